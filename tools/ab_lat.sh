# usage: bash tools/ab_lat.sh "wl1 wl2" v1 v2 ...   (variants built by tools/build_variant.py)
R=$GRAFT_REPO_ROOT; cp $R/enerf_amd/libenerf_hip.so /tmp/lib_orig.so
WLS=$1; shift
for rep in 1 2; do for v in "$@"; do cp $R/enerf_amd/_ab/lib_$v.so $R/enerf_amd/libenerf_hip.so; for w in $WLS; do echo "$v $(cd $R && timeout 200 python tools/ab_options.py $w default= 2>&1 | grep latency | head -1)"; done; done; done
cp /tmp/lib_orig.so $R/enerf_amd/libenerf_hip.so
