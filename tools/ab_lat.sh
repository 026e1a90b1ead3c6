R=$GRAFT_REPO_ROOT; cp $R/enerf_amd/libenerf_hip.so /tmp/lib_orig.so
for rep in 1 2; do for v in base prio; do cp $R/enerf_amd/_ab/lib_$v.so $R/enerf_amd/libenerf_hip.so; for w in zju lego dtu; do echo "$v $(cd $R && timeout 200 python tools/ab_options.py $w default= 2>&1 | grep latency | head -1)"; done; done; done
cp /tmp/lib_orig.so $R/enerf_amd/libenerf_hip.so
