#!/bin/bash
# GPU idle time between frames under the per-frame-sync protocol: kernel trace of bench.py's timed loop -> per-frame busy span and gap.
export TMPDIR=/tmp
TAG=${1:-r03_gap}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pg -o p -- python $R/bench.py --workload ${WL:-dtu} --no-secondary --steps 100 --warmup 10 --no-cpu-baseline --no-stages > $O/bench.log 2>&1
F=$(find /tmp/pg -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$F")))
rows=[r for r in rows if "enerf" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# frames: split at the first kernel of the FeatureNet trunk
frames=[]; cur=[]
for r in rows:
    if "k_conv0_fused" in r["Kernel_Name"] and cur:
        frames.append(cur); cur=[]
    cur.append(r)
frames.append(cur)
frames=frames[-150:]
span=[(int(f[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in f)) for f in frames]
busy=[e-s for s,e in span]
gap=[span[i+1][0]-span[i][1] for i in range(len(span)-1)]
import statistics as st
print("frames", len(frames), "kernels/frame", st.mean(len(f) for f in frames))
print("GPU span per frame us: mean %.1f p50 %.1f" % (st.mean(busy)/1e3, st.median(busy)/1e3))
print("GPU idle between frames us: mean %.1f p50 %.1f min %.1f" % (st.mean(gap)/1e3, st.median(gap)/1e3, min(gap)/1e3))
# one steady-state frame as a timeline: start offset, duration, stream/queue, kernel
f=frames[len(frames)//2]; t0=int(f[0]["Start_Timestamp"])
keys=[k for k in f[0].keys() if "Stream" in k or "Queue" in k]
print("columns:", list(f[0].keys()))
for r in f:
    s_,e_=int(r["Start_Timestamp"])-t0,int(r["End_Timestamp"])-t0
    print("%8.1f %7.1f  %s  %s" % (s_/1e3,(e_-s_)/1e3," ".join(str(r[k]) for k in keys), r["Kernel_Name"].replace("void enerf::","").replace("enerf::","").split("(")[0][:60]))
# time with NO kernel running inside the frame
ev=sorted([(int(r["Start_Timestamp"]),1) for r in f]+[(int(r["End_Timestamp"]),-1) for r in f])
idle=0; run=0; last=None
for t,d in ev:
    if run==0 and last is not None: idle+=t-last
    run+=d; last=t
print("in-frame time with no kernel running: %.1f us" % (idle/1e3))
PY
tail -1 $O/bench.log | cut -c1-200
