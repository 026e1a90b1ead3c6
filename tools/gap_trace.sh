# dev tool: kernel-to-kernel gaps inside a frame (single-stream), from a rocprofv3 kernel trace
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/gt -o p -- python $R/bench.py --workload dtu --steps 12 --warmup 5 --no-cpu-baseline --no-stages --no-sync-per-frame --single-stream > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/gt/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if 'enerf' in r['Kernel_Name'] and 'pack' not in r['Kernel_Name']]
# frames: split at k_conv0_fused
idx = [i for i, r in enumerate(rows) if 'k_conv0_fused' in r['Kernel_Name']]
fr = rows[idx[-4]:idx[-3]]
t0 = int(fr[0]['Start_Timestamp']); busy = 0; gaps = 0
prev_end = None
for r in fr:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) if prev_end is not None else 0
    gaps += max(gap, 0); busy += e - s
    print(f"{(s - t0)/1e3:8.1f} us  dur {(e - s)/1e3:7.1f}  gap {gap/1e3:6.1f}  {r['Kernel_Name'][:70]}")
    prev_end = e
print("frame kernels", len(fr), "busy us", busy / 1e3, "gaps us", gaps / 1e3, "span us", (int(fr[-1]['End_Timestamp']) - t0) / 1e3)
PY
