"""Per-STEP kernel table of the training bench from a rocprofv3 kernel trace (tools/gpu_train_stats.sh).

The process also runs warm-up, capture and verification steps (train_graph.GraphedTrainStep: snapshots, eager twins, gradient
comparisons — hundreds of torch kernels that are not part of a training step), so whole-process statistics say little about the
step.  Here the trace is cut at the launches of the level-1 MLP backward kernel (one per step) and only the LAST `n` steps — graph
replays inside the timed region — are aggregated.   usage: train_step_profile.py kernel_trace.csv [n_steps=8] [out.csv]"""
import csv
import sys
from collections import defaultdict


def main():
    path, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 8
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void enerf::k_mlp_bwd<3")]
    assert len(marks) > n + 1, (len(marks), n)
    lo, hi = marks[-n - 1], marks[-1]                       # n whole steps, delimited at the same point of consecutive steps
    sel = rows[lo:hi]
    span_ms = (int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e6 / n
    acc = defaultdict(lambda: [0, 0])
    for r in sel:
        a = acc[r["Kernel_Name"]]
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot = sum(v[1] for v in acc.values())
    is_torch = lambda k: k.startswith(("void at::native", "at::native", "__amd_rocclr", "void (anonymous namespace)::", "void at_cuda_detail")) or "multi_tensor_apply" in k
    lib = sum(v[1] for k, v in acc.items() if "enerf::" in k)
    tor = {k: v for k, v in acc.items() if "enerf::" not in k}
    print(f"steps {n}  wall per step {span_ms:.3f} ms  kernel time per step {tot / 1e6 / n:.3f} ms  launches per step {len(sel) / n:.1f}")
    print(f"library kernels {lib / 1e6 / n:.3f} ms/step in {sum(v[0] for k, v in acc.items() if 'enerf::' in k) / n:.1f} launches; "
          f"other kernels {sum(v[1] for v in tor.values()) / 1e6 / n:.3f} ms/step in {sum(v[0] for v in tor.values()) / n:.1f} launches")
    out = sorted(acc.items(), key=lambda kv: -kv[1][1])
    if len(sys.argv) > 3:
        with open(sys.argv[3], "w") as f:
            w = csv.writer(f)
            w.writerow(["Name", "CallsPerStep", "UsPerStep", "AvgUs", "Percent"])
            for k, (c, t) in out:
                w.writerow([k, round(c / n, 2), round(t / 1e3 / n, 2), round(t / 1e3 / c, 2), round(100.0 * t / tot, 3)])
    print("---- kernels that are NOT the library's (per step) ----")
    for k, (c, t) in sorted(tor.items(), key=lambda kv: -kv[1][1]):
        print(f"{k[:130]:130s} calls {c / n:6.1f} us {t / 1e3 / n:8.1f}")
    print("---- top library kernels (per step) ----")
    for k, (c, t) in [kv for kv in out if "enerf::" in kv[0]][:30]:
        print(f"{k[:110]:110s} calls {c / n:6.1f} us {t / 1e3 / n:8.1f}")


if __name__ == "__main__":
    main()
