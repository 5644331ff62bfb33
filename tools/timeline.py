"""GPU busy / idle time from a rocprofv3 --kernel-trace run.

    python tools/timeline.py <rocprof output dir> [--tail-frac 0.5 | --anchor grad_accumulate --steps 30]

Reads *_kernel_trace.csv (one row per dispatch with start / end timestamps), keeps the last `tail-frac` of the trace -- or, with
--anchor, the span of the last `steps` + 1 dispatches of the kernel whose name contains the anchor (one per training step), i.e.
exactly `steps` training steps -- and prints: wall time of that window, time at least one kernel was running (union of the dispatch
intervals over all streams), the idle remainder, and per kernel the sum of durations -- the gap a hipGraph could close is
the idle remainder, what overlap already hides is (sum of durations - busy)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    frac = float(sys.argv[sys.argv.index("--tail-frac") + 1]) if "--tail-frac" in sys.argv else 0.5
    f = max(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)  # the newest: gpurun merges earlier runs into the same directory
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    if "--anchor" in sys.argv:
        anchor = sys.argv[sys.argv.index("--anchor") + 1]
        steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 30
        marks = [r[1] for r in rows if anchor in r[2]]
        lo, hi = marks[-steps - 1], marks[-1]
        rows = [r for r in rows if lo <= r[0] and r[1] <= hi + 1]
        print(f"{steps} steps between dispatches of '{anchor}': {(hi - lo) / steps / 1e6:.4f} ms/step")
    else:
        cut = t1 - (t1 - t0) * frac
        rows = [r for r in rows if r[0] >= cut]
    wall = max(r[1] for r in rows) - rows[0][0]
    busy, cur_s, cur_e = 0, None, None
    for s, e, _, _ in rows:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    per = defaultdict(lambda: [0, 0])
    for s, e, k, _ in rows:
        k = re.sub(r"\(anonymous namespace\)::", "", k)[:70]
        per[k][0] += e - s
        per[k][1] += 1
    tot = sum(v[0] for v in per.values())
    if "--dump-step" in sys.argv:  # one step, dispatch by dispatch: start offset, duration, gap to the previous end on the same queue
        marks = [i for i, r in enumerate(rows) if (sys.argv[sys.argv.index("--anchor") + 1] if "--anchor" in sys.argv else "adam_ex") in r[2]]
        a, b = marks[len(marks) // 2], marks[len(marks) // 2 + 1]
        base = rows[a][1]
        last_end = {}
        for s_, e_, k_, q_ in rows[a + 1:b + 1]:
            gap = (s_ - last_end[q_]) / 1e3 if q_ in last_end else float("nan")
            last_end[q_] = e_
            name = k_.replace("(anonymous namespace)::", "")[:60]
            print(f"  q{q_:>3} +{(s_ - base) / 1e3:9.1f} us  dur {(e_ - s_) / 1e3:8.1f} us  gap {gap:7.1f}  {name}")
    print(f"window {wall / 1e6:.3f} ms  busy(union) {busy / 1e6:.3f} ms ({busy / wall:.1%})  idle {(wall - busy) / 1e6:.3f} ms  sum of kernel durations {tot / 1e6:.3f} ms")
    for k, (ns, n) in sorted(per.items(), key=lambda kv: -kv[1][0])[:25]:
        print(f"  {ns / 1e6:9.3f} ms  {n:6d} x  {ns / n / 1e3:9.1f} us  {k}")


if __name__ == "__main__":
    main()
