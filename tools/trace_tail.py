"""Per-kernel GPU time of the LAST `window_ms` of a rocprofv3 --kernel-trace run (the timed window of a script whose set-up dwarfs it):
    python tools/trace_tail.py <dir with *_kernel_trace.csv> <window_ms> [divide_by | auto:<kernel name part>]
auto:<part>: divide by the number of dispatches in the window whose kernel name contains <part> (one per training step, e.g. grad_accumulate)."""
import csv, glob, os, sys
from collections import defaultdict
f = max(glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)  # the newest: gpurun merges earlier runs into the same directory
win = float(sys.argv[2]) * 1e6
div = sys.argv[3] if len(sys.argv) > 3 else "1"
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in csv.DictReader(open(f))]
end = max(r[1] for r in rows)
tail = [r for r in rows if r[0] >= end - win]
div = float(sum(1 for r in tail if div[5:] in r[2])) if div.startswith("auto:") else float(div)
acc, cnt = defaultdict(int), defaultdict(int)
for s, e, k, q in tail:
    acc[k[:90]] += e - s; cnt[k[:90]] += 1
busy = sum(acc.values())
print(f"window {win / 1e6:.1f} ms, {len(tail)} dispatches, kernel time {busy / 1e6:.2f} ms ({busy / win:.2f} of the window; / {div:g}: {busy / 1e6 / div:.4f} ms)")
for k in sorted(acc, key=lambda k: -acc[k])[:28]:
    print(f"{acc[k] / 1e6 / div:9.4f} ms  x{cnt[k] / div:6.2f}  {k}")
