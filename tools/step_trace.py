"""Print the dispatches of one training step (between two lm_lookup<true,false> launches in the middle of a rocprofv3 --kernel-trace
run): start offset, duration, queue, kernel -- to see what overlaps what.   python tools/step_trace.py <rocprof dir> [index from the end]"""
import csv, glob, os, sys
d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 20
f = max(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)  # the newest: gpurun merges earlier runs into the same directory
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in csv.DictReader(open(f))]
rows.sort()
anchor = sys.argv[3] if len(sys.argv) > 3 else "lm_lookup_kernel<true, false>"
marks = [r[0] for r in rows if anchor in r[2]]
lo, hi = marks[-back - 1], marks[-back]
for s, e, k, q in rows:
    if lo <= s < hi:
        print(f"{(s - lo) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  q{q:>3s}  {k[:80]}")
print(f"step {(hi - lo) / 1e3:.1f} us")
