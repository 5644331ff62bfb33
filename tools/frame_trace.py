"""Print the dispatches of the LAST rendered frame of a rocprofv3 --kernel-trace run of bench.py (render sub-record): start offset from
the frame's first patch kernel's tile, duration, queue, kernel.   python tools/frame_trace.py <rocprof dir> [tiles per frame = 15]"""
import csv, glob, os, sys
d = sys.argv[1]
tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 15
f = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in csv.DictReader(open(f))]
rows.sort()
marks = [r[0] for r in rows if "fused_render_patch_kernel" in r[2]]
lo = marks[-tiles] - 4_000_000
hi = max(r[1] for r in rows)
busy = 0
for s, e, k, q in rows:
    if lo <= s <= hi and (e - s) > 20_000:
        print(f"{(s - lo) / 1e3:10.1f} us  +{(e - s) / 1e3:9.1f} us  q{q:>3s}  {k[:70]}")
