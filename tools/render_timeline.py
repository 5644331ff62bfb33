#!/usr/bin/env python3
"""Timeline of the tile pipeline from a rocprofv3 kernel trace of `bench.py --render-frames N`: for the last frame's tiles, when the sampler's
kernels (trace / march / fill, the 32 free CUs) and the inference kernel (224 CUs) start and end, which of the two streams is the critical one
and how long each waits for the other.
    python tools/render_timeline.py <dir with *_kernel_trace.csv>  [n_tiles_back=30]"""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    path = sorted(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True))[-1]
    rows = []
    with open(path) as f:
        rd = csv.DictReader(f)
        cols = {c.lower(): c for c in rd.fieldnames}
        need = ("kernel_name", "start_timestamp", "end_timestamp")
        if any(c not in cols for c in need):
            print("unexpected columns:", rd.fieldnames)
            return
        for r0 in rd:
            r = {k.lower(): v for k, v in r0.items()}
            n = r["kernel_name"]
            k = ("render" if "fused_render_patch" in n else "march" if "march_group" in n else "trace" if "trace_group" in n
                 else "fill" if "fill_from_rec" in n else "depth" if "depth_finalize" in n else None)
            if k:
                rows.append((int(r["start_timestamp"]), int(r["end_timestamp"]), k, int(r.get("stream_id", 0) or 0), int(r.get("grid_size", r.get("grid_size_x", 0)) or 0)))
    rows.sort()
    renders = [r for r in rows if r[2] == "render"]
    if not renders:
        print("no render kernels in", path)
        return
    first = renders[-min(back, len(renders))]
    t0 = first[0]
    sel = [r for r in rows if r[1] >= t0 - 12_000_000]
    print(f"{path}: {len(renders)} inference launches; times in ms from the start of inference launch -{min(back, len(renders))}")
    print("kind    stream  start     end       dur     grid")
    for s, e, k, st, g in sel:
        print(f"{k:7s} {st:6d} {(s - t0) / 1e6:9.3f} {(e - t0) / 1e6:9.3f} {(e - s) / 1e6:7.3f} {g}")
    # cadence and gaps of the inference stream
    last = renders[-min(back, len(renders)):]
    gaps = [(b[0] - a[1]) / 1e6 for a, b in zip(last, last[1:])]
    durs = [(r[1] - r[0]) / 1e6 for r in last]
    print(f"inference: mean duration {sum(durs) / len(durs):.3f} ms, mean gap to the next launch {sum(gaps) / max(len(gaps), 1):.3f} ms, largest gap {max(gaps or [0]):.3f}")
    for kind in ("trace", "march", "fill"):
        ds = [(e - s) / 1e6 for s, e, k, _, g in sel if k == kind and g > 100_000]
        if ds:
            print(f"{kind}: {len(ds)} tile launches, mean {sum(ds) / len(ds):.3f} ms, max {max(ds):.3f}")


if __name__ == "__main__":
    main()
