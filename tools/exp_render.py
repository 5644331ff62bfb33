"""Scratch: full-HD render time against tile size, tiles per graph replay and the CU cap of the fused forward."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rays, tw, P, cus in ((8192, 128, 8, 248), (8192, 128, 8, 244), (8192, 128, 6, 248), (8192, 128, 8, 252)):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--render-frames", "3",
                        "--render-pipeline", str(P), "--render-decoder-cus", str(cus), "--render-tile-rays", str(rays), "--render-tile-width", str(tw)],
                       capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])["render"]
        print(f"tile {rays} rays ({tw} wide) P={P:2d} cus={cus}: {d['value']:.1f} ms/Mpix, {d['ms_per_frame']:.1f} ms/frame, mean acc {d['mean_accumulation']:.5f}", flush=True)
    except Exception as e:
        print(rays, tw, P, cus, "failed", e, r.stderr[-800:], flush=True)
