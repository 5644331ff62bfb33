"""Scratch: full-HD render time against tiles per graph replay and the CU cap of the fused forward."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for P, cus in ((1, 256), (4, 224), (8, 224), (8, 240), (8, 208), (8, 192), (16, 224)):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--render-frames", "3",
                        "--render-pipeline", str(P), "--render-decoder-cus", str(cus)], capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])["render"]
        print(f"P={P:2d} cus={cus}: {d['value']:.1f} ms/Mpix, {d['ms_per_frame']:.1f} ms/frame, mean acc {d['mean_accumulation']:.5f}", flush=True)
    except Exception as e:
        print(P, cus, "failed", e, r.stderr[-1500:], flush=True)
