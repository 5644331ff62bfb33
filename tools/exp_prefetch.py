"""Scratch: step time against the CU cap of the fused backward while the prefetched sampler runs beside it."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for cus in (192, 208, 224, 232, 240, 256):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--render-frames", "0", "--steps", "40", "--prefetch-cus", str(cus)],
                       capture_output=True, text=True)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    print(f"prefetch_cus={cus}: {d['value']:.0f} rays/s, {d['ms_per_step']:.3f} ms/step, {d['roofline']['kernel_ms']}", flush=True)
