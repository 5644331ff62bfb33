"""Scratch: sweep one bench flag."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flag, vals = sys.argv[1], sys.argv[2:]
for v in vals:
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--render-frames", "0", "--steps", "40", flag, v], capture_output=True, text=True)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    print(f"{flag}={v}: {d['value']:.0f} rays/s, {d['ms_per_step']:.3f} ms/step, {d['roofline']['kernel_ms']}", flush=True)
