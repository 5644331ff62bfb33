set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R="--steps 5 --warmup 2 --no-cpu-baseline --global-steps 0 --plugin-steps 0 --psnr-steps 0,0 --trained-seeds= --render-frames 3"
for C in 208 216 224 232 240; do
  timeout -k 10 200 python bench.py $R --render-decoder-cus $C > gpurun_out/r04_render_cus$C.json 2> gpurun_out/r04_render_cus$C.err || exit 1
  python - <<PY
import json
d = json.loads(open("gpurun_out/r04_render_cus$C.json").read().strip().splitlines()[-1])
print("decoder_cus $C render_ms_per_mpix", d["config"].get("render_ms_per_mpix"), "samples/s", d["config"].get("render_samples_per_sec"))
PY
done
grep -o '"roofline": {[^}]*}' gpurun_out/r04_render_cus224.err | tail -1
timeout -k 10 900 python tools/psnr_study.py gpurun_out/r04_psnr_study_p0.json 16 > gpurun_out/r04_psnr_study_p0.log 2>&1 || exit 1
tail -2 gpurun_out/r04_psnr_study_p0.log
