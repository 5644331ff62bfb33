set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests -q -m gpu > gpurun_out/r04_gputests_c.log 2>&1; rc=$?
tail -4 gpurun_out/r04_gputests_c.log
[ $rc -le 1 ] || exit $rc
R="--steps 5 --warmup 2 --no-cpu-baseline --global-steps 0 --plugin-steps 0 --psnr-steps 0,0 --trained-seeds= --render-frames 2"
GFN_RENDER_LEVELS=2 timeout -k 10 200 python bench.py $R > gpurun_out/r04_render_l2.json 2> gpurun_out/r04_render_l2.err || exit 1
GFN_RENDER_LEVELS=4 timeout -k 10 200 python bench.py $R > gpurun_out/r04_render_l4.json 2> gpurun_out/r04_render_l4.err || exit 1
python - <<'PY'
import json
for t in ("l2", "l4"):
    d = json.loads(open(f"gpurun_out/r04_render_{t}.json").read().strip().splitlines()[-1])
    print(t, "render_ms_per_mpix", d["config"].get("render_ms_per_mpix"), "ms/step", d["ms_per_step"])
PY
timeout -k 10 400 python tools/exp_psnr_var.py "dict(leg='fast', seeds=list(range(8)), detail=0.5, lr_hold=1000, steps_global=3000)" "dict(leg='fast', seeds=list(range(8)), detail=0.5, steps_global=3000, lr_final=0.1)" > gpurun_out/r04_psnr_var3.txt 2> gpurun_out/r04_psnr_var3.err || exit 1
echo "variants done"
timeout -k 10 600 bash tools/r04_render_pmc.sh > gpurun_out/r04_render_pmc.log 2>&1 || exit 1
tail -3 gpurun_out/r04_render_pmc.log
