set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -q -m gpu > gpurun_out/r04_gputests_d.log 2>&1; rc=$?
tail -4 gpurun_out/r04_gputests_d.log
[ $rc -le 1 ] || exit $rc
timeout -k 10 400 python bench.py --psnr-steps 0,0 --trained-seeds= --stage-times > gpurun_out/r04_bench_perf1.json 2> gpurun_out/r04_bench_perf1.err || exit 1
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_bench_perf1.json").read().strip().splitlines()[-1])
c = d["config"]
print("focal ms/step", d["ms_per_step"], "global", c.get("global_ms_per_step"), "render", c.get("render_ms_per_mpix"), "plugin", c.get("plugin_ms_per_step"), c.get("plugin_global_ms_per_step"))
print("roofline", d["roofline"])
PY
grep "^\[stage\]" gpurun_out/r04_bench_perf1.err | head -60
