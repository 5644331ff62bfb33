cd $GRAFT_REPO_ROOT
A="--steps 10 --no-cpu-baseline --global-steps 0 --psnr-steps 0,0 --trained-seeds="
for P in 1 0 1 0; do
  for R in 0 1; do
  GFN_SIDE_STREAM_PROBE=$P timeout -k 10 200 python bench.py $A --render-frames $R 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('probe $P render_frames $R: headline', d['ms_per_step'], 'plugin', d['config'].get('plugin_ms_per_step'), d['config'].get('plugin_global_ms_per_step'), 'render', d['config'].get('render_ms_per_mpix'))"
  done
done
