# decoder_cus sweep of the tile pipeline (render_ms_per_mpix per setting); bash tools/exp_render_cus.sh "208 216 224"
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/rcus
mkdir -p $O
LIST=${1:-200 208 212 216 220 224 228 232}
for C in $LIST; do
  timeout -k 10 200 python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --global-steps 0 --plugin-steps 0 --psnr-steps 0,0 --trained-seeds= --render-frames 3 --render-decoder-cus $C > $O/bench$C.json 2> $O/bench$C.err
  python3 -c "
import json,sys
d=json.loads(open('$O/bench$C.json').read()); print('cus $C', d['config'].get('render_ms_per_mpix'))"
done
