# tile shape x decoder_cus x tiles-per-replay sweep of the tile pipeline: bash tools/exp_render_tiles.sh "W:H:CUS:P ..."
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/rtiles
mkdir -p $O
LIST=${1:-384:360:224:15 384:360:180:15}
for S in $LIST; do
  IFS=: read W H C P <<< "$S"
  timeout -k 10 200 python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --global-steps 0 --plugin-steps 0 --psnr-steps 0,0 --trained-seeds= --render-frames 3 \
    --render-tile-rays $((W*H)) --render-tile-width $W --render-decoder-cus $C --render-pipeline $P > $O/b_$W-$H-$C-$P.json 2> $O/b_$W-$H-$C-$P.err || echo "failed $S"
  python3 -c "
import json
d=json.loads(open('$O/b_$W-$H-$C-$P.json').read()); print('tile ${W}x$H cus $C P $P:', d['config'].get('render_ms_per_mpix'))" || true
done
