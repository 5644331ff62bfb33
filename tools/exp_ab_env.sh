# A/B of the focal headline under an environment variable: bash tools/exp_ab_env.sh VAR "v1 v2 ..." [extra bench args]
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/ab
mkdir -p $O
VAR=$1; VALS=$2; shift 2
for rep in 1 2; do
for V in $VALS; do
  env $VAR=$V timeout -k 10 200 python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --global-steps 0 --plugin-steps 0 --psnr-steps 0,0 --trained-seeds= --render-frames 0 "$@" > $O/b_$V.json 2> $O/b_$V.err
  python3 -c "
import json
d=json.loads(open('$O/b_$V.json').read()); print('$VAR=$V ms_per_step', d['ms_per_step'])"
done
done
