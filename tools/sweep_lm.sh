# focal step against the level from which the level-major lookup gathers one address half at a time (phased forward: 2/3 of the samples)
cd $GRAFT_REPO_ROOT
C="--steps 300 --no-cpu-baseline --render-frames 0 --plugin-steps 0 --psnr-steps 0,0 --global-steps 0 --probe-steps 4"
run() { python bench.py $C "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$*', 'ms', round(d['ms_per_step'],4), 'lm', d['roofline']['avg_kernel_ms'])"; }
for rep in 1 2; do
run --lm-halves-from 5
run --lm-halves-from 6
run --lm-halves-from 7
run --lm-halves-from 8
run --lm-halves-from 6 --lm-split-log2 2
done
