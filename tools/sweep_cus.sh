# focal step against the CU cap of the backward while the next sampler runs beside it (one box, back to back, twice)
set -e
cd $GRAFT_REPO_ROOT
C="--steps 300 --no-cpu-baseline --render-frames 0 --plugin-steps 0 --psnr-steps 0,0 --global-steps 0 --probe-steps 1"
for rep in 1 2; do
for cus in 112 128 144 160 176 192 208; do
  python bench.py $C --prefetch-cus $cus 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('cus', $cus, 'ms', round(d['ms_per_step'],4))"
done
done
