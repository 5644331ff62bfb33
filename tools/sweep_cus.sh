# step time against the CU cap of the backward while the next sampler runs beside it (one box, back to back, twice): focal, then global
cd $GRAFT_REPO_ROOT
C="--steps 300 --no-cpu-baseline --render-frames 0 --plugin-steps 0 --psnr-steps 0,0 --probe-steps 1"
for rep in 1 2; do
for cus in 144 152 160 168 176; do
  python bench.py $C --global-steps 0 --prefetch-cus $cus 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('focal cus', $cus, 'ms', round(d['ms_per_step'],4))"
done
for cus in 184 192 200 208 216 224; do
  python bench.py $C --stage global --warmup 60 --prefetch-cus $cus 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('global cus', $cus, 'ms', round(d['ms_per_step'],4))"
done
done
