# full-HD render against (tile shape, CUs of the fused patch kernel): a tile's patches are handed to cus x 8 waves -- how much is lost to the
# last round of patches (patches / waves just above an integer) against tiles whose patch count is a multiple of the waves
cd $GRAFT_REPO_ROOT
C="--steps 20 --no-cpu-baseline --plugin-steps 0 --psnr-steps 0,0 --global-steps 0 --probe-steps 1 --render-frames 3"
run() { python bench.py $C "$@" 2>gpurun_out/sr.err >/dev/null; grep "bench-details" gpurun_out/sr.err | sed 's/^\[bench-details\] //' | python -c "
import sys,json
r=json.loads(sys.stdin.read()).get('render',{})
print('$*'.ljust(90), r.get('value'), r.get('tile'), r.get('tiles_per_replay'))"; }
run
run --render-decoder-cus 225
run --render-tile-rays 230400 --render-tile-width 640 --render-pipeline 9 --render-decoder-cus 225
run --render-tile-rays 230400 --render-tile-width 640 --render-pipeline 9 --render-decoder-cus 224
run --render-tile-rays 207360 --render-tile-width 960 --render-pipeline 10 --render-decoder-cus 224
run --render-tile-rays 207360 --render-tile-width 960 --render-pipeline 10 --render-decoder-cus 216
run --render-tile-rays 691200 --render-tile-width 1920 --render-pipeline 3 --render-decoder-cus 225
run --render-tile-rays 103680 --render-tile-width 480 --render-pipeline 10 --render-decoder-cus 224
