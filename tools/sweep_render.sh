# full-HD render time against the tile size (15 / 6 / 4 tiles per frame): how much of a tile is the tail of its longest patches
cd $GRAFT_REPO_ROOT
C="--steps 20 --no-cpu-baseline --plugin-steps 0 --psnr-steps 0,0 --global-steps 0 --probe-steps 1 --render-frames 3"
run() { python bench.py $C "$@" 2>gpurun_out/sr.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$*', 'ms/Mpix', d['config'].get('render_ms_per_mpix'))"; grep -o '"render": {[^}]*}' gpurun_out/sr.err | cut -c1-400; }
run
run --render-tile-rays 345600 --render-tile-width 640 --render-pipeline 6
run --render-tile-rays 518400 --render-tile-width 960 --render-pipeline 4
