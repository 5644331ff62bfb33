# kernel timeline of the tile pipeline (tools/render_timeline.py on a rocprofv3 kernel trace of two frames): bash tools/render_timeline.sh [W H CUS P = 384 360 224 15]
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/rtl
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
W=${1:-384}; H=${2:-360}; C=${3:-224}; P=${4:-15}
T=${W}x${H}_${C}_$P
rm -rf $O/kt$T
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/kt$T -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --global-steps 0 --plugin-steps 0 --psnr-steps 0,0 --trained-seeds= --render-frames 2 --render-tile-rays $((W*H)) --render-tile-width $W --render-decoder-cus $C --render-pipeline $P > $O/bench$T.json 2> $O/bench$T.err
python3 $GRAFT_REPO_ROOT/tools/render_timeline.py $O/kt$T 40 > $O/timeline$T.txt
find $O -name "*_kernel_trace.csv" -delete
tail -n 4 $O/timeline$T.txt
