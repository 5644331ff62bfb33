set -e
O=$GRAFT_REPO_ROOT/gpurun_out/rtl
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for C in ${CUS:-176 208 224 240}; do
  rm -rf $O/kt$C
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/kt$C -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --global-steps 0 --plugin-steps 0 --psnr-steps 0,0 --trained-seeds= --render-frames 2 --render-decoder-cus $C > $O/bench$C.json 2> $O/bench$C.err
  python3 $GRAFT_REPO_ROOT/tools/render_timeline.py $O/kt$C 30 > $O/timeline$C.txt
  echo "cus $C"; tail -4 $O/timeline$C.txt
  find $O -name "*_kernel_trace.csv" -delete
done
