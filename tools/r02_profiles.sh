# round-2 profile collection (one gpurun call): bench lines, kernel-trace stats (graph and eager launch), separate --pmc passes
set -e
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02p
mkdir -p $O
export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench done"
cd /tmp
COMMON="--no-cpu-baseline --render-frames 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_focal_graph -- python3 $B --steps 30 $COMMON --global-steps 0 > $O/focal_graph_under_rocprof.json 2>$O/kt_focal_graph.err
echo "kt focal graph done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_focal_eager -- python3 $B --steps 30 $COMMON --global-steps 0 --launch eager > $O/focal_eager_under_rocprof.json 2>$O/kt_focal_eager.err
echo "kt focal eager done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_global_graph -- python3 $B --steps 60 --warmup 60 $COMMON --stage global > $O/global_graph_under_rocprof.json 2>$O/kt_global_graph.err
echo "kt global graph done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_global_eager -- python3 $B --steps 60 --warmup 60 $COMMON --stage global --launch eager > $O/global_eager_under_rocprof.json 2>$O/kt_global_eager.err
echo "kt global eager done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_render -- python3 $B --steps 5 --warmup 1 --no-cpu-baseline --global-steps 0 --render-frames 2 > $O/render_under_rocprof.json 2>$O/kt_render.err
echo "kt render done"
for C in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B" "TCC_HIT TCC_MISS" "TCP_TCC_READ_REQ_sum"; do
  T=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d $O/pmc_focal_$T -- python3 $B --steps 3 --warmup 2 $COMMON --global-steps 0 --launch eager > $O/pmc_focal_$T.json 2>$O/pmc_focal_$T.err || echo "pmc $T failed"
  echo "pmc focal $T done"
done
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $O/pmc_global_$C -- python3 $B --steps 3 --warmup 58 $COMMON --stage global --launch eager > $O/pmc_global_$C.json 2>$O/pmc_global_$C.err || echo "pmc global $C failed"
  echo "pmc global $C done"
done
for C in FETCH_SIZE "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B"; do
  T=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d $O/calib_$T -- $GRAFT_REPO_ROOT/tools/fetch_calib.bin > $O/calib_$T.txt 2>$O/calib_$T.err || echo "calib $T failed"
done
echo "calibration done"
# keep the merged output small: the raw traces are not needed, the stats and counter tables are
find $O -name "*_kernel_trace.csv" -delete
find $O -name "*_agent_info.csv" -delete
du -sh $O
