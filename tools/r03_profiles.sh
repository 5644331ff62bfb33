# round-3 profile collection (one gpurun call): the bench line, kernel-trace stats (focal / global graph launch, render, the plugin-surface
# step), separate --pmc passes (one counter group each; never together with a trace: MI355X guide)
set -e
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03p
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err
grep "bench-details" $O/bench.err | sed 's/^\[bench-details\] //' > $O/bench_details.json
echo "bench done"
cd /tmp
COMMON="--no-cpu-baseline --render-frames 0 --plugin-steps 0 --psnr-steps 0,0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_focal_graph -- python3 $B --steps 30 $COMMON --global-steps 0 > $O/focal_graph_under_rocprof.json 2>$O/kt_focal_graph.err
echo "kt focal graph done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_global_graph -- python3 $B --steps 60 --warmup 60 $COMMON --stage global > $O/global_graph_under_rocprof.json 2>$O/kt_global_graph.err
echo "kt global graph done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_render -- python3 $B --steps 5 --warmup 1 --no-cpu-baseline --global-steps 0 --plugin-steps 0 --psnr-steps 0,0 --render-frames 2 > $O/render_under_rocprof.json 2>$O/kt_render.err
echo "kt render done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_plugin_focal -- python3 $GRAFT_REPO_ROOT/tools/plugin_trace.py focal 100 > $O/plugin_focal_under_rocprof.txt 2>$O/kt_plugin_focal.err
echo "kt plugin done"
for C in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B" "TCC_HIT TCC_MISS" "TCP_TCC_READ_REQ_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  T=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d $O/pmc_focal_$T -- python3 $B --steps 3 --warmup 2 $COMMON --global-steps 0 --launch eager > $O/pmc_focal_$T.json 2>$O/pmc_focal_$T.err || echo "pmc $T failed"
  echo "pmc focal $T done"
done
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  T=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d $O/pmc_global_$T -- python3 $B --steps 3 --warmup 58 $COMMON --stage global --launch eager > $O/pmc_global_$T.json 2>$O/pmc_global_$T.err || echo "pmc global $T failed"
  echo "pmc global $T done"
done
# keep the merged output small: the raw traces are not needed, the stats and counter tables are
find $O -name "*_kernel_trace.csv" -delete
find $O -name "*_agent_info.csv" -delete
du -sh $O
