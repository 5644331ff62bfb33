# round-6 profile collection, two gpurun calls (a call lasts 20 minutes at most):
#   bash tools/r06_profiles.sh traces   the bench line + kernel-trace stats (focal / global graph launch, render graph + eager, the plugin-surface step)
#   bash tools/r06_profiles.sh render_traces | pmc_render | busy   after a change to csrc/fused_fwd.hip / x3_device.h's inference parts alone: the bench line + the two render
#                                       traces | the inference kernel's counter passes | the decoder kernels' occupancy passes (then: sync_profiles.py gpurun_out/r06p r06 render)
#   bash tools/r06_profiles.sh pmc      separate --pmc passes (one counter group each; never together with a trace: MI355X guide) for the focal step and
#                                       the global step; pmc_render: the inference kernel's (tiles rendered eagerly: --render-eager); pmc_all: both
# Everything is addressed through $GRAFT_REPO_ROOT (the r03 script mixed it with /root/repo).
set -e
MODE=${1:-traces}
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06p
mkdir -p $O
export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
COMMON="--no-cpu-baseline --render-frames 0 --plugin-steps 0 --psnr-steps 0,0 --trained-seeds= --rays8192-steps 0 --global-x3-steps 0"
RENDER="--steps 2 --warmup 1 --no-cpu-baseline --global-steps 0 --plugin-steps 0 --psnr-steps 0,0 --trained-seeds= --rays8192-steps 0 --global-x3-steps 0 --render-frames 1 --render-eager --no-render-x3"
if [ "$MODE" = "traces" ]; then
  timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err
  grep "bench-details" $O/bench.err | sed 's/^\[bench-details\] //' > $O/bench_details.json
  echo "bench done"
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_focal_graph -- python3 $B --steps 30 $COMMON --global-steps 0 > $O/focal_graph_under_rocprof.json 2>$O/kt_focal_graph.err
  echo "kt focal graph done"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_global_graph -- python3 $B --steps 60 --warmup 60 $COMMON --stage global > $O/global_graph_under_rocprof.json 2>$O/kt_global_graph.err
  echo "kt global graph done"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_render -- python3 $B --steps 5 --warmup 1 --no-cpu-baseline --global-steps 0 --plugin-steps 0 --psnr-steps 0,0 --trained-seeds= --rays8192-steps 0 --render-frames 2 --no-render-x3 > $O/render_under_rocprof.json 2>$O/kt_render.err
  echo "kt render done"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_render_eager -- python3 $B $RENDER > $O/render_eager_under_rocprof.json 2>$O/kt_render_eager.err
  echo "kt render eager done"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_plugin_focal -- python3 $GRAFT_REPO_ROOT/tools/plugin_trace.py focal 100 > $O/plugin_focal_under_rocprof.txt 2>$O/kt_plugin_focal.err
  echo "kt plugin done"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_plugin_trained -- python3 $GRAFT_REPO_ROOT/tools/plugin_trace.py focal 200 trained > $O/plugin_trained_under_rocprof.txt 2>$O/kt_plugin_trained.err
  python3 $GRAFT_REPO_ROOT/tools/trace_tail.py $O/kt_plugin_trained 300 auto:grad_accumulate > $O/plugin_trained_window.txt
  python3 $GRAFT_REPO_ROOT/tools/timeline.py $O/kt_plugin_trained --anchor adam_ex --steps 100 --dump-step > $O/plugin_trained_timeline.txt
  echo "kt plugin trained done"
elif [ "$MODE" = "render_traces" ]; then
  # after a change to csrc/fused_fwd.hip alone: the bench line again and the two render traces (then: pmc_render)
  timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err
  grep "bench-details" $O/bench.err | sed 's/^\[bench-details\] //' > $O/bench_details.json
  echo "bench done"
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_render -- python3 $B --steps 5 --warmup 1 --no-cpu-baseline --global-steps 0 --plugin-steps 0 --psnr-steps 0,0 --trained-seeds= --rays8192-steps 0 --render-frames 2 --no-render-x3 > $O/render_under_rocprof.json 2>$O/kt_render.err
  echo "kt render done"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_render_eager -- python3 $B $RENDER > $O/render_eager_under_rocprof.json 2>$O/kt_render_eager.err
  echo "kt render eager done"
elif [ "$MODE" = "busy" ]; then
  # the decoder kernels' matrix-pipe occupancy in the focal and the global step alone (after a change to csrc/x3_device.h that leaves their code as it was)
  cd /tmp
  C="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; T=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d $O/pmc_focal_$T -- python3 $B --steps 3 --warmup 2 $COMMON --global-steps 0 --launch eager > $O/pmc_focal_$T.json 2>$O/pmc_focal_$T.err || echo "pmc $T failed"
  echo "pmc focal $T done"
  rocprofv3 --pmc $C --output-format csv -d $O/pmc_global_$T -- python3 $B --steps 3 --warmup 58 $COMMON --stage global --launch eager > $O/pmc_global_$T.json 2>$O/pmc_global_$T.err || echo "pmc global $T failed"
  echo "pmc global $T done"
elif [ "$MODE" = "sq_global" ]; then
  # the global step's kernels under the SQ counter groups (what bounds field_bwd_dx_x3_kernel<2, 3> and wgrad_x3_layers_kernel)
  cd /tmp
  for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES" \
           "SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT TCC_MISS" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_WRREQ TCC_EA0_WRREQ_64B"; do
    T=$(echo $C | tr ' ' '_')
    timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $O/sq_global_$T -- python3 $B --steps 3 --warmup 58 $COMMON --stage global --launch eager > $O/sq_global_$T.json 2>$O/sq_global_$T.err || echo "sq global $T failed"
    echo "sq global $T done"
  done
  python3 $GRAFT_REPO_ROOT/tools/prof_summary.py pmc_many $O/sq_global_per_kernel.csv $O/sq_global_*/ || true
else
  cd /tmp
  [ "$MODE" = "pmc_render" ] || for C in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B" "TCC_HIT TCC_MISS" "TCP_TCC_READ_REQ_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    T=$(echo $C | tr ' ' '_')
    rocprofv3 --pmc $C --output-format csv -d $O/pmc_focal_$T -- python3 $B --steps 3 --warmup 2 $COMMON --global-steps 0 --launch eager > $O/pmc_focal_$T.json 2>$O/pmc_focal_$T.err || echo "pmc $T failed"
    echo "pmc focal $T done"
  done
  [ "$MODE" = "pmc_render" ] || for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    T=$(echo $C | tr ' ' '_')
    rocprofv3 --pmc $C --output-format csv -d $O/pmc_global_$T -- python3 $B --steps 3 --warmup 58 $COMMON --stage global --launch eager > $O/pmc_global_$T.json 2>$O/pmc_global_$T.err || echo "pmc global $T failed"
    echo "pmc global $T done"
  done
  [ "$MODE" = "pmc" ] || for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "TCP_TCC_READ_REQ_sum" "TCC_HIT TCC_MISS" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_VMEM"; do
    T=$(echo $C | tr ' ' '_')
    timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $O/pmc_render_$T -- python3 $B $RENDER > $O/pmc_render_$T.json 2>$O/pmc_render_$T.err || echo "pmc render $T failed"
    echo "pmc render $T done"
  done
fi
# keep the merged output small: the raw traces are not needed, the stats and counter tables are
find $O -name "*_kernel_trace.csv" -delete
find $O -name "*_agent_info.csv" -delete
du -sh $O
