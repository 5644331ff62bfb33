# round 6: the plugin-surface iteration on a scene trained through it, under rocprofv3 --kernel-trace: per-iteration kernel table of the last 300 ms
# (trace_tail.py) and ONE iteration dispatch by dispatch with queues and gaps (timeline.py --dump-step).  The raw trace stays on the box.
#   bash tools/r06_plugin_timeline.sh [out dir under gpurun_out]
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r06t}
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $GRAFT_REPO_ROOT/tools/plugin_trace.py focal 200 trained > $O/plugin_trained_under_rocprof.txt 2> $O/kt.err
python3 $GRAFT_REPO_ROOT/tools/trace_tail.py $O/kt 300 auto:grad_accumulate > $O/plugin_trained_window.txt
python3 $GRAFT_REPO_ROOT/tools/timeline.py $O/kt --anchor adam_ex --steps 100 --dump-step > $O/plugin_trained_timeline.txt
find $O -name "*_kernel_trace.csv" -delete
find $O -name "*_agent_info.csv" -delete
tail -3 $O/plugin_trained_under_rocprof.txt | cut -c1-400
