cd $GRAFT_REPO_ROOT
O=gpurun_out/r3fd; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 8 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python tools/trace_timeline.py /tmp/kt/kt_kernel_trace.csv > $O/timeline.txt 2>&1; grep -n "update:\|one mini-batch\|one rollout" $O/timeline.txt
