export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
python -m pytest tests/test_gpu_mlp_tail.py -x -q 2>&1 | tail -3 > $O/r6b_pytest_mlp_tail.log
rm -rf /tmp/prof
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o b -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline > $O/r6b_bench_rocprof.json 2>/dev/null
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} $O/r6b_bench_kernel_stats.csv \;
python $R/tools/trace_timeline.py $(find /tmp/prof -name "*kernel_trace.csv" | head -1) > $O/r6b_timeline.txt 2>&1
cd $R; python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/r6b_bench.json 2>/dev/null
cat $O/r6b_pytest_mlp_tail.log
sed -n 6,24p $O/r6b_timeline.txt | cut -c1-150
cut -c1-250 $O/r6b_bench.json
