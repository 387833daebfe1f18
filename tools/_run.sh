export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
python -m pytest tests/test_gpu_mlp_tail.py -x -q 2>&1 | tail -5 > $O/r6a_pytest_mlp_tail.log
for v in 1 0; do
rm -rf /tmp/prof
cd /tmp && GO2_WGRAD_BELOW=$v timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o b -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline > $O/r6a_bench_rocprof_below$v.json 2>/dev/null
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} $O/r6a_bench_kernel_stats_below$v.csv \;
python $R/tools/trace_timeline.py $(find /tmp/prof -name "*kernel_trace.csv" | head -1) > $O/r6a_timeline_below$v.txt 2>&1
cd $R; GO2_WGRAD_BELOW=$v python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/r6a_bench_below$v.json 2>/dev/null
done
cat $O/r6a_pytest_mlp_tail.log
sed -n 6,24p $O/r6a_timeline_below1.txt | cut -c1-150
cut -c1-250 $O/r6a_bench_below1.json; cut -c1-250 $O/r6a_bench_below0.json
