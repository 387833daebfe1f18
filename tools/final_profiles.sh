#!/bin/bash
# The round's measurement set on one MI355X (run through gpurun; copies to keep go to profiles/ by hand):
#   bash tools/final_profiles.sh [tag]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-final}; mkdir -p $O
cd $R
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
timeout 200 python bench.py --task go2 --no-cpu-baseline > $O/bench_go2.json 2> /dev/null
timeout 200 python bench.py --task go2_cts --steps 50 --no-cpu-baseline > $O/bench_go2_cts.json 2> /dev/null
timeout 200 python bench.py --task go2 --num-envs 32768 --steps 20 --warmup 8 --no-cpu-baseline > $O/bench_go2_32768.json 2> /dev/null
timeout 200 python bench.py --task go2_moe_cts --num-envs 8192 --steps 20 --warmup 8 --no-cpu-baseline > $O/bench_go2_moe_cts_8192.json 2> /dev/null
timeout 200 python bench.py --task go2_moe_cts --num-envs 1024 --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_go2_moe_cts_1024.json 2> /dev/null
timeout 150 python tools/kbench.py 4096 > $O/kbench.txt 2>&1
timeout 150 python tools/kbench.py 4096 rough > $O/kbench_rough.txt 2>&1
timeout 120 python tools/kscale.py 1024 4096 8192 32768 > $O/kscale.txt 2>&1
timeout 120 python tools/termination_check.py "round-3 model (one contact per body group)" > $O/termination_check.txt 2>&1
timeout 100 python tools/policy_bench.py 4096 > $O/policy_bench.txt 2>&1
bash tools/pmc_pass.sh 4096 100 go2_flat > $O/pmc_flat.log 2>&1
bash tools/pmc_pass.sh 4096 100 go2 > $O/pmc_go2.log 2>&1
bash tools/sq_pass.sh 4096 60 go2_flat > $O/sq_flat.log 2>&1
bash tools/sq_pass.sh 4096 60 go2 > $O/sq_go2.log 2>&1
cp $R/gpurun_out/pmc/*.json $R/gpurun_out/pmc/*.csv $O/ 2>/dev/null
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_bench
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python $R/bench.py --steps 30 --warmup 20 --no-cpu-baseline > $O/bench_under_rocprof.json 2> /dev/null
find /tmp/prof_bench -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
cd $R
tail -3 $O/pytest_gpu.log; for f in $O/bench*.json; do echo $f; grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*\|"collection_only": [0-9.]*' $f | tr '\n' ' '; echo; done
tail -3 $O/termination_check.txt; cat $O/kscale.txt | grep -v amdgpu
