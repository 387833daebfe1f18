#!/bin/bash
# The round's measurement set on one MI355X (run through gpurun; copies to keep go to profiles/ by hand):
#   bash tools/final_profiles.sh [tag]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-final}; mkdir -p $O
cd $R
python __graft_entry__.py > $O/build.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 1500 python -m pytest tests -q -m gpu -s -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
GO2_GEMM_SPLIT=0 timeout 200 python bench.py --no-cpu-baseline > $O/bench_fp32_mfma_gemms.json 2> /dev/null
GO2_FUSED_MLP=0 timeout 200 python bench.py --steps 30 --no-cpu-baseline > $O/bench_reference_formulation.json 2> /dev/null
timeout 200 python bench.py --task go2 --no-cpu-baseline > $O/bench_go2.json 2> /dev/null
timeout 200 python bench.py --task go2_cts --steps 50 --no-cpu-baseline > $O/bench_go2_cts.json 2> /dev/null
GO2_GEMM_SPLIT=0 timeout 200 python bench.py --task go2_cts --steps 30 --no-cpu-baseline > $O/bench_go2_cts_fp32_mfma_gemms.json 2> /dev/null
GO2_WGRAD_BELOW=0 timeout 200 python bench.py --no-cpu-baseline > $O/bench_wgrad_below0.json 2> /dev/null
timeout 200 python bench.py --task go2 --num-envs 32768 --steps 20 --warmup 8 --no-cpu-baseline > $O/bench_go2_32768.json 2> /dev/null
timeout 200 python bench.py --num-envs 8192 --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_go2_flat_8192.json 2> /dev/null
timeout 200 python bench.py --task go2_moe_cts --num-envs 8192 --steps 20 --warmup 8 --no-cpu-baseline > $O/bench_go2_moe_cts_8192.json 2> /dev/null
timeout 200 python bench.py --task go2_moe_cts --num-envs 1024 --steps 30 --warmup 8 --no-cpu-baseline > $O/bench_go2_moe_cts_1024.json 2> /dev/null
if [ -z "$SKIP_STEP_KERNEL" ]; then          # (SKIP_STEP_KERNEL=1: libgo2sim_hip.so is unchanged since the last full set — its kbench / PMC / SQ files stay valid, keyed by the library's hash)
timeout 150 python tools/kbench.py 4096 > $O/kbench.txt 2>&1
timeout 150 python tools/kbench.py 4096 rough > $O/kbench_rough.txt 2>&1
fi
{ timeout 100 python tools/policy_bench.py 4096 | tail -1; timeout 100 python tools/policy_bench.py 8192 | tail -1; GO2_GEMM_SPLIT=0 timeout 100 python tools/policy_bench.py 4096 | tail -1; GO2_GEMM_SPLIT=0 timeout 100 python tools/policy_bench.py 8192 | tail -1; } > $O/policy_bench.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -cuid=go2nn -DGO2NN_STAMPS -o build/variants/libgo2nn_stamps.so go2_rl_gym_amd/csrc/go2nn_impl.cpp 2>/dev/null
timeout 100 python tools/policy_bench.py 4096 --stamps 2>&1 | tail -5 >> $O/policy_bench.txt
hipcc -O2 -std=c++17 tools/gemm3_bench.cpp -o /tmp/gemm3_bench -ldl 2>/dev/null
export GEMM3_BENCH=/tmp/gemm3_bench
if [ -z "$SKIP_STEP_KERNEL" ]; then          # (the learner's GEMM kernels likewise)
BX3=1 bash tools/gemm3_pmc.sh bx3_fwd_L2 f 2 10 > /dev/null 2>&1
BX3=1 bash tools/gemm3_pmc.sh bx3_igrad_L2 i 2 10 > /dev/null 2>&1
BX3=1 bash tools/gemm3_pmc.sh bx3_wgrad_L2 w 2 10 > /dev/null 2>&1
BX3=1 BELOW=1 bash tools/gemm3_pmc.sh bx3_igrad_below_L2 i 2 10 > /dev/null 2>&1          # round 6: the input gradient that also leaves the first layers' weight gradients
BX3=1 /tmp/gemm3_bench go2_rl_gym_amd/libgo2nn_hip.so 24576 time > $O/gemm3_time.txt 2>&1
fi
PASSES="1 3" bash tools/policy_pmc.sh mlp3 4096 > $O/policy_pmc.log 2>&1
if [ -z "$SKIP_STEP_KERNEL" ]; then
bash tools/pmc_pass.sh 4096 100 go2_flat > $O/pmc_flat.log 2>&1
bash tools/pmc_pass.sh 4096 100 go2 > $O/pmc_go2.log 2>&1
bash tools/sq_pass.sh 4096 60 go2_flat > $O/sq_flat.log 2>&1
bash tools/sq_pass.sh 4096 60 go2 > $O/sq_go2.log 2>&1
fi
cp $R/gpurun_out/pmc/*.json $R/gpurun_out/pmc/*.csv $O/ 2>/dev/null
cd /tmp && export TMPDIR=/tmp
for spec in "go2_flat 4096 30 20" "go2_cts 4096 12 8" "go2_moe_cts 8192 8 8" "go2_moe_cts 1024 12 8"; do
  set -- $spec; task=$1; n=$2; steps=$3; warm=$4
  rm -rf /tmp/prof_$task
  sfx=""; [ "$n" != "4096" ] && [ "$task" != "go2_moe_cts" -o "$n" != "8192" ] && sfx="_$n"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$task -o b -- python $R/bench.py --task $task --num-envs $n --steps $steps --warmup $warm --no-cpu-baseline > $O/bench_${task}${sfx}_under_rocprof.json 2> /dev/null
  find /tmp/prof_$task -name "*kernel_stats.csv" -exec cp {} $O/bench_${task}${sfx}_kernel_stats.csv \;
  python $R/tools/trace_timeline.py $(find /tmp/prof_$task -name "*kernel_trace.csv" | head -1) > $O/${task}${sfx}_timeline.txt 2>&1
done
cd $R
tail -3 $O/pytest_gpu.log; for f in $O/bench*.json; do echo $f; grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*\|"collection_only": [0-9.]*' $f | tr '\n' ' '; echo; done
