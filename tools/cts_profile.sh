#!/bin/bash
# Kernel statistics + timeline of the CTS / MoE-CTS workloads (BASELINE configs 3 and 5's per-GPU shape) under rocprofv3, on one MI355X through gpurun:
#   bash tools/cts_profile.sh [tag]      -> gpurun_out/<tag>/{bench_go2_cts.json, go2_cts_kernel_stats.csv, go2_cts_timeline.txt, ...moe...}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-cts}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for spec in "go2_cts 4096 12 8" "go2_moe_cts 8192 8 8"; do
  set -- $spec; task=$1; n=$2; steps=$3; warm=$4
  rm -rf /tmp/prof_$task
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$task -o b -- python $R/bench.py --task $task --num-envs $n --steps $steps --warmup $warm --no-cpu-baseline > $O/bench_${task}_under_rocprof.json 2> $O/bench_${task}.err
  find /tmp/prof_$task -name "*kernel_stats.csv" -exec cp {} $O/${task}_kernel_stats.csv \;
  python $R/tools/trace_timeline.py $(find /tmp/prof_$task -name "*kernel_trace.csv" | head -1) > $O/${task}_timeline.txt 2>&1
  timeout 300 python $R/bench.py --task $task --num-envs $n --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_${task}.json 2>> $O/bench_${task}.err
done
for f in $O/bench*.json; do echo $f; grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"collection_only": [0-9.]*' $f | tr '\n' ' '; echo; done
head -25 $O/go2_cts_kernel_stats.csv | cut -c1-200
