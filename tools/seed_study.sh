#!/bin/bash
# is the split-operand learner arithmetic (3 x bf16 planes, six MFMA terms; -1/3 ulp accumulate bias) visible in training?
# 5 seeds x {split, fp32 MFMA} x 1500 iterations on go2_flat at 4096 envs, reward every 250 iterations; one 10 000-iteration pair (seed 1).
#   bash tools/seed_study.sh [tag]   -> gpurun_out/<tag>/seeds_<arith>_<seed>.txt, long_<arith>.txt, cts_*.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-seeds}; mkdir -p $O
cd $R
python __graft_entry__.py > $O/build.log 2>&1
for seed in 1 2 3 4 5; do
  GO2_GEMM_SPLIT=1 timeout 200 python tools/train_curve.py go2_flat 1500 250 $seed 2>/dev/null | grep "^it" > $O/seeds_split_$seed.txt
  GO2_GEMM_SPLIT=0 timeout 200 python tools/train_curve.py go2_flat 1500 250 $seed 2>/dev/null | grep "^it" > $O/seeds_fp32_$seed.txt
done
timeout 200 python tools/train_curve.py go2_cts 300 100 1 2>/dev/null | grep "^it" > $O/cts_go2_cts.txt
timeout 300 python tools/train_curve.py go2_moe_cts 300 100 1 2>/dev/null | grep "^it" > $O/cts_go2_moe_cts.txt
GO2_GEMM_SPLIT=1 timeout 600 python tools/train_curve.py go2_flat 10000 1000 1 2>/dev/null | grep "^it" > $O/long_split.txt
GO2_GEMM_SPLIT=0 timeout 600 python tools/train_curve.py go2_flat 10000 1000 1 2>/dev/null | grep "^it" > $O/long_fp32.txt
tail -n 1 $O/seeds_*.txt $O/long_*.txt $O/cts_*.txt | cut -c1-120
