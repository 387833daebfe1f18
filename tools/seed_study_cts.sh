#!/bin/bash
# the split-operand learner arithmetic against the fp32-MFMA one on the CTS family's own (no-autograd) path — go2_cts at 4096 envs,
# 3 seeds x {split, fp32 MFMA} x 1000 iterations, student AND teacher reward every 250 iterations.
#   bash tools/seed_study_cts.sh [tag]   -> gpurun_out/<tag>/cts_seeds_<arith>_<seed>.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-cts_seeds}; mkdir -p $O
cd $R
for seed in 1 2 3; do
  GO2_GEMM_SPLIT=1 timeout 300 python tools/train_curve.py go2_cts 1000 250 $seed 2>/dev/null | grep "^it" > $O/cts_seeds_split_$seed.txt
  GO2_GEMM_SPLIT=0 timeout 300 python tools/train_curve.py go2_cts 1000 250 $seed 2>/dev/null | grep "^it" > $O/cts_seeds_fp32_$seed.txt
done
for f in $O/cts_seeds_*.txt; do echo "== $(basename $f)"; cut -c1-200 $f; done
