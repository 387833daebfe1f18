export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R && hipcc -O2 -std=c++17 tools/gemm3_bench.cpp -o /tmp/gemm3_bench -ldl 2>/dev/null
python -m pytest tests/test_gpu_mlp_tail.py -x -q -k "below or ppo_heads_path" 2>&1 | tail -3
export BX3=1
BELOW=1 /tmp/gemm3_bench build/variants/libgo2nn_stampsA.so 24576 one i 2 5 2>&1 | grep -v "wave starts"
BELOW=1 /tmp/gemm3_bench build/variants/libgo2nn_stampsA.so 24576 one i 2 5 1 2>&1 | grep -v "wave starts"
python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | cut -c1-250
python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | cut -c1-250
