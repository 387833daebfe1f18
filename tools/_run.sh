export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python -m pytest tests/test_gpu_mlp_tail.py tests/test_gpu_update_golden.py tests/test_gpu_cts_own.py -x -q 2>&1 | tail -3
for v in 0 8192 0 8192; do echo "side rows $v: $(GO2_WGRAD_SIDE_ROWS=$v python bench.py --task go2_moe_cts --num-envs 1024 --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' ')"; done
for v in 0 30000; do echo "headline, side rows $v: $(GO2_WGRAD_SIDE_ROWS=$v python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' ')"; done
