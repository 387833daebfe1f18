cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_gpu_mlp_tail.py -x -q --timeout 120 -k "linear" 2>&1 | tail -2
timeout 100 python tools/gemm_bench.py 2>&1 | grep -v "amdgpu.ids\|weight grad"
