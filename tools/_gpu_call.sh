cd $GRAFT_REPO_ROOT
O=gpurun_out/r3m27; mkdir -p $O
GO2NN_TILE=62 timeout 300 python -m pytest tests/test_gpu_mlp_tail.py -x -q --timeout 120 -k "linear" > $O/pytest62.log 2>&1; echo "pytest62 rc=$?"; tail -3 $O/pytest62.log
for cfg in "default" "62"; do echo "== GO2NN_TILE=$cfg"; GO2NN_TILE=$cfg timeout 100 python tools/gemm_bench.py 2>&1 | grep -v "amdgpu.ids\|weight grad"; done > $O/tile_sweep.txt 2>&1
cat $O/tile_sweep.txt
