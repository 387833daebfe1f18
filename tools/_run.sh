export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python -m pytest tests/test_gpu_cts_own.py tests/test_gpu_update_golden.py -q 2>&1 | tail -3
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -k "cts or golden_sequence or config" 2>&1 | tail -3
