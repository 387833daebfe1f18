export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python -m pytest tests -m gpu -q 2>&1 | grep "passed\|failed\|FAILED" 
python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"collection_only": [0-9.]*' | tr '\n' ' '; echo
python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"collection_only": [0-9.]*' | tr '\n' ' '; echo
python bench.py --task go2_moe_cts --num-envs 1024 --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo
