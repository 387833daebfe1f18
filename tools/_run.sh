export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do for v in 8 4; do echo "min tiles $v: $(GO2NN_WG_MIN_TILES=$v python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' ')"; done; done
GO2NN_WG_MIN_TILES=4 python -m pytest tests/test_gpu_mlp_tail.py -q -k "ppo_heads_path or linear_group_split" 2>&1 | grep "passed\|failed"
