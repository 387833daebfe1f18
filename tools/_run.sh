export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "== MIX"; DBG_DUMP=/tmp/mix python -m pytest tests/test_gpu_parity.py -q -s -k "cts_training_graph_vs_eager and moe" 2>&1 | grep "DBG policy\|Error\|passed\|failed" | cut -c1-400
echo "== MIX_TORCH"; DBG_MIX_TORCH=1 DBG_DUMP=/tmp/tor python -m pytest tests/test_gpu_parity.py -q -s -k "cts_training_graph_vs_eager and moe" 2>&1 | grep "DBG policy\|Error\|passed\|failed" | cut -c1-400
