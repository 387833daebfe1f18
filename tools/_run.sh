export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -q -s -k "cts_training_graph_vs_eager" 2>&1 | grep -v Warning | grep "graph vs eager\|after iteration\|Error\|passed\|failed\|^E " | cut -c1-600 | head -30
