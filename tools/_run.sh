export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -q -s -k "heightfield_one_step_parity_vs_oracle or cts_training_graph_vs_eager" 2>&1 | grep -v Warning | grep "parity\|graph vs eager\|after iteration\|Error\|passed\|failed" | cut -c1-400 | head -40
