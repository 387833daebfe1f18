export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python -m pytest tests/test_gpu_parity.py -q -s -k "cts_training_graph_vs_eager or two_rank_bench" 2>&1 | grep -v Warning | grep "graph vs eager\|per tensor\|passed\|failed\|Error" | cut -c1-1500
