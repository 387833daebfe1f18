cd $GRAFT_REPO_ROOT
O=gpurun_out/r3k; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_update_golden.py -q --timeout=150 -p no:cacheprovider > $O/pytest_a.log 2>&1; echo "rc=$?" >> $O/pytest_a.log
head -c 300 $O/pytest_a.log | head -3; tail -2 $O/pytest_a.log
