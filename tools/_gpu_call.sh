cd $GRAFT_REPO_ROOT
O=gpurun_out/r3m; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_update_golden.py -q --timeout=150 -p no:cacheprovider > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -n "^E " $O/pytest.log | head -8; tail -3 $O/pytest.log
