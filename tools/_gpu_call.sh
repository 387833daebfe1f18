cd $GRAFT_REPO_ROOT
O=gpurun_out/r3p; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -q --timeout=200 -p no:cacheprovider -k "multi_rank or two_rank" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
for ov in 1 0; do GO2_FORCE_COLLECTIVES=1 GO2_OVERLAP_ALLREDUCE=$ov timeout 150 python bench.py --steps 40 --warmup 20 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*\|"all_reduce": [0-9.]*' | tr '\n' ' '; echo " (forced collectives, overlap=$ov)"; done > $O/bench_collectives.txt
timeout 150 python bench.py --steps 40 --warmup 20 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*' >> $O/bench_collectives.txt
grep -n "^E " $O/pytest.log | head -5; tail -3 $O/pytest.log; cat $O/bench_collectives.txt
