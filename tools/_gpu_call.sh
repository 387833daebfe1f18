cd $GRAFT_REPO_ROOT
O=gpurun_out/r3fe; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_mlp_tail.py tests/test_gpu_update_golden.py tests/test_gpu_policy_kernel.py -q --timeout 150 > $O/pytest.log 2>&1; echo "rc=$?"; tail -2 $O/pytest.log
timeout 200 python bench.py --steps 40 --warmup 20 --no-cpu-baseline | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']/1e6,3), round(d['ms_per_step'],2), d['roofline']['traffic'])"
