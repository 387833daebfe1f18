cd $GRAFT_REPO_ROOT
O=gpurun_out/r3fg; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_update_golden.py tests/test_gpu_policy_kernel.py tests/test_gpu_configs.py -q --timeout 150 -x -k "graph or golden or policy or runner or train or rollout or multi_rank or config" > $O/pytest.log 2>&1; echo "rc=$?"; tail -2 $O/pytest.log
timeout 200 python bench.py --steps 40 --warmup 20 --no-cpu-baseline | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']/1e6,3), round(d['ms_per_step'],2), round(d['collection_only']/1e6,2))"
