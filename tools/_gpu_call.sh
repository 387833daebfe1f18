cd $GRAFT_REPO_ROOT
O=gpurun_out/r3m21; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_update_golden.py tests/test_gpu_mlp_tail.py -x -q --timeout 150 -k "loss or adam or golden or sum_rows or graph" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
run() { n=$1; shift
  env "$@" timeout 200 python bench.py --steps 40 --warmup 20 --no-cpu-baseline > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<PY
import json
for l in open('$O/bench_$n.json'):
    if l.startswith('{'):
        d=json.loads(l); print('$n', round(d['value']/1e6,3), round(d['ms_per_step'],2), round(d['roofline']['kernel_ms']*1e3,1))
PY
}
run autoa A=1
run head0 GO2_FUSED_HEAD=0
run autob A=1
