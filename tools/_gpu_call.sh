cd $GRAFT_REPO_ROOT
O=gpurun_out/r3m19; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_mlp_tail.py -x -q --timeout 120 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 100 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm_bench.txt
run() { n=$1; shift
  env "$@" timeout 200 python bench.py --steps 40 --warmup 20 --no-cpu-baseline > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<PY
import json
for l in open('$O/bench_$n.json'):
    if l.startswith('{'):
        d=json.loads(l); print('$n', round(d['value']/1e6,3), round(d['ms_per_step'],2), round(d['roofline']['kernel_ms']*1e3,1))
PY
}
for rep in a b; do
run auto$rep A=1
run fl1$rep GO2_MLP_OWN_F=l3l1
run fall$rep GO2_MLP_OWN_F=all
done
