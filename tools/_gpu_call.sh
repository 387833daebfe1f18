cd $GRAFT_REPO_ROOT
O=gpurun_out/r3m18; mkdir -p $O
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
run inone$rep GO2_MLP_OWN_I=none
run fl3a1$rep GO2_MLP_OWN_F=l3a1
run wk256$rep GO2_MLP_OWN_W=k256
run wall$rep GO2_MLP_OWN_W=all
done
