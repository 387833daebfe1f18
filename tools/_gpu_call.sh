cd $GRAFT_REPO_ROOT
O=gpurun_out/r3m25; mkdir -p $O
run() { n=$1; shift
  env "$@" timeout 200 python bench.py --steps 40 --warmup 20 --no-cpu-baseline > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<PY
import json
ok=False
for l in open('$O/bench_$n.json'):
    if l.startswith('{'):
        d=json.loads(l); ok=True; print('$n', round(d['value']/1e6,3), round(d['ms_per_step'],2))
if not ok: print('$n FAILED'); print(open('$O/bench_$n.err').read()[-600:])
PY
}
run none A=1
run serial GO2_FORCE_COLLECTIVES=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511
run overlap GO2_FORCE_COLLECTIVES=1 GO2_OVERLAP_ALLREDUCE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512
run none2 A=1
