cd $GRAFT_REPO_ROOT
O=gpurun_out/r3fc; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 200 > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log | cut -c1-200
( time timeout 400 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench_time.txt; tail -3 $O/bench_time.txt
python - <<PY
import json
for l in open('$O/bench.json'):
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['collection_only'], d['roofline']['kernel_ms'], d['roofline']['traffic'], d['roofline'].get('traffic_source'), d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
