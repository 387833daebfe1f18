cd $GRAFT_REPO_ROOT
O=gpurun_out/r3final; mkdir -p $O
( time timeout 400 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench_time.txt; cat $O/bench_time.txt | tail -3
timeout 120 python tools/termination_check.py "round-2 model (two contact slots per leg, commit 2c3c234)" build/variants/r2model_1wave.so > $O/termination_check_r2model.txt 2>&1
tail -3 $O/termination_check_r2model.txt; python - <<'PY'
import json
for l in open('gpurun_out/r3final/bench.json'):
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['collection_only'], d['roofline']['kernel_ms'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline'].get('thread_sweep_at_4096'))
PY
