cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ff; mkdir -p $O
for n in 1024 2048 4096 8192 16384; do
  st=30; [ $n -ge 8192 ] && st=12
  timeout 150 python bench.py --num-envs $n --steps $st --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('num_envs %6d: %.3f M env-steps/s, %.2f ms per iteration, collection only %.1f M, step kernel %.1f us' % ($n, d['value']/1e6, d['ms_per_step'], d['collection_only']/1e6, d['roofline']['kernel_ms']*1e3))"
done | tee $O/envs_sweep.txt
