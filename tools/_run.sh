export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
DBG_DUMP=/tmp/mix python -m pytest tests/test_gpu_parity.py -q -k "cts_training_graph_vs_eager and moe" > /dev/null 2>&1
DBG_MIX_TORCH=1 DBG_DUMP=/tmp/tor python -m pytest tests/test_gpu_parity.py -q -k "cts_training_graph_vs_eager and moe" > /dev/null 2>&1
python - <<'PY'
import numpy as np
for mode in (0, 1):
    a, b = np.load("/tmp/mix_%d.npz" % mode), np.load("/tmp/tor_%d.npz" % mode)
    d = {k: float(np.median(np.abs(a[k] - b[k]))) for k in a.files}
    k = max(d, key=d.get)
    print("arm %s: kernel-mix rollout vs torch-mix rollout, weights after iteration 1: largest per-tensor median gap %.2e (%s); actor.network.0.weight %.2e" % ("graph" if mode else "eager", d[k], k, d.get("actor.network.0.weight", -1)))
PY
