"""bench.py's command line on a box without a GPU: `--gpus N > 1` started plainly becomes its own launcher (one rank per GPU under torch.distributed.run) instead of
exiting on argument checking; the ranks then refuse to run without an MI355X — the product has no CPU path (the GPU twin: tests/test_gpu_parity.py::
test_two_rank_bench_rehearsal_on_one_gpu[plain])."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.is_available(), reason="the GPU rehearsal covers the launch on a GPU box")
def test_plain_multi_gpu_start_launches_its_own_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["GO2_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--num-envs", "64"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0                                                       # no GPU here: every rank refuses
    assert "needs torch.distributed.run" not in r.stderr                           # (round 5's exit on argument checking)
    assert r.stderr.count("bench.py needs an MI355X") >= 2, r.stderr[-3000:]       # two ranks were started and each one said so
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]          # and no number came out of a box without a GPU
