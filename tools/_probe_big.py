"""Instrumented run of a task at a large env count (where does it spend its time / hang?): python tools/_probe_big.py task N"""
import faulthandler, os, sys, time
faulthandler.dump_traceback_later(90, repeat=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
task, N = sys.argv[1], int(sys.argv[2])
t0 = time.time()
def say(m):
    print("[%.1fs] %s" % (time.time() - t0, m), flush=True)
from go2_rl_gym_amd.envs import task_registry
from go2_rl_gym_amd.utils import get_args
args = get_args(["--task", task, "--num_envs", str(N), "--headless", "--seed", "1"])
env, env_cfg = task_registry.make_env(task, args); torch.cuda.synchronize(); say("make_env")
runner, _ = task_registry.make_alg_runner(env, task, args, log_root=None); torch.cuda.synchronize(); say("runner")
env.common_step_counter = 0
for i in range(6):
    runner.learn(1, init_at_random_ep_len=(i == 0)); torch.cuda.synchronize()
    say("iteration %d: %.2f M env-steps/s (collection %.1f ms, learn %.1f ms) graphs %s" % (i, runner.last_fps / 1e6, 1e3 * runner.last_collection_time, 1e3 * runner.last_learn_time, runner.graphs_captured()))
env.close(); say("done")
