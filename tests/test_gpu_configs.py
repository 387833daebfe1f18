"""BASELINE.json configs 4 and 5 at their full env counts, on ONE MI355X (config 4 = task go2, 32768 envs = 8 x 4096;
config 5 = task go2_moe_cts, 8192 envs = 8 x 1024).  What a single GPU can show of an 8-GPU configuration:
  * the product path runs at the full size (finite, counters advance, HIP-graph mode for >= 3 iterations) — and at the per-GPU shard size;
  * 8 shards stepped with env_offset r * n are, bit for bit, the rows [r n, (r+1) n) of the one full-size simulator for a whole rollout
    (24 steps) on the tasks' own trimesh terrain: what a rank computes does not depend on how the envs are partitioned.
(One lane mapping serves every size: the measured alternative for large batches — the same lane programs under a 256-register budget, two
waves per SIMD — is 1.5-2.4x SLOWER at every size, profiles/r3_kernel_scaling.txt.)
Run with -m gpu."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from helpers import DeviceSim, heightfield_overrides, load_hip  # noqa: E402


@pytest.fixture(scope="module")
def hip():
    lib = load_hip()
    assert lib.go2sim_is_device_library() == 1
    return lib


@pytest.mark.parametrize("task,N", [("go2", 32768), ("go2", 4096), ("go2_cts", 4096), ("go2_moe_cts", 8192), ("go2_moe_cts", 1024)])
def test_baseline_config_runs_at_full_size_in_graph_mode(hip, task, N):
    """task_registry -> LeggedRobot (HIP library) -> OnPolicyRunner / OnPolicyRunnerCTS at the configuration's env count: 6 iterations, the
    last >= 3 of them replayed from HIP graphs (rollout + every mini-batch step)."""
    import torch
    from go2_rl_gym_amd.envs import task_registry  # noqa: F401
    from go2_rl_gym_amd.utils import get_args
    args = get_args(["--task", task, "--num_envs", str(N), "--headless", "--seed", "1"])
    env, env_cfg = task_registry.make_env(task, args)
    assert env.num_envs == N and env_cfg.terrain.mesh_type == "trimesh" and env.custom_origins
    if task == "go2_moe_cts":
        _, tc = task_registry.get_cfgs(task)
        assert tc.runner.algorithm_class_name == "MoECTS" and (N != 8192 or task_registry.get_cfgs(task)[0].env.num_envs == 8192)      # go2_config.py:33
    torch.manual_seed(1)
    runner, _ = task_registry.make_alg_runner(env, task, args, log_root=None)
    env.common_step_counter = 0
    runner.learn(3, init_at_random_ep_len=True)
    for _ in range(3):
        runner.learn(1)
        torch.cuda.synchronize()
    g = runner.graphs_captured()
    assert g["rollout"] and g["update"], g
    assert env.common_step_counter == 6 * 24
    model = runner.alg.actor_critic if hasattr(runner.alg, "actor_critic") else runner.alg.model
    params = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    assert torch.isfinite(params).all() and torch.isfinite(env.obs_buf).all() and torch.isfinite(env.privileged_obs_buf).all() and torch.isfinite(env.rew_buf).all()
    assert torch.isfinite(env.root_states).all() and float(env.root_states[:, 2].median()) > 0.1
    assert env.measured_heights.abs().max() > 0.02 and runner.last_fps > 0
    # the terrain curriculum round robin covered every level row / type column the task starts on, at this size
    assert int(env.terrain_types.max()) == env_cfg.terrain.num_cols - 1 and int(env.terrain_levels.min()) == 0
    print("%s N=%d: %.2f M env-steps/s (collection %.1f ms, learn %.1f ms)" % (task, N, runner.last_fps / 1e6, 1e3 * runner.last_collection_time, 1e3 * runner.last_learn_time))
    env.close()


KEYS = ("root_states", "dof_state", "obs_buf", "privileged_obs_buf", "rew_buf", "reset_buf", "time_out_buf", "commands", "env_origins", "friction_coeffs", "link_mass_ratio",
        "added_base_mass", "motor_strengths", "episode_length_buf", "terrain_levels", "terrain_types", "contact_forces", "measured_heights", "episode_sums", "torques")


@pytest.mark.parametrize("Ng,n", [(32768, 4096), (8192, 1024)])
def test_eight_shards_are_slices_of_the_full_size_simulator(hip, Ng, n):
    """configs 4 / 5 as the 8 ranks hold them (env_offset = rank * n, num_envs_global = Ng) against ONE simulator of Ng envs: after create,
    reset and each of 24 steps (one rollout) every compared tensor of shard r equals rows [r n, (r+1) n) of the whole, bit for bit."""
    import torch
    ov = heightfield_overrides(Ng, mesh_type="trimesh")[1]
    whole = DeviceSim(hip, num_envs=Ng, seed=5, **ov)
    parts = [DeviceSim(hip, num_envs=n, env_offset=r * n, num_envs_global=Ng, seed=5, **ov) for r in range(Ng // n)]
    assert len(parts) == 8

    def same(tag):
        for k in KEYS:
            w = whole.t[k]
            for r, prt in enumerate(parts):
                sl = w[:, r * n:(r + 1) * n] if k == "episode_sums" else w[r * n:(r + 1) * n]
                assert torch.equal(sl, prt.t[k]), "%s: %s differs between shard %d and the slice of the %d-env simulator" % (tag, k, r, Ng)
    same("create")
    for s_ in [whole] + parts:
        s_.reset_all()
    torch.cuda.synchronize()
    same("reset")
    g = torch.Generator(device="cuda:0"); g.manual_seed(0)
    nreset = 0
    for it in range(24):
        a = torch.randn(Ng, 12, device="cuda:0", generator=g)
        hip.go2sim_step(whole.h, C.c_void_p(a.data_ptr()), whole._st())
        for r, prt in enumerate(parts):
            ar = a[r * n:(r + 1) * n].contiguous()
            hip.go2sim_step(prt.h, C.c_void_p(ar.data_ptr()), prt._st())
        torch.cuda.synchronize()
        same("step %d" % it)
        nreset += int(whole.t["reset_buf"].sum())
    assert torch.isfinite(whole.t["privileged_obs_buf"]).all() and float(whole.t["contact_forces"].abs().max()) > 1.0
    assert nreset > 0 and float(whole.t["measured_heights"].abs().max()) > 0.02      # resets and rough ground were part of what was compared
    for s_ in [whole] + parts:
        s_.close()


@pytest.mark.timeout(240)
@pytest.mark.parametrize("task", ["go2_flat", "go2_cts"])
def test_trains_at_8192_envs_in_the_default_configuration_without_a_second_stream(hip, task, monkeypatch):
    """Round 3 left a hang above 4096 envs per GPU (two chains of vendor GEMMs on two HIP streams: the first iteration never finished).  Since round 5 nothing forks a
    stream any more — PPO's and CTS's mini-batches are grouped launches of own kernels without autograd, the rollouts one / two policy kernels — so the default
    configuration must simply train at 8192 envs: 3 eager + 3 replayed iterations inside the time limit, everything from HIP graphs, and the module-by-module
    formulation that used to fork (_RolloutHeads._pair) is never entered (checked by counting)."""
    import time
    import torch
    from go2_rl_gym_amd.envs import task_registry  # noqa: F401
    from go2_rl_gym_amd.utils import get_args
    made = []
    t0 = time.time()
    args = get_args(["--task", task, "--num_envs", "8192", "--headless", "--seed", "1"])
    env, _ = task_registry.make_env(task, args)
    torch.manual_seed(1)
    runner, _ = task_registry.make_alg_runner(env, task, args, log_root=None)
    alg = runner.alg
    pair = alg._pair

    def counted_pair(f, g, enabled=True):
        made.append(1)
        return pair(f, g, enabled)
    monkeypatch.setattr(alg, "_pair", counted_pair)
    runner.learn(3, init_at_random_ep_len=True)
    for _ in range(3):
        runner.learn(1)
        torch.cuda.synchronize()
    g = runner.graphs_captured()
    assert g["rollout"] and g["update"], g
    assert not made, "the default path went through _RolloutHeads._pair (the module-by-module formulation)"
    if task == "go2_cts":
        assert alg._own_plan() is not None and alg._own_student() and alg._policy_kernel() is not None
    model = alg.actor_critic
    params = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    assert torch.isfinite(params).all() and runner.last_fps > 0 and time.time() - t0 < 180, time.time() - t0
    env.close()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("task,ncoll", [("go2_flat", 21), ("go2_cts", 41)])
def test_multi_rank_code_path_at_4096_envs_replays_with_21_collectives(hip, monkeypatch, task, ncoll):
    """The N > 1 update at the headline size on one GPU: a 1-rank RCCL group with GO2_FORCE_COLLECTIVES=1 runs the shipped serial schedule — per mini-batch
    slot two captured halves with the gradient + KL bucket all-reduced between their replays, plus the 24-byte advantage-statistics all-reduce: 21 collectives
    per iteration — for >= 3 replayed iterations; CTS: the policy steps' and the student steps' buckets, 41 (the own no-autograd mini-batches between the
    halves).  What it cannot show is xGMI behaviour; DESIGN 7 states the predicted 8-rank cost."""
    import socket
    import torch
    import torch.distributed as dist
    from go2_rl_gym_amd.envs import task_registry  # noqa: F401
    from go2_rl_gym_amd.rsl_rl.algorithms._graph import ReducedStep
    from go2_rl_gym_amd.utils import get_args
    monkeypatch.setenv("GO2_FORCE_COLLECTIVES", "1")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda:0"))
    calls = {"n": 0}
    real = dist.all_reduce

    def counted(*a, **k):
        calls["n"] += 1
        return real(*a, **k)
    try:
        args = get_args(["--task", task, "--num_envs", "4096", "--headless", "--seed", "2"])
        env, _ = task_registry.make_env(task, args)
        torch.manual_seed(2)
        runner, _ = task_registry.make_alg_runner(env, task, args, log_root=None)
        runner.learn(4, init_at_random_ep_len=True)          # eager warm-ups + captures
        torch.cuda.synchronize()
        steps = runner.alg._graph if task == "go2_flat" else runner.alg._steps[0] + runner.alg._steps[1]
        if task == "go2_cts":
            assert runner.alg._own_plan() is not None and runner.alg._own_student()
        assert all(isinstance(g, ReducedStep) and g.front.graph is not None and g.back.graph is not None for g in steps), "both halves of every slot are graphs"
        monkeypatch.setattr(dist, "all_reduce", counted)
        for _ in range(3):
            runner.learn(1)
        torch.cuda.synchronize()
        assert calls["n"] == 3 * ncoll, calls
        g = runner.graphs_captured()
        assert g["rollout"] and g["update"], g
        params = torch.cat([p.detach().reshape(-1) for p in runner.alg.actor_critic.parameters()])
        assert torch.isfinite(params).all() and 1e-5 - 1e-12 <= runner.alg.learning_rate <= 1e-2 + 1e-12
        print("%s N=4096, 1-rank RCCL, serial schedule: %.2f M env-steps/s with %d all-reduces per iteration" % (task, runner.last_fps / 1e6, ncoll))
        env.close()
    finally:
        monkeypatch.setattr(dist, "all_reduce", real)
        dist.destroy_process_group()


SWITCH_SCRIPT = """
import sys, torch
sys.path.insert(0, %r)
from go2_rl_gym_amd.envs import task_registry
from go2_rl_gym_amd.utils import get_args
task = sys.argv[1]
args = get_args(["--task", task, "--num_envs", "512", "--headless", "--seed", "4"])
env, _ = task_registry.make_env(task, args)
torch.manual_seed(4)
runner, _ = task_registry.make_alg_runner(env, task, args, log_root=None)
runner.learn(6, init_at_random_ep_len=True)
torch.cuda.synchronize()
model = runner.alg.actor_critic
p = torch.cat([q.detach().reshape(-1) for q in model.parameters()])
g = runner.graphs_captured()
assert torch.isfinite(p).all() and torch.isfinite(env.obs_buf).all() and g["rollout"] and g["update"], g
print("OK %%s %%.3f" %% (task, float(env.rew_buf.mean())))
"""


@pytest.mark.timeout(400)
@pytest.mark.parametrize("switch", ["GO2_FUSE_STEP", "GO2_FUSED_ADAM", "GO2_FUSED_MLP", "GO2_GEMM_SPLIT", "GO2_FUSED_POLICY"])
def test_every_formulation_switch_still_trains(switch):
    """The five switches that select an older / reference formulation (README: =0 each; the other five are GO2_STRICT_GRAPHS, GO2_TUNE_GEMM, GO2_FORCE_COLLECTIVES,
    GO2_DIST_BACKEND — covered by bench.py and the multi-rank tests — and GO2_HIPCC_FLAGS, the build's): PPO and CTS train for 6 iterations at 512 envs in HIP-graph
    mode with the switch off, in a process of its own (the switches are read at import)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for task in ("go2_flat", "go2_flat_cts"):
        r = subprocess.run([sys.executable, "-c", SWITCH_SCRIPT % root, task], env=dict(os.environ, **{switch: "0"}), capture_output=True, text=True, timeout=180)
        assert r.returncode == 0 and ("OK " + task) in r.stdout, (switch, task, r.stdout[-500:], r.stderr[-1500:])
