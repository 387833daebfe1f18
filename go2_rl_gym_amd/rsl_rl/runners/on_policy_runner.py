"""OnPolicyRunner with the reference's interface and semantics (rsl_rl/rsl_rl/runners/on_policy_runner.py:60-309):
24-step rollout -> GAE + advantage normalisation -> 5 epochs x 4 mini-batches of PPO; same checkpoint keys, same
console / scalar names, fps = num_steps_per_env * num_envs / (collection_time + learn_time) (:194).

Differences that do not change results: the per-step `.cpu()` bookkeeping of the reference (:143-153, a host sync every
env step) is replaced by device-side buffers read once per iteration; RoboGauge (:103-111,252-295, an external HTTP
service) is out of scope; with torch.distributed initialised only rank 0 logs and saves.
"""
import os
import statistics
import tempfile
import time
from collections import deque
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist
import yaml

from ...utils.helpers import class_to_dict
from ..algorithms import PPO
from ..algorithms._graph import load_optimizer_state, no_gc, strict_graphs
from ..env import missing_members
from ..modules import ActorCritic

_POLICIES = {"ActorCritic": ActorCritic}
_ALGS = {"PPO": PPO}

for _t, _f in ((np.float32, float), (np.float64, float), (np.int32, int), (np.int64, int)):
    yaml.add_representer(_t, (lambda f: (lambda dumper, data: dumper.represent_float(float(data)) if f is float else dumper.represent_int(int(data))))(_f), Dumper=yaml.SafeDumper)


def _rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def _world():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def _enable_tuned_gemms():
    """Use PyTorch TunableOp with the shipped selection table (go2_rl_gym_amd/tuning/tunableop_gfx950.csv): for the fp32 GEMM
    shapes of this workload (4096-row rollout, 24576-row mini-batches) it picks the fastest hipBLASLt / rocBLAS solution
    (measured 1.6 -> 2.2 M env-steps/s).  Same fp32 arithmetic.  Set PYTORCH_TUNABLEOP_ENABLED yourself to override; set
    GO2_TUNE_GEMM=1 to (re)tune online for other sizes."""
    if os.environ.get("PYTORCH_TUNABLEOP_ENABLED") is not None:
        return
    try:
        import torch.cuda.tunable as tn
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tuning", "tunableop_gfx950.csv")
        tune = os.environ.get("GO2_TUNE_GEMM", "0") == "1"
        if not tune and not os.path.exists(path):
            return
        tn.enable(True)
        tn.tuning_enable(tune)
        tn.record_untuned_enable(False)
        if tune:                       # keeps the shipped selections, tunes the shapes that are not in the table and writes those to the cwd
                                       # (append them to the shipped table)
            if os.path.exists(path):
                tn.read_file(path)
            tn.set_max_tuning_duration(50)
            tn.set_filename(os.path.join(os.getcwd(), "tunableop_results.csv"))
        else:
            ok = tn.read_file(path)
            tn.set_filename(os.path.join(tempfile.gettempdir(), "go2_tunableop_unused.csv"))   # never write over the shipped table
            if not ok:
                tn.enable(False)       # validators (torch / hipBLASLt / arch) do not match this table
    except Exception as e:             # never fail the run over an optional speed-up
        print("[go2_rl_gym_amd] TunableOp not enabled:", e)


class _NullWriter:
    def add_scalar(self, *a, **k):
        pass


def _make_writer(log_dir):
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(log_dir=log_dir, flush_secs=10)
    except Exception:
        return _NullWriter()


class OnPolicyRunner:
    # The reference's PPO runner advances current_learning_iteration only after the loop (on_policy_runner.py:168-171: every intermediate
    # checkpoint carries the START iteration); its CTS runner advances it per iteration before saving (on_policy_runner_cts.py:195), so
    # a mid-run CTS checkpoint resumes the curricula where they were (train.py restores common_step_counter = iter * 24).
    _ITER_IN_CHECKPOINTS = False
    _TERRAIN_TAGS = False       # the PPO runner logs every episode key under 'Episode/' (on_policy_runner.py:191); the CTS runner splits off 'Terrain/' (:221-224)
    def __init__(self, env, train_cfg, log_dir=None, device="cpu", use_graphs=None):
        self.cfg = train_cfg["runner"]
        self.alg_cfg = train_cfg["algorithm"]
        self.policy_cfg = train_cfg["policy"]
        self.device = device
        lacking = missing_members(env)
        if lacking:
            raise TypeError("env does not satisfy the VecEnv contract (rsl_rl/env/vec_env.py); missing: " + ", ".join(lacking))
        self.env = env
        self.lib = getattr(env, "lib", None)
        self._build_algorithm(train_cfg, use_graphs)
        self.num_steps_per_env = self.cfg["num_steps_per_env"]
        self.save_interval = self.cfg["save_interval"]
        self.alg.init_storage(self.env.num_envs, self.num_steps_per_env, [self.env.num_obs], [self.env.num_privileged_obs], [self.env.num_actions])
        self.log_dir = log_dir if _rank() == 0 else None
        self.writer = None
        self.tot_timesteps = 0
        self.tot_time = 0
        self.current_learning_iteration = 0
        self.last_fps = None
        self.last_collection_time = self.last_learn_time = None
        on_gpu = str(device).startswith("cuda") and getattr(getattr(env, "lib", None), "go2sim_is_device_library", lambda: 0)() == 1
        if on_gpu:
            _enable_tuned_gemms()
            # ROCm trap (round 4): since plain PPO no longer issues a single vendor GEMM, nothing initialises hipBLASLt before the first HIP-graph capture, and
            # this PyTorch build then dies inside the capture with "operation not permitted when stream is capturing" (hipblaslt.cpp:171: the library's lazy
            # handle creation).  One 8 x 8 product, eagerly, before anything is captured, restores the order of events every earlier round had.
            _w = torch.zeros(8, 8, device=self.device)
            torch.addmm(_w[0], _w, _w)
            torch.cuda.synchronize(self.device)
        self.use_graphs = bool(on_gpu and self.alg.use_graphs) if use_graphs is None else bool(use_graphs and on_gpu)
        self._rollout_graph, self._graph_ep_infos, self._eager_rollouts = None, None, 0
        self._returns_graph = None
        # one policy step = { policy kernels, ONE env library call }: observations land in the next storage rows, the transition store rides in
        # the step kernel (go2sim_step_rollout); GO2_FUSE_STEP=0 restores copy + store launches
        # (only when the storage rows are tensors the env kernel can write: same device as the env; LeggedRobot.step falls back to the plain
        # step for any destination it cannot write in place)
        def _norm(d):          # 'cuda' and 'cuda:0' name the same device
            d = torch.device(d)
            return torch.device(d.type, torch.cuda.current_device()) if d.type == "cuda" and d.index is None else d
        same_dev = _norm(device) == _norm(getattr(env, "device", device))
        if on_gpu and not same_dev:
            print("[OnPolicyRunner] rl_device %s differs from the env's device %s: the fused env step (observations written into the rollout storage in place) is off" % (device, getattr(env, "device", None)))
        self._fuse_step = (bool(on_gpu or os.environ.get("GO2_FUSE_STEP") == "1") and os.environ.get("GO2_FUSE_STEP", "1") != "0" and hasattr(self.env, "_info_ring")
                           and same_dev)
        N, T = self.env.num_envs, self.num_steps_per_env
        self._rewbuffer, self._lenbuffer = deque(maxlen=100), deque(maxlen=100)
        z = lambda *s_, **k: torch.zeros(*s_, device=self.device, **k)
        self._bk = {"cur_rew": z(N), "cur_len": z(N), "fin_rew": z(T, N), "fin_len": z(T, N), "fin_mask": z(T, N, dtype=torch.bool)}
        _, _ = self.env.reset()
        if self.log_dir is not None and self.env.cfg.env.test is False:
            Path(self.log_dir).mkdir(parents=True, exist_ok=True)
            all_cfg = {"train_cfg": train_cfg, "env_cfg": class_to_dict(self.env.cfg)}
            yaml.safe_dump(all_cfg, open(os.path.join(self.log_dir, "config.yaml"), "w"))

    # ---- hooks the CTS runner overrides ----
    _LOSS_NAMES = (("mean_value_loss", "Loss/value_function", "Value function loss:"), ("mean_surrogate_loss", "Loss/surrogate", "Surrogate loss:"))

    def _build_algorithm(self, train_cfg, use_graphs):
        num_critic_obs = self.env.num_privileged_obs if self.env.num_privileged_obs is not None else self.env.num_obs
        actor_critic = _POLICIES[self.cfg["policy_class_name"]](self.env.num_obs, num_critic_obs, self.env.num_actions, **self.policy_cfg).to(self.device)
        self.alg = _ALGS[self.cfg["algorithm_class_name"]](actor_critic, device=self.device, lib=self.lib, use_graphs=use_graphs, **self.alg_cfg)

    def _begin_learn(self):
        pass

    def _compute_returns(self):
        obs = self.env.get_observations()
        privileged_obs = self.env.get_privileged_observations()
        critic_obs = (privileged_obs if privileged_obs is not None else obs).to(self.device)
        self.alg.compute_returns(critic_obs)

    def _returns_step(self):
        """compute_returns (bootstrap value of the last observations, GAE, advantage normalisation: on_policy_runner.py:158, rollout_storage.py:123-137)
        — replayed from its own small HIP graph when the rollout is (eager it is ~10 launches with host gaps between them); eager whenever a
        collective sits inside it (more than one rank: the advantage-statistics all-reduce)."""
        from ..algorithms.ppo import _collectives_on
        if self._rollout_graph is None or _collectives_on():
            return self._compute_returns()
        if self._returns_graph is None:
            from ..algorithms._graph import CapturedStep
            self._returns_graph = CapturedStep(self._compute_returns, enabled=True, warmup=1, name="compute_returns")
        self._returns_graph()

    def _sync(self):
        if str(self.device).startswith("cuda"):
            torch.cuda.synchronize(self.device)

    # ---- the 24-step rollout body (on_policy_runner.py:135-153); pure enqueue, no host sync ----
    def _rollout(self, bk):
        env, alg, T = self.env, self.alg, self.num_steps_per_env
        obs = env.get_observations()
        privileged_obs = env.get_privileged_observations()
        critic_obs = privileged_obs if privileged_obs is not None else obs
        ep_infos = []
        if hasattr(env, "_info_slot"):
            env._info_slot = 0          # same extras ring slots every iteration (needed for graph replay, harmless otherwise)
        for i in range(T):
            actions = alg.act(obs.to(self.device), critic_obs.to(self.device))
            tg = alg.rollout_targets() if (self._fuse_step and hasattr(alg, "rollout_targets")) else None
            obs, privileged_obs, rewards, dones, infos = env.step(actions, rollout=tg) if tg is not None else env.step(actions)
            critic_obs = privileged_obs if privileged_obs is not None else obs
            rewards, dones = rewards.to(self.device), dones.to(self.device)
            alg.process_env_step(rewards, dones, infos)
            if bk is not None:
                if "episode" in infos:
                    ep_infos.append(infos["episode"])
                bk["cur_rew"] += rewards
                bk["cur_len"] += 1
                bk["fin_mask"][i] = dones
                bk["fin_rew"][i] = bk["cur_rew"]
                bk["fin_len"][i] = bk["cur_len"]
                keep = (~dones).float()
                bk["cur_rew"] *= keep
                bk["cur_len"] *= keep
        return ep_infos

    def learn(self, num_learning_iterations, init_at_random_ep_len=False):
        if self.log_dir is not None and self.writer is None:
            Path(self.log_dir).mkdir(parents=True, exist_ok=True)     # (the reference relies on SummaryWriter to create it)
            self.writer = _make_writer(self.log_dir)
        if init_at_random_ep_len:
            self.env.episode_length_buf = torch.randint_like(self.env.episode_length_buf, high=int(self.env.max_episode_length))
        self.alg.actor_critic.train()
        self._begin_learn()
        rewbuffer, lenbuffer = self._rewbuffer, self._lenbuffer
        N, T = self.env.num_envs, self.num_steps_per_env
        bk = self._bk if self.log_dir is not None else None
        start_iter = self.current_learning_iteration
        tot_iter = start_iter + num_learning_iterations
        it = start_iter
        gpu_clock = str(self.device).startswith("cuda")
        self._tot_time_at_start = self.tot_time         # tot_time accumulates over learn() calls; the ETA is about THIS call
        self._sync()
        for it in range(start_iter, tot_iter):
            # collection / learning time as the reference logs them (on_policy_runner.py:133,154-155,162-163).  On the GPU the split point is a
            # stream EVENT, read after the iteration's one host sync: a device synchronisation between rollout and update would idle the GPU
            # for ~0.1 ms per iteration
            start = time.time()
            if gpu_clock:
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                ev[0].record()
            with torch.inference_mode():
                #  what the captured rollout contains was decided by host-side state at capture time: the CTS module's deployment-side history being
                # all zeros lets process_env_step skip its reset.  If that flag has changed since (act_inference on the live runner, a checkpoint loaded), the graph
                # is dropped and captured again from the next eager rollouts
                sig = bool(getattr(getattr(self.alg, "model", None), "_history_dirty", False))
                if self._rollout_graph is not None and sig != getattr(self, "_rollout_graph_sig", sig):
                    self._rollout_graph, self._returns_graph, self._eager_rollouts = None, None, 0
                self._rollout_graph_sig = sig
                if self._rollout_graph is not None:
                    self._rollout_graph.replay()                       # 24 x (policy, env step kernel, storage) in ONE launch
                    self.env.lib.go2sim_notify_replayed(self.env.handle, T)
                    self.alg.storage.step = T
                    getattr(self.alg, "rollout_replayed", lambda: None)()
                    ep_infos = self._graph_ep_infos
                elif self.use_graphs and self._eager_rollouts >= 2:
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    try:
                        with no_gc(), torch.cuda.graph(g):
                            self._graph_ep_infos = self._rollout(bk)
                    except Exception as e:     # noqa: BLE001 — a capture problem must never stop training: fall back to the eager rollout
                        if strict_graphs():    # ... unless the caller asked for it to (bench.py: no number from a silently degraded run)
                            raise RuntimeError("HIP-graph capture of the rollout failed (%s: %s)" % (type(e).__name__, e)) from e
                        print("[go2_rl_gym_amd] HIP-graph capture of the rollout failed (%s: %s); continuing eagerly" % (type(e).__name__, e))
                        self.use_graphs = False
                        torch.cuda.synchronize()
                        self.alg.storage.step = 0
                        ep_infos = self._rollout(bk)
                    else:
                        self._rollout_graph = g                        # capture executes nothing ...
                        self.alg.storage.step = 0
                        g.replay()                                     # ... so run this iteration's rollout from the graph
                        self.env.lib.go2sim_notify_replayed(self.env.handle, T)
                        self.alg.storage.step = T
                        getattr(self.alg, "rollout_replayed", lambda: None)()
                        ep_infos = self._graph_ep_infos
                else:
                    ep_infos = self._rollout(bk)
                    self._eager_rollouts += 1
                if gpu_clock:
                    ev[1].record()
                else:
                    stop = time.time()
                    collection_time = stop - start
                    start = stop
                self._returns_step()
            losses = self.alg.update()
            mean_value_loss, mean_surrogate_loss = losses[0], losses[1]
            self._sync()
            stop = time.time()
            if gpu_clock:
                collection_time = ev[0].elapsed_time(ev[1]) * 1e-3
                learn_time = max(stop - start - collection_time, 0.0)
            else:
                learn_time = stop - start
            if self.log_dir is not None:
                self._collect_episode_stats(bk)
                self.log(locals())
            self.last_collection_time, self.last_learn_time = collection_time, learn_time
            self.last_fps = T * N * _world() / (collection_time + learn_time)
            if self._ITER_IN_CHECKPOINTS:
                self.current_learning_iteration = it + 1
            if self.log_dir is not None and it % self.save_interval == 0:
                self.save(os.path.join(self.log_dir, "model_{}.pt".format(it)), it, False)
        self.current_learning_iteration = tot_iter
        if self.log_dir is not None:
            self.save(os.path.join(self.log_dir, "model_{}.pt".format(self.current_learning_iteration)), it, True)

    def graphs_captured(self):
        """{"rollout": bool, "update": bool}: which halves of an iteration are being replayed from HIP graphs right now."""
        return {"rollout": self._rollout_graph is not None, "update": bool(getattr(self.alg, "graphs_captured", lambda: False)())}

    def _collect_episode_stats(self, bk):
        m = bk["fin_mask"].cpu().numpy()   # one device->host read per iteration, same deque order as the reference (step-major)
        self._rewbuffer.extend(bk["fin_rew"].cpu().numpy()[m].tolist())
        self._lenbuffer.extend(bk["fin_len"].cpu().numpy()[m].tolist())

    def _reward_stats(self):
        """-> [(tensorboard suffix, console label, rewards deque, lengths deque)]"""
        return [("", "", self._rewbuffer, self._lenbuffer)]

    def log(self, locs, width=80, pad=35):
        self.tot_timesteps += self.num_steps_per_env * self.env.num_envs
        self.tot_time += locs["collection_time"] + locs["learn_time"]
        iteration_time = locs["collection_time"] + locs["learn_time"]
        ep_string = ""
        if locs["ep_infos"]:
            for key in locs["ep_infos"][0]:
                vals = []
                for ep_info in locs["ep_infos"]:
                    v = ep_info[key]
                    v = torch.as_tensor(v, dtype=torch.float, device=self.device).reshape(-1)
                    vals.append(v)
                value = torch.mean(torch.cat(vals))
                self.writer.add_scalar(("Terrain/" if self._TERRAIN_TAGS and "terrain" in key else "Episode/") + key, value, locs["it"])
                ep_string += f"""{f'Mean episode {key}:':>{pad}} {value:.4f}\n"""
        ac = self.alg.actor_critic
        mean_std = ac.std.mean() if hasattr(ac, "std") else ac.action_std.mean()       # MCP-CTS has no std parameter (on_policy_runner_cts.py:218-219)
        fps = int(self.num_steps_per_env * self.env.num_envs / (locs["collection_time"] + locs["learn_time"]))
        w = self.writer
        for (name, tag, _), v in zip(self._LOSS_NAMES, locs["losses"]):
            w.add_scalar(tag, v, locs["it"])
        w.add_scalar("Loss/learning_rate", self.alg.learning_rate, locs["it"])
        if hasattr(ac, "std"):         # not for the MCP actor, whose std is an output of the network (on_policy_runner_cts.py:239-240)
            w.add_scalar("Policy/mean_noise_std", mean_std.item(), locs["it"])
        w.add_scalar("Perf/total_fps", fps, locs["it"])
        w.add_scalar("Perf/collection time", locs["collection_time"], locs["it"])
        w.add_scalar("Perf/learning_time", locs["learn_time"], locs["it"])
        groups = [g for g in self._reward_stats() if len(g[2]) > 0]
        for sfx, _, rb, lb in groups:
            w.add_scalar("Train/mean_%sreward" % sfx, statistics.mean(rb), locs["it"])
            w.add_scalar("Train/mean_%sepisode_length" % sfx, statistics.mean(lb), locs["it"])
            w.add_scalar("Train/mean_%sreward/time" % sfx, statistics.mean(rb), self.tot_time)
            w.add_scalar("Train/mean_%sepisode_length/time" % sfx, statistics.mean(lb), self.tot_time)
        head = f" \033[1m Learning iteration {locs['it']}/{locs['tot_iter']} \033[0m "
        s = (f"""{'#' * width}\n{head.center(width, ' ')}\n\n"""
             f"""{'Computation:':>{pad}} {fps:.0f} steps/s (collection: {locs['collection_time']:.3f}s, learning {locs['learn_time']:.3f}s)\n"""
             )
        for (_, _, label), v in zip(self._LOSS_NAMES, locs["losses"]):
            s += f"""{label:>{pad}} {v:.4f}\n"""
        s += f"""{'Mean action noise std:':>{pad}} {mean_std.item():.2f}\n"""
        for _, label, rb, lb in groups:
            s += (f"""{'Mean %sreward:' % label:>{pad}} {statistics.mean(rb):.2f}\n"""
                  f"""{'Mean %sepisode length:' % label:>{pad}} {statistics.mean(lb):.2f}\n""")
        s += ep_string
        s += (f"""{'-' * width}\n{'Total timesteps:':>{pad}} {self.tot_timesteps}\n{'Iteration time:':>{pad}} {iteration_time:.2f}s\n"""
              f"""{'Total time:':>{pad}} {self.tot_time:.2f}s\n"""
              f"""{'ETA:':>{pad}} {(self.tot_time - self._tot_time_at_start) / (locs['it'] - locs['start_iter'] + 1) * (locs['tot_iter'] - locs['it'] - 1):.1f}s\n""")
        print(s)

    def save(self, path, it=None, last_model=False, infos=None):
        torch.save({"model_state_dict": self.alg.actor_critic.state_dict(), "optimizer_state_dict": self.alg.optimizer.state_dict(),
                    "iter": self.current_learning_iteration, "infos": infos}, path)

    def load(self, path, load_optimizer=True):
        d = torch.load(path, map_location=self.device)
        self.alg.actor_critic.load_state_dict(d["model_state_dict"])
        if load_optimizer:
            load_optimizer_state(self.alg.optimizer, d["optimizer_state_dict"])
            self.alg.rebind_lr()
        self.current_learning_iteration = d["iter"]
        getattr(self.alg, "set_shuffle_counter", lambda it: None)(d["iter"])          # graph mode's keyed permutation continues at the checkpoint's iteration
        return d["infos"]

    def get_inference_policy(self, device=None):
        self.alg.actor_critic.eval()
        if device is not None:
            self.alg.actor_critic.to(device)
        return self.alg.actor_critic.act_inference
