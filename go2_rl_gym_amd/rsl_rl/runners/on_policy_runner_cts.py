"""OnPolicyRunnerCTS (rsl_rl/rsl_rl/runners/on_policy_runner_cts.py:63-360): the PPO runner plus the 5-frame observation
history the student encoder consumes, teacher/student episode statistics and the two-optimizer checkpoint.

The history ring [N, 5, 45] lives in one persistent device buffer updated IN PLACE by the library kernel
go2sim_history_push (zero finished envs, drop the oldest frame, append the new observation — :155-156) instead of a boolean
index write + torch.cat allocation per step; that also makes the whole 24-step rollout capturable in one HIP graph, exactly
like the PPO runner's.  RoboGauge (:103-111,271-330, an external HTTP evaluation service) is out of scope."""
import ctypes as C
from collections import deque

import torch

from ..algorithms import CTS, ACMoECTS, DualMoECTS, MCPCTS, MoECTS, MoENGCTS
from ..modules import ActorCriticACMoECTS, ActorCriticCTS, ActorCriticDualMoECTS, ActorCriticMCPCTS, ActorCriticMoECTS, ActorCriticMoENGCTS
from ..algorithms._graph import load_optimizer_state
from .on_policy_runner import OnPolicyRunner

_POLICIES = {"ActorCriticCTS": ActorCriticCTS, "ActorCriticMoECTS": ActorCriticMoECTS, "ActorCriticMoENGCTS": ActorCriticMoENGCTS,
             "ActorCriticACMoECTS": ActorCriticACMoECTS, "ActorCriticDualMoECTS": ActorCriticDualMoECTS, "ActorCriticMCPCTS": ActorCriticMCPCTS}
_ALGS = {"CTS": CTS, "MoECTS": MoECTS, "MoENGCTS": MoENGCTS, "ACMoECTS": ACMoECTS, "DualMoECTS": DualMoECTS, "MCPCTS": MCPCTS}


class OnPolicyRunnerCTS(OnPolicyRunner):
    _TERRAIN_TAGS = True
    _ITER_IN_CHECKPOINTS = True       # on_policy_runner_cts.py:195: incremented before the save
    _LOSS_NAMES = OnPolicyRunner._LOSS_NAMES + (("mean_entropy_loss", "Loss/entropy", "Entropy loss:"), ("mean_latent_loss", "Loss/latent", "Latent loss:"),
                                                ("mean_load_balance_loss", "Loss/load_balance", "Load balance loss:"),
                                                ("mean_actor_load_balance_loss", "Loss/actor_load_balance", "Actor load balance loss:"))

    def _build_algorithm(self, train_cfg, use_graphs):
        env = self.env
        if env.num_privileged_obs is None:
            raise ValueError("CTS needs privileged observations (on_policy_runner_cts.py:127)")
        if self.lib is None:
            raise RuntimeError("OnPolicyRunnerCTS needs the go2sim library (history ring / GAE / loss kernels); the env did not provide one")
        H = self.history_length = train_cfg["history_length"]
        name = self.cfg["policy_class_name"]
        if name not in _POLICIES or self.cfg["algorithm_class_name"] not in _ALGS:
            raise NotImplementedError("policy %r / algorithm %r: the reference's six CTS-family algorithms are built" % (name, self.cfg["algorithm_class_name"]))
        model = _POLICIES[name](env.num_obs, env.num_privileged_obs, env.num_actions, env.num_envs, H, **self.policy_cfg).to(self.device)
        self.alg = _ALGS[self.cfg["algorithm_class_name"]](model, env.num_envs, H, device=self.device, lib=self.lib, use_graphs=use_graphs, **self.alg_cfg)
        self.history = torch.zeros(env.num_envs, H, env.num_obs, device=self.device)
        self._teacher_mask = torch.zeros(env.num_envs, dtype=torch.bool, device=self.device)
        self._teacher_mask[self.alg.teacher_env_idxs] = True
        self._t_rew, self._t_len, self._s_rew, self._s_len = (deque(maxlen=100) for _ in range(4))

    # ---- history ring ----
    def _push_history(self, obs, dones):
        N, H, D = self.history.shape
        obs = obs.contiguous()
        on_dev = self.lib.go2sim_is_device_library() == 1
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream) if on_dev else None
        d = None if dones is None else C.c_void_p((dones if dones.dtype == torch.uint8 else dones.view(torch.uint8)).data_ptr())
        rc = self.lib.go2sim_history_push(C.c_void_p(self.history.data_ptr()), C.c_void_p(obs.data_ptr()), d, N, H, D, stream)
        if rc != 0:
            raise RuntimeError("go2sim_history_push failed: %s" % self.lib.go2sim_last_error().decode())

    def _begin_learn(self):
        self._push_history(self.env.get_observations().to(self.device), None)       # :129

    def _rollout(self, bk):
        env, alg, T = self.env, self.alg, self.num_steps_per_env
        obs, privileged_obs = env.get_observations().to(self.device), env.get_privileged_observations().to(self.device)
        ep_infos = []
        if hasattr(env, "_info_slot"):
            env._info_slot = 0
        for i in range(T):
            actions = alg.act(obs, privileged_obs, self.history.flatten(1))
            tg = alg.rollout_targets() if self._fuse_step else None
            obs, privileged_obs, rewards, dones, infos = env.step(actions, rollout=tg) if tg is not None else env.step(actions)
            obs, privileged_obs, rewards, dones = obs.to(self.device), privileged_obs.to(self.device), rewards.to(self.device), dones.to(self.device)
            self._push_history(obs, dones)                                         # :155-156
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

    def _compute_returns(self):
        obs, priv = self.env.get_observations().to(self.device), self.env.get_privileged_observations().to(self.device)
        if isinstance(self.alg, ACMoECTS):        # (on_policy_runner_cts.py:174-177)
            self.alg.compute_returns(obs, priv, self.history.flatten(1))
        else:
            self.alg.compute_returns(priv, self.history.flatten(1))

    def _collect_episode_stats(self, bk):
        m = bk["fin_mask"].cpu().numpy()
        rew, ln = bk["fin_rew"].cpu().numpy(), bk["fin_len"].cpu().numpy()
        t = self._teacher_mask.cpu().numpy()[None, :] & m
        s = (~self._teacher_mask.cpu().numpy())[None, :] & m
        self._t_rew.extend(rew[t].tolist()); self._t_len.extend(ln[t].tolist())
        self._s_rew.extend(rew[s].tolist()); self._s_len.extend(ln[s].tolist())

    def _reward_stats(self):
        return [("teacher_", "teacher ", self._t_rew, self._t_len), ("student_", "student ", self._s_rew, self._s_len)]

    # ---- checkpoints: same keys as the reference (:250-258,332-339) ----
    def save(self, path, it=None, last_model=False, infos=None):
        torch.save({"model_state_dict": self.alg.model.state_dict(), "optimizer1_state_dict": self.alg.optimizer1.state_dict(),
                    "optimizer2_state_dict": self.alg.optimizer2.state_dict(), "iter": self.current_learning_iteration, "infos": infos}, path)

    def load(self, path, load_optimizer=True):
        d = torch.load(path, map_location=self.device)
        self.alg.model.load_state_dict(d["model_state_dict"])
        if load_optimizer:
            load_optimizer_state(self.alg.optimizer1, d["optimizer1_state_dict"])
            load_optimizer_state(self.alg.optimizer2, d["optimizer2_state_dict"])
            self.alg.rebind_lr()
        self.current_learning_iteration = d["iter"]
        self.alg.set_shuffle_counter(d["iter"])          # graph mode's keyed permutations continue at the checkpoint's iteration
        return d["infos"]

    def get_inference_policy(self, device=None):
        self.alg.model.eval()
        if device is not None:
            self.alg.model.to(device)
        return self.alg.model.act_inference
