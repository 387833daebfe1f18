"""Rollout storage of the Concurrent Teacher-Student algorithms (rsl_rl/rsl_rl/storage/rollout_storage_cts.py:36-216).

The reference re-orders every transition to [teacher envs | student envs] with ~25 gather/cat launches per env step
(algorithms/cts.py:112-149) and materialises all 20 mini-batches as a list.  Here the rollout stays in ENV order — the
same [T, N, .] tensors as PPO's storage plus the stacked history — and the teacher/student layout exists only as an index
map applied when a mini-batch is gathered: `ref2mine[k]` is the position, in this storage, of sample k of the reference's
flattened (env-major, teacher-first) layout.  A mini-batch is therefore the same SET of samples in the same order as the
reference's for the same permutation (tests/test_cts_golden.py)."""
import torch

from .rollout_storage import RolloutStorage


class RolloutStorageCTS(RolloutStorage):
    class Transition(RolloutStorage.Transition):
        def __init__(self):
            super().__init__()
            self.history = None

    def __init__(self, num_envs, teacher_env_idxs, student_env_idxs, history_length, num_transitions_per_env, obs_shape, privileged_obs_shape,
                 actions_shape, device="cpu", lib=None):
        super().__init__(num_envs, num_transitions_per_env, obs_shape, privileged_obs_shape, actions_shape, device, lib=lib)
        T, N = num_transitions_per_env, num_envs
        self.history_length = history_length
        self.teacher_num_envs, self.student_num_envs = len(teacher_env_idxs), len(student_env_idxs)
        self.history = torch.zeros(T, N, history_length * obs_shape[0], device=device)
        order = torch.cat([teacher_env_idxs, student_env_idxs]).to(device)               # reference row e' -> env
        k = torch.arange(N * T, device=device)
        self.ref2mine = (k % T) * N + order[k // T]                                     # reference flat (e'*T + t) -> this storage's (t*N + env)

    def add_transitions(self, transition):
        s = self.step
        super().add_transitions(transition)
        self.history[s].copy_(transition.history)

    def mini_batch_indices(self, num_mini_batches):
        """One permutation per update, teacher and student samples shuffled separately (rollout_storage_cts.py:152-160);
        -> list of index tensors into the flattened [T*N] storage, each laid out [teacher rows | student rows]."""
        T = self.num_transitions_per_env
        nt, ns = self.teacher_num_envs * T, self.student_num_envs * T
        tb, sb = nt // num_mini_batches, ns // num_mini_batches
        ti = torch.randperm(nt, requires_grad=False, device=self.device)
        si = nt + torch.randperm(ns, requires_grad=False, device=self.device)
        return [self.ref2mine[torch.cat([ti[i * tb:(i + 1) * tb], si[i * sb:(i + 1) * sb]])] for i in range(num_mini_batches)]

    def flat(self):
        f = lambda t: t.flatten(0, 1)
        return {"obs": f(self.observations), "cobs": f(self.privileged_observations) if self.privileged_observations is not None else f(self.observations),
                "act": f(self.actions), "hist": f(self.history), "val": f(self.values), "adv": f(self.advantages), "ret": f(self.returns),
                "logp": f(self.actions_log_prob), "mu": f(self.mu), "sig": f(self.sigma)}

    def mini_batch_generator(self, num_mini_batches, num_epochs=8):
        """The reference's generator interface (same tuple order, :162-216); gathers lazily instead of holding 20 batches."""
        fl = self.flat()
        idx = self.mini_batch_indices(num_mini_batches)
        for _ in range(num_epochs):
            for b in idx:
                yield (fl["obs"][b], fl["cobs"][b], fl["act"][b], fl["hist"][b], fl["val"][b], fl["adv"][b], fl["ret"][b], fl["logp"][b], fl["mu"][b], fl["sig"][b],
                       (None, None), None)
