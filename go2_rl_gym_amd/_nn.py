"""ctypes binding of include/go2nn.h (the policy-side MFMA kernels) and the small host object the algorithms use.

`load_nn()` loads go2_rl_gym_amd/libgo2nn_hip.so and raises if it is missing — like _lib.load_hip there is no CPU fallback in the product;
tests hand `PolicyKernel` the host build (tests/emu/libgo2nn_emu.so) explicitly."""
import ctypes as C
import os

import torch
import torch.nn as nn

_HERE = os.path.dirname(os.path.abspath(__file__))
NN_LIB = os.path.join(_HERE, "libgo2nn_hip.so")
GO2NN_MAX_LAYERS, GO2NN_MAX_WIDTH, GO2NN_ABI_VERSION, GO2NN_MAX_GROUP = 6, 512, 6, 2
_cached = None


class Go2nnSumJob(C.Structure):
    _fields_ = [("part", C.c_void_p), ("out", C.c_void_p), ("nrows", C.c_int32), ("ncols", C.c_int32), ("acc", C.c_void_p), ("nacc", C.c_int32), ("pad_", C.c_int32),
                ("out_w", C.c_int32), ("out_ld", C.c_int32)]          # ABI 6: out_w > 0: the sums go out as rows of out_w floats with pitch out_ld


class Go2nnFwdJob(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("b", C.c_void_p), ("y", C.c_void_p), ("M", C.c_int32), ("K", C.c_int32), ("N", C.c_int32), ("act", C.c_int32),
                ("w_split", C.c_void_p)]          # ABI 4: the weight's split image (go2nn_split_weights) selects the 3 x bf16 kernel; None = fp32 MFMA.  ABI 5: act 1 = no activation


class Go2nnBwdInJob(C.Structure):
    _fields_ = [("gz", C.c_void_p), ("w", C.c_void_p), ("y_prev", C.c_void_p), ("gz_prev", C.c_void_p), ("workspace", C.c_void_p),
                ("M", C.c_int32), ("C", C.c_int32), ("Kin", C.c_int32), ("plain", C.c_int32), ("w_split", C.c_void_p), ("ld", C.c_int32),          # ABI 5: plain 1 = gz W only; ld = pitch of y_prev / gz_prev (0: Kin)
                ("Kx", C.c_int32), ("x_in", C.c_void_p), ("dw_workspace", C.c_void_p), ("ldx", C.c_int32)]          # ABI 6: x_in [M, Kx] = input of the layer below: its weight gradient's partials -> dw_workspace; gz_prev may be None


class Go2nnBwdWJob(C.Structure):
    _fields_ = [("gz", C.c_void_p), ("x", C.c_void_p), ("workspace", C.c_void_p), ("M", C.c_int32), ("C", C.c_int32), ("Kin", C.c_int32), ("split", C.c_int32), ("ldx", C.c_int32)]          # ABI 6: ldx = pitch of x (0: Kin)


class Go2nnSplitJob(C.Structure):
    _fields_ = [("w", C.c_void_p), ("image", C.c_void_p), ("N", C.c_int32), ("K", C.c_int32)]


class Go2nnPpoHeads(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("y_a", "y_c", "w_mu", "b_mu", "w_v", "b_v", "std", "actions", "old_mu", "old_sigma", "old_logp", "adv", "old_values", "returns",
                                          "gz_a", "gz_c", "partials")] + [(k, C.c_int32) for k in ("B", "A", "K", "use_clipped_value_loss")] + \
               [(k, C.c_float) for k in ("clip", "value_loss_coef", "entropy_coef")] + [("surrogate_split", C.c_int32)]          # ABI 5: CTS' teacher rows (0: plain PPO)


class Go2nnMlpIO(C.Structure):
    _fields_ = [("x", C.c_void_p), ("x2", C.c_void_p), ("rows", C.c_void_p), ("y", C.c_void_p)] + [(k, C.c_int32) for k in ("ldx", "ldx2", "kx", "nrows", "ldy", "normalize")]


class Go2nnMlp(C.Structure):
    _fields_ = [("num_layers", C.c_int32), ("dims", C.c_int32 * (GO2NN_MAX_LAYERS + 1)),
                ("weight", C.c_void_p * GO2NN_MAX_LAYERS), ("bias", C.c_void_p * GO2NN_MAX_LAYERS)]


def bind(path):
    lib = C.CDLL(path)
    lib.go2nn_last_error.restype = C.c_char_p
    lib.go2nn_packed_floats.restype = C.c_int64
    lib.go2nn_packed_floats.argtypes = [C.POINTER(Go2nnMlp)]
    lib.go2nn_pack.argtypes = [C.POINTER(Go2nnMlp), C.c_void_p, C.c_void_p]
    lib.go2nn_mlp_arith.argtypes = [C.POINTER(Go2nnMlp)]
    lib.go2nn_mlp_forward.argtypes = [C.POINTER(Go2nnMlp), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.go2nn_policy_act.argtypes = [C.POINTER(Go2nnMlp), C.c_void_p, C.POINTER(Go2nnMlp), C.c_void_p] + [C.c_void_p] * 10 + [C.c_int32, C.c_void_p]
    lib.go2nn_mlp_forward_rows.argtypes = [C.POINTER(C.POINTER(Go2nnMlp)), C.POINTER(C.c_void_p), C.POINTER(Go2nnMlpIO), C.c_int32, C.c_void_p]
    lib.go2nn_policy_act_latent.argtypes = [C.POINTER(Go2nnMlp), C.c_void_p, C.POINTER(Go2nnMlp), C.c_void_p, C.c_void_p, C.c_int32] + [C.c_void_p] * 10 + [C.c_int32, C.c_void_p]
    lib.go2nn_head_backward_workspace.restype = C.c_int64
    lib.go2nn_head_backward_workspace.argtypes = [C.c_int32] * 3
    lib.go2nn_head_backward.argtypes = [C.c_void_p] * 6 + [C.c_int32] * 3 + [C.c_void_p]
    lib.go2nn_linear_elu_forward.argtypes = [C.c_void_p] * 4 + [C.c_int32] * 3 + [C.c_void_p]
    lib.go2nn_linear_backward_workspace.restype = C.c_int64
    lib.go2nn_linear_backward_workspace.argtypes = [C.c_int32] * 3
    lib.go2nn_linear_backward_input.argtypes = [C.c_void_p] * 6 + [C.c_int32] * 3 + [C.c_void_p]
    lib.go2nn_linear_backward_weight.argtypes = [C.c_void_p] * 4 + [C.c_int32] * 3 + [C.c_void_p]
    lib.go2nn_sum_rows.argtypes = [C.POINTER(Go2nnSumJob), C.c_int32, C.c_void_p]
    lib.go2nn_head_backward_rows.argtypes = [C.c_int32] * 3
    lib.go2nn_linear_backward_input_rows.argtypes = [C.c_int32] * 3
    lib.go2nn_linear_elu_forward_group.argtypes = [C.POINTER(Go2nnFwdJob), C.c_int32, C.c_void_p]
    lib.go2nn_linear_backward_input_group_rows.argtypes = [C.c_int32] * 3
    lib.go2nn_linear_backward_input_group.argtypes = [C.POINTER(Go2nnBwdInJob), C.c_int32, C.c_void_p]
    lib.go2nn_linear_backward_input_fused_rows.argtypes = [C.c_int32]
    lib.go2nn_linear_backward_weight_group_rows.argtypes = [C.POINTER(Go2nnBwdWJob), C.c_int32]
    lib.go2nn_linear_backward_weight_group.argtypes = [C.POINTER(Go2nnBwdWJob), C.c_int32, C.c_void_p]
    lib.go2nn_split_weights_bytes.restype = C.c_int64
    lib.go2nn_split_weights_bytes.argtypes = [C.c_int32] * 2
    lib.go2nn_split_weights.argtypes = [C.POINTER(Go2nnSplitJob), C.c_int32, C.c_void_p]
    lib.go2nn_ppo_heads_rows.argtypes = [C.c_int32] * 3
    lib.go2nn_ppo_heads_cols.argtypes = [C.c_int32] * 2
    lib.go2nn_ppo_heads.argtypes = [C.POINTER(Go2nnPpoHeads), C.c_void_p]
    lib.go2nn_l2norm_backward_rows.argtypes = [C.c_int32]
    lib.go2nn_latent_concat.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.go2nn_l2norm_backward.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.go2nn_latent_mse.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p]
    lib.go2nn_moe_usage.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.go2nn_moe_mix_loss.argtypes = [C.c_void_p] * 7 + [C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.go2nn_moe_mix_forward.argtypes = [C.c_void_p] * 5 + [C.c_int32] * 5 + [C.c_void_p]
    if lib.go2nn_abi_version() != GO2NN_ABI_VERSION:
        raise RuntimeError("%s: ABI version %d, expected %d" % (path, lib.go2nn_abi_version(), GO2NN_ABI_VERSION))
    return lib


def load_nn():
    global _cached
    if _cached is None:
        if not os.path.exists(NN_LIB):
            raise RuntimeError("%s is missing: build it with `python -m go2_rl_gym_amd.build` (hipcc --offload-arch=gfx950). There is no CPU fallback." % NN_LIB)
        torch.cuda.is_available()            # torch's HIP runtime first (see _lib.py)
        lib = bind(NN_LIB)
        if lib.go2nn_is_device_library() != 1:
            raise RuntimeError("%s is not the HIP device library" % NN_LIB)
        _cached = lib
    return _cached


def mlp_layers(seq):
    """[(weight, bias)] of an nn.Sequential that is Linear, ELU(alpha=1), ..., Linear (rsl_rl/modules/actor_critic.py:_mlp), else None."""
    mods = list(seq)
    if len(mods) % 2 != 1:
        return None
    out = []
    for k, m in enumerate(mods):
        if k % 2 == 0:
            if not isinstance(m, nn.Linear) or m.bias is None or m.weight.dtype != torch.float32:
                return None
            out.append((m.weight, m.bias))
        elif not (isinstance(m, nn.ELU) and m.alpha == 1.0):
            return None
    dims = [out[0][0].shape[1]] + [w.shape[0] for w, _ in out]
    if len(out) > GO2NN_MAX_LAYERS or max(dims) > GO2NN_MAX_WIDTH:
        return None
    return out


class PackedMlp:
    """One MLP's parameters as the kernels read them: the Go2nnMlp descriptor (pointers to the LIVE parameter tensors) and the packed operand
    buffer, refreshed by pack() — one small launch — whenever the parameters have changed (once per rollout: they only change in update())."""

    def __init__(self, lib, seq, linears=None):
        """seq: an nn.Sequential of Linear / ELU; or linears: the Linear modules of such a stack (wherever they live: nested containers, a normaliser behind them)"""
        layers = mlp_layers(seq) if linears is None else [(m.weight, m.bias) for m in linears]
        if layers is not None and (len(layers) > GO2NN_MAX_LAYERS or max([layers[0][0].shape[1]] + [w.shape[0] for w, _ in layers]) > GO2NN_MAX_WIDTH):
            layers = None
        if layers is None:
            raise ValueError("not a Linear/ELU MLP the kernel supports")
        self.lib, self.layers = lib, layers
        self.desc = Go2nnMlp()
        self.desc.num_layers = len(layers)
        self.desc.dims[0] = layers[0][0].shape[1]
        for l, (w, b) in enumerate(layers):
            assert w.is_contiguous() and b.is_contiguous()
            self.desc.dims[l + 1] = w.shape[0]
            self.desc.weight[l], self.desc.bias[l] = w.data_ptr(), b.data_ptr()
        n = lib.go2nn_packed_floats(C.byref(self.desc))
        if n <= 0:
            raise ValueError(lib.go2nn_last_error().decode())
        self.packed = torch.zeros(int(n), dtype=torch.float32, device=layers[0][0].device)
        self.in_dim, self.out_dim = int(self.desc.dims[0]), int(self.desc.dims[len(layers)])
        self.arith = int(lib.go2nn_mlp_arith(C.byref(self.desc)))          # 3: split-operand kernel, 1: fp32-MFMA kernel, 0: the host test build (include/go2nn.h)

    def _stream(self):
        d = self.packed.device
        return C.c_void_p(torch.cuda.current_stream(d).cuda_stream) if d.type == "cuda" else None

    def still_valid(self):
        return all(self.desc.weight[l] == w.data_ptr() and self.desc.bias[l] == b.data_ptr() for l, (w, b) in enumerate(self.layers))

    def pack(self):
        if not self.still_valid():
            raise RuntimeError("the MLP's parameter tensors were re-allocated after the policy kernel was built (rebuild PolicyKernel)")
        rc = self.lib.go2nn_pack(C.byref(self.desc), C.c_void_p(self.packed.data_ptr()), self._stream())
        if rc != 0:
            raise RuntimeError("go2nn_pack failed: %s" % self.lib.go2nn_last_error().decode())

    def forward(self, x):
        x = x.detach().contiguous().float()
        y = torch.empty(x.shape[0], self.out_dim, dtype=torch.float32, device=x.device)
        rc = self.lib.go2nn_mlp_forward(C.byref(self.desc), C.c_void_p(self.packed.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), x.shape[0], self._stream())
        if rc != 0:
            raise RuntimeError("go2nn_mlp_forward failed: %s" % self.lib.go2nn_last_error().decode())
        return y


class PolicyKernel:
    """PPO.act (rsl_rl/algorithms/ppo.py:90-102) for an ActorCritic of two Linear/ELU MLPs and a state-independent std, as one launch."""

    def __init__(self, lib, actor_critic):
        self.lib, self.ac = lib, actor_critic
        self.actor, self.critic = PackedMlp(lib, actor_critic.actor), PackedMlp(lib, actor_critic.critic)
        if self.actor.out_dim > 32 or self.critic.out_dim != 1:
            raise ValueError("head: up to 32 actions and a scalar value")

    @staticmethod
    def supports(actor_critic):
        if not (hasattr(actor_critic, "actor") and hasattr(actor_critic, "critic") and hasattr(actor_critic, "std") and actor_critic.std.dim() == 1):
            return False
        la, lc = mlp_layers(actor_critic.actor), mlp_layers(actor_critic.critic)
        # the head of go2nn_policy_act: up to 32 actions, a scalar value (the loss head go2sim_ppo_loss takes up to 16 actions; every go2 task has 12)
        return la is not None and lc is not None and la[-1][0].shape[0] <= 32 and lc[-1][0].shape[0] == 1 and actor_critic.std.shape[0] == la[-1][0].shape[0]

    def pack(self):
        self.actor.pack(); self.critic.pack()

    def act(self, obs, critic_obs, eps, a_st=None, mu_st=None, sig_st=None, lp_st=None, v_st=None):
        """-> actions [N, A]; the *_st tensors (storage rows of this step) are filled in place"""
        N, A = obs.shape[0], self.actor.out_dim
        actions = torch.empty(N, A, dtype=torch.float32, device=obs.device)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        for t in (obs, critic_obs, eps, a_st, mu_st, sig_st, lp_st, v_st):
            assert t is None or (t.is_contiguous() and t.dtype == torch.float32), "contiguous float32 tensors"
        rc = self.lib.go2nn_policy_act(C.byref(self.actor.desc), p(self.actor.packed), C.byref(self.critic.desc), p(self.critic.packed), p(obs), p(critic_obs),
                                       p(self.ac.std.detach()), p(eps), p(actions), p(a_st), p(mu_st), p(sig_st), p(lp_st), p(v_st), N, self.actor._stream())
        if rc != 0:
            raise RuntimeError("go2nn_policy_act failed: %s" % self.lib.go2nn_last_error().decode())
        return actions


class PolicyKernelCTS:
    """CTS.act of one rollout step (rsl_rl/rsl_rl/algorithms/cts.py:112-149 over modules/actor_critic_cts.py:146-176) as TWO launches:
    both encoders on their env subsets -> the env-ordered normalised latent (go2nn_mlp_forward_rows), then actor([latent | obs]) + critic([latent | privileged obs]) +
    the sampling head (go2nn_policy_act_latent).  plan: modules/fused_cts.py:CtsPlan (which Linear modules make up the four networks); a model whose student encoder
    is not a plain MLP (the MoE variants) leaves the student rows of the latent to the caller."""

    def __init__(self, lib, model, plan, teacher_idx, student_idx):
        self.lib, self.model, self.L = lib, model, plan.L
        self.enc_t = PackedMlp(lib, None, linears=plan.teacher)
        self.enc_s = PackedMlp(lib, None, linears=plan.student) if plan.student is not None else None
        self.actor, self.critic = PackedMlp(lib, None, linears=plan.actor), PackedMlp(lib, None, linears=plan.critic)
        if self.actor.out_dim > 32 or self.critic.out_dim != 1:
            raise ValueError("head: up to 32 actions and a scalar value")
        self.ti, self.si = teacher_idx.to(torch.int32).contiguous(), student_idx.to(torch.int32).contiguous()
        self._nets = [self.enc_t] + ([self.enc_s] if self.enc_s is not None else [])

    def pack(self):
        for m in self._nets + [self.actor, self.critic]:
            m.pack()

    def _rows(self, nets, ios):
        n = len(nets)
        descs = (C.POINTER(Go2nnMlp) * n)(*[C.pointer(m.desc) for m in nets])
        packed = (C.c_void_p * n)(*[m.packed.data_ptr() for m in nets])
        rc = self.lib.go2nn_mlp_forward_rows(descs, packed, (Go2nnMlpIO * n)(*ios), n, nets[0]._stream())
        if rc != 0:
            raise RuntimeError("go2nn_mlp_forward_rows failed: %s" % self.lib.go2nn_last_error().decode())

    def latents(self, privileged_obs, history, latent):
        """latent [N, L] (env order) <- L2Norm(teacher_encoder(privileged_obs[teacher envs])), L2Norm(student_encoder(history[student envs])) — one launch
        (the student rows only when the student encoder is a plain MLP)"""
        for t in (privileged_obs, history, latent):
            assert t.is_contiguous() and t.dtype == torch.float32
        L = self.L
        ios = [Go2nnMlpIO(privileged_obs.data_ptr(), None, self.ti.data_ptr(), latent.data_ptr(), privileged_obs.shape[1], 0, privileged_obs.shape[1], self.ti.numel(), L, 1)]
        if self.enc_s is not None:
            ios.append(Go2nnMlpIO(history.data_ptr(), None, self.si.data_ptr(), latent.data_ptr(), history.shape[1], 0, history.shape[1], self.si.numel(), L, 1))
        self._rows(self._nets, ios)

    def moe_mix_ok(self, model):
        """the student encoder is a soft mixture whose tail go2nn_moe_mix_forward covers: E <= 16 experts, an L2 normaliser, the latent width the CTS kernels take"""
        enc = getattr(model, "student_moe_encoder", None)
        from .rsl_rl.modules.actor_critic_moe_cts import ActorCriticMoECTS
        from .rsl_rl.modules.utils import L2Norm
        return (enc is not None and getattr(type(model), "student_moe_parts", None) is ActorCriticMoECTS.student_moe_parts and hasattr(enc, "moe") and isinstance(getattr(enc, "norm_layer", None), L2Norm) and enc.moe.experts.expert_num <= 16
                and enc.moe.experts.output_dim == self.L)

    def moe_mix(self, logits, outs, bias, latent, rows=True):
        """latent[student envs] <- normalise(sum_e softmax(logits)_e (outs[e] + bias[e])): logits [n, E], outs [E, n, L] (expert-major, no bias), bias [E * L].
        rows False: latent is [n, L], row r <- source row r (the update's student rows)"""
        E, n, L = outs.shape
        for t in (logits, outs, bias, latent):
            assert t.is_contiguous() and t.dtype == torch.float32
        assert logits.shape == (n, E) and L == self.L and (n == self.si.numel() if rows else latent.shape[0] == n)
        p = lambda t: C.c_void_p(t.data_ptr())
        rc = self.lib.go2nn_moe_mix_forward(p(logits), p(outs), p(bias.detach()), p(self.si) if rows else None, p(latent), latent.shape[1], n, E, L, 1, self.enc_t._stream())
        if rc != 0:
            raise RuntimeError("go2nn_moe_mix_forward failed: %s" % self.lib.go2nn_last_error().decode())

    def value(self, latent, privileged_obs):
        """critic([latent | privileged obs]) -> [N, 1] (the bootstrap value of compute_returns), one launch on the packed weights"""
        N = latent.shape[0]
        v = torch.empty(N, 1, dtype=torch.float32, device=latent.device)
        self._rows([self.critic], [Go2nnMlpIO(latent.data_ptr(), privileged_obs.data_ptr(), None, v.data_ptr(), self.L, privileged_obs.shape[1], self.L, N, 1, 0)])
        return v

    def act(self, latent, obs, privileged_obs, eps, a_st=None, mu_st=None, sig_st=None, lp_st=None, v_st=None):
        N, A = obs.shape[0], self.actor.out_dim
        actions = torch.empty(N, A, dtype=torch.float32, device=obs.device)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        for t in (latent, obs, privileged_obs, eps, a_st, mu_st, sig_st, lp_st, v_st):
            assert t is None or (t.is_contiguous() and t.dtype == torch.float32), "contiguous float32 tensors"
        rc = self.lib.go2nn_policy_act_latent(C.byref(self.actor.desc), p(self.actor.packed), C.byref(self.critic.desc), p(self.critic.packed), p(latent), self.L, p(obs), p(privileged_obs),
                                              p(self.model.std.detach()), p(eps), p(actions), p(a_st), p(mu_st), p(sig_st), p(lp_st), p(v_st), N, self.actor._stream())
        if rc != 0:
            raise RuntimeError("go2nn_policy_act_latent failed: %s" % self.lib.go2nn_last_error().decode())
        return actions
