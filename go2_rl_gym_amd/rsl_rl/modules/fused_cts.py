"""A CTS mini-batch (rsl_rl/rsl_rl/algorithms/cts.py:167-285) as explicit launches of include/go2nn.h — no autograd graph, no vendor GEMM.

The policy step differentiates   teacher_encoder -> L2Norm -> [latent | obs] -> actor,   [latent.detach() | privileged obs] -> critic   through the PPO loss with the
split surrogate (cts.py:228-231); the student step differentiates the student encoder through the latent MSE (cts.py:259-275).  Both are chains of Linear / ELU layers
around three row-wise pieces (the normaliser forward, its backward, the MSE head), so they are the grouped split-operand GEMMs of PPO's mini-batch
(modules/fused.py:ppo_pair_grads) plus go2nn_latent_concat / go2nn_l2norm_backward / go2nn_latent_mse (include/go2nn.h ABI 5):

  policy step  (26 launches):  split weights | teacher encoder forward x3 | latent_concat (zhat straight into the first L columns of both input matrices) |
               actor + critic forward x3 (grouped) | go2nn_ppo_heads (surrogate_split = teacher rows) | weight gradients x3 + input gradients x2 (grouped) |
               the actor's plain input gradient on the teacher rows | l2norm_backward | teacher encoder weight gradients x3 + input gradients x2 | ONE go2nn_sum_rows
  student step (13 launches):  split weights | student + teacher encoder forward x3 (grouped) | latent_mse | student weight gradients x3 + input gradients x2 | sum_rows

What makes the policy step cheap beyond the kernels: optimizer1 never touches the student encoder, so the student rows' latents are constants of an update — they are
written into the input matrices ONCE per update (algorithms/cts.py), not recomputed in each of the 20 policy steps; and obs / privileged obs are gathered straight into
the column blocks behind the latent (go2sim_shuffle_gather's dst_pitch), so no torch.cat ever runs.

Every parameter's .grad is set (replaced, as after zero_grad(set_to_none=True)); all sums have a fixed order."""
import ctypes as C

import torch
import torch.nn as nn

from . import fused


def _leaves(m):
    """the leaf modules of nested nn.Sequential / MLP (modules/utils.py: .network) containers, in order"""
    if isinstance(m, nn.Sequential):
        return [x for c in m for x in _leaves(c)]
    if isinstance(getattr(m, "network", None), nn.Sequential):
        return _leaves(m.network)
    return [m]


def mlp_linears(m, norm=False):
    """[Linear, ELU(1)] x H + Linear (+ an L2Norm when `norm`) with H >= 1 -> the Linear modules, else None"""
    from .utils import L2Norm
    mods = _leaves(m)
    if norm:
        if not mods or not isinstance(mods[-1], L2Norm):
            return None
        mods = mods[:-1]
    if len(mods) < 3 or len(mods) % 2 == 0:
        return None
    lins = []
    for k, x in enumerate(mods):
        if k % 2 == 0:
            if not (isinstance(x, nn.Linear) and x.bias is not None and x.weight.dtype == torch.float32 and x.weight.requires_grad and x.bias.requires_grad
                    and x.in_features >= 4 and x.weight.is_contiguous()):
                return None
            lins.append(x)
        elif not (isinstance(x, nn.ELU) and x.alpha == 1.0):
            return None
    if any(l.out_features % 4 or l.out_features < 4 for l in lins[:-1]):
        return None
    return lins


class CtsPlan:
    """Which Linear modules make up the teacher encoder, the actor, the critic (and the student encoder, when it is a plain MLP) of an ActorCriticCTS-family model."""

    def __init__(self, teacher, actor, critic, student, L):
        self.teacher, self.actor, self.critic, self.student, self.L = teacher, actor, critic, student, L


def _latent_ok(L):
    return L >= 4 and L % 4 == 0 and L <= 128 and ((L // 4) & (L // 4 - 1)) == 0


def cts_plan(model):
    """-> CtsPlan when the no-autograd mini-batch applies to `model`, else None (the algorithm then keeps the autograd formulation): the library pair is loaded, the
    heads are the base class's ([latent | obs] -> actor MLP, [latent | privileged obs] -> critic MLP, one std per action), every network is Linear / ELU with an
    L2-normalised latent, actor and critic share their hidden widths.  (Round 6: also with GO2_GEMM_SPLIT=0, the fp32-MFMA A/B arithmetic — the one product those kernels do
    not have, the plain input gradient of a layer whose width is not a multiple of 4 ([latent | obs] = 77), is then a vendor mm: _plain_input_grad.)"""
    from .actor_critic_cts import ActorCriticCTS
    if not (fused._LIB is not None and fused._NN is not None and isinstance(model, ActorCriticCTS)):
        return None
    t = type(model)
    if any(getattr(t, n) is not getattr(ActorCriticCTS, n) for n in ("policy_mean", "value", "policy_dist", "latents", "evaluate_joint")) or model.state_dependent_std:
        return None
    te, la, lc = mlp_linears(model.teacher_encoder, norm=True), mlp_linears(model.actor), mlp_linears(model.critic)
    if te is None or la is None or lc is None or len(la) != len(lc) or any(a.out_features != c.out_features for a, c in zip(la[:-1], lc[:-1])):
        return None
    L, K = te[-1].out_features, la[-1].in_features
    if not (_latent_ok(L) and lc[-1].out_features == 1 and la[-1].out_features <= 16 and K % 4 == 0 and K <= 256 and model.std.dim() == 1
            and model.std.shape[0] == la[-1].out_features and model.std.requires_grad and la[0].in_features > L and lc[0].in_features > L):
        return None
    st = None
    if t.student_latent is ActorCriticCTS.student_latent and hasattr(model, "student_encoder"):
        st = mlp_linears(model.student_encoder, norm=True)
        if st is not None and (st[-1].out_features != L or len(st) != len(te)):
            # (declined sub-paths are said once — the student step then runs under autograd and the rollout's student encoder as torch modules)
            _note("CTS own path: the student encoder (%d layers -> %d) does not pair with the teacher encoder (%d layers -> %d) in the grouped launches; "
                  "its update step stays on the autograd formulation" % (len(st), st[-1].out_features, len(te), L))
            st = None
    return CtsPlan(te, la, lc, st, L)


_NOTED = set()


def _note(msg):
    if msg not in _NOTED:
        _NOTED.add(msg)
        print("[go2_rl_gym_amd] " + msg)


_Launch = fused._Launch


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def latent_concat(k, z, dst_a, dst_b, inv=None):
    """zhat = z / max(|z|, 1e-12) into the first L columns of dst_a / dst_b (row-major, any pitch >= L; either may be None); inv [n] <- 1 / max(|z|, 1e-12)"""
    n, L = z.shape
    k.check(k.nn.go2nn_latent_concat(_p(z), n, L, _p(dst_a), dst_a.stride(0) if dst_a is not None else 0, _p(dst_b), dst_b.stride(0) if dst_b is not None else 0, _p(inv), k.stream),
            "go2nn_latent_concat")


def _plain_input_grad(k, gz0, lin0, img0):
    """d loss / d x = gz0 W0 of a network's first layer: the split-operand kernel (any width), or — fp32-MFMA arithmetic (GO2_GEMM_SPLIT=0, A/B runs) and a width that is
    not a multiple of 4 — the vendor's mm, as modules/fused.py:_input_grad does for the autograd nodes"""
    if img0 is not None or (lin0.in_features % 4 == 0 and lin0.out_features >= 4):
        return k.bwd_in([(gz0, lin0, None, img0)], plain=True)[0][0]
    return gz0.mm(lin0.weight)


def cts_policy_grads(plan, model, ain, cin, priv_t, batch, n_t, clip, vcoef, ecoef, use_clipped_value_loss, acc=None):
    """One CTS policy mini-batch gradient (cts.py:180-250 + loss.backward()).
    ain [B, L + obs], cin [B, L + priv]: the actor's / critic's input matrices, rows [0, n_t) teacher samples, the rest student samples; columns [L, ...) hold obs /
    privileged obs, the STUDENT rows' first L columns the (constant) student latents; the teacher rows' first L columns are written here.  priv_t [n_t, priv]: the
    teacher rows' privileged observations (the teacher encoder's input).  batch: actions, old values, advantages, returns, old log-probs, old mu, old sigma.
    acc: optional float32[>= 4] — [surrogate, value loss, KL, entropy] of the mini-batch are ADDED to it by the pass's go2nn_sum_rows launch.
    -> stats [surrogate, value loss, KL, entropy] (device tensor; valid after the launches)"""
    k = _Launch(ain.device)
    nn_ = k.nn
    te, la, lc, L = plan.teacher, plan.actor, plan.critic, plan.L
    H = len(la) - 1
    cont = lambda t: t.detach() if t.is_contiguous() else t.detach().contiguous()
    with torch.no_grad():
        imgs = k.images(te + la[:H] + lc[:H])
        ie, ia, ic = imgs[:len(te)], imgs[len(te):len(te) + H], imgs[len(te) + H:]
        # teacher encoder on the teacher rows -> zhat into both input matrices
        eacts = [cont(priv_t)]
        for l, m in enumerate(te):
            eacts.append(k.forward([(eacts[-1], m, ie[l])], act=1 if l == len(te) - 1 else 0)[0])
        inv = k.new(n_t)
        latent_concat(k, eacts[-1], ain, cin, inv)
        # actor + critic: grouped hidden layers, heads + PPO loss with the split surrogate, backward down to the first layers' pre-activations
        tot, gz1 = fused.pair_grads(k, la, lc, ain, cin, model.std, batch, clip, vcoef, ecoef, use_clipped_value_loss, surrogate_split=n_t, acc=acc, imgs=(ia, ic))
        # into the teacher encoder: d loss / d [latent | obs] of the actor on the teacher rows (the critic sees latent.detach()), through the normaliser
        g_in = _plain_input_grad(k, gz1[0][:n_t], la[0], ia[0])
        r = nn_.go2nn_l2norm_backward_rows(n_t)
        dz, zpart, gb_z = k.new(n_t, L), k.new(r * L), k.new(L)
        k.check(nn_.go2nn_l2norm_backward(_p(g_in), g_in.stride(0), _p(ain), ain.stride(0), _p(inv), _p(dz), _p(zpart), n_t, L, k.stream), "go2nn_l2norm_backward")
        k.sums.append((zpart, gb_z, r, L))
        k.chain_backward([{"lins": te, "acts": eacts, "gz": dz, "gb": gb_z, "imgs": ie}], need_gz0=False)
        k.finish()
    return tot[:4]


def cts_student_grads(plan, model, hist_s, priv_s, acc=None):
    """One CTS student mini-batch gradient (cts.py:259-275 + latent_loss.backward()): hist_s [n_s, H * obs], priv_s [n_s, priv] = the student rows of the mini-batch.
    acc: optional float32[>= 1] — the latent loss is ADDED to acc[0].  -> the latent loss (1-element device tensor; valid after the launches)"""
    k = _Launch(hist_s.device)
    nn_ = k.nn
    st, te, L = plan.student, plan.teacher, plan.L
    n = hist_s.shape[0]
    cont = lambda t: t.detach() if t.is_contiguous() else t.detach().contiguous()
    with torch.no_grad():
        imgs = k.images(st + te)
        is_, it = imgs[:len(st)], imgs[len(st):]
        sacts, tx = [cont(hist_s)], cont(priv_s)
        for l in range(len(st)):
            last = 1 if l == len(st) - 1 else 0
            ys = k.forward([(sacts[-1], st[l], is_[l]), (tx, te[l], it[l])], act=last)
            sacts.append(ys[0]); tx = ys[1]
        r = nn_.go2nn_l2norm_backward_rows(n)
        dz, part, tot = k.new(n, L), k.new(r * (L + 4)), k.new(L + 4)          # [loss, 0, 0, 0 | the last bias gradient]
        k.check(nn_.go2nn_latent_mse(_p(sacts[-1]), _p(tx), _p(dz), _p(part), n, L, 1.0, k.stream), "go2nn_latent_mse")
        k.sums.append((part, tot, r, L + 4, acc, 1))
        k.chain_backward([{"lins": st, "acts": sacts, "gz": dz, "gb": tot[4:], "imgs": is_}], need_gz0=False)
        k.finish()
    return tot[:1]


def encoder_forward(lins, x):
    """the un-normalised output z = MLP(x) of a plain encoder (forward only, own kernels: one split launch + one launch per layer)"""
    k = _Launch(x.device)
    with torch.no_grad():
        imgs = k.images(lins)
        h = x if x.is_contiguous() else x.contiguous()
        for l, m in enumerate(lins):
            h = k.forward([(h, m, imgs[l])], act=1 if l == len(lins) - 1 else 0)[0]
    return h


def encoder_latents(plan, lins, x, dst_a, dst_b):
    """zhat = L2Norm(MLP(x)) of a plain encoder straight into the first L columns of dst_a / dst_b — forward only, own kernels"""
    z = encoder_forward(lins, x)
    latent_concat(_Launch(x.device), z, dst_a, dst_b)
    return z


def moe_head_grads(logits, outs, t_hat, lb_coef, acc=None, expert_major=False, bias=None):
    """The loss head of the MoE student step (moe_cts.py:203-214 over modules/utils.py:96-152) with its analytic gradients, as two launches + the reductions:
    logits [n, E] (the gate before its softmax), outs [n, E, L] (the experts' outputs; [E, n, L] when expert_major — the batched GEMM's own output layout, so
    that neither the forward nor the backward pass transposes a [n, E, L] tensor), t_hat [n, L] (the teacher's normalised latent).
    bias (optional, [E * L]): the expert heads' output bias when `outs` comes without it (a plain bmm under autograd): added by the kernel, its gradient returned fourth.
    acc: optional float32[>= 2] — latent loss and load-balance loss are ADDED to acc[0:2].  -> stats [latent loss, load balance], d loss / d logits, d loss / d outs(, d loss / d bias)"""
    k = _Launch(logits.device)
    nn_ = k.nn
    n, E = logits.shape
    L = outs.shape[2]
    assert tuple(outs.shape) == ((E, n, L) if expert_major else (n, E, L))
    cont = lambda t: t.detach() if t.is_contiguous() else t.detach().contiguous()
    logits, outs, t_hat = cont(logits), cont(outs), cont(t_hat)
    with torch.no_grad():
        r = nn_.go2nn_l2norm_backward_rows(n)
        upart, usage = k.new(r * E), k.new(E)
        k.check(nn_.go2nn_moe_usage(_p(logits), _p(upart), n, E, k.stream), "go2nn_moe_usage")
        k.sums.append((upart, usage, r, E))
        k.finish()
        dl, do, part, tot = k.new(n, E), torch.empty_like(outs), k.new(r * 4), k.new(4)
        bpart = dbias = None
        if bias is not None:
            bias = cont(bias).reshape(-1)
            bpart, dbias = k.new(r * E * L), k.new(E * L)
            k.sums.append((bpart, dbias, r, E * L))
        k.check(nn_.go2nn_moe_mix_loss(_p(logits), _p(outs), _p(t_hat), _p(usage), _p(dl), _p(do), _p(part), n, E, L, float(lb_coef), 1 if expert_major else 0, _p(bias), _p(bpart), k.stream),
                "go2nn_moe_mix_loss")
        k.sums.append((part, tot, r, 4, acc, 2))
        k.finish()
    return (tot[:2], dl, do) if bias is None else (tot[:2], dl, do, dbias)


def moe_head_applicable(model, L):
    """the fused MoE loss head needs the library pair, an L2-normalised mixture and the model's `student_moe_parts` hook (gate logits and expert outputs before the mix)"""
    from .utils import L2Norm
    enc = getattr(model, "student_moe_encoder", None)
    return (fused._LIB is not None and fused._NN is not None and hasattr(model, "student_moe_parts") and enc is not None and isinstance(getattr(enc, "norm_layer", None), L2Norm)
            and _latent_ok(L) and getattr(enc, "expert_num", getattr(getattr(enc, "moe", None), "expert_num", 99)) <= 16)

