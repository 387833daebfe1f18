#!/usr/bin/env python3
"""TOOL (GPU): where do graph mode (no-autograd CTS mini-batch) and eager mode (autograd over the same kernels) part during the FIRST update?  Same rollout, same keyed
permutations, same noise (the set-up of tests/test_gpu_parity.py::test_cts_training_graph_vs_eager_on_gpu); the weights are snapshotted after every optimizer step of
both arms and compared step by step.     python tools/debug_cts_gap.py [task]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from helpers import load_hip

task = sys.argv[1] if len(sys.argv) > 1 else "go2_flat_cts"
hip = load_hip()
from go2_rl_gym_amd.envs import task_registry  # noqa: F401
from go2_rl_gym_amd.rsl_rl.modules.actor_critic_cts import ActorCriticCTS
from go2_rl_gym_amd.utils import get_args

N = 512
res = {}
IDX = []
REF64 = []
PREF64 = []
KEEP = {}
for mode in (False, True):
    args = get_args(["--task", task, "--num_envs", str(N), "--headless", "--seed", "3"])
    env, _ = task_registry.make_env(task, args)
    torch.manual_seed(3)
    _, train_cfg = task_registry.get_cfgs(task)
    sched0 = train_cfg.algorithm.schedule
    train_cfg.algorithm.schedule = "fixed"
    try:
        runner, _ = task_registry.make_alg_runner(env, task, args, train_cfg=train_cfg, log_root=None, use_graphs=mode)
    finally:
        train_cfg.algorithm.schedule = sched0
    alg = runner.alg
    T, A = alg.storage.num_transitions_per_env, alg.storage.actions.shape[-1]
    gen, buf, calls = torch.Generator().manual_seed(17), torch.zeros(T, N, A, device=alg.device), [0]

    def noise(self_, like, buf=buf, calls=calls, T=T):
        row = buf[calls[0] % T]; calls[0] += 1
        return row
    ActorCriticCTS._noise = noise
    snaps, grads, norms = [], [], []
    snap = lambda: {n: p.detach().clone() for n, p in alg.model.named_parameters()}
    gsnap = lambda ps: {id(p): (p.grad.detach().clone() if p.grad is not None else None) for p in ps}
    names = {id(p): n for n, p in alg.model.named_parameters()}
    if not mode:
        st = alg.storage
        key = torch.tensor([int((torch.initial_seed() * 0x9E3779B1 + 0x7F4A7C15) & 0x7FFFFFFF), 0, 0, 0], dtype=torch.int32, device=alg.device)
        order = torch.empty(N * T, dtype=torch.int64, device=alg.device)

        def keyed(nmb, st=st, key=key, order=order, T=T):
            nt, ns = st.teacher_num_envs * T, st.student_num_envs * T
            rc = hip.go2sim_cts_minibatch_indices(C.c_void_p(order.data_ptr()), nmb, nt, ns, C.c_void_p(st.ref2mine.data_ptr()), C.c_void_p(key.data_ptr()),
                                                  C.c_void_p(torch.cuda.current_stream(alg.device).cuda_stream))
            assert rc == 0
            rows = (nt // nmb + ns // nmb)
            out_ = [order[i * rows:(i + 1) * rows].clone() for i in range(nmb)]
            IDX.append(out_)
            return out_
        st.mini_batch_indices = keyed
        for opt, ps in ((alg.optimizer1, alg._params1), (alg.optimizer2, alg._params2)):
            def wrapped(*a, _s=opt.step, _ps=ps, **k):
                torch.cuda.synchronize()
                g = {names[i]: v for i, v in gsnap(_ps).items() if v is not None}     # (after clip_grad_norm_)
                r = _s(*a, **k)
                torch.cuda.synchronize()
                grads.append(g); snaps.append(snap())
                return r
            opt.step = wrapped
    else:
        for nm, ps in (("_policy_back", alg._params1), ("_student_back", alg._params2)):
            def wrapped(*a, _f=getattr(alg, nm), _ps=ps, **k):
                torch.cuda.synchronize()
                g = {names[i]: v for i, v in gsnap(_ps).items() if v is not None}     # (BEFORE the clip: the fused kernel clips inside)
                r = _f(*a, **k)
                torch.cuda.synchronize()
                grads.append(g); snaps.append(snap())
                return r
            setattr(alg, nm, wrapped)
    env.common_step_counter = 0
    buf.copy_(torch.randn(buf.shape, generator=gen))
    w0 = snap()
    # the rollout and the update of iteration 1, with the storage captured in between
    store = {}
    upd = alg.update

    def update(*a, **k):
        torch.cuda.synchronize()
        fl = alg.storage.flat()
        for kk, v in fl.items():
            store[kk] = v.detach().clone()
        return upd(*a, **k)
    alg.update = update
    runner.learn(1, init_at_random_ep_len=True)
    torch.cuda.synchronize()
    res[mode] = (w0, snaps, grads, store)
    if mode:
        KEEP.update(alg=alg, n_t=alg._teacher_rows(), env=env)
        break
    if not mode:
        # float64 reference of the first student step's gradient at the eager arm's weights before it (snapshot 19)
        import copy
        m64 = copy.deepcopy(alg.model).double()
        m64.load_state_dict({k: v.double() for k, v in snaps[19].items()}, strict=False)
        n_t = alg._teacher_rows()
        for sstep in (0, 1):
            bs = IDX[0][sstep][n_t:]
            h64, p64 = store["hist"][bs].double(), store["cobs"][bs].double()
            m64.zero_grad()
            sl, _ = m64.student_latent(h64)
            with torch.no_grad():
                tl = m64.teacher_encoder(p64)
            ((tl - sl) ** 2).mean().backward()
            REF64.append({n: q.grad.detach().clone() for n, q in m64.named_parameters() if q.grad is not None})
            if sstep == 0:
                m64.load_state_dict({k: v.double() for k, v in snaps[20].items()}, strict=False)
        # ... and of policy steps 18 and 19 (mini-batch slots 2 and 3 of the last epoch) through the plain torch formulation of the loss (fused_loss off)
        model32, fl0 = alg.model, alg.fused_loss
        alg.model, alg.fused_loss = m64, False
        for pstep in (18, 19):
            m64.load_state_dict({k: v.double() for k, v in snaps[pstep - 1].items()}, strict=False)
            b = IDX[0][pstep % 4]
            m64.zero_grad()
            out_ = alg._policy_losses(*(store[k][b].double() for k in ("obs", "cobs", "hist", "act", "val", "adv", "ret", "logp", "mu", "sig")), n_t)
            out_[0].backward()
            PREF64.append({n: q.grad.detach().clone() for n, q in m64.named_parameters() if q.grad is not None})
            print("fp64 policy step %d: loss %.6f value %.6f surrogate %.6f entropy %.6f kl %.6f" % (pstep, *[float(x) for x in out_]))
        alg.model, alg.fused_loss = model32, fl0
    env.close()

w0e, se, ge, ste = res[False]
w0g, sg, gg, stg = res[True]
print("initial weights identical:", all(torch.equal(w0e[n], w0g[n]) for n in w0e))
for k in ste:
    if k in stg:
        d = (ste[k].double() - stg[k].double()).abs().max().item()
        print("storage %-6s max|diff| %.3e  shape %s" % (k, d, tuple(ste[k].shape)))
print("optimizer steps: eager %d, graph %d" % (len(se), len(sg)))
for i in range(min(len(se), len(sg))):
    med = {n: float((se[i][n].double() - sg[i][n].double()).abs().median()) for n in se[i]}
    mx = {n: float((se[i][n].double() - sg[i][n].double()).abs().max()) for n in se[i]}
    w = max(med, key=med.get)
    # gradient comparison (eager: after clip; graph: before clip) -> compare directions: relative difference after normalising each to unit total norm
    common = [n for n in ge[i] if n in gg[i]]
    ne = torch.sqrt(sum((ge[i][n].double() ** 2).sum() for n in common)); ng = torch.sqrt(sum((gg[i][n].double() ** 2).sum() for n in common))
    rel = {n: float(((ge[i][n].double() / ne) - (gg[i][n].double() / ng)).abs().max() / ((ge[i][n].double() / ne).abs().max() + 1e-30)) for n in common}
    wr = max(rel, key=rel.get)
    print("step %2d  weights: worst median gap %.2e (%s), worst element %.2e | grad norm eager(after clip) %.4f graph(before clip) %.4f | worst direction diff %.2e (%s)"
          % (i, med[w], w, max(mx.values()), float(ne), float(ng), rel[wr], wr))

for j, step in enumerate((20, 21)):
    print("student step %d against the float64 gradient at the eager arm's weights:" % j)
    for n, r in REF64[j].items():
        if n in ge[step] and n in gg[step]:
            sc = float(r.abs().max())
            print("   %-28s |ref| max %.3e   eager err %.2e   graph err %.2e   (relative to |ref| max)" % (n, sc, float((ge[step][n].double() - r).abs().max()) / sc, float((gg[step][n].double() - r).abs().max()) / sc))
for step in (18, 19):
    print("policy step %d, per tensor: direction difference eager vs graph, relative to the tensor's largest element" % step)
    common = [n for n in ge[step] if n in gg[step]]
    ne = torch.sqrt(sum((ge[step][n].double() ** 2).sum() for n in common)); ng = torch.sqrt(sum((gg[step][n].double() ** 2).sum() for n in common))
    for n in common:
        a, b = ge[step][n].double() / ne, gg[step][n].double() / ng
        print("   %-28s %.2e   (|eager| max %.2e, |graph| max %.2e)" % (n, float((a - b).abs().max() / (a.abs().max() + 1e-30)), float(a.abs().max()), float(b.abs().max())))

for j, step in enumerate((18, 19)):
    print("policy step %d against the float64 gradient at the eager arm's weights (eager: after the clip, graph: before it -> both scaled to the reference's norm over the common tensors):" % step)
    common = [n for n in PREF64[j] if n in ge[step] and n in gg[step]]
    nr = torch.sqrt(sum((PREF64[j][n] ** 2).sum() for n in common))
    ne = torch.sqrt(sum((ge[step][n].double() ** 2).sum() for n in common)); ng = torch.sqrt(sum((gg[step][n].double() ** 2).sum() for n in common))
    print("   norms: ref %.5f eager %.5f graph %.5f" % (float(nr), float(ne), float(ng)))
    for n in common:
        r = PREF64[j][n] / nr; sc = float(r.abs().max())
        print("   %-28s eager err %.2e   graph err %.2e" % (n, float((ge[step][n].double() / ne - r).abs().max()) / sc, float((gg[step][n].double() / ng - r).abs().max()) / sc))

# ---- standalone reproducer: the own policy-step gradient at the EAGER arm's weights before step 19 (and 18), on that step's mini-batch, stage by stage against float64 ----
print("=== reproducer ===")
from go2_rl_gym_amd.rsl_rl.modules import fused, fused_cts
alg, store_g, n_t = KEEP["alg"], res[True][3], KEEP["n_t"]
model, plan = alg.model, alg._plan
L = plan.L
import copy
import torch.nn.functional as F
for pstep in (18, 19):
    model.load_state_dict(se[pstep - 1], strict=False)
    b = IDX[0][pstep % 4]
    T_ = {k: store_g[k][b].contiguous() for k in ("obs", "cobs", "hist", "act", "val", "adv", "ret", "logp", "mu", "sig")}
    B = b.numel()
    ain, cin = torch.zeros(B, L + T_["obs"].shape[1], device=b.device), torch.zeros(B, L + T_["cobs"].shape[1], device=b.device)
    ain[:, L:] = T_["obs"]; cin[:, L:] = T_["cobs"]
    k = fused_cts._Launch(b.device)
    z = fused_cts.encoder_forward(plan.student, T_["hist"][n_t:].contiguous())
    fused_cts.latent_concat(k, z, ain[n_t:], cin[n_t:])
    cap = {}
    orig = fused._Launch.chain_backward

    def spy(self, chains, **kw):
        if len(chains) == 2:
            cap["gz_a"], cap["gz_c"], cap["acts_a"] = chains[0]["gz"].clone(), chains[1]["gz"].clone(), [a.clone() for a in chains[0]["acts"]]
        return orig(self, chains, **kw)
    fused._Launch.chain_backward = spy
    for p_ in model.parameters():
        p_.grad = None
    stats = fused_cts.cts_policy_grads(plan, model, ain, cin, T_["cobs"][:n_t].contiguous(), tuple(T_[k_] for k_ in ("act", "val", "adv", "ret", "logp", "mu", "sig")), n_t,
                                       alg.clip_param, alg.value_loss_coef, alg.entropy_coef, alg.use_clipped_value_loss)
    torch.cuda.synchronize()
    fused._Launch.chain_backward = orig
    own = {n: q.grad.detach().double().clone() for n, q in model.named_parameters() if q.grad is not None}
    ref = PREF64[pstep - 18]
    print("policy step %d reproduced standalone: stats %s" % (pstep, [float(x) for x in stats]))
    for n in ("actor.6.weight", "actor.0.weight", "teacher_encoder.4.weight", "std", "critic.6.weight"):
        sc = float(ref[n].abs().max())
        print("   %-26s own err vs fp64 %.2e (relative to the tensor's largest element)" % (n, float((own[n] - ref[n]).abs().max()) / sc))
    # float64 per-row reference of d loss / d (actor's last hidden pre-activation) and of mu, from the kernel's own input matrix ain (latents included)
    with torch.enable_grad():
        la = plan.actor
        x = ain.double()
        hs = [x]
        for l in range(len(la) - 1):
            zl = F.linear(hs[-1], la[l].weight.double(), la[l].bias.double())
            if l == len(la) - 2:
                zl.retain_grad(); z3 = zl
            hs.append(F.elu(zl))
        mu = F.linear(hs[-1], la[-1].weight.double(), la[-1].bias.double())
        sgm = model.std.double().expand_as(mu)
        lp = (-((T_["act"].double() - mu) ** 2) / (2 * sgm * sgm) - sgm.log() - 0.5 * np.log(2 * np.pi)).sum(-1)
        ratio = torch.exp(lp - T_["logp"].double().squeeze(-1))
        adv = T_["adv"].double().squeeze(-1)
        sl = torch.max(-adv * ratio, -adv * torch.clamp(ratio, 1 - alg.clip_param, 1 + alg.clip_param))
        ent = (0.5 + 0.5 * np.log(2 * np.pi) + sgm.log()).sum(-1)
        loss = sl[:n_t].mean() + sl[n_t:].mean() - alg.entropy_coef * ent.mean()
        loss.backward()
    g64 = z3.grad
    d = (cap["gz_a"].double() - g64).abs().max(1).values
    scale = g64.abs().max(1).values
    bad = torch.nonzero(d > 1e-3 * float(g64.abs().max())).flatten()
    print("   activations: max |h_l - fp64| per layer %s" % ["%.2e" % float((cap["acts_a"][l + 1].double() - hs[l + 1]).abs().max()) for l in range(len(la) - 1)])
    print("   heads' output gradient rows: %d of %d rows differ by more than 1e-3 of the largest element (%.3e); worst rows %s" % (bad.numel(), B, float(g64.abs().max()), bad[:12].tolist()))
    for r_ in bad[:8].tolist():
        print("      row %5d (%s): ratio %.6f adv %+.5f  |g64| %.3e |own| %.3e   lp-olp %.6f" % (r_, "teacher" if r_ < n_t else "student", float(ratio[r_]), float(adv[r_]), float(scale[r_]), float(cap["gz_a"][r_].abs().max()), float(lp[r_] - T_["logp"].double()[r_, 0])))
    print("   ratio: min %.4f max %.4f, outside [0.8, 1.2]: %d rows; rows with |ratio - 0.8| < 1e-5 or |ratio - 1.2| < 1e-5: %d" % (float(ratio.min()), float(ratio.max()), int(((ratio < 0.8) | (ratio > 1.2)).sum()), int((((ratio - 0.8).abs() < 1e-5) | ((ratio - 1.2).abs() < 1e-5)).sum())))

# ---- the rows of step 19 whose probability ratio sits at a clip boundary: on which side are they under the eager arm's weights, and under the graph arm's (float64)? ----
print("=== clip boundary rows, policy step 19 ===")
m64 = copy.deepcopy(model).double()
b = IDX[0][3]
T64 = {k: store_g[k][b].double() for k in ("obs", "cobs", "hist", "act", "logp", "adv")}
rat = {}
for name, W in (("eager", se[18]), ("graph", sg[18])):
    m64.load_state_dict({k: v.double() for k, v in W.items()}, strict=False)
    with torch.no_grad():
        lat = m64.latents(T64["cobs"], T64["hist"], n_t)
        m64.update_distribution(torch.cat([lat, T64["obs"]], dim=1))
        rat[name] = torch.exp(m64.get_actions_log_prob(T64["act"]) - T64["logp"].squeeze(-1))
near = torch.nonzero(((rat["eager"] - 0.8).abs() < 1e-5) | ((rat["eager"] - 1.2).abs() < 1e-5)).flatten()
for r_ in near.tolist():
    print("   row %d: ratio under the eager arm's weights %.9f, under the graph arm's %.9f, advantage %+.5f" % (r_, float(rat["eager"][r_]), float(rat["graph"][r_]), float(T64["adv"][r_, 0])))
print("   largest |ratio difference| between the two weight sets over all rows: %.2e" % float((rat["eager"] - rat["graph"]).abs().max()))
