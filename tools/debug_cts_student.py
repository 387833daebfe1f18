#!/usr/bin/env python3
"""TOOL: stage-by-stage errors of the CTS student step (modules/fused_cts.py) on the GPU against torch, for one configuration."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import load_hip
from go2_rl_gym_amd import _nn
from go2_rl_gym_amd.rsl_rl.modules import fused, fused_cts
from test_cts_own import make_cts
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
dev = "cuda:0"
model, alg = make_cts(dev, "CTS")
fused.set_library(load_hip()); fused.set_nn_library(_nn.load_nn())
plan = fused_cts.cts_plan(model)
g = torch.Generator().manual_seed(n)
h, p = torch.randn(n, 225, generator=g).to(dev), torch.randn(n, 60, generator=g).to(dev)
k = fused_cts._Launch(torch.device(dev))
st, te = plan.student, plan.teacher
err = lambda a, b: float((a.double() - b.double()).abs().max())
with torch.no_grad():
    imgs = k.images(st + te); is_, it = imgs[:3], imgs[3:]
    sa, ta = [h], [p]
    for l in range(3):
        ys = k.forward([(sa[-1], st[l], is_[l]), (ta[-1], te[l], it[l])], act=1 if l == 2 else 0)
        rs = torch.nn.functional.linear(sa[-1], st[l].weight, st[l].bias); rt = torch.nn.functional.linear(ta[-1], te[l].weight, te[l].bias)
        if l < 2: rs, rt = torch.nn.functional.elu(rs), torch.nn.functional.elu(rt)
        print("fwd pair layer", l, "student", err(ys[0], rs), "teacher", err(ys[1], rt), "shape", tuple(ys[0].shape))
        sa.append(ys[0]); ta.append(ys[1])
    # single-job forwards for comparison
    x = h
    for l in range(3):
        y = k.forward([(x, st[l], is_[l])], act=1 if l == 2 else 0)[0]
        r = torch.nn.functional.linear(x, st[l].weight, st[l].bias)
        if l < 2: r = torch.nn.functional.elu(r)
        print("fwd single layer", l, err(y, r)); x = r
model.zero_grad()
loss, _ = alg._student_losses(h, p); loss.backward()
want = {kk: q.grad.clone() for kk, q in model.named_parameters() if q.grad is not None}
model.zero_grad(set_to_none=True)
fused_cts.cts_student_grads(plan, model, h, p)
torch.cuda.synchronize()
for kk, q in model.named_parameters():
    if kk in want: print(kk, tuple(q.shape), "err", err(q.grad, want[kk]), "scale", float(want[kk].abs().max()))
