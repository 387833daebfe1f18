"""Forward / backward wall time of the AC-MoE-CTS model at the mini-batch shape, piece by piece (found the N < 32 strided-batched
GEMM cliff and the shared-gate two-stream hang).   python tools/probe_acmoe.py   (GPU)"""
import sys, time, torch
sys.path.insert(0, ".")
from go2_rl_gym_amd.rsl_rl.modules import ActorCriticACMoECTS
torch.manual_seed(0)
m = ActorCriticACMoECTS(45, 263, 12, 4096, 5).cuda()
B, nt = 24576, 18432
obs, priv, hist = torch.randn(B, 45, device="cuda"), torch.randn(B, 263, device="cuda"), torch.randn(B, 225, device="cuda")
def T(label, fn):
    torch.cuda.synchronize(); t0 = time.time(); r = fn(); torch.cuda.synchronize(); print("%-28s %8.1f ms" % (label, 1e3 * (time.time() - t0)), flush=True); return r
for rep in range(2):
    print("rep", rep)
    lat = T("latents", lambda: m.latents(priv, hist, nt))
    xa = torch.cat([lat, obs], 1)
    g = T("gating", lambda: m.actor_moe.gating_network(xa))
    bb = T("actor backbone", lambda: m.actor_moe.experts.backbone(xa))
    eo = T("actor heads", lambda: m.actor_moe.experts.experts(bb))
    mu = T("mixture", lambda: torch.sum(g.unsqueeze(-1) * eo, dim=1))
    v, w = T("value", lambda: m.value(lat, obs, priv))
    loss = (mu ** 2).mean() + (v ** 2).mean() + w.mean()
    T("backward", lambda: loss.backward())
