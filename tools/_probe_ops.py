"""Which op of one PPO mini-batch step is slow at a large mini-batch?  torch.cuda.Event timing per op for M in argv (rows per mini-batch)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from go2_rl_gym_amd.rsl_rl.runners.on_policy_runner import _enable_tuned_gemms
if "--untuned" not in sys.argv:
    _enable_tuned_gemms()
dev = "cuda:0"
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for M in [int(a) for a in sys.argv[1:] if a.isdigit()]:
    print("M =", M, flush=True)
    for (K, N) in ((263, 512), (45, 512), (512, 256), (256, 128), (128, 12), (128, 1), (128, 32)):
        x, W, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(N, device=dev)
        gy = torch.randn(M, N, device=dev)
        r = {"fwd addmm": t(lambda: torch.addmm(b, x, W.t())), "dgrad mm": t(lambda: gy.mm(W)), "wgrad mm": t(lambda: gy.t().mm(x))}
        if M % 8 == 0:
            r["wgrad bmm8"] = t(lambda: torch.bmm(gy.reshape(8, M // 8, -1).transpose(1, 2), x.reshape(8, M // 8, -1)).sum(0))
        print("   K=%4d N=%4d : " % (K, N) + "  ".join("%s %.3f ms" % kv for kv in r.items()), flush=True)
    idx = torch.randperm(M * 4, device=dev)
    src = torch.randn(M * 4, 263, device=dev); out = torch.empty_like(src)
    print("   index_select [4M,263]: %.3f ms; randperm(4M): %.3f ms" % (t(lambda: torch.index_select(src, 0, idx, out=out)), t(lambda: torch.randperm(M * 4, device=dev))), flush=True)
