"""GroupedHeads (the MoE expert heads as one batched GEMM) forward + backward time for 1 / 12 / 32 output columns: strided-batched fp32
GEMMs with N < 32 are ~100x slower on ROCm 7, hence the zero-padding to 32 columns in modules/utils.py.   python tools/probe_heads.py   (GPU)"""
import sys, time, torch
sys.path.insert(0, ".")
from go2_rl_gym_amd.rsl_rl.modules.utils import GroupedHeads
torch.manual_seed(0)
for cout in (1, 12, 32):
    g = GroupedHeads(8, 128, cout).cuda()
    x = torch.randn(24576, 8 * 128, device="cuda", requires_grad=True)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5):
        y = g(x); y.sum().backward()
    torch.cuda.synchronize()
    print("cout", cout, "5 fwd+bwd: %.1f ms" % (1e3 * (time.time() - t0)), flush=True)
