#!/usr/bin/env python3
"""One policy step of the rollout (PPO.act: actor + critic MLPs + sampling head) — the fp32-MFMA kernel of include/go2nn.h against what it
replaces (two hipBLASLt GEMM chains on two streams + go2sim_act_head), both replayed from HIP graphs as in the rollout.   python tools/policy_bench.py [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from go2_rl_gym_amd import _nn
from go2_rl_gym_amd.rsl_rl.modules.actor_critic import ActorCritic
from go2_rl_gym_amd.rsl_rl.runners.on_policy_runner import _enable_tuned_gemms
_enable_tuned_gemms()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = "cuda:0"
ac = ActorCritic(45, 263, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], activation="elu").to(dev)
obs, priv, eps = torch.randn(N, 45, device=dev), torch.randn(N, 263, device=dev), torch.randn(N, 12, device=dev)
pk = _nn.PolicyKernel(_nn.bind(os.environ["GO2NN_LIB"]) if os.environ.get("GO2NN_LIB") else _nn.load_nn(), ac); pk.pack()      # GO2NN_LIB: a build variant (A/B of kernel edits)
rows = [torch.zeros(N, 12, device=dev) for _ in range(3)] + [torch.zeros(N, device=dev), torch.zeros(N, device=dev)]
side = torch.cuda.Stream()
def torch_path():
    with torch.inference_mode():
        cur = torch.cuda.current_stream(); side.wait_stream(cur)
        with torch.cuda.stream(side):
            v = ac.critic(priv)
        mu = ac.actor(obs)
        cur.wait_stream(side)
        a = mu + ac.std * eps
        lp = (-(a - mu) ** 2 / (2 * ac.std ** 2) - ac.std.log() - 0.9189385).sum(-1)
        return a, lp, v
def fused():
    return pk.act(obs, priv, eps, *rows)
def graph_time(fn, reps=24, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * iters)
print("N = %d rows: torch chains + elementwise head %.1f us per policy step | fused MFMA kernel %.1f us | pack (once per rollout) %.1f us" %
      (N, graph_time(torch_path), graph_time(fused), graph_time(pk.pack)))
if "--stamps" in sys.argv:      # per-workgroup phase timestamps (a -DGO2NN_STAMPS build of the library under build/variants/)
    import ctypes as C
    lib = _nn.bind(os.environ.get("GO2NN_STAMPS_LIB") or os.path.join(ROOT, "build", "variants", "libgo2nn_stamps.so"))
    pk2 = _nn.PolicyKernel(lib, ac); pk2.pack()
    nwg = (N + 31) // 32
    buf = torch.zeros(2 * nwg, 8, dtype=torch.int64, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    for _ in range(3):
        a_out = torch.empty(N, 12, device=dev)
        lib.go2nn_policy_act(C.byref(pk2.actor.desc), p(pk2.actor.packed), C.byref(pk2.critic.desc), p(pk2.critic.packed), p(obs), p(priv), p(ac.std.detach()), p(eps), p(a_out),
                             None, None, None, None, None, N, None) if False else None
    # the stamped build takes the stamp buffer through the (otherwise unused in this mode) y pointer: call the raw entry point
    lib.go2nn_policy_act_stamped.argtypes = [C.POINTER(_nn.Go2nnMlp), C.c_void_p, C.POINTER(_nn.Go2nnMlp), C.c_void_p] + [C.c_void_p] * 6 + [C.c_int32, C.c_void_p]
    for _ in range(5):
        lib.go2nn_policy_act_stamped(C.byref(pk2.actor.desc), p(pk2.actor.packed), C.byref(pk2.critic.desc), p(pk2.critic.packed), p(obs), p(priv), p(ac.std.detach()), p(eps), p(a_out), p(buf), N, None)
        torch.cuda.synchronize()
    t = buf.cpu().numpy().astype("float64") / 100.0      # us (100 MHz constant clock)
    for y, name in ((0, "actor"), (1, "critic")):
        d = t[y * nwg:(y + 1) * nwg]
        print("%s workgroups: stage %.1f us | layers %s us | total mean %.1f max %.1f us; first start -> last end %.1f us" %
              (name, (d[:, 1] - d[:, 0]).mean(), " ".join("%.1f" % (d[:, 2 + l] - d[:, 1 + l]).mean() for l in range(4)), (d[:, 5] - d[:, 0]).mean(), (d[:, 5] - d[:, 0]).max(), d[:, 5].max() - t[:, 0].min()))
        if d[:, 6].any():      # (the split-operand kernel: k-loop and epilogue of the wave that holds thread 0, per layer, 10 ns units packed 4 x 16 bits per word)
            raw = buf.cpu().numpy()[y * nwg:(y + 1) * nwg, 6:8]
            f = lambda l, part: ((raw[:, l >> 1] >> (32 * (l & 1) + 16 * part)) & 0xffff)
            print("      thread 0's wave, layers 1-4: k-loop %s us | epilogue %s us (mean over the workgroups where that wave had a tile)" %
                  (" ".join("%.1f" % (f(l, 0)[f(l, 0) > 0].mean() / 100.0 if (f(l, 0) > 0).any() else 0) for l in range(4)),
                   " ".join("%.1f" % (f(l, 1)[f(l, 0) > 0].mean() / 100.0 if (f(l, 0) > 0).any() else 0) for l in range(4))))
