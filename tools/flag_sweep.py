#!/usr/bin/env python3
"""Compiler-option sweep over the step kernel: the SAME source built with one extra hipcc / -mllvm option per variant, each variant
timed like tools/kbench.py's default line (HIP events inside the library, 4096 envs, random actions) on the plane and on the rough
trimesh terrain, and checked against the product build (one step from identical state).  The product library is not touched.
   python tools/flag_sweep.py build      (here: cross-compiles tests/emu/variants/*.so, prints VGPR / AGPR / scratch of the step kernel)
   python tools/flag_sweep.py            (on the GPU box: -> gpurun_out/flag_sweep.txt)"""
import ctypes as C
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
VDIR = os.path.join(ROOT, "tests", "emu", "variants")

VARIANTS = {
    "baseline": [],
    "sched_max_ilp": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
    "sched_iterative_ilp": ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"],
    "sched_max_memory_clause": ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"],
    "metric_bias_0": ["-mllvm", "-amdgpu-schedule-metric-bias=0"],
    "amdgpu_trackers": ["-mllvm", "-amdgpu-use-amdgpu-trackers=1"],
    "no_high_rp_resched": ["-mllvm", "-amdgpu-disable-unclustered-high-rp-reschedule"],
    "no_post_misched": ["-mllvm", "-enable-post-misched=false"],
    "early_ifcvt": ["-mllvm", "-amdgpu-early-ifcvt=1"],
    "no_dpp_combine": ["-mllvm", "-amdgpu-dpp-combine=false"],
    "O2": ["-O2"],
    "no_loop_align": ["-mllvm", "-amdgpu-disable-loop-alignment"],
}


def build():
    from go2_rl_gym_amd import build as b
    os.makedirs(VDIR, exist_ok=True)
    procs = {}
    for name, extra in VARIANTS.items():
        out = os.path.join(VDIR, "libgo2sim_hip_%s.so" % name)
        cmd = [b.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Rpass-analysis=kernel-resource-usage"] + b.EXTRA_FLAGS + extra + ["-o", out, b.SRC]
        procs[name] = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    for name, p in procs.items():
        _, err = p.communicate()
        if p.returncode != 0:
            print("%-26s FAILED: %s" % (name, err.strip().splitlines()[-1] if err.strip() else "?"))
            try:
                os.remove(os.path.join(VDIR, "libgo2sim_hip_%s.so" % name))
            except OSError:
                pass
            continue
        m = re.search(r"Function Name: _Z15go2_step_kernelILi3E.*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+)", err, re.S)
        print("%-26s %s" % (name, "VGPR %s AGPR %s scratch %s occupancy %s" % m.groups() if m else "built"))


def measure():
    import numpy as np
    import torch
    from go2_rl_gym_amd import _abi
    from helpers import STEP_STATE, DeviceSim, heightfield_overrides, load_hip
    N = 4096
    prod = load_hip()
    rough = heightfield_overrides(N, mesh_type="trimesh")[1]
    lines = []

    def timed(lib, **kw):
        s = DeviceSim(lib, num_envs=N, **kw)
        s.reset_all()
        a = torch.randn(N, 12, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(1)) * 0.5
        for _ in range(80):
            lib.go2sim_step(s.h, C.c_void_p(a.data_ptr()), s._st())
        torch.cuda.synchronize(); lib.go2sim_enable_timing(s.h, 1)
        for _ in range(300):
            lib.go2sim_step(s.h, C.c_void_p(a.data_ptr()), s._st())
        ms, n = C.c_double(), C.c_int64(); lib.go2sim_kernel_time(s.h, C.byref(ms), C.byref(n))
        s.close()
        return 1e3 * ms.value / n.value

    def one_step_diff(lib):
        """max |obs| difference to the product build after one step from identical state (256 envs, settled 30 steps)"""
        sp, sv = DeviceSim(prod, num_envs=256, seed=5), DeviceSim(lib, num_envs=256, seed=5)
        sp.reset_all(); sv.reset_all()
        g = torch.Generator(device="cuda:0").manual_seed(2)
        worst = 0.0
        for it in range(40):
            a = torch.randn(256, 12, device="cuda:0", generator=g)
            for k in STEP_STATE:
                getattr(sv, k)[...] = np.asarray(getattr(sp, k))
            sp.step(a.cpu().numpy()); sv.step(a.cpu().numpy())
            d = np.abs(np.asarray(sp.obs_buf, np.float64) - np.asarray(sv.obs_buf, np.float64)).max(1)
            worst = max(worst, float(np.quantile(d, 0.99)))
        sp.close(); sv.close()
        return worst

    base = None
    for name in VARIANTS:
        path = os.path.join(VDIR, "libgo2sim_hip_%s.so" % name)
        if not os.path.exists(path):
            lines.append("%-26s (not built)" % name); continue
        lib = _abi.bind(path, C.c_float)
        assert lib.go2sim_is_device_library() == 1
        tf = [timed(lib) for _ in range(2)]
        tr = timed(lib, **rough)
        diff = one_step_diff(lib)
        if name == "baseline":
            base = (min(tf), tr)
        lines.append("%-26s plane %.1f us (%.1f)  rough %.1f us   vs baseline %+.1f %% / %+.1f %%   one-step obs p99 diff to product %.1e"
                     % (name, min(tf), max(tf), tr, 100 * (min(tf) / base[0] - 1), 100 * (tr / base[1] - 1), diff))
        print(lines[-1], flush=True)
    tp = [timed(prod) for _ in range(2)]
    lines.append("%-26s plane %.1f us (%.1f)  rough %.1f us" % ("PRODUCT library", min(tp), max(tp), timed(prod, **rough)))
    print(lines[-1])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "flag_sweep.txt"), "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    build() if len(sys.argv) > 1 and sys.argv[1] == "build" else measure()
