"""Compile the HIP library (gfx950) in-tree: go2_rl_gym_amd/libgo2sim_hip.so.

hipcc cross-compiles without a GPU, so this runs in the build container; the built .so is git-ignored but
travels with the repo snapshot to the GPU box.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "go2sim_impl.cpp")
OUT = os.path.join(HERE, "libgo2sim_hip.so")
# -ffast-math: rcp/rsq instead of IEEE division sequences in the 6x6 / 3x3 inverses (kernel 174 -> 148 us); parity tests hold
# -fno-slp-vectorize: REQUIRED for correctness with ROCm 7.2's hipcc — with the SLP vectorizer on (-O2 and up), the lane programs that
#   mix packed-f32 candidates with DPP cross-lane moves (go2_lane.h: sliced rows / Gauss-Seidel) are miscompiled: contact impulses come
#   out wrong by orders of magnitude on the device while -O1, or -O3 with this flag, or the same source on the host, agree with the
#   oracle to 1e-5 (bisected on an MI355X, round 2: tools/debug_parity.py over -O1 / -O2 / -O3 x {slp, no-slp, noinline} builds).
#   The GPU parity tests (tests/test_gpu_parity.py::test_one_step_parity_vs_oracle) are what catches a regression here.  Packed f32 VALU
#   is no gain for this latency-bound kernel anyway.
STRUCTURAL_FLAGS = ["-fno-slp-vectorize"]
EXTRA_FLAGS = os.environ.get("GO2_HIPCC_FLAGS", "-ffast-math").split() + STRUCTURAL_FLAGS


def _deps():
    d = [os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc"))]
    d += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    return d


def hipcc_path():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm): the HIP library is the only compute path of this package")


KBENCH_OUT = os.path.join(ROOT, "tests", "emu", "libgo2sim_hip_kbench.so")
PRECISE_OUT = os.path.join(ROOT, "tests", "emu", "libgo2sim_hip_precise.so")


def build_hip_precise(force=False):
    """TEST-ONLY second device build of the same source without -ffast-math (IEEE division / sqrt / no reassociation), next to the host
    emulation under tests/emu/: tests/test_gpu_parity.py uses it to separate fast-math artefacts from fp32 conditioning."""
    return build_hip(force=force, out=PRECISE_OUT, flags=list(STRUCTURAL_FLAGS))


def build_hip_kbench(force=False):
    """TOOL-ONLY device build with the per-wave phase timestamps compiled in (tools/kbench.py, go2sim_debug_clock); the product build carries none."""
    return build_hip(force=force, out=KBENCH_OUT, flags=EXTRA_FLAGS + ["-DGO2_KBENCH_STAMPS"])


def build_hip(force=False, verbose=False, out=OUT, flags=None):
    OUT, EXTRA_FLAGS = out, (globals()["EXTRA_FLAGS"] if flags is None else flags)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= max(os.path.getmtime(p) for p in _deps()):
        return OUT
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + EXTRA_FLAGS + ["-o", OUT, SRC]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stderr[-4000:])
    if verbose:
        print(r.stderr)
    return OUT


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
