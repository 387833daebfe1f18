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
# -cuid=go2sim: hipcc derives a "compilation unit id" from the source PATH and bakes it into ~750 bytes of symbol names; fixed, the library is
#   byte-identical wherever the tree is checked out, so the sha256 that keys profiles/*_pmc / *_sq files to a binary (bench.py
#   roofline.lib_sha256_16) is reproducible by anyone who rebuilds (checked: /root/repo vs a copy under /tmp).
STRUCTURAL_FLAGS = ["-fno-slp-vectorize", "-cuid=go2sim"]
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


def stale(out, deps=None):
    """True when `out` is missing or older than any source it is built from — every file of csrc/ and include/ for both libraries: a header added to
    one of them can never be forgotten in a hand-kept list (round 4's go2nn_bx3.h was)."""
    deps = _deps() if deps is None else deps
    return not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(p) for p in deps)


def build_hip(force=False, verbose=False, out=OUT, flags=None):
    OUT, EXTRA_FLAGS = out, (globals()["EXTRA_FLAGS"] if flags is None else flags)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not force and not stale(OUT):
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


NN_SRC = os.path.join(HERE, "csrc", "go2nn_impl.cpp")
NN_OUT = os.path.join(HERE, "libgo2nn_hip.so")


def build_nn(force=False):
    """The policy-side MFMA kernels (include/go2nn.h) -> go2_rl_gym_amd/libgo2nn_hip.so.  A library of its own, so that the sha256 that keys the
    step kernel's counter profiles to libgo2sim_hip.so does not move when this one changes.  No -ffast-math (the head's log-probability and
    ELU follow the eager formulation's arithmetic)."""
    if not force and not stale(NN_OUT):
        return NN_OUT
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-cuid=go2nn", "-o", NN_OUT, NN_SRC]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stderr[-4000:])
    return NN_OUT


def kernel_resources(so_path=OUT, match="go2_step_kernelILi3E"):
    """Register / scratch / LDS use of a kernel as the CODE OBJECT inside the shared library states it (the .note metadata the loader and
    the dispatcher act on): unbundle the gfx950 code object (llvm-objdump --offloading), read its notes (llvm-readelf --notes).
    -> {"vgpr_count" (the unified VGPR + AGPR allocation request), "agpr_count", "arch_vgpr_count", "sgpr_count", "scratch_bytes_per_lane",
        "lds_bytes_per_workgroup", "waves_per_simd"} of the first kernel whose mangled name contains `match`, or None when the LLVM tools are missing."""
    import re
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    objdump, readelf = os.path.join(llvm, "llvm-objdump"), os.path.join(llvm, "llvm-readelf")
    if not (os.path.exists(objdump) and os.path.exists(readelf) and os.path.exists(so_path)):
        return None
    with tempfile.TemporaryDirectory() as td:
        tmp = os.path.join(td, "lib.so")
        shutil.copy(so_path, tmp)
        if subprocess.run([objdump, "--offloading", tmp], capture_output=True, text=True, cwd=td).returncode != 0:
            return None
        co = [f for f in os.listdir(td) if "amdgcn" in f]
        if not co:
            return None
        notes = subprocess.run([readelf, "--notes", os.path.join(td, co[0])], capture_output=True, text=True).stdout
    # the metadata is a YAML list of kernels; one entry runs from "- .agpr_count" (keys are sorted) to the next
    for ent in re.split(r"\n\s*- \.agpr_count:", "\n" + notes)[1:]:
        ent = ".agpr_count:" + ent
        name = re.search(r"\.name:\s+(\S+)", ent)
        if not name or match not in name.group(1):
            continue
        g = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, ent).group(1))
        total, agpr = g("vgpr_count"), g("agpr_count")
        alloc = (total + 7) // 8 * 8
        return {"kernel": name.group(1), "vgpr_count": total, "agpr_count": agpr, "arch_vgpr_count": total - agpr, "sgpr_count": g("sgpr_count"),
                "scratch_bytes_per_lane": g("private_segment_fixed_size"), "lds_bytes_per_workgroup": g("group_segment_fixed_size"),
                "waves_per_simd": max(1, min(8, 512 // alloc))}
    return None


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
    print(kernel_resources())
    print(build_nn(force=True))
    print(kernel_resources(NN_OUT, match="go2nn_mlp_kernel"))
