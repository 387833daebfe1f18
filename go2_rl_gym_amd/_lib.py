"""Loader of the product's compute library.  There is exactly one: the HIP library for gfx950.

No CPU fallback exists in the product: if the library is missing, cannot be loaded, or is not a device
library, this raises.  (Tests may hand an explicit `lib=` — the CPU oracle or the lane emulation — to the
environment constructor; that path is never taken by train.py / play.py / bench.py's GPU legs.)
"""
import ctypes as C
import os

# PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so).  It must be mapped BEFORE our library is
# dlopen'ed so that both resolve the same libamdhip64 SONAME to ONE runtime; loading the system copy first leaves
# torch with "No HIP GPUs are available" (two runtimes in one process).
import torch  # noqa: F401  (side effect: maps torch's HIP runtime)

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB = os.path.join(_HERE, "libgo2sim_hip.so")
_cached = None


def load_hip():
    global _cached
    if _cached is not None:
        return _cached
    if not os.path.exists(HIP_LIB):
        raise RuntimeError("%s is missing: build it with `python -m go2_rl_gym_amd.build` (hipcc --offload-arch=gfx950). "
                           "There is no CPU fallback." % HIP_LIB)
    torch.cuda.is_available()            # make sure torch's libamdhip64 is resolved first
    lib = _abi.bind(HIP_LIB, C.c_float)
    if lib.go2sim_is_device_library() != 1:
        raise RuntimeError("%s is not the HIP device library" % HIP_LIB)
    _cached = lib
    return lib
