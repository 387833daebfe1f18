"""Test-side helpers: build + load the CPU oracle and wrap an ABI handle with numpy views.

TEST INFRASTRUCTURE: this is the only place (besides __graft_entry__.smoke and bench.py's cpu_baseline
leg) that loads anything from oracle/.
"""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from go2_rl_gym_amd import _abi

_NP = {C.c_float: np.float32, C.c_double: np.float64, C.c_uint8: np.uint8, C.c_int64: np.int64}


def build_oracle():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)


def load_oracle(f64=False):
    path = os.path.join(ROOT, "oracle", "libgo2oracle_f64.so" if f64 else "libgo2oracle_f32.so")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(ROOT, "oracle", "go2_oracle.c")):
        build_oracle()
    lib = _abi.bind(path, C.c_double if f64 else C.c_float)
    assert lib.go2sim_is_device_library() == 0
    lib.go2o_debug_dynamics.argtypes = [C.c_void_p] + [C.c_int] + [C.c_void_p] * 5
    lib.go2o_pd_torques.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    return lib


class HostSim:
    """A handle on a host-memory library (the oracle) with numpy views of every buffer."""

    def __init__(self, lib, **overrides):
        self.lib, self.abi = lib, lib.abi
        self.real = _NP[self.abi.real]
        cfg = self.abi.Cfg()
        lib.go2sim_default_cfg(C.byref(cfg))
        self._keep = []
        for k, v in overrides.items():
            self.set_cfg(cfg, k, v)
        if "num_envs_global" not in overrides:
            cfg.num_envs_global = cfg.env_offset + cfg.num_envs
        self.cfg = cfg
        h = C.c_void_p()
        _abi.check(lib, lib.go2sim_create(C.byref(cfg), 0, C.byref(h)), "go2sim_create")
        self.h = h
        self.N = cfg.num_envs
        b = self.abi.Buffers()
        _abi.check(lib, lib.go2sim_get_buffers(h, C.byref(b)), "go2sim_get_buffers")
        shapes = _abi.buffer_shapes(self.abi, self.N)
        self.buf = {}
        for name, ctype in self.abi.buffer_fields:
            ptr = getattr(b, name)
            arr = np.ctypeslib.as_array(ptr, shape=shapes[name])
            self.buf[name] = arr
            setattr(self, name, arr)

    def set_cfg(self, cfg, k, v):
        cur = getattr(cfg, k)
        if isinstance(v, np.ndarray) and isinstance(cur, C._Pointer):
            self._keep.append(v)
            setattr(cfg, k, v.ctypes.data_as(type(cur)))
        elif hasattr(cur, "__len__"):
            flat = np.asarray(v, dtype=np.float64).ravel()
            C.memmove(cur, np.ascontiguousarray(flat.astype(_NP.get(self.abi.real) if "float" in str(np.ctypeslib.as_array(cur).dtype) else np.ctypeslib.as_array(cur).dtype)).ctypes.data, C.sizeof(cur))
        else:
            setattr(cfg, k, v)

    def close(self):
        if self.h:
            self.lib.go2sim_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset_all(self):
        _abi.check(self.lib, self.lib.go2sim_reset_all(self.h, None), "reset_all")

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=self.real)
        _abi.check(self.lib, self.lib.go2sim_step(self.h, a.ctypes.data, None), "step")

    def simulate(self):
        _abi.check(self.lib, self.lib.go2sim_simulate(self.h, None), "simulate")

    def post_physics(self):
        _abi.check(self.lib, self.lib.go2sim_post_physics(self.h, None), "post_physics")

    def inject(self, u):
        u = np.ascontiguousarray(u, dtype=self.real)
        assert u.shape == (self.N, self.abi.GO2_NUM_UNIFORMS)
        _abi.check(self.lib, self.lib.go2sim_inject_uniforms(self.h, u.ctypes.data, None), "inject")

    def peek(self):
        out = np.zeros((self.N, self.abi.GO2_NUM_UNIFORMS), dtype=self.real)
        _abi.check(self.lib, self.lib.go2sim_peek_uniforms(self.h, out.ctypes.data, None), "peek")
        return out

    def debug_dynamics(self, e, tau):
        tau = np.ascontiguousarray(tau, dtype=self.real)
        a1 = np.zeros(18, self.real); a2 = np.zeros(18, self.real); M = np.zeros((18, 18), self.real); en = np.zeros(6, self.real)
        self.lib.go2o_debug_dynamics(self.h, e, tau.ctypes.data, a1.ctypes.data, a2.ctypes.data, M.ctypes.data, en.ctypes.data)
        return a1, a2, M, en
