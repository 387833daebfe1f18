"""The VecEnv surface of the reference (legged_gym/envs/base/base_task.py:9-86) on top of the go2sim library.

Where the reference allocates torch buffers and an Isaac Gym sim, this class creates one go2sim handle and
exposes the library-owned buffers as torch tensors (zero copy; device memory for the HIP library).
"""
import ctypes as C

import numpy as np
import torch

from ... import _abi

_TYPESTR = {C.c_float: "<f4", C.c_double: "<f8", C.c_uint8: "|u1", C.c_int64: "<i8"}
_NP = {C.c_float: np.float32, C.c_double: np.float64, C.c_uint8: np.uint8, C.c_int64: np.int64}


class _DeviceArray:
    """Minimal __cuda_array_interface__ carrier so torch can wrap a raw device pointer (strided)."""

    def __init__(self, ptr, shape, typestr, strides):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2, "strides": tuple(strides)}


def wrap_buffers(lib, handle, N, device):
    """dict name -> torch tensor viewing the library's buffers with their LOGICAL shapes (the HIP library stores
    per-env fields field-major, so most views are strided transposes)."""
    abi = lib.abi
    b = abi.Buffers()
    _abi.check(lib, lib.go2sim_get_buffers(handle, C.byref(b)), "go2sim_get_buffers")
    shapes = _abi.buffer_shapes(abi, N)
    layout = lib.go2sim_buffer_layout()
    on_device = lib.go2sim_is_device_library() == 1
    out = {}
    for name, ctype in abi.buffer_fields:
        ptr = getattr(b, name)
        base = ctype._type_
        shp = shapes[name]
        itemsize = C.sizeof(base)
        transposed = layout == 1 and name not in _abi.ROW_MAJOR_ALWAYS and len(shp) > 1
        phys = tuple(reversed(shp)) if transposed else shp
        # C-order strides of the physical array, then permuted back to the logical order
        st = [itemsize] * len(phys)
        for i in range(len(phys) - 2, -1, -1):
            st[i] = st[i + 1] * phys[i + 1]
        if transposed:                       # NOT `phys != shp`: a (12, 12) buffer (num_envs == 12) is transposed too
            st = list(reversed(st))
        if on_device:
            addr = C.cast(ptr, C.c_void_p).value
            t = torch.as_tensor(_DeviceArray(addr, shp, _TYPESTR[base], st), device=device)
        else:
            arr = np.ctypeslib.as_array(ptr, shape=phys)
            if transposed:
                arr = arr.transpose()
            t = torch.from_numpy(arr)
        out[name] = t
    return out


class BaseTask:
    def __init__(self, cfg, sim_params, physics_engine, sim_device, headless, lib=None):
        self.sim_params = sim_params
        self.physics_engine = physics_engine
        self.sim_device = sim_device
        self.headless = headless
        if lib is None:
            # product path: the HIP library or nothing
            if not str(sim_device).startswith("cuda"):
                raise RuntimeError("go2_rl_gym_amd simulates on the GPU only (sim_device=%r): there is no CPU simulation path in the product" % (sim_device,))
            from ... import _lib
            lib = _lib.load_hip()
            self.device = sim_device
        else:
            self.device = sim_device if lib.go2sim_is_device_library() == 1 else "cpu"
        self.lib = lib
        self.abi = lib.abi
        self.num_envs = cfg.env.num_envs
        self.num_obs = cfg.env.num_observations
        self.num_privileged_obs = cfg.env.num_privileged_obs
        self.num_actions = cfg.env.num_actions
        self.extras = {}
        self.handle = None
        self.create_sim()
        self.viewer = None      # headless only (the reference's viewer, base_task.py:62-70, is out of scope)

    # -- the four methods of rsl_rl's VecEnv (rsl_rl/env/vec_env.py:36-59) --
    def get_observations(self):
        return self.obs_buf

    def get_privileged_observations(self):
        return self.privileged_obs_buf

    def reset_idx(self, env_ids):
        raise NotImplementedError

    def reset(self):
        """Reset all robots, then one step with zero actions (base_task.py:82-86)."""
        self.reset_idx(torch.arange(self.num_envs, device=self.device))
        obs, privileged_obs, _, _, _ = self.step(torch.zeros(self.num_envs, self.num_actions, device=self.device, requires_grad=False))
        return obs, privileged_obs

    def step(self, actions):
        raise NotImplementedError

    def render(self, sync_frame_time=True):
        return None

    def _stream(self):
        if self.lib.go2sim_is_device_library() == 1:
            return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return None

    def close(self):
        if self.handle is not None:
            self.lib.go2sim_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
