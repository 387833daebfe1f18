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


def build_emu():
    out = os.path.join(ROOT, "tests", "emu", "libgo2sim_emu.so")
    src = os.path.join(ROOT, "go2_rl_gym_amd", "csrc")
    deps = [os.path.join(src, f) for f in os.listdir(src)] + [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.run(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-DGO2_EMU", "-w", "-o", out, os.path.join(src, "go2sim_impl.cpp")], check=True)
    return out


def load_emu():
    """Host emulation of the HIP lane programs (TEST ONLY: same source as the kernels, compiled by g++)."""
    lib = _abi.bind(build_emu(), C.c_float)
    assert lib.go2sim_is_device_library() == 0 and lib.go2sim_buffer_layout() == 1
    return lib


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
        self._wrap(h, cfg.num_envs)

    @classmethod
    def attach(cls, env):
        """View the simulator a LeggedRobot (host layer) created — its config translation is then part of what the test checks."""
        self = cls.__new__(cls)
        self.lib, self.abi, self.real, self._keep, self.cfg, self._env = env.lib, env.lib.abi, _NP[env.lib.abi.real], [], env._c, env
        self._wrap(env.handle, env.num_envs)
        self.close = lambda: None            # the env owns the handle
        return self

    def _wrap(self, h, N):
        lib = self.lib
        self.h = h
        self.N = N
        b = self.abi.Buffers()
        _abi.check(lib, lib.go2sim_get_buffers(h, C.byref(b)), "go2sim_get_buffers")
        shapes = _abi.buffer_shapes(self.abi, self.N)
        self.layout = lib.go2sim_buffer_layout()
        self.buf = {}
        for name, ctype in self.abi.buffer_fields:
            ptr = getattr(b, name)
            shp = shapes[name]
            if self.layout == 1 and name not in _abi.ROW_MAJOR_ALWAYS and len(shp) > 1:
                arr = np.ctypeslib.as_array(ptr, shape=tuple(reversed(shp))).transpose()   # field-major storage, logical view
            else:
                arr = np.ctypeslib.as_array(ptr, shape=shp)
            self.buf[name] = arr
            setattr(self, name, arr)

    def set_cfg(self, cfg, k, v):
        cur = getattr(cfg, k)
        if isinstance(v, np.ndarray) and isinstance(cur, C._Pointer):
            if v.dtype.kind == "f":         # the fp64 oracle build widens every `float*` of the ABI (terrain_origins) to double*
                v = np.ascontiguousarray(v, _NP[type(cur)._type_])
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

    def reset_idx(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        _abi.check(self.lib, self.lib.go2sim_reset_idx(self.h, ids.ctypes.data, len(ids), None), "reset_idx")

    def step_rollout(self, actions, values, gamma):
        """go2sim_step_rollout with every output redirected -> dict of what it wrote"""
        A, N, r = self.abi, self.N, self.real
        a = np.ascontiguousarray(actions, dtype=r); v = np.ascontiguousarray(values, dtype=r)
        out = {"obs": np.zeros((N, A.GO2_NUM_OBS), r), "priv": np.zeros((N, A.GO2_NUM_PRIV_OBS), r), "rewards": np.zeros(N, r), "dones": np.zeros(N, np.uint8),
               "info": np.zeros(A.GO2_EPISODE_INFO_LEN, r)}
        o = A.StepOutputs()
        fp, bp = C.POINTER(A.real), C.POINTER(C.c_uint8)
        o.obs_out, o.priv_out, o.values = out["obs"].ctypes.data_as(fp), out["priv"].ctypes.data_as(fp), v.ctypes.data_as(fp)
        o.rewards_out, o.dones_out, o.episode_info_out, o.gamma = out["rewards"].ctypes.data_as(fp), out["dones"].ctypes.data_as(bp), out["info"].ctypes.data_as(fp), gamma
        _abi.check(self.lib, self.lib.go2sim_step_rollout(self.h, a.ctypes.data, C.byref(o), None), "step_rollout")
        return out

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


# ---- device-memory libraries (the HIP product) -------------------------------------------------------------
class _Proxy:
    """numpy-flavoured access to a torch tensor living in device memory, so the golden/parity tests written for
    HostSim run unchanged on the GPU: `sim.root_states[:] = ndarray`, `np.asarray(sim.obs_buf)`, slicing -> ndarray."""

    def __init__(self, t):
        self.t = t

    def __setitem__(self, idx, val):
        import torch
        self.t[idx] = torch.as_tensor(np.asarray(val), device=self.t.device).to(self.t.dtype)

    def __getitem__(self, idx):
        return self.t[idx].cpu().numpy()

    def __array__(self, dtype=None, copy=None):
        a = self.t.cpu().numpy()
        return a.astype(dtype) if dtype is not None else a

    def any(self):
        return bool(self.t.any().item())

    def sum(self, *a, **k):
        return self.__array__().sum(*a, **k)

    @property
    def shape(self):
        return tuple(self.t.shape)


class DeviceSim:
    """Same interface as HostSim for a device-memory library (go2_rl_gym_amd/libgo2sim_hip.so)."""

    def __init__(self, lib, device="cuda:0", **overrides):
        import torch
        from go2_rl_gym_amd.envs.base.base_task import wrap_buffers
        self.lib, self.abi, self.device = lib, lib.abi, device
        self.real = np.float32
        cfg = self.abi.Cfg()
        lib.go2sim_default_cfg(C.byref(cfg))
        self._keep = []
        for k, v in overrides.items():
            HostSim.set_cfg(self, cfg, k, v)
        if "num_envs_global" not in overrides:
            cfg.num_envs_global = cfg.env_offset + cfg.num_envs
        self.cfg = cfg
        h = C.c_void_p()
        _abi.check(lib, lib.go2sim_create(C.byref(cfg), torch.device(device).index or 0, C.byref(h)), "go2sim_create")
        self._wrap(h, cfg.num_envs, device)

    @classmethod
    def attach(cls, env):
        self = cls.__new__(cls)
        self.lib, self.abi, self.device, self.real, self._keep, self.cfg, self._env = env.lib, env.lib.abi, env.device, np.float32, [], env._c, env
        self._wrap(env.handle, env.num_envs, env.device)
        self.close = lambda: None
        return self

    def _wrap(self, h, N, device):
        import torch
        from go2_rl_gym_amd.envs.base.base_task import wrap_buffers
        lib = self.lib
        self.h, self.N = h, N
        self.t = wrap_buffers(lib, h, self.N, device)
        self.buf = {k: _Proxy(v) for k, v in self.t.items()}
        for k, v in self.buf.items():
            setattr(self, k, v)
        self.torch = torch

    def _st(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if self.h:
            self.torch.cuda.synchronize()
            self.lib.go2sim_destroy(self.h)
            self.h = None

    __del__ = HostSim.__del__

    def reset_all(self):
        _abi.check(self.lib, self.lib.go2sim_reset_all(self.h, self._st()), "reset_all")

    def step_rollout(self, actions, values, gamma):
        t, A, N = self.torch, self.abi, self.N
        a = t.as_tensor(np.ascontiguousarray(actions, dtype=np.float32), device=self.device); v = t.as_tensor(np.ascontiguousarray(values, dtype=np.float32), device=self.device)
        out = {"obs": t.zeros(N, A.GO2_NUM_OBS, device=self.device), "priv": t.zeros(N, A.GO2_NUM_PRIV_OBS, device=self.device), "rewards": t.zeros(N, device=self.device),
               "dones": t.zeros(N, dtype=t.uint8, device=self.device), "info": t.zeros(A.GO2_EPISODE_INFO_LEN, device=self.device)}
        o = A.StepOutputs()
        fp, bp = C.POINTER(C.c_float), C.POINTER(C.c_uint8)
        cast = lambda x, ty: C.cast(C.c_void_p(x.data_ptr()), ty)
        o.obs_out, o.priv_out, o.values = cast(out["obs"], fp), cast(out["priv"], fp), cast(v, fp)
        o.rewards_out, o.dones_out, o.episode_info_out, o.gamma = cast(out["rewards"], fp), cast(out["dones"], bp), cast(out["info"], fp), gamma
        _abi.check(self.lib, self.lib.go2sim_step_rollout(self.h, C.c_void_p(a.data_ptr()), C.byref(o), self._st()), "step_rollout")
        t.cuda.synchronize()
        return {k: x.cpu().numpy() for k, x in out.items()}

    def reset_idx(self, ids):
        ids = self.torch.as_tensor(np.ascontiguousarray(ids, dtype=np.int32), device=self.device)
        _abi.check(self.lib, self.lib.go2sim_reset_idx(self.h, C.c_void_p(ids.data_ptr()), int(ids.numel()), self._st()), "reset_idx")
        self.torch.cuda.synchronize()

    def step(self, actions):
        a = self.torch.as_tensor(np.ascontiguousarray(actions, dtype=np.float32), device=self.device)
        _abi.check(self.lib, self.lib.go2sim_step(self.h, C.c_void_p(a.data_ptr()), self._st()), "step")
        self.torch.cuda.synchronize()

    def simulate(self):
        _abi.check(self.lib, self.lib.go2sim_simulate(self.h, self._st()), "simulate")

    def post_physics(self):
        _abi.check(self.lib, self.lib.go2sim_post_physics(self.h, self._st()), "post_physics")

    def inject(self, u):
        u = self.torch.as_tensor(np.ascontiguousarray(u, dtype=np.float32), device=self.device)
        assert tuple(u.shape) == (self.N, self.abi.GO2_NUM_UNIFORMS)
        _abi.check(self.lib, self.lib.go2sim_inject_uniforms(self.h, C.c_void_p(u.data_ptr()), self._st()), "inject")
        self.torch.cuda.synchronize()

    def peek(self):
        out = self.torch.zeros(self.N, self.abi.GO2_NUM_UNIFORMS, device=self.device)
        _abi.check(self.lib, self.lib.go2sim_peek_uniforms(self.h, C.c_void_p(out.data_ptr()), self._st()), "peek")
        return out.cpu().numpy()


def load_hip():
    from go2_rl_gym_amd import _lib      # imports torch first: one HIP runtime per process (see _lib.py)
    return _lib.load_hip()


def load_hip_kbench():
    """The device library with the per-wave phase timestamps compiled in (build.py:build_hip_kbench) — tools/kbench.py only."""
    from go2_rl_gym_amd import _lib, build  # noqa: F401  (imports torch first)
    lib = _abi.bind(build.build_hip_kbench(), C.c_float)
    assert lib.go2sim_is_device_library() == 1
    return lib


def load_hip_precise():
    """The device library built WITHOUT -ffast-math (go2_rl_gym_amd/build.py:build_hip_precise): a TEST-ONLY second build of the same
    source, used to show which parity differences are fast-math artefacts and which are not.  Never loaded by the product."""
    from go2_rl_gym_amd import _lib, build  # noqa: F401  (imports torch first)
    path = build.build_hip_precise()
    lib = _abi.bind(path, C.c_float)
    assert lib.go2sim_is_device_library() == 1
    return lib


# state that fully determines the next step (copied oracle -> device before a one-step comparison)
STEP_STATE = ["root_states", "dof_state", "last_actions", "last_last_actions", "last_dof_vel", "commands", "commands_resampling_step",
              "commands_xy_accumulation", "episode_length_buf", "motor_strengths", "motor_zero_offsets", "p_gains_multiplier", "d_gains_multiplier",
              "foot_impulse", "last_is_limit_vel", "max_move_distance", "episode_sums", "feet_air_time", "last_contacts", "last_contacts2",
              "friction_coeffs", "restitution_coeffs", "added_base_mass", "added_base_com", "link_mass_ratio", "env_origins", "terrain_levels"]


def heightfield_overrides(num_envs_global, seed=11, mesh_type="heightfield", **terrain_kw):
    """Go2SimCfg overrides for the task=go2 rough terrain, built the way LeggedRobot._fill_cfg does (utils/terrain.py on the host).
    mesh_type 'trimesh' (the registered tasks' default) adds the displaced surface with its vertical faces (hf_cells, hf_walls)."""
    from go2_rl_gym_amd.envs.go2.go2_config import GO2Cfg
    from go2_rl_gym_amd.utils.terrain import Terrain
    tc = GO2Cfg().terrain
    tc.mesh_type = mesh_type
    for k, v in terrain_kw.items():
        setattr(tc, k, v)
    np.random.seed(seed)
    t = Terrain(tc, num_envs_global)
    ov = dict(terrain_mode=1, hf_rows=int(t.tot_rows), hf_cols=int(t.tot_cols), hf_hscale=tc.horizontal_scale, hf_vscale=tc.vertical_scale,
              hf_border=tc.border_size, hf_samples=np.ascontiguousarray(t.heightsamples, np.int16),
              terrain_origins=np.ascontiguousarray(t.env_origins, np.float32), terrain_type_id=np.ascontiguousarray(t.cols2id, np.int32),
              terrain_num_levels=tc.num_rows, terrain_num_types=tc.num_cols, terrain_curriculum=int(tc.curriculum),
              max_init_terrain_level=tc.max_init_terrain_level, measure_heights=int(tc.measure_heights))
    if mesh_type == "trimesh":
        ov.update(hf_cells=np.ascontiguousarray(t.cell_heights, np.int16), hf_walls=1)
    return t, ov


# ---- one-step physics parity: what "within fp32 tolerance" means here --------------------------------------------------------------
# Protocol: every step starts from the fp32 oracle's state, so a comparison is ONE env step (4 substeps of dynamics + contact solve +
# post-physics) from identical inputs.  Two derivations of the same model in fp32 differ by the step's fp32 CONDITIONING — measured, not
# assumed: tools/parity_probe.py (profiles/r2_parity_probe.txt) runs the fp32 oracle against the fp64 oracle on the same inputs and finds
# the same error distribution (p50 / p99 / p99.9 / max) as the HIP kernel against the fp32 oracle, with or without -ffast-math.
#   * plane: ONE bound per tensor, for every env of every step; and a 10x tighter bound for 99 % of them.
#   * rough terrain (facet edges, stair faces: a sphere that changes facet changes its normal): the error distribution is heavy-tailed for
#     the oracle itself, so the gate is relative to it: the kernel's error quantiles stay within a factor of the fp32 oracle's own error
#     against the fp64 oracle, measured in the same run on the same inputs.
# Round 4: one decade tighter where the measured distribution (profiles/r2_parity_probe.txt, re-run as r4_parity_probe.txt) supports it —
# root 1e-3 -> 3e-4, joint state 2e-2 -> 5e-3, torques 1e-2 -> 3e-3, observations 1e-3 -> 3e-4 — with the same three exclusion classes, whose COUNTS the checks now print.
# Round 6: 8 solver sweeps per substep instead of 4 (the body forces' distance to the converged solve: p90 13 % -> 3.8 %).  Twice the sweeps carry the fp32 evaluation-order
# differences between kernel and oracle further: over the 6400 env-steps of test_one_step_parity_vs_oracle the state tensors stay where they were inside their bounds
# (root 8.2e-5, joints 1.7e-3, torques 1.7e-3, obs 8.3e-5) and ONE env-step's reward reaches 3.2e-5 — rewards are sums of ~14 terms of size 1e-3 .. 1e-1 per step, several
# quadratic in joint rates and torques — so the reward bound goes 2e-5 -> 4e-5; nothing else moves.
PLANE_BOUND = {"root_states": 3e-4, "dof_state": 5e-3, "torques": 3e-3, "obs_buf": 3e-4, "privileged_obs_buf": 3e-4, "rew_buf": 4e-5}
# Rough terrain keeps round 3's absolute bound for its well-conditioned env-steps: the tighter plane values are not what its data supports (MI355X, 8000
# env-steps on the trimesh: 99th percentile of root_states 1.5e-4 against a third of 3e-4 — facet and wall switches inside a step)
ROUGH_BOUND = {"root_states": 1e-3, "dof_state": 2e-2, "torques": 1e-2, "obs_buf": 1e-3, "privileged_obs_buf": 1e-3, "rew_buf": 4e-5}


class StepErrors:
    """Accumulates per-env max-abs differences of the named buffers between two simulators over many steps."""

    def __init__(self, keys, bounds=None):
        self.d = {k: [] for k in keys}
        self.bounds = PLANE_BOUND if bounds is None else bounds          # what "ill-conditioned" is measured against (half a bound)

    PENALISED = [4, 5, 8, 9, 12, 13, 16, 17]      # thigh / calf bodies of _reward_collision (legged_robot.py:1277-1279)

    def add(self, a, b, N, ref64=None):
        """a: fp32 oracle, b: the library under test, ref64 (optional): the fp64 oracle stepped from the same state"""
        for k in self.d:
            d = np.abs(np.asarray(getattr(a, k), np.float64) - np.asarray(getattr(b, k), np.float64)).reshape(N, -1).max(1)
            if k == "rew_buf":
                # _reward_collision COUNTS the penalised bodies whose contact force exceeds 0.1 N: a step function of a quantity the two sides agree
                # on to ~1e-2 N only.  Where either side has such a force within 0.05 N of the threshold the reward may differ by one count
                # (scale x dt); those env-steps are excluded from the reward bound and counted instead (check_plane_errors: they must be rare).
                fa = np.linalg.norm(np.asarray(a.contact_forces, np.float64)[:, self.PENALISED], axis=-1)
                fb = np.linalg.norm(np.asarray(b.contact_forces, np.float64)[:, self.PENALISED], axis=-1)
                edge = ((np.abs(fa - 0.1) < 0.05) | (np.abs(fb - 0.1) < 0.05)).any(1)
                self.edge = getattr(self, "edge", 0) + int(edge.sum()); self.rows = getattr(self, "rows", 0) + N
                d = np.where(edge, 0.0, d)
            if ref64 is not None:
                # Conditioning-aware: an env-step at which the fp32 ORACLE itself differs from the fp64 oracle (same inputs) by more than half
                # the bound is an ill-conditioned step (e.g. a robot lying on foot + hip of one leg: two strongly coupled contacts, 4 iterations)
                # and cannot bound a third evaluation; such env-steps are excluded from the fixed bound and counted (check_plane_errors: rare).
                c = np.abs(np.asarray(getattr(a, k), np.float64) - np.asarray(getattr(ref64, k), np.float64)).reshape(N, -1).max(1)
                bound = self.bounds.get(k)
                if bound is not None:
                    ill = c > bound / 2
                    self.ill = getattr(self, "ill", 0) + int(ill.sum())
                    d = np.where(ill, 0.0, d)
            self.d[k].append(d)
        self.steps_seen = getattr(self, "steps_seen", 0) + N

    def all(self, k):
        return np.concatenate(self.d[k])


def ill_conditioned_envs(so, s64):
    """-> boolean [N]: envs whose step is ill-conditioned in fp32 — the fp32 ORACLE's result differs from the fp64 oracle's (same inputs) by more
    than half of PLANE_BOUND in root state, joint state or observations.  Rare (~1e-4 of env-steps under random actions, e.g. a robot lying on
    the foot and the hip of one leg); such an env-step cannot bound a third evaluation and is left out of per-env bounds."""
    bad = np.zeros(np.asarray(so.root_states).shape[0], bool)
    for k in ("root_states", "dof_state", "obs_buf"):
        d = np.abs(np.asarray(getattr(so, k), np.float64) - np.asarray(getattr(s64, k), np.float64))
        bad |= d.reshape(d.shape[0], -1).max(1) > PLANE_BOUND[k] / 2
    return bad


def check_plane_errors(err):
    print("[parity, plane] %d env-steps; excluded: %d tensor-rows of ill-conditioned env-steps (fp32 oracle further than half a bound from the fp64 oracle), %d env-steps with a "
          "_reward_collision force within 0.05 N of its 0.1 N threshold; maxima of the rest: %s" % (getattr(err, "steps_seen", 0), getattr(err, "ill", 0), getattr(err, "edge", 0),
          ", ".join("%s %.1e / %.0e" % (k, float(err.all(k).max()), b) for k, b in PLANE_BOUND.items())))
    for k, bound in PLANE_BOUND.items():
        v = err.all(k)
        assert v.max() < bound, (k, float(v.max()), bound)                                   # every env of every step
        assert np.quantile(v, 0.99) < bound / 10, (k, float(np.quantile(v, 0.99)), bound / 10)
    if getattr(err, "ill", 0):      # ill-conditioned env-steps (see StepErrors.add): at most 1 in 1000 per tensor, counted over all tensors
        assert err.ill <= max(2 * len(PLANE_BOUND), 0.001 * len(PLANE_BOUND) * err.steps_seen), ("ill-conditioned env-steps", err.ill, err.steps_seen)
    if "rew_buf" in PLANE_BOUND and getattr(err, "rows", 0):
        assert err.edge <= 0.01 * err.rows, ("env-steps at the 0.1 N threshold of _reward_collision", err.edge, err.rows)


def check_rough_errors(err, ill_cap=0.005):
    """Rough terrain (err accumulated with ref64): every env-step whose fp32-vs-fp64 ORACLE gap is below half of ROUGH_BOUND — a
    well-conditioned step — is within the plane's absolute bound (99.75 % of them; 99 % within a third of it); the others (a sphere at a facet edge or a
    stair face: the fp64 and fp32 oracles themselves take different facets) stay under the conditioning-relative gate only
    (check_relative_to_conditioning) and their share is capped."""
    print("[parity, rough] %d env-steps; excluded: %d tensor-rows of ill-conditioned env-steps, %d env-steps at the _reward_collision threshold; outside the plane bound: %s"
          % (getattr(err, "steps_seen", 0), getattr(err, "ill", 0), getattr(err, "edge", 0), ", ".join("%s %d" % (k, int((err.all(k) > b).sum())) for k, b in ROUGH_BOUND.items())))
    for k, bound in ROUGH_BOUND.items():
        v = err.all(k)
        # A facet / wall switch is a discontinuity of the model that the fp32-vs-fp64 oracle pair only SAMPLES: a sphere a few ulp from a cell
        # boundary can fall on the other side in a third evaluation although the two oracles agree (measured with the host build of the
        # lanes: 1 env-step in 8000 on the height field; on the MI355X 12 in 8000 on the trimesh, whose vertical faces are the sharper
        # discontinuity).  So: 99.75 % of the well-conditioned env-steps inside the plane's bound (at most 0.25 % + 2 outside, which
        # the conditioning-relative far-tail gate still covers), 99 % inside a third of it (measured on the MI355X: 2.3e-4 for root_states).
        assert (v > bound).sum() <= 2 + 0.0025 * len(v), (k, int((v > bound).sum()), len(v), float(v.max()), bound)
        assert np.quantile(v, 0.99) < bound / 3, (k, float(np.quantile(v, 0.99)), bound / 3)
    assert getattr(err, "ill", 0) <= ill_cap * len(ROUGH_BOUND) * err.steps_seen, ("ill-conditioned env-steps", err.ill, err.steps_seen)
    assert getattr(err, "edge", 0) <= 0.01 * max(getattr(err, "rows", 1), 1)


def check_relative_to_conditioning(err, cond, floors, factor=3.0):
    """err: kernel vs fp32 oracle; cond: fp32 oracle vs fp64 oracle (same inputs).  Median and 99th percentile of the kernel's error within
    `factor` of the oracle's own (plus an absolute floor), and no heavier far tail."""
    for k, floor in floors.items():
        e, c = err.all(k), cond.all(k)
        for q in (0.5, 0.99):
            assert np.quantile(e, q) <= factor * np.quantile(c, q) + floor, (k, q, float(np.quantile(e, q)), float(np.quantile(c, q)))
        far = 1000 * floor
        assert (e > far).mean() <= 2.0 * (c > far).mean() + 2e-3, (k, "tail", float((e > far).mean()), float((c > far).mean()))


# ---- the contact set of a fallen robot (one slot per body group of a leg) ------------------------------------------------
BASE_B, THIGH_B, CALF_B, HIP_B, FOOT_B = 0, [4, 8, 12, 16], [5, 9, 13, 17], [3, 7, 11, 15], [6, 10, 14, 18]


LYING_KW = dict(turn_over=1, turn_over_proportions=np.array([0.0, 0.0, 1.0], np.float32), push_robots=0, seed=3, kp=[0.0] * 12, kd=[0.5] * 12, randomize_action_delay=0)


def lying_robot_batch(sim, rng):
    """Put every env of `sim` (created with LYING_KW: termination off via init_state.turn_over with no env flipped, limp joints) into a random
    LYING pose: trunk 5-9 cm above the plane, roll / pitch within 0.3 rad, every joint anywhere inside its limits, sinking at 0.3 m/s — poses in
    which the trunk box, hips, thighs, calves and feet touch the ground in many combinations at once (about one env in ten has the base AND both
    the thigh and the calf of one leg in contact; up to 14 of the 19 bodies).  The simultaneous contact set the reference reads from PhysX:
    termination on the base force (legged_robot.py:170-173), _reward_collision over 8 thigh / calf bodies (:1277-1279), feet (:1252)."""
    N = sim.N
    lo = np.array([-1.0472, -1.5708, -2.7227] * 2 + [-1.0472, -0.5236, -2.7227] * 2); hi = np.array([1.0472, 3.4907, -0.83776] * 2 + [1.0472, 4.5379, -0.83776] * 2)
    sim.dof_state[:, :, 0] = rng.uniform(lo + 0.06, hi - 0.06, (N, 12)); sim.dof_state[:, :, 1] = 0
    r, p, y = rng.uniform(-0.3, 0.3, N), rng.uniform(-0.3, 0.3, N), rng.uniform(-3, 3, N)
    cr, sr, cp, sp, cy, sy = np.cos(r / 2), np.sin(r / 2), np.cos(p / 2), np.sin(p / 2), np.cos(y / 2), np.sin(y / 2)
    sim.root_states[:, 3:7] = np.stack([sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy], 1)
    sim.root_states[:, 2] = rng.uniform(0.05, 0.09, N); sim.root_states[:, 7:13] = 0; sim.root_states[:, 9] = -0.3
    sim.foot_impulse[...] = 0


def three_body_envs(forces):
    """envs of contact_forces [N,19,3] whose base reports > 1 N (the termination threshold) and, for at least one leg, thigh AND calf > 0.5 N"""
    f = np.linalg.norm(np.asarray(forces, np.float64), axis=2)
    both = (f[:, THIGH_B] > 0.5) & (f[:, CALF_B] > 0.5)
    return np.nonzero((f[:, BASE_B] > 1.0) & both.any(1))[0], f


# ---- the policy-side MFMA kernels (include/go2nn.h) -----------------------------------------------------------------------------------
def load_nn_emu():
    """TEST-ONLY host build of go2_rl_gym_amd/csrc/go2nn_impl.cpp (g++ -DGO2_EMU): same packed operand buffer, same read order, plain loops."""
    from go2_rl_gym_amd import _nn
    out = os.path.join(ROOT, "tests", "emu", "libgo2nn_emu.so")
    src = os.path.join(ROOT, "go2_rl_gym_amd", "csrc", "go2nn_impl.cpp")
    from go2_rl_gym_amd import build as _b
    if _b.stale(out):          # (every file of csrc/ and include/)
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.run(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-DGO2_EMU", "-w", "-o", out, src], check=True)
    lib = _nn.bind(out)
    assert lib.go2nn_is_device_library() == 0
    return lib
