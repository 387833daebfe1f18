"""Scope row f1: the terrain generator (go2_rl_gym_amd/utils/terrain.py) and the height-field contact / height-scan path.

terrain.npz was captured by running the REFERENCE's Terrain class (legged_gym/utils/terrain.py) — its layout, curriculum
orchestration, per-column kinds, origins — over this repo's sub-terrain generators (isaacgym.terrain_utils is a
closed third-party dependency that is absent, so the generators themselves are pinned by their documented shapes below).
CPU only.
"""
import hashlib
import os
import numpy as np
import pytest

from helpers import ROOT, STEP_STATE, HostSim, heightfield_overrides, load_emu, load_oracle
from go2_rl_gym_amd.utils import terrain as T

G = os.path.join(ROOT, "tests", "golden")


def test_terrain_class_matches_reference_orchestration():
    g = dict(np.load(os.path.join(G, "terrain.npz")))
    t, ov = heightfield_overrides(64, seed=int(g["seed"]))
    hf = np.ascontiguousarray(t.height_field_raw)
    assert hf.dtype == np.int16 and tuple(hf.shape) == tuple(g["shape"]) == (1345, 2195)
    assert (t.tot_rows, t.tot_cols) == (int(g["tot_rows"]), int(g["tot_cols"]))
    np.testing.assert_array_equal(np.array(t.cols2id), g["cols2id"])
    np.testing.assert_allclose(t.env_origins, g["env_origins"], atol=1e-9)
    np.testing.assert_array_equal(hf.astype(np.int64).sum(0), g["col_sums"])
    np.testing.assert_array_equal(hf.astype(np.int64).sum(1), g["row_sums"])
    np.testing.assert_array_equal(hf[250:330, 250:330], g["tile_wave"])
    np.testing.assert_array_equal(hf[250 + 9 * 85:330 + 9 * 85, 250 + 8 * 85:330 + 8 * 85], g["tile_stairs"])
    np.testing.assert_array_equal(hf[250 + 5 * 85:330 + 5 * 85, 250 + 14 * 85:330 + 14 * 85], g["tile_obstacles"])
    assert hashlib.sha256(hf.tobytes()).digest() == g["sha256"].tobytes()
    # border stays flat, every tile's origin height is the max of its 2 m centre patch (terrain.py:160-174)
    assert not hf[:250].any() and not hf[:, :250].any() and not hf[-250:].any() and not hf[:, -250:].any()


def _sub(width=80, length=80):
    return T.SubTerrain("t", width=width, length=length, vertical_scale=0.005, horizontal_scale=0.1)


def test_generators_shapes():
    s = T.pyramid_sloped_terrain(_sub(), slope=0.3, platform_size=3.0)
    h = s.height_field_raw
    assert h.dtype == np.int16 and h.shape == (80, 80)
    c = h[40, 40] * 0.005
    assert abs(c - h.max() * 0.005) < 1e-9 and h[0, 0] == 0                       # a pyramid: flat clipped top, zero at the corners
    np.testing.assert_array_equal(h[1:, 1:], h[1:, 1:][::-1, ::-1]); assert np.abs(h.astype(int) - h.T).max() <= 1     # centred on cell 40 of 0..79
    assert abs(c - 0.3 * 4.0 * (1 - 1.5 / 4.0) ** 2) < 0.01                      # slope x half-width x (xx yy) at the 3 m platform's corner
    s = T.pyramid_sloped_terrain(_sub(), slope=-0.3, platform_size=3.0)
    assert s.height_field_raw.min() < 0 and s.height_field_raw.max() == 0
    # stairs: constant treads of step_width, rises of exactly step_height, up then flat platform
    s = T.pyramid_stairs_terrain(_sub(), step_width=0.31, step_height=0.1, platform_size=3.0)
    h = s.height_field_raw
    row = h[:40, 40].astype(int)
    d = np.diff(row)
    assert set(np.unique(d)) <= {0, 20} and (d == 20).sum() >= 5                   # 0.1 m / 0.005
    assert np.all(np.diff(np.flatnonzero(d == 20)) == 3)                         # int(0.31 / 0.1) cells per tread
    np.testing.assert_array_equal(h, h[::-1, ::-1]); np.testing.assert_array_equal(h, h.T)
    s = T.pyramid_stairs_terrain(_sub(), step_width=0.31, step_height=-0.1, platform_size=3.0)
    assert s.height_field_raw.min() < 0 and s.height_field_raw.max() == 0
    # uniform noise: heights on the step grid inside [min, max]
    np.random.seed(0)
    s = T.random_uniform_terrain(_sub(), min_height=-0.05, max_height=0.05, step=0.005, downsampled_scale=0.2)
    h = s.height_field_raw
    assert h.min() >= -10 and h.max() <= 10 and h.std() > 2
    # discrete obstacles: flat platform in the middle, only the listed heights elsewhere
    np.random.seed(0)
    s = T.discrete_obstacles_terrain(_sub(), 0.2, 1.0, 2.0, 20, platform_size=3.0)
    h = s.height_field_raw
    assert not h[26:54, 26:54].any() and set(np.unique(h)) <= {-40, -20, 0, 20, 40} and (h != 0).sum() > 100
    # wave: zero mean-ish, amplitude / 2 peak
    s = T.wave_terrain(_sub(), num_waves=2, amplitude=0.4)
    assert abs(int(s.height_field_raw.max()) - 80) <= 1 and abs(int(s.height_field_raw.min()) + 80) <= 1
    # stepping stones: pits at -depth between stones at 0, platform at 0
    np.random.seed(0)
    s = T.stepping_stones_terrain(_sub(), stone_size=1.0, stone_distance=0.2, max_height=0.0, platform_size=3.0, depth=-10)
    h = s.height_field_raw
    assert h.min() == -2000 and not h[30:50, 30:50].any() and 0.4 < (h >= -1).mean() < 0.95        # stones sit at {-1, 0} raw units
    # gap and pit (legged_gym/utils/terrain.py:176-201)
    s = _sub(); T.gap_terrain(s, gap_size=0.5, platform_size=3.0)
    h = s.height_field_raw
    assert h.min() == -1000 and not h[30:50, 30:50].any() and not h[:10].any()
    s = _sub(); T.pit_terrain(s, depth=0.5, platform_size=4.0)
    assert s.height_field_raw[40, 40] == -100 and s.height_field_raw[0, 0] == 0


def test_selected_and_random_modes():
    from go2_rl_gym_amd.envs.go2.go2_config import GO2Cfg
    tc = GO2Cfg().terrain
    tc.mesh_type, tc.curriculum, tc.num_rows, tc.num_cols = "heightfield", False, 3, 4
    np.random.seed(3)
    t = T.Terrain(tc, 8)
    assert t.height_field_raw.shape == (3 * 80 + 2 * 5 + 500, 4 * 80 + 3 * 5 + 500)      # 8 m tiles, 0.5 m spacing, 25 m border and len(t.cols2id) == 0 and t.env_origins.shape == (3, 4, 3)
    tc.selected, tc.terrain_kwargs = True, {"type": "pyramid_stairs_terrain", "step_width": 0.31, "step_height": 0.1, "platform_size": 3.0}
    t2 = T.Terrain(tc, 8)
    np.testing.assert_array_equal(t2.height_field_raw[250:330, 250:330], t2.height_field_raw[335:415, 335:415])     # every tile the same stairs
    assert t2.height_field_raw.max() > 100
    tc.mesh_type = "plane"
    assert not hasattr(T.Terrain(tc, 8), "height_field_raw")       # plane / none: nothing built (terrain.py:15-16)


# ------------------------------------------------------------------ contact + height scan on the height field
N = 40


def _pair(**kw):
    t, ov = heightfield_overrides(N)
    kw.update(ov)
    return t, HostSim(load_oracle(), num_envs=N, **kw), HostSim(load_emu(), num_envs=N, **kw)


def test_heightfield_one_step_parity_lanes_vs_oracle():
    """Same protocol as test_lane_emulation, on the rough terrain: slopes, stairs, obstacles under the feet,
    terrain curriculum and the 187-point height scan active."""
    from helpers import StepErrors, check_relative_to_conditioning
    t, so, se = _pair()
    s64 = HostSim(load_oracle(f64=True), num_envs=N, **heightfield_overrides(N)[1])
    for k in ("env_origins", "terrain_levels", "terrain_types"):
        np.testing.assert_array_equal(np.asarray(getattr(so, k)), np.asarray(getattr(se, k)), err_msg=k)
    assert len(np.unique(np.asarray(so.terrain_types))) == 20
    so.reset_all(); se.reset_all(); s64.reset_all()
    rng = np.random.default_rng(1)
    contact_seen = tilted = 0
    floors = {"root_states": 2e-5, "dof_state": 2e-4, "torques": 2e-4, "obs_buf": 2e-5, "privileged_obs_buf": 2e-5, "rew_buf": 2e-7}
    err, cond = StepErrors(floors), StepErrors(floors)
    for it in range(90):
        a = rng.normal(0, 0.6, (N, 12)).astype(np.float32)
        for k in STEP_STATE:
            getattr(se, k)[...] = getattr(so, k); getattr(s64, k)[...] = getattr(so, k)
        so.step(a); se.step(a); s64.step(a.astype(np.float64))
        f = np.asarray(so.contact_forces)[:, [6, 10, 14, 18]]
        contact_seen += int((f[..., 2] > 1).sum())
        tilted += int(((f[..., 2] > 1) & (np.abs(f[..., :2]).max(-1) > 0.2 * f[..., 2])).sum())
        err.add(so, se, N); cond.add(so, s64, N)
        assert (np.abs(np.asarray(so.measured_heights) - np.asarray(se.measured_heights)) > 1e-6).sum() <= 2      # scan taken at poses 1e-5 apart
        np.testing.assert_array_equal(np.asarray(so.reset_buf), np.asarray(se.reset_buf))
        np.testing.assert_array_equal(np.asarray(so.terrain_levels), np.asarray(se.terrain_levels))
    # facet edges make a step ill-conditioned in fp32 for any evaluation order: the lane programs' error stays within 3x of the fp32
    # oracle's own error against the fp64 oracle on the same inputs (helpers.check_relative_to_conditioning)
    check_relative_to_conditioning(err, cond, floors)
    assert contact_seen > 1000 and tilted > 50        # feet did load non-horizontal facets
    assert np.abs(np.asarray(so.measured_heights)).max() > 0.05
    s64.close()


@pytest.mark.parametrize("which", ["oracle", "lane_emulation"])
def test_robots_stand_on_every_terrain_kind(which):
    """Zero actions (default pose) for 2 s on every column of the curriculum map: nobody falls through or is launched."""
    lib = load_oracle() if which == "oracle" else load_emu()
    t, ov = heightfield_overrides(N)
    s = HostSim(lib, num_envs=N, push_robots=0, **ov)
    s.reset_all()
    a = np.zeros((N, 12), np.float32)
    resets = 0
    for _ in range(100):
        s.step(a)
        resets += int(np.asarray(s.reset_buf).sum())
    root = np.asarray(s.root_states)
    hgt = root[:, 2] - np.asarray(s.measured_heights)[:, 93]       # centre sample of the 17 x 11 scan
    assert np.isfinite(root).all()
    assert (hgt > 0.12).mean() > 0.9 and (hgt < 0.6).all(), np.sort(hgt)
    assert np.abs(root[:, 7:10]).max() < 3.0
    assert resets <= N // 4


def test_rough_task_through_the_host_layer():
    """task=go2 (mesh_type trimesh -> height field) through LeggedRobot/task_registry, on the host build of the lane programs
    (the GPU run of the same path is tests/test_gpu_parity.py::test_train_rough_terrain_on_gpu)."""
    import torch
    from go2_rl_gym_amd.envs import task_registry
    from go2_rl_gym_amd.utils import get_args
    args = get_args(["--task", "go2", "--num_envs", "40", "--headless", "--sim_device", "cpu", "--rl_device", "cpu"])
    env, cfg = task_registry.make_env("go2", args, lib=load_emu())
    assert cfg.terrain.mesh_type == "trimesh" and env.custom_origins
    assert tuple(env.height_samples.shape) == (1345, 2195) and env.height_samples.dtype == torch.int16
    np.testing.assert_array_equal(env.terrain_ids.numpy(), np.array(env.terrain.cols2id)[env.terrain_types.numpy()])
    np.testing.assert_allclose(env.env_origins.numpy(), env.terrain_origins.numpy()[env.terrain_levels.numpy(), env.terrain_types.numpy()])
    a = torch.zeros(40, 12)
    env.reset()                                   # (extras['episode'] is written by reset_idx only, legged_robot.py:229-237)
    for _ in range(5):
        obs, priv, rew, done, extras = env.step(a)
    ep = extras["episode"]
    assert abs(float(ep["terrain_level_all"]) - float(env.terrain_levels.float().mean())) < 1e-6
    for name, cols in env.terrain.name2cols.items():
        m = torch.isin(env.terrain_types, torch.tensor(sorted(cols)))
        assert abs(float(ep["terrain_level_" + name]) - float(env.terrain_levels[m].float().mean())) < 1e-5
    assert priv.shape == (40, 263) and float(env.measured_heights.abs().max()) > 0.02
    env.close()


def test_heightfield_to_trimesh():
    """convert_heightfield_to_trimesh (the mesh the reference hands PhysX for mesh_type 'trimesh', terrain.py:45-49): counts, the cell
    diagonal the HIP contact query also uses, and the slope-threshold correction — a 1-cell ramp steeper than the threshold becomes a
    vertical wall standing at the HIGH vertex, the low ground extending up to it; gentle slopes are untouched."""
    from go2_rl_gym_amd.utils.terrain import convert_heightfield_to_trimesh
    hs, vs = 0.1, 0.005
    hf = np.zeros((6, 8), np.int16)
    hf[3:, :] = 40                                    # a 0.2 m step across x between rows 2 and 3: slope 2.0 over one cell
    hf[:, 6:] += 4                                    # a 0.02 m step across y between cols 5 and 6: slope 0.2
    v0, t0 = convert_heightfield_to_trimesh(hf, hs, vs, None)
    assert v0.shape == (48, 3) and v0.dtype == np.float32 and t0.shape == (2 * 5 * 7, 3) and t0.dtype == np.uint32
    np.testing.assert_allclose(v0[:, 2].reshape(6, 8), hf * vs, atol=1e-7)
    np.testing.assert_allclose(v0[:, 0].reshape(6, 8), np.arange(6)[:, None] * hs * np.ones((1, 8)), atol=1e-6)
    np.testing.assert_array_equal(t0[0], [0, 9, 1]); np.testing.assert_array_equal(t0[1], [0, 8, 9])     # cell (0,0): split along (0,0)-(1,1)
    assert t0.max() == 47
    v1, t1 = convert_heightfield_to_trimesh(hf, hs, vs, 0.75)
    np.testing.assert_array_equal(t1, t0)
    x1, y1 = v1[:, 0].reshape(6, 8), v1[:, 1].reshape(6, 8)
    np.testing.assert_allclose(x1[2], 0.3, atol=1e-6)            # the low row next to the step moved under the edge: wall at x = 0.3
    np.testing.assert_allclose(x1[[0, 1, 3, 4, 5]], v0[:, 0].reshape(6, 8)[[0, 1, 3, 4, 5]], atol=1e-6)
    y0 = v0[:, 1].reshape(6, 8)
    np.testing.assert_allclose(y1[[0, 1, 3, 4, 5]], y0[[0, 1, 3, 4, 5]], atol=1e-6)                      # the gentle step is left alone
    # the diagonal rule also fires along a straight wall (h[i+1,j+1] - h[i,j] > thr) and, having no y move to defer to, slides the wall-foot
    # vertices one cell along the wall; their heights are those of the foot line, so the surface is the same
    np.testing.assert_allclose(y1[2, :7], y0[2, :7] + hs, atol=1e-6); np.testing.assert_allclose(y1[2, 7], y0[2, 7], atol=1e-6)
    np.testing.assert_array_equal(v1[:, 2], v0[:, 2])
    # a one-sample pit: pulled both ways, stays where it is
    pit = np.full((5, 5), 40, np.int16); pit[2, 2] = 0
    vp, _ = convert_heightfield_to_trimesh(pit, hs, vs, 0.75)
    np.testing.assert_allclose(vp[12, :2], [0.2, 0.2], atol=1e-6)


def test_terrain_class_exposes_the_trimesh_lazily():
    from go2_rl_gym_amd.envs.go2.go2_config import GO2Cfg
    from go2_rl_gym_amd.utils.terrain import Terrain
    tc = GO2Cfg().terrain
    tc.mesh_type, tc.num_rows, tc.num_cols, tc.border_size = "trimesh", 2, 2, 1.0
    np.random.seed(3)
    t = Terrain(tc, 4)
    assert t._trimesh is None
    assert t.vertices.shape == (t.tot_rows * t.tot_cols, 3) and t.triangles.shape == (2 * (t.tot_rows - 1) * (t.tot_cols - 1), 3)
    assert abs(float(t.vertices[:, 2].max()) - float(t.height_field_raw.max()) * tc.vertical_scale) < 1e-6


def _raycast_down(vertices, triangles, pts):
    """Highest intersection of the vertical line through each (x, y) with the triangle mesh (brute force, small meshes only)."""
    v = vertices.astype(np.float64)
    t0, t1, t2 = v[triangles[:, 0]], v[triangles[:, 1]], v[triangles[:, 2]]
    out = np.full(len(pts), -np.inf)
    for k, (x, y) in enumerate(pts):
        det = (t1[:, 1] - t2[:, 1]) * (t0[:, 0] - t2[:, 0]) + (t2[:, 0] - t1[:, 0]) * (t0[:, 1] - t2[:, 1])
        ok = np.abs(det) > 1e-12
        d = np.where(ok, det, 1.0)
        l0 = ((t1[:, 1] - t2[:, 1]) * (x - t2[:, 0]) + (t2[:, 0] - t1[:, 0]) * (y - t2[:, 1])) / d
        l1 = ((t2[:, 1] - t0[:, 1]) * (x - t2[:, 0]) + (t0[:, 0] - t2[:, 0]) * (y - t2[:, 1])) / d
        l2 = 1 - l0 - l1
        ins = ok & (l0 >= -1e-9) & (l1 >= -1e-9) & (l2 >= -1e-9)
        if ins.any():
            out[k] = (l0 * t0[:, 2] + l1 * t1[:, 2] + l2 * t2[:, 2])[ins].max()
    return out


@pytest.mark.parametrize("kind", ["stairs", "obstacles", "slope"])
def test_displaced_cell_heights_reproduce_the_trimesh_surface(kind):
    """utils/terrain.py:displaced_cell_heights (what the contact query reads for mesh_type 'trimesh') against a vertical ray cast onto the
    mesh convert_heightfield_to_trimesh builds WITH the slope_treshold displacement (legged_gym/utils/terrain.py:46-49): same surface
    height at random points; risers are vertical faces (neighbouring cells disagree on their common edge), not 1-cell ramps."""
    from go2_rl_gym_amd.utils.terrain import (SubTerrain, convert_heightfield_to_trimesh, discrete_obstacles_terrain, displaced_cell_heights,
                                              pyramid_sloped_terrain, pyramid_stairs_terrain)
    np.random.seed(3)
    hs, vs, thr = 0.1, 0.005, 0.75
    t = SubTerrain("t", width=60, length=60, vertical_scale=vs, horizontal_scale=hs)
    if kind == "stairs":
        pyramid_stairs_terrain(t, step_width=0.31, step_height=0.15, platform_size=2.0)
    elif kind == "obstacles":
        discrete_obstacles_terrain(t, 0.15, 1.0, 2.0, 20, platform_size=2.0)
    else:
        pyramid_sloped_terrain(t, slope=0.3, platform_size=2.0)
    hf = t.height_field_raw
    cells = displaced_cell_heights(hf, hs, vs, thr)
    assert cells.shape == (59, 59, 4) and cells.dtype == np.int16
    verts, tris = convert_heightfield_to_trimesh(hf, hs, vs, thr)
    rng = np.random.default_rng(0)
    pts = rng.uniform(0.35, 5.55, size=(400, 2))
    pts = pts[(np.abs(pts / hs - np.rint(pts / hs)) > 0.02).all(1)]          # not on a grid line (where a wall makes the height two-valued)
    want = _raycast_down(verts, tris, pts)
    i, j = np.floor(pts[:, 0] / hs).astype(int), np.floor(pts[:, 1] / hs).astype(int)
    u, v = pts[:, 0] / hs - i, pts[:, 1] / hs - j
    c = cells[i, j].astype(np.float64) * vs
    got = np.where(u >= v, c[:, 0] + u * (c[:, 1] - c[:, 0]) + v * (c[:, 3] - c[:, 1]), c[:, 0] + u * (c[:, 3] - c[:, 2]) + v * (c[:, 2] - c[:, 0]))
    d = np.abs(got - want)
    if kind == "slope":
        assert d.max() < 1e-6                                              # (the mesh vertices are float32)
    elif kind == "stairs":
        assert d.max() <= vs * 0.51                                        # planes + vertical faces: exact up to the int16 rounding
    else:
        # box corners: the displaced mesh has small skew triangles there that a cell's two facets only interpolate at the corners
        assert (d <= vs * 0.51).mean() >= 0.97 and d.max() < 0.05, (float((d <= vs * 0.51).mean()), float(d.max()))
    plain = displaced_cell_heights(hf, hs, vs, None)
    if kind == "slope":
        np.testing.assert_array_equal(cells, plain)            # below the threshold nothing moves: continuous cells, no walls
    else:
        # walls: some neighbouring cells disagree on their common edge by a whole step; the un-displaced cells never do
        jump = np.abs(cells[:-1, :, 1].astype(int) - cells[1:, :, 0].astype(int))
        assert jump.max() >= int(0.15 / vs) - 1 and np.abs(plain[:-1, :, 1].astype(int) - plain[1:, :, 0].astype(int)).max() == 0
        # and the 1-cell ramps are gone: inside every displaced cell the surface is (nearly) flat where the raw cell was a ramp
        ramp = (plain.max(-1).astype(int) - plain.min(-1).astype(int)) >= int(0.15 / vs) - 1
        assert ramp.any() and ((cells.max(-1).astype(int) - cells.min(-1).astype(int))[ramp] <= 1).mean() > 0.75      # (the rest: box-corner cells)
