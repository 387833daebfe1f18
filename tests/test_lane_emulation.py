"""The HIP lane programs (go2_rl_gym_amd/csrc/go2_lane.h, go2_post.h), compiled for the host, against the oracle.

CPU only.  The two sides derive the same physical model differently (oracle: Featherstone ABA in link coordinates +
dense CRBA/Cholesky contact problem; lanes: single-frame spatial algebra, per-leg 3x3 block elimination onto the base,
operational-space rows from Ainv / N / Phi), so agreement to fp32 round-off checks both.
"""
import numpy as np
import pytest

from helpers import STEP_STATE, HostSim, load_emu, load_oracle

N = 48


def _pair(**kw):
    return HostSim(load_oracle(), num_envs=N, **kw), HostSim(load_emu(), num_envs=N, **kw)


def test_creation_and_reset_identical():
    so, se = _pair()
    for k in ("friction_coeffs", "restitution_coeffs", "added_base_mass", "added_base_com", "link_mass_ratio", "env_origins"):
        np.testing.assert_array_equal(np.asarray(getattr(so, k)), np.asarray(getattr(se, k)), err_msg=k)
    np.testing.assert_array_equal(so.peek(), se.peek())      # same Philox stream
    so.reset_all(); se.reset_all()
    for k in ("root_states", "dof_state", "commands", "motor_strengths", "motor_zero_offsets", "p_gains_multiplier", "d_gains_multiplier"):
        np.testing.assert_allclose(np.asarray(getattr(so, k)), np.asarray(getattr(se, k)), atol=1e-6, err_msg=k)
    # (go2sim_reset_all is reset_idx(all) only — base_task.py:82-86 follows it with a zero-action step, which is what produces the
    #  observations reset() returns; the buffers in between are not part of the contract)
    a0 = np.zeros((N, 12), np.float32)
    so.step(a0); se.step(a0)
    np.testing.assert_allclose(np.asarray(so.obs_buf), np.asarray(se.obs_buf), atol=2e-5, err_msg="obs_buf after reset + zero-action step")


def test_one_step_parity_through_landing_and_stance():
    """120 steps of random actions; before every step the emulation is synced to the oracle's state, so each comparison
    is ONE step (4 substeps incl. contact solve + post-physics) from identical inputs."""
    from helpers import PLANE_BOUND, StepErrors, check_plane_errors, ill_conditioned_envs
    so, se = _pair()
    s64 = HostSim(load_oracle(f64=True), num_envs=N)      # the fp64 oracle from the same state: tells an ill-conditioned step from an error
    so.reset_all(); se.reset_all(); s64.reset_all()
    rng = np.random.default_rng(0)
    contact_seen = 0
    err = StepErrors(PLANE_BOUND)
    for it in range(120):
        a = rng.normal(0, 1, (N, 12)).astype(np.float32)
        for k in STEP_STATE:
            getattr(se, k)[...] = getattr(so, k); getattr(s64, k)[...] = getattr(so, k)
        so.step(a); se.step(a); s64.step(a.astype(np.float64))
        contact_seen += int((so.contact_forces[:, [6, 10, 14, 18], 2] > 1).sum())
        err.add(so, se, N, ref64=s64)
        ok = ~ill_conditioned_envs(so, s64)
        # forces are impulse / 0.005 s: fp32 noise is amplified 200x, compare relative to the force scale
        fo, fe = np.asarray(so.contact_forces, np.float64)[ok], np.asarray(se.contact_forces, np.float64)[ok]
        de = np.abs(fo - fe).reshape(int(ok.sum()), -1).max(1)
        assert np.median(de) < 5e-3 and de.max() < 2e-3 * max(1.0, np.abs(fo).max()) + 0.5, (it, np.sort(de)[-3:])
        np.testing.assert_array_equal(np.asarray(so.reset_buf), np.asarray(se.reset_buf))
    check_plane_errors(err)             # ONE bound per tensor for every env of every step (helpers.PLANE_BOUND) + 10x tighter for 99 %
    assert contact_seen > 1000          # the robots did land and stand


def test_free_trajectory_stays_statistically_close():
    """Without re-syncing, fp32 round-off grows chaotically once contacts switch; the ensembles must still agree."""
    so, se = _pair(push_robots=0)
    so.reset_all(); se.reset_all()
    a = np.zeros((N, 12), np.float32)
    for _ in range(100):
        so.step(a); se.step(a)
    zo, ze = np.asarray(so.root_states)[:, 2], np.asarray(se.root_states)[:, 2]
    assert abs(zo.mean() - ze.mean()) < 0.01 and 0.15 < ze.mean() < 0.40
    assert abs(np.asarray(so.rew_buf).mean() - np.asarray(se.rew_buf).mean()) < 5e-3


@pytest.mark.parametrize("which", ["oracle", "lane_emulation"])
def test_fine_grained_calls_equal_the_fused_step(which):
    """The Isaac-Gym-shaped path — write actions, go2sim_simulate (the 4-substep torque/physics loop), then go2sim_post_physics — and the
    fused go2sim_step give the same result from the same state; set_*_state_indexed accept the Python-side writes (the API tensors ARE
    the state here, so a row written by the caller is simply what the next call reads)."""
    import ctypes as C
    lib = load_oracle() if which == "oracle" else load_emu()
    a_, b_ = HostSim(lib, num_envs=24, seed=4), HostSim(lib, num_envs=24, seed=4)
    a_.reset_all(); b_.reset_all()
    rng = np.random.default_rng(3)
    for it in range(12):
        act = rng.normal(0, 1, (24, 12)).astype(np.float32)
        if it == 5:       # teleport two robots through the API tensors, as reset code on top of Isaac Gym would
            for s in (a_, b_):
                s.root_states[3, :3] += np.array([0.5, -0.25, 0.1], np.float32); s.dof_state[7, :, 0] *= 0.9
                ids = np.array([3, 7], np.int32)
                assert lib.go2sim_set_root_state_indexed(s.h, ids.ctypes.data, 2, None) == 0
                assert lib.go2sim_set_dof_state_indexed(s.h, ids.ctypes.data, 2, None) == 0
        a_.step(act)
        b_.actions[:] = act; b_.simulate(); b_.post_physics()
        for k in ("root_states", "dof_state", "obs_buf", "privileged_obs_buf", "rew_buf", "reset_buf", "torques", "contact_forces", "commands"):
            np.testing.assert_array_equal(np.asarray(getattr(a_, k)), np.asarray(getattr(b_, k)), err_msg="%s at %d" % (k, it))
    assert lib.go2sim_get_common_step_counter(a_.h) == lib.go2sim_get_common_step_counter(b_.h) == 12
    assert lib.go2sim_set_root_state_indexed(None, None, 0, None) != 0
    a_.close(); b_.close()


def test_curriculum_state_follows_the_counter():
    """reward-curriculum scales, command ranges and zero-command probability as pure functions of common_step_counter // 24
    (get_current_scale :154-168, update_command_ranges, zero_command_curriculum) — values at a few iterations of the go2 config."""
    import ctypes as C
    lib = load_emu()
    s = HostSim(lib, num_envs=4)
    rcs, cr, zp = (C.c_float * lib.abi.GO2_NUM_REWARDS)(), (C.c_float * 8)(), C.c_float()
    names = lib.abi.reward_names

    def state(it):
        lib.go2sim_set_common_step_counter(s.h, it * 24)
        assert lib.go2sim_get_curriculum_state(s.h, rcs, cr, C.byref(zp)) == 0
        return {n: rcs[i] for i, n in enumerate(names)}, [[cr[2 * r], cr[2 * r + 1]] for r in range(4)], zp.value

    r0, c0, z0 = state(0)
    r1, c1, z1 = state(750)
    r2, c2, z2 = state(60000)
    assert abs(z0 - 0.0) < 1e-7 and abs(z1 - 0.05) < 1e-6 and abs(z2 - 0.1) < 1e-7          # zero_command_curriculum 0 -> 0.1 over 1500 iterations
    np.testing.assert_allclose(c0[0], [-0.5, 0.5]); np.testing.assert_allclose(c2[0], [-2.0, 2.0])   # command_range_curriculum at iter 20000 / 50000
    np.testing.assert_allclose(c2[2], [-2.0, 2.0], atol=1e-6)
    cur = [n for n in names if r0[n] != r2[n]]
    assert cur, "the go2 config has reward curricula"                                        # (go2_config.py:161-167)
    for n in names:
        if n not in cur:
            assert r0[n] == r1[n] == r2[n] == 1.0
    s.close()


def test_soak_under_training_like_action_noise():
    """400 steps x 48 envs of N(0, 2) actions (twice the policy's initial exploration noise) through the lane programs: every buffer finite,
    base spin far from the max_angular_velocity safety clamp (DESIGN.md 4 'validity range'), robots keep getting reset and re-spawned."""
    s = HostSim(load_emu(), num_envs=N, seed=11)
    s.reset_all()
    rng = np.random.default_rng(0)
    resets, wmax = 0, 0.0
    for _ in range(400):
        s.step(rng.normal(0, 2.0, (N, 12)).astype(np.float32))
        resets += int(np.asarray(s.reset_buf).sum())
        wmax = max(wmax, float(np.linalg.norm(np.asarray(s.root_states)[:, 10:13], axis=1).max()))
        for k in ("root_states", "dof_state", "obs_buf", "privileged_obs_buf", "rew_buf", "torques", "contact_forces"):
            assert np.isfinite(np.asarray(getattr(s, k))).all(), k
    assert wmax < 60.0 and resets > 10, (wmax, resets)
    s.close()


@pytest.mark.parametrize("which", ["oracle", "lane_emulation"])
def test_step_rollout_equals_step_plus_bookkeeping(which):
    """go2sim_step_rollout: redirected observation rows, the transition store with the time-out bootstrap (ppo.py:107-108) and the extras copy
    equal go2sim_step followed by those operations done by hand (the HIP run: tests/test_gpu_parity.py)."""
    import ctypes as C
    lib = load_oracle() if which == "oracle" else load_emu()
    check_step_rollout(lib, HostSim, 24)


def check_step_rollout(lib, sim, n):
    import ctypes as C
    a_, b_ = sim(lib, num_envs=n, seed=8), sim(lib, num_envs=n, seed=8)
    a_.reset_all(); b_.reset_all()
    el = np.arange(n) % 7 + 1244                            # some envs time out within the run
    a_.episode_length_buf[:] = el; b_.episode_length_buf[:] = el
    rng = np.random.default_rng(1)
    A = lib.abi
    seen_to = 0
    for it in range(10):
        act = rng.normal(0, 1, (n, 12)).astype(np.float32)
        vals = rng.normal(0, 1, n).astype(np.float32)
        a_.step(act)
        out = b_.step_rollout(act, vals, gamma=0.99)
        to = np.asarray(a_.time_out_buf).astype(bool); seen_to += int(to.sum())
        np.testing.assert_array_equal(out["obs"], np.asarray(a_.obs_buf)); np.testing.assert_array_equal(out["priv"], np.asarray(a_.privileged_obs_buf))
        np.testing.assert_allclose(out["rewards"], np.asarray(a_.rew_buf) + np.float32(0.99) * vals * to, atol=1e-7)
        np.testing.assert_array_equal(out["dones"], np.asarray(a_.reset_buf))
        np.testing.assert_allclose(out["info"], np.asarray(a_.episode_info), rtol=2e-6, atol=1e-9)      # (sums of float atomics on the device: order-dependent last bits)
        for k in ("root_states", "dof_state", "rew_buf", "commands"):
            np.testing.assert_array_equal(np.asarray(getattr(a_, k)), np.asarray(getattr(b_, k)), err_msg=k)
    assert seen_to > 0
    a_.close(); b_.close()


def test_reset_idx_subsets_against_the_oracle():
    """go2sim_reset_idx over random subsets (duplicates, out-of-range ids, the empty list, every env) on ragged batch sizes: the lane programs'
    masked reset pass equals the oracle's per-env reset, and nothing outside the listed envs moves."""
    rng = np.random.default_rng(4)
    for n in (1, 5, 16, 37):
        so, se = HostSim(load_oracle(), num_envs=n, seed=3), HostSim(load_emu(), num_envs=n, seed=3)
        so.reset_all(); se.reset_all()
        a = rng.normal(0, 1, (n, 12)).astype(np.float32)
        for _ in range(3):
            so.step(a); se.step(a)
        for k in STEP_STATE:
            getattr(se, k)[...] = getattr(so, k)
        for trial in range(4):
            ids = [np.array([], np.int32), np.arange(n, dtype=np.int32), rng.integers(0, n, max(n // 2, 1)).astype(np.int32),
                   np.concatenate([rng.integers(0, n, 2), [n + 3, -1]]).astype(np.int32)][trial]
            before = {k: np.asarray(getattr(se, k)).copy() for k in ("root_states", "dof_state", "obs_buf", "episode_length_buf", "commands")}
            so.reset_idx(ids); se.reset_idx(ids)
            hit = np.zeros(n, bool); hit[ids[(ids >= 0) & (ids < n)]] = True
            for k in ("root_states", "dof_state", "commands", "motor_strengths", "p_gains_multiplier", "commands_resampling_step"):
                np.testing.assert_allclose(np.asarray(getattr(so, k)), np.asarray(getattr(se, k)), atol=2e-6, err_msg="%s n=%d trial=%d" % (k, n, trial))
            np.testing.assert_array_equal(np.asarray(so.episode_length_buf), np.asarray(se.episode_length_buf))
            np.testing.assert_array_equal(np.asarray(se.obs_buf), before["obs_buf"])
            for k in ("root_states", "dof_state", "episode_length_buf", "commands"):
                np.testing.assert_array_equal(np.asarray(getattr(se, k))[~hit], before[k][~hit], err_msg="untouched envs: " + k)
            if hit.any():
                assert (np.asarray(se.episode_length_buf)[hit] == 0).all() and (np.asarray(se.reset_buf)[hit] == 1).all()
        so.close(); se.close()


def test_fallen_robot_reports_base_thigh_and_calf_at_once():
    """The contact set of DESIGN.md 4 (one slot per body group of a leg: foot, calf, thigh, hip, base share): a robot lying on its trunk AND on
    the thigh and the calf of one leg reports three non-zero body forces in the same step — the base force check_termination reads
    (legged_robot.py:170-173) is not shadowed by a deeper leg link, and _reward_collision (:1277-1279) sees both links of the leg.  Oracle and the
    host build of the lane programs from the same inputs."""
    from helpers import BASE_B, CALF_B, LYING_KW, THIGH_B, lying_robot_batch, three_body_envs
    M = 96
    so, se = HostSim(load_oracle(), num_envs=M, **LYING_KW), HostSim(load_emu(), num_envs=M, **LYING_KW)
    so.reset_all(); se.reset_all()
    rng = np.random.default_rng(0)
    seen3 = most = 0
    for trial in range(3):
        lying_robot_batch(so, rng)
        for k in STEP_STATE:
            getattr(se, k)[...] = getattr(so, k)
        a = np.zeros((M, 12), np.float32)
        so.step(a); se.step(a)
        ids, no = three_body_envs(so.contact_forces)
        _, ne = three_body_envs(se.contact_forces)
        for e in ids:
            legs = [l for l in range(4) if no[e, THIGH_B[l]] > 0.5 and no[e, CALF_B[l]] > 0.5]
            for b in [BASE_B] + [THIGH_B[l] for l in legs] + [CALF_B[l] for l in legs]:
                assert ne[e, b] > 0.25 * min(no[e, b], 4.0), (e, b, no[e, b], ne[e, b])          # the same three bodies report in the lane programs
        seen3 += len(ids); most = max(most, int((no > 0.1).sum(1).max()))
        assert ((no[:, THIGH_B] > 0.1).sum(1) + (no[:, CALF_B] > 0.1).sum(1)).max() >= 5      # _reward_collision can count past the old model's 4
        # all 19 body forces agree (impulse / 5 ms, so relative to the force scale; lying robots are the ill-conditioned case of DESIGN.md 3)
        fo, fe = np.asarray(so.contact_forces, np.float64), np.asarray(se.contact_forces, np.float64)
        d = np.abs(fo - fe).reshape(M, -1).max(1) / (1.0 + np.abs(fo).reshape(M, -1).max(1))
        assert np.median(d) < 1e-4 and np.quantile(d, 0.99) < 1e-2, np.sort(d)[-4:]          # relative to each env's force scale (impulse / 5 ms)
        np.testing.assert_allclose(np.asarray(so.root_states), np.asarray(se.root_states), atol=5e-3)
    assert seen3 >= 10 and most >= 10, (seen3, most)       # many bodies of one robot at once (the round-2 model held at most 8 + the base)
    so.close(); se.close()


@pytest.mark.parametrize("mesh_type", ["heightfield", "trimesh"])
def test_rough_terrain_one_step_parity(mesh_type):
    """CPU twin of tests/test_gpu_parity.py::test_heightfield_one_step_parity_vs_oracle on the curriculum map: every well-conditioned env-step
    (fp32-vs-fp64 oracle gap below half of ROUGH_BOUND) within that absolute bound, the ill-conditioned rest capped."""
    from helpers import ROUGH_BOUND, StepErrors, check_rough_errors, heightfield_overrides
    M = 48
    _, ov = heightfield_overrides(M, mesh_type=mesh_type)
    so, s64, se = HostSim(load_oracle(), num_envs=M, **ov), HostSim(load_oracle(f64=True), num_envs=M, **ov), HostSim(load_emu(), num_envs=M, **ov)
    for s_ in (so, s64, se):
        s_.reset_all()
    rng = np.random.default_rng(2)
    err = StepErrors(ROUGH_BOUND, bounds=ROUGH_BOUND)
    for it in range(50):
        a = rng.normal(0, 0.6, (M, 12)).astype(np.float32)
        for k in STEP_STATE:
            v = np.asarray(getattr(so, k)); getattr(se, k)[...] = v; getattr(s64, k)[...] = v
        so.step(a); se.step(a); s64.step(a.astype(np.float64))
        err.add(so, se, M, ref64=s64)
        np.testing.assert_array_equal(np.asarray(so.reset_buf), np.asarray(se.reset_buf))
    check_rough_errors(err)
    assert np.abs(np.asarray(so.measured_heights)).max() > 0.02
    for s_ in (so, s64, se):
        s_.close()


@pytest.mark.parametrize("mesh_type", ["heightfield", "trimesh"])
def test_candidate_cull_is_conservative_on_rough_terrain(mesh_type):
    """The lane programs skip a body group's collision candidates when the group provably cannot reach the contact margin — on the height
    field against the highest surface within 0.8 m (go2sim_create's hf_top map) scaled by the map's steepest facet.  The oracle tests every
    candidate.  Robots dropped low and tilted all over the curriculum map (every terrain type and level: slopes, stairs, obstacles, gaps),
    where trunk, hips, thighs and calves touch steps and walls: same contact forces and states from the same inputs."""
    from helpers import LYING_KW, heightfield_overrides, lying_robot_batch
    M = 160
    _, ov = heightfield_overrides(M, mesh_type=mesh_type, max_init_terrain_level=9)
    kw = dict(LYING_KW, **ov)
    so, se = HostSim(load_oracle(), num_envs=M, **kw), HostSim(load_emu(), num_envs=M, **kw)
    so.reset_all(); se.reset_all()
    rng = np.random.default_rng(5)
    nonfoot = 0
    for trial in range(3):
        lying_robot_batch(so, rng)
        # onto the env's own piece of terrain: the reset pose's xy (origin + U(-1, 1)), 4-25 cm above the terrain origin's height
        so.root_states[:, 0:2] = np.asarray(so.env_origins)[:, 0:2] + rng.uniform(-1.5, 1.5, (M, 2))
        so.root_states[:, 2] = np.asarray(so.env_origins)[:, 2] + rng.uniform(0.04, 0.25, M)
        for k in STEP_STATE:
            getattr(se, k)[...] = getattr(so, k)
        a = np.zeros((M, 12), np.float32)
        so.step(a); se.step(a)
        fo, fe = np.asarray(so.contact_forces, np.float64), np.asarray(se.contact_forces, np.float64)
        d = np.abs(fo - fe).reshape(M, -1).max(1) / (1.0 + np.abs(fo).reshape(M, -1).max(1))
        # (a facet / wall switch at a cell boundary is the model's own discontinuity: a handful of env-steps may differ)
        assert np.median(d) < 1e-4 and (d > 1e-2).sum() <= 3, np.sort(d)[-5:]
        # the SAME set of bodies is in contact: a culled group that should have been tested would show up as a missing body
        miss = ((np.linalg.norm(fo, axis=2) > 1.0) & (np.linalg.norm(fe, axis=2) == 0.0)).sum()
        assert miss <= 1, miss
        nonfoot += int((np.linalg.norm(fo[:, [0, 1, 2, 3, 4, 5, 7, 8, 9, 11, 12, 13, 15, 16, 17]], axis=2) > 1.0).sum())
    assert nonfoot > 200      # trunk / hip / thigh / calf contacts were what was compared
    so.close(); se.close()
