"""Known-answer tests of the physics model (the part of the path the reference delegates to Isaac Gym, so nothing of
the reference can pin it).  fp64 oracle build unless stated."""
import os

import numpy as np
import pytest

from helpers import ROOT, HostSim, load_oracle

N = 8


def _randomise(s, rng, scale_v=1.0):
    s.reset_all()
    s.dof_state[:, :, 1] = rng.uniform(-3, 3, (s.N, 12)) * scale_v
    s.root_states[:, 7:13] = rng.uniform(-1, 1, (s.N, 6)) * scale_v
    q = rng.normal(size=(s.N, 4)); s.root_states[:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)


def test_aba_equals_dense_inverse_dynamics_fp64():
    """Featherstone's floating-base ABA (RBDA table 9.4) vs M^-1 (tau - C) from CRBA + RNEA + Cholesky: two derivations."""
    s = HostSim(load_oracle(f64=True), num_envs=N)
    rng = np.random.default_rng(1)
    _randomise(s, rng)
    for e in range(N):
        a1, a2, M, en = s.debug_dynamics(e, rng.uniform(-20, 20, 12))
        np.testing.assert_allclose(a1, a2, atol=1e-9, rtol=1e-10)
        np.testing.assert_allclose(M, M.T, atol=1e-14)
        assert np.linalg.eigvalsh(M).min() > 0
        assert abs(en[5] - (15.019 + s.added_base_mass[e] + (s.link_mass_ratio[e] - 1) @ np.array([0.001, 0.001] + [0.678, 1.152, 0.154, 0.04] * 4))) < 1e-9
    s.close()


def _free_flight_drift(dt, steps):
    s = HostSim(load_oracle(f64=True), num_envs=N, gravity=[0, 0, 0], kp=[0] * 12, kd=[0] * 12, push_robots=0, randomize_action_delay=0,
                joint_limit_margin=-10.0, sim_dt=dt, decimation=1)
    rng = np.random.default_rng(2)
    _randomise(s, rng, 0.5)
    s.root_states[:, 2] = 5.0
    e0 = np.array([s.debug_dynamics(e, np.zeros(12))[3] for e in range(N)])
    for _ in range(steps):
        s.simulate()
    e1 = np.array([s.debug_dynamics(e, np.zeros(12))[3] for e in range(N)])
    s.close()
    mom = np.abs(e1[:, 2:5] - e0[:, 2:5]).max() / np.abs(e0[:, 2:5]).max()
    ke = np.abs(e1[:, 0] / e0[:, 0] - 1).max()
    return mom, ke


def test_free_flight_conserves_momentum_and_energy_fp64():
    """No gravity, no contact, zero gains, tumbling base and swinging legs for 0.2 s: world linear momentum and kinetic
    energy are conserved up to the integrator's error, which must be small and shrink ~linearly with dt (first order)."""
    m1, k1 = _free_flight_drift(1e-3, 200)
    m2, k2 = _free_flight_drift(5e-4, 400)
    assert m1 < 2e-3 and k1 < 0.05, (m1, k1)
    assert m2 < 0.65 * m1 and k2 < 0.65 * k1, (m1, m2, k1, k2)


def test_free_fall_is_exact():
    s = HostSim(load_oracle(f64=True), num_envs=N, kp=[0] * 12, kd=[0] * 12, push_robots=0, randomize_action_delay=0, joint_limit_margin=-10.0)
    s.reset_all()
    s.root_states[:, 2] = 10.0; s.root_states[:, 7:13] = 0; s.dof_state[:, :, 1] = 0
    z0 = s.root_states[:, 2].copy()
    for _ in range(5):
        s.simulate()                       # 20 substeps of 5 ms
    # the system COM falls with g; with all joints free and started at rest the base follows to first order
    vz = s.root_states[:, 9]
    assert np.all(np.abs(vz + 9.81 * 0.1) < 0.25)
    s.close()


def test_standing_robot_carries_its_weight():
    """PD at the default pose on the plane: after settling, sum of vertical contact forces = m g within 3 %, base height
    in the standing range, nothing explodes (fp32 build, the one compared with the GPU)."""
    s = HostSim(load_oracle(), num_envs=32, push_robots=0, add_noise=0)
    s.reset_all()
    a = np.zeros((32, 12), np.float32)
    fz = []
    for i in range(150):
        s.step(a)
        if i >= 100:
            fz.append(s.contact_forces[:, :, 2].sum(1).copy())
    alive = ~np.asarray(s.reset_buf).astype(bool)
    w = (15.019 + s.added_base_mass + (s.link_mass_ratio - 1) @ np.array([0.001, 0.001] + [0.678, 1.152, 0.154, 0.04] * 4)) * 9.81
    ratio = (np.mean(fz, axis=0) / w)[alive]
    assert np.abs(np.median(ratio) - 1.0) < 0.03, np.median(ratio)
    z = s.root_states[:, 2][alive]
    assert 0.15 < np.median(z) < 0.40 and np.isfinite(np.asarray(s.obs_buf)).all()
    assert np.abs(s.dof_state[:, :, 1]).max() < 25.0
    s.close()


# The behavioural test with the reference's pretrained policy lives in tests/test_export.py (it runs from committed fixtures,
# through this build's own loader, on the oracle here and on the HIP kernels on the GPU box).


@pytest.mark.parametrize("which", ["oracle", "lane_emulation"])
def test_flipped_robots_settle(which):
    """init_state.turn_over: robots dropped on their back / side from 10-21 cm come to rest on the base, head and hip geometry:
    finite state, small velocities, base stays above the ground and below its drop height; nobody is reset (:174)."""
    from helpers import load_emu
    lib = load_oracle() if which == "oracle" else load_emu()
    N = 24
    s = HostSim(lib, num_envs=N, turn_over=1, turn_over_proportions=np.array([0.5, 0.5, 0.0], np.float32), push_robots=0, seed=5)
    s.reset_all()
    roll0 = np.abs(2 * np.arctan2(np.linalg.norm(np.asarray(s.root_states)[:, 3:5], axis=1), np.abs(np.asarray(s.root_states)[:, 6]) + 1e-9))
    assert (np.asarray(s.turn_over_timer) > 0).all() and (roll0 > 1.0).all()
    a = np.zeros((N, 12), np.float32)
    for _ in range(100):
        s.step(a)
        assert not np.asarray(s.reset_buf).any()
    root = np.asarray(s.root_states)
    assert np.isfinite(root).all() and np.isfinite(np.asarray(s.obs_buf)).all()
    assert (root[:, 2] > 0.02).all() and (root[:, 2] < 0.45).all(), np.sort(root[:, 2])
    assert np.abs(root[:, 7:10]).max() < 2.0
    assert (np.asarray(s.commands)[:, :3] == 0).all()                       # zero commands while the turn-over timer runs (:586-590)
    assert (np.asarray(s.turn_over_timer) > 0).all() and (np.asarray(s.turn_over_timer) <= 3.0 + 1e-3).all()     # 5 s (back) or 3 s (side) minus the 2 s simulated
    s.close()


@pytest.mark.parametrize("which", ["oracle", "lane_emulation"])
def test_coulomb_friction_cone(which):
    """Tilted gravity = a slope; friction randomisation off, so mu = 0.5 * (1 + 1) = 1.  (i) every foot force stays inside the Coulomb cone
    |F_t| <= mu F_n at every step on both slopes; (ii) with a firmer stance (Kp 40, Kd 1: the default 20 / 0.5 sags and tips over on a slope
    without a policy; much stiffer gains leave the stable range of the explicit 200 Hz PD) the robot holds its place on an 11 degree
    slope (tan = 0.2 << mu); (iii) on a 61 degree slope (tan = 1.8 >> mu) it goes downhill."""
    from helpers import load_emu
    lib = load_oracle() if which == "oracle" else load_emu()
    out = {}
    for tan_t in (0.2, 1.8):
        th = np.arctan(tan_t)
        g = [9.81 * np.sin(th), 0.0, -9.81 * np.cos(th)]                  # down-slope = +x
        s = HostSim(lib, num_envs=6, gravity=g, push_robots=0, randomize_friction=0, randomize_restitution=0, randomize_action_delay=0,
                    randomize_pd_gains=0, randomize_motor_strength=0, kp=[40.0] * 12, kd=[1.0] * 12, seed=2)
        s.reset_all()
        s.root_states[:, 3:7] = np.array([0, 0, 0, 1], np.float32); s.root_states[:, 7:13] = 0; s.root_states[:, 2] = 0.34
        a = np.zeros((6, 12), np.float32)
        worst, was_reset = 0.0, np.zeros(6, bool)
        for it in range(100):
            if it == 50:
                x0 = np.asarray(s.root_states)[:, 0].copy()                # settled on the feet by now
            s.step(a)
            was_reset |= np.asarray(s.reset_buf) > 0
            f = np.asarray(s.contact_forces)[:, [6, 10, 14, 18]].astype(np.float64)
            ft, fn = np.hypot(f[..., 0], f[..., 1]), f[..., 2]
            assert (fn >= -1e-3).all()
            worst = max(worst, float((ft - 1.0 * fn).max()))
        out[tan_t] = (np.asarray(s.root_states)[:, 0] - x0, worst, was_reset)
        s.close()
    assert out[0.2][1] < 0.05 and out[1.8][1] < 0.05, (out[0.2][1], out[1.8][1])         # cone respected (N; forces are impulse / 5 ms)
    dx, _, rs = out[0.2]
    assert rs.sum() <= 1 and np.abs(dx[~rs]).max() < 0.02, (dx, rs)                        # 1 s on the gentle slope: holds (an env whose episode
    #                                                                                        ended in the window was re-spawned elsewhere)
    assert (out[1.8][0][~out[1.8][2]] > 0.4).all(), out[1.8][0]


def test_pd_drive_reaches_its_target_in_the_air():
    """Robot held in zero gravity: the PD actuation (:594-618) drives every joint to q0 + action_scale * a and holds it (damped, no overshoot
    beyond a few percent), torques saturating at the URDF effort limits on the way."""
    s = HostSim(load_oracle(), num_envs=4, gravity=[0, 0, 0], push_robots=0, randomize_action_delay=0, randomize_motor_strength=0, randomize_pd_gains=0,
                randomize_motor_zero_offset=0)
    s.reset_all()
    s.root_states[:, 2] = 3.0
    a = np.tile(np.array([0.4, -0.6, 0.8] * 4, np.float32), (4, 1))
    target = np.array(list(s.cfg.default_dof_pos), np.float32) + 0.25 * a[0]
    peak = np.zeros(12)
    for it in range(60):
        s.step(a)
        peak = np.maximum(peak, np.abs(np.asarray(s.torques)).max(0))
    q = np.asarray(s.dof_state)[:, :, 0]
    np.testing.assert_allclose(q, np.tile(target, (4, 1)), atol=5e-3)
    assert np.abs(np.asarray(s.dof_state)[:, :, 1]).max() < 0.05
    assert (peak <= np.array([23.7, 23.7, 35.55] * 4) + 1e-4).all()
    s.close()


@pytest.mark.parametrize("which", ["oracle", "lane_emulation"])
def test_joint_stops_hold_against_saturated_motors(which):
    """Robot floating in zero gravity, one leg per env driven with saturated torque into its upper (envs 0-3) or lower (4-7) URDF stops, arriving
    at the velocity limit (30.1 / 20.07 rad/s): the limit rows of the velocity-level solve stop every joint — transient overshoot below
    one substep's travel at that speed (0.15 rad), then at rest within 0.015 rad of the stop while the motor keeps pushing."""
    from helpers import load_emu
    lib = load_oracle() if which == "oracle" else load_emu()
    lo = np.array([-1.0472, -1.5708, -2.7227, -1.0472, -1.5708, -2.7227, -1.0472, -0.5236, -2.7227, -1.0472, -0.5236, -2.7227])
    hi = np.array([1.0472, 3.4907, -0.83776, 1.0472, 3.4907, -0.83776, 1.0472, 4.5379, -0.83776, 1.0472, 4.5379, -0.83776])
    s = HostSim(lib, num_envs=8, gravity=[0, 0, 0], push_robots=0, randomize_action_delay=0, randomize_motor_strength=0, randomize_pd_gains=0,
                randomize_motor_zero_offset=0)
    s.reset_all()
    s.root_states[:, 2] = 3.0; s.root_states[:, 7:13] = 0
    a = np.zeros((8, 12), np.float32)
    for e in range(8):
        a[e, 3 * (e % 4):3 * (e % 4) + 3] = 40.0 if e < 4 else -40.0          # target 10 rad beyond: the torque stays at the effort limit
    over = np.zeros(8)
    for _ in range(100):
        s.step(a)
        q = np.asarray(s.dof_state)[:, :, 0]
        over = np.maximum(over, np.maximum(q - hi, lo - q).max(1))
    q, qd, tq = np.asarray(s.dof_state)[:, :, 0], np.asarray(s.dof_state)[:, :, 1], np.asarray(s.torques)
    assert over.max() < 0.1, over
    for e in range(8):
        sl = slice(3 * (e % 4), 3 * (e % 4) + 3)
        assert np.abs(q[e, sl] - (hi if e < 4 else lo)[sl]).max() < 0.015 and np.abs(qd[e, sl]).max() < 0.02, (e, q[e, sl], qd[e, sl])
        np.testing.assert_allclose(np.abs(tq[e, sl]), [23.7, 23.7, 35.55], atol=1e-4)
    s.close()


@pytest.mark.parametrize("which", ["oracle", "lane_emulation"])
def test_state_stays_finite_in_a_runaway(which):
    """The worst input the API admits — a free-floating robot, every motor saturated in one direction, then the other, for 160 steps —
    spins the base up until the explicit integration diverges.  asset.max_linear_velocity / max_angular_velocity (legged_robot_config.py:
    131-132, clamped on the base twist as PhysX does) bound it: every buffer stays finite and the clamp holds, so one env can never
    poison a batch with NaN."""
    from helpers import load_emu
    lib = load_oracle() if which == "oracle" else load_emu()
    s = HostSim(lib, num_envs=4, gravity=[0, 0, 0], push_robots=0, randomize_action_delay=0, max_linear_velocity=50.0, max_angular_velocity=100.0)
    s.reset_all()
    s.root_states[:, 2] = 3.0
    for sign in (1.0, -1.0):
        a = np.full((4, 12), sign * 40.0, np.float32)
        for _ in range(80):
            s.step(a)
            for k in ("root_states", "dof_state", "obs_buf", "privileged_obs_buf", "rew_buf", "torques", "contact_forces"):
                assert np.isfinite(np.asarray(getattr(s, k))).all(), k
            r = np.asarray(s.root_states)
            assert (np.linalg.norm(r[:, 7:10], axis=1) <= 50.0 * (1 + 1e-5)).all() and (np.linalg.norm(r[:, 10:13], axis=1) <= 100.0 * (1 + 1e-5)).all()
    s.close()
    d = lib.abi.Cfg(); lib.go2sim_default_cfg(d)
    assert d.max_linear_velocity == 1000.0 and d.max_angular_velocity == 1000.0        # the reference's asset options


@pytest.mark.skipif(not os.path.exists("/root/reference/resources/robots/go2/urdf/go2.urdf"), reason="container-only: needs the reference's URDF")
def test_model_table_is_what_the_generator_derives_from_the_urdf(tmp_path):
    """include/go2_model_data.h (numbers only) is regenerated from the Go2 URDF by tools/gen_go2_model.py and must equal the committed
    file: masses, COMs, inertias, joint origins / axes / limits / efforts and the collision spheres are the URDF's, not hand-typed."""
    import subprocess
    import sys
    out = tmp_path / "go2_model_data.h"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_go2_model.py"), "/root/reference/resources/robots/go2/urdf/go2.urdf", str(out)],
                   check=True, capture_output=True, cwd=ROOT)
    assert out.read_text() == open(os.path.join(ROOT, "include", "go2_model_data.h")).read()
