// go2sim_impl.cpp — the C ABI of include/go2sim.h on top of the lane programs.
//
// Built two ways from this one source:
//   hipcc --offload-arch=gfx950           -> libgo2sim_hip.so : THE PRODUCT.  One HIP kernel per env step
//       (go2_step_kernel), 256-thread workgroups = 16 envs x (4 legs x 4 sub-lanes), one DPP row per env, robot tables
//       staged in LDS, cross-lane sums / exchanges with DPP, all per-env state field-major (SoA) in HBM.
//   g++ -DGO2_EMU                          -> libgo2sim_emu.so : TEST-ONLY host emulation: the very same kernel body
//       (go2_step_body) run lane by lane as fibres with the cross-lane primitives as rendezvous (go2_xlane.h), so the
//       kernel arithmetic can be checked against the oracle on a machine without a GPU.  Never loaded by the product (go2_rl_gym_amd/_lib.py
//       accepts only a library whose go2sim_is_device_library() is 1).
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../../include/go2sim.h"
#ifdef GO2_EMU
#define GO2_SHUFFLE_FN static inline
#else
#include <hip/hip_runtime.h>
#define GO2_SHUFFLE_FN static __host__ __device__ inline
#endif
#include "../../include/go2sim_shuffle.h"
#include "../../include/go2sim_defaults.h"
#include "../../include/go2sim_rng.h"
#define GO2_XLANE_IMPLEMENTATION
#include "go2_lane.h"
#include "go2_post.h"

#ifndef GO2_EMU
#include <hip/hip_runtime.h>
#endif

static thread_local char g_err[512] = "";
#define FAIL(code, ...) do { snprintf(g_err, sizeof(g_err), __VA_ARGS__); return (code); } while (0)

#ifndef GO2_EMU
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) FAIL(GO2SIM_EDEVICE, "%s: %s", #x, hipGetErrorString(e_)); } while (0)
#endif

enum { MODE_PHYS = 1, MODE_POST = 2, MODE_RESET_ALL = 4 };

// ------------------------------------------------------------------------------------------------------
// the per-lane driver: ONE function body, go2_step_body<MODE>, is both the HIP kernel (go2_step_kernel) and — run lane by lane as
// fibres (go2_xlane.h) — the host emulation; only the cross-lane primitives differ between the two builds.
// Workgroup = 256 threads = 4 waves = 16 environments; one 16-lane DPP row = one environment, row lane = leg * 4 + sub.
// ------------------------------------------------------------------------------------------------------
#define GO2_WG_ENVS 16
#define GO2_WG_THREADS (16 * GO2_WG_ENVS)
// The lane context is kept as THREE separate objects (not one struct): the compiler's scalar-replacement pass gives up on a
// single 2.4 KB aggregate with thousands of uses and would leave all of it in scratch memory.
struct LaneAux { float kp[3], kd[3], q0[3], zoff[3], strength[3], act_new[3], act_old[3]; int start; };   // kp/kd: gain x per-env multiplier (loaded once, not per substep)
#define LANE_PARAMS LegPhys& ph_, LegPost& po_, LaneAux& ax
#define GO2_NUM_GROUPS 44     // Philox groups of one env-step (go2sim_rng.h: 0..42) rounded up
// LDS: robot link / collision tables (per-lane leg index -> ds_read), this step's scalars, and per environment the table of drawn uniforms
struct Go2Shared {
  Go2Tables tab; Go2Step S; float ucache[GO2_WG_ENVS][GO2_NUM_GROUPS][4];
  Go2RowsLds rows;      // constraint rows of the non-foot contact slots (go2_tables.h)
  float hcache[GO2_WG_ENVS][GO2_NUM_HEIGHT_POINTS + 5];      // postA's height samples, read back by postB's observation rows
};
static_assert(GO2_WG_LANES == GO2_WG_THREADS, "row storage is per lane of the workgroup");

GO2_HD void lane_load_phys(LANE_PARAMS, const Go2Tables& tab, const Go2PtrsK& p, const Go2Launch& L, const Go2Step& S, const float* actions_in, float u_delay, int e, int lane, int sub) {
  const int N = L.N;
  const LegTab& t = tab.leg[lane];
  LegPhys& ph = ph_;
  ph.leg = lane; ph.sub = sub;
  ph.pw = v3(F2D(p.root, 0, e), F2D(p.root, 1, e), F2D(p.root, 2, e));
  ph.qx = F2D(p.root, 3, e); ph.qy = F2D(p.root, 4, e); ph.qz = F2D(p.root, 5, e); ph.qw = F2D(p.root, 6, e);
  ph.vw = v3(F2D(p.root, 7, e), F2D(p.root, 8, e), F2D(p.root, 9, e));
  ph.ww = v3(F2D(p.root, 10, e), F2D(p.root, 11, e), F2D(p.root, 12, e));
  float cl = L.clip_actions;
  _Pragma("unroll") for (int j = 0; j < 3; ++j) {
    int d = 3 * lane + j;
    ph.q[j] = F2D(p.dof, d, e); ph.qd[j] = F2D(p.dof, 12 + d, e);
    ax.kp[j] = L.kp[d] * F2D(p.kp_mul, d, e); ax.kd[j] = L.kd[d] * F2D(p.kd_mul, d, e); ax.q0[j] = L.q0[d];
    ax.zoff[j] = F2D(p.zero_off, d, e); ax.strength[j] = F2D(p.strength, d, e);
    float a = actions_in ? actions_in[(size_t)e * 12 + d] : F2D(p.actions, d, e);
    a = fminf(fmaxf(a, -cl), cl);                  // legged_robot.py:67-68
    ax.act_new[j] = a; if (sub == 0) F2D(p.actions, d, e) = a;      // (idempotent: a sub-lane that reads the clipped value back clips it to itself)
    ax.act_old[j] = F2D(p.last_actions, d, e);
    ph.lam_foot[0 + j] = F3D(p.foot_impulse, 4, lane, j, e);
  }
  // per-env inertial parameters (legged_robot.py:379-402, recomputeInertia=True modelled as inertia ~ mass)
  float r_hip = F2D(p.mass_ratio, t.mass_ratio_index[0], e), r_thigh = F2D(p.mass_ratio, t.mass_ratio_index[1], e);
  float r_calf = F2D(p.mass_ratio, t.mass_ratio_index[2], e), r_foot = F2D(p.mass_ratio, t.mass_ratio_index[3], e);
  auto rb10 = [](const float* b) { RB r; r.m = b[0]; r.h = v3(b[1], b[2], b[3]); r.J.xx = b[4]; r.J.yy = b[5]; r.J.zz = b[6]; r.J.xy = b[7]; r.J.xz = b[8]; r.J.yz = b[9]; return r; };
  ph.Lhip = r_hip * rb10(t.body[0]); ph.Lthigh = r_thigh * rb10(t.body[1]);
  ph.Lcalf = r_calf * rb10(t.body[2]) + r_foot * rb10(t.body[3]);
  const BaseTab& bt = tab.base;
  float m = bt.m0 + p.added_mass[e], ratio = m / bt.m0;
  V3 cc = v3(bt.c0[0] + F2D(p.added_com, 0, e), bt.c0[1] + F2D(p.added_com, 1, e), bt.c0[2] + F2D(p.added_com, 2, e));
  S3 Ic = {bt.Ic0[0] * ratio, bt.Ic0[1] * ratio, bt.Ic0[2] * ratio, bt.Ic0[3] * ratio, bt.Ic0[4] * ratio, bt.Ic0[5] * ratio};
  ph.Ibase = rb_from_com(m, cc, Ic) + F2D(p.mass_ratio, 0, e) * rb10(bt.head[0]) + F2D(p.mass_ratio, 1, e) * rb10(bt.head[1]);
  ph.mu = 0.5f * (L.terrain_friction + p.friction[e]);
  ph.rest = 0.5f * (L.terrain_restitution + p.restitution[e]);
  // action delay (legged_robot.py:71-78)
  ax.start = 0;
  if (L.rand_delay) {
    const float u = S.injected ? S.injected[(size_t)e * GO2_NUM_UNIFORMS + GO2_U_DELAY] : u_delay;      // slot GO2_U_DELAY = group 0, word 0 (drawn at kernel start)
    ax.start = (int)(u * (float)(L.decimation + 1)); if (ax.start > L.decimation) ax.start = L.decimation;
  }
}

GO2_HD void quat_mul(const float* a, const float* b, float* o) {  // (x,y,z,w)
  o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
  o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}

// after the last substep: forward kinematics at the new state, API tensors, PhysOut.  fbase = this leg's contribution to the base / head
// contact forces (3 bodies x 3), to be summed over the legs by the caller.  Inside a leg's quad, sub-lane k writes the row of the leg's
// body k (hip, thigh, calf, foot) of rigid_body_states / contact_forces; sub-lane 0 writes the leg's DOF state.
GO2_HD void lane_finish_phys(LANE_PARAMS, const Go2Tables& tab, const Go2PtrsK& p, const Go2Launch& L, int e, int lane, int sub, float* fbase) {
  const int N = L.N;
  const LegTab& t = tab.leg[lane];
  LegPhys& ph = ph_; PhysOut& o = po_.o;
  M3 Rwb = quat_to_m3(ph.qx, ph.qy, ph.qz, ph.qw);
  V3 wb = mulT(Rwb, ph.ww), vb = mulT(Rwb, ph.vw);
  float s1, c1, s2, c2, s3, c3; go2_sincos(ph.q[0], &s1, &c1); go2_sincos(ph.q[1], &s2, &c2); go2_sincos(ph.q[2], &s3, &c3);
  float c23 = c2 * c3 - s2 * s3, s23 = s2 * c3 + c2 * s3;
  M3 R1, R2, R3; R1.x = v3(1, 0, 0); R1.y = v3(0, c1, s1); R1.z = v3(0, -s1, c1);
  R2.x = c2 * R1.x - s2 * R1.z; R2.y = R1.y; R2.z = s2 * R1.x + c2 * R1.z;
  R3.x = c23 * R1.x - s23 * R1.z; R3.y = R1.y; R3.z = s23 * R1.x + c23 * R1.z;
  V3 p1 = v3(t.o1[0], t.o1[1], t.o1[2]), p2 = p1 + mul(R1, v3(t.o2[0], t.o2[1], t.o2[2])), p3 = p2 + mul(R2, v3(t.o3[0], t.o3[1], t.o3[2]));
  V3 a1 = v3(1, 0, 0), a2 = R1.y;
  // link spatial velocities in the common frame
  SV V0 = sv(wb, vb);
  SV V1 = V0 + ph.qd[0] * sv(a1, cross(p1, a1)), V2 = V1 + ph.qd[1] * sv(a2, cross(p2, a2)), V3l = V2 + ph.qd[2] * sv(a2, cross(p3, a2));
  V3 pf = p3 + mul(R3, v3(t.foot_off[0], t.foot_off[1], t.foot_off[2]));
  float qb[4] = {ph.qx, ph.qy, ph.qz, ph.qw}, qh[4], qt[4], qc[4];
  float hs1, hc1, hs2, hc2, hs3, hc3; go2_sincos(0.5f * ph.q[0], &hs1, &hc1); go2_sincos(0.5f * ph.q[1], &hs2, &hc2); go2_sincos(0.5f * ph.q[2], &hs3, &hc3);
  float h1[4] = {hs1, 0, 0, hc1}, h2[4] = {0, hs2, 0, hc2}, h3[4] = {0, hs3, 0, hc3};
  quat_mul(qb, h1, qh); quat_mul(qh, h2, qt); quat_mul(qt, h3, qc);
  {   // body `sub` of this leg: hip, thigh, calf, foot
    const V3 org = sel(sub == 0, p1, sel(sub == 1, p2, sel(sub == 2, p3, pf)));
    const SV vel = sv(sel(sub == 0, V1.a, sel(sub == 1, V2.a, V3l.a)), sel(sub == 0, V1.l, sel(sub == 1, V2.l, V3l.l)));
    float qq[4];
    _Pragma("unroll") for (int i = 0; i < 4; ++i) qq[i] = sub == 0 ? qh[i] : (sub == 1 ? qt[i] : qc[i]);
    const int b = t.body_index[0] + sub;
    const V3 pos = ph.pw + mul(Rwb, org);
    const V3 lv = mul(Rwb, vel.l + cross(vel.a, org)), av = mul(Rwb, vel.a);
    const float r13[13] = {pos.x, pos.y, pos.z, qq[0], qq[1], qq[2], qq[3], lv.x, lv.y, lv.z, av.x, av.y, av.z};
    if (sub == 3 || L.full_body_states) _Pragma("unroll") for (int i = 0; i < 13; ++i) F3D(p.rigid, 19, b, i, e) = r13[i];
  }
  {   // the foot row is what post-physics reads: every sub-lane keeps it
    const V3 pos = ph.pw + mul(Rwb, pf);
    o.foot_pos = pos; o.foot_vel = mul(Rwb, V3l.l + cross(V3l.a, pf));
  }
  if (lane < 3 && sub == 0 && L.full_body_states) {  // base, Head_upper, Head_lower rows
    V3 off = v3(tab.base.body_off[lane][0], tab.base.body_off[lane][1], tab.base.body_off[lane][2]);
    V3 pos = ph.pw + mul(Rwb, off); V3 lv = mul(Rwb, vb + cross(wb, off));
    float r13[13] = {pos.x, pos.y, pos.z, ph.qx, ph.qy, ph.qz, ph.qw, lv.x, lv.y, lv.z, ph.ww.x, ph.ww.y, ph.ww.z};
    _Pragma("unroll") for (int i = 0; i < 13; ++i) F3D(p.rigid, 19, lane, i, e) = r13[i];
  }
  // contact forces of this leg's bodies (one contact per body group, compacted into virtual slots: go2_lane.h phaseC); base / head parts go
  // through the leg sum
  V3 zero = v3(0, 0, 0);
  const V3 Fv[GO2_NTYPE] = {ph.virtual_force<0>(L.sim_dt), ph.virtual_force<1>(L.sim_dt), ph.virtual_force<2>(L.sim_dt), ph.virtual_force<3>(L.sim_dt)};
  auto body_force = [&](int b) { V3 f = zero; _Pragma("unroll") for (int k = 0; k < GO2_NTYPE; ++k) f = f + sel(ph.body_v[k] == b, Fv[k], zero); return f; };
  o.Fhip = body_force(t.body_index[0]); o.Fthigh = body_force(t.body_index[1]); o.Fcalf = body_force(t.body_index[2]);
  o.Ffoot = ph.force_foot;
  {
    const V3 f = sel(sub == 0, o.Fhip, sel(sub == 1, o.Fthigh, sel(sub == 2, o.Fcalf, o.Ffoot)));
    const int b = t.body_index[0] + sub; F3D(p.contact, 19, b, 0, e) = f.x; F3D(p.contact, 19, b, 1, e) = f.y; F3D(p.contact, 19, b, 2, e) = f.z;
  }
  _Pragma("unroll") for (int b = 0; b < 3; ++b) { V3 f = body_force(b); fbase[3 * b] = f.x; fbase[3 * b + 1] = f.y; fbase[3 * b + 2] = f.z; }
  o.pw = ph.pw; o.qx = ph.qx; o.qy = ph.qy; o.qz = ph.qz; o.qw = ph.qw; o.vw = ph.vw; o.ww = ph.ww;
  _Pragma("unroll") for (int j = 0; j < 3; ++j) {
    int d = 3 * lane + j;
    o.q[j] = ph.q[j]; o.qd[j] = ph.qd[j]; o.tau[j] = ph.tau[j];
    if (sub == 0) { F2D(p.dof, d, e) = ph.q[j]; F2D(p.dof, 12 + d, e) = ph.qd[j]; F2D(p.torques, d, e) = ph.tau[j]; F3D(p.foot_impulse, 4, lane, j, e) = ph.lam_foot[j]; }
  }
  if (lane == 0 && sub == 0) {
    float r13[13] = {o.pw.x, o.pw.y, o.pw.z, o.qx, o.qy, o.qz, o.qw, o.vw.x, o.vw.y, o.vw.z, o.ww.x, o.ww.y, o.ww.z};
    _Pragma("unroll") for (int k = 0; k < 13; ++k) F2D(p.root, k, e) = r13[k];
  }
}
// fbase_sum = leg-summed base/head forces
GO2_HD void lane_store_base_forces(LANE_PARAMS, const Go2PtrsK& p, const Go2Launch& L, int e, int lane, int sub, const float* fbase_sum) {
  const int N = L.N;
  po_.o.Fbase = v3(fbase_sum[0], fbase_sum[1], fbase_sum[2]);
  if (lane < 3 && sub == 0) for (int k = 0; k < 3; ++k) F3D(p.contact, 19, lane, k, e) = fbase_sum[3 * lane + k];
}
// post-only entry: rebuild PhysOut from the API tensors
GO2_HD void lane_load_physout(LANE_PARAMS, const Go2Tables& tab, const Go2PtrsK& p, const Go2Launch& L, int e, int lane) {
  const int N = L.N; const LegTab& t = tab.leg[lane]; PhysOut& o = po_.o;
  o.pw = v3(F2D(p.root, 0, e), F2D(p.root, 1, e), F2D(p.root, 2, e));
  o.qx = F2D(p.root, 3, e); o.qy = F2D(p.root, 4, e); o.qz = F2D(p.root, 5, e); o.qw = F2D(p.root, 6, e);
  o.vw = v3(F2D(p.root, 7, e), F2D(p.root, 8, e), F2D(p.root, 9, e)); o.ww = v3(F2D(p.root, 10, e), F2D(p.root, 11, e), F2D(p.root, 12, e));
  _Pragma("unroll") for (int j = 0; j < 3; ++j) { int d = 3 * lane + j; o.q[j] = F2D(p.dof, d, e); o.qd[j] = F2D(p.dof, 12 + d, e); o.tau[j] = F2D(p.torques, d, e); }
  auto F = [&](int b) { return v3(F3D(p.contact, 19, b, 0, e), F3D(p.contact, 19, b, 1, e), F3D(p.contact, 19, b, 2, e)); };
  o.Fhip = F(t.body_index[0]); o.Fthigh = F(t.body_index[1]); o.Fcalf = F(t.body_index[2]); o.Ffoot = F(t.body_index[3]); o.Fbase = F(0);
  int fb = t.body_index[3];
  o.foot_pos = v3(F3D(p.rigid, 19, fb, 0, e), F3D(p.rigid, 19, fb, 1, e), F3D(p.rigid, 19, fb, 2, e));
  o.foot_vel = v3(F3D(p.rigid, 19, fb, 7, e), F3D(p.rigid, 19, fb, 8, e), F3D(p.rigid, 19, fb, 9, e));
}
GO2_HD void lane_init_post(LANE_PARAMS, const GO2_AS3 uint8_t* codes, GO2_AS3 float (*uc)[4], GO2_AS3 float* hc, const Go2PtrsK* p, const Go2Launch* L, const Go2Step* S, int e, int lane, int sub) {
  po_.codes = codes; po_.uc = uc; po_.hc = hc;
  po_.e = e; po_.lane = lane; po_.sub = sub; po_.lane16 = lane * 4 + sub; po_.N = L->N; po_.P = p; po_.L = L; po_.S = S;
#if defined(__HIP_DEVICE_COMPILE__) && defined(GO2_KBENCH_STAMPS)
  po_.dbg = nullptr;
#endif
  po_.skip_contact_filters = false; po_.api_reset = false; po_.yaw_seen = false; po_.out = Go2StepOutputs{}; po_.new_lc = po_.new_lc2 = 0; po_.new_fat = 0.f;
}
// reset_idx(all envs) without a step (base_task.py:82-84): postB's reset branch with reset forced on
GO2_HD void lane_reset_all(LANE_PARAMS, const Go2Tables& tab, const Go2PtrsK& p, const Go2Launch& L, const Go2Step& S, GO2_AS3 float (*uc)[4], GO2_AS3 float* hc, int e, int lane, int sub) {
  const int N = L.N;
  lane_load_physout(ph_, po_, ax, tab, p, L, e, lane);
  lane_init_post(ph_, po_, ax, (const GO2_AS3 uint8_t*)tab.slot_code, uc, hc, &p, &L, &S, e, lane, sub);
  LegPost& po = po_;
  po.skip_contact_filters = true; po.api_reset = S.initial_reset == 2;
  // load what postA would have loaded, without advancing any clock
  po.ep_len = p.ep_len[e]; po.timer = p.cmd_timer[e];
  _Pragma("unroll") for (int k = 0; k < 4; ++k) po.cmd[k] = F2D(p.commands, k, e);
  po.acc[0] = F2D(p.cmd_xy_acc, 0, e); po.acc[1] = F2D(p.cmd_xy_acc, 1, e);
  po.stop_heading = p.stop_heading[e]; po.last_limit = p.last_is_limit_vel[e];
  _Pragma("unroll") for (int j = 0; j < 3; ++j) { int d = 3 * lane + j; po.act[j] = 0; po.last_act[j] = 0; po.llast_act[j] = F2D(p.last_last_actions, d, e); po.last_dv[j] = 0; }
  po.max_move = p.max_move[e];
  po.to_timer = L.turn_over ? p.to_timer[e] : 0.f;
  po.load_terrain_fields();
}

// ------------------------------------------------------------------------------------------------------
// The kernel body.  `sh` is the workgroup's LDS block; bid / tid = blockIdx.x / threadIdx.x.
template <int MODE>
GO2_HD void go2_step_body(Go2Shared& sh, const Go2DevBlock* __restrict__ blk, const float* __restrict__ actions_in, int initial_reset, const Go2StepOutputs& outs, int bid, int tid) {
  const Go2PtrsK& p = *(const Go2PtrsK*)&blk->p; const Go2Launch& L = blk->L;   // uniform addresses -> scalar loads; pointers typed global
#if defined(__HIP_DEVICE_COMPILE__)
  // Touch the per-environment state this workgroup is about to read (one 64-byte segment per field-major field: thread k touches field k
  // for the workgroup's first environment) BEFORE the table staging and its barrier: the HBM latency of the state overlaps the staging, the
  // loads of the load phase then hit in L2.  The values are not used.
  float touch = 0.f;
  if (MODE & MODE_PHYS) {
    const int N = L.N; const size_t e0 = (size_t)bid * GO2_WG_ENVS;
#define GO2_TOUCH(base, nf, t0) if (tid >= (t0) && tid < (t0) + (nf)) touch = (base)[(size_t)(tid - (t0)) * N + e0]
    GO2_TOUCH(p.root, 13, 0); GO2_TOUCH(p.dof, 24, 13); GO2_TOUCH(p.kp_mul, 12, 37); GO2_TOUCH(p.kd_mul, 12, 49); GO2_TOUCH(p.zero_off, 12, 61);
    GO2_TOUCH(p.strength, 12, 73); GO2_TOUCH(p.last_actions, 12, 85); GO2_TOUCH(p.foot_impulse, 12, 97); GO2_TOUCH(p.mass_ratio, 18, 109);
    GO2_TOUCH(p.added_com, 3, 127); GO2_TOUCH(p.added_mass, 1, 130); GO2_TOUCH(p.friction, 1, 131); GO2_TOUCH(p.restitution, 1, 132);
#undef GO2_TOUCH
    if (actions_in && tid >= 133 && tid < 133 + 3 && e0 * 12 + (size_t)(tid - 133) * 64 < (size_t)N * 12) touch = actions_in[e0 * 12 + (size_t)(tid - 133) * 64];      // [N,12] row-major: the workgroup's 16 rows = 768 B
  }
#endif
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(GO2_GENERIC(const Go2Tables*, p.tables)); uint32_t* dst = reinterpret_cast<uint32_t*>(&sh.tab);
    for (int i = tid; i < (int)(sizeof(Go2Tables) / 4); i += GO2_WG_THREADS) dst[i] = src[i];
    if (tid < GO2_STEP_SCALAR_PARTS) go2_step_scalars_part(tid, L, blk->dyn, GO2_GENERIC(const float*, p.inj_storage), blk->dyn.common_step_counter + ((MODE & MODE_POST) ? 1 : 0), initial_reset, &sh.S);
  }
  xl::sync();
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" :: "v"(touch));
#endif
  const Go2Tables& tab = sh.tab; const Go2Step& S = sh.S;
  const bool yaw_seen = (MODE & MODE_POST) && S.stage_pending && !blk->dyn.cb_any;      // (rare: Go2Step.stage_pending)
  const int e = bid * GO2_WG_ENVS + (tid >> 4), lane = (tid >> 2) & 3, sub = tid & 3;
  if (e >= L.N) return;   // whole rows (environments) leave together
#if defined(__HIP_DEVICE_COMPILE__) && defined(GO2_KBENCH_STAMPS)
  long long* dbg = p.dbg_clock ? (long long*)p.dbg_clock + (size_t)(bid * 4 + (tid >> 6)) * 32 : nullptr;   // optional phase timestamps per wave (tools/kbench.py)
#define STAMP(k) do { if (dbg && (tid & 63) == 0) dbg[k] = wall_clock64(); } while (0)
#else
#define STAMP(k) do { } while (0)
#endif
  STAMP(0);
  LegPhys ph_; LegPost po_; LaneAux ax;
  ph_.rl = (GO2_AS3 Go2RowsLds*)&sh.rows; ph_.tid = tid; ph_.lid = tid >> 2; ph_.v0_n = v3(0, 0, 1);
  _Pragma("unroll") for (int k = 0; k < GO2_NTYPE; ++k) { ph_.has_v[k] = false; ph_.act_v[k] = 0.f; ph_.body_v[k] = -1; }
  const LegTab& t = tab.leg[lane];
  GO2_AS3 float (*uc)[4] = (GO2_AS3 float (*)[4])sh.ucache[tid >> 4];
  GO2_AS3 float* hc = (GO2_AS3 float*)sh.hcache[tid >> 4];
  if (!S.injected) {
    // the per-step uniforms, one Philox group per lane: observation noise (groups 26..40, go2sim_rng.h) in lanes 0..14, the action delay
    // (group 0) in lane 15
    const int r = tid & 15, g = r < 15 ? 26 + r : 0;
    uint32_t w[4];
    philox4x32_10((uint32_t)(L.env_offset + e), (uint32_t)g, S.step_lo, S.step_hi, L.seed_lo, L.seed_hi, w);
    uc[g][0] = u01_from_bits(w[0]); uc[g][1] = u01_from_bits(w[1]); uc[g][2] = u01_from_bits(w[2]); uc[g][3] = u01_from_bits(w[3]);
  }
  xl::row_sync();
  if (MODE & MODE_RESET_ALL) {
    if (initial_reset == 2 && !p.reset_mask[e]) return;      // go2sim_reset_idx: only the listed environments (whole rows leave)
    lane_reset_all(ph_, po_, ax, tab, p, L, S, uc, hc, e, lane, sub);
    float red[GO2_POST_PARTIALS];
#pragma unroll
    for (int i = 0; i < GO2_POST_PARTIALS; ++i) red[i] = 0.f;
    po_.reset = 1; po_.time_out = 0;
    po_.blv = v3(0, 0, 0); po_.bav = v3(0, 0, 0); po_.pg = v3(0, 0, -1); po_.rpy[0] = po_.rpy[1] = po_.rpy[2] = 0.f;
    po_.own_f2b = 0.f; po_.own_fvel2 = 0.f;
    (void)xl::leg_sum(0.f);    // rendezvous of the row: every lane has read the replicated fields before lane 0 rewrites them
    po_.postB(t, red, 0.f);
    return;
  }
  if (MODE & MODE_PHYS) {
    lane_load_phys(ph_, po_, ax, tab, p, L, S, actions_in, uc[0][0], e, lane, sub);
    LegLoop tl; tl.load(t, sub);
    HotL hl; hl.load(L);
    STAMP(1);
    for (int sb = 0; sb < L.decimation; ++sb) {
      const bool old = L.rand_delay && sb < ax.start;
      const float a[3] = {old ? ax.act_old[0] : ax.act_new[0], old ? ax.act_old[1] : ax.act_new[1], old ? ax.act_old[2] : ax.act_new[2]};
      GO2_MARK(10);
      if (sb == 1) STAMP(16);
      ph_.pd(tl, hl, a, ax.kp, ax.kd, ax.q0, ax.zoff, ax.strength, p.last_dof_vel, L.N, e);
      float part[GO2_QUAD_PARTIALS];
      ph_.phaseA(tl, hl, part);
      GO2_MARK(11);
      if (sb == 1) STAMP(17);
#pragma unroll
      for (int i = 0; i < GO2_QUAD_PARTIALS; ++i) part[i] = xl::leg_sum(part[i]);
      GO2_MARK(12);
      if (sb == 1) STAMP(18);
      ph_.phaseB(hl, part);
      GO2_MARK(13);
      if (sb == 1) STAMP(19);
      ph_.phaseC(tl, t, hl, p.hf_cells, p.hf_top);
      GO2_MARK(14);
      if (sb == 1) STAMP(20);
      // wave-wide row-group activity (ballots -> scalar branches): a group is swept only if some environment of the wave has it active on
      // some leg — typically only the foot contacts are live (an inactive row moves nothing, so skipping is exact)
      const bool af = xl::any(ph_.has_foot()), al = xl::any(ph_.has_limit());
      const bool av[GO2_NTYPE] = {ph_.has_v[0] && xl::any(ph_.has_virtual<0>()), ph_.has_v[1] && xl::any(ph_.has_virtual<1>()),
                                  ph_.has_v[2] && xl::any(ph_.has_virtual<2>()), ph_.has_v[3] && xl::any(ph_.has_virtual<3>())};
      ph_.solve_prepare(af, av, al);
      for (int it = 0; it < L.solver_iterations; ++it) ph_.solve_iteration(af, av, al);
      GO2_MARK(15);
      if (sb == 1) STAMP(21);
      ph_.gather_solution();
      ph_.phaseD(t, hl);
      GO2_MARK(16);
      if (sb == 1) STAMP(22);
    }
    STAMP(2);
    GO2_MARK(20);
    float fb[9];
    {
      int tid1 = tid;          // (indices recomputed behind the loop instead of kept across it: see the note at post-physics below)
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("" : "+v"(tid1));
#endif
      const int e = bid * GO2_WG_ENVS + (tid1 >> 4), lane = (tid1 >> 2) & 3, sub = tid1 & 3;
      lane_finish_phys(ph_, po_, ax, tab, p, L, e, lane, sub, fb);
#pragma unroll
      for (int i = 0; i < 9; ++i) fb[i] = xl::leg_sum(fb[i]);
      lane_store_base_forces(ph_, po_, ax, p, L, e, lane, sub, fb);
    }
  } else {
    lane_load_physout(ph_, po_, ax, tab, p, L, e, lane);
  }
  STAMP(3);
  if (MODE & MODE_POST) {
    GO2_MARK(21);
    // the lane's indices and LDS rows again, from a copy of the thread id the compiler cannot trace back: what post-physics reads of them was live across the
    // substep loop, whose register file is full — the compiler kept them in 48 B of scratch per lane (3 MB written and read back per launch: 1.42 x the
    // algorithmic HBM bytes in the PMC profile); recomputing them costs six integer instructions
    int tid2 = tid;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(tid2));
#endif
    const int e = bid * GO2_WG_ENVS + (tid2 >> 4), lane = (tid2 >> 2) & 3, sub = tid2 & 3;
    const LegTab& t = tab.leg[lane];
    GO2_AS3 float (*uc)[4] = (GO2_AS3 float (*)[4])sh.ucache[tid2 >> 4];
    GO2_AS3 float* hc = (GO2_AS3 float*)sh.hcache[tid2 >> 4];
    lane_init_post(ph_, po_, ax, (const GO2_AS3 uint8_t*)tab.slot_code, uc, hc, &p, &L, &S, e, lane, sub);
    po_.yaw_seen = yaw_seen; po_.out = outs;
#if defined(__HIP_DEVICE_COMPILE__) && defined(GO2_KBENCH_STAMPS)
    po_.dbg = dbg;
#endif
    float part[GO2_POST_PARTIALS];
    po_.postA(t, part);
    GO2_MARK(22);
    STAMP(4);
#pragma unroll
    for (int i = 0; i < GO2_POST_PARTIALS; ++i) part[i] = xl::leg_sum(part[i]);
    float fr = xl::leg_sum(po_.regulation(part));
    po_.postB(t, part, fr);
    STAMP(5);
  }
#undef STAMP
}

#ifndef GO2_EMU
template <int MODE>
__global__ void __launch_bounds__(GO2_WG_THREADS) go2_step_kernel(const Go2DevBlock* __restrict__ blk, const float* __restrict__ actions_in, int initial_reset, const Go2StepOutputs outs) {
  __shared__ Go2Shared sh;
  go2_step_body<MODE>(sh, blk, actions_in, initial_reset, outs, blockIdx.x, threadIdx.x);
}
// ---- test hooks (declared in include/go2sim.h under "test hooks"; no product path calls them) --------------------------------
// legged_robot.py:67-81 on the kernel's own pd() / delay select with caller-supplied DOF states per substep ("fake physics"): what the
// oracle's go2o_torque_trace does, so the reference's golden torques can be compared with the HIP arithmetic directly.
__global__ void __launch_bounds__(64) go2_torque_trace_kernel(const Go2DevBlock* __restrict__ blk, const float* __restrict__ actions_raw, const float* __restrict__ dof, float* __restrict__ out) {
  __shared__ Go2Tables tab; __shared__ Go2Step S;
  const Go2PtrsK& p = *(const Go2PtrsK*)&blk->p; const Go2Launch& L = blk->L;
  { const uint32_t* src = reinterpret_cast<const uint32_t*>(p.tables); uint32_t* dst = reinterpret_cast<uint32_t*>(&tab);
    for (int i = threadIdx.x; i < (int)(sizeof(Go2Tables) / 4); i += 64) dst[i] = src[i];
    if (threadIdx.x == 0) go2_step_scalars(L, blk->dyn, p.inj_storage, blk->dyn.common_step_counter, 0, &S); }
  __syncthreads();
  const int e = blockIdx.x * 16 + (threadIdx.x >> 2), lane = threadIdx.x & 3, N = L.N;
  if (e >= N) return;
  LegPhys ph_; LegPost po_; LaneAux ax;
  const float delay_u = philox_u01((uint32_t)(L.env_offset + e), 0u, S.step_lo, S.step_hi, L.seed_lo, L.seed_hi, GO2_U_DELAY & 3);
    lane_load_phys(ph_, po_, ax, tab, p, L, S, actions_raw, delay_u, e, lane, 0);
  const LegTab& t = tab.leg[lane];
  for (int sub = 0; sub < L.decimation; ++sub) {
    const bool old = L.rand_delay && sub < ax.start;
    const float a[3] = {old ? ax.act_old[0] : ax.act_new[0], old ? ax.act_old[1] : ax.act_new[1], old ? ax.act_old[2] : ax.act_new[2]};
    for (int j = 0; j < 3; ++j) { const float* d = dof + (((size_t)sub * N + e) * 12 + 3 * lane + j) * 2; ph_.q[j] = d[0]; ph_.qd[j] = d[1]; }
    LegLoop tl; tl.load(t, 0);
    ph_.pd(tl, L, a, ax.kp, ax.kd, ax.q0, ax.zoff, ax.strength, p.last_dof_vel, N, e);
    for (int j = 0; j < 3; ++j) { out[((size_t)sub * N + e) * 12 + 3 * lane + j] = ph_.tau[j]; F2D(p.torques, 3 * lane + j, e) = ph_.tau[j]; }
  }
}
// the contact query of the lane program (go2_lane.h contact_query) over caller-supplied spheres: pts [n][4] = centre x, y, z, radius ->
// out [n][4] = gap, normal
__global__ void go2_contact_query_kernel(const Go2DevBlock* __restrict__ blk, const float* __restrict__ pts, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Go2PtrsK& p = *(const Go2PtrsK*)&blk->p;
  LegPhys ph; float gap; V3 nn;
  ph.contact_query(blk->L, p.hf_cells, v3(pts[4 * i], pts[4 * i + 1], pts[4 * i + 2]), pts[4 * i + 3], &gap, &nn);
  out[4 * i] = gap; out[4 * i + 1] = nn.x; out[4 * i + 2] = nn.y; out[4 * i + 3] = nn.z;
}
// HBM-counter calibration probe: the step kernel's access pattern with a KNOWN byte count — 256-thread workgroups of 16 envs, every
// lane of an env reads the same 4 bytes of each of `nread` field-major fields [f][N] (16-byte runs per wave, 64-byte runs per workgroup)
// and lane 0 of the env writes `nwrite` fields.  rocprofv3's FETCH_SIZE / WRITE_SIZE over this kernel against nread*N*4 / nwrite*N*4 give
// the correction factors for THIS pattern (MI355X_MICROARCH.md: "calibrate on a known byte count in your own access pattern").
__global__ void __launch_bounds__(GO2_WG_THREADS) go2_traffic_probe_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int nread, int nwrite) {
  const int e = blockIdx.x * GO2_WG_ENVS + (threadIdx.x >> 4);
  if (e >= N) return;
  float acc = 0.f;
  for (int f = 0; f < nread; ++f) acc += in[(size_t)f * N + e];
  if ((threadIdx.x & 15) == 0) for (int f = 0; f < nwrite; ++f) out[(size_t)f * N + e] = acc + (float)f;
}
// the individually rounded operations of go2_math.h over arrays: out[0..4][n] = a*b, a+b, a-b, a/b, sqrt(|a|)
__global__ void go2_strict_ops_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int n, double inv_b0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = go2_mul_rn(a[i], b[i]); out[n + i] = go2_add_rn(a[i], b[i]); out[2 * (size_t)n + i] = go2_sub_rn(a[i], b[i]);
  out[3 * (size_t)n + i] = go2_div_rn(a[i], b[i]); out[4 * (size_t)n + i] = go2_sqrt_rn(fabsf(a[i])); out[5 * (size_t)n + i] = go2_mul_inv_rn(a[i], inv_b0);
}

// heading_command only, in FRONT of a pass with post-physics: will _post_physics_step_callback's _resample_commands (:408-410) be called with
// >= 1 env in this pass?  (It decides whether the heading clip already sees a command_range_curriculum stage that has just started.)
__global__ void __launch_bounds__(256) go2_cb_scan_kernel(Go2DevBlock* blk) {
  __shared__ int any;
  if (threadIdx.x == 0) any = 0;
  __syncthreads();
  const Go2Launch& L = blk->L;
  const int seen = blk->dyn.cmd_stage_seen < 0 ? -1 : blk->dyn.cmd_stage_seen;
  if (go2_cmd_stage(L, (float)((blk->dyn.common_step_counter + 1) / L.num_steps_per_env)) != seen) {      // uniform
    bool mine = false;
    for (int i = threadIdx.x; i < L.N; i += 256) mine = mine || (blk->p.cmd_timer[i] - 1.f <= 0.f && (float)(blk->p.ep_len[i] + 1) < L.max_episode_length - 1.f);
    if (mine) any = 1;
  }
  __syncthreads();
  if (threadIdx.x == 0) blk->dyn.cb_any = any;
}
// after a step: extras["episode"] means, then advance the device-resident counters
__global__ void go2_finish_kernel(Go2DevBlock* blk, int counter_inc, float* info_out) {
  float* accum = blk->p.ep_accum; float* info = blk->p.episode_info;
  int i = threadIdx.x;
  float cnt = accum[GO2_NUM_REWARDS], track = accum[GO2_REW_TRACKING_LIN_VEL], cb = accum[GO2_NUM_REWARDS + 1];
  __syncthreads();
  if (i <= GO2_NUM_REWARDS) {
    if (cnt > 0.f) info[i] = i < GO2_NUM_REWARDS ? accum[i] / cnt / blk->L.episode_length_s : cnt;
    accum[i] = 0.f;
  }
  if (cnt > 0.f) {      // terrain_level_all / terrain_level_<kind> (legged_robot.py:231-237): written by reset_idx only, i.e. in passes that reset an env
    // per-thread sums in registers (levels are small integers: exact in fp32, order-free), wave shuffles, then one LDS slot per wave
    __shared__ float lsum[GO2_NUM_TERRAIN_KINDS + 1], lcnt[GO2_NUM_TERRAIN_KINDS + 1], wsum[4][GO2_NUM_TERRAIN_KINDS + 1], wcnt[4][GO2_NUM_TERRAIN_KINDS + 1];
    float ps[GO2_NUM_TERRAIN_KINDS + 1], pc[GO2_NUM_TERRAIN_KINDS + 1];
#pragma unroll
    for (int k = 0; k <= GO2_NUM_TERRAIN_KINDS; ++k) { ps[k] = 0.f; pc[k] = 0.f; }
    if (blk->L.terrain_mode != 0) {
      const int N = blk->L.N;
      for (int e0 = i; e0 < N; e0 += 8 * (int)blockDim.x) {      // 8 environments per thread and trip: all 16 loads in flight before the first use
        long long lv8[8]; int kd8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int e = e0 + u * (int)blockDim.x; lv8[u] = e < N ? blk->p.terrain_levels[e] : 0; kd8[u] = e < N ? blk->p.terrain_kind[e] : -2; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float lv = (float)lv8[u]; const int kd = kd8[u];
          ps[0] += kd != -2 ? lv : 0.f; pc[0] += kd != -2 ? 1.f : 0.f;
#pragma unroll
          for (int k = 0; k < GO2_NUM_TERRAIN_KINDS; ++k) { const bool in = kd == k; ps[1 + k] += in ? lv : 0.f; pc[1 + k] += in ? 1.f : 0.f; }
        }
      }
    }
#pragma unroll
    for (int k = 0; k <= GO2_NUM_TERRAIN_KINDS; ++k) {
      for (int o = 32; o > 0; o >>= 1) { ps[k] += __shfl_down(ps[k], o); pc[k] += __shfl_down(pc[k], o); }
      if ((i & 63) == 0 && (i >> 6) < 4) { wsum[i >> 6][k] = ps[k]; wcnt[i >> 6][k] = pc[k]; }
    }
    __syncthreads();
    if (i <= GO2_NUM_TERRAIN_KINDS) { lsum[i] = (wsum[0][i] + wsum[1][i]) + (wsum[2][i] + wsum[3][i]); lcnt[i] = (wcnt[0][i] + wcnt[1][i]) + (wcnt[2][i] + wcnt[3][i]); }
    __syncthreads();
    if (i <= GO2_NUM_TERRAIN_KINDS)
      info[GO2_NUM_REWARDS + 3 + i] = blk->L.terrain_mode == 0 ? (i == 0 ? 0.f : __uint_as_float(0x7fc00000u)) : (lcnt[i] > 0.f ? lsum[i] / lcnt[i] : __uint_as_float(0x7fc00000u));
  }
  if (i == 0) {
    accum[GO2_NUM_REWARDS + 1] = 0.f;
    go2_track_cmd_curriculum(blk->L, blk->dyn, cnt, track, cb, blk->dyn.common_step_counter + counter_inc, info + GO2_NUM_REWARDS + 1);
    blk->dyn.common_step_counter += counter_inc; blk->dyn.step_count += 1; blk->dyn.use_injected = 0;
  }
  if (info_out) {     // extras['episode'] ring slot (go2sim_step_rollout): this step's vector
    __syncthreads();
    if (i < GO2_EPISODE_INFO_LEN) info_out[i] = info[i];
  }
}
__global__ void go2_peek_kernel(float* out, const Go2Tables* tab, int N, int env_offset, uint32_t s0, uint32_t s1, uint32_t k0, uint32_t k1) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * GO2_NUM_UNIFORMS) return;
  int e = i / GO2_NUM_UNIFORMS, slot = i % GO2_NUM_UNIFORMS, code = tab->slot_code[slot];
  out[i] = philox_u01((uint32_t)(env_offset + e), (uint32_t)(code >> 2), s0, s1, k0, k1, code & 3);
}
// GAE(lambda) reverse scan, one lane per env (rollout_storage.py:123-137); block partials -> fp64 atomics
__global__ void __launch_bounds__(256) go2_gae_kernel(const float* rew, const uint8_t* dones, const float* val, const float* last, float* ret, float* adv,
                                                      double* partials, int T, int N, float gamma, float lam) {
  int e = blockIdx.x * 256 + threadIdx.x;
  double s1 = 0, s2 = 0;
  if (e < N) {
    float a = 0.f;
    for (int t = T - 1; t >= 0; --t) {
      float nv = t == T - 1 ? last[e] : val[(size_t)(t + 1) * N + e];
      float nt = 1.f - (dones[(size_t)t * N + e] ? 1.f : 0.f);
      float v = val[(size_t)t * N + e];
      float delta = rew[(size_t)t * N + e] + nt * gamma * nv - v;
      a = delta + nt * gamma * lam * a;
      float r = a + v; ret[(size_t)t * N + e] = r;
      float ad = r - v; adv[(size_t)t * N + e] = ad; s1 += ad; s2 += (double)ad * ad;
    }
  }
  __shared__ double sh1[4], sh2[4];
  for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_down(s1, off); s2 += __shfl_down(s2, off); }
  if ((threadIdx.x & 63) == 0) { sh1[threadIdx.x >> 6] = s1; sh2[threadIdx.x >> 6] = s2; }
  __syncthreads();
  if (threadIdx.x == 0 && partials) {
    atomicAdd(&partials[0], sh1[0] + sh1[1] + sh1[2] + sh1[3]); atomicAdd(&partials[1], sh2[0] + sh2[1] + sh2[2] + sh2[3]);
    if (blockIdx.x == 0) atomicAdd(&partials[2], (double)T * N);
  }
}
__global__ void go2_normalize_kernel(float* adv, const double* partials, int count) {
  double n = partials[2], mean = partials[0] / n, var = (partials[1] - n * mean * mean) / (n - 1.0);
  float sd = (float)sqrt(var > 0 ? var : 0.0), m = (float)mean;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) adv[i] = (adv[i] - m) / (sd + 1e-8f);
}

// ---- clip_grad_norm_ + (adaptive-KL learning rate) + Adam over a list of tensors (go2sim.h go2sim_adam_clip_step) -----------------
// Block b works on one GO2_ADAM_CHUNK-element chunk of one tensor (chunks of tensor i: first[i] .. first[i+1]).
struct Go2AdamLaunch { Go2AdamTensors t; int32_t first[GO2_ADAM_MAX_TENSORS + 1]; };
__device__ __forceinline__ int adam_locate(const Go2AdamLaunch& a, int b) { int i = 0; while (i + 1 < a.t.count && b >= a.first[i + 1]) ++i; return i; }
// A chunk is 1024 quads: a thread owns quads tid, tid + 256, ... and has all of its 16-byte loads in flight before the first use (these launches
// sit on the critical path of every mini-batch, between the backward and the next forward pass: latency, not bandwidth, is what they cost).
// Tensors whose four arrays are not 16-byte aligned, and a chunk's last n % 4 elements, go element by element.
__device__ __forceinline__ bool adam_vec(const Go2AdamLaunch& a, int i) {
  return ((((uintptr_t)a.t.param[i]) | ((uintptr_t)a.t.grad[i]) | ((uintptr_t)a.t.exp_avg[i]) | ((uintptr_t)a.t.exp_avg_sq[i])) & 15) == 0;
}
__device__ __forceinline__ float adam_block_sum(float s, float* sh) {
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
// stage 1: per-chunk sum of squared gradients -> ws[block]; block 0 also takes the learning-rate decision (ppo.py:140-155) and advances the
// tensors' step counters (stage 2 reads the advanced value: t = step, as torch.optim.Adam increments before it forms the bias corrections)
__global__ void __launch_bounds__(256) go2_adam_norm_kernel(const Go2AdamLaunch a, float* __restrict__ ws, float* lr, const float* kl_mean, float desired_kl) {
  __shared__ float sh[4];
  const int i = adam_locate(a, blockIdx.x), off = (blockIdx.x - a.first[i]) * GO2_ADAM_CHUNK, n = min(GO2_ADAM_CHUNK, a.t.numel[i] - off);
  const float* __restrict__ g = a.t.grad[i] + off;
  float s = 0.f;
  if (adam_vec(a, i) && n >= 4) {          // (a chunk of fewer than 4 elements — the critic's 1-element output bias — has no quad to load: element by element)
    const int nq = n >> 2;
    float4 x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) x[u] = reinterpret_cast<const float4*>(g)[min((int)threadIdx.x + 256 * u, max(nq - 1, 0))];
#pragma unroll
    for (int u = 0; u < 4; ++u) if ((int)threadIdx.x + 256 * u < nq) s += (x[u].x * x[u].x + x[u].y * x[u].y) + (x[u].z * x[u].z + x[u].w * x[u].w);
    if ((int)threadIdx.x < (n & 3)) { const float y = g[4 * nq + threadIdx.x]; s += y * y; }
  } else {
    for (int k = threadIdx.x; k < n; k += 256) { const float y = g[k]; s += y * y; }
  }
  s = adam_block_sum(s, sh);
  if (threadIdx.x == 0) {
    ws[blockIdx.x] = s;
    if (blockIdx.x == 0 && kl_mean) {
      const float kl = *kl_mean, r = *lr;
      *lr = kl > desired_kl * 2.f ? fmaxf(1e-5f, r / 1.5f) : ((kl < desired_kl / 2.f && kl > 0.f) ? fminf(1e-2f, r * 1.5f) : r);
    }
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < a.t.count) a.t.step[threadIdx.x][0] += 1.f;
}
__device__ __forceinline__ void adam_one(float& p, float& m, float& v, float g, float coef, float omb1, float beta2, float omb2, float step_size, float bc2s, float eps) {
  const float gk = g * coef;
  m = m + (gk - m) * omb1; v = beta2 * v + omb2 * gk * gk;
  p -= step_size * m / (sqrtf(v) / bc2s + eps);
}
// stage 2: every block re-reduces the chunk sums in the same fixed order (-> the same norm everywhere), then steps its chunk
__global__ void __launch_bounds__(256) go2_adam_step_kernel(const Go2AdamLaunch a, const float* __restrict__ ws, const float* __restrict__ lr, float max_norm,
                                                          double beta1d, double beta2d, float eps) {
  const float beta2 = (float)beta2d, omb1 = (float)(1.0 - beta1d), omb2 = (float)(1.0 - beta2d);
  __shared__ float sh[4];
  const int nb = a.first[a.t.count];
  const int i = adam_locate(a, blockIdx.x), off = (blockIdx.x - a.first[i]) * GO2_ADAM_CHUNK, n = min(GO2_ADAM_CHUNK, a.t.numel[i] - off);
  float* __restrict__ p = a.t.param[i] + off; float* __restrict__ m = a.t.exp_avg[i] + off; float* __restrict__ v = a.t.exp_avg_sq[i] + off; const float* __restrict__ g = a.t.grad[i] + off;
  const bool vec = adam_vec(a, i) && n >= 4;
  const int nq = n >> 2;
  float4 g4[4], m4[4], v4[4], p4[4];
  if (vec) {        // the chunk's loads go out before the norm is re-reduced: they do not depend on it
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = min((int)threadIdx.x + 256 * u, max(nq - 1, 0));
      g4[u] = reinterpret_cast<const float4*>(g)[q]; m4[u] = reinterpret_cast<const float4*>(m)[q]; v4[u] = reinterpret_cast<const float4*>(v)[q]; p4[u] = reinterpret_cast<const float4*>(p)[q];
    }
  }
  float s = 0.f;
  for (int k = threadIdx.x; k < nb; k += 256) s += ws[k];
  const float norm = sqrtf(adam_block_sum(s, sh));
  const float coef = fminf(max_norm / (norm + 1e-6f), 1.f);
  const float t = a.t.step[i][0];                       // this tensor's own step count, as torch.optim.Adam keeps it (advanced by stage 1)
  // the bias corrections in fp64, as torch forms them: 1 - 0.999^t cancels to ~1e-3 in the first steps
  const float bc1 = (float)(1.0 - pow(beta1d, (double)t)), bc2s = (float)sqrt(1.0 - pow(beta2d, (double)t)), step_size = lr[0] / bc1;
  if (vec) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = threadIdx.x + 256 * u;
      if (q < nq) {
        adam_one(p4[u].x, m4[u].x, v4[u].x, g4[u].x, coef, omb1, beta2, omb2, step_size, bc2s, eps); adam_one(p4[u].y, m4[u].y, v4[u].y, g4[u].y, coef, omb1, beta2, omb2, step_size, bc2s, eps);
        adam_one(p4[u].z, m4[u].z, v4[u].z, g4[u].z, coef, omb1, beta2, omb2, step_size, bc2s, eps); adam_one(p4[u].w, m4[u].w, v4[u].w, g4[u].w, coef, omb1, beta2, omb2, step_size, bc2s, eps);
        reinterpret_cast<float4*>(m)[q] = m4[u]; reinterpret_cast<float4*>(v)[q] = v4[u]; reinterpret_cast<float4*>(p)[q] = p4[u];
      }
    }
    const int k = 4 * nq + threadIdx.x;
    if ((int)threadIdx.x < (n & 3)) adam_one(p[k], m[k], v[k], g[k], coef, omb1, beta2, omb2, step_size, bc2s, eps);
  } else {
    for (int k = threadIdx.x; k < n; k += 256) adam_one(p[k], m[k], v[k], g[k], coef, omb1, beta2, omb2, step_size, bc2s, eps);
  }
}

// ---- fused PPO loss head (ppo.py:131-170).  One 16-lane group per sample row, one lane per action dimension (A <= 16): the per-action terms of
// the log-probability, the entropy and the KL divergence are formed side by side and summed over the row by an xor tree, instead of a 12-long
// dependent chain per thread — this launch sits between the forward and the backward pass of every mini-batch, its latency is what it costs.
// A block of 256 threads takes PPO_ROWS rows (4 per group, all their loads issued before the first use). ----------------------------------------
#define PPO_NSTAT 24   // per-block partials: [0..3] surrogate, value loss, kl, entropy ; [4..4+A) grad_std
#define PPO_ROWS 64
__device__ __forceinline__ float row16_sum(float x) {
  x += __shfl_xor(x, 8, 16); x += __shfl_xor(x, 4, 16); x += __shfl_xor(x, 2, 16); x += __shfl_xor(x, 1, 16);
  return x;
}
__global__ void __launch_bounds__(256) go2_ppo_loss_kernel(const float* __restrict__ mu, const float* __restrict__ std_, const float* __restrict__ value,
    const float* __restrict__ actions, const float* __restrict__ old_mu, const float* __restrict__ old_sigma, const float* __restrict__ old_logp,
    const float* __restrict__ adv, const float* __restrict__ tv, const float* __restrict__ ret, float* __restrict__ gmu, float* __restrict__ gval,
    float* __restrict__ part, int B, int A, float clip, float vcoef, int use_clip_v, int split, float w_head, float w_tail) {
  __shared__ float sh[16][PPO_NSTAT];
  constexpr int U = PPO_ROWS / 16;
  const int j = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const bool on = j < A;
  const int jc = on ? j : 0;
  const float LOG2PI = 1.8378770664093453f;
  const float sg = std_[jc], ls = logf(sg), isg2 = 1.f / (sg * sg);
  float a_[U], m_[U], so_[U], mo_[U], olp[U], ad[U], vv[U], tvv[U], rt[U];
  int row[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    row[u] = blockIdx.x * PPO_ROWS + grp * U + u;
    const int r = min(row[u], B - 1); const size_t k = (size_t)r * A + jc;
    a_[u] = actions[k]; m_[u] = mu[k]; so_[u] = old_sigma[k]; mo_[u] = old_mu[k];
    olp[u] = old_logp[r]; ad[u] = adv[r]; vv[u] = value[r]; tvv[u] = tv[r]; rt[u] = ret[r];
  }
  float s_sur = 0.f, s_vl = 0.f, s_kl = 0.f, s_ent = 0.f, s_gs = 0.f;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const float d = a_[u] - m_[u], dm = mo_[u] - m_[u];
    const float lp = row16_sum(on ? -d * d * (0.5f * isg2) - ls - 0.5f * LOG2PI : 0.f);
    const float ent = row16_sum(on ? 0.5f + 0.5f * LOG2PI + ls : 0.f);
    const float kl = row16_sum(on ? logf(sg / so_[u] + 1e-5f) + (so_[u] * so_[u] + dm * dm) * (0.5f * isg2) - 0.5f : 0.f);
    const int i = row[u];
    const float ratio = expf(lp - olp[u]), a = ad[u];
    const float lo = 1.f - clip, hi = 1.f + clip, rc = fminf(fmaxf(ratio, lo), hi), in = (ratio >= lo && ratio <= hi) ? 1.f : 0.f;
    const float s1 = -a * ratio, s2 = -a * rc, sur = fmaxf(s1, s2);
    const float w = s1 > s2 ? 1.f : (s1 < s2 ? in : 0.5f + 0.5f * in);     // torch.max splits ties evenly; clamp passes gradient inside [lo, hi]
    const float wr = i < split ? w_head : w_tail;     // plain PPO: 1/B for every row; CTS: 1/teacher rows, 1/student rows
    const float g_lp = -a * w * ratio * wr;
    const float v = vv[u], dv = v - tvv[u];
    float vl, gv;
    if (use_clip_v) {
      const float dc = fminf(fmaxf(dv, -clip), clip), vin = (dv >= -clip && dv <= clip) ? 1.f : 0.f, vc = tvv[u] + dc;
      const float l1 = (v - rt[u]) * (v - rt[u]), l2 = (vc - rt[u]) * (vc - rt[u]); vl = fmaxf(l1, l2);
      const float g1 = 2.f * (v - rt[u]), g2 = 2.f * (vc - rt[u]) * vin; gv = l1 > l2 ? g1 : (l1 < l2 ? g2 : 0.5f * g1 + 0.5f * g2);
    } else { vl = (rt[u] - v) * (rt[u] - v); gv = 2.f * (v - rt[u]); }
    if (i < B) {
      if (j == 0) gval[i] = vcoef * gv / (float)B;
      if (on) { gmu[(size_t)i * A + j] = g_lp * d * isg2; s_gs += g_lp * (d * d * isg2 / sg - 1.f / sg); }
      s_sur += sur * wr; s_vl += vl; s_kl += kl; s_ent += ent;
    }
  }
  // the block's 16 groups, added in a fixed order
  if (j == 0) { sh[grp][0] = s_sur; sh[grp][1] = s_vl; sh[grp][2] = s_kl; sh[grp][3] = s_ent; }
  if (j < PPO_NSTAT - 4) sh[grp][4 + j] = on ? s_gs : 0.f;
  __syncthreads();
  if (threadIdx.x < PPO_NSTAT) {
    float t[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) t[g] = sh[g][threadIdx.x];
#pragma unroll
    for (int wd = 8; wd >= 1; wd >>= 1)
#pragma unroll
      for (int g = 0; g < wd; ++g) t[g] += t[g + wd];
    part[(size_t)blockIdx.x * PPO_NSTAT + threadIdx.x] = t[0];
  }
}
__global__ void go2_ppo_loss_finish_kernel(const float* __restrict__ part, const float* __restrict__ std_, float* __restrict__ gstd, float* __restrict__ stats,
                                           int nblocks, int B, int A, float vcoef, float ecoef) {
  // 16 x PPO_NSTAT threads: thread (j, k) sums the blocks j, j + 16, ... of statistic k, four loads in flight at a time (this launch sits between the
  // loss kernel and the backward pass: a serial chain of `nblocks` dependent loads per statistic is all it would be), then the 16 sub-sums are added in a
  // fixed order: deterministic
  __shared__ float sh[16][PPO_NSTAT];
  const int k = threadIdx.x % PPO_NSTAT, j = threadIdx.x / PPO_NSTAT;
  if (j < 16) {
    float s = 0.f;
    for (int b0 = j; b0 < nblocks; b0 += 64) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = part[(size_t)min(b0 + 16 * u, nblocks - 1) * PPO_NSTAT + k];
#pragma unroll
      for (int u = 0; u < 4; ++u) if (b0 + 16 * u < nblocks) s += v[u];
    }
    sh[j][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < PPO_NSTAT) {
    float t[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) t[g] = sh[g][k];
#pragma unroll
    for (int wd = 8; wd >= 1; wd >>= 1)
#pragma unroll
      for (int g = 0; g < wd; ++g) t[g] += t[g + wd];
    const float s = t[0];
    if (k < 4) { sh[0][k] = k == 0 ? s : s / (float)B; stats[k] = sh[0][k]; }      // the surrogate partials are already weighted
    else if (k - 4 < A) gstd[k - 4] = s - ecoef / std_[k - 4];
  }
  __syncthreads();
  if (threadIdx.x == 0) stats[4] = sh[0][0] + vcoef * sh[0][1] - ecoef * sh[0][3];
}
// ---- ELU backward + bias gradient in one pass.  Block = 64 column-quads (float4) x 4 row lanes over a 256-col x EB_ROWS-row tile; HBM-bound
// (reads gy, y, writes gz: 12 B per element), so each thread keeps 4 rows = 8 float4 loads in flight.  Column partials per row tile go to a
// workspace and are summed in a fixed order by a second, tree-shaped kernel (deterministic; no atomics) --------------------------------
#define EB_ROWS 64
__device__ __forceinline__ float4 elu_bwd4(const float4 g, const float4 v) {
  float4 o;
  o.x = g.x * (v.x > 0.f ? 1.f : v.x + 1.f); o.y = g.y * (v.y > 0.f ? 1.f : v.y + 1.f);
  o.z = g.z * (v.z > 0.f ? 1.f : v.z + 1.f); o.w = g.w * (v.w > 0.f ? 1.f : v.w + 1.f);
  return o;
}
__global__ void __launch_bounds__(256) go2_elu_bwd_bias_kernel(const float* __restrict__ gy, const float* __restrict__ y, float* __restrict__ gz, float* __restrict__ part, int B, int C) {
  __shared__ float4 sh[4][64];
  const int cq = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c0 = (blockIdx.x * 64 + cq) * 4, r0 = blockIdx.y * EB_ROWS;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c0 < C) {
    const int r1 = min(r0 + EB_ROWS, B);
    int r = r0 + rl;
    for (; r + 12 < r1; r += 16) {          // 4 rows of this row lane per trip, all 8 loads issued before the first use
      float4 g[4], v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const size_t k = (size_t)(r + 4 * u) * C + c0; g[u] = *reinterpret_cast<const float4*>(gy + k); v[u] = *reinterpret_cast<const float4*>(y + k); }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 o = elu_bwd4(g[u], v[u]);
        *reinterpret_cast<float4*>(gz + (size_t)(r + 4 * u) * C + c0) = o;
        acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
      }
    }
    for (; r < r1; r += 4) {
      const size_t k = (size_t)r * C + c0;
      const float4 o = elu_bwd4(*reinterpret_cast<const float4*>(gy + k), *reinterpret_cast<const float4*>(y + k));
      *reinterpret_cast<float4*>(gz + k) = o;
      acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
    }
  }
  sh[rl][cq] = acc;
  __syncthreads();
  if (rl == 0 && c0 < C) {
    const float4 a = sh[0][cq], b = sh[1][cq], c = sh[2][cq], d = sh[3][cq];
    float4 o; o.x = (a.x + b.x) + (c.x + d.x); o.y = (a.y + b.y) + (c.y + d.y); o.z = (a.z + b.z) + (c.z + d.z); o.w = (a.w + b.w) + (c.w + d.w);
    *reinterpret_cast<float4*>(part + (size_t)blockIdx.y * C + c0) = o;
  }
}
// column sums of part[nrows][C]: block = 16 columns x 16 row groups; each thread adds its rows (stride 16) in order, then a fixed LDS tree
__global__ void __launch_bounds__(256) go2_colsum_finish_kernel(const float* __restrict__ part, float* __restrict__ gb, int nrows, int C) {
  __shared__ float sh[16][17];
  const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
  float s = 0.f;
  if (c < C) for (int r = rg; r < nrows; r += 16) s += part[(size_t)r * C + c];
  sh[rg][cl] = s;
  __syncthreads();
  if (rg == 0 && c < C) {
    float t[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = sh[i][cl];
#pragma unroll
    for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
      for (int i = 0; i < w; ++i) t[i] += t[i + w];
    gb[c] = t[0];
  }
}

// ---- rollout heads: PPO.act sampling + storage rows, PPO.process_env_step (one thread per env) ----------------------------
__global__ void __launch_bounds__(256) go2_act_head_kernel(const float* __restrict__ mu, const float* __restrict__ std_, const float* __restrict__ eps, const float* __restrict__ value,
    float* __restrict__ a_out, float* __restrict__ a_st, float* __restrict__ mu_st, float* __restrict__ sig_st, float* __restrict__ lp_st, float* __restrict__ v_st, int N, int A) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= N) return;
  const float HALF_LOG2PI = 0.9189385332046727f;
  float lp = 0.f;
  for (int j = 0; j < A; ++j) {
    const size_t k = (size_t)e * A + j;
    const float m = mu[k], sg = std_[j], a = go2_add_rn(m, go2_mul_rn(sg, eps[k])), d = a - m;      // mean + std * eps as torch's two ops (no FMA): bit-equal to the eager formulation
    lp += -(d * d) / (2.f * sg * sg) - logf(sg) - HALF_LOG2PI;
    a_out[k] = a;
    if (a_st) a_st[k] = a;
    if (mu_st) mu_st[k] = m;
    if (sig_st) sig_st[k] = sg;
  }
  if (lp_st) lp_st[e] = lp;
  if (v_st) v_st[e] = value[e];
}
// A <= 16: 16 lanes per row (coalesced rows, 16x the waves of one thread per row); the log-probability is summed in the same order (j = 0, 1, ..)
__global__ void __launch_bounds__(256) go2_act_head16_kernel(const float* __restrict__ mu, const float* __restrict__ std_, const float* __restrict__ eps, const float* __restrict__ value,
    float* __restrict__ a_out, float* __restrict__ a_st, float* __restrict__ mu_st, float* __restrict__ sig_st, float* __restrict__ lp_st, float* __restrict__ v_st, int N, int A) {
  const int t = blockIdx.x * 256 + threadIdx.x, e = t >> 4, j = t & 15;
  const bool on = e < N && j < A;
  const float HALF_LOG2PI = 0.9189385332046727f;
  float term = 0.f;
  if (on) {
    const size_t k = (size_t)e * A + j;
    const float m = mu[k], sg = std_[j], a = go2_add_rn(m, go2_mul_rn(sg, eps[k])), d = a - m;
    term = -(d * d) / (2.f * sg * sg) - logf(sg) - HALF_LOG2PI;
    a_out[k] = a;
    if (a_st) a_st[k] = a;
    if (mu_st) mu_st[k] = m;
    if (sig_st) sig_st[k] = sg;
  }
  float lp = 0.f;
  const int base = (threadIdx.x & 63) & ~15;
  for (int q = 0; q < A; ++q) lp += __shfl(term, base + q);
  if (e < N && j == 0) { if (lp_st) lp_st[e] = lp; if (v_st) v_st[e] = value[e]; }
}
__global__ void __launch_bounds__(256) go2_store_transition_kernel(const float* __restrict__ rew, const uint8_t* __restrict__ dones, const uint8_t* __restrict__ touts,
    const float* __restrict__ v_st, float* __restrict__ rew_st, uint8_t* __restrict__ dones_st, float gamma, int N) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= N) return;
  float r = rew[e];
  if (touts) r += gamma * (v_st[e] * (touts[e] ? 1.f : 0.f));
  rew_st[e] = r; dones_st[e] = dones[e];
}

// ---- CTS observation-history ring (on_policy_runner_cts.py:155-156): one thread per (env, feature), H values in flight ------
// ---- the update's permutation + gathers (go2sim_shuffle_gather) --------------------------------------------------------------------------
// One wave per output row at a time, its lanes run along the row of every job (263-, 45-, 12- and 1-float rows
// of the PPO storage: 13 load / store pairs per row); two rows' loads are in flight per wave.  The source row pi(r) is computed by the wave itself (no sort,
// no index tensor) unless the caller hands in a permutation.  HBM-bound: 2 x 348 floats per row.
#define GO2_GATHER_ROWS_PER_WG 32
struct Go2GatherArgs {
  const float* src[GO2_GATHER_MAX_JOBS]; float* dst[GO2_GATHER_MAX_JOBS]; int32_t w[GO2_GATHER_MAX_JOBS], dp[GO2_GATHER_MAX_JOBS];          // dp: destination row pitch
  int32_t njobs, rows, h, nclear; const int64_t* indices; uint32_t* key; float* clear;
};
__global__ void __launch_bounds__(256) go2_shuffle_gather_kernel(const Go2GatherArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t seed = 0, counter = 0;
  if (!a.indices) { seed = a.key[0]; counter = a.key[1]; }          // (read before this workgroup's ticket below: the counter only moves once EVERY workgroup holds a ticket)
  const int r0 = blockIdx.x * GO2_GATHER_ROWS_PER_WG;
  for (int rr = wave; rr < GO2_GATHER_ROWS_PER_WG; rr += 8) {       // two rows per trip
    int r[2], sidx[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      r[u] = r0 + rr + 4 * u;
      const int rc = min(r[u], a.rows - 1);
      sidx[u] = a.indices ? (int)a.indices[rc] : (int)go2_shuffle_index((uint32_t)rc, (uint32_t)a.rows, a.h, seed, counter);
    }
    for (int j = 0; j < a.njobs; ++j) {
      const int w = a.w[j];
      const float* __restrict__ s0 = a.src[j] + (size_t)sidx[0] * w; const float* __restrict__ s1 = a.src[j] + (size_t)sidx[1] * w;
      float* __restrict__ d0 = a.dst[j] + (size_t)r[0] * a.dp[j]; float* __restrict__ d1 = a.dst[j] + (size_t)r[1] * a.dp[j];
      for (int e = lane; e < w; e += 64) {
        const float v0 = s0[e], v1 = s1[e];
        if (r[0] < a.rows) d0[e] = v0;
        if (r[1] < a.rows) d1[e] = v1;
      }
    }
  }
  if (blockIdx.x == 0) for (int i = threadIdx.x; i < a.nclear; i += 256) a.clear[i] = 0.f;
  if (!a.indices) {          // the last workgroup to finish advances the counter for the next launch (a replayed HIP graph): every workgroup has read it by then
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned t = atomicAdd(&a.key[2], 1u);
      if (t == gridDim.x - 1) { a.key[2] = 0u; __threadfence(); atomicAdd(&a.key[1], 1u); }
    }
  }
}

__global__ void __launch_bounds__(256) go2_cts_indices_kernel(int64_t* __restrict__ out, int nmb, int nt, int ns, int tb, int sb, int ht, int hs, const int64_t* __restrict__ map, uint32_t* key) {
  const uint32_t seed = key[0], counter = key[1];          // (read before this workgroup's ticket below)
  const int r = blockIdx.x * 256 + threadIdx.x, mb = tb + sb;
  if (r < nmb * mb) {
    const int i = r / mb, j = r - i * mb;
    const int64_t k = j < tb ? (int64_t)go2_shuffle_index((uint32_t)(i * tb + j), (uint32_t)nt, ht, seed, counter)
                             : (int64_t)nt + (int64_t)go2_shuffle_index((uint32_t)(i * sb + j - tb), (uint32_t)ns, hs, seed ^ GO2_SHUFFLE_TAIL_SEED, counter);
    out[r] = map ? map[k] : k;
  }
  __syncthreads();
  if (threadIdx.x == 0) {          // the last workgroup to finish advances the counter (go2_shuffle_gather_kernel's ticket)
    const unsigned t = atomicAdd(&key[2], 1u);
    if (t == gridDim.x - 1) { key[2] = 0u; __threadfence(); atomicAdd(&key[1], 1u); }
  }
}

__global__ void __launch_bounds__(256) go2_history_push_kernel(float* __restrict__ hist, const float* __restrict__ obs, const uint8_t* __restrict__ dones, int N, int H, int D) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N * D) return;
  const int e = i / D, d = i - e * D;
  const bool z = dones != nullptr && dones[e] != 0;
  float* h = hist + (size_t)e * H * D + d;
  for (int k = 0; k + 1 < H; ++k) h[(size_t)k * D] = z ? 0.f : h[(size_t)(k + 1) * D];
  h[(size_t)(H - 1) * D] = obs[i];
}
#endif  // !GO2_EMU

// ------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------
struct Go2Sim {
  Go2SimCfg cfg; Go2SimBuffers b; int N;
  Go2DevBlock h;            // host mirror of the device-resident block (pointers, constants, counters)
  Go2DevBlock* d_blk;       // the block in device memory
  std::vector<void*> allocs;
  float dt, max_episode_length;
  float* inj_storage; Go2Tables* d_tables; int16_t* d_hf; Go2Cell* d_cells; float* d_torigins;
  int timing; double time_ms; int64_t time_launches;
#ifndef GO2_EMU
  std::vector<hipEvent_t> ev; size_t ev_used;
#endif
};

static void* dev_alloc(Go2Sim* s, size_t bytes) {
  void* ptr = nullptr;
#ifdef GO2_EMU
  ptr = calloc(1, bytes ? bytes : 1);
#else
  if (hipMalloc(&ptr, bytes ? bytes : 1) != hipSuccess) return nullptr;
  hipMemset(ptr, 0, bytes ? bytes : 1);
#endif
  if (ptr) s->allocs.push_back(ptr);
  return ptr;
}
static void dev_upload(void* dst, const void* src, size_t bytes) {
#ifdef GO2_EMU
  memcpy(dst, src, bytes);
#else
  hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
#endif
}

static float u01_host(uint64_t seed, uint32_t env, uint32_t slot, uint64_t step) {
  uint32_t r[4]; philox4x32_10(env, slot >> 2, (uint32_t)step, (uint32_t)(step >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), r);
  return u01_from_bits(r[slot & 3]);
}
static float urange_h(float u, float lo, float hi) { return (hi - lo) * u + lo; }

static void fill_tables(Go2Tables* T) {
  static const int kBodyLink[19] = GO2_BODY_LINK_INIT; static const double kOff[19][3] = GO2_BODY_OFFSET_INIT; static const double kMass[19] = GO2_BODY_MASS_INIT;
  static const double kCom[19][3] = GO2_BODY_COM_INIT; static const double kIn[19][6] = GO2_BODY_INERTIA_INIT; static const double kJO[12][3] = GO2_JOINT_ORIGIN_INIT;
  static const double kLo[12] = GO2_JOINT_LOWER_INIT, kHi[12] = GO2_JOINT_UPPER_INIT, kEff[12] = GO2_JOINT_EFFORT_INIT, kVel[12] = GO2_JOINT_VELOCITY_INIT;
  struct Sph { int body, link; double c[3]; double r; };
  static const Sph kFoot[4] = GO2_FOOT_PTS_INIT; static const Sph kOther[4][GO2_LEG_OTHER_PTS] = GO2_LEG_OTHER_PTS_INIT; static const Sph kBase[GO2_BASE_PTS] = GO2_BASE_PTS_INIT;
  (void)kBodyLink;
  memset(T, 0, sizeof(*T));
  T->layout_ok = 1;
  auto body10 = [&](int b, float* out) {   // about the moving-link origin
    V3 c = v3((float)(kOff[b][0] + kCom[b][0]), (float)(kOff[b][1] + kCom[b][1]), (float)(kOff[b][2] + kCom[b][2]));
    S3 Ic = {(float)kIn[b][0], (float)kIn[b][1], (float)kIn[b][2], (float)kIn[b][3], (float)kIn[b][4], (float)kIn[b][5]};
    RB r = rb_from_com((float)kMass[b], c, Ic);
    out[0] = r.m; out[1] = r.h.x; out[2] = r.h.y; out[3] = r.h.z; out[4] = r.J.xx; out[5] = r.J.yy; out[6] = r.J.zz; out[7] = r.J.xy; out[8] = r.J.xz; out[9] = r.J.yz;
  };
  for (int l = 0; l < 4; ++l) {
    LegTab& t = T->leg[l];
    for (int k = 0; k < 3; ++k) { t.o1[k] = (float)kJO[3 * l][k]; t.o2[k] = (float)kJO[3 * l + 1][k]; t.o3[k] = (float)kJO[3 * l + 2][k]; }
    for (int k = 0; k < 4; ++k) { int b = 3 + 4 * l + k; body10(b, t.body[k]); t.body_index[k] = b; t.mass_ratio_index[k] = b - 1; }
    for (int j = 0; j < 3; ++j) { t.lim_lo[j] = (float)kLo[3 * l + j]; t.lim_hi[j] = (float)kHi[3 * l + j]; t.vel_lim[j] = (float)kVel[3 * l + j]; t.eff_lim[j] = (float)kEff[3 * l + j]; }
    for (int k = 0; k < 3; ++k) { t.foot_pt[k] = (float)kFoot[l].c[k]; t.foot_off[k] = (float)kOff[3 + 4 * l + 3][k]; }
    t.foot_pt[3] = (float)kFoot[l].r;
    for (int i = 0; i < GO2_LEG_OTHER_PTS; ++i) {
      for (int k = 0; k < 3; ++k) t.other_pt[i][k] = (float)kOther[l][i].c[k];
      t.other_pt[i][3] = (float)kOther[l][i].r; t.other_link[i] = kOther[l][i].link - 3 * l; t.other_body[i] = kOther[l][i].body;
      const int want = i < GO2_N_HIP_PTS ? 1 : (i < GO2_N_HIP_PTS + GO2_N_THIGH_PTS ? 2 : 3);
      if (t.other_link[i] != want) T->layout_ok = 0;      // go2_lane.h phaseC relies on the per-link order of the candidates
    }
    t.n_base = 0;
    for (int i = 0; i < GO2_BASE_PTS; ++i) if ((i & 3) == l) {
      int k = t.n_base++; for (int a = 0; a < 3; ++a) t.base_pt[k][a] = (float)kBase[i].c[a]; t.base_pt[k][3] = (float)kBase[i].r; t.base_body[k] = kBase[i].body;
    }
    // per-group axis-aligned reach of the spheres in their link frame (go2_lane.h: conservative cull of whole groups on the plane)
    for (int g = 0; g < 4; ++g) for (int a = 0; a < 4; ++a) t.cull_ext[g][a] = 0.f;
    for (int i = 0; i < GO2_LEG_OTHER_PTS; ++i) { const int g = t.other_link[i] - 1; for (int a = 0; a < 3; ++a) t.cull_ext[g][a] = fmaxf(t.cull_ext[g][a], fabsf(t.other_pt[i][a]) + t.other_pt[i][3]); }
    for (int k = 0; k < t.n_base; ++k) for (int a = 0; a < 3; ++a) t.cull_ext[3][a] = fmaxf(t.cull_ext[3][a], fabsf(t.base_pt[k][a]) + t.base_pt[k][3]);
    // deal the candidates to the 4 sub-lanes: slots {thigh x 3, calf x 2, hip (sub-lanes 2, 3), base} (go2_tables.h SubCand)
    for (int sb = 0; sb < 4; ++sb) {
      SubCand& sc = t.cand[sb];
      const int T0 = GO2_N_HIP_PTS, C0 = GO2_N_HIP_PTS + GO2_N_THIGH_PTS;
      const int leg_idx[6] = {T0 + sb, T0 + 4 + sb, T0 + 8 + sb, C0 + sb, C0 + 4 + sb, sb >= 2 ? sb - 2 : -1};
      for (int k = 0; k < 6; ++k) {
        const int i = leg_idx[k];
        if (i < 0) { for (int a = 0; a < 4; ++a) sc.pt[k][a] = 0.f; sc.idx[k] = -1; sc.body[k] = 0; continue; }
        for (int a = 0; a < 4; ++a) sc.pt[k][a] = t.other_pt[i][a];
        sc.idx[k] = i; sc.body[k] = t.other_body[i];
        const int want = k < 3 ? 2 : (k < 5 ? 3 : 1);
        if (t.other_link[i] != want) T->layout_ok = 0;
      }
      if (sb < t.n_base) { for (int a = 0; a < 4; ++a) sc.pt[GO2_SC_BASE][a] = t.base_pt[sb][a]; sc.idx[GO2_SC_BASE] = GO2_NLEG_OTHER + sb; sc.body[GO2_SC_BASE] = t.base_body[sb]; }
      else { for (int a = 0; a < 4; ++a) sc.pt[GO2_SC_BASE][a] = 0.f; sc.idx[GO2_SC_BASE] = -1; sc.body[GO2_SC_BASE] = 0; }
    }
  }
  T->base.m0 = (float)kMass[0];
  for (int k = 0; k < 3; ++k) T->base.c0[k] = (float)kCom[0][k];
  for (int k = 0; k < 6; ++k) T->base.Ic0[k] = (float)kIn[0][k];
  body10(1, T->base.head[0]); body10(2, T->base.head[1]);
  for (int b = 0; b < 3; ++b) for (int k = 0; k < 3; ++k) T->base.body_off[b][k] = (float)kOff[b][k];
  go2_fill_slot_codes(T->slot_code);
  for (int sl = 0; sl < GO2_NUM_UNIFORMS; ++sl) if ((int)T->slot_code[sl] != go2_slot_code(sl)) T->layout_ok = 0;      // go2_post.h's constant-expression form of the same mapping
}



static void blk_sync_dyn(Go2Sim* s, void* stream) {   // host mirror -> device counters (rare: create, set_counter, inject)
#ifdef GO2_EMU
  // only the counters the host owns: the tracked command list behind them (cmd_stage_seen, cmd_x_range) lives on the device alone
  (void)stream; memcpy(&s->d_blk->dyn, &s->h.dyn, offsetof(Go2Dyn, cmd_stage_seen));
#else
  hipMemcpyAsync(&s->d_blk->dyn, &s->h.dyn, offsetof(Go2Dyn, cmd_stage_seen), hipMemcpyHostToDevice, (hipStream_t)stream);
#endif
}

extern "C" {

int go2sim_is_device_library(void) {
#ifdef GO2_EMU
  return 0;
#else
  return 1;
#endif
}
int go2sim_buffer_layout(void) { return 1; }
const char* go2sim_last_error(void) { return g_err; }
void go2sim_default_cfg(Go2SimCfg* cfg) { go2sim_fill_default_cfg(cfg); }

void go2sim_destroy(Go2Sim* s) {
  if (!s) return;
#ifndef GO2_EMU
  for (hipEvent_t e : s->ev) hipEventDestroy(e);
#endif
  for (void* ptr : s->allocs) {
#ifdef GO2_EMU
    free(ptr);
#else
    hipFree(ptr);
#endif
  }
  delete s;
}

// terrain_types = torch.div(arange(N), N / num_cols, rounding_mode="floor") (legged_robot.py:1072): an int64 tensor over a Python
// float computes in fp32, and torch's floor-division is fmod-based — so 3 / (12/20) is 4, not 5.  Restated exactly.
static int64_t terrain_type_of(uint32_t ge, int Ng, int num_types) {
  float a = (float)ge, b = (float)((double)Ng / (double)num_types);
  float mod = fmodf(a, b), div = (a - mod) / b;
  if (mod != 0.f && ((b < 0.f) != (mod < 0.f))) div -= 1.f;
  float fl = 0.f;
  if (div != 0.f) { fl = floorf(div); if (div - fl > 0.5f) fl += 1.f; }
  return (int64_t)fl;
}
int go2sim_create(const Go2SimCfg* cfg, int device_id, Go2Sim** out) {
  if (!cfg || !out) FAIL(GO2SIM_EINVAL, "null argument");
  if (cfg->struct_size != sizeof(Go2SimCfg) || cfg->abi_version != GO2SIM_ABI_VERSION) FAIL(GO2SIM_EINVAL, "cfg size/version mismatch (%u vs %zu)", cfg->struct_size, sizeof(Go2SimCfg));
  if (cfg->num_envs <= 0 || cfg->decimation <= 0 || cfg->num_envs_global < cfg->env_offset + cfg->num_envs) FAIL(GO2SIM_EINVAL, "bad num_envs/decimation");
  if (cfg->control_type < 0 || cfg->control_type > 2) FAIL(GO2SIM_EINVAL, "control_type must be 0 (P), 1 (V) or 2 (T)");
  if (cfg->terrain_mode != 0 && (!cfg->hf_samples || !cfg->terrain_origins || !cfg->terrain_type_id)) FAIL(GO2SIM_EINVAL, "heightfield terrain needs hf_samples, terrain_origins, terrain_type_id");
  if (cfg->limit_vel_comb_count > 36 || cfg->cmd_curriculum_count > 4 || cfg->reward_curriculum_count > 4) FAIL(GO2SIM_EINVAL, "table sizes out of range");
#ifndef GO2_EMU
  HIPCHK(hipSetDevice(device_id));
#else
  (void)device_id;
#endif
  Go2Sim* s = new Go2Sim();
  s->cfg = *cfg; const int N = s->N = cfg->num_envs;
  s->d_hf = nullptr; s->d_cells = nullptr; s->d_torigins = nullptr; s->timing = 0; s->time_ms = 0; s->time_launches = 0;
#ifndef GO2_EMU
  s->ev_used = 0;
#endif
  memset(&s->b, 0, sizeof(s->b)); memset(&s->h, 0, sizeof(s->h));
  bool ok = true;
#define A(field, type, count) do { s->b.field = (type*)dev_alloc(s, sizeof(type) * (size_t)(count)); ok = ok && s->b.field; } while (0)
  A(root_states, float, N * 13); A(dof_state, float, N * 24); A(contact_forces, float, N * 57); A(rigid_body_states, float, N * 19 * 13);
  A(obs_buf, float, N * GO2_NUM_OBS); A(privileged_obs_buf, float, N * GO2_NUM_PRIV_OBS); A(rew_buf, float, N); A(reset_buf, uint8_t, N); A(time_out_buf, uint8_t, N);
  A(episode_length_buf, int64_t, N); A(torques, float, N * 12); A(actions, float, N * 12); A(last_actions, float, N * 12); A(last_last_actions, float, N * 12);
  A(last_dof_vel, float, N * 12); A(last_root_vel, float, N * 6); A(commands, float, N * 4); A(commands_resampling_step, float, N); A(commands_xy_accumulation, float, N * 2);
  A(stop_heading, uint8_t, N); A(last_is_limit_vel, uint8_t, N); A(base_lin_vel, float, N * 3); A(base_ang_vel, float, N * 3); A(projected_gravity, float, N * 3); A(rpy, float, N * 3);
  A(measured_heights, float, N * GO2_NUM_HEIGHT_POINTS); A(max_move_distance, float, N); A(turn_over_timer, float, N); A(feet_air_time, float, N * 4); A(last_contacts, uint8_t, N * 4); A(last_contacts2, uint8_t, N * 4);
  A(motor_strengths, float, N * 12); A(motor_zero_offsets, float, N * 12); A(p_gains_multiplier, float, N * 12); A(d_gains_multiplier, float, N * 12);
  A(env_origins, float, N * 3); A(terrain_levels, int64_t, N); A(terrain_types, int64_t, N); A(episode_sums, float, GO2_NUM_REWARDS * N);
  A(friction_coeffs, float, N); A(restitution_coeffs, float, N); A(added_base_mass, float, N); A(added_base_com, float, N * 3); A(link_mass_ratio, float, N * 18);
  A(episode_info, float, GO2_EPISODE_INFO_LEN); A(foot_impulse, float, N * 12);
#undef A
  Go2Ptrs& p = s->h.p; Go2SimBuffers& b = s->b;
  p.root = b.root_states; p.dof = b.dof_state; p.contact = b.contact_forces; p.rigid = b.rigid_body_states; p.obs = b.obs_buf; p.priv = b.privileged_obs_buf; p.rew = b.rew_buf;
  p.reset = b.reset_buf; p.time_out = b.time_out_buf; p.ep_len = b.episode_length_buf; p.torques = b.torques; p.actions = b.actions; p.last_actions = b.last_actions;
  p.last_last_actions = b.last_last_actions; p.last_dof_vel = b.last_dof_vel; p.last_root_vel = b.last_root_vel; p.commands = b.commands; p.cmd_timer = b.commands_resampling_step;
  p.cmd_xy_acc = b.commands_xy_accumulation; p.stop_heading = b.stop_heading; p.last_is_limit_vel = b.last_is_limit_vel; p.base_lin_vel = b.base_lin_vel; p.base_ang_vel = b.base_ang_vel;
  p.proj_gravity = b.projected_gravity; p.rpy = b.rpy; p.heights = b.measured_heights; p.max_move = b.max_move_distance; p.to_timer = b.turn_over_timer; p.feet_air_time = b.feet_air_time; p.last_contacts = b.last_contacts;
  p.last_contacts2 = b.last_contacts2; p.strength = b.motor_strengths; p.zero_off = b.motor_zero_offsets; p.kp_mul = b.p_gains_multiplier; p.kd_mul = b.d_gains_multiplier; p.origins = b.env_origins;
  p.terrain_levels = b.terrain_levels; p.terrain_types = b.terrain_types; p.ep_sums = b.episode_sums; p.friction = b.friction_coeffs; p.restitution = b.restitution_coeffs;
  p.added_mass = b.added_base_mass; p.added_com = b.added_base_com; p.mass_ratio = b.link_mass_ratio; p.episode_info = b.episode_info; p.foot_impulse = b.foot_impulse;
  p.terrain_kind = (int32_t*)dev_alloc(s, sizeof(int32_t) * N); p.reset_mask = (uint8_t*)dev_alloc(s, (size_t)N); p.ep_accum = (float*)dev_alloc(s, sizeof(float) * (GO2_NUM_REWARDS + 2));
  s->inj_storage = (float*)dev_alloc(s, sizeof(float) * (size_t)N * GO2_NUM_UNIFORMS); p.inj_storage = s->inj_storage;
  s->d_tables = (Go2Tables*)dev_alloc(s, sizeof(Go2Tables));
  s->d_blk = (Go2DevBlock*)dev_alloc(s, sizeof(Go2DevBlock));
  ok = ok && p.terrain_kind && p.reset_mask && p.ep_accum && s->inj_storage && s->d_tables && s->d_blk;
  if (!ok) { go2sim_destroy(s); FAIL(GO2SIM_ENOMEM, "allocation failed"); }
  { Go2Tables T; fill_tables(&T); if (!T.layout_ok) FAIL(GO2SIM_EINVAL, "table layout check failed: collision candidates not tabulated hip/thigh/calf (include/go2_model_data.h vs go2_tables.h), or go2_slot_code != go2sim_rng.h");
    dev_upload(s->d_tables, &T, sizeof(T)); p.tables = s->d_tables; }

  s->dt = (float)cfg->decimation * cfg->sim_dt;
  s->max_episode_length = (float)ceil((double)cfg->episode_length_s / (double)s->dt - 1e-3);   // np.ceil(25/0.02) = 1250 (:1104)

  // ---- launch block constants ----
  Go2Launch& L = s->h.L;
  L.N = N; L.env_offset = cfg->env_offset; L.decimation = cfg->decimation; L.solver_iterations = cfg->solver_iterations;
  L.seed_lo = (uint32_t)cfg->seed; L.seed_hi = (uint32_t)(cfg->seed >> 32);
  L.sim_dt = cfg->sim_dt; L.dt = s->dt; memcpy(L.gravity, cfg->gravity, sizeof(L.gravity));
  L.contact_offset = cfg->contact_offset; L.erp = cfg->erp; L.max_depen_vel = cfg->max_depenetration_velocity; L.bounce_thr = cfg->bounce_threshold_velocity;
  L.cfm = cfg->contact_cfm; L.armature = cfg->joint_armature; L.limit_margin = cfg->joint_limit_margin;
  L.max_lin_vel = cfg->max_linear_velocity; L.max_ang_vel = cfg->max_angular_velocity;
  L.terrain_mode = cfg->terrain_mode; L.hf_walls = (cfg->terrain_mode != 0 && cfg->hf_cells && cfg->hf_walls) ? 1 : 0; L.hf_rows = cfg->hf_rows; L.hf_cols = cfg->hf_cols; L.hf_hscale = cfg->hf_hscale; L.hf_inv_hscale = cfg->hf_hscale != 0.f ? 1.0 / (double)cfg->hf_hscale : 0.0; L.hf_vscale = cfg->hf_vscale; L.hf_border = cfg->hf_border;
  L.terrain_friction = cfg->terrain_friction; L.terrain_restitution = cfg->terrain_restitution; L.terrain_num_levels = cfg->terrain_num_levels; L.terrain_num_types = cfg->terrain_num_types;
  L.terrain_curriculum = cfg->terrain_curriculum; L.move_down_by_acc = cfg->move_down_by_accumulated_xy_command; L.measure_heights = cfg->measure_heights; L.full_body_states = cfg->full_body_states; L.terrain_length = cfg->terrain_length;
  memcpy(L.kp, cfg->kp, sizeof(L.kp)); memcpy(L.kd, cfg->kd, sizeof(L.kd)); memcpy(L.q0, cfg->default_dof_pos, sizeof(L.q0));
  L.control_type = cfg->control_type; L.cmd_track_curr = cfg->cmd_tracking_curriculum; L.cmd_max_curr = cfg->cmd_max_curriculum;
  L.action_scale = cfg->action_scale; L.clip_actions = cfg->clip_actions; L.clip_obs = cfg->clip_observations; memcpy(L.base_init, cfg->base_init_state, sizeof(L.base_init));
  L.rand_strength = cfg->randomize_motor_strength; L.rand_offset = cfg->randomize_motor_zero_offset; L.rand_pd = cfg->randomize_pd_gains; L.push_robots = cfg->push_robots;
  L.push_interval = cfg->push_interval; L.rand_delay = cfg->randomize_action_delay;
  memcpy(L.strength_rng, cfg->motor_strength_range, 8); memcpy(L.offset_rng, cfg->motor_zero_offset_range, 8); memcpy(L.kp_rng, cfg->stiffness_mult_range, 8); memcpy(L.kd_rng, cfg->damping_mult_range, 8);
  L.push_xy = cfg->max_push_vel_xy; L.push_ang = cfg->max_push_ang_vel;
  L.resampling_time = cfg->cmd_resampling_time; L.heading_command = cfg->heading_command; L.dynamic_resample = cfg->dynamic_resample_commands; L.limit_vel_prob = cfg->limit_vel_prob;
  L.limit_invert = cfg->limit_vel_invert_when_continuous; L.stop_heading_at_limit = cfg->stop_heading_at_limit; L.limit_ang_zero_prob = cfg->limit_ang_vel_at_zero_command_prob;
  L.comb_count = cfg->limit_vel_comb_count; memcpy(L.comb, cfg->limit_vel_comb, sizeof(L.comb)); memcpy(L.terrain_max_cmd, cfg->terrain_max_cmd_ranges, sizeof(L.terrain_max_cmd));
  memcpy(L.cmd_ranges0, cfg->cmd_ranges, sizeof(L.cmd_ranges0));
  L.rew_mask_all = 0u;
  for (int t = 0; t < GO2_NUM_REWARDS; ++t) {   // :914-930
    L.rew_scale_dt[t] = cfg->reward_scales[t] * s->dt; L.rew_to_scale_dt[t] = cfg->turn_over ? cfg->turn_over_scales[t] * s->dt : 0.f;
    L.rew_on[t] = (L.rew_scale_dt[t] != 0.f || L.rew_to_scale_dt[t] != 0.f) ? 1 : 0;
    if (L.rew_on[t]) L.rew_mask_all |= 1u << t;
  }
  L.turn_over = cfg->turn_over; L.to_roll_thr = cfg->turn_over_roll_threshold;
  for (int k = 0; k < 3; ++k) L.to_prop[k] = cfg->turn_over_proportions[k];
  for (int k = 0; k < 2; ++k) { L.to_zero_time[k] = cfg->turn_over_zero_time[k]; L.to_height[k][0] = cfg->turn_over_init_heights[k][0]; L.to_height[k][1] = cfg->turn_over_init_heights[k][1]; }
  L.rew_curr_count = cfg->reward_curriculum_count; memcpy(L.rew_curr_term, cfg->reward_curriculum_term, sizeof(L.rew_curr_term)); memcpy(L.rew_curr, cfg->reward_curriculum, sizeof(L.rew_curr));
  L.cmd_curr_count = cfg->cmd_curriculum_count; memcpy(L.cmd_curr, cfg->cmd_curriculum, sizeof(L.cmd_curr));
  L.zero_curr_enabled = cfg->zero_cmd_curriculum_enabled; memcpy(L.zero_curr, cfg->zero_cmd_curriculum, sizeof(L.zero_curr)); L.num_steps_per_env = cfg->num_steps_per_env;
  L.only_positive = cfg->only_positive_rewards; L.tracking_sigma = cfg->tracking_sigma; L.dyn_sigma = cfg->dynamic_sigma_enabled;
  memcpy(L.dyn_sigma_vel, cfg->dynamic_sigma_vel, sizeof(L.dyn_sigma_vel)); memcpy(L.dyn_sigma_max, cfg->dynamic_sigma_max, sizeof(L.dyn_sigma_max));
  L.soft_vel_limit = cfg->soft_dof_vel_limit; L.soft_torque_limit = cfg->soft_torque_limit; L.base_height_target = cfg->base_height_target; L.max_contact_force = cfg->max_contact_force;
  L.min_legs_distance = cfg->min_legs_distance;
  {
    static const double lo[12] = GO2_JOINT_LOWER_INIT, hi[12] = GO2_JOINT_UPPER_INIT;
    for (int j = 0; j < 12; ++j) { float l = (float)lo[j], h = (float)hi[j], m = (l + h) / 2, r = h - l; L.soft_limits[j][0] = m - 0.5f * r * cfg->soft_dof_pos_limit; L.soft_limits[j][1] = m + 0.5f * r * cfg->soft_dof_pos_limit; }   // :372-375
  }
  L.os_lin = cfg->obs_scale_lin_vel; L.os_ang = cfg->obs_scale_ang_vel; L.os_dof_pos = cfg->obs_scale_dof_pos; L.os_dof_vel = cfg->obs_scale_dof_vel; L.os_height = cfg->obs_scale_height;
  L.add_noise = cfg->add_noise;
  { float nl = cfg->noise_level;   // Go2Robot._get_noise_scale_vec (go2_env.py:9-21)
    for (int i = 0; i < 3; ++i) { L.noise_vec[i] = cfg->noise_ang_vel * nl * cfg->obs_scale_ang_vel; L.noise_vec[3 + i] = cfg->noise_gravity * nl; L.noise_vec[6 + i] = 0.f; }
    for (int j = 0; j < 12; ++j) { L.noise_vec[9 + j] = cfg->noise_dof_pos * nl * cfg->obs_scale_dof_pos; L.noise_vec[21 + j] = cfg->noise_dof_vel * nl * cfg->obs_scale_dof_vel; L.noise_vec[33 + j] = 0.f; } }
  L.max_episode_length = s->max_episode_length; L.episode_length_s = cfg->episode_length_s;

  // ---- terrain ----
  std::vector<int32_t> type_id;
  std::vector<float> torig;
  if (cfg->terrain_mode != 0) {
    size_t nh = (size_t)cfg->hf_rows * cfg->hf_cols; s->d_hf = (int16_t*)dev_alloc(s, nh * sizeof(int16_t));
    size_t no = (size_t)cfg->terrain_num_levels * cfg->terrain_num_types * 3; s->d_torigins = (float*)dev_alloc(s, no * sizeof(float));
    if (!s->d_hf || !s->d_torigins) { go2sim_destroy(s); FAIL(GO2SIM_ENOMEM, "allocation failed"); }
    dev_upload(s->d_hf, cfg->hf_samples, nh * sizeof(int16_t)); dev_upload(s->d_torigins, cfg->terrain_origins, no * sizeof(float));
    p.hf = s->d_hf; p.terrain_origins = s->d_torigins;
    {   // the contact surface, cell by cell: the caller's (trimesh: displaced surface with vertical faces) or the continuous one of the samples
      const size_t nr = (size_t)cfg->hf_rows - 1, nc = (size_t)cfg->hf_cols - 1;
      std::vector<Go2Cell> cells(nr * nc);
      for (size_t i = 0; i < nr; ++i) for (size_t j = 0; j < nc; ++j) {
        Go2Cell& q = cells[i * nc + j];
        if (cfg->hf_cells) for (int k = 0; k < 4; ++k) q.h[k] = cfg->hf_cells[(i * nc + j) * 4 + k];
        else { const int16_t* h = cfg->hf_samples; const size_t C_ = (size_t)cfg->hf_cols; q.h[0] = h[i * C_ + j]; q.h[1] = h[(i + 1) * C_ + j]; q.h[2] = h[i * C_ + j + 1]; q.h[3] = h[(i + 1) * C_ + j + 1]; }
      }
      if (L.hf_walls) {        // trimesh: a record per cell that carries the neighbours' heights along its edges and at its corners (go2_tables.h Go2CellW)
        std::vector<Go2CellW> cw(nr * nc);
        for (long i = 0; i < (long)nr; ++i) for (long j = 0; j < (long)nc; ++j) {
          Go2CellW& o = cw[i * nc + j]; const Go2Cell& q = cells[i * nc + j];
          auto nb = [&](long a, long b) -> const Go2Cell* { return (a >= 0 && a < (long)nr && b >= 0 && b < (long)nc) ? &cells[a * nc + b] : nullptr; };
          for (int k = 0; k < 4; ++k) o.h[k] = q.h[k];
          const Go2Cell* c;
          c = nb(i - 1, j); o.xm[0] = c ? c->h[1] : q.h[0]; o.xm[1] = c ? c->h[3] : q.h[2];
          c = nb(i + 1, j); o.xp[0] = c ? c->h[0] : q.h[1]; o.xp[1] = c ? c->h[2] : q.h[3];
          c = nb(i, j - 1); o.ym[0] = c ? c->h[2] : q.h[0]; o.ym[1] = c ? c->h[3] : q.h[1];
          c = nb(i, j + 1); o.yp[0] = c ? c->h[0] : q.h[2]; o.yp[1] = c ? c->h[1] : q.h[3];
          for (int k = 0; k < 4; ++k) { c = nb(i + ((k & 1) ? 1 : -1), j + ((k & 2) ? 1 : -1)); o.dg[k] = c ? c->h[3 - k] : q.h[k]; }
        }
        s->d_cells = (Go2Cell*)dev_alloc(s, cw.size() * sizeof(Go2CellW));
        if (!s->d_cells) { go2sim_destroy(s); FAIL(GO2SIM_ENOMEM, "allocation failed"); }
        dev_upload(s->d_cells, cw.data(), cw.size() * sizeof(Go2CellW)); p.hf_cells = s->d_cells;
      } else {
        s->d_cells = (Go2Cell*)dev_alloc(s, cells.size() * sizeof(Go2Cell));
        if (!s->d_cells) { go2sim_destroy(s); FAIL(GO2SIM_ENOMEM, "allocation failed"); }
        dev_upload(s->d_cells, cells.data(), cells.size() * sizeof(Go2Cell)); p.hf_cells = s->d_cells;
      }
      // the candidate cull's view of the map (go2_lane.h phaseC): per GO2_TOP_CELL^2 block the highest cell corner within GO2_TOP_REACH
      // cells around it (separable running maximum), and the smallest facet n_z of the whole map
      const int B = GO2_TOP_CELL, R = GO2_TOP_REACH; const int tr = (int)((nr + B - 1) / B), tc = (int)((nc + B - 1) / B);
      std::vector<int16_t> cmax(nr * nc), rowmax((size_t)nr * tc), top((size_t)tr * tc);
      double nzmin = 1.0; const double hs = cfg->hf_hscale, vs = cfg->hf_vscale;
      for (size_t i = 0; i < nr * nc; ++i) {
        const int16_t* h = cells[i].h; int16_t m = h[0]; for (int k = 1; k < 4; ++k) m = h[k] > m ? h[k] : m; cmax[i] = m;
        // the cell's two triangles, split along (i,j)-(i+1,j+1): slopes (h10 - h00, h11 - h10) and (h11 - h01, h01 - h00)
        const double s[2][2] = {{(h[1] - h[0]) * vs / hs, (h[3] - h[1]) * vs / hs}, {(h[3] - h[2]) * vs / hs, (h[2] - h[0]) * vs / hs}};
        for (int k = 0; k < 2; ++k) { const double nz = 1.0 / sqrt(s[k][0] * s[k][0] + s[k][1] * s[k][1] + 1.0); nzmin = nz < nzmin ? nz : nzmin; }
      }
      for (size_t i = 0; i < nr; ++i) for (int bj = 0; bj < tc; ++bj) {
        int16_t m = INT16_MIN; const long j0 = (long)bj * B - R, j1 = (long)(bj + 1) * B + R;
        for (long j = j0 < 0 ? 0 : j0; j < j1 && j < (long)nc; ++j) m = cmax[i * nc + j] > m ? cmax[i * nc + j] : m;
        rowmax[i * tc + bj] = m;
      }
      for (int bi = 0; bi < tr; ++bi) for (int bj = 0; bj < tc; ++bj) {
        int16_t m = INT16_MIN; const long i0 = (long)bi * B - R, i1 = (long)(bi + 1) * B + R;
        for (long i = i0 < 0 ? 0 : i0; i < i1 && i < (long)nr; ++i) m = rowmax[i * tc + bj] > m ? rowmax[i * tc + bj] : m;
        top[(size_t)bi * tc + bj] = m;
      }
      int16_t* d_top = (int16_t*)dev_alloc(s, top.size() * sizeof(int16_t));
      if (!d_top) { go2sim_destroy(s); FAIL(GO2SIM_ENOMEM, "allocation failed"); }
      dev_upload(d_top, top.data(), top.size() * sizeof(int16_t)); p.hf_top = d_top;
      L.hf_trows = tr; L.hf_tcols = tc; L.hf_nzmin = (float)(nzmin * (1.0 - 1e-6));
    }
    type_id.assign(cfg->terrain_type_id, cfg->terrain_type_id + cfg->terrain_num_types); torig.assign(cfg->terrain_origins, cfg->terrain_origins + no);
  }
  s->cfg.hf_samples = nullptr; s->cfg.hf_cells = nullptr; s->cfg.terrain_origins = nullptr; s->cfg.terrain_type_id = nullptr;

  // ---- creation-time per-env quantities (legged_robot.py:320-402, :1054-1091), same Philox slots as the oracle ----
  const uint64_t INIT = 0xFFFFFFFFFFFFFFFFull;
  float buckets[64];
  for (int k = 0; k < 64; ++k) buckets[k] = urange_h(u01_host(cfg->seed, 0xFFFFFFFFu, (uint32_t)k, INIT), cfg->friction_range[0], cfg->friction_range[1]);
  std::vector<float> fr(N), re(N), am(N), ac(3 * N), mr(18 * N), org(3 * N), root(13 * N), dof(24 * N), ones(12 * N, 1.f);
  std::vector<int64_t> lv(N, 0), ty(N, 0); std::vector<int32_t> kind(N, -1); std::vector<uint8_t> rb(N, 1);
  const int Ng = cfg->num_envs_global;
  for (int e = 0; e < N; ++e) {
    uint32_t ge = (uint32_t)(cfg->env_offset + e);
    int bucket = (int)(u01_host(cfg->seed, ge, 0, INIT) * 64); bucket = bucket > 63 ? 63 : bucket;
    fr[e] = cfg->randomize_friction ? buckets[bucket] : 1.0f;
    re[e] = cfg->randomize_restitution ? urange_h(u01_host(cfg->seed, ge, 1, INIT), cfg->restitution_range[0], cfg->restitution_range[1]) : 0.f;
    am[e] = cfg->randomize_base_mass ? urange_h(u01_host(cfg->seed, ge, 2, INIT), cfg->added_mass_range[0], cfg->added_mass_range[1]) : 0.f;
    for (int k = 0; k < 3; ++k) ac[(size_t)k * N + e] = cfg->randomize_base_com ? urange_h(u01_host(cfg->seed, ge, 3 + k, INIT), cfg->base_com_range[0], cfg->base_com_range[1]) : 0.f;
    for (int k = 0; k < 18; ++k) mr[(size_t)k * N + e] = cfg->randomize_link_mass ? urange_h(u01_host(cfg->seed, ge, 6 + k, INIT), cfg->link_mass_range[0], cfg->link_mass_range[1]) : 1.f;
    float o3[3];
    if (cfg->terrain_mode == 0) {
      int ncols = (int)floor(sqrt((double)Ng)); o3[0] = cfg->env_spacing * (float)(ge / ncols); o3[1] = cfg->env_spacing * (float)(ge % ncols); o3[2] = 0.f;
    } else {
      int maxl = cfg->terrain_curriculum ? cfg->max_init_terrain_level : cfg->terrain_num_levels - 1;
      lv[e] = ge % (uint32_t)(maxl + 1); ty[e] = terrain_type_of(ge, Ng, cfg->terrain_num_types);
      const float* o = &torig[((size_t)lv[e] * cfg->terrain_num_types + ty[e]) * 3]; o3[0] = o[0]; o3[1] = o[1]; o3[2] = o[2];
      kind[e] = type_id[ty[e]];
    }
    for (int k = 0; k < 3; ++k) org[(size_t)k * N + e] = o3[k];
    for (int k = 0; k < 13; ++k) root[(size_t)k * N + e] = cfg->base_init_state[k] + (k < 3 ? o3[k] : 0.f);
    for (int j = 0; j < 12; ++j) { dof[(size_t)j * N + e] = cfg->default_dof_pos[j]; dof[(size_t)(12 + j) * N + e] = 0.f; }
  }
  dev_upload(b.friction_coeffs, fr.data(), 4 * N); dev_upload(b.restitution_coeffs, re.data(), 4 * N); dev_upload(b.added_base_mass, am.data(), 4 * N);
  dev_upload(b.added_base_com, ac.data(), 12 * N); dev_upload(b.link_mass_ratio, mr.data(), 72 * N); dev_upload(b.env_origins, org.data(), 12 * N);
  dev_upload(b.root_states, root.data(), 52 * N); dev_upload(b.dof_state, dof.data(), 96 * N); dev_upload(b.terrain_levels, lv.data(), 8 * N); dev_upload(b.terrain_types, ty.data(), 8 * N);
  dev_upload(p.terrain_kind, kind.data(), 4 * N); dev_upload(b.reset_buf, rb.data(), N);
  dev_upload(b.motor_strengths, ones.data(), 48 * N); dev_upload(b.p_gains_multiplier, ones.data(), 48 * N); dev_upload(b.d_gains_multiplier, ones.data(), 48 * N);
  s->h.dyn.cmd_stage_seen = -2; s->h.dyn.cmd_x_range[0] = cfg->cmd_ranges[0][0]; s->h.dyn.cmd_x_range[1] = cfg->cmd_ranges[0][1];
  dev_upload(s->d_blk, &s->h, sizeof(Go2DevBlock));
  *out = s;
  return 0;
}

int go2sim_get_buffers(Go2Sim* s, Go2SimBuffers* out) { if (!s || !out) FAIL(GO2SIM_EINVAL, "null argument"); *out = s->b; return 0; }

#ifdef GO2_EMU
struct EmuArgs { Go2Shared* sh; const Go2DevBlock* blk; const float* actions; int initial_reset, bid, mode; const Go2StepOutputs* outs; };
static void emu_thread(void* a_, int tid) {
  EmuArgs& a = *(EmuArgs*)a_;
  if (a.mode == MODE_RESET_ALL) go2_step_body<MODE_RESET_ALL>(*a.sh, a.blk, a.actions, a.initial_reset, *a.outs, a.bid, tid);
  else if (a.mode == (MODE_PHYS | MODE_POST)) go2_step_body<MODE_PHYS | MODE_POST>(*a.sh, a.blk, a.actions, a.initial_reset, *a.outs, a.bid, tid);
  else if (a.mode == MODE_PHYS) go2_step_body<MODE_PHYS>(*a.sh, a.blk, a.actions, a.initial_reset, *a.outs, a.bid, tid);
  else go2_step_body<MODE_POST>(*a.sh, a.blk, a.actions, a.initial_reset, *a.outs, a.bid, tid);
}
static void emu_run(Go2Sim* s, int mode, const float* actions_in, int initial_reset, int counter_inc, const Go2StepOutputs& outs) {
  Go2DevBlock* blk = s->d_blk;
  const Go2Ptrs& p = blk->p; const Go2Launch& L = blk->L;
  static thread_local Go2Shared sh;
  if ((mode & MODE_POST) && L.heading_command) {      // == go2_cb_scan_kernel
    int any = 0; const int seen = blk->dyn.cmd_stage_seen < 0 ? -1 : blk->dyn.cmd_stage_seen;
    if (go2_cmd_stage(L, (float)((blk->dyn.common_step_counter + 1) / L.num_steps_per_env)) != seen)
      for (int i = 0; i < L.N && !any; ++i) any = p.cmd_timer[i] - 1.f <= 0.f && (float)(p.ep_len[i] + 1) < L.max_episode_length - 1.f;
    blk->dyn.cb_any = any;
  }
  for (int bid = 0; bid < (L.N + GO2_WG_ENVS - 1) / GO2_WG_ENVS; ++bid) {      // one workgroup at a time, its 256 threads as fibres
    EmuArgs a = {&sh, blk, actions_in, initial_reset, bid, mode, &outs};
    xl::run_group(GO2_WG_THREADS, emu_thread, &a);
  }
  if (mode != MODE_PHYS) {   // == go2_finish_kernel
    float* acc = p.ep_accum; float cnt = acc[GO2_NUM_REWARDS];
    const float track = acc[GO2_REW_TRACKING_LIN_VEL], cb = acc[GO2_NUM_REWARDS + 1];
    for (int i = 0; i <= GO2_NUM_REWARDS; ++i) { if (cnt > 0.f) p.episode_info[i] = i < GO2_NUM_REWARDS ? acc[i] / cnt / L.episode_length_s : cnt; acc[i] = 0.f; }
    acc[GO2_NUM_REWARDS + 1] = 0.f;
    if (cnt > 0.f) {      // terrain_level_* (:231-237)
      double ls[GO2_NUM_TERRAIN_KINDS + 1] = {0}, lc[GO2_NUM_TERRAIN_KINDS + 1] = {0};
      if (L.terrain_mode != 0) for (int e = 0; e < L.N; ++e) { const double lv = (double)p.terrain_levels[e]; const int kd = p.terrain_kind[e]; ls[0] += lv; lc[0] += 1; if (kd >= 0 && kd < GO2_NUM_TERRAIN_KINDS) { ls[1 + kd] += lv; lc[1 + kd] += 1; } }
      for (int k = 0; k <= GO2_NUM_TERRAIN_KINDS; ++k) p.episode_info[GO2_NUM_REWARDS + 3 + k] = L.terrain_mode == 0 ? (k == 0 ? 0.f : NAN) : (lc[k] > 0 ? (float)(ls[k] / lc[k]) : NAN);
    }
    go2_track_cmd_curriculum(L, blk->dyn, cnt, track, cb, blk->dyn.common_step_counter + counter_inc, p.episode_info + GO2_NUM_REWARDS + 1);
    blk->dyn.common_step_counter += counter_inc; blk->dyn.step_count += 1; blk->dyn.use_injected = 0;
    if (outs.episode_info_out) memcpy(outs.episode_info_out, p.episode_info, sizeof(float) * GO2_EPISODE_INFO_LEN);
  }
}
#endif

// Enqueue one pass.  Nothing computed on the host enters the kernels: the per-step scalars are derived on the
// device from the device-resident counters, so the same enqueue can be captured in a HIP graph and replayed.
static int launch(Go2Sim* s, int mode, const float* actions_in, int initial_reset, int counter_inc, void* stream, const Go2StepOutputs* outs_ = nullptr) {
  const Go2StepOutputs outs = outs_ ? *outs_ : Go2StepOutputs{};
  bool capturing = false;   // a captured enqueue executes nothing now: the host mirror advances on go2sim_notify_replayed instead
#ifdef GO2_EMU
  (void)stream; emu_run(s, mode, actions_in, initial_reset, counter_inc, outs);
#else
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((s->N + GO2_WG_ENVS - 1) / GO2_WG_ENVS), block(GO2_WG_THREADS);
  bool timed = s->timing != 0;
  { hipStreamCaptureStatus cs = hipStreamCaptureStatusNone; if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) { timed = false; capturing = true; } }
  if (timed) {
    if (s->ev_used + 2 > s->ev.size()) { size_t n0 = s->ev.size(); s->ev.resize(n0 + 512); for (size_t i = n0; i < s->ev.size(); ++i) HIPCHK(hipEventCreate(&s->ev[i])); }
    HIPCHK(hipEventRecord(s->ev[s->ev_used], st));
  }
  if ((mode & MODE_POST) && s->h.L.heading_command) hipLaunchKernelGGL(go2_cb_scan_kernel, dim3(1), dim3(256), 0, st, s->d_blk);
  if (mode == MODE_RESET_ALL) hipLaunchKernelGGL(go2_step_kernel<MODE_RESET_ALL>, grid, block, 0, st, s->d_blk, actions_in, initial_reset, outs);
  else if (mode == (MODE_PHYS | MODE_POST)) hipLaunchKernelGGL(go2_step_kernel<MODE_PHYS | MODE_POST>, grid, block, 0, st, s->d_blk, actions_in, initial_reset, outs);
  else if (mode == MODE_PHYS) hipLaunchKernelGGL(go2_step_kernel<MODE_PHYS>, grid, block, 0, st, s->d_blk, actions_in, initial_reset, outs);
  else hipLaunchKernelGGL(go2_step_kernel<MODE_POST>, grid, block, 0, st, s->d_blk, actions_in, initial_reset, outs);
  if (timed) { HIPCHK(hipEventRecord(s->ev[s->ev_used + 1], st)); s->ev_used += 2; }
  if (mode != MODE_PHYS) hipLaunchKernelGGL(go2_finish_kernel, dim3(1), dim3(256), 0, st, s->d_blk, counter_inc, outs.episode_info_out);
  HIPCHK(hipGetLastError());
#endif
  if (mode != MODE_PHYS && !capturing) { s->h.dyn.common_step_counter += counter_inc; s->h.dyn.step_count += 1; s->h.dyn.use_injected = 0; }   // host mirror
  return 0;
}

// oracle-less debugging aid used by tools/kbench.py (not part of include/go2sim.h): per-workgroup phase timestamps
int go2sim_debug_clock(Go2Sim* s, long long* dev_buf) {
  if (!s) return GO2SIM_EINVAL;
  s->h.p.dbg_clock = dev_buf;
  dev_upload(&s->d_blk->p, &s->h.p, sizeof(Go2Ptrs));
  return 0;
}
int go2sim_enable_timing(Go2Sim* s, int en) {
  if (!s) return GO2SIM_EINVAL;
  s->timing = en; s->time_ms = 0; s->time_launches = 0;
#ifndef GO2_EMU
  s->ev_used = 0;
#endif
  return 0;
}
int go2sim_kernel_time(Go2Sim* s, double* ms, int64_t* n) {
  if (!s || !ms || !n) FAIL(GO2SIM_EINVAL, "null argument");
#ifndef GO2_EMU
  for (size_t i = 0; i + 1 < s->ev_used; i += 2) { HIPCHK(hipEventSynchronize(s->ev[i + 1])); float t = 0; HIPCHK(hipEventElapsedTime(&t, s->ev[i], s->ev[i + 1])); s->time_ms += t; s->time_launches++; }
  s->ev_used = 0;
#endif
  *ms = s->time_ms; *n = s->time_launches; s->time_ms = 0; s->time_launches = 0;
  return 0;
}
int go2sim_notify_replayed(Go2Sim* s, int32_t steps) {
  if (!s || steps < 0) return GO2SIM_EINVAL;
  s->h.dyn.common_step_counter += steps; s->h.dyn.step_count += (uint64_t)steps; s->h.dyn.use_injected = 0;
  return 0;
}
int go2sim_reset_all(Go2Sim* s, void* stream) { if (!s) FAIL(GO2SIM_EINVAL, "null handle"); return launch(s, MODE_RESET_ALL, nullptr, 1, 0, stream); }
#ifndef GO2_EMU
__global__ void go2_mark_kernel(uint8_t* mask, const int32_t* ids, int count, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) { const int e = ids[i]; if (e >= 0 && e < N) mask[e] = 1; }
}
#endif
// reset_idx(env_ids) from outside a step: the reset branch of post-physics (go2_post.h) over the listed environments only
int go2sim_reset_idx(Go2Sim* s, const int32_t* env_ids, int32_t count, void* stream) {
  if (!s || count < 0 || (count > 0 && !env_ids)) FAIL(GO2SIM_EINVAL, "bad argument");
  if (count == 0) return 0;                                   // legged_robot.py:189-190
#ifdef GO2_EMU
  memset(s->d_blk->p.reset_mask, 0, (size_t)s->N);
  for (int i = 0; i < count; ++i) if (env_ids[i] >= 0 && env_ids[i] < s->N) s->d_blk->p.reset_mask[env_ids[i]] = 1;
#else
  HIPCHK(hipMemsetAsync(s->h.p.reset_mask, 0, (size_t)s->N, (hipStream_t)stream));
  hipLaunchKernelGGL(go2_mark_kernel, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream, s->h.p.reset_mask, env_ids, count, s->N);
#endif
  return launch(s, MODE_RESET_ALL, nullptr, 2, 0, stream);
}
int go2sim_simulate(Go2Sim* s, void* stream) { if (!s) FAIL(GO2SIM_EINVAL, "null handle"); return launch(s, MODE_PHYS, nullptr, 0, 0, stream); }
int go2sim_post_physics(Go2Sim* s, void* stream) { if (!s) FAIL(GO2SIM_EINVAL, "null handle"); return launch(s, MODE_POST, nullptr, 0, 1, stream); }
int go2sim_step(Go2Sim* s, const float* actions, void* stream) {
  if (!s || !actions) FAIL(GO2SIM_EINVAL, "null argument");
  return launch(s, MODE_PHYS | MODE_POST, actions, 0, 1, stream);   // counter += 1: legged_robot.py:112
}
int go2sim_step_rollout(Go2Sim* s, const float* actions, const Go2StepOutputs* out, void* stream) {
  if (!s || !actions) FAIL(GO2SIM_EINVAL, "null argument");
  return launch(s, MODE_PHYS | MODE_POST, actions, 0, 1, stream, out);
}
// The API tensors ARE the simulator state in this library, so a write is committed as soon as it is made.
int go2sim_set_root_state_indexed(Go2Sim* s, const int32_t*, int32_t, void*) { return s ? 0 : GO2SIM_EINVAL; }
int go2sim_set_dof_state_indexed(Go2Sim* s, const int32_t*, int32_t, void*) { return s ? 0 : GO2SIM_EINVAL; }
int go2sim_set_common_step_counter(Go2Sim* s, int64_t v) { if (!s) return GO2SIM_EINVAL; s->h.dyn.common_step_counter = v; blk_sync_dyn(s, nullptr); return 0; }
int64_t go2sim_get_common_step_counter(Go2Sim* s) { return s ? s->h.dyn.common_step_counter : -1; }
// The curriculum scales are pure functions of common_step_counter // num_steps_per_env here (the reference
// refreshes them when counter % 24 == 0 or when forced, legged_robot.py:144-152 — the same values except
// between a manual counter change and the next refresh), so there is nothing to force.
int go2sim_update_reward_curriculum(Go2Sim* s, int) { return s ? 0 : GO2SIM_EINVAL; }
int go2sim_get_curriculum_state(Go2Sim* s, float* rcs, float cr[4][2], float* zp) {
  if (!s) return GO2SIM_EINVAL;
  Go2Step S; go2_step_scalars(s->h.L, s->h.dyn, nullptr, s->h.dyn.common_step_counter, 0, &S);
  if (rcs) for (int t = 0; t < GO2_NUM_REWARDS; ++t) rcs[t] = s->h.L.rew_scale_dt[t] != 0.f ? S.rew_scale[t] / s->h.L.rew_scale_dt[t] : 1.f;
  if (cr) for (int r = 0; r < 4; ++r) { cr[r][0] = S.cmd_ranges[r][0]; cr[r][1] = S.cmd_ranges[r][1]; }
  if (zp) *zp = S.zero_cmd_proba;
  return 0;
}
int go2sim_inject_uniforms(Go2Sim* s, const float* u, void* stream) {
  if (!s) FAIL(GO2SIM_EINVAL, "null handle");
  if (!u) { s->h.dyn.use_injected = 0; blk_sync_dyn(s, stream); return 0; }
#ifdef GO2_EMU
  memcpy(s->inj_storage, u, sizeof(float) * (size_t)s->N * GO2_NUM_UNIFORMS);
#else
  HIPCHK(hipMemcpyAsync(s->inj_storage, u, sizeof(float) * (size_t)s->N * GO2_NUM_UNIFORMS, hipMemcpyDefault, (hipStream_t)stream));
#endif
  s->h.dyn.use_injected = 1; blk_sync_dyn(s, stream);
  return 0;
}
int go2sim_peek_uniforms(Go2Sim* s, float* out, void* stream) {
  if (!s || !out) FAIL(GO2SIM_EINVAL, "null argument");
  uint64_t sc = s->h.dyn.step_count;
#ifdef GO2_EMU
  (void)stream;
  for (int e = 0; e < s->N; ++e) for (int k = 0; k < GO2_NUM_UNIFORMS; ++k) out[(size_t)e * GO2_NUM_UNIFORMS + k] = u01_host(s->cfg.seed, (uint32_t)(s->cfg.env_offset + e), (uint32_t)s->d_tables->slot_code[k], sc);
#else
  int n = s->N * GO2_NUM_UNIFORMS;
  hipLaunchKernelGGL(go2_peek_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, out, s->d_tables, s->N, s->cfg.env_offset, (uint32_t)sc, (uint32_t)(sc >> 32), (uint32_t)s->cfg.seed, (uint32_t)(s->cfg.seed >> 32));
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

// ---- test hooks --------------------------------------------------------------------------------------------------------------
int go2sim_debug_torque_trace(Go2Sim* s, const float* actions_raw, const float* dof, float* out, void* stream) {
  if (!s || !actions_raw || !dof || !out) FAIL(GO2SIM_EINVAL, "null argument");
#ifdef GO2_EMU
  (void)stream;
  Go2DevBlock* blk = s->d_blk; Go2Tables& tab = *s->d_tables; const Go2Ptrs& p = blk->p; const Go2Launch& L = blk->L; const int N = L.N;
  Go2Step S; go2_step_scalars(L, blk->dyn, p.inj_storage, blk->dyn.common_step_counter, 0, &S);
  for (int e = 0; e < N; ++e) for (int lane = 0; lane < 4; ++lane) {
    static thread_local LegPhys ph_; static thread_local LegPost po_; static thread_local LaneAux ax;
    const float delay_u = philox_u01((uint32_t)(L.env_offset + e), 0u, S.step_lo, S.step_hi, L.seed_lo, L.seed_hi, GO2_U_DELAY & 3);
    lane_load_phys(ph_, po_, ax, tab, p, L, S, actions_raw, delay_u, e, lane, 0);
    for (int sub = 0; sub < L.decimation; ++sub) {
      const bool old = L.rand_delay && sub < ax.start;
      const float a[3] = {old ? ax.act_old[0] : ax.act_new[0], old ? ax.act_old[1] : ax.act_new[1], old ? ax.act_old[2] : ax.act_new[2]};
      for (int j = 0; j < 3; ++j) { const float* d = dof + (((size_t)sub * N + e) * 12 + 3 * lane + j) * 2; ph_.q[j] = d[0]; ph_.qd[j] = d[1]; }
      LegLoop tl; tl.load(tab.leg[lane], 0);
      ph_.pd(tl, L, a, ax.kp, ax.kd, ax.q0, ax.zoff, ax.strength, p.last_dof_vel, N, e);
      for (int j = 0; j < 3; ++j) { out[((size_t)sub * N + e) * 12 + 3 * lane + j] = ph_.tau[j]; F2D(p.torques, 3 * lane + j, e) = ph_.tau[j]; }
    }
  }
#else
  hipLaunchKernelGGL(go2_torque_trace_kernel, dim3((s->N + 15) / 16), dim3(64), 0, (hipStream_t)stream, s->d_blk, actions_raw, dof, out);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}
int go2sim_debug_contact_query(Go2Sim* s, const float* pts, float* out, int32_t n, void* stream) {
  if (!s || !pts || !out || n <= 0) FAIL(GO2SIM_EINVAL, "bad argument");
#ifdef GO2_EMU
  (void)stream;
  for (int i = 0; i < n; ++i) { static thread_local LegPhys ph; float gap; V3 nn;
    ph.contact_query(s->d_blk->L, s->d_blk->p.hf_cells, v3(pts[4 * i], pts[4 * i + 1], pts[4 * i + 2]), pts[4 * i + 3], &gap, &nn);
    out[4 * i] = gap; out[4 * i + 1] = nn.x; out[4 * i + 2] = nn.y; out[4 * i + 3] = nn.z; }
#else
  hipLaunchKernelGGL(go2_contact_query_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, s->d_blk, pts, out, n);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}
int go2sim_debug_traffic_probe(const float* in, float* out, int32_t N, int32_t nread, int32_t nwrite, void* stream) {
  if (!in || !out || N <= 0 || nread < 0 || nwrite < 0) FAIL(GO2SIM_EINVAL, "bad argument");
#ifdef GO2_EMU
  (void)stream;
  for (int e = 0; e < N; ++e) { float acc = 0.f; for (int f = 0; f < nread; ++f) acc += in[(size_t)f * N + e]; for (int f = 0; f < nwrite; ++f) out[(size_t)f * N + e] = acc + (float)f; }
#else
  hipLaunchKernelGGL(go2_traffic_probe_kernel, dim3((N + GO2_WG_ENVS - 1) / GO2_WG_ENVS), dim3(GO2_WG_THREADS), 0, (hipStream_t)stream, in, out, N, nread, nwrite);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}
int go2sim_debug_strict_ops(const float* a, const float* b, float* out, int32_t n, void* stream) {
  if (!a || !b || !out || n <= 0) FAIL(GO2SIM_EINVAL, "bad argument");
#ifdef GO2_EMU
  (void)stream;
  const double inv = 1.0 / (double)b[0];
  for (int i = 0; i < n; ++i) { out[i] = go2_mul_rn(a[i], b[i]); out[n + i] = go2_add_rn(a[i], b[i]); out[2 * (size_t)n + i] = go2_sub_rn(a[i], b[i]);
    out[3 * (size_t)n + i] = go2_div_rn(a[i], b[i]); out[4 * (size_t)n + i] = go2_sqrt_rn(fabsf(a[i])); out[5 * (size_t)n + i] = go2_mul_inv_rn(a[i], inv); }
#else
  float b0 = 0.f; HIPCHK(hipMemcpyAsync(&b0, b, sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream)); HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  hipLaunchKernelGGL(go2_strict_ops_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, b, out, n, 1.0 / (double)b0);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

int go2sim_gae(const float* rewards, const uint8_t* dones, const float* values, const float* last_values, float* returns, float* advantages, double* partials,
               int32_t T, int32_t N, float gamma, float lam, void* stream) {
  if (!rewards || !dones || !values || !last_values || !returns || !advantages || T <= 0 || N <= 0) FAIL(GO2SIM_EINVAL, "bad argument");
#ifdef GO2_EMU
  (void)stream; double s1 = 0, s2 = 0;
  for (int e = 0; e < N; ++e) { float a = 0.f;
    for (int t = T - 1; t >= 0; --t) { float nv = t == T - 1 ? last_values[e] : values[(size_t)(t + 1) * N + e]; float nt = 1.f - (dones[(size_t)t * N + e] ? 1.f : 0.f); float v = values[(size_t)t * N + e];
      float delta = rewards[(size_t)t * N + e] + nt * gamma * nv - v; a = delta + nt * gamma * lam * a; float r = a + v; returns[(size_t)t * N + e] = r; float ad = r - v; advantages[(size_t)t * N + e] = ad; s1 += ad; s2 += (double)ad * ad; } }
  if (partials) { partials[0] += s1; partials[1] += s2; partials[2] += (double)T * N; }
#else
  hipLaunchKernelGGL(go2_gae_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, rewards, dones, values, last_values, returns, advantages, partials, T, N, gamma, lam);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}
int go2sim_normalize_advantages(float* adv, const double* partials, int32_t count, void* stream) {
  if (!adv || !partials || count <= 0) FAIL(GO2SIM_EINVAL, "bad argument");
#ifdef GO2_EMU
  (void)stream; double n = partials[2], mean = partials[0] / n, var = (partials[1] - n * mean * mean) / (n - 1.0); float sd = (float)sqrt(var > 0 ? var : 0.0), m = (float)mean;
  for (int i = 0; i < count; ++i) adv[i] = (adv[i] - m) / (sd + 1e-8f);
#else
  int blocks = (count + 255) / 256; blocks = blocks > 2048 ? 2048 : blocks;
  hipLaunchKernelGGL(go2_normalize_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, adv, partials, count);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

int go2sim_ppo_loss(const float* mu, const float* std_, const float* value, const float* actions, const float* old_mu, const float* old_sigma,
                    const float* old_logp, const float* adv, const float* tv, const float* ret, float* gmu, float* gstd, float* gval, float* stats,
                    float* workspace, int32_t B, int32_t A, float clip, float vcoef, float ecoef, int32_t use_clip_v, int32_t split, void* stream) {
  if (split < 0 || split >= B) split = 0;
  const float w_head = split ? 1.f / (float)split : 1.f / (float)B, w_tail = split ? 1.f / (float)(B - split) : 1.f / (float)B;
  if (!split) split = B;
  if (!mu || !std_ || !value || !actions || !old_mu || !old_sigma || !old_logp || !adv || !tv || !ret || !gmu || !gstd || !gval || !stats || !workspace || B <= 0 || A <= 0 || A > 16)
    FAIL(GO2SIM_EINVAL, "bad argument (1 <= A <= 16)");
#ifdef GO2_EMU
  (void)stream; (void)workspace;
  double s_sur = 0, s_vl = 0, s_kl = 0, s_ent = 0; double gs[20]; for (int j = 0; j < A; ++j) gs[j] = 0;
  const float LOG2PI = 1.8378770664093453f;
  for (int i = 0; i < B; ++i) {
    float lp = 0.f, kl = 0.f, ent = 0.f;
    for (int j = 0; j < A; ++j) { float sg = std_[j], d = actions[(size_t)i * A + j] - mu[(size_t)i * A + j], ls = logf(sg);
      lp += -d * d / (2.f * sg * sg) - ls - 0.5f * LOG2PI; ent += 0.5f + 0.5f * LOG2PI + ls;
      float so = old_sigma[(size_t)i * A + j], dm = old_mu[(size_t)i * A + j] - mu[(size_t)i * A + j]; kl += logf(sg / so + 1e-5f) + (so * so + dm * dm) / (2.f * sg * sg) - 0.5f; }
    float ratio = expf(lp - old_logp[i]), a = adv[i], lo = 1.f - clip, hi = 1.f + clip, rc = fminf(fmaxf(ratio, lo), hi), in = (ratio >= lo && ratio <= hi) ? 1.f : 0.f;
    float s1 = -a * ratio, s2 = -a * rc, sur = fmaxf(s1, s2), w = s1 > s2 ? 1.f : (s1 < s2 ? in : 0.5f + 0.5f * in), wr = i < split ? w_head : w_tail, g_lp = -a * w * ratio * wr;
    float v = value[i], dv = v - tv[i], vl, gv;
    if (use_clip_v) { float dc = fminf(fmaxf(dv, -clip), clip), vin = (dv >= -clip && dv <= clip) ? 1.f : 0.f, vc = tv[i] + dc, l1 = (v - ret[i]) * (v - ret[i]), l2 = (vc - ret[i]) * (vc - ret[i]);
      vl = fmaxf(l1, l2); float g1 = 2.f * (v - ret[i]), g2 = 2.f * (vc - ret[i]) * vin; gv = l1 > l2 ? g1 : (l1 < l2 ? g2 : 0.5f * g1 + 0.5f * g2); }
    else { vl = (ret[i] - v) * (ret[i] - v); gv = 2.f * (v - ret[i]); }
    gval[i] = vcoef * gv / (float)B;
    for (int j = 0; j < A; ++j) { float sg = std_[j], d = actions[(size_t)i * A + j] - mu[(size_t)i * A + j]; gmu[(size_t)i * A + j] = g_lp * d / (sg * sg); gs[j] += g_lp * (d * d / (sg * sg * sg) - 1.f / sg); }
    s_sur += sur * wr; s_vl += vl; s_kl += kl; s_ent += ent;
  }
  for (int j = 0; j < A; ++j) gstd[j] = (float)(gs[j] - ecoef / std_[j]);
  stats[0] = (float)s_sur; stats[1] = (float)(s_vl / B); stats[2] = (float)(s_kl / B); stats[3] = (float)(s_ent / B); stats[4] = stats[0] + vcoef * stats[1] - ecoef * stats[3];
#else
  int nb = (B + PPO_ROWS - 1) / PPO_ROWS;
  hipLaunchKernelGGL(go2_ppo_loss_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, mu, std_, value, actions, old_mu, old_sigma, old_logp, adv, tv, ret, gmu, gval, workspace, B, A, clip, vcoef, use_clip_v, split, w_head, w_tail);
  hipLaunchKernelGGL(go2_ppo_loss_finish_kernel, dim3(1), dim3(16 * PPO_NSTAT), 0, (hipStream_t)stream, workspace, std_, gstd, stats, nb, B, A, vcoef, ecoef);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

int go2sim_elu_backward_bias(const float* gy, const float* y, float* gz, float* gb, float* workspace, int32_t B, int32_t C, void* stream) {
  if (!gy || !y || !gz || !gb || !workspace || B <= 0 || C <= 0 || (C & 3)) FAIL(GO2SIM_EINVAL, "bad argument (C must be a multiple of 4)");
#ifdef GO2_EMU
  (void)stream; (void)workspace;
  for (int c = 0; c < C; ++c) gb[c] = 0.f;
  for (int r = 0; r < B; ++r) for (int c = 0; c < C; ++c) { const size_t k = (size_t)r * C + c; const float o = gy[k] * (y[k] > 0.f ? 1.f : y[k] + 1.f); gz[k] = o; gb[c] += o; }
#else
  const int nr = (B + EB_ROWS - 1) / EB_ROWS;
  hipLaunchKernelGGL(go2_elu_bwd_bias_kernel, dim3((C / 4 + 63) / 64, nr), dim3(256), 0, (hipStream_t)stream, gy, y, gz, workspace, B, C);
  hipLaunchKernelGGL(go2_colsum_finish_kernel, dim3((C + 15) / 16), dim3(256), 0, (hipStream_t)stream, workspace, gb, nr, C);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

static int adam_check(const Go2AdamTensors* t) {
  if (!t || t->count <= 0 || t->count > GO2_ADAM_MAX_TENSORS) return 0;
  for (int i = 0; i < t->count; ++i) if (t->numel[i] <= 0 || !t->param[i] || !t->grad[i] || !t->exp_avg[i] || !t->exp_avg_sq[i] || !t->step[i]) return 0;
  return 1;
}
int go2sim_adam_workspace_len(const Go2AdamTensors* t) {
  if (!adam_check(t)) return GO2SIM_EINVAL;
  int nb = 0; for (int i = 0; i < t->count; ++i) nb += (t->numel[i] + GO2_ADAM_CHUNK - 1) / GO2_ADAM_CHUNK;
  return nb;
}
int go2sim_adam_clip_step(const Go2AdamTensors* t, float* lr, const float* kl_mean, float desired_kl, float max_grad_norm, double beta1_, double beta2_, double eps_,
                          float* workspace, void* stream) {
  if (!adam_check(t) || !lr || !workspace) FAIL(GO2SIM_EINVAL, "bad argument");
  const float beta1 = (float)beta1_, beta2 = (float)beta2_, omb1 = (float)(1.0 - beta1_), omb2 = (float)(1.0 - beta2_), eps = (float)eps_;
#ifdef GO2_EMU
  (void)stream; (void)workspace;
  if (kl_mean) { const float kl = *kl_mean, r = *lr; *lr = kl > desired_kl * 2.f ? fmaxf(1e-5f, r / 1.5f) : ((kl < desired_kl / 2.f && kl > 0.f) ? fminf(1e-2f, r * 1.5f) : r); }
  double ss = 0; for (int i = 0; i < t->count; ++i) for (int k = 0; k < t->numel[i]; ++k) ss += (double)t->grad[i][k] * t->grad[i][k];
  const float coef = fminf(max_grad_norm / ((float)sqrt(ss) + 1e-6f), 1.f);
  for (int i = 0; i < t->count; ++i) {
    const float st = t->step[i][0] + 1.f;
    const float bc1 = (float)(1.0 - pow(beta1_, (double)st)), bc2s = (float)sqrt(1.0 - pow(beta2_, (double)st)), step_size = *lr / bc1;
    for (int k = 0; k < t->numel[i]; ++k) {
      const float gk = t->grad[i][k] * coef; float& m = t->exp_avg[i][k]; float& v = t->exp_avg_sq[i][k];
      m += (gk - m) * omb1; v = beta2 * v + omb2 * gk * gk;
      t->param[i][k] -= step_size * m / (sqrtf(v) / bc2s + eps);
    }
    t->step[i][0] += 1.f;
  }
#else
  Go2AdamLaunch a; a.t = *t; a.first[0] = 0;
  for (int i = 0; i < t->count; ++i) a.first[i + 1] = a.first[i] + (t->numel[i] + GO2_ADAM_CHUNK - 1) / GO2_ADAM_CHUNK;
  const int nb = a.first[t->count];
  hipLaunchKernelGGL(go2_adam_norm_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, a, workspace, lr, kl_mean, desired_kl);
  hipLaunchKernelGGL(go2_adam_step_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, a, workspace, lr, max_grad_norm, beta1_, beta2_, eps);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

int go2sim_act_head(const float* mu, const float* std_, const float* eps, const float* value, float* a_out, float* a_st, float* mu_st, float* sig_st, float* lp_st,
                    float* v_st, int32_t N, int32_t A, void* stream) {
  if (!mu || !std_ || !eps || !a_out || (v_st && !value) || N <= 0 || A <= 0) FAIL(GO2SIM_EINVAL, "bad argument");
#ifdef GO2_EMU
  (void)stream;
  for (int e = 0; e < N; ++e) {
    float lp = 0.f;
    for (int j = 0; j < A; ++j) {
      const size_t k = (size_t)e * A + j; const float m = mu[k], sg = std_[j], a = m + sg * eps[k], d = a - m;
      lp += -(d * d) / (2.f * sg * sg) - logf(sg) - 0.9189385332046727f;
      a_out[k] = a; if (a_st) a_st[k] = a; if (mu_st) mu_st[k] = m; if (sig_st) sig_st[k] = sg;
    }
    if (lp_st) lp_st[e] = lp;
    if (v_st) v_st[e] = value[e];
  }
#else
  if (A <= 16) hipLaunchKernelGGL(go2_act_head16_kernel, dim3((N * 16 + 255) / 256), dim3(256), 0, (hipStream_t)stream, mu, std_, eps, value, a_out, a_st, mu_st, sig_st, lp_st, v_st, N, A);
  else hipLaunchKernelGGL(go2_act_head_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, mu, std_, eps, value, a_out, a_st, mu_st, sig_st, lp_st, v_st, N, A);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

int go2sim_store_transition(const float* rew, const uint8_t* dones, const uint8_t* touts, const float* v_st, float* rew_st, uint8_t* dones_st, float gamma, int32_t N, void* stream) {
  if (!rew || !dones || !rew_st || !dones_st || (touts && !v_st) || N <= 0) FAIL(GO2SIM_EINVAL, "bad argument");
#ifdef GO2_EMU
  (void)stream;
  for (int e = 0; e < N; ++e) { float r = rew[e]; if (touts) r += gamma * (v_st[e] * (touts[e] ? 1.f : 0.f)); rew_st[e] = r; dones_st[e] = dones[e]; }
#else
  hipLaunchKernelGGL(go2_store_transition_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, rew, dones, touts, v_st, rew_st, dones_st, gamma, N);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

uint32_t go2sim_shuffle_index(uint32_t i, uint32_t n, uint32_t seed, uint32_t counter) { return n ? go2_shuffle_index(i, n, go2_shuffle_half_bits(n), seed, counter) : 0; }

int go2sim_shuffle_gather(const Go2GatherJob* jobs, int32_t njobs, int32_t rows, const int64_t* indices, uint32_t* key_state, float* clear, int32_t nclear, void* stream) {
  if (!jobs || njobs <= 0 || njobs > GO2_GATHER_MAX_JOBS || rows <= 0 || (!indices && !key_state) || (nclear > 0 && !clear)) FAIL(GO2SIM_EINVAL, "shuffle gather: bad argument");
  for (int j = 0; j < njobs; ++j) if (!jobs[j].src || !jobs[j].dst || jobs[j].row_floats <= 0 || (jobs[j].dst_pitch != 0 && jobs[j].dst_pitch < jobs[j].row_floats)) FAIL(GO2SIM_EINVAL, "shuffle gather: bad job %d", j);
  const int h = go2_shuffle_half_bits((uint32_t)rows);
#ifdef GO2_EMU
  (void)stream;
  for (int32_t r = 0; r < rows; ++r) {
    const int64_t sidx = indices ? indices[r] : (int64_t)go2_shuffle_index((uint32_t)r, (uint32_t)rows, h, key_state[0], key_state[1]);
    for (int j = 0; j < njobs; ++j) memcpy(jobs[j].dst + (size_t)r * (jobs[j].dst_pitch ? jobs[j].dst_pitch : jobs[j].row_floats), jobs[j].src + (size_t)sidx * jobs[j].row_floats, sizeof(float) * (size_t)jobs[j].row_floats);
  }
  if (!indices) key_state[1] += 1u;
  for (int i = 0; i < nclear; ++i) clear[i] = 0.f;
#else
  Go2GatherArgs a; memset(&a, 0, sizeof(a));
  for (int j = 0; j < njobs; ++j) { a.src[j] = jobs[j].src; a.dst[j] = jobs[j].dst; a.w[j] = jobs[j].row_floats; a.dp[j] = jobs[j].dst_pitch ? jobs[j].dst_pitch : jobs[j].row_floats; }
  a.njobs = njobs; a.rows = rows; a.h = h; a.indices = indices; a.key = key_state; a.clear = clear; a.nclear = nclear;
  const int nwg = (rows + GO2_GATHER_ROWS_PER_WG - 1) / GO2_GATHER_ROWS_PER_WG;
  hipLaunchKernelGGL(go2_shuffle_gather_kernel, dim3(nwg), dim3(256), 0, (hipStream_t)stream, a);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

int go2sim_cts_minibatch_indices(int64_t* out, int32_t nmb, int32_t nt, int32_t ns, const int64_t* map, uint32_t* key_state, void* stream) {
  if (!out || !key_state || nmb <= 0 || nt < nmb || ns < nmb) FAIL(GO2SIM_EINVAL, "cts indices: nmb >= 1 mini-batches of at least one teacher and one student sample each");
  const int tb = nt / nmb, sb = ns / nmb, ht = go2_shuffle_half_bits((uint32_t)nt), hs = go2_shuffle_half_bits((uint32_t)ns);
#ifdef GO2_EMU
  (void)stream;
  for (int i = 0; i < nmb; ++i) for (int j = 0; j < tb + sb; ++j) {
    const int64_t k = j < tb ? (int64_t)go2_shuffle_index((uint32_t)(i * tb + j), (uint32_t)nt, ht, key_state[0], key_state[1])
                             : (int64_t)nt + (int64_t)go2_shuffle_index((uint32_t)(i * sb + j - tb), (uint32_t)ns, hs, key_state[0] ^ GO2_SHUFFLE_TAIL_SEED, key_state[1]);
    out[(size_t)i * (tb + sb) + j] = map ? map[k] : k;
  }
  key_state[1] += 1u;
#else
  hipLaunchKernelGGL(go2_cts_indices_kernel, dim3((nmb * (tb + sb) + 255) / 256), dim3(256), 0, (hipStream_t)stream, out, nmb, nt, ns, tb, sb, ht, hs, map, key_state);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

int go2sim_history_push(float* history, const float* obs, const uint8_t* dones, int32_t N, int32_t H, int32_t D, void* stream) {
  if (!history || !obs || N <= 0 || H <= 0 || D <= 0) FAIL(GO2SIM_EINVAL, "bad argument");
#ifdef GO2_EMU
  (void)stream;
  for (int e = 0; e < N; ++e) for (int d = 0; d < D; ++d) {
    const bool z = dones && dones[e];
    float* h = history + (size_t)e * H * D + d;
    for (int k = 0; k + 1 < H; ++k) h[(size_t)k * D] = z ? 0.f : h[(size_t)(k + 1) * D];
    h[(size_t)(H - 1) * D] = obs[(size_t)e * D + d];
  }
#else
  hipLaunchKernelGGL(go2_history_push_kernel, dim3((N * D + 255) / 256), dim3(256), 0, (hipStream_t)stream, history, obs, dones, N, H, D);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

}  // extern "C"
