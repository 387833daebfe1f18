// go2_lane.h — the per-lane physics program: one lane = one (env, leg).
//
// Replaces gym.simulate + the torque loop of LeggedRobot.step (legged_gym/envs/base/legged_robot.py:73-92)
// with a new articulated-body model (the reference's physics is NVIDIA Isaac Gym / PhysX, which is not part of
// the reference tree; the model is specified in DESIGN.md section 4 and restated independently by the oracle).
//
// Four lanes of a quad own the four legs of one environment and replicate the floating base.  Per substep:
//   A. (lane)  leg kinematics, leg link inertias in the common frame, velocity-product forces, the leg's
//              3x3 joint-space inertia A, its inverse, B = [Ic_i S_i], and the leg's Schur terms
//              K = B A^-1 B^T, r = B A^-1 (tau - C)
//      (quad)  sum {Ic_leg (10), K (21), F_leg (6), r (6)} over the 4 lanes        <- DPP quad shuffles
//   B. (lane, replicated) base articulated inertia IA = I_base + sum Ic - sum K, Phi = IA^-1, base and joint
//              accelerations, unconstrained end-of-substep velocities
//   C. (lane)  contact candidates (foot sphere; deepest of the other leg/base spheres), joint-limit rows;
//              per row: Jc (3), Z = A^-1 Jc^T, G = Ec + Jc N, H = G Phi, diagonal
//      (quad)  projected Gauss-Seidel, lanes take turns, base velocity change w broadcast after each turn
//   D. (lane)  velocities, semi-implicit Euler integration, contact forces
// The cross-lane steps live in the caller (the HIP kernel uses DPP; the host emulation loops over 4 structs).
#pragma once
#include "go2_math.h"
#include "go2_tables.h"

#define GO2_QUAD_PARTIALS 43

struct ContactSlot {
  float Jc[3][3]; float Z[3][3]; SV G[3]; SV H[3]; float dinv[3]; float vfb[3]; float lam[3];
  float mu; float active; int32_t body; V3 nw, t1w, t2w;
};
struct LimitRow { float Z[3]; SV G; SV H; float dinv, vfb, lam, active, sgn; };

struct LegPhys {
  // ---- state (persistent over the substeps of one step) ----
  V3 pw, vw, ww; float qx, qy, qz, qw;
  float q[3], qd[3];
  RB Lhip, Lthigh, Lcalf, Ibase;
  float mu, rest;
  float lam_foot[3];
  // ---- substep temporaries that cross phase boundaries ----
  M3 Rwb, R1, R2, R3; V3 wb, vb, p1, p2, p3, a2;
  SV S1, S2, S3, B1, B2, B3, T1, T2, T3, f0, V0, V0f;
  float Ainv[6];  // 11 12 13 22 23 33
  float u[3], qdf[3];
  float Phi[21];
  ContactSlot cs[2]; LimitRow lr[3];
  SV w; float z[3];
  float tau[3];
  V3 force_foot, force_other; int32_t other_body;

  GO2_HD float ainv(int i, int j) const {
    const int idx[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
    return Ainv[idx[i][j]];
  }

  // ------------------------------------------------------------------------------------------------
  GO2_HD void phaseA(const LegTab& t, const Go2Launch& L, float* part) {
    Rwb = quat_to_m3(qx, qy, qz, qw);
    wb = mulT(Rwb, ww); vb = mulT(Rwb, vw);
    V3 gb = mulT(Rwb, v3(L.gravity[0], L.gravity[1], L.gravity[2]));
    float s1, c1, s2, c2, s3, c3;
    sincosf(q[0], &s1, &c1); sincosf(q[1], &s2, &c2); sincosf(q[2], &s3, &c3);
    float c23 = c2 * c3 - s2 * s3, s23 = s2 * c3 + c2 * s3;
    R1.x = v3(1, 0, 0); R1.y = v3(0, c1, s1); R1.z = v3(0, -s1, c1);
    R2.x = c2 * R1.x - s2 * R1.z; R2.y = R1.y; R2.z = s2 * R1.x + c2 * R1.z;
    R3.x = c23 * R1.x - s23 * R1.z; R3.y = R1.y; R3.z = s23 * R1.x + c23 * R1.z;
    p1 = v3(t.o1[0], t.o1[1], t.o1[2]);
    p2 = p1 + mul(R1, v3(t.o2[0], t.o2[1], t.o2[2]));
    p3 = p2 + mul(R2, v3(t.o3[0], t.o3[1], t.o3[2]));
    V3 a1 = v3(1, 0, 0); a2 = R1.y;
    S1 = sv(a1, cross(p1, a1)); S2 = sv(a2, cross(p2, a2)); S3 = sv(a2, cross(p3, a2));
    RB I1 = to_common(Lhip, R1, p1), I2 = to_common(Lthigh, R2, p2), I3 = to_common(Lcalf, R3, p3);
    RB Ic2 = I2 + I3, Ic1 = I1 + Ic2;
    V0 = sv(wb, vb);
    SV Sq1 = qd[0] * S1, Sq2 = qd[1] * S2, Sq3 = qd[2] * S3;
    SV V1 = V0 + Sq1, V2 = V1 + Sq2, V3_ = V2 + Sq3;
    SV ab0 = sv(v3(0, 0, 0), -gb);
    SV ab1 = ab0 + crm(V1, Sq1), ab2 = ab1 + crm(V2, Sq2), ab3 = ab2 + crm(V3_, Sq3);
    SV f1 = mul(I1, ab1) + crf(V1, mul(I1, V1));
    SV f2 = mul(I2, ab2) + crf(V2, mul(I2, V2));
    SV f3 = mul(I3, ab3) + crf(V3_, mul(I3, V3_));
    SV F2 = f2 + f3, F1 = f1 + F2;
    float C0 = dot(S1, F1), C1 = dot(S2, F2), C2 = dot(S3, f3);
    B1 = mul(Ic1, S1); B2 = mul(Ic2, S2); B3 = mul(I3, S3);
    float A11 = dot(S1, B1) + L.armature, A12 = dot(S1, B2), A13 = dot(S1, B3);
    float A22 = dot(S2, B2) + L.armature, A23 = dot(S2, B3), A33 = dot(S3, B3) + L.armature;
    float m11 = A22 * A33 - A23 * A23, m12 = A13 * A23 - A12 * A33, m13 = A12 * A23 - A13 * A22;
    float det = A11 * m11 + A12 * m12 + A13 * m13, id = 1.0f / det;
    Ainv[0] = m11 * id; Ainv[1] = m12 * id; Ainv[2] = m13 * id;
    Ainv[3] = (A11 * A33 - A13 * A13) * id; Ainv[4] = (A12 * A13 - A11 * A23) * id; Ainv[5] = (A11 * A22 - A12 * A12) * id;
    T1 = Ainv[0] * B1 + Ainv[1] * B2 + Ainv[2] * B3;
    T2 = Ainv[1] * B1 + Ainv[3] * B2 + Ainv[4] * B3;
    T3 = Ainv[2] * B1 + Ainv[4] * B2 + Ainv[5] * B3;
    u[0] = tau[0] - C0; u[1] = tau[1] - C1; u[2] = tau[2] - C2;
    SV rhs = u[0] * T1 + u[1] * T2 + u[2] * T3;
    f0 = mul(Ibase, ab0) + crf(V0, mul(Ibase, V0));
    // partials: Ic1 (10), K (21 packed), F1 (6), rhs (6)
    part[0] = Ic1.m; part[1] = Ic1.h.x; part[2] = Ic1.h.y; part[3] = Ic1.h.z;
    part[4] = Ic1.J.xx; part[5] = Ic1.J.yy; part[6] = Ic1.J.zz; part[7] = Ic1.J.xy; part[8] = Ic1.J.xz; part[9] = Ic1.J.yz;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j)
        part[10 + S6(i, j)] = get(T1, i) * get(B1, j) + get(T2, i) * get(B2, j) + get(T3, i) * get(B3, j);
    part[31] = F1.a.x; part[32] = F1.a.y; part[33] = F1.a.z; part[34] = F1.l.x; part[35] = F1.l.y; part[36] = F1.l.z;
    part[37] = rhs.a.x; part[38] = rhs.a.y; part[39] = rhs.a.z; part[40] = rhs.l.x; part[41] = rhs.l.y; part[42] = rhs.l.z;
  }

  // ------------------------------------------------------------------------------------------------
  GO2_HD void phaseB(const Go2Launch& L, const float* red) {
    RB Ileg; Ileg.m = red[0]; Ileg.h = v3(red[1], red[2], red[3]);
    Ileg.J.xx = red[4]; Ileg.J.yy = red[5]; Ileg.J.zz = red[6]; Ileg.J.xy = red[7]; Ileg.J.xz = red[8]; Ileg.J.yz = red[9];
    RB Itot = Ibase + Ileg;
    float IA[21]; rb_to_s6(Itot, IA);
#pragma unroll
    for (int i = 0; i < 21; ++i) IA[i] -= red[10 + i];
    spd6_inverse(IA, Phi);
    SV p0 = f0 + sv(v3(red[31], red[32], red[33]), v3(red[34], red[35], red[36]));
    SV r = sv(v3(red[37], red[38], red[39]), v3(red[40], red[41], red[42]));
    SV a0 = spd6_mul(Phi, p0 + r); a0 = -1.0f * a0;
    float e0 = u[0] - dot(B1, a0), e1 = u[1] - dot(B2, a0), e2 = u[2] - dot(B3, a0);
    float h = L.sim_dt;
    qdf[0] = qd[0] + h * (Ainv[0] * e0 + Ainv[1] * e1 + Ainv[2] * e2);
    qdf[1] = qd[1] + h * (Ainv[1] * e0 + Ainv[3] * e1 + Ainv[4] * e2);
    qdf[2] = qd[2] + h * (Ainv[2] * e0 + Ainv[4] * e1 + Ainv[5] * e2);
    V0f = V0 + h * a0;
  }

  // terrain height and unit normal under world point (x, y)
  GO2_HD void terrain(const Go2Launch& L, const int16_t* hf, float x, float y, float* hgt, V3* n) const {
    if (L.terrain_mode == 0) { *hgt = 0.f; *n = v3(0, 0, 1); return; }
    float fx = (x + L.hf_border) / L.hf_hscale, fy = (y + L.hf_border) / L.hf_hscale;
    int i = (int)floorf(fx), j = (int)floorf(fy);
    i = i < 0 ? 0 : (i > L.hf_rows - 2 ? L.hf_rows - 2 : i); j = j < 0 ? 0 : (j > L.hf_cols - 2 ? L.hf_cols - 2 : j);
    float uu = fminf(fmaxf(fx - i, 0.f), 1.f), vv = fminf(fmaxf(fy - j, 0.f), 1.f);
    float h00 = hf[i * L.hf_cols + j] * L.hf_vscale, h10 = hf[(i + 1) * L.hf_cols + j] * L.hf_vscale;
    float h01 = hf[i * L.hf_cols + j + 1] * L.hf_vscale, h11 = hf[(i + 1) * L.hf_cols + j + 1] * L.hf_vscale;
    float dx, dy;
    // two triangles per cell split along (i,j)-(i+1,j+1), the diagonal of the reference's trimesh (terrain_utils triangles (0,3,1),(0,2,3))
    if (uu >= vv) { dx = h10 - h00; dy = h11 - h10; }
    else { dx = h11 - h01; dy = h01 - h00; }
    *hgt = h00 + uu * dx + vv * dy;
    float nx = -dx / L.hf_hscale, ny = -dy / L.hf_hscale, inv = 1.0f / sqrtf(nx * nx + ny * ny + 1.f);
    *n = v3(nx * inv, ny * inv, inv);
  }

  GO2_HD void build_slot(ContactSlot& s, const Go2Launch& L, float gap, V3 cb, float rad, int link, int body, V3 nw, bool warm) {
    float h = L.sim_dt;
    s.active = gap < L.contact_offset ? 1.f : 0.f;
    s.body = body; s.mu = mu; s.nw = nw;
    V3 ex = v3(1, 0, 0); float dn = dot(ex, nw);
    V3 t1 = ex - dn * nw; t1 = (1.0f / sqrtf(dot(t1, t1))) * t1;
    s.t1w = t1; s.t2w = cross(nw, t1);
    V3 dirs[3] = {mulT(Rwb, s.nw), mulT(Rwb, s.t1w), mulT(Rwb, s.t2w)};
    V3 rb = cb - rad * dirs[0];
    V3 zero3 = v3(0, 0, 0);
    V3 col1 = sel(link >= 1, cross(v3(1, 0, 0), rb - p1), zero3);
    V3 col2 = sel(link >= 2, cross(a2, rb - p2), zero3);
    V3 col3 = sel(link >= 3, cross(a2, rb - p3), zero3);
    float cfm1 = 1.0f + L.cfm;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      V3 d = dirs[a];
      float j0 = dot(d, col1), j1 = dot(d, col2), j2 = dot(d, col3);
      s.Jc[a][0] = j0; s.Jc[a][1] = j1; s.Jc[a][2] = j2;
      SV Ec = sv(cross(rb, d), d);
      s.G[a] = Ec - (j0 * T1 + j1 * T2 + j2 * T3);
      s.Z[a][0] = Ainv[0] * j0 + Ainv[1] * j1 + Ainv[2] * j2;
      s.Z[a][1] = Ainv[1] * j0 + Ainv[3] * j1 + Ainv[4] * j2;
      s.Z[a][2] = Ainv[2] * j0 + Ainv[4] * j1 + Ainv[5] * j2;
      s.H[a] = spd6_mul(Phi, s.G[a]);
      float d_ = (j0 * s.Z[a][0] + j1 * s.Z[a][1] + j2 * s.Z[a][2] + dot(s.G[a], s.H[a])) * cfm1;
      s.dinv[a] = 1.0f / d_;
      s.vfb[a] = dot(Ec, V0f) + j0 * qdf[0] + j1 * qdf[1] + j2 * qdf[2];
      if (a == 0) {
        float vn_pre = dot(Ec, V0) + j0 * qd[0] + j1 * qd[1] + j2 * qd[2];
        float b = gap >= 0.f ? gap / h : gap * L.erp / h;
        b = fmaxf(b, -L.max_depen_vel);
        if (vn_pre < -L.bounce_thr && gap + vn_pre * h < 0.f) b = fminf(b, rest * vn_pre);
        s.vfb[0] += b;
      }
      s.lam[a] = (warm && s.active > 0.f) ? lam_foot[a] : 0.f;
    }
  }

  // ------------------------------------------------------------------------------------------------
  GO2_HD void phaseC(const LegTab& t, const Go2Launch& L, const int16_t* hf, float* dw_out) {
    // foot
    {
      V3 cb = p3 + mul(R3, v3(t.foot_pt[0], t.foot_pt[1], t.foot_pt[2]));
      V3 cw = pw + mul(Rwb, cb); float hh; V3 n; terrain(L, hf, cw.x, cw.y, &hh, &n);
      float gap = (cw.z - hh) * n.z - t.foot_pt[3];
      build_slot(cs[0], L, gap, cb, t.foot_pt[3], 3, t.body_index[3], n, true);
    }
    // deepest of the other candidates.  They are tabulated per link in a fixed order (hip, thigh, calf, then this lane's share of
    // the base points), so each link's world pose is formed once and a candidate costs one 3x3 transform; only the winner's
    // base-frame position is reconstructed afterwards.
    {
      float best = 1e30f; int bi = 0; V3 bn = v3(0, 0, 1);
      const M3 Rw1 = {mul(Rwb, R1.x), mul(Rwb, R1.y), mul(Rwb, R1.z)}, Rw2 = {mul(Rwb, R2.x), mul(Rwb, R2.y), mul(Rwb, R2.z)}, Rw3 = {mul(Rwb, R3.x), mul(Rwb, R3.y), mul(Rwb, R3.z)};
      const V3 o1 = pw + mul(Rwb, p1), o2 = pw + mul(Rwb, p2), o3 = pw + mul(Rwb, p3);
#define GO2_CAND(idx, PT, RW, OW) { \
        const V3 cw = OW + mul(RW, v3(PT[0], PT[1], PT[2])); float hh; V3 n; terrain(L, hf, cw.x, cw.y, &hh, &n); \
        const float gap = (cw.z - hh) * n.z - PT[3]; \
        if (gap < best) { best = gap; bi = (idx); bn = n; } }
#define GO2_CAND_UNROLL 1     // rolled: unrolling the 19 candidates raises register pressure into scratch (124 vs 117 us, measured)
#pragma unroll GO2_CAND_UNROLL
      for (int i = 0; i < GO2_N_HIP_PTS; ++i) GO2_CAND(i, t.other_pt[i], Rw1, o1)
#pragma unroll GO2_CAND_UNROLL
      for (int i = GO2_N_HIP_PTS; i < GO2_N_HIP_PTS + GO2_N_THIGH_PTS; ++i) GO2_CAND(i, t.other_pt[i], Rw2, o2)
#pragma unroll GO2_CAND_UNROLL
      for (int i = GO2_N_HIP_PTS + GO2_N_THIGH_PTS; i < GO2_NLEG_OTHER; ++i) GO2_CAND(i, t.other_pt[i], Rw3, o3)
#pragma unroll GO2_CAND_UNROLL
      for (int k = 0; k < GO2_LANE_BASE_PTS; ++k) if (k < t.n_base) GO2_CAND(GO2_NLEG_OTHER + k, t.base_pt[k], Rwb, pw)
#undef GO2_CAND
      const bool isb = bi >= GO2_NLEG_OTHER;
      const int kb = isb ? bi - GO2_NLEG_OTHER : 0, ko = isb ? 0 : bi;
      const V3 c = isb ? v3(t.base_pt[kb][0], t.base_pt[kb][1], t.base_pt[kb][2]) : v3(t.other_pt[ko][0], t.other_pt[ko][1], t.other_pt[ko][2]);
      const float brad = isb ? t.base_pt[kb][3] : t.other_pt[ko][3];
      const int bbody = isb ? t.base_body[kb] : t.other_body[ko];
      const int blink = isb ? 0 : (bi < GO2_N_HIP_PTS ? 1 : (bi < GO2_N_HIP_PTS + GO2_N_THIGH_PTS ? 2 : 3));
      const V3 bcb = sel(blink == 0, c, sel(blink == 1, p1 + mul(R1, c), sel(blink == 2, p2 + mul(R2, c), p3 + mul(R3, c))));
      build_slot(cs[1], L, best, bcb, brad, blink, bbody, bn, false);
    }
    // joint limits
    float h = L.sim_dt, cfm1 = 1.0f + L.cfm;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float glo = q[j] - t.lim_lo[j], ghi = t.lim_hi[j] - q[j];
      float sgn = 0.f, gap = 0.f;
      if (glo < L.limit_margin) { sgn = 1.f; gap = glo; } else if (ghi < L.limit_margin) { sgn = -1.f; gap = ghi; }
      LimitRow& r = lr[j];
      r.active = sgn != 0.f ? 1.f : 0.f; r.lam = 0.f; r.sgn = sgn;
      r.Z[0] = sgn * ainv(0, j); r.Z[1] = sgn * ainv(1, j); r.Z[2] = sgn * ainv(2, j);
      SV Tj = sv(sel(j == 0, T1.a, sel(j == 1, T2.a, T3.a)), sel(j == 0, T1.l, sel(j == 1, T2.l, T3.l)));
      r.G = (-sgn) * Tj;
      r.H = spd6_mul(Phi, r.G);
      float d_ = (sgn * r.Z[j] + dot(r.G, r.H)) * cfm1;
      r.dinv = d_ > 0.f ? 1.0f / d_ : 0.f;
      float b = gap >= 0.f ? gap / h : gap * L.erp / h; b = fmaxf(b, -10.0f);
      r.vfb = sgn * qdf[j] + b;
    }
    // warm start contribution
    SV dw = sv(v3(0, 0, 0), v3(0, 0, 0)); z[0] = z[1] = z[2] = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float l = cs[0].lam[a];
      dw = dw + l * cs[0].H[a];
      z[0] += cs[0].Z[a][0] * l; z[1] += cs[0].Z[a][1] * l; z[2] += cs[0].Z[a][2] * l;
    }
    dw_out[0] = dw.a.x; dw_out[1] = dw.a.y; dw_out[2] = dw.a.z; dw_out[3] = dw.l.x; dw_out[4] = dw.l.y; dw_out[5] = dw.l.z;
  }
  GO2_HD bool has_foot() const { return cs[0].active > 0.f; }
  GO2_HD bool has_other() const { return cs[1].active > 0.f; }
  GO2_HD bool has_limit() const { return (lr[0].active + lr[1].active + lr[2].active) > 0.f; }
  GO2_HD void set_w(const float* wsum) { w = sv(v3(wsum[0], wsum[1], wsum[2]), v3(wsum[3], wsum[4], wsum[5])); }

  // one Gauss-Seidel sweep over this lane's rows; `on` = 1 for the lane whose turn it is, else 0.
  // do_slot[s] / do_lim are WAVE-UNIFORM hints: false means no lane of the wave has such a row active this substep, so the
  // group is skipped as a whole (an inactive row contributes exactly +0, so skipping changes no bit of the result).
  GO2_HD void sweep(float on, float* dw_out, bool do_foot = true, bool do_other = true, bool do_lim = true) {
    SV dw = sv(v3(0, 0, 0), v3(0, 0, 0));
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (!(s == 0 ? do_foot : do_other)) continue;
      ContactSlot& c = cs[s];
      float m = on * c.active;
      {
        float v = c.vfb[0] + c.Jc[0][0] * z[0] + c.Jc[0][1] * z[1] + c.Jc[0][2] * z[2] + dot(c.G[0], w);
        float ln = fmaxf(0.f, c.lam[0] - v * c.dinv[0]);
        float dl = m * (ln - c.lam[0]); c.lam[0] += dl;
        z[0] += c.Z[0][0] * dl; z[1] += c.Z[0][1] * dl; z[2] += c.Z[0][2] * dl;
        SV d = dl * c.H[0]; w = w + d; dw = dw + d;
      }
      {
        float v1 = c.vfb[1] + c.Jc[1][0] * z[0] + c.Jc[1][1] * z[1] + c.Jc[1][2] * z[2] + dot(c.G[1], w);
        float v2 = c.vfb[2] + c.Jc[2][0] * z[0] + c.Jc[2][1] * z[1] + c.Jc[2][2] * z[2] + dot(c.G[2], w);
        float l1 = c.lam[1] - v1 * c.dinv[1], l2 = c.lam[2] - v2 * c.dinv[2];
        float lim = c.mu * c.lam[0], nn = sqrtf(l1 * l1 + l2 * l2);
        if (nn > lim) { float sc = nn > 0.f ? lim / nn : 0.f; l1 *= sc; l2 *= sc; }
        float d1 = m * (l1 - c.lam[1]), d2 = m * (l2 - c.lam[2]); c.lam[1] += d1; c.lam[2] += d2;
        z[0] += c.Z[1][0] * d1 + c.Z[2][0] * d2; z[1] += c.Z[1][1] * d1 + c.Z[2][1] * d2; z[2] += c.Z[1][2] * d1 + c.Z[2][2] * d2;
        SV d = d1 * c.H[1] + d2 * c.H[2]; w = w + d; dw = dw + d;
      }
    }
    if (do_lim)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      LimitRow& r = lr[j];
      float v = r.vfb + r.sgn * z[j] + dot(r.G, w);
      float ln = fmaxf(0.f, r.lam - v * r.dinv);
      float dl = on * r.active * (ln - r.lam); r.lam += dl;
      z[0] += r.Z[0] * dl; z[1] += r.Z[1] * dl; z[2] += r.Z[2] * dl;
      SV d = dl * r.H; w = w + d; dw = dw + d;
    }
    dw_out[0] = dw.a.x; dw_out[1] = dw.a.y; dw_out[2] = dw.a.z; dw_out[3] = dw.l.x; dw_out[4] = dw.l.y; dw_out[5] = dw.l.z;
  }
  GO2_HD void add_delta(const float* d) {
    w.a.x += d[0]; w.a.y += d[1]; w.a.z += d[2]; w.l.x += d[3]; w.l.y += d[4]; w.l.z += d[5];
  }
  // after a turn: add what the other lanes contributed (total - own)
  GO2_HD void add_others(const float* tot, const float* own) {
    w.a.x += tot[0] - own[0]; w.a.y += tot[1] - own[1]; w.a.z += tot[2] - own[2];
    w.l.x += tot[3] - own[3]; w.l.y += tot[4] - own[4]; w.l.z += tot[5] - own[5];
  }

  // ------------------------------------------------------------------------------------------------
  GO2_HD void phaseD(const LegTab& t, const Go2Launch& L) {
    float h = L.sim_dt;
    SV V0p = V0f + w;
    float qdp[3] = {qdf[0] + z[0] - dot(T1, w), qdf[1] + z[1] - dot(T2, w), qdf[2] + z[2] - dot(T3, w)};
#pragma unroll
    for (int j = 0; j < 3; ++j) qdp[j] = fminf(fmaxf(qdp[j], -t.vel_lim[j]), t.vel_lim[j]);
    V3 lin_b = V0p.l + h * cross(wb, vb);
    ww = mul(Rwb, V0p.a); vw = mul(Rwb, lin_b);
    {   // asset.max_angular_velocity / max_linear_velocity: the base twist is clamped, so no state can run off to inf (branch-free: x1 when inside)
      ww = fminf(1.f, L.max_ang_vel / sqrtf(fmaxf(dot(ww, ww), 1e-30f))) * ww;
      vw = fminf(1.f, L.max_lin_vel / sqrtf(fmaxf(dot(vw, vw), 1e-30f))) * vw;
    }
    pw = pw + h * vw;
    float th = sqrtf(dot(ww, ww)) * h; float dx, dy, dz, dwq;
    if (th > 1e-9f) { float sc = sinf(0.5f * th) / (th / h); dx = ww.x * sc; dy = ww.y * sc; dz = ww.z * sc; dwq = cosf(0.5f * th); }
    else { dx = ww.x * h * 0.5f; dy = ww.y * h * 0.5f; dz = ww.z * h * 0.5f; dwq = 1.f; }
    float nx = dwq * qx + dx * qw + dy * qz - dz * qy, ny = dwq * qy - dx * qz + dy * qw + dz * qx;
    float nz = dwq * qz + dx * qy - dy * qx + dz * qw, nw_ = dwq * qw - dx * qx - dy * qy - dz * qz;
    float inv = 1.0f / sqrtf(nx * nx + ny * ny + nz * nz + nw_ * nw_);
    qx = nx * inv; qy = ny * inv; qz = nz * inv; qw = nw_ * inv;
#pragma unroll
    for (int j = 0; j < 3; ++j) { qd[j] = qdp[j]; q[j] += h * qdp[j]; }
    float ih = 1.0f / h;
    force_foot = cs[0].active * ih * (cs[0].lam[0] * cs[0].nw + cs[0].lam[1] * cs[0].t1w + cs[0].lam[2] * cs[0].t2w);
    force_other = cs[1].active * ih * (cs[1].lam[0] * cs[1].nw + cs[1].lam[1] * cs[1].t1w + cs[1].lam[2] * cs[1].t2w);
    other_body = cs[1].body;
#pragma unroll
    for (int a = 0; a < 3; ++a) lam_foot[a] = cs[0].active * cs[0].lam[a];
  }

  // _compute_torques (legged_robot.py:594-618, control_type 'P') then *= motor_strengths (:80-81)
  // kp/kd arrive already multiplied by the per-env gain multipliers, q0 = default angle of this leg's joints (hoisted out of the substep loop)
  GO2_HD void pd(const LegTab& t, const Go2Launch& L, const float* act, const float* kp_, const float* kd_, const float* q0, const float* off, const float* strength) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float kp = kp_[j], kd = kd_[j];
      float tq = kp * (act[j] * L.action_scale + q0[j] - q[j] + off[j]) - kd * qd[j];
      tq = fminf(fmaxf(tq, -t.eff_lim[j]), t.eff_lim[j]);
      if (L.rand_strength) tq *= strength[j];
      tau[j] = tq;
    }
  }
};
