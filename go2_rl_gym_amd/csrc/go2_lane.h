// go2_lane.h — the per-lane physics program: one lane = one (env, leg, sub).
//
// Replaces gym.simulate + the torque loop of LeggedRobot.step (legged_gym/envs/base/legged_robot.py:73-92)
// with a new articulated-body model (the reference's physics is NVIDIA Isaac Gym / PhysX, which is not part of
// the reference tree; the model is specified in DESIGN.md section 4 and restated independently by the oracle).
//
// Sixteen lanes — one DPP row — own one environment: row lane = leg * 4 + sub (go2_xlane.h).  The four SUB-lanes of a leg hold the
// same leg state and run the leg's dynamics replicated; what they divide among themselves is the contact problem:
//   * the leg's collision candidates (each sub-lane tests a quarter, a 2-step quad tournament finds the deepest),
//   * every constraint row of the leg, SLICED: a row acts on the 9 numbers (z = change of the leg's 3 joint rates, w = change of the
//     base twist, angular | linear); sub-lane 0 keeps the z slice of every row, sub-lane 1 the w.angular slice, sub-lane 2 the w.linear
//     slice (sub-lane 3 idles there).  A row visit of the projected Gauss-Seidel is then 3 FMAs + one quad sum instead of a 9-term dot
//     product, its state update 3 FMAs instead of 9, and a lane keeps 6 numbers per row instead of 24.
// Per substep:
//   A. (lane, replicated in the quad) leg kinematics, leg link inertias in the common frame, velocity-product forces, the leg's
//              3x3 joint-space inertia A, its inverse, B = [Ic_i S_i], and the leg's Schur terms K = B A^-1 B^T, r = B A^-1 (tau - C)
//      (row)   sum {Ic_leg (10), K (21), F_leg (6), r (6)} over the 4 legs                     <- DPP row rotations
//   B. (lane, replicated) base articulated inertia IA = I_base + sum Ic - sum K, Phi = IA^-1, base and joint accelerations,
//              unconstrained end-of-substep velocities; each sub-lane keeps ITS 3 rows of the map row -> response ([A^-1 | 0], Phi rows)
//   C. (lane)  contact candidates — per leg the foot sphere and ONE candidate per body group: the deepest calf sphere, the deepest thigh
//              point, the deepest hip sphere, the deepest of the leg's share of the base / head points (so a base contact is never shadowed
//              by a leg link and thigh / calf report independently); the groups in contact are compacted into virtual slots 0..3
//              (go2_tables.h); joint-limit rows; per row the lane's slice of J = [Jc | G] and Y = [Z | H] (Z = A^-1 Jc^T, G = Ec + Jc N,
//              H = Phi G), the diagonal by a quad sum.  The rows of the foot and of virtual slot 0 stay in registers; further virtual
//              slots and the limit rows are parked in LDS and pass through ONE register-resident slot while they are swept
//      (row)   projected block iteration: all four legs sweep their own rows (Gauss-Seidel inside a leg) at once, each on its copy of the
//              base-twist slices with the base split n ways (mass splitting); one leg sum per slice and iteration commits the true responses
//   D. (lane, replicated) velocities, semi-implicit Euler integration, contact forces
#pragma once
#include "go2_math.h"
#include "go2_tables.h"
#include "go2_xlane.h"

#define GO2_QUAD_PARTIALS 43

// one constraint row as THIS sub-lane sees it: its 3-number slice of the row vector J = [Jc | G.ang | G.lin] and of the response
// Y = M^-1 J^T = [Z | H.ang | H.lin]; the scalars are replicated in the quad
struct Row { float J[3], Y[3], dinv /* until solve_prepare: this sub-lane's part J . Y of the diagonal */, vfb, lam; };

// The launch constants the substep loop reads, snapshotted ONCE per step.  Read straight from the device block they are "invariant scalar
// loads" to the compiler, which re-issues them at every use instead of keeping them in registers (37 s_load + wait per substep, measured
// in the ISA); passed through an empty asm statement each value becomes an ordinary register value (SGPR, wave-uniform).
struct HotL {
  float sim_dt, gravity[3], contact_offset, erp, max_depen_vel, bounce_thr, cfm, armature, limit_margin, max_lin_vel, max_ang_vel, action_scale;
  float hf_hscale, hf_vscale, hf_border, hf_nzmin; int32_t hf_trows, hf_tcols;
  int32_t terrain_mode, hf_walls, hf_rows, hf_cols, rand_strength, control_type;
#if defined(__HIP_DEVICE_COMPILE__)
  static __device__ __forceinline__ float keep(float x) { asm volatile("" : "+s"(x)); return x; }
  static __device__ __forceinline__ int32_t keep(int32_t x) { asm volatile("" : "+s"(x)); return x; }
#else
  static float keep(float x) { return x; }
  static int32_t keep(int32_t x) { return x; }
#endif
  GO2_HD void load(const Go2Launch& L) {
    sim_dt = keep(L.sim_dt); gravity[0] = keep(L.gravity[0]); gravity[1] = keep(L.gravity[1]); gravity[2] = keep(L.gravity[2]);
    contact_offset = keep(L.contact_offset); erp = keep(L.erp); max_depen_vel = keep(L.max_depen_vel); bounce_thr = keep(L.bounce_thr);
    cfm = keep(L.cfm); armature = keep(L.armature); limit_margin = keep(L.limit_margin); max_lin_vel = keep(L.max_lin_vel);
    max_ang_vel = keep(L.max_ang_vel); action_scale = keep(L.action_scale); hf_hscale = keep(L.hf_hscale); hf_vscale = keep(L.hf_vscale);
    hf_border = keep(L.hf_border); terrain_mode = keep(L.terrain_mode); hf_walls = keep(L.hf_walls); hf_rows = keep(L.hf_rows);
    hf_cols = keep(L.hf_cols); rand_strength = keep(L.rand_strength); control_type = keep(L.control_type);
    hf_nzmin = keep(L.hf_nzmin); hf_trows = keep(L.hf_trows); hf_tcols = keep(L.hf_tcols);
  }
};

// The per-leg constants the substep loop reads, copied ONCE per step from the LDS table into registers (the loop would otherwise wait on
// ~20 ds_reads per substep): joint origins, limits, the foot sphere.
struct LegLoop {
  float o1[3], o2[3], o3[3], eff_lim[3], foot_pt[4];
  GO2_HD void load(const LegTab& t, int sub) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { o1[k] = t.o1[k]; o2[k] = t.o2[k]; o3[k] = t.o3[k]; eff_lim[k] = t.eff_lim[k]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) foot_pt[k] = t.foot_pt[k];
    (void)sub;      // (the sub-lane's candidate spheres, the cull reach, the joint limits and rate limits are read from the LDS table where phase C / D use them: 51 registers that
                    //  would otherwise be live through the whole substep loop — the kernel is at the register file's limit)
  }
};

// one collision candidate as the tournaments carry it: gap, scan-order index (tie-break), facet normal (world), sphere centre in the base
// frame, radius, body index
struct Cand {
  float g; int i; V3 n, c; float r; int b;
  GO2_HD void clear() { g = 1e30f; i = 1 << 20; n = v3(0, 0, 1); c = v3(0, 0, 0); r = 0.f; b = 0; }
  // keep the deeper of the two; ties go to the lower scan-order index (the order of the sequential formulation)
  GO2_HD void take(bool ok, const Cand& o) {
    const bool tk = ok && (o.g < g || (o.g == g && o.i < i));
    g = tk ? o.g : g; i = tk ? o.i : i; n = sel(tk, o.n, n); c = sel(tk, o.c, c); r = tk ? o.r : r; b = tk ? o.b : b;
  }
};

struct LegPhys {
  int leg, sub;
  // ---- state (persistent over the substeps of one step) ----
  V3 pw, vw, ww; float qx, qy, qz, qw;
  float q[3], qd[3];
  RB Lhip, Lthigh, Lcalf, Ibase;
  float mu, rest;
  float lam_foot[3];
  // ---- substep temporaries that cross phase boundaries ----
  M3 Rwb, R1, R2, R3; V3 wb, vb, p1, p2, p3, a2;
  SV B1, B2, B3, T1, T2, T3, f0, V0, V0f;
  float Ainv[6];  // 11 12 13 22 23 33
  float u[3], qdf[3];
  float Msub[3][6];          // this sub-lane's rows of the response map: sub 0 [A^-1 | 0], sub 1 Phi rows 0..2, sub 2 Phi rows 3..5
  Row foot[3], v0[3], tmp[3];      // v0: virtual slot 0; tmp: the parked row set (virtual slots 1..3, joint limits) being built / swept
  float act_foot, act_v[GO2_NTYPE], act_lim[3];
  float lam_v[GO2_NTYPE - 1][3], lam_lim[3];      // impulses of the parked rows (replicated in the quad, like Row.lam)
  bool has_v[GO2_NTYPE];           // WAVE-UNIFORM: some leg of the wave has more than k body groups in contact (virtual slot k was built)
  int32_t body_v[GO2_NTYPE];       // body the contact of virtual slot k belongs to
  V3 v0_n;                         // world normal of virtual slot 0
  GO2_AS3 Go2RowsLds* rl; int tid, lid;      // the workgroup's row storage, this lane's index in it and its leg's (tid >> 2)
  float nsplit, yscale;      // solve_prepare: the split n of the base, and this sub-lane's factor on a response slice (1 joint slice, n base slices)
  float s_act;               // how firmly this leg's rows are active, in [0, 1] (solve_prepare: the smooth count of legs sharing the base): 0 at the activation boundary, 1 a quarter margin inside
  V3 f_n, f_t1, f_t2;     // world directions of the foot's contact frame
  float x[3];                // this sub-lane's slice of the velocity change: sub 0 z, sub 1 w.ang, sub 2 w.lin
  SV w; float z[3];          // the gathered velocity change (after the solve)
  float tau[3];
  V3 force_foot;

  GO2_HD float ainv(int i, int j) const {
    const int idx[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
    return Ainv[idx[i][j]];
  }

  // ------------------------------------------------------------------------------------------------
  template <class LT>
  GO2_HD void phaseA(const LegLoop& t, const LT& L, float* part) {
    Rwb = quat_to_m3(qx, qy, qz, qw);
    wb = mulT(Rwb, ww); vb = mulT(Rwb, vw);
    V3 gb = mulT(Rwb, v3(L.gravity[0], L.gravity[1], L.gravity[2]));
    float s1, c1, s2, c2, s3, c3;
    go2_sincos(q[0], &s1, &c1); go2_sincos(q[1], &s2, &c2); go2_sincos(q[2], &s3, &c3);
    float c23 = c2 * c3 - s2 * s3, s23 = s2 * c3 + c2 * s3;
    R1.x = v3(1, 0, 0); R1.y = v3(0, c1, s1); R1.z = v3(0, -s1, c1);
    R2.x = c2 * R1.x - s2 * R1.z; R2.y = R1.y; R2.z = s2 * R1.x + c2 * R1.z;
    R3.x = c23 * R1.x - s23 * R1.z; R3.y = R1.y; R3.z = s23 * R1.x + c23 * R1.z;
    p1 = v3(t.o1[0], t.o1[1], t.o1[2]);
    p2 = p1 + mul(R1, v3(t.o2[0], t.o2[1], t.o2[2]));
    p3 = p2 + mul(R2, v3(t.o3[0], t.o3[1], t.o3[2]));
    V3 a1 = v3(1, 0, 0); a2 = R1.y;
    SV S1 = sv(a1, cross(p1, a1)), S2 = sv(a2, cross(p2, a2)), S3 = sv(a2, cross(p3, a2));
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
  template <class LT>
  GO2_HD void phaseB(const LT& L, const float* red) {
    RB Ileg; Ileg.m = red[0]; Ileg.h = v3(red[1], red[2], red[3]);
    Ileg.J.xx = red[4]; Ileg.J.yy = red[5]; Ileg.J.zz = red[6]; Ileg.J.xy = red[7]; Ileg.J.xz = red[8]; Ileg.J.yz = red[9];
    RB Itot = Ibase + Ileg;
    float IA[21], Phi[21]; rb_to_s6(Itot, IA);
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
    // this sub-lane's three rows of the response map  [dz; dw] = [A^-1 Jc^T ; Phi G^T] dl
    const bool joint = sub == 0 || sub == 3, lin = sub == 2;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const float ph = lin ? s6at(Phi, 3 + i, k) : s6at(Phi, i, k);
        Msub[i][k] = joint ? (k < 3 ? ainv(i, k) : 0.f) : ph;
      }
  }

  // Contact of a sphere (world centre c, radius r) with the terrain: gap (< 0: penetration) and unit normal of the deepest of
  //   * the facet under the centre — the cell's two triangles, split along (i,j)-(i+1,j+1), the diagonal of the reference's trimesh
  //     (terrain_utils triangles (0,3,1),(0,2,3)); gap = (c.z - h) n.z - r;
  //   * with hf_walls (mesh_type 'trimesh'): the vertical faces on the cell's four edges — a face stands where the NEIGHBOUR cell's heights
  //     along the common edge exceed this cell's (go2sim.h hf_cells); closest point on the face (bottom .. top at the foot of the
  //     perpendicular) gives a horizontal normal beside the face and a slanted one over its top edge;
  //   * and the vertical edge at the nearest cell corner where the DIAGONAL neighbour cell is higher (round 6).
  template <class LT>
  GO2_HD void contact_query(const LT& L, const GO2_AS1 Go2Cell* cells, V3 c, float r, float* gap, V3* n) const {
    if (L.terrain_mode == 0) { *gap = c.z - r; *n = v3(0, 0, 1); return; }
    const float hs = L.hf_hscale, vs = L.hf_vscale;
    float fx = (c.x + L.hf_border) / hs, fy = (c.y + L.hf_border) / hs;
    int i = (int)floorf(fx), j = (int)floorf(fy);
    i = i < 0 ? 0 : (i > L.hf_rows - 2 ? L.hf_rows - 2 : i); j = j < 0 ? 0 : (j > L.hf_cols - 2 ? L.hf_cols - 2 : j);
    const float uu = fminf(fmaxf(fx - i, 0.f), 1.f), vv = fminf(fmaxf(fy - j, 0.f), 1.f);
    const int nc = L.hf_cols - 1;
    // (with hf_walls the record of a cell holds its neighbours' heights along the common edges and corners too — go2_tables.h Go2CellW: one 32-byte load per query;
    // rounds 2-5 fetched the four edge neighbours' cells beside the cell itself, five L2 requests per query)
    Go2CellW w;
    if (L.hf_walls) w = ((const GO2_AS1 Go2CellW*)cells)[i * nc + j];
    else { const Go2Cell q = cells[i * nc + j]; w.h[0] = q.h[0]; w.h[1] = q.h[1]; w.h[2] = q.h[2]; w.h[3] = q.h[3]; }
    const float h00 = w.h[0] * vs, h10 = w.h[1] * vs, h01 = w.h[2] * vs, h11 = w.h[3] * vs;
    float dx, dy;
    if (uu >= vv) { dx = h10 - h00; dy = h11 - h10; }
    else { dx = h11 - h01; dy = h01 - h00; }
    const float hgt = h00 + uu * dx + vv * dy;
    const float nx = -dx / hs, ny = -dy / hs, inv = 1.0f / sqrtf(nx * nx + ny * ny + 1.f);
    float g = (c.z - hgt) * inv - r; V3 nn = v3(nx * inv, ny * inv, inv);
    if (L.hf_walls) {
      // edge k: this cell's heights at the edge ends (a0, a1), the neighbour's (b0, b1), parameter along the edge, distance of the centre from the edge, inward direction
#define GO2_WALL(NB, A0, A1, T, D, NX, NY) { \
        const float bot = (A0) + (T) * ((A1) - (A0)), b0 = (NB)[0] * vs, top = b0 + (T) * ((NB)[1] * vs - b0); \
        if (top - bot > 0.5f * vs) { \
          const float qz = fminf(fmaxf(c.z, bot), top), dz = c.z - qz, dist = sqrtf((D) * (D) + dz * dz), gw = dist - r; \
          if (gw < g) { const float iv = 1.0f / fmaxf(dist, 1e-9f); g = gw; nn = dist > 1e-9f ? v3((NX) * (D) * iv, (NY) * (D) * iv, dz * iv) : v3((NX), (NY), 0.f); } } }
      GO2_WALL(w.xm, h00, h01, vv, uu * hs, 1.f, 0.f)
      GO2_WALL(w.xp, h10, h11, vv, (1.f - uu) * hs, -1.f, 0.f)
      GO2_WALL(w.ym, h00, h10, uu, vv * hs, 0.f, 1.f)
      GO2_WALL(w.yp, h01, h11, uu, (1.f - vv) * hs, 0.f, -1.f)
#undef GO2_WALL
      // the vertical EDGE at the nearest cell corner, where the diagonal neighbour stands higher there than this cell (outside corner of a stair ring
      // or a block): neither of the four faces above needs to exist for it.  Radii < hscale / 2: no other corner's edge can be penetrated from here.
      {
        const bool px = uu >= 0.5f, py = vv >= 0.5f;
        const float bot = px ? (py ? h11 : h10) : (py ? h01 : h00);
        const float top = (px ? (py ? w.dg[3] : w.dg[1]) : (py ? w.dg[2] : w.dg[0])) * vs;
        if (top - bot > 0.5f * vs) {
          const float ex = (px ? 1.f - uu : uu) * hs, ey = (py ? 1.f - vv : vv) * hs;
          const float qz = fminf(fmaxf(c.z, bot), top), dz = c.z - qz, dist = sqrtf(ex * ex + ey * ey + dz * dz), gw = dist - r;
          if (gw < g) { const float iv = 1.0f / fmaxf(dist, 1e-9f), sx = px ? -1.f : 1.f, sy = py ? -1.f : 1.f;
            g = gw; nn = dist > 1e-9f ? v3(sx * ex * iv, sy * ey * iv, dz * iv) : v3(sx * 0.70710678f, sy * 0.70710678f, 0.f); }
        }
      }
    }
    *gap = g; *n = nn;
  }

  // One constraint row from its joint-space row Jc (3) and its base-space row Ec (6): this sub-lane's slices of J = [Jc | G], Y = [Z | H]
  // (G = Ec - sum_j Jc_j T_j, Z = A^-1 Jc, H = Phi G), the inverse diagonal (quad sum of the slice products) and the free velocity.
#ifdef GO2_DBG_NOINLINE_C
  __device__ __attribute__((noinline)) void build_row(Row& r, const float* Jc, const SV& Ec, float cfm1, float bias, float lam0) {
#else
  GO2_HD void build_row(Row& r, const float* Jc, const SV& Ec, float cfm1, float bias, float lam0) {
#endif
    const SV G = Ec - (Jc[0] * T1 + Jc[1] * T2 + Jc[2] * T3);
    const bool joint = sub == 0 || sub == 3;
    const float x6[6] = {joint ? Jc[0] : G.a.x, joint ? Jc[1] : G.a.y, joint ? Jc[2] : G.a.z, joint ? 0.f : G.l.x, joint ? 0.f : G.l.y, joint ? 0.f : G.l.z};
    const float live = sub == 3 ? 0.f : 1.f;      // sub-lane 3 keeps an all-zero slice (it mirrors sub-lane 0's arithmetic, never its contribution)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 6; ++k) s += Msub[i][k] * x6[k];
      r.Y[i] = live * s;
    }
    r.J[0] = live * (sub == 2 ? G.l.x : x6[0]); r.J[1] = live * (sub == 2 ? G.l.y : x6[1]); r.J[2] = live * (sub == 2 ? G.l.z : x6[2]);
    r.dinv = (r.J[0] * r.Y[0] + r.J[1] * r.Y[1] + r.J[2] * r.Y[2]) * cfm1;      // (this sub-lane's part; summed, with the split base, in solve_prepare)
    r.vfb = dot(Ec, V0f) + Jc[0] * qdf[0] + Jc[1] * qdf[1] + Jc[2] * qdf[2] + bias;
    r.lam = lam0;
  }

  // the two tangents of a contact frame: world x made orthogonal to n (world y beside a face that looks along x), and n x t1
  GO2_HD static void tangents(V3 nw, V3* t1, V3* t2) {
    const V3 ex = fabsf(nw.x) < 0.9f ? v3(1, 0, 0) : v3(0, 1, 0); const float dnx = dot(ex, nw);
    V3 t = ex - dnx * nw; t = (1.0f / sqrtf(dot(t, t))) * t;
    *t1 = t; *t2 = cross(nw, t);
  }
  // the three rows (normal, two tangents) of a sphere contact: gap, sphere centre cb in the base frame, radius, link (0 base, 1..3), world normal
  template <class LT>
  GO2_HD void build_slot(Row* rows, float* active, V3* dn, V3* dt1, V3* dt2, const LT& L, float gap, V3 cb, float rad, int link, V3 nw, bool warm) {
    const float h = L.sim_dt, cfm1 = 1.0f + L.cfm;
    const float act = gap < L.contact_offset ? 1.f : 0.f;
    *active = act; *dn = nw;
    V3 t1; tangents(nw, &t1, dt2);
    *dt1 = t1;
    const V3 dirs[3] = {mulT(Rwb, nw), mulT(Rwb, t1), mulT(Rwb, *dt2)};
    const V3 rb = cb - rad * dirs[0];
    const V3 zero3 = v3(0, 0, 0);
    const V3 col1 = sel(link >= 1, cross(v3(1, 0, 0), rb - p1), zero3);
    const V3 col2 = sel(link >= 2, cross(a2, rb - p2), zero3);
    const V3 col3 = sel(link >= 3, cross(a2, rb - p3), zero3);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const V3 d = dirs[a];
      const float Jc[3] = {dot(d, col1), dot(d, col2), dot(d, col3)};
      const SV Ec = sv(cross(rb, d), d);
      float bias = 0.f;
      if (a == 0) {
        const float vn_pre = dot(Ec, V0) + Jc[0] * qd[0] + Jc[1] * qd[1] + Jc[2] * qd[2];
        float b = gap >= 0.f ? gap / h : gap * L.erp / h;
        b = fmaxf(b, -L.max_depen_vel);
        if (vn_pre < -L.bounce_thr && gap + vn_pre * h < 0.f) b = fminf(b, rest * vn_pre);
        bias = b;
      }
      build_row(rows[a], Jc, Ec, cfm1, bias, (warm && act > 0.f) ? lam_foot[a] : 0.f);
    }
  }

  // One body group T of the leg: the deepest candidate of the four sub-lanes (a 2-round quad tournament on gap and scan-order index
  // only; ties go to the lower index); the sub-lane that holds it leaves its data — gap, facet normal, centre in the base frame, radius,
  // body — in LDS for the whole leg.  Skipped (wave-uniform) when no lane of the wave has such a candidate inside the contact margin.
  // -> the group is in contact for this leg
  template <int T, class LT>
  GO2_HD bool group_winner(const LT& L, const Cand& b) {
    bool act = false;
    if (xl::any(b.g < L.contact_offset)) {
      float wg = b.g; int wi = b.i;
      { const float g2 = xl::quad_perm<1, 0, 3, 2>(wg); const int i2 = xl::quad_perm_i<1, 0, 3, 2>(wi); const bool tk = g2 < wg || (g2 == wg && i2 < wi); wg = tk ? g2 : wg; wi = tk ? i2 : wi; }
      { const float g2 = xl::quad_perm<2, 3, 0, 1>(wg); const int i2 = xl::quad_perm_i<2, 3, 0, 1>(wi); const bool tk = g2 < wg || (g2 == wg && i2 < wi); wg = tk ? g2 : wg; wi = tk ? i2 : wi; }
      act = wg < L.contact_offset;
      s_act = fmaxf(s_act, fminf(fmaxf((L.contact_offset - wg) / (0.25f * L.contact_offset), 0.f), 1.f));
      if (act && b.i == wi) {      // (candidate indices are unique within a leg: exactly one sub-lane)
        rl->win[T][0][lid] = b.g; rl->win[T][1][lid] = b.n.x; rl->win[T][2][lid] = b.n.y; rl->win[T][3][lid] = b.n.z;
        rl->win[T][4][lid] = b.c.x; rl->win[T][5][lid] = b.c.y; rl->win[T][6][lid] = b.c.z; rl->win[T][7][lid] = b.r; rl->win[T][8][lid] = (float)b.b;
      }
    }
    return act;
  }
  // rows of a slot on their way into a parked row set P / back into `tmp`
  template <int P>
  GO2_HD void park_rows() {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { rl->jy[P][a][k][tid] = tmp[a].J[k]; rl->jy[P][a][3 + k][tid] = tmp[a].Y[k]; }
      rl->dp[P][a][tid] = tmp[a].dinv;
      if (sub == 0) rl->sc[P][a][0][lid] = tmp[a].vfb;
    }
  }
  template <int P>
  GO2_HD void load_rows(const float* lam) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { tmp[a].J[k] = rl->jy[P][a][k][tid]; tmp[a].Y[k] = rl->jy[P][a][3 + k][tid]; }
      tmp[a].vfb = rl->sc[P][a][0][lid]; tmp[a].dinv = rl->sc[P][a][1][lid]; tmp[a].lam = lam[a];
    }
  }
  // Virtual slot K: the K-th body group (canonical order calf, thigh, hip, base share) that is in contact for this leg — none for a leg
  // with fewer; built wave-uniformly while some leg of the wave has more than K.
  template <int K, class LT>
  GO2_HD void build_virtual(const LT& L, const bool* act, const int* rank) {
    int Tk = -1;
#pragma unroll
    for (int T = GO2_NTYPE - 1; T >= 0; --T) Tk = (act[T] && rank[T] == K) ? T : Tk;
    const bool on = Tk >= 0; const int Ts = on ? Tk : 0;
    // (a group that was not in contact left nothing, or something stale, in its LDS cell: a leg without a K-th contact builds from a harmless
    //  dummy instead — far away, so inactive — and its rows, multiplied by act = 0, move nothing)
    const float g = on ? rl->win[Ts][0][lid] : 1e30f;
    const V3 n = on ? v3(rl->win[Ts][1][lid], rl->win[Ts][2][lid], rl->win[Ts][3][lid]) : v3(0, 0, 1);
    const V3 c = on ? v3(rl->win[Ts][4][lid], rl->win[Ts][5][lid], rl->win[Ts][6][lid]) : v3(0, 0, 0);
    const float r = on ? rl->win[Ts][7][lid] : 0.f;
    const int link = Tk == GO2_T_CALF ? 3 : (Tk == GO2_T_THIGH ? 2 : (Tk == GO2_T_HIP ? 1 : 0));
    body_v[K] = Tk == GO2_T_BASE ? (int)rl->win[GO2_T_BASE][8][lid] : (on ? 3 + 4 * leg + (2 - Tk) : -1);      // bodies of a leg: hip 3 + 4 leg, thigh, calf, foot (go2_model_data.h)
    V3 dn, dt1, dt2;
    if (K == 0) { build_slot(v0, &act_v[0], &dn, &dt1, &dt2, L, g, c, r, link, n, false); v0_n = dn; }
    else {
      build_slot(tmp, &act_v[K], &dn, &dt1, &dt2, L, g, c, r, link, n, false);
      park_rows<(K == 0 ? 0 : K - 1)>();
      if (sub == 0) { rl->nrm[K == 0 ? 0 : K - 1][0][lid] = dn.x; rl->nrm[K == 0 ? 0 : K - 1][1][lid] = dn.y; rl->nrm[K == 0 ? 0 : K - 1][2][lid] = dn.z; }
      lam_v[K == 0 ? 0 : K - 1][0] = lam_v[K == 0 ? 0 : K - 1][1] = lam_v[K == 0 ? 0 : K - 1][2] = 0.f;
    }
  }
  // net contact force of virtual slot K (world): impulses / dt along the slot's frame
  template <int K>
  GO2_HD V3 virtual_force(float h) const {
    V3 f = v3(0, 0, 0);
    if (has_v[K] && xl::any(act_v[K] > 0.f)) {
      const V3 n = K == 0 ? v0_n : v3(rl->nrm[K == 0 ? 0 : K - 1][0][lid], rl->nrm[K == 0 ? 0 : K - 1][1][lid], rl->nrm[K == 0 ? 0 : K - 1][2][lid]);
      V3 t1, t2; tangents(n, &t1, &t2);
      const float l0 = K == 0 ? v0[0].lam : lam_v[K == 0 ? 0 : K - 1][0], l1 = K == 0 ? v0[1].lam : lam_v[K == 0 ? 0 : K - 1][1], l2 = K == 0 ? v0[2].lam : lam_v[K == 0 ? 0 : K - 1][2];
      f = (act_v[K] / h) * (l0 * n + l1 * t1 + l2 * t2);
    }
    return f;
  }

  // ------------------------------------------------------------------------------------------------
  template <class LT>
  GO2_HD void phaseC(const LegLoop& t, const LegTab& tt, const LT& L, const GO2_AS1 Go2Cell* cells, const GO2_AS1 int16_t* top) {
    // foot
    {
      V3 cb = p3 + mul(R3, v3(t.foot_pt[0], t.foot_pt[1], t.foot_pt[2]));
      V3 cw = pw + mul(Rwb, cb); float gap; V3 n; contact_query(L, cells, cw, t.foot_pt[3], &gap, &n);
      build_slot(foot, &act_foot, &f_n, &f_t1, &f_t2, L, gap, cb, t.foot_pt[3], 3, n, true);
      s_act = fminf(fmaxf((L.contact_offset - gap) / (0.25f * L.contact_offset), 0.f), 1.f);
    }
    GO2_MARK(30);
    // The other body groups.  Each sub-lane tests its share of the leg's 22 non-foot points and one of the leg's share of the base / head
    // points (table slots with a fixed link type per slot, go2_tables.h SubCand: thigh x 3, calf x 2, hip, base), keeps the deepest PER
    // GROUP, and a 2-round quad tournament per group carries the group's deepest — with the base-frame position, radius, body and facet
    // normal its rows need — to all four sub-lanes.  Ties go to the lower candidate index (the scan order of the sequential formulation).
    {
      const SubCand& sc = tt.cand[sub];
      const V3 q2 = p2, q3 = p3;
      // A whole link group is skipped when none of its spheres can reach the contact margin in ANY lane of the wave.  On the plane: lowest
      // possible sphere bottom = (link origin height) - sum_axis |world-z component of the link axis| * (group's reach along that axis).
      // Conservative, so skipping cannot change which candidate is inside the margin (only those matter: an inactive slot has no rows).
      // On the height field the same test runs against the highest surface NEAR the leg (hf_top: a coarse map of the highest cell corner
      // within 0.8 m, which covers every wall top too): a sphere whose bottom is d above everything around it has a gap of at least
      // d * (smallest facet n_z of the map) to any facet and of at least d to any wall face.
      bool do_hip = true, do_thigh = true, do_calf = true, do_base = true;
      {
        const V3 gz = v3(Rwb.x.z, Rwb.y.z, Rwb.z.z);        // world z axis in base coordinates
        auto low = [&](const M3& R, V3 o, const float* ext) {
          return pw.z + dot(gz, o) - (fabsf(dot(gz, R.x)) * ext[0] + fabsf(dot(gz, R.y)) * ext[1] + fabsf(dot(gz, R.z)) * ext[2]); };
        const M3 Id = {v3(1, 0, 0), v3(0, 1, 0), v3(0, 0, 1)};
        float top_leg = 0.f, top_base = 0.f, slope = 1.f;
        if (L.terrain_mode != 0) {
          auto top_at = [&](V3 pt) {
            int bi = (int)floorf((pt.x + L.hf_border) / (L.hf_hscale * GO2_TOP_CELL)), bj = (int)floorf((pt.y + L.hf_border) / (L.hf_hscale * GO2_TOP_CELL));
            bi = bi < 0 ? 0 : (bi > L.hf_trows - 1 ? L.hf_trows - 1 : bi); bj = bj < 0 ? 0 : (bj > L.hf_tcols - 1 ? L.hf_tcols - 1 : bj);
            return (float)top[bi * L.hf_tcols + bj] * L.hf_vscale; };
          top_leg = top_at(pw + mul(Rwb, p1)); top_base = top_at(pw); slope = L.hf_nzmin;
        }
        do_hip = xl::any((low(R1, p1, tt.cull_ext[0]) - top_leg) * slope < L.contact_offset); do_thigh = xl::any((low(R2, p2, tt.cull_ext[1]) - top_leg) * slope < L.contact_offset);
        do_calf = xl::any((low(R3, p3, tt.cull_ext[2]) - top_leg) * slope < L.contact_offset); do_base = xl::any((low(Id, v3(0, 0, 0), tt.cull_ext[3]) - top_base) * slope < L.contact_offset);
      }
      auto eval = [&](int k, const M3& R, V3 P) {
        Cand o; o.c = P + mul(R, v3(sc.pt[k][0], sc.pt[k][1], sc.pt[k][2]));
        contact_query(L, cells, pw + mul(Rwb, o.c), sc.pt[k][3], &o.g, &o.n);
        o.i = sc.idx[k]; o.r = sc.pt[k][3]; o.b = sc.body[k];
        return o; };
      // group by group, so that only one or two candidates' data are live at a time: which groups are in contact for this leg, and the
      // winners' data into LDS
      bool act[GO2_NTYPE];
      {
        Cand thigh; thigh.clear();
        if (do_thigh) { thigh.take(sc.idx[0] >= 0, eval(0, R2, q2)); thigh.take(sc.idx[1] >= 0, eval(1, R2, q2)); thigh.take(sc.idx[2] >= 0, eval(2, R2, q2)); }
        act[GO2_T_THIGH] = group_winner<GO2_T_THIGH>(L, thigh);
      }
      GO2_LOAD_FENCE();
      {
        Cand calf; calf.clear();
        if (do_calf) { calf.take(sc.idx[3] >= 0, eval(3, R3, q3)); calf.take(sc.idx[4] >= 0, eval(4, R3, q3)); }
        act[GO2_T_CALF] = group_winner<GO2_T_CALF>(L, calf);
      }
      GO2_LOAD_FENCE();
      {
        Cand hip; hip.clear();
        if (do_hip) hip.take(sc.idx[GO2_SC_HIP] >= 0, eval(GO2_SC_HIP, R1, p1));      // (the two hip spheres: sub-lanes 2 and 3)
        act[GO2_T_HIP] = group_winner<GO2_T_HIP>(L, hip);
      }
      GO2_LOAD_FENCE();
      {
        Cand base; base.clear();
        if (do_base) {   // one of the leg's base / head points (sub-lane < number of points of this leg)
          const M3 Id = {v3(1, 0, 0), v3(0, 1, 0), v3(0, 0, 1)};
          base.take(sc.idx[GO2_SC_BASE] >= 0, eval(GO2_SC_BASE, Id, v3(0, 0, 0)));
        }
        act[GO2_T_BASE] = group_winner<GO2_T_BASE>(L, base);
      }
      GO2_MARK(31);
      // compaction: the K-th group in contact (canonical order) becomes virtual slot K; a virtual slot is built — and later swept — only
      // while some leg of the wave has that many contacts (wave-uniform; an inactive slot's rows move nothing, so skipping is exact)
      const int rank[GO2_NTYPE] = {0, act[0] ? 1 : 0, (act[0] ? 1 : 0) + (act[1] ? 1 : 0), (act[0] ? 1 : 0) + (act[1] ? 1 : 0) + (act[2] ? 1 : 0)};
      const int count = rank[3] + (act[3] ? 1 : 0);
      has_v[0] = xl::any(count > 0); has_v[1] = has_v[0] && xl::any(count > 1); has_v[2] = has_v[1] && xl::any(count > 2); has_v[3] = has_v[2] && xl::any(count > 3);
      _Pragma("unroll") for (int k = 0; k < GO2_NTYPE; ++k) { act_v[k] = 0.f; body_v[k] = -1; }
      // (the parked slots first, virtual slot 0 — whose rows then stay in registers — after the limit rows: while a parked row set is
      //  under construction in `tmp`, only the foot's rows are live beside it)
      if (has_v[0]) {
        xl::row_sync();       // the winners' LDS cells were written by one sub-lane of the leg each
        if (has_v[1]) build_virtual<1>(L, act, rank);
        if (has_v[2]) build_virtual<2>(L, act, rank);
        if (has_v[3]) build_virtual<3>(L, act, rank);
      }
      GO2_MARK(32);
    // joint limits (rare: their rows are parked in LDS)
    {
      float h = L.sim_dt, cfm1 = 1.0f + L.cfm;
      float sgn[3], gap[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        float glo = q[j] - tt.lim_lo[j], ghi = tt.lim_hi[j] - q[j];
        sgn[j] = 0.f; gap[j] = 0.f;
        if (glo < L.limit_margin) { sgn[j] = 1.f; gap[j] = glo; } else if (ghi < L.limit_margin) { sgn[j] = -1.f; gap[j] = ghi; }
        act_lim[j] = sgn[j] != 0.f ? 1.f : 0.f; lam_lim[j] = 0.f;
        if (sgn[j] != 0.f) s_act = fmaxf(s_act, fminf(fmaxf((L.limit_margin - gap[j]) / (0.25f * L.limit_margin), 0.f), 1.f));
      }
      if (xl::any(act_lim[0] + act_lim[1] + act_lim[2] > 0.f)) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const float Jc[3] = {j == 0 ? sgn[j] : 0.f, j == 1 ? sgn[j] : 0.f, j == 2 ? sgn[j] : 0.f};
          float b = gap[j] >= 0.f ? gap[j] / h : gap[j] * L.erp / h; b = fmaxf(b, -10.0f);
          build_row(tmp[j], Jc, sv(v3(0, 0, 0), v3(0, 0, 0)), cfm1, b, 0.f);
        }
        park_rows<GO2_PARK_LIMITS>();
      }
    }
      GO2_MARK(33);
      if (has_v[0]) build_virtual<0>(L, act, rank);
    }
    GO2_MARK(34);
    // warm start: the slice of the velocity change the remembered foot impulses produce.  The joint slice (sub-lane 0) is the leg's
    // own; the base-twist slices are summed over the four legs.
    float c[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 3; ++a) { c[0] += foot[a].Y[0] * foot[a].lam; c[1] += foot[a].Y[1] * foot[a].lam; c[2] += foot[a].Y[2] * foot[a].lam; }
#pragma unroll
    for (int k = 0; k < 3; ++k) { const float s = xl::leg_sum(c[k]); x[k] = sub == 0 ? c[k] : s; }
  }
  GO2_HD bool has_foot() const { return act_foot > 0.f; }
  template <int K> GO2_HD bool has_virtual() const { return act_v[K] > 0.f; }
  GO2_HD bool has_limit() const { return (act_lim[0] + act_lim[1] + act_lim[2]) > 0.f; }

  GO2_HD float row_v(const Row& r) const { return r.vfb + xl::sub_sum(r.J[0] * x[0] + r.J[1] * x[1] + r.J[2] * x[2]); }
  GO2_HD void row_apply(const Row& r, float dl) { const float d = dl * yscale; x[0] += r.Y[0] * d; x[1] += r.Y[1] * d; x[2] += r.Y[2] * d; }      // (the leg's view: base slices x n)
  GO2_HD void sweep_slot(Row* r, float m, float mu_) {
    {
      const float v = row_v(r[0]);
      const float ln = fmaxf(0.f, r[0].lam - v * r[0].dinv);
      const float dl = m * (ln - r[0].lam); r[0].lam += dl;
      row_apply(r[0], dl);
    }
    {
      const float v1 = row_v(r[1]), v2 = row_v(r[2]);
      float l1 = r[1].lam - v1 * r[1].dinv, l2 = r[2].lam - v2 * r[2].dinv;
      // Coulomb cone by radial projection, branch-free: scale = min(1, mu lam_n / |l|)
      const float lim_ = mu_ * r[0].lam, nn2 = l1 * l1 + l2 * l2;
      const float sc = fminf(1.f, lim_ * go2_rsqrt(fmaxf(nn2, 1e-30f)));
      l1 *= sc; l2 *= sc;
      const float d1 = m * (l1 - r[1].lam), d2 = m * (l2 - r[2].lam); r[1].lam += d1; r[2].lam += d2;
      row_apply(r[1], d1); row_apply(r[2], d2);
    }
  }
  // The contact / limit solve: projected block iteration over the LEGS with mass splitting at the base.  The rows of a
  // leg are a block, visited in the fixed order foot, then the body groups in contact — calf, thigh, hip, base share — (n, t each), limits (Gauss-Seidel inside the block); ALL FOUR LEGS sweep their
  // blocks at once, each from the same state, on its own copy of the base-twist slices.  Legs interact only through the base, and each leg is
  // given 1 / n of it: in ITS view the base slices of every response (sub-lanes 1, 2: H = Phi G) are n times larger, the joint slice (sub-lane
  // 0: Z = A^-1 Jc, the leg's own joints with the base held fixed) is as it is.  Committed are the true responses: z as swept, w <- w0 +
  // (1 / n) sum_legs (w_leg - w0) — one leg sum per slice instead of one per leg turn.  n = the number of legs with active rows, counted
  // smoothly (s_act), so the step stays a continuous function of the state.
  // solve_prepare: n, and the rows' inverse diagonals J . Y_view (1 + cfm) with it.
  // parked row set P: the rows' inverse diagonals (quad sum of the sub-lanes' parts with the split base), left in LDS for the sweeps
  template <int P>
  GO2_HD void prepare_parked() {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float d_ = xl::sub_sum(rl->dp[P][a][tid] * yscale), di = d_ > 0.f ? 1.0f / d_ : 0.f;
      if (sub == 0) rl->sc[P][a][1][lid] = di;
    }
  }
  // do_v[K]: virtual slot K is active in some lane of the wave (implies has_v[K]: its rows were built); likewise do_lim
  GO2_HD void solve_prepare(bool do_foot, const bool* do_v, bool do_lim) {
    nsplit = fmaxf(xl::leg_sum(s_act), 1.f);
    yscale = sub == 0 ? 1.f : nsplit;
    auto fin = [&](Row& r) { const float d_ = xl::sub_sum(r.dinv * yscale); r.dinv = d_ > 0.f ? 1.0f / d_ : 0.f; };
    if (do_foot) { fin(foot[0]); fin(foot[1]); fin(foot[2]); }
    if (do_v[0]) { fin(v0[0]); fin(v0[1]); fin(v0[2]); }
    if (do_v[1]) prepare_parked<0>();
    if (do_v[2]) prepare_parked<1>();
    if (do_v[3]) prepare_parked<2>();
    if (do_lim) prepare_parked<GO2_PARK_LIMITS>();
    // what sub-lane 0 of a leg parked in LDS (free velocities in phase C, inverse diagonals here) is read by the leg's other sub-lanes
    if (do_v[1] || do_lim) xl::row_sync();
  }
  template <int K>
  GO2_HD void sweep_parked() {      // virtual slot K >= 1
    load_rows<K - 1>(lam_v[K - 1]);
    sweep_slot(tmp, act_v[K], mu);
    lam_v[K - 1][0] = tmp[0].lam; lam_v[K - 1][1] = tmp[1].lam; lam_v[K - 1][2] = tmp[2].lam;
  }
  // do_* are WAVE-UNIFORM hints: false means no lane of the wave has such a row active this substep, so the group is skipped as a whole
  // (an inactive row moves nothing).
#ifdef GO2_DBG_NOINLINE_GS
  __device__ __attribute__((noinline)) void solve_iteration(bool do_foot, const bool* do_v, bool do_lim) {
#else
  GO2_HD void solve_iteration(bool do_foot, const bool* do_v, bool do_lim) {
#endif
    const float x0[3] = {x[0], x[1], x[2]};
    if (do_foot) sweep_slot(foot, act_foot, mu);
    if (do_v[0]) sweep_slot(v0, act_v[0], mu);
    if (do_v[1]) sweep_parked<1>();
    if (do_v[2]) sweep_parked<2>();
    if (do_v[3]) sweep_parked<3>();
    if (do_lim) {
      load_rows<GO2_PARK_LIMITS>(lam_lim);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float v = row_v(tmp[j]);
        const float ln = fmaxf(0.f, tmp[j].lam - v * tmp[j].dinv);
        const float dl = act_lim[j] * (ln - tmp[j].lam); tmp[j].lam += dl;
        row_apply(tmp[j], dl);
        lam_lim[j] = tmp[j].lam;
      }
    }
    const float inv_n = 1.f / nsplit;
#pragma unroll
    for (int k = 0; k < 3; ++k) { const float s = xl::leg_sum(x[k] - x0[k]); x[k] = sub == 0 ? x[k] : x0[k] + inv_n * s; }
  }
  // after the last turn: every lane gets the whole velocity change of its leg and of the base
  GO2_HD void gather_solution() {
#pragma unroll
    for (int k = 0; k < 3; ++k) z[k] = xl::sub_bcast<0>(x[k]);
    w.a = v3(xl::sub_bcast<1>(x[0]), xl::sub_bcast<1>(x[1]), xl::sub_bcast<1>(x[2]));
    w.l = v3(xl::sub_bcast<2>(x[0]), xl::sub_bcast<2>(x[1]), xl::sub_bcast<2>(x[2]));
  }

  // ------------------------------------------------------------------------------------------------
  template <class LT>
  GO2_HD void phaseD(const LegTab& t, const LT& L) {
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
    if (th > 1e-9f) { float sh, ch; go2_sincos(0.5f * th, &sh, &ch); float sc = sh / (th / h); dx = ww.x * sc; dy = ww.y * sc; dz = ww.z * sc; dwq = ch; }
    else { dx = ww.x * h * 0.5f; dy = ww.y * h * 0.5f; dz = ww.z * h * 0.5f; dwq = 1.f; }
    float nx = dwq * qx + dx * qw + dy * qz - dz * qy, ny = dwq * qy - dx * qz + dy * qw + dz * qx;
    float nz = dwq * qz + dx * qy - dy * qx + dz * qw, nw_ = dwq * qw - dx * qx - dy * qy - dz * qz;
    float inv = 1.0f / sqrtf(nx * nx + ny * ny + nz * nz + nw_ * nw_);
    qx = nx * inv; qy = ny * inv; qz = nz * inv; qw = nw_ * inv;
#pragma unroll
    for (int j = 0; j < 3; ++j) { qd[j] = qdp[j]; q[j] += h * qdp[j]; }
    float ih = 1.0f / h;
    force_foot = act_foot * ih * (foot[0].lam * f_n + foot[1].lam * f_t1 + foot[2].lam * f_t2);
#pragma unroll
    for (int a = 0; a < 3; ++a) lam_foot[a] = act_foot * foot[a].lam;
  }

  // _compute_torques (legged_robot.py:594-618, control_type 'P') then *= motor_strengths (:80-81)
  // kp/kd arrive already multiplied by the per-env gain multipliers, q0 = default angle of this leg's joints (hoisted out of the substep loop)
  template <class LT>
  // last_dv: buffers.last_dof_vel (the rates at the end of the previous policy step, legged_robot.py:141), read only by control type 'V'
  GO2_HD void pd(const LegLoop& t, const LT& L, const float* act, const float* kp_, const float* kd_, const float* q0, const float* off, const float* strength,
                 const GO2_AS1 float* last_dv, int N, int e) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float kp = kp_[j], kd = kd_[j];
      float tq = kp * (act[j] * L.action_scale + q0[j] - q[j] + off[j]) - kd * qd[j];
      if (L.control_type != 0) {   // :612-615 (no go2 task; cold)
        if (L.control_type == 1) tq = kp * (act[j] * L.action_scale - qd[j]) - kd * (qd[j] - last_dv[(size_t)(3 * leg + j) * N + e]) / L.sim_dt;
        else tq = act[j] * L.action_scale;
      }
      tq = fminf(fmaxf(tq, -t.eff_lim[j]), t.eff_lim[j]);
      if (L.rand_strength) tq *= strength[j];
      tau[j] = tq;
    }
  }
};
