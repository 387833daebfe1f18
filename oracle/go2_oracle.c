/* go2_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C CPU restatement of the hot path of wty-yy/go2_rl_gym behind the same C ABI as the HIP
 * library (include/go2sim.h).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it; the product (go2_rl_gym_amd/) never does.
 *
 * What it restates, and how it is pinned:
 *   - the per-env torch logic of the reference (legged_gym/envs/base/legged_robot.py and
 *     legged_gym/envs/go2/go2_env.py; every function below cites the lines it follows).  PINNED by the
 *     golden fixtures under tests/golden/, generated in the build container by importing the reference
 *     with an in-memory isaacgym stub (oracle/gen_golden.py).
 *   - RolloutStorage.compute_returns (rsl_rl/rsl_rl/storage/rollout_storage.py:123-137).  PINNED the
 *     same way.
 *   - the element-wise heads of the PPO / CTS rollout and update that the product runs as library kernels: the fused loss head
 *     (go2sim_ppo_loss: PINNED against torch autograd of the reference's formulation and by the reference's own PPO.update /
 *     CTS.update goldens), PPO.act's sampling head, process_env_step, the CTS history ring, ELU-backward + bias gradient.
 *   - the rigid-body physics.  The reference delegates it to NVIDIA Isaac Gym / PhysX (setup.py:12
 *     `isaacgym`, unpinned, closed source, absent from /root/reference), so there is nothing to restate:
 *     this file states a NEW model in textbook form — Featherstone's
 *     floating-base ABA in link coordinates (RBDA tables 9.4/9.6), dense CRBA + Cholesky for the
 *     contact problem, projected Gauss-Seidel on velocity-level contact/limit rows.  PARITY UNPINNED
 *     against PhysX; pinned only by known-answer tests (tests/test_physics_kat.py).  The HIP kernels
 *     use a different formulation (base-frame single-frame spatial algebra, per-leg block elimination)
 *     of the same model, so HIP-vs-oracle agreement checks two independent derivations.
 *
 * Build: see oracle/Makefile (fp32: libgo2oracle_f32.so, fp64: libgo2oracle_f64.so with -DGO2O_F64,
 * which widens every `float` of the ABI to double).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef GO2O_F64
#define float double
typedef double R;
#define RC(x) (x)
#define SQRT sqrt
#define SIN sin
#define COS cos
#define ATAN2 atan2
#define ASIN asin
#define EXP exp
#define FABS fabs
#define FLOOR floor
#define FMOD fmod
#else
typedef float R;
#define RC(x) (x##f)
#define SQRT sqrtf
#define SIN sinf
#define COS cosf
#define ATAN2 atan2f
#define ASIN asinf
#define EXP expf
#define FABS fabsf
#define FLOOR floorf
#define FMOD fmodf
#endif

#include "../include/go2sim.h"
#include "../include/go2sim_defaults.h"
#include "../include/go2_model_data.h"

#define NB GO2_NUM_BODIES
#define NL 13 /* moving links: base + 12 */
#define NV 18 /* generalized velocity: base spatial (w,v) in base coords + 12 joints */
#define NROWS_MAX (4 * 9)

static char g_err[256] = "";
static const int    kBodyLink[NB] = GO2_BODY_LINK_INIT;
static const double kBodyOffset[NB][3] = GO2_BODY_OFFSET_INIT;
static const double kBodyMass[NB] = GO2_BODY_MASS_INIT;
static const double kBodyCom[NB][3] = GO2_BODY_COM_INIT;
static const double kBodyInertia[NB][6] = GO2_BODY_INERTIA_INIT;
static const int    kJointParent[12] = GO2_JOINT_PARENT_INIT;
static const double kJointOrigin[12][3] = GO2_JOINT_ORIGIN_INIT;
static const int    kJointAxis[12] = GO2_JOINT_AXIS_INIT;
static const double kJointLower[12] = GO2_JOINT_LOWER_INIT;
static const double kJointUpper[12] = GO2_JOINT_UPPER_INIT;
static const double kJointEffort[12] = GO2_JOINT_EFFORT_INIT;
static const double kJointVelocity[12] = GO2_JOINT_VELOCITY_INIT;
typedef struct { int body, link; double c[3]; double r; } Sph;
static const Sph kFootPts[4] = GO2_FOOT_PTS_INIT;
static const Sph kLegOther[4][GO2_LEG_OTHER_PTS] = GO2_LEG_OTHER_PTS_INIT;
static const Sph kBasePts[GO2_BASE_PTS] = GO2_BASE_PTS_INIT;

/* ------------------------------------------------------------------------------------------------
 * Philox4x32-10 (Salmon et al., SC'11), the counter-based generator both libraries use.
 * ---------------------------------------------------------------------------------------------- */
static void philox4x32_10(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4]) {
  uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
  uint32_t k0 = key_in[0], k1 = key_in[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
#define GO2_STEP_INIT 0xFFFFFFFFFFFFFFFFull
#define GO2_ENV_GLOBAL 0xFFFFFFFFu
static float u01(uint64_t seed, uint32_t env_global, uint32_t slot, uint64_t step) {
  uint32_t ctr[4] = {env_global, slot >> 2, (uint32_t)step, (uint32_t)(step >> 32)};
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t o[4];
  philox4x32_10(ctr, key, o);
  return (float)((o[slot & 3] >> 8) * (1.0 / 16777216.0));
}
/* per-step uniforms: slot -> (group, word) by the shared contract table (include/go2sim_rng.h) */
#include "../include/go2sim_rng.h"
static float u01_step(uint64_t seed, uint32_t env_global, uint32_t slot, uint64_t step) {
  static uint8_t code[GO2_NUM_UNIFORMS]; static int init = 0;
  if (!init) { go2_fill_slot_codes(code); init = 1; }
  return u01(seed, env_global, (uint32_t)code[slot], step);   /* code = group*4 + word: u01 splits it as >>2, &3 */
}

/* ------------------------------------------------------------------------------------------------
 * small linear algebra
 * ---------------------------------------------------------------------------------------------- */
static inline void cross3(const R* a, const R* b, R* o) { R x = a[1]*b[2]-a[2]*b[1], y = a[2]*b[0]-a[0]*b[2], z = a[0]*b[1]-a[1]*b[0]; o[0]=x; o[1]=y; o[2]=z; }
static inline R dot3(const R* a, const R* b) { return a[0]*b[0]+a[1]*b[1]+a[2]*b[2]; }
static inline void mat3_vec(const R* M, const R* v, R* o) { R x=M[0]*v[0]+M[1]*v[1]+M[2]*v[2], y=M[3]*v[0]+M[4]*v[1]+M[5]*v[2], z=M[6]*v[0]+M[7]*v[1]+M[8]*v[2]; o[0]=x;o[1]=y;o[2]=z; }
static inline void mat3T_vec(const R* M, const R* v, R* o) { R x=M[0]*v[0]+M[3]*v[1]+M[6]*v[2], y=M[1]*v[0]+M[4]*v[1]+M[7]*v[2], z=M[2]*v[0]+M[5]*v[1]+M[8]*v[2]; o[0]=x;o[1]=y;o[2]=z; }
static inline void mat3_mul(const R* A, const R* B, R* O) { R T[9]; for (int i=0;i<3;++i) for (int j=0;j<3;++j) T[3*i+j]=A[3*i]*B[j]+A[3*i+1]*B[3+j]+A[3*i+2]*B[6+j]; memcpy(O,T,sizeof(T)); }
static void quat_to_mat(const R* q, R* M) { /* (x,y,z,w) -> world_from_body rotation */
  R x=q[0],y=q[1],z=q[2],w=q[3];
  M[0]=1-2*(y*y+z*z); M[1]=2*(x*y-z*w);   M[2]=2*(x*z+y*w);
  M[3]=2*(x*y+z*w);   M[4]=1-2*(x*x+z*z); M[5]=2*(y*z-x*w);
  M[6]=2*(x*z-y*w);   M[7]=2*(y*z+x*w);   M[8]=1-2*(x*x+y*y);
}
static void mat_to_quat(const R* M, R* q) {
  R tr = M[0]+M[4]+M[8];
  if (tr > 0) { R s = SQRT(tr+1)*2; q[3]=s/4; q[0]=(M[7]-M[5])/s; q[1]=(M[2]-M[6])/s; q[2]=(M[3]-M[1])/s; }
  else if (M[0]>M[4] && M[0]>M[8]) { R s=SQRT(1+M[0]-M[4]-M[8])*2; q[3]=(M[7]-M[5])/s; q[0]=s/4; q[1]=(M[1]+M[3])/s; q[2]=(M[2]+M[6])/s; }
  else if (M[4]>M[8]) { R s=SQRT(1+M[4]-M[0]-M[8])*2; q[3]=(M[2]-M[6])/s; q[0]=(M[1]+M[3])/s; q[1]=s/4; q[2]=(M[5]+M[7])/s; }
  else { R s=SQRT(1+M[8]-M[0]-M[4])*2; q[3]=(M[3]-M[1])/s; q[0]=(M[2]+M[6])/s; q[1]=(M[5]+M[7])/s; q[2]=s/4; }
}
/* isaacgym.torch_utils.quat_rotate_inverse: v(2w^2-1) - 2w (q x v) + 2 q (q.v) */
static void quat_rotate_inverse(const R* q, const R* v, R* o) {
  R w=q[3]; R c[3]; cross3(q, v, c); R d = dot3(q, v); R a = 2*w*w-1;
  for (int i=0;i<3;++i) o[i] = v[i]*a - 2*w*c[i] + 2*q[i]*d;
}
/* isaacgym.torch_utils.quat_apply: v + 2w (q x v) + 2 q x (q x v) */
static void quat_apply(const R* q, const R* v, R* o) {
  R c[3], cc[3]; cross3(q, v, c); cross3(q, c, cc);
  for (int i=0;i<3;++i) o[i] = v[i] + 2*q[3]*c[i] + 2*cc[i];
}
/* Cholesky solve of SPD A (n x n, row-major, destroyed) for nrhs right-hand sides B (n x nrhs, row-major). */
static int chol_factor(R* A, int n) {
  for (int j=0;j<n;++j) {
    R d = A[j*n+j]; for (int k=0;k<j;++k) d -= A[j*n+k]*A[j*n+k];
    if (!(d > 0)) return -1;
    d = SQRT(d); A[j*n+j] = d;
    for (int i=j+1;i<n;++i) { R s = A[i*n+j]; for (int k=0;k<j;++k) s -= A[i*n+k]*A[j*n+k]; A[i*n+j] = s/d; }
  }
  return 0;
}
static void chol_solve(const R* L, int n, R* b) {
  for (int i=0;i<n;++i) { R s=b[i]; for (int k=0;k<i;++k) s -= L[i*n+k]*b[k]; b[i]=s/L[i*n+i]; }
  for (int i=n-1;i>=0;--i) { R s=b[i]; for (int k=i+1;k<n;++k) s -= L[k*n+i]*b[k]; b[i]=s/L[i*n+i]; }
}

/* ------------------------------------------------------------------------------------------------
 * spatial algebra in link coordinates (Featherstone, RBDA ch. 2): motion vectors (w; v), force (n; f)
 * Plücker transform X (parent -> child coords): E = child_from_parent rotation, r = child origin in parent.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { R E[9]; R r[3]; } Xf;
static void xf_motion(const Xf* X, const R* v, R* o) { /* o = X v */
  R t[3]; cross3(X->r, v, t); R lin[3] = {v[3]-t[0], v[4]-t[1], v[5]-t[2]};
  R a[3], b[3]; mat3_vec(X->E, v, a); mat3_vec(X->E, lin, b);
  o[0]=a[0];o[1]=a[1];o[2]=a[2];o[3]=b[0];o[4]=b[1];o[5]=b[2];
}
static void xf_force_T(const Xf* X, const R* f, R* o) { /* o = X^T f  (child force -> parent coords) */
  R n[3], l[3], t[3]; mat3T_vec(X->E, f, n); mat3T_vec(X->E, f+3, l); cross3(X->r, l, t);
  o[0]=n[0]+t[0]; o[1]=n[1]+t[1]; o[2]=n[2]+t[2]; o[3]=l[0]; o[4]=l[1]; o[5]=l[2];
}
static void xf_to_mat6(const Xf* X, R* M) { /* 6x6 motion transform */
  R rx[9] = {0,-X->r[2],X->r[1], X->r[2],0,-X->r[0], -X->r[1],X->r[0],0};
  R Erx[9]; mat3_mul(X->E, rx, Erx);
  for (int i=0;i<3;++i) for (int j=0;j<3;++j) { M[6*i+j]=X->E[3*i+j]; M[6*i+3+j]=0; M[6*(3+i)+j]=-Erx[3*i+j]; M[6*(3+i)+3+j]=X->E[3*i+j]; }
}
static void mat6_XtIX(const R* X, const R* I, R* O) { /* O += X^T I X */
  R T[36];
  for (int i=0;i<6;++i) for (int j=0;j<6;++j) { R s=0; for (int k=0;k<6;++k) s += I[6*i+k]*X[6*k+j]; T[6*i+j]=s; }
  for (int i=0;i<6;++i) for (int j=0;j<6;++j) { R s=0; for (int k=0;k<6;++k) s += X[6*k+i]*T[6*k+j]; O[6*i+j]+=s; }
}
static void mat6_vec(const R* M, const R* v, R* o) { R t[6]; for (int i=0;i<6;++i){R s=0; for(int k=0;k<6;++k) s+=M[6*i+k]*v[k]; t[i]=s;} memcpy(o,t,sizeof(t)); }
static void crm(const R* v, const R* m, R* o) { /* v x m (motion) */
  R a[3], b[3], c[3]; cross3(v, m, a); cross3(v, m+3, b); cross3(v+3, m, c);
  o[0]=a[0];o[1]=a[1];o[2]=a[2];o[3]=b[0]+c[0];o[4]=b[1]+c[1];o[5]=b[2]+c[2];
}
static void crf(const R* v, const R* f, R* o) { /* v x* f (force) */
  R a[3], b[3], c[3]; cross3(v, f, a); cross3(v+3, f+3, b); cross3(v, f+3, c);
  o[0]=a[0]+b[0];o[1]=a[1]+b[1];o[2]=a[2]+b[2];o[3]=c[0];o[4]=c[1];o[5]=c[2];
}
static void spatial_inertia(R m, const R* c, const R* Ic6, R* I) { /* about the frame origin, 6x6 */
  R cx[9] = {0,-c[2],c[1], c[2],0,-c[0], -c[1],c[0],0};
  R Ic[9] = {Ic6[0],Ic6[3],Ic6[4], Ic6[3],Ic6[1],Ic6[5], Ic6[4],Ic6[5],Ic6[2]};
  for (int i=0;i<3;++i) for (int j=0;j<3;++j) {
    R cc=0; for (int k=0;k<3;++k) cc += cx[3*i+k]*cx[3*j+k]; /* cx cx^T */
    I[6*i+j] += Ic[3*i+j] + m*cc;
    I[6*i+3+j] += m*cx[3*i+j];
    I[6*(3+i)+j] += m*cx[3*j+i];
    I[6*(3+i)+3+j] += (i==j)? m : 0;
  }
}

/* ------------------------------------------------------------------------------------------------
 * simulator object
 * ---------------------------------------------------------------------------------------------- */
struct Go2Sim {
  Go2SimCfg cfg;
  Go2SimBuffers b;
  int N;
  int16_t* hf; int16_t* cells; int walls; float* terrain_origins; int32_t* terrain_type_id;
  int32_t* terrain_kind;   /* [N] terrain kind 0..8 of each env, -1 on a plane */
  const float* injected;   /* uniforms for the next step, or NULL */
  float* inj_storage;
  uint64_t step_count;     /* Philox step index */
  int64_t common_step_counter;
  /* host scalars the reference keeps in Python */
  R dt;                    /* policy dt = decimation * sim_dt (legged_robot.py:1094) */
  R max_episode_length;    /* ceil(episode_length_s / dt) (:1104) */
  R reward_scale_dt[GO2_NUM_REWARDS];  /* scale * dt (:914-920) */
  R turn_over_scale_dt[GO2_NUM_REWARDS]; /* turn_over_scales * dt (:922-923) */
  R reward_curr_scale[GO2_NUM_REWARDS];/* reward_curriculum_scales (:52-57), 1 if none */
  int reward_has_curr[GO2_NUM_REWARDS];
  R cmd_ranges[4][2]; R max_lin_vel; R zero_command_proba;
  int cmd_curr_done[4];
  double cmd_x_track[2]; int cmd_stage_seen, cb_resamples;
  int yaw_seen;            /* this pass: the heading clip (:411-419) still sees the ranges of cmd_stage_seen (see go2sim_post_physics) */   /* command_ranges['lin_vel_x'] under update_command_curriculum (:728-737) */
  R dof_pos_limits[12][2]; /* soft limits (:372-375) */
  R noise_vec[GO2_NUM_OBS];
  R friction_buckets[64];
  int height_mask[GO2_NUM_HEIGHT_POINTS]; int num_height_mask;
  R height_pts[GO2_NUM_HEIGHT_POINTS][2];
  /* episode info accumulators */
  double ep_sum[GO2_NUM_REWARDS]; int ep_count;
  int timing; double time_ms; int64_t time_launches;
};

/* ---------------- per-env inertial model (legged_robot.py:379-402 + recomputeInertia=True) ------ */
static void build_link_inertias(const Go2Sim* s, int e, R I[NL][36]) {
  memset(I, 0, sizeof(R)*NL*36);
  for (int b = 0; b < NB; ++b) {
    R m, c[3], Ic[6];
    R ratio;
    if (b == 0) {
      m = (R)kBodyMass[0] + (R)s->b.added_base_mass[e];
      ratio = m / (R)kBodyMass[0];
      for (int k=0;k<3;++k) c[k] = (R)kBodyCom[0][k] + (R)s->b.added_base_com[3*e+k];
    } else {
      ratio = (R)s->b.link_mass_ratio[18*e + b - 1];
      m = (R)kBodyMass[b] * ratio;
      for (int k=0;k<3;++k) c[k] = (R)kBodyCom[b][k];
    }
    for (int k=0;k<6;++k) Ic[k] = (R)kBodyInertia[b][k] * ratio;
    for (int k=0;k<3;++k) c[k] += (R)kBodyOffset[b][k];
    spatial_inertia(m, c, Ic, I[kBodyLink[b]]);
  }
}

/* kinematic quantities of one env */
typedef struct {
  R Rw[NL][9]; R pw[NL][3];   /* world pose of each link frame */
  Xf Xup[NL];                 /* parent -> link (index 1..12) */
  R I[NL][36];
  R v[NL][6];                 /* spatial velocity in link coords */
} Kin;

static void joint_rot(int axis, R q, R* M) { /* rotation of the child frame relative to the parent */
  R c = COS(q), s = SIN(q);
  if (axis == 0) { R T[9] = {1,0,0, 0,c,-s, 0,s,c}; memcpy(M,T,sizeof(T)); }
  else { R T[9] = {c,0,s, 0,1,0, -s,0,c}; memcpy(M,T,sizeof(T)); }
}
static void kinematics(const Go2Sim* s, int e, const R* root, const R* q, const R* qd, Kin* k) {
  quat_to_mat(root+3, k->Rw[0]);
  for (int i=0;i<3;++i) k->pw[0][i] = root[i];
  /* base spatial velocity in base coords */
  mat3T_vec(k->Rw[0], root+10, k->v[0]); mat3T_vec(k->Rw[0], root+7, k->v[0]+3);
  build_link_inertias(s, e, k->I);
  for (int j=0;j<12;++j) {
    int l = j+1, p = kJointParent[j];
    R Rq[9]; joint_rot(kJointAxis[j], q[j], Rq);
    for (int a=0;a<3;++a) for (int c=0;c<3;++c) k->Xup[l].E[3*a+c] = Rq[3*c+a]; /* E = Rq^T */
    for (int a=0;a<3;++a) k->Xup[l].r[a] = (R)kJointOrigin[j][a];
    mat3_mul(k->Rw[p], Rq, k->Rw[l]);
    R t[3]; mat3_vec(k->Rw[p], k->Xup[l].r, t);
    for (int a=0;a<3;++a) k->pw[l][a] = k->pw[p][a] + t[a];
    xf_motion(&k->Xup[l], k->v[p], k->v[l]);
    k->v[l][kJointAxis[j]] += qd[j];
  }
}

/* RNEA with zero joint/base acceleration: bias C (18) incl. gravity (RBDA table 5.1 with a_0 = -a_g). */
static void bias_forces(const Go2Sim* s, const Kin* k, const R* qd, R* C) {
  R a[NL][6], f[NL][6];
  R g[3] = {(R)s->cfg.gravity[0], (R)s->cfg.gravity[1], (R)s->cfg.gravity[2]}, gb[3];
  mat3T_vec(k->Rw[0], g, gb);
  for (int i=0;i<3;++i) { a[0][i]=0; a[0][3+i] = -gb[i]; }
  for (int l=0;l<NL;++l) {
    if (l>0) {
      int j=l-1, p=kJointParent[j];
      xf_motion(&k->Xup[l], a[p], a[l]);
      R sq[6]={0,0,0,0,0,0}, c[6]; sq[kJointAxis[j]] = qd[j]; crm(k->v[l], sq, c);
      for (int i=0;i<6;++i) a[l][i] += c[i];
    }
    R Iv[6], Ia[6], vf[6]; mat6_vec(k->I[l], k->v[l], Iv); mat6_vec(k->I[l], a[l], Ia); crf(k->v[l], Iv, vf);
    for (int i=0;i<6;++i) f[l][i] = Ia[i] + vf[i];
  }
  for (int l=NL-1;l>0;--l) {
    int j=l-1, p=kJointParent[j];
    C[6+j] = f[l][kJointAxis[j]];
    R t[6]; xf_force_T(&k->Xup[l], f[l], t);
    for (int i=0;i<6;++i) f[p][i] += t[i];
  }
  for (int i=0;i<6;++i) C[i] = f[0][i];
}
/* CRBA: joint-space inertia M (18x18) (RBDA table 9.5 style). */
static void mass_matrix(const Go2Sim* s, const Kin* k, R* M) {
  R Ic[NL][36]; memcpy(Ic, k->I, sizeof(Ic));
  memset(M, 0, sizeof(R)*NV*NV);
  for (int l=NL-1;l>0;--l) {
    int p = kJointParent[l-1]; R X6[36]; xf_to_mat6(&k->Xup[l], X6);
    mat6_XtIX(X6, Ic[l], Ic[p]);
  }
  for (int i=0;i<6;++i) for (int j=0;j<6;++j) M[NV*i+j] = Ic[0][6*i+j];
  for (int l=1;l<NL;++l) {
    int j=l-1; R F[6]; for (int i=0;i<6;++i) F[i] = Ic[l][6*i+kJointAxis[j]];
    M[NV*(6+j)+(6+j)] = F[kJointAxis[j]] + (R)s->cfg.joint_armature;
    int c = l;
    while (1) {
      int p = kJointParent[c-1]; R t[6]; xf_force_T(&k->Xup[c], F, t); memcpy(F,t,sizeof(t));
      c = p; if (c==0) break;
      int jj=c-1; M[NV*(6+j)+(6+jj)] = M[NV*(6+jj)+(6+j)] = F[kJointAxis[jj]];
    }
    for (int i=0;i<6;++i) M[NV*i+(6+j)] = M[NV*(6+j)+i] = F[i];
  }
}
/* Featherstone's articulated-body algorithm, floating base (RBDA table 9.4).  acc = (a_0 (6), qdd (12)),
 * a_0 = spatial acceleration of the base in base coordinates, gravity included. */
static void aba(const Go2Sim* s, const Kin* k, const R* qd, const R* tau, R* acc) {
  R IA[NL][36], pA[NL][6], c[NL][6], U[NL][6], D[NL], u[NL], a[NL][6];
  memcpy(IA, k->I, sizeof(IA));
  for (int l=0;l<NL;++l) {
    R Iv[6]; mat6_vec(k->I[l], k->v[l], Iv); crf(k->v[l], Iv, pA[l]);
    if (l>0) { R sq[6]={0,0,0,0,0,0}; sq[kJointAxis[l-1]] = qd[l-1]; crm(k->v[l], sq, c[l]); }
  }
  for (int l=NL-1;l>0;--l) {
    int j=l-1, ax=kJointAxis[j], p=kJointParent[j];
    for (int i=0;i<6;++i) U[l][i] = IA[l][6*i+ax];
    D[l] = U[l][ax] + (R)s->cfg.joint_armature;
    u[l] = tau[j] - pA[l][ax];
    R Ia[36], pa[6];
    for (int i=0;i<6;++i) for (int m=0;m<6;++m) Ia[6*i+m] = IA[l][6*i+m] - U[l][i]*U[l][m]/D[l];
    R Iac[6]; mat6_vec(Ia, c[l], Iac);
    for (int i=0;i<6;++i) pa[i] = pA[l][i] + Iac[i] + U[l][i]*u[l]/D[l];
    R X6[36]; xf_to_mat6(&k->Xup[l], X6); mat6_XtIX(X6, Ia, IA[p]);
    R t[6]; xf_force_T(&k->Xup[l], pa, t); for (int i=0;i<6;++i) pA[p][i] += t[i];
  }
  { R L[36]; memcpy(L, IA[0], sizeof(L)); R rhs[6]; for (int i=0;i<6;++i) rhs[i] = -pA[0][i];
    chol_factor(L, 6); chol_solve(L, 6, rhs); memcpy(a[0], rhs, sizeof(rhs)); }
  for (int l=1;l<NL;++l) {
    int j=l-1, ax=kJointAxis[j], p=kJointParent[j];
    R ap[6]; xf_motion(&k->Xup[l], a[p], ap); for (int i=0;i<6;++i) ap[i] += c[l][i];
    R Ua=0; for (int i=0;i<6;++i) Ua += U[l][i]*ap[i];
    R qdd = (u[l] - Ua)/D[l]; acc[6+j] = qdd;
    memcpy(a[l], ap, sizeof(ap)); a[l][ax] += qdd;
  }
  R g[3] = {(R)s->cfg.gravity[0], (R)s->cfg.gravity[1], (R)s->cfg.gravity[2]}, gb[3];
  mat3T_vec(k->Rw[0], g, gb);
  for (int i=0;i<3;++i) { acc[i] = a[0][i]; acc[3+i] = a[0][3+i] + gb[i]; }
}

/* ---------------- terrain ------------------------------------------------------------------------ */
/* Contact of a sphere (world centre c, radius r) with the terrain -> gap (< 0 = penetration) and unit normal n.  The surface is given
 * cell by cell (s->cells: heights at the corners (i,j) (i+1,j) (i,j+1) (i+1,j+1) as seen from inside the cell, go2sim.h hf_cells); the
 * contact is the deepest of the facet under the centre (two triangles per cell split along (i,j)-(i+1,j+1), the diagonal
 * isaacgym.terrain_utils.convert_heightfield_to_trimesh uses) and, with hf_walls, the vertical faces that stand on the cell's edges where
 * the neighbouring cell's edge is higher, and the vertical edge at the nearest cell corner where the DIAGONAL neighbour is higher (mesh_type
 * 'trimesh' with slope_treshold, legged_robot.py:1127-1141). */
static R clampR(R x, R lo, R hi) { return x < lo ? lo : (x > hi ? hi : x); }
static void wall_face(const Go2Sim* s, int inside, int ni, int nj, R a0, R a1, int k0, int k1, R t, R d, R nx, R ny, const R* c, R r, R* g, R* n) {
  if (!inside) return;
  R vs = (R)s->cfg.hf_vscale; int nc = s->cfg.hf_cols - 1;
  const int16_t* b = s->cells + ((size_t)ni*nc + nj)*4;
  R bot = a0 + t*(a1-a0), b0 = b[k0]*vs, top = b0 + t*(b[k1]*vs - b0);
  if (!(top - bot > RC(0.5)*vs)) return;
  R qz = clampR(c[2], bot, top), dz = c[2]-qz, dist = SQRT(d*d + dz*dz), gw = dist - r;
  if (gw < *g) { *g = gw; if (dist > RC(1e-9)) { n[0] = nx*d/dist; n[1] = ny*d/dist; n[2] = dz/dist; } else { n[0]=nx; n[1]=ny; n[2]=0; } }
}
static void contact_query(const Go2Sim* s, const R* c, R r, R* gap, R* n) {
  if (s->cfg.terrain_mode == 0) { *gap = c[2] - r; n[0]=0; n[1]=0; n[2]=1; return; }
  R hs = (R)s->cfg.hf_hscale, vs = (R)s->cfg.hf_vscale;
  R fx = (c[0] + (R)s->cfg.hf_border)/hs, fy = (c[1] + (R)s->cfg.hf_border)/hs;
  int rows = s->cfg.hf_rows, cols = s->cfg.hf_cols, nc = cols-1;
  int i = (int)FLOOR(fx), j = (int)FLOOR(fy);
  if (i<0) i=0; if (i>rows-2) i=rows-2; if (j<0) j=0; if (j>cols-2) j=cols-2;
  R u = clampR(fx - i, 0, 1), v = clampR(fy - j, 0, 1);
  const int16_t* q = s->cells + ((size_t)i*nc + j)*4;
  R h00 = q[0]*vs, h10 = q[1]*vs, h01 = q[2]*vs, h11 = q[3]*vs;
  R dx, dy;
  if (u >= v) { dx = h10-h00; dy = h11-h10; }
  else { dx = h11-h01; dy = h01-h00; }
  R h = h00 + u*dx + v*dy;
  R nx = -dx/hs, ny = -dy/hs, inv = 1/SQRT(nx*nx+ny*ny+1);
  n[0]=nx*inv; n[1]=ny*inv; n[2]=inv;
  R g = (c[2]-h)*inv - r;
  if (s->walls) {
    wall_face(s, i > 0,      i-1, j,   h00, h01, 1, 3, v, u*hs,     1, 0, c, r, &g, n);
    wall_face(s, i < rows-2, i+1, j,   h10, h11, 0, 2, v, (1-u)*hs, -1, 0, c, r, &g, n);
    wall_face(s, j > 0,      i,   j-1, h00, h10, 2, 3, u, v*hs,     0, 1, c, r, &g, n);
    wall_face(s, j < cols-2, i,   j+1, h01, h11, 0, 1, u, (1-v)*hs, 0, -1, c, r, &g, n);
    /* the vertical EDGE at the cell corner nearest to the centre: where the diagonal-neighbour cell stands higher at that corner than this
     * cell (the outside corner of a stair ring or of a block, terrain.py:44-49 -> legged_robot.py:1127-1141), the segment from this cell's
     * height to the diagonal cell's belongs to the mesh although neither of this cell's four faces needs to exist.  Radii are < hscale / 2,
     * so a sphere centred in this cell cannot penetrate the edge at any other corner. */
    int di = u >= RC(0.5) ? 1 : -1, dj = v >= RC(0.5) ? 1 : -1, ii = i+di, jj = j+dj;
    if (ii >= 0 && ii <= rows-2 && jj >= 0 && jj <= cols-2) {
      int kc = (di > 0 ? 1 : 0) + (dj > 0 ? 2 : 0);
      R bot = q[kc]*vs, top = s->cells[((size_t)ii*nc + jj)*4 + (3-kc)]*vs;
      if (top - bot > RC(0.5)*vs) {
        R ex = (di > 0 ? 1-u : u)*hs, ey = (dj > 0 ? 1-v : v)*hs;
        R qz = clampR(c[2], bot, top), dz = c[2]-qz, dist = SQRT(ex*ex + ey*ey + dz*dz), gw = dist - r;
        if (gw < g) { g = gw; if (dist > RC(1e-9)) { n[0] = -di*ex/dist; n[1] = -dj*ey/dist; n[2] = dz/dist; } else { n[0]=-di*RC(0.70710678118654752); n[1]=-dj*RC(0.70710678118654752); n[2]=0; } }
      }
    }
  }
  *gap = g;
}

/* ---------------- one physics substep ------------------------------------------------------------ */
typedef struct {
  int active; int kind;          /* kind 0 = contact triple head (n), handled with its two tangents; 1 = limit */
  R J[3][NV]; R Y[3][NV]; R d[3]; R b; R mu; R lam[3]; R sact;   /* sact: how firmly the row is active, 0 at its activation threshold .. 1 */
  int body; R n[3], t1[3], t2[3];
} Row;

#define NSLOT 5        /* contact slots per leg: foot, calf, thigh, hip, base share */
#define NR (NSLOT+3)   /* + the leg's three joint-limit rows */
/* candidate -> world position of the sphere centre */
static void sphere_world(const Kin* k, const Sph* p, R* c) {
  R l[3] = {(R)p->c[0], (R)p->c[1], (R)p->c[2]}, t[3]; mat3_vec(k->Rw[p->link], l, t);
  for (int i=0;i<3;++i) c[i] = k->pw[p->link][i] + t[i];
}
/* 3x18 Jacobian of the world velocity of world point pw attached to link l, in (base spatial, joints). */
static void point_jacobian(const Kin* k, int l, const R* pw, R Jw[3][NV]) {
  memset(Jw, 0, sizeof(R)*3*NV);
  /* point in link-l coords */
  R d[3] = {pw[0]-k->pw[l][0], pw[1]-k->pw[l][1], pw[2]-k->pw[l][2]}, rl[3]; mat3T_vec(k->Rw[l], d, rl);
  /* chain: columns = X_{l<-c} S_c for each ancestor joint c, and X_{l<-0} for the base */
  R X[36]; for (int i=0;i<36;++i) X[i] = (i%7==0)?1:0;   /* X_{l<-c}, starts as identity (c = l) */
  int c = l;
  while (1) {
    if (c > 0) {
      int j=c-1, ax=kJointAxis[j];
      R col[6]; for (int i=0;i<6;++i) col[i] = X[6*i+ax];
      R vl[3], t[3]; cross3(col, rl, t); for (int i=0;i<3;++i) vl[i] = col[3+i] + t[i];
      R vw[3]; mat3_vec(k->Rw[l], vl, vw); for (int i=0;i<3;++i) Jw[i][6+j] = vw[i];
      R X6[36], T[36]; xf_to_mat6(&k->Xup[c], X6);
      for (int i=0;i<6;++i) for (int m=0;m<6;++m) { R sacc=0; for (int n=0;n<6;++n) sacc += X[6*i+n]*X6[6*n+m]; T[6*i+m]=sacc; }
      memcpy(X, T, sizeof(T));
      c = kJointParent[j];
    } else {
      for (int m=0;m<6;++m) {
        R col[6]; for (int i=0;i<6;++i) col[i] = X[6*i+m];
        R vl[3], t[3]; cross3(col, rl, t); for (int i=0;i<3;++i) vl[i] = col[3+i] + t[i];
        R vw[3]; mat3_vec(k->Rw[l], vl, vw); for (int i=0;i<3;++i) Jw[i][m] = vw[i];
      }
      break;
    }
  }
}

static void pd_torques(const Go2Sim* s, int e, const R* q, const R* qd, const R* act_in, R* tau) {
  /* _compute_torques (legged_robot.py:594-618) then *= motor_strengths (:80-81) */
  const Go2SimBuffers* b = &s->b;
  for (int j=0;j<12;++j) {
    R kp = (R)s->cfg.kp[j]*(R)b->p_gains_multiplier[12*e+j], kd = (R)s->cfg.kd[j]*(R)b->d_gains_multiplier[12*e+j];
    R as = act_in[j]*(R)s->cfg.action_scale, t;
    if (s->cfg.control_type == 1) t = kp*(as - qd[j]) - kd*(qd[j] - (R)b->last_dof_vel[12*e+j])/(R)s->cfg.sim_dt;      /* 'V' :612-613 */
    else if (s->cfg.control_type == 2) t = as;                                                                          /* 'T' :614-615 */
    else t = kp*(as + (R)s->cfg.default_dof_pos[j] - q[j] + (R)b->motor_zero_offsets[12*e+j]) - kd*qd[j];               /* 'P' :610-611 */
    R lim = (R)kJointEffort[j];
    if (t > lim) t = lim; if (t < -lim) t = -lim;
    if (s->cfg.randomize_motor_strength) t *= (R)b->motor_strengths[12*e+j];
    tau[j] = t;
  }
}

static void physics_substep(Go2Sim* s, int e, R* root, R* q, R* qd, const R* tau, R* body_force /*[NB][3]*/, Kin* kout, R (*glam)[NSLOT][3] /* [4]: impulses of the previous substep per leg and body group, or NULL */) {
  const Go2SimCfg* cfg = &s->cfg;
  R h = (R)cfg->sim_dt;
  Kin k; kinematics(s, e, root, q, qd, &k);
  R acc[NV]; aba(s, &k, qd, tau, acc);
  R nu[NV]; for (int i=0;i<6;++i) nu[i] = k.v[0][i]; for (int j=0;j<12;++j) nu[6+j] = qd[j];
  R nu_free[NV]; for (int i=0;i<NV;++i) nu_free[i] = nu[i] + h*acc[i];
  /* dense inverse inertia for the constraint rows */
  R M[NV*NV]; mass_matrix(s, &k, M);
  R Mll[4][3][3];   /* joint-space inertia of each leg with the base held fixed = the leg's diagonal block of M */
  for (int l=0;l<4;++l) for (int i=0;i<3;++i) for (int j=0;j<3;++j) Mll[l][i][j] = M[(6+3*l+i)*NV + 6+3*l+j];
  chol_factor(M, NV);

  R mu = RC(0.5)*((R)cfg->terrain_friction + (R)s->b.friction_coeffs[e]);
  R rest = RC(0.5)*((R)cfg->terrain_restitution + (R)s->b.restitution_coeffs[e]);

  /* rows: per lane (leg) NSLOT contact slots [n,t1,t2] + [limit x3].  The contact set a leg reports: one contact per BODY
   * GROUP — slot 0 the foot sphere, slot 1 the deepest calf sphere, slot 2 the deepest thigh point, slot 3 the deepest hip sphere, slot 4 the
   * deepest of the leg's share of the base / head points — so that a base contact is never shadowed by a leg link (check_termination reads
   * the base force, legged_robot.py:170-173) and thigh and calf report independently (_reward_collision counts 8 bodies, :1277-1279). */
  static _Thread_local Row rows[4][NR];
  for (int lane=0; lane<4; ++lane) {
    for (int slot=0; slot<NSLOT; ++slot) {
      Row* r = &rows[lane][slot]; r->active = 0; r->kind = 0; r->lam[0]=r->lam[1]=r->lam[2]=0;
      const Sph* best = NULL; R best_gap = 0, best_c[3], best_n[3];
      static const int slot_link[NSLOT] = {0, 3, 2, 1, 0};      /* leg-local link of the slot's candidates: calf 3, thigh 2, hip 1 */
      int ncand = slot==0 ? 1 : (slot==NSLOT-1 ? GO2_BASE_PTS : GO2_LEG_OTHER_PTS);
      for (int ci=0; ci<ncand; ++ci) {
        const Sph* p;
        if (slot==0) p = &kFootPts[lane];
        else if (slot==NSLOT-1) { if ((ci & 3) != lane) continue; p = &kBasePts[ci]; }
        else { p = &kLegOther[lane][ci]; if (p->link - 3*lane != slot_link[slot]) continue; }
        R c[3], gap, n[3]; sphere_world(&k, p, c); contact_query(s, c, (R)p->r, &gap, n);
        if (!best || gap < best_gap) { best = p; best_gap = gap; memcpy(best_c,c,sizeof(c)); memcpy(best_n,n,sizeof(n)); }
      }
      if (best && best_gap < (R)cfg->contact_offset) {
        r->active = 1; r->body = best->body; r->mu = mu;
        { R sa = ((R)cfg->contact_offset - best_gap)/(RC(0.25)*(R)cfg->contact_offset); r->sact = sa < 0 ? 0 : (sa > 1 ? 1 : sa); }
        memcpy(r->n, best_n, sizeof(best_n));
        R ex[3] = {1,0,0}; if (!(FABS(r->n[0]) < RC(0.9))) { ex[0]=0; ex[1]=1; }   /* world x, or world y beside a face that looks along x */
        R dn = dot3(ex, r->n); for (int i=0;i<3;++i) r->t1[i] = ex[i]-dn*r->n[i];
        R inv = 1/SQRT(dot3(r->t1,r->t1)); for (int i=0;i<3;++i) r->t1[i]*=inv; cross3(r->n, r->t1, r->t2);
        R pw[3]; for (int i=0;i<3;++i) pw[i] = best_c[i] - (R)best->r*r->n[i];
        R Jw[3][NV]; point_jacobian(&k, best->link, pw, Jw);
        const R* dirs[3] = {r->n, r->t1, r->t2};
        for (int a=0;a<3;++a) for (int c=0;c<NV;++c) r->J[a][c] = dirs[a][0]*Jw[0][c]+dirs[a][1]*Jw[1][c]+dirs[a][2]*Jw[2][c];
        /* bias: gap closing / depenetration / restitution */
        R vn_pre=0; for (int c=0;c<NV;++c) vn_pre += r->J[0][c]*nu[c];
        R b = best_gap >= 0 ? best_gap/h : best_gap*(R)cfg->erp/h;
        if (b < -(R)cfg->max_depenetration_velocity) b = -(R)cfg->max_depenetration_velocity;
        if (vn_pre < -(R)cfg->bounce_threshold_velocity && best_gap + vn_pre*h < 0) { R br = rest*vn_pre; if (br < b) b = br; }
        r->b = b;
        if (slot==0) for (int a=0;a<3;++a) r->lam[a] = (R)s->b.foot_impulse[(4*e+lane)*3+a];
        else if (glam) for (int a=0;a<3;++a) r->lam[a] = glam[lane][slot][a];
      } else if (slot==0) { for (int a=0;a<3;++a) s->b.foot_impulse[(4*e+lane)*3+a] = 0; }
      else if (glam) { for (int a=0;a<3;++a) glam[lane][slot][a] = 0; }
    }
    for (int jj=0;jj<3;++jj) {
      Row* r = &rows[lane][NSLOT+jj]; int j = 3*lane+jj; r->active=0; r->kind=1; r->lam[0]=r->lam[1]=r->lam[2]=0;
      R glo = q[j]-(R)kJointLower[j], ghi = (R)kJointUpper[j]-q[j];
      R sgn = 0, gap = 0;
      if (glo < (R)cfg->joint_limit_margin) { sgn = 1; gap = glo; } else if (ghi < (R)cfg->joint_limit_margin) { sgn = -1; gap = ghi; }
      if (sgn != 0) {
        r->active=1; memset(r->J, 0, sizeof(r->J)); r->J[0][6+j] = sgn;
        { R sa = ((R)cfg->joint_limit_margin - gap)/(RC(0.25)*(R)cfg->joint_limit_margin); r->sact = sa < 0 ? 0 : (sa > 1 ? 1 : sa); }
        R b = gap >= 0 ? gap/h : gap*(R)cfg->erp/h; if (b < -RC(10.0)) b = -RC(10.0);
        r->b = b;
      }
    }
  }
  /* Y = M^-1 J^T, diagonal, warm start */
  R dnu[NV]; memset(dnu, 0, sizeof(dnu));
  for (int lane=0;lane<4;++lane) for (int ri=0;ri<NR;++ri) {
    Row* r = &rows[lane][ri]; if (!r->active) continue;
    int nr = r->kind==0 ? 3 : 1;
    for (int a=0;a<nr;++a) {
      memcpy(r->Y[a], r->J[a], sizeof(R)*NV); chol_solve(M, NV, r->Y[a]);
      R w=0; for (int c=0;c<NV;++c) w += r->J[a][c]*r->Y[a][c];
      r->d[a] = w*(1+(R)cfg->contact_cfm);
      for (int c=0;c<NV;++c) dnu[c] += r->Y[a][c]*r->lam[a];
    }
  }
  /* Projected block iteration over the LEGS with MASS SPLITTING at the base (Tonge et al. 2012).  The constraint rows of a
   * leg form a block (foot n,t; calf n,t; thigh n,t; hip n,t; base share n,t; limits, visited in that order, Gauss-Seidel inside the block).  Per iteration every leg sweeps its
   * block starting from the SAME state; the legs interact only through the base, and each leg is given 1/n of it: in ITS view the base-mediated
   * part of every response, Y - Y_L, is n times larger, the fixed-base part Y_L = M_ll^-1 J_l^T (the leg's own joints) is as it is.  What is
   * committed are the TRUE responses Y dlam of the impulses the legs arrive at.  n = the number of legs with active rows, counted smoothly
   * (a row that has only just become active — within a quarter margin of its threshold, where it still does nothing — counts in
   * proportion), so that the step is a continuous function of the state. */
  int legact[4]; R nsm = 0;
  for (int lane=0;lane<4;++lane) { legact[lane]=0; R sl=0; for (int ri=0;ri<NR;++ri) if (rows[lane][ri].active) { legact[lane]=1; if (rows[lane][ri].sact > sl) sl = rows[lane][ri].sact; } nsm += sl; }
  const R nsplit = nsm > 1 ? nsm : 1;
  static _Thread_local R Yv[4][NR][3][NV], dv[4][NR][3];      /* the leg's view: responses and diagonals with the split base */
  for (int lane=0;lane<4;++lane) for (int ri=0;ri<NR;++ri) {
    Row* r=&rows[lane][ri]; if (!r->active) continue;
    int nr = r->kind==0 ? 3 : 1;
    for (int a=0;a<nr;++a) {
      R A[3][3], bb[3], y[3]; memcpy(A, Mll[lane], sizeof(A)); for (int i=0;i<3;++i) bb[i] = r->J[a][6+3*lane+i];
      R det = A[0][0]*(A[1][1]*A[2][2]-A[1][2]*A[2][1]) - A[0][1]*(A[1][0]*A[2][2]-A[1][2]*A[2][0]) + A[0][2]*(A[1][0]*A[2][1]-A[1][1]*A[2][0]);
      for (int c=0;c<3;++c) { R B_[3][3]; memcpy(B_, A, sizeof(A)); for (int i=0;i<3;++i) B_[i][c]=bb[i];      /* Cramer */
        y[c] = (B_[0][0]*(B_[1][1]*B_[2][2]-B_[1][2]*B_[2][1]) - B_[0][1]*(B_[1][0]*B_[2][2]-B_[1][2]*B_[2][0]) + B_[0][2]*(B_[1][0]*B_[2][1]-B_[1][1]*B_[2][0]))/det; }
      for (int c=0;c<NV;++c) { R yl = (c>=6+3*lane && c<9+3*lane) ? y[c-6-3*lane] : 0; Yv[lane][ri][a][c] = yl + nsplit*(r->Y[a][c]-yl); }
      R w=0; for (int c=0;c<NV;++c) w += r->J[a][c]*Yv[lane][ri][a][c];
      dv[lane][ri][a] = w*(1+(R)cfg->contact_cfm);
    }
  }
  for (int it=0; it<cfg->solver_iterations; ++it) {
    R dnu0[NV], acc[NV]; memcpy(dnu0, dnu, sizeof(dnu)); memset(acc, 0, sizeof(acc));
    for (int lane=0;lane<4;++lane) {
      if (!legact[lane]) continue;
      R dl_[NV]; memcpy(dl_, dnu0, sizeof(dnu0));      /* this leg's view of the velocity change */
      for (int ri=0;ri<NR;++ri) {
        Row* r = &rows[lane][ri]; if (!r->active) continue;
        R lam0[3] = {r->lam[0], r->lam[1], r->lam[2]};
        { R v=0; for (int c=0;c<NV;++c) v += r->J[0][c]*(nu_free[c]+dl_[c]);
          R ln = r->lam[0] - (v + r->b)/dv[lane][ri][0]; if (ln < 0) ln = 0;
          R dl = ln - r->lam[0]; r->lam[0] = ln; for (int c=0;c<NV;++c) dl_[c] += Yv[lane][ri][0][c]*dl; }
        if (r->kind==0) {
          R v1=0, v2=0; for (int c=0;c<NV;++c) { v1 += r->J[1][c]*(nu_free[c]+dl_[c]); v2 += r->J[2][c]*(nu_free[c]+dl_[c]); }
          R l1 = r->lam[1] - v1/dv[lane][ri][1], l2 = r->lam[2] - v2/dv[lane][ri][2];
          R lim = r->mu*r->lam[0], nn = SQRT(l1*l1+l2*l2);
          if (nn > lim) { R sc = nn > 0 ? lim/nn : 0; l1*=sc; l2*=sc; }
          R d1 = l1-r->lam[1], d2 = l2-r->lam[2]; r->lam[1]=l1; r->lam[2]=l2;
          for (int c=0;c<NV;++c) dl_[c] += Yv[lane][ri][1][c]*d1 + Yv[lane][ri][2][c]*d2;
        }
        int nr = r->kind==0 ? 3 : 1;
        for (int a=0;a<nr;++a) for (int c=0;c<NV;++c) acc[c] += r->Y[a][c]*(r->lam[a]-lam0[a]);      /* the true response of what the leg arrived at */
      }
    }
    for (int c=0;c<NV;++c) dnu[c] = dnu0[c] + acc[c];
  }
  /* outputs: contact forces per body (world), warm start */
  memset(body_force, 0, sizeof(R)*NB*3);
  for (int lane=0;lane<4;++lane) for (int slot=0;slot<NSLOT;++slot) {
    Row* r = &rows[lane][slot]; if (!r->active) continue;
    for (int i=0;i<3;++i) body_force[3*r->body+i] += (r->n[i]*r->lam[0] + r->t1[i]*r->lam[1] + r->t2[i]*r->lam[2])/h;
    if (slot==0) for (int a=0;a<3;++a) s->b.foot_impulse[(4*e+lane)*3+a] = (float)r->lam[a];
    else if (glam) for (int a=0;a<3;++a) glam[lane][slot][a] = r->lam[a];
  }
  /* integrate (semi-implicit Euler; world-frame base velocity) */
  R nup[NV]; for (int i=0;i<NV;++i) nup[i] = nu_free[i] + dnu[i];
  for (int j=0;j<12;++j) { R vl = (R)kJointVelocity[j]; if (nup[6+j] > vl) nup[6+j] = vl; if (nup[6+j] < -vl) nup[6+j] = -vl; }
  R wxv[3]; cross3(k.v[0], k.v[0]+3, wxv);
  R lin_b[3] = {nup[3]+h*wxv[0], nup[4]+h*wxv[1], nup[5]+h*wxv[2]};
  R ww[3], vw[3]; mat3_vec(k.Rw[0], nup, ww); mat3_vec(k.Rw[0], lin_b, vw);
  { /* asset.max_angular_velocity / max_linear_velocity (legged_robot.py:974-975): the base twist is clamped, so no state can run off to inf */
    R wm = (R)cfg->max_angular_velocity, vm = (R)cfg->max_linear_velocity, w2 = dot3(ww,ww), v2 = dot3(vw,vw);
    if (w2 > wm*wm) { R sc = wm/SQRT(w2); ww[0]*=sc; ww[1]*=sc; ww[2]*=sc; }
    if (v2 > vm*vm) { R sc = vm/SQRT(v2); vw[0]*=sc; vw[1]*=sc; vw[2]*=sc; } }
  for (int i=0;i<3;++i) { root[7+i] = vw[i]; root[10+i] = ww[i]; root[i] += h*vw[i]; }
  { R th = SQRT(dot3(ww,ww))*h; R dq[4];
    if (th > RC(1e-9)) { R sc = SIN(th/2)/(th/h); dq[0]=ww[0]*sc; dq[1]=ww[1]*sc; dq[2]=ww[2]*sc; dq[3]=COS(th/2); }
    else { dq[0]=ww[0]*h/2; dq[1]=ww[1]*h/2; dq[2]=ww[2]*h/2; dq[3]=1; }
    R* p = root+3; R x=dq[3]*p[0]+dq[0]*p[3]+dq[1]*p[2]-dq[2]*p[1], y=dq[3]*p[1]-dq[0]*p[2]+dq[1]*p[3]+dq[2]*p[0],
       z=dq[3]*p[2]+dq[0]*p[1]-dq[1]*p[0]+dq[2]*p[3], w=dq[3]*p[3]-dq[0]*p[0]-dq[1]*p[1]-dq[2]*p[2];
    R inv = 1/SQRT(x*x+y*y+z*z+w*w); p[0]=x*inv; p[1]=y*inv; p[2]=z*inv; p[3]=w*inv; }
  for (int j=0;j<12;++j) { qd[j] = nup[6+j]; q[j] += h*qd[j]; }
  if (kout) kinematics(s, e, root, q, qd, kout);
}

/* write rigid_body_states[e] from kinematics at the current state */
static void write_body_states(Go2Sim* s, int e, const Kin* k) {
  for (int b=0;b<NB;++b) {
    if (!s->cfg.full_body_states && !(b==6||b==10||b==14||b==18)) continue;   /* feet rows only unless asked (go2sim.h) */
    int l = kBodyLink[b]; float* o = s->b.rigid_body_states + (size_t)(e*NB+b)*13;
    R off[3] = {(R)kBodyOffset[b][0], (R)kBodyOffset[b][1], (R)kBodyOffset[b][2]}, t[3]; mat3_vec(k->Rw[l], off, t);
    R qq[4]; mat_to_quat(k->Rw[l], qq);
    R vl[3], c[3]; cross3(k->v[l], off, c); for (int i=0;i<3;++i) vl[i] = k->v[l][3+i] + c[i];
    R vw[3], ww[3]; mat3_vec(k->Rw[l], vl, vw); mat3_vec(k->Rw[l], k->v[l], ww);
    for (int i=0;i<3;++i) { o[i] = (float)(k->pw[l][i]+t[i]); o[7+i] = (float)vw[i]; o[10+i] = (float)ww[i]; }
    for (int i=0;i<4;++i) o[3+i] = (float)qq[i];
  }
}

/* ------------------------------------------------------------------------------------------------
 * host-side scalars (what the reference keeps as Python floats)
 * ---------------------------------------------------------------------------------------------- */
static R current_scale(const float* c4, int64_t counter, int nsteps) { /* get_current_scale (:154-168) */
  double it = (double)(counter / nsteps);
  double pct = (it - c4[0])/((double)c4[1]-c4[0]); if (pct>1) pct=1; if (pct<0) pct=0;
  return (R)((1.0-pct)*c4[2] + pct*c4[3]);
}
static void update_reward_curriculum(Go2Sim* s, int force) { /* :144-152 */
  if (s->cfg.reward_curriculum_count == 0) return;
  if (s->common_step_counter % s->cfg.num_steps_per_env == 0 || force)
    for (int i=0;i<s->cfg.reward_curriculum_count;++i)
      s->reward_curr_scale[s->cfg.reward_curriculum_term[i]] = current_scale(s->cfg.reward_curriculum[i], s->common_step_counter, s->cfg.num_steps_per_env);
}
static void update_command_scalars(Go2Sim* s) { /* :433-446, :556-557 */
  int64_t it = s->common_step_counter / s->cfg.num_steps_per_env;
  /* the reference sorts by iter descending and applies every entry whose iter has passed, largest first,
   * popping it; applying in ascending order leaves the same final ranges */
  for (int pass=0; pass<s->cfg.cmd_curriculum_count; ++pass) {
    int best=-1; for (int i=0;i<s->cfg.cmd_curriculum_count;++i) if (!s->cmd_curr_done[i] && (double)it >= s->cfg.cmd_curriculum[i][0] && (best<0 || s->cfg.cmd_curriculum[i][0] < s->cfg.cmd_curriculum[best][0])) best=i;
    if (best<0) break;
    s->cmd_curr_done[best]=1;
    for (int r=0;r<4;++r) { s->cmd_ranges[r][0] = (R)s->cfg.cmd_curriculum[best][1+2*r]; s->cmd_ranges[r][1] = (R)s->cfg.cmd_curriculum[best][2+2*r]; }
  }
  R m = FABS(s->cmd_ranges[0][0]); if (FABS(s->cmd_ranges[0][1])>m) m=FABS(s->cmd_ranges[0][1]);
  if (FABS(s->cmd_ranges[1][0])>m) m=FABS(s->cmd_ranges[1][0]); if (FABS(s->cmd_ranges[1][1])>m) m=FABS(s->cmd_ranges[1][1]);
  s->max_lin_vel = m;
  if (s->cfg.zero_cmd_curriculum_enabled) s->zero_command_proba = current_scale(s->cfg.zero_cmd_curriculum, s->common_step_counter, s->cfg.num_steps_per_env);
}
/* env_command_ranges (:861-907): global range clipped by the env's terrain kind */
static void env_cmd_range(const Go2Sim* s, int e, int which, R* lo, R* hi) {
  *lo = s->cmd_ranges[which][0]; *hi = s->cmd_ranges[which][1];
  int kind = s->terrain_kind[e];
  if (kind >= 0 && (which < 3 || s->cfg.heading_command)) {
    R tl = (R)s->cfg.terrain_max_cmd_ranges[kind][which][0], th = (R)s->cfg.terrain_max_cmd_ranges[kind][which][1];
    if (tl > *lo) *lo = tl; if (th < *hi) *hi = th;
  }
}

static inline R uni(const Go2Sim* s, int e, int slot) {
  if (s->injected) return (R)s->injected[(size_t)e*GO2_NUM_UNIFORMS + slot];
  return (R)u01_step(s->cfg.seed, (uint32_t)(s->cfg.env_offset + e), (uint32_t)slot, s->step_count);
}
static inline R urange(R u, R lo, R hi) { return (hi-lo)*u + lo; } /* torch_rand_float */

/* sample_disjoint_intervals (legged_gym/utils/isaacgym_utils.py:32-47) */
static R sample_disjoint(R u01v, R bound, R cmin, R cmax) {
  R wn = -bound - cmin; if (wn<0) wn=0; R wp = cmax - bound; if (wp<0) wp=0;
  R tot = wn + wp + RC(1e-6); R u = u01v*tot;
  return (u < wn) ? cmin + u : cmax - wp + (u - wn);
}

/* _resample_commands for one env (legged_robot.py:423-592); U = first of 7 uniform slots */
static void resample_commands(Go2Sim* s, int e, int U) {
  const Go2SimCfg* c = &s->cfg; Go2SimBuffers* b = &s->b;
  float* cmd = b->commands + 4*e;
  b->stop_heading[e] = 0;
  R ax = (R)b->commands_xy_accumulation[2*e], ay = (R)b->commands_xy_accumulation[2*e+1];
  R remaining = RC(0.625)*(R)c->terrain_length - SQRT(ax*ax+ay*ay)*(R)c->cmd_resampling_time; if (remaining<0) remaining=0;
  b->commands_resampling_step[e] = (float)((R)c->cmd_resampling_time/s->dt);
  R xl,xh,yl,yh,wl,wh,hl,hh; env_cmd_range(s,e,0,&xl,&xh); env_cmd_range(s,e,1,&yl,&yh); env_cmd_range(s,e,2,&wl,&wh); env_cmd_range(s,e,3,&hl,&hh);
  R eplen = (R)b->episode_length_buf[e];
  if (c->dynamic_resample_commands) {
    R vlow = remaining/((s->max_episode_length - eplen + RC(1e-9))*s->dt); if (vlow<0) vlow=0;
    cmd[0] = (float)sample_disjoint(uni(s,e,U+0), vlow, xl, xh);
    cmd[1] = (float)sample_disjoint(uni(s,e,U+1), vlow, yl, yh);
    if (c->heading_command) cmd[3] = (float)urange(uni(s,e,U+2), hl, hh); else cmd[2] = (float)urange(uni(s,e,U+2), wl, wh);
  } else {
    cmd[0] = (float)(xl + uni(s,e,U+0)*(xh-xl)); cmd[1] = (float)(yl + uni(s,e,U+1)*(yh-yl));
    if (c->heading_command) cmd[3] = (float)(hl + uni(s,e,U+2)*(hh-hl)); else cmd[2] = (float)(wl + uni(s,e,U+2)*(wh-wl));
    R nn = SQRT((R)cmd[0]*cmd[0]+(R)cmd[1]*cmd[1]); if (!(nn > RC(0.2))) { cmd[0]=0; cmd[1]=0; }
  }
  R p = uni(s,e,U+3), minp=0, maxp=0;
  if (c->limit_vel_prob > 0) {
    maxp += (R)c->limit_vel_prob;
    int lim = (p>=minp) && (p<maxp);
    if (lim) {
      if (c->limit_vel_invert_when_continuous && b->last_is_limit_vel[e]) { cmd[0]=-cmd[0]; cmd[1]=-cmd[1]; cmd[2]=-cmd[2]; }
      else {
        int idx = (int)(uni(s,e,U+4)*c->limit_vel_comb_count); if (idx >= c->limit_vel_comb_count) idx = c->limit_vel_comb_count-1;
        const float* cb = c->limit_vel_comb[idx];
        cmd[0] = (float)(cb[0]==-1 ? xl : xh); if (cb[0]==0) cmd[0]=0;
        cmd[1] = (float)(cb[1]==-1 ? yl : yh); if (cb[1]==0) cmd[1]=0;
        cmd[2] = (float)(cb[2]==-1 ? wl : wh); if (cb[2]==0) cmd[2]=0;
      }
      if (c->heading_command && c->stop_heading_at_limit) b->stop_heading[e]=1;
    }
    b->last_is_limit_vel[e] = (uint8_t)lim;
    minp += (R)c->limit_vel_prob;
  }
  if (s->zero_command_proba > 0) {
    maxp += s->zero_command_proba;
    R nxt = s->max_episode_length - eplen - remaining/(RC(0.8)*s->max_lin_vel*s->dt + RC(1e-9));
    R top = (R)c->cmd_resampling_time/s->dt; if (nxt<0) nxt=0; if (nxt>top) nxt=top;
    if ((p>=minp) && (p<maxp) && nxt>0) {
      cmd[0]=0; cmd[1]=0; b->commands_resampling_step[e] = (float)nxt;
      if (c->limit_ang_vel_at_zero_command_prob > 0 && uni(s,e,U+5) < (R)c->limit_ang_vel_at_zero_command_prob) {
        cmd[2] = (float)(uni(s,e,U+6) < RC(0.5) ? wl : wh);
        if (c->heading_command) b->stop_heading[e]=1;
      }
    }
    minp += s->zero_command_proba;
  }
  if (c->turn_over && b->turn_over_timer[e] > 0) { cmd[0]=0; cmd[1]=0; cmd[2]=0; b->stop_heading[e]=1; } /* :586-590 */
  b->commands_xy_accumulation[2*e] += cmd[0]; b->commands_xy_accumulation[2*e+1] += cmd[1];
}

/* _get_heights (legged_robot.py:1188-1224) */
static void get_heights(Go2Sim* s, int e) {
  float* mh = s->b.measured_heights + (size_t)e*GO2_NUM_HEIGHT_POINTS;
  if (s->cfg.terrain_mode == 0) { for (int i=0;i<GO2_NUM_HEIGHT_POINTS;++i) mh[i]=0; return; }
  const float* root = s->b.root_states + 13*e;
  /* quat_apply_yaw (utils/math.py:8-12): zero x,y of the quaternion, normalise, apply */
  R qy[4] = {0,0,(R)root[5],(R)root[6]}; R nn = SQRT(qy[2]*qy[2]+qy[3]*qy[3]); if (nn < RC(1e-9)) nn = RC(1e-9); qy[2]/=nn; qy[3]/=nn;
  int rows = s->cfg.hf_rows, cols = s->cfg.hf_cols;
  for (int i=0;i<GO2_NUM_HEIGHT_POINTS;++i) {
    R p[3] = {s->height_pts[i][0], s->height_pts[i][1], 0}, w[3]; quat_apply(qy, p, w);
    R x = (w[0]+(R)root[0]+(R)s->cfg.hf_border)/(R)s->cfg.hf_hscale, y = (w[1]+(R)root[1]+(R)s->cfg.hf_border)/(R)s->cfg.hf_hscale;
    long px = (long)x, py = (long)y;  /* .long() truncates toward zero */
    if (px<0) px=0; if (px>rows-2) px=rows-2; if (py<0) py=0; if (py>cols-2) py=cols-2;
    int h1 = s->hf[px*cols+py], h2 = s->hf[(px+1)*cols+py], h3 = s->hf[px*cols+py+1];
    int hm = h1<h2?h1:h2; if (h3<hm) hm=h3;
    mh[i] = (float)(hm*(R)s->cfg.hf_vscale);
  }
}
static R get_base_height(const Go2Sim* s, int e) { /* :1387-1397 */
  const float* root = s->b.root_states + 13*e;
  if (!s->cfg.measure_heights) return (R)root[2];
  const float* mh = s->b.measured_heights + (size_t)e*GO2_NUM_HEIGHT_POINTS;
  R sum=0; for (int i=0;i<GO2_NUM_HEIGHT_POINTS;++i) sum += (R)mh[i]*(R)s->height_mask[i];
  return (R)root[2] - sum/(R)s->num_height_mask;
}
static R dynamic_sigma(const Go2Sim* s, int e, R vabs, R vmin, R vmax) { /* :1300-1320 */
  R def = (R)s->cfg.tracking_sigma;
  if (!s->cfg.terrain_curriculum || !s->cfg.dynamic_sigma_enabled || s->terrain_kind[e] < 0) return def;
  R target = (R)s->cfg.dynamic_sigma_max[s->terrain_kind[e]], sig = def;
  if (vabs >= vmin && vabs < vmax) sig = def + (vabs-vmin)/(vmax-vmin)*(target-def);
  if (vabs >= vmax) sig = target;
  R ls = EXP(((R)s->b.terrain_levels[e]+1)/10)-1; if (ls>1) ls=1;
  return def + ls*(sig-def);
}

/* compute_reward (legged_robot.py:247-274) for one env */
static void compute_reward(Go2Sim* s, int e) {
  const Go2SimCfg* c = &s->cfg; Go2SimBuffers* b = &s->b;
  const float* root = b->root_states+13*e; const float* ds = b->dof_state+24*e; const float* cf = b->contact_forces+(size_t)e*NB*3;
  const float* rb = b->rigid_body_states+(size_t)e*NB*13; const float* cmd = b->commands+4*e;
  const float* blv = b->base_lin_vel+3*e; const float* bav = b->base_ang_vel+3*e; const float* pg = b->projected_gravity+3*e;
  const float* tq = b->torques+12*e; const float* act = b->actions+12*e; const float* la = b->last_actions+12*e;
  float* lla = b->last_last_actions+12*e; const float* ldv = b->last_dof_vel+12*e;
  static const int feet[4] = {6,10,14,18}; static const int pen[8] = {4,5,8,9,12,13,16,17};
  R raw[GO2_NUM_REWARDS]; memset(raw,0,sizeof(raw));
#define ACTIVE(t) (s->reward_scale_dt[t] != 0 || s->turn_over_scale_dt[t] != 0)   /* names = union of both scale tables (:927-930) */
  if (ACTIVE(GO2_REW_TRACKING_LIN_VEL)) { /* :1322 */
    R sx = (R)c->tracking_sigma, sy = sx;
    if (c->dynamic_sigma_enabled) { sx = dynamic_sigma(s,e,FABS((R)cmd[0]),(R)c->dynamic_sigma_vel[0],(R)c->dynamic_sigma_vel[1]); sy = dynamic_sigma(s,e,FABS((R)cmd[1]),(R)c->dynamic_sigma_vel[0],(R)c->dynamic_sigma_vel[1]); }
    R ex = (R)cmd[0]-(R)blv[0], ey = (R)cmd[1]-(R)blv[1]; raw[GO2_REW_TRACKING_LIN_VEL] = EXP(-(ex*ex/sx + ey*ey/sy)); }
  if (ACTIVE(GO2_REW_TRACKING_ANG_VEL)) { /* :1336 */
    R sg = (R)c->tracking_sigma; if (c->dynamic_sigma_enabled) sg = dynamic_sigma(s,e,FABS((R)cmd[2]),(R)c->dynamic_sigma_vel[2],(R)c->dynamic_sigma_vel[3]);
    R er = (R)cmd[2]-(R)bav[2]; raw[GO2_REW_TRACKING_ANG_VEL] = EXP(-er*er/sg); }
  if (ACTIVE(GO2_REW_LIN_VEL_Z)) raw[GO2_REW_LIN_VEL_Z] = (R)blv[2]*(R)blv[2];                       /* :1228 */
  if (ACTIVE(GO2_REW_ANG_VEL_XY)) raw[GO2_REW_ANG_VEL_XY] = (R)bav[0]*bav[0]+(R)bav[1]*bav[1];       /* :1232 */
  if (ACTIVE(GO2_REW_ORIENTATION)) raw[GO2_REW_ORIENTATION] = (R)pg[0]*pg[0]+(R)pg[1]*pg[1];         /* :1236 */
  if (ACTIVE(GO2_REW_BASE_HEIGHT)) { /* :1245-1259 */
    int nc=0; R fp[3]={0,0,0};
    for (int f=0;f<4;++f) { int ct = (R)cf[3*feet[f]+2] > 1; int filt = ct || b->last_contacts2[4*e+f]; b->last_contacts2[4*e+f]=(uint8_t)ct;
      if (filt) { nc++; for (int i=0;i<3;++i) fp[i] += (R)rb[13*feet[f]+i]; } }
    R den = nc<1?1:nc; R bh=0; for (int i=0;i<3;++i) bh += (fp[i]/den-(R)root[i])*(R)pg[i];
    R d = bh-(R)c->base_height_target; raw[GO2_REW_BASE_HEIGHT] = d*d*(nc>0); }
  if (ACTIVE(GO2_REW_TORQUES)) { R a=0; for (int j=0;j<12;++j) a += (R)tq[j]*tq[j]; raw[GO2_REW_TORQUES]=a; }             /* :1261 */
  if (ACTIVE(GO2_REW_DOF_VEL)) { R a=0; for (int j=0;j<12;++j) a += (R)ds[2*j+1]*ds[2*j+1]; raw[GO2_REW_DOF_VEL]=a; }     /* :1265 */
  if (ACTIVE(GO2_REW_DOF_ACC)) { R a=0; for (int j=0;j<12;++j) { R d=((R)ldv[j]-(R)ds[2*j+1])/s->dt; a+=d*d; } raw[GO2_REW_DOF_ACC]=a; } /* :1269 */
  if (ACTIVE(GO2_REW_ACTION_RATE)) { R a=0; for (int j=0;j<12;++j) { R d=(R)la[j]-(R)act[j]; a+=d*d; } raw[GO2_REW_ACTION_RATE]=a; }    /* :1273 */
  if (ACTIVE(GO2_REW_COLLISION)) { R a=0; for (int p=0;p<8;++p) { const float* f=cf+3*pen[p]; a += SQRT((R)f[0]*f[0]+(R)f[1]*f[1]+(R)f[2]*f[2]) > RC(0.1); } raw[GO2_REW_COLLISION]=a; } /* :1277 */
  if (ACTIVE(GO2_REW_DOF_POS_LIMITS)) { R a=0; for (int j=0;j<12;++j) { R q=(R)ds[2*j]; R lo=q-s->dof_pos_limits[j][0]; if (lo<0) a-=lo; R hi=q-s->dof_pos_limits[j][1]; if (hi>0) a+=hi; } raw[GO2_REW_DOF_POS_LIMITS]=a; } /* :1285 */
  if (ACTIVE(GO2_REW_DOF_VEL_LIMITS)) { R a=0; for (int j=0;j<12;++j) { R d=FABS((R)ds[2*j+1])-(R)kJointVelocity[j]*(R)c->soft_dof_vel_limit; if (d<0) d=0; if (d>1) d=1; a+=d; } raw[GO2_REW_DOF_VEL_LIMITS]=a; } /* :1291 */
  if (ACTIVE(GO2_REW_TORQUE_LIMITS)) { R a=0; for (int j=0;j<12;++j) { R d=FABS((R)tq[j])-(R)kJointEffort[j]*(R)c->soft_torque_limit; if (d<0) d=0; a+=d; } raw[GO2_REW_TORQUE_LIMITS]=a; } /* :1296 */
  if (ACTIVE(GO2_REW_FEET_AIR_TIME)) { /* :1347-1358 */
    R a=0; R cn = SQRT((R)cmd[0]*cmd[0]+(R)cmd[1]*cmd[1]);
    for (int f=0;f<4;++f) { int ct=(R)cf[3*feet[f]+2] > 1; int filt = ct || b->last_contacts[4*e+f]; b->last_contacts[4*e+f]=(uint8_t)ct;
      float* fat = b->feet_air_time+4*e+f; int first = (*fat > 0) && filt; *fat += (float)s->dt; a += ((R)*fat-RC(0.5))*first; if (filt) *fat=0; }
    raw[GO2_REW_FEET_AIR_TIME] = a*(cn > RC(0.1)); }
  if (ACTIVE(GO2_REW_STUMBLE)) { int any=0; for (int f=0;f<4;++f) { const float* F=cf+3*feet[f]; if (SQRT((R)F[0]*F[0]+(R)F[1]*F[1]) > 5*FABS((R)F[2])) any=1; } raw[GO2_REW_STUMBLE]=any; } /* :1360 */
  if (ACTIVE(GO2_REW_STAND_STILL)) { R a=0; for (int j=0;j<12;++j) a += FABS((R)ds[2*j]-(R)c->default_dof_pos[j]); raw[GO2_REW_STAND_STILL] = a*(SQRT((R)cmd[0]*cmd[0]+(R)cmd[1]*cmd[1]) < RC(0.1)); } /* :1365 */
  if (ACTIVE(GO2_REW_FEET_CONTACT_FORCES)) { R a=0; for (int f=0;f<4;++f) { const float* F=cf+3*feet[f]; R d=SQRT((R)F[0]*F[0]+(R)F[1]*F[1]+(R)F[2]*F[2])-(R)c->max_contact_force; if (d>0) a+=d; } raw[GO2_REW_FEET_CONTACT_FORCES]=a; } /* :1369 */
  if (ACTIVE(GO2_REW_ACTION_SMOOTHNESS)) { R a=0; for (int j=0;j<12;++j) { R d=(R)act[j]-2*(R)la[j]+(R)lla[j]; a+=d*d; lla[j]=la[j]; } raw[GO2_REW_ACTION_SMOOTHNESS]=a; } /* :1373 */
  if (ACTIVE(GO2_REW_DOF_POWER)) { R a=0; for (int j=0;j<12;++j) a += FABS((R)tq[j]*(R)ds[2*j+1]); raw[GO2_REW_DOF_POWER]=a; } /* :1381 */
  if (ACTIVE(GO2_REW_CORRECT_BASE_HEIGHT)) { R d = get_base_height(s,e)-(R)c->base_height_target; raw[GO2_REW_CORRECT_BASE_HEIGHT]=d*d; } /* :1399 */
  if (ACTIVE(GO2_REW_FEET_REGULATION)) { /* :1404-1414 */
    R bh = get_base_height(s,e), a=0;
    for (int f=0;f<4;++f) { const float* F = rb+13*feet[f]; R f2b=0; for (int i=0;i<3;++i) f2b += ((R)F[i]-(R)root[i])*(R)pg[i];
      R fh = bh - f2b; if (fh<0) fh=0; a += ((R)F[7]*F[7]+(R)F[8]*F[8])*EXP(-fh/(RC(0.025)*(R)c->base_height_target)); }
    raw[GO2_REW_FEET_REGULATION]=a; }
  if (ACTIVE(GO2_REW_SIMILAR_TO_DEFAULT)) { R a=0; for (int j=0;j<12;++j) a += FABS((R)ds[2*j]-(R)c->default_dof_pos[j]); raw[GO2_REW_SIMILAR_TO_DEFAULT]=a; } /* :1416 */
  if (ACTIVE(GO2_REW_UPRIGHT)) raw[GO2_REW_UPRIGHT] = (-1-(R)pg[2])/2; /* :1420 */
  if (ACTIVE(GO2_REW_LEGS_DISTANCE)) { /* :1423-1441 */
    R ly[4]; for (int f=0;f<4;++f) { R d[3], l[3]; for (int i=0;i<3;++i) d[i]=(R)rb[13*feet[f]+i]-(R)root[i]; R q[4]={(R)root[3],(R)root[4],(R)root[5],(R)root[6]}; quat_rotate_inverse(q,d,l); ly[f]=l[1]; }
    R df = (R)c->min_legs_distance-(ly[0]-ly[1]); if (df<0) df=0; R dr = (R)c->min_legs_distance-(ly[2]-ly[3]); if (dr<0) dr=0; raw[GO2_REW_LEGS_DISTANCE]=df*df+dr*dr; }
  if (ACTIVE(GO2_REW_HIP_TO_DEFAULT)) { R a=0; for (int l=0;l<4;++l) a += FABS((R)ds[2*3*l]-(R)c->default_dof_pos[3*l]); raw[GO2_REW_HIP_TO_DEFAULT]=a; } /* go2_env.py:55 */
  if (ACTIVE(GO2_REW_X_COMMAND_HIP_REGULAR)) { /* go2_env.py:62 */
    R ratio = FABS((R)cmd[0])/SQRT((R)cmd[0]*cmd[0]+(R)cmd[1]*cmd[1]+(R)cmd[2]*cmd[2]);
    raw[GO2_REW_X_COMMAND_HIP_REGULAR] = (FABS((R)ds[0]+(R)ds[6]) + FABS((R)ds[12]+(R)ds[18]))*ratio; }
  R total = 0;
  const int need_turn_over = c->turn_over && FABS((R)b->rpy[3*e]) > (R)c->turn_over_roll_threshold;   /* :263-265 */
  for (int t=0;t<GO2_REW_TERMINATION;++t) if (ACTIVE(t)) {
    R r = raw[t]*(need_turn_over ? s->turn_over_scale_dt[t] : s->reward_scale_dt[t]); if (s->reward_has_curr[t]) r *= s->reward_curr_scale[t];
    total += r; b->episode_sums[(size_t)t*s->N+e] += (float)r;
  }
  if (c->only_positive_rewards && total < 0) total = 0;
  if (ACTIVE(GO2_REW_TERMINATION)) { R r = (R)(b->reset_buf[e] && !b->time_out_buf[e])*s->reward_scale_dt[GO2_REW_TERMINATION]; total += r; b->episode_sums[(size_t)GO2_REW_TERMINATION*s->N+e] += (float)r; }
  b->rew_buf[e] = (float)total;
#undef ACTIVE
}

/* commit API tensors -> nothing to do for the oracle (the AoS tensors are the state) */

/* reset_idx for one env (legged_robot.py:180-245, _reset_dofs :620, _reset_root_states :635, _update_terrain_curriculum :1143) */
static void reset_env(Go2Sim* s, int e, int initial) {
  const Go2SimCfg* c = &s->cfg; Go2SimBuffers* b = &s->b;
  for (int j=0;j<12;++j) {
    if (c->randomize_motor_strength) b->motor_strengths[12*e+j] = (float)urange(uni(s,e,GO2_U_RESET_STRENGTH+j),(R)c->motor_strength_range[0],(R)c->motor_strength_range[1]);
    if (c->randomize_motor_zero_offset) b->motor_zero_offsets[12*e+j] = (float)urange(uni(s,e,GO2_U_RESET_OFFSET+j),(R)c->motor_zero_offset_range[0],(R)c->motor_zero_offset_range[1]);
    if (c->randomize_pd_gains) { b->p_gains_multiplier[12*e+j] = (float)urange(uni(s,e,GO2_U_RESET_KP+j),(R)c->stiffness_mult_range[0],(R)c->stiffness_mult_range[1]);
                                 b->d_gains_multiplier[12*e+j] = (float)urange(uni(s,e,GO2_U_RESET_KD+j),(R)c->damping_mult_range[0],(R)c->damping_mult_range[1]); }
  }
  if (c->terrain_curriculum && c->terrain_mode != 0 && !initial) { /* :1143-1169 */
    R dist = (R)b->max_move_distance[e];
    int up = dist > (R)c->terrain_length/2, down;
    if (c->move_down_by_accumulated_xy_command) { R ax=(R)b->commands_xy_accumulation[2*e], ay=(R)b->commands_xy_accumulation[2*e+1];
      down = (dist < SQRT(ax*ax+ay*ay)*((R)c->cmd_resampling_time*(1-s->zero_command_proba))*RC(0.5)) && !up; }
    else { R cx=(R)b->commands[4*e], cy=(R)b->commands[4*e+1]; down = (dist < SQRT(cx*cx+cy*cy)*(R)c->episode_length_s*RC(0.5)) && !up; }
    int64_t lv = b->terrain_levels[e] + up - down;
    if (lv >= c->terrain_num_levels) { int r = (int)(uni(s,e,GO2_U_RESET_TERRAIN)*c->terrain_num_levels); if (r>=c->terrain_num_levels) r=c->terrain_num_levels-1; lv = r; }
    else if (lv < 0) lv = 0;
    b->terrain_levels[e] = lv;
    const float* o = s->terrain_origins + ((size_t)lv*c->terrain_num_types + b->terrain_types[e])*3;
    for (int i=0;i<3;++i) b->env_origins[3*e+i] = o[i];
    b->max_move_distance[e] = 0;
  }
  for (int j=0;j<12;++j) { b->dof_state[24*e+2*j] = (float)((R)c->default_dof_pos[j]*urange(uni(s,e,GO2_U_RESET_DOF+j),RC(0.5),RC(1.5))); b->dof_state[24*e+2*j+1]=0; }
  float* root = b->root_states+13*e;
  R yaw = urange(uni(s,e,GO2_U_RESET_YAW), -(R)M_PI, (R)M_PI);
  for (int i=0;i<13;++i) root[i] = c->base_init_state[i];
  R roll = 0;
  if (c->turn_over) { /* :642-684: a share of the resets starts on the back (roll pi) or on a side (roll +-pi/2) */
    b->turn_over_timer[e] = 0;
    R pr = uni(s,e,GO2_U_TURN), p0 = (R)c->turn_over_proportions[0], p1 = p0 + (R)c->turn_over_proportions[1];
    if (pr >= 0 && pr < p0) { root[2] = (float)urange(uni(s,e,GO2_U_TURN+1),(R)c->turn_over_init_heights[0][0],(R)c->turn_over_init_heights[0][1]); roll = (R)M_PI; b->turn_over_timer[e] = c->turn_over_zero_time[0]; }
    else if (pr >= p0 && pr < p1) { root[2] = (float)urange(uni(s,e,GO2_U_TURN+2),(R)c->turn_over_init_heights[1][0],(R)c->turn_over_init_heights[1][1]);
      roll = uni(s,e,GO2_U_TURN+3) < RC(0.5) ? (R)M_PI/2 : -(R)M_PI/2; b->turn_over_timer[e] = c->turn_over_zero_time[1]; }
  }
  { /* quat_from_euler_xyz(roll, 0, yaw) (isaacgym torch_utils) */
    R cy=COS(yaw/2), sy=SIN(yaw/2), cr=COS(roll/2), sr=SIN(roll/2);
    root[3]=(float)(cy*sr); root[4]=(float)(sy*sr); root[5]=(float)(sy*cr); root[6]=(float)(cy*cr); }
  for (int i=0;i<3;++i) root[i] += b->env_origins[3*e+i];
  if (c->terrain_mode != 0) { root[0] += (float)urange(uni(s,e,GO2_U_RESET_XY),-1,1); root[1] += (float)urange(uni(s,e,GO2_U_RESET_XY+1),-1,1); }
  for (int i=0;i<6;++i) root[7+i] = (float)urange(uni(s,e,GO2_U_RESET_VEL+i),RC(-0.5),RC(0.5));
  for (int j=0;j<12;++j) { b->actions[12*e+j]=0; b->last_actions[12*e+j]=0; b->last_dof_vel[12*e+j]=0; }
  for (int f=0;f<4;++f) { b->feet_air_time[4*e+f]=0; for (int a=0;a<3;++a) b->foot_impulse[(4*e+f)*3+a]=0; }
  b->episode_length_buf[e]=0; b->reset_buf[e]=1;
  b->commands_resampling_step[e] = (float)((R)c->cmd_resampling_time/s->dt);
  b->commands_xy_accumulation[2*e]=0; b->commands_xy_accumulation[2*e+1]=0;
  resample_commands(s, e, GO2_U_RSB);
  for (int t=0;t<GO2_NUM_REWARDS;++t) { s->ep_sum[t] += b->episode_sums[(size_t)t*s->N+e]; b->episode_sums[(size_t)t*s->N+e]=0; }
  s->ep_count++;
}

/* Go2Robot.compute_observations (go2_env.py:23-53) + the clip of step() (legged_robot.py:96-99) */
static void compute_observations(Go2Sim* s, int e) {
  const Go2SimCfg* c = &s->cfg; Go2SimBuffers* b = &s->b;
  float* o = b->obs_buf+(size_t)e*GO2_NUM_OBS; float* p = b->privileged_obs_buf+(size_t)e*GO2_NUM_PRIV_OBS;
  const float* ds=b->dof_state+24*e; const float* cmd=b->commands+4*e; const float* cf=b->contact_forces+(size_t)e*NB*3;
  static const int feet[4] = {6,10,14,18};
  R cs[3] = {(R)c->obs_scale_lin_vel,(R)c->obs_scale_lin_vel,(R)c->obs_scale_ang_vel};
  R ob[GO2_NUM_OBS];
  for (int i=0;i<3;++i) { ob[i]=(R)b->base_ang_vel[3*e+i]*(R)c->obs_scale_ang_vel; ob[3+i]=(R)b->projected_gravity[3*e+i]; ob[6+i]=(R)cmd[i]*cs[i]; }
  for (int j=0;j<12;++j) { ob[9+j]=((R)ds[2*j]-(R)c->default_dof_pos[j])*(R)c->obs_scale_dof_pos; ob[21+j]=(R)ds[2*j+1]*(R)c->obs_scale_dof_vel; ob[33+j]=(R)b->actions[12*e+j]; }
  R cl = (R)c->clip_observations;
  for (int i=0;i<3;++i) p[i]=(float)((R)b->base_lin_vel[3*e+i]*(R)c->obs_scale_lin_vel);
  for (int i=0;i<GO2_NUM_OBS;++i) p[3+i]=(float)ob[i];
  for (int f=0;f<4;++f) { const float* F=cf+3*feet[f]; p[48+f]=(float)(SQRT((R)F[0]*F[0]+(R)F[1]*F[1]+(R)F[2]*F[2])*RC(1e-3)); }
  for (int j=0;j<12;++j) { p[52+j]=(float)((R)b->torques[12*e+j]/(R)kJointEffort[j]); p[64+j]=(float)(((R)b->last_dof_vel[12*e+j]-(R)ds[2*j+1])/s->dt*RC(1e-4)); }
  for (int i=0;i<GO2_NUM_HEIGHT_POINTS;++i) { R hh=(R)b->root_states[13*e+2]-RC(0.5)-(R)b->measured_heights[(size_t)e*GO2_NUM_HEIGHT_POINTS+i]; if (hh<-1) hh=-1; if (hh>1) hh=1; p[76+i]=(float)(hh*(R)c->obs_scale_height); }
  for (int i=0;i<GO2_NUM_PRIV_OBS;++i) { if (p[i]>cl) p[i]=(float)cl; if (p[i]<-cl) p[i]=(float)-cl; }
  for (int i=0;i<GO2_NUM_OBS;++i) { R v=ob[i]; if (c->add_noise) v += (2*uni(s,e,GO2_U_NOISE+i)-1)*s->noise_vec[i]; if (v>cl) v=cl; if (v<-cl) v=-cl; o[i]=(float)v; }
}

/* post_physics_step (legged_robot.py:102-142) for one env, given the four state tensors */
static void post_physics_env(Go2Sim* s, int e) {
  const Go2SimCfg* c = &s->cfg; Go2SimBuffers* b = &s->b;
  float* root = b->root_states+13*e;
  b->episode_length_buf[e] += 1; b->commands_resampling_step[e] -= 1;
  if (c->turn_over) { R tt = (R)b->turn_over_timer[e] - s->dt; b->turn_over_timer[e] = (float)(tt < 0 ? 0 : tt); }   /* :114-115 */
  R q[4]={(R)root[3],(R)root[4],(R)root[5],(R)root[6]};
  { /* get_euler_xyz (utils/isaacgym_utils.py:11-30) */
    R sr = 2*(q[3]*q[0]+q[1]*q[2]), cr = q[3]*q[3]-q[0]*q[0]-q[1]*q[1]+q[2]*q[2];
    R sp = 2*(q[3]*q[1]-q[2]*q[0]);
    R sy = 2*(q[3]*q[2]+q[0]*q[1]), cy = q[3]*q[3]+q[0]*q[0]-q[1]*q[1]-q[2]*q[2];
    b->rpy[3*e] = (float)ATAN2(sr,cr);
    b->rpy[3*e+1] = (float)(FABS(sp)>=1 ? (sp>0? (R)M_PI/2 : (sp<0 ? -(R)M_PI/2 : 0)) : ASIN(sp));
    b->rpy[3*e+2] = (float)ATAN2(sy,cy); }
  R v[3]={(R)root[7],(R)root[8],(R)root[9]}, w[3]={(R)root[10],(R)root[11],(R)root[12]}, g[3]={0,0,-1}, o[3];
  quat_rotate_inverse(q,v,o); for (int i=0;i<3;++i) b->base_lin_vel[3*e+i]=(float)o[i];
  quat_rotate_inverse(q,w,o); for (int i=0;i<3;++i) b->base_ang_vel[3*e+i]=(float)o[i];
  quat_rotate_inverse(q,g,o); for (int i=0;i<3;++i) b->projected_gravity[3*e+i]=(float)o[i];
  { R dx=(R)root[0]-(R)b->env_origins[3*e], dy=(R)root[1]-(R)b->env_origins[3*e+1]; R d=SQRT(dx*dx+dy*dy); if (d>(R)b->max_move_distance[e]) b->max_move_distance[e]=(float)d; }
  /* _post_physics_step_callback (:404-421) */
  if ((R)b->commands_resampling_step[e] <= 0 && (R)b->episode_length_buf[e] < s->max_episode_length-1) { resample_commands(s, e, GO2_U_RSA); s->cb_resamples++; }
  if (c->heading_command && !b->stop_heading[e]) {
    R f[3]={1,0,0}, fw[3]; quat_apply(q,f,fw); R heading=ATAN2(fw[1],fw[0]);
    R a=(R)b->commands[4*e+3]-heading; a = FMOD(a, 2*(R)M_PI); if (a<0) a += 2*(R)M_PI; if (a>(R)M_PI) a -= 2*(R)M_PI; /* wrap_to_pi (utils/math.py:15-18) */
    R lo,hi; env_cmd_range(s,e,2,&lo,&hi);
    if (s->yaw_seen) { int sn = s->cmd_stage_seen; lo = sn<0 ? (R)c->cmd_ranges[2][0] : (R)c->cmd_curriculum[sn][5]; hi = sn<0 ? (R)c->cmd_ranges[2][1] : (R)c->cmd_curriculum[sn][6];
      int kind = s->terrain_kind[e]; if (kind >= 0) { R tl=(R)c->terrain_max_cmd_ranges[kind][2][0], th=(R)c->terrain_max_cmd_ranges[kind][2][1]; if (tl>lo) lo=tl; if (th<hi) hi=th; } }
    R y=RC(0.5)*a; if (y<lo) y=lo; if (y>hi) y=hi; b->commands[4*e+2]=(float)y; }
  if (c->measure_heights) get_heights(s, e);
  /* check_termination (:170-178): termination body = base (index 0) */
  { const float* F=b->contact_forces+(size_t)e*NB*3; int r = !c->turn_over && SQRT((R)F[0]*F[0]+(R)F[1]*F[1]+(R)F[2]*F[2]) > 1;   /* :174 */
    int to = (R)b->episode_length_buf[e] > s->max_episode_length; b->time_out_buf[e]=(uint8_t)to; b->reset_buf[e]=(uint8_t)(r||to); }
  compute_reward(s, e);
  if (b->reset_buf[e]) reset_env(s, e, 0);
  if (c->push_robots && (b->episode_length_buf[e] % c->push_interval == 0)) { /* _push_robots (:709-724) */
    root[7]=(float)urange(uni(s,e,GO2_U_PUSH),-(R)c->max_push_vel_xy,(R)c->max_push_vel_xy); root[8]=(float)urange(uni(s,e,GO2_U_PUSH+1),-(R)c->max_push_vel_xy,(R)c->max_push_vel_xy);
    for (int i=0;i<3;++i) root[10+i]=(float)urange(uni(s,e,GO2_U_PUSH+2+i),-(R)c->max_push_ang_vel,(R)c->max_push_ang_vel); }
  compute_observations(s, e);
  for (int j=0;j<12;++j) { b->last_actions[12*e+j]=b->actions[12*e+j]; b->last_dof_vel[12*e+j]=b->dof_state[24*e+2*j+1]; }
  for (int i=0;i<6;++i) b->last_root_vel[6*e+i]=root[7+i];
}

static void finish_episode_info(Go2Sim* s) { /* extras["episode"] (:229-242) */
  if (s->ep_count > 0) {
    for (int t=0;t<GO2_NUM_REWARDS;++t) s->b.episode_info[t] = (float)(s->ep_sum[t]/s->ep_count/(double)s->cfg.episode_length_s);
    s->b.episode_info[GO2_NUM_REWARDS] = (float)s->ep_count;
    { /* terrain_level_all / terrain_level_<name> (:231-237): mean level of all envs / of the envs on each terrain kind, NaN for an empty group */
      double ls[GO2_NUM_TERRAIN_KINDS+1] = {0}, lc[GO2_NUM_TERRAIN_KINDS+1] = {0};
      if (s->cfg.terrain_mode != 0) for (int e=0;e<s->N;++e) { double lv=(double)s->b.terrain_levels[e]; int kd=s->terrain_kind[e]; ls[0]+=lv; lc[0]+=1; if (kd>=0 && kd<GO2_NUM_TERRAIN_KINDS) { ls[1+kd]+=lv; lc[1+kd]+=1; } }
      for (int k=0;k<=GO2_NUM_TERRAIN_KINDS;++k) s->b.episode_info[GO2_NUM_REWARDS+3+k] = s->cfg.terrain_mode == 0 ? (k==0 ? 0.0f : (float)NAN) : (lc[k]>0 ? (float)(ls[k]/lc[k]) : (float)NAN); }
  }
  /* The Python list command_ranges['lin_vel_x'] (the reference works on whole batches: callback for all envs, then reset_idx for all
   * reset envs).  Every _resample_commands call with >= 1 env replaces it when a new command_range_curriculum stage has started
   * (:433-446): first the post-physics callback's call (:409-410), then — after update_command_curriculum widened it (:728-737) — the
   * call inside reset_idx.  Nothing samples from the list in this fork (_resample_commands reads env_command_ranges, rebuilt only at a
   * stage start from the replaced list); it is what extras['episode']['max_command_x'] reports (:241-242). */
  { int64_t it = s->common_step_counter / s->cfg.num_steps_per_env; int stage = -1;
    for (int i=0;i<s->cfg.cmd_curriculum_count;++i) if ((double)it >= s->cfg.cmd_curriculum[i][0] && (stage<0 || s->cfg.cmd_curriculum[i][0] > s->cfg.cmd_curriculum[stage][0])) stage=i;
    for (int call=0; call<2; ++call) {
      if (call==0 ? s->cb_resamples == 0 : s->ep_count == 0) continue;
      if (call==1 && s->cfg.cmd_tracking_curriculum &&
          s->ep_sum[GO2_REW_TRACKING_LIN_VEL]/s->ep_count/(double)s->max_episode_length > 0.8*(double)s->cfg.reward_scales[GO2_REW_TRACKING_LIN_VEL]*(double)s->dt) {
        double lo = s->cmd_x_track[0]-0.5, hi = s->cmd_x_track[1]+0.5, m = (double)s->cfg.cmd_max_curriculum;
        s->cmd_x_track[0] = lo < -m ? -m : (lo > 0 ? 0 : lo); s->cmd_x_track[1] = hi < 0 ? 0 : (hi > m ? m : hi);
      }
      if (stage != s->cmd_stage_seen) { s->cmd_stage_seen = stage; if (stage >= 0) { s->cmd_x_track[0] = s->cfg.cmd_curriculum[stage][1]; s->cmd_x_track[1] = s->cfg.cmd_curriculum[stage][2]; } }
    }
    s->cb_resamples = 0; }
  s->b.episode_info[GO2_NUM_REWARDS+1] = (float)s->cmd_x_track[0]; s->b.episode_info[GO2_NUM_REWARDS+2] = (float)s->cmd_x_track[1];
  memset(s->ep_sum,0,sizeof(s->ep_sum)); s->ep_count=0;
}

/* ------------------------------------------------------------------------------------------------
 * ABI
 * ---------------------------------------------------------------------------------------------- */
int go2sim_is_device_library(void) { return 0; }
int go2sim_buffer_layout(void) { return 0; }
const char* go2sim_last_error(void) { return g_err; }
void go2sim_default_cfg(Go2SimCfg* cfg) { go2sim_fill_default_cfg(cfg); }

#define ALLOC(field, type, count) do { s->b.field = (type*)calloc((size_t)(count), sizeof(type)); if (!s->b.field) { go2sim_destroy(s); return GO2SIM_ENOMEM; } } while (0)

/* torch.div(int64 tensor, python float, rounding_mode="floor") (:1072): fp32 arithmetic, c10::div_floor_floating
   (fmod, exact quotient, floor, round-to-nearest guard).  Matters: env 3 of 12 over 20 columns is column 4, not 5. */
static int64_t div_floor_f32(float a, float b) {
  float m = fmodf(a,b), d = (a-m)/b, f = 0;
  if (m != 0 && ((b<0) != (m<0))) d -= 1;
  if (d != 0) { f = floorf(d); if (d-f > 0.5f) f += 1; }
  return (int64_t)f;
}
int go2sim_create(const Go2SimCfg* cfg, int device_id, Go2Sim** out) {
  (void)device_id;
  if (!cfg || !out) { snprintf(g_err,sizeof(g_err),"null argument"); return GO2SIM_EINVAL; }
  if (cfg->struct_size != sizeof(Go2SimCfg) || cfg->abi_version != GO2SIM_ABI_VERSION) { snprintf(g_err,sizeof(g_err),"cfg size/version mismatch (%u vs %zu)", cfg->struct_size, sizeof(Go2SimCfg)); return GO2SIM_EINVAL; }
  if (cfg->num_envs <= 0 || cfg->decimation <= 0 || cfg->num_envs_global < cfg->env_offset + cfg->num_envs) { snprintf(g_err,sizeof(g_err),"bad num_envs/decimation"); return GO2SIM_EINVAL; }
  if (cfg->control_type < 0 || cfg->control_type > 2) { snprintf(g_err,sizeof(g_err),"control_type must be 0 (P), 1 (V) or 2 (T)"); return GO2SIM_EINVAL; }
  if (cfg->terrain_mode != 0 && (!cfg->hf_samples || !cfg->terrain_origins || !cfg->terrain_type_id)) { snprintf(g_err,sizeof(g_err),"heightfield terrain needs hf_samples, terrain_origins, terrain_type_id"); return GO2SIM_EINVAL; }
  Go2Sim* s = (Go2Sim*)calloc(1,sizeof(Go2Sim)); if (!s) return GO2SIM_ENOMEM;
  s->cfg = *cfg; int N = s->N = cfg->num_envs;
  ALLOC(root_states,float,N*13); ALLOC(dof_state,float,N*24); ALLOC(contact_forces,float,N*NB*3); ALLOC(rigid_body_states,float,N*NB*13);
  ALLOC(obs_buf,float,N*GO2_NUM_OBS); ALLOC(privileged_obs_buf,float,N*GO2_NUM_PRIV_OBS); ALLOC(rew_buf,float,N); ALLOC(reset_buf,uint8_t,N); ALLOC(time_out_buf,uint8_t,N);
  ALLOC(episode_length_buf,int64_t,N); ALLOC(torques,float,N*12); ALLOC(actions,float,N*12); ALLOC(last_actions,float,N*12); ALLOC(last_last_actions,float,N*12);
  ALLOC(last_dof_vel,float,N*12); ALLOC(last_root_vel,float,N*6); ALLOC(commands,float,N*4); ALLOC(commands_resampling_step,float,N); ALLOC(commands_xy_accumulation,float,N*2);
  ALLOC(stop_heading,uint8_t,N); ALLOC(last_is_limit_vel,uint8_t,N); ALLOC(base_lin_vel,float,N*3); ALLOC(base_ang_vel,float,N*3); ALLOC(projected_gravity,float,N*3); ALLOC(rpy,float,N*3);
  ALLOC(measured_heights,float,N*GO2_NUM_HEIGHT_POINTS); ALLOC(max_move_distance,float,N); ALLOC(feet_air_time,float,N*4); ALLOC(last_contacts,uint8_t,N*4); ALLOC(last_contacts2,uint8_t,N*4);
  ALLOC(motor_strengths,float,N*12); ALLOC(motor_zero_offsets,float,N*12); ALLOC(p_gains_multiplier,float,N*12); ALLOC(d_gains_multiplier,float,N*12);
  ALLOC(env_origins,float,N*3); ALLOC(terrain_levels,int64_t,N); ALLOC(terrain_types,int64_t,N); ALLOC(episode_sums,float,GO2_NUM_REWARDS*N);
  ALLOC(friction_coeffs,float,N); ALLOC(restitution_coeffs,float,N); ALLOC(added_base_mass,float,N); ALLOC(added_base_com,float,N*3); ALLOC(link_mass_ratio,float,N*18);
  ALLOC(turn_over_timer,float,N); ALLOC(episode_info,float,GO2_EPISODE_INFO_LEN); ALLOC(foot_impulse,float,N*12);
  s->terrain_kind = (int32_t*)malloc(sizeof(int32_t)*N); s->inj_storage = (float*)malloc(sizeof(float)*(size_t)N*GO2_NUM_UNIFORMS);
  s->dt = (R)cfg->decimation*(R)cfg->sim_dt;
  s->max_episode_length = (R)ceil((double)cfg->episode_length_s/(double)s->dt - 1e-3); /* np.ceil(25/0.02) = 1250 (:1104); the slack absorbs fp32 dt */
  for (int t=0;t<GO2_NUM_REWARDS;++t) { s->reward_scale_dt[t] = (R)cfg->reward_scales[t]*s->dt; s->reward_curr_scale[t]=1;
    s->turn_over_scale_dt[t] = cfg->turn_over ? (R)cfg->turn_over_scales[t]*s->dt : 0; }
  for (int i=0;i<cfg->reward_curriculum_count;++i) { int t=cfg->reward_curriculum_term[i]; s->reward_has_curr[t]=1; s->reward_curr_scale[t]=(R)cfg->reward_curriculum[i][2]; }
  for (int r=0;r<4;++r) { s->cmd_ranges[r][0]=(R)cfg->cmd_ranges[r][0]; s->cmd_ranges[r][1]=(R)cfg->cmd_ranges[r][1]; }
  s->cmd_x_track[0]=cfg->cmd_ranges[0][0]; s->cmd_x_track[1]=cfg->cmd_ranges[0][1]; s->cmd_stage_seen=-2;
  for (int j=0;j<12;++j) { /* soft limits (:372-375), computed in fp32 like the reference's torch tensors */
    float lo=(float)kJointLower[j], hi=(float)kJointUpper[j]; float m=(lo+hi)/2, r=hi-lo;
    s->dof_pos_limits[j][0]=(R)(m-0.5f*r*(float)cfg->soft_dof_pos_limit); s->dof_pos_limits[j][1]=(R)(m+0.5f*r*(float)cfg->soft_dof_pos_limit); }
  { /* Go2Robot._get_noise_scale_vec (go2_env.py:9-21) */
    R nl=(R)cfg->noise_level;
    for (int i=0;i<3;++i) { s->noise_vec[i]=(R)cfg->noise_ang_vel*nl*(R)cfg->obs_scale_ang_vel; s->noise_vec[3+i]=(R)cfg->noise_gravity*nl; s->noise_vec[6+i]=0; }
    for (int j=0;j<12;++j) { s->noise_vec[9+j]=(R)cfg->noise_dof_pos*nl*(R)cfg->obs_scale_dof_pos; s->noise_vec[21+j]=(R)cfg->noise_dof_vel*nl*(R)cfg->obs_scale_dof_vel; s->noise_vec[33+j]=0; } }
  { /* _init_height_points (:1172-1186) + scan mask (:790-796): x = -0.8..0.8 (17), y = -0.5..0.5 (11), x-major */
    int n=0; s->num_height_mask=0;
    for (int ix=0;ix<17;++ix) for (int iy=0;iy<11;++iy) { R x=(R)(ix-8)*RC(0.1), y=(R)(iy-5)*RC(0.1); s->height_pts[n][0]=x; s->height_pts[n][1]=y;
      s->height_mask[n] = (ix>=6 && ix<=10 && iy>=4 && iy<=6); /* |x|<=0.2, |y|<=0.15 on the 0.1 grid (fp-safe form) */ s->num_height_mask += s->height_mask[n]; ++n; } }
  if (cfg->terrain_mode != 0) {
    size_t nh=(size_t)cfg->hf_rows*cfg->hf_cols; s->hf=(int16_t*)malloc(nh*sizeof(int16_t)); memcpy(s->hf,cfg->hf_samples,nh*sizeof(int16_t));
    { size_t nr=(size_t)cfg->hf_rows-1, ncell=(size_t)cfg->hf_cols-1, C_=(size_t)cfg->hf_cols; s->cells=(int16_t*)malloc(nr*ncell*4*sizeof(int16_t));
      if (cfg->hf_cells) memcpy(s->cells, cfg->hf_cells, nr*ncell*4*sizeof(int16_t));
      else for (size_t i=0;i<nr;++i) for (size_t j=0;j<ncell;++j) { int16_t* q=s->cells+(i*ncell+j)*4; const int16_t* h=cfg->hf_samples;
        q[0]=h[i*C_+j]; q[1]=h[(i+1)*C_+j]; q[2]=h[i*C_+j+1]; q[3]=h[(i+1)*C_+j+1]; }
      s->walls = (cfg->hf_cells && cfg->hf_walls) ? 1 : 0; }
    size_t no=(size_t)cfg->terrain_num_levels*cfg->terrain_num_types*3; s->terrain_origins=(float*)malloc(no*sizeof(float)); for (size_t i=0;i<no;++i) s->terrain_origins[i]=(float)cfg->terrain_origins[i];
    s->terrain_type_id=(int32_t*)malloc(sizeof(int32_t)*cfg->terrain_num_types); memcpy(s->terrain_type_id,cfg->terrain_type_id,sizeof(int32_t)*cfg->terrain_num_types);
  }
  s->cfg.hf_samples=NULL; s->cfg.hf_cells=NULL; s->cfg.terrain_origins=NULL; s->cfg.terrain_type_id=NULL;
  /* per-env creation-time quantities (legged_robot.py:320-402, :1054-1091) */
  for (int b=0;b<64;++b) s->friction_buckets[b] = urange((R)u01(cfg->seed,GO2_ENV_GLOBAL,(uint32_t)b,GO2_STEP_INIT),(R)cfg->friction_range[0],(R)cfg->friction_range[1]);
  int Ng = cfg->num_envs_global;
  for (int e=0;e<N;++e) {
    uint32_t ge=(uint32_t)(cfg->env_offset+e);
    int bucket=(int)(u01(cfg->seed,ge,0,GO2_STEP_INIT)*64); if (bucket>63) bucket=63;
    s->b.friction_coeffs[e] = cfg->randomize_friction ? (float)s->friction_buckets[bucket] : 1.0f;
    s->b.restitution_coeffs[e] = cfg->randomize_restitution ? (float)urange((R)u01(cfg->seed,ge,1,GO2_STEP_INIT),(R)cfg->restitution_range[0],(R)cfg->restitution_range[1]) : 0.0f;
    s->b.added_base_mass[e] = cfg->randomize_base_mass ? (float)urange((R)u01(cfg->seed,ge,2,GO2_STEP_INIT),(R)cfg->added_mass_range[0],(R)cfg->added_mass_range[1]) : 0.0f;
    for (int k=0;k<3;++k) s->b.added_base_com[3*e+k] = cfg->randomize_base_com ? (float)urange((R)u01(cfg->seed,ge,3+k,GO2_STEP_INIT),(R)cfg->base_com_range[0],(R)cfg->base_com_range[1]) : 0.0f;
    for (int k=0;k<18;++k) s->b.link_mass_ratio[18*e+k] = cfg->randomize_link_mass ? (float)urange((R)u01(cfg->seed,ge,6+k,GO2_STEP_INIT),(R)cfg->link_mass_range[0],(R)cfg->link_mass_range[1]) : 1.0f;
    for (int j=0;j<12;++j) { s->b.motor_strengths[12*e+j]=1; s->b.p_gains_multiplier[12*e+j]=1; s->b.d_gains_multiplier[12*e+j]=1; }
    if (cfg->terrain_mode == 0) { /* grid (:1081-1091) */
      int ncols=(int)floor(sqrt((double)Ng)); s->b.env_origins[3*e]=(float)(cfg->env_spacing*(float)(ge/ncols)); s->b.env_origins[3*e+1]=(float)(cfg->env_spacing*(float)(ge%ncols)); s->b.env_origins[3*e+2]=0;
      s->terrain_kind[e]=-1;
    } else { /* round robin (:1071-1079) */
      int maxl = cfg->terrain_curriculum ? cfg->max_init_terrain_level : cfg->terrain_num_levels-1;
      s->b.terrain_levels[e] = ge % (uint32_t)(maxl+1);
      s->b.terrain_types[e] = div_floor_f32((float)ge, (float)((double)Ng/(double)cfg->terrain_num_types));
      const float* o = s->terrain_origins + ((size_t)s->b.terrain_levels[e]*cfg->terrain_num_types + s->b.terrain_types[e])*3;
      for (int i=0;i<3;++i) s->b.env_origins[3*e+i]=o[i];
      s->terrain_kind[e] = s->terrain_type_id[s->b.terrain_types[e]];
    }
    for (int i=0;i<13;++i) s->b.root_states[13*e+i]=cfg->base_init_state[i];
    for (int i=0;i<3;++i) s->b.root_states[13*e+i]+=s->b.env_origins[3*e+i];
    for (int j=0;j<12;++j) s->b.dof_state[24*e+2*j]=cfg->default_dof_pos[j];
    s->b.reset_buf[e]=1; /* base_task.py:43 */
  }
  update_command_scalars(s);
  *out = s; return 0;
}
void go2sim_destroy(Go2Sim* s) {
  if (!s) return;
  void** p = (void**)&s->b; for (size_t i=0;i<sizeof(Go2SimBuffers)/sizeof(void*);++i) free(p[i]);
  free(s->hf); free(s->cells); free(s->terrain_origins); free(s->terrain_type_id); free(s->terrain_kind); free(s->inj_storage); free(s);
}
int go2sim_get_buffers(Go2Sim* s, Go2SimBuffers* out) { if (!s||!out) return GO2SIM_EINVAL; *out = s->b; return 0; }

/* reset_idx(env_ids) called from outside a step (legged_robot.py:180-245) */
int go2sim_reset_idx(Go2Sim* s, const int32_t* env_ids, int32_t count, void* stream) {
  (void)stream; if (!s || count < 0 || (count > 0 && !env_ids)) return GO2SIM_EINVAL;
  if (count == 0) return 0;                                   /* :189-190 */
  update_command_scalars(s);
  uint8_t* hit = (uint8_t*)calloc((size_t)s->N, 1); if (!hit) return GO2SIM_ENOMEM;
  for (int i=0;i<count;++i) if (env_ids[i] >= 0 && env_ids[i] < s->N) hit[env_ids[i]] = 1;
  for (int e=0;e<s->N;++e) if (hit[e]) { reset_env(s, e, 0); s->b.reset_buf[e] = 1; }
  free(hit);
  finish_episode_info(s);
  s->injected=NULL; s->step_count++;
  return 0;
}
int go2sim_reset_all(Go2Sim* s, void* stream) {
  (void)stream; if (!s) return GO2SIM_EINVAL;
  update_command_scalars(s);
  for (int e=0;e<s->N;++e) reset_env(s, e, 1);
  finish_episode_info(s);
  s->injected=NULL; s->step_count++;
  return 0;
}

static void simulate_env(Go2Sim* s, int e) {
  const Go2SimCfg* c=&s->cfg; Go2SimBuffers* b=&s->b;
  R root[13], q[12], qd[12], tau[12], bf[NB*3];
  for (int i=0;i<13;++i) root[i]=(R)b->root_states[13*e+i];
  for (int j=0;j<12;++j) { q[j]=(R)b->dof_state[24*e+2*j]; qd[j]=(R)b->dof_state[24*e+2*j+1]; }
  int start = 0;
  if (c->randomize_action_delay) { start=(int)(uni(s,e,GO2_U_DELAY)*(c->decimation+1)); if (start>c->decimation) start=c->decimation; } /* :72 */
  Kin k;
  memset(bf,0,sizeof(bf));
  R glam[4][NSLOT][3]; memset(glam,0,sizeof(glam));
  /* EXPERIMENT (tools/solver_convergence.py --both, profiles/r4_solver_convergence.txt): GO2_ORACLE_WARM_GROUPS=1 warm-starts the non-foot slots from the
   * previous substep, matched by body group.  Measured: the 4-sweep body-force gap moves p90 12.7 % -> 10.3 %, p99 116 % -> 143 % — not what closes the gap
   * to the converged solve (16 sweeps do), so the shipped model (kernel and oracle) keeps the foot-only warm start and this stays off. */
  static int warm_groups = -1; if (warm_groups < 0) warm_groups = getenv("GO2_ORACLE_WARM_GROUPS") ? atoi(getenv("GO2_ORACLE_WARM_GROUPS")) : 0;
  for (int i=0;i<c->decimation;++i) {
    R a[12]; for (int j=0;j<12;++j) a[j] = (c->randomize_action_delay && i<start) ? (R)b->last_actions[12*e+j] : (R)b->actions[12*e+j]; /* :74-78 */
    pd_torques(s,e,q,qd,a,tau);
    physics_substep(s,e,root,q,qd,tau,bf,(i==c->decimation-1)?&k:NULL, warm_groups ? glam : NULL);
  }
  for (int i=0;i<13;++i) b->root_states[13*e+i]=(float)root[i];
  for (int j=0;j<12;++j) { b->dof_state[24*e+2*j]=(float)q[j]; b->dof_state[24*e+2*j+1]=(float)qd[j]; b->torques[12*e+j]=(float)tau[j]; }
  for (int i=0;i<NB*3;++i) b->contact_forces[(size_t)e*NB*3+i]=(float)bf[i];
  write_body_states(s,e,&k);
}

#include <time.h>
static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec*1e3 + t.tv_nsec*1e-6; }
int go2sim_simulate(Go2Sim* s, void* stream) {
  (void)stream; if (!s) return GO2SIM_EINVAL;
  double t0 = now_ms();
  #pragma omp parallel for schedule(static)
  for (int e=0;e<s->N;++e) simulate_env(s,e);
  if (s->timing) { s->time_ms += now_ms()-t0; s->time_launches++; }
  return 0;
}
int go2sim_notify_replayed(Go2Sim* s, int32_t steps) { (void)steps; return s ? 0 : GO2SIM_EINVAL; }   /* host library: nothing is replayed */
int go2sim_enable_timing(Go2Sim* s, int en) { if (!s) return GO2SIM_EINVAL; s->timing=en; s->time_ms=0; s->time_launches=0; return 0; }
int go2sim_kernel_time(Go2Sim* s, double* ms, int64_t* n) { if (!s||!ms||!n) return GO2SIM_EINVAL; *ms=s->time_ms; *n=s->time_launches; s->time_ms=0; s->time_launches=0; return 0; }
int go2sim_post_physics(Go2Sim* s, void* stream) {
  (void)stream; if (!s) return GO2SIM_EINVAL;
  s->common_step_counter += 1;            /* :112 */
  update_reward_curriculum(s, 0);         /* :117 */
  update_command_scalars(s);
  { /* The reference picks up a started command_range_curriculum stage inside _resample_commands, when called with >= 1 env (:433-446).
     * Resampled commands therefore always see the current stage (update_command_scalars above); the heading clip (:411-419), applied to
     * every env right after the callback's _resample_commands(resampling_env_ids) (:408-410), sees it only if that BATCH call had an env. */
    int64_t it = s->common_step_counter / s->cfg.num_steps_per_env; int stage = -1, any = 0;
    for (int i=0;i<s->cfg.cmd_curriculum_count;++i) if ((double)it >= s->cfg.cmd_curriculum[i][0] && (stage<0 || s->cfg.cmd_curriculum[i][0] > s->cfg.cmd_curriculum[stage][0])) stage=i;
    for (int e=0;e<s->N && !any;++e) any = ((R)s->b.commands_resampling_step[e]-1 <= 0) && ((R)(s->b.episode_length_buf[e]+1) < s->max_episode_length-1);
    s->yaw_seen = s->cfg.heading_command && stage != (s->cmd_stage_seen < 0 ? -1 : s->cmd_stage_seen) && !any; }
  for (int e=0;e<s->N;++e) post_physics_env(s,e);   /* serial: episode-info accumulation order */
  s->yaw_seen = 0;
  finish_episode_info(s);
  s->injected=NULL; s->step_count++;
  return 0;
}
int go2sim_step(Go2Sim* s, const float* actions, void* stream) {
  if (!s||!actions) return GO2SIM_EINVAL;
  R cl=(R)s->cfg.clip_actions;
  for (int i=0;i<s->N*12;++i) { R a=(R)actions[i]; if (a>cl) a=cl; if (a<-cl) a=-cl; s->b.actions[i]=(float)a; } /* :67-68 */
  const float* inj = s->injected;
  go2sim_simulate(s,stream);
  s->injected = inj;
  return go2sim_post_physics(s,stream);
}
/* go2sim_step + the rollout loop's bookkeeping of one policy step (on_policy_runner.py:135-153, ppo.py:104-114; see go2sim.h) */
int go2sim_step_rollout(Go2Sim* s, const float* actions, const Go2StepOutputs* out, void* stream) {
  int rc = go2sim_step(s, actions, stream);
  if (rc != 0 || !out) return rc;
  const Go2SimBuffers* b = &s->b; int N = s->N;
  if (out->obs_out) memcpy(out->obs_out, b->obs_buf, sizeof(float)*(size_t)N*GO2_NUM_OBS);
  if (out->priv_out) memcpy(out->priv_out, b->privileged_obs_buf, sizeof(float)*(size_t)N*GO2_NUM_PRIV_OBS);
  for (int e=0;e<N;++e) {
    if (out->rewards_out) out->rewards_out[e] = (float)((R)b->rew_buf[e] + ((out->values && b->time_out_buf[e]) ? (R)out->gamma*(R)out->values[e] : 0));
    if (out->dones_out) out->dones_out[e] = b->reset_buf[e];
  }
  if (out->episode_info_out) memcpy(out->episode_info_out, b->episode_info, sizeof(float)*GO2_EPISODE_INFO_LEN);
  return 0;
}
int go2sim_set_root_state_indexed(Go2Sim* s, const int32_t* ids, int32_t count, void* stream) { (void)ids;(void)count;(void)stream; return s ? 0 : GO2SIM_EINVAL; }   /* the API tensors are the state */
int go2sim_set_dof_state_indexed(Go2Sim* s, const int32_t* ids, int32_t count, void* stream) { (void)ids;(void)count;(void)stream; return s ? 0 : GO2SIM_EINVAL; }
int go2sim_set_common_step_counter(Go2Sim* s, int64_t v) { if (!s) return GO2SIM_EINVAL; s->common_step_counter=v; return 0; }
int64_t go2sim_get_common_step_counter(Go2Sim* s) { return s ? s->common_step_counter : -1; }
int go2sim_update_reward_curriculum(Go2Sim* s, int force) { if (!s) return GO2SIM_EINVAL; update_reward_curriculum(s,force); return 0; }
int go2sim_get_curriculum_state(Go2Sim* s, float* rcs, float cr[4][2], float* zp) {
  if (!s) return GO2SIM_EINVAL;
  if (rcs) for (int t=0;t<GO2_NUM_REWARDS;++t) rcs[t]=(float)s->reward_curr_scale[t];
  if (cr) for (int r=0;r<4;++r) { cr[r][0]=(float)s->cmd_ranges[r][0]; cr[r][1]=(float)s->cmd_ranges[r][1]; }
  if (zp) *zp=(float)s->zero_command_proba; return 0;
}
int go2sim_inject_uniforms(Go2Sim* s, const float* u, void* stream) {
  (void)stream; if (!s) return GO2SIM_EINVAL;
  if (!u) { s->injected=NULL; return 0; }
  memcpy(s->inj_storage,u,sizeof(float)*(size_t)s->N*GO2_NUM_UNIFORMS); s->injected=s->inj_storage; return 0;
}
int go2sim_peek_uniforms(Go2Sim* s, float* out, void* stream) {
  (void)stream; if (!s||!out) return GO2SIM_EINVAL;
  for (int e=0;e<s->N;++e) for (int k=0;k<GO2_NUM_UNIFORMS;++k) out[(size_t)e*GO2_NUM_UNIFORMS+k]=u01_step(s->cfg.seed,(uint32_t)(s->cfg.env_offset+e),(uint32_t)k,s->step_count);
  return 0;
}

/* RolloutStorage.compute_returns (rsl_rl/rsl_rl/storage/rollout_storage.py:123-137) */
int go2sim_gae(const float* rewards, const uint8_t* dones, const float* values, const float* last_values,
               float* returns, float* advantages, double* partials, int32_t T, int32_t N, float gamma, float lam, void* stream) {
  (void)stream; if (!rewards||!dones||!values||!last_values||!returns||!advantages||T<=0||N<=0) return GO2SIM_EINVAL;
  double s1=0,s2=0;
  for (int e=0;e<N;++e) {
    R adv=0;
    for (int t=T-1;t>=0;--t) {
      R nv = (t==T-1) ? (R)last_values[e] : (R)values[(size_t)(t+1)*N+e];
      R nt = 1 - (R)(dones[(size_t)t*N+e]!=0);
      R delta = (R)rewards[(size_t)t*N+e] + nt*(R)gamma*nv - (R)values[(size_t)t*N+e];
      adv = delta + nt*(R)gamma*(R)lam*adv;
      R ret = adv + (R)values[(size_t)t*N+e];
      returns[(size_t)t*N+e]=(float)ret;
      float a = (float)(ret - (R)values[(size_t)t*N+e]);   /* self.advantages = self.returns - self.values (:136) */
      advantages[(size_t)t*N+e]=a; s1+=(double)a; s2+=(double)a*(double)a;
    }
  }
  if (partials) { partials[0]+=s1; partials[1]+=s2; partials[2]+=(double)T*N; }
  return 0;
}
int go2sim_normalize_advantages(float* adv, const double* partials, int32_t count, void* stream) {
  (void)stream; if (!adv||!partials||count<=0) return GO2SIM_EINVAL;
  double n=partials[2], mean=partials[0]/n, var=(partials[1]-n*mean*mean)/(n-1); if (var<0) var=0;  /* torch.std is unbiased (:137) */
  double sd=sqrt(var);
  for (int i=0;i<count;++i) adv[i]=(float)(((double)adv[i]-mean)/(sd+1e-8));
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * oracle-only entry points for the known-answer tests (not part of the product ABI)
 * ---------------------------------------------------------------------------------------------- */
/* For env e at its current state with joint torques tau[12]: acc_aba[18] (ABA), acc_dense[18]
 * (M^-1 (tau - C)), M[18*18], energies {kinetic, potential, |linear momentum|... } */
int go2o_debug_dynamics(Go2Sim* s, int e, const float* tau_in, float* acc_aba, float* acc_dense, float* Mout, float* energy) {
  R root[13], q[12], qd[12], tau[12];
  for (int i=0;i<13;++i) root[i]=(R)s->b.root_states[13*e+i];
  for (int j=0;j<12;++j) { q[j]=(R)s->b.dof_state[24*e+2*j]; qd[j]=(R)s->b.dof_state[24*e+2*j+1]; tau[j]=(R)tau_in[j]; }
  Kin k; kinematics(s,e,root,q,qd,&k);
  R a1[NV]; aba(s,&k,qd,tau,a1);
  R M[NV*NV], C[NV], rhs[NV]; mass_matrix(s,&k,M); bias_forces(s,&k,qd,C);
  for (int i=0;i<NV*NV;++i) Mout[i]=(float)M[i];
  R nu[NV]; for (int i=0;i<6;++i) nu[i]=k.v[0][i]; for (int j=0;j<12;++j) nu[6+j]=qd[j];
  R ke=0; for (int i=0;i<NV;++i) for (int j=0;j<NV;++j) ke += RC(0.5)*nu[i]*M[NV*i+j]*nu[j];
  for (int i=0;i<6;++i) rhs[i]=-C[i]; for (int j=0;j<12;++j) rhs[6+j]=tau[j]-C[6+j];
  chol_factor(M,NV); chol_solve(M,NV,rhs);
  /* potential energy and world linear momentum */
  R pe=0, mom[3]={0,0,0}, tm=0;
  for (int l=0;l<NL;++l) { R m=k.I[l][35]; R h[3]={k.I[l][6*2+4], k.I[l][6*0+5], k.I[l][6*1+3]}; /* m*c from the m cx block */
    R hw[3]; mat3_vec(k.Rw[l],h,hw); pe += -(R)s->cfg.gravity[2]*(m*k.pw[l][2]+hw[2]); tm+=m;
    R vl[3], t[3]; cross3(k.v[l],h,t); for (int i=0;i<3;++i) vl[i]=m*k.v[l][3+i]+t[i]; R vw[3]; mat3_vec(k.Rw[l],vl,vw); for (int i=0;i<3;++i) mom[i]+=vw[i]; }
  for (int i=0;i<NV;++i) { acc_aba[i]=(float)a1[i]; acc_dense[i]=(float)rhs[i]; }
  energy[0]=(float)ke; energy[1]=(float)pe; energy[2]=(float)mom[0]; energy[3]=(float)mom[1]; energy[4]=(float)mom[2]; energy[5]=(float)tm;
  return 0;
}
/* _compute_torques for all envs with caller-supplied input actions (golden test of legged_robot.py:594-618,:80-81) */
int go2o_pd_torques(Go2Sim* s, const float* act_in, float* out) {
  for (int e=0;e<s->N;++e) { R q[12],qd[12],a[12],t[12];
    for (int j=0;j<12;++j) { q[j]=(R)s->b.dof_state[24*e+2*j]; qd[j]=(R)s->b.dof_state[24*e+2*j+1]; a[j]=(R)act_in[12*e+j]; }
    pd_torques(s,e,q,qd,a,t); for (int j=0;j<12;++j) out[12*e+j]=(float)t[j]; }
  return 0;
}
int go2o_sizeof_real(void) { return (int)sizeof(R); }
/* Golden test of legged_robot.py:67-81: clip the actions, draw the action delay and, for each substep i,
 * compute the torques from caller-supplied DOF states dof[D][N][12][2] (the 'fake physics').  Leaves the
 * clipped actions in buffers.actions and the last substep's torques in buffers.torques, like step(). */
int go2o_torque_trace(Go2Sim* s, const float* actions_raw, const float* dof, float* out) {
  const Go2SimCfg* c=&s->cfg; int N=s->N; R cl=(R)c->clip_actions;
  for (int i=0;i<N*12;++i) { R a=(R)actions_raw[i]; if (a>cl) a=cl; if (a<-cl) a=-cl; s->b.actions[i]=(float)a; }
  for (int e=0;e<N;++e) {
    int start=0; if (c->randomize_action_delay) { start=(int)(uni(s,e,GO2_U_DELAY)*(c->decimation+1)); if (start>c->decimation) start=c->decimation; }
    for (int i=0;i<c->decimation;++i) {
      R q[12],qd[12],a[12],t[12];
      for (int j=0;j<12;++j) { const float* d=dof+(((size_t)i*N+e)*12+j)*2; q[j]=(R)d[0]; qd[j]=(R)d[1];
        a[j] = (c->randomize_action_delay && i<start) ? (R)s->b.last_actions[12*e+j] : (R)s->b.actions[12*e+j]; }
      pd_torques(s,e,q,qd,a,t);
      for (int j=0;j<12;++j) { out[((size_t)i*N+e)*12+j]=(float)t[j]; s->b.torques[12*e+j]=(float)t[j]; }
    }
  }
  return 0;
}

int go2sim_debug_torque_trace(Go2Sim* s, const float* actions_raw, const float* dof, float* out, void* stream) {
  (void)stream; if (!s||!actions_raw||!dof||!out) return GO2SIM_EINVAL; return go2o_torque_trace(s, actions_raw, dof, out);
}
int go2sim_debug_contact_query(Go2Sim* s, const float* pts, float* out, int32_t n, void* stream) {
  (void)stream; if (!s||!pts||!out||n<=0) return GO2SIM_EINVAL;
  for (int i=0;i<n;++i) { R c[3]={(R)pts[4*i],(R)pts[4*i+1],(R)pts[4*i+2]}, gap, nn[3]; contact_query(s, c, (R)pts[4*i+3], &gap, nn);
    out[4*i]=(float)gap; out[4*i+1]=(float)nn[0]; out[4*i+2]=(float)nn[1]; out[4*i+3]=(float)nn[2]; }
  return 0;
}
int go2sim_debug_traffic_probe(const float* in, float* out, int32_t N, int32_t nread, int32_t nwrite, void* stream) {
  (void)stream; if (!in||!out||N<=0||nread<0||nwrite<0) return GO2SIM_EINVAL;
  for (int e=0;e<N;++e) { float acc=0; for (int f=0;f<nread;++f) acc+=in[(size_t)f*N+e]; for (int f=0;f<nwrite;++f) out[(size_t)f*N+e]=acc+(float)f; }
  return 0;
}
/* the IEEE operations themselves (gcc, x86-64 baseline: no FMA contraction, correctly rounded / and sqrt) */
int go2sim_debug_strict_ops(const float* a, const float* b, float* out, int32_t n, void* stream) {
  (void)stream; if (!a||!b||!out||n<=0) return GO2SIM_EINVAL;
  for (int i=0;i<n;++i) { volatile float x=a[i], y=b[i], y0=b[0]; out[i]=x*y; out[n+i]=x+y; out[2*(size_t)n+i]=x-y; out[3*(size_t)n+i]=x/y;
    out[4*(size_t)n+i]=SQRT(FABS(x)); out[5*(size_t)n+i]=x/y0; }
  return 0;
}

/* Fused PPO loss head, CPU restatement (ppo.py:131-170).  Same contract as the HIP kernel. */
int go2sim_ppo_loss(const float* mu, const float* std, const float* value, const float* actions, const float* old_mu, const float* old_sigma,
                    const float* old_logp, const float* adv, const float* tv, const float* ret, float* gmu, float* gstd, float* gval, float* stats,
                    float* workspace, int32_t B, int32_t A, float clip, float vcoef, float ecoef, int32_t use_clip_v, int32_t split, void* stream) {
  (void)stream; (void)workspace;
  if (split <= 0 || split >= B) split = 0;   /* plain PPO */
  if (!mu||!std||!value||!actions||!old_mu||!old_sigma||!old_logp||!adv||!tv||!ret||!gmu||!gstd||!gval||!stats||B<=0||A<=0||A>64) return GO2SIM_EINVAL;
  double s_sur=0, s_vl=0, s_kl=0, s_ent=0; double gs[64]; for (int j=0;j<A;++j) gs[j]=0;
  const R LOG2PI = RC(1.8378770664093453);
  R ent=0; for (int j=0;j<A;++j) ent += RC(0.5) + RC(0.5)*LOG2PI + (R)log((double)std[j]);
  for (int i=0;i<B;++i) {
    R lp=0, kl=0;
    for (int j=0;j<A;++j) { R sg=(R)std[j], d=(R)actions[(size_t)i*A+j]-(R)mu[(size_t)i*A+j];
      lp += -d*d/(2*sg*sg) - (R)log((double)sg) - RC(0.5)*LOG2PI;
      R so=(R)old_sigma[(size_t)i*A+j], dm=(R)old_mu[(size_t)i*A+j]-(R)mu[(size_t)i*A+j];
      kl += (R)log((double)(sg/so + RC(1e-5))) + (so*so + dm*dm)/(2*sg*sg) - RC(0.5); }
    R ratio = EXP(lp-(R)old_logp[i]), a=(R)adv[i];
    R lo=1-(R)clip, hi=1+(R)clip, rc = ratio<lo?lo:(ratio>hi?hi:ratio); int in = ratio>=lo && ratio<=hi;
    R s1=-a*ratio, s2=-a*rc, sur = s1>s2?s1:s2;
    R w = s1>s2 ? 1 : (s1<s2 ? (R)in : RC(0.5)+RC(0.5)*(R)in);
    /* cts.py:228-231: mean over the teacher rows + mean over the student rows */
    R wr = split ? (i < split ? 1/(R)split : 1/(R)(B-split)) : 1/(R)B;
    R g_lp = -a*w*ratio*wr;
    R v=(R)value[i], dv=v-(R)tv[i], vl, gv;
    if (use_clip_v) { R dc = dv<-(R)clip?-(R)clip:(dv>(R)clip?(R)clip:dv); int vin = dv>=-(R)clip && dv<=(R)clip; R vc=(R)tv[i]+dc;
      R l1=(v-(R)ret[i])*(v-(R)ret[i]), l2=(vc-(R)ret[i])*(vc-(R)ret[i]); vl = l1>l2?l1:l2;
      R g1=2*(v-(R)ret[i]), g2=2*(vc-(R)ret[i])*(R)vin; gv = l1>l2 ? g1 : (l1<l2 ? g2 : RC(0.5)*g1+RC(0.5)*g2); }
    else { vl=((R)ret[i]-v)*((R)ret[i]-v); gv=2*(v-(R)ret[i]); }
    gval[i] = (float)((R)vcoef*gv/(R)B);
    for (int j=0;j<A;++j) { R sg=(R)std[j], d=(R)actions[(size_t)i*A+j]-(R)mu[(size_t)i*A+j];
      gmu[(size_t)i*A+j] = (float)(g_lp*d/(sg*sg));
      gs[j] += (double)(g_lp*(d*d/(sg*sg*sg) - 1/sg)); }
    s_sur += sur*wr; s_vl += vl; s_kl += kl; s_ent += ent;
  }
  for (int j=0;j<A;++j) gstd[j] = (float)(gs[j] - (double)ecoef/(double)std[j]);
  stats[0]=(float)(s_sur); stats[1]=(float)(s_vl/B); stats[2]=(float)(s_kl/B); stats[3]=(float)(s_ent/B);
  stats[4]=(float)(s_sur + (double)vcoef*s_vl/B - (double)ecoef*s_ent/B);
  return 0;
}

/* elu_backward (result form, alpha = 1) + column sum = the bias gradient of the preceding Linear */
int go2sim_elu_backward_bias(const float* gy, const float* y, float* gz, float* gb, float* workspace, int32_t B, int32_t C, void* stream) {
  (void)stream; (void)workspace;
  if (!gy || !y || !gz || !gb || B<=0 || C<=0 || (C&3)) return GO2SIM_EINVAL;
  double* acc = (double*)calloc((size_t)C, sizeof(double));
  for (int r=0;r<B;++r) for (int c=0;c<C;++c) { size_t k=(size_t)r*C+c; float o = gy[k]*(y[k] > 0 ? 1.0f : y[k]+1.0f); gz[k]=o; acc[c]+=o; }
  for (int c=0;c<C;++c) gb[c]=(float)acc[c];
  free(acc);
  return 0;
}

/* clip_grad_norm_ + adaptive-KL learning rate + torch.optim.Adam over a list of tensors (go2sim.h; ppo.py:140-155,178-181), plain loops */
int go2sim_adam_workspace_len(const Go2AdamTensors* t) {
  if (!t || t->count<=0 || t->count>GO2_ADAM_MAX_TENSORS) return GO2SIM_EINVAL;
  int nb=0; for (int i=0;i<t->count;++i) { if (t->numel[i]<=0) return GO2SIM_EINVAL; nb += (t->numel[i]+GO2_ADAM_CHUNK-1)/GO2_ADAM_CHUNK; }
  return nb;
}
int go2sim_adam_clip_step(const Go2AdamTensors* t, float* lr, const float* kl_mean, float desired_kl, float max_grad_norm, double beta1, double beta2, double eps,
                          float* workspace, void* stream) {
  (void)stream; (void)workspace;
  if (go2sim_adam_workspace_len(t) < 0 || !lr) return GO2SIM_EINVAL;
  if (kl_mean) { double kl=*kl_mean, r=*lr, d=desired_kl;
    if (kl > d*2.0) r = r/1.5 > 1e-5 ? r/1.5 : 1e-5; else if (kl < d/2.0 && kl > 0.0) r = r*1.5 < 1e-2 ? r*1.5 : 1e-2;
    *lr = (float)r; }
  double ss=0; for (int i=0;i<t->count;++i) for (int k=0;k<t->numel[i];++k) ss += (double)t->grad[i][k]*(double)t->grad[i][k];
  double coef = (double)max_grad_norm/(sqrt(ss)+1e-6); if (coef > 1.0) coef = 1.0;
  for (int i=0;i<t->count;++i) {
    /* every tensor's own step count, as torch.optim.Adam keeps it (a parameter that got no gradient in some step lags behind) */
    double st = (double)t->step[i][0]+1.0, bc1 = 1.0-pow((double)beta1,st), bc2 = 1.0-pow((double)beta2,st), step_size = (double)*lr/bc1;
    for (int k=0;k<t->numel[i];++k) {
      double g = (double)t->grad[i][k]*coef, m = (double)t->exp_avg[i][k], v = (double)t->exp_avg_sq[i][k];
      m = m + (g-m)*(1.0-(double)beta1); v = (double)beta2*v + (1.0-(double)beta2)*g*g;
      t->exp_avg[i][k]=(float)m; t->exp_avg_sq[i][k]=(float)v;
      t->param[i][k] = (float)((double)t->param[i][k] - step_size*m/(sqrt(v)/sqrt(bc2)+(double)eps));
    }
    t->step[i][0] += 1;
  }
  return 0;
}

/* PPO.act's sampling head (ppo.py:90-102; torch.distributions.Normal.log_prob: -((x-mu)^2)/(2 var) - log(scale) - log(sqrt(2 pi))) */
int go2sim_act_head(const float* mu, const float* std, const float* eps, const float* value, float* a_out, float* a_st, float* mu_st, float* sig_st, float* lp_st,
                    float* v_st, int32_t N, int32_t A, void* stream) {
  (void)stream;
  if (!mu || !std || !eps || !a_out || (v_st && !value) || N<=0 || A<=0) return GO2SIM_EINVAL;
  for (int e=0;e<N;++e) {
    R lp=0;
    for (int j=0;j<A;++j) {
      size_t k=(size_t)e*A+j;
      float a = mu[k] + std[j]*eps[k];              /* the sample is formed in fp32, as torch forms it */
      R d=(R)a-(R)mu[k], sg=(R)std[j];
      lp += -(d*d)/(2*sg*sg) - (R)log((double)sg) - RC(0.9189385332046727);
      a_out[k]=a; if (a_st) a_st[k]=a; if (mu_st) mu_st[k]=mu[k]; if (sig_st) sig_st[k]=std[j];
    }
    if (lp_st) lp_st[e]=(float)lp;
    if (v_st) v_st[e]=value[e];
  }
  return 0;
}
/* PPO.process_env_step (ppo.py:104-114) */
int go2sim_store_transition(const float* rew, const uint8_t* dones, const uint8_t* touts, const float* v_st, float* rew_st, uint8_t* dones_st, float gamma, int32_t N, void* stream) {
  (void)stream;
  if (!rew || !dones || !rew_st || !dones_st || (touts && !v_st) || N<=0) return GO2SIM_EINVAL;
  for (int e=0;e<N;++e) { float r=rew[e]; if (touts) r += gamma*(v_st[e]*(touts[e]?1.0f:0.0f)); rew_st[e]=r; dones_st[e]=dones[e]; }
  return 0;
}

/* on_policy_runner_cts.py:155-156 restated: zero the rows of finished envs, drop the oldest frame, append obs. */
#include "../include/go2sim_shuffle.h"
uint32_t go2sim_shuffle_index(uint32_t i, uint32_t n, uint32_t seed, uint32_t counter) { return n ? go2_shuffle_index(i, n, go2_shuffle_half_bits(n), seed, counter) : 0; }
/* rollout_storage.py:147-183: one permutation, the storage tensors gathered into mini-batch order (see include/go2sim.h) */
int go2sim_shuffle_gather(const Go2GatherJob* jobs, int32_t njobs, int32_t rows, const int64_t* indices, uint32_t* key_state, float* clear, int32_t nclear, void* stream) {
  (void)stream;
  if (!jobs || njobs<=0 || njobs>GO2_GATHER_MAX_JOBS || rows<=0 || (!indices && !key_state) || (nclear>0 && !clear)) return GO2SIM_EINVAL;
  for (int j=0;j<njobs;++j) if (!jobs[j].src || !jobs[j].dst || jobs[j].row_floats<=0 || (jobs[j].dst_pitch!=0 && jobs[j].dst_pitch<jobs[j].row_floats)) return GO2SIM_EINVAL;
  const int h = go2_shuffle_half_bits((uint32_t)rows);
  for (int32_t r=0;r<rows;++r) {
    const int64_t sidx = indices ? indices[r] : (int64_t)go2_shuffle_index((uint32_t)r, (uint32_t)rows, h, key_state[0], key_state[1]);
    for (int j=0;j<njobs;++j) memcpy(jobs[j].dst+(size_t)r*(jobs[j].dst_pitch?jobs[j].dst_pitch:jobs[j].row_floats), jobs[j].src+(size_t)sidx*jobs[j].row_floats, sizeof(float)*(size_t)jobs[j].row_floats);
  }
  if (!indices) key_state[1] += 1u;
  for (int i=0;i<nclear;++i) clear[i]=0.f;
  return 0;
}

int go2sim_cts_minibatch_indices(int64_t* out, int32_t nmb, int32_t nt, int32_t ns, const int64_t* map, uint32_t* key_state, void* stream) {
  /* rollout_storage_cts.py:152-160 with the keyed permutations of include/go2sim_shuffle.h in place of the two torch.randperm draws */
  (void)stream;
  if (!out || !key_state || nmb<=0 || nt<nmb || ns<nmb) return GO2SIM_EINVAL;
  const int tb = nt/nmb, sb = ns/nmb, ht = go2_shuffle_half_bits((uint32_t)nt), hs = go2_shuffle_half_bits((uint32_t)ns);
  for (int i=0;i<nmb;++i) for (int j=0;j<tb+sb;++j) {
    const int64_t k = j<tb ? (int64_t)go2_shuffle_index((uint32_t)(i*tb+j), (uint32_t)nt, ht, key_state[0], key_state[1])
                           : (int64_t)nt + (int64_t)go2_shuffle_index((uint32_t)(i*sb+j-tb), (uint32_t)ns, hs, key_state[0]^GO2_SHUFFLE_TAIL_SEED, key_state[1]);
    out[(size_t)i*(tb+sb)+j] = map ? map[k] : k;
  }
  key_state[1] += 1u;
  return 0;
}

int go2sim_history_push(float* history, const float* obs, const uint8_t* dones, int32_t N, int32_t H, int32_t D, void* stream) {
  (void)stream;
  if (!history || !obs || N<=0 || H<=0 || D<=0) return GO2SIM_EINVAL;
  for (int e=0;e<N;++e) {
    float* h = history + (size_t)e*H*D;
    if (dones && dones[e]) memset(h, 0, sizeof(float)*(size_t)H*D);
    memmove(h, h+D, sizeof(float)*(size_t)(H-1)*D);
    memcpy(h+(size_t)(H-1)*D, obs+(size_t)e*D, sizeof(float)*D);
  }
  return 0;
}
