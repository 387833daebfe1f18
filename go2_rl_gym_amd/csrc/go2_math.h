// go2_math.h — small fixed-size algebra for the Go2 lane programs (device code; also compiled for the
// host by the lane-emulation test build, hence the GO2_HD macro).
//
// Spatial vectors are pairs (ang, lin) of V3 expressed in ONE frame for the whole robot: the inertial
// frame instantaneously coincident with the base frame, origin at the base origin.  In that frame every
// Plücker transform is the identity, a rigid-body inertia is 10 numbers {m, h = m c, J about the origin},
// and composite inertias / forces of the four legs add with plain sums — which is what makes the
// 4-lanes-per-env reduction a handful of quad shuffles.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GO2_HD __host__ __device__ __forceinline__
#else
#define GO2_HD inline
#endif

// static-profile markers (tools/isa_profile.py): comments in the ISA, nothing at run time
#if defined(__HIP_DEVICE_COMPILE__) && defined(GO2_ISA_MARKS)
#define GO2_MARK(n) asm volatile("; GO2MARK " #n)
#else
#define GO2_MARK(n) do { } while (0)
#endif

// a point the compiler does not move LDS / global loads across (device build; nothing at run time): keeps a table value from being loaded long before its use —
// the step kernel sits at the register file's limit
#if defined(__HIP_DEVICE_COMPILE__)
#define GO2_LOAD_FENCE() asm volatile("" ::: "memory")
#else
#define GO2_LOAD_FENCE() do { } while (0)
#endif

// ---- individually rounded fp32 operations -------------------------------------------------------------------------------
// INDEX arithmetic that the reference does in eager torch (one correctly rounded IEEE operation per tensor op) must come out
// bit-identical here: a sample that lands an ulp on the other side of a cell boundary reads a different height.  The device build
// uses -ffast-math (FMA contraction, rcp-based division, 1-ulp sqrt; pragmas do not switch contraction off on this target), so
// these are spelled so that the compiler cannot touch them: single VALU instructions through inline asm, division / sqrt by the
// exactly-rounded constructions below.  Host builds (oracle-side emulation, x86-64 without FMA) use the plain operators.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float go2_mul_rn(float a, float b) { float r; asm("v_mul_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float go2_add_rn(float a, float b) { float r; asm("v_add_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float go2_sub_rn(float a, float b) { float r; asm("v_sub_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// a / b: the quotient formed in fp64 and rounded once more to fp32 is the correctly rounded fp32 quotient (53 >= 2*24 + 2 bits; the
// exact quotient of two fp32 numbers stays >= 2^-49 (relative) away from every fp32 rounding boundary, the fp64 result is within 2^-51).
// The widened operands pass through empty asm statements: -ffast-math would otherwise narrow fptrunc(fdiv(fpext a, fpext b)) back to
// an (rcp-based) fp32 division.
__device__ __forceinline__ double go2_opaque(double x) { asm("" : "+v"(x)); return x; }
__device__ __forceinline__ float go2_div_rn(float a, float b) { return (float)go2_opaque(go2_opaque((double)a) / go2_opaque((double)b)); }
__device__ __forceinline__ float go2_mul_inv_rn(float a, double inv_b) { return (float)go2_opaque(go2_opaque((double)a) * inv_b); }   // a / b with inv_b = 1.0 / (double)b
// sqrt: the hardware's 1-ulp v_sqrt_f32 moved to the correctly rounded neighbour by two exact FMA residual tests
__device__ __forceinline__ float go2_sqrt_rn(float x) {
  float y = __builtin_amdgcn_sqrtf(x);
  const float yd = __int_as_float(__float_as_int(y) - 1), yu = __int_as_float(__float_as_int(y) + 1);
  const float vp = __builtin_fmaf(-yd, y, x), vs = __builtin_fmaf(-yu, y, x);
  y = vp <= 0.f ? yd : y; y = vs > 0.f ? yu : y;
  return x == 0.f ? 0.f : y;
}
#else
GO2_HD float go2_mul_rn(float a, float b) { return a * b; }
GO2_HD float go2_add_rn(float a, float b) { return a + b; }
GO2_HD float go2_sub_rn(float a, float b) { return a - b; }
GO2_HD float go2_div_rn(float a, float b) { return a / b; }
GO2_HD float go2_mul_inv_rn(float a, double inv_b) { return (float)((double)a * inv_b); }
GO2_HD float go2_sqrt_rn(float x) { return sqrtf(x); }
#endif

// sine and cosine of a joint / half rotation angle (|x| < a few pi).  Device: the hardware's v_sin_f32 / v_cos_f32 on x / 2pi (absolute
// error ~1e-6 over this range: far below the fp32 noise of the dynamics it feeds; the libm-grade sincosf costs ~130 instructions and
// there are 21 of them per env step).  Host builds use libm.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void go2_sincos(float x, float* s, float* c) { const float r = x * 0.15915494309189535f; *s = __builtin_amdgcn_sinf(r); *c = __builtin_amdgcn_cosf(r); }
#else
GO2_HD void go2_sincos(float x, float* s, float* c) { *s = sinf(x); *c = cosf(x); }
#endif

// a value that is the same in every lane of the wave, as a scalar (lets the compiler branch on it with s_cbranch)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint32_t go2_uniform_u32(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
#else
GO2_HD uint32_t go2_uniform_u32(uint32_t x) { return x; }
#endif
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float go2_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
#else
GO2_HD float go2_rsqrt(float x) { return 1.0f / sqrtf(x); }
#endif

struct V3 { float x, y, z; };
GO2_HD V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
GO2_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
GO2_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
GO2_HD V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
GO2_HD V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
GO2_HD V3 operator*(V3 a, float s) { return v3(s * a.x, s * a.y, s * a.z); }
GO2_HD V3& operator+=(V3& a, V3 b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
GO2_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
GO2_HD V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
GO2_HD float comp(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
// component-wise select.  NEVER write `cond ? structA : structB` on aggregates in lane code: clang lowers it to a
// select of ADDRESSES, which pins both objects (and the whole lane context they live in) in scratch memory.
GO2_HD V3 sel(bool c, V3 a, V3 b) { return v3(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }

// 3x3 matrix by columns (x, y, z): M v = x v.x + y v.y + z v.z
struct M3 { V3 x, y, z; };
GO2_HD V3 mul(const M3& M, V3 v) { return M.x * v.x + M.y * v.y + M.z * v.z; }
GO2_HD V3 mulT(const M3& M, V3 v) { return v3(dot(M.x, v), dot(M.y, v), dot(M.z, v)); }
GO2_HD M3 quat_to_m3(float qx, float qy, float qz, float qw) {  // world_from_body, columns
  M3 R;
  R.x = v3(1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy + qz * qw), 2 * (qx * qz - qy * qw));
  R.y = v3(2 * (qx * qy - qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz + qx * qw));
  R.z = v3(2 * (qx * qz + qy * qw), 2 * (qy * qz - qx * qw), 1 - 2 * (qx * qx + qy * qy));
  return R;
}

// symmetric 3x3: xx, yy, zz, xy, xz, yz
struct S3 { float xx, yy, zz, xy, xz, yz; };
GO2_HD S3 operator+(S3 a, S3 b) { S3 r = {a.xx + b.xx, a.yy + b.yy, a.zz + b.zz, a.xy + b.xy, a.xz + b.xz, a.yz + b.yz}; return r; }
GO2_HD S3 operator*(float s, S3 a) { S3 r = {s * a.xx, s * a.yy, s * a.zz, s * a.xy, s * a.xz, s * a.yz}; return r; }
GO2_HD V3 mul(const S3& J, V3 v) { return v3(J.xx * v.x + J.xy * v.y + J.xz * v.z, J.xy * v.x + J.yy * v.y + J.yz * v.z, J.xz * v.x + J.yz * v.y + J.zz * v.z); }
// R J R^T
GO2_HD S3 rotate(const M3& R, const S3& J) {
  V3 c0 = R.x * J.xx + R.y * J.xy + R.z * J.xz;  // (R J) columns
  V3 c1 = R.x * J.xy + R.y * J.yy + R.z * J.yz;
  V3 c2 = R.x * J.xz + R.y * J.yz + R.z * J.zz;
  // (R J) R^T: element (i,j) = c0_i R.x_j + c1_i R.y_j + c2_i R.z_j
  S3 o;
  o.xx = c0.x * R.x.x + c1.x * R.y.x + c2.x * R.z.x;
  o.yy = c0.y * R.x.y + c1.y * R.y.y + c2.y * R.z.y;
  o.zz = c0.z * R.x.z + c1.z * R.y.z + c2.z * R.z.z;
  o.xy = c0.x * R.x.y + c1.x * R.y.y + c2.x * R.z.y;
  o.xz = c0.x * R.x.z + c1.x * R.y.z + c2.x * R.z.z;
  o.yz = c0.y * R.x.z + c1.y * R.y.z + c2.y * R.z.z;
  return o;
}

// spatial motion / force vector
struct SV { V3 a, l; };
GO2_HD SV sv(V3 a, V3 l) { SV r; r.a = a; r.l = l; return r; }
GO2_HD SV operator+(SV p, SV q) { return sv(p.a + q.a, p.l + q.l); }
GO2_HD SV operator-(SV p, SV q) { return sv(p.a - q.a, p.l - q.l); }
GO2_HD SV operator*(float s, SV p) { return sv(s * p.a, s * p.l); }
GO2_HD float dot(SV p, SV q) { return dot(p.a, q.a) + dot(p.l, q.l); }
GO2_HD SV crm(SV v, SV m) { return sv(cross(v.a, m.a), cross(v.a, m.l) + cross(v.l, m.a)); }   // v x m
GO2_HD SV crf(SV v, SV f) { return sv(cross(v.a, f.a) + cross(v.l, f.l), cross(v.a, f.l)); }   // v x* f
GO2_HD float get(const SV& s, int i) { return i < 3 ? comp(s.a, i) : comp(s.l, i - 3); }

// rigid-body inertia about the common origin: m, h = m c, J
struct RB { float m; V3 h; S3 J; };
GO2_HD RB operator+(RB a, RB b) { RB r; r.m = a.m + b.m; r.h = a.h + b.h; r.J = a.J + b.J; return r; }
GO2_HD RB operator*(float s, RB a) { RB r; r.m = s * a.m; r.h = s * a.h; r.J = s * a.J; return r; }
GO2_HD SV mul(const RB& I, SV v) { return sv(mul(I.J, v.a) + cross(I.h, v.l), I.m * v.l - cross(I.h, v.a)); }
// move an inertia given in a link frame (about the link origin, link axes) to the common frame:
// link origin at p, link axes R (columns)
GO2_HD RB to_common(const RB& L, const M3& R, V3 p) {
  RB o; o.m = L.m;
  V3 hr = mul(R, L.h);
  o.h = hr + L.m * p;
  S3 J = rotate(R, L.J);
  float pp = dot(p, p), ph = dot(p, hr);
  J.xx += L.m * (pp - p.x * p.x) + 2 * ph - 2 * p.x * hr.x;
  J.yy += L.m * (pp - p.y * p.y) + 2 * ph - 2 * p.y * hr.y;
  J.zz += L.m * (pp - p.z * p.z) + 2 * ph - 2 * p.z * hr.z;
  J.xy += -L.m * p.x * p.y - p.x * hr.y - hr.x * p.y;
  J.xz += -L.m * p.x * p.z - p.x * hr.z - hr.x * p.z;
  J.yz += -L.m * p.y * p.z - p.y * hr.z - hr.y * p.z;
  o.J = J; return o;
}
// {m, com c, inertia about the COM} -> about the frame origin
GO2_HD RB rb_from_com(float m, V3 c, S3 Ic) {
  RB o; o.m = m; o.h = m * c; float cc = dot(c, c);
  o.J.xx = Ic.xx + m * (cc - c.x * c.x); o.J.yy = Ic.yy + m * (cc - c.y * c.y); o.J.zz = Ic.zz + m * (cc - c.z * c.z);
  o.J.xy = Ic.xy - m * c.x * c.y; o.J.xz = Ic.xz - m * c.x * c.z; o.J.yz = Ic.yz - m * c.y * c.z;
  return o;
}

// ---- 6x6 symmetric positive definite: packed lower triangle, index (i,j), i >= j --------------------
#define S6(i, j) ((i) * ((i) + 1) / 2 + (j))
// inverse of an SPD 6x6 given packed (21) -> packed (21); fully unrolled so everything stays in registers
GO2_HD void spd6_inverse(const float* A, float* Ainv) {
  float L[21];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    float d = A[S6(j, j)];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= L[S6(j, k)] * L[S6(j, k)];
    d = sqrtf(fmaxf(d, 1e-20f));
    float inv = 1.0f / d;
    L[S6(j, j)] = d;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      float s = A[S6(i, j)];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= L[S6(i, k)] * L[S6(j, k)];
      L[S6(i, j)] = s * inv;
    }
  }
  // M = L^-1 (lower)
  float M[21];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    M[S6(j, j)] = 1.0f / L[S6(j, j)];
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      float s = 0;
#pragma unroll
      for (int k = j; k < i; ++k) s -= L[S6(i, k)] * M[S6(k, j)];
      M[S6(i, j)] = s / L[S6(i, i)];
    }
  }
  // Ainv = M^T M
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      float s = 0;
#pragma unroll
      for (int k = i; k < 6; ++k) s += M[S6(k, i)] * M[S6(k, j)];
      Ainv[S6(i, j)] = s;
    }
}
GO2_HD float s6at(const float* A, int i, int j) { return i >= j ? A[S6(i, j)] : A[S6(j, i)]; }
GO2_HD SV spd6_mul(const float* A, SV v) {
  float x[6] = {v.a.x, v.a.y, v.a.z, v.l.x, v.l.y, v.l.z}, o[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float s = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) s += s6at(A, i, j) * x[j];
    o[i] = s;
  }
  return sv(v3(o[0], o[1], o[2]), v3(o[3], o[4], o[5]));
}
// packed 6x6 of a rigid-body inertia [[J, hx],[hx^T, m 1]]
GO2_HD void rb_to_s6(const RB& I, float* A) {
  A[S6(0, 0)] = I.J.xx; A[S6(1, 0)] = I.J.xy; A[S6(1, 1)] = I.J.yy; A[S6(2, 0)] = I.J.xz; A[S6(2, 1)] = I.J.yz; A[S6(2, 2)] = I.J.zz;
  // rows 3..5 (lin), cols 0..2 (ang): hx^T = -hx ;  hx = [[0,-hz,hy],[hz,0,-hx],[-hy,hx,0]]
  A[S6(3, 0)] = 0;      A[S6(3, 1)] = I.h.z;  A[S6(3, 2)] = -I.h.y;
  A[S6(4, 0)] = -I.h.z; A[S6(4, 1)] = 0;      A[S6(4, 2)] = I.h.x;
  A[S6(5, 0)] = I.h.y;  A[S6(5, 1)] = -I.h.x; A[S6(5, 2)] = 0;
  A[S6(3, 3)] = I.m; A[S6(4, 3)] = 0; A[S6(4, 4)] = I.m; A[S6(5, 3)] = 0; A[S6(5, 4)] = 0; A[S6(5, 5)] = I.m;
}

// sin and cos of |x| <= pi/2 to fp32 round-off (about 1 ulp), branch-free: the reset path's half-angle quaternion must agree with the
// reference's torch.sin / torch.cos to 1e-6, which the hardware v_sin_f32 / v_cos_f32 (go2_sincos) do not promise.  Reduction to
// [0, pi/4] by sin(x) = cos(pi/2 - x), then the minimax polynomials of the classic fdlibm float kernels.
GO2_HD void go2_sincos_half_pi(float x, float* s, float* c) {
  const float ax = fabsf(x);
  const bool swap = ax > 0.78539816339744831f;
  const float y = swap ? (1.57079632679489662f - ax) + (-4.37113883e-8f) : ax;      // pi/2 = fl(pi/2) - 4.37e-8
  const float z = y * y;
  const float sn = y + y * z * (-0.166666666416265235595f + z * (0.0083333293858894631756f + z * (-0.000198393348360966317347f + z * 0.0000027183114939898219064f)));
  const float cs = 1.f + z * (-0.499999997251031003120f + z * (0.0416666233237390631894f + z * (-0.00138867637746099294692f + z * 0.0000243904487962774090654f)));
  const float sa = swap ? cs : sn;
  *s = x < 0.f ? -sa : sa; *c = swap ? sn : cs;
}

// ---- Philox4x32-10 (Salmon et al. SC'11): the same generator, key and counter layout as the oracle ----
GO2_HD void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
GO2_HD float u01_from_bits(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
// word k (0..3) of a Philox block without indexing a private array dynamically (that would live in scratch memory)
GO2_HD float philox_u01(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, int k) {
  uint32_t r[4];
  philox4x32_10(c0, c1, c2, c3, k0, k1, r);
  uint32_t x = k == 0 ? r[0] : (k == 1 ? r[1] : (k == 2 ? r[2] : r[3]));
  return u01_from_bits(x);
}
