// go2nn_cts.h — the row-wise pieces of the Concurrent Teacher-Student networks around the grouped GEMMs (included by go2nn_impl.cpp; include/go2nn.h ABI 5).
//
// CTS (rsl_rl/rsl_rl/modules/actor_critic_cts.py:49-80,146-176; rsl_rl/rsl_rl/algorithms/cts.py:167-285) puts a 32-wide L2-normalised latent in front of the actor and
// the critic: latent = F.normalize(encoder(x)) (modules/utils.py:24-30), actor input = cat([latent, obs]), critic input = cat([latent.detach(), privileged obs]).  In
// autograd that is a norm reduction, a clamp, a division, two cats and their backward nodes per mini-batch — 15-20 small launches on [B, 32] tensors.  Here:
//   go2nn_latent_concat_kernel   z -> zhat written straight into the first L columns of the two input matrices (+ 1 / |z| kept for the backward pass)
//   go2nn_l2norm_bwd_kernel      dz = (g - zhat (zhat . g)) / |z| from the first L columns of the actor's plain input gradient, + column partial sums (bias gradient)
//   go2nn_latent_mse_kernel      the student step's loss head: both normalisations, the MSE, its gradient through the student's normalisation, column partial sums
// A row is LP = L / 4 lanes (a float4 each; LP a power of two up to 32: L = 4 .. 128), row sums by an xor butterfly inside the LP-lane group; per-workgroup partial rows
// are formed in a fixed order (row lanes through LDS) and finished by go2nn_sum_rows: bit-reproducible.
#pragma once

#define CTS_EPS 1e-12f           /* F.normalize's eps */
#define CTS_ROWS_PER_WG 64       /* rows of one workgroup of the two backward kernels (one partial row each): 18432 teacher rows = 288 workgroups (512 rows = 36 workgroups took 18 us) */
static inline int cts_ok(int n, int L) { return n > 0 && L >= 4 && L <= 128 && (L & 3) == 0 && ((L >> 2) & ((L >> 2) - 1)) == 0; }
static inline int cts_rows(int n) { return (n + CTS_ROWS_PER_WG - 1) / CTS_ROWS_PER_WG; }

#ifndef GO2_EMU
__device__ __forceinline__ float cts_group_sum(float v, int LP) {
  for (int d = 1; d < LP; d <<= 1) v += __shfl_xor(v, d);
  return v;
}
__device__ __forceinline__ float4 cts_ld4(const float* p, bool vec) {
  if (vec) return *reinterpret_cast<const float4*>(p);
  const GmF4u v = *reinterpret_cast<const GmF4u*>(p); return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void cts_st4(float* p, const float4& v, bool vec) {
  if (vec) *reinterpret_cast<float4*>(p) = v;
  else { GmF4u u = {v.x, v.y, v.z, v.w}; *reinterpret_cast<GmF4u*>(p) = u; }
}

__global__ void __launch_bounds__(256) go2nn_latent_concat_kernel(const float* __restrict__ z, int n, int L, float* __restrict__ da, int lda, float* __restrict__ db, int ldb,
                                                                  float* __restrict__ inv_norm) {
  const int LP = L >> 2, rpw = 256 / LP, c = (threadIdx.x % LP) * 4;
  const int r = blockIdx.x * rpw + threadIdx.x / LP;
  const int rc = min(r, n - 1);
  const float4 v = *reinterpret_cast<const float4*>(z + (size_t)rc * L + c);
  const float ss = cts_group_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w, LP);
  const float inv = 1.f / fmaxf(sqrtf(ss), CTS_EPS);
  if (r >= n) return;
  const float4 o = make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
  // (a destination row starts at a multiple of its pitch: 16-byte stores only when the pitch keeps rows aligned)
  if (da) cts_st4(da + (size_t)r * lda + c, o, (lda & 3) == 0 && ((uintptr_t)da & 15) == 0);
  if (db) cts_st4(db + (size_t)r * ldb + c, o, (ldb & 3) == 0 && ((uintptr_t)db & 15) == 0);
  if (inv_norm && c == 0) inv_norm[r] = inv;
}

// One workgroup = CTS_ROWS_PER_WG rows; thread = (column quad, row lane); the row lanes' column sums are combined through LDS in row-lane order.
template <bool MSE>
__global__ void __launch_bounds__(256) go2nn_cts_bwd_kernel(const float* __restrict__ g, int ldg, const float* __restrict__ zh, int ldz, const float* __restrict__ inv_norm,
                                                            const float* __restrict__ zs, const float* __restrict__ zt, float* __restrict__ dz, float* __restrict__ part,
                                                            int n, int L, float scale) {
  __shared__ float4 sh[256];
  __shared__ float shl[256];
  const int LP = L >> 2, RL = 256 / LP, cq = threadIdx.x % LP, rl = threadIdx.x / LP, c = cq * 4;
  const int r0 = blockIdx.x * CTS_ROWS_PER_WG, r1 = min(n, r0 + CTS_ROWS_PER_WG);
  const bool gvec = MSE || ((ldg & 3) == 0 && ((uintptr_t)g & 15) == 0), zvec = MSE || ((ldz & 3) == 0 && ((uintptr_t)zh & 15) == 0);
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
  float loss = 0.f;
  for (int rb = r0; rb < r1; rb += RL) {          // (uniform trip count: the butterflies need every lane of a row group)
    const int r = rb + rl, rc = min(r, r1 - 1);
    const bool live = r < r1;
    float4 gv, zv; float inv;
    if (MSE) {
      const float4 s = *reinterpret_cast<const float4*>(zs + (size_t)rc * L + c), t = *reinterpret_cast<const float4*>(zt + (size_t)rc * L + c);
      const float is = 1.f / fmaxf(sqrtf(cts_group_sum(s.x * s.x + s.y * s.y + s.z * s.z + s.w * s.w, LP)), CTS_EPS);
      const float it = 1.f / fmaxf(sqrtf(cts_group_sum(t.x * t.x + t.y * t.y + t.z * t.z + t.w * t.w, LP)), CTS_EPS);
      zv = make_float4(s.x * is, s.y * is, s.z * is, s.w * is);
      const float4 d = make_float4(t.x * it - zv.x, t.y * it - zv.y, t.z * it - zv.z, t.w * it - zv.w);          // that - shat
      if (live) loss += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
      gv = make_float4(-scale * d.x, -scale * d.y, -scale * d.z, -scale * d.w);                                   // d loss / d shat = 2 (shat - that) / (n L) (scale = 2 / (n L) x the caller's factor)
      inv = is;
    } else {
      gv = cts_ld4(g + (size_t)rc * ldg + c, gvec); zv = cts_ld4(zh + (size_t)rc * ldz + c, zvec); inv = inv_norm[rc];
    }
    const float dot = cts_group_sum(gv.x * zv.x + gv.y * zv.y + gv.z * zv.z + gv.w * zv.w, LP);
    const float4 o = make_float4((gv.x - zv.x * dot) * inv, (gv.y - zv.y * dot) * inv, (gv.z - zv.z * dot) * inv, (gv.w - zv.w * dot) * inv);
    if (live) { *reinterpret_cast<float4*>(dz + (size_t)r * L + c) = o; cs.x += o.x; cs.y += o.y; cs.z += o.z; cs.w += o.w; }
  }
  sh[threadIdx.x] = cs; if (MSE) shl[threadIdx.x] = loss;
  __syncthreads();
  float* prow = part + (size_t)blockIdx.x * (L + (MSE ? 4 : 0));          // MSE: [loss, 0, 0, 0 | column sums] (the sums stay 16-byte aligned behind the scalar)
  if (rl == 0) {
    float4 s = sh[cq];
    for (int j = 1; j < RL; ++j) { const float4 t = sh[j * LP + cq]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
    float* o = prow + (MSE ? 4 : 0) + c; o[0] = s.x; o[1] = s.y; o[2] = s.z; o[3] = s.w;
  }
  if (MSE && threadIdx.x == 0) {
    float s = 0.f;
    for (int j = 0; j < 256; ++j) s += shl[j];
    prow[0] = s / ((float)n * (float)L); prow[1] = prow[2] = prow[3] = 0.f;
  }
}

// ---- the loss head of the MoE student step (rsl_rl/rsl_rl/modules/utils.py:96-152 MoE / StudentMoEEncoder; rsl_rl/rsl_rl/algorithms/moe_cts.py:203-214) -------------
//   w = softmax(logits [n, E]);  y = sum_e w_e outs[:, e, :];  shat = y / max(|y|, 1e-12);  latent loss = mean((that - shat)^2);
//   usage = mean_rows(w);  load-balance loss = mean_e((usage_e - 1 / E)^2);  total = latent + coef * load balance
// In autograd: softmax, a broadcast product, two reductions, the normaliser, two losses and all their backward nodes — ~55 launches on [n, 8] / [n, 32] / [n, 8, 32]
// tensors (250 us of a 1.2 ms student step, profiles/r5b_go2_moe_cts_timeline.txt).  Two passes, because the load-balance gradient needs the batch mean of the gate:
//   go2nn_moe_usage_kernel   column partial sums of softmax(logits)                                  (finished by go2nn_sum_rows -> usage_sum [E])
//   go2nn_moe_mix_kernel     everything else for a row in registers; d loss / d outs [n, E, L] and d loss / d logits [n, E] out, loss partials
#define MOE_MAX_E 16
__global__ void __launch_bounds__(CTS_ROWS_PER_WG) go2nn_moe_usage_kernel(const float* __restrict__ logits, float* __restrict__ part, int n, int E) {
  __shared__ float sh[CTS_ROWS_PER_WG][MOE_MAX_E + 1];
  const int r = blockIdx.x * CTS_ROWS_PER_WG + threadIdx.x;
  float w[MOE_MAX_E];
  if (r < n) {
    float mx = -3.4e38f, sum = 0.f;
#pragma unroll
    for (int e = 0; e < MOE_MAX_E; ++e) { w[e] = e < E ? logits[(size_t)r * E + e] : -3.4e38f; mx = fmaxf(mx, w[e]); }
#pragma unroll
    for (int e = 0; e < MOE_MAX_E; ++e) { w[e] = e < E ? expf(w[e] - mx) : 0.f; sum += w[e]; }
    const float inv = 1.f / sum;
#pragma unroll
    for (int e = 0; e < MOE_MAX_E; ++e) sh[threadIdx.x][e] = w[e] * inv;
  } else {
#pragma unroll
    for (int e = 0; e < MOE_MAX_E; ++e) sh[threadIdx.x][e] = 0.f;
  }
  __syncthreads();
  if ((int)threadIdx.x < E) { float s = 0.f; for (int j = 0; j < CTS_ROWS_PER_WG; ++j) s += sh[j][threadIdx.x]; part[(size_t)blockIdx.x * E + threadIdx.x] = s; }
}

template <int ME>          // ME: 8 or 16 expert slots held in registers (E <= ME)
__global__ void __launch_bounds__(256) go2nn_moe_mix_kernel(const float* __restrict__ logits, const float* __restrict__ outs, const float* __restrict__ that,
                                                            const float* __restrict__ usage_sum, float* __restrict__ dlogits, float* __restrict__ douts, float* __restrict__ part,
                                                            int n, int E, int L, float coef, long long sr, long long se,          // outs / douts element (r, e, c) at r sr + e se + c
                                                            const float* __restrict__ bias, float* __restrict__ dbias_part) {       // optional: the experts' output bias [E, L] is added here
  __shared__ float shl[256];                                                                                                        // (and its gradient's partial rows [E L] left), so autograd sees a plain bmm
  __shared__ float4 shb[256];
  const int LP = L >> 2, RL = 256 / LP, cq = threadIdx.x % LP, rl = threadIdx.x / LP, c = cq * 4;
  const int r0 = blockIdx.x * CTS_ROWS_PER_WG, r1 = min(n, r0 + CTS_ROWS_PER_WG);
  const float invE = 1.f / (float)E, invn = 1.f / (float)n, scale = 2.f / ((float)n * (float)L);
  float lbg[ME];          // d (coef * load balance) / d w[row][e] = coef * 2 (usage_e - 1 / E) / (E n), the same for every row
#pragma unroll
  for (int e = 0; e < ME; ++e) lbg[e] = e < E ? coef * 2.f * (usage_sum[e] * invn - invE) * invE * invn : 0.f;
  float loss = 0.f;
  float4 bq[ME], db[ME];
#pragma unroll
  for (int e = 0; e < ME; ++e) { bq[e] = (bias && e < E) ? *reinterpret_cast<const float4*>(bias + (size_t)e * L + c) : make_float4(0.f, 0.f, 0.f, 0.f); db[e] = make_float4(0.f, 0.f, 0.f, 0.f); }
  for (int rb = r0; rb < r1; rb += RL) {
    const int r = rb + rl, rc = min(r, r1 - 1);
    const bool live = r < r1;
    float w[ME]; float4 o[ME];
    float mx = -3.4e38f, sum = 0.f;
#pragma unroll
    for (int e = 0; e < ME; ++e) {
      w[e] = e < E ? logits[(size_t)rc * E + e] : -3.4e38f; mx = fmaxf(mx, w[e]);
      o[e] = e < E ? *reinterpret_cast<const float4*>(outs + (size_t)rc * sr + (size_t)e * se + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      o[e].x += bq[e].x; o[e].y += bq[e].y; o[e].z += bq[e].z; o[e].w += bq[e].w;
    }
#pragma unroll
    for (int e = 0; e < ME; ++e) { w[e] = e < E ? expf(w[e] - mx) : 0.f; sum += w[e]; }
    const float isum = 1.f / sum;
    float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int e = 0; e < ME; ++e) { w[e] *= isum; y.x = fmaf(w[e], o[e].x, y.x); y.y = fmaf(w[e], o[e].y, y.y); y.z = fmaf(w[e], o[e].z, y.z); y.w = fmaf(w[e], o[e].w, y.w); }
    const float inv = 1.f / fmaxf(sqrtf(cts_group_sum(y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w, LP)), CTS_EPS);
    const float4 sh_ = make_float4(y.x * inv, y.y * inv, y.z * inv, y.w * inv);
    const float4 t = *reinterpret_cast<const float4*>(that + (size_t)rc * L + c);
    const float4 d = make_float4(t.x - sh_.x, t.y - sh_.y, t.z - sh_.z, t.w - sh_.w);
    if (live) loss += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
    const float4 gs = make_float4(-scale * d.x, -scale * d.y, -scale * d.z, -scale * d.w);          // d latent loss / d shat
    const float dot = cts_group_sum(gs.x * sh_.x + gs.y * sh_.y + gs.z * sh_.z + gs.w * sh_.w, LP);
    const float4 dy = make_float4((gs.x - sh_.x * dot) * inv, (gs.y - sh_.y * dot) * inv, (gs.z - sh_.z * dot) * inv, (gs.w - sh_.w * dot) * inv);
    float dw[ME], wd = 0.f;
#pragma unroll
    for (int e = 0; e < ME; ++e) {
      if (e < E) {
        if (live) { const float4 g4 = make_float4(w[e] * dy.x, w[e] * dy.y, w[e] * dy.z, w[e] * dy.w);
          *reinterpret_cast<float4*>(douts + (size_t)r * sr + (size_t)e * se + c) = g4; db[e].x += g4.x; db[e].y += g4.y; db[e].z += g4.z; db[e].w += g4.w; }
        dw[e] = cts_group_sum(dy.x * o[e].x + dy.y * o[e].y + dy.z * o[e].z + dy.w * o[e].w, LP) + lbg[e];
        wd = fmaf(w[e], dw[e], wd);
      } else dw[e] = 0.f;
    }
    if (live && cq == 0) {
#pragma unroll
      for (int e = 0; e < ME; ++e) if (e < E) dlogits[(size_t)r * E + e] = w[e] * (dw[e] - wd);          // softmax backward
    }
  }
  if (dbias_part) {          // the workgroup's partial row of the output bias gradient: row lanes added in a fixed order, expert by expert
#pragma unroll
    for (int e = 0; e < ME; ++e) if (e < E) {
      shb[threadIdx.x] = db[e];
      __syncthreads();
      if (rl == 0) {
        float4 t = shb[cq];
        for (int j = 1; j < RL; ++j) { const float4 u = shb[j * LP + cq]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        *reinterpret_cast<float4*>(dbias_part + (size_t)blockIdx.x * E * L + (size_t)e * L + c) = t;
      }
      __syncthreads();
    }
  }
  shl[threadIdx.x] = loss;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int j = 0; j < 256; ++j) s += shl[j];
    float lb = 0.f;
    if (blockIdx.x == 0) { for (int e = 0; e < E; ++e) { const float u = usage_sum[e] * invn - invE; lb += u * u; } lb *= invE; }
    float* prow = part + (size_t)blockIdx.x * 4;
    prow[0] = s / ((float)n * (float)L); prow[1] = lb; prow[2] = prow[3] = 0.f;
  }
}
// The same mixture FORWARD only (the rollout's student rows, CTS.act: rsl_rl/rsl_rl/algorithms/cts.py:112-149 -> modules/utils.py:96-152): softmax gate, weighted sum of the
// expert outputs (+ their bias), L2 normaliser, and the result scattered to the env-ordered latent — in place of softmax, a broadcast product, two reductions, a clamp,
// a division and an index_copy (8 launches of ~5 us each on [n, 8] / [n, 32] tensors, 24 times per iteration).  rows: optional destination row of source row r.
template <int ME>
__global__ void __launch_bounds__(256) go2nn_moe_mix_forward_kernel(const float* __restrict__ logits, const float* __restrict__ outs, const float* __restrict__ bias,
                                                                    const int* __restrict__ rows, float* __restrict__ z, int ldz, int n, int E, int L, long long sr, long long se) {
  const int LP = L >> 2, RL = 256 / LP, cq = threadIdx.x % LP, rl = threadIdx.x / LP, c = cq * 4;
  const int r0 = blockIdx.x * CTS_ROWS_PER_WG, r1 = min(n, r0 + CTS_ROWS_PER_WG);
  float4 bq[ME];
#pragma unroll
  for (int e = 0; e < ME; ++e) bq[e] = (bias && e < E) ? *reinterpret_cast<const float4*>(bias + (size_t)e * L + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int rb = r0; rb < r1; rb += RL) {
    const int r = rb + rl, rc = min(r, r1 - 1);
    float w[ME]; float4 o[ME];
    float mx = -3.4e38f, sum = 0.f;
#pragma unroll
    for (int e = 0; e < ME; ++e) {
      w[e] = e < E ? logits[(size_t)rc * E + e] : -3.4e38f; mx = fmaxf(mx, w[e]);
      o[e] = e < E ? *reinterpret_cast<const float4*>(outs + (size_t)rc * sr + (size_t)e * se + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      o[e].x += bq[e].x; o[e].y += bq[e].y; o[e].z += bq[e].z; o[e].w += bq[e].w;
    }
#pragma unroll
    for (int e = 0; e < ME; ++e) { w[e] = e < E ? expf(w[e] - mx) : 0.f; sum += w[e]; }
    const float isum = 1.f / sum;
    float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int e = 0; e < ME; ++e) { w[e] *= isum; y.x = fmaf(w[e], o[e].x, y.x); y.y = fmaf(w[e], o[e].y, y.y); y.z = fmaf(w[e], o[e].z, y.z); y.w = fmaf(w[e], o[e].w, y.w); }
    const float inv = 1.f / fmaxf(sqrtf(cts_group_sum(y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w, LP)), CTS_EPS);
    if (r < r1) *reinterpret_cast<float4*>(z + (size_t)(rows ? rows[r] : r) * ldz + c) = make_float4(y.x * inv, y.y * inv, y.z * inv, y.w * inv);
  }
}
#endif  // !GO2_EMU
