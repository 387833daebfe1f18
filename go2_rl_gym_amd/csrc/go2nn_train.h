// go2nn_train.h — the learner-side kernels of include/go2nn.h (included by go2nn_impl.cpp; one translation unit, one library).
//
// PPO.update's backward pass (rsl_rl/rsl_rl/algorithms/ppo.py:120-187 -> autograd over the two MLPs of modules/actor_critic.py:50-75) starts
// at the networks' NARROW ends: the 12-wide action mean and the 1-wide value.  As vendor GEMMs those are the worst shapes of the whole update
// (profiles/r2_timeline_rollout_step_and_minibatch.txt: the value head's weight gradient, a [1,128] = [1,24576] x [24576,128] product, 58 us;
// the policy head's, 15 us + a split-K fix-up + a fill), followed by a column-sum launch for the bias gradient, an ELU-backward pass over the
// [24576,128] activations and its own column-sum pair — ~120 us at the head of the critic's chain, which is the critical path of a mini-batch.
// All of it is ONE streaming pass over the activations: go2nn_head_backward.
#pragma once

#define HB_THREADS 256
#define HB_MAX_C 16

#ifndef GO2_EMU
// One pass over rows [r0, r1) of y [B,K] (the head's input = the last hidden layer's ELU output) and gy [B,C] (gradient of the head's output):
//   gx = gy W                       (the head's input gradient; W [C,K])
//   gz = gx * (y > 0 ? 1 : y + 1)   (through the ELU: gradient of the last hidden layer's pre-activation)   -> stored
//   partial sums over the rows: dW[c][k] += gy[c] y[k]; gb[k] += gz[k]; db[c] += gy[c]                       -> part[wg][(C+1) K + C]
// Thread = (column quad cq, row lane rl): QP = pow2 >= K/4 quads, 256 / QP row lanes; a thread keeps its quad of W (C float4), of dW (C float4)
// and of gb in registers.  HBM-bound (reads y, writes gz: 8 B per element + the narrow gy): 4 rows of loads in flight per thread.
// The row lanes' accumulators are combined through LDS one set at a time, in a fixed order (deterministic; no atomics).
template <int CP>
__global__ void __launch_bounds__(HB_THREADS) go2nn_head_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y, const float* __restrict__ w,
                                                                    float* __restrict__ gz, float* __restrict__ part, int B, int C, int K, int qp_log2, int rows_per_wg) {
  __shared__ float4 sh[HB_THREADS];
  __shared__ float shd[HB_THREADS / 16][HB_MAX_C];
  const int QP = 1 << qp_log2, RL = HB_THREADS >> qp_log2;
  const int cq = threadIdx.x & (QP - 1), rl = threadIdx.x >> qp_log2;
  const bool on = 4 * cq < K;
  const int k0 = on ? 4 * cq : 0;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 wq[CP], dw[CP], gb = zero;
  float db[CP];
#pragma unroll
  for (int c = 0; c < CP; ++c) {
    const float4 t = *reinterpret_cast<const float4*>(w + (size_t)min(c, C - 1) * K + k0);      // (clamped address + select: a conditional load becomes a flat load through a scratch zero)
    wq[c] = (on && c < C) ? t : zero;
    dw[c] = zero; db[c] = 0.f;
  }
  const int r0 = blockIdx.x * rows_per_wg, r1 = min(B, r0 + rows_per_wg);
  constexpr int U = CP > 4 ? 2 : 4;                    // rows in flight per thread (registers: the wide heads hold 2 x CP float4 of W / dW)
  for (int rb = r0 + rl; rb < r1; rb += U * RL) {
    float4 v[U]; float g[U][CP];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = min(rb + u * RL, r1 - 1);           // clamped: the loads of a ragged tail stay in range, its results are dropped below
      v[u] = *reinterpret_cast<const float4*>(y + (size_t)r * K + k0);
#pragma unroll
      for (int c = 0; c < CP; ++c) { const float t = gy[(size_t)r * C + min(c, C - 1)]; g[u][c] = c < C ? t : 0.f; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = rb + u * RL;
      if (r < r1) {
        float4 s = zero;
#pragma unroll
        for (int c = 0; c < CP; ++c) {
          s.x = fmaf(g[u][c], wq[c].x, s.x); s.y = fmaf(g[u][c], wq[c].y, s.y); s.z = fmaf(g[u][c], wq[c].z, s.z); s.w = fmaf(g[u][c], wq[c].w, s.w);
          dw[c].x = fmaf(g[u][c], v[u].x, dw[c].x); dw[c].y = fmaf(g[u][c], v[u].y, dw[c].y); dw[c].z = fmaf(g[u][c], v[u].z, dw[c].z); dw[c].w = fmaf(g[u][c], v[u].w, dw[c].w);
          db[c] += g[u][c];
        }
        float4 o;
        o.x = s.x * (v[u].x > 0.f ? 1.f : v[u].x + 1.f); o.y = s.y * (v[u].y > 0.f ? 1.f : v[u].y + 1.f);
        o.z = s.z * (v[u].z > 0.f ? 1.f : v[u].z + 1.f); o.w = s.w * (v[u].w > 0.f ? 1.f : v[u].w + 1.f);
        if (on) *reinterpret_cast<float4*>(gz + (size_t)r * K + k0) = o;
        gb.x += o.x; gb.y += o.y; gb.z += o.z; gb.w += o.w;
      }
    }
  }
  float* prow = part + (size_t)blockIdx.x * ((size_t)(C + 1) * K + C);
#pragma unroll
  for (int c = 0; c <= CP; ++c) {
    if (c < C || c == CP) {                            // c == CP: the gb set, stored behind the C rows of dW
      sh[threadIdx.x] = c == CP ? gb : dw[c < CP ? c : 0];
      __syncthreads();
      if (rl == 0 && on) {
        float4 s = sh[cq];
        for (int j = 1; j < RL; ++j) { const float4 t = sh[j * QP + cq]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
        *reinterpret_cast<float4*>(prow + (size_t)(c == CP ? C : c) * K + k0) = s;
      }
      __syncthreads();
    }
  }
  if (cq == 0) {
#pragma unroll
    for (int c = 0; c < CP; ++c) shd[rl][c] = db[c];
  }
  __syncthreads();
  if ((int)threadIdx.x < C) {
    float s = shd[0][threadIdx.x];
    for (int j = 1; j < RL; ++j) s += shd[j][threadIdx.x];
    prow[(size_t)(C + 1) * K + threadIdx.x] = s;
  }
}

// Row sums out[c] = sum_r part[r][c] (r ascending within a thread, then a fixed tree) for a LIST of jobs in one launch: the second stage of every
// deterministic reduction of a backward pass — per-workgroup column partials of the head and of the input gradients, the row splits of the weight
// gradients.  Two block shapes: "tall" (many rows, few columns: 16 columns x 16 row groups, LDS tree) and "wide" (<= 32 rows: one thread per
// column, 256 columns per block).
#define SR_MAX_JOBS 16
struct SumRowsArgs {
  const float* part[SR_MAX_JOBS]; float* out[SR_MAX_JOBS];
  int nrows[SR_MAX_JOBS], ncols[SR_MAX_JOBS], first_block[SR_MAX_JOBS + 1];      // blocks [first_block[j], first_block[j+1]) belong to job j
  int njobs;
};
__global__ void __launch_bounds__(256) go2nn_sum_rows_kernel(const SumRowsArgs a) {
  __shared__ float sh[16][17];
  int j = 0;
  while (j + 1 < a.njobs && (int)blockIdx.x >= a.first_block[j + 1]) ++j;
  const float* __restrict__ part = a.part[j]; float* __restrict__ out = a.out[j];
  const int nrows = a.nrows[j], ncols = a.ncols[j], blk = blockIdx.x - a.first_block[j];
  if (nrows <= 32) {
    const int c = blk * 256 + threadIdx.x;
    if (c < ncols) {
      float s = 0.f;
      for (int r0 = 0; r0 < nrows; r0 += 8) {       // eight rows' loads in flight, added in ascending order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(size_t)min(r0 + u, nrows - 1) * ncols + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (r0 + u < nrows) s += v[u];
      }
      out[c] = s;
    }
    return;
  }
  const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4, c = blk * 16 + cl;
  float s = 0.f;
  if (c < ncols) for (int r0 = rg; r0 < nrows; r0 += 64) {          // four of the thread's rows in flight
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = part[(size_t)min(r0 + 16 * u, nrows - 1) * ncols + c];
#pragma unroll
    for (int u = 0; u < 4; ++u) if (r0 + 16 * u < nrows) s += v[u];
  }
  sh[rg][cl] = s;
  __syncthreads();
  if (rg == 0 && c < ncols) {
    float t[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = sh[i][cl];
#pragma unroll
    for (int wd = 8; wd >= 1; wd >>= 1)
#pragma unroll
      for (int i = 0; i < wd; ++i) t[i] += t[i + wd];
    out[c] = t[0];
  }
}
#endif  // !GO2_EMU

// rows per workgroup / number of workgroups of go2nn_head_backward for a batch of B rows (shared by the launch and the workspace size)
static inline void head_bwd_shape(int B, int K, int* qp_log2, int* rows_per_wg, int* nwg) {
  int q = 4;                                           // at least 16 quads
  while ((1 << q) * 4 < K) ++q;
  const int RL = HB_THREADS >> q, step = 4 * RL;      // (a multiple of U RL for both unroll depths)
  int rows = (B + 255) / 256;                          // ~256 workgroups: one per CU
  rows = (rows + step - 1) / step * step;
  *qp_log2 = q; *rows_per_wg = rows; *nwg = (B + rows - 1) / rows;
}
