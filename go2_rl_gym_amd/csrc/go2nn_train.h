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
  __shared__ float4 sh[CP + 1][HB_THREADS];
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
  // (every set through LDS at once, one barrier — go2nn_ppo_heads_kernel; the row lanes are added in the same fixed order: bit-identical partials)
#pragma unroll
  for (int c = 0; c <= CP; ++c) sh[c][threadIdx.x] = c == CP ? gb : dw[c < CP ? c : 0];          // c == CP: the gb set, stored behind the C rows of dW
  __syncthreads();
  for (int task = threadIdx.x; task < ((CP + 1) << qp_log2); task += HB_THREADS) {
    const int c = task >> qp_log2, q = task & (QP - 1);
    if ((c < C || c == CP) && 4 * q < K) {
      float4 s = sh[c][q];
      for (int j = 1; j < RL; ++j) { const float4 t = sh[c][j * QP + q]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
      *reinterpret_cast<float4*>(prow + (size_t)(c == CP ? C : c) * K + 4 * q) = s;
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
#define SR_MAX_JOBS GO2NN_MAX_SUM_JOBS
struct SumRowsArgs {
  const float* part[SR_MAX_JOBS]; float* out[SR_MAX_JOBS]; float* acc[SR_MAX_JOBS];
  int nrows[SR_MAX_JOBS], ncols[SR_MAX_JOBS], nacc[SR_MAX_JOBS], first_block[SR_MAX_JOBS + 1];      // blocks [first_block[j], first_block[j+1]) belong to job j
  int out_w[SR_MAX_JOBS], out_ld[SR_MAX_JOBS];          // out_w > 0: sum c goes to out[(c / out_w) * out_ld + c % out_w]
  int njobs;
};
__global__ void __launch_bounds__(256) go2nn_sum_rows_kernel(const SumRowsArgs a) {
  __shared__ float sh[8][33];
  int j = 0;
  while (j + 1 < a.njobs && (int)blockIdx.x >= a.first_block[j + 1]) ++j;
  const float* __restrict__ part = a.part[j]; float* __restrict__ out = a.out[j];
  const int nrows = a.nrows[j], ncols = a.ncols[j], blk = blockIdx.x - a.first_block[j];
  const int ow = a.out_w[j], old_ = a.out_ld[j];
  auto oidx = [&](int c) { return ow > 0 ? (size_t)(c / ow) * old_ + (c % ow) : (size_t)c; };
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
      out[oidx(c)] = s;
      if (a.acc[j] && c < a.nacc[j]) a.acc[j][c] += s;          // (one thread per column: no race)
    }
    return;
  }
  // tall shape: 32 columns x 8 row groups per block — a row group's load is one 128-byte line per row (round 6: 16 columns = half a line per row left the
  // partial matrices of the weight gradients, 20 - 80 MB per mini-batch, at half the HBM efficiency)
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5, c = blk * 32 + cl;
  float s = 0.f;
  if (c < ncols) for (int r0 = rg; r0 < nrows; r0 += 64) {          // eight of the thread's rows in flight
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(size_t)min(r0 + 8 * u, nrows - 1) * ncols + c];
#pragma unroll
    for (int u = 0; u < 8; ++u) if (r0 + 8 * u < nrows) s += v[u];
  }
  sh[rg][cl] = s;
  __syncthreads();
  if (rg == 0 && c < ncols) {
    float t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = sh[i][cl];
#pragma unroll
    for (int wd = 4; wd >= 1; wd >>= 1)
#pragma unroll
      for (int i = 0; i < wd; ++i) t[i] += t[i + wd];
    out[oidx(c)] = t[0];
    if (a.acc[j] && c < a.nacc[j]) a.acc[j][c] += t[0];
  }
}
#endif  // !GO2_EMU


// ---- go2nn_ppo_heads: the narrow heads of BOTH networks forward, the PPO loss head with its analytic gradients, and the heads backward, in ONE pass ----
// PPO.update (rsl_rl/rsl_rl/algorithms/ppo.py:131-170) ends its forward pass with two degenerate GEMMs (128 -> 12 action means, 128 -> 1 value), evaluates
// ~150 element-wise operators on [B, 12] / [B] tensors (log-prob, ratio, clipped surrogate, clipped value loss, entropy, KL: go2sim_ppo_loss already folds
// those into one kernel), and starts its backward pass at the same narrow ends (go2nn_head_backward above).  Everything in that chain is per ROW of the
// mini-batch and reads or writes the last hidden activations y [B, K] of the two networks: here it is one streaming pass — y_a, y_c in, the pre-activation
// gradients gz_a, gz_c out (8 B per element each way), the rest per-workgroup partial sums — instead of 3 + 3 + 2 launches.
// Thread = (k quad cq, row lane rl) as in go2nn_head_bwd_kernel; the QP lanes of a row form the heads' dot products with an xor butterfly, then every lane of
// the row evaluates the loss terms of the row (A <= 16 values: cheaper than another exchange) and applies them to its own four columns.
#define PH_NSTAT 4              // surrogate, value loss, KL, entropy (means)
struct PpoHeadsArgs {
  const float *y_a, *y_c, *w_mu, *b_mu, *w_v, *b_v, *std_, *actions, *old_mu, *old_sigma, *old_logp, *adv, *old_values, *returns;
  float *gz_a, *gz_c, *part;
  int B, A, K, use_clip_v; float clip, vcoef, ecoef;
  int split; float w_head, w_tail;          // the surrogate's row weights: rows [0, split) w_head, the rest w_tail (plain PPO: split = B, both 1 / B; CTS: 1 / teacher rows, 1 / student rows)
};
#ifdef GO2_EMU
#define PH_HD
#else
#define PH_HD __host__ __device__
#endif
PH_HD static inline int ppo_heads_cols(int A, int K) { return PH_NSTAT + A + (A + 1) * K + A + 2 * K + 1; }
#ifndef GO2_EMU
// what a lane loads for one row: its four columns of both activations, the row's scalars, and the action-dimension values of the lane's ROLE (see below)
struct PhRow { float4 ya, yc; float act, omu, osg, olp, ad, tv, rt; };
template <int CP>
__global__ void __launch_bounds__(256, CP > 12 ? 1 : 2) go2nn_ppo_heads_kernel(const PpoHeadsArgs a, int qp_log2, int rows_per_wg) {          // (13 - 16 actions: 256 registers spilled 36 B per lane)
  static_assert(CP <= 16, "16 value slots in the reduce-scatter");
  __shared__ float4 sh[CP + 3][256];          // (60 KB at 12 actions: two workgroups per CU still fit)
  __shared__ float shs[256 / 16][PH_NSTAT + 2 * HB_MAX_C + 1];
  const int B = a.B, A = a.A, K = a.K;
  const int QP = 1 << qp_log2, RL = 256 >> qp_log2, SH = qp_log2 - 4;          // QP >= 16 lanes per row; lane role c = cq >> SH (2^SH lanes share a role)
  const int cq = threadIdx.x & (QP - 1), rl = threadIdx.x >> qp_log2;
  const bool on = 4 * cq < K;
  const int k0 = on ? 4 * cq : 0;
  const int role = cq >> SH; const bool first = (cq & ((1 << SH) - 1)) == 0, role_on = role < A;
  const int rc = role_on ? role : 0;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  const float LOG2PI = 1.8378770664093453f;
  float4 wq[CP], dw[CP];
#pragma unroll
  for (int c = 0; c < CP; ++c) {
    const float4 t = *reinterpret_cast<const float4*>(a.w_mu + (size_t)min(c, A - 1) * K + k0);
    wq[c] = (on && c < A) ? t : zero; dw[c] = zero;
  }
  // the role's action dimension: sigma, log sigma, 1 / sigma^2, head bias
  const float sg = a.std_[rc], ls = logf(sg), isg2 = 1.f / (sg * sg), bm = a.b_mu[rc];
  const float4 wv = on ? *reinterpret_cast<const float4*>(a.w_v + k0) : zero;
  const float bv = a.b_v[0], invB = 1.f / (float)B, lo = 1.f - a.clip, hi = 1.f + a.clip;
  float4 dwv = zero, gba = zero, gbc = zero;
  float dbv = 0.f, s_sur = 0.f, s_vl = 0.f, s_kl = 0.f, gs = 0.f, dbm = 0.f;
  const int r0 = blockIdx.x * rows_per_wg, r1 = min(B, r0 + rows_per_wg);
  auto fetch = [&](int rb) {
    PhRow x; const int r = min(rb + rl, r1 - 1);
    x.ya = *reinterpret_cast<const float4*>(a.y_a + (size_t)r * K + k0); x.yc = *reinterpret_cast<const float4*>(a.y_c + (size_t)r * K + k0);
    const size_t k = (size_t)r * A + rc;
    x.act = a.actions[k]; x.omu = a.old_mu[k]; x.osg = a.old_sigma[k]; x.olp = a.old_logp[r]; x.ad = a.adv[r]; x.tv = a.old_values[r]; x.rt = a.returns[r];
    return x;
  };
  PhRow nx = fetch(r0);
  for (int rb = r0; rb < r1; rb += RL) {          // (uniform trip count: the exchanges need every lane of a row; rows past the end run on a clamped row and are dropped)
    const PhRow x = nx;
    nx = fetch(min(rb + RL, r1 - 1));                // the next rows' loads fly while this row is worked on (one or two waves per SIMD: nothing else hides them)
    const bool live = rb + rl < r1;
    const float4 ya = x.ya, yc = x.yc;
    // heads forward.  Reduce-scatter of the CP partial dot products over the row's lanes: at each of four levels a lane keeps the half of its values that its
    // lane bit selects and adds the partner's — 8 + 4 + 2 + 1 exchanges instead of 16 x log2(QP) — then the lanes of one role are added up
    float pm[16], pv = yc.x * wv.x + yc.y * wv.y + yc.z * wv.z + yc.w * wv.w;
#pragma unroll
    for (int c = 0; c < 16; ++c) pm[c] = c < CP ? ya.x * wq[c < CP ? c : 0].x + ya.y * wq[c < CP ? c : 0].y + ya.z * wq[c < CP ? c : 0].z + ya.w * wq[c < CP ? c : 0].w : 0.f;
    float mu_r;
    {
      const int d3 = QP >> 1, d2 = QP >> 2, d1 = QP >> 3, d0 = QP >> 4;
      float p8[8], p4[4], p2[2];
      { const bool up = cq & d3;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float keep = up ? pm[8 + i] : pm[i], send = up ? pm[i] : pm[8 + i]; p8[i] = keep + __shfl_xor(send, d3); } }
      { const bool up = cq & d2;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float keep = up ? p8[4 + i] : p8[i], send = up ? p8[i] : p8[4 + i]; p4[i] = keep + __shfl_xor(send, d2); } }
      { const bool up = cq & d1;
#pragma unroll
        for (int i = 0; i < 2; ++i) { const float keep = up ? p4[2 + i] : p4[i], send = up ? p4[i] : p4[2 + i]; p2[i] = keep + __shfl_xor(send, d1); } }
      { const bool up = cq & d0; const float keep = up ? p2[1] : p2[0], send = up ? p2[0] : p2[1]; mu_r = keep + __shfl_xor(send, d0); }
      for (int d = d0 >> 1; d >= 1; d >>= 1) mu_r += __shfl_xor(mu_r, d);
    }
    for (int d = 1; d < QP; d <<= 1) pv += __shfl_xor(pv, d);
    const float v = pv + bv;
    // the loss head of the row (go2sim_ppo_loss's arithmetic: ppo.py:134-170): the role lanes form their action dimension's terms, the row sums go round the row
    const float m = mu_r + bm, dd = x.act - m, dm = x.omu - m;
    float lp = (role_on && first) ? -dd * dd * (0.5f * isg2) - ls - 0.5f * LOG2PI : 0.f;
    float kl = (role_on && first) ? logf(sg / x.osg + 1e-5f) + (x.osg * x.osg + dm * dm) * (0.5f * isg2) - 0.5f : 0.f;
    for (int d = 1; d < QP; d <<= 1) { lp += __shfl_xor(lp, d); kl += __shfl_xor(kl, d); }
    const float ratio = expf(lp - x.olp), rcl = fminf(fmaxf(ratio, lo), hi), in = (ratio >= lo && ratio <= hi) ? 1.f : 0.f;
    const float s1 = -x.ad * ratio, s2 = -x.ad * rcl, sur = fmaxf(s1, s2);
    const float w = s1 > s2 ? 1.f : (s1 < s2 ? in : 0.5f + 0.5f * in);          // torch.max splits ties evenly; clamp passes gradient inside [lo, hi]
    const float wr = rb + rl < a.split ? a.w_head : a.w_tail;
    const float g_lp = live ? -x.ad * w * ratio * wr : 0.f;
    const float dv = v - x.tv;
    float vl, gv;
    if (a.use_clip_v) {
      const float dc = fminf(fmaxf(dv, -a.clip), a.clip), vin = (dv >= -a.clip && dv <= a.clip) ? 1.f : 0.f, vc = x.tv + dc;
      const float l1 = (v - x.rt) * (v - x.rt), l2 = (vc - x.rt) * (vc - x.rt); vl = fmaxf(l1, l2);
      const float g1 = 2.f * (v - x.rt), g2 = 2.f * (vc - x.rt) * vin; gv = l1 > l2 ? g1 : (l1 < l2 ? g2 : 0.5f * g1 + 0.5f * g2);
    } else { vl = (x.rt - v) * (x.rt - v); gv = 2.f * (v - x.rt); }
    const float gval = live ? a.vcoef * gv * invB : 0.f;
    if (live) { s_sur += sur * wr; s_vl += vl * invB; s_kl += kl * invB; }
    const float gm_r = role_on ? g_lp * dd * isg2 : 0.f;          // d loss / d mu[row][role]
    gs += role_on ? g_lp * (dd * dd * isg2 / sg - 1.f / sg) : 0.f;
    dbm += gm_r;
    // heads backward on this lane's four columns: the A gradients of the row come from their role lanes
    float4 gx = zero;
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      const float gm = __shfl(gm_r, c << SH, QP);
      gx.x = fmaf(gm, wq[c].x, gx.x); gx.y = fmaf(gm, wq[c].y, gx.y); gx.z = fmaf(gm, wq[c].z, gx.z); gx.w = fmaf(gm, wq[c].w, gx.w);
      dw[c].x = fmaf(gm, ya.x, dw[c].x); dw[c].y = fmaf(gm, ya.y, dw[c].y); dw[c].z = fmaf(gm, ya.z, dw[c].z); dw[c].w = fmaf(gm, ya.w, dw[c].w);
    }
    float4 oa, oc;
    oa.x = gx.x * (ya.x > 0.f ? 1.f : ya.x + 1.f); oa.y = gx.y * (ya.y > 0.f ? 1.f : ya.y + 1.f); oa.z = gx.z * (ya.z > 0.f ? 1.f : ya.z + 1.f); oa.w = gx.w * (ya.w > 0.f ? 1.f : ya.w + 1.f);
    oc.x = gval * wv.x * (yc.x > 0.f ? 1.f : yc.x + 1.f); oc.y = gval * wv.y * (yc.y > 0.f ? 1.f : yc.y + 1.f);
    oc.z = gval * wv.z * (yc.z > 0.f ? 1.f : yc.z + 1.f); oc.w = gval * wv.w * (yc.w > 0.f ? 1.f : yc.w + 1.f);
    if (live && on) { *reinterpret_cast<float4*>(a.gz_a + (size_t)(rb + rl) * K + k0) = oa; *reinterpret_cast<float4*>(a.gz_c + (size_t)(rb + rl) * K + k0) = oc; }
    gba.x += oa.x; gba.y += oa.y; gba.z += oa.z; gba.w += oa.w; gbc.x += oc.x; gbc.y += oc.y; gbc.z += oc.z; gbc.w += oc.w;
    dwv.x = fmaf(gval, yc.x, dwv.x); dwv.y = fmaf(gval, yc.y, dwv.y); dwv.z = fmaf(gval, yc.z, dwv.z); dwv.w = fmaf(gval, yc.w, dwv.w);
    dbv += gval;
  }
  // the workgroup's partial row: [stats | d loss / d std | dW_mu [A][K] | gb_a [K] | db_mu [A] | dW_v [K] | gb_c [K] | db_v]; row lanes added in a fixed order
  float* prow = a.part + (size_t)blockIdx.x * ppo_heads_cols(A, K);
  float* pa = prow + PH_NSTAT + A; float* pc = pa + (size_t)(A + 1) * K + A;
  // (all CP + 3 quantities through LDS at once, ONE barrier: one quantity per pass cost two barriers each, 30 in all — 6 of the kernel's 28 us at 24576 rows.  The row
  //  lanes of a column quad are added in the same fixed order as before: bit-identical partials.)
#pragma unroll
  for (int c = 0; c < CP + 3; ++c) sh[c][threadIdx.x] = c < CP ? dw[c < CP ? c : 0] : (c == CP ? gba : (c == CP + 1 ? dwv : gbc));
  __syncthreads();
  for (int task = threadIdx.x; task < ((CP + 3) << qp_log2); task += 256) {
    const int c = task >> qp_log2, q = task & (QP - 1);
    if ((c < A || c >= CP) && 4 * q < K) {
      float4 t4 = sh[c][q];
      for (int j = 1; j < RL; ++j) { const float4 t = sh[c][j * QP + q]; t4.x += t.x; t4.y += t.y; t4.z += t.z; t4.w += t.w; }
      float* o = c < CP ? pa + (size_t)c * K : (c == CP ? pa + (size_t)A * K : (c == CP + 1 ? pc : pc + K));
      *reinterpret_cast<float4*>(o + 4 * q) = t4;
    }
  }
  if (cq == 0) { shs[rl][0] = s_sur; shs[rl][1] = s_vl; shs[rl][2] = s_kl; shs[rl][3] = 0.f; shs[rl][PH_NSTAT + 2 * HB_MAX_C] = dbv; }
  if (first && role_on) { shs[rl][PH_NSTAT + role] = gs; shs[rl][PH_NSTAT + HB_MAX_C + role] = dbm; }
  __syncthreads();
  if ((int)threadIdx.x < PH_NSTAT + 2 * HB_MAX_C + 1) {
    const int t = threadIdx.x;
    const int c = t < PH_NSTAT ? 0 : (t < PH_NSTAT + HB_MAX_C ? t - PH_NSTAT : (t < PH_NSTAT + 2 * HB_MAX_C ? t - PH_NSTAT - HB_MAX_C : 0));
    if (c < A) {
      float t1 = shs[0][t];
      for (int j = 1; j < RL; ++j) t1 += shs[j][t];
      if (t < PH_NSTAT) {          // (the entropy of a state-independent std is one number: workgroup 0 writes it)
        float e = 0.f; if (t == 3 && blockIdx.x == 0) for (int q = 0; q < A; ++q) e += 0.5f + 0.5f * LOG2PI + logf(a.std_[q]);
        prow[t] = t == 3 ? e : t1;
      }
      else if (t < PH_NSTAT + HB_MAX_C) prow[PH_NSTAT + c] = t1 - (blockIdx.x == 0 ? a.ecoef / a.std_[c] : 0.f);
      else if (t < PH_NSTAT + 2 * HB_MAX_C) pa[(size_t)(A + 1) * K + c] = t1;
      else pc[2 * K] = t1;
    }
  }
}
#endif

// go2nn_ppo_heads: ~512 workgroups (two per CU: the kernel holds 226 registers, two waves per SIMD cover each other's row loads), whole row groups each
static inline void ppo_heads_shape(int B, int K, int* qp_log2, int* rows_per_wg, int* nwg) {
  int q = 4;
  while ((1 << q) * 4 < K) ++q;
  const int RL = 256 >> q;
  int rows = (B + 511) / 512;
  rows = (rows + RL - 1) / RL * RL;
  *qp_log2 = q; *rows_per_wg = rows; *nwg = (B + rows - 1) / rows;
}
// rows per workgroup / number of workgroups of go2nn_head_backward for a batch of B rows (shared by the launch and the workspace size)
static inline void head_bwd_shape(int B, int K, int* qp_log2, int* rows_per_wg, int* nwg) {
  int q = 4;                                           // at least 16 quads
  while ((1 << q) * 4 < K) ++q;
  const int RL = HB_THREADS >> q, step = 4 * RL;      // (a multiple of U RL for both unroll depths)
  int rows = (B + 255) / 256;                          // ~256 workgroups: one per CU
  rows = (rows + step - 1) / step * step;
  *qp_log2 = q; *rows_per_wg = rows; *nwg = (B + rows - 1) / rows;
}
