// go2nn_mlp3.h — the rollout's policy evaluation (go2nn_mlp_kernel's job: PPO.act, rsl_rl/rsl_rl/algorithms/ppo.py:90-102; CTS: actor_critic_cts.py:146-176) on the
// bf16 matrix pipe with fp32 operands, the learner's arithmetic (go2nn_bx3.h): every weight and every activation is split EXACTLY into three bf16 planes, a product is
// six v_mfma_f32_32x32x16_bf16 terms accumulated in fp32 — 192 matrix-pipe cycles per 16 inputs of a 32 x 32 tile against 512 with v_mfma_f32_32x32x2_f32.
//   * weights: go2nn_pack leaves a plane image behind the fp32 one: [tile of 32 outputs][k-block of 16 inputs][plane][lane][8 bf16] — one 1-KiB wave load per plane,
//     lane (j = l & 31, g = l >> 5) -> W[32 t + j][16 kb + 8 g .. + 7], the B fragment of the MFMA as it stands; zero k-blocks pad a layer to the ring's depth
//   * activations live in LDS AS PLANES ([plane][32 rows][pitch] bf16, pitch = 16 bytes off a multiple of 64): the wave that produces an output splits it ONCE in its
//     epilogue (bias + ELU in fp32 first), and the k-loop's A fragments are three plain ds_read_b128 per k-block shared by the wave's tiles — no VALU work in the loop.
//     6 bytes per activation instead of 4: the two regions (even / odd activations) are sized per network by describe(); what does not fit 160 KB (two 512-wide
//     neighbours) stays on the fp32-MFMA kernel
//   * the last layer's tile is written as fp32: the heads (nn_finish) read it as before
// One workgroup = 32 rows through every layer of one network, 8 waves, a wave owns the output tiles wave, wave + 8 (as go2nn_mlp_kernel).
#pragma once
#ifndef GO2_EMU

// both images of one layer in one launch: the fp32 operand order of go2nn_mlp_kernel (+ the padded bias), then the planes
__global__ void __launch_bounds__(256) go2nn_pack_kernel(const float* __restrict__ W, const float* __restrict__ b, float* __restrict__ out_w, float* __restrict__ out_b, u32x4* __restrict__ out3,
                                                         int K, int N, int KB, int KB3, int NT) {
  const int64_t nw = (int64_t)NT * KB * 256;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < nw + NT * 32; idx += (int64_t)gridDim.x * 256) {
    if (idx < nw) {
      const int e = (int)(idx & 3), lane = (int)((idx >> 2) & 63); const int64_t blk = idx >> 8;
      const int kb = (int)(blk % KB), t = (int)(blk / KB);
      const int n = 32 * t + (lane & 31), k = 8 * kb + 4 * (lane >> 5) + e;
      out_w[idx] = (n < N && k < K) ? W[(int64_t)n * K + k] : 0.f;
    } else {
      const int n = (int)(idx - nw);
      out_b[n] = n < N ? b[n] : 0.f;
    }
  }
  const int64_t total = (int64_t)NT * KB3 * 64;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int lane = (int)(idx & 63); const int64_t blk = idx >> 6;
    const int kb = (int)(blk % KB3), t = (int)(blk / KB3);
    const int n = 32 * t + (lane & 31), k0 = 16 * kb + 8 * (lane >> 5);
    f32x4 v[2];
#pragma unroll
    for (int e = 0; e < 8; ++e) { const int k = k0 + e; v[e >> 2][e & 3] = (n < N && k < K) ? W[(int64_t)n * K + k] : 0.f; }
    u32x2 h0, m0, l0, h1, m1, l1;
    bx3_split4(v[0], h0, m0, l0); bx3_split4(v[1], h1, m1, l1);
    u32x4* o = out3 + blk * 192 + lane;
    o[0] = u32x4{h0[0], h0[1], h1[0], h1[1]}; o[64] = u32x4{m0[0], m0[1], m1[0], m1[1]}; o[128] = u32x4{l0[0], l0[1], l1[0], l1[1]};
  }
}

// two fp32 -> the three planes' packed pairs (low half = a, high half = b)
__device__ __forceinline__ void nn3_split2(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
  h = bx3_pk(a, b);
  const float r0 = a - bx3_lo(h), r1 = b - bx3_hi(h);
  m = bx3_pk(r0, r1);
  l = bx3_pk(r0 - bx3_lo(m), r1 - bx3_hi(m));
}

// One layer for one wave: NTW output tiles, the weights' planes prefetched D k-blocks ahead through a register ring of 2 D slots (KB is a multiple of 2 D: the
// k-blocks past the KV that hold inputs are zero blocks, loaded by the unrolled body but not multiplied).
template <int NTW>
__device__ __forceinline__ void run_layer3(const unsigned char* __restrict__ A, const int apitch, const int KV, unsigned char* __restrict__ B, const int bpitch,
                                           const u32x4* __restrict__ Wp, const float* __restrict__ bp, const int KB, const int NT, const int first, const int lane, const bool last, long long* tacc = nullptr) {
#ifdef GO2NN_STAMPS
  const long long t_in = wall_clock64();
#endif
  constexpr int D = NTW == 1 ? 4 : 2, R = 2 * D;          // prefetch distance in k-blocks; ring slots (block kb sits in slot kb % R)
  const int i = lane & 31, g = lane >> 5;
  f32x16 acc[NTW];
  float bias[NTW];          // (requested in front of the k-loop: at the head of the epilogue its latency — and that of every prefetch still in flight, vmcnt counts in order — was exposed)
#pragma unroll
  for (int j = 0; j < NTW; ++j) bias[j] = bp[32 * (first + NN_WAVES * j < NT ? first + NN_WAVES * j : NT - 1) + i];
  const u32x4* wt[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int t = first + NN_WAVES * j < NT ? first + NN_WAVES * j : NT - 1;      // (a tile past the layer's last is a clamped duplicate whose result is dropped)
    wt[j] = Wp + (int64_t)t * KB * 192 + lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  }
  // The refill of a slot is D blocks behind its consumption and goes to ANOTHER slot than the one being consumed (a ring of D slots refilled in place makes the
  // compiler copy every loaded block — the copies wait for the newest load, vmcnt(0), once per trip: the prefetch is gone)
  u32x4 ring[R][NTW][3];
#pragma unroll
  for (int s = 0; s < D; ++s) {
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p) ring[s][j][p] = wt[j][(s * 3 + p) * 64];
    __builtin_amdgcn_sched_barrier(0);          // (in block order: vmcnt counts in order, and the loop's waits are the stricter of the entry's and the back edge's)
  }
  const unsigned char* arow = A + i * apitch + g * 16;
  const int aplane = NN_ROWS * apitch;
  u32x4 an[3];
#pragma unroll
  for (int p = 0; p < 3; ++p) an[p] = *reinterpret_cast<const u32x4*>(arow + p * aplane);
  // one trip = R blocks; the LAST trip refills nothing past the layer's blocks (a clamped re-load would still be in flight at the epilogue, whose first register
  // reuse then waits for it: the full load latency once per layer)
  auto trip = [&](const int kb0, auto last_c) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last_c)::value;
#pragma unroll
    for (int s = 0; s < R; ++s) {
      const int kb = kb0 + s;
      if (!LAST || s < D) {
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
          for (int p = 0; p < 3; ++p) ring[(s + D) % R][j][p] = wt[j][((kb + D) * 3 + p) * 64];
      }
      u32x4 a[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) a[p] = an[p];
      const int ka = kb + 1 < KV ? kb + 1 : KV - 1;
#pragma unroll
      for (int p = 0; p < 3; ++p) an[p] = *reinterpret_cast<const u32x4*>(arow + p * aplane + ka * 32);
      if (kb < KV) {          // (uniform; the zero blocks that pad the layer to the ring are loaded, not multiplied)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
          for (int p = 0; p < 3; ++p) asm volatile("" : "+v"(ring[s][j][p]));          // (consumed as one 16-byte value: keeps the wave load whole)
        // six terms, small ones first (plane 0 hi, 1 mid, 2 lo); with two tiles consecutive MFMAs go to different accumulators
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
          for (int j = 0; j < NTW; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[PA[q]]), __builtin_bit_cast(bf16x8, ring[s][j][PB[q]]), acc[j], 0, 0, 0);
      }
    }
  };
  for (int kb0 = 0; kb0 + R < KB; kb0 += R) trip(kb0, G3Int<0>{});
  trip(KB - R, G3Int<1>{});
#ifdef GO2NN_STAMPS
  asm volatile("s_nop 0" :: "v"(acc[0][0]));          // (the accumulators are complete before the clock is read)
  const long long t_loop = wall_clock64();
#endif
  // epilogue: C/D layout of the 32x32 tile: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5); registers 2 q, 2 q + 1 are neighbouring rows
  const int bplane = NN_ROWS * bpitch;
#pragma unroll
  for (int j = 0; j < NTW; ++j) if (first + NN_WAVES * j < NT) {
    const int n = 32 * (first + NN_WAVES * j) + i;
    if (last) {
      float* Bf = reinterpret_cast<float*>(B);
#pragma unroll
      for (int r = 0; r < 16; ++r) Bf[((r & 3) + 8 * (r >> 2) + 4 * g) * bpitch + n] = acc[j][r] + bias[j];          // (bpitch in floats here)
    } else {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * g;
        unsigned h, m, l; nn3_split2(elu1(acc[j][r] + bias[j]), elu1(acc[j][r + 1] + bias[j]), h, m, l);
        unsigned char* d = B + row * bpitch + n * 2;
        *reinterpret_cast<unsigned short*>(d) = (unsigned short)h;                  *reinterpret_cast<unsigned short*>(d + bpitch) = (unsigned short)(h >> 16);
        *reinterpret_cast<unsigned short*>(d + bplane) = (unsigned short)m;         *reinterpret_cast<unsigned short*>(d + bplane + bpitch) = (unsigned short)(m >> 16);
        *reinterpret_cast<unsigned short*>(d + 2 * bplane) = (unsigned short)l;     *reinterpret_cast<unsigned short*>(d + 2 * bplane + bpitch) = (unsigned short)(l >> 16);
      }
    }
  }
#ifdef GO2NN_STAMPS
  if (tacc) { __builtin_amdgcn_s_waitcnt(0); const long long t_out = wall_clock64(); tacc[0] = t_loop - t_in; tacc[1] = t_out - t_loop; }
#endif
}

__global__ void __launch_bounds__(NN_THREADS) go2nn_mlp3_kernel(const NNArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[NN3_LDS];
  const NetDesc& nd = a.net[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63, row0 = blockIdx.x * NN_ROWS;
  const int wave = __builtin_amdgcn_readfirstlane(((tid >> 6) + (int)blockIdx.x) & (NN_WAVES - 1));          // (tile ownership rotated by the workgroup index: go2nn_mlp_kernel)
  unsigned char* A = lds; unsigned char* B = lds + nd.lds3;          // even / odd activations
#ifdef GO2NN_STAMPS
  long long* dbg = (a.mode == 1 && a.y) ? (long long*)a.y + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 : nullptr;      // tools/policy_bench.py --stamps
#endif
  NN_STAMP(0);
#ifdef GO2NN_STAMPS
  long long tacc[2] = {0, 0}, tl[2] = {0, 0};
#define NN3_TACC , tacc
#else
#define NN3_TACC
#endif
  if (row0 >= nd.nrows) return;
  int apitch = nn3_pitch(16 * nd.KV3[0]);
  {   // stage the workgroup's input rows as planes, zero-padded to the first layer's k-blocks: a wave takes its share of the rows, a lane two neighbouring columns;
      // every load of the wave is requested before the first value is split (4 rows x up to 4 column pairs per lane)
    const int Kp = nd.KV3[0] * 16, K0 = nd.in_dim, kx = nd.kx, w = tid >> 6, aplane = NN_ROWS * apitch;
    constexpr int RW = NN_ROWS / NN_WAVES, NQ = GO2NN_MAX_WIDTH / 128;
    float v[RW][NQ][2];
    int srow[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) { const int e = min(row0 + w * RW + r, nd.nrows - 1); srow[r] = nd.rows ? nd.rows[e] : e; }          // (the row indices first: one wait, not one per row)
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int i = w * RW + r; const bool live = row0 + i < nd.nrows;
      const int64_t sr = srow[r];
      const float* __restrict__ src = nd.x + sr * nd.ldx;
      const float* __restrict__ src2 = nd.x2 ? nd.x2 + sr * nd.ldx2 - kx : src;          // (indexed with k >= kx only)
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int k = 2 * lane + 128 * q + c, kc = k < K0 ? k : K0 - 1;
          const float* __restrict__ ptr = kc < kx ? src + kc : src2 + kc;          // (a select of addresses, ONE load: `c ? src[k] : src2[k]` compiles to two guarded loads, each waited for)
          const float x = 128 * q < Kp ? *ptr : 0.f;
          v[r][q][c] = (live && k < K0) ? x : 0.f;
        }
    }
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int k = 2 * lane + 128 * q;
        if (k < Kp) {
          unsigned h, m, l; nn3_split2(v[r][q][0], v[r][q][1], h, m, l);
          unsigned char* d = A + (w * RW + r) * apitch + k * 2;
          *reinterpret_cast<unsigned*>(d) = h; *reinterpret_cast<unsigned*>(d + aplane) = m; *reinterpret_cast<unsigned*>(d + 2 * aplane) = l;
        }
      }
  }
  __syncthreads();
  NN_STAMP(1);
  int ldo = 0;
  for (int l = 0; l < nd.nl; ++l) {
    const int KB = nd.KB3[l], NT = nd.NT[l];
    const u32x4* __restrict__ Wp = reinterpret_cast<const u32x4*>(nd.packed + nd.woff3[l]);
    const float* __restrict__ bp = nd.packed + nd.boff[l];
    const bool last = l == nd.nl - 1;
    const int bpitch = last ? 32 * NT + 4 : nn3_pitch(32 * NT);          // (floats for the last layer's fp32 tile, bytes otherwise)
    if (wave < NT) {
      if (NT > NN_WAVES) run_layer3<2>(A, apitch, nd.KV3[l], B, bpitch, Wp, bp, KB, NT, wave, lane, last NN3_TACC);
      else run_layer3<1>(A, apitch, nd.KV3[l], B, bpitch, Wp, bp, KB, NT, wave, lane, last NN3_TACC);
    }
#ifdef GO2NN_STAMPS
    if (l < 4) { tl[l >> 1] |= ((tacc[0] & 0xffff) | ((tacc[1] & 0xffff) << 16)) << (32 * (l & 1)); tacc[0] = tacc[1] = 0; }
#endif
    __syncthreads();
    NN_STAMP(2 + l);
    unsigned char* t_ = A; A = B; B = t_; apitch = bpitch; ldo = bpitch;
  }
#ifdef GO2NN_STAMPS
  if (dbg && tid == 0) { dbg[6] = tl[0]; dbg[7] = tl[1]; }
#endif
  nn_finish(a, nd, reinterpret_cast<float*>(A), ldo, row0, tid);
}
#endif  // !GO2_EMU
