// go2nn_gemm.h — fp32-MFMA GEMMs with the learner's element-wise work in their epilogues (included by go2nn_impl.cpp).
//
// PPO.update (rsl_rl/rsl_rl/algorithms/ppo.py:120-187) is, per mini-batch of M = 24576 rows and per network, three hidden layers
//   y = elu(x W^T + b)                                            (modules/actor_critic.py:50-75)
// forward and backward.  With vendor GEMMs every layer is followed by element-wise passes over the [M, N] activations — ELU, ELU', the bias
// gradient's column sums and their second stage, the sum over the row splits of the weight gradient: a quarter of the update's kernel time
// (profiles/r3_bench_kernel_stats.csv) spent re-reading tensors a GEMM just wrote.  Here those passes are epilogues:
//   forward        Y  = elu(X W^T + b)                                             EPI_BIAS_ELU
//   input grad     Gp = (G W) * elu'(Yp), column partial sums of Gp               EPI_DELU_COLSUM   (the previous layer's pre-activation gradient + bias gradient)
//   weight grad    dW = G^T X  as row-split partials                              EPI_STORE         (summed in a fixed order by go2nn_sum_rows)
//
// One kernel template: C[M,N] = sum_k A(m,k) B(n,k), v_mfma_f32_32x32x2_f32, 256 threads = 4 waves as 2 x 2, a wave owns TM x TN 32x32 tiles
// (workgroup tile 64 TM x 64 TN), k-tiles of 32 or 16 double-buffered through LDS.  Each operand is either "k-contiguous" (rows of k: X, W in the forward
// pass) or "k-strided" (k runs over rows: W in the input gradient, G and X in the weight gradient); LDS keeps the operand's own orientation:
//   k-contiguous: tile [rows][32]      — fragment = one ds_read_b128 per lane at [row][8 kb + 4 g]: the lane's four values feed four MFMAs whose
//                 k index (the lane's g) then stands for input 8 kb + 4 g + e, the same permutation on both operands (as in go2nn_mlp_kernel);
//                 no padding, the 16-byte quads of a row are XOR-swizzled by the row index instead (GmStage)
//   k-strided:    tile [32][rows + 8]  — fragment = four ds_read_b32 at [8 kb + 4 g + e][row]; pitch = 8 mod 16: the two half-waves (g = 0 / 1,
//                 four k-rows apart) fall on disjoint halves of the 64 banks
// No vendor GEMM library.  fp32 in, fp32 accumulate.  Which of a layer's products run here and which on hipBLASLt is decided per shape by measurement
// (go2_rl_gym_amd/rsl_rl/modules/fused.py:_own, profiles/r3_mlp_kernel_choice.txt).
#pragma once

#define GM_BK 32
#define GM_THREADS 256
enum { EPI_STORE = 0, EPI_BIAS_ELU = 1, EPI_DELU_COLSUM = 2, EPI_BIAS = 3, EPI_DELU_WG = 4 };          // EPI_BIAS: a Linear without activation (the encoders' last layer; grouped kernels only)
// EPI_DELU_WG (go2nn_bx3_kernel only): EPI_DELU_COLSUM whose tile never leaves the chip — the weight gradient of the layer BELOW is formed from it in the epilogue

struct GemmArgs {
  const float *A, *B; float* C;
  const float* bias;          // EPI_BIAS_ELU: [N]
  const float* Y;             // EPI_DELU_COLSUM: the ELU output [M][ldc] whose derivative multiplies the product
  float* part;                // EPI_DELU_COLSUM: column partial sums [nbm][N]
  int M, N, K, lda, ldb, ldc;
  int kchunk;                 // contraction range per blockIdx.z (a multiple of 32); K when not split
  long long c_split_stride;   // C of split z starts at C + z * c_split_stride
  int nbm, nbn;
  int c_vec;                  // C (and Y) rows are a multiple of 4 floats long and 16-byte aligned: 16-byte epilogue accesses
  long long* stamps;          // tools only (-DGM_STAMPS)
  int debug;                  // tools only (GO2NN_GEMM_DEBUG): 1 = no epilogue stores, 2 = every workgroup reads tile (0, 0), 4 = no MFMAs
};

#ifndef GO2_EMU
// a 16-byte load from a 4-byte-aligned address (gfx950 takes it as one global_load_dwordx4): rows whose length is not a multiple of 4 floats
typedef float GmF4u __attribute__((ext_vector_type(4), aligned(4)));      // (a vector type: a struct of four floats is split into four loads before the backend sees it)
__device__ __forceinline__ float4 gm_ldu(const float* p) { const GmF4u v = *reinterpret_cast<const GmF4u*>(p); return make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ float gm_keep(float v, bool ok) { return __uint_as_float(__float_as_uint(v) & (ok ? 0xffffffffu : 0u)); }
__device__ __forceinline__ int gm_opaque(int v) { asm volatile("" : "+v"(v)); return v; }      // (the compiler must not see that a clamped index implies the in-range test: it would turn the load back into a branch)
__device__ __forceinline__ float4 gm_masked(const float4 v, unsigned m) {
  return make_float4(gm_keep(v.x, m & 1), gm_keep(v.y, m & 2), gm_keep(v.z, m & 4), gm_keep(v.w, m & 8));
}

// One operand's staging state of a thread: global -> registers -> LDS for k-tiles of BK = 32 or 16 (described for 32).
//   k-contiguous operand (KC): R rows x 32 k; thread = (k quad kq = tid & 7, rows tid >> 3 + 32 p).  LDS tile [R][32] with the quad index XOR-ed
//     by (row >> 1) & 7 (BK = 16: four quads per row, XOR-ed by (row >> 2) & 3): 16 consecutive rows at one k quad — what a ds_read_b128 serves per cycle — fall on 16 distinct bank quads, and so do
//     the 2 rows x 8 quads that 16 consecutive threads write (no padding: the 64 x 128 tile's two stages are 48 KB, three workgroups per CU)
//   k-strided operand:   32 k-rows x R columns; thread = (column quad tid % (R/4), k-rows tid / (R/4) + (1024/R) p).  LDS tile [32][R + 8].
// Loads are branch-free: clamped (valid) addresses, zeroed by a select on the way to LDS — and only in workgroups that touch an edge.
template <int R, bool KC, bool VEC, int BK, int NTHR>
struct GmStage {
  // k-contiguous: QK k-quads per row, RPP rows per pass of the 256 threads; k-strided: QR column quads per k-row, KP k-rows per pass
  static constexpr int QK = BK / 4, RPP = NTHR / QK, QR = R / 4, KP = NTHR / QR, P = KC ? (R + RPP - 1) / RPP : (BK + KP - 1) / KP, LDS_FLOATS = KC ? R * BK : BK * (R + 8);
  static constexpr bool RAGGED = KC ? (R % RPP != 0) : (BK % KP != 0);      // the last pass of the threads covers fewer rows than there are threads (768-thread workgroups)
  static constexpr int SWS = BK == 32 ? 1 : 2;      // swizzle: quad ^= (row >> SWS) & (QK - 1)
  const float* base; int ld, kend; bool edge;
  const float* rowp[P];      // KC: the thread's (clamped) rows
  unsigned rowok;            // KC: bit p = row p in range;  k-strided: 4 column bits
  int c0, ncols;             // k-strided: the thread's (clamped) first column
  float4 buf[P]; unsigned okm;
  __device__ __forceinline__ void init(const float* src, int ld_, int r0, int n, int kend_, int tid) {
    base = src; ld = ld_; kend = kend_; rowok = 0;
    if (KC) {
#pragma unroll
      for (int p = 0; p < P; ++p) { const int lrow = tid / QK + RPP * p, row = r0 + (RAGGED ? min(lrow, R - 1) : lrow);      // (a ragged pass: the spare threads re-read the tile's last row — same lines, no traffic — and do not store)
        rowp[p] = src + (size_t)gm_opaque(min(row, n - 1)) * ld_; rowok |= (unsigned)(row < n) << p; }
      edge = r0 + R > n;
    } else {
      const int col = r0 + 4 * (tid % QR);
      ncols = n; c0 = gm_opaque(VEC ? min(col, n - 4) : min(col, n - 1));
      rowok = VEC ? (col < n ? 15u : 0u) : ((unsigned)(col < n) | (unsigned)(col + 1 < n) << 1 | (unsigned)(col + 2 < n) << 2 | (unsigned)(col + 3 < n) << 3);
      edge = r0 + R > n;
    }
  }
  // one 16-byte piece (p < P) of the k-tile starting at k0 (the kernel issues all pieces of the next tile together, in front of the current tile's MFMAs;
  // issuing them one by one between the MFMA groups was measured slower: DESIGN 10.0)
  template <int p>
  __device__ __forceinline__ void load_piece(int k0, int tid) {
    if (p == 0) okm = 0;
    if (KC) {
      const int k = k0 + 4 * (tid % QK);
      if (VEC) {
        const int kc = gm_opaque(min(k, kend - 4));
        buf[p] = *reinterpret_cast<const float4*>(rowp[p] + kc); okm |= ((rowok >> p & 1) && k < kend ? 15u : 0u) << (4 * p);
      } else if (k0 + BK <= kend) {        // (workgroup-uniform) a full k-tile of unaligned rows: still one 16-byte load per piece
        buf[p] = gm_ldu(rowp[p] + k); okm |= ((rowok >> p & 1) ? 15u : 0u) << (4 * p);
      } else {                              // the ragged last k-tile: element by element, clamped inside the row
        const int k1 = gm_opaque(min(k, kend - 1)), k2 = gm_opaque(min(k + 1, kend - 1)), k3 = gm_opaque(min(k + 2, kend - 1)), k4 = gm_opaque(min(k + 3, kend - 1));
        const unsigned km = (unsigned)(k < kend) | (unsigned)(k + 1 < kend) << 1 | (unsigned)(k + 2 < kend) << 2 | (unsigned)(k + 3 < kend) << 3;
        buf[p] = make_float4(rowp[p][k1], rowp[p][k2], rowp[p][k3], rowp[p][k4]); okm |= ((rowok >> p & 1) ? km : 0u) << (4 * p);
      }
    } else {
      const int k = k0 + (RAGGED ? min(tid / QR + KP * p, BK - 1) : tid / QR + KP * p);
      const float* q = base + (size_t)gm_opaque(min(k, kend - 1)) * ld + c0;
      if (VEC) buf[p] = *reinterpret_cast<const float4*>(q);
      else if (!edge) buf[p] = gm_ldu(q);        // (workgroup-uniform) an interior column tile of unaligned rows
      else { const int n1 = ncols - 1 - c0; buf[p] = make_float4(q[0], q[min(1, n1)], q[min(2, n1)], q[min(3, n1)]); }
      okm |= (k < kend ? rowok : 0u) << (4 * p);
    }
  }
  template <int p = 0>
  __device__ __forceinline__ void load(int k0, int tid) {
    if constexpr (p < P) { load_piece<p>(k0, tid); load<p + 1>(k0, tid); }
  }
  __device__ __forceinline__ void store(float* __restrict__ lds, int k0, int tid) const {
    const bool masked = edge || k0 + BK > kend;          // workgroup-uniform
    if (KC) {
      const int kq = tid % QK, r = tid / QK;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int row = r + RPP * p;
        if (!RAGGED || row < R) *reinterpret_cast<float4*>(lds + row * BK + 4 * (kq ^ ((row >> SWS) & (QK - 1)))) = masked ? gm_masked(buf[p], okm >> (4 * p)) : buf[p];
      }
    } else {
      const int rq = tid % QR, kk = tid / QR;
#pragma unroll
      for (int p = 0; p < P; ++p) if (!RAGGED || kk + KP * p < BK) *reinterpret_cast<float4*>(lds + (kk + KP * p) * (R + 8) + 4 * rq) = masked ? gm_masked(buf[p], okm >> (4 * p)) : buf[p];
    }
  }
  // the MFMA fragment of tile row `r` for k-block kb (8 inputs), lane half g: element e stands for input 8 kb + 4 g + e
  static __device__ __forceinline__ float4 frag(const float* __restrict__ lds, int r, int kb, int g) {
    if (KC) return *reinterpret_cast<const float4*>(lds + r * BK + 4 * ((2 * kb + g) ^ ((r >> SWS) & (QK - 1))));
    const float* q = lds + (8 * kb + 4 * g) * (R + 8) + r;
    return make_float4(q[0], q[R + 8], q[2 * (R + 8)], q[3 * (R + 8)]);
  }
};

#ifdef GM_STAMPS
// TOOL-ONLY (tools/gemm_bench.py --stamps): shader-clock stamps of every workgroup's wave 0 -> g.stamps[workgroup][8]:
// entry, prologue done, k-loop done, end, and the k-loop's time split into load issue / MFMA block / LDS store / barrier wait
#define GM_DECL() long long t0 = 0, t1 = 0, t2 = 0, tl0 = 0, tl1 = 0, tl2 = 0, tl3 = 0, tl4 = 0, s_ld = 0, s_mm = 0, s_st = 0, s_ba = 0
#define GM_T(x) do { __builtin_amdgcn_sched_barrier(0); x = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define GM_ACC() do { s_ld += tl1 - tl0; s_mm += tl2 - tl1; s_st += tl3 - tl2; s_ba += tl4 - tl3; } while (0)
#define GM_OUT() do { if (g.stamps && tid == 0) { long long* o = g.stamps + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 8; o[0] = t0; o[1] = t1; o[2] = t2; o[3] = clock64(); o[4] = s_ld; o[5] = s_mm; o[6] = s_st; o[7] = s_ba; } } while (0)
#else
#define GM_DECL() do { } while (0)
#define GM_T(x) do { } while (0)
#define GM_ACC() do { } while (0)
#define GM_OUT() do { } while (0)
#endif
template <int V> struct GmInt { static constexpr int value = V; };
template <int J0, int J1, class SA, class SB>
__device__ __forceinline__ void gm_issue(SA& sa, SB& sb, int k0, int tid) {
  if constexpr (J0 < J1) {
    if constexpr (J0 < SA::P) sa.template load_piece<J0>(k0, tid); else sb.template load_piece<J0 - SA::P>(k0, tid);
    gm_issue<J0 + 1, J1>(sa, sb, k0, tid);
  }
}

template <int TM, int TN, bool AKC, bool BKC, int EPI, bool VEC, int BK>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) go2nn_gemm_kernel(const GemmArgs g) {      // (>= 3 waves per SIMD: the 128 x 128 tile's
                                                                                                                                  // 64 accumulators + staging would otherwise take 184 registers = 2 waves)
  // 2 x 2 waves; a wave owns TM x TN 32x32 tiles: workgroup tile 64 TM x 64 TN.  (192-row tiles and a twelve-wave 192 x 128 shape were measured and dropped: DESIGN 10.0)
  constexpr int WM = 2, NTHR = 256, NW = 4, BM = 64 * TM, BN = 64 * TN;
  using SA = GmStage<BM, AKC, VEC, BK, NTHR>; using SB = GmStage<BN, BKC, VEC, BK, NTHR>;
  constexpr int ASZ = SA::LDS_FLOATS, BSZ = SB::LDS_FLOATS, LOOP_LDS = 2 * (ASZ + BSZ), EPI_LDS = NW * 32 * 32 * TN;
  __shared__ __attribute__((aligned(16))) float lds[LOOP_LDS > EPI_LDS ? LOOP_LDS : EPI_LDS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, gk = lane >> 5, wm = wave % WM, wn = wave / WM;
  // workgroup -> tile: consecutive workgroup ids go round the 8 XCDs; a XCD gets a contiguous run of tiles (n fastest: the tiles of one row block
  // share their A rows in that XCD's L2)
  const int nb = g.nbm * g.nbn;
  int t = blockIdx.x;
  if ((nb & 7) == 0) t = (t & 7) * (nb >> 3) + (t >> 3);
  const int bm = t / g.nbn, bn = t - bm * g.nbn;
  const int row0 = bm * BM, col0 = bn * BN;
  const int lrow0 = (g.debug & 2) ? 0 : row0, lcol0 = (g.debug & 2) ? 0 : col0;
  const int kbeg = blockIdx.z * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
  const int nk = (kend - kbeg + BK - 1) / BK;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  GM_DECL();
  GM_T(t0);
  SA sa; SB sb;
  sa.init(g.A, g.lda, lrow0, g.M, kend, tid); sb.init(g.B, g.ldb, lcol0, g.N, kend, tid);
  if (nk > 0) { sa.load(kbeg, tid); sb.load(kbeg, tid); sa.store(lds, kbeg, tid); sb.store(lds + ASZ, kbeg, tid); }
  __syncthreads();
  GM_T(t1);
  constexpr int NKB = BK / 8;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1, knext = kbeg + (kt + 1) * BK;
    const bool more = kt + 1 < nk;
    const float* As = lds + cur * (ASZ + BSZ); const float* Bs = As + ASZ;
    GM_T(tl0);
    if (more) { sa.load(knext, tid); sb.load(knext, tid); }          // the next k-tile's global loads fly while this one is multiplied
    GM_T(tl1);
    if (!(g.debug & 4))
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      float4 af[TM], bf[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a) af[a] = SA::frag(As, wm * 32 * TM + a * 32 + i, kb, gk);
#pragma unroll
      for (int b = 0; b < TN; ++b) bf[b] = SB::frag(Bs, wn * 32 * TN + b * 32 + i, kb, gk);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].x, bf[b].x, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].y, bf[b].y, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].z, bf[b].z, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].w, bf[b].w, acc[a][b], 0, 0, 0);
        }
    }
    GM_T(tl2);
    if (kt + 1 < nk) { float* An = lds + (cur ^ 1) * (ASZ + BSZ); sa.store(An, knext, tid); sb.store(An + ASZ, knext, tid); }
    GM_T(tl3);
    __syncthreads();
    GM_T(tl4);
    GM_ACC();
  }
  GM_T(t2);

  // epilogue.  The accumulators leave in the MFMA's C/D layout (column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)): stored as they
  // are that is 16 dword stores per 32x32 tile, each two 128-byte row segments — store-issue-bound, and the input gradient's ELU' operand would come
  // in by as many dword loads.  Instead every wave turns its tile, 32 rows at a time, through its share of the (now free) LDS and works on rows: a lane
  // owns four consecutive columns, reads / writes 16 bytes at a time, 256-byte row segments per 16 lanes.  Column bit 5 is flipped by row bit 2
  // (the half-wave's rows): the two half-waves' dword writes fall on different bank halves without padding.
  {
    constexpr int CT = 32 * TN, LPR = CT / 4, RPI = 64 / LPR, NI = 32 / RPI;      // lanes per row, rows per instruction, instructions per 32-row slab
    float* wl = lds + wave * (32 * CT);                                            // one 32-row slab at a time: 4 x 32 x CT floats fit the k-loop's LDS of every shape
    float* __restrict__ Cz = g.C + (size_t)blockIdx.z * g.c_split_stride;
    const int lc = (lane % LPR) * 4, lr = lane / LPR;
    const int col = col0 + wn * CT + lc;
    const bool cv = g.c_vec != 0 && col + 3 < g.N;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (EPI == EPI_BIAS_ELU) {
      const int c1 = min(col, g.N - 1), c2 = min(col + 1, g.N - 1), c3 = min(col + 2, g.N - 1), c4 = min(col + 3, g.N - 1);
      bias4 = make_float4(g.bias[c1], g.bias[c2], g.bias[c3], g.bias[c4]);
    }
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const int rbase = row0 + wm * 32 * TM + a * 32 + lr;
      float4 y4[NI];
      if (EPI == EPI_DELU_COLSUM) {      // the slab's ELU outputs in flight together (clamped addresses; edge values are dropped below)
#pragma unroll
        for (int n = 0; n < NI; ++n) {
          const float* q = g.Y + (size_t)min(rbase + n * RPI, g.M - 1) * g.ldc;
          if (cv) y4[n] = *reinterpret_cast<const float4*>(q + col);
          else y4[n] = make_float4(q[min(col, g.N - 1)], q[min(col + 1, g.N - 1)], q[min(col + 2, g.N - 1)], q[min(col + 3, g.N - 1)]);
        }
      }
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * gk;
          wl[row * CT + ((b * 32 + i) ^ (gk << 5 & (CT - 1)))] = acc[a][b][r];
        }
#pragma unroll
      for (int n = 0; n < NI; ++n) {
        const int lrow = lr + n * RPI, row = rbase + n * RPI;
        float4 v = *reinterpret_cast<const float4*>(wl + lrow * CT + (lc ^ ((lrow >> 2 & 1) << 5 & (CT - 1))));
        if (EPI == EPI_BIAS_ELU) v = make_float4(elu1(v.x + bias4.x), elu1(v.y + bias4.y), elu1(v.z + bias4.z), elu1(v.w + bias4.w));
        if (EPI == EPI_DELU_COLSUM) {
          const float4 y = y4[n];
          v.x *= y.x > 0.f ? 1.f : y.x + 1.f; v.y *= y.y > 0.f ? 1.f : y.y + 1.f; v.z *= y.z > 0.f ? 1.f : y.z + 1.f; v.w *= y.w > 0.f ? 1.f : y.w + 1.f;
          if (row < g.M) { cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w; }
        }
        if (row < g.M && !(g.debug & 1)) {
          float* o = Cz + (size_t)row * g.ldc + col;
          if (cv) *reinterpret_cast<float4*>(o) = v;
          else { if (col < g.N) o[0] = v.x; if (col + 1 < g.N) o[1] = v.y; if (col + 2 < g.N) o[2] = v.z; if (col + 3 < g.N) o[3] = v.w; }
        }
      }
    }
    if (EPI == EPI_DELU_COLSUM) {
      // column partial sums of this row block: a lane's rows in order -> the lanes that share its columns (xor tree) -> the two row waves; a fixed order
#pragma unroll
      for (int d = LPR; d < 64; d <<= 1) { cs.x += __shfl_xor(cs.x, d); cs.y += __shfl_xor(cs.y, d); cs.z += __shfl_xor(cs.z, d); cs.w += __shfl_xor(cs.w, d); }
      __syncthreads();                      // every wave is done with its LDS quarter
      float* shs = lds;                     // [WM][BN]
      if (lane < LPR) *reinterpret_cast<float4*>(shs + wm * BN + wn * CT + lc) = cs;
      __syncthreads();
      if (tid < BN && col0 + tid < g.N) {
        float t = shs[tid];
#pragma unroll
        for (int w = 1; w < WM; ++w) t += shs[w * BN + tid];
        g.part[(size_t)bm * g.N + col0 + tid] = t;
      }
    }
  }
  GM_OUT();
}

#endif  // !GO2_EMU

// tile shape (TM, TN in 32x32 tiles per wave; workgroup tile 64 TM x 64 TN) for an M x N output with M large: 128 x 128 (16-deep k-tiles, 32 KB of LDS)
// for outputs of >= 512 columns — at M = 24576 that is 768 workgroups = 3 per CU, all resident, at 8 B / clk / CU from the L2; 64 x 128 (48 KB: three
// workgroups per CU, 768 of them for 256 columns) below that; 64 x 64 for narrow outputs or when 128 columns would mostly be padding
static inline int gm_tile_rows(int tm) { return 64 * tm; }
static inline int gm_pick(int n) { return (n + 127) / 128 * 128 <= (n + 63) / 64 * 64 ? 2 : 1; }
static inline void gemm_tile(int M, int N, int* tm, int* tn) {
  static const char* const env = getenv("GO2NN_TILE");        // tools/gemm_bench.py: tile sweep (read once)
  if (env && env[0] >= '1' && env[0] <= '2' && env[1] >= '1' && env[1] <= '2') { *tm = env[0] - '0'; *tn = env[1] - '0'; return; }
  *tn = N > 128 ? gm_pick(N) : 1;
  *tm = (*tn == 2 && N >= 512 && M >= 512) ? 2 : 1;
}
