// go2nn_bx3.h — the learner's GEMMs on the bf16 matrix pipe with fp32 operands: every fp32 value is split EXACTLY into three bf16 planes
// (hi + mid + lo = 8 + 8 + 8 significand bits, round-to-nearest at each level) and a product a . b is formed from six bf16 MFMA terms with fp32 accumulation:
//     a b  ~=  hi hi + hi mid + mid hi + mid mid + hi lo + lo hi          (dropped: mid lo, lo mid, lo lo  <  2^-23 |a b|)
// The dropped terms are below the fp32 rounding of the product itself, and a 512-deep contraction takes 192 accumulator roundings here against 256 with
// v_mfma_f32_32x32x2_f32 — measured against float64 the result is as close as the fp32-MFMA kernels' (tools/gemm3_bench.cpp prints both; tests/test_gpu_mlp_tail.py
// holds the same tolerances for both).  What it buys: v_mfma_f32_32x32x16_bf16 issues every 32 cycles for 16 k, the fp32 form every 64 cycles for 2 k — six terms
// cost 192 cycles per 16 k against 512 (MI355X guide, per-instruction constants).  This is NOT bf16 arithmetic: no operand bit is dropped.
// (included by go2nn_impl.cpp after go2nn_gemm3.h, whose staging class G3Stage and epilogue conventions it shares.)
//
//   go2nn_bx3_split_kernel      weights [N][K] fp32 -> the plane images both GEMM orientations read (once per optimizer step; the matrices are small)
//   go2nn_bx3_kernel<TM, EPI>   forward  Y = elu(X W^T + b)  and input gradient  Gp = (G W) * elu'(Yp) + column sums, grouped (actor + critic tiles in one grid)
//        A operand (activations / gradients, [M][K] fp32 in HBM): staged through registers as in go2nn_gemm3_kernel, split on the way to LDS (22 VALU
//        instructions per 4 values)
//        B operand (weights): read as ready-made plane tiles, a linear 12 KB copy per 128 x 16 tile
//        LDS image of a plane: [row][16 k] bf16 = 32 B per row, the two 16-byte chunks of a row swapped by bit 3 of the row — a fragment read
//        (ds_read_b128: lane i reads chunk g of row i) is conflict-free in each of the instruction's 16-lane groups, and so are the 8-byte stores
// The weight gradient (both operands k-strided, both would need the split) stays on go2nn_wgrad_kernel: DESIGN.md section 5.
#pragma once

#ifndef GO2_EMU
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#endif

#define BX3_BK 16
#ifndef BX3_DA
#define BX3_DA(TM) ((TM) == 1 ? 4 : 2)          /* staging sets of the A operand (tiles of lookahead) */
#endif
#define BX3_SGB_VALU 5          /* VALU issues the scheduler is asked to place behind each MFMA of a tile's first phase */
#define BX3_TILE_BYTES 12288          /* one 128-row x 16-k weight tile: 3 planes x 128 rows x 32 B */

#ifdef GO2_EMU
#define BX3_HD
#else
#define BX3_HD __host__ __device__
#endif
BX3_HD static inline long long bx3_image_bytes(int rows, int con) { return (long long)((rows + 127) / 128) * ((con + BX3_BK - 1) / BX3_BK) * BX3_TILE_BYTES; }

struct Bx3Prob {
  const float* A; const unsigned char* B; float* C; const float* bias; const float* Y; float* part;
  int M, N, K, lda, ldc, nbm, nbn, c_vec, nkt;
  const float* Bf; int ldb;          // weight-gradient mode: the second operand is an activation matrix too (fp32, [rows][N])
  float* wpart; int kx;              // EPI_DELU_WG: Bf [M][ldb] = the input x of the layer below (kx <= 64 columns), wpart [nbm][N][kx] = per-row-tile partials of its weight gradient; C may be NULL
};
struct Bx3Args { Bx3Prob p[2]; int ntiles0, ntiles; long long* stamps; int wg_M, wg_rows; };          // wg_*: the mini-batch's rows and the rows of one slice (weight-gradient mode)
struct Bx3SplitJob { const float* w; unsigned char* img; int N, K; };          // img: forward image, then the transposed one
struct Bx3SplitArgs { Bx3SplitJob j[GO2NN_MAX_SPLIT_JOBS]; int njobs; };

#ifndef GO2_EMU
__device__ __forceinline__ unsigned bx3_pk(float a, float b) { f32x2 v = {a, b}; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }
__device__ __forceinline__ float bx3_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bx3_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
// four fp32 -> three planes of four bf16 (8 bytes each); the residuals are exact (Sterbenz: hi is within a factor 2 of x, so x - hi needs no rounding)
__device__ __forceinline__ void bx3_split4(const f32x4& x, u32x2& h, u32x2& m, u32x2& l) {
  h[0] = bx3_pk(x[0], x[1]); h[1] = bx3_pk(x[2], x[3]);
  const float r0 = x[0] - bx3_lo(h[0]), r1 = x[1] - bx3_hi(h[0]), r2 = x[2] - bx3_lo(h[1]), r3 = x[3] - bx3_hi(h[1]);
  m[0] = bx3_pk(r0, r1); m[1] = bx3_pk(r2, r3);
  const float s0 = r0 - bx3_lo(m[0]), s1 = r1 - bx3_hi(m[0]), s2 = r2 - bx3_lo(m[1]), s3 = r3 - bx3_hi(m[1]);
  l[0] = bx3_pk(s0, s1); l[1] = bx3_pk(s2, s3);
}

// ---- weights -> plane images -----------------------------------------------------------------------------------------------------------------------------
// image of an operand with `rows` rows and a contraction of `con`: [row tile of 128][k tile of 16][plane][row][chunk ^ ((row >> 3) & 1)][8 bf16], zero beyond the
// matrix.  One thread per (tile, row, chunk): 8 values, 3 x 16 bytes out.  blockIdx.y = 2 job + orientation (0: rows = n, con = k; 1: rows = k, con = n).
__global__ void __launch_bounds__(256) go2nn_bx3_split_kernel(const Bx3SplitArgs sa) {
  const int job = blockIdx.y >> 1, tr = blockIdx.y & 1;
  const Bx3SplitJob& j = sa.j[job];
  const int rows = tr ? j.K : j.N, con = tr ? j.N : j.K;
  const int nrt = (rows + 127) / 128, nkt = (con + BX3_BK - 1) / BX3_BK;
  const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
  if (id >= (long long)nrt * nkt * 256) return;
  const int chunk = (int)(id & 1), r = (int)(id >> 1) & 127, tile = (int)(id >> 8), kt = tile % nkt, rt = tile / nkt;
  const int row = rt * 128 + r, c0 = kt * BX3_BK + chunk * 8;
  f32x4 v[2];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = c0 + e;
    const bool in = row < rows && c < con;
    const int rr = min(row, rows - 1), cc = min(c, con - 1);
    const float x = tr ? j.w[(size_t)cc * j.K + rr] : j.w[(size_t)rr * j.K + cc];
    v[e >> 2][e & 3] = in ? x : 0.f;
  }
  u32x2 h0, m0, l0, h1, m1, l1;
  bx3_split4(v[0], h0, m0, l0); bx3_split4(v[1], h1, m1, l1);
  unsigned char* img = j.img + (tr ? bx3_image_bytes(j.N, j.K) : 0) + (size_t)tile * BX3_TILE_BYTES + r * 32 + ((chunk ^ ((r >> 3) & 1)) << 4);
  *reinterpret_cast<u32x4*>(img) = u32x4{h0[0], h0[1], h1[0], h1[1]};
  *reinterpret_cast<u32x4*>(img + 4096) = u32x4{m0[0], m0[1], m1[0], m1[1]};
  *reinterpret_cast<u32x4*>(img + 8192) = u32x4{l0[0], l0[1], l1[0], l1[1]};
}

__device__ __forceinline__ int lane_pos_x(int i, int gk) { return i * 2 + gk; }          // EPI_DELU_WG: position of lane (i, gk)'s 16 bytes inside a 1 KB fragment block of X
// ---- forward / input gradient ---------------------------------------------------------------------------------------------------------------------------
// k-tiles of 16 (one MFMA k-block), two LDS stages, two staging register sets (the loads of tile kt + 2 are issued at the top of tile kt).  One k-tile, per wave:
//     top      global loads of tile kt + 2
//     phase 1  the first three terms' MFMAs on the fragments of tile kt; the staged tile kt + 1 is split and written to the other stage
//     barrier  (tile kt + 1 is in LDS; every wave is past its reads of that stage, they were waited for in front of the MFMAs of tile kt - 1)
//     phase 2  the fragments of tile kt + 1 are requested (inline-asm ds_read_b128, one counted wait at the top of the next tile), the other three terms' MFMAs
template <int OFF> __device__ __forceinline__ void bx3_dsr128(u32x4& d, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF) : "memory"); }
// fragments of one k-tile: hi and mid planes (two sets alternate between the tiles); the lo planes are used by the first two terms only, so ONE set serves every
// tile — it is re-read in phase 2, behind the MFMAs that consumed it
template <int TM> struct Bx3Frags {
  u32x4 a[TM][2], b[2][2];
  __device__ __forceinline__ void opaque() {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int t = 0; t < TM; ++t) g3_opaque(a[t][p]);
#pragma unroll
      for (int t = 0; t < 2; ++t) g3_opaque(b[t][p]);
    }
  }
};
// WG (weight gradient dW [C, Kin] = G^T X over one row slice of the mini-batch): both operands are [rows][columns] fp32 with the CONTRACTION index as their row, so a
// staging thread owns one column and gathers its 8 contraction rows with 8 loads (coalesced across the lanes: 256 B per wave and load), splits them and stores the
// 16 bytes per plane as one fragment chunk; the k-loop, the fragments and the MFMAs are the same.  Output tile = 128 rows of C x 128 columns of Kin of slice `slice`.
template <int TM, int EPI, bool WG = false>
__global__ void __launch_bounds__(256, 2) go2nn_bx3_kernel(const Bx3Args ga) {
  static_assert(!WG || (TM == 2 && EPI == EPI_STORE), "weight-gradient mode: 128 x 128 tiles, plain store");
  static_assert(EPI != EPI_DELU_WG || (TM == 2 && !WG), "input gradient + the weight gradient below: 128-row tiles");
  constexpr int BM = 64 * TM, BN = 128, BK = BX3_BK, TN = 2;
  constexpr int AP = BM * 32, BP = 4096, BOFF = 3 * AP, STAGE = 3 * AP + 3 * BP, LOOP_LDS = 2 * STAGE, EPI_LDS = 4 * 32 * 32 * TN * 4;
  using SA = G3Stage<BM, true, BK>;
  using FR = Bx3Frags<TM>;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LOOP_LDS > EPI_LDS ? LOOP_LDS : EPI_LDS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, gk = lane >> 5, wm = wave & 1, wn = wave >> 1;
  // workgroup -> tile: each problem's tiles dealt to the 8 XCDs in contiguous runs (go2nn_gemm3_kernel)
  int t = blockIdx.x, pi, slice = 0;
  const int n0 = ga.ntiles0, n1 = ga.ntiles - ga.ntiles0;
  if constexpr (WG) {          // workgroup -> (row slice, tile): the slices with index = x mod 8 live on XCD x, all tiles of a slice next to each other (go2nn_wgrad_kernel)
    const int xcd = t & 7, idx = t >> 3;
    slice = (idx / ga.ntiles) * 8 + xcd; t = idx % ga.ntiles;
    pi = t >= n0 ? 1 : 0; t -= pi ? n0 : 0;
  } else if (((n0 | n1) & 7) == 0) { const int x = t & 7, j = t >> 3, c0 = n0 >> 3, c1 = n1 >> 3; pi = j >= c0 ? 1 : 0; t = pi ? x * c1 + (j - c0) : x * c0 + j; }
  else { pi = t >= n0 ? 1 : 0; t -= pi ? n0 : 0; }
  const Bx3Prob& g = ga.p[pi];
  const int bm = t / g.nbn, bn = t - bm * g.nbn;
  const int row0 = bm * BM, col0 = bn * BN;
  const int nk = WG ? (ga.wg_rows + BK - 1) / BK : g.nkt;
  const int m0 = slice * ga.wg_rows;          // (weight-gradient mode: first row of the slice)
  float* const Cout = g.C + (WG ? (size_t)slice * g.M * g.N : 0);

#ifdef GM3_STAMPS
  long long st_[6] = {0, 0, 0, 0, 0, 0};
  const long long wall0_ = wall_clock64();
#endif
  G3_T(0);
  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // the epilogue's geometry and its operands that do not depend on the product: the bias (forward) and the first 32-row slab of the ELU outputs Y whose derivative
  // multiplies the input gradient are requested in front of the k-loop (below, behind the first tiles' loads) — at the head of the epilogue their latency was exposed once per workgroup
  constexpr int CT = 32 * TN, LPR = CT / 4, RPI = 64 / LPR, NI = 32 / RPI;
  const int lc = (lane % LPR) * 4, lr = lane / LPR;
  const int col = col0 + wn * CT + lc;
  const bool cv = g.c_vec != 0 && col + 3 < g.N;
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (EPI == EPI_BIAS_ELU || EPI == EPI_BIAS) {
    const int c1 = min(col, g.N - 1), c2 = min(col + 1, g.N - 1), c3 = min(col + 2, g.N - 1), c4 = min(col + 3, g.N - 1);
    bias4 = make_float4(g.bias[c1], g.bias[c2], g.bias[c3], g.bias[c4]);
  }
  float4 y4[EPI == EPI_DELU_COLSUM ? TM : 1][EPI == EPI_DELU_COLSUM ? NI : 1];
  auto load_y = [&](auto a_c) __attribute__((always_inline)) {
    constexpr int A_ = decltype(a_c)::value;
    if constexpr (EPI == EPI_DELU_COLSUM) {
#pragma unroll
      for (int n = 0; n < NI; ++n) {
        const float* q = g.Y + (size_t)min(row0 + wm * 32 * TM + A_ * 32 + lr + n * RPI, g.M - 1) * g.ldc;
        if (cv) y4[A_][n] = *reinterpret_cast<const float4*>(q + col);
        else y4[A_][n] = make_float4(q[min(col, g.N - 1)], q[min(col + 1, g.N - 1)], q[min(col + 2, g.N - 1)], q[min(col + 3, g.N - 1)]);
      }
    }
  };

  // EPI_DELU_WG: the input x [128 rows of the workgroup][kx] of the layer below, 8 rows at one column per thread and 32-row tile (the epilogue's B fragments)
  const int kx = EPI == EPI_DELU_WG ? g.kx : 0;
  const bool two = kx > 32;
  const int xj = tid & 31, xjt = (tid >> 5) & 1, xhk = (tid >> 6) & 1, xkb = tid >> 7;
  float xv[EPI == EPI_DELU_WG ? 4 : 1][8];
  auto issue_xv = [&]() __attribute__((always_inline)) {
    if constexpr (EPI == EPI_DELU_WG) {
      if (xjt == 0 || two) {
        const float* const xq = g.Bf + min(xjt * 32 + xj, kx - 1);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
          for (int e = 0; e < 8; ++e) xv[rt][e] = xq[(size_t)gm_opaque(min(row0 + rt * 32 + 16 * xkb + (e & 3) + 8 * (e >> 2) + 4 * xhk, g.M - 1)) * g.ldb];
      }
    }
  };
  SA sa; sa.init(g.A, g.lda, row0, g.M, WG ? 4 : g.K, tid);
  // A: DA staging sets — the loads of tile kt + DA are issued at the top of tile kt (HBM latency); B: one set (L2 hits: re-issued as soon as it is committed)
  constexpr int DA = BX3_DA(TM);
  f32x4 ba[DA][WG ? 4 : SA::P]; u32x4 bb[3];          // (weight-gradient mode: a set is 8 values of G's column and 8 of X's)
  // weight-gradient staging: thread -> (column tid % 128 of the tile, contraction half tid / 128).  Buffer loads (round 6): the descriptor of a k-tile starts at its
  // first row and ends with the matrix, so rows beyond M read as zero and a column beyond Kin gets an offset that is out of range — no clamps, no selects, and the
  // per-lane offsets of the 8 rows are loop invariants (the address arithmetic and the edge masks were half of the kernel's 6.3 VALU instructions per MFMA)
  const int scol = tid & 127, skg = tid >> 7;
  unsigned voa[WG ? 8 : 1], vob[WG ? 8 : 1];
  if constexpr (WG) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      voa[j] = (unsigned)(((skg * 8 + j) * g.lda + row0 + scol) * 4);                                        // G [rows][C]: output row = column of G (C in whole tiles)
      vob[j] = col0 + scol < g.N ? (unsigned)(((skg * 8 + j) * g.ldb + col0 + scol) * 4) : 0x7ffffff0u;      // X [rows][Kin]
    }
  }
  auto wg_rsrc = [&](const float* base, int ld, int mrow) __attribute__((always_inline)) {
    const long long left = (long long)(ga.wg_M - mrow) * ld * 4;
    const float* p0 = base + (size_t)mrow * ld;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)p0), hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)p0 >> 32));
    const unsigned nb = __builtin_amdgcn_readfirstlane(left > 0 ? (unsigned)(left < 0x7fffffe0ll ? left : 0x7fffffe0ll) : 0u);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uintptr_t)hi << 32) | lo), 0, nb, 0x00020000);
  };
  auto issue_wg = [&](int kt, f32x4 (&w)[WG ? 4 : SA::P]) __attribute__((always_inline)) {
    if constexpr (WG) {
      const int mrow = m0 + kt * BK;
      const auto ra = wg_rsrc(g.A, g.lda, mrow), rb = wg_rsrc(g.Bf, g.ldb, mrow);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        w[j >> 2][j & 3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, voa[j], 0, 0));
        w[2 + (j >> 2)][j & 3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, vob[j], 0, 0));
      }
    }
  };
  auto commit_wg = [&](int stage, int kt, const f32x4 (&w)[WG ? 4 : SA::P]) __attribute__((always_inline)) {
    if constexpr (WG) {
      unsigned char* st = lds + stage * STAGE + scol * 32 + ((skg ^ ((scol >> 3) & 1)) << 4);
      const f32x4 a0 = w[0], a1 = w[1];          // (rows beyond M were read as zero; a slice is whole k-tiles, so the loop never reaches into the next one)
      u32x2 h0, m0_, l0, h1, m1, l1;
      bx3_split4(a0, h0, m0_, l0); bx3_split4(a1, h1, m1, l1);
      *reinterpret_cast<u32x4*>(st) = u32x4{h0[0], h0[1], h1[0], h1[1]};
      *reinterpret_cast<u32x4*>(st + AP) = u32x4{m0_[0], m0_[1], m1[0], m1[1]};
      *reinterpret_cast<u32x4*>(st + 2 * AP) = u32x4{l0[0], l0[1], l1[0], l1[1]};
      bx3_split4(w[2], h0, m0_, l0); bx3_split4(w[3], h1, m1, l1);
      *reinterpret_cast<u32x4*>(st + BOFF) = u32x4{h0[0], h0[1], h1[0], h1[1]};
      *reinterpret_cast<u32x4*>(st + BOFF + BP) = u32x4{m0_[0], m0_[1], m1[0], m1[1]};
      *reinterpret_cast<u32x4*>(st + BOFF + 2 * BP) = u32x4{l0[0], l0[1], l1[0], l1[1]};
    }
  };
  const unsigned char* bimg = WG ? nullptr : g.B + (size_t)bn * nk * BX3_TILE_BYTES + tid * 16;
  auto issue_b = [&](int kt) __attribute__((always_inline)) {
    if constexpr (WG) return;
    const unsigned char* s = bimg + (size_t)kt * BX3_TILE_BYTES;
#pragma unroll
    for (int p = 0; p < 3; ++p) bb[p] = *reinterpret_cast<const u32x4*>(s + p * 4096);
  };
  // staging thread -> LDS: row = tid / 4 + 64 p, k-quad kq = tid % 4 (half a 16-byte chunk)
  const int kq = tid & 3, srow = tid >> 2;
  const unsigned st_off = (unsigned)(srow * 32 + (((kq >> 1) ^ ((srow >> 3) & 1)) << 4) + ((kq & 1) << 3));          // (+ 64 p rows: the same swizzle bit)
  auto commit = [&](auto fix_c, int stage, int kt, const f32x4 (&ba)[SA::P]) __attribute__((always_inline)) {
    constexpr bool FIX = decltype(fix_c)::value;
    unsigned char* st = lds + stage * STAGE;
#pragma unroll
    for (int p = 0; p < SA::P; ++p) {
      f32x4 v = ba[p];
      if (FIX) {        // the k-tile that crosses K: quads the clamp moved left are shifted back, k >= K is zero (go2nn_gemm3.h: G3Stage::commit_piece<true>)
        const int k = kt * BK + sa.lead, sh = k - min(k, g.K - 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { float tv = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) tv = (q == e + sh) ? ba[p][q] : tv;
          v[e] = k + e < g.K ? tv : 0.f; }
      }
      u32x2 h, m, l; bx3_split4(v, h, m, l);
      unsigned char* d = st + st_off + p * (64 * 32);
      *reinterpret_cast<u32x2*>(d) = h; *reinterpret_cast<u32x2*>(d + AP) = m; *reinterpret_cast<u32x2*>(d + 2 * AP) = l;
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(st + BOFF + p * 4096 + tid * 16) = bb[p];
  };
  // fragment reads: lane i of a 32-row tile reads chunk gk of row i (stage, plane and tile offsets are instruction immediates)
  const unsigned lbase = (unsigned)(uintptr_t)lds;
  const unsigned fch = (unsigned)((gk ^ ((i >> 3) & 1)) << 4);
  const unsigned fa = lbase + (unsigned)((wm * 32 * TM + i) * 32) + fch, fb = lbase + (unsigned)(BOFF + (wn * 64 + i) * 32) + fch;
  FR f0, f1; u32x4 la[TM], lb[2];
  auto read_frags = [&](auto stage_c, FR& f) __attribute__((always_inline)) {
    constexpr int SOFF = decltype(stage_c)::value * STAGE;
    g3_for<0, TM>([&](auto a_c) __attribute__((always_inline)) { constexpr int A = decltype(a_c)::value; bx3_dsr128<SOFF + 2 * AP + A * 32 * 32>(la[A], fa); });
    g3_for<0, 2>([&](auto b_c) __attribute__((always_inline)) { constexpr int B = decltype(b_c)::value; bx3_dsr128<SOFF + 2 * BP + B * 32 * 32>(lb[B], fb); });
    g3_for<0, 2>([&](auto p_c) __attribute__((always_inline)) {
      constexpr int P = decltype(p_c)::value;
      g3_for<0, TM>([&](auto a_c) __attribute__((always_inline)) { constexpr int A = decltype(a_c)::value; bx3_dsr128<SOFF + P * AP + A * 32 * 32>(f.a[A][P], fa); });
      g3_for<0, 2>([&](auto b_c) __attribute__((always_inline)) { constexpr int B = decltype(b_c)::value; bx3_dsr128<SOFF + P * BP + B * 32 * 32>(f.b[B][P], fb); });
    });
  };
  // six terms, small ones first (plane 0 hi, 1 mid, 2 lo); consecutive MFMAs go to different accumulators
  auto mfmas = [&](auto q0_c, auto q1_c, const FR& f) __attribute__((always_inline)) {
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    g3_for<decltype(q0_c)::value, decltype(q1_c)::value>([&](auto q_c) __attribute__((always_inline)) {
      constexpr int Q = decltype(q_c)::value;
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          const u32x4 av = PA[Q] == 2 ? la[a] : f.a[a][PA[Q] & 1], bv = PB[Q] == 2 ? lb[b] : f.b[b][PB[Q] & 1];
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[a][b], 0, 0, 0);
        }
    });
  };
  auto lo_opaque = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < TM; ++t) g3_opaque(la[t]);
#pragma unroll
    for (int t = 0; t < 2; ++t) g3_opaque(lb[t]);
  };

  // One k-tile, held in stage CUR (= kt & 1).  MORE: tile kt + 1 follows.
  // A contraction that is not a multiple of 16 leaves a ragged last tile: the pipelined loop runs the nkp whole tiles, the ragged one is a pass of its own behind it
  // (its displaced quads are put right on the way to LDS: a branch in the loop's commit would keep the scheduler from placing the split between the MFMAs)
  const bool ragged = !WG && (g.K & (BK - 1)) != 0;
  const int nkp = ragged ? nk - 1 : nk;
  // (set indices are compile-time constants: a runtime index would put the staging arrays into scratch) tile kt's data sits in set kt % DA
  auto tile = [&](auto cur_c, auto set_c, const bool MORE, int kt) __attribute__((always_inline)) {
    constexpr int CUR = decltype(cur_c)::value, SET = decltype(set_c)::value;
    auto& a_ld = ba[SET]; auto& a_st = ba[(SET + 1) % DA];          // tile kt was committed during tile kt - 1: its set takes tile kt + DA; tile kt + 1 is committed now
    FR& fc = CUR ? f1 : f0; FR& fn = CUR ? f0 : f1;
    { const int k2 = min(kt + DA, nkp - 1); if constexpr (WG) issue_wg(k2, a_ld); else sa.issue(k2 * BK, a_ld); }          // (past the last tile: a re-read that is never used — straight-line code)
    g3_wait_lgkm<0>();
    fc.opaque(); lo_opaque();
    mfmas(G3Int<0>{}, G3Int<3>{}, fc);
    if (MORE) {          // (workgroup-uniform)
      if constexpr (WG) commit_wg(CUR ^ 1, kt + 1, a_st); else commit(G3Int<0>{}, CUR ^ 1, kt + 1, a_st);
      issue_b(min(kt + 2, nkp - 1));
      // the split's VALU work and the LDS stores between the MFMAs (a bf16 MFMA occupies the matrix pipe for 32 cycles: room for ~6 other issues)
#pragma unroll
      for (int q = 0; q < 6 * TM; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, BX3_SGB_VALU, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        if (q < 3) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      read_frags(G3Int<CUR ^ 1>{}, fn);
      __builtin_amdgcn_sched_barrier(0);
    }
    mfmas(G3Int<3>{}, G3Int<6>{}, fc);
    __builtin_amdgcn_sched_barrier(0);
  };
  constexpr G3Int<0> I0{}; constexpr G3Int<1> I1{};
  if (nkp > 0) {
    issue_b(0);
    g3_for<0, DA>([&](auto d_c) __attribute__((always_inline)) { constexpr int D_ = decltype(d_c)::value;
      if constexpr (WG) issue_wg(min(D_, nkp - 1), ba[D_]); else sa.issue(min(D_, nkp - 1) * BK, ba[D_]); });
    // (BEHIND the first tiles' loads: vmcnt counts in order, in front of them the Y loads stood between the first tile and its commit — prologue 9.6 k instead of 4.9 k ticks)
    if constexpr (TM == 2) load_y(G3Int<0>{});          // (64-row tiles: 32 more registers would cost the third workgroup per CU; 192-row tiles have none to spare)
    issue_xv();
    if constexpr (WG) commit_wg(0, 0, ba[0]); else commit(I0, 0, 0, ba[0]);
    if (nkp > 1) issue_b(1);
    __syncthreads();
    read_frags(I0, f0);
    G3_T(1);
    // tiles in groups of lcm(2, DA): stage parity and staging set are compile-time constants of a tile's code; MORE (tile kt + 1 follows) is a uniform branch
    constexpr int G = (DA % 2) ? 2 * DA : DA;
    for (int kt = 0; kt < nkp; kt += G)
      g3_for<0, G>([&](auto j_c) __attribute__((always_inline)) { constexpr int J = decltype(j_c)::value; if (kt + J < nkp) tile(G3Int<J & 1>{}, G3Int<J % DA>{}, kt + J + 1 < nkp, kt + J); });
  } else if constexpr (TM == 2) {
    issue_xv();
    load_y(G3Int<0>{});          // (a contraction shorter than one k-tile — the 8-wide latent of the toy CTS networks — has no pipelined loop: the first slab's ELU outputs are requested here.
  }                              //  Round 5: found by the CTS student step at K = 8, where the slab was read uninitialised)
  if constexpr (!WG) if (ragged) {
    __syncthreads();
    sa.issue((nk - 1) * BK, ba[0]); issue_b(nk - 1);
    commit(I1, 0, nk - 1, ba[0]);
    __syncthreads();
    read_frags(I0, f0);
    g3_wait_lgkm<0>();
    f0.opaque(); lo_opaque();
    mfmas(G3Int<0>{}, G3Int<6>{}, f0);
  }
  G3_T(2);
  __syncthreads();          // every wave is done with the stages: the epilogue turns tiles through the same LDS
  G3_T(3);

  if constexpr (EPI == EPI_DELU_WG) {
    // ---- input gradient whose tile stays on the chip: Gp = (G W) * elu'(Yp) in the ACCUMULATORS' layout (acc[a][b][r] = row (r & 3) + 8 (r >> 2) + 4 gk of the wave's
    // 32-row tile a, column i of its 32-column tile b), its column sums, and the weight gradient of the layer below  dWp [N, kx] = Gp^T X  of the workgroup's 128 rows:
    // with the contraction index = the row, a lane's accumulator registers 8 kb .. 8 kb + 7 ARE an A fragment of v_mfma_f32_32x32x16_bf16 (m = column i of Gp, the lane's
    // 8 k values = rows {0..3, 8..11} + 4 gk + 16 kb — the order inside a contraction is free as long as X is gathered in the same one).  Gp goes to HBM only if C is set.
#if defined(GM3_STAMPS) && defined(WG_T3)          /* tools only: G3_T(3) / G3_T(5) at phase points WG_T3 / WG_T5 of this epilogue (1: X in LDS, 2: ELU' + sums + stores issued, 3: MFMAs done) */
#define WG_STAMP(n) do { if (WG_T3 == (n)) G3_T(3); if (WG_T5 == (n)) G3_T(5); } while (0)
#else
#define WG_STAMP(n) do { } while (0)
#endif
    const int rw = row0 + wm * 64, cw = col0 + wn * 64;
    // X [128 rows of the workgroup][kx] -> LDS as B fragments, once per workgroup (the loop's stages are free: the barrier above): item = (32-row tile rt, k-block kb,
    // row half hk, column tile jt, column j) = the 8 rows {0..3, 8..11} + 4 hk + 16 kb of tile rt at one column = 16 bytes per plane at
    //   [plane][rt][kb][jt][j][hk]  — a wave's fragment read (lane (i, gk) -> j = i, hk = gk) is 1 KB contiguous.  Thread -> (j, jt, hk, kb) = bits of tid, rt = 0 .. 3.
    // X's loads were issued in front of the k-loop (issue_xv: mostly L2 hits — the four column tiles of a row tile read the same rows); the ELU outputs Y of both
    // 32-row tiles are requested here, all before the first use: ONE exposed latency.
    const bool xon = xjt == 0 || two;
    float y[2][2][16];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* q = g.Y + (size_t)gm_opaque(min(rw + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * gk, g.M - 1)) * g.ldc;
#pragma unroll
        for (int b = 0; b < 2; ++b) y[a][b][r] = q[min(cw + b * 32 + i, g.N - 1)];
      }
    WG_STAMP(1);
    float cs[2] = {0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool in = rw + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * gk < g.M;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const float v = gm_keep(acc[a][b][r] * (y[a][b][r] > 0.f ? 1.f : y[a][b][r] + 1.f), in);
          acc[a][b][r] = v; cs[b] += v;
        }
      }
    // column sums: one partial row per 64 data rows = this wave's rows (the other row half gk is the lane 32 away)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const float t2 = cs[b] + __shfl_xor(cs[b], 32);
      const int c = cw + b * 32 + i;
      if (gk == 0 && c < g.N && (bm * 2 + wm) * 64 < g.M) g.part[(size_t)(bm * 2 + wm) * g.N + c] = t2;
    }
    if (g.C) {          // (workgroup-uniform) Gp is wanted in HBM too: every wave turns its tiles, 32 rows at a time, through its LDS quarter for 16-byte row stores (the plain epilogue's way)
      float* wl = reinterpret_cast<float*>(lds) + wave * (32 * CT);
#pragma unroll
      for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * gk, cc = b * 32 + i;
            wl[row * CT + (cc ^ (gk << 5 & (CT - 1)))] = acc[a][b][r];
          }
#pragma unroll
        for (int n = 0; n < NI; ++n) {
          const int lrow = lr + n * RPI, row = rw + a * 32 + lrow;
          const float4 v = *reinterpret_cast<const float4*>(wl + lrow * CT + (lc ^ ((lrow >> 2 & 1) << 5 & (CT - 1))));
          if (row < g.M) {
            float* o = g.C + (size_t)row * g.ldc + col;
            if (cv) *reinterpret_cast<float4*>(o) = v;
            else { if (col < g.N) o[0] = v.x; if (col + 1 < g.N) o[1] = v.y; if (col + 2 < g.N) o[2] = v.z; if (col + 3 < g.N) o[3] = v.w; }
          }
        }
      }
      __syncthreads();          // the quarters are read: X's fragments go to the same memory
    }
    if (xon) {
      const bool in = xjt * 32 + xj < kx;
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        u32x2 h0, m0_, l0, h1, m1, l1;
        bx3_split4(f32x4{gm_keep(xv[rt][0], in), gm_keep(xv[rt][1], in), gm_keep(xv[rt][2], in), gm_keep(xv[rt][3], in)}, h0, m0_, l0);
        bx3_split4(f32x4{gm_keep(xv[rt][4], in), gm_keep(xv[rt][5], in), gm_keep(xv[rt][6], in), gm_keep(xv[rt][7], in)}, h1, m1, l1);
        unsigned char* d = lds + ((((rt * 2 + xkb) * 2 + xjt) * 32 + xj) * 2 + xhk) * 16;
        *reinterpret_cast<u32x4*>(d) = u32x4{h0[0], h0[1], h1[0], h1[1]};
        *reinterpret_cast<u32x4*>(d + 16384) = u32x4{m0_[0], m0_[1], m1[0], m1[1]};
        *reinterpret_cast<u32x4*>(d + 32768) = u32x4{l0[0], l0[1], l1[0], l1[1]};
      }
    }
    WG_STAMP(2);
    f32x16 o[2][2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[b][jt][r] = 0.f;
    __syncthreads();          // X's fragments are in LDS
    const unsigned xfrag = lbase + (unsigned)(lane_pos_x(i, gk) * 16);
    g3_for<0, 4>([&](auto q_c) __attribute__((always_inline)) {
      constexpr int Q = decltype(q_c)::value, A_ = Q >> 1, KB = Q & 1;
      u32x4 gp[2][3], xp[2][3];
      const unsigned xa = xfrag + (unsigned)((((wm * 2 + A_) * 2 + KB) * 2) * 1024);
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        if (jt == 1 && !two) break;
        g3_for<0, 3>([&](auto p_c) __attribute__((always_inline)) { constexpr int P = decltype(p_c)::value; bx3_dsr128<P * 16384>(xp[jt][P], xa + jt * 1024); });
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        u32x2 h0, m0_, l0, h1, m1, l1;
        bx3_split4(f32x4{acc[A_][b][8 * KB], acc[A_][b][8 * KB + 1], acc[A_][b][8 * KB + 2], acc[A_][b][8 * KB + 3]}, h0, m0_, l0);
        bx3_split4(f32x4{acc[A_][b][8 * KB + 4], acc[A_][b][8 * KB + 5], acc[A_][b][8 * KB + 6], acc[A_][b][8 * KB + 7]}, h1, m1, l1);
        gp[b][0] = u32x4{h0[0], h0[1], h1[0], h1[1]}; gp[b][1] = u32x4{m0_[0], m0_[1], m1[0], m1[1]}; gp[b][2] = u32x4{l0[0], l0[1], l1[0], l1[1]};
      }
      g3_wait_lgkm<0>();
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int p = 0; p < 3; ++p) g3_opaque(xp[jt][p]);
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};          // the six terms of the main loop, small ones first (0 hi, 1 mid, 2 lo)
#pragma unroll
      for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int jt = 0; jt < 2; ++jt) {
            if (jt == 1 && !two) continue;
            o[b][jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, gp[b][PA[t6]]), __builtin_bit_cast(bf16x8, xp[jt][PB[t6]]), o[b][jt], 0, 0, 0);
          }
    });
    WG_STAMP(3);
    __syncthreads();          // every wave is done with X's fragments: the exchange below reuses the memory
    // the two row halves of the workgroup (waves wm = 0, 1) are added through LDS (fixed order), then ONE partial tile [128 columns of Gp][kx] per workgroup
    float* ex = reinterpret_cast<float*>(lds) + wn * (64 * 64);
    if (wm == 1) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
          for (int r = 0; r < 16; ++r) ex[((b * 2 + jt) * 16 + r) * 64 + lane] = o[b][jt][r];
    }
    __syncthreads();
    if (wm == 0) {
      float* const wp = g.wpart + (size_t)bm * g.N * kx;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
          if (jt == 1 && !two) continue;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int c = cw + b * 32 + (r & 3) + 8 * (r >> 2) + 4 * gk, j = jt * 32 + i;
            if (c < g.N && j < kx) wp[(size_t)c * kx + j] = o[b][jt][r] + ex[((b * 2 + jt) * 16 + r) * 64 + lane];
          }
        }
    }
  } else {
  // epilogue: go2nn_gemm3_kernel's (every wave turns its tile, 32 rows at a time, through its LDS quarter; 16-byte row accesses)
  g3_for<(TM == 2 ? 1 : 0), TM>([&](auto a_c) __attribute__((always_inline)) { load_y(a_c); });
  {
    float* wl = reinterpret_cast<float*>(lds) + wave * (32 * CT);
    float4 cs[EPI == EPI_DELU_COLSUM ? TM : 1];          // column sums of the wave's 32-row tiles, one each
#pragma unroll
    for (int a = 0; a < (EPI == EPI_DELU_COLSUM ? TM : 1); ++a) cs[a] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const int rbase = row0 + wm * 32 * TM + a * 32 + lr;
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * gk;
          const int cc = b * 32 + i;
          wl[row * CT + (cc ^ (gk << 5 & (CT - 1)))] = acc[a][b][r];
        }
#pragma unroll
      for (int n = 0; n < NI; ++n) {
        const int lrow = lr + n * RPI, row = rbase + n * RPI;
        float4 v = *reinterpret_cast<const float4*>(wl + lrow * CT + (lc ^ ((lrow >> 2 & 1) << 5 & (CT - 1))));
        if (EPI == EPI_BIAS_ELU) v = make_float4(elu1(v.x + bias4.x), elu1(v.y + bias4.y), elu1(v.z + bias4.z), elu1(v.w + bias4.w));
        if (EPI == EPI_BIAS) v = make_float4(v.x + bias4.x, v.y + bias4.y, v.z + bias4.z, v.w + bias4.w);
        if (EPI == EPI_DELU_COLSUM) {
          const float4 y = y4[EPI == EPI_DELU_COLSUM ? a : 0][EPI == EPI_DELU_COLSUM ? n : 0];
          v.x *= y.x > 0.f ? 1.f : y.x + 1.f; v.y *= y.y > 0.f ? 1.f : y.y + 1.f; v.z *= y.z > 0.f ? 1.f : y.z + 1.f; v.w *= y.w > 0.f ? 1.f : y.w + 1.f;
          if (row < g.M) { float4& c4 = cs[EPI == EPI_DELU_COLSUM ? a : 0]; c4.x += v.x; c4.y += v.y; c4.z += v.z; c4.w += v.w; }
        }
        if (row < g.M) {
          float* o = Cout + (size_t)row * g.ldc + col;
#ifdef BX3_NT_STORE          /* tools only (round 5 prototype: non-temporal epilogue stores, profiles/r5_gemm_store_variants.txt) */
          if (cv) __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(o));
#else
          if (cv) *reinterpret_cast<float4*>(o) = v;
#endif
          else { if (col < g.N) o[0] = v.x; if (col + 1 < g.N) o[1] = v.y; if (col + 2 < g.N) o[2] = v.z; if (col + 3 < g.N) o[3] = v.w; }
        }
      }
    }
    if (EPI == EPI_DELU_COLSUM) {
      // one partial row per 64 data rows whatever the tile height (the caller's row count, go2nn_linear_backward_input_group_rows, does not depend on the kernel): the
      // workgroup's 2 TM row tiles of 32 (wave wm, tile a -> wm TM + a) are summed in pairs
      __syncthreads();
      float* shs = reinterpret_cast<float*>(lds);                     // [2 TM][BN]
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        float4 c4 = cs[EPI == EPI_DELU_COLSUM ? a : 0];
#pragma unroll
        for (int d = LPR; d < 64; d <<= 1) { c4.x += __shfl_xor(c4.x, d); c4.y += __shfl_xor(c4.y, d); c4.z += __shfl_xor(c4.z, d); c4.w += __shfl_xor(c4.w, d); }
        if (lane < LPR) *reinterpret_cast<float4*>(shs + (wm * TM + a) * BN + wn * CT + lc) = c4;
      }
      __syncthreads();
      if (tid < BN && col0 + tid < g.N) {
#pragma unroll
        for (int q = 0; q < TM; ++q)
          if ((bm * TM + q) * 64 < g.M) g.part[(size_t)(bm * TM + q) * g.N + col0 + tid] = shs[(2 * q) * BN + tid] + shs[(2 * q + 1) * BN + tid];
      }
    }
  }
  }
#ifdef GM3_STAMPS
  G3_T(4);
  if (ga.stamps && (tid & 63) == 0) { long long* o = ga.stamps + ((size_t)blockIdx.x * 4 + wave) * 8; o[0] = st_[0]; o[1] = st_[1]; o[2] = st_[2]; o[3] = st_[3]; o[4] = st_[4]; o[5] = wall_clock64(); o[6] = wall0_; o[7] = st_[5]; }
#endif
}
#endif  // !GO2_EMU
