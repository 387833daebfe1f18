// go2nn_gemm3.h — round 4: the learner's GEMMs as GROUPED launches (actor and critic problems of one layer in one grid) with a hand-placed k-loop
// (included by go2nn_impl.cpp after go2nn_gemm.h, whose staging class GmStage and epilogue conventions it shares).
//
// PPO.update (rsl_rl/rsl_rl/algorithms/ppo.py:120-187 -> autograd over modules/actor_critic.py:50-75) runs the SAME layer of two independent MLPs
// back to back; as two launches on two HIP streams the pair takes 2 x one network (profiles/r3_gemm_bench.txt) and needs a second stream.  Here a
// layer's two problems are one grid: tile list = problem 0's tiles, then problem 1's.
//
//   go2nn_gemm3_kernel<TM, TN, BKC, EPI, VEC, BK>    forward (BKC: W k-contiguous) and input gradient (!BKC: W rows are k)
//     k-loop, per k-tile of BK (two LDS stages, as before), per wave:
//       top        the NEXT tile's global loads are issued (compiler-visible loads into the staging registers)
//       k-block kb the fragments of k-block kb + 1 are requested (inline-asm ds_read into the OTHER fragment register set) before the MFMAs of
//                  k-block kb issue; one counted s_waitcnt lgkmcnt(#reads of one k-block) in front of a block's MFMAs — the reads of a block
//                  have had a whole block of MFMAs (1024 cycles for a 2 x 2 wave tile) to land
//       last block the staged registers go to the other LDS stage (behind the compiler's own vmcnt wait), HALF of the block's MFMAs issue, ONE
//                  barrier, the next tile's first fragments are requested, the other half of the MFMAs issue: the barrier and the read latency
//                  sit in the shadow of MFMAs that are already in the pipe
//     MFMAs of a block are ordered k-step outermost, so consecutive instructions hit different accumulators (a same-accumulator chain stalls
//     43 cycles on any instruction slipped in between: MI355X guide, per-instruction constants)
//     !BKC: the k-strided operand keeps its natural [k][n] orientation in LDS and is read 8 bytes per lane at [k][n0 + 2 i]: the two values are
//     the SAME k of two DIFFERENT 32-column tiles (tile b = columns n0 + 2 i + b) — no transpose, half the LDS instructions of four ds_read_b32;
//     the epilogue's LDS turn undoes the column interleave.
//
//   go2nn_wgrad_kernel<TA, TN, VECA, VECB>    weight gradient dW [C, Kin] = G^T X, contraction over the M rows of the mini-batch
//     Both operands have the contraction index as their ROW: a lane's MFMA operand for k-step (rows m, m + 1) is G[m + g][c0 + 2 i .. + 1] /
//     X[m + g][k0 + 4 i .. + 3] — 8 / 16 contiguous bytes, a half-wave reads 256 / 512 contiguous bytes.  So the operands go from global memory
//     STRAIGHT into MFMA registers (a register ring a few k-steps deep), no LDS, no barrier in the loop; the values of one load feed 2 resp. 4
//     column-interleaved tiles.  A workgroup's four waves take four row ranges of ONE output tile (64 x 32 TN) and are summed through LDS in
//     the fixed order w0 + w1 + w2 + w3; the grid's row slices leave [slice][C][Kin] partials that go2nn_sum_rows adds in slice order
//     (bit-reproducible).  Slices of one row range sit on one XCD (workgroup id % 8), so the tiles that re-read the same rows share an L2.
#pragma once

struct Gemm3Prob {
  const float *A, *B; float* C; const float* bias; const float* Y; float* part;
  int M, N, K, lda, ldb, ldc, nbm, nbn, c_vec;
};
struct Gemm3Args { Gemm3Prob p[2]; int ntiles0, ntiles; long long* stamps; };

struct WgProb { const float *G, *X; float* part; int C, Kin, ntc, ntk; };
struct WgArgs { WgProb p[2]; int M, rows_per_slice, nsplit, tiles0, tiles; };

#ifndef GO2_EMU
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float GmF2u __attribute__((ext_vector_type(2), aligned(4)));

template <int V> struct G3Int { static constexpr int value = V; };
template <int I0, int I1, class F>
__device__ __forceinline__ void g3_for(F&& f) {
  if constexpr (I0 < I1) { f(G3Int<I0>{}); g3_for<I0 + 1, I1>(f); }
}
template <int OFF> __device__ __forceinline__ void g3_dsr128(f32x4& d, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF) : "memory"); }
template <int OFF> __device__ __forceinline__ void g3_dsr64(f32x2& d, unsigned addr) { asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF) : "memory"); }
template <int OFF> __device__ __forceinline__ void g3_dsr32(float& d, unsigned addr) { asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF) : "memory"); }
template <int N> __device__ __forceinline__ void g3_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" : : "i"(N) : "memory"); }
template <class T> __device__ __forceinline__ void g3_opaque(T& v) { asm volatile("" : "+v"(v)); }      // (behind a wait: no consumer of v is scheduled above it)

// Staging of one operand's k-tile, global -> registers -> LDS (LDS images as GmStage's: k-contiguous [R][BK] with the 16-byte quads XOR-swizzled by the row,
// k-strided [BK][R + 8]).  Leaner than GmStage — the kernel keeps TWO register sets per operand in flight:
//   * addresses are a wave-uniform base (SGPRs) + one 32-bit offset per piece
//   * every load is a 16-byte load from a 4-byte-aligned address (one global_load_dwordx4 on gfx950 whatever the row pitch: 45- and 263-float rows included),
//     issued unconditionally from a CLAMPED address: rows / columns beyond the matrix re-read its last row / last column quad, and a k-quad that would cross
//     the end of a row starts further left (k <= K - 4) — no branch around a load, so the compiler's vmcnt count stays exact
//   * rows and columns beyond the matrix need no zero fill: they only feed accumulators of outputs that the epilogue never stores (finite values: they are
//     real matrix entries).  Only the CONTRACTION tail matters: in the one k-tile that crosses K, the displaced quads of a k-contiguous operand are shifted
//     back and zero-filled, the rows k >= K of a k-strided operand zeroed, on the way to LDS (commit_piece<true>); every other tile stores what it loaded
// Needs K >= 4, and N a multiple of 4 for a k-strided operand (a column quad is then inside the matrix or entirely outside).
template <int R, bool KC, int BK>
struct G3Stage {
  static constexpr int QK = BK / 4, RPP = 256 / QK, QR = R / 4, KP = 256 / QR, P = KC ? R / RPP : BK / KP, LDS_FLOATS = KC ? R * BK : BK * (R + 8);
  static constexpr int SWS = BK == 32 ? 1 : 2;
  static_assert(KC ? (R % RPP == 0) : (BK % KP == 0), "whole passes of the 256 threads");
  const float* base; int ld, K;
  unsigned off[KC ? P : 1];        // KC: byte offset of the thread's (clamped) row p;  k-strided: byte offset of its (clamped) column quad
  int lead;                        // KC: the thread's first k inside a tile (4 kq);  k-strided: its k-row inside a tile
  __device__ __forceinline__ void init(const float* src, int ld_, int r0, int n_, int K_, int tid) {
    base = src; ld = ld_; K = K_;
    if (KC) {
      lead = 4 * (tid % QK);
#pragma unroll
      for (int p = 0; p < P; ++p) off[p] = (unsigned)gm_opaque(min(r0 + tid / QK + RPP * p, n_ - 1)) * (unsigned)ld_ * 4u;
    } else {
      lead = tid / QR;
      off[0] = (unsigned)gm_opaque(min(r0 + 4 * (tid % QR), n_ - 4)) * 4u;
    }
  }
  __device__ __forceinline__ void issue(int k0, f32x4 (&buf)[P]) const {
    if (KC) {
      const unsigned kc = (unsigned)min(k0 + lead, K - 4) * 4u;
#pragma unroll
      for (int p = 0; p < P; ++p) buf[p] = *reinterpret_cast<const GmF4u*>(reinterpret_cast<const char*>(base) + (off[p] + kc));
    } else {
#pragma unroll
      for (int p = 0; p < P; ++p) { const unsigned kr = (unsigned)min(k0 + lead + KP * p, K - 1);
        buf[p] = *reinterpret_cast<const GmF4u*>(reinterpret_cast<const char*>(base) + (kr * (unsigned)ld * 4u + off[0])); }
    }
  }
  // piece p of a staged k-tile -> LDS.  FIX: the k-tile crosses K.
  template <bool FIX>
  __device__ __forceinline__ void commit_piece(int p, float* __restrict__ lds, int k0, const f32x4& b, int tid) const {
    f32x4 v = b;
    if (KC) {
      const int kq = tid % QK, row = tid / QK + RPP * p;
      if (FIX) {           // element e stands for k + e: v[e + sh] when k + e < K (sh = how far the clamp moved the quad left), else 0
        const int k = k0 + lead, sh = k - min(k, K - 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { float t = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) t = (q == e + sh) ? b[q] : t;
          v[e] = k + e < K ? t : 0.f; }
      }
      *reinterpret_cast<f32x4*>(lds + row * BK + 4 * (kq ^ ((row >> SWS) & (QK - 1)))) = v;
    } else {
      const int rq = tid % QR, kk = lead + KP * p;
      if (FIX) { const bool in = k0 + kk < K; v[0] = in ? b[0] : 0.f; v[1] = in ? b[1] : 0.f; v[2] = in ? b[2] : 0.f; v[3] = in ? b[3] : 0.f; }
      *reinterpret_cast<f32x4*>(lds + kk * (R + 8) + 4 * rq) = v;
    }
  }
};

// One fragment register set: the A fragments of TM row tiles and the B fragments of TN column tiles for one k-block (8 inputs)
template <int TM, int TN, bool BKC>
struct G3Frags {
  f32x4 a[TM];
  f32x4 b[BKC ? TN : 1];            // BKC: per column tile, element e = k-step e
  f32x2 b2[(!BKC && TN == 2) ? 4 : 1];      // !BKC, TN = 2: per k-step e, element = column tile
  float b1[(!BKC && TN == 1) ? 4 : 1];
  static constexpr int NREADS = TM + (BKC ? TN : 4);
  __device__ __forceinline__ void opaque() {
#pragma unroll
    for (int t = 0; t < TM; ++t) g3_opaque(a[t]);
    if constexpr (BKC) {
#pragma unroll
      for (int t = 0; t < TN; ++t) g3_opaque(b[t]);
    } else if constexpr (TN == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) g3_opaque(b2[e]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) g3_opaque(b1[e]);
    }
  }
  template <int E> __device__ __forceinline__ float bval(int t) const {
    if constexpr (BKC) return b[t][E]; else if constexpr (TN == 2) return b2[E][t]; else return b1[E];
  }
};

// waves per SIMD the register allocation is held to (= workgroups per CU): 2 for the 128 x 128 tile (at 3 = 168 registers the compiler spills, and a spill of a
// fragment register between its inline-asm ds_read and the wait stores a value that has not landed: measured wrong results), 3 for the 64 x 128 tile, 4 for 64 x 64
// (round 5: the k-strided B operand's 64 x 128 tile — the fp32 input gradient, GO2_GEMM_SPLIT=0 — spilled 8 - 28 B per lane at 168 registers: held to 2 waves instead)
#ifndef GM3_WAVES
#define GM3_WAVES(TM, TN, BK, BKC) ((TM) * (TN) == 4 ? 2 : (TM) * (TN) == 2 ? ((BKC) ? 3 : 2) : 4)
#endif
template <int TM, int TN, bool BKC, int EPI, int BK>
__global__ void __launch_bounds__(256, GM3_WAVES(TM, TN, BK, BKC)) go2nn_gemm3_kernel(const Gemm3Args ga) {
  static_assert(BKC || TN <= 2, "the natural-orientation B fragments are 4- or 8-byte reads");
  constexpr int NTHR = 256, BM = 64 * TM, BN = 64 * TN, NKB = BK / 8;
  static_assert(NKB % 2 == 0, "the two fragment sets alternate per k-block");
  using SA = G3Stage<BM, true, BK>; using SB = G3Stage<BN, BKC, BK>;
  using FR = G3Frags<TM, TN, BKC>;
  constexpr int ASZ = SA::LDS_FLOATS, BSZ = SB::LDS_FLOATS, STAGE = ASZ + BSZ, LOOP_LDS = 2 * STAGE, EPI_LDS = 4 * 32 * 32 * TN;
  constexpr int BP = BN + 8;        // row pitch of the k-strided stage (GmStage)
  __shared__ __attribute__((aligned(16))) float lds[LOOP_LDS > EPI_LDS ? LOOP_LDS : EPI_LDS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, gk = lane >> 5, wm = wave & 1, wn = wave >> 1;
  // workgroup -> tile of the group's list (problem 0's tiles, then problem 1's); consecutive workgroup ids go round the 8 XCDs, a XCD gets a contiguous run
  // (each problem's tiles are dealt to the XCDs separately: with one run over the concatenated list the XCDs 0-3 would get the actor's short tiles and 4-7 the critic's)
  int t = blockIdx.x, pi;
  const int n0 = ga.ntiles0, n1 = ga.ntiles - ga.ntiles0;
  if (((n0 | n1) & 7) == 0) { const int x = t & 7, j = t >> 3, c0 = n0 >> 3, c1 = n1 >> 3; pi = j >= c0 ? 1 : 0; t = pi ? x * c1 + (j - c0) : x * c0 + j; }
  else { pi = t >= n0 ? 1 : 0; t -= pi ? n0 : 0; }
  const Gemm3Prob& g = ga.p[pi];
  const int bm = t / g.nbn, bn = t - bm * g.nbn;
  const int row0 = bm * BM, col0 = bn * BN;
  const int nk = (g.K + BK - 1) / BK;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // fragment addresses (LDS byte offsets; stage, row-tile and k-step offsets are instruction immediates)
  const unsigned lbase = (unsigned)(uintptr_t)lds;
  constexpr int SWS = SA::SWS, QK = SA::QK;
  unsigned fq[NKB];      // the swizzled quad of k-block kb for this lane: (2 kb + g) ^ ((row >> SWS) & (QK - 1)); row = 32 x + i
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) fq[kb] = (unsigned)(((2 * kb + gk) ^ ((i >> SWS) & (QK - 1))) * 16);
  const unsigned a_row = lbase + (unsigned)((wm * 32 * TM + i) * BK * 4);
  const unsigned b_row = BKC ? lbase + (unsigned)(ASZ * 4 + (wn * 32 * TN + i) * BK * 4)
                             : lbase + (unsigned)(ASZ * 4 + (4 * gk * BP + wn * 32 * TN + TN * i) * 4);

  auto read_frags = [&](auto stage_c, auto kb_c, FR& f) __attribute__((always_inline)) {
    constexpr int S = decltype(stage_c)::value, KB = decltype(kb_c)::value, SOFF = S * STAGE * 4;
    g3_for<0, TM>([&](auto a_c) __attribute__((always_inline)) { constexpr int A = decltype(a_c)::value; g3_dsr128<SOFF + A * 32 * BK * 4>(f.a[A], a_row + fq[KB]); });
    if constexpr (BKC) {
      g3_for<0, TN>([&](auto b_c) __attribute__((always_inline)) { constexpr int B = decltype(b_c)::value; g3_dsr128<SOFF + B * 32 * BK * 4>(f.b[B], b_row + fq[KB]); });
    } else if constexpr (TN == 2) {
      g3_for<0, 4>([&](auto e_c) __attribute__((always_inline)) { constexpr int E = decltype(e_c)::value; g3_dsr64<SOFF + (8 * KB + E) * BP * 4>(f.b2[E], b_row); });
    } else {
      g3_for<0, 4>([&](auto e_c) __attribute__((always_inline)) { constexpr int E = decltype(e_c)::value; g3_dsr32<SOFF + (8 * KB + E) * BP * 4>(f.b1[E], b_row); });
    }
  };
  auto mfmas = [&](auto e0_c, auto e1_c, const FR& f) __attribute__((always_inline)) {
    g3_for<decltype(e0_c)::value, decltype(e1_c)::value>([&](auto e_c) __attribute__((always_inline)) {
      constexpr int E = decltype(e_c)::value;
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[a][E], f.template bval<E>(b), acc[a][b], 0, 0, 0);
    });
  };

#ifdef GM3_STAMPS
  long long st_[6] = {0, 0, 0, 0, 0, 0};
#define G3_T(k) do { __builtin_amdgcn_sched_barrier(0); st_[k] = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define G3_T(k) do { } while (0)
#endif
  G3_T(0);
  // two staging register sets: the loads of k-tile kt + 2 are issued at the top of tile kt and written to LDS during tile kt + 1
  SA sa; SB sb;
  sa.init(g.A, g.lda, row0, g.M, g.K, tid); sb.init(g.B, g.ldb, col0, g.N, g.K, tid);
  f32x4 ba0[SA::P], ba1[SA::P], bb0[SB::P], bb1[SB::P];
  sa.issue(0, ba0); sb.issue(0, bb0);
  if (nk > 1) { sa.issue(BK, ba1); sb.issue(BK, bb1); }
  constexpr int PA = SA::P, NP = SA::P + SB::P, NCB = NKB - 1;       // staged pieces of a k-tile (A's, then B's), dealt to the k-blocks 0 .. NKB - 2
  // (piece indices are compile-time constants throughout: a runtime index would put the staging arrays into scratch)
  auto commit = [&](auto fix_c, auto p_c, float* st, int k0, const f32x4 (&a)[SA::P], const f32x4 (&b)[SB::P]) {
    constexpr bool FIX = decltype(fix_c)::value; constexpr int PP = decltype(p_c)::value;
    if constexpr (PP < PA) sa.template commit_piece<FIX>(PP, st, k0, a[PP], tid); else sb.template commit_piece<FIX>(PP - PA, st + ASZ, k0, b[PP - PA], tid);
  };
  g3_for<0, NP>([&](auto p_c) __attribute__((always_inline)) { commit(G3Int<1>{}, p_c, lds, 0, ba0, bb0); });
  __syncthreads();
  FR f0, f1;
  read_frags(G3Int<0>{}, G3Int<0>{}, f0);
  G3_T(1);

  // One k-tile, held in stage CUR (= kt & 1; its staging set is CUR as well).
  //   top        the loads of tile kt + 2 go into set CUR — unconditionally: past the last tile they re-read it (L2 hits, never used), so every tile of the
  //              loop is the same straight-line code and the compiler's vmcnt counts stay exact
  //   k-block kb the fragments of block kb + 1 are requested, ONE counted wait for this block's fragments, the block's MFMAs; behind them (in the shadow of
  //              the last MFMA) this block's share of tile kt + 1's staged pieces goes to LDS stage CUR ^ 1 (free since the barrier of tile kt - 1) —
  //              a block's LDS writes sit between its fragment reads and the next block's, so the next wait allows them: lgkmcnt(reads + writes)
  //   last block half of the MFMAs, the barrier (all writes of tile kt + 1 are in), the first fragments of tile kt + 1, the other half of the MFMAs
  // MORE: tile kt + 1 follows.  FIX: tile kt + 1 may cross K (its displaced pieces are put right on the way to LDS).
  auto tile = [&](auto cur_c, auto more_c, auto fix_c, int kt) __attribute__((always_inline)) {
    constexpr int CUR = decltype(cur_c)::value; constexpr bool MORE = decltype(more_c)::value;
    auto& a_ld = CUR ? ba1 : ba0; auto& b_ld = CUR ? bb1 : bb0; auto& a_st = CUR ? ba0 : ba1; auto& b_st = CUR ? bb0 : bb1;
    if constexpr (MORE) { const int k2 = min(kt + 2, nk - 1) * BK; sa.issue(k2, a_ld); sb.issue(k2, b_ld); }
    float* nxt = lds + (CUR ^ 1) * STAGE;
    g3_for<0, NKB>([&](auto kb_c) __attribute__((always_inline)) {
      constexpr int KB = decltype(kb_c)::value;
      // pieces written behind block KB: p = KB, KB + NCB, ... (none behind the last block)
      constexpr int WPREV = (MORE && KB > 0) ? (NP - (KB - 1) + NCB - 1) / NCB : 0;
      FR& fc = (KB & 1) ? f1 : f0; FR& fn = (KB & 1) ? f0 : f1;
      if constexpr (KB + 1 < NKB) { read_frags(G3Int<CUR>{}, G3Int<KB + 1>{}, fn); g3_wait_lgkm<FR::NREADS + WPREV>(); }
      else g3_wait_lgkm<WPREV>();
      fc.opaque();
      if constexpr (KB + 1 == NKB && MORE) {
        mfmas(G3Int<0>{}, G3Int<2>{}, fc);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        read_frags(G3Int<CUR ^ 1>{}, G3Int<0>{}, f0);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(G3Int<2>{}, G3Int<4>{}, fc);
      } else {
        mfmas(G3Int<0>{}, G3Int<4>{}, fc);
        if constexpr (MORE && KB < NCB) {
          g3_for<0, NP>([&](auto p_c) __attribute__((always_inline)) { if constexpr (decltype(p_c)::value % NCB == KB) commit(fix_c, p_c, nxt, (kt + 1) * BK, a_st, b_st); });
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  constexpr G3Int<0> I0{}; constexpr G3Int<1> I1{};
  // only the tile that crosses K is fixed on its way to LDS, and it is always committed in the tail below (the loop's tiles commit tiles <= nk - 2)
  // the epilogue's operands (input gradient: the ELU outputs Y of the wave's tile; forward: the bias) are requested in front of the LAST k-tile — the staging
  // registers are free by then — instead of at the head of the epilogue, where their latency was exposed once per workgroup (profiles/r4_gemm3.txt)
  constexpr int CT = 32 * TN, LPR = CT / 4, RPI = 64 / LPR, NI = 32 / RPI;      // epilogue geometry: lanes per row, rows per instruction, instructions per 32-row slab
  const int lc = (lane % LPR) * 4, lr = lane / LPR;
  const int col = col0 + wn * CT + lc;
  const bool cv = g.c_vec != 0 && col + 3 < g.N;
  float4 y4[EPI == EPI_DELU_COLSUM ? TM : 1][EPI == EPI_DELU_COLSUM ? NI : 1];
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
#ifndef GM3_EARLY_DIV
#define GM3_EARLY_DIV 4
#endif
  constexpr int NI_EARLY = GM3_EARLY_DIV ? NI / GM3_EARLY_DIV : 0;          // (all NI rows early costs 32 more registers beside two staging sets: spills at 3 waves per SIMD)
  auto epi_prefetch = [&](auto n0_c, auto n1_c) __attribute__((always_inline)) {
    constexpr int N0 = decltype(n0_c)::value, N1 = decltype(n1_c)::value;
    if ((EPI == EPI_BIAS_ELU || EPI == EPI_BIAS) && N0 == 0) {
      const int c1 = min(col, g.N - 1), c2 = min(col + 1, g.N - 1), c3 = min(col + 2, g.N - 1), c4 = min(col + 3, g.N - 1);
      bias4 = make_float4(g.bias[c1], g.bias[c2], g.bias[c3], g.bias[c4]);
    }
    if (EPI == EPI_DELU_COLSUM) {
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int n = N0; n < N1; ++n) {
          const float* q = g.Y + (size_t)min(row0 + wm * 32 * TM + a * 32 + lr + n * RPI, g.M - 1) * g.ldc;
          if (cv) y4[a][n] = *reinterpret_cast<const float4*>(q + col);
          else y4[a][n] = make_float4(q[min(col, g.N - 1)], q[min(col + 1, g.N - 1)], q[min(col + 2, g.N - 1)], q[min(col + 3, g.N - 1)]);
        }
    }
  };
  {
    int kt = 0;
    for (; kt + 3 < nk; kt += 2) { tile(I0, I1, I0, kt); tile(I1, I1, I0, kt + 1); }
    if (nk - kt == 3) { tile(I0, I1, I0, kt); tile(I1, I1, I1, kt + 1); epi_prefetch(G3Int<0>{}, G3Int<NI_EARLY>{}); tile(I0, I0, I0, kt + 2); }
    else if (nk - kt == 2) { tile(I0, I1, I1, kt); epi_prefetch(G3Int<0>{}, G3Int<NI_EARLY>{}); tile(I1, I0, I0, kt + 1); }
    else { epi_prefetch(G3Int<0>{}, G3Int<NI_EARLY>{}); tile(I0, I0, I0, kt); }
  }
  G3_T(2);
  epi_prefetch(G3Int<NI_EARLY>{}, G3Int<NI>{});
  __syncthreads();          // every wave is done with the stages: the epilogue turns tiles through the same LDS
  G3_T(3);

  // epilogue: as go2nn_gemm_kernel's (every wave turns its tile, 32 rows at a time, through its LDS quarter; 16-byte row accesses).  Column of
  // accumulator (b, lane i): BKC b * 32 + i; !BKC TN * i + b (the natural-orientation fragments interleave the column tiles)
  {
    float* wl = lds + wave * (32 * CT);
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const int rbase = row0 + wm * 32 * TM + a * 32 + lr;
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * gk;
          const int cc = BKC ? b * 32 + i : TN * i + b;
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
          if (row < g.M) { cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w; }
        }
        if (row < g.M) {
          float* o = g.C + (size_t)row * g.ldc + col;
          if (cv) *reinterpret_cast<float4*>(o) = v;
          else { if (col < g.N) o[0] = v.x; if (col + 1 < g.N) o[1] = v.y; if (col + 2 < g.N) o[2] = v.z; if (col + 3 < g.N) o[3] = v.w; }
        }
      }
    }
    if (EPI == EPI_DELU_COLSUM) {
#pragma unroll
      for (int d = LPR; d < 64; d <<= 1) { cs.x += __shfl_xor(cs.x, d); cs.y += __shfl_xor(cs.y, d); cs.z += __shfl_xor(cs.z, d); cs.w += __shfl_xor(cs.w, d); }
      __syncthreads();
      float* shs = lds;                     // [2][BN]
      if (lane < LPR) *reinterpret_cast<float4*>(shs + wm * BN + wn * CT + lc) = cs;
      __syncthreads();
      if (tid < BN && col0 + tid < g.N) g.part[(size_t)bm * g.N + col0 + tid] = shs[tid] + shs[BN + tid];
    }
  }
#ifdef GM3_STAMPS
  G3_T(4);
  if (ga.stamps && (tid & 63) == 0) { long long* o = ga.stamps + ((size_t)blockIdx.x * 4 + wave) * 8; o[0] = st_[0]; o[1] = st_[1]; o[2] = st_[2]; o[3] = st_[3]; o[4] = st_[4]; o[5] = wall_clock64(); }
#endif
}

// ---- weight gradient: operands straight from global memory into MFMA registers ------------------------------------------------------------
// TA / TN: 32-column tiles of G / of X one load of a lane feeds (tile b = columns c0 + T i + b): 2 x 4 for the square layers, 2 x 2 for ragged widths
// (4 x 2 was measured for the input layer and rejected: go2nn_impl.cpp, wgrad3_group_shape).
// VECA / VECB: the operand's rows are aligned to the lane's piece (one load instruction per piece); else 4-byte loads (a 45-float row pitch).
template <int TA, int TN, bool VECA, bool VECB, int D>
__global__ void __launch_bounds__(256, 2) go2nn_wgrad_kernel(const WgArgs wa) {
  static_assert((TA == 2 || TA == 4) && (TN == 2 || TN == 4) && TA * TN <= 8, "8- or 16-byte operand pieces, at most 8 accumulator tiles");
  typedef float AV __attribute__((ext_vector_type(TA)));
  typedef float AVu __attribute__((ext_vector_type(TA), aligned(4)));
  typedef float BV __attribute__((ext_vector_type(TN)));
  typedef float BVu __attribute__((ext_vector_type(TN), aligned(4)));
  __shared__ __attribute__((aligned(16))) BV red[4][16][64];          // one row tile (a) of the four waves' accumulators at a time
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, gk = lane >> 5;
  // workgroup -> (row slice, tile): the slices with index = x mod 8 live on XCD x (workgroup id % 8), all tiles of a slice next to each other
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int slice = (idx / wa.tiles) * 8 + xcd;
  int t = idx % wa.tiles;
  const int pi = t >= wa.tiles0 ? 1 : 0;
  t -= pi ? wa.tiles0 : 0;
  const WgProb& g = wa.p[pi];
  const int tc = t / g.ntk, tk = t - tc * g.ntk;
  const int c0 = tc * 32 * TA, k0 = tk * 32 * TN;
  const int rpw = wa.rows_per_slice >> 2;                      // rows per wave (a multiple of 2)
  const int m0 = slice * wa.rows_per_slice + wave * rpw, mend = min(wa.M, m0 + rpw);
  const int nsteps = (rpw / 2 + D - 1) / D * D;                // k-steps (2 rows each), padded to the ring depth with masked steps

  // the lane's column pieces, clamped into the row (out-of-range pieces are zeroed at use)
  const int ca = c0 + TA * i, cb = k0 + TN * i;
  bool a_ok[TA], b_ok[TN];
#pragma unroll
  for (int e = 0; e < TA; ++e) a_ok[e] = ca + e < g.C;
#pragma unroll
  for (int e = 0; e < TN; ++e) b_ok[e] = cb + e < g.Kin;
  const int cac = gm_opaque(min(ca, g.C - TA < 0 ? 0 : g.C - TA)), cbc = gm_opaque(min(cb, g.Kin - TN < 0 ? 0 : g.Kin - TN));
  // (a clamped piece starts further left: its elements are re-aligned by the shifts below; only edge tiles take that path)
  const bool a_edge = c0 + 32 * TA > g.C, b_edge = k0 + 32 * TN > g.Kin;
  const int a_sh = ca - cac, b_sh = cb - cbc;                  // 0 inside the matrix

  f32x16 acc[TA][TN];
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  AV ra[D]; BV rb[D];
  auto issue = [&](int s, int step) __attribute__((always_inline)) {
    const int m = gm_opaque(min(m0 + 2 * step + gk, wa.M - 1));
    const float* pa = g.G + (size_t)m * g.C + cac; const float* pb = g.X + (size_t)m * g.Kin + cbc;
    if (VECA) ra[s] = *reinterpret_cast<const AV*>(pa); else ra[s] = *reinterpret_cast<const AVu*>(pa);
    if (VECB) rb[s] = *reinterpret_cast<const BV*>(pb); else rb[s] = *reinterpret_cast<const BVu*>(pb);
  };
#pragma unroll
  for (int s = 0; s < D; ++s) issue(s, s);
  for (int j = 0; j < nsteps; j += D) {
#pragma unroll
    for (int s = 0; s < D; ++s) {
      AV av = ra[s]; BV bv = rb[s];
      const bool live = m0 + 2 * (j + s) + gk < mend;
      issue(s, j + s + D);
      if (a_edge) {                // (workgroup-uniform) shifted + masked pieces of an edge tile
        AV w;
#pragma unroll
        for (int e = 0; e < TA; ++e) {
          float v = 0.f;
#pragma unroll
          for (int q = 0; q < TA; ++q) v = (q == e + a_sh) ? av[q] : v;
          w[e] = a_ok[e] ? v : 0.f;
        }
        av = w;
      }
      if (b_edge) {
        BV w;
#pragma unroll
        for (int e = 0; e < TN; ++e) {
          float v = 0.f;
#pragma unroll
          for (int q = 0; q < TN; ++q) v = (q == e + b_sh) ? bv[q] : v;
          w[e] = b_ok[e] ? v : 0.f;
        }
        bv = w;
      }
#pragma unroll
      for (int e = 0; e < TA; ++e) av[e] = live ? av[e] : 0.f;
#pragma unroll
      for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
  }
  // the four waves' partial tiles, summed in wave order, one row tile at a time; accumulator (a, b, r) of lane (j = lane & 31, g) is
  // dW[c0 + TA ((r & 3) + 8 (r >> 2) + 4 g) + a][k0 + TN j + b]: the TN values of one (a, r) are TN consecutive floats of one output row
  float* __restrict__ out = g.part + (size_t)slice * g.C * g.Kin;
  const bool k_vec = (g.Kin % TN == 0) && ((reinterpret_cast<uintptr_t>(g.part) & (4 * TN - 1)) == 0) && (((size_t)g.C * g.Kin) % TN == 0);
#pragma unroll
  for (int a = 0; a < TA; ++a) {
    if (a) __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      BV v;
#pragma unroll
      for (int b = 0; b < TN; ++b) v[b] = acc[a][b][r];
      red[wave][r][lane] = v;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = 4 * wave + q;
      BV v = red[0][r][lane];
#pragma unroll
      for (int w = 1; w < 4; ++w) { const BV u = red[w][r][lane];
#pragma unroll
        for (int b = 0; b < TN; ++b) v[b] += u[b]; }
      const int c = c0 + TA * ((r & 3) + 8 * (r >> 2) + 4 * gk) + a, k = k0 + TN * i;
      if (c < g.C) {
        float* o = out + (size_t)c * g.Kin + k;
        if (k_vec && k + TN <= g.Kin) *reinterpret_cast<BV*>(o) = v;
        else {
#pragma unroll
          for (int b = 0; b < TN; ++b) if (k + b < g.Kin) o[b] = v[b];
        }
      }
    }
  }
}
#endif  // !GO2_EMU

// tile shape of the grouped forward / input-gradient GEMMs for an output of N columns (rows: the mini-batch): 64 x 128 workgroup tiles with 32-deep k-tiles
// (48 KB of LDS, 3 workgroups per CU, no spills: measured best at every shape of the update against 128 x 128 and 16-deep tiles, profiles/r4_gemm3_bench.txt — those
// instantiations are not shipped), 64 x 64 for narrow outputs
static inline void gemm3_tile(int N, int* tm, int* tn, int* bk) {
  *tn = N > 64 ? 2 : 1;
  *tm = 1;
  *bk = 32;
}
// row slices of the grouped weight gradient: ~512 workgroups over all tiles of the group (~256 when the group has 8 tiles or fewer: the 256 -> 128 layer's kernel takes
// the same 33 us with 32 slices as with 64, and the slices' partial tiles are what go2nn_sum_rows has to read — 41 -> 38 us with the sum), a multiple of 8 slices
// (one XCD per slice class), at least 64 rows per slice
static inline void wgrad3_shape(int M, int tiles, int* nsplit, int* rows_per_slice) {
  static const int env_target = getenv("GO2NN_WG3_WGS") ? atoi(getenv("GO2NN_WG3_WGS")) : 0;      // tools only (read once)
  const int target = env_target ? env_target : (tiles <= 8 ? 256 : 512);
  int s = (target / (tiles > 0 ? tiles : 1) + 7) / 8 * 8;
  if (s < 8) s = 8;
  while (s > 8 && (M + s - 1) / s < 64) s -= 8;
  int rows = ((M + s - 1) / s + 7) / 8 * 8;
  *nsplit = s; *rows_per_slice = rows;
}
static inline int wgrad3_tn(int Kin) { return Kin > 64 ? 4 : 2; }
