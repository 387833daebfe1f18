// go2nn_impl.cpp — the C ABI of include/go2nn.h: the rollout's policy evaluation (PPO.act, rsl_rl/rsl_rl/algorithms/ppo.py:90-102) as ONE
// MFMA kernel.  This file holds the fp32-MFMA form (go2nn_mlp_kernel: GO2_GEMM_SPLIT=0, and networks too wide for the other one's LDS layout); the default since round 5 is
// go2nn_mlp3_kernel (go2nn_mlp3.h: the same job on the bf16 matrix pipe with exactly split fp32 operands), chosen in run() below.
//
// Built two ways, like go2sim_impl.cpp:
//   hipcc --offload-arch=gfx950  -> libgo2nn_hip.so : the product
//   g++ -DGO2_EMU                -> libgo2nn_emu.so : TEST-ONLY host build; the same packed operand buffer read in the same order by plain
//                                                     loops (checks the packing / padding / index arithmetic and the host side without a GPU)
//
// Kernel: grid = (ceil(N / 32), networks), 256 threads = 4 waves.  A workgroup carries 32 rows through every layer of ONE network:
//   * activations: two [32][516] fp32 tiles in LDS (ping-pong; row pitch 516 floats = 4 banks off a multiple of 64: the per-lane 16-byte
//     A-operand reads of 16 consecutive rows hit 64 different banks)
//   * weights: streamed from L2 in the operand order of v_mfma_f32_32x32x2_f32, pre-packed by go2nn_pack: for output tile t (32 columns)
//     and k-block kb (8 inputs) one 1-KiB wave load = lane (j = l & 31, g = l >> 5) -> W[32 t + j][8 kb + 4 g .. + 3]; the four elements
//     feed four MFMAs whose k index (0 / 1 = the lane's g) then means input 8 kb + 4 g + e — the A operand is read from LDS with the
//     same permutation, one ds_read_b128 per k-block shared by all the wave's tiles
//   * a wave owns the output tiles wave, wave + 4, ... of a layer (4 accumulator tiles = 64 registers at width 512); bias + ELU in the
//     epilogue, written to the other LDS tile as the next layer's input
//   * the last layer's tile is the head: actor -> mu; a = mu + std * eps, log-probability, storage rows; critic -> value
// No vendor GEMM library; fp32 in, fp32 accumulate (bit-wise a k-ordered fmaf chain per output, MI355X guide).
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#include "../../include/go2nn.h"

#ifndef GO2_EMU
#include <hip/hip_runtime.h>
#endif

static thread_local char g_err[256] = "";
#define FAIL(code, ...) do { snprintf(g_err, sizeof(g_err), __VA_ARGS__); return (code); } while (0)

#define NN_ROWS 32              // rows per workgroup (= the M of the MFMA tile)
#define NN_LD (GO2NN_MAX_WIDTH + 4)
#ifndef NN_THREADS
#define NN_THREADS 512         // 8 waves = two per SIMD: one wave's load / LDS waits are filled with the other's MFMAs (measured: 256 -> 65 us, 512 -> see profiles/r3_policy_kernel.txt)
#endif
#define NN_WAVES (NN_THREADS / 64)
#define NN3_LDS 163840          // the split-operand kernel's LDS: the whole 160 KB of a CU
// row pitch in bytes of a bf16 plane of `width` (a multiple of 16) columns: 16 bytes off a multiple of 64 — the 16-byte fragment reads of 16 consecutive rows
// fall into 16 different 16-byte bank groups
#define nn3_pitch(width) (((width) + 31) / 32 * 64 + 16)

// what the kernel needs of one network: padded shapes and the offsets of its layers in the packed buffer
struct NetDesc {
  int32_t nl, in_dim, out_dim;
  int32_t K[GO2NN_MAX_LAYERS], N[GO2NN_MAX_LAYERS];      // true input / output width of layer l
  int32_t KB[GO2NN_MAX_LAYERS], NT[GO2NN_MAX_LAYERS];    // k-blocks of 8, output tiles of 32
  int64_t woff[GO2NN_MAX_LAYERS], boff[GO2NN_MAX_LAYERS];
  // the split-operand image of the same weights (go2nn_mlp3_kernel): 16-input k-blocks, KV3 of them hold inputs, padded with zero blocks to KB3 (a multiple of the
  // weight ring's depth); lds3 = bytes of the kernel's first LDS region (the even activations), 0 = the network's activations do not fit the 160 KB as planes
  int32_t KB3[GO2NN_MAX_LAYERS], KV3[GO2NN_MAX_LAYERS], lds3;
  int64_t woff3[GO2NN_MAX_LAYERS];
  const float* packed; const float* x;
  // ABI 5 (the CTS rollout): the input row is two segments — columns [0, kx) from x (row pitch ldx), [kx, in_dim) from x2 (pitch ldx2) —, the network runs on
  // the rows `rows[0 .. nrows)` of its inputs (NULL: rows 0 .. nrows - 1), and mode 0 stores y[row * ldy + n], L2-normalised when `normalize`
  const float* x2; const int32_t* rows; float* y;
  int32_t ldx, ldx2, kx, nrows, ldy, normalize;
};
struct NNArgs {
  NetDesc net[2];
  const float *std_, *eps; float *a_out, *a_st, *mu_st, *sig_st, *lp_st, *v_st; float* y;
  int32_t N, A, mode;           // mode 0: forward of every net (grid.y) into its own y; mode 1: policy act (net[0] actor, net[1] critic); N = the largest nrows
};
// the plain case: one dense input matrix, rows 0 .. N - 1, dense output
static inline void net_io(NetDesc& d, const float* x, int N, float* y) {
  d.x = x; d.x2 = nullptr; d.rows = nullptr; d.y = y; d.ldx = d.in_dim; d.ldx2 = 0; d.kx = d.in_dim; d.nrows = N; d.ldy = d.out_dim; d.normalize = 0;
}

static int describe(const Go2nnMlp* m, NetDesc* d, int64_t* total) {
  if (!m || m->num_layers <= 0 || m->num_layers > GO2NN_MAX_LAYERS) return 0;
  int64_t off = 0;
  d->nl = m->num_layers; d->in_dim = m->dims[0]; d->out_dim = m->dims[m->num_layers];
  for (int l = 0; l <= m->num_layers; ++l) if (m->dims[l] <= 0 || m->dims[l] > GO2NN_MAX_WIDTH) return 0;
  for (int l = 0; l < m->num_layers; ++l) {
    d->K[l] = m->dims[l]; d->N[l] = m->dims[l + 1];
    d->KB[l] = (d->K[l] + 31) / 32 * 4; d->NT[l] = (d->N[l] + 31) / 32;      // k-blocks padded to a multiple of 4 = 32 inputs (zero weights): branch-free loops, and 32 NT[l-1] = 8 KB[l]
    d->woff[l] = off; off += (int64_t)d->NT[l] * d->KB[l] * 64 * 4;
    d->boff[l] = off; off += (int64_t)d->NT[l] * 32;
  }
  // (behind the fp32 image: GO2_GEMM_SPLIT=0 and the shapes whose planes do not fit the LDS run the fp32-MFMA kernel from the same buffer)
  int64_t even = 0, odd = 0;
  for (int l = 0; l < m->num_layers; ++l) {
    d->KV3[l] = (d->K[l] + 15) / 16;
    const int ring = d->NT[l] > NN_WAVES ? 4 : 8;          // run_layer3's ring slots
    d->KB3[l] = (d->KV3[l] + ring - 1) / ring * ring;
    d->woff3[l] = off; off += (int64_t)d->NT[l] * d->KB3[l] * 768;
    const int64_t bytes = 3LL * NN_ROWS * nn3_pitch(l == 0 ? 16 * d->KV3[0] : 32 * d->NT[l - 1]);
    ((l & 1) ? odd : even) = std::max((l & 1) ? odd : even, bytes);
  }
  { const int64_t bytes = (int64_t)NN_ROWS * (32 * d->NT[m->num_layers - 1] + 4) * 4;          // the last layer's output tile stays fp32 (the heads read it)
    ((m->num_layers & 1) ? odd : even) = std::max((m->num_layers & 1) ? odd : even, bytes); }
  d->lds3 = even + odd <= NN3_LDS ? (int32_t)even : 0;
  if (total) *total = off;
  return 1;
}

#define HALF_LOG2PI 0.9189385332046727f

#ifndef GO2_EMU
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ELU(alpha = 1): exp(v) - 1 through the hardware exponential (v_exp_f32, ~1 ulp of exp): absolute error <= 1.2e-7 — fp32 round-off of the
// activations' own scale; libm's expm1f costs ~40 instructions and the epilogue of a 512-wide layer evaluates it 64 times per lane
__device__ __forceinline__ float elu1(float v) { return v > 0.f ? v : __expf(v) - 1.0f; }
// a * b and a + b as two separately rounded operations (hipcc contracts `a + b * c`, and HIP's __fmul_rn is a plain product, into an FMA)
__device__ __forceinline__ float mul_rn(float a, float b) { float r; asm("v_mul_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float add_rn(float a, float b) { float r; asm("v_add_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// One layer for one wave: NTW output tiles (t = first + NN_WAVES j), weights prefetched D k-blocks ahead through a register ring, the A operand
// of the next k-block read from LDS while this one's MFMAs issue.  The loop body is branch-free (k-blocks are padded to a multiple of 4, tile
// and block indices are clamped instead of tested) so that the compiler can count outstanding loads exactly: a wave waits for the OLDEST
// wave load only, D NTW - NTW newer ones stay in flight (the one-tile layers have only 4 MFMAs = 256 cycles of work per k-block).
template <int NTW>
__device__ __forceinline__ void run_layer(const float* __restrict__ A, float* __restrict__ B, const float4* __restrict__ Wp, const float* __restrict__ bp,
                                          int KB, int NT, int first, int lane, bool last) {
  constexpr int D = NTW == 4 ? 2 : 4;
  const int i = lane & 31, g = lane >> 5;
  f32x16 acc[NTW];
  const float4* wt[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int t = first + NN_WAVES * j < NT ? first + NN_WAVES * j : NT - 1;      // (a tile past the layer's last is a clamped duplicate whose result is dropped)
    wt[j] = Wp + (int64_t)t * KB * 64 + lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  }
  float4 ring[D][NTW];
#pragma unroll
  for (int s = 0; s < D; ++s)
#pragma unroll
    for (int j = 0; j < NTW; ++j) ring[s][j] = wt[j][s * 64];
  const float* arow = &A[i * NN_LD + 4 * g];
  float4 a_next = *reinterpret_cast<const float4*>(arow);
  for (int kb0 = 0; kb0 < KB; kb0 += D) {
#pragma unroll
    for (int s = 0; s < D; ++s) {
      const int kb = kb0 + s;
      float4 bc[NTW];
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        // the block is consumed as ONE 16-byte value (empty asm on the register quad): left to itself the compiler splits the wave load
        // into four 4-byte loads, one per MFMA, which quarters the efficiency of the L2 -> CU path
        f32x4 q = {ring[s][j].x, ring[s][j].y, ring[s][j].z, ring[s][j].w};
        asm volatile("" : "+v"(q));
        bc[j] = make_float4(q[0], q[1], q[2], q[3]);
      }
      const int kn = kb + D < KB ? kb + D : KB - 1;
#pragma unroll
      for (int j = 0; j < NTW; ++j) ring[s][j] = wt[j][kn * 64];
      const float4 a4 = a_next;
      a_next = *reinterpret_cast<const float4*>(arow + 8 * (kb + 1 < KB ? kb + 1 : KB - 1));
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, bc[j].x, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, bc[j].y, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, bc[j].z, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, bc[j].w, acc[j], 0, 0, 0);
      }
    }
  }
  // epilogue: C/D layout of the 32x32 tile: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int j = 0; j < NTW; ++j) if (first + NN_WAVES * j < NT) {
    const int n = 32 * (first + NN_WAVES * j) + i; const float bias = bp[n];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * g;
      const float v = acc[j][r] + bias;
      B[row * NN_LD + n] = last ? v : elu1(v);
    }
  }
}

// the network's output tile A [32][ld] fp32 -> mode 0: the output rows (L2-normalised on request); mode 1: the sampling head (actor) / the value (critic)
__device__ __forceinline__ void nn_finish(const NNArgs& a, const NetDesc& nd, float* A, const int ld, const int row0, const int tid) {
  if (a.mode == 0) {
    const int No = nd.out_dim;
    if (nd.normalize) {          // F.normalize(x, p=2, dim=-1): x / max(|x|, 1e-12), the scale of row r parked behind its last column (No <= 512 < ld)
      if (tid < NN_ROWS) { float ss = 0.f; for (int n = 0; n < No; ++n) ss += A[tid * ld + n] * A[tid * ld + n]; A[tid * ld + No] = 1.f / fmaxf(sqrtf(ss), 1e-12f); }
      __syncthreads();
    }
    for (int idx = tid; idx < NN_ROWS * No; idx += NN_THREADS) {
      const int r = idx / No, n = idx - r * No;
      if (row0 + r < nd.nrows) {
        const int64_t dr = nd.rows ? nd.rows[row0 + r] : row0 + r;
        nd.y[dr * nd.ldy + n] = nd.normalize ? A[r * ld + n] * A[r * ld + No] : A[r * ld + n];
      }
    }
    return;
  }
  if (tid < NN_ROWS) {
    const int e = row0 + tid;
    if (e >= a.N) return;
    if (blockIdx.y == 0) {      // actor: the sampling head (ppo.py:90-102; go2sim_act_head's arithmetic: a = mu + std * eps as two rounded operations)
      float lp = 0.f;
      for (int j = 0; j < a.A; ++j) {
        const int64_t k = (int64_t)e * a.A + j;
        const float m = A[tid * ld + j], sg = a.std_[j], act = add_rn(m, mul_rn(sg, a.eps[k])), d = act - m;
        lp += -(d * d) / (2.f * sg * sg) - logf(sg) - HALF_LOG2PI;
        a.a_out[k] = act;
        if (a.a_st) a.a_st[k] = act;
        if (a.mu_st) a.mu_st[k] = m;
        if (a.sig_st) a.sig_st[k] = sg;
      }
      if (a.lp_st) a.lp_st[e] = lp;
    } else if (a.v_st) {
      a.v_st[e] = A[tid * ld];
    }
  }
}

__global__ void __launch_bounds__(NN_THREADS) go2nn_mlp_kernel(const NNArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[2 * NN_ROWS * NN_LD];
  const NetDesc& nd = a.net[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63, row0 = blockIdx.x * NN_ROWS;
  // which of a layer's tiles a wave takes is free (tiles are independent outputs): rotated by the workgroup index, so that the workgroups
  // of the chip, which run in step, do not all ask the L2 for the same weight block at the same moment
  const int wave = __builtin_amdgcn_readfirstlane(((tid >> 6) + (int)blockIdx.x) & (NN_WAVES - 1));
  float* A = lds; float* B = lds + NN_ROWS * NN_LD;
#ifdef GO2NN_STAMPS
  long long* dbg = (a.mode == 1 && a.y) ? (long long*)a.y + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 : nullptr;      // tools/policy_bench.py --stamps
#define NN_STAMP(k) do { if (dbg && tid == 0) dbg[k] = wall_clock64(); } while (0)
#else
#define NN_STAMP(k) do { } while (0)
#endif
  NN_STAMP(0);
  if (row0 >= nd.nrows) return;          // (the two networks of a launch may run on different numbers of rows: whole workgroups past a network's rows leave)
  {   // stage the workgroup's input rows, zero-padded to the first layer's k-blocks
    // (a wave takes its share of the rows, its lanes run along the row: coalesced, no division; all of a thread's loads are in flight together)
    const int Kp = nd.KB[0] * 8, K0 = nd.in_dim, kx = nd.kx, w = tid >> 6;
#pragma unroll
    for (int r = 0; r < NN_ROWS / NN_WAVES; ++r) {
      const int i = w * (NN_ROWS / NN_WAVES) + r; const bool live = row0 + i < nd.nrows;
      const int e = live ? row0 + i : nd.nrows - 1;
      const int64_t sr = nd.rows ? nd.rows[e] : e;
      const float* __restrict__ src = nd.x + sr * nd.ldx;
      const float* __restrict__ src2 = nd.x2 ? nd.x2 + sr * nd.ldx2 - kx : src;          // (indexed with k >= kx only)
      for (int k = lane; k < Kp; k += 64) A[i * NN_LD + k] = (live && k < K0) ? (k < kx ? src[k] : src2[k]) : 0.f;
    }
  }
  __syncthreads();
  NN_STAMP(1);
  for (int l = 0; l < nd.nl; ++l) {
    const int KB = nd.KB[l], NT = nd.NT[l];
    const float4* __restrict__ Wp = reinterpret_cast<const float4*>(nd.packed + nd.woff[l]);
    const float* __restrict__ bp = nd.packed + nd.boff[l];
    const bool last = l == nd.nl - 1;
    if (wave < NT) {
      if (NT > 2 * NN_WAVES) run_layer<4>(A, B, Wp, bp, KB, NT, wave, lane, last);
      else if (NT > NN_WAVES) run_layer<2>(A, B, Wp, bp, KB, NT, wave, lane, last);
      else run_layer<1>(A, B, Wp, bp, KB, NT, wave, lane, last);
    }
    __syncthreads();
    NN_STAMP(2 + l);
    float* t_ = A; A = B; B = t_;
  }
  nn_finish(a, nd, A, NN_LD, row0, tid);          // A now holds the network's output tile [32][32 NT_last]
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) FAIL(GO2NN_EDEVICE, "%s: %s", #x, hipGetErrorString(e_)); } while (0)
#endif  // !GO2_EMU

#include "go2nn_train.h"
#include "go2nn_gemm.h"
#include "go2nn_gemm3.h"
#include "go2nn_bx3.h"
#include "go2nn_mlp3.h"
#include "go2nn_cts.h"

#ifdef GO2_EMU
// host restatement: the SAME packed buffer, read in the operand order the kernel uses
static void emu_forward(const NetDesc& nd, int N, int row0, float out[NN_ROWS][GO2NN_MAX_WIDTH]) {
  static thread_local float A[NN_ROWS][NN_LD], B[NN_ROWS][NN_LD];
  memset(A, 0, sizeof(A));
  (void)N;
  for (int i = 0; i < NN_ROWS && row0 + i < nd.nrows; ++i) {
    const int64_t sr = nd.rows ? nd.rows[row0 + i] : row0 + i;
    for (int k = 0; k < nd.in_dim; ++k) A[i][k] = k < nd.kx ? nd.x[sr * nd.ldx + k] : nd.x2[sr * nd.ldx2 + (k - nd.kx)];
  }
  for (int l = 0; l < nd.nl; ++l) {
    const float* Wp = nd.packed + nd.woff[l]; const float* bp = nd.packed + nd.boff[l];
    memset(B, 0, sizeof(B));
    for (int t = 0; t < nd.NT[l]; ++t) for (int i = 0; i < NN_ROWS; ++i) for (int j = 0; j < 32; ++j) {
      float acc = 0.f;
      for (int kb = 0; kb < nd.KB[l]; ++kb) for (int e = 0; e < 4; ++e) for (int g = 0; g < 2; ++g)
        acc = fmaf(A[i][8 * kb + 4 * g + e], Wp[(((int64_t)t * nd.KB[l] + kb) * 64 + (g * 32 + j)) * 4 + e], acc);
      const float v = acc + bp[32 * t + j];
      B[i][32 * t + j] = l == nd.nl - 1 ? v : (v > 0.f ? v : expm1f(v));
    }
    memcpy(A, B, sizeof(A));
  }
  for (int i = 0; i < NN_ROWS; ++i) for (int n = 0; n < nd.out_dim; ++n) out[i][n] = A[i][n];
}
#endif

// ---- the hidden layers of the learner: MFMA GEMMs with fused epilogues (go2nn_gemm.h) ------------------------------------------------
#ifndef GO2_EMU
#ifdef GM_STAMPS
static long long* g_gemm_stamps = nullptr;
extern "C" void go2nn_debug_gemm_stamps(long long* p) { g_gemm_stamps = p; }      // TOOL-ONLY: device buffer [workgroups][8] of the next launches
#endif
// k-tile depth: 16 for the 128 x 128 tile (two stages = 32 KB of LDS: four workgroups per CU, and its k-tile then carries the same 32 MFMAs per wave as
// the 64 x 128 tile's 32-deep one), 32 otherwise.  GO2NN_BK overrides (tools/gemm_bench.py).
static inline int gemm_bk(int tm, int tn) {
  static const int env = getenv("GO2NN_BK") ? atoi(getenv("GO2NN_BK")) : 0;       // (read once)
  if (env == 16 || env == 32) return env;
  return tm >= 2 && tn == 2 ? 16 : 32;
}
template <bool AKC, bool BKC, int EPI, bool VEC>
static void gemm_dispatch2(int tm, int tn, const GemmArgs& g, int splits, hipStream_t st) {
  const dim3 grid(g.nbm * g.nbn, 1, splits), blk(GM_THREADS);
  const int bk = gemm_bk(tm, tn);
  if (tm == 2 && tn == 2 && bk == 16)      hipLaunchKernelGGL((go2nn_gemm_kernel<2, 2, AKC, BKC, EPI, VEC, 16>), grid, blk, 0, st, g);
  else if (tm == 2 && tn == 2)             hipLaunchKernelGGL((go2nn_gemm_kernel<2, 2, AKC, BKC, EPI, VEC, 32>), grid, blk, 0, st, g);
  else if (tm == 1 && tn == 2 && bk == 16) hipLaunchKernelGGL((go2nn_gemm_kernel<1, 2, AKC, BKC, EPI, VEC, 16>), grid, blk, 0, st, g);
  else if (tm == 1 && tn == 2)             hipLaunchKernelGGL((go2nn_gemm_kernel<1, 2, AKC, BKC, EPI, VEC, 32>), grid, blk, 0, st, g);
  else if (tm == 2 && tn == 1)             hipLaunchKernelGGL((go2nn_gemm_kernel<2, 1, AKC, BKC, EPI, VEC, 32>), grid, blk, 0, st, g);
  else                                     hipLaunchKernelGGL((go2nn_gemm_kernel<1, 1, AKC, BKC, EPI, VEC, 32>), grid, blk, 0, st, g);
}
template <bool AKC, bool BKC, int EPI>
static void gemm_dispatch(int tm, int tn, GemmArgs& g, int splits, bool vec, hipStream_t st) {
  static const int dbg = getenv("GO2NN_GEMM_DEBUG") ? atoi(getenv("GO2NN_GEMM_DEBUG")) : 0;      // tools/gemm_bench.py only (read once)
  g.debug = dbg;
#ifdef GM_STAMPS
  g.stamps = g_gemm_stamps;
#endif
  if (vec) gemm_dispatch2<AKC, BKC, EPI, true>(tm, tn, g, splits, st); else gemm_dispatch2<AKC, BKC, EPI, false>(tm, tn, g, splits, st);
}
static inline int aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }
#endif
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
// row splits of the weight gradient dW [C, Kin] = G^T X over M rows: enough workgroups for two per CU, at least 256 rows each
static inline void wgrad_shape(int M, int C, int Kin, int* tm, int* tn, int* splits, int* kchunk) {
  *tm = gm_pick(C); *tn = gm_pick(Kin);              // (no minimum workgroup count here: the row splits supply the parallelism)
  static const char* const wt = getenv("GO2NN_WTILE");                                             // tools/gemm_bench.py sweeps (read once)
  static const int wtarget = getenv("GO2NN_WSPLIT_WGS") ? atoi(getenv("GO2NN_WSPLIT_WGS")) : 512;
  if (wt && wt[0] >= '1' && wt[0] <= '2' && wt[1] >= '1' && wt[1] <= '2') { *tm = wt[0] - '0'; *tn = wt[1] - '0'; }
  const int target = wtarget;
  const int tiles = cdiv(C, 64 * *tm) * cdiv(Kin, 64 * *tn);
  int s = target / tiles; if (s < 1) s = 1;
  int kc = cdiv(cdiv(M, s), GM_BK) * GM_BK; if (kc < 256) kc = 256;
  *kchunk = kc; *splits = cdiv(M, kc);
}
static int lin_check(int M, int C, int Kin) { return M > 0 && C > 0 && Kin > 0 && C <= 4096 && Kin <= 4096; }

extern "C" {

int go2nn_abi_version(void) { return GO2NN_ABI_VERSION; }
const char* go2nn_last_error(void) { return g_err; }
int go2nn_is_device_library(void) {
#ifdef GO2_EMU
  return 0;
#else
  return 1;
#endif
}

int64_t go2nn_packed_floats(const Go2nnMlp* m) {
  NetDesc d; int64_t total = 0;
  if (!describe(m, &d, &total)) FAIL(GO2NN_EINVAL, "unsupported MLP shape (1..%d layers, widths 1..%d)", GO2NN_MAX_LAYERS, GO2NN_MAX_WIDTH);
  return total;
}

#ifndef GO2_EMU
static bool nn_fp32_mfma() { static const bool v = getenv("GO2_GEMM_SPLIT") && atoi(getenv("GO2_GEMM_SPLIT")) == 0; return v; }          // the learner's switch: one arithmetic per run
#endif
int32_t go2nn_mlp_arith(const Go2nnMlp* m) {
  NetDesc d;
  if (!describe(m, &d, nullptr)) FAIL(GO2NN_EINVAL, "unsupported MLP shape (1..%d layers, widths 1..%d)", GO2NN_MAX_LAYERS, GO2NN_MAX_WIDTH);
#ifdef GO2_EMU
  return 0;
#else
  return (!nn_fp32_mfma() && d.lds3 > 0) ? 3 : 1;
#endif
}

int go2nn_pack(const Go2nnMlp* m, float* packed, void* stream) {
  NetDesc d;
  if (!packed || !describe(m, &d, nullptr)) FAIL(GO2NN_EINVAL, "bad argument");
  for (int l = 0; l < d.nl; ++l) {
    if (!m->weight[l] || !m->bias[l]) FAIL(GO2NN_EINVAL, "null weight / bias of layer %d", l);
#ifdef GO2_EMU
    (void)stream;
    float* ow = packed + d.woff[l]; float* ob = packed + d.boff[l];
    for (int t = 0; t < d.NT[l]; ++t) for (int kb = 0; kb < d.KB[l]; ++kb) for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 4; ++e) {
      const int n = 32 * t + (lane & 31), k = 8 * kb + 4 * (lane >> 5) + e;
      ow[(((int64_t)t * d.KB[l] + kb) * 64 + lane) * 4 + e] = (n < d.N[l] && k < d.K[l]) ? m->weight[l][(int64_t)n * d.K[l] + k] : 0.f;
    }
    for (int n = 0; n < d.NT[l] * 32; ++n) ob[n] = n < d.N[l] ? m->bias[l][n] : 0.f;
#else
    const int64_t total = (int64_t)d.NT[l] * d.KB[l] * 256 + d.NT[l] * 32;
    int blocks = (int)((total + 255) / 256); blocks = blocks > 1024 ? 1024 : blocks;
    hipLaunchKernelGGL(go2nn_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, m->weight[l], m->bias[l], packed + d.woff[l], packed + d.boff[l], reinterpret_cast<u32x4*>(packed + d.woff3[l]),
                       d.K[l], d.N[l], d.KB[l], d.KB3[l], d.NT[l]);
#endif
  }
#ifndef GO2_EMU
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

static int run(NNArgs& a, int nets, void* stream) {
#ifdef GO2_EMU
  (void)stream;
  static thread_local float out[2][NN_ROWS][GO2NN_MAX_WIDTH];
  for (int row0 = 0; row0 < a.N; row0 += NN_ROWS) {
    for (int y = 0; y < nets; ++y) if (row0 < a.net[y].nrows) emu_forward(a.net[y], a.N, row0, out[y]);
    if (a.mode == 0) {
      for (int y = 0; y < nets; ++y) { const NetDesc& nd = a.net[y];
        for (int r = 0; r < NN_ROWS && row0 + r < nd.nrows; ++r) {
          float inv = 1.f;
          if (nd.normalize) { float ss = 0.f; for (int n = 0; n < nd.out_dim; ++n) ss += out[y][r][n] * out[y][r][n]; inv = 1.f / fmaxf(sqrtf(ss), 1e-12f); }
          const int64_t dr = nd.rows ? nd.rows[row0 + r] : row0 + r;
          for (int n = 0; n < nd.out_dim; ++n) nd.y[dr * nd.ldy + n] = nd.normalize ? out[y][r][n] * inv : out[y][r][n];
        } }
      continue;
    }
    for (int r = 0; r < NN_ROWS && row0 + r < a.N; ++r) {
      const int e = row0 + r;
      float lp = 0.f;
      for (int j = 0; j < a.A; ++j) {
        const int64_t k = (int64_t)e * a.A + j;
        const float m = out[0][r][j], sg = a.std_[j]; volatile float p = sg * a.eps[k]; const float act = m + p, d = act - m;
        lp += -(d * d) / (2.f * sg * sg) - logf(sg) - HALF_LOG2PI;
        a.a_out[k] = act; if (a.a_st) a.a_st[k] = act; if (a.mu_st) a.mu_st[k] = m; if (a.sig_st) a.sig_st[k] = sg;
      }
      if (a.lp_st) a.lp_st[e] = lp;
      if (a.v_st) a.v_st[e] = out[1][r][0];
    }
  }
#else
  // the split-operand kernel (go2nn_mlp3.h) unless GO2_GEMM_SPLIT=0 asks for fp32 MFMA everywhere (the learner's switch: one arithmetic per run) or a network's planes do not fit the LDS
  bool split = !nn_fp32_mfma();
  for (int y = 0; y < nets; ++y) split = split && a.net[y].lds3 > 0;
  const dim3 grid((a.N + NN_ROWS - 1) / NN_ROWS, nets), blk(NN_THREADS);
  if (split) hipLaunchKernelGGL(go2nn_mlp3_kernel, grid, blk, 0, (hipStream_t)stream, a);
  else       hipLaunchKernelGGL(go2nn_mlp_kernel, grid, blk, 0, (hipStream_t)stream, a);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

int go2nn_mlp_forward(const Go2nnMlp* m, const float* packed, const float* x, float* y, int32_t N, void* stream) {
  NNArgs a; memset(&a, 0, sizeof(a));
  if (!packed || !x || !y || N <= 0 || !describe(m, &a.net[0], nullptr)) FAIL(GO2NN_EINVAL, "bad argument");
  a.net[0].packed = packed; net_io(a.net[0], x, N, y); a.N = N; a.mode = 0;
  return run(a, 1, stream);
}

int go2nn_mlp_forward_rows(const Go2nnMlp* const* nets, const float* const* packed, const Go2nnMlpIO* io, int32_t nnets, void* stream) {
  NNArgs a; memset(&a, 0, sizeof(a));
  if (!nets || !packed || !io || nnets < 1 || nnets > 2) FAIL(GO2NN_EINVAL, "forward rows: 1 or 2 networks");
  for (int j = 0; j < nnets; ++j) {
    NetDesc& d = a.net[j]; const Go2nnMlpIO& q = io[j];
    if (!packed[j] || !describe(nets[j], &d, nullptr)) FAIL(GO2NN_EINVAL, "forward rows: bad network %d", j);
    const int k2 = d.in_dim - q.kx;
    if (!q.x || !q.y || q.nrows <= 0 || q.kx < 1 || q.kx > d.in_dim || q.ldx < q.kx || (k2 > 0 && (!q.x2 || q.ldx2 < k2)) || q.ldy < d.out_dim)
      FAIL(GO2NN_EINVAL, "forward rows: bad input / output description of network %d (segments %d + %d of %d inputs)", j, q.kx, k2, d.in_dim);
    d.packed = packed[j]; d.x = q.x; d.x2 = k2 > 0 ? q.x2 : nullptr; d.rows = q.rows; d.y = q.y; d.ldx = q.ldx; d.ldx2 = q.ldx2; d.kx = q.kx; d.nrows = q.nrows; d.ldy = q.ldy; d.normalize = q.normalize ? 1 : 0;
    a.N = std::max(a.N, q.nrows);
  }
  a.mode = 0;
  return run(a, nnets, stream);
}

int go2nn_policy_act(const Go2nnMlp* actor, const float* actor_packed, const Go2nnMlp* critic, const float* critic_packed,
                     const float* obs, const float* critic_obs, const float* std_, const float* eps,
                     float* a_out, float* a_st, float* mu_st, float* sig_st, float* lp_st, float* v_st, int32_t N, void* stream) {
  NNArgs a; memset(&a, 0, sizeof(a));
  if (!actor_packed || !critic_packed || !obs || !critic_obs || !std_ || !eps || !a_out || N <= 0 || !describe(actor, &a.net[0], nullptr) || !describe(critic, &a.net[1], nullptr))
    FAIL(GO2NN_EINVAL, "bad argument");
  if (a.net[0].out_dim > 32 || a.net[1].out_dim != 1) FAIL(GO2NN_EINVAL, "the head handles up to 32 actions and a scalar value (got %d, %d)", a.net[0].out_dim, a.net[1].out_dim);
  a.net[0].packed = actor_packed; net_io(a.net[0], obs, N, nullptr); a.net[1].packed = critic_packed; net_io(a.net[1], critic_obs, N, nullptr);
  a.std_ = std_; a.eps = eps; a.a_out = a_out; a.a_st = a_st; a.mu_st = mu_st; a.sig_st = sig_st; a.lp_st = lp_st; a.v_st = v_st;
  a.N = N; a.A = a.net[0].out_dim; a.mode = 1;
  return run(a, 2, stream);
}

int go2nn_policy_act_latent(const Go2nnMlp* actor, const float* actor_packed, const Go2nnMlp* critic, const float* critic_packed,
                            const float* latent, int32_t L, const float* obs, const float* critic_obs, const float* std_, const float* eps,
                            float* a_out, float* a_st, float* mu_st, float* sig_st, float* lp_st, float* v_st, int32_t N, void* stream) {
  NNArgs a; memset(&a, 0, sizeof(a));
  if (!actor_packed || !critic_packed || !latent || !obs || !critic_obs || !std_ || !eps || !a_out || N <= 0 || !describe(actor, &a.net[0], nullptr) || !describe(critic, &a.net[1], nullptr))
    FAIL(GO2NN_EINVAL, "bad argument");
  if (a.net[0].out_dim > 32 || a.net[1].out_dim != 1) FAIL(GO2NN_EINVAL, "the head handles up to 32 actions and a scalar value (got %d, %d)", a.net[0].out_dim, a.net[1].out_dim);
  if (L < 1 || L >= a.net[0].in_dim || L >= a.net[1].in_dim) FAIL(GO2NN_EINVAL, "latent width %d against input widths %d / %d", L, a.net[0].in_dim, a.net[1].in_dim);
  for (int j = 0; j < 2; ++j) {
    NetDesc& d = a.net[j];
    d.packed = j ? critic_packed : actor_packed; d.x = latent; d.ldx = L; d.kx = L; d.x2 = j ? critic_obs : obs; d.ldx2 = d.in_dim - L; d.rows = nullptr; d.y = nullptr; d.nrows = N; d.ldy = d.out_dim; d.normalize = 0;
  }
  a.std_ = std_; a.eps = eps; a.a_out = a_out; a.a_st = a_st; a.mu_st = mu_st; a.sig_st = sig_st; a.lp_st = lp_st; a.v_st = v_st;
  a.N = N; a.A = a.net[0].out_dim; a.mode = 1;
  return run(a, 2, stream);
}

int go2nn_sum_rows(const Go2nnSumJob* jobs, int32_t njobs, void* stream) {
  if (!jobs || njobs <= 0 || njobs > GO2NN_MAX_SUM_JOBS) FAIL(GO2NN_EINVAL, "sum rows: 1..%d jobs", GO2NN_MAX_SUM_JOBS);
  for (int j = 0; j < njobs; ++j) if (!jobs[j].part || !jobs[j].out || jobs[j].nrows <= 0 || jobs[j].ncols <= 0 || jobs[j].out_w < 0 ||
                                      (jobs[j].out_w > 0 && (jobs[j].out_ld < jobs[j].out_w || jobs[j].ncols % jobs[j].out_w))) FAIL(GO2NN_EINVAL, "sum rows: bad job %d", j);
#ifdef GO2_EMU
  (void)stream;
  for (int j = 0; j < njobs; ++j) for (int c = 0; c < jobs[j].ncols; ++c) {
    float s = 0.f;
    for (int r = 0; r < jobs[j].nrows; ++r) s += jobs[j].part[(int64_t)r * jobs[j].ncols + c];
    jobs[j].out[jobs[j].out_w > 0 ? (int64_t)(c / jobs[j].out_w) * jobs[j].out_ld + c % jobs[j].out_w : c] = s;
    if (jobs[j].acc && c < jobs[j].nacc) jobs[j].acc[c] += s;
  }
#else
  SumRowsArgs a; memset(&a, 0, sizeof(a));
  a.njobs = njobs; a.first_block[0] = 0;
  for (int j = 0; j < njobs; ++j) {
    a.part[j] = jobs[j].part; a.out[j] = jobs[j].out; a.nrows[j] = jobs[j].nrows; a.ncols[j] = jobs[j].ncols; a.acc[j] = jobs[j].acc; a.nacc[j] = jobs[j].acc ? jobs[j].nacc : 0;
    a.out_w[j] = jobs[j].out_w; a.out_ld[j] = jobs[j].out_ld;
    a.first_block[j + 1] = a.first_block[j] + (jobs[j].nrows <= 32 ? (jobs[j].ncols + 255) / 256 : (jobs[j].ncols + 31) / 32);
  }
  hipLaunchKernelGGL(go2nn_sum_rows_kernel, dim3(a.first_block[njobs]), dim3(256), 0, (hipStream_t)stream, a);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

int32_t go2nn_head_backward_rows(int32_t B, int32_t C, int32_t K) {
  if (go2nn_head_backward_workspace(B, C, K) < 0) return GO2NN_EINVAL;
#ifdef GO2_EMU
  return 1;
#else
  int q, rows, nwg; head_bwd_shape(B, K, &q, &rows, &nwg); return nwg;
#endif
}
int32_t go2nn_linear_backward_input_rows(int32_t M, int32_t C, int32_t Kin) {
  if (!lin_check(M, C, Kin)) FAIL(GO2NN_EINVAL, "linear backward: bad shape");
#ifdef GO2_EMU
  return 1;
#else
  int tm, tn; gemm_tile(M, Kin, &tm, &tn); return cdiv(M, gm_tile_rows(tm));
#endif
}

int64_t go2nn_head_backward_workspace(int32_t B, int32_t C, int32_t K) {
  if (B <= 0 || C <= 0 || C > HB_MAX_C || K <= 0 || K > GO2NN_MAX_WIDTH || (K & 3)) FAIL(GO2NN_EINVAL, "head backward: 1 <= C <= %d, K a multiple of 4 up to %d", HB_MAX_C, GO2NN_MAX_WIDTH);
  int q, rows, nwg; head_bwd_shape(B, K, &q, &rows, &nwg);
  return (int64_t)nwg * ((int64_t)(C + 1) * K + C);
}

int go2nn_head_backward(const float* gy, const float* y, const float* w, float* gz, float* sums, float* workspace, int32_t B, int32_t C, int32_t K, void* stream) {
  if (!gy || !y || !w || !gz || !workspace || go2nn_head_backward_workspace(B, C, K) < 0) FAIL(GO2NN_EINVAL, "head backward: bad argument");
#ifdef GO2_EMU
  (void)stream;
  const int64_t ns = (int64_t)(C + 1) * K + C;
  if (!sums) sums = workspace;           // partials only: the host build has ONE partial row (go2nn_head_backward_rows = 1)
  for (int64_t i = 0; i < ns; ++i) sums[i] = 0.f;
  for (int r = 0; r < B; ++r) {
    for (int c = 0; c < C; ++c) sums[(int64_t)(C + 1) * K + c] += gy[(int64_t)r * C + c];
    for (int k = 0; k < K; ++k) {
      const float v = y[(int64_t)r * K + k];
      float s = 0.f;
      for (int c = 0; c < C; ++c) { const float g = gy[(int64_t)r * C + c]; s = fmaf(g, w[(int64_t)c * K + k], s); sums[(int64_t)c * K + k] = fmaf(g, v, sums[(int64_t)c * K + k]); }
      const float o = s * (v > 0.f ? 1.f : v + 1.f);
      gz[(int64_t)r * K + k] = o; sums[(int64_t)C * K + k] += o;
    }
  }
#else
  int q, rows, nwg; head_bwd_shape(B, K, &q, &rows, &nwg);
  const int ncols = (C + 1) * K + C;
  hipStream_t st = (hipStream_t)stream;
  if (C == 1)       hipLaunchKernelGGL(go2nn_head_bwd_kernel<1>,  dim3(nwg), dim3(HB_THREADS), 0, st, gy, y, w, gz, workspace, B, C, K, q, rows);
  else if (C <= 4)  hipLaunchKernelGGL(go2nn_head_bwd_kernel<4>,  dim3(nwg), dim3(HB_THREADS), 0, st, gy, y, w, gz, workspace, B, C, K, q, rows);
  else if (C <= 8)  hipLaunchKernelGGL(go2nn_head_bwd_kernel<8>,  dim3(nwg), dim3(HB_THREADS), 0, st, gy, y, w, gz, workspace, B, C, K, q, rows);
  else if (C <= 12) hipLaunchKernelGGL(go2nn_head_bwd_kernel<12>, dim3(nwg), dim3(HB_THREADS), 0, st, gy, y, w, gz, workspace, B, C, K, q, rows);
  else              hipLaunchKernelGGL(go2nn_head_bwd_kernel<16>, dim3(nwg), dim3(HB_THREADS), 0, st, gy, y, w, gz, workspace, B, C, K, q, rows);
  HIPCHK(hipGetLastError());
  if (sums) { const Go2nnSumJob job = {workspace, sums, nwg, ncols}; return go2nn_sum_rows(&job, 1, stream); }
#endif
  return 0;
}

int go2nn_linear_elu_forward(const float* x, const float* w, const float* b, float* y, int32_t M, int32_t K, int32_t N, void* stream) {
  if (!x || !w || !b || !y || !lin_check(M, N, K)) FAIL(GO2NN_EINVAL, "linear forward: bad argument");
#ifdef GO2_EMU
  (void)stream;
  for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) {
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc = fmaf(x[(int64_t)m * K + k], w[(int64_t)n * K + k], acc);
    const float v = acc + b[n];
    y[(int64_t)m * N + n] = v > 0.f ? v : expm1f(v);
  }
#else
  GemmArgs g; memset(&g, 0, sizeof(g));
  int tm, tn; gemm_tile(M, N, &tm, &tn);
  g.A = x; g.B = w; g.C = y; g.bias = b; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N;
  const bool vec = (K % 4 == 0) && aligned16(x) && aligned16(w);
  g.kchunk = cdiv(K, GM_BK) * GM_BK; g.nbm = cdiv(M, gm_tile_rows(tm)); g.nbn = cdiv(N, 64 * tn);
  g.c_vec = (N % 4 == 0) && aligned16(y);
  gemm_dispatch<true, true, EPI_BIAS_ELU>(tm, tn, g, 1, vec, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

int64_t go2nn_linear_backward_workspace(int32_t M, int32_t C, int32_t Kin) {
  if (!lin_check(M, C, Kin)) FAIL(GO2NN_EINVAL, "linear backward: bad shape");
  int tm, tn, s, kc; wgrad_shape(M, C, Kin, &tm, &tn, &s, &kc);
  const int64_t wg = (int64_t)s * C * Kin;
  gemm_tile(M, Kin, &tm, &tn);
  const int64_t ig = (int64_t)cdiv(M, 64) * Kin;          // (an upper bound over the tile shapes: 64 rows per partial row at least)
  return wg > ig ? wg : ig;
}

int go2nn_linear_backward_input(const float* gz, const float* w, const float* y_prev, float* gz_prev, float* gb_prev, float* workspace,
                                int32_t M, int32_t C, int32_t Kin, void* stream) {
  if (!gz || !w || !y_prev || !gz_prev || !workspace || !lin_check(M, C, Kin)) FAIL(GO2NN_EINVAL, "linear backward (input): bad argument");
#ifdef GO2_EMU
  (void)stream;
  if (!gb_prev) gb_prev = workspace;     // partials only (one partial row in the host build)
  for (int k = 0; k < Kin; ++k) gb_prev[k] = 0.f;
  for (int m = 0; m < M; ++m) for (int k = 0; k < Kin; ++k) {
    float acc = 0.f;
    for (int c = 0; c < C; ++c) acc = fmaf(gz[(int64_t)m * C + c], w[(int64_t)c * Kin + k], acc);
    const float yv = y_prev[(int64_t)m * Kin + k], o = acc * (yv > 0.f ? 1.f : yv + 1.f);
    gz_prev[(int64_t)m * Kin + k] = o; gb_prev[k] += o;
  }
#else
  GemmArgs g; memset(&g, 0, sizeof(g));
  int tm, tn; gemm_tile(M, Kin, &tm, &tn);
  g.A = gz; g.B = w; g.C = gz_prev; g.Y = y_prev; g.part = workspace; g.M = M; g.N = Kin; g.K = C; g.lda = C; g.ldb = Kin; g.ldc = Kin;
  const bool vec = (C % 4 == 0) && (Kin % 4 == 0) && aligned16(gz) && aligned16(w);
  g.kchunk = cdiv(C, GM_BK) * GM_BK; g.nbm = cdiv(M, gm_tile_rows(tm)); g.nbn = cdiv(Kin, 64 * tn);
  g.c_vec = (Kin % 4 == 0) && aligned16(gz_prev) && aligned16(y_prev);
  gemm_dispatch<true, false, EPI_DELU_COLSUM>(tm, tn, g, 1, vec, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  if (gb_prev) { const Go2nnSumJob job = {workspace, gb_prev, g.nbm, Kin}; return go2nn_sum_rows(&job, 1, stream); }
#endif
  return 0;
}

int go2nn_linear_backward_weight(const float* gz, const float* x, float* dw, float* workspace, int32_t M, int32_t C, int32_t Kin, void* stream) {
  if (!gz || !x || !dw || !workspace || !lin_check(M, C, Kin)) FAIL(GO2NN_EINVAL, "linear backward (weight): bad argument");
#ifdef GO2_EMU
  (void)stream; (void)workspace;
  for (int64_t i = 0; i < (int64_t)C * Kin; ++i) dw[i] = 0.f;
  for (int m = 0; m < M; ++m) for (int c = 0; c < C; ++c) { const float gv = gz[(int64_t)m * C + c]; for (int k = 0; k < Kin; ++k) dw[(int64_t)c * Kin + k] = fmaf(gv, x[(int64_t)m * Kin + k], dw[(int64_t)c * Kin + k]); }
#else
  GemmArgs g; memset(&g, 0, sizeof(g));
  int tm, tn, s, kc; wgrad_shape(M, C, Kin, &tm, &tn, &s, &kc);
  g.A = gz; g.B = x; g.C = s > 1 ? workspace : dw; g.M = C; g.N = Kin; g.K = M; g.lda = C; g.ldb = Kin; g.ldc = Kin;
  const bool vec = (C % 4 == 0) && (Kin % 4 == 0) && aligned16(gz) && aligned16(x);
  g.kchunk = kc; g.c_split_stride = (long long)C * Kin; g.nbm = cdiv(C, 64 * tm); g.nbn = cdiv(Kin, 64 * tn);
  g.c_vec = (Kin % 4 == 0) && aligned16(g.C);
  gemm_dispatch<false, false, EPI_STORE>(tm, tn, g, s, vec, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  if (s > 1) { const Go2nnSumJob job = {workspace, dw, s, C * Kin}; return go2nn_sum_rows(&job, 1, stream); }
#endif
  return 0;
}

#ifdef GO2NN_STAMPS
// TOOL-ONLY (tools/policy_bench.py --stamps): go2nn_policy_act with a per-workgroup timestamp buffer [2 * ceil(N / 32)][8] int64
int go2nn_policy_act_stamped(const Go2nnMlp* actor, const float* actor_packed, const Go2nnMlp* critic, const float* critic_packed,
                             const float* obs, const float* critic_obs, const float* std_, const float* eps, float* a_out, void* stamps, int32_t N, void* stream) {
  NNArgs a; memset(&a, 0, sizeof(a));
  if (!describe(actor, &a.net[0], nullptr) || !describe(critic, &a.net[1], nullptr)) return GO2NN_EINVAL;
  a.net[0].packed = actor_packed; net_io(a.net[0], obs, N, nullptr); a.net[1].packed = critic_packed; net_io(a.net[1], critic_obs, N, nullptr);
  a.std_ = std_; a.eps = eps; a.a_out = a_out; a.y = (float*)stamps; a.N = N; a.A = a.net[0].out_dim; a.mode = 1;
  return run(a, 2, stream);
}
#endif

}  // extern "C"

// ---- ABI 3: grouped layer calls (go2nn_gemm3.h) ---------------------------------------------------------------------------------------------
#ifndef GO2_EMU
#ifdef GM3_STAMPS
static long long* g_gemm3_stamps = nullptr;
extern "C" void go2nn_debug_gemm3_stamps(long long* p) { g_gemm3_stamps = p; }      // TOOL-ONLY: device buffer [workgroups][4 waves][8] of the next launches
#define GM3_SET_STAMPS(a) (a).stamps = g_gemm3_stamps
#else
#define GM3_SET_STAMPS(a) do { } while (0)
#endif
template <bool BKC, int EPI>
static int gemm3_launch(int tm, int tn, int bk, const Gemm3Args& a, hipStream_t st) {
  const dim3 grid(a.ntiles), blk(256);
  if (tm == 1 && tn == 2 && bk == 32)      hipLaunchKernelGGL((go2nn_gemm3_kernel<1, 2, BKC, EPI, 32>), grid, blk, 0, st, a);
  else if (tm == 1 && tn == 1 && bk == 32) hipLaunchKernelGGL((go2nn_gemm3_kernel<1, 1, BKC, EPI, 32>), grid, blk, 0, st, a);
  else FAIL(GO2NN_EINVAL, "gemm3: no kernel for tile %d x %d x %d", tm, tn, bk);
  HIPCHK(hipGetLastError());
  return 0;
}
template <int EPI>
static int bx3_launch(int tm, const Bx3Args& a, hipStream_t st) {
  const dim3 grid(a.ntiles), blk(256);
  if (tm == 3) { if constexpr (EPI == EPI_BIAS_ELU || EPI == EPI_BIAS) hipLaunchKernelGGL((go2nn_bx3_kernel<3, EPI>), grid, blk, 0, st, a); else FAIL(GO2NN_EINVAL, "bx3: no 192-row tile for this epilogue"); }
  else if (tm == 2) hipLaunchKernelGGL((go2nn_bx3_kernel<2, EPI>), grid, blk, 0, st, a);
  else              hipLaunchKernelGGL((go2nn_bx3_kernel<1, EPI>), grid, blk, 0, st, a);
  HIPCHK(hipGetLastError());
  return 0;
}
// rows per workgroup tile of the split-operand kernels (64 tm).  All workgroups of a launch are resident at once only if they fit the chip's slots (2 per CU for
// the 128- and 192-row tiles, 3 for 64 rows); a launch of 768 workgroups on 512 slots runs a second, half-empty round (measured: MFMA busy 37 %).  So: the tile
// height whose workgroup count fills whole rounds best, the larger one on a tie (less weight traffic per MFMA, fewer prologues and epilogues).
static inline int bx3_tm(int M, int N0, int N1, int tm_max) {
  static const char* const env = getenv("GO2NN_BX3_TM");        // tools only
  if (env && env[0] >= '1' && env[0] <= '0' + tm_max) return env[0] - '0';
  int best = 1; double best_eff = -1.0;
  for (int tm = 1; tm <= tm_max; ++tm) {
    const long long tiles = (long long)cdiv(M, 64 * tm) * (cdiv(N0, 128) + (N1 ? cdiv(N1, 128) : 0)), slots = tm == 1 ? 768 : 512;
    const double eff = (double)tiles / (double)((tiles + slots - 1) / slots * slots);
    if (eff >= best_eff - 1e-9) { best_eff = eff > best_eff ? eff : best_eff; best = tm; }
  }
  return best;
}
template <bool VECA, bool VECB>
static int wgrad3_launch2(int ta, int tn, const WgArgs& a, hipStream_t st) {
  const dim3 grid(a.tiles * a.nsplit), blk(256);
  if (tn == 4) hipLaunchKernelGGL((go2nn_wgrad_kernel<2, 4, VECA, VECB, 6>), grid, blk, 0, st, a);
  else              hipLaunchKernelGGL((go2nn_wgrad_kernel<2, 2, VECA, VECB, 8>), grid, blk, 0, st, a);
  HIPCHK(hipGetLastError());
  return 0;
}
#endif
// tile of the grouped weight gradient: 64 x 128 (2 x 4 column tiles per lane load) when every Kin is a multiple of 128, else 64 x 64 (ragged widths waste less:
// 263 -> 320 instead of 384 columns).  Measured and rejected for the input layer's dW [512, 45 | 263] (93-95 us; the padded tiles' MFMAs alone are 70 us at 2.1 GHz):
// 128 x 64 tiles (a quarter fewer bytes through the lanes' loads, but 200 registers = 2 waves per SIMD instead of 4: 117 us) and 256 x 64 workgroup tiles with the X
// columns shared through LDS (16-row chunks, four waves on four column blocks, no cross-wave reduction: 121 us — two barriers per chunk cost more than the loads saved)
static int wgrad3_group_shape(const Go2nnBwdWJob* jobs, int njobs, int* ta, int* tn, int* tiles_of, int* nsplit, int* rows) {
  if (!jobs || njobs < 1 || njobs > GO2NN_MAX_GROUP) return 0;
  *tn = 4; *ta = 2;
  bool split = true; int t128 = 0;
  int kin_sum = 0, kin_pad = 0;
  for (int j = 0; j < njobs; ++j) {
    if (!lin_check(jobs[j].M, jobs[j].C, jobs[j].Kin) || jobs[j].M != jobs[0].M || jobs[j].C < 2 || jobs[j].Kin < 4) return 0;
    if (jobs[j].Kin % 128) *tn = 2;
    if (!jobs[j].split || jobs[j].C % 128) split = false;
    t128 += (jobs[j].C / 128) * cdiv(jobs[j].Kin, 128); kin_sum += jobs[j].Kin; kin_pad += cdiv(jobs[j].Kin, 128) * 128;
  }
#ifndef GO2_EMU
  // split-operand weight gradient (go2nn_bx3_kernel<2, EPI_STORE, true>; *ta = 9 stands for it): asked for by every job, C in whole 128-row tiles, Kin padded to 128-column
  // tiles by at most a factor 2 (measured: dW [512, 45 | 263] 97 -> 87 us although 40 % of its MFMAs are padding; [512, 45 | 48] 40 -> 49 us), and enough tiles that a
  // slice is long (the 256 -> 128 layer's 4 tiles would be 128 slices of 12 k-tiles each, and as many partial tiles to sum: no gain measured)
  if (kin_pad > 2 * kin_sum) split = false;
  static const bool no_wg_split = getenv("GO2NN_WG_SPLIT") && atoi(getenv("GO2NN_WG_SPLIT")) == 0;          // tools only (A/B)
  if (split && t128 >= 8 && !no_wg_split) {
    *ta = 9;
    for (int j = 0; j < njobs; ++j) tiles_of[j] = (jobs[j].C / 128) * cdiv(jobs[j].Kin, 128);
    wgrad3_shape(jobs[0].M, t128, nsplit, rows);
    *rows = (*rows + BX3_BK - 1) / BX3_BK * BX3_BK;
    return t128;
  }
#endif
  int tiles = 0;
  for (int j = 0; j < njobs; ++j) { tiles_of[j] = cdiv(jobs[j].C, 32 * *ta) * cdiv(jobs[j].Kin, 32 * *tn); tiles += tiles_of[j]; }
  wgrad3_shape(jobs[0].M, tiles, nsplit, rows);
  return tiles;
}

extern "C" {

int64_t go2nn_split_weights_bytes(int32_t N, int32_t K) {
  if (N <= 0 || K <= 0 || N > 4096 || K > 4096) FAIL(GO2NN_EINVAL, "split image: bad shape");
  return bx3_image_bytes(N, K) + bx3_image_bytes(K, N);
}

int go2nn_split_weights(const Go2nnSplitJob* jobs, int32_t njobs, void* stream) {
  if (!jobs || njobs < 1 || njobs > GO2NN_MAX_SPLIT_JOBS) FAIL(GO2NN_EINVAL, "split weights: 1..%d jobs", GO2NN_MAX_SPLIT_JOBS);
  for (int j = 0; j < njobs; ++j) if (!jobs[j].w || !jobs[j].image || jobs[j].N <= 0 || jobs[j].K <= 0 || jobs[j].N > 4096 || jobs[j].K > 4096 || ((uintptr_t)jobs[j].image & 15)) FAIL(GO2NN_EINVAL, "split weights: bad job %d", j);
#ifdef GO2_EMU
  return 0;          // (the host build's products read the fp32 weights)
#else
  Bx3SplitArgs a; memset(&a, 0, sizeof(a)); a.njobs = njobs;
  long long most = 0;
  for (int j = 0; j < njobs; ++j) { a.j[j].w = jobs[j].w; a.j[j].img = (unsigned char*)jobs[j].image; a.j[j].N = jobs[j].N; a.j[j].K = jobs[j].K;
    most = std::max(most, std::max(bx3_image_bytes(jobs[j].N, jobs[j].K), bx3_image_bytes(jobs[j].K, jobs[j].N)) / 48); }          // threads: one per 3 x 16 bytes
  hipLaunchKernelGGL(go2nn_bx3_split_kernel, dim3((unsigned)cdiv((int)most, 256), 2 * njobs), dim3(256), 0, (hipStream_t)stream, a);
  HIPCHK(hipGetLastError());
  return 0;
#endif
}

int go2nn_linear_elu_forward_group(const Go2nnFwdJob* jobs, int32_t njobs, void* stream) {
  if (!jobs || njobs < 1 || njobs > GO2NN_MAX_GROUP) FAIL(GO2NN_EINVAL, "forward group: 1..%d jobs", GO2NN_MAX_GROUP);
  for (int j = 0; j < njobs; ++j) if (!jobs[j].x || !jobs[j].w || !jobs[j].b || !jobs[j].y || !lin_check(jobs[j].M, jobs[j].N, jobs[j].K)) FAIL(GO2NN_EINVAL, "forward group: bad job %d", j);
  const int act = jobs[0].act;
  if (act != 0 && act != 1) FAIL(GO2NN_EINVAL, "forward group: act 0 (ELU) or 1 (none)");
  for (int j = 1; j < njobs; ++j) if (jobs[j].act != act) FAIL(GO2NN_EINVAL, "forward group: every job of a group carries the same activation flag");
#ifdef GO2_EMU
  for (int j = 0; j < njobs; ++j) {
    if (!act) { const int rc = go2nn_linear_elu_forward(jobs[j].x, jobs[j].w, jobs[j].b, jobs[j].y, jobs[j].M, jobs[j].K, jobs[j].N, stream); if (rc) return rc; continue; }
    const Go2nnFwdJob& q = jobs[j];
    for (int m = 0; m < q.M; ++m) for (int n = 0; n < q.N; ++n) {
      float acc = 0.f;
      for (int k = 0; k < q.K; ++k) acc = fmaf(q.x[(int64_t)m * q.K + k], q.w[(int64_t)n * q.K + k], acc);
      q.y[(int64_t)m * q.N + n] = acc + q.b[n];
    }
  }
  return 0;
#else
  if (jobs[0].w_split && jobs[njobs - 1].w_split && jobs[0].K >= 4 && jobs[njobs - 1].K >= 4) {          // split-operand kernel (go2nn_bx3.h)
    Bx3Args a; memset(&a, 0, sizeof(a));
    const int tm = bx3_tm(jobs[0].M, jobs[0].N, njobs == 2 ? jobs[1].N : 0, 3);
    for (int j = 0; j < njobs; ++j) {
      Bx3Prob& g = a.p[j]; const Go2nnFwdJob& q = jobs[j];
      g.A = q.x; g.B = (const unsigned char*)q.w_split; g.C = q.y; g.bias = q.b; g.M = q.M; g.N = q.N; g.K = q.K; g.lda = q.K; g.ldc = q.N;
      g.nbm = cdiv(q.M, 64 * tm); g.nbn = cdiv(q.N, 128); g.nkt = cdiv(q.K, BX3_BK); g.c_vec = (q.N % 4 == 0) && aligned16(q.y);
      (j ? a.ntiles : a.ntiles0) = g.nbm * g.nbn;
    }
    a.ntiles = njobs == 2 ? a.ntiles0 + a.ntiles : a.ntiles0;
    GM3_SET_STAMPS(a);
    return act ? bx3_launch<EPI_BIAS>(tm, a, (hipStream_t)stream) : bx3_launch<EPI_BIAS_ELU>(tm, a, (hipStream_t)stream);
  }
  int tm, tn, bk; gemm3_tile(jobs[0].N, &tm, &tn, &bk);
  if (njobs == 2) { int tm1, tn1, bk1; gemm3_tile(jobs[1].N, &tm1, &tn1, &bk1);
    if (tm1 != tm || tn1 != tn || bk1 != bk) { const int rc = go2nn_linear_elu_forward_group(jobs, 1, stream); return rc ? rc : go2nn_linear_elu_forward_group(jobs + 1, 1, stream); } }
  for (int j = 0; j < njobs; ++j) if (jobs[j].K < 4) {          // (the staging's clamped 16-byte loads need 4 floats per row) -> the single-network kernels
    if (act) FAIL(GO2NN_EINVAL, "forward group: a layer without activation needs K >= 4");
    for (int i = 0; i < njobs; ++i) { const int rc = go2nn_linear_elu_forward(jobs[i].x, jobs[i].w, jobs[i].b, jobs[i].y, jobs[i].M, jobs[i].K, jobs[i].N, stream); if (rc) return rc; }
    return 0; }
  Gemm3Args a; memset(&a, 0, sizeof(a));
  for (int j = 0; j < njobs; ++j) {
    Gemm3Prob& g = a.p[j]; const Go2nnFwdJob& q = jobs[j];
    g.A = q.x; g.B = q.w; g.C = q.y; g.bias = q.b; g.M = q.M; g.N = q.N; g.K = q.K; g.lda = q.K; g.ldb = q.K; g.ldc = q.N;
    g.nbm = cdiv(q.M, 64 * tm); g.nbn = cdiv(q.N, 64 * tn); g.c_vec = (q.N % 4 == 0) && aligned16(q.y);
    (j ? a.ntiles : a.ntiles0) = g.nbm * g.nbn;
  }
  a.ntiles = njobs == 2 ? a.ntiles0 + a.ntiles : a.ntiles0;
  GM3_SET_STAMPS(a);
  return act ? gemm3_launch<true, EPI_BIAS>(tm, tn, bk, a, (hipStream_t)stream) : gemm3_launch<true, EPI_BIAS_ELU>(tm, tn, bk, a, (hipStream_t)stream);
#endif
}

int32_t go2nn_linear_backward_input_group_rows(int32_t M, int32_t C, int32_t Kin) {
  if (!lin_check(M, C, Kin)) FAIL(GO2NN_EINVAL, "linear backward: bad shape");
#ifdef GO2_EMU
  return 1;
#else
  int tm, tn, bk; gemm3_tile(Kin, &tm, &tn, &bk); return cdiv(M, 64 * tm);
#endif
}

int32_t go2nn_linear_backward_input_fused_rows(int32_t M) {
  if (M <= 0) FAIL(GO2NN_EINVAL, "linear backward: bad shape");
#ifdef GO2_EMU
  return 1;
#else
  return cdiv(M, 128);
#endif
}

int go2nn_linear_backward_input_group(const Go2nnBwdInJob* jobs, int32_t njobs, void* stream) {
  if (!jobs || njobs < 1 || njobs > GO2NN_MAX_GROUP) FAIL(GO2NN_EINVAL, "input-gradient group: 1..%d jobs", GO2NN_MAX_GROUP);
  const int plain = jobs[0].plain;
  if (plain != 0 && plain != 1) FAIL(GO2NN_EINVAL, "input-gradient group: plain 0 or 1");
  const bool fused = jobs[0].x_in != nullptr;          // ABI 6: + the weight gradient of the layer below
  for (int j = 0; j < njobs; ++j) if (jobs[j].plain != plain || !jobs[j].gz || !jobs[j].w || (!jobs[j].gz_prev && !fused) || (!plain && (!jobs[j].y_prev || !jobs[j].workspace)) || !lin_check(jobs[j].M, jobs[j].C, jobs[j].Kin) ||
                                      (jobs[j].ld != 0 && jobs[j].ld < jobs[j].Kin)) FAIL(GO2NN_EINVAL, "input-gradient group: bad job %d", j);
  if (fused) for (int j = 0; j < njobs; ++j) if (!jobs[j].x_in || !jobs[j].dw_workspace || jobs[j].Kx < 1 || jobs[j].Kx > 64 || plain || jobs[j].ld || !jobs[j].w_split || jobs[j].C < 4 || (jobs[j].ldx != 0 && jobs[j].ldx < jobs[j].Kx))
    FAIL(GO2NN_EINVAL, "input-gradient group with x_in: every job with x_in [M, 1..64], dw_workspace and w_split, plain 0, ld 0 (job %d)", j);
  if (!fused) for (int j = 0; j < njobs; ++j) if (jobs[j].x_in) FAIL(GO2NN_EINVAL, "input-gradient group: x_in on every job of a group or on none");
#ifdef GO2_EMU
  if (fused) {
    for (int j = 0; j < njobs; ++j) {
      const Go2nnBwdInJob& q = jobs[j];
      std::vector<float> gp((size_t)q.M * q.Kin);
      const int rc = go2nn_linear_backward_input(q.gz, q.w, q.y_prev, gp.data(), nullptr, q.workspace, q.M, q.C, q.Kin, stream); if (rc) return rc;
      if (q.gz_prev) memcpy(q.gz_prev, gp.data(), gp.size() * sizeof(float));
      std::vector<float> xd;          // (a column block of a wider input: made dense for the single-network call)
      if (q.ldx && q.ldx != q.Kx) { xd.resize((size_t)q.M * q.Kx); for (int m = 0; m < q.M; ++m) memcpy(&xd[(size_t)m * q.Kx], q.x_in + (size_t)m * q.ldx, q.Kx * sizeof(float)); }
      const int rc2 = go2nn_linear_backward_weight(gp.data(), xd.empty() ? q.x_in : xd.data(), q.dw_workspace, q.dw_workspace, q.M, q.Kin, q.Kx, stream); if (rc2) return rc2;
    }
    return 0;
  }
  for (int j = 0; j < njobs; ++j) {
    if (!plain && !jobs[j].ld) { const int rc = go2nn_linear_backward_input(jobs[j].gz, jobs[j].w, jobs[j].y_prev, jobs[j].gz_prev, nullptr, jobs[j].workspace, jobs[j].M, jobs[j].C, jobs[j].Kin, stream); if (rc) return rc; continue; }
    const Go2nnBwdInJob& q = jobs[j];
    const int64_t ld = q.ld ? q.ld : q.Kin;
    if (!plain) for (int k = 0; k < q.Kin; ++k) q.workspace[k] = 0.f;          // (the host build leaves ONE partial row: go2nn_linear_backward_input_group_rows = 1)
    for (int m = 0; m < q.M; ++m) for (int k = 0; k < q.Kin; ++k) {
      float acc = 0.f;
      for (int c = 0; c < q.C; ++c) acc = fmaf(q.gz[(int64_t)m * q.C + c], q.w[(int64_t)c * q.Kin + k], acc);
      if (!plain) { const float y = q.y_prev[m * ld + k]; acc *= y > 0.f ? 1.f : y + 1.f; q.workspace[k] += acc; }
      q.gz_prev[m * ld + k] = acc;
    }
  }
  return 0;
#else
  if (jobs[0].w_split && jobs[njobs - 1].w_split && jobs[0].C >= 4 && jobs[njobs - 1].C >= 4) {          // split-operand kernel: A = gz [M,C], B = the transposed image (rows k, contraction c)
    Bx3Args a; memset(&a, 0, sizeof(a));
    const int tm = fused ? 2 : bx3_tm(jobs[0].M, jobs[0].Kin, njobs == 2 ? jobs[1].Kin : 0, 2);          // (192-row tiles measured equal for the input gradient: 100.4 against 99.5 us; not instantiated)
    for (int j = 0; j < njobs; ++j) {
      Bx3Prob& g = a.p[j]; const Go2nnBwdInJob& q = jobs[j];
      g.Bf = q.x_in; g.ldb = q.ldx ? q.ldx : q.Kx; g.kx = q.Kx; g.wpart = q.dw_workspace;
      g.A = q.gz; g.B = (const unsigned char*)q.w_split + bx3_image_bytes(q.C, q.Kin); g.C = q.gz_prev; g.Y = q.y_prev; g.part = q.workspace;
      g.M = q.M; g.N = q.Kin; g.K = q.C; g.lda = q.C; g.ldc = q.ld ? q.ld : q.Kin;
      g.nbm = cdiv(q.M, 64 * tm); g.nbn = cdiv(q.Kin, 128); g.nkt = cdiv(q.C, BX3_BK); g.c_vec = (q.Kin % 4 == 0) && (g.ldc % 4 == 0) && aligned16(q.gz_prev) && (plain || aligned16(q.y_prev));
      (j ? a.ntiles : a.ntiles0) = g.nbm * g.nbn;
    }
    a.ntiles = njobs == 2 ? a.ntiles0 + a.ntiles : a.ntiles0;
    GM3_SET_STAMPS(a);
    if (fused) {
      hipLaunchKernelGGL((go2nn_bx3_kernel<2, EPI_DELU_WG>), dim3(a.ntiles), dim3(256), 0, (hipStream_t)stream, a);
      HIPCHK(hipGetLastError());
      return 0;
    }
    return plain ? bx3_launch<EPI_STORE>(tm, a, (hipStream_t)stream) : bx3_launch<EPI_DELU_COLSUM>(tm, a, (hipStream_t)stream);
  }
  int tm, tn, bk; gemm3_tile(jobs[0].Kin, &tm, &tn, &bk);
  if (njobs == 2) { int tm1, tn1, bk1; gemm3_tile(jobs[1].Kin, &tm1, &tn1, &bk1);
    if (tm1 != tm || tn1 != tn || bk1 != bk) { const int rc = go2nn_linear_backward_input_group(jobs, 1, stream); return rc ? rc : go2nn_linear_backward_input_group(jobs + 1, 1, stream); } }
  for (int j = 0; j < njobs; ++j) if (jobs[j].C < 4 || jobs[j].Kin < 4 || jobs[j].Kin % 4) {          // (column quads of W must not straddle Kin) -> the single-network kernels
    for (int i = 0; i < njobs; ++i) if (jobs[i].ld) FAIL(GO2NN_EINVAL, "pitched input gradient on the fp32-MFMA kernels: C >= 4 and Kin a multiple of 4");
    if (plain) FAIL(GO2NN_EINVAL, "plain input gradient on the fp32-MFMA kernels: C >= 4 and Kin a multiple of 4 (the split-operand kernel takes any Kin)");
    // (go2nn_linear_backward_input picks its own tile height — 128 rows for Kin >= 512 — and then leaves HALF the partial rows the group's callers were told
    //  to sum, go2nn_linear_backward_input_group_rows = one per 64 rows: the single-network kernel is launched here with 64-row tiles whatever the shape)
    for (int i = 0; i < njobs; ++i) {
      const Go2nnBwdInJob& q = jobs[i];
      GemmArgs g; memset(&g, 0, sizeof(g));
      int tm1, tn1; gemm_tile(q.M, q.Kin, &tm1, &tn1); tm1 = 1;
      g.A = q.gz; g.B = q.w; g.C = q.gz_prev; g.Y = q.y_prev; g.part = q.workspace; g.M = q.M; g.N = q.Kin; g.K = q.C; g.lda = q.C; g.ldb = q.Kin; g.ldc = q.Kin;
      const bool vec = (q.C % 4 == 0) && (q.Kin % 4 == 0) && aligned16(q.gz) && aligned16(q.w);
      g.kchunk = cdiv(q.C, GM_BK) * GM_BK; g.nbm = cdiv(q.M, gm_tile_rows(tm1)); g.nbn = cdiv(q.Kin, 64 * tn1);
      g.c_vec = (q.Kin % 4 == 0) && aligned16(q.gz_prev) && aligned16(q.y_prev);
      gemm_dispatch<true, false, EPI_DELU_COLSUM>(tm1, tn1, g, 1, vec, (hipStream_t)stream);
      HIPCHK(hipGetLastError());
    }
    return 0; }
  Gemm3Args a; memset(&a, 0, sizeof(a));
  for (int j = 0; j < njobs; ++j) {
    Gemm3Prob& g = a.p[j]; const Go2nnBwdInJob& q = jobs[j];
    g.A = q.gz; g.B = q.w; g.C = q.gz_prev; g.Y = q.y_prev; g.part = q.workspace; g.M = q.M; g.N = q.Kin; g.K = q.C; g.lda = q.C; g.ldb = q.Kin; g.ldc = q.ld ? q.ld : q.Kin;
    g.nbm = cdiv(q.M, 64 * tm); g.nbn = cdiv(q.Kin, 64 * tn); g.c_vec = (q.Kin % 4 == 0) && (g.ldc % 4 == 0) && aligned16(q.gz_prev) && (plain || aligned16(q.y_prev));
    (j ? a.ntiles : a.ntiles0) = g.nbm * g.nbn;
  }
  a.ntiles = njobs == 2 ? a.ntiles0 + a.ntiles : a.ntiles0;
  GM3_SET_STAMPS(a);
  return plain ? gemm3_launch<false, EPI_STORE>(tm, tn, bk, a, (hipStream_t)stream) : gemm3_launch<false, EPI_DELU_COLSUM>(tm, tn, bk, a, (hipStream_t)stream);
#endif
}

int32_t go2nn_linear_backward_weight_group_rows(const Go2nnBwdWJob* jobs, int32_t njobs) {
#ifdef GO2_EMU
  if (!jobs || njobs < 1 || njobs > GO2NN_MAX_GROUP) FAIL(GO2NN_EINVAL, "weight-gradient group: 1..%d jobs", GO2NN_MAX_GROUP);
  return 1;
#else
  int ta, tn, tiles_of[GO2NN_MAX_GROUP], nsplit, rows;
  if (!wgrad3_group_shape(jobs, njobs, &ta, &tn, tiles_of, &nsplit, &rows)) FAIL(GO2NN_EINVAL, "weight-gradient group: 1..%d jobs with one M, C >= 2, Kin >= 4", GO2NN_MAX_GROUP);
  return nsplit;
#endif
}

int go2nn_linear_backward_weight_group(const Go2nnBwdWJob* jobs, int32_t njobs, void* stream) {
  int ta, tn, tiles_of[GO2NN_MAX_GROUP], nsplit, rows;
  const int tiles = wgrad3_group_shape(jobs, njobs, &ta, &tn, tiles_of, &nsplit, &rows);
  if (!tiles) FAIL(GO2NN_EINVAL, "weight-gradient group: 1..%d jobs with one M, C >= 2, Kin >= 4", GO2NN_MAX_GROUP);
  for (int j = 0; j < njobs; ++j) if (!jobs[j].gz || !jobs[j].x || !jobs[j].workspace || (jobs[j].ldx != 0 && jobs[j].ldx < jobs[j].Kin)) FAIL(GO2NN_EINVAL, "weight-gradient group: bad job %d", j);
#ifdef GO2_EMU
  for (int j = 0; j < njobs; ++j) {
    const Go2nnBwdWJob& q = jobs[j];
    std::vector<float> xd;
    if (q.ldx && q.ldx != q.Kin) { xd.resize((size_t)q.M * q.Kin); for (int m = 0; m < q.M; ++m) memcpy(&xd[(size_t)m * q.Kin], q.x + (size_t)m * q.ldx, q.Kin * sizeof(float)); }
    const int rc = go2nn_linear_backward_weight(q.gz, xd.empty() ? q.x : xd.data(), q.workspace, q.workspace, q.M, q.C, q.Kin, stream); if (rc) return rc; }
  return 0;
#else
  for (int j = 0; j < njobs; ++j) if (jobs[j].ldx && jobs[j].ldx != jobs[j].Kin && ta != 9) FAIL(GO2NN_EINVAL, "weight-gradient group: a pitched x (ldx) on the split-operand kernel only (C in whole 128-row tiles, >= 8 tiles)");
  WgArgs a; memset(&a, 0, sizeof(a));
  if (ta == 9) {
    Bx3Args b; memset(&b, 0, sizeof(b));
    for (int j = 0; j < njobs; ++j) {
      Bx3Prob& g = b.p[j]; const Go2nnBwdWJob& q = jobs[j];
      g.A = q.gz; g.lda = q.C; g.Bf = q.x; g.ldb = q.ldx ? q.ldx : q.Kin; g.C = q.workspace; g.ldc = q.Kin; g.M = q.C; g.N = q.Kin; g.K = rows;
      g.nbm = q.C / 128; g.nbn = cdiv(q.Kin, 128); g.c_vec = aligned16(q.workspace) && (q.Kin % 4 == 0);
    }
    b.ntiles0 = tiles_of[0]; b.ntiles = tiles; b.wg_M = jobs[0].M; b.wg_rows = rows;
    hipLaunchKernelGGL((go2nn_bx3_kernel<2, EPI_STORE, true>), dim3(tiles * nsplit), dim3(256), 0, (hipStream_t)stream, b);
    HIPCHK(hipGetLastError());
    return 0;
  }
  bool veca = true, vecb = true;
  for (int j = 0; j < njobs; ++j) {
    WgProb& g = a.p[j]; const Go2nnBwdWJob& q = jobs[j];
    g.G = q.gz; g.X = q.x; g.part = q.workspace; g.C = q.C; g.Kin = q.Kin; g.ntc = cdiv(q.C, 32 * ta); g.ntk = cdiv(q.Kin, 32 * tn);
    veca = veca && (q.C % ta == 0) && (((uintptr_t)q.gz & (4 * ta - 1)) == 0);
    vecb = vecb && (q.Kin % tn == 0) && (((uintptr_t)q.x & (4 * tn - 1)) == 0);
  }
  a.M = jobs[0].M; a.rows_per_slice = rows; a.nsplit = nsplit; a.tiles0 = tiles_of[0]; a.tiles = tiles;
  hipStream_t st = (hipStream_t)stream;
  return veca ? (vecb ? wgrad3_launch2<true, true>(ta, tn, a, st) : wgrad3_launch2<true, false>(ta, tn, a, st))
              : (vecb ? wgrad3_launch2<false, true>(ta, tn, a, st) : wgrad3_launch2<false, false>(ta, tn, a, st));
#endif
}

}  // extern "C"

extern "C" {

static int ppo_heads_ok(int B, int A, int K) { return B > 0 && A >= 1 && A <= HB_MAX_C && K >= 4 && K <= 256 && (K & 3) == 0; }
int32_t go2nn_ppo_heads_cols(int32_t A, int32_t K) {
  if (!ppo_heads_ok(1, A, K)) FAIL(GO2NN_EINVAL, "ppo heads: 1 <= A <= %d, K a multiple of 4 up to 256", HB_MAX_C);
  return ppo_heads_cols(A, K);
}
int32_t go2nn_ppo_heads_rows(int32_t B, int32_t A, int32_t K) {
  if (!ppo_heads_ok(B, A, K)) FAIL(GO2NN_EINVAL, "ppo heads: bad shape");
#ifdef GO2_EMU
  return 1;
#else
  int q, rows, nwg; ppo_heads_shape(B, K, &q, &rows, &nwg); return nwg;
#endif
}
int go2nn_ppo_heads(const Go2nnPpoHeads* h, void* stream) {
  if (!h || !ppo_heads_ok(h->B, h->A, h->K)) FAIL(GO2NN_EINVAL, "ppo heads: bad shape");
  const void* ptrs[] = {h->y_a, h->y_c, h->w_mu, h->b_mu, h->w_v, h->b_v, h->std, h->actions, h->old_mu, h->old_sigma, h->old_logp, h->adv, h->old_values, h->returns, h->gz_a, h->gz_c, h->partials};
  for (const void* q : ptrs) if (!q) FAIL(GO2NN_EINVAL, "ppo heads: null pointer");
  const int B = h->B, A = h->A, K = h->K;
#ifdef GO2_EMU
  (void)stream;
  const float LOG2PI = 1.8378770664093453f, invB = 1.f / (float)B, lo = 1.f - h->clip, hi = 1.f + h->clip;
  const int split = (h->surrogate_split > 0 && h->surrogate_split < B) ? h->surrogate_split : 0;
  const float w_head = split ? 1.f / (float)split : invB, w_tail = split ? 1.f / (float)(B - split) : invB;
  float* S = h->partials; const int nc = ppo_heads_cols(A, K);
  for (int i = 0; i < nc; ++i) S[i] = 0.f;
  float* pa = S + PH_NSTAT + A; float* pc = pa + (size_t)(A + 1) * K + A;
  for (int c = 0; c < A; ++c) { S[3] += 0.5f + 0.5f * LOG2PI + logf(h->std[c]); S[PH_NSTAT + c] = -h->entropy_coef / h->std[c]; }
  for (int r = 0; r < B; ++r) {
    const float* ya = h->y_a + (size_t)r * K; const float* yc = h->y_c + (size_t)r * K;
    float mu[HB_MAX_C], v = h->b_v[0], lp = 0.f, kl = 0.f;
    for (int k = 0; k < K; ++k) v = fmaf(yc[k], h->w_v[k], v);
    for (int c = 0; c < A; ++c) {
      float m = h->b_mu[c]; for (int k = 0; k < K; ++k) m = fmaf(ya[k], h->w_mu[(size_t)c * K + k], m);
      mu[c] = m;
      const float sg = h->std[c], d = h->actions[(size_t)r * A + c] - m, so = h->old_sigma[(size_t)r * A + c], dm = h->old_mu[(size_t)r * A + c] - m, isg2 = 1.f / (sg * sg);
      lp += -d * d * (0.5f * isg2) - logf(sg) - 0.5f * LOG2PI; kl += logf(sg / so + 1e-5f) + (so * so + dm * dm) * (0.5f * isg2) - 0.5f;
    }
    const float ad = h->adv[r], ratio = expf(lp - h->old_logp[r]), rc = fminf(fmaxf(ratio, lo), hi), in = (ratio >= lo && ratio <= hi) ? 1.f : 0.f;
    const float s1 = -ad * ratio, s2 = -ad * rc, sur = fmaxf(s1, s2), w = s1 > s2 ? 1.f : (s1 < s2 ? in : 0.5f + 0.5f * in), wr = (!split || r < split) ? w_head : w_tail, g_lp = -ad * w * ratio * wr;
    const float tv = h->old_values[r], rt = h->returns[r], dv = v - tv;
    float vl, gv;
    if (h->use_clipped_value_loss) {
      const float dc = fminf(fmaxf(dv, -h->clip), h->clip), vin = (dv >= -h->clip && dv <= h->clip) ? 1.f : 0.f, vc = tv + dc;
      const float l1 = (v - rt) * (v - rt), l2 = (vc - rt) * (vc - rt); vl = fmaxf(l1, l2);
      const float g1 = 2.f * (v - rt), g2 = 2.f * (vc - rt) * vin; gv = l1 > l2 ? g1 : (l1 < l2 ? g2 : 0.5f * g1 + 0.5f * g2);
    } else { vl = (rt - v) * (rt - v); gv = 2.f * (v - rt); }
    const float gval = h->value_loss_coef * gv * invB;
    S[0] += sur * wr; S[1] += vl * invB; S[2] += kl * invB;
    float gm[HB_MAX_C];
    for (int c = 0; c < A; ++c) { const float sg = h->std[c], d = h->actions[(size_t)r * A + c] - mu[c], isg2 = 1.f / (sg * sg);
      gm[c] = g_lp * d * isg2; S[PH_NSTAT + c] += g_lp * (d * d * isg2 / sg - 1.f / sg); pa[(size_t)(A + 1) * K + c] += gm[c]; }
    for (int k = 0; k < K; ++k) {
      float gx = 0.f; for (int c = 0; c < A; ++c) { gx = fmaf(gm[c], h->w_mu[(size_t)c * K + k], gx); pa[(size_t)c * K + k] = fmaf(gm[c], ya[k], pa[(size_t)c * K + k]); }
      const float oa = gx * (ya[k] > 0.f ? 1.f : ya[k] + 1.f), oc = gval * h->w_v[k] * (yc[k] > 0.f ? 1.f : yc[k] + 1.f);
      h->gz_a[(size_t)r * K + k] = oa; h->gz_c[(size_t)r * K + k] = oc; pa[(size_t)A * K + k] += oa; pc[K + k] += oc; pc[k] = fmaf(gval, yc[k], pc[k]);
    }
    pc[2 * K] += gval;
  }
#else
  PpoHeadsArgs a;
  a.y_a = h->y_a; a.y_c = h->y_c; a.w_mu = h->w_mu; a.b_mu = h->b_mu; a.w_v = h->w_v; a.b_v = h->b_v; a.std_ = h->std; a.actions = h->actions; a.old_mu = h->old_mu;
  a.old_sigma = h->old_sigma; a.old_logp = h->old_logp; a.adv = h->adv; a.old_values = h->old_values; a.returns = h->returns; a.gz_a = h->gz_a; a.gz_c = h->gz_c; a.part = h->partials;
  a.B = B; a.A = A; a.K = K; a.use_clip_v = h->use_clipped_value_loss; a.clip = h->clip; a.vcoef = h->value_loss_coef; a.ecoef = h->entropy_coef;
  a.split = (h->surrogate_split > 0 && h->surrogate_split < B) ? h->surrogate_split : B;
  a.w_head = a.split < B ? 1.f / (float)a.split : 1.f / (float)B; a.w_tail = a.split < B ? 1.f / (float)(B - a.split) : 1.f / (float)B;
  int q, rows, nwg; ppo_heads_shape(B, K, &q, &rows, &nwg);
  hipStream_t st = (hipStream_t)stream;
  if (A <= 4)       hipLaunchKernelGGL(go2nn_ppo_heads_kernel<4>,  dim3(nwg), dim3(256), 0, st, a, q, rows);
  else if (A <= 8)  hipLaunchKernelGGL(go2nn_ppo_heads_kernel<8>,  dim3(nwg), dim3(256), 0, st, a, q, rows);
  else if (A <= 12) hipLaunchKernelGGL(go2nn_ppo_heads_kernel<12>, dim3(nwg), dim3(256), 0, st, a, q, rows);
  else              hipLaunchKernelGGL(go2nn_ppo_heads_kernel<16>, dim3(nwg), dim3(256), 0, st, a, q, rows);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

}  // extern "C"

// ---- ABI 5: the latent normaliser of the CTS networks (go2nn_cts.h) ---------------------------------------------------------------------------------------
extern "C" {

int32_t go2nn_l2norm_backward_rows(int32_t n) {
  if (n <= 0) FAIL(GO2NN_EINVAL, "l2norm backward: n > 0");
#ifdef GO2_EMU
  return 1;
#else
  return cts_rows(n);
#endif
}

int go2nn_latent_concat(const float* z, int32_t n, int32_t L, float* dst_a, int32_t lda, float* dst_b, int32_t ldb, float* inv_norm, void* stream) {
  if (!z || !cts_ok(n, L) || (!dst_a && !dst_b) || (dst_a && lda < L) || (dst_b && ldb < L)) FAIL(GO2NN_EINVAL, "latent concat: L a multiple of 4 with L / 4 a power of two up to 32, pitches >= L");
#ifdef GO2_EMU
  (void)stream;
  for (int r = 0; r < n; ++r) {
    float ss = 0.f;
    for (int c = 0; c < L; ++c) ss += z[(size_t)r * L + c] * z[(size_t)r * L + c];
    const float inv = 1.f / fmaxf(sqrtf(ss), CTS_EPS);
    for (int c = 0; c < L; ++c) { const float o = z[(size_t)r * L + c] * inv; if (dst_a) dst_a[(size_t)r * lda + c] = o; if (dst_b) dst_b[(size_t)r * ldb + c] = o; }
    if (inv_norm) inv_norm[r] = inv;
  }
#else
  const int rpw = 256 / (L >> 2);
  hipLaunchKernelGGL(go2nn_latent_concat_kernel, dim3(cdiv(n, rpw)), dim3(256), 0, (hipStream_t)stream, z, n, L, dst_a, lda, dst_b, ldb, inv_norm);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

int go2nn_l2norm_backward(const float* g, int32_t ldg, const float* zhat, int32_t ldz, const float* inv_norm, float* dz, float* partials, int32_t n, int32_t L, void* stream) {
  if (!g || !zhat || !inv_norm || !dz || !partials || !cts_ok(n, L) || ldg < L || ldz < L) FAIL(GO2NN_EINVAL, "l2norm backward: bad argument");
#ifdef GO2_EMU
  (void)stream;
  for (int c = 0; c < L; ++c) partials[c] = 0.f;
  for (int r = 0; r < n; ++r) {
    float dot = 0.f;
    for (int c = 0; c < L; ++c) dot += g[(size_t)r * ldg + c] * zhat[(size_t)r * ldz + c];
    for (int c = 0; c < L; ++c) { const float o = (g[(size_t)r * ldg + c] - zhat[(size_t)r * ldz + c] * dot) * inv_norm[r]; dz[(size_t)r * L + c] = o; partials[c] += o; }
  }
#else
  hipLaunchKernelGGL(go2nn_cts_bwd_kernel<false>, dim3(cts_rows(n)), dim3(256), 0, (hipStream_t)stream, g, ldg, zhat, ldz, inv_norm, (const float*)nullptr, (const float*)nullptr, dz, partials, n, L, 0.f);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

int go2nn_latent_mse(const float* z_s, const float* z_t, float* dz_s, float* partials, int32_t n, int32_t L, float grad_scale, void* stream) {
  if (!z_s || !z_t || !dz_s || !partials || !cts_ok(n, L)) FAIL(GO2NN_EINVAL, "latent mse: bad argument");
  const float scale = grad_scale * 2.f / ((float)n * (float)L);
#ifdef GO2_EMU
  (void)stream;
  for (int c = 0; c < L + 4; ++c) partials[c] = 0.f;
  float loss = 0.f;
  for (int r = 0; r < n; ++r) {
    float ss = 0.f, st = 0.f, dot = 0.f, gv[128], zv[128];
    for (int c = 0; c < L; ++c) { ss += z_s[(size_t)r * L + c] * z_s[(size_t)r * L + c]; st += z_t[(size_t)r * L + c] * z_t[(size_t)r * L + c]; }
    const float is = 1.f / fmaxf(sqrtf(ss), CTS_EPS), it = 1.f / fmaxf(sqrtf(st), CTS_EPS);
    for (int c = 0; c < L; ++c) { zv[c] = z_s[(size_t)r * L + c] * is; const float d = z_t[(size_t)r * L + c] * it - zv[c]; loss += d * d; gv[c] = -scale * d; dot += gv[c] * zv[c]; }
    for (int c = 0; c < L; ++c) { const float o = (gv[c] - zv[c] * dot) * is; dz_s[(size_t)r * L + c] = o; partials[4 + c] += o; }
  }
  partials[0] = loss / ((float)n * (float)L);
#else
  hipLaunchKernelGGL(go2nn_cts_bwd_kernel<true>, dim3(cts_rows(n)), dim3(256), 0, (hipStream_t)stream, (const float*)nullptr, 0, (const float*)nullptr, 0, (const float*)nullptr, z_s, z_t, dz_s, partials, n, L, scale);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

int go2nn_moe_usage(const float* logits, float* partials, int32_t n, int32_t E, void* stream) {
  if (!logits || !partials || n <= 0 || E < 1 || E > 16) FAIL(GO2NN_EINVAL, "moe usage: 1 <= E <= 16");
#ifdef GO2_EMU
  (void)stream;
  for (int e = 0; e < E; ++e) partials[e] = 0.f;
  for (int r = 0; r < n; ++r) {
    float mx = -3.4e38f, sum = 0.f, w[16];
    for (int e = 0; e < E; ++e) mx = fmaxf(mx, logits[(size_t)r * E + e]);
    for (int e = 0; e < E; ++e) { w[e] = expf(logits[(size_t)r * E + e] - mx); sum += w[e]; }
    for (int e = 0; e < E; ++e) partials[e] += w[e] / sum;
  }
#else
  hipLaunchKernelGGL(go2nn_moe_usage_kernel, dim3(cts_rows(n)), dim3(CTS_ROWS_PER_WG), 0, (hipStream_t)stream, logits, partials, n, E);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

int go2nn_moe_mix_loss(const float* logits, const float* outs, const float* t_hat, const float* usage_sum, float* d_logits, float* d_outs, float* partials,
                       int32_t n, int32_t E, int32_t L, float lb_coef, int32_t expert_major, const float* bias, float* dbias_partials, void* stream) {
  if (!logits || !outs || !t_hat || !usage_sum || !d_logits || !d_outs || !partials || !cts_ok(n, L) || E < 1 || E > 16 || (dbias_partials && !bias)) FAIL(GO2NN_EINVAL, "moe mix loss: bad argument (1 <= E <= 16)");
  const long long sr = expert_major ? L : (long long)E * L, se = expert_major ? (long long)n * L : L;          // [E, n, L] (the batched GEMM's own output) or [n, E, L]
#ifdef GO2_EMU
  (void)stream;
  const float invE = 1.f / (float)E, invn = 1.f / (float)n, scale = 2.f / ((float)n * (float)L);
  float loss = 0.f, lb = 0.f;
  for (int e = 0; e < E; ++e) { const float u = usage_sum[e] * invn - invE; lb += u * u; }
  lb *= invE;
  for (int r = 0; r < n; ++r) {
    float w[16], y[128], sh[128], dy[128], dw[16], mx = -3.4e38f, sum = 0.f, ss = 0.f, dot = 0.f, wd = 0.f;
    for (int e = 0; e < E; ++e) mx = fmaxf(mx, logits[(size_t)r * E + e]);
    for (int e = 0; e < E; ++e) { w[e] = expf(logits[(size_t)r * E + e] - mx); sum += w[e]; }
    for (int e = 0; e < E; ++e) w[e] /= sum;
    for (int c = 0; c < L; ++c) { y[c] = 0.f; for (int e = 0; e < E; ++e) y[c] = fmaf(w[e], outs[(size_t)r * sr + (size_t)e * se + c] + (bias ? bias[(size_t)e * L + c] : 0.f), y[c]); ss += y[c] * y[c]; }
    const float inv = 1.f / fmaxf(sqrtf(ss), CTS_EPS);
    for (int c = 0; c < L; ++c) { sh[c] = y[c] * inv; const float d = t_hat[(size_t)r * L + c] - sh[c]; loss += d * d; dy[c] = -scale * d; dot += dy[c] * sh[c]; }
    for (int c = 0; c < L; ++c) dy[c] = (dy[c] - sh[c] * dot) * inv;
    for (int e = 0; e < E; ++e) {
      float s = 0.f;
      for (int c = 0; c < L; ++c) { d_outs[(size_t)r * sr + (size_t)e * se + c] = w[e] * dy[c]; s += dy[c] * (outs[(size_t)r * sr + (size_t)e * se + c] + (bias ? bias[(size_t)e * L + c] : 0.f));
        if (dbias_partials) dbias_partials[(size_t)e * L + c] = (r ? dbias_partials[(size_t)e * L + c] : 0.f) + w[e] * dy[c]; }
      dw[e] = s + lb_coef * 2.f * (usage_sum[e] * invn - invE) * invE * invn; wd = fmaf(w[e], dw[e], wd);
    }
    for (int e = 0; e < E; ++e) d_logits[(size_t)r * E + e] = w[e] * (dw[e] - wd);
  }
  partials[0] = loss / ((float)n * (float)L); partials[1] = lb; partials[2] = partials[3] = 0.f;
#else
  if (E <= 8) hipLaunchKernelGGL(go2nn_moe_mix_kernel<8>, dim3(cts_rows(n)), dim3(256), 0, (hipStream_t)stream, logits, outs, t_hat, usage_sum, d_logits, d_outs, partials, n, E, L, lb_coef, sr, se, bias, dbias_partials);
  else        hipLaunchKernelGGL(go2nn_moe_mix_kernel<16>, dim3(cts_rows(n)), dim3(256), 0, (hipStream_t)stream, logits, outs, t_hat, usage_sum, d_logits, d_outs, partials, n, E, L, lb_coef, sr, se, bias, dbias_partials);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

int go2nn_moe_mix_forward(const float* logits, const float* outs, const float* bias, const int32_t* rows, float* z, int32_t ldz, int32_t n, int32_t E, int32_t L,
                          int32_t expert_major, void* stream) {
  if (!logits || !outs || !z || !cts_ok(n, L) || E < 1 || E > 16 || ldz < L || (ldz & 3) || ((uintptr_t)z & 15)) FAIL(GO2NN_EINVAL, "moe mix forward: bad argument (1 <= E <= 16, ldz >= L a multiple of 4)");
  const long long sr = expert_major ? L : (long long)E * L, se = expert_major ? (long long)n * L : L;
#ifdef GO2_EMU
  (void)stream;
  for (int r = 0; r < n; ++r) {
    float w[16], y[128], mx = -3.4e38f, sum = 0.f, ss = 0.f;
    for (int e = 0; e < E; ++e) mx = fmaxf(mx, logits[(size_t)r * E + e]);
    for (int e = 0; e < E; ++e) { w[e] = expf(logits[(size_t)r * E + e] - mx); sum += w[e]; }
    for (int e = 0; e < E; ++e) w[e] /= sum;
    for (int c = 0; c < L; ++c) { y[c] = 0.f; for (int e = 0; e < E; ++e) y[c] = fmaf(w[e], outs[(size_t)r * sr + (size_t)e * se + c] + (bias ? bias[(size_t)e * L + c] : 0.f), y[c]); ss += y[c] * y[c]; }
    const float inv = 1.f / fmaxf(sqrtf(ss), CTS_EPS);
    float* o = z + (size_t)(rows ? rows[r] : r) * ldz;
    for (int c = 0; c < L; ++c) o[c] = y[c] * inv;
  }
#else
  if (E <= 8) hipLaunchKernelGGL(go2nn_moe_mix_forward_kernel<8>, dim3(cts_rows(n)), dim3(256), 0, (hipStream_t)stream, logits, outs, bias, rows, z, ldz, n, E, L, sr, se);
  else        hipLaunchKernelGGL(go2nn_moe_mix_forward_kernel<16>, dim3(cts_rows(n)), dim3(256), 0, (hipStream_t)stream, logits, outs, bias, rows, z, ldz, n, E, L, sr, se);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

}  // extern "C"
