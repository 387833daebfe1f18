// go2_xlane.h — the cross-lane vocabulary of the lane programs, and (host builds only) a fibre runtime that executes the SAME kernel body
// lane by lane so that the device code can be checked against the oracle on a machine without a GPU.
//
// Lane layout of the step kernel (go2sim_impl.cpp): a 256-thread workgroup = 4 waves = 16 environments; one DPP row (16 lanes) = one
// environment:   row lane r = leg * 4 + sub,   leg = r >> 2 (FL, FR, RL, RR),   sub = r & 3.
// The four lanes of a quad are the four SUB-lanes of one leg (they slice the leg's contact rows, split its collision candidates and
// its share of the post-physics work); the four legs of an environment sit 4 lanes apart inside the row.
//
//   device: every primitive is ONE v_*_dpp instruction (the compiler folds the move into the consuming add / min where it can)
//   host  : a rendezvous of the live lanes of the wave (go2_fiber runtime below), then a read of the source lane's slot
//
// xl::quad_perm<a,b,c,d>(x)   x of quad lane {a,b,c,d}[lane & 3]                      (sub-lane exchange inside a leg)
// xl::row_ror<n>(x)           x of row lane (r - n) mod 16                             (n = 4, 8, 12: the same sub-lane of another leg)
// xl::row_shr<n>(x)           x of row lane r - n, 0.0f for r < n
// xl::row_bcast<k>(x)         x of row lane k                                          (row_newbcast)
// xl::sub_sum(x)              sum over the 4 sub-lanes of a leg, bit-identical in the 4 lanes (commutative pairing)
// xl::leg_sum(x)              sum over the 4 legs (same sub-lane), bit-identical in the 4 lanes
// xl::any(p)                  true iff p holds in any live lane of the WAVE
// xl::sync()                  workgroup barrier
#pragma once
#include <stdint.h>

#if defined(__HIP_DEVICE_COMPILE__)
// ------------------------------------------------------------------------------------------------ device
namespace xl {
#ifdef GO2_DBG_NOINLINE_DPP
#define GO2_DPP_INLINE __attribute__((noinline))
#else
#define GO2_DPP_INLINE __forceinline__
#endif
template <int CTRL>
__device__ GO2_DPP_INLINE float dpp(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true)); }
template <int CTRL>
__device__ GO2_DPP_INLINE int dpp_i(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true); }
template <int A, int B, int C_, int D>
__device__ __forceinline__ float quad_perm(float x) { return dpp<A | (B << 2) | (C_ << 4) | (D << 6)>(x); }
template <int A, int B, int C_, int D>
__device__ __forceinline__ int quad_perm_i(int x) { return dpp_i<A | (B << 2) | (C_ << 4) | (D << 6)>(x); }
template <int N>
__device__ __forceinline__ float row_ror(float x) { return dpp<0x120 + N>(x); }
template <int N>
__device__ __forceinline__ float row_shr(float x) { return dpp<0x110 + N>(x); }
template <int K>
__device__ __forceinline__ float row_bcast(float x) { return dpp<0x150 + K>(x); }
template <int K>
__device__ __forceinline__ int row_bcast_i(int x) { return dpp_i<0x150 + K>(x); }
#ifdef GO2_DBG_NOSKIP
__device__ __forceinline__ bool any(bool p) { return true; }
#else
__device__ __forceinline__ bool any(bool p) { return __any(p); }
#endif
__device__ __forceinline__ void sync() { __syncthreads(); }
// LDS hand-off between the lanes of ONE wave (a row of 16 lanes in particular): the LDS executes a wave's accesses in order, so a write by
// one lane is visible to a later read by another lane of the same wave; the fence only stops the compiler from moving the two
__device__ __forceinline__ void row_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
}  // namespace xl
#elif defined(__HIPCC__)
// ------------------------------------------------------------------------------------------------ host pass of a HIP compilation
// (the kernel body is a __host__ __device__ function, so its host instantiation must parse; it is never executed in the device library)
namespace xl {
template <int A, int B, int C_, int D> __host__ __device__ inline float quad_perm(float x) { return x; }
template <int A, int B, int C_, int D> __host__ __device__ inline int quad_perm_i(int x) { return x; }
template <int N> __host__ __device__ inline float row_ror(float x) { return x; }
template <int N> __host__ __device__ inline float row_shr(float x) { return x; }
template <int K> __host__ __device__ inline float row_bcast(float x) { return x; }
template <int K> __host__ __device__ inline int row_bcast_i(int x) { return x; }
__host__ __device__ inline bool any(bool p) { return p; }
__host__ __device__ inline void sync() {}
__host__ __device__ inline void row_sync() {}
__host__ __device__ inline int lane_id() { return 0; }
}  // namespace xl
#else
// ------------------------------------------------------------------------------------------------ host: fibres
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#if !defined(__x86_64__)
#include <ucontext.h>
#endif

namespace xl {
struct Fiber;
struct Group {                       // one workgroup under emulation
  int nthreads, live;                // live = fibres that have not returned yet
  int barrier_arrived; unsigned barrier_gen;
  // rendezvous scopes: the 16-lane DPP row (quad / row primitives: legal wherever whole rows are convergent, e.g. inside a per-env
  // branch) and the wave (xl::any)
  struct Scope { uint32_t slot[2][64]; int arrived, live; unsigned gen; unsigned opid[2]; };
  Scope wave[16], row[64];
  std::vector<Fiber*> fibers;
};
struct Fiber {
  void* sp; char* stack; bool done; int tid; Group* g;
  void (*fn)(void*, int); void* arg;
#if !defined(__x86_64__)
  ucontext_t ctx;
#endif
};
struct Sched { void* main_sp; Fiber* cur;
#if !defined(__x86_64__)
  ucontext_t main_ctx;
#endif
};
inline Sched& sched() { static thread_local Sched s; return s; }

#if defined(__x86_64__)
// minimal System V context switch: callee-saved registers + stack pointer (the emulation never touches signal masks or FP control words)
extern "C" void go2_fiber_switch(void** save_sp, void* load_sp);
#ifdef GO2_XLANE_IMPLEMENTATION
__asm__(".text\n.p2align 4\n.globl go2_fiber_switch\n.hidden go2_fiber_switch\n.type go2_fiber_switch,@function\n"
        "go2_fiber_switch:\n"
        "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
        "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
        "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n"
        ".size go2_fiber_switch, .-go2_fiber_switch\n");
#endif
inline void to_main() { Sched& s = sched(); Fiber* f = s.cur; go2_fiber_switch(&f->sp, s.main_sp); }
inline void to_fiber(Fiber* f) { Sched& s = sched(); s.cur = f; go2_fiber_switch(&s.main_sp, f->sp); }
#else
inline void to_main() { Sched& s = sched(); swapcontext(&s.cur->ctx, &s.main_ctx); }
inline void to_fiber(Fiber* f) { Sched& s = sched(); s.cur = f; swapcontext(&s.main_ctx, &f->ctx); }
#endif
inline void yield() { to_main(); }

inline void fiber_entry() {
  Fiber* f = sched().cur;
  f->fn(f->arg, f->tid);
  f->done = true; f->g->live--; f->g->wave[f->tid >> 6].live--; f->g->row[f->tid >> 4].live--;
  to_main();
  abort();      // a finished fibre is never resumed
}

// Run fn(arg, tid) for tid = 0..nthreads-1 as cooperatively scheduled fibres (round robin; a fibre runs until it reaches a cross-lane
// primitive, a barrier, or returns).  Lock-step is NOT simulated — only the rendezvous points are, which is all the lane programs rely on.
inline void run_group(int nthreads, void (*fn)(void*, int), void* arg) {
  static thread_local std::vector<char*> stacks;
  const size_t STACK = 256 * 1024;
  while ((int)stacks.size() < nthreads) stacks.push_back((char*)malloc(STACK));
  Group g; memset(&g.barrier_arrived, 0, sizeof(int)); g.nthreads = g.live = nthreads; g.barrier_arrived = 0; g.barrier_gen = 0;
  memset(g.wave, 0, sizeof(g.wave)); memset(g.row, 0, sizeof(g.row));
  std::vector<Fiber> fb(nthreads);
  for (int t = 0; t < nthreads; ++t) {
    Fiber& f = fb[t]; f.stack = stacks[t]; f.done = false; f.tid = t; f.g = &g; f.fn = fn; f.arg = arg; g.wave[t >> 6].live++; g.row[t >> 4].live++;
#if defined(__x86_64__)
    uintptr_t top = ((uintptr_t)f.stack + STACK) & ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;                        // fake return address of fiber_entry: entry sees rsp = 8 mod 16, as after a call
    *--sp = (void*)&fiber_entry;            // consumed by the `ret` of go2_fiber_switch
    for (int i = 0; i < 6; ++i) *--sp = nullptr;
    f.sp = sp;
#else
    getcontext(&f.ctx); f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = STACK; f.ctx.uc_link = nullptr;
    makecontext(&f.ctx, (void (*)())fiber_entry, 0);
#endif
    g.fibers.push_back(&f);
  }
  while (g.live > 0)
    for (int t = 0; t < nthreads; ++t) if (!fb[t].done) to_fiber(&fb[t]);
}

inline Fiber* self() { return sched().cur; }
inline int lane_id() { return self()->tid & 63; }

// deposit `bits`, wait until every live lane of the scope (row: wide = false, wave: wide = true) has deposited for this operation,
// return the slot array of the operation indexed by lane-in-wave
inline const uint32_t* exchange(uint32_t bits, unsigned opid, bool wide = false) {
  Fiber* f = self(); Group::Scope& w = wide ? f->g->wave[f->tid >> 6] : f->g->row[f->tid >> 4];
  const unsigned gen = w.gen; const int par = gen & 1;
  if (w.arrived == 0) w.opid[par] = opid;
  else if (w.opid[par] != opid) { fprintf(stderr, "go2_xlane: divergent cross-lane operation (lane %d: op %u vs %u)\n", f->tid, opid, w.opid[par]); abort(); }
  w.slot[par][f->tid & 63] = bits;
  ++w.arrived;
  while (w.gen == gen) {
    if (w.arrived >= w.live) { w.arrived = 0; w.gen = gen + 1; break; }     // everyone still alive has arrived (lanes may have exited meanwhile): next generation
    yield();
  }
  return w.slot[par];
}
inline float f_of(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }
inline uint32_t u_of(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }

template <int A, int B, int C_, int D>
inline float quad_perm(float x) { const int l = lane_id(), p[4] = {A, B, C_, D}; return f_of(exchange(u_of(x), 1)[(l & ~3) | p[l & 3]]); }
template <int A, int B, int C_, int D>
inline int quad_perm_i(int x) { const int l = lane_id(), p[4] = {A, B, C_, D}; return (int)exchange((uint32_t)x, 2)[(l & ~3) | p[l & 3]]; }
template <int N>
inline float row_ror(float x) { const int l = lane_id(); return f_of(exchange(u_of(x), 3)[(l & ~15) | ((l - N) & 15)]); }
template <int N>
inline float row_shr(float x) { const int l = lane_id(); const uint32_t* s = exchange(u_of(x), 4); return (l & 15) >= N ? f_of(s[l - N]) : 0.f; }
template <int K>
inline float row_bcast(float x) { const int l = lane_id(); return f_of(exchange(u_of(x), 5)[(l & ~15) | K]); }
template <int K>
inline int row_bcast_i(int x) { const int l = lane_id(); return (int)exchange((uint32_t)x, 6)[(l & ~15) | K]; }
inline bool any(bool p) {
  Fiber* f = self();
  const uint32_t* s = exchange(p ? 1u : 0u, 7, true);
  bool r = false;
  for (int t = 0; t < 64; ++t) { const int tid = (f->tid & ~63) | t; if (tid < f->g->nthreads && !f->g->fibers[tid]->done && s[t]) r = true; }
  return r;
}
inline void row_sync() { (void)exchange(0u, 8); }
inline void sync() {
  Fiber* f = self(); Group* g = f->g;
  const unsigned gen = g->barrier_gen;
  ++g->barrier_arrived;
  while (g->barrier_gen == gen) {
    if (g->barrier_arrived >= g->live) { g->barrier_arrived = 0; g->barrier_gen = gen + 1; break; }
    yield();
  }
}
}  // namespace xl
#endif

// ---- compositions (same code on both builds) ------------------------------------------------------------------------------------
#if defined(__HIPCC__)
#define GO2_XL __host__ __device__ __forceinline__
#else
#define GO2_XL inline
#endif
namespace xl {
// sum over the quad (the 4 sub-lanes of a leg): both adds pair commutatively, so the 4 lanes end with bit-identical sums
GO2_XL float sub_sum(float x) { const float y = x + quad_perm<1, 0, 3, 2>(x); return y + quad_perm<2, 3, 0, 1>(y); }
// sum over the 4 legs of an environment (lanes 4 apart in the row); same pairing argument (y has period 8 after the first step)
GO2_XL float leg_sum(float x) { const float y = x + row_ror<8>(x); return y + row_ror<4>(y); }
GO2_XL float sub_min(float x) { const float y = fminf(x, quad_perm<1, 0, 3, 2>(x)); return fminf(y, quad_perm<2, 3, 0, 1>(y)); }
template <int S>
GO2_XL float sub_bcast(float x) { return quad_perm<S, S, S, S>(x); }
}  // namespace xl
