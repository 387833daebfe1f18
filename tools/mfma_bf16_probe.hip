// bf16-MFMA issue-rate probe (v_mfma_f32_32x32x16_bf16): W waves per SIMD, A independent accumulators, optionally three ds_read_b128 per six MFMAs (the policy
// kernel's k-block).   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/mfma_bf16_probe tools/mfma_bf16_probe.hip && /tmp/mfma_bf16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int A, bool LDS>
__global__ void probe(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[32 * 1040 * 3];
  for (int i = threadIdx.x; i < (int)sizeof(lds) / 4; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = 0x3f803f80u;
  __syncthreads();
  f32x16 acc[A];
  for (int j = 0; j < A; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const int lane = threadIdx.x & 63;
  const unsigned char* arow = lds + (lane & 31) * 1040 + (lane >> 5) * 16;
  u32x4 a[3] = {u32x4{1, 2, 3, 4}, u32x4{5, 6, 7, 8}, u32x4{9, 10, 11, 12}}, b = {threadIdx.x, 2, 3, 4};
  for (int it = 0; it < iters; ++it) {
    u32x4 an[3];
    if (LDS) {
#pragma unroll
      for (int p = 0; p < 3; ++p) an[p] = *reinterpret_cast<const u32x4*>(arow + p * 32 * 1040 + (it & 31) * 32);
    }
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int j = 0; j < A; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[q % 3]), __builtin_bit_cast(bf16x8, b), acc[j], 0, 0, 0);
    if (LDS) {
#pragma unroll
      for (int p = 0; p < 3; ++p) a[p] = an[p];
    }
  }
  float s = 0.f;
  for (int j = 0; j < A; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int A, bool LDS>
void run(int threads, int iters) {
  float* out; (void)hipMalloc(&out, 256 * 1024 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<A, LDS><<<256, threads>>>(out, 10); hipDeviceSynchronize();
  hipEventRecord(e0); probe<A, LDS><<<256, threads>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 6 * A * (threads / 256.0);          // MFMAs per SIMD
  printf("waves/SIMD %d, %d accumulators%s: %.3f ms -> %.0f TFLOP/s, %.1f ns per MFMA per SIMD (%.1f cycles at 2.4 GHz)\n", threads / 256, A, LDS ? ", 3 ds_read_b128 per 6 MFMAs" : "",
         ms, 256.0 * 4 * n * 32768 / ms / 1e9, ms * 1e6 / n, ms * 1e-3 * 2.4e9 / n);
  hipFree(out);
}
int main() {
  run<1, false>(256, 40000); run<2, false>(256, 20000); run<1, false>(512, 40000); run<2, false>(512, 20000);
  run<1, true>(256, 40000); run<1, true>(512, 40000); run<2, true>(512, 20000);
  run<1, false>(256, 100);      // a short chain: 600 MFMAs per wave (launch + drain visible)
  return 0;
}
