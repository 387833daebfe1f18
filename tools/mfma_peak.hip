// fp32-MFMA issue-rate probe: what does v_mfma_f32_32x32x2_f32 sustain on this chip, with W waves per SIMD and A independent accumulators?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/mfma_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int A>
__global__ void probe(float* out, int iters) {
  f32x16 acc[A];
  for (int j = 0; j < A; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < A; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < A; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int A>
void run(int threads, int iters) {
  float* out; hipMalloc(&out, 256 * 1024 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<A><<<256, threads>>>(out, 10); hipDeviceSynchronize();
  hipEventRecord(e0); probe<A><<<256, threads>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = 256.0 * (threads / 64) * (double)iters * A * 2 * 32 * 32 * 2;
  const double cyc_per_mfma_at_2p4 = ms * 1e-3 * 2.4e9 / ((double)iters * A * (threads / 256.0));
  printf("waves/SIMD %d, %d accumulators: %.3f ms -> %.1f TFLOP/s (fp32 MFMA), %.1f cycles per MFMA per SIMD if the clock were 2.4 GHz\n", threads / 256, A, ms, flop / ms / 1e9, cyc_per_mfma_at_2p4);
  hipFree(out);
}
int main() {
  run<1>(256, 200000); run<4>(256, 50000); run<4>(512, 50000); run<4>(1024, 25000); run<4>(256, 400000);
  return 0;
}
