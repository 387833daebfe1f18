// Device check of the cross-lane primitives (go2_xlane.h) against the lane arithmetic the host emulation implements.
//   hipcc --offload-arch=gfx950 -O3 -o build/xlane_check tools/xlane_check.hip && build/xlane_check
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../go2_rl_gym_amd/csrc/go2_xlane.h"
__global__ void k(float* o) {
  const int l = threadIdx.x; const float x = (float)l;
  o[0 * 64 + l] = xl::quad_perm<1, 0, 3, 2>(x);
  o[1 * 64 + l] = xl::quad_perm<2, 3, 0, 1>(x);
  o[2 * 64 + l] = xl::row_ror<4>(x);
  o[3 * 64 + l] = xl::row_ror<8>(x);
  o[4 * 64 + l] = xl::row_shr<4>(x);
  o[5 * 64 + l] = xl::row_bcast<5>(x);
  o[6 * 64 + l] = xl::sub_sum(x);
  o[7 * 64 + l] = xl::leg_sum(x);
  o[8 * 64 + l] = xl::sub_bcast<2>(x);
  o[9 * 64 + l] = (float)xl::quad_perm_i<1, 0, 3, 2>(l * 7);
  o[10 * 64 + l] = xl::sub_min(100.f - x);
  o[11 * 64 + l] = xl::any(l == 37) ? 1.f : 0.f;
  float s, c; s = __builtin_amdgcn_sinf(x * 0.01f * 0.15915494309189535f); c = __builtin_amdgcn_cosf(x * 0.01f * 0.15915494309189535f);
  o[12 * 64 + l] = s; o[13 * 64 + l] = c;
}
int main() {
  float* d; hipMalloc(&d, 14 * 64 * 4); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); float h[14 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const int q = l & ~3, r = l & ~15, r16 = l & 15;
    const int p1[4] = {1, 0, 3, 2}, p2[4] = {2, 3, 0, 1};
    float want[14];
    want[0] = q | p1[l & 3]; want[1] = q | p2[l & 3]; want[2] = r | ((r16 - 4) & 15); want[3] = r | ((r16 - 8) & 15);
    want[4] = r16 >= 4 ? l - 4 : 0; want[5] = r | 5; want[6] = 4 * q + 6; want[7] = 4 * (r + (l & 3)) + 24; want[8] = q | 2; want[9] = 7 * (q | p1[l & 3]);
    want[10] = 100.f - (q + 3); want[11] = 1.f; want[12] = sinf(l * 0.01f); want[13] = cosf(l * 0.01f);
    for (int k_ = 0; k_ < 14; ++k_) if (fabsf(h[k_ * 64 + l] - want[k_]) > (k_ >= 12 ? 2e-6f : 0.f)) { ++bad; printf("MISMATCH prim %d lane %d: got %g want %g\n", k_, l, h[k_ * 64 + l], want[k_]); }
  }
  printf("xlane_check: %d mismatches\n", bad);
  return bad != 0;
}
