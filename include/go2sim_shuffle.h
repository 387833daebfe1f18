/* go2sim_shuffle.h — the keyed permutation of go2sim_shuffle_gather (include/go2sim.h), stated once for the HIP kernel, the host build and the oracle.
 * A Feistel network is a bijection of 2h-bit numbers whatever its round function; cycle walking (re-encrypt until the value falls below n) restricts it to a
 * bijection of [0, n).  2^(2h) < 4 n, so a walk takes < 4 encryptions on average.  Round function: the lowbias32 integer hash of (half, round key). */
#ifndef GO2SIM_SHUFFLE_H
#define GO2SIM_SHUFFLE_H
#include <stdint.h>
#ifndef GO2_SHUFFLE_FN
#define GO2_SHUFFLE_FN static inline
#endif
GO2_SHUFFLE_FN uint32_t go2_mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
GO2_SHUFFLE_FN int go2_shuffle_half_bits(uint32_t n) { int h = 1; while (h < 16 && (1u << (2 * h)) < n) ++h; return h; }
GO2_SHUFFLE_FN uint32_t go2_shuffle_index(uint32_t i, uint32_t n, int h, uint32_t seed, uint32_t counter) {
  const uint32_t mask = (1u << h) - 1u;
  uint32_t x = i;
  do {
    uint32_t L = x >> h, R = x & mask;
    for (uint32_t r = 0; r < 6u; ++r) {
      const uint32_t F = go2_mix32(R ^ go2_mix32(seed + 0x9e3779b9U * (r + 1u)) ^ (counter * 0x85ebca6bU + r)) & mask;
      const uint32_t t = L ^ F; L = R; R = t;
    }
    x = (L << h) | R;
  } while (x >= n);
  return x;
}
#endif
