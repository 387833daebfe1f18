/* go2sim_rng.h — the random-number contract shared by the HIP library and the oracle (data, not algorithm).
 *
 * Uniform `slot` (GO2_U_* in go2sim.h) of env `e` at Philox step `t` is word `w` of
 *     Philox4x32-10(key = seed, counter = {global env id, group, t_lo, t_hi})
 * with (group, w) = GO2_SLOT_CODE[slot] >> 2, & 3.  Slots that one (env, leg) lane consumes together share a group, so
 * the kernels need one Philox call per group instead of one per slot:
 *   group 0        DELAY
 *   group 1, 2     RSA  x,y,yaw,prob | comb,ang,dir
 *   group 3+4l+g   RESET per-DOF values of leg l: index 3k+j (k = strength, offset, kp, kd, dof; j = joint of the leg) -> g = idx>>2
 *   group 19       RESET terrain, yaw, xy(2)      group 20, 21   RESET vel(6)
 *   group 22, 23   RSB                            group 24, 25   PUSH(5)
 *   group 26,27,28 NOISE ang_vel(3) | gravity(3) | commands(3)
 *   group 29+l, 33+l, 37+l   NOISE dof_pos / dof_vel / actions of leg l (3 each)
 *   group 41       TURN category, backflip height, sideflip height, side sign (init_state.turn_over)
 *   group 42       padding slot
 */
#ifndef GO2SIM_RNG_H
#define GO2SIM_RNG_H
#include <stdint.h>
#include "go2sim.h"

static inline void go2_fill_slot_codes(uint8_t* code /*[GO2_NUM_UNIFORMS]*/) {
  int s, k, d;
  for (s = 0; s < GO2_NUM_UNIFORMS; ++s) code[s] = (uint8_t)(42 * 4);
  for (s = 0; s < 4; ++s) code[GO2_U_TURN + s] = (uint8_t)(41 * 4 + s);
  code[GO2_U_DELAY] = 0;
  for (s = 0; s < 4; ++s) { code[GO2_U_RSA + s] = (uint8_t)(1 * 4 + s); code[GO2_U_RSB + s] = (uint8_t)(22 * 4 + s); }
  for (s = 0; s < 3; ++s) { code[GO2_U_RSA + 4 + s] = (uint8_t)(2 * 4 + s); code[GO2_U_RSB + 4 + s] = (uint8_t)(23 * 4 + s); }
  {
    const int base[5] = {GO2_U_RESET_STRENGTH, GO2_U_RESET_OFFSET, GO2_U_RESET_KP, GO2_U_RESET_KD, GO2_U_RESET_DOF};
    for (k = 0; k < 5; ++k) for (d = 0; d < 12; ++d) { int l = d / 3, j = d % 3, idx = 3 * k + j; code[base[k] + d] = (uint8_t)((3 + 4 * l + (idx >> 2)) * 4 + (idx & 3)); }
  }
  code[GO2_U_RESET_TERRAIN] = 19 * 4 + 0; code[GO2_U_RESET_YAW] = 19 * 4 + 1; code[GO2_U_RESET_XY] = 19 * 4 + 2; code[GO2_U_RESET_XY + 1] = 19 * 4 + 3;
  for (s = 0; s < 6; ++s) code[GO2_U_RESET_VEL + s] = (uint8_t)((20 + (s >> 2)) * 4 + (s & 3));
  for (s = 0; s < 5; ++s) code[GO2_U_PUSH + s] = (uint8_t)((24 + (s >> 2)) * 4 + (s & 3));
  for (s = 0; s < 3; ++s) { code[GO2_U_NOISE + s] = (uint8_t)(26 * 4 + s); code[GO2_U_NOISE + 3 + s] = (uint8_t)(27 * 4 + s); code[GO2_U_NOISE + 6 + s] = (uint8_t)(28 * 4 + s); }
  for (d = 0; d < 12; ++d) { int l = d / 3, j = d % 3;
    code[GO2_U_NOISE + 9 + d] = (uint8_t)((29 + l) * 4 + j); code[GO2_U_NOISE + 21 + d] = (uint8_t)((33 + l) * 4 + j); code[GO2_U_NOISE + 33 + d] = (uint8_t)((37 + l) * 4 + j); }
}
#endif
