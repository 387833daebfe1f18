// go2_post.h — the per-lane post-physics program (one lane = one (env, leg)).
//
// Restates, fused into one pass per env, LeggedRobot.post_physics_step
// (legged_gym/envs/base/legged_robot.py:102-142): counters, derived base quantities (:119-125),
// _post_physics_step_callback (:404-421) with _resample_commands (:423-592) and _get_heights (:1188-1224),
// check_termination (:170-178), compute_reward (:247-274) with every _reward_* (:1228-1441, go2_env.py:55-68),
// reset_idx (:180-245; _reset_dofs :620, _reset_root_states :635, _update_terrain_curriculum :1143),
// _push_robots (:709-724), Go2Robot.compute_observations (go2_env.py:23-53) and the obs clip of step()
// (:96-99), last_* updates (:140-142).  Ordering quirks of the reference are kept.
//
// Work split inside the 16 lanes of an environment (row lane = leg * 4 + sub, go2_xlane.h): every per-joint quantity (3 of the 12 DOFs)
// and the leg's 4 bodies belong to the leg's quad (computed by its 4 sub-lanes alike, written by sub-lane 0); a sixteenth of the 187
// height samples belongs to the lane; per-env scalar logic (commands, termination, root reset) is computed redundantly by the 16 lanes
// (identical inputs, identical results) and written by lane 0.  Cross-lane steps: the leg sum of GO2_POST_PARTIALS reward partial sums
// between postA and postB.  No location that several lanes read is written before that sum (a rendezvous of the whole row).
#pragma once
#include "go2_math.h"
#include "go2_tables.h"
#include "go2_xlane.h"

#define GO2_POST_PARTIALS 24

// field-major (SoA) addressing, see Go2Ptrs
#define F1D(p, e) (p)[(e)]
#define F2D(p, a, e) (p)[(size_t)(a) * N + (e)]
#define F3D(p, A, a, b, e) (p)[((size_t)(b) * (A) + (a)) * N + (e)]

struct PhysOut {
  V3 pw; float qx, qy, qz, qw; V3 vw, ww;
  float q[3], qd[3], tau[3];
  V3 Fhip, Fthigh, Fcalf, Ffoot, Fbase;
  V3 foot_pos, foot_vel;
};

GO2_HD V3 quat_rotate_inverse(float qx, float qy, float qz, float qw, V3 v) {
  V3 qv = v3(qx, qy, qz); V3 c = cross(qv, v); float d = dot(qv, v), a = 2 * qw * qw - 1;
  return a * v - (2 * qw) * c + (2 * d) * qv;
}
GO2_HD V3 quat_apply(float qx, float qy, float qz, float qw, V3 v) {
  V3 qv = v3(qx, qy, qz); V3 c = cross(qv, v); V3 cc = cross(qv, c);
  return v + (2 * qw) * c + 2.0f * cc;
}

// The scalars the reference keeps as Python floats are pure functions of common_step_counter // 24:
// reward curriculum (legged_robot.py:144-168), command-range curriculum (:433-446), zero-command
// probability (:556-557).  `csc` = the counter value this step runs under.
GO2_HD float go2_current_scale(const float* c4, float it) {   // get_current_scale (:154-168)
  float pct = (it - c4[0]) / (c4[1] - c4[0]); pct = fminf(fmaxf(pct, 0.f), 1.f);
  return (1.f - pct) * c4[2] + pct * c4[3];
}
// command_range_curriculum (legged_robot.py:433-446): the entry with the largest `iter` that has passed, -1 before the first
GO2_HD int go2_cmd_stage(const Go2Launch& L, float it) {
  int best = -1;
  for (int i = 0; i < L.cmd_curr_count; ++i) if (it >= L.cmd_curr[i][0] && (best < 0 || L.cmd_curr[i][0] > L.cmd_curr[best][0])) best = i;
  return best;
}
// End of a pass: command_ranges['lin_vel_x'] as the reference's Python list evolves (go2sim.h cmd_tracking_curriculum).  Every
// _resample_commands call with >= 1 env first replaces the list when a new command_range_curriculum stage has started (:433-446); the
// post-physics callback's call (:409-410, `callback_count` envs) comes before reset_idx, whose update_command_curriculum (:728-737)
// widens the list and whose own _resample_commands call checks the stage again.  `csc` = the counter the pass ran with.
// (go2sim_create starts the list at commands.ranges.lin_vel_x with no stage seen.)
GO2_HD void go2_track_cmd_curriculum(const Go2Launch& L, Go2Dyn& dyn, float reset_count, float track_sum, float callback_count, int64_t csc, float* info_lo_hi) {
  const int stage = go2_cmd_stage(L, (float)(csc / L.num_steps_per_env));
  for (int call = 0; call < 2; ++call) {
    if (call == 0 ? callback_count <= 0.f : reset_count <= 0.f) continue;
    if (call == 1 && L.cmd_track_curr && track_sum / reset_count / L.max_episode_length > 0.8f * L.rew_scale_dt[GO2_REW_TRACKING_LIN_VEL]) {
      dyn.cmd_x_range[0] = fminf(fmaxf(dyn.cmd_x_range[0] - 0.5f, -L.cmd_max_curr), 0.f);
      dyn.cmd_x_range[1] = fminf(fmaxf(dyn.cmd_x_range[1] + 0.5f, 0.f), L.cmd_max_curr);
    }
    if (stage != dyn.cmd_stage_seen) {
      dyn.cmd_stage_seen = stage;
      if (stage >= 0) { dyn.cmd_x_range[0] = L.cmd_curr[stage][1]; dyn.cmd_x_range[1] = L.cmd_curr[stage][2]; }
    }
  }
  info_lo_hi[0] = dyn.cmd_x_range[0]; info_lo_hi[1] = dyn.cmd_x_range[1];
}
// The per-step scalars in GO2_STEP_SCALAR_PARTS independent parts, so that a workgroup computes them with one thread per part instead of
// ~700 serial instructions in front of its barrier: part t < GO2_NUM_REWARDS = reward term t's scales, part GO2_NUM_REWARDS = the rest.
#define GO2_STEP_SCALAR_PARTS (GO2_NUM_REWARDS + 1)
GO2_HD void go2_step_scalars_part(int part, const Go2Launch& L, const Go2Dyn& dyn, const float* inj_storage, int64_t csc, int initial_reset, Go2Step* S) {
  const float it = (float)(csc / L.num_steps_per_env);
  if (part < GO2_NUM_REWARDS) {
    const int t = part;
    float sc = L.rew_scale_dt[t], sct = L.rew_to_scale_dt[t];
    for (int i = 0; i < L.rew_curr_count; ++i) if (L.rew_curr_term[i] == t) { const float k = go2_current_scale(L.rew_curr[i], it); sc *= k; sct *= k; }
    S->rew_scale[t] = sc; S->rew_to_scale[t] = sct;
    return;
  }
  S->step_lo = (uint32_t)dyn.step_count; S->step_hi = (uint32_t)(dyn.step_count >> 32);
  S->initial_reset = initial_reset; S->injected = dyn.use_injected ? inj_storage : nullptr;
  const int best = go2_cmd_stage(L, it);
  _Pragma("unroll") for (int r = 0; r < 4; ++r) {
    S->cmd_ranges[r][0] = best < 0 ? L.cmd_ranges0[r][0] : L.cmd_curr[best][1 + 2 * r];
    S->cmd_ranges[r][1] = best < 0 ? L.cmd_ranges0[r][1] : L.cmd_curr[best][2 + 2 * r];
  }
  {
    const int seen = dyn.cmd_stage_seen < 0 ? -1 : dyn.cmd_stage_seen;      // (-2: no _resample_commands call yet = the configured ranges)
    S->stage_pending = (L.heading_command && best != seen) ? 1 : 0;
    S->yaw_range_seen[0] = seen < 0 ? L.cmd_ranges0[2][0] : L.cmd_curr[seen][5]; S->yaw_range_seen[1] = seen < 0 ? L.cmd_ranges0[2][1] : L.cmd_curr[seen][6];
  }
  S->max_lin_vel = fmaxf(fmaxf(fabsf(S->cmd_ranges[0][0]), fabsf(S->cmd_ranges[0][1])), fmaxf(fabsf(S->cmd_ranges[1][0]), fabsf(S->cmd_ranges[1][1])));
  S->zero_cmd_proba = L.zero_curr_enabled ? go2_current_scale(L.zero_curr, it) : 0.f;
  S->rew_mask = initial_reset ? 0u : L.rew_mask_all; S->rew_mask_all = L.rew_mask_all;
}
GO2_HD void go2_step_scalars(const Go2Launch& L, const Go2Dyn& dyn, const float* inj_storage, int64_t csc, int initial_reset, Go2Step* S) {
  for (int part = 0; part < GO2_STEP_SCALAR_PARTS; ++part) go2_step_scalars_part(part, L, dyn, inj_storage, csc, initial_reset, S);
}

// slot -> (Philox group << 2 | word): the mapping of include/go2sim_rng.h (go2_fill_slot_codes) as a constant expression, so that a slot
// known at compile time costs no table lookup (go2sim_create checks the two against each other for every slot)
GO2_HD constexpr int go2_slot_code(int s) {
  return s == GO2_U_DELAY ? 0
       : (s >= GO2_U_RSA && s < GO2_U_RSA + 4) ? 1 * 4 + (s - GO2_U_RSA)
       : (s >= GO2_U_RSA + 4 && s < GO2_U_RSA + 7) ? 2 * 4 + (s - GO2_U_RSA - 4)
       : (s >= GO2_U_RSB && s < GO2_U_RSB + 4) ? 22 * 4 + (s - GO2_U_RSB)
       : (s >= GO2_U_RSB + 4 && s < GO2_U_RSB + 7) ? 23 * 4 + (s - GO2_U_RSB - 4)
       : (s >= GO2_U_RESET_STRENGTH && s < GO2_U_RESET_STRENGTH + 12) ? (3 + 4 * ((s - GO2_U_RESET_STRENGTH) / 3) + ((0 + (s - GO2_U_RESET_STRENGTH) % 3) >> 2)) * 4 + ((0 + (s - GO2_U_RESET_STRENGTH) % 3) & 3)
       : (s >= GO2_U_RESET_OFFSET && s < GO2_U_RESET_OFFSET + 12) ? (3 + 4 * ((s - GO2_U_RESET_OFFSET) / 3) + ((3 + (s - GO2_U_RESET_OFFSET) % 3) >> 2)) * 4 + ((3 + (s - GO2_U_RESET_OFFSET) % 3) & 3)
       : (s >= GO2_U_RESET_KP && s < GO2_U_RESET_KP + 12) ? (3 + 4 * ((s - GO2_U_RESET_KP) / 3) + ((6 + (s - GO2_U_RESET_KP) % 3) >> 2)) * 4 + ((6 + (s - GO2_U_RESET_KP) % 3) & 3)
       : (s >= GO2_U_RESET_KD && s < GO2_U_RESET_KD + 12) ? (3 + 4 * ((s - GO2_U_RESET_KD) / 3) + ((9 + (s - GO2_U_RESET_KD) % 3) >> 2)) * 4 + ((9 + (s - GO2_U_RESET_KD) % 3) & 3)
       : (s >= GO2_U_RESET_DOF && s < GO2_U_RESET_DOF + 12) ? (3 + 4 * ((s - GO2_U_RESET_DOF) / 3) + ((12 + (s - GO2_U_RESET_DOF) % 3) >> 2)) * 4 + ((12 + (s - GO2_U_RESET_DOF) % 3) & 3)
       : s == GO2_U_RESET_TERRAIN ? 19 * 4 + 0 : s == GO2_U_RESET_YAW ? 19 * 4 + 1 : (s == GO2_U_RESET_XY || s == GO2_U_RESET_XY + 1) ? 19 * 4 + 2 + (s - GO2_U_RESET_XY)
       : (s >= GO2_U_RESET_VEL && s < GO2_U_RESET_VEL + 6) ? (20 + ((s - GO2_U_RESET_VEL) >> 2)) * 4 + ((s - GO2_U_RESET_VEL) & 3)
       : (s >= GO2_U_PUSH && s < GO2_U_PUSH + 5) ? (24 + ((s - GO2_U_PUSH) >> 2)) * 4 + ((s - GO2_U_PUSH) & 3)
       : (s >= GO2_U_NOISE && s < GO2_U_NOISE + 9) ? (26 + (s - GO2_U_NOISE) / 3) * 4 + (s - GO2_U_NOISE) % 3
       : (s >= GO2_U_NOISE + 9 && s < GO2_U_NOISE + 45) ? (29 + 4 * ((s - GO2_U_NOISE - 9) / 12) + ((s - GO2_U_NOISE - 9) % 12) / 3) * 4 + (s - GO2_U_NOISE - 9) % 3
       : (s >= GO2_U_TURN && s < GO2_U_TURN + 4) ? 41 * 4 + (s - GO2_U_TURN)
       : 42 * 4;
}

struct LegPost {
  int e, lane, sub, lane16, N;     // lane = leg (0..3), sub = sub-lane of the leg, lane16 = leg * 4 + sub
  const Go2PtrsK* P; const Go2Launch* L; const Go2Step* S;
  PhysOut o;
  // env scalars (replicated)
  int64_t ep_len; float timer, cmd[4], acc[2]; uint8_t stop_heading, last_limit;
  V3 blv, bav, pg; float base_height;
  uint8_t reset, time_out;
  float act[3], last_act[3], llast_act[3], last_dv[3];

  // Uniform `slot` of this env-step (contract: include/go2sim_rng.h: slot -> (Philox group, word)).  The 16 lanes of an environment
  // share ONE table of drawn groups in LDS (uc: [GO2_NUM_GROUPS][4] floats of this env): a group is drawn by exactly one lane — fill()
  // deals up to 16 consecutive groups to the 16 lanes, one Philox call each — and read by whichever lane consumes one of its slots.
  // The per-step groups (action delay, observation noise) are drawn when the kernel starts; the reset / resample / push groups where
  // those (per-environment, hence row-uniform) branches are taken.
  const GO2_AS3 uint8_t* codes; GO2_AS3 float (*uc)[4];      // both live in LDS (ds_read / ds_write, not flat accesses)
  GO2_AS3 float* hc;                                         // this env's height samples of postA, kept in LDS for postB's observation rows (the lane that wrote entry i reads entry i)
  GO2_HD float uni(int slot) const {
    if (S->injected) return S->injected[(size_t)e * GO2_NUM_UNIFORMS + slot];
    const int code = __builtin_constant_p(slot) ? go2_slot_code(slot) : (int)codes[slot];
    return uc[code >> 2][code & 3];
  }
  GO2_HD void draw_group(int g) {
    uint32_t r[4];
    philox4x32_10((uint32_t)(L->env_offset + e), (uint32_t)g, S->step_lo, S->step_hi, L->seed_lo, L->seed_hi, r);
    uc[g][0] = u01_from_bits(r[0]); uc[g][1] = u01_from_bits(r[1]); uc[g][2] = u01_from_bits(r[2]); uc[g][3] = u01_from_bits(r[3]);
  }
  // groups g0 .. g0+n-1 (n <= 16), lane k of the row draws group g0+k; `extra` >= 0: one more group for lane n.  Called by all 16 lanes.
  GO2_HD void fill(int g0, int n, int extra = -1) {
    if (!S->injected) {
      if (lane16 < n) draw_group(g0 + lane16);
      else if (lane16 == n && extra >= 0) draw_group(extra);
    }
    xl::row_sync();
  }
  // the second pass of a reset: terrain / yaw / xy (19), root velocity (20, 21), the resample inside reset (22, 23), the turn-over draws (41)
  // and — a freshly reset env is pushed in the same step (episode clock 0, App. E.4) — the push groups (24, 25): one group per lane
  GO2_HD void fill_reset_tail(bool turn_over, bool push) {
    if (!S->injected) {
      if (lane16 < 5) draw_group(19 + lane16);
      else if (lane16 == 5) { if (turn_over) draw_group(41); }
      else if (lane16 < 8) { if (push) draw_group(24 + (lane16 - 6)); }
    }
    xl::row_sync();
  }
  GO2_HD static float urange(float u, float lo, float hi) { return (hi - lo) * u + lo; }
  GO2_HD void cmd_range(int which, float* lo, float* hi) const {  // env_command_ranges (:861-907)
    *lo = S->cmd_ranges[which][0]; *hi = S->cmd_ranges[which][1];
    int kind = P->terrain_kind[e];
    if (kind >= 0 && (which < 3 || L->heading_command)) {
      *lo = fmaxf(*lo, L->terrain_max_cmd[kind][which][0]); *hi = fminf(*hi, L->terrain_max_cmd[kind][which][1]);
    }
  }
  GO2_HD static float sample_disjoint(float u, float bound, float cmin, float cmax) {  // isaacgym_utils.py:32-47
    float wn = fmaxf(-bound - cmin, 0.f), wp = fmaxf(cmax - bound, 0.f);
    float t = u * (wn + wp + 1e-6f);
    return t < wn ? cmin + t : cmax - wp + (t - wn);
  }
  // _resample_commands (:423-592) for this env; U = first of the 7 uniform slots
  GO2_HD void resample(int U) {
    stop_heading = 0;
    float remaining = fmaxf(0.625f * L->terrain_length - sqrtf(acc[0] * acc[0] + acc[1] * acc[1]) * L->resampling_time, 0.f);
    timer = L->resampling_time / L->dt;
    float xl, xh, yl, yh, wl, wh, hl, hh; cmd_range(0, &xl, &xh); cmd_range(1, &yl, &yh); cmd_range(2, &wl, &wh); cmd_range(3, &hl, &hh);
    float epl = (float)ep_len;
    if (L->dynamic_resample) {
      float vlow = fmaxf(remaining / ((L->max_episode_length - epl + 1e-9f) * L->dt), 0.f);
      cmd[0] = sample_disjoint(uni(U + 0), vlow, xl, xh);
      cmd[1] = sample_disjoint(uni(U + 1), vlow, yl, yh);
      { const float u2 = uni(U + 2), c3 = urange(u2, hl, hh), c2 = urange(u2, wl, wh);      // (selects, not `if (..) cmd[3] = .. else cmd[2] = ..`: a
        cmd[3] = L->heading_command ? c3 : cmd[3]; cmd[2] = L->heading_command ? cmd[2] : c2; }  //  conditional INDEX would put cmd[] in scratch memory)
    } else {
      cmd[0] = xl + uni(U + 0) * (xh - xl); cmd[1] = yl + uni(U + 1) * (yh - yl);
      { const float u2 = uni(U + 2), c3 = hl + u2 * (hh - hl), c2 = wl + u2 * (wh - wl);
        cmd[3] = L->heading_command ? c3 : cmd[3]; cmd[2] = L->heading_command ? cmd[2] : c2; }
      if (!(sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1]) > 0.2f)) { cmd[0] = 0; cmd[1] = 0; }
    }
    float p = uni(U + 3), minp = 0.f, maxp = 0.f;
    if (L->limit_vel_prob > 0.f) {
      maxp += L->limit_vel_prob;
      bool lim = (p >= minp) && (p < maxp);
      if (lim) {
        if (L->limit_invert && last_limit) { cmd[0] = -cmd[0]; cmd[1] = -cmd[1]; cmd[2] = -cmd[2]; }
        else {
          int idx = (int)(uni(U + 4) * L->comb_count); idx = idx >= L->comb_count ? L->comb_count - 1 : idx;
          float c0 = L->comb[idx][0], c1 = L->comb[idx][1], c2 = L->comb[idx][2];
          cmd[0] = c0 == 0.f ? 0.f : (c0 == -1.f ? xl : xh);
          cmd[1] = c1 == 0.f ? 0.f : (c1 == -1.f ? yl : yh);
          cmd[2] = c2 == 0.f ? 0.f : (c2 == -1.f ? wl : wh);
        }
        if (L->heading_command && L->stop_heading_at_limit) stop_heading = 1;
      }
      last_limit = lim ? 1 : 0;
      minp += L->limit_vel_prob;
    }
    if (S->zero_cmd_proba > 0.f) {
      maxp += S->zero_cmd_proba;
      float nxt = L->max_episode_length - epl - remaining / (0.8f * S->max_lin_vel * L->dt + 1e-9f);
      nxt = fminf(fmaxf(nxt, 0.f), L->resampling_time / L->dt);
      if ((p >= minp) && (p < maxp) && nxt > 0.f) {
        cmd[0] = 0; cmd[1] = 0; timer = nxt;
        if (L->limit_ang_zero_prob > 0.f && uni(U + 5) < L->limit_ang_zero_prob) {
          cmd[2] = uni(U + 6) < 0.5f ? wl : wh;
          if (L->heading_command) stop_heading = 1;
        }
      }
    }
    if (L->turn_over && to_timer > 0.f) { cmd[0] = 0.f; cmd[1] = 0.f; cmd[2] = 0.f; stop_heading = 1; }   // :586-590
    acc[0] += cmd[0]; acc[1] += cmd[1];
  }
  float to_timer;
  // one sample of _get_heights (:1188-1224) around base (bx, by) with yaw quaternion (0,0,yz,yw).  INDEX work: every operation is
  // rounded on its own, in the reference's order (quat_apply: t = 2 cross(q, v); v + w t + cross(q, t); then + base, + border, / scale,
  // truncate), so the cell index equals the reference's bit for bit (go2_math.h: go2_*_rn).
  GO2_HD float height_at(int i, float yz, float yw, float bx, float by) const {
    const Go2Launch& c = *L;
    if (c.terrain_mode == 0) return 0.f;
    const int ix = i / 11, iy = i - 11 * ix;
    const float vx = (float)(ix - 8) * 0.1f, vy = (float)(iy - 5) * 0.1f;      // height_points: 0.1 * [-8..8] x 0.1 * [-5..5] (legged_robot_config.py:26-27)
    // q = (0, 0, yz, yw): c = cross(q, v) = (-yz vy, yz vx, 0), cc = cross(q, c) = (-yz c.y, yz c.x, 0); o = (v + (2 yw) c) + 2 cc
    const float cx = -go2_mul_rn(yz, vy), cy = go2_mul_rn(yz, vx), w2 = 2.f * yw;
    const float ccx = -go2_mul_rn(yz, cy), ccy = go2_mul_rn(yz, cx);
    const float wx = go2_add_rn(go2_add_rn(vx, go2_mul_rn(w2, cx)), 2.f * ccx), wy = go2_add_rn(go2_add_rn(vy, go2_mul_rn(w2, cy)), 2.f * ccy);
    const float x = go2_mul_inv_rn(go2_add_rn(go2_add_rn(wx, bx), c.hf_border), c.hf_inv_hscale), y = go2_mul_inv_rn(go2_add_rn(go2_add_rn(wy, by), c.hf_border), c.hf_inv_hscale);
    int px = (int)x, py = (int)y;
    px = px < 0 ? 0 : (px > c.hf_rows - 2 ? c.hf_rows - 2 : px); py = py < 0 ? 0 : (py > c.hf_cols - 2 ? c.hf_cols - 2 : py);
    int h1 = P->hf[px * c.hf_cols + py], h2 = P->hf[(px + 1) * c.hf_cols + py], h3 = P->hf[px * c.hf_cols + py + 1];
    int hm = h1 < h2 ? h1 : h2; hm = h3 < hm ? h3 : hm;
    return hm * c.hf_vscale;
  }
  GO2_HD float dyn_sigma(float vabs, float vmin, float vmax) const {  // :1300-1320
    float def = L->tracking_sigma; int kind = P->terrain_kind[e];
    if (!L->terrain_curriculum || !L->dyn_sigma || kind < 0) return def;
    float target = L->dyn_sigma_max[kind], sig = def;
    if (vabs >= vmin && vabs < vmax) sig = def + (vabs - vmin) / (vmax - vmin) * (target - def);
    if (vabs >= vmax) sig = target;
    float ls = fminf(expf(((float)tlevel + 1.f) / 10.f) - 1.f, 1.f);
    return def + ls * (sig - def);
  }

  // ------------------------------------------------------------------------------------------------
  GO2_HD void postA(const LegTab& t, float* part) {
    const Go2PtrsK& p = *P; const Go2Launch& c = *L;
    ep_len = p.ep_len[e] + 1;                      // :111
    timer = p.cmd_timer[e] - 1.f;                  // :113
    to_timer = c.turn_over ? fmaxf(p.to_timer[e] - c.dt, 0.f) : 0.f;     // :114-115
    _Pragma("unroll") for (int k = 0; k < 4; ++k) cmd[k] = F2D(p.commands, k, e);
    acc[0] = F2D(p.cmd_xy_acc, 0, e); acc[1] = F2D(p.cmd_xy_acc, 1, e);
    stop_heading = p.stop_heading[e]; last_limit = p.last_is_limit_vel[e];
    _Pragma("unroll") for (int j = 0; j < 3; ++j) {
      int d = 3 * lane + j;
      act[j] = F2D(p.actions, d, e); last_act[j] = F2D(p.last_actions, d, e); llast_act[j] = F2D(p.last_last_actions, d, e); last_dv[j] = F2D(p.last_dof_vel, d, e);
    }
    // derived base quantities (:119-125); get_euler_xyz (utils/isaacgym_utils.py:11-30)
    float qx = o.qx, qy = o.qy, qz = o.qz, qw = o.qw;
    float sr = 2 * (qw * qx + qy * qz), cr = qw * qw - qx * qx - qy * qy + qz * qz;
    float sp = 2 * (qw * qy - qz * qx);
    float sy = 2 * (qw * qz + qx * qy), cy = qw * qw + qx * qx - qy * qy - qz * qz;
    float roll = atan2f(sr, cr), pitch = fabsf(sp) >= 1.f ? copysignf(1.57079632679f, sp) : asinf(sp), yaw = atan2f(sy, cy);
    blv = quat_rotate_inverse(qx, qy, qz, qw, o.vw); bav = quat_rotate_inverse(qx, qy, qz, qw, o.ww);
    pg = quat_rotate_inverse(qx, qy, qz, qw, v3(0, 0, -1));
    load_terrain_fields();
    float mm = fmaxf(p.max_move[e], sqrtf((o.pw.x - org_x) * (o.pw.x - org_x) + (o.pw.y - org_y) * (o.pw.y - org_y)));
    // _post_physics_step_callback (:404-421)
    if (timer <= 0.f && (float)ep_len < c.max_episode_length - 1.f) {      // (per-env branch: the whole row takes it)
      fill(1, 2); resample(GO2_U_RSA);
      if ((c.cmd_track_curr || c.heading_command) && lane16 == 0) {      // a _resample_commands call happened before reset_idx (go2_track_cmd_curriculum)
#if defined(__HIP_DEVICE_COMPILE__)
        atomicAdd(&p.ep_accum[GO2_NUM_REWARDS + 1], 1.0f);
#else
        p.ep_accum[GO2_NUM_REWARDS + 1] += 1.0f;
#endif
      }
    }
    if (c.heading_command && !stop_heading) {
      V3 fw = quat_apply(qx, qy, qz, qw, v3(1, 0, 0)); float heading = atan2f(fw.y, fw.x);
      float a = fmodf(cmd[3] - heading, 6.28318530718f); if (a < 0) a += 6.28318530718f; if (a > 3.14159265359f) a -= 6.28318530718f;
      float lo, hi; cmd_range(2, &lo, &hi);
      if (yaw_seen) {      // a started stage that no _resample_commands call of this batch has picked up (Go2Step.stage_pending)
        lo = S->yaw_range_seen[0]; hi = S->yaw_range_seen[1];
        const int kind = p.terrain_kind[e];
        if (kind >= 0) { lo = fmaxf(lo, c.terrain_max_cmd[kind][2][0]); hi = fminf(hi, c.terrain_max_cmd[kind][2][1]); }
      }
      cmd[2] = fminf(fmaxf(0.5f * a, lo), hi);
    }
    // _get_heights (:1188-1224): this lane samples points lane, lane+4, ...
    float hsum = 0.f;
    if (c.measure_heights && c.terrain_mode != 0) {   // on a plane measured_heights stays the all-zero buffer it was created as (:1201-1202)
      // quat_apply_yaw (utils/math.py:8-12): zero x, y, normalise — individually rounded, it feeds the cell index
      const float nn = fmaxf(go2_sqrt_rn(go2_add_rn(go2_mul_rn(qz, qz), go2_mul_rn(qw, qw))), 1e-9f), yz = go2_div_rn(qz, nn), yw = go2_div_rn(qw, nn);
      for (int i = lane16; i < GO2_NUM_HEIGHT_POINTS; i += 16) {
        const float hv = height_at(i, yz, yw, o.pw.x, o.pw.y);
        const int ix = i / 11, iy = i - 11 * ix;
        F2D(p.heights, i, e) = hv; hc[i] = hv;
        if (ix >= 6 && ix <= 10 && iy >= 4 && iy <= 6) hsum += hv;   // base_height_scan_mask (:790-796)
      }
    }
    // check_termination (:170-178)
    bool term = !c.turn_over && sqrtf(dot(o.Fbase, o.Fbase)) > 1.f;      // :174
    time_out = (float)ep_len > c.max_episode_length ? 1 : 0;
    reset = (term || time_out) ? 1 : 0;
    // ---- reward partial sums over this lane's joints / bodies ------------------------------------
    float tq2 = 0, dv2 = 0, dacc = 0, arate = 0, plim = 0, vlim = 0, tlim = 0, still = 0, smooth = 0, power = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      int d = 3 * lane + j;
      tq2 += o.tau[j] * o.tau[j]; dv2 += o.qd[j] * o.qd[j];
      float a = (last_dv[j] - o.qd[j]) / c.dt; dacc += a * a;
      float r = last_act[j] - act[j]; arate += r * r;
      float lo = o.q[j] - c.soft_limits[d][0]; if (lo < 0) plim -= lo; float hi = o.q[j] - c.soft_limits[d][1]; if (hi > 0) plim += hi;
      vlim += fminf(fmaxf(fabsf(o.qd[j]) - t.vel_lim[j] * c.soft_vel_limit, 0.f), 1.f);
      tlim += fmaxf(fabsf(o.tau[j]) - t.eff_lim[j] * c.soft_torque_limit, 0.f);
      still += fabsf(o.q[j] - c.q0[d]);
      float s = act[j] - 2 * last_act[j] + llast_act[j]; smooth += s * s;
      power += fabsf(o.tau[j] * o.qd[j]);
    }
    float coll = (sqrtf(dot(o.Fthigh, o.Fthigh)) > 0.1f ? 1.f : 0.f) + (sqrtf(dot(o.Fcalf, o.Fcalf)) > 0.1f ? 1.f : 0.f);
    float fnorm = sqrtf(dot(o.Ffoot, o.Ffoot));
    bool contact = o.Ffoot.z > 1.f;
    // _reward_base_height (:1245-1259)
    float bh_cnt = 0; V3 bh_pos = v3(0, 0, 0);
    if (c.rew_on[GO2_REW_BASE_HEIGHT]) {
      bool filt = contact || F2D(p.last_contacts2, lane, e); new_lc2 = contact ? 1 : 0;
      if (filt) { bh_cnt = 1; bh_pos = o.foot_pos; }
    }
    // _reward_feet_air_time (:1347-1358)
    float air = 0;
    if (c.rew_on[GO2_REW_FEET_AIR_TIME]) {
      bool filt = contact || F2D(p.last_contacts, lane, e); new_lc = contact ? 1 : 0;
      float fat = F2D(p.feet_air_time, lane, e); bool first = fat > 0.f && filt; fat += c.dt; air = (fat - 0.5f) * (first ? 1.f : 0.f);
      new_fat = filt ? 0.f : fat;
    }
    float stumble = sqrtf(o.Ffoot.x * o.Ffoot.x + o.Ffoot.y * o.Ffoot.y) > 5.f * fabsf(o.Ffoot.z) ? 1.f : 0.f;
    float fcf = fmaxf(fnorm - c.max_contact_force, 0.f);
    V3 dfoot = o.foot_pos - o.pw;
    float f2b = dot(dfoot, pg);                                   // feet_regulation (:1404-1414), finished in postB
    float fvel2 = o.foot_vel.x * o.foot_vel.x + o.foot_vel.y * o.foot_vel.y;
    V3 floc = quat_rotate_inverse(qx, qy, qz, qw, dfoot);          // legs_distance (:1423-1441)
    float hip = o.q[0];
    part[0] = tq2; part[1] = dv2; part[2] = dacc; part[3] = arate; part[4] = coll; part[5] = plim; part[6] = vlim; part[7] = tlim;
    part[8] = air; part[9] = stumble; part[10] = still; part[11] = fcf; part[12] = smooth; part[13] = power;
    part[14] = bh_cnt; part[15] = bh_pos.x; part[16] = bh_pos.y; part[17] = bh_pos.z;
    part[18] = fabsf(hip - c.q0[3 * lane]);                        // hip_to_default (go2_env.py:55)
    part[19] = lane < 2 ? hip : 0.f; part[20] = lane >= 2 ? hip : 0.f;   // x_command_hip_regular (go2_env.py:62)
    part[21] = lane == 0 ? floc.y : (lane == 1 ? -floc.y : 0.f); part[22] = lane == 2 ? floc.y : (lane == 3 ? -floc.y : 0.f);
    part[23] = xl::sub_sum(hsum);   // the leg's samples = its four sub-lanes' shares (every other partial is replicated in the quad)
    own_f2b = f2b; own_fvel2 = fvel2;
    rpy[0] = roll; rpy[1] = pitch; rpy[2] = yaw; max_move = mm;
  }
  float own_f2b, own_fvel2, rpy[3], max_move, org_x, org_y, org_z; int64_t tlevel, ttype;
  bool skip_contact_filters;                   // reset_all runs postB without a postA: nothing to carry over
#if defined(__HIP_DEVICE_COMPILE__) && defined(GO2_KBENCH_STAMPS)
  long long* dbg;                              // per-wave phase timestamps: only in the tools/kbench.py build (-DGO2_KBENCH_STAMPS)
#define GO2_POST_STAMP(k) do { if (dbg && (lane16 == 0) && ((e & 3) == 0)) dbg[k] = wall_clock64(); } while (0)
#else
#define GO2_POST_STAMP(k) do { } while (0)
#endif
  Go2StepOutputs out;                          // go2sim_step_rollout: redirected observations, fused transition store (all null otherwise)
  bool yaw_seen;                               // heading clip against the ranges of the stage last picked up (Go2Step.stage_pending)
  bool api_reset;                              // go2sim_reset_idx: like the reference's reset_idx, leave observations / reward / derived velocities alone
  uint8_t new_lc, new_lc2; float new_fat;      // per-leg read-modify-write fields: read by the 4 sub-lanes in postA, written by sub-lane 0 in postB
  // replicated fields that lane 0 rewrites in postB: every lane reads them BEFORE any lane writes
  GO2_HD void load_terrain_fields() {
    const Go2PtrsK& p = *P;
    org_x = F2D(p.origins, 0, e); org_y = F2D(p.origins, 1, e); org_z = F2D(p.origins, 2, e);
    tlevel = p.terrain_levels[e]; ttype = p.terrain_types[e];
  }

  // feet_regulation needs base_height (a quad quantity) and then a second quad-sum; to keep ONE reduction the
  // caller passes, besides the summed partials, nothing else: each lane recomputes its own term and the
  // four terms are summed with a second tiny reduction (1 float) inside postB's caller.  See regulation().
  GO2_HD float regulation(const float* red) const {
    float bh = base_height_from(red);
    float fh = fmaxf(bh - own_f2b, 0.f);
    return own_fvel2 * expf(-fh / (0.025f * L->base_height_target));
  }
  GO2_HD float base_height_from(const float* red) const {        // _get_base_height (:1387-1397)
    if (!L->measure_heights) return o.pw.z;
    return o.pw.z - red[23] / 15.0f;
  }

  // ------------------------------------------------------------------------------------------------
  // red = quad-summed partials; feet_reg = quad-summed regulation()
  GO2_HD void postB(const LegTab& t, const float* red, float feet_reg) {
    const Go2PtrsK& p = *P; const Go2Launch& c = *L; const int N_ = N; (void)N_;
    float raw[GO2_NUM_REWARDS];
#pragma unroll
    for (int i = 0; i < GO2_NUM_REWARDS; ++i) raw[i] = 0.f;
    {
      float sx = c.tracking_sigma, sy_ = sx, sg = sx;
      if (c.dyn_sigma) { sx = dyn_sigma(fabsf(cmd[0]), c.dyn_sigma_vel[0], c.dyn_sigma_vel[1]); sy_ = dyn_sigma(fabsf(cmd[1]), c.dyn_sigma_vel[0], c.dyn_sigma_vel[1]);
                         sg = dyn_sigma(fabsf(cmd[2]), c.dyn_sigma_vel[2], c.dyn_sigma_vel[3]); }
      float ex = cmd[0] - blv.x, ey = cmd[1] - blv.y, er = cmd[2] - bav.z;
      raw[GO2_REW_TRACKING_LIN_VEL] = expf(-(ex * ex / sx + ey * ey / sy_));   // :1322
      raw[GO2_REW_TRACKING_ANG_VEL] = expf(-er * er / sg);                      // :1336
    }
    raw[GO2_REW_LIN_VEL_Z] = blv.z * blv.z;                                     // :1228
    raw[GO2_REW_ANG_VEL_XY] = bav.x * bav.x + bav.y * bav.y;                    // :1232
    raw[GO2_REW_ORIENTATION] = pg.x * pg.x + pg.y * pg.y;                       // :1236
    {
      float nc = red[14], den = fmaxf(nc, 1.f);
      float bh = (red[15] / den - o.pw.x) * pg.x + (red[16] / den - o.pw.y) * pg.y + (red[17] / den - o.pw.z) * pg.z;
      float d = bh - c.base_height_target; raw[GO2_REW_BASE_HEIGHT] = d * d * (nc > 0.f ? 1.f : 0.f);
    }
    raw[GO2_REW_TORQUES] = red[0]; raw[GO2_REW_DOF_VEL] = red[1]; raw[GO2_REW_DOF_ACC] = red[2]; raw[GO2_REW_ACTION_RATE] = red[3];
    raw[GO2_REW_COLLISION] = red[4]; raw[GO2_REW_DOF_POS_LIMITS] = red[5]; raw[GO2_REW_DOF_VEL_LIMITS] = red[6]; raw[GO2_REW_TORQUE_LIMITS] = red[7];
    float cn = sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1]);
    raw[GO2_REW_FEET_AIR_TIME] = red[8] * (cn > 0.1f ? 1.f : 0.f);
    raw[GO2_REW_STUMBLE] = red[9] > 0.f ? 1.f : 0.f;
    raw[GO2_REW_STAND_STILL] = red[10] * (cn < 0.1f ? 1.f : 0.f);
    raw[GO2_REW_FEET_CONTACT_FORCES] = red[11]; raw[GO2_REW_ACTION_SMOOTHNESS] = red[12]; raw[GO2_REW_DOF_POWER] = red[13];
    { float d = base_height_from(red) - c.base_height_target; raw[GO2_REW_CORRECT_BASE_HEIGHT] = d * d; }   // :1399
    raw[GO2_REW_FEET_REGULATION] = feet_reg;
    raw[GO2_REW_SIMILAR_TO_DEFAULT] = red[10];
    raw[GO2_REW_UPRIGHT] = (-1.f - pg.z) * 0.5f;
    { float df = fmaxf(c.min_legs_distance - red[21], 0.f), dr = fmaxf(c.min_legs_distance - red[22], 0.f); raw[GO2_REW_LEGS_DISTANCE] = df * df + dr * dr; }
    raw[GO2_REW_HIP_TO_DEFAULT] = red[18];
    raw[GO2_REW_X_COMMAND_HIP_REGULAR] = (fabsf(red[19]) + fabsf(red[20])) * fabsf(cmd[0]) / sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1] + cmd[2] * cmd[2]);
    // Sum of the active terms (enum order) and their per-episode sums.  Which terms are active is ONE wave-uniform bit mask (Go2Step),
    // so every test below is a scalar branch.  episode_sums: lane k of the environment's row OWNS terms k and k + 16 — one load, one add,
    // one store per lane, all lanes in parallel — and in the reset branch adds its two sums to the extras accumulators (two atomic
    // instructions per wave instead of one, with its wave reduction and its wait, per term).
    static_assert(GO2_NUM_REWARDS + 1 <= 32, "two owned terms per lane");
    float total = 0.f;
    GO2_AS1 float* es = p.ep_sums;  // [R][N] row-major
    const bool need_to = c.turn_over && fabsf(rpy[0]) > c.to_roll_thr;      // :263-265
    const uint32_t rmask = go2_uniform_u32(S->rew_mask);
    const int own0 = lane16, own1 = lane16 + 16;
    const bool has0 = (rmask >> own0) & 1u, has1 = own1 < GO2_NUM_REWARDS && ((rmask >> own1) & 1u);
    float es0 = has0 ? es[(size_t)own0 * N + e] : 0.f, es1 = has1 ? es[(size_t)own1 * N + e] : 0.f;
    float r0 = 0.f, r1 = 0.f;      // this step's value of the two owned terms
#pragma unroll
    for (int i = 0; i < GO2_REW_TERMINATION; ++i) if ((rmask >> i) & 1u) {
      const float r = raw[i] * (need_to ? S->rew_to_scale[i] : S->rew_scale[i]); total += r;
      if (i < 16) r0 = lane16 == i ? r : r0; else r1 = lane16 == i - 16 ? r : r1;
    }
    if (c.only_positive && total < 0.f) total = 0.f;
    if ((rmask >> GO2_REW_TERMINATION) & 1u) {
      const float r = ((reset && !time_out) ? 1.f : 0.f) * S->rew_scale[GO2_REW_TERMINATION]; total += r;
      if (GO2_REW_TERMINATION < 16) r0 = lane16 == GO2_REW_TERMINATION ? r : r0; else r1 = lane16 == GO2_REW_TERMINATION - 16 ? r : r1;
    }
    es0 += r0; es1 += r1;
    if (c.rew_on[GO2_REW_ACTION_SMOOTHNESS] && !skip_contact_filters)     // (a reset outside a step computes no reward)
      _Pragma("unroll") for (int j = 0; j < 3; ++j) llast_act[j] = last_act[j];       // :1378 (not zeroed on reset, App. E.9)

    GO2_MARK(23);
    GO2_POST_STAMP(6);
    // ---- reset_idx (:180-245) ---------------------------------------------------------------------
    float ox = org_x, oy = org_y, oz = org_z;
    bool es_dirty_all = false;
    if (reset) {
      fill(3, 16);                                  // per-DOF reset groups of the 4 legs (go2sim_rng.h: group 3 + 4 leg + g)
      fill_reset_tail(c.turn_over, c.push_robots && !S->initial_reset);
      GO2_POST_STAMP(8);
      // the four per-DOF tables: sub-lane k of a leg draws and writes table k (strength, offset, kp, kd) for the leg's 3 joints
      {
        const int tb = sub == 0 ? GO2_U_RESET_STRENGTH : (sub == 1 ? GO2_U_RESET_OFFSET : (sub == 2 ? GO2_U_RESET_KP : GO2_U_RESET_KD));
        const bool on = sub == 0 ? c.rand_strength : (sub == 1 ? c.rand_offset : c.rand_pd);
        const float lo = sub == 0 ? c.strength_rng[0] : (sub == 1 ? c.offset_rng[0] : (sub == 2 ? c.kp_rng[0] : c.kd_rng[0]));
        const float hi = sub == 0 ? c.strength_rng[1] : (sub == 1 ? c.offset_rng[1] : (sub == 2 ? c.kp_rng[1] : c.kd_rng[1]));
        GO2_AS1 float* dst = sub == 0 ? p.strength : (sub == 1 ? p.zero_off : (sub == 2 ? p.kp_mul : p.kd_mul));
        if (on) _Pragma("unroll") for (int j = 0; j < 3; ++j) F2D(dst, 3 * lane + j, e) = urange(uni(tb + 3 * lane + j), lo, hi);
      }
      GO2_POST_STAMP(9);
      if (c.terrain_curriculum && c.terrain_mode != 0 && S->initial_reset != 1) {   // _update_terrain_curriculum (:1143-1169)
        float dist = max_move;
        bool up = dist > c.terrain_length * 0.5f, down;
        if (c.move_down_by_acc) down = (dist < sqrtf(acc[0] * acc[0] + acc[1] * acc[1]) * (c.resampling_time * (1.f - S->zero_cmd_proba)) * 0.5f) && !up;
        else down = (dist < sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1]) * c.episode_length_s * 0.5f) && !up;
        int64_t lv = tlevel + (up ? 1 : 0) - (down ? 1 : 0);
        if (lv >= c.terrain_num_levels) { int r = (int)(uni(GO2_U_RESET_TERRAIN) * c.terrain_num_levels); lv = r >= c.terrain_num_levels ? c.terrain_num_levels - 1 : r; }
        else if (lv < 0) lv = 0;
        const float* og = p.terrain_origins + ((size_t)lv * c.terrain_num_types + ttype) * 3;
        ox = og[0]; oy = og[1]; oz = og[2]; max_move = 0.f;
        if (lane16 == 0) { p.terrain_levels[e] = lv; F2D(p.origins, 0, e) = ox; F2D(p.origins, 1, e) = oy; F2D(p.origins, 2, e) = oz; }
      }
      GO2_POST_STAMP(10);
#pragma unroll
      for (int j = 0; j < 3; ++j) {   // _reset_dofs (:620-634)
        int d = 3 * lane + j;
        o.q[j] = c.q0[d] * urange(uni(GO2_U_RESET_DOF + d), 0.5f, 1.5f); o.qd[j] = 0.f;
        act[j] = 0.f; last_act[j] = 0.f; last_dv[j] = 0.f;
      }
      // _reset_root_states (:635-707)
      float yaw = urange(uni(GO2_U_RESET_YAW), -3.14159265358979f, 3.14159265358979f);
      float zinit = c.base_init[2], roll = 0.f;
      if (c.turn_over) {   // :642-684: a share of the resets starts on the back (roll pi) or on a side (roll +-pi/2)
        to_timer = 0.f;
        const float pr = uni(GO2_U_TURN), p0 = c.to_prop[0], p1 = p0 + c.to_prop[1];
        if (pr >= 0.f && pr < p0) { zinit = urange(uni(GO2_U_TURN + 1), c.to_height[0][0], c.to_height[0][1]); roll = 3.14159265358979f; to_timer = c.to_zero_time[0]; }
        else if (pr >= p0 && pr < p1) { zinit = urange(uni(GO2_U_TURN + 2), c.to_height[1][0], c.to_height[1][1]);
          roll = uni(GO2_U_TURN + 3) < 0.5f ? 1.57079632679490f : -1.57079632679490f; to_timer = c.to_zero_time[1]; }
      }
      o.pw = v3(c.base_init[0] + ox, c.base_init[1] + oy, zinit + oz);
      if (c.terrain_mode != 0) { o.pw.x += urange(uni(GO2_U_RESET_XY), -1.f, 1.f); o.pw.y += urange(uni(GO2_U_RESET_XY + 1), -1.f, 1.f); }
      { float cy, sy, cr = 1.f, sr = 0.f; go2_sincos_half_pi(0.5f * yaw, &sy, &cy);       // quat_from_euler_xyz(roll, 0, yaw); |yaw / 2|, |roll / 2| <= pi / 2
        if (c.turn_over) go2_sincos_half_pi(0.5f * roll, &sr, &cr);
        o.qx = cy * sr; o.qy = sy * sr; o.qz = sy * cr; o.qw = cy * cr; }
      o.vw = v3(urange(uni(GO2_U_RESET_VEL), -0.5f, 0.5f), urange(uni(GO2_U_RESET_VEL + 1), -0.5f, 0.5f), urange(uni(GO2_U_RESET_VEL + 2), -0.5f, 0.5f));
      o.ww = v3(urange(uni(GO2_U_RESET_VEL + 3), -0.5f, 0.5f), urange(uni(GO2_U_RESET_VEL + 4), -0.5f, 0.5f), urange(uni(GO2_U_RESET_VEL + 5), -0.5f, 0.5f));
      new_fat = 0.f;
      if (sub == 0) _Pragma("unroll") for (int a = 0; a < 3; ++a) F3D(p.foot_impulse, 4, lane, a, e) = 0.f;
      ep_len = 0;
      timer = c.resampling_time / c.dt; acc[0] = 0.f; acc[1] = 0.f;
      GO2_POST_STAMP(11);
      resample(GO2_U_RSB);
      GO2_POST_STAMP(12);
      {   // extras["episode"] accumulators (:229-242); every term the reference has a sum for, i.e. every computed term; [NUM_REWARDS] counts the envs
        const uint32_t amask = go2_uniform_u32(S->rew_mask_all);
        _Pragma("unroll") for (int h = 0; h < 2; ++h) {
          const int t = lane16 + 16 * h;
          const bool term = t < GO2_NUM_REWARDS && ((amask >> t) & 1u);
          float v = h ? es1 : es0;
          if (term && !rmask) v = es[(size_t)t * N + e];      // (a reset outside a step: nothing was loaded above)
          if (t == GO2_NUM_REWARDS) v = 1.0f;
          if (term || t == GO2_NUM_REWARDS) {
#if defined(__HIP_DEVICE_COMPILE__)
            atomicAdd(&p.ep_accum[t], v);
#else
            p.ep_accum[t] += v;
#endif
          }
        }
        es0 = 0.f; es1 = 0.f;
      }
      es_dirty_all = true;
      GO2_POST_STAMP(13);
    }
    GO2_MARK(24);
    // _push_robots (:709-724): episode clock multiple of the push interval (a fresh reset is pushed at once, App. E.4)
    if (c.push_robots && !S->initial_reset && (ep_len % c.push_interval == 0)) {
      if (!reset) fill(24, 2);                      // (a reset drew them with its own groups)
      o.vw.x = urange(uni(GO2_U_PUSH), -c.push_xy, c.push_xy); o.vw.y = urange(uni(GO2_U_PUSH + 1), -c.push_xy, c.push_xy);
      o.ww = v3(urange(uni(GO2_U_PUSH + 2), -c.push_ang, c.push_ang), urange(uni(GO2_U_PUSH + 3), -c.push_ang, c.push_ang), urange(uni(GO2_U_PUSH + 4), -c.push_ang, c.push_ang));
    }
    GO2_MARK(25);
    GO2_POST_STAMP(7);
    // ---- Go2Robot.compute_observations (go2_env.py:23-53) + clip (:96-99) -----------------------------
    // Every lane writes a share: sub-lanes 0 / 1 / 2 of a leg its joints' position / velocity / action entries (actor rows with noise,
    // critic rows without) plus the critic-only torque / acceleration / foot-force entries, sub-lane 3 of legs 0 / 1 / 2 the base angular
    // velocity / gravity / command triples, sub-lane 3 of leg 3 the critic's base linear velocity; the 187 height entries are dealt round
    // the 16 lanes.  The noise uniforms were drawn at kernel start, one Philox group per lane.
    if (!api_reset) {
    float cl = c.clip_obs;
    GO2_AS1 float* ob = (out.obs_out ? (GO2_AS1 float*)out.obs_out : p.obs) + (size_t)e * GO2_NUM_OBS;
    GO2_AS1 float* pv = (out.priv_out ? (GO2_AS1 float*)out.priv_out : p.priv) + (size_t)e * GO2_NUM_PRIV_OBS;
#define CLIP(x) fminf(fmaxf((x), -cl), cl)
    {
      float v3_[3], w3_[3]; int obase, pbase, wbase, nw;       // 3 values -> ob[obase..] (+noise) and pv[pbase..]; nw extra critic values -> pv[wbase..]
      if (sub == 0) { _Pragma("unroll") for (int j = 0; j < 3; ++j) { v3_[j] = (o.q[j] - c.q0[3 * lane + j]) * c.os_dof_pos; w3_[j] = o.tau[j] / t.eff_lim[j]; }
                      obase = 9 + 3 * lane; pbase = 12 + 3 * lane; wbase = 52 + 3 * lane; nw = 3; }
      else if (sub == 1) { _Pragma("unroll") for (int j = 0; j < 3; ++j) { v3_[j] = o.qd[j] * c.os_dof_vel; w3_[j] = (last_dv[j] - o.qd[j]) / c.dt * 1e-4f; }
                           obase = 21 + 3 * lane; pbase = 24 + 3 * lane; wbase = 64 + 3 * lane; nw = 3; }
      else if (sub == 2) { _Pragma("unroll") for (int j = 0; j < 3; ++j) { v3_[j] = act[j]; w3_[j] = 0.f; }
                           w3_[0] = sqrtf(dot(o.Ffoot, o.Ffoot)) * 1e-3f; obase = 33 + 3 * lane; pbase = 36 + 3 * lane; wbase = 48 + lane; nw = 1; }
      else {
        const V3 tr = lane == 0 ? c.os_ang * bav : (lane == 1 ? pg : (lane == 2 ? v3(cmd[0] * c.os_lin, cmd[1] * c.os_lin, cmd[2] * c.os_ang) : c.os_lin * blv));
        v3_[0] = tr.x; v3_[1] = tr.y; v3_[2] = tr.z; w3_[0] = w3_[1] = w3_[2] = 0.f;
        obase = lane < 3 ? 3 * lane : -1; pbase = lane < 3 ? 3 + 3 * lane : 0; wbase = 0; nw = 0;
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        pv[pbase + j] = CLIP(v3_[j]);
        if (obase >= 0) {
          const int i = obase + j; const float nv = c.noise_vec[i];
          const float nz = (c.add_noise && nv != 0.f) ? (2.f * uni(GO2_U_NOISE + i) - 1.f) * nv : 0.f;
          ob[i] = CLIP(v3_[j] + nz);
        }
        if (j < nw) pv[wbase + j] = CLIP(w3_[j]);
      }
    }
    if (c.terrain_mode == 0 || !c.measure_heights) {      // plane: every sample is 0 (:1201-1202), nothing to read back
      const float hv = CLIP(fminf(fmaxf(o.pw.z - 0.5f, -1.f), 1.f) * c.os_height);
      for (int i = lane16; i < GO2_NUM_HEIGHT_POINTS; i += 16) pv[76 + i] = hv;
    } else {
      for (int i = lane16; i < GO2_NUM_HEIGHT_POINTS; i += 16) {
        // the samples postA took at the pre-reset pose (App. E.3): this lane wrote exactly these entries of measured_heights there (and kept them in LDS)
        float hh = fminf(fmaxf(o.pw.z - 0.5f - (skip_contact_filters ? F2D(p.heights, i, e) : hc[i]), -1.f), 1.f);      // (a reset outside a step has no postA: the buffer's values)
        pv[76 + i] = CLIP(hh * c.os_height);
      }
    }
#undef CLIP
    }
    GO2_MARK(26);
    // ---- write back ---------------------------------------------------------------------------------
    if (sub == 0) {
      if (c.rew_on[GO2_REW_BASE_HEIGHT] && !skip_contact_filters) F2D(p.last_contacts2, lane, e) = new_lc2;
      if (c.rew_on[GO2_REW_FEET_AIR_TIME] && !skip_contact_filters) F2D(p.last_contacts, lane, e) = new_lc;
      if ((c.rew_on[GO2_REW_FEET_AIR_TIME] && !skip_contact_filters) || reset) F2D(p.feet_air_time, lane, e) = new_fat;
    }
    if (sub == 0)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      int d = 3 * lane + j;
      F2D(p.last_actions, d, e) = act[j];                      // :140 (zero for a reset env)
      F2D(p.last_last_actions, d, e) = llast_act[j];
      F2D(p.last_dof_vel, d, e) = o.qd[j];                     // :141
      F2D(p.actions, d, e) = act[j];
      F2D(p.dof, d, e) = o.q[j]; F2D(p.dof, 12 + d, e) = o.qd[j];
    }
    {   // episode_sums: the terms summed this step; after a reset every term (zeroed) — each lane its two
      const uint32_t wmask = es_dirty_all ? go2_uniform_u32(S->rew_mask_all) : rmask;
      if ((wmask >> own0) & 1u) es[(size_t)own0 * N + e] = es0;
      if (own1 < GO2_NUM_REWARDS && ((wmask >> own1) & 1u)) es[(size_t)own1 * N + e] = es1;
    }
    if (lane16 == 0) {
      float r13[13] = {o.pw.x, o.pw.y, o.pw.z, o.qx, o.qy, o.qz, o.qw, o.vw.x, o.vw.y, o.vw.z, o.ww.x, o.ww.y, o.ww.z};
      _Pragma("unroll") for (int k = 0; k < 13; ++k) F2D(p.root, k, e) = r13[k];
      if (!api_reset) _Pragma("unroll") for (int k = 0; k < 6; ++k) F2D(p.last_root_vel, k, e) = r13[7 + k];   // :142
      p.ep_len[e] = ep_len; p.cmd_timer[e] = timer;
      _Pragma("unroll") for (int k = 0; k < 4; ++k) F2D(p.commands, k, e) = cmd[k];
      F2D(p.cmd_xy_acc, 0, e) = acc[0]; F2D(p.cmd_xy_acc, 1, e) = acc[1];
      p.stop_heading[e] = stop_heading; p.last_is_limit_vel[e] = last_limit;
      p.reset[e] = reset; p.max_move[e] = max_move;
      if (c.turn_over) p.to_timer[e] = to_timer;
      if (!api_reset) {
      p.time_out[e] = time_out; p.rew[e] = total;
      if (out.rewards_out) ((GO2_AS1 float*)out.rewards_out)[e] = total + ((out.values && time_out) ? out.gamma * ((const GO2_AS1 float*)out.values)[e] : 0.f);   // ppo.py:107-108
      if (out.dones_out) ((GO2_AS1 uint8_t*)out.dones_out)[e] = reset;
      F2D(p.base_lin_vel, 0, e) = blv.x; F2D(p.base_lin_vel, 1, e) = blv.y; F2D(p.base_lin_vel, 2, e) = blv.z;
      F2D(p.base_ang_vel, 0, e) = bav.x; F2D(p.base_ang_vel, 1, e) = bav.y; F2D(p.base_ang_vel, 2, e) = bav.z;
      F2D(p.proj_gravity, 0, e) = pg.x; F2D(p.proj_gravity, 1, e) = pg.y; F2D(p.proj_gravity, 2, e) = pg.z;
      F2D(p.rpy, 0, e) = rpy[0]; F2D(p.rpy, 1, e) = rpy[1]; F2D(p.rpy, 2, e) = rpy[2];
      }
    }
  }
};
