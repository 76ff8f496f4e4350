// sgw_kernels.hpp -- the generic batched engine kernels, templated on a game family F.
//
// k_engine<F> covers sgw_reset (mode RESET), sgw_step (T = 1) and sgw_rollout (T > 1, state kept
// in registers across steps).  Restates, for 64 envs per wave in lockstep:
//   Environment.step auto-reset + max_iterations   pycolab_interface{,_mo}.py:147-192 / 157-196, 292-319
//   _process_timestep: episode return, termination reason default   safety_game.py:265-304,
//                                                                    safety_game_mo.py:971-1066
#pragma once

#include <cstddef>
#include <type_traits>

#include "sgw_common.hpp"
#include "sgw_pow.hpp"

namespace sgw {

constexpr int TERM_NONE4 = 15;   // 4-bit in-state encoding of "termination_reason key absent"

template <class F, class = void> struct has_idle_round : std::false_type {};
template <class F> struct has_idle_round<F, std::void_t<decltype(&F::idle_round)>> : std::true_type {};
template <class F, class = void> struct has_safety2 : std::false_type {};     // environment_data['safety2_<agent>'] (aintelope_savanna)
template <class F> struct has_safety2<F, std::void_t<decltype(&F::agent_safety2)>> : std::true_type {};
// families that work out every agent's safety value in one pass
template <class F, class = void> struct has_safety_all : std::false_type {};
template <class F> struct has_safety_all<F, std::void_t<decltype(&F::agent_safety_all)>> : std::true_type {};
template <class F>
__device__ inline void agent_safeties(const typename F::State& s, const KSpec& sp, const Lds& l, int (&out)[F::NA]) {
  if constexpr (has_safety_all<F>::value) {
    F::agent_safety_all(s, sp, l, out);
  } else {
#pragma unroll
    for (int ag = 0; ag < F::NA; ++ag) out[ag] = F::agent_safety(s, ag, sp);
  }
}
template <class F, class = void> struct has_init_issue : std::false_type {};
template <class F> struct has_init_issue<F, std::void_t<decltype(&F::init_issue)>> : std::true_type {};
// fused rollout: the COMPUTING wave re-reads the kernel arguments every step (see the loop) unless the family opts out
// (`ROLLOUT_KEEPS_ARGS`: measured faster and inside the register budget)
// one-step kernel: re-read the arguments AFTER the rules for the output phase, so that what only the outputs need (a dozen
// pointers, the LDS plan) is not held in SGPRs across play() -- for the families whose step kernel spills SGPRs (`STEP_REREADS_ARGS`)
template <class F, class = void> struct step_rereads : std::false_type {};
template <class F> struct step_rereads<F, std::void_t<decltype(F::STEP_REREADS_ARGS)>> : std::integral_constant<bool, F::STEP_REREADS_ARGS> {};
template <class F, class = void> struct rollout_rereads : std::true_type {};
template <class F> struct rollout_rereads<F, std::void_t<decltype(F::ROLLOUT_KEEPS_ARGS)>> : std::integral_constant<bool, !F::ROLLOUT_KEEPS_ARGS> {};
// families whose cumulative reward vector (NU doubles per lane: 52 VGPRs in aintelope_savanna) is PARKED IN LDS while the rules
// run: nothing in the rules reads it, so its registers are free for them (`static constexpr bool CUM_IN_LDS = true`)
template <class F, class = void> struct cum_in_lds : std::false_type {};
template <class F> struct cum_in_lds<F, std::void_t<decltype(F::CUM_IN_LDS)>> : std::integral_constant<bool, F::CUM_IN_LDS> {};
// (rows of the stash: one per enabled output column, A * K)
template <class F> __host__ __device__ inline int cum_stash_rows(int A, int K) { return cum_in_lds<F>::value ? A * K : 0; }
template <class F, class = void> struct has_init_args : std::false_type {};
template <class F> struct has_init_args<F, std::void_t<decltype(&F::init_args)>> : std::true_type {};
// families that write their row of the rendered board into the wave's LDS image themselves
template <class F, class = void> struct has_board_stage : std::false_type {};
template <class F> struct has_board_stage<F, std::void_t<decltype(&F::stage_board)>> : std::true_type {};

// ---- a step's outputs: stage (the computing wave, from registers into its LDS buffer) and drain (LDS -> global) ----------
template <class F> constexpr int per_agent_cols(const KSpec& sp) { return F::PER_AGENT ? sp.A : 1; }

// Every requested output of one step is written into the wave's staging buffer in the byte order of the env-major global
// arrays (the wave's 64 rows are contiguous there).  `finished` / `real`: the returns staging of the accumulators.
template <class F, bool SMALL = true, bool BOARD = true>
__device__ inline void emit_stage(const typename F::State& s, const double (&r)[F::NU], double discount, const KArgs& a,
                                  const Lds& l, int lane) {
  const sgw_out& o = a.out;
  const KSpec& sp = a.sp;
  const int nd = a.need;
  const int HW = sp.HW, K = sp.A * sp.K, M = sp.M;   // reward rows hold all agents' vectors: [A][K]
  if (BOARD && (nd & (LN_BOARD | LN_OBS | LN_VIEWS | LN_OBSVIEWS))) {
    if constexpr (has_board_stage<F>::value) {            // boards with dynamic content: the family writes its row itself
      F::stage_board(l, s, sp, lane);
    } else {
      static_assert(!F::CUSTOM_BOARD, "a family with a custom board provides stage_board()");
      int cells[F::NSPRITE]; uint8_t chars[F::NSPRITE];
      const uint8_t* base = F::board_layers(s, sp, l, cells, chars);
      lds_write_board_row<F::NSPRITE>(l.board, HW, lane, base, cells, chars);
    }
  }
  if (nd & LN_REWARD) {
    const StageRow row_r(l.vec_r, l.trash, lane, K);
#pragma unroll
    for (int u = 0; u < F::NU; ++u) *row_r.cell(F::slot(sp, u)) = r[u];
  }
  if (nd & LN_CUMULATIVE) {
    const StageRow row_c(l.vec_c, l.trash, lane, K);
#pragma unroll
    for (int u = 0; u < F::NU; ++u) *row_c.cell(F::slot(sp, u)) = s.cum[u];
  }
  if ((nd & LN_METRICS) && o.metrics && M > 0) {
#pragma unroll
    for (int id = 0; id < F::NMETRIC; ++id) *stage_cell(l.vec_m, l.trash, lane, M, sp.metric_slot[id]) = F::metric(s, id);
  }
  if constexpr (SMALL) {
  if (nd & LN_ST) {
#pragma unroll
    for (int ag = 0; ag < F::NA; ++ag) {
      if constexpr (F::PER_AGENT) l.st[lane * F::NA + ag] = (uint8_t)F::agent_step_type(s, ag);
      else l.st[lane * F::NA + ag] = (uint8_t)s.step_type;
    }
  }
  if (nd & LN_TR) {
    if constexpr (F::PER_AGENT) {
#pragma unroll
      for (int ag = 0; ag < F::NA; ++ag) l.tr[lane * F::NA + ag] = (uint8_t)F::agent_term(s, ag);
    } else {
      l.tr[lane] = (s.step_type == ST_LAST) ? (uint8_t)s.term : (uint8_t)SGW_TERM_NONE;
    }
  }
  if (nd & LN_ACT) {
#pragma unroll
    for (int ag = 0; ag < F::NA; ++ag) l.act[lane * F::NA + ag] = (int8_t)F::actual(s, ag);
  }
  if (nd & LN_POS) {
#pragma unroll
    for (int ag = 0; ag < F::NA; ++ag) {
      int pr, pc; F::agent_pos(s, ag, pr, pc);
      l.pos[(lane * F::NA + ag) * 2] = (uint8_t)pr; l.pos[(lane * F::NA + ag) * 2 + 1] = (uint8_t)pc;
    }
  }
  if (nd & LN_FLG) {
#pragma unroll
    for (int ag = 0; ag < F::NA; ++ag) l.flg[lane * F::NA + ag] = (uint8_t)F::agent_flags(s, ag);
  }
  if (nd & LN_DISC) l.disc[lane] = discount;
  if (nd & LN_HID) l.hid[lane] = F::hidden(s);
  if (nd & LN_SAF) {
    if constexpr (F::PER_AGENT) {
      int saf_all[F::NA];
      agent_safeties<F>(s, sp, l, saf_all);
#pragma unroll
      for (int ag = 0; ag < F::NA; ++ag) l.saf[lane * F::NA + ag] = (int32_t)saf_all[ag];
    } else {
      l.saf[lane] = F::safety(s);
    }
  }
  if (nd & LN_FRM) l.frm[lane] = s.frame;
  }
}

// The wave streams its 64 contiguous rows of every requested output from LDS, 16 B per lane per instruction, no waits in
// between.  coop == false (masked reset): each ACTIVE lane copies only its own rows.
template <class F, bool SMALL = true, bool BOARD = true>
__device__ inline void emit_drain(const KArgs& a, const Lds& l, long long env0, int lane, long long toff, bool coop,
                                  bool lane_active) {
  const sgw_out& o = a.out;
  const KSpec& sp = a.sp;
  const int nd = a.need;
  const int HW = sp.HW, K = sp.A * sp.K, M = sp.M, A = F::NA, PA = F::PER_AGENT ? F::NA : 1;
  const long long env = env0 + lane;
  auto rows = [&](void* base, long long row_bytes, const void* src) {     // base = the output array at time slice toff
    uint8_t* dst = reinterpret_cast<uint8_t*>(base) + toff * row_bytes;
    if (coop) coop_store(dst, env0, (int)row_bytes, src, lane);
    else if (lane_active) lane_store(dst, env, (int)row_bytes, src, lane);
  };
  if (BOARD && (nd & LN_BOARD)) rows(o.board, HW, l.board);
  if (BOARD && (nd & LN_OBS)) {                          // value_mapping LUT (rendering.py:491-549)
    float* dst = o.obs_board + (toff + env0) * HW;
    const uint8_t* img = reinterpret_cast<const uint8_t*>(l.board);
    if (coop) {
      float4* d4 = reinterpret_cast<float4*>(dst);
      for (int c = lane; c < 16 * HW; c += WAVE) {       // 64*HW cells, 4 per lane-iteration
        uint32_t q = l.board[c];
        d4[c] = make_float4(l.value_map[q & 0x7f], l.value_map[(q >> 8) & 0x7f],
                            l.value_map[(q >> 16) & 0x7f], l.value_map[(q >> 24) & 0x7f]);
      }
    } else if (lane_active) {
      for (int i = 0; i < HW; ++i) dst[(long long)lane * HW + i] = l.value_map[img[lane * HW + i] & 0x7f];
    }
  }
  if (nd & LN_REWARD) rows(o.reward, K * 8, l.vec_r);
  if (nd & LN_CUMULATIVE) rows(o.cumulative, K * 8, l.vec_c);
  if ((nd & LN_METRICS) && o.metrics && M > 0) rows(o.metrics, M * 8, l.vec_m);
  if constexpr (SMALL) {
  if (nd & LN_ST) rows(o.step_type, A, l.st);
  if (nd & LN_TR) rows(o.term_reason, PA, l.tr);
  if (nd & LN_ACT) rows(o.actual_action, A, l.act);
  if (nd & LN_POS) rows(o.agent_pos, 2 * A, l.pos);
  if (nd & LN_FLG) rows(o.agent_flags, A, l.flg);
  if (nd & LN_DISC) rows(o.discount, 8, l.disc);
  if (nd & LN_HID) rows(o.hidden, 8, l.hid);
  if (nd & LN_SAF) rows(o.safety, 4 * PA, l.saf);
  if (nd & LN_FRM) rows(o.frame, 4, l.frm);
  }
}

// The per-env scalar outputs straight from registers (one narrow store per lane each): the wave that computed the step
// also drains it, so the LDS round trip of the pipelined rollout would only add instructions here.
template <class F>
__device__ inline void emit_small_direct(const typename F::State& s, double discount, const KArgs& a, const Lds& l, long long env0, int lane,
                                         long long toff, bool active) {
  if (!active) return;
  const sgw_out& o = a.out;
  const KSpec& sp = a.sp;
  const int nd = a.need;
  const long long row = toff + env0 + lane;
  if (nd & LN_ST) {
#pragma unroll
    for (int ag = 0; ag < F::NA; ++ag) {
      if constexpr (F::PER_AGENT) store_wt(o.step_type + row * F::NA + ag, (uint8_t)F::agent_step_type(s, ag));
      else store_wt(o.step_type + row * F::NA + ag, (uint8_t)s.step_type);
    }
  }
  if (nd & LN_TR) {
    if constexpr (F::PER_AGENT) {
#pragma unroll
      for (int ag = 0; ag < F::NA; ++ag) store_wt(o.term_reason + row * F::NA + ag, (uint8_t)F::agent_term(s, ag));
    } else {
      store_wt(o.term_reason + row, (s.step_type == ST_LAST) ? (uint8_t)s.term : (uint8_t)SGW_TERM_NONE);
    }
  }
  if (nd & LN_ACT) {
#pragma unroll
    for (int ag = 0; ag < F::NA; ++ag) store_wt(o.actual_action + row * F::NA + ag, (int8_t)F::actual(s, ag));
  }
  if (nd & LN_POS) {
#pragma unroll
    for (int ag = 0; ag < F::NA; ++ag) {
      int pr, pc; F::agent_pos(s, ag, pr, pc);
      store_wt(o.agent_pos + (row * F::NA + ag) * 2, (uint8_t)pr); store_wt(o.agent_pos + (row * F::NA + ag) * 2 + 1, (uint8_t)pc);
    }
  }
  if (nd & LN_FLG) {
#pragma unroll
    for (int ag = 0; ag < F::NA; ++ag) store_wt(o.agent_flags + row * F::NA + ag, (uint8_t)F::agent_flags(s, ag));
  }
  if (nd & LN_DISC) store_wt(o.discount + row, discount);
  if (nd & LN_HID) store_wt(o.hidden + row, F::hidden(s));
  if (nd & LN_SAF) {
    if constexpr (F::PER_AGENT) {
      int saf_all[F::NA];
      agent_safeties<F>(s, sp, l, saf_all);
#pragma unroll
      for (int ag = 0; ag < F::NA; ++ag) store_wt(o.safety + row * F::NA + ag, (int32_t)saf_all[ag]);
    } else {
      store_wt(o.safety + row, F::safety(s));
    }
  }
  if constexpr (has_safety2<F>::value) {
    if (nd & LN_SAF2) {
#pragma unroll
      for (int ag = 0; ag < F::NA; ++ag) store_wt(o.safety2 + row * F::NA + ag, (int32_t)F::agent_safety2(s, ag, sp));
    }
  }
  if (nd & LN_FRM) store_wt(o.frame + row, s.frame);
}

// sgw_out.done / obs_dir / act_dir: what the wrappers compute from step_type and agent_flags every step, written by the wave that
// holds the state (one byte per lane and agent: a handful of narrow stores per wave, no staging).  The three pointers are read
// from the kernarg segment HERE, through a pointer the compiler cannot see through, and only when asked for: as ordinary
// members of the argument block they were six more SGPRs live across the rules of every step kernel (boat_race_ex: 2 -> 6 SGPR
// spills, 7.07 -> 7.40 us per launch at the mixed suite's shard size) whether or not a launch wanted them.
template <class F>
__device__ inline void emit_decodes_direct(const typename F::State& s, const KArgs& a, long long env0, int lane, long long toff, bool active,
                                           int kargs_off) {
  const int nd = a.need;
  if (!(nd & (LN_DONE | LN_ODIR | LN_ADIR)) || !active) return;
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const sgw_out __attribute__((address_space(4))) * OutSeg;
  OutSeg seg = (OutSeg)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + kargs_off + (int)offsetof(KArgs, out));
  asm volatile("" : "+s"(seg) : : "memory");
  struct { uint8_t *done, *obs_dir, *act_dir; } o = {seg->done, seg->obs_dir, seg->act_dir};
#else
  const sgw_out& o = a.out;
#endif
  const long long row = toff + env0 + lane;
#pragma unroll
  for (int ag = 0; ag < F::NA; ++ag) {
    int st;
    if constexpr (F::PER_AGENT) st = F::agent_step_type(s, ag); else st = s.step_type;
    if (nd & LN_DONE) store_wt(o.done + row * F::NA + ag, (uint8_t)(st >= ST_LAST ? 1 : 0));
    if (nd & (LN_ODIR | LN_ADIR)) {
      const int fl = F::agent_flags(s, ag);
      if (nd & LN_ODIR) store_wt(o.obs_dir + row * F::NA + ag, (uint8_t)((fl >> 3) & 3));
      if (nd & LN_ADIR) store_wt(o.act_dir + row * F::NA + ag, (uint8_t)((fl >> 1) & 3));
    }
  }
}

// workgroup barrier that orders LDS only: __syncthreads() also drains the wave's global stores (s_waitcnt vmcnt(0)),
// which is exactly what the draining wave of the pipelined rollout must not wait for
__device__ inline void lds_workgroup_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// ---- agent-centric windows inside the step launch (sgw_out.views / obs_views; get_agent_perspective, safety_game_moma.py:1996-2101) ----
// The wave's 64 rendered board rows are in LDS already (emit_stage); the windows are assembled next to them in an LDS image laid
// out exactly like the wave's 64 rows of the [N, view_total] output and leave as plain 16-byte-per-lane stores: no second
// launch, no re-read of the board from HBM, no byte-granular global stores.
template <class F, class = void> struct has_views : std::false_type {};
template <class F> struct has_views<F, std::void_t<decltype(F::VIEWS)>> : std::integral_constant<bool, F::VIEWS> {};
template <class F, class = void> struct has_view_dir : std::false_type {};
template <class F> struct has_view_dir<F, std::void_t<decltype(&F::view_dir)>> : std::true_type {};

// rot90 by the agent's observation direction (Directions LEFT=0 RIGHT=1 UP=2 DOWN=3): first crop, then rotate
// (safety_game_moma.py:2085-2096).  (vr, vc) of the OUTPUT -> (row, col) of the crop; square windows only.
__device__ inline void view_unrotate(int dir, int n, int& vr, int& vc) {
  const int r = vr, c = vc;
  if (dir == 3) { vr = n - 1 - r; vc = n - 1 - c; }          // DOWN: rot90 k=2
  else if (dir == 0) { vr = n - 1 - c; vc = r; }             // LEFT: rot90 k=-1 (clockwise)
  else if (dir == 1) { vr = c; vc = n - 1 - r; }             // RIGHT: rot90 k=1 (counter-clockwise)
}
// the inverse: (cr, cc) of the crop -> (vr, vc) of the output
__device__ inline void view_rotate(int dir, int n, int& cr, int& cc) {
  const int r = cr, c = cc;
  if (dir == 3) { cr = n - 1 - r; cc = n - 1 - c; }
  else if (dir == 0) { cr = c; cc = n - 1 - r; }
  else if (dir == 1) { cr = n - 1 - c; cc = r; }
}

// ONE wave assembles the windows of envs [e_lo, e_hi) of its env-wave, an env at a time with all 64 lanes.  Everything that
// does not depend on the env is worked out once per agent, outside the env loop:
//   * a window no larger than the board is gathered (a lane per output byte): the lane's window coordinates are constants,
//     an env adds its agent's position (v_readlane of lane e's registers -> scalars);
//   * a window LARGER than the board (firemaker's supervisor: 33 x 33 around 17 x 17) is mostly padding: the block is filled
//     with the pad byte first and the BOARD's cells are dropped where they land.  The landing offset of board cell (r, c) is
//     (r - pr) * n + (c - pc) unrotated and, rot90-ed by the observation direction, one of +-(r * n + c), +-(c * n - r) plus a
//     term that only depends on the env -- so a lane keeps two constants per cell and an env costs one scalar and one add
//     per cell.  When the window covers the board wherever the agent stands (radius >= board size - 1: the reference's
//     `None` radius) no cell can fall outside and the bounds test is skipped.
// ROT: the env has observation directions (a scalar property of the spec; the caller branches once): without them every `dir`
// below is the constant UP and the rotation arithmetic folds away (firemaker's default mode: 86 instead of 89 us per round).
template <class F, bool ROT>
__device__ inline void views_stage_wave_per_env(const typename F::State& s, const KSpec& sp, const Lds& l, int e_lo, int e_hi, int lane) {
  const int VB = sp.view_total, HW = sp.HW, W = sp.W, H = sp.H;
  const uint32_t pad = (uint32_t)sp.view_pad & 0xffu;
  uint8_t* img = l.views;
  const uint8_t* boards = reinterpret_cast<const uint8_t*>(l.board);
  if (sp.view_prefill) {                                     // (e_hi - e_lo) * VB is a multiple of 8 for blocks of 8 envs
    const uint64_t p8 = 0x0101010101010101ull * pad;
    uint64_t* blk = reinterpret_cast<uint64_t*>(img + e_lo * VB);
    const int n8 = ((e_hi - e_lo) * VB) >> 3;
    for (int i = lane; i < n8; i += WAVE) blk[i] = p8;
  }
  lds_wave_sync();
#pragma unroll
  for (int ag = 0; ag < F::NA; ++ag) {
    const int vh = sp.view_h[ag], vw = sp.view_w[ag], len = vh * vw;
    if (len == 0) continue;                                  // scalar: an agent without a window (absent firemaker agents)
    int prow, pcol, pdir = 2;
    F::agent_pos(s, ag, prow, pcol);
    if constexpr (ROT) pdir = F::view_dir(s, ag);
    const int up = sp.view_up[ag], left = sp.view_left[ag], off = sp.view_off[ag];
    if (len > HW) {
      // board cells k = lane + 64 j, j < 5 (H * W <= 320): row / column and the two rotation constants, once
      constexpr int NP = (SGW_MAX_CELLS + WAVE - 1) / WAVE;
      int rr[NP], cc[NP], P[NP], Q[NP];
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int k = lane + WAVE * j;
        rr[j] = (k * sp.recip_W) >> 16; cc[j] = k - rr[j] * W;
        P[j] = rr[j] * vw + cc[j]; Q[j] = cc[j] * vw - rr[j];
      }
      const bool covers = up >= H - 1 && vh - 1 - up >= H - 1 && left >= W - 1 && vw - 1 - left >= W - 1;
      const int npass = (HW + WAVE - 1) / WAVE;
      for (int e = e_lo; e < e_hi; ++e) {
        const int pr = __builtin_amdgcn_readlane(prow, e) - up, pc = __builtin_amdgcn_readlane(pcol, e) - left;
        const int dir = ROT ? __builtin_amdgcn_readlane(pdir, e) : 2;
        // at = sgn * (rotated ? Q : P) + base   (scalars per env)
        const int n1 = vw - 1;
        const int base = dir == 2 ? -(pr * vw + pc) : (dir == 3 ? (n1 + pr) * vw + n1 + pc : (dir == 0 ? n1 + pr - pc * vw : (n1 + pc) * vw - pr));
        const bool useQ = dir < 2, neg = dir == 3 || dir == 1;
        const uint8_t* src = boards + e * HW;
        uint8_t* dst = img + e * VB + off + base;
        uint8_t val[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) if (j < npass) val[j] = src[lane + WAVE * j < HW ? lane + WAVE * j : 0];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
          if (j < npass) {
            const int t = useQ ? Q[j] : P[j];
            const int at = neg ? -t : t;
            bool ok = lane + WAVE * j < HW;
            if (!covers) { const int cr = rr[j] - pr, c2 = cc[j] - pc; ok = ok && cr >= 0 && cr < vh && c2 >= 0 && c2 < vw; }
            if (ok) dst[at] = val[j];
          }
        }
      }
    } else {
      // output bytes k = lane + 64 j of the window: the lane's window coordinates, once
      constexpr int NP = (SGW_MAX_CELLS + WAVE - 1) / WAVE;
      const int npass = (len + WAVE - 1) / WAVE;
      int wr[NP], wc[NP];
#pragma unroll
      for (int j = 0; j < NP; ++j) { const int k = lane + WAVE * j; wr[j] = (k * (int)sp.view_recip[ag]) >> 16; wc[j] = k - wr[j] * vw; }
      for (int e = e_lo; e < e_hi; ++e) {
        const int pr = __builtin_amdgcn_readlane(prow, e) - up, pc = __builtin_amdgcn_readlane(pcol, e) - left;
        const int dir = ROT ? __builtin_amdgcn_readlane(pdir, e) : 2;
        const uint8_t* src = boards + e * HW;
        uint8_t* dst = img + e * VB + off;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
          if (j < npass) {
            int vr = wr[j], vc = wc[j];
            view_unrotate(dir, vw, vr, vc);                   // dir is a scalar: uniform branches
            const int r = vr + pr, c = vc + pc;
            const bool inside = r >= 0 && r < H && c >= 0 && c < W;
            const uint32_t got = src[inside ? r * W + c : 0];
            if (lane + WAVE * j < len) dst[lane + WAVE * j] = (uint8_t)(inside ? got : pad);
          }
        }
      }
    }
  }
}

// Families whose env-waves are single wavefronts (island_navigation_ex_ma, aintelope_savanna): a LANE assembles its own env's
// windows -- 64 envs x 2 small windows would be 128 serial wave passes the other way.  The window cell index is uniform over
// the wave (scalar row / column); only the agent's position and, where the env has observation directions, the rot90 are per
// lane.  Windows no larger than the board only (a larger one takes the wave-per-env path).
template <class F>
__device__ inline void views_stage_lane_per_env(const typename F::State& s, const KSpec& sp, const Lds& l, int lane) {
  const int VB = sp.view_total, HW = sp.HW, W = sp.W, H = sp.H;
  const uint32_t pad = (uint32_t)sp.view_pad & 0xffu;
  const uint8_t* src = reinterpret_cast<const uint8_t*>(l.board) + lane * HW;
  uint8_t* row = l.views + lane * VB;
#pragma unroll
  for (int ag = 0; ag < F::NA; ++ag) {
    const int vh = sp.view_h[ag], vw = sp.view_w[ag], len = vh * vw;
    if (len == 0) continue;
    int prow, pcol, dir = 2;
    F::agent_pos(s, ag, prow, pcol);
    if constexpr (has_view_dir<F>::value) dir = sp.view_rotates ? F::view_dir(s, ag) : 2;
    const int pr = prow - (int)sp.view_up[ag], pc = pcol - (int)sp.view_left[ag], n1 = vw - 1;
    uint8_t* dst = row + sp.view_off[ag];
    // Board cell of output cell (vr, vc), rot90 by the lane's direction (view_unrotate): LINEAR in the scalar window coordinates,
    //   r = r0 + vr * drr + vc * drv,  c = c0 + vr * dcr + vc * dcv
    // with per-lane constants worked out once per agent -- a cell then costs three multiply-adds, the bounds test and the two LDS
    // accesses (a lone wave per SIMD issues a dependent VALU instruction every ~8 cycles: the select chains of the first version,
    // ~25 instructions per cell, made 50 window bytes cost 11 of island_navigation_ex_ma's 28.6 us)
    const int r0 = pr + ((dir == 3 || dir == 0) ? n1 : 0), c0 = pc + ((dir == 3 || dir == 1) ? n1 : 0);
    const int drv = dir == 0 ? -1 : (dir == 1 ? 1 : 0), dcv = dir == 2 ? 1 : (dir == 3 ? -1 : 0);
    const int drr = dir == 2 ? 1 : (dir == 3 ? -1 : 0), dcr = dir == 0 ? 1 : (dir == 1 ? -1 : 0);
    int vr = 0, vc = 0;                                       // scalar window coordinates of cell k
    constexpr int NB = 8;                                     // cells per batch: eight independent LDS reads in flight, then the eight writes
    for (int k = 0; k < len; k += NB) {
      uint32_t got[NB]; bool inside[NB];
#pragma unroll
      for (int h = 0; h < NB; ++h) {
        const int r = r0 + vr * drr + vc * drv, c = c0 + vr * dcr + vc * dcv;
        inside[h] = (unsigned)r < (unsigned)H && (unsigned)c < (unsigned)W;
        got[h] = src[inside[h] ? r * W + c : 0];
        if (++vc == vw) { vc = 0; ++vr; }                     // (past the window's end the coordinates run on harmlessly: reads are clamped)
      }
#pragma unroll
      for (int h = 0; h < NB; ++h) if (k + h < len) dst[k + h] = (uint8_t)(inside[h] ? got[h] : pad);
    }
  }
}

// LDS view image -> global: `nthr` threads (thread `tid` of them) copy the env-wave's 64 contiguous rows, 16 bytes per lane and
// instruction; the float variant maps four window bytes to four floats per store (value_mapping LUT, rendering.py:491-549)
__device__ inline void views_drain(const KArgs& a, const Lds& l, long long env0, long long toff, int tid, int nthr) {
  const int VB = a.sp.view_total;
  if (a.need & LN_VIEWS) {
    uint4* g = reinterpret_cast<uint4*>(a.out.views + (toff + env0) * VB);
    const uint4* src = reinterpret_cast<const uint4*>(l.views);
    const int n = 4 * VB;
    int j = tid;
    for (; j + 3 * nthr < n; j += 4 * nthr) {                // four LDS reads in flight, then the four stores
      const uint4 v0 = src[j], v1 = src[j + nthr], v2 = src[j + 2 * nthr], v3 = src[j + 3 * nthr];
      store16_wt(g + j, v0); store16_wt(g + j + nthr, v1); store16_wt(g + j + 2 * nthr, v2); store16_wt(g + j + 3 * nthr, v3);
    }
    for (; j < n; j += nthr) store16_wt(g + j, src[j]);
  }
  if (a.need & LN_OBSVIEWS) {
    uint4* g = reinterpret_cast<uint4*>(a.out.obs_views + (toff + env0) * VB);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(l.views);
    const int n = 16 * VB;
    for (int j = tid; j < n; j += nthr) {
      const uint32_t q = src[j];
      const float f0 = l.value_map[q & 0x7f], f1 = l.value_map[(q >> 8) & 0x7f], f2 = l.value_map[(q >> 16) & 0x7f], f3 = l.value_map[(q >> 24) & 0x7f];
      store16_wt(g + j, make_uint4(__float_as_uint(f0), __float_as_uint(f1), __float_as_uint(f2), __float_as_uint(f3)));
    }
  }
}
// masked reset: only the rows of the envs that were reset, a wave per env (slow path)
__device__ inline void views_drain_env(const KArgs& a, const Lds& l, long long env0, int e, int lane) {
  const int VB = a.sp.view_total;
  const uint8_t* src = l.views + e * VB;
  if (a.need & LN_VIEWS) { uint8_t* g = a.out.views + (env0 + e) * VB; for (int k = lane; k < VB; k += WAVE) g[k] = src[k]; }
  if (a.need & LN_OBSVIEWS) { float* g = a.out.obs_views + (env0 + e) * VB; for (int k = lane; k < VB; k += WAVE) g[k] = l.value_map[src[k] & 0x7f]; }
}

// families whose workgroup's waves write the board rows together (cooperative families: every wave holds the same envs)
template <class F, class = void> struct has_board_part : std::false_type {};
template <class F> struct has_board_part<F, std::void_t<decltype(&F::stage_board_part)>> : std::true_type {};
// the env-wave's 64 board rows (and their value-mapped float twins) LDS -> global by `nthr` threads
__device__ inline void board_drain_wg(const KArgs& a, const Lds& l, long long env0, long long toff, int tid, int nthr) {
  const int HW = a.sp.HW;
  if (a.need & LN_BOARD) {
    uint4* g = reinterpret_cast<uint4*>(a.out.board + (toff + env0) * HW);
    const uint4* src = reinterpret_cast<const uint4*>(l.board);
    const int n = 4 * HW;
    int j = tid;
    for (; j + nthr < n; j += 2 * nthr) {
      const uint4 v0 = src[j], v1 = src[j + nthr];
      store16_wt(g + j, v0); store16_wt(g + j + nthr, v1);
    }
    if (j < n) store16_wt(g + j, src[j]);
  }
  if (a.need & LN_OBS) {
    uint4* g = reinterpret_cast<uint4*>(a.out.obs_board + (toff + env0) * HW);
    const int n = 16 * HW;
    for (int j = tid; j < n; j += nthr) {
      const uint32_t q = l.board[j];
      const float f0 = l.value_map[q & 0x7f], f1 = l.value_map[(q >> 8) & 0x7f], f2 = l.value_map[(q >> 16) & 0x7f], f3 = l.value_map[(q >> 24) & 0x7f];
      store16_wt(g + j, make_uint4(__float_as_uint(f0), __float_as_uint(f1), __float_as_uint(f2), __float_as_uint(f3)));
    }
  }
}

// The windows of one step, after the board rows are staged: cooperative families split the env-wave's 64 envs over the
// workgroup's F::WAVES waves (every wave holds the same state; two workgroup barriers: the leader's board rows are visible,
// the image is complete), the others do it within the wave.  `mask_on`: masked reset -- only the rows of the envs whose lane
// has `lane_active` are written.
template <class F, bool BOARD_DRAIN = false>
__device__ inline void views_phase(const typename F::State& s, const KArgs& a_in, uint8_t* smem, int slot, long long env0, int lane, long long t,
                                   bool mask_on, bool lane_active, unsigned kargs_off) {
  // the phase reads its arguments (window geometry, output pointers, LDS plan) from the kernarg segment itself, through a
  // pointer the compiler cannot see through: nothing of it is held in SGPRs across the rules
#if defined(__HIP_DEVICE_COMPILE__)
  KArgs a_seg;
  {
    typedef const KArgs __attribute__((address_space(4))) * KArgsSeg;
    KArgsSeg seg = (KArgsSeg)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + kargs_off);
    asm volatile("" : "+s"(seg) : : "memory");
    a_seg = *seg;
  }
  const KArgs& a = a_seg;
#else
  const KArgs& a = a_in;
#endif
  const Lds l = lds_carve(smem, a.lp, F::LDS_EXTRA, slot);
  const long long toff = a.write_every != 0 ? t * a.n_pad : 0;
  constexpr int NW = F::WAVES, EPW = WAVE / NW;
  static_assert(NW == 1 || (EPW % 8) == 0, "a wave's block of the view image must be 8-byte aligned");
  const int w = NW > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
  if constexpr (NW > 1) lds_workgroup_barrier(); else lds_wave_sync();
  // BOARD_DRAIN: the board rows also leave by the whole workgroup (their stores are in flight while the windows are assembled)
  if constexpr (BOARD_DRAIN) board_drain_wg(a, l, env0, toff, (int)threadIdx.x, NW * WAVE);
  if (!(a.need & (LN_VIEWS | LN_OBSVIEWS))) return;
  if (NW == 1 && !a.sp.view_prefill) views_stage_lane_per_env<F>(s, a.sp, l, lane);
  else if (has_view_dir<F>::value && a.sp.view_rotates) views_stage_wave_per_env<F, has_view_dir<F>::value>(s, a.sp, l, w * EPW, (w + 1) * EPW, lane);
  else views_stage_wave_per_env<F, false>(s, a.sp, l, w * EPW, (w + 1) * EPW, lane);
  if (!mask_on) {
    if constexpr (NW > 1) lds_workgroup_barrier(); else lds_wave_sync();
    views_drain(a, l, env0, toff, NW > 1 ? (int)threadIdx.x : lane, NW * WAVE);
  } else {
    lds_wave_sync();
    const unsigned long long on = __ballot(lane_active);
    for (int e = w * EPW; e < (w + 1) * EPW; ++e) if ((on >> e) & 1ull) views_drain_env(a, l, env0, e, lane);
  }
}

// The wave's accumulator row is updated with no-return f64 atomic adds executed at the memory side: nothing is loaded, so
// the wave never waits for the row (a load + store pair put a memory round trip on the critical path, and its pending
// load made the compiler drain vmcnt in the middle of the output staging).  Only this wave touches the row and it does so
// once per step, in program order: the sums are as deterministic as with plain stores.
__device__ inline void atomic_add_f64_noret(double* p, double v) {
#if defined(__HIP_DEVICE_COMPILE__)
  (void)__builtin_amdgcn_global_atomic_fadd_f64((__attribute__((address_space(1))) double*)p, v);
#else
  *p += v;
#endif
}
// SGW_ACC_PER_ENV: the lanes whose episode just ended add their return vector (and a 1 for the episode count) to their own
// cells of the [A*K+1][n_pad] accumulators: predicated no-return atomics, one writer per cell (deterministic), nothing to wait for
template <class F>
__device__ inline void accumulate_returns_per_env(const typename F::State& s, const KArgs& a, long long env, bool mine) {
  if (!mine) return;
  const int C = a.sp.A * a.sp.K + 1;
#pragma unroll
  for (int u = 0; u < F::NU; ++u) {
    const int q = F::slot(a.sp, u);
    if (q >= 0) atomic_add_f64_noret(&a.ep_acc[(long long)q * a.n_pad + env], s.cum[u]);
  }
  atomic_add_f64_noret(&a.ep_acc[(long long)(C - 1) * a.n_pad + env], 1.0);
}

// Episodic-return accumulators (end-of-batch all-reduce buffer).  Lanes whose episode just ended have staged their return
// vector in l.vec_a (zeros otherwise).  lane = part * 16 + column: each lane sums one column over its part's 16 rows (reads
// batched, a fixed add tree) and adds the sum to the part's own accumulator row -- four rows per env-wave, no exchange between
// lanes; 16 columns per pass (one pass for C <= 16: every single-agent family and firemaker; island_navigation_ex_ma has
// 2K + 1 = 17..25).  Deterministic: every accumulator cell has one writer per launch and launches are ordered.  Traffic: one
// 8-byte RMW per column per 16 envs.
__device__ inline void accumulate_returns(const KArgs& a, const Lds& l, long long wave_id, long long env0, int lane) {
  const int C = a.sp.A * a.sp.K + 1;
  const int part = lane >> 4;
  for (int c0 = 0; c0 < C; c0 += 16) {
    const int col = c0 + (lane & 15);
    double v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int row = part * 16 + j;
      v[j] = (col < C && env0 + row < a.n_envs) ? l.vec_a[row * C + col] : 0.0;
    }
    // a fixed tree over the part's 16 rows (four add levels instead of a chain of sixteen); the four parts keep their own
    // accumulator rows -- no cross-lane exchange -- and k_read_returns adds the rows up in a fixed order
#pragma unroll
    for (int w = 8; w >= 1; w >>= 1) {
#pragma unroll
      for (int j = 0; j < w; ++j) v[j] += v[j + w];
    }
    if (col < C) atomic_add_f64_noret(&a.ep_acc[(wave_id * SGW_ACC_PARTS + part) * C + col], v[0]);
  }
}

// KIND: K_STEP = exactly one step (sgw_step / sgw_step_n), K_ROLLOUT = a.T fused steps, K_RESET = sgw_reset.
// Separate instantiations keep the single-step kernel free of loop-carried scalar state.
enum { K_STEP = 0, K_ROLLOUT = 1, K_RESET = 2 };

// F::WAVES > 1 (cooperative families, firemaker): the workgroup's 64 envs are REPLICATED in every wave -- each wave
// runs the same lane-per-env code on the same state, so control flow (and every s_barrier) is identical across
// the waves -- and the family splits its wave-cooperative phase (one env at a time, one lane per board cell) over
// the waves.  Only wave 0 ("leader") writes outputs, accumulators and state.
//
// Non-cooperative families: the workgroup holds ENV_WAVES independent env-waves (64 envs each, one per SIMD) that share ONE
// LDS copy of the level tables and of the family's read-only tables.  A family may lower the count
// (`static constexpr int ENV_WAVES_MAX`) when four waves' worth of output staging would not fit the CU's 160 KiB of LDS
// with every output requested (island_navigation_ex_ma: 2, aintelope_savanna: 1).
//
// PIPELINED fused rollout (K_ROLLOUT, non-cooperative): every env-wave is a PAIR of wavefronts.  The computing wave keeps
// the state in registers, plays step t and stages its outputs into LDS buffer t & 1; the draining wave copies buffer t & 1
// to global memory and folds the finished episodes into the accumulators while the computing wave is already playing
// step t + 1.  One LDS-only workgroup barrier per step hands a buffer over.  A lone wave issues one instruction per ~4
// cycles whatever it does, and rules and output copy are about as many instructions each: two waves overlap them.
template <class F, class = void> struct family_env_waves { static constexpr int value = ENV_WAVES; };
template <class F> struct family_env_waves<F, std::void_t<decltype(F::ENV_WAVES_MAX)>> { static constexpr int value = F::ENV_WAVES_MAX; };
template <class F, class = void> struct family_pipelines : std::true_type {};
template <class F> struct family_pipelines<F, std::void_t<decltype(F::ROLLOUT_PIPELINED)>> : std::integral_constant<bool, F::ROLLOUT_PIPELINED> {};
// (K_STEP as a pair of wavefronts was measured too: 7.9 us instead of 6.8 at 65 536 envs, 0.53 instead of 0.66 of the roofline at
// 1 M -- the second wave's start-up and the hand-over barrier cost more than the store issue it takes off the first)
template <class F, int KIND> constexpr bool pipelined() { return KIND == K_ROLLOUT && !F::COOPERATIVE && family_pipelines<F>::value; }
// a pipelined workgroup holds at most two env-waves = four wavefronts, one per SIMD, each with the full register file
static_assert(true, "");
template <class F, int KIND> constexpr int env_waves() {
  return F::COOPERATIVE ? 1 : (pipelined<F, KIND>() ? (family_env_waves<F>::value < 2 ? family_env_waves<F>::value : 2) : family_env_waves<F>::value);
}
template <class F, int KIND> constexpr int lds_buffers() { return (pipelined<F, KIND>() && KIND == K_ROLLOUT) ? 2 : 1; }
template <class F, int KIND> constexpr int wg_threads() { return F::WAVES * env_waves<F, KIND>() * WAVE * (pipelined<F, KIND>() ? 2 : 1); }

#ifdef SGW_WAVES_PER_EU      // experiment: force an occupancy target (registers beyond it go to scratch)
#define SGW_OCC __attribute__((amdgpu_waves_per_eu(SGW_WAVES_PER_EU, SGW_WAVES_PER_EU)))
#else
#define SGW_OCC
#endif
// The engine's body: workgroup `block` of a launch whose KArgs block sits `kargs_off` bytes into the kernarg segment (the
// re-reading paths below go back to it there).  k_engine is this and nothing else; k_engine_group runs it for the member of a
// heterogeneous launch that the workgroup belongs to.
template <class F, int KIND>
__device__ __forceinline__ void engine_body(const KArgs& a, const long long block, const unsigned kargs_off) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  constexpr int EW = env_waves<F, KIND>();
  constexpr bool PIPE = pipelined<F, KIND>();
  constexpr int NB = lds_buffers<F, KIND>();
  constexpr int NT = wg_threads<F, KIND>();
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave_in_wg = (F::COOPERATIVE || (EW == 1 && !PIPE)) ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wv = PIPE ? (wave_in_wg >= EW ? wave_in_wg - EW : wave_in_wg) : wave_in_wg;       // env-wave within the workgroup
  const bool drainer = PIPE && wave_in_wg >= EW;
  const bool leader = F::WAVES == 1 || threadIdx.x < WAVE;
  const long long wave_id = block * EW + wv;
  const long long env0 = wave_id * WAVE;
  const long long env = env0 + lane;
  const long long env_id = a.env_id_base + env;
  const bool real = env < a.n_envs;
  const bool wave_live = env0 < a.n_pad;                 // the last workgroup may hold fewer than EW env-waves
#ifdef SGW_STAMPS
  const long long sgw_stamp_wave = wave_live ? wave_id : 0;
#endif

  SGW_STAMP_RT(a, 6);
  SGW_STAMP(a, 0);
  // every global load of the prologue is ISSUED before anything waits (level tables, the family's tables, the env's state,
  // the first actions: one memory round trip, not four); LDS is written afterwards
  TableStage ts;
  lds_tables_issue<NT>(ts, a.tables);
  const Lds l = lds_carve(smem, a.lp, F::LDS_EXTRA, wv * NB);
  typename F::Ctx cx;
  if constexpr (has_init_issue<F>::value) F::init_issue(cx);
  // no control flow up to the barrier: a dead env-wave (past n_pad in the last workgroup) loads env-wave 0's state and the
  // first table bytes instead of branching around its loads, so the compiler keeps every load of the prologue in one block
  typename F::State s;
  F::load(s, a, wave_live ? env : (long long)lane);
  int action0[F::NA];
#pragma unroll
  for (int ag = 0; ag < F::NA; ++ag) {
    const bool have = KIND != K_RESET && a.actions != nullptr && real;
    const int8_t* ap = have ? a.actions + env * F::NA + ag : reinterpret_cast<const int8_t*>(a.tables);
    const int v = (int)*ap;
    action0[ag] = have ? v : 0;
  }
  lds_tables_commit<NT>(ts, smem);
  F::init_ctx(cx, l);
  if constexpr (has_init_args<F>::value) F::init_args(cx, l, a);
  // pipelined rollout with synthetic actions: the Philox stream (~100 instructions per agent and step) is the draining wave's
  // work -- it writes step t + 2's actions into buffer t & 1's inbox while the computing wave plays step t + 1
  if constexpr (PIPE && KIND == K_ROLLOUT) {
    if (drainer && a.actions == nullptr) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const Lds lb = lds_carve(smem, a.lp, F::LDS_EXTRA, wv * NB + b);
#pragma unroll
        for (int ag = 0; ag < F::NA; ++ag)
          lb.ain[ag * WAVE + lane] = (int8_t)synth_action(a.seed, env_id, a.step0 + b, ag, a.sp.action_lo, a.sp.n_actions);
      }
    }
  }
  __syncthreads();
  const int TT = (KIND == K_STEP) ? 1 : a.T;
  if constexpr (PIPE) {
    if (!wave_live || drainer) {
      // the draining wave (and a dead pair, which only keeps the barrier count): one barrier per step hands over buffer t & 1
      for (int t = 0; t < TT; ++t) {
        lds_workgroup_barrier();
        if (!wave_live) continue;
        KArgs a_step;
#if defined(__HIP_DEVICE_COMPILE__)
        {
          typedef const KArgs __attribute__((address_space(4))) * KArgsSeg;
          KArgsSeg seg = (KArgsSeg)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + kargs_off);
          asm volatile("" : "+s"(seg) : : "memory");
          a_step = *seg;
        }
#endif
        const Lds lb = lds_carve(smem, a_step.lp, F::LDS_EXTRA, wv * NB + (NB > 1 ? (t & 1) : 0));
        if (a_step.write_every != 0 || t == TT - 1)
          emit_drain<F>(a_step, lb, env0, lane, a_step.write_every != 0 ? (long long)t * a_step.n_pad : 0, true, true);
        if (!SGW_ACC_PER_ENV && (a_step.need & LN_RETURNS) && lb.flag[0] != 0u) accumulate_returns(a_step, lb, wave_id, env0, lane);
        if (a_step.actions == nullptr && t + 2 < TT) {      // the computing wave read this inbox before the barrier above
#pragma unroll
          for (int ag = 0; ag < F::NA; ++ag)
            lb.ain[ag * WAVE + lane] = (int8_t)synth_action(a_step.seed, a_step.env_id_base + env, a_step.step0 + t + 2, ag,
                                                            a_step.sp.action_lo, a_step.sp.n_actions);
        }
      }
      return;
    }
  } else {
    if (!wave_live) return;
  }
  SGW_STAMP(a, 1);

  if (KIND == K_RESET) {
    const bool m = a.mask ? (real && a.mask[env] != 0) : true;
    double r[F::NU];
#pragma unroll
    for (int u = 0; u < F::NU; ++u) r[u] = 0.0;
    if (m) { F::begin_episode(s, a, l, env, env_id); if (leader) F::store(s, a, env); }
    if (leader) {
      lds_wave_sync();
      emit_stage<F, false>(s, r, __longlong_as_double(0x7ff8000000000000LL), a, l, lane);
    }
    if constexpr (has_views<F>::value) {
      if (a.need & (LN_VIEWS | LN_OBSVIEWS)) views_phase<F>(s, a, smem, wv * NB, env0, lane, 0, a.mask != nullptr, m, kargs_off);
    }
    if (leader) {
      lds_wave_sync();
      emit_drain<F, false>(a, l, env0, lane, 0, a.mask == nullptr, m);
      emit_small_direct<F>(s, __longlong_as_double(0x7ff8000000000000LL), a, l, env0, lane, 0, a.mask == nullptr || m);
      emit_decodes_direct<F>(s, a, env0, lane, 0, a.mask == nullptr || m, kargs_off);
    }
    return;
  }

  const KArgs& a_launch = a;
  const Lds& l_launch = l;
  for (int t = 0; t < TT; ++t) {
    // Fused rollout: nothing of the ARGUMENTS may stay live across the back-edge.  The loop would otherwise keep the whole
    // kernarg block (110 dwords) and everything derived from it in SGPRs -- every K_ROLLOUT instantiation sat at the 106-SGPR
    // ceiling with 200-630 SGPR spills -- so each step re-reads what it needs from the kernarg segment through a pointer the
    // compiler cannot see through (scalar loads, scalar cache hits).  The memory clobber does the same for loop-invariant
    // LDS reads (the family constants: 76 VGPRs in island_navigation_ex_ma).
    KArgs a_step;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (KIND == K_ROLLOUT && rollout_rereads<F>::value) {
      typedef const KArgs __attribute__((address_space(4))) * KArgsSeg;
      KArgsSeg seg = (KArgsSeg)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + kargs_off);
      asm volatile("" : "+s"(seg) : : "memory");
      a_step = *seg;
    } else if constexpr (KIND == K_ROLLOUT) {
      asm volatile("" ::: "memory");
    }
#endif
    const KArgs& a = (KIND == K_ROLLOUT && rollout_rereads<F>::value) ? a_step : a_launch;
    // ... and the LDS carve (a dozen region addresses) is re-derived from this step's arguments for the same reason
    const Lds l_step = lds_carve(smem, a.lp, F::LDS_EXTRA, wv * NB + (NB > 1 ? (t & 1) : 0));
    const Lds& l = KIND == K_ROLLOUT ? l_step : l_launch;   // (K_ROLLOUT: also selects the buffer t & 1)
    double r[F::NU];
#pragma unroll
    for (int u = 0; u < F::NU; ++u) r[u] = 0.0;
    double discount = __longlong_as_double(0x7ff8000000000000LL);   // None at FIRST
    bool over_now = false;
    if constexpr (cum_in_lds<F>::value) {                           // park the cumulative vector (re-read below, on every path)
#pragma unroll
      for (int u = 0; u < F::NU; ++u) { const int q = F::slot(a.sp, u); if (q >= 0) l.cstash[q * WAVE + lane] = s.cum[u]; }   // (a dimension that is not enabled stays 0)
    }
    if constexpr (!F::COOPERATIVE) {
      int action[F::NA];
#pragma unroll
      for (int ag = 0; ag < F::NA; ++ag) {
        if constexpr (KIND == K_STEP) action[ag] = action0[ag];          // sgw_step / sgw_step_n: the caller's actions, always
        else if (a.actions) action[ag] = (t == 0) ? action0[ag]
                                                  : (real ? (int)a.actions[((long long)t * a.n_envs + env) * F::NA + ag] : 0);
        else if constexpr (PIPE) action[ag] = (int)l.ain[ag * WAVE + lane];     // the draining wave's Philox (see the prologue)
        else action[ag] = synth_action(a.seed, env_id, a.step0 + t, ag, a.sp.action_lo, a.sp.n_actions);
      }
      bool idle = false;
      if constexpr (has_idle_round<F>::value) idle = s.step_type >= ST_LAST && !F::reset_requested(s, a, action);
      if (idle) {
        // multi-agent adapter, finished episode, no eligible agent in the submitted dict (PM:173-246: the play loop does not
        // run, so nothing resets): only the per-agent states move on (LAST -> DEAD); rewards are the default zeros
        if constexpr (has_idle_round<F>::value) discount = F::idle_round(s);
        if constexpr (cum_in_lds<F>::value) {
#pragma unroll
          for (int u = 0; u < F::NU; ++u) { const int q = F::slot(a.sp, u); s.cum[u] = q >= 0 ? l.cstash[q * WAVE + lane] : 0.0; }
        }
      } else if (s.step_type >= ST_LAST) {
        // step after LAST (or before any reset): new episode, action discarded (pycolab_interface_mo.py:175-178); the
        // multi-agent adapters still shuffle the discarded actions when more than one was submitted
        F::pre_autoreset(s, a, action);
        F::begin_episode(s, a, l, env, env_id);
      } else {
        discount = F::play(s, action, a, l, r, env);
        const bool over = (discount == 0.0) || (s.frame >= a.sp.max_iterations);   // pycolab_interface.py:292-303
        s.step_type = over ? ST_LAST : ST_MID;
        if (over && s.term == TERM_NONE4) s.term = SGW_MAX_STEPS;                   // safety_game.py:294-296
#pragma unroll
        for (int u = 0; u < F::NU; ++u) {                                          // safety_game_mo.py:996-997
          if constexpr (cum_in_lds<F>::value) { const int q = F::slot(a.sp, u); s.cum[u] = (q >= 0 ? l.cstash[q * WAVE + lane] : 0.0) + r[u]; }
          else s.cum[u] += r[u];
        }
        over_now = over;
      }
    } else {
      // cooperative families run play() with every lane active (their wave-wide phases need full EXEC and every wave
      // must reach the same barriers); lanes that auto-reset this step pass live = false and change nothing
      const bool resetting = s.step_type >= ST_LAST;
      int action[F::NA];
#pragma unroll
      for (int ag = 0; ag < F::NA; ++ag) {
        if constexpr (KIND == K_STEP) action[ag] = action0[ag];          // sgw_step / sgw_step_n: the caller's actions, always
        else if (a.actions) action[ag] = (t == 0) ? action0[ag]
                                                  : (real ? (int)a.actions[((long long)t * a.n_envs + env) * F::NA + ag] : 0);
        else action[ag] = synth_action(a.seed, env_id, a.step0 + t, ag, a.sp.action_lo, a.sp.n_actions);
      }
      if (resetting) { F::pre_autoreset(s, a, action); F::begin_episode(s, a, l, env, env_id); }
      const double d = F::play(s, action, a, l, r, env, !resetting, cx);
      if (!resetting) {
        discount = d;
        const bool over = (discount == 0.0) || (s.frame >= a.sp.max_iterations);
        s.step_type = over ? ST_LAST : ST_MID;
        if (over && s.term == TERM_NONE4) s.term = SGW_MAX_STEPS;
#pragma unroll
        for (int u = 0; u < F::NU; ++u) s.cum[u] += r[u];
        over_now = over;
      }
    }
    SGW_STAMP(a, 2);
    // ---- this step's outputs and finished-episode returns go into the wave's staging buffer ...
    KArgs a_out;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (KIND == K_STEP && step_rereads<F>::value) {
      typedef const KArgs __attribute__((address_space(4))) * KArgsSeg;
      KArgsSeg seg = (KArgsSeg)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + kargs_off);
      asm volatile("" : "+s"(seg) : : "memory");
      a_out = *seg;
    }
#endif
    const KArgs& ae = (KIND == K_STEP && step_rereads<F>::value) ? a_out : a;
    const Lds l_out = lds_carve(smem, ae.lp, F::LDS_EXTRA, wv * NB + (NB > 1 ? (t & 1) : 0));
    const Lds& le = (KIND == K_STEP && step_rereads<F>::value) ? l_out : l;
    const int C = ae.sp.A * ae.sp.K + 1;
    const bool last_t = (t == TT - 1);
    const bool writes = ae.write_every != 0 || last_t;
    bool acc_any = false;
    if (leader) {
      if constexpr (!PIPE) lds_wave_sync();                 // the previous step's cooperative reads are done (program order)
      if (SGW_ACC_PER_ENV && (ae.need & LN_RETURNS)) {
        if (__ballot(over_now && real) != 0ull) accumulate_returns_per_env<F>(s, ae, env, over_now && real);
      } else if (ae.need & LN_RETURNS) {
        acc_any = __ballot(over_now && real) != 0ull;       // wave-uniform
        if (acc_any) {
          const StageRow row_a(le.vec_a, le.trash, lane, C);
#pragma unroll
          for (int u = 0; u < F::NU; ++u) *row_a.cell(F::slot(ae.sp, u)) = over_now ? s.cum[u] : 0.0;
          le.vec_a[lane * C + C - 1] = over_now ? 1.0 : 0.0;
        }
        if constexpr (PIPE) { if (lane == 0) le.flag[0] = acc_any ? 1u : 0u; }
      }
      if (writes) {
        emit_stage<F, PIPE, !has_board_part<F>::value>(s, r, discount, ae, le, lane);
        emit_decodes_direct<F>(s, ae, env0, lane, ae.write_every != 0 ? (long long)t * ae.n_pad : 0, true, kargs_off);
      }
    }
    if constexpr (has_board_part<F>::value) {
      // cooperative family: the WORKGROUP produces the step's big outputs -- every wave writes a slice of the board rows, then
      // (one barrier) a share of the board's and (a second barrier) of the windows' stores; the leader keeps the rest
      static_assert(!PIPE && has_views<F>::value && F::COOPERATIVE, "");
      if (writes && (ae.need & (LN_BOARD | LN_OBS | LN_VIEWS | LN_OBSVIEWS))) {
        F::stage_board_part(le, s, ae.sp, lane, __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)));
        views_phase<F, true>(s, ae, smem, wv * NB, env0, lane, t, false, true, kargs_off);
      }
    } else if constexpr (has_views<F>::value) {
      static_assert(!PIPE, "the pipelined rollout has no window phase (the families with agent views do not pipeline)");
      if (writes && (ae.need & (LN_VIEWS | LN_OBSVIEWS)))
        views_phase<F>(s, ae, smem, wv * NB, env0, lane, t, false, true, kargs_off);
    }
    // ... and leave it: the pair's draining wave takes the buffer over at the barrier (pipelined rollout), or this wave
    // copies it out itself
    if constexpr (PIPE) {
      lds_workgroup_barrier();
    } else if (leader) {
      lds_wave_sync();
      if (writes) {
        emit_drain<F, false, !has_board_part<F>::value>(ae, le, env0, lane, ae.write_every != 0 ? (long long)t * ae.n_pad : 0, true, true);
        emit_small_direct<F>(s, discount, ae, le, env0, lane, ae.write_every != 0 ? (long long)t * ae.n_pad : 0, true);
      }
      SGW_STAMP(ae, 3);
      if (acc_any) accumulate_returns(ae, le, wave_id, env0, lane);
    }
    if constexpr (KIND == K_STEP && step_rereads<F>::value) { if (leader) F::store(s, ae, env); return; }
  }
  SGW_STAMP(a, 4);
  if (leader) F::store(s, a, env);
  SGW_STAMP(a, 5);
  SGW_STAMP_RT(a, 7);
}

template <class F, int KIND>
__global__ SGW_OCC __launch_bounds__((wg_threads<F, KIND>())) void k_engine(uint64_t* hot_state, const uint8_t* hot_tables, const int8_t* hot_actions,
                                                                      long long hot_n_pad, long long hot_n_envs, int hot_words,
                                                                      const KArgs a_in) {
  // the leading scalar arguments repeat the few values the prologue's loads need (state / tables / actions pointers, sizes):
  // as LEADING kernel arguments they are preloaded into SGPRs at wave launch (-mllvm -amdgpu-kernarg-preload-count), so
  // the first global loads issue without waiting for a scalar load of the kernarg segment from memory
  KArgs a = a_in;
  a.state = hot_state; a.tables = hot_tables; a.actions = hot_actions; a.n_pad = hot_n_pad; a.n_envs = hot_n_envs; a.sp.words = hot_words;
  engine_body<F, KIND>(a, (long long)blockIdx.x, SGW_KARGS_OFFSET);
}

// ---- heterogeneous launch: several engines (env families) in ONE grid --------------------------------------------------
// A mixed suite sharded over one GPU (BASELINE config 5: island_navigation_ex + boat_race_ex + safe_interruptibility) is three
// small launches otherwise -- 171 workgroups each on a 256-CU chip, three kernel boundaries per step.  Here workgroup b belongs
// to member m with first_block[m] <= b < first_block[m + 1]; it reads that member's KArgs from the kernarg segment and runs the
// member's family body.  Members are families whose workgroups have the same shape (GROUP_THREADS threads: four env-waves per
// step workgroup, two wave pairs per fused-rollout workgroup); the launch's dynamic LDS is the largest member's.
constexpr int GROUP_MAX = 4, GROUP_THREADS = 256;
enum FamilyTag { TAG_ISLAND_GENERAL, TAG_ISLAND_PACKED, TAG_ISLAND, TAG_BOAT, TAG_SAFEINT, TAG_FIREMAKER, TAG_ISLAND_MA, TAG_TILE, TAG_SOKOBAN,
                 TAG_CONVEYOR, TAG_TOMATO, TAG_FRIEND_FOE, TAG_WHISKY, TAG_ROCKS, TAG_SAVANNA };
struct GroupArgs { int n, pad_; int first_block[GROUP_MAX + 1]; int tag[GROUP_MAX]; int pad2_; KArgs a[GROUP_MAX]; };
template <class F, int KIND> constexpr bool group_member() { return !F::COOPERATIVE && wg_threads<F, KIND>() == GROUP_THREADS; }

// SGW_KARGS_OFFSET (where the fused rollout / the re-reading step kernels find the KArgs block in the kernarg segment) is tied
// to k_engine's parameter list: a struct with the same members in the same order has the same layout as the segment
struct EngineKernargMirror { uint64_t* hot_state; const uint8_t* hot_tables; const int8_t* hot_actions; long long hot_n_pad, hot_n_envs; int hot_words; KArgs a_in; };
static_assert(offsetof(EngineKernargMirror, a_in) == SGW_KARGS_OFFSET, "SGW_KARGS_OFFSET does not match k_engine's leading arguments (SGW_HOT_ARGS)");

// synthetic action stream materialised in HBM: int8 [T, N, A]
__global__ void k_fill_actions(int8_t* out, long long n, int A, int T, unsigned long long seed, long long step0,
                               long long env_id_base, int lo, int nact) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)T * n * A;
  for (; i < total; i += (long long)gridDim.x * blockDim.x) {
    int ag = (int)(i % A);
    long long e = (i / A) % n;
    long long t = i / (A * n);
    out[i] = (int8_t)synth_action(seed, env_id_base + e, step0 + t, ag, lo, nact);
  }
}

// out[c] = sum over waves of acc[wave][c]; one workgroup per column, fixed-order tree => deterministic
__global__ __launch_bounds__(256) void k_read_returns(double* acc, long long n_waves, int C, double* out, int clear) {
  __shared__ double part[256];
  const int c = blockIdx.x;
  double v = 0.0;
#if SGW_ACC_PER_ENV                                           // [C][rows]: a column is contiguous
  for (long long w = threadIdx.x; w < n_waves; w += 256) v += acc[(long long)c * n_waves + w];
  part[threadIdx.x] = v;
  __syncthreads();
  for (int s2 = 128; s2 > 0; s2 >>= 1) {
    if ((int)threadIdx.x < s2) part[threadIdx.x] += part[threadIdx.x + s2];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[c] = part[0];
  if (clear) for (long long w = threadIdx.x; w < n_waves; w += 256) acc[(long long)c * n_waves + w] = 0.0;
  return;
#endif
  for (long long w = threadIdx.x; w < n_waves; w += 256) v += acc[w * C + c];
  part[threadIdx.x] = v;
  __syncthreads();
  for (int s2 = 128; s2 > 0; s2 >>= 1) {
    if ((int)threadIdx.x < s2) part[threadIdx.x] += part[threadIdx.x + s2];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[c] = part[0];
  if (clear) for (long long w = threadIdx.x; w < n_waves; w += 256) acc[w * C + c] = 0.0;
}

// (sum of episode returns, #episodes) over envs whose step_type is LAST
__global__ void k_accumulate_returns(const double* cumulative, const uint8_t* step_type, long long n, int AK,
                                     double* accum) {
  long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool last = e < n && step_type[e] == ST_LAST;
  for (int k = 0; k < AK; ++k) {
    double v = wave_sum(last ? cumulative[e * AK + k] : 0.0);
    if ((threadIdx.x & (WAVE - 1)) == 0 && v != 0.0) atomicAdd(&accum[k], v);
  }
  double c = wave_sum(last ? 1.0 : 0.0);
  if ((threadIdx.x & (WAVE - 1)) == 0 && c != 0.0) atomicAdd(&accum[AK], c);
}

// _episodic_performances bookkeeping per env (safety_game.py:194-263): at a LAST timestep the episode's performance becomes the
// env's last performance and joins its running sum and count
__global__ void k_track_performance(const double* perf, int C, const uint8_t* step_type, int A, int per_agent, long long n, double* last,
                                    double* sum, long long* count, uint8_t* done_out);

__global__ void k_pow(const double* x, double y, double* out, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = sgw_glibc_pow(x[i], y);
}

// numpy PCG64 streams into the firemaker state (words 3..6), buffered-uint32 flag (word 0 bit 27) cleared
__global__ void k_set_rng(uint64_t* state, long long n_pad, long long n, int words, const uint64_t* pcg) {
  long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_pad) return;
  // padding lanes run the same code as real envs: give them a working stream too (an all-zero PCG state returns 0 for
  // ever, and Lemire's rejection loop never leaves on a constant 0)
  const uint64_t pad[4] = {0x9E3779B97F4A7C15ull, (uint64_t)e, 0ull, 1ull};
  for (int k = 0; k < 4; ++k) state[state_index(3 + k, e, words)] = e < n ? pcg[e * 4 + k] : pad[k];
  state[state_index(0, e, words)] &= ~(1ull << 27);
  state[state_index(2, e, words)] &= ~0xffffffffull;
}

// canonical view of the state for sgw_get_state / sgw_set_state: uint64 [words][n_pad] <-> the engine's pair layout
__global__ void k_state_export(const uint64_t* state, long long n_pad, int words, uint64_t* out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)words * n_pad) return;
  const int w = (int)(i / n_pad);
  const long long e = i - (long long)w * n_pad;
  out[i] = state[state_index(w, e, words)];
}
__global__ void k_state_import(uint64_t* state, long long n_pad, int words, const uint64_t* in) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)words * n_pad) return;
  const int w = (int)(i / n_pad);
  const long long e = i - (long long)w * n_pad;
  state[state_index(w, e, words)] = in[i];
}
// every env "never reset": step_type ST_NONE, termination reason absent, everything else zero
__global__ void k_state_init(uint64_t* state, long long n_pad, int words) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= state_alloc_words(words, n_pad)) return;
  const bool word0 = ((i & 127) & 1) == 0 && ((i >> 7) % state_pairs(words)) == 0;
  state[i] = word0 ? (((uint64_t)ST_NONE << 32) | ((uint64_t)15 << 36)) : 0ull;
}

// ---- derived statistics (safety_game_mo.py:1027-1084) in numpy's summation order -----------------------------
// numpy pairwise_sum for a contiguous double array (umath loops): n < 8 sequential from 0.0; n <= 128 eight accumulators
// over blocks of 8, tree-combined, remainder added sequentially; larger n split once (n2 = n/2 rounded down to a
// multiple of 8: both halves <= 128 for n <= 256 = K * K outer differences, K <= 16).  Everything is unrolled at compile time
// for the agent's number of dimensions K (a switch over 1..16): the element index is a constant, so the vectors stay in
// registers and a sum over |d_i - d_j| needs no array of the K * K differences (round 2 kept one per thread: 2.6 KB of scratch
// per lane; a version that streamed the vectors from LDS at run-time indices was bound by serial LDS round trips, 23 us).
template <int T0, int N, class Get> __device__ __forceinline__ double np_block_c(Get& get) {
  if constexpr (N < 8) {
    double r = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) r += get(T0 + i);
    return r;
  } else {
    double r[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) r[u] = get(T0 + u);
#pragma unroll
    for (int i = 8; i < N - (N % 8); i += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) r[u] += get(T0 + i + u);
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
#pragma unroll
    for (int i = N - (N % 8); i < N; ++i) res += get(T0 + i);
    return res;
  }
}
template <int N, class Get> __device__ __forceinline__ double np_sum_c(Get get) {
  if constexpr (N <= 128) return np_block_c<0, N>(get);
  else {
    constexpr int n2 = (N / 2) - ((N / 2) % 8);
    const double lo = np_block_c<0, n2>(get);
    return lo + np_block_c<n2, N - n2>(get);
  }
}
template <int K> __device__ __forceinline__ double np_gini100_c(const double (&v)[K]) {      // gini_coefficient(...) * 100 (1645-1681)
  double mn = v[0];
#pragma unroll
  for (int i = 1; i < K; ++i) mn = v[i] < mn ? v[i] : mn;                                   // python min()
  double d[K];
#pragma unroll
  for (int i = 0; i < K; ++i) d[i] = v[i] - mn;
  const double mad = np_sum_c<K * K>([&](int t) { return fabs(d[t / K] - d[t % K]); }) / (double)(K * K);
  const double rel = mad / (np_sum_c<K>([&](int i) { return d[i]; }) / (double)K + 2.220446049250313e-16);
  return 0.5 * rel * 100.0;
}
template <int K> __device__ __forceinline__ double np_var_c(const double (&v)[K]) {          // np.var(list, ddof=0)
  const double mean = np_sum_c<K>([&](int i) { return v[i]; }) / (double)K;
  double x[K];
#pragma unroll
  for (int i = 0; i < K; ++i) { const double y = v[i] - mean; x[i] = y * y; }
  return np_sum_c<K>([&](int i) { return x[i]; }) / (double)K;
}
struct AgentK { int k[SGW_MAX_AGENTS]; };
// one agent's statistics for this lane's env: vectors from the transposed LDS rows into registers, results into the lane's
// output row in LDS
template <int K> __device__ __noinline__ void derived_agent(const double* Rv, const double* Cv, double denom, double* o, int Kout, int lane) {
  double r[K], c[K], avg[K];
#pragma unroll
  for (int i = 0; i < K; ++i) { r[i] = Rv[i * WAVE + lane]; c[i] = Cv[i * WAVE + lane]; }
#pragma unroll
  for (int i = 0; i < K; ++i) avg[i] = c[i] / denom;
  o[0] = np_gini100_c<K>(r);
  o[1] = np_gini100_c<K>(c);
  o[2] = np_var_c<K>(r);
  o[3] = np_var_c<K>(c);
  o[4] = np_var_c<K>(avg);
#pragma unroll
  for (int j = 0; j < K; ++j) o[5 + j] = avg[j];
  for (int j = K; j < Kout; ++j) o[5 + j] = 0.0;
}
// One wave per 64 envs.  The wave's 64 x A rows of `reward` and `cumulative` are contiguous in global memory: they come in as
// 16-byte loads and are transposed through LDS as [agent][dimension][lane] (a lane then reads its own vector conflict-free);
// the wave's 64 x A x (5 + K) results are staged in LDS and leave as 16-byte stores.
// LDS (dynamic): R [A][K][64] | C [A][K][64] | O [64 * A * (5 + K)] doubles.
__device__ inline void derived_stats_wave(uint8_t* ds_lds, int lane, long long env0, int rows, const double* reward, const double* cumulative, const int* frame,
                                          long long n, int A, int K, const AgentK& ak, int recip_K, double* stats) {
  const long long env = env0 + lane;
  const int AK = A * K, S = 5 + K;
  double* R = reinterpret_cast<double*>(ds_lds);
  double* Cm = R + AK * WAVE;
  double* O = Cm + AK * WAVE;
  // ---- rows in: element e of the wave's block (row-major [64][A][K]) -> [agent * K + dim][lane]
  const long long rows_left = n - env0;
  const int nrow = rows_left < rows ? (int)rows_left : rows;        // a ragged last wave reads only its own rows (rows <= 64 per wave)
  const int nel = nrow * AK;
  const double* gr = reward + env0 * AK;
  const double* gc = cumulative + env0 * AK;
  for (int e0 = 2 * lane; e0 < WAVE * AK; e0 += 2 * WAVE) {
    double r0 = 0.0, r1 = 0.0, c0 = 0.0, c1 = 0.0;
    if (e0 + 1 < nel) {
      const double2 rv = *reinterpret_cast<const double2*>(gr + e0), cv = *reinterpret_cast<const double2*>(gc + e0);
      r0 = rv.x; r1 = rv.y; c0 = cv.x; c1 = cv.y;
    } else if (e0 < nel) { r0 = gr[e0]; c0 = gc[e0]; }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int e = e0 + h;
      const int row = (e * recip_K) >> 18;                          // e / (A * K): exact for e < 64 * 64 (host-computed ceil(2^18 / (A K)))
      const int col = e - row * AK;
      R[col * WAVE + row] = h ? r1 : r0;
      Cm[col * WAVE + row] = h ? c1 : c0;
    }
  }
  const double denom = (double)(((env < n && lane < nrow) ? frame[env] : 0) + 1);
  lds_wave_sync();
  for (int ag = 0; ag < A; ++ag) {
    const double* Rv = R + ag * K * WAVE;
    const double* Cv = Cm + ag * K * WAVE;
    double* o = O + (lane * A + ag) * S;
    switch (ak.k[ag]) {                                             // uniform
#define SGW_DS_CASE(KK) case KK: derived_agent<KK>(Rv, Cv, denom, o, K, lane); break;
      SGW_DS_CASE(1) SGW_DS_CASE(2) SGW_DS_CASE(3) SGW_DS_CASE(4) SGW_DS_CASE(5) SGW_DS_CASE(6) SGW_DS_CASE(7) SGW_DS_CASE(8)
      SGW_DS_CASE(9) SGW_DS_CASE(10) SGW_DS_CASE(11) SGW_DS_CASE(12) SGW_DS_CASE(13) SGW_DS_CASE(14) SGW_DS_CASE(15) SGW_DS_CASE(16)
#undef SGW_DS_CASE
      default: {                                                    // an agent without reward dimensions (an absent slot)
        const double nan = __longlong_as_double(0x7ff8000000000000LL);
        o[0] = 0.0; o[1] = 0.0; o[2] = nan; o[3] = nan; o[4] = nan;
        for (int j = 0; j < K; ++j) o[5 + j] = 0.0;
      }
    }
  }
  lds_wave_sync();
  // ---- results out: the wave's nrow * A * S doubles are contiguous
  const int nout = nrow * A * S;
  double* g = stats + env0 * A * S;
  for (int e0 = 2 * lane; e0 < nout; e0 += 2 * WAVE) {
    if (e0 + 1 < nout) *reinterpret_cast<double2*>(g + e0) = *reinterpret_cast<const double2*>(O + e0);
    else g[e0] = O[e0];
  }
}
__global__ __launch_bounds__(WAVE) void k_derived_stats(const double* reward, const double* cumulative, const int* frame, long long n, int A, int K,
                                                        AgentK ak, int recip_K, double* stats) {
  extern __shared__ __attribute__((aligned(16))) uint8_t ds_lds[];
  derived_stats_wave(ds_lds, (int)threadIdx.x, (long long)blockIdx.x * WAVE, WAVE, reward, cumulative, frame, n, A, K, ak, recip_K, stats);
}

// ---- per-character planes of a rendered board: RGB, occluded layers, unoccluded layers ---------------------------------------
// Every one of these outputs is [N][P planes][H*W] bytes with byte (e, p, c) a function of plane p and of what is at cell c of
// env e.  A workgroup takes 64 envs: their board rows (contiguous) come in as 16-byte loads into LDS, phase 1 reduces a cell to
// a CODE (the ascii code, or the bit mask of the layers that are on there), and phase 2 writes the block's 64 * P * HW
// contiguous output bytes 16 at a time -- a lane walks (env, plane, cell) forward by one from the chunk's first byte, whose
// coordinates come from two reciprocal multiplies, instead of dividing per byte.  (Round 2: one thread per cell, an integer
// division each, P one-byte stores at a stride of HW.)
constexpr int PLANES_THREADS = 256, PLANES_ENVS = 64;      // (PLANES_ENVS: the most envs a workgroup takes; 16 for boards whose staging would crowd the CU's LDS)
struct PlaneGeom { int HW, P, recip_PHW, recip_HW, recip_Q; };       // recip_x = ceil(2^32 / x) as uint32: (f * recip) >> 32 == f / x for f < 2^20; Q = ceil(HW / 4)
__device__ inline uint32_t div_recip(uint32_t f, uint32_t recip) { return recip ? (uint32_t)(((uint64_t)f * recip) >> 32) : f; }   // recip 0: x == 1
// phase 2.  value(p, code) -> output byte
template <class Value>
__device__ inline void planes_expand(uint8_t* out_block, const PlaneGeom& g, int n_env, Value value) {
  const uint32_t per_env = (uint32_t)(g.P * g.HW), total = (uint32_t)n_env * per_env;
  for (uint32_t f0 = 16u * threadIdx.x; f0 < total; f0 += 16u * PLANES_THREADS) {
    uint32_t e = div_recip(f0, (uint32_t)g.recip_PHW), rem = f0 - e * per_env;
    uint32_t p = div_recip(rem, (uint32_t)g.recip_HW), c = rem - p * (uint32_t)g.HW;
    uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      const uint32_t v = f0 + b < total ? (uint32_t)value((int)p, (int)e, (int)c) & 0xffu : 0u;
      w[b >> 2] |= v << (8 * (b & 3));
      if (++c == (uint32_t)g.HW) { c = 0; if (++p == (uint32_t)g.P) { p = 0; ++e; } }
    }
    if (f0 + 16u <= total) *reinterpret_cast<uint4*>(out_block + f0) = make_uint4(w[0], w[1], w[2], w[3]);
    else for (uint32_t b = 0; f0 + b < total; ++b) out_block[f0 + b] = (uint8_t)(w[b >> 2] >> (8 * (b & 3)));
  }
}
// phase 2, the fast form: a lane takes FOUR consecutive cells of an env, fetches their codes once and writes one dword per plane
// -- consecutive lanes write consecutive dwords of a row.  When H*W is not a multiple of 4 the rows do not start on dwords: the
// stores are then unaligned dword stores (global memory takes them; the bytes of a wave's instruction are contiguous all the
// same) and the last, partial group of a row leaves as single bytes.  dword4(e, c, nvalid, put): put(p, the 4 output bytes)
typedef uint32_t __attribute__((aligned(1))) sgw_u32_unaligned;
template <class Dword4>
__device__ inline void planes_expand4(uint8_t* out_block, const PlaneGeom& g, int n_env, Dword4 dword4) {
  const int HW = g.HW, q = (HW + 3) >> 2, items = n_env * q;
  for (int i = threadIdx.x; i < items; i += PLANES_THREADS) {
    const int e = (int)div_recip((uint32_t)i, (uint32_t)g.recip_Q), c = 4 * (i - e * q);
    const int nvalid = HW - c < 4 ? HW - c : 4;
    uint8_t* row = out_block + (size_t)e * g.P * HW + c;
    dword4(e, c, nvalid, [&](int p, uint32_t v) {
      uint8_t* d = row + (size_t)p * HW;
      if (nvalid == 4) {
#if defined(__HIP_DEVICE_COMPILE__)
        // ONE dword store at a byte address (the compiler splits a store it cannot prove aligned into four byte stores; the memory
        // pipeline takes unaligned dwords: the tests compare every byte with the fixtures)
        asm volatile("global_store_dword %0, %1, off" : : "v"(d), "v"(v) : "memory");
#else
        *reinterpret_cast<sgw_u32_unaligned*>(d) = v;
#endif
      } else for (int b = 0; b < nvalid; ++b) d[b] = (uint8_t)(v >> (8 * b));
    });
  }
}
// four consecutive LDS bytes at any alignment (cells past the row's end read the row's last cell: their output is dropped)
__device__ inline uint32_t lds_bytes4(const uint8_t* base, int at, int nvalid) {
  const int i1 = nvalid > 1 ? 1 : 0, i2 = nvalid > 2 ? 2 : i1, i3 = nvalid > 3 ? 3 : i2;
  return (uint32_t)base[at] | ((uint32_t)base[at + i1] << 8) | ((uint32_t)base[at + i2] << 16) | ((uint32_t)base[at + i3] << 24);
}
// the block's board rows -> LDS (16-byte loads; 64 * HW is a multiple of 16 and the block starts on one)
__device__ inline void planes_load_boards(const uint8_t* board, long long env0, int n_env, int HW, uint8_t* lds_board) {
  const uint4* src = reinterpret_cast<const uint4*>(board + env0 * HW);
  const int bytes = n_env * HW, n16 = (reinterpret_cast<uintptr_t>(board) & 15) ? 0 : bytes >> 4;      // a caller's unaligned slice: bytes
  for (int j = threadIdx.x; j < n16; j += PLANES_THREADS) reinterpret_cast<uint4*>(lds_board)[j] = src[j];
  for (int j = (n16 << 4) + threadIdx.x; j < bytes; j += PLANES_THREADS) lds_board[j] = board[env0 * HW + j];     // ragged last block
}

// the RGB planes of a block's boards (in LDS) through the colour table tab[plane][char] (in LDS)
__device__ inline void observe_rgb_expand(const uint8_t* bl, const uint8_t* tab, uint8_t* rgb_block, const PlaneGeom& g_rgb, int n_env) {
  const int HW = g_rgb.HW;
  planes_expand4(rgb_block, g_rgb, n_env, [&](int e, int c, int nvalid, auto put) {
    const uint32_t b4 = (HW & 3) == 0 ? *reinterpret_cast<const uint32_t*>(bl + e * HW + c) : lds_bytes4(bl, e * HW + c, nvalid);
    const uint32_t c0 = b4 & 0x7f, c1 = (b4 >> 8) & 0x7f, c2 = (b4 >> 16) & 0x7f, c3 = (b4 >> 24) & 0x7f;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const uint8_t* t = tab + p * 128;
      put(p, (uint32_t)t[c0] | ((uint32_t)t[c1] << 8) | ((uint32_t)t[c2] << 16) | ((uint32_t)t[c3] << 24));
    }
  });
}

// observation distiller extras from an ascii board (observation_distiller.py:30-91, rendering.py:69-185): RGB planes via the
// colour LUT and OCCLUDED per-character layers (board == char).  LDS: boards [64 * HW] | table [P][128]
__global__ __launch_bounds__(PLANES_THREADS) void k_observe(const uint8_t* board, long long n, int epb, PlaneGeom g_rgb, const uint8_t* rgb_lut, uint8_t* rgb,
                                                            PlaneGeom g_lay, const uint8_t* layer_chars, uint8_t* layers) {
  extern __shared__ __attribute__((aligned(16))) uint8_t pl_lds[];
  const int HW = g_rgb.HW;
  const long long env0 = (long long)blockIdx.x * epb;              // epb envs per workgroup: 64, or 16 for large boards (LDS per workgroup)
  const int n_env = n - env0 < epb ? (int)(n - env0) : epb;
  uint8_t* bl = pl_lds;
  uint8_t* tab = pl_lds + ((epb * HW + 15) & ~15);
  planes_load_boards(board, env0, n_env, HW, bl);
  if (rgb) {
    for (int i = threadIdx.x; i < 3 * 128; i += PLANES_THREADS) tab[i] = rgb_lut[(i & 127) * 3 + (i >> 7)];        // [plane][char]
    __syncthreads();
    observe_rgb_expand(bl, tab, rgb + env0 * 3 * HW, g_rgb, n_env);
  }
  if (layers) {
    __syncthreads();
    for (int i = threadIdx.x; i < g_lay.P; i += PLANES_THREADS) tab[i] = layer_chars[i];
    __syncthreads();
    planes_expand4(layers + env0 * g_lay.P * HW, g_lay, n_env, [&](int e, int c, int nvalid, auto put) {
      const uint32_t b4 = ((HW & 3) == 0 ? *reinterpret_cast<const uint32_t*>(bl + e * HW + c) : lds_bytes4(bl, e * HW + c, nvalid)) & 0x7f7f7f7fu;
      for (int p = 0; p < g_lay.P; ++p) put(p, bytes_equal_mask(b4, 0x01010101u * tab[p]) & 0x01010101u);
    });
  }
}

// unoccluded layers + gap correction (rendering.py:188-302, observation_distiller_ex.py:164-178).  Phase 1 reduces cell (e, c)
// to the bit mask of the layers that are on there: a dynamic layer (stat 2: sprite / moving drape) where the board shows its
// character, a static curtain (stat 1) always, the hidden drape under an agent that covers it (firemaker's fire), and the
// what_lies_beneath layer only where every other layer is blank.  LDS: boards [64 * HW] | masks u32 [64 * HW] | per-cell tables
// dyn u32 [HW], on u32 [HW] | per-char table u32 [128]
__device__ inline void observe_layers_block(uint8_t* pl_lds, long long env0, int epb, bool boards_loaded, const uint8_t* board, long long n, PlaneGeom g, int W,
                                            const uint8_t* chars, const uint8_t* stat, int gap, const uint8_t* pos, const uint8_t* flags, int A,
                                            int hidden, uint8_t* layers) {
  const int HW = g.HW, L = g.P;
  const int n_env = n - env0 < epb ? (int)(n - env0) : epb;
  uint8_t* bl = pl_lds;
  uint32_t* mask = reinterpret_cast<uint32_t*>(pl_lds + ((epb * HW + 15) & ~15));
  uint32_t* dyn = mask + epb * HW;
  uint32_t* on = dyn + HW;
  uint32_t* chm = on + HW;
  if (!boards_loaded) planes_load_boards(board, env0, n_env, HW, bl);
  for (int c = threadIdx.x; c < HW; c += PLANES_THREADS) {
    uint32_t d = 0u, o = 0u;
    for (int k = 0; k < L; ++k) { const uint8_t st = stat[k * HW + c]; d |= (st == 2 ? 1u : 0u) << k; o |= ((st != 0 && st != 2) ? 1u : 0u) << k; }
    dyn[c] = d; on[c] = o;
  }
  for (int ch = threadIdx.x; ch < 128; ch += PLANES_THREADS) {
    uint32_t m = 0u;
    for (int k = 0; k < L; ++k) m |= (chars[k] == ch ? 1u : 0u) << k;
    chm[ch] = m;
  }
  __syncthreads();
  const uint32_t gapbit = gap >= 0 ? 1u << gap : 0u;
  for (int i = threadIdx.x; i < n_env * HW; i += PLANES_THREADS) {
    const int e = (int)div_recip((uint32_t)i, (uint32_t)g.recip_HW), c = i - e * HW;
    mask[i] = ((chm[bl[i] & 0x7f] & dyn[c]) | on[c]) & ~gapbit;      // the gap layer is decided below
  }
  __syncthreads();
  if (pos && hidden >= 0) {                                           // the hidden drape under an agent's sprite
    for (int i = threadIdx.x; i < n_env * A; i += PLANES_THREADS) {
      const long long ea = env0 * A + i;
      const int e = i / A;
      if (flags[ea] & 1) atomicOr(&mask[e * HW + (int)pos[ea * 2] * W + (int)pos[ea * 2 + 1]], 1u << hidden);
    }
    __syncthreads();
  }
  if (gap >= 0) {
    for (int i = threadIdx.x; i < n_env * HW; i += PLANES_THREADS) {
      const int e = (int)div_recip((uint32_t)i, (uint32_t)g.recip_HW), c = i - e * HW;
      const uint32_t m = mask[i];
      const bool gap_static = ((on[c] | dyn[c]) & gapbit) != 0u;              // the layer's curtain entry is not 0
      if (gap_static && (m & ~gapbit) == 0u) mask[i] = m | gapbit;
    }
    __syncthreads();
  }
  planes_expand4(layers + env0 * L * HW, g, n_env, [&](int e, int c, int nvalid, auto put) {
    uint4 m;                                                         // the four cells' layer masks
    if ((HW & 3) == 0) m = *reinterpret_cast<const uint4*>(mask + e * HW + c);      // one 16-byte LDS read
    else {
      const uint32_t* mp = mask + e * HW + c;
      m = make_uint4(mp[0], mp[nvalid > 1 ? 1 : 0], mp[nvalid > 2 ? 2 : 0], mp[nvalid > 3 ? 3 : 0]);
    }
    for (int p = 0; p < L; ++p)
      put(p, ((m.x >> p) & 1u) | (((m.y >> p) & 1u) << 8) | (((m.z >> p) & 1u) << 16) | (((m.w >> p) & 1u) << 24));
  });
}
__global__ __launch_bounds__(PLANES_THREADS) void k_observe_layers(const uint8_t* board, long long n, int epb, PlaneGeom g, int W, const uint8_t* chars,
                                                                   const uint8_t* stat, int gap, const uint8_t* pos, const uint8_t* flags, int A,
                                                                   int hidden, uint8_t* layers) {
  extern __shared__ __attribute__((aligned(16))) uint8_t pl_lds[];
  observe_layers_block(pl_lds, (long long)blockIdx.x * epb, epb, false, board, n, g, W, chars, stat, gap, pos, flags, A, hidden, layers);
}

// ---- sgw_step_full's board-derived outputs in ONE launch: a workgroup takes `epb` envs of the step that just ran and produces, from
// their boards (loaded into LDS once), the RGB planes and the unoccluded layers, then the performance bookkeeping and the `done`
// flags of those envs: one kernel boundary and one graph node instead of three.
struct ExtrasArgs {
  const uint8_t* board; long long n; int epb, HW, W, A, K;
  PlaneGeom g_rgb; const uint8_t* rgb_lut; uint8_t* rgb;
  PlaneGeom g_lay; const uint8_t* chars; const uint8_t* stat; int gap, hidden; const uint8_t* pos; const uint8_t* flags; uint8_t* layers;
  const double* reward; const double* cumulative; const int* frame; AgentK ak; int recip_K; double* stats;
  const double* perf; int perf_cols, per_agent; const uint8_t* step_type; double* last; double* sum; long long* count; uint8_t* done;
  int lds_planes;                                   // bytes of the planes region (the statistics' rows follow it)
};
__device__ inline void track_performance_item(long long i, const double* perf, int C, const uint8_t* step_type, int A, int per_agent, double* last,
                                              double* sum, long long* count, uint8_t* done_out) {
  const long long e = i / C;
  bool done = step_type[e * A] == ST_LAST;
  if (per_agent) { done = true; for (int ag = 0; ag < A; ++ag) done = done && step_type[e * A + ag] >= ST_LAST; }
  if (done_out && i == e * C) done_out[e] = done ? 1 : 0;
  if (!done) return;
  const double v = perf[i];
  if (last) last[i] = v;
  if (sum) sum[i] += v;
  if (count && i == e * C) count[e] += 1;
}
__global__ void k_track_performance(const double* perf, int C, const uint8_t* step_type, int A, int per_agent, long long n, double* last,
                                    double* sum, long long* count, uint8_t* done_out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n * C) track_performance_item(i, perf, C, step_type, A, per_agent, last, sum, count, done_out);
}
__global__ __launch_bounds__(PLANES_THREADS) void k_step_extras(const ExtrasArgs x) {
  extern __shared__ __attribute__((aligned(16))) uint8_t pl_lds[];
  const long long env0 = (long long)blockIdx.x * x.epb;
  const int n_env = x.n - env0 < x.epb ? (int)(x.n - env0) : x.epb;
  const bool planes = x.rgb != nullptr || x.layers != nullptr;
  if (planes) planes_load_boards(x.board, env0, n_env, x.HW, pl_lds);
  if (x.rgb) {
    uint8_t* tab = pl_lds + x.lds_planes - 3 * 128;               // the colour table sits at the end of the planes region
    for (int i = threadIdx.x; i < 3 * 128; i += PLANES_THREADS) tab[i] = x.rgb_lut[(i & 127) * 3 + (i >> 7)];
    __syncthreads();
    observe_rgb_expand(pl_lds, tab, x.rgb + env0 * 3 * x.HW, x.g_rgb, n_env);
  }
  if (x.layers) {
    __syncthreads();
    observe_layers_block(pl_lds, env0, x.epb, true, x.board, x.n, x.g_lay, x.W, x.chars, x.stat, x.gap, x.pos, x.flags, x.A, x.hidden, x.layers);
  }
  // (the derived statistics stay a launch of their own: their unrolled K = 16 body needs ~290 registers, and compiled into this
  // kernel that budget applied to every wave -- one workgroup per CU, 83 instead of 54 us per full step)
  if (x.perf) {
    for (long long i = env0 * x.perf_cols + (int)threadIdx.x; i < (env0 + n_env) * x.perf_cols; i += PLANES_THREADS)
      track_performance_item(i, x.perf, x.perf_cols, x.step_type, x.A, x.per_agent, x.last, x.sum, x.count, x.done);
  }
}

// agent-centric windows of the rendered board (safety_game_moma.py:1996-2101), one thread per output byte
struct ViewSpec { int A, H, W, total; int off[SGW_MAX_AGENTS], up[SGW_MAX_AGENTS], left[SGW_MAX_AGENTS], vh[SGW_MAX_AGENTS], vw[SGW_MAX_AGENTS]; };
// (view_unrotate: above, with the in-kernel windows)
// One WAVE produces one agent's window of one env.  The lanes cover whole window rows (64 / vw rows per pass, one byte per lane) and
// read contiguous board cells (a board row, or a column when the window is rotated); the window is assembled in LDS at the same
// 16-byte phase as its place in the output and leaves as 16-byte stores (head and tail bytes singly): window bytes are not
// aligned to anything (1 139 bytes per firemaker env), and byte-masked partial-line writes were what the earlier versions
// were waiting for.  History (16 384 firemaker envs x 1 139 window bytes): one thread per output byte with its own divisions,
// 46 us; one thread per window row, 49 us; one wave per window with byte stores, 42-44 us.
__device__ inline void view_wave(const uint8_t* src, uint8_t* dst, uint8_t* lds, int H, int W, int vh, int vw, int pr, int pc, int dir,
                                 bool rotate, uint8_t pad, int lane) {
  const int len = vh * vw;
  const int phase = (int)(reinterpret_cast<uintptr_t>(dst) & 15);
  uint8_t* stage = lds + phase;                                   // stage[j] and dst[j] share their 16-byte alignment
  if (len > H * W && vw <= 255) {
    // a window LARGER than the board (firemaker's supervisor: 33 x 33 around a 17 x 17 board) is mostly padding: fill the stage
    // with the pad byte (16-byte LDS stores), then walk the BOARD cells once and drop each where it lands in the window
    const uint32_t p4 = 0x01010101u * pad;
    const uint4 fill = make_uint4(p4, p4, p4, p4);
    for (int k = lane; k < (phase + len + 15) >> 4; k += WAVE) reinterpret_cast<uint4*>(lds)[k] = fill;
    lds_wave_sync();
    const int cells = H * W;
    for (int k0 = 0; k0 < cells; k0 += 4 * WAVE) {                 // four loads in flight
      uint8_t val[4]; int at[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = k0 + q * WAVE + lane;
        const bool in_board = k < cells;
        val[q] = src[in_board ? k : 0];
        const int r = k / W, c = k - r * W;
        const int cr = r - pr, cc = c - pc;                       // crop coordinates
        int vr = cr, vc = cc;                                     // inverse of view_unrotate
        if (rotate) {
          if (dir == 3) { vr = vw - 1 - cr; vc = vw - 1 - cc; }
          else if (dir == 0) { vr = cc; vc = vw - 1 - cr; }
          else if (dir == 1) { vr = vw - 1 - cc; vc = cr; }
        }
        at[q] = (in_board && cr >= 0 && cr < vh && cc >= 0 && cc < vw) ? vr * vw + vc : -1;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) if (at[q] >= 0) stage[at[q]] = val[q];
    }
  } else if (vw > WAVE) {                                         // wider than a wave: plain column chunks (no reference env is)
    for (int vr = 0; vr < vh; ++vr)
      for (int vc = lane; vc < vw; vc += WAVE) {
        int r = vr, c = vc;
        if (rotate) view_unrotate(dir, vw, r, c);
        r += pr; c += pc;
        stage[vr * vw + vc] = (r < 0 || r >= H || c < 0 || c >= W) ? pad : src[r * W + c];
      }
  } else {
    const int rpg = WAVE / vw;                                    // window rows per pass
    const int lr = lane / vw, lc = lane - lr * vw;
    const bool lane_on = lr < rpg;
    for (int vr0 = 0; vr0 < vh; vr0 += 4 * rpg) {                 // four passes in flight: loads first, LDS writes after
      uint8_t val[4]; bool on[4]; int at[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int vr = vr0 + k * rpg + lr;
        on[k] = lane_on && vr < vh;
        int r = on[k] ? vr : 0, c = lc;
        if (rotate) view_unrotate(dir, vw, r, c);
        r += pr; c += pc;
        const bool inside = !(r < 0 || r >= H || c < 0 || c >= W);
        const uint8_t got = src[inside ? r * W + c : 0];           // unconditional load (index clamped), selected afterwards
        val[k] = inside ? got : pad;
        at[k] = vr * vw + lc;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) if (on[k]) stage[at[k]] = val[k];
    }
  }
  lds_wave_sync();
  const int head = (16 - phase) & 15;                             // bytes up to the first 16-byte boundary of the output
  const int nh = head < len ? head : len;
  if (lane < nh) dst[lane] = stage[lane];
  const int body = (len - nh) >> 4;                               // whole 16-byte chunks
  for (int k = lane; k < body; k += WAVE)
    *reinterpret_cast<uint4*>(dst + nh + 16 * k) = *reinterpret_cast<const uint4*>(stage + nh + 16 * k);
  const int done = nh + 16 * body;
  if (lane < len - done) dst[done + lane] = stage[done + lane];
  lds_wave_sync();                                                // the next window of this wave reuses the stage
}
__global__ void k_agent_views(const uint8_t* board, const uint8_t* pos, const uint8_t* flags, long long n, ViewSpec v,
                              uint8_t outside, uint8_t* views, int lds_per_wave) {
  extern __shared__ __attribute__((aligned(16))) uint8_t view_lds[];
  const int lane = threadIdx.x & (WAVE - 1);
  uint8_t* lds = view_lds + (threadIdx.x >> 6) * lds_per_wave;
  const long long stride = ((long long)gridDim.x * blockDim.x) >> 6, windows = n * v.A;
  for (long long wv = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; wv < windows; wv += stride) {
    const long long e = wv / v.A;
    const int ag = (int)(wv - e * v.A);
    if (v.vh[ag] * v.vw[ag] == 0) continue;                      // an agent without a window (absent firemaker agents)
    const int pr = (int)pos[wv * 2] - v.up[ag], pc = (int)pos[wv * 2 + 1] - v.left[ag];
    const int dir = flags ? (flags[wv] >> 3) & 3 : 2;
    view_wave(board + e * (long long)(v.H * v.W), views + e * v.total + v.off[ag], lds, v.H, v.W, v.vh[ag], v.vw[ag], pr, pc, dir,
              flags != nullptr, outside, lane);
  }
}

// Small windows (every window <= 64 cells, e.g. the 5 x 5 windows of island_navigation_ex_ma): a wave per window would leave most
// lanes idle (131 072 waves for 65 536 two-agent envs); one thread per output byte keeps the stores contiguous across lanes
// and costs ~60 instructions per 64 bytes.  `per_env` = bytes per env (window bytes x layers for the layer variant).
__device__ inline uint8_t small_view_byte(const uint8_t* src_env, const uint8_t* pos, const uint8_t* flags, const ViewSpec& v,
                                          uint8_t pad, long long e, int ag, int cellidx) {
  const int vw = v.vw[ag];
  int vr = cellidx / vw, vc = cellidx - vr * vw;
  const long long ea = e * v.A + ag;
  if (flags) view_unrotate((flags[ea] >> 3) & 3, vw, vr, vc);
  const int r = (int)pos[ea * 2] - v.up[ag] + vr, c = (int)pos[ea * 2 + 1] - v.left[ag] + vc;
  return (r < 0 || r >= v.H || c < 0 || c >= v.W) ? pad : src_env[r * v.W + c];
}
__global__ void k_agent_views_small(const uint8_t* board, const uint8_t* pos, const uint8_t* flags, long long n, ViewSpec v,
                                    uint8_t outside, uint8_t* views) {
  const long long total = n * v.total;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long e = i / v.total;
    int b = (int)(i - e * v.total), ag = 0;
#pragma unroll
    for (int k = 1; k < SGW_MAX_AGENTS; ++k) if (k < v.A && b >= v.off[k]) ag = k;
    views[i] = small_view_byte(board + e * (long long)(v.H * v.W), pos, flags, v, outside, e, ag, b - v.off[ag]);
  }
}
__global__ void k_agent_layer_views_small(const uint8_t* layers, const uint8_t* pos, const uint8_t* flags, long long n, ViewSpec v,
                                          const uint8_t* chars, int L, uint8_t outside, uint8_t* out) {
  const long long per_env = (long long)v.total * L, total = n * per_env;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long e = i / per_env;
    int b = (int)(i - e * per_env), ag = 0;
#pragma unroll
    for (int k = 1; k < SGW_MAX_AGENTS; ++k) if (k < v.A && b >= v.off[k] * L) ag = k;
    b -= v.off[ag] * L;
    const int cells = v.vh[ag] * v.vw[ag];
    const int li = b / cells;
    out[i] = small_view_byte(layers + (e * L + li) * (long long)(v.H * v.W), pos, flags, v, (uint8_t)(chars[li] == outside), e, ag,
                             b - li * cells);
  }
}

// per-layer agent windows, windows larger than 64 cells (firemaker's 33 x 33, savanna's 21 x 21): a WORKGROUP takes G envs at a
// time.  Their L layer planes come into LDS, each of the A x L windows of an env is assembled in an LDS image of the env's whole
// output row by one wave (a window larger than the plane: pad fill, then the plane's cells dropped where they land; else a
// gather), and the rows leave as dword stores at their byte address (rows of L * view_bytes bytes are not aligned to anything).
// Why G envs at once: an env is a chain of dependent waits -- planes and positions from global memory, the LDS passes, and the
// next env's loads queue behind this env's stores (vmcnt counts both) -- about 8 us per env whatever its size; with one env per
// wave (the L = 1 form: sgw_agent_views) a CU had 32 such chains in flight and 16 384 firemaker envs took 17.5 us, 65 536
// savanna envs 60 us.  G envs share every wait of the chain.  Round 2's kernel below -- one wave per window, every cell a byte
// load from global memory -- took 208 us for 16 384 firemaker envs' layer cubes (0.13 of HBM).
template <int G>
__global__ __launch_bounds__(256) void k_agent_layer_views_lds(const uint8_t* layers, const uint8_t* pos, const uint8_t* flags, long long n, ViewSpec v,
                                                               const uint8_t* chars, int L, uint8_t outside, uint8_t* out, int lay_bytes, int img_bytes,
                                                               int pad_is_char) {
  // pad_is_char: the planes are ascii boards (L = 1: sgw_agent_views) and cells outside the board read `outside` itself; otherwise
  // they are layer planes and read 1 on the outside character's own layer (agent_perspectives_with_layers)
  // LDS: [ per-agent table: off, up, left, vh, vw x 4 | per-env words of the group: row | col << 8 | flags << 16, x G x 4 |
  //        G x ( [L][H*W] planes | the env's output row [agent][layer][vh][vw] at its 16-byte phase ) ]
  // What a loop indexes at run time (the agent's window, the env's position) is READ from LDS: as members of the kernel
  // argument / as register arrays they became select chains over every agent and env -- 2 200 scalar and 420 spill
  // instructions, 400 branches in the first G-env version, and the kernel was bound by exactly that (75 us for savanna's
  // 65 536 envs, 26 of them with loads, assembly and stores all switched off).
  extern __shared__ __attribute__((aligned(16))) uint8_t view_lds[];
  int* tab = reinterpret_cast<int*>(view_lds);
  uint32_t* ppos = reinterpret_cast<uint32_t*>(view_lds + 128);
  uint8_t* envs = view_lds + 128 + 16 * G;
  const int env_lds = lay_bytes + img_bytes;
  const int HW = v.H * v.W, H = v.H, W = v.W, A = v.A, lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  const int row_bytes = v.total * L;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < SGW_MAX_AGENTS; ++k) { tab[k] = v.off[k] * L; tab[4 + k] = v.up[k]; tab[8 + k] = v.left[k]; tab[12 + k] = v.vh[k]; tab[16 + k] = v.vw[k]; }
  }
  // lane constants, once per kernel (no division inside the env loop): the plane cells this lane drops into a large window
  // (cell k = lane + 64 j: row, column)
  constexpr int NP = (SGW_MAX_CELLS + WAVE - 1) / WAVE;
  int prr[NP], pcc[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) { const int k = lane + WAVE * j; prr[j] = k / W; pcc[j] = k - prr[j] * W; }
  bool inb[NP];                                             // plane cell lane + 64 j exists
#pragma unroll
  for (int j = 0; j < NP; ++j) inb[j] = lane + WAVE * j < HW;
  uint8_t* const trash = view_lds + 124;                    // where the scatter's out-of-window cells land (no exec juggling per store)
  constexpr int PF = G > 1 ? 5 : 12;                        // plane bytes per thread held in registers between the load and the LDS write
  const int nbytes = L * HW;
  const bool in_regs = nbytes <= PF * (int)blockDim.x;
  for (long long e0 = (long long)blockIdx.x * G; e0 < n; e0 += (long long)gridDim.x * G) {
    // ---- every load of the G envs first: planes (into registers), positions and flags (a lane per (env, agent))
    uint32_t pl[G][PF];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const long long e = e0 + g < n ? e0 + g : n - 1;       // (a ragged last group re-reads the last env; nothing of it is stored)
      const uint8_t* src = layers + e * (long long)nbytes;
      if (in_regs) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
          const int idx = (int)threadIdx.x + j * (int)blockDim.x;
          pl[g][j] = idx < nbytes ? src[idx] : 0u;
        }
      }
    }
    if ((int)threadIdx.x < 4 * G) {
      const int g = threadIdx.x >> 2, ag = threadIdx.x & 3;
      const long long e = e0 + g < n ? e0 + g : n - 1;
      uint32_t w = 16u << 16;                                  // (no flags: observation direction UP)
      if (ag < A) {
        const long long ea = e * A + ag;
        w = (uint32_t)*reinterpret_cast<const uint16_t*>(pos + ea * 2) | ((flags ? (uint32_t)flags[ea] : 16u) << 16);
      }
      ppos[threadIdx.x] = w;
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      uint8_t* lay = envs + g * env_lds;
      if (pad_is_char) {                                       // ascii planes: one pad character for every window of the row -- the whole
        uint32_t* im4 = reinterpret_cast<uint32_t*>(lay + lay_bytes);      // image (its phase slack included) is filled here, once
        for (int k = threadIdx.x; k < (img_bytes >> 2); k += blockDim.x) im4[k] = 0x01010101u * outside;
      }
      if (in_regs) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
          const int idx = (int)threadIdx.x + j * (int)blockDim.x;
          if (idx < nbytes) lay[idx] = (uint8_t)pl[g][j];
        }
      } else {
        const long long e = e0 + g < n ? e0 + g : n - 1;
        const uint8_t* src = layers + e * (long long)nbytes;
        for (int i = threadIdx.x; i < nbytes; i += blockDim.x) lay[i] = src[i];
      }
    }
    __syncthreads();
    // ---- the windows of the G envs into their LDS images: agent by agent, so that what only depends on the agent's window
    // (the landing constants of the plane cells) is worked out once for the G envs
    for (int ag = 0; ag < A; ++ag) {                           // scalar loops: agent, env of the group, then this wave's layers
      const int vh = __builtin_amdgcn_readfirstlane(tab[12 + ag]), vw = __builtin_amdgcn_readfirstlane(tab[16 + ag]), cells = vh * vw;
      if (cells == 0) continue;
      const int up = __builtin_amdgcn_readfirstlane(tab[4 + ag]), left = __builtin_amdgcn_readfirstlane(tab[8 + ag]);
      const int off = __builtin_amdgcn_readfirstlane(tab[ag]), n1 = vw - 1;
      const bool big = cells > HW;
      const bool covers = up >= H - 1 && vh - 1 - up >= H - 1 && left >= W - 1 && vw - 1 - left >= W - 1;
      int P[NP], Q[NP];                                        // landing offset of plane cell (r, c): +-(r * n + c) or +-(c * n - r) plus an env scalar
#pragma unroll
      for (int j = 0; j < NP; ++j) { P[j] = prr[j] * vw + pcc[j]; Q[j] = pcc[j] * vw - prr[j]; }
#pragma nounroll
      for (int g = 0; g < G; ++g) {
        const uint8_t* lay = envs + g * env_lds;
        // the image sits at the 16-byte phase of its row in global memory, so that the row's aligned 16-byte chunks are aligned in
        // LDS too (img_bytes has 16 bytes of slack for it)
        const long long eg = e0 + g < n ? e0 + g : n - 1;
        uint8_t* img = envs + g * env_lds + lay_bytes + (int)(reinterpret_cast<uintptr_t>(out + eg * (long long)row_bytes) & 15);
        const uint32_t pw = (uint32_t)__builtin_amdgcn_readfirstlane((int)ppos[g * 4 + ag]);
        const int pr = (int)(pw & 0xffu) - up, pc = (int)((pw >> 8) & 0xffu) - left;
        const int dir = (int)((pw >> 19) & 3u);
        if (big) {
          const int base = dir == 2 ? -(pr * vw + pc) : (dir == 3 ? (n1 + pr) * vw + n1 + pc : (dir == 0 ? n1 + pr - pc * vw : (n1 + pc) * vw - pr));
          const bool useQ = dir < 2, neg = dir == 3 || dir == 1;
          int at[NP]; bool ok[NP];
#pragma unroll
          for (int j = 0; j < NP; ++j) {
            const int t = useQ ? Q[j] : P[j];
            at[j] = (neg ? -t : t) + base;
            ok[j] = inb[j];
            if (!covers) ok[j] = ok[j] && (unsigned)(prr[j] - pr) < (unsigned)vh && (unsigned)(pcc[j] - pc) < (unsigned)vw;
          }
          for (int l = wave; l < L; l += nwave) {
            const uint32_t pad = pad_is_char ? 0x01010101u * outside : (chars[l] == outside ? 0x01010101u : 0u);
            const uint8_t* plane = lay + l * HW;
            uint8_t* dst = img + off + l * cells;
            if (!pad_is_char) {
              // pad fill: bytes up to the first dword boundary, dwords, tail bytes (dst is byte-aligned only)
              const int head = (int)((4 - (reinterpret_cast<uintptr_t>(dst) & 3)) & 3), nh = head < cells ? head : cells;
              if (lane < nh) dst[lane] = (uint8_t)pad;
              uint32_t* d4 = reinterpret_cast<uint32_t*>(dst + nh);
              const int nd = (cells - nh) >> 2;
              for (int k = lane; k < nd; k += WAVE) d4[k] = pad;
              const int done = nh + 4 * nd;
              if (lane < cells - done) dst[done + lane] = (uint8_t)pad;
              lds_wave_sync();
            }
            uint8_t val[NP];
#pragma unroll
            for (int j = 0; j < NP; ++j) val[j] = plane[ok[j] ? lane + WAVE * j : 0];
#pragma unroll
            for (int j = 0; j < NP; ++j) *(ok[j] ? dst + at[j] : trash) = val[j];
          }
        } else {
          for (int l = wave; l < L; l += nwave) {
            const uint8_t pad = pad_is_char ? outside : (uint8_t)(chars[l] == outside);
            const uint8_t* plane = lay + l * HW;
            uint8_t* dst = img + off + l * cells;
            for (int k = lane; k < cells; k += WAVE) {
              int vr = k / vw, vc = k - vr * vw;
              view_unrotate(dir, vw, vr, vc);
              const int r = vr + pr, c = vc + pc;
              dst[k] = ((unsigned)r < (unsigned)H && (unsigned)c < (unsigned)W) ? plane[r * W + c] : pad;
            }
          }
        }
      }
    }
    __syncthreads();
    // ---- the G rows out
#pragma nounroll
    for (int g = 0; g < G; ++g) {
      if (e0 + g < n) {
        uint8_t* gp = out + (e0 + g) * (long long)row_bytes;
        const int phase = (int)(reinterpret_cast<uintptr_t>(gp) & 15);
        const uint8_t* img = envs + g * env_lds + lay_bytes + phase;
        // head bytes up to the first 16-byte boundary, aligned 16-byte chunks, tail bytes
        const int head = ((16 - phase) & 15) < row_bytes ? ((16 - phase) & 15) : row_bytes;
        const int nq = (row_bytes - head) >> 4, done = head + 16 * nq;
        if ((int)threadIdx.x < head) gp[threadIdx.x] = img[threadIdx.x];
        for (int q = threadIdx.x; q < nq; q += blockDim.x)
          *reinterpret_cast<uint4*>(gp + head + 16 * q) = *reinterpret_cast<const uint4*>(img + head + 16 * q);
        if ((int)threadIdx.x < row_bytes - done) gp[done + threadIdx.x] = img[done + threadIdx.x];
      }
    }
    __syncthreads();                                         // the next group reuses the planes and the images
  }
}

// per-layer agent windows: out[e][agent][layer][vr][vc]; one wave per (env, agent, layer)
__global__ void k_agent_layer_views(const uint8_t* layers, const uint8_t* pos, const uint8_t* flags, long long n, ViewSpec v,
                                    const uint8_t* chars, int L, uint8_t outside, uint8_t* out, int lds_per_wave) {
  extern __shared__ __attribute__((aligned(16))) uint8_t view_lds[];
  const int lane = threadIdx.x & (WAVE - 1);
  uint8_t* lds = view_lds + (threadIdx.x >> 6) * lds_per_wave;
  const long long stride = ((long long)gridDim.x * blockDim.x) >> 6, windows = n * v.A * L;
  for (long long wv = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; wv < windows; wv += stride) {
    const long long ea = wv / L;                                 // env * A + agent
    const int li = (int)(wv - ea * L);
    const long long e = ea / v.A;
    const int ag = (int)(ea - e * v.A);
    if (v.vh[ag] * v.vw[ag] == 0) continue;
    const int pr = (int)pos[ea * 2] - v.up[ag], pc = (int)pos[ea * 2 + 1] - v.left[ag];
    const int dir = flags ? (flags[ea] >> 3) & 3 : 2;
    const int cells = v.vh[ag] * v.vw[ag];
    view_wave(layers + (e * L + li) * (long long)(v.H * v.W), out + e * ((long long)v.total * L) + (long long)v.off[ag] * L + (long long)li * cells,
              lds, v.H, v.W, v.vh[ag], v.vw[ag], pr, pc, dir, flags != nullptr, (uint8_t)(chars[li] == outside), lane);
  }
}

}  // namespace sgw
