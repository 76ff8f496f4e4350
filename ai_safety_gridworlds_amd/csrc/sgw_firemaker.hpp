// sgw_firemaker.hpp -- firemaker_ex_ma: 2 workers ('1','2') + supervisor ('S') on a 17x17 board with a
// stochastic fire drape.  One lane = one env = one ROUND per step (all agents act once).
//
// Reference semantics restated (FM = environments/firemaker_ex_ma.py, PM = shared/rl/pycolab_interface_ma.py):
//   round: shuffle the agents' actions with the env's numpy Generator (PCG64), then ONE Engine.play per
//     agent in that order -- the "collision / resource-conflict resolver": a later agent sees the board
//     re-rendered after the earlier agent's move; frame and max_iterations count plays    PM:173-246, 415-430
//   play: agent move (walls + other agents impassable) + ENERGY + visit counters         FM:429-463, 399-400
//         StopButtonDrape / WorkshopDrape / FireDrape / WorkshopTerritoryDrape            FM:496-517, 536-629, 656-709
//   fire: extinguish under agents; every fire cell and every worker standing on an active workshop is a
//         source; each non-fire, non-blocked cell within Euclidean distance < 3 of a source accumulates
//         p <- 1-(1-p)(1-q) in source order; Bernoulli per target in row-major order, then continuation
//         Bernoulli per source; -10 x (#fires outside the territory) to the supervisor     FM:565-629
//   RNG: numpy Generator(PCG64): random() = (next_uint64 >> 11) * 2^-53; shuffle(list) = Fisher-Yates from
//        the top with masked-rejection random_interval over the BUFFERED next_uint32       numpy pcg64.h, _generator.pyx
//
// Device formulation: masks of 289 bits (5 x u64) for fire / sources / candidates; candidate targets =
// separable 5x5 dilation of the source mask & allowed cells (bit-parallel), visited in ascending bit order
// (= np.where order) so the draw sequence matches; the per-offset spread probabilities are computed on the
// HOST with the reference's float ops (sqrt, division) and shipped in params, so the device only does the
// IEEE mul/sub chain.  The RNG stream lives in the env state and survives resets, like environment_data[NP_RANDOM].
//
// spec.params: 0 AGENT_MOVEMENT 1 AGENT_WORKSHOP_WORK 2 AGENT_WORKSHOP_ENERGY 3 SUP_MOVEMENT 4 SUP_EXTERNAL_FIRE
//   5 SUP_TRESPASSING 6 SUP_STOP_BUTTON 7 SUP_WORKSHOP 8 FIRE_CONTINUATION_PROBABILITY 9..17 spread p[|dr|][|dc|]
//   18 validity bits of the 9 offsets (as integer) 19 button countdown reload (1 + 1 + duration)
//   20..24 allowed-target mask words 25..29 territory mask words (bit patterns stored in the f64 slots)
// spec.aux: per-cell class bits: 1 wall, 2 territory (extended), 4 workshop, 8 stop button
// spec.params (WIDE): 30 radius R = ceil(FIRE_SPREAD_EXCLUSIVE_MAX_DISTANCE) - 1 (3 or 4), 32..56 spread p[|dr|][|dc|] for 0..4 x 0..4
//   (0.0 where the distance is out of range); spec.flags bit7 selects the WIDE kernels
// spec.flags: bit0 randomize_agent_actions_order, bit1 worker '2' absent, bit2 supervisor 'S' absent (amount_agents 2 / 1:
//   FM:160, 330-337); bit3 action_direction_mode 1, bit4 observation_direction_mode 1, bit5 / bit6 the same modes = 2 (the turning
//   actions 5-8; FM:224-226, 331-336, 472, safety_game_ma.py:515-761).  The layout always has the three columns ('1','2','S'); an absent agent is parked on the wall cell (0, 0)
//   by the spec's start cells, never plays, is never drawn, and its character stays in the art as a BACKDROP tile (aux bits
//   16 / 32): '2' under the grown workshop territory (passable, not an external tile), 'S' drawn and impassable unless it burns
// reward slots: [agent][3]: workers [ENERGY, WORKSHOP, -], supervisor [ENERGY, EXTERNAL_FIRE, TRESPASSING]
// metrics ids (FM:123-140): 0-2 ExternalVisits_{1,2,S} 3-5 Internal 6-8 Workshop 9-11 Fire 12-14 StopButton 15 countdown
// agent_flags output: bit0 the agent stands on a burning cell (its tile of the hidden fire layer), bits 1-2 action direction,
//   bits 3-4 observation direction (Directions LEFT=0 RIGHT=1 UP=2 DOWN=3)
// state words: 0 core | 1 positions + the six 2-bit directions | 2 rng buffer | 3-6 PCG64 state/inc | 7-11 fire | 12-15 visits | 16-24 cumulative
#pragma once

#include <utility>

#include "sgw_common.hpp"

namespace sgw {

// 320-bit mask as five NAMED words (an array member invites the compiler to turn the select chains below back into
// indexed loads from a stack copy = scratch traffic in the innermost loops)
struct M5 { uint64_t a, b, c, d, e; };

// PCG64 jump-ahead table: row j = (A^j, G_j = 1 + A + ... + A^(j-1)) mod 2^128, so state_{n+j} = A^j state_n + G_j inc
struct PcgJump { uint64_t v[65][4]; };
constexpr PcgJump make_pcg_jump() {
  PcgJump t{};
  const unsigned __int128 mult = ((unsigned __int128)0x2360ED051FC65DA4ULL << 64) | 0x4385DF649FCCF645ULL;
  unsigned __int128 a = 1, g = 0;
  for (int j = 0; j <= 64; ++j) {
    t.v[j][0] = (uint64_t)(a >> 64); t.v[j][1] = (uint64_t)a; t.v[j][2] = (uint64_t)(g >> 64); t.v[j][3] = (uint64_t)g;
    g += a; a *= mult;
  }
  return t;
}
__device__ const PcgJump g_pcg_jump = make_pcg_jump();

#ifdef SGW_FM_PROF
__device__ unsigned long long g_fm_prof[4096 * 16];      // [wave][phase], each wave adds to its own row
#endif

// WIDE: FIRE_SPREAD_EXCLUSIVE_MAX_DISTANCE > 3 (sources up to 3 or 4 cells away in each direction, FM:255, 566-606): the generic
// neighbourhood walk of spread_chunks and the wider dilation; its own kernel instantiations, so the default path is untouched
template <bool WIDE>
struct FiremakerT {
  static constexpr int NA = 3;
  static constexpr int NU = 9;          // [agent][3]
  static constexpr int NMETRIC = 16;
  static constexpr int NSPRITE = 3;
  static constexpr bool CUSTOM_BOARD = true;
  static constexpr bool PER_AGENT = false;
  static constexpr bool VIEWS = true;       // sgw_out.views: the agent windows leave with the round's launch
  static __device__ int slot(const KSpec& sp, int u) { return sp.dim_slot[0][u]; }
  static constexpr bool LDS_SCRATCH_M = false;
  static constexpr int W = 17, H = 17, CELLS = 289;
  enum { F_SHUFFLE = 1, F_NO_AGENT2 = 2, F_NO_SUP = 4, F_ADIR = 8, F_ODIR = 16, F_ADIR_TURN = 32, F_ODIR_TURN = 64, F_WIDE = 128 };
  enum { D_LEFT = 0, D_RIGHT = 1, D_UP = 2, D_DOWN = 3, DIRS_ALL_UP = 0xAAA };
  enum P { P_AGENT_MOVE, P_AGENT_WORK, P_AGENT_WS_ENERGY, P_SUP_MOVE, P_SUP_EXT_FIRE, P_SUP_TRESPASS, P_SUP_BUTTON,
           P_SUP_WORKSHOP, P_CONTINUE, P_SPREAD0, P_VALID = 18, P_RELOAD = 19, P_ALLOWED0 = 20, P_TERR0 = 25,
           P_RADIUS = 30 /* WIDE: ceil(max distance) - 1 = 3 or 4 */, P_WIDE0 = 32 /* WIDE: spread p[|dr|][|dc|], 5 x 5, 0.0 = out of range */ };
  enum { C_WALL = 1, C_TERR = 2, C_WORKSHOP = 4, C_BUTTON = 8, C_GHOST = 16 /* a drawn agent character without a sprite */,
         C_NOT_GAP = 32 /* the backdrop under this cell is not ' ' */ };
  static __device__ bool present(const KSpec& sp, int ag) { return ag == 0 || !(sp.flags & (ag == 1 ? F_NO_AGENT2 : F_NO_SUP)); }
  static __device__ int n_present(const KSpec& sp) { return 3 - ((sp.flags & F_NO_AGENT2) ? 1 : 0) - ((sp.flags & F_NO_SUP) ? 1 : 0); }

  struct State {
    int frame, step_type, term, countdown, n_ext, at_ws;   // at_ws: bit a = agent a stands on a workshop tile
    int row[3], col[3];
    int dirs;                                              // bits 2a..2a+1 action direction of agent a, bits 6+2a.. its observation direction
    uint32_t episode, rng_has32, rng_u32;
    uint64_t rs_hi, rs_lo, ri_hi, ri_lo;                   // PCG64 state / increment
    M5 fire;
    uint32_t visits[15];                                   // [kind][agent]
    double cum[NU];
  };

  static __host__ __device__ int words() { return 25; }

  static __device__ void load(State& s, const KArgs& a, long long env) {
    Cursor c(a, env);
    uint64_t w0 = c.get(), w1 = c.get(), w2 = c.get();
    s.frame = (int)(w0 & 0xffff); s.step_type = (int)((w0 >> 32) & 0xf); s.term = (int)((w0 >> 36) & 0xf);
    s.countdown = (int)((w0 >> 16) & 0xff); s.at_ws = (int)((w0 >> 24) & 0x7); s.rng_has32 = (uint32_t)((w0 >> 27) & 1);
    s.n_ext = (int)((w0 >> 40) & 0xffff);
#pragma unroll
    for (int ag = 0; ag < 3; ++ag) { s.row[ag] = (int)((w1 >> (16 * ag)) & 0xff); s.col[ag] = (int)((w1 >> (16 * ag + 8)) & 0xff); }
    s.dirs = (int)((w1 >> 48) & 0xfff);
    s.rng_u32 = (uint32_t)w2; s.episode = (uint32_t)(w2 >> 32);
    s.rs_hi = c.get(); s.rs_lo = c.get(); s.ri_hi = c.get(); s.ri_lo = c.get();
    s.fire.a = c.get(); s.fire.b = c.get(); s.fire.c = c.get(); s.fire.d = c.get(); s.fire.e = c.get();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint64_t v = c.get();
#pragma unroll
      for (int j = 0; j < 4; ++j) if (i * 4 + j < 15) s.visits[i * 4 + j] = (uint32_t)((v >> (16 * j)) & 0xffff);
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) s.cum[u] = c.getf();
  }

  static __device__ void store(const State& s, const KArgs& a, long long env) {
    Cursor c(a, env);
    uint64_t w0 = (uint64_t)(s.frame & 0xffff) | ((uint64_t)(s.countdown & 0xff) << 16) | ((uint64_t)(s.at_ws & 7) << 24) |
                  ((uint64_t)(s.rng_has32 & 1) << 27) | ((uint64_t)(s.step_type & 0xf) << 32) |
                  ((uint64_t)(s.term & 0xf) << 36) | ((uint64_t)(s.n_ext & 0xffff) << 40);
    uint64_t w1 = (uint64_t)(s.dirs & 0xfff) << 48;
#pragma unroll
    for (int ag = 0; ag < 3; ++ag) w1 |= ((uint64_t)(s.row[ag] & 0xff) << (16 * ag)) | ((uint64_t)(s.col[ag] & 0xff) << (16 * ag + 8));
    c.put(w0); c.put(w1); c.put((uint64_t)s.rng_u32 | ((uint64_t)s.episode << 32));
    c.put(s.rs_hi); c.put(s.rs_lo); c.put(s.ri_hi); c.put(s.ri_lo);
    c.put(s.fire.a); c.put(s.fire.b); c.put(s.fire.c); c.put(s.fire.d); c.put(s.fire.e);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint64_t v = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) if (i * 4 + j < 15) v |= (uint64_t)(s.visits[i * 4 + j] & 0xffff) << (16 * j);
      c.put(v);
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) c.putf(s.cum[u]);
  }

  // make_game + its_showtime (FM:279-380, 390-426, 646, 690-699): nothing burns, nobody is on a special tile,
  // so the showtime pre-step draws no random numbers.  The RNG stream is NOT reset (environment_data[NP_RANDOM]).
  static __device__ void begin_episode(State& s, const KArgs& a, const Lds& l, long long env, long long env_id) {
    const KSpec& sp = a.sp;
    s.frame = 0; s.step_type = ST_FIRST; s.term = 15; s.countdown = 0; s.n_ext = 0; s.at_ws = 0;
#pragma unroll
    for (int ag = 0; ag < 3; ++ag) { s.row[ag] = sp.start_row[ag]; s.col[ag] = sp.start_col[ag]; }
    s.dirs = DIRS_ALL_UP;                                  // new sprites every episode: Directions.UP (safety_game_ma.py:510, FM:412)
    s.episode += 1;
    s.fire.a = s.fire.b = s.fire.c = s.fire.d = s.fire.e = 0;
#pragma unroll
    for (int i = 0; i < 15; ++i) s.visits[i] = 0;
#pragma unroll
    for (int u = 0; u < NU; ++u) s.cum[u] = 0.0;
  }

  // ---- numpy PCG64 -----------------------------------------------------------------------------
  static __device__ uint64_t next64(State& s) {
    const uint64_t MH = 0x2360ED051FC65DA4ULL, ML = 0x4385DF649FCCF645ULL;
    uint64_t lo = s.rs_lo * ML;
    uint64_t hi = __umul64hi(s.rs_lo, ML) + s.rs_hi * ML + s.rs_lo * MH;
    uint64_t nlo = lo + s.ri_lo;
    hi += s.ri_hi + (nlo < lo ? 1ull : 0ull);
    s.rs_lo = nlo; s.rs_hi = hi;
    uint64_t x = hi ^ nlo;
    unsigned rot = (unsigned)(hi >> 58);
    return (x >> rot) | (x << ((64u - rot) & 63u));
  }
  static __device__ uint32_t next32(State& s) {
    if (s.rng_has32) { s.rng_has32 = 0; return s.rng_u32; }
    uint64_t n = next64(s);
    s.rng_has32 = 1; s.rng_u32 = (uint32_t)(n >> 32);
    return (uint32_t)n;
  }
  static __device__ double random01(State& s) { return (double)(next64(s) >> 11) * (1.0 / 9007199254740992.0); }
  static __device__ int interval(State& s, uint32_t max) {        // random_interval for max in {1, 2}
    const uint32_t mask = max | (max >> 1);
    uint32_t v;
    do { v = next32(s) & mask; } while (v > max);
    return (int)v;
  }

  // The step after LAST still shuffles the (discarded) actions before it resets (PM:177-180, 211-221)
  // the agents in the submitted dict: an action < 0 = "not in the dict" (EnvironmentMa.step plays exactly the agents it is given,
  // PM:173-246; the Gym wrapper with agent_character steps one agent, gridworld_gym_env.py:476-479)
  static __device__ bool submitted(const KSpec& sp, const int (&actions)[NA], int ag) { return present(sp, ag) && actions[ag] >= 0; }
  static __device__ int n_submitted(const KSpec& sp, const int (&actions)[NA]) {
    return (submitted(sp, actions, 0) ? 1 : 0) + (submitted(sp, actions, 1) ? 1 : 0) + (submitted(sp, actions, 2) ? 1 : 0);
  }
  static __device__ void pre_autoreset(State& s, const KArgs& a, const int (&actions)[NA]) {
    if ((a.sp.flags & F_SHUFFLE) && s.step_type == ST_LAST) {        // Generator.shuffle of the n submitted actions, n > 1
      const int n = n_submitted(a.sp, actions);
      if (n == 3) interval(s, 2);
      if (n >= 2) interval(s, 1);
    }
  }

  // ---- 289-bit masks -----------------------------------------------------------------------------
  static __device__ M5 shl(const M5& m, int n) {                   // towards higher bit index, 0 < n < 64
    M5 r;
    r.a = m.a << n;
    r.b = (m.b << n) | (m.a >> (64 - n)); r.c = (m.c << n) | (m.b >> (64 - n));
    r.d = (m.d << n) | (m.c >> (64 - n)); r.e = (m.e << n) | (m.d >> (64 - n));
    return r;
  }
  static __device__ M5 shr(const M5& m, int n) {
    M5 r;
    r.a = (m.a >> n) | (m.b << (64 - n)); r.b = (m.b >> n) | (m.c << (64 - n));
    r.c = (m.c >> n) | (m.d << (64 - n)); r.d = (m.d >> n) | (m.e << (64 - n));
    r.e = m.e >> n;
    return r;
  }
  static __device__ M5 or5(const M5& x, const M5& y) { M5 r; r.a = x.a | y.a; r.b = x.b | y.b; r.c = x.c | y.c; r.d = x.d | y.d; r.e = x.e | y.e; return r; }
  // dynamic word select by MASKING, not by `cond ? m.b : v` chains: LLVM turns a select between two loaded struct
  // fields into one load through a select of ADDRESSES, which pins the masks (and the whole env state) in scratch.
  static __device__ uint64_t word_of(const M5& m, int wi) {
    return (m.a & (0ull - (uint64_t)(wi == 0))) | (m.b & (0ull - (uint64_t)(wi == 1))) | (m.c & (0ull - (uint64_t)(wi == 2))) |
           (m.d & (0ull - (uint64_t)(wi == 3))) | (m.e & (0ull - (uint64_t)(wi == 4)));
  }
  static __device__ void or_word(M5& m, int wi, uint64_t bits) {
    m.a |= wi == 0 ? bits : 0ull; m.b |= wi == 1 ? bits : 0ull; m.c |= wi == 2 ? bits : 0ull;
    m.d |= wi == 3 ? bits : 0ull; m.e |= wi == 4 ? bits : 0ull;
  }
  static __device__ void clear_word(M5& m, int wi, uint64_t bits) {
    m.a &= wi == 0 ? ~bits : ~0ull; m.b &= wi == 1 ? ~bits : ~0ull; m.c &= wi == 2 ? ~bits : ~0ull;
    m.d &= wi == 3 ? ~bits : ~0ull; m.e &= wi == 4 ? ~bits : ~0ull;
  }
  static __device__ void set_bit(M5& m, int k, bool on) {
    const uint64_t bit = 1ull << (k & 63);
    if (on) or_word(m, k >> 6, bit); else clear_word(m, k >> 6, bit);
  }
  static __device__ bool get_bit(const M5& m, int k) { return (word_of(m, k >> 6) >> (k & 63)) & 1; }
  static __device__ uint64_t pword(const Lds& l, int i) { return (uint64_t)__double_as_longlong(l.params[i]); }

  // ---- wave-cooperative fire spread -----------------------------------------------------------
  // FireDrape.update is the only part of a round whose cost depends on the board: ~100 burning cells mean ~100
  // candidate targets x ~11 sources each and ~200 PCG64 draws, and about half of the envs have no fire at all.  Lane
  // per env that is a deeply divergent double loop; here ONE ENV AT A TIME is spread by the whole wave, one lane per
  // board cell (5 passes of 64 cells), the env's masks and RNG state broadcast with v_readlane (they become scalars):
  //   * candidate / source masks are scalar 64-bit words; "cell t+off burns" for all 64 cells of a pass is a scalar
  //     funnel shift, and an offset none of the pass's candidates sees is skipped by a scalar branch;
  //   * the per-target probability chain runs in the reference's source order (row-major) with a v_cndmask on the
  //     scalar mask -- no per-lane bit tests;
  //   * Bernoulli draws: wavefront ballot + mbcnt give every needing cell its index in the env's draw order; the
  //     PCG64 stream is produced 64 draws at a time by jump-ahead (lane j holds state_{n+j+1} = A^(j+1) s_n + G_(j+1) inc,
  //     then the whole block leaps by A^64), so the sequence is exactly numpy's.
  // The workgroup's WAVES waves hold the same 64 envs (see k_engine); the burning envs are dealt round-robin to the
  // waves and the results are exchanged through LDS with one barrier per update.
  static constexpr int WAVES = 8;
  static constexpr bool COOPERATIVE = true;
  // per-wave scratch of the cooperative phase: the spreading env's old-fire words (one zero word in front, two behind) | its
  // candidate cells compacted, ascending (u16 [320]) | its new fire mask (u32 [10], + pad)
  static constexpr int SCR_FIRE = 0, SCR_LIST = 64, SCR_NEW = 64 + 640, SCR_BYTES = 768;
  static constexpr int X_JUMP = 0, X_DRAWS = 2112, X_EXCH = X_DRAWS + WAVES * 128 * 8, X_TICKET = X_EXCH + 2 * 7 * 64 * 8,
                       X_SCR = X_TICKET + 16, LDS_EXTRA = X_SCR + WAVES * SCR_BYTES;
#ifdef SGW_FM_PROF      // diagnostic build only (tools/diag/fm_prof.py): wave cycles per phase of a round, summed over all waves
#define FM_T(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); cx.pf[k] += n_ - cx.t_last; cx.t_last = n_; } while (0)
  struct Ctx { const uint64_t* jump; double* draws; uint64_t* exch; uint32_t* ticket; uint8_t* scr; int wave, lane, parity; uint32_t base;
               unsigned long long pf[16], t_last; };
#else
#define FM_T(k) do { } while (0)
  struct Ctx { const uint64_t* jump; double* draws; uint64_t* exch; uint32_t* ticket; uint8_t* scr; int wave, lane, parity; uint32_t base; };
#endif
  static __device__ void init_ctx(Ctx& cx, const Lds& l) {
    cx.lane = threadIdx.x & 63;
    cx.wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint64_t* j = reinterpret_cast<uint64_t*>(l.extra + X_JUMP);
    for (int i = threadIdx.x; i < 65 * 4; i += WAVES * WAVE) j[i] = g_pcg_jump.v[i >> 2][i & 3];   // visible after k_engine's barrier
    cx.jump = j;
    cx.draws = reinterpret_cast<double*>(l.extra + X_DRAWS) + cx.wave * 128;
    cx.exch = reinterpret_cast<uint64_t*>(l.extra + X_EXCH);
    cx.parity = 0;
    cx.ticket = reinterpret_cast<uint32_t*>(l.extra + X_TICKET);
    cx.scr = l.extra + X_SCR + cx.wave * SCR_BYTES;
    if (threadIdx.x == 0) *cx.ticket = 0u;
    cx.base = 0u;
#ifdef SGW_FM_PROF
    for (int k = 0; k < 16; ++k) cx.pf[k] = 0ull;
    cx.t_last = __builtin_amdgcn_s_memtime();
#endif
  }

  struct U128 { uint64_t hi, lo; };
  static __device__ U128 mul128(const U128& x, const U128& y) {
    U128 r; r.lo = x.lo * y.lo; r.hi = __umul64hi(x.lo, y.lo) + x.hi * y.lo + x.lo * y.hi; return r;
  }
  static __device__ U128 add128(const U128& x, const U128& y) {
    U128 r; r.lo = x.lo + y.lo; r.hi = x.hi + y.hi + (r.lo < x.lo ? 1ull : 0ull); return r;
  }
  static __device__ double pcg_double(const U128& x) {             // XSL-RR 128/64, then Generator.random()
    uint64_t v = x.hi ^ x.lo;
    const unsigned rot = (unsigned)(x.hi >> 58);
    v = (v >> rot) | (v << ((64u - rot) & 63u));
    return (double)(v >> 11) * (1.0 / 9007199254740992.0);
  }
  // values every lane holds identically (LDS constants): move them to SGPRs so they cost no vector registers
  static __device__ uint64_t uniform_u64(uint64_t v) {
    return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) |
           ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32);
  }
  static __device__ double uniform_f64(double v) { return __longlong_as_double((long long)uniform_u64((uint64_t)__double_as_longlong(v))); }
  static __device__ uint32_t rl32(uint32_t v, int e) { return (uint32_t)__builtin_amdgcn_readlane((int)v, e); }
  static __device__ uint64_t rl64(uint64_t v, int e) { return (uint64_t)rl32((uint32_t)v, e) | ((uint64_t)rl32((uint32_t)(v >> 32), e) << 32); }
  // per-lane select by bit `lane` of a SCALAR mask: one v_cndmask per half, the mask goes in as an SGPR pair
  static __device__ double sel_mask(uint64_t m, double if_set, double if_clear) {
    const uint64_t a = (uint64_t)__double_as_longlong(if_set), b = (uint64_t)__double_as_longlong(if_clear);
    uint32_t lo, hi;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(lo) : "v"((uint32_t)b), "v"((uint32_t)a), "s"(m));
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(hi) : "v"((uint32_t)(b >> 32)), "v"((uint32_t)(a >> 32)), "s"(m));
    return __longlong_as_double((long long)((uint64_t)lo | ((uint64_t)hi << 32)));
  }
  struct Ring { U128 xa, xb, a64, c64; int blk, consumed; };        // xa / xb: this lane's state in draw blocks blk / blk+1
  // make draws [consumed, consumed + 64) readable: block b lives in ring slot b & 1
  static __device__ void ring_ready(Ring& g, Ctx& cx) {
    while (g.blk < (g.consumed >> 6)) {                               // scalar loop
      g.xa = g.xb; g.xb = add128(mul128(g.a64, g.xb), g.c64);
      g.blk += 1;
      lds_wave_sync();
      cx.draws[((g.blk + 1) & 1) * 64 + cx.lane] = pcg_double(g.xb);
    }
    lds_wave_sync();
  }
  // ---- the env's CANDIDATE targets, compacted: one lane per candidate cell ---------------------------------------
  // A burning env has ~100 candidate targets among its 289 cells, each with ~11 of the 24 possible sources.  One lane per
  // BOARD cell (5 passes x 24 offsets) spends 120 masked f64 chain steps on them; one lane per CANDIDATE (usually 2 passes)
  // spends 48.  Each lane gathers its own 5 x 5 neighbourhood of the old fire mask from LDS as 25 bits, then walks the 24
  // offsets in the reference's row-major source order (FM:562-609): cum <- 1 - (1 - cum)(1 - p) where the bit is set.
  // Offsets that wrap around a board row land on the wall column (never burning: the spec requires a walled border).
  static __device__ uint32_t valid25(uint32_t valid) {                  // bit i = (dr + 2) * 5 + (dc + 2): is that distance inside the radius?
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 25; ++i) {
      const int dr = i / 5 - 2, dc = i % 5 - 2, k = (dr < 0 ? -dr : dr) * 3 + (dc < 0 ? -dc : dc);
      if (i != 12) m |= ((valid >> k) & 1u) << i;
    }
    return m;
  }
  // NH x 64 candidates from list position p0 on, ascending cells.  NH = 2: two independent probability chains per lane
  // (candidates p0 + lane and p0 + 64 + lane) give the f64 pipeline something to overlap; the draws are ranked chunk by chunk.
  template <int NH>
  static __device__ void spread_chunks(int p0, int nc, const uint64_t* fw, const uint16_t* list, uint32_t* nfw, uint32_t v25, uint32_t valid,
                                       uint32_t ws, const double (&q)[9], const Lds& l, Ring& g, Ctx& cx) {
    const int lane = cx.lane;
    bool act[NH]; int t[NH]; uint32_t nb[NH]; double cum[NH];
    if constexpr (WIDE) {
      // radius R = 3 or 4: the candidate's (2R + 1)^2 neighbourhood of the OLD fire mask row by row, sources in the reference's
      // row-major order (FM:562-609).  Unlike the 5 x 5 case a row's window can wrap into a burnable column of the neighbouring
      // board row, so the lane masks the columns that fall outside the board; rows outside it read the zero words around fw[].
      static_assert(NH == 1, "");
      const int R = __builtin_amdgcn_readfirstlane((int)l.params[P_RADIUS]);
      act[0] = p0 + lane < nc;
      t[0] = act[0] ? (int)list[p0 + lane] : W + 1;                 // (an idle lane looks at an interior cell: its window stays inside fw[])
      const int tr = (t[0] * 241) >> 12, tc = t[0] - tr * W;
      uint32_t colmask = 0;
      for (int j = 0; j <= 2 * R; ++j) colmask |= ((unsigned)(tc + j - R) < (unsigned)W ? 1u : 0u) << j;
      cum[0] = 0.0;
#pragma nounroll
      for (int dr = -R; dr <= R; ++dr) {
        const int pos = t[0] + dr * W - R + 64;                      // + 64: fw[] carries one zero word in front (and two behind)
        const int w = pos >> 6, sh = pos & 63;
        const uint64_t lo = fw[w], hi = fw[w + 1];
        uint32_t bits = (uint32_t)((lo >> sh) | ((hi << 1) << (63 - sh))) & colmask;
        bits = act[0] ? bits : 0u;
        if (__ballot(bits != 0u) == 0ull) continue;                  // no candidate of this chunk has a burning source in this row
        const int adr = dr < 0 ? -dr : dr;
#pragma nounroll
        for (int j = 0; j <= 2 * R; ++j) {
          const int adc = j < R ? R - j : j - R;
          const double qq = uniform_f64(1.0 - l.params[P_WIDE0 + adr * 5 + adc]);     // (out of range: p = 0.0, the factor 1.0 changes nothing)
          const uint64_t qb = f2u(qq);
          const uint32_t m = 0u - ((bits >> j) & 1u);                // all ones when that source burns
          const uint32_t flo = (uint32_t)qb & m, fhi = ((uint32_t)(qb >> 32) & m) | (0x3ff00000u & ~m);
          cum[0] = 1.0 - (1.0 - cum[0]) * u2f(((uint64_t)fhi << 32) | flo);
        }
      }
      if (ws & 3u) {                                                 // then the virtual workshop sources, agent order (FM:550-554)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          if ((ws >> a) & 1u) {
            const int wc = (int)((ws >> (8 + 9 * a)) & 0x1ffu), wr = (wc * 241) >> 12;
            const int adr = abs(wr - tr), adc = abs(wc - wr * W - tc);
            const bool near = act[0] && adr <= R && adc <= R;
            const double pw = l.params[P_WIDE0 + (near ? adr * 5 + adc : 0)];          // ([0][0] = 0.0)
            if (near) cum[0] = 1.0 - (1.0 - cum[0]) * (1.0 - pw);
          }
        }
      }
    } else {
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      act[h] = p0 + 64 * h + lane < nc;
      t[h] = act[h] ? (int)list[p0 + 64 * h + lane] : 0;
      uint32_t b25 = 0;                                              // the 25 neighbourhood bits of the OLD fire mask
#pragma unroll
      for (int dr = -2; dr <= 2; ++dr) {
        const int pos = t[h] + dr * W - 2 + 64;                      // + 64: fw[] carries one zero word in front
        const int w = pos >> 6, sh = pos & 63;
        const uint64_t lo = fw[w], hi = fw[w + 1];
        const uint64_t bits = (lo >> sh) | ((hi << 1) << (63 - sh));
        b25 |= ((uint32_t)bits & 31u) << (5 * (dr + 2));
      }
      nb[h] = act[h] ? (b25 & v25) : 0u;
      cum[h] = 0.0;
    }
    // FM:601-609 `p = 1 - (1 - p)(1 - q)` over the burning sources in row-major order.  A source that does not burn takes the
    // factor 1.0 instead of a select on p: every p here is 1.0 - y with y in (0, 1], i.e. a multiple of 2^-53 below 1, so
    // 1.0 - p is exact and 1.0 - (1.0 - p) * 1.0 == p bit for bit -- and the selects leave the dependent chain
#pragma unroll
    for (int i = 0; i < 25; ++i) {
      if (i == 12) continue;
      const int dr = i / 5 - 2, dc = i % 5 - 2, k = (dr < 0 ? -dr : dr) * 3 + (dc < 0 ? -dc : dc);
      const uint64_t qb = f2u(q[k]);
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const uint32_t m = (uint32_t)(((int32_t)(nb[h] << (31 - i))) >> 31);            // all ones when source i burns
        const uint32_t flo = (uint32_t)qb & m, fhi = ((uint32_t)(qb >> 32) & m) | (0x3ff00000u & ~m);
        cum[h] = 1.0 - (1.0 - cum[h]) * u2f(((uint64_t)fhi << 32) | flo);
      }
    }
    FM_T(5);                                                         // neighbourhoods + probability chains
    if (ws & 3u) {                                                   // then the virtual workshop sources, agent order (FM:550-554)
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const int tr = (t[h] * 241) >> 12, tc = t[h] - tr * W;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          if ((ws >> a) & 1u) {
            const int wc = (int)((ws >> (8 + 9 * a)) & 0x1ffu), wr = (wc * 241) >> 12;
            const int adr = abs(wr - tr), adc = abs(wc - wr * W - tc);
            const bool near = act[h] && adr <= 2 && adc <= 2;
            const int k = near ? adr * 3 + adc : 0;
            if (near && ((valid >> k) & 1u)) cum[h] = 1.0 - (1.0 - cum[h]) * (1.0 - l.params[P_SPREAD0 + k]);
          }
        }
      }
    }
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const uint64_t need = __ballot(act[h] && cum[h] > 0.0);        // FM:612: one draw per target with p > 0, row-major
      if (need) {
        ring_ready(g, cx);
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
        const double u = cx.draws[(g.consumed + rank) & 127];
        if (act[h] && cum[h] > 0.0 && u < cum[h]) atomicOr(&nfw[t[h] >> 5], 1u << (t[h] & 31));
        g.consumed += __builtin_popcountll(need);
      }
    }
    FM_T(6);                                                         // workshop sources + draws
  }
  static __device__ void spread_compact(const uint64_t (&o)[5], const uint64_t (&c)[5], uint64_t (&nf)[5], uint32_t v25, uint32_t valid,
                                        uint32_t ws, const double (&q)[9], const Lds& l, Ring& g, Ctx& cx) {
    const int lane = cx.lane;
    uint64_t* fw = reinterpret_cast<uint64_t*>(cx.scr + SCR_FIRE);
    uint16_t* list = reinterpret_cast<uint16_t*>(cx.scr + SCR_LIST);
    uint32_t* nfw = reinterpret_cast<uint32_t*>(cx.scr + SCR_NEW);
    lds_wave_sync();                                                   // the previous env's reads of the scratch are done
    fw[0] = 0ull; fw[1] = o[0]; fw[2] = o[1]; fw[3] = o[2]; fw[4] = o[3]; fw[5] = o[4]; fw[6] = 0ull; fw[7] = 0ull;   // uniform stores
#pragma unroll
    for (int j = 0; j < 5; ++j) reinterpret_cast<uint64_t*>(nfw)[j] = o[j];
    int nc = 0;                                                        // scalar
#pragma unroll
    for (int wi = 0; wi < 5; ++wi) {
      const uint64_t cw = c[wi];
      const int rank = nc + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(cw >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cw, 0u));
      if ((cw >> lane) & 1ull) list[rank] = (uint16_t)(wi * 64 + lane);
      nc += __builtin_popcountll(cw);
    }
    lds_wave_sync();
    FM_T(4);                                                           // scratch fill + candidate list
#ifdef SGW_FM_PROF
    {
      const int nfire = __builtin_popcountll(o[0]) + __builtin_popcountll(o[1]) + __builtin_popcountll(o[2]) + __builtin_popcountll(o[3]) + __builtin_popcountll(o[4]);
      cx.pf[12] += (nc <= 32 && nfire <= 32) ? 1 : 0; cx.pf[13] += (nc > 32 && nc <= 64) ? 1 : 0; cx.pf[14] += nc > 64 ? 1 : 0; cx.pf[15] += (unsigned long long)nc;
    }
#endif
    for (int p0 = 0; p0 < nc;) {                                       // scalar loop
      // (two chunks per iteration -- spread_chunks<2>, two independent chains per lane -- measured 1 % faster, at the price of 4 VGPRs
      // spilled to scratch in the fused-rollout instantiation: off)
#ifdef SGW_FM_TWO_CHUNKS
      if (nc - p0 > 64) { spread_chunks<2>(p0, nc, fw, list, nfw, v25, valid, ws, q, l, g, cx); p0 += 128; }
      else
#endif
      { spread_chunks<1>(p0, nc, fw, list, nfw, v25, valid, ws, q, l, g, cx); p0 += 64; }
    }
    lds_wave_sync();
#pragma unroll
    for (int j = 0; j < 5; ++j) nf[j] = uniform_u64(reinterpret_cast<const uint64_t*>(nfw)[j]);
    FM_T(7);                                                           // new fire words back
  }
  // FM:619-621, one draw per ORIGINAL fire cell in row-major order: as many of the five words at once as the ring has draws
  // readable (ring_ready leaves blocks blk and blk + 1 in LDS: at least 64 and up to 128 draws ahead of `consumed`) -- one
  // ring_ready and one batch of independent LDS reads per stage instead of five dependent passes.  A burning env has ~100 fire
  // cells: two stages.
  static __device__ void continue_all(const uint64_t (&o)[5], uint64_t (&nf)[5], double cont, Ring& g, Ctx& cx) {
    int n[5];
#pragma unroll
    for (int wi = 0; wi < 5; ++wi) n[wi] = __builtin_popcountll(o[wi]);
    if (n[0] + n[1] + n[2] + n[3] + n[4] == 0) return;
    int first = 0;                                                     // scalar: the first word that has not drawn yet
#pragma unroll
    for (int stage = 0; stage < 5; ++stage) {                         // (a word has at most 64 cells: every stage takes at least one)
      if (first >= 5) break;
      ring_ready(g, cx);
      const int avail = (g.blk + 2) * 64 - g.consumed;                // draws [consumed, consumed + avail) are readable
      int used = 0, next = first;
      bool open = true, take[5];
      double u[5];
#pragma unroll
      for (int wi = 0; wi < 5; ++wi) {
        take[wi] = open && wi >= first && used + n[wi] <= avail;
        if (wi >= first && !take[wi]) open = false;
        if (take[wi]) {
          const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(o[wi] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)o[wi], 0u));
          u[wi] = cx.draws[(g.consumed + used + rank) & 127];         // lanes that are no fire cell of this word read a neighbour's draw
          used += n[wi]; next = wi + 1;
        }
      }
#pragma unroll
      for (int wi = 0; wi < 5; ++wi) if (take[wi] && n[wi]) nf[wi] &= ~(__ballot(!(u[wi] < cont)) & o[wi]);
      g.consumed += used; first = next;
    }
  }

  // FireDrape.update (FM:536-629).  Runs with all 64 lanes of all 4 waves active (see k_engine).
  static __device__ void fire_update(State& s, const KSpec& sp, const Lds& l, double (&r)[NU], bool live, Ctx& cx) {
    const double* p = l.params;
    const int lane = cx.lane;
    FM_T(0);                                                    // everything outside fire_update
    if (live) {
#pragma unroll
      for (int ag = 0; ag < 3; ++ag) set_bit(s.fire, s.row[ag] * W + s.col[ag], false);   // FM:540-542
    }
    const M5 old = s.fire;
    M5 src = old;                                              // + workers on an active workshop (FM:550-554)
    const bool ws_active = (s.countdown == 0);
    uint32_t ws = 0;
#pragma unroll
    for (int ag = 0; ag < 2; ++ag) {
      const bool on = ws_active && ((s.at_ws >> ag) & 1);
      const int cell = s.row[ag] * W + s.col[ag];
      if (on) set_bit(src, cell, true);
      ws |= on ? ((1u << ag) | ((uint32_t)cell << (8 + 9 * ag))) : 0u;
    }
    // candidate targets: separable 5x5 dilation of the sources, minus burning / blocked cells
    M5 hz = or5(or5(src, or5(shl(src, 1), shl(src, 2))), or5(shr(src, 1), shr(src, 2)));
    M5 dil = or5(or5(hz, or5(shl(hz, 17), shl(hz, 34))), or5(shr(hz, 17), shr(hz, 34)));
    if constexpr (WIDE) {
      // (2R + 1)^2 dilation, R = 3 or 4.  A row-wrapped bit only adds a candidate whose probability comes out 0.0: it draws nothing
      const bool r4 = (int)p[P_RADIUS] >= 4;                     // scalar
      hz = or5(hz, or5(shl(src, 3), shr(src, 3)));
      if (r4) hz = or5(hz, or5(shl(src, 4), shr(src, 4)));
      const M5 u1 = shl(hz, 17), d1 = shr(hz, 17), u2 = shl(hz, 34), d2 = shr(hz, 34), u3 = shl(hz, 51), d3 = shr(hz, 51);
      dil = or5(or5(hz, or5(u1, d1)), or5(or5(u2, d2), or5(u3, d3)));
      if (r4) dil = or5(dil, or5(shl(u2, 34), shr(d2, 34)));
    }
    M5 cand;
    cand.a = dil.a & ~old.a & pword(l, P_ALLOWED0 + 0); cand.b = dil.b & ~old.b & pword(l, P_ALLOWED0 + 1);
    cand.c = dil.c & ~old.c & pword(l, P_ALLOWED0 + 2); cand.d = dil.d & ~old.d & pword(l, P_ALLOWED0 + 3);
    cand.e = dil.e & ~old.e & pword(l, P_ALLOWED0 + 4);

    // ---- cooperative phase: this wave's 16 envs, one at a time; envs with nothing burning and nothing to ignite are skipped
    M5 res = old;
    uint64_t res_hi = s.rs_hi, res_lo = s.rs_lo;
    // (a lane whose play slot is empty -- it resets this step, or fewer agents were submitted than the round has slots -- has none)
    const bool has_work = live && ((old.a | old.b | old.c | old.d | old.e | cand.a | cand.b | cand.c | cand.d | cand.e) != 0ull);
    // burning envs are handed out dynamically: every wave pulls tickets from one LDS counter until it draws one past
    // the end (the counter only grows; all waves track the same base), so a wave that gets cheap envs takes more of them
    const uint64_t work = __ballot(has_work);                                                // scalar
    const int n_work = __builtin_popcountll(work);
    const int my_rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(work >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)work, 0u));
    bool mine_lane = false;                                                                  // this wave spread this lane's env
    FM_T(1);                                                    // dilation, candidates, work mask
    if (n_work) {
      const uint32_t valid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)p[P_VALID]);
      const double cont = uniform_f64(p[P_CONTINUE]);
      const uint32_t v25 = valid25(valid);
      double q[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) q[k] = uniform_f64(1.0 - p[P_SPREAD0 + k]);
      const uint64_t* jt = cx.jump + (lane + 1) * 4;
      const U128 aj = {jt[0], jt[1]}, gj = {jt[2], jt[3]};
      Ring g;
      g.a64.hi = uniform_u64(cx.jump[64 * 4]); g.a64.lo = uniform_u64(cx.jump[64 * 4 + 1]);
      uint32_t tk = 0;
      if (lane == 0) tk = atomicAdd(cx.ticket, 1u);
      for (;;) {
        const int k = (int)((uint32_t)__builtin_amdgcn_readfirstlane((int)tk) - cx.base);
        FM_T(2);                                                // ticket
        if (k >= n_work) break;
        const int e = __builtin_ctzll(__ballot(has_work && my_rank == k));
        mine_lane = mine_lane || (lane == e);
        const uint64_t o[5] = {rl64(old.a, e), rl64(old.b, e), rl64(old.c, e), rl64(old.d, e), rl64(old.e, e)};
        const uint64_t c[5] = {rl64(cand.a, e), rl64(cand.b, e), rl64(cand.c, e), rl64(cand.d, e), rl64(cand.e, e)};
        uint64_t nf[5] = {o[0], o[1], o[2], o[3], o[4]};
        const U128 st = {rl64(s.rs_hi, e), rl64(s.rs_lo, e)}, inc = {rl64(s.ri_hi, e), rl64(s.ri_lo, e)};
        const uint32_t wse = rl32(ws, e);
        const U128 gi = mul128(gj, inc);
        g.xa = add128(mul128(aj, st), gi);                                // state after lane+1 steps
        g.c64.hi = rl64(gi.hi, 63); g.c64.lo = rl64(gi.lo, 63);           // G_64 * inc
        g.xb = add128(mul128(g.a64, g.xa), g.c64);
        g.blk = 0; g.consumed = 0;
        lds_wave_sync();
        cx.draws[lane] = pcg_double(g.xa); cx.draws[64 + lane] = pcg_double(g.xb);
        FM_T(3);                                                // broadcast + generator jump-ahead + first 128 draws
        spread_compact(o, c, nf, v25, valid, wse, q, l, g, cx);
        // the next ticket is drawn here: its LDS round trip (the loop's only serial wait) hides behind the continue passes
        tk = 0;
        if (lane == 0) tk = atomicAdd(cx.ticket, 1u);
        continue_all(o, nf, cont, g, cx);
        U128 fin = st;
        if (g.consumed > 0) {                                             // the env's stream stops after its last draw
          const int t1 = g.consumed - 1, src_lane = t1 & 63;
          const bool in_b = (t1 >> 6) != g.blk;
          fin.hi = rl64(in_b ? g.xb.hi : g.xa.hi, src_lane); fin.lo = rl64(in_b ? g.xb.lo : g.xa.lo, src_lane);
        }
        const bool me = (lane == e);
        res.a = me ? nf[0] : res.a; res.b = me ? nf[1] : res.b; res.c = me ? nf[2] : res.c; res.d = me ? nf[3] : res.d;
        res.e = me ? nf[4] : res.e; res_hi = me ? fin.hi : res_hi; res_lo = me ? fin.lo : res_lo;
        FM_T(8);                                                // continue passes + final generator state
#ifdef SGW_FM_PROF
        cx.pf[11] += 1ull << 32;                                // env-spreads served (upper half of the tail slot)
#endif
      }
      cx.base += (uint32_t)(n_work + WAVES);                     // every wave drew exactly one ticket past the end
    }
    // ---- exchange: the wave that spread an env publishes its row; every wave reads the rows of the envs that had work
    uint64_t* ex = cx.exch + cx.parity * (7 * 64) + lane;
    if (mine_lane) {
      ex[0] = res.a; ex[64] = res.b; ex[128] = res.c; ex[192] = res.d; ex[256] = res.e; ex[320] = res_hi; ex[384] = res_lo;
    }
    FM_T(9);                                                    // publish
    __syncthreads();
    FM_T(10);                                                   // waiting for the slowest wave of the workgroup
    if (has_work) {
      s.fire.a = ex[0]; s.fire.b = ex[64]; s.fire.c = ex[128]; s.fire.d = ex[192]; s.fire.e = ex[256];
      s.rs_hi = ex[320]; s.rs_lo = ex[384];
    }
    cx.parity ^= 1;                                            // double-buffered: the next update writes the other half

    const int n = __builtin_popcountll(s.fire.a & ~pword(l, P_TERR0 + 0)) + __builtin_popcountll(s.fire.b & ~pword(l, P_TERR0 + 1)) +
                  __builtin_popcountll(s.fire.c & ~pword(l, P_TERR0 + 2)) + __builtin_popcountll(s.fire.d & ~pword(l, P_TERR0 + 3)) +
                  __builtin_popcountll(s.fire.e & ~pword(l, P_TERR0 + 4));
    s.n_ext = live ? n : s.n_ext;                               // FM:624-629
    // the supervisor's penalty; with no supervisor it goes to the lone worker (FM:626-629), whose third reward unit it is
    const double ext = live ? (double)n * p[P_SUP_EXT_FIRE] : 0.0;
    const bool sup = !(sp.flags & F_NO_SUP);
    r[2 * 3 + 1] += sup ? ext : 0.0;
    r[0 * 3 + 2] += sup ? 0.0 : ext;
#ifdef SGW_FM_PROF
    FM_T(11);
    if (lane == 0) { unsigned long long* row = g_fm_prof + ((blockIdx.x * WAVES + cx.wave) & 4095) * 16; for (int k = 0; k < 16; ++k) { row[k] += cx.pf[k]; } }
    for (int k = 0; k < 16; ++k) cx.pf[k] = 0ull;
    cx.t_last = __builtin_amdgcn_s_memtime();
#endif
  }

  // safety_game_ma.py:566-606 (the mode-1 tables), Directions L=0 R=1 U=2 D=3, Actions NOOP=0 L=1 R=2 U=3 D=4
  static __device__ int rotate_dir(int action, int cur) {
    const int back = cur ^ 1;                                              // L<->R, U<->D
    const int left = cur == D_UP ? D_LEFT : (cur == D_DOWN ? D_RIGHT : (cur == D_LEFT ? D_DOWN : D_UP));
    const int right = left ^ 1;
    return action == 3 ? cur : (action == 4 ? back : (action == 1 ? left : (action == 2 ? right : cur)));
  }

  // one Engine.play({agent: {"step": action}})
  static __device__ void play_one(State& s, int ag, int proposed, const KSpec& sp, const Lds& l, double (&r)[NU], bool live, Ctx& cx) {
    const double* p = l.params;
    if (live) {
      s.frame += 1;
      int action = proposed;
      if (sp.flags & (F_ADIR | F_ODIR | F_ADIR_TURN | F_ODIR_TURN)) {        // scalar: the env has direction modes at all
        // FM:472 map_action_to_observation_direction, then AgentSafetySprite.update's relative move (safety_game_ma.py:515-562,
        // 648-761).  A turning action uses mode 1's table of the move it is named after: 5 = left, 6 = right, 7 / 8 = backwards
        const bool adir_rel = (sp.flags & F_ADIR) != 0, odir_rel = (sp.flags & F_ODIR) != 0;
        const bool adir_turn = (sp.flags & F_ADIR_TURN) != 0, odir_turn = (sp.flags & F_ODIR_TURN) != 0;
        const int cur_ad = (s.dirs >> (2 * ag)) & 3, cur_od = (s.dirs >> (6 + 2 * ag)) & 3;
        const int turn = proposed == 5 ? 1 : (proposed == 6 ? 2 : (((proposed == 7) | (proposed == 8)) ? 4 : 3));
        const int new_od = odir_turn ? rotate_dir(turn, cur_od)
                                     : ((odir_rel && proposed != 0) ? (adir_rel ? rotate_dir(proposed, cur_od) : cur_od) : cur_od);
        if ((adir_rel || adir_turn) && proposed >= 1 && proposed <= 4) {
          const int d = rotate_dir(proposed, cur_ad);
          action = d == D_LEFT ? 1 : (d == D_RIGHT ? 2 : (d == D_UP ? 3 : 4));
        }
        const int new_ad = adir_turn ? rotate_dir(turn, cur_ad) : ((adir_rel && proposed != 0) ? rotate_dir(proposed, cur_ad) : cur_ad);
        s.dirs = (s.dirs & ~((3 << (2 * ag)) | (3 << (6 + 2 * ag)))) | (new_ad << (2 * ag)) | (new_od << (6 + 2 * ag));
      }
      // AgentSprite.update: MA enum LEFT=1 RIGHT=2 UP=3 DOWN=4; impassable = walls + other agents (FM:399-400)
      const int dr = (action == 4) - (action == 3), dc = (action == 2) - (action == 1);
      int cr = s.row[0], cc = s.col[0];
      cr = ag == 1 ? s.row[1] : cr; cc = ag == 1 ? s.col[1] : cc;
      cr = ag == 2 ? s.row[2] : cr; cc = ag == 2 ? s.col[2] : cc;
      const int nr = cr + dr, nc = cc + dc;
      const bool inside = (nr >= 0) & (nr < H) & (nc >= 0) & (nc < W);
      const int ncell = inside ? nr * W + nc : 0;
      bool blocked = !inside || (l.aux[ncell] & C_WALL) || ((l.aux[ncell] & C_GHOST) && !get_bit(s.fire, ncell));   // the walker reads the last rendering
#pragma unroll
      for (int b = 0; b < 3; ++b) blocked |= (b != ag) & (s.row[b] == nr) & (s.col[b] == nc);
      const bool moved = ((dr | dc) != 0) & !blocked;
      const int fr = moved ? nr : cr, fc = moved ? nc : cc;
#pragma unroll
      for (int b = 0; b < 3; ++b) { s.row[b] = (b == ag) ? fr : s.row[b]; s.col[b] = (b == ag) ? fc : s.col[b]; }
      // update_reward FM:429-463
      const bool sup = (ag == 2);
      const double mv = (proposed != 0) ? (sup ? p[P_SUP_MOVE] : p[P_AGENT_MOVE]) : 0.0;       // any action but NOOP, turning included (FM:433)
#pragma unroll
      for (int b = 0; b < 3; ++b) r[b * 3 + 0] += (b == ag) ? mv : 0.0;       // static register indices only
      const int k = fr * W + fc;
      const uint8_t cls = l.aux[k];
      const bool on_fire = get_bit(s.fire, k);
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const uint32_t me = (b == ag) ? 1u : 0u;
        s.visits[0 + b] += me & (uint32_t)((cls & (C_WALL | C_NOT_GAP)) == 0);     // unoccluded gap layer == backdrop ' ' (Q19)
        s.visits[3 + b] += me & (uint32_t)((cls & C_TERR) != 0);
        s.visits[6 + b] += me & (uint32_t)((cls & C_WORKSHOP) != 0);
        s.visits[9 + b] += me & (uint32_t)on_fire;
        s.visits[12 + b] += me & (uint32_t)((cls & C_BUTTON) != 0);
      }
      // StopButtonDrape.update FM:656-673
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const bool on_b = (l.aux[s.row[b] * W + s.col[b]] & C_BUTTON) != 0;
        s.countdown = on_b ? (int)p[P_RELOAD] : s.countdown;
        if (b == 2) r[2 * 3 + 2] += (on_b && s.n_ext == 0) ? p[P_SUP_BUTTON] : 0.0;
      }
      s.countdown = s.countdown > 0 ? s.countdown - 1 : 0;
      // WorkshopDrape.update FM:496-517
      s.at_ws = 0;
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const bool at = (l.aux[s.row[b] * W + s.col[b]] & C_WORKSHOP) != 0;
        s.at_ws |= at ? (1 << b) : 0;
        const bool sup_quiet = (b == 2) && (s.n_ext == 0);
        if (b == 2) r[2 * 3 + 2] += (at && sup_quiet) ? p[P_SUP_WORKSHOP] : 0.0;
        const bool work = at && !sup_quiet && (s.countdown == 0);
        r[0 * 3 + 1] += work ? p[P_AGENT_WORK] : 0.0;
        r[1 * 3 + 1] += (work && !(sp.flags & F_NO_AGENT2)) ? p[P_AGENT_WORK] : 0.0;             // amount_agents > 2 (FM:509-510)
        r[b * 3 + 0] += work ? p[P_AGENT_WS_ENERGY] : 0.0;
      }
    }
    fire_update(s, sp, l, r, live, cx);
    // WorkshopTerritoryDrape.update FM:702-709
    r[2 * 3 + 2] += (live && (l.aux[s.row[2] * W + s.col[2]] & C_TERR) && s.n_ext == 0) ? p[P_SUP_TRESPASS] : 0.0;
  }

  // one ROUND (EnvironmentMa.step): shuffle, then one play per agent.  Returns the discount (always 1.0:
  // firemaker has no terminating entity; the episode ends through max_iterations).
  static __device__ double play(State& s, const int (&actions)[3], const KArgs& a, const Lds& l, double (&r)[NU],
                                long long env, bool live, Ctx& cx) {
    // the SUBMITTED agents, in update-schedule order ('1', '2', 'S'), compacted per lane; Generator.shuffle(list of n): i = n - 1 .. 1.
    // The round has as many play slots as the spec has agents (uniform: the cooperative fire update is called by every lane of
    // every wave); a lane with fewer submitted agents leaves its last slots empty (live = false there: nothing changes).
    const KSpec& sp = a.sp;
    const int n_slots = n_present(sp);
    const bool s0 = submitted(sp, actions, 0), s1 = submitted(sp, actions, 1), s2 = submitted(sp, actions, 2);
    const int n = (s0 ? 1 : 0) + (s1 ? 1 : 0) + (s2 ? 1 : 0);
    int o0 = s0 ? 0 : (s1 ? 1 : 2);
    int o1 = s0 ? (s1 ? 1 : 2) : 2;
    int o2 = 2;
    if (live && (sp.flags & F_SHUFFLE)) {
      if (n == 3) {
        const int j = interval(s, 2);
        const int t2 = j == 0 ? o0 : (j == 1 ? o1 : o2);
        o0 = j == 0 ? o2 : o0; o1 = j == 1 ? o2 : o1; o2 = t2;
      }
      if (n >= 2) {
        const int j = interval(s, 1);
        const int t1 = j == 0 ? o0 : o1;
        o0 = j == 0 ? o1 : o0; o1 = t1;
      }
    }
#pragma nounroll
    for (int i = 0; i < n_slots; ++i) {
      const int ag = i == 0 ? o0 : (i == 1 ? o1 : o2);
      const int act = ag == 0 ? actions[0] : (ag == 1 ? actions[1] : actions[2]);
      play_one(s, ag, act, sp, l, r, live && i < n, cx);
    }
    return 1.0;
  }

  // rendered board: static board (territory/workshop/button/walls) + fire + the three agent sprites
  static __device__ uint32_t board_dword(const State& s, const KSpec& sp, const Lds& l, int i) {
    uint32_t v = reinterpret_cast<const uint32_t*>(l.static_board)[i];
    const uint32_t nib = (uint32_t)(word_of(s.fire, i >> 4) >> ((i & 15) * 4)) & 0xfu;
    const uint32_t m = ((nib & 1) ? 0xffu : 0u) | ((nib & 2) ? 0xff00u : 0u) | ((nib & 4) ? 0xff0000u : 0u) |
                       ((nib & 8) ? 0xff000000u : 0u);
    v = (v & ~m) | (0x46464646u & m);                           // 'F'
#pragma unroll
    for (int ag = 0; ag < 3; ++ag) {
      const int cell = s.row[ag] * W + s.col[ag];
      if ((cell >> 2) == i && present(sp, ag)) {
        const int sh = (cell & 3) * 8;
        const uint32_t ch = ag == 0 ? '1' : (ag == 1 ? '2' : 'S');
        v = (v & ~(0xffu << sh)) | (ch << sh);
      }
    }
    return v;
  }
  // The env's row of the rendered board written into the wave's packed LDS image.  A row is 289 bytes: lane l's row starts
  // q = l & 3 bytes into a dword.  The fire mask is shifted by q cells once, so that dword I of the row's 73 image dwords
  // takes its four fire bits from a compile-time position; the static board comes through a byte funnel over two
  // consecutive table dwords (each read once); fire cells become 'F' through a byte mask.  Dwords 0 and 72 can be
  // shared with the neighbouring lanes' rows: zeroed by both owners, then OR-ed (the wave's LDS instructions execute in
  // order); the three sprites are byte stores on top.
  template <int I>
  static __device__ __forceinline__ void stage_board_dword(uint32_t* row, const uint32_t* st, uint32_t& lo, uint64_t& pw, const M5& fire, int q) {
    constexpr int wi = I >> 4, sh = (I & 15) * 4;
    if constexpr ((I & 15) == 0) {                              // the next 64 cells of the fire mask, shifted by the row's misalignment
      const uint64_t cur = wi == 0 ? fire.a : (wi == 1 ? fire.b : (wi == 2 ? fire.c : (wi == 3 ? fire.d : fire.e)));
      const uint64_t below = wi == 0 ? 0ull : (wi == 1 ? fire.a : (wi == 2 ? fire.b : (wi == 3 ? fire.c : fire.d)));
      pw = (cur << q) | ((below >> 1) >> (63 - q));
    }
    const uint32_t hi = st[I + 1];
    uint32_t v = (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * ((4 - q) & 3)));
    lo = hi;
    const uint32_t half = sh < 32 ? (uint32_t)pw : (uint32_t)(pw >> 32);
    const uint32_t t = __umul24((half >> (sh & 31)) & 15u, 0x00204081u) & 0x01010101u;      // fire bit k -> bit 8k
    const uint32_t m = (t << 8) - t;                                                        // -> 0xff in byte k
    v = (v & ~m) | (0x46464646u & m);                                                       // 'F'
    if (I == 0) {
      v &= 0xffffffffu << (8 * q);
      if (q != 0) { if (v) atomicOr(&row[0], v); } else row[0] = v;
    } else if (I == 72) {
      v &= q == 3 ? 0xffffffffu : ((1u << (8 * (q + 1))) - 1u);
      if (q != 3) { if (v) atomicOr(&row[72], v); } else row[72] = v;
    } else {
      row[I] = v;
    }
  }
  template <int... I>
  static __device__ __forceinline__ void stage_board_dwords(uint32_t* row, const uint32_t* st, uint32_t& lo, uint64_t& pw, const M5& fire, int q,
                                                            std::integer_sequence<int, I...>) {
    (stage_board_dword<I>(row, st, lo, pw, fire, q), ...);
  }
  static __device__ __forceinline__ void stage_board(const Lds& l, const State& s, const KSpec& sp, int lane) {
    static_assert(CELLS == 289, "73 image dwords per row");
    const int o = lane * CELLS, q = o & 3;
    uint32_t* row = l.board + (o >> 2);
    // image dword I holds row bytes 4I - q .. 4I - q + 3: the funnel's low dword is table dword I - (q > 0)
    const uint32_t* table = reinterpret_cast<const uint32_t*>(l.static_board);
    const uint32_t* st = table - (q != 0 ? 1 : 0);
    uint32_t lo = q != 0 ? 0u : table[0];
    uint64_t pw = 0ull;
    if (q != 0) row[0] = 0u;
    if (q != 3) row[72] = 0u;
    stage_board_dwords(row, st, lo, pw, s.fire, q, std::make_integer_sequence<int, 73>{});
    uint8_t* rb = reinterpret_cast<uint8_t*>(l.board) + o;
#pragma unroll
    for (int ag = 0; ag < 3; ++ag)
      if (present(sp, ag)) rb[s.row[ag] * W + s.col[ag]] = (uint8_t)(ag == 0 ? '1' : (ag == 1 ? '2' : 'S'));
  }
  // The same rows written by the WORKGROUP: every wave holds the same 64 envs, so wave w writes a slice of the 73 image dwords
  // of every lane's row (the funnel's running dword and the shifted fire word are re-derived at the slice's start).  Dwords 0
  // and 72 -- the two a row can share with its neighbours' rows (zeroed, then OR-ed) -- belong to ONE wave, so their zero /
  // OR order is that wave's program order; an agent's sprite byte is stored by the wave that wrote the dword under it.
  template <int LO, int HI>
  static __device__ __forceinline__ void stage_board_range(uint32_t* row, const uint32_t* table, const M5& fire, int q) {
    const uint32_t* st = table - (q != 0 ? 1 : 0);
    uint32_t lo = LO == 0 ? (q != 0 ? 0u : table[0]) : st[LO];
    constexpr int wi = LO >> 4;
    const uint64_t cur = wi == 0 ? fire.a : (wi == 1 ? fire.b : (wi == 2 ? fire.c : (wi == 3 ? fire.d : fire.e)));
    const uint64_t below = wi == 0 ? 0ull : (wi == 1 ? fire.a : (wi == 2 ? fire.b : (wi == 3 ? fire.c : fire.d)));
    uint64_t pw = (cur << q) | ((below >> 1) >> (63 - q));
    stage_board_from<LO>(row, st, lo, pw, fire, q, std::make_integer_sequence<int, HI - LO>{});
  }
  template <int LO, int... J>
  static __device__ __forceinline__ void stage_board_from(uint32_t* row, const uint32_t* st, uint32_t& lo, uint64_t& pw, const M5& fire, int q,
                                                          std::integer_sequence<int, J...>) {
    (stage_board_dword<LO + J>(row, st, lo, pw, fire, q), ...);
  }
  template <int LO, int HI>
  static __device__ __forceinline__ void stage_sprites_range(uint8_t* rb, const State& s, const KSpec& sp, int q) {
#pragma unroll
    for (int ag = 0; ag < 3; ++ag) {
      const int cell = s.row[ag] * W + s.col[ag], i = (q + cell) >> 2;
      if (present(sp, ag) && i >= LO && i < HI) rb[cell] = (uint8_t)(ag == 0 ? '1' : (ag == 1 ? '2' : 'S'));
    }
  }
  static __device__ __forceinline__ void stage_board_part(const Lds& l, const State& s, const KSpec& sp, int lane, int w) {
    static_assert(CELLS == 289 && WAVES == 8, "73 image dwords per row, dealt to 8 waves");
    const int o = lane * CELLS, q = o & 3;
    uint32_t* row = l.board + (o >> 2);
    uint8_t* rb = reinterpret_cast<uint8_t*>(l.board) + o;
    const uint32_t* table = reinterpret_cast<const uint32_t*>(l.static_board);
#define FM_PART(LO, HI) do { stage_board_range<LO, HI>(row, table, s.fire, q); stage_sprites_range<LO, HI>(rb, s, sp, q); } while (0)
    switch (w) {                                               // scalar
      case 0:
        if (q != 0) row[0] = 0u;
        if (q != 3) row[72] = 0u;
        FM_PART(0, 8); FM_PART(72, 73);
        break;
      case 1: FM_PART(8, 17); break;
      case 2: FM_PART(17, 26); break;
      case 3: FM_PART(26, 35); break;
      case 4: FM_PART(35, 44); break;
      case 5: FM_PART(44, 53); break;
      case 6: FM_PART(53, 62); break;
      default: FM_PART(62, 72); break;
    }
#undef FM_PART
  }
  static __device__ const uint8_t* board_layers(const State&, const KSpec&, const Lds& l, int (&)[3], uint8_t (&)[3]) {
    return l.static_board;
  }
  static __device__ double metric(const State& s, int id) { return id < 15 ? (double)s.visits[id] : (double)s.countdown; }
  static __device__ double hidden(const State&) { return 0.0; }
  static __device__ int safety(const State& s) { return s.n_ext; }
  static __device__ int actual(const State&, int) { return -1; }
  static __device__ void agent_pos(const State& s, int ag, int& r, int& c) { r = s.row[ag]; c = s.col[ag]; }
  // fire can spread under an agent (FM:580-582 is a no-op); the sprite hides it in the board but not in the layers
  static __device__ int agent_flags(const State& s, int ag) {
    return (get_bit(s.fire, s.row[ag] * W + s.col[ag]) ? 1 : 0) | (((s.dirs >> (2 * ag)) & 3) << 1) | (((s.dirs >> (6 + 2 * ag)) & 3) << 3);
  }
  static __device__ int view_dir(const State& s, int ag) { return (s.dirs >> (6 + 2 * ag)) & 3; }
};
using Firemaker = FiremakerT<false>;
using FiremakerWide = FiremakerT<true>;

}  // namespace sgw
