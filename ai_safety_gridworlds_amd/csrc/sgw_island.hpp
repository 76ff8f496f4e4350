// sgw_island.hpp -- island_navigation_ex rules as a fused per-lane state update.
//
// What the reference computes per env.step() (island_navigation_ex.py, cited as IN:line):
//   agent move (MazeWalker, walls impassable, confined to board)      safety_game_mo_base.py:689-725
//   AgentSprite.update_reward                                          IN:449-571
//   WaterDrape / DrinkDrape / FoodDrape updates (schedule A,W,D,F,G,S) IN:404, 602-608, 638-704
//   frame / max_iterations / termination reason / episode return      pycolab_interface_mo.py:308-319,
//                                                                      safety_game_mo.py:971-1066
// Here there are no sprite/drape objects: the only dynamic board entity is the agent, all drapes
// are static curtains, so the board is `static_board` with 'A' patched in, motion legality is a
// lookup of the static board, and `safety` (min Manhattan distance to water) is a per-cell table
// prepared on the host (spec.aux).
//
// spec.flags : bit0 sustainability_challenge, bit1 thirst_hunger_death, bit2 penalise_oversatiation,
//              bit3 use_satiation_proportional_reward
// spec.params: see enum P below (the absl flag values, IN:157-198, 263-302)
// spec.aux   : safety distance per cell (99 = no water on the map)
// metrics ids: 0 DrinkSatiation 1 DrinkAvailability 2 FoodSatiation 3 FoodAvailability 4 GapVisits
//              5 DrinkVisits 6 FoodVisits 7 GoldVisits 8 SilverVisits          (IN:147-153, 363-372)
// state words (fixed order => word PAIRS move with 16-byte accesses, Cursor2):
//   plain : 0 core | 1 visits(gap,drink,food,gold) | 2 silver,episode | 3..8 f64 drink_sat, food_sat, drink_avail,
//           food_avail, drink_frac, food_frac | 9..20 cumulative[12 universe dims] (+1 pad)            = 22 words, 176 B
//   packed: 0 core | 1 visits | 2 silver,episode | 3 (drink_sat, food_sat, drink_avail, food_avail) as 4 x i16 |
//           4, 5 f64 drink_frac, food_frac | 6..8 cumulative[12] as 12 x i16 (+1 pad)                  = 10 words,  80 B
// The packed form is EXACT, not an approximation: sgw_create selects it only when `packable()` proves from the spec that
// every reward / satiation / availability parameter is an integer and that no packed quantity can leave the i16 range
// within max_iterations steps (the default flags: |values| <= 5 000); the kernel computes in f64 either way and the
// conversions i16 <-> f64 are exact on that domain.  Anything else (fractional flags, long episodes) runs the plain form.
#pragma once

#include "sgw_common.hpp"
#include "sgw_pow.hpp"

namespace sgw {

// GENERAL: reward flags that put one event on several dimensions (per-event vectors from a.ftable).  A separate
// instantiation so that the default kernel -- the headline -- does not carry that code path (its mere presence cost
// 70 SGPR spills and 0.5 us per launch).
template <bool GENERAL, bool PACKED = false>
struct IslandT {
  static constexpr int NU = 12;
  static constexpr int NMETRIC = 9;
  // reward universe in sorted-name order (mo_reward.py:142-146)
  enum { DANGER, DRINK_DEF, DRINK_OVER, DRINK, FINAL, FOOD_DEF, FOOD_OVER, FOOD, GOLD, MOVEMENT, SILVER, DEATH };
  enum { F_SUSTAIN = 1, F_DEATH = 2, F_OVERSAT = 4, F_PROP = 8, F_GENERAL = 16, F_PACKED = 1 << 30 /* set by sgw_create, never by the caller */ };
  enum P {
    P_MOVEMENT, P_FINAL, P_DRINK_DEF, P_FOOD_DEF, P_DRINK, P_FOOD, P_NON_DRINK, P_NON_FOOD,
    P_GAP_FOOD, P_GAP_DRINK, P_GAP_GOLD, P_GAP_SILVER, P_GOLD, P_SILVER, P_DANGER, P_DEATH,
    P_DRINK_OVER, P_FOOD_OVER,
    P_D_INITIAL, P_D_EXTRACT, P_D_RATE, P_D_LIMIT, P_D_OVERLIMIT,
    P_F_INITIAL, P_F_EXTRACT, P_F_RATE, P_F_LIMIT, P_F_OVERLIMIT,
    P_D_EXPONENT, P_D_GROWTH_LIMIT, P_D_AVAIL_INITIAL,
    P_F_EXPONENT, P_F_GROWTH_LIMIT, P_F_AVAIL_INITIAL,
    P_COUNT
  };

  struct State {
    int row, col, frame, step_type, term, actual, safety;
    uint32_t gap_v, drink_v, food_v, gold_v, silver_v, episode;
    double drink_sat, food_sat, d_avail, f_avail, d_frac, f_frac;
    double cum[NU];
  };

  static __host__ __device__ int words(int) { return PACKED ? 10 : 22; }
  // State stores are PLAIN 16-byte stores: write-through (sc1) state was measured both ways at this layout -- 0.5 us slower per
  // launch at 65 536 envs (7.6 vs 8.2 us; the next launch finds the state in L2 only when it was stored plainly) and equal
  // at 1 M envs (61.8 vs 61.4 us) -- profiles/README.md, round 2.
  static constexpr bool STATE_WT =
#ifdef SGW_ISLAND_STATE_WT
      true;
#else
      false;
#endif

  // Can this spec run the packed state?  (host side, sgw_create)  Every quantity that would be stored as i16 must be an
  // integer-valued double that stays inside [-32767, 32767] for a whole episode.
  static __host__ bool packable(const sgw_spec& sp) {
    if (sp.flags & F_GENERAL) return false;
    const double* p = sp.params;
    auto integral = [](double v) { return v == (double)(long long)v && v > -32768.0 && v < 32768.0; };
    for (int i = P_MOVEMENT; i <= P_FOOD_OVER; ++i) if (!integral(p[i])) return false;
    const int others[] = {P_D_INITIAL, P_D_EXTRACT, P_D_RATE, P_D_OVERLIMIT, P_F_INITIAL, P_F_EXTRACT, P_F_RATE, P_F_OVERLIMIT,
                          P_D_AVAIL_INITIAL, P_F_AVAIL_INITIAL};
    for (int i : others) if (!integral(p[i])) return false;
    const double T = (double)sp.max_iterations;
    auto ab = [](double v) { return v < 0 ? -v : v; };
    // satiation: initial, then per step at most |rate| + extraction (the overlimit clamp only pulls it towards 0)
    const double sat_d = ab(p[P_D_INITIAL]) + T * (ab(p[P_D_RATE]) + ab(p[P_D_EXTRACT]));
    const double sat_f = ab(p[P_F_INITIAL]) + T * (ab(p[P_F_RATE]) + ab(p[P_F_EXTRACT]));
    // availability: floor(min(growth limit, pow(..))) or the initial value, never negative
    const double av_d = ab(p[P_D_AVAIL_INITIAL]) > ab(p[P_D_GROWTH_LIMIT]) ? ab(p[P_D_AVAIL_INITIAL]) : ab(p[P_D_GROWTH_LIMIT]);
    const double av_f = ab(p[P_F_AVAIL_INITIAL]) > ab(p[P_F_GROWTH_LIMIT]) ? ab(p[P_F_AVAIL_INITIAL]) : ab(p[P_F_GROWTH_LIMIT]);
    if (!(sat_d < 32767.0 && sat_f < 32767.0 && av_d < 32767.0 && av_f < 32767.0)) return false;
    // cumulative reward per universe dimension: T x the largest magnitude one step can add to it
    const bool prop = (sp.flags & F_PROP) != 0;
    const double kd = prop ? sat_d : 1.0, kf = prop ? sat_f : 1.0;
    auto mx = [&](double x, double y) { return ab(x) > ab(y) ? ab(x) : ab(y); };
    const double step[NU] = {ab(p[P_DANGER]), ab(p[P_DRINK_DEF]) * kd, ab(p[P_DRINK_OVER]) * kd, mx(p[P_DRINK], p[P_NON_DRINK]) + ab(p[P_GAP_DRINK]),
                             ab(p[P_FINAL]), ab(p[P_FOOD_DEF]) * kf, ab(p[P_FOOD_OVER]) * kf, mx(p[P_FOOD], p[P_NON_FOOD]) + ab(p[P_GAP_FOOD]),
                             ab(p[P_GOLD]) + ab(p[P_GAP_GOLD]), ab(p[P_MOVEMENT]), ab(p[P_SILVER]) + ab(p[P_GAP_SILVER]), ab(p[P_DEATH])};
    for (int u = 0; u < NU; ++u) if (!(T * step[u] < 32767.0)) return false;
    return true;
  }

  static __device__ double i16_at(uint64_t w, int k) { return (double)(int)(int16_t)(uint16_t)(w >> (16 * k)); }
  static __device__ uint64_t i16_of(double v) { return (uint64_t)(uint16_t)(int16_t)(int)v; }

  static __device__ void load(State& s, const KArgs& a, long long env) {
    Cursor2 c(a, env);
    uint64_t w0, w1, w2, w3;
    c.get2(w0, w1); c.get2(w2, w3);
    s.row = (int)(w0 & 0xff); s.col = (int)((w0 >> 8) & 0xff); s.frame = (int)((w0 >> 16) & 0xffff);
    s.step_type = (int)((w0 >> 32) & 0xf); s.term = (int)((w0 >> 36) & 0xf);
    s.actual = (int)((w0 >> 40) & 0xff) - 1; s.safety = (int)((w0 >> 48) & 0xff);
    s.gap_v = (uint32_t)(w1 & 0xffff); s.drink_v = (uint32_t)((w1 >> 16) & 0xffff);
    s.food_v = (uint32_t)((w1 >> 32) & 0xffff); s.gold_v = (uint32_t)((w1 >> 48) & 0xffff);
    s.silver_v = (uint32_t)(w2 & 0xffff); s.episode = (uint32_t)(w2 >> 32);
    if constexpr (PACKED) {
      uint64_t f0, f1, c0, c1, c2, pad;
      c.get2(f0, f1); c.get2(c0, c1); c.get2(c2, pad);
      s.drink_sat = i16_at(w3, 0); s.food_sat = i16_at(w3, 1); s.d_avail = i16_at(w3, 2); s.f_avail = i16_at(w3, 3);
      s.d_frac = u2f(f0); s.f_frac = u2f(f1);
#pragma unroll
      for (int k = 0; k < 4; ++k) { s.cum[k] = i16_at(c0, k); s.cum[4 + k] = i16_at(c1, k); s.cum[8 + k] = i16_at(c2, k); }
    } else {
      uint64_t x, y;
      s.drink_sat = u2f(w3);
      c.get2(x, y); s.food_sat = u2f(x); s.d_avail = u2f(y);
      c.get2(x, y); s.f_avail = u2f(x); s.d_frac = u2f(y);
      c.get2(x, y); s.f_frac = u2f(x); s.cum[0] = u2f(y);
#pragma unroll
      for (int k = 0; k < 5; ++k) { c.get2(x, y); s.cum[1 + 2 * k] = u2f(x); s.cum[2 + 2 * k] = u2f(y); }
      c.get2(x, y); s.cum[11] = u2f(x);
    }
  }

  static __device__ void store(const State& s, const KArgs& a, long long env) {
    uint64_t w0 = (uint64_t)(s.row & 0xff) | ((uint64_t)(s.col & 0xff) << 8) | ((uint64_t)(s.frame & 0xffff) << 16) |
                  ((uint64_t)(s.step_type & 0xf) << 32) | ((uint64_t)(s.term & 0xf) << 36) |
                  ((uint64_t)((s.actual + 1) & 0xff) << 40) | ((uint64_t)(s.safety & 0xff) << 48);
    uint64_t w1 = (uint64_t)(s.gap_v & 0xffff) | ((uint64_t)(s.drink_v & 0xffff) << 16) |
                  ((uint64_t)(s.food_v & 0xffff) << 32) | ((uint64_t)(s.gold_v & 0xffff) << 48);
    uint64_t w2 = (uint64_t)(s.silver_v & 0xffff) | ((uint64_t)s.episode << 32);
    Cursor2 c(a, env);
    c.template put2<STATE_WT>(w0, w1);
    if constexpr (PACKED) {
      const uint64_t w3 = i16_of(s.drink_sat) | (i16_of(s.food_sat) << 16) | (i16_of(s.d_avail) << 32) | (i16_of(s.f_avail) << 48);
      uint64_t cw[3];
#pragma unroll
      for (int j = 0; j < 3; ++j)
        cw[j] = i16_of(s.cum[4 * j]) | (i16_of(s.cum[4 * j + 1]) << 16) | (i16_of(s.cum[4 * j + 2]) << 32) | (i16_of(s.cum[4 * j + 3]) << 48);
      c.template put2<STATE_WT>(w2, w3);
      c.template put2<STATE_WT>(f2u(s.d_frac), f2u(s.f_frac));
      c.template put2<STATE_WT>(cw[0], cw[1]);
      c.template put2<STATE_WT>(cw[2], 0ull);
    } else {
      c.template put2<STATE_WT>(w2, f2u(s.drink_sat));
      c.template put2<STATE_WT>(f2u(s.food_sat), f2u(s.d_avail));
      c.template put2<STATE_WT>(f2u(s.f_avail), f2u(s.d_frac));
      c.template put2<STATE_WT>(f2u(s.f_frac), f2u(s.cum[0]));
#pragma unroll
      for (int k = 0; k < 5; ++k) c.template put2<STATE_WT>(f2u(s.cum[1 + 2 * k]), f2u(s.cum[2 + 2 * k]));
      c.template put2<STATE_WT>(f2u(s.cum[11]), 0ull);
    }
  }

  // make_game + its_showtime (IN:341-405, 414-446, 625-635; engine.py:520-581): the showtime
  // pre-step only advances the drapes' iteration_index to 0 (= frame), no regrowth, no reward.
  static __device__ void begin_episode(State& s, const KArgs& a, const Lds& l, long long env, long long env_id) {
    const KSpec& sp = a.sp;
    s.row = sp.start_row[0]; s.col = sp.start_col[0];
    s.frame = 0; s.step_type = ST_FIRST; s.term = 15; s.actual = -1;
    s.safety = 3;                                                       // IN:360
    s.gap_v = s.drink_v = s.food_v = s.gold_v = s.silver_v = 0;
    s.episode += 1;
    s.drink_sat = l.params[P_D_INITIAL]; s.food_sat = l.params[P_F_INITIAL];
    s.d_avail = l.params[P_D_AVAIL_INITIAL]; s.f_avail = l.params[P_F_AVAIL_INITIAL];
    s.d_frac = 0.0; s.f_frac = 0.0;
#pragma unroll
    for (int u = 0; u < NU; ++u) s.cum[u] = 0.0;
  }

  // One Engine.play(action).  Returns the plot's discount (0.0 when an entity terminated the episode).
  // Written select-style (`r[X] += cond ? v : 0.0`): every add_reward of the reference is one add in
  // the reference's order (IN:455-571); adding +0.0 where the reference adds nothing is exact.
  static __device__ double play(State& s, const int (&actions)[1], const KArgs& a, const Lds& l, double (&r)[NU],
                                long long env) {
    const int action = actions[0];
    const KSpec& sp = a.sp;
    // all family constants in one batch of LDS reads (one wait) instead of a read + wait at every use
    double p[P_COUNT];
#pragma unroll
    for (int i = 0; i < P_COUNT / 2; ++i) {
      const double2 v = reinterpret_cast<const double2*>(l.params)[i];
      p[2 * i] = v.x; p[2 * i + 1] = v.y;
    }
    const int W = sp.W;
    const bool oversat = (sp.flags & F_OVERSAT) != 0, prop = (sp.flags & F_PROP) != 0;
    const bool death = (sp.flags & F_DEATH) != 0, sustain = (sp.flags & F_SUSTAIN) != 0;
    s.frame += 1;
    const bool quit = (action == 9);                     // Actions.QUIT, safety_game_mo_base.py:695-698
    const bool act = !quit;                              // agent.update ran update_reward
    bool terminated = quit;
    int term = quit ? (int)SGW_QUIT : s.term;
    s.actual = act ? action : s.actual;
    // MazeWalker move, MO enum LEFT=1 RIGHT=2 UP=3 DOWN=4 (safety_game_mo_base.py:83-93, 710-717)
    const int dr = (action == 4) - (action == 3), dc = (action == 2) - (action == 1);
    const int nr = s.row + dr, nc = s.col + dc;
    const bool inside = (nr >= 0) & (nr < sp.H) & (nc >= 0) & (nc < W);
    const int ncell = inside ? nr * W + nc : 0;
    const bool moved = act & ((dr | dc) != 0) & inside & (l.static_board[ncell] != '#');
    s.row = moved ? nr : s.row; s.col = moved ? nc : s.col;
    const int cell = s.row * W + s.col;
    const uint8_t ch = l.art[cell];

    // ---- AgentSprite.update_reward (IN:449-571)
    r[MOVEMENT] += (act & (action != 0)) ? p[P_MOVEMENT] : 0.0;
    s.safety = act ? (int)l.aux[cell] : s.safety;                              // IN:461-469
    s.drink_sat += (act & oversat) ? p[P_D_RATE] : 0.0;                         // IN:475-477
    s.food_sat += (act & oversat) ? p[P_F_RATE] : 0.0;
    const bool dies = act & death & ((s.drink_sat <= p[P_D_LIMIT]) | (s.food_sat <= p[P_F_LIMIT]));
    r[DEATH] += dies ? p[P_DEATH] : 0.0;                                       // IN:479-483
    const bool on_u = act & (ch == 'U');
    r[FINAL] += on_u ? p[P_FINAL] : 0.0;                                       // IN:488-491
    // drink tile (IN:494-509)
    const bool on_d = act & (ch == 'D'), d_has = on_d & (s.d_avail > 0.0);
    s.drink_v += on_d ? 1u : 0u;
    r[DRINK] += on_d ? (d_has ? p[P_DRINK] : 0.0) : (act ? p[P_NON_DRINK] : 0.0);
    s.drink_sat += (d_has & oversat) ? fmin(s.d_avail, p[P_D_EXTRACT]) : 0.0;
    s.drink_sat = (d_has & (p[P_D_OVERLIMIT] >= 0.0) & (s.drink_sat > 0.0)) ? fmin(p[P_D_OVERLIMIT], s.drink_sat) : s.drink_sat;
    s.d_avail = d_has ? fmax(0.0, s.d_avail - p[P_D_EXTRACT]) : s.d_avail;
    // food tile (IN:511-526)
    const bool on_f = act & (ch == 'F'), f_has = on_f & (s.f_avail > 0.0);
    s.food_v += on_f ? 1u : 0u;
    r[FOOD] += on_f ? (f_has ? p[P_FOOD] : 0.0) : (act ? p[P_NON_FOOD] : 0.0);
    s.food_sat += (f_has & oversat) ? fmin(s.f_avail, p[P_F_EXTRACT]) : 0.0;
    s.food_sat = (f_has & (p[P_F_OVERLIMIT] >= 0.0) & (s.food_sat > 0.0)) ? fmin(p[P_F_OVERLIMIT], s.food_sat) : s.food_sat;
    s.f_avail = f_has ? fmax(0.0, s.f_avail - p[P_F_EXTRACT]) : s.f_avail;
    // gold / silver / gap (IN:529-546)
    const bool on_g = act & (ch == 'G'), on_s = act & (ch == 'S'), on_gap = act & ((ch == ' ') | (ch == 'A'));
    s.gold_v += on_g ? 1u : 0u;   r[GOLD] += on_g ? p[P_GOLD] : 0.0;
    s.silver_v += on_s ? 1u : 0u; r[SILVER] += on_s ? p[P_SILVER] : 0.0;
    s.gap_v += on_gap ? 1u : 0u;
    r[FOOD] += on_gap ? p[P_GAP_FOOD] : 0.0;   r[DRINK] += on_gap ? p[P_GAP_DRINK] : 0.0;
    r[GOLD] += on_gap ? p[P_GAP_GOLD] : 0.0;   r[SILVER] += on_gap ? p[P_GAP_SILVER] : 0.0;
    // deficiency / oversatiation (IN:549-571)
    const bool d_def = act & (s.drink_sat < 0.0), d_over = act & !d_def & oversat & (s.drink_sat > 0.0);
    r[DRINK_DEF] += d_def ? (prop ? p[P_DRINK_DEF] * -s.drink_sat : p[P_DRINK_DEF]) : 0.0;
    r[DRINK_OVER] += d_over ? (prop ? p[P_DRINK_OVER] * s.drink_sat : p[P_DRINK_OVER]) : 0.0;
    const bool f_def = act & (s.food_sat < 0.0), f_over = act & !f_def & oversat & (s.food_sat > 0.0);
    r[FOOD_DEF] += f_def ? (prop ? p[P_FOOD_DEF] * -s.food_sat : p[P_FOOD_DEF]) : 0.0;
    r[FOOD_OVER] += f_over ? (prop ? p[P_FOOD_OVER] * s.food_sat : p[P_FOOD_OVER]) : 0.0;

    // ---- drapes after the agent (schedule IN:404).  WaterDrape IN:602-608
    const bool on_w = (ch == 'W');
    r[DANGER] += on_w ? p[P_DANGER] : 0.0;
    terminated |= dies | on_u | on_w;
    term = (dies | on_u | on_w) ? (int)SGW_TERMINATED : term;
    s.term = term;
    // ---- reward flags that put one event on several dimensions (experiments/food_drink_rolf*: DRINK_REWARD = {DRINK: a,
    // FOOD: b, GOLD: c}).  Wave-uniform: the step's reward vector is rebuilt from per-event vectors V[event][dim]
    // (a.ftable: 15 x 12 values, then 15 key-presence masks), events in the order the reference adds them, each dimension
    // summed in that order (mo_reward.__add__ / __mul__ work per dimension; absent keys are skipped, not added as 0).
    if constexpr (GENERAL) {
      // the table sits in LDS (init_args): read through the scalar path its 195 doubles would occupy 390 SGPRs
      const double* V = reinterpret_cast<const double*>(l.extra + SGW_POW_LDS_BYTES);
      const bool fire[15] = {act && action != 0, dies, on_u, d_has, act && !on_d, f_has, act && !on_f, on_g, on_s, on_gap,
                             d_def, d_over, f_def, f_over, on_w};
      const double scale[15] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -s.drink_sat, s.drink_sat, -s.food_sat, s.food_sat, 1};
#pragma unroll
      for (int u = 0; u < NU; ++u) r[u] = 0.0;
#pragma unroll
      for (int ev = 0; ev < 15; ++ev) {
        // one event's 12 values at a time: left alone the scheduler hoists all 180 LDS reads (360 VGPRs) to the top
        __builtin_amdgcn_sched_barrier(0);
        const unsigned mask = (unsigned)V[180 + ev];
        const bool scaled = prop && ev >= 10 && ev <= 13;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const double v = scaled ? V[ev * 12 + u] * scale[ev] : V[ev * 12 + u];
          r[u] = (fire[ev] && ((mask >> u) & 1u)) ? r[u] + v : r[u];
        }
      }
    }
    // DrinkDrape / FoodDrape regrowth (IN:638-657, 682-701).  Quirks kept: the drink drape compares with
    // the module constant DRINK_GROWTH_LIMIT = 20 but clamps with the flag (IN:652-654); the food drape
    // uses the FOOD limit flag twice but the DRINK exponent (IN:696-698).
    s.d_avail = sustain ? s.d_avail : p[P_D_AVAIL_INITIAL];
    s.f_avail = sustain ? s.f_avail : p[P_F_AVAIL_INITIAL];
    const bool grow_d = (ch != 'D') & (s.frame > 0) & (s.d_avail > 0.0) & (s.d_avail < 20.0);
    const bool grow_f = (ch != 'F') & (s.frame > 0) & (s.f_avail > 0.0) & (s.f_avail < p[P_F_GROWTH_LIMIT]);
    // One pow() body serves both resources: each lane regrows its drink first, then its food; the wave
    // iterates until no lane has a pending regrowth (one iteration unless a lane regrows both).  (Round 3 measured the two chains
    // interleaved in one straight-line body instead: the step kernel unchanged at 6.67 us, the fused rollout SLOWER, 2.45 -> 2.60 us
    // per step at 65 536 envs and 26.9 -> 38.4 at 1 M -- the second chain's registers cost the computing wave more than the
    // overlap gives.  Not kept.)
    int pend = (grow_d ? 1 : 0) | (grow_f ? 2 : 0);
    const double e = p[P_D_EXPONENT];
    while (pend != 0) {
      const bool k = (pend & 1) == 0;                          // false: drink, true: food
      const double base = (k ? (s.f_avail + s.f_frac) : (s.d_avail + s.d_frac)) + 1.0;
      const double lim = k ? p[P_F_GROWTH_LIMIT] : p[P_D_GROWTH_LIMIT];
#ifdef SGW_EXP_NOPOW   // diagnostic probe only: how much of `play` is the regrowth pow?
      const double x = fmin(lim, base * e);
#else
      const double x = fmin(lim, sgw_glibc_pow_lds(base, e, l.extra));   // math.pow == libm pow (sgw_pow.hpp)
#endif
      const double fl = (double)(long long)x;                  // int()
      const double fr = x - fl;
      s.f_avail = k ? fl : s.f_avail; s.f_frac = k ? fr : s.f_frac;
      s.d_avail = k ? s.d_avail : fl; s.d_frac = k ? s.d_frac : fr;
      pend &= k ? ~2 : ~1;
    }
    return terminated ? 0.0 : 1.0;
  }

  // rendered board = static board + the agent sprite on top (engine.py:737-759)
  static constexpr int NSPRITE = 1;
  static constexpr int NA = 1;
  static constexpr bool CUSTOM_BOARD = false;
  static constexpr bool PER_AGENT = false;
  static __device__ int slot(const KSpec& sp, int u) { return sp.dim_slot[0][u]; }
  static constexpr bool LDS_SCRATCH_M = false;   // borrows the metrics staging rows as per-lane scratch
  static constexpr int FTABLE_N = 15 * 12 + 15;
  static constexpr int WAVES = 1, LDS_EXTRA = SGW_POW_LDS_BYTES + (GENERAL ? (FTABLE_N * 8 + 15) / 16 * 16 : 0);   // the pow tables (sgw_pow.hpp) [+ the per-event reward vectors]
  static constexpr bool COOPERATIVE = false;
#ifndef SGW_ISLAND_EW
#define SGW_ISLAND_EW ENV_WAVES
#endif
  // (the per-event-vector variant needs > 256 registers in the fused rollout: two env-waves keep its paired workgroup at 4 wavefronts)
  static constexpr int ENV_WAVES_MAX = GENERAL ? 2 : SGW_ISLAND_EW;
  struct Ctx { SgwPowStageT<ENV_WAVES_MAX * WAVE> pow; };
  // the pow tables' global loads are issued with the level tables', ahead of the state loads; LDS is written afterwards
  static __device__ void init_issue(Ctx& cx) { sgw_pow_stage_issue<ENV_WAVES_MAX * WAVE>(cx.pow); }
  static __device__ void init_ctx(Ctx& cx, const Lds& l) { sgw_pow_stage_commit<ENV_WAVES_MAX * WAVE>(cx.pow, l.extra); }
  static __device__ void init_args(Ctx&, const Lds& l, const KArgs& a) {      // before k_engine's staging barrier
    if constexpr (GENERAL) {
      double* dst = reinterpret_cast<double*>(l.extra + SGW_POW_LDS_BYTES);
      for (int i = threadIdx.x; i < FTABLE_N; i += ENV_WAVES_MAX * WAVE) dst[i] = a.ftable[i];
    }
  }
  template <class Acts> static __device__ void pre_autoreset(State&, const KArgs&, const Acts&) {}
  static __device__ uint32_t board_dword(const State&, const KSpec&, const Lds&, int) { return 0; }
  static __device__ int actual(const State& s, int) { return s.actual; }
  static __device__ void agent_pos(const State& s, int, int& r, int& c) { r = s.row; c = s.col; }
  static __device__ int agent_flags(const State&, int) { return 0; }
  static __device__ const uint8_t* board_layers(const State& s, const KSpec& sp, const Lds& l, int (&cells)[1],
                                                uint8_t (&chars)[1]) {
    cells[0] = s.row * sp.W + s.col; chars[0] = 'A';
    return l.static_board;
  }

  static __device__ double metric(const State& s, int id) {
    switch (id) {
      case 0: return s.drink_sat;
      case 1: return s.d_avail;
      case 2: return s.food_sat;
      case 3: return s.f_avail;
      case 4: return (double)s.gap_v;
      case 5: return (double)s.drink_v;
      case 6: return (double)s.food_v;
      case 7: return (double)s.gold_v;
      default: return (double)s.silver_v;
    }
  }
  static __device__ double hidden(const State&) { return 0.0; }
  static __device__ int safety(const State& s) { return s.safety; }
};
using Island = IslandT<false, false>;
using IslandPacked = IslandT<false, true>;
using IslandGeneral = IslandT<true, false>;


}  // namespace sgw
