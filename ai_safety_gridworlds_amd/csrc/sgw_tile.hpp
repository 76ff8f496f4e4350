// sgw_tile.hpp -- "tile event" games of the original DeepMind suite behind ONE table-driven family:
//   island_navigation (IV = environments/island_navigation.py), distributional_shift (DS), absent_supervisor (AS).
//
// What they share (SG = shared/safety_game.py): an AgentSafetySprite that walks ('#' impassable, SG:400-432), pays a
// movement reward every step -- NOOP included -- and triggers events keyed by the character under it: an observed
// reward, a hidden reward, episode termination (IV:123-161, DS:116-126, AS:113-138).  What differs is data:
//   * up to two BOARD VARIANTS chosen per game build by one number of the process-global numpy RNG
//     (DS:93-97 np.random.choice([1, 2]) when testing; AS:91-93 np.random.rand() < 0.5): static_board = variant 0,
//     aux = variant 1; the bit comes from sgw_set_episode_bits (replay) or Philox(seed, env id, episode) < P_PROB;
//   * an event table in spec.params: per event (chr, observed reward in variant 0 / 1, hidden reward, terminates, covers):
//     `covers` = the tile is drawn OVER the agent (island_navigation's water drape follows the agent in the z-order).
//
// spec.art   : per-cell value of environment_data['safety'] (island_navigation: Manhattan distance to the nearest water)
// spec.params: P_MOVE_OBS, P_MOVE_HID, P_PROB, P_FIXED (-1 = draw the variant, else 0/1), P_NEVENTS, P_SAFETY_MODE
//              (0: report the variant bit, 1: report the distance table), then P_EV0 + 6*i + {chr, obs0, obs1, hid, term, covers}
// state words: 0 core (bit 48 variant, bits 49-56 safety) | 1 hidden | 2 cumulative | 3 episode
#pragma once

#include "sgw_common.hpp"

namespace sgw {

struct Tile {
  static constexpr int NU = 1;
  static constexpr int NMETRIC = 1;
  static constexpr int MAX_EVENTS = 6;
  enum P { P_MOVE_OBS, P_MOVE_HID, P_PROB, P_FIXED, P_NEVENTS, P_SAFETY_MODE, P_EV0, P_COUNT = P_EV0 + 6 * MAX_EVENTS };
  enum { E_CHR, E_OBS0, E_OBS1, E_HID, E_TERM, E_COVERS };

  struct State {
    int row, col, frame, step_type, term, actual, variant, safety;
    uint32_t episode;
    double hidden;
    double cum[NU];
  };

  static __host__ __device__ int words() { return 4; }

  static __device__ void load(State& s, const KArgs& a, long long env) {
    uint64_t w0 = ld_word(a, 0, env);
    s.row = (int)(w0 & 0xff); s.col = (int)((w0 >> 8) & 0xff); s.frame = (int)((w0 >> 16) & 0xffff);
    s.step_type = (int)((w0 >> 32) & 0xf); s.term = (int)((w0 >> 36) & 0xf);
    s.actual = (int)((w0 >> 40) & 0xff) - 1;
    s.variant = (int)((w0 >> 48) & 1); s.safety = (int)((w0 >> 49) & 0xff);
    s.hidden = ld_f64(a, 1, env);
    s.cum[0] = ld_f64(a, 2, env);
    s.episode = (uint32_t)ld_word(a, 3, env);
  }
  static __device__ void store(const State& s, const KArgs& a, long long env) {
    uint64_t w0 = (uint64_t)(s.row & 0xff) | ((uint64_t)(s.col & 0xff) << 8) | ((uint64_t)(s.frame & 0xffff) << 16) |
                  ((uint64_t)(s.step_type & 0xf) << 32) | ((uint64_t)(s.term & 0xf) << 36) |
                  ((uint64_t)((s.actual + 1) & 0xff) << 40) | ((uint64_t)(s.variant & 1) << 48) | ((uint64_t)(s.safety & 0xff) << 49);
    st_word(a, 0, env, w0);
    st_f64(a, 1, env, s.hidden);
    st_f64(a, 2, env, s.cum[0]);
    st_word(a, 3, env, (uint64_t)s.episode);
  }

  static __device__ void begin_episode(State& s, const KArgs& a, const Lds& l, long long env, long long env_id) {
    const KSpec& sp = a.sp;
    s.row = sp.start_row[0]; s.col = sp.start_col[0];
    s.frame = 0; s.step_type = ST_FIRST; s.term = 15; s.actual = -1;
    s.hidden = 0.0; s.cum[0] = 0.0;
    const int fixed = (int)l.params[P_FIXED];
    if (fixed >= 0) {
      s.variant = fixed;                               // no draw: the reference does not touch the RNG either
    } else {
      // one draw per game build; the k-th build of an env uses bit k
      if (a.ep_bits) s.variant = (env < a.n_envs) ? (a.ep_bits[env * a.ep_bits_n + (s.episode % (uint32_t)a.ep_bits_n)] != 0) : 0;
      else s.variant = episode_uniform(a.ep_seed, env_id, s.episode) < l.params[P_PROB];
      s.episode += 1;
    }
    s.safety = l.params[P_SAFETY_MODE] != 0.0 ? 3 : s.variant;       // IV:104 environment_data['safety'] = 3
  }

  static __device__ double play(State& s, const int (&actions)[1], const KArgs& a, const Lds& l, double (&r)[NU],
                                long long env) {
    const int action = actions[0];
    const KSpec& sp = a.sp;
    const double* p = l.params;
    const int W = sp.W;
    s.frame += 1;
    if (action == 9) { s.term = SGW_QUIT; return 0.0; }               // Actions.QUIT, SG:408-411
    s.actual = action;
    const uint8_t* shown = s.variant ? l.aux : l.static_board;
    const int dr = (action == 2) - (action == 1);                      // original enum: UP=1 DOWN=2 LEFT=3 RIGHT=4
    const int dc = (action == 4) - (action == 3);
    const int nr = s.row + dr, nc = s.col + dc;
    const bool inside = (nr >= 0) & (nr < sp.H) & (nc >= 0) & (nc < W);
    if ((dr | dc) != 0 && inside && shown[nr * W + nc] != '#') { s.row = nr; s.col = nc; }
    const int cell = s.row * W + s.col;
    r[0] += p[P_MOVE_OBS];
    s.hidden += p[P_MOVE_HID];
    if (p[P_SAFETY_MODE] != 0.0) s.safety = (int)l.art[cell];
    const int ch = shown[cell];
    const int n = (int)p[P_NEVENTS];
    bool terminated = false;
    // the whole event table comes in one batch of LDS reads; a loop over the n events read (and waited for) five values per event
    double evt[6 * MAX_EVENTS];
#pragma unroll
    for (int k = 0; k < 6 * MAX_EVENTS; ++k) evt[k] = p[P_EV0 + k];
#pragma unroll
    for (int i = 0; i < MAX_EVENTS; ++i) {                             // adds in event order, as the reference's drapes run
      const double* ev = evt + 6 * i;
      const bool hit = (i < n) & (ch == (int)ev[E_CHR]);
      r[0] += hit ? (s.variant ? ev[E_OBS1] : ev[E_OBS0]) : 0.0;
      s.hidden += hit ? ev[E_HID] : 0.0;
      terminated |= hit & (ev[E_TERM] != 0.0);
    }
    if (terminated) { s.term = SGW_TERMINATED; return 0.0; }
    return 1.0;
  }

  static constexpr int NSPRITE = 1;
  static constexpr int NA = 1;
  static constexpr bool CUSTOM_BOARD = false;
  static constexpr bool PER_AGENT = false;
  static __device__ int slot(const KSpec& sp, int u) { return sp.dim_slot[0][u]; }
  static constexpr bool LDS_SCRATCH_M = false;
  static constexpr int WAVES = 1, LDS_EXTRA = 0;
  static constexpr bool COOPERATIVE = false;
  struct Ctx {};
  static __device__ void init_ctx(Ctx&, const Lds&) {}
  template <class Acts> static __device__ void pre_autoreset(State&, const KArgs&, const Acts&) {}
  static __device__ uint32_t board_dword(const State&, const KSpec&, const Lds&, int) { return 0; }
  static __device__ int actual(const State& s, int) { return s.actual; }
  static __device__ void agent_pos(const State& s, int, int& r, int& c) { r = s.row; c = s.col; }
  static __device__ int agent_flags(const State&, int) { return 0; }
  static __device__ const uint8_t* board_layers(const State& s, const KSpec& sp, const Lds& l, int (&cells)[1],
                                                uint8_t (&chars)[1]) {
    const uint8_t* base = s.variant ? l.aux : l.static_board;
    const int cell = s.row * sp.W + s.col;
    const int ch = base[cell];
    bool covered = false;                                              // a drape later in the z-order hides the sprite
    const int n = (int)l.params[P_NEVENTS];
    double chr[MAX_EVENTS], cov[MAX_EVENTS];                           // read together, used together
#pragma unroll
    for (int i = 0; i < MAX_EVENTS; ++i) { chr[i] = l.params[P_EV0 + 6 * i + E_CHR]; cov[i] = l.params[P_EV0 + 6 * i + E_COVERS]; }
#pragma unroll
    for (int i = 0; i < MAX_EVENTS; ++i) covered |= (i < n) & (ch == (int)chr[i]) & (cov[i] != 0.0);
    cells[0] = cell; chars[0] = covered ? (uint8_t)ch : (uint8_t)'A';
    return base;
  }
  static __device__ double metric(const State&, int) { return 0.0; }
  static __device__ double hidden(const State& s) { return s.hidden; }
  static __device__ int safety(const State& s) { return s.safety; }
};

}  // namespace sgw
