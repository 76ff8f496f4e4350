// TEST INFRASTRUCTURE ONLY (see hip/hip_runtime.h here): drives the per-lane family code one env at a time through the
// same sequence k_engine runs -- load, [pre_autoreset + begin_episode | play + step bookkeeping], outputs, store -- and
// writes what it produced to a file.  tests/test_host_families.py compares it with the reference fixtures.
//
//   host_families <in> <out>
//   in : int32 family, n, T, reset_proto | sgw_spec bytes | int32 ftable_n | ftable doubles | uint64 rng[n][4] | int8 actions[n][T][A]
//        ... then int32 nb | uint8 ep_bits[n][nb] | int32 nr | f64 rand_stream[n][nr]   (0 = none)
//        reset_proto 1: slot 0 = reset, slot 1 = reset, then T ticks (actions[..][0] == -128: explicit reset)   (multi-agent fixtures)
//        reset_proto 0: slot 0 = reset, then T ticks                                                            (scalar fixtures)
//   out: per env and slot: step_type[A] int32 | frame int32 | reward[A*K] f64 | cumulative[A*K] f64 | board[HW] u8
#define __HIPCC__ 1
#define SGW_PLAIN_STORES 1
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../ai_safety_gridworlds_amd/csrc/sgw_island.hpp"
#include "../../ai_safety_gridworlds_amd/csrc/sgw_island_ma.hpp"
#include "../../ai_safety_gridworlds_amd/csrc/sgw_savanna.hpp"
#include "../../ai_safety_gridworlds_amd/csrc/sgw_boat.hpp"
#include "../../ai_safety_gridworlds_amd/csrc/sgw_sokoban.hpp"
#include "../../ai_safety_gridworlds_amd/csrc/sgw_conveyor.hpp"
#include "../../ai_safety_gridworlds_amd/csrc/sgw_rocks.hpp"
#include "../../ai_safety_gridworlds_amd/csrc/sgw_tile.hpp"
#include "../../ai_safety_gridworlds_amd/csrc/sgw_safeint.hpp"
#include "../../ai_safety_gridworlds_amd/csrc/sgw_tomato.hpp"
#include "../../ai_safety_gridworlds_amd/csrc/sgw_friendfoe.hpp"
#include "../../ai_safety_gridworlds_amd/csrc/sgw_whisky.hpp"

using namespace sgw;

#include <type_traits>
template <class F, class = void> struct has_idle : std::false_type {};
template <class F> struct has_idle<F, std::void_t<decltype(&F::idle_round)>> : std::true_type {};
template <class F, class = void> struct fam_ew { static constexpr int value = ENV_WAVES; };
template <class F> struct fam_ew<F, std::void_t<decltype(F::ENV_WAVES_MAX)>> { static constexpr int value = F::ENV_WAVES_MAX; };
template <class F> constexpr int wg_threads_host() { return fam_ew<F>::value * WAVE; }
template <class F, class = void> struct has_args : std::false_type {};
template <class F> struct has_args<F, std::void_t<decltype(&F::init_args)>> : std::true_type {};
template <class F, class = void> struct has_issue : std::false_type {};
template <class F> struct has_issue<F, std::void_t<decltype(&F::init_issue)>> : std::true_type {};
template <class F, class = void> struct has_prep : std::false_type {};
template <class F> struct has_prep<F, std::void_t<typename F::BoardPrep>> : std::true_type {};
template <class F, class = void> struct has_stage : std::false_type {};
template <class F> struct has_stage<F, std::void_t<decltype(&F::stage_board)>> : std::true_type {};

struct Host {
  sgw_spec spec; KArgs a; Lds l;
  std::vector<uint8_t> tables, lds, bits; std::vector<uint64_t> state; std::vector<double> ftable, stream;
  int nb = 0, nr = 0;
  long long n, n_pad;
};

template <class F> static int words_of(const sgw_spec& sp);
template <> int words_of<Island>(const sgw_spec& sp) { return Island::words(sp.K); }
template <> int words_of<IslandGeneral>(const sgw_spec& sp) { return IslandGeneral::words(sp.K); }
template <> int words_of<IslandPacked>(const sgw_spec& sp) { return IslandPacked::words(sp.K); }
template <> int words_of<IslandMa>(const sgw_spec& sp) { return IslandMa::words(sp.K); }
template <> int words_of<IslandMaWide>(const sgw_spec& sp) { return IslandMaWide::words(sp.K); }
template <> int words_of<Savanna>(const sgw_spec& sp) { return Savanna::words(sp.K); }
template <> int words_of<Boat>(const sgw_spec& sp) { return Boat::words(sp.K, sp.H * sp.W); }
template <> int words_of<Sokoban>(const sgw_spec&) { return Sokoban::words(); }
template <> int words_of<Conveyor>(const sgw_spec&) { return Conveyor::words(); }
template <> int words_of<Rocks>(const sgw_spec&) { return Rocks::words(); }
template <> int words_of<Tile>(const sgw_spec&) { return Tile::words(); }
template <> int words_of<SafeInt>(const sgw_spec&) { return SafeInt::words(); }
template <> int words_of<Tomato>(const sgw_spec&) { return Tomato::words(); }
template <> int words_of<FriendFoe>(const sgw_spec&) { return FriendFoe::words(); }
template <> int words_of<Whisky>(const sgw_spec&) { return Whisky::words(); }

template <class F> static void setup(Host& h, const uint64_t* rng) {
  const sgw_spec& sp = h.spec;
  KSpec& k = h.a.sp;
  std::memset(&h.a, 0, sizeof(h.a));
  k.family = sp.family; k.H = sp.H; k.W = sp.W; k.HW = sp.H * sp.W; k.K = sp.K; k.M = sp.M; k.A = sp.A;
  k.max_iterations = sp.max_iterations; k.flags = sp.flags; k.action_lo = sp.action_lo; k.n_actions = sp.n_actions;
  k.words = words_of<F>(sp);
  std::memcpy(k.start_cell, sp.start_cell, sizeof(k.start_cell));
  kspec_derive(k);
  std::memcpy(k.dim_slot, sp.dim_slot, sizeof(k.dim_slot));
  std::memcpy(k.metric_slot, sp.metric_slot, sizeof(k.metric_slot));
  h.tables.assign(TABLE_BYTES, 0);
  std::memcpy(h.tables.data(), sp.static_board, k.HW);
  std::memcpy(h.tables.data() + SGW_MAX_CELLS, sp.art, k.HW);
  std::memcpy(h.tables.data() + 2 * SGW_MAX_CELLS, sp.aux, k.HW);
  std::memcpy(h.tables.data() + 3 * SGW_MAX_CELLS, sp.value_map, 512);
  std::memcpy(h.tables.data() + 3 * SGW_MAX_CELLS + 512, sp.params, SGW_N_PARAMS * 8);
  h.n_pad = (h.n + 63) / 64 * 64;
  h.state.assign((size_t)state_alloc_words(k.words, h.n_pad), 0);
  for (long long e = 0; e < h.n_pad; ++e) h.state[state_index(0, e, k.words)] = ((uint64_t)ST_NONE << 32) | ((uint64_t)15 << 36);      // sgw_create
  if (rng) for (long long e = 0; e < h.n_pad; ++e) {                                                           // k_set_rng
    const uint64_t pad[4] = {0x9E3779B97F4A7C15ull, (uint64_t)e, 0ull, 1ull};
    for (int q = 0; q < 4; ++q) h.state[state_index(3 + q, e, k.words)] = e < h.n ? rng[e * 4 + q] : pad[q];
  }
  h.a.tables = h.tables.data(); h.a.state = h.state.data(); h.a.n_pad = h.n_pad; h.a.n_envs = h.n; h.a.T = 1;
  h.a.ftable = h.ftable.empty() ? nullptr : h.ftable.data();
  if (h.nb) { h.a.ep_bits = h.bits.data(); h.a.ep_bits_n = h.nb; }
  if (h.nr) { h.a.rand_stream = h.stream.data(); h.a.rand_n = h.nr; }
  // LDS image: the level tables, 64 board rows, and the family's extra region (island: the pow tables)
  const size_t extra = F::LDS_EXTRA;
  h.lds.assign(lds_total_bytes(k.HW, k.A, k.K, k.M, 1, 0, 0, (int)extra, 1, 1) + 64, 0);
  std::memcpy(h.lds.data(), h.tables.data(), TABLE_BYTES);
  h.a.lp = lds_plan(k.HW, k.A, k.K, k.M, 1, 0, 0);
  h.l = lds_carve(h.lds.data(), h.a.lp, (int)extra, 0);
  if (extra) { for (int t = 0; t < wg_threads_host<F>(); ++t) { threadIdx.x = t; typename F::Ctx cx; if constexpr (has_issue<F>::value) F::init_issue(cx); F::init_ctx(cx, h.l); if constexpr (has_args<F>::value) F::init_args(cx, h.l, h.a); } }
}

template <class F> static void emit(Host& h, const typename F::State& s, const double (&r)[F::NU], long long env, FILE* out) {
  const KSpec& sp = h.a.sp;
  for (int ag = 0; ag < sp.A; ++ag) {
    int32_t st;
    if constexpr (F::PER_AGENT) st = F::agent_step_type(s, ag); else st = s.step_type;
    fwrite(&st, 4, 1, out);
  }
  int32_t fr = s.frame; fwrite(&fr, 4, 1, out);
  std::vector<double> rew(sp.A * sp.K, 0.0), cum(sp.A * sp.K, 0.0);
  for (int u = 0; u < F::NU; ++u) { const int sl = F::slot(sp, u); if (sl >= 0) { rew[sl] = r[u]; cum[sl] = s.cum[u]; } }
  fwrite(rew.data(), 8, rew.size(), out); fwrite(cum.data(), 8, cum.size(), out);
  std::vector<uint8_t> board((sp.HW + 3) / 4 * 4, 0);
  if constexpr (has_prep<F>::value) {
    const auto bp = F::board_prepare(s, sp);
    for (int i = 0; i < (sp.HW + 3) / 4; ++i) { const uint32_t v = F::board_dword(bp, s, sp, i); std::memcpy(&board[4 * i], &v, 4); }
  } else if constexpr (F::CUSTOM_BOARD) {
    for (int i = 0; i < (sp.HW + 3) / 4; ++i) { const uint32_t v = F::board_dword(s, sp, h.l, i); std::memcpy(&board[4 * i], &v, 4); }
  } else {
    int cells[F::NSPRITE]; uint8_t chars[F::NSPRITE];
    const uint8_t* base = F::board_layers(s, sp, h.l, cells, chars);
    std::memcpy(board.data(), base, sp.HW);
    for (int q = 0; q < F::NSPRITE; ++q) board[cells[q]] = chars[q];
  }
  if constexpr (has_stage<F>::value) {
    // the family's own LDS row writer, at this env's lane of a packed 64-row image: the row equals the dword rendering,
    // and nothing outside the row's first .. last dword is touched
    const int lane = (int)(env & 63), HW = sp.HW, o = lane * HW;
    std::vector<uint32_t> img((64 * HW + 3) / 4 + 4, 0xAAAAAAAAu);
    Lds li = h.l; li.board = img.data();
    F::stage_board(li, s, sp, lane);
    const uint8_t* b = reinterpret_cast<const uint8_t*>(img.data());
    for (int i = 0; i < HW; ++i)
      if (b[o + i] != board[i]) { std::fprintf(stderr, "stage_board: cell %d of lane %d is %d, want %d\n", i, lane, b[o + i], board[i]); std::exit(3); }
    const int first = o / 4 * 4, end = (o + HW + 3) / 4 * 4;
    for (int i = 0; i < (int)img.size() * 4; ++i)
      if ((i < first || i >= end) && b[i] != 0xAA) { std::fprintf(stderr, "stage_board: byte %d outside the row of lane %d written\n", i, lane); std::exit(3); }
    for (int i = first; i < end; ++i)
      if ((i < o || i >= o + HW) && b[i] != 0) { std::fprintf(stderr, "stage_board: a neighbour's byte %d in a shared dword is not zero\n", i); std::exit(3); }
  }
  if constexpr (!has_stage<F>::value) {
    // the generic static-board-plus-sprites row writer (sgw_common.hpp lds_write_board_row) at this env's lane: same checks
    const int lane = (int)(env & 63), HW = sp.HW, o = lane * HW;
    std::vector<uint32_t> img((64 * HW + 3) / 4 + 8, 0xAAAAAAAAu);
    int cells[F::NSPRITE]; uint8_t chars[F::NSPRITE];
    const uint8_t* base = F::board_layers(s, sp, h.l, cells, chars);
    lds_write_board_row<F::NSPRITE>(img.data(), HW, lane, base, cells, chars);
    const uint8_t* b = reinterpret_cast<const uint8_t*>(img.data());
    for (int i = 0; i < HW; ++i)
      if (b[o + i] != board[i]) { std::fprintf(stderr, "lds_write_board_row: cell %d of lane %d is %d, want %d\n", i, lane, b[o + i], board[i]); std::exit(3); }
    const int first = o / 4 * 4, end = (o + HW + 3) / 4 * 4;
    for (int i = 0; i < (int)img.size() * 4; ++i)
      if ((i < first || i >= end) && b[i] != 0xAA) { std::fprintf(stderr, "lds_write_board_row: byte %d outside the row of lane %d written\n", i, lane); std::exit(3); }
  }
  fwrite(board.data(), 1, sp.HW, out);
  (void)env;
}

template <class F> static void reset_env(Host& h, long long env, FILE* out) {
  threadIdx.x = (unsigned)(env & 63);
  typename F::State s; F::load(s, h.a, env);
  double r[F::NU]; for (int u = 0; u < F::NU; ++u) r[u] = 0.0;
  h.a.mode = MODE_RESET;
  F::begin_episode(s, h.a, h.l, env, env);
  F::store(s, h.a, env);
  emit<F>(h, s, r, env, out);
}

template <class F> static void step_env(Host& h, long long env, const int8_t* act, FILE* out) {   // k_engine, non-cooperative branch
  threadIdx.x = (unsigned)(env & 63);
  typename F::State s; F::load(s, h.a, env);
  double r[F::NU]; for (int u = 0; u < F::NU; ++u) r[u] = 0.0;
  int action[F::NA]; for (int ag = 0; ag < F::NA; ++ag) action[ag] = act[ag];
  h.a.mode = MODE_STEP;
  bool idle = false;
  if constexpr (has_idle<F>::value) idle = s.step_type >= ST_LAST && !F::reset_requested(s, h.a, action);
  if (idle) {
    if constexpr (has_idle<F>::value) F::idle_round(s);
  } else if (s.step_type >= ST_LAST) {
    F::pre_autoreset(s, h.a, action);
    F::begin_episode(s, h.a, h.l, env, env);
  } else {
    const double discount = F::play(s, action, h.a, h.l, r, env);
    const bool over = (discount == 0.0) || (s.frame >= h.a.sp.max_iterations);
    s.step_type = over ? ST_LAST : ST_MID;
    if (over && s.term == 15) s.term = SGW_MAX_STEPS;
    for (int u = 0; u < F::NU; ++u) s.cum[u] += r[u];
  }
  F::store(s, h.a, env);
  emit<F>(h, s, r, env, out);
}

template <class F> static int run(Host& h, int T, int proto, const uint64_t* rng, const int8_t* actions, FILE* out) {
  setup<F>(h, rng);
  const int A = F::NA;
  for (long long e = 0; e < h.n; ++e) {
    reset_env<F>(h, e, out);
    if (proto) reset_env<F>(h, e, out);
    for (int t = 0; t < T; ++t) {
      const int8_t* act = actions + ((size_t)e * T + t) * A;
      if (proto && act[0] == -128) reset_env<F>(h, e, out); else step_env<F>(h, e, act, out);
    }
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  FILE* in = fopen(argv[1], "rb"); FILE* out = fopen(argv[2], "wb");
  if (!in || !out) return 2;
  int32_t hd[4]; if (fread(hd, 4, 4, in) != 4) return 2;
  Host h; h.n = hd[1];
  if (fread(&h.spec, sizeof(h.spec), 1, in) != 1) return 2;
  int32_t fn; if (fread(&fn, 4, 1, in) != 1) return 2;
  h.ftable.resize(fn); if (fn && fread(h.ftable.data(), 8, fn, in) != (size_t)fn) return 2;
  std::vector<uint64_t> rng((size_t)h.n * 4); if (fread(rng.data(), 8, rng.size(), in) != rng.size()) return 2;
  const int A = (hd[0] == SGW_ISLAND_NAVIGATION_EX_MA || hd[0] == SGW_AINTELOPE_SAVANNA) ? 2 : 1;
  std::vector<int8_t> actions((size_t)h.n * hd[2] * A); if (fread(actions.data(), 1, actions.size(), in) != actions.size()) return 2;
  int32_t nb = 0, nr = 0;
  if (fread(&nb, 4, 1, in) == 1 && nb > 0) { h.nb = nb; h.bits.resize((size_t)h.n * nb); if (fread(h.bits.data(), 1, h.bits.size(), in) != h.bits.size()) return 2; }
  if (fread(&nr, 4, 1, in) == 1 && nr > 0) { h.nr = nr; h.stream.resize((size_t)h.n * nr); if (fread(h.stream.data(), 8, h.stream.size(), in) != h.stream.size()) return 2; }
  int rc = 3;
  if (hd[0] == SGW_ISLAND_NAVIGATION_EX) rc = (h.spec.flags & Island::F_GENERAL) ? run<IslandGeneral>(h, hd[2], hd[3], nullptr, actions.data(), out)
                                                                                   : (Island::packable(h.spec) ? run<IslandPacked>(h, hd[2], hd[3], nullptr, actions.data(), out)
                                                                                                                : run<Island>(h, hd[2], hd[3], nullptr, actions.data(), out));
  else if (hd[0] == SGW_ISLAND_NAVIGATION_EX_MA) rc = h.spec.H * h.spec.W > 64 ? run<IslandMaWide>(h, hd[2], hd[3], rng.data(), actions.data(), out)      // 8-word maps
                                                                               : run<IslandMa>(h, hd[2], hd[3], rng.data(), actions.data(), out);
  else if (hd[0] == SGW_AINTELOPE_SAVANNA) rc = run<Savanna>(h, hd[2], hd[3], rng.data(), actions.data(), out);
  else if (hd[0] == SGW_BOAT_RACE_EX || hd[0] == SGW_BOAT_RACE) rc = run<Boat>(h, hd[2], hd[3], nullptr, actions.data(), out);
  else if (hd[0] == SGW_SIDE_EFFECTS_SOKOBAN) rc = run<Sokoban>(h, hd[2], hd[3], nullptr, actions.data(), out);
  else if (hd[0] == SGW_CONVEYOR_BELT) rc = run<Conveyor>(h, hd[2], hd[3], nullptr, actions.data(), out);
  else if (hd[0] == SGW_ROCKS_DIAMONDS) rc = run<Rocks>(h, hd[2], hd[3], nullptr, actions.data(), out);
  else if (hd[0] == SGW_TILE_EVENTS) rc = run<Tile>(h, hd[2], hd[3], nullptr, actions.data(), out);
  else if (hd[0] == SGW_SAFE_INTERRUPTIBILITY) rc = run<SafeInt>(h, hd[2], hd[3], nullptr, actions.data(), out);
  else if (hd[0] == SGW_TOMATO_WATERING) rc = run<Tomato>(h, hd[2], hd[3], nullptr, actions.data(), out);
  else if (hd[0] == SGW_FRIEND_FOE) rc = run<FriendFoe>(h, hd[2], hd[3], nullptr, actions.data(), out);
  else if (hd[0] == SGW_WHISKY_GOLD) rc = run<Whisky>(h, hd[2], hd[3], nullptr, actions.data(), out);
  fclose(in); fclose(out);
  return rc;
}
