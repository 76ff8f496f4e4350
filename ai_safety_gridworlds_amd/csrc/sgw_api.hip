// sgw_api.hip -- host side of libsgw.so: the C ABI declared in include/sgw.h.
// Thin by design: validates arguments, owns the engine's device state, launches k_engine<F>.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/sgw.h"
#include "sgw_boat.hpp"
#include "sgw_conveyor.hpp"
#include "sgw_firemaker.hpp"
#include "sgw_friendfoe.hpp"
#include "sgw_island_ma.hpp"
#include "sgw_island.hpp"
#include "sgw_kernels.hpp"
#include "sgw_rocks.hpp"
#include "sgw_safeint.hpp"
#include "sgw_savanna.hpp"
#include "sgw_sokoban.hpp"
#include "sgw_tile.hpp"
#include "sgw_tomato.hpp"
#include "sgw_whisky.hpp"
#include "sgw_group.hpp"
#include "sgw_savanna_layers.hpp"

using namespace sgw;

// Host copy of the pow tables: sgw_create checks the RUNNING libm against them (see pow_selfcheck).
namespace sgw_host_pow {
#undef SGW_POW_TABLE
#define SGW_POW_TABLE static const
#include "sgw_pow_tables.inc"
}

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, const char* detail = "") {
  snprintf(g_err, sizeof(g_err), fmt, detail);
  return code;
}
#define HIP_TRY(expr)                                                                    \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      snprintf(g_err, sizeof(g_err), "%s failed: %s", #expr, hipGetErrorString(e_));     \
      return SGW_ERR_HIP;                                                                \
    }                                                                                    \
  } while (0)

struct sgw_engine {
  sgw_spec spec;
  KSpec ks;
  int device;
  long long n_envs, n_pad, env_id_base;
  uint8_t* tables_dev;     // static_board | art | aux | value_map
  uint64_t* state_dev;     // pair layout [env-wave][word pair][lane][2] (sgw_common.hpp state_index); canonical [words][n_pad] only through sgw_get/set_state
  const uint8_t* ep_bits;
  int ep_bits_n;
  unsigned long long ep_seed;
  const double* rand_stream;
  int rand_n;
  unsigned long long rand_seed;
  double* acc_dev;         // [n_pad/64 * SGW_ACC_PARTS][A*K+1] episodic-return accumulators, one row per 16 envs (lazily allocated)
  int rng_set;
  double* ftable_dev;      // sgw_set_family_table
  long long ftable_n;
  // sgw_step_n: the T launches of one (actions, T, outputs) call, captured once and replayed as a hipGraph (see step_n_graph)
  struct StepGraph { const int8_t* actions; int T, write_every, accumulate, has_out; long long last_use; sgw_out out; hipGraphExec_t exec; };
  long long graph_tick;
  std::vector<StepGraph> graphs;
  hipStream_t capture_stream;
  unsigned lds_cap_raised; // bit KIND: this engine's k_engine<F, KIND> may use more than 64 KiB of dynamic LDS (set once)
  // sgw_step_full: one captured graph per (actions, out, extras) triple
  struct FullGraph { const int8_t* actions; sgw_out out; sgw_extras ex; long long last_use; hipGraphExec_t exec; };
  std::vector<FullGraph> full_graphs;
};

static void drop_graphs(sgw_engine* e) {      // any setter that changes what a launch's arguments hold invalidates the captures
  for (auto& g : e->graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
  e->graphs.clear();
  for (auto& g : e->full_graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
  e->full_graphs.clear();
}

static size_t acc_bytes(const sgw_engine* e) {              // [env-waves * parts][A*K+1] doubles
  return (size_t)(e->spec.A * e->spec.K + 1) * (size_t)(SGW_ACC_PER_ENV ? e->n_pad : e->n_pad / WAVE * SGW_ACC_PARTS) * 8;
}

// island_navigation_ex: the packed (i16) state when the spec proves it exact (sgw_island.hpp); SGW_ISLAND_PLAIN_STATE in the
// environment forces the plain f64 state (tests run every fixture through both).
static bool island_packable(const sgw_spec& sp) { return !getenv("SGW_ISLAND_PLAIN_STATE") && Island::packable(sp); }

static int family_words(const sgw_spec& sp) {
  switch (sp.family) {
    case SGW_ISLAND_NAVIGATION_EX: return island_packable(sp) ? IslandPacked::words(sp.K) : Island::words(sp.K);
    case SGW_BOAT_RACE_EX:
    case SGW_BOAT_RACE: return Boat::words(sp.K, sp.H * sp.W);
    case SGW_SAFE_INTERRUPTIBILITY: return SafeInt::words();
    case SGW_FIREMAKER_EX_MA: return Firemaker::words();
    case SGW_ISLAND_NAVIGATION_EX_MA: return sp.H * sp.W > 64 ? IslandMaWide::words(sp.K) : IslandMa::words(sp.K);
    case SGW_TILE_EVENTS: return Tile::words();
    case SGW_SIDE_EFFECTS_SOKOBAN: return Sokoban::words();
    case SGW_CONVEYOR_BELT: return Conveyor::words();
    case SGW_TOMATO_WATERING: return Tomato::words();
    case SGW_FRIEND_FOE: return FriendFoe::words();
    case SGW_WHISKY_GOLD: return Whisky::words();
    case SGW_ROCKS_DIAMONDS: return Rocks::words();
    case SGW_AINTELOPE_SAVANNA: return Savanna::words(sp.K);
    default: return -1;
  }
}

// The device pow restates glibc's algorithm over tables extracted from the BUILD host's libm (csrc/sgw_pow_tables.inc);
// the reference's math.pow is the RUNNING host's libm pow.  When the two differ (another glibc, a non-FMA dispatch) parity
// of resource regrowth would fail later and quietly, so the families that regrow resources refuse to construct instead.
// Probe: the regrowth domain (halves and pseudo-random x in [1, 61]) at three exponents.  Returns the mismatch count.
static long pow_selfcheck_run() {
  long bad = 0;
  unsigned long long s = 88172645463325252ULL;
  const double ys[3] = {1.1, 1.05, 1.5};
  for (int i = 0; i < 4096; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    double x = 1.0 + 60.0 * ((double)(s >> 11) / 9007199254740992.0);
    if (i < 240) x = 1.0 + 0.25 * i;
    const double y = ys[i % 3];
    const double want = pow(x, y);
    const double got = sgw_glibc_pow_core(x, y, sgw_host_pow::SGW_POW_LOG_HEAD, sgw_host_pow::SGW_POW_EXP_HEAD,
                                          sgw_host_pow::SGW_POW_LOG_TAB, sgw_host_pow::SGW_POW_EXP_TAB);
    if (memcmp(&want, &got, 8) != 0) ++bad;
  }
  return bad;
}
static long pow_selfcheck() {
  static const long cached = pow_selfcheck_run();           // C++11: initialised once, thread-safe
  return cached;
}

extern "C" {

int sgw_abi_version(void) { return SGW_ABI_VERSION; }
const char* sgw_last_error(void) { return g_err; }
#ifndef SGW_BUILD_COMPILER
#define SGW_BUILD_COMPILER "unknown (built outside ai_safety_gridworlds_amd/build.py: not linted)"
#endif
const char* sgw_build_info(void) { return SGW_BUILD_COMPILER; }
int sgw_sizeof_spec(void) { return (int)sizeof(sgw_spec); }
int sgw_sizeof_out(void) { return (int)sizeof(sgw_out); }
int sgw_sizeof_extras(void) { return (int)sizeof(sgw_extras); }
int64_t sgw_pow_selfcheck(void) { return (int64_t)pow_selfcheck(); }

int sgw_create(const sgw_spec* spec, int64_t n_envs, int64_t env_id_base, int device, sgw_engine** out_engine) {
  if (!spec || !out_engine) return fail(SGW_ERR_ARG, "sgw_create: null argument");
  *out_engine = nullptr;
  if (n_envs <= 0) return fail(SGW_ERR_ARG, "sgw_create: n_envs must be positive");
  const int HW = spec->H * spec->W;
  if (spec->H <= 0 || spec->W <= 0 || HW > SGW_MAX_CELLS || spec->H > 255 || spec->W > 255)
    return fail(SGW_ERR_ARG, "sgw_create: board size out of range");
  if (spec->K < 1 || spec->K > SGW_MAX_K || spec->M < 0 || spec->M > SGW_MAX_M || spec->A < 1 ||
      spec->A > SGW_MAX_AGENTS)
    return fail(SGW_ERR_ARG, "sgw_create: K/M/A out of range");
  if (spec->max_iterations < 1 || spec->max_iterations > 65535)
    return fail(SGW_ERR_ARG, "sgw_create: max_iterations must be in [1, 65535]");
  for (int ag = 0; ag < spec->A; ++ag)
    if (spec->start_cell[ag] < 0 || spec->start_cell[ag] >= HW)
      return fail(SGW_ERR_ARG, "sgw_create: start_cell outside the board");
  for (int ag = 0; ag < SGW_MAX_AGENTS; ++ag)
    for (int u = 0; u < SGW_MAX_K; ++u)
      if (spec->dim_slot[ag][u] >= spec->A * spec->K) return fail(SGW_ERR_ARG, "sgw_create: dim_slot >= A*K");
  for (int m = 0; m < SGW_MAX_M; ++m)
    if (spec->metric_slot[m] >= spec->M && spec->metric_slot[m] >= 0)
      return fail(SGW_ERR_ARG, "sgw_create: metric_slot >= M");
  if (spec->family == SGW_AINTELOPE_SAVANNA && (HW > 192 || spec->A != 2))
    return fail(SGW_ERR_ARG, "sgw_create: aintelope_savanna holds a layer in 3 x 64 bits (H*W <= 192) and lays out two agents (A = 2)");
  if (spec->family == SGW_ISLAND_NAVIGATION_EX_MA && (HW > 128 || spec->A != 2))
    return fail(SGW_ERR_ARG, "sgw_create: island_navigation_ex_ma keeps its map in 4 or 8 x 16 nibbles (H*W <= 128) and has two agents");
  const int words = family_words(*spec);
  if (words < 0) return fail(SGW_ERR_UNSUPPORTED, "sgw_create: unknown game family");
  if ((spec->family == SGW_ISLAND_NAVIGATION_EX || spec->family == SGW_ISLAND_NAVIGATION_EX_MA ||
       spec->family == SGW_AINTELOPE_SAVANNA) && pow_selfcheck() != 0 && !getenv("SGW_ALLOW_LIBM_MISMATCH"))
    return fail(SGW_ERR_UNSUPPORTED, "sgw_create: this host's libm pow() differs from the tables the device pow was built from "
                "(csrc/sgw_pow_tables.inc, glibc 2.35 x86-64 FMA build): resource regrowth would not match the reference's "
                "math.pow here.  Regenerate the tables on this host (tools/gen_pow_tables.py) and rebuild, or set "
                "SGW_ALLOW_LIBM_MISMATCH=1");

  HIP_TRY(hipSetDevice(device));
  sgw_engine* e = new (std::nothrow) sgw_engine();
  if (!e) return fail(SGW_ERR_NOMEM, "sgw_create: out of host memory");
  e->spec = *spec;
  e->device = device;
  e->n_envs = n_envs;
  e->n_pad = (n_envs + SGW_ENV_ALIGN - 1) / SGW_ENV_ALIGN * SGW_ENV_ALIGN;
  e->env_id_base = env_id_base;
  e->ep_bits = nullptr; e->ep_bits_n = 0; e->ep_seed = 0; e->rand_stream = nullptr; e->rand_n = 0; e->rand_seed = 0; e->acc_dev = nullptr; e->rng_set = 0; e->ftable_dev = nullptr; e->ftable_n = 0; e->capture_stream = nullptr; e->graph_tick = 0; e->lds_cap_raised = 0;

  KSpec& k = e->ks;
  memset(&k, 0, sizeof(k));
  k.family = spec->family; k.H = spec->H; k.W = spec->W; k.HW = HW; k.K = spec->K; k.M = spec->M;
  k.A = spec->A; k.max_iterations = spec->max_iterations; k.flags = spec->flags;
  if (spec->family == SGW_ISLAND_NAVIGATION_EX) k.flags = (k.flags & ~Island::F_PACKED) | (island_packable(*spec) ? Island::F_PACKED : 0);
  k.action_lo = spec->action_lo; k.n_actions = spec->n_actions; k.words = words;
  memcpy(k.start_cell, spec->start_cell, sizeof(k.start_cell));
  kspec_derive(k);
  kspec_views(k, *spec);
  if (spec->family == SGW_ISLAND_NAVIGATION_EX_MA) k.view_rotates = (spec->flags & (IslandMa::F_ODIR | IslandMa::F_ODIR_TURN)) ? 1 : 0;
  if (spec->family == SGW_AINTELOPE_SAVANNA) k.view_rotates = (spec->flags & (Savanna::F_ODIR | Savanna::F_ODIR_TURN)) ? 1 : 0;
  if (spec->family == SGW_FIREMAKER_EX_MA) k.view_rotates = (spec->flags & (Firemaker::F_ODIR | Firemaker::F_ODIR_TURN)) ? 1 : 0;
  memcpy(k.dim_slot, spec->dim_slot, sizeof(k.dim_slot));
  memcpy(k.metric_slot, spec->metric_slot, sizeof(k.metric_slot));

  const size_t tbytes = TABLE_BYTES;
  uint8_t host_tables[TABLE_BYTES];
  memset(host_tables, 0, sizeof(host_tables));
  memcpy(host_tables, spec->static_board, HW);
  memcpy(host_tables + SGW_MAX_CELLS, spec->art, HW);
  memcpy(host_tables + 2 * SGW_MAX_CELLS, spec->aux, HW);
  memcpy(host_tables + 3 * SGW_MAX_CELLS, spec->value_map, 512);
  memcpy(host_tables + 3 * SGW_MAX_CELLS + 512, spec->params, SGW_N_PARAMS * 8);

  hipError_t err = hipMalloc((void**)&e->tables_dev, tbytes);
  if (err == hipSuccess) err = hipMemcpy(e->tables_dev, host_tables, tbytes, hipMemcpyHostToDevice);
  const size_t sbytes = (size_t)state_alloc_words(words, e->n_pad) * 8;      // pair layout (sgw_common.hpp): ceil(words / 2) KiB per env-wave
  if (err == hipSuccess) err = hipMalloc((void**)&e->state_dev, sbytes);
  // the episodic-return accumulators (sgw_read_returns): part of the engine from the start, zeroed here with the state --
  // sgw_create ends with a device synchronisation, so no stream of the caller can run ahead of the zeroing
  if (err == hipSuccess) err = hipMalloc((void**)&e->acc_dev, acc_bytes(e));
  if (err == hipSuccess) err = hipMemset(e->acc_dev, 0, acc_bytes(e));
  if (err != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "sgw_create: device allocation failed: %s", hipGetErrorString(err));
    if (e->tables_dev) (void)hipFree(e->tables_dev);
    if (e->state_dev) (void)hipFree(e->state_dev);
    if (e->acc_dev) (void)hipFree(e->acc_dev);
    delete e;
    return SGW_ERR_HIP;
  }
  *out_engine = e;
  // step_type ST_NONE (3) in every env's core word => the first sgw_step auto-resets
  hipLaunchKernelGGL(k_state_init, dim3((unsigned)((sbytes / 8 + 255) / 256)), dim3(256), 0, 0, e->state_dev, e->n_pad, words);
  err = hipGetLastError();
  if (err != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "sgw_create: state init failed: %s", hipGetErrorString(err));
    sgw_destroy(e); *out_engine = nullptr;
    return SGW_ERR_HIP;
  }
  // the state / table initialisation above went through the null stream: finished before the caller can launch on any
  // (possibly non-blocking) stream of its own
  err = hipStreamSynchronize(nullptr);
  if (err != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "sgw_create: initialisation did not complete: %s", hipGetErrorString(err));
    sgw_destroy(e); *out_engine = nullptr;
    return SGW_ERR_HIP;
  }
  return SGW_OK;
}

int sgw_destroy(sgw_engine* e) {
  if (!e) return SGW_OK;
  drop_graphs(e);
  if (e->capture_stream) (void)hipStreamDestroy(e->capture_stream);
  if (e->tables_dev) (void)hipFree(e->tables_dev);
  if (e->state_dev) (void)hipFree(e->state_dev);
  if (e->acc_dev) (void)hipFree(e->acc_dev);
  if (e->ftable_dev) (void)hipFree(e->ftable_dev);
  delete e;
  return SGW_OK;
}

int64_t sgw_n_envs(const sgw_engine* e) { return e ? e->n_envs : 0; }
int64_t sgw_n_pad(const sgw_engine* e) { return e ? e->n_pad : 0; }
int sgw_state_words(const sgw_engine* e) { return e ? e->ks.words : 0; }
int64_t sgw_state_bytes(const sgw_engine* e) { return e ? (int64_t)e->ks.words * e->n_pad * 8 : 0; }

int sgw_set_episode_bits(sgw_engine* e, const uint8_t* bits_dev, int n_per_env, uint64_t seed) {
  if (!e) return fail(SGW_ERR_ARG, "sgw_set_episode_bits: null engine");
  if (bits_dev && n_per_env <= 0) return fail(SGW_ERR_ARG, "sgw_set_episode_bits: n_per_env must be positive");
  e->ep_bits = bits_dev; e->ep_bits_n = bits_dev ? n_per_env : 0; e->ep_seed = seed;
  drop_graphs(e);
  return SGW_OK;
}

int sgw_set_random_stream(sgw_engine* e, const double* u_dev, int n_per_env, uint64_t seed) {
  if (!e) return fail(SGW_ERR_ARG, "sgw_set_random_stream: null engine");
  if (u_dev && n_per_env <= 0) return fail(SGW_ERR_ARG, "sgw_set_random_stream: n_per_env must be positive");
  e->rand_stream = u_dev; e->rand_n = u_dev ? n_per_env : 0; e->rand_seed = seed;
  drop_graphs(e);
  return SGW_OK;
}

int sgw_set_rng_state(sgw_engine* e, const uint64_t* pcg_state_dev) {
  if (!e || !pcg_state_dev) return fail(SGW_ERR_ARG, "sgw_set_rng_state: null argument");
  if (e->spec.family != SGW_FIREMAKER_EX_MA && e->spec.family != SGW_ISLAND_NAVIGATION_EX_MA &&
      e->spec.family != SGW_AINTELOPE_SAVANNA)
    return fail(SGW_ERR_UNSUPPORTED, "sgw_set_rng_state: this game family has no env-side RNG stream");
  HIP_TRY(hipSetDevice(e->device));
  hipLaunchKernelGGL(k_set_rng, dim3((unsigned)((e->n_pad + 255) / 256)), dim3(256), 0, 0, e->state_dev, e->n_pad,
                     e->n_envs, e->ks.words, pcg_state_dev);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(0));
  e->rng_set = 1;
  return SGW_OK;
}

int sgw_set_family_table(sgw_engine* e, const double* table_host, int64_t n) {
  if (!e || !table_host || n <= 0) return fail(SGW_ERR_ARG, "sgw_set_family_table: null / empty argument");
  const bool island_general = e->spec.family == SGW_ISLAND_NAVIGATION_EX && (e->spec.flags & Island::F_GENERAL);
  if (e->spec.family != SGW_AINTELOPE_SAVANNA && !island_general)
    return fail(SGW_ERR_UNSUPPORTED, "sgw_set_family_table: this game family / configuration has no lookup table");
  if (e->spec.family == SGW_AINTELOPE_SAVANNA && n != 2 * ((int64_t)e->spec.max_iterations + 2))
    return fail(SGW_ERR_ARG, "sgw_set_family_table: aintelope_savanna expects 2 * (max_iterations + 2) entries");
  if (island_general && n != 15 * 12 + 15)
    return fail(SGW_ERR_ARG, "sgw_set_family_table: island_navigation_ex per-event reward vectors are 15 x 12 values + 15 masks");
  HIP_TRY(hipSetDevice(e->device));
  drop_graphs(e);
  if (e->ftable_dev) { (void)hipFree(e->ftable_dev); e->ftable_dev = nullptr; e->ftable_n = 0; }
  HIP_TRY(hipMalloc((void**)&e->ftable_dev, (size_t)n * 8));
  HIP_TRY(hipMemcpy(e->ftable_dev, table_host, (size_t)n * 8, hipMemcpyHostToDevice));
  e->ftable_n = n;
  return SGW_OK;
}

#ifdef SGW_SAV_PROF     // diagnostic build only
extern "C" int sgw_debug_sav_prof(unsigned long long* out, int clear) {      // out[4096 * 16]
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sav_prof), 4096 * 16 * 8) != hipSuccess) return -1;
  if (clear) { void* p = nullptr; if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_sav_prof)) != hipSuccess || hipMemset(p, 0, 4096 * 16 * 8) != hipSuccess) return -1; }
  return 0;
}
#endif
#ifdef SGW_PHASE_PROF   // diagnostic build only
extern "C" int sgw_debug_phase_prof(unsigned long long* out, int clear) {    // out[4096 * 8]
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_prof), 4096 * 8 * 8) != hipSuccess) return -1;
  if (clear) { void* p = nullptr; if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_phase_prof)) != hipSuccess || hipMemset(p, 0, 4096 * 8 * 8) != hipSuccess) return -1; }
  return 0;
}
#endif
#ifdef SGW_FM_PROF      // diagnostic build only
extern "C" int sgw_debug_fm_prof(unsigned long long* out, int clear) {      // out[4096 * 16]
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fm_prof), 4096 * 16 * 8) != hipSuccess) return -1;
  if (clear) { void* p = nullptr; if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_fm_prof)) != hipSuccess || hipMemset(p, 0, 4096 * 16 * 8) != hipSuccess) return -1; }
  return 0;
}
#endif

int sgw_pow_f64(const double* x_dev, double y, double* out_dev, int64_t n, int device, void* stream) {
  if (!x_dev || !out_dev || n < 0) return fail(SGW_ERR_ARG, "sgw_pow_f64: null / negative argument");
  if (n == 0) return SGW_OK;
  HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(k_pow, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x_dev, y, out_dev, (long long)n);
  HIP_TRY(hipGetLastError());
  return SGW_OK;
}

// (the accumulators are allocated and zeroed by sgw_create: nothing is allocated on the step path)
static int ensure_acc(sgw_engine* e, hipStream_t) {
  return e->acc_dev ? SGW_OK : fail(SGW_ERR_ARG, "the engine's return accumulators are missing");
}

}  // extern "C"

// ---- launch planning: the same validation and LDS / grid arithmetic for a family's own launch and for a group member ------
template <class F> struct FamilyType { using type = F; };
// fn(FamilyType<F>{}, tag) for the family (and state variant) this engine runs
template <class Fn> static int with_family(const sgw_engine* e, Fn&& fn) {
  switch (e->spec.family) {
    case SGW_ISLAND_NAVIGATION_EX:
      if (e->spec.flags & Island::F_GENERAL) return fn(FamilyType<IslandGeneral>{}, (int)TAG_ISLAND_GENERAL);
      if (e->ks.flags & Island::F_PACKED) return fn(FamilyType<IslandPacked>{}, (int)TAG_ISLAND_PACKED);
      return fn(FamilyType<Island>{}, (int)TAG_ISLAND);
    case SGW_BOAT_RACE_EX:
    case SGW_BOAT_RACE: return fn(FamilyType<Boat>{}, (int)TAG_BOAT);
    case SGW_SAFE_INTERRUPTIBILITY: return fn(FamilyType<SafeInt>{}, (int)TAG_SAFEINT);
    case SGW_FIREMAKER_EX_MA:
      if (e->spec.flags & Firemaker::F_WIDE) return fn(FamilyType<FiremakerWide>{}, (int)TAG_FIREMAKER);     // spread distance > 3
      return fn(FamilyType<Firemaker>{}, (int)TAG_FIREMAKER);
    case SGW_ISLAND_NAVIGATION_EX_MA:
      if (e->spec.H * e->spec.W > 64) return fn(FamilyType<IslandMaWide>{}, (int)TAG_ISLAND_MA);      // resized maps of 65..128 cells
      return fn(FamilyType<IslandMa>{}, (int)TAG_ISLAND_MA);
    case SGW_TILE_EVENTS: return fn(FamilyType<Tile>{}, (int)TAG_TILE);
    case SGW_SIDE_EFFECTS_SOKOBAN: return fn(FamilyType<Sokoban>{}, (int)TAG_SOKOBAN);
    case SGW_CONVEYOR_BELT: return fn(FamilyType<Conveyor>{}, (int)TAG_CONVEYOR);
    case SGW_TOMATO_WATERING: return fn(FamilyType<Tomato>{}, (int)TAG_TOMATO);
    case SGW_FRIEND_FOE: return fn(FamilyType<FriendFoe>{}, (int)TAG_FRIEND_FOE);
    case SGW_WHISKY_GOLD: return fn(FamilyType<Whisky>{}, (int)TAG_WHISKY);
    case SGW_ROCKS_DIAMONDS: return fn(FamilyType<Rocks>{}, (int)TAG_ROCKS);
    case SGW_AINTELOPE_SAVANNA: return fn(FamilyType<Savanna>{}, (int)TAG_SAVANNA);
    default: return fail(SGW_ERR_UNSUPPORTED, "launch: unknown game family");
  }
}

static int launch_kind(const KArgs& a) {                      // K_STEP reads the caller's actions only
  return a.mode == MODE_RESET ? K_RESET : ((a.T == 1 && a.actions) ? K_STEP : K_ROLLOUT);
}

// what has to be true of the engine before any launch, and the engine's own fields of the arguments
static int prepare_args(sgw_engine* e, KArgs& a) {
  if (e->spec.family == SGW_FIREMAKER_EX_MA && !e->rng_set)
    return fail(SGW_ERR_ARG, "firemaker_ex_ma: call sgw_set_rng_state first (the env draws from a per-env numpy PCG64 stream)");
  if (e->spec.family == SGW_ISLAND_NAVIGATION_EX_MA && !e->rng_set &&
      (e->spec.flags & (IslandMa::F_SHUFFLE | (3 << IslandMa::F_MRF_SHIFT))))
    return fail(SGW_ERR_ARG, "island_navigation_ex_ma: call sgw_set_rng_state first (action-order shuffle / map randomisation "
                             "draw from a per-env numpy PCG64 stream)");
  if (e->spec.family == SGW_AINTELOPE_SAVANNA && (!e->rng_set || !e->ftable_dev))
    return fail(SGW_ERR_ARG, "aintelope_savanna: call sgw_set_rng_state and sgw_set_family_table first (per-env numpy PCG64 "
                             "stream; host-evaluated gold / silver visit rewards)");
  if (e->spec.family == SGW_ISLAND_NAVIGATION_EX && (e->spec.flags & Island::F_GENERAL) && !e->ftable_dev)
    return fail(SGW_ERR_ARG, "island_navigation_ex: spec.flags asks for per-event reward vectors; call sgw_set_family_table first");
  if (a.out.safety2 && e->spec.family != SGW_AINTELOPE_SAVANNA)
    return fail(SGW_ERR_UNSUPPORTED, "launch: the safety2 output exists for aintelope_savanna only");
  if ((a.out.obs_dir || a.out.act_dir) && e->spec.family != SGW_ISLAND_NAVIGATION_EX_MA && e->spec.family != SGW_AINTELOPE_SAVANNA &&
      e->spec.family != SGW_FIREMAKER_EX_MA)
    return fail(SGW_ERR_UNSUPPORTED, "launch: the obs_dir / act_dir outputs exist for the families whose agents carry directions "
                                     "(island_navigation_ex_ma, aintelope_savanna, firemaker_ex_ma)");
  a.sp = e->ks; a.tables = e->tables_dev; a.state = e->state_dev; a.ftable = e->ftable_dev;
  a.n_pad = e->n_pad; a.n_envs = e->n_envs; a.env_id_base = e->env_id_base;
  a.ep_bits = e->ep_bits; a.ep_bits_n = e->ep_bits_n; a.ep_seed = e->ep_seed;
  a.rand_stream = e->rand_stream; a.rand_n = e->rand_n; a.rand_seed = e->rand_seed;
  return SGW_OK;
}

struct LaunchPlan { size_t lds_bytes; unsigned blocks, threads; };
// LDS plan + grid of k_engine<F, KIND> for these arguments (fills a.lp / a.need)
template <class F, int KIND> static int plan_launch(const sgw_engine* e, KArgs& a, LaunchPlan& p) {
  constexpr int EW = env_waves<F, KIND>(), NB = lds_buffers<F, KIND>();
  const long long n_waves = e->n_pad / WAVE;
  const int need = lds_need(a, F::LDS_SCRATCH_M), pa = F::PER_AGENT ? F::NA : 1;
  if ((need & (LN_VIEWS | LN_OBSVIEWS)) && (!has_views<F>::value || a.sp.view_total <= 0))
    return fail(SGW_ERR_UNSUPPORTED, "launch: the views / obs_views outputs exist for the families with agent windows only");
  if ((need & (LN_VIEWS | LN_OBSVIEWS)) && F::WAVES == 1 && a.sp.view_prefill)
    return fail(SGW_ERR_UNSUPPORTED, "launch: a window larger than the board is not assembled inside this family's round kernel (one "
                                     "wavefront per 64 envs would walk them env by env): use sgw_agent_views");
  const int vb = a.sp.view_total > 0 ? a.sp.view_total : 0;
  const int CS = cum_stash_rows<F>(a.sp.A, a.sp.K);
  a.lp = lds_plan(a.sp.HW, a.sp.A, a.sp.K, a.sp.M, pa, need, vb, CS); a.need = need;
  p.lds_bytes = lds_total_bytes(a.sp.HW, a.sp.A, a.sp.K, a.sp.M, pa, need, vb, F::LDS_EXTRA, EW, NB, CS);
  p.blocks = (unsigned)((n_waves + EW - 1) / EW);
  p.threads = (unsigned)wg_threads<F, KIND>();
  // the bytes requested == the bytes the plan hands out (checked on every launch)
  if ((size_t)TABLE_BYTES + F::LDS_EXTRA + (size_t)EW * NB * a.lp.wave_bytes != p.lds_bytes || (F::LDS_EXTRA & 15) != 0 ||
      (a.lp.wave_bytes & 15) != 0 || (a.lp.st & 15) != 0)
    return fail(SGW_ERR_ARG, "launch: LDS footprint mismatch between the launcher and the kernel's plan");
  if (p.lds_bytes > 160 * 1024) return fail(SGW_ERR_UNSUPPORTED, "launch: the requested outputs need more than 160 KiB of LDS per workgroup");
  return SGW_OK;
}

template <class F, int KIND> static int launch_as(sgw_engine* e, KArgs& a, hipStream_t st) {
  LaunchPlan p;
  int rc = plan_launch<F, KIND>(e, a, p);
  if (rc) return rc;
  // above the default dynamic-LDS cap: raised ONCE per engine and kernel kind, to the CU's 160 KiB -- a launch that needs it
  // is then never the first inside a stream capture (sgw_step_n)
  if (p.lds_bytes > 65536 && !(e->lds_cap_raised & (1u << KIND))) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_engine<F, KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    e->lds_cap_raised |= 1u << KIND;
  }
  hipLaunchKernelGGL((k_engine<F, KIND>), dim3(p.blocks), dim3(p.threads), p.lds_bytes, st, SGW_HOT_ARGS(a), a);
  return SGW_OK;
}

static int launch(sgw_engine* e, KArgs& a, hipStream_t st) {
  int rc = prepare_args(e, a);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(e->device));
  const int kind = launch_kind(a);
  rc = with_family(e, [&](auto ft, int) {
    using F = typename decltype(ft)::type;
    if (kind == K_STEP) return launch_as<F, K_STEP>(e, a, st);
    if (kind == K_ROLLOUT) return launch_as<F, K_ROLLOUT>(e, a, st);
    return launch_as<F, K_RESET>(e, a, st);
  });
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  return SGW_OK;
}

extern "C" {

int sgw_reset(sgw_engine* e, const uint8_t* mask_dev, const sgw_out* out, void* stream) {
  if (!e) return fail(SGW_ERR_ARG, "sgw_reset: null engine");
  KArgs a; memset(&a, 0, sizeof(a));
  a.mode = MODE_RESET; a.mask = mask_dev; a.T = 1;
  if (out) a.out = *out;
  return launch(e, a, (hipStream_t)stream);
}

int sgw_step(sgw_engine* e, const int8_t* actions_dev, const sgw_out* out, void* stream) {
  if (!e) return fail(SGW_ERR_ARG, "sgw_step: null engine");
  if (!actions_dev) return fail(SGW_ERR_ARG, "sgw_step: null actions");
  KArgs a; memset(&a, 0, sizeof(a));
  a.mode = MODE_STEP; a.actions = actions_dev; a.T = 1;
  if (out) a.out = *out;
  return launch(e, a, (hipStream_t)stream);
}

static long long kspec_view_total(const sgw_spec& sp) {
  long long off = 0;
  for (int a = 0; a < sp.A; ++a) if (sp.view_radius[a][0] >= 0) off += (long long)(sp.view_radius[a][0] + sp.view_radius[a][1] + 1) * (sp.view_radius[a][2] + sp.view_radius[a][3] + 1);
  return off;
}
static void offset_out(sgw_out& o, const sgw_spec& sp, long long n_pad, long long t) {
  const long long HW = sp.H * sp.W, AK = sp.A * sp.K, A = sp.A, M = sp.M, r = t * n_pad;
  const long long PA = (sp.family == SGW_ISLAND_NAVIGATION_EX_MA || sp.family == SGW_AINTELOPE_SAVANNA) ? A : 1;   // term_reason / safety are [N_pad, A] there (IslandMa::PER_AGENT)
  if (o.board) o.board += r * HW;
  if (o.obs_board) o.obs_board += r * HW;
  if (o.reward) o.reward += r * AK;
  if (o.cumulative) o.cumulative += r * AK;
  if (o.step_type) o.step_type += r * A;
  if (o.term_reason) o.term_reason += r * PA;
  if (o.actual_action) o.actual_action += r * A;
  if (o.discount) o.discount += r;
  if (o.hidden) o.hidden += r;
  if (o.safety) o.safety += r * PA;
  if (o.safety2) o.safety2 += r * PA;
  if (o.metrics) o.metrics += r * M;
  if (o.frame) o.frame += r;
  if (o.agent_pos) o.agent_pos += r * A * 2;
  if (o.agent_flags) o.agent_flags += r * A;
  if (o.done) o.done += r * A;
  if (o.obs_dir) o.obs_dir += r * A;
  if (o.act_dir) o.act_dir += r * A;
  const long long VB = kspec_view_total(sp);
  if (o.views) o.views += r * VB;
  if (o.obs_views) o.obs_views += r * VB;
}

static int step_n_launches(sgw_engine* e, const int8_t* actions_dev, int T, int write_every, const sgw_out* out,
                           int accumulate, hipStream_t st) {
  for (int t = 0; t < T; ++t) {
    KArgs a; memset(&a, 0, sizeof(a));
    a.mode = MODE_STEP; a.T = 1; a.ep_acc = accumulate ? e->acc_dev : nullptr;
    a.actions = actions_dev + (long long)t * e->n_envs * e->spec.A;
    if (out) { a.out = *out; if (write_every) offset_out(a.out, e->spec, e->n_pad, t); }
    int rc = launch(e, a, st);
    if (rc) return rc;
  }
  return SGW_OK;
}

// One host launch costs ~5 us of CPU on this runtime -- as much as the smaller families' whole step kernel, and three times
// that for a mixed suite issued from one thread.  A caller that steps from the same device buffers again (the usual loop:
// one actions buffer refilled in place) gets its T launches captured into a hipGraph the second time the same arguments are
// seen and replayed from then on.  SGW_STEP_GRAPHS=0 turns this off.
static int step_graphs_min_T() {
  static int v = -1;
  if (v < 0) { const char* s = getenv("SGW_STEP_GRAPHS"); v = (s && atoi(s) == 0) ? (1 << 30) : 8; }
  return v;
}

int sgw_step_n(sgw_engine* e, const int8_t* actions_dev, int T, int write_every, const sgw_out* out,
               int accumulate, void* stream) {
  if (!e) return fail(SGW_ERR_ARG, "sgw_step_n: null engine");
  if (!actions_dev || T < 1) return fail(SGW_ERR_ARG, "sgw_step_n: bad argument");
  const hipStream_t st = (hipStream_t)stream;
  if (accumulate) { int rc = ensure_acc(e, st); if (rc) return rc; }
  if (T < step_graphs_min_T()) return step_n_launches(e, actions_dev, T, write_every, out, accumulate, st);
  sgw_out key_out; memset(&key_out, 0, sizeof(key_out));
  if (out) key_out = *out;
  sgw_engine::StepGraph* hit = nullptr;
  for (auto& g : e->graphs)
    if (g.actions == actions_dev && g.T == T && g.write_every == (write_every != 0) && g.accumulate == (accumulate != 0) &&
        g.has_out == (out != nullptr) && memcmp(&g.out, &key_out, sizeof(key_out)) == 0) { hit = &g; break; }
  if (!hit) {                                   // first sighting: remember the arguments, launch directly
    // bounded (256 captures, 64 Ki kernel nodes in all): drop the least recently used captures
    auto nodes = [&]() { long long n = T; for (auto& g : e->graphs) n += g.T; return n; };
    while (!e->graphs.empty() && (e->graphs.size() >= 256 || nodes() > 65536)) {
      size_t worst = 0;
      for (size_t i = 1; i < e->graphs.size(); ++i) if (e->graphs[i].last_use < e->graphs[worst].last_use) worst = i;
      if (e->graphs[worst].exec) (void)hipGraphExecDestroy(e->graphs[worst].exec);
      e->graphs.erase(e->graphs.begin() + worst);
    }
    sgw_engine::StepGraph g; memset(&g, 0, sizeof(g));
    g.actions = actions_dev; g.T = T; g.write_every = write_every != 0; g.accumulate = accumulate != 0; g.has_out = out != nullptr;
    g.out = key_out; g.exec = nullptr; g.last_use = ++e->graph_tick;
    e->graphs.push_back(g);
    return step_n_launches(e, actions_dev, T, write_every, out, accumulate, st);
  }
  hit->last_use = ++e->graph_tick;
  if (!hit->exec) {                             // second sighting: capture (on the engine's own stream; nothing executes) and instantiate
    HIP_TRY(hipSetDevice(e->device));
    if (!e->capture_stream) HIP_TRY(hipStreamCreateWithFlags(&e->capture_stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamBeginCapture(e->capture_stream, hipStreamCaptureModeThreadLocal));
    const int rc = step_n_launches(e, actions_dev, T, write_every, out, accumulate, e->capture_stream);
    hipGraph_t graph = nullptr;
    const hipError_t ec = hipStreamEndCapture(e->capture_stream, &graph);
    if (rc || ec != hipSuccess || !graph) {
      if (graph) (void)hipGraphDestroy(graph);
      if (rc) return rc;
      return fail(SGW_ERR_HIP, "sgw_step_n: stream capture of the step launches failed");
    }
    const hipError_t ei = hipGraphInstantiate(&hit->exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ei != hipSuccess) { hit->exec = nullptr; return fail(SGW_ERR_HIP, "sgw_step_n: hipGraphInstantiate failed"); }
  }
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipGraphLaunch(hit->exec, st));
  return SGW_OK;
}

}  // extern "C"

// ---- groups: one heterogeneous launch over several engines (k_engine_group, sgw_group.hpp) -----------------------------------
struct sgw_group {
  std::vector<sgw_engine*> members;
  int device;
  unsigned lds_cap_raised;
  struct Graph { std::vector<const int8_t*> actions; std::vector<sgw_out> outs; int T, write_every, accumulate, has_out; long long last_use; hipGraphExec_t exec; };
  std::vector<Graph> graphs;
  long long graph_tick;
  hipStream_t capture_stream;
};

static bool group_tag(int tag) {
  switch (tag) {
#define SGW_GROUP_TAG(TAG, F) case TAG: return true;
    SGW_GROUP_FAMILIES(SGW_GROUP_TAG)
#undef SGW_GROUP_TAG
    default: return false;
  }
}

// one launch of k_engine_group<KIND> over the members' prepared arguments (a[m].actions / out / T / seed ... set by the caller)
template <int KIND> static int group_launch(sgw_group* g, std::vector<KArgs>& a, hipStream_t st) {
  GroupArgs ga; memset(&ga, 0, sizeof(ga));
  ga.n = (int)g->members.size();
  size_t lds = 0; unsigned blocks = 0;
  for (size_t m = 0; m < g->members.size(); ++m) {
    sgw_engine* e = g->members[m];
    int rc = prepare_args(e, a[m]);
    if (rc) return rc;
    LaunchPlan p{};
    int tag_m = -1;
    rc = with_family(e, [&](auto ft, int tag) {
      using F = typename decltype(ft)::type;
      tag_m = tag;
      if constexpr (group_member<F, K_STEP>() && group_member<F, K_ROLLOUT>()) return plan_launch<F, KIND>(e, a[m], p);
      else return fail(SGW_ERR_UNSUPPORTED, "group launch: this env family is not a group member (its round kernel has its own workgroup shape)");
    });
    if (rc) return rc;
    if (!group_tag(tag_m)) return fail(SGW_ERR_UNSUPPORTED, "group launch: this env family / configuration is not a group member");
    ga.first_block[m] = (int)blocks; ga.tag[m] = tag_m; ga.a[m] = a[m];
    blocks += p.blocks;
    lds = p.lds_bytes > lds ? p.lds_bytes : lds;
  }
  for (size_t m = g->members.size(); m <= GROUP_MAX; ++m) ga.first_block[m] = (int)blocks;
  if (lds > 65536 && !(g->lds_cap_raised & (1u << KIND))) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_engine_group<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    g->lds_cap_raised |= 1u << KIND;
  }
  GroupHot h{ga.n, ga.first_block[1], ga.first_block[2], ga.first_block[3],
             ga.tag[0] | (ga.tag[1] << 8) | (ga.tag[2] << 16) | (ga.tag[3] << 24)};
  hipLaunchKernelGGL((k_engine_group<KIND>), dim3(blocks), dim3(GROUP_THREADS), lds, st, SGW_GROUP_HOT_ARGS(h), ga);
  HIP_TRY(hipGetLastError());
  return SGW_OK;
}

static int group_step_launches(sgw_group* g, const int8_t* const* actions, int T, int write_every, const sgw_out* outs, int accumulate,
                               hipStream_t st) {
  const size_t n = g->members.size();
  std::vector<KArgs> a(n);
  for (int t = 0; t < T; ++t) {
    for (size_t m = 0; m < n; ++m) {
      sgw_engine* e = g->members[m];
      memset(&a[m], 0, sizeof(KArgs));
      a[m].mode = MODE_STEP; a[m].T = 1; a[m].ep_acc = accumulate ? e->acc_dev : nullptr;
      a[m].actions = actions[m] + (long long)t * e->n_envs * e->spec.A;
      if (outs) { a[m].out = outs[m]; if (write_every) offset_out(a[m].out, e->spec, e->n_pad, t); }
    }
    int rc = group_launch<K_STEP>(g, a, st);
    if (rc) return rc;
  }
  return SGW_OK;
}

extern "C" {

int sgw_group_create(sgw_engine* const* engines, int n, sgw_group** out_group) {
  if (!engines || !out_group || n < 1 || n > GROUP_MAX) return fail(SGW_ERR_ARG, "sgw_group_create: 1 to 4 engines");
  *out_group = nullptr;
  for (int m = 0; m < n; ++m) {
    if (!engines[m]) return fail(SGW_ERR_ARG, "sgw_group_create: null engine");
    if (engines[m]->device != engines[0]->device) return fail(SGW_ERR_ARG, "sgw_group_create: the engines of a group live on one device");
    for (int k = 0; k < m; ++k) if (engines[k] == engines[m]) return fail(SGW_ERR_ARG, "sgw_group_create: an engine is listed twice");
    int tag_m = -1;
    (void)with_family(engines[m], [&](auto, int tag) { tag_m = tag; return 0; });
    if (!group_tag(tag_m)) return fail(SGW_ERR_UNSUPPORTED, "sgw_group_create: this env family / configuration is not a group member");
  }
  sgw_group* g = new (std::nothrow) sgw_group();
  if (!g) return fail(SGW_ERR_NOMEM, "sgw_group_create: out of host memory");
  g->members.assign(engines, engines + n);
  g->device = engines[0]->device; g->lds_cap_raised = 0; g->graph_tick = 0; g->capture_stream = nullptr;
  *out_group = g;
  return SGW_OK;
}

int sgw_group_destroy(sgw_group* g) {
  if (!g) return SGW_OK;
  for (auto& gr : g->graphs) if (gr.exec) (void)hipGraphExecDestroy(gr.exec);
  if (g->capture_stream) (void)hipStreamDestroy(g->capture_stream);
  delete g;
  return SGW_OK;
}

int sgw_group_step_n(sgw_group* g, const int8_t* const* actions_dev, int T, int write_every, const sgw_out* outs, int accumulate,
                     void* stream) {
  if (!g || !actions_dev || T < 1) return fail(SGW_ERR_ARG, "sgw_group_step_n: bad argument");
  const hipStream_t st = (hipStream_t)stream;
  const size_t n = g->members.size();
  for (size_t m = 0; m < n; ++m) {
    if (!actions_dev[m]) return fail(SGW_ERR_ARG, "sgw_group_step_n: null actions");
    if (accumulate) { int rc = ensure_acc(g->members[m], st); if (rc) return rc; }
  }
  HIP_TRY(hipSetDevice(g->device));
  if (T < step_graphs_min_T()) return group_step_launches(g, actions_dev, T, write_every, outs, accumulate, st);
  sgw_group::Graph* hit = nullptr;
  for (auto& gr : g->graphs) {
    if (gr.T != T || gr.write_every != (write_every != 0) || gr.accumulate != (accumulate != 0) || gr.has_out != (outs != nullptr)) continue;
    bool same = true;
    for (size_t m = 0; m < n && same; ++m)
      same = gr.actions[m] == actions_dev[m] && (!outs || memcmp(&gr.outs[m], &outs[m], sizeof(sgw_out)) == 0);
    if (same) { hit = &gr; break; }
  }
  if (!hit) {                                   // first sighting: remember, launch directly (as sgw_step_n does)
    while (g->graphs.size() >= 64) {
      size_t worst = 0;
      for (size_t i = 1; i < g->graphs.size(); ++i) if (g->graphs[i].last_use < g->graphs[worst].last_use) worst = i;
      if (g->graphs[worst].exec) (void)hipGraphExecDestroy(g->graphs[worst].exec);
      g->graphs.erase(g->graphs.begin() + worst);
    }
    sgw_group::Graph gr;
    gr.actions.assign(actions_dev, actions_dev + n);
    if (outs) gr.outs.assign(outs, outs + n);
    gr.T = T; gr.write_every = write_every != 0; gr.accumulate = accumulate != 0; gr.has_out = outs != nullptr;
    gr.last_use = ++g->graph_tick; gr.exec = nullptr;
    g->graphs.push_back(gr);
    return group_step_launches(g, actions_dev, T, write_every, outs, accumulate, st);
  }
  hit->last_use = ++g->graph_tick;
  if (!hit->exec) {
    if (!g->capture_stream) HIP_TRY(hipStreamCreateWithFlags(&g->capture_stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamBeginCapture(g->capture_stream, hipStreamCaptureModeThreadLocal));
    const int rc = group_step_launches(g, actions_dev, T, write_every, outs, accumulate, g->capture_stream);
    hipGraph_t graph = nullptr;
    const hipError_t ec = hipStreamEndCapture(g->capture_stream, &graph);
    if (rc || ec != hipSuccess || !graph) {
      if (graph) (void)hipGraphDestroy(graph);
      if (rc) return rc;
      return fail(SGW_ERR_HIP, "sgw_group_step_n: stream capture of the step launches failed");
    }
    const hipError_t ei = hipGraphInstantiate(&hit->exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ei != hipSuccess) { hit->exec = nullptr; return fail(SGW_ERR_HIP, "sgw_group_step_n: hipGraphInstantiate failed"); }
  }
  HIP_TRY(hipGraphLaunch(hit->exec, st));
  return SGW_OK;
}

int sgw_group_rollout(sgw_group* g, int T, uint64_t seed, int64_t step0, int write_every, const sgw_out* outs, int accumulate,
                      void* stream) {
  if (!g || T < 1) return fail(SGW_ERR_ARG, "sgw_group_rollout: bad argument");
  const hipStream_t st = (hipStream_t)stream;
  const size_t n = g->members.size();
  std::vector<KArgs> a(n);
  for (size_t m = 0; m < n; ++m) {
    sgw_engine* e = g->members[m];
    if (e->spec.n_actions < 1) return fail(SGW_ERR_ARG, "sgw_group_rollout: spec has no action range");
    if (accumulate) { int rc = ensure_acc(e, st); if (rc) return rc; }
    memset(&a[m], 0, sizeof(KArgs));
    a[m].mode = MODE_STEP; a[m].actions = nullptr; a[m].T = T; a[m].seed = seed; a[m].step0 = step0;
    a[m].write_every = write_every; a[m].ep_acc = accumulate ? e->acc_dev : nullptr;
    if (outs) a[m].out = outs[m];
  }
  HIP_TRY(hipSetDevice(g->device));
  return group_launch<K_ROLLOUT>(g, a, st);
}

int sgw_read_returns(sgw_engine* e, double* out_dev, int clear, void* stream) {
  if (!e || !out_dev) return fail(SGW_ERR_ARG, "sgw_read_returns: null argument");
  int rc = ensure_acc(e, (hipStream_t)stream);
  if (rc) return rc;
  hipLaunchKernelGGL(k_read_returns, dim3(e->spec.A * e->spec.K + 1), dim3(256), 0, (hipStream_t)stream, e->acc_dev,
                     SGW_ACC_PER_ENV ? e->n_pad : e->n_pad / WAVE * SGW_ACC_PARTS, e->spec.A * e->spec.K + 1, out_dev, clear);
  HIP_TRY(hipGetLastError());
  return SGW_OK;
}

int sgw_rollout(sgw_engine* e, int T, uint64_t seed, int64_t step0, int write_every, const sgw_out* out,
                int accumulate, void* stream) {
  if (!e) return fail(SGW_ERR_ARG, "sgw_rollout: null engine");
  if (T < 1) return fail(SGW_ERR_ARG, "sgw_rollout: T must be >= 1");
  if (e->spec.n_actions < 1) return fail(SGW_ERR_ARG, "sgw_rollout: spec has no action range");
  KArgs a; memset(&a, 0, sizeof(a));
  a.mode = MODE_STEP; a.actions = nullptr; a.T = T; a.seed = seed; a.step0 = step0;
  if (accumulate) { int rc = ensure_acc(e, (hipStream_t)stream); if (rc) return rc; }
  a.write_every = write_every; a.ep_acc = accumulate ? e->acc_dev : nullptr;
  if (out) a.out = *out;
  return launch(e, a, (hipStream_t)stream);
}

int sgw_replay(sgw_engine* e, const int8_t* actions_dev, int T, int write_every, const sgw_out* out, int accumulate, void* stream) {
  if (!e) return fail(SGW_ERR_ARG, "sgw_replay: null engine");
  if (!actions_dev || T < 1) return fail(SGW_ERR_ARG, "sgw_replay: bad argument");
  KArgs a; memset(&a, 0, sizeof(a));
  a.mode = MODE_STEP; a.actions = actions_dev; a.T = T;
  if (accumulate) { int rc = ensure_acc(e, (hipStream_t)stream); if (rc) return rc; }
  a.write_every = write_every; a.ep_acc = accumulate ? e->acc_dev : nullptr;
  if (out) a.out = *out;
  return launch(e, a, (hipStream_t)stream);
}

int sgw_fill_actions(sgw_engine* e, int T, uint64_t seed, int64_t step0, int8_t* actions_dev, void* stream) {
  if (!e || !actions_dev || T < 1) return fail(SGW_ERR_ARG, "sgw_fill_actions: bad argument");
  HIP_TRY(hipSetDevice(e->device));
  long long total = (long long)T * e->n_envs * e->spec.A;
  int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(k_fill_actions, dim3(blocks), dim3(256), 0, (hipStream_t)stream, actions_dev, e->n_envs,
                     e->spec.A, T, seed, step0, e->env_id_base, e->spec.action_lo, e->spec.n_actions);
  HIP_TRY(hipGetLastError());
  return SGW_OK;
}

int sgw_accumulate_returns(sgw_engine* e, const double* cumulative_dev, const uint8_t* step_type_dev,
                           double* ep_accum_dev, void* stream) {
  if (!e || !cumulative_dev || !step_type_dev || !ep_accum_dev)
    return fail(SGW_ERR_ARG, "sgw_accumulate_returns: null argument");
  HIP_TRY(hipSetDevice(e->device));
  int blocks = (int)((e->n_envs + 255) / 256);
  hipLaunchKernelGGL(k_accumulate_returns, dim3(blocks), dim3(256), 0, (hipStream_t)stream, cumulative_dev,
                     step_type_dev, e->n_envs, e->spec.A * e->spec.K, ep_accum_dev);
  HIP_TRY(hipGetLastError());
  return SGW_OK;
}

// ceil(2^32 / x) for div_recip (sgw_kernels.hpp): exact for every f <= f_max, or 0xffffffff... refused by the caller
static bool plane_geom(int HW, int P, PlaneGeom& g) {
  auto recip = [](uint64_t x, uint64_t f_max, uint32_t& out) {
    if (x <= 1) { out = 0; return true; }
    const uint64_t r = ((1ull << 32) + x - 1) / x, err = r * x - (1ull << 32);
    out = (uint32_t)r;
    return f_max * err < (1ull << 32);
  };
  g.HW = HW; g.P = P;
  uint32_t a = 0, b = 0;
  const uint64_t total = (uint64_t)PLANES_ENVS * P * HW;
  uint32_t c = 0;
  const bool ok = recip((uint64_t)P * HW, total, a) && recip((uint64_t)HW, total, b) && recip((uint64_t)(HW + 3) / 4, total, c);
  g.recip_PHW = (int)a; g.recip_HW = (int)b; g.recip_Q = (int)c;
  return ok;
}
static int raise_lds_cap(sgw_engine* e, const void* fn, size_t lds, unsigned bit) {
  if (lds > 160 * 1024) return fail(SGW_ERR_UNSUPPORTED, "observe: the board is too large for the plane kernels' LDS staging");
  if (lds > 65536 && !(e->lds_cap_raised & bit)) {
    HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    e->lds_cap_raised |= bit;
  }
  return SGW_OK;
}

int sgw_observe(sgw_engine* e, const uint8_t* board_dev, const uint8_t* rgb_lut_dev, uint8_t* rgb_dev,
                const uint8_t* layer_chars_dev, int n_layers, uint8_t* layers_dev, void* stream) {
  if (!e || !board_dev) return fail(SGW_ERR_ARG, "sgw_observe: null argument");
  if (rgb_dev && !rgb_lut_dev) return fail(SGW_ERR_ARG, "sgw_observe: rgb requested without a LUT");
  if (layers_dev && (!layer_chars_dev || n_layers < 1 || n_layers > 128)) return fail(SGW_ERR_ARG, "sgw_observe: bad layer list");
  if (!rgb_dev && !layers_dev) return SGW_OK;
  HIP_TRY(hipSetDevice(e->device));
  PlaneGeom g_rgb, g_lay;
  if (!plane_geom(e->ks.HW, 3, g_rgb) || !plane_geom(e->ks.HW, layers_dev ? n_layers : 1, g_lay))
    return fail(SGW_ERR_UNSUPPORTED, "sgw_observe: board / layer count outside the plane kernels' index arithmetic");
  const int epb = PLANES_ENVS;
  const size_t lds = (((size_t)epb * e->ks.HW + 15) & ~(size_t)15) + 3 * 128;
  int rc = raise_lds_cap(e, reinterpret_cast<const void*>(&k_observe), lds, 16u);
  if (rc) return rc;
  const unsigned blocks = (unsigned)((e->n_envs + epb - 1) / epb);
  hipLaunchKernelGGL(k_observe, dim3(blocks), dim3(PLANES_THREADS), lds, (hipStream_t)stream, board_dev, (long long)e->n_envs, epb, g_rgb,
                     rgb_lut_dev, rgb_dev, g_lay, layer_chars_dev, layers_dev);
  HIP_TRY(hipGetLastError());
  return SGW_OK;
}

int sgw_derived_stats(sgw_engine* e, const double* reward_dev, const double* cumulative_dev, const int32_t* frame_dev,
                      const int32_t* k_agent, double* stats_dev, void* stream) {
  if (!e || !reward_dev || !cumulative_dev || !frame_dev || !k_agent || !stats_dev)
    return fail(SGW_ERR_ARG, "sgw_derived_stats: null argument");
  if (((uintptr_t)reward_dev | (uintptr_t)cumulative_dev | (uintptr_t)stats_dev) & 15)
    return fail(SGW_ERR_ARG, "sgw_derived_stats: reward / cumulative / stats must be 16-byte aligned (rows move as 16-byte accesses)");
  AgentK ak; memset(&ak, 0, sizeof(ak));
  for (int a = 0; a < e->spec.A; ++a) {
    if (k_agent[a] < 0 || k_agent[a] > e->spec.K) return fail(SGW_ERR_ARG, "sgw_derived_stats: k_agent out of range");
    ak.k[a] = k_agent[a];
  }
  HIP_TRY(hipSetDevice(e->device));
  const int A = e->spec.A, K = e->spec.K;
  const size_t lds = (size_t)(2 * A * K + A * (5 + K)) * WAVE * 8;           // R | C | O (sgw_kernels.hpp k_derived_stats)
  if (lds > 65536 && !(e->lds_cap_raised & 8u)) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_derived_stats), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    e->lds_cap_raised |= 8u;
  }
  hipLaunchKernelGGL(k_derived_stats, dim3((unsigned)((e->n_envs + WAVE - 1) / WAVE)), dim3(WAVE), lds, (hipStream_t)stream, reward_dev,
                     cumulative_dev, frame_dev, e->n_envs, A, K, ak, ((1 << 18) + A * K - 1) / (A * K), stats_dev);
  HIP_TRY(hipGetLastError());
  return SGW_OK;
}

int sgw_observe_layers(sgw_engine* e, const uint8_t* board_dev, const uint8_t* layer_chars_dev,
                       const uint8_t* layer_static_dev, int n_layers, int gap_index, const uint8_t* agent_pos_dev,
                       const uint8_t* agent_flags_dev, int hidden_layer, uint8_t* layers_dev, void* stream) {
  if ((agent_pos_dev == nullptr) != (agent_flags_dev == nullptr) || hidden_layer >= n_layers)
    return fail(SGW_ERR_ARG, "sgw_observe_layers: agent_pos/agent_flags/hidden_layer mismatch");
  if (!e || !board_dev || !layer_chars_dev || !layer_static_dev || !layers_dev || n_layers < 1 || gap_index >= n_layers)
    return fail(SGW_ERR_ARG, "sgw_observe_layers: bad argument");
  if (n_layers > 32) return fail(SGW_ERR_UNSUPPORTED, "sgw_observe_layers: at most 32 layers (a cell's layers are one 32-bit mask)");
  HIP_TRY(hipSetDevice(e->device));
  PlaneGeom g;
  if (!plane_geom(e->ks.HW, n_layers, g))
    return fail(SGW_ERR_UNSUPPORTED, "sgw_observe_layers: board / layer count outside the plane kernels' index arithmetic");
  const size_t HW = (size_t)e->ks.HW;
  // 64 envs per workgroup, 16 when their boards + masks would take more than a quarter of the CU's LDS (firemaker's 17 x 17: 95 KB
  // for 64 envs = one workgroup per CU and a serial chain of barriers; 24 KB for 16)
  const int epb = (5 * PLANES_ENVS * HW > 40 * 1024) ? 16 : PLANES_ENVS;
  const size_t lds = ((epb * HW + 15) & ~(size_t)15) + epb * HW * 4 + HW * 8 + 512;   // boards | masks | dyn, on | per-char
  int rc = raise_lds_cap(e, reinterpret_cast<const void*>(&k_observe_layers), lds, 32u);
  if (rc) return rc;
  const unsigned blocks = (unsigned)((e->n_envs + epb - 1) / epb);
  hipLaunchKernelGGL(k_observe_layers, dim3(blocks), dim3(PLANES_THREADS), lds, (hipStream_t)stream, board_dev, (long long)e->n_envs, epb, g,
                     e->ks.W, layer_chars_dev, layer_static_dev, gap_index, agent_pos_dev, agent_flags_dev, e->spec.A,
                     agent_pos_dev ? hidden_layer : -1, layers_dev);
  HIP_TRY(hipGetLastError());
  return SGW_OK;
}

int sgw_state_layers(sgw_engine* e, const uint8_t* layer_chars_dev, int n_layers, int gap_only_blank, uint8_t* layers_dev,
                     void* stream) {
  if (!e || !layer_chars_dev || !layers_dev || n_layers < 1) return fail(SGW_ERR_ARG, "sgw_state_layers: bad argument");
  if (e->spec.family != SGW_AINTELOPE_SAVANNA)
    return fail(SGW_ERR_UNSUPPORTED, "sgw_state_layers: this family's layers follow from its board (sgw_observe_layers)");
  if (n_layers > 32) return fail(SGW_ERR_UNSUPPORTED, "sgw_state_layers: at most 32 layers");
  HIP_TRY(hipSetDevice(e->device));
  PlaneGeom g;
  if (!plane_geom(e->ks.HW, n_layers, g)) return fail(SGW_ERR_UNSUPPORTED, "sgw_state_layers: board / layer count outside the plane kernels' index arithmetic");
  const int epb = 16;                                             // 16 envs: 3.5 KB of state words + 10.6 KB of code vectors per workgroup
  const size_t lds = (size_t)epb * SAV_LAYER_WORDS * 8 + (size_t)epb * e->ks.HW * 4 + 32 * 4;
  hipLaunchKernelGGL(k_savanna_layers, dim3((unsigned)((e->n_envs + epb - 1) / epb)), dim3(PLANES_THREADS), lds, (hipStream_t)stream, e->state_dev,
                     e->n_pad, (long long)e->n_envs, e->ks.words, g, e->ks.W, (e->spec.flags & Savanna::F_TWO) ? 1 : 0, layer_chars_dev, gap_only_blank,
                     epb, layers_dev);
  HIP_TRY(hipGetLastError());
  return SGW_OK;
}

static ViewSpec make_viewspec(const sgw_engine* e) {
  ViewSpec v; memset(&v, 0, sizeof(v));
  v.A = e->spec.A; v.H = e->spec.H; v.W = e->spec.W;
  int off = 0;
  for (int a = 0; a < e->spec.A; ++a) {
    const int32_t* rad = e->spec.view_radius[a];
    if (rad[0] < 0) { v.off[a] = off; continue; }
    v.off[a] = off; v.up[a] = rad[0]; v.left[a] = rad[2];
    v.vh[a] = rad[0] + rad[1] + 1; v.vw[a] = rad[2] + rad[3] + 1;
    off += v.vh[a] * v.vw[a];
  }
  v.total = off;
  return v;
}

int sgw_view_bytes(const sgw_engine* e) { return e ? make_viewspec(e).total : 0; }

// LDS stage of one wave of the window kernels: the largest window + one 16-byte phase, rounded to 16
static int view_stage_bytes(const ViewSpec& v) {
  int mx = 0;
  for (int a = 0; a < v.A; ++a) mx = v.vh[a] * v.vw[a] > mx ? v.vh[a] * v.vw[a] : mx;
  return (mx + 16 + 15) / 16 * 16;
}

static bool views_all_small(const ViewSpec& v) {
  for (int a = 0; a < v.A; ++a) if (v.vh[a] * v.vw[a] > WAVE) return false;
  return true;
}

static int check_rotation(const sgw_engine* e, const ViewSpec& v, const uint8_t* flags) {
  if (!flags) return SGW_OK;
  for (int a = 0; a < v.A; ++a)
    if (v.vh[a] != v.vw[a]) return fail(SGW_ERR_UNSUPPORTED, "agent views: rotation by observation direction needs square windows");
  (void)e;
  return SGW_OK;
}

int sgw_agent_views(sgw_engine* e, const uint8_t* board_dev, const uint8_t* agent_pos_dev, const uint8_t* agent_flags_dev,
                    uint8_t outside_chr, uint8_t* views_dev, void* stream) {
  if (!e || !board_dev || !agent_pos_dev || !views_dev) return fail(SGW_ERR_ARG, "sgw_agent_views: null argument");
  ViewSpec v = make_viewspec(e);
  if (v.total <= 0) return fail(SGW_ERR_UNSUPPORTED, "sgw_agent_views: the spec defines no agent views");
  if (check_rotation(e, v, agent_flags_dev)) return SGW_ERR_UNSUPPORTED;
  HIP_TRY(hipSetDevice(e->device));
  if (views_all_small(v)) {                                              // every window <= 64 cells: one thread per byte
    long long total = e->n_envs * (long long)v.total;
    int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(k_agent_views_small, dim3(blocks), dim3(256), 0, (hipStream_t)stream, board_dev, agent_pos_dev,
                       agent_flags_dev, e->n_envs, v, outside_chr, views_dev);
    HIP_TRY(hipGetLastError());
    return SGW_OK;
  }
  {
    // one wave per env: the board and the env's row of windows in LDS, the row out as dword stores at its byte address (the layer-cube
    // kernel with one plane; round 2's k_agent_views -- a wave per window, cells loaded from global memory -- took 26 us for 16 384 envs)
    const int lay_bytes = (e->ks.HW + 15) / 16 * 16, img_bytes = (v.total + 15) / 16 * 16 + 16;      // (+ 16: the image sits at its row's 16-byte phase)
    constexpr int G = 4;      // envs a one-wave workgroup takes per pass (k_agent_layer_views_lds; 16 384 envs: G = 2 18.9, G = 4 17.8, G = 8 24.8 us)
    const size_t lds = 128 + 16 * G + (size_t)G * ((size_t)lay_bytes + (size_t)img_bytes);       // table | per-env words | G x (planes | image)
    if (lds <= 64 * 1024) {
      const long long groups = (e->n_envs + G - 1) / G;
      const int blocks = (int)(groups < 8192 ? groups : 8192);           // (each walks its groups; 4 096..16 384 workgroups measured the same, fewer are slower)
      hipLaunchKernelGGL(k_agent_layer_views_lds<G>, dim3(blocks), dim3(WAVE), lds, (hipStream_t)stream, board_dev, agent_pos_dev, agent_flags_dev,
                         (long long)e->n_envs, v, (const uint8_t*)nullptr, 1, outside_chr, views_dev, lay_bytes, img_bytes, 1);
      HIP_TRY(hipGetLastError());
      return SGW_OK;
    }
  }
  long long total = e->n_envs * (long long)v.A * WAVE;                   // one wave per (env, agent) window
  int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  const int lds_per_wave = view_stage_bytes(v);
  hipLaunchKernelGGL(k_agent_views, dim3(blocks), dim3(256), 4 * lds_per_wave, (hipStream_t)stream, board_dev, agent_pos_dev,
                     agent_flags_dev, e->n_envs, v, outside_chr, views_dev, lds_per_wave);
  HIP_TRY(hipGetLastError());
  return SGW_OK;
}

int sgw_agent_layer_views(sgw_engine* e, const uint8_t* layers_dev, const uint8_t* agent_pos_dev, const uint8_t* agent_flags_dev,
                          const uint8_t* layer_chars_dev, int n_layers, uint8_t outside_chr, uint8_t* out_dev, void* stream) {
  if (!e || !layers_dev || !agent_pos_dev || !layer_chars_dev || !out_dev || n_layers < 1)
    return fail(SGW_ERR_ARG, "sgw_agent_layer_views: bad argument");
  ViewSpec v = make_viewspec(e);
  if (v.total <= 0) return fail(SGW_ERR_UNSUPPORTED, "sgw_agent_layer_views: the spec defines no agent views");
  if (check_rotation(e, v, agent_flags_dev)) return SGW_ERR_UNSUPPORTED;
  HIP_TRY(hipSetDevice(e->device));
  if (views_all_small(v)) {
    long long total = e->n_envs * (long long)v.total * n_layers;
    int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(k_agent_layer_views_small, dim3(blocks), dim3(256), 0, (hipStream_t)stream, layers_dev, agent_pos_dev,
                       agent_flags_dev, e->n_envs, v, layer_chars_dev, n_layers, outside_chr, out_dev);
    HIP_TRY(hipGetLastError());
    return SGW_OK;
  }
  // a workgroup per env: the env's layer planes and its whole output row in LDS (k_agent_layer_views_lds)
  const int lay_bytes = (n_layers * e->ks.HW + 15) / 16 * 16, img_bytes = (v.total * n_layers + 15) / 16 * 16 + 16;
  const size_t lds = 128 + 16 + (size_t)lay_bytes + (size_t)img_bytes;
  if (lds <= 64 * 1024) {
    const int blocks = (int)(e->n_envs < 4096 ? e->n_envs : 4096);
    hipLaunchKernelGGL(k_agent_layer_views_lds<1>, dim3(blocks), dim3(256), lds, (hipStream_t)stream, layers_dev, agent_pos_dev, agent_flags_dev,
                       (long long)e->n_envs, v, layer_chars_dev, n_layers, outside_chr, out_dev, lay_bytes, img_bytes, 0);
    HIP_TRY(hipGetLastError());
    return SGW_OK;
  }
  long long total = e->n_envs * (long long)v.A * n_layers * WAVE;        // (rows too large for LDS) one wave per (env, agent, layer) window
  int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  const int lds_per_wave = view_stage_bytes(v);
  hipLaunchKernelGGL(k_agent_layer_views, dim3(blocks), dim3(256), 4 * lds_per_wave, (hipStream_t)stream, layers_dev, agent_pos_dev,
                     agent_flags_dev, e->n_envs, v, layer_chars_dev, n_layers, outside_chr, out_dev, lds_per_wave);
  HIP_TRY(hipGetLastError());
  return SGW_OK;
}

int sgw_track_performance(sgw_engine* e, const double* perf_dev, int n_cols, const uint8_t* step_type_dev, double* last_dev,
                          double* sum_dev, int64_t* count_dev, uint8_t* done_dev, void* stream) {
  if (!e || !perf_dev || !step_type_dev || n_cols < 1) return fail(SGW_ERR_ARG, "sgw_track_performance: bad argument");
  HIP_TRY(hipSetDevice(e->device));
  const long long total = e->n_envs * n_cols;
  const int per_agent = (e->spec.family == SGW_ISLAND_NAVIGATION_EX_MA || e->spec.family == SGW_AINTELOPE_SAVANNA) ? 1 : 0;
  hipLaunchKernelGGL(k_track_performance, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, perf_dev, n_cols,
                     step_type_dev, e->spec.A, per_agent, (long long)e->n_envs, last_dev, sum_dev, reinterpret_cast<long long*>(count_dev), done_dev);
  HIP_TRY(hipGetLastError());
  return SGW_OK;
}

// RGB + unoccluded layers + performance bookkeeping of the step that just ran, in ONE launch (k_step_extras)
static int step_extras_launch(sgw_engine* e, const sgw_out* out, const sgw_extras* x, bool layers_here, void* stream) {
  const bool perf = x->perf_last || x->perf_sum || x->perf_count || x->done;
  if (!x->rgb && !layers_here && !perf) return SGW_OK;
  if (layers_here && x->n_layers > 32) return fail(SGW_ERR_UNSUPPORTED, "sgw_step_full: at most 32 layers (a cell's layers are one 32-bit mask)");
  ExtrasArgs ea; memset(&ea, 0, sizeof(ea));
  const size_t HW = (size_t)e->ks.HW;
  const int A = e->spec.A, K = e->spec.K;
  const bool planes = x->rgb || layers_here;
  const int epb = (planes && 5 * PLANES_ENVS * HW > 40 * 1024) ? 16 : PLANES_ENVS;
  ea.board = out->board; ea.n = e->n_envs; ea.epb = epb; ea.HW = (int)HW; ea.W = e->ks.W; ea.A = A; ea.K = K;
  if (!plane_geom((int)HW, 3, ea.g_rgb) || !plane_geom((int)HW, layers_here ? x->n_layers : 1, ea.g_lay))
    return fail(SGW_ERR_UNSUPPORTED, "sgw_step_full: board / layer count outside the plane kernels' index arithmetic");
  ea.rgb_lut = x->rgb_lut_dev; ea.rgb = x->rgb;
  if (layers_here) {
    ea.chars = x->layer_chars_dev; ea.stat = x->layer_static_dev; ea.gap = x->gap_index; ea.hidden = x->hidden_layer >= 0 ? x->hidden_layer : -1;
    ea.pos = x->hidden_layer >= 0 ? out->agent_pos : nullptr; ea.flags = x->hidden_layer >= 0 ? out->agent_flags : nullptr; ea.layers = x->layers;
  }
  if (perf) {
    ea.perf = x->perf_from_hidden ? out->hidden : out->cumulative; ea.perf_cols = x->perf_from_hidden ? 1 : A * K;
    ea.per_agent = (e->spec.family == SGW_ISLAND_NAVIGATION_EX_MA || e->spec.family == SGW_AINTELOPE_SAVANNA) ? 1 : 0;
    ea.step_type = out->step_type; ea.last = x->perf_last; ea.sum = x->perf_sum; ea.count = reinterpret_cast<long long*>(x->perf_count); ea.done = x->done;
  }
  // LDS: boards | layer masks | per-cell + per-char tables | colour table
  ea.lds_planes = planes ? (int)((((size_t)epb * HW + 15) & ~(size_t)15) + epb * HW * 4 + HW * 8 + 512 + 3 * 128) : 0;
  const size_t lds = (size_t)ea.lds_planes;
  int rc = raise_lds_cap(e, reinterpret_cast<const void*>(&k_step_extras), lds, 64u);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(e->device));
  hipLaunchKernelGGL(k_step_extras, dim3((unsigned)((e->n_envs + epb - 1) / epb)), dim3(PLANES_THREADS), lds, (hipStream_t)stream, ea);
  HIP_TRY(hipGetLastError());
  return SGW_OK;
}

static int step_full_launches(sgw_engine* e, const int8_t* actions_dev, const sgw_out* out, const sgw_extras* x, void* stream) {
  int rc = sgw_step(e, actions_dev, out, stream);
  if (rc || !x) return rc;
  const bool layers_here = x->layers && e->spec.family != SGW_AINTELOPE_SAVANNA;       // savanna's layers come from its state bitmaps
  rc = step_extras_launch(e, out, x, layers_here, stream);
  if (rc) return rc;
  if (x->layers && !layers_here) { rc = sgw_state_layers(e, x->layer_chars_dev, x->n_layers, 1, x->layers, stream); if (rc) return rc; }
  if (x->stats) { rc = sgw_derived_stats(e, out->reward, out->cumulative, out->frame, x->k_agent, x->stats, stream); if (rc) return rc; }
  if (x->agent_layer_views) {
    const bool rot = e->spec.family != SGW_FIREMAKER_EX_MA || e->ks.view_rotates;   // observation directions come in agent_flags (UP = no rotation)
    rc = sgw_agent_layer_views(e, x->layers, out->agent_pos, rot ? out->agent_flags : nullptr, x->layer_chars_dev, x->n_layers,
                               (uint8_t)(e->ks.view_pad), x->agent_layer_views, stream);
    if (rc) return rc;
  }
  return SGW_OK;
}

int sgw_step_full(sgw_engine* e, const int8_t* actions_dev, const sgw_out* out, const sgw_extras* x, void* stream) {
  if (!e || !actions_dev || !out) return fail(SGW_ERR_ARG, "sgw_step_full: null argument");
  if (x) {
    if ((x->rgb || (x->layers && e->spec.family != SGW_AINTELOPE_SAVANNA)) && !out->board)
      return fail(SGW_ERR_ARG, "sgw_step_full: rgb / layers need the board output");
    if (x->rgb && !x->rgb_lut_dev) return fail(SGW_ERR_ARG, "sgw_step_full: rgb requested without a LUT");
    if (x->layers && (!x->layer_chars_dev || x->n_layers < 1)) return fail(SGW_ERR_ARG, "sgw_step_full: bad layer list");
    if (x->layers && x->hidden_layer >= 0 && (!out->agent_pos || !out->agent_flags))
      return fail(SGW_ERR_ARG, "sgw_step_full: a hidden drape layer needs the agent_pos and agent_flags outputs");
    if (x->agent_layer_views && (!x->layers || !out->agent_pos || !out->agent_flags))
      return fail(SGW_ERR_ARG, "sgw_step_full: agent layer cubes need `layers` and the agent_pos / agent_flags outputs");
    if (x->stats && (!out->reward || !out->cumulative || !out->frame))
      return fail(SGW_ERR_ARG, "sgw_step_full: stats need the reward, cumulative and frame outputs");
    if ((x->perf_last || x->perf_sum || x->perf_count || x->done) && (!out->step_type || !(x->perf_from_hidden ? (const void*)out->hidden : (const void*)out->cumulative)))
      return fail(SGW_ERR_ARG, "sgw_step_full: performance bookkeeping needs step_type and the cumulative (or hidden) output");
  }
  const hipStream_t st = (hipStream_t)stream;
  sgw_extras key_x; memset(&key_x, 0, sizeof(key_x));
  if (x) key_x = *x;
  if (!x || !x->replay || step_graphs_min_T() > 8) return step_full_launches(e, actions_dev, out, x, stream);     // (SGW_STEP_GRAPHS=0: never replay)
  sgw_engine::FullGraph* hit = nullptr;
  for (auto& g : e->full_graphs)
    if (g.actions == actions_dev && memcmp(&g.out, out, sizeof(sgw_out)) == 0 && memcmp(&g.ex, &key_x, sizeof(key_x)) == 0) { hit = &g; break; }
  if (!hit) {
    while (e->full_graphs.size() >= 16) {
      size_t worst = 0;
      for (size_t i = 1; i < e->full_graphs.size(); ++i) if (e->full_graphs[i].last_use < e->full_graphs[worst].last_use) worst = i;
      if (e->full_graphs[worst].exec) (void)hipGraphExecDestroy(e->full_graphs[worst].exec);
      e->full_graphs.erase(e->full_graphs.begin() + worst);
    }
    sgw_engine::FullGraph g; memset(&g, 0, sizeof(g));
    g.actions = actions_dev; g.out = *out; g.ex = key_x; g.last_use = ++e->graph_tick; g.exec = nullptr;
    e->full_graphs.push_back(g);
    return step_full_launches(e, actions_dev, out, x, stream);
  }
  hit->last_use = ++e->graph_tick;
  if (!hit->exec) {
    HIP_TRY(hipSetDevice(e->device));
    if (!e->capture_stream) HIP_TRY(hipStreamCreateWithFlags(&e->capture_stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamBeginCapture(e->capture_stream, hipStreamCaptureModeThreadLocal));
    const int rc = step_full_launches(e, actions_dev, out, x, e->capture_stream);
    hipGraph_t graph = nullptr;
    const hipError_t ec = hipStreamEndCapture(e->capture_stream, &graph);
    if (rc || ec != hipSuccess || !graph) {
      if (graph) (void)hipGraphDestroy(graph);
      if (rc) return rc;
      return fail(SGW_ERR_HIP, "sgw_step_full: stream capture failed");
    }
    const hipError_t ei = hipGraphInstantiate(&hit->exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ei != hipSuccess) { hit->exec = nullptr; return fail(SGW_ERR_HIP, "sgw_step_full: hipGraphInstantiate failed"); }
  }
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipGraphLaunch(hit->exec, st));
  return SGW_OK;
}

int sgw_get_state(sgw_engine* e, uint64_t* state_dev, void* stream) {
  if (!e || !state_dev) return fail(SGW_ERR_ARG, "sgw_get_state: null argument");
  HIP_TRY(hipSetDevice(e->device));
  const long long total = (long long)e->ks.words * e->n_pad;
  hipLaunchKernelGGL(k_state_export, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, e->state_dev,
                     e->n_pad, e->ks.words, state_dev);
  HIP_TRY(hipGetLastError());
  return SGW_OK;
}

int sgw_set_state(sgw_engine* e, const uint64_t* state_dev, void* stream) {
  if (!e || !state_dev) return fail(SGW_ERR_ARG, "sgw_set_state: null argument");
  HIP_TRY(hipSetDevice(e->device));
  const long long total = (long long)e->ks.words * e->n_pad;
  hipLaunchKernelGGL(k_state_import, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, e->state_dev,
                     e->n_pad, e->ks.words, state_dev);
  HIP_TRY(hipGetLastError());
  e->rng_set = 1;          // a saved state carries its envs' generator streams (resume needs no sgw_set_rng_state)
  return SGW_OK;
}

}  // extern "C"
