// sgw_common.hpp -- device-side building blocks shared by every game family (gfx950 / wave64).
//
// Execution model: one lane per env, one 64-lane wavefront per 64 envs ("env-wave"), ENV_WAVES env-waves per workgroup
// (one per SIMD of a CU), grid = ceil(N_pad / 64 / ENV_WAVES).  Everything that is identical for all envs (level tables,
// the family's read-only tables) is staged in LDS ONCE per workgroup and shared by its env-waves; per-env state lives in
// HBM as word pairs interleaved per env-wave (see "state layout") so a wave moves 16 bytes per lane per instruction.
// Env-major outputs ([N, H*W] boards, [N, K] reward vectors) are transposed through LDS so the
// wave writes its 64 rows as one contiguous run of 16-byte-per-lane stores.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sgw.h"

namespace sgw {

constexpr int WAVE = 64;
constexpr int ENV_WAVES = 4;        // independent env-waves per workgroup (non-cooperative families): 256 envs share one table copy
constexpr int TABLE_BYTES = 2048;   // 3*320 level tables + 512 value map + 72*8 params: exactly 2 x 16 B per lane

// ---- kernel arguments (by value => kernarg segment => scalar loads) -------------------------
struct KSpec {
  int family, H, W, HW, K, M, A, max_iterations, flags, action_lo, n_actions, words;
  int start_cell[SGW_MAX_AGENTS];
  int start_row[SGW_MAX_AGENTS], start_col[SGW_MAX_AGENTS];   // start_cell / W and % W (host-computed: no scalar division per launch)
  int8_t dim_slot[SGW_MAX_AGENTS][SGW_MAX_K];
  int8_t metric_slot[SGW_MAX_M];
  // agent-centric windows (sgw_out.views): geometry of agent a's window (0 x 0 = none), its byte offset in the env's row of
  // `view_total` bytes, reciprocals for the row / column split of a cell index (host-computed: no division on the device)
  uint8_t view_up[SGW_MAX_AGENTS], view_left[SGW_MAX_AGENTS], view_h[SGW_MAX_AGENTS], view_w[SGW_MAX_AGENTS];
  uint16_t view_off[SGW_MAX_AGENTS], view_recip[SGW_MAX_AGENTS];     // recip = ceil(65536 / view_w): (k * recip) >> 16 == k / view_w for k < 320
  int view_total, view_pad, view_prefill, recip_W;                   // pad character; any window larger than the board; ceil(65536 / W)
  int view_rotates;                                                  // the env has observation directions (windows are rot90-ed by them)
};

// byte offsets of one env-wave staging buffer's regions, computed ONCE per launch on the host (lds_plan) and read from the
// kernarg segment where needed: deriving them on the device cost a ~30-instruction scalar chain per region with spilled terms
struct LdsPlan { int wave_bytes, vec_r, vec_c, vec_m, vec_a, trash, flag, ain, st, tr, act, pos, flg, disc, hid, saf, frm, views, cstash; };

struct KArgs {
  KSpec sp;
  LdsPlan lp;
  int need;                  // LN_* bits of the requested outputs (host-computed: one scalar instead of 14 pointer tests)
  const uint8_t* tables;     // device copy: static_board | art | aux (3 x SGW_MAX_CELLS) | value_map f32[128] | params f64[48]
  uint64_t* state;           // pair layout [env-wave][word pair][lane][2] (state_index below)
  long long n_pad, n_envs, env_id_base;
  const int8_t* actions;     // [n, A] or nullptr (synthetic)
  const uint8_t* mask;       // reset mask or nullptr
  const uint8_t* ep_bits;    // per-episode external bits [n, ep_bits_n] or nullptr
  int ep_bits_n;
  unsigned long long ep_seed;
  const double* rand_stream; // external in-play random numbers [n, rand_n] (envs that draw from the process-global numpy RNG) or nullptr
  int rand_n;
  unsigned long long rand_seed;
  const double* ftable;      // family lookup table (sgw_set_family_table) or nullptr
  sgw_out out;
  int mode;                  // 0 = step, 1 = reset(mask)
  int T;                     // rollout length (1 for step)
  unsigned long long seed;
  long long step0;
  int write_every;
  double* ep_acc;            // per-wave episodic-return accumulators [n_pad/64][A*K+1], or nullptr
#ifdef SGW_STAMPS
  SGW_STAMP_DECL             // diagnostic build only (tools/diag/stamp_probe.hip): per-wave phase stamps
#endif
};

// leading scalar kernel arguments of k_engine (preloaded into SGPRs at wave launch): what the prologue's loads need
#define SGW_HOT_ARGS(a) (a).state, (a).tables, (a).actions, (a).n_pad, (a).n_envs, (a).sp.words
#define SGW_ACC_PARTS 4          // (SGW_ACC_PER_ENV 0) accumulator rows per env-wave: one per 16 envs (sgw_kernels.hpp accumulate_returns)
// Episodic-return accumulators: 0 = the wave transposes the finished lanes' vectors through LDS and adds 16-env column sums to one
// row per 16 envs (~90 instructions and three LDS round trips per wave and step: 0.75 of a 6.5 us launch at 65 536 envs, 11 of
// 58.6 us at 1 M); 1 = experiment of round 3: one cell per (column, env), [A*K+1][n_pad], added to by the lane whose episode just
// ended (predicated no-return f64 atomics, no LDS, no cross-lane sum).  Measured on one box: island_navigation_ex 7.71 instead of
// 6.67 us per launch at 65 536 envs, 78 instead of 59 us at 1 M (a random agent ends 13 % of its episodes every step: eleven
// scattered atomics per wave and step cost more than the transpose); boat_race_ex, whose episodes are long, 7.50 instead of 7.90.
// Not the default.
#ifndef SGW_ACC_PER_ENV
#define SGW_ACC_PER_ENV 0
#endif
#define SGW_KARGS_OFFSET 48      // byte offset of the KArgs block in k_engine's kernarg segment: 5 x 8 + 4 (+4 padding)

// In-kernel phase stamps: compiled in ONLY by the diagnostic probe (-DSGW_STAMPS); libsgw.so carries none.
#ifdef SGW_STAMPS
#define SGW_STAMP(a, k)                                                                          \
  do {                                                                                           \
    unsigned long long t_;                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                    \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    if ((threadIdx.x & 63) == 0) (a).sgw_stamps[sgw_stamp_wave * 8 + (k)] = t_;                   \
  } while (0)
#define SGW_STAMP_RT(a, k)                                                                       \
  do {                                                                                           \
    unsigned long long t_;                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    if ((threadIdx.x & 63) == 0) (a).sgw_stamps[sgw_stamp_wave * 8 + (k)] = t_;                   \
  } while (0)
#elif defined(SGW_PHASE_PROF)
// diagnostic LIBRARY build (tools/diag/phase_prof.py): wave cycles between the same marks, summed per wave over launches,
// for whatever family runs -- slot k = cycles from mark k-1 to mark k
__device__ unsigned long long g_phase_prof[4096 * 8];
__device__ unsigned long long g_phase_last[4096];
#define SGW_STAMP(a, k)                                                                          \
  do {                                                                                           \
    unsigned long long t_;                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                    \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    const int w_ = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6) & 4095;                   \
    if ((threadIdx.x & 63) == 0) { if ((k) != 0) g_phase_prof[w_ * 8 + (k)] += t_ - g_phase_last[w_]; g_phase_last[w_] = t_; } \
  } while (0)
#define SGW_STAMP_RT(a, k) do { } while (0)
#else
#define SGW_STAMP(a, k) do { } while (0)
#define SGW_STAMP_RT(a, k) do { } while (0)
#endif

__host__ __device__ inline void kspec_derive(KSpec& k) {
  for (int ag = 0; ag < SGW_MAX_AGENTS; ++ag) { k.start_row[ag] = k.W > 0 ? k.start_cell[ag] / k.W : 0; k.start_col[ag] = k.W > 0 ? k.start_cell[ag] % k.W : 0; }
}
// window geometry from the spec's radii (up, down, left, right; < 0 = the agent has no window)
__host__ inline void kspec_views(KSpec& k, const sgw_spec& sp) {
  int off = 0;
  k.view_prefill = 0; k.view_rotates = 0;
  for (int ag = 0; ag < SGW_MAX_AGENTS; ++ag) {
    const int32_t* rad = sp.view_radius[ag];
    k.view_off[ag] = (uint16_t)off; k.view_up[ag] = k.view_left[ag] = k.view_h[ag] = k.view_w[ag] = 0; k.view_recip[ag] = 0;
    if (ag >= sp.A || rad[0] < 0) continue;
    const int vh = rad[0] + rad[1] + 1, vw = rad[2] + rad[3] + 1;
    if (vh > 255 || vw > 255 || off + vh * vw > 65535) { k.view_total = -1; return; }     // refused by the launcher when `views` is asked for
    k.view_up[ag] = (uint8_t)rad[0]; k.view_left[ag] = (uint8_t)rad[2]; k.view_h[ag] = (uint8_t)vh; k.view_w[ag] = (uint8_t)vw;
    k.view_recip[ag] = (uint16_t)((65536 + vw - 1) / vw);
    if (vh * vw > sp.H * sp.W) k.view_prefill = 1;
    off += vh * vw;
  }
  k.view_total = off;
  k.view_pad = sp.view_outside ? sp.view_outside : '#';
  k.recip_W = (65536 + sp.W - 1) / sp.W;
}

enum { MODE_STEP = 0, MODE_RESET = 1 };
enum { ST_FIRST = 0, ST_MID = 1, ST_LAST = 2, ST_NONE = 3 };   // ST_NONE: never reset yet

// ---- LDS layout (dynamic shared memory, carved by the host with the same arithmetic) --------
//   [ level tables 2 KiB | family-private F::LDS_EXTRA (shared by the workgroup) | env-wave 0: buffers | env-wave 1: buffers | ... ]
// An env-wave has one staging buffer (two in the pipelined fused rollout, see k_engine).  EVERY requested output of a step is
// staged in the buffer in exactly the byte order global memory wants for the wave's 64 envs, so that the drain is a plain
// 16-byte-per-lane copy of contiguous bytes -- boards, reward vectors and the one- and four-byte per-env outputs alike
// (a 1-byte-per-lane global store costs ~12x, a 4-byte one ~6x the time per byte of a 16-byte one).
struct Lds {
  uint8_t* static_board;   // [SGW_MAX_CELLS]
  uint8_t* art;            // [SGW_MAX_CELLS]
  uint8_t* aux;            // [SGW_MAX_CELLS]
  float* value_map;        // [128]
  const double* params;    // [SGW_N_PARAMS] family constants (vector registers on demand, not SGPRs)
  uint8_t* extra;          // F::LDS_EXTRA bytes of family-private LDS (16-byte aligned), shared by the workgroup
  uint32_t* board;         // 64*HW bytes (+ slack), THIS wave's 64 board rows, contiguous
  double* vec_r;           // 64*A*K doubles: reward rows
  double* vec_c;           // 64*A*K doubles: cumulative rows
  double* vec_m;           // 64*M doubles: metrics rows
  uint8_t *st, *tr, *pos, *flg;      // step_type [64][A], term_reason [64][PA], agent_pos [64][A][2], agent_flags [64][A]
  int8_t* act;                       // actual_action [64][A]
  double *disc, *hid;                // discount [64], hidden [64]
  int32_t *saf, *frm;                // safety [64][PA], frame [64]
  uint8_t* views;          // 64 * view_total bytes: the wave's 64 rows of agent windows, contiguous as in global memory
  double* cstash;          // [NU][64] doubles: the lanes' cumulative reward vectors parked while the rules run (families with CUM_IN_LDS)
  double* vec_a;           // 64*(A*K+1) doubles: finished-episode returns staged for the per-wave sum
  uint32_t* flag;          // [4] per-step words handed from the computing wave to the draining wave (pipelined rollout)
  int8_t* ain;             // [A][64] synthetic actions handed from the draining wave to the computing wave (pipelined rollout)
  double* trash;           // 64 doubles: where writes of not-enabled reward dimensions / absent metrics land (branch-free)
};

__host__ __device__ inline size_t lds_board_bytes(int HW) { return ((size_t)64 * HW + 15) / 16 * 16 + 16; }
// Regions are carved only for the outputs a launch asked for (host and device evaluate the same arithmetic on the same
// KArgs): the LDS footprint of a wave decides how many waves a CU keeps resident when a launch has more than one per SIMD.
enum { LN_REWARD = 1, LN_CUMULATIVE = 2, LN_METRICS = 4, LN_RETURNS = 8, LN_ST = 16, LN_TR = 32, LN_ACT = 64, LN_POS = 128,
       LN_FLG = 256, LN_DISC = 512, LN_HID = 1024, LN_SAF = 2048, LN_FRM = 4096, LN_BOARD = 8192, LN_OBS = 16384,
       LN_SAF2 = 32768,     // safety2: written straight from registers (emit_small_direct), no staging region
       LN_VIEWS = 65536, LN_OBSVIEWS = 131072,     // agent windows (u8 / value-mapped f32): one LDS image serves both
       LN_DONE = 262144, LN_ODIR = 524288, LN_ADIR = 1048576 };   // the wrappers' decodes: straight from registers (emit_decodes_direct)
__host__ __device__ inline int lds_need(const KArgs& a, bool family_scratch_m) {
  const sgw_out& o = a.out;
  return (o.reward ? LN_REWARD : 0) | (o.cumulative ? LN_CUMULATIVE : 0) | ((o.metrics || family_scratch_m) ? LN_METRICS : 0) |
         (a.ep_acc ? LN_RETURNS : 0) | (o.step_type ? LN_ST : 0) | (o.term_reason ? LN_TR : 0) | (o.actual_action ? LN_ACT : 0) |
         (o.agent_pos ? LN_POS : 0) | (o.agent_flags ? LN_FLG : 0) | (o.discount ? LN_DISC : 0) | (o.hidden ? LN_HID : 0) |
         (o.safety ? LN_SAF : 0) | (o.frame ? LN_FRM : 0) | (o.board ? LN_BOARD : 0) | (o.obs_board ? LN_OBS : 0) |
         (o.safety2 ? LN_SAF2 : 0) | (o.views ? LN_VIEWS : 0) | (o.obs_views ? LN_OBSVIEWS : 0) |
         (o.done ? LN_DONE : 0) | (o.obs_dir ? LN_ODIR : 0) | (o.act_dir ? LN_ADIR : 0);
}
__host__ __device__ inline size_t lds_view_bytes(int vb, int need) { return (need & (LN_VIEWS | LN_OBSVIEWS)) ? ((size_t)64 * vb + 15) / 16 * 16 : 0; }
__host__ __device__ inline size_t lds_rows(int A, int K, int M, int need, int which) {
  const int ak = A * K > 0 ? A * K : 1;
  switch (which) {
    case LN_REWARD: return (need & LN_REWARD) ? ak : 0;
    case LN_CUMULATIVE: return (need & LN_CUMULATIVE) ? ak : 0;
    case LN_METRICS: return (need & LN_METRICS) ? (M > 6 ? M : 6) : 0;      // firemaker parks six mask words here
    default: return ((need & LN_RETURNS) && !SGW_ACC_PER_ENV) ? A * K + 1 : 0;
  }
}
// bytes of the per-env scalar outputs of one wave: `pa` = 1, or A for the families whose term_reason / safety are per agent
__host__ __device__ inline size_t lds_small_bytes(int A, int pa, int need, int which) {
  auto r16 = [](size_t b) { return (b + 15) / 16 * 16; };
  switch (which) {
    case LN_ST: return (need & LN_ST) ? r16((size_t)64 * A) : 0;
    case LN_TR: return (need & LN_TR) ? r16((size_t)64 * pa) : 0;
    case LN_ACT: return (need & LN_ACT) ? r16((size_t)64 * A) : 0;
    case LN_POS: return (need & LN_POS) ? r16((size_t)128 * A) : 0;
    case LN_FLG: return (need & LN_FLG) ? r16((size_t)64 * A) : 0;
    case LN_DISC: return (need & LN_DISC) ? 512 : 0;
    case LN_HID: return (need & LN_HID) ? 512 : 0;
    case LN_SAF: return (need & LN_SAF) ? (size_t)256 * pa : 0;
    default: return (need & LN_FRM) ? 256 : 0;
  }
}
// one staging buffer of one env-wave
__host__ __device__ inline size_t lds_wave_bytes(int HW, int A, int K, int M, int pa, int need, int vb, int cs = 0) {
  const size_t rows = lds_rows(A, K, M, need, LN_REWARD) + lds_rows(A, K, M, need, LN_CUMULATIVE) +
                      lds_rows(A, K, M, need, LN_METRICS) + lds_rows(A, K, M, need, LN_RETURNS) + 1;   // + trash
  size_t small = 16 + (size_t)(64 * A + 15) / 16 * 16;                  // flag words + the synthetic-action inbox (pipelined rollout)
  for (int w = LN_ST; w <= LN_FRM; w <<= 1) small += lds_small_bytes(A, pa, need, w);
  const bool cs_aliased = (need & (LN_REWARD | LN_CUMULATIVE | (SGW_ACC_PER_ENV ? 0 : LN_RETURNS))) != 0;
  return lds_board_bytes(HW) + rows * 64 * 8 + small + lds_view_bytes(vb, need) + (cs_aliased ? 0 : (size_t)cs * 512);
}
__host__ __device__ inline size_t lds_total_bytes(int HW, int A, int K, int M, int pa, int need, int vb, int extra, int env_waves, int buffers, int cs = 0) {
  return TABLE_BYTES + (size_t)extra + (size_t)env_waves * buffers * lds_wave_bytes(HW, A, K, M, pa, need, vb, cs);
}

__host__ __device__ inline LdsPlan lds_plan(int HW, int A, int K, int M, int pa, int need, int vb, int cs = 0) {
  LdsPlan p;
  int o = (int)lds_board_bytes(HW);
  p.vec_r = o; o += 512 * (int)lds_rows(A, K, M, need, LN_REWARD);
  p.vec_c = o; o += 512 * (int)lds_rows(A, K, M, need, LN_CUMULATIVE);
  p.vec_m = o; o += 512 * (int)lds_rows(A, K, M, need, LN_METRICS);
  p.vec_a = o; o += 512 * (int)lds_rows(A, K, M, need, LN_RETURNS);
  p.trash = o; o += 512;
  p.flag = o; o += 16;
  p.ain = o; o += (64 * A + 15) / 16 * 16;
  p.st = o; o += (int)lds_small_bytes(A, pa, need, LN_ST);
  p.tr = o; o += (int)lds_small_bytes(A, pa, need, LN_TR);
  p.act = o; o += (int)lds_small_bytes(A, pa, need, LN_ACT);
  p.pos = o; o += (int)lds_small_bytes(A, pa, need, LN_POS);
  p.flg = o; o += (int)lds_small_bytes(A, pa, need, LN_FLG);
  p.disc = o; o += (int)lds_small_bytes(A, pa, need, LN_DISC);
  p.hid = o; o += (int)lds_small_bytes(A, pa, need, LN_HID);
  p.saf = o; o += (int)lds_small_bytes(A, pa, need, LN_SAF);
  p.frm = o; o += (int)lds_small_bytes(A, pa, need, LN_FRM);
  p.views = o; o += (int)lds_view_bytes(vb, need);
  // the parked cumulative vectors (families with CUM_IN_LDS; cs = A * K rows, indexed by output column) live in a staging region
  // that is only written AFTER the rules -- the reward rows, else the cumulative rows, else the returns rows -- and get rows of
  // their own only when none of those outputs is requested: an extra 11 KB per workgroup took aintelope_savanna from four
  // resident workgroups per CU to three, and a quarter of a 65 536-env launch then waits for a second turn (81 instead of 47 us)
  if (cs > 0 && (need & LN_REWARD)) p.cstash = p.vec_r;
  else if (cs > 0 && (need & LN_CUMULATIVE)) p.cstash = p.vec_c;
  else if (cs > 0 && (need & LN_RETURNS) && !SGW_ACC_PER_ENV) p.cstash = p.vec_a;
  else { p.cstash = o; o += cs * 512; }
  p.wave_bytes = o;
  return p;
}

__host__ __device__ inline Lds lds_carve(uint8_t* smem, const LdsPlan& p, int extra, int slot) {   // slot = wave * buffers + buffer
  Lds l;
  l.static_board = smem;
  l.art = smem + SGW_MAX_CELLS;
  l.aux = smem + 2 * SGW_MAX_CELLS;
  l.value_map = reinterpret_cast<float*>(smem + 3 * SGW_MAX_CELLS);
  l.params = reinterpret_cast<const double*>(smem + 3 * SGW_MAX_CELLS + 512);
  l.extra = smem + TABLE_BYTES;
  uint8_t* w = smem + TABLE_BYTES + extra + slot * p.wave_bytes;
  l.board = reinterpret_cast<uint32_t*>(w);
  l.vec_r = reinterpret_cast<double*>(w + p.vec_r);
  l.vec_c = reinterpret_cast<double*>(w + p.vec_c);
  l.vec_m = reinterpret_cast<double*>(w + p.vec_m);
  l.vec_a = reinterpret_cast<double*>(w + p.vec_a);
  l.trash = reinterpret_cast<double*>(w + p.trash);
  l.flag = reinterpret_cast<uint32_t*>(w + p.flag);
  l.ain = reinterpret_cast<int8_t*>(w + p.ain);
  l.st = w + p.st; l.tr = w + p.tr; l.act = reinterpret_cast<int8_t*>(w + p.act); l.pos = w + p.pos; l.flg = w + p.flg;
  l.disc = reinterpret_cast<double*>(w + p.disc); l.hid = reinterpret_cast<double*>(w + p.hid);
  l.saf = reinterpret_cast<int32_t*>(w + p.saf); l.frm = reinterpret_cast<int32_t*>(w + p.frm);
  l.views = w + p.views;
  l.cstash = reinterpret_cast<double*>(w + p.cstash);
  return l;
}

// tables (2048 bytes = 128 x 16 B) -> LDS in two halves: `issue` only LOADS (unconditionally, index clamped), so the
// caller can put every other global load of the prologue behind it before anything waits; `commit` writes LDS.
struct TableStage { uint4 t0, t1; };
template <int THREADS>
__device__ inline void lds_tables_issue(TableStage& ts, const uint8_t* tables) {
  const uint4* src = reinterpret_cast<const uint4*>(tables);
  if constexpr (THREADS == WAVE) { ts.t0 = src[threadIdx.x]; ts.t1 = src[threadIdx.x + WAVE]; }
  else { ts.t0 = src[threadIdx.x & (TABLE_BYTES / 16 - 1)]; }
}
template <int THREADS>
__device__ inline void lds_tables_commit(const TableStage& ts, uint8_t* smem) {
  uint4* dst = reinterpret_cast<uint4*>(smem);
  if constexpr (THREADS == WAVE) { dst[threadIdx.x] = ts.t0; dst[threadIdx.x + WAVE] = ts.t1; }
  else { dst[threadIdx.x & (TABLE_BYTES / 16 - 1)] = ts.t0; }   // unconditional (upper threads rewrite the same bytes): no branch for the loads to sink into
}

// Within ONE wavefront lanes exchange data through LDS in program order (the LDS pipeline executes
// a wave's ds_* instructions in issue order), so a cross-lane LDS hand-off needs only a COMPILER fence --
// no s_barrier and, unlike __syncthreads(), no s_waitcnt vmcnt(0) that would stall on in-flight global stores.
__device__ inline void lds_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Row element `slot` of this lane's staging row, or the lane's trash cell when the dimension is not enabled: one
// unconditional ds_write instead of a scalar branch around it (a taken/untaken branch costs a lone wave ~20 cycles).
__device__ inline double* stage_cell(double* region, double* trash, int lane, int row_len, int slot) {
  return slot >= 0 ? region + lane * row_len + slot : trash + lane;
}
// the same for a run of cells of one region: the lane's row and trash cell are worked out once, a cell is then one select
// between the two and one add of the (scalar) slot
struct StageRow {
  double* row; double* trash;
  __device__ StageRow(double* region, double* trash_base, int lane, int row_len) : row(region + lane * row_len), trash(trash_base + lane) {}
  __device__ double* cell(int slot) const { return (slot >= 0 ? row : trash) + (slot >= 0 ? slot : 0); }
};

// ---- Philox-4x32-10 (same stream as ai_safety_gridworlds_amd/philox.py) ---------------------
struct U4 { uint32_t x, y, z, w; };
__device__ inline U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                   uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}
enum { TAG_ACTION = 0, TAG_EPISODE = 1 };
__device__ inline int synth_action(unsigned long long seed, long long env_id, long long step,
                                   int agent, int lo, int n) {
  U4 r = philox4x32_10((uint32_t)step, TAG_ACTION, (uint32_t)agent, (uint32_t)((uint64_t)env_id >> 32),
                       (uint32_t)seed, (uint32_t)env_id);
  return lo + (int)(((uint64_t)r.x * (uint32_t)n) >> 32);
}
__device__ inline double episode_uniform(unsigned long long seed, long long env_id, uint32_t episode) {
  U4 r = philox4x32_10(episode, TAG_EPISODE, 0, (uint32_t)((uint64_t)env_id >> 32),
                       (uint32_t)seed, (uint32_t)env_id);
  uint64_t bits = ((uint64_t)r.x << 21) | (r.y >> 11);
  return (double)bits * (1.0 / 9007199254740992.0);
}

// k-th in-play random number of an env: from the caller's stream when one is set (replaying a reference run), otherwise
// Philox: draw k is word pair (k & 1) of Philox(seed, env id, block k >> 1) -- one call yields two uniforms (philox_pair), which
// tomato_watering's drying pass uses (up to 24 draws per step).  The counter is part of the env state (it keeps running across
// episodes, like np.random).
__device__ inline void philox_pair(const KArgs& a, long long env_id, uint32_t block, double& u0, double& u1) {
  U4 r = philox4x32_10(block, TAG_EPISODE, 1u, (uint32_t)((uint64_t)env_id >> 32), (uint32_t)a.rand_seed, (uint32_t)env_id);
  u0 = (double)(((uint64_t)r.x << 21) | (r.y >> 11)) * (1.0 / 9007199254740992.0);
  u1 = (double)(((uint64_t)r.z << 21) | (r.w >> 11)) * (1.0 / 9007199254740992.0);
}
__device__ inline double next_uniform(const KArgs& a, long long env, long long env_id, uint32_t& counter) {
  double u;
  if (a.rand_stream) u = (env < a.n_envs) ? a.rand_stream[env * a.rand_n + (long long)(counter % (uint32_t)a.rand_n)] : 1.0;
  else {
    double u0, u1;
    philox_pair(a, env_id, counter >> 1, u0, u1);
    u = (counter & 1u) ? u1 : u0;
  }
  counter += 1;
  return u;
}

// Write-through stores (`sc1`): the bytes leave the XCD's L2 while the kernel is still running instead of sitting
// dirty until the end-of-kernel write-back, which otherwise serialises ~19 MB of drain behind the last wave
// (MI355X_MICROARCH.md: kernel boundary "+ B / 6 TB/s when the predecessor leaves B bytes dirty"; "publish-large").
typedef unsigned int sgw_u32x4 __attribute__((ext_vector_type(4)));
__device__ inline void store16_wt(void* p, const uint4& v) {
#ifdef SGW_PLAIN_STORES
  *reinterpret_cast<uint4*>(p) = v;
#else
  sgw_u32x4 d = {v.x, v.y, v.z, v.w};
  // s_nop 1: the assembler-level store is invisible to the compiler's hazard recognizer (a VALU write to the data VGPRs of
  // a >64-bit VMEM store needs 2 wait states on gfx940-class parts); the nop makes every build safe whatever is scheduled next
#ifndef SGW_WT_MOD
#define SGW_WT_MOD "sc1"          // experiments: "nt", "sc0 sc1", "sc1 nt" (profiles/README.md records what they measured)
#endif
  asm volatile("global_store_dwordx4 %0, %1, off " SGW_WT_MOD "\n\ts_nop 1" : : "v"(p), "v"(d) : "memory");
#endif
}
template <class T> __device__ inline void store_wt(T* p, T v) {   // 1/4/8-byte scalar outputs
#ifdef SGW_PLAIN_STORES
  *p = v;
#else
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ inline void store8_wt(uint64_t* p, uint64_t v) {
#ifdef SGW_PLAIN_STORES
  *p = v;
#else
  asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
#endif
}

// ---- state layout and access -------------------------------------------------------------------
// One env-wave (64 consecutive envs) owns a contiguous block of ceil(words / 2) KiB: [word pair][lane][2 words].  Lane l's
// words 2p and 2p+1 are the 16 bytes at (p * 64 + l) * 16, so a wave moves a word PAIR of its 64 envs with one
// 16-byte-per-lane instruction (1 KiB, fully coalesced) -- half the vector-memory instructions of word columns, and the
// widest (cheapest per byte) access the memory pipeline has.  Single words are 8-byte accesses at a 16-byte lane stride.
__host__ __device__ inline long long state_pairs(int words) { return (words + 1) >> 1; }
__host__ __device__ inline long long state_index(int w, long long env, int words) {       // index in 8-byte words
  return ((env >> 6) * state_pairs(words) + (w >> 1)) * 128 + ((env & 63) << 1) + (w & 1);
}
__host__ __device__ inline long long state_alloc_words(int words, long long n_pad) { return (n_pad >> 6) * state_pairs(words) * 128; }

__device__ inline uint64_t ld_word(const KArgs& a, int w, long long env) { return a.state[state_index(w, env, a.sp.words)]; }
__device__ inline void st_word(const KArgs& a, int w, long long env, uint64_t v) { a.state[state_index(w, env, a.sp.words)] = v; }
__device__ inline double ld_f64(const KArgs& a, int w, long long env) { return __longlong_as_double((long long)ld_word(a, w, env)); }
__device__ inline void st_f64(const KArgs& a, int w, long long env, double v) { st_word(a, w, env, (uint64_t)__double_as_longlong(v)); }

// Sequential cursor over one env's words 0, 1, 2, ...: alternates between the two halves of a pair (+1 word) and the next
// pair (+127 words).  In straight-line code `odd` folds to a constant; after a conditional advance it is a wave-uniform scalar.
struct Cursor {
  uint64_t* p;
  uint64_t* w0;          // the env's word 0 (always valid: where disabled conditional reads go)
  int odd;
  __device__ Cursor(const KArgs& a, long long env) : p(a.state + state_index(0, env, a.sp.words)), w0(p), odd(0) {}
  __device__ void advance() { p += odd ? 127 : 1; odd ^= 1; }
  __device__ uint64_t get() { uint64_t v = *p; advance(); return v; }
  __device__ double getf() { return __longlong_as_double((long long)get()); }
  __device__ void put(uint64_t v) { *p = v; advance(); }
  __device__ void putf(double v) { put((uint64_t)__double_as_longlong(v)); }
  __device__ void skip(int n) { for (int i = 0; i < n; ++i) advance(); }
  // conditional (wave-uniform) access without a branch: a disabled slot reads the env's word 0 and contributes `fallback`;
  // an enabled one advances the cursor.
  __device__ double getf_if(bool on, double fallback) {
    const uint64_t* q = on ? p : w0;
    const double v = __longlong_as_double((long long)*q);
    p += on ? (odd ? 127 : 1) : 0;
    odd ^= on ? 1 : 0;
    return on ? v : fallback;
  }
};

// Word PAIRS, 16 bytes per lane per instruction (families whose word order is fixed at compile time).
struct Cursor2 {
  uint4* p;
  __device__ Cursor2(const KArgs& a, long long env) : p(reinterpret_cast<uint4*>(a.state + state_index(0, env, a.sp.words))) {}
  __device__ void get2(uint64_t& x, uint64_t& y) {
    const uint4 v = *p; p += 64;
    x = (uint64_t)v.x | ((uint64_t)v.y << 32); y = (uint64_t)v.z | ((uint64_t)v.w << 32);
  }
  // `wt`: write-through (sc1) -- the bytes leave the XCD's L2 while the kernel runs instead of at the kernel boundary
  template <bool WT> __device__ void put2(uint64_t x, uint64_t y) {
    const uint4 v = make_uint4((uint32_t)x, (uint32_t)(x >> 32), (uint32_t)y, (uint32_t)(y >> 32));
    if constexpr (WT) store16_wt(p, v); else *p = v;
    p += 64;
  }
};
// value select by bit masks: `i == 0 ? a : (i == 1 ? b : c)` written with ?: over struct fields is turned by the compiler
// into ONE load through a select of ADDRESSES, which pins the whole State struct in scratch memory
__device__ inline double sel3_f64(int i, double a, double b, double c) {
  const uint64_t ma = 0ull - (uint64_t)(i == 0), mb = 0ull - (uint64_t)(i == 1), mc = ~(ma | mb);
  return __longlong_as_double((long long)(((uint64_t)__double_as_longlong(a) & ma) | ((uint64_t)__double_as_longlong(b) & mb) |
                                          ((uint64_t)__double_as_longlong(c) & mc)));
}
__device__ inline double u2f(uint64_t v) { return __longlong_as_double((long long)v); }
__device__ inline uint64_t f2u(double v) { return (uint64_t)__double_as_longlong(v); }

// ---- cooperative (wave-wide) env-major output stores -----------------------------------------
// The wave's 64 rows of `row_bytes` bytes are contiguous in global memory at dst + env0*row_bytes
// (env0 % 64 == 0 => 16-byte aligned for any row_bytes); copy them from LDS 16 B per lane.
__device__ inline void coop_store(void* dst, long long env0, int row_bytes, const void* lds_src, int lane) {
  uint4* g = reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(dst) + env0 * row_bytes);
  const uint4* s = reinterpret_cast<const uint4*>(lds_src);
  const int nchunk = 4 * row_bytes;               // 64 * row_bytes / 16
  if (row_bytes <= 16) {                          // the per-env scalars (1-16 bytes per env): at most one chunk per lane
    const uint4 v = s[lane < nchunk ? lane : 0];
    if (lane < nchunk) store16_wt(g + lane, v);
    return;
  }
  // Whole 64-chunk blocks (1 KiB, one chunk per lane) first: which blocks exist is the same for every lane, so a batch of
  // four is four LDS reads at fixed offsets from the lane's address (index clamped by SCALAR arithmetic, issued back to
  // back) and up to four stores behind scalar branches -- no per-lane compare, no exec-mask bookkeeping per chunk.  Only
  // the last, partial block (row_bytes not a multiple of 16) is guarded per lane.
  const int nfull = nchunk >> 6, tail = nchunk & 63;
  const uint4* sl = s + lane;
  uint4* gl = g + lane;
  for (int b = 0; b < nfull; b += 4) {
    const int m = nfull - b;                       // blocks left: >= 1
    const int k1 = m > 1 ? 1 : 0, k2 = m > 2 ? 2 : k1, k3 = m > 3 ? 3 : k2;
    const uint4 v0 = sl[b * 64], v1 = sl[(b + k1) * 64], v2 = sl[(b + k2) * 64], v3 = sl[(b + k3) * 64];
    store16_wt(gl + b * 64, v0);
    if (m > 1) store16_wt(gl + (b + 1) * 64, v1);
    if (m > 2) store16_wt(gl + (b + 2) * 64, v2);
    if (m > 3) store16_wt(gl + (b + 3) * 64, v3);
  }
  if (tail != 0) {
    const uint4 v = sl[(lane < tail ? nfull : 0) * 64];
    if (lane < tail) store16_wt(gl + nfull * 64, v);
  }
}
// Slow path (masked reset): each lane copies only its own row.
__device__ inline void lane_store(void* dst, long long env, int row_bytes, const void* lds_src, int lane) {
  uint8_t* g = reinterpret_cast<uint8_t*>(dst) + env * row_bytes;
  const uint8_t* s = reinterpret_cast<const uint8_t*>(lds_src) + (size_t)lane * row_bytes;
  for (int i = 0; i < row_bytes; ++i) g[i] = s[i];
}

// Board rows -> LDS.  Row of lane l occupies bytes [l*HW, (l+1)*HW) of the wave's board image.
// `base` is the board without sprites (LDS, uniform), `cells`/`chars` the sprites painted on top in z-order.
// HW % 4 == 0: copy the base row with the widest aligned LDS accesses, then patch single bytes.
// Otherwise rows are not dword aligned: each lane shifts its row into place and ORs the dwords into a
// zeroed image (boundary dwords are shared by two neighbouring lanes).
// A row written 16 bytes at a time: quad(j) returns bytes 16j .. 16j+15 of the ROW (row-relative; what it holds past the
// row's end is masked away here).  The callback reads its tables as 16-byte LDS loads and keeps everything that does not
// depend on j outside; the four dwords are funnelled to the row's alignment (row `lane` starts at byte lane * HW) and
// stored.  Dwords strictly inside every lane's row take a plain store behind a SCALAR test; only dword 0 and the row's
// last dword can be shared with a neighbouring row (zeroed by both owners, then OR-ed).  Sprites and other single cells
// go on top afterwards as byte stores into the lane's own row (lds_put_cell): the wave's LDS instructions execute in order.
template <class QuadFn>
__device__ inline void lds_write_row_quads(uint32_t* img, int HW, int lane, QuadFn quad) {
  if ((HW & 3) == 0) {                                                 // every row starts on a dword and shares none: plain stores
    const int ndw = HW >> 2, nq = (HW + 15) >> 4;
    uint32_t* row = img + lane * ndw;
    for (int j = 0; j < nq; ++j) {
      const uint4 c = quad(j);
      const int i = 4 * j;                                             // the guards are scalar
      row[i] = c.x;
      if (i + 1 < ndw) row[i + 1] = c.y;
      if (i + 2 < ndw) row[i + 2] = c.z;
      if (i + 3 < ndw) row[i + 3] = c.w;
    }
    return;
  }
  const int o = lane * HW, q = o & 3, sr = 32 - 8 * q;                 // sr = 32 (aligned row), 24, 16, 8
  uint32_t* row = img + (o >> 2);
  const int last = ((o + HW - 1) >> 2) - (o >> 2);                     // index of the last dword this row touches
  const bool head_shared = q != 0, tail_shared = ((o + HW) & 3) != 0;
  if (head_shared) row[0] = 0u;
  if (tail_shared) row[last] = 0u;
  const int nq = (HW + 15) >> 4, inner = (HW - 1) >> 2;                // dwords 1 .. inner - 1 lie inside every lane's row
#if defined(__HIP_DEVICE_COMPILE__)
  auto funnel = [&](uint32_t lo, uint32_t hi) { return q == 0 ? hi : __builtin_amdgcn_alignbit(hi, lo, (uint32_t)sr); };   // one full-rate v_alignbit_b32
#else
  auto funnel = [&](uint32_t lo, uint32_t hi) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> sr); };
#endif
  auto put = [&](int i, uint32_t v) {
    if (i > 0 && i < inner) { row[i] = v; return; }                    // scalar test
    if (i > last) return;
    if ((i == 0 && head_shared) || (i == last && tail_shared)) { if (v) atomicOr(&row[i], v); } else row[i] = v;
  };
  auto keep = [](int bytes) { return bytes >= 4 ? 0xffffffffu : (bytes <= 0 ? 0u : ((1u << (8 * bytes)) - 1u)); };
  uint32_t prev = 0u;
  for (int j = 0; j < nq; ++j) {
    uint4 c = quad(j);
    const int left = HW - 16 * j;                                      // bytes of the row in this quad (scalar)
    if (left < 16) { c.x &= keep(left); c.y &= keep(left - 4); c.z &= keep(left - 8); c.w &= keep(left - 12); }
    put(4 * j, funnel(prev, c.x)); put(4 * j + 1, funnel(c.x, c.y)); put(4 * j + 2, funnel(c.y, c.z)); put(4 * j + 3, funnel(c.z, c.w));
    prev = c.w;
  }
  put(4 * nq, funnel(prev, 0u));
}
// 0xff in every byte of v that equals the byte replicated in x4 (exact per byte: no carry crosses a byte)
__device__ inline uint32_t bytes_equal_mask(uint32_t v, uint32_t x4) {
  const uint32_t y = v ^ x4;
  const uint32_t nz = ((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y;           // bit 7 of a byte set <=> the byte of y is not zero
  const uint32_t one = (~nz & 0x80808080u) >> 7;
  return (one << 8) - one;
}
// bytes of `idx` are table indices + 1 (0 = none): 0xff in every byte whose index has its bit set in `bits`
__device__ inline uint32_t bytes_bit_mask(uint32_t idx, uint32_t bits) {
  uint32_t m = 0u;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t q = (idx >> (8 * k)) & 0xffu;
    m |= (q != 0u && ((bits >> ((q - 1u) & 31u)) & 1u)) ? (0xffu << (8 * k)) : 0u;
  }
  return m;
}
__device__ inline void lds_put_cell(uint32_t* img, int HW, int lane, int cell, uint32_t ch) {
  reinterpret_cast<uint8_t*>(img)[lane * HW + cell] = (uint8_t)ch;
}

template <int NS>
__device__ inline void lds_write_board_row(uint32_t* img, int HW, int lane, const uint8_t* base_bytes,
                                           const int (&cells)[NS], const uint8_t (&chars)[NS]) {
  const uint32_t* base = reinterpret_cast<const uint32_t*>(base_bytes);
  if ((HW & 15) == 0) {
    uint4* row = reinterpret_cast<uint4*>(img) + lane * (HW >> 4);
    const uint4* b4 = reinterpret_cast<const uint4*>(base);
    const int n = HW >> 4;
    for (int i = 0; i < n; i += 4) {                          // four table reads in flight, then the four row writes
      const int i1 = i + 1 < n ? i + 1 : i, i2 = i + 2 < n ? i + 2 : i, i3 = i + 3 < n ? i + 3 : i;
      const uint4 t0 = b4[i], t1 = b4[i1], t2 = b4[i2], t3 = b4[i3];
      row[i] = t0; row[i1] = t1; row[i2] = t2; row[i3] = t3;
    }
    uint8_t* rb = reinterpret_cast<uint8_t*>(img) + lane * HW;
#pragma unroll
    for (int k = 0; k < NS; ++k) rb[cells[k]] = chars[k];
  } else {                                                  // any other size: 16 table bytes per pass, funnelled to the row's alignment
    const uint4* b4 = reinterpret_cast<const uint4*>(base);
    lds_write_row_quads(img, HW, lane, [&](int j) { return b4[j]; });
#pragma unroll
    for (int k = 0; k < NS; ++k) lds_put_cell(img, HW, lane, cells[k], chars[k]);
  }
}
__device__ inline void lds_zero_board(uint32_t* img, int HW) {
  int n = (int)(lds_board_bytes(HW) / 4);
  for (int i = (int)(threadIdx.x & (WAVE - 1)); i < n; i += WAVE) img[i] = 0u;
}

// wave-wide sum (for the episodic-return accumulators)
__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
  return v;
}

}  // namespace sgw
