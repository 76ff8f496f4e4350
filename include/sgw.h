/* sgw.h -- C ABI of libsgw.so, the MI355X-native batched safety-gridworld step engine.
 *
 * This is the drop-in boundary for the hot path of levitation-opensource/ai-safety-gridworlds:
 * what a maintainer of the reference would bind (ctypes) in place of the per-env Python object
 * graph.  One engine advances N independent env instances in lockstep; every entry point is
 * one batched operation over device memory.  No torch types, no C++ types: plain pointers and
 * sizes.  All pointers named *_dev are DEVICE pointers owned by the caller (e.g. a torch
 * tensor's data_ptr()); `stream` is a hipStream_t passed as void* (NULL = default stream).
 * Functions return 0 on success, a negative code on failure; sgw_last_error() describes it.
 * The library never falls back to the CPU: without a usable HIP device every compute entry
 * point fails with SGW_ERR_HIP.
 *
 * Reference interfaces replaced (file:line in the reference tree):
 *   sgw_create   <- SafetyEnvironment*.__init__ + game_factory           (safety_game.py:82-165,
 *                   safety_game_mo.py:148-404; env ctors e.g. island_navigation_ex.py:707-819)
 *   sgw_reset    <- Environment.reset -> make_game -> Engine.its_showtime (pycolab_interface.py:133-145,
 *                   pycolab_interface_mo.py:142-155, safety_game_mo.py:526-724, engine.py:520-581)
 *   sgw_step     <- Environment.step -> Engine.play -> _process_timestep  (pycolab_interface.py:147-192,
 *                   pycolab_interface_mo.py:157-196, _ma.py:173-246, engine.py:583-639,
 *                   safety_game.py:265-304, safety_game_mo.py:971-1066, safety_game_moma.py:1183-1379)
 *   sgw_observe  <- ObservationToArrayWithRGB{,Ex}.__call__              (observation_distiller.py:30-91,
 *                   observation_distiller_ex.py:147-187, rendering.py:188-302, 410-549)
 *   sgw_rollout  <- the user's `for t: env.step(random_action)` loop (fused, benchmark mode)
 */
#ifndef SGW_H_
#define SGW_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGW_ABI_VERSION 8
#define SGW_MAX_CELLS 320      /* >= 17*17 */
#define SGW_MAX_K 16           /* reward dimensions per agent */
#define SGW_MAX_M 32           /* metrics per env */
#define SGW_MAX_AGENTS 4
#define SGW_N_PARAMS 72
#define SGW_ENV_ALIGN 64       /* per-env output buffers must hold round_up(N, 64) envs */

enum sgw_family {
  SGW_ISLAND_NAVIGATION_EX = 0,   /* environments/island_navigation_ex.py */
  SGW_BOAT_RACE_EX = 1,           /* environments/boat_race_ex.py */
  SGW_BOAT_RACE = 2,              /* environments/boat_race.py */
  SGW_SAFE_INTERRUPTIBILITY = 3,  /* environments/safe_interruptibility.py */
  SGW_FIREMAKER_EX_MA = 4,        /* environments/firemaker_ex_ma.py */
  SGW_ISLAND_NAVIGATION_EX_MA = 5, /* environments/island_navigation_ex_ma.py (agents terminate individually) */
  SGW_TILE_EVENTS = 6,             /* island_navigation.py, distributional_shift.py, absent_supervisor.py: one table-driven family */
  SGW_SIDE_EFFECTS_SOKOBAN = 7,    /* environments/side_effects_sokoban.py */
  SGW_CONVEYOR_BELT = 8,           /* environments/conveyor_belt.py */
  SGW_TOMATO_WATERING = 9,         /* environments/tomato_watering.py */
  SGW_FRIEND_FOE = 10,             /* environments/friend_foe.py */
  SGW_WHISKY_GOLD = 11,            /* environments/whisky_gold.py */
  SGW_ROCKS_DIAMONDS = 12,         /* environments/rocks_diamonds.py */
  SGW_AINTELOPE_SAVANNA = 13       /* environments/aintelope/aintelope_savanna.py (one or two agents; per-agent outputs are [N, 2]) */
};

enum sgw_step_type { SGW_FIRST = 0, SGW_MID = 1, SGW_LAST = 2, SGW_DEAD = 3 }; /* rl/environment{,_ma}.py */
enum sgw_term { SGW_TERMINATED = 0, SGW_MAX_STEPS = 1, SGW_INTERRUPTED = 2, SGW_QUIT = 3,
                SGW_TERM_NONE = 255 };  /* termination_reason_enum.py:25-39; 255 = key absent */

enum sgw_error {
  SGW_OK = 0, SGW_ERR_ARG = -1, SGW_ERR_HIP = -2, SGW_ERR_UNSUPPORTED = -3, SGW_ERR_NOMEM = -4
};

/* Game spec: the new framework's equivalent of GAME_ART + Sprite/Drape classes + flags, prepared
 * on the host (ai_safety_gridworlds_amd/specs.py) as flat tables for SoA device state.
 * Per-family meaning of flags/params/aux is documented in csrc/sgw_<family>.hpp. */
typedef struct sgw_spec {
  int32_t family;
  int32_t H, W;
  int32_t K;                 /* enabled reward dimensions (per agent; max over agents) */
  int32_t M;                 /* metrics per env */
  int32_t A;                 /* agents per env (1 unless multi-agent) */
  int32_t max_iterations;
  int32_t start_cell[SGW_MAX_AGENTS];   /* row*W+col of each agent sprite in the art */
  int32_t action_lo, n_actions;         /* action range, used by the synthetic rollout */
  int32_t flags;
  int32_t view_radius[SGW_MAX_AGENTS][4]; /* agent-centric view: tiles visible up, down, left, right (-1 = no view) */
  int32_t view_outside;                 /* what_lies_outside: the character of window cells beyond the board (0 = '#') */
  int32_t reserved[2];
  int8_t dim_slot[SGW_MAX_AGENTS][SGW_MAX_K]; /* reward-universe dim -> output column, -1 = not enabled */
  int8_t metric_slot[SGW_MAX_M];              /* family metric id -> output column, -1 = absent */
  double params[SGW_N_PARAMS];
  float value_map[128];                 /* ascii -> float observation value (value_mapping) */
  uint8_t static_board[SGW_MAX_CELLS];  /* rendered board minus dynamic entities */
  uint8_t art[SGW_MAX_CELLS];           /* original_board */
  uint8_t aux[SGW_MAX_CELLS];           /* family-specific per-cell table */
} sgw_spec;

/* Per-step outputs.  Any pointer may be NULL (that output is skipped).  Env-major arrays, N_pad =
 * round_up(N, 64) rows must be allocated; rows >= N are scratch. */
typedef struct sgw_out {
  uint8_t* board;        /* [N_pad, H*W]  ascii codes of the rendered board (Observation.board) */
  float* obs_board;      /* [N_pad, H*W]  value-mapped float32 board (observation['board']) */
  double* reward;        /* [N_pad, A, K] reward vector in sorted enabled-dimension order; 0 at FIRST */
  double* cumulative;    /* [N_pad, A, K] episode return so far (observation['cumulative_reward']) */
  uint8_t* step_type;    /* [N_pad, A]    sgw_step_type; done == SGW_LAST */
  uint8_t* term_reason;  /* [N_pad]       sgw_term, SGW_TERM_NONE unless LAST; [N_pad, A] for island_navigation_ex_ma (per agent,
                          *               set once EVERY agent is done, safety_game_moma.py:1219-1233) */
  int8_t* actual_action; /* [N_pad, A]    extra_observations['actual_actions'], -1 = absent */
  double* discount;      /* [N_pad]       NaN at FIRST (None) */
  double* hidden;        /* [N_pad]       the_plot['hidden_reward'] of the running episode */
  int32_t* safety;       /* [N_pad]       environment_data['safety'] (island_navigation_ex); [N_pad, A] 'safety_<agent>' for
                          *               island_navigation_ex_ma */
  double* metrics;       /* [N_pad, M]    metrics_dict values in METRICS_LABELS order */
  int32_t* frame;        /* [N_pad]       the_plot.frame */
  uint8_t* agent_pos;    /* [N_pad, A, 2] (row, col) of every agent sprite */
  uint8_t* agent_flags;  /* [N_pad, A]    bit0: a dynamic drape (firemaker: fire) lies hidden under this agent; bits 1-2 action
                          *               direction, bits 3-4 observation direction (Directions LEFT=0 RIGHT=1 UP=2 DOWN=3) */
  int32_t* safety2;      /* [N_pad, A]    environment_data['safety2_<agent>'] (aintelope_savanna.py:619-620, 837-844: Manhattan
                          *               distance to the nearest predator, 99 = none, 3 before the agent's first update);
                          *               aintelope_savanna only, SGW_ERR_UNSUPPORTED elsewhere */
  uint8_t* views;        /* [N_pad, view_bytes] the agent-centric windows of the rendered board (get_agent_perspective,
                          *               safety_game_moma.py:1996-2101) produced INSIDE the step launch from the board rows the
                          *               kernel already holds in LDS: same layout and contents as sgw_agent_views (agent a's window
                          *               at byte offset sum of the previous agents' window sizes, rot90-ed by the observation
                          *               direction where the env has one, cells outside the board = spec.view_outside).  What the
                          *               Zoo wrapper hands to the agents as `obs` (gridworld_zoo_parallel_env.py:541-554): one launch
                          *               per round instead of step + sgw_agent_views.  Families with agent views only
                          *               (firemaker_ex_ma, island_navigation_ex_ma, aintelope_savanna), SGW_ERR_UNSUPPORTED elsewhere */
  float* obs_views;      /* [N_pad, view_bytes] the same windows value-mapped to float32 (ascii_observation_format=False:
                          *               the window of observation['board'], observation_distiller_ex.py:147-187) */
  /* The wrappers' per-step decodes of the outputs above, written by the same launch so that a Python step does not launch a
   * torch kernel per derived tensor (a step from Python is host-bound; every extra launch is ~5 us of host and ~2.5 us of GPU time): */
  uint8_t* done;         /* [N_pad, A]    1 where the (agent's) step_type is LAST or later: the `terminated` of the Gym / Zoo wrappers
                          *               (gridworld_gym_env.py:563-578, gridworld_zoo_parallel_env.py:589-600) */
  uint8_t* obs_dir;      /* [N_pad, A]    the agent's observation direction (bits 3-4 of agent_flags; Directions LEFT=0 RIGHT=1 UP=2
                          *               DOWN=3): the wrappers' INFO_OBSERVATION_DIRECTION (gridworld_zoo_parallel_env.py:325-335).
                          *               Meaningful for the families that carry directions (island_navigation_ex_ma,
                          *               aintelope_savanna, firemaker_ex_ma); 0 elsewhere */
  uint8_t* act_dir;      /* [N_pad, A]    the agent's action direction (bits 1-2 of agent_flags): INFO_ACTION_DIRECTION */
} sgw_out;

typedef struct sgw_engine sgw_engine;

int sgw_abi_version(void);
const char* sgw_last_error(void);
/* The compiler this library was built with ("HIP <version>; <clang version>"): every build is checked for the gfx950
 * register-allocator defect of DESIGN.md §8 before it is installed, so the string names the compiler the lint passed on. */
const char* sgw_build_info(void);

/* Bytes of one sgw_spec / sgw_out as compiled (lets a binding verify its struct mirror). */
int sgw_sizeof_spec(void);
int sgw_sizeof_out(void);
int sgw_sizeof_extras(void);

/* Create an engine for n_envs instances on HIP device `device`.  env_id_base is the global id
 * of env 0 (keys the counter-based RNG so results do not depend on the GPU count).
 * The engine owns its SoA state (hipMalloc).  All envs start "not yet reset". */
int sgw_create(const sgw_spec* spec, int64_t n_envs, int64_t env_id_base, int device,
               sgw_engine** out_engine);
int sgw_destroy(sgw_engine* e);

int64_t sgw_n_envs(const sgw_engine* e);
int64_t sgw_n_pad(const sgw_engine* e);
int64_t sgw_state_bytes(const sgw_engine* e);

/* Per-game-build external inputs: ONE number of the process-global numpy RNG per make_game (safe_interruptibility.py:256-258
 * should_interrupt; absent_supervisor.py:91-93 supervisor present; distributional_shift.py:93-95 test level 1 or 2).
 * bits_dev is uint8 [N, n_per_env]; the k-th game build of env n uses bits[n, k % n_per_env].  NULL => drawn from
 * Philox(seed, env id, build) against the env's probability. */
int sgw_set_episode_bits(sgw_engine* e, const uint8_t* bits_dev, int n_per_env, uint64_t seed);

/* Random numbers for envs that draw from the process-global numpy RNG an input-dependent number of times (tomato_watering.py:
 * 154-156 and tomato_crmdp.py: np.random.random() per watered tomato and step; friend_foe.py:145-155: np.random.choice of the
 * bandit and np.random.rand() per game build; whisky_gold.py:158-163: rand() + choice() per explored step).  A `choice`
 * among k items is floor(k * u).  u_dev double [N, n_per_env]: the k-th draw of env i is
 * u[i][k % n_per_env] (replaying a reference run); NULL: Philox(seed, global env id, k).  The draw counter is env state. */
int sgw_set_random_stream(sgw_engine* e, const double* u_dev, int n_per_env, uint64_t seed);

/* Env-owned numpy Generators (environment_data[NP_RANDOM] = seeding.np_random(seed), safety_game_mo.py:283-291): firemaker_ex_ma
 * (fire spread, agent order), island_navigation_ex_ma (agent order, map randomisation) and aintelope_savanna (map generation,
 * agent order, predators, tile spawning; also Generator.choice / integers: Lemire bounded draws and Floyd's sampler).  pcg_state_dev is uint64 [N, 4] =
 * numpy PCG64 (state_hi, state_lo, inc_hi, inc_lo) per env, as produced by np.random.PCG64(SeedSequence(seed)).  The engine
 * advances the streams exactly like numpy (random(), buffered next_uint32, shuffle, random_interval). */
int sgw_set_rng_state(sgw_engine* e, const uint64_t* pcg_state_dev);

/* Family lookup table in device memory, copied from the host: aintelope_savanna's visit-count rewards
 * [gold_reward[0 .. max_iterations + 1], silver_reward[0 .. max_iterations + 1]], entry v = SCORE * (log(v + 2, base) -
 * log(v + 1, base)) evaluated by the caller with the reference's own math.log (aintelope_savanna.py:956-983), so the
 * device adds the reference's doubles instead of re-deriving logarithms.  Required before the first reset of that family.
 * island_navigation_ex with spec.flags bit 4 (reward flags that put one event on several dimensions, e.g. the experiments/
 * food_drink_rolf* presets): [15 events][12 universe dims] values followed by 15 key-presence masks; events in the order the
 * reference adds them (MOVEMENT, THIRST_HUNGER_DEATH, FINAL, DRINK, NON_DRINK, FOOD, NON_FOOD, GOLD, SILVER, GAP,
 * DRINK_DEFICIENCY, DRINK_OVERSATIATION, FOOD_DEFICIENCY, FOOD_OVERSATIATION, DANGER_TILE; island_navigation_ex.py:449-608). */
int sgw_set_family_table(sgw_engine* e, const double* table_host, int64_t n);

/* out[i] = pow(x[i], y) exactly as the engine's regrowth computes it: the host C library's (glibc) pow algorithm and tables,
 * i.e. the arithmetic of the reference's math.pow (island_navigation_ex.py:603-636, aintelope_savanna.py:1251-1254).
 * x finite, positive, normal; device pointers.  Exposed so that callers and tests can check the device against their libm. */
int sgw_pow_f64(const double* x_dev, double y, double* out_dev, int64_t n, int device, void* stream);

/* Host-side probe, no GPU needed: number of probe inputs (4096 points of the regrowth domain x in [1, 61] at exponents 1.1,
 * 1.05, 1.5) on which the RUNNING host's libm pow() -- the reference's math.pow -- differs from the table-based restatement
 * the device uses (csrc/sgw_pow.hpp over csrc/sgw_pow_tables.inc).  0 on the build host.  sgw_create refuses the families
 * that regrow resources (island_navigation_ex, island_navigation_ex_ma, aintelope_savanna) with SGW_ERR_UNSUPPORTED when it is
 * not 0 (override: environment variable SGW_ALLOW_LIBM_MISMATCH). */
int64_t sgw_pow_selfcheck(void);

/* Start a new episode in every env with mask_dev[n] != 0 (NULL = all) and emit the FIRST
 * timestep into `out` for those envs (other rows of `out` are left untouched). */
int sgw_reset(sgw_engine* e, const uint8_t* mask_dev, const sgw_out* out, void* stream);

/* One env.step() per env: actions_dev int8 [N, A].  Envs whose previous step was LAST are
 * auto-reset instead (action discarded, FIRST emitted) exactly like the reference adapter.
 * Multi-agent families (island_navigation_ex_ma, aintelope_savanna, firemaker_ex_ma): an action < 0 = that agent is not in the
 * submitted dict and does not play this round (EnvironmentMa.step with a subset of the agents, pycolab_interface_ma.py:173-246:
 * the AEC wrapper's way, and the Gym wrapper's with agent_character, gridworld_gym_env.py:476-479). */
int sgw_step(sgw_engine* e, const int8_t* actions_dev, const sgw_out* out, void* stream);

/* T consecutive sgw_step launches (one kernel launch per step, host loop in C): actions_dev int8
 * [T, N, A].  write_every == 0: `out` is overwritten by each step; otherwise arrays are [T, N_pad, ...].
 * accumulate != 0: add the return of every episode that ends to the engine's per-env episodic-return
 * accumulators (see sgw_read_returns).
 * T >= 8: the second call with the same (actions_dev, T, write_every, out pointers, accumulate) captures its T launches into
 * a hipGraph on an engine-owned stream and replays it on `stream` from then on -- refill the action buffer in place to get the
 * benefit (one host launch costs ~5 us, a replayed one 0.2 us); SGW_STEP_GRAPHS=0 in the environment turns this off.  The
 * contents of the buffers are read when the launches run, so replays see the refilled actions. */
int sgw_step_n(sgw_engine* e, const int8_t* actions_dev, int T, int write_every, const sgw_out* out,
               int accumulate, void* stream);

/* Fused benchmark mode: T steps with in-kernel synthetic actions (Philox-4x32-10, key
 * (seed, global env id), counter (step0 + t)); identical to T sgw_step calls fed the same stream.
 * `out` receives the outputs of the LAST of the T steps only when write_every == 0, otherwise
 * arrays are [T, N_pad, ...] and every step is written (rollout-buffer mode).
 * accumulate: as in sgw_step_n. */
int sgw_rollout(sgw_engine* e, int T, uint64_t seed, int64_t step0, int write_every,
                const sgw_out* out, int accumulate, void* stream);

/* The same fused launch over the CALLER's actions: actions_dev int8 [T, N, A] (an action tape: the reference's
 * demonstrations replay, demonstrations.py:65-80, an open-loop plan, a logged episode).  Identical, output for output, to
 * sgw_step_n over the same buffer; the state stays in registers between the steps. */
int sgw_replay(sgw_engine* e, const int8_t* actions_dev, int T, int write_every, const sgw_out* out, int accumulate,
               void* stream);

/* A GROUP: several engines on one device -- different env families, or shards of one -- advanced by ONE kernel launch per step
 * (a heterogeneous grid: each workgroup runs the family body of the engine it belongs to).  This is how a mixed suite sharded over
 * a GPU is stepped (BASELINE config 5: island_navigation_ex + boat_race_ex + safe_interruptibility): one launch and one kernel
 * boundary per step instead of one per family.  Members: the single-agent families (island_navigation_ex with scalar reward
 * flags, boat_race{,_ex}, safe_interruptibility, island_navigation / distributional_shift / absent_supervisor, side_effects_sokoban,
 * conveyor_belt, tomato_watering, friend_foe, whisky_gold, rocks_diamonds), at most 4 engines, all on one device; the round
 * kernels of the multi-agent families fill the chip on their own and are refused (SGW_ERR_UNSUPPORTED).  The group does not own
 * its engines: destroy it before them.  Results are identical to stepping each engine by itself. */
typedef struct sgw_group sgw_group;
int sgw_group_create(sgw_engine* const* engines, int n_engines, sgw_group** out_group);
int sgw_group_destroy(sgw_group* g);
/* sgw_step_n for every member in lockstep: actions_dev[m] int8 [T, N_m, A_m], outs[m] as in sgw_step_n (NULL outs = no outputs).
 * T launches; the second call with the same arguments captures them into a hipGraph and replays it from then on. */
int sgw_group_step_n(sgw_group* g, const int8_t* const* actions_dev, int T, int write_every, const sgw_out* outs,
                     int accumulate, void* stream);
/* sgw_rollout for every member in ONE fused launch (T steps, in-kernel Philox actions keyed by each member's global env ids). */
int sgw_group_rollout(sgw_group* g, int T, uint64_t seed, int64_t step0, int write_every, const sgw_out* outs, int accumulate,
                      void* stream);

/* End-of-batch episodic returns: out_dev double [A*K + 1] = (sum over finished episodes of the
 * episode return vector, number of finished episodes), summed over this engine's envs in a fixed
 * order (deterministic: one accumulator row per 16 envs, each cell with a single adder per launch, + a tree reduction).  This is the
 * buffer a multi-GPU job all-reduces once per batch.  clear != 0 zeroes the accumulators after. */
int sgw_read_returns(sgw_engine* e, double* out_dev, int clear, void* stream);

/* Fill actions_dev int8 [T, N, A] with the same synthetic stream the rollout uses. */
int sgw_fill_actions(sgw_engine* e, int T, uint64_t seed, int64_t step0, int8_t* actions_dev,
                     void* stream);

/* Accumulate (sum of episode returns, #episodes) of envs whose step_type is LAST:
 * ep_accum_dev double [A*K + 1] (atomic adds; exact for integer-valued returns). */
int sgw_accumulate_returns(sgw_engine* e, const double* cumulative_dev, const uint8_t* step_type_dev,
                           double* ep_accum_dev, void* stream);

/* Derived observations from a rendered ascii board (observation distiller, a12):
 * rgb_dev uint8 [N, 3, H*W] via rgb_lut_dev uint8 [128*3]; layers_dev uint8 [N, L, H*W] for the
 * L characters in layer_chars_dev (occluded layers: board == char).  Either may be NULL. */
int sgw_observe(sgw_engine* e, const uint8_t* board_dev, const uint8_t* rgb_lut_dev, uint8_t* rgb_dev,
                const uint8_t* layer_chars_dev, int n_layers, uint8_t* layers_dev, void* stream);

/* Derived per-step statistics of _process_timestep (safety_game_mo.py:1027-1084, gini_coefficient 1645-1681),
 * computed from a step's outputs with numpy's own reduction order (8-accumulator pairwise sums), per env and agent:
 * stats_dev double [N, A, 5 + K] = (gini_index, cumulative_gini_index, mo_variance, cumulative_mo_variance,
 * average_mo_variance, average_reward[K]).  k_agent[A] = number of reward dimensions of each agent (<= K). */
int sgw_derived_stats(sgw_engine* e, const double* reward_dev, const double* cumulative_dev, const int32_t* frame_dev,
                      const int32_t* k_agent, double* stats_dev, void* stream);

/* Per-env performance bookkeeping of SafetyEnvironment{,Mo}: _episodic_performances.append(...) at every LAST timestep
 * (safety_game.py:253-263, 301-302; safety_game_mo.py:1015-1016), get_last_performance (229-251) and get_overall_performance =
 * sum(performances) / len(performances) (194-208, 234-244; safety_game_mo.py:917-938).  perf_dev double [N, C] is the step's
 * performance source (the `cumulative` output, C = A*K, or the `hidden` output, C = 1); where step_type_dev[n, 0..A) is LAST
 * (every agent LAST or DEAD for the families whose agents finish one by one): last_dev[n] = perf[n], sum_dev[n] += perf[n]
 * (the reference's left-to-right sum), count_dev[n] += 1; done_dev[n] = 1 there and 0 elsewhere (the wrapper's `terminated`,
 * gridworld_gym_env.py:563-578).  Any of last / sum / count / done may be NULL. */
int sgw_track_performance(sgw_engine* e, const double* perf_dev, int n_cols, const uint8_t* step_type_dev, double* last_dev,
                          double* sum_dev, int64_t* count_dev, uint8_t* done_dev, void* stream);

/* Everything env.step() returns from ONE host call: sgw_step, then -- on the same stream, chained in C -- whatever of the
 * observation distiller's and _process_timestep's derived outputs `extras` asks for, computed from that step's outputs:
 * RGB (sgw_observe), unoccluded layers (sgw_observe_layers; aintelope_savanna: sgw_state_layers), derived statistics
 * (sgw_derived_stats), per-agent layer cubes (sgw_agent_layer_views), performance bookkeeping (sgw_track_performance).  With
 * extras->replay != 0 the second call with the same (actions_dev, out, extras) pointers captures the launches into a hipGraph and
 * replays it from then on (one graph launch per step instead of up to six kernel launches; refill the action buffer in place);
 * replay == 0 issues the launches directly, which is the faster end to end while the host keeps up.  `out` must carry what the
 * requested extras read: board (rgb, layers), agent_pos + agent_flags (hidden drape, layer cubes), reward + cumulative + frame
 * (stats), cumulative or hidden + step_type (performance).  Reference: gridworld_gym_env.py:455-585, safety_game_mo.py:971-1107,
 * observation_distiller_ex.py:147-187. */
typedef struct sgw_extras {
  uint8_t* rgb;                       /* [N, 3, H*W] or NULL */
  const uint8_t* rgb_lut_dev;         /* uint8 [128*3] */
  uint8_t* layers;                    /* unoccluded [N, L, H*W] or NULL */
  const uint8_t* layer_chars_dev;     /* uint8 [L] */
  const uint8_t* layer_static_dev;    /* uint8 [L, H*W] (unused by aintelope_savanna) */
  int32_t n_layers, gap_index, hidden_layer, perf_from_hidden;
  double* stats;                      /* [N, A, 5 + K] or NULL */
  int32_t k_agent[SGW_MAX_AGENTS];
  uint8_t* agent_layer_views;         /* [N, L * view_bytes] or NULL (needs `layers`) */
  double* perf_last;                  /* [N, C] / [N, C] / int64 [N] / uint8 [N]; all NULL = no bookkeeping */
  double* perf_sum;
  int64_t* perf_count;
  uint8_t* done;
  int32_t replay;                     /* 0: the launches are issued one by one (lowest end-to-end time per step: 38.7 us for the full
                                       * island_navigation_ex set at 65 536 envs, ~25 us of host time); != 0: the chain is captured
                                       * on the second identical call and replayed as ONE hipGraph from the third on (~5 us of host
                                       * time per step for a consumer whose host thread is the bottleneck, at ~5 us more GPU time
                                       * per step: 44.3 us) */
  int32_t reserved_;
} sgw_extras;
int sgw_step_full(sgw_engine* e, const int8_t* actions_dev, const sgw_out* out, const sgw_extras* extras, void* stream);

/* Unoccluded observation layers (BaseUnoccludedObservationRenderer, rendering.py:188-302, which the multi-objective
 * and multi-agent envs use: safety_game_mo_base.py:1157) with the "gap only where every other layer is blank"
 * correction of the distiller (observation_distiller_ex.py:164-178).  layer_static_dev uint8 [L, H*W]: 0/1 = the
 * layer's static curtain (backdrop characters, static drapes), 2 = dynamic (sprite / moving drape: board == char).
 * gap_index = index of the what_lies_beneath layer (-1: no correction).  layers_dev uint8 [N, L, H*W].
 * A dynamic drape hidden under a sprite (fire that spread under an agent) is invisible in the board: pass the
 * step's agent_pos / agent_flags outputs and hidden_layer = that drape's layer index (else NULL, NULL, -1). */
int sgw_observe_layers(sgw_engine* e, const uint8_t* board_dev, const uint8_t* layer_chars_dev,
                       const uint8_t* layer_static_dev, int n_layers, int gap_index, const uint8_t* agent_pos_dev,
                       const uint8_t* agent_flags_dev, int hidden_layer, uint8_t* layers_dev, void* stream);

/* aintelope_savanna: its drapes overlap (a food tile may spawn on gold, a predator may stand on either), so the board
 * shows only the top one and the unoccluded layers come from the state instead: layers_dev uint8 [N, L, H*W] for the
 * characters in layer_chars_dev ('#', ' ', W P D F d f G S, '0', '1'), as of the last step / reset.  gap_only_blank != 0:
 * the ' ' layer is set only where every other layer is blank (observe_gaps_only_where_other_layers_are_blank=True,
 * aintelope_savanna.py:1690), else wherever the backdrop is not a wall. */
int sgw_state_layers(sgw_engine* e, const uint8_t* layer_chars_dev, int n_layers, int gap_only_blank, uint8_t* layers_dev,
                     void* stream);

/* Agent-centric observations (get_agent_perspective, safety_game_moma.py:1996-2101): for every env and
 * agent a, the (up+down+1) x (left+right+1) window of the rendered board centred on the agent, cells
 * outside the board filled with `outside_chr`.  views_dev uint8 [N, view_bytes] with agent a's window at
 * byte offset sum of the previous agents' window sizes (sgw_view_bytes gives the row size).
 * agent_flags_dev (the `agent_flags` output, or NULL): when given, each window is rot90-ed by the agent's observation
 * direction after cropping (observation_direction_mode != 0, safety_game_moma.py:2085-2096; square windows only). */
int sgw_view_bytes(const sgw_engine* e);
int sgw_agent_views(sgw_engine* e, const uint8_t* board_dev, const uint8_t* agent_pos_dev, const uint8_t* agent_flags_dev,
                    uint8_t outside_chr, uint8_t* views_dev, void* stream);

/* The same agent-centric windows for every observation LAYER (agent_perspectives_with_layers,
 * safety_game_moma.py:430-525): layers_dev uint8 [N, L, H*W] (sgw_observe_layers), out uint8 [N, L * view_bytes]
 * laid out per env as agent-major [agent][layer][h][w]; cells outside the board read (layer char == outside_chr). */
int sgw_agent_layer_views(sgw_engine* e, const uint8_t* layers_dev, const uint8_t* agent_pos_dev, const uint8_t* agent_flags_dev,
                          const uint8_t* layer_chars_dev, int n_layers, uint8_t outside_chr, uint8_t* out_dev, void* stream);

/* State copy-out / copy-in (tests, checkpointing) in a layout-independent form: uint64 [words][N_pad], word w of env n at
 * [w][n] (the engine itself keeps word pairs interleaved per 64-env wave; these two calls convert). */
int sgw_state_words(const sgw_engine* e);
int sgw_get_state(sgw_engine* e, uint64_t* state_dev, void* stream);
int sgw_set_state(sgw_engine* e, const uint64_t* state_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif  /* SGW_H_ */
