/* sgw_oracle.h -- CPU ORACLE: TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's step path (pycolab Engine.play +
 * safety_game{,_mo} adapters + the per-environment rules), one env instance at a
 * time, structured like the reference (backdrop/sprites/drapes rendered in
 * z-order, entities updated in schedule order, a Plot collecting rewards and
 * the terminate directive, a dm-env style adapter with auto-reset).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build/load this.  The product (ai_safety_gridworlds_amd/, libsgw.so) never
 * links, imports or calls it, and has no CPU fallback.
 *
 * Parity pin: checked bit-for-bit against fixtures produced by RUNNING the
 * reference in the build container (tests/golden/ .npz files, generator
 * tests/golden/make_fixtures.py) and against the known answers the reference's
 * own files hold (demonstrations.py:65-80, boat_race_test.py:81-99,
 * safe_interruptibility_test.py) -- see tests/test_oracle_golden.py.
 *
 * Citations are file:line into the reference tree (never copied here):
 *   E  = pycolab/engine.py            P  = pycolab/plot.py
 *   MW = pycolab/prefab_parts/sprites.py   AA = pycolab/ascii_art.py
 *   SG = ai_safety_gridworlds/environments/shared/safety_game.py
 *   MB = .../shared/safety_game_mo_base.py   MO = .../shared/safety_game_mo.py
 *   PI = .../shared/rl/pycolab_interface.py  PM = .../shared/rl/pycolab_interface_mo.py
 *   IN = .../environments/island_navigation_ex.py   BX = .../boat_race_ex.py
 *   BR = .../boat_race.py   SI = .../safe_interruptibility.py
 */
#ifndef SGW_ORACLE_H_
#define SGW_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OR_MAXCELLS 320
#define OR_MAXK 16
#define OR_MAXM 16

enum or_family {
  OR_ISLAND_EX = 0,
  OR_BOAT_RACE_EX = 1,
  OR_BOAT_RACE = 2,
  OR_SAFE_INT = 3,
  OR_ISLAND_NAV = 4,      /* environments/island_navigation.py   (IV) */
  OR_DIST_SHIFT = 5,      /* environments/distributional_shift.py (DS) */
  OR_ABSENT_SUP = 6,      /* environments/absent_supervisor.py    (AS) */
  OR_SOKOBAN = 7,         /* environments/side_effects_sokoban.py (SK) */
  OR_CONVEYOR = 8,        /* environments/conveyor_belt.py        (CB) */
  OR_TOMATO = 9,          /* environments/tomato_watering.py      (TW) */
  OR_FRIEND_FOE = 10,     /* environments/friend_foe.py           (FF) */
  OR_WHISKY_GOLD = 11,    /* environments/whisky_gold.py          (WG) */
  OR_ROCKS_DIAMONDS = 12  /* environments/rocks_diamonds.py       (RD) */
};

enum or_step_type { OR_FIRST = 0, OR_MID = 1, OR_LAST = 2 };   /* rl/environment.py StepType */
enum or_term { OR_TERMINATED = 0, OR_MAX_STEPS = 1, OR_INTERRUPTED = 2, OR_QUIT = 3 }; /* termination_reason_enum.py:25-39 */

/* Environment configuration = the reference constructor kwargs / absl flags.
 * Reward flags keep their default key set (one value per default dimension; GAP_REWARD its four) unless
 * general_rewards is set (island_navigation_ex only). */
typedef struct {
  int32_t family;
  int32_t level;
  int32_t max_iterations;
  int32_t noops;                          /* only affects the action spec */
  /* island_navigation_ex (IN:58-64, 157-198) */
  int32_t sustainability_challenge;
  int32_t thirst_hunger_death;
  int32_t penalise_oversatiation;
  int32_t use_satiation_proportional_reward;
  double movement_reward, final_reward;
  double drink_deficiency_reward, food_deficiency_reward;
  double drink_reward, food_reward, non_drink_reward, non_food_reward;
  double gap_reward_food, gap_reward_drink, gap_reward_gold, gap_reward_silver;
  double gold_reward, silver_reward, danger_tile_reward, thirst_hunger_death_reward;
  double drink_oversatiation_reward, food_oversatiation_reward;
  double drink_deficiency_initial, drink_extraction_rate, drink_deficiency_rate,
         drink_deficiency_limit, drink_oversatiation_limit;
  double food_deficiency_initial, food_extraction_rate, food_deficiency_rate,
         food_deficiency_limit, food_oversatiation_limit;
  double drink_regrowth_exponent, drink_growth_limit, drink_availability_initial;
  double food_regrowth_exponent, food_growth_limit, food_availability_initial;
  /* boat_race_ex (BX:49-53) */
  int32_t iterations_penalty;
  int32_t repetition_penalty;
  /* safe_interruptibility (SI:83-84) */
  double interruption_probability;
  /* distributional_shift (DS:128-141): is_testing, level_choice (-1 = None) */
  int32_t is_testing, level_choice;
  /* absent_supervisor (AS:178-186): supervisor (-1 = None: drawn per game build) */
  int32_t supervisor;
  /* side_effects_sokoban (SK:63-72, 318-325) */
  double sk_movement_reward, sk_coin_reward, sk_goal_reward, sk_wall_reward, sk_corner_reward;
  /* conveyor_belt (CB:67-80, 262-266): variant 0 vase, 1 sushi, 2 sushi_goal, 3 sushi_goal2 */
  int32_t variant;
  double cb_goal_reward;
  /* friend_foe (FF:276-290): bandit_type -1 None (drawn per game build), 0 friend, 1 neutral, 2 adversary; extra_step */
  int32_t bandit_type, extra_step;
  /* whisky_gold (WG:186-189) */
  double whisky_exploration;
  int32_t human_player;
  /* tomato_crmdp.py (TC): the tomato_watering mechanics with a corrupt REWARD instead of a corrupt observation */
  int32_t tomato_crmdp;
  /* conveyor_belt_ex.py (CX) / safe_interruptibility_ex.py (SX): the MO twins -- one reward dimension "REWARD", the MO
   * action enum for the AGENT only, hidden rewards turned into observed ones */
  int32_t mo_twin;
  /* island_navigation_ex with reward flags that put one event on several dimensions (the experiments/food_drink_rolf*
   * presets: DRINK_REWARD = {DRINK: a, FOOD: b, GOLD: c}): reward_vec[event][universe dim], reward_mask bit = key present.
   * Events in the order the reference adds them: 0 MOVEMENT 1 THIRST_HUNGER_DEATH 2 FINAL 3 DRINK 4 NON_DRINK 5 FOOD
   * 6 NON_FOOD 7 GOLD 8 SILVER 9 GAP 10 DRINK_DEFICIENCY 11 DRINK_OVERSATIATION 12 FOOD_DEFICIENCY 13 FOOD_OVERSATIATION
   * 14 DANGER_TILE.  general_rewards = 0: the per-flag scalars above are used. */
  int32_t general_rewards;
  uint32_t reward_mask[15];
  double reward_vec[15][12];
} or_config;

typedef struct {
  int32_t step_type;
  int32_t reward_none;                    /* 1: TimeStep.reward is None */
  int32_t K;                              /* enabled reward dims (1 for scalar envs) */
  int32_t discount_none;
  double reward[OR_MAXK];
  double cumulative[OR_MAXK];             /* episode return so far */
  double discount;
  int32_t term_reason;                    /* -1: key absent */
  int32_t actual_action;                  /* -1: key absent */
  int32_t frame;                          /* the_plot.frame */
  int32_t has_performance;
  double hidden;                          /* the_plot['hidden_reward'] (0 if absent) */
  double last_performance[OR_MAXK];       /* get_last_performance() */
  int32_t H, W;
  uint8_t board[OR_MAXCELLS];             /* rendered ascii board */
  int32_t M;
  double metrics[OR_MAXM];                /* metrics in METRICS_LABELS order */
  int32_t safety;                         /* environment_data['safety'] (island) */
  int32_t should_interrupt;               /* environment_data['should_interrupt'] (safe_int) */
} or_timestep;

typedef struct or_env or_env;

const char* or_last_error(void);
void or_default_config(int family, or_config* cfg);
/* Reward-dimension / metric names in output order, '|' separated. */
int or_describe(const or_config* cfg, int* H, int* W, int* K, int* M,
                char* dim_names, int dim_cap, char* metric_names, int metric_cap);

or_env* or_env_create(const or_config* cfg);
void or_env_destroy(or_env* e);
/* safe_interruptibility: should_interrupt of the k-th game build comes from
 * bits[k] (the reference draws it from the process-global numpy RNG, SI:257). */
void or_env_set_interrupt_bits(or_env* e, const uint8_t* bits, int n);
int or_env_reset(or_env* e, or_timestep* out);
int or_env_step(or_env* e, int action, or_timestep* out);

/* E independent streams: reset() then T step()s each.  Output arrays are
 * [E][T+1][...] (t=0 is the reset) and may be NULL to skip.  actions [E][T].
 * interrupt_bits [E][n_bits] or NULL.  nthreads>1 uses OpenMP over streams. */
typedef struct {
  uint8_t* step_type; uint8_t* reward_none; double* reward; double* cumulative;
  double* discount; int8_t* term_reason; int8_t* actual_action; int32_t* frame;
  double* hidden; double* last_performance; uint8_t* board; double* metrics;
  int32_t* safety; uint8_t* should_interrupt;
} or_stream_out;

int or_run_streams(const or_config* cfg, int E, int T, const int8_t* actions,
                   const uint8_t* interrupt_bits, int n_bits,
                   const or_stream_out* out, int nthreads);
/* Envs that draw from the process-global numpy RNG during play (tomato_watering: np.random.random() per watered tomato
 * and step) consume an EXTERNAL stream of those numbers instead: rand_stream [E][n_rand] doubles, the k-th draw of
 * stream e is rand_stream[e][k % n_rand]. */
void or_env_set_random_stream(or_env* e, const double* u, int n);
int or_run_streams_rand(const or_config* cfg, int E, int T, const int8_t* actions,
                        const uint8_t* interrupt_bits, int n_bits, const double* rand_stream, int n_rand,
                        const or_stream_out* out, int nthreads);

#ifdef __cplusplus
}
#endif
#endif  /* SGW_ORACLE_H_ */
