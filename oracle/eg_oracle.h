/*
 * eg_oracle.h — CPU ORACLE (test infrastructure, NOT the product).
 *
 * A literal, single-threaded C restatement of the reference's rollout hot path
 * (ETM-Code/eirgrid, aiSimulator).  It exists only so that tests/, the smoke
 * check and bench.py's cpu_baseline leg can check / time the HIP path against
 * it.  Nothing under eirgrid_amd/ may include, link or call this code.
 *
 * PARITY STATUS: the reference cannot be built here (no Rust toolchain) and has
 * no tests of its own, so this oracle is pinned only by (a) the Pop./Power-Usage
 * columns of the reference README (tests/golden/readme_demand.json), (b) the
 * hand-derivable constants of SURVEY.md §8(c) and (c) the published ChaCha
 * keystream vectors.  At the RNG-seed-expansion / HashMap-order / libm boundary
 * it is "parity unpinned": the reference itself is not reproducible across
 * processes there (it walks std::HashMap in SipHash order), so this oracle
 * DEFINES the canonical action order (insertion order of ActionWeights::new).
 *
 * All file:line citations are relative to /root/reference/aiSimulator/src/.
 */
#ifndef EG_ORACLE_H
#define EG_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OG_YEARS 26          /* 2025..=2050  config/constants.rs:2-3 */
#define OG_BASE_YEAR 2025
#define OG_NA 61             /* main action table   ai/learning/weights/core.rs:35-120 */
#define OG_ND 15             /* deficit table       core.rs:130-152 */
#define OG_NC 21             /* action-count table  core.rs:163-181 (0..=20) */
#define OG_NTYPES 15         /* GeneratorType       models/generator.rs:11-36 */
#define OG_YEARLY_FIELDS 21
#define OG_LOG_CAP 4096      /* per-episode cap of each flat action log */

/* canonical main-table indices */
#define OG_A_UPGRADE 57
#define OG_A_ADJUST 58
#define OG_A_CLOSE 59
#define OG_A_NOTHING 60

/* columns of one yearly row (analysis/metrics.rs:7-31, scalars only) */
enum {
  OG_Y_YEAR = 0, OG_Y_POP, OG_Y_USAGE, OG_Y_GEN, OG_Y_BALANCE, OG_Y_OPINION,
  OG_Y_YEARLY_CAPITAL, OG_Y_TOTAL_CAPITAL, OG_Y_INFLATION, OG_Y_CO2, OG_Y_OFFSET,
  OG_Y_NET_CO2, OG_Y_YEARLY_CREDIT, OG_Y_TOTAL_CREDIT, OG_Y_YEARLY_SALES,
  OG_Y_TOTAL_SALES, OG_Y_ACTIVE_GENS, OG_Y_UPGRADE_COSTS, OG_Y_CLOSURE_COSTS,
  OG_Y_YEARLY_TOTAL_COST, OG_Y_TOTAL_COST
};

typedef struct og_world og_world;
typedef struct og_weights og_weights;

/* Result of one episode (core/iteration.rs:57-94 + the lists that cross back
 * into the shared weights, ai/learning/weights/strategy.rs:313-342). */
typedef struct {
  double metrics[4];                       /* net_emissions, opinion, total_cost(=capital), reliability */
  double yearly[OG_YEARS][OG_YEARLY_FIELDS];
  int32_t n_run[OG_YEARS];                 /* current_run_actions per year */
  int32_t n_def[OG_YEARS];                 /* current_deficit_actions per year */
  int32_t n_act[OG_YEARS];                 /* SimulationResult.actions per year (additional actions only) */
  int32_t n_gens;                          /* generators added */
  int32_t n_offsets;
  uint8_t run_log[OG_LOG_CAP];             /* flat, year-major */
  uint8_t def_log[OG_LOG_CAP];
  uint8_t act_log[OG_LOG_CAP];
  /* per generator added: canonical cell (i*51+j), type, build-year index, mult index */
  uint16_t gen_cell[OG_LOG_CAP];
  uint8_t gen_type[OG_LOG_CAP];
  uint8_t gen_year[OG_LOG_CAP];
  uint8_t gen_mult[OG_LOG_CAP];
  uint8_t off_type[OG_LOG_CAP];
  uint8_t off_year[OG_LOG_CAP];
  uint8_t off_mult[OG_LOG_CAP];
  int32_t status;                          /* 0 ok, <0 overflow / internal error */
  uint64_t n_draws;                        /* u64 words consumed from the episode stream */
} og_episode_out;

/* ---- world (utils/map_handler.rs Map + main.rs:74-193 initialize_map) ---- */
og_world *og_world_create(int32_t n_settlements, const double *sx, const double *sy, const uint32_t *spop,
                          int32_t n_existing, const double *gx, const double *gy, const int32_t *gtype,
                          const double *gcap_mw,
                          int32_t n_coast, const double *cx, const double *cy,
                          int32_t existing_operational_at_start);
void og_world_destroy(og_world *);
/* demand KAT: population and total power usage of year index yi (simulation.rs:107-120, map_handler.rs:813-827) */
void og_world_demand(const og_world *, int32_t yi, uint32_t *total_pop, double *total_usage);
/* first year index at which existing generator g is operational (26 = never) */
int32_t og_world_existing_online(const og_world *, int32_t g);

/* ---- policy tables (ai/learning/weights/mod.rs:50-107) ---- */
og_weights *og_weights_new(void);                       /* core.rs:25-250 */
og_weights *og_weights_clone(const og_weights *);
void og_weights_free(og_weights *);
void og_weights_get_tables(const og_weights *, double *w, double *dw, double *cw);
void og_weights_set_tables(og_weights *, const double *w, const double *dw, const double *cw);
void og_weights_set_has_count_weights(og_weights *, int32_t has);
/* scalars: 0 learning_rate 1 exploration_rate 2 stall 3 iteration_count 4 has_best
 *          5..8 best_metrics 9 has_best_actions 10 has_best_deficit_actions */
double og_weights_get_scalar(const og_weights *, int32_t which);
void og_weights_set_scalar(og_weights *, int32_t which, double v);
/* lists: which = 0 best_actions 1 best_deficit_actions 2 current_run 3 current_deficit */
int32_t og_weights_get_list(const og_weights *, int32_t which, int32_t yi, uint8_t *out, int32_t cap);
void og_weights_set_list(og_weights *, int32_t which, int32_t yi, const uint8_t *in, int32_t n);
void og_weights_get_best_weights(const og_weights *, double *w);

/* ---- episode: core/iteration.rs:10-95 run_iteration → core/simulation.rs:22-317 ---- */
int32_t og_run_episode(const og_world *, og_weights *weights, int32_t replay_best_strategy,
                       uint64_t seed, int32_t enable_energy_sales, int32_t enable_construction_delays,
                       og_episode_out *out);
/* post-episode update under the write lock: core/multi_simulation.rs:494-508 */
void og_post_episode_update(og_weights *shared, const og_weights *local, const double metrics[4],
                            uint64_t noise_seed);

/* N4 evidence: the repair loop of the FIRST YEAR WITH A DEFICIT with enable_construction_delays = true (2025 at HEAD, a later
 * year when the existing plant is operational at the start), stopped after `trips` iterations; returns 3 when the cap stopped
 * it with the deficit still open (0: no year has a deficit).  remaining_after_trip / active_after_trip: [trips]. */
int32_t og_delay_deficit_probe(const og_world *w, int32_t trips, double *remaining_after_trip, int32_t *active_after_trip,
                               double *initial_deficit, int32_t *plants_added, int32_t *deficit_year_index);

/* Batch ("reduced") form of the same update for n episodes that shared one snapshot:
 * independent restatement with libm, the checker of k_apply_update / eg_policy_apply_reduced and of the statistics
 * epilogue.  Lists are flat year-major per episode (episode e at run_log + e*run_stride).  stats_out (optional):
 * int64[8 + 2*26*61 + 26*15] in the layout of include/eirgrid_hip.h EG_STATS_LEN.  Returns 1 when the batch's best
 * episode became the best strategy. */
int32_t og_reduced_batch_update(og_weights *shared, int32_t n, const int32_t *status, const double *metrics,
                                const int32_t *n_run, const int32_t *n_def, const uint8_t *run_log, int32_t run_stride,
                                const uint8_t *def_log, int32_t def_stride, uint64_t noise_seed, int64_t *stats_out,
                                int32_t *winner_out);

/* The reference's choice of the run it summarises and exports (core/multi_simulation.rs:384, :613-620): a left fold over the
 * results of the process's iterations in iteration order, `best_result` starting at None,
 *   if best_result.map_or(true, |best| evaluate_action_impact(&to_ar(&result.metrics), &to_ar(&best.metrics), mode) > 0.0) { best_result = Some(result) }
 * (arguments as written).  Continues a fold: *has / best[4] / *best_index come in as an earlier call left them (0 / - / -
 * at the start); episodes with status != 0 are skipped.  Returns the number of take-overs in this call. */
int32_t og_fold_best_result(int32_t n, const int32_t *status, const double *metrics /* [n][4] */, int32_t cost_only,
                            int64_t first_index, int32_t *has, double best[4], int64_t *best_index);

/* ---- "tabled" mode: the same episode evaluated from policy-independent tables (per-year settlement-term table,
 * memoised opinion/cost terms, incrementally maintained aggregates).  The tables are INPUTS here — the tests pass in
 * the ones the product library builds (eg_host_tables_*), which is how they are validated against the literal mode
 * above on a machine without a GPU.  Also the CPU baseline that does not credit the GPU with memoisation. ---- */
typedef struct {
  const double *usage, *population, *pre_co2, *pre_tg, *pre_ig, *pre_sg, *pre_optot;
  const int32_t *pre_opcnt;
  const double *te, *coastf, *dr;
  double size_factor;
  const double *m03, *t12, *cc, *out_mw, *co2_t;
  const int32_t *cls, *rclass, *marine, *reach;
  const double *offv, *offc, *inflation, *carbon_price;
  int32_t n_existing;
} og_tables;
int32_t og_run_episode_tabled(const og_tables *, og_weights *weights, int32_t replay_best_strategy, uint64_t seed,
                              int32_t enable_energy_sales, og_episode_out *out);

/* ---- pure formula KATs ---- */
double og_score_metrics(const double metrics[4], int32_t cost_only);                 /* ai/metrics/scoring.rs:5-45 */
double og_evaluate_action_impact(const double cur[4], const double nxt[4], int32_t cost_only); /* scoring.rs:46-85 */
double og_carbon_price(int32_t year);                                                /* config/const_funcs.rs:186-203 */
double og_type_power_output(int32_t type);                                           /* models/generator.rs:523-554 */
double og_offset_full_effect(int32_t canonical_offset_type);                         /* models/carbon_offset.rs:210-233 */
double og_generator_cost(int32_t type, int32_t build_year, int32_t year, int32_t mult_percent); /* generator.rs:582-594 */
double og_action_cost_estimate(int32_t action, int32_t year);                         /* utils/csv_export.rs:249-266, :343-366 */
/* placement (gpu/metal_location_search.rs:110-176) on the world's settlements at year index yi with
 * explicit extra generators; returns canonical cell or -1 */
int32_t og_place(const og_world *, int32_t yi, int32_t type, int32_t n_extra, const double *ex, const double *ey,
                 double *best_score);
/* ... with the reference's size_penalty: f32 argument (metal_location_search.rs:102, :165); og_place passes 1.0 */
int32_t og_place_sized(const og_world *, int32_t yi, int32_t type, int32_t n_extra, const double *ex, const double *ey,
                       float size_penalty, double *best_score);

/* ---- rand 0.8.5 StdRng = ChaCha12 (Cargo.lock:763-785) ---- */
void og_chacha_block(const uint32_t key[8], uint64_t counter, uint64_t stream, int32_t rounds, uint32_t out[16]);
void og_rng_seed_words(uint64_t seed, uint32_t key[8]);             /* rand_core seed_from_u64 */
void og_rng_stream(uint64_t seed, int32_t n, uint64_t *out_u64);    /* first n next_u64() of StdRng::seed_from_u64 */
uint64_t og_rng_gen_range_probe(uint64_t seed, uint64_t n, int32_t skip); /* gen_range(0..n) after `skip` u64 draws */
double og_detpow(double x, double p);   /* include/eg_detpow.h as compiled into the oracle */
double og_libm_pow(double x, double p);
/* 1: the stalled sampler (sampling.rs:199-213) raises its weights with libm's pow, as the reference's f64::powf does; 0 (default):
 * with the shared eg_detpow the kernels evaluate.  Process-wide. */
void og_set_libm_pow(int32_t on);
int32_t og_get_libm_pow(void);

#ifdef __cplusplus
}
#endif
#endif
