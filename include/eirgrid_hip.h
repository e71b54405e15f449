/*
 * eirgrid_hip.h — C ABI of the MI355X-native rollout engine (libeirgrid_hip.so).
 *
 * Drop-in boundary for the hot path of ETM-Code/eirgrid's aiSimulator.  The reference has no FFI of its own;
 * each entry point below names the Rust item it replaces (paths relative to aiSimulator/src/):
 *
 *   eg_create / eg_destroy      Map::new + initialize_map              utils/map_handler.rs:351-399, main.rs:74-193
 *   eg_rollout_batch            run_iteration, batched over episodes   core/iteration.rs:10-20 (callers:
 *                                                                       core/multi_simulation.rs:472, :690)
 *   eg_find_suitable_location   MetalLocationSearch::find_suitable_location   gpu/metal_location_search.rs:96-103
 *   eg_place                    the same search as the rollout kernels run it (generators on the 1 km grid), for parity tests
 *   eg_policy_*                 ActionWeights::{new, update_*, apply_*}        ai/learning/weights/ (all files)
 *   eg_policy_apply_episode     the write-locked section               core/multi_simulation.rs:494-508
 *   eg_train_step, eg_policy_push / eg_device_step / eg_policy_pull    the body of the training loop (one batch of
 *                               iterations + the update), on the host or entirely on the device   multi_simulation.rs:425-508
 *   eg_policy_save_json / load_json / append_weight_history / export_improvement_csv
 *                               checkpoints and run-directory files    ai/learning/weights/serialization.rs,
 *                                                                       multi_simulation.rs:166-207, utils/csv_export.rs:155-207
 *
 * Conventions: plain pointers and sizes, caller-allocated host buffers unless a parameter is named d_* (device
 * pointer).  Every function returning int32_t returns EG_OK (0) or a negative EG_ERR_* code; eg_last_error()
 * gives the text.  One eg_ctx per device; a ctx is not thread-safe, independent ctxs may be used from separate
 * host threads.  The library fails loudly (EG_ERR_NO_DEVICE) when no HIP device is present: there is no CPU
 * fallback behind this ABI.
 *
 * Canonical action indices (the reference walks std::HashMap in hash order; this ABI fixes the insertion order
 * of ActionWeights::new, ai/learning/weights/core.rs:35-120):
 *   0..44   AddGenerator(type = idx/3 in models/generator.rs:11-36 order, cost multiplier {100,120,150}[idx%3])
 *   45..56  AddCarbonOffset({Forest, Wetland, ActiveCapture, CarbonCredit}[(idx-45)/3], {100,120,150}[(idx-45)%3])
 *   57 UpgradeEfficiency("")   58 AdjustOperation("",0)   59 CloseGenerator("")   60 DoNothing
 * Deficit table (core.rs:130-152): GasPeaker, GasCombinedCycle, BatteryStorage, PumpedStorage, Biomass,
 *   OnshoreWind, OffshoreWind, UtilitySolar, HydroDam, Nuclear, DomesticSolar, CommercialSolar, TidalGenerator,
 *   WaveEnergy, DoNothing.
 */
#ifndef EIRGRID_HIP_H
#define EIRGRID_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EG_YEARS 26
#define EG_N_ACTIONS 61
#define EG_N_DEFICIT 15
#define EG_N_COUNTS 21
#define EG_N_TYPES 15
#define EG_GRID 51
#define EG_CELLS (EG_GRID * EG_GRID)
#define EG_YEARLY_FIELDS 21

/* Per-episode capacities of the records (an episode that exceeds one ends with status EG_EP_OVERFLOW).  The reference's lists
 * are Vecs (core/simulation.rs:146-162, :406-409; ai/learning/weights/sampling.rs:93-101, :258-266); a replay episode records and
 * applies every action twice, so the lists of a training loop double with every replay episode that becomes the
 * best strategy.  Episodes that replay a long best list (more than 96 actions) run a kernel variant whose lists continue in the
 * episode's own record beyond the on-chip window, up to these capacities — the same 4096 the CPU oracle stops at
 * (oracle/eg_oracle.h OG_LOG_CAP).  Sampled episodes and replays of a short list stay within the on-chip window of
 * EG_ONCHIP_GENS generators / offsets (they place 25-200). */
#define EG_MAX_GENS 4096
#define EG_MAX_OFFSETS 4096
#define EG_RUN_CAP 4096
#define EG_DEF_CAP 4096
#define EG_ACT_CAP 4096
#define EG_ONCHIP_GENS 512

#define EG_OK 0
#define EG_ERR_NO_DEVICE (-1)
#define EG_ERR_BAD_ARG (-2)
#define EG_ERR_HIP (-3)
#define EG_ERR_UNSUPPORTED (-4)
#define EG_ERR_NOMEM (-5)
#define EG_ERR_INTERNAL (-6)

#define EG_EP_OK 0
#define EG_EP_OVERFLOW (-1)
#define EG_EP_NO_LOCATION (-2)
#define EG_EP_INTERNAL (-3)      /* the kernel's helper-wave protocol timed out (a defect, never an input property): eg_fetch* report EG_ERR_INTERNAL */

/* yearly row columns: the scalar fields of YearlyMetrics, analysis/metrics.rs:7-31 */
enum {
  EG_Y_YEAR = 0, EG_Y_POP, EG_Y_USAGE, EG_Y_GEN, EG_Y_BALANCE, EG_Y_OPINION, EG_Y_YEARLY_CAPITAL,
  EG_Y_TOTAL_CAPITAL, EG_Y_INFLATION, EG_Y_CO2, EG_Y_OFFSET, EG_Y_NET_CO2, EG_Y_YEARLY_CREDIT, EG_Y_TOTAL_CREDIT,
  EG_Y_YEARLY_SALES, EG_Y_TOTAL_SALES, EG_Y_ACTIVE_GENS, EG_Y_UPGRADE_COSTS, EG_Y_CLOSURE_COSTS,
  EG_Y_YEARLY_TOTAL_COST, EG_Y_TOTAL_COST
};

typedef struct eg_ctx eg_ctx;
typedef struct eg_policy eg_policy;

/* What initialize_map loads (main.rs:74-193): settlements.json, ireland_generators.csv, coastline_points.json. */
typedef struct {
  int32_t n_settlements;
  const double *settlement_x, *settlement_y;   /* grid metres, clamped to [0, 50000] like data/poi.rs:11-15 */
  const uint32_t *settlement_pop;              /* 2025 population */
  int32_t n_existing;
  const double *existing_x, *existing_y;
  const int32_t *existing_type;                /* generator type index */
  const double *existing_capacity_mw;          /* CSV capacity_mw (data/generators_loader.rs:150-153) */
  int32_t n_coast;
  const double *coast_x, *coast_y;
  int32_t existing_operational_at_start;       /* 0 = reference HEAD: existing plant starts "Planned" in 2024 */
} eg_world;

/* Flags that reach run_simulation (core/simulation.rs:22-31; cli/cli.rs:42-55). */
typedef struct {
  int32_t enable_energy_sales;          /* CLI default true */
  int32_t enable_construction_delays;   /* CLI default false; true is not implemented on the device yet */
  int32_t write_yearly;                 /* 1: fill eg_episode_out.yearly */
} eg_opts;

/* Read-only view of ActionWeights for one batch (ai/learning/weights/mod.rs:50-107). */
typedef struct {
  const double *weights;          /* [26][61] */
  const double *deficit_weights;  /* [26][15] */
  const double *count_weights;    /* [26][21], NULL = absent (dropped by the checkpoint loader) */
  double learning_rate, exploration_rate;
  uint32_t iterations_without_improvement;
  int32_t has_best;
  double best_metrics[4];         /* final_net_emissions, average_public_opinion, total_cost, power_reliability */
  /* replay data (best_actions / best_deficit_actions), flat year-major; NULL when has_best == 0 */
  const int32_t *best_count;          /* [26] */
  const uint8_t *best_actions;        /* sum(best_count) */
  const int32_t *best_deficit_count;  /* [26] */
  const uint8_t *best_deficit_actions;
} eg_policy_snapshot;

/* Host-side result buffers, episode-major.  Any pointer may be NULL (that output is skipped). */
typedef struct {
  double *metrics;      /* [n][4]  SimulationMetrics, core/iteration.rs:69-74 */
  double *yearly;       /* [n][26][21] */
  int32_t *status;      /* [n] EG_EP_* */
  int32_t *n_run;       /* [n][26] current_run_actions per year */
  int32_t *n_def;       /* [n][26] current_deficit_actions per year */
  int32_t *n_act;       /* [n][26] SimulationResult.actions per year */
  uint8_t *run_log;     /* [n][EG_RUN_CAP] flat, year-major */
  uint8_t *def_log;     /* [n][EG_DEF_CAP] */
  uint8_t *act_log;     /* [n][EG_ACT_CAP] */
  int32_t *n_gens;      /* [n] */
  uint16_t *gen_cell;   /* [n][EG_MAX_GENS] placement result: i*51+j on the 1 km grid */
  uint16_t *gen_pack;   /* [n][EG_MAX_GENS] type | build-year index << 4 | multiplier index << 9 */
  int32_t *n_offsets;   /* [n] */
  uint16_t *off_pack;   /* [n][EG_MAX_OFFSETS] offset type | year index << 4 | multiplier index << 9 */
  uint64_t *n_draws;    /* [n] words consumed from the episode stream */
  double *bytes_moved;  /* [n] algorithmic bytes of the episode, SURVEY.md §8(d) formula (bills the whole 2601 x 8 B score field per search) */
  uint32_t *n_chunks;   /* [n] what the episode's placement searches requested from memory, in units of 2 KB: chunks of 64 sorted
                           candidate records (32 B each) — what the branch-and-bound search reads instead of the score field —
                           and, for long (replay) episodes, the entries of their penalty field (eg_rollout.hip place_heavy);
                           bytes touched = bytes_moved - n_gens * 2601 * 8 + n_chunks * 2048 */
} eg_episode_out;

const char *eg_last_error(void);
/* 16 hex digits: sha256 over the sources this library was built from (csrc/Makefile); eirgrid_amd/_native.py compares it
 * with the tree it runs in, so that a stale binary next to edited sources is an import error, not a silent mismatch */
const char *eg_build_hash(void);
int32_t eg_device_count(void);

eg_ctx *eg_create(int32_t device_ordinal, const eg_world *world);
void eg_destroy(eg_ctx *);

/* Run episodes [first_episode_index, first_episode_index + n) against one snapshot.  Episode e draws from
 * StdRng::seed_from_u64(seed + e) (the reference gives every episode the same stream under --seed,
 * core/simulation.rs:50-53; e = 0 reproduces that).  replay_mask[i] != 0 runs episode i with
 * replay_best_strategy = true (core/multi_simulation.rs:461-465).  Results are copied into `out`: the scalar fields and the per-year
 * counts of every episode, and of each list (run_log, def_log, act_log, gen_cell, gen_pack, off_pack) the entries the counts
 * announce — a row's bytes behind the longest list of the batch are not written (they keep what the caller's buffer held). */
int32_t eg_rollout_batch(eg_ctx *, const eg_policy_snapshot *, const eg_opts *, uint64_t seed,
                         uint64_t first_episode_index, uint32_t n_episodes, const uint8_t *replay_mask,
                         eg_episode_out *out);

/* Device-resident variant for the timed path: upload the snapshot once, launch any number of batches (results
 * stay in HBM), then fetch the last batch. */
int32_t eg_upload_snapshot(eg_ctx *, const eg_policy_snapshot *, const eg_opts *);
int32_t eg_rollout_launch(eg_ctx *, uint64_t seed, uint64_t first_episode_index, uint32_t n_episodes,
                          const uint8_t *replay_mask /* host, may be NULL */);
int32_t eg_sync(eg_ctx *);
/* copies the eg_last_batch_size() records of the last launched batch: every non-NULL field of `out` must hold that many */
uint32_t eg_last_batch_size(const eg_ctx *);
int32_t eg_fetch(eg_ctx *, eg_episode_out *out);
/* HIP-event time of the rollout kernel launches since the last call to eg_timing_reset (milliseconds, count). */
int32_t eg_timing_reset(eg_ctx *);
int32_t eg_timing_read(eg_ctx *, double *total_ms, int32_t *n_launches);
/* The same launches seen grid by grid.  A batch with replay episodes is up to three grids on two streams (the replay variants
 * on a side stream, the rest on the null stream): *span_ms = eg_timing_read's total (first start to last end of every batch),
 * *grids_ms = the replay grids' and the lean grid's own durations added up.  Grids that run side by side: span well below the
 * sum; grids that were serialised (e.g. two streams sharing one hardware queue): span == sum. */
int32_t eg_timing_read_grids(eg_ctx *, double *span_ms, double *grids_ms, int32_t *n_launches);
/* Device memory the context holds beyond the 50 KB policy: the world's tables (tab::total, 42 MB: 26 MB of sorted candidate lists, 11.5 MB their compact form, 3.3 MB
 * the per-cell placement prefix the hoisted replay reads), the episode records (one of 41.6 KB per
 * episode of the largest batch so far) and the penalty-field pool of long replay episodes — 126 KB per replay episode of a launch,
 * allocated with the first replay launch unless the host knows the best list to be short (<= 96 actions: it uploaded, rewound or
 * pulled it and no on-device update is in flight), grown to the largest launch since, never beyond
 * EIRGRID_HEAVY_POOL_GB (environment, default 64; an episode without a slot takes the exact scan: slower, same result). */
int32_t eg_memory_report(const eg_ctx *, uint64_t *table_bytes, uint64_t *record_bytes, uint64_t *field_pool_bytes);
/* Batch ("reduced") form of the write-locked update (core/multi_simulation.rs:494-508).
 * eg_update_stats reduces the last launched batch on the device into d_stats (EG_STATS_LEN int64, DEVICE pointer, e.g.
 * a torch tensor): integer sums that do not depend on episode / workgroup / rank order, so ONE sum all-reduce over
 * RCCL is the whole exchange; eg_policy_apply_reduced then applies them on the host.  Layout:
 *   [0] episodes ok  [1] episodes failed  [2] episodes that qualify for contrast learning (learning.rs:160)
 *   [3] NOT a sum: the batch's best score as a sortable integer (bit pattern of the score + 1, 0 = none), a maximum;
 *       used on one GPU to find the best episode inside the update kernel, ignored by every update formula
 *   [8 + y*61 + a]               sum of Q32 ln(penalty_factor), learning.rs:232-239
 *   [8 + 26*61 + y*61 + a]       sum of Q32 ln(mild_penalty),   learning.rs:241-251
 *   [8 + 2*26*61 + y*15 + slot]  deficit actions absent from best_deficit_actions[y], learning.rs:346-352 */
#define EG_STATS_LEN (8 + 2 * EG_YEARS * EG_N_ACTIONS + EG_YEARS * EG_N_DEFICIT)
int32_t eg_update_stats(eg_ctx *, int64_t *d_stats);
/* The timed path fuses all of it: one launch runs the batch, accumulates the statistics in the kernel's epilogue and
 * picks the batch's best episode (highest score, ties to the lowest global index).  `d_packet` (DEVICE pointer,
 * EG_PACKET_BYTES) = int64 stats[EG_STATS_LEN] followed by the candidate record:
 *   f64 score (-1: none) | i64 global index | f64 metrics[4] | i32 n_run[26] | i32 n_def[26] | u8 run_log[EG_RUN_CAP] |
 *   u8 def_log[EG_DEF_CAP]
 * so that one device-to-host copy (after the all-reduce of the stats part when N > 1) feeds eg_policy_apply_reduced. */
#define EG_CANDIDATE_BYTES (8 + 8 + 32 + 4 * EG_YEARS + 4 * EG_YEARS + EG_RUN_CAP + EG_DEF_CAP)
#define EG_PACKET_BYTES (8 * EG_STATS_LEN + EG_CANDIDATE_BYTES)
int32_t eg_rollout_launch_update(eg_ctx *, uint64_t seed, uint64_t first_episode_index, uint32_t n_episodes,
                                 const uint8_t *replay_mask /* host, may be NULL */, void *d_packet);
/* score_metrics of every episode of the last batch (written by eg_update_stats; -1 for failed episodes) */
int32_t eg_fetch_scores(eg_ctx *, double *scores);
/* metrics and action lists of one episode of the last batch (the best-candidate broadcast of SURVEY.md §8(e)) */
int32_t eg_fetch_episode_lists(eg_ctx *, uint32_t episode, double metrics[4], int32_t *n_run, uint8_t *run_log /* EG_RUN_CAP */,
                               int32_t *n_def, uint8_t *def_log /* EG_DEF_CAP */);

/* Full record of ONE episode of the last batch (every non-NULL field of `out`, sized for n = 1). */
int32_t eg_fetch_record(eg_ctx *, uint32_t episode, eg_episode_out *out);
/* The record of the episode that is the policy's best strategy (update_best_strategy, ai/learning/weights/strategy.rs:19-258):
 * with the policy resident on the device, k_apply_update keeps that episode's record next to the policy whenever an
 * update installs a new best strategy.  *state: 0 = no improvement yet, 1 = `out` (n = 1) was filled, 2 = the best
 * episode ran on another rank (ask that rank).  Yearly rows are only meaningful when the policy was pushed with
 * eg_opts.write_yearly = 1.  (NOT the run the reference summarises and exports: that is eg_fetch_best_result below.) */
int32_t eg_fetch_best_run(eg_ctx *, eg_episode_out *out, int32_t *state);

/* The reference's `best_result` (core/multi_simulation.rs:384, :613-620): after its parallel section the reference folds the
 * results of THIS process's iterations in iteration order,
 *     if best_result.map_or(true, |best| evaluate_action_impact(&result.metrics, &best.metrics, optimization_mode) > 0.0)
 *         { best_result = Some(result) }
 * — arguments as written: `result` is the current state, `best` the new one, so a result takes over when the held run is an
 * improvement ON it — and that run is what multi_simulation.rs:821-905 prints and exports (simulation_summary.csv, the detail
 * files).  eg_best_result_track starts the fold at None (mode 1: optimization_mode None, 2: "cost_only" (--cost-only), 0: stop
 * tracking); from then on every batch launched on this context is folded on the device behind its rollout, in global index
 * order, failed episodes skipped (in the reference a failed iteration ends the run).  eg_fetch_best_result copies the held
 * run's record (n = 1; *state 0: none yet, 1: filled; *global_index may be NULL).  One fold per context = per process, as in
 * the reference; eg_evaluate_action_impact is ai/metrics/scoring.rs:46-85 on SimulationMetrics quadruples
 * (metrics_to_action_result, multi_simulation.rs:55-62). */
int32_t eg_best_result_track(eg_ctx *, int32_t mode);
int32_t eg_fetch_best_result(eg_ctx *, eg_episode_out *out, int32_t *state, int64_t *global_index);
double eg_evaluate_action_impact(const double current_metrics[4], const double new_metrics[4], int32_t cost_only);

/* Test hook: fills the LDS of every compute unit with `value` and waits.  LDS is not cleared between workgroups, so a
 * kernel that reads a word before writing it sees what the previous tenant left; the parity tests call this with small
 * integers (the values the helper protocol's sequence flags take) before a rollout. */
int32_t eg_debug_fill_lds(eg_ctx *, uint32_t value);
/* Diagnostic hook: ONE idle workgroup of 256 threads on the library's side stream that stays resident for `cycles` shader cycles;
 * variant 0: 1 KB of LDS, few registers; 1: 150 KB of LDS; 2: 200+ registers a lane; 3: both (the hoisted replay's footprint).  What a
 * resident workgroup costs the grid beside it: scripts/side_kernel_probe.py, profiles/r04_ab_notes.log. */
int32_t eg_debug_occupy(eg_ctx *, int32_t variant, uint64_t cycles);

/* B2: one placement search on the device (settlements of year index `year_index`, the ctx's existing plant plus
 * `n_extra` generators given by grid cell), for parity tests of the arg-max kernel. */
int32_t eg_place(eg_ctx *, int32_t gen_type, int32_t year_index, const uint16_t *extra_cells, int32_t n_extra /* <= EG_ONCHIP_GENS */,
                 int32_t *out_cell, double *out_score);

/* B2 with the reference's own signature — MetalLocationSearch::find_suitable_location(&self, settlements, generators,
 * coastline_points, gen_type, size_penalty: f32) -> Option<Coordinate> (gpu/metal_location_search.rs:96-103): the settlements
 * (with the populations of year `year_index`), the existing plant and the coastline are the ctx's world; `gen_x / gen_y` are
 * the further generators of the caller's map at ARBITRARY coordinates (metres), in list order.  *found = 0 is the
 * reference's None; otherwise (*out_x, *out_y) is the winning candidate's coordinate as Coordinate::new clamps it. */
int32_t eg_find_suitable_location(eg_ctx *, int32_t year_index, int32_t gen_type, const double *gen_x, const double *gen_y,
                                  int32_t n_generators, float size_penalty, double *out_x, double *out_y, int32_t *found,
                                  double *out_score /* may be NULL */);

/* ---- policy-independent host tables (built once per world; no device needed), read-only views for validation.
 * Names: usage population pre_co2 pre_tg pre_ig pre_sg pre_optot te coastf dr m03 t12 cc out_mw co2_t offv offc
 * inflation carbon_price size_factor (f64); pre_opcnt cls rclass marine reach existing_online (i32). ---- */
typedef struct eg_host_tables eg_host_tables;
eg_host_tables *eg_host_tables_create(const eg_world *world);
void eg_host_tables_free(eg_host_tables *);
int32_t eg_host_tables_f64(const eg_host_tables *, const char *name, const double **ptr, int64_t *len);
int32_t eg_host_tables_i32(const eg_host_tables *, const char *name, const int32_t **ptr, int64_t *len);

/* ---- ActionWeights on the host (C++ mirror of ai/learning/weights/) ---- */
eg_policy *eg_policy_new(void);                                     /* core.rs:25-250 */
void eg_policy_free(eg_policy *);
int32_t eg_policy_snapshot_view(const eg_policy *, eg_policy_snapshot *out);  /* pointers live as long as the policy */
int32_t eg_policy_get_tables(const eg_policy *, double *w, double *dw, double *cw);
int32_t eg_policy_set_tables(eg_policy *, const double *w, const double *dw, const double *cw);
double eg_policy_get_scalar(const eg_policy *, int32_t which);     /* same codes as the oracle: see eg_policy.cpp */
int32_t eg_policy_set_scalar(eg_policy *, int32_t which, double v);
int32_t eg_policy_get_list(const eg_policy *, int32_t which, int32_t year_index, uint8_t *out, int32_t cap);
/* multi_simulation.rs:494-508 for one episode: transfer_recorded_actions_from → apply_contrast_learning →
 * update_best_strategy → apply_deficit_contrast_learning.  run/def lists are flat year-major with counts. */
int32_t eg_policy_apply_episode(eg_policy *, const double metrics[4], const int32_t *n_run, const uint8_t *run_log,
                                const int32_t *n_def, const uint8_t *def_log, uint64_t noise_seed);
/* The same three steps for a whole batch that shared one snapshot: `stats` is the (all-reduced) host copy of the
 * eg_update_stats buffer, the candidate is the batch's best episode (highest score, ties to the lowest global index). */
int32_t eg_policy_apply_reduced(eg_policy *, const int64_t *stats, const double cand_metrics[4], const int32_t *cand_n_run,
                                const uint8_t *cand_run_log, const int32_t *cand_n_def, const uint8_t *cand_def_log,
                                uint64_t noise_seed);
/* The same update from update packets: `stats` = the (all-reduced) int64[EG_STATS_LEN] statistics, `candidates` =
 * n_candidates candidate records of EG_CANDIDATE_BYTES each (one per rank, in rank order; layout above).  The winner is
 * the record with the highest score, ties to the lowest global index (index < 0: no candidate).  Returns 1 when the
 * winner became the best strategy, 0 when not, < 0 on error. */
int32_t eg_policy_apply_packet(eg_policy *, const int64_t *stats, const void *candidates, int32_t n_candidates,
                               uint64_t noise_seed);
/* One whole pass of the batch training step on one GPU, one call: snapshot upload, rollout with the statistics
 * epilogue, best pick, ONE packet copy to pinned host memory, eg_policy_apply_packet.  Same return convention. */
int32_t eg_train_step(eg_ctx *, eg_policy *, const eg_opts *opts, uint64_t seed, uint64_t first_episode_index,
                      uint32_t n_episodes, const uint8_t *replay_mask /* host, may be NULL */, uint64_t noise_seed);
/* Device-resident policy.  eg_policy_push uploads the policy once; after that every training step runs on the device with
 * no host synchronisation: eg_device_rollout enqueues the rollout (statistics epilogue, best pick) into `d_packet`
 * (DEVICE, EG_PACKET_BYTES, zeroed once by the caller); eg_device_apply enqueues the batch update from n_packets update
 * packets laid out back to back (DEVICE; with one GPU that is d_packet itself, with N GPUs the result of ONE all-gather of
 * every rank's packet on the same stream — the statistics are integer sums, so adding them up in the kernel is the
 * all-reduce) and zeroes the statistics of `d_own_packet` for the next step.  eg_device_step does both on a
 * library-owned packet (one GPU).  The update is the one of eg_policy_apply_packet, bit for bit (both evaluate
 * csrc/eg_reduced_math.h).  replay_period > 0: the episode with global index i replays the best strategy when
 * i % replay_period == 0 and a best strategy exists (decided on the device).  eg_policy_pull waits for the stream and
 * copies the policy back into `policy` (tables, counters, the best strategy if an on-device update installed one since
 * the push; improvement-history records are appended once per context, to whichever policy pulls first). */
int32_t eg_policy_push(eg_ctx *, const eg_policy *, const eg_opts *opts);
int32_t eg_device_rollout(eg_ctx *, uint64_t seed, uint64_t first_episode_index, uint32_t n_episodes, uint32_t replay_period,
                          void *d_packet);
int32_t eg_device_apply(eg_ctx *, const void *d_packets, int32_t n_packets, void *d_own_packet, uint64_t noise_seed);
int32_t eg_device_step(eg_ctx *, uint64_t seed, uint64_t first_episode_index, uint32_t n_episodes, uint32_t replay_period,
                       uint64_t noise_seed);
int32_t eg_policy_pull(eg_ctx *, eg_policy *);
/* Measurement support (no counterpart in the reference): eg_policy_hold keeps a copy of the device-resident policy as it is
 * now — tables, best strategy, counters — on the device, stream-ordered; eg_policy_rewind puts that copy back.  bench.py
 * rewinds before every batch, so that every batch is the same work on any number of GPUs: left to itself the training loop
 * changes what a replay episode costs, differently for every global batch size. */
int32_t eg_policy_hold(eg_ctx *);
int32_t eg_policy_rewind(eg_ctx *);
/* Replay hoist (no counterpart in the reference — it runs every iteration on its own, core/multi_simulation.rs:425-472).  An episode
 * with replay_best_strategy = true takes every action from the stored lists and reads no seeded draw until a list runs out
 * (ai/learning/weights/sampling.rs:78-101, :242-266; core/simulation.rs:146-162), so all replay episodes of a batch are one and the
 * same computation.  With the hoist on (eg_replay_hoist(ctx, 1), or EIRGRID_REPLAY_HOIST=1 in the environment at eg_create; off by
 * default) a batch computes that script ONCE — a cooperative sixteen-wave workgroup, csrc/eg_replay_coop.h — and hands every replay
 * episode of the batch a copy of the record; records, statistics and update packets are those of the per-episode path, byte for byte
 * (n_chunks excepted: it counts what the search that really ran requested).  A script that needs a fallback draw
 * (sampling.rs:445-528) or would end with a status other than EG_EP_OK is not hoisted: the per-episode kernels run those episodes as
 * before.  eg_replay_hoist_stats waits for the device: *batches_armed = batches launched with the hoist on and replay episodes in
 * them, *last_batch_hoisted = 1 when the last such batch was served by the hoist. */
int32_t eg_replay_hoist(eg_ctx *, int32_t on);
int32_t eg_replay_hoist_stats(eg_ctx *, uint64_t *batches_armed, int32_t *last_batch_hoisted);
/* Diagnostic hook: cycle counts of the hoisted script's phases as a -DEG_COOP_STAMPS build of the library leaves them (zeros in the
 * shipped build): between commands, year start (own chains), waiting for the other waves' chains, command barrier, search, field update. */
int32_t eg_debug_hoist_stamps(eg_ctx *, uint64_t stamps[8]);
/* Checkpoints in the reference's JSON schema (SerializableWeights, ai/learning/serialization.rs:38-51):
 * save_to_file / load_from_file of ai/learning/weights/serialization.rs:29-493.  As in the reference the count table
 * is not part of the file; a loaded policy samples the action count with the heuristic branch (sampling.rs:423-442). */
int32_t eg_policy_save_json(const eg_policy *, const char *path);
eg_policy *eg_policy_load_json(const char *path);
/* --track-weight-history (core/multi_simulation.rs:166-207): append one snapshot {best_score, iteration, timestamp,
 * weights: ActionWeights::to_json()} to the pretty-printed JSON array in `path` (created as needed). */
int32_t eg_policy_append_weight_history(const eg_policy *, const char *path, uint64_t iteration);
/* improvement_history.csv of the reference's best-run export (utils/csv_export.rs:155-207); no file if there is no history */
int32_t eg_policy_export_improvement_csv(const eg_policy *, const char *path);   /* NULL + eg_last_error() on failure */
/* simulation_summary.csv of the best-run export (utils/csv_export.rs:215-432): final metrics, the action list with the
 * exporter's cost estimates, the yearly summary rows.  `run` holds one episode (metrics, yearly, n_act, act_log). */
int32_t eg_export_summary_csv(const eg_episode_out *run, const char *path, const char *timestamp);
/* The detail files of the same export (utils/csv_export.rs:434-1230, called from core/multi_simulation.rs:852-905):
 * <out_dir>/yearly_details/{settlements,generators,carbon_offsets}.csv and <out_dir>/operation_logs/generator_operation_logs.csv,
 * from the best episode's record (`run`: n_gens, gen_pack, n_act, act_log are read) and the world.  Host code, no device.
 * settlement_names: [n_settlements] or NULL ("Settlement_<i>").  offset_seed: carbon-offset coordinates come from thread_rng in
 * the reference (core/actions.rs:142-145); here from StdRng::seed_from_u64(offset_seed).  What the reference really writes —
 * and therefore this function — is described at the top of csrc/eg_export.cpp. */
int32_t eg_export_run_details(const eg_world *world, const char *const *settlement_names, const eg_episode_out *run,
                              const char *out_dir, uint64_t offset_seed);
double eg_score_metrics(const double metrics[4], int32_t cost_only);   /* ai/metrics/scoring.rs:5-45 */

#ifdef __cplusplus
}
#endif
#endif
