// eg_internal.h — layouts shared by the host table builder, the C ABI glue and the HIP kernels.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "eirgrid_hip.h"

namespace eg {

constexpr int kYears = EG_YEARS;
constexpr int kTypes = EG_N_TYPES;
constexpr int kCells = EG_CELLS;
constexpr int kGrid = EG_GRID;
constexpr int kRadiusClasses = 6;
constexpr int kMaxReach = 12;   // cells; the largest penalty radius is 12 km
constexpr int kMults = 3;
constexpr int kOffsetTypes = 4;
constexpr int kPsStride = 2624;   // sorted candidate lists on the device: 41 chunks of 64 (2601 -> 2624 entries)
constexpr int kDrCompact = 352;      // doubles of the compact factor table (the reference's six radii need 333)
constexpr int kLdsGens = EG_ONCHIP_GENS;      // generators / offsets of an episode that the kernels keep in LDS; the long-replay variant goes on in the episode's record
constexpr int kShortReplayMax = 96;   // actions in the best list up to which replay episodes stay on the exact scan (eg_rollout.hip, k_rollout kinds)
constexpr int kMaxVariants = 12;  // distinct (radius class, marine) pairs over the 15 types (8 for the reference's types)
constexpr int kPcStride = 48 * 64;  // entries per list of the compact form (tab::pbase / pcell): the 41 chunks, then zeros — the scan requests a round of four chunks
                                    // ahead without asking whether the list has ended

// Policy-independent tables, built once per world on the host (eg_tables.cpp) and mirrored in HBM.
// Everything a kernel needs that involves sqrt / division by data / pow / exp lives here, so device code only
// performs + - * / compare on table values and reproduces the CPU oracle bit for bit.
struct HostTables {
  // demand (simulation.rs:107-120, map_handler.rs:813-827): identical for every episode
  std::vector<double> usage;        // [26] total power usage
  std::vector<double> population;   // [26] total population (exact integers)
  // existing plant, folded in list order for each year (they precede every new generator in Map::generators)
  std::vector<double> pre_co2, pre_tg, pre_ig, pre_sg, pre_optot;  // [26]
  std::vector<int32_t> pre_opcnt;   // [26]
  // placement (metal_location_search.rs:110-176)
  std::vector<double> te;           // [26][6][2601] settlement product then existing-plant penalties, in reference order
  std::vector<double> coastf;       // [2601] 1/(1+min_p d/5000)
  std::vector<double> dr;           // [6][13][13] d/R for |di|,|dj| in cells; 1.0 where d >= R
  double size_factor;               // 1 - size_penalty*0.1
  // opinion / cost (map_handler.rs:925-965, generator.rs:582-594, const_funcs.rs:28-106)
  std::vector<double> m03;          // [2601] 0.03 * mean_s 1/(1+d/1e4)
  std::vector<double> t12;          // [26][15] 0.12 * type opinion
  std::vector<double> cc;           // [26][15][26][3][2] {cost, 0.82*cost_opinion} at year y for (type, build year, mult)
  std::vector<double> out_mw;       // [15] output of an operational new plant
  std::vector<double> co2_t;        // [15] CO2 of an operational new plant
  std::vector<int32_t> cls;         // [15] 0 dispatchable, 1 intermittent, 2 storage
  std::vector<int32_t> rclass;      // [15] radius class
  std::vector<int32_t> marine;      // [15]
  std::vector<int32_t> reach;       // [6] half-width of the penalty box in cells
  // offsets (carbon_offset.rs:188-260)
  std::vector<double> offv;         // [26][4][26] tonnes offset at year y by an offset of type ot completed in year b
  std::vector<double> offc;         // [26][4][3] cost at year y
  // yearly scalars
  std::vector<double> inflation;    // [26]
  std::vector<double> carbon_price; // [26]
  std::vector<int32_t> existing_online;  // [G0] first operational year index (26 = never)
};

void build_tables(const eg_world& w, HostTables& t);

#ifdef __HIPCC__
#define EG_HD __host__ __device__
#else
#define EG_HD
#endif

// One entry of a sorted candidate list: everything the placement search reads about a candidate, 32 bytes.
struct PsRec { double te, cf, m03; uint32_t cell, pad; };   // pad: the cell's grid coordinates times four, 4 i | 4 j << 16 (small-batch kernel)
static_assert(sizeof(PsRec) == 32, "candidate record");

// Device tables: ONE allocation, fixed layout.  A kernel receives a single base pointer and addresses every table at a
// compile-time offset, so a table access costs no kernel-argument load and no pointer registers.
namespace tab {
constexpr size_t a16(size_t x) { return (x + 15) & ~size_t(15); }
constexpr size_t usage = 0;
constexpr size_t population = usage + 8 * kYears;
constexpr size_t pre_co2 = population + 8 * kYears;
constexpr size_t pre_tg = pre_co2 + 8 * kYears;
constexpr size_t pre_ig = pre_tg + 8 * kYears;
constexpr size_t pre_sg = pre_ig + 8 * kYears;
constexpr size_t pre_optot = pre_sg + 8 * kYears;
constexpr size_t inflation = pre_optot + 8 * kYears;
constexpr size_t carbon_price = inflation + 8 * kYears;
constexpr size_t out_mw = carbon_price + 8 * kYears;
constexpr size_t co2_t = out_mw + 8 * 16;
constexpr size_t pre_opcnt = co2_t + 8 * 16;
constexpr size_t variant = a16(pre_opcnt + 4 * kYears);
constexpr size_t cls = variant + 4 * 16;
constexpr size_t rclass = cls + 4 * 16;
constexpr size_t marine = rclass + 4 * 16;
constexpr size_t reach = marine + 4 * 16;
constexpr size_t dr = a16(reach + 4 * 8);                                   // [6][13][13]
constexpr size_t m03 = a16(dr + 8 * kRadiusClasses * 169);                  // [2601]
constexpr size_t t12 = a16(m03 + 8 * kCells);                               // [26][15]
constexpr size_t offv = a16(t12 + 8 * kYears * kTypes);                     // [26][4][26]
constexpr size_t offc = a16(offv + 8 * kYears * kOffsetTypes * kYears);     // [26][4][3]
constexpr size_t cc = a16(offc + 8 * kYears * kOffsetTypes * kMults);       // [26][15][26][3][2]
constexpr size_t ps = a16(cc + 8 * size_t(kYears) * kTypes * kYears * kMults * 2);   // PsRec [26][kMaxVariants][kPsStride]
constexpr size_t hv_box = a16(ps + sizeof(PsRec) * size_t(kYears) * kMaxVariants * kPsStride);   // u32 [1024 + 16]: heavy episodes, eg_rollout.hip heavy_add
// the throughput kernels' factor table in LDS holds every radius class only up to its own radius (eg_rollout.hip load_factor_table):
// int32 {first entry of class k} [8] | {squared distance from which the factor is 1.0, class k} [8]
constexpr size_t dr_meta = a16(hv_box + 4 * (1024 + 16));
// ... and that table itself as the kernels keep it in LDS (f64 [kDrCompact]: class k's factors by squared distance at entries
// dr_meta[k] .. dr_meta[k] + dr_meta[8 + k], 1.0 everywhere else), so that an episode copies it with one load per lane and block
constexpr size_t dr_compact = a16(dr_meta + 4 * 16);
// the sorted candidates once more as what the approximate scan of the long-replay variant reads of them (eg_rollout.hip place_heavy):
// the unpenalised score (te * cf) * size_factor and the cell, three registers per chunk in flight instead of eight.  Last, so that the
// offsets of the tables the lean kernels read stay small.
constexpr size_t pbase = a16(dr_compact + 8 * size_t(kDrCompact));                     // f64 [26][kMaxVariants][kPcStride]
constexpr size_t pcell = a16(pbase + 8 * size_t(kYears) * kMaxVariants * kPcStride);     // u32 [26][kMaxVariants][kPcStride]
// the field update's entry list (hv_box) packed for every subset of radius classes an episode may keep a field for, as the throughput
// kernel reads it (eg_rollout.hip heavy_add_body: the entry's place in the compact factor table instead of its squared distance),
// padded to a multiple of four entries per lane; hv_quads = that multiple.  (The small-batch kernel packs its own list into LDS.)
constexpr size_t hv_lists = a16(pcell + 4 * size_t(kYears) * kMaxVariants * kPcStride);     // u32 [64][1024]
constexpr size_t hv_quads = a16(hv_lists + 4 * size_t(64) * 1024);                          // i32 [64]
// the placement prefix per CELL (HostTables::te, coastf): what the hoisted replay (eg_replay_coop.h) reads — its sixteen waves hold
// every cell of the grid at once, so it wants the candidates in cell order, not sorted
constexpr size_t te_cell = a16(hv_quads + 4 * 64);                                          // f64 [26][6][2601]
constexpr size_t coastf = a16(te_cell + 8 * size_t(kYears) * kRadiusClasses * kCells);      // f64 [2601]
// ... and the unpenalised score (te * cf) * size_factor of every (year, variant) per cell — tab::pbase in cell order: what the hoisted
// replay's searches multiply the penalty field with (te_cell / coastf: the exact evaluation of tied candidates)
constexpr size_t cbase = a16(coastf + 8 * size_t(kCells));                                  // f64 [26][kMaxVariants][2601]
constexpr size_t total = a16(cbase + 8 * size_t(kYears) * kMaxVariants * kCells);
}  // namespace tab

struct DevTables {
  const uint8_t* base;
  double size_factor;
  int32_t n_existing;
  int32_t n_variants;      // distinct (radius class, marine) pairs among the generator types (tab::variant)
  // heavy episodes (eg_rollout.hip, place_heavy): a pool of per-episode penalty fields [6][2624] f64 in HBM, claimed per
  // launch through `heavy_claim` (launch epoch << 20 | slots handed out)
  uint32_t heavy_slots, heavy_epoch;
  uint8_t* heavy;
  unsigned* heavy_claim;
#define EG_TAB(name, type) EG_HD const type* name() const { return reinterpret_cast<const type*>(base + tab::name); }
  EG_TAB(usage, double) EG_TAB(population, double)
  EG_TAB(pre_co2, double) EG_TAB(pre_tg, double) EG_TAB(pre_ig, double) EG_TAB(pre_sg, double) EG_TAB(pre_optot, double)
  EG_TAB(inflation, double) EG_TAB(carbon_price, double) EG_TAB(out_mw, double) EG_TAB(co2_t, double)
  EG_TAB(pre_opcnt, int32_t) EG_TAB(variant, int32_t) EG_TAB(cls, int32_t) EG_TAB(rclass, int32_t) EG_TAB(marine, int32_t)
  EG_TAB(reach, int32_t)
  EG_TAB(dr, double) EG_TAB(m03, double) EG_TAB(t12, double) EG_TAB(offv, double) EG_TAB(offc, double) EG_TAB(cc, double)
  // placement: candidates of every (year, variant) sorted by unpenalised score, descending (ties: ascending cell)
  EG_TAB(ps, PsRec) EG_TAB(pbase, double) EG_TAB(pcell, uint32_t) EG_TAB(hv_lists, uint32_t) EG_TAB(hv_quads, int32_t) EG_TAB(dr_meta, int32_t) EG_TAB(dr_compact, double)
  EG_TAB(te_cell, double) EG_TAB(coastf, double) EG_TAB(cbase, double)
#undef EG_TAB
};

// Scalars of the policy as the kernels see them.  They live in the snapshot buffer (snap::state), not in kernel
// arguments: the host writes them at upload (eg_reduced_math.h derive_state), an on-device update rewrites them.
struct DevState {
  double learning_rate, exploration_rate;
  double best_metrics[4];
  double rel_improvement, immediate_weight;                   // learning.rs:37-55
  double p_best_score, p_threshold, p_adaptive_lr, p_stagnation;   // contrast step, learning.rs:131-180
  // expressions of the scalars that every action or nudge of an episode evaluates (derive_state): 1 + lr * 0.1 and 1 + lr * 0.2
  // (learning.rs:74-87, deficit.rs:118-127), the stall-scaled exploration rate of sample_action (sampling.rs:150-157), the power
  // of the stalled sampler 1 + 2 min(stall / 1000, 3) (sampling.rs:193-195).  As loads they become scalar registers of the
  // episode; as loop-invariant f64 expressions the compiler hoisted them into vector registers and spilled them.
  double boost_others, boost_noop, eps_main, scaled_power;
  uint32_t stall, iteration_count;
  int32_t has_best, has_cw, noop_boost, has_lists, p_forced;
  uint32_t heur_min, heur_max;                                // sampling.rs:425-427
  uint32_t n_improvements;     // improvements installed by on-device updates since the last upload (log entries written)
  int32_t improved_last;       // the last on-device update installed a new best strategy
  uint32_t failed_total;       // episodes (all ranks) that ended with a status other than EG_EP_OK, as counted by the updates
};
static_assert(sizeof(DevState) % 8 == 0, "state layout");
struct DevImprovement { double score, metrics[4]; uint32_t iteration, pad; };   // on-device improvement log entry

// Policy snapshot in HBM: one packed buffer (filled by one copy from pinned memory), fixed layout.
namespace snap {
constexpr size_t kBestCap = 4096;     // best_actions can hold every replay-doubled year list
// One 128-double block per year carries everything the episode wave loads into LDS at the start of that year:
//   [0,61) main weights   [61] their table-order sum   [62] sum of the first 14 deficit weights   [63] sum of the count row
//   [64,79) deficit weights   [79] unused (was: the stalled sampler's sum of the powered weights; weighted_pick forms its own)
//   [80,101) action-count weights (has_cw)
//   [101,111) the world's scalars of that year (copied from the host tables at upload so that they need no LDS table of
//             their own): existing-plant prefix of CO2 / dispatchable / intermittent / storage output / opinion total,
//             demand, population, inflation, carbon price, existing-plant count   [111,128) zero
constexpr int kPolRow = 128, kPolTotMain = 61, kPolTotDeficit = 62, kPolTotCount = 63, kPolDw = 64, kPolScaledTotal = 79, kPolCw = 80;
constexpr int kPolYear = 101;   // + {0 pre_co2, 1 pre_tg, 2 pre_ig, 3 pre_sg, 4 pre_optot, 5 usage, 6 population, 7 inflation, 8 carbon_price, 9 pre_opcnt}
constexpr size_t pol = 0;                                                   // f64 [26][128]
// stalled sampler (sampling.rs:190-220, stall > 500): per year the weights raised to the power in stable descending
// order and the permutation, evaluated on the device with the shared eg_detpow; valid until the first nudge
constexpr size_t scaled = pol + 8 * EG_YEARS * kPolRow;                     // [26][64]
constexpr size_t scaled_perm = scaled + 8 * EG_YEARS * 64;                  // u8 [26][64]
constexpr size_t best_mask = scaled_perm + 64 * EG_YEARS;                   // u64 [26] bit a: a occurs in best(y) or best_deficit(y)
constexpr size_t bestd_mask = best_mask + 8 * EG_YEARS;                     // u64 [26] bit a: a occurs in best_deficit(y)
constexpr size_t best_off = bestd_mask + 8 * EG_YEARS;                      // i32 [27(+1)] prefix offsets into best_actions
constexpr size_t bestd_off = best_off + 4 * 28;
constexpr size_t best_actions = bestd_off + 4 * 28;                         // u8 [kBestCap]
constexpr size_t bestd_actions = best_actions + kBestCap;
constexpr size_t state = (bestd_actions + kBestCap + 15) & ~size_t(15);     // DevState
constexpr size_t upload_bytes = state + sizeof(DevState);                   // what eg_upload_snapshot copies
// device-resident policy only (on-device updates, eg_policy_pull):
constexpr int kImpLogCap = 256;
constexpr size_t best_w = (upload_bytes + 15) & ~size_t(15);                // f64 [26][61] main weights at the last improvement
constexpr size_t imp_log = best_w + 8 * EG_YEARS * EG_N_ACTIONS;            // DevImprovement [kImpLogCap] (ring)
// Record (rec:: layout) of the best episode, kept by k_apply_update when that episode ran on this device, and its state
// word: 0 = empty, 1 = holds the best episode's record, 2 = the best episode ran on another rank.  Outside the uploaded
// range, so it survives eg_upload_snapshot / eg_policy_push on the same context.
constexpr size_t best_rec_state = (imp_log + sizeof(DevImprovement) * kImpLogCap + 63) & ~size_t(63);   // u32
constexpr size_t best_rec = best_rec_state + 64;
}  // namespace snap

struct DevSnapshot {
  const uint8_t* base;
  int32_t enable_energy_sales;   // run options (host)
  int32_t write_yearly;
  // the fields below are filled IN the kernel from snap::state (load_state); the host leaves them alone
  double learning_rate, exploration_rate;
  uint32_t stall;
  int32_t has_best;
  int32_t has_cw;              // the count table is present (else: heuristic count, sampling.rs:425-427)
  int32_t noop_boost;          // best is net-zero but above 8x the acceptable cost (learning.rs:82)
  double rel_improvement;      // learning.rs:37-49 evaluated on the host
  double immediate_weight;     // learning.rs:54
  int32_t has_best_actions, has_best_deficit;
  uint32_t heur_min, heur_max;   // sampling.rs:425-427 (count table absent)
  // expressions of the scalars above that every action or nudge evaluates (kept as scalars: a loop-invariant f64 expression
  // of uniform values is hoisted by the compiler into VECTOR registers and held there for the whole episode)
  double boost_others, boost_noop;   // 1 + lr * 0.1, 1 + lr * 0.2 (learning.rs:74-87, deficit.rs:118-127)
  double eps_main;                   // exploration rate of sample_action, stall-scaled (sampling.rs:150-157)
  double scaled_power;               // 1 + 2 * min(stall / 1000, 3) (sampling.rs:193-195)
#define EG_SNAP(name, type) EG_HD const type* name() const { return reinterpret_cast<const type*>(base + snap::name); }
  EG_SNAP(pol, double) EG_SNAP(scaled, double) EG_SNAP(scaled_perm, uint8_t)
  EG_SNAP(best_mask, unsigned long long) EG_SNAP(bestd_mask, unsigned long long)
  EG_SNAP(best_off, int32_t) EG_SNAP(bestd_off, int32_t) EG_SNAP(best_actions, uint8_t) EG_SNAP(bestd_actions, uint8_t)
  EG_SNAP(state, DevState)
#undef EG_SNAP
};

// Per-batch outputs in HBM: one record per episode (everything an episode writes is contiguous), fixed layout.
namespace rec {
constexpr size_t metrics = 0;                         // f64 [4]
constexpr size_t score = 32;                          // f64, written by the statistics pass
constexpr size_t bytes_moved = 40;                    // f64
constexpr size_t n_draws = 48;                        // u64
constexpr size_t status = 56;                         // i32
constexpr size_t n_gens = 60;                         // i32
constexpr size_t n_offsets = 64;                      // i32
constexpr size_t n_chunks = 68;                       // u32: chunks of candidate records the placement searches requested
constexpr size_t n_run = 72;                          // i32 [26]
constexpr size_t n_def = n_run + 4 * EG_YEARS;
constexpr size_t n_act = n_def + 4 * EG_YEARS;
constexpr size_t yearly = n_act + 4 * EG_YEARS;       // f64 [26][21]
constexpr size_t run_log = yearly + 8 * EG_YEARS * EG_YEARLY_FIELDS;
constexpr size_t def_log = run_log + EG_RUN_CAP;
constexpr size_t act_log = def_log + EG_DEF_CAP;
constexpr size_t gen_cell = act_log + EG_ACT_CAP;     // u16 [EG_MAX_GENS]
constexpr size_t gen_pack = gen_cell + 2 * EG_MAX_GENS;
constexpr size_t off_pack = gen_pack + 2 * EG_MAX_GENS;
constexpr size_t stride = (off_pack + 2 * EG_MAX_OFFSETS + 63) & ~size_t(63);
static_assert(yearly % 8 == 0 && gen_cell % 2 == 0, "record alignment");
}  // namespace rec
namespace snap { constexpr size_t total = best_rec + rec::stride; }

struct DevOut {
  uint8_t* base;
  double* score_list;      // the episodes' scores once more, back to back (the update's best pick reads all of them: one per 12 KB record is one cache line each)
#define EG_REC(name, type) EG_HD type* name(uint32_t e) const { return reinterpret_cast<type*>(base + size_t(e) * rec::stride + rec::name); }
  EG_REC(metrics, double) EG_REC(score, double) EG_REC(bytes_moved, double) EG_REC(n_draws, unsigned long long)
  EG_REC(status, int32_t) EG_REC(n_gens, int32_t) EG_REC(n_offsets, int32_t) EG_REC(n_chunks, uint32_t)
  EG_REC(n_run, int32_t) EG_REC(n_def, int32_t) EG_REC(n_act, int32_t) EG_REC(yearly, double)
  EG_REC(run_log, uint8_t) EG_REC(def_log, uint8_t) EG_REC(act_log, uint8_t)
  EG_REC(gen_cell, uint16_t) EG_REC(gen_pack, uint16_t) EG_REC(off_pack, uint16_t)
#undef EG_REC
};

void set_error(const std::string& s);
double action_cost_estimate(int action, int year_index);      // eg_tables.cpp; simulation_summary.csv "Estimated Cost"

// launchers implemented in eg_rollout.hip
struct StatsParams;
struct UpdateCandidate;
// How a batch is split over the variants of k_rollout (eg_rollout.hip, EpisodeMap): n_heavy episodes — the ones that replay
// the best strategy — on the two replay variants (the short-replay and the heavy-capable one: the one whose turn it is not
// returns at once), n_lean on the lean one; on two streams side by side when both are present (stream_heavy: the library's
// side stream, stream_lean: the null stream).  mode 1: d_index holds the n_heavy replay episodes followed by the n_lean others; mode 2: the replays are
// off + period * j.  ev: start / stop events of the heavy and of the lean grid (only the ones launched are recorded).
struct HoistInfo;
struct RolloutPlan {
  bool helper_waves;
  void* stream_heavy; void* stream_lean;
  void* go_event;      // not null: recorded on stream_heavy between the two replay variants; the lean grid's stream waits for it
  bool skip_long;      // the host KNOWS the best list to be short: the long-replay variant (which would return at once) is not launched
  uint32_t n_heavy, n_lean, mode;
  const uint32_t* d_index;
  uint32_t off, period;
  void* ev[4];
  // replay hoist (eg_replay_coop.h): hoist_seq != 0 = the replay episodes of this batch are computed ONCE by k_replay_coop into the
  // scratch record `coop_out` and copied to every replay slot by k_replay_broadcast; d_hoist = {u64 word: the sequence number of the
  // last batch whose hoist succeeded, i32 lengths[8]} — the per-episode replay variants are launched in between and return at once
  // when the word says their batch has been served
  unsigned long long hoist_seq;
  HoistInfo* d_hoist;
  uint8_t* coop_out;
  int coop_force;      // test hook (EIRGRID_COOP_FORCE): 1 = every hoisted search evaluates its candidates exactly, 2 = every one is the exact scan
  // statistics epilogue: kStatsReplicas copies of the statistics array, entry-major — an episode adds to copy (workgroup index % kStatsReplicas) and
  // k_fold_stats folds the copies into the packet behind the batch.  16 384 episodes adding to the same few hundred addresses are
  // serialised address by address in L2: with the replay episodes hoisted that was 0.8 ms of a 1.9 ms batch (profiles/r04_ab_notes.log).
  long long* d_stats_rep;
  // per-episode replay kernel (eg_replay_solo.h): solo_seq != 0 = k_replay_solo runs ahead of the long-replay variant and marks what it
  // has completed in d_solo (a word per workgroup of the replay grid)
  unsigned long long solo_seq;
  unsigned long long* d_solo;
};
constexpr int kStatsReplicas = 64;
int launch_fold_stats(long long* d_rep, long long* d_stats, void* stream);
// d_hoist (device): what the kernels of the replay hoist hand each other
struct HoistInfo {
  unsigned long long served_seq;      // written by k_replay_books: the batch whose replay episodes are served by the scratch record
  unsigned long long coop_seq;        // written by k_replay_coop: the batch whose script and placements are complete in the scratch record
  int32_t lens[6];                    // recorded run / deficit / additional actions, generators, offsets, searches beyond the one exchange
  int32_t pad[2];
  int32_t g_end[EG_YEARS], o_end[EG_YEARS];      // generators / offsets at the end of each year
  unsigned long long bytes;           // algorithmic bytes of the episode (SURVEY §8(d) formula, as k_rollout counts them)
  int32_t books_done, pad2;           // k_replay_books: workgroups (years) that have written their row
  double year[EG_YEARS][8];           //   their yearly total cost, credit, sales; net, opinion, total capital, balance
  unsigned long long stamps[8];       // diagnostic builds (-DEG_COOP_STAMPS): cycle counts of the phases
};
constexpr size_t kHoistBytes = (sizeof(HoistInfo) + 63) & ~size_t(63);
int launch_rollout(const DevTables& t, const DevSnapshot& s, const DevOut& o, uint64_t seed, uint64_t first_index,
                   uint32_t n, const uint8_t* d_replay_mask, uint32_t replay_period, long long* d_stats, const RolloutPlan& plan);
int launch_fill_lds(uint32_t value, uint32_t* d_sink, int n_workgroups, void* stream);      // test hook
int launch_occupy(int variant, unsigned long long cycles, uint32_t* d_sink, void* stream);      // diagnostic hook (eg_debug_occupy)
int launch_stalled_tables(uint8_t* d_snap, void* stream);
// d_snap = d_held, except the count of failed episodes, which goes on counting (eg_policy_rewind)
// (`list_len_out`: pinned host word, may be null — the length of the best list as the device now holds it, for the host's launch planning)
int launch_rewind(uint8_t* d_snap, const uint8_t* d_held, uint32_t* list_len_out, void* stream);
// `o`, n_local, first_index: the batch the own packet came from (the winner's record is kept when it is one of them)
// local_pick: one packet, made by this device's last batch — the candidate record is built inside the kernel
int launch_apply_update(uint8_t* d_snap, const void* d_packets, int n_packets, long long* d_zero_stats, uint64_t noise_seed,
                        const DevOut& o, uint32_t n_local, uint64_t first_index, bool local_pick, uint32_t* list_len_out, void* stream);
int launch_place(const DevTables& t, int gen_type, int year_index, const uint16_t* d_cells, int n_extra,
                 int32_t* d_out_cell, double* d_out_score, void* stream);
int launch_place_xy(const DevTables& t, int gen_type, int year_index, const double* d_x, const double* d_y, int n, double radius,
                    double size_term, int32_t* d_out_cell, double* d_out_score, void* stream);
double class_radius(int radius_class);      // eg_tables.cpp: metal_location_search.rs:139-146
// scalars of the contrast step that depend only on the snapshot (learning.rs:131-180); filled in the kernels from
// snap::state
struct StatsParams {
  double best_score;     // score_metrics(best_metrics)
  int32_t has_best;      // best metrics and best action lists present
  double threshold;      // dynamic threshold, learning.rs:146-154
  int32_t forced;        // iterations_without_improvement > 800
  double adaptive_lr;    // learning.rs:174
  double stagnation;     // learning.rs:163-164
};
int launch_update_stats(const DevSnapshot& s, const DevOut& o, uint32_t n, long long* d_stats, void* stream);
// best episode of a batch, laid out right behind the statistics in the update packet (eg_rollout_launch_update)
struct UpdateCandidate {
  double score;            // -1 when the batch has no successful episode
  long long index;         // global episode index
  double metrics[4];
  int32_t n_run[EG_YEARS], n_def[EG_YEARS];
  uint8_t run_log[EG_RUN_CAP], def_log[EG_DEF_CAP];
};
static_assert(sizeof(UpdateCandidate) == EG_CANDIDATE_BYTES, "candidate layout is part of the C ABI");
int launch_pick_best(const DevOut& o, uint32_t n, uint64_t first_index, UpdateCandidate* d_cand, void* stream);
// The reference's `best_result` fold (core/multi_simulation.rs:613-620; eg_rollout.hip k_fold_best): the held run's metrics, global
// index and — kFoldRecord bytes behind the state — its whole record (rec:: layout).
struct FoldState { double metrics[4]; long long index; int32_t has, pad; };
constexpr size_t kFoldRecord = 64;
constexpr size_t kFoldBytes = kFoldRecord + rec::stride;
static_assert(sizeof(FoldState) <= kFoldRecord, "fold state layout");
int launch_fold_best(const DevOut& o, uint32_t n, uint64_t first_index, bool cost_only, uint8_t* d_fold, void* stream);

}  // namespace eg
