// eg_internal.h — layouts shared by the host table builder, the C ABI glue and the HIP kernels.
#pragma once
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
constexpr int kMaxVariants = 12;  // distinct (radius class, marine) pairs over the 15 types (8 for the reference's types)

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

// Device view (raw pointers into HBM), passed to kernels by value.
struct DevTables {
  const double* usage; const double* population;
  const double* pre_co2; const double* pre_tg; const double* pre_ig; const double* pre_sg; const double* pre_optot;
  const int32_t* pre_opcnt;
  // placement: candidates of every (year, variant) sorted by unpenalised score, descending (ties: ascending cell)
  const uint16_t* ps_cell; const double* ps_te; const double* ps_cf; const double* ps_m03;   // [26][n_variants][kPsStride]
  const int32_t* variant;   // [15] type -> variant
  int32_t n_variants;
  const double* dr; double size_factor;
  const double* m03; const double* t12; const double* cc;
  const double* out_mw; const double* co2_t;
  const int32_t* cls; const int32_t* rclass; const int32_t* marine; const int32_t* reach;
  const double* offv; const double* offc;
  const double* inflation; const double* carbon_price;
  int32_t n_existing;
};

// Policy snapshot in HBM
struct DevSnapshot {
  const double* w;    // [26][61]
  const double* dw;   // [26][15]
  const double* cw;   // [26][21] or nullptr
  const double* row_totals;   // [26][3] table-order sums of the w / dw (first 14) / cw rows of each year (host-evaluated)
  // stalled sampler (sampling.rs:190-220, stall > 500): per year the weights raised to the power in stable descending
  // order, the permutation and the sum, evaluated on the host with the shared eg_detpow; valid until the first nudge
  const double* scaled;       // [26][64]
  const uint8_t* scaled_perm; // [26][64]
  const double* scaled_total; // [26]
  double learning_rate, exploration_rate;
  uint32_t stall;
  int32_t has_best;
  int32_t noop_boost;          // best is net-zero but above 8x the acceptable cost (learning.rs:82)
  double rel_improvement;      // learning.rs:37-49 evaluated on the host
  double immediate_weight;     // learning.rs:54
  int32_t has_best_actions, has_best_deficit;
  uint32_t heur_min, heur_max;   // sampling.rs:425-427 evaluated on the host (count table absent)
  const int32_t* best_off;     // [27] prefix offsets into best_actions
  const uint8_t* best_actions;
  const int32_t* bestd_off;    // [27]
  const uint8_t* bestd_actions;
  const unsigned long long* best_mask;   // [26] bit a set: action a occurs in best_actions[y] or best_deficit_actions[y]
  const unsigned long long* bestd_mask;  // [26] bit a set: action a occurs in best_deficit_actions[y]
  int32_t enable_energy_sales;
  int32_t write_yearly;
};

// Per-batch output buffers in HBM (episode-major)
struct DevOut {
  double* metrics; double* yearly; int32_t* status;
  int32_t* n_run; int32_t* n_def; int32_t* n_act;
  uint8_t* run_log; uint8_t* def_log; uint8_t* act_log;
  int32_t* n_gens; uint16_t* gen_cell; uint16_t* gen_pack;
  int32_t* n_offsets; uint16_t* off_pack;
  unsigned long long* n_draws; double* bytes_moved;
  double* score;   // written by k_update_stats
};

void set_error(const std::string& s);

// launchers implemented in eg_rollout.hip
struct StatsParams;
struct UpdateCandidate;
int launch_rollout(const DevTables& t, const DevSnapshot& s, const DevOut& o, uint64_t seed, uint64_t first_index,
                   uint32_t n, const uint8_t* d_replay_mask, const StatsParams& p, long long* d_stats, void* stream,
                   bool helper_waves);
int launch_place(const DevTables& t, int gen_type, int year_index, const uint16_t* d_cells, int n_extra,
                 int32_t* d_out_cell, double* d_out_score, void* stream);
// scalars of the contrast step that depend only on the snapshot (learning.rs:131-180), evaluated on the host
struct StatsParams {
  double best_score;     // score_metrics(best_metrics)
  int32_t has_best;      // best metrics and best action lists present
  double threshold;      // dynamic threshold, learning.rs:146-154
  int32_t forced;        // iterations_without_improvement > 800
  double adaptive_lr;    // learning.rs:174
  double stagnation;     // learning.rs:163-164
};
int launch_update_stats(const DevSnapshot& s, const DevOut& o, const StatsParams& p, uint32_t n, long long* d_stats, void* stream);
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

}  // namespace eg
