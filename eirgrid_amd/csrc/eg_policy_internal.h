// eg_policy_internal.h — the host-side ActionWeights object shared by eg_policy.cpp and eg_checkpoint.cpp.
#pragma once
#include <array>
#include <cstdint>
#include <ctime>
#include <string>
#include <vector>

#include "eirgrid_hip.h"

constexpr int Y = EG_YEARS, NA = EG_N_ACTIONS, ND = EG_N_DEFICIT, NC = EG_N_COUNTS;
using ActionList = std::vector<uint8_t>;

struct ImprovementRecord {   // utils/csv_export.rs ImprovementRecord via ai/learning/serialization.rs:10-19
  uint32_t iteration; double score, net_emissions, total_cost, public_opinion, power_reliability; std::string timestamp;
};

struct eg_policy {
  std::array<std::array<double, NA>, Y> w{};
  std::array<std::array<double, ND>, Y> dw{};
  std::array<std::array<double, NC>, Y> cw{};
  bool has_cw = true;
  double learning_rate = 0.2, exploration_rate = 0.2;   // constants.rs:17-18
  bool has_best = false;
  std::array<double, 4> best_metrics{};
  bool has_best_weights = false;
  std::array<std::array<double, NA>, Y> best_w{};
  bool has_best_actions = false, has_best_deficit = false;
  std::array<ActionList, Y> best_actions, best_deficit, cur_run, cur_def;
  uint32_t iteration_count = 0, stall = 0;
  uint32_t failed_episodes = 0;   // episodes the batch updates saw fail (EG_EP_OVERFLOW, ...): never part of iteration_count
  std::vector<ImprovementRecord> improvement_history;
  void record_improvement(double score, const double m[4]) {   // strategy.rs:71-84
    char buf[32]; std::time_t t = std::time(nullptr); std::tm tmv; localtime_r(&t, &tmv);
    std::strftime(buf, sizeof(buf), "%Y-%m-%d %H:%M:%S", &tmv);
    improvement_history.push_back({iteration_count, score, m[0], m[2], m[1], m[3], buf});
  }
  // flattened replay data handed out by eg_policy_snapshot_view
  mutable std::vector<int32_t> flat_best_count, flat_bestd_count;
  mutable std::vector<uint8_t> flat_best, flat_bestd;
};

