// eg_policy.cpp — host-side ActionWeights: the tabular policy and its post-episode update.
//
// Mirrors ai/learning/weights/{mod,core,learning,strategy}.rs and ai/metrics/scoring.rs of the reference with the
// canonical (insertion-order) action indexing of include/eirgrid_hip.h.  The rollout itself (sampling, in-episode
// nudges) runs on the device; what is here is what the reference does under the shared write lock after each
// episode (core/multi_simulation.rs:494-508) plus construction and snapshot export.  Host code only: the update
// is sequentially dependent across episodes and touches ~2 000 doubles.
#include <array>
#include <cmath>
#include <cstring>
#include <vector>

#include "eg_internal.h"
#include "eg_policy_internal.h"
#define EG_RM static inline
#include "eg_reduced_math.h"
namespace rm = eg::rm;

namespace {

constexpr double kMinW = 0.0001, kMaxW = 0.999;          // ai/learning/constants.rs:14-15
constexpr double kMaxEmissions = 1000000.0, kMaxCost = 50000000000.0;   // config/constants.rs:114-115


// initial weights per generator type, enum order (ai/learning/constants.rs:46-60)
const double kTypeWeight[EG_N_TYPES] = {0.08, 0.08, 0.05, 0.05, 0.08, 0.03, 0.04, 0.06, 0.02, 0.04, 0.06, 0.06, 0.07, 0.05, 0.05};
// deficit table, insertion order (core.rs:130-152; constants.rs:66-77)
const double kDeficitWeight[ND] = {0.15, 0.15, 0.15, 0.10, 0.10, 0.07, 0.07, 0.06, 0.06, 0.05, 0.01, 0.01, 0.01, 0.01, 0.001};
const int kDeficitType[14] = {8, 7, 12, 11, 9, 0, 1, 4, 10, 5, 2, 3, 13, 14};

int deficit_slot(int action) {
  if (action == 60) return 14;
  if (action >= 45 || action % 3 != 0) return -1;
  for (int i = 0; i < 14; ++i) if (kDeficitType[i] == action / 3) return i;
  return -1;
}

// ChaCha12 stream for the stagnation noise (learning.rs:267-280, :356-369 use thread_rng; the canonical
// definition draws from StdRng::seed_from_u64(noise_seed))
struct HostRng {
  uint32_t key[8]; uint64_t counter = 0; uint32_t buf[64]; int index = 64;
  static uint32_t rotl(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }
  explicit HostRng(uint64_t state) {
    for (int i = 0; i < 8; ++i) {
      state = state * 6364136223846793005ull + 11634580027462260723ull;
      uint32_t x = uint32_t(((state >> 18) ^ state) >> 27), rot = uint32_t(state >> 59);
      key[i] = (x >> rot) | (x << ((32 - rot) & 31));
    }
  }
  // rand_chacha refills four consecutive blocks at a time; they are computed side by side (x[word][block]) so that
  // the compiler turns every step into one 4 x u32 vector operation
  void refill() {
    uint32_t s[16][4], x[16][4];
    const uint32_t kc[4] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
    for (int l = 0; l < 4; ++l) {
      for (int i = 0; i < 4; ++i) s[i][l] = kc[i];
      for (int i = 0; i < 8; ++i) s[4 + i][l] = key[i];
      const uint64_t ctr = counter + uint64_t(l);
      s[12][l] = uint32_t(ctr); s[13][l] = uint32_t(ctr >> 32); s[14][l] = 0; s[15][l] = 0;
    }
    std::memcpy(x, s, sizeof(x));
#define EG_QR4(a, b, c, d)                                                                                  \
    for (int l = 0; l < 4; ++l) { x[a][l] += x[b][l]; x[d][l] = rotl(x[d][l] ^ x[a][l], 16); }                \
    for (int l = 0; l < 4; ++l) { x[c][l] += x[d][l]; x[b][l] = rotl(x[b][l] ^ x[c][l], 12); }                \
    for (int l = 0; l < 4; ++l) { x[a][l] += x[b][l]; x[d][l] = rotl(x[d][l] ^ x[a][l], 8); }                 \
    for (int l = 0; l < 4; ++l) { x[c][l] += x[d][l]; x[b][l] = rotl(x[b][l] ^ x[c][l], 7); }
    for (int r = 0; r < 6; ++r) {
      EG_QR4(0, 4, 8, 12) EG_QR4(1, 5, 9, 13) EG_QR4(2, 6, 10, 14) EG_QR4(3, 7, 11, 15)
      EG_QR4(0, 5, 10, 15) EG_QR4(1, 6, 11, 12) EG_QR4(2, 7, 8, 13) EG_QR4(3, 4, 9, 14)
    }
#undef EG_QR4
    for (int l = 0; l < 4; ++l) for (int i = 0; i < 16; ++i) buf[16 * l + i] = x[i][l] + s[i][l];
    counter += 4; index = 0;
  }
  uint64_t next_u64() {
    if (index >= 63) refill();
    uint64_t v = (uint64_t(buf[index + 1]) << 32) | buf[index]; index += 2; return v;
  }
  double next_f64() { return double(next_u64() >> 11) * (1.0 / 9007199254740992.0); }
};

}  // namespace

extern "C" {

double eg_score_metrics(const double m[4], int32_t cost_only) {   // ai/metrics/scoring.rs:5-45
  const double normalized_cost = std::fmax(m[2] / kMaxCost, 1.0);
  const double log_cost = std::log(normalized_cost);
  const double max_expected = std::log(kMaxCost * 100.0 / kMaxCost);
  if (cost_only) return 2.0 - std::fmin(log_cost / max_expected, 1.0);
  if (m[0] > 0.0) return 1.0 - std::fmin(m[0] / kMaxEmissions, 1.0);
  const double cost_score = 1.0 - std::fmin(log_cost / max_expected, 1.0);
  const double cost_weight = normalized_cost > 8.0 ? 0.8 : 0.5;
  const double opinion_weight = 1.0 - cost_weight;
  return 1.0 + (cost_score * cost_weight + m[1] * opinion_weight);
}

double eg_evaluate_action_impact(const double cur[4], const double nxt[4], int32_t cost_only) {   // ai/metrics/scoring.rs:46-85
  // SimulationMetrics -> ActionResult as core/multi_simulation.rs:55-62 maps them: net emissions [0], opinion [1], total cost [2]
  if (cost_only) { const double cost_change = nxt[2] - cur[2]; return -cost_change / std::fmax(std::fabs(cur[2]), 1.0); }
  if (cur[0] > 0.0) return (cur[0] - nxt[0]) / std::fmax(std::fabs(cur[0]), 1.0);
  const double cost_change = nxt[2] - cur[2];
  const double cost_improvement = -cost_change / std::fmax(std::fabs(cur[2]), 1.0);
  const double opinion_improvement = (nxt[1] - cur[1]) / std::fmax(std::fabs(cur[1]), 1.0);
  const double cost_weight = cur[2] > kMaxCost * 8.0 ? 0.8 : 0.5;
  const double opinion_weight = 1.0 - cost_weight;
  return cost_improvement * cost_weight + opinion_improvement * opinion_weight;
}

eg_policy* eg_policy_new(void) {   // core.rs:25-250
  eg_policy* p = new eg_policy();
  for (int y = 0; y < Y; ++y) {
    for (int t = 0; t < EG_N_TYPES; ++t) {
      p->w[y][3 * t] = kTypeWeight[t]; p->w[y][3 * t + 1] = kTypeWeight[t] * 0.5; p->w[y][3 * t + 2] = kTypeWeight[t] * 0.25;
    }
    for (int o = 0; o < 4; ++o) { p->w[y][45 + 3 * o] = 0.02; p->w[y][46 + 3 * o] = 0.02 * 0.5; p->w[y][47 + 3 * o] = 0.02 * 0.25; }
    p->w[y][57] = 0.04; p->w[y][58] = 0.04; p->w[y][59] = 0.02; p->w[y][60] = 0.1;   // constants.rs:61-65
    for (int i = 0; i < ND; ++i) p->dw[y][i] = kDeficitWeight[i];
    double total = 0.0;   // core.rs:158-186
    for (int c = 0; c < NC; ++c) {
      const double bias = c == 0 ? 4.0 : c == 1 ? 3.5 : c == 2 ? 3.0 : c == 3 ? 2.5 : c == 4 ? 2.0 : c == 5 ? 1.5 : 1.0;
      p->cw[y][c] = std::exp(-0.8 * double(c)) * bias;
      total += p->cw[y][c];
    }
    for (int c = 0; c < NC; ++c) p->cw[y][c] /= total;
  }
  return p;
}
void eg_policy_free(eg_policy* p) { delete p; }

int32_t eg_policy_snapshot_view(const eg_policy* p, eg_policy_snapshot* s) {
  if (!p || !s) return EG_ERR_BAD_ARG;
  std::memset(s, 0, sizeof(*s));
  s->weights = &p->w[0][0]; s->deficit_weights = &p->dw[0][0]; s->count_weights = p->has_cw ? &p->cw[0][0] : nullptr;
  s->learning_rate = p->learning_rate; s->exploration_rate = p->exploration_rate;
  s->iterations_without_improvement = p->stall;
  s->has_best = p->has_best ? 1 : 0;
  for (int i = 0; i < 4; ++i) s->best_metrics[i] = p->best_metrics[i];
  if (p->has_best_actions && p->has_best_deficit) {
    p->flat_best_count.assign(Y, 0); p->flat_bestd_count.assign(Y, 0); p->flat_best.clear(); p->flat_bestd.clear();
    for (int y = 0; y < Y; ++y) {
      p->flat_best_count[y] = int32_t(p->best_actions[y].size()); p->flat_bestd_count[y] = int32_t(p->best_deficit[y].size());
      p->flat_best.insert(p->flat_best.end(), p->best_actions[y].begin(), p->best_actions[y].end());
      p->flat_bestd.insert(p->flat_bestd.end(), p->best_deficit[y].begin(), p->best_deficit[y].end());
    }
    p->flat_best.push_back(0); p->flat_bestd.push_back(0);   // keep data() non-null
    s->best_count = p->flat_best_count.data(); s->best_actions = p->flat_best.data();
    s->best_deficit_count = p->flat_bestd_count.data(); s->best_deficit_actions = p->flat_bestd.data();
  }
  return EG_OK;
}
int32_t eg_policy_get_tables(const eg_policy* p, double* w, double* dw, double* cw) {
  if (!p) return EG_ERR_BAD_ARG;
  if (w) std::memcpy(w, &p->w[0][0], sizeof(double) * Y * NA);
  if (dw) std::memcpy(dw, &p->dw[0][0], sizeof(double) * Y * ND);
  if (cw) std::memcpy(cw, &p->cw[0][0], sizeof(double) * Y * NC);
  return EG_OK;
}
int32_t eg_policy_set_tables(eg_policy* p, const double* w, const double* dw, const double* cw) {
  if (!p) return EG_ERR_BAD_ARG;
  if (w) std::memcpy(&p->w[0][0], w, sizeof(double) * Y * NA);
  if (dw) std::memcpy(&p->dw[0][0], dw, sizeof(double) * Y * ND);
  if (cw) std::memcpy(&p->cw[0][0], cw, sizeof(double) * Y * NC);
  return EG_OK;
}
// scalar codes: 0 learning_rate 1 exploration_rate 2 iterations_without_improvement 3 iteration_count 4 has_best
//               5..8 best_metrics 9 has_best_actions 10 has_best_deficit_actions 11 has_count_weights
//               12 improvement_history length (read only) 13 failed_episodes
double eg_policy_get_scalar(const eg_policy* p, int32_t which) {
  switch (which) {
    case 0: return p->learning_rate; case 1: return p->exploration_rate; case 2: return double(p->stall);
    case 3: return double(p->iteration_count); case 4: return p->has_best ? 1.0 : 0.0;
    case 5: case 6: case 7: case 8: return p->best_metrics[which - 5];
    case 9: return p->has_best_actions ? 1.0 : 0.0; case 10: return p->has_best_deficit ? 1.0 : 0.0;
    case 11: return p->has_cw ? 1.0 : 0.0;
    case 12: return double(p->improvement_history.size());
    case 13: return double(p->failed_episodes);
    default: return 0.0;
  }
}
int32_t eg_policy_set_scalar(eg_policy* p, int32_t which, double v) {
  switch (which) {
    case 0: p->learning_rate = v; break; case 1: p->exploration_rate = v; break; case 2: p->stall = uint32_t(v); break;
    case 3: p->iteration_count = uint32_t(v); break; case 4: p->has_best = v != 0.0; break;
    case 5: case 6: case 7: case 8: p->best_metrics[which - 5] = v; break;
    case 9: p->has_best_actions = v != 0.0; break; case 10: p->has_best_deficit = v != 0.0; break;
    case 11: p->has_cw = v != 0.0; break;
    case 13: p->failed_episodes = uint32_t(v); break;
    default: return EG_ERR_BAD_ARG;
  }
  return EG_OK;
}
// list codes: 0 best_actions 1 best_deficit_actions 2 current_run_actions 3 current_deficit_actions 4 best_weights row
int32_t eg_policy_get_list(const eg_policy* p, int32_t which, int32_t yi, uint8_t* out, int32_t cap) {
  if (!p || yi < 0 || yi >= Y) return EG_ERR_BAD_ARG;
  const ActionList& l = which == 0 ? p->best_actions[yi] : which == 1 ? p->best_deficit[yi] : which == 2 ? p->cur_run[yi] : p->cur_def[yi];
  for (size_t i = 0; i < l.size() && int32_t(i) < cap; ++i) out[i] = l[i];
  return int32_t(l.size());
}

int32_t eg_policy_apply_episode(eg_policy* p, const double metrics[4], const int32_t* n_run, const uint8_t* run_log,
                                const int32_t* n_def, const uint8_t* def_log, uint64_t noise_seed) {
  if (!p || !metrics || !n_run || !n_def) return EG_ERR_BAD_ARG;
  HostRng noise(noise_seed);
  // transfer_recorded_actions_from (strategy.rs:313-342)
  for (int y = 0, rp = 0, dp = 0; y < Y; ++y) {
    p->cur_run[y].assign(run_log + rp, run_log + rp + n_run[y]); rp += n_run[y];
    p->cur_def[y].assign(def_log + dp, def_log + dp + n_def[y]); dp += n_def[y];
  }
  const double stall_f = double(p->stall);
  auto randomize = [&](double* row, int n) {
    for (int i = 0; i < n; ++i) {
      const double f = 1.0 + 0.25 * (noise.next_f64() * 2.0 - 1.0);
      double v = row[i] * f;
      row[i] = v < kMinW ? kMinW : (v > kMaxW ? kMaxW : v);
    }
  };
  // apply_contrast_learning (learning.rs:131-283)
  if (p->has_best && p->has_best_actions) {
    const double best_score = eg_score_metrics(p->best_metrics.data(), 0), current_score = eg_score_metrics(metrics, 0);
    const double deterioration = best_score > 0.0 ? (best_score - current_score) / best_score : 0.0;
    const double threshold = 0.1 * std::fmax(std::exp(-stall_f / 500.0), 0.00001 / 0.1);
    if (deterioration > threshold || p->stall > 800) {
      const double stagnation_factor = 1.0 + (0.2 * std::pow(stall_f / 10.0, 1.8));
      const double combined_penalty = std::pow(deterioration, 0.3) * stagnation_factor;
      const double adaptive_lr = p->learning_rate * (1.0 + 0.1 * stall_f);
      const double penalty_factor = 1.0 / (1.0 + adaptive_lr * 1.5 * combined_penalty);
      const double boost_factor = 1.0 + (adaptive_lr * 2.0 * stagnation_factor);
      for (int y = 0; y < Y; ++y) {
        ActionList current = p->cur_run[y]; current.insert(current.end(), p->cur_def[y].begin(), p->cur_def[y].end());
        ActionList best = p->best_actions[y];
        if (p->has_best_deficit) best.insert(best.end(), p->best_deficit[y].begin(), p->best_deficit[y].end());
        auto& row = p->w[y];
        for (uint8_t a : best) row[a] = std::fmin(row[a] * boost_factor, kMaxW);
        for (size_t i = 0; i < current.size(); ++i) {
          const uint8_t a = current[i];
          bool in_best = false;
          for (uint8_t b : best) if (b == a) { in_best = true; break; }
          if (!in_best) row[a] = std::fmax(row[a] * penalty_factor, kMinW);
          else if (i < best.size() && a != best[i]) {
            const double mild = 1.0 / (1.0 + adaptive_lr * combined_penalty * 0.5);
            row[a] = std::fmax(row[a] * mild, kMinW);
          }
        }
      }
      if (p->stall > 1200) for (int y = 0; y < Y; ++y) randomize(p->w[y].data(), NA);
    }
  }
  // update_best_strategy (strategy.rs:19-258)
  const double current_score = eg_score_metrics(metrics, 0);
  p->iteration_count += 1;
  if (!p->has_best || current_score > eg_score_metrics(p->best_metrics.data(), 0)) {
    p->record_improvement(current_score, metrics);
    p->has_best = true; for (int i = 0; i < 4; ++i) p->best_metrics[i] = metrics[i];
    p->has_best_weights = true; p->best_w = p->w;
    p->best_actions = p->cur_run; p->best_deficit = p->cur_def;
    p->has_best_actions = true; p->has_best_deficit = true;
    p->stall = 0;
  } else p->stall += 1;
  // apply_deficit_contrast_learning (learning.rs:285-373)
  if (p->has_best && p->has_best_deficit) {
    const double st = double(p->stall);
    const double deterioration = st / 10.0;
    const double threshold = 0.05 * std::fmax(std::exp(-st / 400.0), 0.00001 / 0.05);
    if (deterioration > threshold || p->stall > 800) {
      const double stagnation_factor = 1.0 + (0.2 * std::pow(st / 10.0, 1.8));
      const double combined_penalty = std::pow(deterioration, 0.3) * stagnation_factor;
      const double adaptive_lr = p->learning_rate * (1.0 + 0.1 * st);
      const double penalty_factor = 1.0 / (1.0 + adaptive_lr * 1.5 * combined_penalty);
      const double boost_factor = 1.0 + (adaptive_lr * 2.0 * stagnation_factor * 1.5);
      for (int y = 0; y < Y; ++y) {
        auto& row = p->dw[y];
        const ActionList& best = p->best_deficit[y];
        for (uint8_t a : best) { const int s = deficit_slot(a); if (s >= 0) row[s] = std::fmin(row[s] * boost_factor, kMaxW); }
        for (uint8_t a : p->cur_def[y]) {
          bool in_best = false;
          for (uint8_t b : best) if (b == a) { in_best = true; break; }
          if (!in_best) { const int s = deficit_slot(a); if (s >= 0) row[s] = std::fmax(row[s] * penalty_factor, kMinW); }
        }
      }
      if (p->stall > 1200) for (int y = 0; y < Y; ++y) randomize(p->dw[y].data(), ND);
    }
  }
  return EG_OK;
}

// Batch form of the three steps above ("reduced mode").  All episodes of the batch were sampled from
// the same snapshot, so they are all contrasted against the best strategy of that snapshot; their multiplicative
// nudges were summed in log space on the device (Q32 integers) and are applied once per stage — all boosts, clamp, all
// penalties, clamp (rm::nudge) — which for a batch of ONE episode is the sequential update itself.  Then the batch's best
// episode competes for the best slot, then the deficit table is contrasted with the stall counter that results — the
// same order as multi_simulation.rs:494-508.
int32_t eg_policy_apply_reduced(eg_policy* p, const int64_t* stats, const double cand_metrics[4], const int32_t* cand_n_run,
                                const uint8_t* cand_run_log, const int32_t* cand_n_def, const uint8_t* cand_def_log,
                                uint64_t noise_seed) {
  // Every formula comes from eg_reduced_math.h, which the device-side form of this update (k_apply_update) shares: the
  // two produce identical policies from identical packets.
  if (!p || !stats) return EG_ERR_BAD_ARG;
  HostRng noise(noise_seed);
  const int64_t n_ok = stats[0], n_qual = stats[2];
  const int64_t* pen = stats + 8; const int64_t* mild = stats + 8 + Y * NA; const int64_t* dcnt = stats + 8 + 2 * Y * NA;
  // the rollout kernels see the best lists only when both are present (eg_policy_snapshot_view); so does this update
  const bool have_lists = p->has_best && p->has_best_actions && p->has_best_deficit;
  if (have_lists && n_qual > 0) {   // apply_contrast_learning
    const double ln_boost = rm::contrast_ln_boost(p->learning_rate, p->stall);
    for (int y = 0; y < Y; ++y) {
      int occ[NA] = {0};
      for (uint8_t a : p->best_actions[y]) occ[a] += 1;
      for (uint8_t a : p->best_deficit[y]) occ[a] += 1;
      double* row = p->w[y].data();
      for (int a = 0; a < NA; ++a) {
        row[a] = rm::nudge(row[a], double(n_qual) * double(occ[a]) * ln_boost, (double(pen[y * NA + a]) + double(mild[y * NA + a])) / 4294967296.0);
      }
    }
    if (p->stall > 1200) for (int y = 0; y < Y; ++y) for (int a = 0; a < NA; ++a) p->w[y][a] = rm::noise(p->w[y][a], noise.next_f64());
  }
  // update_best_strategy with the batch's candidate
  p->iteration_count += uint32_t(n_ok);
  p->failed_episodes += uint32_t(stats[1]);
  bool improved = false;
  if (cand_metrics && cand_n_run && cand_n_def && n_ok > 0)
    improved = !p->has_best || rm::score(cand_metrics) > rm::score(p->best_metrics.data());
  if (improved) {
    p->record_improvement(rm::score(cand_metrics), cand_metrics);
    p->has_best = true; for (int i = 0; i < 4; ++i) p->best_metrics[i] = cand_metrics[i];
    p->has_best_weights = true; p->best_w = p->w;
    for (int y = 0, rp = 0, dp = 0; y < Y; ++y) {
      p->best_actions[y].assign(cand_run_log + rp, cand_run_log + rp + cand_n_run[y]); rp += cand_n_run[y];
      p->best_deficit[y].assign(cand_def_log + dp, cand_def_log + dp + cand_n_def[y]); dp += cand_n_def[y];
      p->cur_run[y] = p->best_actions[y]; p->cur_def[y] = p->best_deficit[y];
    }
    p->has_best_actions = true; p->has_best_deficit = true; p->stall = 0;
  } else p->stall += uint32_t(n_ok);
  // apply_deficit_contrast_learning: the same factor for every episode (it depends on the stall counter only)
  if (!improved && have_lists) {
    const rm::DeficitContrast dc = rm::deficit_contrast(p->learning_rate, p->stall);
    if (dc.active) {
      for (int y = 0; y < Y; ++y) {
        int occ[ND] = {0};
        for (uint8_t a : p->best_deficit[y]) { const int s = deficit_slot(a); if (s >= 0) occ[s] += 1; }
        double* row = p->dw[y].data();
        for (int s = 0; s < ND; ++s) {
          row[s] = rm::nudge(row[s], double(n_ok) * double(occ[s]) * dc.ln_boost, double(dcnt[y * ND + s]) * dc.ln_pen);
        }
      }
      if (p->stall > 1200) for (int y = 0; y < Y; ++y) for (int s = 0; s < ND; ++s) p->dw[y][s] = rm::noise(p->dw[y][s], noise.next_f64());
    }
  }
  return improved ? 1 : EG_OK;
}

int32_t eg_policy_apply_packet(eg_policy* p, const int64_t* stats, const void* candidates, int32_t n_candidates, uint64_t noise_seed) {
  if (!p || !stats || n_candidates < 0 || (n_candidates > 0 && !candidates)) return EG_ERR_BAD_ARG;
  const uint8_t* win = nullptr; double win_score = 0.0; int64_t win_index = 0;
  for (int r = 0; r < n_candidates; ++r) {
    const uint8_t* c = static_cast<const uint8_t*>(candidates) + size_t(r) * EG_CANDIDATE_BYTES;
    double score; int64_t index;
    std::memcpy(&score, c, 8); std::memcpy(&index, c + 8, 8);
    if (index < 0) continue;
    if (!win || score > win_score || (score == win_score && index < win_index)) { win = c; win_score = score; win_index = index; }
  }
  if (!win) return eg_policy_apply_reduced(p, stats, nullptr, nullptr, nullptr, nullptr, nullptr, noise_seed);
  double metrics[4]; int32_t n_run[EG_YEARS], n_def[EG_YEARS];
  std::memcpy(metrics, win + 16, 32); std::memcpy(n_run, win + 48, 4 * EG_YEARS); std::memcpy(n_def, win + 48 + 4 * EG_YEARS, 4 * EG_YEARS);
  const uint8_t* run_log = win + 48 + 8 * EG_YEARS; const uint8_t* def_log = run_log + EG_RUN_CAP;
  return eg_policy_apply_reduced(p, stats, metrics, n_run, run_log, n_def, def_log, noise_seed);
}

}  // extern "C"
