// eg_reduced_math.h — scalar formulas of the batch ("reduced") update, shared by the host (eg_policy.cpp, eg_api.cpp)
// and the device (k_apply_update, the statistics epilogue).  Every transcendental goes through include/eg_detpow.h
// (IEEE + - * / and bit operations only, no FMA contraction), so the host and the device compute the same bits and a
// policy that is updated on the device stays identical to one updated on the host from the same packets.
//
// Define EG_RM before including (`static inline` on the host, `__device__ __forceinline__` on the device); the file
// includes eg_detpow.h with the same qualifier unless it was included already.
#ifndef EG_REDUCED_MATH_H
#define EG_REDUCED_MATH_H

#ifndef EG_RM
#define EG_RM static inline
#endif
#ifndef EG_DETPOW_QUAL
#define EG_DETPOW_QUAL EG_RM
#endif
#include "eg_detpow.h"
#include "eg_internal.h"

namespace eg {
namespace rm {

constexpr double kMinW = 0.0001, kMaxW = 0.999;                 // ai/learning/constants.rs:14-15
constexpr double kMaxCost = 50000000000.0, kMaxEmissions = 1000000.0;   // config/constants.rs:115, scoring.rs

EG_RM double clampw(double v) { return v < kMinW ? kMinW : (v > kMaxW ? kMaxW : v); }
EG_RM double dmaxd(double a, double b) { return a > b ? a : b; }
EG_RM double dmind(double a, double b) { return a < b ? a : b; }
EG_RM double powd(double x, double p) { return x > 0.0 ? eg_detpow(x, p) : 0.0; }         // x >= 0
EG_RM double expd(double y) { return eg_detexp(y < -700.0 ? -700.0 : (y > 700.0 ? 700.0 : y)); }
// exp of a log-space nudge.  |L| is cut at 20: a weight lives in [1e-4, 0.999], so a factor beyond e^+-9.3 ends at
// the clamp either way.
EG_RM double exp_nudge(double L) { return eg_detexp(L < -20.0 ? -20.0 : (L > 20.0 ? 20.0 : L)); }

// ai/metrics/scoring.rs:18-44 (mode None) with the shared logarithm
EG_RM double score(const double* m) {
  if (m[0] > 0.0) return 1.0 - dmind(m[0] / kMaxEmissions, 1.0);
  const double normalized_cost = dmaxd(m[2] / kMaxCost, 1.0);
  const double cost_score = 1.0 - dmind(eg_detlog(normalized_cost) / eg_detlog(kMaxCost * 100.0 / kMaxCost), 1.0);
  const double cost_weight = normalized_cost > 8.0 ? 0.8 : 0.5;
  return 1.0 + (cost_score * cost_weight + m[1] * (1.0 - cost_weight));
}

// Everything a rollout / statistics kernel needs that depends only on the policy's scalars (learning.rs:37-55, :82,
// :131-180; sampling.rs:425-427), evaluated wherever the policy changes: on the host at upload, on the device after an
// on-device update.  The list flags and counters of `s` must be set by the caller.
// In four parts that write disjoint fields: the host runs them one after the other; k_apply_update gives each of the three with a
// transcendental in it (a logarithm, a square root by powd, an exponential and a power: IEEE-only evaluations, about a microsecond or two
// of one thread each) to a wave of its own.
EG_RM void derive_state_score(DevState& s) {
  const double best_score = s.has_best ? score(s.best_metrics) : 0.0;
  // learning.rs:37-55: "relative improvement" compares the best score with itself (Q4)
  const double final_impact = best_score;
  double rel = final_impact;
  if (s.has_best) rel = best_score > 0.0 ? (final_impact - best_score) / best_score : final_impact;
  s.rel_improvement = rel;
  s.immediate_weight = rel > 0.0 ? 0.7 : 0.3;
  s.p_best_score = best_score;
}
EG_RM void derive_state_heur(DevState& s) {
  const double scaled = powd(s.exploration_rate, 0.5);                                                     // sampling.rs:425-427
  const double lo = 2.0 / scaled, hi = 12.0 / scaled;
  s.heur_min = (uint32_t)(long long)(lo + 0.5); s.heur_max = (uint32_t)(long long)(hi + 0.5);               // f64::round, x > 0
}
EG_RM void derive_state_contrast(DevState& s) {
  const double k = (double)s.stall;
  s.p_threshold = 0.1 * dmaxd(expd(-k / 500.0), 0.00001 / 0.1);                                           // learning.rs:146-154
  s.p_stagnation = 1.0 + (0.2 * powd(k / 10.0, 1.8));                                                     // learning.rs:163-164
}
EG_RM void derive_state_rest(DevState& s) {
  const double k = (double)s.stall;
  s.noop_boost = (s.has_best && s.best_metrics[0] <= 0.0 && s.best_metrics[2] > kMaxCost * 8.0) ? 1 : 0;   // learning.rs:82
  s.p_forced = s.stall > 800u ? 1 : 0;
  s.p_adaptive_lr = s.learning_rate * (1.0 + 0.1 * k);                                                    // learning.rs:174
  s.boost_others = 1.0 + (s.learning_rate * 0.1); s.boost_noop = 1.0 + s.learning_rate * 0.2;             // learning.rs:74-87
  s.eps_main = s.stall > 100u ? s.exploration_rate * (1.0 / (1.0 + 0.01 * k)) : s.exploration_rate;       // sampling.rs:150-157
  s.scaled_power = 1.0 + (2.0 * dmind(k / 1000.0, 3.0));                                                  // sampling.rs:193-195
}
EG_RM void derive_state(DevState& s) { derive_state_score(s); derive_state_heur(s); derive_state_contrast(s); derive_state_rest(s); }

// apply_contrast_learning in log space (learning.rs:131-255): the boost of one occurrence in the best lists
EG_RM double contrast_ln_boost(double learning_rate, uint32_t stall) {
  const double k = (double)stall;
  const double stagnation_factor = 1.0 + (0.2 * powd(k / 10.0, 1.8));
  const double adaptive_lr = learning_rate * (1.0 + 0.1 * k);
  return eg_detlog(1.0 + (adaptive_lr * 2.0 * stagnation_factor));
}

// apply_deficit_contrast_learning (learning.rs:285-373): active?, ln(penalty factor), ln(boost factor)
struct DeficitContrast { int active; double ln_pen, ln_boost; };
EG_RM DeficitContrast deficit_contrast(double learning_rate, uint32_t stall) {
  DeficitContrast d;
  const double st = (double)stall;
  const double deterioration = st / 10.0;
  const double threshold = 0.05 * dmaxd(expd(-st / 400.0), 0.00001 / 0.05);
  d.active = (deterioration > threshold || stall > 800u) ? 1 : 0;
  const double stagnation_factor = 1.0 + (0.2 * powd(st / 10.0, 1.8));
  const double combined_penalty = powd(deterioration, 0.3) * stagnation_factor;
  const double adaptive_lr = learning_rate * (1.0 + 0.1 * st);
  d.ln_pen = eg_detlog(1.0 / (1.0 + adaptive_lr * 1.5 * combined_penalty));
  d.ln_boost = eg_detlog(1.0 + (adaptive_lr * 2.0 * stagnation_factor * 1.5));
  return d;
}

// One entry of the main / deficit table after a batch.  The sequential form (learning.rs:214-252, :339-353) first boosts
// every occurrence in the best lists (cap MAX_WEIGHT after each factor), then penalises (floor MIN_WEIGHT after each
// factor); factors >= 1 under a cap and factors <= 1 over a floor commute with their clamp, so the batch form keeps exactly
// that order with ONE clamp per stage: all boosts (Lb >= 0), clamp, all penalties (Lp <= 0), clamp.  A batch of one
// episode is then the sequential update itself (to the Q32 rounding of the summed logarithms).  A stage with a zero
// exponent leaves the entry untouched.
EG_RM double nudge(double w, double Lb, double Lp) {
  double v = w;
  if (Lb != 0.0) v = clampw(v * exp_nudge(Lb));
  if (Lp != 0.0) v = clampw(v * exp_nudge(Lp));
  return v;
}
// An episode that BEATS the best while contrast is forced (stall > 800) has a negative deterioration: the reference's
// deterioration.powf(0.3) is NaN, so is every penalty factor derived from it, and (w * NaN).max(MIN_WEIGHT) is MIN_WEIGHT
// (f64::max returns the other operand).  In log space that is any exponent below ln(MIN/MAX) = -9.2: -20 per occurrence.
constexpr double kLnNanPenalty = -20.0;
// stagnation noise of one entry from a uniform draw u in [0, 1) (learning.rs:267-280, :356-369)
EG_RM double noise(double w, double u) { return clampw(w * (1.0 + 0.25 * (u * 2.0 - 1.0))); }

}  // namespace rm
}  // namespace eg

#endif
