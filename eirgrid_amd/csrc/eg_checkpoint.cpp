// eg_checkpoint.cpp — ActionWeights checkpoints in the reference's JSON schema.
//
// Schema = serde's rendering of SerializableWeights (ai/learning/serialization.rs:38-51): year keys are strings, every
// table entry is a 2-element array [SerializableAction, weight] (ai/actions/serializable_action.rs:6-13), the best_*
// members are nullable, and action_count_weights is NOT part of the file — after a load the count table is absent and
// sample_additional_actions takes the heuristic branch, exactly as in the reference
// (ai/learning/weights/serialization.rs:474, sampling.rs:423-442).  Written pretty-printed with two-space indentation
// like serde_json::to_string_pretty (:133); HashMap order is unspecified in the reference, here years ascend and
// actions follow the canonical table order.  A file written here loads in the reference's load_from_file, and a
// file written by the reference loads here.
#include <cerrno>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <ctime>
#include <fstream>
#include <memory>
#include <sstream>

#include "eg_internal.h"
#include "eg_json.h"
#include "eg_policy_internal.h"

using eg::Json;
using eg::JsonParser;

namespace {

const char* kTypeName[EG_N_TYPES] = {"OnshoreWind", "OffshoreWind", "DomesticSolar", "CommercialSolar", "UtilitySolar", "Nuclear",
                                     "CoalPlant", "GasCombinedCycle", "GasPeaker", "Biomass", "HydroDam", "PumpedStorage",
                                     "BatteryStorage", "TidalGenerator", "WaveEnergy"};   // models/generator.rs:63-83
const char* kOffsetName[4] = {"Forest", "Wetland", "ActiveCapture", "CarbonCredit"};        // canonical order, core.rs:100-114
const int kMultPercent[3] = {100, 120, 150};
const int kDeficitAction[ND] = {24, 21, 36, 33, 27, 0, 3, 12, 30, 15, 6, 9, 39, 42, 60};   // deficit slot -> main-table index

// ---------------------------------------------------------------- writer
std::string fmt_f64(double v) {   // shortest representation that round-trips, with a ".0" on integers like serde_json/ryu
  if (!std::isfinite(v)) return "null";
  char buf[40];
  for (int p = 1; p <= 17; ++p) {
    std::snprintf(buf, sizeof(buf), "%.*g", p, v);
    if (std::strtod(buf, nullptr) == v) break;
  }
  std::string s(buf);
  if (s.find_first_of(".eEn") == std::string::npos) s += ".0";
  return s;
}
std::string fmt_display(double v) {      // Rust `{}` of an f64: shortest digits that round-trip, never an exponent, "5" for 5.0
  if (std::isnan(v)) return "NaN";
  if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
  char buf[400];
  auto r = std::to_chars(buf, buf + sizeof(buf), v, std::chars_format::fixed);
  return std::string(buf, r.ptr);
}
struct Writer {
  std::string out; int depth = 0;
  void nl() { out += '\n'; out.append(size_t(depth) * 2, ' '); }
  void action(int a) {   // SerializableAction::from (serializable_action.rs:15-66)
    const char* type; std::string gen = "null", id = "null", pct = "null", off = "null", mult = "null";
    if (a < 45) { type = "AddGenerator"; gen = std::string("\"") + kTypeName[a / 3] + "\""; mult = std::to_string(kMultPercent[a % 3]); }
    else if (a < 57) { type = "AddCarbonOffset"; off = std::string("\"") + kOffsetName[(a - 45) / 3] + "\""; mult = std::to_string(kMultPercent[(a - 45) % 3]); }
    else if (a == 57) { type = "UpgradeEfficiency"; id = "\"\""; }
    else if (a == 58) { type = "AdjustOperation"; id = "\"\""; pct = "0"; }
    else if (a == 59) { type = "CloseGenerator"; id = "\"\""; }
    else type = "DoNothing";
    out += '{'; ++depth;
    nl(); out += std::string("\"action_type\": \"") + type + "\",";
    nl(); out += "\"generator_type\": " + gen + ",";
    nl(); out += "\"generator_id\": " + id + ",";
    nl(); out += "\"operation_percentage\": " + pct + ",";
    nl(); out += "\"offset_type\": " + off + ",";
    nl(); out += "\"cost_multiplier\": " + mult;
    --depth; nl(); out += '}';
  }
  template <typename Row>
  void table(const Row* rows, int n, const int* action_of) {   // HashMap<u32, Vec<(SerializableAction, f64)>>
    out += '{'; ++depth;
    for (int y = 0; y < Y; ++y) {
      nl(); out += "\"" + std::to_string(2025 + y) + "\": ["; ++depth;
      for (int i = 0; i < n; ++i) {
        nl(); out += '['; ++depth; nl(); action(action_of ? action_of[i] : i); out += ','; nl(); out += fmt_f64(rows[y][i]); --depth; nl(); out += ']';
        if (i + 1 < n) out += ',';
      }
      --depth; nl(); out += ']';
      if (y + 1 < Y) out += ',';
    }
    --depth; nl(); out += '}';
  }
  void lists(const std::array<ActionList, Y>& l) {   // HashMap<u32, Vec<SerializableAction>>
    out += '{'; ++depth;
    for (int y = 0; y < Y; ++y) {
      nl(); out += "\"" + std::to_string(2025 + y) + "\": [";
      if (!l[y].empty()) {
        ++depth;
        for (size_t i = 0; i < l[y].size(); ++i) { nl(); action(l[y][i]); if (i + 1 < l[y].size()) out += ','; }
        --depth; nl();
      }
      out += ']';
      if (y + 1 < Y) out += ',';
    }
    --depth; nl(); out += '}';
  }
};

int action_index(const Json& a) {   // inverse of Writer::action; -1 = not representable in the canonical table
  const Json* t = a.get("action_type");
  if (!t || t->kind != Json::Str) return -1;
  const Json* mult = a.get("cost_multiplier");
  int m = 0;
  if (mult && mult->kind == Json::Num) { int pc = int(mult->num); m = pc == 100 ? 0 : pc == 120 ? 1 : pc == 150 ? 2 : -1; }
  if (t->str == "AddGenerator") {
    const Json* g = a.get("generator_type");
    if (!g || g->kind != Json::Str) return 3 * 8;            // serialization.rs:163-165: GasPeaker, default multiplier
    for (int i = 0; i < EG_N_TYPES; ++i) if (g->str == kTypeName[i]) return m < 0 ? -1 : 3 * i + m;
    return -2;                                               // unknown generator type: an error in the reference too
  }
  if (t->str == "AddCarbonOffset") {
    const Json* o = a.get("offset_type");
    int ot = 0;                                              // unknown names fall back to Forest (serialization.rs:186)
    if (o && o->kind == Json::Str) for (int i = 0; i < 4; ++i) if (o->str == kOffsetName[i]) ot = i;
    return m < 0 ? -1 : 45 + 3 * ot + m;
  }
  const Json* id = a.get("generator_id");
  const bool empty_id = !id || id->kind != Json::Str || id->str.empty();
  if (t->str == "UpgradeEfficiency") return empty_id ? 57 : -1;
  if (t->str == "AdjustOperation") return empty_id ? 58 : -1;
  if (t->str == "CloseGenerator") return empty_id ? 59 : -1;
  if (t->str == "DoNothing") return 60;
  return -2;
}

}  // namespace

extern "C" {

int32_t eg_policy_save_json(const eg_policy* p, const char* path) {   // ai/learning/weights/serialization.rs:29-139
  if (!p || !path) return EG_ERR_BAD_ARG;
  Writer w;
  w.out += '{'; ++w.depth;
  w.nl(); w.out += "\"weights\": "; w.table(p->w.data(), NA, nullptr); w.out += ',';
  w.nl(); w.out += "\"learning_rate\": " + fmt_f64(p->learning_rate) + ",";
  w.nl(); w.out += "\"best_metrics\": ";
  if (p->has_best) {
    w.out += '{'; ++w.depth;
    w.nl(); w.out += "\"final_net_emissions\": " + fmt_f64(p->best_metrics[0]) + ",";
    w.nl(); w.out += "\"average_public_opinion\": " + fmt_f64(p->best_metrics[1]) + ",";
    w.nl(); w.out += "\"total_cost\": " + fmt_f64(p->best_metrics[2]) + ",";
    w.nl(); w.out += "\"power_reliability\": " + fmt_f64(p->best_metrics[3]);
    --w.depth; w.nl(); w.out += '}';
  } else w.out += "null";
  w.out += ',';
  w.nl(); w.out += "\"best_weights\": "; if (p->has_best_weights) w.table(p->best_w.data(), NA, nullptr); else w.out += "null"; w.out += ',';
  w.nl(); w.out += "\"best_actions\": "; if (p->has_best_actions) w.lists(p->best_actions); else w.out += "null"; w.out += ',';
  w.nl(); w.out += "\"iteration_count\": " + std::to_string(p->iteration_count) + ",";
  w.nl(); w.out += "\"iterations_without_improvement\": " + std::to_string(p->stall) + ",";
  w.nl(); w.out += "\"exploration_rate\": " + fmt_f64(p->exploration_rate) + ",";
  w.nl(); w.out += "\"deficit_weights\": "; w.table(p->dw.data(), ND, kDeficitAction); w.out += ',';
  w.nl(); w.out += "\"best_deficit_actions\": "; if (p->has_best_deficit) w.lists(p->best_deficit); else w.out += "null"; w.out += ',';
  w.nl(); w.out += "\"optimization_mode\": null,";     // always None inside ActionWeights (Q3)
  w.nl(); w.out += "\"improvement_history\": ";
  if (p->improvement_history.empty()) w.out += "null";
  else {
    w.out += '['; ++w.depth;
    for (size_t i = 0; i < p->improvement_history.size(); ++i) {
      const ImprovementRecord& r = p->improvement_history[i];
      w.nl(); w.out += '{'; ++w.depth;
      w.nl(); w.out += "\"iteration\": " + std::to_string(r.iteration) + ",";
      w.nl(); w.out += "\"score\": " + fmt_f64(r.score) + ",";
      w.nl(); w.out += "\"net_emissions\": " + fmt_f64(r.net_emissions) + ",";
      w.nl(); w.out += "\"total_cost\": " + fmt_f64(r.total_cost) + ",";
      w.nl(); w.out += "\"public_opinion\": " + fmt_f64(r.public_opinion) + ",";
      w.nl(); w.out += "\"power_reliability\": " + fmt_f64(r.power_reliability) + ",";
      w.nl(); w.out += "\"timestamp\": \"" + r.timestamp + "\"";
      --w.depth; w.nl(); w.out += '}';
      if (i + 1 < p->improvement_history.size()) w.out += ',';
    }
    --w.depth; w.nl(); w.out += ']';
  }
  --w.depth; w.nl(); w.out += '}';
  std::ofstream f(path, std::ios::binary | std::ios::trunc);
  if (!f) { eg::set_error(std::string("eg_policy_save_json: cannot open ") + path); return EG_ERR_BAD_ARG; }
  f << w.out;
  return f.good() ? EG_OK : EG_ERR_BAD_ARG;
}

// --track-weight-history (core/multi_simulation.rs:166-207, :557-560): weight_history.json is a pretty-printed array that
// gains one snapshot {best_score, iteration, timestamp, weights: ActionWeights::to_json()} per checkpoint.  serde_json's
// Map is a BTreeMap here (no preserve_order feature, Cargo.toml:11), so every object lists its keys in byte order; the
// table keys are the Display strings of the actions (ai/actions/grid_action.rs:18-41).
int32_t eg_policy_append_weight_history(const eg_policy* p, const char* path, uint64_t iteration) {
  if (!p || !path) return EG_ERR_BAD_ARG;
  auto display = [](int a) {
    if (a < 45) return std::string("AddGenerator(") + kTypeName[a / 3] + ", " + std::to_string(kMultPercent[a % 3]) + "%)";
    if (a < 57) return std::string("AddCarbonOffset(") + kOffsetName[(a - 45) / 3] + ", " + std::to_string(kMultPercent[(a - 45) % 3]) + "%)";
    if (a == 57) return std::string("UpgradeEfficiency()");
    if (a == 58) return std::string("AdjustOperation(, 0%)");
    if (a == 59) return std::string("CloseGenerator()");
    return std::string("DoNothing");
  };
  Writer w; w.depth = 1;
  auto sorted_table = [&](int n, auto key_of, auto value_of) {      // {"2025": {key: value, ...}, ...}, keys in byte order
    w.out += '{'; ++w.depth;
    for (int y = 0; y < Y; ++y) {
      std::vector<std::pair<std::string, double>> kv;
      for (int i = 0; i < n; ++i) kv.emplace_back(key_of(i), value_of(y, i));
      std::sort(kv.begin(), kv.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
      w.nl(); w.out += "\"" + std::to_string(2025 + y) + "\": {"; ++w.depth;
      for (size_t i = 0; i < kv.size(); ++i) { w.nl(); w.out += "\"" + kv[i].first + "\": " + fmt_f64(kv[i].second); if (i + 1 < kv.size()) w.out += ','; }
      --w.depth; w.nl(); w.out += '}';
      if (y + 1 < Y) w.out += ',';
    }
    --w.depth; w.nl(); w.out += '}';
  };
  const double best_score = p->has_best ? eg_score_metrics(p->best_metrics.data(), 0) : 0.0;
  char stamp[64];
  {  // chrono Local::now().to_rfc3339(): nanoseconds and the local UTC offset
    timespec ts; clock_gettime(CLOCK_REALTIME, &ts);
    std::tm tmv; localtime_r(&ts.tv_sec, &tmv);
    char date[32]; std::strftime(date, sizeof(date), "%Y-%m-%dT%H:%M:%S", &tmv);
    const long off = tmv.tm_gmtoff; const long ao = off < 0 ? -off : off;
    std::snprintf(stamp, sizeof(stamp), "%s.%09ld%c%02ld:%02ld", date, ts.tv_nsec, off < 0 ? '-' : '+', ao / 3600, (ao / 60) % 60);
  }
  w.nl(); w.out += '{'; ++w.depth;
  w.nl(); w.out += "\"best_score\": " + fmt_f64(best_score) + ",";
  w.nl(); w.out += "\"iteration\": " + std::to_string(iteration) + ",";
  w.nl(); w.out += std::string("\"timestamp\": \"") + stamp + "\",";
  w.nl(); w.out += "\"weights\": {"; ++w.depth;
  w.nl(); w.out += "\"action_count_weights\": ";
  if (p->has_cw) sorted_table(NC, [](int i) { return std::to_string(i); }, [&](int y, int i) { return p->cw[y][i]; }); else w.out += "{}";
  w.out += ',';
  w.nl(); w.out += "\"best_score\": " + fmt_f64(best_score) + ",";
  w.nl(); w.out += "\"deficit_weights\": "; sorted_table(ND, [&](int i) { return display(kDeficitAction[i]); }, [&](int y, int i) { return p->dw[y][i]; }); w.out += ',';
  w.nl(); w.out += "\"exploration_rate\": " + fmt_f64(p->exploration_rate) + ",";
  w.nl(); w.out += "\"force_best_actions\": false,";
  w.nl(); w.out += "\"guaranteed_best_actions\": false,";
  w.nl(); w.out += "\"iteration_count\": " + std::to_string(p->iteration_count) + ",";
  w.nl(); w.out += "\"iterations_without_improvement\": " + std::to_string(p->stall) + ",";
  w.nl(); w.out += "\"learning_rate\": " + fmt_f64(p->learning_rate) + ",";
  w.nl(); w.out += "\"optimization_mode\": null,";
  w.nl(); w.out += "\"weights\": "; sorted_table(NA, display, [&](int y, int i) { return p->w[y][i]; });
  --w.depth; w.nl(); w.out += '}';
  --w.depth; w.nl(); w.out += '}';
  // the file is "[]" or "[\n  {...},\n  {...}\n]": append in place instead of re-parsing the whole history
  std::string text;
  { std::ifstream f(path, std::ios::binary); if (f) { std::stringstream ss; ss << f.rdbuf(); text = ss.str(); } }
  size_t end = text.find_last_of(']');
  std::string head = end == std::string::npos ? std::string("[") : text.substr(0, end);
  while (!head.empty() && (head.back() == '\n' || head.back() == ' ')) head.pop_back();
  const bool empty = head == "[" || head.empty();
  if (head.empty()) head = "[";
  std::ofstream f(path, std::ios::binary | std::ios::trunc);
  if (!f) { eg::set_error(std::string("eg_policy_append_weight_history: cannot open ") + path); return EG_ERR_BAD_ARG; }
  f << head << (empty ? "" : ",") << w.out << "\n]";
  return f.good() ? EG_OK : EG_ERR_BAD_ARG;
}

// improvement_history.csv of the best-run export (utils/csv_export.rs:155-207): one row per recorded improvement, the
// score improvement relative to the previous record in per cent.  Nothing is written when there is no history.
int32_t eg_policy_export_improvement_csv(const eg_policy* p, const char* path) {
  if (!p || !path) return EG_ERR_BAD_ARG;
  if (p->improvement_history.empty()) return EG_OK;
  std::ofstream f(path, std::ios::binary | std::ios::trunc);
  if (!f) { eg::set_error(std::string("eg_policy_export_improvement_csv: cannot open ") + path); return EG_ERR_BAD_ARG; }
  f << "Iteration,Score,Net Emissions (tonnes),Total Cost (\xE2\x82\xAC),Public Opinion (%),Power Reliability (%),Score Improvement (%),Timestamp\n";
  double prev = 0.0;
  char line[512];
  for (size_t i = 0; i < p->improvement_history.size(); ++i) {
    const ImprovementRecord& r = p->improvement_history[i];
    const double improvement = (i > 0 && prev > 0.0) ? ((r.score - prev) / prev) * 100.0 : 0.0;
    std::snprintf(line, sizeof(line), "%u,%.6f,%.2f,%.2f,%.2f,%.2f,%.2f,%s\n", r.iteration, r.score, r.net_emissions, r.total_cost,
                  r.public_opinion * 100.0, r.power_reliability * 100.0, improvement, r.timestamp.c_str());
    f << line;
    prev = r.score;
  }
  return f.good() ? EG_OK : EG_ERR_BAD_ARG;
}

// simulation_summary.csv of the best-run export (utils/csv_export.rs:215-432): final metrics, one row per action of
// SimulationResult.actions with the exporter's cost estimate, one row per year of YearlyMetrics.  `run` holds ONE episode
// (the record eg_fetch_best_result / eg_fetch_record return; metrics, yearly, n_act and act_log are read).
int32_t eg_export_summary_csv(const eg_episode_out* run, const char* path, const char* timestamp) {
  if (!run || !path || !run->metrics || !run->yearly || !run->n_act || !run->act_log) {
    eg::set_error("eg_export_summary_csv: metrics, yearly, n_act and act_log are required"); return EG_ERR_BAD_ARG;
  }
  std::ofstream f(path, std::ios::binary | std::ios::trunc);
  if (!f) { eg::set_error(std::string("eg_export_summary_csv: cannot open ") + path); return EG_ERR_BAD_ARG; }
  char line[1024];
  f << "Simulation Summary\n" << "Timestamp," << (timestamp ? timestamp : "") << "\n\n";
  f << "Final Metrics\n";
  f << "Final Net Emissions (tonnes CO2)," << fmt_display(run->metrics[0]) << "\n";
  std::snprintf(line, sizeof(line), "Average Public Opinion (%%),%.2f\nTotal Cost (\xE2\x82\xAC),%.2f\nPower Reliability (%%),%.2f\n\n",
                run->metrics[1] * 100.0, run->metrics[2], run->metrics[3] * 100.0);
  f << line;
  f << "Actions Taken\n";
  f << "Year,Action Type,Generator Type,Generator ID,Operation %,Offset Type,Estimated Cost (\xE2\x82\xAC)\n";
  size_t k = 0;
  for (int y = 0; y < Y; ++y) {
    for (int i = 0; i < run->n_act[y]; ++i, ++k) {
      if (k >= size_t(EG_ACT_CAP)) { eg::set_error("eg_export_summary_csv: action list longer than EG_ACT_CAP"); return EG_ERR_BAD_ARG; }
      const int a = run->act_log[k];
      const char* kind = "DoNothing"; const char* gen = ""; const char* op = ""; const char* off = "";
      if (a < 45) { kind = "AddGenerator"; gen = kTypeName[a / 3]; }
      else if (a < 57) { kind = "AddCarbonOffset"; off = kOffsetName[(a - 45) / 3]; }
      else if (a == 57) kind = "UpgradeEfficiency";
      else if (a == 58) { kind = "AdjustOperation"; op = "0"; }      // AdjustOperation(String::new(), 0), core.rs:118
      else if (a == 59) kind = "CloseGenerator";
      std::snprintf(line, sizeof(line), "%d,%s,%s,,%s,%s,%.2f\n", 2025 + y, kind, gen, op, off, eg::action_cost_estimate(a, y));
      f << line;
    }
  }
  f << "\nYearly Summary Metrics\n";
  f << "Year,Population,PowerUsage,PowerGeneration,PowerBalance,PublicOpinion,YearlyCapitalCost,TotalCapitalCost,Inflation,CO2Emissions,"
       "CarbonOffset,NetEmissions,YearlyRevenue,TotalRevenue,ActiveGenerators,YearlyUpgradeCosts,YearlyClosureCosts,YearlyTotalCost,TotalCost\n";
  for (int y = 0; y < Y; ++y) {
    const double* r = run->yearly + size_t(y) * EG_YEARLY_FIELDS;
    std::snprintf(line, sizeof(line), "%llu,%llu,%.2f,%.2f,%.2f,%.4f,%.2f,%.2f,%.4f,%.2f,%.2f,%.2f,%.2f,%.2f,%llu,%.2f,%.2f,%.2f,%.2f\n",
                  (unsigned long long)r[EG_Y_YEAR], (unsigned long long)r[EG_Y_POP], r[EG_Y_USAGE], r[EG_Y_GEN], r[EG_Y_BALANCE], r[EG_Y_OPINION],
                  r[EG_Y_YEARLY_CAPITAL], r[EG_Y_TOTAL_CAPITAL], r[EG_Y_INFLATION], r[EG_Y_CO2], r[EG_Y_OFFSET], r[EG_Y_NET_CO2],
                  r[EG_Y_YEARLY_CREDIT], r[EG_Y_TOTAL_CREDIT], (unsigned long long)r[EG_Y_ACTIVE_GENS], r[EG_Y_UPGRADE_COSTS],
                  r[EG_Y_CLOSURE_COSTS], r[EG_Y_YEARLY_TOTAL_COST], r[EG_Y_TOTAL_COST]);
    f << line;
  }
  return f.good() ? EG_OK : EG_ERR_BAD_ARG;
}

eg_policy* eg_policy_load_json(const char* path) {   // ai/learning/weights/serialization.rs:140-493
  if (!path) return nullptr;
  std::ifstream f(path, std::ios::binary);
  if (!f) { eg::set_error(std::string("eg_policy_load_json: cannot open ") + path); return nullptr; }
  std::stringstream ss; ss << f.rdbuf();
  const std::string text = ss.str();
  JsonParser ps{text.data(), text.data() + text.size(), {}};
  Json root;
  if (!ps.value(root) || root.kind != Json::Obj) { eg::set_error("eg_policy_load_json: " + (ps.err.empty() ? std::string("not an object") : ps.err)); return nullptr; }
  std::unique_ptr<eg_policy> p(eg_policy_new());
  std::string problem;
  auto load_table = [&](const Json* t, auto& rows, int n, bool deficit, bool strict) -> bool {
    if (!t || t->kind != Json::Obj) return false;
    for (auto& kv : t->obj) {
      const int y = std::atoi(kv.first.c_str()) - 2025;
      if (y < 0 || y >= Y || kv.second.kind != Json::Arr) continue;
      for (const Json& entry : kv.second.arr) {
        if (entry.kind != Json::Arr || entry.arr.size() != 2 || entry.arr[1].kind != Json::Num) continue;
        int a = action_index(entry.arr[0]);
        if (a == -2 && strict) { problem = "unknown action or generator type in weights"; return false; }   // InvalidData in the reference
        if (a < 0) continue;
        if (deficit) { int slot = -1; for (int i = 0; i < ND; ++i) if (kDeficitAction[i] == a) slot = i; a = slot; if (a < 0) continue; }
        if (a < n) rows[y][a] = entry.arr[1].num;
      }
    }
    return true;
  };
  if (!load_table(root.get("weights"), p->w, NA, false, true)) { eg::set_error("eg_policy_load_json: " + (problem.empty() ? std::string("missing weights") : problem)); return nullptr; }
  load_table(root.get("deficit_weights"), p->dw, ND, true, false);   // absent/empty -> defaults (serialization.rs:262-279)
  auto num = [&](const char* k, double d) { const Json* j = root.get(k); return j && j->kind == Json::Num ? j->num : d; };
  p->learning_rate = num("learning_rate", 0.2); p->exploration_rate = num("exploration_rate", 0.2);
  p->iteration_count = uint32_t(num("iteration_count", 0)); p->stall = uint32_t(num("iterations_without_improvement", 0));
  if (const Json* bm = root.get("best_metrics"); bm && bm->kind == Json::Obj) {
    const char* keys[4] = {"final_net_emissions", "average_public_opinion", "total_cost", "power_reliability"};
    p->has_best = true;
    for (int i = 0; i < 4; ++i) { const Json* v = bm->get(keys[i]); p->best_metrics[i] = v && v->kind == Json::Num ? v->num : 0.0; }
  }
  if (const Json* bw = root.get("best_weights"); bw && bw->kind == Json::Obj) { p->best_w = p->w; p->has_best_weights = load_table(bw, p->best_w, NA, false, false); }
  auto load_lists = [&](const Json* t, std::array<ActionList, Y>& l) -> bool {
    if (!t || t->kind != Json::Obj) return false;
    for (auto& kv : t->obj) {
      const int y = std::atoi(kv.first.c_str()) - 2025;
      if (y < 0 || y >= Y || kv.second.kind != Json::Arr) continue;
      l[y].clear();
      for (const Json& a : kv.second.arr) { const int idx = action_index(a); if (idx >= 0) l[y].push_back(uint8_t(idx)); }
    }
    return true;
  };
  p->has_best_actions = load_lists(root.get("best_actions"), p->best_actions);
  p->has_best_deficit = load_lists(root.get("best_deficit_actions"), p->best_deficit);
  if (const Json* h = root.get("improvement_history"); h && h->kind == Json::Arr)
    for (const Json& r : h->arr) {
      if (r.kind != Json::Obj) continue;
      auto g = [&](const char* k) { const Json* v = r.get(k); return v && v->kind == Json::Num ? v->num : 0.0; };
      const Json* ts = r.get("timestamp");
      p->improvement_history.push_back({uint32_t(g("iteration")), g("score"), g("net_emissions"), g("total_cost"), g("public_opinion"),
                                        g("power_reliability"), ts && ts->kind == Json::Str ? ts->str : std::string()});
    }
  p->has_cw = false;   // action_count_weights is not part of the file (serialization.rs:474)
  return p.release();
}

}  // extern "C"
