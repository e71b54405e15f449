// eg_export.cpp — the detail files of the reference's best-run export, host code only:
//   <dir>/yearly_details/settlements.csv       utils/csv_export.rs:457-533
//   <dir>/yearly_details/generators.csv        utils/csv_export.rs:535-985
//   <dir>/yearly_details/carbon_offsets.csv    utils/csv_export.rs:987-1092
//   <dir>/operation_logs/generator_operation_logs.csv   utils/csv_export.rs:1094-1230
// written from the record of the best episode (eg_fetch_best_result / eg_fetch_record) and the world description.
//
// What the reference's exporter really emits (core/multi_simulation.rs:852-905 builds a "final map" = the base map with the
// best episode's SAMPLED actions re-applied, and hands it over with the episode's YearlyMetrics):
//  * settlements.csv — every settlement of the map in every year, population and usage extrapolated from the 2025 values
//    (round(pop * 1.01^k), usage * 1.02^k), not the simulated ones.
//  * generators.csv — `Generator.eol` holds a LIFESPAN (15..100 years: generator.rs:344-362 passed as `eol`, actions.rs:68,
//    generators_loader.rs:194), so the first pass's `year > eol` test (csv_export.rs:704-708) skips every generator of the
//    map in every year: all rows come from the second pass (:807-981), one per year and generator id listed in that year's
//    YearlyMetrics.generator_efficiencies — the generators ACTIVE in the simulated episode — with values estimated from the id
//    string: type by substring, commissioning year = third '_' field (the running index for "Existing_<type>_<n>" ids),
//    nameplate / CO2 / size / cost defaults per type, 99.00 % efficiency, operation 100 (a u8 percentage) x 100 = 10000.00,
//    real coordinates for "Existing_" ids, coordinates hashed from the id otherwise.  The generators re-applied to the final
//    map influence nothing.  Row order within a year is HashMap order in the reference; here: map order (existing plant,
//    then the episode's generators in the order they were added).
//  * carbon_offsets.csv — the offsets of the final map: the sampled AddCarbonOffset actions, ids Offset_<type>_<year>_<n>,
//    from their year on.  The base map has construction delays ON (Q1) and nobody advances the final map's clock, so every
//    offset is still "Planned": CO2 offset 0.00 / -0.00, cost per tonne 0.00.  Their coordinates come from thread_rng in the
//    reference (actions.rs:142-145); here from StdRng::seed_from_u64(offset_seed): x = gen::<f64>() * 50000, then y.
//  * generator_operation_logs.csv — header only: the year loop runs commissioning_year..=min(eol, 2050) with eol a lifespan.
#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include <sys/stat.h>

#include "eg_internal.h"

namespace {

const char* kTypeNames[EG_N_TYPES] = {"OnshoreWind", "OffshoreWind", "DomesticSolar", "CommercialSolar", "UtilitySolar", "Nuclear",
                                      "CoalPlant", "GasCombinedCycle", "GasPeaker", "Biomass", "HydroDam", "PumpedStorage",
                                      "BatteryStorage", "TidalGenerator", "WaveEnergy"};           // models/generator.rs:63-83
const char* kOffsetNames[4] = {"Forest", "Wetland", "ActiveCapture", "CarbonCredit"};                  // canonical order, core.rs:100-114
const int kMultPercent[3] = {100, 120, 150};

double powi(double a, int b) {      // Rust f64::powi == compiler-rt __powidf2
  const bool recip = b < 0;
  double r = 1.0;
  while (true) {
    if (b & 1) r *= a;
    b /= 2;
    if (b == 0) break;
    a *= a;
  }
  return recip ? 1.0 / r : r;
}
std::string display(double v) {      // Rust `{}` of an f64
  char buf[400];
  auto r = std::to_chars(buf, buf + sizeof(buf), v, std::chars_format::fixed);
  return std::string(buf, r.ptr);
}
std::string fixed(double v, int digits) { char b[64]; std::snprintf(b, sizeof(b), "%.*f", digits, v); return b; }

void grid_to_lon_lat(double x, double y, double& lon, double& lat) {      // csv_export.rs:42-84
  const double xv = std::min(std::max(x, 0.0), 50000.0), yv = std::min(std::max(y, 0.0), 50000.0);
  const double lon_range = -5.9 - -10.6, lat_range = 55.4 - 51.4;
  lon = -10.6 + (lon_range * (xv / 50000.0));
  lat = 51.4 + (lat_range * (yv / 50000.0));
}

// csv_export.rs:604-647: the generator type read off an id string
std::string type_of_id(const std::string& id) {
  auto has = [&](const char* s) { return id.find(s) != std::string::npos; };
  if (has("Onshore")) return "OnshoreWind";
  if (has("Offshore")) return "OffshoreWind";
  if (has("DomesticSolar")) return "DomesticSolar";
  if (has("CommercialSolar")) return "CommercialSolar";
  if (has("UtilitySolar")) return "UtilitySolar";
  if (has("Nuclear")) return "Nuclear";
  if (has("Coal")) return "CoalPlant";
  if (has("GasCombinedCycle")) return "GasCombinedCycle";
  if (has("GasPeaker")) return "GasPeaker";
  if (has("Biomass")) return "Biomass";
  if (has("Hydro")) return "HydroDam";
  if (has("PumpedStorage")) return "PumpedStorage";
  if (has("Battery")) return "BatteryStorage";
  if (has("Tidal")) return "TidalGenerator";
  if (has("Wave")) return "WaveEnergy";
  const size_t a = id.find('_');
  if (a == std::string::npos) return "Unknown";
  const size_t b = id.find('_', a + 1);
  return id.substr(a + 1, b == std::string::npos ? std::string::npos : b - a - 1);
}
// csv_export.rs:650-658: third '_' field as u32, else the default
uint32_t commissioning_of_id(const std::string& id, uint32_t fallback) {
  std::vector<std::string> parts; size_t pos = 0;
  while (true) { const size_t n = id.find('_', pos); parts.push_back(id.substr(pos, n == std::string::npos ? std::string::npos : n - pos)); if (n == std::string::npos) break; pos = n + 1; }
  if (parts.size() < 3 || parts[2].empty()) return fallback;
  uint64_t v = 0;
  size_t i = parts[2][0] == '+' ? 1 : 0;      // u32::from_str accepts a leading '+'
  if (i == parts[2].size()) return fallback;
  for (; i < parts[2].size(); ++i) { const char c = parts[2][i]; if (c < '0' || c > '9') return fallback; v = v * 10 + uint64_t(c - '0'); if (v > 0xFFFFFFFFull) return fallback; }
  return uint32_t(v);
}
int type_index(const std::string& name) { for (int t = 0; t < EG_N_TYPES; ++t) if (name == kTypeNames[t]) return t; return 7; }   // :948-952: GasCombinedCycle when unknown

struct TypeDefaults { double power, co2_per_mw, capital_per_mw, reliability; };
// csv_export.rs:661-695 (power, CO2), :873-945 (reliability, size, capital)
TypeDefaults defaults_of(const std::string& t) {
  if (t == "OnshoreWind") return {50.0, 0.0, 1500000.0, 0.35};
  if (t == "OffshoreWind") return {200.0, 0.0, 3500000.0, 0.35};
  if (t == "DomesticSolar") return {0.01, 0.0, 1000000.0, 0.25};
  if (t == "CommercialSolar") return {0.5, 0.0, 800000.0, 0.25};
  if (t == "UtilitySolar") return {50.0, 0.0, 600000.0, 0.25};
  if (t == "Nuclear") return {1000.0, 0.0, 6000000.0, 0.95};
  if (t == "CoalPlant") return {500.0, 3.0, 2000000.0, 0.90};
  if (t == "GasCombinedCycle") return {400.0, 0.4, 1000000.0, 0.85};
  if (t == "GasPeaker") return {100.0, 0.5, 500000.0, 0.90};
  if (t == "Biomass") return {50.0, 0.1, 3000000.0, 0.80};
  if (t == "HydroDam") return {250.0, 0.0, 2500000.0, 0.75};
  if (t == "PumpedStorage") return {200.0, 0.0, 2000000.0, 0.95};
  if (t == "BatteryStorage") return {50.0, 0.0, 400000.0, 0.98};
  if (t == "TidalGenerator") return {30.0, 0.0, 5000000.0, 0.45};
  if (t == "WaveEnergy") return {20.0, 0.0, 4000000.0, 0.40};
  return {100.0, 0.3, 2000000.0, 0.75};
}
double size_estimate(const std::string& t, double power) {      // csv_export.rs:913-920
  if (t == "OnshoreWind") return power / 3.0;
  if (t == "OffshoreWind") return power / 8.0;
  if (t == "DomesticSolar") return power * 8.0;
  if (t == "CommercialSolar") return power * 6.0;
  if (t == "UtilitySolar") return power * 2.0;
  return power / 50.0;
}
// config/tech_type.rs:52-200: planning / construction duration of a generator type commissioned in `year`
void durations(int type, uint32_t year, double& planning, double& construction) {
  // tech: 0 onshore, 1 offshore, 2 solar, 3 gas, 4 coal, 5 nuclear, 6 hydro, 7 storage, 8 biomass, 9 tidal, 10 wave
  static const int tech_of[EG_N_TYPES] = {0, 1, 2, 2, 2, 5, 4, 3, 3, 8, 6, 7, 7, 9, 10};
  static const double plan[11][2] = {{1.5, 0.5}, {3.0, 1.0}, {1.0, 0.3}, {2.0, 1.0}, {2.0, 1.0}, {5.0, 3.0}, {2.5, 1.5}, {1.5, 0.8}, {2.0, 1.0}, {3.0, 1.5}, {3.0, 1.5}};
  static const double build[11][2] = {{1.25, 0.75}, {3.0, 2.0}, {0.5, 0.25}, {2.5, 2.0}, {3.0, 3.0}, {7.0, 4.0}, {4.0, 3.5}, {1.0, 0.5}, {2.0, 1.5}, {2.0, 1.5}, {2.0, 1.5}};
  const int k = tech_of[type];
  const uint32_t cy = std::min<uint32_t>(std::max<uint32_t>(year, 2025u), 2050u);
  const double t = (double(cy) - 2025.0) / (2050.0 - 2025.0);
  planning = std::max(plan[k][0] + t * (plan[k][1] - plan[k][0]), plan[k][1]);
  construction = std::max(build[k][0] + t * (build[k][1] - build[k][0]), build[k][1]);
}

// StdRng::seed_from_u64 + gen::<f64>() (rand 0.8.5: ChaCha12 behind BlockRng; the same restatement as eg_policy.cpp's noise stream)
struct Stream {
  uint32_t key[8]; uint64_t counter = 0; uint32_t buf[64]; int index = 64;
  static uint32_t rotl(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }
  explicit Stream(uint64_t state) {
    for (int i = 0; i < 8; ++i) {
      state = state * 6364136223846793005ull + 11634580027462260723ull;
      const uint32_t x = uint32_t(((state >> 18) ^ state) >> 27), rot = uint32_t(state >> 59);
      key[i] = (x >> rot) | (x << ((32 - rot) & 31));
    }
  }
  void block(uint64_t ctr, uint32_t* out) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u}, x[16];
    for (int i = 0; i < 8; ++i) s[4 + i] = key[i];
    s[12] = uint32_t(ctr); s[13] = uint32_t(ctr >> 32); s[14] = 0; s[15] = 0;
    std::memcpy(x, s, sizeof(x));
    auto qr = [&](int a, int b, int c, int d) {
      x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12);
      x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8); x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
    };
    for (int r = 0; r < 6; ++r) { qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15); qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14); }
    for (int i = 0; i < 16; ++i) out[i] = x[i] + s[i];
  }
  uint64_t next_u64() {
    if (index >= 63) { for (int l = 0; l < 4; ++l) block(counter + uint64_t(l), buf + 16 * l); counter += 4; index = 0; }   // (an even number of words is drawn: no straddle)
    const uint64_t v = (uint64_t(buf[index + 1]) << 32) | buf[index]; index += 2; return v;
  }
  double next_f64() { return double(next_u64() >> 11) * (1.0 / 9007199254740992.0); }
};

void mkdirs(const std::string& p) {
  std::string cur;
  for (size_t i = 0; i <= p.size(); ++i) {
    if (i == p.size() || p[i] == '/') { if (!cur.empty()) ::mkdir(cur.c_str(), 0755); }
    if (i < p.size()) cur += p[i];
  }
}
std::string sanitize(const std::string& id) {      // csv_export.rs:565-569: alphanumeric, whitespace, '_' (ASCII ids here)
  std::string o;
  for (char c : id) if ((c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_' || c == ' ' || c == '\t') o += c;
  return o;
}

}  // namespace

extern "C" int32_t eg_export_run_details(const eg_world* world, const char* const* settlement_names, const eg_episode_out* run,
                                         const char* out_dir, uint64_t offset_seed) {
  if (!world || !run || !out_dir || !run->n_gens || !run->gen_pack || !run->n_act || !run->act_log) {
    eg::set_error("eg_export_run_details: world, out_dir and the run's n_gens, gen_pack, n_act, act_log are required"); return EG_ERR_BAD_ARG;
  }
  const int S = world->n_settlements, G0 = world->n_existing, NG = run->n_gens[0];
  if (NG < 0 || NG > EG_MAX_GENS) { eg::set_error("eg_export_run_details: bad n_gens"); return EG_ERR_BAD_ARG; }
  eg::HostTables H;
  eg::build_tables(*world, H);      // for the years in which the existing plant comes online (Q1)
  const std::string dir(out_dir);
  mkdirs(dir + "/yearly_details"); mkdirs(dir + "/operation_logs");

  {  // ---- settlements.csv (csv_export.rs:457-533)
    std::ofstream f(dir + "/yearly_details/settlements.csv", std::ios::binary | std::ios::trunc);
    if (!f) { eg::set_error("eg_export_run_details: cannot write under " + dir); return EG_ERR_BAD_ARG; }
    f << "Year,Name,Longitude,Latitude,Population,PowerUsage\n";
    for (int k = 0; k <= 25; ++k)
      for (int s = 0; s < S; ++s) {
        double lon, lat; grid_to_lon_lat(world->settlement_x[s], world->settlement_y[s], lon, lat);
        const double pop0 = double(world->settlement_pop[s]);
        const double usage0 = pop0 * (0.001 * std::pow(1.0 + 0.02, 0.0));      // settlements_loader.rs:29, const_funcs.rs:17-26 at 2025
        const uint32_t pop = uint32_t(std::round(pop0 * powi(1.01, k)));
        const std::string name = settlement_names && settlement_names[s] ? settlement_names[s] : "Settlement_" + std::to_string(s);
        f << (2025 + k) << ',' << name << ',' << fixed(lon, 6) << ',' << fixed(lat, 6) << ',' << pop << ',' << display(usage0 * powi(1.02, k)) << '\n';
      }
    if (!f.good()) { eg::set_error("eg_export_run_details: write failed"); return EG_ERR_BAD_ARG; }
  }

  {  // ---- generators.csv: second-pass rows only (see the header of this file)
    std::ofstream f(dir + "/yearly_details/generators.csv", std::ios::binary | std::ios::trunc);
    f << "Year,Generator ID,Type,Longitude,Latitude,Power Output (MW),Efficiency (%),Operation (%),CO2 Output (tonnes),Is Active,Commissioning Year,"
         "End of Life Year,Size,Capital Cost (\xE2\x82\xAC),Operating Cost (\xE2\x82\xAC),Total Annual Cost (\xE2\x82\xAC),Reliability Factor,Planning Time (years),"
         "Construction Time (years),Construction Speed\n";
    int final_map_generators = G0;      // the base map's plant + the sampled AddGenerator actions re-applied to it (multi_simulation.rs:886-890)
    { size_t p = 0; for (int yi = 0; yi < EG_YEARS; ++yi) for (int i = 0; i < run->n_act[yi] && p < size_t(EG_ACT_CAP); ++i, ++p) if (run->act_log[p] < 45) ++final_map_generators; }
    const bool no_generators = final_map_generators == 0;
    if (no_generators) f << "NOTE,No generators found in the simulation\n";      // csv_export.rs:556-560: and nothing else
    // ids of the simulated episode's map: "Existing_<type>_<n>" (generators_loader.rs:188), "Gen_<type>_<year>_<count>" with
    // count = generators in the map when it was added (actions.rs:60)
    struct Gen { std::string id; double x, y; int first_active; bool existing; };
    std::vector<Gen> gens;
    for (int g = 0; g < G0; ++g)
      gens.push_back({std::string("Existing_") + kTypeNames[world->existing_type[g]] + "_" + std::to_string(g), world->existing_x[g], world->existing_y[g],
                      H.existing_online[g], true});
    for (int k = 0; k < NG; ++k) {
      const int pack = run->gen_pack[k], t = pack & 15, yi = (pack >> 4) & 31;
      gens.push_back({std::string("Gen_") + kTypeNames[t] + "_" + std::to_string(2025 + yi) + "_" + std::to_string(G0 + k), 0.0, 0.0, yi, false});
    }
    for (int k = 0; k <= 25 && !no_generators; ++k)
      for (const Gen& g : gens) {
        if (k < g.first_active) continue;      // not in this year's YearlyMetrics.generator_efficiencies (metrics_calculation.rs:90-103)
        const std::string type = type_of_id(g.id);
        const uint32_t commissioning = commissioning_of_id(g.id, 2025u);
        const uint32_t eol = commissioning + 25u;
        const double efficiency = 0.99, operation = double(uint8_t(1.0 * 100.0)) * 100.0;      // generator.rs:703-705, csv_export.rs:846-849
        const TypeDefaults d = defaults_of(type);
        const double co2 = d.co2_per_mw == 0.0 ? 0.0 : d.power * d.co2_per_mw * 8760.0 / 1000.0;
        double x = g.x, y = g.y;
        if (!g.existing) {      // csv_export.rs:856-870: coordinates hashed from the id
          uint32_t h = 0; for (unsigned char c : g.id) h += c;
          x = 5000.0 + double(h % 100u) / 100.0 * (50000.0 - 10000.0);
          y = 5000.0 + double((h / 100u) % 100u) / 100.0 * (50000.0 - 10000.0);
        }
        double lon, lat; grid_to_lon_lat(x, y, lon, lat);
        const double size = size_estimate(type, d.power), capital = d.power * d.capital_per_mw, operating = capital * 0.03;
        double planning, construction; durations(type_index(type), commissioning, planning, construction);
        f << (2025 + k) << ',' << sanitize(g.id) << ',' << kTypeNames[type_index(type)] << ',' << fixed(lon, 6) << ',' << fixed(lat, 6) << ','
          << fixed(d.power, 2) << ',' << fixed(efficiency * 100.0, 2) << ',' << fixed(operation, 2) << ',' << fixed(co2, 2) << ",true," << commissioning << ','
          << eol << ',' << fixed(size * 100.0, 2) << ',' << fixed(capital, 2) << ',' << fixed(operating, 2) << ',' << fixed(capital + operating, 2) << ','
          << fixed(d.reliability, 2) << ',' << fixed(planning, 2) << ',' << fixed(construction, 2) << ",Normal\n";
      }
    if (!f.good()) { eg::set_error("eg_export_run_details: write failed"); return EG_ERR_BAD_ARG; }
  }

  {  // ---- carbon_offsets.csv (csv_export.rs:987-1092): the sampled AddCarbonOffset actions, still "Planned" on the final map
    std::ofstream f(dir + "/yearly_details/carbon_offsets.csv", std::ios::binary | std::ios::trunc);
    f << "Year,Offset ID,Type,X,Y,Size,Capture Efficiency (%),Power Consumption (MW),CO2 Offset (tonnes),Negative CO2 Emissions (tonnes),Cost (\xE2\x82\xAC),"
         "Operating Cost (\xE2\x82\xAC),Total Annual Cost (\xE2\x82\xAC),Cost Per Tonne (\xE2\x82\xAC)\n";
    struct Off { std::string id; int type, year_index, mult; double x, y; };
    std::vector<Off> offs;
    Stream rng(offset_seed);
    size_t pos = 0;
    for (int yi = 0; yi < EG_YEARS; ++yi)
      for (int i = 0; i < run->n_act[yi]; ++i, ++pos) {
        if (pos >= size_t(EG_ACT_CAP)) { eg::set_error("eg_export_run_details: action list longer than EG_ACT_CAP"); return EG_ERR_BAD_ARG; }
        const int a = run->act_log[pos];
        if (a < 45 || a >= 57) continue;
        const int ot = (a - 45) / 3, m = (a - 45) % 3;
        const double x = rng.next_f64() * 50000.0, y = rng.next_f64() * 50000.0;      // actions.rs:142-145 (thread_rng there)
        offs.push_back({std::string("Offset_") + kOffsetNames[ot] + "_" + std::to_string(2025 + yi) + "_" + std::to_string(offs.size()), ot, yi, m, x, y});
      }
    static const double size[4] = {500.0, 300.0, 100.0, 1000.0}, base_cost[4] = {1000000.0, 1000000.0, 1000000000.0, 50000000.0};
    static const double operating[4] = {10000.0, 15000.0, 100000.0, 5000.0}, op_factor[4] = {1.0, 1.01, 0.97, 1.02};      // carbon_offset.rs:196-202
    for (int k = 0; k <= 25; ++k)
      for (const Off& o : offs) {
        if (k < o.year_index) continue;      // the year in the id (csv_export.rs:1021-1036)
        double lon, lat; grid_to_lon_lat(o.x, o.y, lon, lat);
        const double mult = std::min(std::max(double(kMultPercent[o.mult]) / 100.0, 1.0), 5.0);
        const double cost = (base_cost[o.type] * powi(1.0 + 0.0185, k)) * mult;                                  // carbon_offset.rs:186-193
        const double op_cost = operating[o.type] * powi(1.0 + 0.0185, k) * std::pow(op_factor[o.type], double(k));   // :195-207
        const double co2 = 0.0;                                      // construction_status Planned (:241): the final map's clock never moves
        const double power = o.type == 2 ? size[o.type] * 0.5 : 0.0;      // ACTIVE_CAPTURE_POWER_PER_UNIT, carbon_offset.rs:87-90
        f << (2025 + k) << ',' << sanitize(o.id) << ',' << kOffsetNames[o.type] << ',' << fixed(lon, 6) << ',' << fixed(lat, 6) << ',' << display(size[o.type]) << ','
          << fixed(0.85 * 100.0, 2) << ',' << display(power) << ',' << fixed(co2, 2) << ',' << fixed(-co2, 2) << ',' << fixed(cost, 2) << ',' << fixed(op_cost, 2)
          << ',' << fixed(cost + op_cost, 2) << ',' << fixed(0.0, 2) << '\n';
      }
    if (!f.good()) { eg::set_error("eg_export_run_details: write failed"); return EG_ERR_BAD_ARG; }
  }

  {  // ---- generator_operation_logs.csv: the header (see the header of this file)
    std::ofstream f(dir + "/operation_logs/generator_operation_logs.csv", std::ios::binary | std::ios::trunc);
    f << "Year,Month,Day,Hour,Generator ID,Type,Power Output (MW),Operation %,Actual Output (MW),Weather Factor,CO2 Emissions (tonnes)\n";
    if (!f.good()) { eg::set_error("eg_export_run_details: write failed"); return EG_ERR_BAD_ARG; }
  }
  return EG_OK;
}
