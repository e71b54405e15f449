// eg_tables.cpp — host-side construction of the policy-independent tables.
//
// The reference recomputes, for every episode and several times a year, quantities that do not depend on the
// policy at all: settlement demand, the settlement factor of the placement score, the mean settlement opinion of
// a site, what the 59 existing plants contribute to every aggregate, and every powf/powi/exp in the cost and
// opinion formulas.  They are evaluated here once per world, in the reference's own operation order, so that a
// kernel which only looks values up and folds them in list order produces the same bits as a literal evaluation.
// Compiled with -ffp-contract=off -fno-builtin (csrc/Makefile).  Citations: /root/reference/aiSimulator/src/.
#include <cmath>
#include <cstring>

#include "eg_internal.h"

namespace eg {
namespace {

// ---- per-type data, enum order of models/generator.rs:11-36 ----
struct TypeRow {
  double base_cost;      // generator.rs:245-293
  double rate;           // generator.rs:184-202 cost evolution per year
  double mw;             // generator.rs:300-318
  double co2;            // constants.rs:125-128 (tonnes per year at size 1.0)
  double cap_factor;     // generator.rs:538-549; 1.0 = not intermittent
  double urban_factor;   // const_funcs.rs:39-46 with is_urban = can_be_urban(); 0 = flag not set
  int water;             // generator.rs:142-154 requires_water
  int cls;               // 0 dispatchable, 1 intermittent (generator.rs:86-94), 2 storage (:96-101)
  double radius;         // metal_location_search.rs:139-146
  double op_base, op_slope;  // const_funcs.rs:80-90
  int tech;              // tech_type.rs:53-69
};
enum { kOn, kOff, kSolar, kGas, kCoal, kNuc, kHydro, kBio, kTidal, kWave, kStore };
const TypeRow kType[kTypes] = {
    /* OnshoreWind      */ {1.5e6, 0.99, 500.0, 0.0, 0.35, 0.0, 0, 1, 5000.0, 0.83, 0.005, kOn},
    /* OffshoreWind     */ {4.0e6, 0.99, 800.0, 0.0, 0.35, 0.0, 1, 1, 5000.0, 0.83, 0.005, kOff},
    /* DomesticSolar    */ {1.0e7, 0.97, 10.0, 0.0, 0.20, 1.1, 0, 1, 3000.0, 0.89, 0.008, kSolar},
    /* CommercialSolar  */ {4.0e7, 0.97, 50.0, 0.0, 0.20, 1.1, 0, 1, 3000.0, 0.89, 0.008, kSolar},
    /* UtilitySolar     */ {2.4e8, 0.97, 300.0, 0.0, 0.20, 0.0, 0, 1, 3000.0, 0.89, 0.008, kSolar},
    /* Nuclear          */ {1.5e10, 0.99, 1500.0, 0.0, 1.0, 0.0, 0, 0, 12000.0, 0.43, 0.002, kNuc},
    /* CoalPlant        */ {1.5e9, 1.10, 1000.0, 6300.0, 1.0, 0.0, 0, 0, 8000.0, 0.41, -0.015, kCoal},
    /* GasCombinedCycle */ {5.6e8, 1.04, 800.0, 3500.0, 1.0, 0.0, 0, 0, 8000.0, 0.42, -0.008, kGas},
    /* GasPeaker        */ {5.0e8, 1.04, 400.0, 4800.0, 1.0, 0.7, 0, 0, 3000.0, 0.42, -0.008, kGas},
    /* Biomass          */ {1.5e8, 0.99, 50.0, 1500.0, 1.0, 0.0, 0, 0, 3000.0, 0.60, 0.001, kBio},
    /* HydroDam         */ {2.5e9, 1.06, 1200.0, 0.0, 1.0, 0.0, 0, 0, 7000.0, 0.89, 0.004, kHydro},
    /* PumpedStorage    */ {1.2e9, 1.06, 600.0, 0.0, 1.0, 0.0, 0, 2, 7000.0, 0.89, 0.004, kStore},
    /* BatteryStorage   */ {1.5e8, 0.97, 500.0, 0.0, 1.0, 0.0, 0, 2, 3000.0, 0.85, 0.003, kStore},
    /* TidalGenerator   */ {1.0e9, 0.95, 200.0, 0.0, 1.0, 0.0, 1, 0, 6000.0, 0.75, 0.005, kTidal},
    /* WaveEnergy       */ {8.0e8, 0.95, 100.0, 0.0, 1.0, 0.0, 1, 0, 6000.0, 0.75, 0.005, kWave},
};
const double kClassRadius[kRadiusClasses] = {12000.0, 8000.0, 5000.0, 7000.0, 6000.0, 3000.0};
// planning / construction durations in 2025 and 2050 per tech (tech_type.rs:70-200)
const double kPlan[11][2] = {{1.5, 0.5}, {3.0, 1.0}, {1.0, 0.3}, {2.0, 1.0}, {2.0, 1.0}, {5.0, 3.0},
                             {2.5, 1.5}, {2.0, 1.0}, {3.0, 1.5}, {3.0, 1.5}, {1.5, 0.8}};
const double kBuild[11][2] = {{1.25, 0.75}, {3.0, 2.0}, {0.5, 0.25}, {2.5, 2.0}, {3.0, 3.0}, {7.0, 4.0},
                              {4.0, 3.5}, {2.0, 1.5}, {2.0, 1.5}, {2.0, 1.5}, {1.0, 0.5}};
const double kMult[kMults] = {100.0, 120.0, 150.0};  // constants.rs:333-335, as percent

// offsets in canonical order Forest, Wetland, ActiveCapture, CarbonCredit (core.rs:100-114)
struct OffsetRow { double size, rate, base_cost; int natural; };
const OffsetRow kOffset[kOffsetTypes] = {{500.0, 25.0, 1.0e6, 1}, {300.0, 40.0, 1.0e6, 1},
                                         {100.0, 500.0, 1.0e9, 0}, {1000.0, 100.0, 5.0e7, 0}};

double powi(double a, int b) {  // Rust f64::powi == compiler-rt __powidf2
  bool recip = b < 0;
  double r = 1.0;
  while (true) {
    if (b & 1) r *= a;
    b /= 2;
    if (b == 0) break;
    a *= a;
  }
  return recip ? 1.0 / r : r;
}
inline double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }
inline double clamp_coord(double v) { return clampd(v, 0.0, 50000.0); }  // data/poi.rs:11-15
inline double dist(double ax, double ay, double bx, double by) {
  double dx = ax - bx, dy = ay - by;
  return __builtin_sqrt(dx * dx + dy * dy);
}
double duration(const double row[2], int year) {
  int cy = year < 2025 ? 2025 : (year > 2050 ? 2050 : year);
  double t = (double(cy) - 2025.0) / (2050.0 - 2025.0);
  double years = row[0] + t * (row[1] - row[0]);
  return years > row[1] ? years : row[1];
}
double time_reduction(double mult, double k) {  // const_funcs.rs:339-351
  double b = clampd(mult, 1.0, 5.0);
  if (b <= 1.0) return 1.0;
  double r = std::log(b) * k;
  if (r > 0.8) r = 0.8;
  return 1.0 - r;
}
double location_modifier(int t, bool urban, bool coastal) {  // const_funcs.rs:36-54
  double m = 1.0;
  if (urban) m *= (kType[t].urban_factor != 0.0 ? kType[t].urban_factor : 1.0);
  if (kType[t].water && coastal) m *= 1.15;
  return m;
}
double inflation(int y) { return powi(1.0 + 0.0185, y); }
double price_at(double base_cost_field, int t, int y, double loc, double mult) {  // generator.rs:582-594
  double tech = std::pow(kType[t].rate, double(y));
  double c = base_cost_field * inflation(y) * tech * loc;
  return c * mult;
}
double cost_opinion(double cost, int y) {  // const_funcs.rs:95-106
  double ref = 1384000000.0 * inflation(y);
  double n = cost / ref;
  if (n <= 1.0) return 1.0 - n;
  return 0.5 * std::exp(-0.5 * (n - 1.0));
}
double type_opinion(int t, int y) { return clampd(kType[t].op_base + kType[t].op_slope * double(y), 0.0, 1.0); }
bool inside_polygon(double px, double py, const double* x, const double* y, int n) {  // const_funcs.rs:143-158
  bool in = false;
  for (int i = 0, j = n - 1; i < n; j = i++)
    if (((y[i] > py) != (y[j] > py)) && (px < (x[j] - x[i]) * (py - y[i]) / (y[j] - y[i]) + x[i])) in = !in;
  return in;
}
double loader_max_power(int t) {  // generators_loader.rs:118-131
  switch (t) {
    case 0: return 500.0; case 1: return 800.0; case 6: return 1000.0; case 7: return 800.0;
    case 8: return 400.0; case 10: return 1200.0; case 9: return 50.0; default: return 800.0;
  }
}
}  // namespace
double class_radius(int radius_class) { return kClassRadius[radius_class]; }
namespace {
int rclass_of(double radius) {
  for (int k = 0; k < kRadiusClasses; ++k) if (kClassRadius[k] == radius) return k;
  return kRadiusClasses - 1;
}

}  // namespace

void build_tables(const eg_world& w, HostTables& T) {
  const int S = w.n_settlements, G0 = w.n_existing, P = w.n_coast;
  std::vector<double> sx(S), sy(S), gx(G0), gy(G0), px(P), py(P);
  for (int s = 0; s < S; ++s) { sx[s] = clamp_coord(w.settlement_x[s]); sy[s] = clamp_coord(w.settlement_y[s]); }
  for (int g = 0; g < G0; ++g) { gx[g] = clamp_coord(w.existing_x[g]); gy[g] = clamp_coord(w.existing_y[g]); }
  for (int p = 0; p < P; ++p) { px[p] = clamp_coord(w.coast_x[p]); py[p] = clamp_coord(w.coast_y[p]); }

  // ---- per-type scalars ----
  T.out_mw.resize(kTypes); T.co2_t.resize(kTypes); T.cls.resize(kTypes); T.rclass.resize(kTypes); T.marine.resize(kTypes);
  for (int t = 0; t < kTypes; ++t) {
    double base_output = kType[t].mw * 0.99 * 1.0;  // generator.rs:528, new plant: efficiency 0.99, operation 1.0
    T.out_mw[t] = kType[t].cls == 1 ? base_output * kType[t].cap_factor : base_output;
    double co2_out = kType[t].co2 * (double(100) / 100.0);  // actions.rs:50-56
    T.co2_t[t] = co2_out * 1.0 * (1.0 - (0.99 - 0.99));     // generator.rs:625
    T.cls[t] = kType[t].cls; T.rclass[t] = rclass_of(kType[t].radius); T.marine[t] = kType[t].water;
  }
  T.reach.resize(kRadiusClasses);
  T.dr.assign(size_t(kRadiusClasses) * 13 * 13, 1.0);
  for (int k = 0; k < kRadiusClasses; ++k) {
    int reach = 0;
    for (int di = 0; di <= kMaxReach; ++di)
      for (int dj = 0; dj <= kMaxReach; ++dj) {
        double d = dist(double(di) * 1000.0, double(dj) * 1000.0, 0.0, 0.0);
        if (d < kClassRadius[k]) { T.dr[(size_t(k) * 13 + di) * 13 + dj] = d / kClassRadius[k]; if (di > reach) reach = di; }
      }
    T.reach[k] = reach;
  }
  T.size_factor = 1.0 - (double(float(double(100) / 100.0)) * 0.1);  // actions.rs:47 → metal_location_search.rs:165

  // ---- yearly scalars and demand ----
  T.inflation.resize(kYears); T.carbon_price.resize(kYears); T.usage.resize(kYears); T.population.resize(kYears);
  std::vector<uint32_t> pop(w.settlement_pop, w.settlement_pop + S);
  std::vector<std::vector<uint32_t>> pop_by_year(kYears);
  for (int y = 0; y < kYears; ++y) {
    int year = 2025 + y;
    T.inflation[y] = inflation(y);
    if (year < 2030) T.carbon_price[y] = 75.0;                       // const_funcs.rs:186-203
    else if (year < 2040) T.carbon_price[y] = 75.0 + (double(year - 2030) / double(2040 - 2030)) * (130.0 - 75.0);
    else T.carbon_price[y] = 130.0 + (double(year - 2040) / double(2050 - 2040)) * (300.0 - 130.0);
    double per_capita = 0.001 * std::pow(1.0 + 0.02, double(y));     // const_funcs.rs:17-26
    double settlement_usage = 0.0; uint32_t total_pop = 0;
    for (int s = 0; s < S; ++s) {
      if (y > 0) pop[s] = uint32_t(std::round(double(pop[s]) * 1.01));   // simulation.rs:112
      total_pop += pop[s];
      settlement_usage += double(pop[s]) * per_capita;                  // 2025: settlements_loader.rs:31
    }
    pop_by_year[y] = pop;
    T.population[y] = double(total_pop);
    T.usage[y] = settlement_usage * (1.0 + (double(year) - 2024.0) * 0.02);  // map_handler.rs:826
  }

  // ---- geometry: candidate grid vs settlements / coast ----
  std::vector<double> dcs(size_t(kCells) * S);   // distance cell -> settlement
  T.m03.resize(kCells); T.coastf.resize(kCells);
  for (int c = 0; c < kCells; ++c) {
    double cx = double(c / kGrid) * 1000.0, cy = double(c % kGrid) * 1000.0;
    double opinions = 0.0;
    for (int s = 0; s < S; ++s) {
      double d = dist(cx, cy, sx[s], sy[s]);
      dcs[size_t(c) * S + s] = d;
      opinions += 1.0 / (1.0 + d / 10000.0);   // settlement.rs:103-106
    }
    double avg = S > 0 ? opinions / double(S) : 1.0;
    T.m03[c] = 0.03 * avg;
    double min_d = 1.7976931348623157e308;
    for (int p = 0; p < P; ++p) { double d = dist(cx, cy, px[p], py[p]); if (d < min_d) min_d = d; }
    T.coastf[c] = 1.0 / (1.0 + min_d / 5000.0);
  }

  // ---- existing plant (generators_loader.rs:133-206, then Map::add_generator at current_year 2024) ----
  std::vector<double> e_cost_field(G0), e_co2(G0), e_out(G0), e_m03(G0);
  T.existing_online.assign(G0, kYears);
  for (int g = 0; g < G0; ++g) {
    int t = w.existing_type[g];
    double size = clampd(w.existing_capacity_mw[g] / loader_max_power(t), 0.1, 1.0);
    bool coastal = P > 0 && inside_polygon(gx[g], gy[g], px.data(), py.data(), P) && gx[g] < 50000.0 * 0.1;
    double base0 = kType[t].base_cost * std::pow(kType[t].rate, 0.0);
    e_cost_field[g] = base0 * inflation(0) * std::pow(kType[t].rate, 0.0) * location_modifier(t, false, coastal);
    double co2_out = kType[t].co2 * size;
    e_co2[g] = co2_out * 1.0 * (1.0 - (0.99 - 0.99));
    double base_output = w.existing_capacity_mw[g] * 0.99 * 1.0;
    e_out[g] = kType[t].cls == 1 ? base_output * kType[t].cap_factor : base_output;
    double opinions = 0.0;
    for (int s = 0; s < S; ++s) opinions += 1.0 / (1.0 + dist(sx[s], sy[s], gx[g], gy[g]) / 10000.0);
    e_m03[g] = 0.03 * (S > 0 ? opinions / double(S) : 1.0);
    // construction state machine (generator.rs:451-517) from commissioning year 2024, opinion 0.65, multiplier 1.0
    if (w.existing_operational_at_start) { T.existing_online[g] = 0; continue; }
    double plan = duration(kPlan[kType[t].tech], 2024) * (1.0 - 0.65 * 0.5) * time_reduction(1.0, 0.25);
    if (plan < 0.25) plan = 0.25;
    double build = duration(kBuild[kType[t].tech], 2024) * time_reduction(1.0, 0.5);
    if (build < 0.1) build = 0.1;
    int status = 0, start_year = 0;
    for (int y = 0; y < kYears; ++y) {
      int year = 2025 + y;
      if (status == 0) { if (double(year - 2024) >= plan) status = 1; }
      else if (status == 1) { status = 2; start_year = year; }
      else if (status == 2) { if (double(year - start_year) >= build) status = 3; }
      if (status == 3) { T.existing_online[g] = y; break; }
    }
  }
  T.pre_co2.assign(kYears, 0.0); T.pre_tg.assign(kYears, 0.0); T.pre_ig.assign(kYears, 0.0); T.pre_sg.assign(kYears, 0.0);
  T.pre_optot.assign(kYears, 0.0); T.pre_opcnt.assign(kYears, 0);
  for (int y = 0; y < kYears; ++y) {
    double co2 = 0.0, tg = 0.0, ig = 0.0, sg = 0.0, op = 0.0; int cnt = 0;
    for (int g = 0; g < G0; ++g) {
      if (y < T.existing_online[g]) continue;
      int t = w.existing_type[g];
      co2 += e_co2[g];
      if (kType[t].cls == 1) ig += e_out[g]; else if (kType[t].cls == 2) sg += e_out[g]; else tg += e_out[g];
      double cost = price_at(e_cost_field[g], t, y, location_modifier(t, kType[t].urban_factor != 0.0, kType[t].water != 0), 1.0);
      op += e_m03[g] + 0.12 * type_opinion(t, y) + 0.82 * cost_opinion(cost, y);  // map_handler.rs:946-948
      cnt += 1;
    }
    T.pre_co2[y] = co2; T.pre_tg[y] = tg; T.pre_ig[y] = ig; T.pre_sg[y] = sg; T.pre_optot[y] = op; T.pre_opcnt[y] = cnt;
  }

  // ---- placement prefix: settlements, then existing plant, in list order (metal_location_search.rs:126-153) ----
  T.te.resize(size_t(kYears) * kRadiusClasses * kCells);
  std::vector<double> dce(size_t(kCells) * (G0 > 0 ? G0 : 1));
  for (int c = 0; c < kCells; ++c)
    for (int g = 0; g < G0; ++g)
      dce[size_t(c) * G0 + g] = dist(double(c / kGrid) * 1000.0, double(c % kGrid) * 1000.0, gx[g], gy[g]);
  for (int y = 0; y < kYears; ++y) {
    const std::vector<uint32_t>& py_ = pop_by_year[y];
    std::vector<double> popf(S);
    for (int s = 0; s < S; ++s) popf[s] = 1.0 + double(py_[s]) / 1000000.0;
    for (int c = 0; c < kCells; ++c) {
      double score = 1.0;
      const double* d = &dcs[size_t(c) * S];
      for (int s = 0; s < S; ++s) score *= popf[s] / (1.0 + d[s] / 10000.0);
      for (int k = 0; k < kRadiusClasses; ++k) {
        double sc = score; const double R = kClassRadius[k];
        for (int g = 0; g < G0; ++g) { double dd = dce[size_t(c) * G0 + g]; if (dd < R) sc *= dd / R; }
        T.te[(size_t(y) * kRadiusClasses + k) * kCells + c] = sc;
      }
    }
  }

  // ---- opinion and cost of new plant ----
  T.t12.resize(size_t(kYears) * kTypes);
  T.cc.assign(size_t(kYears) * kTypes * kYears * kMults * 2, 0.0);
  for (int y = 0; y < kYears; ++y)
    for (int t = 0; t < kTypes; ++t) {
      T.t12[size_t(y) * kTypes + t] = 0.12 * type_opinion(t, y);
      double loc = location_modifier(t, kType[t].urban_factor != 0.0, kType[t].water != 0);
      for (int b = 0; b < kYears; ++b) {
        double field = kType[t].base_cost * std::pow(kType[t].rate, double(b));   // generator.rs:295-297
        for (int m = 0; m < kMults; ++m) {
          double mult = clampd(clampd(kMult[m] / 100.0, 1.0, 5.0), 1.0, 5.0);      // actions.rs:43-44, generator.rs:727-730
          double cost = price_at(field, t, y, loc, mult);
          size_t i = (((size_t(y) * kTypes + t) * kYears + b) * kMults + m) * 2;
          T.cc[i] = cost; T.cc[i + 1] = 0.82 * cost_opinion(cost, y);
        }
      }
    }

  // ---- offsets ----
  T.offv.assign(size_t(kYears) * kOffsetTypes * kYears, 0.0);
  T.offc.resize(size_t(kYears) * kOffsetTypes * kMults);
  for (int y = 0; y < kYears; ++y)
    for (int o = 0; o < kOffsetTypes; ++o) {
      double base_offset = kOffset[o].size * kOffset[o].rate;
      double eff = clampd(0.85, 0.0, 1.0);
      for (int b = 0; b <= y; ++b) {
        double maturity = kOffset[o].natural ? clampd(1.0 - std::exp(-0.1 * double(y - b)), 0.0, 1.0) : 1.0;
        T.offv[(size_t(y) * kOffsetTypes + o) * kYears + b] = base_offset * eff * maturity;
      }
      for (int m = 0; m < kMults; ++m) {
        double mult = clampd(clampd(kMult[m] / 100.0, 1.0, 5.0), 1.0, 5.0);
        T.offc[(size_t(y) * kOffsetTypes + o) * kMults + m] = (kOffset[o].base_cost * powi(1.0 + 0.0185, y)) * mult;
      }
    }
}


// "Estimated Cost" column of simulation_summary.csv (utils/csv_export.rs:249-266 AddGenerator, :343-366 AddCarbonOffset;
// every other action prints 0.00: the table's generator ids are empty, so the exporter's look-ups find nothing).
// AddGenerator: calc_generator_cost(get_base_cost(year), year, can_be_urban, requires_water, requires_water) * percent / 100.
double action_cost_estimate(int action, int year_index) {
  if (action < 0 || year_index < 0) return 0.0;
  if (action < kTypes * kMults) {
    const int t = action / kMults, m = action - t * kMults;
    const double base = kType[t].base_cost * std::pow(kType[t].rate, double(year_index));      // generator.rs:295-297
    return price_at(base, t, year_index, location_modifier(t, true, true), kMult[m] / 100.0);
  }
  if (action < kTypes * kMults + kOffsetTypes * kMults) {
    const int k = action - kTypes * kMults, o = k / kMults, m = k - o * kMults;
    return (kOffset[o].base_cost * inflation(year_index)) * (kMult[m] / 100.0);
  }
  return 0.0;
}
}  // namespace eg
