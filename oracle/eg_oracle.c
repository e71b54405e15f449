/*
 * eg_oracle.c — CPU ORACLE (test infrastructure, NOT the product).  See eg_oracle.h.
 *
 * Literal restatement: every aggregate is recomputed from the generator /
 * offset / settlement lists exactly where the reference recomputes it, the
 * placement search walks all 100x100 candidates with sqrt+div per factor, and
 * transcendental calls (powf, powi, exp, ln) are made where the reference makes
 * them.  No tables, no memoisation.  Build with -ffp-contract=off -fno-builtin
 * (see oracle/Makefile) so no FMA is formed and no libm call is constant-folded.
 *
 * Citations are relative to /root/reference/aiSimulator/src/.
 */
#include "eg_oracle.h"
#include "../include/eg_detpow.h"   /* shared deterministic x^p, see the header */

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* The stalled sampler's w.powf(p) (sampling.rs:199-213).  Default: the shared IEEE-only eg_detpow, which the kernels evaluate too —
 * so that oracle and kernel agree to the bit by construction.  og_set_libm_pow(1) makes the oracle do what the reference does: call
 * libm's pow.  That mode is what breaks the circle of comparing eg_detpow with itself: the product (eg_detpow on the device) is then
 * held against an oracle that never sees that function (tests/test_detpow.py, tests/test_gpu_parity.py). */
static int g_libm_pow = 0;
void og_set_libm_pow(int32_t on) { g_libm_pow = on ? 1 : 0; }
int32_t og_get_libm_pow(void) { return g_libm_pow; }
static double stalled_pow(double x, double p) { return g_libm_pow ? pow(x, p) : eg_detpow(x, p); }

/* ------------------------------------------------------------------------- */
/* constants (config/constants.rs, ai/learning/constants.rs)                  */
/* ------------------------------------------------------------------------- */
#define END_YEAR 2050
#define MAP_MAX 50000.0               /* constants.rs:6-7 */
#define INFLATION_RATE 0.0185         /* constants.rs:16 */
#define BASE_EFFICIENCY 0.99          /* constants.rs:68 */
#define REFERENCE_ANNUAL_EXPENDITURE 1384000000.0 /* constants.rs:71 */
#define MAX_ACCEPTABLE_EMISSIONS 1000000.0        /* constants.rs:114 */
#define MAX_ACCEPTABLE_COST 50000000000.0         /* constants.rs:115 */

enum { T_ONSHORE, T_OFFSHORE, T_DOMSOLAR, T_COMSOLAR, T_UTILSOLAR, T_NUCLEAR, T_COAL, T_CCGT,
       T_PEAKER, T_BIOMASS, T_HYDRO, T_PUMPED, T_BATTERY, T_TIDAL, T_WAVE };
/* canonical offset order = insertion order of core.rs:100-114 */
enum { O_FOREST, O_WETLAND, O_ACTIVE, O_CREDIT };
enum { ST_PLANNED, ST_GRANTED, ST_BUILDING, ST_OPERATIONAL };

static const int MULT_PERCENT[3] = {100, 120, 150}; /* constants.rs:333-335 */

/* compiler-rt __powidf2: what Rust's f64::powi lowers to */
static double powi_(double a, int b) {
  const int recip = b < 0;
  double r = 1.0;
  for (;;) {
    if (b & 1) r *= a;
    b /= 2;
    if (b == 0) break;
    a *= a;
  }
  return recip ? 1.0 / r : r;
}
static double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }
static double maxd(double a, double b) { return a > b ? a : b; } /* f64::max, no NaNs on this path */
static double mind(double a, double b) { return a < b ? a : b; }
/* f64::round = half away from zero (simulation.rs:112, Q12) */
static double round_half_away(double v) { return round(v); }

/* ------------------------------------------------------------------------- */
/* per-type formulas (models/generator.rs)                                    */
/* ------------------------------------------------------------------------- */
static double cost_evolution_rate(int t) { /* generator.rs:184-202 */
  switch (t) {
    case T_ONSHORE: case T_OFFSHORE: return 0.99;
    case T_DOMSOLAR: case T_COMSOLAR: case T_UTILSOLAR: return 0.97;
    case T_NUCLEAR: return 0.99;
    case T_COAL: return 1.10;
    case T_CCGT: case T_PEAKER: return 1.04;
    case T_BIOMASS: return 0.99;
    case T_HYDRO: case T_PUMPED: return 1.06;
    case T_BATTERY: return 0.97;
    default: return 0.95; /* tidal, wave */
  }
}
static double base_cost_2025(int t) { /* generator.rs:245-293 */
  static const double c[OG_NTYPES] = {1500000.0, 4000000.0, 10000000.0, 40000000.0, 240000000.0,
                                      15000000000.0, 1500000000.0, 560000000.0, 500000000.0, 150000000.0,
                                      2500000000.0, 1200000000.0, 150000000.0, 1000000000.0, 800000000.0};
  return c[t];
}
static double type_base_cost(int t, int year) { /* generator.rs:244-298 get_base_cost */
  double years_from_base = (double)(year - OG_BASE_YEAR);
  return base_cost_2025(t) * pow(cost_evolution_rate(t), years_from_base);
}
static double type_base_power(int t) { /* generator.rs:300-318, constants.rs:164-182 */
  static const double p[OG_NTYPES] = {500.0, 800.0, 10.0, 50.0, 300.0, 1500.0, 1000.0, 800.0,
                                      400.0, 50.0, 1200.0, 600.0, 500.0, 200.0, 100.0};
  return p[t];
}
static int is_intermittent(int t) { return t <= T_UTILSOLAR; }            /* generator.rs:86-94 */
static int is_storage(int t) { return t == T_PUMPED || t == T_BATTERY; }  /* generator.rs:96-101 */
static int can_be_urban(int t) { return t == T_DOMSOLAR || t == T_COMSOLAR || t == T_PEAKER; } /* :132-140 */
static int requires_water(int t) { return t == T_OFFSHORE || t == T_TIDAL || t == T_WAVE; }   /* :142-154 */
static double co2_rate(int t) { /* constants.rs:125-128, actions.rs:50-56 */
  switch (t) { case T_COAL: return 6300.0; case T_CCGT: return 3500.0; case T_PEAKER: return 4800.0;
               case T_BIOMASS: return 1500.0; default: return 0.0; }
}

static double inflation_factor(int year) { /* const_funcs.rs:13-15 */
  return powi_(1.0 + INFLATION_RATE, year - OG_BASE_YEAR);
}
/* const_funcs.rs:28-57 */
static double calc_generator_cost(int t, double base_cost, int year, int is_urban, int is_coastal, int is_river) {
  double inflation = inflation_factor(year);
  double years_from_base = (double)(year - OG_BASE_YEAR);
  double technology_factor = pow(cost_evolution_rate(t), years_from_base);
  double location_modifier = 1.0;
  if (is_urban) {
    double f = 1.0;
    if (t == T_DOMSOLAR || t == T_COMSOLAR) f = 1.1; /* URBAN_SOLAR_BONUS */
    else if (t == T_PEAKER) f = 0.7;                 /* URBAN_PEAKER_PENALTY */
    location_modifier *= f;
  }
  if (requires_water(t)) {
    if (is_coastal) location_modifier *= 1.15;       /* COASTAL_BONUS */
    else if (is_river) location_modifier *= 1.10;
  }
  return base_cost * inflation * technology_factor * location_modifier;
}
static double calc_type_opinion(int t, int year) { /* const_funcs.rs:78-93 */
  double years_passed = (double)(year - OG_BASE_YEAR);
  double base, change;
  switch (t) {
    case T_ONSHORE: case T_OFFSHORE: base = 0.83; change = 0.005; break;
    case T_DOMSOLAR: case T_COMSOLAR: case T_UTILSOLAR: base = 0.89; change = 0.008; break;
    case T_NUCLEAR: base = 0.43; change = 0.002; break;
    case T_COAL: base = 0.41; change = -0.015; break;
    case T_CCGT: case T_PEAKER: base = 0.42; change = -0.008; break;
    case T_HYDRO: case T_PUMPED: base = 0.89; change = 0.004; break;
    case T_TIDAL: case T_WAVE: base = 0.75; change = 0.005; break;
    case T_BATTERY: base = 0.85; change = 0.003; break;
    default: base = 0.60; change = 0.001; break; /* biomass */
  }
  return clampd(base + change * years_passed, 0.0, 1.0);
}
static double calc_cost_opinion(double cost, int year) { /* const_funcs.rs:95-106 */
  double inflation_adjusted_max = REFERENCE_ANNUAL_EXPENDITURE * inflation_factor(year);
  double normalized_cost = cost / inflation_adjusted_max;
  if (normalized_cost <= 1.0) return 1.0 - normalized_cost;
  return 0.5 * exp(-0.5 * (normalized_cost - 1.0));
}

/* config/tech_type.rs:53-200 */
enum { TT_ONSHORE, TT_OFFSHORE, TT_SOLAR, TT_GAS, TT_COAL, TT_NUCLEAR, TT_HYDRO, TT_BIOMASS, TT_TIDAL,
       TT_WAVE, TT_STORAGE };
static int map_to_tech_type(int t) {
  switch (t) {
    case T_ONSHORE: return TT_ONSHORE; case T_OFFSHORE: return TT_OFFSHORE;
    case T_DOMSOLAR: case T_COMSOLAR: case T_UTILSOLAR: return TT_SOLAR;
    case T_CCGT: case T_PEAKER: return TT_GAS; case T_COAL: return TT_COAL;
    case T_NUCLEAR: return TT_NUCLEAR; case T_HYDRO: return TT_HYDRO;
    case T_PUMPED: case T_BATTERY: return TT_STORAGE; case T_BIOMASS: return TT_BIOMASS;
    case T_TIDAL: return TT_TIDAL; default: return TT_WAVE;
  }
}
static double interp_duration(int year, double base_2025, double end_2050) {
  int cy = year < OG_BASE_YEAR ? OG_BASE_YEAR : (year > 2050 ? 2050 : year);
  double t = ((double)cy - (double)OG_BASE_YEAR) / (2050.0 - (double)OG_BASE_YEAR);
  double years = base_2025 + t * (end_2050 - base_2025);
  return maxd(years, end_2050);
}
static double planning_duration(int year, int tech) { /* tech_type.rs:70-133 */
  double b, m;
  switch (tech) {
    case TT_ONSHORE: b = 1.5; m = 0.5; break;   case TT_OFFSHORE: b = 3.0; m = 1.0; break;
    case TT_SOLAR: b = 1.0; m = 0.3; break;     case TT_GAS: case TT_COAL: b = 2.0; m = 1.0; break;
    case TT_NUCLEAR: b = 5.0; m = 3.0; break;   case TT_HYDRO: b = 2.5; m = 1.5; break;
    case TT_STORAGE: b = 1.5; m = 0.8; break;   case TT_BIOMASS: b = 2.0; m = 1.0; break;
    default: b = 3.0; m = 1.5; break; /* tidal, wave */
  }
  return interp_duration(year, b, m);
}
static double construction_duration(int year, int tech) { /* tech_type.rs:136-200 */
  double b, m;
  switch (tech) {
    case TT_ONSHORE: b = 1.25; m = 0.75; break; case TT_OFFSHORE: b = 3.0; m = 2.0; break;
    case TT_SOLAR: b = 0.5; m = 0.25; break;    case TT_GAS: b = 2.5; m = 2.0; break;
    case TT_COAL: b = 3.0; m = 3.0; break;      case TT_NUCLEAR: b = 7.0; m = 4.0; break;
    case TT_HYDRO: b = 4.0; m = 3.5; break;     case TT_STORAGE: b = 1.0; m = 0.5; break;
    default: b = 2.0; m = 1.5; break; /* biomass, tidal, wave */
  }
  return interp_duration(year, b, m);
}
static double time_reduction_factor(double cost_multiplier, double reduction_factor) { /* const_funcs.rs:339-351 */
  double bounded = clampd(cost_multiplier, 1.0, 5.0);
  if (bounded <= 1.0) return 1.0;
  double log_reduction = mind(log(bounded) * reduction_factor, 0.8);
  return 1.0 - log_reduction;
}
static double planning_permission_time(int t, int year, double opinion, double mult) { /* const_funcs.rs:269-283 */
  double base_time = planning_duration(year, map_to_tech_type(t));
  double opinion_factor = 1.0 - (opinion * 0.5);
  double cost_factor = time_reduction_factor(mult, 0.25);
  return maxd(base_time * opinion_factor * cost_factor, 0.25);
}
static double construction_time(int t, int year, double mult) { /* const_funcs.rs:285-295 */
  double base_time = construction_duration(year, map_to_tech_type(t));
  double cost_factor = time_reduction_factor(mult, 0.5);
  return maxd(base_time * cost_factor, 0.1);
}
static double offset_planning_time(int ot, int year, double opinion, double mult) { /* const_funcs.rs:297-317 */
  static const double bt[4] = {1.0, 1.5, 2.0, 0.5}; /* forest, wetland, active, credit; constants.rs:307-310 */
  double years_from_base = (double)(year - OG_BASE_YEAR);
  double year_factor = pow(1.0 - 0.02, years_from_base);
  double opinion_factor = 1.0 - (opinion * 0.5);
  double cost_factor = time_reduction_factor(mult, 0.25);
  return maxd(bt[ot] * year_factor * opinion_factor * cost_factor, 0.25);
}
static double offset_construction_time(int ot, int year, double mult) { /* const_funcs.rs:319-336 */
  static const double bt[4] = {1.0, 2.0, 3.0, 0.2}; /* constants.rs:312-315 */
  double years_from_base = (double)(year - OG_BASE_YEAR);
  double year_factor = pow(1.0 - 0.03, years_from_base);
  double cost_factor = time_reduction_factor(mult, 0.5);
  return maxd(bt[ot] * year_factor * cost_factor, 0.1);
}

double og_carbon_price(int32_t year) { /* const_funcs.rs:186-203 */
  if (year < 2030) return 75.0;
  if (year < 2040) {
    double phase_length = (double)(2040 - 2030);
    double t = (double)(year - 2030) / phase_length;
    return 75.0 + t * (130.0 - 75.0);
  }
  if (year <= 2050) {
    double phase_length = (double)(2050 - 2040);
    double t = (double)(year - 2040) / phase_length;
    return 130.0 + t * (300.0 - 130.0);
  }
  return 300.0;
}

/* ------------------------------------------------------------------------- */
/* entities                                                                    */
/* ------------------------------------------------------------------------- */
typedef struct { double x, y; } coord;
static coord coord_new(double x, double y) { /* data/poi.rs:11-15 */
  coord c; c.x = clampd(x, 0.0, MAP_MAX); c.y = clampd(y, 0.0, MAP_MAX); return c;
}
static double distance_to(coord a, coord b) { /* poi.rs:17-21 */
  double dx = a.x - b.x, dy = a.y - b.y;
  return __builtin_sqrt(dx * dx + dy * dy);
}

typedef struct {
  coord c; int type; int existing;
  double base_cost, power_out, size, co2_out, efficiency, operation_percentage;
  int commissioning_year, is_active_flag, status;
  double planning_permission_time, construction_time;
  int construction_start_year, construction_complete_year, build_year; /* build_year: id-encoded year, generator.rs:689-701 */
  double mult;
} gen_t;

typedef struct {
  int type; double base_cost, size, capture_efficiency, mult;
  int status, commissioning_year, construction_start_year, construction_complete_year;
  double planning_permission_time, construction_time;
} off_t_;

typedef struct { coord c; uint32_t pop; double usage; } settle_t;

struct og_world {
  int S, G0, P;
  settle_t *settle;  /* 2025 state */
  gen_t *existing;   /* state after initialize_map (commissioned 2024) */
  coord *coast;
};

typedef struct {
  const og_world *w;
  settle_t *settle; int S;
  gen_t *gens; int ngens, gcap;
  off_t_ *offs; int noffs, ocap;
  int current_year; int enable_delays;
} map_t;

/* generator.rs:451-480 */
static void gen_initialize_construction(gen_t *g, int year, double opinion, int enable_delays) {
  g->commissioning_year = year;
  if (!enable_delays) {
    g->status = ST_OPERATIONAL; g->construction_start_year = year; g->construction_complete_year = year;
    return;
  }
  g->planning_permission_time = planning_permission_time(g->type, year, opinion, g->mult);
  g->construction_time = construction_time(g->type, year, g->mult);
  g->status = ST_PLANNED;
}
/* generator.rs:482-517 */
static void gen_update_construction_status(gen_t *g, int current_year) {
  if (g->status == ST_OPERATIONAL) return;
  double years_since = (double)(current_year - g->commissioning_year);
  if (g->status == ST_PLANNED) {
    if (years_since >= g->planning_permission_time) g->status = ST_GRANTED;
  } else if (g->status == ST_GRANTED) {
    g->status = ST_BUILDING; g->construction_start_year = current_year;
  } else if (g->status == ST_BUILDING) {
    double since_start = (double)(current_year - g->construction_start_year);
    if (since_start >= g->construction_time) {
      g->status = ST_OPERATIONAL; g->construction_complete_year = current_year; g->is_active_flag = 1;
    }
  }
}
static int gen_is_active(const gen_t *g) { return g->is_active_flag && g->status == ST_OPERATIONAL; } /* :519-521 */
static double gen_power_output(const gen_t *g) { /* generator.rs:523-554, hour = None */
  if (!gen_is_active(g)) return 0.0;
  double base_output = g->power_out * g->efficiency * g->operation_percentage;
  if (is_intermittent(g->type)) {
    if (g->type == T_ONSHORE || g->type == T_OFFSHORE) return base_output * 0.35; /* WIND_CAPACITY_FACTOR */
    return base_output * 0.20;                                                   /* SOLAR_CAPACITY_FACTOR */
  }
  return base_output;
}
static double gen_current_cost(const gen_t *g, int year) { /* generator.rs:582-594 */
  double base = calc_generator_cost(g->type, g->base_cost, year, can_be_urban(g->type), requires_water(g->type),
                                    requires_water(g->type));
  return base * g->mult;
}
static double gen_co2_output(const gen_t *g) { /* generator.rs:618-626 */
  if (!gen_is_active(g)) return 0.0;
  return g->co2_out * g->operation_percentage * (1.0 - (g->efficiency - BASE_EFFICIENCY));
}

/* carbon_offset.rs:115-145 */
static void off_initialize_construction(off_t_ *o, int year, double opinion, int enable_delays) {
  o->commissioning_year = year;
  if (!enable_delays) {
    o->status = ST_OPERATIONAL; o->construction_start_year = year; o->construction_complete_year = year;
    return;
  }
  o->planning_permission_time = offset_planning_time(o->type, year, opinion, o->mult);
  o->construction_time = offset_construction_time(o->type, year, o->mult);
  o->status = ST_PLANNED;
}
/* carbon_offset.rs:147-181 */
static void off_update_construction_status(off_t_ *o, int current_year) {
  if (o->status == ST_OPERATIONAL) return;
  double years_since = (double)(current_year - o->commissioning_year);
  if (o->status == ST_PLANNED) {
    if (years_since >= o->planning_permission_time) o->status = ST_GRANTED;
  } else if (o->status == ST_GRANTED) {
    o->status = ST_BUILDING; o->construction_start_year = current_year;
  } else if (o->status == ST_BUILDING) {
    double since_start = (double)(current_year - o->construction_start_year);
    if (since_start >= o->construction_time) { o->status = ST_OPERATIONAL; o->construction_complete_year = current_year; }
  }
}
static double off_current_cost(const off_t_ *o, int year) { /* carbon_offset.rs:188-195 */
  double infl = powi_(1.0 + INFLATION_RATE, year - OG_BASE_YEAR);
  double base = o->base_cost * infl;
  return base * o->mult;
}
static double off_calc(const off_t_ *o, int year) { /* carbon_offset.rs:210-260 */
  double base_offset;
  switch (o->type) {
    case O_FOREST: base_offset = o->size * 25.0; break;
    case O_ACTIVE: base_offset = o->size * 500.0; break;
    case O_CREDIT: base_offset = o->size * 100.0; break;
    default: base_offset = o->size * 40.0; break; /* wetland */
  }
  if (o->status == ST_OPERATIONAL) {
    double maturity = 1.0;
    if (o->type == O_FOREST || o->type == O_WETLAND) {
      double years_from_start = (double)(year - o->construction_complete_year);
      maturity = clampd(1.0 - exp(-0.1 * years_from_start), 0.0, 1.0);
    }
    return base_offset * o->capture_efficiency * maturity;
  }
  if (o->status == ST_BUILDING) {
    double years_in = (double)(year - o->construction_start_year);
    double progress = clampd(years_in / o->construction_time, 0.0, 1.0);
    double eff;
    if (o->type == O_FOREST || o->type == O_WETLAND) eff = pow(progress, 0.7) * 0.5;
    else if (o->type == O_CREDIT) eff = progress * 0.8;
    else eff = pow(progress, 2.0) * 0.3;
    return base_offset * o->capture_efficiency * eff;
  }
  return 0.0;
}

/* ------------------------------------------------------------------------- */
/* world                                                                       */
/* ------------------------------------------------------------------------- */
static int point_inside_polygon(coord p, const coord *poly, int n) { /* const_funcs.rs:143-158 */
  int inside = 0, j = n - 1;
  for (int i = 0; i < n; ++i) {
    if (((poly[i].y > p.y) != (poly[j].y > p.y)) &&
        (p.x < (poly[j].x - poly[i].x) * (p.y - poly[i].y) / (poly[j].y - poly[i].y) + poly[i].x))
      inside = !inside;
    j = i;
  }
  return inside;
}
static double max_power_for_capacity(int t) { /* data/generators_loader.rs:118-131 */
  switch (t) {
    case T_ONSHORE: return 500.0; case T_OFFSHORE: return 800.0; case T_COAL: return 1000.0;
    case T_CCGT: return 800.0; case T_PEAKER: return 400.0; case T_HYDRO: return 1200.0;
    case T_BIOMASS: return 50.0; default: return 800.0;
  }
}

og_world *og_world_create(int32_t S, const double *sx, const double *sy, const uint32_t *spop,
                          int32_t G0, const double *gx, const double *gy, const int32_t *gtype, const double *gcap,
                          int32_t P, const double *cx, const double *cy, int32_t existing_operational_at_start) {
  og_world *w = (og_world *)calloc(1, sizeof(*w));
  w->S = S; w->G0 = G0; w->P = P;
  w->settle = (settle_t *)calloc((size_t)(S > 0 ? S : 1), sizeof(settle_t));
  w->existing = (gen_t *)calloc((size_t)(G0 > 0 ? G0 : 1), sizeof(gen_t));
  w->coast = (coord *)calloc((size_t)(P > 0 ? P : 1), sizeof(coord));
  for (int i = 0; i < P; ++i) w->coast[i] = coord_new(cx[i], cy[i]);
  for (int i = 0; i < S; ++i) { /* data/settlements_loader.rs:30-33 */
    w->settle[i].c = coord_new(sx[i], sy[i]);
    w->settle[i].pop = spop[i];
    double per_capita = 0.001 * pow(1.0 + 0.02, (double)(2025 - OG_BASE_YEAR)); /* const_funcs.rs:17-26 */
    w->settle[i].usage = (double)spop[i] * per_capita;
  }
  for (int i = 0; i < G0; ++i) { /* generators_loader.rs:133-206 with year = 2025 */
    gen_t *g = &w->existing[i];
    int t = gtype[i];
    g->c = coord_new(gx[i], gy[i]); g->type = t; g->existing = 1;
    g->size = clampd(gcap[i] / max_power_for_capacity(t), 0.1, 1.0);
    int on_land = P > 0 ? point_inside_polygon(g->c, w->coast, P) : 0;
    int is_coastal = on_land && (g->c.x < MAP_MAX * 0.1);
    g->base_cost = calc_generator_cost(t, type_base_cost(t, 2025), 2025, 0, is_coastal, 0);
    g->power_out = gcap[i];
    g->co2_out = co2_rate(t) * g->size;
    g->efficiency = BASE_EFFICIENCY; g->operation_percentage = 1.0; g->is_active_flag = 1;
    g->mult = 1.0; g->build_year = 2020; g->status = ST_PLANNED;
    /* Map::add_generator with current_year = 2024 and delays = true (map_handler.rs:391-394,553-577; Q1).
     * The completion estimate 2024 + p + c is far below 2050 for every type, so nothing is cancelled. */
    if (existing_operational_at_start) gen_initialize_construction(g, 2024, 0.65, 0);
    else gen_initialize_construction(g, 2024, 0.65, 1);
  }
  return w;
}
void og_world_destroy(og_world *w) {
  if (!w) return;
  free(w->settle); free(w->existing); free(w->coast); free(w);
}

static void map_init(map_t *m, const og_world *w) {
  m->w = w; m->S = w->S;
  m->settle = (settle_t *)malloc(sizeof(settle_t) * (size_t)(w->S > 0 ? w->S : 1));
  memcpy(m->settle, w->settle, sizeof(settle_t) * (size_t)w->S);
  m->gcap = w->G0 + 64; m->gens = (gen_t *)malloc(sizeof(gen_t) * (size_t)m->gcap);
  memcpy(m->gens, w->existing, sizeof(gen_t) * (size_t)w->G0);
  m->ngens = w->G0;
  m->ocap = 16; m->offs = (off_t_ *)malloc(sizeof(off_t_) * (size_t)m->ocap); m->noffs = 0;
  m->current_year = 2024; m->enable_delays = 1;
}
static void map_free(map_t *m) { free(m->settle); free(m->gens); free(m->offs); }

/* map_handler.rs:819-827 */
static double map_total_power_usage(const map_t *m, int year) {
  double settlement_usage = 0.0;
  for (int i = 0; i < m->S; ++i) settlement_usage += m->settle[i].usage;
  return settlement_usage * (1.0 + ((double)year - 2024.0) * 0.02);
}
static uint32_t map_total_population(const map_t *m) { /* :813-817 */
  uint32_t s = 0; for (int i = 0; i < m->S; ++i) s += m->settle[i].pop; return s;
}
/* map_handler.rs:829-868 — the intermittent cap is computed and discarded there; the result is a plain sum */
static double map_total_power_generation(const map_t *m) {
  double total_generation = 0.0, intermittent_generation = 0.0, storage_generation = 0.0;
  for (int i = 0; i < m->ngens; ++i) {
    double output = gen_power_output(&m->gens[i]);
    if (is_intermittent(m->gens[i].type)) intermittent_generation += output;
    else if (is_storage(m->gens[i].type)) storage_generation += output;
    else total_generation += output;
  }
  return total_generation + intermittent_generation + storage_generation;
}
static double map_total_co2(const map_t *m) { /* :902-910 */
  double s = 0.0;
  for (int i = 0; i < m->ngens; ++i) if (gen_is_active(&m->gens[i])) s += gen_co2_output(&m->gens[i]);
  return s;
}
static double map_total_offset(const map_t *m, int year) { /* :912-919 */
  double s = 0.0; for (int i = 0; i < m->noffs; ++i) s += off_calc(&m->offs[i], year); return s;
}
static double map_net_co2(const map_t *m, int year) { return map_total_co2(m) - map_total_offset(m, year); }
/* map_handler.rs:925-949 + models/settlement.rs:103-106 */
static double map_new_generator_opinion(const map_t *m, const gen_t *g, int year) {
  double settlement_opinions = 0.0;
  for (int i = 0; i < m->S; ++i) {
    double distance = distance_to(m->settle[i].c, g->c);
    settlement_opinions += 1.0 / (1.0 + distance / 10000.0);
  }
  double avg = m->S > 0 ? settlement_opinions / (double)m->S : 1.0;
  double type_opinion = calc_type_opinion(g->type, year);
  double cost_opinion = calc_cost_opinion(gen_current_cost(g, year), year);
  return 0.03 * avg + 0.12 * type_opinion + 0.82 * cost_opinion; /* constants.rs:11-13 */
}
/* analysis/metrics_calculation.rs:7-30 */
static double average_opinion(const map_t *m, int year, int *active_count) {
  double total = 0.0; int count = 0;
  for (int i = 0; i < m->ngens; ++i)
    if (gen_is_active(&m->gens[i])) { total += map_new_generator_opinion(m, &m->gens[i], year); count += 1; }
  if (active_count) *active_count = count;
  return count > 0 ? total / (double)count : 1.0;
}
static double map_total_capital_cost(const map_t *m, int year) { /* :951-965 */
  double generator_costs = 0.0, offset_costs = 0.0;
  for (int i = 0; i < m->ngens; ++i) if (!m->gens[i].existing) generator_costs += gen_current_cost(&m->gens[i], year);
  for (int i = 0; i < m->noffs; ++i) offset_costs += off_current_cost(&m->offs[i], year);
  return generator_costs + offset_costs;
}
static double map_yearly_capital_cost(const map_t *m, int year) { /* :968-985; offsets' start year parses to 2025 (Q11) */
  double generator_costs = 0.0, offset_costs = 0.0;
  for (int i = 0; i < m->ngens; ++i)
    if (m->gens[i].build_year == year && !m->gens[i].existing) generator_costs += gen_current_cost(&m->gens[i], year);
  for (int i = 0; i < m->noffs; ++i) if (2025 == year) offset_costs += off_current_cost(&m->offs[i], year);
  return generator_costs + offset_costs;
}
static void map_update_construction_status(map_t *m) { /* :1492-1504 */
  for (int i = 0; i < m->ngens; ++i) gen_update_construction_status(&m->gens[i], m->current_year);
  for (int i = 0; i < m->noffs; ++i) off_update_construction_status(&m->offs[i], m->current_year);
}

typedef struct { double net_emissions, public_opinion, power_balance, total_cost; } action_result;
static action_result map_state(const map_t *m, int year) { /* simulation.rs:122-135 */
  action_result r;
  r.net_emissions = map_net_co2(m, year);
  r.public_opinion = average_opinion(m, year, 0);
  r.power_balance = map_total_power_generation(m) - map_total_power_usage(m, year);
  r.total_cost = map_total_capital_cost(m, year);
  return r;
}

/* gpu/metal_location_search.rs:110-176 (the CPU fallback that actually runs) */
static double penalty_radius(int t) {
  switch (t) {
    case T_NUCLEAR: return 12000.0;
    case T_COAL: case T_CCGT: return 8000.0;
    case T_ONSHORE: case T_OFFSHORE: return 5000.0;
    case T_HYDRO: case T_PUMPED: return 7000.0;
    case T_TIDAL: case T_WAVE: return 6000.0;
    default: return 3000.0;
  }
}
static int find_suitable_location(const settle_t *settle, int S, const gen_t *gens, int ngens, const coord *extra,
                                  int nextra, const coord *coast, int P, int t, float size_penalty, coord *out,
                                  double *out_score) {
  const double grid_step = 1000.0;
  const int num_x = (int)(100000.0 / grid_step), num_y = (int)(100000.0 / grid_step); /* file-local MAP_MAX :200-201 */
  double best_score = 0.0; int found = 0;
  for (int i = 0; i < num_x; ++i) {
    for (int j = 0; j < num_y; ++j) {
      coord loc = coord_new((double)i * grid_step, (double)j * grid_step);
      double score = 1.0;
      for (int s = 0; s < S; ++s) {
        double distance = distance_to(loc, settle[s].c);
        double population_factor = (double)settle[s].pop / 1000000.0;
        score *= (1.0 + population_factor) / (1.0 + distance / 10000.0);
      }
      double radius = penalty_radius(t);
      for (int g = 0; g < ngens; ++g) {
        double distance = distance_to(loc, gens[g].c);
        if (distance < radius) score *= distance / radius;
      }
      for (int g = 0; g < nextra; ++g) {
        double distance = distance_to(loc, extra[g]);
        if (distance < radius) score *= distance / radius;
      }
      if (requires_water(t)) {
        double min_d = 1.7976931348623157e308;
        for (int p = 0; p < P; ++p) { double d = distance_to(loc, coast[p]); if (d < min_d) min_d = d; }
        score *= 1.0 / (1.0 + min_d / 5000.0);
      }
      score *= 1.0 - ((double)size_penalty * 0.1);
      if (score > best_score) { best_score = score; *out = loc; found = 1; }
    }
  }
  if (out_score) *out_score = best_score;
  return found;
}
static int cell_of(coord c) { return (int)(c.x / 1000.0) * 51 + (int)(c.y / 1000.0); }

/* ------------------------------------------------------------------------- */
/* RNG: rand 0.8.5 StdRng = rand_chacha 0.3.1 ChaCha12Rng behind BlockRng     */
/* ------------------------------------------------------------------------- */
typedef struct { uint32_t key[8]; uint64_t counter, stream; uint32_t buf[64]; int index; uint64_t words; } rng_t;
#define ROTL32(v, n) (((v) << (n)) | ((v) >> (32 - (n))))
#define QR(a, b, c, d) \
  a += b; d ^= a; d = ROTL32(d, 16); c += d; b ^= c; b = ROTL32(b, 12); \
  a += b; d ^= a; d = ROTL32(d, 8);  c += d; b ^= c; b = ROTL32(b, 7);
void og_chacha_block(const uint32_t key[8], uint64_t counter, uint64_t stream, int32_t rounds, uint32_t out[16]) {
  uint32_t s[16], x[16];
  s[0] = 0x61707865u; s[1] = 0x3320646eu; s[2] = 0x79622d32u; s[3] = 0x6b206574u;
  for (int i = 0; i < 8; ++i) s[4 + i] = key[i];
  s[12] = (uint32_t)counter; s[13] = (uint32_t)(counter >> 32);
  s[14] = (uint32_t)stream;  s[15] = (uint32_t)(stream >> 32);
  memcpy(x, s, sizeof(x));
  for (int r = 0; r < rounds; r += 2) {
    QR(x[0], x[4], x[8], x[12]) QR(x[1], x[5], x[9], x[13]) QR(x[2], x[6], x[10], x[14]) QR(x[3], x[7], x[11], x[15])
    QR(x[0], x[5], x[10], x[15]) QR(x[1], x[6], x[11], x[12]) QR(x[2], x[7], x[8], x[13]) QR(x[3], x[4], x[9], x[14])
  }
  for (int i = 0; i < 16; ++i) out[i] = x[i] + s[i];
}
void og_rng_seed_words(uint64_t state, uint32_t key[8]) { /* rand_core 0.6.4 SeedableRng::seed_from_u64 (PCG32 fill) */
  const uint64_t MUL = 6364136223846793005ull, INC = 11634580027462260723ull;
  for (int i = 0; i < 8; ++i) {
    state = state * MUL + INC;
    uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
    uint32_t rot = (uint32_t)(state >> 59);
    key[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
  }
}
static void rng_seed(rng_t *r, uint64_t seed) {
  og_rng_seed_words(seed, r->key); r->counter = 0; r->stream = 0; r->index = 64; r->words = 0;
}
static void rng_refill(rng_t *r) { /* four consecutive blocks per refill */
  for (int b = 0; b < 4; ++b) og_chacha_block(r->key, r->counter + (uint64_t)b, r->stream, 12, r->buf + 16 * b);
  r->counter += 4;
}
static uint64_t rng_next_u64(rng_t *r) { /* rand_core block.rs BlockRng::next_u64 */
  r->words += 1;
  if (r->index < 63) { uint64_t v = ((uint64_t)r->buf[r->index + 1] << 32) | r->buf[r->index]; r->index += 2; return v; }
  if (r->index >= 64) { rng_refill(r); r->index = 2; return ((uint64_t)r->buf[1] << 32) | r->buf[0]; }
  uint64_t x = r->buf[63]; rng_refill(r); r->index = 1; return ((uint64_t)r->buf[0] << 32) | x;
}
static uint32_t rng_next_u32(rng_t *r) {
  r->words += 1;
  if (r->index >= 64) { rng_refill(r); r->index = 0; }
  return r->buf[r->index++];
}
static double rng_f64(rng_t *r) { return (double)(rng_next_u64(r) >> 11) * (1.0 / 9007199254740992.0); }
static uint64_t rng_range_u64(rng_t *r, uint64_t range) { /* rand uniform.rs sample_single_inclusive, usize */
  int lz = __builtin_clzll(range);
  uint64_t zone = (range << lz) - 1;
  for (;;) {
    uint64_t v = rng_next_u64(r);
    unsigned __int128 m = (unsigned __int128)v * range;
    if ((uint64_t)m <= zone) return (uint64_t)(m >> 64);
  }
}
static uint32_t rng_range_u32(rng_t *r, uint32_t range) { /* same, u32 (sampling.rs:438-441, :478-479) */
  int lz = __builtin_clz(range);
  uint32_t zone = (range << lz) - 1;
  for (;;) {
    uint32_t v = rng_next_u32(r);
    uint64_t m = (uint64_t)v * range;
    if ((uint32_t)m <= zone) return (uint32_t)(m >> 32);
  }
}
void og_rng_stream(uint64_t seed, int32_t n, uint64_t *out) {
  rng_t r; rng_seed(&r, seed); for (int i = 0; i < n; ++i) out[i] = rng_next_u64(&r);
}
uint64_t og_rng_gen_range_probe(uint64_t seed, uint64_t n, int32_t skip) {
  rng_t r; rng_seed(&r, seed); for (int i = 0; i < skip; ++i) rng_next_u64(&r); return rng_range_u64(&r, n);
}

/* ------------------------------------------------------------------------- */
/* ActionWeights (ai/learning/weights/ *.rs)                                   */
/* ------------------------------------------------------------------------- */
typedef struct { int n, cap; uint8_t *a; } list_t;
static void list_clear(list_t *l) { l->n = 0; }
static void list_push(list_t *l, uint8_t v) {
  if (l->n == l->cap) { l->cap = l->cap ? l->cap * 2 : 16; l->a = (uint8_t *)realloc(l->a, (size_t)l->cap); }
  l->a[l->n++] = v;
}
static void list_copy(list_t *d, const list_t *s) { list_clear(d); for (int i = 0; i < s->n; ++i) list_push(d, s->a[i]); }
static int list_contains(const list_t *l, uint8_t v) { for (int i = 0; i < l->n; ++i) if (l->a[i] == v) return 1; return 0; }

struct og_weights {
  double w[OG_YEARS][OG_NA], dw[OG_YEARS][OG_ND], cw[OG_YEARS][OG_NC];
  int has_cw;                 /* action_count_weights present (dropped by the checkpoint loader) */
  double learning_rate, exploration_rate;
  int has_best; double best_metrics[4];
  int has_best_weights; double best_w[OG_YEARS][OG_NA];
  int has_best_actions, has_best_deficit;
  list_t best_actions[OG_YEARS], best_deficit[OG_YEARS];
  uint32_t iteration_count, stall;
  list_t cur_run[OG_YEARS], cur_def[OG_YEARS];
  int force_best;
  int replay_idx[OG_YEARS], replay_def_idx[OG_YEARS];
  rng_t rng; int has_rng;
};

#define MIN_WEIGHT 0.0001
#define MAX_WEIGHT 0.999

static int is_add_generator(int a) { return a < 45; }
static int action_type(int a) { return a / 3; }
/* deficit table order = insertion order core.rs:130-152 */
static const int DEFICIT_TYPE[14] = {T_PEAKER, T_CCGT, T_BATTERY, T_PUMPED, T_BIOMASS, T_ONSHORE, T_OFFSHORE,
                                     T_UTILSOLAR, T_HYDRO, T_NUCLEAR, T_DOMSOLAR, T_COMSOLAR, T_TIDAL, T_WAVE};
static int deficit_index_of(int a) { /* main-table action -> deficit-table slot, -1 if absent */
  if (a == OG_A_NOTHING) return 14;
  if (!is_add_generator(a) || a % 3 != 0) return -1;
  for (int i = 0; i < 14; ++i) if (DEFICIT_TYPE[i] == a / 3) return i;
  return -1;
}

og_weights *og_weights_new(void) { /* core.rs:25-250 */
  og_weights *p = (og_weights *)calloc(1, sizeof(*p));
  static const double tw[OG_NTYPES] = {0.08, 0.08, 0.05, 0.05, 0.08, 0.03, 0.04, 0.06, 0.02, 0.04, 0.06, 0.06,
                                       0.07, 0.05, 0.05}; /* learning/constants.rs:46-60 */
  static const double dwi[OG_ND] = {0.15, 0.15, 0.15, 0.10, 0.10, 0.07, 0.07, 0.06, 0.06, 0.05, 0.01, 0.01, 0.01,
                                    0.01, 0.001}; /* constants.rs:66-77 */
  for (int y = 0; y < OG_YEARS; ++y) {
    for (int t = 0; t < OG_NTYPES; ++t) {
      p->w[y][3 * t] = tw[t]; p->w[y][3 * t + 1] = tw[t] * 0.5; p->w[y][3 * t + 2] = tw[t] * 0.25;
    }
    for (int o = 0; o < 4; ++o) {
      p->w[y][45 + 3 * o] = 0.02; p->w[y][46 + 3 * o] = 0.02 * 0.5; p->w[y][47 + 3 * o] = 0.02 * 0.25;
    }
    p->w[y][OG_A_UPGRADE] = 0.04; p->w[y][OG_A_ADJUST] = 0.04; p->w[y][OG_A_CLOSE] = 0.02; p->w[y][OG_A_NOTHING] = 0.1;
    for (int i = 0; i < OG_ND; ++i) p->dw[y][i] = dwi[i];
    double total_weight = 0.0; /* core.rs:158-186 */
    for (int count = 0; count <= 20; ++count) {
      double base_weight = exp(-0.8 * (double)count);
      double multiplier = count == 0 ? 4.0 : count == 1 ? 3.5 : count == 2 ? 3.0 : count == 3 ? 2.5
                        : count == 4 ? 2.0 : count == 5 ? 1.5 : 1.0;
      double weight = base_weight * multiplier;
      p->cw[y][count] = weight; total_weight += weight;
    }
    for (int count = 0; count <= 20; ++count) p->cw[y][count] /= total_weight;
  }
  p->has_cw = 1; p->learning_rate = 0.2; p->exploration_rate = 0.2;
  return p;
}
og_weights *og_weights_clone(const og_weights *s) {
  og_weights *d = (og_weights *)malloc(sizeof(*d));
  memcpy(d, s, sizeof(*d));
  for (int y = 0; y < OG_YEARS; ++y) {
    list_t *dl[4] = {&d->best_actions[y], &d->best_deficit[y], &d->cur_run[y], &d->cur_def[y]};
    const list_t *sl[4] = {&s->best_actions[y], &s->best_deficit[y], &s->cur_run[y], &s->cur_def[y]};
    for (int k = 0; k < 4; ++k) { dl[k]->a = 0; dl[k]->n = dl[k]->cap = 0; list_copy(dl[k], sl[k]); }
  }
  return d;
}
void og_weights_free(og_weights *p) {
  if (!p) return;
  for (int y = 0; y < OG_YEARS; ++y) { free(p->best_actions[y].a); free(p->best_deficit[y].a); free(p->cur_run[y].a); free(p->cur_def[y].a); }
  free(p);
}
void og_weights_get_tables(const og_weights *p, double *w, double *dw, double *cw) {
  if (w) memcpy(w, p->w, sizeof(p->w));
  if (dw) memcpy(dw, p->dw, sizeof(p->dw));
  if (cw) memcpy(cw, p->cw, sizeof(p->cw));
}
void og_weights_set_tables(og_weights *p, const double *w, const double *dw, const double *cw) {
  if (w) memcpy(p->w, w, sizeof(p->w));
  if (dw) memcpy(p->dw, dw, sizeof(p->dw));
  if (cw) memcpy(p->cw, cw, sizeof(p->cw));
}
void og_weights_set_has_count_weights(og_weights *p, int32_t has) { p->has_cw = has; }
void og_weights_get_best_weights(const og_weights *p, double *w) { memcpy(w, p->best_w, sizeof(p->best_w)); }
double og_weights_get_scalar(const og_weights *p, int32_t which) {
  switch (which) {
    case 0: return p->learning_rate; case 1: return p->exploration_rate; case 2: return (double)p->stall;
    case 3: return (double)p->iteration_count; case 4: return (double)p->has_best;
    case 5: case 6: case 7: case 8: return p->best_metrics[which - 5];
    case 9: return (double)p->has_best_actions; case 10: return (double)p->has_best_deficit;
    default: return 0.0;
  }
}
void og_weights_set_scalar(og_weights *p, int32_t which, double v) {
  switch (which) {
    case 0: p->learning_rate = v; break; case 1: p->exploration_rate = v; break; case 2: p->stall = (uint32_t)v; break;
    case 3: p->iteration_count = (uint32_t)v; break; case 4: p->has_best = (int)v; break;
    case 5: case 6: case 7: case 8: p->best_metrics[which - 5] = v; break;
    case 9: p->has_best_actions = (int)v; break; case 10: p->has_best_deficit = (int)v; break;
    default: break;
  }
}
static list_t *weights_list(og_weights *p, int which, int yi) {
  switch (which) { case 0: return &p->best_actions[yi]; case 1: return &p->best_deficit[yi];
                   case 2: return &p->cur_run[yi]; default: return &p->cur_def[yi]; }
}
int32_t og_weights_get_list(const og_weights *p, int32_t which, int32_t yi, uint8_t *out, int32_t cap) {
  const list_t *l = weights_list((og_weights *)p, which, yi);
  for (int i = 0; i < l->n && i < cap; ++i) out[i] = l->a[i];
  return l->n;
}
void og_weights_set_list(og_weights *p, int32_t which, int32_t yi, const uint8_t *in, int32_t n) {
  list_t *l = weights_list(p, which, yi); list_clear(l); for (int i = 0; i < n; ++i) list_push(l, in[i]);
}

double og_score_metrics(const double m[4], int32_t cost_only) { /* ai/metrics/scoring.rs:5-45 */
  if (cost_only) {
    double normalized_cost = maxd(m[2] / MAX_ACCEPTABLE_COST, 1.0);
    double log_cost = log(normalized_cost);
    double max_expected = log(MAX_ACCEPTABLE_COST * 100.0 / MAX_ACCEPTABLE_COST);
    return 2.0 - mind(log_cost / max_expected, 1.0);
  }
  if (m[0] > 0.0) return 1.0 - mind(m[0] / MAX_ACCEPTABLE_EMISSIONS, 1.0);
  double normalized_cost = maxd(m[2] / MAX_ACCEPTABLE_COST, 1.0);
  double log_cost = log(normalized_cost);
  double max_expected = log(MAX_ACCEPTABLE_COST * 100.0 / MAX_ACCEPTABLE_COST);
  double cost_score = 1.0 - mind(log_cost / max_expected, 1.0);
  double opinion_score = m[1];
  double cost_weight = normalized_cost > 8.0 ? 0.8 : 0.5;
  double opinion_weight = 1.0 - cost_weight;
  return 1.0 + (cost_score * cost_weight + opinion_score * opinion_weight);
}
static double evaluate_impact(const action_result *cur, const action_result *nxt) { /* scoring.rs:46-85, mode None */
  if (cur->net_emissions > 0.0)
    return (cur->net_emissions - nxt->net_emissions) / maxd(fabs(cur->net_emissions), 1.0);
  double cost_change = nxt->total_cost - cur->total_cost;
  double cost_improvement = -cost_change / maxd(fabs(cur->total_cost), 1.0);
  double opinion_improvement = (nxt->public_opinion - cur->public_opinion) / maxd(fabs(cur->public_opinion), 1.0);
  double cost_weight = cur->total_cost > MAX_ACCEPTABLE_COST * 8.0 ? 0.8 : 0.5;
  double opinion_weight = 1.0 - cost_weight;
  return cost_improvement * cost_weight + opinion_improvement * opinion_weight;
}
double og_evaluate_action_impact(const double c[4], const double n[4], int32_t cost_only) {
  if (cost_only) { double cc = n[3] - c[3]; return -cc / maxd(fabs(c[3]), 1.0); }
  action_result a = {c[0], c[1], c[2], c[3]}, b = {n[0], n[1], n[2], n[3]};
  return evaluate_impact(&a, &b);
}

/* core/multi_simulation.rs:613-620 with metrics_to_action_result of :55-62 (power_balance: 0.0) */
int32_t og_fold_best_result(int32_t n, const int32_t *status, const double *metrics, int32_t cost_only, int64_t first_index,
                            int32_t *has, double best[4], int64_t *best_index) {
  int32_t takeovers = 0;
  for (int32_t i = 0; i < n; ++i) {
    if (status && status[i] != 0) continue; /* (a failed iteration ends the reference's run, :611) */
    const double *m = metrics + 4 * (size_t)i;
    int take = 1;
    if (*has) {
      double result_ar[4] = {m[0], m[1], 0.0, m[2]}, best_ar[4] = {best[0], best[1], 0.0, best[2]};
      take = og_evaluate_action_impact(result_ar, best_ar, cost_only) > 0.0;
    }
    if (take) { for (int k = 0; k < 4; ++k) best[k] = m[k]; *has = 1; *best_index = first_index + i; takeovers += 1; }
  }
  return takeovers;
}

/* learning.rs:21-88.  `relative_improvement` compares the best score with itself (Q4) and is therefore 0 whenever the
 * best score is positive; otherwise it equals that (non-positive) score.  The optimisation mode held inside
 * ActionWeights is always None (Q3). */
static void update_weights(og_weights *p, int action, int yi, double improvement) {
  double *yw = p->w[yi];
  double current_weight = yw[action];
  double final_impact_score = p->has_best ? og_score_metrics(p->best_metrics, 0) : 0.0;
  double relative_improvement;
  if (p->has_best) {
    double best_score = og_score_metrics(p->best_metrics, 0);
    relative_improvement = best_score > 0.0 ? (final_impact_score - best_score) / best_score : final_impact_score;
  } else relative_improvement = final_impact_score;
  double immediate_weight = relative_improvement > 0.0 ? 0.7 : 0.3;
  double combined = immediate_weight * improvement + (1.0 - immediate_weight) * relative_improvement;
  double adjustment = combined > 0.0 ? 1.0 + (p->learning_rate * combined)
                                     : 1.0 / (1.0 + (p->learning_rate * fabs(combined)));
  yw[action] = mind(maxd(current_weight * adjustment, MIN_WEIGHT), MAX_WEIGHT);
  if (combined < 0.0) {
    double boost = 1.0 + (p->learning_rate * 0.1);
    for (int a = 0; a < OG_NA; ++a)
      if (a != action && is_add_generator(a)) yw[a] = mind(yw[a] * boost, MAX_WEIGHT);
    if (p->has_best && p->best_metrics[0] <= 0.0 && p->best_metrics[2] > MAX_ACCEPTABLE_COST * 8.0)
      yw[OG_A_NOTHING] = mind(yw[OG_A_NOTHING] * (1.0 + p->learning_rate * 0.2), MAX_WEIGHT);
  }
}
/* deficit.rs:82-135 */
static void update_deficit_weights(og_weights *p, int action, int yi, double improvement) {
  int slot = deficit_index_of(action);
  if (slot < 0) return; /* cannot happen: every deficit action is one of the 14 default-cost generator entries */
  double *yw = p->dw[yi];
  double current_weight = yw[slot];
  double adjustment = improvement > 0.0 ? 1.0 + (p->learning_rate * improvement * 1.5)
                                        : 1.0 / (1.0 + (p->learning_rate * fabs(improvement) * 1.5));
  yw[slot] = mind(maxd(current_weight * adjustment, MIN_WEIGHT), MAX_WEIGHT);
  if (improvement < 0.0) {
    double boost = 1.0 + (p->learning_rate * 0.1);
    for (int i = 0; i < 14; ++i) if (i != slot) yw[i] = mind(yw[i] * boost, MAX_WEIGHT);
  }
}

/* sampling.rs:445-490 — the reference draws from thread_rng here; the canonical definition uses the episode stream */
static int smart_fallback_action(og_weights *p, int year) {
  int acts[7]; uint32_t wts[7];
  acts[0] = 3 * T_ONSHORE; wts[0] = 15; acts[1] = 3 * T_OFFSHORE; wts[1] = 10; acts[2] = 3 * T_UTILSOLAR; wts[2] = 15;
  acts[3] = 3 * T_BATTERY; wts[3] = year < 2035 ? 10 : 20;
  uint32_t offw = year < 2035 ? 5 : (year < 2045 ? 15 : 25);
  acts[4] = 45 + 3 * O_FOREST; wts[4] = offw; acts[5] = 45 + 3 * O_ACTIVE; wts[5] = offw;
  acts[6] = 3 * T_CCGT; wts[6] = year < 2035 ? 15 : (year < 2045 ? 10 : 5);
  uint32_t total = 0; for (int i = 0; i < 7; ++i) total += wts[i];
  uint32_t choice = rng_range_u32(&p->rng, total);
  for (int i = 0; i < 7; ++i) { if (choice < wts[i]) return acts[i]; choice -= wts[i]; }
  return 3 * T_BATTERY;
}
static int smart_deficit_fallback_action(og_weights *p) { /* sampling.rs:492-528 */
  int acts[6] = {3 * T_PEAKER, 3 * T_BATTERY, 3 * T_CCGT, 3 * T_ONSHORE, 3 * T_OFFSHORE, 3 * T_UTILSOLAR};
  uint32_t wts[6] = {30, 30, 20, 10, (uint32_t)(0.07 * 0.5), (uint32_t)(0.06 * 0.5 * 100.0)};
  uint32_t total = 0; for (int i = 0; i < 6; ++i) total += wts[i];
  uint32_t choice = rng_range_u32(&p->rng, total);
  for (int i = 0; i < 6; ++i) { if (choice < wts[i]) return acts[i]; choice -= wts[i]; }
  return 3 * T_BATTERY;
}

static int sample_action(og_weights *p, int yi) { /* sampling.rs:76-238 */
  int year = OG_BASE_YEAR + yi;
  if (p->force_best) {
    int a;
    if (p->has_best_actions) {
      int idx = p->replay_idx[yi];
      if (idx < p->best_actions[yi].n) { a = p->best_actions[yi].a[idx]; p->replay_idx[yi] = idx + 1; }
      else a = smart_fallback_action(p, year);
    } else a = smart_fallback_action(p, year);
    list_push(&p->cur_run[yi], (uint8_t)a);
    return a;
  }
  const double *yw = p->w[yi];
  double current_exploration = p->stall > 100 ? p->exploration_rate * (1.0 / (1.0 + 0.01 * (double)p->stall))
                                               : p->exploration_rate;
  int should_explore = rng_f64(&p->rng) < current_exploration;
  if (should_explore) return (int)rng_range_u64(&p->rng, OG_NA);
  double total_weight = 0.0; for (int a = 0; a < OG_NA; ++a) total_weight += yw[a];
  if (total_weight <= 0.0) return 3 * T_PEAKER;
  if (p->stall > 500) { /* power-scaled selection (sampling.rs:190-220), stable sort by weight descending; powf is
                         * evaluated by the shared eg_detpow (include/eg_detpow.h), < 2e-14 relative from libm's pow — or, in the
                         * literal-libm mode (og_set_libm_pow), by libm's pow itself, as the reference does */
    int order[OG_NA]; for (int a = 0; a < OG_NA; ++a) order[a] = a;
    for (int i = 1; i < OG_NA; ++i) { int k = order[i], j = i - 1; while (j >= 0 && yw[order[j]] < yw[k]) { order[j + 1] = order[j]; --j; } order[j + 1] = k; }
    double stagnation_factor = mind((double)p->stall / 1000.0, 3.0);
    double power_scaling = 1.0 + (2.0 * stagnation_factor);
    double total_scaled = 0.0; for (int i = 0; i < OG_NA; ++i) total_scaled += stalled_pow(yw[order[i]], power_scaling);
    double random_val = rng_f64(&p->rng) * total_scaled;
    for (int i = 0; i < OG_NA; ++i) { random_val -= stalled_pow(yw[order[i]], power_scaling); if (random_val <= 0.0) return order[i]; }
    return order[0];
  }
  double random_val = rng_f64(&p->rng) * total_weight;
  for (int a = 0; a < OG_NA; ++a) { random_val -= yw[a]; if (random_val <= 0.0) return a; }
  return 3 * T_PEAKER;
}
static int sample_deficit_action(og_weights *p, int yi) { /* sampling.rs:240-378 */
  if (p->force_best) {
    int a;
    if (p->has_best_deficit) {
      int idx = p->replay_def_idx[yi];
      if (idx < p->best_deficit[yi].n) { a = p->best_deficit[yi].a[idx]; p->replay_def_idx[yi] = idx + 1; }
      else a = smart_deficit_fallback_action(p);
    } else a = smart_deficit_fallback_action(p);
    list_push(&p->cur_def[yi], (uint8_t)a);
    return a;
  }
  const double *yw = p->dw[yi];
  int should_explore = rng_f64(&p->rng) < p->exploration_rate;
  if (should_explore) return 3 * DEFICIT_TYPE[rng_range_u64(&p->rng, 14)];
  double total_weight = 0.0; for (int i = 0; i < 14; ++i) total_weight += yw[i];
  if (total_weight <= 0.0) return 3 * T_PEAKER;
  double random_val = rng_f64(&p->rng) * total_weight;
  for (int i = 0; i < 14; ++i) { random_val -= yw[i]; if (random_val <= 0.0) return 3 * DEFICIT_TYPE[i]; }
  return 3 * T_PEAKER;
}
static uint32_t sample_additional_actions(og_weights *p, int yi) { /* sampling.rs:380-443 */
  uint32_t deficit_count = (uint32_t)p->cur_def[yi].n;
  uint32_t max_possible = deficit_count >= 20 ? 0 : 20 - deficit_count;
  if (max_possible == 0) return 0;
  double random_val = rng_f64(&p->rng);
  if (p->has_cw) {
    double total_weight = 0.0; for (int c = 0; c < OG_NC; ++c) total_weight += p->cw[yi][c];
    if (total_weight <= 0.0) return 0;
    double random_choice = random_val * total_weight;
    for (uint32_t c = 0; c < OG_NC; ++c) { random_choice -= p->cw[yi][c]; if (random_choice <= 0.0) return c < max_possible ? c : max_possible; }
    return 5 < max_possible ? 5 : max_possible;
  }
  double scaled_exploration = pow(p->exploration_rate, 0.5);
  uint32_t min_actions = (uint32_t)round_half_away(2.0 / scaled_exploration);
  uint32_t max_actions = (uint32_t)round_half_away(12.0 / scaled_exploration);
  uint32_t capped_max = max_actions < max_possible ? max_actions : max_possible;
  uint32_t capped_min = min_actions < capped_max ? min_actions : capped_max;
  if (capped_min == capped_max) return capped_min;
  return capped_min + rng_range_u32(&p->rng, capped_max - capped_min + 1);
}

/* ------------------------------------------------------------------------- */
/* apply_action (core/actions.rs:40-204)                                       */
/* ------------------------------------------------------------------------- */
typedef struct { og_episode_out *out; int overflow; int trip_cap; double *remaining_log; int *active_log; } rec_t;  /* trip_cap: og_delay_deficit_probe only */

static void map_add_generator(map_t *m, gen_t g, rec_t *rec, int mult_idx) { /* map_handler.rs:553-709 */
  int current_year = m->current_year; double public_opinion = 0.65; /* :1518-1522 */
  if (m->enable_delays) {
    double p = planning_permission_time(g.type, current_year, public_opinion, 1.0);
    double c = construction_time(g.type, current_year, 1.0);
    int estimated = (int)ceil((double)current_year + p + c);
    if (estimated > END_YEAR) return; /* cancelled */
  }
  gen_initialize_construction(&g, current_year, public_opinion, m->enable_delays);
  if (m->ngens == m->gcap) { m->gcap *= 2; m->gens = (gen_t *)realloc(m->gens, sizeof(gen_t) * (size_t)m->gcap); }
  m->gens[m->ngens++] = g;
  og_episode_out *o = rec->out;
  if (o->n_gens < OG_LOG_CAP) {
    o->gen_cell[o->n_gens] = (uint16_t)cell_of(g.c); o->gen_type[o->n_gens] = (uint8_t)g.type;
    o->gen_year[o->n_gens] = (uint8_t)(current_year - OG_BASE_YEAR); o->gen_mult[o->n_gens] = (uint8_t)mult_idx;
    o->n_gens++;
  } else rec->overflow = 1;
}
static void map_add_offset(map_t *m, off_t_ o_, rec_t *rec, int mult_idx) { /* map_handler.rs:785-811 */
  int current_year = m->current_year; double public_opinion = 0.65;
  if (m->enable_delays) {
    double p = offset_planning_time(o_.type, current_year, public_opinion, 1.0);
    double c = offset_construction_time(o_.type, current_year, 1.0);
    int estimated = (int)ceil((double)current_year + p + c);
    if (estimated > END_YEAR) return;
  }
  off_initialize_construction(&o_, current_year, public_opinion, m->enable_delays);
  if (m->noffs == m->ocap) { m->ocap *= 2; m->offs = (off_t_ *)realloc(m->offs, sizeof(off_t_) * (size_t)m->ocap); }
  m->offs[m->noffs++] = o_;
  og_episode_out *o = rec->out;
  if (o->n_offsets < OG_LOG_CAP) {
    o->off_type[o->n_offsets] = (uint8_t)o_.type; o->off_year[o->n_offsets] = (uint8_t)(current_year - OG_BASE_YEAR);
    o->off_mult[o->n_offsets] = (uint8_t)mult_idx; o->n_offsets++;
  } else rec->overflow = 1;
}

static void apply_action(map_t *m, int action, int year, rec_t *rec) {
  if (is_add_generator(action)) {
    int t = action_type(action), mi = action % 3;
    double cost_multiplier = clampd((double)MULT_PERCENT[mi] / 100.0, 1.0, 5.0);
    double gen_size = (double)100 / 100.0; /* DEFAULT_GENERATOR_SIZE, constants.rs:209 */
    coord loc;
    /* Map::find_best_generator_location: location_analysis is None in the parallel driver (Q8) → grid search */
    if (find_suitable_location(m->settle, m->S, m->gens, m->ngens, 0, 0, m->w->coast, m->w->P, t, (float)gen_size, &loc, 0)) {
      gen_t g; memset(&g, 0, sizeof(g));
      g.c = loc; g.type = t; g.existing = 0;
      g.base_cost = type_base_cost(t, year); g.power_out = type_base_power(t);
      g.size = clampd(gen_size, 0.1, 1.0); g.co2_out = co2_rate(t) * gen_size;
      g.efficiency = BASE_EFFICIENCY; g.operation_percentage = 1.0; g.is_active_flag = 1; g.status = ST_PLANNED;
      g.mult = clampd(cost_multiplier, 1.0, 5.0); g.build_year = year;
      map_add_generator(m, g, rec, mi);
    } else {
      /* actions.rs:77-89 type fallback: unreachable with the grid search (Q16); flagged instead of recursing */
      rec->overflow = 2;
    }
  } else if (action < OG_A_UPGRADE) {
    int ot = (action - 45) / 3, mi = (action - 45) % 3;
    static const double size[4] = {500.0, 300.0, 100.0, 1000.0};          /* forest, wetland, active, credit */
    static const double base[4] = {1000000.0, 1000000.0, 1000000000.0, 50000000.0};
    off_t_ o; memset(&o, 0, sizeof(o));
    o.type = ot; o.size = size[ot]; o.base_cost = base[ot]; o.capture_efficiency = clampd(0.85, 0.0, 1.0);
    o.mult = clampd(clampd((double)MULT_PERCENT[mi] / 100.0, 1.0, 5.0), 1.0, 5.0); o.status = ST_PLANNED;
    /* the random coordinate (actions.rs:142-145) is read by no metric and is not drawn */
    map_add_offset(m, o, rec, mi);
  }
  /* UpgradeEfficiency / AdjustOperation / CloseGenerator carry an empty id (core.rs:117-119): the lookup fails and
   * nothing changes; DoNothing is a no-op. */
}

/* ------------------------------------------------------------------------- */
/* episode                                                                     */
/* ------------------------------------------------------------------------- */
static void log_push(uint8_t *log, int32_t *counts, int yi, int action, rec_t *rec) {
  int total = 0; for (int y = 0; y < OG_YEARS; ++y) total += counts[y];
  if (total >= OG_LOG_CAP) { rec->overflow = 1; return; }
  log[total] = (uint8_t)action; counts[yi] += 1;
}

/* simulation.rs:319-522 */
static void handle_power_deficit(map_t *m, double deficit, int yi, og_weights *p, rec_t *rec) {
  int year = OG_BASE_YEAR + yi;
  double remaining = deficit; /* Map::handle_power_deficit returns its input: storage never charges (Q13) */
  uint32_t attempts = 0;
  action_result initial_state = map_state(m, year);
  while (remaining > 0.0) {
    attempts += 1;
    int action = attempts < 5 ? sample_deficit_action(p, yi) : 3 * T_BATTERY;
    action_result current_state = map_state(m, year);
    if (is_add_generator(action)) {
      apply_action(m, action, year, rec);
      list_push(&p->cur_def[yi], (uint8_t)action);
      list_push(&p->cur_run[yi], (uint8_t)action);
      action_result new_state = map_state(m, year);
      double overall = evaluate_impact(&current_state, &new_state);
      double emissions_improvement = new_state.net_emissions < current_state.net_emissions
        ? (current_state.net_emissions - new_state.net_emissions) / maxd(fabs(current_state.net_emissions), 1.0) : 0.0;
      double cost_improvement = 0.0;
      if (new_state.net_emissions < 1000.0) {
        double cost_change = new_state.total_cost - current_state.total_cost;
        cost_improvement = -cost_change / maxd(fabs(current_state.total_cost), 1.0);
      }
      double opinion_improvement = new_state.total_cost < MAX_ACCEPTABLE_COST * 8.0
        ? (new_state.public_opinion - current_state.public_opinion) / maxd(1.0 - current_state.public_opinion, 0.1) : 0.0;
      double combined = overall * 0.7 + emissions_improvement * 0.15 + cost_improvement * 0.1 + opinion_improvement * 0.05;
      update_deficit_weights(p, action, yi, combined);
      update_weights(p, action, yi, overall * 0.5);
      remaining = -mind(new_state.power_balance, 0.0);
    }
    if (rec->trip_cap) { /* og_delay_deficit_probe: what the loop has achieved after this trip */
      int active = 0; for (int g = 0; g < m->ngens; ++g) active += gen_is_active(&m->gens[g]);
      rec->remaining_log[attempts - 1] = remaining; rec->active_log[attempts - 1] = active;
      if ((int)attempts >= rec->trip_cap) { rec->overflow = 3; return; }
    }
    if (rec->overflow) return;
  }
  action_result final_state = map_state(m, year);
  double overall_success = evaluate_impact(&initial_state, &final_state);
  if (final_state.power_balance >= 0.0 && overall_success > 0.0 && p->cur_def[yi].n > 0) {
    double success_factor = 0.1 * overall_success;
    list_t snapshot = {0, 0, 0}; list_copy(&snapshot, &p->cur_def[yi]);
    for (int i = 0; i < snapshot.n; ++i) update_deficit_weights(p, snapshot.a[i], yi, success_factor);
    free(snapshot.a);
  }
}

/* analysis/metrics_calculation.rs:32-175 */
static void yearly_metrics(const map_t *m, int year, int enable_energy_sales, const double *prev, double *row) {
  double total_upgrade_costs = 0.0, total_closure_costs = 0.0;
  uint32_t total_pop = map_total_population(m);
  double usage = map_total_power_usage(m, year);
  double gen = map_total_power_generation(m);
  double balance = gen - usage;
  double co2 = map_total_co2(m), offset = map_total_offset(m, year), net = map_net_co2(m, year);
  double credit = net >= 0.0 ? 0.0 : (-net) * og_carbon_price(year); /* const_funcs.rs:206-218 */
  int active = 0; double opinion = average_opinion(m, year, &active);
  double yearly_capital;
  if (year == 2025) yearly_capital = map_yearly_capital_cost(m, year);
  else if (year > 2025) yearly_capital = map_total_capital_cost(m, year) - map_total_capital_cost(m, year - 1);
  else yearly_capital = 0.0;
  double total_capital = map_total_capital_cost(m, year);
  double inflation = inflation_factor(year);
  double sales = 0.0;
  if (enable_energy_sales && balance > 0.0) { double gwh = balance * 8.76; sales = gwh * 50000.0; } /* const_funcs.rs:225-237 */
  double yearly_total = yearly_capital + total_upgrade_costs + total_closure_costs - credit - (enable_energy_sales ? sales : 0.0);
  row[OG_Y_YEAR] = (double)year; row[OG_Y_POP] = (double)total_pop; row[OG_Y_USAGE] = usage; row[OG_Y_GEN] = gen;
  row[OG_Y_BALANCE] = balance; row[OG_Y_OPINION] = opinion; row[OG_Y_YEARLY_CAPITAL] = yearly_capital;
  row[OG_Y_TOTAL_CAPITAL] = total_capital; row[OG_Y_INFLATION] = inflation; row[OG_Y_CO2] = co2; row[OG_Y_OFFSET] = offset;
  row[OG_Y_NET_CO2] = net; row[OG_Y_YEARLY_CREDIT] = credit;
  row[OG_Y_TOTAL_CREDIT] = prev ? prev[OG_Y_TOTAL_CREDIT] + credit : credit;
  row[OG_Y_YEARLY_SALES] = sales; row[OG_Y_TOTAL_SALES] = prev ? prev[OG_Y_TOTAL_SALES] + sales : sales;
  row[OG_Y_ACTIVE_GENS] = (double)active; row[OG_Y_UPGRADE_COSTS] = total_upgrade_costs; row[OG_Y_CLOSURE_COSTS] = total_closure_costs;
  row[OG_Y_YEARLY_TOTAL_COST] = yearly_total; row[OG_Y_TOTAL_COST] = prev ? prev[OG_Y_TOTAL_COST] + yearly_total : yearly_total;
}

int32_t og_run_episode(const og_world *w, og_weights *weights, int32_t replay_best_strategy, uint64_t seed,
                       int32_t enable_energy_sales, int32_t enable_construction_delays, og_episode_out *out) {
  memset(out, 0, sizeof(*out));
  rec_t rec = {out, 0, 0, 0, 0};
  /* run_iteration: iteration.rs:24-42 */
  map_t map; map_init(&map, w);
  for (int y = 0; y < OG_YEARS; ++y) { list_clear(&weights->cur_run[y]); list_clear(&weights->cur_def[y]); weights->replay_idx[y] = 0; weights->replay_def_idx[y] = 0; }
  weights->force_best = replay_best_strategy ? 1 : 0;
  /* run_simulation: simulation.rs:35-57 (local_weights is this same object; the caller's copy is replaced at :313) */
  map.enable_delays = enable_construction_delays;
  og_weights *p = weights;
  rng_seed(&p->rng, seed); p->has_rng = 1;

  for (int yi = 0; yi < OG_YEARS && !rec.overflow; ++yi) {
    int year = OG_BASE_YEAR + yi;
    map.current_year = year;
    map_update_construction_status(&map);
    if (year > OG_BASE_YEAR) { /* simulation.rs:107-120 */
      for (int s = 0; s < map.S; ++s) {
        uint32_t new_pop = (uint32_t)round_half_away((double)map.settle[s].pop * 1.01);
        map.settle[s].pop = new_pop;
        double per_capita = 0.001 * pow(1.0 + 0.02, (double)(year - OG_BASE_YEAR));
        map.settle[s].usage = (double)new_pop * per_capita;
      }
    }
    action_result current_state = map_state(&map, year);
    if (current_state.power_balance < 0.0) handle_power_deficit(&map, -current_state.power_balance, yi, p, &rec);
    if (rec.overflow) break;
    uint32_t n_additional;
    if (p->force_best) n_additional = p->has_best_actions ? (uint32_t)p->best_actions[yi].n : 0; /* simulation.rs:146-162 */
    else n_additional = sample_additional_actions(p, yi);
    for (uint32_t k = 0; k < n_additional && !rec.overflow; ++k) { /* simulation.rs:189-198 */
      int action = sample_action(p, yi);
      apply_action(&map, action, year, &rec);
      log_push(out->act_log, out->n_act, yi, action, &rec);
      list_push(&p->cur_run[yi], (uint8_t)action);
    }
    yearly_metrics(&map, year, enable_energy_sales, yi > 0 ? out->yearly[yi - 1] : 0, out->yearly[yi]);
  }
  /* iteration.rs:57-74 */
  const double *last = out->yearly[OG_YEARS - 1];
  out->metrics[0] = last[OG_Y_NET_CO2]; out->metrics[1] = last[OG_Y_OPINION]; out->metrics[2] = last[OG_Y_TOTAL_CAPITAL];
  out->metrics[3] = last[OG_Y_BALANCE] >= 0.0 ? 1.0 : 0.0;
  int pr = 0, pd = 0;
  for (int y = 0; y < OG_YEARS; ++y) {
    out->n_run[y] = p->cur_run[y].n; out->n_def[y] = p->cur_def[y].n;
    for (int i = 0; i < p->cur_run[y].n; ++i) { if (pr < OG_LOG_CAP) out->run_log[pr++] = p->cur_run[y].a[i]; else rec.overflow = 1; }
    for (int i = 0; i < p->cur_def[y].n; ++i) { if (pd < OG_LOG_CAP) out->def_log[pd++] = p->cur_def[y].a[i]; else rec.overflow = 1; }
  }
  out->n_draws = p->rng.words;
  out->status = rec.overflow ? -rec.overflow : 0;
  map_free(&map);
  return out->status;
}

/* ------------------------------------------------------------------------- */
/* post-episode update (multi_simulation.rs:494-508)                           */
/* ------------------------------------------------------------------------- */
static void transfer_recorded_actions_from(og_weights *s, const og_weights *o) { /* strategy.rs:313-342 */
  for (int y = 0; y < OG_YEARS; ++y) { list_copy(&s->cur_run[y], &o->cur_run[y]); list_copy(&s->cur_def[y], &o->cur_def[y]); }
}
static void randomize_table(double *t, int n, rng_t *r) { /* learning.rs:267-280 / :356-369; thread_rng in the reference */
  for (int i = 0; i < n; ++i) {
    double random_factor = 1.0 + 0.25 * (rng_f64(r) * 2.0 - 1.0);
    t[i] = clampd(t[i] * random_factor, MIN_WEIGHT, MAX_WEIGHT);
  }
}
static void apply_contrast_learning(og_weights *s, const double cur[4], rng_t *noise) { /* learning.rs:131-283 */
  if (!(s->has_best && s->has_best_actions)) return;
  double best_score = og_score_metrics(s->best_metrics, 0), current_score = og_score_metrics(cur, 0);
  double deterioration = best_score > 0.0 ? (best_score - current_score) / best_score : 0.0;
  double iterations = (double)s->stall;
  double dynamic_threshold = 0.1 * maxd(exp(-iterations / 500.0), 0.00001 / 0.1);
  int force_contrast = s->stall > 800;
  if (!(deterioration > dynamic_threshold || force_contrast)) return;
  double stagnation_iterations = (double)s->stall / 10.0;
  double stagnation_factor = 1.0 + (0.2 * pow(stagnation_iterations, 1.8));
  double scaled_deterioration = pow(deterioration, 0.3);
  double combined_penalty = scaled_deterioration * stagnation_factor;
  double adaptive_lr = s->learning_rate * (1.0 + 0.1 * (double)s->stall);
  double penalty_factor = 1.0 / (1.0 + adaptive_lr * 1.5 * combined_penalty);
  double best_boost_factor = 1.0 + (adaptive_lr * 2.0 * stagnation_factor);
  for (int y = 0; y < OG_YEARS; ++y) {
    list_t current = {0, 0, 0}, best = {0, 0, 0};
    list_copy(&current, &s->cur_run[y]); for (int i = 0; i < s->cur_def[y].n; ++i) list_push(&current, s->cur_def[y].a[i]);
    list_copy(&best, &s->best_actions[y]);
    if (s->has_best_deficit) for (int i = 0; i < s->best_deficit[y].n; ++i) list_push(&best, s->best_deficit[y].a[i]);
    double *yw = s->w[y];
    for (int i = 0; i < best.n; ++i) yw[best.a[i]] = mind(yw[best.a[i]] * best_boost_factor, MAX_WEIGHT);
    for (int i = 0; i < current.n; ++i) {
      int a = current.a[i];
      if (!list_contains(&best, (uint8_t)a)) yw[a] = maxd(yw[a] * penalty_factor, MIN_WEIGHT);
      else if (i < best.n && a != best.a[i]) {
        double mild = 1.0 / (1.0 + adaptive_lr * combined_penalty * 0.5);
        yw[a] = maxd(yw[a] * mild, MIN_WEIGHT);
      }
    }
    free(current.a); free(best.a);
  }
  if (s->stall > 1200) for (int y = 0; y < OG_YEARS; ++y) randomize_table(s->w[y], OG_NA, noise);
}
static void update_best_strategy(og_weights *s, const double m[4]) { /* strategy.rs:19-258 */
  double current_score = og_score_metrics(m, 0);
  s->iteration_count += 1;
  int should_update = !s->has_best || current_score > og_score_metrics(s->best_metrics, 0);
  if (should_update) {
    s->has_best = 1; memcpy(s->best_metrics, m, sizeof(s->best_metrics));
    s->has_best_weights = 1; memcpy(s->best_w, s->w, sizeof(s->best_w));
    for (int y = 0; y < OG_YEARS; ++y) { list_copy(&s->best_actions[y], &s->cur_run[y]); list_copy(&s->best_deficit[y], &s->cur_def[y]); }
    s->has_best_actions = 1; s->has_best_deficit = 1;
    s->stall = 0;
  } else s->stall += 1;
}
static void apply_deficit_contrast_learning(og_weights *s, rng_t *noise) { /* learning.rs:285-373 */
  if (!(s->has_best && s->has_best_deficit)) return;
  double deterioration = (double)s->stall / 10.0;
  double iterations = (double)s->stall;
  double dynamic_threshold = 0.05 * maxd(exp(-iterations / 400.0), 0.00001 / 0.05);
  int force_contrast = s->stall > 800;
  if (!(deterioration > dynamic_threshold || force_contrast)) return;
  double stagnation_iterations = (double)s->stall / 10.0;
  double stagnation_factor = 1.0 + (0.2 * pow(stagnation_iterations, 1.8));
  double scaled_deterioration = pow(deterioration, 0.3);
  double combined_penalty = scaled_deterioration * stagnation_factor;
  double adaptive_lr = s->learning_rate * (1.0 + 0.1 * (double)s->stall);
  double penalty_factor = 1.0 / (1.0 + adaptive_lr * 1.5 * combined_penalty);
  double best_boost_factor = 1.0 + (adaptive_lr * 2.0 * stagnation_factor * 1.5);
  for (int y = 0; y < OG_YEARS; ++y) {
    double *yw = s->dw[y];
    const list_t *best = &s->best_deficit[y], *cur = &s->cur_def[y];
    for (int i = 0; i < best->n; ++i) { int slot = deficit_index_of(best->a[i]); if (slot >= 0) yw[slot] = mind(yw[slot] * best_boost_factor, MAX_WEIGHT); }
    for (int i = 0; i < cur->n; ++i)
      if (!list_contains(best, cur->a[i])) { int slot = deficit_index_of(cur->a[i]); if (slot >= 0) yw[slot] = maxd(yw[slot] * penalty_factor, MIN_WEIGHT); }
  }
  if (s->stall > 1200) for (int y = 0; y < OG_YEARS; ++y) randomize_table(s->dw[y], OG_ND, noise);
}
void og_post_episode_update(og_weights *shared, const og_weights *local, const double metrics[4], uint64_t noise_seed) {
  rng_t noise; rng_seed(&noise, noise_seed);
  transfer_recorded_actions_from(shared, local);
  apply_contrast_learning(shared, metrics, &noise);
  update_best_strategy(shared, metrics);
  apply_deficit_contrast_learning(shared, &noise);
}

/* N4 evidence: the 2025 repair loop of simulation.rs:319-522 with
 * enable_construction_delays = true, stopped after `trips` iterations.  Every plant the loop adds starts "Planned"
 * (generator.rs:451-480), is_active() is false until it is Operational (generator.rs:519-521), so the added output is 0
 * and remaining_deficit never moves: the reference's `while remaining_deficit > 0.0` (simulation.rs:359) does not end. */
int32_t og_delay_deficit_probe(const og_world *w, int32_t trips, double *remaining_after_trip, int32_t *active_after_trip,
                               double *initial_deficit, int32_t *plants_added, int32_t *deficit_year_index) {
  og_episode_out *out = (og_episode_out *)calloc(1, sizeof(*out));
  rec_t rec = {out, 0, trips, remaining_after_trip, active_after_trip};
  map_t map; map_init(&map, w);
  og_weights *p = og_weights_new();
  rng_seed(&p->rng, 12345); p->has_rng = 1;
  map.enable_delays = 1;
  *initial_deficit = 0.0; *plants_added = 0; *deficit_year_index = -1;
  /* the year loop of run_simulation (simulation.rs:89-141) without the additional actions, up to the first year with a deficit:
   * 2025 at HEAD (all existing plant still "Planned", Q1), a later year when the existing plant is operational at the start */
  for (int yi = 0; yi < OG_YEARS; ++yi) {
    int year = OG_BASE_YEAR + yi;
    map.current_year = year;
    map_update_construction_status(&map);
    if (year > OG_BASE_YEAR) {
      for (int s = 0; s < map.S; ++s) {
        uint32_t new_pop = (uint32_t)round_half_away((double)map.settle[s].pop * 1.01);
        map.settle[s].pop = new_pop;
        map.settle[s].usage = (double)new_pop * (0.001 * pow(1.0 + 0.02, (double)(year - OG_BASE_YEAR)));
      }
    }
    action_result s0 = map_state(&map, year);
    if (s0.power_balance < 0.0) {
      *initial_deficit = -s0.power_balance; *deficit_year_index = yi;
      int before = map.ngens;
      handle_power_deficit(&map, -s0.power_balance, yi, p, &rec);
      *plants_added = map.ngens - before;
      break;
    }
  }
  int status = rec.overflow;
  og_weights_free(p); map_free(&map); free(out);
  return status; /* 3: stopped by the trip cap with the deficit still open; 0: no year has a deficit */
}

/* ------------------------------------------------------------------------- */
/* batch ("reduced") update — SURVEY.md §8(e) reduced mode, DESIGN.md §2.4      */
/* ------------------------------------------------------------------------- */
/* INDEPENDENT restatement: written from the definition in DESIGN.md §2.4 / include/eirgrid_hip.h, with libm
 * log / exp / pow and none of the product's formula headers (csrc/eg_reduced_math.h, include/eg_detpow.h are NOT
 * used by anything in this section).  It is the checker of k_apply_update / eg_policy_apply_reduced and of the
 * statistics epilogue of k_rollout.
 *
 * Definition.  All n episodes of the batch were sampled from one snapshot P (the state of `s` on entry):
 *  1. statistics — for every successful episode e: score_e = score_metrics(metrics_e).  If P holds a best strategy
 *     with both lists: deterioration_e = (best − score_e) / best (0 when best ≤ 0); e QUALIFIES for the contrast step
 *     when deterioration_e > threshold(stall) or stall > 800 (learning.rs:139-160; an
 *     episode that beats the best while contrast is forced has a NaN penalty factor in the reference, which its
 *     f64::max turns into MIN_WEIGHT: ln = -20 here).  A qualifying episode adds, for
 *     every action occurrence of its year lists (run ++ deficit) that is absent from best ++ best_deficit of that
 *     year, round(ln(penalty_factor_e)·2^32) to PEN[y][a]; for an occurrence that is present but sits at a position
 *     j < len(best lists) holding another action, round(ln(mild_penalty_e)·2^32) to MILD[y][a] (learning.rs:168-177,
 *     :230-251).  Every successful episode (qualifying or not) counts its deficit actions that are absent from
 *     best_deficit[y] in DCNT[y][slot] (learning.rs:346-352).  Integers: the sums do not depend on the order.
 *  2. main table (apply_contrast_learning, once): when at least one episode qualified, every entry becomes
 *     clamp(clamp(w · exp(n_qual·occ(y,a)·ln(boost))) · exp((PEN + MILD)/2^32)) — boosts first, then penalties, as
 *     the sequential form orders them — with occ = occurrences of a in the best lists of y and boost from the
 *     snapshot's stall counter (learning.rs:180); a stage with a zero exponent leaves the entry untouched; if
 *     stall > 1200 every entry then receives the ±25 % noise (learning.rs:267-280) from StdRng(noise_seed), in (y, a) order.
 *  3. best strategy (update_best_strategy, once): iteration_count += n_ok; the candidate is the successful episode
 *     with the highest score (ties: lowest index); it replaces the best when there is none or its score is strictly
 *     greater (stall = 0, best_weights = the table after step 2), else stall += n_ok.
 *  4. deficit table (apply_deficit_contrast_learning, once, only when step 3 did not improve and P had the lists):
 *     with the NEW stall counter, entry (y, s) becomes clamp(clamp(dw · exp(n_ok·occ_d(y,s)·ln(boost_d))) · exp(DCNT·ln(penalty_d)));
 *     noise beyond 1200 continues the same stream.
 * A batch of ONE episode is the sequential update og_post_episode_update itself, up to the Q32 rounding of the
 * logarithms (tests/test_reduced_oracle.py); a larger batch differs from n_ok sequential updates in that every episode
 * is contrasted against the snapshot's best strategy and the clamps are applied once per stage. */
#define OG_STATS_LEN (8 + 2 * OG_YEARS * OG_NA + OG_YEARS * OG_ND)
static double reduced_stage(double w, double L) {
  if (L == 0.0) return w;
  if (L < -20.0) L = -20.0;   /* a weight lives in [1e-4, 0.999]: beyond e^±9.3 the clamp decides either way */
  if (L > 20.0) L = 20.0;
  return clampd(w * exp(L), MIN_WEIGHT, MAX_WEIGHT);
}
/* all boosts, clamp, all penalties, clamp: the order of the sequential form (learning.rs:214-252) */
static double reduced_nudge(double w, double L_boost, double L_penalty) { return reduced_stage(reduced_stage(w, L_boost), L_penalty); }
int32_t og_reduced_batch_update(og_weights *s, int32_t n, const int32_t *status, const double *metrics /* [n][4] */,
                                const int32_t *n_run /* [n][26] */, const int32_t *n_def /* [n][26] */,
                                const uint8_t *run_log, int32_t run_stride, const uint8_t *def_log, int32_t def_stride,
                                uint64_t noise_seed, int64_t *stats_out /* OG_STATS_LEN or NULL */,
                                int32_t *winner_out /* index of the candidate or -1; may be NULL */) {
  int64_t *st = (int64_t *)calloc(OG_STATS_LEN, sizeof(int64_t));
  int64_t *PEN = st + 8, *MILD = st + 8 + OG_YEARS * OG_NA, *DCNT = st + 8 + 2 * OG_YEARS * OG_NA;
  const int have_lists = s->has_best && s->has_best_actions && s->has_best_deficit;
  const double stall = (double)s->stall;
  const double best_score = s->has_best ? og_score_metrics(s->best_metrics, 0) : 0.0;
  const double threshold = 0.1 * maxd(exp(-stall / 500.0), 0.00001 / 0.1);
  const int forced = s->stall > 800;
  const double stagnation = 1.0 + (0.2 * pow(stall / 10.0, 1.8));
  const double adaptive_lr = s->learning_rate * (1.0 + 0.1 * stall);
  int winner = -1; double winner_score = -1.0;
  for (int e = 0; e < n; ++e) {
    if (status[e] != 0) { st[1] += 1; continue; }
    st[0] += 1;
    const double score = og_score_metrics(metrics + 4 * e, 0);
    if (score > winner_score) { winner = e; winner_score = score; }
    if (!have_lists) continue;
    const double det = best_score > 0.0 ? (best_score - score) / best_score : 0.0;
    const int qualifies = det > threshold || forced;
    int64_t q_pen = 0, q_mild = 0;
    if (qualifies) {
      st[2] += 1;
      if (det < 0.0) {   /* powf(negative, 0.3) = NaN, (w * NaN).max(MIN_WEIGHT) = MIN_WEIGHT: an exponent far below ln(MIN/MAX) */
        q_pen = q_mild = (int64_t)(-20.0 * 4294967296.0);
      } else {
        const double combined = pow(det, 0.3) * stagnation;
        q_pen = (int64_t)llrint(log(1.0 / (1.0 + adaptive_lr * 1.5 * combined)) * 4294967296.0);
        q_mild = (int64_t)llrint(log(1.0 / (1.0 + adaptive_lr * combined * 0.5)) * 4294967296.0);
      }
    }
    const uint8_t *run = run_log + (size_t)e * (size_t)run_stride, *def = def_log + (size_t)e * (size_t)def_stride;
    for (int y = 0; y < OG_YEARS; ++y) {
      const int nr = n_run[e * OG_YEARS + y], nd = n_def[e * OG_YEARS + y];
      const list_t *b = &s->best_actions[y], *bd = &s->best_deficit[y];
      for (int j = 0; j < nr + nd; ++j) {
        const uint8_t a = j < nr ? run[j] : def[j - nr];
        if (qualifies) {
          if (!list_contains(b, a) && !list_contains(bd, a)) PEN[y * OG_NA + a] += q_pen;
          else if (j < b->n + bd->n) {
            const uint8_t at_j = j < b->n ? b->a[j] : bd->a[j - b->n];
            if (at_j != a) MILD[y * OG_NA + a] += q_mild;
          }
        }
        if (j >= nr && !list_contains(bd, a)) { const int slot = deficit_index_of(a); if (slot >= 0) DCNT[y * OG_ND + slot] += 1; }
      }
      run += nr; def += nd;
    }
  }
  if (winner >= 0) { int64_t bits; memcpy(&bits, &winner_score, 8); st[3] = bits + 1; }
  if (stats_out) memcpy(stats_out, st, sizeof(int64_t) * OG_STATS_LEN);
  if (winner_out) *winner_out = winner;
  const int64_t n_ok = st[0], n_qual = st[2];
  rng_t noise; rng_seed(&noise, noise_seed);
  /* 2. main table */
  if (have_lists && n_qual > 0) {
    const double ln_boost = log(1.0 + (adaptive_lr * 2.0 * stagnation));
    for (int y = 0; y < OG_YEARS; ++y)
      for (int a = 0; a < OG_NA; ++a) {
        int occ = 0;
        for (int i = 0; i < s->best_actions[y].n; ++i) occ += s->best_actions[y].a[i] == a;
        for (int i = 0; i < s->best_deficit[y].n; ++i) occ += s->best_deficit[y].a[i] == a;
        s->w[y][a] = reduced_nudge(s->w[y][a], (double)n_qual * (double)occ * ln_boost,
                                   ((double)PEN[y * OG_NA + a] + (double)MILD[y * OG_NA + a]) / 4294967296.0);
      }
    if (s->stall > 1200) for (int y = 0; y < OG_YEARS; ++y) randomize_table(s->w[y], OG_NA, &noise);
  }
  /* 3. best strategy */
  s->iteration_count += (uint32_t)n_ok;
  int improved = 0;
  if (winner >= 0) improved = !s->has_best || winner_score > og_score_metrics(s->best_metrics, 0);
  if (improved) {
    s->has_best = 1; memcpy(s->best_metrics, metrics + 4 * winner, sizeof(s->best_metrics));
    s->has_best_weights = 1; memcpy(s->best_w, s->w, sizeof(s->best_w));
    const uint8_t *run = run_log + (size_t)winner * (size_t)run_stride, *def = def_log + (size_t)winner * (size_t)def_stride;
    for (int y = 0; y < OG_YEARS; ++y) {
      const int nr = n_run[winner * OG_YEARS + y], nd = n_def[winner * OG_YEARS + y];
      list_clear(&s->best_actions[y]); for (int i = 0; i < nr; ++i) list_push(&s->best_actions[y], run[i]);
      list_clear(&s->best_deficit[y]); for (int i = 0; i < nd; ++i) list_push(&s->best_deficit[y], def[i]);
      list_copy(&s->cur_run[y], &s->best_actions[y]); list_copy(&s->cur_def[y], &s->best_deficit[y]);
      run += nr; def += nd;
    }
    s->has_best_actions = 1; s->has_best_deficit = 1; s->stall = 0;
  } else s->stall += (uint32_t)n_ok;
  /* 4. deficit table, with the stall counter step 3 left */
  if (!improved && have_lists) {
    const double k = (double)s->stall;
    const double deterioration = k / 10.0;
    const double dthreshold = 0.05 * maxd(exp(-k / 400.0), 0.00001 / 0.05);
    if (deterioration > dthreshold || s->stall > 800) {
      const double dstag = 1.0 + (0.2 * pow(k / 10.0, 1.8));
      const double dcombined = pow(deterioration, 0.3) * dstag;
      const double dlr = s->learning_rate * (1.0 + 0.1 * k);
      const double ln_pen = log(1.0 / (1.0 + dlr * 1.5 * dcombined));
      const double ln_boost = log(1.0 + (dlr * 2.0 * dstag * 1.5));
      for (int y = 0; y < OG_YEARS; ++y)
        for (int sl = 0; sl < OG_ND; ++sl) {
          int occ = 0;
          for (int i = 0; i < s->best_deficit[y].n; ++i) occ += deficit_index_of(s->best_deficit[y].a[i]) == sl;
          s->dw[y][sl] = reduced_nudge(s->dw[y][sl], (double)n_ok * (double)occ * ln_boost, (double)DCNT[y * OG_ND + sl] * ln_pen);
        }
      if (s->stall > 1200) for (int y = 0; y < OG_YEARS; ++y) randomize_table(s->dw[y], OG_ND, &noise);
    }
  }
  free(st);
  return improved;
}

/* ------------------------------------------------------------------------- */
/* KAT helpers                                                                 */
/* ------------------------------------------------------------------------- */
void og_world_demand(const og_world *w, int32_t yi, uint32_t *total_pop, double *total_usage) {
  map_t m; map_init(&m, w);
  for (int k = 1; k <= yi; ++k) {
    int year = OG_BASE_YEAR + k;
    for (int s = 0; s < m.S; ++s) {
      uint32_t new_pop = (uint32_t)round_half_away((double)m.settle[s].pop * 1.01);
      m.settle[s].pop = new_pop;
      m.settle[s].usage = (double)new_pop * (0.001 * pow(1.0 + 0.02, (double)(year - OG_BASE_YEAR)));
    }
  }
  *total_pop = map_total_population(&m); *total_usage = map_total_power_usage(&m, OG_BASE_YEAR + yi);
  map_free(&m);
}
int32_t og_world_existing_online(const og_world *w, int32_t g) {
  gen_t c = w->existing[g];
  for (int yi = 0; yi < OG_YEARS; ++yi) { gen_update_construction_status(&c, OG_BASE_YEAR + yi); if (gen_is_active(&c)) return yi; }
  return OG_YEARS;
}
double og_type_power_output(int32_t t) {
  gen_t g; memset(&g, 0, sizeof(g));
  g.type = t; g.power_out = type_base_power(t); g.efficiency = BASE_EFFICIENCY; g.operation_percentage = 1.0;
  g.is_active_flag = 1; g.status = ST_OPERATIONAL;
  return gen_power_output(&g);
}
double og_offset_full_effect(int32_t ot) {
  static const double size[4] = {500.0, 300.0, 100.0, 1000.0};
  off_t_ o; memset(&o, 0, sizeof(o));
  o.type = ot; o.size = size[ot]; o.capture_efficiency = 0.85; o.status = ST_OPERATIONAL; o.construction_complete_year = 2025;
  if (ot == O_FOREST || ot == O_WETLAND) { /* asymptote: maturity factor -> 1 */
    double base = o.size * (ot == O_FOREST ? 25.0 : 40.0); return base * o.capture_efficiency;
  }
  return off_calc(&o, 2030);
}
double og_generator_cost(int32_t t, int32_t build_year, int32_t year, int32_t mult_percent) {
  gen_t g; memset(&g, 0, sizeof(g));
  g.type = t; g.base_cost = type_base_cost(t, build_year); g.mult = clampd((double)mult_percent / 100.0, 1.0, 5.0);
  return gen_current_cost(&g, year);
}
/* utils/csv_export.rs:249-266 (AddGenerator) and :343-366 (AddCarbonOffset): the "Estimated Cost" column of
 * simulation_summary.csv; every other action prints 0.00 because the table's generator ids are empty. */
double og_action_cost_estimate(int32_t action, int32_t year) {
  if (action >= 0 && action < 45) {
    int t = action / 3;
    static const int percent[3] = {100, 120, 150};
    double base_cost = type_base_cost(t, year);
    double accurate_cost = calc_generator_cost(t, base_cost, year, can_be_urban(t), requires_water(t), requires_water(t));
    return accurate_cost * ((double)percent[action % 3] / 100.0);
  }
  if (action >= 45 && action < 57) {
    static const double base[4] = {1000000.0, 1000000.0, 1000000000.0, 50000000.0}; /* Forest, Wetland, ActiveCapture, CarbonCredit */
    static const int percent[3] = {100, 120, 150};
    double adjusted_cost = base[(action - 45) / 3] * inflation_factor(year);
    return adjusted_cost * ((double)percent[(action - 45) % 3] / 100.0);
  }
  return 0.0;
}
int32_t og_place(const og_world *w, int32_t yi, int32_t type, int32_t n_extra, const double *ex, const double *ey,
                 double *best_score) { return og_place_sized(w, yi, type, n_extra, ex, ey, 1.0f, best_score); }
int32_t og_place_sized(const og_world *w, int32_t yi, int32_t type, int32_t n_extra, const double *ex, const double *ey,
                       float size_penalty, double *best_score) {
  map_t m; map_init(&m, w);
  for (int k = 1; k <= yi; ++k)
    for (int s = 0; s < m.S; ++s) m.settle[s].pop = (uint32_t)round_half_away((double)m.settle[s].pop * 1.01);
  coord *extra = (coord *)malloc(sizeof(coord) * (size_t)(n_extra > 0 ? n_extra : 1));
  for (int i = 0; i < n_extra; ++i) extra[i] = coord_new(ex[i], ey[i]);
  coord loc; double score = 0.0;
  int found = find_suitable_location(m.settle, m.S, m.gens, m.ngens, extra, n_extra, w->coast, w->P, type, size_penalty, &loc, &score);
  if (best_score) *best_score = score;
  free(extra); map_free(&m);
  return found ? cell_of(loc) : -1;
}

/* ------------------------------------------------------------------------- */
/* tabled mode                                                                 */
/* ------------------------------------------------------------------------- */
typedef struct {
  double co2, tg, ig, sg, optot, gcost, ocost, gcost_prev, ocost_prev, offs, usage; int opcnt;
} agg_t;
typedef struct { int cell, type, year, mult; } tgen_t;
typedef struct { int type, year, mult; } toff_t;
typedef struct {
  const og_tables *T; tgen_t *gens; int ngens; toff_t *offs; int noffs; agg_t a; int yi; double *fld;
} tmap_t;

static action_result tstate(const agg_t *a) {
  action_result r;
  r.net_emissions = a->co2 - a->offs;
  r.public_opinion = a->opcnt > 0 ? a->optot / (double)a->opcnt : 1.0;
  r.power_balance = ((a->tg + a->ig) + a->sg) - a->usage;
  r.total_cost = a->gcost + a->ocost;
  return r;
}
static const double *cc_at(const og_tables *T, int yi, int t, int b, int m) {
  return T->cc + ((((size_t)yi * OG_NTYPES + (size_t)t) * OG_YEARS + (size_t)b) * 3 + (size_t)m) * 2;
}
static int tplace(tmap_t *m, int t) {
  const og_tables *T = m->T;
  int rc = T->rclass[t], reach = T->reach[rc];
  const double *te = T->te + ((size_t)m->yi * 6 + (size_t)rc) * 2601;
  const double *dr = T->dr + (size_t)rc * 169;
  for (int c = 0; c < 2601; ++c) m->fld[c] = te[c];
  for (int g = 0; g < m->ngens; ++g) {
    int gi = m->gens[g].cell / 51, gj = m->gens[g].cell % 51;
    for (int di = -reach; di <= reach; ++di) for (int dj = -reach; dj <= reach; ++dj) {
      int ci = gi + di, cj = gj + dj;
      if (ci < 0 || ci > 50 || cj < 0 || cj > 50) continue;
      m->fld[ci * 51 + cj] *= dr[(di < 0 ? -di : di) * 13 + (dj < 0 ? -dj : dj)];
    }
  }
  double best = 0.0; int best_c = -1;
  for (int c = 0; c < 2601; ++c) {
    double s = m->fld[c];
    if (T->marine[t]) s *= T->coastf[c];
    s *= T->size_factor;
    if (s > best) { best = s; best_c = c; }
  }
  return best_c;
}
static void tapply(tmap_t *m, int action, rec_t *rec) {
  const og_tables *T = m->T; agg_t *a = &m->a; int yi = m->yi;
  og_episode_out *o = rec->out;
  if (is_add_generator(action)) {
    int t = action / 3, mi = action % 3;
    int cell = tplace(m, t);
    if (cell < 0) { rec->overflow = 2; return; }
    if (m->ngens >= OG_LOG_CAP) { rec->overflow = 1; return; }
    tgen_t g = {cell, t, yi, mi}; m->gens[m->ngens++] = g;
    o->gen_cell[o->n_gens] = (uint16_t)cell; o->gen_type[o->n_gens] = (uint8_t)t; o->gen_year[o->n_gens] = (uint8_t)yi;
    o->gen_mult[o->n_gens] = (uint8_t)mi; o->n_gens++;
    const double *cc = cc_at(T, yi, t, yi, mi);
    a->gcost += cc[0];
    if (yi > 0) a->gcost_prev += cc_at(T, yi - 1, t, yi, mi)[0];
    a->co2 += T->co2_t[t];
    if (T->cls[t] == 1) a->ig += T->out_mw[t]; else if (T->cls[t] == 2) a->sg += T->out_mw[t]; else a->tg += T->out_mw[t];
    a->optot += (T->m03[cell] + T->t12[(size_t)yi * OG_NTYPES + (size_t)t]) + cc[1];
    a->opcnt += 1;
  } else if (action < OG_A_UPGRADE) {
    int ot = (action - 45) / 3, mi = (action - 45) % 3;
    if (m->noffs >= OG_LOG_CAP) { rec->overflow = 1; return; }
    toff_t f = {ot, yi, mi}; m->offs[m->noffs++] = f;
    o->off_type[o->n_offsets] = (uint8_t)ot; o->off_year[o->n_offsets] = (uint8_t)yi; o->off_mult[o->n_offsets] = (uint8_t)mi; o->n_offsets++;
    a->offs += T->offv[((size_t)yi * 4 + (size_t)ot) * OG_YEARS + (size_t)yi];
    a->ocost += T->offc[((size_t)yi * 4 + (size_t)ot) * 3 + (size_t)mi];
    if (yi > 0) a->ocost_prev += T->offc[((size_t)(yi - 1) * 4 + (size_t)ot) * 3 + (size_t)mi];
  }
}
static void thandle_power_deficit(tmap_t *m, og_weights *p, rec_t *rec) {
  int yi = m->yi;
  action_result initial_state = tstate(&m->a);
  double remaining = -initial_state.power_balance;
  uint32_t attempts = 0;
  while (remaining > 0.0) {
    attempts += 1;
    int action = attempts < 5 ? sample_deficit_action(p, yi) : 3 * T_BATTERY;
    action_result current_state = tstate(&m->a);
    if (is_add_generator(action)) {
      tapply(m, action, rec);
      if (rec->overflow) return;
      list_push(&p->cur_def[yi], (uint8_t)action);
      list_push(&p->cur_run[yi], (uint8_t)action);
      action_result new_state = tstate(&m->a);
      double overall = evaluate_impact(&current_state, &new_state);
      double emissions_improvement = new_state.net_emissions < current_state.net_emissions
        ? (current_state.net_emissions - new_state.net_emissions) / maxd(fabs(current_state.net_emissions), 1.0) : 0.0;
      double cost_improvement = 0.0;
      if (new_state.net_emissions < 1000.0) {
        double cost_change = new_state.total_cost - current_state.total_cost;
        cost_improvement = -cost_change / maxd(fabs(current_state.total_cost), 1.0);
      }
      double opinion_improvement = new_state.total_cost < MAX_ACCEPTABLE_COST * 8.0
        ? (new_state.public_opinion - current_state.public_opinion) / maxd(1.0 - current_state.public_opinion, 0.1) : 0.0;
      double combined = overall * 0.7 + emissions_improvement * 0.15 + cost_improvement * 0.1 + opinion_improvement * 0.05;
      update_deficit_weights(p, action, yi, combined);
      update_weights(p, action, yi, overall * 0.5);
      remaining = -mind(new_state.power_balance, 0.0);
    }
    if (attempts > 100000u) { rec->overflow = 1; return; }
  }
  action_result final_state = tstate(&m->a);
  double overall_success = evaluate_impact(&initial_state, &final_state);
  if (final_state.power_balance >= 0.0 && overall_success > 0.0 && p->cur_def[yi].n > 0) {
    double success_factor = 0.1 * overall_success;
    list_t snapshot = {0, 0, 0}; list_copy(&snapshot, &p->cur_def[yi]);
    for (int i = 0; i < snapshot.n; ++i) update_deficit_weights(p, snapshot.a[i], yi, success_factor);
    free(snapshot.a);
  }
}
int32_t og_run_episode_tabled(const og_tables *T, og_weights *p, int32_t replay, uint64_t seed, int32_t enable_energy_sales,
                              og_episode_out *out) {
  memset(out, 0, sizeof(*out));
  rec_t rec = {out, 0, 0, 0, 0};
  tmap_t m; memset(&m, 0, sizeof(m));
  m.T = T; m.gens = (tgen_t *)malloc(sizeof(tgen_t) * OG_LOG_CAP); m.offs = (toff_t *)malloc(sizeof(toff_t) * OG_LOG_CAP);
  m.fld = (double *)malloc(sizeof(double) * 2601);
  for (int y = 0; y < OG_YEARS; ++y) { list_clear(&p->cur_run[y]); list_clear(&p->cur_def[y]); p->replay_idx[y] = 0; p->replay_def_idx[y] = 0; }
  p->force_best = replay ? 1 : 0;
  rng_seed(&p->rng, seed); p->has_rng = 1;
  double gcost_end = 0.0, ocost_end = 0.0;
  for (int yi = 0; yi < OG_YEARS && !rec.overflow; ++yi) {
    int year = OG_BASE_YEAR + yi; m.yi = yi;
    agg_t *a = &m.a;
    a->co2 = T->pre_co2[yi]; a->tg = T->pre_tg[yi]; a->ig = T->pre_ig[yi]; a->sg = T->pre_sg[yi];
    a->optot = T->pre_optot[yi]; a->opcnt = T->pre_opcnt[yi]; a->usage = T->usage[yi];
    a->gcost = 0.0; a->ocost = 0.0; a->offs = 0.0; a->gcost_prev = gcost_end; a->ocost_prev = ocost_end;
    for (int g = 0; g < m.ngens; ++g) {
      const tgen_t *G = &m.gens[g];
      const double *cc = cc_at(T, yi, G->type, G->year, G->mult);
      a->gcost += cc[0]; a->co2 += T->co2_t[G->type];
      if (T->cls[G->type] == 1) a->ig += T->out_mw[G->type]; else if (T->cls[G->type] == 2) a->sg += T->out_mw[G->type]; else a->tg += T->out_mw[G->type];
      a->optot += (T->m03[G->cell] + T->t12[(size_t)yi * OG_NTYPES + (size_t)G->type]) + cc[1];
      a->opcnt += 1;
    }
    for (int k = 0; k < m.noffs; ++k) {
      const toff_t *F = &m.offs[k];
      a->offs += T->offv[((size_t)yi * 4 + (size_t)F->type) * OG_YEARS + (size_t)F->year];
      a->ocost += T->offc[((size_t)yi * 4 + (size_t)F->type) * 3 + (size_t)F->mult];
    }
    action_result s0 = tstate(a);
    if (s0.power_balance < 0.0) thandle_power_deficit(&m, p, &rec);
    if (rec.overflow) break;
    uint32_t n_additional;
    if (p->force_best) n_additional = p->has_best_actions ? (uint32_t)p->best_actions[yi].n : 0;
    else n_additional = sample_additional_actions(p, yi);
    for (uint32_t k = 0; k < n_additional && !rec.overflow; ++k) {
      int action = sample_action(p, yi);
      tapply(&m, action, &rec);
      log_push(out->act_log, out->n_act, yi, action, &rec);
      list_push(&p->cur_run[yi], (uint8_t)action);
    }
    if (rec.overflow) break;
    action_result s = tstate(a);
    double *row = out->yearly[yi]; const double *prev = yi > 0 ? out->yearly[yi - 1] : 0;
    double gen = (a->tg + a->ig) + a->sg;
    double credit = s.net_emissions >= 0.0 ? 0.0 : (-s.net_emissions) * T->carbon_price[yi];
    double total_capital = a->gcost + a->ocost;
    double yearly_capital = yi == 0 ? total_capital : total_capital - (a->gcost_prev + a->ocost_prev);
    double sales = 0.0;
    if (enable_energy_sales && s.power_balance > 0.0) { double gwh = s.power_balance * 8.76; sales = gwh * 50000.0; }
    double yearly_total = yearly_capital + 0.0 + 0.0 - credit - (enable_energy_sales ? sales : 0.0);
    row[OG_Y_YEAR] = (double)year; row[OG_Y_POP] = T->population[yi]; row[OG_Y_USAGE] = a->usage; row[OG_Y_GEN] = gen;
    row[OG_Y_BALANCE] = s.power_balance; row[OG_Y_OPINION] = s.public_opinion; row[OG_Y_YEARLY_CAPITAL] = yearly_capital;
    row[OG_Y_TOTAL_CAPITAL] = total_capital; row[OG_Y_INFLATION] = T->inflation[yi]; row[OG_Y_CO2] = a->co2; row[OG_Y_OFFSET] = a->offs;
    row[OG_Y_NET_CO2] = s.net_emissions; row[OG_Y_YEARLY_CREDIT] = credit;
    row[OG_Y_TOTAL_CREDIT] = prev ? prev[OG_Y_TOTAL_CREDIT] + credit : credit;
    row[OG_Y_YEARLY_SALES] = sales; row[OG_Y_TOTAL_SALES] = prev ? prev[OG_Y_TOTAL_SALES] + sales : sales;
    row[OG_Y_ACTIVE_GENS] = (double)a->opcnt; row[OG_Y_UPGRADE_COSTS] = 0.0; row[OG_Y_CLOSURE_COSTS] = 0.0;
    row[OG_Y_YEARLY_TOTAL_COST] = yearly_total; row[OG_Y_TOTAL_COST] = prev ? prev[OG_Y_TOTAL_COST] + yearly_total : yearly_total;
    gcost_end = a->gcost; ocost_end = a->ocost;
  }
  const double *last = out->yearly[OG_YEARS - 1];
  out->metrics[0] = last[OG_Y_NET_CO2]; out->metrics[1] = last[OG_Y_OPINION]; out->metrics[2] = last[OG_Y_TOTAL_CAPITAL];
  out->metrics[3] = last[OG_Y_BALANCE] >= 0.0 ? 1.0 : 0.0;
  int pr = 0, pd = 0;
  for (int y = 0; y < OG_YEARS; ++y) {
    out->n_run[y] = p->cur_run[y].n; out->n_def[y] = p->cur_def[y].n;
    for (int i = 0; i < p->cur_run[y].n; ++i) { if (pr < OG_LOG_CAP) out->run_log[pr++] = p->cur_run[y].a[i]; else rec.overflow = 1; }
    for (int i = 0; i < p->cur_def[y].n; ++i) { if (pd < OG_LOG_CAP) out->def_log[pd++] = p->cur_def[y].a[i]; else rec.overflow = 1; }
  }
  out->n_draws = p->rng.words;
  out->status = rec.overflow ? -rec.overflow : 0;
  free(m.gens); free(m.offs); free(m.fld);
  return out->status;
}

double og_detpow(double x, double p) { return eg_detpow(x, p); }
double og_libm_pow(double x, double p) { return pow(x, p); }
