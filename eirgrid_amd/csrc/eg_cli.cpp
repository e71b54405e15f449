// eirgrid-hip — command-line training driver on top of libeirgrid_hip.so.
//
// Reproduces the *contract* of the reference driver, not its code: the 18 flags of cli/cli.rs:5-59 with the same names
// and defaults, the data files initialize_map reads (main.rs:74-193), the run-directory / checkpoint layout of
// run_multi_simulation (core/multi_simulation.rs:160-165, :210-290, :396-404, :544-567, :1160-1164) and its schedule
// (replay of the best strategy in "full" runs: the last 10 % of the iterations, or always when the location cache is
// absent, :38-39, :437-465).  Iterations run on the GPU in batches that share one weights snapshot — the GPU
// counterpart of rayon workers cloning the shared weights (:457-460) — and are folded into the weights either one by
// one in index order (--update sequential: multi_simulation.rs:494-508 verbatim) or with the batch form
// (--update reduced, default).  The best-run CSV export (--enable-csv-export, N3) writes what the
// reference's exporter writes: simulation_summary.csv, improvement_history.csv, yearly_details/{settlements,generators,
// carbon_offsets}.csv and operation_logs/generator_operation_logs.csv (csrc/eg_export.cpp describes their contents).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include <dirent.h>
#include <sys/stat.h>

#include "eg_json.h"
#include "eirgrid_hip.h"

namespace {

struct Args {   // cli/cli.rs:5-59
  uint64_t iterations = 1000; bool parallel = true; bool no_continue = false; std::string checkpoint_dir = "checkpoints";
  uint64_t checkpoint_interval = 5; uint64_t progress_interval = 10; std::string cache_dir = "cache";
  bool force_full_simulation = false, enable_timing = false; bool has_seed = false; uint64_t seed = 0;
  bool verbose_state_logging = false, cost_only = false, enable_energy_sales = true, enable_csv_export = true;
  bool debug_logging = false, debug_weights = false, enable_construction_delays = false, track_weight_history = false;
  // engine-specific
  std::string world_json, assets_dir = "aiSimulator/assets"; uint32_t batch = 1024; std::string update = "reduced"; int device = 0;
  bool existing_operational_at_start = false;
  uint64_t stop_after = 0;      // leave the loop (as an interrupt would) once this many iterations are done and checkpointed
  std::string dump_world;       // write the loaded world (eirgrid_amd JSON form) there and exit: no device needed
  bool replay_hoist = true;     // the replay iterations of a batch computed once (eg_replay_hoist): the same results, the replay phases 5x faster
};

void usage() {
  std::puts("Usage: eirgrid-hip [OPTIONS]\n"
            "  -n, --iterations <N>            [default: 1000]\n  -p, --parallel\n      --no-continue\n"
            "  -c, --checkpoint-dir <DIR>      [default: checkpoints]\n  -i, --checkpoint-interval <N>   [default: 5]\n"
            "  -r, --progress-interval <SECS>  [default: 10]\n  -C, --cache-dir <DIR>           [default: cache]\n"
            "      --force-full-simulation\n      --enable-timing\n      --seed <SEED>\n  -v, --verbose-state-logging\n"
            "      --cost-only\n      --enable-energy-sales\n      --enable-csv-export\n      --debug-logging\n      --debug-weights\n"
            "      --enable-construction-delays\n      --track-weight-history\n"
            "engine options:\n      --world <FILE>       world in eirgrid_amd JSON form (default: read <assets-dir> like the reference)\n"
            "      --assets-dir <DIR>   settlements.json, ireland_generators.csv, coastline_points.json [default: aiSimulator/assets]\n"
            "      --batch <B>          iterations per GPU launch [default: 1024]\n      --update <sequential|reduced>  [default: reduced]\n"
            "      --device <N>         [default: 0]\n      --existing-operational-at-start\n"
            "      --stop-after <N>     stop like an interrupted run once N iterations are done and checkpointed (resume tests)\n"
            "      --dump-world <FILE>  write the world as loaded (the --world JSON form) and exit; needs no GPU\n"
            "      --no-replay-hoist    run every replay iteration of a batch on its own (default: the replay iterations of a batch —\n"
            "                           one and the same computation — are computed once; identical results either way)");
}

bool parse(int argc, char** argv, Args& a) {
  auto need = [&](int& i) -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "error: %s needs a value\n", argv[i]); std::exit(2); } return argv[++i]; };
  for (int i = 1; i < argc; ++i) {
    std::string s = argv[i], val;
    const size_t eq = s.find('=');
    bool has_val = false;
    if (s.rfind("--", 0) == 0 && eq != std::string::npos) { val = s.substr(eq + 1); s = s.substr(0, eq); has_val = true; }
    auto v = [&]() -> std::string { return has_val ? val : std::string(need(i)); };
    if (s == "-n" || s == "--iterations") a.iterations = std::strtoull(v().c_str(), nullptr, 10);
    else if (s == "-p" || s == "--parallel") a.parallel = true;
    else if (s == "--no-continue") a.no_continue = true;
    else if (s == "-c" || s == "--checkpoint-dir") a.checkpoint_dir = v();
    else if (s == "-i" || s == "--checkpoint-interval") a.checkpoint_interval = std::max<uint64_t>(1, std::strtoull(v().c_str(), nullptr, 10));
    else if (s == "-r" || s == "--progress-interval") a.progress_interval = std::strtoull(v().c_str(), nullptr, 10);
    else if (s == "-C" || s == "--cache-dir") a.cache_dir = v();
    else if (s == "--force-full-simulation") a.force_full_simulation = true;
    else if (s == "--enable-timing") a.enable_timing = true;
    else if (s == "--seed") { a.seed = std::strtoull(v().c_str(), nullptr, 10); a.has_seed = true; }
    else if (s == "-v" || s == "--verbose-state-logging") a.verbose_state_logging = true;
    else if (s == "--cost-only") a.cost_only = true;
    else if (s == "--enable-energy-sales") a.enable_energy_sales = true;     // SetTrue flags with default true (Q7)
    else if (s == "--enable-csv-export") a.enable_csv_export = true;
    else if (s == "--debug-logging") a.debug_logging = true;
    else if (s == "--debug-weights") a.debug_weights = true;
    else if (s == "--enable-construction-delays") a.enable_construction_delays = true;
    else if (s == "--track-weight-history") a.track_weight_history = true;
    else if (s == "--world") a.world_json = v();
    else if (s == "--assets-dir") a.assets_dir = v();
    else if (s == "--batch") a.batch = uint32_t(std::max<uint64_t>(1, std::strtoull(v().c_str(), nullptr, 10)));
    else if (s == "--update") a.update = v();
    else if (s == "--device") a.device = std::atoi(v().c_str());
    else if (s == "--existing-operational-at-start") a.existing_operational_at_start = true;
    else if (s == "--stop-after") a.stop_after = std::strtoull(v().c_str(), nullptr, 10);
    else if (s == "--dump-world") a.dump_world = v();
    else if (s == "--no-replay-hoist") a.replay_hoist = false;
    else if (s == "-h" || s == "--help") { usage(); std::exit(0); }
    else { std::fprintf(stderr, "error: unexpected argument '%s'\n", argv[i]); usage(); return false; }
  }
  return a.update == "sequential" || a.update == "reduced";
}

bool read_file(const std::string& path, std::string& out) {
  std::ifstream f(path, std::ios::binary); if (!f) return false;
  std::stringstream ss; ss << f.rdbuf(); out = ss.str(); return true;
}
bool exists(const std::string& p) { struct stat st; return ::stat(p.c_str(), &st) == 0; }
void mkdirs(const std::string& p) { std::string cur; for (size_t i = 0; i <= p.size(); ++i) { if (i == p.size() || p[i] == '/') { if (!cur.empty()) ::mkdir(cur.c_str(), 0755); } if (i < p.size()) cur += p[i]; } }

struct WorldData {
  std::vector<double> sx, sy; std::vector<uint32_t> spop; std::vector<double> gx, gy, gcap; std::vector<int32_t> gtype; std::vector<double> cx, cy;
  std::vector<std::string> names;      // settlement names (settlements.csv of the export); empty: "Settlement_<i>"
  eg_world view(bool at_start) const {
    eg_world w{}; w.n_settlements = int32_t(sx.size()); w.settlement_x = sx.data(); w.settlement_y = sy.data(); w.settlement_pop = spop.data();
    w.n_existing = int32_t(gx.size()); w.existing_x = gx.data(); w.existing_y = gy.data(); w.existing_type = gtype.data(); w.existing_capacity_mw = gcap.data();
    w.n_coast = int32_t(cx.size()); w.coast_x = cx.data(); w.coast_y = cy.data(); w.existing_operational_at_start = at_start ? 1 : 0; return w;
  }
};
bool parse_json(const std::string& path, eg::Json& root) {
  std::string text; if (!read_file(path, text)) return false;
  eg::JsonParser ps{text.data(), text.data() + text.size(), {}};
  return ps.value(root);
}
void nums(const eg::Json* j, std::vector<double>& out) { if (j && j->kind == eg::Json::Arr) for (auto& v : j->arr) out.push_back(v.num); }

bool load_world_json(const std::string& path, WorldData& w) {   // eirgrid_amd.world.World.to_json_dict
  eg::Json r; if (!parse_json(path, r) || r.kind != eg::Json::Obj) return false;
  std::vector<double> pop, type;
  nums(r.get("settlement_x"), w.sx); nums(r.get("settlement_y"), w.sy); nums(r.get("settlement_pop"), pop);
  nums(r.get("existing_x"), w.gx); nums(r.get("existing_y"), w.gy); nums(r.get("existing_type"), type); nums(r.get("existing_capacity"), w.gcap);
  nums(r.get("coast_x"), w.cx); nums(r.get("coast_y"), w.cy);
  for (double p : pop) w.spop.push_back(uint32_t(p));
  for (double t : type) w.gtype.push_back(int32_t(t));
  if (const eg::Json* nm = r.get("settlement_names"); nm && nm->kind == eg::Json::Arr && nm->arr.size() == w.sx.size())
    for (auto& v : nm->arr) w.names.push_back(v.str);
  return !w.sx.empty() && w.sx.size() == w.sy.size() && w.sx.size() == w.spop.size() && w.gx.size() == w.gtype.size();
}
// const_funcs.rs:124-136 + constants.rs:131-134, :270-271
bool lat_lon_to_grid(double lat, double lon, double& x, double& y) {
  if (lat < 51.4 || lat > 55.4 || lon < -10.6 || lon > -5.9) return false;
  x = (lon - -10.6) * 10638.297872340427; y = (lat - 51.4) * 12500.0;
  x = std::min(std::max(x, 0.0), 50000.0); y = std::min(std::max(y, 0.0), 50000.0);
  return true;
}
bool load_reference_assets(const std::string& dir, WorldData& w) {   // main.rs:74-193
  eg::Json s;
  if (!parse_json(dir + "/settlements.json", s)) return false;
  const eg::Json* list = s.get("settlements");
  if (!list || list->kind != eg::Json::Arr) return false;
  for (auto& e : list->arr) {   // data/settlements_loader.rs:23-41
    const eg::Json *lat = e.get("lat"), *lon = e.get("lon"), *pop = e.get("population");
    double x, y;
    if (lat && lon && pop && lat_lon_to_grid(lat->num, lon->num, x, y)) {
      w.sx.push_back(x); w.sy.push_back(y); w.spop.push_back(uint32_t(pop->num));
      const eg::Json* name = e.get("name"); w.names.push_back(name ? name->str : "Settlement_" + std::to_string(w.names.size()));
    }
  }
  std::string csv;
  if (!read_file(dir + "/ireland_generators.csv", csv)) return false;
  std::istringstream in(csv); std::string line; bool header = true;
  while (std::getline(in, line)) {   // data/generators_loader.rs:133-206
    if (header) { header = false; continue; }
    if (line.empty()) continue;
    std::vector<std::string> col; std::stringstream ls(line); std::string c;
    while (std::getline(ls, c, ',')) col.push_back(c);
    if (col.size() < 4) continue;
    std::string fuel = col[3]; while (!fuel.empty() && (fuel.back() == '\r' || fuel.back() == ' ')) fuel.pop_back();
    std::transform(fuel.begin(), fuel.end(), fuel.begin(), ::tolower);
    int t = fuel == "gas" ? 7 : fuel == "coal" ? 6 : fuel == "wind" ? 0 : fuel == "hydro" ? 10 : fuel == "oil" ? 8 : fuel == "biomass" ? 9 : -1;
    if (t < 0) { std::fprintf(stderr, "Invalid fuel type: %s\n", col[3].c_str()); return false; }
    double lat = std::min(std::max(std::atof(col[1].c_str()), 51.4), 55.4), lon = std::min(std::max(std::atof(col[2].c_str()), -10.6), -5.9), x, y;
    lat_lon_to_grid(lat, lon, x, y);
    w.gx.push_back(x); w.gy.push_back(y); w.gtype.push_back(t); w.gcap.push_back(std::atof(col[0].c_str()));
  }
  eg::Json cst;
  if (parse_json(dir + "/coastline_points.json", cst))
    if (const eg::Json* g = cst.get("grid_coords"); g && g->kind == eg::Json::Arr)
      for (auto& pt : g->arr) if (pt.kind == eg::Json::Arr && pt.arr.size() >= 2) { w.cx.push_back(pt.arr[0].num); w.cy.push_back(pt.arr[1].num); }
  return !w.sx.empty();
}

std::string newest_run_dir(const std::string& base) {   // multi_simulation.rs:217-235: 15-character names, year <= 2025
  std::string best;
  if (DIR* d = ::opendir(base.c_str())) {
    while (dirent* e = ::readdir(d)) {
      std::string n = e->d_name;
      if (n.size() != 15 || n[8] != '_') continue;
      if (std::atoi(n.substr(0, 4).c_str()) > 2025) continue;
      if (!exists(base + "/" + n + "/latest_weights.json")) continue;
      if (n > best) best = n;
    }
    ::closedir(d);
  }
  return best.empty() ? best : base + "/" + best;
}

#define CHECK(call) do { int32_t rc_ = (call); if (rc_ < 0) { std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, eg_last_error()); return 1; } } while (0)

}  // namespace

int main(int argc, char** argv) {
  Args a;
  if (!parse(argc, argv, a)) return 2;
  std::puts("EirGrid Power System Simulator (2025-2050) — MI355X rollout engine");
  if (a.enable_construction_delays) { std::fprintf(stderr, "error: --enable-construction-delays is not implemented on the device (DESIGN.md §6)\n"); return 2; }

  WorldData wd;
  if (!a.world_json.empty()) { if (!load_world_json(a.world_json, wd)) { std::fprintf(stderr, "error: cannot read world %s\n", a.world_json.c_str()); return 1; } }
  else if (!load_reference_assets(a.assets_dir, wd)) { std::fprintf(stderr, "error: cannot read %s/{settlements.json,ireland_generators.csv} (use --world or --assets-dir)\n", a.assets_dir.c_str()); return 1; }
  std::printf("World: %zu settlements, %zu existing generators, %zu coastline points\n", wd.sx.size(), wd.gx.size(), wd.cx.size());
  {  // the fuel mix as data/generators_loader.rs:47-57 maps it (gas -> GasCombinedCycle, oil -> GasPeaker, ...)
    static const char* kTypeName[EG_N_TYPES] = {"OnshoreWind", "OffshoreWind", "DomesticSolar", "CommercialSolar", "UtilitySolar", "Nuclear", "CoalPlant",
                                                "GasCombinedCycle", "GasPeaker", "Biomass", "HydroDam", "PumpedStorage", "BatteryStorage", "TidalGenerator", "WaveEnergy"};
    int count[EG_N_TYPES] = {0};
    for (int32_t t : wd.gtype) if (t >= 0 && t < EG_N_TYPES) count[t] += 1;
    std::string mix;
    for (int t = 0; t < EG_N_TYPES; ++t) if (count[t]) mix += (mix.empty() ? "" : ", ") + std::string(kTypeName[t]) + " " + std::to_string(count[t]);
    std::printf("Existing generators by type: %s\n", mix.c_str());
  }
  if (!a.dump_world.empty()) {
    std::ofstream f(a.dump_world);
    if (!f) { std::fprintf(stderr, "error: cannot write %s\n", a.dump_world.c_str()); return 1; }
    auto arr = [&](const char* key, const std::vector<double>& v, bool last = false) {
      f << "\"" << key << "\": [";
      char buf[40];
      for (size_t i = 0; i < v.size(); ++i) { std::snprintf(buf, sizeof(buf), "%.17g", v[i]); f << (i ? ", " : "") << buf; }
      f << "]" << (last ? "" : ", ");
    };
    f << "{";
    arr("settlement_x", wd.sx); arr("settlement_y", wd.sy); arr("settlement_pop", std::vector<double>(wd.spop.begin(), wd.spop.end()));
    arr("existing_x", wd.gx); arr("existing_y", wd.gy); arr("existing_type", std::vector<double>(wd.gtype.begin(), wd.gtype.end()));
    arr("existing_capacity", wd.gcap); arr("coast_x", wd.cx); arr("coast_y", wd.cy);
    f << "\"existing_operational_at_start\": " << (a.existing_operational_at_start ? "true" : "false") << "}\n";
    std::printf("World written to %s\n", a.dump_world.c_str());
    return 0;
  }
  const eg_world world = wd.view(a.existing_operational_at_start);
  eg_ctx* ctx = eg_create(a.device, &world);
  if (!ctx) { std::fprintf(stderr, "eg_create: %s\n", eg_last_error()); return 1; }
  // The reference's replay phases (the last 10 % of a run, --force-full-simulation: core/multi_simulation.rs:38-39, :437-465) run the
  // same replay in every iteration of a batch: computed once unless asked otherwise (worlds the hoist is not sized for: every
  // iteration on its own, silently — the results are the same)
  if (a.replay_hoist) (void)eg_replay_hoist(ctx, 1);

  // run directory and resume (multi_simulation.rs:160-165, :210-290, :396-404)
  eg_policy* policy = nullptr; uint64_t start_iteration = 0; std::string run_dir;
  if (!a.no_continue) {
    run_dir = newest_run_dir(a.checkpoint_dir);
    if (!run_dir.empty()) {
      policy = eg_policy_load_json((run_dir + "/latest_weights.json").c_str());
      if (policy) {
        std::string it; if (read_file(run_dir + "/checkpoint_iteration.txt", it)) start_iteration = std::strtoull(it.c_str(), nullptr, 10);
        std::printf("Loaded weights from %s (iteration %llu)\n", run_dir.c_str(), (unsigned long long)start_iteration);
      } else { std::fprintf(stderr, "warning: %s\n", eg_last_error()); run_dir.clear(); }
    }
  }
  if (!policy) policy = eg_policy_new();
  if (run_dir.empty()) {
    char buf[32]; std::time_t t = std::time(nullptr); std::tm tmv; localtime_r(&t, &tmv);
    std::strftime(buf, sizeof(buf), "%m%d_%H%M%S", &tmv);
    run_dir = a.checkpoint_dir + "/2024" + buf;   // the literal "2024" of multi_simulation.rs:162 (Q17)
  }
  mkdirs(run_dir);
  if (!a.has_seed) { std::random_device rd; a.seed = (uint64_t(rd()) << 32) | rd(); }
  const bool cache_loaded = exists(a.cache_dir + "/location_analysis.json");   // multi_simulation.rs:150-154
  std::printf("Starting multi-simulation optimization with %llu iterations (%llu completed, %llu remaining) in directory %s\n",
              (unsigned long long)a.iterations, (unsigned long long)start_iteration,
              (unsigned long long)(a.iterations > start_iteration ? a.iterations - start_iteration : 0), run_dir.c_str());

  // The run that is summarised and exported at the end: the reference's `best_result`, a fold over this process's iterations in
  // iteration order that starts at None (core/multi_simulation.rs:384, :613-620; --cost-only reaches it as optimization_mode).
  // The library folds every batch on the device behind its rollout (eg_best_result_track) and keeps the held run's record.
  struct BestRun {
    std::vector<double> metrics = std::vector<double>(4), yearly = std::vector<double>(size_t(EG_YEARS) * EG_YEARLY_FIELDS);
    std::vector<int32_t> n_act = std::vector<int32_t>(EG_YEARS); std::vector<uint8_t> act_log = std::vector<uint8_t>(EG_ACT_CAP);
    std::vector<uint16_t> gen_pack = std::vector<uint16_t>(EG_MAX_GENS); int32_t n_gens = 0;
    eg_episode_out view{}; bool valid = false;
    BestRun() { view.metrics = metrics.data(); view.yearly = yearly.data(); view.n_act = n_act.data(); view.act_log = act_log.data();
                view.n_gens = &n_gens; view.gen_pack = gen_pack.data(); }
  } best_run;
  std::vector<uint8_t> mask;
  std::vector<double> metrics; std::vector<int32_t> n_run, n_def; std::vector<uint8_t> run_log, def_log;
  eg_opts opts{a.enable_energy_sales ? 1 : 0, 0, a.enable_csv_export ? 1 : 0};   // the export needs the best episode's yearly rows
  const uint64_t final_full = a.iterations * 10 / 100;   // FULL_RUN_PERCENTAGE, multi_simulation.rs:38, :437
  auto t0 = std::chrono::steady_clock::now(); auto last_progress = t0;
  uint64_t done = start_iteration, last_checkpoint = start_iteration / a.checkpoint_interval;
  const bool reduced = a.update == "reduced";
  unsigned failed_sequential = 0;      // --update sequential: episodes that did not finish (reduced mode counts them on the device)
  // reduced mode keeps the policy on the device: pushed once, every batch is enqueued without a host round trip and the
  // host copy is refreshed (eg_policy_pull) when a checkpoint or a progress line needs it
  if (reduced) CHECK(eg_policy_push(ctx, policy, &opts));
  CHECK(eg_best_result_track(ctx, a.cost_only ? 2 : 1));
  const uint64_t full_from = a.iterations - std::min(a.iterations, final_full);   // multi_simulation.rs:437-465
  while (done < a.iterations) {
    uint32_t n = uint32_t(std::min<uint64_t>(a.batch, a.iterations - done));
    const bool always_full = a.force_full_simulation || !cache_loaded;
    if (reduced) {
      if (!always_full && done < full_from && done + n > full_from) n = uint32_t(full_from - done);   // a batch never straddles the switch
      const bool full = always_full || done >= full_from;
      CHECK(eg_device_step(ctx, a.seed, done, n, full ? 1u : 0u, a.seed + done));   // replay (once a best strategy exists) when full
    } else {
      eg_policy_snapshot snap; CHECK(eg_policy_snapshot_view(policy, &snap));
      CHECK(eg_upload_snapshot(ctx, &snap, &opts));
      mask.assign(n, 0);
      for (uint32_t i = 0; i < n; ++i) {
        const bool full = always_full || done + i >= full_from;
        mask[i] = (full && snap.has_best && snap.best_count) ? 1 : 0;
      }
      metrics.resize(size_t(n) * 4); n_run.resize(size_t(n) * EG_YEARS); n_def.resize(size_t(n) * EG_YEARS);
      run_log.resize(size_t(n) * EG_RUN_CAP); def_log.resize(size_t(n) * EG_DEF_CAP);
      std::vector<int32_t> status(n);
      eg_episode_out out{}; out.metrics = metrics.data(); out.n_run = n_run.data(); out.n_def = n_def.data(); out.run_log = run_log.data();
      out.def_log = def_log.data(); out.status = status.data();
      CHECK(eg_rollout_launch(ctx, a.seed, done, n, mask.data()));
      CHECK(eg_fetch(ctx, &out));
      for (uint32_t i = 0; i < n; ++i)   // multi_simulation.rs:494-508, in iteration order
        if (status[i] != EG_EP_OK) failed_sequential += 1;
        else
          CHECK(eg_policy_apply_episode(policy, &metrics[size_t(i) * 4], &n_run[size_t(i) * EG_YEARS], &run_log[size_t(i) * EG_RUN_CAP],
                                        &n_def[size_t(i) * EG_YEARS], &def_log[size_t(i) * EG_DEF_CAP], a.seed + done + i));
    }
    done += n;
    const auto now = std::chrono::steady_clock::now();
    const bool checkpoint_due = done / a.checkpoint_interval != last_checkpoint || done == a.iterations;
    const bool progress_due = std::chrono::duration<double>(now - last_progress).count() >= double(a.progress_interval) || done == a.iterations;
    if (reduced && (checkpoint_due || progress_due)) CHECK(eg_policy_pull(ctx, policy));
    if (checkpoint_due) {   // multi_simulation.rs:544-567
      last_checkpoint = done / a.checkpoint_interval;
      CHECK(eg_policy_save_json(policy, (run_dir + "/thread_0_weights.json").c_str()));
      CHECK(eg_policy_save_json(policy, (run_dir + "/latest_weights.json").c_str()));
      if (a.track_weight_history) CHECK(eg_policy_append_weight_history(policy, (run_dir + "/weight_history.json").c_str(), done - 1));   // :557-560
      std::ofstream(run_dir + "/checkpoint_iteration.txt") << done;
    }
    if (progress_due) {   // :303-382
      last_progress = now;
      const double secs = std::chrono::duration<double>(now - t0).count();
      double bm[4] = {eg_policy_get_scalar(policy, 5), eg_policy_get_scalar(policy, 6), eg_policy_get_scalar(policy, 7), eg_policy_get_scalar(policy, 8)};
      // (failed: episodes that ran out of the per-episode capacities — replay-doubled lists; they are not part of
      //  iteration_count and never silently dropped from the count)
      std::printf("Progress: %llu/%llu iterations (%.1f%%), %.0f iterations/s | best score %.6f (emissions %.1f t, cost EUR %.2fB, opinion %.1f%%), %u without improvement, %u episodes failed\n",
                  (unsigned long long)done, (unsigned long long)a.iterations, 100.0 * double(done) / double(a.iterations),
                  double(done - start_iteration) / std::max(secs, 1e-9), eg_policy_get_scalar(policy, 4) != 0.0 ? eg_score_metrics(bm, a.cost_only ? 1 : 0) : 0.0,
                  bm[0], bm[2] / 1e9, bm[1] * 100.0, unsigned(eg_policy_get_scalar(policy, 2)), unsigned(eg_policy_get_scalar(policy, 13)) + failed_sequential);
      std::fflush(stdout);
    }
    if (a.stop_after && done >= a.stop_after && done < a.iterations) {
      if (!checkpoint_due) { std::fprintf(stderr, "error: --stop-after needs a checkpoint at the stop (use -i 1 or a divisor)\n"); return 2; }
      std::printf("Stopped after %llu iterations (--stop-after); resume with the same command\n", (unsigned long long)done);
      eg_policy_free(policy); eg_destroy(ctx);
      return 0;
    }
  }
  CHECK(eg_policy_save_json(policy, (run_dir + "/best_weights.json").c_str()));   // multi_simulation.rs:1160-1164
  char stamp[32]; std::string dir;
  if (a.enable_csv_export) {   // multi_simulation.rs:852-859, :912-921; csv_export.rs:114-127 (directory named after the time of export)
    std::time_t t = std::time(nullptr); std::tm tmv; localtime_r(&t, &tmv);
    std::strftime(stamp, sizeof(stamp), "%Y%m%d_%H%M%S", &tmv);
    dir = run_dir + "/enhanced_csv/" + stamp;
    mkdirs(dir);
    CHECK(eg_policy_export_improvement_csv(policy, (dir + "/improvement_history.csv").c_str()));
  }
  int64_t best_index = -1;
  { int32_t state = 0; CHECK(eg_fetch_best_result(ctx, &best_run.view, &state, &best_index)); best_run.valid = state == 1; }
  if (best_run.valid) {   // multi_simulation.rs:821-850
    const double* bm = best_run.metrics.data();
    std::printf("BEST SIMULATION RESULTS SUMMARY (iteration %lld)\nFinal net emissions: %.2f tonnes\nEmissions Status: %s\nAverage public opinion: %.1f%%\n"
                "Total cost: EUR %.2f billion accumulated\nPower reliability: %.1f%%\n", (long long)best_index, bm[0],
                bm[0] <= 0.0 ? "NET ZERO ACHIEVED" : "NET ZERO NOT ACHIEVED", bm[1] * 100.0, bm[2] / 1e9, bm[3] * 100.0);
  }
  if (a.enable_csv_export) {
    if (best_run.valid) {
      CHECK(eg_export_summary_csv(&best_run.view, (dir + "/simulation_summary.csv").c_str(), stamp));
      std::vector<const char*> names;
      for (const std::string& n : wd.names) names.push_back(n.c_str());
      CHECK(eg_export_run_details(&world, names.size() == wd.sx.size() ? names.data() : nullptr, &best_run.view, dir.c_str(), a.seed));
    } else std::puts("note: no iteration finished in this run; simulation_summary.csv and the detail files not written");
  }
  std::printf("Done: %llu iterations in %s (%u episodes failed); best_weights.json, latest_weights.json, checkpoint_iteration.txt written\n",
              (unsigned long long)done, run_dir.c_str(), unsigned(eg_policy_get_scalar(policy, 13)) + failed_sequential);
  eg_policy_free(policy);
  eg_destroy(ctx);
  return 0;
}
