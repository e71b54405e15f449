// eg_api.cpp — C ABI glue: context, HBM residency of tables / snapshot / outputs, launches, timing.
// There is no CPU execution path behind these entry points: without a HIP device eg_create fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <cstdlib>
#include <string>
#include <vector>

#include "eg_internal.h"
#include "eg_policy_internal.h"
#define EG_RM static inline
#include "eg_reduced_math.h"

namespace eg {
namespace {
thread_local std::string g_error;
}
void set_error(const std::string& s) { g_error = s; }
}  // namespace eg

using namespace eg;

#define EG_HIP(call)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      set_error(std::string(#call) + ": " + hipGetErrorString(e_));                           \
      return EG_ERR_HIP;                                                                      \
    }                                                                                         \
  } while (0)

struct eg_host_tables {
  HostTables H;
  std::map<std::string, std::pair<const double*, int64_t>> f64;
  std::map<std::string, std::pair<const int32_t*, int64_t>> i32;
  void index() {
    auto F = [&](const char* n, const std::vector<double>& v) { f64[n] = {v.data(), (int64_t)v.size()}; };
    auto I = [&](const char* n, const std::vector<int32_t>& v) { i32[n] = {v.data(), (int64_t)v.size()}; };
    F("usage", H.usage); F("population", H.population); F("pre_co2", H.pre_co2); F("pre_tg", H.pre_tg); F("pre_ig", H.pre_ig);
    F("pre_sg", H.pre_sg); F("pre_optot", H.pre_optot); F("te", H.te); F("coastf", H.coastf); F("dr", H.dr); F("m03", H.m03);
    F("t12", H.t12); F("cc", H.cc); F("out_mw", H.out_mw); F("co2_t", H.co2_t); F("offv", H.offv); F("offc", H.offc);
    F("inflation", H.inflation); F("carbon_price", H.carbon_price);
    f64["size_factor"] = {&H.size_factor, 1};
    I("pre_opcnt", H.pre_opcnt); I("cls", H.cls); I("rclass", H.rclass); I("marine", H.marine); I("reach", H.reach);
    I("existing_online", H.existing_online);
  }
};

struct eg_ctx {
  int device = 0;
  eg_host_tables tables;
  std::vector<void*> allocs;        // table allocations
  DevTables dev{};
  // snapshot in HBM
  // the whole snapshot lives in ONE device buffer filled by ONE copy from a pinned staging buffer
  uint8_t* d_snap = nullptr; uint8_t* h_snap = nullptr;
  uint8_t* d_snap_held = nullptr;      // eg_policy_hold / eg_policy_rewind
  // the reference's best_result fold (multi_simulation.rs:613-620): 0 = not tracked, 1 = optimization_mode None, 2 = cost_only
  int fold_mode = 0; uint8_t* d_fold = nullptr;
  // Is the best list long (the replay episodes run the heavy-capable variant and are the batch's long pole)?  `list_exact`: the host
  // KNOWS the list the next launch will find on the device (it uploaded, rewound or pulled it and no on-device update has been
  // enqueued since): the replay variant that has nothing to do is then not launched at all.  Otherwise the device may have replaced
  // the list since the host last looked — both variants are launched and decide for themselves — and the hint only orders the
  // launches; it follows the device through `h_list_len`, a pinned host word that k_apply_update and k_rewind write the list's
  // length to (read without synchronising: as old as the launch queue is deep).
  bool long_list_hint = false, long_list_hint_held = false, list_exact = false, list_exact_held = false;
  uint32_t* h_list_len = nullptr; uint32_t* d_list_len = nullptr;      // the same pinned word, host and device address
  DevSnapshot snap{};
  bool snap_valid = false;
  // outputs
  DevOut out{};
  uint32_t out_cap = 0, last_n = 0;
  uint64_t last_first = 0;      // global index of the first episode of the last batch
  uint8_t* d_mask = nullptr; uint32_t mask_cap = 0;
  // timing: a ring of event pairs riding on the rollout dispatches.  A pair is only waited for when the ring comes round to
  // it again (kTimingRing launches later: long finished) or when the caller reads the timing — never inside a training step.
  static constexpr int kTimingRing = 256;
  hipEvent_t ev[kTimingRing][4] = {};       // start / stop of the heavy grid, start / stop of the lean grid (eg_internal.h RolloutPlan)
  uint8_t ev_used[kTimingRing] = {};        // bit 0: the heavy pair was recorded, bit 1: the lean pair
  hipStream_t stream_heavy = nullptr;   // the replay grids of a split batch run beside the lean grid (which stays on the null stream)
  hipEvent_t ev_fork[kTimingRing] = {}, ev_go[kTimingRing] = {}, ev_join[kTimingRing] = {};
  uint32_t* d_index = nullptr; uint32_t index_cap = 0;         // replay / other episode indices of a host-masked batch
  int ring_head = 0, ring_pending = 0;      // next pair to use; pairs recorded and not yet collected (the oldest is head - pending)
  double total_ms = 0.0; int32_t n_launches = 0;
  double grids_ms = 0.0;      // the same launches, every grid's own duration added up (== total_ms when a batch is one grid)
  // eg_place / eg_find_suitable_location: device buffers kept between calls
  uint16_t* d_place_cells = nullptr; int32_t* d_place_cell = nullptr; double* d_place_score = nullptr;
  double* d_place_xy = nullptr; int32_t place_xy_cap = 0;
  uint32_t push_iteration_count = 0;     // iteration counter written into the device state by the next upload
  uint32_t push_failed = 0;              // ... and the failed-episode counter
  uint32_t pulled_improvements = 0;      // on-device improvement log entries already appended to a host policy
  // eg_train_step / eg_device_step: library-owned update packet (device) and its pinned host copy
  uint8_t* d_packet = nullptr; uint8_t* h_packet = nullptr;
  // batches of at most this many episodes run the helper-wave kernel (three waves per episode, all resident at once)
  uint32_t helper_max_episodes = 0;
  // heavy episodes (eg_rollout.hip place_heavy): pool of penalty fields, one slot per episode that outgrows kHeavyGens
  // EIRGRID_HEAVY_POOL_GB (default 64): what the pool may grow to, 126 KB per replay episode of a launch; 131 072 replay episodes
  // (an all-replay batch of configs[3]'s size) want 16 GB.  eg_memory_report tells what is held.
  uint32_t heavy_slots_max = uint32_t((size_t(64) << 30) / (size_t(kRadiusClasses) * 2624 * sizeof(double)));
  uint32_t heavy_slots_wanted = 4096, launch_epoch = 0;
  bool heavy_slots_auto = true;      // (EIRGRID_HEAVY_SLOTS fixes the pool size instead)
  // replay hoist (eg_replay_coop.h; eg_replay_hoist / EIRGRID_REPLAY_HOIST=1): the replay episodes of a batch computed once.
  // d_hoist: {u64 sequence number of the last batch whose hoist succeeded, i32 lengths[5]}; d_coop: the scratch record.
  bool hoist_on = false, hoist_supported = false;
  unsigned long long hoist_seq = 0;
  HoistInfo* d_hoist = nullptr; uint8_t* d_coop = nullptr;
  uint64_t hoist_batches = 0;      // batches launched with the hoist armed (eg_replay_hoist_stats)
  int coop_force = 0;              // EIRGRID_COOP_FORCE (test hook): the hoisted searches' rarely-run paths
  long long* d_stats_rep = nullptr;      // kStatsReplicas copies of the statistics array (RolloutPlan::d_stats_rep); EIRGRID_STATS_REPLICAS=0: none
  // per-episode replay kernel (eg_replay_solo.h; EIRGRID_REPLAY_SOLO=0: off): a word per replay episode of a launch, the launches' sequence
  unsigned long long* d_solo = nullptr; uint32_t solo_cap = 0; unsigned long long solo_seq = 0;
  bool solo_on = true;
};

namespace {

template <typename T>
void put(std::vector<uint8_t>& blob, size_t off, const std::vector<T>& v, size_t max_count) {
  const size_t n = v.size() < max_count ? v.size() : max_count;
  if (n) std::memcpy(blob.data() + off, v.data(), sizeof(T) * n);
}

void free_outputs(eg_ctx* c) {
  if (c->out.base) (void)hipFree(c->out.base);
  c->out = DevOut{}; c->out_cap = 0;
}

int ensure_outputs(eg_ctx* c, uint32_t n) {
  if (n <= c->out_cap) return EG_OK;
  free_outputs(c);
  EG_HIP(hipMalloc((void**)&c->out.base, size_t(n) * rec::stride + size_t(n) * sizeof(double)));
  c->out.score_list = reinterpret_cast<double*>(c->out.base + size_t(n) * rec::stride);
  // zero once so episodes that end early leave defined year counts / rows behind
  EG_HIP(hipMemset(c->out.base, 0, size_t(n) * rec::stride + size_t(n) * sizeof(double)));
  c->out_cap = n;
  return EG_OK;
}

// collects the oldest `count` recorded launches (all of them when count < 0).  A batch that ran as several grids counts as
// ONE launch lasting from the earlier start to the later end.
int collect_timing(eg_ctx* c, int count = -1) {
  if (count < 0 || count > c->ring_pending) count = c->ring_pending;
  for (; count > 0; --count) {
    const int i = (c->ring_head - c->ring_pending + 2 * eg_ctx::kTimingRing) % eg_ctx::kTimingRing;
    const bool heavy = c->ev_used[i] & 1, lean = c->ev_used[i] & 2;
    float best = 0.f, sum = 0.f;
    for (int g = 0; g < 2; ++g)
      if (g ? lean : heavy) {
        EG_HIP(hipEventSynchronize(c->ev[i][2 * g + 1]));
        float ms = 0.f;
        EG_HIP(hipEventElapsedTime(&ms, c->ev[i][2 * g], c->ev[i][2 * g + 1]));
        sum += ms;
      }
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b) {
        if (!(a ? lean : heavy) || !(b ? lean : heavy)) continue;
        EG_HIP(hipEventSynchronize(c->ev[i][2 * b + 1]));
        float ms = 0.f;
        EG_HIP(hipEventElapsedTime(&ms, c->ev[i][2 * a], c->ev[i][2 * b + 1]));
        if (ms > best) best = ms;
      }
    c->total_ms += double(best); c->grids_ms += double(sum); c->n_launches += 1; c->ring_pending -= 1;
  }
  return EG_OK;
}
// before every rollout launch: the field pool holds a slot for every heavy episode of the launch (allocated on first use,
// enlarged when a launch brings more of them: a replay episode without a slot falls back to the exact scan, 20-40x slower,
// and a launch lasts as long as its slowest episode) and the launch has an epoch of its own
int prepare_heavy(eg_ctx* c, uint32_t n_heavy, bool known_short) {
  constexpr size_t kSlotBytes = size_t(kRadiusClasses) * 2624 * sizeof(double);
  // No pool while the host KNOWS the best list to be short: replay episodes of a short list never ask for a slot.  (Not by the
  // pinned hint: the host enqueues a free-running loop many batches ahead of the device, the hint is as old as the queue is deep,
  // and a long replay episode without a slot takes the exact scan — 33 instead of 5 ms per batch, measured.)
  if (known_short && !c->dev.heavy) n_heavy = 0;
  uint32_t want = c->heavy_slots_wanted;
  if (c->heavy_slots_auto && want > 0) {      // 4 096 slots (516 MB) to begin with, then the next power of two, up to the budget
    while (want < n_heavy && want < c->heavy_slots_max) want = want * 2u < c->heavy_slots_max ? want * 2u : c->heavy_slots_max;
  }
  if (want > 0 && n_heavy > 0 && (!c->dev.heavy || want > c->dev.heavy_slots)) {
    void* pool = nullptr;
    if (c->dev.heavy) {      // earlier launches may still use the old pool
      EG_HIP(hipDeviceSynchronize());
      for (auto it = c->allocs.begin(); it != c->allocs.end(); ++it) if (*it == c->dev.heavy) { c->allocs.erase(it); break; }
      (void)hipFree(c->dev.heavy);
      c->dev.heavy = nullptr; c->dev.heavy_slots = 0;
    }
    if (!c->dev.heavy_claim) {
      void* claim = nullptr;
      if (hipMalloc(&claim, 64) != hipSuccess) { (void)hipGetLastError(); c->heavy_slots_wanted = 0; want = 0; }
      else { EG_HIP(hipMemset(claim, 0xFF, 64)); c->allocs.push_back(claim); c->dev.heavy_claim = static_cast<unsigned*>(claim); }      // 0xFF..: an epoch no launch uses
    }
    while (want > 0 && hipMalloc(&pool, kSlotBytes * want) != hipSuccess) {      // no memory for that many: fewer; none: heavy episodes take the exact scan
      (void)hipGetLastError();
      pool = nullptr;
      want = want > 4096u ? want / 2u : 0u;
      c->heavy_slots_auto = false; c->heavy_slots_wanted = want;
    }
    if (pool) { c->allocs.push_back(pool); c->dev.heavy = static_cast<uint8_t*>(pool); c->dev.heavy_slots = want; }
  }
  c->launch_epoch = (c->launch_epoch + 1u) & 0xFFFu;
  if (c->launch_epoch == 0xFFFu) c->launch_epoch = 0u;      // 0xFFF is the "never" epoch the claim word starts with
  c->dev.heavy_epoch = c->launch_epoch;
  return EG_OK;
}
// One batch = up to three grids of k_rollout (eg_internal.h RolloutPlan): the episodes that replay the best strategy on the
// two replay variants (one of which returns at once), the others on the lean one, on two streams side by side.  `host_mask` (n bytes, may be NULL):
// which episodes replay; otherwise `period` (0: none): episode i replays when (first_index + i) % period == 0.
// Uploads the mask and the index lists, takes a slot of the timing ring, gives the launch its field-pool epoch, launches.
int launch_batch(eg_ctx* c, uint64_t seed, uint64_t first_index, uint32_t n, const uint8_t* host_mask, uint32_t period, long long* d_stats) {
  if (n == 0) return EG_OK;
  RolloutPlan plan{};
  plan.helper_waves = n <= c->helper_max_episodes;
  plan.n_lean = n;
  const uint8_t* d_mask = nullptr;
  if (host_mask) {
    if (n > c->mask_cap) { if (c->d_mask) (void)hipFree(c->d_mask); c->d_mask = nullptr; EG_HIP(hipMalloc((void**)&c->d_mask, n)); c->mask_cap = n; }
    if (n > c->index_cap) { if (c->d_index) (void)hipFree(c->d_index); c->d_index = nullptr; EG_HIP(hipMalloc((void**)&c->d_index, sizeof(uint32_t) * n)); c->index_cap = n; }
    std::vector<uint32_t> idx(n);
    uint32_t nh = 0;
    for (uint32_t i = 0; i < n; ++i) if (host_mask[i]) idx[nh++] = i;
    uint32_t k = nh;
    for (uint32_t i = 0; i < n; ++i) if (!host_mask[i]) idx[k++] = i;
    EG_HIP(hipMemcpy(c->d_mask, host_mask, n, hipMemcpyHostToDevice));
    EG_HIP(hipMemcpy(c->d_index, idx.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice));
    d_mask = c->d_mask;
    plan.n_heavy = nh; plan.n_lean = n - nh; plan.mode = 1u; plan.d_index = c->d_index;
  } else if (period == 1u) {
    plan.n_heavy = n; plan.n_lean = 0;
  } else if (period > 1u) {
    plan.off = uint32_t((uint64_t(period) - first_index % period) % period);
    plan.n_heavy = plan.off < n ? (n - plan.off + period - 1u) / period : 0u;
    plan.n_lean = n - plan.n_heavy; plan.mode = 2u; plan.period = period;
  }
  if (c->ring_pending == eg_ctx::kTimingRing) { int rc = collect_timing(c, 1); if (rc != EG_OK) return rc; }
  const int slot = c->ring_head;
  for (int k = 0; k < 4; ++k) plan.ev[k] = c->ev[slot][k];
  // A batch that is one kind of grid stays on the null stream like every other kernel of the library.  With both kinds the lean
  // grid still does; the replay grids go to a (non-blocking) stream of their own that forks off the null stream before them and
  // joins it after the lean grid's launch — two explicit events.  (They were two blocking streams at first: the implicit
  // null-stream synchronisation of those cost 30 us before the grids and 16 us after them, every batch.)
  const bool split = plan.n_heavy > 0 && plan.n_lean > 0;
  plan.stream_heavy = split ? c->stream_heavy : nullptr;
  plan.stream_lean = nullptr;
  // what the host knows about the best list: exactly (then only the replay variant with work is launched), or from the pinned word
  bool list_long = c->long_list_hint;
  if (!c->list_exact && c->h_list_len) list_long = *(volatile uint32_t*)c->h_list_len > uint32_t(kShortReplayMax);
  plan.skip_long = c->list_exact && !list_long;
  int rc = prepare_heavy(c, plan.n_heavy, plan.skip_long);
  if (rc != EG_OK) return rc;
  if (d_stats != nullptr) {      // the statistics epilogue adds to replicated arrays, folded into the packet behind the grids
    static const bool off = [] { const char* e = std::getenv("EIRGRID_STATS_REPLICAS"); return e && e[0] == '0'; }();
    if (!off && n >= 4096u && !c->d_stats_rep) {
      const size_t bytes = sizeof(long long) * size_t(kStatsReplicas) * EG_STATS_LEN;
      EG_HIP(hipMalloc((void**)&c->d_stats_rep, bytes));
      EG_HIP(hipMemsetAsync(c->d_stats_rep, 0, bytes, nullptr));
    }
    // (small batches add directly: a few hundred episodes do not queue up in L2, and the fold is a launch of its own — configs[1],
    //  1 024 episodes: 0.257 ms per batch without it, 0.264 with)
    plan.d_stats_rep = (off || n < 4096u) ? nullptr : c->d_stats_rep;
  }
  // long replay episodes: script / placements / rows, each on its own wave (eg_replay_solo.h) — unless the batch's replays are computed
  // once anyway (what ends that script ends this one as well: the classic variant alone is the fallback then)
  if (plan.n_heavy > 0 && !plan.helper_waves && !plan.skip_long && !c->hoist_on) {
    if (c->solo_on) {
      if (plan.n_heavy > c->solo_cap) {
        if (c->d_solo) (void)hipFree(c->d_solo);
        c->d_solo = nullptr; c->solo_cap = 0;
        EG_HIP(hipMalloc((void**)&c->d_solo, sizeof(unsigned long long) * plan.n_heavy));
        EG_HIP(hipMemsetAsync(c->d_solo, 0, sizeof(unsigned long long) * plan.n_heavy, nullptr));
        c->solo_cap = plan.n_heavy;
      }
      plan.solo_seq = ++c->solo_seq; plan.d_solo = c->d_solo;
    }
  }
  if (c->hoist_on && plan.n_heavy > 0) {      // the replay episodes of this batch are computed once (eg_replay_coop.h)
    plan.hoist_seq = ++c->hoist_seq; plan.d_hoist = c->d_hoist; plan.coop_out = c->d_coop; plan.coop_force = c->coop_force;
    c->hoist_batches += 1;
  }
  if (split) {
    EG_HIP(hipEventRecord(c->ev_fork[slot], nullptr));
    EG_HIP(hipStreamWaitEvent(c->stream_heavy, c->ev_fork[slot], 0));
    // Long replays are the batch's long pole and want the chip first (a 256-register wave that arrives behind 16 384 lean
    // episodes waits for two of them to finish on its SIMD: 5.6 instead of 4.95 ms per batch at 468 generators per replay):
    // the lean grid then waits for an event recorded behind the short-replay variant — which has nothing to do and is gone
    // in microseconds — i.e. until the long variant is being dispatched.  With short replays (or when the host's idea of the
    // list is out of date) nobody waits for anybody.
    // (the hoisted replay is one workgroup that needs a whole CU: it is dispatched ahead of the lean grid in any case)
    if (list_long || plan.hoist_seq != 0ull) plan.go_event = c->ev_go[slot];
  }
  const int lr = launch_rollout(c->dev, c->snap, c->out, seed, first_index, n, d_mask, period, d_stats, plan);
  if (lr != 0) { set_error(std::string("k_rollout launch: ") + hipGetErrorString((hipError_t)lr)); return EG_ERR_HIP; }
  if (split) {
    EG_HIP(hipEventRecord(c->ev_join[slot], c->stream_heavy));
    EG_HIP(hipStreamWaitEvent(nullptr, c->ev_join[slot], 0));
  }
  if (d_stats != nullptr && plan.d_stats_rep != nullptr) {
    const int fs = launch_fold_stats(plan.d_stats_rep, d_stats, nullptr);
    if (fs != 0) { set_error(std::string("k_fold_stats launch: ") + hipGetErrorString((hipError_t)fs)); return EG_ERR_HIP; }
  }
  if (c->fold_mode != 0) {      // behind the batch on the null stream: its results are in iteration order in the records
    const int fr = launch_fold_best(c->out, n, first_index, c->fold_mode == 2, c->d_fold, nullptr);
    if (fr != 0) { set_error(std::string("k_fold_best launch: ") + hipGetErrorString((hipError_t)fr)); return EG_ERR_HIP; }
  }
  c->ev_used[slot] = uint8_t((plan.n_heavy > 0 ? 1 : 0) | (plan.n_lean > 0 ? 2 : 0));
  c->ring_head = (c->ring_head + 1) % eg_ctx::kTimingRing; c->ring_pending += 1;
  c->last_n = n; c->last_first = first_index;
  return EG_OK;
}

}  // namespace

extern "C" {

const char* eg_last_error(void) { return g_error.c_str(); }

int32_t eg_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

eg_ctx* eg_create(int32_t device_ordinal, const eg_world* world) {
  if (!world || world->n_settlements < 0 || world->n_existing < 0 || world->n_coast < 0) { set_error("eg_create: bad world"); return nullptr; }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_error("eg_create: no HIP device (this library has no CPU path)"); return nullptr; }
  if (device_ordinal < 0 || device_ordinal >= n) { set_error("eg_create: device ordinal out of range"); return nullptr; }
  if (hipSetDevice(device_ordinal) != hipSuccess) { set_error("eg_create: hipSetDevice failed"); return nullptr; }
  eg_ctx* c = new eg_ctx();
  {
    // The small-batch kernel holds 3 waves per episode at 3 waves per SIMD: 4 episodes per CU are resident together.
    // EIRGRID_HELPER_WAVES=0 disables it, =all uses it for every batch size (diagnostics / parity tests).
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_ordinal) != hipSuccess) cus = 0;
    c->helper_max_episodes = 4u * (uint32_t)(cus > 0 ? cus : 0);
    // EIRGRID_HEAVY_SLOTS: a fixed number of field slots for heavy episodes (default: 4096 = 516 MB, enlarged to what a launch needs;
    // 0 = every search is the exact scan)
    if (const char* hs = std::getenv("EIRGRID_HEAVY_SLOTS")) { c->heavy_slots_wanted = (uint32_t)std::strtoul(hs, nullptr, 10); c->heavy_slots_auto = false; }
    if (const char* hg = std::getenv("EIRGRID_HEAVY_POOL_GB")) {
      const double gb = std::atof(hg);
      c->heavy_slots_max = uint32_t(std::min(double((1u << 20) - 1u), std::max(0.0, gb) * double(size_t(1) << 30) / double(size_t(kRadiusClasses) * 2624 * sizeof(double))));
    }
    if (c->heavy_slots_max > (1u << 20) - 1u) c->heavy_slots_max = (1u << 20) - 1u;      // (the claim word counts slots in 20 bits)
    if (c->heavy_slots_wanted > c->heavy_slots_max) c->heavy_slots_wanted = c->heavy_slots_max;
    if (const char* hv = std::getenv("EIRGRID_HELPER_WAVES")) {
      if (std::string(hv) == "0") c->helper_max_episodes = 0;
      else if (std::string(hv) == "all") c->helper_max_episodes = 0xFFFFFFFFu;
    }
  }
  c->device = device_ordinal;
  build_tables(*world, c->tables.H);
  c->tables.index();
  const HostTables& H = c->tables.H;
  DevTables& D = c->dev;
  int rc = EG_OK;
  std::vector<uint8_t> blob(tab::total, 0);
  put(blob, tab::usage, H.usage, kYears); put(blob, tab::population, H.population, kYears);
  put(blob, tab::pre_co2, H.pre_co2, kYears); put(blob, tab::pre_tg, H.pre_tg, kYears); put(blob, tab::pre_ig, H.pre_ig, kYears);
  put(blob, tab::pre_sg, H.pre_sg, kYears); put(blob, tab::pre_optot, H.pre_optot, kYears); put(blob, tab::pre_opcnt, H.pre_opcnt, kYears);
  put(blob, tab::inflation, H.inflation, kYears); put(blob, tab::carbon_price, H.carbon_price, kYears);
  put(blob, tab::out_mw, H.out_mw, kTypes); put(blob, tab::co2_t, H.co2_t, kTypes);
  put(blob, tab::cls, H.cls, kTypes); put(blob, tab::rclass, H.rclass, kTypes); put(blob, tab::marine, H.marine, kTypes);
  put(blob, tab::reach, H.reach, kRadiusClasses);
  put(blob, tab::dr, H.dr, size_t(kRadiusClasses) * 169); put(blob, tab::m03, H.m03, kCells); put(blob, tab::t12, H.t12, size_t(kYears) * kTypes);
  put(blob, tab::offv, H.offv, size_t(kYears) * kOffsetTypes * kYears); put(blob, tab::offc, H.offc, size_t(kYears) * kOffsetTypes * kMults);
  put(blob, tab::cc, H.cc, size_t(kYears) * kTypes * kYears * kMults * 2);
  put(blob, tab::te_cell, H.te, size_t(kYears) * kRadiusClasses * kCells); put(blob, tab::coastf, H.coastf, kCells);
  {  // Sorted candidate lists.  final(c) = ((te[c] * prod_g d/R) * cf[c]) * size <= base(c) = (te[c] * cf[c]) * size
     // because every factor is in [0, 1] and IEEE multiplication is monotone, so a scan in descending base order can
     // stop as soon as the next base is below the best final score found (k_rollout / place_search).
    std::vector<std::pair<int, int>> variants;   // (radius class, marine)
    std::vector<int32_t> variant_of(kTypes, 0);
    for (int t = 0; t < kTypes; ++t) {
      std::pair<int, int> key(H.rclass[t], H.marine[t] ? 1 : 0);
      size_t v = 0;
      while (v < variants.size() && variants[v] != key) ++v;
      if (v == variants.size()) variants.push_back(key);
      variant_of[t] = int32_t(v);
    }
    const int NV = int(variants.size());
    if (NV > kMaxVariants) { set_error("eg_create: too many (radius class, marine) variants"); rc = EG_ERR_BAD_ARG; }
    put(blob, tab::variant, variant_of, kTypes);
    D.n_variants = NV;
    PsRec* ps = reinterpret_cast<PsRec*>(blob.data() + tab::ps);   // entries beyond the 2601 candidates stay te = 0
    std::vector<double> base(kCells);
    std::vector<int> order(kCells);
    for (int y = 0; y < kYears && rc == EG_OK; ++y)
      for (int v = 0; v < NV; ++v) {
        const double* te = &H.te[(size_t(y) * kRadiusClasses + variants[v].first) * kCells];
        const bool marine = variants[v].second != 0;
        for (int c2 = 0; c2 < kCells; ++c2) { base[c2] = (te[c2] * (marine ? H.coastf[c2] : 1.0)) * H.size_factor; order[c2] = c2; }
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return base[a] > base[b]; });
        PsRec* list = ps + (size_t(y) * kMaxVariants + v) * kPsStride;
        for (int r = 0; r < kPsStride; ++r) { list[r].te = 0.0; list[r].cf = 1.0; list[r].m03 = 0.0; list[r].cell = 0; list[r].pad = 0; }
        double* pb = reinterpret_cast<double*>(blob.data() + tab::pbase) + (size_t(y) * kMaxVariants + v) * kPcStride;
        uint32_t* pc = reinterpret_cast<uint32_t*>(blob.data() + tab::pcell) + (size_t(y) * kMaxVariants + v) * kPcStride;
        for (int r = 0; r < kPcStride; ++r) { pb[r] = r < kCells ? base[order[r]] : 0.0; pc[r] = r < kCells ? uint32_t(order[r]) : 0u; }
        std::memcpy(blob.data() + tab::cbase + 8 * (size_t(y) * kMaxVariants + v) * kCells, base.data(), 8 * size_t(kCells));      // the same scores per cell
        for (int r = 0; r < kCells; ++r) {
          list[r].te = te[order[r]]; list[r].cf = marine ? H.coastf[order[r]] : 1.0; list[r].m03 = H.m03[order[r]]; list[r].cell = uint32_t(order[r]);
          list[r].pad = uint32_t(4 * (order[r] / kGrid)) | (uint32_t(4 * (order[r] % kGrid)) << 16);
        }
      }
  }
  {  // compact factor table (eg_rollout.hip load_factor_table): class k keeps squared distances 0..cap_k, cap_k = the first at which
     // the factor is 1.0 (d >= R); the factor must depend on the squared distance only and reach 1.0 within 12 cells
    int32_t* meta = reinterpret_cast<int32_t*>(blob.data() + tab::dr_meta);
    int next = 0;
    for (int k = 0; k < kRadiusClasses && rc == EG_OK; ++k) {
      int cap = 1 << 30;
      for (int ai = 0; ai <= kMaxReach; ++ai) for (int aj = 0; aj <= kMaxReach; ++aj)
        if (H.dr[(size_t(k) * 13 + ai) * 13 + aj] == 1.0 && ai * ai + aj * aj < cap) cap = ai * ai + aj * aj;
      bool radial = cap <= kMaxReach * kMaxReach;
      for (int ai = 0; ai <= kMaxReach && radial; ++ai) for (int aj = 0; aj <= kMaxReach; ++aj)
        if ((H.dr[(size_t(k) * 13 + ai) * 13 + aj] == 1.0) != (ai * ai + aj * aj >= cap)) { radial = false; break; }
      if (!radial || cap > 255) { set_error("eg_create: the distance factors of a radius class are not a function of the squared distance that reaches 1.0 within 12 cells"); rc = EG_ERR_BAD_ARG; break; }
      meta[k] = next; meta[8 + k] = cap;
      next += (cap + 1 + 1) & ~1;      // entries 0..cap, every class starts at an even entry
    }
    if (rc == EG_OK && next > kDrCompact) { set_error("eg_create: radii too large for the compact factor table"); rc = EG_ERR_BAD_ARG; }
    if (rc == EG_OK) {      // the table as the kernels hold it in LDS (eg_rollout.hip load_factor_table copies it)
      double* drc = reinterpret_cast<double*>(blob.data() + tab::dr_compact);
      for (int i = 0; i < kDrCompact; ++i) drc[i] = 1.0;
      for (int k = 0; k < kRadiusClasses; ++k)
        for (int ai = 0; ai <= kMaxReach; ++ai) for (int aj = 0; aj <= kMaxReach; ++aj) {
          const int q = ai * ai + aj * aj;
          if (q < meta[8 + k]) drc[meta[k] + q] = H.dr[(size_t(k) * 13 + ai) * 13 + aj];
        }
    }
  }
  {  // heavy episodes (eg_rollout.hip heavy_add): every (class, di, dj) with a factor below 1, i.e. closer than the class radius
    uint32_t* box = reinterpret_cast<uint32_t*>(blob.data() + tab::hv_box);
    int nbox = 0;
    for (int k = 0; k < kRadiusClasses; ++k)
      for (int di = -kMaxReach; di <= kMaxReach; ++di)
        for (int dj = -kMaxReach; dj <= kMaxReach; ++dj) {
          const int ai = di < 0 ? -di : di, aj = dj < 0 ? -dj : dj;
          if (H.dr[(size_t(k) * 13 + ai) * 13 + aj] == 1.0) continue;
          if (nbox < 1024) box[nbox] = uint32_t(di + 16) | (uint32_t(dj + 16) << 5) | (uint32_t(di * di + dj * dj) << 10) | (uint32_t(k) << 19);
          ++nbox;
        }
    if (nbox > 1024) c->heavy_slots_wanted = 0;      // radii the list was not sized for: heavy episodes keep the exact scan
    // (the hoisted replay updates its field with one lane per entry of this list and keeps the scores of eight variants in registers)
    c->hoist_supported = nbox <= 1024 && D.n_variants <= 8;
    for (int i = nbox; i < 1024; ++i) box[i] = 145u << 10;      // padding: class 0, di = dj = -16 (no class reaches that far), q = 145 (factor 1.0)
    for (int k = 0, i = 0; k <= kRadiusClasses; ++k) {      // words 1024..1030: where class k starts (the list is sorted by class), then the end
      while (i < nbox && i < 1024 && int(box[i] >> 19) < k) ++i;
      box[1024 + k] = uint32_t(i);
    }
    // ... and packed for every subset of classes, in the throughput kernel's form (tab::hv_lists): the long-replay variant reads its
    // subset's list from here (a few KB that every long replay of a CU shares) instead of keeping 4 KB of LDS for a copy of its own
    if (rc == EG_OK && nbox <= 1024) {
      const int32_t* meta = reinterpret_cast<const int32_t*>(blob.data() + tab::dr_meta);
      uint32_t* lists = reinterpret_cast<uint32_t*>(blob.data() + tab::hv_lists);
      int32_t* quads = reinterpret_cast<int32_t*>(blob.data() + tab::hv_quads);
      auto place = [&](uint32_t en) -> uint32_t {
        const int k = int(en >> 19), q = int((en >> 10) & 511u);
        return (en & ~(511u << 10)) | (uint32_t(meta[k] + std::min(q, meta[8 + k])) << 10);
      };
      for (int mask = 0; mask < 64; ++mask) {
        uint32_t* l = lists + size_t(mask) * 1024;
        int n = 0;
        for (int k = 0; k < kRadiusClasses; ++k)
          if ((mask >> k) & 1) for (uint32_t i = box[1024 + k]; i < box[1024 + k + 1]; ++i) l[n++] = place(box[i]);
        const int padded = (n + 255) & ~255;
        for (int i = n; i < 1024; ++i) l[i] = place(145u << 10);
        quads[mask] = padded / 256;
      }
    }
  }
  if (rc == EG_OK) {
    void* p = nullptr;
    if (hipMalloc(&p, tab::total) != hipSuccess) { set_error("hipMalloc(tables) failed"); rc = EG_ERR_HIP; }
    else {
      c->allocs.push_back(p);
      if (hipMemcpy(p, blob.data(), tab::total, hipMemcpyHostToDevice) != hipSuccess) { set_error("hipMemcpy(tables) failed"); rc = EG_ERR_HIP; }
      D.base = static_cast<const uint8_t*>(p);
    }
  }
  D.size_factor = H.size_factor; D.n_existing = world->n_existing;
  for (int i = 0; i < eg_ctx::kTimingRing && rc == EG_OK; ++i)
    for (int k = 0; k < 4; ++k)
      if (hipEventCreate(&c->ev[i][k]) != hipSuccess) { set_error("hipEventCreate failed"); rc = EG_ERR_HIP; break; }
  if (rc == EG_OK) {
    // A priority of its own gives the stream a hardware queue of its own.  (A plain stream created after torch / RCCL have made
    // theirs ended up sharing one with the null stream: the grids of a batch then ran one after the other — measured.)
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    const char* sp = std::getenv("EIRGRID_SIDE_STREAM");      // diagnostics: "plain" = no priority
    const bool plain = sp && std::string(sp) == "plain";
    if ((plain ? hipStreamCreateWithFlags(&c->stream_heavy, hipStreamNonBlocking)
               : hipStreamCreateWithPriority(&c->stream_heavy, hipStreamNonBlocking, greatest)) != hipSuccess) { set_error("hipStreamCreate failed"); rc = EG_ERR_HIP; }
  }
  for (int i = 0; i < eg_ctx::kTimingRing && rc == EG_OK; ++i)
    if (hipEventCreateWithFlags(&c->ev_fork[i], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_go[i], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming) != hipSuccess) {
      set_error("hipEventCreate failed"); rc = EG_ERR_HIP;
    }
  if (rc == EG_OK) {
    if (hipMalloc((void**)&c->d_snap, snap::total) != hipSuccess || hipHostMalloc((void**)&c->h_snap, snap::total) != hipSuccess) {
      set_error("hipMalloc(snapshot) failed"); rc = EG_ERR_HIP;
    } else if (hipMemset(c->d_snap, 0, snap::total) != hipSuccess) { set_error("hipMemset(snapshot) failed"); rc = EG_ERR_HIP; }
  }
  if (rc == EG_OK) {      // (a convenience, not a requirement: without it the hint stays what the host last knew)
    if (hipHostMalloc((void**)&c->h_list_len, 64, hipHostMallocMapped) == hipSuccess) {
      *c->h_list_len = 0u;
      if (hipHostGetDevicePointer((void**)&c->d_list_len, c->h_list_len, 0) != hipSuccess) { (void)hipGetLastError(); c->d_list_len = nullptr; }
    } else { (void)hipGetLastError(); c->h_list_len = nullptr; }
  }
  if (rc == EG_OK) {
    if (hipMalloc((void**)&c->d_hoist, kHoistBytes) != hipSuccess || hipMalloc((void**)&c->d_coop, rec::stride + 64) != hipSuccess ||
        hipMemset(c->d_hoist, 0, kHoistBytes) != hipSuccess || hipMemset(c->d_coop, 0, rec::stride + 64) != hipSuccess) {
      set_error("hipMalloc(replay hoist) failed"); rc = EG_ERR_HIP;
    }
    if (const char* rh = std::getenv("EIRGRID_REPLAY_HOIST")) c->hoist_on = c->hoist_supported && rh[0] == '1';
    if (const char* cf = std::getenv("EIRGRID_COOP_FORCE")) c->coop_force = std::atoi(cf);
    if (const char* so = std::getenv("EIRGRID_REPLAY_SOLO")) c->solo_on = so[0] != '0';
  }
  if (rc != EG_OK) { eg_destroy(c); return nullptr; }
  return c;
}

void eg_destroy(eg_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  for (void* p : c->allocs) (void)hipFree(p);
  if (c->d_snap) (void)hipFree(c->d_snap);
  if (c->d_snap_held) (void)hipFree(c->d_snap_held);
  if (c->d_fold) (void)hipFree(c->d_fold);
  if (c->d_hoist) (void)hipFree(c->d_hoist);
  if (c->d_stats_rep) (void)hipFree(c->d_stats_rep);
  if (c->d_solo) (void)hipFree(c->d_solo);
  if (c->d_coop) (void)hipFree(c->d_coop);
  if (c->h_snap) (void)hipHostFree(c->h_snap);
  if (c->h_list_len) (void)hipHostFree(c->h_list_len);
  if (c->d_mask) (void)hipFree(c->d_mask);
  if (c->d_packet) (void)hipFree(c->d_packet);
  if (c->h_packet) (void)hipHostFree(c->h_packet);
  free_outputs(c);
  for (int i = 0; i < eg_ctx::kTimingRing; ++i)
    for (int k = 0; k < 4; ++k) if (c->ev[i][k]) (void)hipEventDestroy(c->ev[i][k]);
  if (c->stream_heavy) (void)hipStreamDestroy(c->stream_heavy);
  for (int i = 0; i < eg_ctx::kTimingRing; ++i) { if (c->ev_fork[i]) (void)hipEventDestroy(c->ev_fork[i]); if (c->ev_go[i]) (void)hipEventDestroy(c->ev_go[i]); if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]); }
  if (c->d_index) (void)hipFree(c->d_index);
  if (c->d_place_cells) (void)hipFree(c->d_place_cells);
  if (c->d_place_cell) (void)hipFree(c->d_place_cell);
  if (c->d_place_score) (void)hipFree(c->d_place_score);
  if (c->d_place_xy) (void)hipFree(c->d_place_xy);
  delete c;
}

eg_host_tables* eg_host_tables_create(const eg_world* world) {
  if (!world || world->n_settlements < 0 || world->n_existing < 0 || world->n_coast < 0) { set_error("eg_host_tables_create: bad world"); return nullptr; }
  eg_host_tables* h = new eg_host_tables();
  build_tables(*world, h->H);
  h->index();
  return h;
}
void eg_host_tables_free(eg_host_tables* h) { delete h; }
int32_t eg_host_tables_f64(const eg_host_tables* h, const char* name, const double** ptr, int64_t* len) {
  if (!h || !name || !ptr || !len) return EG_ERR_BAD_ARG;
  auto it = h->f64.find(name);
  if (it == h->f64.end()) { set_error(std::string("eg_host_tables_f64: unknown table ") + name); return EG_ERR_BAD_ARG; }
  *ptr = it->second.first; *len = it->second.second;
  return EG_OK;
}
int32_t eg_host_tables_i32(const eg_host_tables* h, const char* name, const int32_t** ptr, int64_t* len) {
  if (!h || !name || !ptr || !len) return EG_ERR_BAD_ARG;
  auto it = h->i32.find(name);
  if (it == h->i32.end()) { set_error(std::string("eg_host_tables_i32: unknown table ") + name); return EG_ERR_BAD_ARG; }
  *ptr = it->second.first; *len = it->second.second;
  return EG_OK;
}

int32_t eg_upload_snapshot(eg_ctx* c, const eg_policy_snapshot* s, const eg_opts* o) {
  if (!c || !s || !s->weights || !s->deficit_weights) { set_error("eg_upload_snapshot: bad argument"); return EG_ERR_BAD_ARG; }
  if (o && o->enable_construction_delays) { set_error("enable_construction_delays is not implemented on the device (SURVEY §8(f) N4)"); return EG_ERR_UNSUPPORTED; }
  // the device walks rely on strictly positive weights (the reference clamps every weight to [1e-4, 0.999])
  for (int i = 0; i < EG_YEARS * EG_N_ACTIONS; ++i) if (!(s->weights[i] > 0.0)) { set_error("eg_upload_snapshot: weights must be > 0"); return EG_ERR_BAD_ARG; }
  for (int i = 0; i < EG_YEARS * EG_N_DEFICIT; ++i) if (!(s->deficit_weights[i] > 0.0)) { set_error("eg_upload_snapshot: deficit weights must be > 0"); return EG_ERR_BAD_ARG; }
  EG_HIP(hipSetDevice(c->device));
  const bool have_lists = s->has_best && s->best_count && s->best_actions && s->best_deficit_count && s->best_deficit_actions;
  int32_t off[28] = {0}, offd[28] = {0};
  if (have_lists) for (int y = 0; y < EG_YEARS; ++y) { off[y + 1] = off[y] + s->best_count[y]; offd[y + 1] = offd[y] + s->best_deficit_count[y]; }
  if (size_t(off[26]) > snap::kBestCap || size_t(offd[26]) > snap::kBestCap) { set_error("eg_upload_snapshot: best action lists too long"); return EG_ERR_BAD_ARG; }
  EG_HIP(hipStreamSynchronize(nullptr));   // the pinned staging buffer may still feed the previous copy
  uint8_t* h = c->h_snap;
  {  // packed policy rows; sampling.rs:182, :352-355, :406: the sums the samplers start from, folded in table order
    double* pol = reinterpret_cast<double*>(h + snap::pol);
    std::memset(pol, 0, sizeof(double) * EG_YEARS * snap::kPolRow);
    for (int y = 0; y < EG_YEARS; ++y) {
      double* row = pol + y * snap::kPolRow;
      double a = 0.0, b = 0.0, c2 = 0.0;
      for (int i = 0; i < EG_N_ACTIONS; ++i) { row[i] = s->weights[y * EG_N_ACTIONS + i]; a += row[i]; }
      for (int i = 0; i < EG_N_DEFICIT; ++i) row[snap::kPolDw + i] = s->deficit_weights[y * EG_N_DEFICIT + i];
      for (int i = 0; i < 14; ++i) b += s->deficit_weights[y * EG_N_DEFICIT + i];
      if (s->count_weights) for (int i = 0; i < EG_N_COUNTS; ++i) { row[snap::kPolCw + i] = s->count_weights[y * EG_N_COUNTS + i]; c2 += row[snap::kPolCw + i]; }
      row[snap::kPolTotMain] = a; row[snap::kPolTotDeficit] = b; row[snap::kPolTotCount] = c2;
      const HostTables& H = c->tables.H;      // the year's world scalars ride along (eg_internal.h, snap::kPolYear)
      double* ys = row + snap::kPolYear;
      ys[0] = H.pre_co2[y]; ys[1] = H.pre_tg[y]; ys[2] = H.pre_ig[y]; ys[3] = H.pre_sg[y]; ys[4] = H.pre_optot[y];
      ys[5] = H.usage[y]; ys[6] = H.population[y]; ys[7] = H.inflation[y]; ys[8] = H.carbon_price[y]; ys[9] = double(H.pre_opcnt[y]);
    }
  }
  unsigned long long mask[26] = {0}, dmask[26] = {0};
  if (have_lists)
    for (int y = 0; y < EG_YEARS; ++y) {
      for (int i = off[y]; i < off[y + 1]; ++i) if (s->best_actions[i] < 64) mask[y] |= 1ull << s->best_actions[i];
      for (int i = offd[y]; i < offd[y + 1]; ++i) if (s->best_deficit_actions[i] < 64) { mask[y] |= 1ull << s->best_deficit_actions[i]; dmask[y] |= 1ull << s->best_deficit_actions[i]; }
    }
  std::memcpy(h + snap::best_mask, mask, sizeof(mask)); std::memcpy(h + snap::bestd_mask, dmask, sizeof(dmask));
  std::memcpy(h + snap::best_off, off, sizeof(off)); std::memcpy(h + snap::bestd_off, offd, sizeof(offd));
  c->long_list_hint = have_lists && off[26] > kShortReplayMax; c->list_exact = true;
  // (the pinned word is a HINT only — read when list_exact is false, to order the launches: an update or rewind kernel enqueued before
  //  this upload may still store an older length over this one; nothing but the launch order ever depends on it)
  if (c->h_list_len) *(volatile uint32_t*)c->h_list_len = have_lists ? uint32_t(off[26]) : 0u;
  if (have_lists) { std::memcpy(h + snap::best_actions, s->best_actions, size_t(off[26])); std::memcpy(h + snap::bestd_actions, s->best_deficit_actions, size_t(offd[26])); }
  {  // the policy's scalars as the kernels read them (snap::state)
    DevState st{};
    st.learning_rate = s->learning_rate; st.exploration_rate = s->exploration_rate;
    for (int i = 0; i < 4; ++i) st.best_metrics[i] = s->has_best ? s->best_metrics[i] : 0.0;
    st.stall = s->iterations_without_improvement; st.iteration_count = c->push_iteration_count; st.failed_total = c->push_failed;
    st.has_best = s->has_best ? 1 : 0; st.has_cw = s->count_weights ? 1 : 0; st.has_lists = have_lists ? 1 : 0;
    rm::derive_state(st);
    std::memcpy(h + snap::state, &st, sizeof(st));
  }
  EG_HIP(hipMemcpyAsync(c->d_snap, h, snap::upload_bytes, hipMemcpyHostToDevice, nullptr));   // stream-ordered before the next launch
  {
    int lr = launch_stalled_tables(c->d_snap, nullptr);      // sampling.rs:190-220 on the un-nudged rows (no-op unless stalled)
    if (lr != 0) { set_error(std::string("k_stalled_tables launch: ") + hipGetErrorString((hipError_t)lr)); return EG_ERR_HIP; }
  }
  DevSnapshot& S = c->snap;
  S = DevSnapshot{};
  S.base = c->d_snap;
  S.enable_energy_sales = o ? (o->enable_energy_sales ? 1 : 0) : 1;
  S.write_yearly = o ? (o->write_yearly ? 1 : 0) : 1;
  c->snap_valid = true;
  return EG_OK;
}

int32_t eg_rollout_launch(eg_ctx* c, uint64_t seed, uint64_t first_index, uint32_t n, const uint8_t* replay_mask) {
  if (!c || !c->snap_valid) { set_error("eg_rollout_launch: upload a snapshot first"); return EG_ERR_BAD_ARG; }
  if (n == 0) { c->last_n = 0; return EG_OK; }
  EG_HIP(hipSetDevice(c->device));
  int rc = ensure_outputs(c, n);
  if (rc != EG_OK) return rc;
  return launch_batch(c, seed, first_index, n, replay_mask, 0u, nullptr);
}

int32_t eg_rollout_launch_update(eg_ctx* c, uint64_t seed, uint64_t first_index, uint32_t n, const uint8_t* replay_mask, void* d_packet) {
  if (!c || !c->snap_valid || !d_packet) { set_error("eg_rollout_launch_update: bad argument"); return EG_ERR_BAD_ARG; }
  EG_HIP(hipSetDevice(c->device));
  int rc = ensure_outputs(c, n ? n : 1);
  if (rc != EG_OK) return rc;
  EG_HIP(hipMemsetAsync(d_packet, 0, EG_PACKET_BYTES, nullptr));
  c->last_n = n; c->last_first = first_index;
  rc = launch_batch(c, seed, first_index, n, replay_mask, 0u, (long long*)d_packet);
  if (rc != EG_OK) return rc;
  int lr = launch_pick_best(c->out, n, first_index, reinterpret_cast<UpdateCandidate*>(static_cast<uint8_t*>(d_packet) + 8 * EG_STATS_LEN), nullptr);
  if (lr != 0) { set_error(std::string("k_pick_best launch: ") + hipGetErrorString((hipError_t)lr)); return EG_ERR_HIP; }
  return EG_OK;
}

int32_t eg_debug_fill_lds(eg_ctx* c, uint32_t value) {
  if (!c) { set_error("eg_debug_fill_lds: bad argument"); return EG_ERR_BAD_ARG; }
  EG_HIP(hipSetDevice(c->device));
  int rc = ensure_outputs(c, 1);
  if (rc != EG_OK) return rc;
  hipDeviceProp_t prop;
  EG_HIP(hipGetDeviceProperties(&prop, c->device));
  // 64 KB per workgroup: at most two share a CU's 160 KB, so 4 per CU in flight-order covers every slot several times
  int lr = launch_fill_lds(value, reinterpret_cast<uint32_t*>(c->out.base), prop.multiProcessorCount * 8, nullptr);
  if (lr != 0) { set_error(std::string("k_fill_lds launch: ") + hipGetErrorString((hipError_t)lr)); return EG_ERR_HIP; }
  EG_HIP(hipDeviceSynchronize());
  return EG_OK;
}

int32_t eg_debug_occupy(eg_ctx* c, int32_t variant, uint64_t cycles) {
  if (!c) return EG_ERR_BAD_ARG;
  EG_HIP(hipSetDevice(c->device));
  int rc = ensure_outputs(c, 1);
  if (rc != EG_OK) return rc;
  int lr = launch_occupy(variant, cycles, reinterpret_cast<uint32_t*>(c->out.score_list), c->stream_heavy);      // the library's side stream
  if (lr != 0) { set_error(std::string("k_occupy launch: ") + hipGetErrorString((hipError_t)lr)); return EG_ERR_HIP; }
  return EG_OK;
}

int32_t eg_sync(eg_ctx* c) {
  if (!c) return EG_ERR_BAD_ARG;
  EG_HIP(hipSetDevice(c->device));
  EG_HIP(hipDeviceSynchronize());
  return collect_timing(c);
}

namespace {
// One strided copy per requested field: episode records are rec::stride bytes apart on the device.  The lists of a record have the
// oracle's capacity (4 096 entries: 41.6 KB per episode), an episode fills a fraction of it (a sampled one about 1 KB): the counts
// come first, and every list is then copied only as wide as the longest of the batch needs — the caller's rows keep their full
// pitch, what lies behind an episode's entries is left as the caller passed it — except for the single-record fetches (N == 1:
// eg_fetch_record, eg_fetch_best_run, eg_fetch_best_result), whose rows are zeroed behind the entries: a C caller with an
// uninitialised buffer gets a defined row there, and it costs nothing.  (EIRGRID_FETCH_FULL=1: whole rows, for the
// diagnostic builds that park their cycle stamps at the end of act_log.)
int fetch_records(const uint8_t* d_base, size_t N, eg_episode_out* o) {
#define EG_GET_W(field, count, type, used) \
  if (o->field && (used) > 0) EG_HIP(hipMemcpy2D(o->field, (count) * sizeof(type), d_base + rec::field, rec::stride, (used) * sizeof(type), N, hipMemcpyDeviceToHost)); \
  if (o->field && N == 1 && size_t(used) < size_t(count)) std::memset(o->field + (used), 0, (size_t(count) - size_t(used)) * sizeof(type))      /* one record: its rows end in zeros */
#define EG_GET(field, count, type) EG_GET_W(field, count, type, count)
  EG_GET(metrics, 4, double); EG_GET(yearly, EG_YEARS * EG_YEARLY_FIELDS, double); EG_GET(status, 1, int32_t);
  EG_GET(n_gens, 1, int32_t); EG_GET(n_offsets, 1, int32_t);
  EG_GET(bytes_moved, 1, double);
  EG_GET(n_draws, 1, uint64_t);
  EG_GET(n_chunks, 1, uint32_t);
  static const bool full = [] { const char* f = std::getenv("EIRGRID_FETCH_FULL"); return f && f[0] == '1'; }();
  size_t run = EG_RUN_CAP, def = EG_DEF_CAP, act = EG_ACT_CAP, gens = EG_MAX_GENS, offs = EG_MAX_OFFSETS;
  const bool lists = o->run_log || o->def_log || o->act_log || o->gen_cell || o->gen_pack || o->off_pack;
  std::vector<int32_t> cnt;      // n_run | n_def | n_act [26] each, n_gens, n_offsets: the header of a record, contiguous from rec::status on
  if (lists && !full) {
    constexpr size_t kHead = rec::yearly - rec::status;      // status, n_gens, n_offsets, n_chunks, n_run, n_def, n_act
    static_assert(rec::n_gens == rec::status + 4 && rec::n_offsets == rec::status + 8 && rec::n_run == rec::status + 16, "record header");
    cnt.resize(N * (kHead / 4));
    EG_HIP(hipMemcpy2D(cnt.data(), kHead, d_base + rec::status, rec::stride, kHead, N, hipMemcpyDeviceToHost));
    run = def = act = gens = offs = 0;
    for (size_t e = 0; e < N; ++e) {
      const int32_t* h = cnt.data() + e * (kHead / 4);
      size_t r = 0, d = 0, a = 0;
      for (int y = 0; y < EG_YEARS; ++y) { r += size_t(std::max(h[4 + y], 0)); d += size_t(std::max(h[4 + EG_YEARS + y], 0)); a += size_t(std::max(h[4 + 2 * EG_YEARS + y], 0)); }
      run = std::max(run, r); def = std::max(def, d); act = std::max(act, a);
      gens = std::max(gens, size_t(std::max(h[1], 0))); offs = std::max(offs, size_t(std::max(h[2], 0)));
    }
    run = std::min(run, size_t(EG_RUN_CAP)); def = std::min(def, size_t(EG_DEF_CAP)); act = std::min(act, size_t(EG_ACT_CAP));
    gens = std::min(gens, size_t(EG_MAX_GENS)); offs = std::min(offs, size_t(EG_MAX_OFFSETS));
  }
  EG_GET(n_run, EG_YEARS, int32_t); EG_GET(n_def, EG_YEARS, int32_t); EG_GET(n_act, EG_YEARS, int32_t);
  EG_GET_W(run_log, EG_RUN_CAP, uint8_t, run); EG_GET_W(def_log, EG_DEF_CAP, uint8_t, def); EG_GET_W(act_log, EG_ACT_CAP, uint8_t, act);
  EG_GET_W(gen_cell, EG_MAX_GENS, uint16_t, gens); EG_GET_W(gen_pack, EG_MAX_GENS, uint16_t, gens);
  EG_GET_W(off_pack, EG_MAX_OFFSETS, uint16_t, offs);
#undef EG_GET
#undef EG_GET_W
  return EG_OK;
}
}  // namespace

namespace {
// an episode that ended with EG_EP_INTERNAL is a defect of the kernel's helper-wave protocol, not a property of the input
int check_internal(const int32_t* status, size_t n) {
  if (!status) return EG_OK;
  for (size_t i = 0; i < n; ++i)
    if (status[i] == EG_EP_INTERNAL) { set_error("k_rollout: helper-wave protocol timed out in episode " + std::to_string(i) + " (EG_EP_INTERNAL)"); return EG_ERR_INTERNAL; }
  return EG_OK;
}
}  // namespace

uint32_t eg_last_batch_size(const eg_ctx* c) { return c ? c->last_n : 0u; }

int32_t eg_fetch(eg_ctx* c, eg_episode_out* o) {
  if (!c || !o) return EG_ERR_BAD_ARG;
  int rc = eg_sync(c);
  if (rc != EG_OK) return rc;
  if (c->last_n == 0) return EG_OK;
  rc = fetch_records(c->out.base, c->last_n, o);
  return rc != EG_OK ? rc : check_internal(o->status, c->last_n);
}

int32_t eg_fetch_record(eg_ctx* c, uint32_t episode, eg_episode_out* o) {
  if (!c || !o || episode >= c->last_n) { set_error("eg_fetch_record: bad argument"); return EG_ERR_BAD_ARG; }
  int rc = eg_sync(c);
  if (rc != EG_OK) return rc;
  rc = fetch_records(c->out.base + size_t(episode) * rec::stride, 1, o);
  return rc != EG_OK ? rc : check_internal(o->status, 1);
}

int32_t eg_fetch_best_run(eg_ctx* c, eg_episode_out* o, int32_t* state) {
  if (!c || !o || !state || !c->snap_valid) { set_error("eg_fetch_best_run: push a policy first"); return EG_ERR_BAD_ARG; }
  int rc = eg_sync(c);
  if (rc != EG_OK) return rc;
  uint32_t word = 0;
  EG_HIP(hipMemcpy(&word, c->d_snap + snap::best_rec_state, sizeof(word), hipMemcpyDeviceToHost));
  *state = (int32_t)word;
  if (word != 1u) return EG_OK;
  return fetch_records(c->d_snap + snap::best_rec, 1, o);
}

int32_t eg_memory_report(const eg_ctx* c, uint64_t* table_bytes, uint64_t* record_bytes, uint64_t* field_pool_bytes) {
  if (!c) return EG_ERR_BAD_ARG;
  if (table_bytes) *table_bytes = uint64_t(tab::total);
  if (record_bytes) *record_bytes = uint64_t(c->out_cap) * (rec::stride + sizeof(double));
  if (field_pool_bytes) *field_pool_bytes = uint64_t(c->dev.heavy ? c->dev.heavy_slots : 0u) * uint64_t(kRadiusClasses) * 2624u * sizeof(double);
  return EG_OK;
}

int32_t eg_best_result_track(eg_ctx* c, int32_t mode) {
  if (!c || mode < 0 || mode > 2) { set_error("eg_best_result_track: bad argument"); return EG_ERR_BAD_ARG; }
  EG_HIP(hipSetDevice(c->device));
  if (mode != 0) {
    if (!c->d_fold) EG_HIP(hipMalloc((void**)&c->d_fold, kFoldBytes));
    EG_HIP(hipMemsetAsync(c->d_fold, 0, kFoldBytes, nullptr));      // best_result = None (multi_simulation.rs:384)
  }
  c->fold_mode = mode;
  return EG_OK;
}

int32_t eg_fetch_best_result(eg_ctx* c, eg_episode_out* o, int32_t* state, int64_t* global_index) {
  if (!c || !o || !state) { set_error("eg_fetch_best_result: bad argument"); return EG_ERR_BAD_ARG; }
  if (!c->d_fold) { set_error("eg_fetch_best_result: eg_best_result_track first"); return EG_ERR_BAD_ARG; }
  int rc = eg_sync(c);
  if (rc != EG_OK) return rc;
  FoldState st{};
  EG_HIP(hipMemcpy(&st, c->d_fold, sizeof(st), hipMemcpyDeviceToHost));
  *state = st.has ? 1 : 0;
  if (global_index) *global_index = st.has ? int64_t(st.index) : -1;
  if (!st.has) return EG_OK;
  return fetch_records(c->d_fold + kFoldRecord, 1, o);
}

namespace { int ensure_packet(eg_ctx* c); }
int32_t eg_train_step(eg_ctx* c, eg_policy* p, const eg_opts* o, uint64_t seed, uint64_t first_index, uint32_t n,
                      const uint8_t* replay_mask, uint64_t noise_seed) {
  if (!c || !p) { set_error("eg_train_step: bad argument"); return EG_ERR_BAD_ARG; }
  EG_HIP(hipSetDevice(c->device));
  int prc = ensure_packet(c);
  if (prc != EG_OK) return prc;
  eg_policy_snapshot snap;
  int rc = eg_policy_snapshot_view(p, &snap);
  if (rc != EG_OK) return rc;
  rc = eg_upload_snapshot(c, &snap, o);
  if (rc != EG_OK) return rc;
  rc = eg_rollout_launch_update(c, seed, first_index, n, replay_mask, c->d_packet);
  if (rc != EG_OK) return rc;
  EG_HIP(hipMemcpyAsync(c->h_packet, c->d_packet, EG_PACKET_BYTES, hipMemcpyDeviceToHost, nullptr));
  EG_HIP(hipStreamSynchronize(nullptr));
  return eg_policy_apply_packet(p, reinterpret_cast<const int64_t*>(c->h_packet), c->h_packet + 8 * EG_STATS_LEN, 1, noise_seed);
}

// ---- device-resident policy: push once, step without host synchronisation, pull when needed -----------------------
namespace {
int ensure_packet(eg_ctx* c) {
  if (c->d_packet) return EG_OK;
  EG_HIP(hipMalloc((void**)&c->d_packet, EG_PACKET_BYTES));
  EG_HIP(hipMemset(c->d_packet, 0, EG_PACKET_BYTES));      // the rollout epilogue ADDS to the statistics
  EG_HIP(hipHostMalloc((void**)&c->h_packet, EG_PACKET_BYTES));
  return EG_OK;
}
}  // namespace

int32_t eg_policy_push(eg_ctx* c, const eg_policy* p, const eg_opts* o) {
  if (!c || !p) { set_error("eg_policy_push: bad argument"); return EG_ERR_BAD_ARG; }
  eg_policy_snapshot snap;
  int rc = eg_policy_snapshot_view(p, &snap);
  if (rc != EG_OK) return rc;
  c->push_iteration_count = p->iteration_count; c->push_failed = p->failed_episodes;
  rc = eg_upload_snapshot(c, &snap, o);
  c->push_iteration_count = 0; c->push_failed = 0;
  if (rc != EG_OK) return rc;
  c->pulled_improvements = 0;
  rc = ensure_packet(c);
  if (rc != EG_OK) return rc;
  EG_HIP(hipMemsetAsync(c->d_packet, 0, EG_PACKET_BYTES, nullptr));
  {  // the kept record of the best episode (eg_fetch_best_run) survives a push only when it is the pushed policy's best
     // strategy (checkpoint / resume on the same context); a record left by another policy is dropped
    uint32_t word = 0; double m[4] = {0, 0, 0, 0};
    EG_HIP(hipMemcpy(&word, c->d_snap + snap::best_rec_state, sizeof(word), hipMemcpyDeviceToHost));
    if (word == 1u) EG_HIP(hipMemcpy(m, c->d_snap + snap::best_rec + rec::metrics, sizeof(m), hipMemcpyDeviceToHost));
    const bool same = word == 1u && p->has_best && std::memcmp(m, p->best_metrics.data(), sizeof(m)) == 0;
    if (word != 0u && !same) EG_HIP(hipMemsetAsync(c->d_snap + snap::best_rec_state, 0, sizeof(uint32_t), nullptr));
  }
  // main weights at the last improvement travel with the policy
  if (p->has_best_weights) EG_HIP(hipMemcpyAsync(c->d_snap + snap::best_w, p->best_w.data(), sizeof(double) * EG_YEARS * EG_N_ACTIONS, hipMemcpyHostToDevice, nullptr));
  EG_HIP(hipStreamSynchronize(nullptr));      // p->best_w is pageable host memory
  return EG_OK;
}

namespace {
int device_rollout(eg_ctx* c, uint64_t seed, uint64_t first_index, uint32_t n, uint32_t replay_period, void* d_packet, bool pick);
int device_apply(eg_ctx* c, const void* d_packets, int32_t n_packets, void* d_own_packet, uint64_t noise_seed, bool local_pick);
}  // namespace

int32_t eg_device_rollout(eg_ctx* c, uint64_t seed, uint64_t first_index, uint32_t n, uint32_t replay_period, void* d_packet) {
  return device_rollout(c, seed, first_index, n, replay_period, d_packet, true);
}
int32_t eg_device_apply(eg_ctx* c, const void* d_packets, int32_t n_packets, void* d_own_packet, uint64_t noise_seed) {
  return device_apply(c, d_packets, n_packets, d_own_packet, noise_seed, false);
}
// One GPU: the best episode is found inside k_apply_update (from the best-score key the rollout epilogue leaves in the
// statistics), so a step is three launches: k_rollout, k_apply_update, k_stalled_tables.
int32_t eg_device_step(eg_ctx* c, uint64_t seed, uint64_t first_index, uint32_t n, uint32_t replay_period, uint64_t noise_seed) {
  if (!c) return EG_ERR_BAD_ARG;
  if (n == 0) return EG_OK;
  int rc = ensure_packet(c);
  if (rc != EG_OK) return rc;
  rc = device_rollout(c, seed, first_index, n, replay_period, c->d_packet, false);
  if (rc != EG_OK) return rc;
  return device_apply(c, c->d_packet, 1, c->d_packet, noise_seed, true);
}

namespace {
int device_rollout(eg_ctx* c, uint64_t seed, uint64_t first_index, uint32_t n, uint32_t replay_period, void* d_packet, bool pick) {
  if (!c || !c->snap_valid || !d_packet) { set_error("eg_device_rollout: push a policy first"); return EG_ERR_BAD_ARG; }
  if (n == 0) { c->last_n = 0; return EG_OK; }
  EG_HIP(hipSetDevice(c->device));
  int rc = ensure_outputs(c, n);
  if (rc != EG_OK) return rc;
  rc = launch_batch(c, seed, first_index, n, nullptr, replay_period, (long long*)d_packet);
  if (rc != EG_OK) return rc;
  if (!pick) return EG_OK;
  int lr = launch_pick_best(c->out, n, first_index, reinterpret_cast<UpdateCandidate*>(static_cast<uint8_t*>(d_packet) + 8 * EG_STATS_LEN), nullptr);
  if (lr != 0) { set_error(std::string("k_pick_best launch: ") + hipGetErrorString((hipError_t)lr)); return EG_ERR_HIP; }
  return EG_OK;
}

int device_apply(eg_ctx* c, const void* d_packets, int32_t n_packets, void* d_own_packet, uint64_t noise_seed, bool local_pick) {
  if (!c || !c->snap_valid || !d_packets || n_packets < 1 || !d_own_packet) { set_error("eg_device_apply: bad argument"); return EG_ERR_BAD_ARG; }
  EG_HIP(hipSetDevice(c->device));
  c->list_exact = false;      // from here on the device may hold another best list than the host thinks
  int lr = launch_apply_update(c->d_snap, d_packets, n_packets, (long long*)d_own_packet, noise_seed, c->out, c->last_n, c->last_first,
                               local_pick && n_packets == 1 && c->last_n > 0, c->d_list_len, nullptr);
  if (lr != 0) { set_error(std::string("k_apply_update launch: ") + hipGetErrorString((hipError_t)lr)); return EG_ERR_HIP; }
  // (the stalled sampler's tables of the updated rows are rebuilt inside k_apply_update)
  return EG_OK;
}
}  // namespace

int32_t eg_replay_hoist(eg_ctx* c, int32_t on) {
  if (!c) { set_error("eg_replay_hoist: bad argument"); return EG_ERR_BAD_ARG; }
  if (on && !c->hoist_supported) { set_error("eg_replay_hoist: this world's penalty radii / type variants exceed what the hoisted replay is sized for"); return EG_ERR_UNSUPPORTED; }
  c->hoist_on = on != 0;
  return EG_OK;
}

int32_t eg_replay_hoist_stats(eg_ctx* c, uint64_t* batches_armed, int32_t* last_batch_hoisted) {
  if (!c) return EG_ERR_BAD_ARG;
  EG_HIP(hipSetDevice(c->device));
  EG_HIP(hipDeviceSynchronize());
  unsigned long long word = 0;
  EG_HIP(hipMemcpy(&word, c->d_hoist, sizeof(word), hipMemcpyDeviceToHost));
  if (batches_armed) *batches_armed = c->hoist_batches;
  static_assert(offsetof(HoistInfo, served_seq) == 0, "the served word comes first");
  if (last_batch_hoisted) *last_batch_hoisted = (c->hoist_seq != 0ull && word == c->hoist_seq) ? 1 : 0;
  return EG_OK;
}

int32_t eg_debug_hoist_stamps(eg_ctx* c, uint64_t stamps[8]) {
  if (!c || !stamps) return EG_ERR_BAD_ARG;
  EG_HIP(hipSetDevice(c->device));
  EG_HIP(hipDeviceSynchronize());
  EG_HIP(hipMemcpy(stamps, reinterpret_cast<const uint8_t*>(c->d_hoist) + offsetof(HoistInfo, stamps), 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return EG_OK;
}

int32_t eg_policy_hold(eg_ctx* c) {
  if (!c || !c->snap_valid) { set_error("eg_policy_hold: push a policy first"); return EG_ERR_BAD_ARG; }
  EG_HIP(hipSetDevice(c->device));
  if (!c->d_snap_held) EG_HIP(hipMalloc((void**)&c->d_snap_held, snap::total));
  EG_HIP(hipMemcpyAsync(c->d_snap_held, c->d_snap, snap::total, hipMemcpyDeviceToDevice, nullptr));
  // What the host knows about the list it is holding travels with the copy: a hold behind on-device updates that nobody has pulled
  // (list_exact == false) must not come back from a rewind as "known to be short" — the long-replay variant would not be launched
  // and the replay episodes' records would keep the previous batch's bytes.
  c->long_list_hint_held = c->long_list_hint; c->list_exact_held = c->list_exact;
  return EG_OK;
}

int32_t eg_policy_rewind(eg_ctx* c) {
  if (!c || !c->d_snap_held) { set_error("eg_policy_rewind: nothing held"); return EG_ERR_BAD_ARG; }
  EG_HIP(hipSetDevice(c->device));
  // (the count of failed episodes is a diagnostic of the run, not policy: it goes on counting; one small kernel instead of two copies)
  const int lr = launch_rewind(c->d_snap, c->d_snap_held, c->d_list_len, nullptr);
  if (lr != 0) { set_error(std::string("k_rewind launch: ") + hipGetErrorString((hipError_t)lr)); return EG_ERR_HIP; }
  // the next launch finds the held policy's list: the host knows it exactly when it knew it at the hold; otherwise both replay
  // variants are launched and decide on the device (k_rewind publishes the held list's length through the pinned word for the order)
  c->long_list_hint = c->long_list_hint_held; c->list_exact = c->list_exact_held;
  return EG_OK;
}

int32_t eg_policy_pull(eg_ctx* c, eg_policy* p) {
  if (!c || !p || !c->snap_valid) { set_error("eg_policy_pull: push a policy first"); return EG_ERR_BAD_ARG; }
  EG_HIP(hipSetDevice(c->device));
  EG_HIP(hipStreamSynchronize(nullptr));
  std::vector<uint8_t> h(snap::total);
  EG_HIP(hipMemcpy(h.data(), c->d_snap, snap::total, hipMemcpyDeviceToHost));
  const double* pol = reinterpret_cast<const double*>(h.data() + snap::pol);
  DevState st; std::memcpy(&st, h.data() + snap::state, sizeof(st));
  for (int y = 0; y < EG_YEARS; ++y) {
    const double* row = pol + y * snap::kPolRow;
    for (int a = 0; a < EG_N_ACTIONS; ++a) p->w[y][a] = row[a];
    for (int i = 0; i < EG_N_DEFICIT; ++i) p->dw[y][i] = row[snap::kPolDw + i];
  }
  p->stall = st.stall; p->iteration_count = st.iteration_count; p->failed_episodes = st.failed_total;
  c->long_list_hint = st.has_lists && reinterpret_cast<const int32_t*>(h.data() + snap::best_off)[EG_YEARS] > kShortReplayMax;
  c->list_exact = true;      // (the stream was drained above: nothing is in flight that could change it)
  if (st.n_improvements > 0) {      // at least one on-device improvement since the push: the best strategy is the device's
    p->has_best = true; for (int i = 0; i < 4; ++i) p->best_metrics[i] = st.best_metrics[i];
    const int32_t* off = reinterpret_cast<const int32_t*>(h.data() + snap::best_off);
    const int32_t* offd = reinterpret_cast<const int32_t*>(h.data() + snap::bestd_off);
    for (int y = 0; y < EG_YEARS; ++y) {
      p->best_actions[y].assign(h.data() + snap::best_actions + off[y], h.data() + snap::best_actions + off[y + 1]);
      p->best_deficit[y].assign(h.data() + snap::bestd_actions + offd[y], h.data() + snap::bestd_actions + offd[y + 1]);
      p->cur_run[y] = p->best_actions[y]; p->cur_def[y] = p->best_deficit[y];
    }
    p->has_best_actions = true; p->has_best_deficit = true; p->has_best_weights = true;
    std::memcpy(p->best_w.data(), h.data() + snap::best_w, sizeof(double) * EG_YEARS * EG_N_ACTIONS);
  }
  if (st.n_improvements > c->pulled_improvements) {      // history records are appended once per context, whichever policy pulls
    const DevImprovement* log = reinterpret_cast<const DevImprovement*>(h.data() + snap::imp_log);
    uint32_t from = c->pulled_improvements;
    if (st.n_improvements - from > uint32_t(snap::kImpLogCap)) from = st.n_improvements - uint32_t(snap::kImpLogCap);   // ring overwrote older ones
    const uint32_t keep = p->iteration_count;
    for (uint32_t k = from; k < st.n_improvements; ++k) {
      const DevImprovement& e = log[k % snap::kImpLogCap];
      p->iteration_count = e.iteration;
      p->record_improvement(e.score, e.metrics);
    }
    p->iteration_count = keep;
    c->pulled_improvements = st.n_improvements;
  }
  return EG_OK;
}

int32_t eg_rollout_batch(eg_ctx* c, const eg_policy_snapshot* s, const eg_opts* o, uint64_t seed, uint64_t first_index,
                         uint32_t n, const uint8_t* replay_mask, eg_episode_out* out) {
  int rc = eg_upload_snapshot(c, s, o);
  if (rc != EG_OK) return rc;
  rc = eg_rollout_launch(c, seed, first_index, n, replay_mask);
  if (rc != EG_OK) return rc;
  return eg_fetch(c, out);
}

int32_t eg_timing_reset(eg_ctx* c) {
  if (!c) return EG_ERR_BAD_ARG;
  int rc = collect_timing(c);
  c->total_ms = 0.0; c->grids_ms = 0.0; c->n_launches = 0;
  return rc;
}
int32_t eg_timing_read_grids(eg_ctx* c, double* span_ms, double* grids_ms, int32_t* n_launches) {
  if (!c) return EG_ERR_BAD_ARG;
  int rc = collect_timing(c);
  if (span_ms) *span_ms = c->total_ms;
  if (grids_ms) *grids_ms = c->grids_ms;
  if (n_launches) *n_launches = c->n_launches;
  return rc;
}
int32_t eg_timing_read(eg_ctx* c, double* total_ms, int32_t* n_launches) {
  if (!c) return EG_ERR_BAD_ARG;
  int rc = collect_timing(c);
  if (total_ms) *total_ms = c->total_ms;
  if (n_launches) *n_launches = c->n_launches;
  return rc;
}

int32_t eg_update_stats(eg_ctx* c, int64_t* d_stats) {
  if (!c || !d_stats || !c->snap_valid) { set_error("eg_update_stats: bad argument"); return EG_ERR_BAD_ARG; }
  EG_HIP(hipSetDevice(c->device));
  EG_HIP(hipMemsetAsync(d_stats, 0, sizeof(int64_t) * EG_STATS_LEN, nullptr));
  int lr = launch_update_stats(c->snap, c->out, c->last_n, (long long*)d_stats, nullptr);
  if (lr != 0) { set_error(std::string("k_update_stats launch: ") + hipGetErrorString((hipError_t)lr)); return EG_ERR_HIP; }
  return EG_OK;
}

int32_t eg_fetch_scores(eg_ctx* c, double* scores) {
  if (!c || !scores) return EG_ERR_BAD_ARG;
  EG_HIP(hipSetDevice(c->device));
  if (c->last_n) EG_HIP(hipMemcpy2D(scores, sizeof(double), c->out.base + rec::score, rec::stride, sizeof(double), c->last_n, hipMemcpyDeviceToHost));
  return EG_OK;
}

int32_t eg_fetch_episode_lists(eg_ctx* c, uint32_t i, double metrics[4], int32_t* n_run, uint8_t* run_log, int32_t* n_def,
                               uint8_t* def_log) {
  if (!c || i >= c->last_n || !metrics || !n_run || !run_log || !n_def || !def_log) { set_error("eg_fetch_episode_lists: bad argument"); return EG_ERR_BAD_ARG; }
  EG_HIP(hipSetDevice(c->device));
  EG_HIP(hipMemcpy(metrics, c->out.metrics(i), 4 * sizeof(double), hipMemcpyDeviceToHost));
  EG_HIP(hipMemcpy(n_run, c->out.n_run(i), EG_YEARS * sizeof(int32_t), hipMemcpyDeviceToHost));
  EG_HIP(hipMemcpy(n_def, c->out.n_def(i), EG_YEARS * sizeof(int32_t), hipMemcpyDeviceToHost));
  EG_HIP(hipMemcpy(run_log, c->out.run_log(i), EG_RUN_CAP, hipMemcpyDeviceToHost));
  EG_HIP(hipMemcpy(def_log, c->out.def_log(i), EG_DEF_CAP, hipMemcpyDeviceToHost));
  return EG_OK;
}

int32_t eg_place(eg_ctx* c, int32_t gen_type, int32_t year_index, const uint16_t* extra_cells, int32_t n_extra,
                 int32_t* out_cell, double* out_score) {
  if (!c || gen_type < 0 || gen_type >= EG_N_TYPES || year_index < 0 || year_index >= EG_YEARS || n_extra < 0 || n_extra > EG_ONCHIP_GENS) {
    set_error("eg_place: bad argument"); return EG_ERR_BAD_ARG;
  }
  for (int i = 0; i < n_extra; ++i) if (extra_cells[i] >= EG_CELLS) { set_error("eg_place: cell out of range"); return EG_ERR_BAD_ARG; }
  EG_HIP(hipSetDevice(c->device));
  if (!c->d_place_cells) {      // kept for the life of the context
    EG_HIP(hipMalloc((void**)&c->d_place_cells, sizeof(uint16_t) * EG_MAX_GENS));
    EG_HIP(hipMalloc((void**)&c->d_place_cell, sizeof(int32_t)));
    EG_HIP(hipMalloc((void**)&c->d_place_score, sizeof(double)));
  }
  uint16_t* d_cells = c->d_place_cells; int32_t* d_cell = c->d_place_cell; double* d_score = c->d_place_score;
  if (n_extra) EG_HIP(hipMemcpy(d_cells, extra_cells, sizeof(uint16_t) * n_extra, hipMemcpyHostToDevice));
  int lr = launch_place(c->dev, gen_type, year_index, d_cells, n_extra, d_cell, d_score, nullptr);
  if (lr != 0) { set_error(std::string("k_place launch: ") + hipGetErrorString((hipError_t)lr)); return EG_ERR_HIP; }
  int32_t cell = -1; double score = 0.0;
  EG_HIP(hipMemcpy(&cell, d_cell, sizeof(cell), hipMemcpyDeviceToHost));
  EG_HIP(hipMemcpy(&score, d_score, sizeof(score), hipMemcpyDeviceToHost));
  if (out_cell) *out_cell = cell;
  if (out_score) *out_score = score;
  return EG_OK;
}

int32_t eg_find_suitable_location(eg_ctx* c, int32_t year_index, int32_t gen_type, const double* gen_x, const double* gen_y,
                                  int32_t n_generators, float size_penalty, double* out_x, double* out_y, int32_t* found, double* out_score) {
  if (!c || gen_type < 0 || gen_type >= EG_N_TYPES || year_index < 0 || year_index >= EG_YEARS || n_generators < 0 ||
      (n_generators > 0 && (!gen_x || !gen_y))) { set_error("eg_find_suitable_location: bad argument"); return EG_ERR_BAD_ARG; }
  EG_HIP(hipSetDevice(c->device));
  if (!c->d_place_cell) {
    EG_HIP(hipMalloc((void**)&c->d_place_cells, sizeof(uint16_t) * EG_MAX_GENS));
    EG_HIP(hipMalloc((void**)&c->d_place_cell, sizeof(int32_t)));
    EG_HIP(hipMalloc((void**)&c->d_place_score, sizeof(double)));
  }
  if (n_generators > c->place_xy_cap) {
    if (c->d_place_xy) (void)hipFree(c->d_place_xy);
    c->d_place_xy = nullptr; c->place_xy_cap = 0;
    EG_HIP(hipMalloc((void**)&c->d_place_xy, sizeof(double) * 2 * size_t(n_generators)));
    c->place_xy_cap = n_generators;
  }
  if (n_generators) {
    EG_HIP(hipMemcpy(c->d_place_xy, gen_x, sizeof(double) * n_generators, hipMemcpyHostToDevice));
    EG_HIP(hipMemcpy(c->d_place_xy + c->place_xy_cap, gen_y, sizeof(double) * n_generators, hipMemcpyHostToDevice));
  }
  const double radius = class_radius(c->tables.H.rclass[gen_type]);
  const double size_term = 1.0 - (double(size_penalty) * 0.1);                     // metal_location_search.rs:165
  int lr = launch_place_xy(c->dev, gen_type, year_index, c->d_place_xy, c->d_place_xy + c->place_xy_cap, n_generators, radius, size_term,
                           c->d_place_cell, c->d_place_score, nullptr);
  if (lr != 0) { set_error(std::string("k_place_xy launch: ") + hipGetErrorString((hipError_t)lr)); return EG_ERR_HIP; }
  int32_t cell = -1; double score = 0.0;
  EG_HIP(hipMemcpy(&cell, c->d_place_cell, sizeof(cell), hipMemcpyDeviceToHost));
  EG_HIP(hipMemcpy(&score, c->d_place_score, sizeof(score), hipMemcpyDeviceToHost));
  if (found) *found = cell >= 0 ? 1 : 0;
  if (cell >= 0) { if (out_x) *out_x = double(cell / EG_GRID) * 1000.0; if (out_y) *out_y = double(cell % EG_GRID) * 1000.0; }
  if (out_score) *out_score = score;
  return EG_OK;
}

}  // extern "C"
