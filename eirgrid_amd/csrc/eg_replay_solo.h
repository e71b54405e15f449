// eg_replay_solo.h — a long replay episode on its own wave, taken apart the way the replay hoist takes the batch's one script apart
// (eg_replay_script.h says why that is possible), but for EVERY replay episode by itself: nothing is shared between episodes, the
// record is the episode's own from the first store.  Included by eg_rollout.hip (throughput object) behind k_rollout.
//
// k_rollout<0, kReplayLong> runs a replay the way it runs a sampled episode: year by year, action by action, every action followed by
// the bookkeeping that the NEXT sampled action could depend on (aggregates, state, the repair loop's evaluation, the logs).  For a
// replay of 228 generators that is one serial stream of ~2.2 ms of which the searches are 43 % and the field updates 14 %
// (profiles/r03_ab_notes.log r03ao); the rest is per-action code a replay does not need between two placements.  Here the episode is
//   1. its script (rs::script: actions, logs, counts, the lists' pack words — 64 additional actions at a time),
//   2. its placements, one after the other, with k_rollout's own searches (place_search / place_heavy / place_exact_long, the penalty
//      field of the pool, the lists' window in LDS and their tail in the record — the same code, so the same cells and the same
//      accounting of requested chunks),
//   3. its yearly rows (rs::books_quad, four years at a time), the running totals, the header and the statistics epilogue.
// An episode whose script cannot be finished without a seeded draw or a capacity, or that finds no location, publishes nothing:
// k_rollout<0, kReplayLong>, launched behind this kernel, runs every episode whose word in `done` does not carry the batch's sequence
// number — from the start, into the same record.
#pragma once

namespace solo {

// the script's storage: pack words in the record (read back past the CU's L1: this wave stored them), lengths in LDS — the policy
// row block of Smem, which a replay never samples from
struct Lists {
  unsigned long long gen_pack;
  __device__ __forceinline__ static int* words() { return reinterpret_cast<int*>(sm.pol); }      // [0,26) g_end, [26,52) o_end, [52,57) lens, [58,60) bytes
  __device__ __forceinline__ int gpack_at(int i) const {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // (the script's own stores have left the wave)
    return tail_u16(gen_pack, i);
  }
  __device__ __forceinline__ void gpack_put(int, int) const {}
  __device__ __forceinline__ void opack_put(int, int) const {}
  __device__ __forceinline__ int cls(int t) const { return (sm.type_info[t] >> 12) & 3; }
  __device__ __forceinline__ double out(int t) const { return sm.type_out[t]; }
  __device__ __forceinline__ void year_end(int yi, int ngen, int noff) const { words()[yi] = ngen; words()[EG_YEARS + yi] = noff; }
  __device__ __forceinline__ void finish(int run_pos, int def_pos, int act_pos, int ngen, int noff, unsigned long long bytes) const {
    int* w = words();
    w[52] = run_pos; w[53] = def_pos; w[54] = act_pos; w[55] = ngen; w[56] = noff;
    *reinterpret_cast<unsigned long long*>(w + 58) = bytes;
  }
};
static_assert(snap::kPolRow * 8 >= 60 * 4, "the script's lengths fit the policy row block");

// The field update with the lane's list entries made ready once per class set (k_rollout's heavy_add_body decodes its entries — offsets,
// class, the factor's place in the table — for every generator again: ~20 instructions an entry on the episode's serial path; here an
// entry costs the bounds test, the address, the read-multiply-write).  The first eight entries per lane (the list of up to three or
// four radius classes); lists that are longer go on through heavy_add_body.
constexpr int kEnt = 8;
struct Entries { int base[kEnt], delta[kEnt], dd[kEnt]; double fac[kEnt]; };      // class's first cell, di * 51 + dj, di | dj << 16, d/R
__device__ __forceinline__ void load_entries(Entries& E, int lane) {      // from sh2.box: the episode's list as k_rollout keeps it (8 entries a lane)
#pragma unroll
  for (int k = 0; k < kEnt; ++k) {
    const uint32_t en = sh2.box[k * kWave + lane];      // di + 16 | (dj + 16) << 5 | place in sm.dr << 10 | class << 19
    const int rc = (int)(en >> 19), di = (int)(en & 31u) - 16, dj = (int)((en >> 5) & 31u) - 16;
    E.base[k] = rc * kFieldStride; E.delta[k] = di * kGrid + dj; E.dd[k] = (di & 0xFFFF) | (int)((unsigned)dj << 16);
    E.fac[k] = factor_by_q<false>(rc, (int)offsetof(Smem, dr), (int)((en >> 10) & 511u));
  }
}
// field[class][cell + (di, dj)] *= d/R for the lane's first `n` entries (4 or 8; uniform).  No control flow per entry: an entry that
// falls off the grid goes to its class's spare entry (never read), padding entries multiply by exactly 1.0 (heavy_add_body's rules).
template <int kN>
__device__ __forceinline__ void field_add(unsigned long long field_addr, const Entries& E, int cell) {
  const GlobalF64 base = (GlobalF64)field_addr;
  const int gi = cell / kGrid, gj = cell - gi * kGrid;
  double val[kN]; int off[kN];
#pragma unroll
  for (int k = 0; k < kN; ++k) {
    const int ci = gi + (int)(short)(E.dd[k] & 0xFFFF), cj = gj + (E.dd[k] >> 16);
    const bool inside = (unsigned)ci < (unsigned)kGrid && (unsigned)cj < (unsigned)kGrid;
    off[k] = E.base[k] + (inside ? cell + E.delta[k] : kCells);
    val[k] = field_load(base + off[k]);
  }
#pragma unroll
  for (int k = 0; k < kN; ++k) base[off[k]] = val[k] * E.fac[k];
}

}  // namespace solo

__global__ void __launch_bounds__(kWave, EG_HEAVY_WAVES) k_replay_solo(DevTables T, DevSnapshot S_in, DevOut O, unsigned long long first_index, uint32_t n_episodes,
                                                                       const uint8_t* __restrict__ replay_mask, uint32_t replay_period, long long* stats,
                                                                       EpisodeMap emap) {
  const int lane = threadIdx.x;
  if (blockIdx.x >= emap.count) return;
  if (!(S_in.state()->has_lists && S_in.best_off()[EG_YEARS] > kShortReplayMax)) return;      // (uniform for the whole grid: the short variant's)
  if (emap.hoist_seq != 0ull && *emap.hoist == emap.hoist_seq) return;                        // served by the replay hoist
  const uint32_t e = map_episode(emap, blockIdx.x);
  if (e >= n_episodes) return;
  DevSnapshot S = S_in;
  load_state(S);
  const bool replay = replay_mask != nullptr ? replay_mask[e] != 0 : (replay_period != 0u && S.has_best_actions && (first_index + e) % replay_period == 0ull);
  if (!replay || !S.has_best_actions) return;      // an ordinary episode, or a flagged one without lists (fallback draws from the first action on)
#ifdef EG_SOLO_STAMPS      // diagnostic build (make ab AB=solostamps ABFLAGS=-DEG_SOLO_STAMPS; scripts/solo_stamps.py): cycles of the phases
  unsigned long long cs[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cs_last = __builtin_readcyclecounter();
#define EG_SS(slot) do { const unsigned long long now_ = __builtin_readcyclecounter(); cs[slot] += now_ - cs_last; cs_last = now_; } while (0)
#else
#define EG_SS(slot) do {} while (0)
#endif
  load_static_tables(T, lane, true);
#if defined(EG_SOLO_STAMPS) && defined(EG_STAMPS)
  if (lane < 8) sm.hdbg[lane >> 2][lane & 3] = 0ull;
#endif
  wave_sync();
  EG_SS(0);      // 0: set-up

  // ---- 1. the script ----
  solo::Lists ls{(unsigned long long)O.gen_pack(e)};
  if (!rs::script(T, S, O, e, ls, lane)) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  wave_sync();
  EG_SS(1);      // 1: the script
  const int* words = solo::Lists::words();
  const int n_gens = __builtin_amdgcn_readfirstlane(words[55]);

  // ---- 2. the placements: k_rollout's apply_action for a generator, minus the aggregates ----
  Episode ep;
  ep.ngen = 0; ep.chunks = 0; ep.heavy = -1; ep.heavy_classes = 0; ep.heavy_quads = 0;
  uint16_t* gen_cell = O.gen_cell(e);
  const ListTail tail = {(unsigned long long)gen_cell, (unsigned long long)O.gen_pack(e), (unsigned long long)O.off_pack(e)};
  PrefixCache prefix_cache0 = {0.0, -1, 0};
  uint32_t search_seq = 0;
  int pk_block = 0;
  solo::Entries E;
#pragma unroll
  for (int k = 0; k < solo::kEnt; ++k) { E.base[k] = 0; E.delta[k] = 0; E.dd[k] = 0; E.fac[k] = 1.0; }
  for (int g = 0; g < n_gens; ++g) {
    if ((g & (kWave - 1)) == 0) pk_block = g + lane < n_gens ? tail_u16(tail.gen_pack, g + lane) : 0;      // 64 pack words, one per lane
    const int pk = __builtin_amdgcn_readlane(pk_block, g & (kWave - 1));
    const int t = pk & 15, yi = (pk >> 4) & 31, m = pk >> 9;
    double m03v = 0.0;
    int cell = -1;
    bool placed = false;
    auto between = [] {};
    EG_SS(2);      // 2: placements: the list entry
    if (ep.ngen >= kHeavyGens && ep.heavy != -2) {      // a long list: approximate field + exact evaluation of the few candidates
      const unsigned long long slot_bytes = (unsigned long long)(kRadiusClasses * kFieldStride) * 8ull;
      if (ep.heavy == -1) ep.heavy = heavy_claim(T, lane);
      if (ep.heavy >= 0) {
        const int info = __builtin_amdgcn_readfirstlane(sm.type_info[t]);
        const int hv = info & 15, hrc = (info >> 4) & 15;
        const unsigned long long class_addr = (unsigned long long)T.heavy + (unsigned long long)ep.heavy * slot_bytes + (unsigned long long)(hrc * kFieldStride) * 8ull;
        if (!((ep.heavy_classes >> hrc) & 1)) {      // the first search of this radius class: its field joins
          heavy_build_class<false>(class_addr, tail.gen_cell, lane, hrc, throughput_table(info), (info >> 8) & 15, ep.ngen);
          ep.heavy_classes |= 1 << hrc;
          ep.heavy_quads = __builtin_amdgcn_readfirstlane(T.hv_quads()[ep.heavy_classes]);
          wave_sync();
#pragma unroll
          for (int k = 0; k < kBoxLds / kWave; ++k) sh2.box[k * kWave + lane] = T.hv_lists()[ep.heavy_classes * 1024 + k * kWave + lane];
          wave_sync();
          solo::load_entries(E, lane);      // the lane's entries, ready for every field update until the next class joins
        }
        const size_t yv = (size_t)(yi * kMaxVariants + hv) * kPsStride, yc = (size_t)(yi * kMaxVariants + hv) * kPcStride;
        const int hr = place_heavy<false>((unsigned long long)(T.ps() + yv), (unsigned long long)(T.pbase() + yc), (unsigned long long)(T.pcell() + yc), class_addr,
                                          tail.gen_cell, T.size_factor, lane, hrc, throughput_table(info), ep.ngen);
        if (hr >= 0) { cell = hr & 0xFFFF; ep.chunks += hr >> 16; placed = true; }
      }
    }
    if (!placed && ep.ngen > kLdsGens) {      // the exact scan, for a list beyond the window (place_search walks LDS)
      const int info = __builtin_amdgcn_readfirstlane(sm.type_info[t]);
      const int xr = place_exact_long<false>((unsigned long long)(T.ps() + (size_t)(yi * kMaxVariants + (info & 15)) * kPsStride), tail.gen_cell,
                                             T.size_factor, lane, (info >> 4) & 15, throughput_table(info), ep.ngen);
      cell = xr < 0 ? -1 : (xr & 0xFFFF);
      if (xr >= 0) ep.chunks += xr >> 16;
      placed = true;
    }
    if (!placed) cell = __builtin_amdgcn_readfirstlane(place_search<0>(T, lane, yi, t, ep.ngen, nullptr, &m03v, prefix_cache0, between, ep.chunks, &search_seq));
    EG_SS(3);      // 3: placements: the search
    if (cell >= 0 && ep.heavy_classes != 0) {      // the field of every class the episode keeps, for the new generator
      const unsigned long long field_addr = (unsigned long long)T.heavy + (unsigned long long)ep.heavy * ((unsigned long long)(kRadiusClasses * kFieldStride) * 8ull);
      ep.chunks += 2 * ep.heavy_quads;
      const unsigned long long hv_list = (unsigned long long)(T.hv_lists() + ep.heavy_classes * 1024);
      if (ep.heavy_quads == 1) solo::field_add<4>(field_addr, E, cell);
      else solo::field_add<8>(field_addr, E, cell);
      if (ep.heavy_quads == 3) heavy_add_body<false, 4>(field_addr, hv_list, lane, cell, 8);
      else if (ep.heavy_quads >= 4) heavy_add_body<false, 8>(field_addr, hv_list, lane, cell, 8);
    }
    if (cell < 0) return;      // EG_EP_NO_LOCATION (or a lost search): the classic path reports it
    if (lane == 0) {
      if (ep.ngen < kLdsGens) { sm.gcell[ep.ngen] = (uint16_t)(cell | (t << 12)); sm.gbm[ep.ngen] = (uint8_t)(yi | (m << 5)); }
      gen_cell[ep.ngen] = (uint16_t)cell;
    }
    wave_sync();
    ep.ngen += 1;
    EG_SS(4);      // 4: placements: field update, list entry
  }

  // ---- 3. the yearly rows, the running totals (metrics_calculation.rs:133-153), the header ----
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  wave_sync();
  double total_cost = 0.0, total_credit = 0.0, total_sales = 0.0;
  double last[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int y0 = 0; y0 < kYears; y0 += 4) {      // four years at a time, a row of sixteen lanes a year (rs::books_quad)
    const rs::YearRow q = rs::books_quad(T, S_in, O, e, y0, words, words + EG_YEARS, lane, [](const uint16_t* list, int i) { return tail_u16((unsigned long long)list, i); });
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int yi = y0 + r;
      if (yi >= kYears) break;
#pragma unroll
      for (int k = 0; k < 7; ++k) last[k] = readlane_f64(q.v[k], 16 * r);
      total_cost = yi == 0 ? last[0] : total_cost + last[0];
      total_credit = yi == 0 ? last[1] : total_credit + last[1];
      total_sales = yi == 0 ? last[2] : total_sales + last[2];
      if (S_in.write_yearly && lane == 0) {
        double* row = O.yearly(e) + yi * EG_YEARLY_FIELDS;
        row[EG_Y_TOTAL_COST] = total_cost; row[EG_Y_TOTAL_CREDIT] = total_credit; row[EG_Y_TOTAL_SALES] = total_sales;
      }
    }
  }
  EG_SS(5);      // 5: the yearly rows
  if (lane == 0) {      // SimulationMetrics, iteration.rs:69-74 (Q2: total_cost is the last year's capital cost)
    O.metrics(e)[0] = last[3]; O.metrics(e)[1] = last[4]; O.metrics(e)[2] = last[5];
    O.metrics(e)[3] = last[6] >= 0.0 ? 1.0 : 0.0;
    *O.status(e) = EG_EP_OK; *O.n_gens(e) = words[55]; *O.n_offsets(e) = words[56];
    *O.n_draws(e) = 0ull;      // (a replay that needs no fallback draws nothing)
    *O.bytes_moved(e) = (double)*reinterpret_cast<const unsigned long long*>(words + 58);
    *O.n_chunks(e) = (uint32_t)ep.chunks;
  }
  if (stats != nullptr) {      // the statistics epilogue, as k_rollout runs it
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    wave_sync();
    StatsParams P;
    load_stats_params(S, P);
    episode_update_stats(O, S, P, e, lane, stats, 1ull, emap.stats_rep ? (int)(blockIdx.x % (uint32_t)kStatsReplicas) : -1);
  }
  EG_SS(6);      // 6: header, statistics epilogue
#ifdef EG_SOLO_STAMPS
  if (lane == 0) {      // (diagnostic build only: the cycle counts go to the otherwise unread tail of this episode's act_log buffer)
    unsigned long long* dbg = (unsigned long long*)(O.act_log(e) + EG_ACT_CAP - 256);
    for (int i = 0; i < 8; ++i) dbg[i] = cs[i];
#ifdef EG_STAMPS      // (with -DEG_STAMPS as well: place_heavy's own counters — scan / candidates / exact evaluation cycles; chunks, candidates, searches)
    for (int i = 0; i < 3; ++i) { dbg[8 + i] = sm.hdbg[0][i]; dbg[11 + i] = sm.hdbg[1][i]; }
#endif
  }
#endif
  if (lane == 0) emap.solo[blockIdx.x] = emap.solo_seq;      // (k_rollout<0, kReplayLong> is stream-ordered behind this kernel)
}
