// eg_replay_coop.h — the replay episodes of a batch, computed once.  Included by eg_rollout.hip (inside its anonymous namespace,
// eg_rollout.o only): it uses that file's wave primitives, aggregates and statistics epilogue.
//
// An episode that replays the best strategy (iteration.rs:34-42, force_best_actions) takes its actions from the stored lists
// (sampling.rs:78-101 and :242-266 return before any `gen`; simulation.rs:146-162 takes the year's count from the list) and reads
// no seeded draw until a list runs out (the smart fallbacks, sampling.rs:445-528) — so every replay episode of a batch, whatever its
// index, is the SAME computation from the same snapshot, and in the sustained state of configs[2] 1 638 bit-identical copies of one
// serial instruction stream were the batch's long pole (k_rollout<0,2> 2.37 ms beside the lean grid's 1.27 ms).
//
// That one computation is taken apart by what depends on what (profiles/r04_ab_notes.log has the measurements that led here):
//   1. THE SCRIPT (k_replay_coop, wave 0).  Which actions the episode takes — the repair loop's trips, the forced batteries, the
//      year's additional actions, every log and count — depends on the lists and on the power balance only, and the balance only on
//      the TYPES of the generators placed (map_handler.rs:829-868), never on where they land.  One wave expands the whole episode:
//      k_rollout's replay path statement for statement, minus the placement, the in-episode weight nudges (learning.rs:21-88,
//      deficit.rs:82-135: they only touch the episode's private copy of the tables, which a replay never samples from and which is
//      dropped at its end — Q5) and every aggregate but the three output class sums.
//   2. THE PLACEMENTS (k_replay_coop, all eight waves — two per SIMD of a CU: with four, one per SIMD, a placement took 2 545 cycles; a
//      second wave per SIMD issues into the first one's waits; sixteen are slower again, profiles/r04_ab_notes.log r04z).  Serial by nature: a
//      search sees every generator before it.  A search is an arg-max over all 2 601 candidate cells at once, six cells per lane:
//      the product of the penalty factors of every generator placed so far is kept per radius class and cell in LDS (field[6][2624]
//      f64, 126 KB of the CU's 160 KB; updated for the new generator by one lane per (class, di, dj) entry of the host's list), the
//      year's unpenalised scores (tab::cbase) sit in registers, so approx(c) = base(c) * field[c] is a read and a multiplication per
//      cell.  One exchange through LDS gives the largest approximate score and says whether a second cell comes within 2^-16 of it.
//      If none does — the usual case — the holder IS the reference's arg-max (exact and approximate score are the same real product
//      rounded at most G + 3 times each: they differ by less than 2^-40 relative for the 4 096 generators a list can hold) and only
//      its cell is asked for.  Otherwise every cell within 2^-30 of the exact maximum of the approximate scores is evaluated EXACTLY
//      — te times the factors in list order, a candidate a wave (exact_product_chain's arithmetic) — and the first maximum in cell
//      order wins; subnormal ranges, more than 64 such cells or no positive score fall back to the exact scan of all cells (every
//      lane folds the whole list for its six cells).  The winner is metal_location_search.rs:110-176's, bit for bit, as with
//      place_search / place_heavy.
//   3. THE YEARLY ROWS (k_replay_books, a wave a year).  Every aggregate of a year is a sum over the generators / offsets in list
//      order (map_handler.rs:829-965): started from the existing-plant prefix at the year's start and continued with every addition,
//      i.e. one sequential sum over the list as it stands at the year's end — and the years do not depend on each other once the
//      lists are known.  Sixteen waves fold them side by side (fold2_row16: one DPP multiply-accumulate per list element and sum),
//      one lane then runs the three running totals of metrics_calculation.rs:133-153 through the 26 years.
// The record goes to a scratch slot; k_replay_broadcast copies it into the record of every replay episode of the batch and runs the
// statistics epilogue for each of them (episode_update_stats: the same function, so the update packet is the per-episode path's by
// construction).  If the script cannot be finished without a seeded draw (a fallback), or would end with a status other than
// EG_EP_OK (a capacity, no location), nothing is published and the per-episode replay variants — launched behind these kernels in
// any case — run the episodes as before: they return at once only when HoistInfo::served_seq carries their batch's sequence number.
#pragma once

#ifndef EG_COOP_WAVES
#define EG_COOP_WAVES 8
#endif
namespace coop {

constexpr int kWaves = EG_COOP_WAVES, kThreads = kWaves * kWave;
constexpr int kPer = (kCells + kThreads - 1) / kThreads;      // candidate cells per lane: 6 (cell = tid + 512 k)
constexpr int kEnt = 1024 / kThreads;                         // entries of the field update per lane: 2
constexpr double kKeepCoop = 1.0 - 0x1p-30;
constexpr int kVariants = 8;                                  // (radius class, marine) pairs whose scores a lane keeps in registers — the reference's
                                                              // fifteen types make eight; a world with more is not hoisted (eg_api.cpp)
constexpr int kSpare = kFieldStride - 2;                      // where a field update that falls off the grid goes (class-relative; never a candidate)
static_assert(kPer <= 16 && (kPer - 1) * kThreads < kCells && kFieldStride >= kCells + 2, "cell slots of a lane; padding of a class's field");

struct __align__(16) Smem {
  double field[kRadiusClasses * kFieldStride];      // per radius class and cell: product of the penalty factors of every generator so far
  double dr[kDrCompact];                            // d/R by squared cell distance (tab::dr_compact)
  double type_out[16];
  int tinfo[16];                                    // variant | radius class << 4 | marine << 8 | output class << 9 | cap << 16 | first entry / 2 << 24
  uint16_t gcell[EG_MAX_GENS];                      // the episode's generators: cell (phase 2)
  uint16_t gpack[EG_MAX_GENS];                      //   type | build-year index << 4 | multiplier index << 9 (the record's gen_pack; phase 1)
  uint16_t opack[EG_MAX_OFFSETS];                   // the episode's offsets (the record's off_pack; phase 1)
  struct __align__(16) { uint32_t top1, top2; int cell, pad; } top[kWaves];      // a search's exchange: high words of each wave's two largest approximate scores
  int n_gens, failed;                               // phase 1 -> phase 2
  double xscore[kWaves]; int xcell[kWaves]; int xcnt[kWaves];      // slow paths: per-wave maxima / candidate counts
  int cand[kWave]; double cand_score[kWave];
};
static_assert(sizeof(Smem) <= 160 * 1024, "one workgroup owns the CU's LDS");
__shared__ Smem sc;
// Every barrier of this kernel orders LDS only (wg_barrier_lds: s_waitcnt lgkmcnt(0); s_barrier): __syncthreads() would also wait for
// the wave's outstanding global stores and loads.

#ifdef EG_COOP_STAMPS      // diagnostic build (make ab AB=coopstamps ABFLAGS=-DEG_COOP_STAMPS; scripts/coop_stamps.py)
#define EG_CS(slot) do { const unsigned long long now_ = __builtin_readcyclecounter(); cs[slot] += now_ - cs_last; cs_last = now_; } while (0)
#else
#define EG_CS(slot) do {} while (0)
#endif
#ifdef EG_COOP_STAMPS
#define EG_CS_ARGS , unsigned long long* cs, unsigned long long& cs_last
#define EG_CS_PASS , cs, cs_last
#else
#define EG_CS_ARGS
#define EG_CS_PASS
#endif

// v_max_f64 / v_min_f64 as such (from a > b ? a : b the compiler makes a compare and two selects; the values here are never NaN)
__device__ __forceinline__ double vmax(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double vmin(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

__device__ __forceinline__ double factor_of(int ci, int cj, int gc, int off, int cap) {
  const int gi = gc / kGrid, gj = gc - gi * kGrid;
  int q = (ci - gi) * (ci - gi) + (cj - gj) * (cj - gj);
  q = q < cap ? q : cap;      // (the class's own radius squared: the table holds 1.0 there, and x * 1.0 == x)
  return sc.dr[off + q];
}

// te times the factor of every generator in list order, for one cell (every lane of the wave returns it): exact_product_chain's
// arithmetic — a generator at or beyond the radius multiplies by exactly 1.0 and is skipped
__device__ __forceinline__ double exact_chain(int off, int cap, int ngen, double te, int cell, int lane) {
  const int ci = cell / kGrid, cj = cell - ci * kGrid;
  double s = te;
  for (int gb = 0; gb < ngen; gb += kWave) {
    double f = 1.0;
    if (gb + lane < ngen) f = factor_of(ci, cj, (int)sc.gcell[gb + lane], off, cap);
    unsigned long long near = __ballot(f != 1.0);
    while (near != 0ull) {
      const int j = __ffsll((long long)near) - 1;
      s = s * readlane_f64(f, j);
      near &= near - 1ull;
    }
  }
  return s;
}

// ---- a search, fast path: this lane's cells against the field, the exchange, the decision.  `b`: the lane's unpenalised scores of the
//      (year, variant) — registers; returns the winning cell, or -2: several cells within reach of the maximum / nothing placeable (the
//      caller takes the slow path).  Called by all waves; every lane returns the same value. ----
__device__ __forceinline__ uint32_t med3_u32(uint32_t a, uint32_t b, uint32_t c) { uint32_t r; asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ int scan_fast(const double (&b)[kPer], int rc, int tid, int lane, int wave EG_CS_ARGS) {
  const char* f0 = reinterpret_cast<const char*>(sc.field) + (rc * kFieldStride + tid) * 8;
  // The lane's largest and second largest approximate score, as 32-bit keys: the score's high word (scores are not negative: ordered
  // like their bit patterns) with the cell slot k in its four lowest bits — a key bounds its score within 2^-16, which is all the
  // decision below needs, and a v_max / v_med3 pair per cell keeps value, place and runner-up (l1 >= l2 throughout, so the median of
  // {l1, l2, key} is the new runner-up).
  uint32_t l1 = 0u, l2 = 0u;
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    // (the last slot reaches beyond the grid for most lanes: their score there is 0 and the read is bent to the class's padding)
    const double f = k + 1 < kPer ? *reinterpret_cast<const double*>(f0 + k * kThreads * 8)
                                  : sc.field[rc * kFieldStride + (tid + k * kThreads < kFieldStride ? tid + k * kThreads : kFieldStride - 1)];
    const double a = b[k] * f;
    const uint32_t key = ((uint32_t)__double2hiint(a) & ~15u) | (uint32_t)k;
    l2 = med3_u32(l1, l2, key);
    l1 = l1 > key ? l1 : key;
  }
  EG_CS(3);      // 3: searches: the lane's cells
  const int lcell = tid + (int)(l1 & 15u) * kThreads;
  const uint32_t w1 = wave_max_u32(l1);
  const unsigned long long hold = __ballot(l1 == w1);
  uint32_t w2 = wave_max_u32(l1 == w1 ? l2 : l1);
  if (__popcll(hold) > 1) w2 = w1;
  const int wcell = __builtin_amdgcn_readlane(lcell, __ffsll((long long)hold) - 1);
  if (lane == 0) { sc.top[wave].top1 = w1; sc.top[wave].top2 = w2; sc.top[wave].cell = wcell; }
  EG_CS(4);      // 4: searches: the wave's two largest
  wg_barrier_lds();
  EG_CS(6);      // 6: searches: the exchange's barrier
  // (lane w < 4 reads wave w's entry — keys and cell in ONE 16-byte read: every LDS round trip of a search is on the batch's serial path)
  int4 tv = {0, 0, 0, 0};
  if (lane < kWaves) tv = *reinterpret_cast<const int4*>(&sc.top[lane]);
  const uint32_t r1 = (uint32_t)tv.x, r2 = (uint32_t)tv.y;
  const uint32_t mh = wave_max_u32(r1);
  const double m_lo = __hiloint2double((int)(mh & ~15u), 0);      // a lower bound of the largest approximate score, within 2^-16
  if (!(m_lo >= 1e-250)) return -2;                                // nothing placeable, or subnormal territory
  const double thr = m_lo * kKeepCoop;
  const unsigned long long c1 = __ballot(lane < kWaves && __hiloint2double((int)(r1 | 15u), -1) >= thr);      // (upper bounds of the scores behind the keys)
  const unsigned long long c2 = __ballot(lane < kWaves && __hiloint2double((int)(r2 | 15u), -1) >= thr);
  // ONE candidate: it is the arg-max (the arg-max is among the candidates) and only its cell is asked for
  if (__popcll(c1) + __popcll(c2) == 1) return __builtin_amdgcn_readlane(tv.z, __ffsll((long long)c1) - 1);
  return -2;
}

// ---- a search, slow path (several cells within reach of the maximum, or nothing decided): place_heavy's steps 2 and 3, then the
//      exact scan.  `tb`: the lane's unpenalised scores.  Returns the cell, or -1: no candidate has a positive score. ----
// `skip_candidates` (test hook, EIRGRID_COOP_FORCE=2): go straight to the exact scan.
__device__ __forceinline__ int scan_slow(const DevTables& T, const double (&tb)[kPer], int yi, int t, int ngen, int tid, int lane, int wave, bool skip_candidates) {
  const int info = __builtin_amdgcn_readfirstlane(sc.tinfo[t]);
  const int rc = (info >> 4) & 15, off = (int)((unsigned)info >> 24) << 1, cap = (info >> 16) & 255;
  const bool marine = ((info >> 8) & 1) != 0;
  const double size_factor = T.size_factor;
  double ap[kPer];
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int cell = tid + k * kThreads;
    ap[k] = cell < kCells ? tb[k] * sc.field[rc * kFieldStride + cell] : 0.0;
  }
  {
    double lm = ap[0];
#pragma unroll
    for (int k = 1; k < kPer; ++k) lm = dmax(lm, ap[k]);
    const double wm = wave_max_f64(lm);
    wg_barrier_lds();      // (everybody is through with the fast path's exchange)
    if (lane == 0) sc.xscore[wave] = wm;
    wg_barrier_lds();
    const double M = wave_max_f64(lane < kWaves ? sc.xscore[lane] : 0.0);
    const double thr = M * kKeepCoop;
    unsigned long long fm[kPer]; int mine = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) { fm[k] = __ballot(ap[k] >= thr && ap[k] > 0.0); mine += __popcll(fm[k]); }
    if (lane == 0) sc.xcnt[wave] = mine;
    wg_barrier_lds();
    int total = 0, before = 0;
    for (int w = 0; w < kWaves; ++w) { const int c = sc.xcnt[w]; total += c; before += w < wave ? c : 0; }
    total = __builtin_amdgcn_readfirstlane(total);
    if (M >= 1e-250 && total != 0 && total <= kWave && !skip_candidates) {
      int pos = before;
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        if ((fm[k] >> lane) & 1ull) sc.cand[pos + __popcll(fm[k] & ((1ull << lane) - 1ull))] = tid + k * kThreads;
        pos += __popcll(fm[k]);
      }
      wg_barrier_lds();
      for (int c = wave; c < total; c += kWaves) {      // a candidate a wave
        const int cell = sc.cand[c];
        const double tev = T.te_cell()[(size_t)(yi * kRadiusClasses + rc) * kCells + cell];
        const double cfe = marine ? T.coastf()[cell] : 1.0;
        const double sk = (exact_chain(off, cap, ngen, tev, cell, lane) * cfe) * size_factor;
        if (lane == 0) sc.cand_score[c] = sk;
      }
      wg_barrier_lds();
      const double sl = lane < total ? sc.cand_score[lane] : 0.0;
      const int cl = lane < total ? sc.cand[lane] : kCells;
      const ChunkBest b = chunk_reduce<false>(sl, cl, 0.0);      // highest score, ties to the lowest cell = the first maximum in cell order
      wg_barrier_lds();      // (cand / cand_score are free again)
      if (b.score > 0.0) return b.cell;
    }
  }
  // ---- the exact scan: every lane folds the whole list, in list order, for each of its cells ----
  double lb = 0.0; int lc = kCells;
  for (int k = 0; k < kPer; ++k) {
    const int cell = tid + k * kThreads;
    if (cell >= kCells) break;
    const int ci = cell / kGrid, cj = cell - ci * kGrid;
    double s = T.te_cell()[(size_t)(yi * kRadiusClasses + rc) * kCells + cell];
    for (int g = 0; g < ngen; ++g) s = s * factor_of(ci, cj, (int)sc.gcell[g], off, cap);
    s = (s * (marine ? T.coastf()[cell] : 1.0)) * size_factor;
    if (s > lb) { lb = s; lc = cell; }      // (cells ascend with k: the first maximum stays)
  }
  const ChunkBest wb = chunk_reduce<false>(lb, lc, 0.0);
  wg_barrier_lds();
  if (lane == 0) { sc.xscore[wave] = wb.score; sc.xcell[wave] = wb.score > 0.0 ? wb.cell : kCells; }
  wg_barrier_lds();
  const ChunkBest b = chunk_reduce<false>(lane < kWaves ? sc.xscore[lane] : 0.0, lane < kWaves ? sc.xcell[lane] : kCells, 0.0);
  wg_barrier_lds();
  return b.score > 0.0 ? b.cell : -1;
}

// the lane's entries of the field update (tab::hv_lists, the list of all six classes), made ready: byte offset of the entry's cell
// relative to the generator's, where the class's spare entry is, (di, dj), d/R
struct Entries { int delta[kEnt], spare[kEnt], di[kEnt], dj[kEnt]; double fac[kEnt]; };

// field[class][cell] *= d/R of a generator at `cell`.  No control flow: an entry that falls off the grid goes to its class's spare
// entry (never a candidate), padding entries multiply by exactly 1.0; distinct entries are distinct cells of their class.
__device__ __forceinline__ void field_add(const Entries& E, int cell) {
  const int gi = cell / kGrid, gj = cell - gi * kGrid;
  double v[kEnt]; int at[kEnt];
#pragma unroll
  for (int j = 0; j < kEnt; ++j) {
    const bool inside = (unsigned)(gi + E.di[j]) < (unsigned)kGrid && (unsigned)(gj + E.dj[j]) < (unsigned)kGrid;
    at[j] = inside ? cell * 8 + E.delta[j] : E.spare[j];
    v[j] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(sc.field) + at[j]);
  }
#pragma unroll
  for (int j = 0; j < kEnt; ++j) *reinterpret_cast<double*>(reinterpret_cast<char*>(sc.field) + at[j]) = v[j] * E.fac[j];
}

// the script's storage (rs::script, eg_replay_script.h): the lists in LDS, the lengths in HoistInfo
struct Lists {
  HoistInfo* info;
  __device__ __forceinline__ int gpack_at(int i) const { return (int)sc.gpack[i]; }
  __device__ __forceinline__ void gpack_put(int at, int pk) const { sc.gpack[at] = (uint16_t)pk; }
  __device__ __forceinline__ void opack_put(int at, int pp) const { sc.opack[at] = (uint16_t)pp; }
  __device__ __forceinline__ int cls(int t) const { return (sc.tinfo[t] >> 9) & 3; }
  __device__ __forceinline__ double out(int t) const { return sc.type_out[t]; }
  __device__ __forceinline__ void year_end(int yi, int ngen, int noff) const { info->g_end[yi] = ngen; info->o_end[yi] = noff; }
  __device__ __forceinline__ void finish(int run_pos, int def_pos, int act_pos, int ngen, int noff, unsigned long long bytes) const {
    info->lens[0] = run_pos; info->lens[1] = def_pos; info->lens[2] = act_pos; info->lens[3] = ngen; info->lens[4] = noff;
    info->bytes = bytes;
    sc.n_gens = ngen;
  }
};

}  // namespace coop

// One workgroup: the batch's replay script and its placements, once.  `O`: the scratch record (episode 0 of it).
// `force` (test hook, EIRGRID_COOP_FORCE): 1 = every search takes the slow path (exact evaluation of the candidates), 2 = every search is the
// exact scan of all cells — the rarely-run paths, held against the per-episode kernels and the oracle by tests/test_gpu_replay_hoist.py.
__global__ void __launch_bounds__(coop::kThreads, 1) k_replay_coop(DevTables T, DevSnapshot S_in, DevOut O, unsigned long long hoist_seq, HoistInfo* info, int force) {
  using namespace coop;
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  DevSnapshot S = S_in;
  load_state(S);
  if (!S.has_best_actions) return;      // no best strategy on the device: the flagged episodes are ordinary ones (k_rollout decides per episode)
#ifdef EG_COOP_STAMPS
  unsigned long long cs[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cs_last = __builtin_readcyclecounter();
#endif
  // ---- tables -> LDS, the lane's constants -> registers ----
  for (int i = tid; i < kRadiusClasses * kFieldStride; i += kThreads) sc.field[i] = 1.0;
  for (int i = tid; i < kDrCompact; i += kThreads) sc.dr[i] = T.dr_compact()[i];
  if (tid < kTypes) {
    const int rc = T.rclass()[tid];
    sc.tinfo[tid] = T.variant()[tid] | (rc << 4) | ((T.marine()[tid] ? 1 : 0) << 8) | (T.cls()[tid] << 9) | (T.dr_meta()[8 + rc] << 16) |
                    (int)((unsigned)(T.dr_meta()[rc] >> 1) << 24);
    sc.type_out[tid] = T.out_mw()[tid];
  }
  if (tid == 0) { sc.n_gens = 0; sc.failed = 0; }
  Entries E;
#pragma unroll
  for (int j = 0; j < kEnt; ++j) {      // entry = di + 16 | (dj + 16) << 5 | place in dr << 10 | class << 19
    const uint32_t en = T.hv_lists()[63 * 1024 + j * kThreads + tid];
    const int rc = (int)(en >> 19);
    E.di[j] = (int)(en & 31u) - 16; E.dj[j] = (int)((en >> 5) & 31u) - 16;
    E.delta[j] = (rc * kFieldStride + E.di[j] * kGrid + E.dj[j]) * 8;
    E.spare[j] = (rc * kFieldStride + kSpare) * 8;
    E.fac[j] = T.dr_compact()[(en >> 10) & 511u];
  }
  wg_barrier_lds();
  EG_CS(0);      // 0: set-up

  // ---- phase 1: the script ----
  if (wave == 0) {
    Lists ls{info};
    const bool ok = rs::script(T, S, O, 0u, ls, lane);      // phase 1 of the header: k_rollout's replay path without the placements
    if (!ok && lane == 0) sc.failed = 1;
  }
  wg_barrier_lds();
  EG_CS(1);      // 1: the script
  if (sc.failed) return;      // the script needs a seeded draw, or ends with a status other than EG_EP_OK: the per-episode path runs it

  // ---- phase 2: the placements, in list order ----
  const int n_gens = __builtin_amdgcn_readfirstlane(sc.n_gens);
  const int nv = T.n_variants;
  uint16_t* gen_cell = O.gen_cell(0);
  double base[kVariants][kPer];      // the year's unpenalised scores of the lane's cells, per (radius class, marine) variant
#pragma unroll
  for (int v = 0; v < kVariants; ++v)
#pragma unroll
    for (int k = 0; k < kPer; ++k) base[v][k] = 0.0;
  int cur_year = -1, slow = 0, years = 0;
  bool failed = false;
  // (the list entry and the type's word of a placement are read during the placement before it: two dependent LDS round trips less)
  int pk_next = n_gens > 0 ? (int)sc.gpack[0] : 0;
  int info_next = sc.tinfo[pk_next & 15];
  for (int g = 0; g < n_gens; ++g) {
    const int pk = __builtin_amdgcn_readfirstlane(pk_next);
    const int info_t = __builtin_amdgcn_readfirstlane(info_next);
    const int t = pk & 15, yi = (pk >> 4) & 31;
    pk_next = (int)sc.gpack[g + 1 < n_gens ? g + 1 : g];
    if (yi != cur_year) {      // a new year: its scores (tab::cbase, per cell), one round trip
      cur_year = yi; years += 1;
#pragma unroll
      for (int v = 0; v < kVariants; ++v)
        if (v < nv) {
#pragma unroll
          for (int k = 0; k < kPer; ++k) {
            const int cell = tid + k * kThreads;
            base[v][k] = cell < kCells ? T.cbase()[(size_t)(yi * kMaxVariants + v) * kCells + cell] : 0.0;
          }
        }
      EG_CS(2);      // 2: year changes (requests; the wait lands in the first search)
    }
    const int v = info_t & 15, rc = (info_t >> 4) & 15;
    int cell;
    switch (v) {      // (uniform: each case reads its own registers — no copies)
#define EG_SCAN(V_) case V_: cell = scan_fast(base[V_], rc, tid, lane, wave EG_CS_PASS); break;
      EG_SCAN(1) EG_SCAN(2) EG_SCAN(3) EG_SCAN(4) EG_SCAN(5) EG_SCAN(6) EG_SCAN(7)
      static_assert(kVariants == 8, "a case a variant");
#undef EG_SCAN
      default: cell = scan_fast(base[0], rc, tid, lane, wave EG_CS_PASS); break;
    }
    EG_CS(7);      // 7: searches: the decision
    if (force != 0) cell = -2;
    if (cell == -2) {      // several cells within reach of the maximum, or nothing decided (rare)
      double tb[kPer];
#pragma unroll
      for (int k = 0; k < kPer; ++k) tb[k] = base[0][k];
#pragma unroll
      for (int vv = 1; vv < kVariants; ++vv)
        if (vv == v) { _Pragma("unroll") for (int k = 0; k < kPer; ++k) tb[k] = base[vv][k]; }
      cell = scan_slow(T, tb, yi, t, g, tid, lane, wave, force == 2);
      slow += 1;
    }
    if (cell < 0) { failed = true; break; }      // EG_EP_NO_LOCATION: the per-episode path reports it
    if (tid == 0) { sc.gcell[g] = (uint16_t)cell; gen_cell[g] = (uint16_t)cell; }
    info_next = sc.tinfo[pk_next & 15];      // (pk_next has long arrived)
    field_add(E, cell);
    wg_barrier_lds();
    EG_CS(5);      // 5: field updates + their barrier
  }
  if (!failed && tid == 0) {
    info->lens[5] = slow;
    *O.n_chunks(0) = (uint32_t)(years * ((nv * kCells * 8 + 2047) / 2048) + slow);      // the years' score tables, in units of 2 KB
#ifdef EG_COOP_STAMPS
    for (int i = 0; i < 8; ++i) info->stamps[i] = cs[i];
#else
    info->stamps[7] = (unsigned long long)slow;      // (eg_debug_hoist_stamps: searches that needed more than the one exchange)
#endif
    __threadfence();
    __hip_atomic_store(&info->coop_seq, hoist_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// The yearly rows of the scratch record: a one-wave workgroup a year (a workgroup that needs one wave slot starts the moment any lean
// episode ends — a sixteen-wave workgroup waited half a millisecond for a CU to drain, and held the lean grid's dispatch back while it
// did: profiles/r04_ab_notes.log).  The workgroup that finishes last runs the three running totals through the 26 years, writes the
// record's header, adds the statistics of ALL the batch's replay episodes (`n_replay` identical episodes: every integer sum of
// episode_update_stats times that — 1 638 waves adding to the same few hundred addresses one after the other took 0.3 ms) and
// publishes HoistInfo::served_seq.
__global__ void __launch_bounds__(kWave) k_replay_books(DevTables T, DevSnapshot S_in, DevOut O, unsigned long long hoist_seq, HoistInfo* info,
                                                        long long* stats, uint32_t n_replay, int stats_rep) {
  if (__hip_atomic_load(&info->coop_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != hoist_seq) return;      // the script was not hoisted
  const int lane = threadIdx.x;
  const int yi = blockIdx.x;
  {
    const int G = __builtin_amdgcn_readfirstlane(info->g_end[yi]), NO = __builtin_amdgcn_readfirstlane(info->o_end[yi]);
    const rs::YearRow r = rs::books_year(T, S_in, O, 0u, yi, G, NO, lane, [](const uint16_t* list, int i) { return (int)list[i]; });
    if (lane == 0) {
      double* sy = info->year[yi];
      for (int k = 0; k < 7; ++k) sy[k] = r.v[k];
    }
  }
  // ---- the workgroup that is done last finishes the record ----
  __threadfence();
  int last = 0;
  if (lane == 0) last = atomicAdd(&info->books_done, 1) == kYears - 1 ? 1 : 0;
  if (!__builtin_amdgcn_readfirstlane(last)) return;
  __threadfence();
  if (lane == 0) {      // the running totals (metrics_calculation.rs:133-153) and the record's header
    info->books_done = 0;      // (for the next batch: its kernels are stream-ordered behind this one)
    double total_cost = 0.0, total_credit = 0.0, total_sales = 0.0, last_year[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int y = 0; y < kYears; ++y) {
      double sy[7];
      for (int k = 0; k < 7; ++k) sy[k] = __hip_atomic_load(&info->year[y][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      total_cost = y == 0 ? sy[0] : total_cost + sy[0];
      total_credit = y == 0 ? sy[1] : total_credit + sy[1];
      total_sales = y == 0 ? sy[2] : total_sales + sy[2];
      if (S_in.write_yearly) {
        double* row = O.yearly(0) + y * EG_YEARLY_FIELDS;
        row[EG_Y_TOTAL_COST] = total_cost; row[EG_Y_TOTAL_CREDIT] = total_credit; row[EG_Y_TOTAL_SALES] = total_sales;
      }
      for (int k = 0; k < 7; ++k) last_year[k] = sy[k];
    }
    // SimulationMetrics, iteration.rs:69-74 (Q2: total_cost is the last year's capital cost)
    O.metrics(0)[0] = last_year[3]; O.metrics(0)[1] = last_year[4]; O.metrics(0)[2] = last_year[5];
    O.metrics(0)[3] = last_year[6] >= 0.0 ? 1.0 : 0.0;
    *O.status(0) = EG_EP_OK; *O.n_gens(0) = info->lens[3]; *O.n_offsets(0) = info->lens[4];
    *O.n_draws(0) = 0ull;      // (a replay that needs no fallback draws nothing)
    *O.bytes_moved(0) = (double)info->bytes;
  }
  if (stats != nullptr) {      // the statistics of the batch's n_replay identical replay episodes, at once
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    wave_sync();
    DevSnapshot S = S_in; StatsParams P;
    load_state(S); load_stats_params(S, P);
    episode_update_stats(O, S, P, 0u, lane, stats, (unsigned long long)n_replay, stats_rep ? 0 : -1);
  }
  __threadfence();
  if (lane == 0) __hip_atomic_store(&info->served_seq, hoist_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// The scratch record into the record of every replay episode of the batch (one wave each); the episode's score goes to the batch's
// score list as the statistics epilogue would have left it (the statistics themselves: k_replay_books).
__global__ void __launch_bounds__(kWave) k_replay_broadcast(DevSnapshot S_in, DevOut scratch, DevOut O, uint32_t n_episodes, int with_stats,
                                                            EpisodeMap emap, const HoistInfo* info, unsigned long long hoist_seq) {
  const int lane = threadIdx.x;
  if (blockIdx.x >= emap.count) return;
  if (info->served_seq != hoist_seq) return;      // the script was not hoisted: the per-episode variants have run these episodes
  const uint32_t e = map_episode(emap, blockIdx.x);
  if (e >= n_episodes) return;
  const int32_t* lens = info->lens;
  const uint8_t* src = scratch.base;
  uint8_t* dst = O.base + (size_t)e * rec::stride;
  auto copy = [&](size_t off, size_t bytes) {      // 16 bytes a lane (records and fields start at multiples of 16)
    const size_t lo = off & ~size_t(15), hi = (off + bytes + 15) & ~size_t(15);
    for (size_t i = lo + 16 * (size_t)lane; i < hi; i += 16 * kWave) *reinterpret_cast<uint4*>(dst + i) = *reinterpret_cast<const uint4*>(src + i);
  };
  static_assert(rec::stride % 16 == 0 && rec::yearly % 16 == 0 && rec::run_log % 16 == 0 && rec::gen_cell % 16 == 0, "record layout");
  copy(0, rec::yearly);      // metrics, score, bytes, draws, status, counts, per-year counts
  if (S_in.write_yearly) copy(rec::yearly, 8 * EG_YEARS * EG_YEARLY_FIELDS);
  copy(rec::run_log, (size_t)lens[0]); copy(rec::def_log, (size_t)lens[1]); copy(rec::act_log, (size_t)lens[2]);
  copy(rec::gen_cell, 2 * (size_t)lens[3]); copy(rec::gen_pack, 2 * (size_t)lens[3]); copy(rec::off_pack, 2 * (size_t)lens[4]);
  if (with_stats && lane == 0) O.score_list[e] = scratch.score_list[0];      // (rec::score came with the header)
}
