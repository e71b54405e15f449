// eg_replay_coop.h — the replay episodes of a batch, computed once.  Included by eg_rollout.hip (inside its anonymous namespace,
// eg_rollout.o only): it uses that file's wave primitives, aggregates and statistics epilogue.
//
// An episode that replays the best strategy (iteration.rs:34-42, force_best_actions) takes its actions from the stored lists
// (sampling.rs:78-101 and :242-266 return before any `gen`; simulation.rs:146-162 takes the year's count from the list) and reads
// no seeded draw until a list runs out (the smart fallbacks, sampling.rs:445-528) — so every replay episode of a batch, whatever its
// index, is the SAME computation from the same snapshot, and in the sustained state of configs[2] 1 638 bit-identical copies of one
// serial instruction stream were the batch's long pole (k_rollout<0,2> 2.37 ms beside the lean grid's 1.27 ms).
//
// k_replay_coop runs that one computation as ONE workgroup of four waves — one per SIMD of a CU, each with the SIMD's whole
// register file (the work of a search does not get shorter by spreading it over more waves than there are SIMDs to issue for them:
// every further wave repeats the wave-level reductions) —:
//   * wave 0 is the episode: the year loop, the repair loop and the additional actions of k_rollout's replay path, statement for
//     statement, minus the in-episode weight nudges (learning.rs:21-88, deficit.rs:82-135: they only touch the episode's private copy
//     of the tables, which a replay episode never samples from and which is dropped at its end — Q5);
//   * a placement is an arg-max over all 2 601 candidate cells at once, eleven cells per lane: the product of the penalty factors of
//     every generator placed so far is kept per radius class and cell in LDS (field[6][2624] f64, 126 KB of the CU's 160 KB; updated
//     for the new generator by one lane per (class, di, dj) entry of the host's list), so approx(c) = ((te * cf) * size) * field[c] is
//     three multiplications per cell; te of the year sits in registers.  One exchange through LDS gives the largest approximate score
//     and says whether a second cell comes within 2^-20 of it.  If none does — the usual case — the holder IS the reference's arg-max
//     (exact and approximate score are the same real product rounded at most G + 3 times each: they differ by less than 2^-40
//     relative for the 4 096 generators a list can hold) and only its cell is asked for.  Otherwise every cell within 2^-30 of the
//     exact maximum of the approximate scores is evaluated EXACTLY — te times the factors in list order, a candidate a wave
//     (exact_product_chain's arithmetic) — and the first maximum in cell order wins; subnormal ranges, more than 64 such cells or no
//     positive score fall back to the exact scan of all cells (every lane folds the whole list for its three cells).  The winner is
//     metal_location_search.rs:110-176's, bit for bit, as with place_search / place_heavy;
//   * the year-start sums (map_handler.rs:829-965, in list order) are eight chains — cost, opinion, CO2, three output classes, offset
//     tonnes, offset cost — two per wave, side by side (fold2_row16: one DPP multiply-accumulate per list element and chain).
// The record goes to a scratch slot; k_replay_broadcast copies it into the record of every replay episode of the batch and runs the
// statistics epilogue for each of them (episode_update_stats: the same function, so the update packet is the per-episode path's by
// construction).  If the script cannot be finished without a seeded draw (a fallback), or would end with a status other than
// EG_EP_OK (a capacity), the kernel gives up without publishing anything and the per-episode replay variants — launched behind it
// in any case — run the episodes as before: they return at once only when the hoist word carries their batch's sequence number.
#pragma once

namespace coop {

constexpr int kWaves = 4, kThreads = kWaves * kWave;
constexpr int kPer = (kCells + kThreads - 1) / kThreads;      // candidate cells per lane: 11
constexpr int kEnt = 1024 / kThreads;                         // entries of the field update per lane: 4
constexpr int kDoPlace = 1, kDoYear = 2, kDoExit = 3;
constexpr double kKeepCoop = 1.0 - 0x1p-30;

struct __align__(16) Smem {
  double field[kRadiusClasses * kFieldStride];      // per radius class and cell: product of the penalty factors of every generator so far
  double dr[kDrCompact];                            // d/R by squared cell distance (tab::dr_compact)
  double type_out[16], type_co2[16];
  int tinfo[16];                                    // radius class | marine << 4 | output class << 8 | cap << 16 | first entry << 24 (/2)
  uint16_t gcell[EG_MAX_GENS];                      // the episode's generators: cell
  uint16_t gpack[EG_MAX_GENS];                      //   type | build-year index << 4 | multiplier index << 9 (the record's gen_pack)
  uint16_t opack[EG_MAX_OFFSETS];                   // the episode's offsets (the record's off_pack)
  int cmd[4];                                       // kind, argument (type | year), generators, offsets
  uint32_t top1[kWaves], top2[kWaves];              // high words of the largest / second largest approximate score of each wave
  int cell1[kWaves];                                // ... and the cell of the largest
  double m031[kWaves];                              // ... and that cell's 0.03 * mean settlement opinion
  double chain[8];                                  // year-start sums
  double xscore[kWaves]; int xcell[kWaves]; int xcnt[kWaves];      // slow paths: per-wave maxima / candidate counts
  int cand[kWave]; double cand_score[kWave];
};
static_assert(sizeof(Smem) <= 160 * 1024, "one workgroup owns the CU's LDS");
__shared__ Smem sc;

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

struct Lane {      // what every lane of the workgroup keeps for its three candidate cells and its entry of the field update
  double te[kRadiusClasses][kPer];      // placement prefix of this year (tab::te_cell), 0.0 beyond the grid
  double cf[kPer];                      // coast factor
  double m03[kPer];                     // 0.03 * mean settlement opinion of the cell (tab::m03)
  double fac[kEnt];                     // the lane's entries of the field update: d/R
  int en[kEnt];                         //   di + 16 | (dj + 16) << 5 | class << 19 (tab::hv_lists, the list of all six classes)
};

// The arg-max of metal_location_search.rs:110-176 for generator type t against the ngen generators in sc.gcell; called by all four
// waves with the same arguments, returns the same cell in every lane (-1: no candidate has a positive score, actions.rs:77-89).
// *slow counts the searches that needed more than the one exchange.
__device__ __forceinline__ int place(const DevTables& T, const Lane& L, int tid, int lane, int wave, int yi, int t, int ngen, int& slow, double& m03) {
  const int info = __builtin_amdgcn_readfirstlane(sc.tinfo[t]);
  const int rc = info & 15, off = (int)((unsigned)info >> 24) << 1, cap = (info >> 16) & 255;
  const bool marine = ((info >> 4) & 1) != 0;
  const double size_factor = T.size_factor;
  double tb[kPer];
  switch (rc) {      // (uniform: the year's prefix of this radius class, out of the registers)
#define EG_TE_OF(c_) case c_: { _Pragma("unroll") for (int k = 0; k < kPer; ++k) tb[k] = L.te[c_][k]; } break;
    EG_TE_OF(1) EG_TE_OF(2) EG_TE_OF(3) EG_TE_OF(4) EG_TE_OF(5)
#undef EG_TE_OF
    default: { _Pragma("unroll") for (int k = 0; k < kPer; ++k) tb[k] = L.te[0][k]; } break;
  }
  double ap[kPer]; double cfm[kPer];
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int cell = tid + k * kThreads;
    cfm[k] = marine ? L.cf[k] : 1.0;
    const double base = (tb[k] * cfm[k]) * size_factor;      // (te = 0 beyond the 2 601 cells)
    ap[k] = cell < kCells ? base * sc.field[rc * kFieldStride + (cell < kCells ? cell : 0)] : 0.0;
  }
  // ---- one exchange: high words of each wave's largest and second largest approximate score (scores are not negative: ordered like
  //      their bit patterns; the high word bounds the value within 2^-20) ----
  uint32_t l1 = 0u, l2 = 0u; int lcell = tid; double lm03 = L.m03[0];      // the lane's largest and second largest (a tie counts as two)
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const uint32_t h = (uint32_t)__double2hiint(ap[k]);
    const bool first = h > l1;
    l2 = first ? l1 : (h > l2 ? h : l2);
    l1 = first ? h : l1;
    lcell = first ? tid + k * kThreads : lcell;
    lm03 = first ? L.m03[k] : lm03;
  }
  const uint32_t w1 = wave_max_u32(l1);
  const unsigned long long hold = __ballot(l1 == w1);
  uint32_t w2 = wave_max_u32(l1 == w1 ? l2 : l1);
  if (__popcll(hold) > 1) w2 = w1;
  const int holder = __ffsll((long long)hold) - 1;
  const int wcell = __builtin_amdgcn_readlane(lcell, holder);
  const double wm03 = readlane_f64(lm03, holder);
  if (lane == 0) { sc.top1[wave] = w1; sc.top2[wave] = w2; sc.cell1[wave] = wcell; sc.m031[wave] = wm03; }
  __syncthreads();
  const uint32_t r1 = lane < kWaves ? sc.top1[lane] : 0u, r2 = lane < kWaves ? sc.top2[lane] : 0u;
  const uint32_t mh = wave_max_u32(r1);
  const double m_lo = __hiloint2double((int)mh, 0);      // a lower bound of the largest approximate score, within 2^-20
  bool scan = !(m_lo >= 1e-250);                         // nothing placeable, or subnormal territory: the exact scan decides
  if (!scan) {
    const double thr = m_lo * kKeepCoop;
    const unsigned long long c1 = __ballot(lane < kWaves && __hiloint2double((int)r1, -1) >= thr);
    const unsigned long long c2 = __ballot(lane < kWaves && __hiloint2double((int)r2, -1) >= thr);
    // ONE candidate: it is the arg-max (the arg-max is among the candidates) and only its cell is asked for
    if (__popcll(c1) + __popcll(c2) == 1) { const int w = __ffsll((long long)c1) - 1; m03 = sc.m031[w]; return sc.cell1[w]; }
  }
  slow += 1;
  if (!scan) {
    // ---- several cells within reach of the maximum: the exact maximum M of the approximate scores, then every cell within 2^-30 of
    //      it exactly (place_heavy's steps 2 and 3) ----
    double lm = ap[0];
#pragma unroll
    for (int k = 1; k < kPer; ++k) lm = dmax(lm, ap[k]);
    const double wm = wave_max_f64(lm);
    __syncthreads();      // (everybody has read top1 / top2 / cell1 of this search: the next search may overwrite them... and xscore below)
    if (lane == 0) sc.xscore[wave] = wm;
    __syncthreads();
    const double M = wave_max_f64(lane < kWaves ? sc.xscore[lane] : 0.0);
    const double thr = M * kKeepCoop;
    unsigned long long fm[kPer]; int mine = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) { fm[k] = __ballot(ap[k] >= thr && ap[k] > 0.0); mine += __popcll(fm[k]); }
    if (lane == 0) sc.xcnt[wave] = mine;
    __syncthreads();
    int total = 0, before = 0;
    for (int w = 0; w < kWaves; ++w) { const int c = sc.xcnt[w]; total += c; before += w < wave ? c : 0; }
    total = __builtin_amdgcn_readfirstlane(total);
    if (!(M >= 1e-250) || total == 0 || total > kWave) scan = true;
    else {
      int pos = before;
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        if ((fm[k] >> lane) & 1ull) sc.cand[pos + __popcll(fm[k] & ((1ull << lane) - 1ull))] = tid + k * kThreads;
        pos += __popcll(fm[k]);
      }
      __syncthreads();
      for (int c = wave; c < total; c += kWaves) {      // a candidate a wave
        const int cell = sc.cand[c];
        const double tev = T.te_cell()[(size_t)(yi * kRadiusClasses + rc) * kCells + cell];
        const double cfe = marine ? T.coastf()[cell] : 1.0;
        const double sk = (exact_chain(off, cap, ngen, tev, cell, lane) * cfe) * size_factor;
        if (lane == 0) sc.cand_score[c] = sk;
      }
      __syncthreads();
      const double sl = lane < total ? sc.cand_score[lane] : 0.0;
      const int cl = lane < total ? sc.cand[lane] : kCells;
      const ChunkBest b = chunk_reduce<false>(sl, cl, 0.0);      // highest score, ties to the lowest cell = the first maximum in cell order
      __syncthreads();      // (cand / cand_score are free again)
      if (b.score > 0.0) { m03 = T.m03()[b.cell]; return b.cell; }
      scan = true;
    }
  }
  // ---- the exact scan: every lane folds the whole list, in list order, for each of its cells ----
  {
    double lb = 0.0; int lc = kCells;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int cell = tid + k * kThreads;
      const int ci = cell / kGrid, cj = cell - ci * kGrid;
      double s = tb[k];
      for (int g = 0; g < ngen; ++g) s = s * factor_of(ci, cj, (int)sc.gcell[g], off, cap);
      s = (s * cfm[k]) * size_factor;
      s = cell < kCells ? s : 0.0;
      if (s > lb) { lb = s; lc = cell; }      // (cells ascend with k: the first maximum stays)
    }
    const ChunkBest wb = chunk_reduce<false>(lb, lc, 0.0);
    __syncthreads();
    if (lane == 0) { sc.xscore[wave] = wb.score; sc.xcell[wave] = wb.score > 0.0 ? wb.cell : kCells; }
    __syncthreads();
    const ChunkBest b = chunk_reduce<false>(lane < kWaves ? sc.xscore[lane] : 0.0, lane < kWaves ? sc.xcell[lane] : kCells, 0.0);
    __syncthreads();
    if (b.score > 0.0) m03 = T.m03()[b.cell];
    return b.score > 0.0 ? b.cell : -1;
  }
}

// field[class][cell] *= d/R of a generator at `cell`: the (class, di, dj) entries of the host's list, kEnt per lane (distinct entries
// are distinct cells of their class: no two lanes meet)
__device__ __forceinline__ void field_add(const Lane& L, int cell) {
  const int gi = cell / kGrid, gj = cell - gi * kGrid;
  double v[kEnt]; int at[kEnt]; bool on[kEnt];
#pragma unroll
  for (int j = 0; j < kEnt; ++j) {
    const int ci = gi + (L.en[j] & 31) - 16, cj = gj + ((L.en[j] >> 5) & 31) - 16, rc = (int)((unsigned)L.en[j] >> 19);
    on[j] = (unsigned)ci < (unsigned)kGrid && (unsigned)cj < (unsigned)kGrid && L.fac[j] != 1.0;
    at[j] = on[j] ? rc * kFieldStride + ci * kGrid + cj : 0;
    v[j] = sc.field[at[j]];
  }
#pragma unroll
  for (int j = 0; j < kEnt; ++j) if (on[j]) sc.field[at[j]] = v[j] * L.fac[j];
}

// Year start: this year's placement prefix into the lanes' registers, and the year-start sums (year_fold's, in list order) as eight
// chains on waves 0-7.  arg = year index | carry << 8.
__device__ __forceinline__ void year_work(const DevTables& T, Lane& L, int tid, int lane, int wave, int arg, int ngen, int noff) {
  const int yi = arg & 31;
  const bool carry = ((arg >> 8) & 1) != 0;
#pragma unroll
  for (int rc = 0; rc < kRadiusClasses; ++rc)
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int cell = tid + k * kThreads;
      L.te[rc][k] = cell < kCells ? T.te_cell()[(size_t)(yi * kRadiusClasses + rc) * kCells + cell] : 0.0;
    }
  // chains: 0 capital cost of the generators, 1 opinion total, 2 CO2, 3 / 4 / 5 output of the dispatchable / intermittent / storage
  // class, 6 offset tonnes, 7 offset cost — wave w folds chains 2w and 2w + 1, side by side.  Chains 2-5 only when the
  // existing-plant prefix has changed (otherwise last year's end-of-year sums carry over, bit for bit).
  static_assert(kWaves == 4, "two chains per wave");
  const bool gens = wave < 3;
  int n = gens ? ngen : noff;
  if ((wave == 1 || wave == 2) && carry) n = 0;
  double acc0 = 0.0, acc1 = 0.0;
  if (wave == 0) acc1 = T.pre_optot()[yi];
  else if (wave == 1) { acc0 = T.pre_co2()[yi]; acc1 = T.pre_tg()[yi]; }
  else if (wave == 2) { acc0 = T.pre_ig()[yi]; acc1 = T.pre_sg()[yi]; }
  acc0 = uniform_f64(acc0); acc1 = uniform_f64(acc1);
  const double* ccy = T.cc() + (unsigned)yi * kTypes * kYears * kMults * 2;
  // the two chains' terms of list element i (+0.0 beyond the list: sums of non-negative terms)
  auto terms = [&](int i, double& x0, double& x1) {
    x0 = 0.0; x1 = 0.0;
    if (i >= n) return;
    if (gens) {
      const int pk = sc.gpack[i], ty = pk & 15, b = (pk >> 4) & 31, m = pk >> 9;
      if (wave == 0) {
        const double2 cc = *reinterpret_cast<const double2*>(ccy + ((unsigned)(ty * kYears + b) * kMults + m) * 2);
        x0 = cc.x;
        x1 = (T.m03()[sc.gcell[i]] + T.t12()[(unsigned)yi * kTypes + ty]) + cc.y;
        return;
      }
      const int cls = (sc.tinfo[ty] >> 8) & 3;
      const double out = sc.type_out[ty];
      if (wave == 1) { x0 = sc.type_co2[ty]; x1 = (cls != 1 && cls != 2) ? out : 0.0; }
      else { x0 = cls == 1 ? out : 0.0; x1 = cls == 2 ? out : 0.0; }
      return;
    }
    const int p = sc.opack[i], ot = p & 15, b = (p >> 4) & 31, m = p >> 9;
    x0 = T.offv()[((unsigned)yi * kOffsetTypes + ot) * kYears + b];
    x1 = T.offc()[((unsigned)yi * kOffsetTypes + ot) * kMults + m];
  };
  double x0, x1;
  terms(lane, x0, x1);
  for (int base = 0; base < n; base += kWave) {
    const double c0 = x0, c1 = x1;
    terms(base + kWave + lane, x0, x1);      // the next block's terms are requested before this block is folded
    const int cnt = n - base < kWave ? n - base : kWave;
    for (int r = 0; r * 16 < cnt; ++r) {      // sixteen list elements per step, in list order
      double a0 = acc0, a1 = acc1;
      fold2_row16(a0, c0, a1, c1);
      acc0 = readlane_f64(a0, 16 * r); acc1 = readlane_f64(a1, 16 * r);
    }
  }
  if (lane == 0) { sc.chain[2 * wave] = acc0; sc.chain[2 * wave + 1] = acc1; }
}

}  // namespace coop

// One workgroup: the batch's replay script, once.  `O`: the scratch record (episode 0 of it); `hoist`: {u64 word, i32 lengths[5]}.
__global__ void __launch_bounds__(coop::kThreads, 1) k_replay_coop(DevTables T, DevSnapshot S_in, DevOut O, unsigned long long hoist_seq,
                                                                 unsigned long long* hoist) {
  using namespace coop;
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  DevSnapshot S = S_in;
  load_state(S);
  if (!S.has_best_actions) return;      // no best strategy on the device: the flagged episodes are ordinary ones (k_rollout decides per episode)
  // ---- tables -> LDS, the lane's constants -> registers ----
  for (int i = tid; i < kRadiusClasses * kFieldStride; i += kThreads) sc.field[i] = 1.0;
  for (int i = tid; i < kDrCompact; i += kThreads) sc.dr[i] = T.dr_compact()[i];
  if (tid < kTypes) {
    const int rc = T.rclass()[tid];
    sc.tinfo[tid] = rc | ((T.marine()[tid] ? 1 : 0) << 4) | (T.cls()[tid] << 8) | (T.dr_meta()[8 + rc] << 16) | (int)((unsigned)(T.dr_meta()[rc] >> 1) << 24);
    sc.type_out[tid] = T.out_mw()[tid];
    sc.type_co2[tid] = T.co2_t()[tid];
  }
  Lane L;
#pragma unroll
  for (int k = 0; k < kPer; ++k) { const int cell = tid + k * kThreads; L.cf[k] = cell < kCells ? T.coastf()[cell] : 1.0; L.m03[k] = cell < kCells ? T.m03()[cell] : 0.0; }
#pragma unroll
  for (int j = 0; j < kEnt; ++j) {      // the list of all six classes: entry = di + 16 | (dj + 16) << 5 | place in dr << 10 | class << 19
    L.en[j] = (int)T.hv_lists()[63 * 1024 + j * kThreads + tid];
    L.fac[j] = T.dr_compact()[((unsigned)L.en[j] >> 10) & 511u];
  }
#pragma unroll
  for (int rc = 0; rc < kRadiusClasses; ++rc)
#pragma unroll
    for (int k = 0; k < kPer; ++k) L.te[rc][k] = 0.0;
  __syncthreads();

  int slow = 0;
  if (wave != 0) {      // ---- waves 1-3: serve wave 0's commands ----
    for (;;) {
      __syncthreads();
      const int kind = __builtin_amdgcn_readfirstlane(sc.cmd[0]), arg = __builtin_amdgcn_readfirstlane(sc.cmd[1]);
      const int ngen = __builtin_amdgcn_readfirstlane(sc.cmd[2]), noff = __builtin_amdgcn_readfirstlane(sc.cmd[3]);
      if (kind == kDoExit) return;
      if (kind == kDoYear) { year_work(T, L, tid, lane, wave, arg, ngen, noff); __syncthreads(); continue; }
      double m03_unused;
      const int cell = place(T, L, tid, lane, wave, arg >> 8, arg & 255, ngen, slow, m03_unused);
      if (cell >= 0) field_add(L, cell);
    }
  }

  // ---- wave 0: the episode (k_rollout's replay path; every value below is the same in all 64 lanes) ----
  auto issue = [&](int kind, int arg, int ngen, int noff) {
    if (lane == 0) { sc.cmd[0] = kind; sc.cmd[1] = arg; sc.cmd[2] = ngen; sc.cmd[3] = noff; }
    __syncthreads();
  };
  const int n_existing = T.n_existing;
  const uint32_t carry_mask = (uint32_t)__ballot(lane > 0 && lane < EG_YEARS && T.pre_co2()[lane] == T.pre_co2()[lane - 1] &&
                                                 T.pre_tg()[lane] == T.pre_tg()[lane - 1] && T.pre_ig()[lane] == T.pre_ig()[lane - 1] &&
                                                 T.pre_sg()[lane] == T.pre_sg()[lane - 1]);
  // which per-episode variant would run these episodes decides the capacities (k_rollout: kGenCap / kOffCap)
  const bool long_list = S.best_off()[EG_YEARS] > kShortReplayMax;
  const int gen_cap = long_list ? EG_MAX_GENS : kLdsGens, off_cap = long_list ? EG_MAX_OFFSETS : kLdsGens;
  uint8_t* run_log = O.run_log(0); uint8_t* def_log = O.def_log(0); uint8_t* act_log = O.act_log(0);
  uint16_t* gen_cell = O.gen_cell(0); uint16_t* gen_pack = O.gen_pack(0); uint16_t* off_pack = O.off_pack(0);
  int ngen = 0, noff = 0, run_pos = 0, def_pos = 0, act_pos = 0, chunks = 0;
  unsigned long long bytes = 32ull;
  double tot_cost = 0.0, tot_credit = 0.0, tot_sales = 0.0, last_net = 0.0, last_opinion = 0.0, last_capital = 0.0, last_balance = 0.0;
  double yend[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  bool fail = false;      // the script needs a seeded draw, or ends with a status other than EG_EP_OK: the per-episode path runs it

  for (int yi = 0; yi < kYears && !fail; ++yi) {
    const int year = 2025 + yi;
    const bool carry = ((carry_mask >> yi) & 1u) != 0u;
    issue(kDoYear, yi | ((carry ? 1 : 0) << 8), ngen, noff);
    year_work(T, L, tid, lane, wave, yi | ((carry ? 1 : 0) << 8), ngen, noff);
    chunks += (kRadiusClasses * kCells * 8 + 2047) / 2048;      // this year's placement prefix, in units of 2 KB
    __syncthreads();
    Agg a;
    a.usage = T.usage()[yi];
    a.gcost_prev = yend[0]; a.ocost_prev = yend[1];
    a.gcost = sc.chain[0]; a.optot = sc.chain[1]; a.offs = sc.chain[6]; a.ocost = sc.chain[7];
    a.opcnt = T.pre_opcnt()[yi] + ngen;
    if (carry) { a.co2 = yend[2]; a.tg = yend[3]; a.ig = yend[4]; a.sg = yend[5]; }
    else { a.co2 = sc.chain[2]; a.tg = sc.chain[3]; a.ig = sc.chain[4]; a.sg = sc.chain[5]; }
    bytes += 2ull * (unsigned long long)(n_existing + ngen) * 56ull + 2ull * (unsigned long long)noff * 8ull + 184ull;

    const int rep_lo = S.best_off()[yi], rep_n = S.best_off()[yi + 1] - rep_lo;
    const int repd_lo = S.bestd_off()[yi], repd_n = S.bestd_off()[yi + 1] - repd_lo;
    // this year's lists, the first 128 / 64 entries one per lane (an action is then a v_readlane, not a memory round trip)
    const int rep0 = lane < rep_n ? (int)S.best_actions()[rep_lo + lane] : 0;
    const int rep1 = kWave + lane < rep_n ? (int)S.best_actions()[rep_lo + kWave + lane] : 0;
    const int repd0 = lane < repd_n ? (int)S.bestd_actions()[repd_lo + lane] : 0;
    int replay_idx = 0, replay_def_idx = 0, n_run_y = 0, n_def_y = 0, n_act_y = 0;
    const State year_start = state_of(a);
    int phase = year_start.balance < 0.0 ? 0 : 1;
    double remaining = -year_start.balance;
    uint32_t attempts = 0, n_add = 0, k_add = 0;
    bool n_add_known = false;

    for (int guard = 0; guard < 200000; ++guard) {
      int action;
      if (phase == 0) {      // simulation.rs:319-522
        if (!(remaining > 0.0)) { phase = 1; continue; }      // (the success bonus only touches the episode's private weights)
        attempts += 1;
        if (attempts < 5u) {      // sampling.rs:242-313
          if (replay_def_idx >= repd_n) { fail = true; break; }      // smart_deficit_fallback: a seeded draw
          action = replay_def_idx < kWave ? __builtin_amdgcn_readlane(repd0, replay_def_idx) : (int)S.bestd_actions()[repd_lo + replay_def_idx];
          replay_def_idx += 1;
          if (def_pos >= EG_DEF_CAP || n_def_y >= 128) { fail = true; break; }
          if (lane == 0) def_log[def_pos] = (uint8_t)action;
          def_pos += 1; n_def_y += 1;
        } else action = 3 * kBattery;      // simulation.rs:369-376
        if (action >= kFirstOffset) continue;
      } else {
        if (!n_add_known) { n_add_known = true; n_add = (uint32_t)rep_n; }      // simulation.rs:146-162
        if (k_add >= n_add) break;
        k_add += 1;
        // sampling.rs:78-145 (replay_idx < rep_n: the count IS the list's length)
        action = replay_idx < kWave ? __builtin_amdgcn_readlane(rep0, replay_idx)
                                    : (replay_idx < 2 * kWave ? __builtin_amdgcn_readlane(rep1, replay_idx - kWave) : (int)S.best_actions()[rep_lo + replay_idx]);
        replay_idx += 1;
        if (run_pos >= EG_RUN_CAP) { fail = true; break; }
        if (lane == 0) run_log[run_pos] = (uint8_t)action;
        run_pos += 1; n_run_y += 1;
      }
      action = __builtin_amdgcn_readfirstlane(action);
      if (action < kFirstOffset) {      // actions.rs:42-91
        const int t = action / 3, m = action - 3 * t;
        bytes += (unsigned long long)kCells * 8ull + (unsigned long long)(n_existing + ngen) * 16ull;
        const double2 ccv = *reinterpret_cast<const double2*>(T.cc() + ((((unsigned)yi * kTypes + t) * kYears + yi) * kMults + m) * 2);
        double cc_prev = 0.0;
        if (yi > 0) cc_prev = T.cc()[((((unsigned)(yi - 1) * kTypes + t) * kYears + yi) * kMults + m) * 2];
        const double t12v = T.t12()[(unsigned)yi * kTypes + t];
        issue(kDoPlace, t | (yi << 8), ngen, noff);
        double m03v = 0.0;
        const int cell = place(T, L, tid, lane, wave, yi, t, ngen, slow, m03v);
        if (cell < 0) { fail = true; break; }      // EG_EP_NO_LOCATION
        field_add(L, cell);
        if (ngen >= gen_cap) { fail = true; break; }      // EG_EP_OVERFLOW
        if (lane == 0) {
          const uint16_t pk = (uint16_t)(t | (yi << 4) | (m << 9));
          sc.gcell[ngen] = (uint16_t)cell; sc.gpack[ngen] = pk;
          gen_cell[ngen] = (uint16_t)cell; gen_pack[ngen] = pk;
        }
        ngen += 1;
        a.gcost += ccv.x;
        if (yi > 0) a.gcost_prev += cc_prev;
        a.co2 += sc.type_co2[t];
        const double out = sc.type_out[t];
        const int cls = (sc.tinfo[t] >> 8) & 3;
        if (cls == 1) a.ig += out; else if (cls == 2) a.sg += out; else a.tg += out;
        a.optot += (m03v + t12v) + ccv.y;
        a.opcnt += 1;
      } else if (action < kFirstOther) {      // actions.rs:121-181
        const int ot = (action - kFirstOffset) / 3, m = (action - kFirstOffset) - 3 * ot;
        if (noff >= off_cap) { fail = true; break; }
        const uint16_t p = (uint16_t)(ot | (yi << 4) | (m << 9));
        if (lane == 0) { sc.opack[noff] = p; off_pack[noff] = p; }
        noff += 1;
        a.offs += T.offv()[((unsigned)yi * kOffsetTypes + ot) * kYears + yi];
        a.ocost += T.offc()[((unsigned)yi * kOffsetTypes + ot) * kMults + m];
        if (yi > 0) a.ocost_prev += T.offc()[((unsigned)(yi - 1) * kOffsetTypes + ot) * kMults + m];
      }
      if (phase == 0) {      // simulation.rs:406-486 (record twice, Q15; the nudges are dropped with the private tables)
        if (def_pos >= EG_DEF_CAP || n_def_y >= 128 || run_pos >= EG_RUN_CAP) { fail = true; break; }
        if (lane == 0) { def_log[def_pos] = (uint8_t)action; run_log[run_pos] = (uint8_t)action; }
        def_pos += 1; n_def_y += 1; run_pos += 1; n_run_y += 1;
        const State nxt = state_of(a);
        remaining = -dmin(nxt.balance, 0.0);
      } else {               // simulation.rs:193-197
        if (act_pos >= EG_ACT_CAP || run_pos >= EG_RUN_CAP) { fail = true; break; }
        if (lane == 0) { act_log[act_pos] = (uint8_t)action; run_log[run_pos] = (uint8_t)action; }
        act_pos += 1; n_act_y += 1; run_pos += 1; n_run_y += 1;
      }
    }
    if (fail) break;
    bytes += 2ull * (unsigned long long)(n_act_y + n_def_y);

    // ---- yearly metrics (metrics_calculation.rs:32-175), as k_rollout writes them ----
    const State s = state_of(a);
    const double gen = (a.tg + a.ig) + a.sg;
    const double credit = s.net >= 0.0 ? 0.0 : (-s.net) * T.carbon_price()[yi];
    const double total_capital = a.gcost + a.ocost;
    const double yearly_capital = yi == 0 ? total_capital : total_capital - (a.gcost_prev + a.ocost_prev);
    double sales = 0.0;
    if (S.enable_energy_sales && s.balance > 0.0) { const double gwh = s.balance * 8.76; sales = gwh * 50000.0; }
    const double yearly_total = yearly_capital + 0.0 + 0.0 - credit - (S.enable_energy_sales ? sales : 0.0);
    const double total_cost = yi == 0 ? yearly_total : tot_cost + yearly_total;
    const double total_credit = yi == 0 ? credit : tot_credit + credit;
    const double total_sales = yi == 0 ? sales : tot_sales + sales;
    tot_cost = total_cost; tot_credit = total_credit; tot_sales = total_sales;
    last_net = s.net; last_opinion = s.opinion; last_capital = total_capital; last_balance = s.balance;
    if (S.write_yearly && lane == 0) {
      double* row = O.yearly(0) + yi * EG_YEARLY_FIELDS;
      row[EG_Y_YEAR] = (double)year; row[EG_Y_POP] = T.population()[yi]; row[EG_Y_USAGE] = a.usage; row[EG_Y_GEN] = gen;
      row[EG_Y_BALANCE] = s.balance; row[EG_Y_OPINION] = s.opinion; row[EG_Y_YEARLY_CAPITAL] = yearly_capital;
      row[EG_Y_TOTAL_CAPITAL] = total_capital; row[EG_Y_INFLATION] = T.inflation()[yi]; row[EG_Y_CO2] = a.co2;
      row[EG_Y_OFFSET] = a.offs; row[EG_Y_NET_CO2] = s.net; row[EG_Y_YEARLY_CREDIT] = credit; row[EG_Y_TOTAL_CREDIT] = total_credit;
      row[EG_Y_YEARLY_SALES] = sales; row[EG_Y_TOTAL_SALES] = total_sales; row[EG_Y_ACTIVE_GENS] = (double)a.opcnt;
      row[EG_Y_UPGRADE_COSTS] = 0.0; row[EG_Y_CLOSURE_COSTS] = 0.0;
      row[EG_Y_YEARLY_TOTAL_COST] = yearly_total; row[EG_Y_TOTAL_COST] = total_cost;
    }
    if (lane == 0) { O.n_run(0)[yi] = n_run_y; O.n_def(0)[yi] = n_def_y; O.n_act(0)[yi] = n_act_y; }
    yend[0] = a.gcost; yend[1] = a.ocost; yend[2] = a.co2; yend[3] = a.tg; yend[4] = a.ig; yend[5] = a.sg;
  }

  if (!fail && lane == 0) {      // SimulationMetrics, iteration.rs:69-74 (Q2)
    O.metrics(0)[0] = last_net; O.metrics(0)[1] = last_opinion; O.metrics(0)[2] = last_capital; O.metrics(0)[3] = last_balance >= 0.0 ? 1.0 : 0.0;
    *O.status(0) = EG_EP_OK; *O.n_gens(0) = ngen; *O.n_offsets(0) = noff;
    *O.n_draws(0) = 0ull;      // (a replay that needs no fallback draws nothing)
    *O.bytes_moved(0) = (double)bytes;
    *O.n_chunks(0) = (uint32_t)(chunks + slow);
    int* lens = reinterpret_cast<int*>(hoist + 1);
    lens[0] = run_pos; lens[1] = def_pos; lens[2] = act_pos; lens[3] = ngen; lens[4] = noff;
    __threadfence();
    __hip_atomic_store(hoist, hoist_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);      // this batch's replay episodes are served
  }
  issue(kDoExit, 0, 0, 0);
}

// The scratch record into the record of every replay episode of the batch (one wave each), then that episode's statistics.
__global__ void __launch_bounds__(kWave) k_replay_broadcast(DevSnapshot S_in, DevOut scratch, DevOut O, uint32_t n_episodes, long long* stats,
                                                            EpisodeMap emap, const unsigned long long* hoist, unsigned long long hoist_seq) {
  const int lane = threadIdx.x;
  if (blockIdx.x >= emap.count) return;
  if (*hoist != hoist_seq) return;      // the script was not hoisted: the per-episode variants have run these episodes
  const uint32_t e = map_episode(emap, blockIdx.x);
  if (e >= n_episodes) return;
  const int* lens = reinterpret_cast<const int*>(hoist + 1);
  const uint8_t* src = scratch.base;
  uint8_t* dst = O.base + (size_t)e * rec::stride;
  auto copy = [&](size_t off, size_t bytes) {      // 16 bytes a lane (records and fields start at multiples of 16 or are copied from one)
    const size_t lo = off & ~size_t(15), hi = (off + bytes + 15) & ~size_t(15);
    for (size_t i = lo + 16 * (size_t)lane; i < hi; i += 16 * kWave) *reinterpret_cast<uint4*>(dst + i) = *reinterpret_cast<const uint4*>(src + i);
  };
  static_assert(rec::stride % 16 == 0 && rec::yearly % 8 == 0, "record layout");
  copy(0, rec::yearly);      // metrics, score, bytes, draws, status, counts, per-year counts
  if (S_in.write_yearly) copy(rec::yearly, 8 * EG_YEARS * EG_YEARLY_FIELDS);
  copy(rec::run_log, (size_t)lens[0]); copy(rec::def_log, (size_t)lens[1]); copy(rec::act_log, (size_t)lens[2]);
  copy(rec::gen_cell, 2 * (size_t)lens[3]); copy(rec::gen_pack, 2 * (size_t)lens[3]); copy(rec::off_pack, 2 * (size_t)lens[4]);
  if (stats != nullptr) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    wave_sync();
    DevSnapshot S = S_in; StatsParams P;
    load_state(S); load_stats_params(S, P);
    episode_update_stats(O, S, P, e, lane, stats);
  }
}
