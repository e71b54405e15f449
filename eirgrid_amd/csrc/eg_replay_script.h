// eg_replay_script.h — what a replay episode does apart from placing its generators: the script (which actions it takes, every log
// and count) and the yearly rows.  Included by eg_rollout.hip inside its anonymous namespace, in both of its objects: the replay hoist
// (eg_replay_coop.h: the batch's replay episodes computed once) and the per-episode replay kernel (eg_replay_solo.h: every replay
// episode on its own wave) run the same statements over their own storage.
//
// Why an episode can be taken apart like this (profiles/r04_ab_notes.log has the measurements that led here): a replay takes its
// actions from the stored lists (sampling.rs:78-101 and :242-266 return before any `gen`; simulation.rs:146-162 takes the year's count
// from the list) and reads no seeded draw until a list runs out (the smart fallbacks, sampling.rs:445-528).  Which actions it takes —
// the repair loop's trips, the forced batteries, the year's additional actions — depends on the lists and on the power balance only,
// and the balance only on the TYPES of the generators placed (map_handler.rs:829-868), never on where they land: the script can be
// expanded ahead of the placements.  And every aggregate of a year is a sum over the generators / offsets in list order
// (map_handler.rs:829-965), started from the existing-plant prefix and continued with every addition — one sequential sum over the
// list as it stands at the year's end —, so the yearly rows can be folded behind the placements, the years independent of each other.
//
// `L` is the caller's storage:
//   int    gpack_at(int i)            generator i's word (type | build-year index << 4 | multiplier index << 9), as the script stored it
//   void   gpack_put(int at, int pk)  the caller's own copy (the store to the record is the script's)
//   void   opack_put(int at, int pp)
//   int    cls(int t)                 output class of generator type t (1 intermittent, 2 storage, otherwise dispatchable)
//   double out(int t)                 output of an operational new plant of type t
//   void   year_end(int yi, int ngen, int noff)                 lane 0: the lists' lengths at the end of year yi
//   void   finish(int run, int def, int act, int ngen, int noff, unsigned long long bytes)      lane 0
#pragma once

namespace rs {

// tg / ig / sg (output of the dispatchable / intermittent / storage class, map_handler.rs:829-868) of the first n generators, added in
// list order onto the existing-plant prefix: year_fold's class sums (one wave; +0.0 for the other classes: sums of non-negative terms)
template <class L>
__device__ __forceinline__ void class_sums(L& ls, int n, int lane, double& tg, double& ig, double& sg) {
  for (int base = 0; base < n; base += kWave) {
    double x0 = 0.0, x1 = 0.0, x2 = 0.0;
    if (base + lane < n) {
      const int ty = ls.gpack_at(base + lane) & 15;
      const int cls = ls.cls(ty);
      const double out = ls.out(ty);
      x0 = (cls != 1 && cls != 2) ? out : 0.0; x1 = cls == 1 ? out : 0.0; x2 = cls == 2 ? out : 0.0;
    }
    const int cnt = n - base < kWave ? n - base : kWave;
    for (int r = 0; r * 16 < cnt; ++r) {
      double a0 = tg, a1 = ig;
      fold2_row16(a0, x0, a1, x1);
      const double a2 = fold_row16(sg, x2);
      tg = readlane_f64(a0, 16 * r); ig = readlane_f64(a1, 16 * r); sg = readlane_f64(a2, 16 * r);
    }
  }
}

// ---- the script of episode `e` of `O` (k_rollout's replay path without the placements; one wave, every value the same in all 64
//      lanes): statement for statement, minus the placement, the in-episode weight nudges (learning.rs:21-88, deficit.rs:82-135: they
//      only touch the episode's private copy of the tables, which a replay never samples from and which is dropped at its end — Q5) and
//      every aggregate but the three output class sums.  Leaves logs, per-year counts and the lists' pack words in the record, the
//      lengths with `ls`; returns false when the classic per-episode path must run the episode: a seeded draw is needed (a fallback), or
//      a capacity would end it with a status other than EG_EP_OK. ----
template <class L>
__device__ __forceinline__ bool script(const DevTables& T, const DevSnapshot& S, const DevOut& O, uint32_t e, L& ls, int lane) {
  const int n_existing = T.n_existing;
  const uint32_t carry_mask = (uint32_t)__ballot(lane > 0 && lane < EG_YEARS && T.pre_co2()[lane] == T.pre_co2()[lane - 1] &&
                                                 T.pre_tg()[lane] == T.pre_tg()[lane - 1] && T.pre_ig()[lane] == T.pre_ig()[lane - 1] &&
                                                 T.pre_sg()[lane] == T.pre_sg()[lane - 1]);
  // the world's yearly scalars and the lists' offsets, a year a lane
  const int ly = lane < EG_YEARS ? lane : 0;
  const double usage_l = T.usage()[ly], ptg_l = T.pre_tg()[ly], pig_l = T.pre_ig()[ly], psg_l = T.pre_sg()[ly];
  const int boff_l = lane <= EG_YEARS ? S.best_off()[lane] : 0, bdoff_l = lane <= EG_YEARS ? S.bestd_off()[lane] : 0;
  // which per-episode variant would run these episodes decides the capacities (k_rollout: kGenCap / kOffCap)
  const bool long_list = __builtin_amdgcn_readlane(boff_l, EG_YEARS) > kShortReplayMax;
  const int gen_cap = long_list ? EG_MAX_GENS : kLdsGens, off_cap = long_list ? EG_MAX_OFFSETS : kLdsGens;
  uint8_t* run_log = O.run_log(e); uint8_t* def_log = O.def_log(e); uint8_t* act_log = O.act_log(e);
  uint16_t* gen_pack = O.gen_pack(e); uint16_t* off_pack = O.off_pack(e);
  int ngen = 0, noff = 0, run_pos = 0, def_pos = 0, act_pos = 0;
  unsigned long long bytes = 32ull;
  double tg = 0.0, ig = 0.0, sg = 0.0;
  // this year's lists, the first 128 / 64 entries one per lane (an action is then a v_readlane); next year's are requested a year ahead
  auto load_lists = [&](int yi, int& r0, int& r1, int& d0) {
    const int lo = __builtin_amdgcn_readlane(boff_l, yi), n = __builtin_amdgcn_readlane(boff_l, yi + 1) - lo;
    const int dlo = __builtin_amdgcn_readlane(bdoff_l, yi), dn = __builtin_amdgcn_readlane(bdoff_l, yi + 1) - dlo;
    r0 = lane < n ? (int)S.best_actions()[lo + lane] : 0;
    r1 = kWave + lane < n ? (int)S.best_actions()[lo + kWave + lane] : 0;
    d0 = lane < dn ? (int)S.bestd_actions()[dlo + lane] : 0;
  };
  int nrep0, nrep1, nrepd0;
  load_lists(0, nrep0, nrep1, nrepd0);

  for (int yi = 0; yi < kYears; ++yi) {
    const bool carry = ((carry_mask >> yi) & 1u) != 0u;
    const int rep0 = nrep0, rep1 = nrep1, repd0 = nrepd0;
    if (yi + 1 < kYears) load_lists(yi + 1, nrep0, nrep1, nrepd0);
    const int rep_lo = __builtin_amdgcn_readlane(boff_l, yi), rep_n = __builtin_amdgcn_readlane(boff_l, yi + 1) - rep_lo;
    const int repd_lo = __builtin_amdgcn_readlane(bdoff_l, yi), repd_n = __builtin_amdgcn_readlane(bdoff_l, yi + 1) - repd_lo;
    const double usage = readlane_f64(usage_l, yi);
    if (!carry) {      // the existing-plant prefix has changed: the class sums are folded again (otherwise last year's carry over, bit for bit)
      tg = readlane_f64(ptg_l, yi); ig = readlane_f64(pig_l, yi); sg = readlane_f64(psg_l, yi);
      class_sums(ls, ngen, lane, tg, ig, sg);
    }
    bytes += 2ull * (unsigned long long)(n_existing + ngen) * 56ull + 2ull * (unsigned long long)noff * 8ull + 184ull;
    int replay_def_idx = 0, n_run_y = 0, n_def_y = 0, n_act_y = 0;
    const double balance0 = ((tg + ig) + sg) - usage;      // state_of(a).balance at the start of the year
    // ---- the repair loop (simulation.rs:319-522), one action at a time: every trip depends on the balance the last one left ----
    if (balance0 < 0.0) {
      double remaining = -balance0;
      uint32_t attempts = 0;
      for (int guard = 0; guard < 200000 && remaining > 0.0; ++guard) {      // (the success bonus behind it only touches the episode's private weights)
        int action;
        attempts += 1;
        if (attempts < 5u) {      // sampling.rs:242-313
          if (replay_def_idx >= repd_n) return false;      // smart_deficit_fallback: a seeded draw
          action = replay_def_idx < kWave ? __builtin_amdgcn_readlane(repd0, replay_def_idx) : (int)S.bestd_actions()[repd_lo + replay_def_idx];
          replay_def_idx += 1;
          if (def_pos >= EG_DEF_CAP || n_def_y >= 128) return false;
          if (lane == 0) def_log[def_pos] = (uint8_t)action;
          def_pos += 1; n_def_y += 1;
        } else action = 3 * kBattery;      // simulation.rs:369-376
        action = __builtin_amdgcn_readfirstlane(action);
        if (action >= kFirstOffset) continue;      // only AddGenerator actions are applied in the repair loop (:398)
        const int t = action / 3, m = action - 3 * t;      // actions.rs:42-91: the placement itself comes later
        bytes += (unsigned long long)kCells * 8ull + (unsigned long long)(n_existing + ngen) * 16ull;
        if (ngen >= gen_cap) return false;      // EG_EP_OVERFLOW
        if (lane == 0) { const int pk = t | (yi << 4) | (m << 9); ls.gpack_put(ngen, pk); gen_pack[ngen] = (uint16_t)pk; }
        ngen += 1;
        const double out = ls.out(t);
        const int cls = ls.cls(t);
        if (cls == 1) ig += out; else if (cls == 2) sg += out; else tg += out;
        // simulation.rs:406-486 (recorded twice, Q15; the nudges are dropped with the private tables)
        if (def_pos >= EG_DEF_CAP || n_def_y >= 128 || run_pos >= EG_RUN_CAP) return false;
        if (lane == 0) { def_log[def_pos] = (uint8_t)action; run_log[run_pos] = (uint8_t)action; }
        def_pos += 1; n_def_y += 1; run_pos += 1; n_run_y += 1;
        const double balance = ((tg + ig) + sg) - usage;
        remaining = -dmin(balance, 0.0);
      }
    }
    // ---- the year's additional actions (simulation.rs:144-198): the whole list (its length IS the count, :146-162; sampling.rs:78-145
    //      hands out entry after entry), 64 actions at a time — a lane an action: logs, list entries and counts in parallel, the class
    //      sums folded in list order ----
    for (int done = 0; done < rep_n; done += kWave) {
      const int i = done + lane, cnt = rep_n - done < kWave ? rep_n - done : kWave;
      const bool valid = i < rep_n;
      int a = done == 0 ? rep0 : rep1;
      if (done >= 2 * kWave) a = valid ? (int)S.best_actions()[rep_lo + i] : 0;
      const bool isg = valid && a < kFirstOffset, iso = valid && a >= kFirstOffset && a < kFirstOther;
      const unsigned long long mg = __ballot(isg), mo = __ballot(iso), below = (1ull << lane) - 1ull;
      const int ng = __popcll(mg), no = __popcll(mo);
      // (what the action-by-action loop checks before every store, for the whole block: any of them ends the script)
      if (ngen + ng > gen_cap || noff + no > off_cap || run_pos + 2 * cnt > EG_RUN_CAP || act_pos + cnt > EG_ACT_CAP) return false;
      if (valid) {      // recorded by the sampler and again by the caller (Q15), and as an action of the result
        run_log[run_pos + 2 * lane] = (uint8_t)a; run_log[run_pos + 2 * lane + 1] = (uint8_t)a;
        act_log[act_pos + lane] = (uint8_t)a;
      }
      double x0 = 0.0, x1 = 0.0, x2 = 0.0;
      if (isg) {
        const int t = a / 3, m = a - 3 * t, at = ngen + __popcll(mg & below);
        const int pk = t | (yi << 4) | (m << 9);
        ls.gpack_put(at, pk); gen_pack[at] = (uint16_t)pk;
        const int cls = ls.cls(t);
        const double out = ls.out(t);
        x0 = (cls != 1 && cls != 2) ? out : 0.0; x1 = cls == 1 ? out : 0.0; x2 = cls == 2 ? out : 0.0;
      }
      if (iso) {
        const int ot = (a - kFirstOffset) / 3, m = (a - kFirstOffset) - 3 * ot, at = noff + __popcll(mo & below);
        const int pp = ot | (yi << 4) | (m << 9);
        ls.opack_put(at, pp); off_pack[at] = (uint16_t)pp;
      }
      // Σ over the block's generators of kCells * 8 + (n_existing + generators before it) * 16
      bytes += (unsigned long long)ng * ((unsigned long long)kCells * 8ull + (unsigned long long)(n_existing + ngen) * 16ull) +
               8ull * (unsigned long long)ng * (unsigned long long)(ng > 0 ? ng - 1 : 0);
      if (ng > 0)
        for (int r = 0; r * 16 < cnt; ++r) {      // (+0.0 from every lane that is not a generator of that class: sums of non-negative terms)
          double a0 = tg, a1 = ig;
          fold2_row16(a0, x0, a1, x1);
          const double a2 = fold_row16(sg, x2);
          tg = readlane_f64(a0, 16 * r); ig = readlane_f64(a1, 16 * r); sg = readlane_f64(a2, 16 * r);
        }
      ngen += ng; noff += no; run_pos += 2 * cnt; n_run_y += 2 * cnt; act_pos += cnt; n_act_y += cnt;
    }
    bytes += 2ull * (unsigned long long)(n_act_y + n_def_y);
    if (lane == 0) {
      O.n_run(e)[yi] = n_run_y; O.n_def(e)[yi] = n_def_y; O.n_act(e)[yi] = n_act_y;
      ls.year_end(yi, ngen, noff);
    }
  }
  if (lane == 0) ls.finish(run_pos, def_pos, act_pos, ngen, noff, bytes);
  return true;
}

// ---- the yearly row of year yi of episode `e` (one wave): every aggregate of the year — the existing-plant prefix, then every
//      generator / offset of the list as it stands at the year's end (G, NO entries), in list order: year_fold's sums continued by
//      apply_action's additions, the same additions in the same order —, then the yearly metrics (metrics_calculation.rs:32-175) as
//      k_rollout writes them.  Lane 0 writes the row (without the three running totals, which run through the years:
//      metrics_calculation.rs:133-153) and every lane returns {yearly total, credit, sales, net CO2, opinion, total capital, balance}.
//      `u16_at(list, i)`: the caller's way of reading a list entry of the record (a wave that stored the entries itself reads them
//      past its CU's L1). ----
struct YearRow { double v[7]; };
template <class Load>
__device__ __forceinline__ YearRow books_year(const DevTables& T, const DevSnapshot& S_in, const DevOut& O, uint32_t e, int yi, int G, int NO, int lane, Load&& u16_at) {
  const uint16_t* gen_cell = O.gen_cell(e); const uint16_t* gen_pack = O.gen_pack(e); const uint16_t* off_pack = O.off_pack(e);
  double gcost = 0.0, gprev = 0.0, optot = uniform_f64(T.pre_optot()[yi]), co2 = uniform_f64(T.pre_co2()[yi]);
  double tg = uniform_f64(T.pre_tg()[yi]), ig = uniform_f64(T.pre_ig()[yi]), sg = uniform_f64(T.pre_sg()[yi]);
  const double* ccy = T.cc() + (unsigned)yi * kTypes * kYears * kMults * 2;
  const double* ccp = T.cc() + (unsigned)(yi > 0 ? yi - 1 : 0) * kTypes * kYears * kMults * 2;      // last year's prices (yearly capital, metrics_calculation.rs:109-117)
  struct GT { double c, o, p, e, x0, x1, x2; };
  auto gterms = [&](int i) -> GT {
    GT r = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (i >= G) return r;
    const int pk = u16_at(gen_pack, i), ty = pk & 15, b = (pk >> 4) & 31, m = pk >> 9;
    const unsigned at = ((unsigned)(ty * kYears + b) * kMults + m) * 2;
    const double2 cc = *reinterpret_cast<const double2*>(ccy + at);
    r.c = cc.x;
    r.o = (T.m03()[u16_at(gen_cell, i)] + T.t12()[(unsigned)yi * kTypes + ty]) + cc.y;
    r.p = yi > 0 ? ccp[at] : 0.0;
    r.e = T.co2_t()[ty];
    const int cls = T.cls()[ty]; const double out = T.out_mw()[ty];
    r.x0 = (cls != 1 && cls != 2) ? out : 0.0; r.x1 = cls == 1 ? out : 0.0; r.x2 = cls == 2 ? out : 0.0;
    return r;
  };
  GT x = gterms(lane);
  for (int base = 0; base < G; base += kWave) {
    const GT c = x;
    x = gterms(base + kWave + lane);      // the next block's terms are requested before this block is folded
    const int cnt = G - base < kWave ? G - base : kWave;
    for (int r = 0; r * 16 < cnt; ++r) {
      double a0 = gcost, a1 = optot, a2 = gprev, a3 = co2, a4 = tg, a5 = ig;
      fold2_row16(a0, c.c, a1, c.o); fold2_row16(a2, c.p, a3, c.e); fold2_row16(a4, c.x0, a5, c.x1);
      const double a6 = fold_row16(sg, c.x2);
      gcost = readlane_f64(a0, 16 * r); optot = readlane_f64(a1, 16 * r); gprev = readlane_f64(a2, 16 * r); co2 = readlane_f64(a3, 16 * r);
      tg = readlane_f64(a4, 16 * r); ig = readlane_f64(a5, 16 * r); sg = readlane_f64(a6, 16 * r);
    }
  }
  double offs = 0.0, ocost = 0.0, oprev = 0.0;
  for (int base = 0; base < NO; base += kWave) {
    double x0 = 0.0, x1 = 0.0, x2 = 0.0;
    if (base + lane < NO) {
      const int p = u16_at(off_pack, base + lane), ot = p & 15, b = (p >> 4) & 31, m = p >> 9;
      x0 = T.offv()[((unsigned)yi * kOffsetTypes + ot) * kYears + b];
      x1 = T.offc()[((unsigned)yi * kOffsetTypes + ot) * kMults + m];
      x2 = yi > 0 ? T.offc()[((unsigned)(yi - 1) * kOffsetTypes + ot) * kMults + m] : 0.0;
    }
    const int cnt = NO - base < kWave ? NO - base : kWave;
    for (int r = 0; r * 16 < cnt; ++r) {
      double a0 = offs, a1 = ocost;
      fold2_row16(a0, x0, a1, x1);
      const double a2 = fold_row16(oprev, x2);
      offs = readlane_f64(a0, 16 * r); ocost = readlane_f64(a1, 16 * r); oprev = readlane_f64(a2, 16 * r);
    }
  }
  Agg a;
  a.co2 = co2; a.tg = tg; a.ig = ig; a.sg = sg; a.optot = optot; a.gcost = gcost; a.ocost = ocost; a.gcost_prev = gprev; a.ocost_prev = oprev;
  a.offs = offs; a.usage = uniform_f64(T.usage()[yi]); a.opcnt = __builtin_amdgcn_readfirstlane(T.pre_opcnt()[yi]) + G;
  const State s = state_of(a);
  const double gen = (a.tg + a.ig) + a.sg;
  const double credit = s.net >= 0.0 ? 0.0 : (-s.net) * T.carbon_price()[yi];
  const double total_capital = a.gcost + a.ocost;
  const double yearly_capital = yi == 0 ? total_capital : total_capital - (a.gcost_prev + a.ocost_prev);
  double sales = 0.0;
  if (S_in.enable_energy_sales && s.balance > 0.0) { const double gwh = s.balance * 8.76; sales = gwh * 50000.0; }
  const double yearly_total = yearly_capital + 0.0 + 0.0 - credit - (S_in.enable_energy_sales ? sales : 0.0);
  if (lane == 0 && S_in.write_yearly) {
    double* row = O.yearly(e) + yi * EG_YEARLY_FIELDS;
    row[EG_Y_YEAR] = (double)(2025 + yi); row[EG_Y_POP] = T.population()[yi]; row[EG_Y_USAGE] = a.usage; row[EG_Y_GEN] = gen;
    row[EG_Y_BALANCE] = s.balance; row[EG_Y_OPINION] = s.opinion; row[EG_Y_YEARLY_CAPITAL] = yearly_capital;
    row[EG_Y_TOTAL_CAPITAL] = total_capital; row[EG_Y_INFLATION] = T.inflation()[yi]; row[EG_Y_CO2] = a.co2;
    row[EG_Y_OFFSET] = a.offs; row[EG_Y_NET_CO2] = s.net; row[EG_Y_YEARLY_CREDIT] = credit;
    row[EG_Y_YEARLY_SALES] = sales; row[EG_Y_ACTIVE_GENS] = (double)a.opcnt;
    row[EG_Y_UPGRADE_COSTS] = 0.0; row[EG_Y_CLOSURE_COSTS] = 0.0;
    row[EG_Y_YEARLY_TOTAL_COST] = yearly_total;
  }
  YearRow r;
  r.v[0] = yearly_total; r.v[1] = credit; r.v[2] = sales; r.v[3] = s.net; r.v[4] = s.opinion; r.v[5] = total_capital; r.v[6] = s.balance;
  return r;
}

// ---- the same rows, four years at a time (one wave: row r of its 64 lanes folds year y0 + r, sixteen list entries a step).  A year's
//      sums are chains of dependent additions — one v_fmac_f64_dpp per list entry and sum, sixteen lanes wide —, so one year at a time
//      three quarters of the wave idle through every chain; the years do not depend on each other, and a DPP row operation runs the
//      four rows' chains side by side.  `g_end` / `o_end`: the lists' lengths at the end of each year (non-decreasing).  A row past the
//      last year repeats it and writes nothing.  Every lane returns its row's year. ----
template <class Load>
__device__ __forceinline__ YearRow books_quad(const DevTables& T, const DevSnapshot& S_in, const DevOut& O, uint32_t e, int y0, const int* g_end, const int* o_end,
                                              int lane, Load&& u16_at) {
  const uint16_t* gen_cell = O.gen_cell(e); const uint16_t* gen_pack = O.gen_pack(e); const uint16_t* off_pack = O.off_pack(e);
  const int j = lane & 15, yr = y0 + (lane >> 4);
  const bool live = yr < kYears;
  const int yi = live ? yr : kYears - 1;
  const int ylast = y0 + 3 < kYears ? y0 + 3 : kYears - 1;
  const int G = g_end[yi], NO = o_end[yi];
  const int Gmax = __builtin_amdgcn_readfirstlane(g_end[ylast]), NOmax = __builtin_amdgcn_readfirstlane(o_end[ylast]);
  double gcost = 0.0, gprev = 0.0, optot = T.pre_optot()[yi], co2 = T.pre_co2()[yi];
  double tg = T.pre_tg()[yi], ig = T.pre_ig()[yi], sg = T.pre_sg()[yi];
  const double* ccy = T.cc() + (unsigned)yi * kTypes * kYears * kMults * 2;
  const double* ccp = T.cc() + (unsigned)(yi > 0 ? yi - 1 : 0) * kTypes * kYears * kMults * 2;
  struct GT { double c, o, p, e, x0, x1, x2; };
  auto gterms = [&](int i) -> GT {
    GT r = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (i >= G) return r;
    const int pk = u16_at(gen_pack, i), ty = pk & 15, b = (pk >> 4) & 31, m = pk >> 9;
    const unsigned at = ((unsigned)(ty * kYears + b) * kMults + m) * 2;
    const double2 cc = *reinterpret_cast<const double2*>(ccy + at);
    r.c = cc.x;
    r.o = (T.m03()[u16_at(gen_cell, i)] + T.t12()[(unsigned)yi * kTypes + ty]) + cc.y;
    r.p = yi > 0 ? ccp[at] : 0.0;
    r.e = T.co2_t()[ty];
    const int cls = T.cls()[ty]; const double out = T.out_mw()[ty];
    r.x0 = (cls != 1 && cls != 2) ? out : 0.0; r.x1 = cls == 1 ? out : 0.0; r.x2 = cls == 2 ? out : 0.0;
    return r;
  };
  GT x = gterms(j);
  for (int base = 0; base < Gmax; base += 16) {      // (a row whose year's list has ended adds +0.0: sums of non-negative terms)
    const GT c = x;
    x = gterms(base + 16 + j);      // the next step's terms are requested before this step is folded
    fold2_row16(gcost, c.c, optot, c.o); fold2_row16(gprev, c.p, co2, c.e); fold2_row16(tg, c.x0, ig, c.x1);
    sg = fold_row16(sg, c.x2);
  }
  double offs = 0.0, ocost = 0.0, oprev = 0.0;
  for (int base = 0; base < NOmax; base += 16) {
    double x0 = 0.0, x1 = 0.0, x2 = 0.0;
    if (base + j < NO) {
      const int p = u16_at(off_pack, base + j), ot = p & 15, b = (p >> 4) & 31, m = p >> 9;
      x0 = T.offv()[((unsigned)yi * kOffsetTypes + ot) * kYears + b];
      x1 = T.offc()[((unsigned)yi * kOffsetTypes + ot) * kMults + m];
      x2 = yi > 0 ? T.offc()[((unsigned)(yi - 1) * kOffsetTypes + ot) * kMults + m] : 0.0;
    }
    fold2_row16(offs, x0, ocost, x1);
    oprev = fold_row16(oprev, x2);
  }
  Agg a;
  a.co2 = co2; a.tg = tg; a.ig = ig; a.sg = sg; a.optot = optot; a.gcost = gcost; a.ocost = ocost; a.gcost_prev = gprev; a.ocost_prev = oprev;
  a.offs = offs; a.usage = T.usage()[yi]; a.opcnt = T.pre_opcnt()[yi] + G;
  const State s = state_of(a);
  const double gen = (a.tg + a.ig) + a.sg;
  const double credit = s.net >= 0.0 ? 0.0 : (-s.net) * T.carbon_price()[yi];
  const double total_capital = a.gcost + a.ocost;
  const double yearly_capital = yi == 0 ? total_capital : total_capital - (a.gcost_prev + a.ocost_prev);
  double sales = 0.0;
  if (S_in.enable_energy_sales && s.balance > 0.0) { const double gwh = s.balance * 8.76; sales = gwh * 50000.0; }
  const double yearly_total = yearly_capital + 0.0 + 0.0 - credit - (S_in.enable_energy_sales ? sales : 0.0);
  if (j == 0 && live && S_in.write_yearly) {
    double* row = O.yearly(e) + yi * EG_YEARLY_FIELDS;
    row[EG_Y_YEAR] = (double)(2025 + yi); row[EG_Y_POP] = T.population()[yi]; row[EG_Y_USAGE] = a.usage; row[EG_Y_GEN] = gen;
    row[EG_Y_BALANCE] = s.balance; row[EG_Y_OPINION] = s.opinion; row[EG_Y_YEARLY_CAPITAL] = yearly_capital;
    row[EG_Y_TOTAL_CAPITAL] = total_capital; row[EG_Y_INFLATION] = T.inflation()[yi]; row[EG_Y_CO2] = a.co2;
    row[EG_Y_OFFSET] = a.offs; row[EG_Y_NET_CO2] = s.net; row[EG_Y_YEARLY_CREDIT] = credit;
    row[EG_Y_YEARLY_SALES] = sales; row[EG_Y_ACTIVE_GENS] = (double)a.opcnt;
    row[EG_Y_UPGRADE_COSTS] = 0.0; row[EG_Y_CLOSURE_COSTS] = 0.0;
    row[EG_Y_YEARLY_TOTAL_COST] = yearly_total;
  }
  YearRow r;
  r.v[0] = yearly_total; r.v[1] = credit; r.v[2] = sales; r.v[3] = s.net; r.v[4] = s.opinion; r.v[5] = total_capital; r.v[6] = s.balance;
  return r;
}

}  // namespace rs
