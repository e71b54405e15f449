// eg_rollout.hip — CDNA4 (gfx950) kernels of the rollout engine.
//
// One wavefront runs one 2025–2050 episode from start to finish; a batch is a grid of episodes, one workgroup each
// (k_rollout<0>: the episode wave alone; k_rollout<1>, small batches: plus a helper wave, see helper_loop).  Control
// flow of an episode is wave-uniform (every lane carries the same scalar state); the lanes fan out over the 64
// candidates of a chunk of the placement search, the per-generator / per-offset gathers at the start of a year, the
// weight rows (nudges, prefix scans of the samplers) and the ChaCha blocks.  Sums and products that end up in an output
// are folded in the reference's list order and every transcendental is a host-built table (eg_tables.cpp), so device
// code is + - * / compare only and reproduces the CPU oracle bit for bit; chains that only feed a decision (the samplers'
// sum-and-walk) are replaced by a parallel form where that provably gives the same decision (weighted_pick).
// Build: -ffp-contract=off (no FMA); compiled twice — the throughput kernels as an object of their own, see EG_TU_THROUGHPUT
// below and csrc/Makefile.  Further kernels: k_apply_update (the batch update, on the device),
// k_stalled_tables, k_pick_best, k_update_stats, k_place.
// Two measured facts shape the code: (1) in the small-batch kernel a lone wave issues one instruction per turn of its
// SIMD, whatever the instruction — the hot loops are written for instruction count, scalar and wait instructions
// included (chunk_product_latency); (2) gfx950 allocates LDS in 1280-byte granules — the throughput kernel's Smem is
// held at seven of them (room for eighteen episodes per CU; sixteen run: four waves per SIMD at 128 VGPRs), the small-batch kernel
// spends LDS freely (SmemLatency).
//
// Reference (paths relative to /root/reference/aiSimulator/src/):
//   episode        core/simulation.rs:22-317, core/iteration.rs:57-74
//   deficit loop   core/simulation.rs:319-522
//   sampling       ai/learning/weights/sampling.rs:76-443
//   nudges         ai/learning/weights/learning.rs:21-88, deficit.rs:82-135
//   apply_action   core/actions.rs:40-204
//   placement      gpu/metal_location_search.rs:110-176
//   metrics        analysis/metrics_calculation.rs:7-175, ai/metrics/scoring.rs:46-85
//   RNG            rand 0.8.5 StdRng = ChaCha12 behind rand_core BlockRng (Cargo.lock:763-785)
#include <hip/hip_runtime.h>
#include <cstddef>
#include <hip/hip_ext.h>

#include "eg_internal.h"

#define EG_DETPOW_QUAL __device__ __forceinline__
#include "eg_detpow.h"
#define EG_RM __device__ __forceinline__
#include "eg_reduced_math.h"

namespace eg {
namespace {

constexpr int kWave = 64;
constexpr int kD2Max = 144, kD2Stride = 146;   // factor table by squared cell distance: 0..144 (12 km = the largest radius), padded
constexpr int kCmdYear = 1 << 30;       // helper command: fold next year's starting sums (else: a placement search)
constexpr int kHelperWaves = 1;         // small-batch kernel: waves per episode beyond the episode wave (see helper_loop)
// The episode wave polls an LDS flag for the helper's results (s_sleep 1 = 64 cycles per poll).  The helper's longest job —
// a chunk against 512 generators, or a year's sums over full lists — is a few 10^4 cycles; after 2^20 polls (~30 ms) the
// protocol must have slipped: the episode ends with EG_EP_INTERNAL instead of hanging the GPU.
constexpr int kSpinCap = 1 << 20;
constexpr int kSearchLost = -2;         // place_search: the helper never answered
constexpr double kMinWeight = 0.0001;   // ai/learning/constants.rs:14
constexpr double kMaxWeight = 0.999;    // constants.rs:15
constexpr double kMaxCost = 50000000000.0;   // config/constants.rs:115
constexpr int kBattery = 12, kPeaker = 8;
constexpr int kNothing = 60, kFirstOffset = 45, kFirstOther = 57;

// deficit-table slot -> generator type (core.rs:130-149) and the inverse
__constant__ int c_deficit_type[14] = {8, 7, 12, 11, 9, 0, 1, 4, 10, 5, 2, 3, 13, 14};
__constant__ int c_deficit_slot[15] = {5, 6, 10, 11, 7, 9, -1, 1, 0, 4, 8, 3, 2, 12, 13};
// the same map as an immediate (4 bits per generator type, 15 = none) for loops that cannot afford a constant-memory
// round trip per element: slot of an action index, -1 if the deficit table has no entry for it
__device__ __forceinline__ int deficit_slot_of(int action) {
  constexpr unsigned long long kNibbles = 0xDC238401F97BA65ull;      // type 0 in the low nibble: 5, 6, 10, 11, 7, 9, 15, 1, 0, 4, 8, 3, 2, 12, 13
  if (action == 60) return 14;                                        // DoNothing
  if (action >= 45 || action % 3 != 0) return -1;
  const int s = (int)((kNibbles >> (4 * (action / 3))) & 15ull);
  return s == 15 ? -1 : s;
}

struct __align__(16) Smem {
  double dr[kDrCompact];              // d/R by squared cell distance, class k at entries off_k .. off_k + cap_k (tab::dr_meta): only up to its own
                                      // radius — 1.0 at cap_k, where every larger distance is capped (2.7 instead of 7 KB: with 128 VGPRs that is
                                      // what lets a fourth wave share the SIMD)
  int gstage[2][kWave + 8];           // per wave: (gi, gj) as two int16 of the 64 generators being folded (read back four at a time);
                                      // the tail stays at the off-grid padding value: the pipeline reads up to three groups ahead
  double scaled[64];                  // stalled sampler: weights^p in sorted order
  double type_out[16];                // per generator type: output of an operational new plant
  double type_co2[16];                //                     CO2
  int type_info[16];                  //                     variant | radius class << 4 | reach << 8 | output class << 12
  double yend[6];                     // last year's end-of-year sums: capital (generators, offsets), CO2, output classes
  double ystate[4];                   // net CO2 / opinion / balance / cost at the start of the year (success bonus of the repair loop)
  double acc[8];                      // once-a-year accumulators kept out of registers: total cost / credit / sales, last row
  double pol[snap::kPolRow];          // this year's policy row block (layout: eg_internal.h, namespace snap)
  uint32_t rng[64];                   // ChaCha12 output buffer: four blocks
  uint32_t rng_key[8];                // ChaCha12 key of the episode stream
  unsigned long long rng_counter;     // next block counter
  uint16_t gcell[kLdsGens];           // cell | type << 12      (the first kLdsGens generators / offsets of the episode: all of them,
  uint16_t opack[kLdsGens];           // type | year << 4 | mult << 9     except in the long-replay variant, whose lists go on in
  uint8_t gbm[kLdsGens];              // build-year index | mult << 5     the episode's record in HBM — ListTail)
  uint8_t ydef[192];                  // [0,128) this year's deficit actions (success bonus, simulation.rs:505-519);
                                      // [128,192) sort permutation of the stalled sampler
  // helper waves (small-batch kernel only): search command (double-buffered by sequence parity), results, flags
  int cmd[2][2];                      // {year | variant << 8 | radius class << 12 (or -1: exit), generators in the list}
  uint32_t hflag[2];                  // sequence number of the search whose result is in hres[h]
  uint32_t yflag; int ysum_opcnt;     // year-start sums folded by the helper wave (sequence number, count)
  double ysum[8];                     //   gcost, optot, offs, ocost, co2, tg, ig, sg
  struct { double score, m03; int cell, pad; } hres[2];
  double park[18];                    // long-replay variant: the year's aggregates while the field code runs (see k_rollout)
#ifdef EG_STAMPS
  unsigned long long hdbg[2][4];
#endif
};
// gfx950 hands out LDS in granules of 1280 bytes (160 KB / 128): seven granules per episode leave room for the sixteen workgroups per
// CU that four waves per SIMD (128 VGPRs) allow.  (Ten granules and three waves until the factor table was compacted: 1.305 -> 1.194 ms
// at 16 384 episodes.)
#ifndef EG_STAMPS
static_assert(sizeof(Smem) <= 7 * 1280, "sixteen episodes per CU");
#endif

// One instance per workgroup (= per episode).  File scope so that non-inlined helpers address it as LDS.
__shared__ Smem sm;
// Extra LDS of the small-batch kernel only (it is not referenced by the throughput kernel, so it costs that one nothing).
struct __align__(16) SmemLatency {
  int gpk[kLdsGens + 16];                           // (4 gi, 4 gj) as two int16 per generator of the episode; padding beyond the list
  double dr16[kRadiusClasses * kD2Stride * 2];      // the factor table at a stride of 16 bytes (see chunk_product_latency)
};
__shared__ SmemLatency sl;
constexpr int kGenPad4 = (int)0xE000E000;           // a generator far off the grid, in the x4 coordinates
#define SM_W (sm.pol)                      // main weights [61]
#define SM_DW (sm.pol + snap::kPolDw)      // deficit weights [15]
#define SM_CW (sm.pol + snap::kPolCw)      // action-count weights [21]

// A workgroup is ONE wavefront, and a wave's LDS operations execute in program order, so data written to LDS by one
// lane is visible to every lane's later reads without waiting; what is needed is only that the compiler keeps the
// program order of the LDS accesses.  Unlike wave_sync() this does not drain vmcnt, so global loads requested ahead
// of time and the episode's output stores stay in flight across it.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

struct Rng {      // register part of the stream state; key / counter / buffer live in LDS
  int index;
  unsigned int words;
};
struct Totals {   // (the sums of the rows themselves are taken inside weighted_pick)
  bool scaled_valid;   // the stalled sampler's sorted / powered table in LDS matches the weight row
};

struct Agg {   // aggregates of the map at the current point of the year (map_handler.rs:819-965)
  double co2, tg, ig, sg, optot, gcost, ocost, gcost_prev, ocost_prev, offs, usage;
  int opcnt;
};
struct State { double net, opinion, balance, cost; };   // ActionResult, simulation_metrics.rs:13-18

// A condition that is the same in every lane, made known to the compiler as such (a scalar branch instead of an
// exec-masked region with its save / restore instructions).
#define EG_UNI(c) (__builtin_amdgcn_ballot_w64(c) != 0ull)
__device__ __forceinline__ double dmin(double a, double b) { return a < b ? a : b; }
__device__ __forceinline__ double dmax(double a, double b) { return a > b ? a : b; }
__device__ __forceinline__ double dabs(double a) { return a < 0.0 ? -a : a; }
// v_max_f64 / v_min_f64 as such: from `a > b ? a : b` the compiler makes a compare and two selects (it may not assume that there are no
// NaNs and that the sign of a zero does not matter).  Only where neither can occur: weights and scores (positive, finite), and
// max(|x|, 1.0) (at least 1.0 whatever the sign of a zero x).
__device__ __forceinline__ double vmax64(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double vmin64(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double vmax64_u(double a, double uniform_b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(uniform_b)); return r; }
__device__ __forceinline__ double vmin64_u(double a, double uniform_b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(uniform_b)); return r; }
__device__ __forceinline__ double max_abs_one(double a) { double r; asm("v_max_f64 %0, |%1|, 1.0" : "=v"(r) : "v"(a)); return r; }      // max(|a|, 1.0)
__device__ __forceinline__ uint32_t rotl32(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }

#define EG_QR(a, b, c, d)                                                         \
  a += b; d ^= a; d = rotl32(d, 16); c += d; b ^= c; b = rotl32(b, 12);           \
  a += b; d ^= a; d = rotl32(d, 8);  c += d; b ^= c; b = rotl32(b, 7);

// Four consecutive ChaCha12 blocks (rand_chacha fills 64 words per refill).  Not inlined: the episode code draws from ~10
// places.  Sixteen lanes work on it: lane 4 b + c holds column c of block b — a word of each of the state's four rows — so a column
// round is one quarter round per lane, and a diagonal round is the same quarter round after rotating rows 1, 2, 3 by one, two,
// three lanes inside the quad (DPP quad_perm) and back: 12 + 6 instructions per lane and half-round pair... 180 for the twelve
// rounds, where one lane per block (four lanes, sixteen words each) took 1 200 — a refill every 32 draws, 2.5 per episode.
__device__ __forceinline__ uint32_t quad_rot(uint32_t v, int ctrl_is_1230_2301_3012) {
  // lane i of a quad takes the value of lane (i + k) % 4: quad_perm [1,2,3,0] = 0x39, [2,3,0,1] = 0x4E, [3,0,1,2] = 0x93
  if (ctrl_is_1230_2301_3012 == 1) return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x39, 0xf, 0xf, false);
  if (ctrl_is_1230_2301_3012 == 2) return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xf, 0xf, false);
  return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x93, 0xf, 0xf, false);
}
__device__ __noinline__ void rng_refill(int lane) {
  wave_sync();
  if (lane < 16) {
    const int blk = lane >> 2, col = lane & 3;
    const unsigned long long counter = sm.rng_counter + (unsigned long long)blk;
    const uint32_t a0 = col == 0 ? 0x61707865u : (col == 1 ? 0x3320646eu : (col == 2 ? 0x79622d32u : 0x6b206574u));
    const uint32_t b0 = sm.rng_key[col], c0 = sm.rng_key[4 + col];
    const uint32_t d0 = col == 0 ? (uint32_t)counter : (col == 1 ? (uint32_t)(counter >> 32) : 0u);
    uint32_t a = a0, bq = b0, c = c0, d = d0;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      EG_QR(a, bq, c, d)                                                    // columns
      bq = quad_rot(bq, 1); c = quad_rot(c, 2); d = quad_rot(d, 3);        // row k moves k lanes: the diagonals line up as columns
      EG_QR(a, bq, c, d)                                                    // diagonals
      bq = quad_rot(bq, 3); c = quad_rot(c, 2); d = quad_rot(d, 1);        // ... and back
    }
    uint32_t* out = sm.rng + 16 * blk + col;
    out[0] = a + a0; out[4] = bq + b0; out[8] = c + c0; out[12] = d + d0;
  }
  wave_sync();
  if (lane == 0) sm.rng_counter += 4ull;
  wave_sync();
}

__device__ void rng_seed(Rng& r, unsigned long long state, int lane) {   // rand_core 0.6.4 seed_from_u64
  const unsigned long long MUL = 6364136223846793005ull, INC = 11634580027462260723ull;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    state = state * MUL + INC;
    uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
    uint32_t rot = (uint32_t)(state >> 59);
    if (lane == 0) sm.rng_key[i] = (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
  }
  if (lane == 0) sm.rng_counter = 0ull;
  wave_sync();
  r.index = 64; r.words = 0;
}
__device__ __forceinline__ unsigned long long rng_u64(Rng& r, int lane) {   // BlockRng::next_u64
  r.words += 1;
  unsigned long long lo;
  if (r.index >= 63) {                      // one word left (63) or none (>= 64): refill once
    const bool straddle = r.index == 63;
    const uint32_t tail = sm.rng[63];
    rng_refill(lane);
    if (straddle) { r.index = 1; return ((unsigned long long)sm.rng[0] << 32) | tail; }
    r.index = 0;
  }
  lo = sm.rng[r.index];
  const unsigned long long hi = sm.rng[r.index + 1];
  r.index += 2;
  return (hi << 32) | lo;
}
__device__ __forceinline__ uint32_t rng_u32(Rng& r, int lane) {   // BlockRng::next_u32
  r.words += 1;
  if (r.index >= 64) { rng_refill(lane); r.index = 0; }
  const uint32_t v = sm.rng[r.index];
  r.index += 1;
  return v;
}
__device__ __forceinline__ double rng_f64(Rng& r, int lane) {   // Standard: 53 bits, [0,1)
  return (double)(rng_u64(r, lane) >> 11) * (1.0 / 9007199254740992.0);
}
__device__ unsigned long long rng_range64(Rng& r, int lane, unsigned long long range) {
  const unsigned long long zone = (range << __clzll((long long)range)) - 1ull;   // uniform.rs sample_single_inclusive
  for (int guard = 0; guard < 4096; ++guard) {
    const unsigned long long v = rng_u64(r, lane);
    const unsigned long long hi = __umul64hi(v, range), lo = v * range;
    if (lo <= zone) return hi;
  }
  return 0;
}
__device__ uint32_t rng_range32(Rng& r, int lane, uint32_t range) {
  const uint32_t zone = (range << __clz((int)range)) - 1u;
  for (int guard = 0; guard < 4096; ++guard) {
    const uint32_t v = rng_u32(r, lane);
    const unsigned long long m = (unsigned long long)v * range;
    if ((uint32_t)m <= zone) return (uint32_t)(m >> 32);
  }
  return 0;
}

__device__ __forceinline__ State state_of(const Agg& a) {   // simulation.rs:122-135
  State s;
  s.net = a.co2 - a.offs;
  s.opinion = a.opcnt > 0 ? a.optot / (double)a.opcnt : 1.0;
  s.balance = ((a.tg + a.ig) + a.sg) - a.usage;
  s.cost = a.gcost + a.ocost;
  return s;
}
__device__ double evaluate_impact(const State& cur, const State& nxt) {   // scoring.rs:46-85 (mode None)
  if (cur.net > 0.0) return (cur.net - nxt.net) / max_abs_one(cur.net);
  double cost_change = nxt.cost - cur.cost;
  double cost_improvement = -cost_change / max_abs_one(cur.cost);
  double opinion_improvement = (nxt.opinion - cur.opinion) / max_abs_one(cur.opinion);
  double cost_weight = cur.cost > kMaxCost * 8.0 ? 0.8 : 0.5;
  double opinion_weight = 1.0 - cost_weight;
  return cost_improvement * cost_weight + opinion_improvement * opinion_weight;
}

// small per-type / per-class tables -> LDS (one dependent LDS read instead of chains of L2 round trips)
// The host table is indexed by (|di|, |dj|) <= 12; the distance, hence the factor, only depends on di^2 + dj^2 (grid
// coordinates are whole kilometres, so dx^2 + dy^2 is exact), and the kernels index by that: one dot product instead of
// two absolute values, two clamps and a linearisation.  Sums that are not a sum of two squares are never read.
__device__ __forceinline__ void load_factor_table(const DevTables& T, int lane) {
  // tails of the staging rows of chunk_product: generators far off the grid (their factor is 1.0), never overwritten
  if (lane < 8) { sm.gstage[0][kWave + lane] = (int)0xC000C000; sm.gstage[1][kWave + lane] = (int)0xC000C000; }
  // (the host lays the compact table out as LDS holds it — tab::dr_compact: one independent load per lane and block of 64, one
  //  memory round trip at the start of an episode where re-indexing the [6][13][13] table here was sixteen dependent ones)
  double v[(kDrCompact + kWave - 1) / kWave];
#pragma unroll
  for (int k = 0; k < (kDrCompact + kWave - 1) / kWave; ++k) v[k] = lane + k * kWave < kDrCompact ? T.dr_compact()[lane + k * kWave] : 1.0;
#pragma unroll
  for (int k = 0; k < (kDrCompact + kWave - 1) / kWave; ++k) if (lane + k * kWave < kDrCompact) sm.dr[lane + k * kWave] = v[k];
}
// byte offset of a radius class's factors inside the LDS block | its cap << 16: what chunk_product / factor_by_q take as `table`
// in the throughput kernels (from the type's info word: bits 16-23 cap, 24-31 first entry / 2)
__device__ __forceinline__ int throughput_table(int info) {
  return ((int)offsetof(Smem, dr) + (int)(((unsigned)info >> 24) << 4)) | (((info >> 16) & 255) << 16);
}
// small-batch kernel: the generator list starts as padding everywhere; the factor table goes in at a stride of 16 bytes
__device__ __forceinline__ void load_latency_tables(const DevTables& T, int lane) {
  for (int i = lane; i < kLdsGens + 16; i += kWave) sl.gpk[i] = kGenPad4;
  for (int i = lane; i < kRadiusClasses * kD2Stride; i += kWave) sl.dr16[2 * i] = 1.0;
  for (int i = lane; i < kRadiusClasses * 169; i += kWave) {
    const int rc = i / 169, k = i - rc * 169, di = k / 13, dj = k - di * 13, q = di * di + dj * dj;
    if (q <= kD2Max) sl.dr16[2 * (rc * kD2Stride + q)] = T.dr()[i];
  }
}
__device__ __forceinline__ void load_static_tables(const DevTables& T, int lane, bool with_factors) {
  if (with_factors) load_factor_table(T, lane);
  if (lane < kTypes) {
    const int rc = T.rclass()[lane];
    sm.type_info[lane] = T.variant()[lane] | (rc << 4) | (T.reach()[rc] << 8) | (T.cls()[lane] << 12) | (T.dr_meta()[8 + rc] << 16) |
                         (int)((unsigned)(T.dr_meta()[rc] >> 1) << 24);
    sm.type_out[lane] = T.out_mw()[lane];
    sm.type_co2[lane] = T.co2_t()[lane];
  }
}

// ---- placement: arg-max over the 51x51 distinct candidates (Q10) -------------------------------------------
// Reference (metal_location_search.rs:110-176): score(c) = ((te[c] * prod_{g in list order, d<R} d/R) * coast(c)) * 0.9,
// keep the first strictly greater score in (i, j) order, i.e. the maximum with ties to the lowest cell index.
// Here the candidates of (year, radius class, marine) come pre-sorted by their unpenalised score (host, eg_api.cpp).
// Every penalty factor is in [0, 1] and IEEE multiplication is monotone, so score(c) <= base(c): the scan takes 64
// candidates at a time (one per lane, each lane folding the generator list in order for its own cell) and stops as
// soon as the next chunk's largest base score is below the best score found — the same winner, bit for bit, as the
// exhaustive search, usually after one or two chunks.
__device__ __forceinline__ double readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// max over the 64 lanes of a NON-NEGATIVE double, returned in every lane.  For non-negative IEEE doubles the order of the
// values is the order of their bit patterns, so the maximum is found on the high dwords first and then on the low dwords
// of the lanes that hold that high dword: two 32-bit DPP reductions (row shifts / row broadcasts; a lane without a DPP
// source reads 0, the identity of an unsigned max) instead of one on 64-bit pairs.
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#define EG_DPP_MAX_STEP(ctrl, row_mask, bank_mask)                                                      \
  {                                                                                                     \
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, row_mask, bank_mask, false); \
    v = o > v ? o : v;                                                                                  \
  }
  EG_DPP_MAX_STEP(0x111, 0xf, 0xf)   // row_shr:1
  EG_DPP_MAX_STEP(0x112, 0xf, 0xf)   // row_shr:2
  EG_DPP_MAX_STEP(0x113, 0xf, 0xf)   // row_shr:3
  EG_DPP_MAX_STEP(0x114, 0xf, 0xe)   // row_shr:4 bank_mask:0xe
  EG_DPP_MAX_STEP(0x118, 0xf, 0xc)   // row_shr:8 bank_mask:0xc
  EG_DPP_MAX_STEP(0x142, 0xa, 0xf)   // row_bcast:15 row_mask:0xa
  EG_DPP_MAX_STEP(0x143, 0xc, 0xf)   // row_bcast:31 row_mask:0xc
#undef EG_DPP_MAX_STEP
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ double wave_max_f64(double v) {
  const unsigned hi = (unsigned)__double2hiint(v);
  const unsigned mh = wave_max_u32(hi);
  const unsigned ml = wave_max_u32(hi == mh ? (unsigned)__double2loint(v) : 0u);
  return __hiloint2double((int)mh, (int)ml);
}

// Inclusive prefix sum over the 64 lanes (DPP: four row shifts, two row broadcasts; a lane without a source adds 0.0).
// (The shifts say bound_ctrl: a lane whose source lies outside the row reads 0 — no register has to be zeroed first; the
//  broadcasts only write the rows of their row mask, the others keep the 0 they were given.)
__device__ __forceinline__ double wave_prefix_sum_f64(double x) {
#define EG_DPP_ADD_STEP(ctrl, row_mask, bank_mask, bound)                                                           \
  {                                                                                                                 \
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), ctrl, row_mask, bank_mask, bound);             \
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), ctrl, row_mask, bank_mask, bound);             \
    x = x + __hiloint2double(hi, lo);                                                                               \
  }
  EG_DPP_ADD_STEP(0x111, 0xf, 0xf, true)    // row_shr:1
  EG_DPP_ADD_STEP(0x112, 0xf, 0xf, true)    // row_shr:2
  EG_DPP_ADD_STEP(0x114, 0xf, 0xf, true)    // row_shr:4
  EG_DPP_ADD_STEP(0x118, 0xf, 0xf, true)    // row_shr:8
  EG_DPP_ADD_STEP(0x142, 0xa, 0xf, false)   // row_bcast:15 -> rows 1, 3
  EG_DPP_ADD_STEP(0x143, 0xc, 0xf, false)   // row_bcast:31 -> rows 2, 3
#undef EG_DPP_ADD_STEP
  return x;
}

// The weighted pick of the samplers (sampling.rs:182-233, :352-370, :406-416): total = sum of the table in table order,
// v = u * total, then v -= table[a] in table order; the pick is the number of entries after which v was still positive
// (weights are not negative, so v never grows).  Every addition and subtraction rounds, so sum and walk are sequential
// chains — but the OUTCOME only depends on the signs of the v_a, and v_a differs from u * T - P_a, with the prefix sums
// P_a and their last value T evaluated in any other order, by less than 2^-45 * total (61 roundings of values below
// `total` in the sequential sum, 61 in the walk, a 6-level tree on the other side, one multiplication and one
// subtraction each).  So: prefix sums in parallel, and if every u * T - P_a is further than 2^-40 * T from zero the signs
// — hence the pick — are those of the sequential sum-and-walk.  Otherwise (a draw within 1e-12 of a boundary) the
// sequential chains decide.  tests/test_weighted_pick.py checks the rule on the CPU.
// Not inlined (four call sites; inlined, the rarely-run sequential part costs the episode loop its registers).  The table
// is given by its byte offset inside the workgroup's LDS block so that it is read with LDS instructions.
__device__ __noinline__ int weighted_pick(int table_offset, int n, double u, int lane) {
#if defined(EG_PROBE_SKIP) && EG_PROBE_SKIP == 5
  if (true) return (int)(u * (double)n);
#endif
  const double* table = reinterpret_cast<const double*>(reinterpret_cast<const char*>(&sm) + table_offset);
  const double w = lane < n ? table[lane] : 0.0;
  const double P = wave_prefix_sum_f64(w);
  const double T = readlane_f64(P, kWave - 1);             // lanes >= n added 0.0
  const double d = u * T - P;
  const unsigned long long valid = (1ull << n) - 1ull;      // n < 64
  if ((__ballot(dabs(d) <= T * 0x1p-40) & valid) == 0ull) return __popcll(__ballot(d > 0.0) & valid);
  double total = 0.0;
  for (int a = 0; a < n; ++a) total += table[a];
  int pick = 0; double v = u * total;
  for (int a = 0; a < n; ++a) { v -= table[a]; pick += v > 0.0 ? 1 : 0; }
  return pick;
}

typedef short short2v __attribute__((ext_vector_type(2)));
// factor of one generator for this lane's candidate: (ci - gi, cj - gj) as two int16, their squared length by one dot
// product, capped at 144 (12 km, the largest radius: the table holds 1.0 there)
__device__ __forceinline__ double factor_at(short2v cpk, int gen_packed, int table_offset, int cap) {
  short2v g; __builtin_memcpy(&g, &gen_packed, 4);
  const short2v d = cpk - g;
  // (the three-operand form with the constant 0 as its accumulator: from the builtin the compiler makes the two-operand
  //  accumulating v_dot2c and a v_mov to zero its destination first — one instruction in six of this loop)
  int q, dbits; __builtin_memcpy(&dbits, &d, 4);
  asm("v_dot2_i32_i16 %0, %1, %1, 0" : "=v"(q) : "v"(dbits));
  q = q < cap ? q : cap;      // (the class's own radius squared: the table holds 1.0 there)
  return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(&sm) + (table_offset + q * 8));
}

// A search multiplies te by the factor of every generator, in list order.  Searches of the same (year, variant) revisit
// the same candidates while the list only grows at its end, so the running product of a chunk is kept between searches
// and a later search of that (year, variant) continues it with the generators added since — the same multiplications in
// the same order, hence the same bits.  (Years with many additions — the slow episodes — mostly repeat one variant.)
struct PrefixCache { double product; int key; int count; };      // key: year << 8 | variant, -1 = empty; count: generators folded in

// s_init times the factors of generators [k0, ngen_s) for this lane's candidate cell.  `stage`: 0 on the episode wave, 1 on
// the helper wave (each wave has its own staging row in LDS).
// A lone wave issues one instruction per turn of its SIMD whatever the instruction is, so the loop is written for the
// fewest instructions per generator: the packed coordinates of 64 generators are staged in LDS once and come back four
// at a time as ONE broadcast 128-bit read (instead of a readlane and a lane-index add each); per generator that leaves a
// packed subtract, the dot product, the cap, the address, the table read and the multiply.
// Throughput kernel (one wave per episode, sixteen episodes per CU).  `stage`: the wave's staging row in LDS.
// The packed coordinates of 64 generators are staged in LDS once and come back four at a time as ONE broadcast 128-bit
// read (instead of a readlane and a lane-index add each); per generator that leaves a packed subtract, the dot product,
// the cap, the address, the table read and the multiply.
// `row0_kept` (lean / short-replay kernels): staging row 0 is kept up to date by the episode — a generator's packed coordinates are
// appended to it when the generator is placed (k_rollout), its other entries stay padding — so the first 64 generators need no
// staging at all: a chunk re-staged the same list, cell -> (i, j) division included, ~15 vector instructions, 95 times an episode.
// Blocks beyond the first 64 are staged in row 1 (the helper wave's, unused in those kernels).
// `xy4`: the candidate's packed coordinates times four (PsRec::pad, as the small-batch kernel uses them): a packed shift gives (ci, cj)
// where dividing the cell by the grid's width took eight instructions per chunk.
__device__ __forceinline__ double chunk_product(int table, int lane, int k0, int ngen_s, double s_init, int xy4, int stage, bool row0_kept = false) {
  double s = s_init;
  const short2v cpk = __builtin_bit_cast(short2v, xy4) >> (short)2;      // (4 ci, 4 cj) -> (ci, cj): both halves are small and not negative
  const int dr_off = table & 0xFFFF, cap = table >> 16;   // throughput_table(): the class's factors inside the LDS block, and where they end
  const int k0_s = __builtin_amdgcn_readfirstlane(k0);           // uniform (it comes out of a per-wave cache): scalar loop control
  for (int gb = k0_s; gb < ngen_s; gb += kWave) {                 // generators in list order
    const bool kept = row0_kept && gb == 0;
    int* row = sm.gstage[row0_kept && gb > 0 ? 1 : stage];
    const int4* row4 = reinterpret_cast<const int4*>(row);
    if (!kept) {
      // Lanes beyond the list hold a generator far off the grid: its squared distance caps at 144, where the table is 1.0.
      const int mine = gb + lane < ngen_s ? (int)(sm.gcell[gb + lane] & 0xFFF) : -1;
      const int mi = mine / kGrid;
      const int mp = mine < 0 ? (int)0xC000C000 : (mi | ((mine - mi * kGrid) << 16));   // (gi, gj) as two int16
      row[lane] = mp;
    }
    asm volatile("" ::: "memory");      // LDS executes a wave's accesses in program order: only the compiler must keep it
    const int cnt = ngen_s - gb < kWave ? ngen_s - gb : kWave;
    const int groups = (cnt + 3) >> 2;
    // Branch-free: the factor table holds 1.0 wherever d >= R (including the cap), and x * 1.0 == x exactly, so
    // out-of-range generators and the padding up to a multiple of four multiply by 1.0 instead of branching.
    // Software-pipelined: while four factors are multiplied in list order (only the multiplies form a chain) the next
    // four are on their way from the table; two register sets take turns.
#define EG_FACTORS_LOOSE(g4, f0, f1, f2, f3) { f0 = factor_at(cpk, g4.x, dr_off, cap); f1 = factor_at(cpk, g4.y, dr_off, cap); \
                                               f2 = factor_at(cpk, g4.z, dr_off, cap); f3 = factor_at(cpk, g4.w, dr_off, cap); }
    double a0, a1, a2, a3, b0, b1, b2, b3;
    int4 gc = row4[0], gn = row4[1];
    EG_FACTORS_LOOSE(gc, a0, a1, a2, a3)
    for (int k = 1;;) {
      if (k >= groups) { s = s * a0; s = s * a1; s = s * a2; s = s * a3; break; }
      gc = gn; gn = row4[k + 1];
      EG_FACTORS_LOOSE(gc, b0, b1, b2, b3)
      __builtin_amdgcn_sched_barrier(0);
      s = s * a0; s = s * a1; s = s * a2; s = s * a3;
      ++k;
      if (k >= groups) { s = s * b0; s = s * b1; s = s * b2; s = s * b3; break; }
      gc = gn; gn = row4[k + 1];
      EG_FACTORS_LOOSE(gc, a0, a1, a2, a3)
      __builtin_amdgcn_sched_barrier(0);
      s = s * b0; s = s * b1; s = s * b2; s = s * b3;
      ++k;
    }
#undef EG_FACTORS_LOOSE
    asm volatile("" ::: "memory");
  }
  return s;
}

// Small-batch kernel (an episode wave and a helper wave per episode, at most four episodes per CU).  A lone wave issues
// one instruction per turn of its SIMD whatever the instruction is, so this version is written for the fewest
// instructions per generator, and spends LDS (plentiful at four workgroups per CU) to get there:
//  * the packed coordinates of the episode's generators live in LDS (sl.gpk, appended when a generator is placed, both
//    waves read it): no staging, no cell -> (i, j) division per search, four generators per broadcast 128-bit read;
//  * coordinates are kept times four and the factor table has a stride of 16 bytes, so the dot product of the packed
//    difference with itself, accumulated onto the table's address, IS the address of the factor: per generator a packed
//    subtract, the accumulator move, the dot product, the cap, the table read and the multiply;
//  * one straight-line loop body of two groups, so the wait counts stay exact; padding multiplies by 1.0.
// k0: generators [0, k0) are already in s_init (kept product); the first group is masked accordingly.
// kMaskTail (helper wave): entries behind the list are not trusted to be padding — the episode wave may append the next
// generator while this chunk is still being evaluated — so the last group is taken out of the pipeline and masked.
template <bool kMaskTail>
__device__ __forceinline__ double chunk_product_latency(int class_off, int k0, int ngen_s, double s_init, int xy4) {
  short2v cpk; __builtin_memcpy(&cpk, &xy4, 4);
  double s = s_init;
  const int k0_s = __builtin_amdgcn_readfirstlane(k0);
  const int g0 = k0_s >> 2, skip = k0_s & 3;
  const int groups = ((ngen_s + 3) >> 2) - g0;
  if (groups <= 0) return s;
  const int4* row4 = reinterpret_cast<const int4*>(sl.gpk) + g0;
  const char* t0 = reinterpret_cast<const char*>(sl.dr16);
  const int cap = class_off + 16 * kD2Max;
#define EG_FACTORS(g4, f0, f1, f2, f3) { \
    short2v e0, e1, e2, e3; int q0, q1, q2, q3; \
    { short2v t; __builtin_memcpy(&t, &g4.x, 4); e0 = cpk - t; __builtin_memcpy(&t, &g4.y, 4); e1 = cpk - t; \
      __builtin_memcpy(&t, &g4.z, 4); e2 = cpk - t; __builtin_memcpy(&t, &g4.w, 4); e3 = cpk - t; } \
    __builtin_amdgcn_sched_barrier(0); \
    q0 = __builtin_amdgcn_sdot2(e0, e0, class_off, false); q1 = __builtin_amdgcn_sdot2(e1, e1, class_off, false); \
    q2 = __builtin_amdgcn_sdot2(e2, e2, class_off, false); q3 = __builtin_amdgcn_sdot2(e3, e3, class_off, false); \
    __builtin_amdgcn_sched_barrier(0); \
    q0 = q0 < cap ? q0 : cap; q1 = q1 < cap ? q1 : cap; q2 = q2 < cap ? q2 : cap; q3 = q3 < cap ? q3 : cap; \
    __builtin_amdgcn_sched_barrier(0); \
    f0 = *reinterpret_cast<const double*>(t0 + q0); f1 = *reinterpret_cast<const double*>(t0 + q1); \
    f2 = *reinterpret_cast<const double*>(t0 + q2); f3 = *reinterpret_cast<const double*>(t0 + q3); }
  double a0, a1, a2, a3, b0, b1, b2, b3;
  if constexpr (kMaskTail) {
    const int last = groups - 1, tail = ngen_s & 3;
    if (last > 0) {      // groups [0, last) through the pipeline (its look-ahead reads are never multiplied)
      int4 ga = row4[0], gb4 = row4[1];
      if (skip > 0) ga.x = kGenPad4;
      if (skip > 1) ga.y = kGenPad4;
      if (skip > 2) ga.z = kGenPad4;
      EG_FACTORS(ga, a0, a1, a2, a3)
      const int pairs = last >> 1;
      for (int p = 0; p < pairs; ++p) {
        ga = row4[2 * p + 2];
        EG_FACTORS(gb4, b0, b1, b2, b3)
        __builtin_amdgcn_sched_barrier(0);
        s = s * a0; s = s * a1; s = s * a2; s = s * a3;
        gb4 = row4[2 * p + 3];
        EG_FACTORS(ga, a0, a1, a2, a3)
        __builtin_amdgcn_sched_barrier(0);
        s = s * b0; s = s * b1; s = s * b2; s = s * b3;
      }
      if (last & 1) { s = s * a0; s = s * a1; s = s * a2; s = s * a3; }
    }
    int4 gl = row4[last];
    if (last == 0) { if (skip > 0) gl.x = kGenPad4; if (skip > 1) gl.y = kGenPad4; if (skip > 2) gl.z = kGenPad4; }
    if (tail != 0) { if (tail < 2) gl.y = kGenPad4; if (tail < 3) gl.z = kGenPad4; gl.w = kGenPad4; }
    EG_FACTORS(gl, a0, a1, a2, a3)
    s = s * a0; s = s * a1; s = s * a2; s = s * a3;
    return s;
  }
  int4 ga = row4[0], gb4 = row4[1];
  if (skip > 0) ga.x = kGenPad4;
  if (skip > 1) ga.y = kGenPad4;
  if (skip > 2) ga.z = kGenPad4;
  EG_FACTORS(ga, a0, a1, a2, a3)
  const int pairs = groups >> 1;
  for (int p = 0; p < pairs; ++p) {          // a holds group 2p
    ga = row4[2 * p + 2];
    EG_FACTORS(gb4, b0, b1, b2, b3)
    __builtin_amdgcn_sched_barrier(0);        // the four table reads stay in flight ahead of the multiply chain
    s = s * a0; s = s * a1; s = s * a2; s = s * a3;
    gb4 = row4[2 * p + 3];
    EG_FACTORS(ga, a0, a1, a2, a3)             // (on the last trip of an even count: a group of padding, never multiplied)
    __builtin_amdgcn_sched_barrier(0);
    s = s * b0; s = s * b1; s = s * b2; s = s * b3;
  }
  if (groups & 1) { s = s * a0; s = s * a1; s = s * a2; s = s * a3; }
#undef EG_FACTORS
  return s;
}
// Final score of this lane's candidate (rank r of the sorted list) against the episode's generator list.
// kLatency: small-batch kernel (`table` = byte offset of the radius class inside sl.dr16, xy4 = the candidate's packed
// coordinates); otherwise the throughput kernel (`table` = throughput_table() of the type, `cell`).
template <bool kLatency>
__device__ __forceinline__ double chunk_score(int table, double size_factor, int lane, int ngen_s, int r, double te,
                                              double cf, int cell, int xy4, bool row0_kept = false) {
  double p;
  if constexpr (kLatency) p = chunk_product_latency<false>(table, 0, ngen_s, te, xy4);
  else p = chunk_product(table, lane, 0, ngen_s, te, xy4, 0, row0_kept);
  const double s = (p * cf) * size_factor;
  return r < kCells ? s : 0.0;
}
// ... continuing the product kept from the last search of the same (year, variant) when there is one
template <bool kLatency>
__device__ __forceinline__ double chunk_score(int table, double size_factor, int lane, int ngen_s, int r, double te,
                                              double cf, int cell, int xy4, PrefixCache& cache, int key, int stage) {
  double s = te; int k0 = 0;
  if (cache.key == key && cache.count <= ngen_s) { s = cache.product; k0 = cache.count; }
  if constexpr (kLatency) s = stage ? chunk_product_latency<true>(table, k0, ngen_s, s, xy4) : chunk_product_latency<false>(table, k0, ngen_s, s, xy4);
  else s = chunk_product(table, lane, k0, ngen_s, s, xy4, stage);
  cache.product = s; cache.key = key; cache.count = ngen_s;
  s = (s * cf) * size_factor;
  return r < kCells ? s : 0.0;
}

struct ChunkBest { double score, m03; int cell; };
// maximum of a chunk's scores, ties to the lowest cell, with the winner's 0.03 * mean settlement opinion
template <bool kLatency>
__device__ __forceinline__ ChunkBest chunk_reduce(double s, int cell, double m03) {
  ChunkBest b;
  // Scores are not negative, so their order is the order of their bit patterns.  Almost always a single lane holds the
  // largest high word: that lane is the winner and nothing else has to be reduced.  (Small-batch kernel: fewer instructions on
  // the critical path.  Throughput kernels: fewer vector instructions — they are bound by vector issue; 16 384 sampled episodes
  // 1.002 -> 0.982 ms, profiles/r04_ab_notes.log r04za; the branch-free form had measured equal before the kernel was that tight.)
  const unsigned hi = (unsigned)__double2hiint(s);
  const unsigned mh = wave_max_u32(hi);
  const unsigned long long top = __ballot(hi == mh);
  if (__popcll(top) == 1) {
    const int w = __ffsll((long long)top) - 1;
    b.score = readlane_f64(s, w); b.cell = __builtin_amdgcn_readlane(cell, w); b.m03 = readlane_f64(m03, w);
    return b;
  }
  const unsigned ml = wave_max_u32(hi == mh ? (unsigned)__double2loint(s) : 0u);
  b.score = __hiloint2double((int)mh, (int)ml);
  int win_c = s == b.score ? cell : kCells;
  const unsigned long long holders = __ballot(s == b.score);
  if (__popcll(holders) == 1) win_c = __builtin_amdgcn_readlane(cell, __ffsll((long long)holders) - 1);
  else {
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) { const int other = __shfl_xor(win_c, sh); win_c = other < win_c ? other : win_c; }
  }
  b.cell = win_c;
  const unsigned long long owner = __ballot(s == b.score && cell == win_c);   // the lane that holds the winner
  b.m03 = readlane_f64(m03, __ffsll((long long)owner) - 1);
  return b;
}

// Workgroup barrier that orders LDS only: the episode's output stores and table loads stay in flight across it.
__device__ __forceinline__ void wg_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- lists beyond the on-chip window (long-replay variant only) ------------------------------------------------------
// The reference's lists are Vecs; a replay episode applies and records every action twice, so in a training loop
// the replayed lists double whenever a replay episode becomes the best strategy: 468 generators after seven updates, 965
// actions in the best list soon after at larger batches.  The first kLdsGens generators / offsets of an episode live in LDS
// (gcell / gbm / opack); a long-replay episode that grows past that goes on in its own output record — it appends every
// generator and offset there anyway (gen_cell / gen_pack / off_pack, EG_MAX_GENS entries) — and reads those entries back in
// blocks of 64 wherever a list is walked: the year-start folds, the exact product of a candidate, a field that joins, the
// exact scan.  Blocks start at multiples of 64 and kLdsGens is one, so a block is either all window or all tail.
// The entries were stored by this wave itself: they are read past the CU's L1 (agent scope), like the penalty field.
struct ListTail { unsigned long long gen_cell, gen_pack, off_pack; };      // addresses: they cross non-inlined calls as integers
typedef unsigned short __attribute__((address_space(1)))* GlobalU16;
__device__ __forceinline__ int tail_u16(unsigned long long addr, int i) {
  return (int)__hip_atomic_load((GlobalU16)addr + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// cell of generator g (a lane each), from the window or from the record; `far`: what a lane beyond the list gets
__device__ __forceinline__ int list_cell(unsigned long long tail_cells, int gb, int lane, int ngen_s, int far) {
  if (gb + lane >= ngen_s) return far;
  return gb < kLdsGens ? (int)(sm.gcell[gb + lane] & 0xFFF) : tail_u16(tail_cells, gb + lane);
}

// ---- aggregates at the start of a year: existing plant first (host tables), then every generator in list order.
//      The lanes gather the per-generator terms in parallel (year_gather: requests only); the sums are then folded lane
//      by lane with readlane, i.e. in list order (year_fold).  Output / CO2 terms of a plant never change (delays off),
//      so the class sums carry over from the end of last year whenever the existing-plant prefix did (`carry`), bit for
//      bit; otherwise they are folded here as well. ----
// acc + x[lane J of the row], in every lane of that row.  gfx90a and later have a DPP form of the VOP2 double-precision
// multiply-accumulate (row_newbcast only; v_add_f64 is VOP3 and has none): acc = x[J] * 1.0 + acc is that addition — the product is
// exact and the sum is rounded once — so a sum in list order over 16 lanes is 16 instructions, instead of 16 x (two v_readlane + one
// add) through scalar registers.
// Inline assembly is opaque to the hazard recogniser: a DPP read needs wait states after a VALU write of the register it reads (2) or
// of EXEC (5).  The s_nop and the sixteen accumulations are therefore ONE asm statement with x as an input: whatever produces x, and
// whatever the compiler schedules in front of the statement, is followed by the five wait states before the first DPP read — two
// statements (a bare s_nop, then the DPP instructions) left the scheduler free to put the producer of x between them.
#define EG_FMAC_DPP(acc_, x_, j_) "v_fmac_f64_dpp " acc_ ", " x_ ", %[one] row_newbcast:" #j_ " row_mask:0xf bank_mask:0xf\n\t"
// acc (the same value in every lane) + x[16 row + 0] + ... + x[16 row + 15], added in that order; valid in the lanes of each row for
// that row's sixteen values
__device__ __forceinline__ double fold_row16(double acc, double x) {
  const double one = 1.0;
  asm("s_nop 4\n\t"
      EG_FMAC_DPP("%[a]", "%[x]", 0) EG_FMAC_DPP("%[a]", "%[x]", 1) EG_FMAC_DPP("%[a]", "%[x]", 2) EG_FMAC_DPP("%[a]", "%[x]", 3)
      EG_FMAC_DPP("%[a]", "%[x]", 4) EG_FMAC_DPP("%[a]", "%[x]", 5) EG_FMAC_DPP("%[a]", "%[x]", 6) EG_FMAC_DPP("%[a]", "%[x]", 7)
      EG_FMAC_DPP("%[a]", "%[x]", 8) EG_FMAC_DPP("%[a]", "%[x]", 9) EG_FMAC_DPP("%[a]", "%[x]", 10) EG_FMAC_DPP("%[a]", "%[x]", 11)
      EG_FMAC_DPP("%[a]", "%[x]", 12) EG_FMAC_DPP("%[a]", "%[x]", 13) EG_FMAC_DPP("%[a]", "%[x]", 14) EG_FMAC_DPP("%[a]", "%[x]", 15)
      : [a] "+v"(acc) : [x] "v"(x), [one] "v"(one));
  return acc;
}
// two such sums side by side (the two chains are independent: interleaved, neither waits for the other's accumulation)
__device__ __forceinline__ void fold2_row16(double& a, double xa, double& b, double xb) {
  const double one = 1.0;
#define EG_FMAC2(j_) EG_FMAC_DPP("%[a]", "%[xa]", j_) EG_FMAC_DPP("%[b]", "%[xb]", j_)
  asm("s_nop 4\n\t"
      EG_FMAC2(0) EG_FMAC2(1) EG_FMAC2(2) EG_FMAC2(3) EG_FMAC2(4) EG_FMAC2(5) EG_FMAC2(6) EG_FMAC2(7)
      EG_FMAC2(8) EG_FMAC2(9) EG_FMAC2(10) EG_FMAC2(11) EG_FMAC2(12) EG_FMAC2(13) EG_FMAC2(14) EG_FMAC2(15)
      : [a] "+v"(a), [b] "+v"(b) : [xa] "v"(xa), [xb] "v"(xb), [one] "v"(one));
#undef EG_FMAC2
}

struct YearTerms { double2 g_cc; double g_m03, g_t12, o_v, o_c; int g_t; };
struct YearSums { double gcost, optot, offs, ocost, co2, tg, ig, sg; int opcnt; };

__device__ __forceinline__ void year_gather_gens(const DevTables& T, int lane, int yi, int base, int ngen_s, YearTerms& t) {
  const double* ccy = T.cc() + (unsigned)yi * kTypes * kYears * kMults * 2;
  const int g = base + lane;
  const bool valid = g < ngen_s;
  const int gc = valid ? sm.gcell[g] : 0, bm = valid ? sm.gbm[g] : 0;
  const int cell = gc & 0xFFF, b = bm & 31, m = bm >> 5;
  t.g_t = gc >> 12;
  t.g_cc = *reinterpret_cast<const double2*>(ccy + ((unsigned)(t.g_t * kYears + b) * kMults + m) * 2);
  t.g_m03 = T.m03()[cell]; t.g_t12 = T.t12()[(unsigned)yi * kTypes + t.g_t];
}
__device__ __forceinline__ void year_gather_offsets(const DevTables& T, int lane, int yi, int base, int noff_s, YearTerms& t) {
  const int k = base + lane;
  const int p = k < noff_s ? sm.opack[k] : 0;
  const int ot = p & 15, b = (p >> 4) & 31, m = p >> 9;
  t.o_v = T.offv()[((unsigned)yi * kOffsetTypes + ot) * kYears + b]; t.o_c = T.offc()[((unsigned)yi * kOffsetTypes + ot) * kMults + m];
}
// the same requests for a block of the list's tail (long-replay variant): the entries come from the episode's record
__device__ __forceinline__ void year_gather_gens_tail(const DevTables& T, int lane, int yi, int base, int ngen_s, YearTerms& t, const ListTail& tail) {
  const double* ccy = T.cc() + (unsigned)yi * kTypes * kYears * kMults * 2;
  const int g = base + lane;
  const bool valid = g < ngen_s;
  const int cell = valid ? tail_u16(tail.gen_cell, g) : 0, pk = valid ? tail_u16(tail.gen_pack, g) : 0;      // type | build year << 4 | mult << 9
  const int b = (pk >> 4) & 31, m = pk >> 9;
  t.g_t = pk & 15;
  t.g_cc = *reinterpret_cast<const double2*>(ccy + ((unsigned)(t.g_t * kYears + b) * kMults + m) * 2);
  t.g_m03 = T.m03()[cell]; t.g_t12 = T.t12()[(unsigned)yi * kTypes + t.g_t];
}
__device__ __forceinline__ void year_gather_offsets_tail(const DevTables& T, int lane, int yi, int base, int noff_s, YearTerms& t, const ListTail& tail) {
  const int k = base + lane;
  const int p = k < noff_s ? tail_u16(tail.off_pack, k) : 0;
  const int ot = p & 15, b = (p >> 4) & 31, m = p >> 9;
  t.o_v = T.offv()[((unsigned)yi * kOffsetTypes + ot) * kYears + b]; t.o_c = T.offc()[((unsigned)yi * kOffsetTypes + ot) * kMults + m];
}
// the first 64 entries of each list are requested here; year_fold gathers the rest (rare) itself
__device__ __forceinline__ YearTerms year_gather(const DevTables& T, int lane, int yi, int ngen_s, int noff_s) {
  YearTerms t; t.g_cc.x = 0.0; t.g_cc.y = 0.0; t.g_m03 = 0.0; t.g_t12 = 0.0; t.o_v = 0.0; t.o_c = 0.0; t.g_t = 0;
  if (ngen_s > 0) year_gather_gens(T, lane, yi, 0, ngen_s, t);
  if (noff_s > 0) year_gather_offsets(T, lane, yi, 0, noff_s, t);
  return t;
}
// `s` comes in holding the starting values (0 / the existing-plant prefix) and leaves holding the year-start sums
// kLong (long-replay variant): blocks beyond the on-chip window come from the episode's record (`tail`)
template <bool kLong>
__device__ __forceinline__ void year_fold(const DevTables& T, int lane, int yi, int ngen_s, int noff_s, bool carry, YearTerms t,
                                          YearSums& s, const ListTail& tail) {
  for (int base = 0; base < ngen_s; base += kWave) {
    if (!kLong && base > 0) year_gather_gens(T, lane, yi, base, ngen_s, t);      // beyond the first 64 generators (rare)
    const double2 cc = t.g_cc;
    const int ty = t.g_t;
    const double op = (t.g_m03 + t.g_t12) + cc.y;
    // long-replay variant (hundreds of generators, registers to spare): the next block's terms are requested before this block is
    // folded — a memory round trip per block and year otherwise, on the serial path of the batch's longest episodes
    if (kLong && base + kWave < ngen_s) {
      if (base + kWave >= kLdsGens) year_gather_gens_tail(T, lane, yi, base + kWave, ngen_s, t, tail);
      else year_gather_gens(T, lane, yi, base + kWave, ngen_s, t);
    }
    double out = 0.0, co2 = 0.0; int cls = 0;
    if (!carry) { out = sm.type_out[ty]; co2 = sm.type_co2[ty]; cls = (sm.type_info[ty] >> 12) & 3; }
    const int cnt = ngen_s - base < kWave ? ngen_s - base : kWave;
    if (carry) {            // the common year: two independent chains
      int j = 0;
      {      // sixteen generators at a time through the DPP adder (a row padded with +0.0: sums of positive terms, x + 0.0 == x)
        const double cz = lane < cnt ? cc.x : 0.0, oz = lane < cnt ? op : 0.0;
        for (; cnt - j >= 7; j += 16) {      // (below seven generators the scalar path is fewer instructions)
          double ag = s.gcost, ao = s.optot;
          fold2_row16(ag, cz, ao, oz);
          s.gcost = readlane_f64(ag, j); s.optot = readlane_f64(ao, j);      // lane j = the first lane of the row just folded
        }
      }
      for (; j + 4 <= cnt; j += 4) {
        const double c0 = readlane_f64(cc.x, j), c1 = readlane_f64(cc.x, j + 1), c2 = readlane_f64(cc.x, j + 2), c3 = readlane_f64(cc.x, j + 3);
        const double o0 = readlane_f64(op, j), o1 = readlane_f64(op, j + 1), o2 = readlane_f64(op, j + 2), o3 = readlane_f64(op, j + 3);
        s.gcost += c0; s.optot += o0; s.gcost += c1; s.optot += o1; s.gcost += c2; s.optot += o2; s.gcost += c3; s.optot += o3;
      }
      for (; j < cnt; ++j) { s.gcost += readlane_f64(cc.x, j); s.optot += readlane_f64(op, j); }
    } else {
      for (int j = 0; j < cnt; ++j) {
        s.gcost += readlane_f64(cc.x, j);
        s.optot += readlane_f64(op, j);
        const double oj = readlane_f64(out, j);
        const int cj = __builtin_amdgcn_readlane(cls, j);
        s.co2 += readlane_f64(co2, j);
        if (cj == 1) s.ig += oj; else if (cj == 2) s.sg += oj; else s.tg += oj;
      }
    }
    s.opcnt += cnt;
  }
  for (int base = 0; base < noff_s; base += kWave) {
    if (base > 0) {
      if (kLong && base >= kLdsGens) year_gather_offsets_tail(T, lane, yi, base, noff_s, t, tail);
      else year_gather_offsets(T, lane, yi, base, noff_s, t);
    }
    const double ov = t.o_v, oc = t.o_c;
    const int cnt = noff_s - base < kWave ? noff_s - base : kWave;
    for (int j = 0; j < cnt; ++j) { s.offs += readlane_f64(ov, j); s.ocost += readlane_f64(oc, j); }
  }
}
// The same fold over generators [g_from, g_to) and offsets [o_from, o_to) only (every block gathered here): a year's sums
// can be started before the year's lists are final — they only grow at their ends — and finished afterwards, with the
// additions in the same order (helper wave, small-batch kernel).
__device__ __forceinline__ void year_fold_range(const DevTables& T, int lane, int yi, int g_from, int g_to, int o_from, int o_to,
                                                bool carry, YearSums& s) {
  YearTerms t; t.g_cc.x = 0.0; t.g_cc.y = 0.0; t.g_m03 = 0.0; t.g_t12 = 0.0; t.o_v = 0.0; t.o_c = 0.0; t.g_t = 0;
  if (g_from < g_to) year_gather_gens(T, lane, yi, g_from, g_to, t);      // both requests first, one latency
  if (o_from < o_to) year_gather_offsets(T, lane, yi, o_from, o_to, t);
  for (int base = g_from; base < g_to; base += kWave) {
    if (base > g_from) year_gather_gens(T, lane, yi, base, g_to, t);
    const double2 cc = t.g_cc;
    const int ty = t.g_t;
    const double op = (t.g_m03 + t.g_t12) + cc.y;
    double out = 0.0, co2 = 0.0; int cls = 0;
    if (!carry) { out = sm.type_out[ty]; co2 = sm.type_co2[ty]; cls = (sm.type_info[ty] >> 12) & 3; }
    const int cnt = g_to - base < kWave ? g_to - base : kWave;
    for (int j = 0; j < cnt; ++j) {
      s.gcost += readlane_f64(cc.x, j);
      s.optot += readlane_f64(op, j);
      if (!carry) {
        const double oj = readlane_f64(out, j);
        const int cj = __builtin_amdgcn_readlane(cls, j);
        s.co2 += readlane_f64(co2, j);
        if (cj == 1) s.ig += oj; else if (cj == 2) s.sg += oj; else s.tg += oj;
      }
    }
    s.opcnt += cnt;
  }
  for (int base = o_from; base < o_to; base += kWave) {
    if (base > o_from) year_gather_offsets(T, lane, yi, base, o_to, t);
    const double ov = t.o_v, oc = t.o_c;
    const int cnt = o_to - base < kWave ? o_to - base : kWave;
    for (int j = 0; j < cnt; ++j) { s.offs += readlane_f64(ov, j); s.ocost += readlane_f64(oc, j); }
  }
}
// starting values of the sums for year yi: zero, or the existing-plant prefix of that year (class sums only when they
// do not carry over)
// The helper wave asks for NEXT year's values, which are not in LDS yet: it reads the tables; the episode wave reads the
// copies that came with this year's policy block.
__device__ __forceinline__ YearSums year_sums_init(const DevTables& T, int yi) {
  YearSums s;
  s.gcost = 0.0; s.ocost = 0.0; s.offs = 0.0; s.optot = T.pre_optot()[yi]; s.opcnt = T.pre_opcnt()[yi];
  s.co2 = T.pre_co2()[yi]; s.tg = T.pre_tg()[yi]; s.ig = T.pre_ig()[yi]; s.sg = T.pre_sg()[yi];
  return s;
}
__device__ __forceinline__ YearSums year_sums_init_current() {
  YearSums s;
  const double* ys = sm.pol + snap::kPolYear;
  s.gcost = 0.0; s.ocost = 0.0; s.offs = 0.0; s.optot = ys[4]; s.opcnt = (int)ys[9];
  s.co2 = ys[0]; s.tg = ys[1]; s.ig = ys[2]; s.sg = ys[3];
  return s;
}

// Helper waves (kHelpers > 0, small batches only).  At B <= 4 x CUs every SIMD holds a single episode wave that is
// latency-bound, and a launch lasts as long as its slowest episode, whose time is dominated by placement searches that
// need several chunks.  A helper wave per episode evaluates chunk 1 of every search while the episode wave
// evaluates chunk 0; the episode wave merges the two maxima in chunk order with the same tie rule, so the winner is the
// one the sequential scan finds (a chunk the sequential scan would not have reached only holds candidates whose
// unpenalised score is already below the best, so evaluating it changes nothing).
// One helper, not more: two-wave workgroups leave the CUs a third empty, and the dispatcher needs that slack — with
// exactly-full CUs (three-wave workgroups, 4 per CU) a few workgroups of every launch were queued behind a full CU,
// started late and stretched the launch by a third (scripts/bench_tail.py shows the workgroups-per-CU histogram).
// Letting the helper run further rounds (chunks 3, 5, ...) was measured too and bought nothing.
//   protocol: episode wave writes cmd[seq & 1] and meets the helpers at a barrier; helper h evaluates chunk h and
//   publishes {result, flag = seq}; the episode wave reads a helper's result only if that chunk's bound still reaches
//   the best score, after spinning on its flag.  The next barrier cannot complete before every helper is back.
// The helper reads the episode's generator list (sl.gpk) while the episode wave may already be appending the next
// generator: it masks what lies behind the count it was given (chunk_product_latency<true>).  At the end of a year it
// also keeps the year's starting sums ahead of the episode wave (kCmdYear: the sums of the year after next are started
// over the lists as they are, the next year's — started a year ago — only receive what was added since).
__device__ __forceinline__ void helper_loop(const DevTables& T, int lane, int h) {
  const double size_factor = T.size_factor;
  PrefixCache cache = {0.0, -1, 0};
  YearSums pre; int pre_year = -1, pre_g = 0, pre_o = 0;      // sums of year pre_year over generators [0, pre_g) / offsets [0, pre_o)
  pre.gcost = pre.optot = pre.offs = pre.ocost = pre.co2 = pre.tg = pre.ig = pre.sg = 0.0; pre.opcnt = 0;
  for (uint32_t sq = 1;; ++sq) {
    wg_barrier_lds();
    const int c0 = __builtin_amdgcn_readfirstlane(sm.cmd[sq & 1][0]);
    const int ngen_s = __builtin_amdgcn_readfirstlane(sm.cmd[sq & 1][1]);
    if (c0 < 0) return;
    if (c0 & kCmdYear) {      // next year's starting sums (year_fold), while the episode wave closes the current year
      const int yi = c0 & 31, ngen = ngen_s & 0xFFFF, noff = (ngen_s >> 16) & 0xFFFF;
      const bool carry = ((c0 >> 8) & 1) != 0;
      // The sums of this year were started a year ago over the lists as they were then (see below): only what has been
      // added since is folded now, so the episode wave finds them ready.
      YearSums ys; int g_from = 0, o_from = 0;
      if (pre_year == yi) { ys = pre; g_from = pre_g; o_from = pre_o; }
      else ys = year_sums_init(T, yi);
      year_fold_range(T, lane, yi, g_from, ngen, o_from, noff, carry, ys);
      if (lane == 0) {
        sm.ysum[0] = ys.gcost; sm.ysum[1] = ys.optot; sm.ysum[2] = ys.offs; sm.ysum[3] = ys.ocost;
        sm.ysum[4] = ys.co2; sm.ysum[5] = ys.tg; sm.ysum[6] = ys.ig; sm.ysum[7] = ys.sg; sm.ysum_opcnt = ys.opcnt;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if defined(EG_TEST_DROP_FLAG) && EG_TEST_DROP_FLAG == 2      // negative build: the sums of 2030 are never announced
      if (yi == 5) continue;
#endif
      if (lane == 0) *(volatile uint32_t*)&sm.yflag = sq;
      // ... and the year after that is started over the lists as they are now (they only grow at their ends)
      if (yi + 1 < kYears) {
        pre = year_sums_init(T, yi + 1);
        year_fold_range(T, lane, yi + 1, 0, ngen, 0, noff, ((c0 >> 9) & 1) != 0, pre);
        pre_year = yi + 1; pre_g = ngen; pre_o = noff;
      }
      continue;
    }
    const int yi = c0 & 31, v = (c0 >> 8) & 15, rc = (c0 >> 12) & 15;
    const int r = h * kWave + lane;
#ifdef EG_STAMPS
    const unsigned long long th0 = __builtin_readcyclecounter();
#endif
    const PsRec c = T.ps()[(size_t)(yi * kMaxVariants + v) * kPsStride + r];
#ifdef EG_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long th1 = __builtin_readcyclecounter();
#endif
    const double s = chunk_score<true>(rc * (kD2Stride * 16), size_factor, lane, ngen_s, r, c.te, c.cf, (int)c.cell, (int)c.pad, cache, (yi << 8) | v, 1);
#ifdef EG_STAMPS
    const unsigned long long th2 = __builtin_readcyclecounter();
#endif
    const ChunkBest b = chunk_reduce<true>(s, (int)c.cell, c.m03);
#ifdef EG_STAMPS
    if (lane == 0) { sm.hdbg[h - 1][0] = th1 - th0; sm.hdbg[h - 1][1] = th2 - th1; sm.hdbg[h - 1][2] = __builtin_readcyclecounter() - th2; }
#endif
    if (lane == 0) { sm.hres[h - 1].score = b.score; sm.hres[h - 1].m03 = b.m03; sm.hres[h - 1].cell = b.cell; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if defined(EG_TEST_DROP_FLAG) && EG_TEST_DROP_FLAG == 1      // negative build: search results of 2027 are never announced
    if (yi == 2) continue;
#endif
    if (lane == 0) *(volatile uint32_t*)&sm.hflag[h - 1] = sq;
  }
}

// `between` is called once the candidate records are requested (and, in the small-batch kernel, the helper has its
// command): whatever the caller has to do before it needs the result goes there and runs under the records' latency.
template <int kHelpers, bool kRow0Kept = false, class Between>
__device__ __forceinline__ int place_search(const DevTables& T, int lane, int yi, int type, int ngen, double* best_score,
                                            double* best_m03, PrefixCache& cache0, Between&& between, int& nchunks,
                                            uint32_t* seq = nullptr, unsigned long long* stamps = nullptr) {
  const int info = __builtin_amdgcn_readfirstlane(sm.type_info[type]);      // uniform: list address arithmetic on the scalar unit
  const int v = info & 15, rc = (info >> 4) & 15;
  const PsRec* __restrict__ list = T.ps() + (size_t)(yi * kMaxVariants + v) * kPsStride;
  // factor table of the radius class: byte offset inside sl.dr16 (small-batch kernel) / inside the LDS block (sm.dr)
  const int table = kHelpers > 0 ? rc * (kD2Stride * 16) : throughput_table(info);
  const double size_factor = T.size_factor;
#ifdef EG_PROBE_NO_GEN_LOOP      // diagnostic build only (profiles/r04_ab_notes.log r04u): the searches fold no generator — what the generator loops cost
  const int ngen_s = 0;
#else
  const int ngen_s = __builtin_amdgcn_readfirstlane(ngen);
#endif
  constexpr int kChunks = (kCells + kWave - 1) / kWave;
  double best = 0.0, m03w = 0.0; int best_c = kCells;
  int first = 0;
  bool more = true, lost = false;
  // chunk 0 is loaded here, chunk k+1 while chunk k is being evaluated
  PsRec c = list[lane];
  nchunks += 1 + (kHelpers > 0 ? kHelpers + 1 : 0);      // chunk 0; small-batch kernel: the helpers' chunks and the look-ahead one
  if constexpr (kHelpers > 0) {
    // lanes 0..kHelpers: unpenalised score of the first candidate of chunks 1..kHelpers+1 = the bound of that chunk
    const int bl = (lane <= kHelpers ? lane + 1 : 1) * kWave;
    const PsRec cb = list[bl];
    const double bound = (cb.te * cb.cf) * size_factor;
    *seq += 1;
    const uint32_t sq = *seq;
    if (lane == 0) { sm.cmd[sq & 1][0] = yi | (v << 8) | (rc << 12); sm.cmd[sq & 1][1] = ngen_s; }
    wg_barrier_lds();
    between();
#ifdef EG_STAMPS
    if (stamps) stamps[8] += 1;
    const unsigned long long tg0 = __builtin_readcyclecounter();
#endif
    const double s0 = chunk_score<true>(table, size_factor, lane, ngen_s, lane, c.te, c.cf, (int)c.cell, (int)c.pad, cache0, (yi << 8) | v, 0);
#ifdef EG_STAMPS
    const unsigned long long tg1 = __builtin_readcyclecounter();
    if (stamps) stamps[9] += tg1 - tg0;
#endif
    const ChunkBest b0 = chunk_reduce<true>(s0, (int)c.cell, c.m03);
    if (b0.score > 0.0) { best = b0.score; best_c = b0.cell; m03w = b0.m03; }
    // the records of the first chunk the episode wave would evaluate itself are requested before it waits for the
    // helper: when chunk 1 is needed, that chunk usually is as well
    c = list[(kHelpers + 1) * kWave + lane];
    for (int h = 1; h <= kHelpers && more; ++h) {
      if (!(readlane_f64(bound, h - 1) >= best)) { more = false; break; }
#ifdef EG_STAMPS
      const unsigned long long tw0 = __builtin_readcyclecounter();
#endif
      // (on a timeout the search carries on with whatever the result slot holds — values are only compared — and reports
      //  the fault at its end: an early return here cost the episode loop 18 spilled VGPRs)
      for (int spins = 0; __builtin_amdgcn_readfirstlane((int)*(volatile uint32_t*)&sm.hflag[h - 1]) != (int)sq; ++spins) {
        __builtin_amdgcn_s_sleep(1);
        if (spins >= kSpinCap) { lost = true; break; }
      }
      asm volatile("" ::: "memory");
#ifdef EG_STAMPS
      if (stamps) { stamps[6] += __builtin_readcyclecounter() - tw0; stamps[27] += sm.hdbg[h - 1][0]; stamps[28] += sm.hdbg[h - 1][1]; stamps[29] += sm.hdbg[h - 1][2]; stamps[30] += 1; }
#endif
      const double hs = sm.hres[h - 1].score; const int hc = sm.hres[h - 1].cell;
      if (hs > best || (hs == best && hs > 0.0 && hc < best_c)) { best = hs; best_c = hc; m03w = sm.hres[h - 1].m03; }
#ifdef EG_STAMPS
      if (stamps) stamps[8] += 1;
#endif
    }
    if (more && !(readlane_f64(bound, kHelpers) >= best)) more = false;
    first = kHelpers + 1;
#ifdef EG_STAMPS
    if (stamps) stamps[10] += __builtin_readcyclecounter() - tg1;
#endif
  }
  if constexpr (kHelpers == 0) between();
  for (int chunk = first; more && chunk < kChunks; ++chunk) {
    const int r = chunk * kWave + lane;
    const double base = (c.te * c.cf) * size_factor;      // padded with te = 0 beyond the 2601 candidates
    // (throughput kernel: said to be uniform, this is a scalar branch; measured +4.5 % there, -2 % in the small-batch kernel)
    const bool below = !(readlane_f64(base, 0) >= best);
    if (chunk > 0 && (kHelpers > 0 ? below : EG_UNI(below))) break;      // sorted descending: lane 0 holds the chunk's bound
    const double te_cur = c.te, cf_cur = c.cf, m03_cur = c.m03; const int cell_cur = (int)c.cell, xy_cur = (int)c.pad;
    if (chunk + 1 < kChunks) { c = list[r + kWave]; nchunks += 1; }
#ifdef EG_STAMPS
    if (stamps) stamps[8] += 1;
    const unsigned long long tg0 = __builtin_readcyclecounter();
#endif
    // the single-wave kernel keeps the products of chunks 0 and 1, the episode wave of the helper kernel that of chunk 0
    // (the kept products are used by the small-batch kernel only: in the throughput kernel the extra live registers cost
    //  more than the shorter loops give back)
    const double s = chunk_score<(kHelpers > 0)>(table, size_factor, lane, ngen_s, r, te_cur, cf_cur, cell_cur, xy_cur, kRow0Kept);
#ifdef EG_STAMPS
    const unsigned long long tg1 = __builtin_readcyclecounter();
    if (stamps) stamps[9] += tg1 - tg0;
#endif
    if (__any(s > best || (s == best && s > 0.0 && cell_cur < best_c))) {
      const ChunkBest b = chunk_reduce<(kHelpers > 0)>(s, cell_cur, m03_cur);
      if (b.score > best || (b.score == best && b.cell < best_c)) { best = b.score; best_c = b.cell; m03w = b.m03; }
    }
#ifdef EG_STAMPS
    if (stamps) stamps[10] += __builtin_readcyclecounter() - tg1;
#endif
  }
  if (best_score) *best_score = best;
  if (best_m03) *best_m03 = m03w;
  if (lost) return kSearchLost;
  return best > 0.0 ? best_c : -1;
}


// ---- heavy episodes: placement against a long generator list --------------------------------------------------------
// The branch-and-bound scan above evaluates whole chunks exactly, O(generators) per chunk.  That is the right trade for
// the 25-45 generators of a sampled episode (2-5 chunks per search), but a replay episode places hundreds (the
// reference's replay records and applies every action twice, SURVEY Q15: 228-468 generators once a replay episode has
// become the best strategy): the best-scoring cells are all taken, the scan goes 15-41 chunks deep, and such an episode
// took 40x the time of a sampled one — a launch lasts as long as its slowest episode.
// From kHeavyGens generators on, an episode keeps in global memory, per radius class it searches for, the product of the
// penalty factors of all its generators for every cell: field[rc][cell] (year-independent; folded in any order, it only
// serves a bound).  A class's field is built when the episode first searches for a type of that class and kept up to date
// from then on.
// A search then is
//   1. approx(c) = ((te * cf) * size) * field[rc][c] over the sorted candidates — one gather and three multiplications
//      per candidate instead of a pass over the generator list — with the same stop rule as the exact scan
//      (field <= 1, so approx(c) <= base(c));
//   2. every candidate with approx >= M * (1 - 2^-30), M the largest approx, is evaluated EXACTLY: the reference's
//      product in list order (exact_product_chain; chunk_product when there are many ties), first maximum in cell order.
// Exactness: exact(c) and approx(c) are both the real product te * cf * size * prod f rounded at most G + 3 times each,
// so they differ by less than 2 (G + 3) 2^-53 < 2^-42 relative while no intermediate is subnormal; the arg-max of the
// exact scores (and every cell tied with it) therefore lies within 2^-41 of M and is among the candidates, and a cell
// below the threshold cannot reach the exact score of M's holder.  Subnormal ranges (M < 1e-250), more than 64
// candidates, or no free field slot fall back to the exact scan above.  Results are the exact scan's, bit for bit.
constexpr int kHeavyGens = 8;       // (the heavy variant only runs replay episodes: their exact scans go deep almost at once)
constexpr int kFieldStride = 2624;                   // doubles per radius class of a field slot
constexpr int kSearchFallback = -3;
// The three functions below are not inlined (they would cost the episode loop its registers), and a pointer that crosses a
// call loses its address space: they take ADDRESSES and make global-memory pointers of them, so that the loads stay
// global_load / global_store instead of flat ones.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef const u32x4 __attribute__((address_space(1)))* GlobalVec4;
typedef double __attribute__((address_space(1)))* GlobalF64;
typedef const double __attribute__((address_space(1)))* GlobalF64c;
__device__ __forceinline__ PsRec load_rec(unsigned long long list_addr, int i) {
  const GlobalVec4 p = (GlobalVec4)(list_addr + (unsigned long long)(unsigned)i * sizeof(PsRec));
  const u32x4 lo = p[0], hi = p[1];
  PsRec r;
  __builtin_memcpy(&r, &lo, 16); __builtin_memcpy(reinterpret_cast<char*>(&r) + 16, &hi, 16);
  return r;
}
typedef const int __attribute__((address_space(1)))* GlobalI32c;
__device__ __forceinline__ double field_load(GlobalF64 p) {      // past the CU's L1: the wave wrote this entry itself
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// q: squared cell distance, already capped (kLatency: at kD2Max; otherwise at the class's own cap, table >> 16).  `table`: the
// class's throughput_table() — the small-batch kernel (kLatency) has the full table at a 16-byte stride and goes by `rc`.
template <bool kLatency>
__device__ __forceinline__ double factor_by_q(int rc, int table, int q) {
  if constexpr (kLatency) return sl.dr16[2 * (rc * kD2Stride + q)];
  else return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(&sm) + ((table & 0xFFFF) + q * 8));
}
template <bool kLatency>
__device__ __forceinline__ int factor_cap(int table) { return kLatency ? kD2Max : table >> 16; }
// one slot of the pool for this launch, or -2 (pool exhausted / absent).  The claim word holds launch epoch << 20 | count,
// so no launch has to reset it: the first claim of a launch swaps the new epoch in with a zero count, every claim is then
// ONE fetch-and-add (a compare-and-swap loop here made 1 638 episodes queue behind each other for milliseconds).
__device__ __forceinline__ int heavy_claim(const DevTables& T, int lane) {
  int slot = -2;
  if (lane == 0 && T.heavy != nullptr) {
    unsigned* w = T.heavy_claim;
    const unsigned epoch = T.heavy_epoch & 0xFFFu;
    const unsigned old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((old >> 20) != epoch) atomicCAS(w, old, epoch << 20);      // whoever comes first; a failure means somebody else did it
    const unsigned v = atomicAdd(w, 1u);
    if ((v >> 20) == epoch && (v & 0xFFFFFu) < T.heavy_slots) slot = (int)(v & 0xFFFFFu);
  }
  return __builtin_amdgcn_readfirstlane(slot);
}
// field[rc][c] *= d/R of a generator at `cell`, for every class and every cell closer than the class radius.  The cells
// concerned are the same for every generator up to a translation: the host lists them once (eg_api.cpp, tab::hv_box: 978
// entries {di, dj, squared distance, class} for the reference's six radii, padded to 1024 = 16 per lane) and the heavy
// variant keeps the list in LDS.  One memory round trip: every lane requests its 16 field entries, then multiplies and
// stores.  Entries are distinct, so the order is free.
// There is NO control flow in here, on purpose.  With a branch around every entry (skip cells outside the grid, skip
// classes this episode does not search) the compiler can no longer count which loads are outstanding: it waited for
// vmcnt(0) before every multiplication — that is, for the previous STORE to come back from L2 — and for every list
// entry before the next was requested: 8-12 thousand cycles per generator, measured, for what is one round trip.  So an
// entry that falls outside the grid goes to a spare entry behind its class's cells (kFieldStride > kCells; never read),
// and the padding of the list multiplies a cell no class reaches by exactly 1.0.  Classes the episode keeps no field for
// are not in its list at all: heavy_pack_list rewrites the list in LDS whenever a class joins (bytes count as much as
// round trips here: a thousand and more such episodes share the L2), and the update comes in four lengths.
constexpr int kBoxEntries = 1024;
constexpr uint32_t kBoxPadding = 145u << 10;      // class 0, di = dj = -16, q = 145: a factor of exactly 1.0 (sm.dr is padded with it)
struct __align__(16) SmemHeavy {
  uint32_t box[kBoxEntries];               // di + 16 | (dj + 16) << 5 | q << 10 (9 bits; throughput kernels: place in sm.dr) | class << 19
};
__shared__ SmemHeavy sh;
// Throughput kernel: a copy of the first eight entries per lane only — the list of up to three or four radius classes, the usual case,
// 2 KB: with it the variant stays at nine LDS granules, which still lets sixteen waves share a CU beside the lean grid (eleven granules
// did not) —; entries beyond come from tab::hv_lists itself.
constexpr int kBoxLds = 8 * kWave;
struct __align__(16) SmemHeavyTp { uint32_t box[kBoxLds]; };
__shared__ SmemHeavyTp sh2;
typedef const uint32_t __attribute__((address_space(1)))* GlobalU32c;
// sh.box = the entries of the radius classes in `classes` (the host's list is sorted by class; its words 1024..1030 are where
// each class starts), padded to a multiple of four per lane; returns that multiple (1..4)
// Throughput kernels (!kLatency): the squared distance of an entry is replaced by the entry's place in the compact factor table
// sm.dr (`meta_addr`: tab::dr_meta), so that heavy_add reads the factor without knowing the class's offset and cap.
template <bool kLatency>
__device__ __noinline__ int heavy_pack_list(unsigned long long box_addr, unsigned long long meta_addr, int lane, int classes) {
  const GlobalU32c src = (GlobalU32c)box_addr;
  const GlobalU32c meta = (GlobalU32c)meta_addr;
  auto place = [&](uint32_t en) -> uint32_t {
    if constexpr (kLatency) return en;
    const int rc = (int)(en >> 19), q = (int)((en >> 10) & 511u);
    const int off = (int)meta[rc], cap = (int)meta[8 + rc];
    return (en & ~(511u << 10)) | ((uint32_t)(off + (q < cap ? q : cap)) << 10);
  };
  int n = 0;
  for (int rc = 0; rc < kRadiusClasses; ++rc) {
    if (!((classes >> rc) & 1)) continue;
    const int s0 = __builtin_amdgcn_readfirstlane((int)src[kBoxEntries + rc]), s1 = __builtin_amdgcn_readfirstlane((int)src[kBoxEntries + rc + 1]);
    for (int i = s0 + lane; i < s1; i += kWave) sh.box[n + i - s0] = place(src[i]);
    n += s1 - s0;
  }
  const int padded = (n + 4 * kWave - 1) & ~(4 * kWave - 1);
  for (int i = n + lane; i < padded; i += kWave) sh.box[i] = place(kBoxPadding);
  wave_sync();
  return padded / (4 * kWave);
}
// Throughput kernel: inlined.  A function that is not waits for its stores to be acknowledged before it returns (s_waitcnt vmcnt(0) ahead of
// s_setpc), and waits on entry for whatever its caller had in flight: as a call, the update cost the episode its own loads, then its
// stores' round trip, and before that the cost terms the caller had just requested — three exposed latencies where one is needed.
// `list_addr` (throughput kernel): the entry list of the episode's classes, packed by the host for every subset of classes
// (tab::hv_lists); its first eight entries per lane are in LDS (sh2.box, copied when a class joins).  The small-batch kernel, with LDS
// to spare, packs its own list (sh.box).
template <bool kLatency, int kPerLane>
__device__ __forceinline__ void heavy_add_body(unsigned long long field_addr, unsigned long long list_addr, int lane, int cell, int first) {
#ifdef EG_STAMPS
  const unsigned long long ts0 = __builtin_readcyclecounter();
#endif
  const int gi = cell / kGrid, gj = cell - gi * kGrid;
  const GlobalF64 base = (GlobalF64)field_addr;
  double val[kPerLane], fac[kPerLane]; int off[kPerLane]; uint32_t ens[kPerLane];
#pragma unroll
  for (int k = 0; k < kPerLane; ++k) {
    if constexpr (kLatency) ens[k] = sh.box[(first + k) * kWave + lane];
    else if ((first + k) * kWave < kBoxLds) ens[k] = sh2.box[(first + k) * kWave + lane];      // (`first` is a constant at every call site)
    else ens[k] = ((GlobalU32c)list_addr)[(first + k) * kWave + lane];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (all list entries in one LDS round trip, not one after the other)
#pragma unroll
  for (int k = 0; k < kPerLane; ++k) {
    const uint32_t en = ens[k];
    const int ci = gi + (int)(en & 31u) - 16, cj = gj + (int)((en >> 5) & 31u) - 16, rc = (int)(en >> 19);
    const bool inside = (unsigned)ci < (unsigned)kGrid && (unsigned)cj < (unsigned)kGrid;
    fac[k] = factor_by_q<kLatency>(rc, (int)offsetof(Smem, dr), (int)((en >> 10) & 511u));      // (throughput kernels: the entry's place in sm.dr, see heavy_pack_list)
    off[k] = rc * kFieldStride + (inside ? ci * kGrid + cj : kCells);
    val[k] = field_load(base + off[k]);
  }
#pragma unroll
  for (int k = 0; k < kPerLane; ++k) base[off[k]] = val[k] * fac[k];
  // (the stores are left in flight: whoever reads the field next — place_heavy, heavy_build_class — waits for them first)
#ifdef EG_STAMPS
  if (lane == 0) sm.hdbg[0][3] += __builtin_readcyclecounter() - ts0;
#endif
}
template <bool kLatency, int kPerLane>
__device__ __noinline__ void heavy_add(unsigned long long field_addr, int lane, int cell, int first) { heavy_add_body<kLatency, kPerLane>(field_addr, 0ull, lane, cell, first); }
// A class joins: its field = for every cell the product of the factors of the generators placed so far (any order: the
// field only serves a bound).  The field of the 41 blocks of 64 cells sits in registers while the generators pass by; a
// generator only touches the blocks whose rows come within `reach` of its own row (a scalar test per block).
// (`tail_cells`: the cells of the generators beyond the on-chip window, ListTail)
template <bool kLatency>
__device__ __noinline__ void heavy_build_class(unsigned long long class_addr, unsigned long long tail_cells, int lane, int rc, int table, int reach, int ngen) {
  const GlobalF64 f = (GlobalF64)class_addr;
  constexpr int kChunks = (kCells + kWave - 1) / kWave;
  // blocks per pass over the generator list: all of them with registers to spare (small-batch kernel); six passes of seven on the
  // throughput kernel's 72 registers (fourteen per pass spilled 17 of them: with seven the long-replay kernel needs no scratch memory
  // at all) — it happens six times an episode at most
  constexpr int kPer = kLatency ? kChunks : 7;
  const int cap = factor_cap<kLatency>(table);
  const int ngen_s = __builtin_amdgcn_readfirstlane(ngen);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (a field update of the other classes may still be in flight)
#pragma nounroll
  for (int c0 = 0; c0 < kChunks; c0 += kPer) {
    double p[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) p[k] = 1.0;
    for (int gb = 0; gb < ngen_s; gb += kWave) {      // the list in blocks of 64, a generator per lane; then one generator at a time
      const int mine = list_cell(tail_cells, gb, lane, ngen_s, 0);
      const int cnt = ngen_s - gb < kWave ? ngen_s - gb : kWave;
      for (int j = 0; j < cnt; ++j) {
        const int gc = __builtin_amdgcn_readlane(mine, j);
        const int gi = gc / kGrid, gj = gc - gi * kGrid;
        const int row_lo = gi - reach, row_hi = gi + reach;
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
          const int ch = c0 + k;      // (uniform; a constant when there is one pass)
          if ((ch * kWave + kWave - 1) / kGrid < row_lo || (ch * kWave) / kGrid > row_hi) continue;      // uniform: scalars against scalars
          const int cell = ch * kWave + lane;
          const int ci = cell / kGrid, cj = cell - ci * kGrid;
          int q = (ci - gi) * (ci - gi) + (cj - gj) * (cj - gj);
          q = q < cap ? q : cap;
          p[k] = p[k] * factor_by_q<kLatency>(rc, table, q);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < kPer; ++k) { const int cell = (c0 + k) * kWave + lane; if (cell < kCells) f[cell] = p[k]; }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// The reference's product for ONE candidate cell: te times the factor of every generator in list order.  The lanes take a
// generator each for the factors (64 at a time).  A generator at or beyond the radius has the factor 1.0 and x * 1.0 == x
// exactly, so only the generators inside the radius are multiplied in — in list order, as a sequential chain of v_mul_f64
// fed by v_readlane.  The winner of a search is a cell that few generators reach: a handful of multiplications instead
// of one per generator (chunk_product, which evaluates 64 different candidates at once, cannot skip anything).
template <bool kLatency>
__device__ __forceinline__ double exact_product_chain(int rc, int table, int ngen_s, double te, int cell, int lane, unsigned long long tail_cells) {
  const int ci = cell / kGrid, cj = cell - ci * kGrid;
  const int cap = factor_cap<kLatency>(table);
  double s = te;
  for (int gb = 0; gb < ngen_s; gb += kWave) {
    double f = 1.0;
    if (gb + lane < ngen_s) {
      const int gc = list_cell(tail_cells, gb, lane, ngen_s, 0);
      const int gi = gc / kGrid, gj = gc - gi * kGrid;
      int q = (ci - gi) * (ci - gi) + (cj - gj) * (cj - gj);
      q = q < cap ? q : cap;
      f = factor_by_q<kLatency>(rc, table, q);
    }
    unsigned long long near = __ballot(f != 1.0);
    while (near != 0ull) {
      const int j = __ffsll((long long)near) - 1;
      s = s * readlane_f64(f, j);
      near &= near - 1ull;
    }
  }
  return s;
}
// `list_addr`: the sorted candidate list of (year, variant); `class_addr`: the field of the radius class.
// returns cell | chunks requested << 16, or kSearchFallback (the winner's 0.03 * mean settlement opinion: the caller reads tab::m03)
// The scan reads the compact form of the sorted list, `pb_addr` / `pc_addr` = unpenalised score and cell per rank: three registers per
// chunk in flight instead of the eight of a full record (in the throughput kernel — four waves per SIMD — this function has 72
// registers, k_heavy_register_budget); the full records are only read for candidates that tie.
template <bool kLatency>
__device__ __noinline__ int place_heavy(unsigned long long list_addr, unsigned long long pb_addr, unsigned long long pc_addr, unsigned long long class_addr,
                                        unsigned long long tail_cells, double size_factor, int lane, int rc, int tbl, int ngen) {
#ifdef EG_STAMPS
  const unsigned long long ts0 = __builtin_readcyclecounter();
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the last field update's stores are in L2 before anything gathers
  const int ngen_s = __builtin_amdgcn_readfirstlane(ngen);
  constexpr int kChunks = (kCells + kWave - 1) / kWave;      // 41: the list holds exactly kChunks * 64 records
  constexpr int kGroup = 4, kGroups = (kChunks + kGroup - 1) / kGroup;
  constexpr double kKeep = 1.0 - 0x1p-30;
  // 1. largest approximate score M, scanning in descending order of the unpenalised score, four chunks per memory round
  //    trip (the records of the next group are requested while this group's field entries are on their way).  Branch-free per
  //    candidate: a lane keeps its largest approximate score, with the chunk it was seen in written into the value's six lowest
  //    bits (2^-46 relative: far inside the 2^-30 the candidates are kept by; one v_max then keeps value and place) and that
  //    candidate's cell, and its second largest as a value only.  Between rounds only the high words of the lanes' maxima are
  //    reduced — a lower bound of M, within 2^-20: the stop rule scans a little further at most —, the full maximum once at the end.
  //    Should a lane's second largest too end up within 2^-30 of M (ties en masse), pass 2 collects the candidates.
  double M = 0.0; int K = 0;
  double lm = 0.0, l2 = 0.0; int lcell = 0;
  // (the addresses are the same in every lane: as scalar registers they leave every request a 32-bit lane offset — the 72 registers of
  //  this function have no room for an address pair per request)
  auto uniform_u64 = [](unsigned long long a) { return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)a); };
  const unsigned long long pb_s = uniform_u64(pb_addr), pc_s = uniform_u64(pc_addr), a_s = uniform_u64(class_addr);
  auto pb_at = [&](int r) { return *(GlobalF64c)(pb_s + (unsigned long long)((unsigned)r * 8u)); };
  auto pc_at = [&](int r) { return *(GlobalI32c)(pc_s + (unsigned long long)((unsigned)r * 4u)); };
  auto field_at = [&](int cell) { return __hip_atomic_load((GlobalF64)(a_s + (unsigned long long)((unsigned)cell * 8u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  // Two register sets take turns as "this round's records" and "the next round's" (no copies between rounds); the compact lists
  // are padded with zeros far enough (kPcStride) for a round to be requested ahead without asking whether the list has ended.
  static_assert((kGroups + 1) * kGroup * kWave <= kPcStride, "the scan requests a round ahead");
  double cb[kGroup], nb[kGroup]; int cc[kGroup], nc[kGroup];
#pragma unroll
  for (int j = 0; j < kGroup; ++j) { cb[j] = pb_at(j * kWave + lane); cc[j] = pc_at(j * kWave + lane); }
#define EG_SCAN_ROUND(CB, CC, NB, NC, g_)                                                                                        \
  {                                                                                                                              \
    if ((g_) >= kGroups || ((g_) > 0 && !(readlane_f64(CB[0], 0) >= M * kKeep))) break;      /* sorted descending: lane 0 holds the round's bound */ \
    double ap[kGroup];                                                                                                           \
    _Pragma("unroll") for (int j = 0; j < kGroup; ++j) ap[j] = field_at(CC[j]);                                                  \
    _Pragma("unroll") for (int j = 0; j < kGroup; ++j) { NB[j] = pb_at((((g_) + 1) * kGroup + j) * kWave + lane); NC[j] = pc_at((((g_) + 1) * kGroup + j) * kWave + lane); } \
    _Pragma("unroll") for (int j = 0; j < kGroup; ++j) {                                                                         \
      const double v = CB[j] * ap[j];                                                                                            \
      const double key = __hiloint2double(__double2hiint(v), (__double2loint(v) & ~63) | ((g_) * kGroup + j));                   \
      const bool larger = key > lm;                                                                                              \
      l2 = vmax64(l2, vmin64(lm, key));      /* (scores: not negative, finite) */                                                \
      lm = vmax64(lm, key);                                                                                                      \
      lcell = larger ? CC[j] : lcell;                                                                                            \
    }                                                                                                                            \
    M = __hiloint2double((int)wave_max_u32((unsigned)__double2hiint(lm)), 0);      /* (scores are not negative: ordered like their bit patterns) */ \
    K = ((g_) + 1) * kGroup < kChunks ? ((g_) + 1) * kGroup : kChunks;                                                           \
  }
#pragma nounroll
  for (int g = 0;; g += 2) {
    EG_SCAN_ROUND(cb, cc, nb, nc, g)
    EG_SCAN_ROUND(nb, nc, cb, cc, g + 1)
  }
#undef EG_SCAN_ROUND
  M = wave_max_f64(lm);
  if (!(M >= 1e-250)) return kSearchFallback;      // (nothing placeable, or subnormal territory: the exact scan decides)
#ifdef EG_STAMPS
  const unsigned long long ts1 = __builtin_readcyclecounter();
#endif
  // 2. the candidates: everything within 2^-30 of M (as ranks in the sorted list, in sm.gstage[1])
  const double thr = M * kKeep;
  int ncand = 0, solo = -1;      // solo: the lane that holds the only candidate
  if (__ballot(l2 >= thr) == 0ull) {      // (thr > 0) no lane has seen two: the candidates are the lanes' own maxima
    const unsigned long long m1 = __ballot(lm >= thr);
    ncand = __popcll(m1);
    if (ncand == 1) solo = __ffsll((long long)m1) - 1;
    if ((m1 >> lane) & 1ull) sm.gstage[1][__popcll(m1 & ((1ull << lane) - 1ull))] = (__double2loint(lm) & 63) * kWave + lane;
  } else {
    for (int k = 0; k < K; ++k) {
      const double approx = pb_at(k * kWave + lane) * field_at(pc_at(k * kWave + lane));
      const unsigned long long m = __ballot(approx >= thr && approx > 0.0);
      if (m != 0ull) {
        const int pos = ncand + __popcll(m & ((1ull << lane) - 1ull));
        if (((m >> lane) & 1ull) && pos < kWave) sm.gstage[1][pos] = k * kWave + lane;
        ncand += __popcll(m);
      }
    }
  }
  if (ncand > kWave || ncand == 0) return kSearchFallback;
  wave_sync();
  // 3. exact scores of the candidates: the reference's product in list order, first maximum in cell order
#ifdef EG_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long ts2 = __builtin_readcyclecounter();
#endif
  ChunkBest b; b.score = 0.0; b.m03 = 0.0; b.cell = kCells;
  if (ncand == 1 && solo >= 0) {      // the usual case: ONE candidate.  It is the arg-max — the arg-max is among the candidates — and nothing
                                      // but its cell is asked for: its exact score (the reference's product over the whole list) need not be
                                      // formed at all.  (It is positive: within 2^-42 of an approximate score of at least 1e-250.)
    b.cell = __builtin_amdgcn_readlane(lcell, solo); b.score = 1.0;
  } else if (ncand <= 4 || ngen_s > kLdsGens) {      // one at a time, generator-parallel factors and the sequential product (exact_product_chain)
                                                     // (chunk_product below walks the on-chip window only)
    for (int k = 0; k < ncand; ++k) {
      const int rk = __builtin_amdgcn_readfirstlane(sm.gstage[1][k]);
      const PsRec e = load_rec(list_addr, rk);      // the same record in every lane
      const double sk = (exact_product_chain<kLatency>(rc, tbl, ngen_s, e.te, (int)e.cell, lane, tail_cells) * e.cf) * size_factor;
      if (sk > b.score || (sk == b.score && sk > 0.0 && (int)e.cell < b.cell)) { b.score = sk; b.cell = (int)e.cell; b.m03 = e.m03; }
    }
  } else {               // many ties: 64 candidates at once (chunk_product)
    const int r = lane < ncand ? sm.gstage[1][lane] : kCells;
    PsRec e; e.te = 0.0; e.cf = 1.0; e.m03 = 0.0; e.cell = 0u; e.pad = 0u;
    if (r < kCells) e = load_rec(list_addr, r);
    const int table = kLatency ? rc * (kD2Stride * 16) : tbl;
    const double s = chunk_score<kLatency>(table, size_factor, lane, ngen_s, r, e.te, e.cf, (int)e.cell, (int)e.pad);
    b = chunk_reduce<false>(s, (int)e.cell, e.m03);
  }
  if (!(b.score > 0.0)) return kSearchFallback;
#ifdef EG_STAMPS
  if (lane == 0) { sm.hdbg[0][0] += ts1 - ts0; sm.hdbg[0][1] += ts2 - ts1; sm.hdbg[0][2] += __builtin_readcyclecounter() - ts2; sm.hdbg[1][0] += (unsigned long long)K; sm.hdbg[1][1] += (unsigned long long)ncand; sm.hdbg[1][2] += 1ull; }
#endif
  wave_sync();
  if (lane == 0) sm.hres[1].m03 = b.m03;
  wave_sync();
  // requested, in units of 2 KB: K chunks of the compact list (64 x 12 B each), K x 64 field entries of 8 B, and — only when several
  // candidates tie — their full records (the field update that follows bills itself)
  return b.cell | (((3 * K + 7) / 8 + (K + 3) / 4 + (ncand > 1 ? 1 : 0)) << 16);
}

// The exact scan for a list that has outgrown the on-chip window (long-replay variant; reached when the field path cannot
// decide: no field slot, scores in subnormal territory, ties en masse).  The candidates, their order, the stop rule and the tie
// rule are place_search's; the generators stream past in blocks of 64 (the window from LDS, the rest from the episode's record)
// and every lane folds all of them, in list order, for its own candidate: a v_readlane per generator and chunk.  Slow, and rare.
// returns cell | chunks requested << 16, or -1 (no candidate with a positive score: actions.rs:77-89); the winner's 0.03 * mean
// settlement opinion in sm.hres[1].m03
template <bool kLatency>
__device__ __noinline__ int place_exact_long(unsigned long long list_addr, unsigned long long tail_cells, double size_factor, int lane, int rc, int tbl, int ngen) {
  const int ngen_s = __builtin_amdgcn_readfirstlane(ngen);
  const int cap = factor_cap<kLatency>(tbl);
  constexpr int kChunks = (kCells + kWave - 1) / kWave;
  double best = 0.0, m03w = 0.0; int best_c = kCells, chunks = 0;
  for (int chunk = 0; chunk < kChunks; ++chunk) {
    const int r = chunk * kWave + lane;
    const PsRec c = load_rec(list_addr, r);
    const double base = (c.te * c.cf) * size_factor;      // (te = 0 beyond the 2601 candidates)
    if (chunk > 0 && !(readlane_f64(base, 0) >= best)) break;      // sorted descending: lane 0 holds the chunk's bound
    ++chunks;
    const int ci = (int)c.cell / kGrid, cj = (int)c.cell - ci * kGrid;
    double s = c.te;
    for (int gb = 0; gb < ngen_s; gb += kWave) {
      const int mine = list_cell(tail_cells, gb, lane, ngen_s, 0);
      const int cnt = ngen_s - gb < kWave ? ngen_s - gb : kWave;
      for (int j = 0; j < cnt; ++j) {
        const int gc = __builtin_amdgcn_readlane(mine, j);
        const int gi = gc / kGrid, gj = gc - gi * kGrid;
        int q = (ci - gi) * (ci - gi) + (cj - gj) * (cj - gj);
        q = q < cap ? q : cap;      // (the table holds 1.0 there, and x * 1.0 == x)
        s = s * factor_by_q<kLatency>(rc, tbl, q);
      }
    }
    s = (s * c.cf) * size_factor;
    s = r < kCells ? s : 0.0;
    if (__any(s > best || (s == best && s > 0.0 && (int)c.cell < best_c))) {
      const ChunkBest b = chunk_reduce<false>(s, (int)c.cell, c.m03);
      if (b.score > best || (b.score == best && b.cell < best_c)) { best = b.score; best_c = b.cell; m03w = b.m03; }
    }
  }
  wave_sync();
  if (lane == 0) sm.hres[1].m03 = m03w;
  wave_sync();
  return best > 0.0 ? (best_c | (chunks << 16)) : -1;
}

// ---- weight nudges -----------------------------------------------------------------------------------------
// update_deficit_weights(action, d_improvement) followed by update_weights(action, w_improvement), as the repair loop calls
// them after every applied action (simulation.rs:453-486).  Every table entry receives at most one factor from each
// call (its own adjustment, or the "boost the others" factor, or the do-nothing boost), so both calls become one
// lane-parallel pass — lane l owns w[l] and dw[l] — with the same multiplications and clamps per entry.
__device__ __forceinline__ void nudge_after_repair(const DevSnapshot& S, int lane, int action, double d_improvement, double w_improvement) {
  const int slot = deficit_slot_of(action);
  const double lr = S.learning_rate;
  const double adj_d = d_improvement > 0.0 ? 1.0 + (lr * d_improvement * 1.5) : 1.0 / (1.0 + (lr * dabs(d_improvement) * 1.5));
  const double combined = S.immediate_weight * w_improvement + (1.0 - S.immediate_weight) * S.rel_improvement;
  const double adj_w = combined > 0.0 ? 1.0 + (lr * combined) : 1.0 / (1.0 + (lr * dabs(combined)));
  const double boost = S.boost_others;
  wave_sync();
  if (lane < EG_N_ACTIONS) {
    double v = SM_W[lane];
    if (lane == action) v = vmin64_u(vmax64_u(v * adj_w, kMinWeight), kMaxWeight);      // (weights: positive, finite)
    else if (combined < 0.0 && lane < kFirstOffset) v = vmin64_u(v * boost, kMaxWeight);
    if (combined < 0.0 && S.noop_boost && lane == kNothing) v = vmin64_u(v * S.boost_noop, kMaxWeight);
    SM_W[lane] = v;
  }
  if (slot >= 0 && lane < EG_N_DEFICIT) {
    double v = SM_DW[lane];
    if (lane == slot) v = vmin64_u(vmax64_u(v * adj_d, kMinWeight), kMaxWeight);
    else if (d_improvement < 0.0 && lane < 14) v = vmin64_u(v * boost, kMaxWeight);
    SM_DW[lane] = v;
  }
  wave_sync();
}

// ---- sampling (canonical table order; see include/eirgrid_hip.h) -------------------------------------------
__device__ int smart_fallback(Rng& r, int lane, int year) {   // sampling.rs:445-490
  const uint32_t offw = year < 2035 ? 5u : (year < 2045 ? 15u : 25u);
  const uint32_t w3 = year < 2035 ? 10u : 20u, w6 = year < 2035 ? 15u : (year < 2045 ? 10u : 5u);
  const uint32_t total = 15u + 10u + 15u + w3 + offw + offw + w6;
  uint32_t choice = rng_range32(r, lane, total);
  if (choice < 15u) return 0;              choice -= 15u;     // OnshoreWind
  if (choice < 10u) return 3;              choice -= 10u;     // OffshoreWind
  if (choice < 15u) return 12;             choice -= 15u;     // UtilitySolar
  if (choice < w3) return 3 * kBattery;    choice -= w3;
  if (choice < offw) return kFirstOffset;  choice -= offw;    // Forest
  if (choice < offw) return kFirstOffset + 6; choice -= offw; // ActiveCapture
  if (choice < w6) return 21;                                  // GasCombinedCycle
  return 3 * kBattery;
}
__device__ int smart_deficit_fallback(Rng& r, int lane) {   // sampling.rs:492-528: weights 30, 30, 20, 10, 0, 3
  uint32_t choice = rng_range32(r, lane, 93u);
  if (choice < 30u) return 3 * kPeaker;   choice -= 30u;
  if (choice < 30u) return 3 * kBattery;  choice -= 30u;
  if (choice < 20u) return 21;            choice -= 20u;
  if (choice < 10u) return 0;             choice -= 10u;
  if (choice < 3u) return 12;
  return 3 * kBattery;
}
__device__ int sample_action_weighted(const DevSnapshot& S, Rng& r, Totals& tot, int lane) {   // sampling.rs:147-237
  const bool explore = rng_f64(r, lane) < S.eps_main;
  if (explore) return (int)rng_range64(r, lane, (unsigned long long)EG_N_ACTIONS);
  if (S.stall > 500u) {   // sampling.rs:190-220: stable sort by weight descending, weights raised to power_scaling
    if (!tot.scaled_valid) {   // the row was nudged this year: rebuild the table (otherwise it is the host-built one)
      const double power = S.scaled_power;
      wave_sync();
      if (lane < EG_N_ACTIONS) {   // rank of this entry in the stable descending order; x^p by the shared eg_detpow
        const double mine = SM_W[lane];
        int rank = 0;
#pragma unroll 4
        for (int b = 0; b < EG_N_ACTIONS; ++b) { const double o = SM_W[b]; rank += (o > mine || (o == mine && b < lane)) ? 1 : 0; }
        sm.scaled[rank] = eg_detpow(mine, power);
        sm.ydef[128 + rank] = (uint8_t)lane;
      }
      wave_sync();
      tot.scaled_valid = true;
    }
    const int idx = weighted_pick((int)offsetof(Smem, scaled), EG_N_ACTIONS, rng_f64(r, lane), lane);
    return sm.ydef[128 + (idx < EG_N_ACTIONS ? idx : 0)];
  }
  // Every weight is >= MIN_WEIGHT > 0, so the running value only decreases: the entry at which it first reaches <= 0 is
  // the number of entries after which it was still positive.
  const int pick = weighted_pick((int)offsetof(Smem, pol), EG_N_ACTIONS, rng_f64(r, lane), lane);
  return pick < EG_N_ACTIONS ? pick : 3 * kPeaker;
}
__device__ int sample_deficit_weighted(const DevSnapshot& S, Rng& r, Totals& tot, int lane) {   // sampling.rs:315-377
  const bool explore = rng_f64(r, lane) < S.exploration_rate;
  if (explore) return 3 * c_deficit_type[(int)rng_range64(r, lane, 14ull)];
  const int pick = weighted_pick((int)(offsetof(Smem, pol) + 8 * snap::kPolDw), 14, rng_f64(r, lane), lane);
  return pick < 14 ? 3 * c_deficit_type[pick] : 3 * kPeaker;
}

struct Episode {   // wave-uniform bookkeeping of one episode
  int ngen, noff;
  int run_pos, def_pos, act_pos;      // flat log cursors
  int n_run_y, n_def_y, n_act_y;      // this year's counts
  int status;
  unsigned long long bytes;           // algorithmic bytes of SURVEY §8(d): whole numbers, kept as an integer (scalar registers)
  int chunks;                         // 64-candidate chunks of sorted candidate records (32 B each) the searches requested
  int heavy;                          // field slot of a heavy episode (place_heavy); -1: not asked for yet, -2: none to be had
  int heavy_classes;                  // bit rc: the field of radius class rc is built and kept up to date
  int heavy_quads;                    // list entries per lane / 4 of the field update (heavy_pack_list)
};

// ---- batch ("reduced") update statistics --------------------------------------------------------------------
// The reference applies apply_contrast_learning / apply_deficit_contrast_learning one episode at a time under a lock
// (multi_simulation.rs:494-508).  For a batch that shares one snapshot the same multiplicative nudges are
// accumulated in log space, as integers (Q32 fixed point), so the result does not depend on the order in which
// episodes, workgroups or ranks contribute: one sum all-reduce of this buffer is the whole exchange.
//   stats[0] episodes ok   [1] episodes failed   [2] episodes that qualify for contrast (learning.rs:160)
//   stats[3] best score of the batch as a sortable integer (bits of the score + 1; one-GPU best pick, see k_apply_update)
//   stats[8 + (y*61+a)]              Σ Q32 ln(penalty_factor)  over occurrences of a in year y that are absent from best
//   stats[8 + 26*61 + (y*61+a)]      Σ Q32 ln(mild_penalty)    over right-action-wrong-slot occurrences (learning.rs:241-251)
//   stats[8 + 2*26*61 + (y*15+s)]    number of deficit actions of slot s in year y absent from best_deficit_actions[y]
constexpr int kStatsMain = EG_YEARS * EG_N_ACTIONS;
constexpr double kQ32 = 4294967296.0;

// The policy's scalars as the kernels use them: always read from the snapshot buffer (snap::state), where the host
// upload or the last on-device update left them.
__device__ __forceinline__ void load_stats_params(const DevSnapshot& S, StatsParams& P) {
  const DevState& st = *S.state();
  P.best_score = st.p_best_score; P.has_best = st.has_lists; P.threshold = st.p_threshold; P.forced = st.p_forced;
  P.adaptive_lr = st.p_adaptive_lr; P.stagnation = st.p_stagnation;
}
// (They are the same in every lane but arrive through vector loads: said to be uniform, they live in scalar registers — as
//  per-lane values the four doubles alone were eight vector registers of the 128 the throughput kernels have, spilled to
//  scratch and reloaded around every nudge.)
__device__ __forceinline__ double uniform_f64(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ void load_state(DevSnapshot& S) {
  const DevState& st = *S.state();
  S.learning_rate = uniform_f64(st.learning_rate); S.exploration_rate = uniform_f64(st.exploration_rate);
  S.stall = (uint32_t)__builtin_amdgcn_readfirstlane((int)st.stall); S.has_best = __builtin_amdgcn_readfirstlane(st.has_best);
  S.has_cw = __builtin_amdgcn_readfirstlane(st.has_cw); S.noop_boost = __builtin_amdgcn_readfirstlane(st.noop_boost);
  S.rel_improvement = uniform_f64(st.rel_improvement); S.immediate_weight = uniform_f64(st.immediate_weight);
  S.has_best_actions = S.has_best_deficit = __builtin_amdgcn_readfirstlane(st.has_lists);
  S.heur_min = (uint32_t)__builtin_amdgcn_readfirstlane((int)st.heur_min); S.heur_max = (uint32_t)__builtin_amdgcn_readfirstlane((int)st.heur_max);
  S.boost_others = uniform_f64(st.boost_others); S.boost_noop = uniform_f64(st.boost_noop);
  S.eps_main = uniform_f64(st.eps_main); S.scaled_power = uniform_f64(st.scaled_power);
}

// One wave adds the contributions of episode e (its outputs must be visible in memory).
// `mult`: the episode stands for that many identical ones (the hoisted replay, eg_replay_coop.h): every sum receives mult times its
// contribution — integers, so that IS adding it mult times.
// `rep` >= 0: `stats` is the replicated form — kStatsReplicas copies, ENTRY-major (copy r of entry i at i * kStatsReplicas + r, so that
// k_fold_stats reads the copies of an entry as one 512-byte line) — and this episode adds to copy `rep`.
__device__ void episode_update_stats(const DevOut& O, const DevSnapshot& S, const StatsParams& P, uint32_t e, int lane, long long* stats,
                                     unsigned long long mult = 1ull, int rep = -1) {
  unsigned long long* const st0 = reinterpret_cast<unsigned long long*>(stats);
  const int stride = rep >= 0 ? kStatsReplicas : 1, ro = rep >= 0 ? rep : 0;
#define st(i_) st0[(size_t)(i_) * stride + ro]
  if (*O.status(e) != EG_EP_OK) {
    if (lane == 0) { atomicAdd(&st(1), mult); *O.score(e) = -1.0; O.score_list[e] = -1.0; }
    return;
  }
  const double score = rm::score(O.metrics(e));
  // slot 3: the batch's best score as an integer that sorts like the score (scores are not negative; + 1 so that 0 means
  // "no successful episode"): with one GPU k_apply_update finds the best episode from it without a kernel of its own
  if (lane == 0) { atomicAdd(&st(0), mult); *O.score(e) = score; O.score_list[e] = score; atomicMax(&st(3), (unsigned long long)__double_as_longlong(score) + 1ull); }
  if (!P.has_best) return;
  const double det = P.best_score > 0.0 ? (P.best_score - score) / P.best_score : 0.0;
  const bool qualifies = det > P.threshold || P.forced;                                               // learning.rs:160
  unsigned long long q_pen = 0, q_mild = 0;
  if (qualifies) {
    if (lane == 0) atomicAdd(&st(2), mult);
    if (det < 0.0) {      // forced contrast on an episode that beats the best: powf(negative, 0.3) is NaN in the reference and
                          // (w * NaN).max(MIN_WEIGHT) == MIN_WEIGHT — in log space any exponent below -9.2 (eg_reduced_math.h)
      q_pen = q_mild = (unsigned long long)(long long)(rm::kLnNanPenalty * kQ32);
    } else {              // det == 0: pow gives 0, both factors are 1 and only the boost remains
      const double combined = pow(det, 0.3) * P.stagnation;                                            // learning.rs:168-171
      // (the two logarithms side by side — lane 1 the mild one's argument, every other lane the penalty's: one evaluation of `log`,
      //  ~190 vector instructions, instead of two; the same operations on the same operands in the lanes that are read)
      const double arg_pen = P.adaptive_lr * 1.5 * combined;       // learning.rs:177
      const double arg_mild = P.adaptive_lr * combined * 0.5;      // learning.rs:247
      const long long q = llrint(log(1.0 / (1.0 + (lane == 1 ? arg_mild : arg_pen))) * kQ32);
      q_pen = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(q >> 32), 0) << 32) | (unsigned)__builtin_amdgcn_readlane((int)q, 0);
      q_mild = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(q >> 32), 1) << 32) | (unsigned)__builtin_amdgcn_readlane((int)q, 1);
    }
  }
  // The items are the entries of every year's `current = run ++ deficit` list (learning.rs:196-211), position j of year y against
  // position j of `best ++ best_deficit` of that year.  Flattened over the years — a lane an item — so that the episode's lists are
  // read in two memory round trips (the items and the best lists' offsets / masks of their years, then the best-list entries they
  // are compared with) instead of two per year: this runs at the end of every episode of a training batch, behind nothing to hide it.
  const uint8_t* run = O.run_log(e);
  const uint8_t* def = O.def_log(e);
  int nr = 0, nd = 0;      // lane y: the year's counts
  if (lane < EG_YEARS) { nr = O.n_run(e)[lane]; nd = O.n_def(e)[lane]; }
  // exclusive prefix sums over the years (lane y: where year y starts in run_log, in def_log); items of year y start at rp + dp
  int rp = nr, dp = nd;
#pragma unroll
  for (int sh = 1; sh < 32; sh <<= 1) {
    const int ur = __shfl_up(rp, sh), ud = __shfl_up(dp, sh);
    if (lane >= sh) { rp += ur; dp += ud; }
  }
  const int total = __builtin_amdgcn_readlane(rp, EG_YEARS - 1) + __builtin_amdgcn_readlane(dp, EG_YEARS - 1);
  rp -= nr; dp -= nd;
  const int ip = rp + dp;
  q_pen *= mult; q_mild *= mult;      // (two's-complement products: the sum of mult equal terms)
  for (int base = 0; base < total; base += kWave) {
    const int i = base + lane;
    int y = 0;      // the year of item i: the last year that starts at or before it (empty years share their start with the next one)
#pragma unroll
    for (int yy = 1; yy < EG_YEARS; ++yy) y = i >= __builtin_amdgcn_readlane(ip, yy) ? yy : y;
    // (every lane takes part in the shuffles: a lane masked off would not lend its year's values to the others)
    const int j = i - __shfl(ip, y), nr_y = __shfl(nr, y), rp_y = __shfl(rp, y), dp_y = __shfl(dp, y);
    const bool valid = i < total;
    int a = 0;
    unsigned long long mask = 0, dmask = 0; int b0 = 0, nb = 0, d0 = 0, nbd = 0;
    if (valid) {
      a = j < nr_y ? run[rp_y + j] : def[dp_y + (j - nr_y)];
      mask = S.best_mask()[y]; dmask = S.bestd_mask()[y];
      b0 = S.best_off()[y]; nb = S.best_off()[y + 1] - b0; d0 = S.bestd_off()[y]; nbd = S.bestd_off()[y + 1] - d0;
    }
    if (valid && qualifies) {
      if (!((mask >> a) & 1ull)) atomicAdd(&st(8 + y * EG_N_ACTIONS + a), q_pen);
      else if (j < nb + nbd) {
        const int b = j < nb ? S.best_actions()[b0 + j] : S.bestd_actions()[d0 + (j - nb)];
        if (a != b) atomicAdd(&st(8 + kStatsMain + y * EG_N_ACTIONS + a), q_mild);
      }
    }
    if (valid && j >= nr_y && !((dmask >> a) & 1ull)) {        // learning.rs:346-352
      const int slot = (a < kFirstOffset && a % 3 == 0) ? c_deficit_slot[a / 3] : (a == kNothing ? 14 : -1);
      if (slot >= 0) atomicAdd(&st(8 + 2 * kStatsMain + y * EG_N_DEFICIT + slot), mult);
    }
  }
#undef st
}

#ifdef EG_STAMPS
// every cycle of the episode is attributed: a mark charges the time since the previous mark to `slot`
#define EG_MARK(slot) do { const unsigned long long now_ = __builtin_readcyclecounter(); stamps[slot] += now_ - last_; last_ = now_; } while (0)
#define EG_MARKG(slot) EG_MARK(slot)
#define EG_T0() EG_MARK(6)
#define EG_T1(slot) EG_MARK(slot)
#define EG_TB(slot) EG_MARK(6)
#define EG_TE(slot) EG_MARK(slot)
#else
#define EG_T0() do {} while (0)
#define EG_T1(slot) do {} while (0)
#define EG_TB(slot) do {} while (0)
#define EG_MARKG(slot) do {} while (0)
#define EG_TE(slot) do {} while (0)
#endif

// Which episode of the batch a workgroup runs.  A batch is launched as up to three grids that run side by side on two
// streams: the episodes that replay the best strategy — the ones that grow long generator lists, SURVEY Q15 — on the
// replay variants of the kernel (kReplayLong: approximate-field placement from kHeavyGens generators on; kReplayShort: the
// exact scan, for short lists; see k_rollout), all others on the lean
// variant, whose code and registers are those of a kernel without the heavy path (measured: with the heavy calls compiled
// into the one kernel, sampled episodes ran 8 % / 13 % slower at 1 024 / 16 384 episodes).
struct EpisodeMap {
  const uint32_t* index;      // mode 1: episode = index[workgroup]
  uint32_t mode;              // 0: the workgroup index; 2: off + period * workgroup (the replays of a period); 3: the others
  uint32_t count, off, period;
  // replay hoist (eg_replay_coop.h): when *hoist carries this launch's sequence number, k_replay_coop has computed the batch's replay
  // episodes once and k_replay_broadcast hands every one of them the record: the replay variants have nothing to do (0: no hoist)
  const unsigned long long* hoist;
  unsigned long long hoist_seq;
  uint32_t stats_rep;         // 1: `stats` is kStatsReplicas copies of the statistics array (entry-major); this workgroup adds to copy (index % kStatsReplicas)
  // per-episode replay kernel (eg_replay_solo.h): word b carries solo_seq when k_replay_solo has completed workgroup b's episode — the
  // long-replay variant, launched behind it, then has nothing to do for that episode (0 / null: no such kernel in this launch)
  unsigned long long* solo;
  unsigned long long solo_seq;
};
__device__ __forceinline__ uint32_t map_episode(const EpisodeMap& m, uint32_t b) {
  if (m.mode == 0u) return b;
  if (m.mode == 1u) return m.index[b];
  if (m.mode == 2u) return m.off + m.period * b;
  if (b < m.off) return b;
  const uint32_t q = b - m.off, p1 = m.period - 1u;
  return m.off + (q / p1) * m.period + 1u + q % p1;
}

// (the heavy-capable variant: four waves per SIMD in the throughput kernel — 128 VGPRs, the field code held to 72 by
//  k_heavy_register_budget, the year's aggregates parked in LDS around its calls —, two in the small-batch kernel — 256 VGPRs)
// kKind: kLean = sampled episodes only (no replay code at all); kReplayShort / kReplayLong = the episodes that replay the best
// strategy, on the lean kernel's budget while the replayed list is short and on the heavy-capable variant once it is long.
// Both are launched over the replay episodes of a batch and the one whose turn it is not returns at once: how long the list
// is only the device knows (an on-device update may have replaced it since the host last looked), and the host plans
// launches many batches ahead of the device.
constexpr int kLean = 0, kReplayShort = 1, kReplayLong = 2;
// This file is compiled twice (csrc/Makefile).  eg_rollout.o holds the small-batch kernels (an episode wave and a helper wave) and
// everything that is not a rollout; eg_rollout_tp.o (-DEG_TU_THROUGHPUT) holds the three throughput kernels k_rollout<0, kind>,
// compiled WITHOUT machine-code loop-invariant code motion.  Out of the year and action loops that pass hoists literal constants and
// table addresses into registers — which are then live for the whole episode and across the calls of the field code: the long-replay
// variant takes 157 vector registers with it and 126 without, i.e. four waves per SIMD instead of two (with its aggregates parked in
// LDS around those calls and the field code held to 72 registers, k_heavy_register_budget), so that two long replays leave a SIMD room
// for two lean waves; the lean variant takes 120 instead of 128 + a spill.  Measured (profiles/r03_ab_notes.log): the sustained batch
// 3.07 -> 2.62 ms; 16 384 sampled episodes alone +1.3 %, the seeded batch +1.1 %; the small-batch kernel would lose 3.6 % and keeps
// the pass: hence two objects, and not a flag for the file.
#ifdef EG_TU_THROUGHPUT
#define EG_HEAVY_WAVES 4
#else
#define EG_HEAVY_WAVES 2
#endif
// actions in the best list up to which replay episodes stay on the exact scan (measured at 16 384 x 10 %: a batch with an
// 82-action list — 90 generators per replay episode — takes 2.02 ms this side of the limit and 2.37 ms on the other, one
// with a 109-action list — 117 generators — 2.39 against 2.25)
// (kShortReplayMax = 96, eg_internal.h)
template <int kHelpers, int kKind>
__global__ void __launch_bounds__(kWave * (1 + kHelpers), kKind == kReplayLong ? (kHelpers > 0 ? 2 : EG_HEAVY_WAVES) : (kHelpers > 0 ? 3 : 4)) k_rollout(DevTables T, DevSnapshot S_in, DevOut O, unsigned long long seed,
                                                                    unsigned long long first_index, uint32_t n_episodes,
                                                                    const uint8_t* __restrict__ replay_mask, uint32_t replay_period,
                                                                    long long* stats, EpisodeMap emap) {
  constexpr bool kHeavy = kKind == kReplayLong, kReplay = kKind != kLean;
  constexpr bool kPark = kHeavy && kHelpers == 0 && EG_HEAVY_WAVES >= 4;
  const int lane = threadIdx.x & (kWave - 1);
  if (blockIdx.x >= emap.count) return;
  if constexpr (kReplay) {      // (uniform for the whole grid)
    const bool long_list = S_in.state()->has_lists && S_in.best_off()[EG_YEARS] > kShortReplayMax;
    if (long_list != kHeavy) return;
    if (emap.hoist_seq != 0ull && *emap.hoist == emap.hoist_seq) return;      // served by k_replay_coop / k_replay_broadcast
    if constexpr (kHeavy) if (emap.solo_seq != 0ull && emap.solo[blockIdx.x] == emap.solo_seq) return;      // done by k_replay_solo
  }
  const uint32_t e = map_episode(emap, blockIdx.x);
  if (e >= n_episodes) return;
  // The tail of a large launch.  A grid of several rounds of waves ends when its last-dispatched episodes do, and those run their last
  // stretch on SIMDs that are emptying (a launch of 16 384 sampled episodes keeps 3.1 of its 4 wave slots per SIMD filled on average,
  // profiles/r04_lean_occupancy.txt).  The episodes dispatched last therefore win issue arbitration over the older waves on their
  // SIMD — which are about to end anyway —, in three steps of 640 workgroups: 16 384 episodes 1.028 -> 1.001 ms, 14 746: 0.942 -> 0.914
  // (profiles/r04_ab_notes.log r04x).  Not for grids that are resident all at once.
  if constexpr (kHelpers == 0) {
    constexpr uint32_t kTailStep = 640u;
    if (emap.count > 8192u) {
      const uint32_t behind = emap.count - 1u - blockIdx.x;      // workgroups dispatched after this one
      if (behind < kTailStep) __builtin_amdgcn_s_setprio(3);
      else if (behind < 2u * kTailStep) __builtin_amdgcn_s_setprio(2);
      else if (behind < 3u * kTailStep) __builtin_amdgcn_s_setprio(1);
    }
  }
  uint32_t search_seq = 0, year_seq = 0;      // commands to the helper wave share one sequence
  PrefixCache prefix_cache0 = {0.0, -1, 0};
  if constexpr (kHelpers > 0) {   // waves 1..kHelpers serve the episode wave's placement searches (see helper_loop)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // (the helper also brings the 8 KB factor table into LDS: it has nothing else to do until the first search, and the
    //  barrier of that search orders its LDS writes before anybody's reads)
    if (wave > 0) { load_latency_tables(T, lane); helper_loop(T, lane, wave); return; }
    __builtin_amdgcn_s_setprio(3);      // the episode wave is the critical path: it wins issue arbitration over helper waves
    // LDS arrives with whatever the previous workgroup on this CU left in it: a stale word that happens to equal a
    // sequence number would read as "result ready".  The flags start at 0 (sequence numbers start at 1); the helper
    // cannot touch them before the first command's barrier, which also orders these writes.
#ifndef EG_TEST_NO_FLAG_INIT
    if (lane == 0) { sm.hflag[0] = 0u; sm.hflag[1] = 0u; sm.yflag = 0u; }
#endif
  }
  DevSnapshot S = S_in;
  load_state(S);
  // iteration.rs:34-42: which episodes replay the best strategy comes from the caller's mask or, when the policy lives on
  // the device (the host cannot know whether a best strategy exists yet), from a period over the global episode index
  // (Replay episodes always run the heavy-capable variant — the launch plan sends them there, eg_api.cpp launch_batch —, so
  //  the lean variant carries no replay code at all.)
  const bool replay = kReplay && (replay_mask != nullptr ? replay_mask[e] != 0
                                                        : (replay_period != 0u && S.has_best_actions && (first_index + e) % replay_period == 0ull));
  const int n_existing = T.n_existing;
#ifdef EG_STAMPS
  unsigned long long stamps[32] = {};
  const unsigned long long t_begin = __builtin_readcyclecounter();
  unsigned long long last_ = t_begin;
#endif

#ifdef EG_STAMPS
  if (kHeavy && lane < 8) sm.hdbg[lane >> 2][lane & 3] = 0ull;
#endif
  load_static_tables(T, lane, kHelpers == 0 || kHeavy);      // (the heavy variant's field update reads sm.dr / sl.dr16)
  if constexpr (kHelpers == 0 && !kHeavy) sm.gstage[0][lane] = (int)0xC000C000;      // the kept staging row starts as padding (chunk_product)
  // bit y: the existing-plant prefix sums of year y equal those of year y-1, so last year's end-of-year class sums carry over
  const uint32_t carry_mask = (uint32_t)__ballot(lane > 0 && lane < EG_YEARS && T.pre_co2()[lane] == T.pre_co2()[lane - 1] &&
                                                 T.pre_tg()[lane] == T.pre_tg()[lane - 1] && T.pre_ig()[lane] == T.pre_ig()[lane - 1] &&
                                                 T.pre_sg()[lane] == T.pre_sg()[lane - 1]);
  Rng rng;
  rng_seed(rng, seed + first_index + (unsigned long long)e, lane);   // simulation.rs:50-53, one stream per episode
  EG_MARKG(16);

  Episode ep;
  ep.ngen = 0; ep.noff = 0; ep.run_pos = 0; ep.def_pos = 0; ep.act_pos = 0; ep.status = EG_EP_OK; ep.bytes = 32ull; ep.chunks = 0; ep.heavy = -1; ep.heavy_classes = 0; ep.heavy_quads = 0;
  uint8_t* run_log = O.run_log(e);
  uint8_t* def_log = O.def_log(e);
  uint8_t* act_log = O.act_log(e);
  uint16_t* gen_cell = O.gen_cell(e);
  uint16_t* gen_pack = O.gen_pack(e);
  uint16_t* off_pack = O.off_pack(e);
  // the long-replay variant's lists go on in the record beyond the on-chip window (ListTail); the others end there
  const ListTail tail = {(unsigned long long)gen_cell, (unsigned long long)gen_pack, (unsigned long long)off_pack};
  constexpr int kGenCap = kHeavy ? EG_MAX_GENS : kLdsGens, kOffCap = kHeavy ? EG_MAX_OFFSETS : kLdsGens;
  bool helper_sums = kHelpers > 0;      // (long-replay variant: until a list outgrows the window — the helper only reads LDS)

  // Wave-uniform doubles that are written once a year and read once a year live in LDS, not in (64-lane) registers:
  // sm.acc[0..2] accumulators of metrics_calculation.rs:133-153, [3..6] last yearly row; sm.yend[0..5] last year's
  // end-of-year capital sums (generators, offsets) and CO2 / output class sums; sm.ystate the state the year started in
  if (lane < 8) sm.acc[lane] = 0.0;
  if (lane < 6) sm.yend[lane] = 0.0;

  // The policy row block of a year (128 doubles, + the stalled sampler's tables) goes to LDS at the start of that year.  The
  // small-batch kernel requests it a year ahead — a lone latency-bound wave with registers to spare.  The throughput kernels
  // (kHelpers == 0: 128 registers, four waves per SIMD to hide a latency) request it at the start of its own year, ahead of
  // the year's gathers: held for a year, those seven registers were spilled to scratch and reloaded — the same latency plus
  // the stores.
  constexpr bool kRowAhead = kHelpers > 0;
  double np0 = 0.0, np1 = 0.0;
  if constexpr (kRowAhead) { np0 = S.pol()[lane]; np1 = S.pol()[64 + lane]; }
  const bool stalled = S.stall > 500u;                        // stalled sampler tables travel the same way
  double ns = 0.0; uint8_t nperm = 0;
  if constexpr (kRowAhead) if (stalled) { ns = S.scaled()[lane]; nperm = S.scaled_perm()[lane]; }
  for (int yi = 0; yi < kYears && ep.status == EG_EP_OK; ++yi) {
    const int year = 2025 + yi;
    EG_MARKG(17);
    // ---- requests first: the per-generator / per-offset terms of this year (first 64 of each list).  In the small-batch
    //      kernel the helper wave was asked for this year's sums when last year's actions ended (see below). ----
    const int ngen_s = __builtin_amdgcn_readfirstlane(ep.ngen);
    const int noff_s = __builtin_amdgcn_readfirstlane(ep.noff);
    const bool carry = ((carry_mask >> yi) & 1u) != 0u;
    const bool sums_from_helper = kHelpers > 0 && yi > 0 && (!kHeavy || helper_sums);
    if constexpr (!kRowAhead) {      // this year's policy block: requested first, it lands under the gathers below
      np0 = S.pol()[yi * snap::kPolRow + lane]; np1 = S.pol()[yi * snap::kPolRow + 64 + lane];
      if (stalled) { ns = S.scaled()[yi * 64 + lane]; nperm = S.scaled_perm()[yi * 64 + lane]; }
    }
    YearTerms terms;
    if (!sums_from_helper) terms = year_gather(T, lane, yi, ngen_s, noff_s);
    {  // this year's policy block -> LDS; small-batch kernel: then request next year's
    wave_sync();
    sm.pol[lane] = np0; sm.pol[64 + lane] = np1;
    if (stalled) { sm.scaled[lane] = ns; sm.ydef[128 + lane] = nperm; }
    if constexpr (kRowAhead) if (yi + 1 < kYears) {
      np0 = S.pol()[(yi + 1) * snap::kPolRow + lane]; np1 = S.pol()[(yi + 1) * snap::kPolRow + 64 + lane];
      if (stalled) { ns = S.scaled()[(yi + 1) * 64 + lane]; nperm = S.scaled_perm()[(yi + 1) * 64 + lane]; }
    }
    wave_sync();
    EG_T1(5);
    }
    ep.n_run_y = 0; ep.n_def_y = 0; ep.n_act_y = 0;
    Totals tot; tot.scaled_valid = true;
    EG_TE(14);

    // ---- aggregates at the start of the year (year_gather / year_fold above) ----
    Agg a;
    a.usage = sm.pol[snap::kPolYear + 5];
    a.gcost_prev = sm.yend[0]; a.ocost_prev = sm.yend[1];
    {
      EG_MARKG(18);
      YearSums ys;
      if (sums_from_helper) {      // folded by the helper wave while this wave closed last year
        // (on a timeout the year carries on with whatever the slots hold until the action loop's own status check ends the
        //  episode: a separate way out of the year loop from here cost the small-batch kernel 14 spilled VGPRs)
        for (int spins = 0; __builtin_amdgcn_readfirstlane((int)*(volatile uint32_t*)&sm.yflag) != (int)year_seq; ++spins) {
          __builtin_amdgcn_s_sleep(1);
          if (spins >= kSpinCap) { ep.status = EG_EP_INTERNAL; break; }
        }
        asm volatile("" ::: "memory");
        ys.gcost = sm.ysum[0]; ys.optot = sm.ysum[1]; ys.offs = sm.ysum[2]; ys.ocost = sm.ysum[3];
        ys.co2 = sm.ysum[4]; ys.tg = sm.ysum[5]; ys.ig = sm.ysum[6]; ys.sg = sm.ysum[7]; ys.opcnt = sm.ysum_opcnt;
      } else {
        ys = year_sums_init_current();
#if !defined(EG_PROBE_SKIP) || EG_PROBE_SKIP != 3
        year_fold<kHeavy>(T, lane, yi, ngen_s, noff_s, carry, terms, ys, tail);
#endif
      }
      a.gcost = ys.gcost; a.optot = ys.optot; a.offs = ys.offs; a.ocost = ys.ocost; a.opcnt = ys.opcnt;
      if (carry) { a.co2 = sm.yend[2]; a.tg = sm.yend[3]; a.ig = sm.yend[4]; a.sg = sm.yend[5]; }
      else { a.co2 = ys.co2; a.tg = ys.tg; a.ig = ys.ig; a.sg = ys.sg; }
      EG_T1(0);
    }
    ep.bytes += 2ull * (unsigned long long)(n_existing + ep.ngen) * 56ull + 2ull * (unsigned long long)ep.noff * 8ull + 184ull;

    // ---- the year's actions: phase 0 = deficit repair (simulation.rs:137-141, :319-522),
    //      phase 1 = additional actions (simulation.rs:144-198).  One loop so that apply_action is emitted once. ----
    EG_MARKG(18);
    int replay_idx = 0, replay_def_idx = 0;   // replay_index is keyed per year (sampling.rs:82, :246-247)
    // a replay episode reads this year's lists once: offsets as scalars, the first 128 / 64 entries one per lane (an action is
    // then a v_readlane instead of two dependent global round trips per action)
    int rep_lo = 0, rep_n = 0, repd_lo = 0, repd_n = 0, rep0 = 0, rep1 = 0, repd0 = 0;
    if constexpr (kReplay) if (replay) {
      rep_lo = __builtin_amdgcn_readfirstlane(S.best_off()[yi]); rep_n = __builtin_amdgcn_readfirstlane(S.best_off()[yi + 1]) - rep_lo;
      repd_lo = __builtin_amdgcn_readfirstlane(S.bestd_off()[yi]); repd_n = __builtin_amdgcn_readfirstlane(S.bestd_off()[yi + 1]) - repd_lo;
      if (!S.has_best_actions) rep_n = 0;
      if (!S.has_best_deficit) repd_n = 0;
      rep0 = lane < rep_n ? (int)S.best_actions()[rep_lo + lane] : 0;
      rep1 = kWave + lane < rep_n ? (int)S.best_actions()[rep_lo + kWave + lane] : 0;
      repd0 = lane < repd_n ? (int)S.bestd_actions()[repd_lo + lane] : 0;
    }
    // (the state at the start of the year is only read by the repair loop — most years have no deficit: its division waits for one)
    State year_start;
    year_start.balance = ((a.tg + a.ig) + a.sg) - a.usage;      // state_of(a).balance
    year_start.net = 0.0; year_start.opinion = 0.0; year_start.cost = 0.0;
    int phase = year_start.balance < 0.0 ? 0 : 1;
    if (phase == 0) {
      year_start = state_of(a);
      if (lane == 0) { sm.ystate[0] = year_start.net; sm.ystate[1] = year_start.opinion; sm.ystate[2] = year_start.balance; sm.ystate[3] = year_start.cost; }
    }
    double remaining = -year_start.balance;
    uint32_t attempts = 0, n_add = 0, k_add = 0;
    bool n_add_known = false;
    State cur = year_start;
    EG_TE(15);

    for (int guard = 0; guard < 200000 && ep.status == EG_EP_OK; ++guard) {
      int action;
      if (phase == 0) {
        if (!(remaining > 0.0)) {   // deficit closed: success bonus (simulation.rs:491-519), then go on to phase 1
          const State fin = cur;      // the state after the last repair action (the map has not changed since)
          wave_sync();
          State initial; initial.net = sm.ystate[0]; initial.opinion = sm.ystate[1]; initial.balance = sm.ystate[2]; initial.cost = sm.ystate[3];
          const double success = evaluate_impact(initial, fin);
          if (fin.balance >= 0.0 && success > 0.0 && ep.n_def_y > 0) {
            const double factor = 0.1 * success;
            wave_sync();
            // update_deficit_weights(action_i, factor) for every deficit action of the year, in order (simulation.rs:505-519):
            // factor > 0, so each call only multiplies its own slot and clamps — lane s counts the calls that hit slot s
            // and applies that many multiply-and-clamp steps to dw[s] (the entries do not interact)
            int hits = 0;
            const int mine0 = sm.ydef[lane], mine1 = sm.ydef[64 + lane];      // the year's deficit actions, one per lane
            for (int i = 0; i < ep.n_def_y; ++i) {
              const int da = i < kWave ? __builtin_amdgcn_readlane(mine0, i) : __builtin_amdgcn_readlane(mine1, i - kWave);
              hits += deficit_slot_of(da) == lane ? 1 : 0;
            }
            const double adj = 1.0 + (S.learning_rate * factor * 1.5);
            if (lane < EG_N_DEFICIT && hits > 0) {
              double v = SM_DW[lane];
              for (int k = 0; k < hits; ++k) v = dmin(dmax(v * adj, kMinWeight), kMaxWeight);
              SM_DW[lane] = v;
            }
            wave_sync();
          }
          phase = 1;
          continue;
        }
        attempts += 1;
        EG_MARKG(19);
        if (attempts < 5u) {
          if (replay) {   // sampling.rs:242-313
            if (replay_def_idx < repd_n) {
              action = replay_def_idx < kWave ? __builtin_amdgcn_readlane(repd0, replay_def_idx) : (int)S.bestd_actions()[repd_lo + replay_def_idx];
              replay_def_idx += 1;
            } else action = smart_deficit_fallback(rng, lane);
            if (ep.def_pos >= EG_DEF_CAP || ep.n_def_y >= 128) { ep.status = EG_EP_OVERFLOW; break; }
            if (lane == 0) { def_log[ep.def_pos] = (uint8_t)action; sm.ydef[ep.n_def_y] = (uint8_t)action; }
            ep.def_pos += 1; ep.n_def_y += 1;
          } else action = sample_deficit_weighted(S, rng, tot, lane);
        } else action = 3 * kBattery;   // simulation.rs:369-376
        EG_T1(2);
        if (action >= kFirstOffset) continue;   // only AddGenerator actions are applied in the repair loop (:398)
        // (`cur`, the state before this action, is the state after the previous one: nothing changes the map in between)
      } else {
        if (!n_add_known) {   // simulation.rs:144-187
          n_add_known = true;
          EG_MARKG(24);
          if (replay) {
            n_add = (uint32_t)rep_n;
          } else {   // sampling.rs:380-443
            const uint32_t dcount = (uint32_t)ep.n_def_y;
            const uint32_t cap = dcount >= 20u ? 0u : 20u - dcount;
            if (cap > 0u) {
              const double u = rng_f64(rng, lane);
              if (S.has_cw) {
                const double total = sm.pol[snap::kPolTotCount];   // the count table is never nudged (Q3): its sum is a snapshot constant
                if (total > 0.0) {      // the first count at which the walk reaches <= 0; none: 5 (sampling.rs:406-421)
                  const uint32_t c = (uint32_t)weighted_pick((int)(offsetof(Smem, pol) + 8 * snap::kPolCw), EG_N_COUNTS, u, lane);
                  n_add = c < (uint32_t)EG_N_COUNTS ? (c < cap ? c : cap) : (5u < cap ? 5u : cap);
                }
              } else {   // heuristic branch; min/max actions were evaluated on the host (sampling.rs:425-427)
                const uint32_t hi = S.heur_max < cap ? S.heur_max : cap;
                const uint32_t lo = S.heur_min < hi ? S.heur_min : hi;
                n_add = lo == hi ? lo : lo + rng_range32(rng, lane, hi - lo + 1u);
              }
            }
          }
          EG_T1(2);
        }
        if (k_add >= n_add) break;
        k_add += 1;
        EG_MARKG(19);
        if (replay) {   // sampling.rs:78-145
          if (replay_idx < rep_n) {
            action = replay_idx < kWave ? __builtin_amdgcn_readlane(rep0, replay_idx)
                                        : (replay_idx < 2 * kWave ? __builtin_amdgcn_readlane(rep1, replay_idx - kWave) : (int)S.best_actions()[rep_lo + replay_idx]);
            replay_idx += 1;
          } else action = smart_fallback(rng, lane, year);
          if (ep.run_pos >= EG_RUN_CAP) { ep.status = EG_EP_OVERFLOW; break; }
          if (lane == 0) run_log[ep.run_pos] = (uint8_t)action;
          ep.run_pos += 1; ep.n_run_y += 1;
        } else action = sample_action_weighted(S, rng, tot, lane);
        EG_T1(2);
      }

      // ---- apply_action (actions.rs:40-204), folded into the aggregates ----
      // (the action index is the same in every lane; saying so moves the index arithmetic below to the scalar unit)
      action = __builtin_amdgcn_readfirstlane(action);
      if (action < kFirstOffset) {
        const int t = action / 3, m = action - 3 * t;
        // terms that depend only on (year, type, multiplier) are requested inside the search, right behind its own
        // loads, and land while it runs
        double2 ccv; double cc_prev = 0.0, t12v;
        auto between = [&]() {
          ep.bytes += (unsigned long long)kCells * 8ull + (unsigned long long)(n_existing + ep.ngen) * 16ull;
          ccv = *reinterpret_cast<const double2*>(T.cc() + ((((unsigned)yi * kTypes + t) * kYears + yi) * kMults + m) * 2);
          if (yi > 0) cc_prev = T.cc()[((((unsigned)(yi - 1) * kTypes + t) * kYears + yi) * kMults + m) * 2];
          t12v = T.t12()[(unsigned)yi * kTypes + t];
        };
        EG_MARKG(20);
        double m03v = 0.0;
        int cell = -1;
        bool placed = false;
        // Long-replay variant at four waves per SIMD: the field code below is not inlined, and what is live across a call has to fit
        // beside the callee's registers.  The year's aggregates — thirty-odd registers of uniform doubles — wait in LDS from here until
        // the field update has been issued.
        if constexpr (kPark) {
          wave_sync();
          if (lane == 0) {
            sm.park[0] = a.co2; sm.park[1] = a.tg; sm.park[2] = a.ig; sm.park[3] = a.sg; sm.park[4] = a.optot; sm.park[5] = a.gcost; sm.park[6] = a.ocost;
            sm.park[7] = a.gcost_prev; sm.park[8] = a.ocost_prev; sm.park[9] = a.offs; sm.park[10] = a.usage;
            sm.park[11] = cur.net; sm.park[12] = cur.opinion; sm.park[13] = cur.balance; sm.park[14] = cur.cost; sm.park[15] = remaining;
            sm.park[16] = __hiloint2double(0, a.opcnt);
          }
          wave_sync();
        }
        if constexpr (kHeavy) if (ep.ngen >= kHeavyGens && ep.heavy != -2) {      // a long list: approximate field + exact evaluation of the few candidates
          const unsigned long long slot_bytes = (unsigned long long)(kRadiusClasses * kFieldStride) * 8ull;
          if (ep.heavy == -1) ep.heavy = heavy_claim(T, lane);
          if (ep.heavy >= 0) {
            const int info = __builtin_amdgcn_readfirstlane(sm.type_info[t]);
            const int hv = info & 15, hrc = (info >> 4) & 15;
            const unsigned long long class_addr = (unsigned long long)T.heavy + (unsigned long long)ep.heavy * slot_bytes + (unsigned long long)(hrc * kFieldStride) * 8ull;
            if (!((ep.heavy_classes >> hrc) & 1)) {      // the first search of this radius class: its field joins
              heavy_build_class<(kHelpers > 0)>(class_addr, tail.gen_cell, lane, hrc, throughput_table(info), (info >> 8) & 15, ep.ngen);
              ep.heavy_classes |= 1 << hrc;
              if constexpr (kHelpers > 0) ep.heavy_quads = heavy_pack_list<true>((unsigned long long)(T.base + tab::hv_box), (unsigned long long)(T.base + tab::dr_meta), lane, ep.heavy_classes);
              else {      // the host packed the list of every subset of classes: this episode's is copied (its first eight entries per lane)
                ep.heavy_quads = __builtin_amdgcn_readfirstlane(T.hv_quads()[ep.heavy_classes]);
                wave_sync();
#pragma unroll
                for (int k = 0; k < kBoxLds / kWave; ++k) sh2.box[k * kWave + lane] = T.hv_lists()[ep.heavy_classes * 1024 + k * kWave + lane];
                wave_sync();
              }
            }
#ifdef EG_STAMPS
            const unsigned long long th0 = __builtin_readcyclecounter();
#endif
            const size_t yv = (size_t)(yi * kMaxVariants + hv) * kPsStride, yc = (size_t)(yi * kMaxVariants + hv) * kPcStride;
            const int hr = place_heavy<(kHelpers > 0)>((unsigned long long)(T.ps() + yv), (unsigned long long)(T.pbase() + yc), (unsigned long long)(T.pcell() + yc), class_addr,
                                                        tail.gen_cell, T.size_factor, lane, hrc, throughput_table(info), ep.ngen);
#ifdef EG_STAMPS
            const unsigned long long th1 = __builtin_readcyclecounter();
            stamps[9] += th1 - th0;
#endif
            if (hr >= 0) { cell = hr & 0xFFFF; ep.chunks += hr >> 16; m03v = T.m03()[cell]; placed = true; between(); }
#ifdef EG_STAMPS
            stamps[10] += __builtin_readcyclecounter() - th1;
#endif
          }
        }
        if constexpr (kHeavy) if (!placed && ep.ngen > kLdsGens) {      // the exact scan, for a list beyond the window (place_search walks LDS)
          const int info = __builtin_amdgcn_readfirstlane(sm.type_info[t]);
          const int xr = place_exact_long<(kHelpers > 0)>((unsigned long long)(T.ps() + (size_t)(yi * kMaxVariants + (info & 15)) * kPsStride), tail.gen_cell,
                                                          T.size_factor, lane, (info >> 4) & 15, throughput_table(info), ep.ngen);
          cell = xr < 0 ? -1 : (xr & 0xFFFF);
          if (xr >= 0) { ep.chunks += xr >> 16; m03v = sm.hres[1].m03; }
          placed = true; between();
        }
        if (!placed) {
#ifdef EG_STAMPS
          cell = __builtin_amdgcn_readfirstlane(place_search<kHelpers, (kHelpers == 0 && !kHeavy)>(T, lane, yi, t, ep.ngen, nullptr, &m03v, prefix_cache0, between, ep.chunks, &search_seq, stamps));
          stamps[11] += 1;
#else
          cell = __builtin_amdgcn_readfirstlane(place_search<kHelpers, (kHelpers == 0 && !kHeavy)>(T, lane, yi, t, ep.ngen, nullptr, &m03v, prefix_cache0, between, ep.chunks, &search_seq));
#endif
        }
        EG_T1(1);
        // the field of every class the episode keeps, for the new generator — requested here, ahead of the bookkeeping, so that it is
        // one call region with the search (the aggregates are parked once); eight list entries per lane and call
        if constexpr (kHeavy) if (cell >= 0 && ep.ngen < kGenCap && ep.heavy_classes != 0) {
          const unsigned long long field_addr = (unsigned long long)T.heavy + (unsigned long long)ep.heavy * ((unsigned long long)(kRadiusClasses * kFieldStride) * 8ull);
          ep.chunks += 2 * ep.heavy_quads;      // 256 entries of 8 B read and written per quad = 2 units of 2 KB
          if constexpr (kHelpers > 0) {
            switch (ep.heavy_quads) {      // (uniform)
              case 1: heavy_add<true, 4>(field_addr, lane, cell, 0); break;
              case 2: heavy_add<true, 8>(field_addr, lane, cell, 0); break;
              case 3: heavy_add<true, 12>(field_addr, lane, cell, 0); break;
              default: heavy_add<true, 16>(field_addr, lane, cell, 0); break;
            }
          } else {
            const unsigned long long hv_list = (unsigned long long)(T.hv_lists() + ep.heavy_classes * 1024);
            if (ep.heavy_quads == 1) heavy_add_body<false, 4>(field_addr, hv_list, lane, cell, 0);
            else heavy_add_body<false, 8>(field_addr, hv_list, lane, cell, 0);
            if (ep.heavy_quads == 3) heavy_add_body<false, 4>(field_addr, hv_list, lane, cell, 8);
            else if (ep.heavy_quads >= 4) heavy_add_body<false, 8>(field_addr, hv_list, lane, cell, 8);
          }
        }
        if constexpr (kPark) {
          wave_sync();
          a.co2 = sm.park[0]; a.tg = sm.park[1]; a.ig = sm.park[2]; a.sg = sm.park[3]; a.optot = sm.park[4]; a.gcost = sm.park[5]; a.ocost = sm.park[6];
          a.gcost_prev = sm.park[7]; a.ocost_prev = sm.park[8]; a.offs = sm.park[9]; a.usage = sm.park[10];
          cur.net = sm.park[11]; cur.opinion = sm.park[12]; cur.balance = sm.park[13]; cur.cost = sm.park[14]; remaining = sm.park[15];
          a.opcnt = __double2loint(sm.park[16]);
        }
        if (cell < 0) { ep.status = cell == kSearchLost ? EG_EP_INTERNAL : EG_EP_NO_LOCATION; break; }   // actions.rs:77-89 is unreachable here (Q16)
        EG_MARKG(21);
        if (ep.ngen >= kGenCap) { ep.status = EG_EP_OVERFLOW; break; }
        if (lane == 0) {
          if (!kHeavy || ep.ngen < kLdsGens) {
            sm.gcell[ep.ngen] = (uint16_t)(cell | (t << 12));
            sm.gbm[ep.ngen] = (uint8_t)(yi | (m << 5));
          }
          if constexpr (kHelpers == 0 && !kHeavy) if (ep.ngen < kWave) {      // the searches' staging row, kept (chunk_product)
            const int gi = cell / kGrid;
            sm.gstage[0][ep.ngen] = gi | ((cell - gi * kGrid) << 16);
          }
          gen_cell[ep.ngen] = (uint16_t)cell;
          gen_pack[ep.ngen] = (uint16_t)(t | (yi << 4) | (m << 9));
        }
        wave_sync();
        ep.ngen += 1;
        a.gcost += ccv.x;
        if (yi > 0) a.gcost_prev += cc_prev;
        a.co2 += sm.type_co2[t];
        const double out = sm.type_out[t];
        const int cls = (sm.type_info[t] >> 12) & 3;
        if (cls == 1) a.ig += out; else if (cls == 2) a.sg += out; else a.tg += out;
        a.optot += (m03v + t12v) + ccv.y;
        a.opcnt += 1;
        if constexpr (kHelpers > 0) {      // the searches of both waves read the list from here (chunk_product_latency)
          // (the helper may still be evaluating its chunk of the search that just ended: it masks what lies behind the
          //  list it was given, chunk_product_latency<true>, so the new entry may appear under it)
          if (lane == 0 && (!kHeavy || ep.ngen <= kLdsGens)) {
            const int gi = cell / kGrid;
            sl.gpk[ep.ngen - 1] = (4 * gi) | ((4 * (cell - gi * kGrid)) << 16);
          }
        }
        EG_TE(12);
      } else if (action < kFirstOther) {
        EG_MARKG(20);
        const int ot = (action - kFirstOffset) / 3, m = (action - kFirstOffset) - 3 * ot;
        if (ep.noff >= kOffCap) { ep.status = EG_EP_OVERFLOW; break; }
        const uint16_t p = (uint16_t)(ot | (yi << 4) | (m << 9));
        if (lane == 0) { if (!kHeavy || ep.noff < kLdsGens) sm.opack[ep.noff] = p; off_pack[ep.noff] = p; }
        wave_sync();
        ep.noff += 1;
        a.offs += T.offv()[((unsigned)yi * kOffsetTypes + ot) * kYears + yi];
        a.ocost += T.offc()[((unsigned)yi * kOffsetTypes + ot) * kMults + m];
        if (yi > 0) a.ocost_prev += T.offc()[((unsigned)(yi - 1) * kOffsetTypes + ot) * kMults + m];
        EG_TE(13);
      }
      // 57..59 carry an empty generator id (core.rs:117-119): the lookup fails, nothing changes.  60: DoNothing.

      if (phase == 0) {   // simulation.rs:406-486
        if (ep.def_pos >= EG_DEF_CAP || ep.n_def_y >= 128 || ep.run_pos >= EG_RUN_CAP) { ep.status = EG_EP_OVERFLOW; break; }
        if (lane == 0) { def_log[ep.def_pos] = (uint8_t)action; sm.ydef[ep.n_def_y] = (uint8_t)action; run_log[ep.run_pos] = (uint8_t)action; }
        ep.def_pos += 1; ep.n_def_y += 1; ep.run_pos += 1; ep.n_run_y += 1;
        EG_MARKG(22);
        const State nxt = state_of(a);
#if defined(EG_PROBE_SKIP) && EG_PROBE_SKIP == 2      // diagnostic builds only (profiles/r04_ab_notes.log r04zb): what a piece costs in vector instructions
        remaining = -dmin(nxt.balance, 0.0); cur = nxt;
        if (true) continue;
#endif
        // evaluate_action_impact (scoring.rs:46-85) and the emission / cost terms of simulation.rs:420-452 divide the same
        // differences by the same denominators: each quotient is formed once
        const bool net_positive = cur.net > 0.0;
        double q_net = 0.0, q_cost = 0.0;
        if (net_positive || nxt.net < cur.net) q_net = (cur.net - nxt.net) / max_abs_one(cur.net);
        if (!net_positive || nxt.net < 1000.0) { const double cost_change = nxt.cost - cur.cost; q_cost = -cost_change / max_abs_one(cur.cost); }
        double overall;
        if (net_positive) overall = q_net;
        else {
          const double opinion_improvement = (nxt.opinion - cur.opinion) / max_abs_one(cur.opinion);
          const double cost_weight = cur.cost > kMaxCost * 8.0 ? 0.8 : 0.5;
          overall = q_cost * cost_weight + opinion_improvement * (1.0 - cost_weight);
        }
        const double em = nxt.net < cur.net ? q_net : 0.0;
        const double ci = nxt.net < 1000.0 ? q_cost : 0.0;
        const double oi = nxt.cost < kMaxCost * 8.0 ? (nxt.opinion - cur.opinion) / dmax(1.0 - cur.opinion, 0.1) : 0.0;
        const double combined = overall * 0.7 + em * 0.15 + ci * 0.1 + oi * 0.05;
#if !defined(EG_PROBE_SKIP) || EG_PROBE_SKIP != 1
        nudge_after_repair(S, lane, action, combined, overall * 0.5);
#endif
        tot.scaled_valid = false;
        remaining = -dmin(nxt.balance, 0.0);
        cur = nxt;
        EG_T1(3);
      } else {            // simulation.rs:193-197
        if (ep.act_pos >= EG_ACT_CAP || ep.run_pos >= EG_RUN_CAP) { ep.status = EG_EP_OVERFLOW; break; }
        if (lane == 0) { act_log[ep.act_pos] = (uint8_t)action; run_log[ep.run_pos] = (uint8_t)action; }
        ep.act_pos += 1; ep.n_act_y += 1; ep.run_pos += 1; ep.n_run_y += 1;
        EG_MARKG(23);
      }
    }
    if (ep.status != EG_EP_OK) break;
    ep.bytes += 2ull * (unsigned long long)(ep.n_act_y + ep.n_def_y);
    if constexpr (kHelpers > 0 && kHeavy) if (ep.ngen > kLdsGens || ep.noff > kLdsGens) helper_sums = false;
    if constexpr (kHelpers > 0) {      // the lists are final for this year: the helper folds next year's starting sums meanwhile
      if (yi + 1 < kYears && (!kHeavy || helper_sums)) {
        search_seq += 1; year_seq = search_seq;
        if (lane == 0) {
          sm.cmd[search_seq & 1][0] = kCmdYear | (yi + 1) | ((int)((carry_mask >> (yi + 1)) & 1u) << 8) | ((int)((carry_mask >> (yi + 2)) & 1u) << 9);
          sm.cmd[search_seq & 1][1] = ep.ngen | (ep.noff << 16);
        }
        wg_barrier_lds();
      }
    }

    // ---- yearly metrics (metrics_calculation.rs:32-175) ----
    EG_MARKG(25);
#if defined(EG_PROBE_SKIP) && EG_PROBE_SKIP == 4
    if (true) continue;
#endif
    const State s = state_of(a);
    const double gen = (a.tg + a.ig) + a.sg;
    const double credit = s.net >= 0.0 ? 0.0 : (-s.net) * sm.pol[snap::kPolYear + 8];
    const double total_capital = a.gcost + a.ocost;
    const double yearly_capital = yi == 0 ? total_capital : total_capital - (a.gcost_prev + a.ocost_prev);
    double sales = 0.0;
    if (S.enable_energy_sales && s.balance > 0.0) { const double gwh = s.balance * 8.76; sales = gwh * 50000.0; }
    const double yearly_total = yearly_capital + 0.0 + 0.0 - credit - (S.enable_energy_sales ? sales : 0.0);
    wave_sync();
    const double total_cost = yi == 0 ? yearly_total : sm.acc[0] + yearly_total;
    const double total_credit = yi == 0 ? credit : sm.acc[1] + credit;
    const double total_sales = yi == 0 ? sales : sm.acc[2] + sales;
    wave_sync();
    if (lane == 0) {
      sm.acc[0] = total_cost; sm.acc[1] = total_credit; sm.acc[2] = total_sales;
      sm.acc[3] = s.net; sm.acc[4] = s.opinion; sm.acc[5] = total_capital; sm.acc[6] = s.balance;
    }
    if (S.write_yearly && lane == 0) {   // one lane: 21 adjacent 8-byte stores (merged pairwise)
      double* row = O.yearly(e) + yi * EG_YEARLY_FIELDS;
      row[EG_Y_YEAR] = (double)year; row[EG_Y_POP] = sm.pol[snap::kPolYear + 6]; row[EG_Y_USAGE] = a.usage; row[EG_Y_GEN] = gen;
      row[EG_Y_BALANCE] = s.balance; row[EG_Y_OPINION] = s.opinion; row[EG_Y_YEARLY_CAPITAL] = yearly_capital;
      row[EG_Y_TOTAL_CAPITAL] = total_capital; row[EG_Y_INFLATION] = sm.pol[snap::kPolYear + 7]; row[EG_Y_CO2] = a.co2;
      row[EG_Y_OFFSET] = a.offs; row[EG_Y_NET_CO2] = s.net; row[EG_Y_YEARLY_CREDIT] = credit; row[EG_Y_TOTAL_CREDIT] = total_credit;
      row[EG_Y_YEARLY_SALES] = sales; row[EG_Y_TOTAL_SALES] = total_sales; row[EG_Y_ACTIVE_GENS] = (double)a.opcnt;
      row[EG_Y_UPGRADE_COSTS] = 0.0; row[EG_Y_CLOSURE_COSTS] = 0.0;   // identically 0 (simulation.rs:41-42)
      row[EG_Y_YEARLY_TOTAL_COST] = yearly_total; row[EG_Y_TOTAL_COST] = total_cost;
    }
    if (lane == 0) {
      O.n_run(e)[yi] = ep.n_run_y;
      O.n_def(e)[yi] = ep.n_def_y;
      O.n_act(e)[yi] = ep.n_act_y;
    }
    if (lane == 0) { sm.yend[0] = a.gcost; sm.yend[1] = a.ocost; sm.yend[2] = a.co2; sm.yend[3] = a.tg; sm.yend[4] = a.ig; sm.yend[5] = a.sg; }
    EG_T1(4);
  }

  wave_sync();
  if (lane == 0) {   // SimulationMetrics, iteration.rs:69-74 (Q2: total_cost is the last year's capital cost)
    O.metrics(e)[0] = sm.acc[3];
    O.metrics(e)[1] = sm.acc[4];
    O.metrics(e)[2] = sm.acc[5];
    O.metrics(e)[3] = sm.acc[6] >= 0.0 ? 1.0 : 0.0;
    *O.status(e) = ep.status;
    *O.n_gens(e) = ep.ngen;
    *O.n_offsets(e) = ep.noff;
    *O.n_draws(e) = (unsigned long long)rng.words;
    *O.bytes_moved(e) = (double)ep.bytes;
    *O.n_chunks(e) = (uint32_t)ep.chunks;
#ifdef EG_STAMPS
    EG_MARKG(26);
    stamps[7] = __builtin_readcyclecounter() - t_begin;
    if constexpr (kHeavy) {      // heavy searches: scan / candidates + records / exact evaluation / field update cycles; chunks, candidates, searches
      stamps[27] = sm.hdbg[0][0]; stamps[28] = sm.hdbg[0][1]; stamps[29] = sm.hdbg[0][2]; stamps[30] = sm.hdbg[0][3];
      stamps[24] = sm.hdbg[1][0]; stamps[25] = sm.hdbg[1][1]; stamps[26] = sm.hdbg[1][2];
    }
    // where did this workgroup run?  HW_ID (se / sh / cu / simd / wave slot) and XCC_ID: more workgroups on one CU than fit
    // at once means some of them had to wait for a slot (second round) — scripts/bench_tail.py
    stamps[31] = (unsigned long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                 ((unsigned long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    // diagnostic build only: cycle shares go to the (otherwise unread) tail of this episode's act_log buffer
    unsigned long long* dbg = (unsigned long long*)(act_log + EG_ACT_CAP - 256);
    for (int i = 0; i < 32; ++i) dbg[i] = stamps[i];
#endif
  }
  if constexpr (kHelpers > 0) {   // release the helper waves
    search_seq += 1;
    if (lane == 0) {
      sm.cmd[search_seq & 1][0] = -1;
      // after a protocol fault the helper's sequence number cannot be trusted: it finds the exit command in either buffer
      // (it has read the command it is working on into registers, so overwriting that buffer is harmless)
      if (ep.status == EG_EP_INTERNAL) sm.cmd[(search_seq & 1) ^ 1][0] = -1;
    }
    wg_barrier_lds();
  }
  if (stats != nullptr) {   // fused batch-update statistics: this episode's lists are re-read by all lanes
    // The wave re-reads what it stored itself: its stores only have to have left the wave (workgroup-scope release =
    // s_waitcnt vmcnt(0)); they went through to L2 and nothing of these buffers was ever loaded into this CU's L1.  A
    // device-scope fence here would write back and invalidate the XCD's L2 once per episode.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    wave_sync();
    StatsParams P;      // only needed here: not held in registers through the episode
    load_stats_params(S, P);
    episode_update_stats(O, S, P, e, lane, stats, 1ull, emap.stats_rep ? (int)(blockIdx.x % (uint32_t)kStatsReplicas) : -1);
  }
}

#ifdef EG_TU_THROUGHPUT
// Never launched.  A device function's register budget is the tightest launch bound among the kernels that can reach it, and the
// attributes that would say so directly are for kernels only: this kernel's bound of seven waves per SIMD gives the long-replay
// variant's field code 72 registers, which — with the episode's aggregates parked in LDS around those calls — is what lets that
// variant run four waves per SIMD (128 registers) beside the lean grid.
__global__ void __launch_bounds__(kWave, 7) k_heavy_register_budget(unsigned long long a, unsigned long long b, unsigned long long c, double d, int i, int* out) {
  const int lane = threadIdx.x;
  int r = place_heavy<false>(a, b, c, a + b, b + c, d, lane, i, i, i);
  r += place_exact_long<false>(a, c, d, lane, i, i, i);
  heavy_build_class<false>(a, c, lane, i, i, i, i);
  out[lane] = r;
}
#endif

#include "eg_replay_script.h"      // a replay episode's script and yearly rows (both objects)
#ifdef EG_TU_THROUGHPUT
#include "eg_replay_solo.h"        // k_replay_solo: a long replay episode on its own wave, script / placements / rows one after the other
#endif

#ifndef EG_TU_THROUGHPUT      // (everything from here to the launchers lives in eg_rollout.o only)
#include "eg_replay_coop.h"      // k_replay_coop, k_replay_broadcast: the replay episodes of a batch, computed once

// ---- B2: a single placement search, for parity tests of the arg-max --------------------------------------------
__global__ void __launch_bounds__(kWave) k_place(DevTables T, int type, int yi, const uint16_t* __restrict__ cells,
                                                 int n_extra, int32_t* out_cell, double* out_score) {
  const int lane = threadIdx.x;
  for (int g = lane; g < n_extra; g += kWave) sm.gcell[g] = cells[g];
  load_static_tables(T, lane, true);
  __syncthreads();
  double score = 0.0;
  PrefixCache pc0 = {0.0, -1, 0};
  int nchunks = 0;
  const int cell = place_search<0>(T, lane, yi, type, n_extra, &score, nullptr, pc0, []() {}, nchunks);
  if (lane == 0) { *out_cell = cell; *out_score = score; }
}

// ---- B2 with the reference's own signature: generators at arbitrary coordinates, a size penalty --------------------
// MetalLocationSearch::find_suitable_location (metal_location_search.rs:96-176) for the ctx's settlements (populations of
// year `yi`) and existing plant — that part of every candidate's product is the host table te, in the reference's order —
// times, for each of the caller's generators in list order, distance / radius when closer than the radius (sqrt and division
// are IEEE correctly rounded here as in the reference), times the coast factor, times 1 - size_penalty * 0.1.  All 2601
// distinct candidates are evaluated (no generator sits on the 1 km grid, so nothing of the rollout's tables applies);
// first maximum in (i, j) order = highest score, ties to the lowest cell.
__global__ void __launch_bounds__(kWave) k_place_xy(DevTables T, int type, int yi, const double* __restrict__ gx, const double* __restrict__ gy,
                                                    int n, double radius, double size_term, int32_t* out_cell, double* out_score) {
  const int lane = threadIdx.x;
  const int rc = T.rclass()[type], v = T.variant()[type];
  const PsRec* __restrict__ list = T.ps() + (size_t)(yi * kMaxVariants + v) * kPsStride;
  (void)rc;
  double best = 0.0; int best_c = kCells;
  for (int r = lane; r < kCells; r += kWave) {
    const PsRec c = list[r];
    const int ci = (int)c.cell / kGrid, cj = (int)c.cell - ci * kGrid;
    const double x = (double)ci * 1000.0, y = (double)cj * 1000.0;
    double score = c.te;
    for (int g = 0; g < n; ++g) {
      const double dx = x - gx[g], dy = y - gy[g];
      const double distance = __builtin_sqrt(dx * dx + dy * dy);
      if (distance < radius) score = score * (distance / radius);
    }
    score = score * c.cf;            // 1.0 for the types without the coast term (x * 1.0 == x)
    score = score * size_term;
    if (score > best || (score == best && score > 0.0 && (int)c.cell < best_c)) { best = score; best_c = (int)c.cell; }
  }
  const ChunkBest b = chunk_reduce<false>(best, best_c, 0.0);
  if (lane == 0) { *out_cell = b.score > 0.0 ? b.cell : -1; *out_score = b.score; }
}

// Stalled sampler tables of a freshly uploaded snapshot (sampling.rs:190-220 on the un-nudged rows): per year the
// weights in stable descending order raised to the power (shared eg_detpow), the permutation and the table-order sum.
// One workgroup per year; runs on the stream right behind the snapshot copy.
// (one wave per year; `s_w`, `s_scaled`: 64 doubles of LDS each, the wave's own)
__device__ __forceinline__ void stalled_tables_year(uint8_t* snap_base, int y, int lane, uint32_t stall, double* s_w, double* s_scaled) {
  const double stagnation = rm::dmind((double)stall / 1000.0, 3.0);
  const double power = 1.0 + (2.0 * stagnation);
  double* row = reinterpret_cast<double*>(snap_base + snap::pol) + y * snap::kPolRow;
  const double* w = row;
  double* scaled = reinterpret_cast<double*>(snap_base + snap::scaled) + y * 64;
  uint8_t* perm = snap_base + snap::scaled_perm + y * 64;
  const double mine = lane < EG_N_ACTIONS ? w[lane] : 0.0;
  wave_sync();
  s_w[lane] = mine;
  wave_sync();
  if (lane < EG_N_ACTIONS) {
    int rank = 0;
    for (int b = 0; b < EG_N_ACTIONS; ++b) { const double o = s_w[b]; rank += (o > mine || (o == mine && b < lane)) ? 1 : 0; }
    const double v = eg_detpow(mine, power);
    s_scaled[rank] = v; scaled[rank] = v; perm[rank] = (uint8_t)lane;
  }
  // (the table-order sum of the powered weights is not kept: weighted_pick forms it from the table it is handed — a lane adding 61
  //  LDS words one after the other was 4 of k_apply_update's microseconds in every stalled update; slot kPolScaledTotal stays 0)
}
__global__ void __launch_bounds__(kWave) k_stalled_tables(uint8_t* snap_base) {
  __shared__ double s_w[64], s_scaled[64];
  const uint32_t stall = reinterpret_cast<const DevState*>(snap_base + snap::state)->stall;
  if (stall <= 500u) return;
  stalled_tables_year(snap_base, blockIdx.x, threadIdx.x, stall, s_w, s_scaled);
}

// statistics of a finished batch without re-running it (same accumulation as the k_rollout epilogue)
__global__ void __launch_bounds__(kWave) k_update_stats(DevOut O, DevSnapshot S_in, uint32_t n, long long* stats) {
  const uint32_t e = blockIdx.x;
  if (e >= n) return;
  DevSnapshot S = S_in; StatsParams P;
  load_state(S); load_stats_params(S, P);
  episode_update_stats(O, S, P, e, threadIdx.x, stats);
}

// best episode of the batch: highest score, ties to the lowest index; its metrics and action lists are copied behind
// the statistics so that one transfer carries everything the host-side update needs
__global__ void __launch_bounds__(1024) k_pick_best(DevOut O, uint32_t n, unsigned long long first_index, UpdateCandidate* cand) {
  __shared__ double s_score[1024];
  __shared__ int s_idx[1024];
  const int tid = threadIdx.x;
  double best = -1.0; int best_i = -1;
  for (uint32_t i = tid; i < n; i += 1024) {
    const double sc = O.score_list[i];
    if (sc > best) { best = sc; best_i = (int)i; }      // ascending i per thread: first maximum
  }
  s_score[tid] = best; s_idx[tid] = best_i;
  __syncthreads();
  for (int w = 512; w >= 1; w >>= 1) {
    if (tid < w) {
      const double o = s_score[tid + w]; const int oi = s_idx[tid + w];
      if (oi >= 0 && (o > s_score[tid] || (o == s_score[tid] && (s_idx[tid] < 0 || oi < s_idx[tid])))) { s_score[tid] = o; s_idx[tid] = oi; }
    }
    __syncthreads();
  }
  const int win = s_idx[0];
  if (tid == 0) { cand->score = win >= 0 ? s_score[0] : -1.0; cand->index = win >= 0 ? (long long)(first_index + (unsigned long long)win) : -1ll; }
  if (win < 0) return;
  if (tid < 4) cand->metrics[tid] = O.metrics(win)[tid];
  if (tid < EG_YEARS) { cand->n_run[tid] = O.n_run(win)[tid]; cand->n_def[tid] = O.n_def(win)[tid]; }
  for (int i = tid; i < EG_RUN_CAP; i += 1024) cand->run_log[i] = O.run_log(win)[i];
  for (int i = tid; i < EG_DEF_CAP; i += 1024) cand->def_log[i] = O.def_log(win)[i];
}

// ---- the reference's `best_result` (core/multi_simulation.rs:384, :613-620): which run is summarised and exported ------
// After its parallel section the reference folds the results in iteration order:
//     best_result = None;  for result in results { if best_result.map_or(true, |best|
//         evaluate_action_impact(result.metrics -> best.metrics, optimization_mode) > 0.0) { best_result = Some(result) } }
// with the arguments as written — `result` is the "current state" and `best` the "new state", so a result takes over when
// the held one is an IMPROVEMENT on it.  That is the run multi_simulation.rs:821-905 prints and exports, and it is not the
// policy's best strategy (score_metrics, strategy.rs:19-258).  The fold is sequentially dependent but a result rarely takes
// over (the held run drifts towards the worst: a running extreme), so it is evaluated speculatively: 1024 results at a time
// against the held run, the FIRST that takes over (lowest index) becomes the held run and the rest of the window is
// examined again — the sequential fold's answer, in n / 1024 + (take-overs) rounds.  Failed episodes are skipped (in the
// reference a failed iteration ends the run, multi_simulation.rs:611).  The winner's whole record is kept behind the state.
__device__ __forceinline__ double fold_impact(const double* cur, const double* nxt, int cost_only) {      // scoring.rs:46-85
  State c, x;
  c.net = cur[0]; c.opinion = cur[1]; c.balance = 0.0; c.cost = cur[2];      // metrics_to_action_result, multi_simulation.rs:55-62
  x.net = nxt[0]; x.opinion = nxt[1]; x.balance = 0.0; x.cost = nxt[2];
  if (cost_only) { const double cost_change = x.cost - c.cost; return -cost_change / dmax(dabs(c.cost), 1.0); }      // scoring.rs:52-58
  return evaluate_impact(c, x);
}
__global__ void __launch_bounds__(1024) k_fold_best(DevOut O, uint32_t n, unsigned long long first_index, int cost_only, uint8_t* fold) {
  constexpr int kPer = 8;      // results per thread and tile, in registers
  __shared__ double s_best[4];
  __shared__ int s_has, s_first, s_win;
  const int tid = threadIdx.x;
  FoldState* st = reinterpret_cast<FoldState*>(fold);
  if (tid < 4) s_best[tid] = st->metrics[tid];
  if (tid == 0) { s_has = st->has; s_win = -1; s_first = 0x7FFFFFFF; }
  __syncthreads();
  for (uint32_t tile = 0; tile < n; tile += kPer * 1024u) {
    double m[kPer][4]; bool ok[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const uint32_t i = tile + (uint32_t)k * 1024u + (uint32_t)tid;
      ok[k] = i < n && *O.status(i) == EG_EP_OK;
#pragma unroll
      for (int j = 0; j < 4; ++j) m[k][j] = ok[k] ? O.metrics(i)[j] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < kPer; ++k) {      // window k: results tile + 1024 k + [0, 1024), in index order = thread order
      int from = 0;
      for (;;) {
        double b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = s_best[j];
        const bool take = tid >= from && ok[k] && (s_has == 0 || fold_impact(m[k], b, cost_only) > 0.0);
        if (take) atomicMin(&s_first, tid);
        __syncthreads();
        const int f = s_first;
        __syncthreads();      // everybody has read it
        if (f == 0x7FFFFFFF) break;
        if (tid == f) {
#pragma unroll
          for (int j = 0; j < 4; ++j) s_best[j] = m[k][j];
          s_has = 1; s_win = (int)(tile + (uint32_t)k * 1024u) + f; s_first = 0x7FFFFFFF;
        }
        from = f + 1;
        __syncthreads();
      }
    }
  }
  __syncthreads();
  const int win = s_win;
  if (win < 0) return;      // the held run stays
  const unsigned long long* src = reinterpret_cast<const unsigned long long*>(O.base + (size_t)win * rec::stride);
  unsigned long long* dst = reinterpret_cast<unsigned long long*>(fold + kFoldRecord);
  for (int i = tid; i < (int)(rec::stride / 8); i += 1024) dst[i] = src[i];
  if (tid == 0) {
    for (int j = 0; j < 4; ++j) st->metrics[j] = s_best[j];
    st->index = (long long)(first_index + (unsigned long long)win); st->has = 1;
  }
}

// ---- on-device batch update (eg_policy_apply_reduced, eg_policy.cpp, restated for one 1024-thread workgroup) -----------
// The policy lives in the snapshot buffer; this kernel consumes the (all-reduced) statistics and the candidate records
// of a batch and leaves the buffer as eg_upload_snapshot would have written it after the host update: weights, row
// sums, best lists / offsets / masks, scalars (derive_state) — the same bits, because every formula comes from
// eg_reduced_math.h.  The stalled sampler tables follow in k_stalled_tables.  The statistics are zeroed for the next batch.
__device__ void chacha12_block(const uint32_t* key, unsigned long long counter, uint32_t* out) {
  uint32_t s[16], x[16];
  s[0] = 0x61707865u; s[1] = 0x3320646eu; s[2] = 0x79622d32u; s[3] = 0x6b206574u;
#pragma unroll
  for (int i = 0; i < 8; ++i) s[4 + i] = key[i];
  s[12] = (uint32_t)counter; s[13] = (uint32_t)(counter >> 32); s[14] = 0u; s[15] = 0u;
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = s[i];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    EG_QR(x[0], x[4], x[8], x[12]) EG_QR(x[1], x[5], x[9], x[13]) EG_QR(x[2], x[6], x[10], x[14]) EG_QR(x[3], x[7], x[11], x[15])
    EG_QR(x[0], x[5], x[10], x[15]) EG_QR(x[1], x[6], x[11], x[12]) EG_QR(x[2], x[7], x[8], x[13]) EG_QR(x[3], x[4], x[9], x[14])
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) out[i] = x[i] + s[i];
}

// `packets` = n_packets update packets of EG_PACKET_BYTES (one per rank, in rank order; the gathered copies when N > 1):
// the statistics are summed here (integers: any order gives the same sum), the candidate records sit behind them.
__global__ void __launch_bounds__(1024) k_apply_update(uint8_t* snap_base, const uint8_t* packets, int n_cands, long long* zero_stats,
                                                      unsigned long long noise_seed, const uint8_t* out_base, const double* score_list, uint32_t n_local,
                                                      unsigned long long first_index, int local_pick, uint32_t* list_len_out) {
  constexpr int NA = EG_N_ACTIONS, ND = EG_N_DEFICIT, Y = EG_YEARS;
  constexpr int kMainDraws = Y * NA, kDefDraws = Y * ND, kBlocks = (2 * (kMainDraws + kDefDraws) + 15) / 16;
  __shared__ uint32_t s_noise[kBlocks * 16];
  __shared__ uint32_t s_key[8];
  __shared__ DevState st;
  __shared__ int s_winner, s_improved, s_randomized_main, s_owned;
  __shared__ __align__(16) uint8_t s_best[snap::kBestCap]; __shared__ __align__(16) uint8_t s_bestd[snap::kBestCap];      // the best lists as they were before this update
  __shared__ int s_off[2][Y + 1];
  __shared__ double s_ln_boost;
  __shared__ rm::DeficitContrast s_dc;
  __shared__ int s_prefix[2][Y + 1];
  const int tid = threadIdx.x;
#ifdef EG_STAMPS
  __shared__ unsigned long long dbg_t[16];      // diagnostic: phase boundaries (100 MHz counter), see scripts/apply_phases.py
  if (tid == 0) dbg_t[0] = wall_clock64();
#endif
  double* pol = reinterpret_cast<double*>(snap_base + snap::pol);
  int32_t* best_off = reinterpret_cast<int32_t*>(snap_base + snap::best_off);
  int32_t* bestd_off = reinterpret_cast<int32_t*>(snap_base + snap::bestd_off);
  uint8_t* best_actions = snap_base + snap::best_actions;
  uint8_t* bestd_actions = snap_base + snap::bestd_actions;
  DevState* gstate = reinterpret_cast<DevState*>(snap_base + snap::state);

  // serial pieces run on different waves side by side: state + winner + "is it an improvement" on wave 1, the noise key on
  // wave 0
  const uint8_t* cands = packets + 8 * EG_STATS_LEN;      // record r at cands + r * EG_PACKET_BYTES
  auto stat = [&](int i) {
    long long v = 0;
    for (int r = 0; r < n_cands; ++r) v += reinterpret_cast<const long long*>(packets + (size_t)r * EG_PACKET_BYTES)[i];
    return v;
  };
  // Everything the contrast steps read is requested now and lands while the serial pieces run: the old best lists go to
  // LDS (the steps walk them per table entry), each thread's table entries and statistics into its registers.
  constexpr int kPen = 8, kMild = 8 + Y * NA, kDcnt = 8 + 2 * Y * NA;
  {
    const uint32_t* b4 = reinterpret_cast<const uint32_t*>(best_actions); const uint32_t* d4 = reinterpret_cast<const uint32_t*>(bestd_actions);
    reinterpret_cast<uint32_t*>(s_best)[tid] = b4[tid]; reinterpret_cast<uint32_t*>(s_bestd)[tid] = d4[tid];      // kBestCap = 4 x 1024 bytes
    if (tid <= Y) { s_off[0][tid] = best_off[tid]; s_off[1][tid] = bestd_off[tid]; }
  }
  double w_in[2] = {0.0, 0.0}, pen_in[2] = {0.0, 0.0}, dw_in = 0.0; long long dcnt_in = 0;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int i = tid + 1024 * k;
    if (i < Y * NA) { const int y = i / NA, a = i - y * NA; w_in[k] = pol[y * snap::kPolRow + a]; pen_in[k] = (double)stat(kPen + i) + (double)stat(kMild + i); }
  }
  if (tid < Y * ND) { const int y = tid / ND, sl = tid - y * ND; dw_in = pol[y * snap::kPolRow + snap::kPolDw + sl]; dcnt_in = stat(kDcnt + tid); }
  if (local_pick) {
    // One GPU: the candidate record of the packet is made here instead of by k_pick_best (a launch of 5.5 us per step).
    // The rollout epilogue left the batch's best score in statistics slot 3; the best episode is the lowest index that
    // holds it (highest score, ties to the lowest index), its metrics and lists are copied behind the statistics.
    UpdateCandidate* cw = reinterpret_cast<UpdateCandidate*>(const_cast<uint8_t*>(cands));
    const DevOut O{const_cast<uint8_t*>(out_base), const_cast<double*>(score_list)};
    const unsigned long long key = reinterpret_cast<const unsigned long long*>(packets)[3];
    if (tid == 0) s_winner = 0x7FFFFFFF;
    __syncthreads();
    if (key != 0ull)      // (from the back-to-back copy of the scores: one per 12 KB record cost 25 us at 16 384 episodes)
      for (uint32_t base = 0; base < n_local; base += 16u * 1024u) {
        double sc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) { const uint32_t i = base + (uint32_t)k * 1024u + (uint32_t)tid; sc[k] = i < n_local ? O.score_list[i] : -1.0; }
#pragma unroll
        for (int k = 0; k < 16; ++k)
          if (sc[k] >= 0.0 && (unsigned long long)__double_as_longlong(sc[k]) + 1ull == key) atomicMin(&s_winner, (int)(base + (uint32_t)k * 1024u + (uint32_t)tid));
      }
    __syncthreads();
    const int win = s_winner != 0x7FFFFFFF ? s_winner : -1;
    if (tid == 0) { cw->score = win >= 0 ? *O.score(win) : -1.0; cw->index = win >= 0 ? (long long)(first_index + (unsigned long long)win) : -1ll; }
    if (win >= 0) {
      if (tid < 4) cw->metrics[tid] = O.metrics(win)[tid];
      if (tid < EG_YEARS) { cw->n_run[tid] = O.n_run(win)[tid]; cw->n_def[tid] = O.n_def(win)[tid]; }
      // (four bytes a thread: the lists start at multiples of four in the record and in the packet)
      static_assert(rec::run_log % 4 == 0 && rec::def_log % 4 == 0 && rec::stride % 4 == 0 && offsetof(UpdateCandidate, run_log) % 4 == 0 &&
                    offsetof(UpdateCandidate, def_log) % 4 == 0 && EG_RUN_CAP % 4 == 0 && EG_DEF_CAP % 4 == 0, "word copies of the winner's lists");
      for (int i = tid; i < EG_RUN_CAP / 4; i += 1024) reinterpret_cast<uint32_t*>(cw->run_log)[i] = reinterpret_cast<const uint32_t*>(O.run_log(win))[i];
      for (int i = tid; i < EG_DEF_CAP / 4; i += 1024) reinterpret_cast<uint32_t*>(cw->def_log)[i] = reinterpret_cast<const uint32_t*>(O.def_log(win))[i];
    }
    __syncthreads();      // the record is in place for every thread of this workgroup
  }
  if (tid == 64) {
    st = *gstate;
    // the batch's candidate: highest score, ties to the lowest global index (eg_policy_apply_packet)
    int win = -1; double ws = 0.0; long long wi = 0;
    for (int r = 0; r < n_cands; ++r) {
      const UpdateCandidate* c = reinterpret_cast<const UpdateCandidate*>(cands + (size_t)r * EG_PACKET_BYTES);
      if (c->index < 0) continue;
      if (win < 0 || c->score > ws || (c->score == ws && c->index < wi)) { win = r; ws = c->score; wi = c->index; }
    }
    s_winner = win;
    bool improved = false;
    if (win >= 0 && stat(0) > 0) {
      const UpdateCandidate* c = reinterpret_cast<const UpdateCandidate*>(cands + (size_t)win * EG_PACKET_BYTES);
      improved = !st.has_best || rm::score(c->metrics) > rm::score(st.best_metrics);
    }
    s_improved = improved ? 1 : 0;
  }
  // the two transcendental-heavy scalars of the contrast steps depend on the old state and the episode count only: each is
  // evaluated once, on a wave of its own, beside the pieces above (every thread used to evaluate them for itself)
  if (tid == 128) s_ln_boost = rm::contrast_ln_boost(gstate->learning_rate, gstate->stall);
  if (tid == 192) s_dc = rm::deficit_contrast(gstate->learning_rate, gstate->stall + (uint32_t)stat(0));   // the stall counter after a batch without improvement
  if (tid == 0) {
    unsigned long long state = noise_seed;      // rand_core seed_from_u64 (HostRng)
    for (int i = 0; i < 8; ++i) {
      state = state * 6364136223846793005ull + 11634580027462260723ull;
      const uint32_t x = (uint32_t)(((state >> 18) ^ state) >> 27), rot = (uint32_t)(state >> 59);
      s_key[i] = (x >> rot) | (x << ((32u - rot) & 31u));
    }
  }
  __syncthreads();
#ifdef EG_STAMPS
  if (tid == 0) dbg_t[1] = wall_clock64();
#endif
  const long long n_ok = stat(0), n_qual = stat(2);
  const bool contrast = st.has_best && st.has_lists && n_qual > 0;
  const bool randomize_main = contrast && st.stall > 1200u;
  // the noise stream is only needed beyond 1200 stalled episodes; the deficit table may need it even if the main one
  // did not (its stall counter is the updated one), so the blocks are made whenever either could
  const bool maybe_noise = st.stall + (uint32_t)n_ok > 1200u;
  if (maybe_noise && tid < kBlocks) chacha12_block(s_key, (unsigned long long)tid, s_noise + 16 * tid);
  __syncthreads();
#ifdef EG_STAMPS
  if (tid == 0) dbg_t[2] = wall_clock64();
#endif
  auto draw = [&](int j) {      // j-th gen::<f64>() of the stream
    const unsigned long long v = ((unsigned long long)s_noise[2 * j + 1] << 32) | s_noise[2 * j];
    return (double)(v >> 11) * (1.0 / 9007199254740992.0);
  };

  // ---- apply_contrast_learning over the batch ----
  if (contrast) {
    const double ln_boost = s_ln_boost;
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      const int i = tid + 1024 * k2;
      if (i >= Y * NA) break;
      const int y = i / NA, a = i - y * NA;
      int occ = 0;
      for (int k = s_off[0][y]; k < s_off[0][y + 1]; ++k) occ += s_best[k] == a ? 1 : 0;
      for (int k = s_off[1][y]; k < s_off[1][y + 1]; ++k) occ += s_bestd[k] == a ? 1 : 0;
      double w = rm::nudge(w_in[k2], (double)n_qual * (double)occ * ln_boost, pen_in[k2] / 4294967296.0);
      if (randomize_main) w = rm::noise(w, draw(i));
      pol[y * snap::kPolRow + a] = w;
    }
  }
  __syncthreads();
#ifdef EG_STAMPS
  if (tid == 0) dbg_t[3] = wall_clock64();
#endif

  // ---- update_best_strategy with the batch's candidate ----
  if (tid == 0) {
    const bool improved = s_improved != 0;
    st.iteration_count += (uint32_t)n_ok;
    st.failed_total += (uint32_t)stat(1);
    s_randomized_main = randomize_main ? 1 : 0;
    if (improved) {
      const UpdateCandidate* c = reinterpret_cast<const UpdateCandidate*>(cands + (size_t)s_winner * EG_PACKET_BYTES);
      DevImprovement* log = reinterpret_cast<DevImprovement*>(snap_base + snap::imp_log) + (st.n_improvements % snap::kImpLogCap);
      log->score = rm::score(c->metrics); log->iteration = st.iteration_count; log->pad = 0;
      for (int k = 0; k < 4; ++k) { log->metrics[k] = c->metrics[k]; st.best_metrics[k] = c->metrics[k]; }
      st.n_improvements += 1;
      st.has_best = 1; st.has_lists = 1; st.stall = 0;
      int a = 0, b = 0;
      for (int y = 0; y < Y; ++y) { s_prefix[0][y] = a; s_prefix[1][y] = b; a += c->n_run[y]; b += c->n_def[y]; }
      s_prefix[0][Y] = a; s_prefix[1][Y] = b;
      const long long local = c->index - (long long)first_index;
      s_owned = (out_base && local >= 0 && local < (long long)n_local) ? 1 : 0;
      *reinterpret_cast<uint32_t*>(snap_base + snap::best_rec_state) = s_owned ? 1u : 2u;
    } else st.stall += (uint32_t)n_ok;
    st.improved_last = improved ? 1 : 0;
  }
  __syncthreads();
#ifdef EG_STAMPS
  if (tid == 0) dbg_t[4] = wall_clock64();
#endif
  const bool improved = s_improved != 0;
  if (improved) {      // the candidate's lists become the best lists; the main weights of this moment are kept beside them
    const UpdateCandidate* c = reinterpret_cast<const UpdateCandidate*>(cands + (size_t)s_winner * EG_PACKET_BYTES);
    const int nr = s_prefix[0][Y], nd = s_prefix[1][Y];
    for (int i = tid; i < nr && i < (int)snap::kBestCap; i += 1024) best_actions[i] = c->run_log[i];
    for (int i = tid; i < nd && i < (int)snap::kBestCap; i += 1024) bestd_actions[i] = c->def_log[i];
    if (tid <= Y) { best_off[tid] = s_prefix[0][tid]; bestd_off[tid] = s_prefix[1][tid]; }
    if (tid < Y) {
      unsigned long long m = 0, dm = 0;
      for (int k = s_prefix[0][tid]; k < s_prefix[0][tid + 1]; ++k) if (c->run_log[k] < 64) m |= 1ull << c->run_log[k];
      for (int k = s_prefix[1][tid]; k < s_prefix[1][tid + 1]; ++k) if (c->def_log[k] < 64) { m |= 1ull << c->def_log[k]; dm |= 1ull << c->def_log[k]; }
      reinterpret_cast<unsigned long long*>(snap_base + snap::best_mask)[tid] = m;
      reinterpret_cast<unsigned long long*>(snap_base + snap::bestd_mask)[tid] = dm;
    }
    double* best_w = reinterpret_cast<double*>(snap_base + snap::best_w);
    for (int i = tid; i < Y * NA; i += 1024) { const int y = i / NA, a = i - y * NA; best_w[i] = pol[y * snap::kPolRow + a]; }
    // the whole record of the winning episode (yearly rows, action list, placements) stays with the policy when the
    // episode ran here (eg_fetch_best_run; the run the reference exports is another one, see k_fold_best)
    if (s_owned) {
      const unsigned long long* src = reinterpret_cast<const unsigned long long*>(out_base + (size_t)(c->index - (long long)first_index) * rec::stride);
      unsigned long long* dst = reinterpret_cast<unsigned long long*>(snap_base + snap::best_rec);
      for (int i = tid; i < (int)(rec::stride / 8); i += 1024) dst[i] = src[i];
    }
  }

  // ---- apply_deficit_contrast_learning: the same factor for every episode (it depends on the stall counter only) ----
  if (!improved && st.has_best && st.has_lists) {
    const rm::DeficitContrast dc = s_dc;      // st.stall is the old counter + n_ok here (no improvement)
    if (dc.active) {
      const bool randomize = st.stall > 1200u;
      const int first_draw = s_randomized_main ? kMainDraws : 0;
      for (int i = tid; i < Y * ND; i += 1024) {      // (Y * ND < 1024: one entry per thread, the one requested at the start)
        const int y = i / ND, sl = i - y * ND;
        int occ = 0;
        for (int k = s_off[1][y]; k < s_off[1][y + 1]; ++k) {
          const int a = s_bestd[k];
          const int slot = (a < kFirstOffset && a % 3 == 0) ? c_deficit_slot[a / 3] : (a == kNothing ? 14 : -1);
          occ += slot == sl ? 1 : 0;
        }
        double w = rm::nudge(dw_in, (double)n_ok * (double)occ * dc.ln_boost, (double)dcnt_in * dc.ln_pen);
        if (randomize) w = rm::noise(w, draw(first_draw + i));
        pol[y * snap::kPolRow + snap::kPolDw + sl] = w;
      }
    }
  }
  __syncthreads();
#ifdef EG_STAMPS
  if (tid == 0) dbg_t[5] = wall_clock64();
#endif

  // ---- what eg_upload_snapshot derives: row sums in table order, scalars ----
  // (the rows come to LDS with one load per thread and entry; a thread a year then adds its 61 + 14 entries in table order from
  //  there — summed straight from global memory, 75 dependent-latency loads per thread, this was 10 of the kernel's 22 us)
  __shared__ double s_scratch[2 * 16 * 64];      // [26][76] here; the stalled sampler's per-wave tables further down
  {
    constexpr int kPer = NA + 14 + 1;      // 76: a row's main weights, its first 14 deficit weights, padding
    for (int i = tid; i < Y * (NA + 14); i += 1024) {
      const int y = i / (NA + 14), k = i - y * (NA + 14);
      s_scratch[y * kPer + k] = pol[y * snap::kPolRow + (k < NA ? k : snap::kPolDw + (k - NA))];
    }
    __syncthreads();
    if (tid < Y) {
      const double* r = s_scratch + tid * kPer;
      double a = 0.0, b = 0.0;
      for (int i = 0; i < NA; ++i) a += r[i];
      for (int i = 0; i < 14; ++i) b += r[NA + i];
      double* row = pol + tid * snap::kPolRow;
      row[snap::kPolTotMain] = a; row[snap::kPolTotDeficit] = b;      // the count row is never nudged: its sum stays
    }
  }
  // (derive_state in its four parts, a wave each, beside the row sums of wave 0; the state goes out behind the next barrier)
  if (tid == 64) rm::derive_state_score(st);
  if (tid == 128) rm::derive_state_heur(st);
  if (tid == 192) rm::derive_state_contrast(st);
  if (tid == 256) rm::derive_state_rest(st);
  // the length of the best list goes to a pinned host word: the host plans its launches by it (which replay variant is the long
  // pole, whether a field pool is needed) without ever waiting for the device
  // (only when the list has changed — an improvement —: a store to host memory is a PCIe write the kernel's end waits for)
  if (tid == 65 && list_len_out && s_improved != 0) {
    const uint32_t len = st.has_lists ? (uint32_t)s_prefix[0][Y] : 0u;
    __hip_atomic_store(list_len_out, len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // this rank's statistics buffer is ready for the next batch's epilogue (when it is also `packets`, every read of it
  // happened before the barriers above)
  __syncthreads();
  if (tid == 64) *gstate = st;
#ifdef EG_STAMPS
  if (tid == 0) dbg_t[6] = wall_clock64();
#endif
  for (int i = tid; i < EG_STATS_LEN; i += 1024) zero_stats[i] = 0;
  // the stalled sampler's tables of the rows as they are now (k_stalled_tables's work: a launch of its own — 5 us and a dependency gap
  // behind every update, for something that only happens beyond 500 iterations without improvement — until round 3): a wave a year
  if (st.stall > 500u) {
    static_assert(Y * (NA + 15) <= 2 * 16 * 64, "one LDS buffer serves the row sums and the stalled tables");
    double* s_sw = s_scratch + (tid >> 6) * 64; double* s_ss = s_scratch + 16 * 64 + (tid >> 6) * 64;      // (the barrier above is behind the row sums)
    for (int y = tid >> 6; y < Y; y += 16) stalled_tables_year(snap_base, y, tid & 63, st.stall, s_sw, s_ss);
  }
#ifdef EG_STAMPS
  __syncthreads();      // the unused statistics slots 4..7 carry the phase durations out (after the zeroing above)
  if (tid == 0) {
    const unsigned long long t_end = wall_clock64();
    zero_stats[4] = (long long)(dbg_t[2] - dbg_t[0]);      // (best pick,) state, winner, noise key, noise blocks
    zero_stats[5] = (long long)(dbg_t[4] - dbg_t[2]);      // contrast step on the main table + best-strategy bookkeeping
    zero_stats[6] = (long long)(dbg_t[5] - dbg_t[4]);      // list copies / deficit contrast
    zero_stats[7] = (long long)(t_end - dbg_t[5]);         // row sums, derive_state, zeroing
  }
#endif
}

#endif      // EG_TU_THROUGHPUT
}  // namespace

// the throughput kernels (one wave per episode), from their own object (see the top of the kernel section)
int launch_rollout_throughput(int kind, const DevTables& t, const DevSnapshot& s, const DevOut& o, uint64_t seed, uint64_t first_index, uint32_t n,
                              const uint8_t* d_replay_mask, uint32_t replay_period, long long* d_stats, uint32_t count, uint32_t mode,
                              const uint32_t* index, uint32_t off, uint32_t period, const unsigned long long* hoist, unsigned long long hoist_seq,
                              uint32_t stats_rep, unsigned long long* solo, unsigned long long solo_seq, void* stream, void* ev0, void* ev1);
#ifdef EG_TU_THROUGHPUT
int launch_rollout_throughput(int kind, const DevTables& t, const DevSnapshot& s, const DevOut& o, uint64_t seed, uint64_t first_index, uint32_t n,
                              const uint8_t* d_replay_mask, uint32_t replay_period, long long* d_stats, uint32_t count, uint32_t mode,
                              const uint32_t* index, uint32_t off, uint32_t period, const unsigned long long* hoist, unsigned long long hoist_seq,
                              uint32_t stats_rep, unsigned long long* solo, unsigned long long solo_seq, void* stream, void* ev0, void* ev1) {
  EpisodeMap map{};
  map.count = count; map.mode = mode; map.index = index; map.off = off; map.period = period; map.hoist = hoist; map.hoist_seq = hoist_seq; map.stats_rep = stats_rep;
  map.solo = solo; map.solo_seq = solo_seq;
  // the timing events ride on the dispatch packet itself (no separate barrier packets around the kernel)
#define EG_LAUNCH_TP(kKind) hipExtLaunchKernelGGL((k_rollout<0, kKind>), dim3(map.count), dim3(kWave), 0, (hipStream_t)stream, (hipEvent_t)ev0, (hipEvent_t)ev1, 0, \
                                                  t, s, o, (unsigned long long)seed, (unsigned long long)first_index, n, d_replay_mask, replay_period, d_stats, map)
  if (kind == kReplayLong) {
    // every long replay episode on its own wave, script / placements / rows one after the other (eg_replay_solo.h); the classic variant
    // behind it runs what that kernel left undone (a script that needs a seeded draw or hits a capacity) and carries the stop event
    if (solo_seq != 0ull)
      hipLaunchKernelGGL(k_replay_solo, dim3(map.count), dim3(kWave), 0, (hipStream_t)stream, t, s, o, (unsigned long long)first_index, n, d_replay_mask, replay_period,
                         d_stats, map);
    EG_LAUNCH_TP(kReplayLong);
  }
  else if (kind == kReplayShort) EG_LAUNCH_TP(kReplayShort);
  else EG_LAUNCH_TP(kLean);
#undef EG_LAUNCH_TP
  return (int)hipGetLastError();
}
#else
namespace {
template <int kKind>
void launch_variant(bool helper_waves, const DevTables& t, const DevSnapshot& s, const DevOut& o, uint64_t seed, uint64_t first_index, uint32_t n,
                    const uint8_t* d_replay_mask, uint32_t replay_period, long long* d_stats, const EpisodeMap& map, void* stream, void* ev0, void* ev1) {
  if (helper_waves)
    hipExtLaunchKernelGGL((k_rollout<kHelperWaves, kKind>), dim3(map.count), dim3(kWave * (1 + kHelperWaves)), 0, (hipStream_t)stream,
                          (hipEvent_t)ev0, (hipEvent_t)ev1, 0, t, s, o, (unsigned long long)seed, (unsigned long long)first_index, n,
                          d_replay_mask, replay_period, d_stats, map);
  else
    (void)launch_rollout_throughput(kKind, t, s, o, seed, first_index, n, d_replay_mask, replay_period, d_stats, map.count, map.mode, map.index, map.off, map.period,
                                    map.hoist, map.hoist_seq, map.stats_rep, map.solo, map.solo_seq, stream, ev0, ev1);
}
}  // namespace

int launch_rollout(const DevTables& t, const DevSnapshot& s, const DevOut& o, uint64_t seed, uint64_t first_index,
                   uint32_t n, const uint8_t* d_replay_mask, uint32_t replay_period, long long* d_stats_packet, const RolloutPlan& p) {
  if (n == 0) return 0;
  // the statistics go to the replicated array when the plan brings one (k_fold_stats, launched by the caller behind the grids, folds it
  // into the packet)
  const bool rep = d_stats_packet != nullptr && p.d_stats_rep != nullptr;
  long long* d_stats = rep ? p.d_stats_rep : d_stats_packet;
  if (p.n_heavy > 0) {      // first, so that the long episodes start first: both replay variants, one of which returns at once
    EpisodeMap m{};
    m.count = p.n_heavy; m.stats_rep = rep ? 1u : 0u;
    if (p.n_lean == 0) m.mode = 0u;
    else if (p.mode == 1u) { m.mode = 1u; m.index = p.d_index; }
    else { m.mode = 2u; m.off = p.off; m.period = p.period; }
    if (!p.helper_waves && p.solo_seq != 0ull) { m.solo = p.d_solo; m.solo_seq = p.solo_seq; }
    const bool hoist = p.hoist_seq != 0ull;
    if (hoist) {
      // Replay hoist: ONE workgroup computes the batch's replay script and its placements (k_replay_coop, four waves with the whole
      // register file and LDS of a CU — it must be dispatched BEFORE the lean grid fills every CU: the lean grid's stream waits for an
      // event recorded right in front of it), k_replay_books the yearly rows; the per-episode variants follow and return at once when
      // that has succeeded, k_replay_broadcast hands out the record.
      m.hoist = reinterpret_cast<const unsigned long long*>(p.d_hoist); m.hoist_seq = p.hoist_seq;      // (HoistInfo::served_seq comes first)
      if (p.go_event && p.n_lean > 0) {
        (void)hipEventRecord((hipEvent_t)p.go_event, (hipStream_t)p.stream_heavy);
        (void)hipStreamWaitEvent((hipStream_t)p.stream_lean, (hipEvent_t)p.go_event, 0);
      }
      const DevOut scratch{p.coop_out, reinterpret_cast<double*>(p.coop_out + rec::stride)};
      hipExtLaunchKernelGGL(k_replay_coop, dim3(1), dim3(coop::kThreads), 0, (hipStream_t)p.stream_heavy, (hipEvent_t)p.ev[0], nullptr, 0,
                            t, s, scratch, p.hoist_seq, p.d_hoist, p.coop_force);
      hipLaunchKernelGGL(k_replay_books, dim3(EG_YEARS), dim3(kWave), 0, (hipStream_t)p.stream_heavy, t, s, scratch, p.hoist_seq, p.d_hoist, d_stats, p.n_heavy, rep ? 1 : 0);
    }
    // (the short one first: when it is the one that returns at once it finds the chip empty and is gone in microseconds; a
    //  256-register wave of the long one, when IT has nothing to do, must wait until a SIMD full of lean waves has drained two
    //  of them, and whatever is queued behind it on the stream waits with it — measured: 0.6 ms)
    launch_variant<kReplayShort>(p.helper_waves, t, s, o, seed, first_index, n, d_replay_mask, replay_period, d_stats, m, p.stream_heavy, hoist ? nullptr : p.ev[0], nullptr);
    if (!hoist && p.go_event && p.n_lean > 0) {
      (void)hipEventRecord((hipEvent_t)p.go_event, (hipStream_t)p.stream_heavy);
      (void)hipStreamWaitEvent((hipStream_t)p.stream_lean, (hipEvent_t)p.go_event, 0);
    }
    // (the short variant's launch carries the start event, the long one's the stop event; without the long one a marker does)
    if (!p.skip_long) launch_variant<kReplayLong>(p.helper_waves, t, s, o, seed, first_index, n, d_replay_mask, replay_period, d_stats, m, p.stream_heavy, nullptr, hoist ? nullptr : p.ev[1]);
    else if (!hoist) (void)hipEventRecord((hipEvent_t)p.ev[1], (hipStream_t)p.stream_heavy);
    if (hoist) {
      const DevOut scratch{p.coop_out, reinterpret_cast<double*>(p.coop_out + rec::stride)};
      hipExtLaunchKernelGGL(k_replay_broadcast, dim3(m.count), dim3(kWave), 0, (hipStream_t)p.stream_heavy, nullptr, (hipEvent_t)p.ev[1], 0,
                            s, scratch, o, n, d_stats != nullptr ? 1 : 0, m, (const HoistInfo*)p.d_hoist, p.hoist_seq);
    }
  }
  if (p.n_lean > 0) {
    EpisodeMap m{};
    m.count = p.n_lean; m.stats_rep = rep ? 1u : 0u;
    if (p.n_heavy == 0) m.mode = 0u;
    else if (p.mode == 1u) { m.mode = 1u; m.index = p.d_index + p.n_heavy; }
    else { m.mode = 3u; m.off = p.off; m.period = p.period; }
    launch_variant<kLean>(p.helper_waves, t, s, o, seed, first_index, n, d_replay_mask, replay_period, d_stats, m, p.stream_lean, p.ev[2], p.ev[3]);
  }
  return (int)hipGetLastError();
}
// The replicated statistics of a batch into its packet (sums; slot 3 is a maximum), and the copies cleared for the next batch: a wave
// an entry — its 64 copies are one 512-byte line.
__global__ void __launch_bounds__(256) k_fold_stats(long long* rep, long long* stats) {
  static_assert(kStatsReplicas == kWave, "a lane a copy");
  const int lane = threadIdx.x & (kWave - 1);
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= EG_STATS_LEN) return;
  unsigned long long v = (unsigned long long)rep[(size_t)i * kStatsReplicas + lane];
  rep[(size_t)i * kStatsReplicas + lane] = 0;
#pragma unroll
  for (int sh = 32; sh >= 1; sh >>= 1) {
    const unsigned long long o = (unsigned long long)__shfl_xor((long long)v, sh);
    v = i == 3 ? (o > v ? o : v) : v + o;
  }
  if (lane == 0) {
    const unsigned long long old = (unsigned long long)stats[i];
    stats[i] = (long long)(i == 3 ? (v > old ? v : old) : old + v);
  }
}
int launch_fold_stats(long long* d_rep, long long* d_stats, void* stream) {
  hipLaunchKernelGGL(k_fold_stats, dim3((EG_STATS_LEN + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_rep, d_stats);
  return (int)hipGetLastError();
}
int launch_place(const DevTables& t, int gen_type, int year_index, const uint16_t* d_cells, int n_extra,
                 int32_t* d_out_cell, double* d_out_score, void* stream) {
  hipLaunchKernelGGL(k_place, dim3(1), dim3(kWave), 0, (hipStream_t)stream, t, gen_type, year_index, d_cells, n_extra,
                     d_out_cell, d_out_score);
  return (int)hipGetLastError();
}
int launch_place_xy(const DevTables& t, int gen_type, int year_index, const double* d_x, const double* d_y, int n, double radius,
                    double size_term, int32_t* d_out_cell, double* d_out_score, void* stream) {
  hipLaunchKernelGGL(k_place_xy, dim3(1), dim3(kWave), 0, (hipStream_t)stream, t, gen_type, year_index, d_x, d_y, n, radius, size_term,
                     d_out_cell, d_out_score);
  return (int)hipGetLastError();
}
// Test hook: leaves `value` in every LDS word a later workgroup can be handed (tests/test_gpu_parity.py uses it to show
// that no kernel reads LDS it has not written).  One workgroup of 64 KB per slot, enough of them to cover every CU.
__global__ void __launch_bounds__(256) k_fill_lds(uint32_t value, uint32_t* sink) {
  __shared__ uint32_t words[16384];
  for (int i = threadIdx.x; i < 16384; i += 256) words[i] = value;
  __syncthreads();
  if (words[(threadIdx.x * 61u + blockIdx.x) & 16383u] != value) *sink = 1u;      // keeps the stores alive
}
// Diagnostic (eg_debug_occupy; scripts/side_kernel_probe.py): ONE workgroup that holds `kBytes` of LDS and its registers for `cycles` shader
// cycles and does nothing — what does a resident workgroup of that footprint cost the grid that runs beside it?
template <int kBytes, int kRegs, int kBusy = 0>
__global__ void __launch_bounds__(256, 1) k_occupy(unsigned long long cycles, uint32_t* sink) {
  __shared__ uint32_t words[kBytes / 4];
  double r[kRegs];
#pragma unroll
  for (int i = 0; i < kRegs; ++i) r[i] = (double)(threadIdx.x + i);
  words[threadIdx.x] = threadIdx.x;
  const unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < cycles) {
#pragma unroll
    for (int i = 0; i < kRegs; ++i) r[i] = r[i] * 1.0000001 + (double)words[(threadIdx.x + i) & 255];
    if (kBusy == 0) __builtin_amdgcn_s_sleep(8);
    if (kBusy >= 2) { words[(threadIdx.x * 7 + 1) & 255] = (uint32_t)r[0]; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
  }
  double acc = 0.0;
#pragma unroll
  for (int i = 0; i < kRegs; ++i) acc += r[i];
  if (acc == 12345.678) *sink = 1u;      // keeps the registers alive
}
int launch_occupy(int variant, unsigned long long cycles, uint32_t* d_sink, void* stream) {
  if (variant == 0) hipLaunchKernelGGL((k_occupy<1024, 4>), dim3(1), dim3(256), 0, (hipStream_t)stream, cycles, d_sink);            // small in every way
  else if (variant == 1) hipLaunchKernelGGL((k_occupy<150 * 1024, 4>), dim3(1), dim3(256), 0, (hipStream_t)stream, cycles, d_sink);  // the whole LDS of a CU
  else if (variant == 2) hipLaunchKernelGGL((k_occupy<1024, 100>), dim3(1), dim3(256), 0, (hipStream_t)stream, cycles, d_sink);       // 200+ registers a lane
  else if (variant == 3) hipLaunchKernelGGL((k_occupy<150 * 1024, 100>), dim3(1), dim3(256), 0, (hipStream_t)stream, cycles, d_sink);  // both
  else if (variant == 4) hipLaunchKernelGGL((k_occupy<150 * 1024, 100, 1>), dim3(1), dim3(256), 0, (hipStream_t)stream, cycles, d_sink);  // both, busy (no sleep)
  else hipLaunchKernelGGL((k_occupy<150 * 1024, 100, 2>), dim3(1), dim3(256), 0, (hipStream_t)stream, cycles, d_sink);                   // both, busy, LDS writes and barriers
  return (int)hipGetLastError();
}
int launch_fill_lds(uint32_t value, uint32_t* d_sink, int n_workgroups, void* stream) {
  hipLaunchKernelGGL(k_fill_lds, dim3(n_workgroups), dim3(256), 0, (hipStream_t)stream, value, d_sink);
  return (int)hipGetLastError();
}
__global__ void __launch_bounds__(1024) k_rewind(uint8_t* snap_base, const uint8_t* held, uint32_t* list_len_out) {
  static_assert(snap::total % 16 == 0 && (snap::state + offsetof(DevState, failed_total)) % 4 == 0, "snapshot layout");
  constexpr uint32_t kFailedWord = (uint32_t)((snap::state + offsetof(DevState, failed_total)) / 4);
  const uint32_t failed = reinterpret_cast<const uint32_t*>(snap_base)[kFailedWord];
  __syncthreads();      // (one workgroup: everybody has read the counter before anybody overwrites it)
  const uint4* src = reinterpret_cast<const uint4*>(held);
  uint4* dst = reinterpret_cast<uint4*>(snap_base);
  for (uint32_t i = threadIdx.x; i < (uint32_t)(snap::total / 16); i += 1024u) dst[i] = src[i];
  __syncthreads();
  if (threadIdx.x == 0) reinterpret_cast<uint32_t*>(snap_base)[kFailedWord] = failed;
  if (threadIdx.x == 64 && list_len_out) {
    const DevState* hs = reinterpret_cast<const DevState*>(held + snap::state);
    const uint32_t len = hs->has_lists ? (uint32_t)reinterpret_cast<const int32_t*>(held + snap::best_off)[EG_YEARS] : 0u;
    __hip_atomic_store(list_len_out, len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
int launch_rewind(uint8_t* d_snap, const uint8_t* d_held, uint32_t* list_len_out, void* stream) {
  hipLaunchKernelGGL(k_rewind, dim3(1), dim3(1024), 0, (hipStream_t)stream, d_snap, d_held, list_len_out);
  return (int)hipGetLastError();
}
int launch_stalled_tables(uint8_t* d_snap, void* stream) {
  hipLaunchKernelGGL(k_stalled_tables, dim3(EG_YEARS), dim3(kWave), 0, (hipStream_t)stream, d_snap);
  return (int)hipGetLastError();
}
int launch_apply_update(uint8_t* d_snap, const void* d_packets, int n_packets, long long* d_zero_stats, uint64_t noise_seed,
                        const DevOut& o, uint32_t n_local, uint64_t first_index, bool local_pick, uint32_t* list_len_out, void* stream) {
  hipLaunchKernelGGL(k_apply_update, dim3(1), dim3(1024), 0, (hipStream_t)stream, d_snap, (const uint8_t*)d_packets, n_packets,
                     d_zero_stats, (unsigned long long)noise_seed, (const uint8_t*)o.base, (const double*)o.score_list, n_local, (unsigned long long)first_index,
                     local_pick ? 1 : 0, list_len_out);
  return (int)hipGetLastError();
}
int launch_update_stats(const DevSnapshot& s, const DevOut& o, uint32_t n, long long* d_stats, void* stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_update_stats, dim3(n), dim3(kWave), 0, (hipStream_t)stream, o, s, n, d_stats);
  return (int)hipGetLastError();
}
int launch_fold_best(const DevOut& o, uint32_t n, uint64_t first_index, bool cost_only, uint8_t* d_fold, void* stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_fold_best, dim3(1), dim3(1024), 0, (hipStream_t)stream, o, n, (unsigned long long)first_index, cost_only ? 1 : 0, d_fold);
  return (int)hipGetLastError();
}
int launch_pick_best(const DevOut& o, uint32_t n, uint64_t first_index, UpdateCandidate* d_cand, void* stream) {
  hipLaunchKernelGGL(k_pick_best, dim3(1), dim3(1024), 0, (hipStream_t)stream, o, n, (unsigned long long)first_index, d_cand);
  return (int)hipGetLastError();
}

#endif      // EG_TU_THROUGHPUT

}  // namespace eg
