"""ctypes binding of the CPU oracle (oracle/libeg_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under eirgrid_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libeg_oracle.so")

YEARS, NA, ND, NC = 26, 61, 15, 21
YEARLY_FIELDS = 21
LOG_CAP = 4096


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (seconds)."""
    src = os.path.join(_HERE, "eg_oracle.c")
    deps = [src, os.path.join(_HERE, "eg_oracle.h"), os.path.join(_HERE, "Makefile"),
            os.path.join(os.path.dirname(_HERE), "include", "eg_detpow.h")]
    stale = (not os.path.exists(_LIB_PATH)) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(d) for d in deps)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B", "libeg_oracle.so"], check=True, capture_output=True)
    return _LIB_PATH


class EpisodeOut(C.Structure):
    _fields_ = [
        ("metrics", C.c_double * 4),
        ("yearly", (C.c_double * YEARLY_FIELDS) * YEARS),
        ("n_run", C.c_int32 * YEARS),
        ("n_def", C.c_int32 * YEARS),
        ("n_act", C.c_int32 * YEARS),
        ("n_gens", C.c_int32),
        ("n_offsets", C.c_int32),
        ("run_log", C.c_uint8 * LOG_CAP),
        ("def_log", C.c_uint8 * LOG_CAP),
        ("act_log", C.c_uint8 * LOG_CAP),
        ("gen_cell", C.c_uint16 * LOG_CAP),
        ("gen_type", C.c_uint8 * LOG_CAP),
        ("gen_year", C.c_uint8 * LOG_CAP),
        ("gen_mult", C.c_uint8 * LOG_CAP),
        ("off_type", C.c_uint8 * LOG_CAP),
        ("off_year", C.c_uint8 * LOG_CAP),
        ("off_mult", C.c_uint8 * LOG_CAP),
        ("status", C.c_int32),
        ("n_draws", C.c_uint64),
    ]


class Tables(C.Structure):
    _fields_ = [(n, C.POINTER(C.c_double)) for n in ("usage", "population", "pre_co2", "pre_tg", "pre_ig", "pre_sg", "pre_optot")] + \
               [("pre_opcnt", C.POINTER(C.c_int32))] + \
               [(n, C.POINTER(C.c_double)) for n in ("te", "coastf", "dr")] + [("size_factor", C.c_double)] + \
               [(n, C.POINTER(C.c_double)) for n in ("m03", "t12", "cc", "out_mw", "co2_t")] + \
               [(n, C.POINTER(C.c_int32)) for n in ("cls", "rclass", "marine", "reach")] + \
               [(n, C.POINTER(C.c_double)) for n in ("offv", "offc", "inflation", "carbon_price")] + [("n_existing", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    dp, u32p, i32p, u8p, u64p = (C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.POINTER(C.c_int32),
                                 C.POINTER(C.c_uint8), C.POINTER(C.c_uint64))
    L.og_world_create.restype = C.c_void_p
    L.og_world_create.argtypes = [C.c_int32, dp, dp, u32p, C.c_int32, dp, dp, i32p, dp, C.c_int32, dp, dp, C.c_int32]
    L.og_world_destroy.argtypes = [C.c_void_p]
    L.og_world_demand.argtypes = [C.c_void_p, C.c_int32, u32p, dp]
    L.og_world_existing_online.restype = C.c_int32
    L.og_world_existing_online.argtypes = [C.c_void_p, C.c_int32]
    L.og_weights_new.restype = C.c_void_p
    L.og_weights_clone.restype = C.c_void_p
    L.og_weights_clone.argtypes = [C.c_void_p]
    L.og_weights_free.argtypes = [C.c_void_p]
    L.og_weights_get_tables.argtypes = [C.c_void_p, dp, dp, dp]
    L.og_weights_set_tables.argtypes = [C.c_void_p, dp, dp, dp]
    L.og_weights_set_has_count_weights.argtypes = [C.c_void_p, C.c_int32]
    L.og_weights_get_scalar.restype = C.c_double
    L.og_weights_get_scalar.argtypes = [C.c_void_p, C.c_int32]
    L.og_weights_set_scalar.argtypes = [C.c_void_p, C.c_int32, C.c_double]
    L.og_weights_get_list.restype = C.c_int32
    L.og_weights_get_list.argtypes = [C.c_void_p, C.c_int32, C.c_int32, u8p, C.c_int32]
    L.og_weights_set_list.argtypes = [C.c_void_p, C.c_int32, C.c_int32, u8p, C.c_int32]
    L.og_weights_get_best_weights.argtypes = [C.c_void_p, dp]
    L.og_run_episode.restype = C.c_int32
    L.og_run_episode.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint64, C.c_int32, C.c_int32, C.POINTER(EpisodeOut)]
    L.og_run_episode_tabled.restype = C.c_int32
    L.og_run_episode_tabled.argtypes = [C.POINTER(Tables), C.c_void_p, C.c_int32, C.c_uint64, C.c_int32, C.POINTER(EpisodeOut)]
    L.og_post_episode_update.argtypes = [C.c_void_p, C.c_void_p, dp, C.c_uint64]
    L.og_delay_deficit_probe.restype = C.c_int32
    L.og_delay_deficit_probe.argtypes = [C.c_void_p, C.c_int32, dp, i32p, dp, i32p, i32p]
    L.og_reduced_batch_update.restype = C.c_int32
    L.og_reduced_batch_update.argtypes = [C.c_void_p, C.c_int32, i32p, dp, i32p, i32p, u8p, C.c_int32, u8p, C.c_int32,
                                          C.c_uint64, C.POINTER(C.c_int64), i32p]
    L.og_score_metrics.restype = C.c_double
    L.og_score_metrics.argtypes = [dp, C.c_int32]
    L.og_evaluate_action_impact.restype = C.c_double
    L.og_evaluate_action_impact.argtypes = [dp, dp, C.c_int32]
    L.og_fold_best_result.restype = C.c_int32
    L.og_fold_best_result.argtypes = [C.c_int32, C.POINTER(C.c_int32), dp, C.c_int32, C.c_int64, C.POINTER(C.c_int32), dp, C.POINTER(C.c_int64)]
    L.og_carbon_price.restype = C.c_double
    L.og_carbon_price.argtypes = [C.c_int32]
    L.og_type_power_output.restype = C.c_double
    L.og_type_power_output.argtypes = [C.c_int32]
    L.og_offset_full_effect.restype = C.c_double
    L.og_offset_full_effect.argtypes = [C.c_int32]
    L.og_generator_cost.restype = C.c_double
    L.og_generator_cost.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    L.og_action_cost_estimate.restype = C.c_double
    L.og_action_cost_estimate.argtypes = [C.c_int32, C.c_int32]
    L.og_place.restype = C.c_int32
    L.og_place.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, dp, dp, dp]
    L.og_place_sized.restype = C.c_int32
    L.og_place_sized.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, dp, dp, C.c_float, dp]
    L.og_chacha_block.argtypes = [u32p, C.c_uint64, C.c_uint64, C.c_int32, u32p]
    L.og_rng_seed_words.argtypes = [C.c_uint64, u32p]
    L.og_rng_stream.argtypes = [C.c_uint64, C.c_int32, u64p]
    L.og_rng_gen_range_probe.restype = C.c_uint64
    L.og_rng_gen_range_probe.argtypes = [C.c_uint64, C.c_uint64, C.c_int32]
    L.og_detpow.restype = C.c_double
    L.og_detpow.argtypes = [C.c_double, C.c_double]
    L.og_libm_pow.restype = C.c_double
    L.og_libm_pow.argtypes = [C.c_double, C.c_double]
    L.og_set_libm_pow.argtypes = [C.c_int32]
    L.og_get_libm_pow.restype = C.c_int32
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class libm_pow:
    """Context manager: inside it the oracle's stalled sampler (sampling.rs:199-213) calls libm's pow, as the reference's f64::powf
    does, instead of the shared eg_detpow (oracle/eg_oracle.c og_set_libm_pow)."""

    def __enter__(self):
        self.before = lib().og_get_libm_pow()
        lib().og_set_libm_pow(1)
        return self

    def __exit__(self, *exc):
        lib().og_set_libm_pow(self.before)
        return False


class OracleWorld:
    def __init__(self, world):
        L = lib()
        self._keep = [np.ascontiguousarray(world.settlement_x, dtype=np.float64),
                      np.ascontiguousarray(world.settlement_y, dtype=np.float64),
                      np.ascontiguousarray(world.settlement_pop, dtype=np.uint32),
                      np.ascontiguousarray(world.existing_x, dtype=np.float64),
                      np.ascontiguousarray(world.existing_y, dtype=np.float64),
                      np.ascontiguousarray(world.existing_type, dtype=np.int32),
                      np.ascontiguousarray(world.existing_capacity, dtype=np.float64),
                      np.ascontiguousarray(world.coast_x, dtype=np.float64),
                      np.ascontiguousarray(world.coast_y, dtype=np.float64)]
        k = self._keep
        self.n_existing = len(k[3])
        self.h = L.og_world_create(len(k[0]), _dp(k[0]), _dp(k[1]), k[2].ctypes.data_as(C.POINTER(C.c_uint32)),
                                   len(k[3]), _dp(k[3]), _dp(k[4]), k[5].ctypes.data_as(C.POINTER(C.c_int32)), _dp(k[6]),
                                   len(k[7]), _dp(k[7]), _dp(k[8]), int(world.existing_operational_at_start))

    def __del__(self):
        if getattr(self, "h", None):
            lib().og_world_destroy(self.h)
            self.h = None

    def demand(self, yi):
        pop, usage = C.c_uint32(), C.c_double()
        lib().og_world_demand(self.h, yi, C.byref(pop), C.byref(usage))
        return pop.value, usage.value

    def existing_online(self):
        return [lib().og_world_existing_online(self.h, g) for g in range(self.n_existing)]

    def place(self, yi, gen_type, extra_xy=(), size_penalty=1.0):
        ex = np.array([p[0] for p in extra_xy], dtype=np.float64)
        ey = np.array([p[1] for p in extra_xy], dtype=np.float64)
        score = C.c_double()
        cell = lib().og_place_sized(self.h, yi, gen_type, len(ex), _dp(ex), _dp(ey), C.c_float(size_penalty), C.byref(score))
        return cell, score.value


class OracleWeights:
    """ActionWeights (ai/learning/weights/mod.rs:50-107) held by the oracle."""
    SC = dict(learning_rate=0, exploration_rate=1, stall=2, iteration_count=3, has_best=4,
              best_net_emissions=5, best_opinion=6, best_cost=7, best_reliability=8,
              has_best_actions=9, has_best_deficit_actions=10)

    def __init__(self, handle=None):
        self.h = handle if handle is not None else lib().og_weights_new()

    def __del__(self):
        if getattr(self, "h", None):
            lib().og_weights_free(self.h)
            self.h = None

    def clone(self):
        return OracleWeights(lib().og_weights_clone(self.h))

    def tables(self):
        w = np.zeros((YEARS, NA)); dw = np.zeros((YEARS, ND)); cw = np.zeros((YEARS, NC))
        lib().og_weights_get_tables(self.h, _dp(w), _dp(dw), _dp(cw))
        return w, dw, cw

    def set_tables(self, w=None, dw=None, cw=None):
        args = []
        for a, shape in ((w, (YEARS, NA)), (dw, (YEARS, ND)), (cw, (YEARS, NC))):
            if a is None:
                args.append(None)
            else:
                a = np.ascontiguousarray(a, dtype=np.float64)
                assert a.shape == shape
                self._tmp = getattr(self, "_tmp", []) + [a]
                args.append(_dp(a))
        lib().og_weights_set_tables(self.h, *args)

    def best_weights(self):
        w = np.zeros((YEARS, NA))
        lib().og_weights_get_best_weights(self.h, _dp(w))
        return w

    def get(self, name):
        return lib().og_weights_get_scalar(self.h, self.SC[name])

    def set(self, name, v):
        lib().og_weights_set_scalar(self.h, self.SC[name], float(v))

    def set_has_count_weights(self, has):
        lib().og_weights_set_has_count_weights(self.h, int(has))

    def get_list(self, which, yi):
        buf = (C.c_uint8 * LOG_CAP)()
        n = lib().og_weights_get_list(self.h, which, yi, buf, LOG_CAP)
        return list(buf[:n])

    def set_list(self, which, yi, values):
        arr = (C.c_uint8 * max(1, len(values)))(*values)
        lib().og_weights_set_list(self.h, which, yi, arr, len(values))

    def lists(self, which):
        return [self.get_list(which, y) for y in range(YEARS)]


def run_episode(world: OracleWorld, weights: OracleWeights, seed: int, replay: bool = False,
                energy_sales: bool = True, delays: bool = False):
    """core/iteration.rs:10-95 run_iteration.  `weights` is mutated exactly as the reference mutates it."""
    out = EpisodeOut()
    st = lib().og_run_episode(world.h, weights.h, int(replay), C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), int(energy_sales),
                              int(delays), C.byref(out))
    return st, out


class OracleTables:
    """Tables for the oracle's tabled mode.  `source` provides f64(name)/i32(name) arrays — in the tests that is the
    product library's eg_host_tables view, so running the tabled mode validates those tables."""
    F64 = ("usage", "population", "pre_co2", "pre_tg", "pre_ig", "pre_sg", "pre_optot", "te", "coastf", "dr", "m03", "t12",
           "cc", "out_mw", "co2_t", "offv", "offc", "inflation", "carbon_price")
    I32 = ("pre_opcnt", "cls", "rclass", "marine", "reach")

    def __init__(self, source, n_existing):
        self.arrays = {}
        t = Tables()
        for n in self.F64:
            a = np.ascontiguousarray(source.f64(n), dtype=np.float64); self.arrays[n] = a
            setattr(t, n, a.ctypes.data_as(C.POINTER(C.c_double)))
        for n in self.I32:
            a = np.ascontiguousarray(source.i32(n), dtype=np.int32); self.arrays[n] = a
            setattr(t, n, a.ctypes.data_as(C.POINTER(C.c_int32)))
        t.size_factor = float(source.f64("size_factor")[0])
        t.n_existing = int(n_existing)
        self.t = t


def run_episode_tabled(tables: OracleTables, weights: OracleWeights, seed: int, replay: bool = False, energy_sales: bool = True):
    out = EpisodeOut()
    st = lib().og_run_episode_tabled(C.byref(tables.t), weights.h, int(replay), C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF),
                                     int(energy_sales), C.byref(out))
    return st, out


def split_log(log, counts):
    res, pos = [], 0
    for n in counts:
        res.append(list(log[pos:pos + n]))
        pos += n
    return res


def post_episode_update(shared: OracleWeights, local: OracleWeights, metrics, noise_seed: int = 0):
    m = np.ascontiguousarray(metrics, dtype=np.float64)
    lib().og_post_episode_update(shared.h, local.h, _dp(m), C.c_uint64(noise_seed))


def delay_deficit_probe(world: OracleWorld, trips: int):
    """The repair loop of the first year with a deficit, with construction delays on, stopped after `trips` iterations
    (oracle/eg_oracle.c og_delay_deficit_probe): (status, initial deficit MW, remaining deficit after each trip, active plant after
    each trip, plants added, year index of that year)."""
    rem = np.zeros(trips); act = np.zeros(trips, np.int32); d0 = C.c_double(); added = C.c_int32(); year = C.c_int32()
    st = lib().og_delay_deficit_probe(world.h, trips, _dp(rem), act.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(d0), C.byref(added), C.byref(year))
    return st, d0.value, rem, act, added.value, year.value


STATS_LEN = 8 + 2 * YEARS * NA + YEARS * ND


def reduced_batch_update(shared: OracleWeights, status, metrics, n_run, n_def, run_log, def_log, noise_seed: int = 0):
    """Batch ("reduced") update of SURVEY.md §8(e) / DESIGN.md §2.4 for n episodes that shared the snapshot `shared`
    (independent libm restatement, oracle/eg_oracle.c og_reduced_batch_update).  Arrays are episode-major: status [n],
    metrics [n,4], n_run / n_def [n,26], run_log / def_log [n, stride] flat year-major.
    Returns (improved, stats int64[STATS_LEN], winner index or -1)."""
    st = np.ascontiguousarray(status, dtype=np.int32); n = st.shape[0]
    m = np.ascontiguousarray(metrics, dtype=np.float64).reshape(n, 4)
    nr = np.ascontiguousarray(n_run, dtype=np.int32).reshape(n, YEARS); nd = np.ascontiguousarray(n_def, dtype=np.int32).reshape(n, YEARS)
    rl = np.ascontiguousarray(run_log, dtype=np.uint8).reshape(n, -1); dl = np.ascontiguousarray(def_log, dtype=np.uint8).reshape(n, -1)
    ok = st == 0
    assert (nr[ok].sum(axis=1) <= rl.shape[1]).all() and (nd[ok].sum(axis=1) <= dl.shape[1]).all()
    stats = np.zeros(STATS_LEN, np.int64); winner = C.c_int32(-1)
    i32p, u8p = C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
    improved = lib().og_reduced_batch_update(shared.h, n, st.ctypes.data_as(i32p), _dp(m), nr.ctypes.data_as(i32p), nd.ctypes.data_as(i32p),
                                             rl.ctypes.data_as(u8p), rl.shape[1], dl.ctypes.data_as(u8p), dl.shape[1],
                                             C.c_uint64(noise_seed & 0xFFFFFFFFFFFFFFFF), stats.ctypes.data_as(C.POINTER(C.c_int64)),
                                             C.byref(winner))
    return bool(improved), stats, winner.value


def score_metrics(metrics, cost_only=False):
    m = np.ascontiguousarray(metrics, dtype=np.float64)
    return lib().og_score_metrics(_dp(m), int(cost_only))


def evaluate_action_impact(cur, nxt, cost_only=False):
    a = np.ascontiguousarray(cur, dtype=np.float64); b = np.ascontiguousarray(nxt, dtype=np.float64)
    return lib().og_evaluate_action_impact(_dp(a), _dp(b), int(cost_only))


class BestResultFold:
    """The reference's `best_result` (core/multi_simulation.rs:384, :613-620), folded batch after batch."""

    def __init__(self, cost_only=False):
        self.cost_only = bool(cost_only); self.has = C.c_int32(0); self.best = np.zeros(4); self.index = C.c_int64(-1); self.takeovers = 0

    def feed(self, status, metrics, first_index=0):
        st = np.ascontiguousarray(status, dtype=np.int32); m = np.ascontiguousarray(metrics, dtype=np.float64).reshape(-1, 4)
        assert len(st) == len(m)
        self.takeovers += lib().og_fold_best_result(len(st), st.ctypes.data_as(C.POINTER(C.c_int32)), _dp(m), int(self.cost_only), int(first_index),
                                                    C.byref(self.has), _dp(self.best), C.byref(self.index))
        return self

    @property
    def winner(self):
        return int(self.index.value) if self.has.value else None


def chacha_block(key_words, counter, stream, rounds):
    k = (C.c_uint32 * 8)(*key_words); out = (C.c_uint32 * 16)()
    lib().og_chacha_block(k, counter, stream, rounds, out)
    return list(out)


def rng_stream(seed, n):
    out = (C.c_uint64 * n)()
    lib().og_rng_stream(C.c_uint64(seed), n, out)
    return list(out)
