"""Host-side mirror of the reference interface for the rollout hot path, on top of the C ABI.

Names follow the reference: `ActionWeights` (ai/learning/weights/mod.rs:50-107), `run_iteration`
(core/iteration.rs:10-20), `SimulationMetrics` (ai/metrics/simulation_metrics.rs:5-11), `find_suitable_location`
(gpu/metal_location_search.rs:96-103).  All compute happens in libeirgrid_hip.so on the GPU; this file only
marshals numpy buffers.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _native as N
from .world import World

BASE_YEAR, END_YEAR = 2025, 2050

YEARLY_COLUMNS = [
    "year", "total_population", "total_power_usage", "total_power_generation", "power_balance",
    "average_public_opinion", "yearly_capital_cost", "total_capital_cost", "inflation_factor", "total_co2_emissions",
    "total_carbon_offset", "net_co2_emissions", "yearly_carbon_credit_revenue", "total_carbon_credit_revenue",
    "yearly_energy_sales_revenue", "total_energy_sales_revenue", "active_generators", "yearly_upgrade_costs",
    "yearly_closure_costs", "yearly_total_cost", "total_cost",
]


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _world_struct(world: World):
    keep = [np.ascontiguousarray(world.settlement_x, dtype=np.float64), np.ascontiguousarray(world.settlement_y, dtype=np.float64),
            np.ascontiguousarray(world.settlement_pop, dtype=np.uint32),
            np.ascontiguousarray(world.existing_x, dtype=np.float64), np.ascontiguousarray(world.existing_y, dtype=np.float64),
            np.ascontiguousarray(world.existing_type, dtype=np.int32), np.ascontiguousarray(world.existing_capacity, dtype=np.float64),
            np.ascontiguousarray(world.coast_x, dtype=np.float64), np.ascontiguousarray(world.coast_y, dtype=np.float64)]
    w = N.EgWorld(len(keep[0]), _p(keep[0], C.c_double), _p(keep[1], C.c_double), _p(keep[2], C.c_uint32),
                  len(keep[3]), _p(keep[3], C.c_double), _p(keep[4], C.c_double), _p(keep[5], C.c_int32), _p(keep[6], C.c_double),
                  len(keep[7]), _p(keep[7], C.c_double), _p(keep[8], C.c_double), int(world.existing_operational_at_start))
    return w, keep


class HostTables:
    """The policy-independent tables the library builds on the host for a world (no device needed)."""

    def __init__(self, world: World):
        w, self._keep = _world_struct(world)
        self.h = N.lib().eg_host_tables_create(C.byref(w))
        if not self.h:
            raise N.EirgridError(N.lib().eg_last_error().decode())

    def __del__(self):
        if getattr(self, "h", None):
            N.lib().eg_host_tables_free(self.h)
            self.h = None

    def f64(self, name: str) -> np.ndarray:
        ptr, n = C.POINTER(C.c_double)(), C.c_int64()
        N.check(N.lib().eg_host_tables_f64(self.h, name.encode(), C.byref(ptr), C.byref(n)), "eg_host_tables_f64")
        return np.ctypeslib.as_array(ptr, shape=(n.value,)).copy()

    def i32(self, name: str) -> np.ndarray:
        ptr, n = C.POINTER(C.c_int32)(), C.c_int64()
        N.check(N.lib().eg_host_tables_i32(self.h, name.encode(), C.byref(ptr), C.byref(n)), "eg_host_tables_i32")
        return np.ctypeslib.as_array(ptr, shape=(n.value,)).copy()


class ActionWeights:
    """ActionWeights held by the library (eg_policy): tables, best strategy, counters."""
    SC = dict(learning_rate=0, exploration_rate=1, iterations_without_improvement=2, iteration_count=3, has_best=4,
              best_net_emissions=5, best_opinion=6, best_cost=7, best_reliability=8, has_best_actions=9,
              has_best_deficit_actions=10, has_count_weights=11, improvement_history_len=12, failed_episodes=13)

    def __init__(self, handle=None):
        self.h = handle if handle is not None else N.lib().eg_policy_new()

    def save_to_file(self, path: str) -> None:
        """ai/learning/weights/serialization.rs:29-139 — pretty JSON in the reference's SerializableWeights schema."""
        N.check(N.lib().eg_policy_save_json(self.h, str(path).encode()), "eg_policy_save_json")

    @staticmethod
    def load_from_file(path: str) -> "ActionWeights":
        """ai/learning/weights/serialization.rs:140-493."""
        h = N.lib().eg_policy_load_json(str(path).encode())
        if not h:
            raise N.EirgridError(N.lib().eg_last_error().decode())
        return ActionWeights(h)

    def __del__(self):
        if getattr(self, "h", None):
            try:
                N.lib().eg_policy_free(self.h)
            except TypeError:      # interpreter shutdown: module globals are already gone
                pass
            self.h = None

    def tables(self):
        w = np.zeros((N.YEARS, N.N_ACTIONS)); dw = np.zeros((N.YEARS, N.N_DEFICIT)); cw = np.zeros((N.YEARS, N.N_COUNTS))
        N.check(N.lib().eg_policy_get_tables(self.h, _p(w, C.c_double), _p(dw, C.c_double), _p(cw, C.c_double)))
        return w, dw, cw

    def set_tables(self, w=None, dw=None, cw=None):
        args, keep = [], []
        for a in (w, dw, cw):
            if a is None:
                args.append(None)
            else:
                a = np.ascontiguousarray(a, dtype=np.float64); keep.append(a); args.append(_p(a, C.c_double))
        N.check(N.lib().eg_policy_set_tables(self.h, *args))

    def get(self, name: str) -> float:
        return N.lib().eg_policy_get_scalar(self.h, self.SC[name])

    def set(self, name: str, v) -> None:
        N.check(N.lib().eg_policy_set_scalar(self.h, self.SC[name], float(v)))

    def get_list(self, which: int, yi: int):
        buf = (C.c_uint8 * 4096)()
        n = N.lib().eg_policy_get_list(self.h, which, yi, buf, 4096)
        return list(buf[:n])

    def lists(self, which: int):
        return [self.get_list(which, y) for y in range(N.YEARS)]

    def snapshot(self) -> N.EgPolicySnapshot:
        s = N.EgPolicySnapshot()
        N.check(N.lib().eg_policy_snapshot_view(self.h, C.byref(s)))
        return s

    def apply_episode(self, metrics, n_run, run_log, n_def, def_log, noise_seed: int = 0):
        """core/multi_simulation.rs:494-508 for one finished episode."""
        m = np.ascontiguousarray(metrics, dtype=np.float64)
        nr = np.ascontiguousarray(n_run, dtype=np.int32); nd = np.ascontiguousarray(n_def, dtype=np.int32)
        rl = np.ascontiguousarray(run_log, dtype=np.uint8); dl = np.ascontiguousarray(def_log, dtype=np.uint8)
        N.check(N.lib().eg_policy_apply_episode(self.h, _p(m, C.c_double), _p(nr, C.c_int32), _p(rl, C.c_uint8),
                                                _p(nd, C.c_int32), _p(dl, C.c_uint8), C.c_uint64(noise_seed)))


def apply_reduced(weights: "ActionWeights", stats, candidate, noise_seed: int = 0) -> bool:
    """Batch form of core/multi_simulation.rs:494-508 (SURVEY.md §8(e) reduced mode).  `stats` is the all-reduced host copy
    of the eg_update_stats buffer (int64[STATS_LEN]); `candidate` = (metrics, n_run, run_log, n_def, def_log) of the batch's
    best episode or None.  Returns True when the candidate became the best strategy."""
    st = np.ascontiguousarray(stats, dtype=np.int64)
    assert st.shape == (N.STATS_LEN,)
    if candidate is None:
        rc = N.lib().eg_policy_apply_reduced(weights.h, _p(st, C.c_int64), None, None, None, None, None, C.c_uint64(noise_seed))
    else:
        m = np.ascontiguousarray(candidate[0], dtype=np.float64)
        nr = np.ascontiguousarray(candidate[1], dtype=np.int32); rl = np.ascontiguousarray(candidate[2], dtype=np.uint8)
        nd = np.ascontiguousarray(candidate[3], dtype=np.int32); dl = np.ascontiguousarray(candidate[4], dtype=np.uint8)
        rc = N.lib().eg_policy_apply_reduced(weights.h, _p(st, C.c_int64), _p(m, C.c_double), _p(nr, C.c_int32), _p(rl, C.c_uint8),
                                             _p(nd, C.c_int32), _p(dl, C.c_uint8), C.c_uint64(noise_seed))
    if rc < 0:
        N.check(rc, "eg_policy_apply_reduced")
    return rc == 1


def apply_packet(weights: "ActionWeights", stats, candidates, noise_seed: int = 0) -> bool:
    """eg_policy_apply_packet: `stats` int64[STATS_LEN] (all-reduced), `candidates` uint8[W, CANDIDATE_BYTES] candidate
    records, one per rank.  The winner (highest score, ties to the lowest global index) competes for the best slot."""
    st = np.ascontiguousarray(stats, dtype=np.int64)
    cd = np.ascontiguousarray(candidates, dtype=np.uint8).reshape(-1, N.CANDIDATE_BYTES)
    assert st.shape == (N.STATS_LEN,)
    rc = N.lib().eg_policy_apply_packet(weights.h, st.ctypes.data, cd.ctypes.data, cd.shape[0], C.c_uint64(noise_seed))
    if rc < 0:
        N.check(rc, "eg_policy_apply_packet")
    return rc == 1


def evaluate_action_impact(current_metrics, new_metrics, cost_only: bool = False) -> float:
    """ai/metrics/scoring.rs:46-85 on SimulationMetrics quadruples (metrics_to_action_result, multi_simulation.rs:55-62)."""
    a = np.ascontiguousarray(current_metrics, dtype=np.float64); b = np.ascontiguousarray(new_metrics, dtype=np.float64)
    return N.lib().eg_evaluate_action_impact(_p(a, C.c_double), _p(b, C.c_double), int(cost_only))


def score_metrics(metrics, cost_only: bool = False) -> float:
    m = np.ascontiguousarray(metrics, dtype=np.float64)
    return N.lib().eg_score_metrics(_p(m, C.c_double), int(cost_only))


@dataclass
class BatchResult:
    metrics: np.ndarray       # [n,4] final_net_emissions, average_public_opinion, total_cost, power_reliability
    yearly: np.ndarray        # [n,26,21]
    status: np.ndarray
    n_run: np.ndarray; n_def: np.ndarray; n_act: np.ndarray
    run_log: np.ndarray; def_log: np.ndarray; act_log: np.ndarray
    n_gens: np.ndarray; gen_cell: np.ndarray; gen_pack: np.ndarray
    n_offsets: np.ndarray; off_pack: np.ndarray
    n_draws: np.ndarray; bytes_moved: np.ndarray
    n_chunks: np.ndarray = None      # [n] chunks of 64 candidate records the searches requested (see include/eirgrid_hip.h)

    @staticmethod
    def alloc(n: int) -> "BatchResult":
        z = np.zeros
        return BatchResult(z((n, 4)), z((n, N.YEARS, N.YEARLY_FIELDS)), z(n, np.int32), z((n, N.YEARS), np.int32),
                           z((n, N.YEARS), np.int32), z((n, N.YEARS), np.int32), z((n, N.RUN_CAP), np.uint8),
                           z((n, N.DEF_CAP), np.uint8), z((n, N.ACT_CAP), np.uint8), z(n, np.int32),
                           z((n, N.MAX_GENS), np.uint16), z((n, N.MAX_GENS), np.uint16), z(n, np.int32),
                           z((n, N.MAX_OFFSETS), np.uint16), z(n, np.uint64), z(n), z(n, np.uint32))

    def struct(self) -> N.EgEpisodeOut:
        return N.EgEpisodeOut(_p(self.metrics, C.c_double), _p(self.yearly, C.c_double), _p(self.status, C.c_int32),
                              _p(self.n_run, C.c_int32), _p(self.n_def, C.c_int32), _p(self.n_act, C.c_int32),
                              _p(self.run_log, C.c_uint8), _p(self.def_log, C.c_uint8), _p(self.act_log, C.c_uint8),
                              _p(self.n_gens, C.c_int32), _p(self.gen_cell, C.c_uint16), _p(self.gen_pack, C.c_uint16),
                              _p(self.n_offsets, C.c_int32), _p(self.off_pack, C.c_uint16), _p(self.n_draws, C.c_uint64),
                              _p(self.bytes_moved, C.c_double), _p(self.n_chunks, C.c_uint32))

    def export_summary_csv(self, path: str, timestamp: str = "", episode: int = 0) -> None:
        """simulation_summary.csv (utils/csv_export.rs:215-432) of one episode of this result."""
        one = BatchResult(*[np.ascontiguousarray(getattr(self, f)[episode:episode + 1]) for f in
                            ("metrics", "yearly", "status", "n_run", "n_def", "n_act", "run_log", "def_log", "act_log", "n_gens",
                             "gen_cell", "gen_pack", "n_offsets", "off_pack", "n_draws", "bytes_moved", "n_chunks")])
        out = one.struct()
        N.check(N.lib().eg_export_summary_csv(C.byref(out), path.encode(), timestamp.encode()), "eg_export_summary_csv")

    def export_run_details(self, world: World, out_dir: str, settlement_names=None, offset_seed: int = 0, episode: int = 0) -> None:
        """yearly_details/{settlements,generators,carbon_offsets}.csv + operation_logs/generator_operation_logs.csv of one
        episode of this result (utils/csv_export.rs:434-1230; eg_export_run_details)."""
        one = BatchResult(*[np.ascontiguousarray(getattr(self, f)[episode:episode + 1]) for f in
                            ("metrics", "yearly", "status", "n_run", "n_def", "n_act", "run_log", "def_log", "act_log", "n_gens",
                             "gen_cell", "gen_pack", "n_offsets", "off_pack", "n_draws", "bytes_moved", "n_chunks")])
        out = one.struct()
        w, keep = _world_struct(world)
        names = None
        if settlement_names is not None:
            names = (C.c_char_p * len(settlement_names))(*[n.encode() for n in settlement_names])
        N.check(N.lib().eg_export_run_details(C.byref(w), names, C.byref(out), str(out_dir).encode(), C.c_uint64(offset_seed & (2**64 - 1))),
                "eg_export_run_details")

    def bytes_touched(self) -> np.ndarray:
        """Per episode: the SURVEY §8(d) bytes with the score field of every search (2601 x 8 B, never read by the
        branch-and-bound search) replaced by the candidate records the search requested (n_chunks x 64 x 32 B)."""
        return self.bytes_moved - self.n_gens.astype(np.float64) * (N.CELLS * 8.0) + self.n_chunks.astype(np.float64) * (64 * 32.0)

    def bytes_requested(self) -> np.ndarray:
        """Per episode: the bytes the kernel REQUESTS from the memory system (L2 or HBM), by construction of the code — without
        the SURVEY formula's state term `2 (59 + G) 56` per year (state that lives in LDS: 3 bytes per generator) and without the
        generator coordinates a search reads (LDS as well):
          n_chunks x 2048                       sorted candidate records in chunks of 64 x 32 B; long-replay episodes: + the entries of
                                                their penalty field they gather and update (counted by the kernel)
          32 x sum_g (25 - b_g)                 year-start gathers per generator and later year: {cost, cost opinion} 16 B, m03 8 B, t12 8 B
          16 x sum_o (25 - b_o)                 per carbon offset and later year: tonnes 8 B, cost 8 B
          32 x n_gens + 24 x n_offsets          the terms of an addition itself
          26 x 1024                             the policy row block of every year (128 doubles)
          26 x 168 + 3 x 104 + 72               stores: yearly rows, per-year counts, record header
          2 x (run + def) + act + 4 x n_gens + 2 x n_offsets    stores: logs (run and def re-read by the statistics epilogue), placements"""
        n = len(self.status)
        out = np.zeros(n)
        for e in range(n):
            g, o = int(self.n_gens[e]), int(self.n_offsets[e])
            by = (self.gen_pack[e, :g].astype(np.int64) >> 4) & 31
            oy = (self.off_pack[e, :o].astype(np.int64) >> 4) & 31
            logs = 2.0 * (self.n_run[e].sum() + self.n_def[e].sum()) + self.n_act[e].sum()
            out[e] = (self.n_chunks[e] * 2048.0 + 32.0 * (25 - by).sum() + 16.0 * (25 - oy).sum() + 32.0 * g + 24.0 * o + 26 * 1024.0
                      + 26 * 168.0 + 3 * 104.0 + 72.0 + logs + 4.0 * g + 2.0 * o)
        return out

    def lists(self, e: int, which: str):
        log = {"run": self.run_log, "def": self.def_log, "act": self.act_log}[which][e]
        cnt = {"run": self.n_run, "def": self.n_def, "act": self.n_act}[which][e]
        out, pos = [], 0
        for c in cnt:
            out.append(log[pos:pos + c].tolist()); pos += int(c)
        return out


class Engine:
    """One eg_ctx: a world resident in the HBM of one MI355X."""

    def __init__(self, world: World, device: int = 0):
        L = N.lib()
        if L.eg_device_count() <= 0:
            raise N.EirgridError("no HIP device visible: eirgrid_amd runs on MI355X (gfx950) only and has no CPU path")
        w, self._keep = _world_struct(world)
        self.h = L.eg_create(device, C.byref(w))
        if not self.h:
            raise N.EirgridError(L.eg_last_error().decode())
        self.world = world

    def close(self):
        if getattr(self, "h", None):
            try:
                N.lib().eg_destroy(self.h)
            except TypeError:      # interpreter shutdown: module globals are already gone
                pass
            self.h = None

    __del__ = close

    @staticmethod
    def _opts(enable_energy_sales=True, enable_construction_delays=False, write_yearly=True):
        return N.EgOpts(int(enable_energy_sales), int(enable_construction_delays), int(write_yearly))

    def rollout_batch(self, weights: ActionWeights, seed: int, n_episodes: int, first_episode_index: int = 0,
                      replay_mask=None, enable_energy_sales=True, enable_construction_delays=False,
                      write_yearly=True, out: "BatchResult" = None) -> BatchResult:
        """Batched `run_iteration` (core/iteration.rs:10-20): episodes first..first+n against one weights snapshot.
        `out`: a BatchResult of the same size to fill again (a caller in a loop keeps its buffers; list entries behind an episode's
        counts keep whatever they held)."""
        res = out if out is not None else BatchResult.alloc(n_episodes)
        assert res.status.shape == (n_episodes,)
        snap = weights.snapshot()
        opts = self._opts(enable_energy_sales, enable_construction_delays, write_yearly)
        mask = None
        if replay_mask is not None:
            self._mask = np.ascontiguousarray(replay_mask, dtype=np.uint8)
            assert self._mask.shape == (n_episodes,)
            mask = _p(self._mask, C.c_uint8)
        out = res.struct()
        N.check(N.lib().eg_rollout_batch(self.h, C.byref(snap), C.byref(opts), C.c_uint64(seed & (2**64 - 1)),
                                         C.c_uint64(first_episode_index), n_episodes, mask, C.byref(out)), "eg_rollout_batch")
        return res

    def run_iteration(self, iteration: int, weights: ActionWeights, replay_best_strategy: bool, seed: int,
                      enable_energy_sales: bool = True, enable_construction_delays: bool = False) -> BatchResult:
        """Single-episode form with the reference's argument order (core/iteration.rs:10-20)."""
        mask = np.array([1 if replay_best_strategy else 0], dtype=np.uint8)
        return self.rollout_batch(weights, seed, 1, iteration, mask, enable_energy_sales, enable_construction_delays)

    # device-resident path used by bench.py
    def upload_snapshot(self, weights: ActionWeights, enable_energy_sales=True, write_yearly=True):
        snap = weights.snapshot()
        opts = self._opts(enable_energy_sales, False, write_yearly)
        N.check(N.lib().eg_upload_snapshot(self.h, C.byref(snap), C.byref(opts)), "eg_upload_snapshot")

    def launch(self, seed: int, first_episode_index: int, n_episodes: int, replay_mask=None):
        mask = None
        if replay_mask is not None:
            self._mask = np.ascontiguousarray(replay_mask, dtype=np.uint8)
            mask = _p(self._mask, C.c_uint8)
        N.check(N.lib().eg_rollout_launch(self.h, C.c_uint64(seed & (2**64 - 1)), C.c_uint64(first_episode_index),
                                          n_episodes, mask), "eg_rollout_launch")

    def launch_update(self, seed: int, first_episode_index: int, n_episodes: int, d_packet_ptr: int, replay_mask=None):
        """Fused step: rollout + update statistics + best-candidate pick into a PACKET_BYTES device buffer."""
        mask = None
        if replay_mask is not None:
            self._mask = np.ascontiguousarray(replay_mask, dtype=np.uint8)
            mask = _p(self._mask, C.c_uint8)
        N.check(N.lib().eg_rollout_launch_update(self.h, C.c_uint64(seed & (2**64 - 1)), C.c_uint64(first_episode_index),
                                                 n_episodes, mask, C.c_void_p(d_packet_ptr)), "eg_rollout_launch_update")

    def train_step(self, weights: ActionWeights, seed: int, first_episode_index: int, n_episodes: int, replay_mask=None,
                   noise_seed: int = 0, enable_energy_sales=True, write_yearly=True) -> bool:
        """eg_train_step: the whole single-GPU training step in one library call.  Returns True on a new best strategy."""
        mask = None
        if replay_mask is not None:
            self._mask = np.ascontiguousarray(replay_mask, dtype=np.uint8)
            mask = _p(self._mask, C.c_uint8)
        opts = self._opts(enable_energy_sales, False, write_yearly)
        rc = N.lib().eg_train_step(self.h, weights.h, C.byref(opts), C.c_uint64(seed & (2**64 - 1)), C.c_uint64(first_episode_index),
                                   n_episodes, mask, C.c_uint64(noise_seed & (2**64 - 1)))
        if rc < 0:
            N.check(rc, "eg_train_step")
        return rc == 1

    # ---- device-resident policy (include/eirgrid_hip.h: eg_policy_push ... eg_policy_pull) ----
    def push(self, weights: ActionWeights, enable_energy_sales=True, write_yearly=True):
        opts = self._opts(enable_energy_sales, False, write_yearly)
        N.check(N.lib().eg_policy_push(self.h, weights.h, C.byref(opts)), "eg_policy_push")

    def device_rollout(self, seed: int, first_episode_index: int, n_episodes: int, replay_period: int, d_packet_ptr: int):
        N.check(N.lib().eg_device_rollout(self.h, C.c_uint64(seed & (2**64 - 1)), C.c_uint64(first_episode_index), n_episodes,
                                          replay_period, C.c_void_p(d_packet_ptr)), "eg_device_rollout")

    def device_apply(self, d_packets_ptr: int, n_packets: int, d_own_packet_ptr: int, noise_seed: int):
        N.check(N.lib().eg_device_apply(self.h, C.c_void_p(d_packets_ptr), n_packets, C.c_void_p(d_own_packet_ptr),
                                        C.c_uint64(noise_seed & (2**64 - 1))), "eg_device_apply")

    def device_step(self, seed: int, first_episode_index: int, n_episodes: int, replay_period: int, noise_seed: int):
        N.check(N.lib().eg_device_step(self.h, C.c_uint64(seed & (2**64 - 1)), C.c_uint64(first_episode_index), n_episodes,
                                       replay_period, C.c_uint64(noise_seed & (2**64 - 1))), "eg_device_step")

    def hold(self):
        """Keep a device-side copy of the device-resident policy (eg_policy_hold)."""
        N.check(N.lib().eg_policy_hold(self.h), "eg_policy_hold")

    def rewind(self):
        """Put the held copy back (eg_policy_rewind)."""
        N.check(N.lib().eg_policy_rewind(self.h), "eg_policy_rewind")

    def replay_hoist(self, on: bool = True) -> None:
        """Compute the replay episodes of every batch once instead of once per episode (eg_replay_hoist; include/eirgrid_hip.h)."""
        N.check(N.lib().eg_replay_hoist(self.h, int(bool(on))), "eg_replay_hoist")

    def replay_hoist_stats(self):
        """(batches launched with the hoist armed, whether the last of them was served by it).  Synchronises."""
        n, last = C.c_uint64(), C.c_int32()
        N.check(N.lib().eg_replay_hoist_stats(self.h, C.byref(n), C.byref(last)), "eg_replay_hoist_stats")
        return int(n.value), bool(last.value)

    def pull(self, weights: ActionWeights):
        N.check(N.lib().eg_policy_pull(self.h, weights.h), "eg_policy_pull")

    def sync(self):
        N.check(N.lib().eg_sync(self.h), "eg_sync")

    def fetch(self, n_episodes: int = None) -> BatchResult:
        """Every output of the last launched batch (eg_fetch copies eg_last_batch_size() records)."""
        last = int(N.lib().eg_last_batch_size(self.h))
        if n_episodes is not None and n_episodes != last:
            raise ValueError(f"fetch({n_episodes}): the last launched batch holds {last} episodes")
        res = BatchResult.alloc(last)
        out = res.struct()
        N.check(N.lib().eg_fetch(self.h, C.byref(out)), "eg_fetch")
        return res

    def timing_reset(self):
        N.check(N.lib().eg_timing_reset(self.h))

    def timing_read(self):
        ms, n = C.c_double(), C.c_int32()
        N.check(N.lib().eg_timing_read(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def memory_report(self):
        """{tables, records, field_pool} in bytes of device memory (eg_memory_report)."""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        N.check(N.lib().eg_memory_report(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return {"tables": a.value, "records": b.value, "field_pool": c.value}

    def timing_read_grids(self):
        """(span ms, sum of the grids' own ms, launches): grids of a batch that ran side by side show span < sum."""
        span, grids, n = C.c_double(), C.c_double(), C.c_int32()
        N.check(N.lib().eg_timing_read_grids(self.h, C.byref(span), C.byref(grids), C.byref(n)))
        return span.value, grids.value, n.value

    def update_stats(self, d_stats_ptr: int):
        """Reduce the last launched batch into an int64[STATS_LEN] DEVICE buffer (e.g. torch tensor .data_ptr())."""
        N.check(N.lib().eg_update_stats(self.h, C.c_void_p(d_stats_ptr)), "eg_update_stats")

    def fetch_record(self, episode: int) -> BatchResult:
        """Every output of one episode of the last batch, as a one-episode BatchResult."""
        res = BatchResult.alloc(1)
        out = res.struct()
        N.check(N.lib().eg_fetch_record(self.h, episode, C.byref(out)), "eg_fetch_record")
        return res

    def fetch_best_run(self):
        """(state, record): the record of the episode that is the policy's best strategy (strategy.rs:19-258), kept by the
        on-device update.  state 0 = no improvement yet, 1 = record valid, 2 = it ran on another rank.  (The run the reference
        exports is another one: fetch_best_result.)"""
        res = BatchResult.alloc(1)
        out = res.struct()
        state = C.c_int32(0)
        N.check(N.lib().eg_fetch_best_run(self.h, C.byref(out), C.byref(state)), "eg_fetch_best_run")
        return state.value, (res if state.value == 1 else None)

    def track_best_result(self, cost_only: bool = False, on: bool = True) -> None:
        """Start the reference's `best_result` fold at None (core/multi_simulation.rs:384, :613-620): every batch launched from
        now on is folded on the device in iteration order (eg_best_result_track)."""
        N.check(N.lib().eg_best_result_track(self.h, 0 if not on else (2 if cost_only else 1)), "eg_best_result_track")

    def fetch_best_result(self):
        """(global index, record) of the run the reference would summarise and export, or (None, None) before any result."""
        res = BatchResult.alloc(1)
        out = res.struct()
        state = C.c_int32(0); index = C.c_int64(-1)
        N.check(N.lib().eg_fetch_best_result(self.h, C.byref(out), C.byref(state), C.byref(index)), "eg_fetch_best_result")
        return (int(index.value), res) if state.value == 1 else (None, None)

    def fetch_scores(self, n_episodes: int) -> np.ndarray:
        s = np.zeros(n_episodes)
        N.check(N.lib().eg_fetch_scores(self.h, _p(s, C.c_double)), "eg_fetch_scores")
        return s

    def fetch_episode_lists(self, episode: int):
        m = np.zeros(4); nr = np.zeros(N.YEARS, np.int32); nd = np.zeros(N.YEARS, np.int32)
        rl = np.zeros(N.RUN_CAP, np.uint8); dl = np.zeros(N.DEF_CAP, np.uint8)
        N.check(N.lib().eg_fetch_episode_lists(self.h, episode, _p(m, C.c_double), _p(nr, C.c_int32), _p(rl, C.c_uint8),
                                               _p(nd, C.c_int32), _p(dl, C.c_uint8)), "eg_fetch_episode_lists")
        return m, nr, rl, nd, dl

    def debug_fill_lds(self, value: int) -> None:
        """Test hook: leave `value` in every LDS word of every CU (LDS is not cleared between workgroups)."""
        N.check(N.lib().eg_debug_fill_lds(self.h, C.c_uint32(value & 0xFFFFFFFF)), "eg_debug_fill_lds")

    def find_suitable_location_xy(self, gen_type: int, generators_xy=(), size_penalty: float = 1.0, year_index: int = 0):
        """MetalLocationSearch::find_suitable_location with the reference's signature (gpu/metal_location_search.rs:96-103):
        further generators at arbitrary coordinates, an f32 size penalty.  Returns ((x, y) or None, best score)."""
        gx = np.ascontiguousarray([p[0] for p in generators_xy], dtype=np.float64)
        gy = np.ascontiguousarray([p[1] for p in generators_xy], dtype=np.float64)
        x, y, score, found = C.c_double(), C.c_double(), C.c_double(), C.c_int32()
        N.check(N.lib().eg_find_suitable_location(self.h, year_index, gen_type, _p(gx, C.c_double) if len(gx) else None,
                                                  _p(gy, C.c_double) if len(gy) else None, len(gx), C.c_float(size_penalty),
                                                  C.byref(x), C.byref(y), C.byref(found), C.byref(score)), "eg_find_suitable_location")
        return ((x.value, y.value) if found.value else None), score.value

    def find_suitable_location(self, gen_type: int, year_index: int = 0, extra_cells=()):
        """The same search as the rollout kernels run it (generators on the 1 km grid): returns (cell or -1, best score)."""
        cells = np.ascontiguousarray(list(extra_cells), dtype=np.uint16)
        cell, score = C.c_int32(), C.c_double()
        N.check(N.lib().eg_place(self.h, gen_type, year_index, _p(cells, C.c_uint16) if len(cells) else None, len(cells),
                                 C.byref(cell), C.byref(score)), "eg_place")
        return cell.value, score.value
