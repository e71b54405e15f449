"""CPU: the product's host-built tables (eg_host_tables_*) are validated by running the oracle's tabled mode on them
and demanding bit-identical episodes to the oracle's literal mode (no table, sqrt/div/pow at every use)."""
import numpy as np
import pytest

from eirgrid_amd.engine import HostTables
from eirgrid_amd.world import World, synthetic_world
from oracle import api as O


def _same(a, b):
    return (np.array(a.yearly).tobytes() == np.array(b.yearly).tobytes() and list(a.metrics) == list(b.metrics)
            and bytes(a.run_log) == bytes(b.run_log) and bytes(a.def_log) == bytes(b.def_log) and bytes(a.act_log) == bytes(b.act_log)
            and list(a.gen_cell) == list(b.gen_cell) and list(a.n_run) == list(b.n_run) and a.n_draws == b.n_draws
            and a.n_gens == b.n_gens and a.n_offsets == b.n_offsets and a.status == b.status)


def _check(world, seeds, replay_from=None, tweak=None):
    ow = O.OracleWorld(world)
    ot = O.OracleTables(HostTables(world), len(world.existing_x))
    for seed in seeds:
        wa, wb = O.OracleWeights(), O.OracleWeights()
        for w in (wa, wb):
            if tweak:
                tweak(w)
            if replay_from is not None:
                w.set("has_best", 1); w.set("has_best_actions", 1); w.set("has_best_deficit_actions", 1)
                for y in range(26):
                    w.set_list(0, y, replay_from[0][y]); w.set_list(1, y, replay_from[1][y])
        sa, a = O.run_episode(ow, wa, seed, replay=replay_from is not None)
        sb, b = O.run_episode_tabled(ot, wb, seed, replay=replay_from is not None)
        assert sa == sb == 0
        assert _same(a, b), f"seed {seed}: tabled episode differs from the literal one"
        for x, y in zip(wa.tables(), wb.tables()):
            assert x.tobytes() == y.tobytes(), "in-episode weight nudges differ"
    return a


def test_tables_match_literal_oracle(world):
    _check(world, range(12345, 12345 + 12))


def test_demand_and_online_tables(world, oracle_world):
    ht = HostTables(world)
    for yi in range(26):
        pop, usage = oracle_world.demand(yi)
        assert ht.f64("population")[yi] == pop and ht.f64("usage")[yi] == usage
    assert ht.i32("existing_online").tolist() == oracle_world.existing_online()
    assert ht.i32("reach").tolist() == [11, 7, 4, 6, 5, 2]
    assert ht.f64("size_factor")[0] == 0.9


def test_tables_replay_episode(world, oracle_world):
    """Replay (force_best_actions) of config 1's episode, including the double recording of Q15."""
    wts = O.OracleWeights()
    st, first = O.run_episode(oracle_world, wts, 12345)
    best = (O.split_log(first.run_log, first.n_run), O.split_log(first.def_log, first.n_def))
    out = _check(world, [7, 8], replay_from=best)
    # Q15: the replayed episode records every additional action twice
    assert sum(out.n_run) > sum(first.n_run)


def test_tables_existing_operational_switch(built):
    _check(synthetic_world(existing_operational_at_start=True), [1, 2, 3])


def test_tables_ragged_worlds(built):
    """Edge shapes: a single settlement, no existing plant, no coastline; and a different size."""
    base = synthetic_world()
    tiny = World(base.settlement_x[:1], base.settlement_y[:1], np.array([500000], dtype=np.uint32), np.zeros(0), np.zeros(0),
                 np.zeros(0, np.int32), np.zeros(0), np.zeros(0), np.zeros(0))
    _check(tiny, [11, 12])
    other = synthetic_world(seed=99, n_settlements=37, n_coast=16)
    _check(other, [5, 6])


def test_tables_stalled_and_best_state(world):
    """Snapshot scalars that change sampling / nudges: stall in (100, 500], a net-zero-but-expensive best (no-op boost),
    no count table (checkpoint-loaded weights take the heuristic branch, sampling.rs:423-442)."""
    def tweak(w):
        w.set("stall", 250); w.set("has_best", 1)
        w.set("best_net_emissions", -10.0); w.set("best_opinion", 0.7); w.set("best_cost", 9e11); w.set("best_reliability", 1.0)
    _check(world, [21, 22, 23], tweak=tweak)
    _check(world, [31, 32], tweak=lambda w: w.set_has_count_weights(0))
    for stall in (600, 2500):      # power-scaled sampler
        _check(world, [41, 42], tweak=lambda w, s=stall: w.set("stall", s))


def test_tables_fuzzed_replay_fallbacks(world):
    """Literal vs tabled oracle on replay episodes whose best lists are random (fallback samplers, non-generator repairs)."""
    rng = np.random.default_rng(7)
    ow = O.OracleWorld(world)
    ot = O.OracleTables(HostTables(world), len(world.existing_x))
    for trial in range(4):
        run = [rng.integers(0, 61, int(rng.choice([0, 1, 3, 9]))).tolist() for _ in range(26)]
        dfl = [rng.integers(0, 61, int(rng.choice([0, 1, 2]))).tolist() for _ in range(26)]
        outs = []
        for mode in (0, 1):
            w = O.OracleWeights()
            w.set("has_best", 1); w.set("has_best_actions", 1); w.set("has_best_deficit_actions", 1); w.set("stall", 30 * trial)
            for y in range(26):
                w.set_list(0, y, run[y]); w.set_list(1, y, dfl[y])
            st, out = (O.run_episode(ow, w, 555 + trial, replay=True) if mode == 0 else O.run_episode_tabled(ot, w, 555 + trial, replay=True))
            assert st == 0
            outs.append(out)
        assert _same(*outs), f"trial {trial}"
        assert outs[0].n_draws > 0
