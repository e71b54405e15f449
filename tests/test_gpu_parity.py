"""GPU: the HIP rollout path, called through the C ABI, against the CPU oracle on the same seeded inputs.
Bar: bit-exact action indices/counts/placement cells AND bit-exact floats (which implies the north star's 1e-5)."""
import json
import os

import numpy as np
import pytest

from eirgrid_amd.engine import ActionWeights, Engine, HostTables
from eirgrid_amd.world import World, synthetic_world
from oracle import api as O
from tests.helpers import assert_episode_equal, oracle_weights_like

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _tabled(world):
    return O.OracleTables(HostTables(world), len(world.existing_x))


def test_native_library_is_loaded(engine):
    """The HIP extension in-tree is the thing that ran (no silent fallback)."""
    maps = open("/proc/self/maps").read()
    assert "libeirgrid_hip.so" in maps


def test_golden_episode_config1(engine):
    """BASELINE config 1 against the committed fixture (made by the literal CPU oracle)."""
    g = json.load(open(os.path.join(GOLDEN, "episode_v1.json")))
    res = engine.run_iteration(0, ActionWeights(), False, 12345)
    assert res.status[0] == 0
    assert [float.hex(v) for v in res.metrics[0]] == g["metrics"]
    assert [[float.hex(v) for v in row] for row in res.yearly[0]] == g["yearly"]
    assert res.lists(0, "run") == g["run"] and res.lists(0, "def") == g["deficit"] and res.lists(0, "act") == g["actions"]
    assert res.gen_cell[0, :res.n_gens[0]].tolist() == g["gen_cell"]
    assert int(res.n_draws[0]) == g["n_draws"]


def test_batch_vs_literal_oracle(engine, oracle_world):
    """16 episodes against the literal oracle (full 100x100 grid search, G x S opinion loops)."""
    res = engine.rollout_batch(ActionWeights(), 12345, 16)
    for e in range(16):
        st, ref = O.run_episode(oracle_world, O.OracleWeights(), 12345 + e)
        assert_episode_equal(res, e, ref, "literal")


def test_config2_1024_episodes_vs_tabled_oracle(engine, world):
    """BASELINE config 2: 1,024 parallel episodes, every one compared bit for bit."""
    tb = _tabled(world)
    res = engine.rollout_batch(ActionWeights(), 12345, 1024)
    assert (res.status == 0).all()
    for e in range(1024):
        st, ref = O.run_episode_tabled(tb, O.OracleWeights(), 12345 + e)
        assert_episode_equal(res, e, ref, "config2")
    # eg_fetch_record: one episode's record on its own
    one = engine.fetch_record(777)
    for name in ("metrics", "yearly", "status", "n_run", "n_def", "n_act", "run_log", "def_log", "n_gens", "gen_cell", "gen_pack",
                 "n_offsets", "off_pack", "n_draws", "bytes_moved"):
        assert getattr(one, name)[0].tobytes() == getattr(res, name)[777].tobytes(), name
    with pytest.raises(Exception):
        engine.fetch_record(1024)


def test_replay_mask_and_best_lists(engine, world):
    """Config-3 style batch: 10 % of the episodes replay the best strategy (Q15 double recording included)."""
    tb = _tabled(world)
    pol = ActionWeights()
    first = engine.run_iteration(0, pol, False, 12345)
    pol.apply_episode(first.metrics[0], first.n_run[0], first.run_log[0, :first.n_run[0].sum()], first.n_def[0],
                      first.def_log[0, :first.n_def[0].sum()])
    assert pol.get("has_best") == 1 and pol.get("iterations_without_improvement") == 0
    n = 256
    mask = (np.arange(n) % 10 == 3).astype(np.uint8)
    res = engine.rollout_batch(pol, 777, n, first_episode_index=5000, replay_mask=mask)
    assert (res.status == 0).all()
    for e in range(n):
        ow = oracle_weights_like(pol)
        st, ref = O.run_episode_tabled(tb, ow, 777 + 5000 + e, replay=bool(mask[e]))
        assert_episode_equal(res, e, ref, "replay" if mask[e] else "sampled")
    e = int(np.argmax(mask))
    assert res.n_run[e].sum() > first.n_run[0].sum()          # every replayed additional action is recorded twice
    assert res.n_draws[e] == 0 or res.n_draws[e] < res.n_draws[(e + 1) % n]   # replay takes no seeded draws unless it falls back


def test_snapshot_variants(engine, world):
    """stall in (100, 500] (scaled exploration), expensive net-zero best (DoNothing boost), missing count table."""
    tb = _tabled(world)
    variants = []
    p = ActionWeights(); p.set("iterations_without_improvement", 250); variants.append(p)
    for stall in (501, 900, 1500, 3500):          # power-scaled sampling (sampling.rs:190-220) via the shared eg_detpow
        p = ActionWeights(); p.set("iterations_without_improvement", stall); variants.append(p)
    p = ActionWeights(); p.set("has_best", 1); p.set("best_net_emissions", -10.0); p.set("best_opinion", 0.7)
    p.set("best_cost", 9e11); p.set("best_reliability", 1.0); variants.append(p)
    p = ActionWeights(); p.set("has_count_weights", 0); variants.append(p)
    p = ActionWeights(); p.set("learning_rate", 0.35); p.set("exploration_rate", 0.6); variants.append(p)
    rng = np.random.default_rng(3)
    p = ActionWeights(); w, dw, cw = p.tables()
    p.set_tables(np.clip(w * rng.uniform(0.2, 5, w.shape), 1e-4, 0.999), np.clip(dw * rng.uniform(0.2, 5, dw.shape), 1e-4, 0.999),
                 cw * rng.uniform(0.5, 2, cw.shape)); variants.append(p)
    for k, pol in enumerate(variants):
        res = engine.rollout_batch(pol, 4242 + k, 48)
        for e in range(48):
            st, ref = O.run_episode_tabled(tb, oracle_weights_like(pol), 4242 + k + e)
            assert_episode_equal(res, e, ref, f"variant {k}")


def test_stalled_sampler_against_the_oracle_with_libm_pow(engine, world):
    """stall > 500: the reference's sampler raises the sorted weights with f64::powf (sampling.rs:199-213).  The kernels evaluate the
    shared eg_detpow (IEEE basic operations only) — and so does the oracle by default, which made the bitwise claim for this branch a
    comparison of that function with itself.  Here the oracle calls libm's pow, as the reference does (og_set_libm_pow): the
    product's episodes at stall 501 / 900 / 1500 / 3500 must still be the oracle's, every action index and every float.  (A pick can
    only differ when a draw lands within 2e-14 of a boundary of the powered table; none does in these 4 x 48 episodes.)"""
    tb = _tabled(world)
    for k, stall in enumerate((501, 900, 1500, 3500)):
        pol = ActionWeights(); pol.set("iterations_without_improvement", stall)
        res = engine.rollout_batch(pol, 4242 + k, 48)
        with O.libm_pow():
            for e in range(48):
                st, ref = O.run_episode_tabled(tb, oracle_weights_like(pol), 4242 + k + e)
                assert_episode_equal(res, e, ref, f"stall {stall}, oracle with libm pow")
        assert O.lib().og_get_libm_pow() == 0


def test_energy_sales_off(engine, world):
    tb = _tabled(world)
    res = engine.rollout_batch(ActionWeights(), 99, 8, enable_energy_sales=False)
    for e in range(8):
        st, ref = O.run_episode_tabled(tb, O.OracleWeights(), 99 + e, energy_sales=False)
        assert_episode_equal(res, e, ref, "no sales")
    assert (res.yearly[:, :, 14] == 0).all()


def test_other_worlds(built):
    """existing_operational_at_start (Q1 switch), a tiny world (1 settlement, no plant, no coast), a different size."""
    base = synthetic_world()
    worlds = [synthetic_world(existing_operational_at_start=True),
              World(base.settlement_x[:1], base.settlement_y[:1], np.array([500000], dtype=np.uint32), np.zeros(0), np.zeros(0),
                    np.zeros(0, np.int32), np.zeros(0), np.zeros(0), np.zeros(0)),
              synthetic_world(seed=99, n_settlements=37, n_coast=16)]
    for k, w in enumerate(worlds):
        eng = Engine(w)
        ow = O.OracleWorld(w)
        res = eng.rollout_batch(ActionWeights(), 31 + k, 6)
        for e in range(6):
            st, ref = O.run_episode(ow, O.OracleWeights(), 31 + k + e)
            assert_episode_equal(res, e, ref, f"world {k}")
        eng.close()


def test_place_kernel_vs_oracle(engine, oracle_world):
    """B2 seam: arg-max kernel vs metal_location_search.rs:110-176 for every type, with random extra plant."""
    rng = np.random.default_rng(1)
    for t in range(15):
        for n_extra in (0, 1, 17, 120):
            cells = rng.integers(0, 2601, n_extra).tolist()
            yi = int(rng.integers(0, 26))
            cell, score = engine.find_suitable_location(t, yi, cells)
            ref_cell, ref_score = oracle_world.place(yi, t, [((c // 51) * 1000.0, (c % 51) * 1000.0) for c in cells])
            assert (cell, score) == (ref_cell, ref_score), (t, n_extra, yi)


def test_shard_invariance_and_determinism(engine):
    """Episodes are keyed by global index: one batch of 64 == batches of 1 + 31 + 32; a batch of 1 == its episode."""
    pol = ActionWeights()
    whole = engine.rollout_batch(pol, 2024, 64)
    parts = [engine.rollout_batch(pol, 2024, n, first_episode_index=f) for f, n in ((0, 1), (1, 31), (32, 32))]
    for name in ("metrics", "yearly", "n_run", "n_def", "n_act", "n_gens", "n_draws", "status"):
        joined = np.concatenate([getattr(p, name) for p in parts])
        assert joined.tobytes() == getattr(whole, name).tobytes(), name
    e = 0
    for p in parts:      # log bytes beyond the per-year counts are unspecified; compare the lists
        for k in range(len(p.status)):
            for which in ("run", "def", "act"):
                assert p.lists(k, which) == whole.lists(e, which)
            assert p.gen_cell[k, :p.n_gens[k]].tolist() == whole.gen_cell[e, :whole.n_gens[e]].tolist()
            e += 1
    again = engine.rollout_batch(pol, 2024, 64)
    assert again.yearly.tobytes() == whole.yearly.tobytes() and again.n_run.tobytes() == whole.n_run.tobytes()
    assert all(again.lists(k, "run") == whole.lists(k, "run") for k in range(64))


def test_full_size_batch_properties(engine, world):
    """BASELINE config 3 size (16,384 episodes, 10 % replay): size-independent properties + sampled bit parity."""
    tb = _tabled(world)
    pol = ActionWeights()
    first = engine.run_iteration(0, pol, False, 12345)
    pol.apply_episode(first.metrics[0], first.n_run[0], first.run_log[0, :first.n_run[0].sum()], first.n_def[0],
                      first.def_log[0, :first.n_def[0].sum()])
    n = 16384
    mask = (np.arange(n) % 10 == 0).astype(np.uint8)
    res = engine.rollout_batch(pol, 12345, n, replay_mask=mask)
    assert (res.status == 0).all()
    y = res.yearly
    assert (y[:, :, 0] == np.arange(2025, 2051)).all()
    assert (y[:, :, 4] >= 0).all()                                  # deficit loop always closes the gap
    assert (y[:, :, 11] == y[:, :, 9] - y[:, :, 10]).all()          # net = co2 - offsets, exactly
    assert np.allclose(np.cumsum(y[:, :, 19], axis=1), y[:, :, 20], rtol=1e-12)
    assert (res.metrics[:, 0] == y[:, 25, 11]).all() and (res.metrics[:, 2] == y[:, 25, 7]).all()
    assert (res.n_run.sum(1) >= res.n_def.sum(1)).all()
    samp = ~mask.astype(bool)
    assert (res.n_act[samp] + res.n_def[samp] <= np.maximum(20, res.n_def[samp])).all()   # 20-action cap per year
    assert (res.n_gens == (np.take_along_axis(res.gen_pack, np.zeros((n, 1), int), 1)[:, 0] * 0 + res.n_gens)).all()
    # all replay episodes follow the same script and take no seeded draws -> identical
    rep = np.flatnonzero(mask)
    assert (res.yearly[rep] == res.yearly[rep[0]]).all()
    for e in list(range(0, n, 997)) + [n - 1]:
        st, ref = O.run_episode_tabled(tb, oracle_weights_like(pol), 12345 + e, replay=bool(mask[e]))
        assert_episode_equal(res, e, ref, "config3")


def test_unsupported_inputs_fail_loudly(engine):
    from eirgrid_amd import _native as N
    pol = ActionWeights()
    with pytest.raises(N.EirgridError):
        engine.rollout_batch(pol, 1, 4, enable_construction_delays=True)


def test_fuzzed_snapshots_and_replay_fallbacks(engine, world):
    """Randomised policies: perturbed tables, random stall / rates, and replay against RANDOM best lists (any of the 61
    actions, wrong lengths) so that non-generator deficit replays, the 20-action overshoot of Q15 and the smart fallbacks
    (sampling.rs:445-528, next_u32 draws) are exercised.  Every episode bit-identical to the tabled oracle."""
    tb = _tabled(world)
    rng = np.random.default_rng(2024)
    for trial in range(6):
        pol = ActionWeights()
        # a synthetic "best strategy": random lists, sometimes empty years, sometimes long
        run = [rng.integers(0, 61, int(rng.choice([0, 0, 1, 2, 5, 12]))).tolist() for _ in range(26)]
        dfl = [(3 * rng.choice([8, 7, 12, 11, 9, 0, 1, 4, 10, 5, 2, 3, 13, 14], int(rng.choice([0, 1, 2, 3])))).tolist() for _ in range(26)]
        if trial >= 2:
            dfl[0] = rng.integers(0, 61, 3).tolist()          # includes non-generator actions: skipped by the repair loop
        nr = np.array([len(l) for l in run], np.int32); nd = np.array([len(l) for l in dfl], np.int32)
        metrics = [float(rng.choice([-5e4, 3e5])), 0.7, float(rng.choice([4e10, 9e11])), 1.0]
        pol.apply_episode(metrics, nr, np.array([a for l in run for a in l], np.uint8), nd, np.array([a for l in dfl for a in l], np.uint8))
        assert pol.get("has_best_actions") == 1
        w, dw, cw = pol.tables()
        pol.set_tables(np.clip(w * 10 ** rng.uniform(-1.5, 1.0, w.shape), 1e-4, 0.999),
                       np.clip(dw * 10 ** rng.uniform(-1.5, 1.0, dw.shape), 1e-4, 0.999), cw * rng.uniform(0.2, 3.0, cw.shape))
        pol.set("iterations_without_improvement", int(rng.choice([0, 50, 150, 480, 520, 1400, 4000])))
        pol.set("learning_rate", float(rng.uniform(0.05, 0.5))); pol.set("exploration_rate", float(rng.uniform(0.0, 0.9)))
        if trial % 2 == 1:
            pol.set("has_count_weights", 0)
        n = 96
        mask = (rng.uniform(size=n) < 0.4).astype(np.uint8)
        res = engine.rollout_batch(pol, 31337 + trial, n, first_episode_index=100 * trial, replay_mask=mask)
        for e in range(n):
            st, ref = O.run_episode_tabled(tb, oracle_weights_like(pol), 31337 + trial + 100 * trial + e, replay=bool(mask[e]))
            assert_episode_equal(res, e, ref, f"fuzz {trial}")
        assert (res.n_draws[mask == 1] > 0).any(), "some replay episode must have fallen back to seeded draws"


def test_helper_wave_kernel_equals_single_wave_kernel(world):
    """Small batches run two waves per episode (a helper wave evaluates chunk 1 of every placement search and folds the
    next year's sums); large batches run one.  Both kernels, forced through the same 1,536 episodes (fresh and stalled policy, with
    replays), must agree on every output byte — and a sample of them with the tabled oracle."""
    tb = _tabled(world)
    engines = {}
    for mode in ("0", "all"):
        os.environ["EIRGRID_HELPER_WAVES"] = mode
        try:
            engines[mode] = Engine(world, device=0)
        finally:
            del os.environ["EIRGRID_HELPER_WAVES"]
    try:
        for stall in (0, 1200):
            pol = ActionWeights()
            pol.set("iterations_without_improvement", stall)
            n = 1536
            mask = (np.arange(n) % 7 == 3).astype(np.uint8)
            a = engines["0"].rollout_batch(pol, 2468, n, first_episode_index=77, replay_mask=mask)
            b = engines["all"].rollout_batch(pol, 2468, n, first_episode_index=77, replay_mask=mask)
            assert (a.status == 0).all()
            for name in ("status", "metrics", "yearly", "n_run", "n_def", "n_act", "n_gens", "n_offsets", "n_draws", "bytes_moved"):
                assert getattr(a, name).tobytes() == getattr(b, name).tobytes(), name
            for e in range(n):
                g = int(a.n_gens[e])
                assert a.gen_cell[e, :g].tobytes() == b.gen_cell[e, :g].tobytes() and a.gen_pack[e, :g].tobytes() == b.gen_pack[e, :g].tobytes()
                assert a.lists(e, "run") == b.lists(e, "run") and a.lists(e, "def") == b.lists(e, "def") and a.lists(e, "act") == b.lists(e, "act")
            for e in range(0, n, 97):
                st, ref = O.run_episode_tabled(tb, oracle_weights_like(pol), 2468 + 77 + e, replay=bool(mask[e]))
                assert_episode_equal(b, e, ref, f"helper kernel, stall {stall}")
    finally:
        for eng in engines.values():
            eng.close()


def test_rollout_ignores_what_is_left_in_lds(engine, world):
    """LDS is handed from workgroup to workgroup uncleared.  The helper protocol polls sequence flags in LDS, so a stale
    word equal to a live sequence number would read as "result ready" (seen once as a garbage year-1 opinion after a
    kernel that leaves small integers in LDS).  Fill every CU's LDS with each value a flag takes early in an episode,
    then roll out: every output byte must equal the clean run's."""
    pol = ActionWeights()
    n = 1024                                         # the two-wave kernel
    clean = engine.rollout_batch(pol, 97531, n)
    tb = _tabled(world)
    for e in range(0, n, 101):
        st, ref = O.run_episode_tabled(tb, O.OracleWeights(), 97531 + e)
        assert_episode_equal(clean, e, ref, "clean run")
    names = ("status", "metrics", "yearly", "n_run", "n_def", "n_act", "n_gens", "n_offsets", "n_draws", "run_log", "def_log", "act_log", "gen_cell")
    for value in list(range(1, 25)) + [0xFFFFFFFF, 0x3FF00000]:
        engine.debug_fill_lds(value)
        res = engine.rollout_batch(pol, 97531, n)
        for name in names:
            assert getattr(res, name).tobytes() == getattr(clean, name).tobytes(), (value, name)
    engine.debug_fill_lds(3)
    big = engine.rollout_batch(pol, 97531, 4096)     # the one-wave kernel on the same episodes
    for name in ("status", "metrics", "yearly", "n_run", "n_def", "n_act", "n_gens", "n_draws"):
        assert getattr(big, name)[:n].tobytes() == getattr(clean, name).tobytes(), name


def test_both_kernels_agree_on_random_policies(world):
    """A short version of scripts/soak.py inside the suite: random policies (perturbed tables, stall counters, best lists,
    20 % replays) through both kernels, identical bytes demanded.  Seed 1 is the stream on which a list-append race of
    the small-batch kernel (the helper wave still reading the generator list when the next generator was appended)
    first showed, in its fourth policy."""
    engines = {}
    for mode in ("0", "all"):
        os.environ["EIRGRID_HELPER_WAVES"] = mode
        try:
            engines[mode] = Engine(world, device=0)
        finally:
            del os.environ["EIRGRID_HELPER_WAVES"]
    rng = np.random.default_rng(1)
    try:
        for trial in range(8):
            pol = ActionWeights()
            run = [rng.integers(0, 61, int(rng.choice([0, 0, 1, 2, 5, 9]))).tolist() for _ in range(26)]
            dfl = [(3 * rng.choice([8, 7, 12, 11, 9, 0, 1, 4, 10, 5, 2, 3, 13, 14], int(rng.choice([0, 1, 2, 3])))).tolist() for _ in range(26)]
            nr = np.array([len(l) for l in run], np.int32); nd = np.array([len(l) for l in dfl], np.int32)
            pol.apply_episode([float(rng.choice([-5e4, 3e5])), 0.7, float(rng.choice([4e10, 9e11])), 1.0], nr,
                              np.array([a for l in run for a in l], np.uint8), nd, np.array([a for l in dfl for a in l], np.uint8))
            w, dw, cw = pol.tables()
            pol.set_tables(np.clip(w * 10 ** rng.uniform(-1.5, 1.0, w.shape), 1e-4, 0.999),
                           np.clip(dw * 10 ** rng.uniform(-1.5, 1.0, dw.shape), 1e-4, 0.999), cw * rng.uniform(0.2, 3.0, cw.shape))
            pol.set("iterations_without_improvement", int(rng.choice([0, 50, 150, 480, 520, 1400, 4000])))
            pol.set("learning_rate", float(rng.uniform(0.05, 0.5))); pol.set("exploration_rate", float(rng.uniform(0.0, 0.9)))
            if trial % 3 == 2:
                pol.set("has_count_weights", 0)
            n = 1536; seed = int(rng.integers(1, 2**40)); first = int(rng.integers(0, 2**20))
            mask = (rng.uniform(size=n) < 0.2).astype(np.uint8)
            a = engines["0"].rollout_batch(pol, seed, n, first_episode_index=first, replay_mask=mask)
            b = engines["all"].rollout_batch(pol, seed, n, first_episode_index=first, replay_mask=mask)
            for name in ("status", "metrics", "yearly", "n_run", "n_def", "n_act", "n_gens", "n_offsets", "n_draws", "bytes_moved"):
                assert getattr(a, name).tobytes() == getattr(b, name).tobytes(), (trial, name)
            for e in range(n):
                g = int(a.n_gens[e])
                assert a.gen_cell[e, :g].tobytes() == b.gen_cell[e, :g].tobytes(), (trial, e)
    finally:
        for eng in engines.values():
            eng.close()


def _engines_by_helper_mode(world, slots=None):
    engines = {}
    for mode in ("0", "all"):
        os.environ["EIRGRID_HELPER_WAVES"] = mode
        if slots is not None:
            os.environ["EIRGRID_HEAVY_SLOTS"] = slots
        try:
            engines[mode] = Engine(world, device=0)
        finally:
            del os.environ["EIRGRID_HELPER_WAVES"]
            os.environ.pop("EIRGRID_HEAVY_SLOTS", None)
    return engines


def _policy_with_best_lists(rng, per_year, types, offsets_per_year=0, deficit_max=3):
    """An ActionWeights whose best strategy adds `per_year` generators (of `types`) and `offsets_per_year` carbon offsets in
    every year, in random order, plus a few deficit actions."""
    pol = ActionWeights()
    run = []
    for _ in range(26):
        year = [int(3 * rng.choice(types) + rng.integers(0, 3)) for _ in range(per_year)]
        year += [int(45 + rng.integers(0, 12)) for _ in range(offsets_per_year)]
        rng.shuffle(year)
        run.append([int(a) for a in year])
    dfl = [[int(3 * rng.choice([8, 7, 12, 11])) for _ in range(int(rng.integers(0, deficit_max)))] for _ in range(26)]
    nr = np.array([len(l) for l in run], np.int32); nd = np.array([len(l) for l in dfl], np.int32)
    pol.apply_episode([-5e4, 0.7, 4e10, 1.0], nr, np.array([a for l in run for a in l], np.uint8), nd,
                      np.array([a for l in dfl for a in l], np.uint8))
    return pol


_ALL_FIELDS = ("status", "metrics", "yearly", "n_run", "n_def", "n_act", "n_gens", "n_offsets", "n_draws", "bytes_moved",
               "run_log", "def_log", "act_log", "gen_cell", "gen_pack", "off_pack")


def _used(res, name):
    """A list field with everything behind each episode's used length zeroed (the records are not cleared between batches: what
    lies behind a list is whatever an earlier batch of that engine left there)."""
    a = getattr(res, name)
    if a.ndim != 2 or name in ("metrics", "n_run", "n_def", "n_act"):
        return a
    length = {"run_log": res.n_run.sum(axis=1), "def_log": res.n_def.sum(axis=1), "act_log": res.n_act.sum(axis=1),
              "gen_cell": res.n_gens, "gen_pack": res.n_gens, "off_pack": res.n_offsets}[name]
    return np.where(np.arange(a.shape[1])[None, :] < length[:, None], a, 0).astype(a.dtype)


def test_capacity_overflow_is_reported_not_hidden(world):
    """Maximum sizes.  The record capacities are the oracle's (4096 entries per list, include/eirgrid_hip.h): a best strategy
    with 80 generator additions in every year, replayed (and double-recorded), would record 4160 actions.  Such an
    episode must end with EG_EP_OVERFLOW in both kernels AND in the oracle, with identical bytes between the kernels, never
    with a silently truncated result; seeded episodes of the same batch are untouched."""
    engines = _engines_by_helper_mode(world)
    try:
        pol = ActionWeights()
        nr = np.full(26, 80, np.int32); nd = np.zeros(26, np.int32)
        pol.apply_episode([-5e4, 0.7, 4e10, 1.0], nr, np.full(26 * 80, 36, np.uint8), nd, np.zeros(0, np.uint8))   # 36 = BatteryStorage 100 %
        n = 64
        mask = (np.arange(n) % 2 == 0).astype(np.uint8)
        a = engines["0"].rollout_batch(pol, 5, n, replay_mask=mask)
        b = engines["all"].rollout_batch(pol, 5, n, replay_mask=mask)
        assert (a.status[mask == 1] == -1).all(), "EG_EP_OVERFLOW expected for every replay episode"     # include/eirgrid_hip.h
        assert (a.status[mask == 0] == 0).all()
        for name in ("status", "metrics", "n_run", "n_def", "n_act", "n_gens", "n_offsets", "n_draws"):
            assert getattr(a, name).tobytes() == getattr(b, name).tobytes(), name
        assert (a.n_gens <= 4096).all() and (a.n_run.sum(axis=1) <= 4096).all()
        tb = _tabled(world)
        st, ref = O.run_episode_tabled(tb, oracle_weights_like(pol), 5, replay=True)
        assert st == -1, "the oracle overflows on the same episode"
        for e in (1, 33, 63):      # the seeded episodes of the batch against the oracle
            st, ref = O.run_episode_tabled(tb, oracle_weights_like(pol), 5 + e, replay=False)
            assert_episode_equal(a, e, ref, "beside overflowing episodes")
    finally:
        for eng in engines.values():
            eng.close()


def test_long_replay_episodes_beyond_the_onchip_window_match_the_oracle(world):
    """The reference's lists are Vecs (core/simulation.rs:146-162, :406-409; sampling.rs:93-101, :258-266) and a replay doubles
    them: configs[3] as a free-running loop reaches a 965-action best list by its eighth update.  Replay episodes of
    ~600, ~1000 and ~2000 generators (and ~700 carbon offsets): beyond the 512 entries the kernels keep in LDS the long-replay
    variant goes on in the episode's own record (eg_rollout.hip ListTail).  Every output bit-identical to the tabled oracle, in
    both helper modes, with a field pool for every episode, for three of them (the others take place_exact_long) and none."""
    tb = _tabled(world)
    rng = np.random.default_rng(2026)
    policies = [_policy_with_best_lists(rng, 23, [0, 4, 12, 7, 5], offsets_per_year=2),          # ~600 generators, three radius classes
                _policy_with_best_lists(rng, 12, [0, 1, 13], offsets_per_year=27),               # ~310 generators, ~700 offsets
                _policy_with_best_lists(rng, 39, list(range(15)), offsets_per_year=1),           # ~1000 generators, every class
                _policy_with_best_lists(rng, 76, [12, 2, 3, 8, 9, 4], offsets_per_year=0, deficit_max=2)]   # ~2000 generators, 3 km class
    n = 24
    mask = (np.arange(n) % 3 != 1).astype(np.uint8)
    results = {}
    for slots, which in ((None, range(4)), ("3", (0, 2)), ("0", (0,))):
        engines = _engines_by_helper_mode(world, slots)
        try:
            for mode, eng in engines.items():
                results[(mode, slots)] = {k: eng.rollout_batch(policies[k], 77 + k, n, first_episode_index=100 * k, replay_mask=mask) for k in which}
        finally:
            for eng in engines.values():
                eng.close()
    ref_runs = results[("0", None)]
    sizes = []
    for k, pol in enumerate(policies):
        a = ref_runs[k]
        assert (a.status == 0).all(), (k, a.status.tolist())
        sizes.append((int(a.n_gens[mask == 1].min()), int(a.n_gens.max()), int(a.n_offsets.max()), int(a.n_run.sum(axis=1).max())))
        for key, runs in results.items():
            if k not in runs:
                continue
            for name in _ALL_FIELDS:
                assert _used(a, name).tobytes() == _used(runs[k], name).tobytes(), (k, key, name)
        for e in (0, 1, 2, 9, 23):
            st, ref = O.run_episode_tabled(tb, oracle_weights_like(pol), 77 + k + 100 * k + e, replay=bool(mask[e]))
            assert_episode_equal(a, e, ref, f"policy {k}, {'replay' if mask[e] else 'sampled'}")
    print("generators (min replay, max) / offsets / recorded actions per policy:", sizes)
    assert sizes[0][0] > 550 and sizes[1][2] > 650 and sizes[2][0] > 950 and sizes[3][0] > 1900 and sizes[3][3] > 3900


@pytest.mark.parametrize("variant", [1, 2])
def test_helper_protocol_timeout_is_an_error_not_a_hang(variant):
    """The episode wave polls LDS flags for the helper wave's results.  Negative builds (csrc/Makefile `negative`) make
    the helper withhold a flag — variant 1: the results of every search of 2027, variant 2: the starting sums of 2030.
    The kernel must finish (bounded polling), every episode that needed the withheld result must end with
    EG_EP_INTERNAL, eg_fetch must return EG_ERR_INTERNAL, and the process must stay usable."""
    import subprocess, sys, time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "eirgrid_amd", f"libeirgrid_hip_neg{variant}.so")
    assert os.path.exists(lib), f"{lib} missing: make -C eirgrid_amd/csrc negative"
    code = r"""
import sys, time
sys.path.insert(0, %r)
import numpy as np
from eirgrid_amd import synthetic_world, _native as N
from eirgrid_amd.engine import ActionWeights, Engine
eng = Engine(synthetic_world(), device=0)
pol = ActionWeights()
t0 = time.time()
try:
    eng.rollout_batch(pol, 12345, 256)
    print("RESULT no-error")
except N.EirgridError as e:
    code = "internal" if "code %%d" %% N.EG_ERR_INTERNAL in str(e) else "other"
    eng.sync()
    from eirgrid_amd.engine import BatchResult
    import ctypes as C
    st = np.zeros(256, np.int32)
    out = N.EgEpisodeOut(); out.status = st.ctypes.data_as(C.POINTER(C.c_int32))
    N.lib().eg_fetch(eng.h, C.byref(out))
    print("RESULT", code, int((st == N.EG_EP_INTERNAL).sum()), int((st == 0).sum()), "%%.1f" %% (time.time() - t0))
big = eng.rollout_batch(pol, 12345, 2048)      # the one-wave kernel has no helper: unaffected
print("BIG", int((big.status == 0).sum()))
eng.close()
""" % root
    env = dict(os.environ, EIRGRID_LIB=lib)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    res = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][0].split()
    assert res[1] == "internal", r.stdout
    n_internal, n_ok, seconds = int(res[2]), int(res[3]), float(res[4])
    assert n_internal + n_ok == 256 and n_internal >= 50 and seconds < 60
    assert [l for l in r.stdout.splitlines() if l.startswith("BIG")][0].split()[1] == "2048"


def test_heavy_episodes_match_the_oracle(world):
    """Replay episodes with hundreds of generators (what the reference's replay phase grows into).  From 64
    generators on a search runs through the approximate penalty field + exact evaluation of the few candidates
    (eg_rollout.hip place_heavy) instead of the exact branch-and-bound scan.  Every output must stay bit-identical: to the
    tabled oracle, between the two kernels, and between a pool of field slots that covers every heavy episode, one that
    runs out after three (the rest fall back to the exact scan) and no pool at all."""
    tb = _tabled(world)
    rng = np.random.default_rng(11)
    policies = []
    # (the field update comes in four lengths, by the entries of the radius classes an episode searches — heavy_pack_list: the type
    #  lists below make it 4, 1, 4, 1, 2 and 3 quads of the entry list)
    for per_year, types in ((12, list(range(15))), (9, [0, 4, 12]), (15, [1, 13, 14, 5, 7]), (17, [12]), (9, [5]), (10, [5, 6, 0])):
        pol = ActionWeights()
        run = [[int(3 * rng.choice(types) + rng.integers(0, 3)) for _ in range(per_year)] for _ in range(26)]
        dfl = [[int(3 * rng.choice([8, 7, 12, 11])) for _ in range(int(rng.integers(0, 3)))] for _ in range(26)]
        nr = np.array([len(l) for l in run], np.int32); nd = np.array([len(l) for l in dfl], np.int32)
        pol.apply_episode([-5e4, 0.7, 4e10, 1.0], nr, np.array([a for l in run for a in l], np.uint8), nd,
                          np.array([a for l in dfl for a in l], np.uint8))
        policies.append(pol)
    n = 96
    mask = (np.arange(n) % 3 != 1).astype(np.uint8)
    results, kernel_ms = {}, {}
    for mode, slots in (("0", None), ("all", None), ("0", "3"), ("0", "0"), ("all", "0")):
        os.environ["EIRGRID_HELPER_WAVES"] = mode
        if slots is not None:
            os.environ["EIRGRID_HEAVY_SLOTS"] = slots
        try:
            eng = Engine(world, device=0)
        finally:
            del os.environ["EIRGRID_HELPER_WAVES"]
            os.environ.pop("EIRGRID_HEAVY_SLOTS", None)
        try:
            eng.timing_reset()
            results[(mode, slots)] = [eng.rollout_batch(pol, 321 + k, n, first_episode_index=40 * k, replay_mask=mask) for k, pol in enumerate(policies)]
            kernel_ms[(mode, slots)] = eng.timing_read()[0]
        finally:
            eng.close()
    ref_runs = results[("0", None)]
    heavy = 0
    for k, pol in enumerate(policies):
        a = ref_runs[k]
        assert (a.status == 0).all(), (k, np.bincount(-a.status))
        heavy += int((a.n_gens >= 200).sum())
        for key, runs in results.items():
            b = runs[k]
            for name in ("status", "metrics", "yearly", "n_run", "n_def", "n_act", "n_gens", "n_offsets", "n_draws", "bytes_moved",
                         "run_log", "def_log", "act_log", "gen_cell", "gen_pack", "off_pack"):
                assert getattr(a, name).tobytes() == getattr(b, name).tobytes(), (k, key, name)
        for e in list(range(0, n, 7)) + [2, 5]:
            st, ref = O.run_episode_tabled(tb, oracle_weights_like(pol), 321 + k + 40 * k + e, replay=bool(mask[e]))
            assert_episode_equal(a, e, ref, f"policy {k}, {'replay' if mask[e] else 'sampled'}")
    assert heavy >= 100
    # the field path really ran: the same episodes take a fraction of the exact scan's time
    print("k_rollout ms for the six batches:", {f"{k[0]}/{k[1]}": round(v, 2) for k, v in kernel_ms.items()})
    assert kernel_ms[("0", None)] < 0.5 * kernel_ms[("0", "0")] and kernel_ms[("all", None)] < 0.5 * kernel_ms[("all", "0")]


def test_fetch_copies_the_lists_as_wide_as_the_batch_needs(engine):
    """eg_rollout_batch copies the counts and, of every list, only the entries the longest list of the batch announces
    (include/eirgrid_hip.h): into a caller's buffer full of a marker byte the entries within an episode's counts are the episode's,
    and nothing behind the batch's longest list is touched."""
    from eirgrid_amd.engine import BatchResult
    n = 48
    ref = engine.rollout_batch(ActionWeights(), 4242, n)
    out = BatchResult.alloc(n)
    for name in ("run_log", "def_log", "act_log", "gen_cell", "gen_pack", "off_pack"):
        getattr(out, name).view(np.uint8)[:] = 0xAB
    res = engine.rollout_batch(ActionWeights(), 4242, n, out=out)
    assert res is out and (res.status == 0).all()
    for name in ("metrics", "yearly", "status", "n_run", "n_def", "n_act", "n_gens", "n_offsets", "n_draws"):
        assert getattr(res, name).tobytes() == getattr(ref, name).tobytes(), name
    widths = {"run_log": res.n_run.sum(axis=1), "def_log": res.n_def.sum(axis=1), "act_log": res.n_act.sum(axis=1),
              "gen_cell": res.n_gens, "gen_pack": res.n_gens, "off_pack": res.n_offsets}
    for name, used in widths.items():
        a, b = getattr(res, name), getattr(ref, name)
        w = int(used.max())
        assert 0 < w < a.shape[1] or name == "off_pack"
        for e in range(n):
            assert (a[e, :used[e]] == b[e, :used[e]]).all(), (name, e)
        assert (a[:, w:].view(np.uint8) == 0xAB).all(), f"{name}: written behind the batch's longest list"


@pytest.mark.gpu
def test_heavy_searches_with_tied_candidates_match_the_oracle():
    """A world that is symmetric about both centre lines and the diagonals (one settlement in the middle of the map, no plant, no
    coast): every placement score is shared by up to eight cells, bit for bit.  A long replay's search then finds several
    candidates within 2^-30 of the largest approximate score — the single-candidate rule (its cell is the answer, no exact
    evaluation) does not apply —, evaluates them exactly and must take the lowest cell of the tied maxima, as the reference's
    first-strictly-greater scan does (metal_location_search.rs:168-171).  Against the tabled oracle, and against an engine
    without the field pool (every search the exact scan)."""
    w = World(np.array([25000.0]), np.array([25000.0]), np.array([400000], dtype=np.uint32), np.zeros(0), np.zeros(0),
              np.zeros(0, np.int32), np.zeros(0), np.zeros(0), np.zeros(0))
    tb = O.OracleTables(HostTables(w), 0)
    rng = np.random.default_rng(5)
    pol = ActionWeights()
    types = [0, 4, 12, 7, 8]
    run = [[int(3 * rng.choice(types) + rng.integers(0, 3)) for _ in range(9)] for _ in range(26)]
    nr = np.array([len(l) for l in run], np.int32)
    pol.apply_episode([-5e4, 0.7, 4e10, 1.0], nr, np.array([a for l in run for a in l], np.uint8), np.zeros(26, np.int32), np.zeros(0, np.uint8))
    n = 24
    mask = np.ones(n, np.uint8)
    results = {}
    for mode, slots in (("0", None), ("all", None), ("0", "0")):
        os.environ["EIRGRID_HELPER_WAVES"] = mode
        if slots is not None:
            os.environ["EIRGRID_HEAVY_SLOTS"] = slots
        try:
            eng = Engine(w, device=0)
        finally:
            del os.environ["EIRGRID_HELPER_WAVES"]
            os.environ.pop("EIRGRID_HEAVY_SLOTS", None)
        try:
            results[(mode, slots)] = eng.rollout_batch(pol, 77, n, replay_mask=mask)
        finally:
            eng.close()
    a = results[("0", None)]
    assert (a.status == 0).all() and a.n_gens.min() >= 200
    for key, b in results.items():
        for name in ("status", "metrics", "yearly", "n_run", "n_def", "n_act", "n_gens", "n_offsets", "n_draws", "run_log", "def_log", "act_log",
                     "gen_cell", "gen_pack", "off_pack"):
            assert getattr(a, name).tobytes() == getattr(b, name).tobytes(), (key, name)
    for e in (0, 7, 23):
        st, ref = O.run_episode_tabled(tb, oracle_weights_like(pol), 77 + e, replay=True)
        assert_episode_equal(a, e, ref, "symmetric world, replay")
    # the ties are real: mirror cells (i, j) / (50 - i, j) / (j, i) ... carry the same unpenalised score, so the early placements
    # of an episode take the lowest cell of a tied set — cells in the first half of the grid
    first = a.gen_cell[0, :4].astype(int)
    assert ((first // 51) <= 25).all(), first


def test_field_pool_is_only_held_once_the_best_list_is_long(world):
    """The penalty-field pool of long replay episodes (126 KB per replay episode of a launch) is allocated when the best list is
    known to be longer than 96 actions, not with the first replay episode: replays of config 1's episode never ask for a slot."""
    eng = Engine(world, device=0)
    try:
        pol = ActionWeights()
        first = eng.rollout_batch(pol, 12345, 1)
        pol.apply_episode(first.metrics[0], first.n_run[0], first.run_log[0], first.n_def[0], first.def_log[0])
        assert sum(len(l) for l in pol.lists(0)) <= 96
        mask = np.ones(64, np.uint8)
        a = eng.rollout_batch(pol, 12345, 64, replay_mask=mask)
        assert (a.status == 0).all() and eng.memory_report()["field_pool"] == 0
        rng = np.random.default_rng(3)
        long_pol = _policy_with_best_lists(rng, 9, [0, 4, 12])
        b = eng.rollout_batch(long_pol, 12345, 64, replay_mask=mask)
        rep = eng.memory_report()
        assert (b.status == 0).all() and rep["field_pool"] == 4096 * 6 * 2624 * 8 and rep["tables"] > 20e6 and rep["records"] >= 64 * 41000
    finally:
        eng.close()


def test_field_pool_grows_with_the_launch(world):
    """More replay episodes in one launch than the field pool starts with (4 096 slots): the pool is enlarged before the launch,
    so none of them falls back to the exact scan (20-40x slower, and a launch lasts as long as its slowest episode), and
    episodes beyond the old pool size are still the oracle's, bit for bit."""
    tb = _tabled(world)
    rng = np.random.default_rng(23)
    pol = ActionWeights()
    run = [[int(3 * rng.choice([0, 4, 12, 7]) + rng.integers(0, 3)) for _ in range(7)] for _ in range(26)]
    nr = np.array([len(l) for l in run], np.int32); nd = np.zeros(26, np.int32)
    pol.apply_episode([-5e4, 0.7, 4e10, 1.0], nr, np.array([a for l in run for a in l], np.uint8), nd, np.zeros(0, np.uint8))
    n = 4608
    mask = np.ones(n, np.uint8)
    out, ms = {}, {}
    for slots in (None, "4096"):      # the default (grows with the launch) and a pool held at its initial size
        if slots is not None:
            os.environ["EIRGRID_HEAVY_SLOTS"] = slots
        try:
            eng = Engine(world, device=0)
        finally:
            os.environ.pop("EIRGRID_HEAVY_SLOTS", None)
        try:
            eng.rollout_batch(pol, 99, 64, replay_mask=mask[:64])      # (the pool exists at its initial size now)
            eng.timing_reset()
            out[slots] = eng.rollout_batch(pol, 99, n, replay_mask=mask)
            t, launches = eng.timing_read()
            ms[slots] = t / max(launches, 1)
        finally:
            eng.close()
    res = out[None]
    assert (res.status == 0).all() and int(res.n_gens.min()) >= 150
    for name in ("status", "metrics", "yearly", "n_run", "n_def", "n_gens", "bytes_moved"):
        assert getattr(res, name).tobytes() == getattr(out["4096"], name).tobytes(), name
    for e in (0, 4095, 4096, 4300, n - 1):
        st, ref = O.run_episode_tabled(tb, oracle_weights_like(pol), 99 + e, replay=True)
        assert_episode_equal(res, e, ref, f"episode {e} of {n} replays")
    print(f"{n} replay episodes of {res.n_gens.mean():.0f} generators: {ms[None]:.2f} ms; with 512 of them on the exact scan: {ms['4096']:.2f} ms")
    assert ms[None] < 0.75 * ms["4096"]


def test_find_suitable_location_with_the_reference_signature(engine, oracle_world):
    """B2 as the reference declares it (metal_location_search.rs:96-103): generators at arbitrary coordinates (off the 1 km
    grid, on it, on top of candidates, outside every radius) and an f32 size penalty — cell AND score bit-identical to the
    literal 100 x 100 search of the oracle, for all 15 types, several years, 0 to 300 generators."""
    rng = np.random.default_rng(5)
    checked = 0
    for trial in range(24):
        n = int(rng.choice([0, 1, 3, 17, 80, 300]))
        xy = np.column_stack([rng.uniform(0.0, 50000.0, n), rng.uniform(0.0, 50000.0, n)])
        if n >= 3:
            xy[0] = (12000.0, 31000.0)                 # exactly on a candidate: its score becomes 0
            xy[1] = np.round(xy[1] / 1000.0) * 1000.0   # on the grid
            xy[2] = (49999.999, 0.001)
        for t in range(15):
            yi = int(rng.integers(0, 26)); sp = float(rng.choice([1.0, 0.0, 0.35, 2.5]))
            got_xy, got_score = engine.find_suitable_location_xy(t, [tuple(p) for p in xy], size_penalty=sp, year_index=yi)
            cell, score = oracle_world.place(yi, t, [tuple(p) for p in xy], size_penalty=sp)
            assert np.float64(got_score).tobytes() == np.float64(score).tobytes(), (trial, t, yi, n, got_score, score)
            assert (got_xy is None) == (cell < 0)
            if cell >= 0:
                assert got_xy == (float(cell // 51) * 1000.0, float(cell % 51) * 1000.0), (trial, t)
            checked += 1
    assert checked == 24 * 15
    # a size penalty of 10 makes every score 0: None, as in the reference
    assert engine.find_suitable_location_xy(0, [], size_penalty=10.0)[0] is None
