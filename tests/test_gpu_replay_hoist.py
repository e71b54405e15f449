"""GPU: the replay hoist (csrc/eg_replay_coop.h; include/eirgrid_hip.h eg_replay_hoist) against the per-episode path.

Every replay episode of a batch takes its actions from the stored lists (sampling.rs:78-101, :242-266; simulation.rs:146-162) and
reads no seeded draw until a list runs out — the replay episodes of a batch are one computation.  With the hoist on that script is
computed once by a cooperative workgroup and its record handed to every replay episode.  Bar: every output byte of every episode
(n_chunks excepted: it counts what the search that really ran requested), the update packets and the policies of a training loop
identical to the per-episode path's — and the tabled oracle's — whether the hoist takes or has to leave the episodes to the
per-episode kernels (a fallback draw, a capacity)."""
import os

import numpy as np
import pytest

from eirgrid_amd.engine import ActionWeights, Engine, HostTables
from eirgrid_amd.parallel import BatchTrainer
from eirgrid_amd.world import World
from oracle import api as O
from tests.helpers import assert_episode_equal, oracle_weights_like
from tests.test_gpu_parity import _ALL_FIELDS, _policy_with_best_lists, _used

pytestmark = pytest.mark.gpu


def _pair(world, helper=None):
    """(engine without the hoist, engine with it)"""
    if helper is not None:
        os.environ["EIRGRID_HELPER_WAVES"] = helper
    try:
        plain, hoisted = Engine(world, device=0), Engine(world, device=0)
    finally:
        os.environ.pop("EIRGRID_HELPER_WAVES", None)
    hoisted.replay_hoist(True)
    return plain, hoisted


def _same_records(a, b, what):
    for name in _ALL_FIELDS:
        assert _used(a, name).tobytes() == _used(b, name).tobytes(), (what, name)


def _full_script(rng, per_year, types, offsets_per_year=0):
    """A best strategy that a replay can follow to the end without a fallback draw: four deficit actions in every year (the repair
    loop asks the list at most four times a year, simulation.rs:362-377), `per_year` generators and some offsets."""
    pol = ActionWeights()
    run = []
    for _ in range(26):
        year = [int(3 * rng.choice(types) + rng.integers(0, 3)) for _ in range(per_year)] + [int(45 + rng.integers(0, 12)) for _ in range(offsets_per_year)]
        year += [int(rng.choice([57, 58, 59, 60]))] * int(rng.integers(0, 2))
        rng.shuffle(year)
        run.append([int(a) for a in year])
    dfl = [[int(3 * rng.choice([8, 7, 12, 11, 0, 45 // 3 + 1])) for _ in range(4)] for _ in range(26)]      # (48 = an offset: skipped by the repair loop)
    pol.apply_episode([-5e4, 0.7, 4e10, 1.0], np.array([len(l) for l in run], np.int32), np.array([a for l in run for a in l], np.uint8),
                      np.full(26, 4, np.int32), np.array([a for l in dfl for a in l], np.uint8))
    return pol


def _seeded(engine):
    """config 1's episode installed as the best strategy by the reference's own sequential update"""
    pol = ActionWeights()
    first = engine.run_iteration(0, pol, False, 12345)
    pol.apply_episode(first.metrics[0], first.n_run[0], first.run_log[0, :first.n_run[0].sum()], first.n_def[0],
                      first.def_log[0, :first.n_def[0].sum()])
    return pol


def test_hoisted_replays_are_the_per_episode_records(world):
    """configs[2]'s shape — every 10th episode replays — from the seeded policy (a short list: 35 generators per replay) and from
    lists of ~120 and ~600 generators (beyond the per-episode kernels' on-chip window), in the large-batch and the small-batch launch
    shape; each batch against the per-episode path byte for byte, the replay episodes against the tabled oracle, and the hoist must
    really have served them."""
    tb = O.OracleTables(HostTables(world), len(world.existing_x))
    rng = np.random.default_rng(404)
    for helper in ("0", "all"):
        plain, hoisted = _pair(world, helper)
        try:
            policies = [_seeded(plain), _full_script(rng, 5, [0, 4, 12, 7], offsets_per_year=1),
                        _full_script(rng, 23, [0, 4, 12, 7, 5, 1, 13], offsets_per_year=2)]
            for k, pol in enumerate(policies):
                n = 320
                mask = (np.arange(n) % 10 == 3).astype(np.uint8)
                a = plain.rollout_batch(pol, 900 + k, n, first_episode_index=1000 * k, replay_mask=mask)
                b = hoisted.rollout_batch(pol, 900 + k, n, first_episode_index=1000 * k, replay_mask=mask)
                assert (a.status == 0).all()
                _same_records(a, b, (helper, k))
                armed, served = hoisted.replay_hoist_stats()
                assert served, f"policy {k}: the script needs no fallback draw — the hoist must have taken it"
                reps = np.flatnonzero(mask)
                assert (b.n_draws[reps] == 0).all()
                for name in _ALL_FIELDS:      # the replay episodes of a batch are one computation
                    u = _used(b, name)
                    assert all(u[reps[0]].tobytes() == u[e].tobytes() for e in reps), name
                for e in (int(reps[0]), int(reps[-1]), 0):
                    st, ref = O.run_episode_tabled(tb, oracle_weights_like(pol), 900 + k + 1000 * k + e, replay=bool(mask[e]))
                    assert_episode_equal(b, e, ref, f"hoisted, policy {k}")
            print("generators per replay episode:", [int(hoisted.rollout_batch(p, 1, 10, replay_mask=np.ones(10, np.uint8)).n_gens[0]) for p in policies])
        finally:
            plain.close(); hoisted.close()


def test_scripts_that_need_a_fallback_draw_are_left_to_the_per_episode_kernels(world):
    """Random best lists (any of the 61 actions, wrong lengths, empty years): the repair loop runs out of replayed deficit actions and
    takes the smart fallback (sampling.rs:492-528) — a seeded draw, different in every episode.  The hoist gives up at that point and
    the per-episode kernels run the episodes: same bytes as without the hoist, and as the tabled oracle."""
    tb = O.OracleTables(HostTables(world), len(world.existing_x))
    rng = np.random.default_rng(2025)
    plain, hoisted = _pair(world)
    try:
        fell_back = 0
        for trial in range(5):
            pol = ActionWeights()
            run = [rng.integers(0, 61, int(rng.choice([0, 0, 1, 2, 5, 12]))).tolist() for _ in range(26)]
            dfl = [(3 * rng.choice([8, 7, 12, 11, 9, 0, 1, 4, 10, 5, 2, 3, 13, 14], int(rng.choice([0, 1, 2, 3])))).tolist() for _ in range(26)]
            if trial >= 2:
                dfl[0] = rng.integers(0, 61, 3).tolist()
            nr = np.array([len(l) for l in run], np.int32); nd = np.array([len(l) for l in dfl], np.int32)
            pol.apply_episode([-5e4, 0.7, 4e10, 1.0], nr, np.array([a for l in run for a in l], np.uint8), nd, np.array([a for l in dfl for a in l], np.uint8))
            n = 96
            mask = (rng.uniform(size=n) < 0.4).astype(np.uint8)
            a = plain.rollout_batch(pol, 31337 + trial, n, first_episode_index=100 * trial, replay_mask=mask)
            b = hoisted.rollout_batch(pol, 31337 + trial, n, first_episode_index=100 * trial, replay_mask=mask)
            _same_records(a, b, trial)
            served = hoisted.replay_hoist_stats()[1]
            drew = bool((b.n_draws[mask == 1] > 0).any())
            assert served != drew, "a script is hoisted exactly when its replay episodes draw nothing"
            fell_back += int(not served)
            for e in np.flatnonzero(mask)[:4]:
                st, ref = O.run_episode_tabled(tb, oracle_weights_like(pol), 31337 + trial + 100 * trial + int(e), replay=True)
                assert_episode_equal(b, int(e), ref, f"fallback trial {trial}")
        assert fell_back >= 2
    finally:
        plain.close(); hoisted.close()


def test_capacities_are_the_per_episode_kernels(world):
    """A replay that would end with EG_EP_OVERFLOW (4 160 recorded actions) is not hoisted: the per-episode kernels report it, as before;
    a list of ~2 000 generators (within the capacity) is."""
    plain, hoisted = _pair(world)
    try:
        pol = ActionWeights()
        pol.apply_episode([-5e4, 0.7, 4e10, 1.0], np.full(26, 80, np.int32), np.full(26 * 80, 36, np.uint8), np.zeros(26, np.int32), np.zeros(0, np.uint8))
        n = 32
        mask = (np.arange(n) % 2 == 0).astype(np.uint8)
        a = plain.rollout_batch(pol, 5, n, replay_mask=mask)
        b = hoisted.rollout_batch(pol, 5, n, replay_mask=mask)
        assert (b.status[mask == 1] == -1).all() and (b.status[mask == 0] == 0).all()
        for name in ("status", "metrics", "n_run", "n_def", "n_act", "n_gens", "n_offsets", "n_draws"):
            assert getattr(a, name).tobytes() == getattr(b, name).tobytes(), name
        assert not hoisted.replay_hoist_stats()[1]
        rng = np.random.default_rng(2026)
        big = _full_script(rng, 76, [12, 2, 3, 8, 9, 4])
        a = plain.rollout_batch(big, 6, 12, replay_mask=np.ones(12, np.uint8))
        b = hoisted.rollout_batch(big, 6, 12, replay_mask=np.ones(12, np.uint8))
        assert (a.status == 0).all() and a.n_gens.min() > 1900
        _same_records(a, b, "2000 generators")
        assert hoisted.replay_hoist_stats()[1]
    finally:
        plain.close(); hoisted.close()


def test_tied_candidates_take_the_lowest_cell():
    """The symmetric world of test_heavy_searches_with_tied_candidates_match_the_oracle: up to eight cells share every score bit for
    bit, so the hoisted search meets several candidates within reach of the maximum, evaluates them exactly and must take the lowest
    cell — the reference's first-strictly-greater scan (metal_location_search.rs:168-171)."""
    w = World(np.array([25000.0]), np.array([25000.0]), np.array([400000], dtype=np.uint32), np.zeros(0), np.zeros(0),
              np.zeros(0, np.int32), np.zeros(0), np.zeros(0), np.zeros(0))
    tb = O.OracleTables(HostTables(w), 0)
    rng = np.random.default_rng(5)
    pol = ActionWeights()
    run = [[int(3 * rng.choice([0, 4, 12, 7, 8]) + rng.integers(0, 3)) for _ in range(9)] for _ in range(26)]
    dfl = [[int(3 * rng.choice([8, 7, 12, 0])) for _ in range(4)] for _ in range(26)]      # (four a year: the repair loop never runs out of them)
    pol.apply_episode([-5e4, 0.7, 4e10, 1.0], np.array([len(l) for l in run], np.int32), np.array([a for l in run for a in l], np.uint8),
                      np.full(26, 4, np.int32), np.array([a for l in dfl for a in l], np.uint8))
    plain, hoisted = _pair(w)
    try:
        n = 8
        a = plain.rollout_batch(pol, 77, n, replay_mask=np.ones(n, np.uint8))
        b = hoisted.rollout_batch(pol, 77, n, replay_mask=np.ones(n, np.uint8))
        assert (a.status == 0).all() and a.n_gens.min() >= 200
        _same_records(a, b, "symmetric world")
        assert hoisted.replay_hoist_stats()[1]
        st, ref = O.run_episode_tabled(tb, oracle_weights_like(pol), 77, replay=True)
        assert_episode_equal(b, 0, ref, "symmetric world, hoisted replay")
    finally:
        plain.close(); hoisted.close()


def test_training_loops_with_and_without_the_hoist_hold_the_same_policy(world):
    """The device-resident loop of configs[2] (16 384 episodes per update, every 10th a replay, from the seeded policy — the replayed
    list doubles with every replay that wins) for 12 updates, and the all-replay phase of the CLI's last 10 %
    (1 024 x 100 %): the policies after the last update — tables, best lists, counters — are equal, i.e. every update packet was."""
    for episodes, fraction, steps in ((16384, 0.1, 12), (1024, 1.0, 8)):
        states = []
        for hoist in (False, True):
            eng = Engine(world, device=0)
            try:
                eng.replay_hoist(hoist)
                pol = _seeded(eng)
                tr = BatchTrainer(eng, pol, episodes, 12345, replay_fraction=fraction)
                for _ in range(steps):
                    tr.step()
                tr.sync()
                res = eng.fetch(episodes)
                w, dw, cw = pol.tables()
                states.append((w.tobytes(), dw.tobytes(), repr(pol.lists(0)), repr(pol.lists(1)), pol.get("iteration_count"),
                               pol.get("iterations_without_improvement"), pol.get("best_cost"), pol.get("failed_episodes"),
                               tuple(_used(res, name).tobytes() for name in _ALL_FIELDS)))
                if hoist:
                    armed, served = eng.replay_hoist_stats()
                    assert armed >= steps and served
                    print(f"{episodes} x {fraction}: best list {sum(len(l) for l in pol.lists(0))} actions after {steps} updates, "
                          f"{int(res.n_gens[0])} generators per replay episode")
            finally:
                eng.close()
        assert states[0] == states[1], (episodes, fraction)


def test_the_rarely_run_paths_of_the_hoisted_search(world):
    """A hoisted search decides by one exchange when a single cell holds the maximum; several cells within reach of it are evaluated
    exactly (the reference's product in list order), and subnormal ranges / more than 64 such cells / no positive score take the exact
    scan of all cells.  EIRGRID_COOP_FORCE = 1 / 2 sends EVERY search down the second / third path: same bytes as the per-episode
    kernels for lists of ~140 and ~600 generators, other options (no energy sales, no yearly rows) and another world included."""
    from eirgrid_amd.world import synthetic_world
    rng = np.random.default_rng(777)
    other = synthetic_world(seed=0xE16D0003)
    for wld, kwargs in ((world, {}), (world, {"enable_energy_sales": False, "write_yearly": False}), (other, {})):
        policies = [_full_script(rng, 5, [0, 4, 12, 7, 1], offsets_per_year=1), _full_script(rng, 23, [0, 4, 12, 7, 5, 13, 9], offsets_per_year=2)]
        plain = Engine(wld, device=0)
        forced = {}
        for f in ("0", "1", "2"):
            os.environ["EIRGRID_COOP_FORCE"] = f
            try:
                forced[f] = Engine(wld, device=0)
            finally:
                del os.environ["EIRGRID_COOP_FORCE"]
            forced[f].replay_hoist(True)
        try:
            for k, pol in enumerate(policies):
                n = 40
                mask = (np.arange(n) % 4 == 1).astype(np.uint8)
                a = plain.rollout_batch(pol, 55 + k, n, replay_mask=mask, **kwargs)
                assert (a.status == 0).all()
                for f, eng in forced.items():
                    b = eng.rollout_batch(pol, 55 + k, n, replay_mask=mask, **kwargs)
                    for name in _ALL_FIELDS:
                        if name == "yearly" and not kwargs.get("write_yearly", True):
                            continue
                        assert _used(a, name).tobytes() == _used(b, name).tobytes(), (f, k, name, kwargs)
                    assert eng.replay_hoist_stats()[1], (f, k)
        finally:
            plain.close()
            for eng in forced.values():
                eng.close()
