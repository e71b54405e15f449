"""GPU: the per-episode replay kernel (csrc/eg_replay_solo.h: a long replay episode as script / placements / yearly rows on its own
wave) against the classic long-replay variant of k_rollout, which runs a replay year by year and action by action.

Bar: EVERY output byte of every episode — n_chunks included: the placements are k_rollout's own searches, so they request the same
candidate records —, the update packets and the policies of a training loop identical to the classic path's (EIRGRID_REPLAY_SOLO=0)
and to the tabled oracle's; a script that needs a seeded draw or hits a capacity is left to the classic variant."""
import os

import numpy as np
import pytest

from eirgrid_amd.engine import ActionWeights, Engine, HostTables
from eirgrid_amd.parallel import BatchTrainer
from oracle import api as O
from tests.helpers import assert_episode_equal, oracle_weights_like
from tests.test_gpu_parity import _ALL_FIELDS, _used
from tests.test_gpu_replay_hoist import _full_script, _seeded

pytestmark = pytest.mark.gpu


def _pair(world, pool=True):
    """(engine with the classic long-replay variant only, engine with k_replay_solo ahead of it); large-batch launch shape — the
    small-batch kernel (helper waves) has no such kernel.  pool=False: no penalty-field pool (EIRGRID_HEAVY_SLOTS=0) — every search
    of a long list is the exact scan"""
    os.environ["EIRGRID_HELPER_WAVES"] = "0"
    if not pool:
        os.environ["EIRGRID_HEAVY_SLOTS"] = "0"
    try:
        os.environ["EIRGRID_REPLAY_SOLO"] = "0"
        classic = Engine(world, device=0)
        os.environ["EIRGRID_REPLAY_SOLO"] = "1"
        solo = Engine(world, device=0)
    finally:
        os.environ.pop("EIRGRID_REPLAY_SOLO", None); os.environ.pop("EIRGRID_HELPER_WAVES", None); os.environ.pop("EIRGRID_HEAVY_SLOTS", None)
    return classic, solo


def _same_records(a, b, what):
    for name in _ALL_FIELDS + ("n_chunks",):
        assert _used(a, name).tobytes() == _used(b, name).tobytes(), (what, name)


def test_solo_replays_are_the_classic_records(world):
    """Lists of ~120, ~600 and ~2 000 generators (within the on-chip window, beyond it, near the capacity), with offsets and no-op
    actions; every 10th episode replays.  Against the classic variant byte for byte, against the tabled oracle, and every replay
    episode of a batch is the same computation."""
    tb = O.OracleTables(HostTables(world), len(world.existing_x))
    rng = np.random.default_rng(515)
    classic, solo = _pair(world)
    try:
        policies = [_full_script(rng, 5, [0, 4, 12, 7], offsets_per_year=1), _full_script(rng, 23, [0, 4, 12, 7, 5, 1, 13], offsets_per_year=2),
                    _full_script(rng, 72, [0, 4, 12, 7, 5, 1, 13, 2, 9], offsets_per_year=3)]
        for k, pol in enumerate(policies):
            n = 160
            mask = (np.arange(n) % 10 == 7).astype(np.uint8)
            a = classic.rollout_batch(pol, 700 + k, n, first_episode_index=1000 * k, replay_mask=mask)
            b = solo.rollout_batch(pol, 700 + k, n, first_episode_index=1000 * k, replay_mask=mask)
            assert (a.status == 0).all(), np.unique(a.status)
            _same_records(a, b, k)
            reps = np.flatnonzero(mask)
            assert (b.n_draws[reps] == 0).all()
            for name in _ALL_FIELDS:
                u = _used(b, name)
                assert all(u[reps[0]].tobytes() == u[e].tobytes() for e in reps), name
            for e in (int(reps[0]), int(reps[-1])):
                st, ref = O.run_episode_tabled(tb, oracle_weights_like(pol), 700 + k + 1000 * k + e, replay=True)
                assert_episode_equal(b, e, ref, f"solo, policy {k}")
        print("generators per replay episode:", [int(solo.rollout_batch(p, 1, 10, replay_mask=np.ones(10, np.uint8)).n_gens[0]) for p in policies])
    finally:
        classic.close(); solo.close()


def test_scripts_the_solo_kernel_leaves_to_the_classic_variant(world):
    """Random long best lists (any of the 61 actions, wrong lengths, empty years): the repair loop runs out of replayed deficit actions
    and takes the smart fallback (sampling.rs:492-528) — a seeded draw.  k_replay_solo publishes nothing for such an episode and the
    classic variant runs it from the start: the same bytes as without the kernel, and as the tabled oracle's.  Also a list that ends
    in EG_EP_OVERFLOW (4 160 recorded actions)."""
    tb = O.OracleTables(HostTables(world), len(world.existing_x))
    rng = np.random.default_rng(77)
    classic, solo = _pair(world)
    try:
        drew = 0
        for trial in range(4):
            pol = ActionWeights()
            run = [rng.integers(0, 61, int(rng.choice([0, 3, 8, 12, 20]))).tolist() for _ in range(26)]
            dfl = [(3 * rng.choice([8, 7, 12, 11, 9, 0, 1, 4, 10, 5, 2, 3, 13, 14], int(rng.choice([0, 1, 2, 3])))).tolist() for _ in range(26)]
            nr = np.array([len(l) for l in run], np.int32); nd = np.array([len(l) for l in dfl], np.int32)
            pol.apply_episode([-5e4, 0.7, 4e10, 1.0], nr, np.array([a for l in run for a in l], np.uint8), nd, np.array([a for l in dfl for a in l], np.uint8))
            assert nr.sum() > 96      # a long list: the long-replay variant's
            n = 64
            mask = (rng.uniform(size=n) < 0.4).astype(np.uint8)
            a = classic.rollout_batch(pol, 4242 + trial, n, first_episode_index=100 * trial, replay_mask=mask)
            b = solo.rollout_batch(pol, 4242 + trial, n, first_episode_index=100 * trial, replay_mask=mask)
            _same_records(a, b, trial)
            drew += int((b.n_draws[mask == 1] > 0).any())
            for e in np.flatnonzero(mask)[:3]:
                st, ref = O.run_episode_tabled(tb, oracle_weights_like(pol), 4242 + trial + 100 * trial + int(e), replay=True)
                assert_episode_equal(b, int(e), ref, f"fallback trial {trial}")
        assert drew >= 2
        pol = ActionWeights()
        pol.apply_episode([-5e4, 0.7, 4e10, 1.0], np.full(26, 80, np.int32), np.full(26 * 80, 36, np.uint8), np.zeros(26, np.int32), np.zeros(0, np.uint8))
        mask = np.ones(8, np.uint8)
        a = classic.rollout_batch(pol, 5, 8, replay_mask=mask); b = solo.rollout_batch(pol, 5, 8, replay_mask=mask)
        assert (a.status != 0).all()
        assert a.status.tobytes() == b.status.tobytes()
    finally:
        classic.close(); solo.close()


def test_training_loops_with_and_without_the_solo_kernel(world):
    """The device-resident loop of configs[2] (16 384 episodes per update, every 10th a replay, from the seeded policy — the replayed
    list doubles with every replay that wins) for 12 updates: the policies after the last update — tables, best lists,
    counters — and the last batch's records are equal, i.e. every update packet was."""
    states = []
    for eng in _pair(world):
        try:
            pol = _seeded(eng)
            tr = BatchTrainer(eng, pol, 16384, 12345, replay_fraction=0.1)
            for _ in range(12):
                tr.step()
            tr.sync()
            res = eng.fetch(16384)
            w, dw, cw = pol.tables()
            states.append((w.tobytes(), dw.tobytes(), repr(pol.lists(0)), repr(pol.lists(1)), pol.get("iteration_count"),
                           pol.get("iterations_without_improvement"), pol.get("best_cost"), pol.get("failed_episodes"),
                           tuple(_used(res, name).tobytes() for name in _ALL_FIELDS + ("n_chunks",))))
            print(f"best list {sum(len(l) for l in pol.lists(0))} actions after 12 updates, {int(res.n_gens[0])} generators per replay episode")
        finally:
            eng.close()
    assert states[0] == states[1]
    assert sum(len(l) for l in pol.lists(0)) > 96, "the loop has reached the long-replay variant"


def test_other_options_another_world_and_no_field_pool(world):
    """The run options that change what a replay writes (no energy sales, no yearly rows), another synthetic world, and engines without
    a penalty-field pool (every search of a long list is place_search / place_exact_long): same bytes as the classic variant."""
    from eirgrid_amd.world import synthetic_world
    rng = np.random.default_rng(919)
    other = synthetic_world(seed=0xE16D0003)
    for wld, kwargs, pool in ((world, {"enable_energy_sales": False, "write_yearly": False}, True), (other, {}, True), (world, {}, False)):
        classic, solo = _pair(wld, pool)
        try:
            for k, pol in enumerate((_full_script(rng, 6, [0, 4, 12, 7, 1], offsets_per_year=1), _full_script(rng, 26, [0, 4, 12, 7, 5, 13, 9], offsets_per_year=2))):
                n = 48
                mask = (np.arange(n) % 4 == 1).astype(np.uint8)
                a = classic.rollout_batch(pol, 66 + k, n, replay_mask=mask, **kwargs)
                b = solo.rollout_batch(pol, 66 + k, n, replay_mask=mask, **kwargs)
                assert (a.status == 0).all()
                for name in _ALL_FIELDS + ("n_chunks",):
                    if name == "yearly" and not kwargs.get("write_yearly", True):
                        continue
                    assert _used(a, name).tobytes() == _used(b, name).tobytes(), (k, name, kwargs, pool)
        finally:
            classic.close(); solo.close()
