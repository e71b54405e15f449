"""CPU: the batch ("reduced") update of the product (eg_policy_apply_reduced, formulas of csrc/eg_reduced_math.h through
include/eg_detpow.h) against the oracle's independent libm restatement (oracle/eg_oracle.c og_reduced_batch_update),
and the reduced update of ONE episode against the literal sequential update (multi_simulation.rs:494-508)."""
import numpy as np
import pytest

from eirgrid_amd import _native as N
from eirgrid_amd.engine import ActionWeights, HostTables, apply_reduced
from oracle import api as O
from tests.helpers import oracle_weights_like

SCALARS = ("iterations_without_improvement", "iteration_count", "has_best", "best_net_emissions", "best_opinion", "best_cost",
           "best_reliability", "has_best_actions", "has_best_deficit_actions")
OSC = dict(iterations_without_improvement="stall")


def oracle_batch(tables, snapshot: O.OracleWeights, seed, first, n, replay_period):
    """n tabled-oracle episodes (global indices first..first+n) against clones of one snapshot, as episode-major arrays."""
    status = np.zeros(n, np.int32); metrics = np.zeros((n, 4)); n_run = np.zeros((n, 26), np.int32); n_def = np.zeros((n, 26), np.int32)
    run_log = np.zeros((n, O.LOG_CAP), np.uint8); def_log = np.zeros((n, O.LOG_CAP), np.uint8)
    has_lists = snapshot.get("has_best_actions") == 1
    for e in range(n):
        replay = bool(replay_period and has_lists and (first + e) % replay_period == 0)
        st, out = O.run_episode_tabled(tables, snapshot.clone(), seed + first + e, replay=replay)
        status[e] = st; metrics[e] = out.metrics; n_run[e] = out.n_run; n_def[e] = out.n_def
        run_log[e] = np.frombuffer(out.run_log, np.uint8); def_log[e] = np.frombuffer(out.def_log, np.uint8)
    return status, metrics, n_run, n_def, run_log, def_log


def assert_policies_match(pol: ActionWeights, ow: O.OracleWeights, rtol, what):
    for x, y, name in zip(pol.tables()[:2], ow.tables()[:2], ("weights", "deficit weights")):
        np.testing.assert_allclose(x, y, rtol=rtol, atol=0, err_msg=f"{what}: {name}")
    assert pol.lists(0) == ow.lists(0) and pol.lists(1) == ow.lists(1), f"{what}: best lists"
    for name in SCALARS:
        assert pol.get(name) == ow.get(OSC.get(name, name)), f"{what}: {name}"
    if pol.get("has_best") == 1:
        pass


@pytest.fixture(scope="module")
def tables(world):
    return O.OracleTables(HostTables(world), len(world.existing_x))


def test_host_reduced_update_matches_independent_restatement(tables):
    """100 chained batches of 32 episodes (10 % ... 25 % replays once a best strategy exists): the product's host update,
    fed with the statistics the restatement derived from the episodes, ends every step within 1e-12 of the restatement
    (libm vs the shared IEEE-only exp/log/pow), with identical lists, counters and best metrics — through the first
    improvement, forced contrast (> 800), the stalled regime and the stagnation noise (> 1200)."""
    pol, ow = ActionWeights(), O.OracleWeights()
    n, period, seed = 32, 4, 777
    seen_stall = 0; improvements = 0; qualified = 0
    for step in range(100):
        first = step * n
        batch = oracle_batch(tables, ow, seed, first, n, period)
        assert (batch[0] == 0).all()
        improved_o, stats, winner = O.reduced_batch_update(ow, *batch, noise_seed=5000 + step)
        cand = (batch[1][winner], batch[2][winner], batch[4][winner][:N.RUN_CAP], batch[3][winner], batch[5][winner][:N.DEF_CAP])
        improved_p = apply_reduced(pol, stats, cand, noise_seed=5000 + step)
        assert improved_p == improved_o, f"step {step}"
        assert_policies_match(pol, ow, 1e-12, f"step {step}")
        improvements += improved_o; qualified += int(stats[2]); seen_stall = max(seen_stall, int(ow.get("stall")))
        assert stats[0] == n and stats[1] == 0
    assert improvements >= 1 and qualified > 0 and seen_stall > 1200
    w, dw, _ = pol.tables()
    assert w.min() >= 1e-4 and w.max() <= 0.999 and dw.min() >= 1e-4 and dw.max() <= 0.999


@pytest.mark.parametrize("stall0", [0, 95, 790, 1195])
def test_reduced_update_of_one_episode_is_the_sequential_update(tables, stall0):
    """Batch of ONE: og_reduced_batch_update (and the product's eg_policy_apply_reduced) against the literal sequential
    section multi_simulation.rs:494-508 (og_post_episode_update), step by step from identical states: every table entry
    (to the Q32 rounding of the logarithms: 2^-33 relative per action occurrence), every list, counter and best metric —
    including the reference's NaN-penalty quirk when an episode beats the best under forced contrast (stall > 800) and
    entries whose boosts saturate at MAX_WEIGHT before their mild penalties."""
    seq, red = O.OracleWeights(), O.OracleWeights()
    pol = ActionWeights()
    seed = 31337
    steps = 120
    saturating = 0; nan_quirk = 0; improved_n = 0
    for step in range(steps):
        if step >= 1 and stall0 and seq.get("stall") < stall0:      # keep the chain in the regime under test (an improvement resets the counter)
            for p in (seq, red):
                p.set("stall", stall0)
            pol.set("iterations_without_improvement", stall0)
        snapshot = seq.clone()
        replay = snapshot.get("has_best_actions") == 1 and step % 5 == 0
        st, out = O.run_episode_tabled(tables, snapshot, seed + step, replay=replay)      # `snapshot` now carries the episode's lists
        assert st == 0
        w0, dw0, _ = seq.tables()
        best0 = [seq.get_list(0, y) + seq.get_list(1, y) for y in range(26)]
        had_lists = seq.get("has_best_actions") == 1
        stall_before = seq.get("stall")
        s_best = O.score_metrics([seq.get(k) for k in ("best_net_emissions", "best_opinion", "best_cost", "best_reliability")]) if seq.get("has_best") else 0.0
        s_cur = O.score_metrics(list(out.metrics))
        det = (s_best - s_cur) / s_best if s_best > 0 else 0.0
        O.post_episode_update(seq, snapshot, list(out.metrics), noise_seed=9000 + step)
        batch = (np.array([0], np.int32), np.array([out.metrics]), np.array([out.n_run], np.int32), np.array([out.n_def], np.int32),
                 np.frombuffer(out.run_log, np.uint8)[None, :], np.frombuffer(out.def_log, np.uint8)[None, :])
        improved, stats, winner = O.reduced_batch_update(red, *batch, noise_seed=9000 + step)
        cand = (batch[1][0], batch[2][0], batch[4][0][:N.RUN_CAP], batch[3][0], batch[5][0][:N.DEF_CAP])
        assert apply_reduced(pol, stats, cand, noise_seed=9000 + step) == improved
        improved_n += improved
        # lists, counters, best metrics: always identical
        assert red.lists(0) == seq.lists(0) and red.lists(1) == seq.lists(1), f"step {step}"
        for name in ("stall", "iteration_count", "has_best", "best_net_emissions", "best_opinion", "best_cost", "best_reliability"):
            assert red.get(name) == seq.get(name), f"step {step}: {name}"
        w_s, dw_s, _ = seq.tables(); w_r, dw_r, _ = red.tables()
        if had_lists and stall_before > 800 and det < 0.0:
            nan_quirk += 1
        if had_lists and stats[2] == 1 and det > 0.0:      # boosted above the cap, then mildly penalised: the order of the clamps matters
            k = stall_before
            boost = 1.0 + seq.get("learning_rate") * (1.0 + 0.1 * k) * 2.0 * (1.0 + 0.2 * (k / 10.0) ** 1.8)
            mild = stats[8 + 26 * 61:8 + 2 * 26 * 61].reshape(26, 61)
            saturating += int(any(mild[y, a] != 0 and w0[y, a] * boost ** best0[y].count(a) > 0.999 for y in range(26) for a in set(best0[y])))
        tol = 1e-8
        np.testing.assert_allclose(w_r, w_s, rtol=tol, err_msg=f"step {step}: main table")
        np.testing.assert_allclose(dw_r, dw_s, rtol=tol, err_msg=f"step {step}: deficit table")
        # the product's host update follows the restatement to rounding
        np.testing.assert_allclose(pol.tables()[0], w_r, rtol=1e-12); np.testing.assert_allclose(pol.tables()[1], dw_r, rtol=1e-12)
        # next step starts from the sequential state in all three (so every step is an independent one-step comparison)
        red.set_tables(w_s, dw_s, None); pol.set_tables(w_s, dw_s, None)
    assert improved_n >= 1 and saturating > 0
    if stall0 > 800:
        assert nan_quirk > 0
    print(f"stall0={stall0}: {steps} steps, {improved_n} improvements; steps with saturating boosts + mild penalties {saturating}, with the NaN quirk {nan_quirk}")
