"""Which run the reference summarises and exports: the `best_result` fold of core/multi_simulation.rs:384, :613-620
 — NOT the policy's best strategy.  CPU: the oracle's fold against a literal restatement and hand cases,
the product's host formula against the oracle's.  GPU: the on-device fold (k_fold_best) against the oracle's over the same
episodes, batch after batch."""
import numpy as np
import pytest

from oracle import api as O

MAX_COST = 50_000_000_000.0      # config/constants.rs:115


def _impact(cur, new, cost_only):
    """ai/metrics/scoring.rs:46-85 on SimulationMetrics quadruples, written out once more in Python."""
    if cost_only:
        return -(new[2] - cur[2]) / max(abs(cur[2]), 1.0)
    if cur[0] > 0.0:
        return (cur[0] - new[0]) / max(abs(cur[0]), 1.0)
    cost_improvement = -(new[2] - cur[2]) / max(abs(cur[2]), 1.0)
    opinion_improvement = (new[1] - cur[1]) / max(abs(cur[1]), 1.0)
    cw = 0.8 if cur[2] > MAX_COST * 8.0 else 0.5
    return cost_improvement * cw + opinion_improvement * (1.0 - cw)


def _fold(status, metrics, cost_only, best=None, index=None, first=0):
    """core/multi_simulation.rs:613-620, literally: results in iteration order, arguments as written."""
    for i, (st, m) in enumerate(zip(status, metrics)):
        if st != 0:
            continue
        if best is None or _impact(m, best, cost_only) > 0.0:
            best, index = m, first + i
    return best, index


def _random_metrics(rng, n):
    net = np.where(rng.uniform(size=n) < 0.5, rng.uniform(1.0, 9e5, n), -rng.uniform(0.0, 5e5, n))
    return np.stack([net, rng.uniform(0.2, 0.95, n), 10.0 ** rng.uniform(9.5, 12.2, n), (rng.uniform(size=n) < 0.9).astype(float)], axis=1)


def test_the_fold_keeps_the_run_the_held_one_improves_on(built):
    """Hand case.  Three runs above net zero with emissions 500, 300, 800: evaluate_action_impact(result -> best) > 0 means the
    HELD run has lower emissions than the newcomer, and then the newcomer takes over: 500 stays against 300, 800 takes over.
    The exported run is the one with the highest emissions, not the best-scoring one (300)."""
    m = np.array([[500.0, 0.7, 4e10, 1.0], [300.0, 0.7, 4e10, 1.0], [800.0, 0.7, 4e10, 1.0]])
    f = O.BestResultFold().feed(np.zeros(3, np.int32), m)
    assert f.winner == 2 and f.takeovers == 2
    assert int(np.argmax([O.score_metrics(r) for r in m])) == 1
    # cost_only reaches the fold as optimization_mode (multi_simulation.rs:616, main.rs:62): the dearest run is kept
    c = np.array([[-10.0, 0.7, 4e10, 1.0], [-10.0, 0.7, 9e10, 1.0], [-10.0, 0.7, 2e10, 1.0]])
    assert O.BestResultFold(cost_only=True).feed(np.zeros(3, np.int32), c).winner == 1
    # below net zero, default mode: half cost, half opinion (both relative to the newcomer)
    z = np.array([[-10.0, 0.5, 4e10, 1.0], [-10.0, 0.9, 4e10, 1.0], [-10.0, 0.6, 4e10, 1.0]])
    assert O.BestResultFold().feed(np.zeros(3, np.int32), z).winner == 0      # 0.9 and 0.6 are improvements on 0.5: it stays


def test_oracle_fold_against_the_literal_restatement(built):
    rng = np.random.default_rng(5)
    for cost_only in (False, True):
        for trial in range(20):
            n = int(rng.integers(1, 400))
            m = _random_metrics(rng, n)
            st = np.where(rng.uniform(size=n) < 0.1, -1, 0).astype(np.int32)
            f = O.BestResultFold(cost_only)
            best, index = None, None
            cut = int(rng.integers(0, n + 1))      # two batches: the fold carries over
            f.feed(st[:cut], m[:cut], 1000); f.feed(st[cut:], m[cut:], 1000 + cut)
            best, index = _fold(st, m, cost_only, first=1000)
            assert f.winner == index, (cost_only, trial)
            if index is not None:
                assert f.best.tobytes() == np.asarray(best).tobytes()


def test_host_impact_formula_is_the_oracles(built):
    from eirgrid_amd.engine import evaluate_action_impact
    rng = np.random.default_rng(9)
    m = _random_metrics(rng, 400)
    for cost_only in (False, True):
        for a, b in zip(m[:-1], m[1:]):
            want = O.evaluate_action_impact([a[0], a[1], 0.0, a[2]], [b[0], b[1], 0.0, b[2]], cost_only)
            got = evaluate_action_impact(a, b, cost_only)
            assert np.float64(got).tobytes() == np.float64(want).tobytes()
            assert np.float64(got).tobytes() == np.float64(_impact(a, b, cost_only)).tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("cost_only", [False, True])
def test_device_fold_is_the_sequential_fold(world, cost_only):
    """k_fold_best (1024 results at a time against the held run, the first that takes over becomes it, the window is examined
    again) must return what the sequential fold returns: index, metrics and the whole record of the held run, through several
    batches of uneven sizes (tiles of 8192, partial windows), replay episodes and a batch in which nothing takes over."""
    from eirgrid_amd.engine import ActionWeights, Engine
    eng = Engine(world, device=0)
    try:
        pol = ActionWeights()
        seed = 4242
        eng.track_best_result(cost_only)
        assert eng.fetch_best_result() == (None, None)
        fold = O.BestResultFold(cost_only)
        first, kept, kept_index = 0, None, None
        for n, replay in ((3000, False), (1024, True), (9000, False), (7, True), (1, False)):
            if replay and not pol.get("has_best_actions"):
                res0 = eng.rollout_batch(pol, seed, 1, first_episode_index=0)      # (folded too: the oracle sees it as well)
                fold.feed(res0.status, res0.metrics, 0)
                if fold.winner == 0 and kept_index != 0:
                    kept, kept_index = res0, 0
                pol.apply_episode(res0.metrics[0], res0.n_run[0], res0.run_log[0], res0.n_def[0], res0.def_log[0])
            mask = (np.arange(n) % 5 == 0).astype(np.uint8) if replay else None
            res = eng.rollout_batch(pol, seed, n, first_episode_index=first, replay_mask=mask)
            before = fold.winner
            fold.feed(res.status, res.metrics, first)
            if fold.winner != before:
                e = fold.winner - first
                kept = type(res)(*[np.ascontiguousarray(getattr(res, f.name)[e:e + 1]) for f in res.__dataclass_fields__.values()])
                kept_index = fold.winner
            idx, rec = eng.fetch_best_result()
            assert idx == fold.winner, (n, idx, fold.winner)
            assert rec.metrics[0].tobytes() == fold.best.tobytes()
            for name in ("yearly", "n_run", "n_def", "n_act", "n_gens", "n_offsets", "n_draws", "status"):
                assert getattr(rec, name).tobytes() == getattr(kept, name).tobytes(), name
            g = int(rec.n_gens[0])
            assert rec.gen_cell[0, :g].tobytes() == kept.gen_cell[0, :g].tobytes() and rec.lists(0, "act") == kept.lists(0, "act")
            first += n
        assert fold.takeovers >= 4, fold.takeovers      # (several restarts of a window were exercised)
        # the held run is not the best-scoring one: that is the point of restating the fold
        eng.track_best_result(cost_only)      # a new process starts at None again
        assert eng.fetch_best_result() == (None, None)
    finally:
        eng.close()
