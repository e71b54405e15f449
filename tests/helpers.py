"""Shared comparison helpers: the HIP path's BatchResult vs the oracle's EpisodeOut, bit for bit."""
import numpy as np

from oracle import api as O


def assert_episode_equal(res, e, ref, what=""):
    """Bit-exact: action indices/counts, placement cells, draw count, every yearly float and the 4 metrics."""
    tag = f"{what} episode {e}"
    assert res.status[e] == ref.status, tag
    assert O.split_log(ref.run_log, ref.n_run) == res.lists(e, "run"), f"{tag}: current_run_actions differ"
    assert O.split_log(ref.def_log, ref.n_def) == res.lists(e, "def"), f"{tag}: current_deficit_actions differ"
    assert O.split_log(ref.act_log, ref.n_act) == res.lists(e, "act"), f"{tag}: recorded actions differ"
    assert res.n_gens[e] == ref.n_gens and res.n_offsets[e] == ref.n_offsets, tag
    n = ref.n_gens
    assert res.gen_cell[e, :n].tolist() == list(ref.gen_cell[:n]), f"{tag}: placement cells differ"
    pack = [t | (y << 4) | (m << 9) for t, y, m in zip(ref.gen_type[:n], ref.gen_year[:n], ref.gen_mult[:n])]
    assert res.gen_pack[e, :n].tolist() == pack, tag
    no = ref.n_offsets
    opack = [t | (y << 4) | (m << 9) for t, y, m in zip(ref.off_type[:no], ref.off_year[:no], ref.off_mult[:no])]
    assert res.off_pack[e, :no].tolist() == opack, tag
    assert int(res.n_draws[e]) == ref.n_draws, f"{tag}: RNG draw count differs"
    ya, yb = np.array(ref.yearly), res.yearly[e]
    if ya.tobytes() != yb.tobytes():
        bad = np.argwhere(ya != yb)
        raise AssertionError(f"{tag}: yearly rows differ first at (year, field) {bad[0]}: {ya[tuple(bad[0])]!r} vs {yb[tuple(bad[0])]!r}")
    assert np.array(ref.metrics).tobytes() == res.metrics[e].tobytes(), f"{tag}: metrics differ"
    # the north star's tolerance for float reward terms is 1e-5 relative; the bar here is bitwise, which implies it
    np.testing.assert_allclose(res.metrics[e], np.array(ref.metrics), rtol=1e-5)


def oracle_weights_like(policy):
    """An oracle ActionWeights equal to the product's eg_policy (tables, scalars, best lists)."""
    ow = O.OracleWeights()
    w, dw, cw = policy.tables()
    ow.set_tables(w, dw, cw)
    for name_o, name_p in (("learning_rate", "learning_rate"), ("exploration_rate", "exploration_rate"),
                           ("stall", "iterations_without_improvement"), ("iteration_count", "iteration_count"),
                           ("has_best", "has_best"), ("best_net_emissions", "best_net_emissions"),
                           ("best_opinion", "best_opinion"), ("best_cost", "best_cost"),
                           ("best_reliability", "best_reliability"), ("has_best_actions", "has_best_actions"),
                           ("has_best_deficit_actions", "has_best_deficit_actions")):
        ow.set(name_o, policy.get(name_p))
    ow.set_has_count_weights(int(policy.get("has_count_weights")))
    for which in (0, 1):
        for y in range(26):
            ow.set_list(which, y, policy.get_list(which, y))
    return ow
