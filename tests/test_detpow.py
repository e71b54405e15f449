"""CPU: the shared deterministic x^p (include/eg_detpow.h) against glibc's pow on the weight domain."""
import numpy as np

from oracle import api as O


def test_detpow_close_to_libm(built):
    L = O.lib()
    rng = np.random.default_rng(0)
    worst = 0.0
    xs = np.concatenate([10 ** rng.uniform(-4, 0, 4000), [1e-4, 0.999, 1.0, 0.5, 0.08, 0.02]])
    for x in xs:
        for p in (1.0, 2.0, 2.002, 3.0, 4.6, 7.0):
            a, b = L.og_detpow(float(x), p), L.og_libm_pow(float(x), p)
            worst = max(worst, abs(a - b) / b)
    assert worst < 2e-14, worst
    assert L.og_detpow(1.0, 5.0) == 1.0
    assert L.og_detpow(0.5, 2.0) == 0.25


def test_stalled_sampler_with_libm_pow_picks_the_same_actions(built, world):
    """The reference's stalled sampler calls f64::powf = libm's pow (sampling.rs:199-213); oracle and kernels evaluate the shared
    eg_detpow instead, so that they agree to the bit.  This test takes eg_detpow OUT of the oracle (og_set_libm_pow: libm's pow, as
    the reference) and runs the stalled policies again: a pick differs only if a draw lands within 2e-14 of a boundary of the powered
    table — the test reports it if one does; none does in 4 x 48 episodes (about 9 000 stalled picks), and every output — action
    lists, placements, every float — is the same in both modes."""
    from eirgrid_amd.engine import HostTables
    tb = O.OracleTables(HostTables(world), len(world.existing_x))
    picks = 0
    for k, stall in enumerate((501, 900, 1500, 3500)):
        for e in range(48):
            outs = []
            for libm in (False, True):
                w = O.OracleWeights(); w.set("stall", stall)
                if libm:
                    with O.libm_pow():
                        st, ep = O.run_episode_tabled(tb, w, 4242 + k + e)
                else:
                    st, ep = O.run_episode_tabled(tb, w, 4242 + k + e)
                assert st == 0
                outs.append(ep)
            a, b = outs
            assert O.split_log(a.run_log, a.n_run) == O.split_log(b.run_log, b.n_run), f"stall {stall} episode {e}: a pick differs between eg_detpow and libm pow"
            assert O.split_log(a.act_log, a.n_act) == O.split_log(b.act_log, b.n_act)
            assert list(a.gen_cell[:a.n_gens]) == list(b.gen_cell[:b.n_gens]) and a.n_draws == b.n_draws
            assert np.array(a.yearly).tobytes() == np.array(b.yearly).tobytes() and list(a.metrics) == list(b.metrics)
            picks += int(sum(a.n_act))
    assert picks > 1000
