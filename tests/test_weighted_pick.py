"""The device samplers pick by parallel prefix sums when that is provably the sequential sum-and-walk's answer
(eg_rollout.hip: weighted_pick).  This checks the bound the kernel relies on, on the CPU, with the same rule:
with P_a = prefix sums in tree order and T their last value, if every u * T - P_a is further than 2^-40 * T from zero, the
number of positive u * T - P_a equals the number of positive values of the reference's chains: total = w_0 + w_1 + ...
in order, v = u * total, v -= w_a in order."""
import numpy as np


def _tree_prefix(w):
    """Inclusive prefix sums in the order of the DPP scan: Hillis-Steele inside rows of 16, then row 15 -> rows 1,3 and
    lane 31 -> rows 2,3."""
    x = np.zeros(64); x[:len(w)] = w
    for d in (1, 2, 4, 8):
        y = x.copy()
        for i in range(64):
            if i % 16 >= d:
                y[i] = x[i] + x[i - d]
        x = y
    y = x.copy()
    for i in list(range(16, 32)) + list(range(48, 64)):
        y[i] = x[i] + x[(i // 16) * 16 - 1]
    x = y; y = x.copy()
    for i in range(32, 64):
        y[i] = x[i] + x[31]
    return y


def _sequential_pick(w, u):
    total = 0.0
    for x in w:
        total += x
    v = u * total; pick = 0
    for a in range(len(w)):
        v = v - w[a]; pick += v > 0.0
    return pick


def test_parallel_pick_agrees_whenever_it_is_trusted():
    rng = np.random.default_rng(5)
    trusted = ambiguous = 0
    for trial in range(4000):
        n = int(rng.choice([14, 21, 61]))
        w = np.clip(10 ** rng.uniform(-4, 0, n), 1e-4, 0.999)
        total = 0.0
        for x in w:
            total += x
        mode = trial % 4
        if mode == 0:
            u = rng.uniform()
        else:
            k = int(rng.integers(0, n)); b = 0.0
            for x in w[:k + 1]:
                b += x
            if mode == 1:         # on or within a few ulps of a boundary of the sequential walk: must be left to it
                u = float(np.nextafter(b / total, rng.choice([0.0, 1.0])) if rng.integers(0, 2) else b / total)
            else:                 # just outside the tolerance: the closest the parallel rule is ever trusted
                u = b / total + rng.choice([-1.0, 1.0]) * 2.0 ** -40 * (1.0 + 3.0 * rng.uniform())
            u = min(max(u, 0.0), float(np.nextafter(1.0, 0.0)))
        P = _tree_prefix(w)
        T = P[63]
        d = u * T - P[:n]
        if (np.abs(d) <= T * 2.0 ** -40).any():
            ambiguous += 1        # the kernel evaluates the sequential chains here
            continue
        trusted += 1
        assert int((d > 0).sum()) == _sequential_pick(w, u), (trial, n, u)
    assert trusted > 2500 and ambiguous > 900
