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
