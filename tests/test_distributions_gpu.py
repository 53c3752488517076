"""Device samplers / moments (stand-alone C-ABI hooks) against the oracle and the
reference-generated vectors."""
import math

import numpy as np
import pytest

from bnmtf_amd import distributions as D
from oracle import rng

pytestmark = pytest.mark.gpu


def test_tn_draws_equal_oracle_candidate_sequence():
    n = 50000
    rs = np.random.RandomState(1)
    mu = rs.normal(0, 3, n); tau = rs.gamma(2.0, 2.0, n)
    x = D.TN_vector_draw(mu, tau, seed=99, it=4, col=7)
    y = rng.tn_draw(mu, tau, np.arange(n), 7, 4, rng.STREAM_HOOK, 99)
    rel = np.abs(x - y) / (1e-4 + np.abs(y))
    assert np.mean(rel < 1e-3) > 0.995          # fp32 vs fp64 candidate arithmetic
    assert (np.asarray(x) >= 0).all()


def test_tn_draw_distribution(golden):
    from scipy import stats
    g = golden("distributions.npz").case("draw")
    n = 200000
    for (mu, tau) in g["pairs"]:
        x = np.sort(D.TN_vector_draw(np.full(n, mu), np.full(n, tau), seed=5, it=0, col=1))
        sd = 1.0 / math.sqrt(tau)
        d = stats.truncnorm(-mu / sd, np.inf, loc=mu, scale=sd)
        ks = np.abs(d.cdf(x) - (np.arange(n) + 0.5) / n).max()
        assert ks < 2.7 / math.sqrt(n), (mu, tau, ks)
    # guards (truncated_normal_vector.py:41-45; test_draw :35-41)
    for _ in range(3):
        v1, v2 = D.TN_vector_draw([1.0, 0.32], [3.0, 0.0])
        assert v1 >= 0.0 and v2 == 0.0
    assert D.TN_draw(1.0, 3.0) >= 0 and D.TN_draw(0.3, 0.0) == 0.0


def test_tn_moments_match_reference(golden):
    g = golden("distributions.npz").case("mom")
    e = D.TN_vector_expectation(g["mu"], g["tau"]); v = D.TN_vector_variance(g["mu"], g["tau"])
    np.testing.assert_allclose(e, g["exp"], rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(v, g["var"], rtol=2e-6, atol=1e-300)   # 1 - lam*(lam-x) cancels near the -30 sigma switch
    e2 = D.TN_vector_expectation([1.0, -1], [3.0, 2000]); v2 = D.TN_vector_variance([1.0, -1], [3.0, 2000])
    assert e2[1] == 1. / 2000. and v2[1] == (1. / 2000.) ** 2
    assert abs(D.TN_expectation(1.0, 3.0) - e2[0]) < 1e-15 and D.TN_mode(-2.0) == 0.0 and D.TN_mode(1.0) == 1.0


def test_gamma_draw_matches_oracle():
    for i, (a, b) in enumerate([(30.0, 2.0), (0.5, 2.0), (3.6e6, 1.2e6), (2.0, 3.0)]):
        for it in range(5):
            x = D.gamma_draw(a, b, seed=7, it=it)
            y = rng.gamma_draw(a, b, it, 7)
            assert abs(x - y) <= 1e-9 * y
    assert D.gamma_expectation(2.0, 3.0) == 2.0 / 3.0
    assert D.gamma_expectation_log(2.0, 3.0) == -0.67582795356964265
    assert D.gamma_mode(2.0, 3.0) == 1. / 3.
