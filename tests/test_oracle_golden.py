"""The oracle (oracle/bnmtf_oracle.py) against vectors produced by the reference
itself (tests/golden/make_golden.py) and the reference's own known-answer tests.
CPU only."""
import math

import numpy as np
import pytest

from oracle import bnmtf_oracle as O
from oracle import rng

TOL = 1e-12


def _bnmf(c):
    pri = dict(alpha=float(c["alpha"]), beta=float(c["beta"]), lambdaU=c["lambdaU"], lambdaV=c["lambdaV"])
    b = O.BNMFGibbsOracle(c["R"], c["M"], int(c["K"]), pri)
    b.U, b.V, b.tau = c["U"].copy(), c["V"].copy(), float(c["tau"])
    return b


@pytest.mark.parametrize("name", ["t5x3", "toy", "r37x29", "r40x33"])
def test_bnmf_conditionals_match_reference(golden, name):
    c = golden("bnmf_gibbs_cond.npz").case(name)
    b = _bnmf(c)
    assert b.alpha_s() == float(c["alpha_s"])
    assert abs(b.beta_s() - float(c["beta_s"])) <= TOL * abs(float(c["beta_s"]))
    for k in range(b.K):
        tU = b.tauU(k); tV = b.tauV(k)
        np.testing.assert_allclose(tU, c["tauU"][k], rtol=TOL, atol=0)
        np.testing.assert_allclose(b.muU(tU, k), c["muU"][k], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(tV, c["tauV"][k], rtol=TOL, atol=0)
        np.testing.assert_allclose(b.muV(tV, k), c["muV"][k], rtol=1e-10, atol=1e-12)
    p = b.predict_while_running()
    np.testing.assert_allclose([p["MSE"], p["R^2"], p["Rp"]], c["perf"], rtol=1e-12, equal_nan=True)
    # bit-exact integer counts
    assert int(b.size_Omega) == int(c["size_Omega"])
    assert np.array_equal(b.M.sum(axis=1).astype(np.int64), c["row_counts"])
    assert np.array_equal(b.M.sum(axis=0).astype(np.int64), c["col_counts"])
    # sums form of the metrics (what the HIP path returns) agrees with the direct form
    s = O.metric_sums(b.M, b.R, b.U @ b.V.T)
    m = O.metrics_from_sums(s)
    np.testing.assert_allclose([m["MSE"], m["R^2"]], c["perf"][:2], rtol=1e-9)


@pytest.mark.parametrize("name", ["t5x3", "toy", "r37x29"])
def test_bnmf_postrun_matches_reference(golden, name):
    c = golden("bnmf_gibbs_cond.npz").case(name)
    b = _bnmf(c)
    b.all_U, b.all_V, b.all_tau = list(c["all_U"]), list(c["all_V"]), list(c["all_tau"])
    eU, eV, et = b.approx_expectation(2, 3)
    np.testing.assert_allclose(eU, c["expU"], rtol=1e-14); np.testing.assert_allclose(eV, c["expV"], rtol=1e-14)
    assert abs(et - float(c["exptau"])) < 1e-14
    p = b.predict(c["M_test"], 2, 3)
    np.testing.assert_allclose([p["MSE"], p["R^2"], p["Rp"]], c["predict"], rtol=1e-12)
    q = [b.quality(m, 2, 3) for m in ["loglikelihood", "BIC", "AIC", "MSE", "ELBO"]]
    np.testing.assert_allclose(q, c["quality"], rtol=1e-12)
    with pytest.raises(AssertionError) as e:
        b.quality("FAIL", 2, 3)
    assert str(e.value) == "Unrecognised metric for model quality: FAIL."


def test_bnmf_known_answers_of_reference_tests():
    """tests/code/test_bnmf_gibbs_optimised.py:144-203 closed forms."""
    I, J, K = 5, 3, 2
    R = np.ones((I, J)); M = np.ones((I, J)); M[0, 0] = M[2, 2] = M[3, 1] = 0
    pri = dict(alpha=3, beta=1, lambdaU=2 * np.ones((I, K)), lambdaV=3 * np.ones((J, K)))
    b = O.BNMFGibbsOracle(R, M, K, pri); b.initialise("exp")
    assert (b.U == 0.5).all() and (b.V == 1 / 3.).all()
    assert b.alpha_s() == 3 + 6.
    assert abs(b.beta_s() - (1 + .5 * (12 * (2. / 3.) ** 2))) < 1e-15
    b.tau = 3.
    tauU = 3. * np.array([[2. / 9.] * 2, [1. / 3.] * 2, [2. / 9.] * 2, [2. / 9.] * 2, [1. / 3.] * 2])
    muU = 1. / tauU * (3. * np.array([[2. * (5. / 6.) * (1. / 3.), 10. / 18.], [15. / 18.] * 2, [10. / 18.] * 2, [10. / 18.] * 2, [15. / 18.] * 2]) - 2.)
    for k in range(K):
        assert np.array_equal(b.tauU(k), tauU[:, k])
        assert np.abs(b.muU(tauU[:, k], k) - muU[:, k]).max() < 1e-15
        assert np.array_equal(b.tauV(k), 3. * np.ones(J))
        assert np.abs(b.muV(3. * np.ones(J), k) - (1. / 3.) * (3. * 4. * (5. / 6.) * .5 - 3.)).max() < 1e-15


def test_constructor_messages():
    """Exact assertion strings of bnmf_gibbs_optimised.py:59-62,77-78,88,90."""
    pri = dict(alpha=3, beta=1, lambdaU=np.ones((5, 1)), lambdaV=np.ones((3, 1)))
    with pytest.raises(AssertionError) as e:
        O.BNMFGibbsOracle(np.ones(3), np.ones((2, 3)), 1, pri)
    assert str(e.value) == "Input matrix R is not a two-dimensional array, but instead 1-dimensional."
    with pytest.raises(AssertionError) as e:
        O.BNMFGibbsOracle(np.ones((3, 2)), np.ones((2, 3)), 1, pri)
    assert str(e.value) == "Input matrix R is not of the same size as the indicator matrix M: (3, 2) and (2, 3) respectively."
    pri2 = dict(alpha=3, beta=1, lambdaU=np.ones((3, 1)), lambdaV=np.ones((3, 1)))
    with pytest.raises(AssertionError) as e:
        O.BNMFGibbsOracle(np.ones((2, 3)), np.ones((2, 3)), 1, pri2)
    assert str(e.value) == "Prior matrix lambdaU has the wrong shape: (3, 1) instead of (2, 1)."
    pri3 = dict(alpha=3, beta=1, lambdaU=np.ones((2, 1)), lambdaV=np.ones((3, 1)))
    with pytest.raises(AssertionError) as e:
        O.BNMFGibbsOracle(np.ones((2, 3)), [[1, 1, 1], [0, 0, 0]], 1, pri3)
    assert str(e.value) == "Fully unobserved row in R, row 1."
    with pytest.raises(AssertionError) as e:
        O.BNMFGibbsOracle(np.ones((2, 3)), [[1, 1, 0], [1, 0, 0]], 1, pri3)
    assert str(e.value) == "Fully unobserved column in R, column 2."


@pytest.mark.parametrize("name", ["t5x3", "toy", "r37x29"])
def test_bnmtf_conditionals_match_reference(golden, name):
    c = golden("bnmtf_gibbs_cond.npz").case(name)
    pri = dict(alpha=float(c["alpha"]), beta=float(c["beta"]), lambdaF=c["lambdaF"], lambdaS=c["lambdaS"], lambdaG=c["lambdaG"])
    b = O.BNMTFGibbsOracle(c["R"], c["M"], int(c["K"]), int(c["L"]), pri)
    b.F, b.S, b.G, b.tau = c["F"].copy(), c["S"].copy(), c["G"].copy(), float(c["tau"])
    assert abs(b.beta_s() - float(c["beta_s"])) <= 1e-12 * abs(float(c["beta_s"]))
    for k in range(b.K):
        t = b.tauF(k)
        np.testing.assert_allclose(t, c["tauF"][k], rtol=TOL)
        np.testing.assert_allclose(b.muF(t, k), c["muF"][k], rtol=1e-10, atol=1e-12)
        for l in range(b.L):
            ts = b.tauS(k, l)
            assert abs(ts - c["tauS"][k, l]) <= 1e-12 * abs(ts)
            assert abs(b.muS(ts, k, l) - c["muS"][k, l]) <= 1e-10 * max(1.0, abs(c["muS"][k, l]))
    for l in range(b.L):
        t = b.tauG(l)
        np.testing.assert_allclose(t, c["tauG"][l], rtol=TOL)
        np.testing.assert_allclose(b.muG(t, l), c["muG"][l], rtol=1e-10, atol=1e-12)
    p = b.predict_while_running()
    np.testing.assert_allclose([p["MSE"], p["R^2"], p["Rp"]], c["perf"], rtol=1e-12, equal_nan=True)
    b.all_F, b.all_S, b.all_G, b.all_tau = list(c["all_F"]), list(c["all_S"]), list(c["all_G"]), list(c["all_tau"])
    pp = b.predict(c["M_test"], 2, 3)
    np.testing.assert_allclose([pp["MSE"], pp["R^2"], pp["Rp"]], c["predict"], rtol=1e-12)
    q = [b.quality(m, 2, 3) for m in ["loglikelihood", "BIC", "AIC", "MSE", "ELBO"]]
    np.testing.assert_allclose(q, c["quality"], rtol=1e-12)


def test_vb_toy_trajectory_matches_reference(golden):
    g = golden("bnmf_vb.npz").case("toy")
    t = golden("toy_data.npz").case("bnmf")
    I, J = t["R"].shape; K = 10
    b = O.BNMFVBOracle(t["R"], t["M"], K, dict(alpha=1., beta=1., lambdaU=0.1 * np.ones((I, K)), lambdaV=0.1 * np.ones((J, K))))
    b.initialise("exp")
    assert abs(b.exptau - float(g["init_exptau"])) < 1e-12 * b.exptau
    np.testing.assert_allclose(b.expU, g["init_expU"], rtol=1e-13)
    assert abs(b.exp_square_diff() - float(g["init_esd"])) < 1e-12 * float(g["init_esd"])
    b.run(20)
    np.testing.assert_allclose(b.all_performances["MSE"], g["mse"], rtol=1e-9)
    np.testing.assert_allclose(b.all_exp_tau, g["exptau"], rtol=1e-9)
    np.testing.assert_allclose(b.all_elbo, g["elbo"], rtol=1e-9)
    for nm in ["expU", "expV", "varU", "varV", "muU", "muV", "tauU", "tauV"]:
        np.testing.assert_allclose(getattr(b, nm), g["it20/" + nm], rtol=1e-7, atol=1e-12)
    # numbers quoted in SURVEY.md 8(c)(3)
    assert abs(b.all_performances["MSE"][0] - 19416.2628562074) < 1e-6
    assert abs(b.all_performances["MSE"][19] - 1.5678561910685924) < 1e-9
    q = [b.quality(m) for m in ["loglikelihood", "BIC", "AIC", "MSE", "ELBO"]]
    np.testing.assert_allclose(q, g["quality"], rtol=1e-9)


def test_vb_ragged_case_matches_reference(golden):
    g = golden("bnmf_vb.npz").case("r31x23")
    K = 4
    b = O.BNMFVBOracle(g["R"], g["M"], K, dict(alpha=2., beta=.5, lambdaU=g["lambdaU"], lambdaV=g["lambdaV"]))
    b.initialise("exp", {"tauU": g["tauU0"], "tauV": g["tauV0"]})
    b.run(10)
    np.testing.assert_allclose(b.all_performances["MSE"], g["mse"], rtol=1e-9)
    np.testing.assert_allclose(b.all_elbo, g["elbo"], rtol=1e-9)
    np.testing.assert_allclose(b.expU, g["it10/expU"], rtol=1e-7, atol=1e-12)


def test_vb_known_answers_of_reference_tests():
    """tests/code/test_bnmf_vb_optimised.py:230-238 (exp_square_diff) and
    tests/code/distributions/test_gamma.py:18-23."""
    I, J, K = 5, 3, 2
    R = np.ones((I, J)); M = np.ones((I, J)); M[0, 0] = M[2, 2] = M[3, 1] = 0
    b = O.BNMFVBOracle(R, M, K, dict(alpha=3, beta=1, lambdaU=2 * np.ones((I, K)), lambdaV=3 * np.ones((J, K))))
    b.expU = 1. / 2 * np.ones((I, K)); b.expV = 1. / 3 * np.ones((J, K))
    b.varU = 2 * np.ones((I, K)); b.varV = 3 * np.ones((J, K))
    assert b.exp_square_diff() == 172.66666666666666
    b.update_tau()
    assert b.alpha_s == 3 + 12. / 2. and b.beta_s == 1 + 172.66666666666666 / 2.
    b.exptau = 3.; b.muU = np.zeros((I, K)); b.tauU = np.zeros((I, K))
    b.update_U(0)   # test_update_U :250-264
    for i in range(I):
        w = (M[i] * (b.expV[:, 0] ** 2 + b.varV[:, 0])).sum()
        assert b.tauU[i, 0] == 3. * w
        assert abs(b.muU[i, 0] - (1. / (3. * w)) * (-2. + 3. * (M[i] * ((R[i] - b.expU[i] @ b.expV.T + b.expU[i, 0] * b.expV[:, 0]) * b.expV[:, 0])).sum())) < 1e-15
    b2 = O.BNMFVBOracle(R, M, K, dict(alpha=3, beta=1, lambdaU=2 * np.ones((I, K)), lambdaV=3 * np.ones((J, K))))
    b2.initialise()   # test_update_exp_U / test_update_exp_tau :284-311
    assert abs(b2.exptau - (3 + 12. / 2.) / (1 + 35.4113198623 / 2.)) < 1e-12
    assert abs(b2.explogtau - (2.1406414779556 - math.log(1 + 35.4113198623 / 2.))) < 1e-12
    b2.tauU = 4 * np.ones((I, K)); b2.update_exp_U(0)
    assert np.abs(b2.expU[:, 0] - (0.5 + 0.5 * 0.2876155949126352)).max() < 1e-5
    assert np.abs(b2.varU[:, 0] - 0.25 * (1. - 0.37033832534958433)).max() < 1e-5
    assert O.gamma_expectation_log(2.0, 3.0) == -0.67582795356964265
    assert O.gamma_expectation(2.0, 3.0) == 2.0 / 3.0
    assert O.gamma_mode(2.0, 3.0) == 1. / 3.


def test_tn_moments_match_reference(golden):
    g = golden("distributions.npz").case("mom")
    with np.errstate(all="ignore"):
        e = O.tn_expectation(g["mu"], g["tau"]); v = O.tn_variance(g["mu"], g["tau"])
    np.testing.assert_allclose(e, g["exp"], rtol=1e-13, atol=0)
    np.testing.assert_allclose(v, g["var"], rtol=1e-13, atol=0)
    # tests/code/distributions/test_truncated_normal_vector.py:13-31
    e2 = O.tn_expectation([1.0, -1], [3.0, 2000]); v2 = O.tn_variance([1.0, -1], [3.0, 2000])
    assert e2[1] == 1. / 2000. and v2[1] == (1. / 2000.) ** 2
    gg = golden("distributions.npz").case("gamma")
    for (a, bb), ex, el, mo in zip(gg["ab"], gg["exp"], gg["explog"], gg["mode"]):
        assert O.gamma_expectation(a, bb) == ex and O.gamma_expectation_log(a, bb) == el and O.gamma_mode(a, bb) == mo


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    def h(x):
        return [int(v) for v in x]
    assert h(rng.philox4x32_10(0, 0, 0, 0, 0)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert h(rng.philox4x32_10(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffffffffffff)) == \
        [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert h(rng.philox4x32_10(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, (0x299f31d0 << 32) | 0xa4093822)) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def _tn_exact(mu, tau):
    from scipy import stats
    sd = 1.0 / math.sqrt(tau)
    return stats.truncnorm(-mu / sd, np.inf, loc=mu, scale=sd)


def test_tn_sampler_matches_reference_distribution(golden):
    """The new sampler against quantiles of 2e5 draws of the reference's
    TN_vector_draw per (mu,tau) pair, all three rtnorm regimes
    (a = -mu*sqrt(tau) in {-10,-2.5,-1,0,.2,.3,1,3,3.6,8,40}, tau in {1,37}).

    Finding pinned here: the reference's vendored table sampler (rtnorm.py:128-217)
    is itself off the exact truncated normal near its right-tail cell
    (x ~ 3.2..3.49 sigma): tiny for a in [0,1] (about 5e-4 of mass missing beyond
    3.2 sigma) and gross for a in (2.9, 3.4867) (CDF error up to 0.12 at a = 3).
    The new sampler follows the exact distribution the reference documents
    (truncated_normal_vector.py:1-5); it is compared with the reference's
    quantiles wherever the reference agrees with the exact CDF, and with the
    exact CDF everywhere."""
    g = golden("distributions.npz").case("draw")
    n = 200000; nref = int(g["n"]); p = g["probs"]
    n_ref_ok = 0
    for (mu, tau), q, mom in zip(g["pairs"], g["quantiles"], g["moments"]):
        x = rng.tn_draw(np.full(n, mu), np.full(n, tau), np.arange(n), 3, 11, rng.STREAM_HOOK, 2718)
        assert (x >= 0).all()
        d = _tn_exact(mu, tau)
        xs = np.sort(x)
        # (1) exact CDF, every pair: Kolmogorov distance below the 1e-6-level critical value
        ks = np.abs(d.cdf(xs) - (np.arange(n) + 0.5) / n).max()
        assert ks < 2.7 / math.sqrt(n), (mu, tau, ks)
        # (2) reference quantiles, where the reference itself is on the exact CDF
        ref_dev = np.abs(d.cdf(q) - p)
        ref_ok = ref_dev <= 4.5 * np.sqrt(p * (1 - p) / nref) + 1.0 / nref
        emp = np.searchsorted(xs, q, side="right") / float(n)
        band = 4.5 * np.sqrt(p * (1 - p) * (1. / n + 1. / nref)) + 2.0 / n
        assert (np.abs(emp - p)[ref_ok] <= band[ref_ok]).all(), (mu, tau)
        a = -mu * math.sqrt(tau)
        if abs(a - 3.0) < 1e-9:
            assert ref_dev.max() > 0.05      # the reference's own deviation, documented above
        else:
            assert ref_ok.sum() >= len(p) - 3, (mu, tau, ref_ok.sum())
            n_ref_ok += 1
            sd = math.sqrt(mom[1])
            assert abs(x.mean() - mom[0]) < 6 * sd * math.sqrt(2. / n)
    assert n_ref_ok == 20
    # guards of truncated_normal_vector.py:41-45
    x = rng.tn_draw([1.0, 0.32, np.nan], [3.0, 0.0, 1.0], [0, 1, 2], 0, 0, rng.STREAM_HOOK, 1)
    assert x[0] >= 0 and x[1] == 0.0 and x[2] == 0.0


def test_gamma_sampler_distribution():
    from scipy import stats
    for shape, rate in [(30.0, 2.0), (0.5, 2.0), (3.6e3, 1.2e3)]:
        g = np.array([rng.gamma_draw(shape, rate, i, 7) for i in range(4000)])
        assert stats.kstest(g, stats.gamma(shape, scale=1.0 / rate).cdf).pvalue > 1e-4


def test_gibbs_toy_trajectory_within_reference_bands(golden):
    """Oracle Gibbs (new sampler) on the toy set vs 10 seeded reference runs:
    early iterations inside a widened min/max band, converged level equal."""
    t = golden("toy_data.npz").case("bnmf")
    g = golden("gibbs_trajectories.npz").case("bnmf")
    I, J = t["R"].shape; K = 10
    pri = dict(alpha=1., beta=1., lambdaU=0.1 * np.ones((I, K)), lambdaV=0.1 * np.ones((J, K)))
    b = O.BNMFGibbsOracle(t["R"], t["M"], K, pri, seed=5)
    b.U, b.V, b.tau = g["U0_seed0"].copy(), g["V0_seed0"].copy(), float(g["tau0_seed0"])
    b.run(200)
    mse = np.array(b.all_performances["MSE"]); ref = g["mse"]
    lo, hi = ref.min(axis=0), ref.max(axis=0)
    assert (mse[:60] > lo[:60] / 2.5).all() and (mse[:60] < hi[:60] * 2.5).all()
    # converged regime: mean of last 50 iterations within the spread of the reference seeds
    m_ref = ref[:, 150:].mean(axis=1)
    assert m_ref.min() * 0.97 < mse[150:].mean() < m_ref.max() * 1.03
    assert abs(np.mean(b.all_tau[150:]) - g["tau"][:, 150:].mean()) < 0.05


def test_oracle_matches_the_reference_at_ranks_above_64(golden):
    """tests/golden/wide_rank.npz (round 6): the reference's conditional parameters for every column of a K = 96 model and its
    nmf_icm trajectory at K = 70 -- what pins the oracle where the device runs column blocks (tests/test_wide_rank_gpu.py)."""
    c = golden("wide_rank.npz").case("k96")
    K = int(c["K"])
    pri = dict(alpha=float(c["alpha"]), beta=float(c["beta"]), lambdaU=c["lambdaU"], lambdaV=c["lambdaV"])
    o = O.BNMFGibbsOracle(c["R"], c["M"], K, pri)
    o.U, o.V, o.tau = c["U"].copy(), c["V"].copy(), float(c["tau"])
    for k in (0, 63, 64, 95):
        t = o.tauU(k)
        np.testing.assert_allclose(t, c["tauU"][k], rtol=1e-12)
        np.testing.assert_allclose(o.muU(t, k), c["muU"][k], rtol=1e-9, atol=1e-12)
        t = o.tauV(k)
        np.testing.assert_allclose(t, c["tauV"][k], rtol=1e-12)
        np.testing.assert_allclose(o.muV(t, k), c["muV"][k], rtol=1e-9, atol=1e-12)
    assert abs(o.beta_s() / float(c["beta_s"]) - 1) < 1e-12
    g = golden("wide_rank.npz").case("icm70")
    I, J = g["R"].shape
    pri = dict(alpha=1.0, beta=1.0, lambdaU=np.ones((I, 70)), lambdaV=np.ones((J, 70)))
    m = O.NMFICMOracle(g["R"], g["M"], 70, pri)
    m.U, m.V, m.tau = g["U0"].copy(), g["V0"].copy(), float(g["tau0"])
    m.run(6, minimum_TN=0.01)
    np.testing.assert_allclose(m.all_tau, g["all_tau"], rtol=1e-9)
    np.testing.assert_allclose(m.U, g["U"], rtol=1e-8, atol=1e-10)


def test_tri_oracle_matches_the_reference_at_ranks_above_64(golden):
    """tests/golden/wide_tri.npz (round 6): the reference's conditional parameters of a tri-factorisation with K = 70 and / or
    L = 66, and its nmtf_icm trajectories there -- what pins the oracle where the device runs blocks of S (tests/test_wide_tri_gpu.py)."""
    for tag in ("k70l5", "k6l66", "k70l66"):
        c = golden("wide_tri.npz").case(tag)
        K, L = c["S"].shape
        pri = dict(alpha=float(c["alpha"]), beta=float(c["beta"]), lambdaF=c["lambdaF"], lambdaS=c["lambdaS"], lambdaG=c["lambdaG"])
        o = O.BNMTFGibbsOracle(c["R"], c["M"], K, L, pri)
        o.F, o.S, o.G, o.tau = c["F"].copy(), c["S"].copy(), c["G"].copy(), float(c["tau"])
        for k in (0, K - 1, K // 2):
            t = o.tauF(k)
            np.testing.assert_allclose(t, c["tauF"][k], rtol=1e-12)
            np.testing.assert_allclose(o.muF(t, k), c["muF"][k], rtol=1e-9, atol=1e-12)
        for l in (0, L - 1):
            t = o.tauG(l)
            np.testing.assert_allclose(t, c["tauG"][l], rtol=1e-12)
            np.testing.assert_allclose(o.muG(t, l), c["muG"][l], rtol=1e-9, atol=1e-12)
        for i, (k, l) in enumerate(c["kl"][:12]):
            t = o.tauS(int(k), int(l))
            assert abs(t / c["tauS"][i] - 1) < 1e-12 and abs(o.muS(t, int(k), int(l)) - c["muS"][i]) < 1e-9 * (1 + abs(c["muS"][i]))
        assert abs(o.beta_s() / float(c["beta_s"]) - 1) < 1e-12
    g = golden("wide_tri.npz").case("icm_k70l4")
    I, J = g["R"].shape; K, L = g["S0"].shape
    pri = dict(alpha=1.0, beta=1.0, lambdaF=0.1 * np.ones((I, K)), lambdaS=0.1 * np.ones((K, L)), lambdaG=0.1 * np.ones((J, L)))
    m = O.NMTFICMOracle(g["R"], g["M"], K, L, pri)
    m.F, m.S, m.G, m.tau = g["F0"].copy(), g["S0"].copy(), g["G0"].copy(), float(g["tau0"])
    m.run(int(g["iterations"]), minimum_TN=float(g["minimum_TN"]))
    np.testing.assert_allclose(m.all_tau, g["all_tau"], rtol=1e-9)
    np.testing.assert_allclose(m.F, g["F"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(m.S, g["S"], rtol=1e-8, atol=1e-10)


def test_vb_oracle_matches_the_reference_at_k70(golden):
    c = golden("wide_rank.npz").case("vb70")
    pri = dict(alpha=1.0, beta=1.0, lambdaU=c["lambdaU"], lambdaV=c["lambdaV"])
    o = O.BNMFVBOracle(c["R"], c["M"], 70, pri)
    o.initialise("exp")
    assert abs(o.exptau / float(c["exptau0"]) - 1) < 1e-12
    o.run(5)
    np.testing.assert_allclose(o.all_performances["MSE"], c["mse"], rtol=1e-9)
    np.testing.assert_allclose(o.all_exp_tau, c["exptau"], rtol=1e-9)
    np.testing.assert_allclose(o.expU, c["expU"], rtol=1e-7, atol=1e-12)
