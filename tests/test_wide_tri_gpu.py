"""The tri-factorisations with K or L above 64 (the reference takes any rank: bnmtf_gibbs_optimised.py:56-84, nmtf_icm.py), run
as blocks (bnmtf_amd/_blocked.py: TriBlocks; csrc: bnmtf_set_s_block, bnmtf_s_rows, bnmf_set_residual_data with a BNMTF target):
F's and G's column blocks are BNMF models against the effective factors, block (b, c) of S a BNMTF model on the data minus what
the other blocks of S explain.  Against the reference's own numbers (tests/golden/wide_tri.npz) and the fp64 oracle."""
import numpy as np
import pytest

from bnmtf_amd import bnmtf_gibbs_optimised, nmtf_icm
from oracle import bnmtf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["k70l5", "k6l66", "k70l66"])
def test_conditional_parameters_match_the_reference(golden, tag):
    """tauF / muF (:195-199), tauS / muS (:201-205), tauG / muG (:207-211), beta_s and the metrics of a random state."""
    c = golden("wide_tri.npz").case(tag)
    K, L = c["S"].shape
    pri = dict(alpha=float(c["alpha"]), beta=float(c["beta"]), lambdaF=c["lambdaF"], lambdaS=c["lambdaS"], lambdaG=c["lambdaG"])
    b = bnmtf_gibbs_optimised(c["R"], c["M"], K, L, pri, verbose=False, seed=1)
    b.F, b.S, b.G, b.tau = c["F"].copy(), c["S"].copy(), c["G"].copy(), float(c["tau"])
    assert "blocks of F" in b.describe()
    for k in sorted({0, 63, 64, K - 1} & set(range(K))):
        t = b.tauF(k)
        np.testing.assert_allclose(t, c["tauF"][k], rtol=5e-6)
        assert np.abs(b.muF(t, k) - c["muF"][k]).max() < 5e-5 * (np.abs(c["muF"][k]).max() + 1.0)
    for l in sorted({0, 63, 64, L - 1} & set(range(L))):
        t = b.tauG(l)
        np.testing.assert_allclose(t, c["tauG"][l], rtol=5e-6)
        assert np.abs(b.muG(t, l) - c["muG"][l]).max() < 5e-5 * (np.abs(c["muG"][l]).max() + 1.0)
    for i, (k, l) in enumerate(c["kl"][:16]):
        t = b.tauS(int(k), int(l))
        assert abs(t / c["tauS"][i] - 1) < 5e-6
        assert abs(b.muS(t, int(k), int(l)) - c["muS"][i]) < 5e-5 * (np.abs(c["muS"]).max() + 1.0)
    assert abs(b.beta_s() / float(c["beta_s"]) - 1) < 1e-6
    p = b.predict_while_running()
    np.testing.assert_allclose([p["MSE"], p["R^2"], p["Rp"]], c["perf"], rtol=1e-6)
    b.close()


@pytest.mark.parametrize("tag", ["icm_k70l4", "icm_k70l66"])
def test_icm_trajectories_match_the_reference(golden, tag):
    """nmtf_icm.py:132-173 end to end (deterministic): one column block of S (a step = a row block) and two by two blocks
    (a step = one row of one block)."""
    g = golden("wide_tri.npz").case(tag)
    I, J = g["R"].shape; K, L = g["S0"].shape
    pri = dict(alpha=1.0, beta=1.0, lambdaF=0.1 * np.ones((I, K)), lambdaS=0.1 * np.ones((K, L)), lambdaG=0.1 * np.ones((J, L)))
    m = nmtf_icm(g["R"], g["M"], K, L, pri, verbose=False)
    m.F, m.S, m.G, m.tau = g["F0"].copy(), g["S0"].copy(), g["G0"].copy(), float(g["tau0"])
    m.run(int(g["iterations"]), minimum_TN=float(g["minimum_TN"]))
    np.testing.assert_allclose(m.all_tau, g["all_tau"], rtol=2e-3)
    np.testing.assert_allclose(m.all_tau[:2], g["all_tau"][:2], rtol=1e-4)
    np.testing.assert_allclose(m.all_performances["MSE"], g["mse"], rtol=2e-3)
    for got, ref in ((m.F, g["F"]), (m.S, g["S"]), (m.G, g["G"])):
        assert np.abs(got - ref).max() < 2e-2 * np.abs(ref).max()
    assert abs(m.quality("MSE") / g["mse"][-1] - 1) < 2e-3
    m.close()


def test_draws_follow_the_oracles_chain_and_a_narrow_model_agrees_with_its_blocks():
    """Gibbs draws at K = 70, L = 66: the same Philox keys as the oracle's chain (rows / columns of F and G by their wide column
    index, S_kl by k L + l), so the first sweeps agree element-wise but for decisions on a rounding boundary; mode updates
    (deterministic) follow the oracle closely for several iterations."""
    rs = np.random.RandomState(9)
    I, J, K, L = 46, 41, 70, 66
    R = rs.exponential(1.0, (I, 5)) @ rs.exponential(1.0, (5, 4)) @ rs.exponential(1.0, (J, 4)).T + rs.normal(0, 1, (I, J))
    M = (rs.rand(I, J) >= 0.15).astype(float)
    M[np.arange(I), rs.randint(0, J, I)] = 1.0; M[rs.randint(0, I, J), np.arange(J)] = 1.0
    pri = dict(alpha=1.0, beta=1.0, lambdaF=0.5, lambdaS=0.5, lambdaG=0.5)
    F0 = rs.exponential(0.4, (I, K)); S0 = rs.exponential(0.4, (K, L)); G0 = rs.exponential(0.4, (J, L))
    b = bnmtf_gibbs_optimised(R, M, K, L, pri, verbose=False, seed=31)
    b.F, b.S, b.G, b.tau = F0.copy(), S0.copy(), G0.copy(), 0.9
    b.run(2)
    o = O.BNMTFGibbsOracle(R, M, K, L, pri, seed=31)
    o.F, o.S, o.G, o.tau = F0.copy(), S0.copy(), G0.copy(), 0.9
    o.run(2)
    for dev, ora in ((b.all_F[0], o.all_F[0]), (b.all_S[0], o.all_S[0]), (b.all_G[0], o.all_G[0])):
        d = np.abs(dev - ora) / (1e-3 + np.abs(ora))
        assert np.mean(d < 2e-3) > 0.98
    assert abs(b.all_tau[0] / o.all_tau[0] - 1) < 2e-3
    np.testing.assert_allclose(b.all_performances["MSE"][:1], o.all_performances["MSE"][:1], rtol=2e-3)
    b.close()
    b = bnmtf_gibbs_optimised(R, M, K, L, pri, verbose=False, seed=31)
    b.F, b.S, b.G, b.tau = F0.copy(), S0.copy(), G0.copy(), 0.9
    b.run(3, update="mode")
    o = O.BNMTFGibbsOracle(R, M, K, L, pri, seed=31)
    o.F, o.S, o.G, o.tau = F0.copy(), S0.copy(), G0.copy(), 0.9
    o.run(3, draw=False)
    np.testing.assert_allclose(b.all_performances["MSE"], o.all_performances["MSE"], rtol=2e-3)
    np.testing.assert_allclose(b.all_tau, o.all_tau, rtol=5e-4)           # (mode updates take the Gamma's mean: tools/r06/fuzz_wide_tri.py found a draw here)
    assert np.abs(b.all_S[0] - o.all_S[0]).max() < 2e-3 * np.abs(o.all_S[0]).max()
    assert np.abs(b.all_F[0] - o.all_F[0]).max() < 2e-3 * np.abs(o.all_F[0]).max()
    # the posterior means accumulated by a blocked run, and predict() on a held-out mask
    b.run(4, store_samples=False, expectation=(1, 1))
    eF, eS, eG, et = b.approx_expectation(1, 1)
    assert eF.shape == (I, K) and eS.shape == (K, L) and eG.shape == (J, L) and et > 0
    p = b.predict(1.0 - M, 1, 1)
    assert np.isfinite(p["MSE"])
    b.close()
