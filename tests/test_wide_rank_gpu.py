"""Ranks above 64 on the device (round 6): the reference takes any K (code/models/bnmf_gibbs_optimised.py:54-78); the device runs a
wider model as column blocks of at most 64 (bnmtf_amd/_blocked.py; csrc: bnmf_half_sweep, bnmf_set_residual_data,
bnmf_set_column_block, bnmtf_metric_sums_wide).  Checked against the reference's own numbers at K = 96 and K = 70
(tests/golden/wide_rank.npz, made by tests/golden/make_golden.py from the reference) and against the oracle."""
import numpy as np
import pytest

from bnmtf_amd import bnmf_gibbs_optimised, nmf_icm
from oracle import bnmtf_oracle as O

pytestmark = pytest.mark.gpu


def _pri(c):
    return dict(alpha=float(c["alpha"]), beta=float(c["beta"]), lambdaU=c["lambdaU"], lambdaV=c["lambdaV"])


def test_conditional_parameters_at_k96_match_the_reference(golden):
    c = golden("wide_rank.npz").case("k96")
    K = int(c["K"])
    b = bnmf_gibbs_optimised(c["R"], c["M"], K, _pri(c), verbose=False)
    assert "column blocks [(0, 64), (64, 96)]" in b.describe()
    b.U, b.V, b.tau = c["U"].copy(), c["V"].copy(), float(c["tau"])
    for k in (0, 1, 31, 63, 64, 65, 95):                  # both blocks, both sides of the cut
        tU = b.tauU(k); tV = b.tauV(k)
        np.testing.assert_allclose(tU, c["tauU"][k], rtol=2e-6)
        np.testing.assert_allclose(tV, c["tauV"][k], rtol=2e-6)
        # mu: absolute, in units of the cancelling terms of the numerator (fp32 contractions, as at K <= 64)
        sU = np.abs(c["muU"][k]).max() + 1.0; sV = np.abs(c["muV"][k]).max() + 1.0
        assert np.abs(b.muU(c["tauU"][k], k) - c["muU"][k]).max() < 1e-4 * sU
        assert np.abs(b.muV(c["tauV"][k], k) - c["muV"][k]).max() < 1e-4 * sV
    assert abs(b.beta_s() / float(c["beta_s"]) - 1) < 1e-6
    p = b.predict_while_running()
    np.testing.assert_allclose([p["MSE"], p["R^2"], p["Rp"]], c["perf"], rtol=1e-6)
    tot, row, col = b.omega_counts()
    assert tot == int(c["M"].sum()) and (row == c["M"].sum(axis=1)).all() and (col == c["M"].sum(axis=0)).all()
    b.close()


def test_icm_trajectory_at_k70_matches_the_reference(golden):
    """nmf_icm (deterministic): six iterations with the minimum_TN clamp, 64 + 6 columns."""
    c = golden("wide_rank.npz").case("icm70")
    I, J = c["R"].shape; K = 70
    pri = dict(alpha=1.0, beta=1.0, lambdaU=np.ones((I, K)), lambdaV=np.ones((J, K)))
    b = nmf_icm(c["R"], c["M"], K, pri, verbose=False)
    b.U, b.V, b.tau = c["U0"].copy(), c["V0"].copy(), float(c["tau0"])
    b.run(6, minimum_TN=0.01)
    np.testing.assert_allclose(b.all_tau, c["all_tau"], rtol=1e-3)
    np.testing.assert_allclose(b.all_performances["MSE"], c["mse"], rtol=1e-3)
    assert np.abs(b.U - c["U"]).max() < 5e-3 * np.abs(c["U"]).max() and np.abs(b.V - c["V"]).max() < 5e-3 * np.abs(c["V"]).max()
    assert (b.U >= 0.01 - 1e-7).all()
    b.close()


@pytest.mark.parametrize("I,J,K", [(150, 120, 96), (300, 260, 130)])
def test_mode_trajectory_and_draws_follow_the_oracle(I, J, K):
    """Mode updates (deterministic) over four iterations against the fp64 oracle; draws: the first sweep of the oracle's Philox
    chain element-wise -- the candidate streams are keyed by the WIDE model's column index, so columns 64 ... draw what the
    oracle draws for them."""
    rs = np.random.RandomState(K)
    R = rs.exponential(1.0, (I, 10)) @ rs.exponential(1.0, (J, 10)).T + rs.randn(I, J)
    M = (rs.uniform(size=(I, J)) >= 0.15).astype(float); M[np.arange(I), rs.randint(0, J, I)] = 1; M[rs.randint(0, I, J), np.arange(J)] = 1
    pri = dict(alpha=1., beta=1., lambdaU=0.5, lambdaV=0.5)
    U0 = rs.exponential(0.3, (I, K)); V0 = rs.exponential(0.3, (J, K))
    b = bnmf_gibbs_optimised(R, M, K, pri, verbose=False, seed=21)
    b.U, b.V, b.tau = U0.copy(), V0.copy(), 1.3
    b.run(4, update="mode")
    o = O.BNMFGibbsOracle(R, M, K, pri, seed=21)
    o.U, o.V, o.tau = U0.copy(), V0.copy(), 1.3
    o.run(4, draw=False)
    sU = max(1.0, np.abs(o.all_U[0]).max()); sV = max(1.0, np.abs(o.all_V[0]).max())
    assert np.abs(b.all_U[0] - o.all_U[0]).max() < 5e-4 * sU and np.abs(b.all_V[0] - o.all_V[0]).max() < 5e-4 * sV
    np.testing.assert_allclose(b.all_performances["MSE"], o.all_performances["MSE"], rtol=5e-4)
    np.testing.assert_allclose(b.all_tau, o.all_tau, rtol=5e-4)           # (the mean of the Gamma in mode updates, not a draw)
    # draws
    b2 = bnmf_gibbs_optimised(R, M, K, pri, verbose=False, seed=21)
    b2.U, b2.V, b2.tau = U0.copy(), V0.copy(), 1.3
    b2.run(3)
    o2 = O.BNMFGibbsOracle(R, M, K, pri, seed=21)
    o2.U, o2.V, o2.tau = U0.copy(), V0.copy(), 1.3
    o2.run(1)
    for dev, ora in ((b2.all_U[0], o2.all_U[0]), (b2.all_V[0], o2.all_V[0])):
        d0 = np.abs(dev - ora) / (1e-3 + np.abs(ora))
        assert np.mean(d0 < 1e-3) > 0.99
        for c0 in range(0, K, 64):                         # every block by itself, too
            assert np.mean(d0[:, c0:c0 + 64] < 1e-3) > 0.98
    assert abs(b2.all_tau[0] / o2.all_tau[0] - 1) < 1e-3
    assert abs(b2.all_performances["MSE"][0] / o2.all_performances["MSE"][0] - 1) < 1e-3
    b.close(); b2.close()


def test_post_run_api_and_device_side_expectation_of_a_wide_model():
    I, J, K = 120, 90, 80
    rs = np.random.RandomState(3)
    R = rs.exponential(1.0, (I, 5)) @ rs.exponential(1.0, (J, 5)).T + rs.randn(I, J)
    M = (rs.uniform(size=(I, J)) >= 0.1).astype(float); M[:, 0] = 1; M[0, :] = 1
    pri = dict(alpha=1., beta=1., lambdaU=1.0, lambdaV=1.0)
    np.random.seed(1)
    b = bnmf_gibbs_optimised(R, M, K, pri, verbose=False, seed=4)
    b.initialise("random")
    b.run(12)
    assert b.all_U.shape == (12, I, K) and b.all_V.shape == (12, J, K) and len(b.all_tau) == 12
    assert b.all_performances["MSE"][-1] < b.all_performances["MSE"][0]
    eU, eV, et = b.approx_expectation(4, 2)
    q = b.quality("loglikelihood", 4, 2); p = b.predict(1 - M, 4, 2)
    assert np.isfinite(q) and np.isfinite(p["MSE"])
    # the same chain with the posterior sums kept beside the run instead of the samples (what the model-selection drivers ask for)
    np.random.seed(1)
    c = bnmf_gibbs_optimised(R, M, K, pri, verbose=False, seed=4)
    c.initialise("random")
    c.run(12, store_samples=False, expectation=(4, 2))
    fU, fV, ft = c.approx_expectation(4, 2)
    np.testing.assert_allclose(fU, eU, rtol=1e-5, atol=1e-6); np.testing.assert_allclose(fV, eV, rtol=1e-5, atol=1e-6)
    assert abs(ft / et - 1) < 1e-9
    assert abs(c.quality("loglikelihood", 4, 2) / q - 1) < 1e-6
    b.close(); c.close()


def test_vb_trajectory_at_k70_matches_the_reference(golden):
    """bnmf_vb_optimised beyond 64 columns (deterministic from init='exp'): five iterations against the reference's own numbers
    (tests/golden/wide_rank.npz: vb70) -- MSE, exptau, the ELBO, the moments -- and single updates through the class hooks."""
    from bnmtf_amd import bnmf_vb_optimised
    c = golden("wide_rank.npz").case("vb70")
    K = 70
    pri = dict(alpha=1.0, beta=1.0, lambdaU=c["lambdaU"], lambdaV=c["lambdaV"])
    b = bnmf_vb_optimised(c["R"], c["M"], K, pri, verbose=False)
    b.initialise("exp")
    assert abs(b.exptau / float(c["exptau0"]) - 1) < 1e-6
    b.run(5)
    np.testing.assert_allclose(b.all_performances["MSE"], c["mse"], rtol=1e-3)
    np.testing.assert_allclose(b.all_exp_tau, c["exptau"], rtol=1e-3)
    assert np.isneginf(b.all_elbo[0]) == np.isneginf(c["elbo"][0])
    np.testing.assert_allclose(b.all_elbo[1:], c["elbo"][1:], rtol=1e-3)
    for name in ("expU", "expV", "varU", "tauU", "muV"):
        ref = c[name]
        assert np.abs(getattr(b, name) - ref).max() < 3e-3 * max(1e-3, np.abs(ref).max()), name
    # single updates on the second block, against the oracle from the state the run left
    o = O.BNMFVBOracle(c["R"], c["M"], K, pri)
    for n in ("muU", "tauU", "expU", "varU", "muV", "tauV", "expV", "varV"):
        setattr(o, n, getattr(b, n).copy())
    o.exptau = b.exptau
    o.update_U(66); b.update_U(66)
    np.testing.assert_allclose(b.tauU[:, 66], o.tauU[:, 66], rtol=5e-6)
    assert np.abs(b.muU[:, 66] - o.muU[:, 66]).max() < 1e-4 * (1.0 + np.abs(o.muU[:, 66]).max())
    assert abs(b.exp_square_diff() / o.exp_square_diff() - 1) < 1e-6
    b.close()
