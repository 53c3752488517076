"""BASELINE.json's other full-size configurations on the device, through size-independent properties (the oracle does not
finish at these sizes; the closed-form conditionals of a few entries are evaluated directly in NumPy fp64):
  cfg4  BNMTF Gibbs 4096 x 4096, K = L = 32   (bnmtf_gibbs_optimised.py:195-211)
  cfg5  BNMF VB     8192 x 8192, K = 64       (bnmf_vb_optimised.py:165-215)"""
import numpy as np
import pytest

from bnmtf_amd import bnmf_vb_optimised, bnmtf_gibbs_optimised
from bnmtf_amd.synthetic import generate_bnmf, generate_bnmtf

pytestmark = pytest.mark.gpu


def test_bnmtf_gibbs_4096_conditionals_and_chain():
    I = J = 4096; K = L = 32
    R, M, _, _, _ = generate_bnmtf(I, J, K, L, 0.1, tau=1.0, seed_data=0, seed_mask=1)
    R = R.astype(np.float64); M = M.astype(np.float64)
    pri = dict(alpha=1., beta=1., lambdaF=0.1, lambdaS=0.1, lambdaG=0.1)
    b = bnmtf_gibbs_optimised(R, M, K, L, pri, verbose=False, seed=3)
    np.random.seed(1); b.initialise("random", "random")
    F, S, G, tau = b.F.copy(), b.S.copy(), b.G.copy(), b.tau
    # closed-form conditional parameters of a few S entries, F columns and G columns at the full size
    res = M * (R - F @ S @ G.T)
    for (k, l) in [(0, 0), (5, 17), (31, 31)]:
        fg2 = (F[:, k] ** 2) @ M @ (G[:, l] ** 2)
        t_ref = tau * fg2
        m_ref = (-0.1 + tau * (F[:, k] @ res @ G[:, l] + S[k, l] * fg2)) / t_ref
        t = b.tauS(k, l)
        assert abs(t - t_ref) < 2e-5 * t_ref
        assert abs(b.muS(t, k, l) - m_ref) < 2e-4 * (abs(m_ref) + 1.0 / np.sqrt(t_ref))
    for k in (0, 13):
        sg = S[k] @ G.T
        t_ref = tau * (M * sg ** 2).sum(axis=1)
        m_ref = (-0.1 + tau * ((res + M * np.outer(F[:, k], sg)) * sg).sum(axis=1)) / t_ref
        t = b.tauF(k)
        np.testing.assert_allclose(t, t_ref, rtol=2e-5)
        assert np.abs(b.muF(t, k) - m_ref).max() < 2e-4 * np.abs(m_ref).max()
    for l in (7,):
        fs = F @ S[:, l]
        t_ref = tau * (M.T * fs ** 2).T.sum(axis=0)
        m_ref = (-0.1 + tau * ((res + M * np.outer(fs, G[:, l])).T * fs).T.sum(axis=0)) / t_ref
        t = b.tauG(l)
        np.testing.assert_allclose(t, t_ref, rtol=2e-5)
        assert np.abs(b.muG(t, l) - m_ref).max() < 2e-4 * np.abs(m_ref).max()
    # the chain: towards the noise floor, non-negative finite draws, metric identities vs the direct kernel
    b.run(40, store_samples=False)
    mse = np.array(b.all_performances["MSE"])
    assert mse[-1] < mse[0] / 50 and np.all(np.isfinite(mse))
    p = b.predict_while_running()
    assert abs(p["MSE"] - mse[-1]) < 2e-4 * mse[-1]
    for X in (b.F, b.S, b.G):
        assert np.isfinite(X).all() and X.min() >= 0


def test_bnmf_vb_8192_fixed_point_properties():
    I = J = 8192; K = 64
    R, M, _, _ = generate_bnmf(I, J, K, 0.1, tau=1.0, seed_data=0, seed_mask=1)
    b = bnmf_vb_optimised(R, M, K, dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1), verbose=False)
    b.initialise("exp")
    b.run(12)
    assert "vb_sweep=masked" in b.describe()      # this configuration takes the on-chip sweep with the masked sums from the matrix cores by itself
    elbo = np.array(b.all_elbo); mse = np.array(b.all_performances["MSE"])
    # coordinate ascent: the bound does not go down (fp32 storage of the factors: allow 1e-7 of its size).  The bound is
    # -inf, as in the reference, while some entry has mu sqrt(tau) < -37.5 (log of an underflown erfc,
    # bnmf_vb_optimised.py:160-163) -- at 2 x 524 288 entries with tau_ik ~ 1e4 that is the normal state
    fin = np.isfinite(elbo)
    assert not np.isnan(elbo).any() and not (elbo == np.inf).any()
    assert np.all(np.diff(elbo[fin]) > -1e-7 * np.abs(elbo[fin][1:]))
    assert mse[-1] < mse[0] / 20 and np.all(np.diff(mse) < 0)
    # the Gram-identity exp_square_diff used inside run() against the direct fp64 kernel over all observed entries
    esd = b.exp_square_diff()
    assert abs(b.beta_s - (1. + 0.5 * esd)) < 5e-5 * b.beta_s
    assert abs(b.alpha_s - (1. + 0.5 * M.sum())) < 1e-9 * b.alpha_s
    for X in (b.expU, b.expV, b.varU, b.varV, b.tauU, b.tauV):
        assert np.isfinite(X).all() and X.min() >= 0
    # moments are those of the stored (mu, tau): a sample of entries against the oracle's formula
    from oracle import bnmtf_oracle as O
    sl = (slice(0, 8192, 257), slice(None))
    np.testing.assert_allclose(b.expU[sl], O.tn_expectation(b.muU[sl], b.tauU[sl]), rtol=2e-6, atol=1e-30)
    np.testing.assert_allclose(b.varU[sl], O.tn_variance(b.muU[sl], b.tauU[sl]), rtol=2e-4, atol=1e-30)
