"""BASELINE.json configs[1] at its full size on the device: BNMF Gibbs, synthetic R 4096 x 4096, K = 32, 10 % missing
(the 8-wave block shape of the on-chip sweep kernel is what this size selects).  The oracle does not finish here, so:
observed counts bit-exact, the conditional parameters of three columns of U and V against the reference's closed forms
(bnmf_gibbs_optimised.py:167-177) evaluated in NumPy fp64, the on-chip kernel against the generic kernel over the first
sweeps (same Philox counters), the Gram-identity metrics against the direct fp64 metric kernel, the chain on its way to
the noise floor."""
import numpy as np
import pytest

from bnmtf_amd import bnmf_gibbs_optimised
from bnmtf_amd.synthetic import generate_bnmf

pytestmark = pytest.mark.gpu

PRI = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)


def test_bnmf_gibbs_4096_k32_full_size():
    I = J = 4096; K = 32
    R, M, _, _ = generate_bnmf(I, J, K, 0.1, tau=1.0, seed_data=0, seed_mask=1)
    R = R.astype(np.float64); M = M.astype(np.float64)
    b = bnmf_gibbs_optimised(R, M, K, PRI, verbose=False, seed=0)
    tot, row, col = b.omega_counts()
    assert tot == I * J - int(0.1 * I * J) == 15099495                      # n_Omega of cfg2
    assert np.array_equal(row, M.sum(axis=1).astype(np.uint32)) and np.array_equal(col, M.sum(axis=0).astype(np.uint32))
    assert "rows[n=4096" in b.describe()
    np.random.seed(0); b.initialise("random")
    U, V, tau = b.U.copy(), b.V.copy(), b.tau
    # closed forms (fp64) of three columns of each factor
    res = M * (R - U @ V.T)
    for k in (0, 13, K - 1):
        t_ref = tau * (M @ (V[:, k] ** 2))
        m_ref = (-0.1 + tau * ((res @ V[:, k]) + U[:, k] * (M @ (V[:, k] ** 2)))) / t_ref
        t = b.tauU(k)
        np.testing.assert_allclose(t, t_ref, rtol=2e-6)
        scale = np.abs(tau * (np.abs(res) @ np.abs(V[:, k])) / t_ref).max()   # size of the cancelling terms
        assert np.abs(b.muU(t, k) - m_ref).max() < 2e-5 * scale
        t_ref = tau * (M.T @ (U[:, k] ** 2))
        m_ref = (-0.1 + tau * ((res.T @ U[:, k]) + V[:, k] * (M.T @ (U[:, k] ** 2)))) / t_ref
        t = b.tauV(k)
        np.testing.assert_allclose(t, t_ref, rtol=2e-6)
        scale = np.abs(tau * (np.abs(res).T @ np.abs(U[:, k])) / t_ref).max()
        assert np.abs(b.muV(t, k) - m_ref).max() < 2e-5 * scale
    del res
    # on-chip kernel == generic kernel: same candidates, same acceptance rule
    runs = {}
    for path in ("fast", "generic"):
        c = bnmf_gibbs_optimised(R, M, K, PRI, verbose=False, seed=0)
        c.U, c.V, c.tau = U.copy(), V.copy(), tau
        if path == "generic":
            c.set_sweep_path(False)
        c.run(2)
        runs[path] = (c.all_U.copy(), c.all_V.copy(), c.all_tau.copy(), np.array(c.all_performances["MSE"]))
        c.close()
    f, g = runs["fast"], runs["generic"]
    d0 = np.abs(f[0][0] - g[0][0]) / (np.abs(g[0][0]) + 1e-3)
    assert np.mean(d0 < 1e-3) > 0.995
    np.testing.assert_allclose(f[3], g[3], rtol=2e-3)
    np.testing.assert_allclose(f[2], g[2], rtol=2e-3)
    # the chain, and the metric identities against the direct kernel
    b.run(60, store_samples=False)
    mse = np.array(b.all_performances["MSE"])
    assert mse[0] > 100 * mse[-1] and 0.9 < mse[-1] < 4.0 and np.all(np.diff(mse[20:]) < 0)
    p = b.predict_while_running()
    assert abs(p["MSE"] - mse[-1]) < 1e-4 * mse[-1]
    assert abs(p["R^2"] - b.all_performances["R^2"][-1]) < 1e-5 and abs(p["Rp"] - b.all_performances["Rp"][-1]) < 1e-5
    assert np.isfinite(b.U).all() and np.isfinite(b.V).all() and b.U.min() >= 0 and b.V.min() >= 0
