"""Edge shapes of the hot path on the device against the oracle (mode updates: every step deterministic):
a mask with nothing missing (no slots at all: the sweeps reduce to the Gram terms), rank one, a single row / a single
column, and a mask with most entries missing (slot rows beyond what the on-chip kernels take: the generic kernel)."""
import numpy as np
import pytest

from bnmtf_amd import bnmf_gibbs_optimised, bnmf_vb_optimised, bnmtf_gibbs_optimised
from oracle import bnmtf_oracle as O

pytestmark = pytest.mark.gpu

PRI2 = dict(alpha=1., beta=1., lambdaU=0.3, lambdaV=0.2)
PRI3 = dict(alpha=1., beta=1., lambdaF=0.3, lambdaS=0.2, lambdaG=0.1)


def _data(I, J, K, frac_missing, seed):
    rs = np.random.RandomState(seed)
    R = rs.exponential(1.0, (I, K)) @ rs.exponential(1.0, (J, K)).T + 0.3 * rs.randn(I, J)
    M = (rs.rand(I, J) >= frac_missing).astype(float)
    M[rs.randint(I, size=J), np.arange(J)] = 1; M[np.arange(I), rs.randint(J, size=I)] = 1      # no empty row / column
    return R, M, rs


@pytest.mark.parametrize("handover", ["0", "1"], ids=["prepass", "handover"])
@pytest.mark.parametrize("I,J,K,miss", [(70, 90, 6, 0.0), (45, 33, 1, 0.2), (1, 40, 3, 0.1), (50, 1, 2, 0.0), (160, 1500, 5, 0.9)])
def test_bnmf_gibbs_edge_shapes(monkeypatch, I, J, K, miss, handover):
    """handover = "1": q handed over between the half sweeps wherever its tables can be built -- with nothing missing
    the regions are empty, with 90 % missing the units need more slot rows than the on-chip kernels take and the hand-over has to
    stay off by itself."""
    monkeypatch.setenv("BNMTF_HANDOVER", handover)
    R, M, rs = _data(I, J, K, miss, 3)
    U0, V0 = rs.exponential(1.0, (I, K)), rs.exponential(1.0, (J, K))
    o = O.BNMFGibbsOracle(R, M, K, PRI2)
    o.U, o.V, o.tau = U0.copy(), V0.copy(), 0.8
    o.run(4, draw=False)
    b = bnmf_gibbs_optimised(R, M, K, PRI2, verbose=False, seed=1)
    if handover == "0" or miss >= 0.9:
        assert "handover=0" in b.describe()
    b.U, b.V, b.tau = U0.copy(), V0.copy(), 0.8
    b.run(4, update='mode')
    # the masked SSE comes from Gram identities in fp32 products: its error scales with the terms that cancel (sum R^2), not
    # with the SSE -- which a single row fitted by three factors drives to ~1e-4 of them
    scale = (M * R ** 2).sum() / M.sum()
    np.testing.assert_allclose(b.all_performances['MSE'], o.all_performances['MSE'], rtol=5e-4, atol=2e-6 * scale)
    np.testing.assert_allclose(1.0 / np.asarray(b.all_tau), 1.0 / np.asarray(o.all_tau), rtol=5e-4, atol=2e-6 * scale)
    for x, y in ((b.all_U[-1], o.all_U[-1]), (b.all_V[-1], o.all_V[-1])):
        assert np.abs(np.asarray(x) - np.asarray(y)).max() <= 2e-3 * max(1.0, np.abs(y).max())
    b.run(3)                                        # and draws stay finite and non-negative on these shapes
    assert np.isfinite(b.U).all() and np.isfinite(b.V).all() and b.U.min() >= 0 and b.V.min() >= 0


@pytest.mark.parametrize("I,J,K,miss", [(60, 75, 4, 0.0), (40, 52, 1, 0.15)])
def test_bnmf_vb_edge_shapes(I, J, K, miss):
    R, M, rs = _data(I, J, K, miss, 5)
    o = O.BNMFVBOracle(R, M, K, PRI2)
    b = bnmf_vb_optimised(R, M, K, PRI2, verbose=False)
    np.random.seed(2); o.initialise('exp')
    np.random.seed(2); b.initialise('exp')
    o.run(8); b.run(8)
    np.testing.assert_allclose(b.all_exp_tau, o.all_exp_tau, rtol=2e-3)
    np.testing.assert_allclose(b.all_performances['MSE'], o.all_performances['MSE'], rtol=2e-3)


@pytest.mark.parametrize("I,J,K,L,miss", [(55, 64, 3, 4, 0.0), (48, 41, 1, 1, 0.2), (30, 35, 1, 5, 0.1)])
def test_bnmtf_gibbs_edge_shapes(I, J, K, L, miss):
    rs = np.random.RandomState(9)
    R = rs.exponential(1.0, (I, K)) @ rs.exponential(1.0, (K, L)) @ rs.exponential(1.0, (J, L)).T + 0.3 * rs.randn(I, J)
    M = (rs.rand(I, J) >= miss).astype(float)
    M[rs.randint(I, size=J), np.arange(J)] = 1; M[np.arange(I), rs.randint(J, size=I)] = 1
    F0, S0, G0 = rs.exponential(1.0, (I, K)), rs.exponential(1.0, (K, L)), rs.exponential(1.0, (J, L))
    o = O.BNMTFGibbsOracle(R, M, K, L, PRI3)
    o.F, o.S, o.G, o.tau = F0.copy(), S0.copy(), G0.copy(), 0.6
    o.run(4, draw=False)
    b = bnmtf_gibbs_optimised(R, M, K, L, PRI3, verbose=False, seed=1)
    b.F, b.S, b.G, b.tau = F0.copy(), S0.copy(), G0.copy(), 0.6
    b.run(4, update='mode')
    scale = (M * R ** 2).sum() / M.sum()
    np.testing.assert_allclose(b.all_performances['MSE'], o.all_performances['MSE'], rtol=1e-3, atol=2e-6 * scale)
    np.testing.assert_allclose(1.0 / np.asarray(b.all_tau), 1.0 / np.asarray(o.all_tau), rtol=1e-3, atol=2e-6 * scale)
    assert np.abs(np.asarray(b.all_S[-1]) - np.asarray(o.all_S[-1])).max() <= 3e-3 * max(1.0, np.abs(o.all_S[-1]).max())
    b.run(3)
    assert np.isfinite(b.S).all() and b.S.min() >= 0
