"""The residual-form restatement the full-size GPU parity tests check against (tests/_residual_form.py) equals the as-written
oracle (oracle/bnmtf_oracle.py, itself pinned by the reference's vectors in tests/test_oracle_golden.py) -- fp64, small shapes."""
import numpy as np

from oracle import bnmtf_oracle as O
from _residual_form import mode_sweep, s_step_mode

LAM = 0.1


def _data(I, J, seed, frac=0.2):
    rs = np.random.RandomState(seed)
    R = rs.exponential(1.0, (I, 3)) @ rs.exponential(1.0, (3, J)) + 0.1 * rs.rand(I, J)
    M = (rs.rand(I, J) > frac).astype(np.float64)
    M[0, :] = 1.0; M[:, 1] = 1.0; M[np.arange(min(I, J)), np.arange(min(I, J))] = 1.0      # (the constructor refuses empty rows / columns)
    return R, M, rs


def test_bnmf_mode_iteration_equals_the_as_written_oracle():
    I, J, K = 23, 17, 5
    R, M, rs = _data(I, J, 0)
    o = O.BNMFGibbsOracle(R, M, K, dict(alpha=1., beta=1., lambdaU=LAM, lambdaV=LAM), seed=0)
    o.U, o.V, o.tau = rs.exponential(1.0, (I, K)), rs.exponential(1.0, (J, K)), 0.7
    U, V, tau = o.U.copy(), o.V.copy(), 0.7
    for _ in range(3):
        E = M * (R - U @ V.T)
        mode_sweep(E, U, V, M, tau, LAM)
        Et = np.ascontiguousarray(E.T)
        mode_sweep(Et, V, U, np.ascontiguousarray(M.T), tau, LAM)
        tau = (1. + M.sum() / 2.) / (1. + 0.5 * (Et ** 2).sum())
        o.sweep(draw=False)
        np.testing.assert_allclose(U, o.U, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(V, o.V, rtol=1e-9, atol=1e-12)
        assert (U > 0).mean() > 0.3 and (V > 0).mean() > 0.3
        assert abs(tau - o.tau) <= 1e-10 * o.tau


def test_bnmtf_mode_iteration_equals_the_as_written_oracle():
    I, J, K, L = 19, 16, 4, 3
    R, M, rs = _data(I, J, 1)
    o = O.BNMTFGibbsOracle(R, M, K, L, dict(alpha=1., beta=1., lambdaF=LAM, lambdaS=LAM, lambdaG=LAM), seed=0)
    o.F, o.S, o.G, o.tau = rs.exponential(1.0, (I, K)), rs.exponential(1.0, (K, L)), rs.exponential(1.0, (J, L)), 0.7
    F, S, G, tau = o.F.copy(), o.S.copy(), o.G.copy(), 0.7
    for _ in range(3):
        Veff = G @ S.T
        E = M * (R - F @ Veff.T)
        mode_sweep(E, F, Veff, M, tau, LAM)
        for k in range(K):
            for l in range(L):
                s_step_mode(E, F, S, G, M, tau, LAM, k, l)
        Ueff = F @ S
        Et = np.ascontiguousarray((M * (R - Ueff @ G.T)).T)
        mode_sweep(Et, G, Ueff, np.ascontiguousarray(M.T), tau, LAM)
        tau = (1. + M.sum() / 2.) / (1. + 0.5 * (Et ** 2).sum())
        o.sweep(draw=False)
        np.testing.assert_allclose(F, o.F, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(S, o.S, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(G, o.G, rtol=1e-9, atol=1e-12)
        assert (F > 0).mean() > 0.3 and (S > 0).mean() > 0.3 and (G > 0).mean() > 0.3
        assert abs(tau - o.tau) <= 1e-10 * o.tau
