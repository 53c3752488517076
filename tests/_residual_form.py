"""The reference's column loops restated on the masked residual, NumPy fp64 -- the checker the full-size parity tests use
(tests/test_full_size_parity_gpu.py), where the as-written oracle (oracle/bnmtf_oracle.py: a pass over R per column) would take
minutes per iteration.  tests/test_residual_form_cpu.py pins these forms to the as-written oracle, which the reference's own
vectors pin (tests/test_oracle_golden.py).

E = M (R - prediction) is kept current by rank-one updates, so every column sees the new values of the columns before it:
bnmf_gibbs_optimised.py:167-177 (tauU/muU, tauV/muV), bnmtf_gibbs_optimised.py:195-211 (F, S, G), mode update = nmf_icm.py:124-134."""
import numpy as np


def mode_sweep(E, X, Y, Mm, tau, lam):
    """columns of X given Y, mode update; E = M (R - X Y^T) is (rows of X) x (rows of Y) and kept current"""
    for k in range(X.shape[1]):
        a = Mm @ (Y[:, k] ** 2)
        num = E @ Y[:, k] + X[:, k] * a
        with np.errstate(divide="ignore", invalid="ignore"):
            mu = (-lam + tau * num) / (tau * a)
        new = np.where(a > 0, np.maximum(mu, 0.0), 0.0)
        E -= Mm * np.outer(new - X[:, k], Y[:, k])
        X[:, k] = new


def s_step_mode(E, F, S, G, Mm, tau, lam, k, l):
    """one entry of S (bnmtf_gibbs_optimised.py:201-205), mode update: a = sum M (F_k^2 x G_l^2), num = F_k^T E G_l + S_kl a;
    returns the new value, E and S updated in place"""
    a = (F[:, k] ** 2) @ (Mm @ (G[:, l] ** 2))
    num = F[:, k] @ (E @ G[:, l]) + S[k, l] * a
    new = max((-lam + tau * num) / (tau * a), 0.0) if a > 0 else 0.0
    E -= Mm * ((new - S[k, l]) * np.outer(F[:, k], G[:, l]))
    S[k, l] = new
    return new
