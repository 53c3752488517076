"""Whole-iteration parity at BASELINE.json's full sizes, for the configurations that had only spot checks (cfg3 has
tests/test_wide_sweep_gpu.py::test_headline_shape_whole_sweep_against_fp64_closed_forms): the kernels the bench times, one
iteration in a deterministic update, against the reference's column loops restated on the masked residual in NumPy fp64
(E = M (R - prediction) kept current by rank-one updates, so every column sees the new values of the columns before it).

  cfg2  BNMF Gibbs 4096 x 4096, K = 32        bnmf_gibbs_optimised.py:134-142, :167-177   both half sweeps, tau, MSE
  cfg5  BNMF VB    8192 x 8192, K = 64        bnmf_vb_optimised.py:121-153, :181-215      update_U/V + moments of all 2 x 64 columns, exptau, ELBO pieces
  cfg4  BNMTF Gibbs 4096 x 4096, K = L = 32   bnmtf_gibbs_optimised.py:152-167, :195-211  F sweep, the first 64 steps of the S chain, G sweep

The residual form lives in tests/_residual_form.py and is itself pinned: tests/test_residual_form_cpu.py holds it to the as-written
oracle (oracle/bnmtf_oracle.py, which the reference's vectors pin) at 1e-9 on small shapes; oracle.BNMFGibbsFairCPU is the same
form with draws.

Tolerances: factors 5e-4 of their scale (fp32 contractions against fp64), masked MSE / exptau 3e-4 relative -- the MSE-identity
tolerance stated in DESIGN.md section 5."""
import numpy as np
import pytest

from bnmtf_amd import bnmf_gibbs_optimised, bnmf_vb_optimised, bnmtf_gibbs_optimised
from bnmtf_amd.synthetic import generate_bnmf, generate_bnmtf
from oracle import bnmtf_oracle as O
from _residual_form import mode_sweep, s_step_mode

pytestmark = pytest.mark.gpu

LAM = 0.1


def _mode_sweep(E, X, Y, Mm, tau, lam=LAM):
    mode_sweep(E, X, Y, Mm, tau, lam)


def test_cfg2_whole_iteration_against_fp64_closed_forms():
    I = J = 4096; K = 32
    R, M, _, _ = generate_bnmf(I, J, K, 0.1, tau=1.0, seed_data=0, seed_mask=1)
    pri = dict(alpha=1., beta=1., lambdaU=LAM, lambdaV=LAM)
    b = bnmf_gibbs_optimised(R, M, K, pri, verbose=False, seed=0)
    rs = np.random.RandomState(0)
    b.U, b.V, b.tau = rs.exponential(1.0, (I, K)), rs.exponential(1.0, (J, K)), 0.7      # (factors of the data's scale: no column collapses to zero)
    assert not b.is_small()
    U, V, tau = b.U.copy(), b.V.copy(), float(b.tau)
    b.run(1, update="mode")
    M64 = M.astype(np.float64)
    E = M64 * (R.astype(np.float64) - U @ V.T)
    _mode_sweep(E, U, V, M64, tau)
    assert np.abs(b.all_U[0] - U).max() <= 5e-4 * np.abs(U).max()
    Et = np.ascontiguousarray(E.T); Mt = np.ascontiguousarray(M64.T)
    del E
    _mode_sweep(Et, V, U, Mt, tau)
    assert np.abs(b.all_V[0] - V).max() <= 5e-4 * np.abs(V).max()
    sse = float((Et ** 2).sum()); n = float(Mt.sum())
    assert abs(b.all_performances["MSE"][0] - sse / n) <= 3e-4 * sse / n
    # tau of the mode harness = alpha_s / beta_s (:161-165 with the new factors)
    tau_ref = (1.0 + 0.5 * n) / (1.0 + 0.5 * sse)
    assert abs(b.all_tau[0] - tau_ref) <= 3e-4 * tau_ref


def test_cfg5_whole_vb_iteration_against_fp64_closed_forms():
    """One whole iteration of bnmf_vb_optimised.run (:133-146): update_U(k) + update_exp_U(k) for k = 0..63, the same for V,
    update_tau + update_exp_tau; then the pieces of the ELBO that do not underflow."""
    I = J = 8192; K = 64
    R, M, _, _ = generate_bnmf(I, J, K, 0.1, tau=1.0, seed_data=0, seed_mask=1)
    pri = dict(alpha=1., beta=1., lambdaU=LAM, lambdaV=LAM)
    b = bnmf_vb_optimised(R, M, K, pri, verbose=False)
    b.initialise("exp")
    rs = np.random.RandomState(1)                  # q(U), q(V) of the data's scale instead of mu = 1 / lambda = 10
    b.muU, b.muV = rs.exponential(1.0, (I, K)), rs.exponential(1.0, (J, K))
    b.tauU, b.tauV = 1.0 + rs.rand(I, K), 1.0 + rs.rand(J, K)
    b.expU, b.varU = O.tn_expectation(b.muU, b.tauU), O.tn_variance(b.muU, b.tauU)
    b.expV, b.varV = O.tn_expectation(b.muV, b.tauV), O.tn_variance(b.muV, b.tauV)
    b.exptau = 0.8
    eU, vU, eV, vV, exptau = b.expU.copy(), b.varU.copy(), b.expV.copy(), b.varV.copy(), float(b.exptau)
    b.run(1)
    M64 = M.astype(np.float64)
    E = M64 * (R.astype(np.float64) - eU @ eV.T)

    def vb_sweep(E, ex, var, exo, varo, Mm):
        mus, taus = np.zeros_like(ex), np.zeros_like(ex)
        for k in range(K):
            t = exptau * (Mm @ (varo[:, k] + exo[:, k] ** 2))                       # :189-190 / :193-194
            num = E @ exo[:, k] + ex[:, k] * (Mm @ (exo[:, k] ** 2))
            mu = (-LAM + exptau * num) / t
            new_e = O.tn_expectation(mu, t); new_v = O.tn_variance(mu, t)          # :199-211
            E -= Mm * np.outer(new_e - ex[:, k], exo[:, k])
            ex[:, k], var[:, k], mus[:, k], taus[:, k] = new_e, new_v, mu, t
        return mus, taus
    muU, tauU = vb_sweep(E, eU, vU, eV, vV, M64)
    sU = np.abs(eU).max()
    assert np.abs(b.expU - eU).max() <= 5e-4 * sU
    np.testing.assert_allclose(b.tauU, tauU, rtol=2e-5)
    assert np.abs(b.muU - muU).max() <= 5e-4 * max(np.abs(muU).max(), sU)
    assert np.abs(b.varU - vU).max() <= 2e-3 * np.abs(vU).max()
    Et = np.ascontiguousarray(E.T); Mt = np.ascontiguousarray(M64.T)
    del E
    muV, tauV = vb_sweep(Et, eV, vV, eU, vU, Mt)
    sV = np.abs(eV).max()
    assert np.abs(b.expV - eV).max() <= 5e-4 * sV
    # (tauV sums the device's own moments of U, which agree with the fp64 ones to 5e-4 of their scale, not of their value: columns
    # of U that have all but collapsed carry that as a relative error)
    np.testing.assert_allclose(b.tauV, tauV, rtol=3e-4, atol=1e-9 * tauV.max())
    assert np.abs(b.varV - vV).max() <= 2e-3 * np.abs(vV).max()
    # update_tau / update_exp_tau (:181-187, :213-215): exp_square_diff = sum_Omega [(R - E[U]E[V]^T)^2 + S2U S2V^T - E[U]^2 E[V]^2^T]
    S2U, S2V = vU + eU ** 2, vV + eV ** 2
    esd = float((Et ** 2).sum()) + float(((Mt @ S2U) * S2V).sum()) - float(((Mt @ (eU ** 2)) * (eV ** 2)).sum())
    n = float(Mt.sum())
    alpha_s, beta_s = 1.0 + 0.5 * n, 1.0 + 0.5 * esd
    assert abs(b.alpha_s - alpha_s) <= 1e-12 * alpha_s
    assert abs(b.beta_s - beta_s) <= 3e-4 * beta_s
    assert abs(b.exptau - alpha_s / beta_s) <= 3e-4 * alpha_s / beta_s
    assert abs(b.all_performances["MSE"][0] - float((Et ** 2).sum()) / n) <= 3e-4 * float((Et ** 2).sum()) / n
    # the sums elbo() takes over the factor entries (:168-176), but for log erfc (underflows to -inf at this size, as in the reference)
    t = b.all_elbo_terms[0]
    assert abs(t[0] - esd) <= 3e-4 * esd
    for got, want in [(t[2], float(0.5 * (tauU * (vU + (eU - muU) ** 2)).sum())), (t[4], float(np.log(tauU).sum())), (t[5], float((LAM * eU).sum())),
                      (t[6], float(0.5 * (tauV * (vV + (eV - muV) ** 2)).sum())), (t[8], float(np.log(tauV).sum())), (t[9], float((LAM * eV).sum()))]:
        assert abs(got - want) <= 1e-3 * abs(want), (got, want)


def test_cfg4_f_sweep_first_s_steps_and_g_sweep_against_fp64_closed_forms():
    """One BNMTF iteration in the mode update (bnmtf_gibbs_optimised.py:152-167): the 32 F columns, the first 64 of the 1 024
    sequential S entries walked in NumPy (an entry's value after the iteration is its value right after its own step), eight
    later entries up to the last one each checked as one step from the device's own earlier entries, and the 32 G columns
    -- the latter from the device's own (F, S)."""
    I = J = 4096; K = L = 32
    R, M, _, _, _ = generate_bnmtf(I, J, K, L, 0.1, tau=1.0, seed_data=0, seed_mask=1)
    pri = dict(alpha=1., beta=1., lambdaF=LAM, lambdaS=LAM, lambdaG=LAM)
    b = bnmtf_gibbs_optimised(R, M, K, L, pri, verbose=False, seed=3)
    rs = np.random.RandomState(2)                  # (factors of the data's scale: F S G^T ~ R, no column collapses to zero)
    b.F, b.S, b.G, b.tau = rs.exponential(1.0, (I, K)), rs.exponential(1.0, (K, L)), rs.exponential(1.0, (J, L)), 0.7
    F, S, G, tau = b.F.copy(), b.S.copy(), b.G.copy(), float(b.tau)
    S_start = S.copy()
    b.run(1, update="mode")
    R64 = R.astype(np.float64); M64 = M.astype(np.float64)
    # ---- F columns: the U sweep with V := G S^T (:195-199)
    Veff = G @ S.T
    E = M64 * (R64 - F @ Veff.T)
    _mode_sweep(E, F, Veff, M64, tau)
    sF = np.abs(F).max()
    assert np.abs(b.all_F[0] - F).max() <= 5e-4 * sF
    # ---- the first 64 S entries, row-major (:201-205): a = sum M (F_k^2 x G_l^2), num = F_k^T E G_l + S_kl a
    Sd = b.all_S[0]
    for step in range(64):
        k, l = divmod(step, L)
        new = s_step_mode(E, F, S, G, M64, tau, LAM, k, l)
        assert abs(Sd[k, l] - new) <= 1e-3 * max(abs(new), np.abs(S[:2]).max() * 1e-1), (k, l, Sd[k, l], new)
    # ---- later S entries, up to the last one, each from the device's own earlier entries (an entry sees the new values of the
    # entries before it and the old values of those behind it): the closed form of that one step, the whole chain length covered
    Sold = S_start
    for step in (64, 65, 257, 511, 512, 800, 1022, 1023):
        k, l = divmod(step, L)
        Smix = np.where((np.arange(K * L) < step).reshape(K, L), Sd.astype(np.float64), Sold)
        Emix = M64 * (R64 - (F @ Smix) @ G.T)
        new = s_step_mode(Emix, F, Smix, G, M64, tau, LAM, k, l)
        assert abs(Sd[k, l] - new) <= 1e-3 * max(abs(new), np.abs(Sold).max() * 1e-1), (k, l, Sd[k, l], new)
    del Emix
    # ---- G columns from the device's (F, S): the V sweep with U := F S (:207-211)
    Fd, Sd64 = b.all_F[0].astype(np.float64), Sd.astype(np.float64)
    Ueff = Fd @ Sd64
    Et = np.ascontiguousarray((M64 * (R64 - Ueff @ G.T)).T)
    Mt = np.ascontiguousarray(M64.T)
    del E
    _mode_sweep(Et, G, Ueff, Mt, tau)
    assert np.abs(b.all_G[0] - G).max() <= 5e-4 * np.abs(G).max()
    sse = float((Et ** 2).sum()); n = float(Mt.sum())
    assert abs(b.all_performances["MSE"][0] - sse / n) <= 3e-4 * sse / n
