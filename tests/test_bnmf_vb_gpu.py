"""BNMF VB on the device: deterministic trajectory (init='exp') against the reference's own run
(tests/golden/bnmf_vb.npz), and the reference's known-answer tests."""
import math

import numpy as np
import pytest

from bnmtf_amd import bnmf_vb_optimised

pytestmark = pytest.mark.gpu

# BNMTF_WIDE=1 (read when the handle is created) selects the 16-unit-wave block shape of the VB sweep -- the shape the
# 8192 x 8192 configuration runs, sweep_vb_kernel<., 16, 0> with the fp32 moments routine on wave 0 -- on any size
# -- with q handed over between the half sweeps (the default there) and, "1-prepass", rebuilt by every sweep's
# pre-pass (BNMTF_HANDOVER=0)
# "...-masked" (BNMTF_VB_PATH=masked, read when the model is built): the sweep on the Gibbs sweep's on-chip kernels (sweep_chip.inc, MODE =
# kSweepVB) with the two chain-independent masked sums of every (unit, column) from kernel_maskgemm.hip (bits x int8 digit
# planes) -- the path the 8192 x 8192, K = 64 configuration takes by itself; the others force the pair-panel kernel
SHAPES = pytest.mark.parametrize("wide", [None, "1", "1-prepass", "masked", "1-masked", "1-prepass-masked"],
                                 ids=["8wave", "16wave", "16wave-prepass", "8wave-masked", "16wave-masked", "16wave-prepass-masked"])


def _shape(monkeypatch, wide):
    masked = wide is not None and wide.endswith("masked")
    monkeypatch.setenv("BNMTF_VB_PATH", "masked" if masked else "pairs")
    if masked:
        wide = wide[:-len("masked")].rstrip("-") or None
    if wide is not None:
        monkeypatch.setenv("BNMTF_WIDE", "1")
        monkeypatch.setenv("BNMTF_HANDOVER", "0" if wide.endswith("prepass") else "1")
    return wide


@SHAPES
def test_toy_trajectory_matches_reference(golden, monkeypatch, wide):
    """Config-5 algorithm on config-1 data: MSE / exptau / ELBO per iteration vs the reference.
    The q-parameters live on the device in fp32 (reductions fp64); 20 deterministic fixed-point
    iterations amplify that rounding (the run passes through a fast transition around iterations 8-12): rel 1e-3 on
    MSE and exptau, 2e-4 on the ELBO; the first iterations, before any amplification, 2e-5."""
    wide = _shape(monkeypatch, wide)
    g = golden("bnmf_vb.npz").case("toy")
    t = golden("toy_data.npz").case("bnmf")
    I, J = t["R"].shape; K = 10
    b = bnmf_vb_optimised(t["R"], t["M"], K, dict(alpha=1., beta=1., lambdaU=0.1 * np.ones((I, K)), lambdaV=0.1 * np.ones((J, K))), verbose=False)
    b.initialise('exp')
    assert abs(b.exptau - float(g["init_exptau"])) < 2e-6 * b.exptau
    np.testing.assert_allclose(b.expU, g["init_expU"], rtol=1e-9)
    assert abs(b.exp_square_diff() - float(g["init_esd"])) < 2e-6 * float(g["init_esd"])
    b.run(20)
    np.testing.assert_allclose(b.all_performances['MSE'], g["mse"], rtol=1e-3)
    np.testing.assert_allclose(b.all_performances['MSE'][:5], g["mse"][:5], rtol=2e-5)
    np.testing.assert_allclose(b.all_exp_tau, g["exptau"], rtol=1e-3)
    np.testing.assert_allclose(b.all_elbo, g["elbo"], rtol=2e-4)
    np.testing.assert_allclose(b.all_elbo[:6], g["elbo"][:6], rtol=2e-5)
    for nm in ["expU", "expV", "muU", "muV", "tauU", "tauV"]:
        ref = g["it20/" + nm]
        assert np.abs(getattr(b, nm) - ref).max() < 2e-3 * np.abs(ref).max(), nm
    assert abs(b.elbo() - g["elbo"][-1]) < 2e-4 * abs(g["elbo"][-1])
    q = [b.quality(m) for m in ["loglikelihood", "BIC", "AIC", "MSE", "ELBO"]]
    np.testing.assert_allclose(q, g["quality"], rtol=1e-3)
    p = b.predict(t["M"])
    np.testing.assert_allclose([p["MSE"], p["R^2"], p["Rp"]], g["final_perf"], rtol=1e-3)
    assert len(b.all_times) == 20


@SHAPES
def test_ragged_case_matches_reference(golden, monkeypatch, wide):
    masked = wide is not None and wide.endswith("masked")
    wide = _shape(monkeypatch, wide)
    g = golden("bnmf_vb.npz").case("r31x23")
    K = 4
    b = bnmf_vb_optimised(g["R"], g["M"], K, dict(alpha=2., beta=.5, lambdaU=g["lambdaU"], lambdaV=g["lambdaV"]), verbose=False)
    b.initialise('exp', {"tauU": g["tauU0"], "tauV": g["tauV0"]})
    assert ("sweep_nw=16" in b.describe()) == (wide is not None) and ("handover=1" in b.describe()) == (wide == "1")   # (small problems: only when asked for)
    b.run(10)
    assert ("vb_sweep=masked" if masked else "vb_sweep=pairs") in b.describe()      # the path that was asked for is the one that ran
    np.testing.assert_allclose(b.all_performances['MSE'], g["mse"], rtol=1e-3)
    np.testing.assert_allclose(b.all_elbo, g["elbo"], rtol=1e-4)
    assert np.abs(b.expU - g["it10/expU"]).max() < 2e-3 * np.abs(g["it10/expU"]).max()


@pytest.mark.parametrize("path", ["pairs", "masked"])
@pytest.mark.parametrize("handover", ["1", "0"])
def test_a_run_split_in_two_calls_is_the_same_trajectory(monkeypatch, handover, path):
    """run(3); run(4) is run(7), bit for bit: the second call finds the device state it left (no upload) and, with the hand-over,
    q of the missing entries where the first call's last half sweep put it."""
    from bnmtf_amd.synthetic import generate_bnmf
    monkeypatch.setenv("BNMTF_WIDE", "1")
    monkeypatch.setenv("BNMTF_HANDOVER", handover)
    monkeypatch.setenv("BNMTF_VB_PATH", path)
    I, J, K = 600, 500, 20
    R, M, _, _ = generate_bnmf(I, J, K, 0.15, seed_data=3, seed_mask=4)
    pri = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)
    def model():
        b = bnmf_vb_optimised(R, M, K, pri, verbose=False)
        b.initialise('exp')
        return b
    one = model(); one.run(7)
    two = model(); two.run(3); two.run(4)
    assert two.all_exp_tau == one.all_exp_tau[3:] and two.all_elbo == one.all_elbo[3:]
    assert np.array_equal(two.expU, one.expU) and np.array_equal(two.tauV, one.tauV)


def test_known_answers_of_reference_tests():
    """tests/code/test_bnmf_vb_optimised.py:218-311."""
    I, J, K = 5, 3, 2
    R = np.ones((I, J)); M = np.ones((I, J)); M[0, 0] = M[2, 2] = M[3, 1] = 0
    lambdaU = 2 * np.ones((I, K)); lambdaV = 3 * np.ones((J, K))
    pri = dict(alpha=3, beta=1, lambdaU=lambdaU, lambdaV=lambdaV)
    b = bnmf_vb_optimised(R, M, K, pri, verbose=False)
    b.expU = 1. / lambdaU; b.expV = 1. / lambdaV; b.varU = 2 * np.ones((I, K)); b.varV = 3 * np.ones((J, K))
    assert abs(b.exp_square_diff() - 172.66666666666666) < 2e-5      # expV = 1/3 is rounded to fp32 on the device
    b.update_tau()
    assert b.alpha_s == 3 + 12. / 2. and abs(b.beta_s - (1 + 172.66666666666666 / 2.)) < 2e-5
    for k in range(K):
        b = bnmf_vb_optimised(R, M, K, pri, verbose=False)
        b.muU = np.zeros((I, K)); b.tauU = np.zeros((I, K)); b.muV = np.zeros((J, K)); b.tauV = np.zeros((J, K))
        b.expU = 1. / lambdaU; b.expV = 1. / lambdaV; b.varU = 2 * np.ones((I, K)); b.varV = 3 * np.ones((J, K))
        b.exptau = 3.
        b.update_U(k)
        for i in range(I):
            w = (M[i] * (b.expV[:, k] ** 2 + b.varV[:, k])).sum()
            assert abs(b.tauU[i, k] - 3. * w) < 1e-5 * 3. * w
            ref = (1. / (3. * w)) * (-2. + 3. * (M[i] * ((R[i] - b.expU[i] @ b.expV.T + b.expU[i, k] * b.expV[:, k]) * b.expV[:, k])).sum())
            assert abs(b.muU[i, k] - ref) < 1e-5
        b.update_V(k)
        for j in range(J):
            w = (M[:, j] * (b.expU[:, k] ** 2 + b.varU[:, k])).sum()
            assert abs(b.tauV[j, k] - 3. * w) < 1e-5 * 3. * w
    b = bnmf_vb_optimised(R, M, K, pri, verbose=False)
    b.initialise()
    assert abs(b.exptau - (3 + 12. / 2.) / (1 + 35.4113198623 / 2.)) < 1e-6
    assert abs(b.explogtau - (2.1406414779556 - math.log(1 + 35.4113198623 / 2.))) < 1e-6
    b.tauU = 4 * np.ones((I, K)); b.update_exp_U(0)
    assert np.abs(b.expU[:, 0] - (0.5 + 0.5 * 0.2876155949126352)).max() < 1e-5
    assert np.abs(b.varU[:, 0] - 0.25 * (1. - 0.37033832534958433)).max() < 1e-5
    with pytest.raises(AssertionError) as e:
        b.quality('FAIL')
    assert str(e.value) == "Unrecognised metric for model quality: FAIL."


def test_large_shape_identity():
    """1536 x 1024, K=16: the Gram-identity exp_square_diff used inside run() equals the direct fp64 kernel."""
    from bnmtf_amd.synthetic import generate_bnmf
    I, J, K = 1536, 1024, 16
    R, M, _, _ = generate_bnmf(I, J, K, 0.1, seed_data=7, seed_mask=8)
    b = bnmf_vb_optimised(R, M, K, dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1), verbose=False)
    b.initialise('exp')
    b.run(15)
    mse = b.all_performances['MSE']
    assert mse[-1] < mse[0] / 100
    esd = b.exp_square_diff()
    assert abs(b.beta_s - (1. + 0.5 * esd)) < 5e-5 * b.beta_s
    p = b.predict(M)
    assert abs(p["MSE"] - mse[-1]) < 1e-4 * mse[-1]


@SHAPES
def test_fast_vb_sweep_equals_generic_sweep(monkeypatch, wide):
    """The register/LDS-resident VB sweep (kernel_sweep_vb.hip + vb_pieces_kernel) against the generic kernel on a
    ragged problem: same fixed-point iteration, fp32 rounding differences only."""
    from bnmtf_amd.synthetic import generate_bnmf
    wide = _shape(monkeypatch, wide)
    I, J, K = 600, 500, 20
    R, M, _, _ = generate_bnmf(I, J, K, 0.15, seed_data=3, seed_mask=4)
    pri = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)
    res = {}
    for mode in ("fast", "generic"):
        if mode == "generic":
            monkeypatch.setenv("BNMTF_VB_GENERIC", "1")
        b = bnmf_vb_optimised(R, M, K, pri, verbose=False)
        b.initialise('exp')
        assert ("sweep_nw=16" in b.describe()) == (wide is not None)
        b.run(6)
        res[mode] = (np.array(b.all_performances['MSE']), np.array(b.all_exp_tau), np.array(b.all_elbo), b.expU.copy(), b.varU.copy(), b.tauV.copy())
    f, g = res["fast"], res["generic"]
    np.testing.assert_allclose(f[0], g[0], rtol=2e-5)
    np.testing.assert_allclose(f[1], g[1], rtol=2e-5)
    np.testing.assert_allclose(f[2], g[2], rtol=5e-6)
    assert np.abs(f[3] - g[3]).max() < 2e-3 * np.abs(g[3]).max()
    assert np.abs(f[4] - g[4]).max() < 5e-3 * np.abs(g[4]).max()
    np.testing.assert_allclose(f[5], g[5], rtol=1e-4)


@pytest.mark.parametrize("I,J,K,frac", [(515, 389, 40, 0.15), (600, 500, 20, 0.3), (130, 97, 7, 0.6)])
def test_masked_sums_from_the_matrix_cores_against_fp64(I, J, K, frac):
    """csrc/kernel_maskgemm.hip through the hook bnmf_vb_masked_sums: sum over a unit's MISSING entries of the other factor's
    S2 = var + exp^2 and of its exp^2 (the chain-independent parts of tauU / muU, bnmf_vb_optimised.py:189-199), formed from the
    mask's bits and the moments on a per-column 22-bit fixed-point grid with integer accumulation.  Stated error: an element
    is off by at most 2^(e_c - 23) with 2^(e_c - 1) <= max of the column < 2^e_c, i.e. <= max_c * 2^-22; the sum of n elements by
    at most n times that (in practice ~sqrt(n)), plus one fp32 rounding per slab.  The moments span six decades here."""
    from bnmtf_amd.synthetic import generate_bnmf
    R, M, _, _ = generate_bnmf(I, J, K, frac, seed_data=11, seed_mask=12)
    rs = np.random.RandomState(5)
    b = bnmf_vb_optimised(R, M, K, dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1), verbose=False)
    b.initialise('exp')
    b.expU = 10. ** rs.uniform(-4, 1, (I, K)); b.varU = 10. ** rs.uniform(-6, 0, (I, K))
    b.expV = 10. ** rs.uniform(-4, 1, (J, K)); b.varV = 10. ** rs.uniform(-6, 0, (J, K))
    miss = 1.0 - M
    for which, (mm, ex, var) in enumerate([(miss, b.expV, b.varV), (miss.T, b.expU, b.varU)]):
        e32 = ex.astype(np.float32).astype(np.float64); v32 = var.astype(np.float32).astype(np.float64)     # what the device holds
        s2 = (v32.astype(np.float32) + (e32 * e32).astype(np.float32)).astype(np.float64)                     # S2 as the device forms it (fp32)
        e2 = (e32 * e32).astype(np.float32).astype(np.float64)
        asq, vsq = b.masked_sums(which)
        n_miss = mm.sum(1)[:, None]
        for got, op in ((asq, s2), (vsq, e2)):
            ref = mm @ op
            bound = n_miss * op.max(0)[None, :] * 2.0 ** -22 + 3e-7 * ref + 1e-30
            assert np.all(np.abs(got - ref) <= bound), float((np.abs(got - ref) / bound).max())
            # ... and in practice far inside it: the relative error of the sums that matter (those not dwarfed by the column's maximum)
            big = ref > 1e-3 * op.max(0)[None, :] * np.maximum(n_miss, 1)
            assert np.abs(got - ref)[big].max() <= 2e-6 * ref[big].max()
