"""BNMF Gibbs: the HIP path (through the C ABI / Python class) against the oracle, the
golden vectors generated from the reference and the reference's known answers.

Tolerances (fp32 device arithmetic vs the reference's fp64):
  tau* (a sum of squares, no cancellation)      rel 2e-6
  mu*  numerators                               abs 2e-5 * scale, scale = tau * sum_j |R~ V| (the cancelling terms)
  masked SSE / MSE from Gram identities         rel 2e-5
"""
import math

import numpy as np
import pytest

import bnmtf_amd
from bnmtf_amd import bnmf_gibbs_optimised
from oracle import bnmtf_oracle as O
from oracle import rng

pytestmark = pytest.mark.gpu

CASES = ["t5x3", "toy", "r37x29", "r40x33"]


def _pri(c):
    return dict(alpha=float(c["alpha"]), beta=float(c["beta"]), lambdaU=c["lambdaU"], lambdaV=c["lambdaV"])


def _mu_scale(M, R, U, V, tau, k, rows=True):
    """magnitude of the terms that cancel inside the numerator of mu (for the abs tolerance)"""
    if rows:
        return tau * ((M * np.abs(R)) @ np.abs(V[:, k]) + np.abs(U) @ np.abs(V.T @ V[:, k]))
    return tau * ((M * np.abs(R)).T @ np.abs(U[:, k]) + np.abs(V) @ np.abs(U.T @ U[:, k]))


@pytest.mark.parametrize("name", CASES)
def test_conditional_parameters_match_reference(golden, name):
    c = golden("bnmf_gibbs_cond.npz").case(name)
    b = bnmf_gibbs_optimised(c["R"], c["M"], int(c["K"]), _pri(c), verbose=False)
    b.U, b.V, b.tau = c["U"].copy(), c["V"].copy(), float(c["tau"])
    tot, row, col = b.omega_counts()
    assert tot == int(c["size_Omega"]) and np.array_equal(row, c["row_counts"]) and np.array_equal(col, c["col_counts"])
    assert b.alpha_s() == float(c["alpha_s"])
    assert abs(b.beta_s() - float(c["beta_s"])) <= 2e-6 * abs(float(c["beta_s"]))
    for k in range(b.K):
        tU = b.tauU(k)
        np.testing.assert_allclose(tU, c["tauU"][k], rtol=2e-6)
        sc = _mu_scale(c["M"], c["R"], c["U"], c["V"], float(c["tau"]), k, True) / c["tauU"][k]
        assert (np.abs(b.muU(c["tauU"][k], k) - c["muU"][k]) <= 2e-5 * sc + 1e-6).all()
        tV = b.tauV(k)
        np.testing.assert_allclose(tV, c["tauV"][k], rtol=2e-6)
        sc = _mu_scale(c["M"], c["R"], c["U"], c["V"], float(c["tau"]), k, False) / c["tauV"][k]
        assert (np.abs(b.muV(c["tauV"][k], k) - c["muV"][k]) <= 2e-5 * sc + 1e-6).all()
    p = b.predict_while_running()
    np.testing.assert_allclose([p["MSE"], p["R^2"]], c["perf"][:2], rtol=2e-6)
    if np.isfinite(c["perf"][2]):
        assert abs(p["Rp"] - c["perf"][2]) < 1e-6


def test_known_answers_of_reference_tests():
    """tests/code/test_bnmf_gibbs_optimised.py:144-203 on the device."""
    I, J, K = 5, 3, 2
    R = np.ones((I, J)); M = np.ones((I, J)); M[0, 0] = M[2, 2] = M[3, 1] = 0
    pri = dict(alpha=3, beta=1, lambdaU=2 * np.ones((I, K)), lambdaV=3 * np.ones((J, K)))
    b = bnmf_gibbs_optimised(R, M, K, pri, verbose=False)
    b.initialise('exp')
    assert (b.U == 0.5).all() and (b.V == 1. / 3.).all() and b.tau >= 0.0
    assert b.alpha_s() == 3 + 6.
    assert abs(b.beta_s() - (1 + .5 * (12 * (2. / 3.) ** 2))) < 1e-6
    b.tau = 3.
    tauU = 3. * np.array([[2. / 9.] * 2, [1. / 3.] * 2, [2. / 9.] * 2, [2. / 9.] * 2, [1. / 3.] * 2])
    muU = 1. / tauU * (3. * np.array([[2. * (5. / 6.) * (1. / 3.), 10. / 18.], [15. / 18.] * 2, [10. / 18.] * 2, [10. / 18.] * 2, [15. / 18.] * 2]) - 2.)
    for k in range(K):
        assert np.abs(b.tauU(k) - tauU[:, k]).max() < 1e-6
        assert np.abs(b.muU(tauU[:, k], k) - muU[:, k]).max() < 1e-5
        assert np.abs(b.tauV(k) - 3.).max() < 1e-6
        assert np.abs(b.muV(3. * np.ones(J), k) - (1. / 3.) * (3. * 4. * (5. / 6.) * .5 - 3.)).max() < 1e-5
    # test_run :207-236: shapes and "values changed"
    I, J, K = 10, 5, 2
    R = np.ones((I, J)); M = np.ones((I, J)); M[0, 0] = M[2, 2] = M[3, 1] = 0
    pri = dict(alpha=3, beta=1, lambdaU=2 * np.ones((I, K)), lambdaV=3 * np.ones((J, K)))
    b = bnmf_gibbs_optimised(R, M, K, pri, verbose=False, seed=3)
    b.initialise('exp')
    Us, Vs, taus = b.run(15)
    assert b.all_U.shape == (15, I, K) and b.all_V.shape == (15, J, K) and b.all_tau.shape == (15,)
    assert (Us[0] != 0.5).all() and (Vs[0] != 1. / 3.).all() and taus[1] != 3.
    assert (Us >= 0).all() and (Vs >= 0).all() and (taus > 0).all()
    assert len(b.all_times) == 15 and all(np.diff(b.all_times) > 0)


@pytest.mark.parametrize("name", ["t5x3", "toy", "r37x29"])
def test_postrun_api_matches_reference(golden, name):
    c = golden("bnmf_gibbs_cond.npz").case(name)
    b = bnmf_gibbs_optimised(c["R"], c["M"], int(c["K"]), _pri(c), verbose=False)
    b.all_U, b.all_V, b.all_tau = list(c["all_U"]), list(c["all_V"]), list(c["all_tau"])
    eU, eV, et = b.approx_expectation(2, 3)
    np.testing.assert_allclose(eU, c["expU"], rtol=1e-14)
    p = b.predict(c["M_test"], 2, 3)
    np.testing.assert_allclose([p["MSE"], p["R^2"], p["Rp"]], c["predict"], rtol=2e-6)
    q = [b.quality(m, 2, 3) for m in ["loglikelihood", "BIC", "AIC", "MSE", "ELBO"]]
    np.testing.assert_allclose(q, c["quality"], rtol=2e-6)
    with pytest.raises(AssertionError) as e:
        b.quality('FAIL', 2, 3)
    assert str(e.value) == "Unrecognised metric for model quality: FAIL."


def _chip_shape(monkeypatch, path):
    """Problems of this size run the on-chip sweep with one unit per wave (kernel_sweep_unit.hip: "chip"); "chip-pairs" switches
    that shape off, which leaves the pair layout's blocks (sweep_chip.inc: two units per wave).  Decided when the model is built."""
    if path == "chip-pairs":
        monkeypatch.setenv("BNMTF_UNIT", "0")
        return "chip"
    return path


def _set_path(b, path):
    """small: the one-launch kernel (kernel_small.hip); chip: the multi-launch path with the on-chip sweep; generic: with the generic sweep"""
    b.set_small_path(path == "small")
    b.set_sweep_path(path != "generic")
    assert b.is_small() == (path == "small")


@pytest.mark.parametrize("fast", ["small", "chip", "chip-pairs", "generic"])
@pytest.mark.parametrize("name", ["toy", "r37x29", "r40x33"])
def test_mode_update_trajectory_matches_oracle(golden, name, fast, monkeypatch):
    """Deterministic end-to-end parity: with every draw replaced by the mode
    max(0,mu) (the ICM update, nmf_icm.py:124-134) the whole data path -- both
    contractions, the sequential column loop, q maintenance, Gram-identity SSE, tau,
    metrics -- must follow the fp64 oracle."""
    c = golden("bnmf_gibbs_cond.npz").case(name)
    o = O.BNMFGibbsOracle(c["R"], c["M"], int(c["K"]), _pri(c))
    o.U, o.V, o.tau = c["U"].copy(), c["V"].copy(), float(c["tau"])
    o.run(8, draw=False)
    fast = _chip_shape(monkeypatch, fast)
    b = bnmf_gibbs_optimised(c["R"], c["M"], int(c["K"]), _pri(c), verbose=False)
    b.U, b.V, b.tau = c["U"].copy(), c["V"].copy(), float(c["tau"])
    _set_path(b, fast)
    b.run(8, update='mode')
    if fast == "chip":
        import os
        assert ("unit_sweep[rows=1" in b.describe()) == (os.environ.get("BNMTF_UNIT") != "0"), b.describe()
    np.testing.assert_allclose(b.all_performances['MSE'], o.all_performances['MSE'], rtol=2e-4)
    np.testing.assert_allclose(b.all_tau, o.all_tau, rtol=2e-4)
    assert np.abs(b.all_U[0] - o.all_U[0]).max() < 2e-4 * max(1.0, np.abs(o.all_U[0]).max())
    assert np.abs(b.all_V[0] - o.all_V[0]).max() < 2e-4 * max(1.0, np.abs(o.all_V[0]).max())
    assert np.abs(b.all_U[7] - o.all_U[7]).max() < 5e-3 * max(1.0, np.abs(o.all_U[7]).max())
    # state after run == last sample
    assert np.allclose(b.U, b.all_U[-1]) and np.allclose(b.V, b.all_V[-1]) and abs(b.tau - b.all_tau[-1]) < 1e-12


@pytest.mark.parametrize("fast", ["small", "chip", "chip-pairs", "generic"])
def test_gibbs_draws_follow_oracle_with_same_philox_stream(golden, fast, monkeypatch):
    """Same seed, same counters: the device sampler reproduces the oracle's draws
    (first sweep: every element within fp32 noise unless an accept/reject decision
    sits on a rounding boundary) and the MSE trajectory stays together."""
    t = golden("toy_data.npz").case("bnmf")
    g = golden("gibbs_trajectories.npz").case("bnmf")
    I, J = t["R"].shape; K = 10
    pri = dict(alpha=1., beta=1., lambdaU=0.1 * np.ones((I, K)), lambdaV=0.1 * np.ones((J, K)))
    o = O.BNMFGibbsOracle(t["R"], t["M"], K, pri, seed=77)
    o.U, o.V, o.tau = g["U0_seed0"].copy(), g["V0_seed0"].copy(), float(g["tau0_seed0"])
    o.run(30)
    fast = _chip_shape(monkeypatch, fast)
    b = bnmf_gibbs_optimised(t["R"], t["M"], K, pri, verbose=False, seed=77)
    b.U, b.V, b.tau = g["U0_seed0"].copy(), g["V0_seed0"].copy(), float(g["tau0_seed0"])
    _set_path(b, fast)
    b.run(30)
    d0 = np.abs(b.all_U[0] - o.all_U[0]) / (1e-3 + np.abs(o.all_U[0]))
    assert np.mean(d0 < 1e-3) > 0.99
    assert abs(b.all_tau[0] - o.all_tau[0]) < 1e-3 * o.all_tau[0]
    np.testing.assert_allclose(b.all_performances['MSE'][:3], o.all_performances['MSE'][:3], rtol=1e-3)
    # later iterations: same level (chains may decouple at a flipped accept)
    assert abs(np.mean(b.all_performances['MSE'][20:]) / np.mean(o.all_performances['MSE'][20:]) - 1) < 0.1


def test_gibbs_toy_trajectory_within_reference_bands(golden):
    """Config 1 (toy 100x80, K=10): masked-MSE trajectory against 10 seeded runs of the reference."""
    t = golden("toy_data.npz").case("bnmf")
    g = golden("gibbs_trajectories.npz").case("bnmf")
    I, J = t["R"].shape; K = 10
    pri = dict(alpha=1., beta=1., lambdaU=0.1 * np.ones((I, K)), lambdaV=0.1 * np.ones((J, K)))
    b = bnmf_gibbs_optimised(t["R"], t["M"], K, pri, verbose=False, seed=11)
    b.U, b.V, b.tau = g["U0_seed0"].copy(), g["V0_seed0"].copy(), float(g["tau0_seed0"])
    b.run(200)
    mse = np.array(b.all_performances['MSE']); ref = g["mse"]
    lo, hi = ref.min(axis=0), ref.max(axis=0)
    assert (mse[:60] > lo[:60] / 2.5).all() and (mse[:60] < hi[:60] * 2.5).all()
    m_ref = ref[:, 150:].mean(axis=1)
    assert m_ref.min() * 0.97 < mse[150:].mean() < m_ref.max() * 1.03
    assert abs(np.mean(b.all_tau[150:]) - g["tau"][:, 150:].mean()) < 0.05
    # metrics reported by the run (Gram identities) == direct fp64 evaluation of the final sample
    p = b.predict_while_running()
    assert abs(p["MSE"] - mse[-1]) < 2e-5 * mse[-1]
    assert abs(p["R^2"] - b.all_performances['R^2'][-1]) < 1e-5 and abs(p["Rp"] - b.all_performances['Rp'][-1]) < 1e-5
    # held-out prediction quality like the reference's (posterior mean, burn-in 100, thinning 2)
    eU, eV, _ = b.approx_expectation(100, 2)
    held = ((1 - t["M"]) * (t["R_true"] - eU @ eV.T) ** 2).sum() / (1 - t["M"]).sum()
    ref_held = g["heldout_mse_vs_Rtrue"]
    assert ref_held.min() * 0.8 < held < ref_held.max() * 1.2


def test_random_init_matches_numpy_stream():
    """initialise('random') consumes numpy's global stream like the reference's loop
    (bnmf_gibbs_optimised.py:106-109): same seed -> same U, V as the oracle restatement."""
    I, J, K = 12, 9, 3
    rs = np.random.RandomState(5)
    R = rs.rand(I, J) + 1; M = np.ones((I, J)); M[1, 2] = 0
    pri = dict(alpha=1., beta=1., lambdaU=0.5, lambdaV=2.0)
    np.random.seed(42); b = bnmf_gibbs_optimised(R, M, K, pri, verbose=False); b.initialise('random')
    np.random.seed(42); o = O.BNMFGibbsOracle(R, M, K, pri); o.initialise('random')
    assert np.array_equal(b.U, o.U) and np.array_equal(b.V, o.V)
    assert abs(b.tau - o.tau) < 2e-6 * o.tau


def test_large_shape_properties():
    """2048 x 1536, K=32: size-independent checks at a shape the oracle cannot sweep in seconds:
    (1) tauU/muU of a few columns vs the oracle formulas, (2) Gram-identity metrics ==
    direct fp64 metrics of the same sample, (3) MSE falls to the noise floor."""
    from bnmtf_amd.synthetic import generate_bnmf
    I, J, K = 2048, 1536, 32
    R, M, _, _ = generate_bnmf(I, J, K, 0.1, seed_data=3, seed_mask=4)
    pri = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)
    b = bnmf_gibbs_optimised(R, M, K, pri, verbose=False, seed=9)
    tot, row, col = b.omega_counts()
    assert tot == I * J - int(0.1 * I * J) == int(M.sum())
    assert np.array_equal(row, M.sum(axis=1)) and np.array_equal(col, M.sum(axis=0))
    np.random.seed(1); b.initialise('random')
    o = O.BNMFGibbsOracle(R.astype(np.float64), M, K, pri)
    o.U, o.V, o.tau = b.U.copy(), b.V.copy(), b.tau
    for k in (0, 17, 31):
        tU = o.tauU(k)
        np.testing.assert_allclose(b.tauU(k), tU, rtol=5e-6)
        sc = _mu_scale(o.M, o.R, o.U, o.V, o.tau, k, True) / tU
        assert (np.abs(b.muU(tU, k) - o.muU(tU, k)) <= 2e-5 * sc).all()
        tV = o.tauV(k)
        np.testing.assert_allclose(b.tauV(k), tV, rtol=5e-6)
        sc = _mu_scale(o.M, o.R, o.U, o.V, o.tau, k, False) / tV
        assert (np.abs(b.muV(tV, k) - o.muV(tV, k)) <= 2e-5 * sc).all()
    b.run(150)
    mse = b.all_performances['MSE']
    # the generic sweep kernel gives the same chain (same Philox counters): first iterations agree
    g2 = bnmf_gibbs_optimised(R, M, K, pri, verbose=False, seed=9)
    g2.U, g2.V, g2.tau = o.U.copy(), o.V.copy(), o.tau
    g2.set_sweep_path(False)
    g2.run(3)
    np.testing.assert_allclose(g2.all_performances['MSE'], mse[:3], rtol=2e-3)
    print('MSE trajectory', [round(float(m), 3) for m in mse[::10]])
    assert mse[0] > 10 * mse[-1] and 0.8 < mse[-1] < 1.3
    p = b.predict_while_running()
    assert abs(p["MSE"] - mse[-1]) < 5e-5 * mse[-1]
    assert abs(p["Rp"] - b.all_performances['Rp'][-1]) < 1e-5


def test_rccl_exchange_path_with_one_rank(golden, monkeypatch):
    """The multi-GPU exchange code (dlopen'd RCCL: in-place all-gather of the factor blocks, all-reduce of the
    three sums) run with a 1-rank communicator must leave the chain unchanged."""
    c = golden("bnmf_gibbs_cond.npz").case("r37x29")
    res = []
    for force in (False, True):
        if force:
            monkeypatch.setenv("BNMTF_FORCE_COMM", "1")
        b = bnmf_gibbs_optimised(c["R"], c["M"], int(c["K"]), _pri(c), verbose=False, seed=5)
        b.U, b.V, b.tau = c["U"].copy(), c["V"].copy(), float(c["tau"])
        b.set_small_path(False)            # (both runs on the multi-launch path: the exchange is part of that one)
        b.run(5)
        res.append((b.all_U.copy(), b.all_tau.copy(), list(b.all_performances['MSE'])))
    # the exchange path folds the per-block sums in a different order (fp64 rounding): tau agrees to ~1e-15 relative
    assert np.array_equal(res[0][0][0], res[1][0][0])
    np.testing.assert_allclose(res[0][1], res[1][1], rtol=1e-10)
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(res[0][2], res[1][2], rtol=1e-8)


def test_pinned_sample_arrays_are_reused(monkeypatch):
    """run()'s page-locked sample arrays go back to a pool when their last view goes away and are handed out again for the same
    size (page-locking is tens of milliseconds per call otherwise); BNMTF_PIN_POOL_MB bounds what waits there."""
    from bnmtf_amd import _lib
    _lib._pin_pool_clear()
    a = _lib.sample_buffer((7, 33, 5))
    addr = a.ctypes.data
    a[:] = 3.0
    del a
    assert _lib._PIN_POOL_BYTES[0] == 7 * 33 * 5 * 4
    b = _lib.sample_buffer((7, 33, 5))
    assert b.ctypes.data == addr and _lib._PIN_POOL_BYTES[0] == 0
    c = _lib.sample_buffer((7, 33, 5))                    # a second one of the same size while the first is alive: a new block
    assert c.ctypes.data != addr
    monkeypatch.setenv("BNMTF_PIN_POOL_MB", "0")
    del b, c
    assert _lib._PIN_POOL_BYTES[0] == 0 and not any(_lib._PIN_POOL.values())
