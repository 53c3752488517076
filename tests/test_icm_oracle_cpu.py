"""ICM (nmf_icm / nmtf_icm): the oracle restatement against trajectories produced by the reference itself
(tests/golden/icm.npz, made by tests/golden/make_golden.py icm).  ICM is deterministic given the initial
state, so the whole run is compared: final factors, tau per iteration, the three metrics, quality()."""
import numpy as np
import pytest

from oracle import bnmtf_oracle as O

NMF_CASES = ["nmf_conv", "nmf_min", "nmf_collapse"]
NMTF_CASES = ["nmtf_conv", "nmtf_min"]


def _toy(golden, which):
    t = golden("toy_data.npz").case(which)
    return t["R"], t["M"]


@pytest.mark.parametrize("name", NMF_CASES)
def test_nmf_icm_oracle_matches_reference(golden, name):
    c = golden("icm.npz").case(name)
    R, M = _toy(golden, "bnmf")
    K, lam, mtn, iters = int(c["cfg"][0]), float(c["cfg"][1]), float(c["cfg"][2]), int(c["cfg"][3])
    o = O.NMFICMOracle(R, M, K, dict(alpha=1.0, beta=1.0, lambdaU=lam, lambdaV=lam))
    o.U, o.V = c["U0"].copy(), c["V0"].copy()
    o.tau = O.gamma_mode(o.alpha_s(), o.beta_s())
    assert o.tau == pytest.approx(float(c["tau0"]), rel=1e-12)
    with np.errstate(all="ignore"):
        o.run(iters, minimum_TN=mtn)
        q = [o.quality(m) for m in ["loglikelihood", "BIC", "AIC", "MSE", "ELBO"]]
        p = o.predict(c["Mpred"])
    np.testing.assert_allclose(o.U, c["U"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(o.V, c["V"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(o.all_tau, c["all_tau"], rtol=1e-10)
    np.testing.assert_allclose(o.all_performances["MSE"], c["mse"], rtol=1e-10)
    np.testing.assert_allclose(o.all_performances["R^2"], c["r2"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(o.all_performances["Rp"], c["rp"], rtol=1e-9, equal_nan=True)
    np.testing.assert_allclose(q, c["quality"], rtol=1e-10)
    np.testing.assert_allclose([p["MSE"], p["R^2"], p["Rp"]], c["pred"], rtol=1e-9, equal_nan=True)
    if mtn > 0:
        assert o.U.min() >= mtn and o.V.min() >= mtn


@pytest.mark.parametrize("name", NMTF_CASES)
def test_nmtf_icm_oracle_matches_reference(golden, name):
    c = golden("icm.npz").case(name)
    R, M = _toy(golden, "bnmtf")
    K, lam, mtn, iters = int(c["cfg"][0]), float(c["cfg"][1]), float(c["cfg"][2]), int(c["cfg"][3])
    o = O.NMTFICMOracle(R, M, K, K, dict(alpha=1.0, beta=1.0, lambdaF=lam, lambdaS=lam, lambdaG=lam))
    o.F, o.S, o.G = c["F0"].copy(), c["S0"].copy(), c["G0"].copy()
    o.tau = O.gamma_mode(o.alpha_s(), o.beta_s())
    assert o.tau == pytest.approx(float(c["tau0"]), rel=1e-12)
    with np.errstate(all="ignore"):
        o.run(iters, minimum_TN=mtn)
        q = [o.quality(m) for m in ["loglikelihood", "BIC", "AIC", "MSE", "ELBO"]]
    np.testing.assert_allclose(o.F, c["F"], rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(o.S, c["S"], rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(o.G, c["G"], rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(o.all_tau, c["all_tau"], rtol=1e-9)
    np.testing.assert_allclose(o.all_performances["MSE"], c["mse"], rtol=1e-9)
    np.testing.assert_allclose(q, c["quality"], rtol=1e-9)


def test_icm_known_answers():
    """gamma_mode / TN mode as the reference's distribution tests state them (tests/code/distributions)."""
    assert O.gamma_mode(2.0, 3.0) == pytest.approx(1.0 / 3.0)
    np.testing.assert_array_equal(O.tn_mode(np.array([-1.5, 0.0, 2.0])), np.array([0.0, 0.0, 2.0]))
