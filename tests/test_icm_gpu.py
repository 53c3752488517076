"""ICM on the device (classes nmf_icm / nmtf_icm through the C ABI, update rule BNMTF_UPDATE_ICM) against the
reference's own trajectories (tests/golden/icm.npz).  Deterministic, so the comparison is end to end; fp32 device
arithmetic against the reference's fp64: factors rel 2e-3 of their scale after 10-12 iterations (the updates
contract: errors do not grow), tau and MSE rel 5e-4."""
import numpy as np
import pytest

from bnmtf_amd import nmf_icm, nmtf_icm

pytestmark = pytest.mark.gpu


def _toy(golden, which):
    t = golden("toy_data.npz").case(which)
    return t["R"], t["M"]


@pytest.mark.parametrize("name", ["nmf_conv", "nmf_min", "nmf_collapse"])
def test_nmf_icm_trajectory_matches_reference(golden, name):
    c = golden("icm.npz").case(name)
    R, M = _toy(golden, "bnmf")
    K, lam, mtn, iters = int(c["cfg"][0]), float(c["cfg"][1]), float(c["cfg"][2]), int(c["cfg"][3])
    b = nmf_icm(R, M, K, dict(alpha=1.0, beta=1.0, lambdaU=lam, lambdaV=lam), verbose=False)
    b.initialise("exp")
    b.U, b.V = c["U0"].copy(), c["V0"].copy()
    b.tau = (b.alpha_s() - 1) / b.beta_s()
    assert b.tau == pytest.approx(float(c["tau0"]), rel=2e-5)
    assert b.run(iters, minimum_TN=mtn) is None
    np.testing.assert_allclose(b.all_tau, c["all_tau"], rtol=5e-4)
    np.testing.assert_allclose(b.all_performances["MSE"], c["mse"], rtol=5e-4)
    np.testing.assert_allclose(b.all_performances["R^2"], c["r2"], rtol=5e-4, atol=1e-4)
    sU, sV = max(np.abs(c["U"]).max(), 1e-3), max(np.abs(c["V"]).max(), 1e-3)
    assert np.abs(b.U - c["U"]).max() <= 2e-3 * sU and np.abs(b.V - c["V"]).max() <= 2e-3 * sV
    if mtn > 0:
        assert b.U.min() >= mtn * (1 - 1e-6) and b.V.min() >= mtn * (1 - 1e-6)
    q = [b.quality(m) for m in ["loglikelihood", "BIC", "AIC", "MSE", "ELBO"]]
    np.testing.assert_allclose(q, c["quality"], rtol=1e-3)
    p = b.predict(c["Mpred"])
    np.testing.assert_allclose([p["MSE"], p["R^2"]], c["pred"][:2], rtol=1e-3, atol=1e-4)
    assert len(b.all_times) == iters and all(np.diff(b.all_times) >= 0)


@pytest.mark.parametrize("name", ["nmtf_conv", "nmtf_min"])
def test_nmtf_icm_trajectory_matches_reference(golden, name):
    c = golden("icm.npz").case(name)
    R, M = _toy(golden, "bnmtf")
    K, lam, mtn, iters = int(c["cfg"][0]), float(c["cfg"][1]), float(c["cfg"][2]), int(c["cfg"][3])
    b = nmtf_icm(R, M, K, K, dict(alpha=1.0, beta=1.0, lambdaF=lam, lambdaS=lam, lambdaG=lam), verbose=False)
    b.initialise("exp", "exp")
    b.F, b.S, b.G = c["F0"].copy(), c["S0"].copy(), c["G0"].copy()
    b.tau = (b.alpha_s() - 1) / b.beta_s()
    assert b.tau == pytest.approx(float(c["tau0"]), rel=2e-5)
    assert b.run(iters, minimum_TN=mtn) is None
    np.testing.assert_allclose(b.all_tau, c["all_tau"], rtol=2e-3)
    np.testing.assert_allclose(b.all_performances["MSE"], c["mse"], rtol=2e-3)
    for got, ref in [(b.F, c["F"]), (b.S, c["S"]), (b.G, c["G"])]:
        assert np.abs(got - ref).max() <= 1e-2 * np.abs(ref).max()
    if mtn > 0:
        assert min(b.F.min(), b.S.min(), b.G.min()) >= mtn * (1 - 1e-6)
    q = [b.quality(m) for m in ["loglikelihood", "BIC", "AIC", "MSE", "ELBO"]]
    np.testing.assert_allclose(q, c["quality"], rtol=5e-3)


def test_icm_rejects_bad_arguments(golden):
    R, M = _toy(golden, "bnmf")
    b = nmf_icm(R, M, 3, dict(alpha=1.0, beta=1.0, lambdaU=1.0, lambdaV=1.0), verbose=False)
    with pytest.raises(AssertionError) as e:
        b.initialise("bad")
    assert str(e.value) == "Unknown initialisation option: bad. Should be 'random' or 'exp'."
    b.initialise("exp")
    with pytest.raises(AssertionError) as e:
        b.quality("FAIL")
    assert str(e.value) == "Unrecognised metric for model quality: FAIL."
