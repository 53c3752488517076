"""oracle.BNMTFVBOracle (the NumPy restatement of code/models/bnmtf_vb_optimised.py) pinned against vectors the reference
itself produced (tests/golden/bnmtf_vb.npz, generator tests/golden/make_golden.py `trivb`) and against the known answers
of the reference's own tests (tests/code/test_bnmtf_vb_optimised.py)."""
import itertools
import math
import random

import numpy as np

from oracle import bnmtf_oracle as O

NAMES = ["muF", "tauF", "expF", "varF", "muS", "tauS", "expS", "varS", "muG", "tauG", "expG", "varG"]


def _t5x3():
    I, J, K, L = 5, 3, 2, 4
    R = np.ones((I, J)); M = np.ones((I, J)); M[0, 0] = M[2, 2] = M[3, 1] = 0
    pri = dict(alpha=3, beta=1, lambdaF=2 * np.ones((I, K)), lambdaS=3 * np.ones((K, L)), lambdaG=4 * np.ones((J, L)))
    return R, M, K, L, pri


def _orders(g, tag, L):
    return [([(int(a) // L, int(a) % L) for a in oS], [int(x) for x in oF], [int(x) for x in oG])
            for oS, oF, oG in zip(g[tag + "order_S"], g[tag + "order_F"], g[tag + "order_G"])]


def test_known_answers_of_the_reference_tests():
    """test_bnmtf_vb_optimised.py:281-300 (exp_square_diff = 2749 + 5/6, update_tau) and :232-279 (the ELBO constant)."""
    R, M, K, L, pri = _t5x3()
    I, J = R.shape
    b = O.BNMTFVBOracle(R, M, K, L, pri)
    b.expF = 1. / pri["lambdaF"]; b.expS = 1. / pri["lambdaS"]; b.expG = 1. / pri["lambdaG"]
    b.varF = np.ones((I, K)) * 2; b.varS = np.ones((K, L)) * 3; b.varG = np.ones((J, L)) * 4
    assert abs(b.exp_square_diff() - (2749 + 5. / 6.)) < 1e-12
    b.update_tau()
    assert b.alpha_s == 3 + 12. / 2. and abs(b.beta_s - (1 + (2749 + 5. / 6.) / 2.)) < 1e-12
    b.expF = 5 * np.ones((I, K)); b.expS = 6 * np.ones((K, L)); b.expG = 7 * np.ones((J, L))
    b.varF = 11 * np.ones((I, K)); b.varS = 12 * np.ones((K, L)); b.varG = 13 * np.ones((J, L))
    b.exptau, b.explogtau = 8., 9.
    b.muF = 14 * np.ones((I, K)); b.muS = 15 * np.ones((K, L)); b.muG = 16 * np.ones((J, L))
    b.tauF = np.ones((I, K)) / 100.; b.tauS = np.ones((K, L)) / 101.; b.tauG = np.ones((J, L)) / 102.
    b.alpha_s, b.beta_s = 20., 21.
    ELBO = 12. / 2. * (9. - math.log(2 * math.pi)) - 8. / 2. * (33828492 + 12763008) \
        + 5 * 2 * (math.log(2.) - 2. * 5.) + 2 * 4 * (math.log(3.) - 3. * 6.) + 3 * 4 * (math.log(4.) - 4. * 7.) \
        + 3. * np.log(1.) - np.log(math.gamma(3.)) + 2. * 9. - 1. * 8. \
        - 20. * np.log(21.) + np.log(math.gamma(20.)) - 19. * 9. + 21. * 8. \
        - 0.5 * 5 * 2 * math.log(1. / 100.) + 0.5 * 5 * 2 * math.log(2 * math.pi) + 5 * 2 * math.log(1. - 0.080756659233771066) \
        + 0.5 * 5 * 2 * 1. / 100. * (11. + 81.) \
        - 0.5 * 4 * 2 * math.log(1. / 101.) + 0.5 * 4 * 2 * math.log(2 * math.pi) + 4 * 2 * math.log(1. - 0.067776752211548219) \
        + 0.5 * 4 * 2 * 1. / 101. * (12. + 81.) \
        - 0.5 * 4 * 3 * math.log(1. / 102.) + 0.5 * 4 * 3 * math.log(2 * math.pi) + 4 * 3 * math.log(1. - 0.056570004076003155) \
        + 0.5 * 4 * 3 * 1. / 102. * (13. + 81.)
    assert abs(b.elbo() - ELBO) < 1e-9 * abs(ELBO)


def _case(golden, tag):
    g = golden("bnmtf_vb.npz").case(tag)
    if tag == "t5x3":
        R, M, K, L, pri = _t5x3()
    else:
        R, M = g["R"], g["M"]
        K, L = g["lambdaS"].shape
        pri = dict(alpha=2.0, beta=0.5, lambdaF=g["lambdaF"], lambdaS=g["lambdaS"], lambdaG=g["lambdaG"])
    return g, R, M, K, L, pri


def test_single_updates_match_the_reference(golden):
    for tag in ("t5x3", "r33x27"):
        g, R, M, K, L, pri = _case(golden, tag)
        b = O.BNMTFVBOracle(R, M, K, L, pri)
        for n in NAMES:
            setattr(b, n, g["state/" + n].copy())
        b.exptau = float(g["state/exptau"])
        assert abs(b.exp_square_diff() - float(g["esd"])) < 1e-10 * float(g["esd"])
        for k in range(K):
            b.update_F(k)
        np.testing.assert_allclose(b.tauF, g["upd/tauF"], rtol=1e-12); np.testing.assert_allclose(b.muF, g["upd/muF"], rtol=1e-9, atol=1e-12)
        for k, l in itertools.product(range(K), range(L)):
            b.update_S(k, l)
        np.testing.assert_allclose(b.tauS, g["upd/tauS"], rtol=1e-12); np.testing.assert_allclose(b.muS, g["upd/muS"], rtol=1e-9, atol=1e-12)
        for l in range(L):
            b.update_G(l)
        np.testing.assert_allclose(b.tauG, g["upd/tauG"], rtol=1e-12); np.testing.assert_allclose(b.muG, g["upd/muG"], rtol=1e-9, atol=1e-12)


def test_runs_match_the_reference_with_stored_orders_and_with_the_python_random_stream(golden):
    # ragged case: ten iterations from a given tau initialisation
    g, R, M, K, L, pri = _case(golden, "r33x27")
    for mode in ("stored", "stream"):
        b = O.BNMTFVBOracle(R, M, K, L, pri)
        b.initialise("exp", "exp", {"tauF": g["init/tauF"], "tauS": g["init/tauS"], "tauG": g["init/tauG"]})
        assert abs(b.exptau - float(g["init_exptau"])) < 1e-12 * b.exptau
        if mode == "stored":
            b.run(10, orders=_orders(g, "", L))
        else:
            random.seed(int(g["seed"]))                 # the reference's own shuffles (bnmtf_vb_optimised.py:172-186)
            b.run(10)
        np.testing.assert_allclose(b.all_performances["MSE"], g["mse"], rtol=1e-9)
        np.testing.assert_allclose(b.all_exp_tau, g["exptau"], rtol=1e-9)
        np.testing.assert_allclose(b.all_elbo, g["elbo"], rtol=1e-9)
        for n in NAMES:
            np.testing.assert_allclose(getattr(b, n), g["final/" + n], rtol=1e-7, atol=1e-12)
        np.testing.assert_allclose([b.quality(m) for m in ["loglikelihood", "BIC", "AIC", "MSE", "ELBO"]], g["quality"], rtol=1e-9)
    # toy set: 20 iterations from a random initialisation
    t = golden("toy_data.npz").case("bnmtf")
    g = golden("bnmtf_vb.npz").case("toy")
    I, J = t["R"].shape; K = L = 5
    pri = dict(alpha=1.0, beta=1.0, lambdaF=0.1 * np.ones((I, K)), lambdaS=0.1 * np.ones((K, L)), lambdaG=0.1 * np.ones((J, L)))
    b = O.BNMTFVBOracle(t["R"], t["M"], K, L, pri)
    b.tauF, b.tauS, b.tauG = np.ones((I, K)), np.ones((K, L)), np.ones((J, L))
    b.muF, b.muS, b.muG = g["init/muF"].copy(), g["init/muS"].copy(), g["init/muG"].copy()
    b.finish_initialise()
    assert abs(b.exptau - float(g["init_exptau"])) < 1e-12 * b.exptau and abs(b.elbo() - float(g["init_elbo"])) < 1e-10 * abs(float(g["init_elbo"]))
    b.run(20, orders=_orders(g, "", L))
    np.testing.assert_allclose(b.all_performances["MSE"], g["mse"], rtol=1e-8)
    np.testing.assert_allclose(b.all_elbo, g["elbo"], rtol=1e-9)
    p = b.predict(t["M"])
    np.testing.assert_allclose([p["MSE"], p["R^2"], p["Rp"]], g["final_perf"], rtol=1e-8)
