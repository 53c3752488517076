#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE.

Runs only in the build container (needs /root/reference).  The reference is
Python-2 source, so a throw-away copy is converted with lib2to3 under
/tmp/oracle (never inside this repo, never shipped to the GPU box) and
imported from there -- recipe of SURVEY.md Appendix B.  What is committed is
DATA ONLY: inputs and the reference's outputs for them.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz
"""
import contextlib
import io
import os
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
TMP = "/tmp/oracle"


def import_reference():
    if not os.path.isdir(os.path.join(TMP, "BNMTF")):
        os.makedirs(TMP, exist_ok=True)
        shutil.copytree(REF, os.path.join(TMP, "BNMTF"))
        subprocess.run("chmod -R u+w %s/BNMTF" % TMP, shell=True, check=True)
        subprocess.run([sys.executable, "-m", "lib2to3", "-w", "-n",
                        TMP + "/BNMTF/code", TMP + "/BNMTF/tests", TMP + "/BNMTF/data_toy"],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    sys.path.insert(0, TMP)
    import matplotlib
    matplotlib.use("Agg")


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()):
        yield


def rand_mask(rs, I, J, frac):
    while True:
        M = (rs.rand(I, J) >= frac).astype(float)
        if M.sum(axis=0).min() > 0 and M.sum(axis=1).min() > 0:
            return M


def bnmf_cases():
    rs = np.random.RandomState(12345)
    cases = {}
    # the reference's own 5x3 known-answer matrix (tests/code/test_bnmf_gibbs_optimised.py:144-153)
    I, J, K = 5, 3, 2
    R = np.ones((I, J)); M = np.ones((I, J)); M[0, 0] = M[2, 2] = M[3, 1] = 0
    cases["t5x3"] = dict(R=R, M=M, K=K, alpha=3.0, beta=1.0, lambdaU=2 * np.ones((I, K)), lambdaV=3 * np.ones((J, K)))
    # toy data set (config 1)
    R = np.loadtxt(REF + "/data_toy/bnmf/R.txt"); M = np.loadtxt(REF + "/data_toy/bnmf/M.txt")
    I, J = R.shape; K = 10
    cases["toy"] = dict(R=R, M=M, K=K, alpha=1.0, beta=1.0, lambdaU=0.1 * np.ones((I, K)), lambdaV=0.1 * np.ones((J, K)))
    # ragged random case, non-constant lambdas
    I, J, K = 37, 29, 6
    R = rs.exponential(1.0, (I, K)) @ rs.exponential(1.0, (J, K)).T + rs.normal(0, 1, (I, J))
    M = rand_mask(rs, I, J, 0.2)
    cases["r37x29"] = dict(R=R, M=M, K=K, alpha=2.0, beta=0.5, lambdaU=rs.uniform(0.05, 2.0, (I, K)), lambdaV=rs.uniform(0.05, 2.0, (J, K)))
    # heavily masked case (60 % missing)
    I, J, K = 40, 33, 5
    R = rs.exponential(1.0, (I, K)) @ rs.exponential(1.0, (J, K)).T + rs.normal(0, 1, (I, J))
    M = rand_mask(rs, I, J, 0.6)
    cases["r40x33"] = dict(R=R, M=M, K=K, alpha=1.0, beta=1.0, lambdaU=0.3 * np.ones((I, K)), lambdaV=0.7 * np.ones((J, K)))
    return cases, rs


def make_bnmf_cond():
    from BNMTF.code.models.bnmf_gibbs_optimised import bnmf_gibbs_optimised
    cases, rs = bnmf_cases()
    out = {}
    for name, c in cases.items():
        pri = dict(alpha=c["alpha"], beta=c["beta"], lambdaU=c["lambdaU"], lambdaV=c["lambdaV"])
        b = bnmf_gibbs_optimised(c["R"], c["M"], c["K"], pri)
        I, J, K = b.I, b.J, b.K
        if name == "t5x3":
            b.initialise("exp"); b.tau = 3.0
        else:
            b.U = rs.exponential(1.0, (I, K)); b.V = rs.exponential(1.0, (J, K)); b.tau = float(rs.gamma(2.0, 0.5))
        for k_, v in c.items():
            out["%s/%s" % (name, k_)] = np.asarray(v)
        out[name + "/U"], out[name + "/V"], out[name + "/tau"] = b.U.copy(), b.V.copy(), np.float64(b.tau)
        tU = np.array([b.tauU(k) for k in range(K)]); mU = np.array([b.muU(tU[k], k) for k in range(K)])
        tV = np.array([b.tauV(k) for k in range(K)]); mV = np.array([b.muV(tV[k], k) for k in range(K)])
        out[name + "/tauU"], out[name + "/muU"], out[name + "/tauV"], out[name + "/muV"] = tU, mU, tV, mV
        out[name + "/alpha_s"], out[name + "/beta_s"] = np.float64(b.alpha_s()), np.float64(b.beta_s())
        p = b.predict_while_running()
        out[name + "/perf"] = np.array([p["MSE"], p["R^2"], p["Rp"]])
        out[name + "/size_Omega"] = np.int64(b.size_Omega)
        out[name + "/row_counts"] = b.M.sum(axis=1).astype(np.int64)
        out[name + "/col_counts"] = b.M.sum(axis=0).astype(np.int64)
        # post-run API on a hand-made sample list (mirrors tests :240-360)
        n = 10
        b.all_U = [rs.exponential(1.0, (I, K)) for _ in range(n)]
        b.all_V = [rs.exponential(1.0, (J, K)) for _ in range(n)]
        b.all_tau = [float(rs.gamma(2.0, 0.5)) for _ in range(n)]
        out[name + "/all_U"], out[name + "/all_V"], out[name + "/all_tau"] = np.array(b.all_U), np.array(b.all_V), np.array(b.all_tau)
        eU, eV, et = b.approx_expectation(2, 3)
        out[name + "/expU"], out[name + "/expV"], out[name + "/exptau"] = eU, eV, np.float64(et)
        Mt = rand_mask(rs, I, J, 0.7) if I > 5 else np.array([[0, 0, 1], [0, 1, 0], [0, 0, 0], [1, 1, 0], [0, 0, 1]], dtype=float)
        pp = b.predict(Mt, 2, 3)
        out[name + "/M_test"] = Mt
        out[name + "/predict"] = np.array([pp["MSE"], pp["R^2"], pp["Rp"]])
        out[name + "/quality"] = np.array([b.quality(m, 2, 3) for m in ["loglikelihood", "BIC", "AIC", "MSE", "ELBO"]])
    np.savez_compressed(os.path.join(HERE, "bnmf_gibbs_cond.npz"), **out)


def make_bnmtf_cond():
    from BNMTF.code.models.bnmtf_gibbs_optimised import bnmtf_gibbs_optimised
    rs = np.random.RandomState(777)
    cases = {}
    I, J, K, L = 5, 3, 2, 4   # tests/code/test_bnmtf_gibbs_optimised.py known-answer shape
    R = np.ones((I, J)); M = np.ones((I, J)); M[0, 0] = M[2, 2] = M[3, 1] = 0
    cases["t5x3"] = dict(R=R, M=M, K=K, L=L, alpha=3.0, beta=1.0, lambdaF=2 * np.ones((I, K)), lambdaS=3 * np.ones((K, L)), lambdaG=5 * np.ones((J, L)))
    R = np.loadtxt(REF + "/data_toy/bnmtf/R.txt"); M = np.loadtxt(REF + "/data_toy/bnmtf/M.txt")
    I, J = R.shape; K = L = 5
    cases["toy"] = dict(R=R, M=M, K=K, L=L, alpha=1.0, beta=1.0, lambdaF=0.1 * np.ones((I, K)), lambdaS=0.1 * np.ones((K, L)), lambdaG=0.1 * np.ones((J, L)))
    I, J, K, L = 37, 29, 4, 3
    R = rs.exponential(1.0, (I, K)) @ rs.exponential(1.0, (K, L)) @ rs.exponential(1.0, (J, L)).T + rs.normal(0, 1, (I, J))
    M = rand_mask(rs, I, J, 0.25)
    cases["r37x29"] = dict(R=R, M=M, K=K, L=L, alpha=2.0, beta=0.5, lambdaF=rs.uniform(0.05, 2, (I, K)), lambdaS=rs.uniform(0.05, 2, (K, L)), lambdaG=rs.uniform(0.05, 2, (J, L)))
    out = {}
    for name, c in cases.items():
        pri = dict(alpha=c["alpha"], beta=c["beta"], lambdaF=c["lambdaF"], lambdaS=c["lambdaS"], lambdaG=c["lambdaG"])
        b = bnmtf_gibbs_optimised(c["R"], c["M"], c["K"], c["L"], pri)
        I, J, K, L = b.I, b.J, b.K, b.L
        if name == "t5x3":
            b.initialise("exp", "exp"); b.tau = 3.0
        else:
            b.F = rs.exponential(1.0, (I, K)); b.S = rs.exponential(1.0, (K, L)); b.G = rs.exponential(1.0, (J, L))
            b.tau = float(rs.gamma(2.0, 0.5))
        for k_, v in c.items():
            out["%s/%s" % (name, k_)] = np.asarray(v)
        out[name + "/F"], out[name + "/S"], out[name + "/G"], out[name + "/tau"] = b.F.copy(), b.S.copy(), b.G.copy(), np.float64(b.tau)
        tF = np.array([b.tauF(k) for k in range(K)]); mF = np.array([b.muF(tF[k], k) for k in range(K)])
        tS = np.array([[b.tauS(k, l) for l in range(L)] for k in range(K)])
        mS = np.array([[b.muS(tS[k, l], k, l) for l in range(L)] for k in range(K)])
        tG = np.array([b.tauG(l) for l in range(L)]); mG = np.array([b.muG(tG[l], l) for l in range(L)])
        out[name + "/tauF"], out[name + "/muF"] = tF, mF
        out[name + "/tauS"], out[name + "/muS"] = tS, mS
        out[name + "/tauG"], out[name + "/muG"] = tG, mG
        out[name + "/alpha_s"], out[name + "/beta_s"] = np.float64(b.alpha_s()), np.float64(b.beta_s())
        p = b.predict_while_running()
        out[name + "/perf"] = np.array([p["MSE"], p["R^2"], p["Rp"]])
        out[name + "/size_Omega"] = np.int64(b.size_Omega)
        n = 10
        b.all_F = [rs.exponential(1.0, (I, K)) for _ in range(n)]
        b.all_S = [rs.exponential(1.0, (K, L)) for _ in range(n)]
        b.all_G = [rs.exponential(1.0, (J, L)) for _ in range(n)]
        b.all_tau = [float(rs.gamma(2.0, 0.5)) for _ in range(n)]
        out[name + "/all_F"], out[name + "/all_S"], out[name + "/all_G"], out[name + "/all_tau"] = \
            np.array(b.all_F), np.array(b.all_S), np.array(b.all_G), np.array(b.all_tau)
        Mt = rand_mask(rs, I, J, 0.7) if I > 5 else np.array([[0, 0, 1], [0, 1, 0], [0, 0, 0], [1, 1, 0], [0, 0, 1]], dtype=float)
        pp = b.predict(Mt, 2, 3)
        out[name + "/M_test"] = Mt
        out[name + "/predict"] = np.array([pp["MSE"], pp["R^2"], pp["Rp"]])
        out[name + "/quality"] = np.array([b.quality(m, 2, 3) for m in ["loglikelihood", "BIC", "AIC", "MSE", "ELBO"]])
    np.savez_compressed(os.path.join(HERE, "bnmtf_gibbs_cond.npz"), **out)


def make_vb():
    from BNMTF.code.models.bnmf_vb_optimised import bnmf_vb_optimised
    out = {}
    R = np.loadtxt(REF + "/data_toy/bnmf/R.txt"); M = np.loadtxt(REF + "/data_toy/bnmf/M.txt")
    I, J = R.shape; K = 10
    pri = dict(alpha=1.0, beta=1.0, lambdaU=0.1 * np.ones((I, K)), lambdaV=0.1 * np.ones((J, K)))
    b = bnmf_vb_optimised(R, M, K, pri)
    b.initialise("exp")
    out["toy/init_exptau"] = np.float64(b.exptau); out["toy/init_expU"] = b.expU.copy(); out["toy/init_varU"] = b.varU.copy()
    out["toy/init_esd"] = np.float64(b.exp_square_diff())
    mse, exptau, elbo = [], [], []
    for it in range(20):
        with quiet():
            b.run(1)
        mse.append(b.all_performances["MSE"][0]); exptau.append(b.exptau); elbo.append(b.elbo())
        if it + 1 in (1, 2, 20):
            for nm in ["expU", "expV", "varU", "varV", "muU", "muV", "tauU", "tauV"]:
                out["toy/it%d/%s" % (it + 1, nm)] = getattr(b, nm).copy()
    out["toy/mse"], out["toy/exptau"], out["toy/elbo"] = np.array(mse), np.array(exptau), np.array(elbo)
    p = b.predict(M)
    out["toy/final_perf"] = np.array([p["MSE"], p["R^2"], p["Rp"]])
    out["toy/quality"] = np.array([b.quality(m) for m in ["loglikelihood", "BIC", "AIC", "MSE", "ELBO"]])
    # ragged random case with non-trivial tauUV init and one update of each kind
    rs = np.random.RandomState(4242)
    I, J, K = 31, 23, 4
    R = rs.exponential(1.0, (I, K)) @ rs.exponential(1.0, (J, K)).T + rs.normal(0, 1, (I, J))
    Mr = rand_mask(rs, I, J, 0.3)
    lU, lV = rs.uniform(0.1, 2, (I, K)), rs.uniform(0.1, 2, (J, K))
    b = bnmf_vb_optimised(R, Mr, K, dict(alpha=2.0, beta=0.5, lambdaU=lU, lambdaV=lV))
    b.initialise("exp", {"tauU": rs.uniform(0.5, 3, (I, K)), "tauV": rs.uniform(0.5, 3, (J, K))})
    out["r31x23/R"], out["r31x23/M"], out["r31x23/lambdaU"], out["r31x23/lambdaV"] = R, Mr, lU, lV
    out["r31x23/tauU0"], out["r31x23/tauV0"] = b.tauU.copy(), b.tauV.copy()
    mse, exptau, elbo = [], [], []
    for it in range(10):
        with quiet():
            b.run(1)
        mse.append(b.all_performances["MSE"][0]); exptau.append(b.exptau); elbo.append(b.elbo())
    out["r31x23/mse"], out["r31x23/exptau"], out["r31x23/elbo"] = np.array(mse), np.array(exptau), np.array(elbo)
    for nm in ["expU", "expV", "varU", "varV", "muU", "muV", "tauU", "tauV"]:
        out["r31x23/it10/%s" % nm] = getattr(b, nm).copy()
    np.savez_compressed(os.path.join(HERE, "bnmf_vb.npz"), **out)


def make_tn():
    from BNMTF.code.models.distributions.truncated_normal_vector import TN_vector_draw, TN_vector_expectation, TN_vector_variance
    from BNMTF.code.models.distributions.gamma import gamma_expectation, gamma_expectation_log, gamma_mode
    out = {}
    mus = np.array([-60., -31., -30.5, -29.5, -10., -3., -1., -0.3, 0., 0.2, 1., 4., 25., -1., 1e-3, -1e3, 5.0, -0.5])
    taus = np.array([1., 1., 1., 1., 1., 1., 2000., 1., 1., 3., 3., 0.25, 1., 0.01, 1e6, 1e-4, 0.0, 1e-12])
    G1, G2 = np.meshgrid(np.linspace(-45, 12, 58), np.array([0.05, 0.5, 1.0, 7.0, 400.0]))
    mus = np.concatenate([mus, G1.ravel()]); taus = np.concatenate([taus, G2.ravel()])
    with np.errstate(all="ignore"):
        out["mom/mu"], out["mom/tau"] = mus, taus
        out["mom/exp"] = np.array(TN_vector_expectation(mus, taus), dtype=float)
        out["mom/var"] = np.array(TN_vector_variance(mus, taus), dtype=float)
    out["gamma/ab"] = np.array([[2.0, 3.0], [3600.5, 1234.25], [1.0, 1e-3], [7.5e6, 3.1e7]])
    out["gamma/exp"] = np.array([gamma_expectation(a, b) for a, b in out["gamma/ab"]])
    out["gamma/explog"] = np.array([gamma_expectation_log(a, b) for a, b in out["gamma/ab"]])
    out["gamma/mode"] = np.array([gamma_mode(a, b) for a, b in out["gamma/ab"]])
    # distribution of the reference sampler: quantiles of n draws per (a = -mu*sqrt(tau)) regime
    n = 200000
    probs = np.concatenate([[0.0005, 0.001, 0.005], np.linspace(0.01, 0.99, 99), [0.995, 0.999, 0.9995]])
    pairs = []
    for a in [-10., -2.5, -1., 0., 0.2, 0.3, 1., 3., 3.6, 8., 40.]:
        for tau in [1.0, 37.0]:
            pairs.append((-a / np.sqrt(tau), tau))
    np.random.seed(2024)
    q, mom = [], []
    for mu, tau in pairs:
        d = np.array(TN_vector_draw(np.full(n, mu), np.full(n, tau)), dtype=float)
        q.append(np.quantile(d, probs)); mom.append([d.mean(), d.var()])
    out["draw/pairs"], out["draw/probs"], out["draw/quantiles"], out["draw/moments"], out["draw/n"] = \
        np.array(pairs), probs, np.array(q), np.array(mom), np.int64(n)
    np.savez_compressed(os.path.join(HERE, "distributions.npz"), **out)


def make_trajectories():
    """Seeded reference Gibbs runs on the toy sets: per-iteration MSE for 10 seeds."""
    from BNMTF.code.models.bnmf_gibbs_optimised import bnmf_gibbs_optimised
    from BNMTF.code.models.bnmtf_gibbs_optimised import bnmtf_gibbs_optimised
    import random
    out = {}
    R = np.loadtxt(REF + "/data_toy/bnmf/R.txt"); M = np.loadtxt(REF + "/data_toy/bnmf/M.txt")
    I, J = R.shape; K = 10
    pri = dict(alpha=1.0, beta=1.0, lambdaU=0.1 * np.ones((I, K)), lambdaV=0.1 * np.ones((J, K)))
    rs = np.random.RandomState(99)
    Mtest = (rs.rand(I, J) < 0.5).astype(float) * (1 - M)      # held-out = half of the unobserved entries
    Rtrue = np.loadtxt(REF + "/data_toy/bnmf/R_true.txt")
    mse, tau, held, init_mse = [], [], [], []
    for s in range(10):
        np.random.seed(s); random.seed(s)
        b = bnmf_gibbs_optimised(R, M, K, pri)
        b.initialise("random")
        if s == 0:
            out["bnmf/U0_seed0"], out["bnmf/V0_seed0"], out["bnmf/tau0_seed0"] = b.U.copy(), b.V.copy(), np.float64(b.tau)
        with quiet():
            b.run(200)
        mse.append(b.all_performances["MSE"]); tau.append(b.all_tau.copy())
        eU, eV, _ = b.approx_expectation(100, 2)
        held.append(((1 - M) * (Rtrue - eU @ eV.T) ** 2).sum() / (1 - M).sum())
    out["bnmf/mse"], out["bnmf/tau"], out["bnmf/heldout_mse_vs_Rtrue"] = np.array(mse), np.array(tau), np.array(held)
    R = np.loadtxt(REF + "/data_toy/bnmtf/R.txt"); M = np.loadtxt(REF + "/data_toy/bnmtf/M.txt")
    I, J = R.shape; K = L = 5
    pri = dict(alpha=1.0, beta=1.0, lambdaF=0.1 * np.ones((I, K)), lambdaS=0.1 * np.ones((K, L)), lambdaG=0.1 * np.ones((J, L)))
    mse, tau = [], []
    for s in range(6):
        np.random.seed(s); random.seed(s)
        b = bnmtf_gibbs_optimised(R, M, K, L, pri)
        b.initialise("random", "random")
        with quiet():
            b.run(200)
        mse.append(b.all_performances["MSE"]); tau.append(b.all_tau.copy())
    out["bnmtf/mse"], out["bnmtf/tau"] = np.array(mse), np.array(tau)
    np.savez_compressed(os.path.join(HERE, "gibbs_trajectories.npz"), **out)


def make_icm():
    """nmf_icm / nmtf_icm are deterministic given the initial state: whole trajectories of the reference
    (a converging run, a run with the minimum_TN clamp, and the collapse to zero from init='exp')."""
    from BNMTF.code.models.nmf_icm import nmf_icm
    from BNMTF.code.models.nmtf_icm import nmtf_icm
    out = {}
    R = np.loadtxt(REF + "/data_toy/bnmf/R.txt"); M = np.loadtxt(REF + "/data_toy/bnmf/M.txt")
    I, J = R.shape
    for name, K, lam, init, mtn, seed, iters in [("nmf_conv", 10, 1.0, "random", 0.0, 3, 12), ("nmf_min", 10, 0.1, "random", 0.1, 7, 12),
                                                 ("nmf_collapse", 10, 0.1, "exp", 0.0, None, 3)]:
        pri = dict(alpha=1.0, beta=1.0, lambdaU=lam * np.ones((I, K)), lambdaV=lam * np.ones((J, K)))
        if seed is not None:
            np.random.seed(seed)
        b = nmf_icm(R, M, K, pri)
        b.initialise(init)
        out[name + "/U0"], out[name + "/V0"], out[name + "/tau0"] = b.U.copy(), b.V.copy(), np.array(b.tau)
        with quiet(), np.errstate(all="ignore"):
            b.run(iters, minimum_TN=mtn)
            out[name + "/quality"] = np.array([b.quality(m) for m in ["loglikelihood", "BIC", "AIC", "MSE", "ELBO"]])
            Mp = rand_mask(np.random.RandomState(3), I, J, 0.5)
            pr = b.predict(Mp)
        out[name + "/U"], out[name + "/V"], out[name + "/all_tau"] = b.U.copy(), b.V.copy(), b.all_tau.copy()
        out[name + "/mse"] = np.array(b.all_performances["MSE"]); out[name + "/r2"] = np.array(b.all_performances["R^2"])
        out[name + "/rp"] = np.array(b.all_performances["Rp"])
        out[name + "/cfg"] = np.array([K, lam, mtn, iters])
        out[name + "/Mpred"], out[name + "/pred"] = Mp, np.array([pr["MSE"], pr["R^2"], pr["Rp"]])
    R = np.loadtxt(REF + "/data_toy/bnmtf/R.txt"); M = np.loadtxt(REF + "/data_toy/bnmtf/M.txt")
    I, J = R.shape; K = L = 5
    for name, lam, mtn, seed, iters in [("nmtf_conv", 1.0, 0.0, 11, 10), ("nmtf_min", 0.1, 0.05, 12, 10)]:
        pri = dict(alpha=1.0, beta=1.0, lambdaF=lam * np.ones((I, K)), lambdaS=lam * np.ones((K, L)), lambdaG=lam * np.ones((J, L)))
        np.random.seed(seed)
        b = nmtf_icm(R, M, K, L, pri)
        b.initialise("random", "random")
        out[name + "/F0"], out[name + "/S0"], out[name + "/G0"], out[name + "/tau0"] = b.F.copy(), b.S.copy(), b.G.copy(), np.array(b.tau)
        with quiet(), np.errstate(all="ignore"):
            b.run(iters, minimum_TN=mtn)
            out[name + "/quality"] = np.array([b.quality(m) for m in ["loglikelihood", "BIC", "AIC", "MSE", "ELBO"]])
        out[name + "/F"], out[name + "/S"], out[name + "/G"], out[name + "/all_tau"] = b.F.copy(), b.S.copy(), b.G.copy(), b.all_tau.copy()
        out[name + "/mse"] = np.array(b.all_performances["MSE"]); out[name + "/r2"] = np.array(b.all_performances["R^2"])
        out[name + "/rp"] = np.array(b.all_performances["Rp"])
        out[name + "/cfg"] = np.array([K, lam, mtn, iters])
    np.savez_compressed(os.path.join(HERE, "icm.npz"), **out)


def make_bnmtf_vb():
    """bnmtf_vb_optimised (tests/code/test_bnmtf_vb_optimised.py's inputs, the toy set, a ragged case): single updates
    from hand-set states, and whole runs.  run() shuffles its three index lists with Python's `random.shuffle`
    (bnmtf_vb_optimised.py:52,172-186): `random.seed` before each run pins the orders, and they are stored too."""
    import itertools
    import random
    from BNMTF.code.models.bnmtf_vb_optimised import bnmtf_vb_optimised
    out = {}
    rs = np.random.RandomState(777)

    def record_updates(tag, b):
        """update_F(k) / update_S(k,l) / update_G(l) each from the SAME state (restored in between) + esd, elbo pieces"""
        names = ["muF", "tauF", "expF", "varF", "muS", "tauS", "expS", "varS", "muG", "tauG", "expG", "varG"]
        state = {n: getattr(b, n).copy() for n in names}
        for n in names:
            out["%s/state/%s" % (tag, n)] = state[n]
        out[tag + "/state/exptau"] = np.float64(b.exptau)
        out[tag + "/esd"] = np.float64(b.exp_square_diff())
        tF, mF = np.zeros((b.I, b.K)), np.zeros((b.I, b.K))
        for k in range(b.K):
            b.update_F(k); tF[:, k], mF[:, k] = b.tauF[:, k], b.muF[:, k]
            b.tauF, b.muF = state["tauF"].copy(), state["muF"].copy()
        tS, mS = np.zeros((b.K, b.L)), np.zeros((b.K, b.L))
        for k, l in itertools.product(range(b.K), range(b.L)):
            b.update_S(k, l); tS[k, l], mS[k, l] = b.tauS[k, l], b.muS[k, l]
            b.tauS, b.muS = state["tauS"].copy(), state["muS"].copy()
        tG, mG = np.zeros((b.J, b.L)), np.zeros((b.J, b.L))
        for l in range(b.L):
            b.update_G(l); tG[:, l], mG[:, l] = b.tauG[:, l], b.muG[:, l]
            b.tauG, b.muG = state["tauG"].copy(), state["muG"].copy()
        for n, v in (("tauF", tF), ("muF", mF), ("tauS", tS), ("muS", mS), ("tauG", tG), ("muG", mG)):
            out["%s/upd/%s" % (tag, n)] = v

    def record_run(tag, b, iters, seed):
        random.seed(seed)
        state = random.getstate()
        orders = []
        for it in range(iters):          # the three shuffles of every iteration, as run() will make them
            oS = list(itertools.product(range(b.K), range(b.L))); random.shuffle(oS)
            oF = list(range(b.K)); random.shuffle(oF)
            oG = list(range(b.L)); random.shuffle(oG)
            orders.append((oS, oF, oG))
        random.setstate(state)
        mse, exptau, elbo = [], [], []
        for it in range(iters):
            with quiet():
                b.run(1)
            mse.append(b.all_performances["MSE"][0]); exptau.append(b.exptau); elbo.append(b.elbo())
        out[tag + "/seed"] = np.int64(seed)
        out[tag + "/order_S"] = np.array([[k * b.L + l for k, l in o[0]] for o in orders], dtype=np.int32)
        out[tag + "/order_F"] = np.array([o[1] for o in orders], dtype=np.int32)
        out[tag + "/order_G"] = np.array([o[2] for o in orders], dtype=np.int32)
        out[tag + "/mse"], out[tag + "/exptau"], out[tag + "/elbo"] = np.array(mse), np.array(exptau), np.array(elbo)
        for n in ["muF", "tauF", "expF", "varF", "muS", "tauS", "expS", "varS", "muG", "tauG", "expG", "varG"]:
            out["%s/final/%s" % (tag, n)] = getattr(b, n).copy()
        p = b.predict(b.M)
        out[tag + "/final_perf"] = np.array([p["MSE"], p["R^2"], p["Rp"]])
        out[tag + "/quality"] = np.array([b.quality(m) for m in ["loglikelihood", "BIC", "AIC", "MSE", "ELBO"]])

    # the reference tests' 5 x 3 matrix and hand-set state (test_bnmtf_vb_optimised.py:283-300)
    I, J, K, L = 5, 3, 2, 4
    R = np.ones((I, J)); M = np.ones((I, J)); M[0, 0] = M[2, 2] = M[3, 1] = 0
    pri = dict(alpha=3, beta=1, lambdaF=2 * np.ones((I, K)), lambdaS=3 * np.ones((K, L)), lambdaG=4 * np.ones((J, L)))
    b = bnmtf_vb_optimised(R, M, K, L, pri)
    b.muF = rs.uniform(0.1, 2, (I, K)); b.muS = rs.uniform(0.1, 2, (K, L)); b.muG = rs.uniform(0.1, 2, (J, L))
    b.tauF = rs.uniform(0.5, 3, (I, K)); b.tauS = rs.uniform(0.5, 3, (K, L)); b.tauG = rs.uniform(0.5, 3, (J, L))
    b.expF = 1. / pri["lambdaF"]; b.expS = 1. / pri["lambdaS"]; b.expG = 1. / pri["lambdaG"]
    b.varF = np.ones((I, K)) * 2; b.varS = np.ones((K, L)) * 3; b.varG = np.ones((J, L)) * 4
    b.exptau = 3.
    record_updates("t5x3", b)
    # a ragged random case, non-constant lambdas, random q state
    I, J, K, L = 33, 27, 5, 4
    R = rs.exponential(1.0, (I, K)) @ rs.exponential(1.0, (K, L)) @ rs.exponential(1.0, (J, L)).T + rs.normal(0, 1, (I, J))
    Mr = rand_mask(rs, I, J, 0.25)
    pri = dict(alpha=2.0, beta=0.5, lambdaF=rs.uniform(0.1, 2, (I, K)), lambdaS=rs.uniform(0.1, 2, (K, L)), lambdaG=rs.uniform(0.1, 2, (J, L)))
    out["r33x27/R"], out["r33x27/M"] = R, Mr
    for n in ("lambdaF", "lambdaS", "lambdaG"):
        out["r33x27/" + n] = pri[n]
    b = bnmtf_vb_optimised(R, Mr, K, L, pri)
    b.initialise("exp", "exp", {"tauF": rs.uniform(0.5, 3, (I, K)), "tauS": rs.uniform(0.5, 3, (K, L)), "tauG": rs.uniform(0.5, 3, (J, L))})
    out["r33x27/init/tauF"], out["r33x27/init/tauS"], out["r33x27/init/tauG"] = b.tauF.copy(), b.tauS.copy(), b.tauG.copy()
    out["r33x27/init_exptau"] = np.float64(b.exptau)
    record_updates("r33x27", b)
    record_run("r33x27", b, 10, seed=11)
    # the toy set (data_toy/bnmtf: 100 x 80, K = L = 5), init exp / exp
    R = np.loadtxt(REF + "/data_toy/bnmtf/R.txt"); M = np.loadtxt(REF + "/data_toy/bnmtf/M.txt")
    I, J = R.shape; K = L = 5
    pri = dict(alpha=1.0, beta=1.0, lambdaF=0.1 * np.ones((I, K)), lambdaS=0.1 * np.ones((K, L)), lambdaG=0.1 * np.ones((J, L)))
    b = bnmtf_vb_optimised(R, M, K, L, pri)
    np.random.seed(5)
    b.initialise("random", "random")
    for n in ("muF", "muS", "muG"):
        out["toy/init/" + n] = getattr(b, n).copy()
    out["toy/init_exptau"] = np.float64(b.exptau); out["toy/init_esd"] = np.float64(b.exp_square_diff())
    out["toy/init_elbo"] = np.float64(b.elbo())
    record_run("toy", b, 20, seed=3)
    np.savez_compressed(os.path.join(HERE, "bnmtf_vb.npz"), **out)


def make_masks():
    """code/cross_validation/mask.py under random.seed: the masks / folds the reference draws (the vectorised
    bnmtf_amd.cross_validation.mask consumes the same random stream)."""
    import random
    sys.path.insert(0, TMP + "/BNMTF/code/cross_validation")
    import mask as ref_mask
    out = {}
    random.seed(1); out["generate_M"] = ref_mask.generate_M(7, 5, 0.3)
    M = np.ones((9, 6)); M[0, 1] = M[3, 3] = M[8, 5] = M[4, 0] = 0
    out["M"] = M
    random.seed(2); out["folds"] = np.array(ref_mask.compute_folds(9, 6, 4, M))
    random.seed(3); out["folds_attempts"] = np.array(ref_mask.compute_folds_attempts(9, 6, 3, 50, M))
    out["Ms"] = np.array(ref_mask.compute_Ms(list(out["folds"])))
    random.seed(4); tr, te = ref_mask.generate_M_from_M(M, 0.3); out["split_train"], out["split_test"] = tr, te
    random.seed(5); tr, te = ref_mask.try_generate_M_from_M(M, 0.4, 20); out["try_train"], out["try_test"] = tr, te
    random.seed(6); rows = ref_mask.compute_crossval_folds_rows_attempts(M, 5, 3, 50)
    out["rows_train"] = np.array([a for a, b in rows]); out["rows_test"] = np.array([b for a, b in rows])
    random.seed(7); cols = ref_mask.compute_crossval_folds_columns_attempts(M, 4, 2, 50)
    out["cols_train"] = np.array([a for a, b in cols]); out["cols_test"] = np.array([b for a, b in cols])
    out["inverse"] = ref_mask.calc_inverse_M(M)
    out["nz"] = np.array(ref_mask.nonzero_indices(M))
    np.savez_compressed(os.path.join(HERE, "masks.npz"), **out)


def make_kmeans():
    """code/models/kmeans/kmeans.py under random.seed: starting centroids, every iteration's assignments, the final
    centroids / masks / distances / clustering_results -- on clustered data with missing values, on a case that empties a
    cluster ('singleton' refill, kmeans.py:137-152), on one with an unobserved column and a point that shares no
    coordinate with some centroid, and on the toy BNMTF matrix as initialise(init_FG='kmeans') uses it (rows and columns)."""
    from BNMTF.code.models.kmeans.kmeans import KMeans
    rs = np.random.RandomState(77)
    cases = {}
    # three well separated groups, 25 % missing
    cen = np.array([[0., 0, 0, 0, 0, 0], [5, 5, 5, 5, 5, 5], [-4, 6, -4, 6, -4, 6]])
    X = np.vstack([c + 0.3 * rs.randn(14, 6) for c in cen]); M = (rs.rand(*X.shape) > 0.25).astype(float)
    M[np.arange(len(X)), rs.randint(6, size=len(X))] = 1
    cases["groups"] = (X, M, 3, 5)
    # more clusters than natural groups and duplicated points: clusters run empty and are refilled
    X = np.vstack([np.tile(rs.randn(1, 4), (6, 1)), 4 + 0.01 * rs.randn(5, 4), rs.randn(4, 4) * 3]); M = np.ones(X.shape)
    M[1, 0] = M[7, 2] = M[12, 3] = 0
    cases["empties"] = (X, M, 6, 3)
    # an unobserved column (dropped), sparse rows, a point that overlaps few centroid coordinates
    X = rs.randn(20, 7) * 2; M = (rs.rand(20, 7) > 0.55).astype(float); M[:, 4] = 0
    M[np.arange(20), rs.randint(4, size=20)] = 1
    cases["sparse"] = (X, M, 4, 11)
    R = np.loadtxt(REF + "/data_toy/bnmtf/R.txt"); Mt = np.loadtxt(REF + "/data_toy/bnmtf/M.txt")
    cases["toy_rows"] = (R, Mt, 5, 0)
    cases["toy_cols"] = (R.T.copy(), Mt.T.copy(), 5, 1)
    out = {}
    for name, (X, M, K, seed) in cases.items():
        with quiet():
            km = KMeans(X, M, K)
            km.initialise(seed)
            out[name + "/X"] = X; out[name + "/M"] = M; out[name + "/K"] = np.array(K); out[name + "/seed"] = np.array(seed)
            out[name + "/centroids0"] = np.array(km.centroids, dtype=float)
            # the loop of cluster() (kmeans.py:70-84), keeping every iteration's state
            hist = []
            iteration = 1; change = True
            while change:
                iteration += 1
                change = km.assignment()
                km.update()
                hist.append(np.array(km.cluster_assignments, dtype=int).copy())
                if iteration >= 200:
                    break
            km.create_matrix()
        out[name + "/assign_hist"] = np.array(hist)
        out[name + "/centroids"] = np.array([np.asarray(c, dtype=float) for c in km.centroids])
        out[name + "/mask_centroids"] = np.array(km.mask_centroids, dtype=float)
        out[name + "/distances"] = np.array(km.distances, dtype=float)
        out[name + "/clustering_results"] = km.clustering_results
    np.savez_compressed(os.path.join(HERE, "kmeans.npz"), **out)


def make_gdsc():
    """data_drug_sensitivity/gdsc/load_data.py:15-54 on the reference's own ic50 file: what load_gdsc returns for the
    first 12 cell lines (the excerpt travels as a fixture: header + 12 lines of the data file) and summary numbers of the
    whole file (checked where /root/reference is present)."""
    sys.path.insert(0, TMP + "/BNMTF/data_drug_sensitivity/gdsc")
    import load_data as ref_load
    src = REF + "/data_drug_sensitivity/gdsc/ic50_excl_empty_filtered_cell_lines_drugs.txt"
    lines = open(src, "r").readlines()
    excerpt = os.path.join(HERE, "gdsc_excerpt.txt")
    with open(excerpt, "w") as f:
        f.writelines(lines[:13])
    out = {}
    X, X_min, M, drugs, cells, cancers, tissues = ref_load.load_gdsc(location=excerpt)
    out["ex/X"], out["ex/X_min"], out["ex/M"] = X, X_min, M
    out["ex/n_drugs"] = np.array(len(drugs)); out["ex/n_cells"] = np.array(len(cells))
    out["ex/negated"] = ref_load.negate_gdsc(X, M)
    X, X_min, M, drugs, cells, cancers, tissues = ref_load.load_gdsc()
    out["full/shape"] = np.array(X.shape); out["full/M_sum"] = np.array(M.sum()); out["full/X_sum"] = np.array(X.sum())
    out["full/X_min_sum"] = np.array(X_min.sum()); out["full/minimum"] = np.array(X.min())
    out["full/row_obs"] = M.sum(axis=1); out["full/col_obs"] = M.sum(axis=0)
    ii = np.array([0, 5, 17, 100, 333, 621]); jj = np.array([0, 3, 77, 137, 50, 9])
    out["full/ii"], out["full/jj"] = ii, jj
    out["full/X_at"], out["full/M_at"], out["full/X_min_at"] = X[ii, jj], M[ii, jj], X_min[ii, jj]
    np.savez_compressed(os.path.join(HERE, "gdsc.npz"), **out)


def make_wide_rank():
    """Ranks above 64 (round 6: the device runs them as column blocks): the reference's conditional parameters for every column
    of a K = 96 model from a random state, and a whole nmf_icm trajectory at K = 70 (deterministic)."""
    from BNMTF.code.models.bnmf_gibbs_optimised import bnmf_gibbs_optimised
    from BNMTF.code.models.nmf_icm import nmf_icm
    rs = np.random.RandomState(2026)
    out = {}
    I, J, K = 57, 44, 96
    R = rs.exponential(1.0, (I, 8)) @ rs.exponential(1.0, (J, 8)).T + rs.normal(0, 1, (I, J))
    M = rand_mask(rs, I, J, 0.25)
    pri = dict(alpha=2.0, beta=0.5, lambdaU=rs.uniform(0.05, 2.0, (I, K)), lambdaV=rs.uniform(0.05, 2.0, (J, K)))
    b = bnmf_gibbs_optimised(R, M, K, pri)
    b.U = rs.exponential(0.3, (I, K)); b.V = rs.exponential(0.3, (J, K)); b.tau = 0.8
    for k_, v in dict(R=R, M=M, K=K, U=b.U, V=b.V, tau=b.tau, **pri).items():
        out["k96/" + k_] = np.asarray(v)
    tU = np.array([b.tauU(k) for k in range(K)]); mU = np.array([b.muU(tU[k], k) for k in range(K)])
    tV = np.array([b.tauV(k) for k in range(K)]); mV = np.array([b.muV(tV[k], k) for k in range(K)])
    out["k96/tauU"], out["k96/muU"], out["k96/tauV"], out["k96/muV"] = tU, mU, tV, mV
    out["k96/beta_s"] = np.float64(b.beta_s())
    p = b.predict_while_running()
    out["k96/perf"] = np.array([p["MSE"], p["R^2"], p["Rp"]])
    I, J, K = 60, 48, 70
    R = rs.exponential(1.0, (I, 6)) @ rs.exponential(1.0, (J, 6)).T + rs.normal(0, 1, (I, J))
    M = rand_mask(rs, I, J, 0.15)
    pri = dict(alpha=1.0, beta=1.0, lambdaU=np.ones((I, K)), lambdaV=np.ones((J, K)))
    np.random.seed(5)
    c = nmf_icm(R, M, K, pri)
    c.initialise("random")
    out["icm70/R"], out["icm70/M"], out["icm70/U0"], out["icm70/V0"], out["icm70/tau0"] = R, M, c.U.copy(), c.V.copy(), np.array(c.tau)
    with quiet(), np.errstate(all="ignore"):
        c.run(6, minimum_TN=0.01)
    out["icm70/U"], out["icm70/V"], out["icm70/all_tau"] = c.U.copy(), c.V.copy(), c.all_tau.copy()
    out["icm70/mse"] = np.array(c.all_performances["MSE"])
    # the variational model at K = 70 (deterministic from init='exp'): five iterations of the reference
    from BNMTF.code.models.bnmf_vb_optimised import bnmf_vb_optimised
    I, J, K = 52, 41, 70
    R = rs.exponential(1.0, (I, 6)) @ rs.exponential(1.0, (J, 6)).T + rs.normal(0, 1, (I, J))
    M = rand_mask(rs, I, J, 0.2)
    pri = dict(alpha=1.0, beta=1.0, lambdaU=rs.uniform(0.5, 2.0, (I, K)), lambdaV=rs.uniform(0.5, 2.0, (J, K)))
    v = bnmf_vb_optimised(R, M, K, pri)
    v.initialise("exp")
    out["vb70/R"], out["vb70/M"], out["vb70/lambdaU"], out["vb70/lambdaV"] = R, M, pri["lambdaU"], pri["lambdaV"]
    out["vb70/exptau0"] = np.float64(v.exptau)
    elbos = []
    with quiet(), np.errstate(all="ignore"):
        for _ in range(5):
            v.run(1)
            elbos.append(v.elbo())
    # (run(1) five times: all_* hold the last call only; the trajectory is collected call by call)
    v2 = bnmf_vb_optimised(R, M, K, pri)
    v2.initialise("exp")
    with quiet(), np.errstate(all="ignore"):
        v2.run(5)
    out["vb70/mse"] = np.array(v2.all_performances["MSE"]); out["vb70/exptau"] = np.array(v2.all_exp_tau)
    out["vb70/elbo"] = np.array(elbos)
    out["vb70/expU"], out["vb70/expV"], out["vb70/varU"], out["vb70/tauU"], out["vb70/muV"] = v2.expU.copy(), v2.expV.copy(), v2.varU.copy(), v2.tauU.copy(), v2.muV.copy()
    np.savez_compressed(os.path.join(HERE, "wide_rank.npz"), **out)


def make_wide_tri():
    """The tri-factorisation with K or L above 64 (round 6: the device runs it as blocks of S): the reference's conditional
    parameters from random states (every column of F and G, a sample of the entries of S) and whole nmtf_icm trajectories
    (deterministic) -- one with a single column block of S, one with two row and two column blocks."""
    from BNMTF.code.models.bnmtf_gibbs_optimised import bnmtf_gibbs_optimised
    from BNMTF.code.models.nmtf_icm import nmtf_icm
    rs = np.random.RandomState(4052)
    out = {}
    for tag, (I, J, K, L) in (("k70l5", (48, 37, 70, 5)), ("k6l66", (41, 52, 6, 66)), ("k70l66", (45, 39, 70, 66))):
        R = rs.exponential(1.0, (I, 5)) @ rs.exponential(1.0, (5, 4)) @ rs.exponential(1.0, (J, 4)).T + rs.normal(0, 1, (I, J))
        M = rand_mask(rs, I, J, 0.2)
        pri = dict(alpha=2.0, beta=0.5, lambdaF=rs.uniform(0.05, 2.0, (I, K)), lambdaS=rs.uniform(0.05, 2.0, (K, L)), lambdaG=rs.uniform(0.05, 2.0, (J, L)))
        b = bnmtf_gibbs_optimised(R, M, K, L, pri)
        b.F = rs.exponential(0.3, (I, K)); b.S = rs.exponential(0.3, (K, L)); b.G = rs.exponential(0.3, (J, L)); b.tau = 0.7
        for k_, v in dict(R=R, M=M, F=b.F, S=b.S, G=b.G, tau=b.tau, **pri).items():
            out[tag + "/" + k_] = np.asarray(v)
        tF = np.array([b.tauF(k) for k in range(K)]); mF = np.array([b.muF(tF[k], k) for k in range(K)])
        tG = np.array([b.tauG(l) for l in range(L)]); mG = np.array([b.muG(tG[l], l) for l in range(L)])
        kl = np.array([(k, l) for k in range(K) for l in range(L)])
        kl = kl[rs.permutation(len(kl))[:60]]
        tS = np.array([b.tauS(k, l) for k, l in kl]); mS = np.array([b.muS(tS[i], k, l) for i, (k, l) in enumerate(kl)])
        out[tag + "/tauF"], out[tag + "/muF"], out[tag + "/tauG"], out[tag + "/muG"] = tF, mF, tG, mG
        out[tag + "/kl"], out[tag + "/tauS"], out[tag + "/muS"] = kl, tS, mS
        out[tag + "/beta_s"] = np.float64(b.beta_s())
        p = b.predict_while_running()
        out[tag + "/perf"] = np.array([p["MSE"], p["R^2"], p["Rp"]])
    for tag, (I, J, K, L, its, mtn, seed) in (("icm_k70l4", (50, 40, 70, 4, 5, 0.001, 21)), ("icm_k70l66", (44, 38, 70, 66, 3, 0.001, 22))):
        R = rs.exponential(1.0, (I, 5)) @ rs.exponential(1.0, (5, 4)) @ rs.exponential(1.0, (J, 4)).T + rs.normal(0, 1, (I, J))
        M = rand_mask(rs, I, J, 0.15)
        pri = dict(alpha=1.0, beta=1.0, lambdaF=0.1 * np.ones((I, K)), lambdaS=0.1 * np.ones((K, L)), lambdaG=0.1 * np.ones((J, L)))
        np.random.seed(seed)
        c = nmtf_icm(R, M, K, L, pri)
        c.initialise("exp", "exp")
        # (a start on the data's scale: from the priors' own draws every entry falls to minimum_TN in the first sweep)
        a0 = (R[M > 0].mean() / (K * L)) ** (1.0 / 3.0)
        c.F = rs.exponential(a0, (I, K)); c.S = rs.exponential(a0, (K, L)); c.G = rs.exponential(a0, (J, L))
        c.tau = (c.alpha_s() - 1.0) / c.beta_s()
        for k_, v in dict(R=R, M=M, F0=c.F.copy(), S0=c.S.copy(), G0=c.G.copy(), tau0=np.array(c.tau), minimum_TN=np.array(mtn), iterations=np.array(its)).items():
            out[tag + "/" + k_] = np.asarray(v)
        with quiet(), np.errstate(all="ignore"):
            c.run(its, minimum_TN=mtn)
        out[tag + "/F"], out[tag + "/S"], out[tag + "/G"], out[tag + "/all_tau"] = c.F.copy(), c.S.copy(), c.G.copy(), np.array(c.all_tau)
        out[tag + "/mse"] = np.array(c.all_performances["MSE"])
    np.savez_compressed(os.path.join(HERE, "wide_tri.npz"), **out)


def make_toy_data():
    """The reference's toy inputs (data files its own tests/experiments hold) as one fixture."""
    out = {}
    for m in ["R", "M", "U", "V", "R_true"]:
        out["bnmf/" + m] = np.loadtxt(REF + "/data_toy/bnmf/%s.txt" % m)
    for m in ["R", "M", "F", "S", "G", "R_true"]:
        out["bnmtf/" + m] = np.loadtxt(REF + "/data_toy/bnmtf/%s.txt" % m)
    np.savez_compressed(os.path.join(HERE, "toy_data.npz"), **out)


if __name__ == "__main__":
    import_reference()
    which = sys.argv[1:] or ["toy", "bnmf", "bnmtf", "vb", "tn", "traj", "icm", "trivb", "masks", "kmeans", "gdsc", "wide", "widetri"]
    if "toy" in which: make_toy_data()
    if "bnmf" in which: make_bnmf_cond()
    if "bnmtf" in which: make_bnmtf_cond()
    if "vb" in which: make_vb()
    if "tn" in which: make_tn()
    if "traj" in which: make_trajectories()
    if "icm" in which: make_icm()
    if "trivb" in which: make_bnmtf_vb()
    if "masks" in which: make_masks()
    if "kmeans" in which: make_kmeans()
    if "gdsc" in which: make_gdsc()
    if "wide" in which: make_wide_rank()
    if "widetri" in which: make_wide_tri()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
