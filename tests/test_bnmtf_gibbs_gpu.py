"""BNMTF Gibbs on the device vs the reference-generated vectors and the oracle."""
import numpy as np
import pytest

from bnmtf_amd import bnmtf_gibbs_optimised
from oracle import bnmtf_oracle as O

pytestmark = pytest.mark.gpu


def _pri(c):
    return dict(alpha=float(c["alpha"]), beta=float(c["beta"]), lambdaF=c["lambdaF"], lambdaS=c["lambdaS"], lambdaG=c["lambdaG"])


def _model(c, **kw):
    b = bnmtf_gibbs_optimised(c["R"], c["M"], int(c["K"]), int(c["L"]), _pri(c), verbose=False, **kw)
    b.F, b.S, b.G, b.tau = c["F"].copy(), c["S"].copy(), c["G"].copy(), float(c["tau"])
    return b


@pytest.mark.parametrize("name", ["t5x3", "toy", "r37x29"])
def test_conditional_parameters_match_reference(golden, name):
    """tauF/muF, tauS/muS, tauG/muG (bnmtf_gibbs_optimised.py:195-211) through the hot-path kernels.
    mu tolerances are absolute, scaled by the size of the cancelling terms (fp32 products of O(|R| |F| |G|))."""
    c = golden("bnmtf_gibbs_cond.npz").case(name)
    b = _model(c)
    K, L = b.K, b.L
    M, R, F, S, G, tau = c["M"], c["R"], c["F"], c["S"], c["G"], float(c["tau"])
    assert abs(b.beta_s() - float(c["beta_s"])) <= 2e-6 * abs(float(c["beta_s"]))
    P = np.abs(F) @ np.abs(S) @ np.abs(G).T
    for k in range(K):
        np.testing.assert_allclose(b.tauF(k), c["tauF"][k], rtol=5e-6)
        sg = np.abs(S[k] @ G.T)
        sc = tau * ((M * (np.abs(R) + P)) @ sg) / c["tauF"][k]
        assert (np.abs(b.muF(c["tauF"][k], k) - c["muF"][k]) <= 3e-5 * sc + 1e-6).all()
        for l in range(L):
            assert abs(b.tauS(k, l) - c["tauS"][k, l]) <= 1e-5 * c["tauS"][k, l]
            sc = tau * (M * (np.abs(R) + P) * np.outer(np.abs(F[:, k]), np.abs(G[:, l]))).sum() / c["tauS"][k, l]
            assert abs(b.muS(c["tauS"][k, l], k, l) - c["muS"][k, l]) <= 3e-5 * sc + 1e-6
    for l in range(L):
        np.testing.assert_allclose(b.tauG(l), c["tauG"][l], rtol=5e-6)
        fs = np.abs(F @ S[:, l])
        sc = tau * ((M * (np.abs(R) + P)).T @ fs) / c["tauG"][l]
        assert (np.abs(b.muG(c["tauG"][l], l) - c["muG"][l]) <= 3e-5 * sc + 1e-6).all()
    p = b.predict_while_running()
    np.testing.assert_allclose([p["MSE"], p["R^2"]], c["perf"][:2], rtol=3e-6)
    b.all_F, b.all_S, b.all_G, b.all_tau = list(c["all_F"]), list(c["all_S"]), list(c["all_G"]), list(c["all_tau"])
    pp = b.predict(c["M_test"], 2, 3)
    np.testing.assert_allclose([pp["MSE"], pp["R^2"], pp["Rp"]], c["predict"], rtol=3e-6)
    q = [b.quality(m, 2, 3) for m in ["loglikelihood", "BIC", "AIC", "MSE", "ELBO"]]
    np.testing.assert_allclose(q, c["quality"], rtol=3e-6)


@pytest.mark.parametrize("fast", [True, False])
@pytest.mark.parametrize("name", ["toy", "r37x29"])
def test_mode_update_trajectory_matches_oracle(golden, name, fast):
    """Deterministic parity of the whole F / S / G / tau data path (draws replaced by the mode)."""
    c = golden("bnmtf_gibbs_cond.npz").case(name)
    o = O.BNMTFGibbsOracle(c["R"], c["M"], int(c["K"]), int(c["L"]), _pri(c))
    o.F, o.S, o.G, o.tau = c["F"].copy(), c["S"].copy(), c["G"].copy(), float(c["tau"])
    o.run(6, draw=False)
    b = _model(c)
    b.set_sweep_path(fast)
    b.run(6, update='mode')
    np.testing.assert_allclose(b.all_performances['MSE'], o.all_performances['MSE'], rtol=5e-4)
    np.testing.assert_allclose(b.all_tau, o.all_tau, rtol=5e-4)
    assert np.abs(b.all_S[0] - o.all_S[0]).max() < 5e-4 * max(1.0, np.abs(o.all_S[0]).max())
    assert np.abs(b.all_F[0] - o.all_F[0]).max() < 5e-4 * max(1.0, np.abs(o.all_F[0]).max())
    assert np.abs(b.all_G[0] - o.all_G[0]).max() < 5e-4 * max(1.0, np.abs(o.all_G[0]).max())


def test_gibbs_draws_follow_oracle_and_reference_bands(golden):
    t = golden("toy_data.npz").case("bnmtf")
    g = golden("gibbs_trajectories.npz").case("bnmtf")
    I, J = t["R"].shape; K = L = 5
    pri = dict(alpha=1., beta=1., lambdaF=0.1 * np.ones((I, K)), lambdaS=0.1 * np.ones((K, L)), lambdaG=0.1 * np.ones((J, L)))
    np.random.seed(3)
    b = bnmtf_gibbs_optimised(t["R"], t["M"], K, L, pri, verbose=False, seed=21)
    b.initialise('random', 'random')
    o = O.BNMTFGibbsOracle(t["R"], t["M"], K, L, pri, seed=21)
    o.F, o.S, o.G, o.tau = b.F.copy(), b.S.copy(), b.G.copy(), b.tau
    o.run(3)
    b.run(200)
    # same Philox stream: first sweep agrees element-wise up to fp32 noise
    d0 = np.abs(b.all_F[0] - o.all_F[0]) / (1e-3 + np.abs(o.all_F[0]))
    assert np.mean(d0 < 2e-3) > 0.98
    assert np.abs(b.all_S[0] - o.all_S[0]).max() < 5e-3 * np.abs(o.all_S[0]).max()
    np.testing.assert_allclose(b.all_performances['MSE'][:2], o.all_performances['MSE'][:2], rtol=2e-3)
    # converged level vs the seeded reference runs (tests/golden/make_golden.py)
    mse = np.array(b.all_performances['MSE']); ref = g["mse"]
    assert ref[:, 150:].mean(axis=1).min() * 0.9 < mse[150:].mean() < ref[:, 150:].mean(axis=1).max() * 1.1
    p = b.predict_while_running()
    assert abs(p["MSE"] - mse[-1]) < 5e-5 * mse[-1]
    assert b.all_F.shape == (200, I, K) and b.all_S.shape == (200, K, L) and b.all_G.shape == (200, J, L)


def test_kmeans_initialisation_runs():
    from bnmtf_amd.synthetic import generate_bnmtf
    R, M, _, _, _ = generate_bnmtf(60, 40, 3, 2, 0.1, seed_data=5, seed_mask=6)
    pri = dict(alpha=1., beta=1., lambdaF=0.1, lambdaS=0.1, lambdaG=0.1)
    b = bnmtf_gibbs_optimised(R, M, 3, 2, pri, verbose=False, seed=1)
    import random; random.seed(0); np.random.seed(0)
    b.initialise('random', 'kmeans')
    assert set(np.unique(b.F)) <= {0.2, 1.2} and np.allclose(b.F.sum(axis=1), 1.0 + 0.2 * 3)
    assert set(np.unique(b.G)) <= {0.2, 1.2}
    b.run(5)
    assert np.isfinite(b.all_performances['MSE']).all()


@pytest.mark.parametrize("K,L", [(6, 40), (40, 8)])
def test_factors_wider_than_32_take_the_64_wide_paths(K, L):
    """L > 32 (G stored 64 wide): the S preparation falls back to the generic q kernel and Omega is formed 64 wide; K > 32
    (F stored 64 wide): the 64-lane w kernel.  The conditional parameters still match the closed forms
    (bnmtf_gibbs_optimised.py:195-211) evaluated by the oracle."""
    rs = np.random.RandomState(11)
    I, J = 70, 95
    F0 = rs.exponential(1.0, (I, K)); S0 = rs.exponential(1.0, (K, L)); G0 = rs.exponential(1.0, (J, L))
    R = F0 @ S0 @ G0.T + rs.randn(I, J)
    M = (rs.rand(I, J) > 0.2).astype(float)
    M[rs.randint(I, size=J), np.arange(J)] = 1; M[np.arange(I), rs.randint(J, size=I)] = 1
    pri = dict(alpha=1., beta=1., lambdaF=0.3, lambdaS=0.2, lambdaG=0.1)
    b = bnmtf_gibbs_optimised(R, M, K, L, pri, verbose=False, seed=2)
    b.F, b.S, b.G, b.tau = rs.exponential(1.0, (I, K)), rs.exponential(1.0, (K, L)), rs.exponential(1.0, (J, L)), 0.7
    o = O.BNMTFGibbsOracle(R, M, K, L, pri)
    o.F, o.S, o.G, o.tau = b.F.copy(), b.S.copy(), b.G.copy(), b.tau
    for (k, l) in [(0, 0), (K // 2, L - 5), (K - 1, L - 1)]:
        t = b.tauS(k, l); to = o.tauS(k, l)
        assert abs(t - to) < 5e-6 * to
        assert abs(b.muS(t, k, l) - o.muS(to, k, l)) < 2e-4 * (abs(o.muS(to, k, l)) + 1 / np.sqrt(to))
    for l in (0, L - 7, L - 1):
        t = b.tauG(l); to = o.tauG(l)
        np.testing.assert_allclose(t, to, rtol=5e-6)
        assert np.abs(b.muG(t, l) - o.muG(to, l)).max() < 2e-4 * np.abs(o.muG(to, l)).max()
    b.run(5)
    assert np.isfinite(b.S).all() and b.S.min() >= 0 and b.S.shape == (K, L)


@pytest.mark.parametrize("I,J,K,L", [(150, 130, 32, 32), (120, 90, 32, 17), (90, 140, 9, 32)])
def test_full_width_S_system_follows_the_oracle(I, J, K, L):
    """K and / or L = 32: the packed (k <= k') x (l <= l') GEMM runs with full 64-wide tile groups and partly filled ones
    (528, 153 and 45 pairs), the chain walks 32-entry rows.  Mode updates: every step deterministic, compared with the
    oracle's sequential conditionals (bnmtf_gibbs_optimised.py:157-160)."""
    rs = np.random.RandomState(5)
    F0 = rs.exponential(1.0, (I, K)); S0 = rs.exponential(1.0, (K, L)) / np.sqrt(K * L / 25.0); G0 = rs.exponential(1.0, (J, L))
    R = F0 @ S0 @ G0.T + rs.randn(I, J)
    M = (rs.rand(I, J) > 0.15).astype(float)
    M[rs.randint(I, size=J), np.arange(J)] = 1; M[np.arange(I), rs.randint(J, size=I)] = 1
    pri = dict(alpha=1., beta=1., lambdaF=0.3, lambdaS=0.2, lambdaG=0.1)
    Fi, Si, Gi = rs.exponential(1.0, (I, K)), rs.exponential(0.2, (K, L)), rs.exponential(1.0, (J, L))
    o = O.BNMTFGibbsOracle(R, M, K, L, pri)
    o.F, o.S, o.G, o.tau = Fi.copy(), Si.copy(), Gi.copy(), 0.5
    o.run(3, draw=False)
    b = bnmtf_gibbs_optimised(R, M, K, L, pri, verbose=False, seed=4)
    b.F, b.S, b.G, b.tau = Fi.copy(), Si.copy(), Gi.copy(), 0.5
    b.run(3, update='mode')
    np.testing.assert_allclose(b.all_performances['MSE'], o.all_performances['MSE'], rtol=1e-3)
    np.testing.assert_allclose(b.all_tau, o.all_tau, rtol=1e-3)
    for name in ("all_F", "all_S", "all_G"):
        x, y = np.asarray(getattr(b, name)[0]), np.asarray(getattr(o, name)[0])
        assert np.abs(x - y).max() < 1e-3 * max(1.0, np.abs(y).max()), name


def test_device_kmeans_reproduces_the_reference():
    """bnmtf_amd.kmeans.KMeans (assignment distances and per-cluster sums on the GPU, fp64) against what the reference's
    KMeans itself produced under random.seed (tests/golden/kmeans.npz: every iteration's assignments, final centroids,
    masks, clustering_results -- incl. the cases that empty a cluster and the reference's centroid-is-a-view-of-X
    behaviour), and against the CPU oracle on a larger clustered case."""
    import os
    from bnmtf_amd.kmeans import KMeans
    from oracle.kmeans_oracle import KMeansOracle
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kmeans.npz"))
    for name in sorted(set(k.split("/")[0] for k in g.files)):
        km = KMeans(g[name + "/X"], g[name + "/M"], int(g[name + "/K"]), device=0)
        km.initialise(int(g[name + "/seed"]))
        assert np.array_equal(np.array(km.centroids), g[name + "/centroids0"]), name
        km.cluster()
        assert np.array_equal(np.array(km.assign_hist), g[name + "/assign_hist"]), name
        np.testing.assert_allclose(np.array(km.centroids), g[name + "/centroids"], rtol=1e-12, atol=1e-12, err_msg=name)
        assert np.array_equal(km.mask_centroids, g[name + "/mask_centroids"]), name
        assert np.array_equal(km.clustering_results, g[name + "/clustering_results"]), name
        d_ref = g[name + "/distances"]
        np.testing.assert_allclose(np.where(np.isfinite(km.distances), km.distances, np.nan), d_ref, rtol=1e-11, atol=1e-12, equal_nan=True, err_msg=name)
        km.close()
    rs = np.random.RandomState(5)
    n, d, K = 700, 90, 6
    centres = rs.normal(0, 4, (K, d))
    lab = rs.randint(K, size=n)
    X = centres[lab] + rs.normal(0, 0.5, (n, d))
    M = (rs.rand(n, d) > 0.3).astype(float)
    M[np.arange(n), rs.randint(d, size=n)] = 1
    for Kc, seed, nn in ((K, 11, n), (9, 3, 60), (45, 2, 300)):          # natural groups; clusters that run empty; K above one sums pass (40)
        km = KMeans(X[:nn], M[:nn], Kc, device=0); km.initialise(seed=seed); km.cluster()
        ko = KMeansOracle(X[:nn], M[:nn], Kc); ko.initialise(seed=seed); ko.cluster()
        assert np.array_equal(np.array(km.assign_hist), np.array(ko.assign_hist))
        np.testing.assert_allclose(np.array(km.centroids), np.array(ko.centroids), rtol=1e-11, atol=1e-11)
        assert np.array_equal(km.mask_centroids, ko.mask_centroids)
        if Kc == K:
            rows = km.clustering_results.argmax(axis=1)
            assert all(len(set(rows[lab == gq])) == 1 for gq in range(K))     # each true group ends in one cluster
        km.close()
