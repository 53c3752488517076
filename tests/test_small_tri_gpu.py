"""The one-launch path of the tri-factorisation (csrc/kernel_small.hip, TRI instantiations): a block per model runs the F sweep,
the K L entries of S, the G sweep, tau and the metrics of every iteration of a run() call.  Checked against the oracle, against
the multi-launch path, and batch against solo."""
import numpy as np
import pytest

import bnmtf_amd
from bnmtf_amd import bnmtf_gibbs_optimised
from bnmtf_amd.synthetic import generate_bnmtf
from oracle import bnmtf_oracle as O

pytestmark = pytest.mark.gpu

PRI = dict(alpha=1., beta=1., lambdaF=0.1, lambdaS=0.1, lambdaG=0.1)


def _pri(c):
    return dict(alpha=float(c["alpha"]), beta=float(c["beta"]), lambdaF=c["lambdaF"], lambdaS=c["lambdaS"], lambdaG=c["lambdaG"])


@pytest.mark.parametrize("name", ["t5x3", "toy", "r37x29"])
def test_mode_update_trajectory_matches_oracle_on_the_small_path(golden, name):
    """Deterministic parity of the whole F / S / G / tau data path of the one-launch kernel (draws replaced by the mode), on the
    reference-generated states (bnmtf_gibbs_optimised.py:138-180)."""
    c = golden("bnmtf_gibbs_cond.npz").case(name)
    o = O.BNMTFGibbsOracle(c["R"], c["M"], int(c["K"]), int(c["L"]), _pri(c))
    o.F, o.S, o.G, o.tau = c["F"].copy(), c["S"].copy(), c["G"].copy(), float(c["tau"])
    o.run(6, draw=False)
    b = bnmtf_gibbs_optimised(c["R"], c["M"], int(c["K"]), int(c["L"]), _pri(c), verbose=False)
    b.F, b.S, b.G, b.tau = c["F"].copy(), c["S"].copy(), c["G"].copy(), float(c["tau"])
    assert b.is_small() and "small[" in b.describe()
    b.run(6, update='mode')
    np.testing.assert_allclose(b.all_performances['MSE'], o.all_performances['MSE'], rtol=5e-4)
    np.testing.assert_allclose(b.all_tau, o.all_tau, rtol=5e-4)
    for it in (0, 5):
        assert np.abs(b.all_S[it] - o.all_S[it]).max() < 5e-4 * (it + 1) * max(1.0, np.abs(o.all_S[it]).max())
        assert np.abs(b.all_F[it] - o.all_F[it]).max() < 5e-4 * (it + 1) * max(1.0, np.abs(o.all_F[it]).max())
        assert np.abs(b.all_G[it] - o.all_G[it]).max() < 5e-4 * (it + 1) * max(1.0, np.abs(o.all_G[it]).max())
    np.testing.assert_allclose(b.F, b.all_F[-1]); np.testing.assert_allclose(b.S, b.all_S[-1]); np.testing.assert_allclose(b.G, b.all_G[-1])


def test_draws_follow_the_oracle_with_the_same_philox_stream(golden):
    t = golden("toy_data.npz").case("bnmtf")
    g = golden("gibbs_trajectories.npz").case("bnmtf")
    I, J = t["R"].shape; K = L = 5
    pri = dict(alpha=1., beta=1., lambdaF=0.1 * np.ones((I, K)), lambdaS=0.1 * np.ones((K, L)), lambdaG=0.1 * np.ones((J, L)))
    np.random.seed(3)
    b = bnmtf_gibbs_optimised(t["R"], t["M"], K, L, pri, verbose=False, seed=21)
    b.initialise('random', 'random')
    assert b.is_small()
    o = O.BNMTFGibbsOracle(t["R"], t["M"], K, L, pri, seed=21)
    o.F, o.S, o.G, o.tau = b.F.copy(), b.S.copy(), b.G.copy(), b.tau
    o.run(3)
    b.run(200)
    # same Philox stream: the first sweeps agree element-wise up to fp32 noise
    d0 = np.abs(b.all_F[0] - o.all_F[0]) / (1e-3 + np.abs(o.all_F[0]))
    assert np.mean(d0 < 2e-3) > 0.98
    assert np.abs(b.all_S[0] - o.all_S[0]).max() < 5e-3 * np.abs(o.all_S[0]).max()
    dG = np.abs(b.all_G[0] - o.all_G[0]) / (1e-3 + np.abs(o.all_G[0]))
    assert np.mean(dG < 5e-3) > 0.95
    np.testing.assert_allclose(b.all_performances['MSE'][:2], o.all_performances['MSE'][:2], rtol=2e-3)
    np.testing.assert_allclose(b.all_tau[:2], o.all_tau[:2], rtol=2e-3)
    # converged level vs the seeded reference runs (tests/golden/make_golden.py)
    mse = np.array(b.all_performances['MSE']); ref = g["mse"]
    assert ref[:, 150:].mean(axis=1).min() * 0.9 < mse[150:].mean() < ref[:, 150:].mean(axis=1).max() * 1.1
    p = b.predict_while_running()
    assert abs(p["MSE"] - mse[-1]) < 5e-5 * mse[-1]
    assert b.all_F.shape == (200, I, K) and b.all_S.shape == (200, K, L) and b.all_G.shape == (200, J, L)


@pytest.mark.parametrize("I,J,K,L,miss", [(100, 80, 5, 5, 0.1), (150, 130, 10, 7, 0.2), (622, 138, 10, 10, 0.19), (90, 140, 9, 32, 0.1), (120, 90, 32, 17, 0.15)])
def test_the_first_iterations_agree_with_the_multi_launch_path(I, J, K, L, miss):
    """Same conditionals, same Philox keying: the two paths draw the same chain up to fp32 summation order (a draw near an
    acceptance boundary may take another candidate: a few entries)."""
    R, M, _, _, _ = generate_bnmtf(I, J, K, L, miss, seed_data=I + K, seed_mask=J + L)
    runs = []
    for small in (True, False):
        np.random.seed(4)
        b = bnmtf_gibbs_optimised(R, M, K, L, PRI, seed=11, verbose=False)
        b.initialise('random', 'random')
        b.set_small_path('always' if small else False)
        assert b.is_small() == small
        b.run(2)
        runs.append(b)
    a, c = runs
    for x, y in ((a.all_F[0], c.all_F[0]), (a.all_S[0], c.all_S[0]), (a.all_G[0], c.all_G[0])):
        d = np.abs(x - y) / (1e-3 + np.abs(y))
        assert np.mean(d < 5e-3) > 0.97, (np.mean(d < 5e-3), d.max())
    np.testing.assert_allclose(a.all_performances['MSE'][0], c.all_performances['MSE'][0], rtol=5e-3)
    np.testing.assert_allclose(a.all_tau[0], c.all_tau[0], rtol=5e-3)


def test_a_run_split_in_two_calls_is_the_same_chain():
    R, M, _, _, _ = generate_bnmtf(90, 70, 6, 4, 0.15, seed_data=5, seed_mask=6)
    outs = []
    for split in ((7,), (3, 4), (1, 1, 5)):
        np.random.seed(3)
        b = bnmtf_gibbs_optimised(R, M, 6, 4, PRI, seed=21, verbose=False)
        b.initialise('random', 'random')
        Fs, Ss, taus = [], [], []
        for n in split:
            b.run(n)
            Fs.append(b.all_F.copy()); Ss.append(b.all_S.copy()); taus.append(b.all_tau.copy())
        outs.append((np.concatenate(Fs), np.concatenate(Ss), np.concatenate(taus)))
    for F, S, tau in outs[1:]:
        assert np.array_equal(F, outs[0][0]) and np.array_equal(S, outs[0][1]) and np.array_equal(tau, outs[0][2])


def test_run_many_is_every_models_own_run():
    """A batch in one launch: every model ends with exactly the chain its own run() draws -- different shapes, ranks, masks and
    seeds in one grid (so: other block sizes and slot classes than alone), a model of the multi-launch path and a two-factor model
    among them."""
    from bnmtf_amd import bnmf_gibbs_optimised
    from bnmtf_amd.synthetic import generate_bnmf
    specs = [(100, 80, 5, 5, 0.1, 1), (60, 90, 4, 7, 0.3, 2), (37, 29, 5, 3, 0.0, 3), (150, 40, 12, 6, 0.2, 4), (300, 200, 8, 8, 0.1, 5),
             (622, 138, 10, 10, 0.19, 6), (1100, 64, 6, 5, 0.1, 7)]
    def build():
        ms = []
        for (I, J, K, L, miss, seed) in specs:
            R, M, _, _, _ = generate_bnmtf(I, J, K, L, miss, seed_data=seed, seed_mask=seed + 50)
            np.random.seed(seed)
            m = bnmtf_gibbs_optimised(R, M, K, L, PRI, seed=seed, verbose=False)
            m.initialise('random', 'random')
            m.set_small_path('always')          # (alone, the models that fill a CU would take the multi-launch path: another summation order)
            ms.append(m)
        R, M, _, _ = generate_bnmf(80, 60, 5, 0.1, seed_data=9, seed_mask=10)
        np.random.seed(9)
        m = bnmf_gibbs_optimised(R, M, 5, dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1), seed=9, verbose=False)
        m.initialise('random')
        ms.append(m)
        return ms
    solo = build()
    for m in solo:
        m.run(5)
    batch = build()
    assert [m.is_small() for m in batch] == [True] * 6 + [False, True]
    res = bnmtf_amd.run_many(batch, 5)
    assert len(res) == len(batch)
    for a, b in zip(solo[:-1], batch[:-1]):
        assert np.array_equal(a.all_F, b.all_F) and np.array_equal(a.all_S, b.all_S) and np.array_equal(a.all_G, b.all_G)
        assert np.array_equal(a.all_tau, b.all_tau) and a.all_performances == b.all_performances
        assert np.array_equal(a.F, b.F) and np.array_equal(a.S, b.S) and np.array_equal(a.G, b.G) and a.tau == b.tau
    assert np.array_equal(solo[-1].all_U, batch[-1].all_U)
    # and a second batched call continues every chain
    for m in solo:
        m.run(3)
    bnmtf_amd.run_many(batch, 3, store_samples=False)
    for a, b in zip(solo[:-1], batch[:-1]):
        assert np.array_equal(a.F, b.F) and np.array_equal(a.S, b.S) and np.array_equal(a.all_tau, b.all_tau)


def test_posterior_means_on_the_device_equal_the_host_means():
    R, M, _, _, _ = generate_bnmtf(80, 60, 5, 4, 0.2, seed_data=1, seed_mask=2)
    np.random.seed(1)
    b = bnmtf_gibbs_optimised(R, M, 5, 4, PRI, seed=3, verbose=False)
    b.initialise('random', 'random')
    b.run(30, expectation=(10, 2))
    dev = b.approx_expectation(10, 2)
    b._dev_expect = None
    host = b.approx_expectation(10, 2)
    for x, y in zip(dev, host):
        np.testing.assert_allclose(x, y, rtol=1e-6)


def test_switching_paths_mid_chain_keeps_the_state():
    R, M, _, _, _ = generate_bnmtf(70, 50, 4, 3, 0.1, seed_data=2, seed_mask=3)
    np.random.seed(2)
    b = bnmtf_gibbs_optimised(R, M, 4, 3, PRI, seed=5, verbose=False)
    b.initialise('random', 'random')
    b.run(5)
    F5, S5 = b.F.copy(), b.S.copy()
    t = b.tauS(1, 2)                               # a multi-launch entry point: builds that path, takes the state over
    assert np.isfinite(t) and np.array_equal(b.F, F5) and np.array_equal(b.S, S5)
    b.set_small_path(False)
    b.run(3)
    b.set_small_path('always')
    b.run(3)
    assert np.isfinite(b.all_performances['MSE']).all() and b.all_performances['MSE'][-1] < 2.0 * np.var(R)
    p = b.predict_while_running()
    assert abs(p["MSE"] - b.all_performances['MSE'][-1]) < 1e-4 * p["MSE"]


def test_batched_greedy_search_cross_validation_gives_the_unbatched_results(tmp_path):
    """The folds' walks side by side on a batched pool: the steps they have open together are one device call (a block per model),
    and every model is the chain its own run() draws -- the same held-out performances, to the bit."""
    from bnmtf_amd.cross_validation.greedy_search_cross_validation import GreedySearchCrossValidation
    from bnmtf_amd.cross_validation.replicas import ReplicaPool
    import random
    R, M, _, _, _ = generate_bnmtf(40, 30, 3, 2, 0.1, seed_data=3, seed_mask=4)
    perf = []
    for batched in (False, True):
        random.seed(0); np.random.seed(0)
        pool = ReplicaPool(devices=[0], shared={"R": R}, batched=batched)
        calls = []
        pmap = pool.map
        pool.map = lambda fn, jobs, *a, **k: (calls.append(len(list(jobs))), pmap(fn, jobs, *a, **k))[1]
        cv = GreedySearchCrossValidation(classifier=bnmtf_gibbs_optimised, R=R, M=M, values_K=[2, 3, 4], values_L=[2, 3], folds=3, priors=PRI,
                                         init_S="random", init_FG="kmeans", iterations=40, restarts=1, quality_metric="AIC",
                                         file_performance=str(tmp_path / ("perf%d.txt" % batched)), pool=pool, seed=7)
        cv.run(burn_in=20, thinning=2)
        pool.close()
        perf.append(cv.performances)
        if batched:
            assert calls[0] == 3 and max(calls) >= 6, calls        # the three folds' first models in one call, their steps of up to three models together
    assert perf[0] == perf[1]


def test_the_path_is_chosen_by_what_the_call_runs():
    """A tri-factorisation that fills a CU (1024-thread block) runs faster alone on the multi-launch path, and on a CU of its own as
    soon as a call runs two of them; the small ones always take the one launch.  (std_built: were the multi-launch structures built?)"""
    R, M, _, _, _ = generate_bnmtf(622, 138, 6, 6, 0.19, seed_data=1, seed_mask=2)
    def model(seed):
        np.random.seed(seed)
        b = bnmtf_gibbs_optimised(R, M, 6, 6, PRI, seed=seed, verbose=False)
        b.initialise('random', 'random')
        return b
    lone = model(1)
    assert not lone.is_small()
    lone.run(3)
    assert "std_built=1" in lone.describe()
    pair = [model(2), model(3)]
    bnmtf_amd.run_many(pair, 3)
    assert all("std_built=0" in m.describe() and "block=1024" in m.describe() for m in pair)
    assert all(np.isfinite(m.all_performances['MSE']).all() for m in pair + [lone])
    Rs, Ms, _, _, _ = generate_bnmtf(100, 80, 5, 5, 0.1, seed_data=1, seed_mask=2)
    np.random.seed(4)
    small = bnmtf_gibbs_optimised(Rs, Ms, 5, 5, PRI, seed=4, verbose=False)
    small.initialise('random', 'random')
    assert small.is_small()
    small.run(3)
    assert "std_built=0" in small.describe()
    # ranks above 10: the S step's sequential form (~2 us per entry of S) -- alone the multi-launch path's chain is several times faster,
    # a batch that is worth the K L = 256 entries takes the one launch
    wide = []
    for seed in range(5):
        np.random.seed(seed)
        w = bnmtf_gibbs_optimised(Rs, Ms, 16, 16, PRI, seed=seed, verbose=False)
        w.initialise('random', 'random')
        wide.append(w)
    assert not wide[0].is_small()
    wide[0].run(2)
    assert "std_built=1" in wide[0].describe()
    bnmtf_amd.run_many(wide[1:], 2)
    assert all("std_built=0" in w.describe() for w in wide[1:])


def test_random_shapes_ranks_and_masks_agree_with_the_multi_launch_path():
    """tools/fuzz_small_tri.py in short: edge shapes (one row, one column, rank one, nothing missing, 45 % missing), both forms of the
    S step, every block size -- first iteration against the multi-launch path, draws and mode updates."""
    rng = np.random.RandomState(5)
    done = 0
    for case in range(60):
        I = int(rng.choice([1, 2, 7, 33, 65, 100, 257, 400, 622])); J = int(rng.choice([1, 3, 16, 31, 80, 138, 300]))
        if I + J > 900: J = max(1, 900 - I)
        K = int(rng.choice([1, 2, 5, 10, 11, 32])); L = int(rng.choice([1, 3, 10, 12, 32]))
        miss = float(rng.choice([0.0, 0.05, 0.2, 0.45]))
        try:
            R, M, _, _, _ = generate_bnmtf(I, J, K, L, miss, seed_data=case, seed_mask=case + 1)
        except RuntimeError:                     # (no mask without an empty row or column at this shape and density)
            continue
        for upd in ("draw", "mode"):
            runs = []
            for small in (True, False):
                np.random.seed(case)
                b = bnmtf_gibbs_optimised(R, M, K, L, PRI, seed=case + 5, verbose=False)
                b.initialise('random', 'random')
                b.set_small_path('always' if small else False)
                if small and not b.is_small():
                    break
                b.run(2, update=upd)
                runs.append(b)
            if len(runs) < 2:
                break
            a, c = runs
            for nm in ("all_F", "all_S", "all_G"):
                x, y = getattr(a, nm)[0], getattr(c, nm)[0]
                d = np.abs(x - y) / (1e-3 + np.abs(y))
                assert np.isfinite(x).all() and np.mean(d < 5e-3) >= (0.9 if upd == "draw" else 0.999), (case, I, J, K, L, miss, upd, nm, np.mean(d < 5e-3), d.max())
            np.testing.assert_allclose(a.all_performances['MSE'][0], c.all_performances['MSE'][0], rtol=2e-2 if upd == "draw" else 1e-3)
            for m in runs:
                m.close()
        else:
            done += 1
        if done >= 16:
            break
    assert done >= 16
