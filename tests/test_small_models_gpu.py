"""The one-launch path for small models (csrc/kernel_small.hip: the whole run() of a BNMF Gibbs / ICM model in one launch, one
block per model; bnmtf_amd.run_many: a batch of models in one grid) against the oracle, the reference's golden vectors and
the multi-launch path.  Same tolerances as tests/test_bnmf_gibbs_gpu.py (fp32 device arithmetic vs the reference's fp64)."""
import numpy as np
import pytest

import bnmtf_amd
from bnmtf_amd import bnmf_gibbs_optimised, nmf_icm
from bnmtf_amd.synthetic import generate_bnmf
from oracle import bnmtf_oracle as O

pytestmark = pytest.mark.gpu

PRI = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)


def _pri(c):
    return dict(alpha=float(c["alpha"]), beta=float(c["beta"]), lambdaU=c["lambdaU"], lambdaV=c["lambdaV"])


@pytest.mark.parametrize("name", ["t5x3", "toy", "r37x29", "r40x33"])
def test_mode_update_trajectory_matches_oracle_on_the_small_path(golden, name):
    """Deterministic end-to-end parity of the one-launch kernel: the block's own MFMA contraction with the K x K term, the column
    loop with its running numerator bases, q handed over between the half sweeps, fp64 MFMA Gram, Gram-identity SSE, tau."""
    c = golden("bnmf_gibbs_cond.npz").case(name)
    o = O.BNMFGibbsOracle(c["R"], c["M"], int(c["K"]), _pri(c))
    o.U, o.V, o.tau = c["U"].copy(), c["V"].copy(), float(c["tau"])
    o.run(8, draw=False)
    b = bnmf_gibbs_optimised(c["R"], c["M"], int(c["K"]), _pri(c), verbose=False)
    b.U, b.V, b.tau = c["U"].copy(), c["V"].copy(), float(c["tau"])
    assert b.is_small()
    b.run(8, update='mode')
    assert "std_built=0" in b.describe()            # nothing of the multi-launch path was needed
    np.testing.assert_allclose(b.all_performances['MSE'], o.all_performances['MSE'], rtol=2e-4)
    np.testing.assert_allclose(b.all_performances['R^2'], o.all_performances['R^2'], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(b.all_performances['Rp'], o.all_performances['Rp'], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(b.all_tau, o.all_tau, rtol=2e-4)
    assert np.abs(b.all_U[0] - o.all_U[0]).max() < 2e-4 * max(1.0, np.abs(o.all_U[0]).max())
    assert np.abs(b.all_V[0] - o.all_V[0]).max() < 2e-4 * max(1.0, np.abs(o.all_V[0]).max())
    assert np.abs(b.all_U[7] - o.all_U[7]).max() < 5e-3 * max(1.0, np.abs(o.all_U[7]).max())
    assert np.allclose(b.U, b.all_U[-1]) and np.allclose(b.V, b.all_V[-1]) and abs(b.tau - b.all_tau[-1]) < 1e-12
    assert len(b.all_times) == 8 and all(np.diff(b.all_times) > 0) and b.all_times[0] > 0


def test_draws_follow_the_oracle_with_the_same_philox_stream(golden):
    """Same seed, same counters: the small path draws the oracle's chain (first sweep element-wise but for accept / reject decisions on a
    rounding boundary), and it is the chain of the multi-launch path."""
    t = golden("toy_data.npz").case("bnmf")
    g = golden("gibbs_trajectories.npz").case("bnmf")
    I, J = t["R"].shape; K = 10
    pri = dict(alpha=1., beta=1., lambdaU=0.1 * np.ones((I, K)), lambdaV=0.1 * np.ones((J, K)))
    o = O.BNMFGibbsOracle(t["R"], t["M"], K, pri, seed=77)
    o.U, o.V, o.tau = g["U0_seed0"].copy(), g["V0_seed0"].copy(), float(g["tau0_seed0"])
    o.run(30)
    runs = []
    for small in (True, False):
        b = bnmf_gibbs_optimised(t["R"], t["M"], K, pri, verbose=False, seed=77)
        b.U, b.V, b.tau = g["U0_seed0"].copy(), g["V0_seed0"].copy(), float(g["tau0_seed0"])
        b.set_small_path(small)
        assert b.is_small() == small
        b.run(30)
        runs.append(b)
    b, b2 = runs
    d0 = np.abs(b.all_U[0] - o.all_U[0]) / (1e-3 + np.abs(o.all_U[0]))
    assert np.mean(d0 < 1e-3) > 0.99
    assert abs(b.all_tau[0] - o.all_tau[0]) < 1e-3 * o.all_tau[0]
    np.testing.assert_allclose(b.all_performances['MSE'][:3], o.all_performances['MSE'][:3], rtol=1e-3)
    assert abs(np.mean(b.all_performances['MSE'][20:]) / np.mean(o.all_performances['MSE'][20:]) - 1) < 0.1
    d1 = np.abs(b.all_U[0] - b2.all_U[0]) / (1e-3 + np.abs(b2.all_U[0]))
    assert np.mean(d1 < 1e-3) > 0.99
    np.testing.assert_allclose(b.all_performances['MSE'][:3], b2.all_performances['MSE'][:3], rtol=1e-3)


def test_toy_trajectory_within_reference_bands(golden):
    """Config 1 (toy 100x80, K=10) on the small path against 10 seeded runs of the reference, 200 iterations (three refreshes of q)."""
    t = golden("toy_data.npz").case("bnmf")
    g = golden("gibbs_trajectories.npz").case("bnmf")
    I, J = t["R"].shape; K = 10
    pri = dict(alpha=1., beta=1., lambdaU=0.1 * np.ones((I, K)), lambdaV=0.1 * np.ones((J, K)))
    b = bnmf_gibbs_optimised(t["R"], t["M"], K, pri, verbose=False, seed=11)
    b.U, b.V, b.tau = g["U0_seed0"].copy(), g["V0_seed0"].copy(), float(g["tau0_seed0"])
    assert b.is_small()
    b.run(200)
    mse = np.array(b.all_performances['MSE']); ref = g["mse"]
    lo, hi = ref.min(axis=0), ref.max(axis=0)
    assert (mse[:60] > lo[:60] / 2.5).all() and (mse[:60] < hi[:60] * 2.5).all()
    m_ref = ref[:, 150:].mean(axis=1)
    assert m_ref.min() * 0.97 < mse[150:].mean() < m_ref.max() * 1.03
    assert abs(np.mean(b.all_tau[150:]) - g["tau"][:, 150:].mean()) < 0.05
    # the metrics the kernel reports (Gram identities, q handed over 199 times) == direct fp64 evaluation of the final sample
    p = b.predict_while_running()
    assert abs(p["MSE"] - mse[-1]) < 2e-5 * mse[-1]
    assert abs(p["R^2"] - b.all_performances['R^2'][-1]) < 1e-5 and abs(p["Rp"] - b.all_performances['Rp'][-1]) < 1e-5
    eU, eV, _ = b.approx_expectation(100, 2)
    held = ((1 - t["M"]) * (t["R_true"] - eU @ eV.T) ** 2).sum() / (1 - t["M"]).sum()
    ref_held = g["heldout_mse_vs_Rtrue"]
    assert ref_held.min() * 0.8 < held < ref_held.max() * 1.2


def test_a_run_split_in_two_calls_is_the_same_chain():
    """run(3); run(4) == run(7) bit for bit: q of the missing entries is handed from call to call, the refresh goes by the iteration number."""
    R, M, _, _ = generate_bnmf(90, 70, 6, 0.15, seed_data=5, seed_mask=6)
    outs = []
    for split in ((7,), (3, 4), (1, 1, 5)):
        np.random.seed(3)
        b = bnmf_gibbs_optimised(R, M, 6, PRI, seed=21, verbose=False)
        b.initialise('random')
        Us, taus = [], []
        for n in split:
            b.run(n)
            Us.append(b.all_U.copy()); taus.append(b.all_tau.copy())
        outs.append((np.concatenate(Us), np.concatenate(taus)))
    for U, tau in outs[1:]:
        assert np.array_equal(U, outs[0][0]) and np.array_equal(tau, outs[0][1])


def test_run_many_is_every_models_own_run():
    """A batch in one launch: every model ends with exactly the chain its own run() draws -- different shapes, ranks, masks and
    seeds in one grid, a model of the multi-launch path among them."""
    specs = [(100, 80, 10, 0.1, 1), (60, 90, 4, 0.3, 2), (37, 29, 5, 0.0, 3), (150, 40, 12, 0.2, 4), (300, 200, 8, 0.1, 5), (622, 138, 25, 0.19, 6),
             (1100, 64, 6, 0.1, 7)]
    def build():
        ms = []
        for (I, J, K, miss, seed) in specs:
            R, M, _, _ = generate_bnmf(I, J, K, miss, seed_data=seed, seed_mask=seed + 50)
            np.random.seed(seed)
            m = bnmf_gibbs_optimised(R, M, K, PRI, seed=seed, verbose=False)
            m.initialise('random')
            m.set_small_path('always')          # (alone, the models that fill a CU would take the multi-launch path: another summation order)
            ms.append(m)
        return ms
    solo = build()
    for m in solo:
        m.run(6)
    batch = build()
    assert [m.is_small() for m in batch] == [True] * 6 + [False]
    res = bnmtf_amd.run_many(batch, 6)
    assert len(res) == len(batch)
    for a, b in zip(solo, batch):
        assert np.array_equal(a.all_U, b.all_U) and np.array_equal(a.all_V, b.all_V) and np.array_equal(a.all_tau, b.all_tau)
        assert a.all_performances == b.all_performances
        assert np.array_equal(a.U, b.U) and a.tau == b.tau
    # and a second batched call continues every chain
    for m in solo:
        m.run(3)
    bnmtf_amd.run_many(batch, 3, store_samples=False)
    for a, b in zip(solo, batch):
        assert np.array_equal(a.U, b.U) and np.array_equal(a.all_tau, b.all_tau)


def test_posterior_means_on_the_device_equal_the_host_means():
    R, M, _, _ = generate_bnmf(80, 60, 5, 0.1, seed_data=2, seed_mask=3)
    np.random.seed(0)
    a = bnmf_gibbs_optimised(R, M, 5, PRI, seed=9, verbose=False); a.initialise('random')
    np.random.seed(0)
    b = bnmf_gibbs_optimised(R, M, 5, PRI, seed=9, verbose=False); b.initialise('random')
    a.run(40)
    b.run(40, store_samples=False, expectation=(10, 3))
    assert a.is_small() and b.is_small()
    eU, eV, et = a.approx_expectation(10, 3)
    dU, dV, dt = b.approx_expectation(10, 3)
    np.testing.assert_allclose(dU, eU, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(dV, eV, rtol=1e-6, atol=1e-7)
    assert abs(dt - et) < 1e-12 * et
    assert abs(a.quality('AIC', 10, 3) - b.quality('AIC', 10, 3)) < 1e-6 * abs(a.quality('AIC', 10, 3))


def test_switching_paths_mid_chain_keeps_the_state():
    """small -> multi-launch -> small: the state follows (device-to-device), conditional parameters come from the multi-launch kernels."""
    R, M, _, _ = generate_bnmf(70, 50, 4, 0.2, seed_data=8, seed_mask=9)
    np.random.seed(4)
    b = bnmf_gibbs_optimised(R, M, 4, PRI, seed=2, verbose=False); b.initialise('random')
    b.run(5)
    U5, V5 = b.U.copy(), b.V.copy()
    o = O.BNMFGibbsOracle(R.astype(np.float64), M, 4, PRI)
    o.U, o.V, o.tau = U5.copy(), V5.copy(), b.tau
    np.testing.assert_allclose(b.tauU(1), o.tauU(1), rtol=5e-6)           # builds the multi-launch structures, hands the state over
    assert "std_built=1" in b.describe()
    b.set_small_path(False); b.run(2)
    b.set_small_path(True); b.run(2)
    assert np.isfinite(b.all_performances['MSE']).all() and b.all_performances['MSE'][-1] < 5.0
    p = b.predict_while_running()
    assert abs(p["MSE"] - b.all_performances['MSE'][-1]) < 5e-5 * p["MSE"]


def test_edge_shapes_on_the_small_path():
    """Nothing missing, rank one, a single row, a single column, most entries missing: mode-update trajectories against the oracle."""
    rs = np.random.RandomState(0)
    cases = []
    R = rs.rand(40, 30) * 3 + 0.5
    cases.append((R, np.ones((40, 30)), 3))                                   # nothing missing
    M = (rs.rand(40, 30) < 0.8).astype(float); M[:, 0] = 1; M[0, :] = 1
    cases.append((R, M, 1))                                                   # rank one
    cases.append((R[:1, :], np.ones((1, 30)), 2))                             # one row
    cases.append((R[:, :1], np.ones((40, 1)), 2))                             # one column
    M = (rs.rand(40, 30) < 0.15).astype(float); M[np.arange(40), np.arange(40) % 30] = 1; M[np.arange(30) % 40, np.arange(30)] = 1
    cases.append((R, M, 3))                                                   # ~85 % missing
    for R, M, K in cases:
        pri = dict(alpha=1., beta=1., lambdaU=0.5, lambdaV=0.5)
        b = bnmf_gibbs_optimised(R, M, K, pri, verbose=False, seed=1)
        assert b.is_small()
        U0 = rs.rand(R.shape[0], K) + 0.5; V0 = rs.rand(R.shape[1], K) + 0.5
        b.U, b.V, b.tau = U0.copy(), V0.copy(), 1.0
        o = O.BNMFGibbsOracle(R, M, K, pri)
        o.U, o.V, o.tau = U0.copy(), V0.copy(), 1.0
        o.run(5, draw=False); b.run(5, update='mode')
        scale = (M * R ** 2).sum() / M.sum()
        assert np.abs(np.array(b.all_performances['MSE']) - np.array(o.all_performances['MSE'])).max() < 2e-5 * scale
        np.testing.assert_allclose(b.all_tau, o.all_tau, rtol=5e-4)
        assert np.abs(b.all_U[-1] - o.all_U[-1]).max() < 2e-3 * max(1.0, np.abs(o.all_U[-1]).max())
        b.run(5)
        assert np.isfinite(b.all_U).all() and (b.all_U >= 0).all() and (b.all_tau > 0).all()


def test_icm_takes_the_small_path(golden):
    """nmf_icm (update rule ICM: minimum_TN clamp, tau = gamma mode) runs through the one-launch kernel; its trajectories are
    compared with the reference's in tests/test_icm_gpu.py."""
    t = golden("toy_data.npz").case("bnmf")
    m = nmf_icm(t["R"], t["M"], 5, dict(alpha=1.0, beta=1.0, lambdaU=0.1, lambdaV=0.1), verbose=False)
    m.initialise("random")
    assert m.is_small()
    m.run(10, minimum_TN=0.1)
    assert "std_built=0" in m.describe()
    assert m.U.min() >= 0.1 * (1 - 1e-6) and m.all_performances["MSE"][-1] < m.all_performances["MSE"][0]


def test_batched_replica_pool_gives_the_sequential_results():
    """ReplicaPool(batched=True): the jobs of a model search fitted as one device call (bnmtf_amd.run_many) == one after the other."""
    from bnmtf_amd.cross_validation.replicas import ReplicaPool, fit_model
    R, M, _, _ = generate_bnmf(60, 50, 4, 0.2, seed_data=1, seed_mask=2)
    rs = np.random.RandomState(3)
    test = ((rs.rand(60, 50) < 0.5) & (M == 0)).astype(float)
    jobs = [dict(classifier=bnmf_gibbs_optimised, args=(K, PRI), init={"init": "random"}, iterations=30, burn_in=10, thinning=2, minimum_TN=None,
                 M=M, test=test, metrics=["loglikelihood", "AIC", "MSE"], seed=100 + K) for K in (2, 3, 4, 5, 6)]
    seq = ReplicaPool(devices=[0], shared={"R": R}).map(fit_model, jobs)
    bat = ReplicaPool(devices=[0], shared={"R": R}, batched=True).map(fit_model, jobs)
    for a, b in zip(seq, bat):
        for k in a["quality"]:
            assert a["quality"][k] == pytest.approx(b["quality"][k], rel=1e-12)       # (the metric kernel sums with fp64 atomics)
        for k in a["performance"]:
            assert a["performance"][k] == pytest.approx(b["performance"][k], rel=1e-12)


def test_batched_replica_pool_fits_icm_jobs_by_icm():
    """nmf_icm inherits the Gibbs class: a batched pool used to send its jobs (minimum_TN=None, the search drivers' default) through
    run_many, i.e. fit them by Gibbs draws (round 4's advice).  They are run one by one now: batched == unbatched, and the batch
    entry point refuses the class."""
    from bnmtf_amd import nmf_icm
    from bnmtf_amd.cross_validation.replicas import ReplicaPool, fit_model
    R, M, _, _ = generate_bnmf(60, 50, 4, 0.2, seed_data=1, seed_mask=2)
    jobs = [dict(classifier=nmf_icm, args=(K, PRI), init={"init": "random"}, iterations=12, burn_in=None, thinning=None, minimum_TN=None,
                 M=M, test=None, metrics=["loglikelihood", "MSE"], seed=100 + K) for K in (2, 3, 4)]
    seq = ReplicaPool(devices=[0], shared={"R": R}).map(fit_model, jobs)
    bat = ReplicaPool(devices=[0], shared={"R": R}, batched=True).map(fit_model, jobs)
    for a, b in zip(seq, bat):
        for k in a["quality"]:
            assert a["quality"][k] == pytest.approx(b["quality"][k], rel=1e-12)
    m = nmf_icm(R, M, 3, PRI)
    m.initialise('random')
    with pytest.raises(TypeError):
        bnmtf_amd.run_many([m], 3)
    g = bnmf_gibbs_optimised(R, M, 3, PRI, verbose=False, seed=5)
    g.initialise('random')
    U0 = g.U.copy()
    bnmtf_amd.run_many([g], 0)                  # (used to install the batch's zero-filled placeholders as the model's state)
    assert np.array_equal(g.U, U0)


@pytest.mark.parametrize("miss", [0.27, 0.34, 0.45])
def test_the_slot_classes_above_32_follow_the_oracle(miss):
    """Masks with more missing entries than 1024 threads hold at 32 slots each (a training fold of a 19 %-missing 622 x 138 matrix has
    27 %): the 40-, 48- and 64-slot classes, which gather the previous column's values again instead of keeping them in registers."""
    import re
    R, M, _, _ = generate_bnmf(622, 138, 6, miss, seed_data=4, seed_mask=5)
    rs = np.random.RandomState(2)
    U0 = rs.rand(622, 6) + 0.3; V0 = rs.rand(138, 6) + 0.3
    b = bnmf_gibbs_optimised(R, M, 6, PRI, verbose=False, seed=3)
    b.set_small_path('always')
    assert b.is_small()
    b.U, b.V, b.tau = U0.copy(), V0.copy(), 1.0
    slots = [int(x) for x in re.search(r"slots=(\d+)/(\d+)", b.describe()).groups()]
    assert max(slots) == {0.27: 40, 0.34: 48, 0.45: 64}[miss], b.describe()
    o = O.BNMFGibbsOracle(R.astype(np.float64), M, 6, PRI)
    o.U, o.V, o.tau = U0.copy(), V0.copy(), 1.0
    o.run(4, draw=False); b.run(4, update='mode')
    np.testing.assert_allclose(b.all_performances['MSE'], o.all_performances['MSE'], rtol=2e-4)
    np.testing.assert_allclose(b.all_tau, o.all_tau, rtol=2e-4)
    assert np.abs(b.all_U[-1] - o.all_U[-1]).max() < 2e-3 * max(1.0, np.abs(o.all_U[-1]).max())
    b.run(10)
    p = b.predict_while_running()
    assert abs(p["MSE"] - b.all_performances['MSE'][-1]) < 5e-5 * p["MSE"]


def test_batched_line_search_cross_validation_takes_the_final_models_along(tmp_path):
    """LineSearchCrossValidation on a batched pool with an explicit seed: the folds' final models ride in the search's batch (for
    every candidate K; the losing ones are dropped) -- one device call instead of two, the same folds, ranks and scores."""
    import random
    import re
    from bnmtf_amd.cross_validation import LineSearchCrossValidation, ReplicaPool
    R, M, _, _ = generate_bnmf(50, 40, 3, 0.1, seed_data=7, seed_mask=8)
    out = []
    for batched in (False, True):
        random.seed(5); np.random.seed(5)
        f = str(tmp_path / ("ls%d.txt" % batched))
        pool = ReplicaPool(devices=[0], shared={"R": np.asarray(R, dtype=float)}, batched=batched)
        cv = LineSearchCrossValidation(classifier=bnmf_gibbs_optimised, R=R, M=M, values_K=[2, 3, 5], folds=3, priors=PRI, init_UV="random",
                                       iterations=60, restarts=2, quality_metric="AIC", file_performance=f, pool=pool, seed=9)
        cv.run(burn_in=30, thinning=2)
        pool.close()
        txt = open(f).read()
        out.append((re.findall(r"Best K for fold \d+: (\d+)\.", txt), cv.performances))
    assert out[0][0] == out[1][0] and len(out[0][0]) == 3
    for m in ("MSE", "R^2", "Rp"):
        np.testing.assert_allclose(out[0][1][m], out[1][1][m], rtol=1e-10)


def test_the_path_is_chosen_by_what_the_call_runs():
    """'auto' (the default): a model of a 256- / 512-thread block takes the one-launch path alone; a model that fills a CU (622 x 138
    at 19 % missing: a 1024-thread block) takes the multi-launch path alone -- the whole chip is faster for one such model -- and the
    one-launch path from three models per run_many call on."""
    R, M, _, _ = generate_bnmf(100, 80, 10, 0.1, seed_data=1, seed_mask=2)
    toy = bnmf_gibbs_optimised(R, M, 10, PRI, verbose=False, seed=1)
    toy.initialise('random')
    assert toy.is_small() and "block=256" in toy.describe()
    R, M, _, _ = generate_bnmf(622, 138, 25, 0.19, seed_data=3, seed_mask=4)
    ms = []
    for s in range(3):
        np.random.seed(s)
        m = bnmf_gibbs_optimised(R, M, 25, PRI, verbose=False, seed=s)
        m.initialise('random')
        assert "block=1024" in m.describe() and not m.is_small()
        ms.append(m)
    ms[0].run(2)
    assert "std_built=1" in ms[0].describe()                     # alone: the multi-launch path (its structures were built for it)
    bnmtf_amd.run_many(ms[1:], 2)
    assert all("std_built=1" in m.describe() for m in ms[1:])    # two models: one after the other on the multi-launch path
    ms2 = []
    for s in range(3):
        np.random.seed(s)
        m = bnmf_gibbs_optimised(R, M, 25, PRI, verbose=False, seed=s)
        m.initialise('random')
        ms2.append(m)
    bnmtf_amd.run_many(ms2, 2)
    assert all("std_built=0" in m.describe() for m in ms2)       # three: one grid, a block each
    # the same chain either way (first sweep element-wise but for flipped accept / reject decisions)
    d = np.abs(ms2[0].all_U[0] - ms[0].all_U[0]) / (1e-3 + np.abs(ms[0].all_U[0]))
    assert np.mean(d < 1e-3) > 0.99
