"""bnmtf_amd.cross_validation without a GPU: the vectorised mask helpers against masks the reference drew
(tests/golden/masks.npz), and the replica scheduling / selection logic of the search and cross-validation drivers with a
stand-in classifier (tests/cv_fakes.py)."""
import os
import random
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
os.environ["PYTHONPATH"] = os.pathsep.join([HERE, os.path.dirname(HERE), os.environ.get("PYTHONPATH", "")])   # for spawned workers

from bnmtf_amd.cross_validation import (GreedySearch, GridSearch, LineSearch, LineSearchCrossValidation, MatrixCrossValidation,
                                         ParallelMatrixCrossValidation, ReplicaPool, mask)
from bnmtf_amd.cross_validation.replicas import fit_model
from cv_fakes import FakeModel, FakeTri


def test_masks_and_folds_are_the_reference_s_under_random_seed(golden):
    g = golden("masks.npz")._z
    M = g["M"]
    random.seed(1); assert np.array_equal(mask.generate_M(7, 5, 0.3), g["generate_M"])
    random.seed(2); folds = mask.compute_folds(9, 6, 4, M); assert np.array_equal(np.array(folds), g["folds"])
    assert np.array_equal(np.array(mask.compute_Ms(folds)), g["Ms"])
    random.seed(3); assert np.array_equal(np.array(mask.compute_folds_attempts(9, 6, 3, 50, M)), g["folds_attempts"])
    random.seed(4); tr, te = mask.generate_M_from_M(M, 0.3); assert np.array_equal(tr, g["split_train"]) and np.array_equal(te, g["split_test"])
    random.seed(5); tr, te = mask.try_generate_M_from_M(M, 0.4, 20); assert np.array_equal(tr, g["try_train"]) and np.array_equal(te, g["try_test"])
    random.seed(6); rows = mask.compute_crossval_folds_rows_attempts(M, 5, 3, 50)
    assert np.array_equal(np.array([a for a, b in rows]), g["rows_train"]) and np.array_equal(np.array([b for a, b in rows]), g["rows_test"])
    random.seed(7); cols = mask.compute_crossval_folds_columns_attempts(M, 4, 2, 50)
    assert np.array_equal(np.array([a for a, b in cols]), g["cols_train"]) and np.array_equal(np.array([b for a, b in cols]), g["cols_test"])
    assert np.array_equal(mask.calc_inverse_M(M), g["inverse"]) and np.array_equal(np.array(mask.nonzero_indices(M)), g["nz"])
    assert mask.check_empty_rows_columns(M) and not mask.check_empty_rows_columns(np.zeros((3, 3)))
    assert mask.nonzero_row_indices(M)[0] == [0, 2, 3, 4, 5] and mask.nonzero_column_indices(M)[0] == [0, 1, 2, 3, 5, 6, 7, 8]
    assert len(mask.recover_predictions(M, np.ones((9, 6)), np.zeros((9, 6)))) == 4


def test_replica_pool_deals_jobs_to_worker_processes_and_reports_failures():
    R = np.ones((6, 5)); M = np.ones((6, 5))
    jobs = [dict(classifier=FakeModel, args=(K, {}), init={"init": "random"}, iterations=5, burn_in=2, thinning=1, minimum_TN=None,
                 M=M, test=M, metrics=["loglikelihood", "MSE"], seed=K) for K in range(1, 9)]
    with ReplicaPool(devices=[0, 1, 0], shared={"R": R}) as pool:
        res = pool.map(fit_model, jobs)
        assert [r["quality"]["MSE"] for r in res] == [FakeModel(R, M, K, {}, seed=K).quality("MSE") for K in range(1, 9)]   # job order kept
        assert {r["performance"]["device"] for r in res} <= {0, 1} and len({r["performance"]["pid"] for r in res}) >= 2
        assert all(r["performance"]["expectation_burn_in"] == 2 for r in res)          # sampled models accumulate on the device
        assert os.getpid() not in {r["performance"]["pid"] for r in res}
        with pytest.raises(RuntimeError) as e:
            pool.map(fit_model, jobs + [dict(jobs[0], args=(13, {}))])
        assert "unlucky K" in str(e.value) and "1 of 9" in str(e.value)
    serial = ReplicaPool(devices=[0], shared={"R": R}).map(fit_model, jobs[:2])      # one slot: in this process
    assert serial[0]["performance"]["pid"] == os.getpid()


def test_line_grid_and_greedy_search_select_like_the_reference():
    R = np.ones((6, 5)); M = np.ones((6, 5))
    ls = LineSearch(FakeModel, [2, 3, 4, 5, 6], R, M, {"alpha": 1}, "random", iterations=10, restarts=3, seed=0)
    ls.search(burn_in=3, thinning=2)
    assert ls.best_value("BIC") == 4 and ls.best_value("MSE") == 4 and len(ls.all_values("AIC")) == 5
    # of the three restarts (seeds s, s+1, s+2) the one with the largest log-likelihood (seed % 3 == 2) is kept
    assert ls.all_values("loglikelihood")[2] == 2.0
    with pytest.raises(AssertionError) as e:
        ls.all_values("FAIL")
    assert str(e.value) == "Unrecognised metric name: FAIL."
    pri = {"alpha": 1, "lambdaF": 0.1, "lambdaS": 0.2, "lambdaG": 0.3}
    gs = GridSearch(FakeTri, [1, 2, 3, 4], [4, 5, 6], R, M, pri, "random", "random", iterations=5)
    gs.search()
    assert gs.best_value("BIC") == (3, 5) and gs.all_values("MSE").shape == (4, 3) and gs.all_values("MSE")[0, 0] == 5
    gr = GreedySearch(FakeTri, [1, 2, 3, 4, 5], [3, 4, 5, 6, 7], R, M, pri, "random", "random", iterations=5)
    gr.search("AIC")
    assert gr.best_value("AIC") == (3, 5)
    tried = {(K, L) for K, L, _ in gr.all_values("AIC")}
    assert (1, 3) in tried and (5, 7) not in tried and len(tried) < 25          # a walk, not the whole grid


def test_line_search_cross_validation_and_matrix_cross_validation(tmp_path):
    rs = np.random.RandomState(0)
    R = rs.rand(12, 10); M = np.ones((12, 10)); M[0, 0] = M[5, 5] = 0
    random.seed(0)
    f = str(tmp_path / "perf.txt")
    with ReplicaPool(devices=[0, 0], shared={"R": R}) as pool:
        cv = LineSearchCrossValidation(FakeModel, R, M, [3, 4, 5], folds=3, priors={}, init_UV="random", iterations=8, restarts=2,
                                       quality_metric="AIC", file_performance=f, pool=pool, seed=1)
        cv.run(burn_in=2, thinning=2)
    txt = open(f).read()
    assert txt.count("Best K for fold") == 3 and "Best K for fold 1: 4." in txt and "Average performance:" in txt
    assert abs(cv.average_performance["MSE"] - 0.504) < 1e-12 and len(cv.performances["Rp"]) == 3
    # parameter search over K with the serial and the parallel driver: same folds (same random seed), same numbers
    out = []
    for cls, extra in ((MatrixCrossValidation, {}), (ParallelMatrixCrossValidation, {"P": 2, "devices": [0, 0]})):
        random.seed(5)
        f2 = str(tmp_path / (cls.__name__ + ".txt"))
        c = cls(FakeModel, R, M, 4, [{"K": 2, "priors": {}}, {"K": 6, "priors": {}}], {"iterations": 3}, f2, **extra)
        c.run()
        best = c.find_best_parameters("MSE", low_better=True)
        assert best[0] == {"K": 2, "priors": {}} and abs(best[1] - 0.502) < 1e-12
        out.append((c.performances["MSE"], c.all_performances[c.JSON({"K": 6, "priors": {}})]["n_test"]))
        assert "Best performances" in open(f2).read()
    assert out[0] == out[1]


def test_greedy_search_keeps_the_references_stale_variable_behind_a_switch():
    """greedy_search_bnmtf.py:165: along the K edge a successful step sets performance_so_far to the main loop's LAST
    performance_new_L, not to the step's own value.  as_written=True (default) reproduces that -- including the NameError
    the reference raises when the main loop never ran -- as_written=False is the symmetric rule."""
    R = np.ones((6, 5)); M = np.ones((6, 5))
    pri = {"alpha": 1, "lambdaF": 0.1, "lambdaS": 0.2, "lambdaG": 0.3}

    class Ridge(FakeTri):                  # quality falls along K while L is already at its edge value
        def quality(self, metric, burn_in=None, thinning=None):
            v = {1: 10.0, 2: 8.0, 3: 9.5, 4: 9.4, 5: 1.0}[self.K] + 100.0 * (self.L != 7)
            return {"loglikelihood": -v, "BIC": v, "AIC": v, "MSE": v, "ELBO": 0.0}[metric]
    # a single L: the main loop never runs, the K-edge loop takes its first successful step -> NameError as in the reference
    gw = GreedySearch(Ridge, [1, 2, 3, 4, 5], [7], R, M, pri, "random", "random", iterations=2)
    with pytest.raises(NameError) as e:
        gw.search("AIC")
    assert str(e.value) == "name 'performance_new_L' is not defined"
    gf = GreedySearch(Ridge, [1, 2, 3, 4, 5], [7], R, M, pri, "random", "random", iterations=2, as_written=False)
    gf.search("AIC")
    assert [K for K, L, _ in gf.all_values("AIC")] == [1, 2, 3] and gf.best_value("AIC") == (2, 7)     # stops when K = 3 is worse than K = 2
    # two values of L: the main loop runs once and ends with L at its edge; its performance_new_L (the value at (1, 7)) is
    # what the K-edge loop then compares with
    class Ridge2(Ridge):
        def quality(self, metric, burn_in=None, thinning=None):
            v = {1: 10.0, 2: 8.0, 3: 9.5, 4: 9.4, 5: 1.0}[self.K] + (50.0 if self.L == 6 else 0.0)
            return {"loglikelihood": -v, "BIC": v, "AIC": v, "MSE": v, "ELBO": 0.0}[metric]
    ga = GreedySearch(Ridge2, [1, 2, 3, 4, 5], [6, 7], R, M, pri, "random", "random", iterations=2)
    ga.search("AIC")
    gb = GreedySearch(Ridge2, [1, 2, 3, 4, 5], [6, 7], R, M, pri, "random", "random", iterations=2, as_written=False)
    gb.search("AIC")
    tried_a = sorted({(K, L) for K, L, _ in ga.all_values("AIC")}); tried_b = sorted({(K, L) for K, L, _ in gb.all_values("AIC")})
    # main loop: (1,6) -> best of (2,6), (1,7), (2,7) is (2,7) = 8: both indices advance, L is at its edge; then K-edge:
    # K = 3 (9.5) is worse than 8 -> both stop.  Same walk here; the switch matters from the second K-edge step on (above).
    assert tried_a == tried_b and ga.best_value("AIC") == (2, 7)


def test_greedy_search_cross_validation(tmp_path):
    from bnmtf_amd.cross_validation import GreedySearchCrossValidation
    rs = np.random.RandomState(1)
    R = rs.rand(12, 10); M = np.ones((12, 10)); M[0, 0] = M[5, 5] = 0
    random.seed(3)
    f = str(tmp_path / "greedy.txt")
    pri = {"alpha": 1, "lambdaF": 0.1, "lambdaS": 0.2, "lambdaG": 0.3}
    with ReplicaPool(devices=[0, 0], shared={"R": R}) as pool:
        cv = GreedySearchCrossValidation(FakeTri, R, M, [1, 2, 3, 4], [4, 5, 6], folds=3, priors=pri, init_S="random", init_FG="random",
                                         iterations=4, restarts=2, quality_metric="AIC", file_performance=f, pool=pool, seed=2)
        cv.run(burn_in=1, thinning=1)
    txt = open(f).read()
    assert txt.count("Best K,L for fold") == 3 and "Best K,L for fold 2: (3, 5)." in txt and txt.count("Performance: ") == 3
    assert "Average performance:" in txt and len(cv.performances["MSE"]) == 3 and abs(cv.average_performance["MSE"] - 0.503) < 1e-12
    with pytest.raises(AssertionError):
        GreedySearchCrossValidation(FakeTri, R, M, [1], [1], 2, pri, "random", "random", 1, 1, "loglikelihood", f)


def test_matrix_cross_validation_logs_a_failing_setting_and_carries_on(tmp_path):
    """matrix_cross_validation.py:79-81: a setting that raises is logged and skipped; the others are recorded."""
    rs = np.random.RandomState(0)
    R = rs.rand(12, 10); M = np.ones((12, 10))
    random.seed(5)
    f = str(tmp_path / "mcv.txt")
    c = MatrixCrossValidation(FakeModel, R, M, 3, [{"K": 2, "priors": {}}, {"K": 13, "priors": {}}], {"iterations": 3}, f)
    c.run()
    txt = open(f).read()
    assert "Tried parameters {'K': 13, 'priors': {}} but got exception: unlucky K. \n" in txt
    # (as in the reference, find_best_parameters indexes parameter_search by the position among the RECORDED settings, :125-127:
    # a failed setting ahead of the best one would shift it -- here the failed one is last)
    assert list(c.all_performances) == [c.JSON({"K": 2, "priors": {}})] and c.find_best_parameters("MSE", True)[0] == {"K": 2, "priors": {}}


def test_nested_matrix_cross_validation(tmp_path):
    """nested_matrix_cross_validation.py:78-140: per outer fold an inner cross-validation on the outer training mask names the parameters
    (one log file each), the outer model is fitted with them and scored on the outer test fold; the closing log lines."""
    from bnmtf_amd.cross_validation import MatrixNestedCrossValidation
    rs = np.random.RandomState(0)
    R = rs.rand(14, 12); M = np.ones((14, 12)); M[1, 1] = M[7, 3] = 0
    random.seed(11)
    files = [str(tmp_path / ("inner%d.txt" % i)) for i in range(3)]
    f = str(tmp_path / "nested.txt")
    search = [{"K": 2, "priors": {}}, {"K": 4, "priors": {}}, {"K": 6, "priors": {}}]
    n = MatrixNestedCrossValidation(FakeModel, R, M, 3, 2, search, {"iterations": 2}, f, files, devices=[0, 0])
    n.run()
    txt = open(f).read()
    assert txt.startswith("Average performances: ") and "\nAll performances: " in txt
    # FakeModel.predict: MSE = 0.5 + 0.001 K, the same for every K's test fold -- the inner search's minimum is its first setting, K = 2
    assert n.all_performances["MSE"] == [0.502] * 3 and abs(n.average_performances["MSE"] - 0.502) < 1e-12
    assert sum(n.all_performances["n_test"]) == M.sum()                  # the outer test folds partition the observed entries
    for fi in files:
        inner = open(fi).read()
        assert inner.count("Performances so far") >= 1 or "Average performances" in inner or len(inner) > 0
    with pytest.raises(AssertionError):
        MatrixNestedCrossValidation(FakeModel, R, M[:, :5], 3, 2, search, {"iterations": 2}, f, files)
