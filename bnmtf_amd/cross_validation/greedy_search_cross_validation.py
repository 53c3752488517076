"""code/cross_validation/greedy_search_cross_validation.py (class GreedySearchCrossValidation): per fold a greedy (K, L)
search, then `restarts` final tri-factorisations with the fold's best (K, L), scored on the held-out entries.

Kept as the reference has it: the per-fold GreedySearch is given the FULL mask `M` (greedy_search_cross_validation.py:68-
78 passes `M=self.M`, not the fold's training mask -- so every fold searches the same data and only the final models see
the split); the log lines; `run(burn_in, thinning, minimum_TN)`.  What is organised differently: the folds' searches are
walks of dependent steps, but the folds do not depend on each other, and neither do the final models -- all final models
(folds x restarts) run as one batch on the replica pool."""
import numpy

from . import mask
from .greedy_search_bnmtf import GreedySearch
from .replicas import ReplicaPool, fit_model

metrics = ['MSE', 'AIC', 'BIC']
measures = ['R^2', 'MSE', 'Rp']
attempts_generate_M = 1000


class GreedySearchCrossValidation(object):
    def __init__(self, classifier, R, M, values_K, values_L, folds, priors, init_S, init_FG, iterations, restarts, quality_metric, file_performance,
                 *, pool=None, seed=None):
        self.classifier = classifier
        self.R = numpy.array(R, dtype=float)
        self.M = numpy.array(M)
        self.values_K, self.values_L = values_K, values_L
        self.folds = folds
        self.priors = priors
        self.init_S, self.init_FG = init_S, init_FG
        self.iterations, self.restarts = iterations, restarts
        self.quality_metric = quality_metric
        self.fout = open(file_performance, 'w')
        (self.I, self.J) = self.R.shape
        assert (self.R.shape == self.M.shape), "R and M are of different shapes: %s and %s respectively." % (self.R.shape, self.M.shape)
        assert self.quality_metric in metrics
        self.pool, self.seed = pool, seed
        self.performances = {}

    def run(self, burn_in=None, thinning=None, minimum_TN=None):
        """:58-104."""
        folds_test = mask.compute_folds_attempts(I=self.I, J=self.J, no_folds=self.folds, attempts=attempts_generate_M, M=self.M)
        folds_training = mask.compute_Ms(folds_test)
        own = self.pool is None
        pool = self.pool or ReplicaPool(devices=[0], shared={"R": self.R})
        pool.shared.setdefault("R", self.R)
        try:
            side_by_side = self.seed is not None and (len(getattr(pool, "devices", [])) > 1 or getattr(pool, "batched", False))
            # (a batched pool: the steps the folds have open at the same time go to the device as one call, a block per model)
            walk_pool = pool.joint(self.folds) if side_by_side and getattr(pool, "batched", False) else pool

            def fold_search(fi):
                try:
                    search = GreedySearch(classifier=self.classifier, values_K=self.values_K, values_L=self.values_L, R=self.R, M=self.M,
                                          priors=self.priors, initS=self.init_S, initFG=self.init_FG, iterations=self.iterations, restarts=self.restarts,
                                          pool=walk_pool, seed=None if self.seed is None else self.seed + 104729 * fi)
                    search.search(self.quality_metric, burn_in=burn_in, thinning=thinning, minimum_TN=minimum_TN)
                finally:
                    if walk_pool is not pool:
                        walk_pool.leave()
                best = search.best_value(metric=self.quality_metric)
                # (written below, fold by fold, each followed by its Performance line: the reference's order, greedy_search_cross_validation.py:68-100)
                return best, ("All model fits for fold %s, metric %s: %s.\n" % (fi + 1, self.quality_metric, search.all_values(metric=self.quality_metric))
                              + "Best K,L for fold %s: %s.\n" % (fi + 1, best))
            # A fold's walk is a chain of dependent steps of one to three fits, but the folds do not depend on each other: with
            # several replica slots and seeded candidates (every fit then draws from its own stream, whatever runs beside it) the
            # walks advance side by side, each from a thread of its own that hands its steps to the shared pool.  Unseeded, the
            # candidates draw from the workers' global streams in the order they run: fold by fold, as the reference.
            if side_by_side:
                from concurrent.futures import ThreadPoolExecutor
                with ThreadPoolExecutor(max_workers=self.folds) as ex:
                    done = list(ex.map(fold_search, range(self.folds)))
            else:
                done = [fold_search(fi) for fi in range(self.folds)]
            best_KL, fold_lines = [d[0] for d in done], [d[1] for d in done]
            final = pool.map(fit_model, [job for fi, (train, test) in enumerate(zip(folds_training, folds_test))
                                         for job in self._final_jobs(train, test, best_KL[fi][0], best_KL[fi][1], burn_in, thinning, minimum_TN, fi)])
        finally:
            if own:
                pool.close()
        performances_test = {measure: [] for measure in measures}
        for fi in range(self.folds):
            performance = self._best(final[fi * self.restarts:(fi + 1) * self.restarts])
            self.fout.write(fold_lines[fi])
            self.fout.write("Performance: %s.\n\n" % performance)
            self.fout.flush()
            for measure in measures:
                performances_test[measure].append(performance[measure])
        average_performance_test = self.compute_average_performance(performances_test)
        self.fout.write("Average performance: %s. \nPerformances test: %s." % (average_performance_test, performances_test))
        self.fout.flush()
        self.performances = performances_test
        self.average_performance = average_performance_test

    def _final_jobs(self, train, test, K, L, burn_in, thinning, minimum_TN, fold=0):
        return [dict(classifier=self.classifier, args=(K, L, self.priors), init={"init_S": self.init_S, "init_FG": self.init_FG},
                     iterations=self.iterations, burn_in=burn_in, thinning=thinning, minimum_TN=minimum_TN, M=train, test=test,
                     metrics=["loglikelihood"], seed=None if self.seed is None else self.seed + 15485863 + 104729 * fold + r)
                for r in range(self.restarts)]

    @staticmethod
    def _best(results):
        best = results[0]
        for r in results[1:]:                 # the first strictly better restart wins (:131-133)
            if r["quality"]["loglikelihood"] > best["quality"]["loglikelihood"]:
                best = r
        return best["performance"]

    def compute_average_performance(self, performances):
        """:108-109."""
        return {measure: (sum(values) / float(len(values))) for measure, values in performances.items()}

    def run_model(self, train, test, K, L, burn_in=None, thinning=None, minimum_TN=None):
        """:113-142: `restarts` models with this (K, L), the prediction of the one with the best log-likelihood."""
        pool = self.pool or ReplicaPool(devices=[0], shared={"R": self.R})
        pool.shared.setdefault("R", self.R)
        return self._best(pool.map(fit_model, self._final_jobs(train, test, K, L, burn_in, thinning, minimum_TN)))
