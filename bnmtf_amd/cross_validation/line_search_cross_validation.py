"""code/cross_validation/line_search_cross_validation.py (class LineSearchCrossValidation): per fold a line search over
K on the training mask, then `restarts` final models with the best K, scored on the held-out fold.  Folds x K values x
restarts are independent models: all folds' line searches run on the replica pool together, then all final models."""
import numpy

from . import mask
from ._search import METRICS, best_of_restarts
from .replicas import ReplicaPool, fit_model

metrics = ['MSE', 'AIC', 'BIC']
measures = ['R^2', 'MSE', 'Rp']
attempts_generate_M = 100


class LineSearchCrossValidation:
    def __init__(self, classifier, R, M, values_K, folds, priors, init_UV, iterations, restarts, quality_metric, file_performance,
                 *, pool=None, seed=None):
        self.classifier = classifier
        self.R = numpy.array(R, dtype=float)
        self.M = numpy.array(M)
        self.values_K = values_K
        self.folds = folds
        self.priors = priors
        self.init_UV = init_UV
        self.iterations = iterations
        self.restarts = restarts
        self.quality_metric = quality_metric
        self.fout = open(file_performance, 'w')
        (self.I, self.J) = self.R.shape
        assert (self.R.shape == self.M.shape), "R and M are of different shapes: %s and %s respectively." % (self.R.shape, self.M.shape)
        assert self.quality_metric in metrics
        self.pool, self.seed = pool, seed
        self.performances = {}

    def run(self, burn_in=None, thinning=None, minimum_TN=None):
        """:54-99."""
        folds_test = mask.compute_folds_attempts(I=self.I, J=self.J, no_folds=self.folds, attempts=attempts_generate_M, M=self.M)
        folds_training = mask.compute_Ms(folds_test)
        own = self.pool is None
        pool = self.pool or ReplicaPool(devices=[0], shared={"R": self.R})
        pool.shared.setdefault("R", self.R)
        try:
            # every (fold, K, restart) of the line searches at once
            def final_jobs(fi, K):
                return [dict(classifier=self.classifier, args=(K, self.priors), init={"init": self.init_UV}, iterations=self.iterations,
                             burn_in=burn_in, thinning=thinning, minimum_TN=minimum_TN, M=folds_training[fi], test=folds_test[fi], metrics=["loglikelihood"],
                             seed=None if self.seed is None else self.seed + 15485863 + 104729 * fi + r) for r in range(self.restarts)]
            jobs = []
            for fi, train in enumerate(folds_training):
                for ki, K in enumerate(self.values_K):
                    for r in range(self.restarts):
                        jobs.append(dict(classifier=self.classifier, args=(K, self.priors), init={"init": self.init_UV}, iterations=self.iterations,
                                         burn_in=burn_in, thinning=thinning, minimum_TN=minimum_TN, M=train, test=None, metrics=METRICS,
                                         seed=None if self.seed is None else self.seed + 104729 * fi + 7919 * ki + r))
            # A pool that fits its jobs as ONE device batch (ReplicaPool(batched=True): small models are a block each, a GPU has 256
            # CUs) takes the folds' final models along with the search -- for EVERY candidate K, the ones of the losing K are dropped:
            # the job's wall time is one batch instead of two.  Only with explicit seeds (each job then seeds its own streams; with
            # the global NumPy stream the extra models would shift every later draw).
            speculative = bool(getattr(pool, "batched", False)) and self.seed is not None
            n_search = len(jobs)
            if speculative:
                for fi in range(self.folds):
                    for K in self.values_K:
                        jobs += final_jobs(fi, K)
            res_all = pool.map(fit_model, jobs)
            res = res_all[:n_search]
            best_K = []
            for fi in range(self.folds):
                vals = []
                for ki in range(len(self.values_K)):
                    rs = res[(fi * len(self.values_K) + ki) * self.restarts:(fi * len(self.values_K) + ki + 1) * self.restarts]
                    best = rs[0]
                    for r in rs[1:]:
                        if r["quality"]["loglikelihood"] > best["quality"]["loglikelihood"]:
                            best = r
                    vals.append(best["quality"][self.quality_metric])
                self.fout.write("All model fits for fold %s, metric %s: %s.\n" % (fi + 1, self.quality_metric, vals))
                best_K.append(self.values_K[vals.index(min(vals))])
                self.fout.write("Best K for fold %s: %s.\n" % (fi + 1, best_K[-1]))
                self.fout.flush()
            # the final models of every fold (restarts each), scored on the held-out entries
            if speculative:
                res = []
                for fi in range(self.folds):
                    o = n_search + (fi * len(self.values_K) + list(self.values_K).index(best_K[fi])) * self.restarts
                    res += res_all[o:o + self.restarts]
            else:
                res = pool.map(fit_model, [job for fi in range(self.folds) for job in final_jobs(fi, best_K[fi])])
        finally:
            if own:
                pool.close()
        performances_test = {measure: [] for measure in measures}
        for fi in range(self.folds):
            rs = res[fi * self.restarts:(fi + 1) * self.restarts]
            best = rs[0]
            for r in rs[1:]:
                if r["quality"]["loglikelihood"] > best["quality"]["loglikelihood"]:
                    best = r
            self.fout.write("Performance: %s.\n\n" % best["performance"])
            self.fout.flush()
            for measure in measures:
                performances_test[measure].append(best["performance"][measure])
        average_performance_test = self.compute_average_performance(performances_test)
        message = "Average performance: %s. \nPerformances test: %s." % (average_performance_test, performances_test)
        self.fout.write(message)
        self.fout.flush()
        self.performances = performances_test
        self.average_performance = average_performance_test

    def compute_average_performance(self, performances):
        """:103-104."""
        return {measure: (sum(values) / float(len(values))) for measure, values in performances.items()}

    def run_model(self, train, test, K, burn_in=None, thinning=None, minimum_TN=None):
        """:108-135: `restarts` models with this K, the prediction of the one with the best log-likelihood."""
        pool = self.pool or ReplicaPool(devices=[0], shared={"R": self.R})
        pool.shared.setdefault("R", self.R)
        jobs = [dict(classifier=self.classifier, args=(K, self.priors), init={"init": self.init_UV}, iterations=self.iterations, burn_in=burn_in,
                     thinning=thinning, minimum_TN=minimum_TN, M=train, test=test, metrics=["loglikelihood"],
                     seed=None if self.seed is None else self.seed + r) for r in range(self.restarts)]
        rs = pool.map(fit_model, jobs)
        best = rs[0]
        for r in rs[1:]:
            if r["quality"]["loglikelihood"] > best["quality"]["loglikelihood"]:
                best = r
        return best["performance"]
