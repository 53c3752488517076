"""code/cross_validation/grid_search_bnmtf.py (class GridSearch): every (K, L) of the grid, `restarts` models each, on
the replica pool."""
import numpy

from ._search import METRICS, best_of_restarts

metrics = METRICS


class GridSearch:
    def __init__(self, classifier, values_K, values_L, R, M, priors, initS, initFG, iterations, restarts=1, *, pool=None, seed=None):
        self.classifier = classifier
        self.values_K = values_K
        self.values_L = values_L
        self.R = R
        self.M = M
        (self.I, self.J) = self.R.shape
        self.priors = priors
        self.initS = initS
        self.initFG = initFG
        self.iterations = iterations
        self.restarts = restarts
        assert self.restarts > 0, "Need at least 1 restart."
        self.pool, self.seed = pool, seed
        self.all_performances = {metric: numpy.empty((len(self.values_K), len(self.values_L))) for metric in metrics}

    def search(self, burn_in=None, thinning=None):
        """:60-91 (priors given as scalars are broadcast per (K, L), :66-69)."""
        cands = []
        for K in self.values_K:
            for L in self.values_L:
                priors = self.priors.copy()
                priors['lambdaF'] = self.priors['lambdaF'] * numpy.ones((self.I, K))
                priors['lambdaS'] = self.priors['lambdaS'] * numpy.ones((K, L))
                priors['lambdaG'] = self.priors['lambdaG'] * numpy.ones((self.J, L))
                cands.append((K, L, priors))
        q = best_of_restarts(self.pool, self.classifier, self.R, self.M, cands, {"init_S": self.initS, "init_FG": self.initFG},
                             self.iterations, self.restarts, burn_in, thinning, None, self.seed)
        for n, quality in enumerate(q):
            ik, il = n // len(self.values_L), n % len(self.values_L)
            for metric in metrics:
                self.all_performances[metric][ik, il] = quality[metric]

    def all_values(self, metric):
        assert metric in metrics, "Unrecognised metric name: %s." % metric
        return self.all_performances[metric]

    def best_value(self, metric):
        """:104-107."""
        assert metric in metrics, "Unrecognised metric name: %s." % metric
        index, row_length = numpy.argmin(self.all_values(metric)), len(self.values_L)
        return (self.values_K[index // row_length], self.values_L[index % row_length])
