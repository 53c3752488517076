"""code/cross_validation/line_search_bnmf.py (class LineSearch): try each K, `restarts` models per K, keep the one with
the best log-likelihood, record BIC / AIC / loglikelihood / MSE / ELBO.  The (K, restart) models are independent: they
run on the replica pool."""
from ._search import METRICS, best_of_restarts

metrics = METRICS


class LineSearch:
    def __init__(self, classifier, values_K, R, M, priors, initUV, iterations, restarts=1, *, pool=None, seed=None):
        self.classifier = classifier
        self.values_K = values_K
        self.R = R
        self.M = M
        (self.I, self.J) = self.R.shape
        self.priors = priors
        self.initUV = initUV
        self.iterations = iterations
        self.restarts = restarts
        assert self.restarts > 0, "Need at least 1 restart."
        self.pool, self.seed = pool, seed
        self.all_performances = {metric: [] for metric in metrics}

    def search(self, burn_in=None, thinning=None, minimum_TN=None):
        """:52-79."""
        q = best_of_restarts(self.pool, self.classifier, self.R, self.M, [(K, self.priors) for K in self.values_K],
                             {"init": self.initUV}, self.iterations, self.restarts, burn_in, thinning, minimum_TN, self.seed)
        for quality in q:
            for metric in metrics:
                self.all_performances[metric].append(quality[metric])

    def all_values(self, metric):
        assert metric in metrics, "Unrecognised metric name: %s." % metric
        return self.all_performances[metric]

    def best_value(self, metric):
        """:86-88: the K with the LOWEST value of the metric."""
        assert metric in metrics, "Unrecognised metric name: %s." % metric
        return self.values_K[self.all_values(metric).index(min(self.all_values(metric)))]
