"""code/cross_validation/greedy_search_bnmtf.py (class GreedySearch): walk the (K, L) grid from its corner, at each
point trying K+1, L+1 and both.  The walk is sequential; the up to three new points of a step (x restarts) are
independent models and run together on the replica pool.

One line of the reference is kept AS WRITTEN behind `as_written=True` (the default): walking along the last L (the K edge,
greedy_search_bnmtf.py:155-165) a successful step sets `performance_so_far = performance_new_L` -- the value left over from
the main loop, not the step's own `performance_new_K` -- so the stopping rule of the following steps compares against a
stale number (and when the main loop never ran, e.g. a single value of L, the reference raises NameError at that line:
so does this class).  `as_written=False` uses `performance_new_K`, what the symmetric L-edge loop (:140-152) does."""
from ._search import METRICS, best_of_restarts

metrics = METRICS


class GreedySearch:
    def __init__(self, classifier, values_K, values_L, R, M, priors, initS, initFG, iterations, restarts=1, *, pool=None, seed=None, as_written=True):
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
        self.pool, self.seed, self.as_written = pool, seed, as_written
        self.all_performances = {metric: [] for metric in metrics}

    def search(self, search_metric, burn_in=None, thinning=None, minimum_TN=None):
        """:69-168."""
        assert search_metric in metrics, "Unrecognised metric name: %s." % search_metric

        def try_KLs(pairs):
            """performance of every (K, L) of `pairs` under search_metric; new ones are fitted together"""
            new = [p for p in dict.fromkeys(pairs) if not self.find_KL(search_metric, *p)]
            if new:
                q = best_of_restarts(self.pool, self.classifier, self.R, self.M, [(K, L, self.priors) for K, L in new],
                                     {"init_S": self.initS, "init_FG": self.initFG}, self.iterations, self.restarts,
                                     burn_in, thinning, minimum_TN, self.seed)
                for (K, L), quality in zip(new, q):
                    for metric in metrics:
                        self.all_performances[metric].append((K, L, quality[metric]))
            return [self.find_KL(search_metric, K, L)[0][2] for K, L in pairs]

        if self.as_written and len(self.values_L) == 1 and len(self.values_K) > 1:
            # the reference's NameError (greedy_search_bnmtf.py:165: `performance_new_L` is only bound inside the main loop, which a
            # single value of L never enters) comes after the first SUCCESSFUL step along K, i.e. after models have been fitted:
            # said here, before any is
            import warnings
            warnings.warn("GreedySearch(as_written=True) with a single value in values_L raises the reference's NameError at the first "
                          "successful step along K (greedy_search_bnmtf.py:165); as_written=False walks the K edge by the symmetric rule")
        ik, il = 0, 0
        current_K, current_L = self.values_K[ik], self.values_L[il]
        performance_so_far = try_KLs([(current_K, current_L)])[0]
        stale_new_L = None                   # the main loop's last performance_new_L (see the header)
        while ik < len(self.values_K) - 1 and il < len(self.values_L) - 1:
            new_K, new_L = self.values_K[ik + 1], self.values_L[il + 1]
            performance_new_K, performance_new_L, performance_new_KL = try_KLs([(new_K, current_L), (current_K, new_L), (new_K, new_L)])
            stale_new_L = performance_new_L
            if performance_so_far < min(performance_new_K, performance_new_L, performance_new_KL):
                break
            if performance_new_K < performance_new_L and performance_new_K < performance_new_KL:
                ik += 1; current_K = new_K; performance_so_far = performance_new_K
            elif performance_new_L < performance_new_KL:
                il += 1; current_L = new_L; performance_so_far = performance_new_L
            else:
                ik += 1; il += 1; current_K, current_L = new_K, new_L; performance_so_far = performance_new_KL
        # at an edge of the grid the walk continues along it (:140-166)
        if ik == len(self.values_K) - 1:
            while il < len(self.values_L) - 1:
                new_L = self.values_L[il + 1]
                performance_new_L = try_KLs([(current_K, new_L)])[0]
                if performance_so_far < performance_new_L:
                    break
                il += 1; current_L = new_L; performance_so_far = performance_new_L
        elif il == len(self.values_L) - 1:
            while ik < len(self.values_K) - 1:
                new_K = self.values_K[ik + 1]
                performance_new_K = try_KLs([(new_K, current_L)])[0]
                if performance_so_far < performance_new_K:
                    break
                ik += 1; current_K = new_K
                if not self.as_written:
                    performance_so_far = performance_new_K
                elif stale_new_L is None:
                    raise NameError("name 'performance_new_L' is not defined")       # greedy_search_bnmtf.py:165 before any main-loop step
                else:
                    performance_so_far = stale_new_L

    def all_values(self, metric):
        assert metric in metrics, "Unrecognised metric name: %s." % metric
        return self.all_performances[metric]

    def find_KL(self, metric, K, L):
        """:175-177."""
        return list(filter(lambda x: (x[0], x[1]) == (K, L), self.all_values(metric)))

    def best_value(self, metric):
        """:180-183."""
        assert metric in metrics, "Unrecognised metric name: %s." % metric
        (best_K, best_L, best_metric) = min(self.all_performances[metric], key=lambda x: x[2])
        return (best_K, best_L)
