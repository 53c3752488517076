"""Shared by the search classes: fit (candidate x restart) models on the replica pool and keep, per candidate, the
restart with the best log-likelihood -- the rule of line_search_bnmf.py:63-74 / grid_search_bnmtf.py:66-78."""
from .replicas import ReplicaPool, fit_model

METRICS = ['BIC', 'AIC', 'loglikelihood', 'MSE', 'ELBO']


def best_of_restarts(pool, classifier, R, M, candidates, init, iterations, restarts, burn_in, thinning, minimum_TN, seed=None):
    """candidates: list of positional-argument tuples after (R, M).  Returns one {metric: value} per candidate."""
    own = pool is None
    pool = pool or ReplicaPool(devices=[0], shared={"R": R})
    if "R" not in pool.shared:
        pool.shared["R"] = R
    jobs = []
    for ci, args in enumerate(candidates):
        for r in range(restarts):
            jobs.append(dict(classifier=classifier, args=args, init=init, iterations=iterations, burn_in=burn_in, thinning=thinning,
                             minimum_TN=minimum_TN, M=M, test=None, metrics=METRICS,
                             seed=None if seed is None else seed + 7919 * ci + r))
    try:
        results = pool.map(fit_model, jobs)
    finally:
        if own:
            pool.close()
    out = []
    for ci in range(len(candidates)):
        rs = results[ci * restarts:(ci + 1) * restarts]
        best = rs[0]
        for r in rs[1:]:                                   # first strictly better restart wins, as in the reference loop
            if r["quality"]["loglikelihood"] > best["quality"]["loglikelihood"]:
                best = r
        out.append(best["quality"])
    return out
