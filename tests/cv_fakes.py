"""A CPU-only stand-in classifier for the scheduling tests of bnmtf_amd.cross_validation (no GPU, no library): same
duck-typed surface the drivers use (ctor(R, M, K, priors), initialise, run, quality, predict, train), deterministic
numbers that depend on (K, seed) and record where the model ran."""
import os
import time

import numpy as np


class FakeModel(object):
    def __init__(self, R, M, K, priors, *, seed=None, device=0, verbose=True):
        self.R, self.M, self.K, self.priors, self.seed, self.device = np.asarray(R), np.asarray(M), K, priors, seed, device
        self.ran = None

    def initialise(self, init='random'):
        assert init in ('random', 'exp')
        self.init = init

    def run(self, iterations, minimum_TN=None, expectation=None, store_samples=True):
        if self.K == 13:
            raise ValueError("unlucky K")
        time.sleep(0.05)                      # a fit takes a while: the other workers get their share of the queue
        self.ran = dict(iterations=iterations, expectation=expectation, store_samples=store_samples, minimum_TN=minimum_TN)

    def train(self, iterations, init='random'):
        self.initialise(init); self.run(iterations)

    def quality(self, metric, burn_in=None, thinning=None):
        # best K is 4; among restarts the log-likelihood grows with (seed mod 3)
        r = 0 if self.seed is None else self.seed % 3
        base = {"loglikelihood": -100.0 * (self.K - 4) ** 2 + r, "BIC": 10.0 * (self.K - 4) ** 2 - r, "AIC": 9.0 * (self.K - 4) ** 2 - r,
                "MSE": 1.0 + (self.K - 4) ** 2 - 0.01 * r, "ELBO": 0.0}
        return base[metric]

    def predict(self, M_pred, burn_in=None, thinning=None):
        n = float(np.asarray(M_pred).sum())
        return {"MSE": 0.5 + 0.001 * self.K, "R^2": 0.9, "Rp": 0.95, "n_test": n, "device": self.device, "pid": os.getpid(),
                "expectation_burn_in": -1 if (self.ran is None or self.ran["expectation"] is None) else self.ran["expectation"][0]}


class FakeTri(FakeModel):
    def __init__(self, R, M, K, L, priors, *, seed=None, device=0, verbose=True):
        FakeModel.__init__(self, R, M, K, priors, seed=seed, device=device)
        self.L = L

    def initialise(self, init_S='random', init_FG='random'):
        pass

    def quality(self, metric, burn_in=None, thinning=None):
        v = (self.K - 3) ** 2 + (self.L - 5) ** 2            # minimum at (3, 5)
        return {"loglikelihood": -v, "BIC": v, "AIC": v, "MSE": v, "ELBO": 0.0}[metric]
