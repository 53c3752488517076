"""Drop-in for code/models/nmf_icm.py (class nmf_icm): Iterated Conditional Modes for MAP
non-negative matrix factorisation.  Same conditional parameters as the Gibbs sampler
(tauU/muU/tauV/muV, nmf_icm.py:159-169), but every update takes the mode of the truncated
normal, max(0, mu), clamped from below by minimum_TN (:128-135), and tau takes the Gamma
mode (alpha_s - 1) / beta_s (:137, distributions/gamma.py:27-29).  The whole trajectory is
deterministic, so it is compared with the reference end to end.

    NMF = nmf_icm(R, M, K, priors)
    NMF.initialise(init)                    # 'random' | 'exp'
    NMF.run(iterations, minimum_TN=0.)      # fills all_tau, all_times, all_performances
    NMF.predict(M_pred); NMF.quality(metric)
"""
import math

import numpy as np

from . import _lib
from ._base import metrics_from_sums
from .bnmf_gibbs import bnmf_gibbs_optimised


def gamma_mode(alpha, beta):
    """distributions/gamma.py:27-29."""
    return (float(alpha) - 1) / float(beta)


class nmf_icm(bnmf_gibbs_optimised):
    def initialise(self, init='random'):
        """:93-111 ('random' consumes numpy.random.exponential in the reference's (i,k) order)."""
        assert init in ['random', 'exp'], "Unknown initialisation option: %s. Should be 'random' or 'exp'." % init
        if init == 'random':
            self.U = self._rng().exponential(scale=1.0 / self.lambdaU)
            self.V = self._rng().exponential(scale=1.0 / self.lambdaV)
        else:
            self.U = 1.0 / self.lambdaU
            self.V = 1.0 / self.lambdaV
        self.tau = gamma_mode(self.alpha_s(), self.beta_s())

    def run(self, iterations, minimum_TN=0.):
        """:114-150.  One device call runs all iterations; returns None like the reference."""
        it = int(iterations)
        if self._blocks is not None:            # ranks above 64: column blocks (_blocked.py), ICM rules
            self._run_blocked(it, _lib.UPDATE_ICM, False, None, minimum_TN=float(minimum_TN), icm=True)
            return
        self._push()
        taus = np.zeros(it); perf = np.zeros((it, 3)); times = np.zeros(it)
        L = _lib.lib()
        _lib.check(L.bnmtf_set_minimum_tn(self._handle(), float(minimum_TN)))
        _lib.check(L.bnmf_gibbs_run(self._handle(), it, _lib.UPDATE_ICM, None, None, _lib.ptr(taus), _lib.ptr(perf), _lib.ptr(times)))
        self._pull()
        self.all_tau = taus
        self.all_times = list(times)
        self.all_performances = {'MSE': list(perf[:, 0]), 'R^2': list(perf[:, 1]), 'Rp': list(perf[:, 2])}
        if self.verbose:
            for i in range(it):
                print("Iteration %s. MSE: %s. R^2: %s. Rp: %s." % (i + 1, perf[i, 0], perf[i, 1], perf[i, 2]))
        return

    def predict(self, M_pred):
        """:173-178: metrics of the current point estimate on M_pred."""
        return metrics_from_sums(self._metric_sums(M_pred, self.U, None, self.V))

    def quality(self, metric):
        """:201-217."""
        assert metric in ['loglikelihood', 'BIC', 'AIC', 'MSE', 'ELBO'], 'Unrecognised metric for model quality: %s.' % metric
        log_likelihood = self.log_likelihood()
        if metric == 'loglikelihood':
            return log_likelihood
        elif metric == 'BIC':
            return - 2 * log_likelihood + (self.I * self.K + self.J * self.K) * math.log(self.size_Omega)
        elif metric == 'AIC':
            return - 2 * log_likelihood + 2 * (self.I * self.K + self.J * self.K)
        elif metric == 'MSE':
            return metrics_from_sums(self._metric_sums(None, self.U, None, self.V))['MSE']
        elif metric == 'ELBO':
            return 0.

    def log_likelihood(self):
        """:219-222."""
        s = self._metric_sums(None, self.U, None, self.V)
        sse = s[2] - 2.0 * s[5] + s[4]
        return self.size_Omega / 2. * (math.log(self.tau) - math.log(2 * math.pi)) - self.tau / 2. * sse
