"""Drop-in for code/models/nmtf_icm.py (class nmtf_icm): Iterated Conditional Modes for MAP
non-negative matrix tri-factorisation R ~ F.S.G^T.  Sweep order F columns, S row-major, G
columns, tau (nmtf_icm.py:141-162); every update is max(0, mu) clamped from below by
minimum_TN, tau is the Gamma mode.  Deterministic: compared with the reference end to end.
"""
import math

import numpy as np

from . import _lib
from ._base import metrics_from_sums
from .bnmtf_gibbs import bnmtf_gibbs_optimised
from .nmf_icm import gamma_mode


class nmtf_icm(bnmtf_gibbs_optimised):
    def initialise(self, init_S='random', init_FG='random'):
        """:100-128: the Gibbs class's initial F, S, G; tau = gamma_mode."""
        super().initialise(init_S=init_S, init_FG=init_FG)
        self.tau = gamma_mode(self.alpha_s(), self.beta_s())

    def run(self, iterations, minimum_TN=0.):
        """:132-173; returns None like the reference."""
        it = int(iterations)
        if self._blocks is not None:           # K or L above 64: blocks (_blocked.py), the same updates with the Gamma mode for tau
            self._run_blocked(it, _lib.UPDATE_ICM, False, None, minimum_TN=float(minimum_TN), icm=True)
            return
        self._push()
        taus = np.zeros(it); perf = np.zeros((it, 3)); times = np.zeros(it)
        L = _lib.lib()
        _lib.check(L.bnmtf_set_minimum_tn(self._handle(), float(minimum_TN)))
        _lib.check(L.bnmtf_gibbs_run(self._handle(), it, _lib.UPDATE_ICM, None, None, None, _lib.ptr(taus), _lib.ptr(perf), _lib.ptr(times)))
        self._pull()
        self.all_tau = taus
        self.all_times = list(times)
        self.all_performances = {'MSE': list(perf[:, 0]), 'R^2': list(perf[:, 1]), 'Rp': list(perf[:, 2])}
        if self.verbose:
            for i in range(it):
                print("Iteration %s. MSE: %s. R^2: %s. Rp: %s." % (i + 1, perf[i, 0], perf[i, 1], perf[i, 2]))
        return

    def predict(self, M_pred):
        """:218-223."""
        return metrics_from_sums(self._metric_sums(M_pred, self.F, self.S, self.G))

    def quality(self, metric):
        """:246-264."""
        assert metric in ['loglikelihood', 'BIC', 'AIC', 'MSE', 'ELBO'], 'Unrecognised metric for model quality: %s.' % metric
        log_likelihood = self.log_likelihood()
        if metric == 'loglikelihood':
            return log_likelihood
        elif metric == 'BIC':
            return - 2 * log_likelihood + (self.I * self.K + self.K * self.L + self.J * self.L) * math.log(self.size_Omega)
        elif metric == 'AIC':
            return - 2 * log_likelihood + 2 * (self.I * self.K + self.K * self.L + self.J * self.L)
        elif metric == 'MSE':
            return metrics_from_sums(self._metric_sums(None, self.F, self.S, self.G))['MSE']
        elif metric == 'ELBO':
            return 0.

    def log_likelihood(self):
        """:266-269."""
        s = self._metric_sums(None, self.F, self.S, self.G)
        sse = s[2] - 2.0 * s[5] + s[4]
        return self.size_Omega / 2. * (math.log(self.tau) - math.log(2 * math.pi)) - self.tau / 2. * sse
