"""Drop-in for code/models/bnmtf_vb_optimised.py (class bnmtf_vb_optimised): variational Bayes for
Bayesian non-negative matrix tri-factorisation R ~ F.S.G^T on an MI355X.

    BNMTF = bnmtf_vb_optimised(R, M, K, L, priors)
    BNMTF.initialise(init_S, init_FG, tauFSG={})     # init_S: 'random'|'exp'; init_FG: 'random'|'exp'|'kmeans'
    BNMTF.run(iterations)

Per iteration (bnmtf_vb_optimised.py:170-192): the K.L entries of S, the K columns of F and the L columns of G are
each updated in a freshly shuffled order.  The orders are drawn HERE with `random.shuffle` -- the same three calls on
the same lists as the reference, so `random.seed(s)` reproduces its trajectory -- and handed to the device, which
runs all iterations in one call.  K, L <= 32."""
import ctypes as C
import itertools
import math
import random

import numpy as np
import scipy.special

from . import _lib
from ._base import DeviceModel, broadcast_lambda, check_rank, check_R_M, metrics_from_sums
from .distributions import TN_vector_expectation, TN_vector_variance, gamma_expectation, gamma_expectation_log
from .kmeans import KMeans


class bnmtf_vb_optimised(DeviceModel):
    def __init__(self, R, M, K, L, priors, *, device=0, verbose=True, rank=0, world=1, comm_id=None):
        self.R = np.array(R, dtype=float)
        self.M = np.array(M, dtype=float)
        self.K, self.L = K, L
        check_R_M(self.R, self.M)
        check_rank("bnmtf_vb_optimised", 32, K=self.K, L=self.L)
        (self.I, self.J) = self.R.shape
        self.size_Omega = self.M.sum()
        self.alpha, self.beta = float(priors['alpha']), float(priors['beta'])
        self.lambdaF = broadcast_lambda(priors['lambdaF'], (self.I, self.K), "lambdaF")
        self.lambdaS = broadcast_lambda(priors['lambdaS'], (self.K, self.L), "lambdaS")
        self.lambdaG = broadcast_lambda(priors['lambdaG'], (self.J, self.L), "lambdaG")
        self.verbose = verbose
        # (VB draws nothing on the device; rank / world / comm_id: rows of F and columns of G over several GPUs, round 6)
        self._init_device(0, device, rank, world, comm_id)

    def _lambda_arrays(self):
        return self.lambdaF, self.lambdaG, self.lambdaS

    # -- state hand-off -------------------------------------------------------
    _NAMES = ("muF", "tauF", "expF", "varF", "muS", "tauS", "expS", "varS", "muG", "tauG", "expG", "varG")

    def _shapes(self):
        return [(self.I, self.K)] * 4 + [(self.K, self.L)] * 4 + [(self.J, self.L)] * 4

    def _push(self):
        for n, s in zip(self._NAMES, self._shapes()):      # the reference's tests set only some of the attributes
            if not hasattr(self, n):
                setattr(self, n, np.ones(s))
        arrs = [_lib.f64(getattr(self, n)) for n in self._NAMES]
        _lib.check(_lib.lib().bnmtf_vb_set_state(self._handle(), *[_lib.ptr(a) for a in arrs], float(getattr(self, "exptau", 1.0))))

    def _pull(self):
        arrs = [np.zeros(s) for s in self._shapes()]
        _lib.check(_lib.lib().bnmtf_vb_get_state(self._handle(), *[_lib.ptr(a) for a in arrs]))
        for n, a in zip(self._NAMES, arrs):
            setattr(self, n, a)

    # -- initialise / run -------------------------------------------------------
    def train(self, init_S, init_FG, iterations):
        """:100-102."""
        self.initialise(init_S, init_FG)
        return self.run(iterations)

    def initialise(self, init_S='random', init_FG='random', tauFSG={}):
        """:107-156."""
        self.tauF = np.array(tauFSG['tauF'], dtype=float) if 'tauF' in tauFSG else np.ones((self.I, self.K))
        self.tauS = np.array(tauFSG['tauS'], dtype=float) if 'tauS' in tauFSG else np.ones((self.K, self.L))
        self.tauG = np.array(tauFSG['tauG'], dtype=float) if 'tauG' in tauFSG else np.ones((self.J, self.L))
        assert init_S in ['exp', 'random'], "Unrecognised init option for S: %s." % init_S
        self.muS = 1. / self.lambdaS
        if init_S == 'random':
            self.muS = self._rng().exponential(scale=1.0 / self.lambdaS)
        assert init_FG in ['exp', 'random', 'kmeans'], "Unrecognised init option for F,G: %s." % init_FG
        self.muF, self.muG = 1. / self.lambdaF, 1. / self.lambdaG
        if init_FG == 'random':
            self.muF = self._rng().exponential(scale=1.0 / self.lambdaF)
            self.muG = self._rng().exponential(scale=1.0 / self.lambdaG)
        elif init_FG == 'kmeans':
            if self.verbose: print("Initialising F using KMeans.")
            kmeans_F = KMeans(self.R, self.M, self.K, device=self._device)
            kmeans_F.initialise()
            kmeans_F.cluster()
            self.muF = kmeans_F.clustering_results
            if self.verbose: print("Initialising G using KMeans.")
            kmeans_G = KMeans(self.R.T, self.M.T, self.L, device=self._device)
            kmeans_G.initialise()
            kmeans_G.cluster()
            self.muG = kmeans_G.clustering_results
        self.expF, self.varF = np.zeros((self.I, self.K)), np.zeros((self.I, self.K))
        self.expS, self.varS = np.zeros((self.K, self.L)), np.zeros((self.K, self.L))
        self.expG, self.varG = np.zeros((self.J, self.L)), np.zeros((self.J, self.L))
        for k in range(self.K):
            self.update_exp_F(k)
        for k, l in itertools.product(range(self.K), range(self.L)):
            self.update_exp_S(k, l)
        for l in range(self.L):
            self.update_exp_G(l)
        self.update_tau()
        self.update_exp_tau()

    def _draw_orders(self, iterations):
        """The three shuffles of every iteration, as the reference's run() makes them (:171-186)."""
        K, L = self.K, self.L
        out = np.zeros((iterations, K * L + K + L), dtype=np.int32)
        for it in range(iterations):
            indices_kl = list(itertools.product(range(K), range(L)))
            random.shuffle(indices_kl)
            indices_k = list(range(K))
            random.shuffle(indices_k)
            indices_l = list(range(L))
            random.shuffle(indices_l)
            out[it, :K * L] = [k * L + l for k, l in indices_kl]
            out[it, K * L:K * L + K] = indices_k
            out[it, K * L + K:] = indices_l
        return out

    def run(self, iterations, orders=None):
        """:160-205.  orders (optional): int array [iterations][K L + K + L] of update orders instead of fresh shuffles."""
        it = int(iterations)
        orders = self._draw_orders(it) if orders is None else np.ascontiguousarray(orders, dtype=np.int32)
        assert orders.shape == (it, self.K * self.L + self.K + self.L)
        self._push()
        exptau = np.zeros(it); perf = np.zeros((it, 3)); terms = np.zeros((it, 10)); times = np.zeros(it)
        _lib.check(_lib.lib().bnmtf_vb_run(self._handle(), it, _lib.ptr(orders), _lib.ptr(exptau), _lib.ptr(perf), _lib.ptr(terms), _lib.ptr(times)))
        self._pull()
        self.all_exp_tau = list(exptau)
        self.all_times = list(times)
        self.all_performances = {'MSE': list(perf[:, 0]), 'R^2': list(perf[:, 1]), 'Rp': list(perf[:, 2])}
        if it > 0:
            self.alpha_s = self.alpha + self.size_Omega / 2.0
            self.beta_s = terms[-1, 1]
            self.update_exp_tau()
        # the ELBO the reference prints per iteration: only the last one can be finished here (the K.L terms of S need
        # the q(S) of that iteration); earlier iterations carry the F / G / tau part
        self.all_elbo_terms = terms
        if self.verbose:
            for i in range(it):
                print("Iteration %s. MSE: %s. R^2: %s. Rp: %s." % (i + 1, perf[i, 0], perf[i, 1], perf[i, 2]))
        return

    # -- ELBO -------------------------------------------------------------------
    def elbo(self):
        """:208-223: exp_square_diff on the device, the O(IK + KL + JL) sums on the host."""
        with np.errstate(all='ignore'):
            v = self.size_Omega / 2. * (self.explogtau - math.log(2 * math.pi)) - self.exptau / 2. * self.exp_square_diff() \
                + self.alpha * math.log(self.beta) - scipy.special.gammaln(self.alpha) \
                + (self.alpha - 1.) * self.explogtau - self.beta * self.exptau \
                - self.alpha_s * math.log(self.beta_s) + scipy.special.gammaln(self.alpha_s) \
                - (self.alpha_s - 1.) * self.explogtau + self.beta_s * self.exptau
            for lam, e, var, mu, tau, n in ((self.lambdaF, self.expF, self.varF, self.muF, self.tauF, self.I * self.K),
                                            (self.lambdaS, self.expS, self.varS, self.muS, self.tauS, self.K * self.L),
                                            (self.lambdaG, self.expG, self.varG, self.muG, self.tauG, self.J * self.L)):
                v += np.log(lam).sum() - (lam * e).sum()
                v += -.5 * np.log(tau).sum() + n / 2. * math.log(2 * math.pi) \
                    + np.log(0.5 * scipy.special.erfc(-mu * np.sqrt(tau) / math.sqrt(2))).sum() + (tau / 2. * (var + (e - mu) ** 2)).sum()
            return v

    def triple_dot(self, M1, M2, M3):
        """:226-227 (host utility on explicit arrays)."""
        return np.dot(M1, np.dot(M2, M3))

    # -- updates ----------------------------------------------------------------
    def update_tau(self):
        """:231-233."""
        self.alpha_s = self.alpha + self.size_Omega / 2.0
        self.beta_s = self.beta + 0.5 * self.exp_square_diff()

    def _esd_and_sums(self):
        self._push()
        esd = C.c_double(); sums = np.zeros(6)
        _lib.check(_lib.lib().bnmtf_vb_exp_square_diff(self._handle(), C.byref(esd), _lib.ptr(sums)))
        return esd.value, sums

    def exp_square_diff(self):
        """:235-239 (fp64 on the device)."""
        return self._esd_and_sums()[0]

    def _update(self, which, k, l, moments):
        self._push()
        _lib.check(_lib.lib().bnmtf_vb_update(self._handle(), which, int(k), int(l), int(moments)))
        self._pull()

    def update_F(self, k):
        """:241-250."""
        self._update(0, k, 0, 0)

    def update_S(self, k, l):
        """:252-262."""
        self._update(1, k, l, 0)

    def update_G(self, l):
        """:264-273."""
        self._update(2, 0, l, 0)

    def update_exp_F(self, k):
        """:276-278."""
        self.expF[:, k] = TN_vector_expectation(self.muF[:, k], self.tauF[:, k])
        self.varF[:, k] = TN_vector_variance(self.muF[:, k], self.tauF[:, k])

    def update_exp_S(self, k, l):
        """:280-282."""
        self.expS[k, l] = TN_vector_expectation([self.muS[k, l]], [self.tauS[k, l]])[0]
        self.varS[k, l] = TN_vector_variance([self.muS[k, l]], [self.tauS[k, l]])[0]

    def update_exp_G(self, l):
        """:284-285."""
        self.expG[:, l] = TN_vector_expectation(self.muG[:, l], self.tauG[:, l])
        self.varG[:, l] = TN_vector_variance(self.muG[:, l], self.tauG[:, l])

    def update_exp_tau(self):
        """:286-288."""
        self.exptau = gamma_expectation(self.alpha_s, self.beta_s)
        self.explogtau = gamma_expectation_log(self.alpha_s, self.beta_s)

    # -- prediction / model quality ----------------------------------------------
    def predict(self, M_pred):
        """:292-297."""
        return metrics_from_sums(self._metric_sums(M_pred, self.expF, self.expS, self.expG))

    def quality(self, metric):
        """:320-337."""
        assert metric in ['loglikelihood', 'BIC', 'AIC', 'MSE', 'ELBO'], 'Unrecognised metric for model quality: %s.' % metric
        log_likelihood = self.log_likelihood()
        npar = self.I * self.K + self.K * self.L + self.J * self.L
        if metric == 'loglikelihood':
            return log_likelihood
        elif metric == 'BIC':
            return - 2 * log_likelihood + npar * math.log(self.size_Omega)
        elif metric == 'AIC':
            return - 2 * log_likelihood + 2 * npar
        elif metric == 'MSE':
            return metrics_from_sums(self._metric_sums(None, self.expF, self.expS, self.expG))['MSE']
        elif metric == 'ELBO':
            return self.elbo()

    def log_likelihood(self):
        """:339-342."""
        s = self._metric_sums(None, self.expF, self.expS, self.expG)
        sse = s[2] - 2.0 * s[5] + s[4]
        return self.size_Omega / 2. * (self.explogtau - math.log(2 * math.pi)) - self.exptau / 2. * sse


bnmtf_vb = bnmtf_vb_optimised
