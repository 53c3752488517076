"""Drop-in for code/models/bnmf_vb_optimised.py (class bnmf_vb_optimised): variational Bayes for
Bayesian NMF, with the column updates, TN moments and exp_square_diff evaluated by libbnmtf_hip.so.

    BNMF = bnmf_vb_optimised(R, M, K, priors)
    BNMF.initialise(init='exp', tauUV={})
    BNMF.run(iterations)
"""
import ctypes as C
import math

import numpy as np
import scipy.special

from . import _lib
from ._base import DeviceModel, broadcast_lambda, check_rank, check_R_M, metrics_from_sums
from .distributions import TN_vector_expectation, TN_vector_variance, gamma_expectation, gamma_expectation_log
from ._blocked import BLOCK, MAX_BLOCKS, VBColumnBlocks


class bnmf_vb_optimised(DeviceModel):
    def __init__(self, R, M, K, priors, *, device=0, verbose=True, rank=0, world=1, comm_id=None):
        self.R = np.array(R, dtype=float)
        self.M = np.array(M, dtype=float)
        self.K = K
        check_R_M(self.R, self.M)
        check_rank("bnmf_vb_optimised", BLOCK * MAX_BLOCKS, K=self.K)
        (self.I, self.J) = self.R.shape
        self.size_Omega = self.M.sum()
        self.alpha, self.beta = float(priors['alpha']), float(priors['beta'])
        self.lambdaU = broadcast_lambda(priors['lambdaU'], (self.I, self.K), "lambdaU")
        self.lambdaV = broadcast_lambda(priors['lambdaV'], (self.J, self.K), "lambdaV")
        self.verbose = verbose
        self._init_device(0, device, rank, world, comm_id)      # VB draws nothing: the key is unused, 0 on every rank
        # ranks above 64 (the reference has no limit, :53-77): column blocks of at most 64, one device model each (_blocked.py)
        self._blocks = None
        if self.K > BLOCK:
            assert world == 1, "ranks above %d run on one GPU (column blocks: DESIGN.md section 8)" % BLOCK
            self._blocks = VBColumnBlocks(self, bnmf_vb_optimised)

    def _lambda_arrays(self):
        return self.lambdaU, self.lambdaV, None

    def close(self):
        if getattr(self, "_blocks", None) is not None:
            self._blocks.close()
        super(bnmf_vb_optimised, self).close()

    def _handle(self):
        if getattr(self, "_blocks", None) is not None:       # shape-only entry points: the first block's handle
            return self._blocks.handles()[0]
        return super(bnmf_vb_optimised, self)._handle()

    def _metric_sums(self, M_pred, A, S, B):
        if self._blocks is not None:
            if M_pred is not None:
                Mp_ = np.asarray(M_pred)
                assert ((Mp_ == 0) | (Mp_ == 1)).all(), "The indicator matrix M_pred must contain only 0 and 1."
            return self._blocks.metric_sums(M_pred, self.expU if A is None else A, self.expV if B is None else B)
        return super(bnmf_vb_optimised, self)._metric_sums(M_pred, A, S, B)

    def describe(self):
        if self._blocks is not None:
            self._blocks._prepare()
            return "column blocks %s: " % (self._blocks.ranges,) + " | ".join(ch.describe() for ch in self._blocks.children)
        return super(bnmf_vb_optimised, self).describe()

    # -- state hand-off -------------------------------------------------------
    _NAMES = ("muU", "tauU", "expU", "varU", "muV", "tauV", "expV", "varV")

    def _push(self):
        exptau = float(getattr(self, "exptau", 1.0))
        if self._blocks is not None:
            held = getattr(self, "_device_state", None)
            if held is not None and held[1] == exptau and all(np.array_equal(getattr(self, n), a) for n, a in zip(self._NAMES, held[2])):
                return
            self._blocks.push(exptau)
            self._device_state = (None, exptau, [np.array(getattr(self, n), dtype=float) for n in self._NAMES])
            return
        # the state the device holds already (nothing touched the q parameters since the last run() pulled them): no upload,
        # and the device keeps what it carries between its half sweeps -- run(a); run(b) is the trajectory of run(a + b)
        held = getattr(self, "_device_state", None)
        if held is not None and held[0] is self._h and held[1] == exptau and all(np.array_equal(getattr(self, n), a) for n, a in zip(self._NAMES, held[2])):
            return
        if held is not None and held[0] is self._h and all(np.array_equal(getattr(self, n), a) for n, a in zip(self._NAMES, held[2])):
            # only exptau moved (update_tau / update_exp_tau between two device calls): the q parameters on the device stay
            _lib.check(_lib.lib().bnmtf_set_tau(self._handle(), exptau))
            self._device_state = (held[0], exptau, held[2])
            return
        self._device_state = None
        arrs = [_lib.f64(getattr(self, n)) for n in self._NAMES]
        _lib.check(_lib.lib().bnmf_vb_set_state(self._handle(), *[_lib.ptr(a) for a in arrs], exptau))
        self._device_state = (self._h, exptau, [np.array(a, dtype=float) for a in arrs])      # (what the device holds now -- in fp32; a push of the same arrays would give it the same)

    def _pull(self):
        if self._blocks is not None:
            got = self._blocks.pull()
            for n in self._NAMES:
                setattr(self, n, got[n])
            self._device_state = (None, float(getattr(self, "exptau", 1.0)), [got[n].copy() for n in self._NAMES])
            return
        shapes = [(self.I, self.K)] * 4 + [(self.J, self.K)] * 4
        arrs = [np.zeros(s) for s in shapes]
        _lib.check(_lib.lib().bnmf_vb_get_state(self._handle(), *[_lib.ptr(a) for a in arrs]))
        for n, a in zip(self._NAMES, arrs):
            setattr(self, n, a)
        self._device_state = (self._h, None, [a.copy() for a in arrs])       # (exptau: filled in by run() once it has been formed from the device's beta_s)

    def initialise(self, init='exp', tauUV={}):
        """bnmf_vb_optimised.py:93-117."""
        self.tauU = np.array(tauUV['tauU'], dtype=float) if 'tauU' in tauUV else np.ones((self.I, self.K))
        self.tauV = np.array(tauUV['tauV'], dtype=float) if 'tauV' in tauUV else np.ones((self.J, self.K))
        assert init in ['exp', 'random'], "Unrecognised init option for F,G: %s." % init
        self.muU, self.muV = 1. / self.lambdaU, 1. / self.lambdaV
        if init == 'random':
            self.muU = self._rng().exponential(scale=1.0 / self.lambdaU)
            self.muV = self._rng().exponential(scale=1.0 / self.lambdaV)
        # (update_exp_U(k) / update_exp_V(k) of every column, :111-115: the moments are element-wise -- one device call per factor
        # instead of two per column; a model search builds dozens of models and their set-up was most of its wall time)
        from .distributions import _moments
        muU, muV = np.broadcast_to(self.muU, (self.I, self.K)), np.broadcast_to(self.muV, (self.J, self.K))
        e, v = _moments(np.ravel(muU), np.ravel(self.tauU))
        self.expU, self.varU = e.reshape(self.I, self.K), v.reshape(self.I, self.K)
        e, v = _moments(np.ravel(muV), np.ravel(self.tauV))
        self.expV, self.varV = e.reshape(self.J, self.K), v.reshape(self.J, self.K)
        self.update_tau()
        self.update_exp_tau()

    def run(self, iterations):
        """:121-153.  all_elbo (the value the reference only prints) is kept as an extra attribute."""
        it = int(iterations)
        if self._blocks is not None:
            return self._run_blocked(it)
        self._push()
        exptau = np.zeros(it); perf = np.zeros((it, 3)); terms = np.zeros((it, 10)); times = np.zeros(it)
        _lib.check(_lib.lib().bnmf_vb_run(self._handle(), it, _lib.ptr(exptau), _lib.ptr(perf), _lib.ptr(terms), _lib.ptr(times)))
        self._run_finish(it, exptau, perf, terms, times)

    def _run_finish(self, it, exptau, perf, terms, times):
        """What run() does behind the device call (batch.run_many: behind the call that ran this model among others)."""
        self._pull()
        self.all_exp_tau = list(exptau)
        self.all_times = list(times)
        self.all_performances = {'MSE': list(perf[:, 0]), 'R^2': list(perf[:, 1]), 'Rp': list(perf[:, 2])}
        self.all_elbo_terms = terms        # per iteration: exp_square_diff, beta_s, then (quad, log erfc, log tau, lambda E) sums of U and of V
        self._all_elbo = None              # (all_elbo: finished from the terms when somebody reads it -- a model search reads the last state only)
        if it > 0:
            self.alpha_s = self.alpha + self.size_Omega / 2.0
            self.beta_s = terms[-1, 1]
            self.update_exp_tau()
            if getattr(self, "_device_state", None) is not None:      # alpha_s / beta_s from the device's own beta_s: the exptau it holds
                self._device_state = (self._device_state[0], float(self.exptau), self._device_state[2])
        if self.verbose:
            for i in range(it):
                print("Iteration %s. ELBO: %s. MSE: %s. R^2: %s. Rp: %s." % (i + 1, self.all_elbo[i], perf[i, 0], perf[i, 1], perf[i, 2]))
        return

    def _run_blocked(self, it):
        """run() of a model wider than 64 columns (_blocked.py): per iteration the blocks' half sweeps in turn (:134-141), then
        update_tau / update_exp_tau (:143-144) from the full-width exp_square_diff, the metrics and the ELBO (:146-150)."""
        import time
        blocks = self._blocks
        self._push()
        self.all_exp_tau, self.all_times, self.all_elbo = [], [], []
        self.all_performances = {'MSE': [], 'R^2': [], 'Rp': []}
        t0 = time.time()
        for i in range(it):
            blocks.sweep_both()
            self._pull()
            esd, s = blocks.esd(self.expU, self.expV)
            self.alpha_s = self.alpha + self.size_Omega / 2.0
            self.beta_s = self.beta + 0.5 * esd
            self.update_exp_tau()
            blocks.set_tau(self.exptau)
            self._device_state = (None, float(self.exptau), self._device_state[2])
            perf = metrics_from_sums(s)
            elbo = self._elbo_given_esd(esd)
            for m in ('MSE', 'R^2', 'Rp'):
                self.all_performances[m].append(perf[m])
            self.all_exp_tau.append(self.exptau); self.all_elbo.append(elbo); self.all_times.append(time.time() - t0)
            if self.verbose:
                print("Iteration %s. ELBO: %s. MSE: %s. R^2: %s. Rp: %s." % (i + 1, elbo, perf['MSE'], perf['R^2'], perf['Rp']))
        return

    def train(self, iterations, init_UV='random'):
        """:157-159, as written in the reference (initialise has no init_UV keyword -> TypeError there too)."""
        self.initialise(init_UV=init_UV)
        self.run(iterations=iterations)

    # -- ELBO -------------------------------------------------------------------
    def _elbo_scalar_part(self, esd, alpha_s, beta_s, exptau, explogtau, sums_u, sums_v):
        quad_u, lerfc_u, ltau_u, lamx_u = sums_u
        quad_v, lerfc_v, ltau_v, lamx_v = sums_v
        return self.size_Omega / 2. * (explogtau - math.log(2 * math.pi)) - exptau / 2. * esd \
            + np.log(self.lambdaU).sum() - lamx_u + np.log(self.lambdaV).sum() - lamx_v \
            + self.alpha * math.log(self.beta) - scipy.special.gammaln(self.alpha) \
            + (self.alpha - 1.) * explogtau - self.beta * exptau \
            - alpha_s * math.log(beta_s) + scipy.special.gammaln(alpha_s) \
            - (alpha_s - 1.) * explogtau + beta_s * exptau \
            - .5 * ltau_u + self.I * self.K / 2. * math.log(2 * math.pi) + lerfc_u + quad_u \
            - .5 * ltau_v + self.J * self.K / 2. * math.log(2 * math.pi) + lerfc_v + quad_v

    @property
    def all_elbo(self):
        """The ELBO of every iteration of the last run() (the value the reference prints, :148-150)."""
        if getattr(self, "_all_elbo", None) is None:
            terms = getattr(self, "all_elbo_terms", None)
            self._all_elbo = [] if terms is None else [self._elbo_from_terms(t) for t in terms]
        return self._all_elbo

    @all_elbo.setter
    def all_elbo(self, value):
        self._all_elbo = value

    def _elbo_from_terms(self, t):
        esd, beta_s = t[0], t[1]
        alpha_s = self.alpha + self.size_Omega / 2.0
        return self._elbo_scalar_part(esd, alpha_s, beta_s, gamma_expectation(alpha_s, beta_s),
                                      gamma_expectation_log(alpha_s, beta_s), t[2:6], t[6:10])

    def elbo(self):
        """:163-177 for the current attributes: exp_square_diff on the device, the O((I+J)K) sums on the host."""
        return self._elbo_given_esd(self.exp_square_diff())

    def _elbo_given_esd(self, esd):
        def sums(mu, tau, ex, var, lam):
            with np.errstate(all='ignore'):
                return ((tau / 2. * (var + (ex - mu) ** 2)).sum(),
                        np.log(0.5 * scipy.special.erfc(-mu * np.sqrt(tau) / math.sqrt(2))).sum(),
                        np.log(tau).sum(), (lam * ex).sum())
        return self._elbo_scalar_part(esd, self.alpha_s, self.beta_s, self.exptau, self.explogtau,
                                      sums(self.muU, self.tauU, self.expU, self.varU, self.lambdaU),
                                      sums(self.muV, self.tauV, self.expV, self.varV, self.lambdaV))

    # -- updates ----------------------------------------------------------------
    def update_tau(self):
        """:181-183."""
        self.alpha_s = self.alpha + self.size_Omega / 2.0
        self.beta_s = self.beta + 0.5 * self.exp_square_diff()

    def masked_sums(self, which):
        """Hook (tests): sum over the MISSING entries of a unit of the other factor's S2 = var + exp^2 and of its exp^2, per column --
        the chain-independent parts of tauU / muU (which = 0) or tauV / muV (which = 1), from the matrix-core product run() uses."""
        assert self._blocks is None, "masked_sums is a hook of the single-block model (K <= %d)" % BLOCK
        self._push()
        n = self.I if which == 0 else self.J
        asq = np.zeros((n, self.K)); vsq = np.zeros((n, self.K))
        _lib.check(_lib.lib().bnmf_vb_masked_sums(self._handle(), int(which), asq.ctypes.data, vsq.ctypes.data))
        return asq, vsq

    def exp_square_diff(self):
        """:185-187 (fp64 on the device)."""
        for n in ("muU", "tauU", "muV", "tauV"):          # the test-suite sets exp/var only
            if not hasattr(self, n):
                setattr(self, n, np.ones((self.I if n.endswith("U") else self.J, self.K)))
        self._push()
        if self._blocks is not None:
            return self._blocks.esd(self.expU, self.expV)[0]
        out = C.c_double()
        _lib.check(_lib.lib().bnmf_vb_exp_square_diff(self._handle(), C.byref(out)))
        return out.value

    def _update(self, which, k, moments):
        self._push()
        if self._blocks is not None:
            self._blocks.update(which, k, moments)
            self._pull()
            return
        _lib.check(_lib.lib().bnmf_vb_update(self._handle(), which, int(k), int(moments)))
        self._pull()

    def update_U(self, k):
        """:189-191."""
        self._update(0, k, 0)

    def update_V(self, k):
        """:193-195."""
        self._update(1, k, 0)

    def update_exp_U(self, k):
        """:199-204."""
        self.expU[:, k] = TN_vector_expectation(self.muU[:, k], self.tauU[:, k])
        self.varU[:, k] = TN_vector_variance(self.muU[:, k], self.tauU[:, k])

    def update_exp_V(self, k):
        """:206-211."""
        self.expV[:, k] = TN_vector_expectation(self.muV[:, k], self.tauV[:, k])
        self.varV[:, k] = TN_vector_variance(self.muV[:, k], self.tauV[:, k])

    def update_exp_tau(self):
        """:213-215."""
        self.exptau = gamma_expectation(self.alpha_s, self.beta_s)
        self.explogtau = gamma_expectation_log(self.alpha_s, self.beta_s)

    # -- prediction / model quality ----------------------------------------------
    def predict(self, M_pred):
        """:219-224."""
        return metrics_from_sums(self._metric_sums(M_pred, self.expU, None, self.expV))

    def quality(self, metric):
        """:247-262."""
        assert metric in ['loglikelihood', 'BIC', 'AIC', 'MSE', 'ELBO'], 'Unrecognised metric for model quality: %s.' % metric
        log_likelihood = self.log_likelihood()
        if metric == 'loglikelihood':
            return log_likelihood
        elif metric == 'BIC':
            return - 2 * log_likelihood + (self.I * self.K + self.J * self.K) * math.log(self.size_Omega)
        elif metric == 'AIC':
            return - 2 * log_likelihood + 2 * (self.I * self.K + self.J * self.K)
        elif metric == 'MSE':
            return metrics_from_sums(self._metric_sums(None, self.expU, None, self.expV))['MSE']
        elif metric == 'ELBO':
            return self.elbo()

    def log_likelihood(self):
        """:264-266."""
        s = self._metric_sums(None, self.expU, None, self.expV)
        sse = s[2] - 2.0 * s[5] + s[4]
        return self.size_Omega / 2. * (self.explogtau - math.log(2 * math.pi)) - self.exptau / 2. * sse


bnmf_vb = bnmf_vb_optimised
