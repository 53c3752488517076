"""Drop-in for code/models/bnmf_gibbs_optimised.py (class bnmf_gibbs_optimised):
Gibbs sampler for Bayesian non-negative matrix factorisation R ~ U.V^T, with the
per-iteration work done by libbnmtf_hip.so on an MI355X.

    BNMF = bnmf_gibbs_optimised(R, M, K, priors)
    BNMF.initialise(init)          # 'random' | 'exp'
    BNMF.run(iterations)           # -> (all_U, all_V, all_tau)
    BNMF.approx_expectation(burn_in, thinning); BNMF.predict(M_pred, burn_in, thinning)
    BNMF.quality(metric, burn_in, thinning)

Same constructor arguments, attributes (R, M, I, J, K, size_Omega, alpha, beta,
lambdaU, lambdaV, U, V, tau, all_U, all_V, all_tau, all_times, all_performances),
methods and assertion messages as the reference.  Build-only extras are
keyword-only: seed (Philox key; default drawn from numpy.random so that
numpy.random.seed() makes runs reproducible, as in the reference), device,
verbose, rank/world/comm_id (row/column sharding over several GPUs).
"""
import ctypes as C
import math

import numpy as np

from . import _lib
from ._base import DeviceModel, broadcast_lambda, check_rank, check_R_M, metrics_from_sums
from ._blocked import BLOCK, MAX_BLOCKS, ColumnBlocks

MAX_RANK_BLOCKED = BLOCK * MAX_BLOCKS      # 256: ranks above 64 run as column blocks (_blocked.py)


class bnmf_gibbs_optimised(DeviceModel):
    def __init__(self, R, M, K, priors, *, seed=None, device=0, verbose=True, rank=0, world=1, comm_id=None):
        self.R = np.array(R, dtype=float)
        self.M = np.array(M, dtype=float)
        self.K = K
        check_R_M(self.R, self.M)
        check_rank("bnmf_gibbs_optimised", MAX_RANK_BLOCKED, K=self.K)
        (self.I, self.J) = self.R.shape
        self.size_Omega = self.M.sum()
        self.alpha, self.beta = float(priors['alpha']), float(priors['beta'])
        self.lambdaU = broadcast_lambda(priors['lambdaU'], (self.I, self.K), "lambdaU")
        self.lambdaV = broadcast_lambda(priors['lambdaV'], (self.J, self.K), "lambdaV")
        self.verbose = verbose
        self._init_device(seed, device, rank, world, comm_id)
        # ranks above 64 (the reference has no limit, :54-78): column blocks of at most 64, one device model each (_blocked.py)
        self._blocks = None
        if self.K > BLOCK:
            assert world == 1, "ranks above %d run on one GPU (column blocks: DESIGN.md section 8)" % BLOCK
            self._blocks = ColumnBlocks(self, bnmf_gibbs_optimised)

    def _lambda_arrays(self):
        return self.lambdaU, self.lambdaV, None

    def close(self):
        if getattr(self, "_blocks", None) is not None:
            self._blocks.close()
        super(bnmf_gibbs_optimised, self).close()

    def _handle(self):
        if getattr(self, "_blocks", None) is not None:       # shape-only entry points (omega_counts, ...): the first block's handle
            return self._blocks.handles()[0]
        return super(bnmf_gibbs_optimised, self)._handle()

    # Initialise and run the sampler (bnmf_gibbs_optimised.py:94-96)
    def train(self, init, iterations):
        self.initialise(init=init)
        return self.run(iterations)

    def initialise(self, init='random'):
        """:100-117.  'random' consumes numpy.random.exponential in (i,k) row-major order,
        i.e. the same values as the reference's scalar loop for the same numpy seed."""
        assert init in ['random', 'exp'], "Unknown initialisation option: %s. Should be 'random' or 'exp'." % init
        if init == 'random':
            self.U = self._rng().exponential(scale=1.0 / self.lambdaU)
            self.V = self._rng().exponential(scale=1.0 / self.lambdaV)
        else:
            self.U = 1.0 / self.lambdaU
            self.V = 1.0 / self.lambdaV
        self.tau = self.alpha_s() / self.beta_s()

    def _push(self, tau=None):
        tau = getattr(self, "tau", 1.0) if tau is None else tau
        if self._blocks is not None:
            held = getattr(self, "_device_state", None)
            if held is not None and float(tau) == held[3] and np.array_equal(self.U, held[1]) and np.array_equal(self.V, held[2]):
                return
            self._blocks.push(np.asarray(self.U, dtype=float), np.asarray(self.V, dtype=float), float(tau))
            self._device_state = (None, np.array(self.U, dtype=float), np.array(self.V, dtype=float), float(tau))
            return
        # the state the device holds already (nothing touched U, V, tau since the last run() pulled them): no upload -- and the
        # device keeps what it carries between its half sweeps, so run(a); run(b) is the chain of run(a + b)
        held = getattr(self, "_device_state", None)
        if held is not None and held[0] is self._h and float(tau) == held[3] and np.array_equal(self.U, held[1]) and np.array_equal(self.V, held[2]):
            return
        self._device_state = None
        _lib.check(_lib.lib().bnmf_set_state(self._handle(), _lib.ptr(_lib.f64(self.U)), _lib.ptr(_lib.f64(self.V)), float(tau)))

    def _pull(self):
        U = np.zeros((self.I, self.K)); V = np.zeros((self.J, self.K)); tau = C.c_double()
        _lib.check(_lib.lib().bnmf_get_state(self._handle(), _lib.ptr(U), _lib.ptr(V), C.byref(tau)))
        self.U, self.V, self.tau = U, V, tau.value
        self._device_state = (self._h, U.copy(), V.copy(), tau.value)

    def run(self, iterations, update='draw', store_samples=True, expectation=None):
        """:121-157.  One device call runs all iterations; samples, tau, metrics and
        cumulative times come back afterwards.  store_samples=False skips the all_U/all_V
        hand-off (device-resident benchmark mode); update='mode' runs the ICM harness;
        expectation=(burn_in, thinning) also accumulates the posterior means of exactly that
        approx_expectation(burn_in, thinning) on the device (what the model-selection drivers
        need: with store_samples=False no sample ever crosses to the host)."""
        if self._blocks is not None:
            return self._run_blocked(iterations, _lib.UPDATE_MODE if update == 'mode' else _lib.UPDATE_DRAW, store_samples, expectation)
        bufs = self._run_prepare(iterations, store_samples, expectation)
        it, U_out, V_out, taus, perf, times = bufs
        _lib.check(_lib.lib().bnmf_gibbs_run(self._handle(), it, _lib.UPDATE_MODE if update == 'mode' else _lib.UPDATE_DRAW,
                                             _lib.ptr(U_out), _lib.ptr(V_out), _lib.ptr(taus), _lib.ptr(perf), _lib.ptr(times)))
        return self._run_finish(bufs, store_samples)

    def _run_blocked(self, iterations, update, store_samples, expectation, minimum_TN=0.0, icm=False):
        """run() of a model wider than 64 columns: the blocks' half sweeps in turn (_blocked.py); tau by the update rule's own
        law -- a Gamma(alpha_s, beta_s) draw keyed like the single-handle loop's (seed, iteration); mode updates (the deterministic
        harness) take its mean, ICM the Gamma mode (nmf_icm.py:137)."""
        from .distributions import gamma_draw
        it = int(iterations)
        self._push()
        blocks = self._blocks
        all_U = np.zeros((it, self.I, self.K), dtype=np.float32) if store_samples else None
        all_V = np.zeros((it, self.J, self.K), dtype=np.float32) if store_samples else None
        self._dev_expect = None
        acc = None
        if expectation is not None:
            burn_in, thinning = int(expectation[0]), int(expectation[1])
            assert 0 <= burn_in < it and thinning >= 1, "expectation=(burn_in, thinning) needs 0 <= burn_in < iterations, thinning >= 1"
            acc = {"U": np.zeros((self.I, self.K)), "V": np.zeros((self.J, self.K)), "tau": 0.0, "n": 0, "sel": set(range(burn_in, it, thinning))}
        alpha_s = self.alpha_s()

        def tau_rule(iteration, sse):
            beta_s = self.beta + 0.5 * sse
            if icm:
                return (alpha_s - 1.0) / beta_s
            if update == _lib.UPDATE_MODE:          # the deterministic harness: the mean, as the single-handle loop takes it (csrc finish_kernel)
                return alpha_s / beta_s
            return gamma_draw(alpha_s, beta_s, seed=self._seed, it=iteration, device=self._device)

        def store(i, U, V):
            all_U[i] = U; all_V[i] = V

        def each(i, U, V, tau):
            if i in acc["sel"]:
                acc["U"] += U; acc["V"] += V; acc["tau"] += tau; acc["n"] += 1

        taus, perf, times = blocks.run(it, update, tau_rule, minimum_TN=minimum_TN, store=store if store_samples else None,
                                       each=each if acc is not None else None)
        if it > 0:
            self.U, self.V, self.tau = blocks.last
            self._device_state = (None, self.U.copy(), self.V.copy(), float(self.tau))
        if acc is not None and acc["n"] > 0:
            self._host_expect = ((burn_in, thinning), acc["U"] / acc["n"], acc["V"] / acc["n"], acc["tau"] / acc["n"])
        self.all_U = all_U if store_samples else np.zeros((0, self.I, self.K))
        self.all_V = all_V if store_samples else np.zeros((0, self.J, self.K))
        self.all_tau = taus
        self.all_times = list(times)
        self.all_performances = {'MSE': list(perf[:, 0]), 'R^2': list(perf[:, 1]), 'Rp': list(perf[:, 2])}
        if self.verbose:
            for i in range(it):
                print("Iteration %s. MSE: %s. R^2: %s. Rp: %s." % (i + 1, perf[i, 0], perf[i, 1], perf[i, 2]))
        return (self.all_U, self.all_V, self.all_tau)

    def _device_expectation(self, burn_in, thinning):
        if self._blocks is not None:           # (the blocked run keeps the posterior sums on the host)
            he = getattr(self, "_host_expect", None)
            if he is not None and he[0] == (int(burn_in), int(thinning)) and len(getattr(self, "all_U", ())) == 0:
                return (he[1], None, he[2], he[3])
            return None
        return super(bnmf_gibbs_optimised, self)._device_expectation(burn_in, thinning)

    def _metric_sums(self, M_pred, A, S, B):
        if self._blocks is not None:
            if M_pred is not None:
                Mp_ = np.asarray(M_pred)
                assert ((Mp_ == 0) | (Mp_ == 1)).all(), "The indicator matrix M_pred must contain only 0 and 1."
            return self._blocks.metric_sums(M_pred, self.U if A is None else A, self.V if B is None else B)
        return super(bnmf_gibbs_optimised, self)._metric_sums(M_pred, A, S, B)

    def describe(self):
        if self._blocks is not None:
            self._blocks._prepare()
            return "column blocks %s: " % (self._blocks.ranges,) + " | ".join(ch.describe() for ch in self._blocks.children)
        return super(bnmf_gibbs_optimised, self).describe()

    def _run_prepare(self, iterations, store_samples, expectation):
        """State on the device, expectation switch, output arrays of one run() call (also used by bnmtf_amd.run_many)."""
        it = int(iterations)
        self._push()
        self._set_expectation(expectation, it)
        # page-locked sample arrays: the device-to-host copy of iteration t overlaps the sweeps of iteration t+1
        U_out = _lib.sample_buffer((it, self.I, self.K)) if store_samples else None
        V_out = _lib.sample_buffer((it, self.J, self.K)) if store_samples else None
        return (it, U_out, V_out, np.zeros(it), np.zeros((it, 3)), np.zeros(it))

    def _run_finish(self, bufs, store_samples, state=None):
        it, U_out, V_out, taus, perf, times = bufs
        if state is None:
            self._pull()
        else:                       # (run_many fetched the final states of the whole batch with one synchronisation)
            self.U, self.V, self.tau = state[0], state[1], float(state[2][0])
            self._device_state = (self._h, self.U.copy(), self.V.copy(), self.tau)
        # the samples are what the device drew: fp32 (the reference's arrays are fp64; every reduction below sums in fp64)
        self.all_U = U_out if store_samples else np.zeros((0, self.I, self.K))
        self.all_V = V_out if store_samples else np.zeros((0, self.J, self.K))
        self.all_tau = taus
        self.all_times = list(times)
        self.all_performances = {'MSE': list(perf[:, 0]), 'R^2': list(perf[:, 1]), 'Rp': list(perf[:, 2])}
        if self.verbose:
            for i in range(it):
                print("Iteration %s. MSE: %s. R^2: %s. Rp: %s." % (i + 1, perf[i, 0], perf[i, 1], perf[i, 2]))
        return (self.all_U, self.all_V, self.all_tau)

    # Parameters of the conditional posteriors (:161-177), evaluated on the device
    def alpha_s(self):
        return self.alpha + self.size_Omega / 2.0

    def beta_s(self):
        if self._blocks is not None:           # :164-165 from the full-width masked SSE
            s = self._metric_sums(None, np.asarray(self.U, dtype=float), None, np.asarray(self.V, dtype=float))
            return self.beta + 0.5 * (s[2] - 2.0 * s[5] + s[4])
        self._push()
        out = C.c_double()
        _lib.check(_lib.lib().bnmtf_beta_s(self._handle(), C.byref(out)))
        return out.value

    def _cond(self, which, k):
        self._push()
        if self._blocks is not None:
            return self._blocks.cond(which, k)
        n = self.I if which == 0 else self.J
        numer = np.zeros(n); tauk = np.zeros(n)
        _lib.check(_lib.lib().bnmf_cond_params(self._handle(), which, int(k), _lib.ptr(numer), _lib.ptr(tauk)))
        return numer, tauk

    def tauU(self, k):
        return self._cond(0, k)[1]

    def muU(self, tauUk, k):
        return 1. / np.asarray(tauUk, dtype=float) * self._cond(0, k)[0]

    def tauV(self, k):
        return self._cond(1, k)[1]

    def muV(self, tauVk, k):
        return 1. / np.asarray(tauVk, dtype=float) * self._cond(1, k)[0]

    # Posterior means from the stored samples (:182-187); host fp64, accepts lists
    def approx_expectation(self, burn_in, thinning):
        dev = self._device_expectation(burn_in, thinning)
        if dev is not None:
            return (dev[0], dev[2], dev[3])
        indices = range(burn_in, len(self.all_U), thinning)
        exp_U = np.array([self.all_U[i] for i in indices], dtype=np.float64).sum(axis=0) / float(len(indices))
        exp_V = np.array([self.all_V[i] for i in indices], dtype=np.float64).sum(axis=0) / float(len(indices))
        exp_tau = sum([self.all_tau[i] for i in indices]) / float(len(indices))
        return (exp_U, exp_V, exp_tau)

    def predict(self, M_pred, burn_in, thinning):
        """:191-197 (the I x J x K product and the masked sums run on the device in fp64)."""
        (exp_U, exp_V, _) = self.approx_expectation(burn_in, thinning)
        return metrics_from_sums(self._metric_sums(M_pred, exp_U, None, exp_V))

    def predict_while_running(self):
        """:199-204."""
        return metrics_from_sums(self._metric_sums(None, self.U, None, self.V))

    def quality(self, metric, burn_in, thinning):
        """:227-245."""
        assert metric in ['loglikelihood', 'BIC', 'AIC', 'MSE', 'ELBO'], 'Unrecognised metric for model quality: %s.' % metric
        (expU, expV, exptau) = self.approx_expectation(burn_in, thinning)
        log_likelihood = self.log_likelihood(expU, expV, exptau)
        if metric == 'loglikelihood':
            return log_likelihood
        elif metric == 'BIC':
            return - 2 * log_likelihood + (self.I * self.K + self.J * self.K) * math.log(self.size_Omega)
        elif metric == 'AIC':
            return - 2 * log_likelihood + 2 * (self.I * self.K + self.J * self.K)
        elif metric == 'MSE':
            return metrics_from_sums(self._metric_sums(None, expU, None, expV))['MSE']
        elif metric == 'ELBO':
            return 0.

    def log_likelihood(self, expU, expV, exptau):
        """:247-251."""
        s = self._metric_sums(None, expU, None, expV)
        sse = s[2] - 2.0 * s[5] + s[4]
        explogtau = math.log(exptau)
        return self.size_Omega / 2. * (explogtau - math.log(2 * math.pi)) - exptau / 2. * sse


bnmf_gibbs = bnmf_gibbs_optimised
