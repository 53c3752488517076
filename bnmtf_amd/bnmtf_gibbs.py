"""Drop-in for code/models/bnmtf_gibbs_optimised.py (class bnmtf_gibbs_optimised): Gibbs sampler
for Bayesian non-negative matrix tri-factorisation R ~ F.S.G^T on an MI355X.

    BNMTF = bnmtf_gibbs_optimised(R, M, K, L, priors)
    BNMTF.initialise(init_S, init_FG)      # init_S: 'random'|'exp'; init_FG: 'random'|'exp'|'kmeans'
    BNMTF.run(iterations)                  # -> (all_F, all_S, all_G, all_tau)
"""
import ctypes as C
import math

import numpy as np

from . import _lib
from ._base import DeviceModel, broadcast_lambda, check_rank, check_R_M, metrics_from_sums
from .kmeans import KMeans
from ._blocked import BLOCK, MAX_BLOCKS, TriBlocks

MAX_RANK_BLOCKED = BLOCK * MAX_BLOCKS      # 256: K or L above 64 run as blocks of S (_blocked.py: TriBlocks)


class bnmtf_gibbs_optimised(DeviceModel):
    def __init__(self, R, M, K, L, priors, *, seed=None, device=0, verbose=True, rank=0, world=1, comm_id=None):
        self.R = np.array(R, dtype=float)
        self.M = np.array(M, dtype=float)
        self.K = K
        self.L = L
        check_R_M(self.R, self.M)
        check_rank("bnmtf_gibbs_optimised", MAX_RANK_BLOCKED, K=self.K, L=self.L)
        (self.I, self.J) = self.R.shape
        self.size_Omega = self.M.sum()
        self.alpha, self.beta = float(priors['alpha']), float(priors['beta'])
        self.lambdaF = broadcast_lambda(priors['lambdaF'], (self.I, self.K), "lambdaF")
        self.lambdaS = broadcast_lambda(priors['lambdaS'], (self.K, self.L), "lambdaS")
        self.lambdaG = broadcast_lambda(priors['lambdaG'], (self.J, self.L), "lambdaG")
        self.verbose = verbose
        self._init_device(seed, device, rank, world, comm_id)
        # K or L above 64 (the reference has no limit, :56-84): F's and G's column blocks and the blocks of S, one device model each
        self._blocks = None
        if self.K > BLOCK or self.L > BLOCK:
            assert world == 1, "ranks above %d run on one GPU (blocks: DESIGN.md section 8)" % BLOCK
            from .bnmf_gibbs import bnmf_gibbs_optimised
            self._blocks = TriBlocks(self, bnmf_gibbs_optimised, bnmtf_gibbs_optimised)

    def _lambda_arrays(self):
        return self.lambdaF, self.lambdaG, self.lambdaS

    def close(self):
        if getattr(self, "_blocks", None) is not None:
            self._blocks.close()
        super(bnmtf_gibbs_optimised, self).close()

    def _handle(self):
        if getattr(self, "_blocks", None) is not None:       # shape-only entry points (omega_counts, ...): the first F block's handle
            self._blocks._prepare()
            return self._blocks.Fch[0]._handle()
        return super(bnmtf_gibbs_optimised, self)._handle()

    def describe(self):
        if self._blocks is not None:
            self._blocks._prepare()
            return "blocks of F %s, of G %s, of S their product: " % (self._blocks.kr, self._blocks.lr) + " | ".join(ch.describe() for ch in self._blocks.children())
        return super(bnmtf_gibbs_optimised, self).describe()

    def train(self, init, iterations):
        """bnmtf_gibbs_optimised.py:100-102 (as written there: initialise(init=init) is not a valid
        keyword of initialise and raises TypeError in the reference too)."""
        self.initialise(init=init)
        return self.run(iterations)

    def initialise(self, init_S='random', init_FG='random'):
        """:106-134."""
        assert init_S in ['random', 'exp'], "Unknown initialisation option for S: %s. Should be 'random' or 'exp'." % init_S
        assert init_FG in ['random', 'exp', 'kmeans'], "Unknown initialisation option for S: %s. Should be 'random', 'exp', or 'kmeans." % init_FG
        self.S = 1. / self.lambdaS
        if init_S == 'random':
            self.S = self._rng().exponential(scale=1.0 / self.lambdaS)
        self.F, self.G = 1. / self.lambdaF, 1. / self.lambdaG
        if init_FG == 'random':
            self.F = self._rng().exponential(scale=1.0 / self.lambdaF)
            self.G = self._rng().exponential(scale=1.0 / self.lambdaG)
        elif init_FG == 'kmeans':
            if self.verbose: print("Initialising F using KMeans.")
            kmeans_F = KMeans(self.R, self.M, self.K, device=self._device)
            kmeans_F.initialise()
            kmeans_F.cluster()
            self.F = kmeans_F.clustering_results + 0.2
            if self.verbose: print("Initialising G using KMeans.")
            kmeans_G = KMeans(self.R.T, self.M.T, self.L, device=self._device)
            kmeans_G.initialise()
            kmeans_G.cluster()
            self.G = kmeans_G.clustering_results + 0.2
        self.tau = self.alpha_s() / self.beta_s()

    def _push(self):
        if self._blocks is not None:           # (a blocked model's state lives on the host between the phases of its iterations)
            return
        tau = float(getattr(self, "tau", 1.0))
        # the state the device holds already (nothing touched F, S, G, tau since the last run() pulled them): no upload -- and the
        # device keeps what it carries between its half sweeps, so run(a); run(b) is the chain of run(a + b)
        held = getattr(self, "_device_state", None)
        if (held is not None and held[0] is self._h and tau == held[4] and np.array_equal(self.F, held[1])
                and np.array_equal(self.S, held[2]) and np.array_equal(self.G, held[3])):
            return
        self._device_state = None
        _lib.check(_lib.lib().bnmtf_set_state(self._handle(), _lib.ptr(_lib.f64(self.F)), _lib.ptr(_lib.f64(self.S)),
                                              _lib.ptr(_lib.f64(self.G)), tau))

    def _pull(self):
        F = np.zeros((self.I, self.K)); S = np.zeros((self.K, self.L)); G = np.zeros((self.J, self.L)); tau = C.c_double()
        _lib.check(_lib.lib().bnmtf_get_state(self._handle(), _lib.ptr(F), _lib.ptr(S), _lib.ptr(G), C.byref(tau)))
        self.F, self.S, self.G, self.tau = F, S, G, tau.value
        self._device_state = (self._h, F.copy(), S.copy(), G.copy(), tau.value)

    def run(self, iterations, update='draw', store_samples=True, expectation=None):
        """:138-180.  expectation=(burn_in, thinning): posterior means accumulated on the device (see bnmf_gibbs_optimised.run)."""
        if self._blocks is not None:
            return self._run_blocked(iterations, _lib.UPDATE_MODE if update == 'mode' else _lib.UPDATE_DRAW, store_samples, expectation)
        bufs = self._run_prepare(iterations, store_samples, expectation)
        it, F_out, S_out, G_out, taus, perf, times = bufs
        _lib.check(_lib.lib().bnmtf_gibbs_run(self._handle(), it, _lib.UPDATE_MODE if update == 'mode' else _lib.UPDATE_DRAW,
                                              _lib.ptr(F_out), _lib.ptr(S_out), _lib.ptr(G_out), _lib.ptr(taus), _lib.ptr(perf), _lib.ptr(times)))
        return self._run_finish(bufs, store_samples)

    def _run_blocked(self, iterations, update, store_samples, expectation, minimum_TN=0.0, icm=False):
        """run() of a model with K or L above 64 (_blocked.py: TriBlocks): tau by the update rule's own law -- a Gamma(alpha_s,
        beta_s) draw keyed like the single-handle loop's (seed, iteration); mode updates take its mean, ICM the Gamma mode (nmtf_icm.py:160)."""
        from .distributions import gamma_draw
        it = int(iterations)
        blocks = self._blocks
        all_F = np.zeros((it, self.I, self.K), dtype=np.float32) if store_samples else None
        all_S = np.zeros((it, self.K, self.L), dtype=np.float32) if store_samples else None
        all_G = np.zeros((it, self.J, self.L), dtype=np.float32) if store_samples else None
        acc = None
        if expectation is not None:
            burn_in, thinning = int(expectation[0]), int(expectation[1])
            assert 0 <= burn_in < it and thinning >= 1, "expectation=(burn_in, thinning) needs 0 <= burn_in < iterations, thinning >= 1"
            acc = {"F": np.zeros((self.I, self.K)), "S": np.zeros((self.K, self.L)), "G": np.zeros((self.J, self.L)), "tau": 0.0, "n": 0, "sel": set(range(burn_in, it, thinning))}
        alpha_s = self.alpha_s()
        blocks._prepare()

        def tau_rule(iteration, sse):
            beta_s = self.beta + 0.5 * sse
            if icm:
                return (alpha_s - 1.0) / beta_s
            if update == _lib.UPDATE_MODE:          # the deterministic harness: the mean, as the single-handle loop takes it (csrc finish_kernel)
                return alpha_s / beta_s
            return gamma_draw(alpha_s, beta_s, seed=self._seed, it=iteration, device=self._device)

        def store(i, F, S, G):
            all_F[i] = F; all_S[i] = S; all_G[i] = G

        def each(i, F, S, G, tau):
            if i in acc["sel"]:
                acc["F"] += F; acc["S"] += S; acc["G"] += G; acc["tau"] += tau; acc["n"] += 1

        taus, perf, times = blocks.run(it, update, tau_rule, minimum_TN=minimum_TN, store=store if store_samples else None,
                                       each=each if acc is not None else None)
        if it > 0:
            self.F, self.S, self.G, self.tau = blocks.last
        self._host_expect = None
        if acc is not None and acc["n"] > 0:
            self._host_expect = ((burn_in, thinning), acc["F"] / acc["n"], acc["S"] / acc["n"], acc["G"] / acc["n"], acc["tau"] / acc["n"])
        self.all_F = all_F if store_samples else np.zeros((0, self.I, self.K))
        self.all_S = all_S if store_samples else np.zeros((0, self.K, self.L))
        self.all_G = all_G if store_samples else np.zeros((0, self.J, self.L))
        self.all_tau = taus
        self.all_times = list(times)
        self.all_performances = {'MSE': list(perf[:, 0]), 'R^2': list(perf[:, 1]), 'Rp': list(perf[:, 2])}
        if self.verbose:
            for i in range(it):
                print("Iteration %s. MSE: %s. R^2: %s. Rp: %s." % (i + 1, perf[i, 0], perf[i, 1], perf[i, 2]))
        return (self.all_F, self.all_S, self.all_G, self.all_tau)

    def _device_expectation(self, burn_in, thinning):
        if self._blocks is not None:           # (the blocked run keeps the posterior sums on the host)
            he = getattr(self, "_host_expect", None)
            if he is not None and he[0] == (int(burn_in), int(thinning)) and len(getattr(self, "all_F", ())) == 0:
                return (he[1], he[2], he[3], he[4])
            return None
        return super(bnmtf_gibbs_optimised, self)._device_expectation(burn_in, thinning)

    def _metric_sums(self, M_pred, A, S, B):
        if self._blocks is not None:
            if M_pred is not None:
                Mp_ = np.asarray(M_pred)
                assert ((Mp_ == 0) | (Mp_ == 1)).all(), "The indicator matrix M_pred must contain only 0 and 1."
            return self._blocks.metric_sums(M_pred, self.F if A is None else A, self.S if S is None else S, self.G if B is None else B)
        return super(bnmtf_gibbs_optimised, self)._metric_sums(M_pred, A, S, B)

    # run() in two halves, so that bnmtf_amd.run_many can put many models' device part into one call
    def _run_prepare(self, iterations, store_samples, expectation):
        it = int(iterations)
        self._push()
        self._set_expectation(expectation, it)
        F_out = _lib.sample_buffer((it, self.I, self.K)) if store_samples else None
        S_out = _lib.sample_buffer((it, self.K, self.L)) if store_samples else None
        G_out = _lib.sample_buffer((it, self.J, self.L)) if store_samples else None
        return (it, F_out, S_out, G_out, np.zeros(it), np.zeros((it, 3)), np.zeros(it))

    def _run_finish(self, bufs, store_samples, state=None):
        it, F_out, S_out, G_out, taus, perf, times = bufs
        if state is None:
            self._pull()
        else:                       # (run_many fetched the final states of the whole batch with one synchronisation)
            self.F, self.S, self.G, self.tau = state[0], state[1], state[2], float(state[3][0])
            self._device_state = (self._h, self.F.copy(), self.S.copy(), self.G.copy(), self.tau)
        self.all_F = F_out if store_samples else np.zeros((0, self.I, self.K))
        self.all_S = S_out if store_samples else np.zeros((0, self.K, self.L))
        self.all_G = G_out if store_samples else np.zeros((0, self.J, self.L))
        self.all_tau = taus
        self.all_times = list(times)
        self.all_performances = {'MSE': list(perf[:, 0]), 'R^2': list(perf[:, 1]), 'Rp': list(perf[:, 2])}
        if self.verbose:
            for i in range(it):
                print("Iteration %s. MSE: %s. R^2: %s. Rp: %s." % (i + 1, perf[i, 0], perf[i, 1], perf[i, 2]))
        return (self.all_F, self.all_S, self.all_G, self.all_tau)

    def triple_dot(self, M1, M2, M3):
        """:184-185 (host utility on explicit arrays)."""
        return np.dot(M1, np.dot(M2, M3))

    def alpha_s(self):
        return self.alpha + self.size_Omega / 2.0

    def beta_s(self):
        """:192-193."""
        if self._blocks is not None:           # from the full-width masked SSE
            s = self._metric_sums(None, self.F, self.S, self.G)
            return self.beta + 0.5 * (s[2] - 2.0 * s[5] + s[4])
        self._push()
        out = C.c_double()
        _lib.check(_lib.lib().bnmtf_beta_s(self._handle(), C.byref(out)))
        return out.value

    def _cond(self, which, k, l, n):
        if self._blocks is not None:
            return self._blocks.cond(which, k, l, self.F, self.S, self.G, float(getattr(self, "tau", 1.0)))
        self._push()
        numer = np.zeros(n); tauk = np.zeros(n)
        _lib.check(_lib.lib().bnmtf_cond_params(self._handle(), which, int(k), int(l), _lib.ptr(numer), _lib.ptr(tauk)))
        return numer, tauk

    def tauF(self, k):
        return self._cond(0, k, 0, self.I)[1]

    def muF(self, tauFk, k):
        return 1. / np.asarray(tauFk, dtype=float) * self._cond(0, k, 0, self.I)[0]

    def tauS(self, k, l):
        return float(self._cond(1, k, l, 1)[1][0])

    def muS(self, tauSkl, k, l):
        return 1. / tauSkl * float(self._cond(1, k, l, 1)[0][0])

    def tauG(self, l):
        return self._cond(2, 0, l, self.J)[1]

    def muG(self, tauGl, l):
        return 1. / np.asarray(tauGl, dtype=float) * self._cond(2, 0, l, self.J)[0]

    def approx_expectation(self, burn_in, thinning):
        """:216-223."""
        dev = self._device_expectation(burn_in, thinning)
        if dev is not None:
            return dev
        indices = range(burn_in, len(self.all_F), thinning)
        exp_F = np.array([self.all_F[i] for i in indices], dtype=np.float64).sum(axis=0) / float(len(indices))
        exp_S = np.array([self.all_S[i] for i in indices], dtype=np.float64).sum(axis=0) / float(len(indices))
        exp_G = np.array([self.all_G[i] for i in indices], dtype=np.float64).sum(axis=0) / float(len(indices))
        exp_tau = sum([self.all_tau[i] for i in indices]) / float(len(indices))
        return (exp_F, exp_S, exp_G, exp_tau)

    def predict(self, M_pred, burn_in, thinning):
        """:226-232."""
        (exp_F, exp_S, exp_G, _) = self.approx_expectation(burn_in, thinning)
        return metrics_from_sums(self._metric_sums(M_pred, exp_F, exp_S, exp_G))

    def predict_while_running(self):
        """:234-239."""
        return metrics_from_sums(self._metric_sums(None, self.F, self.S, self.G))

    def quality(self, metric, burn_in, thinning):
        """:262-280."""
        assert metric in ['loglikelihood', 'BIC', 'AIC', 'MSE', 'ELBO'], 'Unrecognised metric for model quality: %s.' % metric
        (expF, expS, expG, exptau) = self.approx_expectation(burn_in, thinning)
        log_likelihood = self.log_likelihood(expF, expS, expG, exptau)
        npar = self.I * self.K + self.K * self.L + self.J * self.L
        if metric == 'loglikelihood':
            return log_likelihood
        elif metric == 'BIC':
            return - 2 * log_likelihood + npar * math.log(self.size_Omega)
        elif metric == 'AIC':
            return - 2 * log_likelihood + 2 * npar
        elif metric == 'MSE':
            return metrics_from_sums(self._metric_sums(None, expF, expS, expG))['MSE']
        elif metric == 'ELBO':
            return 0.

    def log_likelihood(self, expF, expS, expG, exptau):
        """:282-285."""
        s = self._metric_sums(None, expF, expS, expG)
        sse = s[2] - 2.0 * s[5] + s[4]
        return self.size_Omega / 2. * (math.log(exptau) - math.log(2 * math.pi)) - exptau / 2. * sse


bnmtf_gibbs = bnmtf_gibbs_optimised
