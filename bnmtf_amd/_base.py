"""Shared host-side logic of the model classes: argument checks with the reference's
exact assertion messages, the device handle, metric finishing (fp64)."""
import ctypes as C
import hashlib
import math

import numpy as np

from . import _lib


def check_rank(what, limit, **ranks):
    """The reference accepts any rank; this build's kernels hold a latent factor per wave lane (K, L <= 64) and the
    variational tri-factorisation has only the dense S system (K, L <= 32).  Said at construction, not at the first device
    call of a search that has already fitted its smaller candidates."""
    from ._lib import BnmtfError
    for name, v in ranks.items():
        if not (1 <= int(v) <= limit):
            raise BnmtfError("%s: %s = %s is outside what this build runs (1 <= %s <= %d; DESIGN.md section 1, limits)" % (what, name, v, name, limit))


def check_R_M(R, M):
    """bnmf_gibbs_optimised.py:59-62 and check_empty_rows_columns :82-90 (same text in
    bnmtf_gibbs_optimised.py:62-65,88-96 and bnmf_vb_optimised.py:58-61,81-89)."""
    assert len(R.shape) == 2, "Input matrix R is not a two-dimensional array, " \
        "but instead %s-dimensional." % len(R.shape)
    assert R.shape == M.shape, "Input matrix R is not of the same size as " \
        "the indicator matrix M: %s and %s respectively." % (R.shape, M.shape)
    for i, c in enumerate(M.sum(axis=1)):
        assert c != 0, "Fully unobserved row in R, row %s." % i
    for j, c in enumerate(M.sum(axis=0)):
        assert c != 0, "Fully unobserved column in R, column %s." % j


def broadcast_lambda(value, shape, name):
    """Scalar-or-array prior rate -> full array (bnmf_gibbs_optimised.py:68-78)."""
    lam = np.array(value)
    if lam.shape == ():
        lam = lam * np.ones(shape)
    assert lam.shape == shape, "Prior matrix %s has the wrong shape: %s instead of (%s, %s)." % (
        name, lam.shape, shape[0], shape[1])
    return lam


def compute_MSE(M, R, R_pred):
    """bnmf_gibbs_optimised.py:208-209 (explicit R_pred supplied by the caller: host fp64)."""
    return (M * (R - R_pred) ** 2).sum() / float(M.sum())


def compute_R2(M, R, R_pred):
    """:211-215."""
    mean = (M * R).sum() / float(M.sum())
    SS_total = float((M * (R - mean) ** 2).sum())
    SS_res = float((M * (R - R_pred) ** 2).sum())
    return 1. - SS_res / SS_total if SS_total != 0. else np.inf


def compute_Rp(M, R, R_pred):
    """:217-223."""
    mean_real = (M * R).sum() / float(M.sum())
    mean_pred = (M * R_pred).sum() / float(M.sum())
    covariance = (M * (R - mean_real) * (R_pred - mean_pred)).sum()
    variance_real = (M * (R - mean_real) ** 2).sum()
    variance_pred = (M * (R_pred - mean_pred) ** 2).sum()
    with np.errstate(all="ignore"):
        return covariance / float(math.sqrt(variance_real) * math.sqrt(variance_pred))


def metrics_from_sums(s):
    """MSE / R^2 / Rp from the six masked sums the device returns
    (n, sum R, sum R^2, sum P, sum P^2, sum R*P); same quantities as :208-223."""
    n, sr, srr, sp, spp, srp = [float(v) for v in s]
    sse = srr - 2.0 * srp + spp
    ss_tot = srr - sr * sr / n
    cov = srp - sr * sp / n
    vp = spp - sp * sp / n
    with np.errstate(all="ignore"):
        rp = np.float64(cov) / np.float64(math.sqrt(max(ss_tot, 0.0)) * math.sqrt(max(vp, 0.0)))
    return {"MSE": sse / n, "R^2": (1.0 - sse / ss_tot) if ss_tot != 0.0 else np.inf, "Rp": float(rp)}


class DeviceModel(object):
    """Owns the bnmtf_handle of one model instance (created lazily, after any fork)."""

    def _init_device(self, seed, device, rank, world, comm_id):
        self._h = None
        if world > 1 and seed is None:
            # every rank must draw the same chain (same Philox key, same tau variates) and start from the same replicated
            # factors: without an explicit seed the key comes from the communicator id, which all ranks share
            assert comm_id is not None, "world > 1 needs the comm_id all ranks share"
            seed = int.from_bytes(hashlib.sha256(bytes(comm_id)).digest()[:8], "little") >> 2
        self._seed = seed
        self._device = device
        self._rank, self._world, self._comm_id = rank, world, comm_id
        self._init_rs = None

    def _rng(self):
        """Source of the initialisation draws: numpy.random's global stream for a single-GPU model (what the reference
        consumes, so numpy.random.seed() reproduces its initial factors); a RandomState seeded with the shared key for a
        sharded model, so that every rank holds the same replicated U, V."""
        if self._world == 1:
            return np.random
        if self._init_rs is None:
            self._init_rs = np.random.RandomState(self._seed % (2 ** 32))
        return self._init_rs

    def _lambda_arrays(self):
        raise NotImplementedError

    def _handle(self):
        if self._h is None:
            L = _lib.lib()
            if self._seed is None:      # follow NumPy's global seeding like the reference's samplers do
                self._seed = int(np.random.randint(0, 2 ** 62))
            lr, lc, ls = self._lambda_arrays()
            # the device holds a 0/1 mask; the reference would weight by the values of M (all its callers pass 0/1)
            Mb = np.ascontiguousarray(self.M != 0)              # one byte per entry already: handed over as uint8 without a copy
            assert (self.M == Mb).all(), "The indicator matrix M must contain only 0 and 1."
            self._keep = (np.ascontiguousarray(self.R, dtype=np.float32),
                          Mb.view(np.uint8),
                          _lib.f64(lr), _lib.f64(lc), None if ls is None else _lib.f64(ls),
                          None if self._comm_id is None else np.frombuffer(bytes(self._comm_id), dtype=np.uint8).copy())
            R32, M8, lr, lc, ls, cid = self._keep
            p = _lib.Problem(self.I, self.J, self.K, getattr(self, "L", 0) if ls is not None else 0,
                             _lib.ptr(R32), _lib.ptr(M8), _lib.ptr(lr), _lib.ptr(lc), _lib.ptr(ls),
                             float(self.alpha), float(self.beta), C.c_uint64(self._seed & (2 ** 64 - 1)),
                             int(self._device), int(self._rank), int(self._world), _lib.ptr(cid))
            h = C.c_void_p()
            _lib.check(L.bnmtf_create(C.byref(p), C.byref(h)))
            self._h = h
            self._keep = None           # the library copied everything it needs
        return self._h

    def close(self):
        if getattr(self, "_h", None) is not None:
            _lib.lib().bnmtf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- posterior means on the device ------------------------------------------
    def _set_expectation(self, expectation, iterations):
        """Before a run(): switch the device-side accumulation of approx_expectation(burn_in, thinning) on or off."""
        self._dev_expect = None
        if expectation is None:
            _lib.check(_lib.lib().bnmtf_set_expectation(self._handle(), -1, 1))
            return
        burn_in, thinning = int(expectation[0]), int(expectation[1])
        assert 0 <= burn_in < iterations and thinning >= 1, "expectation=(burn_in, thinning) needs 0 <= burn_in < iterations, thinning >= 1"
        _lib.check(_lib.lib().bnmtf_set_expectation(self._handle(), burn_in, thinning))
        self._dev_expect = (burn_in, thinning)

    def _device_expectation(self, burn_in, thinning):
        """(exp_A, exp_S or None, exp_B, exp_tau) from the device sums when the last run() accumulated exactly this
        (burn_in, thinning); None otherwise (the caller then averages the stored samples)."""
        if getattr(self, "_dev_expect", None) != (int(burn_in), int(thinning)):
            return None
        L = getattr(self, "L", 0)
        A = np.zeros((self.I, self.K)); B = np.zeros((self.J, L if L else self.K)); S = np.zeros((self.K, L)) if L else None
        tau = C.c_double(); cnt = C.c_uint64()
        _lib.check(_lib.lib().bnmtf_get_expectation(self._handle(), _lib.ptr(A), _lib.ptr(S), _lib.ptr(B), C.byref(tau), C.byref(cnt)))
        return (A, S, B, tau.value)

    # -- device facts ---------------------------------------------------------
    def omega_counts(self):
        """(size_Omega, per-row, per-column observed counts) as the device holds them."""
        tot = C.c_uint64()
        row = np.zeros(self.I, dtype=np.uint32); col = np.zeros(self.J, dtype=np.uint32)
        _lib.check(_lib.lib().bnmtf_omega_counts(self._handle(), C.byref(tot), _lib.ptr(row), _lib.ptr(col)))
        return int(tot.value), row, col

    def describe(self):
        buf = C.create_string_buffer(1024)
        _lib.check(_lib.lib().bnmtf_describe(self._handle(), buf, 1024))
        return buf.value.decode()

    def set_profiling(self, enable=True, kernel=None, every=1):
        """Bracket kernel launches with HIP events (all listed kernels, or only `kernel`, and then only in every
        `every`-th iteration)."""
        code = 0 if not enable else (1 if kernel is None else 2 + int(kernel) + 32 * (max(int(every), 1) - 1))
        _lib.check(_lib.lib().bnmtf_set_profiling(self._handle(), code))

    def set_sweep_path(self, fast=True):
        """fast=False forces the generic sweep kernel (test hook; results are the same)."""
        _lib.check(_lib.lib().bnmtf_set_sweep_path(self._handle(), int(bool(fast))))

    def set_small_path(self, on='auto'):
        """The one-launch path for small models (csrc/kernel_small.hip): 'auto' (default) takes it when it is the faster way to
        run the call; True / 'always' forces it for a model that qualifies, False sends run() down the multi-launch path (test /
        A-B hooks; the chain is the same up to fp32 summation order)."""
        mode = {"auto": 1, "always": 2, True: 2, False: 0, 0: 0, 1: 1, 2: 2}[on]
        _lib.check(_lib.lib().bnmtf_set_small_path(self._handle(), int(mode)))

    def is_small(self):
        """Does run() take the one-launch path (kernel_small.hip: the whole run in one launch, one block per model)?"""
        out = C.c_int()
        _lib.check(_lib.lib().bnmtf_is_small(self._handle(), C.byref(out)))
        return bool(out.value)

    def kernel_stats(self, kernel):
        ms = C.c_double(); n = C.c_uint64()
        _lib.check(_lib.lib().bnmtf_kernel_stats(self._handle(), int(kernel), C.byref(ms), C.byref(n)))
        return ms.value, int(n.value)

    def _metric_sums(self, M_pred, A, S, B):
        out = np.zeros(6)
        if M_pred is not None:
            Mp_ = np.asarray(M_pred)
            assert ((Mp_ == 0) | (Mp_ == 1)).all(), "The indicator matrix M_pred must contain only 0 and 1."
        Mp = None if M_pred is None else np.ascontiguousarray(np.asarray(M_pred) != 0, dtype=np.uint8)
        A = None if A is None else _lib.f64(A)
        S = None if S is None else _lib.f64(S)
        B = None if B is None else _lib.f64(B)
        _lib.check(_lib.lib().bnmtf_metric_sums(self._handle(), _lib.ptr(Mp), _lib.ptr(A), _lib.ptr(S), _lib.ptr(B), _lib.ptr(out)))
        return out

    # Functions for computing MSE, R^2, Rp given an explicit prediction matrix
    def compute_MSE(self, M, R, R_pred):
        return compute_MSE(np.asarray(M), np.asarray(R), np.asarray(R_pred))

    def compute_R2(self, M, R, R_pred):
        return compute_R2(np.asarray(M), np.asarray(R), np.asarray(R_pred))

    def compute_Rp(self, M, R, R_pred):
        return compute_Rp(np.asarray(M), np.asarray(R), np.asarray(R_pred))
