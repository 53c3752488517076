"""Ranks above 64: a factorisation run as COLUMN BLOCKS (round 6; the tri-factorisations: TriBlocks at the end of the file).

The reference takes any K (code/models/bnmf_gibbs_optimised.py:54-78).  The device kernels hold one latent factor per wave lane
(K <= 64), so a wider model is cut into ceil(K / 64) blocks of columns, each an ordinary device model (one handle) of rank
K_b <= 64 on the same R, M.  Given the other blocks, the conditionals of block b's columns are those of a rank-K_b model on the
residual data R - sum_{b' != b} U_b' V_b'^T (the other blocks' part of U V^T moves to the data side of
M . (R - U V^T + U_k V_k^T), bnmf_gibbs_optimised.py:170-171, 176-177).  So one iteration of run() (:133-155) is

    for every block b in order:  residual data of b  ->  half sweep of U_b          (:134-137, columns in order)
    for every block b in order:  residual data of b  ->  half sweep of V_b          (:139-142)
    tau ~ Gamma(alpha_s, beta_s)  with beta_s from the full-width masked SSE         (:144, :161-165)
    the three metrics of the full-width sample on the training mask                  (:147-150)

-- the reference's sequential column order exactly, with the draws keyed by the wide model's column index (the Philox column
word of block b's column k is 64 b + k: csrc bnmf_set_column_block), so the chain is the oracle's chain for the same seed.
The blocks' half sweeps, residual updates and the full-width metric sums are device calls (include/bnmtf_hip.h: bnmf_half_sweep,
bnmf_set_residual_data, bnmtf_metric_sums_wide); this file only orders them.  One GPU; Gibbs draws, mode and ICM updates.
Not a fast path: every half sweep of a block is preceded by a pass over R (the residual), and the per-iteration control is
Python -- it exists so that a search over value lists that reach past 64 runs instead of failing."""
import ctypes as C
import time

import numpy as np

from . import _lib

BLOCK = 64
MAX_BLOCKS = 4          # csrc kernels.h kMaxOtherBlocks + 1: ranks up to 256


def block_ranges(K):
    return [(c0, min(c0 + BLOCK, K)) for c0 in range(0, K, BLOCK)]


class ColumnBlocks(object):
    """The child models of a wide bnmf_gibbs / nmf_icm instance `owner` (which keeps the full-width U, V, tau attributes)."""

    def __init__(self, owner, child_cls, seeded=True):
        self.owner = owner
        self.ranges = block_ranges(owner.K)
        assert len(self.ranges) <= MAX_BLOCKS
        self.children = []
        for (c0, c1) in self.ranges:
            pri = {"alpha": owner.alpha, "beta": owner.beta, "lambdaU": owner.lambdaU[:, c0:c1], "lambdaV": owner.lambdaV[:, c0:c1]}
            kw = {"seed": owner._seed} if seeded else {}
            ch = child_cls(owner.R, owner.M, c1 - c0, pri, device=owner._device, verbose=False, **kw)
            self.children.append(ch)
        self._ready = False
        self.iteration = 0

    # -- device plumbing ----------------------------------------------------------
    def _prepare(self):
        if self._ready:
            return
        L = _lib.lib()
        if self.owner._seed is None:                      # one Philox key for all blocks: drawn once, like a single handle's
            self.owner._seed = int(np.random.randint(0, 2 ** 62))
        for ch, (c0, _) in zip(self.children, self.ranges):
            ch._seed = self.owner._seed
            _lib.check(L.bnmf_set_column_block(ch._handle(), int(c0)))
        self._ready = True

    def handles(self):
        self._prepare()
        return [ch._handle() for ch in self.children]

    def push(self, U, V, tau):
        self._prepare()
        for ch, (c0, c1) in zip(self.children, self.ranges):
            ch.U, ch.V, ch.tau = np.ascontiguousarray(U[:, c0:c1]), np.ascontiguousarray(V[:, c0:c1]), float(tau)
            ch._device_state = None
            ch._push()

    def pull(self):
        I, J, K = self.owner.I, self.owner.J, self.owner.K
        U = np.zeros((I, K)); V = np.zeros((J, K))
        for ch, (c0, c1) in zip(self.children, self.ranges):
            ch._pull()
            U[:, c0:c1] = ch.U; V[:, c0:c1] = ch.V
        return U, V

    def _others(self, b):
        hs = self.handles()
        o = [hs[i] for i in range(len(hs)) if i != b]
        arr = (C.c_void_p * max(len(o), 1))(*[h.value for h in o])
        return arr, len(o)

    def residual(self, b):
        arr, n = self._others(b)
        _lib.check(_lib.lib().bnmf_set_residual_data(self.handles()[b], arr, n))

    def set_tau(self, tau):
        for ch in self.children:
            # (factors stay where they are: only the scalar changes)
            _lib.check(_lib.lib().bnmtf_set_tau(ch._handle(), float(tau)))
            ch.tau = float(tau)

    def set_iteration(self, it):
        for ch in self.children:
            _lib.check(_lib.lib().bnmtf_set_iteration(ch._handle(), C.c_uint64(int(it))))

    def metric_sums(self, M_pred, A, B):
        out = np.zeros(6)
        Mp = None if M_pred is None else np.ascontiguousarray(np.asarray(M_pred) != 0, dtype=np.uint8)
        A = _lib.f64(A); B = _lib.f64(B)
        _lib.check(_lib.lib().bnmtf_metric_sums_wide(self.handles()[0], _lib.ptr(Mp), _lib.ptr(A), _lib.ptr(B), int(A.shape[1]), _lib.ptr(out)))
        return out

    def cond(self, which, k):
        """(numerator, tau_k) of the conditional of column k of U (which = 0) or V: block b's hook on its residual data."""
        b = int(k) // BLOCK
        self.residual(b)
        n = self.owner.I if which == 0 else self.owner.J
        numer = np.zeros(n); tauk = np.zeros(n)
        _lib.check(_lib.lib().bnmf_cond_params(self.handles()[b], which, int(k) - BLOCK * b, _lib.ptr(numer), _lib.ptr(tauk)))
        return numer, tauk

    # -- the iteration --------------------------------------------------------------
    def run(self, iterations, update, tau_rule, minimum_TN=0.0, store=None, each=None):
        """`iterations` iterations from the state that was pushed.  update: _lib.UPDATE_DRAW / MODE / ICM; tau_rule(it, sse) ->
        the new tau; store(it, U, V): sample hand-off (or None); each(it, U, V, tau): anything else per iteration (posterior sums).
        Returns (taus, perf [n][3], times)."""
        L = _lib.lib()
        hs = self.handles()
        nb = len(hs)
        taus = np.zeros(iterations); perf = np.zeros((iterations, 3)); times = np.zeros(iterations)
        for ch in self.children:
            _lib.check(L.bnmtf_set_minimum_tn(ch._handle(), float(minimum_TN)))
        t0 = time.time()
        from ._base import metrics_from_sums
        for it in range(iterations):
            self.set_iteration(self.iteration)
            for which in (0, 1):
                for b in range(nb):
                    if nb > 1:
                        self.residual(b)
                    _lib.check(L.bnmf_half_sweep(hs[b], which, int(update)))
            U, V = self.pull()
            s = self.metric_sums(None, U, V)
            sse = s[2] - 2.0 * s[5] + s[4]
            tau = tau_rule(self.iteration, sse)
            self.set_tau(tau)
            m = metrics_from_sums(s)
            taus[it] = tau; perf[it] = (m["MSE"], m["R^2"], m["Rp"]); times[it] = time.time() - t0
            if store is not None:
                store(it, U, V)
            if each is not None:
                each(it, U, V, tau)
            self.iteration += 1
        self.last = (U, V, float(tau)) if iterations > 0 else None
        return taus, perf, times

    def close(self):
        for ch in self.children:
            ch.close()


class VBColumnBlocks(ColumnBlocks):
    """The same for bnmf_vb_optimised (bnmf_vb_optimised.py:121-153): update_U(k) / update_exp_U(k) of block b's columns are the
    single-block updates on the residual data R - sum_{b' != b} E[U_b'] E[V_b']^T (the other blocks' part of
    M . (R - E[U] E[V]^T + ...), :189-195, moves to the data side; tauU of a column involves that column alone);
    exp_square_diff (:185-187) = the full-width masked SSE + the blocks' second-moment sums (a sum over columns)."""
    NAMES = ("muU", "tauU", "expU", "varU", "muV", "tauV", "expV", "varV")

    def __init__(self, owner, child_cls):
        ColumnBlocks.__init__(self, owner, child_cls, seeded=False)

    def _prepare(self):
        if self._ready:
            return
        for ch, (c0, _) in zip(self.children, self.ranges):
            _lib.check(_lib.lib().bnmf_set_column_block(ch._handle(), int(c0)))
        self._ready = True

    def push(self, exptau):
        self._prepare()
        o = self.owner
        for ch, (c0, c1) in zip(self.children, self.ranges):
            for n in self.NAMES:
                setattr(ch, n, np.ascontiguousarray(np.asarray(getattr(o, n), dtype=float)[:, c0:c1]))
            ch.exptau = float(exptau)
            ch._device_state = None
            ch._push()

    def pull(self):
        o = self.owner
        out = {n: np.zeros((o.I if n.endswith("U") else o.J, o.K)) for n in self.NAMES}
        for ch, (c0, c1) in zip(self.children, self.ranges):
            ch._pull()
            for n in self.NAMES:
                out[n][:, c0:c1] = getattr(ch, n)
        return out

    def esd(self, expU, expV):
        """exp_square_diff of the full-width model for the state the children hold."""
        s = self.metric_sums(None, expU, expV)
        corr = 0.0
        t = np.zeros(2)
        for h in self.handles():
            _lib.check(_lib.lib().bnmf_vb_esd_terms(h, _lib.ptr(t)))
            corr += t[1]
        return (s[2] - 2.0 * s[5] + s[4]) + corr, s

    def update(self, which, k, moments):
        b = int(k) // BLOCK
        if len(self.children) > 1:
            self.residual(b)
        _lib.check(_lib.lib().bnmf_vb_update(self.handles()[b], int(which), int(k) - BLOCK * b, int(moments)))

    def sweep_both(self):
        L = _lib.lib()
        hs = self.handles()
        for which in (0, 1):
            for b in range(len(hs)):
                if len(hs) > 1:
                    self.residual(b)
                _lib.check(L.bnmf_vb_half_sweep(hs[b], which))


class TriBlocks(object):
    """The child models of a wide bnmtf_gibbs / nmtf_icm instance `owner` (K or L above 64; the reference takes any,
    bnmtf_gibbs_optimised.py:56-84).  With K cut into row blocks b of S and L into column blocks c:

      * F's columns of block b (:146-155, 195-199): a BNMF model (U = F_b, V = G S_b^T) on the data minus what F's other blocks
        explain -- the F children, residual data from each other (bnmf_set_residual_data), draws keyed by the wide column index;
      * G's columns of block c (:162-171, 207-211): a BNMF model (U = F S_c, V = G_c) likewise -- the G children;
      * block (b, c) of S (:157-160, 201-205): a BNMTF model (F_b, S_bc, G_c) on R - F S G^T + F_b S_bc G_c^T
        = R - sum_c' W_c' G_c'^T with W_c' = (F S)_c' for c' != c and (F S)_c - F_b S_bc for c' = c: the G children carry those
        products (their U is set to W_c'), the S child takes them off its data and walks its rows (bnmtf_s_rows).  S is row-major
        over the WIDE matrix: row k visits the blocks (b, 0), (b, 1), ... in turn, so with more than one column block a step is
        one row of one block (K * ceil(L / 64) steps per iteration, a pass over R each); with one column block a step is a whole
        row block.

    Not a fast path: Python orders device calls, the effective factors are formed on the host."""

    def __init__(self, owner, bnmf_cls, bnmtf_cls):
        self.owner = o = owner
        self.kr, self.lr = block_ranges(o.K), block_ranges(o.L)
        assert len(self.kr) <= MAX_BLOCKS and len(self.lr) <= MAX_BLOCKS
        kw = dict(device=o._device, verbose=False, seed=o._seed)
        ab = {"alpha": o.alpha, "beta": o.beta}
        self.Fch = [bnmf_cls(o.R, o.M, k1 - k0, dict(ab, lambdaU=o.lambdaF[:, k0:k1], lambdaV=np.ones((o.J, k1 - k0))), **kw) for (k0, k1) in self.kr]
        self.Gch = [bnmf_cls(o.R, o.M, l1 - l0, dict(ab, lambdaU=np.ones((o.I, l1 - l0)), lambdaV=o.lambdaG[:, l0:l1]), **kw) for (l0, l1) in self.lr]
        self.Sch = [[bnmtf_cls(o.R, o.M, k1 - k0, l1 - l0, dict(ab, lambdaF=o.lambdaF[:, k0:k1], lambdaS=o.lambdaS[k0:k1, l0:l1], lambdaG=o.lambdaG[:, l0:l1]), **kw)
                     for (l0, l1) in self.lr] for (k0, k1) in self.kr]
        self._ready = False
        self.iteration = 0

    def children(self):
        return self.Fch + self.Gch + [ch for row in self.Sch for ch in row]

    def _prepare(self):
        if self._ready:
            return
        L = _lib.lib()
        o = self.owner
        if o._seed is None:
            o._seed = int(np.random.randint(0, 2 ** 62))
        for ch in self.children():
            ch._seed = o._seed
        for ch, (k0, _) in zip(self.Fch, self.kr):
            _lib.check(L.bnmf_set_column_block(ch._handle(), int(k0)))
        for ch, (l0, _) in zip(self.Gch, self.lr):
            _lib.check(L.bnmf_set_column_block(ch._handle(), int(l0)))
        for row, (k0, _) in zip(self.Sch, self.kr):
            for ch, (l0, _) in zip(row, self.lr):
                _lib.check(L.bnmtf_set_s_block(ch._handle(), int(k0), int(l0), int(o.L)))
        self._ready = True

    # -- plumbing -------------------------------------------------------------------
    @staticmethod
    def _set2(ch, U, V, tau):
        ch.U, ch.V, ch.tau = np.ascontiguousarray(U, dtype=float), np.ascontiguousarray(V, dtype=float), float(tau)
        ch._device_state = None
        ch._push()

    @staticmethod
    def _set3(ch, F, S, G, tau):
        ch.F, ch.S, ch.G, ch.tau = np.ascontiguousarray(F, dtype=float), np.ascontiguousarray(S, dtype=float), np.ascontiguousarray(G, dtype=float), float(tau)
        ch._device_state = None
        ch._push()

    @staticmethod
    def _residual(target, others):
        arr = (C.c_void_p * max(len(others), 1))(*[h._handle().value for h in others])
        _lib.check(_lib.lib().bnmf_set_residual_data(target._handle(), arr, len(others)))

    def _for_all(self, fn, *a):
        for ch in self.children():
            _lib.check(fn(ch._handle(), *a))

    def metric_sums(self, M_pred, F, S, G):
        out = np.zeros(6)
        Mp = None if M_pred is None else np.ascontiguousarray(np.asarray(M_pred) != 0, dtype=np.uint8)
        A = _lib.f64(np.dot(np.asarray(F, dtype=float), np.asarray(S, dtype=float))); B = _lib.f64(G)
        self._prepare()
        _lib.check(_lib.lib().bnmtf_metric_sums_wide(self.Fch[0]._handle(), _lib.ptr(Mp), _lib.ptr(A), _lib.ptr(B), int(A.shape[1]), _lib.ptr(out)))
        return out

    # -- the three phases of an iteration ---------------------------------------------
    def _f_prepare(self, b, F, S, G, tau):
        k0, k1 = self.kr[b]
        self._set2(self.Fch[b], F[:, k0:k1], np.dot(G, S[k0:k1, :].T), tau)

    def _f_phase(self, F, S, G, tau, update):
        L = _lib.lib()
        for b in range(len(self.kr)):
            self._f_prepare(b, F, S, G, tau)
        for b, ch in enumerate(self.Fch):
            if len(self.Fch) > 1:
                self._residual(ch, [x for x in self.Fch if x is not ch])
            _lib.check(L.bnmf_half_sweep(ch._handle(), 0, int(update)))
        for ch, (k0, k1) in zip(self.Fch, self.kr):
            ch._pull()
            F[:, k0:k1] = ch.U

    def _carrier(self, c, U, G, tau):
        l0, l1 = self.lr[c]
        self._set2(self.Gch[c], U, G[:, l0:l1], tau)

    def _s_target(self, b, c, F, S, G, FS, tau):
        """the carriers for block (b, c) of S and the block's own state"""
        k0, k1 = self.kr[b]; l0, l1 = self.lr[c]
        self._carrier(c, FS[:, l0:l1] - np.dot(F[:, k0:k1], S[k0:k1, l0:l1]), G, tau)
        self._residual(self.Sch[b][c], self.Gch)

    def _s_phase(self, F, S, G, tau, update):
        L = _lib.lib()
        nb, nc = len(self.kr), len(self.lr)
        FS = np.dot(F, S)
        for c, (l0, l1) in enumerate(self.lr):
            self._carrier(c, FS[:, l0:l1], G, tau)
        for b, (k0, k1) in enumerate(self.kr):
            for c, (l0, l1) in enumerate(self.lr):
                self._set3(self.Sch[b][c], F[:, k0:k1], S[k0:k1, l0:l1], G[:, l0:l1], tau)
        for b, (k0, k1) in enumerate(self.kr):
            # one column block: nothing outside the row block changes while its rows are walked -- one step per row block
            steps = [(0, k1 - k0)] if nc == 1 else [(k, k + 1) for k in range(k1 - k0)]
            for (r0, r1) in steps:
                for c, (l0, l1) in enumerate(self.lr):
                    ch = self.Sch[b][c]
                    self._s_target(b, c, F, S, G, FS, tau)
                    _lib.check(L.bnmtf_s_rows(ch._handle(), int(r0), int(r1), int(update)))
                    Sb = np.zeros((k1 - k0, l1 - l0))
                    _lib.check(L.bnmtf_get_state(ch._handle(), None, _lib.ptr(Sb), None, None))
                    S[k0 + r0:k0 + r1, l0:l1] = Sb[r0:r1]
                    FS[:, l0:l1] = np.dot(F, S[:, l0:l1])
                    self._carrier(c, FS[:, l0:l1], G, tau)
        return FS

    def _g_phase(self, F, S, G, FS, tau, update):
        L = _lib.lib()
        for c, ch in enumerate(self.Gch):
            if len(self.Gch) > 1:
                self._residual(ch, [x for x in self.Gch if x is not ch])
            _lib.check(L.bnmf_half_sweep(ch._handle(), 1, int(update)))
        for ch, (l0, l1) in zip(self.Gch, self.lr):
            ch._pull()
            G[:, l0:l1] = ch.V

    def run(self, iterations, update, tau_rule, minimum_TN=0.0, store=None, each=None):
        """`iterations` iterations (:138-180) from the owner's F, S, G, tau; returns (taus, perf [n][3], times)."""
        from ._base import metrics_from_sums
        self._prepare()
        L = _lib.lib()
        o = self.owner
        F, S, G, tau = np.array(o.F, dtype=float), np.array(o.S, dtype=float), np.array(o.G, dtype=float), float(o.tau)
        taus = np.zeros(iterations); perf = np.zeros((iterations, 3)); times = np.zeros(iterations)
        self._for_all(L.bnmtf_set_minimum_tn, float(minimum_TN))
        t0 = time.time()
        for it in range(iterations):
            self._for_all(L.bnmtf_set_iteration, C.c_uint64(int(self.iteration)))
            self._f_phase(F, S, G, tau, update)
            FS = self._s_phase(F, S, G, tau, update)
            self._g_phase(F, S, G, FS, tau, update)
            s = self.metric_sums(None, F, S, G)
            sse = s[2] - 2.0 * s[5] + s[4]
            tau = float(tau_rule(self.iteration, sse))
            m = metrics_from_sums(s)
            taus[it] = tau; perf[it] = (m["MSE"], m["R^2"], m["Rp"]); times[it] = time.time() - t0
            if store is not None:
                store(it, F, S, G)
            if each is not None:
                each(it, F, S, G, tau)
            self.iteration += 1
        self.last = (F, S, G, tau)
        return taus, perf, times

    def cond(self, which, k, l, F, S, G, tau):
        """(numerator, tau_kl) of the conditional of column k of F (which = 0), entry (k, l) of S (1) or column l of G (2)."""
        self._prepare()
        L = _lib.lib()
        F, S, G = np.asarray(F, dtype=float), np.asarray(S, dtype=float), np.asarray(G, dtype=float)
        if which == 0:
            b = int(k) // BLOCK
            for bb in range(len(self.kr)):
                self._f_prepare(bb, F, S, G, tau)
            if len(self.Fch) > 1:
                self._residual(self.Fch[b], [x for x in self.Fch if x is not self.Fch[b]])
            n = self.owner.I; numer = np.zeros(n); tauk = np.zeros(n)
            _lib.check(L.bnmf_cond_params(self.Fch[b]._handle(), 0, int(k) - BLOCK * b, _lib.ptr(numer), _lib.ptr(tauk)))
            return numer, tauk
        FS = np.dot(F, S)
        for c, (l0, l1) in enumerate(self.lr):
            self._carrier(c, FS[:, l0:l1], G, tau)
        c = int(l) // BLOCK
        if which == 2:
            if len(self.Gch) > 1:
                self._residual(self.Gch[c], [x for x in self.Gch if x is not self.Gch[c]])
            n = self.owner.J; numer = np.zeros(n); tauk = np.zeros(n)
            _lib.check(L.bnmf_cond_params(self.Gch[c]._handle(), 1, int(l) - BLOCK * c, _lib.ptr(numer), _lib.ptr(tauk)))
            return numer, tauk
        b = int(k) // BLOCK
        (k0, k1), (l0, l1) = self.kr[b], self.lr[c]
        self._set3(self.Sch[b][c], F[:, k0:k1], S[k0:k1, l0:l1], G[:, l0:l1], tau)
        self._s_target(b, c, F, S, G, FS, tau)
        numer = np.zeros(1); tauk = np.zeros(1)
        _lib.check(L.bnmtf_cond_params(self.Sch[b][c]._handle(), 1, int(k) - k0, int(l) - l0, _lib.ptr(numer), _lib.ptr(tauk)))
        return numer, tauk

    def close(self):
        for ch in self.children():
            ch.close()
