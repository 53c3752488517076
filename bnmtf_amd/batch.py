"""Independent models in one device call.

    bnmtf_amd.run_many(models, iterations)      # every model ends as if its own run(iterations) had been called

The reference fits the candidates of a model search -- folds x ranks x restarts -- one after the other
(code/cross_validation/line_search_cross_validation.py:54-131, line_search_bnmf.py:53-76) or in a process pool
(parallel_matrix_cross_validation.py:40-74).  Small BNMF and BNMTF Gibbs models run on the device as ONE block each
(csrc/kernel_small.hip), so a list of them is one launch per kind; models that do not qualify are run in turn.  Variational models
(bnmf_vb_optimised, any size its 8-wave kernels serve) walk their iterations in lock-step: every kernel of an iteration is ONE
launch for all of them (csrc/many.h, api_many.inc: bnmf_vb_run_many), each model ending with the bits of its own run().  ICM models
(nmf_icm: their own run(), update rule and minimum_TN) are not taken: ReplicaPool runs them one by one."""
import ctypes as C
import time

import numpy as np

from . import _lib


def takes(model):
    """Is `model` one that run_many fits as its own run() would?  (A Gibbs class itself -- or a subclass that keeps its run().)"""
    return _kind(model) is not None


def _kind(model):
    from .bnmf_gibbs import bnmf_gibbs_optimised
    from .bnmtf_gibbs import bnmtf_gibbs_optimised
    if isinstance(model, bnmf_gibbs_optimised) and type(model).run is bnmf_gibbs_optimised.run:
        return "bnmf"
    if isinstance(model, bnmtf_gibbs_optimised) and type(model).run is bnmtf_gibbs_optimised.run:
        return "bnmtf"
    from .bnmf_vb import bnmf_vb_optimised
    if isinstance(model, bnmf_vb_optimised) and type(model).run is bnmf_vb_optimised.run:
        return "vb"
    return None


def run_many(models, iterations, update='draw', store_samples=True, expectation=None):
    """run(iterations, update, store_samples, expectation) of every model in `models` (bnmf_gibbs_optimised and / or
    bnmtf_gibbs_optimised instances), with the models of the one-launch path that share a device and a kind sharing a single
    launch; bnmf_vb_optimised instances (their run(iterations)): the models of a device walk their iterations in lock-step, one
    launch per kernel for all of them (csrc/api_many.inc).  Returns the list of the runs' results, in the order of `models`."""
    models = list(models)
    if not models:
        return []
    if not all(takes(m) for m in models):
        # (nmf_icm inherits the Gibbs class and overrides run(): its update rule, minimum_TN and Gamma mode are not what the
        # batched entry point runs -- it would come back fitted by Gibbs draws)
        raise TypeError("run_many takes models whose run() is bnmf_gibbs_optimised.run or bnmtf_gibbs_optimised.run (got %s)"
                        % sorted({type(m).__name__ for m in models if not takes(m)}))
    if int(iterations) == 0:                  # run(0) changes nothing (the C entry points return before they fill the final states)
        return [None for _ in models]
    out = [None] * len(models)
    upd = _lib.UPDATE_MODE if update == 'mode' else _lib.UPDATE_DRAW
    _run_many_vb([m for m in models if _kind(m) == "vb"], int(iterations))
    for kind in ("bnmf", "bnmtf"):
        idx = [i for i, m in enumerate(models) if _kind(m) == kind]
        if not idx:
            continue
        ms = [models[i] for i in idx]
        bufs = [m._run_prepare(iterations, store_samples, expectation) for m in ms]
        n = len(ms)
        arr = lambda xs: (C.c_void_p * n)(*[None if x is None else x.ctypes.data for x in xs])
        hs = (C.c_void_p * n)(*[m._handle().value for m in ms])
        if kind == "bnmf":
            states = [(np.zeros((m.I, m.K)), np.zeros((m.J, m.K)), np.zeros(1)) for m in ms]      # what every model ends with
            _lib.check(_lib.lib().bnmf_gibbs_run_many(hs, n, bufs[0][0], upd, arr([b[1] for b in bufs]), arr([b[2] for b in bufs]),
                                                      arr([b[3] for b in bufs]), arr([b[4] for b in bufs]), arr([b[5] for b in bufs]),
                                                      arr([s[0] for s in states]), arr([s[1] for s in states]), arr([s[2] for s in states])))
        else:
            states = [(np.zeros((m.I, m.K)), np.zeros((m.K, m.L)), np.zeros((m.J, m.L)), np.zeros(1)) for m in ms]
            _lib.check(_lib.lib().bnmtf_gibbs_run_many(hs, n, bufs[0][0], upd, arr([b[1] for b in bufs]), arr([b[2] for b in bufs]),
                                                       arr([b[3] for b in bufs]), arr([b[4] for b in bufs]), arr([b[5] for b in bufs]),
                                                       arr([b[6] for b in bufs]),
                                                       arr([s[0] for s in states]), arr([s[1] for s in states]), arr([s[2] for s in states]),
                                                       arr([s[3] for s in states])))
        for i, m, b, st in zip(idx, ms, bufs, states):
            out[i] = m._run_finish(b, store_samples, state=st)
    return out


def _run_many_vb(ms, it):
    """bnmf_vb_optimised.run(it) of every model of `ms` (bnmf_vb_optimised.py:121-153): per device one bnmf_vb_run_many call;
    models wider than 64 columns (column blocks) run on their own."""
    by_device = {}
    for m in ms:
        if m._blocks is not None:
            m.run(it)
        else:
            by_device.setdefault(m._device, []).append(m)
    for group in by_device.values():
        n = len(group)
        for m in group:
            m._push()
        hs = (C.c_void_p * n)(*[m._handle().value for m in group])
        exptau = np.zeros((n, it)); perf = np.zeros((n, it, 3)); terms = np.zeros((n, it, 10)); times = np.zeros((n, it))
        info = np.zeros(2, dtype=np.int32)
        t0 = time.perf_counter()
        _lib.check(_lib.lib().bnmf_vb_run_many(hs, n, it, _lib.ptr(exptau), _lib.ptr(perf), _lib.ptr(terms), _lib.ptr(times), _lib.ptr(info)))
        dt = time.perf_counter() - t0
        for i, m in enumerate(group):
            m._run_finish(it, exptau[i], perf[i], terms[i], times[i])
            m._many_info = (int(info[0]), int(info[1]), dt)     # models that shared launches, argument-list uploads, seconds of the device call
