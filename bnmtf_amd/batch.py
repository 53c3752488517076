"""Independent models in one device call.

    bnmtf_amd.run_many(models, iterations)      # every model ends as if its own run(iterations) had been called

The reference fits the candidates of a model search -- folds x ranks x restarts -- one after the other
(code/cross_validation/line_search_cross_validation.py:54-131, line_search_bnmf.py:53-76) or in a process pool
(parallel_matrix_cross_validation.py:40-74).  Small BNMF Gibbs / ICM models run on the device as ONE block each
(csrc/kernel_small.hip), so a list of them is one launch; models that do not qualify are run in turn.  ICM models (nmf_icm:
their own run(), update rule and minimum_TN) are not taken: ReplicaPool runs them one by one."""
import ctypes as C

import numpy as np

from . import _lib


def takes(model):
    """Is `model` one that run_many fits as its own run() would?  (The Gibbs class itself -- or a subclass that keeps its run().)"""
    from .bnmf_gibbs import bnmf_gibbs_optimised
    return isinstance(model, bnmf_gibbs_optimised) and type(model).run is bnmf_gibbs_optimised.run


def run_many(models, iterations, update='draw', store_samples=True, expectation=None):
    """run(iterations, update, store_samples, expectation) of every model in `models` (bnmf_gibbs_optimised instances, all on
    one device), with the models of the one-launch path sharing a single launch.  Returns the list of the runs' results."""
    models = list(models)
    if not models:
        return []
    if not all(takes(m) for m in models):
        # (nmf_icm inherits the Gibbs class and overrides run(): its update rule, minimum_TN and Gamma mode are not what the
        # batched entry point runs -- it would come back fitted by Gibbs draws)
        raise TypeError("run_many takes models whose run() is bnmf_gibbs_optimised.run (got %s)" % sorted({type(m).__name__ for m in models if not takes(m)}))
    if int(iterations) == 0:                  # run(0) changes nothing (the C entry point returns before it fills the final states)
        return [None for _ in models]
    bufs = [m._run_prepare(iterations, store_samples, expectation) for m in models]
    n = len(models)
    states = [(np.zeros((m.I, m.K)), np.zeros((m.J, m.K)), np.zeros(1)) for m in models]      # what every model ends with
    arr = lambda xs: (C.c_void_p * n)(*[None if x is None else x.ctypes.data for x in xs])
    hs = (C.c_void_p * n)(*[m._handle().value for m in models])
    _lib.check(_lib.lib().bnmf_gibbs_run_many(hs, n, bufs[0][0], _lib.UPDATE_MODE if update == 'mode' else _lib.UPDATE_DRAW,
                                              arr([b[1] for b in bufs]), arr([b[2] for b in bufs]), arr([b[3] for b in bufs]),
                                              arr([b[4] for b in bufs]), arr([b[5] for b in bufs]),
                                              arr([s[0] for s in states]), arr([s[1] for s in states]), arr([s[2] for s in states])))
    return [m._run_finish(b, store_samples, state=s) for m, b, s in zip(models, bufs, states)]
