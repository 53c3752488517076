"""Many variational models in one launch per kernel (csrc/many.h, api_many.inc; bnmtf_amd.run_many with bnmf_vb_optimised models):
the list-form kernels run the single-model kernels' bodies, so every model must end with the BITS of its own run() -- the q
parameters, exptau, the metrics and the ELBO terms of every iteration (bnmf_vb_optimised.py:121-153)."""
import ctypes as C

import numpy as np
import pytest

from bnmtf_amd import _lib, bnmf_vb_optimised, run_many
from bnmtf_amd.cross_validation.replicas import fit_model, fit_models
from bnmtf_amd.synthetic import generate_bnmf

pytestmark = pytest.mark.gpu

PRI = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)
NAMES = ("muU", "tauU", "expU", "varU", "muV", "tauV", "expV", "varV")


def _models(specs, init="random"):
    """specs: (I, J, K, missing fraction, mask seed)"""
    out = []
    for n, (I, J, K, frac, seed) in enumerate(specs):
        R, M, _, _ = generate_bnmf(I, J, min(K, 10), frac, seed_data=1, seed_mask=seed)
        np.random.seed(1000 + n)
        b = bnmf_vb_optimised(R, M, K, PRI, verbose=False)
        b.initialise(init)
        out.append(b)
    return out


def _same(a, b):
    for n in NAMES:
        np.testing.assert_array_equal(getattr(a, n), getattr(b, n), err_msg=n)
    assert a.all_exp_tau == b.all_exp_tau
    assert a.all_performances == b.all_performances
    np.testing.assert_array_equal(a.all_elbo_terms, b.all_elbo_terms)
    assert a.all_elbo == b.all_elbo
    assert a.exptau == b.exptau and a.beta_s == b.beta_s


@pytest.mark.parametrize("specs", [
    [(622, 138, 25, 0.19, s) for s in (2, 3, 4)],                                    # the folds of one rank: one launch per kernel
    [(622, 138, K, 0.19, 5 + i) for i, K in enumerate((15, 20, 25, 30, 15, 20, 25, 30))],      # the line search's ranks together (KP = 32 for all)
    [(300, 200, 8, 0.1, 1), (210, 150, 8, 0.3, 2), (300, 200, 40, 0.1, 3), (64, 500, 5, 0.5, 4)],      # shapes and a 64-wide model: launches split where grids / kernels differ
], ids=["folds", "ranks", "shapes"])
def test_models_run_together_end_with_the_bits_of_their_own_runs(specs):
    alone = _models(specs); together = _models(specs)
    for m in alone:
        m.run(12)
    assert run_many(together, 12) == [None] * len(specs)
    for a, b in zip(alone, together):
        _same(a, b)
    shared, uploads, _ = together[0]._many_info
    assert shared == len(specs)
    assert uploads <= 3 * 11 * len(specs)                  # argument lists are uploaded when they change: not per iteration
    # a second call continues where the first stopped, as run(); run() does
    for m in alone:
        m.run(5)
    run_many(together, 5)
    for a, b in zip(alone, together):
        _same(a, b)


def test_argument_lists_are_not_uploaded_per_iteration():
    ms = _models([(622, 138, 20, 0.19, s) for s in (2, 3, 4, 5)], init="exp")
    run_many(ms, 40)
    assert ms[0]._many_info[1] <= 16, ms[0]._many_info        # 11 launch sites, the first iteration's two or three differences


def test_models_that_cannot_share_launches_run_one_by_one_in_the_same_call():
    specs = [(622, 138, 10, 0.19, 2), (622, 138, 10, 0.19, 3), (622, 138, 10, 0.19, 4)]
    alone = _models(specs); together = _models(specs)
    _lib.check(_lib.lib().bnmtf_set_profiling(together[1]._handle(), 1))     # per-kernel timers: this one stays out of the batch
    for m in alone:
        m.run(6)
    run_many(together, 6)
    for a, b in zip(alone, together):
        _same(a, b)
    assert together[0]._many_info[0] == 2
    # a single batchable model: its own run
    one = _models(specs[:1]); run_many(one, 6)
    _same(alone[0], one[0])
    assert one[0]._many_info[0] == 0


def test_c_entry_point_argument_checks():
    ms = _models([(100, 80, 5, 0.1, 1), (100, 80, 5, 0.1, 2)])
    L = _lib.lib()                          # (include/bnmtf_hip.h: BNMTF_OK 0, BNMTF_EINVAL -1, BNMTF_ESTATE -5)
    for m in ms:
        m._push()
    hs = (C.c_void_p * 2)(ms[0]._handle().value, ms[0]._handle().value)
    assert L.bnmf_vb_run_many(hs, 2, 3, None, None, None, None, None) == -1          # the same model twice
    hs = (C.c_void_p * 2)(ms[0]._handle().value, ms[1]._handle().value)
    assert L.bnmf_vb_run_many(hs, 2, 0, None, None, None, None, None) == 0
    assert L.bnmf_vb_run_many(hs, 2, -1, None, None, None, None, None) == -1
    fresh = bnmf_vb_optimised(ms[0].R, ms[0].M, 5, PRI, verbose=False)
    hs = (C.c_void_p * 2)(ms[0]._handle().value, fresh._handle().value)
    assert L.bnmf_vb_run_many(hs, 2, 3, None, None, None, None, None) == -5          # no state set


def test_fit_models_of_a_batched_pool_gives_fit_models_results():
    R, M, _, _ = generate_bnmf(200, 90, 6, 0.2, seed_data=3, seed_mask=4)
    rs = np.random.RandomState(0)
    jobs = []
    for i in range(5):
        held = (rs.rand(*M.shape) < 0.1) * M
        jobs.append(dict(classifier=bnmf_vb_optimised, args=([4, 6, 8, 6, 4][i], PRI), init={"init": "random"}, iterations=30, burn_in=None, thinning=None,
                         minimum_TN=None, M=M - held, test=held, metrics=["loglikelihood", "AIC", "MSE"], seed=10 + i))
    shared = {"R": np.asarray(R, dtype=float)}
    one_by_one = [fit_model(j, shared) for j in jobs]
    batch = fit_models(jobs, shared)
    for a, b in zip(one_by_one, batch):       # (the same fitted bits; the metric passes sum with fp64 atomics: equal to rounding)
        for m in a["quality"]:
            assert abs(a["quality"][m] - b["quality"][m]) <= 1e-10 * abs(a["quality"][m])
        for m in a["performance"]:
            assert abs(a["performance"][m] - b["performance"][m]) <= 1e-10 * abs(a["performance"][m])
