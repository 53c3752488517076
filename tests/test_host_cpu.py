"""CPU-only checks of the product's host side: the C-ABI library loads and exports every symbol
the header declares, the classes reproduce the reference's constructor contract (exact assertion
messages), the NumPy-only post-run helpers, and failing loudly without a GPU."""
import os
import re

import numpy as np
import pytest

import bnmtf_amd
from bnmtf_amd import _lib, bnmf_gibbs_optimised, bnmtf_gibbs_optimised, bnmf_vb_optimised
from bnmtf_amd.comm import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "bnmtf_hip.h")).read()
    declared = set(re.findall(r"^BNMTF_API\s+(?:int|const char\*)\s+(bnmt?f_\w+)\s*\(", hdr, flags=re.M))
    assert declared, "no declarations parsed"
    lib = bnmtf_amd.lib()
    for name in declared:
        assert hasattr(lib, name), "libbnmtf_hip.so does not export %s" % name
    assert declared == set(_lib.EXPORTS), (declared ^ set(_lib.EXPORTS))
    assert lib.bnmtf_version() >= 100
    # ... and nothing else: -fvisibility=hidden and a linker version script (csrc/exports.map) leave the header's entry points as
    # the library's only dynamic symbols
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", bnmtf_amd.LIB_PATH], capture_output=True, text=True, check=True).stdout
    syms = {l.split()[-1] for l in out.splitlines() if l.strip()}
    assert syms == declared, sorted(syms ^ declared)[:20]
    assert lib.bnmtf_has_experiments() == 0, "the shipped build has no experiment kernels (make EXPERIMENTS=1 is a tools/ build)"


def test_no_cpu_fallback_without_gpu():
    if bnmtf_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    R = np.ones((4, 3)); M = np.ones((4, 3))
    b = bnmf_gibbs_optimised(R, M, 2, dict(alpha=1, beta=1, lambdaU=1., lambdaV=1.), verbose=False)
    b.U = np.ones((4, 2)); b.V = np.ones((3, 2)); b.tau = 1.0
    with pytest.raises(bnmtf_amd.BnmtfError):
        b.run(1)
    with pytest.raises(bnmtf_amd.BnmtfError):
        b.beta_s()
    from bnmtf_amd import distributions as D
    with pytest.raises(bnmtf_amd.BnmtfError):
        D.TN_vector_draw([1.0], [1.0])


@pytest.mark.parametrize("cls", ["bnmf", "vb"])
def test_constructor_contract_bnmf(cls):
    """tests/code/test_bnmf_gibbs_optimised.py:14-105 (same text for bnmf_vb_optimised)."""
    mk = (lambda R, M, K, p: bnmf_gibbs_optimised(R, M, K, p, verbose=False)) if cls == "bnmf" else \
         (lambda R, M, K, p: bnmf_vb_optimised(R, M, K, p, verbose=False))
    M = np.ones((2, 3)); I, J, K = 5, 3, 1
    priors = {'alpha': 3, 'beta': 1, 'lambdaU': np.ones((I, K)), 'lambdaV': np.ones((J, K))}
    with pytest.raises(AssertionError) as e:
        mk(np.ones(3), M, K, priors)
    assert str(e.value) == "Input matrix R is not a two-dimensional array, but instead 1-dimensional."
    with pytest.raises(AssertionError) as e:
        mk(np.ones((4, 3, 2)), M, K, priors)
    assert str(e.value) == "Input matrix R is not a two-dimensional array, but instead 3-dimensional."
    with pytest.raises(AssertionError) as e:
        mk(np.ones((3, 2)), M, K, priors)
    assert str(e.value) == "Input matrix R is not of the same size as the indicator matrix M: (3, 2) and (2, 3) respectively."
    R4 = np.ones((2, 3))
    with pytest.raises(AssertionError) as e:
        mk(R4, M, K, {'alpha': 3, 'beta': 1, 'lambdaU': np.ones((3, 1)), 'lambdaV': np.ones((3, 1))})
    assert str(e.value) == "Prior matrix lambdaU has the wrong shape: (3, 1) instead of (2, 1)."
    with pytest.raises(AssertionError) as e:
        mk(R4, M, K, {'alpha': 3, 'beta': 1, 'lambdaU': np.ones((2, 1)), 'lambdaV': np.ones((4, 1))})
    assert str(e.value) == "Prior matrix lambdaV has the wrong shape: (4, 1) instead of (3, 1)."
    pr = {'alpha': 3, 'beta': 1, 'lambdaU': np.ones((2, 1)), 'lambdaV': np.ones((3, 1))}
    with pytest.raises(AssertionError) as e:
        mk(R4, [[1, 1, 1], [0, 0, 0]], K, pr)
    assert str(e.value) == "Fully unobserved row in R, row 1."
    with pytest.raises(AssertionError) as e:
        mk(R4, [[1, 1, 0], [1, 0, 0]], K, pr)
    assert str(e.value) == "Fully unobserved column in R, column 2."
    I, J, K = 3, 2, 2
    R5 = 2 * np.ones((I, J)); M5 = np.ones((I, J))
    b = mk(R5, M5, K, {'alpha': 3, 'beta': 1, 'lambdaU': 3., 'lambdaV': 4.})
    assert np.array_equal(b.R, R5) and np.array_equal(b.M, M5) and (b.I, b.J, b.K) == (I, J, K)
    assert b.size_Omega == I * J and b.alpha == 3 and b.beta == 1
    assert np.array_equal(b.lambdaU, 3. * np.ones((I, K))) and np.array_equal(b.lambdaV, 4. * np.ones((J, K)))
    # keyword form used by the cross-validation drivers (line_search_cross_validation.py:109-114)
    b = (bnmf_gibbs_optimised if cls == "bnmf" else bnmf_vb_optimised)(R=R5, M=M5, K=K, priors={'alpha': 3, 'beta': 1, 'lambdaU': 3., 'lambdaV': 4.})
    assert b.K == K


def test_constructor_contract_bnmtf():
    """tests/code/test_bnmtf_gibbs_optimised.py constructor checks."""
    I, J, K, L = 5, 3, 1, 2
    pri = {'alpha': 3, 'beta': 1, 'lambdaF': np.ones((I, K)), 'lambdaS': np.ones((K, L)), 'lambdaG': np.ones((J, L))}
    with pytest.raises(AssertionError) as e:
        bnmtf_gibbs_optimised(np.ones(3), np.ones((2, 3)), K, L, pri)
    assert str(e.value) == "Input matrix R is not a two-dimensional array, but instead 1-dimensional."
    R4 = np.ones((2, 3)); M = np.ones((2, 3))
    with pytest.raises(AssertionError) as e:
        bnmtf_gibbs_optimised(R4, M, K, L, {'alpha': 3, 'beta': 1, 'lambdaF': np.ones((3, 1)), 'lambdaS': np.ones((1, 2)), 'lambdaG': np.ones((3, 2))})
    assert str(e.value) == "Prior matrix lambdaF has the wrong shape: (3, 1) instead of (2, 1)."
    with pytest.raises(AssertionError) as e:
        bnmtf_gibbs_optimised(R4, M, K, L, {'alpha': 3, 'beta': 1, 'lambdaF': np.ones((2, 1)), 'lambdaS': np.ones((2, 2)), 'lambdaG': np.ones((3, 2))})
    assert str(e.value) == "Prior matrix lambdaS has the wrong shape: (2, 2) instead of (1, 2)."
    with pytest.raises(AssertionError) as e:
        bnmtf_gibbs_optimised(R4, M, K, L, {'alpha': 3, 'beta': 1, 'lambdaF': np.ones((2, 1)), 'lambdaS': np.ones((1, 2)), 'lambdaG': np.ones((4, 2))})
    assert str(e.value) == "Prior matrix lambdaG has the wrong shape: (4, 2) instead of (3, 2)."
    b = bnmtf_gibbs_optimised(R4, M, K, L, {'alpha': 3, 'beta': 1, 'lambdaF': 2., 'lambdaS': 3., 'lambdaG': 4.}, verbose=False)
    assert np.array_equal(b.lambdaS, 3. * np.ones((K, L))) and b.L == L
    with pytest.raises(AssertionError) as e:
        b.initialise(init_S='nope')
    assert str(e.value) == "Unknown initialisation option for S: nope. Should be 'random' or 'exp'."


def test_postrun_host_helpers(golden):
    """approx_expectation accepts lists (tests :240-267); compute_MSE/R2/Rp closed forms (:308-328)."""
    I, J, K = 5, 3, 2
    b = bnmf_gibbs_optimised(np.ones((I, J)), np.ones((I, J)), K, {'alpha': 3, 'beta': 1, 'lambdaU': 2., 'lambdaV': 3.}, verbose=False)
    b.all_U = [np.ones((I, K)) * 3 * m ** 2 for m in range(1, 11)]
    b.all_V = [np.ones((J, K)) * 2 * m ** 2 for m in range(1, 11)]
    b.all_tau = [m ** 2 for m in range(1, 11)]
    eU, eV, et = b.approx_expectation(2, 3)
    assert et == (9. + 36. + 81.) / 3. and np.array_equal(eU, (9. + 36. + 81.) * np.ones((I, K)))
    assert np.array_equal(eV, (9. + 36. + 81.) * (2. / 3.) * np.ones((J, K)))
    R = np.array([[1, 2], [3, 4]], dtype=float); Mp = np.array([[0, 0], [1, 1]])
    Rp = np.array([[500, 550], [1220, 1342]], dtype=float)
    import math
    assert b.compute_MSE(Mp, R, Rp) == (1217 ** 2 + 1338 ** 2) / 2.0
    assert b.compute_R2(Mp, R, Rp) == 1. - (1217 ** 2 + 1338 ** 2) / (0.5 ** 2 + 0.5 ** 2)
    assert b.compute_Rp(Mp, R, Rp) == 61. / (math.sqrt(.5) * math.sqrt(7442.))
    with pytest.raises(AssertionError) as e:
        b.initialise('nope')
    assert str(e.value) == "Unknown initialisation option: nope. Should be 'random' or 'exp'."
    with pytest.raises(AssertionError) as e:
        b.quality('FAIL', 0, 1)
    assert str(e.value) == "Unrecognised metric for model quality: FAIL."
    from bnmtf_amd._base import metrics_from_sums
    from oracle import bnmtf_oracle as O
    c = golden("bnmf_gibbs_cond.npz").case("r37x29")
    m = metrics_from_sums(O.metric_sums(c["M"], c["R"], c["U"] @ c["V"].T))
    np.testing.assert_allclose([m["MSE"], m["R^2"], m["Rp"]], c["perf"], rtol=1e-9)


def test_shard_ranges_partition_everything():
    for n in (1, 7, 100, 8192, 8191):
        for world in (1, 2, 3, 8):
            if world > n:
                continue
            pos = 0
            for r in range(world):
                first, count = shard_range(n, r, world)
                assert first == pos and count >= n // world
                pos += count
            assert pos == n


def test_product_kmeans_has_no_cpu_path():
    """bnmtf_amd.kmeans.KMeans runs its two O(points x coordinates x K) passes in libbnmtf_hip.so only: without a GPU the
    first assignment raises (the NumPy restatement that checks it is oracle/kmeans_oracle.py, test infrastructure)."""
    from bnmtf_amd.kmeans import KMeans
    from bnmtf_amd import BnmtfError, device_count
    if device_count() > 0:
        pytest.skip("a GPU is present")
    rs = np.random.RandomState(0)
    X = rs.randn(12, 4); Mk = np.ones((12, 4))
    km = KMeans(X, Mk, 3); km.initialise(seed=1)
    with pytest.raises(BnmtfError):
        km.cluster()


def test_sharded_model_without_seed_gets_one_shared_key_and_one_shared_initialisation():
    """Every rank of a sharded model must run the same chain: with seed=None the Philox key is derived from the
    communicator id all ranks hold (not from each process' own NumPy stream), and initialise('random') draws from a
    stream seeded with it, so the replicated U, V are the same on every rank.  (bnmtf_create re-checks the key across
    the ranks through the communicator.)"""
    R = np.ones((6, 5)); M = np.ones((6, 5))
    pri = dict(alpha=1, beta=1, lambdaU=1., lambdaV=1.)
    cid = bytes(range(128))
    ranks = []
    for r in range(2):
        np.random.seed(100 + r)                       # the processes' own global streams differ
        ranks.append(bnmf_gibbs_optimised(R, M, 3, pri, verbose=False, rank=r, world=2, comm_id=cid))
    assert ranks[0]._seed == ranks[1]._seed and ranks[0]._seed is not None
    a, b = ranks[0]._rng().exponential(size=(6, 3)), ranks[1]._rng().exponential(size=(6, 3))
    assert np.array_equal(a, b)
    other = bnmf_gibbs_optimised(R, M, 3, pri, verbose=False, rank=0, world=2, comm_id=bytes(128))
    assert other._seed != ranks[0]._seed
    single = bnmf_gibbs_optimised(R, M, 3, pri, verbose=False)
    assert single._rng() is np.random                 # single GPU: the reference's global stream
    with pytest.raises(AssertionError):
        bnmf_gibbs_optimised(R, M, 3, pri, verbose=False, rank=0, world=2)
    explicit = bnmf_gibbs_optimised(R, M, 3, pri, verbose=False, seed=5, rank=1, world=2, comm_id=cid)
    assert explicit._seed == 5


def test_non_binary_mask_is_rejected_before_the_device_sees_it():
    R = np.ones((4, 3)); M = np.ones((4, 3)); M[1, 1] = 0.5
    b = bnmf_gibbs_optimised(R, M, 2, dict(alpha=1, beta=1, lambdaU=1., lambdaV=1.), verbose=False)
    b.U = np.ones((4, 2)); b.V = np.ones((3, 2)); b.tau = 1.0
    with pytest.raises(AssertionError) as e:
        b.run(1)
    assert str(e.value) == "The indicator matrix M must contain only 0 and 1."


def test_gdsc_text_loader_round_trip(tmp_path):
    """bnmtf_amd.data: the drug-sensitivity text format of data_drug_sensitivity/gdsc/load_data.py:15-86."""
    from bnmtf_amd import data
    f = str(tmp_path / "gdsc.txt")
    X = np.array([[1.5, 0.0, -2.0], [0.0, 3.25, 4.0]]); M = np.array([[1., 0., 1.], [0., 1., 1.]])
    data.store_gdsc(f, X, M, ["d1", "d2", "d3"], ["c1", "c2"], ["t1", "t2"], ["s1", "s2"])
    X2, Xmin, M2, drugs, cells, cancers, tissues = data.load_gdsc(f, sep="\t")
    assert np.array_equal(M2, M) and np.array_equal(X2, X * M) and drugs == ["d1", "d2", "d3"] and cells == ["c1", "c2"] and tissues == ["s1", "s2"]
    assert np.array_equal(Xmin, np.where(M == 1, X - (-2.0 - 1), 0.0))
    assert np.array_equal(data.negate_gdsc(X2, M2), np.where(M == 1, -X + 4.0, 0.0))


def test_bench_clock_reading_is_optional():
    """bench.py's `clock` object: rocm-smi read (by a helper process started before the GPU is touched) beside an untimed loop.
    Without a GPU (or without rocm-smi) it is None and nothing raises; without a helper likewise."""
    import importlib.util, os, time
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench._clock_beside(None, lambda: time.sleep(0.01), lambda: None, seconds=0.1) is None
    got = bench._clock_beside(bench._clock_helper_start(), lambda: time.sleep(0.01), lambda: None, seconds=0.3)
    assert got is None or (got["sclk_mhz_median"] > 0 and got["readings"] >= 1)


def test_run_many_takes_gibbs_models_only():
    """The batch entry point fits a model as bnmf_gibbs_optimised.run would: a subclass with its own run() (nmf_icm: another
    update rule, minimum_TN, the Gamma mode) is refused before anything reaches the device, and the replica pool's batching asks
    the same question (round 4's advice: such jobs came back fitted by Gibbs draws)."""
    from bnmtf_amd import nmf_icm, run_many
    from bnmtf_amd.batch import takes
    rs = np.random.RandomState(0)
    R = rs.rand(6, 5); M = np.ones((6, 5))
    pri = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)
    assert takes(bnmf_gibbs_optimised(R, M, 2, pri)) and not takes(nmf_icm(R, M, 2, pri)) and not takes(object())
    with pytest.raises(TypeError):
        run_many([nmf_icm(R, M, 2, pri)], 3)
    assert run_many([], 3) == []


def test_joint_map_merges_the_calls_the_threads_have_open():
    """ReplicaPool.joint: the map() calls of the participating threads that are open at the same time reach the pool as ONE map();
    a thread that is done leaves; results go back to their callers in their own order."""
    import threading
    from bnmtf_amd.cross_validation.replicas import ReplicaPool, _Joint

    class Recorder(object):
        shared, devices, batched = {}, [0], True
        def __init__(self):
            self.calls = []
        def map(self, fn, jobs, errors="raise"):
            self.calls.append(len(jobs))
            return [fn(j, None) for j in jobs]

    rec = Recorder()
    joint = _Joint(rec, 3)
    out = {}
    def walker(name, steps):
        try:
            res = []
            for s in range(steps):
                res.append(joint.map(lambda j, _: (name, j), [10 * s + i for i in range(name + 1)]))
            out[name] = res
        finally:
            joint.leave()
    ts = [threading.Thread(target=walker, args=(n, st)) for n, st in ((0, 1), (1, 3), (2, 2))]
    for t in ts: t.start()
    for t in ts: t.join(timeout=30)
    assert not any(t.is_alive() for t in ts)
    # (the one lambda per call is its own fn: the recorder sees one map per fn and round -- 3 + 2 + 1 calls of 1..3 jobs)
    assert sorted(rec.calls) == sorted([1, 2, 3, 2, 3, 2])
    assert out[1] == [[(1, 0), (1, 1)], [(1, 10), (1, 11)], [(1, 20), (1, 21)]] and out[0] == [[(0, 0)]]
    # with one shared fn the open calls are one map
    rec2 = Recorder(); joint2 = _Joint(rec2, 2)
    f = lambda j, _: j * 2
    got = {}
    def w2(name):
        got[name] = joint2.map(f, [name, name + 1]); joint2.leave()
    ts = [threading.Thread(target=w2, args=(n,)) for n in (1, 5)]
    for t in ts: t.start()
    for t in ts: t.join(timeout=30)
    assert rec2.calls == [4] and got == {1: [2, 4], 5: [10, 12]}
    assert isinstance(ReplicaPool(devices=[0]).joint(2), _Joint)


def test_joint_map_survives_a_walk_that_fails():
    """A thread that raises between its map() calls leaves (the caller's `finally`): the others are not left waiting for it; an
    exception inside the merged call reaches every caller of that call."""
    import threading
    from bnmtf_amd.cross_validation.replicas import _Joint

    class Pool(object):
        shared, devices, batched = {}, [0], True
        def map(self, fn, jobs, errors="raise"):
            if any(j == "boom" for j in jobs):
                raise ValueError("the device call failed")
            return [fn(j, None) for j in jobs]

    f = lambda j, _: j
    joint = _Joint(Pool(), 3)
    out = {}
    def good(name):
        try:
            out[name] = [joint.map(f, [name]), joint.map(f, [name + 10])]
        finally:
            joint.leave()
    def bad():
        try:
            joint.map(f, [0])
            raise RuntimeError("this walk stops here")
        except RuntimeError:
            out["bad"] = "stopped"
        finally:
            joint.leave()
    ts = [threading.Thread(target=good, args=(1,)), threading.Thread(target=good, args=(2,)), threading.Thread(target=bad)]
    for t in ts: t.start()
    for t in ts: t.join(timeout=30)
    assert not any(t.is_alive() for t in ts)
    assert out == {1: [[1], [11]], 2: [[2], [12]], "bad": "stopped"}
    joint = _Joint(Pool(), 2)
    errs = []
    def failing(job):
        try:
            joint.map(f, [job])
        except ValueError as e:
            errs.append(str(e))
        finally:
            joint.leave()
    ts = [threading.Thread(target=failing, args=(j,)) for j in ("boom", "fine")]
    for t in ts: t.start()
    for t in ts: t.join(timeout=30)
    assert not any(t.is_alive() for t in ts) and errs == ["the device call failed"] * 2


def test_rank_limits_are_said_at_construction_and_wide_bnmf_models_are_column_blocks():
    """The reference takes any rank (bnmf_gibbs_optimised.py:54-78).  Here: bnmf_gibbs / nmf_icm up to 256 (column blocks of 64,
    _blocked.py: built without touching a device), the Gibbs / ICM tri-factorisations up to 256 as blocks of S, the variational
    one up to 32 -- and what lies beyond is refused when
    the class is constructed, with the limit in the message, not at the first device call of a search."""
    import bnmtf_amd
    from bnmtf_amd._lib import BnmtfError
    from bnmtf_amd._blocked import block_ranges
    R = np.ones((6, 5)); M = np.ones((6, 5))
    pri = dict(alpha=1., beta=1., lambdaU=1., lambdaV=1.)
    b = bnmtf_amd.bnmf_gibbs_optimised(R, M, 200, pri, verbose=False)
    assert b._blocks is not None and b._blocks.ranges == [(0, 64), (64, 128), (128, 192), (192, 200)] == block_ranges(200)
    assert [c.K for c in b._blocks.children] == [64, 64, 64, 8] and b._blocks.children[3].lambdaU.shape == (6, 8)
    assert bnmtf_amd.nmf_icm(R, M, 65, pri, verbose=False)._blocks.ranges == [(0, 64), (64, 65)]
    assert bnmtf_amd.bnmf_gibbs_optimised(R, M, 64, pri, verbose=False)._blocks is None
    with pytest.raises(BnmtfError, match="K = 257 is outside what this build runs"):
        bnmtf_amd.bnmf_gibbs_optimised(R, M, 257, pri, verbose=False)
    pri3 = dict(alpha=1., beta=1., lambdaF=1., lambdaS=1., lambdaG=1.)
    t = bnmtf_amd.bnmtf_gibbs_optimised(R, M, 65, 130, pri3, verbose=False)              # blocks of S: 2 x 3 (TriBlocks)
    assert (t._blocks.kr, t._blocks.lr) == ([(0, 64), (64, 65)], [(0, 64), (64, 128), (128, 130)])
    assert [c.K for c in t._blocks.Fch] == [64, 1] and [c.K for c in t._blocks.Gch] == [64, 64, 2]
    assert [[(c.K, c.L) for c in row] for row in t._blocks.Sch] == [[(64, 64), (64, 64), (64, 2)], [(1, 64), (1, 64), (1, 2)]]
    assert bnmtf_amd.nmtf_icm(R, M, 3, 65, pri3, verbose=False)._blocks is not None and bnmtf_amd.bnmtf_gibbs_optimised(R, M, 64, 64, pri3, verbose=False)._blocks is None
    with pytest.raises(BnmtfError, match="outside what this build runs"):
        bnmtf_amd.bnmtf_gibbs_optimised(R, M, 257, 3, pri3, verbose=False)
    with pytest.raises(BnmtfError, match="outside what this build runs"):
        bnmtf_amd.bnmtf_vb_optimised(R, M, 33, 3, pri3, verbose=False)
    assert bnmtf_amd.bnmf_vb_optimised(R, M, 65, pri, verbose=False)._blocks.ranges == [(0, 64), (64, 65)]
    with pytest.raises(BnmtfError, match="outside what this build runs"):
        bnmtf_amd.bnmf_vb_optimised(R, M, 257, pri, verbose=False)
