"""The row/column-sharded path on ONE GPU: two (or three) ranks of one process, one host thread each, joined by the
library's in-process transport (communicator id "BNMTFLOC...": device-to-device copies and a host rendezvous in place of
RCCL's all-gather / all-reduce).  Everything else is the multi-GPU code path as it runs on a node: shard ranges, the
kernels' row offsets, the placement of the gathered factor blocks, the reduction of the three SSE sums, the Philox
counters keyed by global indices.  Expected: every rank ends with the same replicated chain, and it is the chain of the
single-rank run (sweeps are bit-identical; tau can differ in its last fp64 bits through the order of the partial sums)."""
import threading

import numpy as np
import pytest

from bnmtf_amd import bnmf_gibbs_optimised
from bnmtf_amd.synthetic import generate_bnmf

pytestmark = pytest.mark.gpu

PRI = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)


def _run_ranks(R, M, K, U0, V0, tau0, world, iters, update, token):
    cid = (b"BNMTFLOC" + token).ljust(128, b"\0")
    out, err = [None] * world, [None] * world

    def work(rank):
        try:
            b = bnmf_gibbs_optimised(R, M, K, PRI, verbose=False, seed=7, rank=rank, world=world, comm_id=cid)
            b.U, b.V, b.tau = U0.copy(), V0.copy(), tau0
            b.run(iters, update=update)
            out[rank] = (b.all_U.copy(), b.all_V.copy(), b.all_tau.copy(), np.array(b.all_performances["MSE"]))
            b.close()
        except Exception as e:      # noqa: BLE001 -- reported by the main thread
            err[rank] = e

    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in ts), "a rank hung"
    assert all(e is None for e in err), err
    return out


@pytest.mark.parametrize("exchange", ["streams", "serial"])
@pytest.mark.parametrize("I,J,K,world", [(640, 512, 24, 2), (515, 389, 40, 3)])
def test_sharded_run_equals_single_rank_run(monkeypatch, I, J, K, world, exchange):
    """exchange = "streams": the collectives on the exchange stream beside the own-rows Gram, the relayout and the contraction's
    own-rows slices (the default); "serial" (BNMTF_EXCHANGE=serial): all of them on the compute stream in program order."""
    if exchange == "serial":
        monkeypatch.setenv("BNMTF_EXCHANGE", "serial")
    R, M, _, _ = generate_bnmf(I, J, K, 0.12, seed_data=5, seed_mask=6)
    rs = np.random.RandomState(3)
    U0 = rs.exponential(1.0, (I, K)); V0 = rs.exponential(1.0, (J, K)); tau0 = 0.7
    for update in ("mode", "draw"):
        single = bnmf_gibbs_optimised(R, M, K, PRI, verbose=False, seed=7)      # fresh handle: iteration counter (Philox word) 0
        single.U, single.V, single.tau = U0.copy(), V0.copy(), tau0
        single.run(5, update=update)
        ranks = _run_ranks(R, M, K, U0, V0, tau0, world, 5, update, ("%d%s" % (world, update)).encode())
        for r in range(1, world):                      # replicated state: identical on every rank
            for a, b in zip(ranks[0], ranks[r]):
                assert np.array_equal(a, b)
        sU, sV, stau, smse = single.all_U, single.all_V, single.all_tau, np.array(single.all_performances["MSE"])
        # first sweep: same operation order in every kernel -> bit-identical rows and columns
        assert np.array_equal(ranks[0][0][0], sU[0]) and np.array_equal(ranks[0][1][0], sV[0])
        np.testing.assert_allclose(ranks[0][2], stau, rtol=1e-9)
        np.testing.assert_allclose(ranks[0][3], smse, rtol=1e-6)
        assert np.abs(ranks[0][0][-1] - sU[-1]).max() <= 1e-4 * np.abs(sU[-1]).max()


def test_sharded_chain_is_the_same_every_time():
    """Three ranks share the one GPU, so every kernel runs beside two others and its LDS-DMA pieces take longer than
    usual: a barrier that published a staged panel without waiting for the issuing wave's vector-memory counter showed
    up here as a chain that left the single-rank one in 4 runs of 12 (fixed: sync_with_dma, sweep_common.h).  Mode
    updates: nothing random, every run must give the same bits."""
    I, J, K, world = 515, 389, 40, 3
    R, M, _, _ = generate_bnmf(I, J, K, 0.12, seed_data=5, seed_mask=6)
    rs = np.random.RandomState(3)
    U0 = rs.exponential(1.0, (I, K)); V0 = rs.exponential(1.0, (J, K)); tau0 = 0.7
    ref = None
    for n in range(8):
        ranks = _run_ranks(R, M, K, U0, V0, tau0, world, 5, "mode", ("rep%d" % n).encode())
        if ref is None:
            ref = ranks[0]
        for a, b in zip(ref, ranks[0]):
            assert np.array_equal(a, b), "run %d differs from run 0" % n


def _threads(world, work):
    out, err = [None] * world, [None] * world

    def run(rank):
        try:
            out[rank] = work(rank)
        except Exception as e:      # noqa: BLE001 -- reported by the main thread
            err[rank] = e

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in ts), "a rank hung"
    assert all(e is None for e in err), err
    return out


@pytest.mark.parametrize("path", ["pairs", "masked"])
@pytest.mark.parametrize("I,J,K,world", [(640, 512, 24, 2), (515, 389, 40, 3)])
def test_sharded_vb_run_equals_single_rank_run(monkeypatch, I, J, K, world, path):
    """BNMF VB, rows / columns sharded (BASELINE configs[4] "1 vs 8 GPUs"): (E, S2) blocks gathered after each half sweep,
    the SSE-identity sums and the ELBO pieces exchanged as 20 doubles; deterministic, so every rank holds the single-rank
    trajectory (sums are formed in a different order: 1e-9 on the scalars, fp32 noise on the factors).  path: the pair-panel
    sweep, or the on-chip sweep with the masked sums from kernel_maskgemm.hip (BNMTF_VB_PATH, read when the model is built)."""
    from bnmtf_amd import bnmf_vb_optimised
    monkeypatch.setenv("BNMTF_VB_PATH", path)
    R, M, _, _ = generate_bnmf(I, J, K, 0.12, seed_data=5, seed_mask=6)
    pri = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)
    single = bnmf_vb_optimised(R, M, K, pri, verbose=False)
    single.initialise("exp")
    single.run(8)
    cid = (b"BNMTFLOC" + b"vb%d" % world).ljust(128, b"\0")

    def work(rank):
        b = bnmf_vb_optimised(R, M, K, pri, verbose=False, rank=rank, world=world, comm_id=cid)
        b.initialise("exp")
        b.run(8)
        res = (np.array(b.all_exp_tau), np.array(b.all_performances["MSE"]), np.array(b.all_elbo), b.expU.copy(), b.expV.copy(),
               b.muU.copy(), b.tauV.copy(), b.varU.copy())
        b.close()
        return res

    ranks = _threads(world, work)
    for r in range(1, world):
        for a, b in zip(ranks[0], ranks[r]):
            assert np.array_equal(a, b)
    np.testing.assert_allclose(ranks[0][0], single.all_exp_tau, rtol=1e-6)
    np.testing.assert_allclose(ranks[0][1], single.all_performances["MSE"], rtol=1e-6)
    np.testing.assert_allclose(ranks[0][2], single.all_elbo, rtol=1e-6)
    for got, ref in zip(ranks[0][3:], (single.expU, single.expV, single.muU, single.tauV, single.varU)):
        assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max()


@pytest.mark.parametrize("I,J,K,L,world", [(640, 512, 8, 6, 2), (515, 389, 12, 9, 3), (1300, 1100, 32, 32, 2)])
def test_sharded_bnmtf_vb_run_equals_single_rank_run(I, J, K, L, world):
    """bnmtf_vb over several ranks (round 6): rows of F and columns of G updated by their owners, the blocks of (E, var, S2)
    gathered behind each half sweep, the S system summed over the ranks' column ranges and walked by every rank, the iteration's
    sums exchanged as 21 doubles.  Deterministic: every rank holds the same trajectory, the single-rank one up to the order of sums."""
    from bnmtf_amd import bnmtf_vb_optimised
    from bnmtf_amd.synthetic import generate_bnmtf
    R, M, _, _, _ = generate_bnmtf(I, J, K, L, 0.12, seed_data=5, seed_mask=6)
    pri = dict(alpha=1., beta=1., lambdaF=0.1, lambdaS=0.1, lambdaG=0.1)
    rs = np.random.RandomState(4)
    n_it = 4
    orders = np.array([np.concatenate([rs.permutation(K * L), rs.permutation(K), rs.permutation(L)]) for _ in range(n_it)], dtype=np.int32)

    names = ["muF", "tauF", "expF", "varF", "muS", "tauS", "expS", "varS", "muG", "tauG", "expG", "varG"]
    b0 = bnmtf_vb_optimised(R, M, K, L, pri, verbose=False)
    np.random.seed(3)
    b0.initialise("random", "random")
    init = {n: getattr(b0, n).copy() for n in names}
    init_exptau = float(b0.exptau)
    b0.close()

    def fit(**kw):
        b = bnmtf_vb_optimised(R, M, K, L, pri, verbose=False, **kw)
        for n in names:                      # (NumPy's global stream is not a thing to draw from in several threads)
            setattr(b, n, init[n].copy())
        b.exptau = init_exptau
        b.run(n_it, orders=orders)
        res = (np.array(b.all_exp_tau), np.array(b.all_performances["MSE"]), b.expF.copy(), b.expS.copy(), b.expG.copy(), b.muF.copy(), b.tauG.copy(), b.varG.copy(),
               np.array(b.beta_s), np.array(b.exp_square_diff()))
        b.close()
        return res

    single = fit()
    cid = (b"BNMTFLOC" + ("tvb%d_%d" % (world, K)).encode()).ljust(128, b"\0")
    ranks = _threads(world, lambda rank: fit(rank=rank, world=world, comm_id=cid))
    for r in range(1, world):
        for a, b in zip(ranks[0][:9], ranks[r][:9]):
            assert np.array_equal(a, b)
        assert abs(float(ranks[0][9]) / float(ranks[r][9]) - 1) < 1e-10       # (the direct pass adds its blocks' sums with atomics)
    np.testing.assert_allclose(ranks[0][0], single[0], rtol=2e-4)
    np.testing.assert_allclose(ranks[0][1], single[1], rtol=2e-4)
    for got, ref in zip(ranks[0][2:8], single[2:8]):
        assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max()
    # update_tau of the last iteration (the exchanged sums) against the direct exp_square_diff of the gathered state
    assert abs(float(ranks[0][8]) - (1.0 + 0.5 * float(ranks[0][9]))) < 5e-5 * float(ranks[0][8])


@pytest.mark.parametrize("I,J,K,L,world", [(512, 640, 12, 9, 2), (389, 515, 20, 32, 3)])
def test_sharded_bnmtf_run_equals_single_rank_run(I, J, K, L, world):
    """BNMTF Gibbs sharded: F rows / G columns drawn by their owners and gathered; the S step's (A, b) summed over the
    ranks' column ranges with one all-reduce (the K.L Gram exchange), then every rank walks the same K.L conditionals."""
    from bnmtf_amd import bnmtf_gibbs_optimised
    from bnmtf_amd.synthetic import generate_bnmtf
    R, M, _, _, _ = generate_bnmtf(I, J, K, L, 0.12, seed_data=5, seed_mask=6)
    pri = dict(alpha=1., beta=1., lambdaF=0.1, lambdaS=0.1, lambdaG=0.1)
    rs = np.random.RandomState(3)
    F0 = rs.exponential(1.0, (I, K)); S0 = rs.exponential(1.0, (K, L)); G0 = rs.exponential(1.0, (J, L)); tau0 = 0.7
    for update in ("mode", "draw"):
        single = bnmtf_gibbs_optimised(R, M, K, L, pri, verbose=False, seed=7)
        single.F, single.S, single.G, single.tau = F0.copy(), S0.copy(), G0.copy(), tau0
        single.run(4, update=update)
        cid = (b"BNMTFLOC" + ("tri%d%s" % (world, update)).encode()).ljust(128, b"\0")

        def work(rank):
            b = bnmtf_gibbs_optimised(R, M, K, L, pri, verbose=False, seed=7, rank=rank, world=world, comm_id=cid)
            b.F, b.S, b.G, b.tau = F0.copy(), S0.copy(), G0.copy(), tau0
            b.run(4, update=update)
            res = (b.all_F.copy(), b.all_S.copy(), b.all_G.copy(), b.all_tau.copy(), np.array(b.all_performances["MSE"]))
            b.close()
            return res

        ranks = _threads(world, work)
        for r in range(1, world):
            for a, b in zip(ranks[0], ranks[r]):
                assert np.array_equal(a, b)
        # F of the first sweep: same operation order in every kernel -> bit-identical; S goes through sums over column
        # ranges whose order depends on the split (fp32 rounding), everything after it inherits that
        assert np.array_equal(ranks[0][0][0], single.all_F[0])
        # (a draw whose acceptance test sits within that rounding of its threshold takes the next candidate: rare)
        close = np.abs(ranks[0][1][0] - single.all_S[0]) <= 2e-4 * np.abs(single.all_S[0]).max()
        assert close.all() if update == "mode" else close.mean() > 0.97
        np.testing.assert_allclose(ranks[0][4][:2], np.array(single.all_performances["MSE"])[:2], rtol=2e-3)
        if update == "mode":
            np.testing.assert_allclose(ranks[0][3], single.all_tau, rtol=1e-3)
            assert np.abs(ranks[0][2][-1] - single.all_G[-1]).max() <= 5e-3 * np.abs(single.all_G[-1]).max()


def test_eight_ranks_at_the_headline_shard_shapes_draw_the_single_rank_chain():
    """The multi-GPU code at the REAL shard shapes of the headline configuration (8192 x 8192, K = 64, rows 8 x 1024): eight ranks
    of one process on ONE GPU (in-process transport) against the single-rank run.  Every kernel launch, shard range, row offset,
    the contraction's part-1 / part-2 split around the exchange and the exchange stream's dependencies are the code an 8-GPU
    run executes (exchange_factor, api.hip).  Every rank ends with the same bits; the chain is the single-rank chain -- the first
    sweep of U element-wise but for flipped accept / reject decisions (the one GPU hands q over between its half sweeps, the
    shards rebuild it), the first MSE to 1e-5, the trajectory together afterwards.  (Until round 5 a tool:
    tools/shard_check_8192.py.)"""
    I = J = 8192; K = 64; world = 8; iters = 4
    R, M, _, _ = generate_bnmf(I, J, K, 0.1, tau=1.0, seed_data=0, seed_mask=1)
    rs = np.random.RandomState(3)
    U0 = rs.exponential(10.0, (I, K)); V0 = rs.exponential(10.0, (J, K)); tau0 = 1.0
    s = bnmf_gibbs_optimised(R, M, K, PRI, verbose=False, seed=7)
    s.U, s.V, s.tau = U0.copy(), V0.copy(), tau0
    s.run(iters)
    sm = np.array(s.all_performances["MSE"]); sU0 = s.all_U[0].copy()
    s.close()
    cid = b"BNMTFLOC8192T".ljust(128, b"\0")
    out, err = [None] * world, [None] * world

    def work(rank):
        try:
            b = bnmf_gibbs_optimised(R, M, K, PRI, verbose=False, seed=7, rank=rank, world=world, comm_id=cid)
            b.U, b.V, b.tau = U0.copy(), V0.copy(), tau0
            b.run(iters, store_samples=(rank == 0))
            out[rank] = (np.array(b.all_performances["MSE"]), b.U.copy(), b.all_U[0].copy() if rank == 0 else None)
            b.close()
        except Exception as e:      # noqa: BLE001
            err[rank] = e
    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not any(err), err
    for r in range(1, world):
        assert np.array_equal(out[0][1], out[r][1]) and np.array_equal(out[0][0], out[r][0]), r
    rel = np.abs(out[0][0] / sm - 1)
    d0 = np.abs(out[0][2] - sU0) / (1e-3 + np.abs(sU0))
    assert float(np.mean(d0 < 1e-3)) > 0.999 and rel[0] < 1e-5 and rel.max() < 5e-3, (float(np.mean(d0 < 1e-3)), rel)
