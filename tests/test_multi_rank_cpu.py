"""world_size-2 (gloo, CPU) check of the multi-GPU scheme libbnmtf_hip.so implements with RCCL:
rows of R split over the ranks for the U sweep, columns for the V sweep (bnmtf_shard_range), the
freshly drawn factor blocks all-gathered after each half sweep, three scalars all-reduced for the
Gram-identity SSE, tau drawn redundantly from the same Philox counter.  The per-shard arithmetic is
the oracle's (no GPU here); what is verified is the partition / exchange / RNG-keying design:
two ranks reproduce the single-process chain."""
import os
import socket

import numpy as np
import pytest


class _Lazy(object):
    """torch, imported when a test of THIS module first touches it.  A `pytest tests -m gpu` session imports every test module at
    collection; torch brings the HIP runtime bundled with its wheel into the process, and libbnmtf_hip.so -- loaded later -- would
    then run on that runtime instead of /opt/rocm's (the one bench.py, smoke() and every user of the library run on)."""
    def __init__(self, name):
        self._name, self._mod = name, None

    def __getattr__(self, attr):
        if attr.startswith("_"):                  # (pytest's collection probes every module-level object for __test__, pytestmark, ...)
            raise AttributeError(attr)
        if self._mod is None:
            import importlib
            self._mod = importlib.import_module(self._name)
        return getattr(self._mod, attr)


torch = _Lazy("torch")
dist = _Lazy("torch.distributed")
mp = _Lazy("torch.multiprocessing")

from bnmtf_amd.comm import shard_range
from oracle import bnmtf_oracle as O
from oracle import rng


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _problem():
    rs = np.random.RandomState(11)
    I, J, K = 23, 17, 4
    R = rs.exponential(1.0, (I, K)) @ rs.exponential(1.0, (J, K)).T + rs.normal(0, 1, (I, J))
    M = (rs.rand(I, J) > 0.25).astype(float); M[:, 0] = 1; M[0, :] = 1
    pri = dict(alpha=1., beta=1., lambdaU=0.3, lambdaV=0.6)
    U0 = rs.exponential(1.0, (I, K)); V0 = rs.exponential(1.0, (J, K))
    return R, M, K, pri, U0, V0


def _gather_blocks(local_block, n, cols, rank, world):
    """all-gather of ragged row blocks (what comm_allgather_factor does with ncclAllGather / grouped broadcasts)"""
    out = np.zeros((n, cols))
    bufs = [None] * world
    dist.all_gather_object(bufs, local_block)
    for r in range(world):
        f, c = shard_range(n, r, world)
        out[f:f + c] = bufs[r]
    return out


def _sharded_run(rank, world, port, iters, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    R, M, K, pri, U0, V0 = _problem()
    o = O.BNMFGibbsOracle(R, M, K, pri, seed=99)
    o.U, o.V = U0.copy(), V0.copy()
    o.tau = o.alpha_s() / o.beta_s()
    I, J = R.shape
    r0, nr = shard_range(I, rank, world); c0, nc = shard_range(J, rank, world)
    rows = np.arange(r0, r0 + nr); cols = np.arange(c0, c0 + nc)
    Mbar = 1.0 - M
    taus, mses = [], []
    for it in range(iters):
        for k in range(K):                       # this rank's rows only; counters use GLOBAL row indices
            t = o.tauU(k)[rows]; m = o.muU(o.tauU(k), k)[rows]
            o.U[rows, k] = rng.tn_draw(m, t, rows, k, it, rng.STREAM_ROWS, o.seed)
        o.U = _gather_blocks(o.U[rows], I, K, rank, world)
        for k in range(K):
            t = o.tauV(k)[cols]; m = o.muV(o.tauV(k), k)[cols]
            o.V[cols, k] = rng.tn_draw(m, t, cols, k, it, rng.STREAM_COLS, o.seed)
        Vloc = o.V[cols]
        o.V = _gather_blocks(Vloc, J, K, rank, world)
        # the three per-rank partial sums the cols sweep produces (over this rank's columns)
        Pv = (M * R).T[cols] @ o.U                                   # R~^T U, own columns
        qm = (Mbar * (o.U @ o.V.T))[:, cols]                         # q on the missing entries of own columns
        acc = torch.tensor([(Pv * Vloc).sum(), qm.sum(), (qm ** 2).sum()], dtype=torch.float64)
        dist.all_reduce(acc)
        srp, sq, sq2 = acc.tolist()
        spp = ((o.U.T @ o.U) * (o.V.T @ o.V)).sum() - sq2
        sse = (M * R * R).sum() - 2.0 * srp + spp
        o.tau = rng.gamma_draw(o.alpha_s(), o.beta + 0.5 * sse, it, o.seed)
        taus.append(o.tau); mses.append(sse / M.sum())
    if rank == 0:
        q.put((o.U, o.V, taus, mses))
    # every rank holds the same replicated state
    chk = torch.tensor([o.U.sum(), o.V.sum(), o.tau], dtype=torch.float64)
    lo = chk.clone(); hi = chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    assert torch.equal(lo, hi)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharded_sweep_reproduces_single_process_chain():
    iters = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_run, args=(r, 2, port, iters, q)) for r in range(2)]
    for p in procs: p.start()
    U, V, taus, mses = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60); assert p.exitcode == 0
    R, M, K, pri, U0, V0 = _problem()
    o = O.BNMFGibbsOracle(R, M, K, pri, seed=99)
    o.U, o.V = U0.copy(), V0.copy(); o.tau = o.alpha_s() / o.beta_s()
    o.run(iters)
    np.testing.assert_allclose(U, o.U, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(V, o.V, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(taus, o.all_tau, rtol=1e-9)
    np.testing.assert_allclose(mses, o.all_performances["MSE"], rtol=1e-9)


# ---------------------------------------------------------------------------------------------------------------
# BNMF VB and BNMTF Gibbs over ranks (api_models.inc): what is exchanged, checked with the oracle's arithmetic
# ---------------------------------------------------------------------------------------------------------------
def _vb_and_tri_run(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    R, M, K, pri, U0, V0 = _problem()
    I, J = R.shape
    r0, nr = shard_range(I, rank, world); c0, nc = shard_range(J, rank, world)
    rows = slice(r0, r0 + nr); cols = slice(c0, c0 + nc)
    # ---- BNMF VB: rank r updates q(U) of its rows, q(V) of its columns; (E, S2) blocks are gathered after each half
    # sweep; exp_square_diff (bnmf_vb_optimised.py:185-187) is the all-reduced sum of the ranks' masked pieces
    v = O.BNMFVBOracle(R, M, K, pri)
    v.initialise("exp")
    full = O.BNMFVBOracle(R, M, K, pri); full.initialise("exp")
    for it in range(3):
        for k in range(K):
            v.update_U(k); v.update_exp_U(k)         # (the oracle updates every row; a rank keeps its own block ...)
        for name in ("muU", "tauU", "expU", "varU"):
            blk = getattr(v, name)[rows].copy()
            setattr(v, name, _gather_blocks(blk, I, K, rank, world))          # ... and receives the others' blocks
        for k in range(K):
            v.update_V(k); v.update_exp_V(k)
        for name in ("muV", "tauV", "expV", "varV"):
            blk = getattr(v, name)[cols].copy()
            setattr(v, name, _gather_blocks(blk, J, K, rank, world))
        S2U, S2V = v.varU + v.expU ** 2, v.varV + v.expV ** 2
        piece = (M[rows] * ((R[rows] - v.expU[rows] @ v.expV.T) ** 2 + S2U[rows] @ S2V.T - (v.expU[rows] ** 2) @ (v.expV ** 2).T)).sum()
        t = torch.tensor([piece], dtype=torch.float64)
        dist.all_reduce(t)
        v.alpha_s = v.alpha + v.size_Omega / 2.0; v.beta_s = v.beta + 0.5 * float(t.item()); v.update_exp_tau()
        full.sweep()
        assert abs(v.exptau - full.exptau) < 1e-12 * full.exptau
        assert np.abs(v.expU - full.expU).max() < 1e-12 and np.abs(v.varV - full.varV).max() < 1e-12
    # ---- BNMTF: the S step's dense system.  A[(k,l),(k',l')] = sum_ij M_ij F_ik G_jl F_ik' G_jl' and b = sum_ij M_ij R_ij
    # F_ik G_jl are sums over columns j: a rank forms them over ITS columns, one all-reduce gives every rank the whole system
    rs = np.random.RandomState(5)
    Kt, Lt = 3, 4
    F = rs.exponential(1.0, (I, Kt)); S = rs.exponential(1.0, (Kt, Lt)); G = rs.exponential(1.0, (J, Lt))
    def system(jsl):
        A = np.zeros((Kt * Lt, Kt * Lt)); b = np.zeros(Kt * Lt)
        for j in range(*jsl.indices(J)):
            W = (F * M[:, j:j + 1]).T @ F                          # W_j[k][k'] = sum_i M_ij F_ik F_ik'
            A += np.kron(W, np.outer(G[j], G[j]))
            b += np.kron((M[:, j] * R[:, j]) @ F, G[j])
        return A, b
    A_loc, b_loc = system(cols)
    t = torch.tensor(np.concatenate([A_loc.ravel(), b_loc]), dtype=torch.float64)
    dist.all_reduce(t)                                             # the "K x L Gram" exchange: one all-reduce of (A, b)
    A = t.numpy()[:A_loc.size].reshape(A_loc.shape); b = t.numpy()[A_loc.size:]
    A_full, b_full = system(slice(0, J))
    assert np.allclose(A, A_full, rtol=1e-12) and np.allclose(b, b_full, rtol=1e-12)
    # the coordinate step on (A, b) is the reference's tauS / muS (bnmtf_gibbs_optimised.py:201-205)
    tri = O.BNMTFGibbsOracle(R, M, Kt, Lt, dict(alpha=1., beta=1., lambdaF=0.3, lambdaS=0.2, lambdaG=0.6))
    tri.F, tri.S, tri.G, tri.tau = F, S.copy(), G, 0.7
    s = S.ravel()
    for (k, l) in [(0, 0), (1, 2), (2, 3)]:
        a = k * Lt + l
        tauS = tri.tau * A[a, a]
        muS = (-0.2 + tri.tau * (b[a] - A[a] @ s + A[a, a] * s[a])) / tauS
        assert abs(tauS - tri.tauS(k, l)) < 1e-10 * tauS and abs(muS - tri.muS(tri.tauS(k, l), k, l)) < 1e-9 * (abs(muS) + 1)
    q.put((rank, True))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_vb_and_tri_factorisation_exchanges():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_vb_and_tri_run, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(q.get(timeout=5)[0] for _ in range(2)) == [0, 1]
