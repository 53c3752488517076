"""Two processes, two GPUs, REAL RCCL (ncclCommInitRank with world = 2, ncclAllGather of the factor blocks, ncclAllReduce of
the sums): the row/column-sharded BNMF Gibbs and VB runs end with the same replicated chain on both ranks, and it is the
single-GPU chain.  Needs two visible devices; on a one-GPU box the test is skipped (the same code path runs there through
the in-process transport, tests/test_sharded_gpu.py, and with a one-rank RCCL communicator, BNMTF_FORCE_COMM)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from bnmtf_amd import bnmf_gibbs_optimised, bnmf_vb_optimised, device_count
from bnmtf_amd.synthetic import generate_bnmf

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_ranks_over_rccl_draw_the_single_gpu_chain(tmp_path):
    if device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    out = str(tmp_path / "chain")
    # a child starts the two ranks (bnmtf_amd.comm.spawn_local): this process has touched the GPU already
    code = "import sys; sys.path.insert(0, %r); from bnmtf_amd import comm; sys.exit(comm.spawn_local(2, [%r, %r]))" % (
        os.path.dirname(HERE), os.path.join(HERE, "rccl_worker.py"), out)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", code], env=env, timeout=600, capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    r0, r1 = np.load(out + ".rank0.npz"), np.load(out + ".rank1.npz")
    for k in r0.files:
        assert np.array_equal(r0[k], r1[k]), k            # replicated state: the same bits on both ranks
    I, J, K = 640, 512, 24
    R, M, _, _ = generate_bnmf(I, J, K, 0.12, seed_data=5, seed_mask=6)
    rs = np.random.RandomState(3)
    U0 = rs.exponential(1.0, (I, K)); V0 = rs.exponential(1.0, (J, K))
    pri = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)
    for update in ("mode", "draw"):
        s = bnmf_gibbs_optimised(R, M, K, pri, verbose=False, seed=7)
        s.U, s.V, s.tau = U0.copy(), V0.copy(), 0.7
        s.run(5, update=update)
        assert np.array_equal(r0[update + "_U"][0], s.all_U[0])          # first sweep: same operation order in every kernel
        np.testing.assert_allclose(r0[update + "_tau"], s.all_tau, rtol=1e-9)
        np.testing.assert_allclose(r0[update + "_mse"], s.all_performances["MSE"], rtol=1e-6)
        assert np.abs(r0[update + "_U"][-1] - s.all_U[-1]).max() <= 1e-4 * np.abs(s.all_U[-1]).max()
        s.close()
    v = bnmf_vb_optimised(R, M, K, pri, verbose=False)
    v.initialise("exp"); v.run(6)
    np.testing.assert_allclose(r0["vb_mse"], v.all_performances["MSE"], rtol=1e-6)
    np.testing.assert_allclose(r0["vb_exptau"], v.all_exp_tau, rtol=1e-6)


def test_every_sharded_model_family_runs_its_exchanges_through_a_one_rank_rccl_communicator(monkeypatch):
    """One GPU: BNMTF_FORCE_COMM gives a model a 1-rank RCCL communicator, so its run() takes the multi-GPU code path -- the dlopen'd
    RCCL calls (in-place all-gathers of factor blocks and moments, the S system's all-reduce, the exchanged sums) on the library's
    streams -- with collectives that move nothing.  bnmf_gibbs has had this since round 2 (tests/test_bnmf_gibbs_gpu.py); here
    the other families: bnmf_vb, bnmtf_gibbs, bnmtf_vb (sharded since round 6).  The trajectories must be the unsharded ones up
    to the order of the exchanged sums."""
    from bnmtf_amd import bnmf_vb_optimised, bnmtf_gibbs_optimised, bnmtf_vb_optimised
    from bnmtf_amd.synthetic import generate_bnmf, generate_bnmtf
    I, J, K, L = 300, 260, 12, 9
    R2, M2, _, _ = generate_bnmf(I, J, K, 0.12, seed_data=1, seed_mask=2)
    R3, M3, _, _, _ = generate_bnmtf(I, J, K, L, 0.12, seed_data=3, seed_mask=4)
    pri2 = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)
    pri3 = dict(alpha=1., beta=1., lambdaF=0.1, lambdaS=0.1, lambdaG=0.1)
    rs = np.random.RandomState(0)
    F0 = rs.exponential(1.0, (I, K)); S0 = rs.exponential(1.0, (K, L)); G0 = rs.exponential(1.0, (J, L))
    orders = np.array([np.concatenate([rs.permutation(K * L), rs.permutation(K), rs.permutation(L)]) for _ in range(4)], dtype=np.int32)
    out = {}
    for force in (False, True):
        if force:
            monkeypatch.setenv("BNMTF_FORCE_COMM", "1")
        v = bnmf_vb_optimised(R2, M2, K, pri2, verbose=False)
        v.initialise("exp"); v.run(5)
        g = bnmtf_gibbs_optimised(R3, M3, K, L, pri3, verbose=False, seed=3)
        g.set_small_path(False)
        g.F, g.S, g.G, g.tau = F0.copy(), S0.copy(), G0.copy(), 0.8
        g.run(4, update="mode")
        t = bnmtf_vb_optimised(R3, M3, K, L, pri3, verbose=False)
        np.random.seed(2); t.initialise("random", "random")
        t.run(4, orders=orders)
        out[force] = (np.array(v.all_exp_tau), v.expU.copy(), np.array(g.all_tau), g.all_S[-1].copy(), np.array(t.all_exp_tau), t.expS.copy(), t.expF.copy())
        for m in (v, g, t):
            m.close()
    for a, b in zip(out[False], out[True]):
        assert np.abs(a - b).max() <= 2e-4 * np.abs(a).max()
