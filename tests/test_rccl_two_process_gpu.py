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
