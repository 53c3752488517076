"""Worker of tests/test_rccl_two_process_gpu.py: one rank of a row/column-sharded BNMF Gibbs run over REAL RCCL (one
process per GPU, started by bnmtf_amd.comm.spawn_local).  Writes its replicated chain to <out>.rank<r>.npz."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    out = sys.argv[1]
    from bnmtf_amd import bnmf_gibbs_optimised, bnmf_vb_optimised, comm
    from bnmtf_amd.synthetic import generate_bnmf
    rank, world, local_rank, cid, cp = comm.init_from_env()
    I, J, K = 640, 512, 24
    R, M, _, _ = generate_bnmf(I, J, K, 0.12, seed_data=5, seed_mask=6)
    rs = np.random.RandomState(3)
    U0 = rs.exponential(1.0, (I, K)); V0 = rs.exponential(1.0, (J, K))
    pri = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)
    res = {}
    for update in ("mode", "draw"):
        b = bnmf_gibbs_optimised(R, M, K, pri, verbose=False, seed=7, device=local_rank, rank=rank, world=world, comm_id=cid)
        b.U, b.V, b.tau = U0.copy(), V0.copy(), 0.7
        b.run(5, update=update)
        res[update + "_U"], res[update + "_V"], res[update + "_tau"] = b.all_U.copy(), b.all_V.copy(), b.all_tau.copy()
        res[update + "_mse"] = np.array(b.all_performances["MSE"])
        b.close()
        cp.barrier()
    v = bnmf_vb_optimised(R, M, K, pri, verbose=False, device=local_rank, rank=rank, world=world, comm_id=cid)
    v.initialise("exp")
    v.run(6)
    res["vb_mse"] = np.array(v.all_performances["MSE"]); res["vb_exptau"] = np.array(v.all_exp_tau)
    v.close()
    np.savez(out + ".rank%d.npz" % rank, **res)
    cp.barrier()
    cp.close()


if __name__ == "__main__":
    main()
