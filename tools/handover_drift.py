"""q hand-over between the half sweeps: does q handed back and forth drift from U.V?  The sum-of-squares identity behind the reported
MSE takes sum q and sum q^2 of the missing entries from the sweep's registers: reported MSE against the fp64 MSE of the final (U, V)
on the host, with the pre-pass every iteration (BNMTF_HANDOVER=0) and with hand-over at several refresh intervals."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bnmtf_amd
from bnmtf_amd.synthetic import generate_bnmf
I = J = int(os.environ.get("N", "8192")); K = 64
R, M, _, _ = generate_bnmf(I, J, K, 0.1, seed_data=3, seed_mask=4)
pri = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)
def run(env, n):
    for k in ("BNMTF_HANDOVER", "BNMTF_HANDOVER_REFRESH"): os.environ.pop(k, None)
    os.environ.update(env)
    b = bnmtf_amd.bnmf_gibbs_optimised(R, M, K, pri, verbose=False, seed=11)
    np.random.seed(2); b.initialise('random')
    b.run(n, store_samples=False)
    U, V = np.asarray(b.U, dtype=np.float64), np.asarray(b.V, dtype=np.float64)
    mse = float((M * (R - U @ V.T) ** 2).sum() / M.sum())
    return b.all_performances["MSE"][-1], mse
for n in (16, 200, 1000):
    for env in ({"BNMTF_HANDOVER": "0"}, {"BNMTF_HANDOVER_REFRESH": "8"}, {"BNMTF_HANDOVER_REFRESH": "64"}, {"BNMTF_HANDOVER_REFRESH": "1000000"}):
        rep, exact = run(env, n)
        print("n", n, env, "reported MSE %.9g  fp64 MSE of (U, V) %.9g  rel %.3g" % (rep, exact, abs(rep - exact) / exact), flush=True)
