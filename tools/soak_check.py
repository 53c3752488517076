"""Soak check at the headline shape: 1 500 Gibbs iterations reach the noise floor (MSE -> 1/tau = 1, tau -> 1), with a VB
handle alive and running beside the Gibbs one."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bnmtf_amd import bnmf_gibbs_optimised, bnmf_vb_optimised, bnmtf_gibbs_optimised
from bnmtf_amd.synthetic import generate_bnmf
R, M, _, _ = generate_bnmf(8192, 8192, 64, 0.1, tau=1.0, seed_data=0, seed_mask=1)
pri = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)
a = bnmf_gibbs_optimised(R, M, 64, pri, verbose=False, seed=0)
b = bnmf_vb_optimised(R, M, 64, pri, verbose=False)          # a second handle alive at the same time
np.random.seed(0); a.initialise("random"); b.initialise("exp")
t0 = time.time(); a.run(1500, store_samples=False); t1 = time.time()
b.run(30)
mse = np.array(a.all_performances["MSE"])
print("gibbs 1500 its %.2fs  MSE first %.3f  it300 %.4f  it1500 %.4f  tau %.4f  finite %s" % (t1 - t0, mse[0], mse[299], mse[-1], a.all_tau[-1], np.isfinite(mse).all()))
print("vb 30 its MSE %.4f -> %.4f" % (b.all_performances["MSE"][0], b.all_performances["MSE"][-1]))
a.run(200, store_samples=False)                                # a handle keeps working after another one ran
print("gibbs +200: MSE %.4f" % a.all_performances["MSE"][-1])
