import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from bnmtf_amd import bnmf_gibbs_optimised, bnmf_vb_optimised
from bnmtf_amd.synthetic import generate_bnmf
R, M, _, _ = generate_bnmf(8192, 8192, 64, 0.1, tau=1.0, seed_data=0, seed_mask=1)
pri = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)
# determinism of the new paths: the same run twice, bit for bit (Gibbs: same seed; VB: deterministic)
res = []
for rep in range(2):
    a = bnmf_gibbs_optimised(R, M, 64, pri, verbose=False, seed=3)
    np.random.seed(0); a.initialise("random"); a.run(300, store_samples=False)
    res.append((np.array(a.all_performances["MSE"]), a.U.copy(), a.V.copy()))
    a.close()
print("gibbs 300 its twice: identical MSE trajectory %s, identical U %s V %s; MSE it300 %.4f" % (np.array_equal(res[0][0], res[1][0]), np.array_equal(res[0][1], res[1][1]), np.array_equal(res[0][2], res[1][2]), res[0][0][-1]))
vres = []
for rep in range(2):
    b = bnmf_vb_optimised(R, M, 64, pri, verbose=False)
    b.initialise("exp"); b.run(150)
    vres.append((np.array(b.all_performances["MSE"]), np.array(b.all_elbo), b.expU.copy(), b.tauV.copy()))
    print(b.describe().split()[-1])
    b.close()
print("vb 150 its twice: identical MSE %s ELBO %s expU %s tauV %s; MSE %.4f -> %.4f; ELBO finite tail %s" % (np.array_equal(vres[0][0], vres[1][0]), np.array_equal(vres[0][1], vres[1][1], equal_nan=True), np.array_equal(vres[0][2], vres[1][2]), np.array_equal(vres[0][3], vres[1][3]), vres[0][0][0], vres[0][0][-1], np.isfinite(vres[0][1][-5:]).all()))
