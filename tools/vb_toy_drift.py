"""Drift of the device VB trajectory from the reference's on the toy case (tests/golden/bnmf_vb.npz), per path:
BNMTF_VB_PATH=pairs (pair-panel kernel) against BNMTF_VB_PATH=masked (the on-chip kernel with the masked sums from
kernel_maskgemm.hip): 1.5e-3 / 1.2e-3 on expU (three bf16 planes: 6.7e-4; two: 2.0e-3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bnmtf_amd import bnmf_vb_optimised
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "bnmf_vb.npz")))
t = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "toy_data.npz")))
R, M = t["bnmf/R"], t["bnmf/M"]
I, J = R.shape; K = 10
b = bnmf_vb_optimised(R, M, K, dict(alpha=1., beta=1., lambdaU=0.1 * np.ones((I, K)), lambdaV=0.1 * np.ones((J, K))), verbose=False)
b.initialise('exp')
b.run(20)
out = []
for nm in ["expU", "expV", "muU", "muV", "tauU", "tauV"]:
    ref = g["toy/it20/" + nm]
    out.append("%s %.2e" % (nm, np.abs(getattr(b, nm) - ref).max() / np.abs(ref).max()))
print(os.environ.get("BNMTF_VB_PATH", "auto"), os.environ.get("BNMTF_WIDE", "-"), " ".join(out),
      "mse %.2e elbo %.2e" % (np.abs(b.all_performances['MSE'] / g["toy/mse"] - 1).max(), np.abs(np.array(b.all_elbo) / g["toy/elbo"] - 1).max()))
print(b.describe())
