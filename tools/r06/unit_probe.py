import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from bnmtf_amd import bnmf_gibbs_optimised
from bnmtf_amd.synthetic import generate_bnmf
rows = int(sys.argv[1])
R, M, _, _ = generate_bnmf(rows, 8192, 64, 0.1, tau=1.0, seed_data=0, seed_mask=1)
s = bnmf_gibbs_optimised(R, M, 64, dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1), verbose=False, seed=7)
s.initialise("random")
s.run(3, store_samples=False)
s.close()
