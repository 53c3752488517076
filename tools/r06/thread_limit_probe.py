"""bnmtf_create's layout passes when no more threads can be had (RLIMIT_NPROC at the floor): the model must build and run."""
import os, resource, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bnmtf_amd import bnmf_gibbs_optimised
from bnmtf_amd.synthetic import generate_bnmf
pri = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)
R, M, _, _ = generate_bnmf(1500, 1200, 16, 0.1, seed_data=1, seed_mask=2)
a = bnmf_gibbs_optimised(R, M, 16, pri, verbose=False, seed=1); a.initialise("random"); a.run(2)      # HIP runtime up, its threads made
soft, hard = resource.getrlimit(resource.RLIMIT_NPROC)
resource.setrlimit(resource.RLIMIT_NPROC, (1, hard))
print("uid", os.getuid(), "RLIMIT_NPROC", soft, hard, "-> 1", flush=True)
b = bnmf_gibbs_optimised(R, M, 16, pri, verbose=False, seed=1); b.initialise("random"); b.run(2)
resource.setrlimit(resource.RLIMIT_NPROC, (soft, hard))
print("built and ran with no threads to be had: MSE", b.all_performances["MSE"][-1], a.all_performances["MSE"][-1], flush=True)
