"""Phase stamps of the small-model kernel (tools/variant.sh small_timing kernel_small.hip -DBNMTF_SMALL_TIMING; BNMTF_LIB=tools/lib_small_timing.so)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bnmtf_amd
from bnmtf_amd.synthetic import generate_bnmf
PRI = dict(alpha=1.0, beta=1.0, lambdaU=0.1, lambdaV=0.1)
for (I, J, K, miss) in [(622, 138, 25, 0.19)]:
    R, M, _, _ = generate_bnmf(I, J, K, miss, seed_data=3, seed_mask=4)
    np.random.seed(1)
    b = bnmtf_amd.bnmf_gibbs_optimised(R, M, K, PRI, seed=5, verbose=False)
    b.initialise('random')
    b.set_small_path('always')
    b.run(50, store_samples=False)
    print("== %dx%d K=%d miss %.2f" % (I, J, K, miss), flush=True)
    b.run(100, store_samples=False)
    print("   device clock %.1f us/it" % (1e6 * b.all_times[-1] / 100), flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "tri":
    from bnmtf_amd.synthetic import generate_bnmtf
    PT = dict(alpha=1., beta=1., lambdaF=0.1, lambdaS=0.1, lambdaG=0.1)
    for (I, J, K, L, miss) in [(622, 138, 10, 10, 0.19), (100, 80, 5, 5, 0.1)]:
        R, M, _, _, _ = generate_bnmtf(I, J, K, L, miss, seed_data=1, seed_mask=2)
        np.random.seed(1)
        b = bnmtf_amd.bnmtf_gibbs_optimised(R, M, K, L, PT, seed=3, verbose=False)
        b.initialise('random', 'random')
        b.set_small_path('always')
        b.run(20, store_samples=False)
        print("== tri %dx%d K=%d L=%d miss %.2f  %s" % (I, J, K, L, miss, b.describe()[-120:]), flush=True)
        b.run(100, store_samples=False)
        print("   device clock %.1f us/it" % (1e6 * b.all_times[-1] / 100), flush=True)
