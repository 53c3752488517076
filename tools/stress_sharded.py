"""Is the sharded chain (in-process transport, one GPU) the same every time?  python tools/stress_sharded.py [runs] [world] [update]
Repeats the 515 x 389, K=40 case of tests/test_sharded_gpu.py and reports where a run first leaves the single-rank chain."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from bnmtf_amd import bnmf_gibbs_optimised
from bnmtf_amd.synthetic import generate_bnmf
import test_sharded_gpu as T

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
world = int(sys.argv[2]) if len(sys.argv) > 2 else 3
update = sys.argv[3] if len(sys.argv) > 3 else "mode"
I, J, K = 515, 389, 40
R, M, _, _ = generate_bnmf(I, J, K, 0.12, seed_data=5, seed_mask=6)
rs = np.random.RandomState(3)
U0 = rs.exponential(1.0, (I, K)); V0 = rs.exponential(1.0, (J, K)); tau0 = 0.7
single = bnmf_gibbs_optimised(R, M, K, T.PRI, verbose=False, seed=7)
single.U, single.V, single.tau = U0.copy(), V0.copy(), tau0
single.run(5, update=update)
sU, sV = single.all_U, single.all_V
bad = 0
for n in range(runs):
    ranks = T._run_ranks(R, M, K, U0, V0, tau0, world, 5, update, ("s%d" % n).encode())
    msg = []
    for it in range(5):
        for name, a, b in (("U", ranks[0][0][it], sU[it]), ("V", ranks[0][1][it], sV[it])):
            if not np.array_equal(a, b):
                d = np.argwhere(a != b)
                msg.append("it %d %s: %d entries differ, rows %s..%s cols %s, max |d| %.3g" % (
                    it, name, len(d), d[:, 0].min(), d[:, 0].max(), sorted(set(d[:, 1].tolist()))[:6], np.abs(a - b).max()))
                break
        if msg:
            break
    same_ranks = all(np.array_equal(x, y) for r in range(1, world) for x, y in zip(ranks[0], ranks[r]))
    if msg or not same_ranks:
        bad += 1
        print("run %d: %s%s" % (n, msg[0] if msg else "", "" if same_ranks else "  (ranks disagree with each other)"))
print("bad %d of %d (world %d, %s)" % (bad, runs, world, update))
