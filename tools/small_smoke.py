"""First checks of the one-launch path for small models (csrc/kernel_small.hip) against the oracle and the multi-launch path."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bnmtf_amd
from bnmtf_amd.synthetic import generate_bnmf
from oracle import bnmtf_oracle as O

PRI = dict(alpha=1.0, beta=1.0, lambdaU=0.1, lambdaV=0.1)

def case(I, J, K, miss, iters=8):
    R, M, _, _ = generate_bnmf(I, J, K, miss, seed_data=3, seed_mask=4)
    np.random.seed(1)
    b = bnmtf_amd.bnmf_gibbs_optimised(R, M, K, PRI, seed=5, verbose=False)
    b.initialise('random')
    b.set_small_path('always')
    print(b.describe()[-90:], 'small:', b.is_small())
    o = O.BNMFGibbsOracle(R.astype(np.float64), M, K, PRI, seed=5)
    o.U, o.V, o.tau = b.U.copy(), b.V.copy(), b.tau
    U0, V0, t0 = b.U.copy(), b.V.copy(), b.tau
    o.run(iters, draw=False)
    b.run(iters, update='mode')
    mse_b, mse_o = np.array(b.all_performances['MSE']), np.array(o.all_performances['MSE'])
    print("%dx%d K=%d miss=%.2f  mode: MSE rel err %.2e  tau rel err %.2e  U err it1 %.2e it%d %.2e" % (
        I, J, K, miss, np.abs(mse_b / mse_o - 1).max(), np.abs(b.all_tau / o.all_tau - 1).max(),
        np.abs(b.all_U[0] - o.all_U[0]).max(), iters, np.abs(b.all_U[-1] - o.all_U[-1]).max()))
    # draws: same chain as the oracle's sampler
    o2 = O.BNMFGibbsOracle(R.astype(np.float64), M, K, PRI, seed=5)
    o2.U, o2.V, o2.tau = U0.copy(), V0.copy(), t0
    o2.run(5)
    b.U, b.V, b.tau = U0.copy(), V0.copy(), t0
    import ctypes as C
    from bnmtf_amd import _lib
    _lib.check(_lib.lib().bnmtf_set_iteration(b._handle(), 0))
    b.run(5)
    d0 = np.abs(b.all_U[0] - o2.all_U[0]) / (1e-3 + np.abs(o2.all_U[0]))
    print("   draws: frac within 1e-3 first sweep %.4f, MSE %s vs %s" % (np.mean(d0 < 1e-3), np.round(b.all_performances['MSE'][:3], 4), np.round(o2.all_performances['MSE'][:3], 4)))
    p = b.predict_while_running()
    print("   identity MSE %.6f direct %.6f" % (b.all_performances['MSE'][-1], p['MSE']))
    # the multi-launch path from the same state
    b2 = bnmtf_amd.bnmf_gibbs_optimised(R, M, K, PRI, seed=5, verbose=False)
    b2.U, b2.V, b2.tau = U0.copy(), V0.copy(), t0
    b2.set_small_path(False)
    b2.run(5)
    print("   vs multi-launch path: U it1 max diff %.2e  MSE %s" % (np.abs(b2.all_U[0] - b.all_U[0]).max(), np.round(b2.all_performances['MSE'][:3], 4)))
    t = time.perf_counter(); b.run(200); dt = time.perf_counter() - t
    print("   200 iterations: %.1f us/it wall, %.1f us/it device clock" % (1e6 * dt / 200, 1e6 * b.all_times[-1] / 200))

for args in [(100, 80, 10, 0.1), (100, 80, 10, 0.0), (37, 29, 5, 0.2), (300, 200, 8, 0.1), (622, 138, 25, 0.19), (512, 512, 32, 0.05)]:
    case(*args)
