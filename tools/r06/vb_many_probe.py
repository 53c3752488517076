import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bnmtf_amd import bnmf_vb_optimised, run_many
from bnmtf_amd.synthetic import generate_bnmf
I, J = 622, 138
pri = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)
def build(n, Ks):
    ms = []
    for i in range(n):
        K = Ks[i % len(Ks)]
        R, M, _, _ = generate_bnmf(I, J, 10, 0.19, seed_data=1, seed_mask=2 + i)
        np.random.seed(100 + i)
        b = bnmf_vb_optimised(R, M, K, pri, verbose=False)
        b.initialise("random")
        ms.append(b)
    return ms
for n, Ks in ((3, [25]), (8, [15, 20, 25, 30])):
    a = build(n, Ks); b = build(n, Ks)
    for m in a: m.run(20)
    run_many(b, 20)
    ok = True
    for x, y in zip(a, b):
        for name in ("muU", "tauU", "expU", "varU", "muV", "tauV", "expV", "varV"):
            if not np.array_equal(getattr(x, name), getattr(y, name)): ok = False; print("differs", name, np.abs(getattr(x, name) - getattr(y, name)).max())
        if x.all_exp_tau != y.all_exp_tau or x.all_performances != y.all_performances: ok = False; print("records differ")
    print(n, Ks, "identical" if ok else "DIFFERENT", b[0]._many_info)
ms = build(40, [15, 20, 25, 30])
for m in ms: m.run(3)
t0 = time.perf_counter(); run_many(ms, 500); dt = time.perf_counter() - t0
print("40 models x 500 iterations: %.3f s = %.1f us per model-iteration, info %s" % (dt, dt / (40 * 500) * 1e6, ms[0]._many_info))
t0 = time.perf_counter()
for m in ms[:8]: m.run(500)
dt = time.perf_counter() - t0
print("one by one: %.1f us per model-iteration" % (dt / (8 * 500) * 1e6))
