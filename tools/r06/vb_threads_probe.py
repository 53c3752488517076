"""N GDSC-shaped bnmf_vb models, one host thread and one stream each, in ONE process: how far do their kernels share the GPU?
(The replica pool's slots are processes.)   [GRAPH=1] [GPU_MAX_HW_QUEUES=n] python tools/r06/vb_threads_probe.py"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bnmtf_amd import bnmf_vb_optimised
from bnmtf_amd.synthetic import generate_bnmf
I, J, K = 622, 138, 25
R, M, _, _ = generate_bnmf(I, J, K, 0.19, seed_data=1, seed_mask=2)
pri = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)
models = []
for n in range(16):
    b = bnmf_vb_optimised(R, M, K, pri, verbose=False)
    b.initialise("exp")
    if os.environ.get("GRAPH") == "1":
        b.set_graph_replay(True)
    b.run(5)
    models.append(b)
for nt in (1, 2, 4, 8, 16):
    ts = [threading.Thread(target=lambda b=b: b.run(500)) for b in models[:nt]]
    t0 = time.perf_counter()
    [t.start() for t in ts]; [t.join() for t in ts]
    dt = time.perf_counter() - t0
    print("%2d threads: %.0f ms for %d x 500 iterations = %.1f us per model-iteration" % (nt, dt * 1e3, nt, dt / (nt * 500) * 1e6))
