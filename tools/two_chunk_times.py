#!/usr/bin/env python3
"""Sweep / contraction times of a problem whose inner extents need two LDS panels (9 185 .. ~18 000), with the two-chunk
on-chip kernel and with the generic fall-back (BNMTF_NO_CHUNKS=1):  python tools/two_chunk_times.py [N] [K]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bnmtf_amd
from bnmtf_amd import _lib
from bnmtf_amd.synthetic import generate_bnmf

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
R, M, _, _ = generate_bnmf(N, N, K, 0.1, tau=1.0, seed_data=0, seed_mask=1)
pri = dict(alpha=1.0, beta=1.0, lambdaU=0.1, lambdaV=0.1)
for env in ("0", "1"):
    if env == "1":
        os.environ["BNMTF_NO_CHUNKS"] = "1"
    np.random.seed(0)
    b = bnmtf_amd.bnmf_gibbs_optimised(R, M, K, pri, seed=0, verbose=False)
    b.initialise("random")
    b.run(3, store_samples=False)
    b.set_profiling(True)
    t0 = time.perf_counter()
    n = 10 if env == "0" else 3
    b.run(n, store_samples=False)
    dt = (time.perf_counter() - t0) / n
    names = {_lib.KERNEL_GEMM_ROWS: "gemm_rows", _lib.KERNEL_GEMM_COLS: "gemm_cols", _lib.KERNEL_SWEEP_ROWS: "sweep_rows", _lib.KERNEL_SWEEP_COLS: "sweep_cols"}
    st = {nm: round(1e3 * b.kernel_stats(k)[0] / max(b.kernel_stats(k)[1], 1), 1) for k, nm in names.items()}
    print("no_chunks=%s  %dx%d K=%d: %.2f ms/iteration  kernels(us): %s  MSE %.4g  %s" % (env, N, N, K, 1e3 * dt, st, b.all_performances["MSE"][-1], b.describe().split("rows[")[1][:110]))
    b.close()
