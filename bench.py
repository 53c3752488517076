#!/usr/bin/env python3
"""Headline benchmark: Gibbs iterations/sec of BNMF on a synthetic I=J=8192, K=64 matrix
with a 10 % missing mask (BASELINE.json metric / configs[2]; it fits one GPU).

    python bench.py --gpus N --steps K --warmup W

One process per GPU (for N > 1 launched by torch.distributed.run; rows of R are split
over the ranks for the U sweep, columns for the V sweep, factor blocks exchanged with
RCCL inside libbnmtf_hip.so).  A step is one full Gibbs iteration exactly as the
reference's run() defines it (bnmf_gibbs_optimised.py:133-155): K column updates of U,
K of V, the tau draw and the three training-mask metrics; inputs are resident in HBM
when the timed region starts, samples stay on the device (PCIe-inclusive rate: DESIGN.md).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak (the f32 kernel, BNMTF_GEMM=f32)
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA peak (the contraction takes 6 bf16 products per fp32 product)
PEAK_HBM_GBS = 8000.0

WORKLOADS = {
    "bnmf_8192_k64": dict(I=8192, J=8192, K=64),          # BASELINE.json metric config (headline)
    "bnmf_4096_k32": dict(I=4096, J=4096, K=32),          # configs[1]
    "bnmf_1024_k16": dict(I=1024, J=1024, K=16),
    "bnmtf_4096_k32": dict(I=4096, J=4096, K=32, L=32),   # configs[3]
    "vb_8192_k64": dict(I=8192, J=8192, K=64, vb=True),   # configs[4]
    "vb_4096_k32": dict(I=4096, J=4096, K=32, vb=True),
}


def cpu_baseline(R, M, K, pri, seed):
    """The oracle (NumPy restatement of the reference, fp64, as written) on a bounded
    sample of the same workload: ten U-column and ten V-column updates (tau*, mu*, draws),
    beta_s + tau draw, and the metrics, at full size; an iteration is K of each."""
    from oracle import bnmtf_oracle as O
    from oracle import rng as orng
    try:
        import threadpoolctl
        cores = max(i["num_threads"] for i in threadpoolctl.threadpool_info() if i.get("user_api") == "blas")
    except Exception:
        cores = os.cpu_count()
    o = O.BNMFGibbsOracle(R, M, K, pri, seed=seed)
    np.random.seed(0)
    o.U = np.random.exponential(10.0, (o.I, K)); o.V = np.random.exponential(10.0, (o.J, K)); o.tau = 1.0
    ncol = min(K, 10)                 # a bounded sample: ncol of the K column updates of each factor (~10-20 s of CPU work)
    t0 = time.perf_counter()
    for k in range(ncol):
        t = o.tauU(k); m = o.muU(t, k); o.U[:, k] = orng.tn_draw(m, t, np.arange(o.I), k, 0, orng.STREAM_ROWS, seed)
    t1 = time.perf_counter()
    for k in range(ncol):
        t = o.tauV(k); m = o.muV(t, k); o.V[:, k] = orng.tn_draw(m, t, np.arange(o.J), k, 0, orng.STREAM_COLS, seed)
    t2 = time.perf_counter()
    o.tau = orng.gamma_draw(o.alpha_s(), o.beta_s(), 0, seed)
    o.predict_while_running()
    t3 = time.perf_counter()
    sec_per_iter = K * ((t1 - t0) + (t2 - t1)) / ncol + (t3 - t2)
    return {"value": 1.0 / sec_per_iter, "unit": "Gibbs iterations/s", "cores": int(cores), "kind": "port",
            "sample": "oracle/bnmtf_oracle.py (NumPy fp64, as written): %d of %d U-column updates %.2fs, %d of %d V-column "
                      "updates %.2fs, tau+metrics %.2fs at full size; iteration = %d/%d*(U+V)+tail = %.1fs"
                      % (ncol, K, t1 - t0, ncol, K, t2 - t1, t3 - t2, K, ncol, sec_per_iter)}

def side_workload(a, w, rank, world):
    """BNMTF Gibbs / BNMF VB timing lines (single GPU; not the headline metric)."""
    import bnmtf_amd
    from bnmtf_amd import _lib
    from bnmtf_amd.synthetic import generate_bnmf, generate_bnmtf
    assert world == 1, "BNMTF / VB workloads are single-GPU"
    I, J, K = w["I"], w["J"], w["K"]
    L = _lib.lib()
    np.random.seed(0)
    if w.get("vb"):
        R, M, _, _ = generate_bnmf(I, J, K, 0.1, seed_data=0, seed_mask=1)
        m = bnmtf_amd.bnmf_vb_optimised(R, M, K, dict(alpha=1.0, beta=1.0, lambdaU=0.1, lambdaV=0.1), verbose=False)
        m.initialise("exp")
        m._push()
        run = lambda n, perf=None: _lib.check(L.bnmf_vb_run(m._handle(), n, None, _lib.ptr(perf), None, None))
        name = "BNMF VB iterations/sec"
    else:
        R, M, _, _, _ = generate_bnmtf(I, J, K, w["L"], 0.1, seed_data=0, seed_mask=1)
        m = bnmtf_amd.bnmtf_gibbs_optimised(R, M, K, w["L"], dict(alpha=1.0, beta=1.0, lambdaF=0.1, lambdaS=0.1, lambdaG=0.1), seed=0, verbose=False)
        m.initialise("random", "random")
        m._push()
        run = lambda n, perf=None: _lib.check(L.bnmtf_gibbs_run(m._handle(), n, 0, None, None, None, None, _lib.ptr(perf), None))
        name = "BNMTF Gibbs iterations/sec"
    run(a.warmup)
    perf = np.zeros((a.steps, 3))
    _lib.check(L.bnmtf_sync(m._handle()))
    t0 = time.perf_counter()
    run(a.steps, perf)
    _lib.check(L.bnmtf_sync(m._handle()))
    dt = time.perf_counter() - t0
    print(json.dumps({"metric": "%s (%s)" % (name, a.workload), "value": a.steps / dt, "unit": "iterations/s", "n_gpus": 1,
                      "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True,
                      "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                      "config": {"workload": a.workload}, "mse_first_last": [float(perf[0, 0]), float(perf[-1, 0])]}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="bnmf_8192_k64", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--with-samples", action="store_true", help="also time the PCIe sample hand-off (all_U/all_V)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`" % (a.gpus, a.gpus))
        a.gpus = world

    import bnmtf_amd
    from bnmtf_amd import _lib
    from bnmtf_amd.synthetic import generate_bnmf

    dist = None
    comm_id = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="gloo")         # control plane only; the data path is RCCL in the library
        ids = [None]
        if rank == 0:
            buf = np.zeros(128, dtype=np.uint8)
            _lib.check(_lib.lib().bnmtf_comm_unique_id(_lib.ptr(buf)))
            ids = [bytes(buf)]
        dist.broadcast_object_list(ids, src=0)
        comm_id = ids[0]

    w = WORKLOADS[a.workload]
    I, J, K = w["I"], w["J"], w["K"]
    if "L" in w or w.get("vb"):
        return side_workload(a, w, rank, world)
    R, M, _, _ = generate_bnmf(I, J, K, 0.1, tau=1.0, seed_data=0, seed_mask=1)
    pri = dict(alpha=1.0, beta=1.0, lambdaU=0.1, lambdaV=0.1)

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(R.astype(np.float64), M.astype(np.float64), K, pri, 0)

    model = bnmtf_amd.bnmf_gibbs_optimised(R, M, K, pri, seed=0, device=local_rank, verbose=False,
                                           rank=rank, world=world, comm_id=comm_id)
    np.random.seed(0)
    model.initialise("random")
    model._push()
    h = model._handle()
    L = _lib.lib()

    def sync():
        _lib.check(L.bnmtf_sync(h))
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    def run(n, perf=None):
        _lib.check(L.bnmf_gibbs_run(h, n, _lib.UPDATE_DRAW, None, None, None, _lib.ptr(perf), None))

    run(a.warmup)
    # timed region: HIP events bracket the roofline kernel only (two records per iteration on its own stream)
    model.set_profiling(True, kernel=_lib.KERNEL_GEMM_COLS)
    perf = np.zeros((a.steps, 3))
    sync()
    t0 = time.perf_counter()
    run(a.steps, perf)
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    stats = {}
    names = {0: "gemm_rows(R~.V)", 1: "gemm_cols(R~^T.U)", 2: "sweep_rows", 3: "sweep_cols"}
    ms, n = model.kernel_stats(_lib.KERNEL_GEMM_COLS)
    stats[names[1]] = {"avg_us": 1e3 * ms / max(n, 1), "launches": n}
    # the other kernels of the iteration: a short untimed run with every timer on
    model.set_profiling(True)
    run(min(a.steps, 10))
    sync()
    for kid, nm in names.items():
        if kid == _lib.KERNEL_GEMM_COLS:
            continue
        ms, n = model.kernel_stats(kid)
        stats[nm] = {"avg_us": 1e3 * ms / max(n, 1), "launches": n}
    model.set_profiling(False)

    pcie = None
    if a.with_samples and world == 1:
        n = min(a.steps, 20)
        U = np.zeros((n, I, K), dtype=np.float32); V = np.zeros((n, J, K), dtype=np.float32)
        sync(); t1 = time.perf_counter()
        _lib.check(L.bnmf_gibbs_run(h, n, _lib.UPDATE_DRAW, _lib.ptr(U), _lib.ptr(V), None, None, None))
        sync(); pcie = n / (time.perf_counter() - t1)

    if rank == 0:
        ms_step = 1e3 * dt / a.steps
        # roofline of the dense kernel of the path, "the U^T.R step" (SURVEY.md 8(d)): algorithmic 4*I*J bytes (R~ read
        # once) and 2*I*J*K flop per launch (per rank: its column shard).  Since the contraction moved to the bf16 matrix
        # cores (three-term operand splits, 6 MFMA products per fp32 product) it is a stream of R~ from HBM.
        g = stats["gemm_cols(R~^T.U)"]
        flops = 2.0 * I * (J / world) * K
        achieved = flops / (g["avg_us"] * 1e-6) / 1e12 if g["avg_us"] > 0 else 0.0
        bytes_alg = 4.0 * I * (J / world)
        # whole-iteration bound: max(t_MFMA, t_HBM) of the two contractions, SURVEY.md 8(d)
        f32_gemm = os.environ.get("BNMTF_GEMM") == "f32"
        t_mfma = (4.0 * I * J * K / (PEAK_F32_MFMA_TFLOPS * 1e12) if f32_gemm else 6 * 4.0 * I * J * K / (PEAK_BF16_MFMA_TFLOPS * 1e12)) / world
        t_hbm = (2.0 * I * J * 4.125 + 8.0 * (I + J) * K) / (PEAK_HBM_GBS * 1e9) / world
        traffic = None
        try:     # HBM bytes per launch of this kernel from the committed rocprofv3 PMC passes (profiles/traffic.json)
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            if a.workload == "bnmf_8192_k64" and world == 1:
                traffic = tj["hbm_bytes_per_launch"]
        except Exception:
            pass
        out = {
            "metric": "Gibbs iterations/sec (BNMF, I=J=8192, K=64)" if a.workload == "bnmf_8192_k64" else "Gibbs iterations/sec (%s)" % a.workload,
            "value": a.steps / dt, "unit": "Gibbs iterations/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BNMF Gibbs, synthetic R %dx%d K=%d, 10%% missing mask, priors alpha=beta=1 lambda=0.1, init random" % (I, J, K),
                       "parallelism": "rows/cols split x%d, RCCL all-gather of factor blocks" % world if world > 1 else "single GPU",
                       "samples": "device-resident"},
            "roofline": ({"bound": "mfma", "kernel": "gemm_cols: Pv = R~^T.U (f32 MFMA 32x32x2)", "achieved": achieved,
                          "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                          "algorithmic_per_launch": {"flop": flops, "bytes": bytes_alg}, "avg_launch_us": g["avg_us"]}
                         if f32_gemm else
                         {"bound": "hbm", "kernel": "gemm_cols: Pv = R~^T.U (bf16x3 MFMA 32x32x16, fp32-exact products)",
                          "achieved": bytes_alg / (g["avg_us"] * 1e-6) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                          "frac": bytes_alg / (g["avg_us"] * 1e-6) / 1e9 / PEAK_HBM_GBS, "traffic": traffic,
                          "algorithmic_per_launch": {"flop": flops, "bytes": bytes_alg}, "avg_launch_us": g["avg_us"]}),
            "iteration_bound": {"t_mfma_us": 1e6 * t_mfma, "t_hbm_us": 1e6 * t_hbm,
                                "frac_of_bound": max(t_mfma, t_hbm) / (dt / a.steps)},
            "kernels": stats,
            "mse_first_last": [float(perf[0, 0]), float(perf[-1, 0])],
            "cpu_baseline": cpu,
        }
        if pcie is not None:
            out["pcie_inclusive_iterations_per_s"] = pcie
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
