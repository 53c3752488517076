"""Synthetic problems for tests and bench.py, following the reference's toy generator
(data_toy/bnmf/generate_bnmf.py:27-83, code/cross_validation/mask.py:9-14):
U, V ~ Exp(1), R = U.V^T + N(0, 1/tau), mask with exactly floor(fraction*I*J) zeros
placed uniformly without replacement and redrawn until no row/column is empty."""
import numpy as np


def generate_mask(I, J, fraction_unknown, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    n_zero = int(fraction_unknown * I * J)
    for _ in range(1000):
        M = np.ones(I * J, dtype=np.uint8)
        M[rng.choice(I * J, size=n_zero, replace=False)] = 0
        M = M.reshape(I, J)
        if M.sum(axis=0).min() > 0 and M.sum(axis=1).min() > 0:
            return M
    raise RuntimeError("could not draw a mask without empty rows/columns")


def generate_bnmf(I, J, K, fraction_unknown=0.1, tau=1.0, seed_data=0, seed_mask=1, dtype=np.float32):
    rng = np.random.Generator(np.random.PCG64(seed_data))
    U = rng.exponential(1.0, (I, K)).astype(dtype)
    V = rng.exponential(1.0, (J, K)).astype(dtype)
    R = U @ V.T
    R += rng.normal(0.0, 1.0 / np.sqrt(tau), (I, J)).astype(dtype)
    return R, generate_mask(I, J, fraction_unknown, seed_mask), U, V


def generate_bnmtf(I, J, K, L, fraction_unknown=0.1, tau=1.0, seed_data=0, seed_mask=1, dtype=np.float32):
    rng = np.random.Generator(np.random.PCG64(seed_data))
    F = rng.exponential(1.0, (I, K)).astype(dtype)
    S = rng.exponential(1.0, (K, L)).astype(dtype)
    G = rng.exponential(1.0, (J, L)).astype(dtype)
    R = (F @ S) @ G.T
    R += rng.normal(0.0, 1.0 / np.sqrt(tau), (I, J)).astype(dtype)
    return R, generate_mask(I, J, fraction_unknown, seed_mask), F, S, G
