"""Counter-based RNG + samplers: the CPU statement of what the HIP kernels draw.

TEST INFRASTRUCTURE (see oracle/__init__.py).  fp64 NumPy, vectorised.

The reference draws from NumPy's process-global MT19937 stream
(`code/models/distributions/rtnorm.py:13`, `gamma.py:7`, `exponential.py:4`),
one scalar at a time, through the vendored Chopin/Mazet table sampler
(`rtnorm.py:90-218`).  That stream cannot be reproduced on a GPU and the
reference's own tests only assert `draw >= 0`
(`tests/code/distributions/test_truncated_normal_vector.py:35-41`), so the
new sampler is defined here, once, and both the oracle and the HIP kernels
implement exactly this definition:

* Philox-4x32-10 (Salmon et al., SC'11), key = 64-bit seed,
  counter = (element index, column index, iteration, stream | candidate<<4).
  Every draw is therefore a pure function of (seed, iteration, element,
  column): results do not depend on launch geometry or on how many GPUs the
  rows are split over.
* TN(mu, tau) on [0, inf): standardised lower bound a = -mu*sqrt(tau).
  a <  A0: plain normal rejection (Box-Muller proposal, accept z >= a);
  a >= A0: Robert (1995) translated-exponential rejection with the optimal
           rate lam = (a + sqrt(a^2+4))/2.  The accepted value is returned
           as e/sqrt(tau) (e = the exponential excess) because mu + sigma*a
           is identically 0 (the reference computes `r*sigma+mu`,
           `rtnorm.py:78`, which cancels catastrophically in the tail).
  Candidates c = 0,1,2,... are a fixed sequence; the draw is the FIRST
  accepted candidate, so evaluating 16 or 64 candidates in parallel (one per
  lane) gives the same value as this serial statement.
  Guards follow `truncated_normal_vector.py:41-45`: tau == 0 -> 0;
  negative / non-finite -> 0.
* Gamma(shape, rate): Marsaglia-Tsang (2000) squeeze-free form in fp64.
"""
import numpy as np

PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = 0x9E3779B9
PHILOX_W1 = 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)

# stream ids (low 4 bits of counter word 3)
STREAM_ROWS = 0      # U (BNMF) / F (BNMTF) sweep
STREAM_COLS = 1      # V / G sweep
STREAM_S = 2         # S sweep (BNMTF)
STREAM_TAU = 3       # noise precision
STREAM_HOOK = 8      # stand-alone tn_sample / gamma_sample test hooks

TN_A0 = 0.25         # switch between normal and exponential proposals
TWO_PI = 6.283185307179586


def philox4x32_10(c0, c1, c2, c3, seed):
    """Vectorised Philox-4x32-10.  Inputs broadcast; returns 4 uint32 arrays."""
    c0, c1, c2, c3 = np.broadcast_arrays(
        np.asarray(c0, dtype=np.uint64), np.asarray(c1, dtype=np.uint64),
        np.asarray(c2, dtype=np.uint64), np.asarray(c3, dtype=np.uint64))
    c0 = c0 & MASK32; c1 = c1 & MASK32; c2 = c2 & MASK32; c3 = c3 & MASK32
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    k0 = seed & 0xFFFFFFFF
    k1 = (seed >> 32) & 0xFFFFFFFF
    for _ in range(10):
        p0 = PHILOX_M0 * c0          # 64-bit products (operands < 2^32)
        p1 = PHILOX_M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0), lo1,
                          hi0 ^ c3 ^ np.uint64(k1), lo0)
        k0 = (k0 + PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + PHILOX_W1) & 0xFFFFFFFF
    return (c0.astype(np.uint32), c1.astype(np.uint32),
            c2.astype(np.uint32), c3.astype(np.uint32))


def u23(r):
    """uint32 -> uniform in (0,1) from the top 23 bits: (n + 0.5) * 2^-23, n < 2^23, is exact in
    fp32 (24 significant bits), strictly inside (0,1), and the same number on the device."""
    return ((r >> np.uint32(9)).astype(np.float64) + 0.5) * (1.0 / 8388608.0)


def u32(r):
    """uint32 -> uniform in (0,1) with 32 bits (fp64 only: tau draw)."""
    return (r.astype(np.float64) + 0.5) * (1.0 / 4294967296.0)


def tn_draw(mu, tau, elem, col, it, stream, seed, max_cand=4096):
    """First-accepted-candidate TN(mu,tau) draw on [0,inf) for each element.

    mu, tau, elem, col broadcast together; `it`, `stream`, `seed` scalars.
    Replaces `TN_vector_draw` (`truncated_normal_vector.py:37-50`) /
    `TN_draw` (`truncated_normal.py:37-44`).
    """
    mu, tau, elem, col = np.broadcast_arrays(
        np.asarray(mu, dtype=np.float64), np.asarray(tau, dtype=np.float64),
        np.asarray(elem), np.asarray(col))
    shape = mu.shape
    mu = mu.ravel(); tau = tau.ravel()
    elem = elem.ravel().astype(np.uint64); col = col.ravel().astype(np.uint64)
    out = np.zeros(mu.shape, dtype=np.float64)
    live = tau > 0.0                      # tau == 0 (or NaN / negative) -> 0
    with np.errstate(invalid="ignore", divide="ignore"):
        rt = np.sqrt(np.where(live, tau, 1.0))
        a = -mu * rt
        live &= np.isfinite(a)
        d = 2.0 / (np.sqrt(a * a + 4.0) + a)  # = lam - a, no cancellation
        lam = a + d
        tail = a >= TN_A0
        todo = live.copy()
        cand = 0
        while todo.any() and cand < max_cand:
            idx = np.nonzero(todo)[0]
            r0, r1, _, _ = philox4x32_10(elem[idx], col[idx], it,
                                         int(stream) + 16 * cand, seed)
            u1 = u23(r0); u2 = u23(r1)
            nl = -np.log(u1)
            t = tail[idx]
            # normal proposal
            z = np.sqrt(2.0 * nl) * np.cos(TWO_PI * u2)
            acc_n = z >= a[idx]
            x_n = mu[idx] + z / rt[idx]
            # exponential proposal
            e = nl / lam[idx]
            acc_e = u2 <= np.exp(-0.5 * (e - d[idx]) ** 2)
            x_e = e / rt[idx]
            acc = np.where(t, acc_e, acc_n)
            x = np.where(t, x_e, x_n)
            got = idx[acc]
            out[got] = x[acc]
            todo[got] = False
            cand += 1
    bad = ~np.isfinite(out) | (out < 0.0)
    out[bad] = 0.0
    return out.reshape(shape)


def gamma_draw(shape, rate, it, seed, stream=STREAM_TAU, max_cand=4096):
    """One Gamma(shape, scale=1/rate) draw (replaces `gamma.py:11-14`)."""
    shape = float(shape); rate = float(rate)
    boost = shape < 1.0
    a = shape + 1.0 if boost else shape
    d = a - 1.0 / 3.0
    c = 1.0 / np.sqrt(9.0 * d)
    for cand in range(max_cand):
        r0, r1, r2, r3 = philox4x32_10(0, 0, it, int(stream) + 16 * cand, seed)
        u1 = float(u32(r0)); u2 = float(u32(r1)); u3 = float(u32(r2)); u4 = float(u32(r3))
        x = np.sqrt(-2.0 * np.log(u1)) * np.cos(TWO_PI * u2)
        v = 1.0 + c * x
        if v <= 0.0:
            continue
        v = v * v * v
        if np.log(u3) < 0.5 * x * x + d - d * v + d * np.log(v):
            g = d * v
            if boost:
                g *= u4 ** (1.0 / shape)
            return g / rate
    raise RuntimeError("gamma_draw: no candidate accepted")
