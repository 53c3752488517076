"""The device's fp64 TN-moments routine (bnmtf_amd/csrc/device_rng.h: tn_moments) restated in NumPy with the committed
coefficients (bnmtf_amd/csrc/tn_moments_coeffs.h), against the oracle (= the reference's formula,
truncated_normal_vector.py:53-73) and against 50-digit values.  The GPU side of the same check: tests/test_distributions_gpu.py."""
import importlib.util
import os
import re

import numpy as np
import pytest

from oracle import bnmtf_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gen():
    spec = importlib.util.spec_from_file_location("gen_tn", os.path.join(ROOT, "tools", "gen_tn_moments_coeffs.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _header_coeffs():
    txt = open(os.path.join(ROOT, "bnmtf_amd", "csrc", "tn_moments_coeffs.h")).read()
    body = re.search(r"kErfcxP\[\d+\] = \{(.*?)\};", txt, re.S).group(1)
    P = [float(x) for x in body.replace("\n", " ").split(",") if x.strip()]
    deg = int(re.search(r"kErfcxDeg = (\d+)", txt).group(1))
    assert len(P) == deg + 1
    return P


def test_routine_matches_the_oracle_formula():
    g = _gen()
    P = _header_coeffs()
    rs = np.random.RandomState(1)
    mu = np.concatenate([rs.randn(4000) * 4, -np.abs(rs.randn(1000)) * 25, np.linspace(-35, 35, 141)])
    tau = np.exp(rs.randn(mu.size))
    e, v = g.moments_fp64(mu, tau, P)
    eo, vo = O.tn_expectation(mu, tau), O.tn_variance(mu, tau)
    np.testing.assert_allclose(e, eo, rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(v, vo, rtol=2e-6, atol=1e-300)      # 1 - lam (lam - x) cancels like x^4 near the -30 sigma switch


def test_erfcx_polynomial_against_mpmath():
    mp = pytest.importorskip("mpmath")
    g = _gen()
    P = _header_coeffs()
    mp.mp.dps = 40
    worst = 0.0
    for t in np.concatenate([np.linspace(0, 10, 201), [15.0, 30.0, 100.0, 1e4]]):
        s = (t - g.A) / (t + g.A)
        f = P[-1]
        for c in P[-2::-1]:
            f = f * s + c
        ref = mp.exp(mp.mpf(float(t)) ** 2) * mp.erfc(mp.mpf(float(t)))
        worst = max(worst, abs((mp.mpf(f / (1 + 2 * t)) - ref) / ref))
    assert worst < 6e-16


def _gen32():
    spec = importlib.util.spec_from_file_location("gen_tn32", os.path.join(ROOT, "tools", "gen_tn_moments_f32_coeffs.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _header_f32():
    txt = open(os.path.join(ROOT, "bnmtf_amd", "csrc", "tn_moments_f32_coeffs.h")).read()
    out = {}
    for name in ("kTnF32R", "kTnF32H", "kTnF32E"):
        body = re.search(name + r"\[\d+\] = \{(.*?)\};", txt, re.S).group(1)
        out[name] = [float(x.strip().rstrip("f")) for x in body.replace("\n", " ").split(",") if x.strip()]
        assert len(out[name]) == int(re.search(name + r"Deg = (\d+)", txt).group(1)) + 1
    return out["kTnF32R"], out["kTnF32H"], out["kTnF32E"]


def test_fp32_sweep_routine_matches_the_oracle_formula():
    """tn_moments_f32 (the routine inside the VB sweeps) in NumPy fp32 with the committed coefficients: fp32-level agreement
    with the reference's formula over both signs of mu, including mu << 0 where that formula itself needs fp64."""
    g = _gen32()
    PR, PH, PE = _header_f32()
    rs = np.random.RandomState(2)
    x = np.concatenate([rs.uniform(-12, 29.5, 6000), np.linspace(-40, 29.9, 700)])
    tau = np.exp(rs.randn(x.size)).astype(np.float32)
    mu = (-x / np.sqrt(tau.astype(np.float64))).astype(np.float32)
    e, v = g.moments_f32(mu, tau, PR, PH, PE)
    eo = O.tn_expectation(mu.astype(np.float64), tau.astype(np.float64))
    vo = O.tn_variance(mu.astype(np.float64), tau.astype(np.float64))
    xr = -mu.astype(np.float64) * np.sqrt(tau.astype(np.float64))
    ok = xr < 29.99                     # away from the reference's -30 sigma switch (the two sides differ by 1e-3 there)
    np.testing.assert_allclose(e[ok], eo[ok], rtol=2e-6, atol=1e-30)
    # the oracle's own fp64 variance loses ~x^4 eps for mu << 0; compare where that is below the fp32 level
    okv = ok & (xr < 12)
    np.testing.assert_allclose(v[okv], vo[okv], rtol=5e-6, atol=1e-30)


def test_segment_table_of_the_mean_matches_the_oracle_formula():
    """tn_mean_table.h (the S chain of the variational tri-factorisation reads a step's mean out of it, kernel_trivb.hip): the
    committed table evaluated as the device evaluates it -- fp32 segment coordinate, floor / fract, fp32 Horner -- against the
    reference's formula (truncated_normal.py:42-52 through the oracle) on x in [-6, 26), and the file is what the generator writes."""
    spec = importlib.util.spec_from_file_location("gen_tab", os.path.join(ROOT, "tools", "gen_tn_mean_table.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    txt = open(os.path.join(ROOT, "bnmtf_amd", "csrc", "tn_mean_table.h")).read()
    rows = re.findall(r"^\s*\{(.*?)\},\s*$", txt, re.M)
    tab = np.array([[float(v.strip().rstrip("f")) for v in r.split(",")] for r in rows], dtype=np.float32)
    assert tab.shape == (g.NSEG, g.DEG + 1) == (64, 6)
    assert "kTnMeanX0 = %.1ff, kTnMeanInvW = %.1ff" % (g.X0, 1.0 / g.W) in txt
    rs = np.random.RandomState(4)
    x = np.concatenate([rs.uniform(-6, 25.99, 20000), -6 + 0.5 * np.arange(64) + 1e-6]).astype(np.float32)
    got = g.eval_f32(tab, x).astype(np.float64)
    # E[TN(mu, 1)] with mu = -x: sigma = 1, so the mean IS r(x); the oracle's formula loses digits like x^2 eps for x >> 0
    ref = O.tn_expectation(-x.astype(np.float64), np.ones(x.size))
    lo = x < 12
    np.testing.assert_allclose(got[lo], ref[lo], rtol=5e-7)
    np.testing.assert_allclose(got[~lo], ref[~lo], rtol=2e-5)
    # ... and against 40-digit values over the whole range, a coarser sample (the generator's own check)
    assert g.max_rel_error(tab, n=1500, seed=1) < 3e-7
    np.testing.assert_array_equal(tab, g.table().astype(np.float32))
