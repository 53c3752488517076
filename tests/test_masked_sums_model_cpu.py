"""CPU model of the arithmetic in bnmtf_amd/csrc/kernel_maskgemm.hip (the variational sweep's chain-independent masked sums,
bnmf_vb_optimised.py:189-199): per-column fixed-point grid, balanced base-256 digits, integer accumulation, one rounding at the
end.  Checks the statements the kernel's header makes -- digit ranges (int8 operands), exact reconstruction, the accumulators'
range (int32), the error bound against the fp64 sum -- without a GPU; tests/test_bnmf_vb_gpu.py holds the device against the
same bound through bnmf_vb_masked_sums."""
import numpy as np


def planes(x):
    """vb_colmax_kernel + vb_planes_kernel: x [rows][cols] non-negative fp32 -> (e[cols], d0, d1, d2 int8-range arrays)."""
    x = np.asarray(x, dtype=np.float32)
    mx = x.max(0)
    bits = mx.view(np.uint32)
    e = np.where(bits != 0, ((bits >> 23) & 255).astype(np.int64) - 126, 0)           # 2^e > max (exponent field + 1)
    n = np.rint(np.ldexp(x.astype(np.float64), (22 - e)[None, :])).astype(np.int64)  # 0 .. 2^22 (ldexpf + v_cvt_i32_f32, RNE)
    d2 = ((n + 128) & 255) - 128
    n1 = (n - d2) >> 8
    d1 = ((n1 + 128) & 255) - 128
    d0 = (n1 - d1) >> 8
    return e, d0, d1, d2, n


def masked_sums(miss, x):
    """maskgemm_kernel: integer sums per digit plane, combined as fmaf(d0, 65536, fmaf(d1, 256, d2)) in fp32, scaled by 2^(e - 22)."""
    e, d0, d1, d2, _ = planes(x)
    mi = miss.astype(np.int64)
    D0, D1, D2 = mi @ d0, mi @ d1, mi @ d2
    inner = (np.float32(256.0) * D1.astype(np.float32) + D2.astype(np.float32)).astype(np.float32)        # exact products, one rounding
    tot = (D0.astype(np.float64) * 65536.0 + inner.astype(np.float64)).astype(np.float32)                  # the outer FMA: one rounding
    return np.ldexp(tot.astype(np.float64), (e - 22)[None, :]).astype(np.float32), (D0, D1, D2)


def test_digit_planes_are_int8_and_exact():
    rs = np.random.RandomState(0)
    x = (10.0 ** rs.uniform(-8, 3, (4096, 24))).astype(np.float32)
    x[::7, 3] = 0.0; x[:, 5] = 0.0                                                    # zeros and an all-zero column
    x[11, 7] = np.float32(2.0) ** 100; x[12, 8] = np.float32(2.0) ** -120             # a huge and a tiny column maximum
    e, d0, d1, d2, n = planes(x)
    assert d0.min() >= 0 and d0.max() <= 64 and d1.min() >= -128 and d1.max() <= 127 and d2.min() >= -128 and d2.max() <= 127
    assert np.array_equal(d0 * 65536 + d1 * 256 + d2, n) and n.min() >= 0 and n.max() <= 2 ** 22
    # the grid: 2^(e-1) <= max < 2^e for every non-zero column; an element is off by at most half a grid step
    mx = x.max(0).astype(np.float64)
    nz = mx > 0
    assert np.all(mx[nz] < np.ldexp(1.0, e[nz])) and np.all(mx[nz] >= np.ldexp(1.0, e[nz] - 1))
    err = np.abs(np.ldexp(n.astype(np.float64), (e - 22)[None, :]) - x.astype(np.float64))
    assert np.all(err <= np.ldexp(1.0, e - 23)[None, :] * (1 + 1e-12))


def test_masked_sums_model_against_fp64_and_accumulator_range():
    rs = np.random.RandomState(1)
    rows, units, cols = 8192, 96, 16
    x = (10.0 ** rs.uniform(-5, 1, (rows, cols))).astype(np.float32)
    miss = rs.rand(units, rows) < 0.1
    miss[0] = True                                                                     # a unit with every entry missing: the accumulators' worst case
    got, (D0, D1, D2) = masked_sums(miss, x)
    for D in (D0, D1, D2):
        assert np.abs(D).max() < 2 ** 31                                               # fits the int32 accumulator tiles (8192 x 128 at most)
        assert np.abs(D).max() < 2 ** 24                                               # ... and converts to fp32 exactly
    ref = miss.astype(np.float64) @ x.astype(np.float64)
    n_miss = miss.sum(1)[:, None]
    bound = n_miss * x.max(0).astype(np.float64)[None, :] * 2.0 ** -22 + 1.2e-7 * ref
    assert np.all(np.abs(got.astype(np.float64) - ref) <= bound)
    # the order of the additions does not matter (integers): any split of the inner range gives the same digits sums
    half = rows // 2
    e, d0, d1, d2, _ = planes(x)
    mi = miss.astype(np.int64)
    assert np.array_equal(mi[:, :half] @ d0[:half] + mi[:, half:] @ d0[half:], D0)
    # typical accuracy: ~sqrt(n) half-steps -- an order below the fp32 rounding of the sums the sweep forms beside it
    big = ref > 1e-2 * ref.max()
    assert (np.abs(got.astype(np.float64) - ref)[big] / ref[big]).max() < 3e-7
