"""The masked contraction (K1 / K2, kernel_gemm.hip) in its two arithmetic forms: fp32-exact products on the bf16 matrix
cores (three-term operand splits, the default) and the plain f32 MFMA kernel (BNMTF_GEMM=f32).  Both feed the same
conditional parameters (bnmf_gibbs_optimised.py:167-177): each is compared with the oracle, and with the other."""
import numpy as np
import pytest

from bnmtf_amd import bnmf_gibbs_optimised
from bnmtf_amd.synthetic import generate_bnmf
from oracle import bnmtf_oracle as O

pytestmark = pytest.mark.gpu

PRI = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)


@pytest.mark.parametrize("I,J,K", [(700, 900, 24), (1300, 640, 64)])
def test_f32_mfma_contraction_equals_bf16x3_contraction_and_oracle(monkeypatch, I, J, K):
    """(the f32-MFMA kernel is an experiment: `make EXPERIMENTS=1`; the shipped build checks the bf16x3 kernel against the oracle)"""
    from bnmtf_amd import _lib
    modes = ("bf16x3", "f32") if _lib.lib().bnmtf_has_experiments() else ("bf16x3",)
    R, M, _, _ = generate_bnmf(I, J, K, 0.15, seed_data=11, seed_mask=12)
    rs = np.random.RandomState(1)
    U0 = rs.exponential(1.0, (I, K)); V0 = rs.exponential(1.0, (J, K))
    o = O.BNMFGibbsOracle(R.astype(np.float64), M, K, PRI)
    o.U, o.V, o.tau = U0.copy(), V0.copy(), 0.8
    got = {}
    for mode in modes:
        if mode == "f32":
            monkeypatch.setenv("BNMTF_GEMM", "f32")
        b = bnmf_gibbs_optimised(R, M, K, PRI, verbose=False, seed=2)
        b.U, b.V, b.tau = U0.copy(), V0.copy(), 0.8
        out = []
        for k in (0, K // 2, K - 1):
            tU = o.tauU(k); tV = o.tauV(k)
            mU, mV = b.muU(tU, k), b.muV(tV, k)
            su = np.abs(o.muU(tU, k)).max() + 1.0; sv = np.abs(o.muV(tV, k)).max() + 1.0
            assert np.abs(mU - o.muU(tU, k)).max() < 5e-5 * su, (mode, k)
            assert np.abs(mV - o.muV(tV, k)).max() < 5e-5 * sv, (mode, k)
            out += [mU, mV]
        b.run(3, update="mode")
        out += [b.U.copy(), b.V.copy(), np.array(b.all_tau)]
        got[mode] = out
        b.close()
    if "f32" in got:                                # (the shipped build has one form: nothing to compare it with but the oracle, above)
        for x, y in zip(got["bf16x3"], got["f32"]):
            assert np.abs(x - y).max() <= 2e-5 * (np.abs(y).max() + 1.0)
