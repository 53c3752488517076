"""The on-chip half sweep with one unit per 64-lane wave (csrc/kernel_sweep_unit.hip, round 6): the shape of few units per CU --
the shards of a multi-GPU run (8192 x 8192 over 8 ranks: 1 024 units per rank and direction) and small single-GPU problems.
The reference's column loops (bnmf_gibbs_optimised.py:134-142; nmf_icm.py:124-134 in the mode update) are what it computes;
checked here against the fp64 oracle and against the pair-layout shapes (sweep_chip.inc) it replaces at these sizes.  The small
golden cases of tests/test_bnmf_gibbs_gpu.py run it as well ("chip")."""
import numpy as np
import pytest

from bnmtf_amd import bnmf_gibbs_optimised, bnmtf_gibbs_optimised
from bnmtf_amd.synthetic import generate_bnmf, generate_bnmtf
from oracle import bnmtf_oracle as O

pytestmark = pytest.mark.gpu

PRI = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)
PRI3 = dict(alpha=1., beta=1., lambdaF=0.1, lambdaS=0.1, lambdaG=0.1)


def _ragged_mask(rs, I, J, lo, hi):
    frac = rs.uniform(lo, hi, size=I)
    M = (rs.uniform(size=(I, J)) >= frac[:, None]).astype(np.float64)
    M[np.arange(I), rs.randint(0, J, I)] = 1.0
    M[rs.randint(0, I, J), np.arange(J)] = 1.0
    return M


@pytest.mark.parametrize("I,J,K,lo,hi", [(300, 200, 8, 0.05, 0.15), (515, 389, 40, 0.0, 0.6), (1100, 700, 64, 0.05, 0.3), (130, 2500, 33, 0.02, 0.2)])
def test_mode_updates_follow_the_oracle_and_the_pair_layout_shapes(monkeypatch, I, J, K, lo, hi):
    """Six mode updates (deterministic): the unit-per-wave sweep against the fp64 oracle, and against the pair-layout kernels
    (BNMTF_UNIT=0) it stands in for -- ragged masks, so several slot classes per launch; 1 100 rows: eight unit waves per block."""
    rs = np.random.RandomState(I + 7 * J)
    U0 = rs.exponential(1.0, (I, K)); V0 = rs.exponential(1.0, (J, K))
    R = U0 @ V0.T + rs.randn(I, J)
    M = _ragged_mask(rs, I, J, lo, hi)
    Us = rs.exponential(1.0, (I, K)); Vs = rs.exponential(1.0, (J, K))
    out = {}
    for unit in ("1", "0"):
        monkeypatch.setenv("BNMTF_UNIT", unit)
        b = bnmf_gibbs_optimised(R, M, K, PRI, verbose=False, seed=5)
        b.set_small_path(False)
        b.U, b.V, b.tau = Us.copy(), Vs.copy(), 0.8
        b.run(6, update="mode")
        d = b.describe()
        assert ("unit_sweep[rows=1" in d) == (unit == "1") and ("cols=1" in d.split("unit_sweep[")[1]) == (unit == "1" and J <= 2048), d
        if unit == "1" and I > 1024:
            assert "unit_sweep[rows=1/8" in d, d
        out[unit] = (b.all_U.copy(), b.all_V.copy(), b.all_tau.copy(), np.array(b.all_performances["MSE"]))
        b.close()
    o = O.BNMFGibbsOracle(R, M, K, PRI)
    o.U, o.V, o.tau = Us.copy(), Vs.copy(), 0.8
    o.run(6, draw=False)
    u = out["1"]
    sU = max(1.0, np.abs(o.all_U[0]).max()); sV = max(1.0, np.abs(o.all_V[0]).max())
    assert np.abs(u[0][0] - o.all_U[0]).max() < 3e-4 * sU and np.abs(u[1][0] - o.all_V[0]).max() < 3e-4 * sV
    assert np.abs(u[0][-1] - o.all_U[-1]).max() < 5e-3 * max(1.0, np.abs(o.all_U[-1]).max())
    np.testing.assert_allclose(u[2], o.all_tau, rtol=3e-4)
    np.testing.assert_allclose(u[3], o.all_performances["MSE"], rtol=3e-4)
    p = out["0"]                       # the pair-layout shapes: same numbers up to the order of the fp32 sums
    assert np.abs(u[0][0] - p[0][0]).max() < 2e-4 * sU and np.abs(u[1][0] - p[1][0]).max() < 2e-4 * sV
    np.testing.assert_allclose(u[3], p[3], rtol=2e-4)


def test_draws_follow_the_oracles_philox_chain():
    """Draw mode: the candidate sequence, keyed by (seed; row, column, iteration, stream | candidate), is the oracle's; the first
    sweep agrees element-wise but for decisions that sit on a rounding boundary, the statistics-based MSE with the direct one."""
    I, J, K = 640, 512, 24
    R, M, _, _ = generate_bnmf(I, J, K, 0.12, seed_data=5, seed_mask=6)
    rs = np.random.RandomState(3)
    U0 = rs.exponential(1.0, (I, K)); V0 = rs.exponential(1.0, (J, K))
    b = bnmf_gibbs_optimised(R, M, K, PRI, verbose=False, seed=77)
    b.set_small_path(False)
    b.U, b.V, b.tau = U0.copy(), V0.copy(), 0.7
    b.run(12)
    assert "unit_sweep[rows=1/4" in b.describe()
    o = O.BNMFGibbsOracle(R.astype(np.float64), M, K, PRI, seed=77)
    o.U, o.V, o.tau = U0.copy(), V0.copy(), 0.7
    o.run(3)
    for dev, ora in ((b.all_U[0], o.all_U[0]), (b.all_V[0], o.all_V[0])):
        d0 = np.abs(dev - ora) / (1e-3 + np.abs(ora))
        assert np.mean(d0 < 1e-3) > 0.99
    np.testing.assert_allclose(b.all_performances["MSE"][:2], o.all_performances["MSE"][:2], rtol=1e-3)
    assert abs(b.all_tau[0] - o.all_tau[0]) < 1e-3 * o.all_tau[0]
    # the iteration's metrics come from the sweep's own sums (Gram identities): they are the metrics of the state it left
    direct = b.predict_while_running()
    assert abs(direct["MSE"] / b.all_performances["MSE"][-1] - 1) < 1e-4
    assert (b.U >= 0).all() and np.isfinite(b.U).all() and b.all_performances["MSE"][-1] < b.all_performances["MSE"][0]


def test_a_run_in_two_calls_and_a_second_model_draw_the_same_chain():
    """The sweep depends on a unit's own missing list, the seed and the iteration counter only: run(3); run(4) is run(7), and a
    second handle draws the same bits."""
    I, J, K = 400, 330, 16
    R, M, _, _ = generate_bnmf(I, J, K, 0.1, seed_data=1, seed_mask=2)
    rs = np.random.RandomState(0)
    U0 = rs.exponential(1.0, (I, K)); V0 = rs.exponential(1.0, (J, K))
    runs = []
    for split in ((7,), (3, 4), (7,)):
        b = bnmf_gibbs_optimised(R, M, K, PRI, verbose=False, seed=9)
        b.set_small_path(False)
        b.U, b.V, b.tau = U0.copy(), V0.copy(), 1.0
        for n in split:
            b.run(n)
        runs.append((b.U.copy(), b.V.copy(), b.tau))
        b.close()
    for r in runs[1:]:
        assert np.array_equal(r[0], runs[0][0]) and np.array_equal(r[1], runs[0][1]) and r[2] == runs[0][2]


def test_tri_factorisation_sweeps_run_it_too():
    """bnmtf_gibbs: the F and G sweeps are this kernel against the effective factors G S^T and F S (bnmtf_gibbs_optimised.py:
    195-199, 207-211); mode updates against the oracle."""
    I, J, K, L = 260, 300, 6, 9
    R, M, _, _, _ = generate_bnmtf(I, J, K, L, 0.1, seed_data=3, seed_mask=4)
    rs = np.random.RandomState(1)
    F0 = rs.exponential(1.0, (I, K)); S0 = rs.exponential(1.0, (K, L)); G0 = rs.exponential(1.0, (J, L))
    b = bnmtf_gibbs_optimised(R, M, K, L, PRI3, verbose=False, seed=3)
    b.set_small_path(False)
    b.F, b.S, b.G, b.tau = F0.copy(), S0.copy(), G0.copy(), 0.9
    b.run(4, update="mode")
    assert "unit_sweep[rows=1" in b.describe()
    o = O.BNMTFGibbsOracle(R.astype(np.float64), M, K, L, PRI3)
    o.F, o.S, o.G, o.tau = F0.copy(), S0.copy(), G0.copy(), 0.9
    o.run(4, draw=False)
    assert np.abs(b.all_F[0] - o.all_F[0]).max() < 5e-4 * max(1.0, np.abs(o.all_F[0]).max())
    assert np.abs(b.all_G[0] - o.all_G[0]).max() < 5e-4 * max(1.0, np.abs(o.all_G[0]).max())
    np.testing.assert_allclose(b.all_performances["MSE"], o.all_performances["MSE"], rtol=1e-3)
    np.testing.assert_allclose(b.all_tau, o.all_tau, rtol=1e-3)
