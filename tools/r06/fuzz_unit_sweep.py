"""Random shapes, ranks and masks through the unit-per-wave sweep (csrc/kernel_sweep_unit.hip) against the pair-layout kernels
(BNMTF_UNIT=0, a child process): three mode updates (deterministic; agreement to the order of fp32 sums) and two draws (the same
Philox chain: entries equal but for decisions on a rounding boundary).   python tools/r06/fuzz_unit_sweep.py [seconds] [seed]"""
import json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run(c):
    from bnmtf_amd import bnmf_gibbs_optimised
    rs = np.random.RandomState(c["seed"])
    I, J, K = c["I"], c["J"], c["K"]
    U0 = rs.exponential(1.0, (I, K)); V0 = rs.exponential(1.0, (J, K))
    R = U0 @ V0.T + rs.randn(I, J)
    fr = rs.uniform(c["lo"], c["hi"], size=I)
    M = (rs.uniform(size=(I, J)) >= fr[:, None]).astype(float)
    M[np.arange(I), rs.randint(0, J, I)] = 1.0; M[rs.randint(0, I, J), np.arange(J)] = 1.0
    pri = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)
    Us = rs.exponential(1.0, (I, K)); Vs = rs.exponential(1.0, (J, K))
    out = {}
    for mode in ("mode", "draw"):
        b = bnmf_gibbs_optimised(R, M, K, pri, verbose=False, seed=c["seed"] % 997)
        b.set_small_path(False)
        b.U, b.V, b.tau = Us.copy(), Vs.copy(), 0.8
        b.run(3 if mode == "mode" else 2, update=mode)
        # (draws: the FIRST iteration -- one decision that falls the other way changes everything drawn after it)
        pick = -1 if mode == "mode" else 0
        out[mode + "_U"] = b.all_U[pick].copy(); out[mode + "_V"] = b.all_V[pick].copy(); out[mode + "_mse"] = np.array(b.all_performances["MSE"])
        out["desc"] = b.describe()
        b.close()
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        o = run(json.loads(sys.argv[2]))
        np.savez(sys.argv[3], **{k: v for k, v in o.items() if k != "desc"})
        sys.exit(0)
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    t0 = time.time(); n = 0; nunit = 0; worst = 0.0
    while time.time() - t0 < budget:
        lo = float(rs.uniform(0.0, 0.2))
        c = dict(I=int(rs.randint(64, 2300)), J=int(rs.randint(64, 2300)), K=int(rs.randint(1, 65)), lo=lo, hi=lo + float(rs.uniform(0.02, 0.5)), seed=int(rs.randint(1 << 30)))
        f = run(c)
        tmp = "/tmp/fuzz_unit_%d.npz" % os.getpid()
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", json.dumps(c), tmp], env=dict(os.environ, BNMTF_UNIT="0"), capture_output=True, text=True)
        assert r.returncode == 0, (c, r.stderr[-2000:])
        g = np.load(tmp)
        errs = {}
        for k in ("mode_U", "mode_V"):
            errs[k] = float(np.abs(f[k] - g[k]).max() / max(1.0, np.abs(g[k]).max()))
        errs["mode_mse"] = float(np.abs(f["mode_mse"] / g["mode_mse"] - 1).max())
        for k in ("draw_U", "draw_V"):
            d = np.abs(f[k] - g[k]) / (1e-3 + np.abs(g[k]))
            errs[k] = 1.0 - float(np.mean(d < 2e-3))
        n += 1; nunit += "unit_sweep[rows=1" in f["desc"] or "cols=1" in f["desc"]
        worst = max(worst, errs["mode_U"], errs["mode_V"])
        bad = {k: v for k, v in errs.items() if not (v < (0.03 if k.startswith("draw") else 1e-3))}
        if bad:
            print("MISMATCH", c, errs, f["desc"][-80:]); sys.exit(1)
    print("fuzz_unit_sweep: %d cases (%d with a unit-per-wave direction), worst relative difference of the mode updates %.2e" % (n, nunit, worst))
