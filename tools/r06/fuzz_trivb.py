"""Random shapes, masks and ranks through bnmtf_vb's round-6 paths (the on-chip F / G sweeps with the covariance term, the blocked S
pass with its table, exp_square_diff from the sweeps' sums, the contractions beside the S pass) against the generic kernels
(BNMTF_VB_GENERIC=1, BNMTF_VB_CHAIN=steps, BNMTF_TRI_OVERLAP=0 in a child process) and the direct exp_square_diff.
    python tools/r06/fuzz_trivb.py [seconds] [seed]"""
import os, subprocess, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def case(rs):
    I = int(rs.randint(200, 2600)); J = int(rs.randint(200, 2600))
    K = int(rs.randint(1, 33)); L = int(rs.randint(1, 33))
    frac = float(rs.uniform(0.02, 0.35))
    ragged = bool(rs.rand() < 0.4)
    return dict(I=I, J=J, K=K, L=L, frac=frac, ragged=ragged, seed=int(rs.randint(1 << 30)), its=int(rs.randint(1, 4)))


def run(c):
    from bnmtf_amd import bnmtf_vb_optimised
    rs = np.random.RandomState(c["seed"])
    I, J, K, L = c["I"], c["J"], c["K"], c["L"]
    F0 = rs.exponential(1.0, (I, K)); S0 = rs.exponential(1.0, (K, L)); G0 = rs.exponential(1.0, (J, L))
    R = F0 @ S0 @ G0.T + rs.randn(I, J)
    if c["ragged"]:
        fr = rs.uniform(0.0, 2 * c["frac"], size=I)
        M = (rs.uniform(size=(I, J)) >= fr[:, None]).astype(float)
    else:
        M = (rs.uniform(size=(I, J)) >= c["frac"]).astype(float)
    M[np.arange(I), rs.randint(0, J, I)] = 1.0; M[rs.randint(0, I, J), np.arange(J)] = 1.0
    pri = dict(alpha=1., beta=1., lambdaF=0.1, lambdaS=0.1, lambdaG=0.1)
    b = bnmtf_vb_optimised(R, M, K, L, pri, verbose=False)
    np.random.seed(c["seed"] % 1000)
    b.initialise("random", "random")
    orders = np.array([np.concatenate([rs.permutation(K * L), rs.permutation(K), rs.permutation(L)]) for _ in range(c["its"])], dtype=np.int32)
    b.run(c["its"], orders=orders)
    esd_direct = b.exp_square_diff()
    out = dict(expF=b.expF, expS=b.expS, expG=b.expG, tauF=b.tauF, tauS=b.tauS, tauG=b.tauG, exptau=np.array(b.all_exp_tau), mse=np.array(b.all_performances["MSE"]),
               beta_s=np.array(b.beta_s), esd=np.array(esd_direct), path=b.describe().split("]")[-1])
    b.close()
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        c = json.loads(sys.argv[2])
        o = run(c)
        np.savez(sys.argv[3], **{k: v for k, v in o.items() if k != "path"})
        sys.exit(0)
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    t0 = time.time(); n = 0; worst = 0.0; fastn = 0
    while time.time() - t0 < budget:
        c = case(rs)
        f = run(c)
        env = dict(os.environ, BNMTF_VB_GENERIC="1", BNMTF_VB_CHAIN="steps", BNMTF_TRI_OVERLAP="0")
        tmp = "/tmp/fuzz_trivb_%d.npz" % os.getpid()
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", json.dumps(c), tmp], env=env, capture_output=True, text=True)
        assert r.returncode == 0, (c, r.stderr[-2000:])
        g = np.load(tmp)
        errs = {}
        for k in ("expF", "expS", "expG", "tauF", "tauS", "tauG", "exptau", "mse"):
            errs[k] = float(np.abs(f[k] - g[k]).max() / (np.abs(g[k]).max() + 1e-30))
        errs["beta_vs_direct"] = float(abs(f["beta_s"] - (1.0 + 0.5 * f["esd"])) / f["beta_s"])
        bad = {k: v for k, v in errs.items() if not (v < (2e-3 if k != "beta_vs_direct" else 1e-4))}
        n += 1; fastn += "pairs+cov" in f["path"]
        worst = max(worst, max(errs.values()))
        if bad:
            print("MISMATCH", c, errs, f["path"]); sys.exit(1)
    print("fuzz_trivb: %d cases (%d on the on-chip sweeps), worst relative difference %.2e" % (n, fastn, worst))
