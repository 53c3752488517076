import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from bnmtf_amd import bnmf_vb_optimised
from bnmtf_amd.synthetic import generate_bnmf
R, M, _, _ = generate_bnmf(8192, 8192, 64, 0.1, tau=1.0, seed_data=0, seed_mask=1)
pri = dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1)
b = bnmf_vb_optimised(R, M, 64, pri, verbose=False)
b.initialise("exp"); b.run(150)
e = np.array(b.all_elbo); t = np.array(b.all_elbo_terms)
print("elbo finite:", np.isfinite(e).sum(), "of", len(e), "first non-finite at", int(np.argmax(~np.isfinite(e))) if (~np.isfinite(e)).any() else None)
i = int(np.argmax(~np.isfinite(e))) if (~np.isfinite(e)).any() else len(e) - 1
print("terms at", i, t[i]); print("terms before", t[max(i - 1, 0)])
print("min tauU*muU^2-ish: muU min %.3g max %.3g tauU min %.3g max %.3g" % (b.muU.min(), b.muU.max(), b.tauU.min(), b.tauU.max()))
x = b.muU * np.sqrt(b.tauU)
print("mu*sqrt(tau) min %.4g ; units below -37: %d" % (x.min(), (x < -37).sum()))
