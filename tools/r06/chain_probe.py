import numpy as np, random, sys
sys.path.insert(0, '.')
from bnmtf_amd import bnmtf_vb_optimised
from bnmtf_amd.synthetic import generate_bnmtf
I=J=2048; K=L=32
R,M,_,_,_=generate_bnmtf(I,J,K,L,0.1,seed_data=1,seed_mask=2)
pri=dict(alpha=1.,beta=1.,lambdaF=0.1,lambdaS=0.1,lambdaG=0.1)
b=bnmtf_vb_optimised(R,M,K,L,pri,verbose=False)
np.random.seed(0); random.seed(0)
b.initialise("random","random")
import os
n=int(os.environ.get("NIT","1"))
b.run(n)
x = -b.muS * np.sqrt(b.tauS)
print("x quantiles", np.percentile(x, [0, 5, 25, 50, 75, 90, 95, 99, 100]).round(2))
h, e = np.histogram(x, bins=[-1e9, -6, -3, 0, 2, 4, 6, 8, 10, 15, 20, 30, 1e9])
print("x histogram", list(zip(e[:-1], h)))
