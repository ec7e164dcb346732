import sys, numpy as np
sys.path.insert(0,'.')
from xmca_amd import _hip
n=int(sys.argv[1]); cplx = len(sys.argv)>2
rng=np.random.default_rng(0); X=rng.standard_normal((n,2*n)); 
if cplx: X = X + 1j*rng.standard_normal((n,2*n))
G=X@X.conj().T
h=_hip.Handle(0); h.eigh(G, vectors=False)
