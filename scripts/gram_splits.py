import sys; sys.path.insert(0,'.')
import numpy as np
from xmca_amd import _hip
h=_hip.Handle(0)
T,N=2920,10000
fl=T*(T+1)*N
for s in [1,2,3,4,5,6,7,8,9,10,11,12,14]:
    ms=h.bench_gemm(T,T,N,np.float64,a_kfast=True,b_nfast=False,upper_only=True,splits=s,reps=5)
    print("splits",s,"ms %.3f"%ms,"TF %.1f"%(fl/ms/1e9), "frac %.3f"%(fl/ms/1e9/78.6))
ms=h.bench_gemm(T,T,N,np.float64,a_kfast=True,b_nfast=False,upper_only=True,splits=0,reps=5)
print("auto", ms, fl/ms/1e9)
