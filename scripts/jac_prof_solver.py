import numpy as np
a=np.loadtxt('gpurun_out/jac_prof.txt',dtype=np.int64).reshape(512,12,12)[:,:,2:]
for w in [0,5,20,40]:
    print(w, "assemble", a[w,0,0]-a[w,0,8], "pre+step0", a[w,1,2]-a[w,0,0], "steps1-11", a[w,11,5]-a[w,1,2], "steps 12-31", a[w,0,6]-a[w,11,5], "writes", a[w,0,1]-a[w,0,6], "total", a[w,0,1]-a[w,0,0])
    print(w, [(int(a[w,s,3]-a[w,s,2]), int(a[w,s,4]-a[w,s,3]), int(a[w,s,5]-a[w,s,4])) for s in range(1,8)])
