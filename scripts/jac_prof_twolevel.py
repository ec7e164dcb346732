import numpy as np
a=np.loadtxt('gpurun_out/jac_prof.txt',dtype=np.int64).reshape(512,12,12)[:,:,2:]
for w in [0,5,20,40]:
    print(w, "assemble", a[w,0,0]-a[w,0,8], [(int(a[w,r,3]-a[w,r,2]), int(a[w,r,4]-a[w,r,3]), int(a[w,r,5]-a[w,r,4])) for r in range(1,5)], "sweep total", a[w,4,5]-a[w,1,2])
