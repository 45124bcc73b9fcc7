import sys, time, numpy as np
sys.path.insert(0,'.')
import rsparse_amd as rs
from rsparse_amd import synth
A = synth.laplacian_2d(1024)
b = np.ones(A[0])
for _ in range(2):
    x=b.copy(); t=time.time(); rs.cholsol(A,x,0); print("one shot", time.time()-t)
