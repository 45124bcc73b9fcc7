import sys, os, numpy as np, torch
sys.path.insert(0, '.')
import rsparse_amd as rs
from rsparse_amd import synth
g = int(sys.argv[1]) if len(sys.argv) > 1 else 64
k = 256
A = synth.laplacian_3d(g); n = A[0]
F = rs.CholHandle(A, rs.schol(A, 0)); L = F.factor(); T = (n, n, L.p, L.i.astype(np.int64), L.x)
lo, up = rs.TriHandle(T, 0), rs.TriHandle(T, 1)
X = torch.rand((n, k), dtype=torch.float64, device='cuda')
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    rs.solve_pair_dev(lo, up, X.data_ptr(), k, k, st)
torch.cuda.synchronize()
print(lo.info(), up.info())
