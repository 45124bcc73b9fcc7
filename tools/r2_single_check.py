"""Residuals of narrow blocks (1, 7, 8, 16 columns) through the resident Cholesky solve: usage r2_single_check.py <grid>"""
import sys, numpy as np, torch, scipy.sparse as sp
sys.path.insert(0, '.')
import rsparse_amd as rs
from rsparse_amd import synth
g = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
A = synth.laplacian_2d(g); n = A[0]
Asp = sp.csc_matrix((A[4], A[3], A[2]), shape=(n, n))
F = rs.CholHandle(A, rs.schol(A, 0))
st = torch.cuda.current_stream().cuda_stream
for k in (1, 7, 8, 16, 64):
    xs = 2.0 * synth.splitmix64(3, n * k).reshape(n, k) - 1.0
    B = Asp @ xs
    xb = torch.tensor(B, device='cuda').contiguous()
    F.solve_dev(xb.data_ptr(), rs.api.ROW_MAJOR, k, k, True, st); torch.cuda.synchronize()
    X = xb.cpu().numpy()
    print(g, k, "rows: residual", np.abs(Asp @ X - B).max() / np.abs(B).max(), "x err", np.abs(X - xs).max(), flush=True)
b = B[:, 0].copy()
F.solve(b)
print("host 1 rhs: x err", np.abs(b - xs[:, 0]).max())
