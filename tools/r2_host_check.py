"""Host-buffer solves (rsp_chol_solve_f64) of 1 and 3 columns, contiguous and strided: usage r2_host_check.py <grid>"""
import sys, numpy as np, scipy.sparse as sp
sys.path.insert(0, '.')
import rsparse_amd as rs
from rsparse_amd import synth
g = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
A = synth.laplacian_2d(g); n = A[0]
Asp = sp.csc_matrix((A[4], A[3], A[2]), shape=(n, n))
F = rs.CholHandle(A, rs.schol(A, 0))
xs = 2.0 * synth.splitmix64(1, n) - 1.0
b = Asp @ xs
for rep in range(3):
    x = b.copy(); F.solve(x)
    print("1 rhs, call", rep, "x err", np.abs(x - xs).max(), flush=True)
X = np.asfortranarray(np.stack([b, 2 * b, -b], axis=1)); F.solve(X)
print("3 rhs contiguous: x err", np.abs(X[:, 1] - 2 * xs).max(), flush=True)
big = np.zeros((n + 5, 3), order='F'); big[:n] = np.stack([b, 2 * b, -b], axis=1)
F.solve(big[:n])
print("3 rhs strided: x err", np.abs(big[:n, 2] + xs).max(), "tail untouched", float(np.abs(big[n:]).max()), flush=True)
