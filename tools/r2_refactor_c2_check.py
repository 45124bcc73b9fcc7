"""C2-size refactorisation with OTHER values (chains of 1 024 unknowns inverted on the device): residuals of the solves."""
import sys, numpy as np, scipy.sparse as sp
sys.path.insert(0, '.')
import rsparse_amd as rs
from rsparse_amd import synth
A = synth.laplacian_2d(1024); n = A[0]
F = rs.CholHandle(A, rs.schol(A, 0))
m, nn, p, i, x = A
diag = np.asarray(i, np.int64) == np.repeat(np.arange(nn), np.diff(p))
for s in range(3):
    x2 = x * (1.0 + 0.37 * s)
    x2[diag] += 0.5 * s * synth.splitmix64(77 + s, nn)
    As = (m, nn, p, i, x2)
    if s: F.refactor(As)
    M = sp.csc_matrix((x2, i, p), shape=(n, n))
    xs = 2.0 * synth.splitmix64(5 + s, n * 3).reshape(n, 3) - 1.0
    B = np.asfortranarray(M @ xs)
    X = B.copy(order="F"); F.solve(X)
    x1 = B[:, 1].copy(); F.solve(x1)
    print(s, "x error 3 cols", np.abs(X - xs).max(), "1 col", np.abs(x1 - xs[:, 1]).max(), "residual", np.abs(M @ X - B).max() / np.abs(B).max(), flush=True)
