"""Stress of the refactorise-and-solve path against the oracle: many patterns, several value sets each, 1 / 5 / 70 columns."""
import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import rsparse_amd as rs
from rsparse_amd import synth
from oracle import pyoracle as orc

def revalue(A, s):
    m, n, p, i, x = A
    x2 = x * (1.0 + 0.4 * synth.splitmix64(1000 + s, 1)[0])
    diag = np.asarray(i, np.int64) == np.repeat(np.arange(n), np.diff(p))
    x2 = x2.copy(); x2[diag] += 0.2 * s * synth.splitmix64(2000 + s, n)
    return (m, n, p, i, x2)

worst = 0.0
cases = [("circuit", lambda k: synth.circuit_spd(900 + 700 * k, 5 + k % 3, 50 + k)) for k in range(5)]
cases += [("lap2d", lambda k: synth.laplacian_2d(60 + 37 * k, 45 + 29 * k)) for k in range(5)]
cases += [("lap3d", lambda k: synth.laplacian_3d(9 + 3 * k)) for k in range(4)]
for idx, (name, gen) in enumerate(cases):
    A0 = gen(idx % 5)
    n = A0[0]
    F = rs.CholHandle(A0, rs.schol(A0, 0))
    B = np.asfortranarray((2.0 * synth.splitmix64(7 + idx, n * 70) - 1.0).reshape(n, 70))
    for s in range(3):
        A = revalue(A0, s) if s else A0
        if s:
            F.refactor(A)
        X = B.copy(order="F"); F.solve(X)
        X5 = np.asfortranarray(B[:, :5].copy()); F.solve(X5)
        x1 = B[:, 69].copy(); F.solve(x1)
        for c, got in ((0, X[:, 0]), (33, X[:, 33]), (69, X[:, 69]), (4, X5[:, 4]), (69, x1)):
            st, want = orc.cholsol(A, B[:, c].copy(), 0)
            err = np.abs(got - want).max() / np.abs(want).max()
            worst = max(worst, err)
            assert st == 0 and err <= 1e-8, (name, idx, s, c, err)
    print(name, idx, n, "ok", flush=True)
print("worst relative error", worst)
