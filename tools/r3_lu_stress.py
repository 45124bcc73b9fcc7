"""Round 3: the GPU numeric LU against the host routine, bit for bit, on a spread of matrices, orderings and tolerances
(whatever path the device check chooses, the factors must be the host routine's)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rsparse_amd as rs  # noqa: E402
from rsparse_amd import synth  # noqa: E402

cases = []
for seed, (n, per_col, band) in enumerate([(500, 3, 4), (2000, 9, 50), (5000, 5, 200), (12000, 7, 30), (30000, 4, 12), (3000, 12, 300), (1000, 2, 999)]):
    cases.append((f"banded n={n} per_col={per_col} band={band}", synth.banded_unsymmetric(n, per_col, band, 40 + seed)))
for n in (300, 1500, 2200):
    cases.append((f"circuit n={n}", synth.circuit_spd(n, 6 + n, 100 + n)))
cases.append(("laplacian_2d 70x45", synth.laplacian_2d(70, 45)))
cases.append(("laplacian_3d 14", synth.laplacian_3d(14)))
for n, dens, seed in ((400, 0.01, 3), (1500, 0.004, 4)):
    cases.append((f"random n={n}", synth.random_sparse(n, dens, seed=seed)))
    cases.append((f"random spd n={n}", synth.random_sparse(n, dens, seed=seed, symmetric_pd=True)))

bad = 0
on_dev = 0
t_start = time.time()
for name, A in cases:
    for order in (-1, 0, 1, 2):
        for tol in (1.0, 0.1):
            S = rs.sqr(A, order)
            os.environ["RSP_LU_GPU"] = "0"
            try:
                H = rs.lu(A, S, tol)
            except rs.NoPivot:
                H = None
            os.environ["RSP_LU_GPU"] = "1"
            try:
                G = rs.lu(A, S, tol)
            except rs.NoPivot:
                G = None
            if (H is None) != (G is None):
                print("MISMATCH (pivot error)", name, order, tol, flush=True)
                bad += 1
                continue
            if H is None:
                continue
            on_dev += int(G.on_device)
            same = (np.array_equal(H.pinv, G.pinv) and np.array_equal(H.l.p, G.l.p) and np.array_equal(H.l.i, G.l.i)
                    and np.array_equal(H.u.p, G.u.p) and np.array_equal(H.u.i, G.u.i)
                    and np.array_equal(H.l.x.view(np.int64), G.l.x.view(np.int64)) and np.array_equal(H.u.x.view(np.int64), G.u.x.view(np.int64)))
            if not same:
                print("MISMATCH", name, order, tol, "on_device", G.on_device, flush=True)
                bad += 1
    print(f"{name}: done ({time.time() - t_start:.0f} s)", flush=True)
print(f"{len(cases) * 8} factorisations, {on_dev} on the device, {bad} mismatches")
sys.exit(1 if bad else 0)
