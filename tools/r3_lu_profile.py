"""Round 3: the GPU numeric LU at C4 (banded 200 000, order 1, tol 1.0) under rocprofv3: one factorisation and two refactorisations."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rsparse_amd as rs  # noqa: E402
from rsparse_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
A = rs.Sprs.from_tuple(synth.banded_unsymmetric(n))
S = rs.sqr(A, 1)
t0 = time.time(); N = rs.lu(A, S, 1.0); t_lu = time.time() - t0
assert N.on_device
A2 = rs.Sprs(A.m, A.n, A.p, A.i, A.x * 1.01)
times = []
for M in (A2, A):
    t0 = time.time(); N.refactor(M, 1.0); times.append(time.time() - t0)
print(json.dumps({"n": n, "nnz_L": int(N.l.p[-1]), "nnz_U": int(N.u.p[-1]), "lu_s": t_lu, "refactor_s": times}), flush=True)
