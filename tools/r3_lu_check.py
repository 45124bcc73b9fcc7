"""GPU numeric LU (static pivots) against the host routine: bit-exactness and time.  Run on a GPU box."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rsparse_amd as rs  # noqa: E402
os.environ["RSP_LU_TRACE"] = "1"
from rsparse_amd import synth  # noqa: E402


def run(name, A, order, tol):
    S = rs.sqr(A, order)
    os.environ["RSP_LU_GPU"] = "0"
    t = time.time(); H = rs.lu(A, S, tol); th = time.time() - t
    _, depth = rs.lu_static_pattern(A, S)
    os.environ["RSP_LU_GPU"] = "1"
    rs.lu(A, S, tol)  # warm (module load)
    t = time.time(); G = rs.lu(A, S, tol); tg = time.time() - t
    tr = float("nan")
    if G.on_device:
        A2 = rs.Sprs(A.m, A.n, A.p, A.i, A.x * 1.01)
        t = time.time(); G.refactor(A2, tol); tr = time.time() - t
        G.refactor(A, tol)
    os.environ["RSP_LU_GPU"] = "2"
    t = time.time(); D = rs.lu(A, S, tol); td = time.time() - t
    same = (np.array_equal(H.pinv, G.pinv) and np.array_equal(H.l.p, G.l.p) and np.array_equal(H.l.i, G.l.i)
            and np.array_equal(H.u.p, G.u.p) and np.array_equal(H.u.i, G.u.i)
            and np.array_equal(H.l.x.view(np.int64), G.l.x.view(np.int64)) and np.array_equal(H.u.x.view(np.int64), G.u.x.view(np.int64)))
    if not same and G.on_device:
        pat = (np.array_equal(H.pinv, G.pinv) and np.array_equal(H.l.p, G.l.p) and np.array_equal(H.l.i, G.l.i)
               and np.array_equal(H.u.p, G.u.p) and np.array_equal(H.u.i, G.u.i))
        dl = np.flatnonzero(H.l.x.view(np.int64) != G.l.x.view(np.int64))
        du = np.flatnonzero(H.u.x.view(np.int64) != G.u.x.view(np.int64))
        print("   patterns equal:", pat, " L values differing:", dl.size, " U values differing:", du.size,
              " max |dL|:", float(np.max(np.abs(H.l.x - G.l.x))), " max |dU|:", float(np.max(np.abs(H.u.x - G.u.x))),
              " first:", dl[:3], du[:3], flush=True)
    print(f"{name:28s} n={A.n:7d} order={order:2d} lnz={len(H.l.i):9d} depth={depth:6d} host {th*1e3:9.1f} ms  gpu {tg*1e3:9.1f} ms"
          f" (on_device={G.on_device})  refactor {tr*1e3:8.1f} ms  auto {td*1e3:9.1f} ms (on_device={D.on_device})  bit-identical={same}", flush=True)
    return same or not G.on_device


ok = True
ok &= run("banded 3000", rs.Sprs.from_tuple(synth.banded_unsymmetric(3000, 5, 10, 1)), 1, 1.0)
ok &= run("banded 50000", rs.Sprs.from_tuple(synth.banded_unsymmetric(50000)), 1, 1.0)
ok &= run("laplacian_2d 60", rs.Sprs.from_tuple(synth.laplacian_2d(60)), 1, 1.0)
ok &= run("laplacian_2d 300", rs.Sprs.from_tuple(synth.laplacian_2d(300)), 1, 1.0)
ok &= run("laplacian_3d 20", rs.Sprs.from_tuple(synth.laplacian_3d(20)), 1, 1.0)
ok &= run("circuit 2000", rs.Sprs.from_tuple(synth.circuit_spd(2000, 6, 100)), 1, 1.0)
ok &= run("banded 200000 (C4)", rs.Sprs.from_tuple(synth.banded_unsymmetric(200000)), 1, 1.0)
ok &= run("circuit 2000 natural", rs.Sprs.from_tuple(synth.circuit_spd(2000, 6, 100)), -1, 0.001)
print("ALL OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
