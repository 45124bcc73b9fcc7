import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rsparse_amd as rs
from rsparse_amd import synth
os.environ["RSP_LU_TRACE"] = "1"
for n in (8, 64, 300, 3000):
    A = rs.Sprs.from_tuple(synth.banded_unsymmetric(n, 3, min(6, n - 1), 1))
    S = rs.sqr(A, -1)
    os.environ["RSP_LU_GPU"] = "0"
    H = rs.lu(A, S, 1.0)
    os.environ["RSP_LU_GPU"] = "1"
    print("n", n, "host done", flush=True)
    try:
        t = time.time(); G = rs.lu(A, S, 1.0); dt = time.time() - t
        print("  gpu", G.on_device, "%.1f ms" % (dt * 1e3), "L same", np.array_equal(H.l.x.view(np.int64), G.l.x.view(np.int64)),
              "U same", np.array_equal(H.u.x.view(np.int64), G.u.x.view(np.int64)), flush=True)
        if n <= 8:
            print(H.l.x, G.l.x, H.u.x, G.u.x, sep="\n")
    except Exception as e:
        print("  FAILED:", e, flush=True)
