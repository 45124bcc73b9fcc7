"""Round 3: in-kernel timestamps of the 8-wavefront blocks of the latency-bound levels of a single-RHS solve at C2 (library built
with -DRSP_TRI_CLOCKS; RSP_TRI_CLOCKS=1 makes run_window synchronise after every level and print record / gather / barrier /
finish times, 10 ns ticks, of workgroup 0)."""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
import rsparse_amd as rs  # noqa: E402
from rsparse_amd import synth  # noqa: E402

A = rs.Sprs.from_tuple(synth.laplacian_2d(1024))
S = rs.schol(A, 0)
F = rs.CholHandle(A, S)
b = 2.0 * synth.splitmix64(1, A.n) - 1.0
x = b.copy(); F.solve(x)          # programmes resident
os.environ["RSP_TRI_CLOCKS"] = "1"
x = b.copy(); F.solve(x)
