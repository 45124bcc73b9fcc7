"""Round 3, VERDICT item 1d: one step of the multi-RHS solve (and the single-RHS cholsol solve phase) captured in a HIP graph
(torch.cuda.graph around the unchanged C-ABI calls) against the same calls launched eagerly.  C2 by default."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import rsparse_amd as rs  # noqa: E402
from rsparse_amd import synth  # noqa: E402

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nrhs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
rs.lib().rsp_set_device(0)
n = grid * grid
A = synth.laplacian_2d(grid)
S = rs.schol(A, 0)
F = rs.CholHandle(A, S)
L = F.factor()
Lt = (n, n, L.p, L.i.astype(np.int64), L.x)
lower = rs.TriHandle(Lt, rs.api.LSOLVE)
upper = rs.TriHandle(Lt, rs.api.LTSOLVE)
X = synth.splitmix64_torch(2, n * nrhs, dev, -1.0, 1.0).reshape(n, nrhs)
X0 = X.clone()
out = {"grid": grid, "nrhs": nrhs}


def timed(fn, reps):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(torch.cuda.current_stream())
    for _ in range(reps):
        fn()
    e1.record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, (time.perf_counter() - t0) * 1e3 / reps


side = torch.cuda.Stream()
with torch.cuda.stream(side):
    st = side.cuda_stream

    def eager():
        rs.solve_pair_dev(lower, upper, X.data_ptr(), nrhs, nrhs, st)

    X.copy_(X0); eager(); torch.cuda.synchronize()
    want = X.clone()
    for _ in range(3):
        eager()
    out["pair_eager_ms"], out["pair_eager_wall_ms"] = timed(eager, 10)
    try:
        g = torch.cuda.CUDAGraph()
        X.copy_(X0)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            eager()
        torch.cuda.synchronize()
        X.copy_(X0); g.replay(); torch.cuda.synchronize()
        out["pair_graph_bitwise_equal_to_eager"] = bool(torch.equal(X, want))
        for _ in range(3):
            g.replay()
        out["pair_graph_ms"], out["pair_graph_wall_ms"] = timed(g.replay, 10)
    except Exception as e:  # noqa: BLE001
        out["pair_graph_error"] = repr(e)[:300]
    print(json.dumps(out), flush=True)

    # single right-hand side, the cholsol solve phase on the resident factor (P, lsolve, ltsolve, P')
    xb = synth.splitmix64_torch(7, n, dev, -1.0, 1.0)
    x0 = xb.clone()

    def eager1():
        F.solve_dev(xb.data_ptr(), rs.api.COL_MAJOR, n, 1, True, st)

    eager1(); torch.cuda.synchronize()
    want1 = xb.clone()
    for _ in range(3):
        eager1()
    out["one_rhs_eager_ms"], out["one_rhs_eager_wall_ms"] = timed(eager1, 20)
    try:
        g1 = torch.cuda.CUDAGraph()
        xb.copy_(x0)
        torch.cuda.synchronize()
        with torch.cuda.graph(g1, stream=side):
            eager1()
        torch.cuda.synchronize()
        xb.copy_(x0); g1.replay(); torch.cuda.synchronize()
        out["one_rhs_graph_bitwise_equal_to_eager"] = bool(torch.equal(xb, want1))
        for _ in range(3):
            g1.replay()
        out["one_rhs_graph_ms"], out["one_rhs_graph_wall_ms"] = timed(g1.replay, 20)
    except Exception as e:  # noqa: BLE001
        out["one_rhs_graph_error"] = repr(e)[:300]
print(json.dumps(out), flush=True)
