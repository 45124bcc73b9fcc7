"""Timing probe of BASELINE.json's full-size configurations C3, C4, C5 (stage times), to size the -m gpu property tests."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import rsparse_amd as rs
from rsparse_amd import synth
which = sys.argv[1]
t = time.time()
def lap(msg):
    global t
    print(f"{msg}: {time.time() - t:.2f}s", flush=True); t = time.time()
if which == 'c3':
    g = int(sys.argv[2]) if len(sys.argv) > 2 else 80
    A = synth.laplacian_3d(g); n = A[0]; lap(f"generate n={n}")
    S = rs.schol(A, 0); lap(f"schol lnz={S.lnz}")
    F = rs.CholHandle(A, S); lap("chol analyse+factor"); print(F.info(), flush=True)
    k = 256
    M = rs.MatHandle(A); lap("mat")
    X = torch.rand((n, k), dtype=torch.float64, device='cuda') * 2 - 1
    B = X.clone()
    st = torch.cuda.current_stream().cuda_stream
    F.solve_dev(X.data_ptr(), rs.api.ROW_MAJOR, k, k, True, st); torch.cuda.synchronize(); lap("first solve (programme build)")
    R = B.clone()
    M.gaxpy_dev(X.data_ptr(), k, R.data_ptr(), k, k, rs.api.ROW_MAJOR, -1.0, st); torch.cuda.synchronize(); lap("residual")
    print("max rel residual", float((R.norm(dim=0) / B.norm(dim=0)).max()), flush=True)
    X2 = B.clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); F.solve_dev(X2.data_ptr(), rs.api.ROW_MAJOR, k, k, True, st); e1.record(); torch.cuda.synchronize()
    print("second solve ms", e0.elapsed_time(e1), "info", F.info(), flush=True)
elif which == 'c4':
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
    A = synth.banded_unsymmetric(n); lap(f"generate n={n}")
    S = rs.sqr(A, 1); lap("sqr")
    N = rs.lu(A, S, 1.0); lap("lu (host)")
    b = synth.splitmix64(5, n); x = b.copy()
    N.solve(x); lap("solve 1")
    import scipy.sparse as sp
    Asp = sp.csc_matrix((A[4], A[3], A[2]), shape=(n, n))
    print("res", np.linalg.norm(Asp @ x - b) / np.linalg.norm(b), flush=True)
    k = 256
    X = torch.rand((n, k), dtype=torch.float64, device='cuda'); B = X.clone()
    st = torch.cuda.current_stream().cuda_stream
    N.solve_dev(X.data_ptr(), rs.api.ROW_MAJOR, k, k, st); torch.cuda.synchronize(); lap("solve_dev 256 (build)")
    M = rs.MatHandle(A); R = B.clone()
    M.gaxpy_dev(X.data_ptr(), k, R.data_ptr(), k, k, rs.api.ROW_MAJOR, -1.0, st); torch.cuda.synchronize()
    print("max rel residual", float((R.norm(dim=0) / B.norm(dim=0)).max()), flush=True)
    X2 = B.clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); N.solve_dev(X2.data_ptr(), rs.api.ROW_MAJOR, k, k, st); e1.record(); torch.cuda.synchronize()
    print("second solve ms", e0.elapsed_time(e1), flush=True)
elif which == 'c5':
    nsys = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    n = 2000
    A0 = synth.circuit_spd(n, 6, 100)
    nnz = len(A0[4])
    vals = np.empty((nsys, nnz)); B = np.empty((nsys, n))
    for s in range(nsys):
        vals[s] = synth.circuit_spd(n, 6, 100 + s)[4]; B[s] = synth.splitmix64(1000 + s, n)
    lap(f"generate {nsys} systems")
    got = B.copy()
    status = rs.cholsol_batch(A0, vals, got, 0); lap("cholsol_batch")
    print("bad", int((status != 0).sum()), flush=True)
    import scipy.sparse as sp
    worst = 0.0
    for s in range(0, nsys, max(1, nsys // 64)):
        Asp = sp.csc_matrix((vals[s], A0[3], A0[2]), shape=(n, n))
        worst = max(worst, np.linalg.norm(Asp @ got[s] - B[s]) / np.linalg.norm(B[s]))
    lap("residuals"); print("worst", worst)
