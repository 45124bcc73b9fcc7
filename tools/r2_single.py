import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import rsparse_amd as rs
from rsparse_amd import synth
A = synth.laplacian_2d(1024); n = A[0]
F = rs.CholHandle(A, rs.schol(A, 0))
st = torch.cuda.current_stream().cuda_stream
for k, layout, ld in ((1, rs.api.COL_MAJOR, n), (8, rs.api.ROW_MAJOR, 8), (16, rs.api.ROW_MAJOR, 16), (7, rs.api.ROW_MAJOR, 7)):
    xb = torch.rand(n * k, dtype=torch.float64, device='cuda')
    F.solve_dev(xb.data_ptr(), layout, ld, k, True, st); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): F.solve_dev(xb.data_ptr(), layout, ld, k, True, st)
    e1.record(); torch.cuda.synchronize()
    print(k, "rhs: ms per cholsol solve phase", e0.elapsed_time(e1) / 20, flush=True)
