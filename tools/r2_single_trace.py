"""One single-RHS cholsol solve phase at C2 under the kernel trace: launches, workgroups, durations."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import rsparse_amd as rs
from rsparse_amd import synth
A = synth.laplacian_2d(1024); n = A[0]
F = rs.CholHandle(A, rs.schol(A, 0))
st = torch.cuda.current_stream().cuda_stream
xb = torch.rand(n, dtype=torch.float64, device='cuda')
for _ in range(3):
    F.solve_dev(xb.data_ptr(), rs.api.COL_MAJOR, n, 1, True, st); torch.cuda.synchronize()
