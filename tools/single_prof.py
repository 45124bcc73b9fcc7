"""Single right-hand-side cholsol solve phase at C2 (factor resident), a few repetitions: for rocprofv3 kernel traces."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import rsparse_amd as rs
from rsparse_amd import synth

A = synth.laplacian_2d(int(sys.argv[1]) if len(sys.argv) > 1 else 1024)
F = rs.CholHandle(A, rs.schol(A, 0))
b = torch.rand(A[0], dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(4):
    F.solve_dev(b.data_ptr(), rs.api.COL_MAJOR, A[0], 1, True, st)
torch.cuda.synchronize()
