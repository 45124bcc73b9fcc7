"""Refactor + first solve at C2 with RSP_TIMING=1: where the host time after a refactorisation goes."""
import os, sys, time, numpy as np, torch
os.environ["RSP_TIMING"] = "1"
sys.path.insert(0, '.')
import rsparse_amd as rs
from rsparse_amd import synth
arg = sys.argv[1] if len(sys.argv) > 1 else '1024'
A = synth.laplacian_3d(int(arg[2:])) if arg.startswith('3d') else synth.laplacian_2d(int(arg)); n = A[0]
F = rs.CholHandle(A, rs.schol(A, 0))
X = torch.rand(n, 256, dtype=torch.float64, device='cuda')
st = torch.cuda.current_stream().cuda_stream
t0 = time.time(); F.solve_dev(X.data_ptr(), rs.api.ROW_MAJOR, 256, 256, True, st); torch.cuda.synchronize()
print("first solve (programmes built)", time.time() - t0, flush=True)
for rep in range(2):
    t0 = time.time(); F.refactor(A); t1 = time.time()
    F.solve_dev(X.data_ptr(), rs.api.ROW_MAJOR, 256, 256, True, st); torch.cuda.synchronize()
    print("refactor", t1 - t0, "solve after refactor", time.time() - t1, flush=True)
