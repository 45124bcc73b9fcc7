"""C3 / C4 solve phase (256 right-hand sides) under the blocked and the supernodal engine: usage r2_cfg_sn.py c3|c4 [size]"""
import sys, time, os, numpy as np, torch
sys.path.insert(0, '.')
import rsparse_amd as rs
from rsparse_amd import synth
which = sys.argv[1]
k = 256
st = torch.cuda.current_stream().cuda_stream
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
if which == 'c3':
    g = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    A = synth.laplacian_3d(g); n = A[0]
    F = rs.CholHandle(A, rs.schol(A, 0)); L = F.factor(); T = (n, n, L.p, L.i.astype(np.int64), L.x)
    pairs = [(T, 0), (T, 1)]
else:
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
    A = synth.banded_unsymmetric(n); S = rs.sqr(A, 1); N = rs.lu(A, S, 1.0)
    pairs = [(N.l.as_tuple(), 0), (N.u.as_tuple(), 2)]
M = rs.MatHandle(A)
for mode in ("0", "1"):
    os.environ["RSP_SUPERNODAL"] = mode
    t0 = time.time()
    hs = [rs.TriHandle(T, kind) for T, kind in pairs]
    X = torch.rand((n, k), dtype=torch.float64, device='cuda')
    ms = timed(lambda: rs.solve_pair_dev(hs[0], hs[1], X.data_ptr(), k, k, st))
    print(which, "supernodal" if mode == "1" else "blocked", f"ms/step {ms:.2f}", "launches", [h.info()["launches"] for h in hs], f"setup {time.time()-t0:.1f}s", flush=True)
