"""Numeric Cholesky at C2 only (no triangular-solve analysis): for rocprofv3 kernel traces of the chol kernels."""
import sys, time
sys.path.insert(0, ".")
import rsparse_amd as rs
from rsparse_amd import synth

arg = sys.argv[1] if len(sys.argv) > 1 else "1024"
A = synth.laplacian_3d(int(arg[2:])) if arg.startswith("3d") else synth.laplacian_2d(int(arg))
S = rs.schol(A, 0)
for _ in range(2):
    F = rs.CholHandle(A, S)
    fi = F.info()
    print({k: fi[k] for k in ("factor_ms", "flops", "chol_levels")})
