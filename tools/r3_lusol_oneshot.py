"""Round 3: the one-shot lusol at C4 -- first call (stage timings with RSP_TIMING=1 RSP_LU_TRACE=1), then the same call again."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import rsparse_amd as rs  # noqa: E402
from rsparse_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
A = rs.Sprs.from_tuple(synth.banded_unsymmetric(n))
b = synth.splitmix64(5, A.n)
rs.lib().rsp_set_device(0)
w = rs.Sprs.from_tuple(synth.banded_unsymmetric(5000)); xw = np.ones(5000); rs.lusol(w, xw, 1, 1.0)  # code objects loaded
time.sleep(0.5)
rs.cache_clear()
out = {}
x = b.copy(); t0 = time.time(); rs.lusol(A, x, 1, 1.0); out["first_call_s"] = time.time() - t0
r = rs.gaxpy(A, x, -b)
out["residual_rel"] = float(np.linalg.norm(r) / np.linalg.norm(b))
t0 = time.time()
while rs.cache_stats()["pending"] and time.time() - t0 < 120:
    time.sleep(0.005)
out["background_s"] = time.time() - t0
for k in ("first_hit_s", "second_hit_s"):
    y = b.copy(); t0 = time.time(); rs.lusol(A, y, 1, 1.0); out[k] = time.time() - t0
out["hit_vs_miss_max_rel_diff"] = float(np.max(np.abs(x - y)) / np.max(np.abs(x)))
print(json.dumps(out), flush=True)
rs.cache_clear()
