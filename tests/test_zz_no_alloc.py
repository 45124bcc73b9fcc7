"""rsp_tri_solve_dev_noalloc_f64: the row-major device solve whose scratch rows live in the library's fixed __device__
array (no hipMalloc in the call) must give exactly what rsp_tri_solve_dev_f64 gives -- same kernels, same programme, same
order of every sum -- on two calls in a row, the second with a smaller input."""
import numpy as np
import pytest

import rsparse_amd as rs
from oracle import pyoracle as orc
from rsparse_amd import synth

pytestmark = pytest.mark.gpu


def test_noalloc_entry_point_matches_the_allocating_one():
    torch = pytest.importorskip("torch")
    A = synth.laplacian_2d(96)
    st, L = orc.chol(A, orc.schol(A, 0))
    assert st == orc.OK
    n = L[0]
    stream = torch.cuda.current_stream().cuda_stream
    for kind, name in ((rs.api.LSOLVE, "lsolve"), (rs.api.LTSOLVE, "ltsolve")):
        h = rs.TriHandle(L, kind)
        # before any solve the programme is not on the device: the entry point refuses instead of allocating
        probe = torch.zeros((n, 4), dtype=torch.float64, device="cuda:0")
        with pytest.raises(rs.DeviceError):
            h.solve_dev_noalloc(probe.data_ptr(), 4, 4, stream)
        first = True
        for nrhs in (96, 40):  # two calls in a row, the second with a smaller input
            B = (2.0 * synth.splitmix64(31 + nrhs, n * nrhs) - 1.0).reshape(n, nrhs)
            Xa = torch.tensor(B, device="cuda:0")
            h.solve_dev(Xa.data_ptr(), rs.api.ROW_MAJOR, nrhs, nrhs, stream)  # the existing entry point (uploads the programme)
            torch.cuda.synchronize()
            if first:
                info = h.info()
                assert info["scratch_rows"] > 0  # the kernels really use the scratch buffer
                first = False
            Xn = torch.tensor(B, device="cuda:0")
            torch.cuda.synchronize()
            free_before = torch.cuda.mem_get_info()[0]  # hipMemGetInfo: the device's free bytes, whoever allocates
            h.solve_dev_noalloc(Xn.data_ptr(), nrhs, nrhs, stream)
            torch.cuda.synchronize()
            assert torch.cuda.mem_get_info()[0] >= free_before  # nothing was allocated by the call
            assert torch.equal(Xa, Xn)  # bit for bit
            # a block with a row stride larger than its width: the columns beyond nrhs are not touched
            ld = nrhs + 8
            Xw = torch.full((n, ld), 7.25, dtype=torch.float64, device="cuda:0")
            Xw[:, :nrhs] = torch.tensor(B, device="cuda:0")
            h.solve_dev_noalloc(Xw.data_ptr(), ld, nrhs, stream)
            torch.cuda.synchronize()
            assert torch.equal(Xw[:, :nrhs], Xa)
            assert bool((Xw[:, nrhs:] == 7.25).all())
            want = orc.trisolve(name, L, B[:, 3].copy())
            got = Xn[:, 3].cpu().numpy()
            assert np.max(np.abs(got - want)) / np.max(np.abs(want)) <= 1e-8
        # sizes beyond the fixed array are rejected with an error, not served by an allocation
        too_wide = (4 << 20) // info["scratch_rows"] + 1
        big = torch.zeros((n, too_wide), dtype=torch.float64, device="cuda:0")
        with pytest.raises(rs.DeviceError):
            h.solve_dev_noalloc(big.data_ptr(), too_wide, too_wide, stream)
