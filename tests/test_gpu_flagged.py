"""RSP_FLAGGED_LEVELS=1 (opt-in, one right-hand side): runs of consecutive latency-bound levels of the blocked programme go out as
ONE launch each (dev::solve_levels_flagged: tickets in level order, a flag per workgroup, host-built producer lists, coherent
row accesses).  Same programme, same order of every sum: the result must equal the default path's bit for bit, and the
oracle's to the usual tolerance."""
import numpy as np
import pytest

import rsparse_amd as rs
from oracle import pyoracle as orc
from rsparse_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("maker", [lambda: synth.laplacian_2d(160), lambda: synth.laplacian_3d(16), lambda: synth.circuit_spd(3000)])
def test_flagged_levels_equal_the_level_launches(maker, monkeypatch):
    A = maker()
    st, L = orc.chol(A, orc.schol(A, 0))
    assert st == orc.OK
    n = L[0]
    b = 2.0 * synth.splitmix64(77, n) - 1.0
    for kind, name in ((rs.api.LSOLVE, "lsolve"), (rs.api.LTSOLVE, "ltsolve")):
        monkeypatch.delenv("RSP_FLAGGED_LEVELS", raising=False)
        plain = rs.TriHandle(L, kind)
        x0 = b.copy()
        plain.solve(x0)
        monkeypatch.setenv("RSP_FLAGGED_LEVELS", "1")
        flagged = rs.TriHandle(L, kind)
        for _ in range(3):  # the flags carry an epoch from call to call
            x1 = b.copy()
            flagged.solve(x1)
            assert np.array_equal(x0.view(np.int64), x1.view(np.int64))
        want = orc.trisolve(name, L, b.copy())
        assert np.max(np.abs(x1 - want)) / np.max(np.abs(want)) <= 1e-8
        # several columns take the ordinary launches on the same handle
        B = (2.0 * synth.splitmix64(78, n * 3) - 1.0).reshape(3, n).T.copy(order="F")
        X = B.copy(order="F")
        flagged.solve(X)
        for j in range(3):
            wj = orc.trisolve(name, L, B[:, j].copy())
            assert np.max(np.abs(X[:, j] - wj)) / np.max(np.abs(wj)) <= 1e-8
