"""GPU tests added in round 2 (run on the MI355X box with -m gpu), all through the C ABI:
   * gaxpy (lib.rs:411-421) on one vector and on device-resident blocks, against the oracle,
   * multi-right-hand-side lsolve / ltsolve and chol on 3-D Laplacians (wide separators: the >1024-source partial sums
     and the inverted chains in pieces dominate there), against the oracle,
   * BASELINE.json's full-size configurations C3 (80^3), C4 (200 000 unsymmetric) and C5 (4 096 systems) through
     size-independent properties (residual of every column, wide block against one-column solves, linearity),
   * the device-resident lusol solve, handles shared between host threads, and the hardened boundary.
Tolerances are north_star's: ||x - x_ref||inf / ||x_ref||inf <= 1e-8, relative residual <= 1e-10."""
import threading

import numpy as np
import pytest

import rsparse_amd as rs
from helpers import load_sprs
from oracle import pyoracle as orc
from rsparse_amd import synth

pytestmark = pytest.mark.gpu

REL_X = 1e-8
REL_RES = 1e-10


def rel_inf(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def _torch():
    return pytest.importorskip("torch")


def _block_residuals(A, X_dev, B_dev, nrhs):
    """max over columns of ||A x - b|| / ||b|| for row-major device blocks, computed on the device (rsp_mat_gaxpy_dev_f64)."""
    torch = _torch()
    M = rs.MatHandle(A)
    R = B_dev.clone()
    M.gaxpy_dev(X_dev.data_ptr(), nrhs, R.data_ptr(), nrhs, nrhs, rs.api.ROW_MAJOR, -1.0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return float((R.norm(dim=0) / B_dev.norm(dim=0)).max())


# ---- gaxpy ---------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("name", ["lap2d", "band", "lusol_6", "dup"])
def test_gaxpy_vs_oracle(name):
    torch = _torch()
    if name == "lap2d":
        A = synth.laplacian_2d(37, 23)
    elif name == "band":
        A = synth.banded_unsymmetric(1500, band=25)
    elif name == "lusol_6":
        A = load_sprs("lusol_6.sprs")
    else:  # duplicate entries in one column are summed, as the reference's loop does
        A = (3, 3, np.array([0, 3, 5, 6], np.int64), np.array([0, 0, 2, 1, 1, 2], np.int64), np.array([1.0, 2.0, -1.0, 4.0, 0.5, 3.0]))
    m, n = A[0], A[1]
    x = 2.0 * synth.splitmix64(3, n) - 1.0
    y = 2.0 * synth.splitmix64(4, m) - 1.0
    want = orc.gaxpy(A, x, y)
    got = rs.gaxpy(A, x, y)
    assert np.array_equal(got, want)  # same products added in the same order: the same bits
    # blocks, both layouts, beta = 1 / -1 / 0
    k = 70
    Xh = (2.0 * synth.splitmix64(5, n * k) - 1.0).reshape(n, k)
    Yh = (2.0 * synth.splitmix64(6, m * k) - 1.0).reshape(m, k)
    M = rs.MatHandle(A)
    st = torch.cuda.current_stream().cuda_stream
    for beta in (1.0, -1.0, 0.0):
        ref = np.stack([orc.gaxpy(A, Xh[:, c].copy(), np.zeros(m)) for c in range(k)], axis=1) + beta * Yh
        X = torch.tensor(Xh, device="cuda:0")
        Y = torch.tensor(Yh, device="cuda:0")
        M.gaxpy_dev(X.data_ptr(), k, Y.data_ptr(), k, k, rs.api.ROW_MAJOR, beta, st)
        torch.cuda.synchronize()
        assert rel_inf(Y.cpu().numpy(), ref) <= 1e-14
        Xc = torch.tensor(np.asfortranarray(Xh).T.copy(), device="cuda:0")  # k x n contiguous = n x k column-major
        Yc = torch.tensor(np.asfortranarray(Yh).T.copy(), device="cuda:0")
        M.gaxpy_dev(Xc.data_ptr(), n, Yc.data_ptr(), m, k, rs.api.COL_MAJOR, beta, st)
        torch.cuda.synchronize()
        assert rel_inf(Yc.cpu().numpy().T, ref) <= 1e-14


# ---- 3-D Laplacians: multi-right-hand-side solves and chol against the oracle ----------------------------------------

@pytest.mark.parametrize("grid,nrhs", [(24, 64), (24, 256), (28, 64)])
def test_multi_rhs_on_3d_laplacian_vs_oracle(grid, nrhs):
    A = synth.laplacian_3d(grid)
    n = A[0]
    So = orc.schol(A, 0)
    st, Lo = orc.chol(A, So)
    assert st == orc.OK
    S = rs.schol(A, 0)
    F = rs.CholHandle(A, S)
    L = F.factor()
    assert np.array_equal(L.p, Lo[2]) and np.array_equal(L.i.astype(np.int64), Lo[3])  # the pattern is exact
    assert rel_inf(L.x, Lo[4]) <= 1e-11
    T = Lo
    B = np.asfortranarray((2.0 * synth.splitmix64(31, n * nrhs) - 1.0).reshape(nrhs, n).T)
    cols = sorted(set([0, 1, nrhs // 3, nrhs // 2, nrhs - 2, nrhs - 1] + list(range(5, nrhs, 37))))
    for kind, kid in (("lsolve", 0), ("ltsolve", 1)):
        want = np.stack([orc.trisolve(kind, T, B[:, c].copy()) for c in cols], axis=1)
        h = rs.TriHandle(T, kid)
        X = B.copy(order="F")
        h.solve(X)
        assert rel_inf(X[:, cols], want) <= REL_X
        assert h.info()["scratch_rows"] >= 0 and h.info()["launches"] > 0
    # cholsol on the whole block: every column's residual, three columns against the oracle
    X = B.copy(order="F")
    F.solve(X)
    import scipy.sparse as sp
    Asp = sp.csc_matrix((A[4], A[3], A[2]), shape=(n, n))
    R = Asp @ X - B
    assert np.max(np.linalg.norm(R, axis=0) / np.linalg.norm(B, axis=0)) <= REL_RES
    for c in (0, nrhs // 2, nrhs - 1):
        st, want = orc.cholsol(A, B[:, c].copy(), 0)
        assert st == orc.OK and rel_inf(X[:, c], want) <= REL_X
    info = F.info()
    assert info["lsolve_launches"] > 0 and info["lsolve_tasks"] > 0 and info["ltsolve_launches"] > 0 and info["ltsolve_tasks"] > 0


# ---- BASELINE.json's full-size configurations: properties that need no reference run ---------------------------------

def test_full_size_c3_laplacian_3d_80():
    """C3: 3-D 7-point Laplacian 80^3 (n = 512 000, nnz(L) = 535.8 M), Cholesky factor + 256 right-hand sides."""
    torch = _torch()
    A = synth.laplacian_3d(80)
    n, k = A[0], 256
    S = rs.schol(A, 0)
    F = rs.CholHandle(A, S)
    assert F.info()["lnz"] == 535760267  # an integer result: the same on every run and every machine
    # X = 512 000 x 256, column-major, U(-1, 1) from SplitMix64 seed 3 (column c = the stream's c-th run of n doubles)
    B = synth.splitmix64_torch(3, n * k, "cuda:0", -1.0, 1.0).reshape(k, n).T.contiguous()
    X = B.clone()
    st = torch.cuda.current_stream().cuda_stream
    F.solve_dev(X.data_ptr(), rs.api.ROW_MAJOR, k, k, True, st)
    torch.cuda.synchronize()
    assert _block_residuals(A, X, B, k) <= REL_RES  # all 256 columns
    # columns of the wide block against one-column solves, and linearity
    for c in (7, 200):
        x1 = B[:, c].clone().contiguous()
        F.solve_dev(x1.data_ptr(), rs.api.COL_MAJOR, n, 1, True, st)
        torch.cuda.synchronize()
        assert float((x1 - X[:, c]).abs().max() / X[:, c].abs().max()) <= 1e-10
    mix = (2.0 * B[:, 1] - 3.0 * B[:, 2]).contiguous()
    F.solve_dev(mix.data_ptr(), rs.api.COL_MAJOR, n, 1, True, st)
    torch.cuda.synchronize()
    want = 2.0 * X[:, 1] - 3.0 * X[:, 2]
    assert float((mix - want).abs().max() / want.abs().max()) <= 1e-10


def test_full_size_c4_unsymmetric_200k():
    """C4: unsymmetric 200 000 x 200 000, ~10 entries per column, lusol with partial pivoting tol = 1.0."""
    torch = _torch()
    import scipy.sparse as sp
    A = synth.banded_unsymmetric(200000)
    n, k = A[0], 256
    Asp = sp.csc_matrix((A[4], A[3], A[2]), shape=(n, n))
    b = synth.splitmix64(5, n)
    x = b.copy()
    rs.lusol(A, x, 1, 1.0)
    assert np.linalg.norm(Asp @ x - b) / np.linalg.norm(b) <= REL_RES
    S = rs.sqr(A, 1)
    N = rs.lu(A, S, 1.0)
    # b ~ U(0, 1) from SplitMix64 seed 5; the block's column c is the stream's c-th run of n doubles (column 0 = b)
    B = synth.splitmix64_torch(5, n * k, "cuda:0").reshape(k, n).T.contiguous()
    assert np.array_equal(B[:, 0].cpu().numpy(), b)
    X = B.clone()
    st = torch.cuda.current_stream().cuda_stream
    N.solve_dev(X.data_ptr(), rs.api.ROW_MAJOR, k, k, st)
    torch.cuda.synchronize()
    assert _block_residuals(A, X, B, k) <= REL_RES
    x1 = B[:, 77].cpu().numpy().copy()
    N.solve(x1)  # the host entry point, one column
    assert rel_inf(X[:, 77].cpu().numpy(), x1) <= 1e-10
    mix = (2.0 * B[:, 1] - 3.0 * B[:, 2]).contiguous()
    N.solve_dev(mix.data_ptr(), rs.api.COL_MAJOR, n, 1, st)
    torch.cuda.synchronize()
    want = 2.0 * X[:, 1] - 3.0 * X[:, 2]
    assert float((mix - want).abs().max() / want.abs().max()) <= 1e-10


def test_full_size_c5_batch_4096():
    """C5: 4 096 independent SPD systems of 2 000 unknowns (circuit-style, one pattern), cholsol on the whole batch."""
    import scipy.sparse as sp
    n, nsys = 2000, 4096
    A0 = synth.circuit_spd(n, 6, 100)
    nnz = len(A0[4])
    vals = np.empty((nsys, nnz))
    B = np.empty((nsys, n))
    for s in range(nsys):
        vals[s] = synth.circuit_spd(n, 6, 100 + s)[4]
        B[s] = synth.splitmix64(1000 + s, n)
    got = B.copy()
    status = rs.cholsol_batch(A0, vals, got, 0)
    assert not status.any()
    worst = 0.0
    for s in range(nsys):  # every system's residual (a 2 000 x 2 000 product each)
        Asp = sp.csc_matrix((vals[s], A0[3], A0[2]), shape=(n, n))
        worst = max(worst, np.linalg.norm(Asp @ got[s] - B[s]) / np.linalg.norm(B[s]))
    assert worst <= REL_RES
    for s in (0, 1777, 4095):  # against single calls and the oracle
        one = B[s].copy()
        As = (n, n, A0[2], A0[3], vals[s])
        rs.cholsol(As, one, 0)
        assert rel_inf(got[s], one) <= 1e-11
        st, want = orc.cholsol(As, B[s].copy(), 0)
        assert st == orc.OK and rel_inf(got[s], want) <= REL_X


def test_chol_refactor_twice_vs_oracle():
    """rsp_chol_refactor_f64: new values on the analysed pattern, factor and solves against the oracle each time."""
    n = 1500
    mats = [synth.circuit_spd(n, 6, 100 + s) for s in range(3)]
    S = rs.schol(mats[0], 0)
    F = rs.CholHandle(mats[0], S)
    b = 2.0 * synth.splitmix64(77, n) - 1.0
    for k, A in enumerate(mats + [mats[0]]):  # back to the first values at the end
        if k:
            F.refactor(A)
        st, Lo = orc.chol(A, orc.schol(A, 0))
        L = F.factor()
        assert st == orc.OK and np.array_equal(L.i.astype(np.int64), Lo[3]) and rel_inf(L.x, Lo[4]) <= 1e-11
        x = b.copy()
        F.solve(x)
        st, want = orc.cholsol(A, b.copy(), 0)
        assert rel_inf(x, want) <= REL_X
        X = np.asfortranarray(np.stack([b, 2.0 * b, -b], axis=1))
        F.solve(X)  # the multi-column programme is rebuilt from the new values too
        assert rel_inf(X[:, 1], 2.0 * want) <= REL_X
    bad = list(mats[1])
    bad[4] = bad[4].copy()
    bad[4][bad[2][17]:bad[2][18]] *= -1.0  # column 17 gets a negative diagonal
    with pytest.raises(rs.NotPositiveDefinite):
        F.refactor(tuple(bad))
    F.refactor(mats[2])  # and the handle recovers
    x = b.copy()
    F.solve(x)
    st, want = orc.cholsol(mats[2], b.copy(), 0)
    assert rel_inf(x, want) <= REL_X
    with pytest.raises(rs.DeviceError):
        F.refactor(synth.circuit_spd(n + 1, 6, 100))  # another size: bad input


@pytest.mark.parametrize("matrix", ["circuit", "grid"])
@pytest.mark.parametrize("refill", ["device", "host", "rebuild"])
def test_refactor_keeps_or_rebuilds_the_solve_programmes(refill, matrix, monkeypatch):
    """After rsp_chol_refactor_f64 the resident solve programmes receive the new values in place -- by kernels straight
    from the factor on the device (default), or on the host through the same recipe (RSP_REFILL_DEVICE=0) -- or are
    analysed and emitted again (RSP_REFILL=0): wide, narrow and single right-hand sides against the oracle every way,
    and the launch summary of the programmes does not change.  "grid" is a 2-D mesh whose separators become chains that
    are replaced by their inverses (widths that are no multiples of 64), "circuit" has none."""
    if refill == "host":
        monkeypatch.setenv("RSP_REFILL_DEVICE", "0")
    if refill == "rebuild":
        monkeypatch.setenv("RSP_REFILL", "0")
    if matrix == "circuit":
        n = 2600
        mats = [synth.circuit_spd(n, 6, 300 + s) for s in range(3)]
    else:
        base = synth.laplacian_2d(150, 131)
        n = base[0]
        mats = []
        for s in range(3):  # same pattern, other values: scaled, and the diagonal pushed up unevenly
            m, nn, p, i, x = base
            x2 = x * (1.0 + 0.5 * s)
            diag = np.asarray(i, np.int64) == np.repeat(np.arange(nn), np.diff(p))
            x2[diag] += 0.3 * s * synth.splitmix64(400 + s, nn)
            mats.append((m, nn, p, i, x2))
    F = rs.CholHandle(mats[0], rs.schol(mats[0], 0))
    B = np.asfortranarray((2.0 * synth.splitmix64(301, n * 40) - 1.0).reshape(n, 40))
    first_info = None
    for k, A in enumerate(mats + [mats[1]]):
        if k:
            F.refactor(A)
        X = B.copy(order="F")
        F.solve(X)                       # 40 columns: the blocked programme, 64-column wavefronts
        x1 = B[:, 3].copy()
        F.solve(x1)                      # one column: 16-column wavefronts on the same programme
        for c in (0, 3, 39):
            st, want = orc.cholsol(A, B[:, c].copy(), 0)
            assert st == orc.OK and rel_inf(X[:, c], want) <= REL_X, (k, c)
        assert rel_inf(x1, orc.cholsol(A, B[:, 3].copy(), 0)[1]) <= REL_X
        fi = F.info()
        keys = ("lsolve_launches", "lsolve_tasks", "ltsolve_launches", "ltsolve_tasks")
        if first_info is None:
            first_info = [fi[key] for key in keys]
            assert all(v > 0 for v in first_info)
        else:
            assert [fi[key] for key in keys] == first_info


def test_factor_view_is_the_resident_factor():
    """rsp_chol_factor_view: the host pattern and the device values the solves read, without a copy; equal to
    rsp_chol_get_factor's copy, and it follows a refactor."""
    n = 1200
    A, A2 = synth.circuit_spd(n, 6, 31), synth.circuit_spd(n, 6, 32)
    F = rs.CholHandle(A, rs.schol(A, 0))
    for M in (A, A2):
        if M is A2:
            F.refactor(M)
        L = F.factor()
        Lp, Li, dLx = F.factor_view()
        assert np.array_equal(Lp, L.p) and np.array_equal(Li, L.i) and dLx
        host = np.empty(len(L.x))
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        assert hip.hipMemcpy(ctypes.c_void_p(host.ctypes.data), ctypes.c_void_p(dLx), ctypes.c_size_t(8 * len(L.x)), 2) == 0  # device to host
        assert np.array_equal(host, L.x)
    # the text and binary formats carry the factor exactly
    import tempfile, os
    with tempfile.TemporaryDirectory() as d:
        L.save(os.path.join(d, "l.sprs"))
        L.save_bin(os.path.join(d, "l.bin"))
        for back in (rs.Sprs.load(os.path.join(d, "l.sprs")), rs.Sprs.load_bin(os.path.join(d, "l.bin"))):
            assert np.array_equal(back.p, L.p) and np.array_equal(back.i, L.i) and np.array_equal(back.x.view(np.uint64), L.x.view(np.uint64))
            x = 2.0 * synth.splitmix64(5, n) - 1.0
            want = orc.trisolve("lsolve", L.as_tuple(), x)
            rs.lsolve(back, x)
            assert rel_inf(x, want) <= REL_X


def test_host_blocks_with_a_leading_dimension_vs_oracle():
    """rsp_chol_solve_f64 / rsp_tri_solve_f64 on host blocks whose leading dimension exceeds n (the C ABI's ldb / ldx):
    every column against the oracle, the rows between the columns untouched."""
    n, k, pad = 900, 3, 7
    A = synth.circuit_spd(n, 6, 41)
    F = rs.CholHandle(A, rs.schol(A, 0))
    cols = [2.0 * synth.splitmix64(50 + j, n) - 1.0 for j in range(k)]
    big = np.full((n + pad, k), 123.0, order="F")
    for j in range(k):
        big[:n, j] = cols[j]
    rs.api._check(rs.api.lib().rsp_chol_solve_f64(F._h, big.ctypes.data, n + pad, k))
    assert np.all(big[n:] == 123.0)
    for j in range(k):
        st, want = orc.cholsol(A, cols[j].copy(), 0)
        assert st == orc.OK and rel_inf(big[:n, j], want) <= REL_X
    L = F.factor()
    T = rs.TriHandle(L, rs.api.LSOLVE)
    big = np.full((n + pad, k), -7.0, order="F")
    for j in range(k):
        big[:n, j] = cols[j]
    rs.api._check(rs.api.lib().rsp_tri_solve_f64(T._h, big.ctypes.data, n + pad, k))
    assert np.all(big[n:] == -7.0)
    for j in range(k):
        assert rel_inf(big[:n, j], orc.trisolve("lsolve", L.as_tuple(), cols[j])) <= REL_X


@pytest.mark.parametrize("layout", ["row", "col"])
def test_refine_dev_vs_oracle(layout):
    """rsp_chol_refine_dev_f64: from X = 0 one step IS the solve (against the oracle's cholsol); from a perturbed solution
    one step comes back to it; a second step leaves the residual where it is; padding of the blocks is not touched."""
    import torch
    n, k = 1300, 5
    A = synth.circuit_spd(n, 6, 61)
    F = rs.CholHandle(A, rs.schol(A, 0))
    M = rs.MatHandle(A)
    Bh = 2.0 * synth.splitmix64(62, n * k).reshape(n, k) - 1.0
    want = np.stack([orc.cholsol(A, Bh[:, j].copy(), 0)[1] for j in range(k)], axis=1)
    row = layout == "row"
    lay, ld, shape = (rs.api.ROW_MAJOR, k + 3, (n, k + 3)) if row else (rs.api.COL_MAJOR, n + 3, (k, n + 3))
    def block(vals):
        t = torch.full(shape, 77.0, dtype=torch.float64, device="cuda")
        if row:
            t[:, :k] = torch.tensor(vals, device="cuda")
        else:
            t[:, :n] = torch.tensor(vals.T.copy(), device="cuda")
        return t
    def values(t):
        return t[:, :k].cpu().numpy() if row else t[:, :n].cpu().numpy().T
    def padding(t):
        return t[:, k:] if row else t[:, n:]
    st = torch.cuda.current_stream().cuda_stream
    B, W = block(Bh), block(np.zeros((n, k)))
    X = block(np.zeros((n, k)))
    F.refine_dev(M, B.data_ptr(), X.data_ptr(), W.data_ptr(), lay, ld, k, 1, st)
    torch.cuda.synchronize()
    assert rel_inf(values(X), want) <= REL_X and bool((padding(X) == 77.0).all()) and bool((padding(B) == 77.0).all())
    noise = 1e-5 * (2.0 * synth.splitmix64(63, n * k).reshape(n, k) - 1.0)
    X = block(want + noise)
    F.refine_dev(M, B.data_ptr(), X.data_ptr(), W.data_ptr(), lay, ld, k, 1, st)
    torch.cuda.synchronize()
    once = values(X).copy()
    assert rel_inf(once, want) <= REL_X
    F.refine_dev(M, B.data_ptr(), X.data_ptr(), W.data_ptr(), lay, ld, k, 2, st)
    torch.cuda.synchronize()
    Ad = np.zeros((n, n))
    m_, n_, p, i, x = A
    for j in range(n):
        Ad[i[p[j]:p[j + 1]].astype(np.int64), j] = x[p[j]:p[j + 1]]
    res = lambda Xv: np.abs(Ad @ Xv - Bh).max() / np.abs(Bh).max()
    assert res(values(X)) <= max(2.0 * res(once), 1e-14) and rel_inf(values(X), want) <= REL_X
    # bad arguments: aliased blocks, a matrix of another size
    with pytest.raises(rs.DeviceError):
        F.refine_dev(M, B.data_ptr(), X.data_ptr(), X.data_ptr(), lay, ld, k, 1, st)
    with pytest.raises(rs.DeviceError):
        F.refine_dev(rs.MatHandle(synth.circuit_spd(n + 1, 6, 61)), B.data_ptr(), X.data_ptr(), W.data_ptr(), lay, ld, k, 1, st)


# ---- device-resident lusol solve -------------------------------------------------------------------------------------

@pytest.mark.parametrize("order", [1, -1])
def test_lu_solve_dev_vs_oracle(order):
    torch = _torch()
    A = synth.banded_unsymmetric(3000, band=30)
    n, k = A[0], 48
    S = rs.sqr(A, order)
    N = rs.lu(A, S, 1.0)
    Bh = (2.0 * synth.splitmix64(8, n * k) - 1.0).reshape(n, k)
    cols = (0, 17, 47)
    want = {}
    for c in cols:
        st, w = orc.lusol(A, Bh[:, c].copy(), order, 1.0)
        assert st == orc.OK
        want[c] = w
    st_ = torch.cuda.current_stream().cuda_stream
    X = torch.tensor(Bh, device="cuda:0")
    N.solve_dev(X.data_ptr(), rs.api.ROW_MAJOR, k, k, st_)
    torch.cuda.synchronize()
    Xr = X.cpu().numpy()
    Xc = torch.tensor(Bh.T.copy(), device="cuda:0")  # k x n contiguous = column-major n x k
    N.solve_dev(Xc.data_ptr(), rs.api.COL_MAJOR, n, k, st_)
    torch.cuda.synchronize()
    Xcm = Xc.cpu().numpy().T
    for c in cols:
        assert rel_inf(Xr[:, c], want[c]) <= REL_X
        assert rel_inf(Xcm[:, c], want[c]) <= REL_X


# ---- thread safety: two host threads on two handles, and on one shared handle ----------------------------------------

def test_handles_from_two_host_threads():
    A = synth.laplacian_2d(60, 45)
    n = A[0]
    st, L = orc.chol(A, orc.schol(A, 0))
    assert st == orc.OK
    nrhs = 24
    B = np.asfortranarray((2.0 * synth.splitmix64(41, n * nrhs) - 1.0).reshape(nrhs, n).T)
    want = {kind: np.stack([orc.trisolve(kind, L, B[:, c].copy()) for c in range(nrhs)], axis=1) for kind in ("lsolve", "ltsolve")}
    shared = rs.TriHandle(L, 0)
    own = [rs.TriHandle(L, 0), rs.TriHandle(L, 1)]
    errors = []

    def work(t):
        try:
            for it in range(6):
                X = B.copy(order="F")
                shared.solve(X)  # both threads, one handle
                if rel_inf(X, want["lsolve"]) > REL_X:
                    errors.append(("shared", t, it))
                Y = B.copy(order="F")
                own[t].solve(Y)  # each thread its own handle
                if rel_inf(Y, want["lsolve" if t == 0 else "ltsolve"]) > REL_X:
                    errors.append(("own", t, it))
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_shared_handle_on_two_streams():
    """Consecutive device calls on one handle from different streams: the second waits for the first's kernels."""
    torch = _torch()
    A = synth.laplacian_2d(60, 45)
    n = A[0]
    st, L = orc.chol(A, orc.schol(A, 0))
    nrhs = 64
    Bh = (2.0 * synth.splitmix64(43, n * nrhs) - 1.0).reshape(n, nrhs)
    want = np.stack([orc.trisolve("lsolve", L, Bh[:, c].copy()) for c in (0, 31, 63)], axis=1)
    h = rs.TriHandle(L, 0)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    X1 = torch.tensor(Bh, device="cuda:0")
    X2 = torch.tensor(Bh, device="cuda:0")
    torch.cuda.synchronize()
    for _ in range(4):
        X1.copy_(torch.tensor(Bh, device="cuda:0"))
        X2.copy_(torch.tensor(Bh, device="cuda:0"))
        torch.cuda.synchronize()
        h.solve_dev(X1.data_ptr(), rs.api.ROW_MAJOR, nrhs, nrhs, s1.cuda_stream)
        h.solve_dev(X2.data_ptr(), rs.api.ROW_MAJOR, nrhs, nrhs, s2.cuda_stream)
        torch.cuda.synchronize()
        assert rel_inf(X1.cpu().numpy()[:, [0, 31, 63]], want) <= REL_X
        assert rel_inf(X2.cpu().numpy()[:, [0, 31, 63]], want) <= REL_X


# ---- every kernel variant an environment switch selects, against the oracle ------------------------------------------

ITEMS_EVERYWHERE = {"RSP_ITEM_WAVES": "1", "RSP_ITEM_MIN_SOURCES": "1", "RSP_ITEM_MIN_FILL": "0"}
VARIANTS = [
    {},                                                   # defaults
    {"RSP_LEAN_WGS": "1"},                                # every level through the lean variant
    {"RSP_LEAN_WGS": "1000000"},                          # every level through the deep-ring variant, 16-column wavefronts
    {"RSP_LEAN_WGS": "1000000", "RSP_LAT_NARROW": "0"},   # ... 64-column wavefronts
    {"RSP_SMALL_WAVES": "1"},                             # small-block tasks always in their own lean launch
    {"RSP_SMALL_WAVES": "1000000000"},                    # ... never
    {"RSP_CLUSTER_CPW": "1"}, {"RSP_CLUSTER_CPW": "4"},   # column chunks of a cluster per workgroup
    {"RSP_PAIR_OVERLAP": "0"}, {"RSP_PAIR_PARTS": "3"}, {"RSP_PAIR_PARTS": "4"},
    {"RSP_TRI_BLOCKED": "0"},                             # the plain level programme for blocks of right-hand sides
    {"RSP_TRI_SINGLE": "1"},                              # ... for one right-hand side
    {"RSP_SPLIT_SOURCES": "32", "RSP_SPLIT8_SOURCES": "64", "RSP_MAX_BLOCK_SOURCES": "128"},  # more split blocks, partial sums
    {"RSP_TASK_BLOCKS": "3"},                             # several dependent blocks per wavefront (long unsplit blocks)
    # round 3: wide items over every chain (however short and sparse), run as items (64 x 64 register tiles) on every level
    {**ITEMS_EVERYWHERE},
    {**ITEMS_EVERYWHERE, "RSP_ITEM_KS": "2"},             # ... K-steps of an item shared by two wavefronts
    {**ITEMS_EVERYWHERE, "RSP_ITEM_KS": "4"},             # ... by four
    {**ITEMS_EVERYWHERE, "RSP_ITEM_WAVES": "0"},          # ... emitted but never run: always the twin 16-row blocks
    {**ITEMS_EVERYWHERE, "RSP_ITEM_ROWS": "32"},          # items of two row tiles
    {**ITEMS_EVERYWHERE, "RSP_SORT_ROWS": "0", "RSP_MAX_BLOCK_SOURCES": "64"},  # rows in natural order, items over partial sums
    {"RSP_ITEM_ROWS": "16"},                              # no items at all
]


@pytest.mark.parametrize("env", VARIANTS, ids=lambda e: ",".join(f"{k[4:]}={v}" for k, v in e.items()) or "defaults")
def test_every_switch_variant_vs_oracle(env, monkeypatch):
    torch = _torch()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    A = synth.laplacian_2d(70, 53)
    st, L = orc.chol(A, orc.schol(A, 0))
    B = synth.banded_unsymmetric(2500, band=40)
    st2, f = orc.lu(B, 1, 1.0)
    assert st == st2 == orc.OK
    stream = torch.cuda.current_stream().cuda_stream
    for T, kinds in ((L, ("lsolve", "ltsolve")), (f["L"], ("lsolve",)), (f["U"], ("usolve", "utsolve"))):
        n = T[0]
        for kind in kinds:
            h = rs.TriHandle(T, {"lsolve": 0, "ltsolve": 1, "usolve": 2, "utsolve": 3}[kind])
            for nrhs in (1, 4, 8, 16, 20, 128):  # 4..16: aligned blocks narrower than one 16-column chunk ... one 64-column chunk
                Bh = (2.0 * synth.splitmix64(90 + nrhs, n * nrhs) - 1.0).reshape(n, nrhs)
                cols = sorted({0, nrhs // 2, nrhs - 1})
                want = np.stack([orc.trisolve(kind, T, Bh[:, c].copy()) for c in cols], axis=1)
                X = torch.tensor(Bh, device="cuda:0")
                h.solve_dev(X.data_ptr(), rs.api.ROW_MAJOR, nrhs, nrhs, stream)
                torch.cuda.synchronize()
                assert rel_inf(X.cpu().numpy()[:, cols], want) <= REL_X, (kind, nrhs)
    # the pair entry point (column parts on several streams) on the Cholesky factor
    n, nrhs = L[0], 192
    Bh = (2.0 * synth.splitmix64(7, n * nrhs) - 1.0).reshape(n, nrhs)
    want = np.stack([orc.trisolve("ltsolve", L, orc.trisolve("lsolve", L, Bh[:, c].copy())) for c in (0, 100, 191)], axis=1)
    lo, up = rs.TriHandle(L, 0), rs.TriHandle(L, 1)
    X = torch.tensor(Bh, device="cuda:0")
    rs.solve_pair_dev(lo, up, X.data_ptr(), nrhs, nrhs, stream)
    torch.cuda.synchronize()
    assert rel_inf(X.cpu().numpy()[:, [0, 100, 191]], want) <= REL_X


def test_wide_items_vs_oracle(monkeypatch):
    """The wide items (dev::solve_items) on a deeper factor than the variant test's: every level with items runs them as
    items, 64 / 128 / 256 columns (one to four chunks per workgroup), both sweeps, against the oracle on spread columns and
    against the same handle's twin-block path (RSP_ITEM_WAVES=0 builds nothing new: the choice is made per launch)."""
    torch = _torch()
    monkeypatch.setenv("RSP_ITEM_MIN_SOURCES", "1")  # (read when the programme is built: items over every chain)
    monkeypatch.setenv("RSP_ITEM_MIN_FILL", "0")
    A = synth.laplacian_2d(160, 160)
    n = A[0]
    st, L = orc.chol(A, orc.schol(A, 0))
    assert st == orc.OK
    stream = torch.cuda.current_stream().cuda_stream
    for kind, name in ((0, "lsolve"), (1, "ltsolve")):
        h = rs.TriHandle(L, kind)
        for nrhs in (64, 128, 256):
            Bh = (2.0 * synth.splitmix64(300 + nrhs, n * nrhs) - 1.0).reshape(n, nrhs)
            cols = sorted({0, 1, 33, 63, nrhs // 2, nrhs - 1})
            want = np.stack([orc.trisolve(name, L, Bh[:, c].copy()) for c in cols], axis=1)
            got = {}
            for waves in ("1", "0"):
                monkeypatch.setenv("RSP_ITEM_WAVES", waves)
                X = torch.tensor(Bh, device="cuda:0")
                h.solve_dev(X.data_ptr(), rs.api.ROW_MAJOR, nrhs, nrhs, stream)
                torch.cuda.synchronize()
                got[waves] = X.cpu().numpy()
                assert rel_inf(got[waves][:, cols], want) <= REL_X, (name, nrhs, waves)
            assert rel_inf(got["1"], got["0"]) <= 1e-12, (name, nrhs)
        info = h.info()
        assert info["items"] > 0, info


@pytest.mark.parametrize("env", [{"RSP_CHOL_GROUPED": "0"}, {"RSP_CHOL_TINY": "0"}, {"RSP_CHOL_FUSE_WAVES": "0"}, {"RSP_CHOL_FUSE_WAVES": "100000"},
                                 {"RSP_CHOL_SHORT_GROUPS": "0"}, {"RSP_CHOL_BLOCK_PANELS": "1"}],
                         ids=lambda e: ",".join(f"{k[4:]}={v}" for k, v in e.items()))
def test_every_cholesky_variant_vs_oracle(env, monkeypatch):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for A in (synth.laplacian_2d(90, 70), synth.laplacian_3d(14), synth.circuit_spd(2500)):
        st, Lo = orc.chol(A, orc.schol(A, 0))
        L = rs.CholHandle(A, rs.schol(A, 0)).factor()
        assert st == orc.OK and np.array_equal(L.i.astype(np.int64), Lo[3]) and rel_inf(L.x, Lo[4]) <= 1e-11


def test_batch_sharded_over_devices():
    """rsp_set_devices: the batch's systems go to the listed devices in contiguous shares, one host thread each.  On a
    one-GPU box the same device is listed twice and three times (two / three shares): same results as one share."""
    n, nsys = 600, 11
    mats = [synth.circuit_spd(n, 6, 100 + s) for s in range(nsys)]
    values = np.ascontiguousarray(np.stack([m[4] for m in mats]))
    values[7, mats[0][2][5]:mats[0][2][6]] *= -1.0  # system 7 is not positive definite
    B = np.ascontiguousarray(np.stack([synth.splitmix64(1000 + s, n) for s in range(nsys)]))
    one = B.copy()
    st1 = rs.cholsol_batch(mats[0], values, one, 0)
    assert st1.tolist() == [0] * 7 + [1] + [0] * 3
    try:
        for devs in ([0, 0], [0, 0, 0]):
            rs.set_devices(devs)
            got = B.copy()
            st = rs.cholsol_batch(mats[0], values, got, 0)
            assert st.tolist() == st1.tolist() and np.array_equal(got, one)
        with pytest.raises(rs.DeviceError):
            rs.set_devices([0, 99])
    finally:
        rs.set_devices([])


def test_rhs_blocks_sharded_over_devices():
    """rsp_set_devices and the host-pointer solves (rsp_chol_solve_f64, rsp_tri_solve_f64): column blocks per listed device,
    every device but the handle's own on a REPLICA of the resident programmes copied device to device (no second analysis).
    On the one-GPU box device 0 is listed two and three times; the shares run the same programme on the same columns' data,
    so the result equals the unsharded one bit for bit.  A refactorisation drops the replicas."""
    _torch()
    A = synth.laplacian_2d(90, 77)
    n, k = A[0], 200
    F = rs.CholHandle(A, rs.schol(A, 0))
    B = np.asfortranarray((2.0 * synth.splitmix64(41, n * k) - 1.0).reshape(k, n).T)
    want = B.copy(order="F")
    F.solve(want)
    st, w0 = orc.cholsol(A, B[:, 5].copy(), 0)
    assert rel_inf(want[:, 5], w0) <= REL_X
    st2, L = orc.chol(A, orc.schol(A, 0))
    T = rs.TriHandle(L, 0)
    tw = B.copy(order="F")
    T.solve(tw)
    try:
        for devs in ([0, 0], [0, 0, 0]):
            rs.set_devices(devs)
            got = B.copy(order="F")
            F.solve(got)
            assert np.array_equal(got, want), devs
            tg = B.copy(order="F")
            T.solve(tg)
            assert np.array_equal(tg, tw), devs
        # new values: the replicas follow (they are made again from the refactorised programmes)
        A2 = (A[0], A[1], A[2], A[3], A[4] * 1.5)
        F.refactor(A2)
        got = B.copy(order="F")
        F.solve(got)
        assert rel_inf(got, want / 1.5) <= 1e-12
        # fewer columns than two per device: the plain path
        small = B[:, :3].copy(order="F")
        F.solve(small)
        assert rel_inf(small, want[:, :3] / 1.5) <= 1e-12
    finally:
        rs.set_devices([])


# ---- qrsol: the consumer of usolve / utsolve (lib.rs:927-956) -------------------------------------------
from helpers import qr_case, reference_cases, to_dense  # noqa: E402

QR_SOLVER = reference_cases()["solver"]
QR_CASES = [k for k, v in QR_SOLVER.items() if v["call"].startswith("qrsol")]


@pytest.mark.parametrize("name", QR_CASES)
def test_qrsol_reference_cases(name):
    """Every qrsol test of the reference (tests/solver_tests.rs:687-968): its own assertions on the vector left in b,
    and the same vector from the oracle's qrsol to REL_X (both branches: m >= n through usolve, m < n through utsolve)."""
    c = QR_SOLVER[name]
    A = qr_case(c)
    m, n = A[0], A[1]
    b0 = np.asarray(c["b"], np.float64)
    got = b0.copy()
    rs.qrsol(A, got, c["order"])
    if "x" in c:
        assert np.abs(got[:len(c["x"])] - np.asarray(c["x"])).max() <= c["tol"]
    if "residual" in c["check"]:
        assert np.abs(to_dense(A) @ got[:n] - b0[:m]).max() <= c["res_tol"]
    if "agree" in c["check"]:
        other = b0.copy()
        if "cholsol" in c["call"]:
            rs.cholsol(A, other, c["order_chol"])
        else:
            rs.lusol(A, other, c["order_lu"], c["lutol"])
        assert np.abs(got - other).max() <= c["tol"]
    want = orc.qrsol(A, b0, c["order"])
    assert rel_inf(got[:n], want[:n]) <= REL_X


def test_qr_handle_solves_many():
    """rsp_qr_f64 once, rsp_qr_solve_f64 for several right-hand sides, least squares (tall matrix) and minimum norm
    (its transpose factored, transposed = 1), against the oracle's qrsol and numpy's lstsq."""
    tall = qr_case(dict(QR_SOLVER["qrsol_9"]))          # 201 x 200
    wide = qr_case(dict(QR_SOLVER["qrsol_8"]))          # 200 x 201
    S = rs.sqr(tall, 2, True)
    N = rs.qr(tall, S)
    D = to_dense(tall)
    for seed in (1, 2, 3):
        b = 2.0 * synth.splitmix64(seed, 201) - 1.0
        got = b.copy()
        N.solve(got)
        want = orc.qrsol(tall, b, 2)
        assert rel_inf(got[:200], want[:200]) <= REL_X
        ls = np.linalg.lstsq(D, b, rcond=None)[0]
        assert rel_inf(got[:200], ls) <= 1e-6
        # the same handle factors wide' = tall: minimum-norm solution of wide y = c
        c = np.zeros(201)
        c[:200] = 2.0 * synth.splitmix64(10 + seed, 200) - 1.0
        y = c.copy()
        N.solve(y, transposed=True)
        want = orc.qrsol(wide, c, 2)
        assert rel_inf(y, want) <= REL_X
        assert np.abs(to_dense(wide) @ y - c[:200]).max() <= 1e-9


# ---- round 3: the analysis cache behind the one-shot cholsol (rsp_cholsol_f64) -------------------------------------------

def _wait_cache_idle(limit_s=120.0):
    import time
    t0 = time.time()
    while rs.cache_stats()["pending"] and time.time() - t0 < limit_s:
        time.sleep(0.02)
    assert rs.cache_stats()["pending"] == 0


def test_cholsol_analysis_cache(monkeypatch):
    """Same pattern, new values: a hit (refactorisation on the cached analysis) with the oracle's answer; one changed
    index: a miss; RSP_ANALYSIS_CACHE=0: no cache at all; the entry budget evicts the least recently used pattern."""
    _torch()
    monkeypatch.setenv("RSP_ANALYSIS_CACHE", "2")
    rs.cache_clear()
    base = rs.cache_stats()
    A = synth.laplacian_2d(90, 70)
    n, p, i, x = A[0], A[2], A[3], A[4]
    b = 2.0 * synth.splitmix64(11, n) - 1.0

    def check(M, tag):
        st, want = orc.cholsol(M, b.copy(), 0)
        assert st == orc.OK
        got = b.copy()
        rs.cholsol(M, got, 0)
        assert rel_inf(got, want) <= REL_X, tag

    check(A, "first call")
    s = rs.cache_stats()
    assert s["misses"] == base["misses"] + 1 and s["hits"] == base["hits"]
    _wait_cache_idle()
    assert rs.cache_stats()["entries"] == 1
    # values only: hits, each against the oracle's own cholsol of those values
    for k in range(3):
        xk = x.copy()
        diag = np.flatnonzero(np.repeat(np.arange(n), np.diff(p)) == i)
        xk[diag] += 0.25 * (k + 1)
        check((A[0], A[1], p, i, xk), f"hit {k}")
    s = rs.cache_stats()
    assert s["hits"] == base["hits"] + 3 and s["misses"] == base["misses"] + 1
    # a hit that is not positive definite reports it, and the next hit works again
    xbad = x.copy()
    xbad[np.flatnonzero(np.repeat(np.arange(n), np.diff(p)) == i)[n // 2]] = -4.0
    with pytest.raises(rs.NotPositiveDefinite):
        rs.cholsol((A[0], A[1], p, i, xbad), b.copy(), 0)
    check(A, "hit after a failed hit")
    # one changed index (an off-diagonal entry moved to a free slot of its column): another pattern, a miss
    B = synth.laplacian_2d(90, 71)
    before = rs.cache_stats()
    bb = 2.0 * synth.splitmix64(12, B[0]) - 1.0
    st, want = orc.cholsol(B, bb.copy(), 0)
    got = bb.copy()
    rs.cholsol(B, got, 0)
    assert rel_inf(got, want) <= REL_X
    assert rs.cache_stats()["misses"] == before["misses"] + 1
    _wait_cache_idle()
    # a third pattern: the budget of two entries evicts the least recently used one (A was used last before B)
    Cm = synth.laplacian_2d(64, 64)
    bc = 2.0 * synth.splitmix64(13, Cm[0]) - 1.0
    rs.cholsol(Cm, bc.copy(), 0)
    _wait_cache_idle()
    s = rs.cache_stats()
    assert s["entries"] == 2 and s["evicted"] >= base["evicted"] + 1
    # another ordering of a cached pattern is another key
    before = rs.cache_stats()
    got = bc.copy()
    rs.cholsol(Cm, got, -1)
    st, want = orc.cholsol(Cm, bc.copy(), -1)
    assert rel_inf(got, want) <= REL_X and rs.cache_stats()["misses"] == before["misses"] + 1
    _wait_cache_idle()
    # switched off: neither hits nor misses are counted, nothing is kept
    rs.cache_clear()
    monkeypatch.setenv("RSP_ANALYSIS_CACHE", "0")
    before = rs.cache_stats()
    check(A, "cache off")
    check(A, "cache off, again")
    s = rs.cache_stats()
    assert s["entries"] == 0 and s["hits"] == before["hits"] and s["misses"] == before["misses"]


def test_cholsol_analysis_cache_two_threads(monkeypatch):
    """Two host threads calling cholsol with one pattern at once: an entry is out of the cache while a call uses it, so one
    thread hits and the other takes the plain path; both get the oracle's answer."""
    _torch()
    import threading
    monkeypatch.setenv("RSP_ANALYSIS_CACHE", "2")
    rs.cache_clear()
    A = synth.laplacian_2d(80, 80)
    n = A[0]
    b = 2.0 * synth.splitmix64(21, n) - 1.0
    st, want = orc.cholsol(A, b.copy(), 0)
    rs.cholsol(A, b.copy(), 0)
    _wait_cache_idle()
    out = [None, None]
    errs = []

    def work(k):
        try:
            for _ in range(3):
                g = b.copy()
                rs.cholsol(A, g, 0)
                out[k] = g
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs
    assert rel_inf(out[0], want) <= REL_X and rel_inf(out[1], want) <= REL_X
    _wait_cache_idle()
    rs.cache_clear()


def test_handle_after_a_failed_refactorisation():
    """A refactorisation that fails (not positive definite) leaves NO usable factor: solve, factor download and factor
    view report NotPositiveDefinite until a refactorisation succeeds; values on another pattern of the same size are bad
    input, not a silently wrong factor (round-2 advisor finding)."""
    _torch()
    A = synth.laplacian_2d(40, 30)
    n, p, i, x = A[0], A[2], A[3], A[4]
    F = rs.CholHandle(A, rs.schol(A, 0))
    b = 2.0 * synth.splitmix64(31, n) - 1.0
    st, want = orc.cholsol(A, b.copy(), 0)
    X = b.copy()
    F.solve(X)
    assert rel_inf(X, want) <= REL_X
    diag = np.flatnonzero(np.repeat(np.arange(n), np.diff(p)) == i)
    xbad = x.copy()
    xbad[diag[n // 3]] = -1.0
    with pytest.raises(rs.NotPositiveDefinite):
        F.refactor((A[0], A[1], p, i, xbad))
    with pytest.raises(rs.NotPositiveDefinite):
        F.solve(b.copy())
    with pytest.raises(rs.NotPositiveDefinite):
        F.factor()
    # same size, another pattern: two row indices of one column swapped places with their neighbours' rows
    i2 = i.copy()
    c = n // 2
    cols = np.repeat(np.arange(n), np.diff(p))
    free = sorted(set(range(n)) - set(i[p[c]:p[c + 1]].tolist()))
    off = [k for k in range(p[c], p[c + 1]) if i[k] != c][0]
    i2[off] = free[0]
    with pytest.raises(rs.DeviceError):
        F.refactor((A[0], A[1], p, i2, x))
    F.refactor(A)
    X = b.copy()
    F.solve(X)
    assert rel_inf(X, want) <= REL_X
