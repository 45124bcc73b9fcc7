"""GPU parity tests (run on the MI355X box with -m gpu): the HIP path, called through the C ABI, against
   (a) the reference's golden vectors, (b) the CPU oracle on the same seeded inputs.
Tolerances are north_star's: ||x - x_ref||inf / ||x_ref||inf <= 1e-8, relative residual <= 1e-10;
integer structure (pattern of L) is compared exactly."""
import numpy as np
import pytest

import rsparse_amd as rs
from helpers import case_matrix, load_sprs, reference_cases
from oracle import pyoracle as orc
from rsparse_amd import synth

pytestmark = pytest.mark.gpu

CASES = reference_cases()
SOLVER = CASES["solver"]
REL_X = 1e-8
REL_RES = 1e-10


def rel_inf(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def _abs_close(a, b, tol):
    a, b = np.asarray(a), np.asarray(b)
    assert np.all(np.abs(a[: len(b)] - b[: len(a)]) <= tol), float(np.max(np.abs(a - b)))


# ---- (a) the reference's own tests, through the product API -----------------------------------------

@pytest.mark.parametrize("name", [k for k, v in SOLVER.items() if v["call"] in ("lsolve", "ltsolve", "usolve", "utsolve")])
def test_reference_triangular_tests(name):
    c = SOLVER[name]
    A = rs.Sprs.from_vec(c["dense"])
    b = np.array(c["b"])
    getattr(rs, c["call"])(A, b)
    _abs_close(b, c["x"], c["tol"])
    assert rel_inf(b, orc.trisolve(c["call"], A.as_tuple(), c["b"])) <= REL_X


@pytest.mark.parametrize("name", [k for k, v in SOLVER.items() if v["call"] in ("cholsol", "lusol")])
def test_reference_driver_tests(name):
    c = SOLVER[name]
    A = rs.Sprs.from_tuple(case_matrix(c))
    b = np.array(c["b"], dtype=np.float64)
    if c["check"] == "error":
        with pytest.raises(rs.NotPositiveDefinite):
            rs.cholsol(A, b, c["order"])
        return
    if c["call"] == "cholsol":
        rs.cholsol(A, b, c["order"])
        _, want = orc.cholsol(A.as_tuple(), c["b"], c["order"])
    else:
        rs.lusol(A, b, c["order"], c["lutol"])
        _, want = orc.lusol(A.as_tuple(), c["b"], c["order"], c["lutol"])
    if c["check"] == "x":
        _abs_close(b, c["x"], c["tol"])
    elif c["check"] == "x_bitwise":
        # README.md:182. The GPU solves fuse multiply-adds, so bitwise equality is not expected; 1e-13 relative is.
        assert rel_inf(b, np.array(c["x"])) <= 1e-13
    elif c["check"] == "residual":
        _abs_close(orc.gaxpy(A.as_tuple(), b, np.zeros(len(b))), c["b"], c["tol"])
    if name != "lusol_3":  # ill-conditioned (the reference itself only asks for 0.5)
        assert rel_inf(b, want) <= REL_X


def test_reference_allsol_1():
    c = SOLVER["allsol_1"]
    A = rs.Sprs.from_tuple(case_matrix(c))
    b1, b2 = np.array(c["b"]), np.array(c["b"])
    rs.cholsol(A, b1, c["order_chol"])
    rs.lusol(A, b2, c["order_lu"], c["lutol"])
    _abs_close(b1, b2, c["tol"])


# ---- (b) oracle parity on seeded synthetic inputs ----------------------------------------------------

SPD = {
    "lap2d_40x33": lambda: synth.laplacian_2d(40, 33),
    "lap2d_96": lambda: synth.laplacian_2d(96),
    "lap3d_12": lambda: synth.laplacian_3d(12),
    "circuit_2000": lambda: synth.circuit_spd(2000),
    "cholsol_5": lambda: load_sprs("cholsol_5.sprs"),
    "randspd_400": lambda: synth.random_sparse(400, 0.01, seed=21, symmetric_pd=True),
}


@pytest.mark.parametrize("name", sorted(SPD))
@pytest.mark.parametrize("order", [0, -1])
def test_chol_pattern_exact_values_close(name, order):
    A = SPD[name]()
    So = orc.schol(A, order)
    st, Lo = orc.chol(A, So)
    assert st == orc.OK
    S = rs.schol(A, order)
    N = rs.chol(A, S)
    assert N.l.p.tolist() == Lo[2].tolist()                      # L.p == cp
    assert N.l.i.astype(np.int64).tolist() == Lo[3].tolist()     # diagonal first, rows ascending
    assert np.max(np.abs(N.l.x - Lo[4]) / np.maximum(np.abs(Lo[4]), 1e-300)) <= 1e-9
    assert rel_inf(N.l.x, Lo[4]) <= 1e-12


@pytest.mark.parametrize("name", sorted(SPD))
def test_cholsol_vs_oracle(name):
    A = SPD[name]()
    n = A[0]
    xstar = 2.0 * synth.splitmix64(1, n) - 1.0
    b = orc.gaxpy(A, xstar, np.zeros(n))
    st, want = orc.cholsol(A, b, 0)
    got = b.copy()
    rs.cholsol(A, got, 0)
    assert rel_inf(got, want) <= REL_X
    r = orc.gaxpy(A, got, -b)
    assert np.linalg.norm(r) / np.linalg.norm(b) <= REL_RES


def test_chol_upper_triangle_only_semantics():
    """cholsol_7 (tests/solver_tests.rs:672): a non-symmetric input 'works' because only i <= j is read."""
    A = load_sprs("cholsol_7.sprs")
    S = rs.schol(A, 0)
    N = rs.chol(A, S)
    st, Lo = orc.chol(A, orc.schol(A, 0))
    assert N.l.i.astype(np.int64).tolist() == Lo[3].tolist()
    assert rel_inf(N.l.x, Lo[4]) <= 1e-12


def test_chol_duplicate_entries_last_one_wins():
    """ereach assigns x[i] = a.x[p] (lib.rs:2003): for duplicate entries the last stored one is used."""
    m, n, p, i, x = synth.laplacian_2d(6)
    # duplicate the first entry of every column with a different value placed BEFORE the original
    pi, ii, xi = [0], [], []
    for j in range(n):
        rows, vals = i[p[j]:p[j + 1]], x[p[j]:p[j + 1]]
        ii += [rows[0]] + rows.tolist()
        xi += [vals[0] + 7.0] + vals.tolist()
        pi.append(len(ii))
    A = (m, n, np.array(pi, np.int64), np.array(ii, np.int64), np.array(xi))
    st, Lo = orc.chol(A, orc.schol(A, -1))
    N = rs.chol(A, rs.schol(A, -1))
    assert st == orc.OK and rel_inf(N.l.x, Lo[4]) <= 1e-12


def test_chol_not_positive_definite():
    A = list(synth.laplacian_2d(8))
    x = A[4].copy()
    x[A[2][20]:A[2][21]] *= -1.0  # column 20 now has a negative diagonal
    A[4] = x
    A = tuple(A)
    st, _ = orc.chol(A, orc.schol(A, 0))
    assert st == orc.NOT_POSITIVE_DEFINITE
    with pytest.raises(rs.NotPositiveDefinite):
        rs.chol(A, rs.schol(A, 0))


def _factors():
    A = synth.laplacian_2d(50, 37)
    st, L = orc.chol(A, orc.schol(A, 0))
    B = synth.banded_unsymmetric(3000, band=30)
    st2, f = orc.lu(B, 1, 1.0)
    C = load_sprs("lusol_6.sprs")
    st3, g = orc.lu(C, 1, 1e-12)
    assert st == st2 == st3 == orc.OK
    return {"chol_L": L, "band_L": f["L"], "band_U": f["U"], "lusol6_L": g["L"], "lusol6_U": g["U"]}


FACTORS = None


def factors():
    global FACTORS
    if FACTORS is None:
        FACTORS = _factors()
    return FACTORS


KINDS = [("lsolve", "chol_L"), ("ltsolve", "chol_L"), ("lsolve", "band_L"), ("ltsolve", "band_L"), ("usolve", "band_U"),
         ("utsolve", "band_U"), ("lsolve", "lusol6_L"), ("usolve", "lusol6_U"), ("utsolve", "lusol6_U")]
KIND_ID = {"lsolve": 0, "ltsolve": 1, "usolve": 2, "utsolve": 3}


@pytest.mark.parametrize("kind,fac", KINDS)
def test_trisolve_single_rhs_vs_oracle(kind, fac):
    T = factors()[fac]
    n = T[0]
    b = 2.0 * synth.splitmix64(5, n) - 1.0
    want = orc.trisolve(kind, T, b)
    got = b.copy()
    getattr(rs, kind)(T, got)
    assert rel_inf(got, want) <= REL_X


@pytest.mark.parametrize("mode", ["0", "1", "2"])
@pytest.mark.parametrize("kind,fac", KINDS)
def test_one_shot_trisolve_both_programmes(kind, fac, mode, monkeypatch):
    """One-shot calls run rows with flags in one persistent launch (solve_rows_flagged, the default and RSP_SINGLE_USE=2);
    0 forces the blocked programme, 1 the plain level programme; all must give the oracle's result."""
    monkeypatch.setenv("RSP_SINGLE_USE", mode)
    T = factors()[fac]
    b = 2.0 * synth.splitmix64(6, T[0]) - 1.0
    want = orc.trisolve(kind, T, b)
    got = b.copy()
    getattr(rs, kind)(T, got)
    assert rel_inf(got, want) <= REL_X


@pytest.mark.parametrize("mode", ["0", "1", "2"])
def test_one_shot_drivers_both_programmes(mode, monkeypatch):
    monkeypatch.setenv("RSP_SINGLE_USE", mode)
    A = synth.laplacian_2d(40, 33)
    b = orc.gaxpy(A, 2.0 * synth.splitmix64(1, A[0]) - 1.0, np.zeros(A[0]))
    for order in (0, -1):
        st, want = orc.cholsol(A, b, order)
        got = b.copy()
        rs.cholsol(A, got, order)
        assert st == orc.OK and rel_inf(got, want) <= REL_X
    B = synth.banded_unsymmetric(2000, band=20)
    c = synth.splitmix64(5, B[0])
    st, want = orc.lusol(B, c, 1, 1.0)
    got = c.copy()
    rs.lusol(B, got, 1, 1.0)
    assert st == orc.OK and rel_inf(got, want) <= REL_X


@pytest.mark.parametrize("kind,fac", KINDS)
@pytest.mark.parametrize("nrhs", [2, 7, 64, 130, 256])
def test_trisolve_multi_rhs_vs_oracle(kind, fac, nrhs):
    T = factors()[fac]
    n = T[0]
    B = np.asfortranarray((2.0 * synth.splitmix64(7, n * nrhs) - 1.0).reshape(nrhs, n).T)
    want = np.stack([orc.trisolve(kind, T, B[:, r]) for r in range(nrhs)], axis=1)
    h = rs.TriHandle(T, KIND_ID[kind])
    X = B.copy(order="F")
    h.solve(X)
    assert rel_inf(X, want) <= REL_X
    # linearity (size-independent property): solve(a*b1 + b2) == a*solve(b1) + solve(b2)
    if nrhs >= 2:
        y = np.asfortranarray((3.0 * B[:, 0] + B[:, 1]).reshape(n, 1))
        h.solve(y)
        assert rel_inf(y[:, 0], 3.0 * X[:, 0] + X[:, 1]) <= 1e-9


# the blocked programme with wide items forced on every chain of every kind of factor, run as items and as their twin blocks
@pytest.mark.parametrize("waves", ["1", "0"])
@pytest.mark.parametrize("kind,fac", KINDS)
def test_trisolve_items_and_twins_vs_oracle(kind, fac, waves, monkeypatch):
    monkeypatch.setenv("RSP_ITEM_MIN_SOURCES", "1")
    monkeypatch.setenv("RSP_ITEM_MIN_FILL", "0")
    monkeypatch.setenv("RSP_ITEM_WAVES", waves)
    T = factors()[fac]
    n = T[0]
    for nrhs in (3, 64, 200, 256):
        B = np.asfortranarray((2.0 * synth.splitmix64(11, n * nrhs) - 1.0).reshape(nrhs, n).T)
        want = np.stack([orc.trisolve(kind, T, B[:, r]) for r in (0, nrhs // 2, nrhs - 1)], axis=1)
        h = rs.TriHandle(T, KIND_ID[kind])
        X = B.copy(order="F")
        h.solve(X)
        assert rel_inf(X[:, [0, nrhs // 2, nrhs - 1]], want) <= REL_X
        assert (h.info()["launches"] > 0)


@pytest.mark.parametrize("nrhs", [64, 96, 200, 256])
def test_pair_solve_pipelined_halves(nrhs):
    """rsp_tri_solve_pair_dev_f64: lsolve + ltsolve with the two column halves on two streams == the oracle's loops."""
    torch = pytest.importorskip("torch")
    T = factors()["chol_L"]
    n = T[0]
    B = (2.0 * synth.splitmix64(23, n * nrhs) - 1.0).reshape(n, nrhs)
    want = np.stack([orc.trisolve("ltsolve", T, orc.trisolve("lsolve", T, B[:, r].copy())) for r in range(nrhs)], axis=1)
    lo, up = rs.TriHandle(T, 0), rs.TriHandle(T, 1)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):  # repeated calls reuse the auxiliary stream and events
        X = torch.tensor(B, device="cuda:0")
        rs.solve_pair_dev(lo, up, X.data_ptr(), nrhs, nrhs, st)
        torch.cuda.synchronize()
        assert rel_inf(X.cpu().numpy(), want) <= REL_X


def test_trisolve_device_row_major():
    torch = pytest.importorskip("torch")
    T = factors()["chol_L"]
    n, nrhs = T[0], 96
    B = (2.0 * synth.splitmix64(9, n * nrhs) - 1.0).reshape(n, nrhs)
    want = np.stack([orc.trisolve("lsolve", T, B[:, r].copy()) for r in range(nrhs)], axis=1)
    X = torch.tensor(B, device="cuda:0")
    h = rs.TriHandle(T, 0)
    h.solve_dev(X.data_ptr(), rs.api.ROW_MAJOR, nrhs, nrhs, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert rel_inf(X.cpu().numpy(), want) <= REL_X


def test_chol_handle_multi_rhs_and_refactor():
    A = synth.laplacian_2d(64)
    n, nrhs = A[0], 40
    S = rs.schol(A, 0)
    F = rs.CholHandle(A, S)
    B = np.asfortranarray((2.0 * synth.splitmix64(3, n * nrhs) - 1.0).reshape(nrhs, n).T)
    X = B.copy(order="F")
    F.solve(X)
    import scipy.sparse.linalg as spla
    from helpers import to_scipy
    M = to_scipy(A)
    R = M @ X - B
    assert np.max(np.linalg.norm(R, axis=0) / np.linalg.norm(B, axis=0)) <= REL_RES
    st, want = orc.cholsol(A, B[:, 3].copy(), 0)
    assert rel_inf(X[:, 3], want) <= REL_X
    info = F.info()
    assert info["n"] == n and info["lnz"] == S.lnz


def test_empty_and_tiny():
    A = rs.Sprs.from_vec([[4.0]])
    b = np.array([2.0])
    rs.cholsol(A, b, 0)
    assert b[0] == 0.5
    b = np.array([2.0])
    rs.lusol(A, b, 1, 1.0)
    assert b[0] == 0.5
    b = np.array([3.0])
    rs.lsolve(A, b)
    assert b[0] == 0.75


# ---- batched cholsol: many systems, one pattern (BASELINE.json's batched configuration) -----------------

def _circuit_batch(n, nsys, topology_seed=6):
    mats = [synth.circuit_spd(n, topology_seed, 100 + s) for s in range(nsys)]
    p, i = mats[0][2], mats[0][3]
    assert all(np.array_equal(m[2], p) and np.array_equal(m[3], i) for m in mats)  # shared pattern
    values = np.ascontiguousarray(np.stack([m[4] for m in mats]))
    B = np.ascontiguousarray(np.stack([synth.splitmix64(1000 + s, n) for s in range(nsys)]))
    return mats, values, B


@pytest.mark.parametrize("n,order", [(300, 0), (300, -1), (2000, 0)])
def test_cholsol_batch_vs_oracle(n, order):
    mats, values, B = _circuit_batch(n, 6)
    want = []
    for s, A in enumerate(mats):
        st, x = orc.cholsol(A, B[s].copy(), order)
        assert st == orc.OK
        want.append(x)
    got = B.copy()
    status = rs.cholsol_batch(mats[0], values, got, order)
    assert status.tolist() == [0] * len(mats)
    for s, A in enumerate(mats):
        assert rel_inf(got[s], want[s]) <= REL_X
        r = orc.gaxpy(A, got[s], -B[s])
        assert np.linalg.norm(r) / np.linalg.norm(B[s]) <= REL_RES


def test_cholsol_batch_matches_single_calls_and_flags_bad_systems():
    mats, values, B = _circuit_batch(400, 5)
    values[3, mats[0][2][17]:mats[0][2][18]] *= -1.0  # system 3: column 17 gets a negative diagonal
    got = B.copy()
    status = rs.cholsol_batch(mats[0], values, got, 0)
    assert status.tolist() == [0, 0, 0, 1, 0]
    assert np.array_equal(got[3], B[3])  # cholsol returns NotPositiveDefinite before it touches b (lib.rs:380)
    for s in (0, 1, 2, 4):
        one = B[s].copy()
        rs.cholsol(mats[s], one, 0)
        assert rel_inf(got[s], one) <= 1e-12


def test_cholsol_batch_empty_and_single():
    mats, values, B = _circuit_batch(64, 1)
    got = B.copy()
    assert rs.cholsol_batch(mats[0], values, got, 0).tolist() == [0]
    st, want = orc.cholsol(mats[0], B[0].copy(), 0)
    assert rel_inf(got[0], want) <= REL_X
    assert rs.cholsol_batch(mats[0], values[:0], got[:0], 0).tolist() == []


# ---- BASELINE.json's full size (2-D Laplacian 1024^2): properties that need no reference run ---------------------------

def test_full_size_properties():
    """n = 1 048 576, nnz(L) = 47.5 M: the oracle would take minutes here, so the checks are size-independent ones:
    the residual of A x = b, linearity of the solve, and columns of a wide block agreeing with one-column solves."""
    import scipy.sparse as sp
    A = synth.laplacian_2d(1024)
    n = A[0]
    Asp = sp.csc_matrix((A[4], A[3], A[2]), shape=(n, n))
    S = rs.schol(A, 0)
    F = rs.CholHandle(A, S)
    assert F.info()["lnz"] == 47496134  # the pattern is an integer result: the same on every run and every machine
    k = 192  # three 64-column chunks: the pair pipeline cuts it into unequal parts
    B = np.asfortranarray((2.0 * synth.splitmix64(7, n * k) - 1.0).reshape(k, n).T)
    X = B.copy(order="F")
    F.solve(X)
    for c in (0, 63, 64, 191):
        r = Asp @ X[:, c] - B[:, c]
        assert np.linalg.norm(r) / np.linalg.norm(B[:, c]) <= REL_RES
    # one-column solves of the same right-hand sides
    for c in (5, 130):
        x1 = B[:, c].copy()
        F.solve(x1)
        assert rel_inf(X[:, c], x1) <= 1e-11
    # linearity: solve(2 b1 - 3 b2) = 2 solve(b1) - 3 solve(b2)
    mix = 2.0 * B[:, 1] - 3.0 * B[:, 2]
    F.solve(mix)
    assert rel_inf(mix, 2.0 * X[:, 1] - 3.0 * X[:, 2]) <= 1e-10
    # the factor itself: L L' = P A P' on a probe vector, through the triangular products of scipy
    L = F.factor()
    Lsp = sp.csc_matrix((L.x, L.i.astype(np.int64), L.p), shape=(n, n))
    v = 2.0 * synth.splitmix64(9, n) - 1.0
    pinv = np.array(S.pinv)
    pv = np.empty(n)
    pv[pinv] = v  # (P v)[pinv[k]] = v[k]
    lhs = Lsp @ (Lsp.T @ pv)
    rhs = np.empty(n)
    rhs[pinv] = Asp @ v
    assert np.linalg.norm(lhs - rhs) / np.linalg.norm(rhs) <= 1e-13


def test_large_banded_lu_properties():
    """BASELINE.json's unsymmetric configuration at a quarter of its size (n = 50 000; the host LU is the slow part):
    residual of A x = b through lusol, and the handle's multi-column solve against its one-column solve."""
    import scipy.sparse as sp
    A = synth.banded_unsymmetric(50000)
    n = A[0]
    Asp = sp.csc_matrix((A[4], A[3], A[2]), shape=(n, n))
    b = synth.splitmix64(5, n)
    x = b.copy()
    rs.lusol(A, x, 1, 1.0)
    assert np.linalg.norm(Asp @ x - b) / np.linalg.norm(b) <= REL_RES
    S = rs.sqr(A, 1)
    N = rs.lu(A, S, 1.0)
    B = np.asfortranarray(synth.splitmix64(6, n * 70).reshape(70, n).T)
    X = B.copy(order="F")
    N.solve(X)
    for c in (0, 64, 69):
        assert np.linalg.norm(Asp @ X[:, c] - B[:, c]) / np.linalg.norm(B[:, c]) <= REL_RES
    x1 = B[:, 33].copy()
    N.solve(x1)
    assert rel_inf(X[:, 33], x1) <= 1e-11


def test_cholsol_batch_large_systems_take_the_ordinary_path():
    """Systems too large for the LDS-resident batched solve (n > 16384) are factored and solved one by one."""
    import scipy.sparse as sp
    mats, values, B = _circuit_batch(17000, 2)
    got = B.copy()
    assert rs.cholsol_batch(mats[0], values, got, 0).tolist() == [0, 0]
    for s, A in enumerate(mats):
        Asp = sp.csc_matrix((A[4], A[3], A[2]), shape=(A[0], A[0]))
        assert np.linalg.norm(Asp @ got[s] - B[s]) / np.linalg.norm(B[s]) <= REL_RES
        one = B[s].copy()
        rs.cholsol(A, one, 0)
        assert rel_inf(got[s], one) <= 1e-11


@pytest.mark.skipif(__import__("os").environ.get("RSP_CHOL_GROUPED") == "0", reason="the source-owned variant sums with atomics")
def test_chol_is_bit_reproducible():
    """The numeric factorisation owns every target tile with one wavefront and sums in a host-fixed order (no atomics):
    two factorisations of the same matrix give the same bits, as the sequential reference does."""
    for A in (synth.laplacian_2d(160), synth.laplacian_3d(20), synth.circuit_spd(3000)):
        S = rs.schol(A, 0)
        L1 = rs.CholHandle(A, S).factor()
        L2 = rs.CholHandle(A, S).factor()
        assert np.array_equal(L1.x, L2.x)
        st, Lo = orc.chol(A, orc.schol(A, 0))
        assert st == orc.OK and L1.i.astype(np.int64).tolist() == Lo[3].tolist()
        assert rel_inf(L1.x, Lo[4]) <= 1e-11
