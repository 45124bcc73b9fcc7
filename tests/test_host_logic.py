"""CPU-only checks of the product's host layer (no GPU compute calls):
   * the C-ABI library loads and exports every symbol include/rsparse_hip.h declares;
   * integer structure computed by the product's own host code (amd, schol, lu pivoting and patterns,
     level sets) is BIT-EXACT against the oracle and against the reference's integer pin (qr_1's q)."""
import os
import re

import numpy as np
import pytest

import rsparse_amd as rs
from helpers import case_matrix, load_sprs, reference_cases
from oracle import pyoracle as orc
from rsparse_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = reference_cases()


def _matrices():
    out = {}
    for name in ("cholsol_1.sprs", "cholsol_2.sprs", "cholsol_5.sprs", "cholsol_7.sprs", "lusol_3.sprs", "lusol_4.sprs",
                 "lusol_6.sprs"):
        out[name] = load_sprs(name)
    out["lap2d_24x17"] = synth.laplacian_2d(24, 17)
    out["lap2d_64"] = synth.laplacian_2d(64)
    out["lap3d_9"] = synth.laplacian_3d(9)
    out["band_600"] = synth.banded_unsymmetric(600, band=20)
    out["circuit_500"] = synth.circuit_spd(500)
    out["rand_300"] = synth.random_sparse(300, 0.02, seed=11)
    out["randspd_200"] = synth.random_sparse(200, 0.02, seed=12, symmetric_pd=True)
    return out


MATS = _matrices()


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "rsparse_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(rsp_[A-Za-z0-9_]+)\s*\(", header)))
    L = rs.lib()
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(L._declared) == declared  # the Python mirror binds exactly the header's surface


@pytest.mark.parametrize("name", sorted(MATS))
@pytest.mark.parametrize("order", [-1, 0, 1, 2])
def test_amd_bit_exact(name, order):
    A = MATS[name]
    want = orc.amd(A, order)
    got = rs.amd(A, order)
    if want is None:
        assert got is None
    else:
        assert got.tolist() == want.tolist()


def test_amd_matches_reference_pin_qr_1():
    pin = CASES["structure"]["qr_1"]  # tests/basic_tests.rs:1282
    assert rs.amd(load_sprs(pin["asset"]), 2).tolist() == pin["sym.q"]


@pytest.mark.parametrize("name", [k for k in sorted(MATS) if MATS[k][0] == MATS[k][1]])
@pytest.mark.parametrize("order", [-1, 0])
def test_schol_bit_exact(name, order):
    A = MATS[name]
    want = orc.schol(A, order)
    got = rs.schol(A, order)
    assert (got.pinv is None) == (want["pinv"] is None)
    if want["pinv"] is not None:
        assert got.pinv.tolist() == want["pinv"].tolist()
    assert got.parent.tolist() == want["parent"].tolist()
    assert got.cp.tolist() == want["cp"].tolist()
    assert got.lnz == got.unz == want["lnz"]


@pytest.mark.parametrize("name", ["lusol_3.sprs", "lusol_4.sprs", "lusol_6.sprs", "cholsol_7.sprs", "band_600", "rand_300",
                                  "circuit_500", "lap2d_24x17"])
@pytest.mark.parametrize("order,tol", [(1, 1e-12), (1, 1.0), (-1, 1e-6), (0, 0.5)])
def test_lu_bit_exact(name, order, tol):
    A = MATS[name]
    st, want = orc.lu(A, order, tol)
    assert st == orc.OK
    S = rs.sqr(A, order)
    N = rs.lu(A, S, tol)
    assert N.pinv.tolist() == want["pinv"].tolist()              # pivot sequence
    for got, ref in ((N.l, want["L"]), (N.u, want["U"])):
        assert got.p.tolist() == ref[2].tolist()
        assert got.i.astype(np.int64).tolist() == ref[3].tolist()  # pattern, in storage order
        assert np.array_equal(got.x, ref[4])                        # values, bitwise
    q = S.q[: A[1]] if S.q is not None else np.arange(A[1])
    assert q.tolist() == want["q"].tolist()


@pytest.mark.parametrize("name", ["band_600", "circuit_500", "lap2d_24x17", "lap2d_64", "lap3d_9", "randspd_200"])
@pytest.mark.parametrize("order", [1, -1, 0])
def test_lu_static_pattern_is_the_pattern_of_the_reference_when_diagonals_pivot(name, order):
    """Integer half of the GPU numeric LU (host/lu_static.cpp): on matrices whose every column keeps its diagonal as pivot,
    pinv and the patterns of L and U in storage order equal the oracle's lu (lib.rs:519-622)."""
    A = MATS[name]
    st, want = orc.lu(A, order, 1.0)
    assert st == orc.OK
    n = A[1]
    assert np.array_equal(want["pinv"][want["q"]], np.arange(n)), "test matrix must keep its diagonal pivots"
    N, depth = rs.lu_static_pattern(A, rs.sqr(A, order))
    assert 1 <= depth <= n
    assert N.pinv.tolist() == want["pinv"].tolist()
    for got, ref in ((N.l, want["L"]), (N.u, want["U"])):
        assert got.p.tolist() == ref[2].tolist()
        assert got.i.astype(np.int64).tolist() == ref[3].tolist()
        assert not got.x.any()  # values are the device's business


def test_lu_static_pattern_refuses_what_the_device_cannot_do():
    # no structural diagonal in column 0: there is no static pivot sequence
    A = (2, 2, np.array([0, 1, 3], np.int64), np.array([1, 0, 1], np.int64), np.array([1.0, 2.0, 3.0]))
    with pytest.raises(rs.DeviceError):
        rs.lu_static_pattern(A, rs.sqr(A, -1))
    # a duplicate entry inside a column (the reference keeps the last value; the device scatters in parallel)
    D = (2, 2, np.array([0, 3, 4], np.int64), np.array([0, 1, 0, 1], np.int64), np.array([1.0, 2.0, 3.0, 4.0]))
    with pytest.raises(rs.DeviceError):
        rs.lu_static_pattern(D, rs.sqr(D, -1))


@pytest.mark.parametrize("threads", ["1", "3", "8"])
def test_lu_static_pattern_does_not_depend_on_the_thread_count(threads, monkeypatch):
    """The search runs on the host threads by elimination-tree levels from 4 096 columns on (RSP_LU_SEARCH_THREADS): pinv, the
    patterns in the reference's storage order and the dependency depth equal the oracle's lu whatever the thread count."""
    monkeypatch.setenv("RSP_LU_SEARCH_THREADS", threads)
    for A, order in ((synth.banded_unsymmetric(12000), 1), (synth.laplacian_2d(80), 1), (synth.banded_unsymmetric(9000), -1)):
        st, want = orc.lu(A, order, 1.0)
        assert st == orc.OK
        assert np.array_equal(want["pinv"][want["q"]], np.arange(A[1])), "test matrix must keep its diagonal pivots"
        N, depth = rs.lu_static_pattern(A, rs.sqr(A, order))
        assert N.pinv.tolist() == want["pinv"].tolist()
        for got, ref in ((N.l, want["L"]), (N.u, want["U"])):
            assert np.array_equal(got.p, ref[2])
            assert np.array_equal(got.i.astype(np.int64), ref[3])
        if threads == "1":
            test_lu_static_pattern_does_not_depend_on_the_thread_count.depth = getattr(
                test_lu_static_pattern_does_not_depend_on_the_thread_count, "depth", {})
            test_lu_static_pattern_does_not_depend_on_the_thread_count.depth[(A[1], order)] = depth
        else:
            assert depth == test_lu_static_pattern_does_not_depend_on_the_thread_count.depth[(A[1], order)]
    # a duplicate entry inside a column far from the first one: every thread stops, the call refuses
    m, n, p, i, x = synth.banded_unsymmetric(9000)
    j = 7000
    i2 = np.insert(i, p[j + 1], i[p[j]])
    x2 = np.insert(x, p[j + 1], 0.5)
    p2 = p.copy()
    p2[j + 1:] += 1
    B = (m, n, p2, i2, x2)
    with pytest.raises(rs.DeviceError):
        rs.lu_static_pattern(B, rs.sqr(B, -1))


def test_lu_no_pivot_error():
    A = rs.Sprs.from_vec([[1.0, 2.0], [2.0, 4.0]])  # singular
    with pytest.raises(rs.NoPivot):
        rs.lu(A, rs.sqr(A, -1), 1.0)
    st, _ = orc.lu(A.as_tuple(), -1, 1.0)
    assert st == orc.NO_PIVOT


def _levels_python(kind, T):
    m, n, p, i, x = T
    lvl = np.zeros(n, np.int64)
    if kind in (rs.api.LSOLVE, rs.api.USOLVE):  # scatter loops of the reference: column j feeds rows i[p]
        order = range(n) if kind == rs.api.LSOLVE else range(n - 1, -1, -1)
        for j in order:
            rows = i[p[j] + 1: p[j + 1]] if kind == rs.api.LSOLVE else i[p[j]: p[j + 1] - 1]
            for r in rows:
                lvl[r] = max(lvl[r], lvl[j] + 1)
    else:  # gather loops: unknown j needs every x[i[p]]
        order = range(n - 1, -1, -1) if kind == rs.api.LTSOLVE else range(n)
        for j in order:
            rows = i[p[j] + 1: p[j + 1]] if kind == rs.api.LTSOLVE else i[p[j]: p[j + 1] - 1]
            for r in rows:
                lvl[j] = max(lvl[j], lvl[r] + 1)
    return lvl


@pytest.mark.parametrize("name", ["cholsol_2.sprs", "cholsol_5.sprs", "lap2d_24x17", "circuit_500"])
def test_level_sets_bit_exact_cholesky_factor(name):
    A = MATS[name]
    S = orc.schol(A, 0)
    st, L = orc.chol(A, S)
    assert st == orc.OK
    for kind in (rs.api.LSOLVE, rs.api.LTSOLVE):
        h = rs.TriHandle(L, kind)
        assert h.levels().tolist() == _levels_python(kind, L).tolist()


@pytest.mark.parametrize("name", ["lusol_4.sprs", "band_600"])
def test_level_sets_bit_exact_lu_factors(name):
    st, f = orc.lu(MATS[name], 1, 1.0)
    assert st == orc.OK
    for kind, T in ((rs.api.LSOLVE, f["L"]), (rs.api.USOLVE, f["U"]), (rs.api.UTSOLVE, f["U"]), (rs.api.LTSOLVE, f["L"])):
        assert rs.TriHandle(T, kind).levels().tolist() == _levels_python(kind, T).tolist()


def test_bad_input_is_reported_not_crashed():
    # a "lower" matrix whose first entry in a column is not followed by strictly lower rows
    T = rs.Sprs(2, 2, [0, 2, 3], [1, 0, 1], [1.0, 2.0, 3.0])
    with pytest.raises(rs.DeviceError):
        rs.TriHandle(T, rs.api.LSOLVE)
    # empty column: the reference would index out of bounds (panic)
    T = rs.Sprs(2, 2, [0, 1, 1], [0], [1.0])
    with pytest.raises(rs.DeviceError):
        rs.TriHandle(T, rs.api.LSOLVE)


def test_from_vec_matches_reference_constructor():
    c = CASES["solver"]["lsolve_1"]
    A = rs.Sprs.from_vec(c["dense"])
    m, n, p, i, x = case_matrix(c)
    assert A.p.tolist() == p.tolist() and A.i.tolist() == i.tolist() and np.array_equal(A.x, x)


# ---- multi-right-hand-side programme (clusters + blocks): host-side validator, no GPU -----------------------------

def _program_factors():
    A = synth.laplacian_2d(50, 37)
    st, L = orc.chol(A, orc.schol(A, 0))
    B = synth.banded_unsymmetric(3000, band=30)
    st2, f = orc.lu(B, 1, 1.0)
    C = load_sprs("lusol_6.sprs")
    st3, g = orc.lu(C, 1, 1e-12)
    assert st == st2 == st3 == orc.OK
    return [(0, L), (1, L), (0, f["L"]), (1, f["L"]), (2, f["U"]), (3, f["U"]), (0, g["L"]), (2, g["U"]), (3, g["U"])]


# cluster_slots > 0: group-record clusters of that many LDS slots; 0: the library default
@pytest.mark.parametrize("task_blocks,cluster_slots", [(0, 0), (1, 16), (4, 96), (2, 1)])
def test_block_programme_is_valid_and_exact(task_blocks, cluster_slots):
    """Every id in range, every row read only after the level that writes it, every unknown written exactly once,
    and the replayed programme agrees with sequential substitution (rsp_tri_validate_program)."""
    for kind, T in _program_factors():
        rep = rs.validate_program(T, kind, task_blocks, cluster_slots)
        assert rep["bad_index"] == 0 and rep["order_violations"] == 0 and rep["unwritten"] == 0, (kind, rep)
        assert rep["max_rel_error"] <= 1e-12, (kind, rep)
        if cluster_slots == 1:
            assert rep["clusters"] == 0


@pytest.mark.parametrize("env", [{"RSP_ITEM_MIN_SOURCES": "1", "RSP_ITEM_MIN_FILL": "0"},
                                 {"RSP_ITEM_MIN_SOURCES": "1", "RSP_ITEM_MIN_FILL": "0", "RSP_ITEM_ROWS": "32", "RSP_SORT_ROWS": "0"},
                                 {"RSP_ITEM_MIN_SOURCES": "1", "RSP_ITEM_MIN_FILL": "0", "RSP_MAX_BLOCK_SOURCES": "48", "RSP_BIG_CHAIN": "8"},
                                 {"RSP_ITEM_ROWS": "16"}])
def test_block_programme_with_wide_items_is_valid_in_both_views(env, monkeypatch):
    """Round 3: wide items over every chain (however short and sparse, over partial sums too), the programme replayed once
    with the items as their twin blocks and once as items (masks, compressed coefficient blocks): both views valid and
    equal to sequential substitution; rows of a chain regrouped by similarity or in natural order."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for kind, T in _program_factors():
        rep = rs.validate_program(T, kind, 0, 0)
        assert rep["bad_index"] == 0 and rep["order_violations"] == 0 and rep["unwritten"] == 0, (kind, rep)
        assert rep["max_rel_error"] <= 1e-12, (kind, rep)


def test_block_programme_with_duplicate_entries():
    """The reference does not sum duplicate entries (src/data.rs:345-367): its loops apply each one.  The programme must
    add them up wherever it builds coefficients (blocks, in-block triangles, inverse diagonal blocks)."""
    A = synth.laplacian_2d(30, 25)
    st, L = orc.chol(A, orc.schol(A, 0))
    n, _, p, i, x = L
    pi, ii, xi = [0], [], []
    for j in range(n):  # every off-diagonal entry twice, with different halves
        rows, vals = i[p[j]:p[j + 1]], x[p[j]:p[j + 1]]
        ii += [rows[0]] + [r for r in rows[1:] for _ in (0, 1)]
        xi += [vals[0]] + [v * f for v in vals[1:] for f in (0.25, 0.75)]
        pi.append(len(ii))
    D = (n, n, np.array(pi, np.int64), np.array(ii, np.int64), np.array(xi))
    for kind in (0, 1):
        rep = rs.validate_program(D, kind)
        assert rep["bad_index"] == 0 and rep["order_violations"] == 0 and rep["unwritten"] == 0, rep
        assert rep["max_rel_error"] <= 1e-12, rep


def test_batched_cholsol_rejects_bad_input_and_needs_a_gpu():
    import torch
    from rsparse_amd import synth
    A = rs.Sprs.from_tuple(synth.circuit_spd(30, 6, 100))
    vals = np.ascontiguousarray(A.x[None, :].copy())
    B = np.ones((1, A.n))
    # an empty batch is a no-op everywhere
    assert rs.cholsol_batch(A, vals[:0], B[:0], 0).tolist() == []
    # a non-square pattern is refused before anything touches a device
    R = rs.Sprs(3, 2, [0, 1, 2], [0, 1], [1.0, 1.0])
    with pytest.raises(rs.DeviceError):
        rs.cholsol_batch(R, np.ones((1, 2)), np.ones((1, 2)), 0)
    if not torch.cuda.is_available():  # no CPU fallback: the numeric path fails loudly without a GPU
        with pytest.raises(rs.DeviceError):
            rs.cholsol_batch(A, vals, B, 0)


@pytest.mark.parametrize("count,threads", [(0, 4), (1, 4), (1000, 8), (200_000, 1), (200_000, 3), (1_000_000, 8), (1_000_003, 16)])
def test_threaded_host_helpers_agree_with_sequential(count, threads):
    """host/parallel.hpp (threaded sort and ranges used by the analyses): same result as the sequential code for any
    thread count, including chunk counts that do not divide the input."""
    assert rs.lib().rsp_selftest_host_threads(count, threads, 12345 + count) == 0


def test_symb_from_parts_rejects_corrupt_vectors():
    """rsp_symb_from_parts: vectors that are later used as indices (cp by chol, q by lu) are validated up front -- where the
    reference would panic on a bounds check the library returns RSP_BAD_INPUT, never heap corruption."""
    A = synth.laplacian_2d(6, 5)
    n = A[0]
    S = rs.schol(A, 0)
    pinv, parent, cp = np.array(S.pinv), np.array(S.parent), np.array(S.cp)
    assert rs.Symb.from_parts(n, pinv=pinv, parent=parent, cp=cp).lnz == S.lnz  # the honest vectors pass
    bad = []
    c = cp.copy(); c[0] = 1; bad.append(dict(pinv=pinv, parent=parent, cp=c))                 # does not start at 0
    c = cp.copy(); c[3] = c[2]; bad.append(dict(pinv=pinv, parent=parent, cp=c))              # an empty column (no diagonal)
    c = cp.copy(); c[5] = c[4] - 2; bad.append(dict(pinv=pinv, parent=parent, cp=c))          # not monotone
    c = cp.copy(); c[n] = c[n - 1] + 7; bad.append(dict(pinv=pinv, parent=parent, cp=c))      # last column longer than n - k
    c = cp.copy(); c[1:] += 10 ** 15; bad.append(dict(pinv=pinv, parent=parent, cp=c))        # absurd sizes
    p = pinv.copy(); p[2] = p[3]; bad.append(dict(pinv=p, parent=parent, cp=cp))              # not a permutation
    p = pinv.copy(); p[0] = n; bad.append(dict(pinv=p, parent=parent, cp=cp))                 # out of range
    q = np.arange(n); q[4] = -1; bad.append(dict(q=q))
    q = np.arange(n); q[1] = q[2]; bad.append(dict(q=q))
    t = parent.copy(); t[0] = n; bad.append(dict(pinv=pinv, parent=t, cp=cp))
    for kw in bad:
        with pytest.raises(rs.DeviceError, match="status -1"):
            rs.Symb.from_parts(n, **kw)


def test_selftest_leaves_the_environment_alone(monkeypatch):
    """rsp_selftest_host_threads overrides the thread count for the calling thread only (no setenv in a library call)."""
    monkeypatch.setenv("RSP_HOST_THREADS", "3")
    for threads in (1, 2, 5, 0):
        assert rs.lib().rsp_selftest_host_threads(300000, threads, 7) == 0
        assert os.environ["RSP_HOST_THREADS"] == "3"


# ---- interchange formats (Sprs::save / Sprs::load, data.rs:414-516, and the binary format beside it) ----
ASSETS = ("cholsol_1.sprs", "cholsol_2.sprs", "cholsol_5.sprs", "cholsol_7.sprs", "lusol_3.sprs", "lusol_4.sprs", "lusol_6.sprs", "qrsol_8.sprs")


@pytest.mark.parametrize("name", ASSETS)
def test_sprs_text_format_reproduces_reference_files(name, tmp_path):
    """The .sprs assets were written by the reference's Sprs::save: load must read them and save must write them back
    byte for byte (integer lists and shortest round-trip floats in Rust's Debug form)."""
    import rsparse_amd as ra
    src = os.path.join(os.path.dirname(__file__), "golden", "assets", name)
    A = ra.Sprs.load(src)
    m, n, p, i, x = load_sprs(name)
    assert (A.m, A.n) == (m, n) and np.array_equal(A.p, p)
    nz = int(p[n])
    assert np.array_equal(A.i[:nz].astype(np.int64), i) and np.array_equal(A.x[:nz], x)
    out = tmp_path / name
    A.save(out)
    assert open(out).read() == open(src).read()


def test_sprs_text_format_edge_cases(tmp_path):
    import rsparse_amd as ra
    # values whose Python and Rust spellings differ: exponents, negative zero, subnormals
    x = [1e-7, 1e16, -0.0, 5e-324, 1.5, 123456789.125, 1e21, 1.2345e-10]
    A = ra.Sprs(8, 1, [0, 8], list(range(8)), x)
    A.save(tmp_path / "a.sprs")
    text = open(tmp_path / "a.sprs").read()
    assert "x: [1e-7, 1e16, -0.0, 5e-324, 1.5, 123456789.125, 1e21, 1.2345e-10]" in text
    B = ra.Sprs.load(tmp_path / "a.sprs")
    assert np.array_equal(B.x.view(np.uint64), A.x.view(np.uint64)) and np.array_equal(B.p, A.p) and np.array_equal(B.i, A.i)
    # the reference returns the empty matrix when nzmax, m or n is 0 (data.rs:445-447)
    ra.Sprs(3, 3, [0, 0, 0, 0], [], []).save(tmp_path / "z.sprs")
    Z = ra.Sprs.load(tmp_path / "z.sprs")
    assert (Z.m, Z.n, Z.nzmax) == (0, 0, 0)
    with open(tmp_path / "bad.sprs", "w") as f:
        f.write("nzmax: 2\nm: 2\nn: 2\np: [0, 1, 2]\ni: [0, 1]\n")  # no x line
    with pytest.raises(ValueError):
        ra.Sprs.load(tmp_path / "bad.sprs")


def test_sprs_binary_format(tmp_path):
    import rsparse_amd as ra
    for name in ("cholsol_5.sprs", "lusol_6.sprs"):
        A = ra.Sprs.from_tuple(load_sprs(name))
        A.save_bin(tmp_path / "a.bin")
        B = ra.Sprs.load_bin(tmp_path / "a.bin")
        assert (B.m, B.n) == (A.m, A.n)
        assert np.array_equal(B.p, A.p) and np.array_equal(B.i, A.i) and np.array_equal(B.x.view(np.uint64), A.x.view(np.uint64))
    # the layout the header documents: magic, m, n, nnz, then the three arrays
    raw = open(tmp_path / "a.bin", "rb").read()
    assert raw[:8] == b"RSPSPRS1" and np.frombuffer(raw[8:32], np.int64).tolist() == [A.m, A.n, len(A.x)]
    assert len(raw) == 32 + 8 * (A.n + 1) + 16 * len(A.x)
    # empty matrix round trip
    ra.Sprs(0, 0, [0], [], []).save_bin(tmp_path / "e.bin")
    E = ra.Sprs.load_bin(tmp_path / "e.bin")
    assert (E.m, E.n, len(E.x)) == (0, 0, 0)
    # rejected: missing file, wrong magic, truncated, row index out of range
    with pytest.raises(ValueError):
        ra.Sprs.load_bin(tmp_path / "nope.bin")
    open(tmp_path / "m.bin", "wb").write(b"XXXXXXXX" + raw[8:])
    with pytest.raises(ValueError):
        ra.Sprs.load_bin(tmp_path / "m.bin")
    open(tmp_path / "t.bin", "wb").write(raw[:-8])
    with pytest.raises(ValueError):
        ra.Sprs.load_bin(tmp_path / "t.bin")
    bad = bytearray(raw)
    off = 32 + 8 * (A.n + 1)
    bad[off:off + 8] = np.uint64(A.m).tobytes()
    open(tmp_path / "r.bin", "wb").write(bytes(bad))
    with pytest.raises(ValueError):
        ra.Sprs.load_bin(tmp_path / "r.bin")


# ---- host QR (sqr(qr=true), qr): integers against the reference's pins and the oracle, values against the oracle ----
def test_qr_symbolic_and_numeric_match_reference_pins():
    import rsparse_amd as ra
    pins = reference_cases()["structure"]
    pin = pins["qr_1"]
    A = load_sprs(pin["asset"])
    S = ra.sqr(A, pin["order"], True)
    assert S.q.tolist() == pin["sym.q"] and S.parent.tolist() == pin["sym.parent"] and S.cp.tolist() == pin["sym.cp"]
    assert S.pinv.tolist() == pin["sym.pinv"] and (S.m2, S.lnz, S.unz) == (pin["sym.m2"], pin["sym.lnz"], pin["sym.unz"])
    for name, nb in (("qr_1", 50), ("qr_2", 10)):
        pin = pins[name]
        A = load_sprs(pin["asset"])
        N = ra.qr(A, ra.sqr(A, pin["order"], True))
        V, R = N.l, N.u
        assert V.i.tolist() == pin["qr.l.i"] and V.p.tolist() == pin["qr.l.p"] and R.i.tolist() == pin["qr.u.i"] and R.p.tolist() == pin["qr.u.p"]
        assert (V.m, V.n, R.m, R.n) == (pin["qr.l.m"], pin["qr.l.n"], pin["qr.u.m"], pin["qr.u.n"])
        nv = len(V.x)
        assert np.abs(V.x - np.array(pin["qr.l.x~1e-6"][:nv])).max() <= 1e-6 and np.abs(R.x - np.array(pin["qr.u.x~1e-6"])).max() <= 1e-6
        assert np.abs(N.b - np.array(pin["qr.b~1e-6"][:nb])).max() <= 1e-6


@pytest.mark.parametrize("name,order", [("cholsol_5.sprs", 2), ("lusol_4.sprs", 2), ("qrsol_8.sprs", 2), ("cholsol_2.sprs", 0), ("lusol_3.sprs", -1)])
def test_qr_bitwise_vs_oracle(name, order):
    """Same integers AND the same values bit for bit as the oracle's restatement of sqr + qr (lib.rs:788-877, 1114-1133):
    one order of operations, no contraction on either side.  qrsol_8 is 200 x 201: factored as its transpose."""
    import rsparse_amd as ra
    from oracle import pyoracle as orc
    from helpers import transpose_csc
    A = load_sprs(name)
    if A[0] < A[1]:
        A = transpose_csc(A)
    S = ra.sqr(A, order, True)
    N = ra.qr(A, S)
    o = orc.sqr_qr(A, order, numeric=True)
    assert (S.q is None) == (o["q"] is None) and (S.q is None or np.array_equal(S.q, o["q"]))
    assert np.array_equal(S.parent, o["parent"]) and np.array_equal(S.cp, o["cp"]) and np.array_equal(S.pinv, o["pinv"])
    assert (S.m2, S.lnz, S.unz) == (o["m2"], o["lnz"], o["unz"])
    V, R = N.l, N.u
    nv, nr = int(o["Vp"][-1]), int(o["Rp"][-1])
    assert np.array_equal(V.p, o["Vp"]) and np.array_equal(V.i.astype(np.int64), o["Vi"][:nv]) and np.array_equal(R.p, o["Rp"])
    assert np.array_equal(R.i.astype(np.int64), o["Ri"][:nr])
    assert np.array_equal(V.x.view(np.uint64), o["Vx"][:nv].view(np.uint64)) and np.array_equal(R.x.view(np.uint64), o["Rx"][:nr].view(np.uint64))
    assert np.array_equal(N.b.view(np.uint64), o["beta"].view(np.uint64))


def test_qr_rejects_foreign_symbolic():
    import rsparse_amd as ra
    A = load_sprs("cholsol_5.sprs")
    with pytest.raises(ra.DeviceError):
        ra.qr(A, ra.schol(A, 0))            # a Cholesky analysis: no QR row permutation
    with pytest.raises(ra.DeviceError):
        ra.qr(A, ra.sqr(load_sprs("cholsol_2.sprs"), 2, True))  # another matrix


# ---- refill of a programme with new values on the same pattern (refactorisation) ----
@pytest.mark.parametrize("case", ["lap2d", "lap3d", "lu_L", "lu_U", "band"])
def test_programme_refill_is_bitwise_a_fresh_emission(case):
    """New values poured into a programme through its recipe == the programme emitted from scratch for those values, bit
    for bit (coefficients, 16 x 16 triangle inverses, chain inverses, cluster records), for every solve kind that applies;
    several builder settings (small clusters, several blocks per task)."""
    import rsparse_amd as ra
    from rsparse_amd import synth
    from oracle import pyoracle as orc
    if case in ("lap2d", "lap3d"):
        A = synth.laplacian_2d(60, 47) if case == "lap2d" else synth.laplacian_3d(12)
        st, T = orc.chol(A, orc.schol(A, 0))
        kinds = (0, 1)
    elif case in ("lu_L", "lu_U"):
        st, f = orc.lu(synth.banded_unsymmetric(1800, band=30), 1, 1.0)
        T, kinds = (f["L"], (0, 1)) if case == "lu_L" else (f["U"], (2, 3))
    else:  # a dense band: long chains (inverses in pieces) and source lists cut into partial sums
        n, p, i, x = 1400, [0], [], []
        for j in range(n):
            rows = list(range(j, min(n, j + 1101)))
            i += rows
            x += [4.0 + j % 3] + [0.01 * ((r * 7 + j) % 11 - 5) for r in rows[1:]]
            p.append(len(i))
        T, st, kinds = (n, n, np.array(p), np.array(i), np.array(x)), 0, (0, 1)
    assert st == 0
    nz = int(T[2][-1])
    new_values = np.asarray(T[4][:nz]) * (1.0 + 0.3 * synth.splitmix64(9, nz))
    for kind in kinds:
        for opts in ({}, {"cluster_slots": 8}, {"task_blocks": 3}):
            r = ra.validate_refill(T, new_values, kind, **opts)
            assert r["has_recipe"] == 1 and r["coef_mismatches"] == 0 and r["group_mismatches"] == 0 and r["coef_entries"] > 0, (kind, opts, r)


def test_programme_without_recipe_is_reported():
    """Duplicate entries add up inside one coefficient: such a programme carries no recipe (and is emitted again)."""
    import rsparse_amd as ra
    p, i, x = [0], [], []
    for j in range(60):
        i.append(j)
        x.append(2.0)
        if j + 1 < 60:
            i += [j + 1, j + 1]
            x += [0.1, 0.2]
        p.append(len(i))
    r = ra.validate_refill((60, 60, np.array(p), np.array(i), np.array(x)), np.array(x) * 1.5, 0)
    assert r["has_recipe"] == 0


@pytest.mark.parametrize("case", ["lap2d", "lap3d", "lu_U"])
def test_programme_does_not_depend_on_emitter_threads(case, monkeypatch):
    """The groups of a long chain are ranked on the host threads and appended in order: the programme (pattern,
    coefficients, clusters, recipe) is bit for bit the one the sequential emitter writes, for any number of threads and
    with small part / split thresholds (partial sums everywhere)."""
    import rsparse_amd as ra
    from rsparse_amd import synth
    from oracle import pyoracle as orc
    if case == "lap2d":
        A = synth.laplacian_2d(140, 121)
        st, T = orc.chol(A, orc.schol(A, 0))
        kinds = (0, 1)
    elif case == "lap3d":
        A = synth.laplacian_3d(13)
        st, T = orc.chol(A, orc.schol(A, 0))
        kinds = (0, 1)
    else:
        st, f = orc.lu(synth.banded_unsymmetric(2000, band=36), 1, 1.0)
        T, kinds = f["U"], (2, 3)
    assert st == 0
    for env in ({}, {"RSP_MAX_BLOCK_SOURCES": "96", "RSP_SPLIT_SOURCES": "32"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for kind in kinds:
            sums = []
            for threads in ("0", "1", "5", "16"):
                monkeypatch.setenv("RSP_EMIT_THREADS", threads)
                sums.append(ra.program_checksum(T, kind))
            assert all(s == sums[0] for s in sums), (case, kind, env)


def test_gather_form_does_not_depend_on_host_threads(monkeypatch):
    """The factor's columns are cut into pieces of the visiting order, one per host thread; every unknown still receives
    its terms in the reference's order: programmes built from the gather form are bit-identical for 1, 3 and 8 threads
    (a factor large enough for the threaded path: more than 2^19 entries per thread piece), all four solve kinds."""
    import rsparse_amd as ra
    from rsparse_amd import synth
    from oracle import pyoracle as orc
    A = synth.laplacian_2d(220, 220)
    st, L = orc.chol(A, orc.schol(A, 0))
    st2, f = orc.lu(synth.banded_unsymmetric(30000, band=40), 1, 1.0)
    assert st == 0 and st2 == 0 and int(L[2][-1]) > 2 ** 20 and int(f["U"][2][-1]) > 2 ** 20
    for T, kinds in ((L, (0, 1)), (f["U"], (2, 3))):
        for kind in kinds:
            sums = []
            for threads in ("1", "3", "8"):
                monkeypatch.setenv("RSP_HOST_THREADS", threads)
                sums.append(ra.program_checksum(T, kind))
            assert sums[0] == sums[1] == sums[2], kind
