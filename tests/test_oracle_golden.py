"""Pin the CPU oracle (oracle/rsparse_oracle.c) against every golden vector the reference holds for
the numeric solve path: tests/solver_tests.rs hot-path tests, the README
full-precision answer, and the exact-integer pins of tests/basic_tests.rs qr_1 / qr_2."""
import numpy as np
import pytest

from helpers import case_matrix, load_sprs, reference_cases
from oracle import pyoracle as orc

CASES = reference_cases()
SOLVER = CASES["solver"]


def _abs_close(a, b, tol):
    a, b = np.asarray(a), np.asarray(b)
    assert np.all(np.abs(a[: len(b)] - b[: len(a)]) <= tol), float(np.max(np.abs(a - b)))


@pytest.mark.parametrize("name", [k for k, v in SOLVER.items() if v["call"] in ("lsolve", "ltsolve", "usolve", "utsolve")])
def test_triangular_golden(name):
    c = SOLVER[name]
    got = orc.trisolve(c["call"], case_matrix(c), c["b"])
    _abs_close(got, c["x"], c["tol"])


@pytest.mark.parametrize("name", [k for k, v in SOLVER.items() if v["call"] in ("cholsol", "lusol")])
def test_drivers_golden(name):
    c = SOLVER[name]
    A = case_matrix(c)
    if c["call"] == "cholsol":
        st, x = orc.cholsol(A, c["b"], c["order"])
    else:
        st, x = orc.lusol(A, c["b"], c["order"], c["lutol"])
    if c["check"] == "error":
        assert st == orc.NOT_POSITIVE_DEFINITE
        return
    assert st == orc.OK
    if c["check"] == "x":
        _abs_close(x, c["x"], c["tol"])
    elif c["check"] == "x_bitwise":
        # README.md:182 prints the f64 result with round-trip precision: equality must be exact
        assert [float(v) for v in x] == c["x"]
    elif c["check"] == "residual":
        r = orc.gaxpy(A, x, np.zeros(len(x)))
        _abs_close(r, c["b"], c["tol"])


def test_allsol_1_cholsol_lusol_agree():
    c = SOLVER["allsol_1"]
    A = case_matrix(c)
    st1, x1 = orc.cholsol(A, c["b"], c["order_chol"])
    st2, x2 = orc.lusol(A, c["b"], c["order_lu"], c["lutol"])
    assert st1 == st2 == orc.OK
    _abs_close(x1, x2, c["tol"])


def test_qr_1_integer_pins():
    """tests/basic_tests.rs:1282-1303 -- the only bit-exact pins of amd/etree/post/counts in the reference."""
    pin = CASES["structure"]["qr_1"]
    out = orc.sqr_qr(load_sprs(pin["asset"]), pin["order"])
    assert out["q"].tolist() == pin["sym.q"]
    assert out["parent"].tolist() == pin["sym.parent"]
    assert out["cp"].tolist() == pin["sym.cp"]
    assert out["pinv"].tolist() == pin["sym.pinv"]
    assert (out["m2"], out["lnz"], out["unz"]) == (pin["sym.m2"], pin["sym.lnz"], pin["sym.unz"])
    assert out["Vi"].tolist() == pin["qr.l.i"] and out["Vp"].tolist() == pin["qr.l.p"]
    assert out["Ri"].tolist() == pin["qr.u.i"] and out["Rp"].tolist() == pin["qr.u.p"]
    _abs_close(out["Vx"], pin["qr.l.x~1e-6"], 1e-6)
    _abs_close(out["Rx"], pin["qr.u.x~1e-6"], 1e-6)
    _abs_close(out["beta"], pin["qr.b~1e-6"][:50], 1e-6)


def test_qr_2_integer_pins():
    pin = CASES["structure"]["qr_2"]
    out = orc.sqr_qr(load_sprs(pin["asset"]), pin["order"])
    assert out["Vi"].tolist() == pin["qr.l.i"] and out["Vp"].tolist() == pin["qr.l.p"]
    assert out["Ri"].tolist() == pin["qr.u.i"] and out["Rp"].tolist() == pin["qr.u.p"]
    assert (out["lnz"], out["unz"]) == (pin["qr.l.nzmax"], pin["qr.u.nzmax"])
    _abs_close(out["Vx"], pin["qr.l.x~1e-6"][:23], 1e-6)
    _abs_close(out["Rx"], pin["qr.u.x~1e-6"], 1e-6)
    _abs_close(out["beta"], pin["qr.b~1e-6"][:10], 1e-6)


# ---- qrsol: the consumer of usolve / utsolve; tests/solver_tests.rs:687-968 ----
from helpers import qr_case, to_dense  # noqa: E402

QR_CASES = [k for k, v in SOLVER.items() if v["call"].startswith("qrsol")]


def check_qrsol_case(c, A, got, solve_other):
    """The reference test's own assertions on the vector qrsol left in b."""
    m, n = A[0], A[1]
    b = np.asarray(c["b"], np.float64)
    if "x" in c:
        _abs_close(got, c["x"], c["tol"])
    if "residual" in c["check"]:
        assert np.abs(to_dense(A) @ got[:n] - b[:m]).max() <= c["res_tol"]
    if "agree" in c["check"]:
        _abs_close(got, solve_other(), c["tol"])


@pytest.mark.parametrize("name", QR_CASES)
def test_qrsol_golden(name):
    c = SOLVER[name]
    A = qr_case(c)
    got = orc.qrsol(A, c["b"], c["order"])
    def other():
        if "cholsol" in c["call"]:
            return orc.cholsol(A, c["b"], c["order_chol"])[1]
        return orc.lusol(A, c["b"], c["order_lu"], c["lutol"])[1]
    check_qrsol_case(c, A, got, other)
