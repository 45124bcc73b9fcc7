"""ctypes front-end of the CPU oracle (oracle/rsparse_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the rsparse_amd package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")

OK, NOT_POSITIVE_DEFINITE, NO_PIVOT = 0, 1, 2


def build(native=False):
    """Compile the oracle with gcc (seconds).  native=True adds -O3 -march=native (cpu_baseline timing)."""
    out = os.path.join(_HERE, "librsparse_oracle_native.so" if native else "librsparse_oracle.so")
    src = os.path.join(_HERE, "rsparse_oracle.c")
    if os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
        return out
    flags = ["-O3", "-march=native"] if native else ["-O2"]
    subprocess.check_call(["gcc", *flags, "-std=c99", "-fPIC", "-ffp-contract=off", "-shared", "-o", out, src, "-lm"])
    return out


def lib(native=False):
    global _LIB
    if _LIB is None or native:
        path = os.path.join(_HERE, "librsparse_oracle_native.so" if native else "librsparse_oracle.so")
        if not os.path.exists(path):
            path = build(native)
        L = C.CDLL(path)
        for name in ("orc_lsolve", "orc_ltsolve", "orc_usolve", "orc_utsolve"):
            getattr(L, name).argtypes = [C.c_int64, i64p, i64p, f64p, f64p]
            getattr(L, name).restype = None
        L.orc_gaxpy.argtypes = [C.c_int64, C.c_int64, i64p, i64p, f64p, f64p, f64p, f64p]
        L.orc_gaxpy.restype = None
        L.orc_amd.argtypes = [C.c_int64, C.c_int64, i64p, i64p, C.c_int, i64p]
        L.orc_schol.argtypes = [C.c_int64, i64p, i64p, C.c_int, i64p, i64p, i64p]
        L.orc_chol.argtypes = [C.c_int64, i64p, i64p, f64p, C.c_void_p, i64p, i64p, i64p, i64p, f64p]
        L.orc_cholsol.argtypes = [C.c_int64, i64p, i64p, f64p, f64p, C.c_int]
        L.orc_lu_new.argtypes = [C.c_int64, i64p, i64p, f64p, C.c_int, C.c_double, C.POINTER(C.c_int), i64p]
        L.orc_lu_new.restype = C.c_void_p
        L.orc_lu_nnz.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.orc_lu_get.argtypes = [C.c_void_p, i64p, i64p, f64p, i64p, i64p, f64p, i64p]
        L.orc_lu_free.argtypes = [C.c_void_p]
        L.orc_lusol.argtypes = [C.c_int64, i64p, i64p, f64p, f64p, C.c_int, C.c_double]
        L.orc_sqr_qr.argtypes = [C.c_int64, C.c_int64, i64p, i64p, f64p, C.c_int, i64p, i64p, i64p, i64p, i64p,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_solve_llt_many.argtypes = [C.c_int64, i64p, i64p, f64p, f64p, C.c_int64, C.c_int64]
        L.orc_solve_llt_many.restype = None
        if native:
            return L
        _LIB = L
    return _LIB


def _csc(A):
    """A = (m, n, p, i, x) -> contiguous int64/int64/float64 arrays."""
    m, n, p, i, x = A
    return (int(m), int(n), np.ascontiguousarray(p, np.int64), np.ascontiguousarray(i, np.int64),
            np.ascontiguousarray(x, np.float64))


def trisolve(kind, T, b):
    m, n, p, i, x = _csc(T)
    out = np.array(b, dtype=np.float64)
    getattr(lib(), "orc_" + kind)(n, p, i, x, out)
    return out


def gaxpy(A, x, y):
    m, n, p, i, v = _csc(A)
    r = np.zeros(m)
    lib().orc_gaxpy(m, n, p, i, v, np.ascontiguousarray(x, np.float64), np.ascontiguousarray(y, np.float64), r)
    return r


def amd(A, order):
    m, n, p, i, _ = _csc(A)
    P = np.zeros(n + 1, np.int64)
    has = lib().orc_amd(m, n, p, i, order, P)
    return P if has else None


def schol(A, order):
    m, n, p, i, _ = _csc(A)
    pinv = np.zeros(n, np.int64)
    parent = np.zeros(n, np.int64)
    cp = np.zeros(n + 1, np.int64)
    has = lib().orc_schol(n, p, i, order, pinv, parent, cp)
    return dict(pinv=pinv if has else None, parent=parent, cp=cp, lnz=int(cp[n]), unz=int(cp[n]))


def chol(A, S):
    m, n, p, i, x = _csc(A)
    lnz = int(S["cp"][n])
    Lp = np.zeros(n + 1, np.int64)
    Li = np.zeros(max(lnz, 1), np.int64)
    Lx = np.zeros(max(lnz, 1), np.float64)
    pinv = S["pinv"]
    pv = None if pinv is None else np.ascontiguousarray(pinv, np.int64)
    st = lib().orc_chol(n, p, i, x, None if pv is None else pv.ctypes.data, np.ascontiguousarray(S["parent"], np.int64),
                        np.ascontiguousarray(S["cp"], np.int64), Lp, Li, Lx)
    return st, (n, n, Lp, Li[:lnz], Lx[:lnz])


def cholsol(A, b, order):
    m, n, p, i, x = _csc(A)
    out = np.array(b, dtype=np.float64)
    st = lib().orc_cholsol(n, p, i, x, out, order)
    return st, out


def lu(A, order, tol):
    m, n, p, i, x = _csc(A)
    st = C.c_int(0)
    q = np.zeros(n, np.int64)
    h = lib().orc_lu_new(n, p, i, x, order, tol, C.byref(st), q)
    res = None
    if st.value == OK:
        lnz, unz = C.c_int64(0), C.c_int64(0)
        lib().orc_lu_nnz(h, C.byref(lnz), C.byref(unz))
        Lp = np.zeros(n + 1, np.int64); Up = np.zeros(n + 1, np.int64)
        Li = np.zeros(max(lnz.value, 1), np.int64); Ui = np.zeros(max(unz.value, 1), np.int64)
        Lx = np.zeros(max(lnz.value, 1)); Ux = np.zeros(max(unz.value, 1))
        pinv = np.zeros(n, np.int64)
        lib().orc_lu_get(h, Lp, Li, Lx, Up, Ui, Ux, pinv)
        res = dict(L=(n, n, Lp, Li[:lnz.value], Lx[:lnz.value]), U=(n, n, Up, Ui[:unz.value], Ux[:unz.value]),
                   pinv=pinv, q=q)
    lib().orc_lu_free(h)
    return st.value, res


def lusol(A, b, order, tol):
    m, n, p, i, x = _csc(A)
    out = np.array(b, dtype=np.float64)
    st = lib().orc_lusol(n, p, i, x, out, order, tol)
    return st, out


def sqr_qr(A, order, numeric=True):
    """sqr(a, order, qr=true) [+ qr]: only used to run the reference's integer pins."""
    m, n, p, i, x = _csc(A)
    q = np.zeros(n + 1, np.int64); parent = np.zeros(n, np.int64); cp = np.zeros(n, np.int64)
    pinv = np.zeros(2 * m + n, np.int64); scal = np.zeros(3, np.int64)
    has = lib().orc_sqr_qr(m, n, p, i, x, order, q, parent, cp, pinv, scal, None, None, None, None, None, None, None)
    out = dict(q=q if has else None, parent=parent, cp=cp, pinv=pinv, m2=int(scal[0]), lnz=int(scal[1]), unz=int(scal[2]))
    if numeric:
        Vp = np.zeros(n + 1, np.int64); Rp = np.zeros(n + 1, np.int64)
        Vi = np.zeros(out["lnz"], np.int64); Vx = np.zeros(out["lnz"])
        Ri = np.zeros(out["unz"], np.int64); Rx = np.zeros(out["unz"])
        beta = np.zeros(n)
        lib().orc_sqr_qr(m, n, p, i, x, order, q, parent, cp, pinv, scal, Vp.ctypes.data, Vi.ctypes.data, Vx.ctypes.data,
                         Rp.ctypes.data, Ri.ctypes.data, Rx.ctypes.data, beta.ctypes.data)
        out.update(Vp=Vp, Vi=Vi, Vx=Vx, Rp=Rp, Ri=Ri, Rx=Rx, beta=beta)
    return out


def _transpose(A):
    """transpose (lib.rs:1178-1200): rows of every column of the result ascend."""
    m, n, p, i, x = _csc(A)
    nz = int(p[n])
    cols = np.repeat(np.arange(n, dtype=np.int64), np.diff(p))
    order = np.argsort(i[:nz], kind="stable")
    tp = np.zeros(m + 1, np.int64)
    np.add.at(tp, i[:nz] + 1, 1)
    return (n, m, np.cumsum(tp), cols[order], x[:nz][order])


def _happly(f, k, x):
    """happly (lib.rs:2099-2111) with column k of V: sums in entry order, one product and one add at a time."""
    lo, hi = int(f["Vp"][k]), int(f["Vp"][k + 1])
    rows, v = f["Vi"][lo:hi], f["Vx"][lo:hi]
    tau = 0.0
    for e in range(hi - lo):
        tau += v[e] * x[rows[e]]
    tau *= f["beta"][k]
    for e in range(hi - lo):
        x[rows[e]] -= v[e] * tau


def qrsol(A, b, order):
    """qrsol (lib.rs:927-956) from the oracle's sqr + qr (orc_sqr_qr), happly above and orc_usolve / orc_utsolve.
    b holds max(m, n) entries; returns the overwritten copy."""
    m, n, p, i, x = _csc(A)
    b = np.array(b, dtype=np.float64)
    if m >= n:
        f = sqr_qr(A, order, numeric=True)
        R = (n, n, f["Rp"], f["Ri"], f["Rx"])
        w = np.zeros(f["m2"])
        w[f["pinv"][:m]] = b[:m]                                  # ipvec(m, pinv, b, x)
        for k in range(n):
            _happly(f, k, w)
        w[:n] = trisolve("usolve", R, w[:n])
        q = f["q"][:n] if f["q"] is not None else np.arange(n)
        b[q] = w[:n]                                              # ipvec(n, q, x, b)
    else:
        At = _transpose(A)
        f = sqr_qr(At, order, numeric=True)                       # At is n x m
        R = (m, m, f["Rp"], f["Ri"], f["Rx"])
        w = np.zeros(f["m2"])
        q = f["q"][:m] if f["q"] is not None else np.arange(m)
        w[:m] = b[q]                                              # pvec(m, q, b, x)
        w[:m] = trisolve("utsolve", R, w[:m])
        for k in range(m - 1, -1, -1):
            _happly(f, k, w)
        b[:n] = w[f["pinv"][:n]]                                  # pvec(n, pinv, x, b)
    return b
