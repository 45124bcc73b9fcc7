"""ctypes binding of include/rsparse_hip.h, shaped like rsparse's public Rust API (src/lib.rs, src/data.rs)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librsparse_hip.so")

RSP_OK, RSP_NOT_POSITIVE_DEFINITE, RSP_NO_PIVOT, RSP_BAD_INPUT, RSP_DEVICE_ERROR, RSP_INTERNAL_ERROR = 0, 1, 2, -1, -2, -3
LSOLVE, LTSOLVE, USOLVE, UTSOLVE = 0, 1, 2, 3
COL_MAJOR, ROW_MAJOR = 0, 1


class Error(Exception):
    """rsparse::Error (src/lib.rs:189-194)."""


class NotPositiveDefinite(Error):
    pass


class NoPivot(Error):
    pass


class DeviceError(RuntimeError):
    """No counterpart in the reference: bad input or HIP failure (negative status)."""


class _Csc(C.Structure):
    _fields_ = [("nzmax", C.c_int64), ("m", C.c_int64), ("n", C.c_int64), ("p", C.c_void_p), ("i", C.c_void_p),
                ("x", C.c_void_p)]


_lib = None


def lib():
    """The loaded C-ABI library. Raises if it was not built: the product has no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise ImportError(f"{_LIB_PATH} is missing: build it with `make -C rsparse_amd/csrc` (hipcc, gfx950)")
        # torch bundles its own libamdhip64.so.7; two HIP runtimes in one process cannot both own the GPU.
        # Loading torch first makes the dynamic loader resolve our DT_NEEDED against torch's copy.
        try:
            import torch  # noqa: F401  (plumbing only: device memory / streams / torch.distributed)
        except ImportError:
            pass
        L = C.CDLL(_LIB_PATH)
        vp, i64, dbl, cint = C.c_void_p, C.c_int64, C.c_double, C.c_int
        csc = C.POINTER(_Csc)
        sig = {
            "rsp_status_string": (C.c_char_p, [cint]),
            "rsp_device_count": (cint, []),
            "rsp_set_device": (cint, [cint]),
            "rsp_set_devices": (cint, [vp, cint]),
            "rsp_lsolve_f64": (cint, [csc, vp]),
            "rsp_ltsolve_f64": (cint, [csc, vp]),
            "rsp_usolve_f64": (cint, [csc, vp]),
            "rsp_utsolve_f64": (cint, [csc, vp]),
            "rsp_tri_analyze": (cint, [csc, cint, C.POINTER(vp)]),
            "rsp_tri_solve_f64": (cint, [vp, vp, i64, i64]),
            "rsp_tri_solve_dev_f64": (cint, [vp, vp, cint, i64, i64, vp]),
            "rsp_tri_solve_pair_dev_f64": (cint, [vp, vp, vp, i64, i64, vp]),
            "rsp_tri_solve_dev_noalloc_f64": (cint, [vp, vp, i64, i64, vp]),
            "rsp_tri_info": (cint, [vp, vp]),
            "rsp_tri_levels": (cint, [vp, vp]),
            "rsp_tri_free": (None, [vp]),
            "rsp_tri_validate_program": (cint, [csc, cint, i64, cint, vp, vp]),
            "rsp_tri_validate_refill": (cint, [csc, vp, cint, i64, cint, vp]),
            "rsp_tri_program_checksum": (cint, [csc, cint, i64, cint, vp]),
            "rsp_cache_stats": (cint, [vp]),
            "rsp_cache_clear": (cint, []),
            "rsp_selftest_host_threads": (cint, [i64, cint, C.c_uint64]),
            "rsp_schol": (cint, [csc, C.c_int8, C.POINTER(vp)]),
            "rsp_sqr_qr": (cint, [csc, C.c_int8, C.POINTER(vp)]),
            "rsp_symb_m2": (i64, [vp]),
            "rsp_symb_pinv_len": (i64, [vp]),
            "rsp_symb_cp_len": (i64, [vp]),
            "rsp_qr_f64": (cint, [csc, vp, C.POINTER(vp)]),
            "rsp_qr_V": (cint, [vp, csc]),
            "rsp_qr_R": (cint, [vp, csc]),
            "rsp_qr_beta": (vp, [vp]),
            "rsp_qr_solve_f64": (cint, [vp, vp, vp, cint]),
            "rsp_qr_free": (None, [vp]),
            "rsp_qrsol_f64": (cint, [csc, vp, C.c_int8]),
            "rsp_sqr_lu": (cint, [csc, C.c_int8, C.POINTER(vp)]),
            "rsp_symb_from_parts": (cint, [i64, vp, vp, vp, vp, C.POINTER(vp)]),
            "rsp_symb_n": (i64, [vp]),
            "rsp_symb_pinv": (vp, [vp]),
            "rsp_symb_q": (vp, [vp]),
            "rsp_symb_parent": (vp, [vp]),
            "rsp_symb_cp": (vp, [vp]),
            "rsp_symb_lnz": (i64, [vp]),
            "rsp_symb_unz": (i64, [vp]),
            "rsp_symb_free": (None, [vp]),
            "rsp_amd": (cint, [csc, C.c_int8, vp, C.POINTER(cint)]),
            "rsp_chol_f64": (cint, [csc, vp, vp, vp, vp]),
            "rsp_chol_factor_f64": (cint, [csc, vp, C.POINTER(vp)]),
            "rsp_chol_refactor_f64": (cint, [vp, csc]),
            "rsp_chol_refine_dev_f64": (cint, [vp, vp, vp, vp, vp, cint, i64, i64, cint, vp]),
            "rsp_chol_solve_f64": (cint, [vp, vp, i64, i64]),
            "rsp_chol_solve_dev_f64": (cint, [vp, vp, cint, i64, i64, cint, vp]),
            "rsp_chol_lnz": (i64, [vp]),
            "rsp_chol_get_factor": (cint, [vp, vp, vp, vp]),
            "rsp_chol_factor_view": (cint, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]),
            "rsp_sprs_save_bin": (cint, [csc, C.c_char_p]),
            "rsp_sprs_load_bin_dims": (cint, [C.c_char_p, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]),
            "rsp_sprs_load_bin": (cint, [C.c_char_p, vp, vp, vp]),
            "rsp_chol_info": (cint, [vp, vp]),
            "rsp_chol_factor_ms": (dbl, [vp]),
            "rsp_chol_free": (None, [vp]),
            "rsp_cholsol_f64": (cint, [csc, vp, C.c_int8]),
            "rsp_cholsol_batch_f64": (cint, [csc, vp, vp, i64, C.c_int8, vp]),
            "rsp_lu_f64": (cint, [csc, vp, dbl, C.POINTER(vp)]),
            "rsp_lu_L": (cint, [vp, csc]),
            "rsp_lu_U": (cint, [vp, csc]),
            "rsp_lu_pinv": (vp, [vp]),
            "rsp_lu_on_device": (cint, [vp]),
            "rsp_lu_refactor_f64": (cint, [vp, csc, dbl]),
            "rsp_lu_static_pattern_f64": (cint, [csc, vp, C.POINTER(vp), C.POINTER(i64)]),
            "rsp_lu_solve_f64": (cint, [vp, vp, vp, i64, i64]),
            "rsp_lu_solve_dev_f64": (cint, [vp, vp, vp, cint, i64, i64, vp]),
            "rsp_gaxpy_f64": (cint, [csc, vp, vp, vp]),
            "rsp_mat_create_f64": (cint, [csc, C.POINTER(vp)]),
            "rsp_mat_gaxpy_dev_f64": (cint, [vp, vp, i64, vp, i64, i64, cint, dbl, vp]),
            "rsp_mat_free": (None, [vp]),
            "rsp_lu_free": (None, [vp]),
            "rsp_lusol_f64": (cint, [csc, vp, C.c_int8, dbl]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        L._declared = sorted(sig)
        _lib = L
    return _lib


def _check(status):
    if status == RSP_OK:
        return
    if status == RSP_NOT_POSITIVE_DEFINITE:
        raise NotPositiveDefinite(lib().rsp_status_string(status).decode())
    if status == RSP_NO_PIVOT:
        raise NoPivot(lib().rsp_status_string(status).decode())
    raise DeviceError(f"rsparse_hip status {status}: {lib().rsp_status_string(status).decode()}")


def device_count():
    return lib().rsp_device_count()


def set_devices(ids):
    """Devices that cholsol_batch spreads its systems over, and CholHandle.solve / TriHandle.solve their right-hand-side columns
    (empty list: the current device only)."""
    a = np.ascontiguousarray(ids, np.int32)
    _check(lib().rsp_set_devices(a.ctypes.data if len(a) else None, len(a)))


def _vec(ptr, n, dtype):
    if not ptr:
        return None
    ct = C.c_int64 if dtype == np.int64 else C.c_double
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,)).copy()


class Sprs:
    """rsparse::data::Sprs<f64> (src/data.rs:194-208): CSC with p: i64[n+1], i: u64[nnz], x: f64[nnz]."""

    def __init__(self, m=0, n=0, p=None, i=None, x=None):
        self.m, self.n = int(m), int(n)
        self.p = np.ascontiguousarray(p if p is not None else np.zeros(n + 1), dtype=np.int64)
        self.i = np.ascontiguousarray(i if i is not None else [], dtype=np.uint64)
        self.x = np.ascontiguousarray(x if x is not None else [], dtype=np.float64)
        self.nzmax = len(self.x)

    @classmethod
    def from_tuple(cls, t):
        m, n, p, i, x = t
        return cls(m, n, p, i, x)

    @classmethod
    def from_vec(cls, rows):
        """Sprs::from_vec (src/data.rs:289-314): column by column, exact zeros dropped."""
        a = np.asarray(rows, dtype=np.float64)
        m, n = a.shape
        mask = (a != 0.0).T  # column-major traversal
        cols, rws = np.nonzero(mask)
        p = np.zeros(n + 1, np.int64)
        np.add.at(p, cols + 1, 1)
        return cls(m, n, np.cumsum(p), rws, a.T[mask])

    def as_tuple(self):
        return (self.m, self.n, self.p, self.i.astype(np.int64), self.x)

    def save(self, path):
        """Sprs::save (src/data.rs:414-431): six text lines `nzmax: ..`, `m: ..`, `n: ..`, `p: [..]`, `i: [..]`, `x: [..]`, the
        vectors in Rust's Debug form (floats in their shortest round-trip decimal, so the values survive exactly)."""
        def flt(v):
            r = repr(float(v))
            if "e" in r:  # Rust prints 1e-7 / 1e16 where Python prints 1e-07 / 1e+16
                mant, ex = r.split("e")
                r = mant + "e" + str(int(ex))
            return r
        with open(path, "w") as f:
            f.write(f"nzmax: {self.nzmax}\nm: {self.m}\nn: {self.n}\n")
            f.write("p: [" + ", ".join(str(int(v)) for v in self.p) + "]\n")
            f.write("i: [" + ", ".join(str(int(v)) for v in self.i) + "]\n")
            f.write("x: [" + ", ".join(flt(v) for v in self.x) + "]\n")

    @classmethod
    def load(cls, path):
        """Sprs::load (src/data.rs:433-516): parse the six lines Sprs::save writes; like the reference, a file whose nzmax, m
        or n is 0 yields the empty matrix, and a missing or malformed line is an error (ValueError here)."""
        fields = {}
        with open(path) as f:
            for line in f:
                key, sep, val = line.partition(":")
                if sep:
                    fields[key.strip()] = val.strip()
        try:
            nzmax, m, n = int(fields["nzmax"]), int(fields["m"]), int(fields["n"])
            if nzmax == 0 or m == 0 or n == 0:
                return cls()
            vec = lambda key, conv: [conv(t) for t in fields[key].strip("[]").split(",") if t.strip()]
            p, i, x = vec("p", int), vec("i", int), vec("x", float)
        except (KeyError, ValueError) as e:
            raise ValueError(f"{path}: not in the Sprs text format ({e})") from None
        if len(p) != n + 1 or len(i) != len(x) or p[-1] > len(x):
            raise ValueError(f"{path}: inconsistent Sprs (n={n}, len(p)={len(p)}, len(i)={len(i)}, len(x)={len(x)})")
        return cls(m, n, p, i, x)

    def save_bin(self, path):
        """Binary interchange format (rsp_sprs_save_bin, include/rsparse_hip.h): header + the three arrays as they sit in memory."""
        _check(lib().rsp_sprs_save_bin(C.byref(self._view()), os.fsencode(path)))

    @classmethod
    def load_bin(cls, path):
        """ValueError for a file that is missing, truncated, inconsistent or not in this format (RSP_BAD_INPUT below)."""
        m, n, nnz = C.c_int64(), C.c_int64(), C.c_int64()
        if lib().rsp_sprs_load_bin_dims(os.fsencode(path), C.byref(m), C.byref(n), C.byref(nnz)) != RSP_OK:
            raise ValueError(f"{path}: cannot be read as an RSPSPRS1 file")
        p, i, x = np.zeros(n.value + 1, np.int64), np.zeros(nnz.value, np.uint64), np.zeros(nnz.value)
        if lib().rsp_sprs_load_bin(os.fsencode(path), p.ctypes.data, i.ctypes.data if nnz.value else None,
                                   x.ctypes.data if nnz.value else None) != RSP_OK:
            raise ValueError(f"{path}: truncated or inconsistent RSPSPRS1 file")
        return cls(m.value, n.value, p, i, x)

    def _view(self):
        v = _Csc(self.nzmax, self.m, self.n, self.p.ctypes.data, self.i.ctypes.data, self.x.ctypes.data)
        v._keep = self
        return v


def _as_sprs(a):
    return a if isinstance(a, Sprs) else Sprs.from_tuple(a)


def _one_shot(name, t, x):
    t = _as_sprs(t)
    if not (isinstance(x, np.ndarray) and x.dtype == np.float64 and x.flags.c_contiguous):
        raise TypeError("x must be a contiguous float64 numpy array (it is overwritten, like the reference's &mut [f64])")
    _check(getattr(lib(), name)(C.byref(t._view()), x.ctypes.data))


def lsolve(l, x):
    """rsparse::lsolve (src/lib.rs:464): x <- L^-1 x, diagonal first in every column."""
    _one_shot("rsp_lsolve_f64", l, x)


def ltsolve(l, x):
    """rsparse::ltsolve (src/lib.rs:505): x <- L^-T x."""
    _one_shot("rsp_ltsolve_f64", l, x)


def usolve(u, x):
    """rsparse::usolve (src/lib.rs:1230): x <- U^-1 x, diagonal last in every column."""
    _one_shot("rsp_usolve_f64", u, x)


def utsolve(u, x):
    """rsparse::utsolve (src/lib.rs:1271): x <- U^-T x."""
    _one_shot("rsp_utsolve_f64", u, x)


class Symb:
    """rsparse::data::Symb (src/data.rs:1022-1038), owned by the library."""

    def __init__(self, handle):
        self._h = C.c_void_p(handle)
        L = lib()
        self.n = L.rsp_symb_n(self._h)
        self.pinv = _vec(L.rsp_symb_pinv(self._h), L.rsp_symb_pinv_len(self._h), np.int64)  # n (schol) or 2m + n (sqr, qr)
        self.q = _vec(L.rsp_symb_q(self._h), self.n + 1, np.int64)
        self.parent = _vec(L.rsp_symb_parent(self._h), self.n, np.int64)
        self.cp = _vec(L.rsp_symb_cp(self._h), L.rsp_symb_cp_len(self._h), np.int64)  # n + 1 (schol) or n (sqr, qr)
        self.lnz = L.rsp_symb_lnz(self._h)
        self.unz = L.rsp_symb_unz(self._h)
        self.m2 = L.rsp_symb_m2(self._h)

    @classmethod
    def from_parts(cls, n, pinv=None, q=None, parent=None, cp=None):
        def ptr(a):
            if a is None:
                return None, None
            a = np.ascontiguousarray(a, np.int64)
            return a, a.ctypes.data
        keep = [ptr(pinv), ptr(q), ptr(parent), ptr(cp)]
        h = C.c_void_p()
        _check(lib().rsp_symb_from_parts(n, keep[0][1], keep[1][1], keep[2][1], keep[3][1], C.byref(h)))
        return cls(h.value)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.rsp_symb_free(self._h)
            self._h = None


def schol(a, order):
    """rsparse::schol (src/lib.rs:968): ordering + symbolic Cholesky analysis (host)."""
    a = _as_sprs(a)
    h = C.c_void_p()
    _check(lib().rsp_schol(C.byref(a._view()), order, C.byref(h)))
    return Symb(h.value)


def sqr(a, order, qr=False):
    """rsparse::sqr (src/lib.rs:1114): ordering and symbolic analysis for LU (qr=False) or QR (qr=True), on the host."""
    a = _as_sprs(a)
    h = C.c_void_p()
    _check((lib().rsp_sqr_qr if qr else lib().rsp_sqr_lu)(C.byref(a._view()), order, C.byref(h)))
    return Symb(h.value)


def amd(a, order):
    """rsparse's private amd (src/lib.rs:1292): returns the n+1-entry permutation or None."""
    a = _as_sprs(a)
    P = np.zeros(a.n + 1, np.int64)
    has = C.c_int(0)
    _check(lib().rsp_amd(C.byref(a._view()), order, P.ctypes.data, C.byref(has)))
    return P if has.value else None


class Nmrc:
    """rsparse::data::Nmrc (src/data.rs:1064-1074)."""

    def __init__(self, l=None, u=None, pinv=None):
        self.l, self.u, self.pinv, self.b = l, u, pinv, []


def chol(a, s):
    """rsparse::chol (src/lib.rs:278): numeric Cholesky on the GPU; returns Nmrc with .l (pinv None)."""
    a = _as_sprs(a)
    n, lnz = a.n, int(s.cp[a.n])
    Lp = np.zeros(n + 1, np.int64)
    Li = np.zeros(max(lnz, 1), np.uint64)
    Lx = np.zeros(max(lnz, 1), np.float64)
    _check(lib().rsp_chol_f64(C.byref(a._view()), s._h, Lp.ctypes.data, Li.ctypes.data, Lx.ctypes.data))
    return Nmrc(l=Sprs(n, n, Lp, Li[:lnz], Lx[:lnz]))


def cholsol(a, b, order):
    """rsparse::cholsol (src/lib.rs:377): b <- A^-1 b."""
    a = _as_sprs(a)
    assert b.dtype == np.float64 and b.flags.c_contiguous
    _check(lib().rsp_cholsol_f64(C.byref(a._view()), b.ctypes.data, order))


def cholsol_batch(a, values, B, order):
    """cholsol (src/lib.rs:377) for every system of a batch that shares the pattern of `a`.

    values: (nsys, nnz) array, row s = the values of system s in a's entry order; B: (nsys, n), overwritten by the
    solutions.  Returns an int32 array of per-system status (0 = ok, 1 = NotPositiveDefinite, right-hand side kept).
    """
    a = _as_sprs(a)
    nsys = values.shape[0]
    assert values.dtype == np.float64 and values.flags.c_contiguous and values.shape == (nsys, int(a.p[-1]))
    assert B.dtype == np.float64 and B.flags.c_contiguous and B.shape == (nsys, a.n)
    status = np.zeros(nsys, np.int32)
    _check(lib().rsp_cholsol_batch_f64(C.byref(a._view()), values.ctypes.data, B.ctypes.data, nsys, order, status.ctypes.data))
    return status


class LuHandle:
    def __init__(self, handle, symb):
        self._h = C.c_void_p(handle)
        self._s = symb

    def _mat(self, getter):
        v = _Csc()
        _check(getter(self._h, C.byref(v)))
        nz = v.nzmax
        p = _vec(v.p, v.n + 1, np.int64)
        i = _vec(v.i, nz, np.int64) if nz else np.zeros(0, np.int64)
        x = _vec(v.x, nz, np.float64) if nz else np.zeros(0)
        return Sprs(v.m, v.n, p, i, x)

    @property
    def l(self):
        return self._mat(lib().rsp_lu_L)

    @property
    def u(self):
        return self._mat(lib().rsp_lu_U)

    @property
    def pinv(self):
        return _vec(lib().rsp_lu_pinv(self._h), self._s.n, np.int64)

    @property
    def on_device(self):
        """True when the values were computed by the GPU numeric LU (static pivots held), False: by the host routine."""
        return bool(lib().rsp_lu_on_device(self._h))

    def refactor(self, a, tol):
        """Factor `a` (same pattern as the matrix this handle was made from, new values) in place of the old factors."""
        a = _as_sprs(a)
        _check(lib().rsp_lu_refactor_f64(self._h, C.byref(a._view()), tol))

    def solve(self, B):
        """B: float64, shape (n,) or Fortran-ordered (n, nrhs); overwritten with A^-1 B."""
        n = self._s.n
        nrhs = 1 if B.ndim == 1 else B.shape[1]
        assert B.dtype == np.float64 and (B.ndim == 1 or B.flags.f_contiguous)
        _check(lib().rsp_lu_solve_f64(self._h, self._s._h, B.ctypes.data, n, nrhs))

    def solve_dev(self, ptr, layout, ld, nrhs, stream=None):
        """Device-resident block (device pointer `ptr`) overwritten with A^-1 B; asynchronous on `stream`."""
        _check(lib().rsp_lu_solve_dev_f64(self._h, self._s._h, ptr, layout, ld, nrhs, stream))

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.rsp_lu_free(self._h)
            self._h = None


def lu(a, s, tol):
    """rsparse::lu (src/lib.rs:519): LU with the reference's pivot rule -- on the GPU when the diagonal pivots hold, else on
    the host; returns a handle with .l .u .pinv (.on_device tells which)."""
    a = _as_sprs(a)
    h = C.c_void_p()
    _check(lib().rsp_lu_f64(C.byref(a._view()), s._h, tol, C.byref(h)))
    return LuHandle(h.value, s)


def lu_static_pattern(a, s):
    """The integer half of the GPU numeric LU on the host: (handle with the patterns of L, U and pinv for diagonal pivots,
    values zero; depth of the column dependency levels)."""
    a = _as_sprs(a)
    h = C.c_void_p()
    depth = C.c_int64(0)
    _check(lib().rsp_lu_static_pattern_f64(C.byref(a._view()), s._h, C.byref(h), C.byref(depth)))
    return LuHandle(h.value, s), depth.value


def gaxpy(a, x, y):
    """rsparse::gaxpy (src/lib.rs:411-421): returns A x + y (one vector), computed on the GPU."""
    a = _as_sprs(a)
    x = np.ascontiguousarray(x, np.float64)
    y = np.ascontiguousarray(y, np.float64)
    assert x.shape == (a.n,) and y.shape == (a.m,)
    r = np.empty(a.m)
    _check(lib().rsp_gaxpy_f64(C.byref(a._view()), x.ctypes.data, y.ctypes.data, r.ctypes.data))
    return r


class MatHandle:
    """A sparse matrix resident on the device for gaxpy over blocks of right-hand sides: Y <- A X + beta Y."""

    def __init__(self, a):
        a = _as_sprs(a)
        self.m, self.n = a.m, a.n
        h = C.c_void_p()
        _check(lib().rsp_mat_create_f64(C.byref(a._view()), C.byref(h)))
        self._h = h

    def gaxpy_dev(self, x_ptr, ldx, y_ptr, ldy, nrhs, layout, beta=1.0, stream=None):
        _check(lib().rsp_mat_gaxpy_dev_f64(self._h, x_ptr, ldx, y_ptr, ldy, nrhs, layout, beta, stream))

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.rsp_mat_free(self._h)
            self._h = None


class QrHandle:
    """rsparse::data::Nmrc of a QR factorisation (.l = V, .u = R, .b = beta), owned by the library."""

    def __init__(self, handle, symb):
        self._h = C.c_void_p(handle)
        self._s = symb

    _mat = LuHandle._mat

    @property
    def l(self):
        return self._mat(lib().rsp_qr_V)

    @property
    def u(self):
        return self._mat(lib().rsp_qr_R)

    @property
    def b(self):
        return _vec(lib().rsp_qr_beta(self._h), self._s.n, np.float64)

    def solve(self, b, transposed=False):
        """The solve phase of qrsol (src/lib.rs:936-942, or :949-954 when the handle factors the transpose); b is overwritten."""
        assert b.dtype == np.float64 and b.flags.c_contiguous and b.ndim == 1
        _check(lib().rsp_qr_solve_f64(self._h, self._s._h, b.ctypes.data, 1 if transposed else 0))

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.rsp_qr_free(self._h)
            self._h = None


def qr(a, s):
    """rsparse::qr (src/lib.rs:788): Householder QR on the host; returns a handle with .l (V), .u (R), .b (beta)."""
    a = _as_sprs(a)
    h = C.c_void_p()
    _check(lib().rsp_qr_f64(C.byref(a._view()), s._h, C.byref(h)))
    return QrHandle(h.value, s)


def qrsol(a, b, order):
    """rsparse::qrsol (src/lib.rs:927): least-squares (m >= n) or minimum-norm (m < n) solution in b, which holds
    max(m, n) entries; the triangular solve with R runs on the GPU."""
    a = _as_sprs(a)
    assert b.dtype == np.float64 and b.flags.c_contiguous and b.shape == (max(a.m, a.n),)
    _check(lib().rsp_qrsol_f64(C.byref(a._view()), b.ctypes.data, order))


def transpose(a):
    """rsparse::transpose (src/lib.rs:1178): host-side helper of the test-suite."""
    a = _as_sprs(a)
    nz = int(a.p[a.n])
    p = np.zeros(a.m + 1, np.int64)
    np.add.at(p, a.i[:nz].astype(np.int64) + 1, 1)
    p = np.cumsum(p)
    cols = np.repeat(np.arange(a.n), np.diff(a.p))
    order = np.argsort(a.i[:nz].astype(np.int64), kind="stable")
    return Sprs(a.n, a.m, p, cols[order], a.x[:nz][order])


def lusol(a, b, order, tol):
    """rsparse::lusol (src/lib.rs:672): b <- A^-1 b."""
    a = _as_sprs(a)
    assert b.dtype == np.float64 and b.flags.c_contiguous
    _check(lib().rsp_lusol_f64(C.byref(a._view()), b.ctypes.data, order, tol))


def validate_program(t, kind, task_blocks=0, cluster_slots=0):
    """Host-only check of the multi-RHS programme built for `t` (rsp_tri_validate_program); no GPU needed."""
    t = _as_sprs(t)
    rep = np.zeros(8, np.int64)
    err = C.c_double(0.0)
    _check(lib().rsp_tri_validate_program(C.byref(t._view()), kind, task_blocks, cluster_slots, rep.ctypes.data, C.byref(err)))
    keys = ("bad_index", "order_violations", "unwritten", "levels", "tasks", "clusters", "cluster_rows", "launches")
    d = dict(zip(keys, rep.tolist()))
    d["max_rel_error"] = err.value
    return d


def validate_refill(t, new_values, kind, task_blocks=0, cluster_slots=0):
    """Host-only check that new values poured into the programme of `t` through its recipe give, bit for bit, the programme
    emitted from scratch for those values (rsp_tri_validate_refill); no GPU needed."""
    t = _as_sprs(t)
    v = np.ascontiguousarray(new_values, np.float64)
    assert v.shape == (int(t.p[t.n]),)
    rep = np.zeros(4, np.int64)
    _check(lib().rsp_tri_validate_refill(C.byref(t._view()), v.ctypes.data, kind, task_blocks, cluster_slots, rep.ctypes.data))
    return dict(zip(("has_recipe", "coef_mismatches", "group_mismatches", "coef_entries"), rep.tolist()))


def program_checksum(t, kind, task_blocks=0, cluster_slots=0):
    """Host-only fingerprint (four 64-bit hashes) of the multi-RHS programme built for `t` (rsp_tri_program_checksum)."""
    t = _as_sprs(t)
    out = np.zeros(4, np.uint64)
    _check(lib().rsp_tri_program_checksum(C.byref(t._view()), kind, task_blocks, cluster_slots, out.ctypes.data))
    return tuple(int(v) for v in out)


def cache_stats():
    """The analysis cache behind cholsol (rsp_cache_stats)."""
    a = np.zeros(7, np.int64)
    _check(lib().rsp_cache_stats(a.ctypes.data))
    return dict(zip(("hits", "misses", "entries", "bytes", "pending", "inserted", "evicted"), a.tolist()))


def cache_clear():
    _check(lib().rsp_cache_clear())


def solve_pair_dev(first, second, ptr, ld, nrhs, stream=None):
    """first.solve then second.solve on one row-major device block (rsp_tri_solve_pair_dev_f64)."""
    _check(lib().rsp_tri_solve_pair_dev_f64(first._h, second._h, ptr, ld, nrhs, stream))


class TriHandle:
    """Analyse-once / solve-many triangular solver (rsp_tri_*)."""

    def __init__(self, t, kind):
        t = _as_sprs(t)
        h = C.c_void_p()
        _check(lib().rsp_tri_analyze(C.byref(t._view()), kind, C.byref(h)))
        self._h, self.n, self.kind = h, t.n, kind

    def solve(self, X):
        """X float64: (n,) or Fortran-ordered (n, nrhs) host array, overwritten."""
        nrhs = 1 if X.ndim == 1 else X.shape[1]
        assert X.dtype == np.float64 and (X.ndim == 1 or X.flags.f_contiguous)
        _check(lib().rsp_tri_solve_f64(self._h, X.ctypes.data, self.n, nrhs))

    def solve_dev(self, ptr, layout, ld, nrhs, stream=None):
        _check(lib().rsp_tri_solve_dev_f64(self._h, ptr, layout, ld, nrhs, stream))

    def solve_dev_noalloc(self, ptr, ld, nrhs, stream=None):
        """Row-major device block, scratch rows in the library's fixed device array (rsp_tri_solve_dev_noalloc_f64)."""
        _check(lib().rsp_tri_solve_dev_noalloc_f64(self._h, ptr, ld, nrhs, stream))

    def info(self):
        a = np.zeros(11, np.int64)
        _check(lib().rsp_tri_info(self._h, a.ctypes.data))
        keys = ("n", "nnz", "row_levels", "tasks", "launches", "device_bytes", "bytes_1rhs", "inverted_rows", "scratch_rows", "items",
                "item_entries")
        return dict(zip(keys, a.tolist()))

    def levels(self):
        a = np.zeros(self.n, np.int64)
        _check(lib().rsp_tri_levels(self._h, a.ctypes.data))
        return a

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.rsp_tri_free(self._h)
            self._h = None


class CholHandle:
    """Device-resident Cholesky factor (rsp_chol_*): factor once, solve many."""

    def __init__(self, a, s):
        a = _as_sprs(a)
        h = C.c_void_p()
        _check(lib().rsp_chol_factor_f64(C.byref(a._view()), s._h, C.byref(h)))
        self._h, self.n = h, a.n

    def solve(self, B):
        nrhs = 1 if B.ndim == 1 else B.shape[1]
        assert B.dtype == np.float64 and (B.ndim == 1 or B.flags.f_contiguous)
        _check(lib().rsp_chol_solve_f64(self._h, B.ctypes.data, self.n, nrhs))

    def refactor(self, a):
        """New values on the analysed pattern (chol(a, &mut s) called again, lib.rs:278): numeric factorisation only."""
        a = _as_sprs(a)
        _check(lib().rsp_chol_refactor_f64(self._h, C.byref(a._view())))

    def solve_dev(self, ptr, layout, ld, nrhs, permute=True, stream=None):
        _check(lib().rsp_chol_solve_dev_f64(self._h, ptr, layout, ld, nrhs, 1 if permute else 0, stream))

    def factor(self):
        lnz = lib().rsp_chol_lnz(self._h)
        Lp = np.zeros(self.n + 1, np.int64)
        Li = np.zeros(max(lnz, 1), np.uint64)
        Lx = np.zeros(max(lnz, 1))
        _check(lib().rsp_chol_get_factor(self._h, Lp.ctypes.data, Li.ctypes.data, Lx.ctypes.data))
        return Sprs(self.n, self.n, Lp, Li[:lnz], Lx[:lnz])

    def refine_dev(self, mat, b_ptr, x_ptr, w_ptr, layout, ld, nrhs, steps=1, stream=None):
        """`steps` rounds of iterative refinement of the solved block at x_ptr (right-hand sides at b_ptr, scratch at
        w_ptr; device addresses, one layout and leading dimension), asynchronous on `stream`."""
        _check(lib().rsp_chol_refine_dev_f64(self._h, mat._h, b_ptr, x_ptr, w_ptr, layout, ld, nrhs, steps, stream))

    def factor_view(self):
        """Zero-copy view of the factor: (Lp, Li) numpy views of the library's host pattern, and the DEVICE address of the lnz
        values (wrap it with torch / pass it to a kernel).  Borrowed: valid until the handle is freed or refactored."""
        Lp, Li, dLx = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(lib().rsp_chol_factor_view(self._h, C.byref(Lp), C.byref(Li), C.byref(dLx)))
        lnz = lib().rsp_chol_lnz(self._h)
        as_np = lambda ptr, cnt, ty, dt: np.frombuffer((ty * cnt).from_address(ptr.value), dtype=dt) if cnt else np.zeros(0, dt)
        return as_np(Lp, self.n + 1, C.c_int64, np.int64), as_np(Li, lnz, C.c_uint64, np.uint64), dLx.value

    def info(self):
        a = np.zeros(10, np.int64)
        _check(lib().rsp_chol_info(self._h, a.ctypes.data))
        keys = ("n", "lnz", "supernodes", "chol_levels", "flops", "bytes_chol", "lsolve_launches", "lsolve_tasks",
                "ltsolve_launches", "ltsolve_tasks")
        d = dict(zip(keys, a.tolist()))
        d["factor_ms"] = lib().rsp_chol_factor_ms(self._h)
        return d

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.rsp_chol_free(self._h)
            self._h = None
