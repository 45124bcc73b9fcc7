// UNVERIFIED: there is no Rust toolchain in the image this library is built and tested in, so this file has never
// been compiled.  It is the binding INTEGRATION.md section 2 describes, kept as a file so that a maintainer can start
// from it; the verified consumers of the same C ABI are rsparse_amd/api.py (ctypes) and the tests that drive it.
// Place it at src/hip.rs of the rsparse crate and add `mod hip;` to src/lib.rs.
use crate::data::{Sprs, Symb, Nmrc};
use crate::Error;

#[repr(C)]
struct Csc { nzmax: i64, m: i64, n: i64, p: *const isize, i: *const usize, x: *const f64 }   // rsp_csc_f64
#[repr(C)] struct RspSymb { _p: [u8; 0] }
#[repr(C)] struct RspChol { _p: [u8; 0] }
#[repr(C)] struct RspLu { _p: [u8; 0] }

extern "C" {
    fn rsp_lsolve_f64 (l: *const Csc, x: *mut f64) -> i32;     // replaces lsolve   src/lib.rs:464
    fn rsp_ltsolve_f64(l: *const Csc, x: *mut f64) -> i32;     // replaces ltsolve  src/lib.rs:505
    fn rsp_usolve_f64 (u: *const Csc, x: *mut f64) -> i32;     // replaces usolve   src/lib.rs:1230
    fn rsp_utsolve_f64(u: *const Csc, x: *mut f64) -> i32;     // replaces utsolve  src/lib.rs:1271
    fn rsp_cholsol_f64(a: *const Csc, b: *mut f64, order: i8) -> i32;            // cholsol src/lib.rs:377
    fn rsp_lusol_f64  (a: *const Csc, b: *mut f64, order: i8, tol: f64) -> i32;  // lusol   src/lib.rs:672
    fn rsp_qrsol_f64  (a: *const Csc, b: *mut f64, order: i8) -> i32;            // qrsol   src/lib.rs:927 (b: max(m, n) entries)
    // a loop of cholsol calls (src/lib.rs:377) over systems with ONE pattern: one analysis, one launch sequence
    fn rsp_cholsol_batch_f64(a: *const Csc, ax_all: *const f64, b: *mut f64, nsys: i64, order: i8, status: *mut i32) -> i32;
    fn rsp_symb_from_parts(n: i64, pinv: *const isize, q: *const isize, parent: *const isize, cp: *const isize,
                           s: *mut *mut RspSymb) -> i32;
    fn rsp_chol_f64(a: *const Csc, s: *const RspSymb, lp: *mut isize, li: *mut usize, lx: *mut f64) -> i32; // chol :278
    fn rsp_symb_free(s: *mut RspSymb);
    // lu (src/lib.rs:519) on a Symb from sqr: the reference's factors bit for bit (values from the GPU when every column keeps
    // its diagonal as pivot, else from the host routine); L, U, pinv are borrowed views of Nmrc.l / .u / .pinv
    fn rsp_sqr_lu(a: *const Csc, order: i8, s: *mut *mut RspSymb) -> i32;                      // sqr(a, order, false) :1114
    fn rsp_lu_f64(a: *const Csc, s: *const RspSymb, tol: f64, n: *mut *mut RspLu) -> i32;      // lu :519
    fn rsp_lu_L(n: *const RspLu, l: *mut Csc) -> i32;
    fn rsp_lu_U(n: *const RspLu, u: *mut Csc) -> i32;
    fn rsp_lu_pinv(n: *const RspLu) -> *const isize;
    fn rsp_lu_refactor_f64(n: *mut RspLu, a: *const Csc, tol: f64) -> i32;                     // new values, same pattern (no counterpart)
    fn rsp_lu_free(n: *mut RspLu);
}

fn view(a: &Sprs<f64>) -> Csc {
    Csc { nzmax: a.nzmax as i64, m: a.m as i64, n: a.n as i64, p: a.p.as_ptr(), i: a.i.as_ptr(), x: a.x.as_ptr() }
}

/// 0 -> Ok, 1 -> Error::NotPositiveDefinite, 2 -> Error::NoPivot (src/lib.rs:189-194).
/// Negative codes (bad input, no GPU) have no variant in the public enum: the caller falls back to the generic loop.
fn status(code: i32) -> Option<Result<(), Error>> {
    match code { 0 => Some(Ok(())), 1 => Some(Err(Error::NotPositiveDefinite)), 2 => Some(Err(Error::NoPivot)), _ => None }
}

pub(crate) fn lsolve_f64(l: &Sprs<f64>, x: &mut [f64]) -> bool { unsafe { rsp_lsolve_f64(&view(l), x.as_mut_ptr()) == 0 } }
pub(crate) fn cholsol_f64(a: &Sprs<f64>, b: &mut [f64], order: i8) -> Option<Result<(), Error>> {
    status(unsafe { rsp_cholsol_f64(&view(a), b.as_mut_ptr(), order) })
}
/// true when the library solved the system; false (negative code): the caller runs the generic qrsol body.
pub(crate) fn qrsol_f64(a: &Sprs<f64>, b: &mut [f64], order: i8) -> bool {
    b.len() >= a.m.max(a.n) && unsafe { rsp_qrsol_f64(&view(a), b.as_mut_ptr(), order) } == 0
}
/// New public helper a maintainer may expose next to `cholsol`: `systems` share `pattern`'s p/i, `values[s]` are the
/// entries of system s, `b[s]` its right-hand side.  Per-system result as `cholsol` would have returned it.
pub fn cholsol_batch(pattern: &Sprs<f64>, values: &[f64], b: &mut [f64], order: i8) -> Option<Vec<Result<(), Error>>> {
    let (n, nnz) = (pattern.n, pattern.p[pattern.n] as usize);
    let nsys = b.len() / n.max(1);
    assert!(values.len() == nsys * nnz && b.len() == nsys * n);
    let mut st = vec![0i32; nsys];
    let rc = unsafe { rsp_cholsol_batch_f64(&view(pattern), values.as_ptr(), b.as_mut_ptr(), nsys as i64, order, st.as_mut_ptr()) };
    if rc != 0 { return None; } // negative codes: caller falls back to the generic CPU loop
    Some(st.iter().map(|&c| if c == 0 { Ok(()) } else { Err(Error::NotPositiveDefinite) }).collect())
}
pub(crate) fn chol_f64(a: &Sprs<f64>, s: &Symb) -> Option<Result<Nmrc<f64>, Error>> {
    let n = a.n;
    let mut h = std::ptr::null_mut();
    let pinv = s.pinv.as_ref().map_or(std::ptr::null(), |v| v.as_ptr());
    if unsafe { rsp_symb_from_parts(n as i64, pinv, std::ptr::null(), s.parent.as_ptr(), s.cp.as_ptr(), &mut h) } != 0 {
        return None;
    }
    let lnz = s.cp[n] as usize;
    let mut l = Sprs::<f64>::zeros(n, n, lnz);
    let code = unsafe { rsp_chol_f64(&view(a), h, l.p.as_mut_ptr(), l.i.as_mut_ptr(), l.x.as_mut_ptr()) };
    unsafe { rsp_symb_free(h) };
    status(code).map(|r| r.map(|_| { let mut nm = Nmrc::new(); nm.l = l; nm }))
}
