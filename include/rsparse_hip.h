/*
 * rsparse_hip.h -- C ABI of librsparse_hip.so: an MI355X (gfx950) implementation of the
 * numeric solve path of RLado/rsparse (sparse triangular solves, numeric Cholesky, and the
 * cholsol / lusol drivers).  Symbolic analysis runs on the host CPU, numeric work on the GPU.
 *
 * The reference has no FFI; its boundary is its public Rust functions.  Every entry point
 * below names the reference function (file:line in rsparse v1.2.1) it replaces.  All
 * pointers are HOST pointers unless the name says `_dev`.  Index types are those of the
 * reference on x86-64: `isize` -> int64_t, `usize` -> uint64_t.
 *
 * Rules: nothing throws or aborts across this boundary; every function returns an
 * rsp_status.  RSP_NOT_POSITIVE_DEFINITE / RSP_NO_PIVOT are the reference's two
 * `Error` variants (src/lib.rs:189-194); negative codes have no Rust counterpart
 * (bad input, no usable GPU) and are reported, never papered over with a CPU path.
 */
#ifndef RSPARSE_HIP_H
#define RSPARSE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Borrowed view of rsparse::data::Sprs<f64> (src/data.rs:194-208): CSC, nnz = p[n] (never nzmax),
 * row indices inside a column in any order unless stated. */
typedef struct {
    int64_t nzmax, m, n;
    const int64_t *p;  /* n+1 column pointers (Vec<isize>) */
    const uint64_t *i; /* row indices           (Vec<usize>) */
    const double *x;   /* values                (Vec<f64>)   */
} rsp_csc_f64;

enum rsp_status {
    RSP_OK = 0,
    RSP_NOT_POSITIVE_DEFINITE = 1, /* Error::NotPositiveDefinite, src/lib.rs:191 */
    RSP_NO_PIVOT = 2,              /* Error::NoPivot,             src/lib.rs:193 */
    RSP_BAD_INPUT = -1,            /* the reference would panic on a bounds check */
    RSP_DEVICE_ERROR = -2,         /* HIP error / no gfx950 device visible */
    RSP_INTERNAL_ERROR = -3        /* a defect of the library itself (an unexpected exception), never caused by the input */
};

/* which triangular system a handle solves; names follow the reference functions */
enum rsp_tri_kind {
    RSP_LSOLVE = 0,  /* L x = b,  L lower, diagonal FIRST in each column (src/lib.rs:464) */
    RSP_LTSOLVE = 1, /* L' x = b, same storage                           (src/lib.rs:505) */
    RSP_USOLVE = 2,  /* U x = b,  U upper, diagonal LAST in each column  (src/lib.rs:1230) */
    RSP_UTSOLVE = 3  /* U' x = b, same storage                           (src/lib.rs:1271) */
};

/* layout of a device-resident block of right-hand sides */
enum rsp_layout {
    RSP_COL_MAJOR = 0, /* X[i + r*ld]: each right-hand side contiguous (a Rust `&mut [f64]` per column) */
    RSP_ROW_MAJOR = 1  /* X[i*ld + r]: the nrhs values of one unknown contiguous (the kernels' native layout) */
};

const char *rsp_status_string(int status);
int rsp_device_count(void);     /* number of visible HIP devices (0 if none) */
int rsp_set_device(int device); /* device used by the calling thread's later calls */
/* Devices that the entry points which shard INDEPENDENT work spread it over, in contiguous shares, one host thread per
 * listed device, no exchange between them: the systems of rsp_cholsol_batch_f64, and the right-hand-side COLUMNS of the
 * host-pointer solves rsp_chol_solve_f64 and rsp_tri_solve_f64 (from two columns per device).  For the solves every
 * device but the handle's own works on a replica of the handle's resident solve programmes (and of pinv), copied device
 * to device when first needed -- nothing is analysed twice, nothing visits the host -- and dropped by the next
 * refactorisation.  The first listed device should be the handle's own.  n = 0 restores the default (the calling
 * thread's current device only); a device listed twice gets two shares.  Process-wide. */
int rsp_set_devices(const int *ids, int n);

/* ---- one-shot triangular solves, x overwritten in place -------------------------------------
 * rsparse::lsolve / ltsolve / usolve / utsolve (src/lib.rs:464-471, 505-512, 1230-1237, 1271-1278).
 * Same conventions: lsolve/ltsolve need the diagonal first in every column, usolve/utsolve last. */
int rsp_lsolve_f64(const rsp_csc_f64 *L, double *x);
int rsp_ltsolve_f64(const rsp_csc_f64 *L, double *x);
int rsp_usolve_f64(const rsp_csc_f64 *U, double *x);
int rsp_utsolve_f64(const rsp_csc_f64 *U, double *x);

/* ---- analyse-once / solve-many triangular handle (no reference counterpart: the reference
 *      re-walks the matrix on every call; this caches the device copy and the level schedule) ----
 * Thread safety (all handle types): any number of host threads may call into the library at once, on the same handle
 * too.  Calls on ONE handle are serialised by a mutex in the handle, and a call on another stream than the handle's
 * previous call first waits (on the device) for that call's kernels, because a handle's scratch rows and workspaces
 * are shared by all of its calls.  Different handles are independent.  rsp_tri_solve_dev_noalloc_f64 is the one
 * exception: it records nothing, see there. */
typedef struct rsp_tri rsp_tri;
int rsp_tri_analyze(const rsp_csc_f64 *T, int kind, rsp_tri **out);
/* X = host, column-major n x nrhs with leading dimension ldx >= n; overwritten with the solutions */
int rsp_tri_solve_f64(rsp_tri *h, double *X, int64_t ldx, int64_t nrhs);
/* dX = device pointer; layout/ld as rsp_layout; runs on `stream` (a hipStream_t, NULL = default)
 * and returns without synchronising */
int rsp_tri_solve_dev_f64(rsp_tri *h, double *dX, int layout, int64_t ld, int64_t nrhs, void *stream);
/* No counterpart in the reference.  For callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc): rsp_tri_solve_dev_f64 for a ROW-MAJOR device block, with the
 * programme's scratch rows in a fixed device array of the library (32 MiB per GPU, held until the process ends) instead
 * of a buffer allocated on demand.  Calls no hipMalloc / hipFree / allocating routine.  The handle must have solved
 * once through rsp_tri_solve_dev_f64 or rsp_tri_solve_f64 with more than one right-hand side (that call uploads the
 * programme); returns RSP_BAD_INPUT otherwise, and when scratch rows x nrhs exceeds 4 Mi doubles (rsp_tri_info
 * reports the scratch rows).  Every call on a GPU shares that GPU's array: calls that overlap on one GPU race. */
int rsp_tri_solve_dev_noalloc_f64(rsp_tri *h, double *dX, int64_t ld, int64_t nrhs, void *stream);
/* first then second on the same device block (lsolve + ltsolve of cholsol, lsolve + usolve of lusol), row-major only:
 * wide blocks are cut into two column halves that go through both sweeps on two streams, so the nearly empty levels
 * at the top of one half's dependency tree run beside the other half's heavy levels.  Ordered after earlier work
 * on `stream`; later work on `stream` is ordered after both halves.  Not re-entrant per handle. */
int rsp_tri_solve_pair_dev_f64(rsp_tri *first, rsp_tri *second, double *dX, int64_t ld, int64_t nrhs, void *stream);
/* info[0..10] = n, nnz, dependency levels of single unknowns, tasks, kernel launches per sweep, device bytes of the
 *               programme, algorithmic bytes for one sweep of 1 rhs, unknowns in inverted chains, scratch rows, wide items
 *               (64-row products over the chains' blocks), multiply-adds per right-hand side of the wide items */
int rsp_tri_info(const rsp_tri *h, int64_t *info);
/* integer schedule, for bit-exact comparison with the host oracle: level_of_row[n] */
int rsp_tri_levels(const rsp_tri *h, int64_t *level_of_row);
void rsp_tri_free(rsp_tri *h);
/* Host-only validator of the multi-right-hand-side programme built for T (debug aid; no reference counterpart, needs
 * no GPU and solves nothing for the caller): replays the programme on one synthetic right-hand side with every index
 * bounds-checked and every read checked against the level that produces it.
 * report[0..7] = bad indices, order violations, unknowns not written exactly once, levels, block tasks, clusters,
 *                unknowns inside clusters, kernel launches per sweep;  *max_rel_error = against sequential substitution.
 * task_blocks <= 0 and cluster_slots == 0 select the library defaults; cluster_slots > 0 builds group-record clusters of
 * that many LDS slots (the bottom of the forest).  The programme is checked in both of its views: wide items as their
 * twin 16-row blocks, and as items. */
int rsp_tri_validate_program(const rsp_csc_f64 *T, int kind, int64_t task_blocks, int cluster_slots, int64_t *report,
                             double *max_rel_error);
/* Host-only check of the REFILL of a programme (debug aid, no GPU): the programme of T is built, the values `new_values`
 * (same pattern, nnz entries) are poured into it through its recipe, and the result is compared bit for bit with a
 * programme emitted from scratch for the new values.  report[0] = 1 if the programme carries a recipe (0: duplicate
 * entries or an oversized inverse pool -- such programmes are always emitted again), report[1] = coefficient entries that
 * differ, report[2] = cluster records that differ, report[3] = coefficient entries. */
int rsp_tri_validate_refill(const rsp_csc_f64 *T, const double *new_values, int kind, int64_t task_blocks, int cluster_slots, int64_t *report);
/* Host-only fingerprint of the programme built for T (debug aid, no GPU): FNV-1a hashes of out[0] its pattern (headers,
 * ids, prologue records, task and level tables), out[1] its coefficients, out[2] its clusters and group records,
 * out[3] its refill recipe.  The tests use it to show that the programme does not depend on how many host threads
 * emitted it. */
int rsp_tri_program_checksum(const rsp_csc_f64 *T, int kind, int64_t task_blocks, int cluster_slots, uint64_t *out);


/* Host-only self-test of the threaded helpers the analyses use (host/parallel.hpp; debug aid, no reference counterpart,
 * no GPU): sorts `count` pseudo-random keys (seeded) with the threaded sort and fills an array over threaded ranges, on
 * `threads` host threads (<= 0: the library's default), and compares with the sequential result.
 * Returns RSP_OK when both agree, RSP_BAD_INPUT otherwise. */
int rsp_selftest_host_threads(int64_t count, int threads, uint64_t seed);

/* ---- binary interchange format of Sprs (beside the reference's text format, Sprs::save / Sprs::load, data.rs:414-516) ----
 * 32-byte header (magic "RSPSPRS1", m, n, nnz as little-endian int64) followed by p[n+1], i[nnz], x[nnz] as they sit in
 * memory.  Loading is two calls: the dimensions, then the arrays into caller-allocated storage (validated: monotone p,
 * indices < m).  RSP_BAD_INPUT for a file that cannot be opened, is truncated, or is not in this format. */
int rsp_sprs_save_bin(const rsp_csc_f64 *A, const char *path);
int rsp_sprs_load_bin_dims(const char *path, int64_t *m, int64_t *n, int64_t *nnz);
int rsp_sprs_load_bin(const char *path, int64_t *p, uint64_t *i, double *x);

/* ---- gaxpy ------------------------------------------------------------------------------------
 * rsparse::gaxpy (src/lib.rs:411-421): r = A x + y for one vector (host pointers; r may alias y). */
int rsp_gaxpy_f64(const rsp_csc_f64 *A, const double *x, const double *y, double *r);
/* The same for device-resident BLOCKS of right-hand sides (no reference counterpart: the reference has one vector at
 * a time): A is uploaded once (row-wise), then  Y <- A X + beta Y  for X (n x nrhs) and Y (m x nrhs) in the given
 * rsp_layout, asynchronously on `stream`.  beta = 1 is gaxpy; beta = -1 with Y = B gives the residual A X - B of a
 * solved block; beta = 0 ignores Y's contents.  Entries of a row are added in ascending column order, as the
 * reference's loop adds them. */
typedef struct rsp_mat rsp_mat;
int rsp_mat_create_f64(const rsp_csc_f64 *A, rsp_mat **M);
int rsp_mat_gaxpy_dev_f64(const rsp_mat *M, const double *dX, int64_t ldx, double *dY, int64_t ldy, int64_t nrhs, int layout, double beta,
                          void *stream);
void rsp_mat_free(rsp_mat *M);

/* ---- symbolic analysis (host CPU) -------------------------------------------------------------
 * rsparse::schol (src/lib.rs:968-986) and the LU branch of rsparse::sqr (src/lib.rs:1114-1119,1134-1137).
 * rsp_symb mirrors rsparse::data::Symb (src/data.rs:1022-1038). */
typedef struct rsp_symb rsp_symb;
int rsp_schol(const rsp_csc_f64 *A, int8_t order, rsp_symb **S);
int rsp_sqr_lu(const rsp_csc_f64 *A, int8_t order, rsp_symb **S);
/* sqr(a, order, qr = true) (src/lib.rs:1114-1133): q = amd(order) on A'A, parent = column elimination tree, cp = the n
 * row counts of R, pinv = the 2m + n entries vcount returns (row permutation, n zeros, leftmost), m2, lnz = nnz(V),
 * unz = nnz(R).  A may be rectangular (m >= n for a factorisation; qrsol transposes otherwise). */
int rsp_sqr_qr(const rsp_csc_f64 *A, int8_t order, rsp_symb **S);
/* build a Symb from caller-held vectors (pinv/q may be NULL = None) -- for chol(a, &mut s) with any s.
 * RSP_BAD_INPUT unless pinv and q are permutations of 0..n-1, parent[k] is -1 or < n, cp[0] == 0 and every column of
 * cp holds 1..n-k entries (the reference would panic on a bounds check where these are used as indices). */
int rsp_symb_from_parts(int64_t n, const int64_t *pinv, const int64_t *q, const int64_t *parent, const int64_t *cp,
                        rsp_symb **S);
int64_t rsp_symb_n(const rsp_symb *S);
const int64_t *rsp_symb_pinv(const rsp_symb *S);   /* n entries, or NULL for None */
const int64_t *rsp_symb_q(const rsp_symb *S);      /* n+1 entries as amd returns them, or NULL for None */
const int64_t *rsp_symb_parent(const rsp_symb *S); /* n entries (schol only) */
const int64_t *rsp_symb_cp(const rsp_symb *S);     /* n+1 entries (schol only) */
int64_t rsp_symb_lnz(const rsp_symb *S);
int64_t rsp_symb_unz(const rsp_symb *S);
int64_t rsp_symb_m2(const rsp_symb *S);       /* sqr_qr only: rows of V and R (m plus fictitious rows) */
int64_t rsp_symb_pinv_len(const rsp_symb *S); /* n (schol), 2m + n (sqr_qr), 0 for None */
int64_t rsp_symb_cp_len(const rsp_symb *S);   /* n + 1 (schol), n (sqr_qr), 0 (sqr_lu) */
void rsp_symb_free(rsp_symb *S);
/* the ordering alone: rsparse's private amd (src/lib.rs:1292-1752). Returns RSP_OK and *has_perm = 0
 * for the natural ordering (order < 0), else writes n+1 entries to P. */
int rsp_amd(const rsp_csc_f64 *A, int8_t order, int64_t *P, int *has_perm);

/* ---- numeric Cholesky -------------------------------------------------------------------------
 * rsparse::chol (src/lib.rs:278-337): L has L.p == S.cp, diagonal first then rows ascending.
 * Caller allocates Lp[n+1], Li[cp[n]], Lx[cp[n]] (the fields of Nmrc.l). Pattern on host, values on GPU. */
int rsp_chol_f64(const rsp_csc_f64 *A, const rsp_symb *S, int64_t *Lp, uint64_t *Li, double *Lx);

/* factor kept on the device for repeated solves (no reference counterpart; cholsol refactors each call) */
typedef struct rsp_chol rsp_chol;
int rsp_chol_factor_f64(const rsp_csc_f64 *A, const rsp_symb *S, rsp_chol **F);
/* New values, same pattern (the loop a time stepper or a Newton iteration runs: the reference calls chol(a, &mut s) again,
 * lib.rs:278): numeric factorisation alone, on the analysis F already holds.  A must have the pattern F was created
 * with (only n and nnz are verified).  RSP_NOT_POSITIVE_DEFINITE leaves F without a usable factor until the next
 * successful call.  Solves after it rebuild the solve programmes' values. */
int rsp_chol_refactor_f64(rsp_chol *F, const rsp_csc_f64 *A);
/* Iterative refinement of an already solved block, entirely on the device (no reference counterpart: the
 * reference solves once): `steps` times  W = A X - B (rsp_mat_gaxpy_dev_f64),  W = A^-1 W with the resident factor (the
 * cholsol solve phase, lib.rs:383-386),  X = X - W.  dB, dX, dW: three distinct n x nrhs device blocks of the same
 * layout and leading dimension; dB is read only, dW is scratch.  M: the matrix A that F factors (rsp_mat_create_f64).
 * Asynchronous on `stream`. */
int rsp_chol_refine_dev_f64(rsp_chol *F, const rsp_mat *M, const double *dB, double *dX, double *dW, int layout, int64_t ld, int64_t nrhs,
                            int steps, void *stream);
/* B host column-major n x nrhs: B <- A^-1 B through P, lsolve, ltsolve, P' (src/lib.rs:383-386) */
int rsp_chol_solve_f64(rsp_chol *F, double *B, int64_t ldb, int64_t nrhs);
/* device-resident variant; `permute` = 0 skips P/P' (solves L L' y = b in the permuted ordering) */
int rsp_chol_solve_dev_f64(rsp_chol *F, double *dB, int layout, int64_t ld, int64_t nrhs, int permute, void *stream);
int64_t rsp_chol_lnz(const rsp_chol *F);
int rsp_chol_get_factor(const rsp_chol *F, int64_t *Lp, uint64_t *Li, double *Lx);
/* Zero-copy view of the factor where it lives (Nmrc.l without the round trip through host Vecs): the pattern on the
 * host (Lp: n+1, Li: lnz, exactly the reference's layout), the values ON THE DEVICE (dLx: lnz doubles).  Borrowed: valid
 * until F is freed or refactored. */
int rsp_chol_factor_view(const rsp_chol *F, const int64_t **Lp, const uint64_t **Li, const double **dLx);
/* info[0..9] = n, lnz, supernodes, supernodal levels, flops (sum of squared column counts), algorithmic bytes,
 *              lsolve launches per sweep, lsolve tasks, ltsolve launches per sweep, ltsolve tasks (of the solve
 *              programmes actually built: zeros before the first solve) */
int rsp_chol_info(const rsp_chol *F, int64_t *info);
/* last numeric-factorisation time measured with HIP events on the library's stream, milliseconds */
double rsp_chol_factor_ms(const rsp_chol *F);
void rsp_chol_free(rsp_chol *F);

/* rsparse::cholsol (src/lib.rs:377-389): b overwritten with A\b */
int rsp_cholsol_f64(const rsp_csc_f64 *A, double *b, int8_t order);
/* The analysis cache behind rsp_cholsol_f64 (no counterpart in the reference, whose cholsol, lib.rs:377-389, carries no handle
 * and repeats ordering and symbolic analysis on every call): after a call with a new pattern, a background thread of the
 * library analyses that pattern as a multi-use handle; later calls with the same (device, n, order, pattern of A) --
 * compared entry by entry -- refactorise on it and solve (C2: 1.7 s -> tens of milliseconds).  RSP_ANALYSIS_CACHE = number
 * of patterns kept (default 2, 0 switches the cache off), RSP_ANALYSIS_CACHE_MB = device memory they may hold (default
 * 16384), RSP_ANALYSIS_CACHE_MIN_N = smallest system that is cached (default 4000 unknowns); least recently used entries leave first.  rsp_cache_stats: out[0..6] = hits, misses, entries, estimated device
 * bytes, background analyses in flight, inserted, evicted.  rsp_cache_clear waits for the background analyses and frees
 * every cached handle (call it before the process ends if device memory must be returned in an orderly way).
 * rsp_lusol_f64 shares the cache (and its counters): a call whose factorisation ran on the device (diagonal pivots) leaves
 * that factorisation's device half behind, a later call on the same pattern is rsp_lu_refactor_f64 + a solve; an entry
 * whose new values no longer keep their diagonal pivots is factored by the host routine and leaves the cache. */
int rsp_cache_stats(int64_t *out);
int rsp_cache_clear(void);
/* cholsol over `nsys` systems that share ONE pattern (A->p, A->i; A->x is ignored) and differ in their values:
 * Ax_all[s * nnz .. ) are the values of system s in A's entry order, B[s * n .. ) its right-hand side, overwritten
 * by the solution.  What the reference does by calling cholsol (lib.rs:377-389) in a loop -- BASELINE.json's batched
 * configuration -- with the ordering and the symbolic analysis done once and every kernel launched once for the
 * whole batch.  status (may be NULL): per system RSP_OK or RSP_NOT_POSITIVE_DEFINITE; such a system keeps its
 * right-hand side, as cholsol returns before touching b.  Returns RSP_OK when the batch ran (with status == NULL:
 * RSP_NOT_POSITIVE_DEFINITE if any system failed).  With several devices set by rsp_set_devices the systems are split
 * over them (north_star: independent batched systems are sharded per GPU). */
int rsp_cholsol_batch_f64(const rsp_csc_f64 *A, const double *Ax_all, double *B, int64_t nsys, int8_t order, int32_t *status);

/* ---- LU ---------------------------------------------------------------------------------------
 * rsparse::lu (src/lib.rs:519-622): left-looking LU with threshold partial pivoting.  The values are computed on the GPU
 * when every column keeps its DIAGONAL as pivot under the reference's rule (|diag| >= tol * max|x|, lib.rs:585-589) -- the
 * patterns then follow from integer work on the host, the columns run on the device in one persistent launch (one
 * wavefront per column, columns taken by dependency level, each in the reference's order of operations), every column checks the rule, and the factors are
 * bit-identical to the reference's, values included.  Any column that breaks the rule, a matrix with a structurally missing
 * diagonal or duplicate entries in a column, tol > 1, a column of A itself that already fails the rule, or too little work
 * to pay for the device set-up (< 2e7 multiply-subtracts) send the whole factorisation to the host routine instead (the reference's algorithm, sequential; the
 * same bits).  RSP_LU_GPU=0 forces the host, =1 the GPU whenever the static pivots apply.  rsparse::lusol (src/lib.rs:672-683) then solves with L and U on the GPU. */
typedef struct rsp_lu rsp_lu;
int rsp_lu_f64(const rsp_csc_f64 *A, const rsp_symb *S, double tol, rsp_lu **N);
int rsp_lu_L(const rsp_lu *N, rsp_csc_f64 *L); /* borrowed views, valid until rsp_lu_free */
int rsp_lu_U(const rsp_lu *N, rsp_csc_f64 *U);
const int64_t *rsp_lu_pinv(const rsp_lu *N);   /* n entries */
/* the integer half of the GPU numeric LU alone, on the host (tests; no GPU needed): the patterns of L and U and pinv that
 * rsparse::lu produces when every column pivots on its diagonal, values all zero; *nlevels = depth of the column dependency
 * levels.  RSP_BAD_INPUT when a diagonal entry is structurally missing (no static pivot sequence exists). */
int rsp_lu_static_pattern_f64(const rsp_csc_f64 *A, const rsp_symb *S, rsp_lu **N, int64_t *nlevels);
int rsp_lu_on_device(const rsp_lu *N); /* 1: the values came from the GPU numeric LU, 0: from the host routine */
/* New values on the pattern the handle was made with: N becomes the factorisation of A (same p and i as the matrix given to
 * rsp_lu_f64, RSP_BAD_INPUT otherwise; the column order of that call is kept).  No counterpart in the reference, which calls
 * lu again (lib.rs:519); the result is what that call would give, bit for bit.  When N came from the GPU numeric LU the
 * kernel alone runs again -- no depth-first search, no new analysis, resident solve programmes get the new coefficients --
 * as long as every column still keeps its diagonal as pivot; otherwise the host routine factors A (the pivot sequence may
 * change; solves are analysed again at their next use).  RSP_NO_PIVOT (a singular matrix) leaves the handle WITHOUT a
 * factorisation: solves and views return RSP_NO_PIVOT until a refactorisation succeeds.  Not concurrent with solves on the
 * same handle.  The host views (rsp_lu_L / rsp_lu_U) fetch the new values from the device when they are next asked for. */
int rsp_lu_refactor_f64(rsp_lu *N, const rsp_csc_f64 *A, double tol);
/* b <- A^-1 b for nrhs host column-major right-hand sides (ipvec, lsolve, usolve, ipvec: src/lib.rs:678-681) */
int rsp_lu_solve_f64(rsp_lu *N, const rsp_symb *S, double *B, int64_t ldb, int64_t nrhs);
/* the same for a device-resident block in either rsp_layout, asynchronously on `stream`: both permutations run on the
 * device, nothing is copied to the host and nothing synchronises (S must be the Symb the factorisation was made with) */
int rsp_lu_solve_dev_f64(rsp_lu *N, const rsp_symb *S, double *dB, int layout, int64_t ld, int64_t nrhs, void *stream);
void rsp_lu_free(rsp_lu *N);
int rsp_lusol_f64(const rsp_csc_f64 *A, double *b, int8_t order, double tol);

/* ---- QR: the consumer of usolve / utsolve --------------------------------------------------------
 * rsparse::qr (src/lib.rs:788-877): Householder QR on the host (like lu: sequential, order of operations kept), Nmrc.l = V
 * (diagonal first in every column), Nmrc.u = R (m2 x n, diagonal last), Nmrc.b = beta; borrowed views as for rsp_lu.
 * rsp_qr_solve_f64 is the solve phase of rsparse::qrsol (src/lib.rs:931-955) with the triangular solve on the GPU:
 *   transposed == 0 (the handle factors A, m >= n): b[0, m) in, least-squares solution in b[0, n)
 *       ipvec(pinv), happly k = 0 .. n-1, usolve(R), ipvec(q)                         (lib.rs:936-942)
 *   transposed == 1 (the handle factors A' of an underdetermined A): b[0, rows of A) in, minimum-norm solution in
 *       b[0, columns of A):  pvec(q), utsolve(R), happly k = n-1 .. 0, pvec(pinv)     (lib.rs:949-954)
 * rsp_qrsol_f64 = rsparse::qrsol (src/lib.rs:927-956): b holds max(m, n) entries; no error variants (the reference
 * returns ()), negative codes only. */
typedef struct rsp_qr rsp_qr;
int rsp_qr_f64(const rsp_csc_f64 *A, const rsp_symb *S, rsp_qr **N);
int rsp_qr_V(const rsp_qr *N, rsp_csc_f64 *V);
int rsp_qr_R(const rsp_qr *N, rsp_csc_f64 *R);
const double *rsp_qr_beta(const rsp_qr *N); /* n entries */
int rsp_qr_solve_f64(rsp_qr *N, const rsp_symb *S, double *b, int transposed);
void rsp_qr_free(rsp_qr *N);
int rsp_qrsol_f64(const rsp_csc_f64 *A, double *b, int8_t order);

#ifdef __cplusplus
}
#endif
#endif /* RSPARSE_HIP_H */
