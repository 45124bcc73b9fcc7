// HIP kernels (gfx950) for the level-scheduled sparse triangular solves.
//
// Data model (built on the host by schedule.cpp, uploaded once per factor):
//   op   : one unknown.  x[row] = (x[row] - sum_p val[p] * x[idx[p]]) / diag, p in [ptr[op], ptr[op+1])
//   task : a run of ops executed sequentially by ONE wavefront (ops are stored in execution order)
//   level: a set of tasks with no dependencies among them -> one kernel launch
// Right-hand sides are ROW-MAJOR on the device (the nrhs values of one unknown are contiguous), so a
// wavefront reads/writes 64*V consecutive doubles per unknown: every global access is fully coalesced and
// no lane ever needs another lane's data (each lane solves its own V right-hand sides).
// All per-entry metadata (idx, val, row, diag) is wave-uniform and comes through the scalar cache.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rsp {
namespace dev {

struct TriArrays {
    const int32_t *task_ptr;  // ntasks+1 : op range of each task (tasks sorted by level)
    const int32_t *op_row;    // nops
    const int32_t *op_ptr;    // nops+1   : entry range of each op
    const double *op_diag;    // nops
    const int32_t *ent_idx;   // nnz
    const double *ent_val;    // nnz
};

// ---- many right-hand sides: one wavefront per (task, block of 64*V columns) -------------------
template <int V>
__global__ __launch_bounds__(256) void solve_level_multi(TriArrays T, int task_begin, int ntasks, double *__restrict__ X,
                                                          int64_t ld, int nrhs, int nchunks)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const int t = wave / nchunks;
    if (t >= ntasks) return;
    const int chunk = wave - t * nchunks;
    const int col0 = (chunk * 64 + lane) * V;
    if (col0 >= nrhs) return;
    // nrhs is a multiple of V (the launcher picks V accordingly), so a lane is either fully in or fully out
    double *Xc = X + col0;
    const int op_end = T.task_ptr[task_begin + t + 1];
    for (int op = T.task_ptr[task_begin + t]; op < op_end; ++op) {
        const int64_t row = T.op_row[op];
        const int pb = T.op_ptr[op], pe = T.op_ptr[op + 1];
        double acc[V];
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] = Xc[row * ld + v];
        int p = pb;
        for (; p + 4 <= pe; p += 4) {  // 4 independent gathers in flight per wave
            const int64_t j0 = T.ent_idx[p], j1 = T.ent_idx[p + 1], j2 = T.ent_idx[p + 2], j3 = T.ent_idx[p + 3];
            const double a0 = T.ent_val[p], a1 = T.ent_val[p + 1], a2 = T.ent_val[p + 2], a3 = T.ent_val[p + 3];
            double x0[V], x1[V], x2[V], x3[V];
#pragma unroll
            for (int v = 0; v < V; ++v) {
                x0[v] = Xc[j0 * ld + v];
                x1[v] = Xc[j1 * ld + v];
                x2[v] = Xc[j2 * ld + v];
                x3[v] = Xc[j3 * ld + v];
            }
#pragma unroll
            for (int v = 0; v < V; ++v) {
                acc[v] -= a0 * x0[v];
                acc[v] -= a1 * x1[v];
                acc[v] -= a2 * x2[v];
                acc[v] -= a3 * x3[v];
            }
        }
        for (; p < pe; ++p) {
            const int64_t j = T.ent_idx[p];
            const double a = T.ent_val[p];
#pragma unroll
            for (int v = 0; v < V; ++v) acc[v] -= a * Xc[j * ld + v];
        }
        const double d = T.op_diag[op];
#pragma unroll
        for (int v = 0; v < V; ++v) Xc[row * ld + v] = acc[v] / d;
    }
}

// ---- one right-hand side: one wavefront per task, lanes stride over the entries of an unknown and
//      the partial sums are combined with a DPP/shuffle wave reduction --------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ __launch_bounds__(256) void solve_level_single(TriArrays T, int task_begin, int ntasks, double *x)
{
    const int lane = threadIdx.x & 63;
    const int t = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    if (t >= ntasks) return;
    const int op_end = T.task_ptr[task_begin + t + 1];
    for (int op = T.task_ptr[task_begin + t]; op < op_end; ++op) {
        const int row = T.op_row[op];
        const int pb = T.op_ptr[op], pe = T.op_ptr[op + 1];
        double s = 0.0;
        for (int p = pb + lane; p < pe; p += 64) s += T.ent_val[p] * x[T.ent_idx[p]];
        if (pe - pb > 1) s = wave_sum(s);  // wave-uniform branch
        else s = __shfl(s, 0, 64);
        if (lane == 0) x[row] = (x[row] - s) / T.op_diag[op];
        // later ops of this task may read x[row] from other lanes: same wavefront, same CU, in-order
        // vector memory pipeline and a write-through L1 shared by the CU, so only the compiler must not reorder
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
}

// ---- one vector, one use: rows with flags instead of levels --------------------------------------------------------------
// A handle that solves ONE vector and is dropped (the reference's one-shot lsolve ... cholsol, lusol) cannot pay for a
// programme: building the blocked one costs ~20 ns per entry on the host, and the plain level programme is one launch per
// dependency level (13 500 launches, 0.3 s, at a 1024^2 Laplacian).  This kernel needs the gather form and the rows sorted by
// dependency level: ONE persistent launch, wavefronts take "grabs" -- up to kRowsPerGrab rows of ONE level, listed by the
// host -- from a counter; a row whose source x[j] is still in flight waits on that row's flag.  (By level, because in a
// postordered factor row r mostly depends on row r - 1: rows in index order in flight form one chain.  Several rows per
// grab, because same-address atomics serialise at ~0.2 us each: a million of them would be the kernel's duration.  The
// WHOLE wavefront polls ONE flag at a time -- the first unready one -- because 64 lanes polling 64 different flags from
// thousands of waiting wavefronts saturate the fabric and slow every hand-over down tenfold.)  A row only waits for rows
// of lower levels, which come earlier in the order and were taken by wavefronts that are running: no deadlock; every wait
// is bounded besides (a flag that never rises is a defect: the wait gives up and marks the run, every other wait sees the
// mark, the grid drains, the host reports RSP_DEVICE_ERROR).  x and the flags travel as agent-scope atomics.  Sums are
// per lane in entry order, then a fixed butterfly: deterministic, independent of timing.
constexpr int kRowsPerGrab = 8;

// values of a gather form out of the matrix it was built from, where a factorisation left them on the device: entry e
// comes from Tx[src[e]], the diagonal of row r from the first (lower storage) or last entry of column r
__global__ __launch_bounds__(256) void rows_gather_values(const int32_t *__restrict__ src, const double *__restrict__ Tx, double *__restrict__ val,
                                                          int64_t nnz, const int64_t *__restrict__ Tp, double *__restrict__ diag, int64_t n,
                                                          int lower_storage)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz || e < n; e += stride) {
        if (e < nnz) val[e] = Tx[src[e]];
        if (e < n) diag[e] = Tx[lower_storage ? Tp[e] : Tp[e + 1] - 1];
    }
}

struct RowsControl {
    int next;    // next grab
    int failed;  // a wait gave up
    int stuck_row, stuck_source;
};

__global__ __launch_bounds__(64) void solve_rows_flagged(const int64_t *ptr, const int32_t *idx, const double *val, const double *diag, double *x,
                                                         int *done, RowsControl *ctl, const int32_t *order, const int2 *grabs, int ngrabs,
                                                         unsigned spin_limit, int reverse)
{
    const int lane = threadIdx.x;
    for (;;) {
        // (every lane adds, 1 or 0, and readfirstlane states the uniformity: see lu_columns for what `if (lane == 0)` did here)
        int g = atomicAdd(&ctl->next, lane == 0 ? 1 : 0);
        g = __builtin_amdgcn_readfirstlane(g);
        if (g >= ngrabs) break;  // every wavefront gets here: the counter only grows
        const int2 span = grabs[g];
        for (int t = span.x; t < span.y; ++t) {
            const int r = __builtin_amdgcn_readfirstlane(order[t]);
            const double xr = x[r], dr = diag[r];  // (issued ahead of the waits below)
            double s = 0.0;
            // 64 entries at a time; reverse (the descending solves, whose rows list their NEAREST source first): from the
            // last piece to the first, so that the source that was computed a moment ago -- the one worth waiting for --
            // is met when everything else has been added up.  (Walking a 1 000-entry row of a top supernode front to
            // back waited at the first piece and then paid every other piece's round trips on the critical path:
            // ltsolve at C2 93 ms, lsolve, whose nearest source comes last anyway, 18 ms.)
            const int64_t row_begin = ptr[r], row_end = ptr[r + 1];
            const int64_t pieces = (row_end - row_begin + 63) >> 6;
            for (int64_t c = 0; c < pieces; ++c) {
                const int64_t p0 = row_begin + ((reverse ? pieces - 1 - c : c) << 6);
                const bool in = p0 + lane < row_end;
                const int j = in ? idx[p0 + lane] : -1;
                const double v = in ? val[p0 + lane] : 0.0;
                bool ready = j < 0 || __hip_atomic_load(done + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                unsigned long long waiting = __ballot(!ready);
                unsigned spins = 0;
                while (waiting) {  // (uniform) the whole wavefront polls the first unready source's flag
                    const int first = __builtin_ctzll(waiting);
                    const int jw = __builtin_amdgcn_readlane(j, first);
                    while (__hip_atomic_load(done + jw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                        __builtin_amdgcn_s_sleep(4);
                        if ((++spins & 1023u) == 0) {
                            if (spins >= spin_limit && lane == 0 && atomicExch(&ctl->failed, 1) == 0) {
                                ctl->stuck_row = r;
                                ctl->stuck_source = jw;
                            }
                            if (__hip_atomic_load(&ctl->failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                        }
                    }
                    if (__hip_atomic_load(&ctl->failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                    ready = j < 0 || __hip_atomic_load(done + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                    waiting = __ballot(!ready);
                }
                if (in) s += v * __hip_atomic_load(x + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            s = wave_sum(s);
            if (lane == 0) __hip_atomic_store(x + r, (xr - s) / dr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): x[r] is in the L2 before its flag is
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (lane == 0) __hip_atomic_store(done + r, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- layout / permutation helpers ----------------------------------------------------------------
// out (row-major n x k, ld_out) = in (column-major n x k, ld_in), with optional row permutation:
//   scatter != 0 : out[perm[i], :] = in[i, :]   (ipvec, lib.rs:2151)
//   scatter == 0 : out[i, :] = in[perm[i], :]   (pvec,  lib.rs:2244)
// Tiled through LDS so that both the reads (along i) and the writes (along r) are coalesced.
__global__ __launch_bounds__(256) void colmajor_to_rowmajor(const double *__restrict__ in, int64_t ld_in, double *__restrict__ out,
                                                            int64_t ld_out, int64_t n, int k, const int64_t *__restrict__ perm,
                                                            int scatter)
{
    __shared__ double tile[32][33];
    const int64_t i0 = (int64_t)blockIdx.x * 32;
    const int r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int rr = ty; rr < 32; rr += 8) {
        const int64_t i = i0 + tx;
        const int r = r0 + rr;
        if (i < n && r < k) {
            const int64_t src = (perm && !scatter) ? perm[i] : i;
            tile[rr][tx] = in[src + (int64_t)r * ld_in];
        }
    }
    __syncthreads();
    for (int ii = ty; ii < 32; ii += 8) {
        const int64_t i = i0 + ii;
        const int r = r0 + tx;
        if (i < n && r < k) {
            const int64_t dst = (perm && scatter) ? perm[i] : i;
            out[dst * ld_out + r] = tile[tx][ii];
        }
    }
}

// out (column-major) = in (row-major), same permutation options (applied to the row index of `out`/`in`)
__global__ __launch_bounds__(256) void rowmajor_to_colmajor(const double *__restrict__ in, int64_t ld_in, double *__restrict__ out,
                                                            int64_t ld_out, int64_t n, int k, const int64_t *__restrict__ perm,
                                                            int scatter)
{
    __shared__ double tile[32][33];
    const int64_t i0 = (int64_t)blockIdx.x * 32;
    const int r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int ii = ty; ii < 32; ii += 8) {
        const int64_t i = i0 + ii;
        const int r = r0 + tx;
        if (i < n && r < k) {
            const int64_t src = (perm && !scatter) ? perm[i] : i;
            tile[ii][tx] = in[src * ld_in + r];
        }
    }
    __syncthreads();
    for (int rr = ty; rr < 32; rr += 8) {
        const int64_t i = i0 + tx;
        const int r = r0 + rr;
        if (i < n && r < k) {
            const int64_t dst = (perm && scatter) ? perm[i] : i;
            out[dst + (int64_t)r * ld_out] = tile[tx][rr];
        }
    }
}

// row-major -> row-major row permutation (device-resident RHS blocks): out[perm[i],:] = in[i,:] or out[i,:] = in[perm[i],:]
__global__ __launch_bounds__(256) void permute_rows(const double *__restrict__ in, double *__restrict__ out, int64_t ld, int64_t n, int k,
                                                    const int64_t *__restrict__ perm, int scatter)
{
    const int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n) return;
    const int64_t p = perm ? perm[i] : i;  // no permutation = identity
    const int64_t src = scatter ? i : p, dst = scatter ? p : i;
    for (int r = threadIdx.x & 63; r < k; r += 64) out[dst * ld + r] = in[src * ld + r];
}

// one vector (ipvec / pvec, lib.rs:2151 / 2244): out[perm[i]] = in[i] or out[i] = in[perm[i]]; perm may be null
__global__ __launch_bounds__(256) void permute_vector(const double *__restrict__ in, double *__restrict__ out, int64_t n,
                                                      const int64_t *__restrict__ perm, int scatter)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t p = perm ? perm[i] : i;
    if (scatter) out[p] = in[i];
    else out[i] = in[p];
}

}  // namespace dev
}  // namespace rsp

