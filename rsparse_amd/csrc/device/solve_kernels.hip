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
    const int64_t src = scatter ? i : perm[i], dst = scatter ? perm[i] : i;
    for (int r = threadIdx.x & 63; r < k; r += 64) out[dst * ld + r] = in[src * ld + r];
}

}  // namespace dev
}  // namespace rsp

// ==== blocked multi-right-hand-side programme (host/blocked.cpp) ===========================================
namespace rsp {
namespace dev {

struct BlockArrays {
    const int64_t *task_ptr;  // blocks of task t (tasks sorted by level)
    const int32_t *rc;        // per block: class size 1/2/4/8/16
    const int32_t *has_tri;
    const int32_t *nsrc;
    const int64_t *src_off, *coef_off, *row_off;
    const int32_t *rows;      // rc dest ids then rc init ids; id >= 0 row of X, id < 0 scratch row ~id, INT32_MIN none
    const int32_t *src_idx;
    const double *coef;
};

#define RSP_NONE INT32_MIN

template <int V>
__device__ __forceinline__ const double *row_ptr(int32_t id, const double *X, int64_t ldx, const double *S, int64_t lds)
{
    return id >= 0 ? X + (int64_t)id * ldx : S + (int64_t)(~id) * lds;
}

// One block: RC unknowns x V right-hand sides per lane, all in registers.
template <int RC, int V>
__device__ __forceinline__ void run_block(const BlockArrays &B, int64_t b, double *X, int64_t ldx, double *S, int64_t lds, int col0)
{
    const int32_t *rows = B.rows + B.row_off[b];
    const int ns = B.nsrc[b];
    const int32_t *sidx = B.src_idx + B.src_off[b];
    const double *cf = B.coef + B.coef_off[b];
    double acc[RC][V];
#pragma unroll
    for (int r = 0; r < RC; ++r) {
        const int32_t id = rows[RC + r];
        if (id != RSP_NONE) {
            const double *p = row_ptr<V>(id, X, ldx, S, lds) + col0;
#pragma unroll
            for (int v = 0; v < V; ++v) acc[r][v] = p[v];
        } else {
#pragma unroll
            for (int v = 0; v < V; ++v) acc[r][v] = 0.0;
        }
    }
    // U source rows in flight per wavefront: all ids (scalar), then all row loads, then the multiply-adds
    constexpr int U = (RC * V >= 32) ? 4 : 8;
    int e = 0;
    for (; e + U <= ns; e += U) {
        double x[U][V];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double *p = row_ptr<V>(sidx[e + u], X, ldx, S, lds) + col0;
#pragma unroll
            for (int v = 0; v < V; ++v) x[u][v] = p[v];
        }
        const double *c0 = cf + (int64_t)e * RC;
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int r = 0; r < RC; ++r) {
                const double a = c0[u * RC + r];
#pragma unroll
                for (int v = 0; v < V; ++v) acc[r][v] -= a * x[u][v];
            }
        }
    }
    for (; e < ns; ++e) {
        const double *p0 = row_ptr<V>(sidx[e], X, ldx, S, lds) + col0;
        const double *c0 = cf + (int64_t)e * RC;
        double x0[V];
#pragma unroll
        for (int v = 0; v < V; ++v) x0[v] = p0[v];
#pragma unroll
        for (int r = 0; r < RC; ++r) {
            const double a0 = c0[r];
#pragma unroll
            for (int v = 0; v < V; ++v) acc[r][v] -= a0 * x0[v];
        }
    }
    if (B.has_tri[b]) {
        const double *tri = cf + (int64_t)ns * RC;
#pragma unroll
        for (int r = 0; r < RC; ++r) {
#pragma unroll
            for (int t = 0; t < r; ++t) {
                const double a = tri[r * RC + t];
#pragma unroll
                for (int v = 0; v < V; ++v) acc[r][v] -= a * acc[t][v];
            }
            const double d = tri[r * RC + r];
#pragma unroll
            for (int v = 0; v < V; ++v) acc[r][v] /= d;
        }
    }
#pragma unroll
    for (int r = 0; r < RC; ++r) {
        const int32_t id = rows[r];
        if (id != RSP_NONE) {
            double *p = const_cast<double *>(row_ptr<V>(id, X, ldx, S, lds)) + col0;
#pragma unroll
            for (int v = 0; v < V; ++v) p[v] = acc[r][v];
        }
    }
}

typedef double f64x4 __attribute__((ext_vector_type(4)));

// 16-unknown block on the matrix cores.  acc (16 unknowns x 64*V right-hand sides) lives in the C/D layout of
// v_mfma_f64_16x16x4_f64: lane (g = lane >> 4, j = lane & 15), register s  <->  unknown 4*s + g, columns
// col0 + j*NT + nt (NT = 4*V tiles, so every lane owns NT CONSECUTIVE doubles of a row: wide loads/stores).
// One K-step consumes 4 sources: A = coefficients (lane (g, i): coef[source 4k+g][unknown i], a fully coalesced
// 512-byte read of the coefficient stream), B = the 4 gathered source rows (lane (g, j): NT doubles of source
// 4k+g).  No scalar loads and no cross-lane traffic; the in-block triangular solve is 4 more K-steps with
// A = inverse of the block's 16 x 16 triangle and B = acc itself (C/D register s IS the B operand of K-step s).
template <int V>
__device__ __forceinline__ void run_block_mfma(const BlockArrays &B, int64_t b, double *X, int64_t ldx, double *S, int64_t lds,
                                               int col0, int nrhs, int lane)
{
    constexpr int NT = 4 * V;
    const int g = lane >> 4, j = lane & 15;
    const int cb = col0 + j * NT;  // first column of this lane
    const bool full = cb + NT <= nrhs;
    const int32_t *rows = B.rows + B.row_off[b];
    const int ns = B.nsrc[b];  // padded to a multiple of 4 by the host
    const int32_t *sidx = B.src_idx + B.src_off[b];
    const double *cf = B.coef + B.coef_off[b];
    f64x4 acc[NT];
    // init: register s <- row (4s + g)
#pragma unroll
    for (int sidx4 = 0; sidx4 < 4; ++sidx4) {
        const int32_t id = rows[16 + 4 * sidx4 + g];
        double v[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) v[nt] = 0.0;
        if (id != RSP_NONE) {
            const double *p = row_ptr<V>(id, X, ldx, S, lds) + cb;
            if (full) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) v[nt] = p[nt];
            } else {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    if (cb + nt < nrhs) v[nt] = p[nt];
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt][sidx4] = v[nt];
    }
#pragma unroll 2
    for (int e = 0; e < ns; e += 4) {
        const int32_t id = sidx[e + g];
        const double a = -cf[(int64_t)e * 16 + lane];
        const double *p = row_ptr<V>(id, X, ldx, S, lds) + cb;
        double xb[NT];
        if (full) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) xb[nt] = p[nt];
        } else {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) xb[nt] = (cb + nt < nrhs) ? p[nt] : 0.0;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xb[nt], acc[nt], 0, 0, 0);
    }
    if (B.has_tri[b]) {
        const double *tinv = cf + (int64_t)ns * 16;  // 4 x 64 doubles in A-operand order
        double a4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) a4[k] = tinv[k * 64 + lane];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f64x4 r = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < 4; ++k) r = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[k], acc[nt][k], r, 0, 0, 0);
            acc[nt] = r;
        }
    }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const int32_t id = rows[4 * s4 + g];
        if (id != RSP_NONE) {
            double *p = const_cast<double *>(row_ptr<V>(id, X, ldx, S, lds)) + cb;
            if (full) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) p[nt] = acc[nt][s4];
            } else {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    if (cb + nt < nrhs) p[nt] = acc[nt][s4];
            }
        }
    }
}

// one wavefront per (task, block of 64*V columns); blocks of a task run in order
template <int V>
__global__ __launch_bounds__(256, 4) void solve_level_blocked(BlockArrays B, int64_t task_begin, int ntasks, double *X, int64_t ldx,
                                                            double *S, int64_t lds, int nrhs, int nchunks)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const int t = wave / nchunks;
    if (t >= ntasks) return;
    const int chunk = wave - t * nchunks;
    const int cw = chunk * 64 * V;          // first column of this wavefront
    const int col0 = cw + lane * V;         // first column of this lane in the vector (rc <= 4) path
    const int64_t bend = B.task_ptr[task_begin + t + 1];
    for (int64_t b = B.task_ptr[task_begin + t]; b < bend; ++b) {
        const int rc = B.rc[b];
        if (rc == 16) {
            run_block_mfma<V>(B, b, X, ldx, S, lds, cw, nrhs, lane);  // every lane takes part (EXEC all ones)
        } else if (col0 < nrhs) {
            switch (rc) {
                case 1: run_block<1, V>(B, b, X, ldx, S, lds, col0); break;
                case 2: run_block<2, V>(B, b, X, ldx, S, lds, col0); break;
                default: run_block<4, V>(B, b, X, ldx, S, lds, col0); break;
            }
        }
    }
}

}  // namespace dev
}  // namespace rsp
