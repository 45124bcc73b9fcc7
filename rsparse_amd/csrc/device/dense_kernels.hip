// HIP kernel (gfx950) of the DENSE TASKS of the multi-right-hand-side triangular solve (programmes: host/blocked.cpp
// with dense clusters, host/supernodal.cpp; layout: BlockProgram / SnProgram in host/sparse.hpp).
//
// A task is a run of consecutive unknowns -- whole subtrees at the bottom of the dependency forest, a few hundred
// unknowns -- whose rows of X (its own unknowns plus the external rows they read) fit in LDS.  One wavefront owns
// (task, 16 * NT right-hand sides) and is a workgroup of its own, so nothing is shared and the only barrier is the
// wavefront's own.  The wavefront
//   1. copies the task's rows of X into LDS slots (three dependent round trips: header, row ids, rows; every row piece
//      is 128 * NT contiguous bytes and all of them are in flight together); slots that stand for scratch rows
//      (id < 0: partial sums the task sends upwards) start at zero,
//   2. runs the task's stream of matrix-core K-STEPS out of LDS: the host has cut the unknowns into nodes of up to 16
//      and folded each node's own triangle into its coefficients ( x_node = [T^-1 | -T^-1 Cf] * [b_node ; x_sources] ),
//      so a node is a chain of v_mfma_f64_16x16x4_f64 whose B operand is four LDS rows per step and whose A operand
//      (64 doubles per step, in lane order) streams from global memory through a ring of registers 32 steps ahead of
//      its use; nothing in a task waits on global memory once its rows are in LDS,
//   3. writes its own slots back (rows of X, or scratch rows).
// Algorithmic traffic only: every row of X is read once and written once per sweep; the coefficient stream is read once
// per 16 * NT columns and is shared by the column chunks of a task through the XCD's L2 (consecutive work items).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../host/sparse.hpp"

namespace rsp {
namespace dev {

struct DenseArrays {
    const ClusterHeader *hdr;  // sorted by level; ngroups = k-steps, grp_off = first k-step
    const int32_t *ids;        // row behind every slot (external rows first, then the task's own): >= 0 row of X, < 0 scratch row ~id
    const uint32_t *steps;     // 4 per k-step: bits 0-15 slot of B-operand row g; bits 16-20: rows of the node on its LAST k-step, else 0
    const double *coef;        // 64 per k-step
};

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

struct DenseQuad {  // four k-steps of operands
    double a[4];
    uint32_t step[4];
};

__device__ __forceinline__ void dense_load_quad(DenseQuad &Q, const DenseArrays &D, int64_t k0, int ks, int nks, int lane)
{
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t k = k0 + min(ks + u, nks - 1);  // clamped: every load is unconditional
        Q.a[u] = D.coef[k * 64 + lane];
        Q.step[u] = D.steps[k * 4 + (lane >> 4)];
    }
}

// FAST: whole chunks of 16 * NT columns, rows 16-byte aligned (ld even, X and S aligned).
template <int NT, bool FAST>
__global__ __launch_bounds__(64) void solve_dense_tasks(DenseArrays D, int64_t task_begin, int ntasks, int nchunks, double *__restrict__ X,
                                                         int64_t ldx, double *__restrict__ S, int64_t lds, int nrhs)
{
    extern __shared__ double xs[];  // (slots + 1) rows of RB doubles
    constexpr int RB = 16 * NT;     // doubles per LDS row
    constexpr int LPR = RB / 2;     // lanes per row (16 bytes each)
    constexpr int RPI = 64 / LPR;   // rows per wave-wide load
    constexpr int DEPTH = 8;        // quads of k-steps in flight
    const int lane = threadIdx.x;
    // XCD-aware placement: blockIdx % 8 share an XCD; every XCD walks a contiguous range of items, and the column
    // chunks of one task are consecutive items: they read the same coefficient stream out of that XCD's L2 and
    // touch the same rows of X at about the same time (DRAM page locality)
    const int per_xcd = (int)(gridDim.x >> 3);
    const int item = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    const int c = item / nchunks;
    if (c >= ntasks) return;
    const int chunk = item - c * nchunks;
    const int c0 = chunk * RB;
    const ClusterHeader h = D.hdr[task_begin + c];
    const int32_t *ids = D.ids + h.ids_off;
    const int nslots = h.n_ext + h.n_own;
    const int nks = h.ngroups;
    const int64_t k0 = h.grp_off;

    // the operands do not depend on the rows: fill the ring before anything else
    DenseQuad R[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) dense_load_quad(R[d], D, k0, 4 * d, nks, lane);

    // ---- rows -> LDS ---------------------------------------------------------------------------------------------
    // Two dependent round trips for ALL rows: the ids of every slot (one per lane and group of RPI slots, kept in
    // registers for the write-back), then every row piece at once.  FAST: the pieces go straight to LDS
    // (global_load_lds_dwordx4: wave-uniform LDS base + 16 bytes per lane, which is exactly the slot layout), no
    // registers are held while they travel; scratch slots (id < 0) are not loaded but cleared.
    const int sub = lane % LPR, rsel = lane / LPR;
    constexpr int MAXG = 40;  // groups of RPI slots per pass (40 KiB of LDS per pass)
    int32_t id[MAXG];
#pragma unroll
    for (int u = 0; u < MAXG; ++u) id[u] = ids[min(u * RPI + rsel, nslots - 1)];
    if (FAST && nslots <= MAXG * RPI) {
#pragma unroll
        for (int u = 0; u < MAXG; ++u) {
            if (u * RPI < nslots) {  // wave-uniform
                const int s = u * RPI + rsel;
                if (s < nslots && id[u] >= 0)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(X + (int64_t)id[u] * ldx + c0 + 2 * sub),
                                                     (__attribute__((address_space(3))) void *)(xs + (size_t)(u * RPI) * RB), 16, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < MAXG; ++u) {
            const int s = u * RPI + rsel;
            if (u * RPI < nslots && s < nslots && id[u] < 0) *reinterpret_cast<d2 *>(xs + (size_t)s * RB + 2 * sub) = d2{0.0, 0.0};
        }
    } else {
        for (int s0 = 0; s0 < nslots; s0 += 8 * RPI) {
            int32_t idr[8];
            d2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) idr[u] = ids[min(s0 + u * RPI + rsel, nslots - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v[u] = d2{0.0, 0.0};
                if (idr[u] >= 0) {  // scratch slots start at zero
                    const double *p = X + (int64_t)idr[u] * ldx + c0 + 2 * sub;
                    if (FAST) {
                        v[u] = *reinterpret_cast<const d2 *>(p);
                    } else {
                        if (c0 + 2 * sub < nrhs) v[u].x = p[0];
                        if (c0 + 2 * sub + 1 < nrhs) v[u].y = p[1];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s = s0 + u * RPI + rsel;
                if (s < nslots) *reinterpret_cast<d2 *>(xs + (size_t)s * RB + 2 * sub) = v[u];
            }
        }
    }
    if (lane < LPR) *reinterpret_cast<d2 *>(xs + (size_t)nslots * RB + 2 * lane) = d2{0.0, 0.0};  // the row of zeros
    __syncthreads();  // (also waits for the rows in flight)

    // ---- the k-step stream -----------------------------------------------------------------------------------
    const int g = lane >> 4, j = lane & 15;
    d4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
    int out = h.n_ext;  // nodes write consecutive slots
    for (int base = 0; base < nks; base += 4 * DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ks = base + 4 * d + u;
                if (ks < nks) {  // wave-uniform
                    const uint32_t st = R[d].step[u];
                    const double *row = xs + (size_t)(st & 0xFFFFu) * RB + j;
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(R[d].a[u], row[16 * t], acc[t], 0, 0, 0);
                    const int w = (int)(__builtin_amdgcn_readfirstlane(st) >> 16) & 31;
                    if (w) {  // last k-step of a node: rows 4 s + g of the result go to slots out + 4 s + g
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            const int r = 4 * s + g;
                            if (r < w) {
#pragma unroll
                                for (int t = 0; t < NT; ++t) xs[(size_t)(out + r) * RB + 16 * t + j] = acc[t][s];
                            }
                        }
#pragma unroll
                        for (int t = 0; t < NT; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
                        out += w;
                        __syncthreads();  // one wavefront: orders the stores before the next node's reads
                    }
                }
            }
            dense_load_quad(R[d], D, k0, base + 4 * (DEPTH + d), nks, lane);  // the ring slot's next contents
        }
    }

    // ---- the task's own slots -> X / scratch --------------------------------------------------------------------
    if (nslots <= MAXG * RPI) {
#pragma unroll
        for (int u = 0; u < MAXG; ++u) {
            const int s = u * RPI + rsel;
            if (u * RPI < nslots && s >= h.n_ext && s < nslots) {
                const d2 v = *reinterpret_cast<const d2 *>(xs + (size_t)s * RB + 2 * sub);
                double *p = (id[u] >= 0 ? X + (int64_t)id[u] * ldx : S + (int64_t)(~id[u]) * lds) + c0 + 2 * sub;
                if (FAST) {
                    *reinterpret_cast<d2 *>(p) = v;
                } else {
                    if (c0 + 2 * sub < nrhs) p[0] = v.x;
                    if (c0 + 2 * sub + 1 < nrhs) p[1] = v.y;
                }
            }
        }
    } else {
        for (int s = h.n_ext + rsel; s < nslots; s += RPI) {
            const int32_t ido = ids[s];
            const d2 v = *reinterpret_cast<const d2 *>(xs + (size_t)s * RB + 2 * sub);
            double *p = (ido >= 0 ? X + (int64_t)ido * ldx : S + (int64_t)(~ido) * lds) + c0 + 2 * sub;
            if (FAST) {
                *reinterpret_cast<d2 *>(p) = v;
            } else {
                if (c0 + 2 * sub < nrhs) p[0] = v.x;
                if (c0 + 2 * sub + 1 < nrhs) p[1] = v.y;
            }
        }
    }
}

}  // namespace dev
}  // namespace rsp
