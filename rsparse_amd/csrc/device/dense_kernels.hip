// HIP kernel (gfx950) of the DENSE TASKS of the multi-right-hand-side triangular solve (programme: host/blocked.cpp,
// layout: BlockProgram in host/sparse.hpp).
//
// A task is a run of consecutive unknowns -- whole subtrees at the bottom of the dependency forest, a few hundred
// unknowns -- whose rows of X (its own unknowns plus the external rows they read) fit in LDS.  One wavefront owns
// (task, 16 * NT right-hand sides) and is a workgroup of its own, so nothing is shared and the only barrier is the
// wavefront's own.  The wavefront
//   1. copies the task's rows of X into LDS slots (three dependent round trips: header, row ids, rows; every row piece
//      is 128 * NT contiguous bytes and all of them are in flight together),
//   2. runs the task's stream of matrix-core K-STEPS out of LDS: the host has cut the unknowns into nodes of up to 16
//      and folded each node's own triangle into its coefficients ( x_node = [T^-1 | -T^-1 Cf] * [b_node ; x_sources] ),
//      so a node is a chain of v_mfma_f64_16x16x4_f64 whose B operand is four LDS rows per step and whose A operand
//      (64 doubles per step, in lane order) streams from global memory several steps ahead of its use; nothing in a
//      task waits on global memory once its rows are in LDS,
//   3. writes its own unknowns back.
// Algorithmic traffic only: every row of X is read once and written once per sweep; the coefficient stream is read once
// per 16 * NT columns and is shared by the column chunks of a task through the XCD's L2 (consecutive work items).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../host/sparse.hpp"

namespace rsp {
namespace dev {

struct DenseArrays {
    const ClusterHeader *hdr;  // sorted by level; ngroups = k-steps, grp_off = first k-step
    const int32_t *ids;        // row of X behind every slot: external rows first, then the task's unknowns
    const uint16_t *slots;     // 4 per k-step
    const int32_t *meta;       // 1 per k-step
    const double *coef;        // 64 per k-step
};

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

template <int NT>
struct DenseQuad {  // four k-steps of operands
    double a[4];
    int slot[4];
    int meta[4];
};

template <int NT>
__device__ __forceinline__ void dense_load_quad(DenseQuad<NT> &Q, const DenseArrays &D, int64_t k0, int ks, int nks, int lane)
{
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t k = k0 + min(ks + u, nks - 1);  // clamped: every load is unconditional
        Q.a[u] = D.coef[k * 64 + lane];
        Q.slot[u] = D.slots[k * 4 + (lane >> 4)];
        Q.meta[u] = D.meta[k];
    }
}

// FAST: whole chunks of 16 * NT columns, rows 16-byte aligned (ld even, X aligned).
template <int NT, bool FAST>
__global__ __launch_bounds__(64) void solve_dense_tasks(DenseArrays D, int64_t task_begin, int ntasks, int nchunks, double *__restrict__ X,
                                                         int64_t ldx, int nrhs)
{
    extern __shared__ double xs[];  // (slots + 1) rows of RB doubles
    constexpr int RB = 16 * NT;     // doubles per LDS row
    constexpr int LPR = RB / 2;     // lanes per row (16 bytes each)
    constexpr int RPI = 64 / LPR;   // rows per wave-wide load
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

    // the first operands do not depend on the rows: issue them before anything else
    DenseQuad<NT> Q, Qn;
    dense_load_quad<NT>(Q, D, k0, 0, nks, lane);

    // ---- rows of X -> LDS ------------------------------------------------------------------------------------
    const int sub = lane % LPR, rsel = lane / LPR;
    for (int s0 = 0; s0 < nslots; s0 += 8 * RPI) {
        int32_t id[8];
        d2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) id[u] = ids[min(s0 + u * RPI + rsel, nslots - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double *p = X + (int64_t)id[u] * ldx + c0 + 2 * sub;
            if (FAST) {
                v[u] = *reinterpret_cast<const d2 *>(p);
            } else {
                v[u].x = c0 + 2 * sub < nrhs ? p[0] : 0.0;
                v[u].y = c0 + 2 * sub + 1 < nrhs ? p[1] : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int s = s0 + u * RPI + rsel;
            if (s < nslots) *reinterpret_cast<d2 *>(xs + (size_t)s * RB + 2 * sub) = v[u];
        }
    }
    if (lane < LPR) *reinterpret_cast<d2 *>(xs + (size_t)nslots * RB + 2 * lane) = d2{0.0, 0.0};  // the row of zeros
    __syncthreads();

    // ---- the k-step stream -----------------------------------------------------------------------------------
    const int g = lane >> 4, j = lane & 15;
    d4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
    for (int ks = 0; ks < nks; ks += 4) {
        dense_load_quad<NT>(Qn, D, k0, ks + 4, nks, lane);  // next quad in flight during this one
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (ks + u < nks) {  // wave-uniform
                const double *row = xs + (size_t)Q.slot[u] * RB + j;
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(Q.a[u], row[16 * t], acc[t], 0, 0, 0);
                const int m = __builtin_amdgcn_readfirstlane(Q.meta[u]);
                if (m < 0) {  // last k-step of a node: rows 4 s + g of the result go to slots out + 4 s + g
                    const int w = (m >> 20) & 31, out = m & 0xFFFFF;
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
                    __syncthreads();  // one wavefront: orders the stores before the next node's reads
                }
            }
        }
        Q = Qn;
    }

    // ---- the task's unknowns -> X -----------------------------------------------------------------------------
    for (int s0 = h.n_ext; s0 < nslots; s0 += 8 * RPI) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int s = s0 + u * RPI + rsel;
            if (s < nslots) {
                const int32_t id = ids[s];
                const d2 v = *reinterpret_cast<const d2 *>(xs + (size_t)s * RB + 2 * sub);
                double *p = X + (int64_t)id * ldx + c0 + 2 * sub;
                if (FAST) {
                    *reinterpret_cast<d2 *>(p) = v;
                } else {
                    if (c0 + 2 * sub < nrhs) p[0] = v.x;
                    if (c0 + 2 * sub + 1 < nrhs) p[1] = v.y;
                }
            }
        }
    }
}

}  // namespace dev
}  // namespace rsp
