// HIP kernel (gfx950) of the blocked multi-right-hand-side triangular solve (programme: host/blocked.cpp).
//
// One wavefront owns (task, 64 right-hand sides) and executes the task's blocks in order.  A block computes
//     acc[r] = init[r] - sum_e coef[e][r] * src[e]      (+ the block's own small triangular solve)
// for up to 16 unknowns r at once, so every gathered source row is used for all unknowns of the block.
// The kernel is latency-, not bandwidth-bound unless the dependent loads are kept short and deep, hence:
//   * one 32-byte header per block (a single scalar load) instead of one load per field;
//   * source ids are fetched 64 (vector path: 16) at a time with ONE coalesced vector load and handed out with
//     ds_bpermute / v_readlane, so the gathers of a whole batch can be issued back to back;
//   * coefficients are read as coalesced vector loads (MFMA A operands, or broadcast by v_readlane);
//   * 16-unknown blocks run on the matrix cores (v_mfma_f64_16x16x4_f64), which keeps the register
//     footprint small enough for 5 wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../host/sparse.hpp"

namespace rsp {
namespace dev {

#define RSP_NONE INT32_MIN

using BlockHeader = ::rsp::BlockHeader;  // 32 bytes, one s_load_dwordx8 (host/sparse.hpp)

struct BlockArrays {
    const int64_t *task_ptr;     // blocks of task t (tasks sorted by level)
    const BlockHeader *hdr;
    const int32_t *src_idx;      // id >= 0 row of X, id < 0 scratch row ~id, INT32_MIN none
    const double *coef;
};

__device__ __forceinline__ const double *row_ptr(int32_t id, const double *X, int64_t ldx, const double *S, int64_t lds)
{
    return id >= 0 ? X + (int64_t)id * ldx : S + (int64_t)(~id) * lds;
}

__device__ __forceinline__ double readlane_f64(double v, int src_lane)  // src_lane wave-uniform
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

// ---- vector path: RC <= 4 unknowns, one right-hand side per lane ---------------------------------------
template <int RC>
__device__ __forceinline__ void run_block_vec(const BlockHeader h, const BlockArrays &B, double *X, int64_t ldx, double *S, int64_t lds,
                                              int col, bool live, int lane)
{
    const int32_t *ids = B.src_idx + h.src_off;
    const double *cf = B.coef + h.coef_off;
    // dest / init ids: lanes 0..2RC-1 load them at once
    const int32_t hid = ids[lane < 2 * RC ? lane : 0];
    double acc[RC];
#pragma unroll
    for (int r = 0; r < RC; ++r) {
        const int32_t id = __builtin_amdgcn_readlane(hid, RC + r);
        acc[r] = (id != RSP_NONE && live) ? row_ptr(id, X, ldx, S, lds)[col] : 0.0;
    }
    const int32_t *sidx = ids + 2 * RC;
    constexpr int SB = 16;  // sources per batch: 16 ids and 16*RC <= 64 coefficients, one vector load each
    for (int e0 = 0; e0 < h.nsrc; e0 += SB) {
        const int nb = min(SB, h.nsrc - e0);
        const int32_t idv = sidx[e0 + (lane < nb ? lane : 0)];
        const double cv = cf[(int64_t)e0 * RC + (lane < nb * RC ? lane : 0)];
        double x[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            if (u < nb) {  // wave-uniform
                const int32_t id = __builtin_amdgcn_readlane(idv, u);
                x[u] = live ? row_ptr(id, X, ldx, S, lds)[col] : 0.0;
            } else {
                x[u] = 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            if (u < nb) {
#pragma unroll
                for (int r = 0; r < RC; ++r) acc[r] -= readlane_f64(cv, u * RC + r) * x[u];
            }
        }
    }
    if (h.has_tri) {
        const double tv = cf[(int64_t)h.nsrc * RC + (lane < RC * RC ? lane : 0)];
#pragma unroll
        for (int r = 0; r < RC; ++r) {
#pragma unroll
            for (int t = 0; t < r; ++t) acc[r] -= readlane_f64(tv, r * RC + t) * acc[t];
            acc[r] /= readlane_f64(tv, r * RC + r);
        }
    }
#pragma unroll
    for (int r = 0; r < RC; ++r) {
        const int32_t id = __builtin_amdgcn_readlane(hid, r);
        if (id != RSP_NONE && live) const_cast<double *>(row_ptr(id, X, ldx, S, lds))[col] = acc[r];
    }
}

typedef double f64x4 __attribute__((ext_vector_type(4)));

// ---- matrix-core path: 16 unknowns x 64 right-hand sides ----------------------------------------------------
// acc lives in the C/D layout of v_mfma_f64_16x16x4_f64: lane (g = lane >> 4, j = lane & 15), register s <-> unknown
// 4*s + g, columns cw + 4*j + nt for the 4 tiles nt (every lane owns 4 CONSECUTIVE doubles of a row: 32-byte accesses).
// One K-step consumes 4 sources: A = coefficients (lane (g, i): coef[source 4k+g][unknown i] -- a fully coalesced
// 512-byte read of the coefficient stream), B = the gathered rows (lane (g, j): 4 doubles of source 4k+g).
// The in-block triangular solve is 4 more K-steps with A = inverse of the block's 16 x 16 triangle and B = acc itself
// (C/D register s IS the B operand of K-step s), so nothing ever crosses lanes.
__device__ __forceinline__ void run_block_mfma(const BlockHeader h, const BlockArrays &B, double *X, int64_t ldx, double *S, int64_t lds,
                                               int cw, int nrhs, int lane)
{
    const int g = lane >> 4, j = lane & 15;
    const int cb = cw + 4 * j;
    const bool full = cb + 4 <= nrhs;
    const int32_t *ids = B.src_idx + h.src_off;
    const double *cf = B.coef + h.coef_off;
    const int32_t hid = ids[lane < 32 ? lane : 0];  // 16 dest ids, 16 init ids
    f64x4 acc[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int32_t id = __shfl(hid, 16 + 4 * s + g, 64);
        double v[4] = {0.0, 0.0, 0.0, 0.0};
        if (id != RSP_NONE) {
            const double *p = row_ptr(id, X, ldx, S, lds) + cb;
            if (full) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) v[nt] = p[nt];
            } else {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    if (cb + nt < nrhs) v[nt] = p[nt];
            }
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt][s] = v[nt];
    }
    const int32_t *sidx = ids + 32;
    for (int e0 = 0; e0 < h.nsrc; e0 += 64) {
        const int nb = min(64, h.nsrc - e0);  // multiple of 16
        const int32_t idv = sidx[e0 + (lane < nb ? lane : 0)];
        for (int k0 = 0; k0 < nb; k0 += 16) {  // 4 K-steps = 16 sources: all gathers first, then 16 MFMAs
            double a[4], xb[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int32_t id = __shfl(idv, k0 + 4 * q + g, 64);
                a[q] = -cf[(int64_t)(e0 + k0 + 4 * q) * 16 + lane];
                const double *p = row_ptr(id, X, ldx, S, lds) + cb;
                if (full) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) xb[q][nt] = p[nt];
                } else {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) xb[q][nt] = (cb + nt < nrhs) ? p[nt] : 0.0;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], xb[q][nt], acc[nt], 0, 0, 0);
        }
    }
    if (h.has_tri) {
        const double *tinv = cf + (int64_t)h.nsrc * 16;  // 4 x 64 doubles in A-operand order
        double a4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) a4[k] = tinv[k * 64 + lane];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            f64x4 r = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < 4; ++k) r = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[k], acc[nt][k], r, 0, 0, 0);
            acc[nt] = r;
        }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int32_t id = __shfl(hid, 4 * s + g, 64);
        if (id != RSP_NONE) {
            double *p = const_cast<double *>(row_ptr(id, X, ldx, S, lds)) + cb;
            if (full) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) p[nt] = acc[nt][s];
            } else {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    if (cb + nt < nrhs) p[nt] = acc[nt][s];
            }
        }
    }
}

// one wavefront per (task, 64 columns); the blocks of a task run in order
__global__ __launch_bounds__(256, 4) void solve_level_blocked(BlockArrays B, int64_t task_begin, int ntasks, double *X, int64_t ldx,
                                                               double *S, int64_t lds, int nrhs, int nchunks)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const int t = wave / nchunks;
    if (t >= ntasks) return;
    const int chunk = wave - t * nchunks;
    const int cw = chunk * 64;
    const int col = cw + lane;
    const bool live = col < nrhs;
    const int64_t bend = B.task_ptr[task_begin + t + 1];
    for (int64_t b = B.task_ptr[task_begin + t]; b < bend; ++b) {
        const BlockHeader h = B.hdr[b];
        switch (h.rc) {
            case 16: run_block_mfma(h, B, X, ldx, S, lds, cw, nrhs, lane); break;
            case 1: run_block_vec<1>(h, B, X, ldx, S, lds, col, live, lane); break;
            case 2: run_block_vec<2>(h, B, X, ldx, S, lds, col, live, lane); break;
            default: run_block_vec<4>(h, B, X, ldx, S, lds, col, live, lane); break;
        }
    }
}

}  // namespace dev
}  // namespace rsp
