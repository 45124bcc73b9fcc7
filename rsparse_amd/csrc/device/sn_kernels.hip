// HIP kernel (gfx950) of the product ITEMS of the supernodal multi-right-hand-side programme (host/supernodal.cpp,
// SnProgram in host/sparse.hpp):  OUT = INIT + A * IN  over rows named by ids (row of X, or scratch row).
//
// One workgroup of four wavefronts owns (item, 64 right-hand sides); an item has at most 64 output rows, so a
// wavefront's accumulators are a 64 x 64 tile of doubles in the C/D layout of v_mfma_f64_16x16x4_f64 (4 row tiles x
// 4 column tiles; lane (g = lane >> 4, j = lane & 15), register s <-> row 16 rt + 4 s + g, columns cw + 4 j + nt: every
// lane owns 4 CONSECUTIVE doubles of a row, 32-byte accesses).  One K-step consumes 4 input rows: A = up to four
// 512-byte coefficient tiles (coalesced, lane order), B = the gathered row pieces (2 x 16 bytes per lane), 4 x m_tiles
// MFMAs for each of the 4 column tiles.  With four row tiles a K-step is 4 KiB of loads for 16 MFMAs -- a quarter of the
// bytes per flop of a 16-row tile.  The loads run three K-steps ahead of the arithmetic (ids six ahead), all of them
// unconditional (clamped step numbers), so the compiler's vmcnt bookkeeping stays exact.
// Long input lists are shared by `nsplit` wavefronts of the workgroup (contiguous ranges of K-steps); their partial
// tiles meet in LDS one row tile at a time and wavefront 0 stores.  Every wavefront of a workgroup reaches every barrier.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../host/sparse.hpp"

namespace rsp {
namespace dev {

struct SnArrays {
    const SnItem *items;  // sorted by level
    const int32_t *ids;
    const double *coef;
};

typedef double sd4 __attribute__((ext_vector_type(4)));
typedef double sd2 __attribute__((ext_vector_type(2)));

struct SnStep {
    double a[4];
    sd2 lo, hi;
};

struct SnCtx {
    const int32_t *in_ids;
    const double *cf;
    const double *X, *S;
    int64_t ldx, lds;
    int mt, g, lane, cb, nrhs, last;
    bool in;
};

__device__ __forceinline__ const double *sn_row(int32_t id, const double *X, int64_t ldx, const double *S, int64_t lds)
{
    return id >= 0 ? X + (int64_t)id * ldx : S + (int64_t)(~id) * lds;
}

__device__ __forceinline__ int32_t sn_id(const SnCtx &c, int ks) { return c.in_ids[4 * min(ks, c.last) + c.g]; }

template <bool FAST>
__device__ __forceinline__ void sn_load(SnStep &R, const SnCtx &c, int ks, int32_t id)
{
    const int k = min(ks, c.last);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) R.a[rt] = c.cf[((int64_t)k * c.mt + min(rt, c.mt - 1)) * 64 + c.lane];
    const double *p = sn_row(id, c.X, c.ldx, c.S, c.lds);
    if (FAST) {
        p += c.in ? c.cb : 0;
        R.lo = *reinterpret_cast<const sd2 *>(p);
        R.hi = *reinterpret_cast<const sd2 *>(p + 2);
    } else {
        p += c.cb;
        R.lo.x = c.cb + 0 < c.nrhs ? p[0] : 0.0;
        R.lo.y = c.cb + 1 < c.nrhs ? p[1] : 0.0;
        R.hi.x = c.cb + 2 < c.nrhs ? p[2] : 0.0;
        R.hi.y = c.cb + 3 < c.nrhs ? p[3] : 0.0;
    }
}

__device__ __forceinline__ void sn_mfma(sd4 (&acc)[4][4], const SnStep &R, int mt)
{
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        if (rt < mt) {  // wave-uniform
            acc[rt][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(R.a[rt], R.lo.x, acc[rt][0], 0, 0, 0);
            acc[rt][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(R.a[rt], R.lo.y, acc[rt][1], 0, 0, 0);
            acc[rt][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(R.a[rt], R.hi.x, acc[rt][2], 0, 0, 0);
            acc[rt][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(R.a[rt], R.hi.y, acc[rt][3], 0, 0, 0);
        }
    }
}

// One level: items [item_begin, ...) sorted by nsplit: n4 items shared by four wavefronts, then n2 shared by two, then n1
// run by one.  A workgroup hosts one, two or four (item, 64-column chunk) pairs accordingly, so all four wavefronts work.
struct SnLevel {
    int64_t item_begin;
    int n4, n2, n1;
};

// (group of 8 items, chunk, item) order of pair q over n items
__device__ __forceinline__ void sn_pair(int q, int n, int nchunks, int &item, int &chunk)
{
    const int grp = q / (8 * nchunks), r = q - grp * (8 * nchunks);
    const int in_grp = min(8, n - grp * 8);
    chunk = r / in_grp;
    item = grp * 8 + (r - chunk * in_grp);
}

template <bool FAST>
__global__ __launch_bounds__(256, 2) void sn_gemm(SnArrays P, SnLevel L, int nchunks, double *__restrict__ X, int64_t ldx,
                                                   double *__restrict__ S, int64_t lds, int nrhs)
{
    __shared__ double red[3 * 1024];  // partial tiles of the non-leading wavefronts, one row tile at a time
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // XCD-aware placement (blockIdx % 8 share an XCD, speed only): every XCD walks a contiguous range of work, ordered
    // (group of 8 items, column chunk, item): the workgroups an XCD runs together are consecutive items -- row groups of
    // one node, which gather the same input rows -- on the same columns
    const int per_xcd = (int)(gridDim.x >> 3);
    const int work = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    const int w4 = L.n4 * nchunks, w2 = (L.n2 * nchunks + 1) / 2, w1 = (L.n1 * nchunks + 3) / 4;
    if (work >= w4 + w2 + w1) return;  // whole workgroup
    int ns, wsub, item, chunk;
    bool valid = true;
    if (work < w4) {
        ns = 4;
        wsub = wv;
        sn_pair(work, L.n4, nchunks, item, chunk);
    } else if (work < w4 + w2) {
        ns = 2;
        wsub = wv & 1;
        const int q = (work - w4) * 2 + (wv >> 1);
        valid = q < L.n2 * nchunks;
        sn_pair(valid ? q : 0, L.n2, nchunks, item, chunk);
        item += L.n4;
    } else {
        ns = 1;
        wsub = 0;
        const int q = (work - w4 - w2) * 4 + wv;
        valid = q < L.n1 * nchunks;
        sn_pair(valid ? q : 0, L.n1, nchunks, item, chunk);
        item += L.n4 + L.n2;
    }
    const SnItem I = P.items[L.item_begin + item];
    const int mt = I.m_tiles;
    const bool active = valid;
    const int ks0 = active ? (I.k_steps * wsub) / ns : 0, ks1 = active ? (I.k_steps * (wsub + 1)) / ns : 0;

    SnCtx c;
    c.in_ids = P.ids + I.in_off;
    c.cf = P.coef + I.a_off;
    c.X = X;
    c.S = S;
    c.ldx = ldx;
    c.lds = lds;
    c.mt = mt;
    c.g = lane >> 4;
    c.lane = lane;
    c.cb = chunk * 64 + 4 * (lane & 15);
    c.nrhs = nrhs;
    c.last = max(ks1 - 1, 0);
    c.in = c.cb + 4 <= nrhs;

    sd4 acc[4][4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[rt][nt] = sd4{0.0, 0.0, 0.0, 0.0};

    if (ks0 < ks1) {  // wave-uniform
        SnStep A, B, Cc;
        int32_t n0 = sn_id(c, ks0 + 3), n1 = sn_id(c, ks0 + 4), n2 = sn_id(c, ks0 + 5);
        {
            const int32_t i0 = sn_id(c, ks0), i1 = sn_id(c, ks0 + 1), i2 = sn_id(c, ks0 + 2);
            sn_load<FAST>(A, c, ks0, i0);
            sn_load<FAST>(B, c, ks0 + 1, i1);
            sn_load<FAST>(Cc, c, ks0 + 2, i2);
        }
        for (int ks = ks0; ks < ks1; ks += 3) {
            const int32_t m0 = sn_id(c, ks + 6), m1 = sn_id(c, ks + 7), m2 = sn_id(c, ks + 8);
            sn_mfma(acc, A, mt);
            sn_load<FAST>(A, c, ks + 3, n0);
            if (ks + 1 < ks1) sn_mfma(acc, B, mt);
            sn_load<FAST>(B, c, ks + 4, n1);
            if (ks + 2 < ks1) sn_mfma(acc, Cc, mt);
            sn_load<FAST>(Cc, c, ks + 5, n2);
            n0 = m0;
            n1 = m1;
            n2 = m2;
        }
    }

    // ---- partial tiles of the non-leading wavefronts meet in LDS, one row tile at a time -------------------------------------
    // (ns is the same for the whole workgroup; every wavefront reaches every barrier)
    if (ns > 1) {
        for (int rt = 0; rt < 4; ++rt) {
            if (active && wsub > 0 && rt < mt) {
                double *mine = red + (size_t)(wv - 1) * 1024 + lane;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q == rt) {
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                            for (int s = 0; s < 4; ++s) mine[(nt * 4 + s) * 64] = acc[q][nt][s];
                    }
                }
            }
            __syncthreads();
            if (active && wsub == 0 && rt < mt) {
                for (int o = 0; o < ns - 1; ++o) {
                    const double *theirs = red + (size_t)(wv + o) * 1024 + lane;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (q == rt) {
#pragma unroll
                            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                                for (int s = 0; s < 4; ++s) acc[q][nt][s] += theirs[(nt * 4 + s) * 64];
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
    if (!active || wsub != 0) return;

    // ---- INIT rows are added at the end (they are only read here, after every input row), then the store ---------------------
    const int32_t *out_ids = P.ids + I.out_off, *init_ids = P.ids + I.init_off;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        if (rt < mt) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int row = 16 * rt + 4 * s + c.g;
                const int32_t oid = out_ids[row];
                if (oid == SnProgram::kNone) continue;
                double v[4] = {acc[rt][0][s], acc[rt][1][s], acc[rt][2][s], acc[rt][3][s]};
                if (I.has_init) {
                    const int32_t iid = init_ids[row];
                    if (iid != SnProgram::kNone) {
                        const double *p = sn_row(iid, X, ldx, S, lds) + c.cb;
                        if (FAST) {
                            if (c.in) {
                                const sd2 lo = *reinterpret_cast<const sd2 *>(p), hi = *reinterpret_cast<const sd2 *>(p + 2);
                                v[0] += lo.x; v[1] += lo.y; v[2] += hi.x; v[3] += hi.y;
                            }
                        } else {
#pragma unroll
                            for (int nt = 0; nt < 4; ++nt)
                                if (c.cb + nt < nrhs) v[nt] += p[nt];
                        }
                    }
                }
                double *p = const_cast<double *>(sn_row(oid, X, ldx, S, lds)) + c.cb;
                if (FAST) {
                    if (c.in) {
                        *reinterpret_cast<sd2 *>(p) = sd2{v[0], v[1]};
                        *reinterpret_cast<sd2 *>(p + 2) = sd2{v[2], v[3]};
                    }
                } else {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        if (c.cb + nt < nrhs) p[nt] = v[nt];
                }
            }
        }
    }
}

}  // namespace dev
}  // namespace rsp
