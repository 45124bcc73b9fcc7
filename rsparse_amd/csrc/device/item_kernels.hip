// HIP kernel (gfx950) of the WIDE ITEMS of the blocked multi-right-hand-side programme (host/blocked.cpp, WideItem in
// host/sparse.hpp): up to 64 unknowns of one chain phase (4 row tiles of 16) over the union of the tiles' source lists,
//     OUT[r] = INIT[r] - sum_e A[r][e] * IN[e]          r < 16 m_tiles, e < 4 k_steps
// over rows named by ids (row of X, or scratch row ~id).  The same arithmetic also exists as ordinary 16-row blocks (the
// item's twins), which is how the latency-bound levels at the top of the tree and narrow blocks run it; this kernel is
// for the levels that fill the chip.
//
// One wavefront owns (item, 64 right-hand sides): its accumulators are a 64 x 64 tile of doubles in the C/D layout of
// v_mfma_f64_16x16x4_f64 (4 row tiles x 4 column tiles), so one K-step (4 sources) is up to 16 MFMAs -- 1024 cycles of
// matrix-core time -- for 4 KiB of loads, a quarter of the bytes per flop of a 16-row block and four times the arithmetic
// behind every dependent round trip.  A 4-bit mask per K-step names the row tiles that have entries in it: only those
// 16 x 4 coefficient blocks are stored (K-step by K-step) and multiplied, so the union list costs loads, not MFMAs.
// The four wavefronts of a workgroup take four column chunks of the SAME item (or share its K-steps, KS): they read the
// same coefficient blocks at about the same time, so three of the four find them in the compute unit's L1.
// Column tiles are interleaved so that every global access is 16 bytes per lane and contiguous across lanes: lane
// (g = lane >> 4, j = lane & 15) holds columns cw + 2 j, cw + 2 j + 1 (tiles 0, 1) and cw + 32 + 2 j, cw + 33 + 2 j
// (tiles 2, 3) of source row 4 k + g.
// Loads run D K-steps ahead of the arithmetic in a ring of compile-time registers; every load is unconditional (K-steps
// past the end re-read the last one; a K-step always loads four blocks, the ones past its own belong to the next K-step)
// so that the hardware's load counter stays exact (blocked_kernels.hip).  Source ids arrive 64 at a time (16 K-steps)
// with one coalesced load, two windows ahead, and reach their lanes by ds_bpermute; masks and block counts are scalars.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../host/sparse.hpp"

namespace rsp {
namespace dev {

struct ItemArrays {
    const WideItem *items;  // sorted by level
    const int32_t *ids;     // BlockProgram::src_idx
    const double *coef;     // BlockProgram::coef
    const uint32_t *masks;  // BlockProgram::item_masks
};

typedef double wd4 __attribute__((ext_vector_type(4)));
typedef double wd2 __attribute__((ext_vector_type(2)));

struct ItemStep {
    double a[4];  // the K-step's coefficient blocks, in storage order (then whatever follows)
    wd2 lo, hi;
};

struct ItemCtx {
    const double *cf;      // this lane's word of the item's first block
    const char *Xl, *Sl;   // this lane's first piece of row 0 of X / of the scratch rows
    uint32_t ldx8, lds8;   // row pitch in bytes
    int g, last;
};

// K-step k (clamped to the last one) of the wavefront's range: blocks [blk, blk + 4) and the piece of source row 4 k + g
__device__ __forceinline__ void item_load(ItemStep &R, const ItemCtx &c, int k, int blk, int32_t idw)
{
    const int kk = min(k, c.last);
    const int32_t id = __builtin_amdgcn_ds_bpermute(((4 * kk + c.g) & 63) << 2, idw);
    const double *a = c.cf + (int64_t)blk * 64;
#pragma unroll
    for (int q = 0; q < 4; ++q) R.a[q] = a[q * 64];
    const bool x = id >= 0;
    const uint32_t row = (uint32_t)(x ? id : ~id);
    const char *p = (x ? c.Xl : c.Sl) + (uint64_t)row * (uint64_t)(x ? c.ldx8 : c.lds8);
    R.lo = *reinterpret_cast<const wd2 *>(p);
    R.hi = *reinterpret_cast<const wd2 *>(p + 256);
}

// bits: the K-step's mask (wave-uniform); tile rt's block is the popcount(bits below rt)-th of the K-step
__device__ __forceinline__ void item_mfma(wd4 (&acc)[4][4], const ItemStep &R, unsigned bits)
{
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        if ((bits >> rt) & 1u) {  // wave-uniform
            const int idx = __builtin_popcount(bits & ((1u << rt) - 1u));
            double a = R.a[0];
            if (rt >= 1) a = idx == 1 ? R.a[1] : a;
            if (rt >= 2) a = idx == 2 ? R.a[2] : a;
            if (rt >= 3) a = idx == 3 ? R.a[3] : a;
            acc[rt][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, R.lo.x, acc[rt][0], 0, 0, 0);
            acc[rt][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, R.lo.y, acc[rt][1], 0, 0, 0);
            acc[rt][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, R.hi.x, acc[rt][2], 0, 0, 0);
            acc[rt][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, R.hi.y, acc[rt][3], 0, 0, 0);
        }
    }
}

__device__ __forceinline__ uint64_t window_masks(const uint32_t *mw, int w, int nwin)  // (wave-uniform: scalar loads)
{
    const uint32_t *p = mw + 4 * min(w, nwin - 1);
    const uint64_t m = (uint64_t)p[0] | ((uint64_t)p[1] << 32);
    return w < nwin ? m : 0;
}

// nrhs: a multiple of 64 (whole chunks), rows 16-byte aligned.  gridDim.x: a multiple of 8.
// KS: wavefronts that share one (item, chunk) pair's K-steps (1, 2 or 4; whole windows of 16 K-steps each): levels with
// few, long items are split so that they still fill the chip; the partial tiles meet in LDS one row tile at a time and
// the pair's first wavefront stores.  The workgroup's four wavefronts are 4 / KS column chunks of one item (KS = 1 and
// fewer than four chunks: of several items).
template <int D, int MINB, int KS>
__global__ __launch_bounds__(256, MINB) void solve_items(ItemArrays P, int64_t item_begin, int nitems, int nchunks, double *__restrict__ X, int64_t ldx,
                                                          double *__restrict__ S, int64_t lds)
{
    static_assert(16 % D == 0, "ring slots must keep their place from one id window to the next");
    __shared__ double red[KS > 1 ? 3 * 1024 : 1];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // XCD-aware placement (blockIdx % 8 share an XCD): every XCD walks a contiguous range of items -- row groups of one
    // chain, which gather the same rows
    const int per_xcd = (int)(gridDim.x >> 3);
    const int work = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    const int cpw = min(nchunks, 4 / KS);          // column chunks of one item in a workgroup
    const int ipw = 4 / (KS * cpw);                // items per workgroup
    const int groups = (nchunks + cpw - 1) / cpw;  // workgroups per item (or per ipw items)
    const int slot = work / groups, cg = work - slot * groups;
    const int pair = wv / KS, sub = wv - pair * KS;  // (item, chunk) pair inside the workgroup, share of its K-steps
    const int item = slot * ipw + pair / cpw, chunk = cg * cpw + pair % cpw;
    const bool active = item < nitems && chunk < nchunks;
    if (KS == 1 && !active) return;  // (no barrier in that variant)
    const WideItem I = P.items[item_begin + (active ? item : 0)];
    const int mt = I.m_tiles;
    // this wavefront's windows of 16 K-steps: [w0, w1), K-steps [ks0, ks0 + ks)
    const int nwin_all = (I.k_steps + 15) >> 4;
    const int w0 = (nwin_all * sub) / KS, w1 = (nwin_all * (sub + 1)) / KS;
    const int ks0 = 16 * w0, ks = active ? min(I.k_steps, 16 * w1) - ks0 : 0;
    const int nwin = w1 - w0;
    const int32_t *ids = P.ids + I.ids_off;
    const int32_t *in_ids = ids + 32 * mt + 4 * ks0;
    const uint32_t *mw = P.masks + I.mask_off + 4 * w0;

    ItemCtx c;
    c.g = lane >> 4;
    c.last = max(ks - 1, 0);
    c.cf = P.coef + I.coef_off + lane;
    const int cb = (active ? chunk : 0) * 64 + 2 * (lane & 15);
    c.Xl = reinterpret_cast<const char *>(X + cb);
    c.Sl = reinterpret_cast<const char *>(S + cb);
    c.ldx8 = (uint32_t)(ldx * 8);
    c.lds8 = (uint32_t)(lds * 8);

    const int32_t out_id = ids[min(lane, 16 * mt - 1)], init_id = ids[16 * mt + min(lane, 16 * mt - 1)];
    wd4 acc[4][4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[rt][nt] = wd4{0.0, 0.0, 0.0, 0.0};

    if (ks > 0) {  // wave-uniform
        const int nid = 4 * ks;
        int32_t idw = in_ids[min(lane, nid - 1)];        // window 0
        int32_t idn = in_ids[min(64 + lane, nid - 1)];   // window 1
        uint64_t mcur = window_masks(mw, 0, nwin), mnext = window_masks(mw, 1, nwin);
        int blk = (int)mw[2];  // first block of the next K-step to be loaded
        int blk_last = blk;    // ... of the last K-step that exists (K-steps past the end re-read it)
        ItemStep ring[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {  // (D <= 16: inside window 0)
            const bool real = d <= c.last;
            blk_last = real ? blk : blk_last;
            item_load(ring[d], c, d, blk_last, idw);
            blk += real ? __builtin_popcount((unsigned)(mcur >> (4 * d)) & 15u) : 0;
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int k0 = 0; k0 < ks; k0 += 16) {
            // K-steps multiplied in this pass: window k0 / 16 (mcur); loaded: [k0 + D, k0 + 16 + D), i.e. that window up to
            // j = 16 - D, then the next one (mnext)
            const int32_t idc = idw, idx = idn;
            idw = idn;
            idn = in_ids[min(2 * 64 + 4 * k0 + lane, nid - 1)];
            const uint64_t mc = mcur, mx = mnext;
            mcur = mnext;
            mnext = window_masks(mw, (k0 >> 4) + 2, nwin);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (k0 + j < ks) item_mfma(acc, ring[j % D], (unsigned)(mc >> (4 * j)) & 15u);  // wave-uniform
                const int kl = k0 + j + D;
                const bool real = kl <= c.last;
                blk_last = real ? blk : blk_last;
                item_load(ring[j % D], c, kl, blk_last, j + D < 16 ? idc : idx);
                blk += real ? __builtin_popcount((unsigned)((j + D < 16 ? mc : mx) >> (4 * ((j + D) & 15))) & 15u) : 0;
                __builtin_amdgcn_sched_barrier(0);  // keep the re-fill right behind the slot's last use
            }
        }
    }

    if (KS > 1) {  // every wavefront of the workgroup reaches every barrier
        for (int rt = 0; rt < 4; ++rt) {
            if (sub > 0 && rt < mt) {
                double *mine = red + (size_t)(wv - pair - 1) * 1024 + lane;  // non-leading wavefronts: 4 - 4 / KS of them
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (q == rt) {
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                            for (int s = 0; s < 4; ++s) mine[(nt * 4 + s) * 64] = acc[q][nt][s];
                    }
            }
            __syncthreads();
            if (sub == 0 && rt < mt) {
                for (int o = 0; o < KS - 1; ++o) {
                    const double *theirs = red + (size_t)(wv - pair + o) * 1024 + lane;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (q == rt) {
#pragma unroll
                            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                                for (int s = 0; s < 4; ++s) acc[q][nt][s] += theirs[(nt * 4 + s) * 64];
                        }
                }
            }
            __syncthreads();
        }
        if (!active || sub != 0) return;
    }

    // ---- OUT = INIT - acc: the initial rows are read here, after every input row ----------------------------------------
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        if (rt < mt) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int row = 16 * rt + 4 * s + c.g;
                const int32_t oid = __builtin_amdgcn_ds_bpermute(row << 2, out_id);
                const int32_t iid = __builtin_amdgcn_ds_bpermute(row << 2, init_id);
                if (oid == BlockProgram::kNone) continue;
                wd2 lo = {0.0, 0.0}, hi = {0.0, 0.0};
                if (iid != BlockProgram::kNone) {
                    const char *p = (iid >= 0 ? c.Xl : c.Sl) + (uint64_t)(uint32_t)(iid >= 0 ? iid : ~iid) * (uint64_t)(iid >= 0 ? c.ldx8 : c.lds8);
                    lo = *reinterpret_cast<const wd2 *>(p);
                    hi = *reinterpret_cast<const wd2 *>(p + 256);
                }
                char *q = const_cast<char *>(oid >= 0 ? c.Xl : c.Sl) + (uint64_t)(uint32_t)(oid >= 0 ? oid : ~oid) * (uint64_t)(oid >= 0 ? c.ldx8 : c.lds8);
                *reinterpret_cast<wd2 *>(q) = wd2{lo.x - acc[rt][0][s], lo.y - acc[rt][1][s]};
                *reinterpret_cast<wd2 *>(q + 256) = wd2{hi.x - acc[rt][2][s], hi.y - acc[rt][3][s]};
            }
        }
    }
}

// ksplit: 1, 2 or 4 (see solve_items)
inline void launch_items(int ksplit, const ItemArrays &P, int64_t item_begin, int nitems, int nchunks, double *X, int64_t ldx, double *S, int64_t lds,
                         hipStream_t stream)
{
    const int cpw = nchunks < 4 / ksplit ? nchunks : 4 / ksplit, ipw = 4 / (ksplit * cpw), groups = (nchunks + cpw - 1) / cpw;
    const int64_t works = (int64_t)((nitems + ipw - 1) / ipw) * groups;
    const unsigned wgs = (unsigned)((works + 7) / 8 * 8);
    if (ksplit == 4) hipLaunchKernelGGL((solve_items<4, 2, 4>), dim3(wgs), dim3(256), 0, stream, P, item_begin, nitems, nchunks, X, ldx, S, lds);
    else if (ksplit == 2) hipLaunchKernelGGL((solve_items<4, 2, 2>), dim3(wgs), dim3(256), 0, stream, P, item_begin, nitems, nchunks, X, ldx, S, lds);
    else hipLaunchKernelGGL((solve_items<4, 2, 1>), dim3(wgs), dim3(256), 0, stream, P, item_begin, nitems, nchunks, X, ldx, S, lds);
}

}  // namespace dev
}  // namespace rsp
