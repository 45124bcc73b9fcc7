// HIP kernel (gfx950) of the blocked multi-right-hand-side triangular solve (programme: host/blocked.cpp).
//
// One wavefront owns (task, 64 right-hand sides) and executes the task's blocks in order.  A block computes
//     acc[r] = init[r] - sum_e coef[e][r] * src[e]      (+ the block's own small triangular solve)
// for up to 16 unknowns r at once, so every gathered source row is used for all unknowns of the block.
// Under load a dependent global round trip costs microseconds, so every path below is written as THREE of them:
//   1. the 32-byte block header (one scalar load),
//   2. everything the header points to that is an index or a coefficient: destination / initial-value ids, the ids
//      and coefficients of the first sources, the block's triangle -- all issued together,
//   3. the rows themselves (initial values and first sources together),
// after which long source lists run as a software pipeline (loads of the next stage in flight during the arithmetic).
// 16-unknown blocks run on the matrix cores (v_mfma_f64_16x16x4_f64); blocks of 1, 2 or 4 unknowns in per-lane registers.
// The host pads src_idx and coef by one wavefront's worth of entries, so clamped speculative fetches stay in bounds.
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
    const int32_t *prologue;     // BlockProgram::kPrologueWords int32 per record (host/sparse.hpp)
};

// A prologue record spread over the wavefront: lane l holds words l and 64 + l.  Header, destination / initial-value ids
// and the ids of the first stage(s) arrive in ONE round trip; after it the wavefront goes straight to the rows of X.
struct Prologue {
    int32_t r0, r1;
};

template <bool SECOND>
__device__ __forceinline__ Prologue load_prologue(const BlockArrays &B, int64_t rec, int lane)
{
    const int32_t *p = B.prologue + rec * BlockProgram::kPrologueWords;
    Prologue R;
    R.r0 = p[lane];
    R.r1 = SECOND ? p[64 + lane] : 0;
    return R;
}

__device__ __forceinline__ BlockHeader header_of(const Prologue &R)  // wave-uniform: lives in scalar registers
{
    BlockHeader h;
    h.rc = __builtin_amdgcn_readlane(R.r0, 0);
    h.nsrc = __builtin_amdgcn_readlane(R.r0, 1);
    h.has_tri = __builtin_amdgcn_readlane(R.r0, 2);
    h.pad = 0;
    h.src_off = (int64_t)(uint32_t)__builtin_amdgcn_readlane(R.r0, 4) | ((int64_t)__builtin_amdgcn_readlane(R.r0, 5) << 32);
    h.coef_off = (int64_t)(uint32_t)__builtin_amdgcn_readlane(R.r0, 6) | ((int64_t)__builtin_amdgcn_readlane(R.r0, 7) << 32);
    return h;
}

__device__ __forceinline__ int32_t word_at(int32_t r, int idx)  // word `idx` (0..63, per lane) of a lane-distributed vector
{
    return __builtin_amdgcn_ds_bpermute(idx << 2, r);
}

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
__device__ __forceinline__ void run_block_vec(const BlockHeader h, const Prologue &R, const BlockArrays &B, double *X, int64_t ldx, double *S,
                                              int64_t lds, int col, bool live, int lane)
{
    const double *cf = B.coef + h.coef_off;
    const int32_t *sidx = B.src_idx + h.src_off + 2 * RC;
    constexpr int SB = 16;  // sources per batch: 16 ids and 16*RC <= 64 coefficients, one vector load each
    // the record already holds the dest / init ids (words 8.., 24..) and the ids of the first batch (words 40..):
    // round trip 2 fetches the rows together with the first coefficients and the triangle
    int nb = min(SB, h.nsrc);
    int32_t idv = word_at(R.r0, (40 + lane) & 63);  // lane u < 16: id of source u
    double cv = cf[lane < nb * RC ? lane : 0];
    const double tv = cf[(int64_t)h.nsrc * RC + (lane < RC * RC ? lane : 0)];  // padded pool: in bounds without a triangle
    double acc[RC];
#pragma unroll
    for (int r = 0; r < RC; ++r) {
        const int32_t id = __builtin_amdgcn_readlane(R.r0, 24 + r);
        acc[r] = (id != RSP_NONE && live) ? row_ptr(id, X, ldx, S, lds)[col] : 0.0;
    }
    for (int e0 = 0; e0 < h.nsrc; e0 += SB) {
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
        // ids and coefficients of the next batch travel while this one is multiplied
        const int nb_next = min(SB, h.nsrc - e0 - SB);
        const int32_t idv_next = sidx[e0 + SB + (lane < nb_next ? lane : 0)];
        const double cv_next = cf[(int64_t)(e0 + SB) * RC + (lane < nb_next * RC ? lane : 0)];
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            if (u < nb) {
#pragma unroll
                for (int r = 0; r < RC; ++r) acc[r] -= readlane_f64(cv, u * RC + r) * x[u];
            }
        }
        nb = nb_next;
        idv = idv_next;
        cv = cv_next;
    }
    if (h.has_tri) {
#pragma unroll
        for (int r = 0; r < RC; ++r) {
#pragma unroll
            for (int t = 0; t < r; ++t) acc[r] -= readlane_f64(tv, r * RC + t) * acc[t];
            acc[r] /= readlane_f64(tv, r * RC + r);
        }
    }
#pragma unroll
    for (int r = 0; r < RC; ++r) {
        const int32_t id = __builtin_amdgcn_readlane(R.r0, 8 + r);
        if (id != RSP_NONE && live) const_cast<double *>(row_ptr(id, X, ldx, S, lds))[col] = acc[r];
    }
}

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// ---- matrix-core path: 16 unknowns x 64 right-hand sides ----------------------------------------------------
// acc lives in the C/D layout of v_mfma_f64_16x16x4_f64: lane (g = lane >> 4, j = lane & 15), register s <-> unknown
// 4*s + g, columns cw + 4*j + nt for the 4 tiles nt (every lane owns 4 CONSECUTIVE doubles of a row: 32-byte accesses).
// One K-step consumes 4 sources: A = coefficients (lane (g, i): coef[source 4k+g][unknown i] -- a fully coalesced
// 512-byte read of the coefficient stream), B = the gathered rows (lane (g, j): 4 doubles of source 4k+g).
// The in-block triangular solve is 4 more K-steps with A = inverse of the block's 16 x 16 triangle and B = acc itself
// (C/D register s IS the B operand of K-step s), so nothing ever crosses lanes.
//
// Sources are consumed in STAGES of 16 (4 K-steps, 16 MFMAs).  The loads of stage s+1 (16 row pieces, 4 coefficient
// vectors) and the ids of stage s+2 are issued before the MFMAs of stage s; every load is unconditional (stage numbers
// are clamped), so the compiler's vmcnt bookkeeping stays exact and a stage's MFMAs wait for that stage's loads only.
struct Stage {
    double a[4];
    f64x2 lo[4], hi[4];
};

struct PanelCtx {
    const int32_t *sidx;  // source ids of the block
    const double *cf;     // source coefficients of the block
    const double *X, *S;
    int64_t ldx, lds;
    int g, lane, cb, nrhs;
    bool in;              // this lane's 4 columns exist (FAST: all or none)
};

__device__ __forceinline__ void load_ids(int32_t (&id)[4], const PanelCtx &c, int st)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) id[q] = c.sidx[st * 16 + 4 * q + c.g];
}

template <bool FAST>
__device__ __forceinline__ void load_stage(Stage &R, const int32_t (&id)[4], const PanelCtx &c, int st)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        R.a[q] = -c.cf[(int64_t)(st * 16 + 4 * q) * 16 + c.lane];
        const double *p = row_ptr(id[q], c.X, c.ldx, c.S, c.lds);
        if (FAST) {  // 16-byte aligned rows, nrhs a multiple of 4: lanes without columns re-read column 0
            p += c.in ? c.cb : 0;
            R.lo[q] = *reinterpret_cast<const f64x2 *>(p);
            R.hi[q] = *reinterpret_cast<const f64x2 *>(p + 2);
        } else {
            p += c.cb;
            R.lo[q].x = c.cb + 0 < c.nrhs ? p[0] : 0.0;
            R.lo[q].y = c.cb + 1 < c.nrhs ? p[1] : 0.0;
            R.hi[q].x = c.cb + 2 < c.nrhs ? p[2] : 0.0;
            R.hi[q].y = c.cb + 3 < c.nrhs ? p[3] : 0.0;
        }
    }
}

__device__ __forceinline__ void mfma_stage(f64x4 (&acc)[4], const Stage &R)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(R.a[q], R.lo[q].x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(R.a[q], R.lo[q].y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(R.a[q], R.hi[q].x, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(R.a[q], R.hi[q].y, acc[3], 0, 0, 0);
    }
}

// ids and triangle of a block, fetched in the same round trip as the first source ids
struct PanelIds {
    int32_t init[4], dest[4];  // unknown 4s + g for s = 0..3
    double tinv[4];            // A operands of the inverse triangle
};

__device__ __forceinline__ void load_panel_tinv(PanelIds &P, const BlockHeader h, const PanelCtx &c)
{
    const double *tinv = c.cf + (int64_t)h.nsrc * 16;  // padded pool: in bounds without a triangle
#pragma unroll
    for (int k = 0; k < 4; ++k) P.tinv[k] = tinv[k * 64 + c.lane];
}

__device__ __forceinline__ void panel_ids_from(PanelIds &P, const Prologue &R, const PanelCtx &c, bool lead)
{
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        P.dest[s] = word_at(R.r0, 8 + 4 * s + c.g);
        const int32_t init = word_at(R.r0, 24 + 4 * s + c.g);
        P.init[s] = lead ? init : RSP_NONE;
    }
}

// rows of the block -> acc (register s <- unknown 4s + g); kNone = zero
template <bool FAST>
__device__ __forceinline__ void panel_load_rows(f64x4 (&acc)[4], const int32_t (&id4)[4], const PanelCtx &c)
{
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int32_t id = id4[s];
        double v[4] = {0.0, 0.0, 0.0, 0.0};
        if (id != RSP_NONE) {
            const double *p = row_ptr(id, c.X, c.ldx, c.S, c.lds) + c.cb;
            if (FAST) {
                if (c.in) {
                    const f64x2 lo = *reinterpret_cast<const f64x2 *>(p), hi = *reinterpret_cast<const f64x2 *>(p + 2);
                    v[0] = lo.x; v[1] = lo.y; v[2] = hi.x; v[3] = hi.y;
                }
            } else {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    if (c.cb + nt < c.nrhs) v[nt] = p[nt];
            }
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt][s] = v[nt];
    }
}

// acc = init rows - coef[st_begin*16 .. st_end*16) * sources.  `lead`: this wavefront loads the initial values.
// PIPE: two stages of registers, the next stage's loads in flight during this stage's MFMAs -- the shortest critical
// path, for levels with little work.  !PIPE: one stage of registers (half the footprint, twice the wavefronts per CU):
// a stage's loads are only issued after the previous stage's MFMAs, and it is the OTHER wavefronts of the CU that
// cover the wait -- the better trade on levels with thousands of blocks, where throughput is what counts.
// Lean variant with HALF stages (PIPE == 2): the same 16 sources' worth of registers as the one-stage lean variant, but as
// two half stages of 8 that alternate -- while one half is multiplied the other one is in flight, and the ids run two
// halves ahead, so a wavefront never sits with nothing outstanding.
struct HalfStage {
    double a[2];
    f64x2 lo[2], hi[2];
};

__device__ __forceinline__ void load_half_ids(int32_t (&id)[2], const PanelCtx &c, int hs)
{
#pragma unroll
    for (int q = 0; q < 2; ++q) id[q] = c.sidx[hs * 8 + 4 * q + c.g];
}

template <bool FAST>
__device__ __forceinline__ void load_half(HalfStage &R, const int32_t (&id)[2], const PanelCtx &c, int hs)
{
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        R.a[q] = -c.cf[(int64_t)(hs * 8 + 4 * q) * 16 + c.lane];
        const double *p = row_ptr(id[q], c.X, c.ldx, c.S, c.lds);
        if (FAST) {
            p += c.in ? c.cb : 0;
            R.lo[q] = *reinterpret_cast<const f64x2 *>(p);
            R.hi[q] = *reinterpret_cast<const f64x2 *>(p + 2);
        } else {
            p += c.cb;
            R.lo[q].x = c.cb + 0 < c.nrhs ? p[0] : 0.0;
            R.lo[q].y = c.cb + 1 < c.nrhs ? p[1] : 0.0;
            R.hi[q].x = c.cb + 2 < c.nrhs ? p[2] : 0.0;
            R.hi[q].y = c.cb + 3 < c.nrhs ? p[3] : 0.0;
        }
    }
}

__device__ __forceinline__ void mfma_half(f64x4 (&acc)[4], const HalfStage &R)
{
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(R.a[q], R.lo[q].x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(R.a[q], R.lo[q].y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(R.a[q], R.hi[q].x, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(R.a[q], R.hi[q].y, acc[3], 0, 0, 0);
    }
}

// stages [st_begin, st_end) of 16 sources = half stages [2 st_begin, 2 st_end); the record holds the ids of the first two
template <bool FAST>
__device__ __forceinline__ void gather_halves(f64x4 (&acc)[4], const Prologue &R, const PanelCtx &c, int st_begin, int st_end)
{
    if (st_begin >= st_end) return;  // wave-uniform
    const int h0 = 2 * st_begin, hend = 2 * st_end, hlast = hend - 1;
    int32_t idA[2], idB[2], idA2[2], idB2[2];
    HalfStage A, B;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        idA[q] = word_at(R.r0, 40 + 4 * q + c.g);
        idB[q] = word_at(R.r0, 48 + 4 * q + c.g);
    }
    load_half_ids(idA2, c, min(h0 + 2, hlast));
    load_half_ids(idB2, c, min(h0 + 3, hlast));
    load_half<FAST>(A, idA, c, h0);
    load_half<FAST>(B, idB, c, h0 + 1);
    for (int hs = h0; hs < hend; hs += 2) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            idA[q] = idA2[q];
            idB[q] = idB2[q];
        }
        load_half_ids(idA2, c, min(hs + 4, hlast));
        load_half_ids(idB2, c, min(hs + 5, hlast));
        mfma_half(acc, A);
        load_half<FAST>(A, idA, c, min(hs + 2, hlast));
        mfma_half(acc, B);
        load_half<FAST>(B, idB, c, min(hs + 3, hlast));
    }
}

// ONE right-hand side: nothing but latency counts (a level is a handful of wavefronts waiting for a handful of doubles),
// and registers are plentiful because only column 0 exists.  All stages of the wavefront (at most DEPTH per batch) are
// fetched at once -- every id in one round trip, then every row and coefficient in one more -- instead of one
// dependent gather per stage; only the first of the four column tiles is computed.
template <int DEPTH>
__device__ __forceinline__ void panel_gather_single(f64x4 (&acc)[4], const PanelCtx &c, int st_begin, int st_end)
{
    const bool mine = (c.lane & 15) == 0;  // lanes (g, 0) hold the B operand of column 0
    for (int s0 = st_begin; s0 < st_end; s0 += DEPTH) {
        int32_t id[DEPTH][4];
        double a[DEPTH][4], b[DEPTH][4];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int st = min(s0 + d, st_end - 1);  // clamped: loads stay unconditional
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                id[d][q] = c.sidx[st * 16 + 4 * q + c.g];
                a[d][q] = -c.cf[(int64_t)(st * 16 + 4 * q) * 16 + c.lane];
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int q = 0; q < 4; ++q) b[d][q] = mine ? row_ptr(id[d][q], c.X, c.ldx, c.S, c.lds)[0] : 0.0;
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (s0 + d < st_end) {  // wave-uniform
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[d][q], b[d][q], acc[0], 0, 0, 0);
            }
    }
}

template <bool FAST, int PIPE>
__device__ __forceinline__ void panel_gather(f64x4 (&acc)[4], PanelIds &P, const Prologue &R, const BlockHeader h, const PanelCtx &c,
                                             int st_begin, int st_end, bool lead)
{
    if (!FAST && c.nrhs == 1) {  // wave-uniform
        panel_ids_from(P, R, c, lead);
        panel_load_rows<false>(acc, P.init, c);
        load_panel_tinv(P, h, c);
        panel_gather_single<PIPE == 1 ? 8 : 2>(acc, c, st_begin, st_end);
        return;
    }
    const int last = st_end - 1;
    const bool any = st_begin < st_end;  // wave-uniform; a block may have no sources at all
    int32_t idA[4];
    Stage A;
    // the record holds the block's ids and those of this wavefront's first stage (and the one after it)
    panel_ids_from(P, R, c, lead);
#pragma unroll
    for (int q = 0; q < 4; ++q) idA[q] = word_at(R.r0, 40 + 4 * q + c.g);
    if (PIPE == 2) {
        panel_load_rows<FAST>(acc, P.init, c);
        gather_halves<FAST>(acc, R, c, st_begin, st_end);
        load_panel_tinv(P, h, c);
    } else if (PIPE == 1) {
        int32_t idB[4];
        Stage B;
#pragma unroll
        for (int q = 0; q < 4; ++q) idB[q] = word_at(R.r1, 4 * q + c.g);  // used only if the wavefront has a second stage
        // round trip 2
        panel_load_rows<FAST>(acc, P.init, c);
        load_panel_tinv(P, h, c);
        if (!any) return;
        load_stage<FAST>(A, idA, c, st_begin);
        for (int s = st_begin; s < st_end; s += 2) {
            load_stage<FAST>(B, idB, c, min(s + 1, last));
            load_ids(idA, c, min(s + 2, last));
            mfma_stage(acc, A);
            load_stage<FAST>(A, idA, c, min(s + 2, last));
            load_ids(idB, c, min(s + 3, last));
            if (s + 1 < st_end) mfma_stage(acc, B);
        }
    } else {
        panel_load_rows<FAST>(acc, P.init, c);
        if (any) {
            load_stage<FAST>(A, idA, c, st_begin);
            for (int s = st_begin; s < st_end; ++s) {
                load_ids(idA, c, min(s + 1, last));
                mfma_stage(acc, A);
                load_stage<FAST>(A, idA, c, min(s + 1, last));
            }
        }
        load_panel_tinv(P, h, c);
    }
}

// in-block triangle (as its inverse) and the store of the 16 unknowns
template <bool FAST>
__device__ __forceinline__ void panel_finish(f64x4 (&acc)[4], const PanelIds &P, const BlockHeader h, const PanelCtx &c, double *X, double *S)
{
    if (h.has_tri) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            f64x4 r = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < 4; ++k) r = __builtin_amdgcn_mfma_f64_16x16x4f64(P.tinv[k], acc[nt][k], r, 0, 0, 0);
            acc[nt] = r;
        }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int32_t id = P.dest[s];
        if (id != RSP_NONE) {
            double *p = const_cast<double *>(row_ptr(id, X, c.ldx, S, c.lds)) + c.cb;
            if (FAST) {
                if (c.in) {
                    *reinterpret_cast<f64x2 *>(p) = f64x2{acc[0][s], acc[1][s]};
                    *reinterpret_cast<f64x2 *>(p + 2) = f64x2{acc[2][s], acc[3][s]};
                }
            } else {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    if (c.cb + nt < c.nrhs) p[nt] = acc[nt][s];
            }
        }
    }
}

__device__ __forceinline__ PanelCtx make_ctx(const BlockHeader h, const BlockArrays &B, const double *X, int64_t ldx, const double *S,
                                             int64_t lds, int cw, int nrhs, int lane)
{
    PanelCtx c;
    c.sidx = B.src_idx + h.src_off + 32;
    c.cf = B.coef + h.coef_off;
    c.X = X;
    c.S = S;
    c.ldx = ldx;
    c.lds = lds;
    c.g = lane >> 4;
    c.lane = lane;
    c.cb = cw + 4 * (lane & 15);
    c.nrhs = nrhs;
    c.in = c.cb + 4 <= nrhs;
    return c;
}

template <bool FAST, int PIPE>
__device__ __forceinline__ void run_block_mfma(const BlockHeader h, const Prologue &R, const BlockArrays &B, double *X, int64_t ldx, double *S,
                                               int64_t lds, int cw, int nrhs, int lane)
{
    const PanelCtx c = make_ctx(h, B, X, ldx, S, lds, cw, nrhs, lane);
    PanelIds P;
    f64x4 acc[4];
    panel_gather<FAST, PIPE>(acc, P, R, h, c, 0, h.nsrc >> 4, true);
    panel_finish<FAST>(acc, P, h, c, X, S);
}

// ---- one launch per level ----------------------------------------------------------------------------------------
// A level holds block tasks of three split classes (host/blocked.cpp): 0 = one wavefront per (task, 64 columns),
// 1 = four wavefronts share the sources of one matrix-core block, 2 = eight wavefronts.  All of them run in ONE
// launch of 512-thread workgroups; a workgroup's role follows from its index.
//
// XCD-aware placement: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 share an XCD, speed only), each
// with its own 4 MiB L2.  Work item i = (blockIdx % 8) * per_xcd + blockIdx / 8, so every XCD walks a CONTIGUOUS
// range of items; inside the split classes items are ordered (group of 8 blocks, 64-column chunk, block), so the
// workgroups an XCD runs together are consecutive blocks of one supernode on the SAME columns -- they gather the same
// rows of X, and all but the first find them in that XCD's L2.
struct LevelDesc {
    int64_t task_begin;      // class 0: first task
    int64_t block_begin[2];  // classes 1, 2: first block (one block per task)
    int64_t blk0, rec0;      // class 0: first block and its prologue record (records follow the block order)
    int64_t rec_begin[2];    // classes 1, 2: record of the first block (4 / 8 records per block)
    int ntasks, nblk[2];
    int wg0, wg1, wg2;       // workgroups per class
    int single;              // every class-0 task is one block: block = blk0 + task, no task_ptr lookup
};

// class 0: the blocks of a task run in order.  t = task number inside the launch.
template <bool FAST, int PIPE>
__device__ __forceinline__ void run_task(const BlockArrays &B, int64_t task_begin, int t, int64_t blk0, int64_t rec0, bool single, int chunk,
                                         double *X, int64_t ldx, double *S, int64_t lds, int nrhs, int lane)
{
    const int cw = chunk * 64;
    const int col = cw + lane;
    const bool live = col < nrhs;
    int64_t b = blk0 + t, bend = b + 1;
    if (!single) {
        b = B.task_ptr[task_begin + t];
        bend = B.task_ptr[task_begin + t + 1];
    }
    for (; b < bend; ++b) {
        const Prologue R = load_prologue<PIPE == 1>(B, rec0 + (b - blk0), lane);
        const BlockHeader h = header_of(R);
        switch (h.rc) {
            case 16: run_block_mfma<FAST, PIPE>(h, R, B, X, ldx, S, lds, cw, nrhs, lane); break;
            case 1: run_block_vec<1>(h, R, B, X, ldx, S, lds, col, live, lane); break;
            case 2: run_block_vec<2>(h, R, B, X, ldx, S, lds, col, live, lane); break;
            default: run_block_vec<4>(h, R, B, X, ldx, S, lds, col, live, lane); break;
        }
    }
}

// (group of 8 blocks, chunk, block) order of item q over n blocks
__device__ __forceinline__ void split_item(int q, int n, int nchunks, int &blk, int &chunk)
{
    const int per_group = 8 * nchunks;
    const int group = q / per_group, r = q - group * per_group;
    const int in_group = min(8, n - group * 8);
    chunk = r / in_group;
    blk = group * 8 + (r - chunk * in_group);
}

// W wavefronts (w = 0..W-1) share one matrix-core block: contiguous ranges of its stages each, partial accumulators
// meet in LDS (red: (W - 1) x 8 KiB), wavefront 0 finishes.  Every wavefront of the workgroup reaches the barrier.
template <int W, bool FAST, int PIPE>
__device__ __forceinline__ void run_split(const BlockArrays &B, bool valid, int64_t rec, int chunk, int w, double *red, double *X,
                                          int64_t ldx, double *S, int64_t lds, int nrhs, int lane)
{
    PanelIds P;
    f64x4 acc[4];
    BlockHeader h;
    PanelCtx c;
    if (valid) {
        const Prologue R = load_prologue<PIPE == 1>(B, rec + w, lane);  // this wavefront's record of the block
        h = header_of(R);
        c = make_ctx(h, B, X, ldx, S, lds, chunk * 64, nrhs, lane);
        const int nst = h.nsrc >> 4;
        panel_gather<FAST, PIPE>(acc, P, R, h, c, (nst * w) / W, (nst * (w + 1)) / W, w == 0);
        if (w > 0) {
            double *mine = red + (size_t)(w - 1) * 1024 + lane;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int s = 0; s < 4; ++s) mine[(nt * 4 + s) * 64] = acc[nt][s];
        }
    }
    __syncthreads();
    if (valid && w == 0) {
        for (int o = 0; o < W - 1; ++o) {
            const double *theirs = red + (size_t)o * 1024 + lane;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[nt][s] += theirs[(nt * 4 + s) * 64];
        }
        panel_finish<FAST>(acc, P, h, c, X, S);
    }
}

// Levels with thousands of small-block tasks (the rows just above the clusters) are bound by how many row loads the
// chip keeps in flight: this lean kernel (no matrix-core path, a third of the registers) runs them at 3-4x the occupancy.
__global__ __launch_bounds__(256) void solve_level_small(BlockArrays B, int64_t task_begin, int ntasks, double *X, int64_t ldx, double *S,
                                                         int64_t lds, int nrhs, int nchunks, int64_t blk0, int64_t rec0, int single)
{
    const int lane = threadIdx.x & 63;
    const int per_xcd = (int)(gridDim.x >> 3);  // XCD-aware placement as in solve_level: consecutive tasks share an XCD
    const int item = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    const int pair = __builtin_amdgcn_readfirstlane((int)(item * 4 + (threadIdx.x >> 6)));
    const int t = pair / nchunks;
    if (t >= ntasks) return;
    const int col = (pair - t * nchunks) * 64 + lane;
    const bool live = col < nrhs;
    int64_t b = blk0 + t, bend = b + 1;
    if (!single) {
        b = B.task_ptr[task_begin + t];
        bend = B.task_ptr[task_begin + t + 1];
    }
    for (; b < bend; ++b) {
        const Prologue R = load_prologue<false>(B, rec0 + (b - blk0), lane);
        const BlockHeader h = header_of(R);
        switch (h.rc) {
            case 1: run_block_vec<1>(h, R, B, X, ldx, S, lds, col, live, lane); break;
            case 2: run_block_vec<2>(h, R, B, X, ldx, S, lds, col, live, lane); break;
            default: run_block_vec<4>(h, R, B, X, ldx, S, lds, col, live, lane); break;
        }
    }
}

template <bool FAST, int PIPE>
__global__ __launch_bounds__(512, PIPE == 1 ? 2 : 4) void solve_level(BlockArrays B, LevelDesc L, double *X, int64_t ldx, double *S, int64_t lds, int nrhs,
                                                    int nchunks)
{
    extern __shared__ double red[];  // 7 x 8 KiB
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // 0..7
    const int per_xcd = (int)(gridDim.x >> 3);                              // the grid is a multiple of 8 workgroups
    const int item = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    // longest work first: the 8-wavefront blocks, then the 4-wavefront ones, then the small tasks fill the tail
    if (item < L.wg2) {  // 1 (block, chunk) pair, 8 wavefronts
        int blk, chunk;
        split_item(item, L.nblk[1], nchunks, blk, chunk);
        run_split<8, FAST, PIPE>(B, true, L.rec_begin[1] + (int64_t)blk * 8, chunk, w, red, X, ldx, S, lds, nrhs, lane);
    } else if (item < L.wg2 + L.wg1) {  // 2 (block, chunk) pairs, 4 wavefronts each
        const int q = (item - L.wg2) * 2 + (w >> 2);
        const bool valid = q < L.nblk[0] * nchunks;
        int blk = 0, chunk = 0;
        if (valid) split_item(q, L.nblk[0], nchunks, blk, chunk);
        run_split<4, FAST, PIPE>(B, valid, L.rec_begin[0] + (int64_t)blk * 4, chunk, w & 3, red + (size_t)(w >> 2) * 3 * 1024, X, ldx, S, lds, nrhs, lane);
    } else if (item < L.wg2 + L.wg1 + L.wg0) {  // 8 (task, chunk) pairs
        const int pair = (item - L.wg2 - L.wg1) * 8 + w;
        const int t = pair / nchunks;
        if (t < L.ntasks) run_task<FAST, PIPE>(B, L.task_begin, t, L.blk0, L.rec0, L.single != 0, pair - t * nchunks, X, ldx, S, lds, nrhs, lane);
    }
}

}  // namespace dev
}  // namespace rsp
