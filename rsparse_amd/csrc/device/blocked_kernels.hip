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

#ifdef RSP_TRI_CLOCKS
// debug aid: phase timestamps (10 ns ticks) of wavefront 0 of workgroup 0 of the class-3 part of a level launch
__device__ unsigned long long g_tri_clk[16];
#define TCLK(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_tri_clk[i] = wall_clock64(); } while (0)
#else
#define TCLK(i)
#endif

using BlockHeader = ::rsp::BlockHeader;  // 32 bytes, one s_load_dwordx8 (host/sparse.hpp)

struct BlockArrays {
    const int64_t *task_ptr;     // blocks of task t (tasks sorted by level)
    const BlockHeader *hdr;
    const int32_t *src_idx;      // id >= 0 row of X, id < 0 scratch row ~id, INT32_MIN none
    const double *coef;
    const int32_t *prologue;     // BlockProgram::kPrologueWords int32 per record (host/sparse.hpp)
};

// A prologue record spread over the wavefront (host/sparse.hpp: lines of 64 words): lane l holds word l of line 0 (header,
// destination / initial-value ids, a vector block's first source ids) and word l of the id lines (matrix-core blocks: the
// ids of all sources of this wavefront).  Everything arrives in ONE round trip; after it the wavefront goes straight to X.
struct Prologue {
    int32_t r0, ids0, ids1;
};

typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x3 __attribute__((ext_vector_type(3)));

// LINES: lines of the record (2 or 3, interleaved: lane l owns words l * LINES ...); NL: id lines wanted (0: a vector block)
template <int LINES, int NL>
__device__ __forceinline__ Prologue load_prologue(const BlockArrays &B, int64_t line, int lane)
{
    const int32_t *p = B.prologue + line * BlockProgram::kLineWords + lane * LINES;
    Prologue R;
    if (NL == 0) {
        R.r0 = p[0];
        R.ids0 = R.ids1 = 0;
    } else if (LINES == 2) {
        const i32x2 v = *reinterpret_cast<const i32x2 *>(p);  // ONE load: the ids must not become a second round trip
        R.r0 = v.x;
        R.ids0 = v.y;
        R.ids1 = 0;
    } else {
        const i32x3 v = *reinterpret_cast<const i32x3 *>(p);
        R.r0 = v.x;
        R.ids0 = v.y;
        R.ids1 = v.z;
    }
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

// COH (solve_levels_flagged: several levels in one launch): rows of X and scratch rows written by another wavefront of the SAME
// launch are read past the caches that are not coherent across XCDs (agent-scope accesses).  Plain loads would not be safe
// even though every row is written once: narrow blocks put 16 rows into one cache line, and an XCD that fetched the line
// for a finished neighbour would serve a stale copy of a row that was still pending.
template <bool COH>
__device__ __forceinline__ double load_x(const double *p)
{
    return COH ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
template <bool COH>
__device__ __forceinline__ void store_x(double *p, double v)
{
    if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// ---- vector path: RC <= 4 unknowns, one right-hand side per lane ---------------------------------------
template <int RC, bool COH = false>
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
        acc[r] = (id != RSP_NONE && live) ? load_x<COH>(row_ptr(id, X, ldx, S, lds) + col) : 0.0;
    }
    for (int e0 = 0; e0 < h.nsrc; e0 += SB) {
        double x[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            if (u < nb) {  // wave-uniform
                const int32_t id = __builtin_amdgcn_readlane(idv, u);
                x[u] = live ? load_x<COH>(row_ptr(id, X, ldx, S, lds) + col) : 0.0;
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
        if (id != RSP_NONE && live) store_x<COH>(const_cast<double *>(row_ptr(id, X, ldx, S, lds)) + col, acc[r]);
    }
}

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// ---- matrix-core path: 16 unknowns x 16 NT right-hand sides ------------------------------------------------
// acc lives in the C/D layout of v_mfma_f64_16x16x4_f64: lane (g = lane >> 4, j = lane & 15), register s <-> unknown
// 4*s + g, columns cw + NT*j + nt for the NT tiles nt (every lane owns NT CONSECUTIVE doubles of a row: one 8-, 16- or
// two 16-byte accesses).  One K-step consumes 4 sources: A = coefficients (lane (g, i): coef[source 4k+g][unknown i] --
// a fully coalesced 512-byte read of the coefficient stream), B = the gathered rows (lane (g, j): NT doubles of source
// 4k+g).  The in-block triangular solve is 4 more K-steps with A = inverse of the block's 16 x 16 triangle and B = acc
// itself (C/D register s IS the B operand of K-step s), so nothing ever crosses lanes.
//
// NT = 4 (64 columns per wavefront) where a level has work for the whole chip; NT = 1 on the long tail of levels with
// a few dozen blocks: there a workgroup's own matrix-core time IS the launch's duration (one f64 MFMA occupies its SIMD
// for 64 cycles; a 1024-source block is 1024 of them per 64 columns), and 16-column wavefronts put four times as many
// compute units to work on it.
//
// What else bounds a block is the number of DEPENDENT round trips to memory, so the path is written as two of them:
//   1. the record (header, destination / initial-value ids, the ids of every source of this wavefront),
//   2. the initial rows, the triangle, and the rows and coefficients of the first D slots of the RING below,
// after which the ring keeps D slots of Q K-steps in flight: a slot is multiplied and immediately re-filled with the
// loads of the slot D places ahead.  The rules that keep the hardware's load counter exact (and the waits minimal):
// every load is unconditional (slots past the end read row 0), no loaded value is post-processed before its use (the
// sum is accumulated with the coefficients' own sign: acc = sum - init, negated once at the end), ring slots are
// compile-time registers (the loop is unrolled over the ring), and sched_barriers pin the order in which loads are
// issued.  Source ids never travel again: they sit in the wavefront's record registers and reach the lane that needs
// them through ds_bpermute.
template <int NT>
struct Piece {  // NT consecutive doubles of a row
    double v[NT];
};

template <int Q, int NT>
struct Slot {
    double a[Q];
    Piece<NT> b[Q];
};

struct PanelCtx {
    const int32_t *sidx;  // source ids of the block (pool)
    const double *cf;     // source coefficients of the block
    const double *X, *S;
    const char *Xl, *Sl;  // FAST: this lane's piece of row 0 of X / of the scratch rows (lanes without columns: column 0)
    int64_t ldx, lds;
    uint32_t ldx8, lds8;  // row pitch in bytes
    int g, lane, cb, nrhs;
    bool in;              // this lane's NT columns exist (FAST: all or none)
};

// this lane's piece of row `id`: FAST layout (aligned, whole pieces) with one 64-bit multiply-add per address
template <bool FAST, int NT, bool COH = false>
__device__ __forceinline__ Piece<NT> load_piece(int32_t id, const PanelCtx &c)
{
    Piece<NT> r;
    if (FAST) {
        const bool x = id >= 0;
        const uint32_t row = (uint32_t)(x ? id : ~id);
        const char *p = (x ? c.Xl : c.Sl) + (uint64_t)row * (uint64_t)(x ? c.ldx8 : c.lds8);
        if constexpr (NT == 4) {
            const f64x2 lo = *reinterpret_cast<const f64x2 *>(p), hi = *reinterpret_cast<const f64x2 *>(p + 16);
            r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = hi.x; r.v[3] = hi.y;
        } else if constexpr (NT == 2) {
            const f64x2 lo = *reinterpret_cast<const f64x2 *>(p);
            r.v[0] = lo.x; r.v[1] = lo.y;
        } else {
            r.v[0] = load_x<COH>(reinterpret_cast<const double *>(p));
        }
    } else {
        const double *p = row_ptr(id, c.X, c.ldx, c.S, c.lds) + c.cb;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) r.v[nt] = c.cb + nt < c.nrhs ? load_x<COH>(p + nt) : 0.0;
    }
    return r;
}

// K-steps [slot * Q, ...) of the current WINDOW, whose ids the record registers (ids0, ids1) hold (16 K-steps per
// register); `slot` and `nslots` count inside the window.  Slots past the end read row 0 of X (never used).
template <bool FAST, int Q, int NT, bool COH = false>
__device__ __forceinline__ void load_slot(Slot<Q, NT> &R, int32_t ids0, int32_t ids1, const PanelCtx &c, const double *cfw, int slot, int nslots)
{
    const bool valid = slot < nslots;  // wave-uniform
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int k = slot * Q + q;
        int32_t id = __builtin_amdgcn_ds_bpermute(((4 * k + c.g) & 63) << 2, (k & 16) ? ids1 : ids0);
        id = valid ? id : 0;
        R.a[q] = cfw[(valid ? (int64_t)k * 64 : 0) + c.lane];
        R.b[q] = load_piece<FAST, NT, COH>(id, c);
    }
}

template <int Q, int NT>
__device__ __forceinline__ void mfma_slot(f64x4 (&acc)[NT], const Slot<Q, NT> &R)
{
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(R.a[q], R.b[q].v[nt], acc[nt], 0, 0, 0);
}

// ids and triangle of a block
struct PanelIds {
    int32_t dest[4];  // unknown 4s + g for s = 0..3
    double tinv[4];   // A operands of the inverse triangle
};

__device__ __forceinline__ void load_panel_tinv(PanelIds &P, const BlockHeader h, const PanelCtx &c)
{
    const double *tinv = c.cf + (int64_t)h.nsrc * 16;  // padded pool: in bounds without a triangle
#pragma unroll
    for (int k = 0; k < 4; ++k) P.tinv[k] = tinv[k * 64 + c.lane];
}

// initial rows of the block -> v (register s <- unknown 4s + g); the loads are unconditional (rows that do not exist
// read row 0 and are zeroed afterwards), so all of them travel together with the ring's first slots
template <bool FAST, int NT, bool COH = false>
__device__ __forceinline__ void panel_load_init(f64x4 (&v)[NT], const Prologue &R, const PanelCtx &c, bool lead)
{
    int32_t id[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        id[s] = word_at(R.r0, 24 + 4 * s + c.g);
        if (!lead) id[s] = RSP_NONE;
    }
    Piece<NT> piece[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) piece[s] = load_piece<FAST, NT, COH>(id[s] != RSP_NONE ? id[s] : 0, c);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const bool have = id[s] != RSP_NONE && (!FAST || c.in);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) v[nt][s] = have ? piece[s].v[nt] : 0.0;
    }
}

// acc = coef[st_begin*16 .. st_end*16) * sources - initial rows (the caller negates).  `lead`: this wavefront owns the
// initial values.  LAT: the latency variant (levels with few workgroups: a deep ring, the triangle fetched up front);
// otherwise the lean one (4 slots of 4 sources, half the registers, twice the wavefronts per CU, initial rows and
// triangle fetched at the end).  NL: id lines of the record.  One right-hand side takes the same path (16-column
// wavefronts, element-wise loads): a special all-stages-at-once gather that fetched its ids from the pool cost one more
// dependent round trip per block than the record's id lines (2.53 against 2.39 ms per cholsol solve phase at C2).
template <bool FAST, bool LAT, int NL, int NT, bool COH = false>
__device__ __forceinline__ void panel_gather(f64x4 (&acc)[NT], PanelIds &P, const Prologue &R, const BlockHeader h, const PanelCtx &c,
                                             int st_begin, int st_end, bool lead)
{
#pragma unroll
    for (int s = 0; s < 4; ++s) P.dest[s] = word_at(R.r0, 8 + 4 * s + c.g);
    // K-steps per slot, slots of the ring (NT = 4: 10 registers per K-step; NT = 1: 4)
    constexpr int Q = LAT ? 2 : 1, D = LAT ? 7 : 4;
    constexpr int WIN = 16 * NL;  // K-steps whose ids the record registers hold
    const int nks = (st_end - st_begin) * 4;  // K-steps of this wavefront
    const double *cfw = c.cf + (int64_t)st_begin * 256;
    // LAT: the initial rows travel with the ring's first slots and start the sum (acc = -init); lean: registers are
    // short, the initial rows are fetched at the end together with the triangle (one round trip either way)
    // (the sched_barriers pin the ORDER in which loads are issued: loads return in order, and the compiler's scheduler
    // would otherwise happily issue the slot that is needed first last)
    if (LAT) {
        panel_load_init<FAST, NT, COH>(acc, R, c, lead);
        load_panel_tinv(P, h, c);
        __builtin_amdgcn_sched_barrier(0);
    }
    int32_t ids0 = R.ids0, ids1 = NL >= 2 ? R.ids1 : 0;
    Slot<Q, NT> ring[D];
    {
        const int nslots = min(nks, WIN) / Q;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            load_slot<FAST, Q, NT, COH>(ring[d], ids0, ids1, c, cfw, d, nslots);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = LAT ? -acc[nt] : f64x4{0.0, 0.0, 0.0, 0.0};
    for (int w0 = 0;;) {
        const int nslots = min(nks - w0, WIN) / Q;
        for (int base = 0; base < nslots; base += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (base + d < nslots) mfma_slot<Q, NT>(acc, ring[d]);  // wave-uniform
                load_slot<FAST, Q, NT, COH>(ring[d], ids0, ids1, c, cfw, base + d + D, nslots);
                __builtin_amdgcn_sched_barrier(0);  // keep the re-fill where it is: right behind the slot's last use
            }
        }
        w0 += WIN;
        if (w0 >= nks) break;
        // a wavefront with more sources than its record holds (only when one wavefront runs a long block alone): the
        // next window's ids come from the pool, the ring starts over
        cfw += (int64_t)WIN * 64;
        const int32_t *sw = c.sidx + (int64_t)st_begin * 16 + (int64_t)w0 * 4;
        const int left = (nks - w0) * 4;
        ids0 = sw[min(c.lane, left - 1)];
        if (NL >= 2) ids1 = sw[min(64 + c.lane, left - 1)];
        const int nnext = min(nks - w0, WIN) / Q;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            load_slot<FAST, Q, NT, COH>(ring[d], ids0, ids1, c, cfw, d, nnext);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (!LAT && lead) {  // wave-uniform
        f64x4 init[NT];
        panel_load_init<FAST, NT, COH>(init, R, c, true);
        load_panel_tinv(P, h, c);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] -= init[nt];
    }
}

// acc -> -acc, in-block triangle (as its inverse) and the store of the 16 unknowns
template <bool FAST, int NT, bool COH = false>
__device__ __forceinline__ void panel_finish(f64x4 (&acc)[NT], const PanelIds &P, const BlockHeader h, const PanelCtx &c, double *X, double *S)
{
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = -acc[nt];
    if (h.has_tri) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
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
                    if constexpr (NT == 4) {
                        *reinterpret_cast<f64x2 *>(p) = f64x2{acc[0][s], acc[1][s]};
                        *reinterpret_cast<f64x2 *>(p + 2) = f64x2{acc[2][s], acc[3][s]};
                    } else if constexpr (NT == 2) {
                        *reinterpret_cast<f64x2 *>(p) = f64x2{acc[0][s], acc[1][s]};
                    } else {
                        store_x<COH>(p, acc[0][s]);
                    }
                }
            } else {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    if (c.cb + nt < c.nrhs) store_x<COH>(p + nt, acc[nt][s]);
            }
        }
    }
}

template <int NT>
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
    c.cb = cw + NT * (lane & 15);
    c.nrhs = nrhs;
    c.in = c.cb + NT <= nrhs;
    c.ldx8 = (uint32_t)(ldx * 8);
    c.lds8 = (uint32_t)(lds * 8);
    c.Xl = reinterpret_cast<const char *>(X + (c.in ? c.cb : 0));
    c.Sl = reinterpret_cast<const char *>(S + (c.in ? c.cb : 0));
    return c;
}

template <bool FAST, bool LAT, int NT, bool COH = false>
__device__ __forceinline__ void run_block_mfma(const BlockHeader h, const Prologue &R, const BlockArrays &B, double *X, int64_t ldx, double *S,
                                               int64_t lds, int cw, int nrhs, int lane)
{
    const PanelCtx c = make_ctx<NT>(h, B, X, ldx, S, lds, cw, nrhs, lane);
    PanelIds P;
    f64x4 acc[NT];
    panel_gather<FAST, LAT, 1, NT, COH>(acc, P, R, h, c, 0, h.nsrc >> 4, true);
    panel_finish<FAST, NT, COH>(acc, P, h, c, X, S);
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
    int64_t blk0, rec0;      // class 0: first block and the LINE of its prologue record (records follow the block order, 2 lines each)
    int64_t rec_begin[2];    // split classes: line of the first block's first record (4 records of 2 lines / 8 records of 3 lines per block)
    int ntasks, nblk[2];
    int wg0, wg1, wg2;       // workgroups per class
    int single;              // every class-0 task is one block: block = blk0 + task, no task_ptr lookup
};

// class 0: the blocks of a task run in order.  t = task number inside the launch.  A wavefront owns 16 NT columns; the
// vector path (one column per lane) then uses the first 16 NT lanes only.
template <bool FAST, bool LAT, int NT, bool COH = false>
__device__ __forceinline__ void run_task(const BlockArrays &B, int64_t task_begin, int t, int64_t blk0, int64_t rec0, bool single, int chunk,
                                         double *X, int64_t ldx, double *S, int64_t lds, int nrhs, int lane)
{
    const int cw = chunk * 16 * NT;
    const int col = cw + lane;
    const bool live = col < nrhs && lane < 16 * NT;
    int64_t b = blk0 + t, bend = b + 1;
    if (!single) {
        b = B.task_ptr[task_begin + t];
        bend = B.task_ptr[task_begin + t + 1];
    }
    for (; b < bend; ++b) {
        const Prologue R = load_prologue<2, 1>(B, rec0 + 2 * (b - blk0), lane);
        const BlockHeader h = header_of(R);
        switch (h.rc) {
            case 16: run_block_mfma<FAST, LAT, NT, COH>(h, R, B, X, ldx, S, lds, cw, nrhs, lane); break;
            case 1: run_block_vec<1, COH>(h, R, B, X, ldx, S, lds, col, live, lane); break;
            case 2: run_block_vec<2, COH>(h, R, B, X, ldx, S, lds, col, live, lane); break;
            default: run_block_vec<4, COH>(h, R, B, X, ldx, S, lds, col, live, lane); break;
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
// meet in LDS (red: (W - 1) x 2 NT KiB), wavefront 0 finishes.  Every wavefront of the workgroup reaches the barrier.
template <int W, bool FAST, bool LAT, int NT, bool COH = false>
__device__ __forceinline__ void run_split(const BlockArrays &B, bool valid, int64_t rec, int chunk, int w, double *red, double *X,
                                          int64_t ldx, double *S, int64_t lds, int nrhs, int lane)
{
    PanelIds P;
    f64x4 acc[NT];
    BlockHeader h;
    PanelCtx c;
    if (valid) {
        constexpr int NL = W == 8 ? 2 : 1;
        if (W == 8) TCLK(0);
        const Prologue R = load_prologue<NL + 1, NL>(B, rec + (int64_t)w * (NL + 1), lane);  // this wavefront's record of the block
        h = header_of(R);
        if (W == 8) TCLK(1);
        c = make_ctx<NT>(h, B, X, ldx, S, lds, chunk * 16 * NT, nrhs, lane);
        const int nst = h.nsrc >> 4;
        panel_gather<FAST, LAT, NL, NT, COH>(acc, P, R, h, c, (nst * w) / W, (nst * (w + 1)) / W, w == 0);
#ifdef RSP_TRI_CLOCKS
        if (W == 8) {
            double sm = 0;
            for (int nt = 0; nt < NT; ++nt) sm += acc[nt][0] + acc[nt][1] + acc[nt][2] + acc[nt][3];
            if (sm == 1.2345e300) X[0] = 0;
            TCLK(2);
            if (blockIdx.x == 0 && threadIdx.x == 0) g_tri_clk[8] = (unsigned long long)nst;
        }
#endif
        if (w > 0) {
            double *mine = red + (size_t)(w - 1) * 256 * NT + lane;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int s = 0; s < 4; ++s) mine[(nt * 4 + s) * 64] = acc[nt][s];
        }
    }
    __syncthreads();
    if (W == 8) TCLK(3);
    if (valid && w == 0) {
        for (int o = 0; o < W - 1; ++o) {
            const double *theirs = red + (size_t)o * 256 * NT + lane;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[nt][s] += theirs[(nt * 4 + s) * 64];
        }
        panel_finish<FAST, NT, COH>(acc, P, h, c, X, S);
        if (W == 8) TCLK(4);
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
        const Prologue R = load_prologue<2, 0>(B, rec0 + 2 * (b - blk0), lane);
        const BlockHeader h = header_of(R);
        switch (h.rc) {
            case 1: run_block_vec<1>(h, R, B, X, ldx, S, lds, col, live, lane); break;
            case 2: run_block_vec<2>(h, R, B, X, ldx, S, lds, col, live, lane); break;
            default: run_block_vec<4>(h, R, B, X, ldx, S, lds, col, live, lane); break;
        }
    }
}

// One workgroup of a level: `item` is its (XCD-aware) number inside the level.  nchunks counts chunks of 16 NT columns.
template <bool FAST, bool LAT, int NT, bool COH = false>
__device__ __forceinline__ void level_workgroup(const BlockArrays &B, const LevelDesc &L, int item, int w, double *red, double *X, int64_t ldx,
                                                double *S, int64_t lds, int nrhs, int nchunks, int lane)
{
    // longest work first: the 8-wavefront blocks, then the 4-wavefront ones, then the small tasks fill the tail
    if (item < L.wg2) {  // 1 (block, chunk) pair, 8 wavefronts
        int blk, chunk;
        split_item(item, L.nblk[1], nchunks, blk, chunk);
        run_split<8, FAST, LAT, NT, COH>(B, true, L.rec_begin[1] + (int64_t)blk * 24, chunk, w, red, X, ldx, S, lds, nrhs, lane);
    } else if (item < L.wg2 + L.wg1) {  // 2 (block, chunk) pairs, 4 wavefronts each
        const int q = (item - L.wg2) * 2 + (w >> 2);
        const bool valid = q < L.nblk[0] * nchunks;
        int blk = 0, chunk = 0;
        if (valid) split_item(q, L.nblk[0], nchunks, blk, chunk);
        run_split<4, FAST, LAT, NT, COH>(B, valid, L.rec_begin[0] + (int64_t)blk * 8, chunk, w & 3, red + (size_t)(w >> 2) * 3 * 256 * NT, X, ldx, S, lds, nrhs,
                                    lane);
    } else if (item < L.wg2 + L.wg1 + L.wg0) {  // 8 (task, chunk) pairs
        const int pair = (item - L.wg2 - L.wg1) * 8 + w;
        const int t = pair / nchunks;
        if (t < L.ntasks) run_task<FAST, LAT, NT, COH>(B, L.task_begin, t, L.blk0, L.rec0, L.single != 0, pair - t * nchunks, X, ldx, S, lds, nrhs, lane);
    }
}

template <bool FAST, bool LAT, int NT>
__global__ __launch_bounds__(512, LAT ? 2 : 4) void solve_level(BlockArrays B, LevelDesc L, double *X, int64_t ldx, double *S, int64_t lds, int nrhs,
                                                                int nchunks)
{
    extern __shared__ double red[];  // 7 x 2 NT KiB
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // 0..7
    const int per_xcd = (int)(gridDim.x >> 3);                              // the grid is a multiple of 8 workgroups
    const int item = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    level_workgroup<FAST, LAT, NT>(B, L, item, w, red, X, ldx, S, lds, nrhs, nchunks, lane);
}

// ---- several consecutive latency-bound levels in ONE launch (narrow blocks: at most 16 columns) --------------------------
// A launch per level costs ~9 us when the level is a handful of workgroups whose own chain is ~4 us.  Here the workgroups of a
// RUN of such levels are one grid.  A workgroup takes a TICKET from a counter (tickets are numbered level by level, so that
// everything a ticket waits for was taken earlier, by a workgroup that is running), looks up its (level, item), waits for the
// tickets that produce its sources -- a host-built list per ticket, only producers of the same run; a flag per ticket, the
// whole first wavefront polling one flag at a time -- runs the level's ordinary code with coherent row accesses (COH) and
// raises its own flag once its stores have landed.  Nobody counts arrivals and nothing is fenced: the hand-over is one
// store and one poll, as in solve_rows_flagged.  Every wait is bounded (a flag that never rises is a defect: the wait gives
// up and marks the run, every other wait sees the mark, the grid drains, the host reports RSP_DEVICE_ERROR).
struct FlagRun {
    const LevelDesc *levels;   // of the run
    const int2 *tickets;       // (level of the run, item inside the level)
    const int32_t *wait_ptr;   // producers of ticket k: wait_list[wait_ptr[k] .. wait_ptr[k + 1]) (flag indices)
    const int32_t *wait_list;
    int *flags;                // flags[flag_base + k] == epoch: ticket k is done
    int *counter;              // next ticket (zero before the launch)
    int *failed;
    int flag_base, ntickets, epoch;
    unsigned spin_limit;
};

template <bool FAST>
__global__ __launch_bounds__(512, 2) void solve_levels_flagged(BlockArrays B, FlagRun F, double *X, int64_t ldx, double *S, int64_t lds, int nrhs)
{
    extern __shared__ double red[];  // 7 x 2 KiB
    __shared__ int ticket;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (threadIdx.x == 0) ticket = atomicAdd(F.counter, 1);
    __syncthreads();
    const int k = __builtin_amdgcn_readfirstlane(ticket);
    if (k >= F.ntickets) return;  // (the whole workgroup)
    const int2 where = F.tickets[k];
    const LevelDesc L = F.levels[where.x];
    if (w == 0) {
        const int begin = F.wait_ptr[k], end = F.wait_ptr[k + 1];
        unsigned spins = 0;
        bool gave_up = false;
        for (int p0 = begin; p0 < end && !gave_up; p0 += 64) {
            const bool in = p0 + lane < end;
            const int f = in ? F.wait_list[p0 + lane] : 0;
            bool ready = !in || __hip_atomic_load(F.flags + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == F.epoch;
            unsigned long long waiting = __ballot(!ready);
            while (waiting) {  // (uniform) the whole wavefront polls the first missing producer
                const int fw = __builtin_amdgcn_readlane(f, __builtin_ctzll(waiting));
                while (__hip_atomic_load(F.flags + fw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != F.epoch) {
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 1023u) == 0) {
                        if (spins >= F.spin_limit && lane == 0) atomicExch(F.failed, 1);
                        if (__hip_atomic_load(F.failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                            gave_up = true;
                            break;
                        }
                    }
                }
                if (gave_up) break;
                ready = !in || __hip_atomic_load(F.flags + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == F.epoch;
                waiting = __ballot(!ready);
            }
        }
    }
    __syncthreads();
    level_workgroup<FAST, true, 1, true>(B, L, where.y, w, red, X, ldx, S, lds, nrhs, 1, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wavefront's rows are in the L2 / on their way out before the flag is
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(F.flags + F.flag_base + k, F.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace dev
}  // namespace rsp
