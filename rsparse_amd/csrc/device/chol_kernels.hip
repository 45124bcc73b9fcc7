// HIP kernels (gfx950) for the supernodal right-looking numeric Cholesky (see chol_engine.hpp).
//
// Storage is the reference's CSC layout of L itself (Lp == cp, diagonal first, rows ascending), so the
// result needs no conversion.  A panel is a run of <= 32 consecutive columns c0..c1-1 with the supernode
// property  rows(col c0+t) = {c0+t, .., c1-1} U R  (R = rows below the panel, shared by all its columns).
// "Position" pos of a row in the panel: pos = row - c0 inside the diagonal block, w + index-in-R below it.
// Column c0+t holds len0 - t entries (len0 = w + |R|), so with base = Lp[c0]
//     entry (pos, t)  lives at  Lx[base + t*len0 - t(t-1)/2 + pos - t]        (pos >= t)
// -- a closed form: no kernel below ever chases Lp through memory.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rsp {
namespace dev {

__device__ __forceinline__ double readlane_f64(double v, int src_lane);  // blocked_kernels.hip

__device__ __forceinline__ int64_t panel_entry(int64_t base, int len0, int pos, int t)
{
    return base + (int64_t)t * len0 - (int64_t)((t * (t - 1)) >> 1) + (pos - t);
}

// L = 0 was memset; place the entries of triu(P A P') (duplicates resolved on the host: dst < 0 = ignored)
// Every kernel of the factorisation takes a batch dimension: blockIdx.y = system, all systems share the pattern
// (and therefore every index array) and differ in their values only: Ax + y * ax_stride, Lx + y * lx_stride.
__global__ __launch_bounds__(256) void chol_scatter_A(const double *__restrict__ Ax, const int64_t *__restrict__ dst, int64_t nnz,
                                                      double *__restrict__ Lx, int64_t ax_stride, int64_t lx_stride)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const int64_t d = dst[e];
    if (d >= 0) Lx[(int64_t)blockIdx.y * lx_stride + d] = Ax[(int64_t)blockIdx.y * ax_stride + e];
}

// ---- rank update: factored columns subtract their outer product from the ancestor panels they touch --------------
// A supernode S (W consecutive columns, rows R_S below them) is one dense trapezoid in the closed form above; a pair
// names a range of its columns [t_begin, t_end), the rows below them (`rs` rows: the q index space, row q sits at
// position len0_S - rs + q of S), and a target panel s: rows q0 .. q0 + ncols of the q space are columns of s (bit t of
// `mask`: column c0_s + t).  The update is
//     L_s(row q, column q0 + j)  -=  sum_{t in [t_begin, t_end)} L_S(q, t) * L_S(q0 + j, t)        for q >= q0 + j.
// Two kinds of pair (host: chol_engine.hip):
//   * INSIDE a supernode, panel by panel (t range = one panel of <= 32 columns, target = a later panel of S, issued at
//     the panel's level): nothing else writes S while its panels are being factored, so the result is subtracted with
//     plain loads and stores;
//   * FROM a whole supernode to a panel of an ANCESTOR supernode (t range = all W columns, issued once S is complete):
//     accumulated over all of S in registers and subtracted with ONE hardware f64 atomic per entry, because
//     different supernodes of a level may update the same ancestor entry -- the only place where the order of a
//     floating-point summation is not fixed (values agree to rounding; the integer structure is exact).
// One wavefront owns (pair, 16 consecutive q): a 32 x 16 tile on the matrix cores (v_mfma_f64_16x16x4_f64, A = the
// <= 32 column rows q0 + j, B = the 16 rows q, K = the t range in chunks of 32), the operands of a chunk fetched in
// one round trip.
struct CholPair {        // 64 bytes
    int64_t src_base;    // Lp[first column of S]
    int64_t tgt_base;    // Lp[c0_s]
    int64_t rel_off;     // into rel: position in panel s of row q, q = q0 ...
    int32_t len_d;       // len0 of S
    int32_t len_s;       // len0 of the target panel
    int32_t q0;
    uint32_t mask;
    int32_t rs;          // rows of the q space
    int32_t t_begin, t_end;
    int32_t atomic;
    int32_t pad[2];
};

struct CholPanel;
struct CholUpdateArgs {
    const int32_t *item_pair;  // work items of all levels: (pair, first q of the tile); pair < 0: copy the parked
    const int32_t *item_q;     //   diagonal block of panel ~pair from the side buffer into L (see chol_panel_factor)
    const CholPair *pairs;
    const int32_t *rel;
    double *Lx;
    const CholPanel *panels;
    const double *dbuf;
};

__device__ void chol_unpark_diagonal(const CholUpdateArgs &A, int panel, int lane);

typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void chol_rank_update(CholUpdateArgs A, int64_t item_begin, int64_t nitems, int64_t lx_stride,
                                                        int64_t dbuf_stride)
{
    A.Lx += (int64_t)blockIdx.y * lx_stride;
    A.dbuf += (int64_t)blockIdx.y * dbuf_stride;
    const int lane = threadIdx.x & 63;
    const int64_t it =
        item_begin + (int64_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    if (it >= item_begin + nitems) return;
    const int pr = A.item_pair[it], qt = A.item_q[it];
    if (pr < 0) {  // wave-uniform
        chol_unpark_diagonal(A, ~pr, lane);
        return;
    }
    const CholPair P = A.pairs[pr];
    const int g = lane >> 4, j = lane & 15;
    const int ncols = __popc(P.mask);
    const bool two = ncols > 16;
    const int qr = qt + j, rowbase = P.len_d - P.rs;  // position in S of row q = rowbase + q
    const bool row_ok = qr < P.rs, col0_ok = j < ncols, col1_ok = two && 16 + j < ncols;
    const int pos_r = rowbase + qr, pos_c0 = rowbase + P.q0 + j, pos_c1 = pos_c0 + 16;

    const int relpos = row_ok ? A.rel[P.rel_off + (qr - P.q0)] : 0;
    // C layout: lane (g, j), register s  <->  column number 4s + g (+ 16) of the pair, row qt + j.  One store / atomic
    // instruction therefore covers 16 consecutive rows of 4 target columns: whole 128-byte runs of L_s.
    // tc[c]: target column (bit position in the mask) of column number 4c + g, c = 0..7
    int tc[8];
    {
        uint32_t m = P.mask;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            tc[c] = 0;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const int t = m ? __builtin_ctz(m) : 0;  // wave-uniform
                if (gg == g) tc[c] = t;
                m &= m - 1;
            }
        }
    }

    f64x4 c0 = {0.0, 0.0, 0.0, 0.0}, c1 = {0.0, 0.0, 0.0, 0.0};
    for (int t0 = P.t_begin; t0 < P.t_end; t0 += 32) {  // wave-uniform
        const int wc = min(32, P.t_end - t0);
        const int ksteps = (wc + 3) >> 2;
        // one round trip per chunk: A = the rows that are target COLUMNS, B = the tile's rows
        double a0[8], a1[8], b[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            a0[k] = a1[k] = b[k] = 0.0;
            if (k < ksteps) {
                const int u = 4 * k + g;
                const bool col = u < wc;
                if (col && row_ok) b[k] = A.Lx[panel_entry(P.src_base, P.len_d, pos_r, t0 + u)];
                if (col && col0_ok) a0[k] = A.Lx[panel_entry(P.src_base, P.len_d, pos_c0, t0 + u)];
                if (col && col1_ok) a1[k] = A.Lx[panel_entry(P.src_base, P.len_d, pos_c1, t0 + u)];
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < ksteps) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[k], b[k], c0, 0, 0, 0);
                if (two) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[k], b[k], c1, 0, 0, 0);
            }
    }

    if (!row_ok) return;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int cn = 4 * s + g;  // column number inside the pair
        if (cn < ncols && qr >= P.q0 + cn) {
            double *dst = A.Lx + panel_entry(P.tgt_base, P.len_s, relpos, tc[s]);
            if (P.atomic) unsafeAtomicAdd(dst, -c0[s]);
            else *dst -= c0[s];
        }
        if (two && 16 + cn < ncols && qr >= P.q0 + 16 + cn) {
            double *dst = A.Lx + panel_entry(P.tgt_base, P.len_s, relpos, tc[4 + s]);
            if (P.atomic) unsafeAtomicAdd(dst, -c1[s]);
            else *dst -= c1[s];
        }
    }
}

// ---- panel factorisation -------------------------------------------------------------------------------------
struct CholPanel {   // 24 bytes
    int64_t base;    // Lp[c0]
    int32_t w, r;    // columns, rows below the diagonal block
    int32_t dslot;   // tall panels (r > 64): slot of the factored diagonal block in the side buffer; else -1
    int32_t pad;
};

struct CholFactorItem {  // 32 bytes: the panel's record travels with the item (one round trip before the data)
    CholPanel panel;
    int32_t block;       // row block of 64, or -1: diagonal block only
    int32_t c0;          // first column (for the status word)
};

struct CholFactorArgs {
    const CholFactorItem *items;
    double *Lx;
    int32_t *status;  // smallest column whose pivot d <= 0 (lib.rs:325)
    double *dbuf;     // side buffer: 1024 doubles per tall panel (mode 2)
};

__device__ void chol_unpark_diagonal(const CholUpdateArgs &A, int panel, int lane)
{
    const CholPanel P = A.panels[panel];
    const int len0 = P.w + P.r;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int idx = 64 * k + lane, rp = idx >> 5, t = idx & 31;
        if (rp < P.w && t <= rp) A.Lx[panel_entry(P.base, len0, rp, t)] = A.dbuf[(int64_t)P.dslot * 1024 + rp * 32 + t];
    }
}

// One wavefront per work item (panel, block).  Three kinds of item, chosen on the host so that no wavefront
// reads a diagonal block while another one overwrites it:
//   block >= 0, trsm_only == 0 : panel with <= 64 rows below the block: factor the w x w diagonal block in LDS,
//                                write it back, then solve this panel's rows (all in one wavefront);
//   block == -1                : panel with more rows: factor the diagonal block and write it back, nothing else;
//   trsm_only == 1 (next launch): load the already factored block and solve rows [64*block, 64*block + 64).
// Row solve: lane = one row below the block,  x <- x * D^-T  (L(k,i) = x_i / L(i,i), lib.rs:310).
// Mode 2 (levels with few panels, where a launch costs more than arithmetic): ONE launch; every row-block wavefront
// of a tall panel factors the diagonal block itself, from the unfactored entries in L, and solves its rows.  Nobody
// may overwrite those entries meanwhile, so wavefront 0 parks the factored block in a side buffer, and the level's
// update launch copies it into L (chol_rank_update, item_pair < 0).
#ifdef RSP_CHOL_CLOCKS
__device__ unsigned long long g_chol_clk[16];
#define CLK(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && nitems >= 6 && nitems <= 8) g_chol_clk[i] = wall_clock64(); } while (0)
#else
#define CLK(i)
#endif

__global__ __launch_bounds__(256) void chol_panel_factor(CholFactorArgs A, int64_t item_begin, int64_t nitems, int trsm_only,
                                                         int64_t lx_stride, int64_t dbuf_stride)
{
    A.Lx += (int64_t)blockIdx.y * lx_stride;
    A.dbuf += (int64_t)blockIdx.y * dbuf_stride;
    A.status += blockIdx.y;  // one status word per system
    __shared__ __attribute__((aligned(16))) double tiles[4][32][34];  // rows of 34: 16-byte aligned runs for the row solve
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int64_t it = item_begin + (int64_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wv));
    if (it >= item_begin + nitems) return;
    double(*D)[34] = tiles[wv];
    CLK(0);
    const CholFactorItem I = A.items[it];
    const CholPanel P = I.panel;
    const int blk = I.block, w = P.w, r = P.r, len0 = w + r;

    // the diagonal block and (when this item solves rows) this lane's row: all loads of one round trip
    const int rr = blk * 64 + lane;
    const bool solve = blk >= 0 && rr < r;
    const int pos = w + rr;
    double x[32];
#pragma unroll
    for (int t = 0; t < 32; ++t) x[t] = (solve && t < w) ? A.Lx[panel_entry(P.base, len0, pos, t)] : 0.0;
    if (trsm_only != 1) {
        CLK(1);
        // dense Cholesky of the diagonal block in registers: lane = row, d[t] = entry (lane, t); the pivot column
        // travels by readlane, so the 32 column steps are straight-line code without a single memory access
        double d[32];
#pragma unroll
        for (int t = 0; t < 32; ++t) d[t] = (lane < w && t <= lane) ? A.Lx[panel_entry(P.base, len0, lane, t)] : 0.0;
#ifdef RSP_CHOL_CLOCKS
        { double sum = 0; for (int t = 0; t < 32; ++t) sum += d[t] + x[t]; if (sum == 1.2345e300) A.status[0] = 1; }
        CLK(2);
#endif
#pragma unroll
        for (int t = 0; t < 32; ++t) {  // all loads land here: without this fence the compiler interleaves the phases and
            asm volatile("" : "+v"(d[t]));  // spills 600 registers (measured: 26 ms instead of 11.7 for the factorisation)
            asm volatile("" : "+v"(x[t]));
        }
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            if (t < w) {  // wave-uniform
                const double dtt = readlane_f64(d[t], t);
                if (dtt <= 0.0 && lane == 0 && blk <= 0) atomicMin(A.status, I.c0 + t);
                // 1 / sqrt(d) from the hardware's estimate and two Newton steps, sqrt(d) = d * rsqrt(d) corrected once (full
                // double precision): the library's sqrt and division are two dependent ~60-instruction sequences per column
                // on the critical path of the whole factorisation (measured with in-kernel timestamps: 14 of the 27 us of a
                // chain-level launch were this loop).  L(k,i) = x_i / L(i,i) (lib.rs:310) is a multiplication by rl.
                double rl = __builtin_amdgcn_rsq(dtt);
                rl = rl * fma(-0.5 * dtt * rl, rl, 1.5);
                rl = rl * fma(-0.5 * dtt * rl, rl, 1.5);
                const double l0 = dtt * rl;
                const double l = fma(fma(-l0, l0, dtt), 0.5 * rl, l0);
                d[t] = lane == t ? l : d[t] * rl;
                if (lane == 0) D[t][t] = rl;  // the row solve multiplies by the reciprocal diagonal
#pragma unroll
                for (int jj = t + 1; jj < 32; ++jj)  // no test of jj < w: lanes >= w hold zeros, the update is then a no-op
                    d[jj] -= d[t] * readlane_f64(d[t], jj);  // above the diagonal (lane < jj) this collects values nothing reads
                // (the pivot column through LDS broadcasts instead of v_readlane: 9.3 us for the 32 columns against 7.1)
            }
        }
        CLK(3);
        __builtin_amdgcn_sched_barrier(0);
        const bool parked = trsm_only == 2 && P.dslot >= 0;  // mode 2, tall panel: L keeps the unfactored block for now
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            if (lane < w && t <= lane) {
                if (!parked) A.Lx[panel_entry(P.base, len0, lane, t)] = d[t];
                else if (blk == 0) A.dbuf[(int64_t)P.dslot * 1024 + lane * 32 + t] = d[t];
            }
            if (lane < 32 && (lane != t || t >= w)) D[t][lane] = d[t];  // TRANSPOSED: D[column][row]; the diagonal already holds its reciprocal
        }
        if (blk < 0) return;
    } else {
        double dv[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int idx = 64 * k + lane, t = idx >> 5, rp = idx & 31;
            dv[k] = (t < w && rp < w && rp >= t) ? A.Lx[panel_entry(P.base, len0, rp, t)] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int idx = 64 * k + lane;
            D[idx >> 5][idx & 31] = ((idx & 31) == (idx >> 5) && (idx >> 5) < w) ? 1.0 / dv[k] : dv[k];  // D[column][row]
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_sched_barrier(0);
    CLK(4);

    if (solve) {
#pragma unroll
        // by columns (right-looking): once x[u] is final the 31 - u updates it feeds are INDEPENDENT multiply-adds; the
        // row-wise form is one dependent chain of 496 (measured 4.6 us against 1.x).  No test of u < w: rows and columns
        // >= w of D are zeros, those x stay zero and are not stored.
        for (int u = 0; u < 32; ++u) {
            x[u] *= D[u][u];
#pragma unroll
            for (int t = u + 1; t < 32; ++t) x[t] -= x[u] * D[u][t];  // a run of one LDS row: 16-byte broadcast reads
        }
#ifdef RSP_CHOL_CLOCKS
        { double sum = 0; for (int t = 0; t < 32; ++t) sum += x[t]; if (sum == 1.2345e300) A.status[0] = 1; }
        CLK(5);
#endif
#pragma unroll
        for (int t = 0; t < 32; ++t) asm volatile("" : "+v"(x[t]));
#pragma unroll
        for (int t = 0; t < 32; ++t)
            if (t < w) A.Lx[panel_entry(P.base, len0, pos, t)] = x[t];
    }
    CLK(6);
}

// ---- rank update, target-owned (atomics-free) --------------------------------------------------------------------
// The same contributions as chol_rank_update, regrouped by what they WRITE: a 16-row x 32-column tile of a target
// panel is owned, at every level, by one wavefront, which walks the (host-ordered) list of source ranges of that level
// that touch the tile, accumulates them in TARGET coordinates -- A operand: column i of the panel <-> the
// popcount(mask below i)-th column row of the source; B operand: row j of the tile <-> the source row the host's
// 16-entry map names, or none -- and subtracts the tile from L once, with plain loads and stores.  No two wavefronts
// of a launch write the same entry and the order of every sum is fixed by the host: the factor is bit-reproducible.
struct CholSub {        // 64 bytes: one source range's contribution to one target tile
    int64_t src_base;   // Lp[first column of the source supernode]
    int32_t len_d, rs;  // its len0; rows of the q space
    int32_t q0;         // first row of the q space that is a column of the target panel
    uint32_t mask;      // bit i: column c0_s + i of the target panel is a row of the source
    int32_t t_begin, t_end;
    int32_t qa;         // first source row that falls into the tile
    int32_t pad;
    uint32_t rowmap[4]; // byte j: source row of tile row j, relative to qa, or 0xff
    int32_t pad2[2];
};

struct CholGroup {      // 32 bytes: one target tile at one level
    int64_t tgt_base;   // Lp[c0_s]
    int64_t sub_begin;
    int32_t len_s, w_s; // w_s < 0: -w_s columns, and every source range of the group has at most 4 columns (fast path)
    int32_t r16;        // rows 16 r16 .. 16 r16 + 15 of the panel; -1: copy the parked diagonal block of panel sub_count
    int32_t sub_count;
};

struct CholTileArgs {
    const CholGroup *groups;
    const CholSub *subs;
    double *Lx;
    const CholPanel *panels;
    const double *dbuf;
};

// L(pos, i) -= c for the 8 entries of the tile this lane owns.  All eight loads first, then all eight stores: written as
// eight `-=` the compiler must assume the stores alias the later loads and serialises eight round trips (measured with
// in-kernel timestamps: 5 of the 10 us of a chain-level update launch).
__device__ __forceinline__ void chol_subtract_tile(double *Lx, int64_t tgt_base, int len_s, int w_s, int pos, int g, const f64x4 &c0,
                                                   const f64x4 &c1)
{
    double *p0[4], *p1[4];
    double v0[4], v1[4];
    bool ok0[4], ok1[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int i0 = 4 * s + g, i1 = i0 + 16;
        ok0[s] = i0 < w_s && i0 <= pos;
        ok1[s] = i1 < w_s && i1 <= pos;
        p0[s] = Lx + panel_entry(tgt_base, len_s, pos, ok0[s] ? i0 : 0);  // (pos, 0) always exists: the loads are unconditional
        p1[s] = Lx + panel_entry(tgt_base, len_s, pos, ok1[s] ? i1 : 0);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        v0[s] = *p0[s];
        v1[s] = *p1[s];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (ok0[s]) *p0[s] = v0[s] - c0[s];
        if (ok1[s]) *p1[s] = v1[s] - c1[s];
    }
}

__global__ __launch_bounds__(256) void chol_tile_update(CholTileArgs A, int64_t group_begin, int64_t ngroups, int64_t lx_stride,
                                                        int64_t dbuf_stride)
{
    A.Lx += (int64_t)blockIdx.y * lx_stride;
    A.dbuf += (int64_t)blockIdx.y * dbuf_stride;
    const int lane = threadIdx.x & 63;
    const int64_t it =
        group_begin + (int64_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    if (it >= group_begin + ngroups) return;
#ifdef RSP_CHOL_CLOCKS
#define CLKU(i) do { if (it == group_begin + ngroups - 1 && lane == 0 && ngroups >= 6 && ngroups <= 40) g_chol_clk[8 + i] = wall_clock64(); } while (0)
#else
#define CLKU(i)
#endif
    CLKU(0);
    const CholGroup G = A.groups[it];
    if (G.r16 < 0) {  // wave-uniform
        CholUpdateArgs U{nullptr, nullptr, nullptr, nullptr, A.Lx, A.panels, A.dbuf};
        chol_unpark_diagonal(U, G.sub_count, lane);
        return;
    }
    const int g = lane >> 4, j = lane & 15;
    const uint32_t below0 = (1u << j) - 1u, below1 = (1u << (16 + j)) - 1u;
    f64x4 c0 = {0.0, 0.0, 0.0, 0.0}, c1 = {0.0, 0.0, 0.0, 0.0};
    if (G.w_s < 0) {
        // every source range of this group has at most 4 columns (the tiles next to the leaves: several one-column
        // contributions each): four ranges per round trip instead of one
        const int w_s = -G.w_s;
        for (int k0 = 0; k0 < G.sub_count; k0 += 4) {
            double a0[4], a1[4], b[4];
            bool two[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool have = k0 + u < G.sub_count;  // wave-uniform
                const CholSub S = A.subs[G.sub_begin + min(k0 + u, G.sub_count - 1)];
                const int rowbase = S.len_d - S.rs;
                const uint32_t word = j < 8 ? (j < 4 ? S.rowmap[0] : S.rowmap[1]) : (j < 12 ? S.rowmap[2] : S.rowmap[3]);
                const int rm = (int)((word >> (8 * (j & 3))) & 0xffu);
                const bool col = have && S.t_begin + g < S.t_end;
                two[u] = have && (S.mask >> 16) != 0u;
                a0[u] = a1[u] = b[u] = 0.0;
                if (col && rm != 0xff) b[u] = A.Lx[panel_entry(S.src_base, S.len_d, rowbase + S.qa + rm, S.t_begin + g)];
                if (col && ((S.mask >> j) & 1u))
                    a0[u] = A.Lx[panel_entry(S.src_base, S.len_d, rowbase + S.q0 + __popc(S.mask & below0), S.t_begin + g)];
                if (col && ((S.mask >> (16 + j)) & 1u))
                    a1[u] = A.Lx[panel_entry(S.src_base, S.len_d, rowbase + S.q0 + __popc(S.mask & below1), S.t_begin + g)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (k0 + u < G.sub_count) {
                    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b[u], c0, 0, 0, 0);
                    if (two[u]) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b[u], c1, 0, 0, 0);
                }
        }
        const int pos = 16 * G.r16 + j;
        if (pos >= G.len_s) return;
        chol_subtract_tile(A.Lx, G.tgt_base, G.len_s, w_s, pos, g, c0, c1);
        return;
    }
    CLKU(1);
    CholSub next = A.subs[G.sub_begin];  // wave-uniform (scalar loads); a group has at least one source range
    for (int k = 0; k < G.sub_count; ++k) {
        const CholSub S = next;
        next = A.subs[G.sub_begin + min(k + 1, G.sub_count - 1)];  // in flight while this range is gathered and multiplied
        const int rowbase = S.len_d - S.rs;
        const uint32_t word = j < 8 ? (j < 4 ? S.rowmap[0] : S.rowmap[1]) : (j < 12 ? S.rowmap[2] : S.rowmap[3]);
        const int rm = (int)((word >> (8 * (j & 3))) & 0xffu);
        const bool row_ok = rm != 0xff;
        const bool col0_ok = (S.mask >> j) & 1u, col1_ok = (S.mask >> (16 + j)) & 1u;
        const bool two = (S.mask >> 16) != 0u;
        const int pos_r = rowbase + S.qa + rm;
        const int pos_c0 = rowbase + S.q0 + __popc(S.mask & below0), pos_c1 = rowbase + S.q0 + __popc(S.mask & below1);
        for (int t0 = S.t_begin; t0 < S.t_end; t0 += 32) {  // wave-uniform
            const int wc = min(32, S.t_end - t0);
            const int ksteps = (wc + 3) >> 2;
            double a0[8], a1[8], b[8];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                a0[kk] = a1[kk] = b[kk] = 0.0;
                if (kk < ksteps) {
                    const int u = 4 * kk + g;
                    const bool col = u < wc;
                    if (col && row_ok) b[kk] = A.Lx[panel_entry(S.src_base, S.len_d, pos_r, t0 + u)];
                    if (col && col0_ok) a0[kk] = A.Lx[panel_entry(S.src_base, S.len_d, pos_c0, t0 + u)];
                    if (two && col && col1_ok) a1[kk] = A.Lx[panel_entry(S.src_base, S.len_d, pos_c1, t0 + u)];
                }
            }
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
                if (kk < ksteps) {
                    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[kk], b[kk], c0, 0, 0, 0);
                    if (two) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[kk], b[kk], c1, 0, 0, 0);
                }
        }
    }
    // C layout: lane (g, j), register s  <->  panel column 4s + g (+ 16), tile row j
#ifdef RSP_CHOL_CLOCKS
    { double sm = c0[0] + c0[1] + c0[2] + c0[3] + c1[0] + c1[1] + c1[2] + c1[3]; if (sm == 1.2345e300) A.Lx[0] = 0; }
    CLKU(2);
    if (it == group_begin + ngroups - 1 && lane == 0 && ngroups >= 6 && ngroups <= 40) g_chol_clk[12] = (unsigned long long)G.sub_count;
#endif
    const int pos = 16 * G.r16 + j;
    if (pos >= G.len_s) return;
    chol_subtract_tile(A.Lx, G.tgt_base, G.len_s, G.w_s, pos, g, c0, c1);
}

// ---- tiny panels: factor, row solve and update of a whole leaf supernode in a quarter of a wavefront ----------------
// Half of the columns of a 2-D mesh factor are one-column leaves with two or three rows below them; a wavefront per
// panel would idle 60 of its 64 lanes.  Panels with w <= 4 columns and w + r <= 16 rows that are a supernode of
// their own run FOUR to a wavefront: a group of 16 lanes owns a panel, lane = row position (diagonal-block rows
// first, then the rows below), the <= 4 columns live in registers and the pivot column travels by ds_bpermute.  The
// same lanes then subtract the outer product of the rows below from the ancestors (hardware f64 atomics, as for every
// update that leaves a supernode); the target entry of every (row, row') pair was tabulated by the host.
struct CholTiny {     // 32 bytes
    int64_t base;     // Lp[c0]
    int64_t tgt_off;  // into tiny_tgt: entry index in Lx of (R[q], R[q']), q' <= q, at tgt_off + q (q + 1) / 2 + q'
    int32_t w, r, c0, pad;
};

struct CholTinyArgs {
    const CholTiny *items;
    const int64_t *tgt;
    double *Lx;
    int32_t *status;
};

__global__ __launch_bounds__(256) void chol_tiny_panels(CholTinyArgs A, int64_t item_begin, int64_t nitems, int64_t lx_stride, int update)
{
    double *Lx = A.Lx + (int64_t)blockIdx.y * lx_stride;
    int32_t *status = A.status + blockIdx.y;
    const int lane = threadIdx.x & 63, li = lane & 15, gbase = lane & 48;
    const int64_t it = item_begin + ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / 16;
    const bool have = it < item_begin + nitems;
    CholTiny P = {0, 0, 0, 0, 0, 0};
    if (have) P = A.items[it];
    const int w = P.w, r = P.r, len0 = w + r, pos = li;
    const bool row = have && pos < len0;
    double d[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) d[t] = (row && t < w && t <= pos) ? Lx[panel_entry(P.base, len0, pos, t)] : 0.0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const double dtt = __shfl(d[t], gbase + t, 64);  // every lane takes part in every exchange
        double piv[4];
        if (t < w) {
            if (dtt <= 0.0 && li == 0) atomicMin(status, P.c0 + t);
            const double l = sqrt(dtt);
            if (row && pos >= t) d[t] = pos == t ? l : d[t] / l;  // L(k,i) = x_i / L(i,i), lib.rs:310
        }
#pragma unroll
        for (int jj = t + 1; jj < 4; ++jj) piv[jj] = __shfl(d[t], gbase + jj, 64);
#pragma unroll
        for (int jj = t + 1; jj < 4; ++jj)
            if (t < w && jj < w && row && pos >= jj) d[jj] -= d[t] * piv[jj];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
        if (row && t < w && t <= pos) Lx[panel_entry(P.base, len0, pos, t)] = d[t];
    if (!update) return;  // wave-uniform: the target-owned update kernel does it
    // rank update from the rows below: lane = row q, loop over the rows q' <= q
    const int q = pos - w;
    const bool below = row && q >= 0;
    const int64_t tbase = P.tgt_off + ((int64_t)q * (q + 1) >> 1);
    for (int qq = 0; qq < 15; ++qq) {  // w >= 1, so r <= 15; uniform bound, lanes drop out by predicate
        double v = 0.0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const double other = __shfl(d[t], (gbase + w + qq) & 63, 64);
            if (t < w) v += d[t] * other;
        }
        if (below && qq <= q && qq < r) unsafeAtomicAdd(Lx + A.tgt[tbase + qq], -v);
    }
}

// ---- batched cholsol solve phase: many small systems with one pattern --------------------------------------------
// One workgroup per system; the vector lives in LDS for the whole of  x = P b, L x = b, L' x = b, b = P' x
// (lib.rs:383-386).  Both sweeps are level-scheduled gathers (one thread per unknown of the level, a workgroup
// barrier between levels): lsolve reads row r of L through a CSR index built once on the host (columns ascending:
// the same subtraction order as the reference's column sweep), ltsolve reads column j of the CSC as it is.
struct CholBatchSolveArgs {
    const int64_t *Lp;        // n + 1 (diagonal first in every column)
    const int32_t *Li;        // lnz, row of every entry
    const int32_t *row_ptr;   // n + 1: strictly-lower entries of row r ...
    const int32_t *row_col;   // ... their columns (ascending) ...
    const int32_t *row_pos;   // ... and their positions in Lx
    const int32_t *lev_ptr_l, *lev_rows_l;  // lsolve: rows by level
    const int32_t *lev_ptr_t, *lev_cols_t;  // ltsolve: columns by level
    const int64_t *pinv;      // may be null
    const double *Lx;
    double *B;                // nsys x n
    int64_t lx_stride;
    int32_t n, nlev_l, nlev_t;
};

__global__ __launch_bounds__(256) void chol_batch_solve(CholBatchSolveArgs A)
{
    extern __shared__ double xs[];
    const double *Lx = A.Lx + (int64_t)blockIdx.x * A.lx_stride;
    double *b = A.B + (int64_t)blockIdx.x * A.n;
    const int tid = threadIdx.x, n = A.n;
    for (int k = tid; k < n; k += 256) xs[A.pinv ? A.pinv[k] : k] = b[k];  // ipvec, lib.rs:2151
    __syncthreads();
    for (int l = 0; l < A.nlev_l; ++l) {
        for (int k = A.lev_ptr_l[l] + tid; k < A.lev_ptr_l[l + 1]; k += 256) {
            const int r = A.lev_rows_l[k];
            double acc = xs[r];
            for (int e = A.row_ptr[r]; e < A.row_ptr[r + 1]; ++e) acc -= Lx[A.row_pos[e]] * xs[A.row_col[e]];
            xs[r] = acc / Lx[A.Lp[r]];
        }
        __syncthreads();
    }
    for (int l = 0; l < A.nlev_t; ++l) {
        for (int k = A.lev_ptr_t[l] + tid; k < A.lev_ptr_t[l + 1]; k += 256) {
            const int j = A.lev_cols_t[k];
            const int64_t p0 = A.Lp[j], p1 = A.Lp[j + 1];
            double acc = xs[j];
            for (int64_t p = p0 + 1; p < p1; ++p) acc -= Lx[p] * xs[A.Li[p]];
            xs[j] = acc / Lx[p0];
        }
        __syncthreads();
    }
    for (int k = tid; k < n; k += 256) b[k] = xs[A.pinv ? A.pinv[k] : k];  // pvec, lib.rs:2244
}

}  // namespace dev
}  // namespace rsp
