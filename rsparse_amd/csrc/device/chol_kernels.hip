// HIP kernels (gfx950) for the supernodal left-looking numeric Cholesky (see chol_engine.hpp).
//
// Storage is the reference's CSC layout of L itself (Lp == cp, diagonal first, rows ascending), so the
// result needs no conversion.  A panel is a run of <= 32 consecutive columns c0..c1-1 with the supernode
// property  rows(col c0+t) = {c0+t, .., c1-1} U R  (R = rows below the panel, shared by all its columns).
// "Position" pos of a row in the panel: pos = row - c0 inside the diagonal block, w + index-in-R below it.
// Entry (pos, t) of the panel lives at  Lx[Lp[c0+t] + pos - t]   (pos >= t).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rsp {
namespace dev {

// L = 0 was memset; place the entries of triu(P A P') (duplicates resolved on the host: dst < 0 = ignored)
__global__ __launch_bounds__(256) void chol_scatter_A(const double *__restrict__ Ax, const int64_t *__restrict__ dst, int64_t nnz,
                                                      double *__restrict__ Lx)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const int64_t d = dst[e];
    if (d >= 0) Lx[d] = Ax[e];
}

struct CholUpdateArgs {
    const int32_t *slot_panel;  // active slots of this launch
    const int32_t *slot_pos;
    const int64_t *slot_tptr;   // tuples of slot a: [slot_tptr[a], slot_tptr[a+1])
    const int32_t *tuple_pair;
    const int32_t *tuple_q;
    const int32_t *pair_src;    // source panel d
    const int32_t *pair_q0;     // first row of R_d that lies in the target panel
    const uint32_t *pair_mask;  // bit t set: column c0_target + t is a row of R_d
    const int32_t *panel_c0;
    const int64_t *Lp;
    double *Lx;
};

// One wavefront per (target panel s, row position pos).  Lane l serves target column t = l & 31; the two
// half-waves split the columns u of each source panel.  For every descendant panel d that contains this row
// (tuple = (pair d->s, index q of the row in R_d)):
//     acc[t] += sum_u L_d(q, u) * L_d(qq(t), u)        qq(t) = index in R_d of column c0_s + t, if present, qq <= q
// and finally L_s(pos, t) -= acc[t].  Each entry of L is written by exactly one lane of one wavefront of the
// launch: no atomics, and the summation order is fixed by the host-built tuple order.
__global__ __launch_bounds__(256) void chol_panel_update(CholUpdateArgs A, int64_t slot_begin, int64_t nslots)
{
    const int lane = threadIdx.x & 63;
    const int64_t a =
        slot_begin + (int64_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    if (a >= slot_begin + nslots) return;
    const int s = A.slot_panel[a], pos = A.slot_pos[a];
    const int c0s = A.panel_c0[s], ws = A.panel_c0[s + 1] - c0s;
    const int t = lane & 31, half = lane >> 5;
    const uint32_t below = (1u << t) - 1u;
    double acc = 0.0;
    const int64_t tend = A.slot_tptr[a + 1];
    for (int64_t tp = A.slot_tptr[a]; tp < tend; ++tp) {
        const int pr = A.tuple_pair[tp], q = A.tuple_q[tp];
        const int d = A.pair_src[pr], q0 = A.pair_q0[pr];
        const uint32_t mask = A.pair_mask[pr];
        const int c0d = A.panel_c0[d], wd = A.panel_c0[d + 1] - c0d;
        const int qq = q0 + __popc(mask & below);
        const bool active = ((mask >> t) & 1u) && qq <= q;
        for (int u = half; u < wd; u += 2) {
            const int64_t base = A.Lp[c0d + u] + (wd - u);  // first entry of R_d in column c0d+u
            const double lq = A.Lx[base + q];
            if (active) acc += lq * A.Lx[base + qq];
        }
    }
    acc += __shfl_xor(acc, 32, 64);
    if (half == 0 && t < ws && t <= pos) A.Lx[A.Lp[c0s + t] + pos - t] -= acc;
}

struct CholFactorArgs {
    const int32_t *item_panel;
    const int32_t *item_block;
    const int32_t *panel_c0;
    const int32_t *panel_rows;
    const int64_t *Lp;
    double *Lx;
    int32_t *status;  // smallest column whose pivot d <= 0 (lib.rs:325)
};

// One wavefront per work item (panel, block).  Three kinds of item, chosen on the host so that no wavefront
// reads a diagonal block while another one overwrites it:
//   block >= 0, trsm_only == 0 : panel with <= 64 rows below the block: factor the w x w diagonal block in LDS,
//                                write it back, then solve this panel's rows (all in one wavefront);
//   block == -1                : panel with more rows: factor the diagonal block and write it back, nothing else;
//   trsm_only == 1 (next launch): load the already factored block and solve rows [64*block, 64*block + 64).
// Row solve: lane = one row below the block,  x <- x * D^-T  (L(k,i) = x_i / L(i,i), lib.rs:310).
__global__ __launch_bounds__(256) void chol_panel_factor(CholFactorArgs A, int64_t item_begin, int64_t nitems, int trsm_only)
{
    __shared__ double tiles[4][32][33];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int64_t it = item_begin + (int64_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wv));
    if (it >= item_begin + nitems) return;
    double(*D)[33] = tiles[wv];
    const int s = A.item_panel[it], blk = A.item_block[it];
    const int c0 = A.panel_c0[s], w = A.panel_c0[s + 1] - c0, r = A.panel_rows[s];

    if (lane < w)
        for (int t = 0; t <= lane; ++t) D[lane][t] = A.Lx[A.Lp[c0 + t] + lane - t];
    __builtin_amdgcn_wave_barrier();
    if (!trsm_only) {
        for (int t = 0; t < w; ++t) {
            const double dtt = D[t][t];
            if (dtt <= 0.0 && lane == 0) atomicMin(A.status, c0 + t);
            const double l = sqrt(dtt);
            __builtin_amdgcn_wave_barrier();
            if (lane == t) D[t][t] = l;
            if (lane > t && lane < w) D[lane][t] /= l;
            __builtin_amdgcn_wave_barrier();
            if (lane > t && lane < w) {
                const double lit = D[lane][t];
                for (int j = t + 1; j <= lane; ++j) D[lane][j] -= lit * D[j][t];
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (lane < w)
            for (int t = 0; t <= lane; ++t) A.Lx[A.Lp[c0 + t] + lane - t] = D[lane][t];
        if (blk < 0) return;
    }

    const int rr = blk * 64 + lane;
    if (rr < r) {
        const int pos = w + rr;
        double x[32];
#pragma unroll
        for (int t = 0; t < 32; ++t) x[t] = (t < w) ? A.Lx[A.Lp[c0 + t] + pos - t] : 0.0;
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            if (t < w) {
                double v = x[t];
#pragma unroll
                for (int u = 0; u < t; ++u) v -= x[u] * D[t][u];
                x[t] = v / D[t][t];
            }
        }
#pragma unroll
        for (int t = 0; t < 32; ++t)
            if (t < w) A.Lx[A.Lp[c0 + t] + pos - t] = x[t];
    }
}

}  // namespace dev
}  // namespace rsp
