// HIP kernels (gfx950) that pour NEW VALUES of a factor into a resident multi-right-hand-side programme without the
// host: the device half of BlockProgram::Refill (host/sparse.hpp).  The reference has no counterpart: chol(a, &mut s)
// called again (lib.rs:278) simply produces a new factor that lsolve / ltsolve read directly; here the solves read a
// programme derived from the factor (regrouped coefficients, inverted triangles), and after a refactorisation that
// programme follows the factor at memory speed -- nothing crosses PCIe.
//
// A source CODE names where a value of the programme comes from: code < split ( = nnz + n of the gather form):
// Tx[map[code]] (an off-diagonal entry or a diagonal of the factor; map composes the gather form's order with the
// factor's storage order), code >= split: -pool[code - split] (an entry of a chain inverse), code < 0: a constant that
// stays as the host wrote it.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../host/sparse.hpp"

namespace rsp {
namespace dev {

struct RefillSource {
    const int32_t *map;  // split entries
    const double *Tx;    // the factor's values on the device
    const double *pool;  // chain inverses
    int32_t split, nnz;  // nnz: codes [nnz, split) are diagonals
    __device__ __forceinline__ double value(int32_t code) const { return code < split ? Tx[map[code]] : -pool[code - split]; }
};

__global__ __launch_bounds__(256) void refill_coef(double *__restrict__ coef, const int32_t *__restrict__ coef_src, int64_t count, RefillSource S)
{
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < count; e += (int64_t)gridDim.x * 256) {
        const int32_t c = coef_src[e];
        if (c >= 0) coef[e] = S.value(c);
    }
}

// one thread per group record: its up to four coefficients are consecutive entries of the gather form
__global__ __launch_bounds__(256) void refill_groups(GroupRec *__restrict__ groups, const int32_t *__restrict__ first_entry,
                                                     const int32_t *__restrict__ row, int64_t count, RefillSource S)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= count) return;
    const int cnt = groups[g].flags & 7;
    const int32_t base = first_entry[g];
#pragma unroll
    for (int e = 0; e < 4; ++e) groups[g].coef[e] = e < cnt ? S.Tx[S.map[base + e]] : 0.0;
    groups[g].rdiag = 1.0 / S.Tx[S.map[S.nnz + row[g]]];
}

// 16 x 16 block triangles: 16 lanes per record, lane c computes column c of the inverse by forward substitution and
// stores it in the A-operand order of v_mfma_f64_16x16x4_f64 (K-step k, lane l holds Tinv[l & 15][4k + (l >> 4)])
__global__ __launch_bounds__(256) void refill_tri16(double *__restrict__ coef, const BlockProgram::Refill::Tri16 *__restrict__ recs, int64_t count,
                                                    RefillSource S)
{
    __shared__ double Ts[16][16][17];  // [record of the workgroup][row][column]
    const int sub = threadIdx.x >> 4, c = threadIdx.x & 15;
    const int64_t k = (int64_t)blockIdx.x * 16 + sub;
    const bool live = k < count;
    const BlockProgram::Refill::Tri16 *R = recs + (live ? k : 0);
    double(*T)[17] = Ts[sub];
    // lane c loads column c of the triangle (rows c..15)
    for (int r = 0; r < 16; ++r) {
        double v = r == c ? 1.0 : 0.0;
        if (live && r >= c && r < R->rows) {
            const int32_t code = R->src[r * (r + 1) / 2 + c];
            v = code >= 0 ? S.value(code) : 0.0;
        }
        T[r][c] = v;
    }
    __syncthreads();
    if (!live) return;
    double z[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (r == c) z[r] = 1.0 / T[r][r];
        else if (r > c) {
            double sum = 0.0;
#pragma unroll
            for (int t = 0; t < 16; ++t)
                if (t >= c && t < r) sum += T[r][t] * z[t];
            z[r] = -sum / T[r][r];
        }
    }
    double *out = coef + R->coef_base;
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(c >> 2) * 64 + r + 16 * (c & 3)] = z[r];
}

// chain matrices: M (zeroed before) receives the entries that fall inside a chain's diagonal block
__global__ __launch_bounds__(256) void chain_fill(double *__restrict__ M, const int32_t *__restrict__ pos, const int32_t *__restrict__ code,
                                                  int64_t count, RefillSource S)
{
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < count; e += (int64_t)gridDim.x * 256) M[pos[e]] = S.Tx[S.map[code[e]]];
}

struct ChainDesc {
    int64_t pool_off;
    int32_t w, first_tile;  // column tiles of 64 of all chains are numbered consecutively
};

// Inverse of every chain's lower triangle: one wavefront per (chain, 64 columns), lane = column c of the inverse, which
// is an independent forward substitution  M z = e_c.  Rows go by blocks of 64: the part of the sum that comes from
// earlier row blocks is accumulated in 64 registers per lane (one coalesced load of z per t, the entries of M are the
// same address in every lane), then the block's own 64 x 64 triangle is solved in registers.  (The first version, one
// row and one term at a time, paid a memory round trip per term: 136 ms for the chains of the C2 factor, 34 ms with
// eight terms in flight.)
// diag_only: only the tile's own 64 x 64 diagonal block is inverted (the blocked inverse below builds the rest from products).
__global__ __launch_bounds__(64) void chain_invert(const double *__restrict__ Mpool, double *__restrict__ pool, const ChainDesc *__restrict__ chains,
                                                   int nchains, int diag_only)
{
    int lo = 0, hi = nchains;  // chain of this tile
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (chains[mid].first_tile <= (int)blockIdx.x) lo = mid;
        else hi = mid;
    }
    const ChainDesc C = chains[lo];
    const int w = C.w, c0 = ((int)blockIdx.x - C.first_tile) * 64, cl = (int)threadIdx.x, c = c0 + cl;
    const double *M = Mpool + C.pool_off;
    double *Z = pool + C.pool_off;
    const bool live = c < w;
    const int cc = min(c, w - 1);
    __shared__ double stage[64][8];
    const int wend = diag_only ? min(w, c0 + 64) : w;
    for (int I0 = c0; I0 < wend; I0 += 64) {
        const int rows = min(64, w - I0);  // wave-uniform
        double acc[64];
#pragma unroll
        for (int r = 0; r < 64; ++r) acc[r] = 0.0;
        // steps of eight columns of M: lane r stages its row's eight entries in LDS (the next step's loads are in flight
        // while this one is multiplied), every lane then reads all 64 x 8 of them as broadcasts
        const double *mrow = M + (int64_t)(I0 + min(cl, rows - 1)) * w;
        double mnext[8], znext[8];
        if (c0 < I0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                mnext[k] = mrow[c0 + k];
                znext[k] = Z[(int64_t)(c0 + k) * w + cc];  // rows above the diagonal hold zeros
            }
        }
        for (int t0 = c0; t0 < I0; t0 += 8) {  // c0 and I0 are multiples of 64: whole steps of eight
            double z[8];
            __builtin_amdgcn_wave_barrier();  // (one wavefront: the previous step's reads are done)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                stage[cl][k] = mnext[k];
                z[k] = znext[k];
            }
            if (t0 + 8 < I0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    mnext[k] = mrow[t0 + 8 + k];
                    znext[k] = Z[(int64_t)(t0 + 8 + k) * w + cc];
                }
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the staged entries are in LDS
#pragma unroll
            for (int r = 0; r < 64; ++r)
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[r] += stage[r][k] * z[k];  // rows beyond `rows` repeat the last one: unused
        }
        double z[64];
#pragma unroll
        for (int r = 0; r < 64; ++r) {
            z[r] = 0.0;
            if (r < rows) {
                const double *mr = M + (int64_t)(I0 + r) * w + I0;
                double sum = I0 == c0 ? (r == cl ? 1.0 : 0.0) : -acc[r];
#pragma unroll
                for (int t = 0; t < r; ++t) sum -= mr[t] * z[t];
                z[r] = sum / mr[r];
            }
        }
        if (live) {
#pragma unroll
            for (int r = 0; r < 64; ++r)
                if (r < rows) Z[(int64_t)(I0 + r) * w + c] = z[r];
        }
        __builtin_amdgcn_s_waitcnt(0);  // the rows just written are read back by the next row blocks
    }
}

// ---- blocked inverse of a chain's triangle on the matrix cores ---------------------------------------------------------
// M = [A 0; C B]  =>  M^-1 = [A^-1 0; -B^-1 C A^-1, B^-1]: the 64 x 64 diagonal blocks are inverted by chain_invert(diag_only),
// then blocks are paired level by level (1 + 1, 2 + 2, 4 + 4 ... blocks of 64), each level two launches of 64 x 64 output
// tiles: T = C Z11 (mode 1: left operand from M, right from the inverse so far, into a scratch pool) and Z21 = -Z22 T
// (mode 2: left from the inverse, right from the scratch pool, into the inverse).  The host lists the tiles of every level
// once per programme (ChainTile).  One wavefront per tile: 4 x 4 accumulators of v_mfma_f64_16x16x4_f64, K restricted to
// where the triangular operand is not zero.  (One wavefront per 64 columns walking all row blocks in turn -- chain_invert
// without diag_only -- took 3.7 ms per triangle at a 1024^2 Laplacian, the 1024-wide root separator's first tile alone.)
struct ChainTile {
    int64_t pool_off;
    int32_t w, r0, c0, k0, k1, pad;
};

typedef double rf_f64x4 __attribute__((ext_vector_type(4)));

// (no __restrict__ on the pools: mode 2 reads and writes the inverse -- disjoint blocks of it)
__global__ __launch_bounds__(64) void chain_gemm(const double *Lpool, const double *Rpool, double *Opool, const ChainTile *__restrict__ tiles, int negate)
{
    const ChainTile T = tiles[blockIdx.x];
    const int lane = threadIdx.x, m = lane & 15, g = lane >> 4, w = T.w;
    const double *L = Lpool + T.pool_off, *R = Rpool + T.pool_off;
    double *O = Opool + T.pool_off;
    rf_f64x4 acc[4][4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = rf_f64x4{0.0, 0.0, 0.0, 0.0};
    // rows and columns beyond the chain (w is any number >= 25) read a valid address and contribute zeros
    int64_t lrow[4];
    bool lin[4];
    int rcol[4];
    bool rin[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int r = T.r0 + 16 * t + m, c = T.c0 + 16 * t + m;
        lin[t] = r < w;
        lrow[t] = (int64_t)min(r, w - 1) * w;
        rin[t] = c < w;
        rcol[t] = min(c, w - 1);
    }
    // four K-steps per pass: their 32 loads are in flight together (one K-step per round trip made a 512-long product 200 us)
    for (int k = T.k0; k < T.k1; k += 16) {
        double a[4][4], b[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kk = k + 4 * u + g;
            const bool kin = kk < T.k1;
            const int kc = min(kk, w - 1);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double av = L[lrow[t] + kc], bv = R[(int64_t)kc * w + rcol[t]];
                a[u][t] = kin && lin[t] ? av : 0.0;
                b[u][t] = kin && rin[t] ? bv : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][rt], b[u][ct], acc[rt][ct], 0, 0, 0);
    }
    // element s of a lane is row 4 s + g, column m of its 16 x 16 tile
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int r = T.r0 + 16 * rt + 4 * s + g;
            if (r < w) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    const int c = T.c0 + 16 * ct + m;
                    if (c < w) O[(int64_t)r * w + c] = negate ? -acc[rt][ct][s] : acc[rt][ct][s];
                }
            }
        }
}

}  // namespace dev
}  // namespace rsp
