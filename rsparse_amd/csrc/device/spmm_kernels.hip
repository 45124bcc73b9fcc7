// HIP kernels (gfx950) of Y = A * X + beta * Y for a sparse A held row-wise (CSR) and dense blocks X, Y: the reference's
// gaxpy (lib.rs:411-421: y + A x, one vector) widened to blocks of right-hand sides, so that residuals of a whole solved
// block are checked at memory speed on the device instead of one column at a time on the host.
// Row-major blocks: one wavefront per (row of A, 64 columns); every entry of the row contributes one coalesced 512-byte
// piece of a row of X.  Column-major blocks (the reference's layout, normally one vector): one lane per (row, column).
// Entries of a row are applied in the order of the CSC columns they came from (ascending column), which is the order
// the reference's loop adds them in for that row: the result is bit-identical to the oracle's for beta = 1.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rsp {
namespace dev {

struct CsrArrays {
    const int64_t *ptr;  // rows + 1
    const int32_t *col;
    const double *val;
};

__global__ __launch_bounds__(256) void gaxpy_rowmajor(CsrArrays A, int64_t rows, const double *__restrict__ X, int64_t ldx, double *__restrict__ Y,
                                                      int64_t ldy, int nrhs, int nchunks, double beta)
{
#pragma clang fp contract(off)  // never fused: the reference rounds the product and the sum separately
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t r = item / nchunks;
    if (r >= rows) return;
    const int c = (int)(item - r * nchunks) * 64 + lane;
    if (c >= nrhs) return;
    double acc = beta == 0.0 ? 0.0 : beta * Y[r * ldy + c];
    const int64_t e0 = A.ptr[r], e1 = A.ptr[r + 1];
    for (int64_t e = e0; e < e1; ++e) acc += A.val[e] * X[(int64_t)A.col[e] * ldx + c];
    Y[r * ldy + c] = acc;
}

__global__ __launch_bounds__(256) void gaxpy_colmajor(CsrArrays A, int64_t rows, const double *__restrict__ X, int64_t ldx, double *__restrict__ Y,
                                                      int64_t ldy, int nrhs, double beta)
{
#pragma clang fp contract(off)
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    if (r >= rows || c >= nrhs) return;
    double acc = beta == 0.0 ? 0.0 : beta * Y[r + c * ldy];
    const int64_t e0 = A.ptr[r], e1 = A.ptr[r + 1];
    const double *x = X + c * ldx;
    for (int64_t e = e0; e < e1; ++e) acc += A.val[e] * x[A.col[e]];
    Y[r + c * ldy] = acc;
}

// dst = a * src + b * dst on a rows x cols block whose contiguous dimension is cols (b == 0: dst is only written).  The
// glue of iterative refinement: W = B before the residual, X = X - W after the correction solve.
__global__ __launch_bounds__(256) void block_axpby(double *__restrict__ dst, int64_t ld_dst, const double *__restrict__ src, int64_t ld_src,
                                                   int64_t rows, int64_t cols, double a, double b)
{
#pragma clang fp contract(off)
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols, c = i - r * cols;
        const double v = a * src[r * ld_src + c];
        dst[r * ld_dst + c] = b == 0.0 ? v : v + b * dst[r * ld_dst + c];
    }
}

}  // namespace dev
}  // namespace rsp
