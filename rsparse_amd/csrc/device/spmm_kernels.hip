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

}  // namespace dev
}  // namespace rsp
