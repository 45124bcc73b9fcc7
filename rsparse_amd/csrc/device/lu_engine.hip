// GPU numeric LU with STATIC (diagonal) pivots, verified column by column.
//
// rsparse::lu (/root/reference/src/lib.rs:519-622) is a left-looking LU with threshold partial pivoting: column k is
// x = L \ A(:, q[k]) over the depth-first reach of the column, then the pivot is the diagonal row q[k] whenever
// |x[q[k]]| >= tol * max|x| over the rows that are not pivotal yet (lib.rs:585-589).  When that holds for EVERY column --
// diagonally dominant matrices, or a small tol -- the pivot sequence is known before any value is: pinv[q[k]] = k, and the
// patterns of L and U in the reference's own storage order follow from integer work alone (host/lu_static.cpp).
// This engine computes the values on those patterns, left-looking like the reference, one wavefront per column, WITHOUT
// level launches: one persistent launch, wavefronts take the columns in ascending order from a counter and wait on a
// per-column "done" flag for each dependency that is still in flight (ascending order makes that deadlock-free: a column
// only ever waits for smaller columns, and those were taken earlier by wavefronts that are running).  Deep dependency chains
// therefore cost a flag hand-over per link instead of a launch, and a column's updates from finished columns overlap with the
// columns still in flight.  A column applies its dependencies in the stored order of U(:,k) (the reach order the reference
// uses), each update x[i] -= L(i,j) * x[j] as a separate multiply and subtract (the reference never fuses them), so every
// entry receives the SAME operations in the SAME order: the factors are bit-identical to rsparse::lu's whenever the
// assumption holds.  Every column checks it (best > 0, |pivot| >= tol * best, lib.rs:584-589) and reports the smallest column
// that breaks it; the caller then discards everything and runs host/lu.cpp.
// What stays on the host is the depth-first search: it costs as many integer steps as the numeric part costs flops (the
// reach of column k walks every entry of every L column it touches), so a one-shot lu can gain at most about a factor two.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rsp {
namespace dev {

struct LuArrays {
    const int64_t *Ap;     // column pointers of A
    const int32_t *Arow;   // pinv[A.i]: rows of A in pivot order
    const double *Ax;
    const int32_t *acol;   // q[k]: the column of A behind column k of the factors
    const int32_t *Lp, *Li;
    double *Lx;
    const int32_t *Up, *Ui;
    double *Ux;
};

// The work vector is read and written by different lanes of one wavefront in turns, and the factor's values by different
// wavefronts: those accesses go to the L2 (agent-scope atomics: no stale line in a compute unit's L1), and turn() waits until
// the wavefront's earlier loads and stores have completed there before the next turn issues its own.
__device__ __forceinline__ double xload(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void xstore(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void turn()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) alone (gfx9 encoding: vmcnt in bits 3:0 and 15:14, expcnt 6:4 and lgkmcnt 11:8 left at "no wait")
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

constexpr unsigned kLuSpinLimit = 1u << 26;  // polls of a dependency's flag before a wait gives up (tens of seconds)

struct LuControl {
    int next;       // the next column to take
    int first_bad;  // smallest column whose diagonal failed the pivot rule (INT_MAX: none)
};

// persistent: gridDim.x wavefronts (one per workgroup), each with its own zeroed work vector of n doubles (left zeroed);
// done[k] != 0 once column k of L and U is complete in memory
__global__ __launch_bounds__(64) void lu_columns(LuArrays M, double *work, int64_t n, double tol, LuControl *ctl, int *done)
{
    const int lane = threadIdx.x;
    double *x = work + (int64_t)blockIdx.x * n;
    for (;;) {
        int k = 0;
        if (lane == 0) k = atomicAdd(&ctl->next, 1);
        k = __shfl(k, 0, 64);
        if (k >= n) break;  // every wavefront gets here: the counter only grows
        const int ac = M.acol[k];
        for (int64_t e = M.Ap[ac] + lane; e < M.Ap[ac + 1]; e += 64) xstore(x + M.Arow[e], M.Ax[e]);  // x = A(:, q[k])  (lib.rs:2346-2348)
        const int ub = M.Up[k], ud = M.Up[k + 1] - 1;  // U(:,k): dependencies [ub, ud), the diagonal at ud
        for (int t0 = ub; t0 < ud; t0 += 64) {         // x = L \ x over the reach, in reach order (lib.rs:2350-2362)
            const int mine = t0 + lane < ud ? M.Ui[t0 + lane] : -1;
            const bool ready = mine < 0 || __hip_atomic_load(done + mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            const unsigned long long ready_mask = __ballot(ready);
            const int cnt = min(64, ud - t0);
            for (int s = 0; s < cnt; ++s) {
                const int j = __shfl(mine, s, 64);
                if (!((ready_mask >> s) & 1ull)) {  // column j (< k) is in flight on a wavefront that is running: wait for it
                    // (bounded: were a flag never raised -- a defect -- the wait gives up after seconds, marks the run as failed,
                    // and every other wait sees the mark, so the grid always drains)
                    unsigned spins = 0;
                    while (__hip_atomic_load(done + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                        __builtin_amdgcn_s_sleep(2);
                        if ((++spins & 1023u) == 0) {
                            if (spins >= kLuSpinLimit && lane == 0) atomicMin(&ctl->first_bad, -1);
                            if (__hip_atomic_load(&ctl->first_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0) break;
                        }
                    }
                }
                turn();
                const double xj = xload(x + j);
                for (int p = M.Lp[j] + 1 + lane; p < M.Lp[j + 1]; p += 64) {
                    const int r = M.Li[p];
                    xstore(x + r, __dsub_rn(xload(x + r), __dmul_rn(xload(M.Lx + p), xj)));  // two roundings, as in the reference (no fused multiply-add)
                }
            }
        }
        turn();
        const double pivot = xload(x + k);
        // the pivot rule (lib.rs:566-589): the largest |x| among the rows that are not pivotal yet -- the rows of L(:,k), diagonal
        // included -- must be positive, and the diagonal within tol of it
        double best = fabs(pivot);
        for (int p = M.Lp[k] + 1 + lane; p < M.Lp[k + 1]; p += 64) {
            const double a = fabs(xload(x + M.Li[p]));
            best = a > best ? a : best;
        }
        for (int off = 32; off > 0; off >>= 1) {
            const double o = __shfl_xor(best, off, 64);
            best = o > best ? o : best;
        }
        const bool ok = best > 0.0 && fabs(pivot) >= best * tol;  // (a NaN anywhere fails one of the two)
        if (!ok && lane == 0) atomicMin(&ctl->first_bad, k);
        for (int t = ub + lane; t < ud; t += 64) {
            M.Ux[t] = xload(x + M.Ui[t]);
            xstore(x + M.Ui[t], 0.0);  // the vector goes back to zeros for the next column that uses it (lib.rs:607)
        }
        if (lane == 0) {
            M.Ux[ud] = pivot;                // U(k,k) last (lib.rs:593-595)
            xstore(M.Lx + M.Lp[k], 1.0);     // L(k,k) = 1 first (lib.rs:597-599)
            xstore(x + k, 0.0);
        }
        for (int p = M.Lp[k] + 1 + lane; p < M.Lp[k + 1]; p += 64) {
            xstore(M.Lx + p, __ddiv_rn(xload(x + M.Li[p]), pivot));  // lib.rs:604
            xstore(x + M.Li[p], 0.0);
        }
        turn();  // L(:,k) is in the L2 before the flag is (a failed column raises it too: nobody may wait for ever)
        if (lane == 0) __hip_atomic_store(done + k, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace dev

constexpr double kLuMinUpdates = 2e6;  // below this many multiply-subtracts the host routine is done in a few milliseconds
constexpr int kLuDeclined = 100;  // lu_numeric_device: not applicable or not worth it here -- use sparse_lu

// kOk: out = the factors (bit-identical to sparse_lu's); kLuDeclined: see above; throws DeviceError.
// mode: 0 never, 1 always when applicable, anything else: when there is enough numeric work to pay for the device set-up
int lu_numeric_device(const CscView &A, const i64 *q, double tol, LuFactors &out, int mode)
{
    const i64 n = A.n;
    if (mode == 0 || n == 0 || !(tol <= 1.0)) return kLuDeclined;  // (tol > 1 can make the reference refuse a diagonal that is the maximum)
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1) return kLuDeclined;
    std::vector<int32_t> level;
    LuFactors f;
    if (!lu_static_pattern(A, q, f, level)) return kLuDeclined;
    // the numeric work the host would do: one update per entry of L(:,j) per dependency j of a column
    double updates = 0;
    for (i64 k = 0; k < n; ++k)
        for (i64 t = f.U.p[(size_t)k]; t + 1 < f.U.p[(size_t)k + 1]; ++t) updates += (double)(f.L.p[(size_t)f.U.i[(size_t)t] + 1] - f.L.p[(size_t)f.U.i[(size_t)t]] - 1);
    if (mode != 1 && updates < kLuMinUpdates) return kLuDeclined;  // the host finishes sooner than the device set-up costs
    if (A.nnz() >= INT32_MAX) return kLuDeclined;
    std::vector<int32_t> arow((size_t)A.nnz()), acol((size_t)n), Lp((size_t)n + 1), Li(f.L.i.size()), Up((size_t)n + 1), Ui(f.U.i.size());
    for (i64 e = 0; e < A.nnz(); ++e) arow[(size_t)e] = (int32_t)f.pinv[(size_t)A.i[e]];
    for (i64 k = 0; k < n; ++k) acol[(size_t)k] = (int32_t)(q ? q[k] : k);
    for (i64 k = 0; k <= n; ++k) { Lp[(size_t)k] = (int32_t)f.L.p[(size_t)k]; Up[(size_t)k] = (int32_t)f.U.p[(size_t)k]; }
    for (size_t e = 0; e < Li.size(); ++e) Li[e] = (int32_t)f.L.i[e];
    for (size_t e = 0; e < Ui.size(); ++e) Ui[e] = (int32_t)f.U.i[e];
    DeviceBuffer<int64_t> dAp;
    DeviceBuffer<int32_t> dArow, dAcol, dLp, dLi, dUp, dUi;
    DeviceBuffer<double> dAx, dLx, dUx, dWork;
    DeviceBuffer<int> dDone;
    DeviceBuffer<dev::LuControl> dCtl;
    dAp.upload(A.p, (size_t)n + 1);
    dArow.upload(arow);
    dAx.upload(A.x, (size_t)A.nnz());
    dAcol.upload(acol);
    dLp.upload(Lp); dLi.upload(Li); dUp.upload(Up); dUi.upload(Ui);
    dLx.resize(Li.size());
    dUx.resize(Ui.size());
    // one work vector per wavefront in flight: as many as 4 GiB hold, at most 8 per SIMD of the device (all resident at once)
    hipDeviceProp_t prop;
    int device = 0;
    RSP_HIP(hipGetDevice(&device));
    RSP_HIP(hipGetDeviceProperties(&prop, device));
    const i64 resident = (i64)prop.multiProcessorCount * 32;
    const i64 slots = std::max<i64>(1, std::min<i64>({resident, n, ((i64)4 << 30) / (8 * n)}));
    dWork.resize((size_t)(slots * n));
    RSP_HIP(hipMemsetAsync(dWork.get(), 0, (size_t)(slots * n) * sizeof(double), nullptr));
    dDone.resize((size_t)n);
    RSP_HIP(hipMemsetAsync(dDone.get(), 0, (size_t)n * sizeof(int), nullptr));
    const dev::LuControl start{0, INT_MAX};
    dCtl.upload(&start, 1);
    const dev::LuArrays M{dAp.get(), dArow.get(), dAx.get(), dAcol.get(), dLp.get(), dLi.get(), dLx.get(), dUp.get(), dUi.get(), dUx.get()};
    hipLaunchKernelGGL(dev::lu_columns, dim3((unsigned)slots), dim3(64), 0, nullptr, M, dWork.get(), n, tol, dCtl.get(), dDone.get());
    RSP_HIP(hipGetLastError());
    dev::LuControl end{};
    dCtl.download(&end, 1);  // (synchronises)
    if (end.first_bad < 0) throw DeviceError("lu_columns: a dependency flag was never raised");
    if (end.first_bad != INT_MAX) return kLuDeclined;  // some column would not have taken its diagonal: the pivot sequence is not this one
    dLx.download(f.L.x.data(), f.L.x.size());
    dUx.download(f.U.x.data(), f.U.x.size());
    out = std::move(f);
    return kOk;
}

}  // namespace rsp
