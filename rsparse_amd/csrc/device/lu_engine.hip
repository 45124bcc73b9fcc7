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

#include <chrono>
#include <memory>

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

// The factor's values are written by one wavefront and read by others, anywhere on the chip: those accesses are agent-scope
// atomics (they go to the L2 and beyond, no stale line in a compute unit's L1 or another XCD's L2).  The work vector belongs to
// ONE wavefront, whose lanes read and write it in turns: plain loads and stores (one compute unit's L1 is coherent for its own
// accesses), and turn() -- a compiler fence plus s_waitcnt vmcnt(0) -- keeps a turn's accesses behind the previous turn's.
__device__ __forceinline__ double xload(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void xstore(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void turn()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) alone (gfx9 encoding: vmcnt in bits 3:0 and 15:14, expcnt 6:4 and lgkmcnt 11:8 left at "no wait")
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

struct LuControl {
    int next;       // the next column to take
    int first_bad;  // smallest column whose diagonal failed the pivot rule (INT_MAX: none; -1: a wait gave up)
    int stuck_k, stuck_j;  // the wait that gave up: column stuck_k on its dependency stuck_j
};

// persistent: gridDim.x wavefronts (one per workgroup), each with its own zeroed work vector of n doubles (left zeroed);
// done[k] != 0 once column k of L and U is complete in memory
__global__ __launch_bounds__(64) void lu_columns(LuArrays M, double *work, int64_t n, double tol, LuControl *ctl, int *done, unsigned spin_limit)
{
#pragma clang fp contract(off)  // never fused: the reference rounds the product and the difference separately.  (Plain operators under
                                // this pragma, not __dmul_rn/__dsub_rn: those are inline a * b and a - b from a header compiled with
                                // contraction allowed, and came out as one v_fma_f64.)
    const int lane = threadIdx.x;
    double *x = work + (int64_t)blockIdx.x * n;
    for (;;) {
        // the next column, for the whole wavefront.  No `if (lane == 0)` around the atomic: the compiler unswitched this loop on
        // that (loop-invariant) condition and sent lanes 1..63 round a back edge of their own, reading k = 0 for ever; an add
        // of 0 from the other lanes leaves the counter alone, and readfirstlane tells the compiler that k is uniform
        int k = atomicAdd(&ctl->next, lane == 0 ? 1 : 0);
        k = __builtin_amdgcn_readfirstlane(k);
        if (k >= n) break;  // every wavefront gets here: the counter only grows
        const int ac = M.acol[k];
        for (int64_t e = M.Ap[ac] + lane; e < M.Ap[ac + 1]; e += 64) x[M.Arow[e]] = M.Ax[e];  // x = A(:, q[k])  (lib.rs:2346-2348)
        const int ub = M.Up[k], ud = M.Up[k + 1] - 1;  // U(:,k): dependencies [ub, ud), the diagonal at ud
        for (int t0 = ub; t0 < ud; t0 += 64) {         // x = L \ x over the reach, in reach order (lib.rs:2350-2362)
            const int mine = t0 + lane < ud ? M.Ui[t0 + lane] : -1;
            const bool ready = mine < 0 || __hip_atomic_load(done + mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            const unsigned long long ready_mask = __ballot(ready);
            const int cnt = min(64, ud - t0);
            for (int s = 0; s < cnt; ++s) {
                const int j = __builtin_amdgcn_readlane(mine, s);  // uniform, and known to be
                if (!((ready_mask >> s) & 1ull)) {  // column j (< k) is in flight on a wavefront that is running: wait for it
                    // (bounded: were a flag never raised -- a defect -- the wait gives up after seconds, marks the run as failed,
                    // and every other wait sees the mark, so the grid always drains)
                    unsigned spins = 0;
                    while (__hip_atomic_load(done + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                        __builtin_amdgcn_s_sleep(2);
                        if ((++spins & 1023u) == 0) {
                            if (spins >= spin_limit && lane == 0 && atomicMin(&ctl->first_bad, -1) >= 0) {
                                ctl->stuck_k = k;
                                ctl->stuck_j = j;
                            }
                            if (__hip_atomic_load(&ctl->first_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0) break;
                        }
                    }
                }
                turn();
                const double xj = x[j];
                for (int p = M.Lp[j] + 1 + lane; p < M.Lp[j + 1]; p += 64) {
                    const int r = M.Li[p];
                    const double lx = xload(M.Lx + p);
                    const double product = lx * xj;  // rounded on its own, as in the reference
                    x[r] = x[r] - product;
                }
            }
        }
        turn();
        const double pivot = x[k];
        // the pivot rule (lib.rs:566-589): the largest |x| among the rows that are not pivotal yet -- the rows of L(:,k), diagonal
        // included -- must be positive, and the diagonal within tol of it
        double best = fabs(pivot);
        for (int p = M.Lp[k] + 1 + lane; p < M.Lp[k + 1]; p += 64) {
            const double a = fabs(x[M.Li[p]]);
            best = a > best ? a : best;
        }
        for (int off = 32; off > 0; off >>= 1) {
            const double o = __shfl_xor(best, off, 64);
            best = o > best ? o : best;
        }
        const bool ok = best > 0.0 && fabs(pivot) >= best * tol;  // (a NaN anywhere fails one of the two)
        if (!ok && lane == 0) atomicMin(&ctl->first_bad, k);
        for (int t = ub + lane; t < ud; t += 64) {
            M.Ux[t] = x[M.Ui[t]];
            x[M.Ui[t]] = 0.0;  // the vector goes back to zeros for the next column that uses it (lib.rs:607)
        }
        if (lane == 0) {
            M.Ux[ud] = pivot;                // U(k,k) last (lib.rs:593-595)
            xstore(M.Lx + M.Lp[k], 1.0);     // L(k,k) = 1 first (lib.rs:597-599)
            x[k] = 0.0;
        }
        for (int p = M.Lp[k] + 1 + lane; p < M.Lp[k + 1]; p += 64) {
            xstore(M.Lx + p, x[M.Li[p]] / pivot);  // lib.rs:604 (IEEE division)
            x[M.Li[p]] = 0.0;
        }
        turn();  // L(:,k) is in the L2 before the flag is (a failed column raises it too: nobody may wait for ever)
        if (lane == 0) __hip_atomic_store(done + k, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace dev

constexpr double kLuMinUpdates = 1e8;  // below this many multiply-subtracts the host routine finishes as soon as patterns plus device set-up
constexpr int kLuDeclined = 100;       // lu_numeric_device: not applicable or not worth it here -- use sparse_lu

// Device half of one factorisation with static pivots: the patterns of A, L and U resident, the values computed by
// lu_columns.  Kept by the handle, so that a matrix with the SAME pattern and new values is factored again by the kernel
// alone (rsp_lu_refactor_f64: nothing of the depth-first search is repeated).
class LuDevice {
   public:
    // f: patterns from lu_static_pattern (p, i, pinv); A: the matrix they belong to; q: its column order (nullptr: natural)
    void setup(const CscView &A, const i64 *q, const LuFactors &f)
    {
        n_ = A.n;
        nnz_ = A.nnz();
        pattern_ = pattern_hash(A);
        q_.assign((size_t)n_, 0);
        for (i64 k = 0; k < n_; ++k) q_[(size_t)k] = q ? q[k] : k;
        has_q_ = q != nullptr;
        std::vector<int32_t> arow((size_t)nnz_), acol((size_t)n_), Lp((size_t)n_ + 1), Li(f.L.i.size()), Up((size_t)n_ + 1), Ui(f.U.i.size());
        for (i64 e = 0; e < nnz_; ++e) arow[(size_t)e] = (int32_t)f.pinv[(size_t)A.i[e]];
        for (i64 k = 0; k < n_; ++k) acol[(size_t)k] = (int32_t)q_[(size_t)k];
        for (i64 k = 0; k <= n_; ++k) { Lp[(size_t)k] = (int32_t)f.L.p[(size_t)k]; Up[(size_t)k] = (int32_t)f.U.p[(size_t)k]; }
        for (size_t e = 0; e < Li.size(); ++e) Li[e] = (int32_t)f.L.i[e];
        for (size_t e = 0; e < Ui.size(); ++e) Ui[e] = (int32_t)f.U.i[e];
        dAp_.upload(A.p, (size_t)n_ + 1);
        dArow_.upload(arow);
        dAcol_.upload(acol);
        dLp_.upload(Lp); dLi_.upload(Li); dUp_.upload(Up); dUi_.upload(Ui);
        dAx_.resize((size_t)nnz_);
        dLx_.resize(Li.size());
        dUx_.resize(Ui.size());
        // one work vector per wavefront in flight: as many as 4 GiB hold, at most 8 per SIMD of the device (all resident at once)
        hipDeviceProp_t prop;
        int device = 0;
        RSP_HIP(hipGetDevice(&device));
        RSP_HIP(hipGetDeviceProperties(&prop, device));
        slots_ = std::max<i64>(1, std::min<i64>({(i64)prop.multiProcessorCount * 32, n_, ((i64)4 << 30) / (8 * n_)}));
        dWork_.resize((size_t)(slots_ * n_));
        RSP_HIP(hipMemsetAsync(dWork_.get(), 0, (size_t)(slots_ * n_) * sizeof(double), nullptr));  // lu_columns leaves it zeroed
        dDone_.resize((size_t)n_);
        dCtl_.resize(1);
    }
    // values Ax (host, in the order of the pattern given to setup): runs the factorisation.  kOk, or kLuDeclined when some
    // column would not have taken its diagonal as pivot (the values on the device are then meaningless).  Synchronises.
    int factor(const double *Ax, double tol)
    {
        dAx_.upload(Ax, (size_t)nnz_);
        RSP_HIP(hipMemsetAsync(dDone_.get(), 0, (size_t)n_ * sizeof(int), nullptr));
        const dev::LuControl start{0, INT_MAX, -1, -1};
        dCtl_.upload(&start, 1);
        // polls of one dependency's flag before a wait gives up (a poll is about a microsecond: tens of seconds by default)
        const char *spin_env = getenv("RSP_LU_SPIN_LIMIT");
        const unsigned spin_limit = spin_env && *spin_env ? (unsigned)strtoul(spin_env, nullptr, 10) : (1u << 25);
        const dev::LuArrays M{dAp_.get(), dArow_.get(), dAx_.get(), dAcol_.get(), dLp_.get(), dLi_.get(), dLx_.get(), dUp_.get(), dUi_.get(), dUx_.get()};
        hipLaunchKernelGGL(dev::lu_columns, dim3((unsigned)slots_), dim3(64), 0, nullptr, M, dWork_.get(), n_, tol, dCtl_.get(), dDone_.get(), spin_limit);
        RSP_HIP(hipGetLastError());
        dev::LuControl end{};
        dCtl_.download(&end, 1);  // (synchronises)
        if (end.first_bad < 0)
            throw DeviceError("lu_columns: column " + std::to_string(end.stuck_k) + " gave up waiting for column " + std::to_string(end.stuck_j) +
                              " (next " + std::to_string(end.next) + ")");
        return end.first_bad != INT_MAX ? kLuDeclined : kOk;
    }
    void download(LuFactors &f) const
    {
        dLx_.download(f.L.x.data(), f.L.x.size());
        dUx_.download(f.U.x.data(), f.U.x.size());
    }
    bool same_pattern(const CscView &A) const { return A.n == n_ && A.m == n_ && A.nnz() == nnz_ && pattern_hash(A) == pattern_; }
    const i64 *q() const { return has_q_ ? q_.data() : nullptr; }
    i64 wavefronts() const { return slots_; }
    i64 device_bytes() const
    {
        return (i64)(dAp_.bytes() + dArow_.bytes() + dAcol_.bytes() + dLp_.bytes() + dLi_.bytes() + dUp_.bytes() + dUi_.bytes() + dAx_.bytes() +
                     dLx_.bytes() + dUx_.bytes() + dWork_.bytes() + dDone_.bytes());
    }

   private:
    i64 n_ = 0, nnz_ = 0, slots_ = 0;
    uint64_t pattern_ = 0;
    std::vector<i64> q_;
    bool has_q_ = false;
    DeviceBuffer<int64_t> dAp_;
    DeviceBuffer<int32_t> dArow_, dAcol_, dLp_, dLi_, dUp_, dUi_;
    DeviceBuffer<double> dAx_, dLx_, dUx_, dWork_;
    DeviceBuffer<int> dDone_;
    DeviceBuffer<dev::LuControl> dCtl_;
};

// kOk: out = the factors (bit-identical to sparse_lu's), *keep = the device half when keep != nullptr; kLuDeclined: see above;
// throws DeviceError.
// mode: 0 never, 1 always when applicable, anything else: when there is enough numeric work to pay for the device set-up
int lu_numeric_device(const CscView &A, const i64 *q, double tol, LuFactors &out, int mode, std::unique_ptr<LuDevice> *keep)
{
    const i64 n = A.n;
    if (mode == 0 || n == 0 || !(tol <= 1.0)) return kLuDeclined;  // (tol > 1 can make the reference refuse a diagonal that is the maximum)
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1) return kLuDeclined;
    if (A.nnz() >= INT32_MAX) return kLuDeclined;
    std::vector<int32_t> level;
    LuFactors f;
    const bool trace = getenv("RSP_LU_TRACE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = now();
    if (!lu_static_pattern(A, q, f, level)) return kLuDeclined;
    const auto t1 = now();
    // the numeric work the host would do: one update per entry of L(:,j) per dependency j of a column
    double updates = 0;
    for (i64 k = 0; k < n; ++k)
        for (i64 t = f.U.p[(size_t)k]; t + 1 < f.U.p[(size_t)k + 1]; ++t) updates += (double)(f.L.p[(size_t)f.U.i[(size_t)t] + 1] - f.L.p[(size_t)f.U.i[(size_t)t]] - 1);
    if (mode != 1 && updates < kLuMinUpdates) return kLuDeclined;
    std::unique_ptr<LuDevice> D(new LuDevice);
    D->setup(A, q, f);
    const auto t2 = now();
    const int st = D->factor(A.x, tol);
    const auto t3 = now();
    if (st != kOk) return st;
    D->download(f);
    if (trace)
        fprintf(stderr, "[rsp lu] n %lld updates %.3g: patterns (host) %.1f ms, set-up and upload %.1f ms, lu_columns on %lld wavefronts %.1f ms, download %.1f ms\n",
                (long long)n, updates, ms(t0, t1), ms(t1, t2), (long long)D->wavefronts(), ms(t2, t3), ms(t3, now()));
    out = std::move(f);
    if (keep) *keep = std::move(D);
    return kOk;
}

}  // namespace rsp
