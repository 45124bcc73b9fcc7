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
#include <cmath>
#include <memory>

namespace rsp {
namespace dev {

struct LuArrays {
    const int64_t *Ap;     // column pointers of A
    const int32_t *Arow;   // pinv[A.i]: rows of A in pivot order
    const double *Ax;
    const int32_t *acol;   // q[k]: the column of A behind column k of the factors
    const int32_t *order;  // the columns sorted by dependency level: the order in which wavefronts take them
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

#ifdef RSP_LU_CLOCKS
// -DRSP_LU_CLOCKS: wall-clock ticks (10 ns) of the phases of ONE column in the middle of the matrix, to see where a link of the
// dependency chain spends its time
__device__ unsigned long long g_lu_clk[16];
#define LUCLK(i) \
    if (k == (int)(n / 2) && lane == 0) g_lu_clk[i] = wall_clock64()
#else
#define LUCLK(i)
#endif

struct LuControl {
    int next;       // the next column to take
    int first_bad;  // smallest column whose diagonal failed the pivot rule (INT_MAX: none; -1: a wait gave up)
    int stuck_k, stuck_j;  // the wait that gave up: column stuck_k on its dependency stuck_j
};

// persistent: gridDim.x wavefronts (one per workgroup), each with its own zeroed work vector of n doubles (left zeroed);
// ctl->next counts POSITIONS in M.order;
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
        int t = atomicAdd(&ctl->next, lane == 0 ? 1 : 0);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= n) break;  // every wavefront gets here: the counter only grows
        // Columns are taken by dependency level, not by number: in a postordered factor column k mostly depends on k - 1, so
        // consecutive columns in flight were one chain and most wavefronts waited (147 ms at C4; by level: see DESIGN).  A
        // column's dependencies have lower levels, hence earlier positions: they were taken earlier, by running wavefronts.
        const int k = __builtin_amdgcn_readfirstlane(M.order[t]);
        LUCLK(0);
        const int ac = M.acol[k];
        for (int64_t e = M.Ap[ac] + lane; e < M.Ap[ac + 1]; e += 64) x[M.Arow[e]] = M.Ax[e];  // x = A(:, q[k])  (lib.rs:2346-2348)
        const int ub = M.Up[k], ud = M.Up[k + 1] - 1;  // U(:,k): dependencies [ub, ud), the diagonal at ud
        // the rows of L(:,k) behind the diagonal: [lp0, lp1); the first 128 row numbers are fetched now, for the end of the column
        const int lp0 = M.Lp[k] + 1, lp1 = M.Lp[k + 1];
        int my_r[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) my_r[c] = lp0 + 64 * c + lane < lp1 ? M.Li[lp0 + 64 * c + lane] : 0;
        for (int t0 = ub; t0 < ud; t0 += 64) {         // x = L \ x over the reach, in reach order (lib.rs:2350-2362)
            // 64 dependencies at a time: their flags and the extents of their columns of L are fetched by the lanes at once (one
            // round trip for the lot, where `Lp[j]` per dependency was a round trip of its own to cold memory)
            const int mine = t0 + lane < ud ? M.Ui[t0 + lane] : -1;
            const bool ready = mine < 0 || __hip_atomic_load(done + mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            const int my_p0 = mine >= 0 ? M.Lp[mine] + 1 : 0, my_p1 = mine >= 0 ? M.Lp[mine + 1] : 0;
            const unsigned long long ready_mask = __ballot(ready);
            const int cnt = min(64, ud - t0);
            // The first 64 entries of the NEXT dependency's column (row index and value) travel while the current one is
            // applied -- when that column was complete at the batch's start; one still in flight is fetched after its wait.
            int nr = 0;
            double nlx = 0.0;
            bool fetched = false;  // (uniform) nr / nlx hold the first piece of dependency s
            for (int s = 0; s < cnt; ++s) {
                const int j = __builtin_amdgcn_readlane(mine, s);  // uniform, and known to be
                const int p0 = __builtin_amdgcn_readlane(my_p0, s), p1 = __builtin_amdgcn_readlane(my_p1, s);
                if (!((ready_mask >> s) & 1ull)) {  // column j (< k) is in flight on a wavefront that is running: wait for it
                    // (bounded: were a flag never raised -- a defect -- the wait gives up after seconds, marks the run as failed,
                    // and every other wait sees the mark, so the grid always drains)
                    unsigned spins = 0;
                    LUCLK(1);  // (the last wait that starts)
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
                const bool in0 = p0 + lane < p1;
                int r0 = nr;
                double lx0 = nlx;
                if (!fetched) {  // (uniform)
                    r0 = in0 ? M.Li[p0 + lane] : 0;
                    lx0 = in0 ? xload(M.Lx + p0 + lane) : 0.0;
                }
                LUCLK(2);  // (the last dependency's turn starts)
                turn();
                const double xj = x[j];
                const double x0 = in0 ? x[r0] : 0.0;
                __builtin_amdgcn_sched_barrier(0);  // (the fetch below is issued BEHIND these loads: loads return in order)
                fetched = s + 1 < cnt && ((ready_mask >> (s + 1)) & 1ull);
                if (fetched) {
                    const int q0 = __builtin_amdgcn_readlane(my_p0, s + 1), q1 = __builtin_amdgcn_readlane(my_p1, s + 1);
                    const bool inn = q0 + lane < q1;
                    nr = inn ? M.Li[q0 + lane] : 0;
                    nlx = inn ? xload(M.Lx + q0 + lane) : 0.0;
                }
                __builtin_amdgcn_sched_barrier(0);
                if (in0) {
                    const double product = lx0 * xj;  // rounded on its own, as in the reference
                    x[r0] = x0 - product;
                }
                for (int p = p0 + 64 + lane; p < p1; p += 64) {
                    const int r = M.Li[p];
                    const double lx = xload(M.Lx + p);
                    const double product = lx * xj;
                    x[r] = x[r] - product;
                }
            }
        }
        LUCLK(3);
        turn();
        // What the columns waiting for this one need is L(:,k) and the flag: those first; U(:,k)'s values and the clean-up of
        // the work vector follow once the flag is up.
        const double pivot = x[k];
        double v[2];
        bool in[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            in[c] = lp0 + 64 * c + lane < lp1;
            v[c] = in[c] ? x[my_r[c]] : 0.0;
        }
        // the pivot rule (lib.rs:566-589): the largest |x| among the rows that are not pivotal yet -- the rows of L(:,k), diagonal
        // included -- must be positive, and the diagonal within tol of it
        double best = fabs(pivot);
#pragma unroll
        for (int c = 0; c < 2; ++c) best = fabs(v[c]) > best ? fabs(v[c]) : best;
        for (int p = lp0 + 128 + lane; p < lp1; p += 64) {
            const double a = fabs(x[M.Li[p]]);
            best = a > best ? a : best;
        }
        for (int off = 32; off > 0; off >>= 1) {
            const double o = __shfl_xor(best, off, 64);
            best = o > best ? o : best;
        }
        LUCLK(4);
        const bool ok = best > 0.0 && fabs(pivot) >= best * tol;  // (a NaN anywhere fails one of the two)
        if (!ok && lane == 0) atomicMin(&ctl->first_bad, k);
        if (lane == 0) xstore(M.Lx + lp0 - 1, 1.0);  // L(k,k) = 1 first (lib.rs:597-599)
#pragma unroll
        for (int c = 0; c < 2; ++c)
            if (in[c]) xstore(M.Lx + lp0 + 64 * c + lane, v[c] / pivot);  // lib.rs:604 (IEEE division)
        for (int p = lp0 + 128 + lane; p < lp1; p += 64) xstore(M.Lx + p, x[M.Li[p]] / pivot);
        LUCLK(6);
        turn();  // L(:,k) is in the L2 before the flag is (a failed column raises it too: nobody may wait for ever)
        LUCLK(7);
        if (lane == 0) __hip_atomic_store(done + k, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef RSP_LU_CLOCKS
        if (k == (int)(n / 2) - 1 && lane == 0) g_lu_clk[8] = wall_clock64();  // the previous column raises its flag
        if (k == (int)(n / 2) && lane == 0) g_lu_clk[9] = (unsigned long long)(ud - ub), g_lu_clk[10] = (unsigned long long)(lp1 - lp0 + 1);
#endif
        // behind the flag: U(:,k) (read by the host alone) and the work vector back to zeros for this wavefront's next column
        // (lib.rs:607); four pieces' loads together, then their stores (one by one each store would wait for its own load)
        for (int t = ub + lane; t < ud; t += 256) {
            int u[4];
            double w[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) u[c] = t + 64 * c < ud ? M.Ui[t + 64 * c] : -1;
#pragma unroll
            for (int c = 0; c < 4; ++c) w[c] = u[c] >= 0 ? x[u[c]] : 0.0;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (u[c] >= 0) {
                    M.Ux[t + 64 * c] = w[c];
                    x[u[c]] = 0.0;
                }
        }
        LUCLK(5);
        if (lane == 0) {
            M.Ux[ud] = pivot;  // U(k,k) last (lib.rs:593-595)
            x[k] = 0.0;
        }
#pragma unroll
        for (int c = 0; c < 2; ++c)
            if (in[c]) x[my_r[c]] = 0.0;
        for (int p = lp0 + 128 + lane; p < lp1; p += 64) x[M.Li[p]] = 0.0;
        turn();  // (the zeros have landed before the next column scatters into the same vector)
    }
}

}  // namespace dev

constexpr double kLuMinUpdates = 2e7;  // below this many multiply-subtracts the host routine finishes as soon as patterns plus device
                                       // set-up (measured: 5.6e6 updates 7.0 ms host / 5.9 ms GPU path, 4.3e5 updates 1.6 / 3.9 ms)
constexpr int kLuDeclined = 100;       // lu_numeric_device: not applicable or not worth it here -- use sparse_lu

// Device half of one factorisation with static pivots: the patterns of A, L and U resident, the values computed by
// lu_columns.  Kept by the handle, so that a matrix with the SAME pattern and new values is factored again by the kernel
// alone (rsp_lu_refactor_f64: nothing of the depth-first search is repeated).
class LuDevice {
   public:
    // f: patterns from lu_static_pattern (p, i, pinv); A: the matrix they belong to; q: its column order (nullptr: natural)
    void setup(const CscView &A, const i64 *q, const LuFactors &f, const std::vector<int32_t> &level)
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
        {  // columns by dependency level, by number inside a level (a counting sort)
            int32_t nlev = 0;
            for (int32_t l : level) nlev = std::max(nlev, l + 1);
            std::vector<i64> at((size_t)nlev + 1, 0);
            for (int32_t l : level) at[(size_t)l + 1]++;
            for (int32_t l = 0; l < nlev; ++l) at[(size_t)l + 1] += at[(size_t)l];
            std::vector<int32_t> order((size_t)n_);
            for (i64 k = 0; k < n_; ++k) order[(size_t)at[(size_t)level[(size_t)k]]++] = (int32_t)k;
            dOrder_.upload(order);
        }
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
        // (8 and 16 GiB, 5 368 and 8 192 wavefronts at C4, were measured: no faster -- the kernel follows its dependency chain)
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
        const dev::LuArrays M{dAp_.get(), dArow_.get(), dAx_.get(), dAcol_.get(), dOrder_.get(), dLp_.get(), dLi_.get(), dLx_.get(), dUp_.get(), dUi_.get(), dUx_.get()};
        hipLaunchKernelGGL(dev::lu_columns, dim3((unsigned)slots_), dim3(64), 0, nullptr, M, dWork_.get(), n_, tol, dCtl_.get(), dDone_.get(), spin_limit);
        RSP_HIP(hipGetLastError());
        dev::LuControl end{};
        dCtl_.download(&end, 1);  // (synchronises)
#ifdef RSP_LU_CLOCKS
        {
            unsigned long long c[16] = {};
            RSP_HIP(hipMemcpyFromSymbol(c, HIP_SYMBOL(dev::g_lu_clk), sizeof c));
            fprintf(stderr, "[rsp lu clocks] column n/2: %llu dependencies, %llu rows of L; 10 ns ticks since the column was taken: last wait starts %lld, "
                            "previous column's flag raised %lld, last turn starts %lld, dependencies done %lld, pivot rule %lld, U written %lld, L written %lld, "
                            "stores landed %lld\n",
                    c[9], c[10], (long long)(c[1] - c[0]), (long long)(c[8] - c[0]), (long long)(c[2] - c[0]), (long long)(c[3] - c[0]),
                    (long long)(c[4] - c[0]), (long long)(c[5] - c[0]), (long long)(c[6] - c[0]), (long long)(c[7] - c[0]));
        }
#endif
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
    const double *dLx() const { return dLx_.get(); }  // the factors' values where lu_columns left them
    const double *dUx() const { return dUx_.get(); }
    i64 wavefronts() const { return slots_; }
    i64 device_bytes() const
    {
        return (i64)(dAp_.bytes() + dArow_.bytes() + dAcol_.bytes() + dOrder_.bytes() + dLp_.bytes() + dLi_.bytes() + dUp_.bytes() + dUi_.bytes() + dAx_.bytes() +
                     dLx_.bytes() + dUx_.bytes() + dWork_.bytes() + dDone_.bytes());
    }

   private:
    i64 n_ = 0, nnz_ = 0, slots_ = 0;
    uint64_t pattern_ = 0;
    std::vector<i64> q_;
    bool has_q_ = false;
    DeviceBuffer<int64_t> dAp_;
    DeviceBuffer<int32_t> dArow_, dAcol_, dOrder_, dLp_, dLi_, dUp_, dUi_;
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
    if (mode != 1) {
        // A cheap look before the expensive search (which costs as much as half a host factorisation and is wasted when the
        // device check fails): the pivot rule on the columns of A itself.  It is exact for columns that no earlier column
        // updates and a heuristic for the rest; a matrix that fails it here would almost surely fail it on the device.
        for (i64 c = 0; c < n; ++c) {
            double diag = 0.0, largest = 0.0;
            for (i64 e = A.p[c]; e < A.p[c + 1]; ++e) {
                const double a = std::fabs(A.x[e]);
                if (A.i[e] == c) diag = a;
                largest = a > largest ? a : largest;
            }
            if (!(largest > 0.0) || !(diag >= largest * tol)) return kLuDeclined;
        }
    }
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
    try {
        D->setup(A, q, f, level);
    } catch (const DeviceError &) {  // no room for the work vectors and patterns on the device: the host routine needs none
        (void)hipGetLastError();
        return kLuDeclined;
    }
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
