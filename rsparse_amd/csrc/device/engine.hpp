// Device-side objects of librsparse_hip: RAII buffers, the triangular-solve engine and the numeric
// Cholesky engine.  Everything here runs on the GPU; there is no CPU fallback for numeric work.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../host/sparse.hpp"

namespace rsp {

struct DeviceError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define RSP_HIP(expr)                                                                                         \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess)                                                                                 \
            throw ::rsp::DeviceError(std::string(#expr) + ": " + hipGetErrorString(e_));                      \
    } while (0)

// An n x nrhs column-major block between host (leading dimension ld_host) and device (packed, leading dimension n): one
// plain copy when the host block is packed too, else one per column.  Not hipMemcpy2D: under rocprofv3 (kernel trace or
// counters) its host-to-device form, inside this library's process, returns hipSuccess and copies NOTHING every other
// call or so on this ROCm (7.2.0; first seen as a residual of 4.9 in a one-column solve that is exact outside the
// profiler).  Established in round 3 with the diagnostic below (RSP_COPY_2D_CHECK=1, tools/r3_copy2d_incident.sh;
// profiles/r03_memcpy2d_incident.txt): the block read back right after the call, and again after hipDeviceSynchronize,
// is the buffer's old contents -- lost, not late; a standalone single-threaded program does not show it.
inline void copy_block_to_device(double *d, const double *h, i64 n, i64 ld_host, i64 nrhs)
{
    if (const char *dbg = std::getenv("RSP_COPY_2D_CHECK")) {  // diagnostic: the hipMemcpy2D form, read back and compared
        if (dbg[0] == '1') {
            RSP_HIP(hipMemcpy2D(d, (size_t)n * sizeof(double), h, (size_t)ld_host * sizeof(double), (size_t)n * sizeof(double), (size_t)nrhs,
                                hipMemcpyHostToDevice));
            std::vector<double> back((size_t)n * (size_t)nrhs);
            RSP_HIP(hipMemcpy(back.data(), d, back.size() * sizeof(double), hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (i64 j = 0; j < nrhs; ++j)
                for (i64 i = 0; i < n; ++i) bad += back[(size_t)(j * n + i)] != h[(size_t)(j * ld_host + i)];
            std::fprintf(stderr, "copy check: hipMemcpy2D of %lld x %lld (ld %lld): %zu entries differ right after the copy\n", (long long)n,
                         (long long)nrhs, (long long)ld_host, bad);
            if (bad) {  // late or lost?  once more after the whole device has drained
                RSP_HIP(hipDeviceSynchronize());
                RSP_HIP(hipMemcpy(back.data(), d, back.size() * sizeof(double), hipMemcpyDeviceToHost));
                size_t still = 0, zeros = 0;
                for (i64 j = 0; j < nrhs; ++j)
                    for (i64 i = 0; i < n; ++i) {
                        still += back[(size_t)(j * n + i)] != h[(size_t)(j * ld_host + i)];
                        zeros += back[(size_t)(j * n + i)] == 0.0;
                    }
                std::fprintf(stderr, "copy check:   after hipDeviceSynchronize: %zu entries differ (%zu of the block's entries are 0.0); repaired with a plain copy\n",
                             still, zeros);
            }
            if (!bad) return;  // (else: the plain copies below)
        }
    }
    if (nrhs == 1 || ld_host == n) {
        RSP_HIP(hipMemcpy(d, h, (size_t)n * (size_t)nrhs * sizeof(double), hipMemcpyHostToDevice));
        return;
    }
    for (i64 j = 0; j < nrhs; ++j) RSP_HIP(hipMemcpy(d + j * n, h + j * ld_host, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
}
inline void copy_block_to_host(double *h, const double *d, i64 n, i64 ld_host, i64 nrhs)
{
    if (nrhs == 1 || ld_host == n) {
        RSP_HIP(hipMemcpy(h, d, (size_t)n * (size_t)nrhs * sizeof(double), hipMemcpyDeviceToHost));
        return;
    }
    for (i64 j = 0; j < nrhs; ++j) RSP_HIP(hipMemcpy(h + j * ld_host, d + j * n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
}

template <class T>
class DeviceBuffer {
   public:
    DeviceBuffer() = default;
    explicit DeviceBuffer(size_t count) { resize(count); }
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    DeviceBuffer(DeviceBuffer &&o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
    DeviceBuffer &operator=(DeviceBuffer &&o) noexcept
    {
        if (this != &o) { release(); p_ = o.p_; n_ = o.n_; o.p_ = nullptr; o.n_ = 0; }
        return *this;
    }
    ~DeviceBuffer() { release(); }
    void resize(size_t count)
    {
        if (count <= n_ && p_) return;
        release();
        RSP_HIP(hipMalloc((void **)&p_, (count ? count : 1) * sizeof(T)));
        n_ = count;
    }
    void upload(const T *src, size_t count)
    {
        resize(count);
        if (count) RSP_HIP(hipMemcpy(p_, src, count * sizeof(T), hipMemcpyHostToDevice));
    }
    void upload(const std::vector<T> &v) { upload(v.data(), v.size()); }
    void download(T *dst, size_t count) const
    {
        if (count) RSP_HIP(hipMemcpy(dst, p_, count * sizeof(T), hipMemcpyDeviceToHost));
    }
    T *get() const { return p_; }
    size_t size() const { return n_; }
    size_t bytes() const { return n_ * sizeof(T); }

   private:
    void release()
    {
        if (p_) (void)hipFree(p_);
        p_ = nullptr;
        n_ = 0;
    }
    T *p_ = nullptr;
    size_t n_ = 0;
};

// One level-scheduled program (see solve_kernels.hip) resident on the device.
struct TriProgram {
    i64 ntasks = 0, nlevels = 0, longest_task = 0;
    DeviceBuffer<int32_t> task_ptr, op_row, op_ptr, ent_idx;
    DeviceBuffer<double> op_diag, ent_val;
    std::vector<std::pair<int, int>> levels;  // (first task, number of tasks)
    size_t device_bytes() const
    {
        return task_ptr.bytes() + op_row.bytes() + op_ptr.bytes() + ent_idx.bytes() + op_diag.bytes() + ent_val.bytes();
    }
};

// Host image of a TriProgram before its upload (built without any device call, so it can be made on a worker thread).
struct TriProgramHost {
    i64 ntasks = 0, nlevels = 0, longest_task = 0;
    std::vector<int32_t> task_ptr, op_row, op_ptr, ent_idx;
    std::vector<double> op_diag, ent_val;
    std::vector<std::pair<int, int>> levels;
};

// The blocked multi-right-hand-side programme (host/blocked.cpp) resident on the device.
struct BlockedProgram {
    i64 ntasks = 0, nlevels = 0, nblocks = 0, scratch_rows = 0, big_chains = 0, inverted_rows = 0, padded_entries = 0;
    i64 nclusters = 0, cluster_rows = 0, cluster_entries = 0, launches = 0;
    int cluster_slots = 0;
    DeviceBuffer<i64> task_ptr;
    DeviceBuffer<BlockHeader> hdr;
    DeviceBuffer<int32_t> src_idx;
    DeviceBuffer<double> coef;
    DeviceBuffer<int32_t> prologue;
    DeviceBuffer<WideItem> items;       // wide items (item_kernels.hip): the chains' blocks again, 64 rows at a time
    DeviceBuffer<uint32_t> item_masks;
    i64 nitems = 0, item_entries = 0;
    bool single_block_tasks = false;
    DeviceBuffer<ClusterHeader> clusters;
    DeviceBuffer<int32_t> cluster_ids;
    DeviceBuffer<GroupRec> groups;
    struct Level {
        i64 task_begin[BlockProgram::kClasses];  // by class (BlockProgram::level_class_ptr)
        int ntasks[BlockProgram::kClasses];
        i64 block_begin[BlockProgram::kClasses];  // first block of each class (split classes: one block per task)
        i64 rec_begin[BlockProgram::kClasses];  // prologue record of the first block of each class
        i64 cluster_begin;
        int nclusters;
        i64 item_begin;
        int nitems, item_ksteps_max;
    };
    std::vector<Level> levels;
    // RSP_FLAGGED_LEVELS=1 (one right-hand side): runs of consecutive latency-bound levels that go out as ONE launch each
    // (dev::solve_levels_flagged): per run its level descriptors and tickets, per ticket the flags it waits for
    struct FlagRunHost {
        size_t level_begin, level_end;  // levels [begin, end) of the programme
        i64 level_off, ticket_off;      // into fr_levels (descriptors) and the per-ticket arrays
        int nlevels, ntickets;
    };
    std::vector<FlagRunHost> flag_runs;
    DeviceBuffer<char> fr_levels, fr_tickets;  // dev::LevelDesc, int2
    DeviceBuffer<int32_t> fr_wait_ptr, fr_wait_list;
    DeviceBuffer<int> fr_flags, fr_counters, fr_failed;
    int fr_epoch = 0;
    bool fr_ran = false;
    size_t device_bytes() const
    {
        return task_ptr.bytes() + hdr.bytes() + prologue.bytes() + src_idx.bytes() + coef.bytes() + clusters.bytes() + cluster_ids.bytes() +
               groups.bytes() + items.bytes() + item_masks.bytes();
    }
};

// Analyse-once / solve-many engine for one triangular factor and one of the four solve kinds.
class TriSolver {
   public:
    // throws DeviceError; returns kOk / kBadInput
    int analyze(const CscView &T, int kind);
    // T.x == nullptr analyses the pattern alone (no device call, any thread); set_values() with the same matrix and its
    // values must follow before anything is solved or staged
    void set_values(const CscView &T);
    // dX device pointer. layout: 0 column-major, 1 row-major (include/rsparse_hip.h).  Asynchronous on `stream`.
    void solve_device(double *dX, int layout, i64 ld, i64 nrhs, hipStream_t stream);
    // X host column-major
    void solve_host(double *X, i64 ldx, i64 nrhs);

    i64 n() const { return n_; }
    i64 nnz() const { return nnz_; }
    const std::vector<i64> &level_of_row() const { return level_of_row_; }
    i64 num_row_levels() const { return num_row_levels_; }
    const TriProgram &multi_program() const { return multi_; }
    const TriProgram &single_program() const { return single_; }
    const BlockedProgram &blocked_program() const { return blocked_; }
    // kernel launches per sweep and tasks of the programme that has actually been built (zeros before the first solve)
    void summary(i64 &launches, i64 &tasks) const
    {
        if (have_blocked_) { launches = blocked_.launches; tasks = blocked_.ntasks; }
        else if (have_single_) { launches = single_.nlevels; tasks = single_.ntasks; }
        else if (have_multi_) { launches = multi_.nlevels; tasks = multi_.ntasks; }
        else { launches = 0; tasks = 0; }
    }
    i64 algorithmic_bytes(i64 nrhs) const { return nnz_ * 12 + (n_ + 1) * 4 + 16 * n_ * nrhs; }
    // pieces of the multi-right-hand-side path that solve_pair_device drives on two streams
    void prepare_blocked(i64 total_cols);
    void stage_blocked();  // host-only: builds the multi-RHS programme ahead of its upload
    // The caller analyses, solves ONE vector and discards (cholsol, lusol, the one-shot lsolve...): take the plain level
    // programme (one launch per dependency level, nothing to build) when its launches cost less than building the
    // blocked programme for a single use.  Handles that solve many times never call this.
    void set_single_use();
    bool single_use_levels() const { return single_use_levels_ || single_use_rows_; }
    // the flagged row kernel of a single-use handle ran and one of its waits gave up (a defect): call after the stream
    // has been synchronised; throws DeviceError
    void check_rows_run();
    // The same kernel with the values taken on the device (cholsol's first call on a pattern: the factor never visits the
    // host).  After a pattern-only analyze(): rows_wanted() says whether a single-use handle would take the row kernel at
    // all (RSP_SINGLE_USE); prepare_rows_pattern() uploads the pattern half -- pointers, indices, the source position of
    // every value, rows by level, grabs -- from any thread with the device current, e.g. beside a factorisation;
    // rows_values_from_device() gathers values and diagonal out of dTx (dTp: the matrix's column pointers on the
    // device) and makes the handle a single-use one.  Asynchronous on `stream`.  false: nothing was prepared.
    static bool rows_wanted();
    void prepare_rows_pattern();
    bool rows_prepared() const { return rows_.pattern_only; }
    bool rows_values_from_device(const int64_t *dTp, const double *dTx, hipStream_t stream);
    // NEW VALUES ON THE ANALYSED PATTERN (a refactorisation).  keep_for_refill() before analyze(): the gather form keeps
    // where each of its values comes from and the blocked programme keeps its recipe (BlockProgram::Refill) and a host
    // copy of its coefficients.  refresh_host(T) then pours the values of T (same pattern) into that copy -- host work
    // only, any thread -- and refresh_upload() sends coefficients and cluster records to the resident programme.
    // refresh_host returns false when there is nothing to refresh in place (no resident blocked programme, no recipe):
    // the caller analyses again.
    void keep_for_refill() { G_.keep_src = true; }
    bool refresh_host(const CscView &T);
    void refresh_upload();
    // The same on the device, from the factor's values where the factorisation left them (dTx; T gives the pattern
    // only): gather kernels for coefficients and cluster records, the 16 x 16 triangles and the chain blocks inverted by
    // kernels (refill_kernels.hip).  The recipe goes to the device at the first call.  Asynchronous on `stream`.  false:
    // not available for this programme (no recipe, duplicate entries inside a chain) -- use refresh_host.  The host copy
    // of the gather form keeps the OLD values afterwards: the plain level programmes cannot be built from it any more
    // (DeviceError) until a refresh_host.
    bool refresh_device(const CscView &T, const double *dTx, hipStream_t stream);
    // PROGRAMME AHEAD OF THE VALUES: after a pattern-only analyze() (keep_for_refill() on), emit the blocked programme with
    // placeholder values -- host work, any thread, e.g. beside a numeric factorisation that is still running.  Every
    // value in it is rewritten by refresh_device() before the first solve; adopt_staged() uploads it (false: nothing
    // usable was staged).
    void stage_blocked_pattern();
    bool adopt_staged();
    // sends the recipe to the device ahead of the first refresh_device (which would do it otherwise); T: the pattern
    void prepare_refresh_device(const CscView &T)
    {
        if (have_blocked_ && kept_ && !recipe_.built) build_device_recipe(T);
    }
    void run_window(double *X_rowmajor, i64 ld, i64 nrhs, i64 col_off, i64 total_cols, hipStream_t stream, hipEvent_t after_heavy,
                    double *scratch_override = nullptr);
    // Row-major solve that allocates nothing: the blocked programme must already be resident (one earlier solve through
    // solve_device) and its scratch rows live in a file-scope __device__ array of kNoAllocScratchDoubles doubles.
    // Returns kBadInput when the programme is not resident or scratch_rows * nrhs exceeds the array.
    int solve_device_noalloc(double *X_rowmajor, i64 ld, i64 nrhs, hipStream_t stream);
    // A REPLICA of this solver's resident blocked programme on another device (or again on the same one): every device
    // array is copied device to device (hipMemcpyPeer), nothing is analysed or emitted again, nothing visits the host.
    // The current device must be dst_device; this solver's programme must be resident (prepare_blocked) and idle.  The
    // replica can solve and nothing else (no refresh: make a new one after a refactorisation).  false: the programme in
    // use here is not the blocked one.
    bool clone_resident(TriSolver &dst, int src_device, int dst_device) const;

   private:
    void build_program(i64 max_task_entries, TriProgram &P);
    TriProgramHost stage_program(i64 max_task_entries) const;  // host half of build_program
    void run_multi(double *X_rowmajor, i64 ld, i64 nrhs, hipStream_t stream);
    void run_blocked(double *X_rowmajor, i64 ld, i64 nrhs, hipStream_t stream);
    void build_blocked();
    void run_single(double *x, hipStream_t stream);

    i64 n_ = 0, nnz_ = 0, num_row_levels_ = 0;
    int kind_ = 0;
    GatherForm G_;  // kept on the host so the second program can be built lazily
    std::vector<i64> level_of_row_;
    TriProgram multi_, single_;
    BlockedProgram blocked_;
    bool have_multi_ = false, have_single_ = false, have_blocked_ = false, single_use_levels_ = false;
    // single use, one vector: rows with flags (solve_rows_flagged) straight from the gather form -- no programme to build
    bool single_use_rows_ = false, rows_ran_ = false;
    struct RowsDevice {
        DeviceBuffer<int64_t> ptr;
        DeviceBuffer<int32_t> idx, order, src;
        DeviceBuffer<double> val, diag;
        DeviceBuffer<int> done;
        DeviceBuffer<char> ctl, grabs;  // dev::RowsControl, int2 (first, end) positions of each grab
        i64 ngrabs = 0;
        bool uploaded = false, pattern_only = false;
    } rows_;
    void upload_rows_schedule();
    void build_flag_runs(const BlockProgram &P);  // (the host programme, before its arrays are moved to the device)
    void run_rows(double *x, hipStream_t stream);
    std::unique_ptr<BlockProgram> staged_;
    std::unique_ptr<BlockProgram> kept_;  // keep_for_refill(): coef, groups and the recipe of the resident programme
    struct DeviceRecipe {
        bool built = false, usable = false;
        DeviceBuffer<int32_t> map, coef_src, grp_first, grp_row, chain_pos, chain_code;
        DeviceBuffer<BlockProgram::Refill::Tri16> tri16;
        DeviceBuffer<char> chains;  // dev::ChainDesc
        DeviceBuffer<double> pool, mpool, tpool;  // chain inverses, the chains' triangles, scratch of the blocked inverse
        DeviceBuffer<char> tiles1, tiles2;        // dev::ChainTile lists of the blocked inverse: T = C Z11 and Z21 = -Z22 T per level
        std::vector<std::pair<i64, i64>> tile_levels;  // (first, count) of each level in both lists
        i64 ncoef = 0, ngroups = 0, ntri = 0, nchain_entries = 0, pool_size = 0;
        int nchains = 0, ntiles = 0;
        int32_t split = 0, nnz = 0;
    } recipe_;
    bool host_values_stale_ = false;  // refresh_device ran since the gather form last received values
    void build_device_recipe(const CscView &T);
    std::unique_ptr<TriProgramHost> staged_single_;  // set_single_use(): the level programme, ready for upload
    DeviceBuffer<double> scratch_;  // scratch rows of the blocked programme (row-major, ld = nrhs)
    DeviceBuffer<double> work_;  // row-major staging for column-major callers
    i64 scratch_cols_ = 0;
};

// first.solve then second.solve on one row-major device block, column halves pipelined over two streams
// a.analyze(Ta, kind_a) and b.analyze(Tb, kind_b) plus both programme builds, on two host threads
// single_use: see TriSolver::set_single_use()
int analyze_pair(TriSolver &a, const CscView &Ta, int kind_a, TriSolver &b, const CscView &Tb, int kind_b, bool single_use = false);
void solve_pair_device(TriSolver &first, TriSolver &second, double *X_rowmajor, i64 ld, i64 nrhs, hipStream_t stream);

// tunables (entries = off-diagonal nonzeros processed sequentially by one wavefront)
constexpr i64 kTaskEntriesMulti = 2048;   // unblocked fallback programme
constexpr i64 kTaskEntriesSingle = 512;
constexpr i64 kLeanLaunchWorkgroups = 128;  // levels with at least this many 512-thread workgroups run the lean variant
constexpr i64 kPairParts = 2;  // column parts of the pair pipeline (C2, 256 right-hand sides: 3 parts are 2 % faster but re-read the
                               // coefficients once more -- 35 instead of 25 GB of fabric traffic per step; 4 parts lose)
constexpr i64 kLeanLaunchWorkgroupsSingle = 128;  // the same switch for a single right-hand side (deep-fetch gather)
constexpr i64 kNoAllocScratchDoubles = 4 << 20;  // 32 MiB per GPU, held until the process ends (rsp_tri_solve_dev_noalloc_f64)
constexpr i64 kItemLaunchWaves = 512;  // a level runs its wide items as items (dev::solve_items) from this many (item, 64 columns) pairs,
                                       // as their twin 16-row blocks below it
constexpr i64 kSmallLaunchWaves = 2048;  // small-block tasks of a level get their own launch from this many wavefronts

}  // namespace rsp
