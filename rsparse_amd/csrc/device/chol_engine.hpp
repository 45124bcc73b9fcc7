// Numeric sparse Cholesky on the GPU (rsparse::chol, /root/reference/src/lib.rs:278-337).
//
// The reference is up-looking and strictly sequential in the row index.  Here the same factor is
// produced by a supernodal RIGHT-looking schedule: columns are grouped into panels (fundamental supernodes
// cut to at most kPanelWidth columns), panels are levelled by the elimination tree, and each level is up to
// three kernel launches:
//   1. factor  : dense Cholesky of the panel's diagonal block + triangular solve of the rows below it
//   2. update  : the outer products of the level's finished column ranges are subtracted from the ancestor panels
//                they touch (16 x 32 tiles on the matrix cores); every target tile is owned by one wavefront, which
//                sums its contributions in a host-fixed order and stores once: no atomics, bit-reproducible
//                (RSP_CHOL_GROUPED=0: the source-owned variant with f64 atomics, ~9 % faster at C2)
// The pattern of L (Lp == cp, diagonal first, rows ascending) is computed on the host exactly as the
// reference lays it out, so the integer structure is bit-identical; values agree to rounding.
#pragma once
#include <thread>

#include "engine.hpp"
#include "chol_kernels.hip"

namespace rsp {

constexpr int kPanelWidth = 32;
constexpr i64 kCholFuseWaves = 256;  // levels with at most this many row-block wavefronts factor and solve in one launch
constexpr i64 kCholGrouped = 1;      // 1: target-owned tile updates (no atomics, reproducible); 0: source-owned with atomics
constexpr i64 kCholBlockPanels = 4;  // panels per block of the in-supernode updates (see chol_engine.hip)
constexpr i64 kBatchMaxN = 16384;  // batched solve keeps the vector of a system in LDS (128 KiB)

class CholFactor {
   public:
    // symbolic work + uploads; needs only the pattern of A. kOk / kBadInput.
    // single_use: the factor will be solved against once and dropped (cholsol); the pattern analysis of L for the
    // triangular solves then starts on host threads as soon as the pattern of L is known, beside the rest of this call
    int analyze(const CscView &A, const Symbolic &S, bool single_use = false);
    // numeric factorisation of values A.x (same pattern as analysed). kOk / kNotPositiveDefinite.
    int factorize(const CscView &A);
    // batched variant (same pattern, nsys value sets): see chol_engine.hip.  n must not exceed kBatchMaxN for solve_batch.
    void factorize_values(const double *Ax_all, i64 nsys, int32_t *first_bad);
    void solve_batch(double *B, i64 nsys);
    i64 annz() const { return annz_; }
    // waits for the host threads that analyse() left emitting the solve programmes (otherwise the first solve waits for them)
    void finish_analysis() { join_pattern_jobs(); }
    // false after a factorisation that failed (not positive definite): the values of L are unusable until one succeeds
    bool valid() const { return factor_valid_; }
    // hash of the analysed pattern of A (pattern_hash): a refactorisation must bring the same pattern
    uint64_t pattern() const { return pattern_hash_; }
    void solve_device(double *dB, int layout, i64 ld, i64 nrhs, bool permute, hipStream_t stream);
    void solve_host(double *B, i64 ldb, i64 nrhs);
    // The same with the columns of B shared out over `devices` (contiguous blocks, one host thread per entry; an id may be
    // listed more than once): every other device works on a REPLICA of the resident solve programmes and of pinv, copied
    // device to device when first needed (TriSolver::clone_resident) and dropped by the next refactorisation.  No
    // exchange between the devices: the blocks are independent (north_star: right-hand-side blocks are sharded per GPU).
    void solve_host_sharded(double *B, i64 ldb, i64 nrhs, const std::vector<int> &devices);
    i64 replicas() const { return (i64)replicas_.size(); }
    void download(i64 *Lp, i64 *Li, double *Lx) const;
    void info(i64 *out) const;
    i64 n() const { return n_; }
    i64 lnz() const { return lnz_; }
    double factor_ms() const { return factor_ms_; }
    // borrowed views of the factor where it lives: pattern on the host (as the reference stores it), values on the device
    const i64 *host_Lp() const { return Lp_.data(); }
    const i64 *host_Li() const { return Li_.data(); }
    const double *device_Lx() const { return d_Lx.get(); }
    // the factor is solved against once and discarded (cholsol): see TriSolver::set_single_use()
    void set_single_use(bool v) { single_use_ = v; }
    // before analyze(): this factor's solves never go through the solve programmes (the batched LDS solve): do not emit them
    void no_solve_programmes() { want_programmes_ = false; }

   private:
    void build_solvers();

    i64 n_ = 0, lnz_ = 0, annz_ = 0;
    bool has_pinv_ = false, single_use_ = false, factor_valid_ = false;
    uint64_t pattern_hash_ = 0;
    std::vector<i64> pinv_, Lp_;  // host copies
    RawVector<i64> Li_;           // (as the reference stores it)
    double flops_ = 0.0;
    double factor_ms_ = 0.0;
    i64 npanels_ = 0, nlevels_ = 0, nsuper_ = 0;

    // device: pattern + values of L
    DeviceBuffer<i64> d_Lp;       // n+1 (entry offsets; 64-bit because lnz may exceed 2^31 on 288 GB parts)
    DeviceBuffer<double> d_Lx;    // lnz
    DeviceBuffer<i64> d_pinv;     // n
    // device: scatter of A into L
    DeviceBuffer<double> d_Ax;    // nnz(A)
    DeviceBuffer<i64> d_Adst;     // nnz(A): destination entry of L, or -1
    // device: panels, update pairs (source panel -> ancestor panel) and the per-level work items
    DeviceBuffer<int32_t> d_panel_c0;         // npanels+1
    DeviceBuffer<dev::CholPanel> d_panels;    // npanels
    DeviceBuffer<dev::CholPair> d_pairs;
    DeviceBuffer<int32_t> d_rel;              // per pair: position in the target panel of every updated row
    DeviceBuffer<int32_t> d_upd_pair, d_upd_q;  // update items: (pair, first row of the 16-row tile)
    DeviceBuffer<dev::CholFactorItem> d_items;         // factor items: (panel record, row block)
    DeviceBuffer<dev::CholTiny> d_tiny;                // tiny leaf supernodes, by level
    DeviceBuffer<i64> d_tiny_tgt;                      // their update targets (entry indices of L)
    DeviceBuffer<int32_t> d_status;  // [0] = smallest column with a non-positive pivot, or INT_MAX
    struct Level {
        i64 item_begin, item_count;  // factor work: whole small panels + diagonal blocks of large ones
        i64 trsm_begin, trsm_count;  // second launch: row blocks of the large panels
        i64 upd_begin, upd_count;    // third launch: rank updates from this level's panels
        i64 tiny_begin, tiny_count;  // leaf supernodes handled whole by chol_tiny_panels
        bool fused;                  // factor and row solve in one launch (chol_panel_factor mode 2)
    };
    std::vector<Level> levels_;
    i64 update_items_ = 0, ntall_ = 0;
    bool grouped_ = false;                  // target-owned (atomics-free) updates
    DeviceBuffer<dev::CholSub> d_subs;      // ... their source ranges, grouped by target tile
    DeviceBuffer<dev::CholGroup> d_groups;  // ... and the tiles, by level
    DeviceBuffer<double> d_dbuf;  // parked diagonal blocks of tall panels (fused levels)
    // batched solve phase: CSR index of L and the level sets of both sweeps
    i64 batch_ = 1;
    bool want_programmes_ = true;
    bool pat_ahead_ = false;  // the pattern jobs also emit the blocked programmes (values follow on the device)
    bool solvers_stale_ = false;  // refactorised since the solve programmes received their values
    int32_t nlev_l_ = 0, nlev_t_ = 0;
    DeviceBuffer<int32_t> d_Li32, d_row_ptr, d_row_col, d_row_pos, d_lev_ptr_l, d_lev_rows_l, d_lev_ptr_t, d_lev_cols_t;

    std::unique_ptr<TriSolver> lower_, upper_;
    DeviceBuffer<double> work_;
    int device_ = 0;  // the device the handle lives on (current when it was analysed)
    struct Replica {
        int device = 0;
        TriSolver lower, upper;
        DeviceBuffer<i64> pinv;
        DeviceBuffer<double> work;
    };
    std::vector<std::unique_ptr<Replica>> replicas_;  // by slot of the device list they were made for
    // single use: the pattern analysis of L for the two sweeps, started by analyze() on two host threads (no device
    // calls there); build_solvers() joins them and adds the values
    std::unique_ptr<TriSolver> pat_lower_, pat_upper_;
    int pat_status_[2] = {kOk, kOk};
    std::thread pat_thread_[2];
    void join_pattern_jobs();

   public:
    CholFactor() = default;
    CholFactor(const CholFactor &) = delete;
    CholFactor &operator=(const CholFactor &) = delete;
    ~CholFactor() { join_pattern_jobs(); }
};

}  // namespace rsp
