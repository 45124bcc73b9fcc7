// Numeric sparse Cholesky on the GPU (rsparse::chol, /root/reference/src/lib.rs:278-337).
//
// The reference is up-looking and strictly sequential in the row index.  Here the same factor is
// produced by a supernodal LEFT-looking schedule: columns are grouped into panels (fundamental supernodes
// cut to at most kPanelWidth columns), panels are levelled by the elimination tree, and each level is two
// kernel launches:
//   1. update  : every (panel, row) pair owned by exactly one wavefront gathers the contributions
//                - sum_u L(row,u) L(col,u) of all descendant panels  (atomics-free, deterministic order)
//   2. factor  : dense Cholesky of the panel's diagonal block + triangular solve of the rows below it
// The pattern of L (Lp == cp, diagonal first, rows ascending) is computed on the host exactly as the
// reference lays it out, so the integer structure is bit-identical; values agree to rounding.
#pragma once
#include "engine.hpp"

namespace rsp {

constexpr int kPanelWidth = 32;

class CholFactor {
   public:
    // symbolic work + uploads; needs only the pattern of A. kOk / kBadInput.
    int analyze(const CscView &A, const Symbolic &S);
    // numeric factorisation of values A.x (same pattern as analysed). kOk / kNotPositiveDefinite.
    int factorize(const CscView &A);
    void solve_device(double *dB, int layout, i64 ld, i64 nrhs, bool permute, hipStream_t stream);
    void solve_host(double *B, i64 ldb, i64 nrhs);
    void download(i64 *Lp, i64 *Li, double *Lx) const;
    void info(i64 *out) const;
    i64 n() const { return n_; }
    i64 lnz() const { return lnz_; }
    double factor_ms() const { return factor_ms_; }

   private:
    void build_solvers();

    i64 n_ = 0, lnz_ = 0, annz_ = 0;
    bool has_pinv_ = false;
    std::vector<i64> pinv_, Lp_, Li_;  // host copies (Li_ as the reference stores it)
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
    // device: panels
    DeviceBuffer<int32_t> d_panel_c0;   // npanels+1
    DeviceBuffer<int32_t> d_panel_rows; // npanels: rows below the panel (r)
    // update programme: slots (panel,row position), tuples (pair, q), pairs (source panel, q0, column mask)
    DeviceBuffer<int32_t> d_slot_panel, d_slot_pos;  // per active slot
    DeviceBuffer<i64> d_slot_tptr;                    // per active slot + 1
    DeviceBuffer<int32_t> d_tuple_pair, d_tuple_q;
    DeviceBuffer<int32_t> d_pair_src, d_pair_q0;
    DeviceBuffer<uint32_t> d_pair_mask;
    // factor programme: work items (panel, row block)
    DeviceBuffer<int32_t> d_item_panel, d_item_block;
    DeviceBuffer<int32_t> d_status;  // [0] = smallest column with a non-positive pivot, or INT_MAX
    struct Level {
        i64 slot_begin, slot_count;  // update work (active slots)
        i64 item_begin, item_count;  // factor work: whole small panels + diagonal blocks of large ones
        i64 trsm_begin, trsm_count;  // second launch: row blocks of the large panels
    };
    std::vector<Level> levels_;

    std::unique_ptr<TriSolver> lower_, upper_;
    DeviceBuffer<double> work_;
};

}  // namespace rsp
