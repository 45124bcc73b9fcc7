// Host-side data model and symbolic layer of librsparse_hip (product code, runs on the CPU by design:
// north_star keeps ordering / etree / patterns / LU pivoting on the host).
// Mirrors rsparse's src/data.rs types; every routine cites the reference lines whose RESULT it must
// reproduce.  Independent of oracle/ (which is test infrastructure and never linked here).
#pragma once
#include <climits>
#include <cstdint>
#include <memory>
#include <vector>

namespace rsp {

using i64 = int64_t;

// A vector whose resize() leaves new elements UNINITIALISED: for large index arrays that are written exactly once, by
// several host threads -- value-initialising 400 MB on one thread costs more than filling it (every fresh page faults in
// on that thread).
template <class T>
struct DefaultInitAllocator : std::allocator<T> {
    template <class U>
    struct rebind {
        using other = DefaultInitAllocator<U>;
    };
    using std::allocator<T>::allocator;
    template <class U>
    void construct(U *p)
    {
        ::new (static_cast<void *>(p)) U;
    }
    template <class U, class... Args>
    void construct(U *p, Args &&...args)
    {
        ::new (static_cast<void *>(p)) U(std::forward<Args>(args)...);
    }
};
template <class T>
using RawVector = std::vector<T, DefaultInitAllocator<T>>;

// borrowed CSC view == the arrays of rsparse::data::Sprs<f64> (data.rs:194-208)
struct CscView {
    i64 m = 0, n = 0;
    const i64 *p = nullptr;
    const i64 *i = nullptr;
    const double *x = nullptr;  // may be null for pattern-only work
    i64 nnz() const { return p[n]; }
};

// owning CSC
struct Csc {
    i64 m = 0, n = 0;
    std::vector<i64> p, i;
    std::vector<double> x;
    CscView view() const { return CscView{m, n, p.data(), i.data(), x.empty() ? nullptr : x.data()}; }
};

// pattern-only CSC (orderings never look at values)
struct Pattern {
    i64 m = 0, n = 0;
    std::vector<i64> p, i;
};

// rsparse::data::Symb (data.rs:1022-1038)
struct Symbolic {
    i64 n = 0;
    bool has_pinv = false, has_q = false;
    std::vector<i64> pinv;    // schol: inverse fill-reducing permutation (n)
    std::vector<i64> q;       // sqr: column permutation as amd returns it (n+1)
    std::vector<i64> parent;  // elimination tree
    std::vector<i64> cp;      // column pointers of L
    i64 m2 = 0, lnz = 0, unz = 0;
};

// rsparse::data::Nmrc for LU (data.rs:1064-1074)
struct LuFactors {
    Csc L, U;
    std::vector<i64> pinv;
};

enum : int { kOk = 0, kNotPositiveDefinite = 1, kNoPivot = 2, kBadInput = -1, kDeviceError = -2 };

// ---- ordering.cpp -------------------------------------------------------------------------------
// amd (lib.rs:1292-1752): returns false for the natural ordering, else fills perm (n+1 entries).
bool approximate_minimum_degree(const CscView &A, int order, std::vector<i64> &perm);

// ---- symbolic.cpp -------------------------------------------------------------------------------
Pattern transpose_pattern(const Pattern &A);
// symperm, lib.rs:2369-2408; source (optional) receives, for every entry of C, the entry of A it came from
Csc symmetric_permute_upper(const CscView &A, const i64 *pinv, std::vector<i64> *source = nullptr);
std::vector<i64> elimination_tree(const CscView &C);             // etree(ata=false), lib.rs:2026-2071
// schol (lib.rs:968-986): pinv, parent, cp, lnz = unz = cp[n]
Symbolic symbolic_cholesky(const CscView &A, int order);
// sqr, LU branch (lib.rs:1114-1119, 1134-1137): q = amd(order), lnz = unz = 4*nnz + n
Symbolic symbolic_lu(const CscView &A, int order);

// Pattern of the Cholesky factor exactly as rsparse::chol lays it out (lib.rs:295-334): Lp == cp,
// diagonal first in each column, remaining rows strictly ascending.  Li has cp[n] entries.
// Returns false if cp disagrees with the pattern that parent generates (a corrupt Symb).
bool cholesky_pattern(const CscView &C_upper, const std::vector<i64> &parent, const std::vector<i64> &cp,
                      RawVector<i64> &Li);

// supernode partition of a Cholesky factor: super_start has nsuper+1 entries; consecutive columns j, j+1
// share a supernode when parent[j] == j+1 and count[j] == count[j+1] + 1.
std::vector<i64> fundamental_supernodes(const std::vector<i64> &parent, const std::vector<i64> &cp);

// ---- lu.cpp -------------------------------------------------------------------------------------
// lu (lib.rs:519-622) with threshold partial pivoting; q may be null (natural). Bit-exact pivot sequence,
// L/U storage order and values (compiled with -ffp-contract=off).
int sparse_lu(const CscView &A, const i64 *q, double tol, LuFactors &out);

// ---- lu_static.cpp --------------------------------------------------------------------------------
// Patterns of L and U (reference storage order) and pinv of rsparse::lu when every column pivots on its diagonal (row q[k]),
// plus the dependency level of every column; integer work only.  false: not applicable (see lu_static.cpp).
bool lu_static_pattern(const CscView &A, const i64 *q, LuFactors &out, std::vector<int32_t> &level);

// ---- schedule.cpp -------------------------------------------------------------------------------
// 64-bit hash of a matrix's PATTERN (m, n, p[0..n], i[0..nnz)): what an analysis depends on.  Computed on the host threads
// (chunks hashed independently and combined in order: the value does not depend on the thread count).
uint64_t pattern_hash(const CscView &A);
// A triangular solve in "gather" form: unknown r is computed from earlier unknowns,
//   x[r] = (b[r] - sum_p val[p] * x[idx[p]]) / diag[r],   p in [ptr[r], ptr[r+1])
// processed in ascending r (lsolve, utsolve) or descending r (ltsolve, usolve).
struct GatherForm {
    i64 n = 0;
    bool descending = false;
    std::vector<i64> ptr;      // n+1
    std::vector<int32_t> idx;  // off-diagonal column of each entry
    std::vector<double> val;
    std::vector<double> diag;  // n
    std::vector<int32_t> src;  // pattern-only build: position in T.x of each entry, until fill_gather_values()
    bool placeholder = false;  // a pattern-only build whose values have not been filled in yet (val and diag are zeros)
    bool keep_src = false;     // true: src is recorded by every build and survives fill_gather_values(), so that new values
                               // of the same pattern can be poured in again (refactorisation)
};
// kind: 0 lsolve, 1 ltsolve, 2 usolve, 3 utsolve (include/rsparse_hip.h rsp_tri_kind). Returns kOk or kBadInput.
// T.x == nullptr builds the pattern only (val and diag stay zero, src remembers where every value will come from).
int build_gather_form(const CscView &T, int kind, GatherForm &out);
// the values of a pattern-only gather form, from the same matrix with T.x present
void fill_gather_values(const CscView &T, int kind, GatherForm &G);

// Level schedule over "tasks".  A task is a run of consecutive unknowns (in processing order) solved
// sequentially by one wavefront; all of its dependencies outside the run belong to tasks of lower levels.
struct Schedule {
    std::vector<i64> level_of_row;    // canonical level sets: 0 if no dependency, else 1 + max level of deps
    i64 num_row_levels = 0;
    std::vector<int32_t> task_first;  // first unknown of task t (in processing order: highest index if descending)
    std::vector<int32_t> task_rows;   // number of unknowns in task t
    std::vector<i64> level_ptr;       // tasks of level l are task ids [level_ptr[l], level_ptr[l+1])
    i64 longest_task = 0;
};
// max_task_entries bounds the sequential work of one task (number of off-diagonal entries).
Schedule build_schedule(const GatherForm &G, i64 max_task_entries);

// ---- blocked.cpp ----------------------------------------------------------------------------------
// The multi-right-hand-side programme.  Unknowns are split into two kinds of work:
//
// CLUSTERS (bottom of the dependency forest): a run of consecutive unknowns -- whole small subtrees -- that one
// wavefront solves start to finish with the rows of X it touches held in LDS ("slots": external sources first, then
// the cluster's own unknowns in processing order).  The arithmetic is streamed as 64-byte GROUP records (up to four
// multiply-subtracts each) that the wavefront prefetches several records ahead; nothing on the dependent chain of a
// cluster waits on global memory, so the thousands of short dependent rows at the bottom cost LDS latency, not HBM latency.
//
// BLOCKS (everything above): up to 16 unknowns that are consecutive in processing order and share most of their
// sources (supernode rows).  A block is one unit of work for a wavefront:
//     acc[r] = init[r]  (a row of X, a scratch row, or 0)                         r < rc
//     acc[r] -= coef[e][r] * src[e]     for every source e of the block (a row of X or a scratch row)
//     if has_tri:  acc[r] = (acc[r] - sum_{t<r} tri[r][t] * acc[t]) / tri[r][r]   (the block's own chain)
//     dest[r] = acc[r]
// so every source row is loaded once and used for rc unknowns (register reuse).  Long dense chains (supernodes wider
// than kBigChain) are not solved sequentially: their diagonal block M is inverted on the host once, and the chain
// becomes two sets of independent blocks,
//     Y = X_J - L[J, ext] X_ext   (scratch)      and      X_J = M^-1 Y,
// which removes the chain from the critical path.
struct BlockHeader {   // 32 bytes: the device reads it with one scalar load
    int32_t rc;        // class size: 1, 2, 4 (vector path) or 16 (matrix-core path); rows padded with kNone
    int32_t nsrc;      // number of sources; rc == 16: padded to a multiple of 16 with zero coefficients
    int32_t has_tri;   // 1: in-block triangular solve with the tri coefficients
    int32_t pad;
    i64 src_off;       // into src_idx: [rc destination ids][rc init ids][nsrc source ids]
    i64 coef_off;      // into coef: nsrc*rc source coefficients, then the triangle (rc*rc, or 256 for rc == 16)
};
struct ClusterHeader {  // 32 bytes
    int32_t n_ext;      // slots [0, n_ext): external sources (read only)
    int32_t n_own;      // slots [n_ext, n_ext + n_own): the cluster's unknowns, in processing order (written back)
    int32_t ngroups;
    int32_t pad;
    i64 ids_off;        // into cluster_ids: the row of X behind every slot
    i64 grp_off;        // into groups
};
struct GroupRec {       // 64 bytes: four 16-byte loads, the same address in every lane
    int32_t off[4];     // LDS byte offset (slot * 512) of each source; unused entries repeat a valid slot with coef 0
    double coef[4];
    double rdiag;       // 1 / diagonal, applied when kLast is set
    int32_t own_off;    // LDS byte offset of the unknown this group belongs to
    int32_t flags;      // bits 0-2: number of valid entries (0..4); kFirst: start of an unknown; kLast: divide and store
    static constexpr int32_t kFirst = 8, kLast = 16;
};
// A WIDE ITEM: up to 64 unknowns of one chain phase (4 row tiles of 16, each tile one ranked 16-row group) over the UNION of
// the tiles' source lists,
//     OUT[r] = INIT[r] - sum_e A[r][e] * IN[e].
// A 4-bit mask per K-step (4 sources) names the row tiles that have entries in it; only those 16 x 4 coefficient BLOCKS are
// stored (64 doubles each: lane i + 16 g <-> A[16 t + i][4 k_step + g]), K-step by K-step, tiles ascending.  Executed by
// dev::solve_items (64 x 64 register tiles) on the levels that fill the chip; the same arithmetic exists as ordinary
// 16-row blocks (the item's TWINS, task classes 2 / 4 / 6), which is how latency-bound levels and narrow blocks run it.
struct WideItem {          // 32 bytes
    int32_t m_tiles;       // 1..4
    int32_t k_steps;       // sources / 4 (the list is padded to a multiple of 16 sources, as for blocks)
    i64 coef_off;          // into coef: the item's first block
    i64 ids_off;           // into src_idx: [16 m_tiles destination ids][16 m_tiles initial-value ids][4 k_steps source ids]
    i64 mask_off;          // into item_masks: 4 words per window of 16 K-steps: the masks (4 bits per K-step, two words), the number of
                           // coefficient blocks before the window, 0
};
struct BlockProgram {
    static constexpr int32_t kNone = INT32_MIN;
    // task classes inside a level, in task order: 0 vector-path blocks only; 1 one wavefront per (task, 64 columns), at least one
    // matrix-core block; 3 / 5 one matrix-core block whose sources are split over 4 / 8 wavefronts; 2, 4, 6: the same three
    // for blocks that are TWINS OF WIDE ITEMS (one block per task), so that a level can run its items either as items
    // (dev::solve_items; the block kernel then skips classes 2, 4, 6) or as blocks (classes 1+2, 3+4, 5+6 are contiguous)
    static constexpr int kClasses = 7;
    std::vector<BlockHeader> hdr;    // per block
    // pools
    std::vector<int32_t> src_idx;    // id >= 0: row of X; id < 0 (and != kNone): scratch row ~id
    std::vector<double> coef;
    // PROLOGUE records: what a wavefront needs before it can touch X, in one contiguous piece so that it costs ONE
    // round trip instead of dependent ones (header, then ids, then more ids).  A record is 2 or 3 LINES of kLineWords
    // int32 (a line is one coalesced 256-byte load of the wavefront):
    //   line 0: [0, 8) the BlockHeader itself, [8, 24) destination ids, [24, 40) initial-value ids (first rc of each
    //           used), [40, 56) vector blocks (rc < 16) only: the ids of the first 16 sources
    //   line 1 (and 2): matrix-core blocks only: the ids of ALL sources of the record's wavefront, in order (the last
    //           one repeated to the end of the line); a wavefront with more than 64 (two lines: 128) sources fetches
    //           the later ones from the pool, 64 (128) at a time.
    // A block of split class 2 / 3 has 4 / 8 consecutive records, one per wavefront w, which owns the 16-source stages
    // [nst * w / W, nst * (w + 1) / W) of the block; all other blocks have one record.  Records of classes 0-2 have two
    // lines, those of class 3 three (record_lines).  block_record[b] = LINE index of block b's first record; records
    // follow the block order, so inside one (level, class) range they are an arithmetic sequence.  In memory the lines
    // of a record are INTERLEAVED (word l of line i sits at l * lines + i), so that lane l fetches its word of every
    // line with a single 8- or 12-byte load.
    static constexpr int kLineWords = 64;
    static constexpr int record_lines(int waves) { return waves == 8 ? 3 : 2; }
    std::vector<int32_t> prologue;
    std::vector<i64> block_record;   // per block, plus one past the end
    // clusters, sorted by level
    std::vector<ClusterHeader> clusters;
    std::vector<int32_t> cluster_ids;
    std::vector<GroupRec> groups;
    int32_t cluster_slots = 0;       // LDS slots a wavefront needs (max n_ext + n_own over all clusters)
    // schedule
    std::vector<i64> task_ptr;       // blocks of task t: [task_ptr[t], task_ptr[t+1]) (executed in order by one wavefront)
    std::vector<i64> level_ptr;      // block tasks of level l
    std::vector<i64> level_class_ptr;  // kClasses entries per level (see kClasses)
    std::vector<WideItem> items;       // sorted by level
    std::vector<uint32_t> item_masks;
    std::vector<i64> item_level_ptr;   // items of level l
    std::vector<i64> cluster_level_ptr;  // clusters of level l (same level numbering)
    i64 scratch_rows = 0;
    i64 big_chains = 0, inverted_rows = 0, padded_entries = 0, cluster_rows = 0, cluster_entries = 0;
    i64 item_padded_entries = 0;     // multiply-adds (per right-hand side) of the items view: 256 per (K-step, row tile) with entries
    // REFILL RECIPE: where every value of the programme comes from, so that new values on the same pattern (a
    // refactorisation) are poured into coef and groups without emitting the programme again (refill_block_program).
    // A source code is -1 (a constant: padding, the -1 of a partial-sum combine), an entry of the gather form
    // [0, nnz): G.val[code], a diagonal [nnz, nnz + n): G.diag[code - nnz], or an entry of a chain inverse
    // [nnz + n, ...): -pool[code - nnz - n].  valid == false (duplicate entries that add up, more than 2^31
    // codes): the programme is emitted again instead.
    struct Refill {
        bool valid = false;
        std::vector<int32_t> coef_src;  // one code per entry of coef
        struct Tri16 {                  // a 16 x 16 block triangle whose INVERSE is stored (matrix-core order) at coef_base
            i64 coef_base;
            int32_t rows;
            int32_t src[136];           // code of T[r][t], t <= r, at r (r + 1) / 2 + t
        };
        std::vector<Tri16> tri16;
        struct Chain { i64 t0, t1, pool_off; };  // positions [t0, t1) inverted as one dense triangle into pool
        std::vector<Chain> chains;
        i64 pool_size = 0;
        struct Cluster { i64 lo, hi, grp_off; };  // unknowns [lo, hi] and the first group record, in emission order
        std::vector<Cluster> clusters;
    } refill;
};
// new values (G refilled by fill_gather_values) into the coef and groups of a programme built from the same pattern;
// false if the programme carries no valid recipe
bool refill_block_program(const GatherForm &G, BlockProgram &P);
constexpr i64 kBigChain = 24;        // chains longer than this are replaced by their inverse
constexpr i64 kItemRows = 64;        // unknowns per wide item of a chain (32, 48 or 64; 16: no items)
constexpr i64 kItemMinSources = 256; // ... only over row groups with at least this many sources (shorter ones stay plain blocks)
constexpr i64 kMaxChain = 1024;      // ... in pieces of at most this many unknowns (host cost O(w^3) per piece)
constexpr i64 kSplitSources = 64;    // matrix-core blocks with at least this many sources are split over 4 wavefronts,
                                     // with more than 4x this over 8
constexpr i64 kMaxBlockSources = 1024;  // longer source lists become partial sums (scratch rows) plus a combine block
constexpr int kClusterSlots = 24;    // default LDS slots per cluster (12 KB per wavefront at 64 right-hand sides; measured
                                     // at C2: 24 slots beat 16, 32, 40 and 48 -- occupancy outweighs coverage)
constexpr i64 kTaskBlocks = 1;       // default number of dependent blocks one wavefront runs back to back (measured: 1 beats 2, 3)
BlockProgram build_block_program(const GatherForm &G, i64 max_task_blocks, int cluster_slots);

// Host-side validator of a programme (debug aid, never a solve path: it takes no user data).  Replays the programme
// level by level on ONE synthetic right-hand side with every index bounds-checked and every read checked against the
// level that produces it, and compares with plain sequential substitution over the gather form.
struct ProgramReport {
    i64 bad_index = 0;        // ids / offsets outside their arrays
    i64 order_violations = 0; // a row read before the level (or the earlier block of the same task) that writes it
    i64 unwritten = 0;        // unknowns no block or cluster writes, or written more than once
    double max_rel_error = 0; // against sequential substitution
};
// items_view: the wide items run as items (dev::solve_items) instead of as their row tiles
ProgramReport check_block_program(const GatherForm &G, const BlockProgram &P, bool items_view = false);

}  // namespace rsp
