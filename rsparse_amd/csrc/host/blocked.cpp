// Builder of the multi-right-hand-side programme (see BlockProgram in sparse.hpp).
// Integer/host work only; no counterpart in the reference, whose solves are sequential loops
// (/root/reference/src/lib.rs:464-512, 1230-1278).  The arithmetic each unknown receives is the same
// set of multiply-subtracts as in those loops, regrouped; for chains replaced by an explicit inverse
// the diagonal block is applied as M^-1 instead of by substitution (results agree to rounding).
//
// Everything here is valid for ANY triangular pattern (LU factors with unsorted rows included): the forest
// used to pick clusters is only a heuristic, dependencies and levels are always taken from the real entries.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <cstdio>
#include <thread>

#include "parallel.hpp"
#include "sparse.hpp"

namespace rsp {
namespace {

constexpr int32_t kNone = BlockProgram::kNone;

struct StageClock {  // RSP_TIMING=1: where the host time of a programme build goes (stderr)
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(int k)
    {
        const auto t1 = std::chrono::steady_clock::now();
        acc[k] += std::chrono::duration<double>(t1 - t0).count();
        t0 = t1;
    }
};

// 1, 2, 4: vector path (registers per lane); more than 4 unknowns: a 16-row matrix-core block (MFMA 16x16x4 f64)
inline int class_of(i64 rows) { return rows <= 1 ? 1 : rows <= 2 ? 2 : rows <= 4 ? 4 : 16; }

struct Triple {
    int32_t pos;   // processing position of the source (or scratch ordinal for phase-B blocks)
    int32_t slot;  // row slot inside the block
    double val;
    int32_t code = -1;  // where val comes from (BlockProgram::Refill), -1: a constant
};

// ---- numeric pieces shared by the emitter and refill_block_program: one code, bit-identical results ----------------
// Inverse of a chain's dense diagonal block.  Positions [t0, t1) of the processing order; M in position coordinates
// (duplicate entries add up), then one forward substitution per column of the inverse.  Columns are dealt round-robin
// to the host threads (column c costs (w - c)^2 / 2, so interleaving balances them); the result does not depend on
// the number of threads.
void chain_inverse(const GatherForm &G, i64 t0, i64 t1, std::vector<double> &Minv)
{
    const i64 n = G.n, w = t1 - t0;
    auto row_at = [&](i64 t) { return G.descending ? n - 1 - t : t; };
    auto pos_of = [&](i64 r) { return G.descending ? n - 1 - r : r; };
    std::vector<double> M((size_t)w * w, 0.0);
    for (i64 t = t0; t < t1; ++t) {
        const i64 r = row_at(t);
        M[(size_t)(t - t0) * w + (t - t0)] = G.diag[r];
        for (i64 p = G.ptr[r]; p < G.ptr[r + 1]; ++p) {
            const i64 ps = pos_of(G.idx[p]);
            if (ps >= t0) M[(size_t)(t - t0) * w + (ps - t0)] += G.val[p];
        }
    }
    Minv.assign((size_t)w * w, 0.0);
    auto invert_columns = [&](i64 first, i64 stride) {
        std::vector<double> z(w);
        for (i64 c = first; c < w; c += stride) {
            z[c] = 1.0 / M[(size_t)c * w + c];
            for (i64 r = c + 1; r < w; ++r) {
                const double *Mr = &M[(size_t)r * w];
                double sum = 0.0;
                for (i64 t = c; t < r; ++t) sum += Mr[t] * z[t];
                z[r] = -sum / Mr[r];
            }
            for (i64 r = c; r < w; ++r) Minv[(size_t)r * w + c] = z[r];
        }
    };
    // starting a thread costs tens of microseconds: only chains whose inverse takes milliseconds are shared out
    const i64 nthreads = w >= 384 ? std::min<i64>(host_threads(), w / 64) : 1;
    if (nthreads <= 1) {
        invert_columns(0, 1);
    } else {
        std::vector<std::thread> pool;
        for (i64 k = 0; k < nthreads; ++k) pool.emplace_back(invert_columns, k, nthreads);
        for (std::thread &th : pool) th.join();
    }
}

// inverse of a block's 16 x 16 lower triangle T (row-major, unit rows for padding), stored in the A-operand order of
// v_mfma_f64_16x16x4_f64: K-step k, lane l holds Tinv[l & 15][4k + (l >> 4)]
void store_tri16_inverse(const double (&T)[16][16], double *out)
{
    double Ti[16][16];
    for (int r = 0; r < 16; ++r)
        for (int t = 0; t < 16; ++t) Ti[r][t] = 0.0;
    for (int c = 0; c < 16; ++c) {
        Ti[c][c] = 1.0 / T[c][c];
        for (int r = c + 1; r < 16; ++r) {
            double sum = 0.0;
            for (int t = c; t < r; ++t) sum += T[r][t] * Ti[t][c];
            Ti[r][c] = -sum / T[r][r];
        }
    }
    for (int k = 0; k < 4; ++k)
        for (int l = 0; l < 64; ++l) out[(size_t)k * 64 + l] = Ti[l & 15][4 * k + (l >> 4)];
}

// the group records of one cluster row: coefficients and reciprocal diagonal (the offsets are pattern)
inline void fill_group_values(const GatherForm &G, i64 r, GroupRec *recs)
{
    const i64 pb = G.ptr[r], ne = G.ptr[r + 1] - pb;
    const i64 ng = std::max<i64>(1, (ne + 3) / 4);
    for (i64 g = 0; g < ng; ++g) {
        const i64 cnt = std::max<i64>(0, std::min<i64>(4, ne - 4 * g));
        for (i64 e = 0; e < 4; ++e) recs[g].coef[e] = e < cnt ? G.val[pb + 4 * g + e] : 0.0;
        recs[g].rdiag = 1.0 / G.diag[r];
    }
}

inline void sort_triples(std::vector<Triple> &src)  // by (pos, slot); most callers hand over sorted lists
{
    auto less = [](const Triple &a, const Triple &b) { return a.pos != b.pos ? a.pos < b.pos : a.slot < b.slot; };
    if (!std::is_sorted(src.begin(), src.end(), less)) std::sort(src.begin(), src.end(), less);
}


class Builder {
   public:
    Builder(const GatherForm &G, i64 max_blocks, int slots)
        : G_(G), n_(G.n), max_blocks_(std::max<i64>(1, max_blocks)), slots_(std::min(slots, 64)),  // group-record clusters: one id per lane
          recipe_ok_(G.keep_src)
    {
        avail_.assign(n_, 0);
        task_of_.assign(n_, -1);
        stamp_.assign(n_, -1);
        cluster_of_.assign(n_, -1);
        auto tune = [](const char *name, i64 &v) {  // experiments only; the defaults are the measured optimum at C2
            if (const char *e = std::getenv(name)) v = std::atoll(e);
        };
        tune("RSP_SPLIT_SOURCES", split_sources_);
        split8_sources_ = 4 * split_sources_;
        tune("RSP_SPLIT8_SOURCES", split8_sources_);
        tune("RSP_BIG_CHAIN", big_chain_);
        tune("RSP_MAX_BLOCK_SOURCES", max_block_sources_);
        emit_threads_ = host_threads();  // RSP_EMIT_THREADS=0: the sequential path (the programme does not depend on it)
        tune("RSP_EMIT_THREADS", emit_threads_);
        tune("RSP_ITEM_ROWS", item_rows_);
        item_rows_ = item_rows_ >= 64 ? 64 : item_rows_ >= 48 ? 48 : item_rows_ >= 32 ? 32 : 16;
        tune("RSP_SORT_ROWS", sort_rows_);
        tune("RSP_ITEM_MIN_SOURCES", item_min_sources_);
        if (const char *e = std::getenv("RSP_ITEM_MIN_FILL")) item_min_fill_ = std::atof(e);
    }

    BlockProgram run()
    {
        clock_.lap(7);
        // the pools grow block by block: room for the usual size up front (1.2 - 1.7 entries per entry of the factor at C2) spares the
        // copies of a vector that doubles its way up to gigabytes
        P_.coef.reserve(G_.idx.size() + 3 * (G_.idx.size() / 4));
        if (recipe_ok_) P_.refill.coef_src.reserve(G_.idx.size() + 3 * (G_.idx.size() / 4));
        P_.src_idx.reserve(G_.idx.size() / 4);
        if (slots_ >= 2) select_clusters();
        clock_.lap(0);
        std::vector<char> emitted(cluster_lo_.size(), 0);
        for (i64 t = 0; t < n_;) {
            const i64 r = row_at(t);
            const int32_t c = cluster_of_[r];
            if (c >= 0) {
                if (!emitted[c]) {
                    emit_cluster(c);
                    emitted[c] = 1;
                }
                ++t;
                continue;
            }
            i64 t1 = t;  // maximal segment of consecutive unknowns that are not in a cluster
            while (t1 < n_ && cluster_of_[row_at(t1)] < 0) ++t1;
            emit_segment(t, t1);
            t = t1;
        }
        clock_.lap(1);
        finish();
        P_.refill.valid = recipe_ok_ && (i64)G_.idx.size() + n_ + P_.refill.pool_size < INT32_MAX;
        if (!P_.refill.valid) P_.refill = BlockProgram::Refill();
        clock_.lap(2);
        if (std::getenv("RSP_TIMING"))
            std::fprintf(stderr, "programme build: clusters %.2fs, emission %.2fs (of which chains: matrix %.2fs, inverse %.2fs, blocks %.2fs + %.2fs), finish %.2fs\n",
                         clock_.acc[0], clock_.acc[1], clock_.acc[3], clock_.acc[4], clock_.acc[5], clock_.acc[6], clock_.acc[2]);
        return std::move(P_);
    }

   private:
    i64 row_at(i64 t) const { return G_.descending ? n_ - 1 - t : t; }
    i64 pos_of(i64 r) const { return G_.descending ? n_ - 1 - r : r; }

    // ---- clusters ------------------------------------------------------------------------------------------
    // Forest in index space with the root towards larger indices: parent[j] = smallest index > j related to j
    // (the first unknown that uses x[j] in ascending solves, the nearest dependency in descending ones).  For a
    // Cholesky factor this is the elimination tree, whose subtrees are index ranges (the ordering is a postorder).
    void select_clusters()
    {
        std::vector<i64> parent(n_, -1);
        if (!G_.descending) {
            for (i64 i = 0; i < n_; ++i)
                for (i64 p = G_.ptr[i]; p < G_.ptr[i + 1]; ++p)
                    if (parent[G_.idx[p]] < 0) parent[G_.idx[p]] = i;
        } else {
            for (i64 j = 0; j < n_; ++j)
                for (i64 p = G_.ptr[j]; p < G_.ptr[j + 1]; ++p)
                    if (parent[j] < 0 || G_.idx[p] < parent[j]) parent[j] = G_.idx[p];
        }
        std::vector<i64> size(n_, 1), first(n_);
        for (i64 j = 0; j < n_; ++j) first[j] = j;
        for (i64 j = 0; j < n_; ++j) {
            const i64 p = parent[j];
            if (p < 0) continue;
            size[p] += size[j];
            first[p] = std::min(first[p], first[j]);
        }
        // maximal subtrees that are index ranges and fit the slot budget (own unknowns + estimated external sources)
        std::vector<i64> unit_root(n_, -1);
        for (i64 j = n_ - 1; j >= 0; --j) {
            const i64 p = parent[j];
            if (p >= 0 && unit_root[p] >= 0) {
                unit_root[j] = unit_root[p];
                continue;
            }
            const i64 ext = G_.descending ? G_.ptr[j + 1] - G_.ptr[j] : 0;
            if (first[j] == j - size[j] + 1 && size[j] + ext <= slots_) unit_root[j] = j;
        }
        // pack index-adjacent units into clusters while the exact slot count fits.  A cluster is always an index
        // range, so its external sources lie strictly before it in processing order and cannot depend on it.
        i64 a = -1, b = -1;  // current cluster range
        auto close = [&]() {
            if (a < 0) return;
            const int32_t c = (int32_t)cluster_lo_.size();
            cluster_lo_.push_back(a);
            cluster_hi_.push_back(b);
            for (i64 r = a; r <= b; ++r) cluster_of_[r] = c;
            a = b = -1;
        };
        for (i64 i = 0; i < n_;) {
            const i64 root = unit_root[i];
            if (root < 0) {
                close();
                ++i;
                continue;
            }
            const i64 lo = i, hi = root;  // the unit (i == first[root] by construction)
            if (a >= 0 && lo == b + 1 && slots_needed(a, hi) <= slots_) {
                b = hi;
            } else {
                close();
                if (slots_needed(lo, hi) <= slots_) {
                    a = lo;
                    b = hi;
                }
            }
            i = hi + 1;
        }
        close();
    }

    // unknowns of [lo, hi] plus their distinct sources outside the range
    i64 slots_needed(i64 lo, i64 hi)
    {
        if (hi - lo + 1 > slots_) return hi - lo + 1;
        ++stamp_id_;
        i64 ext = 0;
        for (i64 r = lo; r <= hi; ++r)
            for (i64 p = G_.ptr[r]; p < G_.ptr[r + 1]; ++p) {
                const i64 s = G_.idx[p];
                if (s >= lo && s <= hi) continue;
                if (stamp_[s] != stamp_id_) {
                    stamp_[s] = stamp_id_;
                    ++ext;
                }
            }
        return hi - lo + 1 + ext;
    }

    void emit_cluster(int32_t c)
    {
        close_task();
        const i64 lo = cluster_lo_[c], hi = cluster_hi_[c];
        const i64 n_own = hi - lo + 1;
        if (slot_.empty()) slot_.assign(n_, -1);
        // external sources in first-use order
        std::vector<int32_t> ext;
        ++stamp_id_;
        i64 level = 0;
        for (i64 k = 0; k < n_own; ++k) {
            const i64 r = G_.descending ? hi - k : lo + k;
            for (i64 p = G_.ptr[r]; p < G_.ptr[r + 1]; ++p) {
                const i64 s = G_.idx[p];
                if (s >= lo && s <= hi) continue;
                if (stamp_[s] != stamp_id_) {
                    stamp_[s] = stamp_id_;
                    slot_[s] = (int32_t)ext.size();
                    ext.push_back((int32_t)s);
                    level = std::max<i64>(level, avail_[s] + 1);
                }
            }
        }
        ClusterHeader h{(int32_t)ext.size(), (int32_t)n_own, 0, 0, (i64)P_.cluster_ids.size(), (i64)P_.groups.size()};
        P_.refill.clusters.push_back(BlockProgram::Refill::Cluster{lo, hi, h.grp_off});
        P_.cluster_ids.insert(P_.cluster_ids.end(), ext.begin(), ext.end());
        for (i64 k = 0; k < n_own; ++k) {
            const i64 r = G_.descending ? hi - k : lo + k;
            slot_[r] = (int32_t)(ext.size() + k);
            P_.cluster_ids.push_back((int32_t)r);
        }
        for (i64 k = 0; k < n_own; ++k) {
            const i64 r = G_.descending ? hi - k : lo + k;
            const int32_t own_off = slot_[r] * 512;
            const i64 pb = G_.ptr[r], ne = G_.ptr[r + 1] - pb;
            const i64 ng = std::max<i64>(1, (ne + 3) / 4);
            const size_t first_rec = P_.groups.size();
            for (i64 g = 0; g < ng; ++g) {
                GroupRec rec;
                const i64 cnt = std::max<i64>(0, std::min<i64>(4, ne - 4 * g));
                for (i64 e = 0; e < 4; ++e) rec.off[e] = e < cnt ? slot_[G_.idx[pb + 4 * g + e]] * 512 : own_off;
                rec.own_off = own_off;
                rec.flags = (int32_t)cnt | (g == 0 ? GroupRec::kFirst : 0) | (g == ng - 1 ? GroupRec::kLast : 0);
                P_.groups.push_back(rec);
            }
            fill_group_values(G_, r, P_.groups.data() + first_rec);
            avail_[r] = (int32_t)level;
            task_of_[r] = -2;
            P_.cluster_entries += ne;
        }
        h.ngroups = (int32_t)((i64)P_.groups.size() - h.grp_off);
        P_.clusters.push_back(h);
        cluster_level_.push_back((int32_t)level);
        emit_order_.push_back(c);
        P_.cluster_rows += n_own;
        P_.cluster_slots = std::max<int32_t>(P_.cluster_slots, h.n_ext + h.n_own);
    }

    // ---- segments of unknowns outside clusters: maximal runs in which each unknown depends on ALL earlier ones ----
    void emit_segment(i64 s0, i64 s1)
    {
        if (mark_.empty()) mark_.assign(n_, -1);
        i64 t0 = s0;
        ++run_id_;
        for (i64 t = s0; t <= s1; ++t) {
            bool cut = (t == s1);
            if (!cut && t > t0) {
                const i64 r = row_at(t);
                i64 inside = 0;
                ++stamp_id_;  // distinct members of the run among the sources (the reference allows duplicate entries)
                for (i64 p = G_.ptr[r]; p < G_.ptr[r + 1]; ++p) {
                    const i64 d = G_.idx[p];
                    if (mark_[d] == run_id_ && stamp_[d] != stamp_id_) {
                        stamp_[d] = stamp_id_;
                        ++inside;
                    }
                }
                cut = inside != t - t0;
            }
            if (cut) {
                if (t - t0 > big_chain_) {
                    // the inverse of a chain costs O(w^3) on the host: very long chains are cut into pieces, each piece
                    // seeing the earlier ones as ordinary sources (two more levels per piece)
                    for (i64 c0 = t0; c0 < t; c0 += kMaxChain) emit_big_chain(c0, std::min(c0 + kMaxChain, t));
                } else {
                    emit_normal_run(t0, t);
                }
                t0 = t;
                ++run_id_;
            }
            if (t < s1) mark_[row_at(t)] = run_id_;
        }
    }

    // ---- refill recipe (BlockProgram::Refill): a source code beside every value that is written -----------------
    // (only for gather forms that keep their value sources: nobody can refill the others)
    void grow_coef(size_t new_size)
    {
        P_.coef.resize(new_size, 0.0);
        if (recipe_ok_) P_.refill.coef_src.resize(new_size, -1);
    }
    void set_code(size_t at, int32_t code)
    {
        if (code < 0 || !recipe_ok_) return;
        int32_t &c = P_.refill.coef_src[at];
        if (c != -1) recipe_ok_ = false;  // two values add up in one slot (duplicate entries of the matrix)
        c = code;
    }
    int32_t diag_code(i64 r) const { return (int32_t)((i64)G_.idx.size() + r); }
    int32_t pool_code(i64 at) const
    {
        const i64 c = (i64)G_.idx.size() + n_ + at;
        return c < INT32_MAX ? (int32_t)c : -1;  // (run() invalidates the recipe when the pool does not fit)
    }

    // ---- low-level emission ------------------------------------------------------------------------
    // sources: triples sorted by (pos, slot); ids of sources given by id_of(pos).  Returns block index.
    // tri / tri_code: the block's own 16-stride triangle and the codes of its entries (refill recipe), or null.
    template <class IdOf>
    i64 push_block(int rows, const int32_t *dest, const int32_t *init, std::vector<Triple> &src, IdOf id_of, const double *tri,
                   const int32_t *tri_code, i64 level, bool own_task)
    {
        // distinct sources in ascending pos order.  Long lists (the rows of a supernode share their sources) are not
        // sorted as triples: the distinct positions are collected through the stamp array, sorted alone (an order of
        // magnitude fewer), and every triple then finds its source by rank.
        bool ranked = src.size() > 64;
        if (ranked)
            for (const Triple &x : src)
                if (x.pos < 0 || x.pos >= n_) { ranked = false; break; }
        i64 distinct = 0;
        if (ranked) {
            if (rank_.empty()) rank_.assign(n_, 0);
            ++stamp_id_;
            order_.clear();
            for (const Triple &x : src)
                if (stamp_[x.pos] != stamp_id_) {
                    stamp_[x.pos] = stamp_id_;
                    order_.push_back(x.pos);
                }
            std::sort(order_.begin(), order_.end());
            for (size_t k = 0; k < order_.size(); ++k) rank_[order_[k]] = (int32_t)k;
            distinct = (i64)order_.size();
        } else {
            sort_triples(src);
            for (size_t k = 0; k < src.size(); ++k) distinct += (k == 0 || src[k].pos != src[k - 1].pos);
        }
        // long source lists always take the matrix-core path: it is pipelined and can be split over wavefronts
        const int rc = distinct >= split_sources_ ? 16 : class_of(rows);
        const i64 b = (i64)P_.hdr.size();
        BlockHeader h{rc, 0, tri ? 1 : 0, 0, (i64)P_.src_idx.size(), (i64)P_.coef.size()};
        for (int r = 0; r < rc; ++r) P_.src_idx.push_back(r < rows ? dest[r] : kNone);
        for (int r = 0; r < rc; ++r) P_.src_idx.push_back(r < rows ? init[r] : kNone);
        i64 ns = distinct;
        const size_t cbase = P_.coef.size();
        if (ranked) {
            for (int32_t pos : order_) P_.src_idx.push_back(id_of(pos));
            grow_coef(cbase + (size_t)distinct * rc);
            for (const Triple &x : src) {
                const size_t at = cbase + (size_t)rank_[x.pos] * rc + x.slot;
                P_.coef[at] += x.val;  // duplicates add up
                set_code(at, x.code);
            }
        } else {
            for (size_t k = 0; k < src.size();) {
                const int32_t pos = src[k].pos;
                P_.src_idx.push_back(id_of(pos));
                const size_t base = P_.coef.size();
                grow_coef(base + rc);
                for (; k < src.size() && src[k].pos == pos; ++k) {
                    P_.coef[base + src[k].slot] += src[k].val;  // duplicates add up
                    set_code(base + src[k].slot, src[k].code);
                }
            }
        }
        if (rc == 16 && ns > 0)  // the matrix-core path consumes sources sixteen at a time: pad with zero coefficients
            for (; ns % 16 != 0; ++ns) {
                P_.src_idx.push_back(P_.src_idx.back());
                grow_coef(P_.coef.size() + rc);
            }
        h.nsrc = (int32_t)ns;
        P_.hdr.push_back(h);
        if (tri && rc < 16) {
            const size_t base = P_.coef.size();
            grow_coef(base + (size_t)rc * rc);
            for (int r = 0; r < rc; ++r)
                for (int t = 0; t <= r; ++t) {
                    P_.coef[base + (size_t)r * rc + t] = (r < rows) ? tri[r * 16 + t] : (t == r ? 1.0 : 0.0);
                    if (r < rows && tri_code) set_code(base + (size_t)r * rc + t, tri_code[r * 16 + t]);
                }
        } else if (tri) {
            // inverse of the block's 16 x 16 lower triangle (unit rows for padding), in matrix-core order
            double T[16][16];
            for (int r = 0; r < 16; ++r)
                for (int t = 0; t < 16; ++t) T[r][t] = (r < rows && t <= r) ? tri[r * 16 + t] : (t == r ? 1.0 : 0.0);
            const size_t base = P_.coef.size();
            grow_coef(base + 256);
            store_tri16_inverse(T, P_.coef.data() + base);
            if (!tri_code) recipe_ok_ = false;
            if (recipe_ok_) {
                BlockProgram::Refill::Tri16 rec;
                rec.coef_base = (i64)base;
                rec.rows = rows;
                for (int r = 0; r < 16; ++r)
                    for (int t = 0; t <= r; ++t) rec.src[r * (r + 1) / 2 + t] = r < rows ? tri_code[r * 16 + t] : -1;
                P_.refill.tri16.push_back(rec);
            }
        }
        P_.padded_entries += ns * rc;
        if (own_task) {
            close_task();
            task_first_blk_.push_back(b);
            task_level_.push_back((int32_t)level);
            task_nblk_.push_back(1);
        }
        return b;
    }

    void close_task() { cur_task_ = -1; }

    // ---- normal runs: blocks of up to 16 unknowns solved in registers -----------------------------------
    void emit_normal_run(i64 t0, i64 t1)
    {
        for (i64 a = t0; a < t1;) {
            i64 take = 1;
            const i64 left = t1 - a;
            for (i64 cand : {(i64)16, (i64)4, (i64)2}) {
                if (cand > left && class_of(left) != cand) continue;
                const i64 rows = std::min(cand, left);
                if (rows <= 1) continue;
                if (efficient(a, a + rows)) {
                    take = rows;
                    break;
                }
            }
            emit_normal_block(a, a + take);
            a += take;
        }
    }

    // does grouping positions [a,b) waste less than ~40 % of the multiply-adds?
    bool efficient(i64 a, i64 b)
    {
        const i64 rows = b - a;
        const int rc = class_of(rows);
        ++stamp_id_;
        i64 uni = 0, ext = 0;
        for (i64 t = a; t < b; ++t) {
            const i64 r = row_at(t);
            for (i64 p = G_.ptr[r]; p < G_.ptr[r + 1]; ++p) {
                const i64 s = G_.idx[p];
                if (pos_of(s) >= a) continue;
                ++ext;
                if (stamp_[s] != stamp_id_) {
                    stamp_[s] = stamp_id_;
                    ++uni;
                }
            }
        }
        if (uni == 0) return true;
        // a matrix-core block costs the same whether 5 or 16 of its rows are used, and it is cheap: accept more padding
        return (double)ext >= (rc == 16 ? 0.3 : 0.6) * (double)rc * (double)uni;
    }

    void emit_normal_block(i64 a, i64 b)
    {
        const int rows = (int)(b - a);
        int32_t dest[16];
        double tri[256];
        int32_t tri_code[256];
        std::memset(tri, 0, sizeof(tri));
        std::fill(tri_code, tri_code + 256, -1);
        src_.clear();
        i64 need = 0;
        bool chained = false;
        for (i64 t = a; t < b; ++t) {
            const i64 r = row_at(t);
            const int slot = (int)(t - a);
            dest[slot] = (int32_t)r;
            tri[slot * 16 + slot] = G_.diag[r];
            tri_code[slot * 16 + slot] = diag_code(r);
            for (i64 p = G_.ptr[r]; p < G_.ptr[r + 1]; ++p) {
                const i64 s = G_.idx[p];
                const i64 ps = pos_of(s);
                if (ps >= a) {
                    tri[slot * 16 + (ps - a)] += G_.val[p];
                    if (tri_code[slot * 16 + (ps - a)] != -1) recipe_ok_ = false;  // duplicate entries add up
                    tri_code[slot * 16 + (ps - a)] = (int32_t)p;
                } else {
                    src_.push_back(Triple{(int32_t)ps, slot, G_.val[p], (int32_t)p});
                    if (cur_task_ >= 0 && task_of_[s] == cur_task_) chained = true;
                    else need = std::max<i64>(need, avail_[s] + 1);
                }
            }
        }
        // a wavefront runs at most max_blocks_ dependent blocks back to back: every block costs a chain of dependent
        // global loads (header, ids, rows), so longer tasks stretch the level they run in
        // blocks with many sources are split over the wavefronts of a workgroup on the device: a task of their own
        const bool wide = distinct_sources(src_) >= split_sources_;
        const bool fits = !wide && cur_task_ >= 0 && task_nblk_[cur_task_] < max_blocks_;
        i64 level;
        if (fits && need <= task_level_[cur_task_] && (chained || need == task_level_[cur_task_])) {
            level = task_level_[cur_task_];
            task_nblk_[cur_task_]++;
        } else {
            if (cur_task_ >= 0 && chained) need = std::max<i64>(need, task_level_[cur_task_] + 1);
            level = need;
            cur_task_ = (i64)task_level_.size();
            task_first_blk_.push_back((i64)P_.hdr.size());
            task_level_.push_back((int32_t)level);
            task_nblk_.push_back(1);
        }
        const i64 keep_task = cur_task_;
        push_block(rows, dest, dest, src_, [&](int32_t pos) { return (int32_t)row_at(pos); }, tri, tri_code, level, false);
        cur_task_ = keep_task;
        for (i64 t = a; t < b; ++t) {
            avail_[row_at(t)] = (int32_t)level;
            task_of_[row_at(t)] = (int32_t)cur_task_;
        }
        if (wide) close_task();  // nothing is appended to it
    }

    i64 distinct_sources(const std::vector<Triple> &src)
    {
        ++stamp_id_;
        i64 k = 0;
        for (const Triple &x : src) {
            const i64 key = x.pos;  // positions are unique per source row
            if (stamp_[key] != stamp_id_) {
                stamp_[key] = stamp_id_;
                ++k;
            }
        }
        return k;
    }

    // ---- the same blocks from sources that are ALREADY RANKED (long chains: the ranking of a group of rows is a pure
    //      function of the gather form, done on the host threads; only these appends are sequential) -----------------
    struct RankedGroup {
        int rows = 0;
        int32_t dest[16], init[16];
        std::vector<Triple> src;     // pos = rank of the source among the group's distinct sources; by part when parts > 1
        std::vector<int32_t> order;  // the distinct sources, ascending: what id_of() is asked for
        std::vector<i64> part_off;   // parts + 1 offsets into src
        i64 need = 0, parts = 1, per = 0;
    };

    // one block of its own task, no triangle: exactly what push_block(.., nullptr, nullptr, level, true) appends for the
    // sources order[0, distinct) and the triples src[0, nsrc) whose ranks start at rank_base
    template <class IdOf>
    void append_block(int rows, const int32_t *dest, const int32_t *init, const int32_t *order, i64 distinct, const Triple *src, size_t nsrc,
                      int32_t rank_base, IdOf id_of, i64 level, bool twin = false)
    {
        const int rc = distinct >= split_sources_ ? 16 : class_of(rows);
        const i64 b = (i64)P_.hdr.size();
        BlockHeader h{rc, 0, 0, 0, (i64)P_.src_idx.size(), (i64)P_.coef.size()};
        for (int r = 0; r < rc; ++r) P_.src_idx.push_back(r < rows ? dest[r] : kNone);
        for (int r = 0; r < rc; ++r) P_.src_idx.push_back(r < rows ? init[r] : kNone);
        i64 ns = distinct;
        const size_t cbase = P_.coef.size();
        for (i64 k = 0; k < distinct; ++k) P_.src_idx.push_back(id_of(order[k]));
        grow_coef(cbase + (size_t)distinct * rc);
        for (size_t k = 0; k < nsrc; ++k) {
            const size_t at = cbase + (size_t)(src[k].pos - rank_base) * rc + src[k].slot;
            P_.coef[at] += src[k].val;  // duplicates add up
            set_code(at, src[k].code);
        }
        if (rc == 16 && ns > 0)
            for (; ns % 16 != 0; ++ns) {
                P_.src_idx.push_back(P_.src_idx.back());
                grow_coef(P_.coef.size() + rc);
            }
        h.nsrc = (int32_t)ns;
        P_.hdr.push_back(h);
        P_.padded_entries += ns * rc;
        close_task();
        task_first_blk_.push_back(b);
        task_level_.push_back((int32_t)level);
        task_nblk_.push_back(1);
        task_tile_.resize(task_level_.size(), 0);
        task_tile_.back() = twin ? 1 : 0;
    }


    // one row tile of an item: a ranked 16-row group, or one PART of it (sources order[0, cnt), triples whose ranks start at rank_base)
    struct TileView {
        bool present = false;  // (a tile with fewer parts than its neighbours is absent from the later part items)
        int rows = 0;
        const int32_t *dest = nullptr, *init = nullptr, *order = nullptr;
        i64 cnt = 0;
        const Triple *src = nullptr;
        size_t nsrc = 0;
        int32_t rank_base = 0;
    };

    // ONE WIDE ITEM (sparse.hpp) over up to four tiles of one chain phase, all at `level`: the union of their source lists,
    // a 4-bit mask per source (which tiles use it), sources ordered by mask (then by position) so that the K-steps of a
    // tile's zero blocks bunch together and are skipped as whole K-steps.  The same arithmetic is also emitted as the
    // tiles' own 16-row blocks (the item's TWINS, task classes 2 / 4 / 6); a level runs one view or the other
    // (dev::solve_items or the block kernel), decided at launch time.
    template <class IdOf>
    void append_item(const TileView *g, int mt, IdOf id_of, i64 level)
    {
        // union of the sorted lists, with masks
        item_union_.clear();
        item_mask_.clear();
        i64 at[4] = {0, 0, 0, 0};
        while (true) {
            int32_t next = INT32_MAX;
            for (int q = 0; q < mt; ++q)
                if (g[q].present && at[q] < g[q].cnt) next = std::min(next, g[q].order[at[q]]);
            if (next == INT32_MAX) break;
            uint8_t m = 0;
            for (int q = 0; q < mt; ++q)
                if (g[q].present && at[q] < g[q].cnt && g[q].order[at[q]] == next) {
                    m |= (uint8_t)(1 << q);
                    ++at[q];
                }
            item_union_.push_back(next);
            item_mask_.push_back(m);
        }
        const i64 ns = (i64)item_union_.size();
        // order by mask: the bucket of mask 15 first, then the others by descending value; positions ascending inside a bucket
        i64 bucket[17] = {0};
        for (uint8_t m : item_mask_) bucket[16 - m]++;  // (mask m -> bucket 15 - m, shifted by one for the prefix sum)
        for (int k = 1; k <= 16; ++k) bucket[k] += bucket[k - 1];
        item_perm_.resize((size_t)ns);
        for (i64 u = 0; u < ns; ++u) item_perm_[(size_t)u] = (int32_t)bucket[15 - item_mask_[(size_t)u]]++;
        const i64 ns4 = (ns + 3) / 4 * 4, ks = ns4 / 4;
        const i64 nwin = (ks + 15) / 16;
        // which row tiles have entries in each K-step (the OR of its four sources' masks); only those 16 x 4 coefficient blocks
        // are stored, K-step by K-step, tiles ascending inside a K-step
        item_kmask_.assign((size_t)ks, 0);
        for (i64 u = 0; u < ns; ++u) item_kmask_[(size_t)(item_perm_[(size_t)u] / 4)] |= item_mask_[(size_t)u];
        item_kfirst_.resize((size_t)ks + 1);
        item_kfirst_[0] = 0;
        for (i64 k = 0; k < ks; ++k) item_kfirst_[(size_t)k + 1] = item_kfirst_[(size_t)k] + __builtin_popcount(item_kmask_[(size_t)k]);
        const i64 nblocks = item_kfirst_[(size_t)ks];
        const size_t cbase = P_.coef.size();
        grow_coef(cbase + (size_t)nblocks * 64);
        for (int q = 0; q < mt; ++q) {
            if (!g[q].present) continue;
            // rank inside the tile's list -> rank inside the union (both lists ascending)
            item_map_.resize((size_t)g[q].cnt);
            size_t u = 0;
            for (i64 e = 0; e < g[q].cnt; ++e) {
                while (item_union_[u] != g[q].order[e]) ++u;
                item_map_[(size_t)e] = item_perm_[u];
            }
            const unsigned below = (1u << q) - 1u;
            for (size_t k = 0; k < g[q].nsrc; ++k) {
                const Triple &x = g[q].src[k];
                const i64 e = item_map_[(size_t)(x.pos - g[q].rank_base)];
                const i64 blk = item_kfirst_[(size_t)(e / 4)] + __builtin_popcount(item_kmask_[(size_t)(e / 4)] & below);
                const size_t c = cbase + (size_t)blk * 64 + (size_t)(e % 4) * 16 + (size_t)x.slot;
                P_.coef[c] += x.val;  // duplicates add up
                set_code(c, x.code);
            }
        }
        WideItem it{mt, (int32_t)ks, (i64)cbase, (i64)P_.src_idx.size(), (i64)P_.item_masks.size()};
        for (int q = 0; q < mt; ++q)
            for (int r = 0; r < 16; ++r) P_.src_idx.push_back(g[q].present && r < g[q].rows ? g[q].dest[r] : kNone);
        for (int q = 0; q < mt; ++q)
            for (int r = 0; r < 16; ++r) P_.src_idx.push_back(g[q].present && r < g[q].rows ? g[q].init[r] : kNone);
        const size_t in0 = P_.src_idx.size();
        P_.src_idx.resize(in0 + (size_t)ns4, 0);
        for (i64 u = 0; u < ns; ++u) P_.src_idx[in0 + (size_t)item_perm_[(size_t)u]] = id_of(item_union_[(size_t)u]);
        for (i64 u = ns; u < ns4; ++u) P_.src_idx[in0 + (size_t)u] = P_.src_idx[in0 + (size_t)ns - 1];  // padding: zero coefficients
        for (i64 w = 0; w < nwin; ++w) {  // per window of 16 K-steps: 4 mask bits per K-step (two words), then the blocks before the window
            uint64_t m = 0;
            for (i64 k = 16 * w; k < std::min(ks, 16 * w + 16); ++k) m |= (uint64_t)item_kmask_[(size_t)k] << (4 * (k - 16 * w));
            P_.item_masks.push_back((uint32_t)(m & 0xffffffffu));
            P_.item_masks.push_back((uint32_t)(m >> 32));
            P_.item_masks.push_back((uint32_t)item_kfirst_[(size_t)(16 * w)]);
            P_.item_masks.push_back(0);
        }
        P_.item_padded_entries += 64 * nblocks;
        P_.items.push_back(it);
        item_level_.push_back((int32_t)level);
    }

    // Up to four ranked 16-row groups of one chain phase, all gathered at `level`: their blocks (partial sums into scratch
    // rows plus a combine block at level + 1 for source lists longer than max_block_sources_) and, over the same rows and
    // the same partial sums, the wide items -- one per part -- whose twins those blocks are.  Returns the level at which
    // every destination row is complete.
    template <class IdOf>
    i64 emit_tiles(const RankedGroup *const *g, int mt, IdOf id_of, i64 level)
    {
        // An item pays where its wavefronts have long runs of K-steps ahead of them, most of them with work for every row
        // tile (measured: the chains of the 2-D Laplacian 1024^2 average 30 K-steps per item, and lsolve's row tiles share
        // half of their union list -- the twin blocks are faster there; 3-D 80^3: items 12 % faster overall).
        bool item = false;
        if (item_rows_ >= 32 && mt >= 2) {
            i64 longest = 0, parts = 1, own = 0, dense = 0;
            for (int q = 0; q < mt; ++q) {
                longest = std::max<i64>(longest, std::min<i64>((i64)g[q]->order.size(), g[q]->parts > 1 ? g[q]->per : INT32_MAX));
                parts = std::max(parts, g[q]->parts);
                own += (i64)g[q]->order.size();
            }
            if (longest >= item_min_sources_) {
                for (i64 p = 0; p < parts; ++p) {  // size of the union list of every part
                    const int32_t *at[4], *end[4];
                    for (int q = 0; q < mt; ++q) {
                        const RankedGroup &G = *g[q];
                        const i64 ns = (i64)G.order.size();
                        const i64 first = G.parts == 1 ? 0 : p * G.per, cnt = p >= G.parts ? 0 : G.parts == 1 ? ns : std::min(G.per, ns - first);
                        at[q] = G.order.data() + std::min(first, ns);
                        end[q] = at[q] + cnt;
                    }
                    while (true) {
                        int32_t next = INT32_MAX;
                        for (int q = 0; q < mt; ++q)
                            if (at[q] < end[q]) next = std::min(next, *at[q]);
                        if (next == INT32_MAX) break;
                        for (int q = 0; q < mt; ++q)
                            if (at[q] < end[q] && *at[q] == next) ++at[q];
                        dense += mt;
                    }
                }
                item = (double)own >= item_min_fill_ * (double)dense;
            }
        }
        i64 max_parts = 1, done = level;
        std::vector<int32_t> pdest[4];
        int32_t none[16];
        std::fill(none, none + 16, kNone);
        for (int q = 0; q < mt; ++q) {
            const RankedGroup &G = *g[q];
            const i64 ns = (i64)G.order.size();
            max_parts = std::max(max_parts, G.parts);
            if (G.parts == 1) {
                append_block(G.rows, G.dest, G.init, G.order.data(), ns, G.src.data(), G.src.size(), 0, id_of, level, item);
                continue;
            }
            const int rows = G.rows;
            pdest[q].resize((size_t)G.parts * rows);
            for (i64 p = 0; p < G.parts; ++p) {
                for (int r = 0; r < rows; ++r) pdest[q][(size_t)p * rows + r] = ~(int32_t)(P_.scratch_rows++);
                const i64 first = p * G.per, cnt = std::min(G.per, ns - first);
                append_block(rows, &pdest[q][(size_t)p * rows], p == 0 ? G.init : none, G.order.data() + first, cnt, G.src.data() + G.part_off[p],
                             (size_t)(G.part_off[p + 1] - G.part_off[p]), (int32_t)first, id_of, level, item);
            }
            // combine: dest = partial_0 + sum_{p>=1} partial_p   (acc -= (-1) * partial); runs in both views
            std::vector<Triple> comb;
            for (i64 p = 1; p < G.parts; ++p)
                for (int r = 0; r < rows; ++r) comb.push_back(Triple{(int32_t)((p - 1) * rows + r), r, -1.0});
            const int32_t *p1 = pdest[q].data() + rows;
            push_block(rows, G.dest, pdest[q].data(), comb, [&](int32_t o) { return p1[o]; }, nullptr, nullptr, level + 1, true);
            done = level + 1;
        }
        if (item)
            for (i64 p = 0; p < max_parts; ++p) {
                TileView v[4];
                for (int q = 0; q < mt; ++q) {
                    const RankedGroup &G = *g[q];
                    if (p >= G.parts) continue;
                    const i64 ns = (i64)G.order.size();
                    const i64 first = G.parts == 1 ? 0 : p * G.per, cnt = G.parts == 1 ? ns : std::min(G.per, ns - first);
                    v[q].present = true;
                    v[q].rows = G.rows;
                    v[q].dest = G.parts == 1 ? G.dest : &pdest[q][(size_t)p * G.rows];
                    v[q].init = (G.parts == 1 || p == 0) ? G.init : none;
                    v[q].order = G.order.data() + first;
                    v[q].cnt = cnt;
                    v[q].src = G.parts == 1 ? G.src.data() : G.src.data() + G.part_off[p];
                    v[q].nsrc = G.parts == 1 ? G.src.size() : (size_t)(G.part_off[p + 1] - G.part_off[p]);
                    v[q].rank_base = (int32_t)first;
                }
                append_item(v, mt, id_of, level);
            }
        return done;
    }

    // distinct sources and ranks of a group's triples (pos = processing position on entry, rank on return), cut into
    // parts of at most max_block_sources_ sources; scratch: a stamp and a rank per unknown, private to the calling thread
    void rank_group(RankedGroup &g, std::vector<i64> &stamp, i64 &stamp_id, std::vector<int32_t> &rank) const
    {
        ++stamp_id;
        g.order.clear();
        for (const Triple &x : g.src)
            if (stamp[x.pos] != stamp_id) {
                stamp[x.pos] = stamp_id;
                g.order.push_back(x.pos);
            }
        std::sort(g.order.begin(), g.order.end());
        for (size_t k = 0; k < g.order.size(); ++k) rank[g.order[k]] = (int32_t)k;
        for (Triple &x : g.src) x.pos = rank[x.pos];
        const i64 ns = (i64)g.order.size();
        g.parts = ns <= max_block_sources_ ? 1 : (ns + max_block_sources_ - 1) / max_block_sources_;
        g.per = (ns + g.parts - 1) / g.parts;
        g.part_off.clear();
        if (g.parts > 1) {  // counting sort by part, the order inside a part kept
            g.part_off.assign((size_t)g.parts + 1, 0);
            for (const Triple &x : g.src) g.part_off[(size_t)(x.pos / g.per) + 1]++;
            for (i64 p = 0; p < g.parts; ++p) g.part_off[p + 1] += g.part_off[p];
            std::vector<i64> fill(g.part_off.begin(), g.part_off.end() - 1);
            std::vector<Triple> sorted(g.src.size());
            for (const Triple &x : g.src) sorted[(size_t)fill[(size_t)(x.pos / g.per)]++] = x;
            g.src.swap(sorted);
        }
    }

    // ---- gather into dest with the source list cut into partial sums of at most kMaxBlockSources sources ----------
    // src triples: pos = ordinal understood by id_of.  Every block is its own task.  Returns the level at which dest
    // is complete.
    template <class IdOf>
    i64 emit_gather(int rows, const int32_t *dest, const int32_t *init, std::vector<Triple> &src, IdOf id_of, i64 level)
    {
        // the common case needs the NUMBER of distinct sources only (push_block orders them itself)
        bool in_range = true;
        for (const Triple &x : src)
            if (x.pos < 0 || x.pos >= n_) { in_range = false; break; }
        i64 ns = 0;
        if (in_range) {
            ns = distinct_sources(src);
        } else {
            sort_triples(src);
            for (size_t k = 0; k < src.size(); ++k) ns += (k == 0 || src[k].pos != src[k - 1].pos);
        }
        if (ns <= max_block_sources_) {
            push_block(rows, dest, init, src, id_of, nullptr, nullptr, level, true);
            return level;
        }
        const i64 parts = (ns + max_block_sources_ - 1) / max_block_sources_;
        const i64 per = (ns + parts - 1) / parts;
        std::vector<int32_t> pdest((size_t)parts * rows);
        // part of a triple = rank of its source / per.  With in-range positions the ranks come from the sorted list of
        // DISTINCT sources and the triples are bucketed in one pass; otherwise they are sorted and cut.
        std::vector<std::vector<Triple>> chunks((size_t)parts);
        if (in_range) {
            if (rank_.empty()) rank_.assign(n_, 0);
            ++stamp_id_;
            order_.clear();
            for (const Triple &x : src)
                if (stamp_[x.pos] != stamp_id_) {
                    stamp_[x.pos] = stamp_id_;
                    order_.push_back(x.pos);
                }
            std::sort(order_.begin(), order_.end());
            for (size_t k = 0; k < order_.size(); ++k) rank_[order_[k]] = (int32_t)k;
            for (const Triple &x : src) chunks[(size_t)(rank_[x.pos] / per)].push_back(x);
        } else {
            sort_triples(src);
            size_t k = 0;
            for (i64 p = 0; p < parts; ++p) {
                i64 taken = 0;
                while (k < src.size()) {
                    const bool new_src = (chunks[p].empty() || src[k].pos != chunks[p].back().pos);
                    if (new_src && taken == per) break;
                    taken += new_src;
                    chunks[p].push_back(src[k++]);
                }
            }
        }
        for (i64 p = 0; p < parts; ++p) {
            int32_t none[16];
            for (int r = 0; r < rows; ++r) {
                pdest[(size_t)p * rows + r] = ~(int32_t)(P_.scratch_rows++);
                none[r] = kNone;
            }
            push_block(rows, &pdest[(size_t)p * rows], p == 0 ? init : none, chunks[p], id_of, nullptr, nullptr, level, true);
        }
        // combine: dest = partial_0 + sum_{p>=1} partial_p   (acc -= (-1) * partial)
        std::vector<Triple> comb;
        for (i64 p = 1; p < parts; ++p)
            for (int r = 0; r < rows; ++r) comb.push_back(Triple{(int32_t)((p - 1) * rows + r), r, -1.0});
        const int32_t *p1 = pdest.data() + rows;
        push_block(rows, dest, pdest.data(), comb, [&](int32_t o) { return p1[o]; }, nullptr, nullptr, level + 1, true);
        return level + 1;
    }

    // ---- long dense chains: Y = X_J - L[J,ext] X_ext ; X_J = M^-1 Y ------------------------------------------
    // Both phases are emitted as 16-row blocks and, over the same rows, as WIDE ITEMS of up to item_rows_ unknowns (the
    // blocks are then the item's twins: append_item).  The rows of phase A are independent of each other, so they are
    // free to be GROUPED in any order: rows with similar source lists go into the same group (similar_row_order), which
    // keeps the union list of a group -- every row of a group pays for every source of the union -- close to the rows'
    // own lists (lsolve / usolve: the rows of a supernode do not share their sources).
    void emit_big_chain(i64 t0, i64 t1)
    {
        close_task();
        StageClock ck;
        const i64 w = t1 - t0;
        P_.big_chains++;
        P_.inverted_rows += w;
        // dense diagonal block in position coordinates, then its inverse (forward substitution per column)
        std::vector<double> Minv;
        if (G_.placeholder) Minv.assign((size_t)w * w, 0.0);  // a programme emitted ahead of its values: the refill computes the inverse
        else chain_inverse(G_, t0, t1, Minv);
        const i64 pool_off = P_.refill.pool_size;
        P_.refill.chains.push_back(BlockProgram::Refill::Chain{t0, t1, pool_off});
        P_.refill.pool_size += w * w;
        ck.lap(4);
        const i64 y0 = P_.scratch_rows;
        P_.scratch_rows += w;
        const i64 tiles_per_item = std::max<i64>(1, item_rows_ / 16);
        // phase A per group of 16 unknowns: all groups independent
        std::vector<i64> rows_in_order((size_t)w);
        for (i64 k = 0; k < w; ++k) rows_in_order[(size_t)k] = t0 + k;
        if (sort_rows_ && w > 16) similar_row_order(t0, t1, rows_in_order);
        i64 ready = 0;
        {
            // the triples of a group, its distinct sources and their ranks are a pure function of the gather form (and of
            // avail_, which this chain does not change before its end): batches of groups are prepared on the host
            // threads, then appended in order -- the programme does not depend on the number of threads
            const i64 ngroups = (w + 15) / 16;
            const i64 T = std::max<i64>(1, std::min<i64>(emit_threads_, ngroups));
            if ((i64)tstamp_.size() < T) {
                tstamp_.resize((size_t)T);
                trank_.resize((size_t)T);
                tstamp_id_.resize((size_t)T, 0);
            }
            const i64 batch = 4 * T;  // (a multiple of 4: the groups of an item are prepared together)
            std::vector<RankedGroup> work((size_t)std::min(batch, ngroups));
            for (i64 g0 = 0; g0 < ngroups; g0 += batch) {
                const i64 g1 = std::min(ngroups, g0 + batch);
                auto prepare = [&](i64 k) {
                    if (tstamp_[k].empty()) {
                        tstamp_[k].assign(n_, -1);
                        trank_[k].assign(n_, 0);
                    }
                    for (i64 g = g0 + k; g < g1; g += T) {
                        RankedGroup &G = work[(size_t)(g - g0)];
                        const i64 a = 16 * g, b = std::min(a + 16, w);
                        G.rows = (int)(b - a);
                        G.src.clear();
                        G.need = 0;
                        for (i64 q = a; q < b; ++q) {
                            const i64 t = rows_in_order[(size_t)q];
                            const i64 r = row_at(t);
                            const int slot = (int)(q - a);
                            G.dest[slot] = ~(int32_t)(y0 + (t - t0));
                            G.init[slot] = (int32_t)r;
                            for (i64 p = G_.ptr[r]; p < G_.ptr[r + 1]; ++p) {
                                const i64 s = G_.idx[p];
                                const i64 ps = pos_of(s);
                                if (ps >= t0) continue;
                                G.src.push_back(Triple{(int32_t)ps, slot, G_.val[p], (int32_t)p});
                                G.need = std::max<i64>(G.need, avail_[s] + 1);
                            }
                        }
                        rank_group(G, tstamp_[k], tstamp_id_[k], trank_[k]);
                    }
                };
                if (T == 1) {
                    prepare(0);
                } else {
                    std::vector<std::thread> pool;
                    for (i64 k = 1; k < T; ++k) pool.emplace_back(prepare, k);
                    prepare(0);
                    for (std::thread &th : pool) th.join();
                }
                ck.lap(5);
                auto id_of = [&](int32_t pos) { return (int32_t)row_at(pos); };
                for (i64 g = g0; g < g1; g += tiles_per_item) {
                    const int mt = (int)std::min<i64>(tiles_per_item, g1 - g);
                    const RankedGroup *grp[4];
                    i64 level = 0;
                    for (int q = 0; q < mt; ++q) {
                        grp[q] = &work[(size_t)(g + q - g0)];
                        level = std::max(level, grp[q]->need);
                    }
                    ready = std::max(ready, emit_tiles(grp, mt, id_of, level));  // (the tiles of an item share its level)
                }
                ck.lap(6);
            }
        }
        // phase B: X_J = Minv * Y, all groups independent.  The sources of a group are the scratch rows 0 .. b - t0 - 1, every
        // one of them used: a source's rank is its number, and triples generated source by source are already grouped by part
        i64 final_level = ready + 1;
        auto id_of_y = [&](int32_t o) { return ~(int32_t)(y0 + o); };
        for (i64 a0 = t0; a0 < t1; a0 += 16 * tiles_per_item) {
            const int mt = (int)std::min<i64>(tiles_per_item, (t1 - a0 + 15) / 16);
            const RankedGroup *grp[4];
            for (int q = 0; q < mt; ++q) {
                const i64 a = a0 + 16 * q, b = std::min(a + 16, t1);
                RankedGroup &G = chain_groups_[q];
                grp[q] = &G;
                G.rows = (int)(b - a);
                for (i64 t = a; t < b; ++t) {
                    G.dest[t - a] = (int32_t)row_at(t);
                    G.init[t - a] = kNone;
                }
                const i64 ns = b - t0;
                G.order.resize((size_t)ns);
                for (i64 u = 0; u < ns; ++u) G.order[(size_t)u] = (int32_t)u;
                G.parts = ns <= max_block_sources_ ? 1 : (ns + max_block_sources_ - 1) / max_block_sources_;
                G.per = (ns + G.parts - 1) / G.parts;
                G.src.clear();
                G.part_off.assign((size_t)G.parts + 1, 0);
                for (i64 u = 0; u < ns; ++u) {
                    for (i64 t = std::max(a, t0 + u); t < b; ++t)
                        G.src.push_back(Triple{(int32_t)u, (int)(t - a), -Minv[(size_t)(t - t0) * w + u], pool_code(pool_off + (t - t0) * w + u)});
                    G.part_off[(size_t)(u / G.per) + 1] = (i64)G.src.size();
                }
                for (i64 p = 0; p < G.parts; ++p) G.part_off[p + 1] = std::max(G.part_off[p + 1], G.part_off[p]);
            }
            final_level = std::max(final_level, emit_tiles(grp, mt, id_of_y, ready + 1));
        }
        for (i64 t = t0; t < t1; ++t) {
            avail_[row_at(t)] = (int32_t)final_level;
            task_of_[row_at(t)] = -2;  // never "chained" to a running task
        }
        close_task();
        ck.lap(5);
        for (int k = 3; k <= 6; ++k) clock_.acc[k] += ck.acc[k];
    }

    // Order of the rows of a chain (positions [t0, t1)) for phase A, such that the rows of one item have similar source
    // lists.  Greedy: an item starts from the unassigned row with the longest list and takes, one at a time, the
    // unassigned row whose list adds the fewest new sources to the item's union (ties: the longer list, then the lower
    // position).  The lists of a supernode's rows are unions of whole runs of consecutive positions (descendant
    // supernodes), so they are compared as lists of SEGMENTS -- maximal runs of consecutive source positions that belong
    // to the same set of rows -- weighted by their length: tens of segments per row instead of hundreds of sources.
    // Pure function of the gather form.
    void similar_row_order(i64 t0, i64 t1, std::vector<i64> &out)
    {
        const i64 w = t1 - t0, R = 16;  // (the rows of a block; an item takes four consecutive blocks)
        if (rank_.empty()) rank_.assign(n_, 0);
        // distinct sources of the chain, ascending
        ++stamp_id_;
        order_.clear();
        bool all_same = true;
        i64 first_len = -1;
        for (i64 k = 0; k < w; ++k) {
            const i64 r = row_at(t0 + k);
            i64 len = 0;
            for (i64 p = G_.ptr[r]; p < G_.ptr[r + 1]; ++p) {
                const i64 ps = pos_of(G_.idx[p]);
                if (ps >= t0) continue;
                ++len;
                if (stamp_[ps] != stamp_id_) {
                    stamp_[ps] = stamp_id_;
                    order_.push_back((int32_t)ps);
                    if (k > 0) all_same = false;  // a source the first row does not have
                }
            }
            if (k == 0) first_len = len;
            else if (len != first_len) all_same = false;
        }
        if (all_same || order_.empty()) return;  // (ltsolve / utsolve on a supernode: every row has the same list)
        std::sort(order_.begin(), order_.end());
        const i64 nu = (i64)order_.size();
        for (i64 k = 0; k < nu; ++k) rank_[order_[(size_t)k]] = (int32_t)k;
        // membership hash per source, then segments
        std::vector<uint64_t> hash((size_t)nu, 0);
        auto mix = [](uint64_t z) {
            z += 0x9e3779b97f4a7c15ull;
            z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
            z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
            return z ^ (z >> 31);
        };
        for (i64 k = 0; k < w; ++k) {
            const i64 r = row_at(t0 + k);
            const uint64_t hk = mix((uint64_t)k);
            for (i64 p = G_.ptr[r]; p < G_.ptr[r + 1]; ++p) {
                const i64 ps = pos_of(G_.idx[p]);
                if (ps < t0) hash[(size_t)rank_[ps]] += hk;  // (a duplicate entry counts twice, in every comparison alike)
            }
        }
        std::vector<int32_t> seg_of((size_t)nu);
        std::vector<int32_t> seg_len;
        for (i64 k = 0; k < nu; ++k) {
            if (k == 0 || order_[(size_t)k] != order_[(size_t)k - 1] + 1 || hash[(size_t)k] != hash[(size_t)k - 1]) seg_len.push_back(0);
            seg_of[(size_t)k] = (int32_t)seg_len.size() - 1;
            seg_len.back()++;
        }
        const i64 nseg = (i64)seg_len.size();
        // per row: its segments (each once) and the length of its list
        std::vector<i64> row_ptr((size_t)w + 1, 0), row_len((size_t)w, 0);
        std::vector<int32_t> row_seg;
        std::vector<i64> seg_stamp((size_t)nseg, -1);
        for (i64 k = 0; k < w; ++k) {
            const i64 r = row_at(t0 + k);
            for (i64 p = G_.ptr[r]; p < G_.ptr[r + 1]; ++p) {
                const i64 ps = pos_of(G_.idx[p]);
                if (ps >= t0) continue;
                const int32_t sg = seg_of[(size_t)rank_[ps]];
                if (seg_stamp[(size_t)sg] != k) {
                    seg_stamp[(size_t)sg] = k;
                    row_seg.push_back(sg);
                    row_len[(size_t)k] += seg_len[(size_t)sg];
                }
            }
            row_ptr[(size_t)k + 1] = (i64)row_seg.size();
        }
        std::vector<char> used((size_t)w, 0), in_union((size_t)nseg, 0);
        std::vector<int32_t> touched;
        i64 at = 0;
        while (at < w) {
            i64 seed = -1;
            for (i64 k = 0; k < w; ++k)
                if (!used[(size_t)k] && (seed < 0 || row_len[(size_t)k] > row_len[(size_t)seed])) seed = k;
            touched.clear();
            auto take = [&](i64 k) {
                used[(size_t)k] = 1;
                out[(size_t)at++] = t0 + k;
                for (i64 q = row_ptr[(size_t)k]; q < row_ptr[(size_t)k + 1]; ++q)
                    if (!in_union[(size_t)row_seg[(size_t)q]]) {
                        in_union[(size_t)row_seg[(size_t)q]] = 1;
                        touched.push_back(row_seg[(size_t)q]);
                    }
            };
            take(seed);
            for (i64 m = 1; m < R && at < w; ++m) {
                i64 best = -1, best_add = 0;
                for (i64 k = 0; k < w; ++k) {
                    if (used[(size_t)k]) continue;
                    i64 add = 0;
                    for (i64 q = row_ptr[(size_t)k]; q < row_ptr[(size_t)k + 1]; ++q)
                        if (!in_union[(size_t)row_seg[(size_t)q]]) add += seg_len[(size_t)row_seg[(size_t)q]];
                    if (best < 0 || add < best_add || (add == best_add && row_len[(size_t)k] > row_len[(size_t)best])) {
                        best = k;
                        best_add = add;
                    }
                }
                take(best);
            }
            for (int32_t sg : touched) in_union[(size_t)sg] = 0;
        }
    }

    // ---- order tasks and clusters by level ------------------------------------------------------------------
    void finish()
    {
        const i64 ntasks = (i64)task_level_.size(), nclusters = (i64)cluster_level_.size();
        i64 nlev = 0;
        for (int32_t l : task_level_) nlev = std::max<i64>(nlev, l + 1);
        for (int32_t l : cluster_level_) nlev = std::max<i64>(nlev, l + 1);
        // clusters nothing else reads (the leaves of a transposed solve) all run in ONE launch at the last level
        if (nclusters) {
            std::vector<char> read(nclusters, 0);
            for (int32_t id : P_.src_idx)
                if (id >= 0 && cluster_of_[id] >= 0) read[cluster_of_[id]] = 1;
            for (i64 c = 0; c < nclusters; ++c) {
                const ClusterHeader &h = P_.clusters[c];
                for (int32_t e = 0; e < h.n_ext; ++e) {
                    const int32_t id = P_.cluster_ids[h.ids_off + e];
                    if (cluster_of_[id] >= 0) read[cluster_of_[id]] = 1;
                }
            }
            // P_.clusters / cluster_level_ are in emission (processing) order, read[] is indexed by cluster id
            for (i64 c = 0; c < nclusters; ++c)
                if (!read[emit_order_[c]]) cluster_level_[c] = (int32_t)(nlev - 1);
        }
        // inside a level, tasks are grouped by class (BlockProgram::kClasses): 0 = only small (vector-path) blocks, 1 = one
        // wavefront per (task, 64 columns) with at least one matrix-core block, 3 / 5 = one matrix-core block whose sources
        // are split over 4 / 8 wavefronts; 2, 4, 6 = the same for the row tiles of wide items
        constexpr int NC = BlockProgram::kClasses;
        task_tile_.resize((size_t)ntasks, 0);
        for (int32_t l : item_level_) nlev = std::max<i64>(nlev, l + 1);
        auto task_class = [&](i64 t) {
            const BlockHeader &h = P_.hdr[task_first_blk_[t]];
            const int tile = task_tile_[(size_t)t] ? 1 : 0;
            if (task_nblk_[t] == 1 && h.rc == 16 && h.nsrc >= split_sources_) return (h.nsrc > split8_sources_ ? 5 : 3) + tile;
            if (tile) return 2;
            for (i64 b = task_first_blk_[t]; b < task_first_blk_[t] + task_nblk_[t]; ++b)
                if (P_.hdr[b].rc == 16) return 1;
            return 0;
        };
        P_.level_class_ptr.assign(NC * nlev + 1, 0);
        for (i64 t = 0; t < ntasks; ++t) P_.level_class_ptr[NC * task_level_[t] + task_class(t) + 1]++;
        for (i64 k = 0; k < NC * nlev; ++k) P_.level_class_ptr[k + 1] += P_.level_class_ptr[k];
        P_.level_ptr.assign(nlev + 1, 0);
        for (i64 l = 0; l <= nlev; ++l) P_.level_ptr[l] = P_.level_class_ptr[NC * l];
        std::vector<i64> fill(P_.level_class_ptr.begin(), P_.level_class_ptr.end() - 1), order(ntasks);
        for (i64 t = 0; t < ntasks; ++t) order[fill[NC * task_level_[t] + task_class(t)]++] = t;
        std::vector<BlockHeader> hdr(P_.hdr.size());
        P_.task_ptr.assign(ntasks + 1, 0);
        i64 at = 0;
        for (i64 k = 0; k < ntasks; ++k) {
            const i64 t = order[k];
            P_.task_ptr[k] = at;
            for (i64 b = task_first_blk_[t]; b < task_first_blk_[t] + task_nblk_[t]; ++b, ++at) hdr[at] = P_.hdr[b];
        }
        P_.task_ptr[ntasks] = at;
        P_.hdr.swap(hdr);
        // the kernels fetch ids, coefficients and the triangle speculatively (clamped lane offsets): keep those in bounds
        P_.src_idx.resize(P_.src_idx.size() + 128, 0);
        grow_coef(P_.coef.size() + 320);
        // prologue records (sparse.hpp), in block order
        {
            constexpr int LW = BlockProgram::kLineWords;
            const i64 nblocks = (i64)P_.hdr.size();
            P_.block_record.assign(nblocks + 1, 0);
            std::vector<int> waves(nblocks, 1);
            for (i64 l = 0; l < nlev; ++l)
                for (int c = 3; c < NC; ++c)
                    for (i64 t = P_.level_class_ptr[NC * l + c]; t < P_.level_class_ptr[NC * l + c + 1]; ++t)
                        waves[P_.task_ptr[t]] = c <= 4 ? 4 : 8;  // split classes: one block per task
            for (i64 b = 0; b < nblocks; ++b) P_.block_record[b + 1] = P_.block_record[b] + waves[b] * BlockProgram::record_lines(waves[b]);
            P_.prologue.assign((size_t)(P_.block_record[nblocks] + 1) * LW, 0);  // one spare line: the kernels load whole records
            for (i64 b = 0; b < nblocks; ++b) {
                const BlockHeader &h = P_.hdr[b];
                const int32_t *ids = P_.src_idx.data() + h.src_off;
                const i64 nst = h.rc == 16 ? h.nsrc / 16 : 0;
                const int lines = BlockProgram::record_lines(waves[b]);
                for (int w = 0; w < waves[b]; ++w) {
                    int32_t *rec = P_.prologue.data() + (size_t)(P_.block_record[b] + (i64)w * lines) * LW;
                    auto word = [&](int line, int l) -> int32_t & { return rec[(size_t)l * lines + line]; };  // interleaved: lane l owns words l*lines..
                    int32_t head[8];
                    std::memcpy(head, &h, sizeof(BlockHeader));
                    for (int k = 0; k < 8; ++k) word(0, k) = head[k];
                    for (int r = 0; r < 16; ++r) {
                        word(0, 8 + r) = r < h.rc ? ids[r] : kNone;
                        word(0, 24 + r) = r < h.rc ? ids[h.rc + r] : kNone;
                    }
                    const int32_t *sidx = ids + 2 * h.rc;
                    if (h.rc == 16) {
                        // the wavefront's own stages [first, last): every source id, in order
                        // (a wavefront with more sources than its record holds fetches the later ones from the pool)
                        const i64 first = (nst * w) / waves[b], last = (nst * (w + 1)) / waves[b];
                        const i64 cnt = (last - first) * 16;
                        for (i64 e = 0; e < (i64)(lines - 1) * LW; ++e)
                            word(1 + (int)(e / LW), (int)(e % LW)) = cnt ? sidx[first * 16 + std::min(e, cnt - 1)] : 0;
                    } else {
                        for (int e = 0; e < 16; ++e) word(0, 40 + e) = e < h.nsrc ? sidx[e] : 0;  // the vector path's first batch
                    }
                }
            }
        }
        // clusters
        P_.cluster_level_ptr.assign(nlev + 1, 0);
        for (int32_t l : cluster_level_) P_.cluster_level_ptr[l + 1]++;
        for (i64 l = 0; l < nlev; ++l) P_.cluster_level_ptr[l + 1] += P_.cluster_level_ptr[l];
        std::vector<i64> cfill(P_.cluster_level_ptr.begin(), P_.cluster_level_ptr.end() - 1);
        std::vector<ClusterHeader> ch(nclusters);
        for (i64 c = 0; c < nclusters; ++c) ch[cfill[cluster_level_[c]]++] = P_.clusters[c];
        P_.clusters.swap(ch);
        // wide items, by level (emission order kept inside a level: the items of one chain stay together)
        P_.item_level_ptr.assign(nlev + 1, 0);
        for (int32_t l : item_level_) P_.item_level_ptr[l + 1]++;
        for (i64 l = 0; l < nlev; ++l) P_.item_level_ptr[l + 1] += P_.item_level_ptr[l];
        std::vector<i64> ifill(P_.item_level_ptr.begin(), P_.item_level_ptr.end() - 1);
        std::vector<WideItem> sorted(P_.items.size());
        for (size_t k = 0; k < P_.items.size(); ++k) sorted[(size_t)ifill[item_level_[k]]++] = P_.items[k];
        P_.items.swap(sorted);
    }

    const GatherForm &G_;
    const i64 n_, max_blocks_;
    const int slots_;
    i64 split_sources_ = kSplitSources, split8_sources_ = 4 * kSplitSources, big_chain_ = kBigChain, max_block_sources_ = kMaxBlockSources;
    BlockProgram P_;
    std::vector<int32_t> avail_, task_of_, cluster_of_, slot_;
    std::vector<i64> stamp_, mark_;
    i64 stamp_id_ = 0, run_id_ = 0;
    std::vector<Triple> src_;
    std::vector<int32_t> order_, rank_;  // push_block: distinct source positions and their ranks
    // emit_big_chain: private scratch of the threads that rank the groups of a chain
    i64 emit_threads_ = 0;
    i64 item_rows_ = kItemRows, sort_rows_ = 1, item_min_sources_ = kItemMinSources;
    double item_min_fill_ = 0.2;        // an item's tiles must use at least this share of its union list, on average (0.5 / 0.3 /
                                        // 0.2 / 0.1 at C3: 24.5 / 23.4 / 23.3 / 23.2 ms; C2 indifferent: blocks of zeros cost nothing)
    std::vector<char> task_tile_;       // per block task: 1 = a row tile of a wide item
    std::vector<int32_t> item_level_;   // per item, in emission order
    RankedGroup chain_groups_[4];
    std::vector<int32_t> item_union_, item_perm_, item_map_;  // append_item scratch
    std::vector<uint8_t> item_mask_, item_kmask_;
    std::vector<i64> item_kfirst_;
    std::vector<std::vector<i64>> tstamp_;
    std::vector<std::vector<int32_t>> trank_;
    std::vector<i64> tstamp_id_;
    std::vector<i64> cluster_lo_, cluster_hi_;
    std::vector<int32_t> cluster_level_;  // in emission order
    std::vector<int32_t> emit_order_;     // emission order -> cluster id (index into cluster_lo_)
    // block tasks in creation order
    std::vector<i64> task_first_blk_, task_nblk_;
    std::vector<int32_t> task_level_;
    i64 cur_task_ = -1;
    bool recipe_ok_;
    StageClock clock_;
};

}  // namespace

bool refill_block_program(const GatherForm &G, BlockProgram &P)
{
    const BlockProgram::Refill &R = P.refill;
    if (!R.valid || R.coef_src.size() != P.coef.size()) return false;
    const i64 n = G.n, nnz = (i64)G.idx.size();
    // chain inverses: the large ones one after the other (each shares its columns out over the host threads), the many
    // small ones side by side
    std::vector<double> pool((size_t)R.pool_size);
    {
        std::vector<double> Minv;
        for (const BlockProgram::Refill::Chain &c : R.chains)
            if (c.t1 - c.t0 >= 384) {
                chain_inverse(G, c.t0, c.t1, Minv);
                std::copy(Minv.begin(), Minv.end(), pool.begin() + c.pool_off);
            }
    }
    parallel_ranges((i64)R.chains.size(), 4, [&](i64 lo, i64 hi) {
        std::vector<double> Minv;
        for (i64 k = lo; k < hi; ++k) {
            const BlockProgram::Refill::Chain &c = R.chains[k];
            if (c.t1 - c.t0 >= 384) continue;
            chain_inverse(G, c.t0, c.t1, Minv);
            std::copy(Minv.begin(), Minv.end(), pool.begin() + c.pool_off);
        }
    });
    auto value = [&](int32_t code) {
        return code < nnz ? G.val[code] : code < nnz + n ? G.diag[code - nnz] : -pool[(size_t)(code - nnz - n)];
    };
    parallel_ranges((i64)R.coef_src.size(), 1 << 18, [&](i64 lo, i64 hi) {
        for (i64 e = lo; e < hi; ++e)
            if (R.coef_src[e] >= 0) P.coef[e] = value(R.coef_src[e]);
    });
    parallel_ranges((i64)R.tri16.size(), 256, [&](i64 lo, i64 hi) {
        for (i64 k = lo; k < hi; ++k) {
            const BlockProgram::Refill::Tri16 &t = R.tri16[k];
            double T[16][16];
            for (int r = 0; r < 16; ++r)
                for (int c = 0; c < 16; ++c) {
                    const int32_t code = c <= r ? t.src[r * (r + 1) / 2 + c] : -1;
                    T[r][c] = (r < t.rows && c <= r) ? (code >= 0 ? value(code) : 0.0) : (c == r ? 1.0 : 0.0);
                }
            store_tri16_inverse(T, P.coef.data() + t.coef_base);
        }
    });
    parallel_ranges((i64)R.clusters.size(), 256, [&](i64 lo, i64 hi) {
        for (i64 c = lo; c < hi; ++c) {
            const BlockProgram::Refill::Cluster &cl = R.clusters[c];
            GroupRec *rec = P.groups.data() + cl.grp_off;
            for (i64 k = 0; k <= cl.hi - cl.lo; ++k) {
                const i64 r = G.descending ? cl.hi - k : cl.lo + k;
                fill_group_values(G, r, rec);
                rec += std::max<i64>(1, (G.ptr[r + 1] - G.ptr[r] + 3) / 4);
            }
        }
    });
    return true;
}

BlockProgram build_block_program(const GatherForm &G, i64 max_task_blocks, int cluster_slots)
{
    Builder b(G, max_task_blocks, cluster_slots);
    return b.run();
}

ProgramReport check_block_program(const GatherForm &G, const BlockProgram &P, bool items_view)
{
    ProgramReport R;
    constexpr int NC = BlockProgram::kClasses;
    if ((i64)P.level_class_ptr.size() != NC * ((i64)P.level_ptr.size() - 1) + 1 || P.item_level_ptr.size() != P.level_ptr.size()) {
        R.bad_index = 1;
        return R;
    }
    const i64 n = G.n;
    // synthetic right-hand side and the sequential answer
    std::vector<double> b(n), want(n);
    uint64_t z = 0x9E3779B97F4A7C15ull;
    for (i64 i = 0; i < n; ++i) {
        z ^= z << 13; z ^= z >> 7; z ^= z << 17;
        b[i] = (double)(z >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
    }
    want = b;
    for (i64 t = 0; t < n; ++t) {
        const i64 r = G.descending ? n - 1 - t : t;
        double acc = want[r];
        for (i64 p = G.ptr[r]; p < G.ptr[r + 1]; ++p) acc -= G.val[p] * want[G.idx[p]];
        want[r] = acc / G.diag[r];
    }
    std::vector<double> x = b, scratch((size_t)std::max<i64>(P.scratch_rows, 1), 0.0);
    // level at which a row of X / scratch becomes final (-1: initial data, INT_MAX: not yet written)
    std::vector<i64> xdone(n, INT32_MAX), sdone((size_t)std::max<i64>(P.scratch_rows, 1), INT32_MAX);
    std::vector<int> writes(n, 0);
    std::vector<i64> xtask(n, -1), stask((size_t)std::max<i64>(P.scratch_rows, 1), -1);  // block task that wrote the row
    auto producer = [&](int32_t id) -> i64 & { return id >= 0 ? xtask[id] : stask[~id]; };
    const i64 nlev = (i64)P.level_ptr.size() - 1;
    auto valid = [&](int32_t id) { return id >= 0 ? id < n : (id != BlockProgram::kNone && (i64)(~id) < P.scratch_rows); };
    auto value = [&](int32_t id) -> double & { return id >= 0 ? x[id] : scratch[~id]; };
    auto done = [&](int32_t id) -> i64 & { return id >= 0 ? xdone[id] : sdone[~id]; };
    for (i64 l = 0; l < nlev; ++l) {
        // ---- clusters of this level ----
        for (i64 c = P.cluster_level_ptr[l]; c < P.cluster_level_ptr[l + 1]; ++c) {
            const ClusterHeader &h = P.clusters[c];
            const i64 ns = h.n_ext + h.n_own;
            const i64 stream_len = (i64)P.groups.size();
            if (ns > P.cluster_slots || h.ids_off < 0 || h.ids_off + ns > (i64)P.cluster_ids.size() || h.grp_off < 0 ||
                h.grp_off + h.ngroups > stream_len || h.ngroups < 1) {
                ++R.bad_index;
                continue;
            }
            std::vector<double> lds(ns);
            bool ok = true;
            for (i64 s = 0; s < ns; ++s) {
                const int32_t id = P.cluster_ids[h.ids_off + s];
                if (id < 0 || id >= n) { ++R.bad_index; ok = false; break; }
                if (s < h.n_ext && xdone[id] >= l) ++R.order_violations;   // external source not final yet
                if (s >= h.n_ext && xdone[id] != INT32_MAX) ++R.unwritten;  // own unknown already written
                lds[s] = x[id];
            }
            if (!ok) continue;
            double acc = 0.0;
            for (i64 g = 0; g < h.ngroups; ++g) {
                const GroupRec &rec = P.groups[h.grp_off + g];
                const i64 own = rec.own_off / 512;
                if (rec.own_off % 512 || own < h.n_ext || own >= ns) { ++R.bad_index; break; }
                if (rec.flags & GroupRec::kFirst) acc = lds[own];
                if (rec.flags & 7)
                    for (int e = 0; e < 4; ++e) {
                        const i64 s = rec.off[e] / 512;
                        if (rec.off[e] % 512 || s < 0 || s >= ns) { ++R.bad_index; continue; }
                        acc -= rec.coef[e] * lds[s];
                    }
                if (rec.flags & GroupRec::kLast) { acc *= rec.rdiag; lds[own] = acc; }
            }
            for (i64 s = h.n_ext; s < ns; ++s) {
                const int32_t id = P.cluster_ids[h.ids_off + s];
                x[id] = lds[s];
                xdone[id] = l;
                writes[id]++;
            }
        }
        // ---- wide items of this level (the items view; the tile view runs them as the blocks of classes 2, 4, 6) ----
        for (i64 k = P.item_level_ptr[l]; items_view && k < P.item_level_ptr[l + 1]; ++k) {
            const WideItem &I = P.items[k];
            const i64 nr = 16 * (i64)I.m_tiles, ns = 4 * (i64)I.k_steps, nwin = ((i64)I.k_steps + 15) / 16;
            if (I.m_tiles < 1 || I.m_tiles > 4 || I.k_steps < 0 || I.ids_off < 0 || I.ids_off + 2 * nr + ns > (i64)P.src_idx.size() || I.mask_off < 0 ||
                I.mask_off + 4 * nwin > (i64)P.item_masks.size() || I.coef_off < 0) {
                ++R.bad_index;
                continue;
            }
            const int32_t *ids = P.src_idx.data() + I.ids_off;
            const uint32_t *mw = P.item_masks.data() + I.mask_off;
            std::vector<double> acc((size_t)nr, 0.0);
            for (i64 r = 0; r < nr; ++r) {
                const int32_t id = ids[nr + r];
                if (id == BlockProgram::kNone) continue;
                if (!valid(id)) { ++R.bad_index; continue; }
                if (id < 0 && done(id) >= l) ++R.order_violations;
                acc[(size_t)r] = value(id);
            }
            // replay of dev::solve_items: the blocks of a K-step follow those of the K-step before; the window words give the
            // masks and the number of blocks before each window
            i64 blk = 0;
            bool ok = true;
            for (i64 k = 0; k < I.k_steps && ok; ++k) {
                if (k % 16 == 0 && (i64)mw[4 * (k / 16) + 2] != blk) { ++R.bad_index; ok = false; break; }
                const unsigned mask = (mw[4 * (k / 16) + ((k % 16) >= 8 ? 1 : 0)] >> (4 * (k % 8))) & 15u;
                if (mask >> I.m_tiles) { ++R.bad_index; ok = false; break; }
                for (int q = 0; q < I.m_tiles; ++q) {
                    if (!((mask >> q) & 1u)) continue;
                    if (I.coef_off + (blk + 1) * 64 > (i64)P.coef.size()) { ++R.bad_index; ok = false; break; }
                    const double *cb = P.coef.data() + I.coef_off + blk * 64;
                    ++blk;
                    for (int gsrc = 0; gsrc < 4; ++gsrc) {
                        const int32_t id = ids[2 * nr + 4 * k + gsrc];
                        if (!valid(id)) { ++R.bad_index; continue; }
                        bool zero = true;
                        for (int r = 0; r < 16; ++r) zero = zero && cb[gsrc * 16 + r] == 0.0;
                        if (!zero && done(id) >= l) ++R.order_violations;  // (an item never reads what its own level writes)
                        for (int r = 0; r < 16; ++r) acc[(size_t)(16 * q + r)] -= cb[gsrc * 16 + r] * value(id);
                    }
                }
            }
            if (!ok) continue;
            for (i64 r = 0; r < nr; ++r) {
                const int32_t id = ids[r];
                if (id == BlockProgram::kNone) continue;
                if (!valid(id)) { ++R.bad_index; continue; }
                value(id) = acc[(size_t)r];
                done(id) = l;
                if (id >= 0) writes[id]++;
                producer(id) = -3;
            }
        }
        // ---- block tasks of this level ----
        for (i64 t = P.level_ptr[l]; t < P.level_ptr[l + 1]; ++t) {
            if (items_view) {  // classes 2, 4, 6: row tiles of the items above
                int c = 0;
                while (c + 1 < NC && t >= P.level_class_ptr[NC * l + c + 1]) ++c;
                if (c == 2 || c == 4 || c == 6) continue;
            }
            for (i64 bk = P.task_ptr[t]; bk < P.task_ptr[t + 1]; ++bk) {
                const BlockHeader &h = P.hdr[bk];
                const int rc = h.rc;
                if ((rc != 1 && rc != 2 && rc != 4 && rc != 16) || h.src_off < 0 ||
                    h.src_off + 2 * rc + h.nsrc > (i64)P.src_idx.size() || h.coef_off < 0 ||
                    h.coef_off + (i64)h.nsrc * rc + (h.has_tri ? (rc == 16 ? 256 : rc * rc) : 0) > (i64)P.coef.size() ||
                    (rc == 16 && h.nsrc % 16)) {
                    ++R.bad_index;
                    continue;
                }
                const int32_t *ids = P.src_idx.data() + h.src_off;
                const double *cf = P.coef.data() + h.coef_off;
                // the prologue records the kernels actually read must say what the pools say
                {
                    constexpr int LW = BlockProgram::kLineWords;
                    const bool have = (i64)P.block_record.size() == (i64)P.hdr.size() + 1 && P.block_record[bk] >= 0 &&
                                      P.block_record[bk + 1] > P.block_record[bk] &&
                                      (size_t)P.block_record[bk + 1] * LW <= P.prologue.size();
                    const i64 span = have ? P.block_record[bk + 1] - P.block_record[bk] : 0;
                    const i64 W = span == 2 ? 1 : span == 8 ? 4 : span == 24 ? 8 : 0;
                    if (!have || !W) {
                        ++R.bad_index;
                    } else {
                        const i64 nst = rc == 16 ? h.nsrc / 16 : 0, lines = BlockProgram::record_lines((int)W);
                        for (i64 w = 0; w < W; ++w) {
                            const int32_t *rec = P.prologue.data() + (size_t)(P.block_record[bk] + w * lines) * LW;
                            auto word = [&](i64 line, i64 lane) { return rec[lane * lines + line]; };
                            int32_t head[8];
                            std::memcpy(head, &h, sizeof(BlockHeader));
                            bool same = true;
                            for (int k = 0; k < 8; ++k) same = same && word(0, k) == head[k];
                            for (int r = 0; r < rc && r < 16; ++r) same = same && word(0, 8 + r) == ids[r] && word(0, 24 + r) == ids[rc + r];
                            if (rc == 16) {
                                const i64 first = (nst * w) / W, last = (nst * (w + 1)) / W;
                                for (i64 e = 0; same && e < std::min<i64>((last - first) * 16, (lines - 1) * LW); ++e)
                                    same = word(1 + e / LW, e % LW) == ids[32 + first * 16 + e];
                            } else {
                                for (int e = 0; e < 16 && e < h.nsrc; ++e) same = same && word(0, 40 + e) == ids[2 * rc + e];
                            }
                            if (!same) ++R.bad_index;
                        }
                    }
                }
                double acc[16];
                for (int r = 0; r < rc; ++r) {
                    const int32_t id = ids[rc + r];
                    acc[r] = 0.0;
                    if (id == BlockProgram::kNone) continue;
                    if (!valid(id)) { ++R.bad_index; continue; }
                    // init rows: initial data of X (never written before) or a scratch row final at a lower level
                    if (id < 0 && done(id) >= l) ++R.order_violations;
                    acc[r] = value(id);
                }
                for (i64 e = 0; e < h.nsrc; ++e) {
                    const int32_t id = ids[2 * rc + e];
                    if (!valid(id)) { ++R.bad_index; continue; }
                    bool zero = true;
                    for (int r = 0; r < rc; ++r) zero = zero && cf[e * rc + r] == 0.0;
                    // final at a lower level, or written earlier by this very task (marked with level l during the task)
                    if (!zero && done(id) > l) ++R.order_violations;
                    if (!zero && done(id) == l && producer(id) != t) ++R.order_violations;
                    for (int r = 0; r < rc; ++r) acc[r] -= cf[e * rc + r] * value(id);
                }
                if (h.has_tri) {
                    const double *tri = cf + (i64)h.nsrc * rc;
                    if (rc < 16) {
                        for (int r = 0; r < rc; ++r) {
                            for (int u = 0; u < r; ++u) acc[r] -= tri[r * rc + u] * acc[u];
                            acc[r] /= tri[r * rc + r];
                        }
                    } else {
                        double out[16];
                        for (int r = 0; r < 16; ++r) {
                            double s = 0.0;
                            for (int u = 0; u < 16; ++u) s += tri[(u / 4) * 64 + (u % 4) * 16 + r] * acc[u];
                            out[r] = s;
                        }
                        std::copy(out, out + 16, acc);
                    }
                }
                for (int r = 0; r < rc; ++r) {
                    const int32_t id = ids[r];
                    if (id == BlockProgram::kNone) continue;
                    if (!valid(id)) { ++R.bad_index; continue; }
                    value(id) = acc[r];
                    done(id) = l;
                    if (id >= 0) writes[id]++;
                    producer(id) = t;
                }
            }
        }
    }
    double top = 1e-300, err = 0.0;
    for (i64 i = 0; i < n; ++i) {
        if (writes[i] != 1) ++R.unwritten;
        top = std::max(top, std::fabs(want[i]));
        const double e = std::fabs(x[i] - want[i]);
        if (!(e <= err)) err = e;  // NaN propagates
    }
    R.max_rel_error = err / top;
    return R;
}

}  // namespace rsp
