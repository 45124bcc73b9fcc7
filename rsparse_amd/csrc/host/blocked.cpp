// Builder of the blocked multi-right-hand-side programme (see BlockProgram in sparse.hpp).
// Integer/host work only; no counterpart in the reference, whose solves are sequential loops
// (/root/reference/src/lib.rs:464-512, 1230-1278).  The arithmetic each unknown receives is the same
// set of multiply-subtracts as in those loops, regrouped; for chains replaced by an explicit inverse
// the diagonal block is applied as M^-1 instead of by substitution (results agree to rounding).
#include <algorithm>
#include <cstring>

#include "sparse.hpp"

namespace rsp {
namespace {

constexpr int32_t kNone = BlockProgram::kNone;

// 1, 2, 4: vector path (registers per lane); more than 4 unknowns: a 16-row matrix-core block (MFMA 16x16x4 f64)
inline int class_of(i64 rows) { return rows <= 1 ? 1 : rows <= 2 ? 2 : rows <= 4 ? 4 : 16; }

struct Triple {
    int32_t pos;   // processing position of the source (or scratch ordinal for B blocks)
    int32_t slot;  // row slot inside the block
    double val;
};

class Builder {
   public:
    Builder(const GatherForm &G, i64 cap) : G_(G), n_(G.n), cap_(cap)
    {
        avail_.assign(n_, 0);
        task_of_.assign(n_, -1);
        stamp_.assign(n_, -1);
    }

    BlockProgram run()
    {
        find_runs();
        for (size_t k = 0; k + 1 < run_begin_.size(); ++k) {
            const i64 t0 = run_begin_[k], t1 = run_begin_[k + 1];
            if (t1 - t0 > kBigChain) emit_big_chain(t0, t1);
            else emit_normal_run(t0, t1);
        }
        finish();
        return std::move(P_);
    }

   private:
    i64 row_at(i64 t) const { return G_.descending ? n_ - 1 - t : t; }
    i64 pos_of(i64 r) const { return G_.descending ? n_ - 1 - r : r; }

    // maximal runs of consecutive unknowns in which each one depends on ALL earlier ones of the run
    void find_runs()
    {
        std::vector<i64> mark(n_, -1);
        i64 t0 = 0, run = 0;
        run_begin_.push_back(0);
        for (i64 t = 0; t < n_; ++t) {
            const i64 r = row_at(t);
            i64 inside = 0;
            for (i64 p = G_.ptr[r]; p < G_.ptr[r + 1]; ++p) inside += mark[G_.idx[p]] == run;
            if (t > t0 && inside != t - t0) {
                run_begin_.push_back(t);
                t0 = t;
                ++run;
            }
            mark[r] = run;
        }
        if (n_ > 0) run_begin_.push_back(n_);
    }

    // ---- low-level emission ------------------------------------------------------------------------
    // sources: triples sorted by (pos, slot); ids of sources given by id_of(pos).  Returns block index.
    template <class IdOf>
    i64 push_block(int rows, const int32_t *dest, const int32_t *init, std::vector<Triple> &src, IdOf id_of, const double *tri,
                   i64 level, bool own_task)
    {
        const int rc = class_of(rows);
        std::sort(src.begin(), src.end(), [](const Triple &a, const Triple &b) { return a.pos != b.pos ? a.pos < b.pos : a.slot < b.slot; });
        const i64 b = (i64)P_.hdr.size();
        BlockHeader h{rc, 0, tri ? 1 : 0, 0, (i64)P_.src_idx.size(), (i64)P_.coef.size()};
        for (int r = 0; r < rc; ++r) P_.src_idx.push_back(r < rows ? dest[r] : kNone);
        for (int r = 0; r < rc; ++r) P_.src_idx.push_back(r < rows ? init[r] : kNone);
        i64 ns = 0;
        for (size_t k = 0; k < src.size();) {
            const int32_t pos = src[k].pos;
            P_.src_idx.push_back(id_of(pos));
            const size_t base = P_.coef.size();
            P_.coef.resize(base + rc, 0.0);
            for (; k < src.size() && src[k].pos == pos; ++k) P_.coef[base + src[k].slot] = src[k].val;
            ++ns;
        }
        if (rc == 16)  // the matrix-core path consumes sources sixteen at a time: pad with zero coefficients
            for (; ns % 16 != 0; ++ns) {
                P_.src_idx.push_back(P_.src_idx.back());
                P_.coef.resize(P_.coef.size() + rc, 0.0);
            }
        h.nsrc = (int32_t)ns;
        P_.hdr.push_back(h);
        if (tri && rc < 16) {
            const size_t base = P_.coef.size();
            P_.coef.resize(base + (size_t)rc * rc, 0.0);
            for (int r = 0; r < rc; ++r)
                for (int t = 0; t <= r; ++t) P_.coef[base + (size_t)r * rc + t] = (r < rows) ? tri[r * 16 + t] : (t == r ? 1.0 : 0.0);
        } else if (tri) {
            // inverse of the block's 16 x 16 lower triangle (unit rows for padding), stored in the A-operand order of
            // v_mfma_f64_16x16x4_f64: K-step k, lane l holds Tinv[l & 15][4k + (l >> 4)]
            double T[16][16], Ti[16][16];
            for (int r = 0; r < 16; ++r)
                for (int t = 0; t < 16; ++t) {
                    T[r][t] = (r < rows && t <= r) ? tri[r * 16 + t] : (t == r ? 1.0 : 0.0);
                    Ti[r][t] = 0.0;
                }
            for (int c = 0; c < 16; ++c) {
                Ti[c][c] = 1.0 / T[c][c];
                for (int r = c + 1; r < 16; ++r) {
                    double sum = 0.0;
                    for (int t = c; t < r; ++t) sum += T[r][t] * Ti[t][c];
                    Ti[r][c] = -sum / T[r][r];
                }
            }
            const size_t base = P_.coef.size();
            P_.coef.resize(base + 256, 0.0);
            for (int k = 0; k < 4; ++k)
                for (int l = 0; l < 64; ++l) P_.coef[base + (size_t)k * 64 + l] = Ti[l & 15][4 * k + (l >> 4)];
        }
        P_.padded_entries += ns * rc;
        blk_entries_.push_back(ns * rc + (tri ? rc * rc / 2 : 0));
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
        std::memset(tri, 0, sizeof(tri));
        src_.clear();
        i64 need = 0;
        bool chained = false;
        for (i64 t = a; t < b; ++t) {
            const i64 r = row_at(t);
            const int slot = (int)(t - a);
            dest[slot] = (int32_t)r;
            tri[slot * 16 + slot] = G_.diag[r];
            for (i64 p = G_.ptr[r]; p < G_.ptr[r + 1]; ++p) {
                const i64 s = G_.idx[p];
                const i64 ps = pos_of(s);
                if (ps >= a) {
                    tri[slot * 16 + (ps - a)] = G_.val[p];
                } else {
                    src_.push_back(Triple{(int32_t)ps, slot, G_.val[p]});
                    if (cur_task_ >= 0 && task_of_[s] == cur_task_) chained = true;
                    else need = std::max<i64>(need, avail_[s] + 1);
                }
            }
        }
        // distinct sources (for the task's work bound)
        i64 uniq = 0;
        {
            ++stamp_id_;
            for (const Triple &x : src_) {
                const i64 s = row_at(x.pos);
                if (stamp_[s] != stamp_id_) { stamp_[s] = stamp_id_; ++uniq; }
            }
        }
        const i64 work = uniq * class_of(rows) + rows * rows / 2 + 1;
        const bool fits = cur_task_ >= 0 && cur_work_ + work <= cap_;
        i64 level;
        if (fits && need <= task_level_[cur_task_] && (chained || need == task_level_[cur_task_])) {
            level = task_level_[cur_task_];
            task_nblk_[cur_task_]++;
            cur_work_ += work;
        } else {
            if (cur_task_ >= 0 && chained) need = std::max<i64>(need, task_level_[cur_task_] + 1);
            level = need;
            cur_task_ = (i64)task_level_.size();
            task_first_blk_.push_back((i64)P_.hdr.size());
            task_level_.push_back((int32_t)level);
            task_nblk_.push_back(1);
            cur_work_ = work;
        }
        const i64 keep_task = cur_task_;
        push_block(rows, dest, dest, src_, [&](int32_t pos) { return (int32_t)row_at(pos); }, tri, level, false);
        cur_task_ = keep_task;
        for (i64 t = a; t < b; ++t) {
            avail_[row_at(t)] = (int32_t)level;
            task_of_[row_at(t)] = (int32_t)cur_task_;
        }
    }

    // ---- gather with optional split into partial sums; every block is its own task -------------------------
    // src triples: pos = ordinal understood by id_of.  Returns the level at which dest is complete.
    template <class IdOf>
    i64 emit_gather(int rows, const int32_t *dest, const int32_t *init, std::vector<Triple> &src, IdOf id_of, i64 level)
    {
        std::sort(src.begin(), src.end(), [](const Triple &a, const Triple &b) { return a.pos != b.pos ? a.pos < b.pos : a.slot < b.slot; });
        // count distinct sources
        i64 ns = 0;
        for (size_t k = 0; k < src.size(); ++k) ns += (k == 0 || src[k].pos != src[k - 1].pos);
        if (ns <= kChunkSources) {
            push_block(rows, dest, init, src, id_of, nullptr, level, true);
            return level;
        }
        const i64 parts = (ns + kChunkSources - 1) / kChunkSources;
        const i64 per = (ns + parts - 1) / parts;
        std::vector<int32_t> pdest((size_t)parts * rows);
        std::vector<Triple> chunk;
        size_t k = 0;
        for (i64 p = 0; p < parts; ++p) {
            chunk.clear();
            i64 taken = 0;
            while (k < src.size()) {
                const bool new_src = (chunk.empty() || src[k].pos != chunk.back().pos);
                if (new_src && taken == per) break;
                taken += new_src;
                chunk.push_back(src[k++]);
            }
            int32_t none[16];
            for (int r = 0; r < rows; ++r) {
                pdest[(size_t)p * rows + r] = ~(int32_t)(P_.scratch_rows++);
                none[r] = kNone;
            }
            push_block(rows, &pdest[(size_t)p * rows], p == 0 ? init : none, chunk, id_of, nullptr, level, true);
        }
        // combine: dest = partial_0 + sum_{p>=1} partial_p   (acc -= (-1) * partial)
        std::vector<Triple> comb;
        for (i64 p = 1; p < parts; ++p)
            for (int r = 0; r < rows; ++r) comb.push_back(Triple{(int32_t)((p - 1) * rows + r), r, -1.0});
        const int32_t *p1 = pdest.data() + rows;
        push_block(rows, dest, pdest.data(), comb, [&](int32_t o) { return p1[o]; }, nullptr, level + 1, true);
        return level + 1;
    }

    // ---- long dense chains: Y = X_J - L[J,ext] X_ext ; X_J = M^-1 Y ------------------------------------------
    void emit_big_chain(i64 t0, i64 t1)
    {
        close_task();
        const i64 w = t1 - t0;
        P_.big_chains++;
        P_.inverted_rows += w;
        // dense diagonal block in position coordinates, then its inverse (forward substitution per column)
        std::vector<double> M((size_t)w * w, 0.0);
        for (i64 t = t0; t < t1; ++t) {
            const i64 r = row_at(t);
            M[(size_t)(t - t0) * w + (t - t0)] = G_.diag[r];
            for (i64 p = G_.ptr[r]; p < G_.ptr[r + 1]; ++p) {
                const i64 ps = pos_of(G_.idx[p]);
                if (ps >= t0) M[(size_t)(t - t0) * w + (ps - t0)] = G_.val[p];
            }
        }
        std::vector<double> Minv((size_t)w * w, 0.0);
        std::vector<double> z(w);
        for (i64 c = 0; c < w; ++c) {
            z[c] = 1.0 / M[(size_t)c * w + c];
            for (i64 r = c + 1; r < w; ++r) {
                const double *Mr = &M[(size_t)r * w];
                double s = 0.0;
                for (i64 t = c; t < r; ++t) s += Mr[t] * z[t];
                z[r] = -s / Mr[r];
            }
            for (i64 r = c; r < w; ++r) Minv[(size_t)r * w + c] = z[r];
        }
        const i64 y0 = P_.scratch_rows;
        P_.scratch_rows += w;
        // phase A per group of 16 unknowns
        i64 ready = 0;
        for (i64 a = t0; a < t1; a += 16) {
            const i64 b = std::min(a + 16, t1);
            const int rows = (int)(b - a);
            int32_t dest[16], init[16];
            src_.clear();
            i64 need = 0;
            for (i64 t = a; t < b; ++t) {
                const i64 r = row_at(t);
                const int slot = (int)(t - a);
                dest[slot] = ~(int32_t)(y0 + (t - t0));
                init[slot] = (int32_t)r;
                for (i64 p = G_.ptr[r]; p < G_.ptr[r + 1]; ++p) {
                    const i64 s = G_.idx[p];
                    const i64 ps = pos_of(s);
                    if (ps >= t0) continue;
                    src_.push_back(Triple{(int32_t)ps, slot, G_.val[p]});
                    need = std::max<i64>(need, avail_[s] + 1);
                }
            }
            const i64 done = emit_gather(rows, dest, init, src_, [&](int32_t pos) { return (int32_t)row_at(pos); }, need);
            ready = std::max(ready, done);
        }
        // phase B: X_J = Minv * Y, all groups independent
        i64 final_level = ready + 1;
        for (i64 a = t0; a < t1; a += 16) {
            const i64 b = std::min(a + 16, t1);
            const int rows = (int)(b - a);
            int32_t dest[16], init[16];
            src_.clear();
            for (i64 t = a; t < b; ++t) {
                const int slot = (int)(t - a);
                dest[slot] = (int32_t)row_at(t);
                init[slot] = kNone;
                const double *Mi = &Minv[(size_t)(t - t0) * w];
                for (i64 u = 0; u <= t - t0; ++u) src_.push_back(Triple{(int32_t)u, slot, -Mi[u]});
            }
            const i64 done = emit_gather(rows, dest, init, src_, [&](int32_t o) { return ~(int32_t)(y0 + o); }, ready + 1);
            final_level = std::max(final_level, done);
        }
        for (i64 t = t0; t < t1; ++t) {
            avail_[row_at(t)] = (int32_t)final_level;
            task_of_[row_at(t)] = -2;  // never "chained" to a running task
        }
        close_task();
    }

    // ---- order tasks by level and blocks by task ---------------------------------------------------------
    void finish()
    {
        const i64 ntasks = (i64)task_level_.size();
        i64 nlev = 0;
        for (int32_t l : task_level_) nlev = std::max<i64>(nlev, l + 1);
        P_.level_ptr.assign(nlev + 1, 0);
        for (int32_t l : task_level_) P_.level_ptr[l + 1]++;
        for (i64 l = 0; l < nlev; ++l) P_.level_ptr[l + 1] += P_.level_ptr[l];
        std::vector<i64> fill(P_.level_ptr.begin(), P_.level_ptr.end() - 1), order(ntasks);
        for (i64 t = 0; t < ntasks; ++t) order[fill[task_level_[t]]++] = t;
        // permute the per-block arrays into task order
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
    }

    const GatherForm &G_;
    const i64 n_, cap_;
    BlockProgram P_;
    std::vector<i64> run_begin_;
    std::vector<int32_t> avail_, task_of_;
    std::vector<i64> stamp_;
    i64 stamp_id_ = 0;
    std::vector<Triple> src_;
    std::vector<i64> blk_entries_;
    // tasks in creation order
    std::vector<i64> task_first_blk_, task_nblk_;
    std::vector<int32_t> task_level_;
    i64 cur_task_ = -1, cur_work_ = 0;
};

}  // namespace

BlockProgram build_block_program(const GatherForm &G, i64 max_task_entries)
{
    Builder b(G, max_task_entries);
    return b.run();
}

}  // namespace rsp
