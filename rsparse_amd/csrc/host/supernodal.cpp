// Builder and host validator of the SUPERNODAL multi-right-hand-side programme (SnProgram in sparse.hpp).
// Integer/host work plus the small dense algebra of the nodes (inverse diagonal blocks); no counterpart in the
// reference, whose solves are sequential loops (/root/reference/src/lib.rs:464-512, 1230-1278).  Every unknown
// receives the same multiply-subtracts as in those loops, regrouped by supernode; diagonal blocks are applied as
// explicit inverses (results agree to rounding).
//
// Everything is valid for ANY triangular pattern: the forest that picks tasks and the nesting test that forms nodes
// are heuristics, dependencies and levels are always taken from the real entries, and check_supernodal_program()
// replays a programme on the host with every index and every read-after-write checked.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "parallel.hpp"
#include "sparse.hpp"

namespace rsp {
namespace {

constexpr int32_t kNone = SnProgram::kNone;
constexpr int32_t kZeroRow = ~(int32_t)0;  // scratch row 0

class SnBuilder {
   public:
    SnBuilder(const GatherForm &R, const GatherForm *C, bool scatter, const SnOptions &opt)
        : R_(R), S_(scatter ? *C : R), n_(R.n), scatter_(scatter), flip_(R.descending == scatter), opt_(opt)
    {
        P_.n = n_;
        P_.scatter = scatter;
    }

    SnProgram run()
    {
        select_tasks();
        define_units();
        if (scatter_) emit_scatter();
        else emit_gather();
        finish();
        return std::move(P_);
    }

   private:
    // canonical positions: the structure S(p) of every column lies at positions > p (L: p = index, U: p = n - 1 - index)
    i64 row_of(i64 p) const { return flip_ ? n_ - 1 - p : p; }
    i64 pos_of(i64 r) const { return flip_ ? n_ - 1 - r : r; }

    struct Unit {
        bool task;
        i64 lo, hi;       // positions [lo, hi]
        i64 e_off, e_cnt; // into epool_: positions of the rows below (sorted ascending)
        i64 sid = 0;      // scatter kinds: first partial scratch row (one per E row)
        i64 ysid = 0;     // nodes: first Y scratch row (0: none)
    };

    // ---- tasks: maximal subtrees of the column forest that are position ranges and fit the LDS slot budget ----------
    void select_tasks()
    {
        task_of_.assign(n_, -1);
        stamp_.assign(n_, -1);
        if (opt_.task_slots < 2) return;
        std::vector<i64> parent(n_, -1), size(n_, 1), first(n_), cnt(n_, 0);
        for (i64 p = 0; p < n_; ++p) {
            const i64 r = row_of(p);
            first[p] = p;
            cnt[p] = S_.ptr[r + 1] - S_.ptr[r];
            for (i64 e = S_.ptr[r]; e < S_.ptr[r + 1]; ++e) {
                const i64 q = pos_of(S_.idx[e]);
                if (parent[p] < 0 || q < parent[p]) parent[p] = q;
            }
        }
        for (i64 p = 0; p < n_; ++p)
            if (parent[p] >= 0) {
                size[parent[p]] += size[p];
                first[parent[p]] = std::min(first[parent[p]], first[p]);
            }
        // scatter kinds: a task must receive nothing from outside, i.e. every source of its rows lies inside it
        std::vector<i64> minsrc;
        if (scatter_) {
            minsrc.assign(n_, n_);
            for (i64 p = 0; p < n_; ++p) {
                const i64 r = row_of(p);
                for (i64 e = R_.ptr[r]; e < R_.ptr[r + 1]; ++e) minsrc[p] = std::min(minsrc[p], pos_of(R_.idx[e]));
            }
        }
        std::vector<i64> unit_root(n_, -1);
        for (i64 p = n_ - 1; p >= 0; --p) {
            const i64 q = parent[p];
            if (q >= 0 && unit_root[q] >= 0) {
                unit_root[p] = unit_root[q];
                continue;
            }
            if (first[p] == p - size[p] + 1 && size[p] + cnt[p] <= opt_.task_slots) unit_root[p] = p;
        }
        i64 a = -1, b = -1;
        auto range_ok = [&](i64 lo, i64 hi) {
            if (slots_needed(lo, hi) > opt_.task_slots) return false;
            if (scatter_)
                for (i64 p = lo; p <= hi; ++p)
                    if (minsrc[p] < lo) return false;
            return true;
        };
        auto close = [&]() {
            if (a < 0) return;
            const int32_t t = (int32_t)task_lo_.size();
            task_lo_.push_back(a);
            task_hi_.push_back(b);
            for (i64 p = a; p <= b; ++p) task_of_[p] = t;
            a = b = -1;
        };
        for (i64 p = 0; p < n_;) {
            const i64 root = unit_root[p];
            if (root < 0) {
                close();
                ++p;
                continue;
            }
            const i64 lo = p, hi = root;
            if (a >= 0 && lo == b + 1 && range_ok(a, hi)) {
                b = hi;
            } else {
                close();
                if (range_ok(lo, hi)) {
                    a = lo;
                    b = hi;
                }
            }
            p = hi + 1;
        }
        close();
    }

    i64 slots_needed(i64 lo, i64 hi)
    {
        if (hi - lo + 1 > opt_.task_slots) return hi - lo + 1;
        ++stamp_id_;
        i64 ext = 0;
        for (i64 p = lo; p <= hi; ++p) {
            const i64 r = row_of(p);
            for (i64 e = S_.ptr[r]; e < S_.ptr[r + 1]; ++e) {
                const i64 q = pos_of(S_.idx[e]);
                if (q <= hi) continue;
                if (stamp_[q] != stamp_id_) {
                    stamp_[q] = stamp_id_;
                    ++ext;
                }
            }
        }
        return hi - lo + 1 + ext;
    }

    // sorted positions of the structure of [lo, hi] that lie beyond hi
    void structure_below(i64 lo, i64 hi, std::vector<i64> &out)
    {
        out.clear();
        ++stamp_id_;
        for (i64 p = lo; p <= hi; ++p) {
            const i64 r = row_of(p);
            for (i64 e = S_.ptr[r]; e < S_.ptr[r + 1]; ++e) {
                const i64 q = pos_of(S_.idx[e]);
                if (q <= hi) continue;
                if (stamp_[q] != stamp_id_) {
                    stamp_[q] = stamp_id_;
                    out.push_back(q);
                }
            }
        }
        std::sort(out.begin(), out.end());
    }

    // ---- units in position order: tasks, and nodes formed from runs of consecutive columns with nested structure -----
    void define_units()
    {
        std::vector<i64> E, Sq, merged;
        for (i64 p = 0; p < n_;) {
            if (task_of_[p] >= 0) {
                const int32_t t = task_of_[p];
                structure_below(task_lo_[t], task_hi_[t], E);
                units_.push_back(Unit{true, task_lo_[t], task_hi_[t], (i64)epool_.size(), (i64)E.size()});
                epool_.insert(epool_.end(), E.begin(), E.end());
                P_.task_rows += task_hi_[t] - task_lo_[t] + 1;
                p = task_hi_[t] + 1;
                continue;
            }
            const i64 pa = p;
            structure_below(pa, pa, E);
            i64 w = 1, real = 1 + (i64)E.size(), zeros = 0;
            while (pa + w < n_ && task_of_[pa + w] < 0 && w < opt_.max_node && !E.empty() && E.front() == pa + w) {
                const i64 q = pa + w;
                structure_below(q, q, Sq);
                // E' = (E \\ {q}) u Sq;  zeros: the old columns take the rows only the new column has, and vice versa
                merged.clear();
                std::set_union(E.begin() + 1, E.end(), Sq.begin(), Sq.end(), std::back_inserter(merged));
                const i64 extra = (i64)merged.size() - ((i64)E.size() - 1), missing = (i64)merged.size() - (i64)Sq.size();
                const i64 added = w * extra + missing;
                if ((double)(zeros + added) > opt_.relax * (double)(real + 1 + (i64)Sq.size())) break;
                zeros += added;
                real += 1 + (i64)Sq.size();
                E.swap(merged);
                ++w;
            }
            units_.push_back(Unit{false, pa, pa + w - 1, (i64)epool_.size(), (i64)E.size()});
            epool_.insert(epool_.end(), E.begin(), E.end());
            P_.nnodes++;
            P_.top_rows += w;
            if (w > opt_.small_node) P_.big_nodes++;
            p = pa + w;
        }
    }

    // ---- dense algebra of a node ----------------------------------------------------------------------------------------
    // D (w x w, lower in position space), its inverse, and T_E (m x w); all row-major
    void node_matrices(const Unit &u, std::vector<double> &Dinv, std::vector<double> &TE)
    {
        const i64 w = u.hi - u.lo + 1, m = u.e_cnt;
        if (eslot_.empty()) eslot_.assign(n_, -1);
        for (i64 k = 0; k < m; ++k) eslot_[epool_[u.e_off + k]] = (int32_t)k;
        std::vector<double> D((size_t)w * w, 0.0);
        TE.assign((size_t)m * w, 0.0);
        for (i64 j = 0; j < w; ++j) {
            const i64 r = row_of(u.lo + j);
            D[(size_t)j * w + j] = S_.diag[r];
            for (i64 e = S_.ptr[r]; e < S_.ptr[r + 1]; ++e) {
                const i64 q = pos_of(S_.idx[e]);
                if (q <= u.hi) D[(size_t)(q - u.lo) * w + j] += S_.val[e];  // duplicates add up, as the loops would apply both
                else TE[(size_t)eslot_[q] * w + j] += S_.val[e];
            }
        }
        Dinv.assign((size_t)w * w, 0.0);
        auto invert_columns = [&](i64 first, i64 stride) {
            std::vector<double> z(w);
            for (i64 c = first; c < w; c += stride) {
                z[c] = 1.0 / D[(size_t)c * w + c];
                for (i64 r = c + 1; r < w; ++r) {
                    const double *Dr = &D[(size_t)r * w];
                    double s = 0.0;
                    for (i64 t = c; t < r; ++t) s += Dr[t] * z[t];
                    z[r] = -s / Dr[r];
                }
                for (i64 r = c; r < w; ++r) Dinv[(size_t)r * w + c] = z[r];
            }
        };
        const i64 nthreads = w >= 384 ? std::min<i64>(host_threads(), w / 64) : 1;
        if (nthreads <= 1) {
            invert_columns(0, 1);
        } else {  // independent columns, dealt round-robin: the result does not depend on the number of threads
            std::vector<std::thread> pool;
            for (i64 k = 0; k < nthreads; ++k) pool.emplace_back(invert_columns, k, nthreads);
            for (std::thread &th : pool) th.join();
        }
    }

    // W = T_E D^-1 (m x w)
    static void fold_inverse(const std::vector<double> &TE, const std::vector<double> &Dinv, i64 m, i64 w, std::vector<double> &W)
    {
        W.assign((size_t)m * w, 0.0);
        // rows of W are independent of each other: shared out over the host threads when the product is worth it
        parallel_ranges(m, std::max<i64>(1, (i64)4000000 / std::max<i64>(1, w * w)), [&](i64 k0, i64 k1) {
            for (i64 k = k0; k < k1; ++k) {
                const double *t = &TE[(size_t)k * w];
                double *o = &W[(size_t)k * w];
                for (i64 j = 0; j < w; ++j) {
                    const double tj = t[j];
                    if (tj == 0.0) continue;
                    const double *di = &Dinv[(size_t)j * w];
                    for (i64 c = 0; c <= j; ++c) o[c] += tj * di[c];
                }
            }
        });
    }

    // ---- items --------------------------------------------------------------------------------------------------------------
    // rows [r0, r1) of  OUT = INIT + A * IN  restricted to input rows [k0, k1)  (k0 a multiple of 4)
    template <class AF>
    void push_item(i64 r0, i64 r1, const int32_t *out_ids, const int32_t *init_ids, i64 k0, i64 k1, const int32_t *in_ids, AF &&A,
                   i64 level)
    {
        const i64 rows = r1 - r0, mt = (rows + 15) / 16, ks = (k1 - k0 + 3) / 4;
        SnItem it{(int32_t)mt, (int32_t)ks, 1, init_ids ? 1 : 0, (i64)P_.coef.size(), (i64)P_.ids.size(), 0, 0};
        it.nsplit = ks >= 48 ? 4 : ks >= 16 ? 2 : 1;
        for (i64 i = 0; i < 16 * mt; ++i) P_.ids.push_back(i < rows ? out_ids[r0 + i] : kNone);
        it.in_off = (i64)P_.ids.size();
        for (i64 k = 0; k < 4 * ks; ++k) P_.ids.push_back(k0 + k < k1 ? in_ids[k0 + k] : kZeroRow);
        it.init_off = (i64)P_.ids.size();
        if (init_ids)
            for (i64 i = 0; i < 16 * mt; ++i) P_.ids.push_back(i < rows ? init_ids[r0 + i] : kNone);
        P_.coef.resize(P_.coef.size() + (size_t)(ks * mt * 64), 0.0);
        double *cf = P_.coef.data() + it.a_off;
        for (i64 i = 0; i < rows; ++i)
            for (i64 k = k0; k < k1; ++k) {
                const double v = A(r0 + i, k);
                if (v != 0.0) cf[(((k - k0) >> 2) * mt + (i >> 4)) * 64 + ((k - k0) & 3) * 16 + (i & 15)] = v;
            }
        items_.push_back(it);
        item_level_.push_back((int32_t)level);
        P_.padded_flops_per_col += 2 * 16 * mt * 4 * ks;
        last_level_ = std::max(last_level_, level);
    }

    // OUT[i] = INIT[i] + sum_k A(i, k) IN[k], i < M.  krange(r0, r1) -> [k0, k1): the input rows the group r0..r1 uses
    // (triangular operands).  Returns the level at which OUT is complete.
    template <class AF, class KR>
    i64 emit_product(i64 M, const int32_t *out_ids, const int32_t *init_ids, const int32_t *in_ids, AF &&A, KR &&krange, i64 level,
                     i64 group = 64, i64 max_k = 0)
    {
        i64 done = level;
        if (max_k <= 0) max_k = opt_.max_item_k;
        for (i64 r0 = 0; r0 < M; r0 += group) {
            const i64 r1 = std::min(M, r0 + group), rows = r1 - r0;
            i64 k0, k1;
            krange(r0, r1, k0, k1);
            k0 &= ~(i64)3;
            const i64 K = k1 - k0;
            if (K <= max_k) {
                push_item(r0, r1, out_ids, init_ids, k0, k1, in_ids, A, level);
                continue;
            }
            // partial products into scratch rows, then one combine item
            const i64 parts = (K + max_k - 1) / max_k;
            const i64 per = ((K + parts - 1) / parts + 3) & ~(i64)3;
            std::vector<int32_t> tmp((size_t)(parts * rows));
            for (i64 q = 0; q < parts * rows; ++q) tmp[q] = ~(int32_t)(P_.scratch_rows++);
            for (i64 q = 0; q < parts; ++q) {
                const i64 a = k0 + q * per, b = std::min(k1, a + per);
                // rows are addressed relative to r0 in tmp: shift the id arrays
                push_item(r0, r1, tmp.data() + q * rows - r0, q == 0 && init_ids ? init_ids : nullptr, a, b, in_ids, A, level);
            }
            auto ones = [&](i64 i, i64 k) { return (k % rows) == (i - r0) ? 1.0 : 0.0; };
            push_item(r0, r1, out_ids, tmp.data() - r0, 0, (parts - 1) * rows, tmp.data() + rows, ones, level + 1);
            done = level + 1;
        }
        return done;
    }

    // ---- tasks --------------------------------------------------------------------------------------------------------------
    // own unknowns in processing order (scatter: ascending positions, gather: descending); gather kinds read the rows E
    // (ext slots), scatter kinds write one partial row per row of E (slots after the own ones, scratch rows sid...)
    void emit_task(const Unit &u, i64 level)
    {
        const i64 lo = u.lo, hi = u.hi, n_own = hi - lo + 1, m = u.e_cnt;
        if (slot_.empty()) slot_.assign(n_, -1);
        const int32_t n_ext = scatter_ ? 0 : (int32_t)m;
        const int32_t n_total = (int32_t)(n_own + m);  // slots in use
        const int32_t zero_slot = n_total;
        ClusterHeader h{n_ext, (int32_t)(n_total - n_ext), 0, 0, (i64)P_.task_ids.size(), (i64)P_.task_recs.size() / 4};
        auto pos_at = [&](i64 k) { return scatter_ ? lo + k : hi - k; };
        if (!scatter_)
            for (i64 k = 0; k < m; ++k) {
                slot_[epool_[u.e_off + k]] = (int32_t)k;
                P_.task_ids.push_back((int32_t)row_of(epool_[u.e_off + k]));
            }
        for (i64 k = 0; k < n_own; ++k) {
            slot_[pos_at(k)] = n_ext + (int32_t)k;
            P_.task_ids.push_back((int32_t)row_of(pos_at(k)));
        }
        if (scatter_)
            for (i64 k = 0; k < m; ++k) {
                slot_[epool_[u.e_off + k]] = (int32_t)(n_own + k);
                P_.task_ids.push_back(~(int32_t)(u.sid + k));
            }
        // ---- rows of the task: its own unknowns (x = (b - sum T x_src) / d) and, for the scatter kinds, the partial rows
        //      u_q = sum_c T[q, c] x_c over the task's columns; each with the level of its latest source inside the task ----
        struct Row { int32_t slot, level; i64 e0, e1; double scale; bool partial; };
        std::vector<Row> rows;
        std::vector<std::pair<int32_t, double>> ent;  // (source slot, coefficient as subtracted)
        std::vector<int32_t> slot_level((size_t)n_total + 1, -1);  // level at which a slot's value is final (-1: on arrival)
        for (i64 k = 0; k < n_own; ++k) {
            const i64 r = row_of(pos_at(k));
            Row row{n_ext + (int32_t)k, 0, (i64)ent.size(), 0, 1.0 / R_.diag[r], false};
            for (i64 e = R_.ptr[r]; e < R_.ptr[r + 1]; ++e) {
                const int32_t sl = slot_[pos_of(R_.idx[e])];
                ent.emplace_back(sl, R_.val[e]);
                row.level = std::max(row.level, slot_level[sl] + 1);
            }
            row.e1 = (i64)ent.size();
            slot_level[row.slot] = row.level;
            rows.push_back(row);
        }
        if (scatter_ && m > 0) {
            struct Tr { int32_t k, sl; double v; };
            std::vector<Tr> tr;
            for (i64 p = lo; p <= hi; ++p) {
                const i64 r = row_of(p);
                for (i64 e = S_.ptr[r]; e < S_.ptr[r + 1]; ++e) {
                    const i64 q = pos_of(S_.idx[e]);
                    if (q > hi) tr.push_back(Tr{slot_[q] - (int32_t)n_own, slot_[p], S_.val[e]});
                }
            }
            std::stable_sort(tr.begin(), tr.end(), [](const Tr &x, const Tr &y) { return x.k < y.k; });
            for (size_t at = 0; at < tr.size();) {
                Row row{(int32_t)n_own + tr[at].k, 0, (i64)ent.size(), 0, 1.0, true};
                const int32_t k = tr[at].k;
                for (; at < tr.size() && tr[at].k == k; ++at) {
                    ent.emplace_back(tr[at].sl, -tr[at].v);  // acc -= (-T) x: the partial row is +sum T x
                    row.level = std::max(row.level, slot_level[tr[at].sl] + 1);
                }
                row.e1 = (i64)ent.size();
                rows.push_back(row);
            }
        }
        // ---- level by level: the level's rows are dealt to the four lane groups (longest first, to the least loaded
        //      group), a group runs its rows one after the other, and the level is padded to whole windows ----
        int32_t nlev = 0;
        for (const Row &row : rows) nlev = std::max(nlev, row.level + 1);
        std::vector<std::vector<int32_t>> by_level((size_t)nlev);
        for (size_t x = 0; x < rows.size(); ++x) by_level[rows[x].level].push_back((int32_t)x);
        const SnTaskRec nothing{0.0, (uint32_t)zero_slot, SnTaskRec::kNothing};
        std::vector<SnTaskRec> seq[4];
        for (int32_t l = 0; l < nlev; ++l) {
            std::vector<int32_t> &rl = by_level[l];
            std::stable_sort(rl.begin(), rl.end(), [&](int32_t x, int32_t y) { return rows[x].e1 - rows[x].e0 > rows[y].e1 - rows[y].e0; });
            for (int g = 0; g < 4; ++g) seq[g].clear();
            for (int32_t x : rl) {
                int g = 0;
                for (int q = 1; q < 4; ++q)
                    if (seq[q].size() < seq[g].size()) g = q;
                const Row &row = rows[x];
                seq[g].push_back(SnTaskRec{0.0, (uint32_t)(row.partial ? zero_slot : row.slot), SnTaskRec::kBegin});
                for (i64 e = row.e0; e < row.e1; ++e) seq[g].push_back(SnTaskRec{ent[e].second, (uint32_t)ent[e].first, SnTaskRec::kEntry});
                seq[g].push_back(SnTaskRec{row.scale, (uint32_t)row.slot, SnTaskRec::kEnd});
            }
            size_t len = 0;
            for (int g = 0; g < 4; ++g) len = std::max(len, seq[g].size());
            len = (len + kTaskWindow - 1) / kTaskWindow * kTaskWindow;
            for (size_t st = 0; st < len; ++st)
                for (int g = 0; g < 4; ++g) P_.task_recs.push_back(st < seq[g].size() ? seq[g][st] : nothing);
            P_.padded_flops_per_col += 2 * 4 * (i64)len;
        }
        // whole rings of windows: the kernel's loop body is unconditional (exact vmcnt bookkeeping needs straight-line code)
        while (((i64)P_.task_recs.size() / 4 - h.grp_off) % (kTaskWindow * kTaskDepth))
            for (int g = 0; g < 4; ++g) P_.task_recs.push_back(nothing);
        h.ngroups = (int32_t)((i64)P_.task_recs.size() / 4 - h.grp_off);
        tasks_.push_back(h);
        task_level_.push_back((int32_t)level);
        P_.task_slots_max = std::max<int32_t>(P_.task_slots_max, n_total);
        last_level_ = std::max(last_level_, level);
    }

    // ---- scatter kinds (lsolve, usolve): ascending positions, fan-out through partial rows -------------------------------
    void emit_scatter()
    {
        // partial rows: one per (unit, row of its E); incoming[q] = the partial rows that target position q
        std::vector<i64> inc_ptr(n_ + 1, 0);
        for (Unit &u : units_) {
            u.sid = P_.scratch_rows;
            P_.scratch_rows += u.e_cnt;
            P_.partial_rows += u.e_cnt;
            for (i64 k = 0; k < u.e_cnt; ++k) inc_ptr[epool_[u.e_off + k] + 1]++;
        }
        for (i64 q = 0; q < n_; ++q) inc_ptr[q + 1] += inc_ptr[q];
        std::vector<int32_t> inc((size_t)inc_ptr[n_]);
        {
            std::vector<i64> fill(inc_ptr.begin(), inc_ptr.end() - 1);
            for (const Unit &u : units_)
                for (i64 k = 0; k < u.e_cnt; ++k) inc[fill[epool_[u.e_off + k]]++] = (int32_t)(u.sid + k);
        }
        std::vector<int32_t> part_level((size_t)P_.scratch_rows, 0);  // level at which a partial row is complete
        std::vector<double> Dinv, TE, W;
        std::vector<int32_t> own, yid, eid, in;
        for (Unit &u : units_) {
            if (u.task) {
                emit_task(u, 0);
                for (i64 k = 0; k < u.e_cnt; ++k) part_level[u.sid + k] = 0;
                continue;
            }
            const i64 w = u.hi - u.lo + 1, m = u.e_cnt;
            own.resize(w);
            yid.resize(w);
            eid.resize(m);
            u.ysid = P_.scratch_rows;
            P_.scratch_rows += w;
            for (i64 j = 0; j < w; ++j) {
                own[j] = (int32_t)row_of(u.lo + j);
                yid[j] = ~(int32_t)(u.ysid + j);
            }
            for (i64 k = 0; k < m; ++k) eid[k] = ~(int32_t)(u.sid + k);
            // y = b - (incoming partial rows), 16 rows at a time
            i64 lr = -1;
            for (i64 j = 0; j < w; ++j)
                for (i64 x = inc_ptr[u.lo + j]; x < inc_ptr[u.lo + j + 1]; ++x) lr = std::max<i64>(lr, part_level[inc[x]]);
            const i64 red_level = lr + 1;
            i64 y_level = red_level;
            for (i64 j0 = 0; j0 < w; j0 += 16) {
                const i64 j1 = std::min(w, j0 + 16);
                in.clear();
                std::vector<int32_t> tgt;
                for (i64 j = j0; j < j1; ++j)
                    for (i64 x = inc_ptr[u.lo + j]; x < inc_ptr[u.lo + j + 1]; ++x) {
                        in.push_back(~inc[x]);
                        tgt.push_back((int32_t)j);
                    }
                auto minus_one = [&](i64 i, i64 k) { return tgt[k] == i ? -1.0 : 0.0; };
                emit_product_rows(j0, j1, yid.data(), own.data(), (i64)in.size(), in.data(), minus_one, red_level);
                y_level = std::max(y_level, last_product_level_);
            }
            const i64 g_level = y_level + 1;
            i64 p_level;
            node_matrices(u, Dinv, TE);
            {
                // [x ; u] = [D^-1 ; T_E D^-1] y.  Wide nodes sit at the top of the forest, where a level holds a handful of
                // them: 16-row items with short input lists, so that a level is many short wavefronts instead of a few long ones
                fold_inverse(TE, Dinv, m, w, W);
                std::vector<int32_t> out(own);
                out.insert(out.end(), eid.begin(), eid.end());
                auto A = [&](i64 i, i64 k) { return i < w ? (k <= i ? Dinv[(size_t)i * w + k] : 0.0) : W[(size_t)(i - w) * w + k]; };
                auto tri = [&](i64, i64 r1, i64 &k0, i64 &k1) { k0 = 0; k1 = std::min(w, r1); };
                const bool wide = w > opt_.small_node;
                p_level = emit_product(w + m, out.data(), nullptr, yid.data(), A, tri, g_level, wide ? opt_.wide_rows : 64,
                                                 wide ? opt_.wide_item_k : 0);
            }
            for (i64 k = 0; k < m; ++k) part_level[u.sid + k] = (int32_t)p_level;
        }
    }

    // one group of at most 64 output rows [r0, r1) with the whole input list (used by the reductions)
    template <class AF>
    void emit_product_rows(i64 r0, i64 r1, const int32_t *out_ids, const int32_t *init_ids, i64 K, const int32_t *in_ids, AF &&A, i64 level)
    {
        last_product_level_ = level;
        if (K <= opt_.max_item_k) {
            push_item(r0, r1, out_ids, init_ids, 0, K, in_ids, A, level);
            return;
        }
        const i64 rows = r1 - r0, parts = (K + opt_.max_item_k - 1) / opt_.max_item_k;
        const i64 per = ((K + parts - 1) / parts + 3) & ~(i64)3;
        std::vector<int32_t> tmp((size_t)(parts * rows));
        for (i64 q = 0; q < parts * rows; ++q) tmp[q] = ~(int32_t)(P_.scratch_rows++);
        for (i64 q = 0; q < parts; ++q)
            push_item(r0, r1, tmp.data() + q * rows - r0, q == 0 ? init_ids : nullptr, q * per, std::min(K, (q + 1) * per), in_ids, A, level);
        auto ones = [&](i64 i, i64 k) { return (k % rows) == (i - r0) ? 1.0 : 0.0; };
        push_item(r0, r1, out_ids, tmp.data() - r0, 0, (parts - 1) * rows, tmp.data() + rows, ones, level + 1);
        last_product_level_ = level + 1;
    }

    // ---- gather kinds (ltsolve, utsolve): descending positions, every unit reads the finished rows below it --------------
    void emit_gather()
    {
        std::vector<int32_t> avail(n_, 0);  // level at which x at a position is final
        std::vector<double> Dinv, TE, W;
        std::vector<int32_t> own, yid, eid, in;
        for (i64 ui = (i64)units_.size() - 1; ui >= 0; --ui) {
            Unit &u = units_[ui];
            const i64 w = u.hi - u.lo + 1, m = u.e_cnt;
            i64 level = 0;
            for (i64 k = 0; k < m; ++k) level = std::max<i64>(level, avail[epool_[u.e_off + k]] + 1);
            if (u.task) {
                emit_task(u, level);
                for (i64 p = u.lo; p <= u.hi; ++p) avail[p] = (int32_t)level;
                continue;
            }
            own.resize(w);
            eid.resize(m);
            for (i64 j = 0; j < w; ++j) own[j] = (int32_t)row_of(u.lo + j);
            for (i64 k = 0; k < m; ++k) eid[k] = (int32_t)row_of(epool_[u.e_off + k]);
            node_matrices(u, Dinv, TE);
            i64 done;
            fold_inverse(TE, Dinv, m, w, W);
            auto A = [&](i64 i, i64 k) { return k < w ? (k >= i ? Dinv[(size_t)k * w + i] : 0.0) : -W[(size_t)(k - w) * w + i]; };
            if (w <= opt_.small_node) {
                // x = D^-T b - (T_E D^-1)^T x_E in ONE item per node (in place: the item reads b before it writes x)
                in.assign(own.begin(), own.end());
                in.insert(in.end(), eid.begin(), eid.end());
                auto all = [&](i64, i64, i64 &k0, i64 &k1) { k0 = 0; k1 = w + m; };
                done = emit_product(w, own.data(), nullptr, in.data(), A, all, level);
            } else {
                // several items write x while others still read b: b is copied to scratch rows first (level 0: nothing
                // writes these rows of X before the node itself does)
                yid.resize(w);
                u.ysid = P_.scratch_rows;
                P_.scratch_rows += w;
                for (i64 j = 0; j < w; ++j) yid[j] = ~(int32_t)(u.ysid + j);
                auto none = [&](i64, i64) { return 0.0; };
                auto empty = [&](i64, i64, i64 &k0, i64 &k1) { k0 = 0; k1 = 0; };
                emit_product(w, yid.data(), own.data(), eid.data(), none, empty, 0);
                in.assign(yid.begin(), yid.end());
                in.insert(in.end(), eid.begin(), eid.end());
                // D^-T is upper triangular: row group [r0, r1) uses y rows from r0 on, and all of x_E
                auto upper = [&](i64 r0, i64, i64 &k0, i64 &k1) { k0 = r0; k1 = w + m; };
                done = emit_product(w, own.data(), nullptr, in.data(), A, upper, std::max<i64>(level, 1), opt_.wide_rows, opt_.wide_item_k);
            }
            for (i64 p = u.lo; p <= u.hi; ++p) avail[p] = (int32_t)done;
        }
    }

    // ---- order items and tasks by level -------------------------------------------------------------------------------------
    void finish()
    {
        const i64 nlev = last_level_ + 1;
        // gather kinds: tasks whose unknowns nothing else reads (the bottom of the forest) all run in ONE launch at the last level
        if (!scatter_ && !tasks_.empty()) {
            std::vector<char> read(task_lo_.size(), 0);
            auto mark = [&](int32_t id) {
                if (id >= 0 && task_of_[pos_of(id)] >= 0) read[task_of_[pos_of(id)]] = 1;
            };
            for (const SnItem &it : items_)
                for (i64 k = 0; k < 4 * (i64)it.k_steps; ++k) mark(P_.ids[it.in_off + k]);
            for (const ClusterHeader &h : tasks_)
                for (int32_t e = 0; e < h.n_ext; ++e) mark(P_.task_ids[h.ids_off + e]);
            for (size_t c = 0; c < tasks_.size(); ++c) {
                const int32_t own_row = P_.task_ids[tasks_[c].ids_off + tasks_[c].n_ext];
                if (!read[task_of_[pos_of(own_row)]]) task_level_[c] = (int32_t)last_level_;
            }
        }
        auto bucket = [&](const std::vector<int32_t> &level, std::vector<i64> &ptr, std::vector<i64> &order) {
            ptr.assign(nlev + 1, 0);
            for (int32_t l : level) ptr[l + 1]++;
            for (i64 l = 0; l < nlev; ++l) ptr[l + 1] += ptr[l];
            std::vector<i64> fill(ptr.begin(), ptr.end() - 1);
            order.resize(level.size());
            for (size_t k = 0; k < level.size(); ++k) order[fill[level[k]]++] = (i64)k;
        };
        std::vector<i64> order;
        bucket(item_level_, P_.item_level_ptr, order);
        // inside a level: items shared by four wavefronts first, then by two, then the rest (dev::SnLevel)
        for (i64 l = 0; l < nlev; ++l)
            std::stable_sort(order.begin() + P_.item_level_ptr[l], order.begin() + P_.item_level_ptr[l + 1],
                             [&](i64 a, i64 b) { return items_[a].nsplit > items_[b].nsplit; });
        P_.items.resize(items_.size());
        for (size_t k = 0; k < order.size(); ++k) P_.items[k] = items_[order[k]];
        bucket(task_level_, P_.task_level_ptr, order);
        P_.tasks.resize(tasks_.size());
        for (size_t k = 0; k < order.size(); ++k) P_.tasks[k] = tasks_[order[k]];
        // the kernels prefetch with clamped indices and whole vectors: keep those reads in bounds
        P_.ids.resize(P_.ids.size() + 64, kZeroRow);
        P_.coef.resize(P_.coef.size() + 256, 0.0);
    }

    const GatherForm &R_, &S_;
    const i64 n_;
    const bool scatter_, flip_;
    const SnOptions opt_;
    SnProgram P_;
    std::vector<int32_t> task_of_, slot_, eslot_;
    std::vector<i64> task_lo_, task_hi_, stamp_, epool_;
    i64 stamp_id_ = 0, last_level_ = 0, last_product_level_ = 0;
    std::vector<Unit> units_;
    std::vector<SnItem> items_;
    std::vector<int32_t> item_level_, task_level_;
    std::vector<ClusterHeader> tasks_;
};

}  // namespace

SnProgram build_supernodal_program(const GatherForm &R, const GatherForm *C, bool scatter, const SnOptions &opt)
{
    SnBuilder b(R, C, scatter, opt);
    return b.run();
}

ProgramReport check_supernodal_program(const GatherForm &G, const SnProgram &P)
{
    ProgramReport R;
    const i64 n = G.n;
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
    // per row: level and unit of its (single) write, level and unit of its latest read.  Units of one level run
    // concurrently on the device, so within a level nobody may read what another unit writes, or write what another reads.
    struct Track { i64 wlevel = -1, wunit = -1, rlevel = -1, runit = -1; int writes = 0; };
    std::vector<Track> tx(n), ts((size_t)std::max<i64>(P.scratch_rows, 1));
    auto valid = [&](int32_t id) { return id >= 0 ? id < n : (id != SnProgram::kNone && (i64)(~id) < P.scratch_rows); };
    auto value = [&](int32_t id) -> double & { return id >= 0 ? x[id] : scratch[~id]; };
    auto track = [&](int32_t id) -> Track & { return id >= 0 ? tx[id] : ts[~id]; };
    auto note_read = [&](int32_t id, i64 level, i64 unit) {
        Track &t = track(id);
        if (t.wlevel == level && t.wunit != unit) ++R.order_violations;  // written by a concurrent unit
        if (t.rlevel != level || t.runit != unit) {
            if (t.rlevel == level && t.runit != unit) t.runit = -2;  // several readers at this level
            else { t.rlevel = level; t.runit = unit; }
        }
    };
    auto note_write = [&](int32_t id, i64 level, i64 unit) {
        Track &t = track(id);
        if (id == kZeroRow) ++R.bad_index;  // the row of zeros is never written
        if (t.rlevel == level && t.runit != unit) ++R.order_violations;  // a concurrent unit reads it at this level
        if (t.rlevel > level) ++R.order_violations;
        t.wlevel = level;
        t.wunit = unit;
        t.writes++;
    };
    const i64 nlev = (i64)P.item_level_ptr.size() - 1;
    if ((i64)P.task_level_ptr.size() - 1 != nlev) { ++R.bad_index; return R; }
    for (i64 l = 0; l < nlev; ++l) {
        // ---- tasks of this level ----
        for (i64 c = P.task_level_ptr[l]; c < P.task_level_ptr[l + 1]; ++c) {
            const ClusterHeader &h = P.tasks[c];
            const i64 unit = -10 - c;
            const i64 ns = h.n_ext + h.n_own;
            if (ns > P.task_slots_max || ns + 1 > 65535 || h.ids_off < 0 || h.ids_off + ns > (i64)P.task_ids.size() || h.grp_off < 0 ||
                (h.grp_off + h.ngroups) * 4 > (i64)P.task_recs.size() || h.ngroups < 0 || h.ngroups % kTaskWindow) {
                ++R.bad_index;
                continue;
            }
            std::vector<double> lds(ns + 1, 0.0);
            bool ok = true;
            for (i64 s = 0; s < ns; ++s) {
                const int32_t id = P.task_ids[h.ids_off + s];
                if (!valid(id) || (s < h.n_ext && id < 0)) { ++R.bad_index; ok = false; break; }
                if (id >= 0) {
                    note_read(id, l, unit);
                    lds[s] = x[id];
                }
            }
            if (!ok) continue;
            // replay of dev::solve_sparse_tasks: per window, every group reads its four operands, then runs its four steps.
            // written[s]: window in which slot s received its final value (a read in the same or an earlier window than the
            // write it depends on would see the old value on the device)
            std::vector<i64> written(ns + 1, -1);
            std::vector<char> done(ns + 1, 0);
            double acc[4] = {0, 0, 0, 0};
            for (i64 w0 = 0; w0 < h.ngroups; w0 += kTaskWindow) {
                double t[4][kTaskWindow];
                for (int g = 0; g < 4; ++g)
                    for (int u = 0; u < kTaskWindow; ++u) {
                        const SnTaskRec &rec = P.task_recs[(h.grp_off + w0 + u) * 4 + g];
                        if (rec.slot > ns || rec.op > SnTaskRec::kEnd) { ++R.bad_index; t[g][u] = 0.0; continue; }
                        t[g][u] = lds[rec.slot];
                        if (rec.op == SnTaskRec::kEntry && (i64)rec.slot >= h.n_ext && !(done[rec.slot] && written[rec.slot] < w0)) ++R.order_violations;
                        if (rec.op == SnTaskRec::kBegin && rec.slot < ns && done[rec.slot]) ++R.order_violations;
                    }
                for (int u = 0; u < kTaskWindow; ++u)
                    for (int g = 0; g < 4; ++g) {
                        const SnTaskRec &rec = P.task_recs[(h.grp_off + w0 + u) * 4 + g];
                        if (rec.slot > ns) continue;
                        if (rec.op == SnTaskRec::kBegin) acc[g] = t[g][u];
                        else if (rec.op == SnTaskRec::kEntry) acc[g] -= rec.coef * t[g][u];
                        else if (rec.op == SnTaskRec::kEnd) {
                            if ((i64)rec.slot < h.n_ext || (i64)rec.slot >= ns || done[rec.slot]) { ++R.bad_index; continue; }
                            lds[rec.slot] = acc[g] * rec.coef;
                            done[rec.slot] = 1;
                            written[rec.slot] = w0;
                        }
                    }
            }
            for (i64 s = h.n_ext; s < ns; ++s)
                if (!done[s]) ++R.unwritten;
            for (i64 s = h.n_ext; s < ns; ++s) {
                const int32_t id = P.task_ids[h.ids_off + s];
                note_write(id, l, unit);
                value(id) = lds[s];
            }
        }
        // ---- items of this level ----
        for (i64 k = P.item_level_ptr[l]; k < P.item_level_ptr[l + 1]; ++k) {
            const SnItem &it = P.items[k];
            const i64 rows = 16 * (i64)it.m_tiles, K = 4 * (i64)it.k_steps;
            if (it.m_tiles < 1 || it.m_tiles > 4 || it.k_steps < 0 || (it.nsplit != 1 && it.nsplit != 2 && it.nsplit != 4) || it.a_off < 0 ||
                it.a_off + K * rows > (i64)P.coef.size() || it.out_off < 0 || it.out_off + rows > (i64)P.ids.size() || it.in_off < 0 ||
                it.in_off + K > (i64)P.ids.size() || (it.has_init && (it.init_off < 0 || it.init_off + rows > (i64)P.ids.size()))) {
                ++R.bad_index;
                continue;
            }
            std::vector<double> acc(rows, 0.0);
            if (it.has_init)
                for (i64 i = 0; i < rows; ++i) {
                    const int32_t id = P.ids[it.init_off + i];
                    if (id == SnProgram::kNone) continue;
                    if (!valid(id)) { ++R.bad_index; continue; }
                    note_read(id, l, k);
                    acc[i] = value(id);
                }
            for (i64 kk = 0; kk < K; ++kk) {
                const int32_t id = P.ids[it.in_off + kk];
                if (!valid(id)) { ++R.bad_index; continue; }
                const double *cf = P.coef.data() + it.a_off + ((kk >> 2) * it.m_tiles) * 64 + (kk & 3) * 16;
                bool used = false;
                for (i64 i = 0; i < rows; ++i) {
                    const double c = cf[(i >> 4) * 64 + (i & 15)];
                    used = used || c != 0.0;
                    acc[i] += c * value(id);
                }
                if (used || id != kZeroRow) note_read(id, l, k);
            }
            for (i64 i = 0; i < rows; ++i) {
                const int32_t id = P.ids[it.out_off + i];
                if (id == SnProgram::kNone) continue;
                if (!valid(id)) { ++R.bad_index; continue; }
                note_write(id, l, k);
                value(id) = acc[i];
            }
        }
    }
    double top = 1e-300, err = 0.0;
    for (i64 i = 0; i < n; ++i) {
        if (tx[i].writes != 1) ++R.unwritten;
        top = std::max(top, std::fabs(want[i]));
        const double e = std::fabs(x[i] - want[i]);
        if (!(e <= err)) err = e;
    }
    for (i64 s = 1; s < P.scratch_rows; ++s)
        if (ts[s].writes > 1) ++R.order_violations;  // scratch rows are written at most once
    R.max_rel_error = err / top;
    return R;
}

}  // namespace rsp
