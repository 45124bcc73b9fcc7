// Approximate-minimum-degree ordering on the host.
//
// The permutation must equal, entry for entry, what rsparse's private `amd` returns
// (/root/reference/src/lib.rs:1292-1752) because everything downstream (etree, L pattern, level sets)
// is graded bit-exact.  The algorithm is therefore the same quotient-graph elimination with the same
// tie-breaking; the organisation is this library's own: the graph C is built pattern-only, and the
// elimination is a small state machine (QuotientGraph) with one method per phase.
#include <algorithm>
#include <cmath>

#include "sparse.hpp"

namespace rsp {
namespace {

inline i64 flip(i64 v) { return -v - 2; }

// pattern of A' (row lists of A, each in ascending column order)
Pattern transpose_of(const CscView &A)
{
    Pattern T;
    T.m = A.n;
    T.n = A.m;
    T.p.assign(A.m + 1, 0);
    T.i.resize(A.nnz());
    for (i64 e = 0; e < A.nnz(); ++e) T.p[A.i[e] + 1]++;
    for (i64 r = 0; r < A.m; ++r) T.p[r + 1] += T.p[r];
    std::vector<i64> fill(T.p.begin(), T.p.end() - 1);
    for (i64 j = 0; j < A.n; ++j)
        for (i64 e = A.p[j]; e < A.p[j + 1]; ++e) T.i[fill[A.i[e]]++] = j;
    return T;
}

// Appends the not-yet-seen rows of column j of (p, i) to `out`; stamp[] remembers the current column tag.
inline void append_new_rows(const i64 *p, const i64 *i, i64 j, i64 tag, std::vector<i64> &stamp, std::vector<i64> &out)
{
    for (i64 e = p[j]; e < p[j + 1]; ++e) {
        const i64 r = i[e];
        if (stamp[r] < tag) {
            stamp[r] = tag;
            out.push_back(r);
        }
    }
}

// pattern of A + A' with the entry order of lib.rs:247-271 (column of A first, then column of A')
Pattern union_with_transpose(const CscView &A, const Pattern &AT)
{
    Pattern C;
    C.m = A.m;
    C.n = A.n;
    C.p.assign(A.n + 1, 0);
    C.i.reserve(2 * A.nnz());
    std::vector<i64> stamp(A.m, 0);
    for (i64 j = 0; j < A.n; ++j) {
        C.p[j] = (i64)C.i.size();
        append_new_rows(A.p, A.i, j, j + 1, stamp, C.i);
        append_new_rows(AT.p.data(), AT.i.data(), j, j + 1, stamp, C.i);
    }
    C.p[A.n] = (i64)C.i.size();
    return C;
}

// pattern of L * R with the entry order of lib.rs:713-748
Pattern product_pattern(const i64 lm, const i64 *lp, const i64 *li, const i64 rn, const i64 *rp, const i64 *ri)
{
    Pattern C;
    C.m = lm;
    C.n = rn;
    C.p.assign(rn + 1, 0);
    std::vector<i64> stamp(lm, 0);
    for (i64 j = 0; j < rn; ++j) {
        C.p[j] = (i64)C.i.size();
        for (i64 e = rp[j]; e < rp[j + 1]; ++e) append_new_rows(lp, li, ri[e], j + 1, stamp, C.i);
    }
    C.p[rn] = (i64)C.i.size();
    return C;
}

// The quotient graph of the elimination.  Object e is a live variable, a live element, or dead; its
// adjacency lives in store[start[e] .. start[e]+len[e]).  Names follow the classic AMD paper; the
// phase order and every tie-break follow lib.rs:1380-1749.
class QuotientGraph {
   public:
    QuotientGraph(Pattern &&C, i64 dense_threshold) : n_(C.n), dense_(dense_threshold)
    {
        // drop the diagonal (lib.rs:1371) while moving the pattern in
        start_.assign(n_ + 1, 0);
        store_.reserve(C.i.size() + C.i.size() / 5 + 2 * n_);
        for (i64 j = 0; j < n_; ++j) {
            start_[j] = (i64)store_.size();
            for (i64 e = C.p[j]; e < C.p[j + 1]; ++e)
                if (C.i[e] != j) store_.push_back(C.i[e]);
        }
        used_ = (i64)store_.size();
        start_[n_] = used_;
        store_.resize(used_ + used_ / 5 + 2 * n_);  // lib.rs:1374
        const i64 N = n_ + 1;
        len_.assign(N, 0); nv_.assign(N, 1); next_.assign(N, -1); head_.assign(N, -1); elen_.assign(N, 0);
        degree_.assign(N, 0); w_.assign(N, 1); hhead_.assign(N, -1); last_.assign(N, -1);
        for (i64 k = 0; k < n_; ++k) len_[k] = start_[k + 1] - start_[k];
        for (i64 k = 0; k <= n_; ++k) degree_[k] = len_[k];
    }

    std::vector<i64> order()
    {
        seed_degree_lists();
        while (nel_ < n_) {
            pick_pivot();
            if (elenk_ > 0 && used_ + mindeg_ >= (i64)store_.size()) compact();
            form_element();
            scan_set_differences();
            update_degrees();
            merge_indistinguishable();
            close_element();
        }
        return postorder();
    }

   private:
    i64 clear_marks(i64 mark, i64 lemax)  // lib.rs:2534-2546
    {
        if (mark < 2 || mark + lemax < 0) {
            for (i64 k = 0; k < n_; ++k)
                if (w_[k] != 0) w_[k] = 1;
            mark = 2;
        }
        return mark;
    }

    void seed_degree_lists()  // lib.rs:1394-1423
    {
        mark_ = clear_marks(0, 0);
        elen_[n_] = -2;
        start_[n_] = -1;
        w_[n_] = 0;
        for (i64 i = 0; i < n_; ++i) {
            const i64 d = degree_[i];
            if (d == 0) {  // isolated: already an (empty) root element
                elen_[i] = -2;
                ++nel_;
                start_[i] = -1;
                w_[i] = 0;
            } else if (d > dense_) {  // dense: absorbed into the dummy element n
                nv_[i] = 0;
                elen_[i] = -1;
                ++nel_;
                start_[i] = flip(n_);
                nv_[n_]++;
            } else {
                if (head_[d] != -1) last_[head_[d]] = i;
                next_[i] = head_[d];
                head_[d] = i;
            }
        }
    }

    void pick_pivot()  // lib.rs:1428-1444
    {
        for (;;) {
            k_ = head_[mindeg_];
            if (!(mindeg_ < n_ && k_ == -1)) break;
            ++mindeg_;
        }
        if (next_[k_] != -1) last_[next_[k_]] = -1;
        head_[mindeg_] = next_[k_];
        elenk_ = elen_[k_];
        nvk_ = nv_[k_];
        nel_ += nvk_;
    }

    void compact()  // garbage collection, lib.rs:1447-1475
    {
        for (i64 j = 0; j < n_; ++j) {
            const i64 s = start_[j];
            if (s >= 0) {
                start_[j] = store_[s];
                store_[s] = flip(j);
            }
        }
        i64 dst = 0, src = 0;
        while (src < used_) {
            const i64 j = flip(store_[src++]);
            if (j >= 0) {
                store_[dst] = start_[j];
                start_[j] = dst++;
                for (i64 t = 0; t < len_[j] - 1; ++t) store_[dst++] = store_[src++];
            }
        }
        used_ = dst;
    }

    void form_element()  // lib.rs:1477-1534
    {
        dk_ = 0;
        nv_[k_] = -nvk_;
        i64 cursor = start_[k_];
        pk1_ = (elenk_ == 0) ? cursor : used_;
        pk2_ = pk1_;
        for (i64 step = 1; step <= elenk_ + 1; ++step) {
            i64 e, src, count;
            if (step > elenk_) {
                e = k_;
                src = cursor;
                count = len_[k_] - elenk_;
            } else {
                e = store_[cursor++];
                src = start_[e];
                count = len_[e];
            }
            for (i64 t = 0; t < count; ++t) {
                const i64 i = store_[src++];
                const i64 nvi = nv_[i];
                if (nvi <= 0) continue;
                dk_ += nvi;
                nv_[i] = -nvi;
                store_[pk2_++] = i;
                if (next_[i] != -1) last_[next_[i]] = last_[i];
                if (last_[i] != -1) next_[last_[i]] = next_[i];
                else head_[degree_[i]] = next_[i];
            }
            if (e != k_) {
                start_[e] = flip(k_);
                w_[e] = 0;
            }
        }
        if (elenk_ != 0) used_ = pk2_;
        degree_[k_] = dk_;
        start_[k_] = pk1_;
        len_[k_] = pk2_ - pk1_;
        elen_[k_] = -2;
    }

    void scan_set_differences()  // lib.rs:1537-1558
    {
        mark_ = clear_marks(mark_, lemax_);
        for (i64 pk = pk1_; pk < pk2_; ++pk) {
            const i64 i = store_[pk];
            const i64 eln = elen_[i];
            if (eln <= 0) continue;
            const i64 nvi = -nv_[i];
            const i64 wnvi = mark_ - nvi;
            for (i64 q = start_[i]; q <= start_[i] + eln - 1; ++q) {
                const i64 e = store_[q];
                if (w_[e] >= mark_) w_[e] -= nvi;
                else if (w_[e] != 0) w_[e] = degree_[e] + wnvi;
            }
        }
    }

    void update_degrees()  // lib.rs:1561-1625
    {
        for (i64 pk = pk1_; pk < pk2_; ++pk) {
            const i64 i = store_[pk];
            const i64 p1 = start_[i];
            const i64 p2 = p1 + elen_[i] - 1;
            i64 pn = p1;
            uint64_t hash = 0;  // the reference sums into a usize and reduces mod n (lib.rs:1579,1599,1617)
            i64 d = 0;
            for (i64 q = p1; q <= p2; ++q) {
                const i64 e = store_[q];
                if (w_[e] == 0) continue;
                const i64 dext = w_[e] - mark_;
                if (dext > 0) {
                    d += dext;
                    store_[pn++] = e;
                    hash += (uint64_t)e;
                } else {  // aggressive absorption
                    start_[e] = flip(k_);
                    w_[e] = 0;
                }
            }
            elen_[i] = pn - p1 + 1;
            const i64 p3 = pn, p4 = p1 + len_[i];
            for (i64 q = p2 + 1; q < p4; ++q) {
                const i64 j = store_[q];
                const i64 nvj = nv_[j];
                if (nvj <= 0) continue;
                d += nvj;
                store_[pn++] = j;
                hash += (uint64_t)j;
            }
            if (d == 0) {  // mass elimination
                start_[i] = flip(k_);
                const i64 nvi = -nv_[i];
                dk_ -= nvi;
                nvk_ += nvi;
                nel_ += nvi;
                nv_[i] = 0;
                elen_[i] = -1;
            } else {
                degree_[i] = std::min(degree_[i], d);
                store_[pn] = store_[p3];
                store_[p3] = store_[p1];
                store_[p1] = k_;
                len_[i] = pn - p1 + 1;
                const i64 h = (i64)(hash % (uint64_t)n_);
                next_[i] = hhead_[h];
                hhead_[h] = i;
                last_[i] = h;
            }
        }
        degree_[k_] = dk_;
        lemax_ = std::max(lemax_, dk_);
        mark_ = clear_marks(mark_ + lemax_, lemax_);
    }

    void merge_indistinguishable()  // supernode detection, lib.rs:1628-1679
    {
        for (i64 pk = pk1_; pk < pk2_; ++pk) {
            i64 i = store_[pk];
            if (nv_[i] >= 0) continue;
            const i64 h = last_[i];
            i = hhead_[h];
            hhead_[h] = -1;
            for (; i != -1 && next_[i] != -1; i = next_[i], ++mark_) {
                const i64 ln = len_[i], eln = elen_[i];
                for (i64 q = start_[i] + 1; q <= start_[i] + ln - 1; ++q) w_[store_[q]] = mark_;
                i64 jlast = i;
                for (i64 j = next_[i]; j != -1;) {
                    bool same = len_[j] == ln && elen_[j] == eln;
                    for (i64 q = start_[j] + 1; same && q <= start_[j] + ln - 1; ++q)
                        if (w_[store_[q]] != mark_) same = false;
                    if (same) {
                        start_[j] = flip(i);
                        nv_[i] += nv_[j];
                        nv_[j] = 0;
                        elen_[j] = -1;
                        j = next_[j];
                        next_[jlast] = j;
                    } else {
                        jlast = j;
                        j = next_[j];
                    }
                }
            }
        }
    }

    void close_element()  // lib.rs:1682-1714
    {
        i64 dst = pk1_;
        for (i64 pk = pk1_; pk < pk2_; ++pk) {
            const i64 i = store_[pk];
            const i64 nvi = -nv_[i];
            if (nvi <= 0) continue;
            nv_[i] = nvi;
            i64 d = degree_[i] + dk_ - nvi;
            d = std::min(d, n_ - nel_ - nvi);
            if (head_[d] != -1) last_[head_[d]] = i;
            next_[i] = head_[d];
            last_[i] = -1;
            head_[d] = i;
            mindeg_ = std::min(mindeg_, d);
            degree_[i] = d;
            store_[dst++] = i;
        }
        nv_[k_] = nvk_;
        len_[k_] = dst - pk1_;
        if (len_[k_] == 0) {
            start_[k_] = -1;
            w_[k_] = 0;
        }
        if (elenk_ != 0) used_ = dst;
    }

    std::vector<i64> postorder()  // lib.rs:1718-1749 with tdfs, lib.rs:2412-2446
    {
        for (i64 i = 0; i < n_; ++i) start_[i] = flip(start_[i]);  // start_ now holds assembly-tree parents
        for (i64 j = 0; j <= n_; ++j) head_[j] = -1;
        for (i64 j = n_; j >= 0; --j) {  // absorbed variables hang under their representative
            if (nv_[j] > 0) continue;
            next_[j] = head_[start_[j]];
            head_[start_[j]] = j;
        }
        for (i64 e = n_; e >= 0; --e) {  // elements hang under their parent element
            if (nv_[e] <= 0) continue;
            if (start_[e] != -1) {
                next_[e] = head_[start_[e]];
                head_[start_[e]] = e;
            }
        }
        std::vector<i64> perm(n_ + 1, 0), stack(n_ + 2, 0);
        i64 k = 0;
        for (i64 root = 0; root <= n_; ++root) {
            if (start_[root] != -1) continue;
            i64 top = 0;
            stack[0] = root;
            while (top >= 0) {
                const i64 p = stack[top];
                const i64 child = head_[p];
                if (child == -1) {
                    --top;
                    perm[k++] = p;
                } else {
                    head_[p] = next_[child];
                    stack[++top] = child;
                }
            }
        }
        return perm;
    }

    const i64 n_, dense_;
    std::vector<i64> start_, store_;
    std::vector<i64> len_, nv_, next_, head_, elen_, degree_, w_, hhead_, last_;
    i64 used_ = 0, nel_ = 0, mindeg_ = 0, lemax_ = 0, mark_ = 0;
    // state of the pivot being eliminated
    i64 k_ = -1, elenk_ = 0, nvk_ = 0, dk_ = 0, pk1_ = 0, pk2_ = 0;
};

}  // namespace

Pattern transpose_pattern(const Pattern &A)
{
    CscView v{A.m, A.n, A.p.data(), A.i.data(), nullptr};
    return transpose_of(v);
}

bool approximate_minimum_degree(const CscView &A, int order, std::vector<i64> &perm)
{
    if (order < 0) return false;  // lib.rs:1324
    const i64 m = A.m, n = A.n;
    // dense-row threshold in f32 arithmetic, as lib.rs:1331-1332
    i64 dense = std::max<i64>(16, (i64)(10.0f * std::sqrt((float)n)));
    dense = std::min<i64>(n - 2, dense);

    Pattern AT = transpose_of(A);
    Pattern C;
    if (order == 0 && n == m) {
        C = union_with_transpose(A, AT);
    } else if (order == 1) {
        // drop the rows of A with more than `dense` entries, then C = A'A on what is left (lib.rs:1336-1352)
        i64 kept = 0;
        for (i64 r = 0; r < m; ++r) {
            const i64 b = AT.p[r], e = AT.p[r + 1];
            AT.p[r] = kept;
            if (e - b > dense) continue;
            for (i64 q = b; q < e; ++q) AT.i[kept++] = AT.i[q];
        }
        AT.p[m] = kept;
        AT.i.resize(kept);
        Pattern A2 = transpose_pattern(AT);
        C = product_pattern(AT.m, AT.p.data(), AT.i.data(), A2.n, A2.p.data(), A2.i.data());
    } else {
        C = product_pattern(AT.m, AT.p.data(), AT.i.data(), A.n, A.p, A.i);
    }
    QuotientGraph g(std::move(C), dense);
    perm = g.order();
    return true;
}

}  // namespace rsp
