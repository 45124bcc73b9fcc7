// Host symbolic layer for the Cholesky path: C = triu(P A P'), elimination tree, pattern of L, column
// pointers, supernodes.  Given the permutation, every result here is mathematically unique, so any
// correct algorithm reproduces the reference's integers (schol, /root/reference/src/lib.rs:968-986,
// and the pattern that chol builds, lib.rs:295-334).  Column counts are obtained from the row
// subtrees directly (one pass that counts, one that fills) instead of the skeleton-matrix algorithm the
// reference uses (lib.rs:1797-1897); the fill pass is needed anyway to hand the pattern to the GPU.
#include <algorithm>
#include <thread>

#include "parallel.hpp"
#include "sparse.hpp"

namespace rsp {

// symperm (lib.rs:2369-2408): C = upper triangle of A(p,p) given pinv; only entries i <= j of A are read.
Csc symmetric_permute_upper(const CscView &A, const i64 *pinv, std::vector<i64> *source)
{
    const i64 n = A.n;
    Csc C;
    C.m = C.n = n;
    C.p.assign(n + 1, 0);
    for (i64 j = 0; j < n; ++j) {
        const i64 j2 = pinv ? pinv[j] : j;
        for (i64 e = A.p[j]; e < A.p[j + 1]; ++e) {
            const i64 i = A.i[e];
            if (i > j) continue;
            const i64 i2 = pinv ? pinv[i] : i;
            C.p[std::max(i2, j2) + 1]++;
        }
    }
    for (i64 j = 0; j < n; ++j) C.p[j + 1] += C.p[j];
    C.i.resize(C.p[n]);
    C.x.resize(C.p[n]);
    if (source) source->assign(C.p[n], 0);
    std::vector<i64> fill(C.p.begin(), C.p.end() - 1);
    for (i64 j = 0; j < n; ++j) {
        const i64 j2 = pinv ? pinv[j] : j;
        for (i64 e = A.p[j]; e < A.p[j + 1]; ++e) {
            const i64 i = A.i[e];
            if (i > j) continue;
            const i64 i2 = pinv ? pinv[i] : i;
            const i64 q = fill[std::max(i2, j2)]++;
            C.i[q] = std::min(i2, j2);
            C.x[q] = A.x ? A.x[e] : 0.0;
            if (source) (*source)[q] = e;
        }
    }
    return C;
}

// Liu's algorithm with path compression; result == etree(C, false), lib.rs:2026-2071
std::vector<i64> elimination_tree(const CscView &C)
{
    const i64 n = C.n;
    std::vector<i64> parent(n, -1), ancestor(n, -1);
    for (i64 k = 0; k < n; ++k) {
        for (i64 e = C.p[k]; e < C.p[k + 1]; ++e) {
            i64 i = C.i[e];
            while (i != -1 && i < k) {
                const i64 up = ancestor[i];
                ancestor[i] = k;
                if (up == -1) parent[i] = k;
                i = up;
            }
        }
    }
    return parent;
}

namespace {

// Walks the row subtree of row k (the nonzero columns of L(k,:), lib.rs:1985-2022) and calls f(col) for
// each column < k in it.  mark[] must hold values != k for unvisited nodes.
template <class F>
inline void for_each_in_row_subtree(const CscView &C, const std::vector<i64> &parent, i64 k, std::vector<i64> &mark, F &&f)
{
    mark[k] = k;
    for (i64 e = C.p[k]; e < C.p[k + 1]; ++e) {
        i64 i = C.i[e];
        if (i > k) continue;
        for (; i >= 0 && mark[i] != k; i = parent[i]) {
            mark[i] = k;
            f(i);
        }
    }
}

}  // namespace

Symbolic symbolic_cholesky(const CscView &A, int order)
{
    const i64 n = A.n;
    Symbolic S;
    S.n = n;
    std::vector<i64> perm;
    if (approximate_minimum_degree(A, order, perm)) {  // pinvert, lib.rs:2196-2209
        S.has_pinv = true;
        S.pinv.assign(n, 0);
        for (i64 k = 0; k < n; ++k) S.pinv[perm[k]] = k;
    }
    CscView Ap = A;
    Ap.x = nullptr;  // pattern only
    Csc C = symmetric_permute_upper(Ap, S.has_pinv ? S.pinv.data() : nullptr);
    S.parent = elimination_tree(C.view());
    // column counts of L: column i gains one entry for every row k whose subtree contains i, plus the diagonal
    std::vector<i64> count(n, 1), mark(n, -1);
    const CscView Cv = C.view();
    for (i64 k = 0; k < n; ++k) for_each_in_row_subtree(Cv, S.parent, k, mark, [&](i64 col) { count[col]++; });
    S.cp.assign(n + 1, 0);
    for (i64 j = 0; j < n; ++j) S.cp[j + 1] = S.cp[j] + count[j];
    S.lnz = S.unz = S.cp[n];
    return S;
}

Symbolic symbolic_lu(const CscView &A, int order)
{
    Symbolic S;
    S.n = A.n;
    S.has_q = approximate_minimum_degree(A, order, S.q);
    S.unz = S.lnz = 4 * A.nnz() + A.n;  // lib.rs:1135-1136
    return S;
}

// Column structures through the supernodes instead of row by row (which is 47 M scattered appends at C2: the slowest
// stage of a Cholesky analysis).  With parent the elimination tree of C,
//     struct(j) = {j}  U  {k > j : C(j,k) != 0}  U  union over the children c of j of struct(c) \ {c},
// and inside a supernode (parent[j] == j + 1, count[j + 1] == count[j] - 1) struct(j + 1) = struct(j) \ {j}.  So only the
// FIRST column of every supernode is merged (from its children's lists, which are tails of first columns already done:
// the sum of those tails is a small fraction of nnz(L)); the other columns are tails of it, copied on the host threads.
// Everything the partition assumes is verified on the way -- parent against the elimination tree of C, every merged
// list against cp, every other contribution to a later column of a supernode against the first column's list -- so a
// Symb that does not describe this matrix is refused, as before.
bool cholesky_pattern(const CscView &C, const std::vector<i64> &parent, const std::vector<i64> &cp, RawVector<i64> &Li)
{
    const i64 n = C.n;
    if ((i64)parent.size() != n || (i64)cp.size() != n + 1 || cp[0] != 0) return false;
    for (i64 j = 0; j < n; ++j)
        if (cp[j + 1] <= cp[j] || cp[j + 1] - cp[j] > n - j) return false;
    if (n >= INT32_MAX) return false;
    if (parent != elimination_tree(C)) return false;
    // rows of C (upper triangle): row j lists the k > j with C(j,k) != 0
    std::vector<i64> rp((size_t)n + 1, 0);
    for (i64 k = 0; k < n; ++k)
        for (i64 e = C.p[k]; e < C.p[k + 1]; ++e) {
            if (C.i[e] < 0 || C.i[e] > k) return false;
            if (C.i[e] < k) rp[(size_t)C.i[e] + 1]++;
        }
    for (i64 j = 0; j < n; ++j) rp[(size_t)j + 1] += rp[(size_t)j];
    std::vector<int32_t> rk((size_t)rp[(size_t)n]);
    {
        std::vector<i64> at(rp.begin(), rp.end() - 1);
        for (i64 k = 0; k < n; ++k)
            for (i64 e = C.p[k]; e < C.p[k + 1]; ++e)
                if (C.i[e] < k) rk[(size_t)at[(size_t)C.i[e]]++] = (int32_t)k;
    }
    // children lists (ascending)
    std::vector<int32_t> child_head((size_t)n, -1), child_next((size_t)n, -1);
    for (i64 j = n - 1; j >= 0; --j)
        if (parent[(size_t)j] >= 0) {
            child_next[(size_t)j] = child_head[(size_t)parent[(size_t)j]];
            child_head[(size_t)parent[(size_t)j]] = (int32_t)j;
        }
    Li.resize((size_t)cp[(size_t)n]);  // (uninitialised)
    // fresh pages fault in on whoever touches them first: on all threads at once, not one by one in the merge below
    parallel_ranges((i64)Li.size(), (i64)1 << 22, [&](i64 b, i64 e) {
        for (i64 k = b; k < e; k += 512) Li[(size_t)k] = 0;
    });
    std::vector<int32_t> first_of((size_t)n);   // first column of the supernode of column j
    std::vector<int32_t> mark((size_t)n, -1);   // mark[r] == f: r is in struct(f), f the first column being worked on
    std::vector<i64> list;
    // the rows below the diagonal of a finished column c: a tail of its supernode's first column
    const auto below = [&](i64 c, const i64 *&b, const i64 *&e) {
        const i64 f = first_of[(size_t)c];
        b = Li.data() + cp[(size_t)f] + (c - f) + 1;
        e = Li.data() + cp[(size_t)f + 1];
    };
    for (i64 f = 0; f < n;) {
        i64 last = f;
        while (last + 1 < n && parent[(size_t)last] == last + 1 &&
               cp[(size_t)last + 1] - cp[(size_t)last] == cp[(size_t)last + 2] - cp[(size_t)last + 1] + 1)
            ++last;
        for (i64 j = f; j <= last; ++j) first_of[(size_t)j] = (int32_t)f;
        // struct(f): every source is ascending, so what it adds is an ascending run; runs that do not simply continue the
        // list are merged in (a column usually has one child that brings nearly everything)
        list.clear();
        mark[(size_t)f] = (int32_t)f;
        size_t runs = 0;
        bool ascending = true;
        const auto close_run = [&](size_t start) {
            if (start == list.size()) return;
            ++runs;
            if (start > 0 && list[start] < list[start - 1]) {
                if (runs <= 8) std::inplace_merge(list.begin(), list.begin() + (std::ptrdiff_t)start, list.end());
                else ascending = false;
            }
        };
        for (int32_t c = child_head[(size_t)f]; c >= 0; c = child_next[(size_t)c]) {
            const i64 *b, *e;
            below(c, b, e);
            const size_t start = list.size();
            for (; b < e; ++b)
                if (mark[(size_t)*b] != f) {
                    mark[(size_t)*b] = (int32_t)f;
                    list.push_back(*b);
                }
            close_run(start);
        }
        {
            const size_t start = list.size();
            for (i64 e = rp[(size_t)f]; e < rp[(size_t)f + 1]; ++e) {
                const int32_t k = rk[(size_t)e];
                if (mark[(size_t)k] != f) {
                    mark[(size_t)k] = (int32_t)f;
                    list.push_back(k);
                }
            }
            close_run(start);
        }
        if ((i64)list.size() != cp[(size_t)f + 1] - cp[(size_t)f] - 1) return false;
        if (!ascending) std::sort(list.begin(), list.end());
        i64 *out = Li.data() + cp[(size_t)f];
        out[0] = f;
        std::copy(list.begin(), list.end(), out + 1);
        // the later columns of the supernode: their own entries of C and their other children add nothing new
        for (i64 j = f + 1; j <= last; ++j) {
            if (out[j - f] != j) return false;
            for (i64 e = rp[(size_t)j]; e < rp[(size_t)j + 1]; ++e)
                if (mark[(size_t)rk[(size_t)e]] != f) return false;
            for (int32_t c = child_head[(size_t)j]; c >= 0; c = child_next[(size_t)c]) {
                if (c == j - 1) continue;
                const i64 *b, *e;
                below(c, b, e);
                for (; b < e; ++b)
                    if (mark[(size_t)*b] != f) return false;
            }
        }
        f = last + 1;
    }
    // the tails: column j of a supernode that starts at f is struct(f) without its first j - f rows; threads by entries
    const i64 threads = std::max<i64>(1, std::min<i64>(host_threads(), cp[(size_t)n] >> 18));
    std::vector<i64> cut((size_t)threads + 1, n);
    cut[0] = 0;
    for (i64 t = 1; t < threads; ++t)
        cut[(size_t)t] = std::lower_bound(cp.begin(), cp.end() - 1, cp[(size_t)n] * t / threads) - cp.begin();
    const auto tails = [&](i64 j0, i64 j1) {
        for (i64 j = j0; j < j1; ++j) {
            const i64 f = first_of[(size_t)j];
            if (f != j) std::copy(Li.data() + cp[(size_t)f] + (j - f), Li.data() + cp[(size_t)f + 1], Li.data() + cp[(size_t)j]);
        }
    };
    std::vector<std::thread> pool;
    for (i64 t = 1; t < threads; ++t) pool.emplace_back(tails, cut[(size_t)t], cut[(size_t)t + 1]);
    tails(cut[0], cut[1]);
    for (std::thread &th : pool) th.join();
    return true;
}

std::vector<i64> fundamental_supernodes(const std::vector<i64> &parent, const std::vector<i64> &cp)
{
    const i64 n = (i64)parent.size();
    std::vector<i64> start;
    start.push_back(0);
    for (i64 j = 0; j + 1 < n; ++j) {
        const bool chained = parent[j] == j + 1 && (cp[j + 1] - cp[j]) == (cp[j + 2] - cp[j + 1]) + 1;
        if (!chained) start.push_back(j + 1);
    }
    if (n > 0) start.push_back(n);
    return start;
}

}  // namespace rsp
