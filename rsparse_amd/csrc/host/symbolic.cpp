// Host symbolic layer for the Cholesky path: C = triu(P A P'), elimination tree, pattern of L, column
// pointers, supernodes.  Given the permutation, every result here is mathematically unique, so any
// correct algorithm reproduces the reference's integers (schol, /root/reference/src/lib.rs:968-986,
// and the pattern that chol builds, lib.rs:295-334).  Column counts are obtained from the row
// subtrees directly (one pass that counts, one that fills) instead of the skeleton-matrix algorithm the
// reference uses (lib.rs:1797-1897); the fill pass is needed anyway to hand the pattern to the GPU.
#include <algorithm>

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

bool cholesky_pattern(const CscView &C, const std::vector<i64> &parent, const std::vector<i64> &cp, std::vector<i64> &Li)
{
    const i64 n = C.n;
    Li.assign(cp[n], 0);
    std::vector<i64> fill(cp.begin(), cp.end() - 1), mark(n, -1);
    bool ok = true;
    // rows are appended in increasing k, and the diagonal of column k is written when row k is finished, i.e.
    // before any later row can append to column k: diagonal first, then ascending rows (lib.rs:316-319, 329-332)
    for (i64 k = 0; k < n && ok; ++k) {
        for_each_in_row_subtree(C, parent, k, mark, [&](i64 col) {
            if (fill[col] < cp[col + 1]) Li[fill[col]++] = k;
            else ok = false;
        });
        if (fill[k] != cp[k]) ok = false;  // the diagonal must be the first entry of its column
        else Li[fill[k]++] = k;
    }
    for (i64 j = 0; j < n && ok; ++j) ok = fill[j] == cp[j + 1];
    return ok;
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
