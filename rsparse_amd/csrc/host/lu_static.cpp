// Symbolic half of the GPU numeric LU: the patterns of L and U that rsparse::lu
// (/root/reference/src/lib.rs:519-622) produces WHEN EVERY COLUMN TAKES ITS DIAGONAL AS PIVOT (row q[k] for column k --
// what threshold pivoting does whenever |diagonal| >= tol * max|column|, lib.rs:585-589), in the reference's own
// storage order: U(:,k) = the already pivotal rows of the column's depth-first reach in reach order, diagonal last;
// L(:,k) = the pivot first, then the other rows of the reach in reach order (lib.rs:577-609), row indices of L in pivot
// order (lib.rs:615-617).  Integer work only: the same depth-first search as host/lu.cpp (lib.rs:2256-2277, 1916-1975;
// neighbours visited in storage order) with the pivot of column k known in advance, pinv[q[k]] = k.
// The device computes the values on these patterns (device/lu_engine.hip) and checks the assumption column by column;
// when it fails anywhere the caller falls back to sparse_lu.
#include <algorithm>

#include "sparse.hpp"

namespace rsp {

// out.L / out.U: p and i final, x sized and zero; out.pinv; level[k] = 0 for a column that depends on no other, else
// 1 + the largest level among the columns in U(:,k).  false: some column's diagonal row is not in its reach (a
// structurally zero pivot), duplicate entries inside a column of A, or sizes beyond 32-bit device indices.
bool lu_static_pattern(const CscView &A, const i64 *q, LuFactors &out, std::vector<int32_t> &level)
{
    const i64 n = A.n;
    if (A.m != n || n >= INT32_MAX) return false;
    Csc &L = out.L, &U = out.U;
    L.m = L.n = U.m = U.n = n;
    L.p.assign(n + 1, 0);
    U.p.assign(n + 1, 0);
    L.i.clear();
    U.i.clear();
    const i64 guess = 4 * A.nnz() + n;
    L.i.reserve(guess);
    U.i.reserve(guess);
    std::vector<i64> &pinv = out.pinv;
    pinv.assign(n, -1);
    // the pivot row of every column is known: inverse of q
    std::vector<i64> col_of_row(n, -1);  // row r is the pivot of column col_of_row[r]
    for (i64 k = 0; k < n; ++k) {
        const i64 r = q ? q[k] : k;
        if (r < 0 || r >= n || col_of_row[r] >= 0) return false;
        col_of_row[r] = k;
    }
    level.assign(n, 0);
    std::vector<i64> order(n), stack(n), resume(n), seen(n, -1), dup(n, -1);
    for (i64 k = 0; k < n; ++k) {
        L.p[k] = (i64)L.i.size();
        U.p[k] = (i64)U.i.size();
        const i64 col = q ? q[k] : k;
        // reach of A(:,col) through the columns of L built so far; rows are ORIGINAL row numbers here, as in lu.cpp: row j is
        // pivotal iff pinv[j] >= 0, and then column pinv[j] of L (stored with original row numbers until the end) continues the search
        i64 top = n;
        for (i64 e = A.p[col]; e < A.p[col + 1]; ++e) {
            const i64 root = A.i[e];
            if (root < 0 || root >= n) return false;
            if (seen[root] == k) {
                // (a second entry with the same row is either a duplicate or already reached: duplicates are checked below)
                continue;
            }
            i64 depth = 0;
            stack[0] = root;
            while (depth >= 0) {
                const i64 j = stack[depth];
                const i64 jcol = pinv[j];
                if (seen[j] != k) {
                    seen[j] = k;
                    resume[depth] = jcol < 0 ? 0 : L.p[jcol];
                }
                const i64 end = jcol < 0 ? 0 : L.p[jcol + 1];  // (jcol < k: final)
                bool finished = true;
                for (i64 p = resume[depth]; p < end; ++p) {
                    const i64 i = L.i[p];
                    if (seen[i] == k) continue;
                    resume[depth] = p;
                    stack[++depth] = i;
                    finished = false;
                    break;
                }
                if (finished) {
                    --depth;
                    order[--top] = j;
                }
            }
        }
        // duplicate row indices inside the column of A: the reference keeps the LAST value (x[i] = a.x[p] in order); the device
        // scatters a column's entries in parallel, so such matrices stay on the host
        for (i64 e = A.p[col]; e < A.p[col + 1]; ++e) {
            if (dup[A.i[e]] == k) return false;
            dup[A.i[e]] = k;
        }
        bool diagonal_reached = false;
        for (i64 t = top; t < n; ++t) {
            const i64 i = order[t];
            if (pinv[i] >= 0) {
                U.i.push_back(pinv[i]);
                level[k] = std::max(level[k], level[pinv[i]] + 1);
            } else if (i == col) {
                diagonal_reached = true;
            }
        }
        if (!diagonal_reached) return false;
        U.i.push_back(k);  // U(k,k) last
        pinv[col] = k;
        L.i.push_back(col);  // L(k,k) first
        for (i64 t = top; t < n; ++t) {
            const i64 i = order[t];
            if (pinv[i] < 0) L.i.push_back(i);
        }
        L.p[k + 1] = (i64)L.i.size();
        if ((i64)L.i.size() >= INT32_MAX || (i64)U.i.size() >= INT32_MAX) return false;
    }
    L.p[n] = (i64)L.i.size();
    U.p[n] = (i64)U.i.size();
    // every row became a pivot (q is a permutation); row indices of L in pivot order
    for (auto &r : L.i) r = pinv[r];
    L.x.assign(L.i.size(), 0.0);
    U.x.assign(U.i.size(), 0.0);
    return true;
}

}  // namespace rsp
