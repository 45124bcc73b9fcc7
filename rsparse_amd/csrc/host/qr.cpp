// Host sparse QR for the qrsol driver (the consumer of usolve / utsolve, lib.rs:927-956): symbolic analysis
// sqr(a, order, qr = true) (lib.rs:1114-1133: amd on A'A, column elimination tree, row counts of R, the row permutation
// of vcount lib.rs:2450-2530) and the left-looking Householder factorisation qr (lib.rs:788-877) with happly / house
// (lib.rs:2099-2147).  Integers (q, parent, cp, pinv, m2, lnz, unz, the patterns of V and R and their storage order)
// reproduce the reference's; values follow its order of operations (compiled without contraction).  The triangular
// solves with R run on the GPU (capi.hip); everything here is host work by design, like lu.
#include <cmath>

#include "qr.hpp"

namespace rsp {

namespace {

// etree(a, ata = true), lib.rs:2026-2071: the elimination tree of (A(:,q))' A(:,q) without forming the product.  Two
// columns that share a row are related; prev[row] remembers the last column seen in that row.
template <class ColOf>
std::vector<i64> column_elimination_tree(const CscView &A, ColOf col_of)
{
    std::vector<i64> parent(A.n, -1), ancestor(A.n, -1), prev(A.m, -1);
    for (i64 k = 0; k < A.n; ++k) {
        const i64 c = col_of(k);
        for (i64 e = A.p[c]; e < A.p[c + 1]; ++e) {
            i64 j = prev[A.i[e]];
            while (j != -1 && j < k) {
                const i64 up = ancestor[j];
                ancestor[j] = k;
                if (up == -1) parent[j] = k;
                j = up;
            }
            prev[A.i[e]] = k;
        }
    }
    return parent;
}

}  // namespace

Symbolic symbolic_qr(const CscView &A, int order)
{
    const i64 m = A.m, n = A.n;
    Symbolic S;
    S.n = n;
    S.has_q = approximate_minimum_degree(A, order, S.q);
    auto col_of = [&](i64 k) { return S.has_q ? S.q[k] : k; };
    S.parent = column_elimination_tree(A, col_of);

    // Symb.pinv of the QR analysis holds three things (vcount, lib.rs:2450-2530): [0, m) the row permutation,
    // [m, m + n) unused zeros, [m + n, 2m + n) leftmost[i] = first column (of A(:,q)) with an entry in row i
    S.has_pinv = true;
    S.pinv.assign(2 * m + n, 0);
    i64 *rowperm = S.pinv.data(), *leftmost = S.pinv.data() + m + n;
    for (i64 i = 0; i < m; ++i) leftmost[i] = -1;
    for (i64 k = n - 1; k >= 0; --k) {
        const i64 c = col_of(k);
        for (i64 e = A.p[c]; e < A.p[c + 1]; ++e) leftmost[A.i[e]] = k;
    }
    // Every column k owns a queue of rows: first the rows whose leftmost column is k, ascending.  Column k takes the
    // head of its queue as its pivot row (a fictitious row m2++ if the queue is empty); what is left of the queue moves
    // to the FRONT of the parent's queue.  V(:,k) has one entry per row that was in the queue.
    std::vector<i64> next(m, -1), head(n, -1), tail(n, -1), queued(n, 0);
    for (i64 i = m - 1; i >= 0; --i) {
        rowperm[i] = -1;
        const i64 k = leftmost[i];
        if (k < 0) continue;  // an empty row
        if (queued[k]++ == 0) tail[k] = i;
        next[i] = head[k];
        head[k] = i;
    }
    S.m2 = m;
    S.lnz = 0;
    for (i64 k = 0; k < n; ++k) {
        i64 pivot = head[k];
        S.lnz += 1;
        if (pivot < 0) pivot = S.m2++;
        if (pivot < m) rowperm[pivot] = k;
        if (--queued[k] <= 0) continue;
        S.lnz += queued[k];
        const i64 pa = S.parent[k];
        if (pa == -1) continue;
        if (queued[pa] == 0) tail[pa] = tail[k];
        next[tail[k]] = head[pa];
        head[pa] = next[pivot];
        queued[pa] += queued[k];
    }
    for (i64 i = 0, k = n; i < m; ++i)
        if (rowperm[i] < 0) rowperm[i] = k++;

    // cp[j] = entries in ROW j of R (= column j of the Cholesky factor of A'A, what counts(.., ata = true) returns,
    // lib.rs:1797-1897), counted from the patterns the factorisation will build: R(:,k) is k plus the union of the tree
    // paths from leftmost[i] towards k over the rows i of column k.
    S.cp.assign(n, 0);
    std::vector<i64> stamp(n, -1);
    S.unz = 0;
    for (i64 k = 0; k < n; ++k) {
        stamp[k] = k;
        S.cp[k] += 1;
        const i64 c = col_of(k);
        for (i64 e = A.p[c]; e < A.p[c + 1]; ++e)
            for (i64 j = leftmost[A.i[e]]; j != -1 && stamp[j] != k; j = S.parent[j]) {
                stamp[j] = k;
                S.cp[j] += 1;
            }
    }
    for (i64 k = 0; k < n; ++k) S.unz += S.cp[k];
    return S;
}

namespace {

// happly, lib.rs:2099-2111: x -= v (beta v'x) with v = V(:,j)
inline void apply_reflection(const Csc &V, i64 j, i64 end, double beta, double *x)
{
    double tau = 0.0;
    for (i64 e = V.p[j]; e < end; ++e) tau += V.x[e] * x[V.i[e]];
    tau *= beta;
    for (i64 e = V.p[j]; e < end; ++e) x[V.i[e]] -= V.x[e] * tau;
}

// house, lib.rs:2116-2147: v (in place), beta and the norm s such that (I - beta v v') x = s e1
inline double make_reflector(double *x, i64 len, double &beta)
{
    double sigma = 0.0;
    for (i64 i = 1; i < len; ++i) sigma += x[i] * x[i];
    if (sigma == 0.0) {
        const double s = std::fabs(x[0]);
        beta = x[0] <= 0.0 ? 2.0 : 0.0;
        x[0] = 1.0;
        return s;
    }
    const double s = std::sqrt(x[0] * x[0] + sigma);
    x[0] = x[0] <= 0.0 ? x[0] - s : -sigma / (x[0] + s);
    beta = -1.0 / (s * x[0]);
    return s;
}

}  // namespace

int sparse_qr(const CscView &A, const Symbolic &S, QrFactors &F)
{
    const i64 m = A.m, n = A.n, m2 = S.m2;
    if (S.n != n || !S.has_pinv || (i64)S.pinv.size() != 2 * m + n || (i64)S.parent.size() != n || m2 < m || S.lnz < n || S.unz < n ||
        (S.has_q && (i64)S.q.size() < n))
        return kBadInput;
    const i64 *rowperm = S.pinv.data(), *leftmost = S.pinv.data() + m + n;
    for (i64 i = 0; i < m; ++i)
        if (rowperm[i] < 0 || rowperm[i] >= m2 || leftmost[i] < -1 || leftmost[i] >= n) return kBadInput;
    Csc &V = F.V, &R = F.R;
    V.m = R.m = m2;
    V.n = R.n = n;
    V.p.assign(n + 1, 0);
    R.p.assign(n + 1, 0);
    V.i.assign(S.lnz, 0);
    V.x.assign(S.lnz, 0.0);
    R.i.assign(S.unz, 0);
    R.x.assign(S.unz, 0.0);
    F.beta.assign(n, 0.0);
    std::vector<i64> stamp(m2, -1), chain(n), path(n);
    std::vector<double> x(m2, 0.0);
    i64 vnz = 0, rnz = 0;
    for (i64 k = 0; k < n; ++k) {
        V.p[k] = vnz;
        R.p[k] = rnz;
        const i64 v0 = vnz;
        if (vnz >= S.lnz) return kBadInput;
        stamp[k] = k;
        V.i[vnz++] = k;  // V(k,k) leads its column
        i64 top = n;
        const i64 col = S.has_q ? S.q[k] : k;
        if (col < 0 || col >= n) return kBadInput;
        for (i64 e = A.p[col]; e < A.p[col + 1]; ++e) {
            // pattern of R(:,k): climb from the leftmost column of this row to the first node already reached for k;
            // each climb goes below the earlier ones on the stack, in climbing order
            i64 len = 0;
            for (i64 j = leftmost[A.i[e]]; j != -1 && stamp[j] != k; j = S.parent[j]) {
                chain[len++] = j;
                stamp[j] = k;
            }
            while (len > 0) path[--top] = chain[--len];
            const i64 r = rowperm[A.i[e]];
            x[r] = A.x[e];
            if (r > k && stamp[r] < k) {  // below the diagonal: a row of V(:,k)
                if (vnz >= S.lnz) return kBadInput;
                V.i[vnz++] = r;
                stamp[r] = k;
            }
        }
        for (i64 t = top; t < n; ++t) {
            const i64 j = path[t];  // R(j,k) is an entry: apply reflection j, then x(j) is final
            apply_reflection(V, j, V.p[j + 1], F.beta[j], x.data());
            if (rnz >= S.unz) return kBadInput;
            R.i[rnz] = j;
            R.x[rnz++] = x[j];
            x[j] = 0.0;
            if (S.parent[j] == k)  // the rows of a child's reflector carry over (scatter_no_x, lib.rs:2310-2329)
                for (i64 e = V.p[j]; e < V.p[j + 1]; ++e) {
                    const i64 r = V.i[e];
                    if (stamp[r] < k) {
                        if (vnz >= S.lnz) return kBadInput;
                        stamp[r] = k;
                        V.i[vnz++] = r;
                    }
                }
        }
        for (i64 e = v0; e < vnz; ++e) {
            V.x[e] = x[V.i[e]];
            x[V.i[e]] = 0.0;
        }
        if (rnz >= S.unz) return kBadInput;
        R.i[rnz] = k;  // the diagonal closes the column (usolve's convention)
        V.p[k + 1] = vnz;
        R.x[rnz++] = make_reflector(V.x.data() + v0, vnz - v0, F.beta[k]);
    }
    V.p[n] = vnz;
    R.p[n] = rnz;
    V.i.resize(vnz);
    V.x.resize(vnz);
    R.i.resize(rnz);
    R.x.resize(rnz);
    return kOk;
}

void qr_apply_reflections(const QrFactors &F, bool reverse, double *x)
{
    const i64 n = F.V.n;
    if (!reverse)
        for (i64 k = 0; k < n; ++k) apply_reflection(F.V, k, F.V.p[k + 1], F.beta[k], x);
    else
        for (i64 k = n - 1; k >= 0; --k) apply_reflection(F.V, k, F.V.p[k + 1], F.beta[k], x);
}

// transpose (lib.rs:1178-1200): rows of every column of the result ascend
Csc transpose_csc(const CscView &A)
{
    Csc T;
    T.m = A.n;
    T.n = A.m;
    T.p.assign(A.m + 1, 0);
    for (i64 e = 0; e < A.p[A.n]; ++e) T.p[A.i[e] + 1]++;
    for (i64 i = 0; i < A.m; ++i) T.p[i + 1] += T.p[i];
    T.i.resize(A.p[A.n]);
    T.x.resize(A.p[A.n]);
    std::vector<i64> fill(T.p.begin(), T.p.end() - 1);
    for (i64 j = 0; j < A.n; ++j)
        for (i64 e = A.p[j]; e < A.p[j + 1]; ++e) {
            const i64 q = fill[A.i[e]]++;
            T.i[q] = j;
            T.x[q] = A.x ? A.x[e] : 0.0;
        }
    return T;
}

}  // namespace rsp
