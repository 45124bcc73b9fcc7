// Left-looking sparse LU with threshold partial pivoting on the host.
//
// north_star keeps this stage on the CPU: its pivot sequence, the storage order of L and U and their
// values must equal rsparse::lu (/root/reference/src/lib.rs:519-622) bit for bit.  That fixes the
// algorithm (a sparse triangular solve over the depth-first reach of each column, neighbours visited in
// storage order) and the floating-point operation order; this file must be compiled with
// -ffp-contract=off.  The organisation is this library's own: visit marks live in a stamp array instead
// of flipped column pointers, and the reach/solve/pivot steps are methods of one workspace object.  The depth-first
// search -- it walks every entry of every column of L it reaches -- reads a 32-bit copy of L's row indices and 32-bit
// marks (6-8 % off the whole factorisation; the result is the same integers).
#include <climits>
#include <cmath>

#include "sparse.hpp"

namespace rsp {
namespace {

class LeftLookingLU {
   public:
    LeftLookingLU(const CscView &A, const i64 *q, double tol, LuFactors &out)
        : A_(A), q_(q), tol_(tol), n_(A.n), L_(out.L), U_(out.U), pinv_(out.pinv)
    {
        const i64 guess = 4 * A.nnz() + n_;  // sqr's estimate, lib.rs:1135-1136
        L_.m = L_.n = U_.m = U_.n = n_;
        L_.p.assign(n_ + 1, 0);
        U_.p.assign(n_ + 1, 0);
        L_.i.reserve(guess); L_.x.reserve(guess);
        U_.i.reserve(guess); U_.x.reserve(guess);
        pinv_.assign(n_, -1);
        x_.assign(n_, 0.0);
        order_.assign(n_, 0);
        stack_.assign(n_, 0);
        resume_.assign(n_, 0);
        seen_.assign(n_, -1);
        Li32_.reserve(guess);
    }

    int run()
    {
        for (i64 k = 0; k < n_; ++k) {
            L_.p[k] = (i64)L_.i.size();
            U_.p[k] = (i64)U_.i.size();
            const i64 col = q_ ? q_[k] : k;
            const i64 top = solve_column(col, k);
            // pivot search (lib.rs:566-589): first strictly largest |x| among non-pivotal rows, in reach order;
            // rows already pivotal go to U(:,k)
            i64 ipiv = -1;
            double best = -1.0;
            for (i64 t = top; t < n_; ++t) {
                const i64 i = order_[t];
                if (pinv_[i] < 0) {
                    const double a = std::fabs(x_[i]);
                    if (a > best) { best = a; ipiv = i; }
                } else {
                    U_.i.push_back(pinv_[i]);
                    U_.x.push_back(x_[i]);
                }
            }
            if (ipiv == -1 || best <= 0.0) return kNoPivot;
            if (pinv_[col] < 0 && std::fabs(x_[col]) >= best * tol_) ipiv = col;  // prefer the diagonal
            const double pivot = x_[ipiv];
            U_.i.push_back(k);  // U(k,k) is the LAST entry of column k
            U_.x.push_back(pivot);
            pinv_[ipiv] = k;
            L_.i.push_back(ipiv);  // L(k,k) = 1 is the FIRST entry of column k
            Li32_.push_back((int32_t)ipiv);
            L_.x.push_back(1.0);
            for (i64 t = top; t < n_; ++t) {
                const i64 i = order_[t];
                if (pinv_[i] < 0) {
                    L_.i.push_back(i);
                    Li32_.push_back((int32_t)i);
                    L_.x.push_back(x_[i] / pivot);
                }
                x_[i] = 0.0;
            }
        }
        L_.p[n_] = (i64)L_.i.size();
        U_.p[n_] = (i64)U_.i.size();
        for (auto &r : L_.i) r = pinv_[r];  // row indices of L in pivot order (lib.rs:615-617)
        return kOk;
    }

   private:
    // order_[top..n) = rows reachable from the pattern of A(:,col) through the columns of L built so far, in
    // reverse depth-first finishing order (lib.rs:2256-2277, 1916-1975); then x = L \ A(:,col) (lib.rs:2333-2365)
    i64 solve_column(i64 col, i64 k)
    {
        i64 top = n_;
        for (i64 e = A_.p[col]; e < A_.p[col + 1]; ++e) {
            const i64 root = A_.i[e];
            const int32_t mark = (int32_t)k;
            if (seen_[root] == mark) continue;
            i64 depth = 0;
            stack_[0] = root;
            const int32_t *li = Li32_.data();
            while (depth >= 0) {
                const i64 j = stack_[depth];
                const i64 jcol = pinv_[j];  // column of L that row j pivots, or < 0
                if (seen_[j] != mark) {
                    seen_[j] = mark;
                    resume_[depth] = jcol < 0 ? 0 : L_.p[jcol];
                }
                const i64 end = jcol < 0 ? 0 : (jcol + 1 <= k ? column_end(jcol, k) : 0);
                bool finished = true;
                for (i64 p = resume_[depth]; p < end; ++p) {
                    const int32_t i = li[p];
                    if (seen_[i] == mark) continue;
                    resume_[depth] = p;
                    stack_[++depth] = i;
                    finished = false;
                    break;
                }
                if (finished) {
                    --depth;
                    order_[--top] = j;
                }
            }
        }
        for (i64 t = top; t < n_; ++t) x_[order_[t]] = 0.0;
        for (i64 e = A_.p[col]; e < A_.p[col + 1]; ++e) x_[A_.i[e]] = A_.x[e];
        for (i64 t = top; t < n_; ++t) {
            const i64 j = order_[t];
            const i64 jcol = pinv_[j];
            if (jcol < 0) continue;
            const double xj = x_[j];
            const i64 end = column_end(jcol, k);
            for (i64 p = L_.p[jcol] + 1; p < end; ++p) x_[L_.i[p]] -= L_.x[p] * xj;
        }
        return top;
    }

    // end of column c of L while column k is being built (L_.p[c+1] is final for c+1 <= k)
    i64 column_end(i64 c, i64 k) const { return c + 1 <= k ? L_.p[c + 1] : (i64)L_.i.size(); }

    const CscView &A_;
    const i64 *q_;
    const double tol_;
    const i64 n_;
    Csc &L_, &U_;
    std::vector<i64> &pinv_;
    std::vector<double> x_;
    std::vector<i64> order_, stack_, resume_;
    std::vector<int32_t> seen_;  // column number of the last search that reached the row
    std::vector<int32_t> Li32_;  // L_.i again, 32 bits wide: what the search reads
};

}  // namespace

int sparse_lu(const CscView &A, const i64 *q, double tol, LuFactors &out)
{
    if (A.m != A.n || A.n >= INT32_MAX) return kBadInput;  // (the device side takes no larger systems either)
    LeftLookingLU f(A, q, tol, out);
    return f.run();
}

}  // namespace rsp
