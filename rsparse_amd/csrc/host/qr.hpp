// Host sparse QR (qr.cpp): the consumer of usolve / utsolve, rsparse::qrsol (lib.rs:927-956).
#pragma once
#include "sparse.hpp"

namespace rsp {

// rsparse::data::Nmrc for QR (data.rs:1064-1074): l = V (Householder vectors, diagonal first in every column),
// u = R (m2 x n, diagonal last in every column), b = beta
struct QrFactors {
    Csc V, R;
    std::vector<double> beta;
};

// sqr(a, order, qr = true), lib.rs:1114-1133: q, parent (column etree), cp (n ROW counts of R), pinv (2m + n entries:
// row permutation, n zeros, leftmost), m2, lnz = nnz(V), unz = nnz(R)
Symbolic symbolic_qr(const CscView &A, int order);
// qr, lib.rs:788-877.  kBadInput when S does not belong to A.
int sparse_qr(const CscView &A, const Symbolic &S, QrFactors &out);
// x <- H(n-1) .. H(0) x (reverse == false: the order qrsol uses for m >= n, lib.rs:937-940) or H(0) .. H(n-1) x
void qr_apply_reflections(const QrFactors &F, bool reverse, double *x);
Csc transpose_csc(const CscView &A);  // transpose, lib.rs:1178-1200

}  // namespace rsp
