"""Deterministic synthetic inputs for the benchmark configurations.

Random numbers come from SplitMix64 (u = (z >> 11) * 2^-53), never numpy's RNG, so any other
implementation of the same generator reproduces the inputs bit for bit.  Matrices are returned as
(m, n, p, i, x) CSC tuples with sorted rows and, for SPD cases, full symmetric storage.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed, count):
    """count uniform doubles in [0,1) from SplitMix64 seeded with `seed` (vectorised: state_k = seed + (k+1)*gamma)."""
    with np.errstate(over="ignore"):
        k = np.arange(1, count + 1, dtype=np.uint64)
        z = (np.uint64(seed) + k * np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def splitmix64_torch(seed, count, device, lo=0.0, hi=1.0, chunk=1 << 24):
    """The same `count` doubles as splitmix64(seed, count), mapped to [lo, hi), generated on `device` as a torch tensor
    (int64 arithmetic wraps like uint64; logical right shifts are arithmetic shifts with the sign extension masked off),
    so that a benchmark block of gigabytes never exists on the host."""
    import torch

    def s64(v):  # a uint64 constant as the int64 with the same bits
        return v - (1 << 64) if v >= (1 << 63) else v

    def lsr(z, k):
        return (z >> k) & ((1 << (64 - k)) - 1)

    out = torch.empty(count, dtype=torch.float64, device=device)
    gamma, m1, m2 = s64(0x9E3779B97F4A7C15), s64(0xBF58476D1CE4E5B9), s64(0x94D049BB133111EB)
    for start in range(0, count, chunk):
        stop = min(count, start + chunk)
        k = torch.arange(start + 1, stop + 1, dtype=torch.int64, device=device)
        z = k * gamma + s64(seed & 0xFFFFFFFFFFFFFFFF)
        z = (z ^ lsr(z, 30)) * m1
        z = (z ^ lsr(z, 27)) * m2
        z = z ^ lsr(z, 31)
        u = lsr(z, 11).to(torch.float64) * (1.0 / 9007199254740992.0)
        out[start:stop] = u * (hi - lo) + lo if (lo != 0.0 or hi != 1.0) else u
    return out


def _csc_from_coo(n, rows, cols, vals):
    import scipy.sparse as sp
    A = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsc()  # sums duplicates
    A.sort_indices()
    return (n, n, A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.astype(np.float64))


def laplacian_2d(nx, ny=None):
    """5-point Laplacian, Dirichlet, idx = y*nx + x, diag 4, off-diag -1 (config C2 at nx = ny = 1024)."""
    ny = nx if ny is None else ny
    n = nx * ny
    idx = np.arange(n, dtype=np.int64).reshape(ny, nx)
    rows, cols, vals = [idx.ravel()], [idx.ravel()], [np.full(n, 4.0)]
    for a, b in ((idx[:, 1:], idx[:, :-1]), (idx[1:, :], idx[:-1, :])):
        rows += [a.ravel(), b.ravel()]
        cols += [b.ravel(), a.ravel()]
        vals += [np.full(a.size, -1.0)] * 2
    return _csc_from_coo(n, np.concatenate(rows), np.concatenate(cols), np.concatenate(vals))


def laplacian_3d(nx):
    """7-point Laplacian nx^3, idx = (z*nx + y)*nx + x, diag 6, off-diag -1 (config C3 at nx = 80)."""
    n = nx ** 3
    idx = np.arange(n, dtype=np.int64).reshape(nx, nx, nx)
    rows, cols, vals = [idx.ravel()], [idx.ravel()], [np.full(n, 6.0)]
    for a, b in ((idx[:, :, 1:], idx[:, :, :-1]), (idx[:, 1:, :], idx[:, :-1, :]), (idx[1:, :, :], idx[:-1, :, :])):
        rows += [a.ravel(), b.ravel()]
        cols += [b.ravel(), a.ravel()]
        vals += [np.full(a.size, -1.0)] * 2
    return _csc_from_coo(n, np.concatenate(rows), np.concatenate(cols), np.concatenate(vals))


def banded_unsymmetric(n, per_col=9, band=50, seed=4):
    """Config C4: per column `per_col` off-diagonals at rows (j + d) mod n, d ~ U{-band..band}\\{0 allowed, summed},
    values U(-1,1), diagonal = 1 + sum |column| (strict column diagonal dominance)."""
    u = splitmix64(seed, 2 * n * per_col)
    d = np.floor(u[: n * per_col] * (2 * band + 1)).astype(np.int64) - band
    v = 2.0 * u[n * per_col:] - 1.0
    cols = np.repeat(np.arange(n, dtype=np.int64), per_col)
    rows = (cols + d) % n
    off = rows != cols
    rows, cols, v = rows[off], cols[off], v[off]
    colsum = np.zeros(n)
    np.add.at(colsum, cols, np.abs(v))
    rows = np.concatenate([rows, np.arange(n, dtype=np.int64)])
    cols = np.concatenate([cols, np.arange(n, dtype=np.int64)])
    v = np.concatenate([v, 1.0 + colsum])
    return _csc_from_coo(n, rows, cols, v)


def circuit_spd(n, seed_topology=6, seed_values=100):
    """Config C5: chain edges (i,i+1) + 2 random long edges per node, conductances U(0.1,1),
    A = graph Laplacian + diag U(0.01,0.1)  => SPD."""
    ut = splitmix64(seed_topology, 2 * n)
    far = np.floor(ut * n).astype(np.int64)
    a = np.concatenate([np.arange(n - 1, dtype=np.int64), np.repeat(np.arange(n, dtype=np.int64), 2)])
    b = np.concatenate([np.arange(1, n, dtype=np.int64), far])
    keep = a != b
    a, b = a[keep], b[keep]
    uv = splitmix64(seed_values, len(a) + n)
    g = 0.1 + 0.9 * uv[: len(a)]
    shift = 0.01 + 0.09 * uv[len(a):]
    deg = np.zeros(n)
    np.add.at(deg, a, g)
    np.add.at(deg, b, g)
    rows = np.concatenate([a, b, np.arange(n, dtype=np.int64)])
    cols = np.concatenate([b, a, np.arange(n, dtype=np.int64)])
    vals = np.concatenate([-g, -g, deg + shift])
    return _csc_from_coo(n, rows, cols, vals)


def random_sparse(n, density, seed, symmetric_pd=False):
    """Small random test matrices (unsorted-row variants are made by the tests themselves)."""
    nnz = max(1, int(density * n * n))
    u = splitmix64(seed, 3 * nnz)
    rows = np.floor(u[:nnz] * n).astype(np.int64)
    cols = np.floor(u[nnz: 2 * nnz] * n).astype(np.int64)
    vals = 2.0 * u[2 * nnz:] - 1.0
    if symmetric_pd:
        rows, cols, vals = np.concatenate([rows, cols]), np.concatenate([cols, rows]), np.concatenate([vals, vals])
    A = _csc_from_coo(n, rows, cols, vals)
    import scipy.sparse as sp
    M = sp.csc_matrix((A[4], A[3], A[2]), shape=(n, n))
    colsum = np.asarray(abs(M).sum(axis=0)).ravel()
    M = (M + sp.diags(colsum + 1.0)).tocsc()
    M.sort_indices()
    return (n, n, M.indptr.astype(np.int64), M.indices.astype(np.int64), M.data.astype(np.float64))
