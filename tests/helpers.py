"""Shared test helpers: fixture loading and matrix construction (data only, no solver logic)."""
import ast
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")


def load_sprs(name):
    """Parse the reference's text matrix format (data.rs:414-424): six `key: value` lines."""
    vals = {}
    with open(os.path.join(GOLDEN, "assets", name)) as f:
        for line in f:
            key, _, val = line.partition(":")
            vals[key.strip()] = ast.literal_eval(val.strip())
    m, n = int(vals["m"]), int(vals["n"])
    p = np.array(vals["p"], np.int64)
    nz = int(p[n])
    return (m, n, p, np.array(vals["i"], np.int64)[:nz], np.array(vals["x"], np.float64)[:nz])


def from_dense(rows):
    """Sprs::from_vec (data.rs:289-314): column by column, rows ascending, exact zeros skipped."""
    a = np.array(rows, dtype=np.float64)
    m, n = a.shape
    p, i, x = [0], [], []
    for j in range(n):
        for r in range(m):
            if a[r, j] != 0.0:
                i.append(r)
                x.append(a[r, j])
        p.append(len(i))
    return (m, n, np.array(p, np.int64), np.array(i, np.int64), np.array(x, np.float64))


def reference_cases():
    with open(os.path.join(GOLDEN, "reference_cases.json")) as f:
        return json.load(f)


def case_matrix(case):
    return load_sprs(case["asset"]) if "asset" in case else from_dense(case["dense"])


def to_scipy(A):
    import scipy.sparse as sp
    m, n, p, i, x = A
    return sp.csc_matrix((x, i, p), shape=(m, n))


def to_dense(A):
    m, n, p, i, x = A
    out = np.zeros((m, n))
    for j in range(n):
        out[np.asarray(i[p[j]:p[j + 1]], np.int64), j] = x[p[j]:p[j + 1]]
    return out


def transpose_csc(A):
    """Data helper for the one reference test that transposes its asset first (solver_tests.rs:921)."""
    m, n, p, i, x = A
    nz = int(p[n])
    cols = np.repeat(np.arange(n, dtype=np.int64), np.diff(p))
    order = np.argsort(np.asarray(i[:nz], np.int64), kind="stable")
    tp = np.zeros(m + 1, np.int64)
    np.add.at(tp, np.asarray(i[:nz], np.int64) + 1, 1)
    return (n, m, np.cumsum(tp), cols[order], np.asarray(x[:nz])[order])


def qr_case(case):
    A = case_matrix(case)
    return transpose_csc(A) if case.get("transpose") else A
