/*
 * rsparse_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded CPU restatement of the numeric solve path of
 * RLado/rsparse v1.2.1 (itself a Rust port of CSparse v1).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * shipped library (rsparse_amd/csrc) never links or calls anything in here.
 *
 * Fidelity rules (what makes it an oracle):
 *   - every index is int64 where the reference uses isize/usize;
 *   - loop orders, tie-breaks and storage orders follow the cited reference
 *     lines, so integer structure (AMD permutation, etree, column counts,
 *     L/U patterns, LU pivot sequence) is reproduced exactly;
 *   - build with -ffp-contract=off: the reference never fuses a*b+c, and LU
 *     pivot ties / the README known answer depend on that;
 *   - chol's diagonal uses sqrt(d); the reference writes powf(d, 0.5)
 *     (src/lib.rs:332) which differs by at most 1 ulp.
 *
 * Pinned by (tests/test_oracle_golden.py): every hot-path test of the
 * reference's tests/solver_tests.rs, the README full-precision lusol answer
 * (bitwise), and the exact-integer sqr/qr pins of tests/basic_tests.rs:1272-1336.
 * amd(order 0/1), etree/counts(ata=false), the chol L pattern and the LU pivot
 * sequence have no integer fixture in the reference ("parity unpinned" for those
 * integers); they are pinned indirectly through solution vectors.
 *
 * All citations "lib.rs:N" are /root/reference/src/lib.rs, "data.rs:N" is src/data.rs.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t idx;

#define ORC_OK 0
#define ORC_NOT_POSITIVE_DEFINITE 1 /* lib.rs:191 */
#define ORC_NO_PIVOT 2              /* lib.rs:193 */

/* owning CSC matrix, data.rs:194-208; x may be NULL for pattern-only work */
typedef struct {
    idx nzmax, m, n;
    idx *p, *i;
    double *x;
} spm;

static idx imax(idx a, idx b) { return a > b ? a : b; }
static idx imin(idx a, idx b) { return a < b ? a : b; }
static idx flip(idx i) { return -i - 2; }               /* lib.rs:2551 */
static idx unflip(idx i) { return i < 0 ? flip(i) : i; } /* lib.rs:2556 */

static void *xcalloc(idx n, size_t sz) { return calloc((size_t)(n > 0 ? n : 1), sz); }

static spm *spm_new(idx m, idx n, idx nzmax, int values) /* data.rs:226 */
{
    spm *a = (spm *)calloc(1, sizeof(spm));
    a->m = m; a->n = n; a->nzmax = nzmax;
    a->p = (idx *)xcalloc(n + 1, sizeof(idx));
    a->i = (idx *)xcalloc(nzmax, sizeof(idx));
    a->x = values ? (double *)xcalloc(nzmax, sizeof(double)) : NULL;
    return a;
}
static void spm_free(spm *a)
{
    if (!a) return;
    free(a->p); free(a->i); free(a->x); free(a);
}
static void spm_resize(spm *a, idx nzmax)
{
    a->i = (idx *)realloc(a->i, sizeof(idx) * (size_t)(nzmax > 0 ? nzmax : 1));
    if (a->x) a->x = (double *)realloc(a->x, sizeof(double) * (size_t)(nzmax > 0 ? nzmax : 1));
    a->nzmax = nzmax;
}
/* borrowed view -> the oracle never writes through these pointers */
static spm view(idx m, idx n, const idx *p, const idx *i, const double *x)
{
    spm a; a.m = m; a.n = n; a.nzmax = p[n]; a.p = (idx *)p; a.i = (idx *)i; a.x = (double *)x;
    return a;
}

/* lib.rs:1901-1911 */
static idx cumsum(idx *p, idx *c, idx n)
{
    idx nz = 0;
    for (idx k = 0; k < n; k++) { p[k] = nz; nz += c[k]; c[k] = p[k]; }
    p[n] = nz;
    return nz;
}

/* lib.rs:1178-1197 */
static spm *transpose(const spm *a, int values)
{
    idx *w = (idx *)xcalloc(a->m, sizeof(idx));
    spm *c = spm_new(a->n, a->m, a->p[a->n], values && a->x);
    for (idx p = 0; p < a->p[a->n]; p++) w[a->i[p]]++;
    cumsum(c->p, w, a->m);
    for (idx j = 0; j < a->n; j++)
        for (idx p = a->p[j]; p < a->p[j + 1]; p++) {
            idx q = w[a->i[p]]++;
            c->i[q] = j;
            if (c->x) c->x[q] = a->x[p];
        }
    free(w);
    return c;
}

/* lib.rs:2281-2306, pattern part only (values never influence the pattern) */
static idx scatter_pat(const spm *a, idx j, idx *w, idx mark, spm *c, idx nz)
{
    for (idx p = a->p[j]; p < a->p[j + 1]; p++) {
        idx i = a->i[p];
        if (w[i] < mark) { w[i] = mark; c->i[nz++] = i; }
    }
    return nz;
}

/* pattern of lib.rs:247-271 (amd calls add(a, at, 0, 0): values are all zero) */
static spm *add_pat(const spm *a, const spm *b)
{
    idx m = a->m, n = b->n, nz = 0;
    idx *w = (idx *)xcalloc(m, sizeof(idx));
    spm *c = spm_new(m, n, a->p[a->n] + b->p[n], 0);
    for (idx j = 0; j < n; j++) {
        c->p[j] = nz;
        nz = scatter_pat(a, j, w, j + 1, c, nz);
        nz = scatter_pat(b, j, w, j + 1, c, nz);
    }
    c->p[n] = nz;
    free(w);
    return c;
}

/* pattern of lib.rs:713-748 */
static spm *multiply_pat(const spm *a, const spm *b)
{
    idx nz = 0, m = a->m;
    idx *w = (idx *)xcalloc(m, sizeof(idx));
    spm *c = spm_new(m, b->n, 2 * (a->p[a->n] + b->p[b->n]) + m, 0);
    for (idx j = 0; j < b->n; j++) {
        if (nz + m > c->nzmax) spm_resize(c, 2 * c->nzmax + m);
        c->p[j] = nz;
        for (idx p = b->p[j]; p < b->p[j + 1]; p++) nz = scatter_pat(a, b->i[p], w, j + 1, c, nz);
    }
    c->p[b->n] = nz;
    free(w);
    return c;
}

/* lib.rs:2075-2095 with the `diag` predicate of lib.rs:1979 */
static idx drop_diagonal(spm *a)
{
    idx nz = 0;
    for (idx j = 0; j < a->n; j++) {
        idx p = a->p[j];
        a->p[j] = nz;
        for (; p < a->p[j + 1]; p++)
            if (a->i[p] != j) a->i[nz++] = a->i[p];
    }
    a->p[a->n] = nz;
    return nz;
}

/* lib.rs:2412-2446 */
static idx tdfs(idx j, idx k, idx *head, const idx *next, idx *post, idx *stack)
{
    idx top = 0;
    stack[0] = j;
    while (top >= 0) {
        idx p = stack[top];
        idx i = head[p];
        if (i == -1) { top--; post[k++] = p; }
        else { head[p] = next[i]; stack[++top] = i; }
    }
    return k;
}

/* lib.rs:2534-2546 */
static idx wclear(idx mark, idx lemax, idx *w, idx n)
{
    if (mark < 2 || mark + lemax < 0) {
        for (idx k = 0; k < n; k++) if (w[k] != 0) w[k] = 1;
        mark = 2;
    }
    return mark;
}

/* lib.rs:1292-1752.  Returns NULL for the natural ordering, else n+1 entries. */
static idx *amd(const spm *A, int order)
{
    if (order < 0) return NULL;                                     /* :1324 */
    idx m = A->m, n = A->n;
    spm *AT = transpose(A, 0), *C;
    idx dense = imax(16, (idx)(10.0f * sqrtf((float)n)));            /* :1331, f32 arithmetic */
    dense = imin(n - 2, dense);                                     /* :1332 */
    if (order == 0 && n == m) {
        C = add_pat(A, AT);                                         /* :1335 */
    } else if (order == 1) {                                        /* :1336-1352 */
        idx p2 = 0;
        for (idx j = 0; j < m; j++) {
            idx p = AT->p[j];
            AT->p[j] = p2;
            if (AT->p[j + 1] - p > dense) continue;
            for (; p < AT->p[j + 1]; p++) AT->i[p2++] = AT->i[p];
        }
        AT->p[m] = p2;
        spm *A2 = transpose(AT, 0);
        C = multiply_pat(AT, A2);
        spm_free(A2);
    } else {
        C = multiply_pat(AT, A);                                    /* :1354 */
    }
    spm_free(AT);

    idx *P = (idx *)xcalloc(n + 1, sizeof(idx));
    idx *len = (idx *)xcalloc(8 * (n + 1), sizeof(idx));            /* :1359-1369 */
    idx *nv = len + (n + 1), *next = len + 2 * (n + 1), *head = len + 3 * (n + 1);
    idx *elen = len + 4 * (n + 1), *degree = len + 5 * (n + 1), *w = len + 6 * (n + 1);
    idx *hhead = len + 7 * (n + 1), *last = P;
    idx *stack = (idx *)xcalloc(n + 2, sizeof(idx));

    drop_diagonal(C);                                               /* :1371 */
    idx cnz = C->p[n];
    spm_resize(C, cnz + cnz / 5 + 2 * n);                           /* :1374 */
    idx *Cp = C->p, *Ci = C->i;

    /* initialise quotient graph, :1380-1397 */
    for (idx k = 0; k < n; k++) len[k] = Cp[k + 1] - Cp[k];
    len[n] = 0;
    for (idx i = 0; i <= n; i++) {
        head[i] = -1; last[i] = -1; next[i] = -1; hhead[i] = -1;
        nv[i] = 1; w[i] = 1; elen[i] = 0; degree[i] = len[i];
    }
    idx mark = wclear(0, 0, w, n);
    elen[n] = -2; Cp[n] = -1; w[n] = 0;

    /* initialise degree lists, :1400-1423 */
    idx nel = 0;
    for (idx i = 0; i < n; i++) {
        idx d = degree[i];
        if (d == 0) { elen[i] = -2; nel++; Cp[i] = -1; w[i] = 0; }
        else if (d > dense) { nv[i] = 0; elen[i] = -1; nel++; Cp[i] = flip(n); nv[n]++; }
        else {
            if (head[d] != -1) last[head[d]] = i;
            next[i] = head[d];
            head[d] = i;
        }
    }

    idx mindeg = 0, lemax = 0;
    while (nel < n) {                                               /* :1425 */
        /* select node of minimum approximate degree, :1428-1444 */
        idx k;
        for (;;) {
            k = head[mindeg];
            if (!(mindeg < n && k == -1)) break;
            mindeg++;
        }
        if (next[k] != -1) last[next[k]] = -1;
        head[mindeg] = next[k];
        idx elenk = elen[k], nvk = nv[k];
        nel += nvk;

        /* garbage collection, :1447-1475 */
        if (elenk > 0 && cnz + mindeg >= C->nzmax) {
            for (idx j = 0; j < n; j++) {
                idx p = Cp[j];
                if (p >= 0) { Cp[j] = Ci[p]; Ci[p] = flip(j); }
            }
            idx q = 0, p = 0;
            while (p < cnz) {
                idx j = flip(Ci[p++]);
                if (j >= 0) {
                    Ci[q] = Cp[j];
                    Cp[j] = q++;
                    for (idx k3 = 0; k3 < len[j] - 1; k3++) Ci[q++] = Ci[p++];
                }
            }
            cnz = q;
        }

        /* construct new element, :1477-1534 */
        idx dk = 0;
        nv[k] = -nvk;
        idx p = Cp[k];
        idx pk1 = (elenk == 0) ? p : cnz;
        idx pk2 = pk1;
        for (idx k1 = 1; k1 <= elenk + 1; k1++) {
            idx e, pj, ln;
            if (k1 > elenk) { e = k; pj = p; ln = len[k] - elenk; }
            else { e = Ci[p++]; pj = Cp[e]; ln = len[e]; }
            for (idx k2 = 1; k2 <= ln; k2++) {
                idx i = Ci[pj++];
                idx nvi = nv[i];
                if (nvi <= 0) continue;
                dk += nvi;
                nv[i] = -nvi;
                Ci[pk2++] = i;
                if (next[i] != -1) last[next[i]] = last[i];
                if (last[i] != -1) next[last[i]] = next[i];
                else head[degree[i]] = next[i];
            }
            if (e != k) { Cp[e] = flip(k); w[e] = 0; }
        }
        if (elenk != 0) cnz = pk2;
        degree[k] = dk; Cp[k] = pk1; len[k] = pk2 - pk1; elen[k] = -2;

        /* find set differences, :1537-1558 */
        mark = wclear(mark, lemax, w, n);
        for (idx pk = pk1; pk < pk2; pk++) {
            idx i = Ci[pk];
            idx eln = elen[i];
            if (eln <= 0) continue;
            idx nvi = -nv[i];
            idx wnvi = mark - nvi;
            for (idx q = Cp[i]; q <= Cp[i] + eln - 1; q++) {
                idx e = Ci[q];
                if (w[e] >= mark) w[e] -= nvi;
                else if (w[e] != 0) w[e] = degree[e] + wnvi;
            }
        }

        /* degree update, :1561-1625 */
        for (idx pk = pk1; pk < pk2; pk++) {
            idx i = Ci[pk];
            idx p1 = Cp[i];
            idx p2 = p1 + elen[i] - 1;
            idx pn = p1;
            uint64_t h = 0;                                          /* usize hash, :1567,:1579 */
            idx d = 0;
            for (idx q = p1; q <= p2; q++) {
                idx e = Ci[q];
                if (w[e] != 0) {
                    idx dext = w[e] - mark;
                    if (dext > 0) { d += dext; Ci[pn++] = e; h += (uint64_t)e; }
                    else { Cp[e] = flip(k); w[e] = 0; }
                }
            }
            elen[i] = pn - p1 + 1;
            idx p3 = pn, p4 = p1 + len[i];
            for (idx q = p2 + 1; q < p4; q++) {
                idx j = Ci[q];
                idx nvj = nv[j];
                if (nvj <= 0) continue;
                d += nvj;
                Ci[pn++] = j;
                h += (uint64_t)j;
            }
            if (d == 0) {
                Cp[i] = flip(k);
                idx nvi = -nv[i];
                dk -= nvi; nvk += nvi; nel += nvi;
                nv[i] = 0; elen[i] = -1;
            } else {
                degree[i] = imin(degree[i], d);
                Ci[pn] = Ci[p3];
                Ci[p3] = Ci[p1];
                Ci[p1] = k;
                len[i] = pn - p1 + 1;
                h %= (uint64_t)n;                                    /* :1617 */
                next[i] = hhead[h];
                hhead[h] = i;
                last[i] = (idx)h;
            }
        }
        degree[k] = dk;
        lemax = imax(lemax, dk);
        mark = wclear(mark + lemax, lemax, w, n);

        /* supernode detection, :1628-1679 */
        for (idx pk = pk1; pk < pk2; pk++) {
            idx i = Ci[pk];
            if (nv[i] >= 0) continue;
            idx h = last[i];
            i = hhead[h];
            hhead[h] = -1;
            while (i != -1 && next[i] != -1) {
                idx ln = len[i], eln = elen[i];
                for (idx q = Cp[i] + 1; q <= Cp[i] + ln - 1; q++) w[Ci[q]] = mark;
                idx jlast = i;
                idx j = next[i];
                while (j != -1) {
                    int ok = (len[j] == ln) && (elen[j] == eln);
                    for (idx q = Cp[j] + 1; ok && q < Cp[j] + ln; q++)
                        if (w[Ci[q]] != mark) ok = 0;
                    if (ok) {
                        Cp[j] = flip(i);
                        nv[i] += nv[j];
                        nv[j] = 0;
                        elen[j] = -1;
                        j = next[j];
                        next[jlast] = j;
                    } else {
                        jlast = j;
                        j = next[j];
                    }
                }
                i = next[i];
                mark++;
            }
        }

        /* finalize new element, :1682-1714 */
        p = pk1;
        for (idx pk = pk1; pk < pk2; pk++) {
            idx i = Ci[pk];
            idx nvi = -nv[i];
            if (nvi <= 0) continue;
            nv[i] = nvi;
            idx d = degree[i] + dk - nvi;
            d = imin(d, n - nel - nvi);
            if (head[d] != -1) last[head[d]] = i;
            next[i] = head[d];
            last[i] = -1;
            head[d] = i;
            mindeg = imin(mindeg, d);
            degree[i] = d;
            Ci[p++] = i;
        }
        nv[k] = nvk;
        len[k] = p - pk1;
        if (len[k] == 0) { Cp[k] = -1; w[k] = 0; }
        if (elenk != 0) cnz = p;
    }

    /* postorder the assembly tree, :1718-1749 */
    for (idx i = 0; i < n; i++) Cp[i] = flip(Cp[i]);
    for (idx j = 0; j <= n; j++) head[j] = -1;
    for (idx j = n; j >= 0; j--) {
        if (nv[j] > 0) continue;
        next[j] = head[Cp[j]];
        head[Cp[j]] = j;
    }
    for (idx e = n; e >= 0; e--) {
        if (nv[e] <= 0) continue;
        if (Cp[e] != -1) { next[e] = head[Cp[e]]; head[Cp[e]] = e; }
    }
    idx k = 0;
    for (idx i = 0; i <= n; i++)
        if (Cp[i] == -1) k = tdfs(i, k, head, next, P, stack);

    free(stack); free(len); spm_free(C);
    return P;
}

/* lib.rs:2196-2209 */
static idx *pinvert(const idx *p, idx n)
{
    if (!p) return NULL;
    idx *pinv = (idx *)xcalloc(n, sizeof(idx));
    for (idx k = 0; k < n; k++) pinv[p[k]] = k;
    return pinv;
}

/* lib.rs:2369-2408 */
static spm *symperm(const spm *a, const idx *pinv)
{
    idx n = a->n;
    spm *c = spm_new(n, n, a->p[n], 1);
    idx *w = (idx *)xcalloc(n, sizeof(idx));
    for (idx j = 0; j < n; j++) {
        idx j2 = pinv ? pinv[j] : j;
        for (idx p = a->p[j]; p < a->p[j + 1]; p++) {
            idx i = a->i[p];
            if (i > j) continue;
            idx i2 = pinv ? pinv[i] : i;
            w[imax(i2, j2)]++;
        }
    }
    cumsum(c->p, w, n);
    for (idx j = 0; j < n; j++) {
        idx j2 = pinv ? pinv[j] : j;
        for (idx p = a->p[j]; p < a->p[j + 1]; p++) {
            idx i = a->i[p];
            if (i > j) continue;
            idx i2 = pinv ? pinv[i] : i;
            idx q = w[imax(i2, j2)]++;
            c->i[q] = imin(i2, j2);
            c->x[q] = a->x ? a->x[p] : 0.0;
        }
    }
    free(w);
    return c;
}

/* lib.rs:2026-2071 */
static idx *etree(const spm *a, int ata)
{
    idx n = a->n, m = a->m;
    idx *parent = (idx *)xcalloc(n, sizeof(idx));
    idx *ancestor = (idx *)xcalloc(n + (ata ? m : 0), sizeof(idx));
    idx *prev = ancestor + n;
    if (ata) for (idx i = 0; i < m; i++) prev[i] = -1;
    for (idx k = 0; k < n; k++) {
        parent[k] = -1;
        ancestor[k] = -1;
        for (idx p = a->p[k]; p < a->p[k + 1]; p++) {
            idx i = ata ? prev[a->i[p]] : a->i[p];
            while (i != -1 && i < k) {
                idx inext = ancestor[i];
                ancestor[i] = k;
                if (inext == -1) parent[i] = k;
                i = inext;
            }
            if (ata) prev[a->i[p]] = k;
        }
    }
    free(ancestor);
    return parent;
}

/* lib.rs:2213-2240 */
static idx *postorder(idx n, const idx *parent)
{
    idx *post = (idx *)xcalloc(n, sizeof(idx));
    idx *head = (idx *)xcalloc(3 * n, sizeof(idx));
    idx *next = head + n, *stack = head + 2 * n;
    for (idx j = 0; j < n; j++) head[j] = -1;
    for (idx j = n - 1; j >= 0; j--) {
        if (parent[j] == -1) continue;
        next[j] = head[parent[j]];
        head[parent[j]] = j;
    }
    idx k = 0;
    for (idx j = 0; j < n; j++)
        if (parent[j] == -1) k = tdfs(j, k, head, next, post, stack);
    free(head);
    return post;
}

/* lib.rs:1756-1793 */
static void cedge(idx j, idx i, const idx *first, idx *maxfirst, idx *delta, idx *prevleaf, idx *ancestor)
{
    if (i <= j || first[j] <= maxfirst[i]) return;
    maxfirst[i] = first[j];
    idx jprev = prevleaf[i];
    delta[j]++;
    if (jprev != -1) {
        idx q = jprev;
        while (q != ancestor[q]) q = ancestor[q];
        for (idx s = jprev; s != q;) {
            idx sparent = ancestor[s];
            ancestor[s] = q;
            s = sparent;
        }
        delta[q]--;
    }
    prevleaf[i] = j;
}

/* lib.rs:1797-1897 */
static idx *counts(const spm *a, const idx *parent, const idx *post, int ata)
{
    idx m = a->m, n = a->n;
    idx s = ata ? 4 * n + n + m + 1 : 4 * n;
    idx *w = (idx *)xcalloc(s, sizeof(idx));
    idx *ancestor = w, *maxfirst = w + n, *prevleaf = w + 2 * n, *first = w + 3 * n;
    idx *head = ata ? w + 4 * n : NULL, *next = ata ? w + 5 * n + 1 : NULL;
    idx *delta = (idx *)xcalloc(n, sizeof(idx));
    spm *at = transpose(a, 0);
    for (idx k = 0; k < s; k++) w[k] = -1;
    for (idx k = 0; k < n; k++) {
        idx j = post[k];
        delta[j] = (first[j] == -1) ? 1 : 0;
        for (; j != -1 && first[j] == -1; j = parent[j]) first[j] = k;
    }
    if (ata) {
        for (idx k = 0; k < n; k++) w[post[k]] = k;   /* invert post (aliases ancestor, as the reference) */
        for (idx i = 0; i < m; i++) {
            idx k = n;
            for (idx p = at->p[i]; p < at->p[i + 1]; p++) k = imin(k, w[at->i[p]]);
            next[i] = head[k];
            head[k] = i;
        }
    }
    for (idx i = 0; i < n; i++) ancestor[i] = i;
    for (idx k = 0; k < n; k++) {
        idx j = post[k];
        if (parent[j] != -1) delta[parent[j]]--;
        if (ata) {
            for (idx ii = head[k]; ii != -1; ii = next[ii])
                for (idx p = at->p[ii]; p < at->p[ii + 1]; p++)
                    cedge(j, at->i[p], first, maxfirst, delta, prevleaf, ancestor);
        } else {
            for (idx p = at->p[j]; p < at->p[j + 1]; p++)
                cedge(j, at->i[p], first, maxfirst, delta, prevleaf, ancestor);
        }
        if (parent[j] != -1) ancestor[j] = parent[j];
    }
    for (idx j = 0; j < n; j++)
        if (parent[j] != -1) delta[parent[j]] += delta[j];
    spm_free(at);
    free(w);
    return delta;
}

/* symbolic object, data.rs:1022-1038 */
typedef struct {
    idx *pinv, *q, *parent, *cp;
    idx m2, lnz, unz;
    idx pinv_len;
} symb;

static void symb_free(symb *s)
{
    if (!s) return;
    free(s->pinv); free(s->q); free(s->parent); free(s->cp); free(s);
}

/* lib.rs:968-986 */
static symb *schol(const spm *a, int order)
{
    idx n = a->n;
    symb *s = (symb *)calloc(1, sizeof(symb));
    idx *p = amd(a, order);
    s->pinv = pinvert(p, n);
    s->pinv_len = s->pinv ? n : 0;
    free(p);
    spm *c = symperm(a, s->pinv);
    s->parent = etree(c, 0);
    idx *post = postorder(n, s->parent);
    idx *cnt = counts(c, s->parent, post, 0);
    free(post);
    spm_free(c);
    s->cp = (idx *)xcalloc(n + 1, sizeof(idx));
    s->unz = cumsum(s->cp, cnt, n);
    s->lnz = s->unz;
    free(cnt);
    return s;
}

/* lib.rs:1985-2022 */
static idx ereach(const spm *a, idx k, const idx *parent, idx *s, idx *w, double *x, idx top)
{
    for (idx p = a->p[k]; p < a->p[k + 1]; p++) {
        idx i = a->i[p];
        if (i > k) continue;
        x[i] = a->x[p];
        idx len = 0;
        for (; w[i] != k; i = parent[i]) { s[len++] = i; w[i] = k; }
        for (idx j = 1; j <= len; j++) s[--top] = s[len - j];
    }
    return top;
}

/* lib.rs:278-337.  L is written into caller storage Lp[n+1], Li[cp[n]], Lx[cp[n]]. */
static int chol_numeric(const spm *a, const idx *pinv, const idx *parent, const idx *cp, idx *Lp, idx *Li, double *Lx)
{
    idx n = a->n;
    idx *w = (idx *)xcalloc(3 * n, sizeof(idx));
    idx *s = w + n, *c = w + 2 * n;
    double *x = (double *)xcalloc(n, sizeof(double));
    spm *C = pinv ? symperm(a, pinv) : NULL;
    const spm *Cm = C ? C : a;
    int status = ORC_OK;
    for (idx k = 0; k < n; k++) {
        Lp[k] = c[k] = cp[k];
        x[k] = 0.0;
        w[k] = k;
        idx top = ereach(Cm, k, parent, s, w, x, n);
        double d = x[k];
        x[k] = 0.0;
        for (; top < n; top++) {
            idx i = s[top];
            double lki = x[i] / Lx[Lp[i]];
            x[i] = 0.0;
            for (idx p = Lp[i] + 1; p < c[i]; p++) x[Li[p]] -= Lx[p] * lki;
            d -= lki * lki;
            idx p = c[i]++;
            Li[p] = k;
            Lx[p] = lki;
        }
        if (d <= 0.0) { status = ORC_NOT_POSITIVE_DEFINITE; break; }   /* :325 */
        idx p = c[k]++;
        Li[p] = k;
        Lx[p] = sqrt(d);                                               /* :332 powf(d, 0.5) */
    }
    if (status == ORC_OK) Lp[n] = cp[n];
    spm_free(C);
    free(w); free(x);
    return status;
}

/* lib.rs:464-471 */
void orc_lsolve(idx n, const idx *Lp, const idx *Li, const double *Lx, double *x)
{
    for (idx j = 0; j < n; j++) {
        x[j] /= Lx[Lp[j]];
        for (idx p = Lp[j] + 1; p < Lp[j + 1]; p++) x[Li[p]] -= Lx[p] * x[j];
    }
}
/* lib.rs:505-512 */
void orc_ltsolve(idx n, const idx *Lp, const idx *Li, const double *Lx, double *x)
{
    for (idx j = n - 1; j >= 0; j--) {
        for (idx p = Lp[j] + 1; p < Lp[j + 1]; p++) x[j] -= Lx[p] * x[Li[p]];
        x[j] /= Lx[Lp[j]];
    }
}
/* lib.rs:1230-1237 */
void orc_usolve(idx n, const idx *Up, const idx *Ui, const double *Ux, double *x)
{
    for (idx j = n - 1; j >= 0; j--) {
        x[j] /= Ux[Up[j + 1] - 1];
        for (idx p = Up[j]; p < Up[j + 1] - 1; p++) x[Ui[p]] -= Ux[p] * x[j];
    }
}
/* lib.rs:1271-1278 */
void orc_utsolve(idx n, const idx *Up, const idx *Ui, const double *Ux, double *x)
{
    for (idx j = 0; j < n; j++) {
        for (idx p = Up[j]; p < Up[j + 1] - 1; p++) x[j] -= Ux[p] * x[Ui[p]];
        x[j] /= Ux[Up[j + 1] - 1];
    }
}
/* lib.rs:411-421: r = A x + y */
void orc_gaxpy(idx m, idx n, const idx *Ap, const idx *Ai, const double *Ax, const double *x, const double *y, double *r)
{
    for (idx i = 0; i < m; i++) r[i] = y[i];
    for (idx j = 0; j < n; j++)
        for (idx p = Ap[j]; p < Ap[j + 1]; p++) r[Ai[p]] += Ax[p] * x[j];
}
/* lib.rs:2151-2159 */
static void ipvec(idx n, const idx *p, const double *b, double *x)
{
    for (idx k = 0; k < n; k++) x[p ? p[k] : k] = b[k];
}
/* lib.rs:2244-2251 */
static void pvec(idx n, const idx *p, const double *b, double *x)
{
    for (idx k = 0; k < n; k++) x[k] = b[p ? p[k] : k];
}

/* ---- LU ------------------------------------------------------------------------ */

/* lib.rs:1916-1975; Gp = L.p is used for the visit marks (flipped), as in the reference */
static idx dfs(idx j, idx *Gp, const idx *Gi, idx top, idx *xi, idx *pstack, const idx *pinv)
{
    idx head = 0;
    xi[0] = j;
    while (head >= 0) {
        j = xi[head];
        idx jnew = pinv ? pinv[j] : j;
        if (Gp[j] >= 0) {                                 /* !marked */
            Gp[j] = flip(Gp[j]);
            pstack[head] = (jnew < 0) ? 0 : unflip(Gp[jnew]);
        }
        int done = 1;
        idx p2 = (jnew < 0) ? 0 : unflip(Gp[jnew + 1]);
        for (idx p = pstack[head]; p < p2; p++) {
            idx i = Gi[p];
            if (Gp[i] < 0) continue;
            pstack[head] = p;
            xi[++head] = i;
            done = 0;
            break;
        }
        if (done) { head--; xi[--top] = j; }
    }
    return top;
}

/* lib.rs:2256-2277 */
static idx reach(idx n, idx *Gp, const idx *Gi, const spm *b, idx k, idx *xi, const idx *pinv)
{
    idx top = n;
    for (idx p = b->p[k]; p < b->p[k + 1]; p++)
        if (Gp[b->i[p]] >= 0) top = dfs(b->i[p], Gp, Gi, top, xi, xi + n, pinv);
    for (idx p = top; p < n; p++) Gp[xi[p]] = flip(Gp[xi[p]]);
    return top;
}

/* lib.rs:2333-2365 */
static idx splsolve(spm *L, const spm *b, idx k, idx *xi, double *x, const idx *pinv)
{
    idx n = L->n;
    idx top = reach(n, L->p, L->i, b, k, xi, pinv);
    for (idx p = top; p < n; p++) x[xi[p]] = 0.0;
    for (idx p = b->p[k]; p < b->p[k + 1]; p++) x[b->i[p]] = b->x[p];
    for (idx px = top; px < n; px++) {
        idx j = xi[px];
        idx jnew = pinv ? pinv[j] : j;
        if (jnew < 0) continue;
        for (idx p = L->p[jnew] + 1; p < L->p[jnew + 1]; p++) x[L->i[p]] -= L->x[p] * x[j];
    }
    return top;
}

typedef struct {
    spm *L, *U;
    idx *pinv;
    idx n;
} lufact;

static void lufact_free(lufact *f)
{
    if (!f) return;
    spm_free(f->L); spm_free(f->U); free(f->pinv); free(f);
}

/* lib.rs:519-622; q may be NULL (natural); lnz/unz initial guesses as sqr gives them (:1135-1136) */
static lufact *lu_numeric(const spm *a, const idx *q, double tol, int *status)
{
    idx n = a->n;
    idx guess = 4 * a->p[n] + n;
    lufact *f = (lufact *)calloc(1, sizeof(lufact));
    f->n = n;
    f->L = spm_new(n, n, guess, 1);
    f->U = spm_new(n, n, guess, 1);
    f->pinv = (idx *)xcalloc(n, sizeof(idx));
    spm *L = f->L, *U = f->U;
    idx *pinv = f->pinv;
    double *x = (double *)xcalloc(n, sizeof(double));
    idx *xi = (idx *)xcalloc(2 * n, sizeof(idx));
    for (idx i = 0; i < n; i++) pinv[i] = -1;
    for (idx k = 0; k <= n; k++) L->p[k] = 0;
    idx lnz = 0, unz = 0;
    *status = ORC_OK;
    for (idx k = 0; k < n; k++) {
        L->p[k] = lnz;
        U->p[k] = unz;
        if (lnz + n > L->nzmax) spm_resize(L, 2 * L->nzmax + n);
        if (unz + n > U->nzmax) spm_resize(U, 2 * U->nzmax + n);
        idx col = q ? q[k] : k;
        idx top = splsolve(L, a, col, xi, x, pinv);
        /* find pivot, :566-589 */
        idx ipiv = -1;
        double a_f = -1.0;
        for (idx p = top; p < n; p++) {
            idx i = xi[p];
            if (pinv[i] < 0) {
                double t = fabs(x[i]);
                if (t > a_f) { a_f = t; ipiv = i; }
            } else {
                U->i[unz] = pinv[i];
                U->x[unz++] = x[i];
            }
        }
        if (ipiv == -1 || a_f <= 0.0) { *status = ORC_NO_PIVOT; break; }
        if (pinv[col] < 0 && fabs(x[col]) >= a_f * tol) ipiv = col;
        /* divide by pivot, :592-609 */
        double pivot = x[ipiv];
        U->i[unz] = k;
        U->x[unz++] = pivot;
        pinv[ipiv] = k;
        L->i[lnz] = ipiv;
        L->x[lnz++] = 1.0;
        for (idx p = top; p < n; p++) {
            idx i = xi[p];
            if (pinv[i] < 0) {
                L->i[lnz] = i;
                L->x[lnz++] = x[i] / pivot;
            }
            x[i] = 0.0;
        }
    }
    if (*status == ORC_OK) {
        L->p[n] = lnz;
        U->p[n] = unz;
        for (idx p = 0; p < lnz; p++) L->i[p] = pinv[L->i[p]];   /* :615-617 */
        spm_resize(L, lnz);                                    /* quick_trim, data.rs:391 */
        spm_resize(U, unz);
    }
    free(x); free(xi);
    return f;
}

/* ---- exported drivers ------------------------------------------------------------ */

/* amd alone; returns 0 if the ordering is natural (None), 1 if P[0..n] was written */
int orc_amd(idx m, idx n, const idx *Ap, const idx *Ai, int order, idx *P)
{
    spm a = view(m, n, Ap, Ai, NULL);
    idx *p = amd(&a, order);
    if (!p) return 0;
    memcpy(P, p, sizeof(idx) * (size_t)(n + 1));
    free(p);
    return 1;
}

/* schol, lib.rs:968-986.  Returns 1 if pinv was written (0 = natural order). */
int orc_schol(idx n, const idx *Ap, const idx *Ai, int order, idx *pinv, idx *parent, idx *cp)
{
    spm a = view(n, n, Ap, Ai, NULL);
    symb *s = schol(&a, order);
    int has = s->pinv != NULL;
    if (has) memcpy(pinv, s->pinv, sizeof(idx) * (size_t)n);
    memcpy(parent, s->parent, sizeof(idx) * (size_t)n);
    memcpy(cp, s->cp, sizeof(idx) * (size_t)(n + 1));
    symb_free(s);
    return has;
}

/* chol, lib.rs:278-337; pinv may be NULL */
int orc_chol(idx n, const idx *Ap, const idx *Ai, const double *Ax, const idx *pinv, const idx *parent, const idx *cp,
             idx *Lp, idx *Li, double *Lx)
{
    spm a = view(n, n, Ap, Ai, Ax);
    return chol_numeric(&a, pinv, parent, cp, Lp, Li, Lx);
}

/* cholsol, lib.rs:377-389 */
int orc_cholsol(idx n, const idx *Ap, const idx *Ai, const double *Ax, double *b, int order)
{
    spm a = view(n, n, Ap, Ai, Ax);
    symb *s = schol(&a, order);
    idx lnz = s->cp[n];
    idx *Lp = (idx *)xcalloc(n + 1, sizeof(idx)), *Li = (idx *)xcalloc(lnz, sizeof(idx));
    double *Lx = (double *)xcalloc(lnz, sizeof(double)), *x = (double *)xcalloc(n, sizeof(double));
    int st = chol_numeric(&a, s->pinv, s->parent, s->cp, Lp, Li, Lx);
    if (st == ORC_OK) {
        ipvec(n, s->pinv, b, x);
        orc_lsolve(n, Lp, Li, Lx, x);
        orc_ltsolve(n, Lp, Li, Lx, x);
        pvec(n, s->pinv, x, b);
    }
    free(Lp); free(Li); free(Lx); free(x);
    symb_free(s);
    return st;
}

/* lu with the ordering of sqr(a, order, false), lib.rs:1114-1119,:1134-1137 ; handle API because nnz is unknown */
void *orc_lu_new(idx n, const idx *Ap, const idx *Ai, const double *Ax, int order, double tol, int *status, idx *q_out)
{
    spm a = view(n, n, Ap, Ai, Ax);
    idx *q = amd(&a, order);
    if (q_out) {
        if (q) memcpy(q_out, q, sizeof(idx) * (size_t)n);
        else for (idx k = 0; k < n; k++) q_out[k] = k;
    }
    lufact *f = lu_numeric(&a, q, tol, status);
    free(q);
    return f;
}
void orc_lu_nnz(void *h, idx *lnz, idx *unz)
{
    lufact *f = (lufact *)h;
    *lnz = f->L->p[f->n];
    *unz = f->U->p[f->n];
}
void orc_lu_get(void *h, idx *Lp, idx *Li, double *Lx, idx *Up, idx *Ui, double *Ux, idx *pinv)
{
    lufact *f = (lufact *)h;
    idx n = f->n;
    memcpy(Lp, f->L->p, sizeof(idx) * (size_t)(n + 1));
    memcpy(Up, f->U->p, sizeof(idx) * (size_t)(n + 1));
    memcpy(Li, f->L->i, sizeof(idx) * (size_t)Lp[n]);
    memcpy(Lx, f->L->x, sizeof(double) * (size_t)Lp[n]);
    memcpy(Ui, f->U->i, sizeof(idx) * (size_t)Up[n]);
    memcpy(Ux, f->U->x, sizeof(double) * (size_t)Up[n]);
    memcpy(pinv, f->pinv, sizeof(idx) * (size_t)n);
}
void orc_lu_free(void *h) { lufact_free((lufact *)h); }

/* lusol, lib.rs:672-683 */
int orc_lusol(idx n, const idx *Ap, const idx *Ai, const double *Ax, double *b, int order, double tol)
{
    spm a = view(n, n, Ap, Ai, Ax);
    idx *q = amd(&a, order);
    int st;
    lufact *f = lu_numeric(&a, q, tol, &st);
    if (st == ORC_OK) {
        double *x = (double *)xcalloc(n, sizeof(double));
        ipvec(n, f->pinv, b, x);
        orc_lsolve(n, f->L->p, f->L->i, f->L->x, x);
        orc_usolve(n, f->U->p, f->U->i, f->U->x, x);
        ipvec(n, q, x, b);
        free(x);
    }
    lufact_free(f);
    free(q);
    return st;
}

/* ---- QR symbolic + numeric: present ONLY to run the reference's exact-integer pins
 *      (tests/basic_tests.rs:1272-1336), which exercise amd / etree / post / counts ------ */

/* lib.rs:2163-2192 */
static spm *permute(const spm *a, const idx *pinv, const idx *q)
{
    idx nz = 0;
    spm *c = spm_new(a->m, a->n, a->p[a->n], 1);
    for (idx k = 0; k < a->n; k++) {
        c->p[k] = nz;
        idx j = q ? q[k] : k;
        for (idx p = a->p[j]; p < a->p[j + 1]; p++) {
            c->x[nz] = a->x[p];
            c->i[nz++] = pinv ? pinv[a->i[p]] : a->i[p];
        }
    }
    c->p[a->n] = nz;
    return c;
}

/* lib.rs:2450-2530 */
static idx *vcount(const spm *a, const idx *parent, idx *m2, idx *vnz)
{
    idx n = a->n, m = a->m;
    idx *pinv = (idx *)xcalloc(2 * m + n, sizeof(idx));
    idx *leftmost = pinv + m + n;
    idx *w = (idx *)xcalloc(m + 3 * n, sizeof(idx));
    idx *next = w, *head = w + m, *tail = w + m + n, *nque = w + m + 2 * n;
    for (idx k = 0; k < n; k++) { head[k] = -1; tail[k] = -1; nque[k] = 0; }
    for (idx i = 0; i < m; i++) leftmost[i] = -1;
    for (idx k = n - 1; k >= 0; k--)
        for (idx p = a->p[k]; p < a->p[k + 1]; p++) leftmost[a->i[p]] = k;
    for (idx i = m - 1; i >= 0; i--) {
        pinv[i] = -1;
        idx k = leftmost[i];
        if (k == -1) continue;
        if (nque[k]++ == 0) tail[k] = i;
        next[i] = head[k];
        head[k] = i;
    }
    *vnz = 0;
    *m2 = m;
    for (idx k = 0; k < n; k++) {
        idx i = head[k];
        (*vnz)++;
        if (i < 0) i = (*m2)++;
        pinv[i] = k;
        if (--nque[k] <= 0) continue;
        *vnz += nque[k];
        idx pa = parent[k];
        if (pa != -1) {
            if (nque[pa] == 0) tail[pa] = tail[k];
            next[tail[k]] = head[pa];
            head[pa] = next[i];
            nque[pa] += nque[k];
        }
    }
    idx k = n;
    for (idx i = 0; i < m; i++)
        if (pinv[i] < 0) pinv[i] = k++;
    free(w);
    return pinv;
}

/* sqr(a, order, qr=true), lib.rs:1114-1133 */
static symb *sqr_qr(const spm *a, int order)
{
    symb *s = (symb *)calloc(1, sizeof(symb));
    s->q = amd(a, order);
    spm *c = (order >= 0) ? permute(a, NULL, s->q) : NULL;
    const spm *cm = c ? c : a;
    s->parent = etree(cm, 1);
    idx *post = postorder(a->n, s->parent);
    s->cp = counts(cm, s->parent, post, 1);
    free(post);
    s->pinv = vcount(cm, s->parent, &s->m2, &s->lnz);
    s->pinv_len = 2 * a->m + a->n;
    s->unz = 0;
    for (idx k = 0; k < a->n; k++) s->unz += s->cp[k];
    spm_free(c);
    return s;
}

/* lib.rs:2099-2111 */
static void happly(const spm *v, idx i, double beta, double *x)
{
    double tau = 0.0;
    for (idx p = v->p[i]; p < v->p[i + 1]; p++) tau += v->x[p] * x[v->i[p]];
    tau *= beta;
    for (idx p = v->p[i]; p < v->p[i + 1]; p++) x[v->i[p]] -= v->x[p] * tau;
}

/* lib.rs:2116-2147 */
static double house(double *x, double *beta, idx n)
{
    double s, sigma = 0.0;
    for (idx i = 1; i < n; i++) sigma += x[i] * x[i];
    if (sigma != 0.0) {
        s = sqrt(x[0] * x[0] + sigma);
        x[0] = (x[0] <= 0.0) ? x[0] - s : -sigma / (x[0] + s);
        *beta = 1.0 / (-s * x[0]);
    } else {
        s = fabs(x[0]);
        *beta = (x[0] <= 0.0) ? 2.0 : 0.0;
        x[0] = 1.0;
    }
    return s;
}

/* sqr(order, qr=true) + qr, lib.rs:788-877.  Outputs sized by the caller:
 * q[n+1], parent[n], cp[n], pinv[2m+n], scal[3]={m2,lnz,unz}, Vp/Rp[n+1], Vi/Vx[lnz], Ri/Rx[unz], beta[n]. */
int orc_sqr_qr(idx m, idx n, const idx *Ap, const idx *Ai, const double *Ax, int order, idx *q, idx *parent, idx *cp,
               idx *pinv, idx *scal, idx *Vp, idx *Vi, double *Vx, idx *Rp, idx *Ri, double *Rx, double *beta)
{
    spm a = view(m, n, Ap, Ai, Ax);
    symb *s = sqr_qr(&a, order);
    int has_q = s->q != NULL;
    if (has_q) memcpy(q, s->q, sizeof(idx) * (size_t)(n + 1));
    memcpy(parent, s->parent, sizeof(idx) * (size_t)n);
    memcpy(cp, s->cp, sizeof(idx) * (size_t)n);
    memcpy(pinv, s->pinv, sizeof(idx) * (size_t)(2 * m + n));
    scal[0] = s->m2; scal[1] = s->lnz; scal[2] = s->unz;
    if (Vp) {
        idx m2 = s->m2, vnz = 0, rnz = 0;
        const idx *leftmost = s->pinv + m + n;
        idx *w = (idx *)xcalloc(m2 + n, sizeof(idx));
        idx *ws = w + m2;
        double *x = (double *)xcalloc(m2, sizeof(double));
        spm V, R;
        V.m = R.m = m2; V.n = R.n = n; V.p = Vp; V.i = Vi; V.x = Vx; R.p = Rp; R.i = Ri; R.x = Rx;
        for (idx k = 0; k < n; k++) beta[k] = 0.0;
        for (idx i = 0; i < m2; i++) w[i] = -1;
        for (idx k = 0; k < n; k++) {
            Rp[k] = rnz;
            Vp[k] = vnz;
            idx p1 = vnz;
            w[k] = k;
            Vi[vnz++] = k;
            idx top = n;
            idx col = s->q ? s->q[k] : k;
            for (idx p = Ap[col]; p < Ap[col + 1]; p++) {
                idx i = leftmost[Ai[p]];
                idx len = 0;
                for (; w[i] != k; i = s->parent[i]) { ws[len++] = i; w[i] = k; }
                for (idx j = 1; j <= len; j++) ws[--top] = ws[len - j];
                i = s->pinv[Ai[p]];
                x[i] = Ax[p];
                if (i > k && w[i] < k) { Vi[vnz++] = i; w[i] = k; }
            }
            for (idx p = top; p < n; p++) {
                idx i = ws[p];
                happly(&V, i, beta[i], x);
                Ri[rnz] = i;
                Rx[rnz++] = x[i];
                x[i] = 0.0;
                if (s->parent[i] == k) {                      /* scatter_no_x, lib.rs:2310-2329 */
                    for (idx pp = Vp[i]; pp < Vp[i + 1]; pp++) {
                        idx r = Vi[pp];
                        if (w[r] < k) { w[r] = k; Vi[vnz++] = r; }
                    }
                }
            }
            for (idx p = p1; p < vnz; p++) { Vx[p] = x[Vi[p]]; x[Vi[p]] = 0.0; }
            Ri[rnz] = k;
            Rx[rnz++] = house(Vx + p1, beta + k, vnz - p1);
            Vp[k + 1] = vnz;   /* so happly on column k sees its extent (the reference finalises v.p at the end; */
            Rp[k + 1] = rnz;   /*  v.p[k+1] is written as v.p[k] of the next iteration before it is read) */
        }
        free(w); free(x);
    }
    symb_free(s);
    return has_q;
}

/* ---- timing helper for bench.py's cpu_baseline leg: nrep x (lsolve + ltsolve) on one RHS each ---- */
void orc_solve_llt_many(idx n, const idx *Lp, const idx *Li, const double *Lx, double *X, idx ldx, idx nrhs)
{
    for (idx r = 0; r < nrhs; r++) {
        orc_lsolve(n, Lp, Li, Lx, X + r * ldx);
        orc_ltsolve(n, Lp, Li, Lx, X + r * ldx);
    }
}
