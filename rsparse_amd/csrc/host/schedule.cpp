// Turns a CSC triangular factor into the "gather" form the kernels consume, and builds the level
// schedule (integer-only, host).  No counterpart in the reference, whose solves are bare sequential
// loops (/root/reference/src/lib.rs:464-471, 505-512, 1230-1237, 1271-1278); the dependency structure
// encoded here is exactly the order those loops impose.
#include <algorithm>
#include <limits>
#include <thread>

#include "parallel.hpp"
#include "sparse.hpp"

namespace rsp {

uint64_t pattern_hash(const CscView &A)
{
    auto mix = [](uint64_t z) {
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    };
    // fixed chunks of 2^16 words, each hashed sequentially (position-dependent), chunk hashes combined in chunk order
    auto hash_array = [&](const i64 *v, i64 count, uint64_t seed) {
        constexpr i64 kChunk = 1 << 16;
        const i64 nchunks = (count + kChunk - 1) / kChunk;
        std::vector<uint64_t> part((size_t)nchunks, 0);
        parallel_ranges(nchunks, 8, [&](i64 lo, i64 hi) {
            for (i64 c = lo; c < hi; ++c) {
                uint64_t h = seed + (uint64_t)c * 0x9e3779b97f4a7c15ull;
                const i64 e = std::min(count, (c + 1) * kChunk);
                for (i64 k = c * kChunk; k < e; ++k) h = (h ^ (uint64_t)v[k]) * 0x100000001b3ull + (h >> 29);
                part[(size_t)c] = mix(h);
            }
        });
        uint64_t h = seed;
        for (uint64_t x : part) h = mix(h ^ x) + 0x9e3779b97f4a7c15ull;
        return h;
    };
    uint64_t h = mix((uint64_t)A.m * 0x9e3779b97f4a7c15ull ^ (uint64_t)A.n);
    h = hash_array(A.p, A.n + 1, h);
    h = hash_array(A.i, A.p[A.n], h ^ 0x5851f42d4c957f2dull);
    return h;
}


// The columns of T are cut into one contiguous piece of the VISITING order per host thread (ascending j for lsolve /
// utsolve, descending for ltsolve / usolve -- the order in which the reference's loop meets the entries).  Thread k
// counts its entries per unknown, the counts become (unknown, thread) offsets, and every thread then writes its entries
// behind those of the threads that visit earlier columns: every unknown keeps its terms in the reference's order, and
// the result does not depend on the number of threads.
int build_gather_form(const CscView &T, int kind, GatherForm &G)
{
    const i64 n = T.n;
    if (T.m != n || kind < 0 || kind > 3) return kBadInput;
    if (n >= std::numeric_limits<int32_t>::max() || T.nnz() >= std::numeric_limits<int32_t>::max()) return kBadInput;
    const bool lower_storage = (kind == 0 || kind == 1);  // L: diagonal first; U: diagonal last
    const bool by_rows = (kind == 0 || kind == 2);        // lsolve/usolve gather along ROWS of the stored factor
    const bool pattern_only = T.x == nullptr;
    G.n = n;
    G.descending = (kind == 1 || kind == 2);
    G.diag.assign(n, 0.0);
    G.ptr.assign(n + 1, 0);
    const i64 threads = std::max<i64>(1, std::min<i64>(host_threads(), T.nnz() >> 18));
    auto piece = [&](i64 k, i64 &lo, i64 &hi) {  // positions [lo, hi) of the visiting order
        lo = n * k / threads;
        hi = n * (k + 1) / threads;
    };
    auto run = [&](auto &&body) {
        std::vector<std::thread> pool;
        for (i64 k = 1; k < threads; ++k) pool.emplace_back(body, k);
        body((i64)0);
        for (std::thread &th : pool) th.join();
    };
    // validate + diagonal + entry counts per (unknown, thread)
    std::vector<std::vector<int32_t>> count((size_t)threads);
    std::vector<char> bad((size_t)threads, 0);
    run([&](i64 k) {
        if (by_rows && threads > 1) count[k].assign(n, 0);
        i64 lo, hi;
        piece(k, lo, hi);
        for (i64 t = lo; t < hi; ++t) {
            const i64 j = G.descending ? n - 1 - t : t;
            const i64 b = T.p[j], e = T.p[j + 1];
            if (e <= b || b < 0 || e > T.nnz()) {  // the reference indexes the diagonal unconditionally
                bad[k] = 1;
                return;
            }
            const i64 dpos = lower_storage ? b : e - 1;
            if (!pattern_only) G.diag[j] = T.x[dpos];
            const i64 first = lower_storage ? b + 1 : b, last = lower_storage ? e : e - 1;
            for (i64 p = first; p < last; ++p) {
                const i64 i = T.i[p];
                // off-diagonal entries must lie strictly inside the triangle, otherwise the loops of the reference
                // do not describe a triangular solve (it would silently compute something else)
                if (lower_storage ? !(i > j && i < n) : !(i >= 0 && i < j)) {
                    bad[k] = 1;
                    return;
                }
                if (by_rows && threads > 1) count[k][i]++;
                else if (by_rows) G.ptr[i + 1]++;
            }
            if (!by_rows) G.ptr[j + 1] = last - first;
        }
    });
    for (char b : bad)
        if (b) return kBadInput;
    if (by_rows && threads > 1) {
        parallel_ranges(n, 1 << 16, [&](i64 lo, i64 hi) {
            for (i64 r = lo; r < hi; ++r) {
                i64 total = 0;
                for (i64 k = 0; k < threads; ++k) total += count[k][r];
                G.ptr[r + 1] = total;
            }
        });
    }
    for (i64 r = 0; r < n; ++r) G.ptr[r + 1] += G.ptr[r];
    const i64 nz = G.ptr[n];
    G.idx.resize(nz);
    G.val.assign(nz, 0.0);
    G.src.clear();
    G.placeholder = pattern_only;
    const bool want_src = pattern_only || G.keep_src;
    if (want_src) G.src.resize(nz);
    if (by_rows && threads > 1) {
        // count[k][r] becomes the first slot of thread k in unknown r (threads in visiting order)
        parallel_ranges(n, 1 << 16, [&](i64 lo, i64 hi) {
            for (i64 r = lo; r < hi; ++r) {
                i64 at = G.ptr[r];
                for (i64 k = 0; k < threads; ++k) {
                    const int32_t c = count[k][r];
                    count[k][r] = (int32_t)at;
                    at += c;
                }
            }
        });
    }
    std::vector<i64> fill;
    if (by_rows && threads == 1) fill.assign(G.ptr.begin(), G.ptr.end() - 1);
    // visit source columns in the order the reference's loop visits them, so each unknown accumulates its terms
    // in the reference's order: ascending j for lsolve/utsolve, descending j for ltsolve/usolve
    run([&](i64 k) {
        i64 lo, hi;
        piece(k, lo, hi);
        for (i64 t = lo; t < hi; ++t) {
            const i64 j = G.descending ? n - 1 - t : t;
            const i64 b = T.p[j], e = T.p[j + 1];
            i64 own = G.ptr[j];
            for (i64 p = (lower_storage ? b + 1 : b); p < (lower_storage ? e : e - 1); ++p) {
                const i64 i = T.i[p];
                const i64 q = !by_rows ? own++ : threads > 1 ? count[k][i]++ : fill[i]++;
                G.idx[q] = (int32_t)(by_rows ? j : i);
                if (want_src) G.src[q] = (int32_t)p;
                if (!pattern_only) G.val[q] = T.x[p];
            }
        }
    });
    return kOk;
}

void fill_gather_values(const CscView &T, int kind, GatherForm &G)
{
    const bool lower_storage = (kind == 0 || kind == 1);
    for (i64 j = 0; j < G.n; ++j) G.diag[j] = T.x[lower_storage ? T.p[j] : T.p[j + 1] - 1];
    const double *x = T.x;
    parallel_ranges((i64)G.src.size(), 1 << 18, [&](i64 lo, i64 hi) {
        for (i64 q = lo; q < hi; ++q) G.val[q] = x[G.src[q]];
    });
    if (!G.keep_src) std::vector<int32_t>().swap(G.src);
    G.placeholder = false;
}

Schedule build_schedule(const GatherForm &G, i64 max_task_entries)
{
    const i64 n = G.n;
    Schedule S;
    S.level_of_row.assign(n, 0);
    std::vector<int32_t> task_of(n, -1);
    std::vector<int32_t> task_level;  // level of each task, in creation order
    std::vector<int32_t> first, rows;
    i64 cur = -1, cur_entries = 0;
    for (i64 t = 0; t < n; ++t) {
        const i64 r = G.descending ? n - 1 - t : t;
        i64 row_level = 0;   // canonical level of the unknown
        i64 need = 0;        // lowest task level that satisfies the dependencies outside the current task
        bool chained = false;
        for (i64 p = G.ptr[r]; p < G.ptr[r + 1]; ++p) {
            const i64 d = G.idx[p];
            row_level = std::max(row_level, S.level_of_row[d] + 1);
            if (task_of[d] == cur) chained = true;
            else need = std::max<i64>(need, task_level[task_of[d]] + 1);
        }
        S.level_of_row[r] = row_level;
        S.num_row_levels = std::max(S.num_row_levels, row_level + 1);
        const i64 nent = G.ptr[r + 1] - G.ptr[r];
        const bool fits = cur >= 0 && cur_entries + nent <= max_task_entries;
        if (fits && need <= task_level[cur] && (chained || need == task_level[cur])) {
            rows[cur]++;
            cur_entries += nent;
        } else {
            if (cur >= 0 && chained) need = std::max<i64>(need, task_level[cur] + 1);
            cur = (i64)first.size();
            first.push_back((int32_t)r);
            rows.push_back(1);
            task_level.push_back((int32_t)need);
            cur_entries = nent;
        }
        task_of[r] = (int32_t)cur;
    }
    // bucket tasks by level (stable: creation order inside a level)
    i64 nlevels = 0;
    for (int32_t l : task_level) nlevels = std::max<i64>(nlevels, l + 1);
    S.level_ptr.assign(nlevels + 1, 0);
    for (int32_t l : task_level) S.level_ptr[l + 1]++;
    for (i64 l = 0; l < nlevels; ++l) S.level_ptr[l + 1] += S.level_ptr[l];
    std::vector<i64> fill(S.level_ptr.begin(), S.level_ptr.end() - 1);
    const i64 ntasks = (i64)first.size();
    S.task_first.resize(ntasks);
    S.task_rows.resize(ntasks);
    for (i64 k = 0; k < ntasks; ++k) {
        const i64 q = fill[task_level[k]]++;
        S.task_first[q] = first[k];
        S.task_rows[q] = rows[k];
        S.longest_task = std::max<i64>(S.longest_task, rows[k]);
    }
    return S;
}

}  // namespace rsp
