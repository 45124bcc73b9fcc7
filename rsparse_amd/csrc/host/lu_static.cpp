// Symbolic half of the GPU numeric LU: the patterns of L and U that rsparse::lu
// (/root/reference/src/lib.rs:519-622) produces WHEN EVERY COLUMN TAKES ITS DIAGONAL AS PIVOT (row q[k] for column k --
// what threshold pivoting does whenever |diagonal| >= tol * max|column|, lib.rs:585-589), in the reference's own
// storage order: U(:,k) = the already pivotal rows of the column's depth-first reach in reach order, diagonal last;
// L(:,k) = the pivot first, then the other rows of the reach in reach order (lib.rs:577-609), row indices of L in pivot
// order (lib.rs:615-617).  Integer work only: the same depth-first search as host/lu.cpp (lib.rs:2256-2277, 1916-1975;
// neighbours visited in storage order) with the pivot of column k known in advance, pinv[q[k]] = k.
// The device computes the values on these patterns (device/lu_engine.hip) and checks the assumption column by column;
// when it fails anywhere the caller falls back to sparse_lu.
//
// The search costs as many steps as the numeric factorisation costs multiply-subtracts (the reach of column k walks every
// entry of every column of L it touches) and is what a GPU factorisation spends most of its time in.  One column's search
// is sequential by nature, and the reference's ORDER of a reach (which fixes the order of every sum) only comes out of that
// search; but columns in different subtrees of the elimination tree of B + B' (B = A in pivot coordinates) never meet, so
// the columns are searched on the host threads in the order of that tree's levels, a column starting when its children
// are complete (a counter per column) -- the scheme of lu_columns and of the one-shot row kernel.  Two earlier attempts
// with the columns in ASCENDING order (each waiting for the patterns it met) were slower than one thread: consecutive
// columns are one chain.  Patterns are searched in 32-bit storage (1.7 times faster than the 64-bit search of host/lu.cpp).
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <memory>
#include <thread>

#include "parallel.hpp"
#include "sparse.hpp"

namespace rsp {
namespace {

// int32 blocks handed out to one thread; addresses stay put once handed out
class Arena {
   public:
    int32_t *take(size_t count)
    {
        if (count > left_) {
            const size_t size = std::max<size_t>(count, (size_t)1 << 20);
            blocks_.emplace_back(new int32_t[size]);
            at_ = blocks_.back().get();
            left_ = size;
        }
        int32_t *p = at_;
        at_ += count;
        left_ -= count;
        return p;
    }

   private:
    std::vector<std::unique_ptr<int32_t[]>> blocks_;
    int32_t *at_ = nullptr;
    size_t left_ = 0;
};

struct Column {
    const int32_t *l = nullptr;  // L(:,k): ORIGINAL row numbers, the pivot row first
    const int32_t *u = nullptr;  // U(:,k): column numbers (pivot order), the diagonal last
    int32_t llen = 0, ulen = 0;
};

// One thread's search state; column(k) is the reference's reach of column k with every pivot known in advance.
struct Searcher {
    const CscView &A;
    const i64 *q;
    const std::vector<int32_t> &col_of_row;
    Column *cols;
    std::vector<int32_t> &level;
    const std::atomic<uint8_t> *done;  // null: sequential (every earlier column is complete)
    std::atomic<bool> *failed;
    std::vector<int32_t> order, stack, resume, seen, dup;
    Arena arena;  // (the patterns live in it until they are assembled)

    Searcher(const CscView &A_, const i64 *q_, const std::vector<int32_t> &cr, Column *c, std::vector<int32_t> &lv, const std::atomic<uint8_t> *d,
             std::atomic<bool> *f)
        : A(A_), q(q_), col_of_row(cr), cols(c), level(lv), done(d), failed(f), order((size_t)A_.n), stack((size_t)A_.n), resume((size_t)A_.n),
          seen((size_t)A_.n, -1), dup((size_t)A_.n, -1)
    {
    }

    // an earlier column met by the search must be complete: by construction it is (a descendant in the elimination tree);
    // should that ever not hold, wait for it -- bounded -- rather than read half a pattern
    bool complete(int32_t jcol) const
    {
        if (!done || done[jcol].load(std::memory_order_acquire)) return true;
        for (i64 spins = 0; spins < ((i64)1 << 26); ++spins) {
            if (done[jcol].load(std::memory_order_acquire)) return true;
            if (failed->load(std::memory_order_relaxed)) return false;
            std::this_thread::yield();
        }
        return false;
    }

    bool column(i64 k)
    {
        const i64 n = A.n;
        const i64 col = q ? q[k] : k;
        // reach of A(:,col) through the columns of L that exist when column k is factored (columns < k): rows are ORIGINAL
        // row numbers, as in lu.cpp; row j is pivotal iff its column col_of_row[j] < k, and then that column of L continues
        // the search
        i64 top = n;
        for (i64 e = A.p[col]; e < A.p[col + 1]; ++e) {
            const int32_t root = (int32_t)A.i[e];
            if (seen[(size_t)root] == k) continue;  // (reached already, or a duplicate: checked below)
            i64 depth = 0;
            stack[0] = root;
            while (depth >= 0) {
                const int32_t j = stack[(size_t)depth];
                const int32_t jcol = col_of_row[(size_t)j];
                const int32_t *list = nullptr;
                int32_t end = 0;
                if (seen[(size_t)j] != k) {
                    seen[(size_t)j] = (int32_t)k;
                    resume[(size_t)depth] = 0;
                    if (jcol < k && !complete(jcol)) return false;
                }
                if (jcol < k) {
                    const Column &c = cols[(size_t)jcol];
                    list = c.l;
                    end = c.llen;
                }
                bool finished = true;
                for (int32_t p = resume[(size_t)depth]; p < end; ++p) {
                    const int32_t i = list[p];
                    if (seen[(size_t)i] == k) continue;
                    resume[(size_t)depth] = p;
                    stack[(size_t)++depth] = i;
                    finished = false;
                    break;
                }
                if (finished) {
                    --depth;
                    order[(size_t)--top] = j;
                }
            }
        }
        // duplicate row indices inside the column of A: the reference keeps the LAST value (x[i] = a.x[p] in order); the device
        // scatters a column's entries in parallel, so such matrices stay on the host
        bool good = true;
        for (i64 e = A.p[col]; e < A.p[col + 1] && good; ++e) {
            if (dup[(size_t)A.i[e]] == k) good = false;
            dup[(size_t)A.i[e]] = (int32_t)k;
        }
        i64 ulen = 1, llen = 1;
        bool diagonal_reached = false;
        for (i64 t = top; t < n; ++t) {
            const int32_t i = order[(size_t)t];
            if (col_of_row[(size_t)i] < k) ++ulen;
            else if (i == col) diagonal_reached = true;
            else ++llen;
        }
        if (!good || !diagonal_reached) return false;
        int32_t *u = arena.take((size_t)ulen), *l = arena.take((size_t)llen);
        i64 un = 0, ln = 0;
        int32_t lev = 0;
        l[ln++] = (int32_t)col;  // L(k,k) first
        for (i64 t = top; t < n; ++t) {
            const int32_t i = order[(size_t)t];
            const int32_t c = col_of_row[(size_t)i];
            if (c < k) {
                u[un++] = c;
                lev = std::max(lev, level[(size_t)c] + 1);
            } else if (i != col) {
                l[ln++] = i;
            }
        }
        u[un++] = (int32_t)k;  // U(k,k) last
        level[(size_t)k] = lev;
        Column &mine = cols[(size_t)k];
        mine.l = l;
        mine.u = u;
        mine.llen = (int32_t)llen;
        mine.ulen = (int32_t)ulen;
        return true;
    }
};

}  // namespace

// out.L / out.U: p and i final, x sized and zero; out.pinv; level[k] = 0 for a column that depends on no other, else
// 1 + the largest level among the columns in U(:,k).  false: some column's diagonal row is not in its reach (a
// structurally zero pivot), duplicate entries inside a column of A, or sizes beyond 32-bit device indices.
bool lu_static_pattern(const CscView &A, const i64 *q, LuFactors &out, std::vector<int32_t> &level)
{
    const i64 n = A.n;
    if (A.m != n || n >= INT32_MAX) return false;
    // the pivot row of every column is known: inverse of q
    std::vector<int32_t> col_of_row((size_t)n, -1);  // row r is the pivot of column col_of_row[r]
    for (i64 k = 0; k < n; ++k) {
        const i64 r = q ? q[k] : k;
        if (r < 0 || r >= n || col_of_row[(size_t)r] >= 0) return false;
        col_of_row[(size_t)r] = (int32_t)k;
    }
    for (i64 e = 0; e < A.nnz(); ++e)
        if (A.i[e] < 0 || A.i[e] >= n) return false;
    level.assign((size_t)n, 0);
    std::unique_ptr<Column[]> cols(new Column[(size_t)n]);
    std::atomic<bool> failed{false};
    const char *env = std::getenv("RSP_LU_SEARCH_THREADS");
    const i64 threads = std::max<i64>(1, std::min<i64>(env && *env ? std::atoll(env) : host_threads(), n / 2048));
    std::vector<std::unique_ptr<Searcher>> searchers;  // (they own the arenas: alive until the patterns are assembled)
    if (threads <= 1) {
        searchers.emplace_back(new Searcher(A, q, col_of_row, cols.get(), level, nullptr, &failed));
        for (i64 k = 0; k < n; ++k)
            if (!searchers[0]->column(k)) return false;
    } else {
        // elimination tree of B + B' (B = A in pivot coordinates: column k, row col_of_row[row]), Liu's algorithm
        std::vector<i64> lp((size_t)n + 1, 0);
        for (i64 k = 0; k < n; ++k) {
            const i64 col = q ? q[k] : k;
            for (i64 e = A.p[col]; e < A.p[col + 1]; ++e) {
                const i64 r = col_of_row[(size_t)A.i[e]];
                if (r != k) lp[(size_t)std::max(r, k) + 1]++;
            }
        }
        for (i64 k = 0; k < n; ++k) lp[(size_t)k + 1] += lp[(size_t)k];
        std::vector<int32_t> lo((size_t)lp[(size_t)n]);
        {
            std::vector<i64> at(lp.begin(), lp.end() - 1);
            for (i64 k = 0; k < n; ++k) {
                const i64 col = q ? q[k] : k;
                for (i64 e = A.p[col]; e < A.p[col + 1]; ++e) {
                    const i64 r = col_of_row[(size_t)A.i[e]];
                    if (r != k) lo[(size_t)at[(size_t)std::max(r, k)]++] = (int32_t)std::min(r, k);
                }
            }
        }
        std::vector<int32_t> parent((size_t)n, -1), ancestor((size_t)n, -1);
        for (i64 k = 0; k < n; ++k)
            for (i64 e = lp[(size_t)k]; e < lp[(size_t)k + 1]; ++e) {
                int32_t i = lo[(size_t)e];
                while (i != -1 && i < k) {
                    const int32_t up = ancestor[(size_t)i];
                    ancestor[(size_t)i] = (int32_t)k;
                    if (up == -1) parent[(size_t)i] = (int32_t)k;
                    i = up;
                }
            }
        // columns by the height of their subtree (children have smaller numbers), ascending inside a level
        std::vector<int32_t> height((size_t)n, 0);
        std::unique_ptr<std::atomic<int32_t>[]> pending(new std::atomic<int32_t>[(size_t)n]);
        std::unique_ptr<std::atomic<uint8_t>[]> done(new std::atomic<uint8_t>[(size_t)n]);
        for (i64 k = 0; k < n; ++k) {
            pending[(size_t)k].store(0, std::memory_order_relaxed);
            done[(size_t)k].store(0, std::memory_order_relaxed);
        }
        i64 heights = 0;
        for (i64 k = 0; k < n; ++k) {
            const int32_t p = parent[(size_t)k];
            if (p >= 0) {
                height[(size_t)p] = std::max(height[(size_t)p], height[(size_t)k] + 1);
                pending[(size_t)p].fetch_add(1, std::memory_order_relaxed);
            }
            heights = std::max<i64>(heights, height[(size_t)k] + 1);
        }
        std::vector<i64> at((size_t)heights + 1, 0);
        for (i64 k = 0; k < n; ++k) at[(size_t)height[(size_t)k] + 1]++;
        for (i64 h = 0; h < heights; ++h) at[(size_t)h + 1] += at[(size_t)h];
        std::vector<int32_t> sequence((size_t)n);
        for (i64 k = 0; k < n; ++k) sequence[(size_t)at[(size_t)height[(size_t)k]]++] = (int32_t)k;
        std::atomic<i64> next{0};
        for (i64 t = 0; t < threads; ++t) searchers.emplace_back(new Searcher(A, q, col_of_row, cols.get(), level, done.get(), &failed));
        const auto work = [&](Searcher &S) {
            for (;;) {
                const i64 t = next.fetch_add(1, std::memory_order_relaxed);
                if (t >= n || failed.load(std::memory_order_relaxed)) return;
                const i64 k = sequence[(size_t)t];
                // the children were taken earlier (lower height) by threads that are running: this wait ends
                for (i64 spins = 0; pending[(size_t)k].load(std::memory_order_acquire) != 0; ++spins) {
                    if (failed.load(std::memory_order_relaxed)) return;
                    if (spins > 64) std::this_thread::yield();
                }
                if (!S.column(k)) {
                    failed.store(true, std::memory_order_relaxed);
                    return;
                }
                done[(size_t)k].store(1, std::memory_order_release);
                if (parent[(size_t)k] >= 0) pending[(size_t)parent[(size_t)k]].fetch_sub(1, std::memory_order_acq_rel);
            }
        };
        std::vector<std::thread> pool;
        for (i64 t = 1; t < threads; ++t) pool.emplace_back(work, std::ref(*searchers[(size_t)t]));
        work(*searchers[0]);
        for (std::thread &th : pool) th.join();
        if (failed.load()) return false;
    }
    // assemble: column pointers, then the indices (row indices of L in pivot order: every row is the pivot of col_of_row[row])
    Csc &L = out.L, &U = out.U;
    L.m = L.n = U.m = U.n = n;
    L.p.assign((size_t)n + 1, 0);
    U.p.assign((size_t)n + 1, 0);
    for (i64 k = 0; k < n; ++k) {
        L.p[(size_t)k + 1] = L.p[(size_t)k] + cols[(size_t)k].llen;
        U.p[(size_t)k + 1] = U.p[(size_t)k] + cols[(size_t)k].ulen;
    }
    if (L.p[(size_t)n] >= INT32_MAX || U.p[(size_t)n] >= INT32_MAX) return false;
    L.i.resize((size_t)L.p[(size_t)n]);
    U.i.resize((size_t)U.p[(size_t)n]);
    parallel_ranges(n, 4096, [&](i64 k0, i64 k1) {
        for (i64 k = k0; k < k1; ++k) {
            const Column &c = cols[(size_t)k];
            i64 *li = L.i.data() + L.p[(size_t)k], *ui = U.i.data() + U.p[(size_t)k];
            for (int32_t t = 0; t < c.llen; ++t) li[t] = col_of_row[(size_t)c.l[t]];
            for (int32_t t = 0; t < c.ulen; ++t) ui[t] = c.u[t];
        }
    });
    out.pinv.assign(col_of_row.begin(), col_of_row.end());
    L.x.assign(L.i.size(), 0.0);
    U.x.assign(U.i.size(), 0.0);
    return true;
}

}  // namespace rsp
