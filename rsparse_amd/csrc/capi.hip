// extern "C" surface of librsparse_hip.so (declared in include/rsparse_hip.h).
// Nothing throws across this boundary: every entry point maps exceptions to an rsp_status.
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <utility>
#include <vector>
#include <new>

#include "../../include/rsparse_hip.h"
#include "device/chol_engine.hpp"
#include "device/engine.hpp"
#include "host/parallel.hpp"
#include "host/qr.hpp"

using namespace rsp;

// Every handle that solves carries a guard: a mutex around the host-side state a call touches (lazily built programmes,
// workspace buffers) and an event that orders the DEVICE work of consecutive calls on different streams -- a handle's
// scratch rows and workspaces are shared by all of its calls, so a call on another stream first waits for the previous
// call's kernels.  Calls on one handle are therefore safe from any number of host threads; they do not overlap.
struct UseGuard {
    std::mutex m;
    hipEvent_t done = nullptr;
    hipStream_t last = nullptr;
    bool used = false;
    ~UseGuard()
    {
        if (done) (void)hipEventDestroy(done);
    }
    void begin(hipStream_t stream)
    {
        if (used && last != stream) RSP_HIP(hipStreamWaitEvent(stream, done, 0));
    }
    void end(hipStream_t stream)
    {
        if (!done) RSP_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming));
        RSP_HIP(hipEventRecord(done, stream));
        last = stream;
        used = true;
    }
};

struct rsp_tri {
    TriSolver solver;
    UseGuard guard;
    int device = -1;                                  // where the handle lives: the device current at its first solve
    std::vector<std::unique_ptr<TriSolver>> replicas;  // rsp_set_devices: copies of the resident programme, by slot of the device list
};
// rsp_set_devices: the devices that entry points with independent work spread it over
static std::mutex g_devices_guard;
static std::vector<int> g_devices;
static std::vector<int> listed_devices()
{
    std::lock_guard<std::mutex> lock(g_devices_guard);
    return g_devices;
}
struct rsp_symb {
    Symbolic s;
};
struct rsp_chol {
    CholFactor f;
    UseGuard guard;
};
struct rsp_mat {  // a sparse matrix resident on the device, row-wise (gaxpy / residuals of whole blocks)
    int64_t m = 0, n = 0, nnz = 0;
    DeviceBuffer<int64_t> ptr;
    DeviceBuffer<int32_t> col;
    DeviceBuffer<double> val;
};
struct rsp_lu {
    LuFactors f;
    std::unique_ptr<TriSolver> lower, upper;  // built on first solve
    bool single_use = false;                  // lusol: one solve, then dropped
    bool on_device = false;                   // the values came from the GPU numeric LU (static pivots held), not from host/lu.cpp
    std::unique_ptr<LuDevice> device;         // ... and its device half, for rsp_lu_refactor_f64
    std::vector<int64_t> q_factor;            // the column order the factorisation was made with (empty: natural)
    bool valid = true;                        // false after a refactorisation that found no pivot: the handle holds NO
                                              // factorisation until a refactorisation succeeds (solves return RSP_NO_PIVOT)
    bool values_on_device_only = false;       // a refactorisation left the new values of L and U on the device: the host copy
                                              // (f.L.x, f.U.x) is brought up to date when somebody asks for it
    uint64_t pattern = 0;                     // pattern_hash of the matrix that was factored, and its entry count:
    int64_t pattern_nnz = -1;                 // rsp_lu_refactor_f64 takes new values on that pattern only
    DeviceBuffer<int64_t> d_pinv, d_q;        // rsp_lu_solve_dev_f64: the two permutations on the device
    std::vector<int64_t> q_host;              // ... and what d_q holds
    DeviceBuffer<double> work;
    UseGuard guard;
};

struct rsp_qr {
    QrFactors f;
    int64_t m = 0, n = 0;                // of the matrix that was factored
    std::unique_ptr<TriSolver> upper[2];  // usolve / utsolve with R, built on first use
    UseGuard guard;
};

namespace {

CscView view_of(const rsp_csc_f64 *A)
{
    return CscView{A->m, A->n, A->p, reinterpret_cast<const int64_t *>(A->i), A->x};
}

bool sane(const rsp_csc_f64 *A)
{
    if (!A || A->n < 0 || A->m < 0 || !A->p) return false;
    if (A->p[0] != 0 || A->p[A->n] < 0) return false;
    if (A->p[A->n] > 0 && (!A->i || !A->x)) return false;
    return true;
}

bool sane_indices(const rsp_csc_f64 *A)
{
    if (!sane(A)) return false;
    for (int64_t j = 0; j < A->n; ++j)
        if (A->p[j + 1] < A->p[j]) return false;
    const int64_t nz = A->p[A->n];
    for (int64_t e = 0; e < nz; ++e)
        if (A->i[e] >= (uint64_t)A->m) return false;
    return true;
}

template <class F>
int guarded(F &&f)
{
    try {
        return f();
    } catch (const DeviceError &) {
        return RSP_DEVICE_ERROR;
    } catch (const std::bad_alloc &) {
        return RSP_DEVICE_ERROR;
    } catch (const std::invalid_argument &) {  // what the library throws for input it rejects late
        return RSP_BAD_INPUT;
    } catch (const std::exception &e) {  // anything else is a defect of the library, not of the caller's input
        if (getenv("RSP_TIMING") || getenv("RSP_DEBUG")) fprintf(stderr, "rsparse_hip: internal error: %s\n", e.what());
        return RSP_INTERNAL_ERROR;
    } catch (...) {
        return RSP_INTERNAL_ERROR;
    }
}

int require_device()
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return RSP_DEVICE_ERROR;
    return RSP_OK;
}

int one_shot(const rsp_csc_f64 *T, int kind, double *x)
{
    if (!sane(T) || !x) return RSP_BAD_INPUT;
    return guarded([&] {
        if (int st = require_device()) return st;
        TriSolver s;
        if (int st = s.analyze(view_of(T), kind)) return st;
        s.set_single_use();
        s.solve_host(x, T->n, 1);
        return (int)RSP_OK;
    });
}

// x(P) = b for every right-hand side column (ipvec, lib.rs:2151-2159); P == nullptr is the identity
void scatter_rows(int64_t n, const int64_t *p, const double *b, double *x, int64_t ld, int64_t nrhs)
{
    for (int64_t r = 0; r < nrhs; ++r)
        for (int64_t k = 0; k < n; ++k) x[(p ? p[k] : k) + r * n] = b[k + r * ld];
}

}  // namespace

extern "C" {

const char *rsp_status_string(int status)
{
    switch (status) {
        case RSP_OK: return "ok";
        case RSP_NOT_POSITIVE_DEFINITE:
            return "Could not complete Cholesky factorization. Please provide a positive definite matrix";
        case RSP_NO_PIVOT: return "Could not find a pivot";
        case RSP_BAD_INPUT: return "bad input";
        case RSP_DEVICE_ERROR: return "HIP device error (is a gfx950 GPU visible?)";
        case RSP_INTERNAL_ERROR: return "internal error of the library (not caused by the input; please report)";
        default: return "unknown status";
    }
}

int rsp_device_count(void)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}

int rsp_set_device(int device) { return hipSetDevice(device) == hipSuccess ? RSP_OK : RSP_DEVICE_ERROR; }

int rsp_lsolve_f64(const rsp_csc_f64 *L, double *x) { return one_shot(L, RSP_LSOLVE, x); }
int rsp_ltsolve_f64(const rsp_csc_f64 *L, double *x) { return one_shot(L, RSP_LTSOLVE, x); }
int rsp_usolve_f64(const rsp_csc_f64 *U, double *x) { return one_shot(U, RSP_USOLVE, x); }
int rsp_utsolve_f64(const rsp_csc_f64 *U, double *x) { return one_shot(U, RSP_UTSOLVE, x); }

int rsp_tri_analyze(const rsp_csc_f64 *T, int kind, rsp_tri **out)
{
    if (!sane(T) || !out) return RSP_BAD_INPUT;
    *out = nullptr;
    return guarded([&] {
        std::unique_ptr<rsp_tri> h(new rsp_tri);
        if (int st = h->solver.analyze(view_of(T), kind)) return st;
        *out = h.release();
        return (int)RSP_OK;
    });
}

int rsp_tri_solve_f64(rsp_tri *h, double *X, int64_t ldx, int64_t nrhs)
{
    if (!h || !X || nrhs < 0 || ldx < h->solver.n()) return RSP_BAD_INPUT;
    return guarded([&] {
        if (int st = require_device()) return st;
        std::lock_guard<std::mutex> lock(h->guard.m);
        if (h->device < 0) RSP_HIP(hipGetDevice(&h->device));
        const std::vector<int> devs = listed_devices();
        const int64_t G = (int64_t)devs.size(), n = h->solver.n();
        if (G <= 1 || nrhs < 2 * G || n == 0) {
            h->solver.solve_host(X, ldx, nrhs);  // synchronous: returns with the device idle
            return (int)RSP_OK;
        }
        // column blocks over the listed devices, one host thread per entry; every device but the handle's own works on a replica
        // of the resident programme, copied device to device when first needed (TriSolver::clone_resident)
        h->solver.prepare_blocked((nrhs + G - 1) / G);
        RSP_HIP(hipDeviceSynchronize());
        if ((int64_t)h->replicas.size() < G) h->replicas.resize((size_t)G);
        std::vector<std::string> errors((size_t)G);
        auto share = [&](int64_t g) {
            try {
                const int64_t lo = nrhs * g / G, hi = nrhs * (g + 1) / G, cols = hi - lo;
                RSP_HIP(hipSetDevice(devs[(size_t)g]));
                TriSolver *s = &h->solver;
                if (!(g == 0 && devs[0] == h->device)) {
                    std::unique_ptr<TriSolver> &R = h->replicas[(size_t)g];
                    if (!R) {
                        R.reset(new TriSolver);
                        if (!h->solver.clone_resident(*R, h->device, devs[(size_t)g])) throw DeviceError("this handle's programme cannot be replicated");
                        RSP_HIP(hipDeviceSynchronize());
                    }
                    s = R.get();
                }
                DeviceBuffer<double> d((size_t)n * (size_t)cols);
                copy_block_to_device(d.get(), X + lo * ldx, n, ldx, cols);
                hipStream_t stream = nullptr;
                RSP_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
                try {
                    s->solve_device(d.get(), 0, n, cols, stream);
                    RSP_HIP(hipStreamSynchronize(stream));
                } catch (...) {
                    (void)hipStreamDestroy(stream);
                    throw;
                }
                (void)hipStreamDestroy(stream);
                copy_block_to_host(X + lo * ldx, d.get(), n, ldx, cols);
            } catch (const std::exception &e) {
                errors[(size_t)g] = e.what()[0] ? e.what() : "error";
            }
        };
        int before = 0;
        RSP_HIP(hipGetDevice(&before));
        std::vector<std::thread> pool;
        for (int64_t g = 1; g < G; ++g) pool.emplace_back(share, g);
        share(0);
        for (std::thread &t : pool) t.join();
        (void)hipSetDevice(before);
        for (const std::string &e : errors)
            if (!e.empty()) throw DeviceError("sharded solve: " + e);
        return (int)RSP_OK;
    });
}

int rsp_tri_solve_dev_f64(rsp_tri *h, double *dX, int layout, int64_t ld, int64_t nrhs, void *stream)
{
    if (!h || !dX || nrhs < 0 || (layout != RSP_COL_MAJOR && layout != RSP_ROW_MAJOR)) return RSP_BAD_INPUT;
    if (layout == RSP_COL_MAJOR ? ld < h->solver.n() : ld < nrhs) return RSP_BAD_INPUT;
    return guarded([&] {
        std::lock_guard<std::mutex> lock(h->guard.m);
        if (h->device < 0) RSP_HIP(hipGetDevice(&h->device));
        h->guard.begin((hipStream_t)stream);
        h->solver.solve_device(dX, layout, ld, nrhs, (hipStream_t)stream);
        h->guard.end((hipStream_t)stream);
        return (int)RSP_OK;
    });
}

int rsp_tri_solve_dev_noalloc_f64(rsp_tri *h, double *dX, int64_t ld, int64_t nrhs, void *stream)
{
    if (!h || !dX || nrhs < 0 || ld < nrhs) return RSP_BAD_INPUT;
    return guarded([&] {
        std::lock_guard<std::mutex> lock(h->guard.m);  // host state only: no event is created or recorded here (nothing may allocate)
        return h->solver.solve_device_noalloc(dX, ld, nrhs, (hipStream_t)stream);
    });
}

int rsp_tri_solve_pair_dev_f64(rsp_tri *first, rsp_tri *second, double *dX, int64_t ld, int64_t nrhs, void *stream)
{
    if (!first || !second || !dX || nrhs < 0 || ld < nrhs || first->solver.n() != second->solver.n()) return RSP_BAD_INPUT;
    return guarded([&] {
        // both handles, in address order (two threads pairing the same handles the other way round must not deadlock)
        std::unique_lock<std::mutex> la(first < second ? first->guard.m : second->guard.m);
        std::unique_lock<std::mutex> lb;
        if (first != second) lb = std::unique_lock<std::mutex>(first < second ? second->guard.m : first->guard.m);
        first->guard.begin((hipStream_t)stream);
        if (first != second) second->guard.begin((hipStream_t)stream);
        solve_pair_device(first->solver, second->solver, dX, ld, nrhs, (hipStream_t)stream);
        first->guard.end((hipStream_t)stream);
        if (first != second) second->guard.end((hipStream_t)stream);
        return (int)RSP_OK;
    });
}

int rsp_tri_info(const rsp_tri *h, int64_t *info)
{
    if (!h || !info) return RSP_BAD_INPUT;
    std::lock_guard<std::mutex> lock(const_cast<rsp_tri *>(h)->guard.m);
    const TriSolver &s = h->solver;
    const TriProgram &P = s.multi_program();
    info[0] = s.n();
    info[1] = s.nnz();
    info[2] = s.num_row_levels();
    const BlockedProgram &Bp = s.blocked_program();
    const bool blocked = Bp.nblocks > 0;  // the programme the multi-rhs path actually ran
    info[3] = blocked ? Bp.ntasks : P.ntasks;
    info[4] = blocked ? Bp.launches : P.nlevels;
    info[5] = (int64_t)(blocked ? Bp.device_bytes() : P.device_bytes());
    info[6] = s.algorithmic_bytes(1);
    info[7] = blocked ? Bp.inverted_rows : 0;
    info[8] = blocked ? Bp.scratch_rows : 0;
    info[9] = blocked ? Bp.nitems : 0;
    info[10] = blocked ? Bp.item_entries : 0;
    return RSP_OK;
}

int rsp_tri_levels(const rsp_tri *h, int64_t *level_of_row)
{
    if (!h || !level_of_row) return RSP_BAD_INPUT;
    const auto &lv = h->solver.level_of_row();
    std::memcpy(level_of_row, lv.data(), lv.size() * sizeof(int64_t));
    return RSP_OK;
}

void rsp_tri_free(rsp_tri *h) { delete h; }

int rsp_selftest_host_threads(int64_t count, int threads, uint64_t seed)
{
    if (count < 0) return RSP_BAD_INPUT;
    return guarded([&] {
        struct Override {  // this thread only; the process environment is left alone (other threads read it)
            int64_t keep;
            explicit Override(int t) : keep(host_threads_override()) { if (t > 0) host_threads_override() = t; }
            ~Override() { host_threads_override() = keep; }
        } scoped(threads);
        std::vector<std::pair<uint64_t, uint32_t>> a((size_t)count);
        uint64_t z = seed;
        for (int64_t k = 0; k < count; ++k) {  // SplitMix64, keys with many ties, unique ordinals
            z += 0x9e3779b97f4a7c15ull;
            uint64_t v = z;
            v = (v ^ (v >> 30)) * 0xbf58476d1ce4e5b9ull;
            v = (v ^ (v >> 27)) * 0x94d049bb133111ebull;
            a[k] = {(v ^ (v >> 31)) % 1000, (uint32_t)k};
        }
        auto b = a;
        std::sort(b.begin(), b.end());
        parallel_sort(a.begin(), a.end());
        std::vector<int64_t> f((size_t)count, -1);
        parallel_ranges(count, 1, [&](int64_t lo, int64_t hi) {
            for (int64_t k = lo; k < hi; ++k) f[k] += k + 2;  // every index exactly once -> k + 1
        });
        bool ok = a == b;
        for (int64_t k = 0; k < count && ok; ++k) ok = f[k] == k + 1;
        return ok ? (int)RSP_OK : (int)RSP_BAD_INPUT;
    });
}

int rsp_tri_validate_program(const rsp_csc_f64 *T, int kind, int64_t task_blocks, int cluster_slots, int64_t *report,
                             double *max_rel_error)
{
    if (!sane(T) || !report || !max_rel_error) return RSP_BAD_INPUT;
    return guarded([&] {
        GatherForm G;
        if (int st = build_gather_form(view_of(T), kind, G)) return st;
        const int slots = cluster_slots > 0 ? cluster_slots : kClusterSlots;
        const BlockProgram P = build_block_program(G, task_blocks > 0 ? task_blocks : kTaskBlocks, slots);
        // both views of the programme: the wide items as their twin blocks, and as items
        ProgramReport R = check_block_program(G, P, false);
        if (!P.items.empty()) {
            const ProgramReport R2 = check_block_program(G, P, true);
            R.bad_index += R2.bad_index;
            R.order_violations += R2.order_violations;
            R.unwritten += R2.unwritten;
            if (!(R2.max_rel_error <= R.max_rel_error)) R.max_rel_error = R2.max_rel_error;
        }
        report[0] = R.bad_index;
        report[1] = R.order_violations;
        report[2] = R.unwritten;
        report[3] = (int64_t)P.level_ptr.size() - 1;
        report[4] = (int64_t)P.task_ptr.size() - 1;
        report[5] = (int64_t)P.clusters.size();
        report[6] = P.cluster_rows;
        int64_t launches = 0;
        for (size_t l = 0; l + 1 < P.level_ptr.size(); ++l)
            launches += (P.level_ptr[l + 1] > P.level_ptr[l]) + (P.cluster_level_ptr[l + 1] > P.cluster_level_ptr[l]);
        report[7] = launches;
        *max_rel_error = R.max_rel_error;
        return (int)RSP_OK;
    });
}

int rsp_tri_program_checksum(const rsp_csc_f64 *T, int kind, int64_t task_blocks, int cluster_slots, uint64_t *out)
{
    if (!sane(T) || !out) return RSP_BAD_INPUT;
    return guarded([&] {
        GatherForm G;
        G.keep_src = true;
        if (int st = build_gather_form(view_of(T), kind, G)) return st;
        const BlockProgram P = build_block_program(G, task_blocks > 0 ? task_blocks : kTaskBlocks, cluster_slots > 0 ? cluster_slots : kClusterSlots);
        auto fnv = [](const void *data, size_t bytes) {
            uint64_t h = 1469598103934665603ull;
            const unsigned char *p = static_cast<const unsigned char *>(data);
            for (size_t k = 0; k < bytes; ++k) h = (h ^ p[k]) * 1099511628211ull;
            return h;
        };
        out[0] = fnv(P.hdr.data(), P.hdr.size() * sizeof(BlockHeader)) ^ fnv(P.src_idx.data(), P.src_idx.size() * 4) ^
                 fnv(P.prologue.data(), P.prologue.size() * 4) ^ fnv(P.task_ptr.data(), P.task_ptr.size() * 8) ^
                 fnv(P.level_class_ptr.data(), P.level_class_ptr.size() * 8) ^ fnv(P.items.data(), P.items.size() * sizeof(WideItem)) ^
                 fnv(P.item_masks.data(), P.item_masks.size() * 4) ^ fnv(P.item_level_ptr.data(), P.item_level_ptr.size() * 8);
        out[1] = fnv(P.coef.data(), P.coef.size() * 8);
        out[2] = fnv(P.groups.data(), P.groups.size() * sizeof(GroupRec)) ^ fnv(P.clusters.data(), P.clusters.size() * sizeof(ClusterHeader)) ^
                 fnv(P.cluster_ids.data(), P.cluster_ids.size() * 4);
        out[3] = fnv(P.refill.coef_src.data(), P.refill.coef_src.size() * 4) ^ (uint64_t)P.scratch_rows ^ ((uint64_t)P.refill.pool_size << 20);
        return (int)RSP_OK;
    });
}

int rsp_tri_validate_refill(const rsp_csc_f64 *T, const double *new_values, int kind, int64_t task_blocks, int cluster_slots, int64_t *report)
{
    if (!sane(T) || !new_values || !report) return RSP_BAD_INPUT;
    return guarded([&] {
        const int slots = cluster_slots > 0 ? cluster_slots : kClusterSlots;
        const int64_t blocks = task_blocks > 0 ? task_blocks : kTaskBlocks;
        GatherForm G;
        G.keep_src = true;
        if (int st = build_gather_form(view_of(T), kind, G)) return st;
        BlockProgram P = build_block_program(G, blocks, slots);
        CscView T2 = view_of(T);
        T2.x = new_values;
        report[0] = P.refill.valid ? 1 : 0;
        report[1] = report[2] = 0;
        report[3] = (int64_t)P.coef.size();
        if (!P.refill.valid) return (int)RSP_OK;
        fill_gather_values(T2, kind, G);
        if (!refill_block_program(G, P)) return (int)RSP_INTERNAL_ERROR;
        GatherForm G2;
        if (int st = build_gather_form(T2, kind, G2)) return st;
        const BlockProgram Q = build_block_program(G2, blocks, slots);
        if (Q.coef.size() != P.coef.size() || Q.groups.size() != P.groups.size()) return (int)RSP_INTERNAL_ERROR;
        auto same = [](double a, double b) { return std::memcmp(&a, &b, sizeof(double)) == 0; };
        for (size_t e = 0; e < P.coef.size(); ++e) report[1] += !same(P.coef[e], Q.coef[e]);
        for (size_t g = 0; g < P.groups.size(); ++g) {
            bool ok = same(P.groups[g].rdiag, Q.groups[g].rdiag);
            for (int e = 0; e < 4; ++e) ok = ok && same(P.groups[g].coef[e], Q.groups[g].coef[e]);
            report[2] += !ok;
        }
        return (int)RSP_OK;
    });
}

// ---- binary interchange format of Sprs ------------------------------------------------------------
// 32-byte header: magic "RSPSPRS1", m, n, nnz (int64, little endian), then p[n+1] (int64), i[nnz] (uint64), x[nnz] (f64):
// the reference's three Vecs as they sit in memory, so a factor of 5e8 entries loads at disk speed instead of being parsed
// from text (the reference's Sprs::save / Sprs::load, data.rs:414-516, write and parse decimal text).
namespace {
constexpr char kSprsMagic[8] = {'R', 'S', 'P', 'S', 'P', 'R', 'S', '1'};
struct FileCloser {
    void operator()(FILE *f) const { if (f) fclose(f); }
};
}  // namespace

int rsp_sprs_save_bin(const rsp_csc_f64 *A, const char *path)
{
    if (!sane(A) || !path) return RSP_BAD_INPUT;
    return guarded([&] {
        std::unique_ptr<FILE, FileCloser> f(fopen(path, "wb"));
        if (!f) return (int)RSP_BAD_INPUT;
        const int64_t nnz = A->p[A->n], dims[3] = {A->m, A->n, nnz};
        bool ok = fwrite(kSprsMagic, 1, 8, f.get()) == 8 && fwrite(dims, 8, 3, f.get()) == 3 &&
                  fwrite(A->p, 8, (size_t)A->n + 1, f.get()) == (size_t)A->n + 1;
        if (nnz) ok = ok && fwrite(A->i, 8, (size_t)nnz, f.get()) == (size_t)nnz && fwrite(A->x, 8, (size_t)nnz, f.get()) == (size_t)nnz;
        return ok ? (int)RSP_OK : (int)RSP_BAD_INPUT;
    });
}

int rsp_sprs_load_bin_dims(const char *path, int64_t *m, int64_t *n, int64_t *nnz)
{
    if (!path || !m || !n || !nnz) return RSP_BAD_INPUT;
    return guarded([&] {
        std::unique_ptr<FILE, FileCloser> f(fopen(path, "rb"));
        char magic[8];
        int64_t dims[3];
        if (!f || fread(magic, 1, 8, f.get()) != 8 || std::memcmp(magic, kSprsMagic, 8) != 0 || fread(dims, 8, 3, f.get()) != 3 || dims[0] < 0 ||
            dims[1] < 0 || dims[2] < 0)
            return (int)RSP_BAD_INPUT;
        // the header must account for the file's length exactly, so a corrupt count cannot make the caller allocate wildly
        if (fseeko(f.get(), 0, SEEK_END) != 0) return (int)RSP_BAD_INPUT;
        const unsigned long long len = (unsigned long long)ftello(f.get());
        if (dims[1] > (int64_t)(len / 8) || dims[2] > (int64_t)(len / 16) || len != 32ull + 8ull * (dims[1] + 1) + 16ull * dims[2]) return (int)RSP_BAD_INPUT;
        *m = dims[0];
        *n = dims[1];
        *nnz = dims[2];
        return (int)RSP_OK;
    });
}

int rsp_sprs_load_bin(const char *path, int64_t *p, uint64_t *i, double *x)
{
    if (!path || !p) return RSP_BAD_INPUT;
    return guarded([&] {
        std::unique_ptr<FILE, FileCloser> f(fopen(path, "rb"));
        char magic[8];
        int64_t dims[3];
        if (!f || fread(magic, 1, 8, f.get()) != 8 || std::memcmp(magic, kSprsMagic, 8) != 0 || fread(dims, 8, 3, f.get()) != 3 || dims[0] < 0 ||
            dims[1] < 0 || dims[2] < 0)
            return (int)RSP_BAD_INPUT;
        const size_t n = (size_t)dims[1], nnz = (size_t)dims[2];
        if (nnz && (!i || !x)) return (int)RSP_BAD_INPUT;
        bool ok = fread(p, 8, n + 1, f.get()) == n + 1;
        if (nnz) ok = ok && fread(i, 8, nnz, f.get()) == nnz && fread(x, 8, nnz, f.get()) == nnz;
        if (!ok || p[0] != 0 || p[n] != dims[2]) return (int)RSP_BAD_INPUT;  // truncated or inconsistent file
        for (size_t j = 0; j < n; ++j)
            if (p[j + 1] < p[j]) return (int)RSP_BAD_INPUT;
        for (size_t e = 0; e < nnz; ++e)
            if (i[e] >= (uint64_t)dims[0]) return (int)RSP_BAD_INPUT;
        return (int)RSP_OK;
    });
}

// ---- gaxpy on device-resident blocks ------------------------------------------------------------

int rsp_mat_create_f64(const rsp_csc_f64 *A, rsp_mat **M)
{
    if (!sane_indices(A) || !M || A->n > INT32_MAX) return RSP_BAD_INPUT;
    *M = nullptr;
    return guarded([&] {
        if (int st = require_device()) return st;
        std::unique_ptr<rsp_mat> h(new rsp_mat);
        const int64_t m = A->m, n = A->n, nnz = A->p[n];
        // CSC -> CSR by counting; inside a row the entries keep the order of their columns (the order the reference's
        // column loop adds them to that row in), duplicates included
        std::vector<int64_t> ptr((size_t)m + 1, 0);
        for (int64_t e = 0; e < nnz; ++e) ptr[A->i[e] + 1]++;
        for (int64_t r = 0; r < m; ++r) ptr[r + 1] += ptr[r];
        std::vector<int64_t> fill(ptr.begin(), ptr.end() - 1);
        std::vector<int32_t> col((size_t)nnz);
        std::vector<double> val((size_t)nnz);
        for (int64_t j = 0; j < n; ++j)
            for (int64_t e = A->p[j]; e < A->p[j + 1]; ++e) {
                const int64_t at = fill[A->i[e]]++;
                col[at] = (int32_t)j;
                val[at] = A->x[e];
            }
        h->m = m;
        h->n = n;
        h->nnz = nnz;
        h->ptr.upload(ptr);
        h->col.upload(col);
        h->val.upload(val);
        *M = h.release();
        return (int)RSP_OK;
    });
}

int rsp_mat_gaxpy_dev_f64(const rsp_mat *M, const double *dX, int64_t ldx, double *dY, int64_t ldy, int64_t nrhs, int layout, double beta,
                          void *stream)
{
    if (!M || !dX || !dY || nrhs < 0 || nrhs > INT32_MAX || (layout != RSP_COL_MAJOR && layout != RSP_ROW_MAJOR)) return RSP_BAD_INPUT;
    if (layout == RSP_COL_MAJOR ? (ldx < M->n || ldy < M->m) : (ldx < nrhs || ldy < nrhs)) return RSP_BAD_INPUT;
    if (nrhs == 0 || M->m == 0) return RSP_OK;
    return guarded([&] {
        const dev::CsrArrays A{M->ptr.get(), M->col.get(), M->val.get()};
        if (layout == RSP_ROW_MAJOR) {
            const int nchunks = (int)((nrhs + 63) / 64);
            const int64_t wgs = (M->m * nchunks + 3) / 4;
            if (wgs > INT32_MAX) return (int)RSP_BAD_INPUT;
            hipLaunchKernelGGL(dev::gaxpy_rowmajor, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, A, M->m, dX, ldx, dY, ldy, (int)nrhs,
                               nchunks, beta);
        } else {
            if (nrhs > 65535) return (int)RSP_BAD_INPUT;
            hipLaunchKernelGGL(dev::gaxpy_colmajor, dim3((unsigned)((M->m + 255) / 256), (unsigned)nrhs), dim3(256), 0, (hipStream_t)stream, A,
                               M->m, dX, ldx, dY, ldy, (int)nrhs, beta);
        }
        RSP_HIP(hipGetLastError());
        return (int)RSP_OK;
    });
}

// Iterative refinement of a solved block, all on the device: per step  W = A X - B,  W = A^-1 W (the resident factor:
// P, lsolve, ltsolve, P' of cholsol, lib.rs:383-386),  X = X - W.
int rsp_chol_refine_dev_f64(rsp_chol *F, const rsp_mat *M, const double *dB, double *dX, double *dW, int layout, int64_t ld, int64_t nrhs,
                            int steps, void *stream)
{
    if (!F || !M || !dB || !dX || !dW || nrhs < 0 || nrhs > INT32_MAX || steps < 0 || (layout != RSP_COL_MAJOR && layout != RSP_ROW_MAJOR))
        return RSP_BAD_INPUT;
    const int64_t n = F->f.n();
    if (M->m != n || M->n != n || (layout == RSP_COL_MAJOR ? ld < n : ld < nrhs)) return RSP_BAD_INPUT;
    if (dW == dX || dW == dB || dX == dB) return RSP_BAD_INPUT;
    if (nrhs == 0 || n == 0) return RSP_OK;
    return guarded([&] {
        std::lock_guard<std::mutex> lock(F->guard.m);
        if (!F->f.valid()) return (int)RSP_NOT_POSITIVE_DEFINITE;
        hipStream_t st = (hipStream_t)stream;
        F->guard.begin(st);
        const int64_t rows = layout == RSP_ROW_MAJOR ? n : nrhs, cols = layout == RSP_ROW_MAJOR ? nrhs : n;
        const unsigned grid = (unsigned)std::min<int64_t>((rows * cols + 255) / 256, 1 << 16);
        for (int s = 0; s < steps; ++s) {
            hipLaunchKernelGGL(dev::block_axpby, dim3(grid), dim3(256), 0, st, dW, ld, dB, ld, rows, cols, 1.0, 0.0);
            if (int e = rsp_mat_gaxpy_dev_f64(M, dX, ld, dW, ld, nrhs, layout, -1.0, stream)) return e;
            F->f.solve_device(dW, layout, ld, nrhs, true, st);
            hipLaunchKernelGGL(dev::block_axpby, dim3(grid), dim3(256), 0, st, dX, ld, dW, ld, rows, cols, -1.0, 1.0);
        }
        RSP_HIP(hipGetLastError());
        F->guard.end(st);
        return (int)RSP_OK;
    });
}

void rsp_mat_free(rsp_mat *M) { delete M; }

int rsp_gaxpy_f64(const rsp_csc_f64 *A, const double *x, const double *y, double *r)
{
    if (!sane_indices(A) || !x || !y || !r) return RSP_BAD_INPUT;
    rsp_mat *M = nullptr;
    int st = rsp_mat_create_f64(A, &M);
    if (st == RSP_OK)
        st = guarded([&] {
            DeviceBuffer<double> dx, dy;
            dx.upload(x, (size_t)A->n);
            dy.upload(y, (size_t)A->m);
            if (int s2 = rsp_mat_gaxpy_dev_f64(M, dx.get(), A->n, dy.get(), A->m, 1, RSP_COL_MAJOR, 1.0, nullptr)) return s2;
            RSP_HIP(hipDeviceSynchronize());
            dy.download(r, (size_t)A->m);
            return (int)RSP_OK;
        });
    rsp_mat_free(M);
    return st;
}

// ---- symbolic -----------------------------------------------------------------------------------

int rsp_schol(const rsp_csc_f64 *A, int8_t order, rsp_symb **S)
{
    if (!sane_indices(A) || !S || A->m != A->n) return RSP_BAD_INPUT;
    *S = nullptr;
    return guarded([&] {
        std::unique_ptr<rsp_symb> h(new rsp_symb);
        h->s = symbolic_cholesky(view_of(A), order);
        *S = h.release();
        return (int)RSP_OK;
    });
}

int rsp_sqr_lu(const rsp_csc_f64 *A, int8_t order, rsp_symb **S)
{
    if (!sane_indices(A) || !S) return RSP_BAD_INPUT;
    *S = nullptr;
    return guarded([&] {
        std::unique_ptr<rsp_symb> h(new rsp_symb);
        h->s = symbolic_lu(view_of(A), order);
        *S = h.release();
        return (int)RSP_OK;
    });
}

int rsp_sqr_qr(const rsp_csc_f64 *A, int8_t order, rsp_symb **S)
{
    if (!sane_indices(A) || !S) return RSP_BAD_INPUT;
    *S = nullptr;
    return guarded([&] {
        std::unique_ptr<rsp_symb> h(new rsp_symb);
        h->s = symbolic_qr(view_of(A), order);
        *S = h.release();
        return (int)RSP_OK;
    });
}

int rsp_symb_from_parts(int64_t n, const int64_t *pinv, const int64_t *q, const int64_t *parent, const int64_t *cp, rsp_symb **S)
{
    if (n < 0 || !S) return RSP_BAD_INPUT;
    *S = nullptr;
    // The vectors are used as indices later (chol writes Li[cp[..]++], lu reads A.p[q[k]]): reject anything that is not
    // what schol / sqr would have produced -- the reference would panic on a bounds check there.
    auto permutation = [n](const int64_t *v) {
        std::vector<char> seen((size_t)n, 0);
        for (int64_t k = 0; k < n; ++k) {
            if (v[k] < 0 || v[k] >= n || seen[v[k]]) return false;
            seen[v[k]] = 1;
        }
        return true;
    };
    if (pinv && !permutation(pinv)) return RSP_BAD_INPUT;
    if (q && !permutation(q)) return RSP_BAD_INPUT;
    if (parent)
        for (int64_t k = 0; k < n; ++k)
            if (parent[k] < -1 || parent[k] >= n) return RSP_BAD_INPUT;
    if (cp) {
        if (cp[0] != 0) return RSP_BAD_INPUT;
        for (int64_t k = 0; k < n; ++k)  // every column of L holds at least its diagonal, at most n - k rows
            if (cp[k + 1] < cp[k] + 1 || cp[k + 1] - cp[k] > n - k) return RSP_BAD_INPUT;
    }
    return guarded([&] {
        std::unique_ptr<rsp_symb> h(new rsp_symb);
        Symbolic &s = h->s;
        s.n = n;
        if (pinv) { s.has_pinv = true; s.pinv.assign(pinv, pinv + n); }
        if (q) { s.has_q = true; s.q.assign(q, q + n); s.q.push_back(n); }
        if (parent) s.parent.assign(parent, parent + n);
        if (cp) { s.cp.assign(cp, cp + n + 1); s.lnz = s.unz = cp[n]; }
        *S = h.release();
        return (int)RSP_OK;
    });
}

int64_t rsp_symb_n(const rsp_symb *S) { return S ? S->s.n : 0; }
const int64_t *rsp_symb_pinv(const rsp_symb *S) { return (S && S->s.has_pinv) ? S->s.pinv.data() : nullptr; }
const int64_t *rsp_symb_q(const rsp_symb *S) { return (S && S->s.has_q) ? S->s.q.data() : nullptr; }
const int64_t *rsp_symb_parent(const rsp_symb *S) { return (S && !S->s.parent.empty()) ? S->s.parent.data() : nullptr; }
const int64_t *rsp_symb_cp(const rsp_symb *S) { return (S && !S->s.cp.empty()) ? S->s.cp.data() : nullptr; }
int64_t rsp_symb_lnz(const rsp_symb *S) { return S ? S->s.lnz : 0; }
int64_t rsp_symb_m2(const rsp_symb *S) { return S ? S->s.m2 : 0; }
int64_t rsp_symb_pinv_len(const rsp_symb *S) { return (S && S->s.has_pinv) ? (int64_t)S->s.pinv.size() : 0; }
int64_t rsp_symb_cp_len(const rsp_symb *S) { return S ? (int64_t)S->s.cp.size() : 0; }
int64_t rsp_symb_unz(const rsp_symb *S) { return S ? S->s.unz : 0; }
void rsp_symb_free(rsp_symb *S) { delete S; }

int rsp_amd(const rsp_csc_f64 *A, int8_t order, int64_t *P, int *has_perm)
{
    if (!sane_indices(A) || !P || !has_perm) return RSP_BAD_INPUT;
    return guarded([&] {
        std::vector<int64_t> perm;
        *has_perm = approximate_minimum_degree(view_of(A), order, perm) ? 1 : 0;
        if (*has_perm) std::memcpy(P, perm.data(), perm.size() * sizeof(int64_t));
        return (int)RSP_OK;
    });
}

// ---- Cholesky -----------------------------------------------------------------------------------

static int chol_factor_impl(const rsp_csc_f64 *A, const rsp_symb *S, rsp_chol **F, bool single_use)
{
    if (!sane_indices(A) || !S || !F || A->m != A->n || S->s.n != A->n || S->s.parent.size() != (size_t)A->n ||
        S->s.cp.size() != (size_t)A->n + 1)
        return RSP_BAD_INPUT;
    *F = nullptr;
    return guarded([&] {
        if (int st = require_device()) return st;
        std::unique_ptr<rsp_chol> h(new rsp_chol);
        if (int st = h->f.analyze(view_of(A), S->s, single_use)) return st;
        if (int st = h->f.factorize(view_of(A))) return st;
        *F = h.release();
        return (int)RSP_OK;
    });
}

int rsp_chol_factor_f64(const rsp_csc_f64 *A, const rsp_symb *S, rsp_chol **F) { return chol_factor_impl(A, S, F, false); }

// pattern_verified: the caller has compared A's pattern with the analysed one entry by entry (the analysis cache)
static int chol_refactor_impl(rsp_chol *F, const rsp_csc_f64 *A, bool pattern_verified)
{
    if (!F || !sane(A) || A->m != A->n || A->n != F->f.n() || A->p[A->n] != F->f.annz()) return RSP_BAD_INPUT;
    return guarded([&] {
        std::lock_guard<std::mutex> lock(F->guard.m);
        // the values are scattered through the analysed pattern's maps: another pattern of the same size would give a wrong
        // factor without any error
        if (!pattern_verified && pattern_hash(view_of(A)) != F->f.pattern()) return (int)RSP_BAD_INPUT;
        return F->f.factorize(view_of(A));  // same analysis, same launches; the solve programmes receive the values on the next solve
    });
}
int rsp_chol_refactor_f64(rsp_chol *F, const rsp_csc_f64 *A) { return chol_refactor_impl(F, A, false); }

int rsp_chol_f64(const rsp_csc_f64 *A, const rsp_symb *S, int64_t *Lp, uint64_t *Li, double *Lx)
{
    if (!Lp || !Li || !Lx) return RSP_BAD_INPUT;
    rsp_chol *F = nullptr;
    int st = rsp_chol_factor_f64(A, S, &F);
    if (st == RSP_OK) st = rsp_chol_get_factor(F, Lp, Li, Lx);
    rsp_chol_free(F);
    return st;
}

int rsp_chol_get_factor(const rsp_chol *F, int64_t *Lp, uint64_t *Li, double *Lx)
{
    if (!F || !Lp || !Li || !Lx) return RSP_BAD_INPUT;
    return guarded([&] {
        std::lock_guard<std::mutex> lock(const_cast<rsp_chol *>(F)->guard.m);
        if (!F->f.valid()) return (int)RSP_NOT_POSITIVE_DEFINITE;  // the last (re)factorisation failed
        F->f.download(Lp, reinterpret_cast<int64_t *>(Li), Lx);
        return (int)RSP_OK;
    });
}

int rsp_chol_solve_f64(rsp_chol *F, double *B, int64_t ldb, int64_t nrhs)
{
    if (!F || !B || nrhs < 0 || ldb < F->f.n()) return RSP_BAD_INPUT;
    return guarded([&] {
        std::lock_guard<std::mutex> lock(F->guard.m);
        if (!F->f.valid()) return (int)RSP_NOT_POSITIVE_DEFINITE;  // the last (re)factorisation failed: nothing to solve with
        const std::vector<int> devs = listed_devices();
        if (devs.size() > 1) F->f.solve_host_sharded(B, ldb, nrhs, devs);  // column blocks per device, replicas of the programmes
        else F->f.solve_host(B, ldb, nrhs);
        return (int)RSP_OK;
    });
}

int rsp_chol_solve_dev_f64(rsp_chol *F, double *dB, int layout, int64_t ld, int64_t nrhs, int permute, void *stream)
{
    if (!F || !dB || nrhs < 0 || (layout != RSP_COL_MAJOR && layout != RSP_ROW_MAJOR)) return RSP_BAD_INPUT;
    if (layout == RSP_COL_MAJOR ? ld < F->f.n() : ld < nrhs) return RSP_BAD_INPUT;
    return guarded([&] {
        std::lock_guard<std::mutex> lock(F->guard.m);
        if (!F->f.valid()) return (int)RSP_NOT_POSITIVE_DEFINITE;
        F->guard.begin((hipStream_t)stream);
        F->f.solve_device(dB, layout, ld, nrhs, permute != 0, (hipStream_t)stream);
        F->guard.end((hipStream_t)stream);
        return (int)RSP_OK;
    });
}

int64_t rsp_chol_lnz(const rsp_chol *F) { return F ? F->f.lnz() : 0; }

int rsp_chol_factor_view(const rsp_chol *F, const int64_t **Lp, const uint64_t **Li, const double **dLx)
{
    if (!F || !Lp || !Li || !dLx) return RSP_BAD_INPUT;
    std::lock_guard<std::mutex> lock(const_cast<rsp_chol *>(F)->guard.m);
    if (!F->f.valid()) return RSP_NOT_POSITIVE_DEFINITE;
    *Lp = F->f.host_Lp();
    *Li = reinterpret_cast<const uint64_t *>(F->f.host_Li());
    *dLx = F->f.device_Lx();
    return RSP_OK;
}

int rsp_chol_info(const rsp_chol *F, int64_t *info)
{
    if (!F || !info) return RSP_BAD_INPUT;
    std::lock_guard<std::mutex> lock(const_cast<rsp_chol *>(F)->guard.m);
    F->f.info(info);
    return RSP_OK;
}

double rsp_chol_factor_ms(const rsp_chol *F) { return F ? F->f.factor_ms() : 0.0; }

void rsp_chol_free(rsp_chol *F) { delete F; }

// ---- analysis cache behind the one-shot signature --------------------------------------------------------
// The reference's cholsol(a, b, order) (lib.rs:377-389) carries no handle: a caller that solves with ONE pattern again and
// again (time stepping, Newton iterations) would pay ordering + symbolic analysis + programme emission on every call.
// rsp_cholsol_f64 therefore keeps the analysed handles of the last few patterns: after a call that missed, a background
// thread analyses the pattern again as a MULTI-USE handle (blocked solve programmes emitted ahead of the values) and puts
// it into the cache; later calls with the same (device, n, order, pattern) -- compared entry by entry, not by hash alone --
// refactorise on that analysis and solve, nothing else.  Least recently used entries leave when the entry or byte budget
// is exceeded (RSP_ANALYSIS_CACHE entries, default 2, 0 = off; RSP_ANALYSIS_CACHE_MB device memory, default 16384).
// An entry is taken OUT of the cache while a call uses it, so two threads never share a handle; the second thread of a
// pair takes the plain path.  The results of a hit and of a miss differ only by the order of the solves' additions.
extern "C++" {
namespace {
struct CholCacheEntry {
    int kind = 0;  // 0: cholsol (S + F), 1: lusol (S + N)
    uint64_t hash = 0;
    int device = 0;
    int8_t order = 0;
    int64_t n = 0, nnz = 0;
    std::vector<int64_t> p, i;  // the pattern itself: a hit is verified entry by entry
    std::unique_ptr<rsp_symb> S;
    std::unique_ptr<rsp_chol> F;
    std::unique_ptr<rsp_lu> N;
    int64_t bytes = 0;  // estimate of the device memory behind F / N
    uint64_t stamp = 0;
};
struct CholCache {
    std::mutex m;
    std::vector<std::unique_ptr<CholCacheEntry>> entries;
    std::vector<uint64_t> pending;  // hashes a background analysis is running for
    std::vector<std::thread> workers;
    uint64_t clock = 0;
    int64_t hits = 0, misses = 0, inserted = 0, evicted = 0;
    ~CholCache()
    {
        for (std::thread &t : workers)
            if (t.joinable()) t.join();
        // the process is ending: the HIP runtime may already be gone, the handles' device memory goes with the process
        for (auto &e : entries) {
            (void)e->F.release();
            (void)e->N.release();
            (void)e->S.release();
        }
    }
};
CholCache &chol_cache()
{
    static CholCache c;
    return c;
}
int64_t cache_env(const char *name, int64_t fallback)
{
    const char *e = getenv(name);
    return e ? std::atoll(e) : fallback;
}
int64_t chol_cache_capacity() { return std::max<int64_t>(0, cache_env("RSP_ANALYSIS_CACHE", 2)); }
int64_t chol_cache_budget() { return std::max<int64_t>(0, cache_env("RSP_ANALYSIS_CACHE_MB", 16384)) << 20; }
// systems smaller than this are not cached: a whole call costs a few milliseconds there, and a stream of small systems with
// different patterns (BASELINE.json's batch variant B) would only feed the background thread
int64_t chol_cache_min_n() { return std::max<int64_t>(1, cache_env("RSP_ANALYSIS_CACHE_MIN_N", 4000)); }

// entry by entry, on the host threads (50 MB at a million unknowns: 5 ms on one thread, a sixth of a cached call)
bool same_pattern(const CholCacheEntry &e, const rsp_csc_f64 *A)
{
    if (e.n != A->n || e.nnz != A->p[A->n]) return false;
    std::atomic<bool> same{true};
    auto compare = [&](const int64_t *a, const int64_t *b, int64_t count) {
        parallel_ranges(count, 1 << 18, [&](int64_t lo, int64_t hi) {
            if (std::memcmp(a + lo, b + lo, (size_t)(hi - lo) * sizeof(int64_t)) != 0) same.store(false);
        });
    };
    compare(e.p.data(), A->p, A->n + 1);
    if (same.load()) compare(e.i.data(), reinterpret_cast<const int64_t *>(A->i), e.nnz);
    return same.load();
}

// entries beyond the budgets leave, least recently used first (the caller holds the cache's mutex); returns what must be freed
// OUTSIDE the lock (freeing a handle synchronises the device)
std::vector<std::unique_ptr<CholCacheEntry>> chol_cache_trim(CholCache &C, int64_t capacity, int64_t budget)
{
    std::vector<std::unique_ptr<CholCacheEntry>> out;
    auto total = [&] {
        int64_t b = 0;
        for (auto &e : C.entries) b += e->bytes;
        return b;
    };
    while (!C.entries.empty() && ((int64_t)C.entries.size() > capacity || total() > budget)) {
        size_t oldest = 0;
        for (size_t k = 1; k < C.entries.size(); ++k)
            if (C.entries[k]->stamp < C.entries[oldest]->stamp) oldest = k;
        out.push_back(std::move(C.entries[oldest]));
        C.entries.erase(C.entries.begin() + (long)oldest);
        C.evicted++;
    }
    return out;
}

// background job of a miss: the multi-use analysis of the pattern (no values needed: the first hit brings them)
void chol_cache_analyse(std::unique_ptr<CholCacheEntry> e)
{
    CholCache &C = chol_cache();
    const uint64_t hash = e->hash;
    bool ok = false;
    try {
        if (hipSetDevice(e->device) == hipSuccess) {
            CscView A{e->n, e->n, e->p.data(), e->i.data(), nullptr};
            if (!e->S) {  // (a call that missed hands its own ordering and symbolic analysis over)
                e->S.reset(new rsp_symb);
                e->S->s = symbolic_cholesky(A, e->order);
            }
            e->F.reset(new rsp_chol);
            ok = e->F->f.analyze(A, e->S->s, false) == kOk;
            if (ok) e->F->f.finish_analysis();  // the solve programmes are emitted too: the first hit finds everything ready
            e->bytes = e->S->s.lnz * 48 + e->nnz * 24;
        }
    } catch (...) {
        ok = false;
    }
    std::vector<std::unique_ptr<CholCacheEntry>> dead;
    {
        std::lock_guard<std::mutex> lock(C.m);
        C.pending.erase(std::remove(C.pending.begin(), C.pending.end(), hash), C.pending.end());
        if (ok && e->bytes <= chol_cache_budget() && chol_cache_capacity() > 0) {
            e->stamp = ++C.clock;
            C.entries.push_back(std::move(e));
            C.inserted++;
            dead = chol_cache_trim(C, chol_cache_capacity(), chol_cache_budget());
        }
    }
    dead.clear();
    e.reset();
}
}  // namespace
}  // extern "C++"

int rsp_cache_clear(void)
{
    return guarded([&] {
        CholCache &C = chol_cache();
        std::vector<std::thread> workers;
        {
            std::lock_guard<std::mutex> lock(C.m);
            workers.swap(C.workers);
        }
        for (std::thread &t : workers)
            if (t.joinable()) t.join();
        std::vector<std::unique_ptr<CholCacheEntry>> dead;
        {
            std::lock_guard<std::mutex> lock(C.m);
            dead.swap(C.entries);
        }
        dead.clear();
        return (int)RSP_OK;
    });
}

int rsp_cache_stats(int64_t *out)
{
    if (!out) return RSP_BAD_INPUT;
    CholCache &C = chol_cache();
    std::lock_guard<std::mutex> lock(C.m);
    out[0] = C.hits;
    out[1] = C.misses;
    out[2] = (int64_t)C.entries.size();
    out[3] = 0;
    for (auto &e : C.entries) out[3] += e->bytes;
    out[4] = (int64_t)C.pending.size();
    out[5] = C.inserted;
    out[6] = C.evicted;
    return RSP_OK;
}

int rsp_cholsol_f64(const rsp_csc_f64 *A, double *b, int8_t order)
{
    if (!sane_indices(A) || !b || A->m != A->n) return RSP_BAD_INPUT;
    const auto clock = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = clock();
    // ---- the analysis cache: same pattern as an earlier call? ----
    const int64_t capacity = chol_cache_capacity();
    uint64_t hash = 0;
    int device = 0;
    bool spawn = false;
    if (capacity > 0 && A->n >= chol_cache_min_n()) {
        int st = guarded([&] {
            if (int rd = require_device()) return rd;
            RSP_HIP(hipGetDevice(&device));
            return (int)RSP_OK;
        });
        if (st != RSP_OK) return st;
        hash = pattern_hash(view_of(A));
        CholCache &C = chol_cache();
        std::unique_ptr<CholCacheEntry> hit;
        {
            std::lock_guard<std::mutex> lock(C.m);
            for (size_t k = 0; k < C.entries.size(); ++k) {
                CholCacheEntry &e = *C.entries[k];
                if (e.kind == 0 && e.hash == hash && e.device == device && e.order == order && e.n == A->n && e.nnz == A->p[A->n]) {
                    hit = std::move(C.entries[k]);  // out of the cache while this call uses it
                    C.entries.erase(C.entries.begin() + (long)k);
                    break;
                }
            }
        }
        if (hit && !same_pattern(*hit, A)) {  // a hash collision: not this pattern, back it goes
            std::lock_guard<std::mutex> lock(C.m);
            C.entries.push_back(std::move(hit));
            hit.reset();
        }
        if (hit) {
            st = chol_refactor_impl(hit->F.get(), A, true);  // numeric Cholesky on the cached analysis (GPU)
            const double t1 = clock();
            if (st == RSP_OK) st = rsp_chol_solve_f64(hit->F.get(), b, A->n, 1);  // P, lsolve, ltsolve, P' through the resident programmes
            if (getenv("RSP_TIMING")) fprintf(stderr, "cholsol (cached analysis): refactor %.3fs, solve %.3fs\n", t1 - t0, clock() - t1);
            std::vector<std::unique_ptr<CholCacheEntry>> dead;
            {
                std::lock_guard<std::mutex> lock(C.m);
                C.hits++;
                hit->stamp = ++C.clock;
                C.entries.push_back(std::move(hit));
                dead = chol_cache_trim(C, capacity, chol_cache_budget());
            }
            return st;
        }
        {
            std::lock_guard<std::mutex> lock(C.m);
            C.misses++;
            // one background analysis per pattern, a few at a time
            spawn = std::find(C.pending.begin(), C.pending.end(), hash) == C.pending.end() && C.pending.size() < 2;
            if (spawn) C.pending.push_back(hash);
        }
    }
    rsp_symb *S = nullptr;
    rsp_chol *F = nullptr;
    int st = rsp_schol(A, order, &S);                    // ordering and symbolic analysis (host)
    const double t1 = clock();
    if (st == RSP_OK) st = chol_factor_impl(A, S, &F, true);  // numeric Cholesky (GPU); one solve, then the factor is dropped
    const double t2 = clock();
    if (st == RSP_OK) F->f.set_single_use(true);         // one solve, then the factor is dropped
    if (st == RSP_OK) st = rsp_chol_solve_f64(F, b, A->n, 1);  // P, lsolve, ltsolve, P' (GPU)
    const double t3 = clock();
    rsp_chol_free(F);
    std::thread worker;
    if (spawn && st < 0) {  // bad input or a device failure: nothing worth analysing again
        std::lock_guard<std::mutex> lock(chol_cache().m);
        auto &pend = chol_cache().pending;
        pend.erase(std::remove(pend.begin(), pend.end(), hash), pend.end());
        spawn = false;
    }
    if (spawn) {
        // AFTER this call's own work (the two would share the host's cores: started first, the background analysis made the call
        // itself 10 % slower): the pattern is copied (the caller's arrays are only borrowed), this call's ordering and symbolic
        // analysis go along, and a thread of the library builds the multi-use analysis while the caller goes on
        std::unique_ptr<CholCacheEntry> e;
        bool started = false;
        try {
            e.reset(new CholCacheEntry);
            e->hash = hash;
            e->device = device;
            e->order = order;
            e->n = A->n;
            e->nnz = A->p[A->n];
            e->p.assign(A->p, A->p + A->n + 1);
            e->i.assign(reinterpret_cast<const int64_t *>(A->i), reinterpret_cast<const int64_t *>(A->i) + e->nnz);
            if (S) {
                e->S.reset(S);
                S = nullptr;
            }
            CholCacheEntry *raw = e.release();
            try {
                worker = std::thread([raw] { chol_cache_analyse(std::unique_ptr<CholCacheEntry>(raw)); });
                started = true;
            } catch (...) {
                delete raw;
            }
        } catch (...) {
            started = false;
        }
        if (!started) {
            std::lock_guard<std::mutex> lock(chol_cache().m);
            auto &pend = chol_cache().pending;
            pend.erase(std::remove(pend.begin(), pend.end(), hash), pend.end());
        }
    }
    rsp_symb_free(S);
    if (worker.joinable()) {
        std::lock_guard<std::mutex> lock(chol_cache().m);
        auto &ws = chol_cache().workers;
        ws.push_back(std::move(worker));
        if (ws.size() > 4) {  // the oldest job has long finished (at most two run at a time): collect it, so the list stays short
            std::thread old = std::move(ws.front());
            ws.erase(ws.begin());
            if (old.joinable()) old.join();
        }
    }
    if (getenv("RSP_TIMING"))
        fprintf(stderr, "cholsol: schol %.2fs, chol (analysis + factor) %.2fs, solve (analysis + solve) %.2fs, free %.2fs\n", t1 - t0,
                t2 - t1, t3 - t2, clock() - t3);
    return st;
}

// the whole batch on the calling thread's current device
static int cholsol_batch_on_device(const rsp_csc_f64 *A, const double *Ax_all, double *B, int64_t nsys, int8_t order, int32_t *status)
{
    return guarded([&] {
        if (int st = require_device()) return st;
        const int64_t n = A->n, nnz = A->p[n];
        const Symbolic S = symbolic_cholesky(view_of(A), order);  // one ordering and one analysis for the whole batch
        CholFactor F;
        if (n <= kBatchMaxN) F.no_solve_programmes();  // the batch is solved out of LDS, straight from the factors
        if (int st = F.analyze(view_of(A), S)) return st;
        bool any_bad = false;
        std::vector<int32_t> bad;
        std::vector<double> keep((size_t)n);
        if (n > kBatchMaxN) {  // too large for the LDS-resident solve: one system at a time through the ordinary path
            for (int64_t s = 0; s < nsys; ++s) {
                CscView V = view_of(A);
                V.x = Ax_all + s * nnz;
                F.set_single_use(true);
                const int st = F.factorize(V);
                if (st == kOk) F.solve_host(B + s * n, n, 1);
                else if (st != kNotPositiveDefinite) return st;
                if (status) status[s] = st;
                any_bad = any_bad || st != kOk;
            }
            return (!status && any_bad) ? (int)RSP_NOT_POSITIVE_DEFINITE : (int)RSP_OK;
        }
        // chunks bounded by device memory for the factors (8 GiB) and by the grid's second dimension
        const int64_t per = std::max<int64_t>(1, std::min<int64_t>(32768, (int64_t)(8ll << 30) / (8 * std::max<int64_t>(1, F.lnz()))));
        for (int64_t s0 = 0; s0 < nsys; s0 += per) {
            const int64_t cnt = std::min(per, nsys - s0);
            bad.assign((size_t)cnt, 0);
            F.factorize_values(Ax_all + s0 * nnz, cnt, bad.data());
            // a system that is not positive definite keeps its right-hand side (cholsol returns before touching b, lib.rs:380)
            std::vector<std::pair<int64_t, std::vector<double>>> saved;
            for (int64_t s = 0; s < cnt; ++s)
                if (bad[s] != INT_MAX) saved.emplace_back(s, std::vector<double>(B + (s0 + s) * n, B + (s0 + s + 1) * n));
            F.solve_batch(B + s0 * n, cnt);
            for (auto &sv : saved) std::copy(sv.second.begin(), sv.second.end(), B + (s0 + sv.first) * n);
            for (int64_t s = 0; s < cnt; ++s) {
                if (status) status[s0 + s] = bad[s] == INT_MAX ? RSP_OK : RSP_NOT_POSITIVE_DEFINITE;
                any_bad = any_bad || bad[s] != INT_MAX;
            }
        }
        return (!status && any_bad) ? (int)RSP_NOT_POSITIVE_DEFINITE : (int)RSP_OK;
    });
}


// devices the multi-device entry points spread over (rsp_set_devices); empty = the calling thread's current device

int rsp_set_devices(const int *ids, int n)
{
    if (n < 0 || (n > 0 && !ids)) return RSP_BAD_INPUT;
    const int count = rsp_device_count();
    for (int k = 0; k < n; ++k)
        if (ids[k] < 0 || ids[k] >= count) return RSP_BAD_INPUT;
    std::lock_guard<std::mutex> lock(g_devices_guard);
    g_devices.assign(ids, ids + n);
    return RSP_OK;
}

int rsp_cholsol_batch_f64(const rsp_csc_f64 *A, const double *Ax_all, double *B, int64_t nsys, int8_t order, int32_t *status)
{
    if (!sane_indices(A) || A->m != A->n || nsys < 0 || (nsys > 0 && (!Ax_all || !B))) return RSP_BAD_INPUT;
    if (nsys == 0 || A->n == 0) return RSP_OK;
    std::vector<int> devs;
    devs = listed_devices();
    if (devs.size() <= 1 || nsys < 2 * (int64_t)devs.size()) {
        int before = -1;
        if (devs.size() == 1) {
            if (hipGetDevice(&before) != hipSuccess || hipSetDevice(devs[0]) != hipSuccess) return RSP_DEVICE_ERROR;
        }
        const int st = cholsol_batch_on_device(A, Ax_all, B, nsys, order, status);
        if (before >= 0) (void)hipSetDevice(before);  // the caller's thread keeps the device it came with
        return st;
    }
    // independent systems: contiguous shares, one host thread per device, no exchange between them (every device
    // repeats the ordering and the analysis for itself: they depend on the pattern alone and run side by side)
    const int64_t G = (int64_t)devs.size(), n = A->n, nnz = A->p[n];
    std::vector<int> rc((size_t)G, RSP_OK);
    std::vector<int32_t> local;
    int32_t *st = status;
    if (!st) {
        local.assign((size_t)nsys, 0);
        st = local.data();
    }
    std::vector<std::thread> pool;
    for (int64_t g = 0; g < G; ++g) {
        const int64_t lo = nsys * g / G, hi = nsys * (g + 1) / G;
        pool.emplace_back([&, g, lo, hi] {
            if (hipSetDevice(devs[g]) != hipSuccess) { rc[g] = RSP_DEVICE_ERROR; return; }
            rc[g] = cholsol_batch_on_device(A, Ax_all + lo * nnz, B + lo * n, hi - lo, order, st + lo);
        });
    }
    for (auto &t : pool) t.join();
    for (int64_t g = 0; g < G; ++g)
        if (rc[g] < 0) return rc[g];
    if (!status)
        for (int64_t s = 0; s < nsys; ++s)
            if (st[s] != RSP_OK) return RSP_NOT_POSITIVE_DEFINITE;
    return RSP_OK;
}

// ---- LU -----------------------------------------------------------------------------------------

int rsp_lu_f64(const rsp_csc_f64 *A, const rsp_symb *S, double tol, rsp_lu **N)
{
    if (!sane_indices(A) || !S || !N || A->m != A->n || S->s.n != A->n) return RSP_BAD_INPUT;
    *N = nullptr;
    return guarded([&] {
        std::unique_ptr<rsp_lu> h(new rsp_lu);
        const int64_t *q = S->s.has_q ? S->s.q.data() : nullptr;
        // the numeric part on the GPU when every column keeps its diagonal as pivot (checked there) and the columns' dependency
        // levels are wide enough to pay (RSP_LU_GPU: 0 never, 1 whenever applicable); bit-identical to the host's factors, which
        // are what every other case gets
        const char *mode = getenv("RSP_LU_GPU");
        int st = lu_numeric_device(view_of(A), q, tol, h->f, mode && *mode ? atoi(mode) : 2, &h->device);
        h->on_device = st == kOk;
        if (q) h->q_factor.assign(q, q + A->n);
        h->pattern = pattern_hash(view_of(A));
        h->pattern_nnz = A->p[A->n];
        if (st == kLuDeclined) st = sparse_lu(view_of(A), q, tol, h->f);
        if (st) return st;
        *N = h.release();
        return (int)RSP_OK;
    });
}

static void fill_view(const Csc &M, rsp_csc_f64 *out)
{
    out->nzmax = M.p[M.n];
    out->m = M.m;
    out->n = M.n;
    out->p = M.p.data();
    out->i = reinterpret_cast<const uint64_t *>(M.i.data());
    out->x = M.x.data();
}

// the host copy of the factors' values, after a refactorisation that left them on the device (caller holds the guard)
static void lu_pull_values(rsp_lu *N)
{
    if (!N->values_on_device_only) return;
    N->device->download(N->f);
    N->values_on_device_only = false;
}
static int lu_host_view(const rsp_lu *N_, bool lower, rsp_csc_f64 *out)
{
    if (!N_ || !out) return RSP_BAD_INPUT;
    rsp_lu *N = const_cast<rsp_lu *>(N_);  // (the handle is logically unchanged: its host copy catches up with the device)
    return guarded([&] {
        std::lock_guard<std::mutex> lock(N->guard.m);
        if (!N->valid) return (int)RSP_NO_PIVOT;
        lu_pull_values(N);
        fill_view(lower ? N->f.L : N->f.U, out);
        return (int)RSP_OK;
    });
}
int rsp_lu_L(const rsp_lu *N, rsp_csc_f64 *L) { return lu_host_view(N, true, L); }
int rsp_lu_U(const rsp_lu *N, rsp_csc_f64 *U) { return lu_host_view(N, false, U); }
const int64_t *rsp_lu_pinv(const rsp_lu *N) { return N ? N->f.pinv.data() : nullptr; }
int rsp_lu_static_pattern_f64(const rsp_csc_f64 *A, const rsp_symb *S, rsp_lu **N, int64_t *nlevels)
{
    if (!sane_indices(A) || !S || !N || A->m != A->n || S->s.n != A->n) return RSP_BAD_INPUT;
    *N = nullptr;
    return guarded([&] {
        std::unique_ptr<rsp_lu> h(new rsp_lu);
        std::vector<int32_t> level;
        if (!lu_static_pattern(view_of(A), S->s.has_q ? S->s.q.data() : nullptr, h->f, level)) return (int)RSP_BAD_INPUT;
        int32_t depth = 0;
        for (int32_t l : level) depth = std::max(depth, l + 1);
        if (nlevels) *nlevels = depth;
        *N = h.release();
        return (int)RSP_OK;
    });
}
int rsp_lu_on_device(const rsp_lu *N) { return N && N->on_device ? 1 : 0; }

// New values on the pattern the handle was made with (a refactorisation; no counterpart in the reference, which runs lu
// again).  With the device half present the kernel alone runs -- no depth-first search, nothing analysed again, the
// solve programmes receive the new coefficients in place -- and the factors are again bit-identical to rsparse::lu's.
// When a column no longer keeps its diagonal as pivot, or the handle came from the host routine, the host routine
// factors the new matrix (the pivot sequence may then differ; solves are analysed again at their next use).
int rsp_lu_refactor_f64(rsp_lu *N, const rsp_csc_f64 *A, double tol)
{
    if (!N || !sane_indices(A) || A->m != A->n || A->n != N->f.L.n) return RSP_BAD_INPUT;
    return guarded([&] {
        std::lock_guard<std::mutex> lock(N->guard.m);
        const CscView Av = view_of(A);
        if (N->pattern_nnz != Av.nnz() || N->pattern != pattern_hash(Av)) return (int)RSP_BAD_INPUT;
        if (N->device) {
            RSP_HIP(hipDeviceSynchronize());  // solves still in flight read the coefficients that are about to change
            if (N->device->factor(A->x, tol) == kOk) {
                N->values_on_device_only = true;  // (downloaded when the host view, a host refill or a new analysis asks for them)
                if (N->lower) {
                    // resident solve programmes take the new coefficients in place: on the device, straight from the values
                    // lu_columns left there (refill kernels), else on the host through the same recipe, else they are dropped
                    // and analysed again at the next solve
                    const bool on_device = !(getenv("RSP_REFILL_DEVICE") && getenv("RSP_REFILL_DEVICE")[0] == '0') &&
                                           N->lower->refresh_device(N->f.L.view(), N->device->dLx(), nullptr) &&
                                           N->upper->refresh_device(N->f.U.view(), N->device->dUx(), nullptr);
                    if (on_device) {
                        RSP_HIP(hipDeviceSynchronize());
                    } else {
                        lu_pull_values(N);
                        if (N->lower->refresh_host(N->f.L.view()) && N->upper->refresh_host(N->f.U.view())) {
                            N->lower->refresh_upload();
                            N->upper->refresh_upload();
                        } else {
                            N->lower.reset();
                            N->upper.reset();
                        }
                    }
                }
                return (int)RSP_OK;
            }
            N->device.reset();
        }
        // the host routine, with the column order of the first factorisation
        if (N->lower) RSP_HIP(hipDeviceSynchronize());
        LuFactors f;
        const int st = sparse_lu(Av, N->q_factor.empty() ? nullptr : N->q_factor.data(), tol, f);
        N->values_on_device_only = false;
        N->on_device = false;
        N->lower.reset();
        N->upper.reset();
        N->d_pinv = DeviceBuffer<int64_t>();
        N->valid = st == kOk;  // (no pivot: what the handle held is gone with the programmes, and nothing replaces it)
        if (st) return st;
        N->f = std::move(f);
        return (int)RSP_OK;
    });
}

// the two triangular handles of a factorisation, built on its first solve (caller holds the guard)
static int lu_solvers(rsp_lu *N)
{
    if (N->lower) return RSP_OK;
    lu_pull_values(N);  // (the analysis reads the values from the host copy)
    std::unique_ptr<TriSolver> lo(new TriSolver), up(new TriSolver);
    if (N->device && !N->single_use) {  // a refactorisation may follow: the programmes keep the recipe for new coefficients
        lo->keep_for_refill();
        up->keep_for_refill();
    }
    if (int st = analyze_pair(*lo, N->f.L.view(), RSP_LSOLVE, *up, N->f.U.view(), RSP_USOLVE, N->single_use)) return st;
    N->lower = std::move(lo);
    N->upper = std::move(up);
    return RSP_OK;
}

int rsp_lu_solve_f64(rsp_lu *N, const rsp_symb *S, double *B, int64_t ldb, int64_t nrhs)
{
    if (!N || !S || !B || nrhs < 0 || ldb < N->f.L.n || S->s.n != N->f.L.n) return RSP_BAD_INPUT;
    return guarded([&] {
        if (int st = require_device()) return st;
        std::lock_guard<std::mutex> lock(N->guard.m);
        if (!N->valid) return (int)RSP_NO_PIVOT;
        const int64_t n = N->f.L.n;
        if (int st = lu_solvers(N)) return st;
        // x = P b (host gather, 16 MB per million unknowns), L and U solves on the GPU, b = Q x
        std::vector<double> x((size_t)n * (size_t)nrhs);
        scatter_rows(n, N->f.pinv.data(), B, x.data(), ldb, nrhs);
        DeviceBuffer<double> d(x.size());
        d.upload(x);
        N->lower->solve_device(d.get(), RSP_COL_MAJOR, n, nrhs, nullptr);
        N->upper->solve_device(d.get(), RSP_COL_MAJOR, n, nrhs, nullptr);
        RSP_HIP(hipDeviceSynchronize());
        N->lower->check_rows_run();  // (single-use handles: the flagged row kernel reports an abandoned wait here)
        N->upper->check_rows_run();
        d.download(x.data(), x.size());
        const int64_t *q = S->s.has_q ? S->s.q.data() : nullptr;
        for (int64_t r = 0; r < nrhs; ++r)
            for (int64_t k = 0; k < n; ++k) B[(q ? q[k] : k) + r * ldb] = x[k + r * n];
        return (int)RSP_OK;
    });
}

int rsp_lu_solve_dev_f64(rsp_lu *N, const rsp_symb *S, double *dB, int layout, int64_t ld, int64_t nrhs, void *stream_)
{
    if (!N || !S || !dB || nrhs < 0 || nrhs > INT32_MAX || S->s.n != N->f.L.n || (layout != RSP_COL_MAJOR && layout != RSP_ROW_MAJOR))
        return RSP_BAD_INPUT;
    if (layout == RSP_COL_MAJOR ? ld < N->f.L.n : ld < nrhs) return RSP_BAD_INPUT;
    if (nrhs == 0 || N->f.L.n == 0) return RSP_OK;
    return guarded([&] {
        std::lock_guard<std::mutex> lock(N->guard.m);
        if (!N->valid) return (int)RSP_NO_PIVOT;
        hipStream_t stream = (hipStream_t)stream_;
        const int64_t n = N->f.L.n;
        if (int st = lu_solvers(N)) return st;
        if (!N->d_pinv.size()) N->d_pinv.upload(N->f.pinv);
        // the column permutation comes with the call (as in rsp_lu_solve_f64): the device copy follows it
        if (S->s.has_q && (N->q_host.size() != (size_t)n || !std::equal(N->q_host.begin(), N->q_host.end(), S->s.q.begin()))) {
            RSP_HIP(hipDeviceSynchronize());  // (an earlier solve on another stream may still read the old copy)
            N->q_host.assign(S->s.q.begin(), S->s.q.begin() + n);
            N->d_q.upload(N->q_host.data(), (size_t)n);
        }
        N->guard.begin(stream);
        // x(pinv[k]) = b[k]  (ipvec, lib.rs:678), into a row-major workspace whatever the caller's layout
        const int64_t ldw = layout == RSP_COL_MAJOR ? nrhs : ld;
        N->work.resize((size_t)n * (size_t)ldw);
        double *W = N->work.get();
        const dim3 tgrid((unsigned)((n + 31) / 32), (unsigned)((nrhs + 31) / 32)), tblock(256);
        const dim3 pgrid((unsigned)((n + 3) / 4)), pblock(256);
        if (layout == RSP_COL_MAJOR)
            hipLaunchKernelGGL(dev::colmajor_to_rowmajor, tgrid, tblock, 0, stream, dB, ld, W, nrhs, n, (int)nrhs, N->d_pinv.get(), 1);
        else
            hipLaunchKernelGGL(dev::permute_rows, pgrid, pblock, 0, stream, dB, W, ld, n, (int)nrhs, N->d_pinv.get(), 1);
        solve_pair_device(*N->lower, *N->upper, W, ldw, nrhs, stream);  // lsolve, usolve (lib.rs:679-680)
        // b(q[k]) = x[k]  (ipvec with q, lib.rs:681); no q = identity
        const int64_t *q = S->s.has_q ? N->d_q.get() : nullptr;
        if (layout == RSP_COL_MAJOR)
            hipLaunchKernelGGL(dev::rowmajor_to_colmajor, tgrid, tblock, 0, stream, W, nrhs, dB, ld, n, (int)nrhs, q, 1);
        else
            hipLaunchKernelGGL(dev::permute_rows, pgrid, pblock, 0, stream, W, dB, ld, n, (int)nrhs, q, 1);
        RSP_HIP(hipGetLastError());
        N->guard.end(stream);
        return (int)RSP_OK;
    });
}

void rsp_lu_free(rsp_lu *N) { delete N; }

// ---- QR: the consumer of usolve / utsolve (qrsol, lib.rs:927-956) ------------------------------------------------
int rsp_qr_f64(const rsp_csc_f64 *A, const rsp_symb *S, rsp_qr **N)
{
    if (!sane_indices(A) || !S || !N || S->s.n != A->n) return RSP_BAD_INPUT;
    *N = nullptr;
    return guarded([&] {
        std::unique_ptr<rsp_qr> h(new rsp_qr);
        h->m = A->m;
        h->n = A->n;
        if (int st = sparse_qr(view_of(A), S->s, h->f)) return st;
        *N = h.release();
        return (int)RSP_OK;
    });
}

int rsp_qr_V(const rsp_qr *N, rsp_csc_f64 *V)
{
    if (!N || !V) return RSP_BAD_INPUT;
    fill_view(N->f.V, V);
    return RSP_OK;
}
int rsp_qr_R(const rsp_qr *N, rsp_csc_f64 *R)
{
    if (!N || !R) return RSP_BAD_INPUT;
    fill_view(N->f.R, R);
    return RSP_OK;
}
const double *rsp_qr_beta(const rsp_qr *N) { return N ? N->f.beta.data() : nullptr; }

// x <- R^-1 x (kind 2, usolve) or R^-T x (kind 3, utsolve) on the GPU; R is m2 x n with every row index below n, so the
// n x n view is the triangular matrix the solve needs (usolve only touches x[0, n), lib.rs:1230-1237)
static int qr_triangular(rsp_qr *N, int kind, double *x, bool single_use)
{
    std::unique_ptr<TriSolver> &T = N->upper[kind - RSP_USOLVE];
    if (!T) {
        const Csc &R = N->f.R;
        std::unique_ptr<TriSolver> t(new TriSolver);
        if (int st = t->analyze(CscView{R.n, R.n, R.p.data(), R.i.data(), R.x.data()}, kind)) return st;
        if (single_use) t->set_single_use();
        T = std::move(t);
    }
    T->solve_host(x, N->n, 1);
    return RSP_OK;
}

int rsp_qr_solve_f64(rsp_qr *N, const rsp_symb *S, double *b, int transposed)
{
    if (!N || !S || !b || S->s.n != N->n || S->s.m2 != N->f.V.m || (int64_t)S->s.pinv.size() != 2 * N->m + N->n) return RSP_BAD_INPUT;
    if (N->m < N->n) return RSP_BAD_INPUT;  // the factored matrix must have at least as many rows as columns (qrsol factors A' otherwise)
    return guarded([&] {
        if (int st = require_device()) return st;
        std::lock_guard<std::mutex> lock(N->guard.m);
        const int64_t m = N->m, n = N->n;
        const int64_t *rowperm = S->s.pinv.data(), *q = S->s.has_q ? S->s.q.data() : nullptr;
        std::vector<double> x((size_t)S->s.m2, 0.0);
        if (!transposed) {  // least squares with the factored matrix itself, lib.rs:931-942
            for (int64_t i = 0; i < m; ++i) x[rowperm[i]] = b[i];          // ipvec(m, pinv, b, x)
            qr_apply_reflections(N->f, false, x.data());                    // happly, k ascending
            if (int st = qr_triangular(N, RSP_USOLVE, x.data(), false)) return st;  // x = R \ x
            for (int64_t k = 0; k < n; ++k) b[q ? q[k] : k] = x[k];         // ipvec(n, q, x, b)
        } else {  // the handle factors A' (m x n here = n_A x m_A): minimum-norm solution of A y = b, lib.rs:943-955
            for (int64_t k = 0; k < n; ++k) x[k] = b[q ? q[k] : k];         // pvec(m_A, q, b, x)
            if (int st = qr_triangular(N, RSP_UTSOLVE, x.data(), false)) return st;  // x = R' \ x
            qr_apply_reflections(N->f, true, x.data());                     // happly, k descending
            for (int64_t i = 0; i < m; ++i) b[i] = x[rowperm[i]];           // pvec(n_A, pinv, x, b)
        }
        return (int)RSP_OK;
    });
}

void rsp_qr_free(rsp_qr *N) { delete N; }

int rsp_qrsol_f64(const rsp_csc_f64 *A, double *b, int8_t order)
{
    if (!sane_indices(A) || !b) return RSP_BAD_INPUT;
    rsp_symb *S = nullptr;
    rsp_qr *N = nullptr;
    int st;
    if (A->m >= A->n) {
        st = rsp_sqr_qr(A, order, &S);
        if (st == RSP_OK) st = rsp_qr_f64(A, S, &N);
        if (st == RSP_OK) st = rsp_qr_solve_f64(N, S, b, 0);
    } else {
        Csc At;
        st = guarded([&] {
            At = transpose_csc(view_of(A));
            return (int)RSP_OK;
        });
        rsp_csc_f64 at{(int64_t)At.x.size(), At.m, At.n, At.p.data(), reinterpret_cast<const uint64_t *>(At.i.data()), At.x.data()};
        if (st == RSP_OK) st = rsp_sqr_qr(&at, order, &S);
        if (st == RSP_OK) st = rsp_qr_f64(&at, S, &N);
        if (st == RSP_OK) st = rsp_qr_solve_f64(N, S, b, 1);
    }
    rsp_qr_free(N);
    rsp_symb_free(S);
    return st;
}

// background job after a lusol call whose factorisation ran on the device: the handle becomes a multi-use one (blocked solve
// programmes with their refill recipes, resident after one solve of a zero vector) and enters the analysis cache
extern "C++" {
namespace {
void lu_cache_prepare(std::unique_ptr<CholCacheEntry> e)
{
    CholCache &C = chol_cache();
    const uint64_t hash = e->hash;
    bool ok = false;
    try {
        if (hipSetDevice(e->device) == hipSuccess && e->N->device) {
            rsp_lu *N = e->N.get();
            {
                std::lock_guard<std::mutex> lock(N->guard.m);
                RSP_HIP(hipDeviceSynchronize());
                N->single_use = false;
                N->lower.reset();
                N->upper.reset();
            }
            std::vector<double> zero((size_t)e->n, 0.0);
            ok = rsp_lu_solve_f64(N, e->S.get(), zero.data(), e->n, 1) == RSP_OK;
            if (ok) {  // the refill recipes go to the device now, not at the first hit
                std::lock_guard<std::mutex> lock(N->guard.m);
                if (N->lower && N->upper) {
                    N->lower->prepare_refresh_device(N->f.L.view());
                    N->upper->prepare_refresh_device(N->f.U.view());
                }
            }
            e->bytes = (int64_t)(N->f.L.i.size() + N->f.U.i.size()) * 40 + N->device->device_bytes();
        }
    } catch (...) {
        ok = false;
    }
    std::vector<std::unique_ptr<CholCacheEntry>> dead;
    {
        std::lock_guard<std::mutex> lock(C.m);
        C.pending.erase(std::remove(C.pending.begin(), C.pending.end(), hash), C.pending.end());
        if (ok && e->bytes <= chol_cache_budget() && chol_cache_capacity() > 0) {
            e->stamp = ++C.clock;
            C.entries.push_back(std::move(e));
            C.inserted++;
            dead = chol_cache_trim(C, chol_cache_capacity(), chol_cache_budget());
        }
    }
    dead.clear();
    e.reset();
}
}  // namespace
}  // extern "C++"

// rsparse::lusol (lib.rs:672-683).  Like rsp_cholsol_f64 it keeps what a call on the same pattern can reuse -- here only when
// the factorisation ran on the device (static diagonal pivots): the patterns of L and U then depend on the pattern of A alone,
// so a later call with the same (device, n, order, pattern) is rsp_lu_refactor_f64 (the kernel alone, the pivot rule checked
// again with this call's tol) + a solve through the resident programmes.  A hit whose values no longer keep their diagonal
// pivots is factored by the host routine, as any call would be, and its entry leaves the cache.
int rsp_lusol_f64(const rsp_csc_f64 *A, double *b, int8_t order, double tol)
{
    if (!sane_indices(A) || !b || A->m != A->n) return RSP_BAD_INPUT;
    const int64_t capacity = chol_cache_capacity();
    uint64_t hash = 0;
    int device = 0;
    bool may_insert = false;
    if (capacity > 0 && A->n >= chol_cache_min_n()) {
        int st = guarded([&] {
            if (int rd = require_device()) return rd;
            RSP_HIP(hipGetDevice(&device));
            return (int)RSP_OK;
        });
        if (st != RSP_OK) return st;
        hash = pattern_hash(view_of(A));
        CholCache &C = chol_cache();
        std::unique_ptr<CholCacheEntry> hit;
        {
            std::lock_guard<std::mutex> lock(C.m);
            for (size_t k = 0; k < C.entries.size(); ++k) {
                CholCacheEntry &e = *C.entries[k];
                if (e.kind == 1 && e.hash == hash && e.device == device && e.order == order && e.n == A->n && e.nnz == A->p[A->n]) {
                    hit = std::move(C.entries[k]);  // out of the cache while this call uses it
                    C.entries.erase(C.entries.begin() + (long)k);
                    break;
                }
            }
        }
        if (hit && !same_pattern(*hit, A)) {  // a hash collision: not this pattern, back it goes
            std::lock_guard<std::mutex> lock(C.m);
            C.entries.push_back(std::move(hit));
            hit.reset();
        }
        if (hit) {
            st = rsp_lu_refactor_f64(hit->N.get(), A, tol);
            if (st == RSP_OK) st = rsp_lu_solve_f64(hit->N.get(), hit->S.get(), b, A->n, 1);
            std::vector<std::unique_ptr<CholCacheEntry>> dead;
            {
                std::lock_guard<std::mutex> lock(C.m);
                C.hits++;
                if (st == RSP_OK && hit->N->device) {
                    hit->stamp = ++C.clock;
                    C.entries.push_back(std::move(hit));
                    dead = chol_cache_trim(C, capacity, chol_cache_budget());
                } else {
                    C.evicted++;
                }
            }
            dead.clear();
            hit.reset();  // (an entry that went to the host routine is no better than a fresh call)
            return st;
        }
        {
            std::lock_guard<std::mutex> lock(C.m);
            C.misses++;
            may_insert = std::find(C.pending.begin(), C.pending.end(), hash) == C.pending.end() && C.pending.size() < 2;
        }
    }
    rsp_symb *S = nullptr;
    rsp_lu *N = nullptr;
    const double t0 = now_s();
    int st = rsp_sqr_lu(A, order, &S);               // ordering (host)
    const double t1 = now_s();
    if (st == RSP_OK) st = rsp_lu_f64(A, S, tol, &N);  // numeric LU: GPU when the diagonal pivots hold, else host; same bits
    const double t2 = now_s();
    if (st == RSP_OK) N->single_use = true;
    if (st == RSP_OK) st = rsp_lu_solve_f64(N, S, b, A->n, 1);  // lsolve + usolve (GPU)
    if (getenv("RSP_TIMING"))
        fprintf(stderr, "lusol: sqr %.2fs, lu %.2fs (%s), solve (analysis + solve) %.2fs\n", t1 - t0, t2 - t1, N && N->device ? "device" : "host", now_s() - t2);
    if (st == RSP_OK && may_insert && N->device) {
        // keep the factorisation's device half and its patterns for the next call on this pattern
        CholCache &C = chol_cache();
        std::unique_ptr<CholCacheEntry> e;
        bool claimed = false;
        try {
            e.reset(new CholCacheEntry);
            e->kind = 1;
            e->hash = hash;
            e->device = device;
            e->order = order;
            e->n = A->n;
            e->nnz = A->p[A->n];
            e->p.assign(A->p, A->p + A->n + 1);
            e->i.assign(reinterpret_cast<const int64_t *>(A->i), reinterpret_cast<const int64_t *>(A->i) + e->nnz);
            std::lock_guard<std::mutex> lock(C.m);
            if (std::find(C.pending.begin(), C.pending.end(), hash) == C.pending.end()) {
                C.pending.push_back(hash);
                claimed = true;
            }
        } catch (...) {
            claimed = false;
        }
        if (claimed) {
            e->S.reset(S);
            e->N.reset(N);
            CholCacheEntry *raw = e.release();
            std::thread worker;
            try {
                worker = std::thread([raw] { lu_cache_prepare(std::unique_ptr<CholCacheEntry>(raw)); });
            } catch (...) {  // no thread: the entry is dropped (with S and N), the pattern is free to be claimed again
                delete raw;
                std::lock_guard<std::mutex> lock(C.m);
                C.pending.erase(std::remove(C.pending.begin(), C.pending.end(), hash), C.pending.end());
                return st;
            }
            std::lock_guard<std::mutex> lock(C.m);
            C.workers.push_back(std::move(worker));
            if (C.workers.size() > 4) {  // the oldest job has long finished (at most two run at a time): collect it
                std::thread old = std::move(C.workers.front());
                C.workers.erase(C.workers.begin());
                if (old.joinable()) old.join();
            }
            return st;  // S and N belong to the cache now
        }
    }
    rsp_lu_free(N);
    rsp_symb_free(S);
    return st;
}

}  // extern "C"
