// TriSolver: uploads a triangular factor in execution order and runs the level-scheduled kernels.
#include <cstdio>
#include <algorithm>
#include <chrono>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <cstdlib>

#include "engine.hpp"
#include "solve_kernels.hip"
#include "blocked_kernels.hip"
#include "item_kernels.hip"
#include "cluster_kernels.hip"
#include "refill_kernels.hip"

namespace rsp {

static double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int TriSolver::analyze(const CscView &T, int kind)
{
    const double t0 = now_s();
    const int st = build_gather_form(T, kind, G_);
    if (st != kOk) return st;
    const double t1 = now_s();
    kind_ = kind;
    n_ = G_.n;
    nnz_ = (i64)G_.idx.size() + n_;
    // canonical level sets (reported through rsp_tri_levels for bit-exact comparison)
    Schedule S = build_schedule(G_, 0);
    level_of_row_ = std::move(S.level_of_row);
    num_row_levels_ = S.num_row_levels;
    if (getenv("RSP_TIMING")) fprintf(stderr, "tri analyze: gather form %.2fs, level sets %.2fs\n", t1 - t0, now_s() - t1);
    return kOk;
}

TriProgramHost TriSolver::stage_program(i64 max_task_entries) const
{
    if (host_values_stale_) throw DeviceError("the level programme needs the factor's values on the host: refreshed on the device only");
    const Schedule S = build_schedule(G_, max_task_entries);
    TriProgramHost H;
    const i64 ntasks = (i64)S.task_first.size();
    H.task_ptr.assign(ntasks + 1, 0);
    H.op_row.resize(n_);
    H.op_ptr.assign(n_ + 1, 0);
    H.ent_idx.resize(G_.idx.size());
    H.op_diag.resize(n_);
    H.ent_val.resize(G_.val.size());
    i64 op = 0, ent = 0;
    for (i64 t = 0; t < ntasks; ++t) {
        H.task_ptr[t] = (int32_t)op;
        for (i64 s = 0; s < S.task_rows[t]; ++s) {
            const i64 r = G_.descending ? S.task_first[t] - s : S.task_first[t] + s;
            H.op_row[op] = (int32_t)r;
            H.op_diag[op] = G_.diag[r];
            H.op_ptr[op] = (int32_t)ent;
            const i64 len = G_.ptr[r + 1] - G_.ptr[r];
            std::copy(G_.idx.begin() + G_.ptr[r], G_.idx.begin() + G_.ptr[r + 1], H.ent_idx.begin() + ent);
            std::copy(G_.val.begin() + G_.ptr[r], G_.val.begin() + G_.ptr[r + 1], H.ent_val.begin() + ent);
            ent += len;
            ++op;
        }
    }
    H.task_ptr[ntasks] = (int32_t)op;
    H.op_ptr[n_] = (int32_t)ent;
    H.ntasks = ntasks;
    H.nlevels = (i64)S.level_ptr.size() - 1;
    H.longest_task = S.longest_task;
    for (i64 l = 0; l < H.nlevels; ++l) H.levels.emplace_back((int)S.level_ptr[l], (int)(S.level_ptr[l + 1] - S.level_ptr[l]));
    return H;
}

static void upload_program(TriProgramHost &H, TriProgram &P)
{
    P.ntasks = H.ntasks;
    P.nlevels = H.nlevels;
    P.longest_task = H.longest_task;
    P.levels = std::move(H.levels);
    P.task_ptr.upload(H.task_ptr);
    P.op_row.upload(H.op_row);
    P.op_ptr.upload(H.op_ptr);
    P.op_diag.upload(H.op_diag);
    P.ent_idx.upload(H.ent_idx);
    P.ent_val.upload(H.ent_val);
}

void TriSolver::set_values(const CscView &T)
{
    fill_gather_values(T, kind_, G_);
}

void TriSolver::build_program(i64 max_task_entries, TriProgram &P)
{
    TriProgramHost H = stage_program(max_task_entries);
    upload_program(H, P);
}

static dev::TriArrays arrays_of(const TriProgram &P)
{
    return dev::TriArrays{P.task_ptr.get(), P.op_row.get(), P.op_ptr.get(), P.op_diag.get(), P.ent_idx.get(), P.ent_val.get()};
}

void TriSolver::run_multi(double *X, i64 ld, i64 nrhs, hipStream_t stream)
{
    if (!have_multi_) {
        build_program(kTaskEntriesMulti, multi_);
        have_multi_ = true;
    }
    const dev::TriArrays A = arrays_of(multi_);
    const int wpb = 4;  // wavefronts per 256-thread block
    for (const auto &lv : multi_.levels) {
        // widest per-lane vector that still gives the chip enough wavefronts for this level
        int V = 1;
        if (nrhs % 4 == 0 && (i64)lv.second * ((nrhs + 255) / 256) >= 2048) V = 4;
        else if (nrhs % 2 == 0 && (i64)lv.second * ((nrhs + 127) / 128) >= 2048) V = 2;
        const int nchunks = (int)((nrhs + 64 * V - 1) / (64 * V));
        const i64 waves = (i64)lv.second * nchunks;
        const dim3 grid((unsigned)((waves + wpb - 1) / wpb)), block(64 * wpb);
        if (V == 4)
            hipLaunchKernelGGL(dev::solve_level_multi<4>, grid, block, 0, stream, A, lv.first, lv.second, X, ld, (int)nrhs, nchunks);
        else if (V == 2)
            hipLaunchKernelGGL(dev::solve_level_multi<2>, grid, block, 0, stream, A, lv.first, lv.second, X, ld, (int)nrhs, nchunks);
        else
            hipLaunchKernelGGL(dev::solve_level_multi<1>, grid, block, 0, stream, A, lv.first, lv.second, X, ld, (int)nrhs, nchunks);
    }
    RSP_HIP(hipGetLastError());
}

static i64 env_or(const char *name, i64 fallback)
{
    const char *v = getenv(name);
    return v && *v ? atol(v) : fallback;
}

// host half of build_blocked(): no device call, so two solvers can be staged on two host threads
void TriSolver::stage_blocked()
{
    if (host_values_stale_) throw DeviceError("the programme needs the factor's values on the host: refreshed on the device only");
    staged_.reset(new BlockProgram(build_block_program(G_, env_or("RSP_TASK_BLOCKS", kTaskBlocks), (int)env_or("RSP_CLUSTER_SLOTS", kClusterSlots))));
}

void TriSolver::set_single_use()
{
    // measured at C2 (7 307 row levels, 47.5 M entries): a level launch costs about 6 us, the programme build about 20 ns per
    // entry; the levels win when they are cheaper by a factor of two.  RSP_SINGLE_USE=0 always keeps the blocked
    // programme, =1 always takes the levels (read per call: the tests run both ways in one process).
    // RSP_SINGLE_USE=2: rows with flags in one persistent launch (solve_rows_flagged), which needs no programme at all.
    const char *mode = getenv("RSP_SINGLE_USE");
    single_use_rows_ = false;
    // Since round 3 that kernel is the default (=2 or unset); 0 and 1 force the two older choices.
    const bool rows_possible = n_ < INT32_MAX && !G_.placeholder;
    if (mode && (mode[0] == '0' || mode[0] == '1')) single_use_levels_ = mode[0] == '1';
    else if (rows_possible) single_use_rows_ = true, single_use_levels_ = false;
    else single_use_levels_ = (double)num_row_levels_ * 6e-6 < (double)nnz_ * 1e-8;
    // host half of the level programme right away: analyze_pair() calls this on its worker threads
    if (single_use_levels_ && !have_single_ && !staged_single_) staged_single_.reset(new TriProgramHost(stage_program(kTaskEntriesSingle)));
}

// analyse two factors (or one factor in two directions) side by side on two host threads, programmes included
int analyze_pair(TriSolver &a, const CscView &Ta, int kind_a, TriSolver &b, const CscView &Tb, int kind_b, bool single_use)
{
    int sa = kOk, sb = kOk;
    std::exception_ptr ea, eb;
    auto work = [single_use](TriSolver &t, const CscView &T, int kind, int &st, std::exception_ptr &e) {
        try {
            st = t.analyze(T, kind);
            if (st == kOk && single_use) t.set_single_use();
            if (st == kOk && !t.single_use_levels()) t.stage_blocked();
        } catch (...) {
            e = std::current_exception();
        }
    };
    std::thread other(work, std::ref(b), std::cref(Tb), kind_b, std::ref(sb), std::ref(eb));
    work(a, Ta, kind_a, sa, ea);
    other.join();
    if (ea) std::rethrow_exception(ea);
    if (eb) std::rethrow_exception(eb);
    return sa != kOk ? sa : sb;
}

// what the block kernel of one level runs for ONE column chunk, twin blocks of wide items included, no separate small-block
// launch (the description run_window makes for a narrow block); returns the level's workgroups
static int describe_narrow_level(const BlockedProgram::Level &lv, bool single_block_tasks, dev::LevelDesc &L)
{
    const int n01 = lv.ntasks[0] + lv.ntasks[1] + lv.ntasks[2];
    L.task_begin = lv.task_begin[0];
    L.block_begin[0] = lv.block_begin[3];
    L.block_begin[1] = lv.block_begin[5];
    L.blk0 = lv.block_begin[0];
    L.rec0 = lv.rec_begin[0];
    L.rec_begin[0] = lv.rec_begin[3];
    L.rec_begin[1] = lv.rec_begin[5];
    L.single = single_block_tasks ? 1 : 0;
    L.ntasks = n01;
    L.nblk[0] = lv.ntasks[3] + lv.ntasks[4];
    L.nblk[1] = lv.ntasks[5] + lv.ntasks[6];
    L.wg0 = (n01 + 7) / 8;
    L.wg1 = (L.nblk[0] + 1) / 2;
    L.wg2 = L.nblk[1];
    return L.wg0 + L.wg1 + L.wg2;
}

void TriSolver::build_flag_runs(const BlockProgram &P)
{
    BlockedProgram &D = blocked_;
    D.flag_runs.clear();
    const char *on = getenv("RSP_FLAGGED_LEVELS");
    if (!on || on[0] != '1' || !D.single_block_tasks || n_ >= INT32_MAX) return;
    const i64 lean_single = env_or("RSP_LEAN_WGS_SINGLE", kLeanLaunchWorkgroupsSingle);
    const i64 small_waves = env_or("RSP_SMALL_WAVES", kSmallLaunchWaves);
    // 0: nothing to launch (transparent inside a run), 1: a latency-bound block launch and nothing else, 2: anything else
    const auto kind_of = [&](const BlockedProgram::Level &lv) {
        dev::LevelDesc L;
        const int wgs = describe_narrow_level(lv, true, L);
        if (lv.nclusters == 0 && wgs == 0) return 0;
        if (lv.nclusters || (lv.ntasks[0] > 0 && lv.ntasks[0] >= small_waves)) return 2;
        return (wgs + 7) / 8 * 8 < lean_single ? 1 : 2;
    };
    std::vector<dev::LevelDesc> descs;
    std::vector<int2> tickets;
    std::vector<int32_t> wait_ptr(1, 0), wait_list;
    std::vector<int32_t> owner_x((size_t)n_, -1), owner_s((size_t)std::max<i64>(1, P.scratch_rows), -1);
    const auto owner_of = [&](int32_t id) -> int32_t & { return id >= 0 ? owner_x[(size_t)id] : owner_s[(size_t)(~id)]; };
    // the blocks of ticket (L, item): one, two or up to eight
    const auto blocks_of = [&](const dev::LevelDesc &L, int item, i64 (&b)[8]) {
        int nb = 0;
        if (item < L.wg2) {
            b[nb++] = L.block_begin[1] + item;
        } else if (item < L.wg2 + L.wg1) {
            for (int h = 0; h < 2; ++h) {
                const int q = (item - L.wg2) * 2 + h;
                if (q < L.nblk[0]) b[nb++] = L.block_begin[0] + q;
            }
        } else {
            for (int w = 0; w < 8; ++w) {
                const int t = (item - L.wg2 - L.wg1) * 8 + w;
                if (t < L.ntasks) b[nb++] = L.blk0 + t;
            }
        }
        return nb;
    };
    std::vector<int32_t> stamp;
    for (size_t l0 = 0; l0 < D.levels.size();) {
        if (kind_of(D.levels[l0]) != 1) {
            ++l0;
            continue;
        }
        size_t l1 = l0, last = l0;
        int members = 0;
        for (; l1 < D.levels.size(); ++l1) {
            const int kd = kind_of(D.levels[l1]);
            if (kd == 2) break;
            if (kd == 1) ++members, last = l1;
        }
        l1 = last + 1;
        if (members < 2) {
            l0 = l1;
            continue;
        }
        BlockedProgram::FlagRunHost run;
        run.level_begin = l0;
        run.level_end = l1;
        run.level_off = (i64)descs.size();
        run.ticket_off = (i64)tickets.size();
        run.nlevels = 0;
        for (size_t l = l0; l < l1; ++l) {
            dev::LevelDesc L;
            const int wgs = describe_narrow_level(D.levels[l], true, L);
            if (wgs == 0) continue;
            for (int item = 0; item < wgs; ++item) tickets.push_back(int2{run.nlevels, item});
            descs.push_back(L);
            run.nlevels++;
        }
        run.ntickets = (int)((i64)tickets.size() - run.ticket_off);
        // producers: who writes each row inside this run, then, per ticket, the earlier tickets whose rows it reads
        for (int k = 0; k < run.ntickets; ++k) {
            const int2 tk = tickets[(size_t)(run.ticket_off + k)];
            i64 b[8];
            const int nb = blocks_of(descs[(size_t)(run.level_off + tk.x)], tk.y, b);
            for (int q = 0; q < nb; ++q) {
                const BlockHeader &h = P.hdr[(size_t)b[q]];
                for (int r = 0; r < h.rc; ++r) {
                    const int32_t id = P.src_idx[(size_t)(h.src_off + r)];
                    if (id != BlockProgram::kNone) owner_of(id) = k;
                }
            }
        }
        stamp.assign((size_t)run.ntickets, -1);
        for (int k = 0; k < run.ntickets; ++k) {
            const int2 tk = tickets[(size_t)(run.ticket_off + k)];
            i64 b[8];
            const int nb = blocks_of(descs[(size_t)(run.level_off + tk.x)], tk.y, b);
            for (int q = 0; q < nb; ++q) {
                const BlockHeader &h = P.hdr[(size_t)b[q]];
                for (i64 e = h.rc; e < 2 * (i64)h.rc + h.nsrc; ++e) {  // initial-value rows and sources
                    const int32_t id = P.src_idx[(size_t)(h.src_off + e)];
                    if (id == BlockProgram::kNone) continue;
                    const int32_t o = owner_of(id);
                    // (o >= k: the row is written later in the run or by this ticket -- a padding entry with a zero
                    // coefficient, or the ticket's own initial value)
                    if (o < 0 || o >= k || stamp[(size_t)o] == k) continue;
                    stamp[(size_t)o] = k;
                    wait_list.push_back((int32_t)(run.ticket_off + o));
                }
            }
            wait_ptr.push_back((int32_t)wait_list.size());
        }
        for (int k = 0; k < run.ntickets; ++k) {  // clear the owners for the next run
            const int2 tk = tickets[(size_t)(run.ticket_off + k)];
            i64 b[8];
            const int nb = blocks_of(descs[(size_t)(run.level_off + tk.x)], tk.y, b);
            for (int q = 0; q < nb; ++q) {
                const BlockHeader &h = P.hdr[(size_t)b[q]];
                for (int r = 0; r < h.rc; ++r) {
                    const int32_t id = P.src_idx[(size_t)(h.src_off + r)];
                    if (id != BlockProgram::kNone) owner_of(id) = -1;
                }
            }
        }
        D.flag_runs.push_back(run);
        l0 = l1;
    }
    if (D.flag_runs.empty()) return;
    D.fr_levels.resize(descs.size() * sizeof(dev::LevelDesc));
    RSP_HIP(hipMemcpy(D.fr_levels.get(), descs.data(), descs.size() * sizeof(dev::LevelDesc), hipMemcpyHostToDevice));
    D.fr_tickets.resize(tickets.size() * sizeof(int2));
    RSP_HIP(hipMemcpy(D.fr_tickets.get(), tickets.data(), tickets.size() * sizeof(int2), hipMemcpyHostToDevice));
    D.fr_wait_ptr.upload(wait_ptr);
    wait_list.push_back(0);  // (never empty)
    D.fr_wait_list.upload(wait_list);
    D.fr_flags.resize(tickets.size());
    RSP_HIP(hipMemset(D.fr_flags.get(), 0, tickets.size() * sizeof(int)));
    D.fr_counters.resize(D.flag_runs.size());
    D.fr_failed.resize(1);
    RSP_HIP(hipMemset(D.fr_failed.get(), 0, sizeof(int)));
    D.fr_epoch = 0;
    if (getenv("RSP_TIMING")) {
        i64 lv = 0;
        for (const auto &r : D.flag_runs) lv += r.nlevels;
        fprintf(stderr, "flagged levels: %zu runs holding %lld levels, %zu tickets, %zu waits\n", D.flag_runs.size(), (long long)lv, tickets.size(),
                wait_list.size() - 1);
    }
}

void TriSolver::build_blocked()
{
    const double tb0 = now_s();
    if (!staged_) stage_blocked();
    BlockProgram P = std::move(*staged_);
    staged_.reset();
    const double tb1 = now_s();
    BlockedProgram &D = blocked_;
    D.ntasks = (i64)P.task_ptr.size() - 1;
    D.nlevels = (i64)P.level_ptr.size() - 1;
    D.nblocks = (i64)P.hdr.size();
    D.scratch_rows = P.scratch_rows;
    D.big_chains = P.big_chains;
    D.inverted_rows = P.inverted_rows;
    D.padded_entries = P.padded_entries;
    D.nclusters = (i64)P.clusters.size();
    D.cluster_rows = P.cluster_rows;
    D.cluster_entries = P.cluster_entries;
    D.cluster_slots = P.cluster_slots;
    D.levels.clear();
    D.launches = 0;
    for (i64 l = 0; l < D.nlevels; ++l) {
        BlockedProgram::Level lv;
        constexpr int NC = BlockProgram::kClasses;
        for (int c = 0; c < NC; ++c) {
            lv.task_begin[c] = P.level_class_ptr[NC * l + c];
            lv.ntasks[c] = (int)(P.level_class_ptr[NC * l + c + 1] - P.level_class_ptr[NC * l + c]);
            lv.block_begin[c] = P.task_ptr[lv.task_begin[c]];
            lv.rec_begin[c] = P.block_record[lv.block_begin[c]];
        }
        int any = 0;
        for (int c = 0; c < NC; ++c) any += lv.ntasks[c];
        D.launches += any > 0;
        lv.cluster_begin = P.cluster_level_ptr[l];
        lv.nclusters = (int)(P.cluster_level_ptr[l + 1] - P.cluster_level_ptr[l]);
        lv.item_begin = P.item_level_ptr[l];
        lv.nitems = (int)(P.item_level_ptr[l + 1] - P.item_level_ptr[l]);
        lv.item_ksteps_max = 0;
        for (i64 k = P.item_level_ptr[l]; k < P.item_level_ptr[l + 1]; ++k) lv.item_ksteps_max = std::max(lv.item_ksteps_max, (int)P.items[k].k_steps);
        D.launches += lv.nclusters > 0;
        D.levels.push_back(lv);
    }
    if (getenv("RSP_DUMP_LEVELS")) {  // per level: tasks by class, blocks, 16-source stages of matrix-core blocks, vector entries
        for (i64 l = 0; l < D.nlevels; ++l) {
            i64 nb = 0, stages = 0, vec = 0, vblk = 0, maxst = 0, rows = 0;
            for (i64 t = P.level_ptr[l]; t < P.level_ptr[l + 1]; ++t)
                for (i64 b = P.task_ptr[t]; b < P.task_ptr[t + 1]; ++b) {
                    const BlockHeader &h = P.hdr[b];
                    ++nb;
                    if (h.rc == 16) { stages += h.nsrc / 16; maxst = std::max<i64>(maxst, h.nsrc / 16); }
                    else { vec += (i64)h.nsrc * h.rc; ++vblk; }
                    for (int r = 0; r < h.rc; ++r) rows += P.src_idx[h.src_off + r] != BlockProgram::kNone;
                }
            fprintf(stderr, "L%lld tasks %d/%d+%d/%d+%d/%d+%d blocks %lld rows %lld mfma_stages %lld max %lld vec_blocks %lld vec_entries %lld clusters %d items %d (max %d K-steps)\n",
                    (long long)l, D.levels[l].ntasks[0], D.levels[l].ntasks[1], D.levels[l].ntasks[2], D.levels[l].ntasks[3], D.levels[l].ntasks[4],
                    D.levels[l].ntasks[5], D.levels[l].ntasks[6], (long long)nb, (long long)rows, (long long)stages, (long long)maxst, (long long)vblk,
                    (long long)vec, D.levels[l].nclusters, D.levels[l].nitems, D.levels[l].item_ksteps_max);
        }
    }
    D.single_block_tasks = D.ntasks == D.nblocks;
    build_flag_runs(P);
    D.task_ptr.upload(P.task_ptr);
    D.hdr.upload(P.hdr);
    D.prologue.upload(P.prologue);
    D.items.upload(P.items);
    D.item_masks.upload(P.item_masks);
    D.nitems = (i64)P.items.size();
    D.item_entries = P.item_padded_entries;
    D.single_block_tasks = D.ntasks == D.nblocks;
    // the kernels fetch ids and coefficients a wavefront at a time (clamped lane offsets), never past these ends
    D.src_idx.upload(P.src_idx);
    D.coef.upload(P.coef);
    D.clusters.upload(P.clusters);
    D.cluster_ids.upload(P.cluster_ids);
    D.groups.upload(P.groups);
    RSP_HIP(hipFuncSetAttribute((const void *)dev::solve_clusters, hipFuncAttributeMaxDynamicSharedMemorySize,
                                4 * (std::min(D.cluster_slots, 64) * 512 + 1024)));
    if (getenv("RSP_TIMING")) fprintf(stderr, "tri programme: build %.2fs, upload %.2fs\n", tb1 - tb0, now_s() - tb1);
    kept_.reset();
    if (G_.keep_src && P.refill.valid) {
        kept_.reset(new BlockProgram);
        kept_->coef = std::move(P.coef);
        kept_->groups = std::move(P.groups);
        kept_->refill = std::move(P.refill);
    }
}

void TriSolver::stage_blocked_pattern()
{
    if (G_.src.empty()) return;
    staged_.reset(new BlockProgram(build_block_program(G_, env_or("RSP_TASK_BLOCKS", kTaskBlocks), (int)env_or("RSP_CLUSTER_SLOTS", kClusterSlots))));
    if (!staged_->refill.valid) staged_.reset();
    host_values_stale_ = true;  // the gather form holds no values at all yet
}

bool TriSolver::adopt_staged()
{
    if (have_blocked_) return kept_ != nullptr;  // already uploaded
    if (!staged_) return false;
    build_blocked();
    have_blocked_ = true;
    return kept_ != nullptr;
}

bool TriSolver::refresh_host(const CscView &T)
{
    if (!have_blocked_ || !kept_ || G_.src.empty() || T.n != n_ || T.nnz() + 0 != (i64)G_.src.size() + n_) return false;
    fill_gather_values(T, kind_, G_);
    host_values_stale_ = false;
    return refill_block_program(G_, *kept_);
}

void TriSolver::build_device_recipe(const CscView &T)
{
    DeviceRecipe &D = recipe_;
    D.built = true;
    D.usable = false;
    const BlockProgram::Refill &R = kept_->refill;
    const i64 nnz = (i64)G_.src.size(), n = n_;
    const bool lower_storage = kind_ == 0 || kind_ == 1;
    // code -> entry of the factor: the gather form's sources, then the diagonals
    std::vector<int32_t> map((size_t)(nnz + n));
    std::copy(G_.src.begin(), G_.src.end(), map.begin());
    for (i64 r = 0; r < n; ++r) map[(size_t)(nnz + r)] = (int32_t)(lower_storage ? T.p[r] : T.p[r + 1] - 1);
    // group records: first entry of the gather form and row of each
    const i64 ngroups = (i64)kept_->groups.size();
    std::vector<int32_t> first((size_t)ngroups, -1), row((size_t)ngroups, 0);
    for (const BlockProgram::Refill::Cluster &c : R.clusters) {
        i64 g = c.grp_off;
        for (i64 k = 0; k <= c.hi - c.lo; ++k) {
            const i64 r = G_.descending ? c.hi - k : c.lo + k;
            const i64 pb = G_.ptr[r], ng = std::max<i64>(1, (G_.ptr[r + 1] - pb + 3) / 4);
            for (i64 gg = 0; gg < ng; ++gg, ++g) {
                if (g >= ngroups) return;
                first[g] = (int32_t)std::min<i64>(pb + 4 * gg, std::max<i64>(nnz - 1, 0));
                row[g] = (int32_t)r;
            }
        }
    }
    for (int32_t f : first)
        if (f < 0) return;  // a record no cluster owns: not a programme this recipe describes
    // chains: descriptors, and the entries of the gather form that fall inside each diagonal block
    std::vector<dev::ChainDesc> chains;
    std::vector<int32_t> pos, code;
    int tiles = 0;
    std::vector<char> seen;
    for (const BlockProgram::Refill::Chain &c : R.chains) {
        const i64 w = c.t1 - c.t0;
        chains.push_back(dev::ChainDesc{c.pool_off, (int32_t)w, tiles});
        tiles += (int)((w + 63) / 64);
        seen.assign((size_t)w, 0);
        for (i64 t = c.t0; t < c.t1; ++t) {
            const i64 r = G_.descending ? n - 1 - t : t;
            std::fill(seen.begin(), seen.end(), 0);
            pos.push_back((int32_t)(c.pool_off + (t - c.t0) * w + (t - c.t0)));
            code.push_back((int32_t)(nnz + r));
            seen[(size_t)(t - c.t0)] = 1;
            for (i64 p = G_.ptr[r]; p < G_.ptr[r + 1]; ++p) {
                const i64 ps = G_.descending ? n - 1 - G_.idx[p] : G_.idx[p];
                if (ps < c.t0) continue;
                if (seen[(size_t)(ps - c.t0)]) return;  // duplicate entries add up: the host refill handles those
                seen[(size_t)(ps - c.t0)] = 1;
                pos.push_back((int32_t)(c.pool_off + (t - c.t0) * w + (ps - c.t0)));
                code.push_back((int32_t)p);
            }
        }
    }
    D.map.upload(map);
    D.coef_src.upload(R.coef_src);
    D.grp_first.upload(first);
    D.grp_row.upload(row);
    D.tri16.upload(R.tri16);
    D.chain_pos.upload(pos);
    D.chain_code.upload(code);
    D.chains.resize(std::max<size_t>(1, chains.size() * sizeof(dev::ChainDesc)));
    if (!chains.empty()) RSP_HIP(hipMemcpy(D.chains.get(), chains.data(), chains.size() * sizeof(dev::ChainDesc), hipMemcpyHostToDevice));
    D.pool.resize((size_t)std::max<i64>(R.pool_size, 1));
    D.mpool.resize((size_t)std::max<i64>(R.pool_size, 1));
    D.tpool.resize((size_t)std::max<i64>(R.pool_size, 1));
    {
        // the blocked inverse's tiles: blocks of 64 paired level by level (refill_kernels.hip: chain_gemm)
        std::vector<dev::ChainTile> t1, t2;
        D.tile_levels.clear();
        i64 max_blocks = 0;
        for (const dev::ChainDesc &c : chains) max_blocks = std::max<i64>(max_blocks, (c.w + 63) / 64);
        for (i64 half = 1; half < max_blocks; half *= 2) {
            const i64 first = (i64)t1.size();
            for (const dev::ChainDesc &c : chains) {
                const i64 nb = (c.w + 63) / 64;
                for (i64 g0 = 0; g0 < nb; g0 += 2 * half) {
                    const i64 a0 = g0, a1 = std::min(g0 + half, nb), b0 = a1, b1 = std::min(g0 + 2 * half, nb);
                    for (i64 i = b0; i < b1; ++i)
                        for (i64 j = a0; j < a1; ++j) {
                            // T[i][j] = sum_k C[i][k] Z11[k][j], k from column block j on (Z11 is lower triangular)
                            t1.push_back(dev::ChainTile{c.pool_off, c.w, (int32_t)(64 * i), (int32_t)(64 * j), (int32_t)(64 * j),
                                                        (int32_t)std::min<i64>(64 * a1, c.w), 0});
                            // Z21[i][j] = -sum_k Z22[i][k] T[k][j], k up to row block i (Z22 is lower triangular)
                            t2.push_back(dev::ChainTile{c.pool_off, c.w, (int32_t)(64 * i), (int32_t)(64 * j), (int32_t)(64 * b0),
                                                        (int32_t)std::min<i64>(64 * (i + 1), c.w), 0});
                        }
                }
            }
            D.tile_levels.emplace_back(first, (i64)t1.size() - first);
        }
        D.tiles1.resize(std::max<size_t>(1, t1.size() * sizeof(dev::ChainTile)));
        D.tiles2.resize(std::max<size_t>(1, t2.size() * sizeof(dev::ChainTile)));
        if (!t1.empty()) {
            RSP_HIP(hipMemcpy(D.tiles1.get(), t1.data(), t1.size() * sizeof(dev::ChainTile), hipMemcpyHostToDevice));
            RSP_HIP(hipMemcpy(D.tiles2.get(), t2.data(), t2.size() * sizeof(dev::ChainTile), hipMemcpyHostToDevice));
        }
    }
    D.ncoef = (i64)R.coef_src.size();
    D.ngroups = ngroups;
    D.ntri = (i64)R.tri16.size();
    D.nchain_entries = (i64)pos.size();
    D.pool_size = R.pool_size;
    D.nchains = (int)chains.size();
    D.ntiles = tiles;
    D.split = (int32_t)(nnz + n);
    D.nnz = (int32_t)nnz;
    D.usable = true;
}

bool TriSolver::refresh_device(const CscView &T, const double *dTx, hipStream_t stream)
{
    if (!have_blocked_ || !kept_ || G_.src.empty() || T.n != n_ || T.nnz() != (i64)G_.src.size() + n_) return false;
    if (!recipe_.built) build_device_recipe(T);
    DeviceRecipe &D = recipe_;
    if (!D.usable || D.ncoef != (i64)blocked_.coef.size() || D.ngroups > (i64)blocked_.groups.size()) return false;
    const dev::RefillSource S{D.map.get(), dTx, D.pool.get(), D.split, D.nnz};
    auto blocks = [](i64 count) { return dim3((unsigned)std::min<i64>((count + 255) / 256, 1 << 20)); };
    if (D.pool_size) {
        RSP_HIP(hipMemsetAsync(D.mpool.get(), 0, (size_t)D.pool_size * sizeof(double), stream));
        RSP_HIP(hipMemsetAsync(D.pool.get(), 0, (size_t)D.pool_size * sizeof(double), stream));
        hipLaunchKernelGGL(dev::chain_fill, blocks(D.nchain_entries), dim3(256), 0, stream, D.mpool.get(), D.chain_pos.get(), D.chain_code.get(),
                           D.nchain_entries, S);
        // the inverses: 64 x 64 diagonal blocks by substitution, the rest by matrix-core products level by level
        // (RSP_CHAIN_INVERT_BLOCKED=0: one wavefront per 64 columns walks all row blocks, the round-2 kernel)
        const bool blocked_inverse = !(getenv("RSP_CHAIN_INVERT_BLOCKED") && getenv("RSP_CHAIN_INVERT_BLOCKED")[0] == '0');
        hipLaunchKernelGGL(dev::chain_invert, dim3((unsigned)D.ntiles), dim3(64), 0, stream, D.mpool.get(), D.pool.get(),
                           reinterpret_cast<const dev::ChainDesc *>(D.chains.get()), D.nchains, blocked_inverse ? 1 : 0);
        if (blocked_inverse)
            for (const auto &lv : D.tile_levels) {
                if (!lv.second) continue;
                const dev::ChainTile *t1 = reinterpret_cast<const dev::ChainTile *>(D.tiles1.get()) + lv.first;
                const dev::ChainTile *t2 = reinterpret_cast<const dev::ChainTile *>(D.tiles2.get()) + lv.first;
                hipLaunchKernelGGL(dev::chain_gemm, dim3((unsigned)lv.second), dim3(64), 0, stream, D.mpool.get(), D.pool.get(), D.tpool.get(), t1, 0);
                hipLaunchKernelGGL(dev::chain_gemm, dim3((unsigned)lv.second), dim3(64), 0, stream, D.pool.get(), D.tpool.get(), D.pool.get(), t2, 1);
            }
    }
    if (D.ntri)
        hipLaunchKernelGGL(dev::refill_tri16, dim3((unsigned)((D.ntri + 15) / 16)), dim3(256), 0, stream, blocked_.coef.get(), D.tri16.get(), D.ntri, S);
    if (D.ncoef) hipLaunchKernelGGL(dev::refill_coef, blocks(D.ncoef), dim3(256), 0, stream, blocked_.coef.get(), D.coef_src.get(), D.ncoef, S);
    if (D.ngroups)
        hipLaunchKernelGGL(dev::refill_groups, dim3((unsigned)((D.ngroups + 255) / 256)), dim3(256), 0, stream, blocked_.groups.get(), D.grp_first.get(),
                           D.grp_row.get(), D.ngroups, S);
    RSP_HIP(hipGetLastError());
    host_values_stale_ = true;
    have_single_ = have_multi_ = false;
    staged_single_.reset();
    return true;
}

void TriSolver::refresh_upload()
{
    blocked_.coef.upload(kept_->coef);
    blocked_.groups.upload(kept_->groups);
    // the plain level programmes carry values of their own: rebuilt from the gather form at their next use
    have_single_ = have_multi_ = false;
    staged_single_.reset();
}

void TriSolver::run_blocked(double *X, i64 ld, i64 nrhs, hipStream_t stream)
{
    run_window(X, ld, nrhs, 0, nrhs, stream, nullptr);
}

static double *noalloc_scratch_for_current_device();

template <class T>
static void clone_buffer(DeviceBuffer<T> &dst, const DeviceBuffer<T> &src, int src_device, int dst_device)
{
    dst.resize(src.size());
    if (src.size()) RSP_HIP(hipMemcpyPeer(dst.get(), dst_device, src.get(), src_device, src.bytes()));
}

bool TriSolver::clone_resident(TriSolver &dst, int src_device, int dst_device) const
{
    if (!have_blocked_) return false;
    dst.n_ = n_;
    dst.nnz_ = nnz_;
    dst.kind_ = kind_;
    dst.num_row_levels_ = num_row_levels_;
    const BlockedProgram &S = blocked_;
    BlockedProgram &D = dst.blocked_;
    D.ntasks = S.ntasks; D.nlevels = S.nlevels; D.nblocks = S.nblocks; D.scratch_rows = S.scratch_rows; D.big_chains = S.big_chains;
    D.inverted_rows = S.inverted_rows; D.padded_entries = S.padded_entries; D.nclusters = S.nclusters; D.cluster_rows = S.cluster_rows;
    D.cluster_entries = S.cluster_entries; D.launches = S.launches; D.cluster_slots = S.cluster_slots; D.nitems = S.nitems;
    D.item_entries = S.item_entries; D.single_block_tasks = S.single_block_tasks;
    D.levels = S.levels;
    clone_buffer(D.task_ptr, S.task_ptr, src_device, dst_device);
    clone_buffer(D.hdr, S.hdr, src_device, dst_device);
    clone_buffer(D.src_idx, S.src_idx, src_device, dst_device);
    clone_buffer(D.coef, S.coef, src_device, dst_device);
    clone_buffer(D.prologue, S.prologue, src_device, dst_device);
    clone_buffer(D.items, S.items, src_device, dst_device);
    clone_buffer(D.item_masks, S.item_masks, src_device, dst_device);
    clone_buffer(D.clusters, S.clusters, src_device, dst_device);
    clone_buffer(D.cluster_ids, S.cluster_ids, src_device, dst_device);
    clone_buffer(D.groups, S.groups, src_device, dst_device);
    RSP_HIP(hipFuncSetAttribute((const void *)dev::solve_clusters, hipFuncAttributeMaxDynamicSharedMemorySize,
                                4 * (std::min(D.cluster_slots, 64) * 512 + 1024)));  // (per device)
    dst.have_blocked_ = true;
    return true;
}

void TriSolver::prepare_blocked(i64 total_cols)
{
    // (the allocation-free entry point only READS the address of its scratch array: looked up here, where allocating is allowed --
    // the first lookup on a device may materialise the code object's globals)
    (void)noalloc_scratch_for_current_device();
    if (!have_blocked_) {
        build_blocked();
        have_blocked_ = true;
    }
    const i64 rows = blocked_.scratch_rows;
    if (rows) {
        const size_t want = (size_t)rows * (size_t)total_cols;
        scratch_.resize(want);
        scratch_cols_ = total_cols;
    }
}

// Columns [col_off, col_off + nrhs) of a row-major block of `total_cols` columns whose first column is at X - col_off:
// the scratch rows use the same column window, so two windows of one block can be in flight on two streams.
// `after_heavy`, if given, is recorded on `stream` once 90 % of the sweep's workgroups have been launched.
void TriSolver::run_window(double *X, i64 ld, i64 nrhs, i64 col_off, i64 total_cols, hipStream_t stream, hipEvent_t after_heavy,
                           double *scratch_override)
{
    if (!scratch_override) prepare_blocked(total_cols);
    const BlockedProgram &P = blocked_;
    double *const Sbase = scratch_override ? scratch_override : scratch_.get();
    double *const Sw = Sbase ? Sbase + col_off : nullptr;
    const i64 lds = total_cols;
    const dev::BlockArrays A{P.task_ptr.get(), P.hdr.get(), P.src_idx.get(), P.coef.get(), P.prologue.get()};
    const dev::ClusterArrays C{P.clusters.get(), P.cluster_ids.get(), P.groups.get()};
    // 64 right-hand sides per wavefront: 16-unknown blocks run as 4 MFMA tiles of 16 columns, small blocks one column per lane
    const int nchunks = (int)((nrhs + 63) / 64);
    // 16-byte row pieces need aligned rows and whole groups of 4 columns; anything else takes the element-wise path
    const bool fast = nrhs % 4 == 0 && ld % 2 == 0 && lds % 2 == 0 && ((uintptr_t)X % 16) == 0 && ((uintptr_t)Sw % 16) == 0;
    const i64 cluster_cpw = env_or("RSP_CLUSTER_CPW", 2);
    const i64 lean_wgs = env_or("RSP_LEAN_WGS", kLeanLaunchWorkgroups);
    const i64 lean_wgs_single = env_or("RSP_LEAN_WGS_SINGLE", kLeanLaunchWorkgroupsSingle);  // one right-hand side
    const i64 small_waves = env_or("RSP_SMALL_WAVES", kSmallLaunchWaves);
    const i64 item_waves = env_or("RSP_ITEM_WAVES", kItemLaunchWaves);
    const i64 item_ks = env_or("RSP_ITEM_KS", 0);  // 1, 2, 4: forced K-split of the item kernel (0: by the level's size)
    const dev::ItemArrays IA{P.items.get(), P.src_idx.get(), P.coef.get(), P.item_masks.get()};
    // A level's wide items run as items (64 x 64 register tiles) when the block is made of whole, aligned 64-column chunks and
    // the level has enough of them to fill the chip; otherwise as their twin 16-row blocks
    auto items_mode = [&](const BlockedProgram::Level &lv) {
        return lv.nitems > 0 && fast && nrhs % 64 == 0 && item_waves > 0 && (i64)lv.nitems * nchunks >= item_waves;
    };
    // workgroups of the whole sweep (a proxy for its work), to find the point where the heavy part is over
    i64 total = 0, done = 0;
    auto level_work = [&](const BlockedProgram::Level &lv) {
        const bool it = items_mode(lv);
        return (i64)lv.nclusters * nchunks + ((i64)(lv.ntasks[0] + lv.ntasks[1] + (it ? 0 : lv.ntasks[2])) * nchunks + 7) / 8 +
               ((i64)(lv.ntasks[3] + (it ? 0 : lv.ntasks[4])) * nchunks + 1) / 2 + (i64)(lv.ntasks[5] + (it ? 0 : lv.ntasks[6])) * nchunks +
               (it ? (i64)lv.nitems * nchunks : 0);
    };
    if (after_heavy)
        for (const auto &lv : P.levels) total += level_work(lv);
    // what the block kernel of one level runs, and with how many workgroups, for `chunks` column chunks per block.  twins:
    // the twin blocks of the level's wide items are included (classes 1+2, 3+4, 5+6 are contiguous in task and block order)
    auto small_alone_of = [&](const BlockedProgram::Level &lv) { return lv.ntasks[0] > 0 && (i64)lv.ntasks[0] * nchunks >= small_waves; };
    auto describe = [&](const BlockedProgram::Level &lv, bool small_alone, bool twins, int chunks, dev::LevelDesc &L) {
        const int n01 = lv.ntasks[1] + (twins ? lv.ntasks[2] : 0) + (small_alone ? 0 : lv.ntasks[0]);  // classes 0, 1 (and 2) are consecutive tasks
        L.task_begin = small_alone ? lv.task_begin[1] : lv.task_begin[0];
        L.block_begin[0] = lv.block_begin[3];
        L.block_begin[1] = lv.block_begin[5];
        L.blk0 = small_alone ? lv.block_begin[1] : lv.block_begin[0];
        L.rec0 = small_alone ? lv.rec_begin[1] : lv.rec_begin[0];
        L.rec_begin[0] = lv.rec_begin[3];
        L.rec_begin[1] = lv.rec_begin[5];
        L.single = P.single_block_tasks ? 1 : 0;
        L.ntasks = n01;
        L.nblk[0] = lv.ntasks[3] + (twins ? lv.ntasks[4] : 0);
        L.nblk[1] = lv.ntasks[5] + (twins ? lv.ntasks[6] : 0);
        L.wg0 = (int)(((i64)n01 * chunks + 7) / 8);
        L.wg1 = (int)(((i64)L.nblk[0] * chunks + 1) / 2);
        L.wg2 = (int)((i64)L.nblk[1] * chunks);
        return (unsigned)(((i64)L.wg0 + L.wg1 + L.wg2 + 7) / 8 * 8);  // whole rounds over the 8 XCDs
    };
    const bool narrow_block = nrhs <= 16;
    const i64 lean_threshold = nrhs == 1 ? lean_wgs_single : lean_wgs;

    // one right-hand side, RSP_FLAGGED_LEVELS=1: runs of latency-bound levels as one launch each (build_flag_runs)
    const bool flagged = nrhs == 1 && !scratch_override && !P.flag_runs.empty();
    size_t next_run = 0;
    if (flagged) {
        ++blocked_.fr_epoch;
        blocked_.fr_ran = true;
        RSP_HIP(hipMemsetAsync(P.fr_counters.get(), 0, P.flag_runs.size() * sizeof(int), stream));
    }
    bool recorded = false;
    for (size_t li = 0; li < P.levels.size(); ++li) {
        if (flagged && next_run < P.flag_runs.size() && li == P.flag_runs[next_run].level_begin) {
            const auto &run = P.flag_runs[next_run];
            dev::FlagRun F;
            F.levels = reinterpret_cast<const dev::LevelDesc *>(P.fr_levels.get()) + run.level_off;
            F.tickets = reinterpret_cast<const int2 *>(P.fr_tickets.get()) + run.ticket_off;
            F.wait_ptr = P.fr_wait_ptr.get() + run.ticket_off;
            F.wait_list = P.fr_wait_list.get();
            F.flags = P.fr_flags.get();
            F.counter = P.fr_counters.get() + next_run;
            F.failed = P.fr_failed.get();
            F.flag_base = (int)run.ticket_off;
            F.ntickets = run.ntickets;
            F.epoch = P.fr_epoch;
            F.spin_limit = (unsigned)env_or("RSP_FLAGGED_SPIN_LIMIT", 1 << 22);
            if (fast) hipLaunchKernelGGL(dev::solve_levels_flagged<true>, dim3((unsigned)run.ntickets), dim3(512), 7 * 2048, stream, A, F, X, ld, Sw, lds, (int)nrhs);
            else hipLaunchKernelGGL(dev::solve_levels_flagged<false>, dim3((unsigned)run.ntickets), dim3(512), 7 * 2048, stream, A, F, X, ld, Sw, lds, (int)nrhs);
            li = run.level_end - 1;
            ++next_run;
            continue;
        }
        const auto &lv = P.levels[li];
        if (after_heavy && !recorded && done * 10 >= total * 9) {
            RSP_HIP(hipEventRecord(after_heavy, stream));
            recorded = true;
        }
        done += level_work(lv);
        if (lv.nclusters) {  // clusters and block tasks of one level are independent of each other
            const int cpw = (int)std::min<i64>(std::min<i64>(nchunks, 4), cluster_cpw);
            const size_t per_wave = (size_t)P.cluster_slots * 512 + 1024;
            const i64 wgs = ((i64)lv.nclusters * ((nchunks + cpw - 1) / cpw) + 7) / 8 * 8;  // whole rounds over the 8 XCDs
            hipLaunchKernelGGL(dev::solve_clusters, dim3((unsigned)wgs), dim3(64 * cpw), per_wave * cpw, stream, C, lv.cluster_begin,
                               lv.nclusters, nchunks, X, ld, (int)nrhs, P.cluster_slots, cpw);
        }
        // many small-block tasks: their own lean, high-occupancy launch; a few ride along with class 1
        const bool small_alone = small_alone_of(lv);
        if (small_alone)
            hipLaunchKernelGGL(dev::solve_level_small, dim3((unsigned)((((i64)lv.ntasks[0] * nchunks + 3) / 4 + 7) / 8 * 8)), dim3(256), 0,
                               stream, A, lv.task_begin[0], lv.ntasks[0], X, ld, Sw, lds, (int)nrhs, nchunks, lv.block_begin[0], lv.rec_begin[0],
                               P.single_block_tasks ? 1 : 0);
        // the level's wide items, as items: whole K-steps per wavefront when there are enough (item, chunk) pairs to fill the
        // chip, else shared by two or four wavefronts (levels of few, long items)
        const bool as_items = items_mode(lv);
        if (as_items) {
            const i64 pairs = (i64)lv.nitems * nchunks;
            int ksplit = pairs >= 2048 || lv.item_ksteps_max < 64 ? 1 : pairs >= 1024 || lv.item_ksteps_max < 128 ? 2 : 4;
            if (item_ks == 1 || item_ks == 2 || item_ks == 4) ksplit = (int)item_ks;
            dev::launch_items(ksplit, IA, lv.item_begin, lv.nitems, nchunks, X, ld, Sw, lds, stream);
        }
        dev::LevelDesc L;
        unsigned wgs = describe(lv, small_alone, !as_items, nchunks, L);
        if (wgs) {
            // Few workgroups: a workgroup's own chain (round trips, then its matrix-core time) is the launch's duration --
            // the deep-ring variant, and 16 columns per wavefront instead of 64 so that four times as many compute units
            // share the work.  Many: the lean variant at twice the occupancy.
            const bool pipe = wgs < (unsigned)lean_threshold;
            const i64 narrow = env_or("RSP_LAT_NARROW", 1);
            int level_chunks = nchunks;
            const bool nt1 = pipe && narrow && nrhs > 16;  // (blocks of at most 16 columns: see narrow_block)
            if (nt1) {
                level_chunks = (int)((nrhs + 15) / 16);
                wgs = describe(lv, small_alone, !as_items, level_chunks, L);
            }
            // Blocks of at most 16 columns (one right-hand side among them) always run 16-column wavefronts (one column
            // tile: a quarter of the registers), lean on heavy levels.  Wider UNALIGNED blocks (element-wise path) have no
            // lean variant: their selects need the registers of the deep-ring one.
            auto kernel = narrow_block ? (fast ? (pipe ? dev::solve_level<true, true, 1> : dev::solve_level<true, false, 1>)
                                               : (pipe ? dev::solve_level<false, true, 1> : dev::solve_level<false, false, 1>))
                        : fast ? (nt1 ? dev::solve_level<true, true, 1> : pipe ? dev::solve_level<true, true, 4> : dev::solve_level<true, false, 4>)
                                       : (nt1 ? dev::solve_level<false, true, 1> : dev::solve_level<false, true, 4>);
            if (narrow_block && !nt1) {
                level_chunks = 1;
                wgs = describe(lv, small_alone, !as_items, level_chunks, L);
            }
            hipLaunchKernelGGL(kernel, dim3(wgs), dim3(512), 7 * 8192, stream, A, L, X, ld, Sw, lds, (int)nrhs, level_chunks);
#ifdef RSP_TRI_CLOCKS
            if (getenv("RSP_TRI_CLOCKS") && L.wg2 > 0) {
                RSP_HIP(hipStreamSynchronize(stream));
                unsigned long long clk[16];
                RSP_HIP(hipMemcpyFromSymbol(clk, HIP_SYMBOL(dev::g_tri_clk), sizeof(clk)));
                fprintf(stderr, "level wgs %u (class-3 %d) nt1 %d pipe %d: record %lld gather %lld barrier %lld finish %lld  stages %lld\n", wgs, L.wg2, (int)nt1,
                        (int)pipe, (long long)(clk[1] - clk[0]), (long long)(clk[2] - clk[0]), (long long)(clk[3] - clk[0]), (long long)(clk[4] - clk[0]),
                        (long long)clk[8]);
            }
#endif
        }
    }
    if (after_heavy && !recorded) RSP_HIP(hipEventRecord(after_heavy, stream));
    RSP_HIP(hipGetLastError());
}

// first.solve then second.solve on the same row-major block (lsolve + ltsolve of cholsol, lsolve + usolve of lusol).
// Wide blocks are cut into column parts (kPairParts) that travel through both sweeps on streams of their own, each part
// starting when the heavy levels of the previous part's first sweep are over: the long chains of nearly empty levels
// at the top of the tree (latency-bound, the chip idles) then run beside another part's heavy levels.
void solve_pair_device(TriSolver &first, TriSolver &second, double *X, i64 ld, i64 nrhs, hipStream_t stream)
{
    if (nrhs <= 0 || first.n() == 0) return;
    const bool overlap = !(getenv("RSP_PAIR_OVERLAP") && getenv("RSP_PAIR_OVERLAP")[0] == '0');
    const i64 nchunks = (nrhs + 63) / 64;
    if (!overlap || nchunks < 2) {
        first.solve_device(X, 1, ld, nrhs, stream);
        second.solve_device(X, 1, ld, nrhs, stream);
        return;
    }
    // auxiliary streams and events per (thread, device), created on first use and kept
    constexpr int kMaxParts = 4;
    struct Aux {
        hipStream_t stream[kMaxParts] = {};
        hipEvent_t start = nullptr, heavy[kMaxParts] = {}, done[kMaxParts] = {};
    };
    static thread_local std::vector<Aux> per_device;
    const i64 want_parts = std::max<i64>(2, std::min<i64>(kMaxParts, env_or("RSP_PAIR_PARTS", kPairParts)));
    int device = 0;
    RSP_HIP(hipGetDevice(&device));
    if ((size_t)device >= per_device.size()) per_device.resize(device + 1);
    Aux &ax = per_device[device];
    if (!ax.start) {
        RSP_HIP(hipEventCreateWithFlags(&ax.start, hipEventDisableTiming));
        for (int k = 0; k < kMaxParts; ++k) {
            if (k) RSP_HIP(hipStreamCreateWithFlags(&ax.stream[k], hipStreamNonBlocking));
            RSP_HIP(hipEventCreateWithFlags(&ax.heavy[k], hipEventDisableTiming));
            RSP_HIP(hipEventCreateWithFlags(&ax.done[k], hipEventDisableTiming));
        }
    }
    // parts of whole 64-column chunks, the earlier ones one chunk larger when they do not divide evenly
    const int parts = (int)std::min<i64>(want_parts, nchunks);
    i64 col[kMaxParts + 1] = {0};
    for (int k = 0; k < parts; ++k) col[k + 1] = std::min<i64>(nrhs, col[k] + (nchunks / parts + (k < nchunks % parts ? 1 : 0)) * 64);
    col[parts] = nrhs;
    first.prepare_blocked(nrhs);   // allocations happen before anything is in flight
    second.prepare_blocked(nrhs);
    RSP_HIP(hipEventRecord(ax.start, stream));
    for (int k = 0; k < parts; ++k) {
        const hipStream_t sk = k ? ax.stream[k] : stream;
        if (k) {
            RSP_HIP(hipStreamWaitEvent(sk, ax.start, 0));
            RSP_HIP(hipStreamWaitEvent(sk, ax.heavy[k - 1], 0));  // the heavy levels of the previous part are under way
        }
        first.run_window(X + col[k], ld, col[k + 1] - col[k], col[k], nrhs, sk, k + 1 < parts ? ax.heavy[k] : nullptr);
        second.run_window(X + col[k], ld, col[k + 1] - col[k], col[k], nrhs, sk, nullptr);
        if (k) RSP_HIP(hipEventRecord(ax.done[k], sk));
    }
    for (int k = 1; k < parts; ++k) RSP_HIP(hipStreamWaitEvent(stream, ax.done[k], 0));
}

// ---- the entry point that allocates nothing (rsp_tri_solve_dev_noalloc_f64) ------------------------------------------
// Scratch rows of the blocked programme in a file-scope device array instead of a hipMalloc'ed buffer.  In host code the
// array's name is only a stand-in: its device address is looked up once per device with hipGetSymbolAddress and THAT
// pointer is what the kernels receive.  Every call on a GPU shares that GPU's array: calls that overlap on one GPU race.
// Nothing in it has to start at zero except scratch row 0 of the supernodal programme (a row of zeros, cleared by every
// call); every other scratch row is written by the programme before it is read.
static __device__ __align__(256) double g_noalloc_scratch[kNoAllocScratchDoubles];

static double *noalloc_scratch_for_current_device()
{
    constexpr int kMaxDevices = 64;
    static std::mutex guard;
    static double *ptr[kMaxDevices] = {};
    int device = 0;
    RSP_HIP(hipGetDevice(&device));
    if (device < 0 || device >= kMaxDevices) throw DeviceError("device index out of range");
    std::lock_guard<std::mutex> lock(guard);
    if (!ptr[device]) {
        double *p = nullptr;
        RSP_HIP(hipGetSymbolAddress((void **)&p, HIP_SYMBOL(g_noalloc_scratch)));  // this device's copy of the array
        ptr[device] = p;
    }
    return ptr[device];
}

int TriSolver::solve_device_noalloc(double *X, i64 ld, i64 nrhs, hipStream_t stream)
{
    if (nrhs <= 0 || n_ == 0) return kOk;
    if (!have_blocked_) return kBadInput;  // building and uploading the programme allocates: an earlier solve must have done it
    const i64 rows = blocked_.scratch_rows;
    if (rows > 0 && rows > kNoAllocScratchDoubles / nrhs) return kBadInput;
    double *S = noalloc_scratch_for_current_device();
    run_window(X, ld, nrhs, 0, nrhs, stream, nullptr, S);
    return kOk;
}

void TriSolver::run_single(double *x, hipStream_t stream)
{
    if (!have_single_) {
        const double t0 = now_s();
        if (staged_single_) {
            upload_program(*staged_single_, single_);
            staged_single_.reset();
        } else {
            build_program(kTaskEntriesSingle, single_);
        }
        have_single_ = true;
        if (getenv("RSP_TIMING"))
            fprintf(stderr, "tri level programme: %.2fs, %lld levels, %lld tasks\n", now_s() - t0, (long long)single_.nlevels, (long long)single_.ntasks);
    }
    const dev::TriArrays A = arrays_of(single_);
    const int wpb = 4;
    for (const auto &lv : single_.levels) {
        const dim3 grid((unsigned)((lv.second + wpb - 1) / wpb)), block(64 * wpb);
        hipLaunchKernelGGL(dev::solve_level_single, grid, block, 0, stream, A, lv.first, lv.second, x);
    }
    RSP_HIP(hipGetLastError());
}

// rows by dependency level, index order inside a level (a counting sort of level_of_row_); grabs: runs of one level
void TriSolver::upload_rows_schedule()
{
    std::vector<i64> at((size_t)num_row_levels_ + 1, 0);
    for (i64 r = 0; r < n_; ++r) at[(size_t)level_of_row_[(size_t)r] + 1]++;
    for (i64 l = 0; l < num_row_levels_; ++l) at[(size_t)l + 1] += at[(size_t)l];
    std::vector<int2> grabs;
    for (i64 l = 0; l < num_row_levels_; ++l)
        for (i64 t = at[(size_t)l]; t < at[(size_t)l + 1]; t += dev::kRowsPerGrab)
            grabs.push_back(int2{(int)t, (int)std::min<i64>(at[(size_t)l + 1], t + dev::kRowsPerGrab)});
    std::vector<int32_t> order((size_t)n_);
    {
        std::vector<i64> fill(at.begin(), at.end() - 1);
        for (i64 r = 0; r < n_; ++r) order[(size_t)fill[(size_t)level_of_row_[(size_t)r]]++] = (int32_t)r;
    }
    rows_.order.upload(order);
    rows_.ngrabs = (i64)grabs.size();
    rows_.grabs.resize(std::max<size_t>(1, grabs.size() * sizeof(int2)));
    if (!grabs.empty()) RSP_HIP(hipMemcpy(rows_.grabs.get(), grabs.data(), grabs.size() * sizeof(int2), hipMemcpyHostToDevice));
    rows_.done.resize((size_t)n_);
    rows_.ctl.resize(sizeof(dev::RowsControl));
}

bool TriSolver::rows_wanted()
{
    const char *mode = getenv("RSP_SINGLE_USE");
    return !(mode && (mode[0] == '0' || mode[0] == '1'));
}

void TriSolver::prepare_rows_pattern()
{
    if (!G_.placeholder || G_.src.size() != G_.idx.size() || n_ >= INT32_MAX || rows_.uploaded) return;
    rows_.ptr.upload(G_.ptr);
    rows_.idx.upload(G_.idx);
    rows_.src.upload(G_.src);
    rows_.val.resize(std::max<size_t>(1, G_.idx.size()));
    rows_.diag.resize((size_t)n_);
    upload_rows_schedule();
    rows_.pattern_only = true;
}

bool TriSolver::rows_values_from_device(const int64_t *dTp, const double *dTx, hipStream_t stream)
{
    if (!rows_.pattern_only) return false;
    const i64 count = std::max<i64>((i64)G_.idx.size(), n_);
    hipLaunchKernelGGL(dev::rows_gather_values, dim3((unsigned)std::min<i64>((count + 255) / 256, 1 << 20)), dim3(256), 0, stream, rows_.src.get(),
                       dTx, rows_.val.get(), (int64_t)G_.idx.size(), dTp, rows_.diag.get(), (int64_t)n_, (kind_ == 0 || kind_ == 1) ? 1 : 0);
    RSP_HIP(hipGetLastError());
    rows_.uploaded = true;
    single_use_rows_ = true;
    single_use_levels_ = false;
    host_values_stale_ = true;  // the host's gather form holds no values: nothing else can be built from it
    return true;
}

void TriSolver::run_rows(double *x, hipStream_t stream)
{
    if (!rows_.uploaded) {
        if (host_values_stale_ || G_.placeholder) throw DeviceError("the row kernel needs the factor's values on the host: refreshed on the device only");
        rows_.ptr.upload(G_.ptr);
        rows_.idx.upload(G_.idx);
        rows_.val.upload(G_.val);
        rows_.diag.upload(G_.diag);
        upload_rows_schedule();
        rows_.uploaded = true;
    }
    RSP_HIP(hipMemsetAsync(rows_.done.get(), 0, (size_t)n_ * sizeof(int), stream));
    RSP_HIP(hipMemsetAsync(rows_.ctl.get(), 0, sizeof(dev::RowsControl), stream));
    hipDeviceProp_t prop;
    int device = 0;
    RSP_HIP(hipGetDevice(&device));
    RSP_HIP(hipGetDeviceProperties(&prop, device));
    const i64 waves = std::max<i64>(1, std::min<i64>(rows_.ngrabs, (i64)prop.multiProcessorCount * 32));
    const unsigned spin_limit = (unsigned)env_or("RSP_ROWS_SPIN_LIMIT", 1 << 24);
    hipLaunchKernelGGL(dev::solve_rows_flagged, dim3((unsigned)waves), dim3(64), 0, stream, rows_.ptr.get(), rows_.idx.get(), rows_.val.get(),
                       rows_.diag.get(), x, rows_.done.get(), reinterpret_cast<dev::RowsControl *>(rows_.ctl.get()), rows_.order.get(),
                       reinterpret_cast<const int2 *>(rows_.grabs.get()), (int)rows_.ngrabs, spin_limit, G_.descending ? 1 : 0);
    RSP_HIP(hipGetLastError());
    rows_ran_ = true;
}

void TriSolver::check_rows_run()
{
    if (blocked_.fr_ran) {  // RSP_FLAGGED_LEVELS=1: a wait inside a flagged launch gave up (a defect)
        blocked_.fr_ran = false;
        int failed = 0;
        RSP_HIP(hipMemcpy(&failed, blocked_.fr_failed.get(), sizeof failed, hipMemcpyDeviceToHost));
        if (failed) {
            RSP_HIP(hipMemset(blocked_.fr_failed.get(), 0, sizeof(int)));
            throw DeviceError("solve_levels_flagged: a workgroup gave up waiting for its producers");
        }
    }
    if (!rows_ran_) return;
    rows_ran_ = false;
    dev::RowsControl c{};
    RSP_HIP(hipMemcpy(&c, rows_.ctl.get(), sizeof c, hipMemcpyDeviceToHost));
    if (c.failed)
        throw DeviceError("solve_rows_flagged: row " + std::to_string(c.stuck_row) + " gave up waiting for row " + std::to_string(c.stuck_source));
}

void TriSolver::solve_device(double *dX, int layout, i64 ld, i64 nrhs, hipStream_t stream)
{
    if (nrhs <= 0 || n_ == 0) return;
    if (nrhs == 1 && (layout == 0 || ld == 1) && single_use_rows_) {
        run_rows(dX, stream);
        return;
    }
    if (nrhs == 1 && (layout == 0 || ld == 1)) {
        // one right-hand side: a column vector is its own row-major n x 1 block.  The blocked programme needs ~170
        // launches per sweep where the plain level programme needs one per dependency level (thousands).
        const bool single_levels = getenv("RSP_TRI_SINGLE") && getenv("RSP_TRI_SINGLE")[0] == '1';
        if (single_levels || single_use_levels_) run_single(dX, stream);
        else run_blocked(dX, 1, 1, stream);
        return;
    }
    const bool blocked = !(getenv("RSP_TRI_BLOCKED") && getenv("RSP_TRI_BLOCKED")[0] == '0');
    auto run = [&](double *X, i64 ldx) { blocked ? run_blocked(X, ldx, nrhs, stream) : run_multi(X, ldx, nrhs, stream); };
    if (layout == 1) {
        run(dX, ld);
        return;
    }
    // column-major caller: stage through a row-major workspace (two extra passes over X)
    work_.resize((size_t)n_ * (size_t)nrhs);
    const dim3 grid((unsigned)((n_ + 31) / 32), (unsigned)((nrhs + 31) / 32)), block(256);
    hipLaunchKernelGGL(dev::colmajor_to_rowmajor, grid, block, 0, stream, dX, ld, work_.get(), nrhs, n_, (int)nrhs,
                       (const int64_t *)nullptr, 0);
    run(work_.get(), nrhs);
    hipLaunchKernelGGL(dev::rowmajor_to_colmajor, grid, block, 0, stream, work_.get(), nrhs, dX, ld, n_, (int)nrhs,
                       (const int64_t *)nullptr, 0);
    RSP_HIP(hipGetLastError());
}

void TriSolver::solve_host(double *X, i64 ldx, i64 nrhs)
{
    if (nrhs <= 0 || n_ == 0) return;
    DeviceBuffer<double> d((size_t)n_ * (size_t)nrhs);
    copy_block_to_device(d.get(), X, n_, ldx, nrhs);
    solve_device(d.get(), 0, n_, nrhs, nullptr);
    RSP_HIP(hipDeviceSynchronize());
    check_rows_run();
    copy_block_to_host(X, d.get(), n_, ldx, nrhs);
}

}  // namespace rsp
