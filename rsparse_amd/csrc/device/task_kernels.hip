// HIP kernel (gfx950) of the TASKS of the supernodal multi-right-hand-side programme (host/supernodal.cpp, SnProgram in
// host/sparse.hpp): the bottom of the dependency forest, where an unknown has a handful of entries.
//
// A task is a run of consecutive unknowns -- whole subtrees, one or two hundred unknowns -- whose rows of X (its own
// unknowns plus the external rows they read, or the partial rows they send upwards) fit in LDS.  One wavefront owns
// (task, 16 right-hand sides) and is a workgroup of its own: nothing is shared, there is no barrier after the load.
//   1. Rows -> LDS slots: header, then the row ids of every slot (kept in registers for the write-back), then every row
//      piece at once, straight into LDS (global_load_lds_dwordx4: wave-uniform LDS base + 16 bytes per lane is exactly
//      the slot layout; 8 slots of 128 bytes per instruction, no registers held while they travel).  Scratch slots
//      (id < 0: partial sums for the levels above) are cleared instead.
//   2. The arithmetic is a stream of 16-byte records, one per (step, lane group): the wavefront's four groups of 16 lanes
//      work on four DIFFERENT unknowns of one dependency level, lane j of a group on right-hand side j.  A record is
//      begin (acc = x[slot]), entry (acc -= coef * x[slot]) or end (x[slot] = acc / diag).  Records stream from global
//      memory through a ring of registers 24 steps ahead of their use; the LDS operands of a window of 4 steps are
//      read before any of them is used, which is safe because the host pads every level to whole windows (operands
//      of a level are final before its first window).  Nothing waits on global memory once the rows are in LDS.
//   3. Own slots -> X (or scratch rows).
// Algorithmic traffic only: every row of X is read once and written once per sweep; records are 16 bytes per entry.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../host/sparse.hpp"

namespace rsp {
namespace dev {

struct TaskArrays {
    const ClusterHeader *hdr;  // sorted by level; ngroups = steps, grp_off = first step
    const int32_t *ids;        // row behind every slot: >= 0 row of X, < 0 scratch row ~id
    const SnTaskRec *recs;     // 4 per step
};

typedef int tv4 __attribute__((ext_vector_type(4)));
typedef double td2 __attribute__((ext_vector_type(2)));

struct TaskWindow {
    tv4 rec[kTaskWindow];
};

__device__ __forceinline__ void task_load_window(TaskWindow &W, const tv4 *recs, int64_t s0, int step, int nsteps, int g)
{
#pragma unroll
    for (int u = 0; u < kTaskWindow; ++u) W.rec[u] = recs[(s0 + min(step + u, nsteps - 1)) * 4 + g];  // clamped: unconditional loads
}

// FAST: whole chunks of 16 columns, rows 16-byte aligned (ld even, X and S aligned).
template <bool FAST>
__global__ __launch_bounds__(64) void solve_sparse_tasks(TaskArrays D, int64_t task_begin, int ntasks, int nchunks, double *__restrict__ X,
                                                          int64_t ldx, double *__restrict__ S, int64_t lds, int nrhs)
{
    extern __shared__ double xs[];  // (slots + 1) rows of 16 doubles
    constexpr int RB = 16;          // doubles per LDS row
    constexpr int LPR = 8;          // lanes per row (16 bytes each)
    constexpr int RPI = 8;          // rows per wave-wide load
    constexpr int DEPTH = kTaskDepth;  // windows in flight; nsteps is a multiple of kTaskWindow * DEPTH
    const int lane = threadIdx.x;
    // XCD-aware placement: blockIdx % 8 share an XCD; every XCD walks a contiguous range of items, and the column chunks
    // of one task are consecutive items: they read the same records out of that XCD's L2 and touch the same rows of X at
    // about the same time (DRAM page locality)
    const int per_xcd = (int)(gridDim.x >> 3);
    const int item = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    const int c = item / nchunks;
    if (c >= ntasks) return;
    const int chunk = item - c * nchunks;
    const int c0 = chunk * RB;
    const ClusterHeader h = D.hdr[task_begin + c];
    const int32_t *ids = D.ids + h.ids_off;
    const int nslots = h.n_ext + h.n_own;
    const int nsteps = h.ngroups;
    const int g = lane >> 4, j = lane & 15;
    const tv4 *recs = reinterpret_cast<const tv4 *>(D.recs);

    TaskWindow R[DEPTH];

    // ---- rows -> LDS ---------------------------------------------------------------------------------------------
    const int sub = lane % LPR, rsel = lane / LPR;
    constexpr int MAXG = 40;  // groups of RPI slots: 320 slots, 40 KiB of LDS
    int32_t id[MAXG];
#pragma unroll
    for (int u = 0; u < MAXG; ++u) id[u] = ids[min(u * RPI + rsel, nslots - 1)];
    if (FAST) {
#pragma unroll
        for (int u = 0; u < MAXG; ++u) {
            if (u * RPI < nslots) {  // wave-uniform
                const int s = u * RPI + rsel;
                if (s < nslots && id[u] >= 0)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(X + (int64_t)id[u] * ldx + c0 + 2 * sub),
                                                     (__attribute__((address_space(3))) void *)(xs + (size_t)(u * RPI) * RB), 16, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < MAXG; ++u) {
            const int s = u * RPI + rsel;
            if (u * RPI < nslots && s < nslots && id[u] < 0) *reinterpret_cast<td2 *>(xs + (size_t)s * RB + 2 * sub) = td2{0.0, 0.0};
        }
    } else {
#pragma unroll
        for (int u = 0; u < MAXG; ++u) {
            const int s = u * RPI + rsel;
            if (u * RPI < nslots && s < nslots) {
                td2 v = td2{0.0, 0.0};
                if (id[u] >= 0) {
                    const double *p = X + (int64_t)id[u] * ldx + c0 + 2 * sub;
                    if (c0 + 2 * sub < nrhs) v.x = p[0];
                    if (c0 + 2 * sub + 1 < nrhs) v.y = p[1];
                }
                *reinterpret_cast<td2 *>(xs + (size_t)s * RB + 2 * sub) = v;
            }
        }
    }
    if (lane < LPR) *reinterpret_cast<td2 *>(xs + (size_t)nslots * RB + 2 * lane) = td2{0.0, 0.0};  // the row of zeros
    __syncthreads();  // one wavefront; also waits for the rows in flight

    // ---- the record stream ---------------------------------------------------------------------------------------
    // The ring is filled HERE, after the wait above, although the records do not depend on the rows: with loads already
    // outstanding at loop entry in exactly the order the loop leaves them at its back edge, the compiler's vmcnt
    // bookkeeping stays exact (vmcnt(20) per window); filled before the row loads it merged the two states into a
    // vmcnt(0) in every pass of the loop (measured: 2.6 ms instead of 0.x ms per launch).
    if (nsteps > 0) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) task_load_window(R[d], recs, h.grp_off, kTaskWindow * d, nsteps, g);
    }
    double acc = 0.0;
    double *xj = xs + j;
    for (int base = 0; base < nsteps; base += kTaskWindow * DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            {
                double t[kTaskWindow];
#pragma unroll
                for (int u = 0; u < kTaskWindow; ++u) t[u] = xj[(size_t)(uint32_t)R[d].rec[u].z * RB];
#pragma unroll
                for (int u = 0; u < kTaskWindow; ++u) {
                    const tv4 r = R[d].rec[u];
                    const double coef = __hiloint2double(r.y, r.x);
                    const uint32_t op = (uint32_t)r.w;
                    const double sub_acc = fma(-coef, t[u], acc);
                    acc = op == SnTaskRec::kEntry ? sub_acc : acc;
                    acc = op == SnTaskRec::kBegin ? t[u] : acc;
                    if (op == SnTaskRec::kEnd) {
                        acc *= coef;
                        xj[(size_t)(uint32_t)r.z * RB] = acc;
                    }
                }
            }
            task_load_window(R[d], recs, h.grp_off, base + kTaskWindow * (DEPTH + d), nsteps, g);  // the ring slot's next contents
        }
    }
    __syncthreads();

    // ---- the task's own slots -> X / scratch --------------------------------------------------------------------
#pragma unroll
    for (int u = 0; u < MAXG; ++u) {
        const int s = u * RPI + rsel;
        if (u * RPI < nslots && s >= h.n_ext && s < nslots) {
            const td2 v = *reinterpret_cast<const td2 *>(xs + (size_t)s * RB + 2 * sub);
            double *p = (id[u] >= 0 ? X + (int64_t)id[u] * ldx : S + (int64_t)(~id[u]) * lds) + c0 + 2 * sub;
            if (FAST) {
                *reinterpret_cast<td2 *>(p) = v;
            } else {
                if (c0 + 2 * sub < nrhs) p[0] = v.x;
                if (c0 + 2 * sub + 1 < nrhs) p[1] = v.y;
            }
        }
    }
}

}  // namespace dev
}  // namespace rsp
