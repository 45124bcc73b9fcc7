// HIP kernel (gfx950) of the cluster part of the multi-right-hand-side triangular solve (programme: host/blocked.cpp).
//
// One wavefront owns (cluster, 64 right-hand sides).  A cluster is a set of small dependency subtrees: dozens of
// unknowns with a handful of entries each, every one depending on the previous ones.  Solved through global memory
// that is one HBM/L2 round trip per unknown; here the wavefront
//   1. copies the rows of X the cluster touches (external sources, then its own unknowns) into LDS slots.  Under load
//      a dependent global round trip costs microseconds, so the kernel makes exactly three of them before it computes:
//      header, then ids + first records, then all rows (up to 32 coalesced 512-byte row loads in flight),
//   2. runs the whole dependent chain out of LDS.  The arithmetic arrives as 64-byte group records (four
//      multiply-subtracts each): 16 records (1 KiB) per coalesced vector load, fetched one block ahead of their use
//      and parked in a 1 KiB LDS ring, from which every lane reads the current record at the same address
//      (broadcast).  Per entry the chain costs one ds_read_b64 and one v_fma_f64; the divide is a multiply by the
//      reciprocal diagonal computed on the host,
//   3. writes its own unknowns back, again as whole 512-byte rows.
// Algorithmic traffic only: every row of X is read once and written once, every coefficient read once per 64 columns.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../host/sparse.hpp"

namespace rsp {
namespace dev {

struct ClusterArrays {
    const ClusterHeader *hdr;  // sorted by level
    const int32_t *ids;
    const GroupRec *groups;
};

typedef int v4i __attribute__((ext_vector_type(4)));

struct GroupRegs {  // one GroupRec in registers (identical in every lane)
    v4i q0, q1, q2, q3;
};

__device__ __forceinline__ GroupRegs read_group(const v4i *ring, int j)  // j wave-uniform: broadcast LDS reads
{
    GroupRegs r;
    r.q0 = ring[4 * j + 0];
    r.q1 = ring[4 * j + 1];
    r.q2 = ring[4 * j + 2];
    r.q3 = ring[4 * j + 3];
    return r;
}

__device__ __forceinline__ double lds_get(const double *xs, int byte_off)
{
    return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(xs) + byte_off);
}

// acc: running value of the current unknown; xs_lane = xs + lane (this lane's column of every slot)
__device__ __forceinline__ void run_group(const GroupRegs &r, double *xs_lane, double &acc)
{
    const int fl = __builtin_amdgcn_readfirstlane(r.q3.w);
    const int own = r.q3.z;
    if (fl & GroupRec::kFirst) acc = lds_get(xs_lane, own);
    if (fl & 7) {
        const double x0 = lds_get(xs_lane, r.q0.x), x1 = lds_get(xs_lane, r.q0.y);
        const double x2 = lds_get(xs_lane, r.q0.z), x3 = lds_get(xs_lane, r.q0.w);
        const double c0 = __hiloint2double(r.q1.y, r.q1.x), c1 = __hiloint2double(r.q1.w, r.q1.z);
        const double c2 = __hiloint2double(r.q2.y, r.q2.x), c3 = __hiloint2double(r.q2.w, r.q2.z);
        // two half-sums: a shorter dependent chain than four serial multiply-subtracts
        const double s01 = fma(c1, x1, c0 * x0);
        const double s23 = fma(c3, x3, c2 * x2);
        acc -= s01 + s23;
    }
    if (fl & GroupRec::kLast) {
        acc *= __hiloint2double(r.q3.y, r.q3.x);
        *reinterpret_cast<double *>(reinterpret_cast<char *>(xs_lane) + own) = acc;
    }
}

// One wavefront per (cluster, 64 columns), wavefronts completely independent of each other (no barrier).  A workgroup
// holds the `cpw` column chunks of ONE cluster side by side, so the chip reads and writes whole contiguous rows of X
// at about the same time (DRAM page locality) instead of a 512-byte piece every 2 KiB.
// Dynamic LDS = cpw * (slots * 512 + 1024) bytes.
__global__ __launch_bounds__(256) void solve_clusters(ClusterArrays C, int64_t cluster_begin, int nclusters, int nchunks,
                                                      double *__restrict__ X, int64_t ldx, int nrhs, int slots, int cpw)
{
    extern __shared__ double lds_all[];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    double *xs = lds_all + (size_t)wv * (slots * 64 + 128);
    v4i *ring = reinterpret_cast<v4i *>(xs + (size_t)slots * 64);  // 16 group records
    const int lane = threadIdx.x & 63;
    // XCD-aware placement (blockIdx % 8 share an XCD and its L2): every XCD walks a contiguous range of work items, and
    // items are ordered (group of 16 clusters, chunk group, cluster), so the workgroups an XCD runs together are
    // neighbouring clusters on the same columns -- in a transposed solve they read the same ancestor rows.
    const int per = (nchunks + cpw - 1) / cpw;  // workgroups per cluster
    const int per_xcd = (int)(gridDim.x >> 3);  // the grid is a multiple of 8 workgroups
    const int item = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (item >= nclusters * per) return;
    const int grp = item / (16 * per), r = item - grp * (16 * per);
    const int in_grp = min(16, nclusters - grp * 16);
    const int cg = r / in_grp;
    const int c = grp * 16 + (r - cg * in_grp);
    const int chunk = cg * cpw + wv;
    if (chunk >= nchunks) return;
    const int col = chunk * 64 + lane;
    const bool live = col < nrhs;
    const ClusterHeader h = C.hdr[cluster_begin + c];
    const int32_t *ids = C.ids + h.ids_off;
    const int nslots = h.n_ext + h.n_own;
    const v4i *gsrc = reinterpret_cast<const v4i *>(C.groups + h.grp_off);
    const int ng = h.ngroups, nq = 4 * ng;

    // ONE round trip for everything the header points to: the first 16 records and the row id of every slot
    // (lane s holds the id of slot s; the builder keeps clusters within 64 slots)
    v4i blk = gsrc[min(lane, nq - 1)];
    const int32_t idv = ids[min(lane, nslots - 1)];

    // ONE more for the rows themselves: up to 32 coalesced 512-byte row loads in flight per wavefront
    for (int s0 = 0; s0 < nslots; s0 += 32) {
        double v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            if (s0 + u < nslots) {  // wave-uniform
                const int64_t id = __builtin_amdgcn_readlane(idv, s0 + u);
                v[u] = live ? X[id * ldx + col] : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < 32; ++u)
            if (s0 + u < nslots) xs[(s0 + u) * 64 + lane] = v[u];
    }

    double *xs_lane = xs + lane;
    double acc = 0.0;
    for (int b0 = 0; b0 < ng; b0 += 16) {
        ring[lane] = blk;
        blk = gsrc[min(4 * (b0 + 16) + lane, nq - 1)];  // next block: in flight while this one is executed
        const int nb = min(16, ng - b0);
        GroupRegs cur = read_group(ring, 0);
        for (int j = 0; j < nb; ++j) {
            const GroupRegs nxt = read_group(ring, min(j + 1, nb - 1));
            run_group(cur, xs_lane, acc);
            cur = nxt;
        }
    }

    // the ids are still in registers: the write-back costs no further round trip
    for (int s = h.n_ext; s < nslots; ++s) {
        const int64_t id = __builtin_amdgcn_readlane(idv, s);
        if (live) X[id * ldx + col] = xs[s * 64 + lane];
    }
}

}  // namespace dev
}  // namespace rsp
