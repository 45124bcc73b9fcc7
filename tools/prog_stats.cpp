// Dev tool (host only, no GPU): structure of the blocked multi-RHS programme of the C2 factor.
//   g++ -std=c++17 -O2 -I rsparse_amd/csrc/host tools/prog_stats.cpp rsparse_amd/csrc/host/{ordering,symbolic,schedule,blocked}.o -lpthread -o /tmp/prog_stats
//   /tmp/prog_stats [grid=1024] [kind=0|1]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>

#include "parallel.hpp"
#include "sparse.hpp"

using namespace rsp;

int main(int argc, char **argv)
{
    const i64 g = argc > 1 ? atoll(argv[1]) : 1024;
    const int kind = argc > 2 ? atoi(argv[2]) : 0;
    const i64 n = g * g;
    Csc A;
    A.m = A.n = n;
    A.p.push_back(0);
    for (i64 y = 0; y < g; ++y)
        for (i64 x = 0; x < g; ++x) {
            if (y > 0) { A.i.push_back((y - 1) * g + x); A.x.push_back(-1.0); }
            if (x > 0) { A.i.push_back(y * g + x - 1); A.x.push_back(-1.0); }
            A.i.push_back(y * g + x); A.x.push_back(4.0);
            if (x + 1 < g) { A.i.push_back(y * g + x + 1); A.x.push_back(-1.0); }
            if (y + 1 < g) { A.i.push_back((y + 1) * g + x); A.x.push_back(-1.0); }
            A.p.push_back((i64)A.i.size());
        }
    Symbolic S = symbolic_cholesky(A.view(), 0);
    Csc C = symmetric_permute_upper(A.view(), S.pinv.data());
    std::vector<i64> Li;
    if (!cholesky_pattern(C.view(), S.parent, S.cp, Li)) return 1;
    Csc L;
    L.m = L.n = n;
    L.p = S.cp;
    L.i = Li;
    L.x.assign(Li.size(), -0.001);
    for (i64 j = 0; j < n; ++j) L.x[L.p[j]] = 2.0;
    fprintf(stderr, "n %lld nnz(L) %lld\n", (long long)n, (long long)Li.size());
    GatherForm G;
    G.keep_src = true;
    if (build_gather_form(L.view(), kind, G)) return 2;
    BlockProgram P = build_block_program(G, kTaskBlocks, kClusterSlots, false);
    const i64 nlev = (i64)P.level_ptr.size() - 1;
    constexpr int NC = BlockProgram::kClasses;
    fprintf(stderr, "levels %lld blocks %zu clusters %zu cluster_rows %lld scratch_rows %lld big_chains %lld inverted_rows %lld padded %lld coef %zu\n",
            (long long)nlev, P.hdr.size(), P.clusters.size(), (long long)P.cluster_rows, (long long)P.scratch_rows, (long long)P.big_chains,
            (long long)P.inverted_rows, (long long)P.padded_entries, P.coef.size());
    i64 tot[NC] = {}, pad[NC] = {};
    for (i64 l = 0; l < nlev; ++l) {
        i64 cnt[NC], st[NC] = {}, mx = 0;
        i64 runs = 0, run_blocks = 0;  // class 2/3: runs of consecutive blocks with identical source lists
        for (int c = 0; c < NC; ++c) {
            cnt[c] = P.level_class_ptr[NC * l + c + 1] - P.level_class_ptr[NC * l + c];
            const BlockHeader *prev = nullptr;
            for (i64 t = P.level_class_ptr[NC * l + c]; t < P.level_class_ptr[NC * l + c + 1]; ++t)
                for (i64 b = P.task_ptr[t]; b < P.task_ptr[t + 1]; ++b) {
                    const BlockHeader &h = P.hdr[b];
                    st[c] += (i64)h.nsrc * h.rc;
                    if (h.rc == 16) mx = std::max<i64>(mx, h.nsrc / 16);
                    if (false) {
                        bool same = prev && prev->nsrc == h.nsrc;
                        if (same)
                            for (int e = 0; e < h.nsrc && same; ++e)
                                same = P.src_idx[prev->src_off + 32 + e] == P.src_idx[h.src_off + 32 + e];
                        if (!same) ++runs;
                        ++run_blocks;
                        prev = &h;
                    }
                }
            tot[c] += cnt[c];
            pad[c] += st[c];
        }
        printf("L%lld clusters %lld items %lld tasks", (long long)l, (long long)(P.cluster_level_ptr[l + 1] - P.cluster_level_ptr[l]),
               (long long)(P.item_level_ptr[l + 1] - P.item_level_ptr[l]));
        for (int c = 0; c < NC; ++c) printf("%s%lld", c ? "/" : " ", (long long)cnt[c]);
        printf(" padded");
        for (int c = 0; c < NC; ++c) printf("%s%lld", c ? "/" : " ", (long long)st[c]);
        {
            i64 sk = 0, sb = 0, mk = 0, smt = 0;
            for (i64 k = P.item_level_ptr[l]; k < P.item_level_ptr[l + 1]; ++k) {
                const WideItem &I = P.items[k];
                sk += I.k_steps; mk = std::max<i64>(mk, I.k_steps); smt += I.m_tiles;
                const i64 nw = (I.k_steps + 15) / 16;
                for (i64 w = 0; w < nw; ++w) sb += __builtin_popcount(P.item_masks[I.mask_off + 4 * w]) + __builtin_popcount(P.item_masks[I.mask_off + 4 * w + 1]);
            }
            const i64 ni = P.item_level_ptr[l + 1] - P.item_level_ptr[l];
            if (ni) printf(" | items: ksteps sum %lld avg %lld max %lld blocks %lld tiles/item %.1f", (long long)sk, (long long)(sk / ni), (long long)mk, (long long)sb, (double)smt / ni);
        }
        printf(" maxst %lld\n", (long long)mx);
        (void)runs; (void)run_blocks;
    }

    // classification of matrix-core blocks: phase A of a chain (dest = scratch, sources = X), phase B (sources = scratch), normal
    {
        i64 padded[3][8] = {}, real[3][8] = {}, count[3][8] = {};
        for (size_t b = 0; b < P.hdr.size(); ++b) {
            const BlockHeader &h = P.hdr[b];
            if (h.rc != 16) continue;
            const int32_t d0 = P.src_idx[h.src_off];
            const int32_t s0 = h.nsrc ? P.src_idx[h.src_off + 32] : 0;
            const int kind_b = s0 < 0 ? 1 : (d0 < 0 ? 0 : 2);
            int bucket = 0;
            while ((16 << bucket) < h.nsrc && bucket < 7) ++bucket;
            i64 r = 0;
            for (i64 e = 0; e < (i64)h.nsrc * 16; ++e) r += P.refill.valid ? (P.refill.coef_src[h.coef_off + e] != -1) : (P.coef[h.coef_off + e] != 0.0);
            padded[kind_b][bucket] += (i64)h.nsrc * 16;
            real[kind_b][bucket] += r;
            count[kind_b][bucket]++;
        }
        const char *names[3] = {"phaseA(X->scratch)", "phaseB(scratch->X)", "normal"};
        for (int k = 0; k < 3; ++k)
            for (int bk = 0; bk < 8; ++bk)
                if (count[k][bk]) printf("%s nsrc<=%d: blocks %lld padded %lld real %lld\n", names[k], 16 << bk, (long long)count[k][bk], (long long)padded[k][bk], (long long)real[k][bk]);
    }
    printf("total tasks");
    for (int c = 0; c < NC; ++c) printf("%s%lld", c ? "/" : " ", (long long)tot[c]);
    printf(" padded entries");
    for (int c = 0; c < NC; ++c) printf("%s%lld", c ? "/" : " ", (long long)pad[c]);
    i64 item_pad = 0;
    for (const WideItem &I : P.items) item_pad += (i64)I.m_tiles * 16 * I.k_steps * 4;
    printf("\nitems %zu dense entries %lld masked entries %lld; blocks view padded_entries %lld (twins %lld)\n", P.items.size(), (long long)item_pad,
           (long long)P.item_padded_entries, (long long)P.padded_entries, (long long)(pad[2] + pad[4] + pad[6]));
    if (argc > 3) {
        for (int view = 0; view < 2; ++view) {
            const ProgramReport R = check_block_program(G, P, view == 1);
            printf("check (%s view): bad_index %lld order_violations %lld unwritten %lld max_rel_error %.3e\n", view ? "items" : "tiles", (long long)R.bad_index,
                   (long long)R.order_violations, (long long)R.unwritten, R.max_rel_error);
        }
    }
    return 0;
}
