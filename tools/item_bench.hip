// Dev microbenchmark (round 3): the wide-item kernel (device/item_kernels.hip) on synthetic "heavy level" items.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/item_bench.hip -o build/item_bench
//   ./item_bench <nitems> <K sources> <group (items sharing one source list)> <ncols> <ksplit> <m_tiles> <percent of K-steps with a partial mask>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../rsparse_amd/csrc/device/item_kernels.hip"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

using namespace rsp;

int main(int argc, char **argv)
{
    const int nitems = argc > 1 ? atoi(argv[1]) : 2048;
    const int K = argc > 2 ? atoi(argv[2]) : 256;
    const int group = argc > 3 ? atoi(argv[3]) : 8;
    const int ncols = argc > 4 ? atoi(argv[4]) : 256;
    const int ksplit = argc > 5 ? atoi(argv[5]) : 1;
    const int mt = argc > 6 ? atoi(argv[6]) : 4;
    const int partial = argc > 7 ? atoi(argv[7]) : 0;
    const int64_t span = argc > 8 ? atoll(argv[8]) : 100000;
    const int64_t n = 1 << 20, ld = ncols;
    const int ks = K / 4, nwin = (ks + 15) / 16;
    std::mt19937_64 rng(7);
    std::vector<WideItem> items(nitems);
    std::vector<int32_t> ids;
    std::vector<uint32_t> masks;
    std::vector<double> coef;
    std::vector<int32_t> srcs;
    double mfmas = 0;
    for (int i = 0; i < nitems; ++i) {
        if (i % group == 0) {
            srcs.clear();
            const int64_t base = rng() % (n - span);
            for (int e = 0; e < K; ++e) srcs.push_back((int32_t)(base + rng() % span));
            std::sort(srcs.begin(), srcs.end());
        }
        WideItem &I = items[i];
        I.m_tiles = mt;
        I.k_steps = ks;
        I.coef_off = (int64_t)coef.size();
        I.ids_off = (int64_t)ids.size();
        I.mask_off = (int64_t)masks.size();
        for (int r = 0; r < 16 * mt; ++r) ids.push_back(~(int32_t)((int64_t)i * 64 + r));  // scratch rows
        for (int r = 0; r < 16 * mt; ++r) ids.push_back((int32_t)(rng() % n));
        ids.insert(ids.end(), srcs.begin(), srcs.end());
        int blocks = 0;
        for (int w = 0; w < nwin; ++w) {
            uint64_t m = 0;
            const int first = blocks;
            for (int k = 16 * w; k < std::min(ks, 16 * w + 16); ++k) {
                unsigned bits = (1u << mt) - 1u;
                if ((int)(rng() % 100) < partial) bits = 1u + (unsigned)(rng() % ((1u << mt) - 1u));
                m |= (uint64_t)bits << (4 * (k - 16 * w));
                blocks += __builtin_popcount(bits);
            }
            masks.push_back((uint32_t)m);
            masks.push_back((uint32_t)(m >> 32));
            masks.push_back((uint32_t)first);
            masks.push_back(0);
        }
        mfmas += blocks;
        for (int b = 0; b < blocks * 64; ++b) coef.push_back((double)(rng() % 1000) * 1e-6);
    }
    ids.resize(ids.size() + 256, 0);
    coef.resize(coef.size() + 320, 0.0);
    double *dX, *dS, *dcoef;
    int32_t *dids;
    uint32_t *dmasks;
    WideItem *ditems;
    const size_t srows = (size_t)nitems * 64 + 1;
    CK(hipMalloc(&dX, (size_t)n * ld * 8));
    CK(hipMalloc(&dS, srows * ld * 8));
    CK(hipMalloc(&dcoef, coef.size() * 8));
    CK(hipMalloc(&dids, ids.size() * 4));
    CK(hipMalloc(&dmasks, masks.size() * 4));
    CK(hipMalloc(&ditems, items.size() * sizeof(WideItem)));
    std::vector<double> hx((size_t)n * ld);
    for (auto &v : hx) v = (double)(rng() % 2001) * 1e-3 - 1.0;
    CK(hipMemcpy(dX, hx.data(), hx.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(dS, 0, srows * ld * 8));
    CK(hipMemcpy(dcoef, coef.data(), coef.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dmasks, masks.data(), masks.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(ditems, items.data(), items.size() * sizeof(WideItem), hipMemcpyHostToDevice));
    const int nchunks = ncols / 64;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const dev::ItemArrays IA{ditems, dids, dcoef, dmasks};
    auto launch = [&]() { dev::launch_items(ksplit, IA, 0, nitems, nchunks, dX, ld, dS, ld, 0); };
    for (int w = 0; w < 3; ++w) launch();
    CK(hipDeviceSynchronize());
    const int reps = 10;
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double flops = 2.0 * mfmas * 64 * ncols;
    std::vector<double> hs(srows * ld);
    CK(hipMemcpy(hs.data(), dS, hs.size() * 8, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int it : {0, 3, nitems - 1})
        for (int r : {0, 5, 16 * mt - 1})
            for (int col : {0, 1, 33, 63, ncols - 1}) {
                const WideItem &I = items[it];
                double want = hx[(size_t)ids[I.ids_off + 16 * mt + r] * ld + col];
                int blk = 0;
                for (int k = 0; k < ks; ++k) {
                    const unsigned bits = (masks[I.mask_off + 4 * (k / 16) + ((k % 16) >= 8)] >> (4 * (k % 8))) & 15u;
                    for (int q = 0; q < mt; ++q) {
                        if (!((bits >> q) & 1u)) continue;
                        if (q == r / 16)
                            for (int g = 0; g < 4; ++g)
                                want -= coef[I.coef_off + (size_t)blk * 64 + g * 16 + r % 16] * hx[(size_t)ids[I.ids_off + 32 * mt + 4 * k + g] * ld + col];
                        ++blk;
                    }
                }
                const double got = hs[(size_t)(~ids[I.ids_off + r]) * ld + col];
                worst = std::max(worst, std::abs(got - want));
            }
    printf("ksplit %d items %d mt %d K %d group %d cols %d partial %d%%: %.1f us  %.1f TFLOP/s (stored blocks)  max abs error on samples %.2e\n", ksplit, nitems, mt,
           K, group, ncols, partial, ms * 1e3, flops / ms / 1e9, worst);
    return worst < 1e-9 ? 0 : 1;
}
