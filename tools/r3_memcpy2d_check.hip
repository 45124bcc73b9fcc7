// Round 3, ADVICE item 5: is hipMemcpy2D (host to device, pageable host memory) the cause of the wrong block seen under
// rocprofv3 in round 2?  Standalone: no library code.  A column-major n x k host block with leading dimension ld goes to a
// packed device block by hipMemcpy2D, comes back by a plain hipMemcpy and is compared; a kernel touches the block in
// between so that a kernel trace has something to trace.  Run plain and under rocprofv3 (tools/r3_memcpy2d_check.sh).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 2; } } while (0)

__global__ void scale(double *x, size_t count, double f)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) x[i] *= f;
}

int main()
{
    int bad_total = 0;
    const size_t shapes[][3] = {{1048576, 1, 1048576}, {1048576, 1, 1048576 + 64}, {300000, 4, 300000 + 8}, {4096, 7, 5000}, {1048576, 3, 1048576}};
    for (int rep = 0; rep < 3; ++rep)
        for (const auto &sh : shapes) {
            const size_t n = sh[0], k = sh[1], ld = sh[2];
            std::vector<double> h(ld * k), back(n * k);
            for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 1000003) + 0.25 * rep;
            double *d = nullptr;
            CK(hipMalloc((void **)&d, n * k * sizeof(double)));
            CK(hipMemset(d, 0xff, n * k * sizeof(double)));
            CK(hipMemcpy2D(d, n * sizeof(double), h.data(), ld * sizeof(double), n * sizeof(double), k, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(scale, dim3((unsigned)((n * k + 255) / 256)), dim3(256), 0, nullptr, d, n * k, 2.0);
            CK(hipGetLastError());
            CK(hipMemcpy(back.data(), d, n * k * sizeof(double), hipMemcpyDeviceToHost));
            size_t bad = 0, first = 0;
            for (size_t j = 0; j < k; ++j)
                for (size_t i = 0; i < n; ++i)
                    if (back[j * n + i] != 2.0 * h[j * ld + i]) {
                        if (!bad) first = j * n + i;
                        ++bad;
                    }
            std::printf("rep %d  n %zu k %zu ld %zu: %zu wrong entries%s\n", rep, n, k, ld, bad, bad ? " <-- MISMATCH" : "");
            if (bad) std::printf("   first wrong entry %zu: got %.17g, want %.17g\n", first, back[first], 2.0 * h[(first / n) * ld + first % n]);
            bad_total += bad != 0;
            CK(hipFree(d));
        }
    std::printf(bad_total ? "hipMemcpy2D: %d of 15 blocks arrived wrong\n" : "hipMemcpy2D: all 15 blocks arrived intact%.0d\n", bad_total);
    return bad_total ? 1 : 0;
}
