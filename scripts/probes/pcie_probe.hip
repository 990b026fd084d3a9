// PCIe ingest probe: the copy engine (hipMemcpyAsync out of pinned memory) against kernels that read the pinned memory themselves.
//   hipcc --offload-arch=gfx950 -O3 -o build/pcie_probe scripts/probes/pcie_probe.hip && build/pcie_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <int UNROLL>
__global__ __launch_bounds__(256) void k_ingest(const int4 *__restrict__ src, int4 *__restrict__ dst, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < n16; i0 += stride * UNROLL) {
        int4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { const size_t i = i0 + u * stride; if (i < n16) v[u] = src[i]; }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { const size_t i = i0 + u * stride; if (i < n16) dst[i] = v[u]; }
    }
}

int main()
{
    const size_t MB = 1 << 20, total = 8 * MB;
    char *host = nullptr, *hdev = nullptr, *dev = nullptr;
    CK(hipHostMalloc((void **)&host, total, hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void **)&hdev, host, 0));
    CK(hipMalloc((void **)&dev, total));
    std::memset(host, 1, total);
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const size_t sizes[] = { 442040, 884080, 1768160, 3536320, 4420400 };
    for (size_t b : sizes) {
        for (int rep = 0; rep < 3; ++rep) { CK(hipMemcpyAsync(dev, host, b, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); }
        double best = 1e9;
        for (int rep = 0; rep < 10; ++rep) { const double t = now_us(); CK(hipMemcpyAsync(dev, host, b, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); best = std::min(best, now_us() - t); }
        std::printf("DMA   %8zu B: %7.1f us  %5.1f GB/s (enqueue -> synchronised)\n", b, best, b / best / 1e3);
    }
    {   // six copies back to back, as the upload issues them
        const size_t parts[6] = { 442040, 442040, 3360, 480000, 1768160, 884080 };
        double best = 1e9;
        for (int rep = 0; rep < 10; ++rep) {
            const double t = now_us(); size_t off = 0;
            for (size_t p : parts) { CK(hipMemcpyAsync(dev + off, host + off, p, hipMemcpyHostToDevice, s1)); off += (p + 255) & ~(size_t)255; }
            CK(hipStreamSynchronize(s1)); best = std::min(best, now_us() - t);
        }
        std::printf("DMA   six copies (4.02 MB): %7.1f us\n", best);
    }
    const int grids[] = { 64, 128, 256, 512, 1024 };
    for (size_t b : sizes) for (int g : grids) {
        const size_t n16 = b / 16;
        auto run = [&](int unroll) {
            for (int rep = 0; rep < 3; ++rep) { if (unroll == 4) hipLaunchKernelGGL(k_ingest<4>, dim3(g), dim3(256), 0, s1, (const int4 *)hdev, (int4 *)dev, n16); else hipLaunchKernelGGL(k_ingest<8>, dim3(g), dim3(256), 0, s1, (const int4 *)hdev, (int4 *)dev, n16); (void)hipStreamSynchronize(s1); }
            double best = 1e9;
            for (int rep = 0; rep < 10; ++rep) {
                const double t = now_us();
                if (unroll == 4) hipLaunchKernelGGL(k_ingest<4>, dim3(g), dim3(256), 0, s1, (const int4 *)hdev, (int4 *)dev, n16); else hipLaunchKernelGGL(k_ingest<8>, dim3(g), dim3(256), 0, s1, (const int4 *)hdev, (int4 *)dev, n16);
                (void)hipStreamSynchronize(s1); best = std::min(best, now_us() - t);
            }
            return best;
        };
        const double t4 = run(4), t8 = run(8);
        std::printf("KERNEL %8zu B grid %4d: unroll 4 %7.1f us %5.1f GB/s   unroll 8 %7.1f us %5.1f GB/s\n", b, g, t4, b / t4 / 1e3, t8, b / t8 / 1e3);
    }
    {   // two kernels on two streams at once: index arrays (0.88 MB) beside the rest (3.5 MB)
        double best_a = 1e9, best_b = 1e9;
        for (int rep = 0; rep < 10; ++rep) {
            const double t = now_us();
            hipLaunchKernelGGL(k_ingest<4>, dim3(128), dim3(256), 0, s1, (const int4 *)hdev, (int4 *)dev, (size_t)884080 / 16);
            hipLaunchKernelGGL(k_ingest<4>, dim3(256), dim3(256), 0, s2, (const int4 *)(hdev + MB), (int4 *)(dev + MB), (size_t)3536320 / 16);
            (void)hipStreamSynchronize(s1); const double ta = now_us() - t;
            (void)hipStreamSynchronize(s2); const double tb = now_us() - t;
            best_a = std::min(best_a, ta); best_b = std::min(best_b, tb);
        }
        std::printf("KERNEL two streams: 0.88 MB done %7.1f us, 3.5 MB done %7.1f us\n", best_a, best_b);
    }
    return 0;
}
