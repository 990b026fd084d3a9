// Latency of dependent operations for a LONE wave on an MI355X compute unit (what every single-workgroup solver of this library is
// made of): cycles (s_memtime) per link of chains of N = 256 links, one workgroup of one wave (and of eight for the barrier).
//   hipcc --offload-arch=gfx950 -O3 -o build/chain_probe scripts/probes/chain_probe.hip && build/chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 256
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double readfirst_f64(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v)
{
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false));
}
__global__ void k_probe(double *out, unsigned long long *cyc, double seed)
{
    __shared__ double lds[1024];
    const int lane = threadIdx.x & 63;
    double x = seed + lane * 1e-3, a = 1.0000001, b = 1e-9;
    unsigned long long t0, t1;
    int q = 0;
    // (the stamps are scalar instructions: each is made to depend on the chain's value through a v_readfirstlane)
#define STAMP(t) { const int d_ = __builtin_amdgcn_readfirstlane(__double2loint(x)); asm volatile("s_nop 7\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "s"(d_) : "memory"); }
#define BEGIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); STAMP(t0); x = __builtin_fma((double)(unsigned)(t0 & 1ull), 1e-300, x);
#define END() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); STAMP(t1); if (threadIdx.x == 0) cyc[q] = t1 - t0; ++q; out[threadIdx.x] += x;
    BEGIN();                                             // 0: empty
    END();
    BEGIN();                                             // 1: dependent fma
#pragma unroll
    for (int i = 0; i < N; ++i) x = __builtin_fma(x, a, b);
    END();
    BEGIN();                                             // 2: readlane (lane 3) -> fma
#pragma unroll
    for (int i = 0; i < N; ++i) x = __builtin_fma(readlane_f64(x, 3), a, x * 1e-30);
    END();
    BEGIN();                                             // 3: readfirstlane -> fma
#pragma unroll
    for (int i = 0; i < N; ++i) x = __builtin_fma(readfirst_f64(x), a, b);
    END();
    BEGIN();                                             // 4: dpp row_newbcast:3 -> fma
#pragma unroll
    for (int i = 0; i < N; ++i) x = __builtin_fma(dpp_f64<0x153>(x), a, b);
    END();
    BEGIN();                                             // 5: rcp_f64 -> fma
#pragma unroll
    for (int i = 0; i < N; ++i) x = __builtin_fma(__builtin_amdgcn_rcp(x), a, 1.5);
    END();
    BEGIN();                                             // 6: rsq_f64 -> fma
#pragma unroll
    for (int i = 0; i < N; ++i) x = __builtin_fma(__builtin_amdgcn_rsq(x), a, 1.5);
    END();
    BEGIN();                                             // 7: LDS store -> wave barrier -> load of the neighbour's
#pragma unroll 8
    for (int i = 0; i < N; ++i) {
        lds[threadIdx.x] = x;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        x = lds[threadIdx.x ^ 1] * a;
    }
    END();
    BEGIN();                                             // 8: __syncthreads
#pragma unroll 8
    for (int i = 0; i < N; ++i) { __syncthreads(); x = x * a; }
    END();
    BEGIN();                                             // 9: v_cmp -> branch on the result (uniform) -> fma
#pragma unroll 8
    for (int i = 0; i < N; ++i) { if (__builtin_amdgcn_readfirstlane(x > 1e300 ? 1 : 0)) x = 1.0; x = __builtin_fma(x, a, b); }
    END();
    BEGIN();                                             // 10: ds_bpermute broadcast of lane 3 -> fma
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int lo = __builtin_amdgcn_ds_bpermute(12, __double2loint(x)), hi = __builtin_amdgcn_ds_bpermute(12, __double2hiint(x));
        x = __builtin_fma(__hiloint2double(hi, lo), a, b);
    }
    END();
    BEGIN();                                             // 11: two independent fma chains interleaved (per pair of links)
    { double y = x + 1.0;
#pragma unroll
      for (int i = 0; i < N; ++i) { x = __builtin_fma(x, a, b); y = __builtin_fma(y, a, b); }
      x += y; }
    END();
}
int main()
{
    double *out; unsigned long long *cyc, h[16];
    hipMalloc((void **)&out, 8 * 512); hipMalloc((void **)&cyc, 8 * 16); hipMemset(out, 0, 8 * 512);
    const char *names[] = { "empty", "fma", "v_readlane -> fma", "v_readfirstlane -> fma", "dpp row_newbcast -> fma", "v_rcp_f64 -> fma", "v_rsq_f64 -> fma",
                            "LDS store, wave barrier, load", "__syncthreads", "compare -> uniform branch -> fma", "ds_bpermute -> fma", "two fma chains side by side" };
    for (int waves : { 1, 8 }) {
        for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(k_probe, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.25); hipDeviceSynchronize(); }
        hipMemcpy(h, cyc, 8 * 16, hipMemcpyDeviceToHost);
        std::printf("%d wave(s) in the workgroup, cycles per link (chain of %d):\n", waves, N);
        for (int q = 1; q < 12; ++q) std::printf("  %-34s %7.1f\n", names[q], (double)(h[q] - h[0]) / N);
    }
    return 0;
}
