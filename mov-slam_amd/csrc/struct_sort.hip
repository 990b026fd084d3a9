// Device side of the structure pass for windows beyond k_struct_pairs' reach (struct_kernels.hip keeps a pair-bin mask per
// chunk of 64 map points in LDS: up to 80 free keyframes).  Same output - for every upper-triangle pose pair (i < j) the
// (slot of i's edge, slot of j's edge, map point) entries of the points both keyframes observe, in ascending point order,
// the pairs in row-major order - by a different route:
//   count   one thread per map point walks the couples of its free observers: the pair bins' totals (integer atomics: the
//           result does not depend on their order) and the point's number of couples;
//   emit    after an exclusive scan over the points, every point writes its couples (key = pair bin, value = the packed
//           entry) at its own offset: the emission order is point order;
//   sort    a STABLE radix sort by pair bin (rocPRIM's radix_sort_pairs, called directly): entries of a bin keep the emission order, i.e. point
//           order, and the bins come out in row-major order - the sorted values ARE the entry lists, written straight
//           into the arena.  Nothing depends on scheduling: bit-identical to the host builder (structure.cpp) whatever the run.
// What it replaces in the reference is the block-pattern part of g2o's BlockSolver::buildStructure (called from
// optimizer.initializeOptimization(), /root/reference/src/Optimizer.cc:754); on the host it cost 3.9 ms of a 9 ms call at
// 150 free keyframes x 60 000 points.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "device_types.h"
#include "kernels.h"

namespace movba {

namespace {

constexpr int kCoupleBlock = 256;

// the hessian indices of a point's first kCache observers, fetched once with all their loads in flight (edge -> keyframe ->
// hessian index: two dependent levels instead of two per couple); observers past the cache are read where they are needed
constexpr int kCache = 16;

__device__ __forceinline__ int obs_h(const StructDev &sd, const int (&hc)[kCache], int b, int g)
{
    const int k = g - b;
    int h = -1;
    if (k < kCache) {
#pragma unroll
        for (int q = 0; q < kCache; ++q) h = k == q ? hc[q] : h;
        return h;
    }
    return sd.hidx[sd.g_pose[g]];
}

__global__ __launch_bounds__(kCoupleBlock) void k_couple_count(StructDev sd, int32_t *cnt_pt)
{
    const int l = blockIdx.x * kCoupleBlock + threadIdx.x;
    if (l >= sd.P) return;
    const int b = sd.pt_start[l], e = sd.pt_start[l + 1], nf = sd.nfree;
    int hc[kCache];
#pragma unroll
    for (int q = 0; q < kCache; ++q) hc[q] = b + q < e ? sd.g_pose[b + q] : -1;
#pragma unroll
    for (int q = 0; q < kCache; ++q) hc[q] = hc[q] >= 0 ? sd.hidx[hc[q]] : -1;
    int f = 0;
    for (int ga = b; ga < e; ++ga) {
        const int ha = obs_h(sd, hc, b, ga);
        if (ha < 0) continue;
        ++f;
        atomicAdd(&sd.cnt[(size_t)ha * nf + ha], 1);
        for (int gb = ga + 1; gb < e; ++gb) {
            const int hb = obs_h(sd, hc, b, gb);
            if (hb < 0) continue;
            if (hb == ha) { *sd.error = 1; continue; }          // a keyframe observes the point twice
            const int lo = ha < hb ? ha : hb, hi = ha < hb ? hb : ha;
            atomicAdd(&sd.cnt[(size_t)lo * nf + hi], 1);
        }
    }
    cnt_pt[l] = f * (f - 1) / 2;
}

__global__ __launch_bounds__(kCoupleBlock) void k_couple_emit(StructDev sd, const int32_t *off, unsigned *keys, unsigned long long *vals)
{
    const int l = blockIdx.x * kCoupleBlock + threadIdx.x;
    if (l >= sd.P) return;
    const int b = sd.pt_start[l], e = sd.pt_start[l + 1], nf = sd.nfree;
    int hc[kCache];
#pragma unroll
    for (int q = 0; q < kCache; ++q) hc[q] = b + q < e ? sd.g_pose[b + q] : -1;
#pragma unroll
    for (int q = 0; q < kCache; ++q) hc[q] = hc[q] >= 0 ? sd.hidx[hc[q]] : -1;
    int k = off[l];
    for (int ga = b; ga < e; ++ga) {
        const int ha = obs_h(sd, hc, b, ga);
        if (ha < 0) continue;
        const int sa = sd.slot[ga];
        for (int gb = ga + 1; gb < e; ++gb) {
            const int hb = obs_h(sd, hc, b, gb);
            if (hb < 0 || hb == ha) continue;
            const int sb = sd.slot[gb];
            const bool up = ha < hb;
            keys[k] = (unsigned)((up ? ha : hb) * nf + (up ? hb : ha));
            vals[k] = ent_pack(up ? sa : sb, up ? sb : sa, l);
            ++k;
        }
    }
}

int key_bits(int nf)
{
    int bits = 1;
    while (((long long)1 << bits) < (long long)nf * nf) ++bits;
    return bits;
}

}  // namespace

hipError_t launch_couple_count(const StructDev &sd, int32_t *cnt_pt, hipStream_t s)
{
    hipLaunchKernelGGL(k_couple_count, dim3((sd.P + kCoupleBlock - 1) / kCoupleBlock), dim3(kCoupleBlock), 0, s, sd, cnt_pt);
    return hipGetLastError();
}

// temporary storage of the scan over P points and of the sort of noff (key, value) pairs, whichever is larger
size_t sorted_fill_temp_bytes(int P, long long noff, int nfree)
{
    size_t a = 0, b = 0;
    (void)rocprim::exclusive_scan(nullptr, a, (const int32_t *)nullptr, (int32_t *)nullptr, (int32_t)0, (size_t)P, rocprim::plus<int32_t>(), (hipStream_t)nullptr);
    (void)rocprim::radix_sort_pairs(nullptr, b, (const unsigned *)nullptr, (unsigned *)nullptr, (const unsigned long long *)nullptr,
                                    (unsigned long long *)nullptr, (size_t)noff, 0u, (unsigned)key_bits(nfree), (hipStream_t)nullptr);
    return (a > b ? a : b) + 256;
}

// sd.ent64 <- the entry lists (sd.slot must hold the completed pose-major slots)
hipError_t launch_sorted_fill(const StructDev &sd, const int32_t *cnt_pt, int32_t *off, unsigned *keys_in, unsigned *keys_out,
                              unsigned long long *vals_in, void *tmp, size_t tmp_bytes, long long noff, hipStream_t s)
{
    if (noff <= 0) return hipSuccess;
    size_t tb = tmp_bytes;
    hipError_t e = rocprim::exclusive_scan(tmp, tb, cnt_pt, off, (int32_t)0, (size_t)sd.P, rocprim::plus<int32_t>(), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_couple_emit, dim3((sd.P + kCoupleBlock - 1) / kCoupleBlock), dim3(kCoupleBlock), 0, s, sd, off, keys_in, vals_in);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    tb = tmp_bytes;
    return rocprim::radix_sort_pairs(tmp, tb, keys_in, keys_out, vals_in, sd.ent64, (size_t)noff, 0u, (unsigned)key_bits(sd.nfree), s);
}

}  // namespace movba
