// Device side of the structure pass: the per-pair entry lists of the schur kernel —
// for every upper-triangle pose pair (i <= j), the (edge of i, edge of j) couples of the map points both
// keyframes observe, in ascending point order — are counted and filled on the GPU instead of on the host.
//
// What it replaces in the reference is the part of g2o's BlockSolver::buildStructure that lays out the Hpl /
// Hschur block pattern (called from optimizer.initializeOptimization(), /root/reference/src/Optimizer.cc:754);
// on the host this pass cost as much as the whole GPU solve.
//
// One wave handles a chunk of 64 consecutive map points (one per lane).  For every pair bin (i * nf + j) the
// wave builds, in LDS, the 64-bit mask of its points that contribute to the bin (atomic OR: the result does not
// depend on the order of the atomics).  The rank of a point inside its chunk is the popcount of the lower lanes'
// bits, the offset of the chunk inside the pair's list is an exclusive scan of the per-chunk counts, so the entry
// order (pair, then point) is exactly the host builder's and does not depend on scheduling.
#include <hip/hip_runtime.h>

#include "device_types.h"
#include "kernels.h"

namespace movba {

constexpr int kDegCap = 16;     // observers per point whose hessian indices are cached in LDS (longer tracks read the rest from memory)

// bins[b] |= bit of this lane, for every unordered couple of free observers (a <= b) of the lane's point
template <bool FILL>
__global__ __launch_bounds__(64) void k_struct_pairs(StructDev sd)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long masks[];     // nf x nf
    const int lane = threadIdx.x;
    const int chunk = blockIdx.x;
    const int nf = sd.nfree, nbins = nf * nf;
    for (int b = lane; b < nbins; b += 64) masks[b] = 0ull;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int l = chunk * 64 + lane;
    int begin = 0, end = 0;
    if (l < sd.P) { begin = sd.pt_start[l]; end = sd.pt_start[l + 1]; }
    const unsigned long long bit = 1ull << lane;
    // the hessian indices of the lane's first kDegCap observers, gathered ONCE into LDS (the pair loops below would
    // otherwise repeat the dependent g_pose -> hidx loads d^2 / 2 times per point: the kernel was a chain of memory round trips)
    int *hcache = reinterpret_cast<int *>(masks + nbins) + lane * kDegCap;
    for (int k = 0; k < kDegCap && begin + k < end; ++k) hcache[k] = sd.hidx[sd.g_pose[begin + k]];
    auto hof = [&](int e) { const int k = e - begin; return k < kDegCap ? hcache[k] : sd.hidx[sd.g_pose[e]]; };
    for (int a = begin; a < end; ++a) {
        const int ha = hof(a);
        if (ha < 0) continue;
        for (int b = a; b < end; ++b) {
            const int hb = hof(b);
            if (hb < 0) continue;
            if (b != a && hb == ha) { *sd.error = 1; continue; }          // same keyframe observing a point twice
            const int lo = ha < hb ? ha : hb, hi = ha < hb ? hb : ha;
            atomicOr(&masks[lo * nf + hi], bit);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (!FILL) {
        // per-chunk counts, chunk-major: the wave's nbins values leave as contiguous lines (bin-major, every lane wrote a
        // line of its own: 24 MB of write traffic for 3 MB of counts at cfg3), and the fill pass finds its chunk's row in one
        // 10 KB stretch
        for (int b = lane; b < nbins; b += 64) sd.cntw[(size_t)chunk * nbins + b] = __popcll(masks[b]);
    } else {
        const unsigned long long lower = bit - 1ull;
        for (int a = begin; a < end; ++a) {
            const int ha = hof(a);
            if (ha < 0) continue;
            const int sa = sd.slot[a];
            for (int b = a + 1; b < end; ++b) {              // (a, a): diagonal entries are their slot, nothing to store
                const int hb = hof(b);
                if (hb < 0 || hb == ha) continue;
                const int lo = ha < hb ? ha : hb, hi = ha < hb ? hb : ha;
                const int bin = lo * nf + hi;
                const int pos = sd.pair_ptr[sd.pid[bin]] + sd.cntw[(size_t)chunk * nbins + bin] + __popcll(masks[bin] & lower) - sd.n_diag;
                // pose-major slots of the two edges, the one of the lower hessian index first
                const int sb = sd.slot[b];
                sd.ent_i[pos] = (ha <= hb) ? sa : sb;
                sd.ent_j[pos] = (ha <= hb) ? sb : sa;
                sd.ent_l[pos] = l;
            }
        }
    }
}

// exclusive scan over the chunks of every bin's counts (in place) and the bin totals.  A workgroup takes 64 bins (one per
// lane: coalesced across bins) and cuts the chunks into 16 segments, one per wave: segment sums, a 16-step prefix through
// LDS, then the running prefixes written back; 8 loads in flight per lane in both passes.
__global__ __launch_bounds__(1024) void k_struct_scan(StructDev sd)
{
    __shared__ int seg_tot[16][64];
    const int nbins = sd.nfree * sd.nfree;
    const int tx = threadIdx.x & 63, sy = threadIdx.x >> 6;
    const int bin = blockIdx.x * 64 + tx;
    const bool live = bin < nbins;
    const int L = (sd.nchunks + 15) / 16, c_beg = sy * L, c_end = min(sd.nchunks, c_beg + L);
    int32_t *col = sd.cntw + (live ? bin : 0);
    int sum = 0;
    for (int c0 = c_beg; c0 < c_end && live; c0 += 8) {
        int v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = c0 + u < c_end ? col[(size_t)(c0 + u) * nbins] : 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) sum += v[u];
    }
    seg_tot[sy][tx] = sum;
    __syncthreads();
    int carry = 0, total = 0;
    for (int q = 0; q < 16; ++q) { const int t = seg_tot[q][tx]; carry += q < sy ? t : 0; total += t; }
    for (int c0 = c_beg; c0 < c_end && live; c0 += 8) {
        int v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = c0 + u < c_end ? col[(size_t)(c0 + u) * nbins] : 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (c0 + u < c_end) col[(size_t)(c0 + u) * nbins] = carry;
            carry += v[u];
        }
    }
    if (live && sy == 0) sd.cnt[bin] = total;
}

// map point of every pose-major slot (what a diagonal schur entry needs besides its slot)
__global__ __launch_bounds__(256) void k_slot_point(const int32_t *slot, const int32_t *g_point, int32_t *slot_point, int E)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g < E) { const int sl = slot[g]; if (sl >= 0) slot_point[sl] = g_point[g]; }
}

hipError_t launch_slot_point(const int32_t *slot, const int32_t *g_point, int32_t *slot_point, int E, hipStream_t s)
{
    if (E > 0) hipLaunchKernelGGL(k_slot_point, dim3((E + 255) / 256), dim3(256), 0, s, slot, g_point, slot_point, E);
    return hipGetLastError();
}

hipError_t launch_struct_count(const StructDev &sd, hipStream_t s)
{
    const size_t lds = sizeof(unsigned long long) * (size_t)sd.nfree * sd.nfree + sizeof(int) * 64 * kDegCap;
    hipLaunchKernelGGL(k_struct_pairs<false>, dim3(sd.nchunks), dim3(64), lds, s, sd);
    hipLaunchKernelGGL(k_struct_scan, dim3((sd.nfree * sd.nfree + 63) / 64), dim3(1024), 0, s, sd);
    return hipGetLastError();
}

hipError_t launch_struct_fill(const StructDev &sd, hipStream_t s)
{
    const size_t lds = sizeof(unsigned long long) * (size_t)sd.nfree * sd.nfree + sizeof(int) * 64 * kDegCap;
    hipLaunchKernelGGL(k_struct_pairs<true>, dim3(sd.nchunks), dim3(64), lds, s, sd);
    return hipGetLastError();
}

hipError_t configure_struct_kernels()
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_struct_pairs<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k_struct_pairs<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
}

}  // namespace movba
