// Device side of the structure pass: the per-pair entry lists of the schur kernel —
// for every upper-triangle pose pair (i <= j), the (edge of i, edge of j) couples of the map points both
// keyframes observe, in ascending point order — are counted and filled on the GPU instead of on the host.
//
// What it replaces in the reference is the part of g2o's BlockSolver::buildStructure that lays out the Hpl /
// Hschur block pattern (called from optimizer.initializeOptimization(), /root/reference/src/Optimizer.cc:754);
// on the host this pass cost as much as the whole GPU solve.  The fill needs nothing from the host but the edges'
// pose-major slots: it runs while the host lays out the pairs (finish_pairs) from the counts.
//
// One workgroup handles a chunk of 64 consecutive map points (one per lane of each of its waves).  For every pair bin
// (i * nf + j) it builds, in LDS, the 64-bit mask of its points that contribute to the bin (atomic OR: the result does not
// depend on the order of the atomics).  The rank of a point inside its chunk is the popcount of the lower lanes'
// bits, the offset of the chunk inside the pair's list is an exclusive scan of the per-chunk counts, so the entry
// order (pair, then point) is exactly the host builder's and does not depend on scheduling.
#include <hip/hip_runtime.h>

#include "device_types.h"
#include "kernels.h"

namespace movba {

constexpr int kDegCap = 16;     // observers per point whose hessian indices are cached in LDS (longer tracks read the rest from memory)

// bins[b] |= bit of the point, for every unordered couple of free observers (a <= b) of the point.
// One workgroup of kStructWaves waves per chunk: lane l of EVERY wave stands for point l of the chunk, and the waves deal the
// point's couples among themselves (couple number mod kStructWaves): the kernel is a chain of LDS atomics and scattered
// memory operations per couple, so a chunk's time is the longest chain of one lane (one wave per chunk: 39 us for the fill
// at cfg3, with 313 waves on 256 CUs).
constexpr int kStructWaves = 4;

// LDS image of a chunk's workgroup: pair-bin masks, the caches of the points' first kDegCap observers (hessian index,
// and for the fill their pose-major slot), the window's hessian-index table, and for the fill the chunk's row of scanned
// counts and the bins' first entries.  Everything the couple loops look up is in LDS before they start: the dependent
// global loads are pt_start -> (g_pose, slot), two round trips (five before: the launch boundary leaves the caches cold,
// and the kernel was little more than that chain).
__host__ __device__ inline size_t struct_lds_bytes(int nf, int NP, bool fill)
{
    const size_t nbins = (size_t)nf * nf;
    return sizeof(unsigned long long) * nbins + sizeof(int) * (64 * kDegCap * (fill ? 2 : 1) + (size_t)NP + (fill ? 2 * nbins : 0));
}

template <bool FILL>
__global__ __launch_bounds__(64 * kStructWaves) void k_struct_pairs(StructDev sd)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long masks[];     // nf x nf
    constexpr int NT = 64 * kStructWaves;
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int chunk = blockIdx.x;
    const int nf = sd.nfree, nbins = nf * nf;
    int *hcache_all = reinterpret_cast<int *>(masks + nbins);
    int *scache_all = hcache_all + 64 * kDegCap;                            // (fill only)
    int *htab = hcache_all + 64 * kDegCap * (FILL ? 2 : 1);                 // NP
    int *cnt_row = htab + sd.NP, *ent0_row = cnt_row + nbins;               // (fill only)
    const int l = chunk * 64 + lane;
    // (behind a grouping pass of the device that found the caller's edges unusable as they are: the points' ranges were never
    //  written - whatever the arena held before stands there - and the host is about to do the pass itself)
    if (sd.abort && (sd.abort[0] | sd.abort[1])) return;
    int begin = 0, end = 0;
    if (l < sd.P) { begin = sd.pt_start[l]; end = sd.pt_start[l + 1]; }
    for (int b = threadIdx.x; b < nbins; b += NT) masks[b] = 0ull;
    for (int k = threadIdx.x; k < sd.NP; k += NT) htab[k] = sd.hidx[k];
    if (FILL)
        for (int b = threadIdx.x; b < nbins; b += NT) { cnt_row[b] = sd.cntw[(size_t)chunk * nbins + b]; ent0_row[b] = sd.ent0[b]; }
    const unsigned long long bit = 1ull << lane;
    int *hcache = hcache_all + lane * kDegCap, *scache = scache_all + lane * kDegCap;
    int gp_reg[kDegCap / kStructWaves];
#pragma unroll
    for (int u = 0; u < kDegCap / kStructWaves; ++u) {
        const int k = part + u * kStructWaves;
        gp_reg[u] = begin + k < end ? sd.g_pose[begin + k] : -1;
        if (FILL && begin + k < end) scache[k] = sd.slot[begin + k];
    }
    __syncthreads();                                                        // htab is in place
#pragma unroll
    for (int u = 0; u < kDegCap / kStructWaves; ++u) if (gp_reg[u] >= 0) hcache[part + u * kStructWaves] = htab[gp_reg[u]];
    __syncthreads();
    // The couples among a point's first kDegCap observers are walked with the observers' hessian indices (and slots) in
    // registers and the couple loops fully unrolled: couple number k (in a fixed order) belongs to wave k mod kStructWaves.
    // As loops over LDS-resident lists they cost ~100 cycles of LDS latency per couple and lane (8 of the count kernel's
    // 16 us).  Observers past kDegCap (long tracks) take the generic loops below.
    const int deg = end - begin, dcap = deg < kDegCap ? deg : kDegCap;
    int hreg[kDegCap], sreg[kDegCap];
#pragma unroll
    for (int k = 0; k < kDegCap; ++k) { hreg[k] = k < dcap ? hcache[k] : -1; sreg[k] = (FILL && k < dcap) ? scache[k] : 0; }
    auto hof = [&](int e) { const int k = e - begin; return k < kDegCap ? hcache[k] : htab[sd.g_pose[e]]; };
    auto sof = [&](int e) { const int k = e - begin; return k < kDegCap ? scache[k] : sd.slot[e]; };
    const unsigned long long lower = bit - 1ull;
    // one couple (a <= b) of observers with hessian indices ha, hb >= 0
    auto mark = [&](int ha, int hb, bool same_edge) {
        if (!same_edge && hb == ha) { *sd.error = 1; return; }               // same keyframe observing a point twice
        const int lo = ha < hb ? ha : hb, hi = ha < hb ? hb : ha;
        atomicOr(&masks[lo * nf + hi], bit);
    };
    auto emit = [&](int ha, int hb, int sa, int sb) {
        const int lo = ha < hb ? ha : hb, hi = ha < hb ? hb : ha;
        const int bin = lo * nf + hi;
        const int pos = ent0_row[bin] + cnt_row[bin] + __popcll(masks[bin] & lower);
        // pose-major slots of the two edges, the one of the lower hessian index first
        const int si = (ha <= hb) ? sa : sb, sj = (ha <= hb) ? sb : sa;
        if (sd.ent64) sd.ent64[pos] = ent_pack(si, sj, l);            // one 8-byte store instead of three scattered 4-byte ones
        else { sd.ent_i[pos] = si; sd.ent_j[pos] = sj; sd.ent_l[pos] = l; }
    };
#if !defined(MOVBA_STRUCT_SKIP) || MOVBA_STRUCT_SKIP != 1
    {
        int k = 0;
#pragma unroll
        for (int a = 0; a < kDegCap; ++a)
#pragma unroll
            for (int b = a; b < kDegCap; ++b, ++k)
                if ((k & (kStructWaves - 1)) == part && hreg[a] >= 0 && hreg[b] >= 0) mark(hreg[a], hreg[b], a == b);
        if (deg > kDegCap) {
            int turn = 0;
            for (int a = begin; a < end; ++a) {
                const int ha = hof(a);
                if (ha < 0) continue;
                for (int b = max(a, begin + kDegCap); b < end; ++b) {
                    const int hb = hof(b);
                    if (hb < 0) continue;
                    if ((turn++ & (kStructWaves - 1)) != part) continue;
                    mark(ha, hb, a == b);
                }
            }
        }
    }
#endif
    __syncthreads();
    if (!FILL) {
        // per-chunk counts, chunk-major: the chunk's nbins values leave as contiguous lines, and the fill pass finds its
        // chunk's row in one 10 KB stretch
        // ... and the bins' totals by integer atomics (order-independent): the host waits for exactly these, the scan over
        // the chunks (needed by the fill only) runs behind their copy
        for (int b = threadIdx.x; b < nbins; b += NT) {
            const int pc = __popcll(masks[b]);
            sd.cntw[(size_t)chunk * nbins + b] = pc;
            if (pc) atomicAdd(&sd.cnt[b], pc);
        }
    } else {
#if !defined(MOVBA_STRUCT_SKIP) || MOVBA_STRUCT_SKIP != 2
        int k = 0;
#pragma unroll
        for (int a = 0; a < kDegCap; ++a)
#pragma unroll
            for (int b = a + 1; b < kDegCap; ++b, ++k)            // (a, a): diagonal entries are their slot, nothing to store
                if ((k & (kStructWaves - 1)) == part && hreg[a] >= 0 && hreg[b] >= 0 && hreg[a] != hreg[b]) emit(hreg[a], hreg[b], sreg[a], sreg[b]);
        if (deg > kDegCap) {
            int turn = 0;
            for (int a = begin; a < end; ++a) {
                const int ha = hof(a);
                if (ha < 0) continue;
                const int sa = sof(a);
                for (int b = max(a + 1, begin + kDegCap); b < end; ++b) {
                    const int hb = hof(b);
                    if (hb < 0 || hb == ha) continue;
                    if ((turn++ & (kStructWaves - 1)) != part) continue;
                    emit(ha, hb, sa, sof(b));
                }
            }
        }
#endif
    }
}

// exclusive scan over the chunks of every bin's counts (in place).  A workgroup takes 64 bins (one per
// lane: coalesced across bins) and cuts the chunks into 16 segments, one per wave: segment sums, a 16-step prefix through
// LDS, then the running prefixes written back; 8 loads in flight per lane in both passes.
__device__ __forceinline__ void scan_columns(int32_t *tab, int nbins, int nrows)
{
    __shared__ int seg_tot[16][64];
    const int tx = threadIdx.x & 63, sy = threadIdx.x >> 6;
    const int bin = blockIdx.x * 64 + tx;
    const bool live = bin < nbins;
    const int L = (nrows + 15) / 16, c_beg = sy * L, c_end = min(nrows, c_beg + L);
    int32_t *col = tab + (live ? bin : 0);
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
    int carry = 0;
    for (int q = 0; q < sy; ++q) carry += seg_tot[q][tx];
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
}

__global__ __launch_bounds__(1024) void k_struct_scan(StructDev sd) { scan_columns(sd.cntw, sd.nfree * sd.nfree, sd.nchunks); }
// ... and of the grouping pass's edges per (workgroup, keyframe): H becomes "edges of the keyframe in the workgroups before"
__global__ __launch_bounds__(1024) void k_basic_scan(BasicDev bd) { scan_columns(bd.H, bd.NP, bd.nblk); }

hipError_t launch_basic_scan(const BasicDev &bd, hipStream_t s)
{
    hipLaunchKernelGGL(k_basic_scan, dim3((bd.NP + 63) / 64), dim3(1024), 0, s, bd);
    return hipGetLastError();
}

// ---- the grouping pass on the device (BasicDev, device_types.h) ----
// One thread per edge, 256 edges per workgroup.  Validation and the grouped-order test are per edge; a point's range starts at
// the first edge whose predecessor belongs to another point; the rank of an edge among its keyframe's edges is assembled from
// three exclusive counts - the workgroups before (H, scanned by k_basic_scan), the waves before inside the workgroup, the
// lanes before inside the wave (one ballot per distinct keyframe of the wave: 64 consecutive edges belong to ~10 neighbouring
// points, hence to a dozen keyframes) - so that it is the host builder's rank whatever the scheduling.
__global__ __launch_bounds__(kBasicBlock) void k_basic_hist(BasicDev bd)
{
    extern __shared__ int whist[];                      // (kBasicBlock / 64) x NP
    constexpr int NW = kBasicBlock / 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int e = blockIdx.x * kBasicBlock + threadIdx.x;
    for (int k = threadIdx.x; k < NW * bd.NP; k += kBasicBlock) whist[k] = 0;
    const bool live = e < bd.E;
    // (the caller's arrays are read ONCE, across the bus, and left in the arena for every later kernel; a lane's predecessor is
    //  its neighbour's value, the first lane of a wave reads one more word)
    int kf = live ? bd.src_pose[e] : -1, l = live ? bd.src_point[e] : 0;
    int lp = __shfl_up(l, 1);
    if (lane == 0) lp = (live && e > 0) ? bd.src_point[e - 1] : -1;
    if (live) { bd.edge_pose[e] = kf; bd.edge_point[e] = l; }
    bool bad = false;
    if (live) {
        if ((unsigned)kf >= (unsigned)bd.NP || (unsigned)l >= (unsigned)bd.P) { bad = true; kf = -1; }
        else {
            if (l < lp) bd.info[1] = 1;                 // not in ascending point order: the host groups such a window itself
            else for (int q = max(lp, -1) + 1; q <= l; ++q) bd.pt_start[q] = e;       // (points nobody observes start where the next one does)
            if (e == bd.E - 1) for (int q = l + 1; q <= bd.P; ++q) bd.pt_start[q] = bd.E;
        }
    }
    if (bad) bd.info[0] = 1;
    __syncthreads();
    int r = 0;
    unsigned long long todo = __ballot(kf >= 0);
    while (todo) {
        const int lead = __ffsll((long long)todo) - 1;
        const int k = __builtin_amdgcn_readlane(kf, lead);
        const unsigned long long m = __ballot(kf == k);
        if (kf == k) r = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == lead) whist[wv * bd.NP + k] = __popcll(m);
        todo &= ~m;
    }
    __syncthreads();
    if (kf >= 0) {
        for (int q = 0; q < wv; ++q) r += whist[q * bd.NP + kf];
        bd.rank[e] = r;
    } else if (live) bd.rank[e] = 0;
    for (int k = threadIdx.x; k < bd.NP; k += kBasicBlock) {
        int t = 0;
#pragma unroll
        for (int q = 0; q < NW; ++q) t += whist[q * bd.NP + k];
        bd.H[(size_t)blockIdx.x * bd.NP + k] = t;
        if (t) atomicAdd(&bd.pose_edges[k], t);        // (integer totals: order-independent)
    }
}

// hessian indices, first pose-major slots and the free-pose list from the edges per keyframe, as build_basic numbers them:
// free keyframes with at least one edge, in caller order.  One workgroup; keyframes in rounds of 1024.
__global__ __launch_bounds__(1024) void k_basic_index(BasicDev bd)
{
    __shared__ int sc_a[1024], sc_e[1024];
    __shared__ int carry[3];
    if (threadIdx.x == 0) { carry[0] = 0; carry[1] = 0; carry[2] = 0; }
    __syncthreads();
    for (int i0 = 0; i0 < bd.NP; i0 += 1024) {
        const int i = i0 + threadIdx.x;
        const bool in = i < bd.NP;
        const int pe = in ? bd.pose_edges[i] : 0;
        const bool fixed = in && bd.pose_fixed[i] != 0;
        const bool act = in && !fixed && pe > 0;
        sc_a[threadIdx.x] = act ? 1 : 0; sc_e[threadIdx.x] = act ? pe : 0;
        const int nfix = __syncthreads_count(fixed);
        for (int d = 1; d < 1024; d <<= 1) {            // Hillis-Steele, inclusive
            const int va = threadIdx.x >= d ? sc_a[threadIdx.x - d] : 0, ve = threadIdx.x >= d ? sc_e[threadIdx.x - d] : 0;
            __syncthreads();
            sc_a[threadIdx.x] += va; sc_e[threadIdx.x] += ve;
            __syncthreads();
        }
        const int h = carry[0] + sc_a[threadIdx.x] - (act ? 1 : 0), b = carry[1] + sc_e[threadIdx.x] - (act ? pe : 0);
        if (in) { bd.hidx[i] = act ? h : -1; bd.base[i] = act ? b : -1; if (act) bd.free_pose[h] = i; }
        __syncthreads();
        if (threadIdx.x == 1023) { carry[0] += sc_a[1023]; carry[1] += sc_e[1023]; }
        if (threadIdx.x == 0) carry[2] += nfix;
        __syncthreads();
    }
    if (threadIdx.x == 0) { bd.base[bd.NP] = -1; bd.info[2] = carry[0]; bd.info[3] = carry[1]; bd.info[4] = carry[2]; }
}

// map point of every pose-major slot (what a diagonal schur entry needs besides its slot).  With `base` the slot array
// arrives holding each edge's rank among its keyframe's edges (structure.h, build_basic): the slots are completed here.
// (hx: the grouping pass ran on the device, the rank counts inside the edge's workgroup of kBasicBlock edges only: the
//  workgroups before contribute hx[workgroup x NP + keyframe])
__global__ __launch_bounds__(kBasicBlock) void k_slot_point(int32_t *slot, const int32_t *g_pose, const int32_t *base, const int32_t *g_point, int32_t *slot_point, int E,
                                                              const int32_t *hx, int NP)
{
    const int g = blockIdx.x * kBasicBlock + threadIdx.x;
    if (g >= E) return;
    int sl = slot[g];
    if (base) {
        const int kf = g_pose[g];
        const int b = base[kf];
        if (hx && b >= 0) sl += hx[(size_t)blockIdx.x * NP + kf];
        sl = b >= 0 ? b + sl : -1;
        slot[g] = sl;
    }
    if (sl >= 0) slot_point[sl] = g_point[g];
}

hipError_t launch_slot_point(int32_t *slot, const int32_t *g_pose, const int32_t *base, const int32_t *g_point, int32_t *slot_point, int E, const int32_t *hx, int NP, hipStream_t s)
{
    if (E > 0) hipLaunchKernelGGL(k_slot_point, dim3((E + kBasicBlock - 1) / kBasicBlock), dim3(kBasicBlock), 0, s, slot, g_pose, base, g_point, slot_point, E, hx, NP);
    return hipGetLastError();
}

hipError_t launch_basic(const BasicDev &bd, hipStream_t s)
{
    hipLaunchKernelGGL(k_basic_hist, dim3(bd.nblk), dim3(kBasicBlock), sizeof(int) * (kBasicBlock / 64) * (size_t)bd.NP, s, bd);
    hipLaunchKernelGGL(k_basic_index, dim3(1), dim3(1024), 0, s, bd);
    return hipGetLastError();
}

hipError_t launch_struct_count(const StructDev &sd, hipStream_t s)
{
    const size_t lds = struct_lds_bytes(sd.nfree, sd.NP, false);
    hipLaunchKernelGGL(k_struct_pairs<false>, dim3(sd.nchunks), dim3(64 * kStructWaves), lds, s, sd);
    return hipGetLastError();
}

// One workgroup behind the count kernel:
// (1) the bins' totals and the error word straight into the host's pinned buffer, then a sequence number the host polls (a
//     device-to-host copy plus an event wait cost ~15 us more than these stores across the bus);
// (2) the first off-diagonal entry of every pair bin: the pairs are numbered diagonal first, then the off-diagonal bins with
//     entries in row-major order (structure.cpp, finish_pairs), and the off-diagonal entry lists follow that order, so
//     ent0[bin] is the exclusive prefix sum of cnt over the bins (i, j), i < j, taken row-major: every thread sums a run of
//     consecutive bins, the run totals are scanned through LDS.  The fill kernel therefore needs nothing from the host.
// (basic_pe / basic_info: the grouping pass ran on the device too - its edges per keyframe and its info words travel behind the
//  sequence number's place: host_cnt[nbins + 2 ...] = kBasicInfo words, then NP counts)
__global__ __launch_bounds__(1024) void k_struct_counts_out(StructDev sd, int32_t *host_cnt, int seq, const int32_t *basic_pe, const int32_t *basic_info)
{
    __shared__ int tot[1024];
    const int nf = sd.nfree, nbins = nf * nf;
    if (host_cnt) {         // (null on the second pass of a renumbered window: the host has permuted its copy itself)
        for (int b = threadIdx.x; b < nbins; b += 1024) host_cnt[b] = sd.cnt[b];
        if (threadIdx.x == 0) host_cnt[nbins] = *sd.error;
        if (basic_pe) {
            if (threadIdx.x < kBasicInfo) host_cnt[nbins + 2 + threadIdx.x] = basic_info[threadIdx.x];
            for (int k = threadIdx.x; k < sd.NP; k += 1024) host_cnt[nbins + 2 + kBasicInfo + k] = basic_pe[k];
        }
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(host_cnt + nbins + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }

    const int per = (nbins + 1023) / 1024, b0 = threadIdx.x * per, b1 = min(nbins, b0 + per);
    int sum = 0;
    for (int b = b0; b < b1; ++b) sum += (b / nf < b % nf) ? sd.cnt[b] : 0;
    tot[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {           // Hillis-Steele, inclusive
        const int v = threadIdx.x >= d ? tot[threadIdx.x - d] : 0;
        __syncthreads();
        tot[threadIdx.x] += v;
        __syncthreads();
    }
    int run = tot[threadIdx.x] - sum;
    for (int b = b0; b < b1; ++b) {
        sd.ent0[b] = run;
        run += (b / nf < b % nf) ? sd.cnt[b] : 0;
    }
}

hipError_t launch_struct_counts_out(const StructDev &sd, int32_t *host_cnt_dev, int seq, hipStream_t s, const int32_t *basic_pe, const int32_t *basic_info)
{
    hipLaunchKernelGGL(k_struct_counts_out, dim3(1), dim3(1024), 0, s, sd, host_cnt_dev, seq, basic_pe, basic_info);
    return hipGetLastError();
}

hipError_t launch_struct_scan(const StructDev &sd, hipStream_t s)
{
    hipLaunchKernelGGL(k_struct_scan, dim3((sd.nfree * sd.nfree + 63) / 64), dim3(1024), 0, s, sd);
    return hipGetLastError();
}

bool struct_lds_fits(int nfree, int NP) { return struct_lds_bytes(nfree, NP, true) <= 150 * 1024; }

hipError_t launch_struct_fill(const StructDev &sd, hipStream_t s)
{
    const size_t lds = struct_lds_bytes(sd.nfree, sd.NP, true);
    hipLaunchKernelGGL(k_struct_pairs<true>, dim3(sd.nchunks), dim3(64 * kStructWaves), lds, s, sd);
    return hipGetLastError();
}

hipError_t configure_struct_kernels()
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_struct_pairs<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k_struct_pairs<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
}

}  // namespace movba
