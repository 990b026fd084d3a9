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

#include <algorithm>

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
__device__ __forceinline__ int scan_columns(int32_t *tab, int nbins, int nrows, int blk)
{
    __shared__ int seg_tot[16][64];
    const int tx = threadIdx.x & 63, sy = threadIdx.x >> 6;
    const int bin = blk * 64 + tx;
    const bool live = bin < nbins;
    const int L = (nrows + 15) / 16, c_beg = sy * L, c_end = min(nrows, c_beg + L);
    int32_t *col = tab + (live ? bin : 0);
    // Segments of up to kKeep rows (cfg3: 27 of the grouping pass's table, 20 of the pair bins') stay in registers between the
    // two passes, all their loads in flight at once: the table was written by the launch in front, by workgroups on every XCD,
    // so each round of loads is a round trip past this XCD's L2 (~1 us) - eight at a time and read twice that was 8 of them.
    constexpr int kKeep = 32;
    if (L <= kKeep) {
        int v[kKeep];
        // (one running 32-bit offset from the table: kKeep separate 64-bit addresses spilled under the 128 registers of a
        //  1 024-thread workgroup)
        const unsigned step = (unsigned)nbins, first = (unsigned)c_beg * step + (unsigned)(live ? bin : 0);
        const int nrow = live ? max(c_end - c_beg, 0) : 0;
        unsigned o = first;
#pragma unroll
        for (int u = 0; u < kKeep; ++u) { v[u] = u < nrow ? tab[o] : 0; o += step; }
        int sum = 0;
#pragma unroll
        for (int u = 0; u < kKeep; ++u) sum += v[u];
        seg_tot[sy][tx] = sum;
        __syncthreads();
        int carry = 0;
        for (int q = 0; q < sy; ++q) carry += seg_tot[q][tx];
        o = first;
#pragma unroll
        for (int u = 0; u < kKeep; ++u) {
            if (u < nrow) tab[o] = carry;
            carry += v[u];
            o += step;
        }
        return carry;
    }
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
    return carry;       // (the last segment's threads: the column's total)
}

__global__ __launch_bounds__(1024) void k_struct_scan(StructDev sd) { scan_columns(sd.cntw, sd.nfree * sd.nfree, sd.nchunks, blockIdx.x); }

// ---- the caller's arrays out of mapped host memory into the arena (IngestArgs, device_types.h) ----
// Measured on an MI355X box (scripts/probes/pcie_probe.hip): 64 workgroups with four 16-byte loads per lane in flight take in
// 48 GB/s, what one large copy-engine command gets (51 GB/s) - but the upload's six arrays as six copy commands take 125 us for
// 4 MB (each command costs ~8 us of latency the next one waits behind), two launches of this kernel ~95 us.
constexpr int kIngestGrid = 64, kIngestBlock = 256, kIngestFly = 4;
__global__ __launch_bounds__(kIngestBlock) void k_ingest(IngestArgs a)
{
    if (a.wait_for) {
        // a hint, not a dependency: after ~200 us the launch goes ahead whatever the counter says
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while ((int)(__hip_atomic_load(a.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.wait_for) < 0 &&
               __builtin_amdgcn_s_memrealtime() - t0 < 20000ull) __builtin_amdgcn_s_sleep(8);
    }
    const size_t stride = (size_t)gridDim.x * kIngestBlock;
    for (size_t i = (size_t)blockIdx.x * kIngestBlock + threadIdx.x; i < a.zero_words; i += stride) a.zero[i] = 0u;
    for (int q = 0; q < a.nseg; ++q) {
        const IngestSeg sg = a.seg[q];
        const bool wide = ((reinterpret_cast<size_t>(sg.src) | reinterpret_cast<size_t>(sg.dst)) & 15) == 0;
        const size_t n16 = wide ? sg.bytes / 16 : 0;
        // (non-temporal stores: 3.5 MB of dirty lines in the L2s would be written back by the release at the END of every kernel
        //  that finishes on another stream meanwhile - the grouping and count kernels took 3 - 4 times their time beside a
        //  write-back ingest)
        typedef int v4i __attribute__((ext_vector_type(4)));
        const v4i *src = static_cast<const v4i *>(sg.src);
        v4i *dst = static_cast<v4i *>(sg.dst);
        for (size_t i0 = (size_t)blockIdx.x * kIngestBlock + threadIdx.x; i0 < n16; i0 += stride * kIngestFly) {
            v4i v[kIngestFly];
#pragma unroll
            for (int u = 0; u < kIngestFly; ++u) { const size_t i = i0 + u * stride; if (i < n16) v[u] = src[i]; }
#pragma unroll
            for (int u = 0; u < kIngestFly; ++u) { const size_t i = i0 + u * stride; if (i < n16) __builtin_nontemporal_store(v[u], dst + i); }
        }
        // the tail (and a piece that is not 16-byte aligned as a whole): 4 bytes at a time
        const unsigned *s4 = static_cast<const unsigned *>(sg.src);
        unsigned *d4 = static_cast<unsigned *>(sg.dst);
        for (size_t i = n16 * 4 + (size_t)blockIdx.x * kIngestBlock + threadIdx.x; i < sg.bytes / 4; i += stride) d4[i] = s4[i];
    }
    __syncthreads();
    if (threadIdx.x == 0 && a.counter) atomicAdd(a.counter, 1u);
}

hipError_t launch_ingest(const IngestArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(k_ingest, dim3(kIngestGrid), dim3(kIngestBlock), 0, s, a);
    return hipGetLastError();
}
int ingest_workgroups() { return kIngestGrid; }

// ---- the grouping pass on the device (BasicDev, device_types.h) ----
// One thread per edge, 256 edges per workgroup.  Validation and the grouped-order test are per edge; a point's range starts at
// the first edge whose predecessor belongs to another point; the rank of an edge among its keyframe's edges is assembled from
// three exclusive counts - the workgroups before (H, scanned by k_basic_scan), the waves before inside the workgroup, the
// lanes before inside the wave (one ballot per distinct keyframe of the wave: 64 consecutive edges belong to ~10 neighbouring
// points, hence to a dozen keyframes) - so that it is the host builder's rank whatever the scheduling.
// hessian indices, first pose-major slots and the free-pose list from the edges per keyframe, as build_basic numbers them:
// free keyframes with at least one edge, in caller order.  One workgroup (the LAST one of k_basic_scan to finish); keyframes
// in rounds of its threads.
template <int NT>
__device__ __forceinline__ void basic_index(const BasicDev &bd, const int *pe_lds /* NP */, int *sc /* 2 x (NT / 64) + 4 ints of LDS */)
{
    constexpr int NWV = NT / 64;
    int *wa = sc, *we = sc + NWV, *carry = we + NWV;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) { carry[0] = 0; carry[1] = 0; carry[2] = 0; }
    __syncthreads();
    for (int i0 = 0; i0 < bd.NP; i0 += NT) {
        const int i = i0 + threadIdx.x;
        const bool in = i < bd.NP;
        const int pe = in ? pe_lds[i] : 0;
        const bool fixed = in && bd.pose_fixed[i] != 0;
        const bool act = in && !fixed && pe > 0;
        // inclusive scans over the workgroup: inside a wave by shuffles, the waves' totals through LDS
        int ia = act ? 1 : 0, ie = act ? pe : 0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int ta = __shfl_up(ia, d), te = __shfl_up(ie, d);
            if (lane >= d) { ia += ta; ie += te; }
        }
        if (lane == 63) { wa[wv] = ia; we[wv] = ie; }
        const int nfix = __syncthreads_count(fixed);
        int oa = 0, oe = 0, ta = 0, te = 0;
#pragma unroll
        for (int q = 0; q < NWV; ++q) { const int a = wa[q], e = we[q]; if (q < wv) { oa += a; oe += e; } ta += a; te += e; }
        const int h = carry[0] + oa + ia - (act ? 1 : 0), b = carry[1] + oe + ie - (act ? pe : 0);
        if (in) { bd.hidx[i] = act ? h : -1; bd.base[i] = act ? b : -1; if (act) bd.free_pose[h] = i; }
        __syncthreads();
        if (threadIdx.x == 0) { carry[0] += ta; carry[1] += te; carry[2] += nfix; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { bd.base[bd.NP] = -1; bd.info[2] = carry[0]; bd.info[3] = carry[1]; bd.info[4] = carry[2]; }
}

// H becomes "edges of the keyframe in the workgroups before" (what the slots need), the columns' totals are the edges per
// keyframe, and the keyframes are numbered from them: ONE workgroup for all of it (64 columns per turn), so that nothing has to
// travel between workgroups - a ticket counter with its two device-scope fences cost this kernel 10 of its 17 us (a release
// fence writes back the whole L2 of its XCD on this chip).
__global__ __launch_bounds__(1024) void k_basic_scan(BasicDev bd)
{
    __shared__ int sc[2 * 16 + 4];
    __shared__ int pe_lds[1024];
    for (int g = 0; g * 64 < bd.NP; ++g) {
        const int total = scan_columns(bd.H, bd.NP, bd.nblk, g);
        const int col = g * 64 + (threadIdx.x & 63);
        if ((threadIdx.x >> 6) == 15 && col < bd.NP) { pe_lds[col] = total; bd.pose_edges[col] = total; }
        __syncthreads();                                // (seg_tot of scan_columns is reused by the next turn)
    }
    basic_index<1024>(bd, pe_lds, sc);
}

__global__ __launch_bounds__(kBasicBlock) void k_basic_hist(BasicDev bd)
{
    extern __shared__ int whist[];                      // (kBasicBlock / 64) x NP ints
    __shared__ int slice[kBasicBlock];
    __shared__ int lp_first;
    constexpr int NW = kBasicBlock / 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int e0 = blockIdx.x * kBasicBlock, e = e0 + threadIdx.x;
    for (int k = threadIdx.x; k < NW * bd.NP; k += kBasicBlock) whist[k] = 0;
    // (the index arrays are in the arena: the upload's first copies, which this launch waits for)
    if (threadIdx.x == 0) lp_first = e0 > 0 && e0 < bd.E ? bd.edge_point[e0 - 1] : -1;
    const bool live = e < bd.E;
    int kf = live ? bd.edge_pose[e] : -1;
    const int l = live ? bd.edge_point[e] : 0;
    slice[threadIdx.x] = l;
    __syncthreads();
    bool bad = false;
    if (live && ((unsigned)kf >= (unsigned)bd.NP || (unsigned)l >= (unsigned)bd.P)) { bad = true; kf = -1; }
    if (bad) bd.info[0] = 1;
    if (live && !bad) {
        const int lp = threadIdx.x > 0 ? slice[threadIdx.x - 1] : lp_first;
        if (l < lp) bd.info[1] = 1;                     // not in ascending point order: the host groups such a window itself
        else for (int q = max(lp, -1) + 1; q <= l; ++q) bd.pt_start[q] = e;       // (points nobody observes start where the next one does)
        if (e == bd.E - 1) for (int q = l + 1; q <= bd.P; ++q) bd.pt_start[q] = bd.E;
    }
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
        // (the keyframes' totals are the scan's by-product: 432 workgroups adding into 60 words by device-scope atomics took 35 us)
        bd.H[(size_t)blockIdx.x * bd.NP + k] = t;
    }
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
    hipLaunchKernelGGL(k_basic_scan, dim3(1), dim3(1024), 0, s, bd);
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
