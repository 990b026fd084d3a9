// k_dense_persist: the direct solve of the reduced camera system in ONE launch.
//
// The reference's step is an exact sparse Cholesky (LinearSolverCSparse, /root/reference/src/Optimizer.cc:535; a failed
// factorisation rejects the LM trial).  dense_solve.hip does the same arithmetic with one launch per 48-wide block column
// (28 us each, whatever the size); here the whole solve - assembly of S from the schur partials, blocked Cholesky,
// forward and back substitution, pose update - is one launch of up to 248 workgroups that hand finished tiles to each
// other through L2:
//   * every 48 x 48 tile of the lower block triangle has an OWNER workgroup (dense_plan.h) that keeps it in LDS from
//     assembly to the end: tile (I, K) -= L(I, k) L(K, k)^T for every block column k < K as soon as that column's tiles
//     are published (fp64 matrix cores);
//   * workgroup K owns the diagonal tile (K, K) and the one to its left.  It factors D_K ONCE, over the identity
//     (sweep_inverse: [D; I] -> [L; L^-T]), and publishes the inverse factor W_K = L(K, K)^-T: the tiles below take
//     L(I, K) = tile W_K as a matrix product, the forward substitution is y_K = W_K^T r_K, the back substitution
//     x_J = W_J (y_J - sum c(I, J)).  L(K, K) itself is never stored.  The dependent chain
//         W_(K-1) -> L(K, K-1) -> D_K -> W_K
//     crosses workgroups once per block column and is one task of the owner (DT_COL);
//   * the right-hand side row r_K follows the factorisation one block column behind (DT_RUP), the back substitution runs
//     from the last block row up with one 48-vector per tile in flight, the one on its chain as tagged records;
//   * hand-offs of tiles follow cdna_hip_programming.md Guideline 16 in its write-through form: payload by sc1 stores, every
//     storing wave drains (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane stores the flag (= the launch's epoch, so flags
//     are never reset between launches); the consumer's wave 0 polls the flag with sc1 loads, the workgroup barrier behind
//     the match releases the other waves, and EVERY load of handed-off bytes is an sc1 load (no acquire fence needed: one
//     workgroup per CU, hipMalloc memory, 8- / 16-byte accesses - the first row of MI355X_MICROARCH.md's hand-off table);
//   * every wait is bounded by the 100 MHz clock: a workgroup that waits 20 ms gives up, marks the solve failed (the LM
//     trial is rejected, the download returns MOVBA_ERR_DEVICE_WAIT through Ctrl::n_sync_timeouts) and leaves; so do the
//     others.  The host keeps two such launches from sharing the device (DenseGate, api.cpp).
// Fixed summation order everywhere (each tile has one owner, updates in column order): bit-reproducible run to run.
#include <hip/hip_runtime.h>

#include "dense_plan.h"
#include "dense_tile.h"
#include "device_math.h"
#include "device_types.h"
#include "kernels.h"

namespace movba {

using namespace dense;

namespace {

constexpr int kPT = 256;                        // threads: 4 waves, one per SIMD
constexpr int kTileLds = NB * LD;               // doubles of one LDS tile image
constexpr int kTaskCache = 128;                 // tasks of the workgroup's list held in LDS at a time (32 bytes each)
__host__ __device__ constexpr int kTaskOff(int nslots) { return (2 + nslots) * kTileLds + 672; }       // (doubles) behind the carve
__host__ __device__ constexpr int kBadOff(int nslots) { return (2 + nslots) * kTileLds + 660; }     // LDS word of sweep_inverse's bad-pivot flag (in the carve's spare doubles)

// every word that travels between workgroups is a GLOBAL agent-scope access (global_load / global_store ... sc1, never flat_)
typedef __attribute__((address_space(1))) long long g_i64;
typedef __attribute__((address_space(1))) unsigned g_u32;
__device__ __forceinline__ double ld_sc1(const double *p)
{
    return __longlong_as_double(__hip_atomic_load((const g_i64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_sc1(double *p, double v)
{
    __hip_atomic_store((g_i64 *)p, __double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned ld_flag(const unsigned *p) { return __hip_atomic_load((const g_u32 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_flag(unsigned *p, unsigned v) { __hip_atomic_store((g_u32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct Lds {
    double *A, *B;          // two scratch tiles (MFMA operands; the sweeps' column table spans both)
    double *slots;          // the workgroup's own tiles
    double *rrow, *yv, *xv, *xs, *ps, *pivb, *rinvb, *red;
    lds_vint *prog;         // pivots published by the factorising wave
    int *abort;
    unsigned long long wait_ticks;      // bound of every wait (DevWindow::wait_ticks: 20 ms; a test hook shortens it)
};

__device__ __forceinline__ Lds carve(double *sm, int nslots, unsigned long long wait_ticks)
{
    Lds l;
    l.wait_ticks = wait_ticks;
    l.A = sm; l.B = sm + kTileLds; l.slots = sm + 2 * kTileLds;
    double *p = l.slots + (size_t)nslots * kTileLds;
    l.rrow = p; l.yv = p + 64; l.xv = p + 128; l.xs = p + 192; l.ps = p + 256; l.pivb = p + 512; l.rinvb = p + 576; l.red = p + 640;
    l.prog = (lds_vint *)(p + 656);
    l.abort = reinterpret_cast<int *>(p + 657);
    return l;
}

// global tile (row-major NB x NB) <-> LDS image (row stride LD): 16-byte sc1 (write-through / L1-bypassing) accesses through a
// buffer descriptor of the tile, all of a thread's accesses in flight (a tile is 1 152 x 16 bytes: 4.5 per thread)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kTileBytes = NB * NB * 8, kTileVec = kTileBytes / 16;
// TRI = 2: an upper triangular tile (the inverse factor W_K) travels without its three 16 x 16 blocks below the diagonal
// (TRI = 1: without those above)
template <int TRI>
__device__ __forceinline__ bool tri_skip(int row, int c) { return TRI == 1 ? (c >> 4) > (row >> 4) : (TRI == 2 ? (c >> 4) < (row >> 4) : false); }
template <int TRI = 0>
__device__ __forceinline__ void fetch_tile(const double *g, double *lds, int tid)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(g), 0, kTileBytes, 0x00020000);
    u32x4 v[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int e = tid + kPT * q, row = (2 * e) / NB, c = 2 * e - row * NB;
        const bool want = e < kTileVec && !tri_skip<TRI>(row, c);
        v[q] = __builtin_amdgcn_raw_buffer_load_b128(r, want ? e * 16 : kTileBytes, 0, 16);       // (past the descriptor's range: no memory access)
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int e = tid + kPT * q;
        if (e < kTileVec) {
            const int row = (2 * e) / NB, c = 2 * e - row * NB;
            if (tri_skip<TRI>(row, c)) continue;
            lds[row * LD + c] = __hiloint2double((int)v[q].y, (int)v[q].x); lds[row * LD + c + 1] = __hiloint2double((int)v[q].w, (int)v[q].z);
        }
    }
}

template <int TRI = 0>
__device__ __forceinline__ void publish_tile(double *g, const double *lds, int tid)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(g, 0, kTileBytes, 0x00020000);
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int e = tid + kPT * q;
        if (e < kTileVec) {
            const int row = (2 * e) / NB, c = 2 * e - row * NB;
            if (tri_skip<TRI>(row, c)) continue;
            const double a = lds[row * LD + c], b = lds[row * LD + c + 1];
            const u32x4 v = { (unsigned)__double2loint(a), (unsigned)__double2hiint(a), (unsigned)__double2loint(b), (unsigned)__double2hiint(b) };
            __builtin_amdgcn_raw_buffer_store_b128(v, r, e * 16, 0, 16);
        }
    }
}

// A 48-vector handed over as 48 self-announcing 16-byte records (value, the launch's epoch, a check word): one sc1 store per
// lane, no drain, no flag; the consumer's lanes poll their own record until epoch and check word fit.  (A 16-byte store of one
// lane is one request to one cache line; the check word is there so that a torn read could only ever cause another look.)
constexpr unsigned kTagSalt = 0x9e3779b9u;
__device__ __forceinline__ void st_tagged(unsigned *rec_base, int nrec_bytes, int t, double v, unsigned epoch)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(rec_base, 0, nrec_bytes, 0x00020000);
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const u32x4 q = { lo, hi, epoch, epoch ^ lo ^ hi ^ kTagSalt };
    __builtin_amdgcn_raw_buffer_store_b128(q, r, t * 16, 0, 16);
}
// wave 0, lanes below 48: false (to the whole workgroup, through Lds::abort) when the records did not come within kWaitTicks
__device__ __forceinline__ bool ld_tagged(const unsigned *rec_base, int nrec_bytes, int tid, unsigned epoch, double &v, const Lds &l)
{
    if (tid < 64) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(rec_base), 0, nrec_bytes, 0x00020000);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const bool mine = tid < NB;
        bool good = true;
        u32x4 q;
        for (;;) {
            q = __builtin_amdgcn_raw_buffer_load_b128(r, mine ? tid * 16 : 0, 0, 16);
            const bool ok = q.z == epoch && q.w == (epoch ^ q.x ^ q.y ^ kTagSalt);
            if (__all(ok || !mine)) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > l.wait_ticks) { good = false; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        v = __hiloint2double((int)q.y, (int)q.x);
        if (!good && tid == 0) *l.abort = 1;
    }
    __syncthreads();
    return *l.abort == 0;
}

// every storing wave drains its stores, the workgroup meets, ONE lane stores the flag
__device__ __forceinline__ void set_flag(unsigned *flags, int idx, unsigned epoch, int tid)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) st_flag(flags + idx, epoch);
}

// wave 0 polls n flags (lane i the i-th, 64 per pass) until all carry the epoch; the barrier behind releases the others.
// false: gave up after kWaitTicks (the workgroup then leaves the solve)
template <typename IdxFn>
__device__ __forceinline__ bool wg_wait(const unsigned *flags, unsigned epoch, int n, IdxFn idx, const Lds &l, int tid)
{
    if (tid < 64) {
        bool good = true;
        for (int base = 0; base < n && good; base += 64) {
            const int i = base + tid;
            const bool mine = i < n;
            const unsigned *f = flags + idx(mine ? i : base);
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                const bool ok = ld_flag(f) == epoch;
                if (__all(ok || !mine)) break;
                if (__builtin_amdgcn_s_memrealtime() - t0 > l.wait_ticks) { good = false; break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        if (!good && tid == 0) *l.abort = 1;
    }
    __syncthreads();
    return *l.abort == 0;
}

// two flags, the first one awaited: 0 = gave up, 1 = the first is set, 3 = both are (the caller fetches both operands at once;
// otherwise the first operand travels while the second is still being produced)
__device__ __forceinline__ int wg_wait2(const unsigned *flags, unsigned epoch, int fa, int fb, const Lds &l, int tid)
{
    if (tid < 64) {
        const unsigned *f = flags + (tid == 1 ? fb : fa);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned long long m;
        bool good = true;
        for (;;) {
            m = __ballot(ld_flag(f) == epoch);
            if (m & 1) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > l.wait_ticks) { good = false; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (tid == 0) { if (!good) *l.abort = 1; l.abort[1] = (int)(m & 3); }
    }
    __syncthreads();
    return *l.abort == 0 ? l.abort[1] : 0;
}

// r -= L y for one 48 x 48 tile (LDS image) and y in xs: five column groups, combined in fixed order
__device__ __forceinline__ void rhs_minus_tile_times(const double *L, const Lds &l, int tid)
{
    const int g = tid / NB, cc = tid - g * NB;
    if (g < 5) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 10; ++q) { const int col = g + 5 * q; if (col < NB) s += L[cc * LD + col] * l.xs[col]; }
        l.ps[g * NB + cc] = s;
    }
    __syncthreads();
    if (tid < NB) l.rrow[tid] -= (((l.ps[tid] + l.ps[NB + tid]) + l.ps[2 * NB + tid]) + l.ps[3 * NB + tid]) + l.ps[4 * NB + tid];
    __syncthreads();
}

// tile <- tile W with W upper triangular (blocks (kb, cb), kb <= cb): wave rb < 3 takes row block rb, all three products in
// registers before the first of them is written back (in place: the row block is read and written by this wave only)
__device__ __forceinline__ void tile_times_upper(double *U, const double *W, int wv, int lane)
{
    if (wv >= 3) return;
    const double *ap = U + (wv * 16 + (lane & 15)) * LD + (lane >> 4);          // A[i = lane & 15][k = lane >> 4]
    const double *bp = W + (lane >> 4) * LD + (lane & 15);                      // B[k = lane >> 4][j = lane & 15]
    // every operand in registers first (36 LDS reads in flight), then the 24 matrix instructions back to back
    double a[12], b0[4], b1[8], b2[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) { a[q] = ap[4 * q]; b2[q] = bp[4 * q * LD + 32]; }
#pragma unroll
    for (int q = 0; q < 8; ++q) b1[q] = bp[4 * q * LD + 16];
#pragma unroll
    for (int q = 0; q < 4; ++q) b0[q] = bp[4 * q * LD];
    dbl4 x0 = dbl4{ 0.0, 0.0, 0.0, 0.0 }, x1 = x0, x2 = x0;
#pragma unroll
    for (int q = 0; q < 12; ++q) {
        if (q < 4) x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b0[q], x0, 0, 0, 0);
        if (q < 8) x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b1[q], x1, 0, 0, 0);
        x2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b2[q], x2, 0, 0, 0);
    }
    double *cp = U + (wv * 16 + (lane >> 4)) * LD + (lane & 15);                // C: col = lane & 15, row = (lane >> 4) + 4 reg
    cp[0] = x0.x; cp[4 * LD] = x0.y; cp[8 * LD] = x0.z; cp[12 * LD] = x0.w;
    cp[16] = x1.x; cp[4 * LD + 16] = x1.y; cp[8 * LD + 16] = x1.z; cp[12 * LD + 16] = x1.w;
    cp[32] = x2.x; cp[4 * LD + 32] = x2.y; cp[8 * LD + 32] = x2.z; cp[12 * LD + 32] = x2.w;
}

// out[c] = sum_r T[r][c] v[r] (TRANSPOSED) or sum_c' T[c][c'] v[c'] for a 48 x 48 LDS tile and v in xs: five groups of 48
// threads, partial sums combined in fixed order; the result to every thread below 48 (after the barriers inside)
template <bool TRANSPOSED>
__device__ __forceinline__ double tile_matvec(const double *T, const Lds &l, int tid)
{
    const int g = tid / NB, cc = tid - g * NB;
    if (g < 5) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 10; ++q) { const int r = g + 5 * q; if (r < NB) s += (TRANSPOSED ? T[r * LD + cc] : T[cc * LD + r]) * l.xs[r]; }
        l.ps[g * NB + cc] = s;
    }
    __syncthreads();
    return tid < NB ? (((l.ps[tid] + l.ps[NB + tid]) + l.ps[2 * NB + tid]) + l.ps[3 * NB + tid]) + l.ps[4 * NB + tid] : 0.0;
}

// slot -= A B^T, the nine 16 x 16 MFMA tiles dealt round-robin to the four waves; LOWER (a diagonal tile, A == B): the six
// tiles on and below the diagonal only
template <bool LOWER>
__device__ __forceinline__ void tile_update(double *C, const double *As, const double *Bs, int wv, int lane, double *Cout = nullptr)
{
    if (!Cout) Cout = C;                // (the last update of a diagonal tile lands in scratch tile B, where the sweep takes it from)
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int q = wv + 4 * u;
        if (q < (LOWER ? 6 : 9)) {
            // (lower: tiles (0,0) (1,0) (1,1) (2,0) (2,1) (2,2) in that order)
            const int mt = LOWER ? (q >= 3 ? 2 : (q >= 1 ? 1 : 0)) : q / 3, ntc = LOWER ? q - mt * (mt + 1) / 2 : q - mt * 3;
            const int off = (mt * 16 + (lane >> 4)) * LD + ntc * 16 + (lane & 15);
            const double *cp = C + off;
            dbl4 acc = dbl4{ cp[0], cp[4 * LD], cp[8 * LD], cp[12 * LD] };
            acc = tile_mfma(As, Bs, mt, ntc, lane, acc);
            double *op = Cout + off;
            op[0] = acc.x; op[4 * LD] = acc.y; op[8 * LD] = acc.z; op[12 * LD] = acc.w;
        }
    }
}

// S (damped) of tile (I, K) from the schur work-item partials into an LDS image: the element map and the item order of
// k_dense_assemble (dense_solve.hip), so both direct solvers and the PCG see the same matrix to the last bit
struct AsmView { const double *part; const int32_t *prange; double *bp; int32_t nf, n; unsigned long long *st; };      // (by value: the kernel's argument block stays out of scratch)
__device__ __attribute__((noinline)) void assemble_tile(AsmView av, int I, int K, double lambda, int dst_off, int rrow_off, int tid)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];     // (LDS addressed from its own symbol: ds_ instructions, not flat_)
    double *dst = sm + dst_off;
    const int nf = av.nf, n = av.n;
    // (global pointers said to be global: plain global_load instead of flat_load)
    typedef const __attribute__((address_space(1))) double *gdp;
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    typedef const __attribute__((address_space(1))) i32x2 *grp;
    const gdp part = (gdp)av.part;
    const grp prange = (grp)av.prange;
    // Two levels of dependent loads in all (these come right behind a launch boundary: ~2.5 us each, cold): the item ranges
    // of the thread's nine elements and of the tile's diagonal pairs, then the partial records themselves.  Everything is
    // loaded unconditionally from an address that is always valid and masked afterwards: written with a branch per element
    // the compiler serialised the nine round trips of every level (11 us per diagonal tile).
    constexpr int kPer = NB * NB / kPT;
    int kk[kPer], uu[kPer], i0[kPer], i1[kPer];
    bool live[kPer];                    // the element lies inside the system (not in the padding of the last tile)
    double pad[kPer];
    i32x2 rg[kPer];
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
        const int e = tid + kPT * q, r = e / NB, cc = e - r * NB;
        const int gr = I * NB + r, gc = K * NB + cc;
        live[q] = gr < n && gc < n;
        const int bi = live[q] ? gr / 6 : 0, bj = live[q] ? gc / 6 : 0;
        const int a = gr - (gr / 6) * 6, b = gc - (gc / 6) * 6;
        const int lo = bi < bj ? bi : bj, hi = bi < bj ? bj : bi;
        rg[q] = prange[(size_t)lo * nf + hi];
        kk[q] = bi <= bj ? a * 6 + b : b * 6 + a;
        const bool dg = live[q] && bi == bj;
        uu[q] = dg ? 42 + (a <= b ? ut6(a, b) : ut6(b, a)) : -1;
        pad[q] = live[q] ? (dg && a == b ? lambda : 0.0) : (gr == gc ? 1.0 : 0.0);
    }
    // A diagonal tile holds eight diagonal pairs, each cut into several work items (four or five at cfg3) whose records are
    // summed for 57 of the tile's elements and for the right-hand side: their records (contiguous: the diagonal pairs come
    // first, in order) are staged in the two scratch tiles by one coalesced pass and summed from LDS.
    int item0 = 0, nstage = 0;
    const int b0 = K * (NB / 6), b1 = min(b0 + NB / 6, nf);
    if (I == K) {
        item0 = prange[(size_t)b0 * nf + b0].x;
        nstage = prange[(size_t)(b1 - 1) * nf + (b1 - 1)].y - item0;
        if (nstage * kPartStride > 2 * kTileLds) nstage = 0;       // (a keyframe with tens of thousands of edges: summed from memory)
    }
    bool staged[kPer];
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
        i0[q] = live[q] ? rg[q].x : 0; i1[q] = live[q] ? rg[q].y : 0;
        staged[q] = uu[q] >= 0 && nstage > 0;
    }
    if (av.st && tid == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); av.st[1] = __builtin_amdgcn_s_memrealtime(); }
    // the first item of every element that is not staged, issued beside the staging pass
    double sacc[kPer], hpp[kPer];
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
        const size_t base = (size_t)(staged[q] ? 0 : i0[q]) * kPartStride;           // (item 0 exists; read and dropped where not wanted)
        sacc[q] = part[base + kk[q]];
        hpp[q] = part[base + (uu[q] >= 0 ? uu[q] : 0)];
    }
    if (I == K) {
        // (eight loads in flight per thread: one at a time the pass took a memory round trip per 256 doubles)
        const int ntot = nstage * kPartStride;
        for (int base = 0; base < ntot; base += 8 * kPT) {
            double tmp[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int e = base + tid + u * kPT; tmp[u] = part[(size_t)item0 * kPartStride + (e < ntot ? e : 0)]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int e = base + tid + u * kPT; if (e < ntot) sm[e] = tmp[u]; }
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
        const bool first = !staged[q] && i1[q] > i0[q];
        sacc[q] = first ? 0.0 + sacc[q] : 0.0;                     // (0 + v: the sum as k_dense_assemble forms it, to the sign of a zero)
        hpp[q] = (first && uu[q] >= 0) ? 0.0 + hpp[q] : 0.0;
    }
    if (av.st && tid == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); av.st[2] = __builtin_amdgcn_s_memrealtime(); }
    // sums in item order.  Staged elements: from LDS, the thread's elements step by step together (reads unconditional,
    // adds masked) ...
    if (nstage > 0) {
        int len = 0;
#pragma unroll
        for (int q = 0; q < kPer; ++q) len = max(len, staged[q] ? i1[q] - i0[q] : 0);
        for (int st = 0; st < len; ++st) {
            double va[kPer], vh[kPer];
#pragma unroll
            for (int q = 0; q < kPer; ++q) {
                const bool on = staged[q] && i0[q] + st < i1[q];
                const int it = on ? i0[q] + st - item0 : 0;
                va[q] = sm[it * kPartStride + kk[q]];
                vh[q] = sm[it * kPartStride + (uu[q] >= 0 ? uu[q] : 0)];
            }
#pragma unroll
            for (int q = 0; q < kPer; ++q)
                if (staged[q] && i0[q] + st < i1[q]) { sacc[q] += va[q]; hpp[q] += vh[q]; }
        }
    }
    // ... the further items of an element summed from memory (rare: an off-diagonal pair of more than 2 048 shared points, or
    // a diagonal tile too large to stage), step by step as well
    {
        int len = 0;
#pragma unroll
        for (int q = 0; q < kPer; ++q) len = max(len, staged[q] ? 0 : i1[q] - i0[q]);
        for (int st = 1; st < len; ++st) {
            double va[kPer], vh[kPer];
#pragma unroll
            for (int q = 0; q < kPer; ++q) {
                const bool on = !staged[q] && i0[q] + st < i1[q];
                const size_t base = (size_t)(on ? i0[q] + st : 0) * kPartStride;
                va[q] = part[base + kk[q]];
                vh[q] = part[base + (uu[q] >= 0 ? uu[q] : 0)];
            }
#pragma unroll
            for (int q = 0; q < kPer; ++q)
                if (!staged[q] && i0[q] + st < i1[q]) { sacc[q] += va[q]; if (uu[q] >= 0) hpp[q] += vh[q]; }
        }
    }
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
        double v = pad[q];
        if (live[q] && i1[q] > i0[q]) v = uu[q] >= 0 ? (hpp[q] + pad[q]) - sacc[q] : -sacc[q];
        const int e = tid + kPT * q, r = e / NB, cc = e - r * NB;
        dst[r * LD + cc] = v;
    }
    if (av.st && tid == 0) av.st[3] = __builtin_amdgcn_s_memrealtime();
    if (I == K && tid < NB) {
        // right-hand side b_S = b_p - sum B Dinv b_l of this block column; b_p is kept for computeScale
        const int gc = K * NB + tid;
        double v = 0.0;
        if (gc < n) {
            const int bj = gc / 6, a = gc - bj * 6;
            double bb = 0.0, cb = 0.0;
            const i32x2 jr = prange[(size_t)bj * nf + bj];
            const int j0 = jr.x, j1 = jr.y;
            if (nstage > 0) {
                for (int itx = j0; itx < j1; ++itx) { bb += sm[(itx - item0) * kPartStride + 63 + a]; cb += sm[(itx - item0) * kPartStride + 36 + a]; }
            } else {
                for (int itx = j0; itx < j1; ++itx) { bb += part[(size_t)itx * kPartStride + 63 + a]; cb += part[(size_t)itx * kPartStride + 36 + a]; }
            }
            st_sc1(av.bp + gc, bb);
            v = bb - cb;
        }
        sm[rrow_off + tid] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// D (48 x 48, LDS image at d_off, overwritten) -> W = L^-T (upper triangular, LDS image at w_off; its blocks below the
// diagonal are NOT written) with D = L L^T: the stacked sweep [D; I] -> [L; L^-T] in three panels of 16 columns.
//   chain   the panel's 16 pivots with one ROW PER LANE and the panel's 16 entries of the row in registers; pivot row k is
//           lane k, broadcast entry by entry with v_readlane (no LDS round trip, no barrier inside the chain):
//           p_rc -= (p_rk / p_kk) p_kc for c > k.  Wave 0 holds the rows of D from the panel's diagonal block down; wave 1
//           holds that diagonal block ONCE MORE on its lanes 0..15 (the same arithmetic, bit for bit) and behind it the rows
//           of the identity's image: row blocks 0 .. panel, the last of them still the identity itself (made in registers:
//           nothing below the diagonal of [I] is ever touched), so neither wave needs anything from the other in a panel;
//   trail   one round of three 16 x 16 blocks on the fp64 matrix cores, column block by column block (left-looking: block
//           column 2 takes both panels' updates at once): D(1,1), D(2,1), W(0,1) after panel 0; D(2,2), W(0,2), W(1,2)
//           after panel 1.
// L itself is not kept (only the rows below a panel, which the trailing products read): the substitutions and the tiles
// below use W.  The upper triangle of a diagonal block of D is carried as its symmetric image.
// Returns true (to every thread) when a pivot was not positive: the caller marks the factorisation failed.
// ---------------------------------------------------------------------------------------------------------------------
// One pivot of the chain.  What limits a panel is the dependent sequence
//     entry (K, K) final -> v_readlane (pivot) -> v_rcp_f64 -> e -> t -> td = p_K / pivot -> p_(K+1) -= td s_(K+1) -> next pivot
// so everything else is issued INTO that sequence's latency, in an order pinned by scheduling barriers (left to itself the
// scheduler put the thirty broadcasts and fifteen updates of a pivot in front of the reciprocal's refinement):
//   * pivot K's remaining updates (columns K + 2 ...) are deferred into pivot K + 1's reciprocal chain (tdp, sp = the previous
//     pivot's multiplier and row), two between each pair of dependent operations;
//   * p_K / pivot by ONE cubic step from the hardware reciprocal (relative error 2^-24 -> 2^-72): three dependent operations;
//   * the pivots are collected on their own lanes (mine): 1 / sqrt(pivot) is computed ONCE behind the chain, lane-parallel.
// A pivot that is not positive and finite is only recorded: the numbers that follow are garbage, the caller marks the
// factorisation failed and nothing of it is used.
#ifdef MOVBA_CHAIN_PINNED
#define CHAIN_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define CHAIN_FENCE() do { } while (0)
#endif
// lane LANE of old <- the wave-uniform v (two v_writelane_b32 from scalar registers)
template <int LANE>
__device__ __forceinline__ double writelane_f64(double v, double old)
{
    int lo = __double2loint(old), hi = __double2hiint(old);
    asm("v_writelane_b32 %0, %1, %2" : "+v"(lo) : "s"(__double2loint(v)), "n"(LANE));
    asm("v_writelane_b32 %0, %1, %2" : "+v"(hi) : "s"(__double2hiint(v)), "n"(LANE));
    return __hiloint2double(hi, lo);
}
template <int K>
struct PanelStep {
    static __device__ __forceinline__ void run(double (&p)[16], double &mine, bool &bad, int lane, double piv, double r0, double tdp, const double (&sp)[16])
    {
        const double e = __builtin_fma(-piv, r0, 1.0);
        const double q0 = p[K] * r0;
        CHAIN_FENCE();
        if constexpr (K >= 1 && K + 1 < 16) p[K + 1] -= tdp * sp[K + 1];
        if constexpr (K >= 1 && K + 2 < 16) p[K + 2] -= tdp * sp[K + 2];
        CHAIN_FENCE();
        const double t = __builtin_fma(e, e, e);
        CHAIN_FENCE();
        if constexpr (K >= 1 && K + 3 < 16) p[K + 3] -= tdp * sp[K + 3];
        if constexpr (K >= 1 && K + 4 < 16) p[K + 4] -= tdp * sp[K + 4];
        CHAIN_FENCE();
        const double td = __builtin_fma(q0, t, q0);
        CHAIN_FENCE();
        if constexpr (K >= 1) {
#pragma unroll
            for (int c = K + 5; c < 16; ++c) p[c] -= tdp * sp[c];
        }
        mine = writelane_f64<K>(piv, mine);           // (two v_writelane from the pivot's scalar registers)
        double s[16];
#pragma unroll
        for (int c = K + 1; c < 16; ++c) s[c] = readlane_f64(p[c], K);
        CHAIN_FENCE();
        if constexpr (K < 15) {
            p[K + 1] -= td * s[K + 1];
            CHAIN_FENCE();
            const double pivn = readlane_f64(p[K + 1], K + 1);
            const double r0n = __builtin_amdgcn_rcp(pivn);
            CHAIN_FENCE();
            PanelStep<K + 1>::run(p, mine, bad, lane, pivn, r0n, td, s);
        }
    }
};

__device__ __attribute__((noinline)) bool sweep_inverse(int d_off, int w_off, int flag_off)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];     // (LDS addressed from its own symbol: ds_ instructions, not flat_)
    double *D = sm + d_off, *W = sm + w_off;
    int *s_bad = reinterpret_cast<int *>(sm + flag_off);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) *s_bad = 0;
#ifdef MOVBA_SWEEP_PROFILE
    unsigned long long *prof = reinterpret_cast<unsigned long long *>(sm + flag_off - 660 + 256);      // (Lds::ps: idle during a sweep)
#define SWEEP_MARK(i) do { if (tid == 64) prof[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SWEEP_MARK(i) do { } while (0)
#endif
    SWEEP_MARK(0);
    bool bad_any = false;
#pragma unroll 1
    for (int pn = 0; pn < 3; ++pn) {
        const int c0 = 16 * pn;
        double pr[16], pk[16];
        double *rowp = D + c0 * LD + c0;                // (lanes without a row of their own shadow the pivot block's first row)
        bool writes = false;
        int ident_c = -1;
        const bool chain = wv < 2;
        if (wv == 0) {
            const int nr = NB - c0;
            if (lane < nr) rowp = D + (c0 + lane) * LD + c0;
            writes = lane >= 16 && lane < nr;           // (the rows below the pivot block: L21, read by the trailing products)
        } else if (wv == 1) {
            const int r = lane - 16;
            if (lane < 16) rowp = D + (c0 + lane) * LD + c0;
            else if (r < c0 + 16) { rowp = W + r * LD + c0; writes = true; if (r >= c0) ident_c = r - c0; }
        }
        if (chain) {
#pragma unroll
            for (int c = 0; c < 16; ++c) pr[c] = rowp[c];               // (all sixteen reads in flight; an identity row reads and drops)
            if (__any(ident_c >= 0)) {
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    // (bitwise selects: the value read may be anything)
                    const long long keep = ident_c >= 0 ? 0ll : -1ll, one = c == ident_c ? 0x3ff0000000000000ll : 0ll;
                    pr[c] = __longlong_as_double((__double_as_longlong(pr[c]) & keep) | one);
                }
            }
            SWEEP_MARK(1 + 4 * pn);
            {
                const double piv0 = readlane_f64(pr[0], 0);
                const double s0[16] = {};
                double mine = 1.0;
                bool bad = false;
                PanelStep<0>::run(pr, mine, bad, lane, piv0, __builtin_amdgcn_rcp(piv0), 0.0, s0);
                bad_any |= __any(!(mine > 0.0 && mine < __builtin_inf()));
                // 1 / sqrt(pivot) by v_rsq_f64 and two Newton steps, pivot c on lane c; broadcast for the scaling below
                double rs = __builtin_amdgcn_rsq(mine);
                const double hp = 0.5 * mine;
                rs = rs * __builtin_fma(-hp * rs, rs, 1.5);
                rs = rs * __builtin_fma(-hp * rs, rs, 1.5);
#pragma unroll
                for (int c = 0; c < 16; ++c) pk[c] = readlane_f64(rs, c);
            }
            SWEEP_MARK(2 + 4 * pn);
            if (writes) {
#pragma unroll
                for (int c = 0; c < 16; ++c) rowp[c] = pr[c] * pk[c];
            }
        }
        __syncthreads();
        SWEEP_MARK(3 + 4 * pn);
        if (pn == 2) break;
        if (wv < 3) {
            // pn == 0: D(1,1), D(2,1), W(0,1) over the panel's columns;  pn == 1: D(2,2), W(0,2) over both panels', W(1,2) over this one's
            const int cb = pn + 1;
            const bool isW = pn == 0 ? wv == 2 : wv >= 1;
            const int rb = pn == 0 ? (wv == 0 ? 1 : (wv == 1 ? 2 : 0)) : (wv == 0 ? 2 : wv - 1);
            const int k0 = (pn == 1 && wv < 2) ? 0 : c0, k1 = c0 + 16;
            const double *Ab = (isW ? W : D) + (16 * rb) * LD;
            double *Cb = (isW ? W : D) + (16 * rb) * LD + 16 * cb;
            const double *ap = Ab + (lane & 15) * LD + (lane >> 4);
            const double *bp = D + (16 * cb + (lane & 15)) * LD + (lane >> 4);
            double *cp = Cb + (lane >> 4) * LD + (lane & 15);
            dbl4 acc = isW ? dbl4{ 0.0, 0.0, 0.0, 0.0 } : dbl4{ cp[0], cp[4 * LD], cp[8 * LD], cp[12 * LD] };
            double av[8], bv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { const int k = k0 + 4 * q; av[q] = k < k1 ? -ap[k] : 0.0; bv[q] = k < k1 ? bp[k] : 0.0; }
#pragma unroll
            for (int q = 0; q < 8; ++q) if (k0 + 4 * q < k1) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], acc, 0, 0, 0);
            cp[0] = acc.x; cp[4 * LD] = acc.y; cp[8 * LD] = acc.z; cp[12 * LD] = acc.w;
        }
        __syncthreads();
        SWEEP_MARK(4 + 4 * pn);
    }
    if (bad_any && lane == 0) *s_bad = 1;
    __syncthreads();
    return *s_bad != 0;
}


}  // namespace

__global__ __launch_bounds__(kPT) void k_dense_persist(DevWindow w, unsigned epoch)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    Ctrl *c = w.ctrl;
    if (c->done == 1) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // everything the task loop needs from the argument block, once, in registers
    const int nt = w.dense.ntile, n = w.dense.n, nslots = w.dense.slots;
    double *const tiles = w.dense.tiles, *const contrib = w.dense.contrib, *const xsol = w.dense.xsol;
    unsigned *const flags = w.dense.flags, *const failw = w.dense.failw, *const ctag = w.dense.ctag;
    unsigned long long *const stamps = w.dense.stamps;
    const DenseTask *const gtasks = w.dense.tasks;
    AsmView av = { w.part, w.dense.prange, w.bp, w.nfree, w.dense.n, nullptr };
    const Lds l = carve(sm, nslots, w.wait_ticks);
    const double lambda = c->lambda;
    if (tid == 0) { *l.abort = 0; *l.prog = 0; }
    __syncthreads();
    bool aborted = false;
    // a tile's publication whose flag is still to be set: the stores are issued, the drain + flag follow behind the next piece of
    // work that needs nothing from memory (set before any wait, any further publication, and at the end)
    int pending = -1;
    auto flush = [&]() { if (pending >= 0) { set_flag(flags, pending, epoch, tid); pending = -1; } };

    const int t0 = w.dense.task_ptr[blockIdx.x], t1 = w.dense.task_ptr[blockIdx.x + 1];
    int32_t *tcache = reinterpret_cast<int32_t *>(sm + kTaskOff(nslots));       // the next kTaskCache tasks of this workgroup, in LDS
    for (int t = t0; t < t1 && !aborted; ++t) {
        if ((t - t0) % kTaskCache == 0) {
            __syncthreads();
            const int nload = min(kTaskCache, t1 - t) * 8;
            for (int q = tid; q < nload; q += kPT) tcache[q] = reinterpret_cast<const int32_t *>(gtasks + t)[q];
            __syncthreads();
        }
        const int32_t *tw = tcache + ((t - t0) % kTaskCache) * 8;
        DenseTask tk;
        tk.op = __builtin_amdgcn_readfirstlane(tw[0]); tk.slot = __builtin_amdgcn_readfirstlane(tw[1]); tk.I = __builtin_amdgcn_readfirstlane(tw[2]);
        tk.K = __builtin_amdgcn_readfirstlane(tw[3]); tk.k = __builtin_amdgcn_readfirstlane(tw[4]);
        tk.pad[0] = __builtin_amdgcn_readfirstlane(tw[5]); tk.pad[1] = __builtin_amdgcn_readfirstlane(tw[6]); tk.pad[2] = __builtin_amdgcn_readfirstlane(tw[7]);
        const int I = tk.I, K = tk.K, k = tk.k;
        if (stamps && tid == 0) { stamps[6 * (size_t)t] = __builtin_amdgcn_s_memrealtime(); for (int q = 1; q < 5; ++q) stamps[6 * (size_t)t + q] = 0; }
        const int slot_off = (2 + tk.slot) * kTileLds;
        double *slot = sm + slot_off;
        switch (tk.op) {
        case DT_ASM: {
            av.st = stamps ? stamps + 6 * (size_t)t : nullptr;
            assemble_tile(av, I, K, lambda, slot_off, (2 + nslots) * kTileLds, tid);      // (the right-hand side row into Lds::rrow)
            __syncthreads();
            break;
        }
        case DT_UPD: {
            // operands this workgroup produced itself are read where they lie (its own LDS slots: pad[0] / pad[1] of the task);
            // the others are waited for and fetched
            const int sa = tk.pad[0], sb = I == K ? tk.pad[0] : tk.pad[1];
            const bool getA = sa == -1, getB = I != K && sb == -1;          // (-2: in the scratch tile since the update before)
            const int fa = dense_flag_F(nt, I, k), fb = dense_flag_F(nt, K, k);
            if (getA || getB) flush();
            if (getA && getB) {
                const int m = wg_wait2(flags, epoch, fa, fb, l, tid);
                if (m == 0) { aborted = true; break; }
                if (stamps && tid == 0) stamps[6 * (size_t)t + 1] = __builtin_amdgcn_s_memrealtime();
                fetch_tile(tiles + tile_off(I, k), l.A, tid);
                if (m != 3 && !wg_wait(flags, epoch, 1, [&](int) { return fb; }, l, tid)) { aborted = true; break; }
                fetch_tile(tiles + tile_off(K, k), l.B, tid);
            } else if (getA || getB) {
                if (!wg_wait(flags, epoch, 1, [&](int) { return getA ? fa : fb; }, l, tid)) { aborted = true; break; }
                if (stamps && tid == 0) stamps[6 * (size_t)t + 1] = __builtin_amdgcn_s_memrealtime();
                if (getA) fetch_tile(tiles + tile_off(I, k), l.A, tid); else fetch_tile(tiles + tile_off(K, k), l.B, tid);
            }
            if (getA || getB) __syncthreads();
            const double *Ap = sa < 0 ? l.A : sm + (2 + sa) * kTileLds;
            const double *Bp = I == K ? Ap : (sb < 0 ? l.B : sm + (2 + sb) * kTileLds);
            if (I == K) tile_update<true>(slot, Ap, Bp, wv, lane); else tile_update<false>(slot, Ap, Bp, wv, lane);
            __syncthreads();
            break;
        }
        case DT_UPD2: {
            // block column k into the diagonal tile (six blocks on and below the diagonal) and the tile to its left (nine)
            const int sa = tk.pad[0], sb = tk.pad[1];
            const int fa = dense_flag_F(nt, K, k), fb = dense_flag_F(nt, K - 1, k);
            if (sa < 0 || sb < 0) flush();
            if (sa < 0 && sb < 0) {
                if (!wg_wait(flags, epoch, 2, [&](int i) { return i == 0 ? fa : fb; }, l, tid)) { aborted = true; break; }
            } else if (sa < 0 || sb < 0) {
                if (!wg_wait(flags, epoch, 1, [&](int) { return sa < 0 ? fa : fb; }, l, tid)) { aborted = true; break; }
            }
            if (stamps && tid == 0) stamps[6 * (size_t)t + 1] = __builtin_amdgcn_s_memrealtime();
            if (sa < 0) fetch_tile(tiles + tile_off(K, k), l.A, tid);
            if (sb < 0) fetch_tile(tiles + tile_off(K - 1, k), l.B, tid);
            if (sa < 0 || sb < 0) __syncthreads();
            const double *Ap = sa < 0 ? l.A : sm + (2 + sa) * kTileLds;
            const double *Bp = sb < 0 ? l.B : sm + (2 + sb) * kTileLds;
            double *sub = sm + (2 + tk.pad[2]) * kTileLds;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = wv + 4 * u;
                if (q < 15) {
                    const bool dg = q < 6;
                    const int qq = dg ? q : q - 6;
                    const int mt = dg ? (qq >= 3 ? 2 : (qq >= 1 ? 1 : 0)) : qq / 3, ntc = dg ? qq - mt * (mt + 1) / 2 : qq - mt * 3;
                    double *cp = (dg ? slot : sub) + (mt * 16 + (lane >> 4)) * LD + ntc * 16 + (lane & 15);
                    dbl4 acc = dbl4{ cp[0], cp[4 * LD], cp[8 * LD], cp[12 * LD] };
                    acc = tile_mfma(Ap, dg ? Ap : Bp, mt, ntc, lane, acc);
                    cp[0] = acc.x; cp[4 * LD] = acc.y; cp[8 * LD] = acc.z; cp[12 * LD] = acc.w;
                }
            }
            __syncthreads();
            break;
        }
        case DT_DIAG: {
            // D_K -> W_K = L(K, K)^-T: the stacked sweep of D_K over the identity (sweep_inverse).  W_K is what
            // travels - the tiles below take  L(I, K) = S(I, K) W_K  as one matrix product instead of factoring D_K once more
            // each - and what this workgroup keeps, in D_K's slot: both substitutions are products with it.
            flush();
            // (D_K lies in scratch tile B: its last update wrote it there - block column 0's is copied - so that W_K can be
            //  written straight into the slot)
            if (K == 0) {
                for (int e = tid; e < NB * NB; e += kPT) { const int r = e / NB, cc = e - r * NB; l.B[r * LD + cc] = slot[r * LD + cc]; }
                __syncthreads();
            }
            if (stamps && tid == 0) stamps[6 * (size_t)t + 2] = __builtin_amdgcn_s_memrealtime();
            if (sweep_inverse(kTileLds, slot_off, kBadOff(nslots)) && tid == 0) st_flag(failw, epoch);
            if (stamps && tid == 0) stamps[6 * (size_t)t + 3] = __builtin_amdgcn_s_memrealtime();
            publish_tile<2>(tiles + tile_off(K, K), slot, tid);
            set_flag(flags, dense_flag_PD(nt, K), epoch, tid);
            // (behind the flag: what is left of D_K below the diagonal blocks goes, the substitutions multiply with the whole slot)
            for (int e = tid; e < NB * NB; e += kPT) { const int r = e / NB, cc = e - r * NB; if (tri_skip<2>(r, cc)) slot[r * LD + cc] = 0.0; }
            __syncthreads();
            break;
        }
        case DT_COL: {
            // the diagonal owner's chain of one block column: L(K, K-1) = S(K, K-1) W_(K-1), D_K -= L L^T, D_K -> W_K
            flush();
            if (!wg_wait(flags, epoch, 1, [&](int) { return dense_flag_PD(nt, K - 1); }, l, tid)) { aborted = true; break; }
            if (stamps && tid == 0) stamps[6 * (size_t)t + 1] = __builtin_amdgcn_s_memrealtime();
            fetch_tile<2>(tiles + tile_off(K - 1, K - 1), l.B, tid);      // W_(K-1)
            __syncthreads();
            double *sub = sm + (2 + tk.pad[2]) * kTileLds;
            tile_times_upper(sub, l.B, wv, lane);
            __syncthreads();
            publish_tile(tiles + tile_off(K, K - 1), sub, tid);         // (stores issued; drained behind the update below)
            tile_update<true>(slot, sub, sub, wv, lane, l.B);           // D_K into scratch tile B (W_(K-1) is through)
            set_flag(flags, dense_flag_F(nt, K, K - 1), epoch, tid);
            if (stamps && tid == 0) stamps[6 * (size_t)t + 2] = __builtin_amdgcn_s_memrealtime();
            if (sweep_inverse(kTileLds, slot_off, kBadOff(nslots)) && tid == 0) st_flag(failw, epoch);
            if (stamps && tid == 0) stamps[6 * (size_t)t + 3] = __builtin_amdgcn_s_memrealtime();
#ifdef MOVBA_SWEEP_PROFILE
            if (stamps && tid == 0 && K == 1) {
                const unsigned long long *pf = reinterpret_cast<const unsigned long long *>(l.ps);
                printf("sweep D_1: p0 load %llu chain %llu store+sync %llu trail %llu | p1 load %llu chain %llu store+sync %llu trail %llu | p2 load %llu chain %llu store+sync %llu  (x10 ns)\n",
                       pf[1] - pf[0], pf[2] - pf[1], pf[3] - pf[2], pf[4] - pf[3], pf[5] - pf[4], pf[6] - pf[5], pf[7] - pf[6], pf[8] - pf[7], pf[9] - pf[8], pf[10] - pf[9], pf[11] - pf[10]);
            }
#endif
            publish_tile<2>(tiles + tile_off(K, K), slot, tid);
            set_flag(flags, dense_flag_PD(nt, K), epoch, tid);
            for (int e = tid; e < NB * NB; e += kPT) { const int r = e / NB, cc = e - r * NB; if (tri_skip<2>(r, cc)) slot[r * LD + cc] = 0.0; }
            __syncthreads();
            break;
        }
        case DT_OFF: {
            flush();
            if (!wg_wait(flags, epoch, 1, [&](int) { return dense_flag_PD(nt, K); }, l, tid)) { aborted = true; break; }
            if (stamps && tid == 0) stamps[6 * (size_t)t + 1] = __builtin_amdgcn_s_memrealtime();
            fetch_tile<2>(tiles + tile_off(K, K), l.B, tid);      // W_K
            __syncthreads();
            if (stamps && tid == 0) stamps[6 * (size_t)t + 2] = __builtin_amdgcn_s_memrealtime();
            tile_times_upper(slot, l.B, wv, lane);
            __syncthreads();
            if (stamps && tid == 0) stamps[6 * (size_t)t + 3] = __builtin_amdgcn_s_memrealtime();
            // Its stores are issued here; the drain and the flag follow behind the next piece of work (for the diagonal owner: the
            // update of D_I from its own LDS slots), so the write acknowledgements are not waited for on the chain.
            publish_tile(tiles + tile_off(I, K), slot, tid);
            pending = dense_flag_F(nt, I, K);
            break;
        }
        case DT_RHS: {
            // r_K -= L(K, K-1) y_(K-1) (the block columns up to K - 2 were applied by the DT_RUP tasks), then y_K = W_K^T r_K
            flush();
            if (K >= 1) {
                if (!wg_wait(flags, epoch, 1, [&](int) { return dense_flag_F(nt, nt, K - 1); }, l, tid)) { aborted = true; break; }
                if (stamps && tid == 0) stamps[6 * (size_t)t + 1] = __builtin_amdgcn_s_memrealtime();
                if (tid < NB) l.xs[tid] = ld_sc1(tiles + tile_off(nt, K - 1) + tid);
                __syncthreads();
                rhs_minus_tile_times(sm + (2 + tk.pad[0]) * kTileLds, l, tid);       // L(K, K-1): the workgroup's own tile (pad[0] = its slot)
            }
            if (tid < NB) l.xs[tid] = l.rrow[tid];
            __syncthreads();
            const double yk = tile_matvec<true>(slot, l, tid);
            if (tid < NB) { l.yv[tid] = yk; st_sc1(tiles + tile_off(nt, K) + tid, yk); }
            set_flag(flags, dense_flag_F(nt, nt, K), epoch, tid);
            break;
        }
        case DT_RUP: {
            flush();
            if (!wg_wait(flags, epoch, 1, [&](int) { return dense_flag_F(nt, nt, k); }, l, tid)) { aborted = true; break; }
            if (stamps && tid == 0) stamps[6 * (size_t)t + 1] = __builtin_amdgcn_s_memrealtime();
            fetch_tile(tiles + tile_off(K, k), l.A, tid);
            if (tid < NB) l.xs[tid] = ld_sc1(tiles + tile_off(nt, k) + tid);
            __syncthreads();
            rhs_minus_tile_times(l.A, l, tid);
            break;
        }
        case DT_BSX: {
            // the contributions of the block rows from J + 2 down were ready long ago: summed first, so that the chain
            // x_(J+1) -> c(J+1, J) -> x_J carries one flag and one 48-vector
            const int J = K, nc = nt - 1 - J;
            flush();
            double acc = 0.0;
            if (nc > 1) {
                if (!wg_wait(flags, epoch, nc - 1, [&](int i) { return dense_flag_FC(nt, J + 2 + i, J); }, l, tid)) { aborted = true; break; }
                if (tid < NB)
                    for (int i0 = J + 2; i0 < nt; i0 += 8) {
                        double v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = i0 + u < nt ? ld_sc1(contrib + ((size_t)(i0 + u) * nt + J) * NB + tid) : 0.0;
#pragma unroll
                        for (int u = 0; u < 8; ++u) acc += v[u];
                    }
            }
            // (the one contribution on the chain, c(J+1, J), comes as tagged records: no flag, no separate load)
            double cj = 0.0;
            if (nc > 0 && !ld_tagged(ctag + (size_t)J * NB * 4, NB * 16, tid, epoch, cj, l)) { aborted = true; break; }
            if (stamps && tid == 0) stamps[6 * (size_t)t + 1] = __builtin_amdgcn_s_memrealtime();
            if (tid < NB) {
                if (nc > 0) acc += cj;
                l.xs[tid] = l.yv[tid] - acc;
            }
            __syncthreads();
            const double xj = tile_matvec<false>(slot, l, tid);       // x_J = W_J (y_J - sum c(I, J))
            __syncthreads();                                                   // (every thread is through with xs and ps)
            if (tid < NB) { l.xv[tid] = xj; l.xs[tid] = xj; st_sc1(xsol + J * NB + tid, xj); }
            if (J >= 1) {
                // c(J, J-1) = L(J, J-1)^T x_J straight away (the owner's own tile), handed to block row J - 1 as tagged records
                __syncthreads();
                const double cv = tile_matvec<true>(sm + (2 + tk.pad[0]) * kTileLds, l, tid);
                if (tid < NB) st_tagged(ctag + (size_t)(J - 1) * NB * 4, NB * 16, tid, cv, epoch);
            }
            // x_J's flag, for the other tiles of block row J: behind the tagged records (not on the chain)
            set_flag(flags, dense_flag_FX(nt, J), epoch, tid);
            break;
        }
        case DT_BSC: {
            // c(I, J) = L(I, J)^T x_I with J = K: column sums over the tile's rows in five row groups, combined in fixed order
            // (x_I of the workgroup's own diagonal tile is still in LDS: pad[0] of the task)
            if (tk.pad[0] != 1) {
                flush();
                if (!wg_wait(flags, epoch, 1, [&](int) { return dense_flag_FX(nt, I); }, l, tid)) { aborted = true; break; }
                if (stamps && tid == 0) stamps[6 * (size_t)t + 1] = __builtin_amdgcn_s_memrealtime();
                if (tid < NB) l.xs[tid] = ld_sc1(xsol + I * NB + tid);
            } else if (tid < NB) l.xs[tid] = l.xv[tid];
            __syncthreads();
            const int g = tid / NB, cc = tid - g * NB;
            if (g < 5) {
                double s = 0.0;
#pragma unroll
                for (int q = 0; q < 10; ++q) { const int r = g + 5 * q; if (r < NB) s += slot[r * LD + cc] * l.xs[r]; }
                l.ps[g * NB + cc] = s;
            }
            __syncthreads();
            const double cv = tid < NB ? (((l.ps[tid] + l.ps[NB + tid]) + l.ps[2 * NB + tid]) + l.ps[3 * NB + tid]) + l.ps[4 * NB + tid] : 0.0;
            if (tid < NB) st_sc1(contrib + ((size_t)I * nt + K) * NB + tid, cv);
            set_flag(flags, dense_flag_FC(nt, I, K), epoch, tid);
            if (pending >= 0) { if (tid == 0) st_flag(flags + pending, epoch); pending = -1; }      // (x_I's stores were drained by the same wait)
            break;
        }
        case DT_EPI: {
            flush();
            if (!wg_wait(flags, epoch, nt, [&](int i) { return dense_flag_FX(nt, i); }, l, tid)) { aborted = true; break; }
            if (stamps && tid == 0) stamps[6 * (size_t)t + 1] = __builtin_amdgcn_s_memrealtime();
            break;      // (the outputs follow the loop)
        }
        default: break;
        }
        if (stamps && tid == 0) stamps[6 * (size_t)t + 5] = __builtin_amdgcn_s_memrealtime();
    }
    flush();
    if (blockIdx.x != 0) {
        if (aborted && tid == 0) st_flag(failw + 1, epoch);
        return;
    }

    // ---- workgroup 0: increment, pose part of computeScale(), trial poses (VertexSE3Expmap::oplusImpl); releases a parked
    // solve (Ctrl::done 2 -> 0).  A solve that gave up on a wait is reported as a failed factorisation AND counted. ----
    __syncthreads();
    const bool fail = aborted || ld_flag(failw) == epoch || ld_flag(failw + 1) == epoch;
    double *x = l.A;                                // n <= 2 tiles of scratch (dense_persist_supported)
    double sc = 0.0;
    for (int idx = tid; idx < n; idx += kPT) {
        const double xv = fail ? 0.0 : ld_sc1(xsol + idx);
        const double bpv = fail ? 0.0 : ld_sc1(w.bp + idx);
        w.xp[idx] = xv;
        sc += xv * (lambda * xv + bpv);
        x[idx] = xv;
    }
    const double scs = block_reduce<kPT / 64, false>(sc, l.red);
    const int cur = c->cur;
    const DevState &S0 = w.st[cur];
    const DevState &S1 = w.st[cur ^ 1];
    for (int i = tid; i < w.NP; i += kPT) {
        double T[7], Tn[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) T[q] = S0.pose[7 * i + q];
        const int h = w.hidx[i];
        if (h >= 0 && !fail) {          // (a failed factorisation moves nothing: g2o returns from solve() before its update)
            double u[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) u[q] = x[6 * h + q];
            se3_oplus(u, T, Tn);
        } else {
#pragma unroll
            for (int q = 0; q < 7; ++q) Tn[q] = T[q];
        }
        double R[9];
        quat_to_R(Tn, R);
#pragma unroll
        for (int q = 0; q < 7; ++q) S1.pose[7 * i + q] = Tn[q];
#pragma unroll
        for (int q = 0; q < 9; ++q) S1.Rt[12 * i + q] = R[q];
        S1.Rt[12 * i + 9] = Tn[4]; S1.Rt[12 * i + 10] = Tn[5]; S1.Rt[12 * i + 11] = Tn[6];
    }
    if (tid == 0) {
        w.scale_part[w.n_pt_blocks] = scs;
        c->pcg_fail = fail ? 1 : 0;
        c->pcg_last_iters = -1;                     // trace marker: this trial was solved directly
        c->n_direct += 1;
        if (fail) c->n_chol_fail += 1;
        if (aborted || ld_flag(failw + 1) == epoch) c->n_sync_timeouts += 1;
        if (c->done == 2) c->done = 0;              // the solve was parked for this: the kernels behind run again
    }
}

size_t dense_persist_lds_bytes(int slots) { return ((size_t)(2 + slots) * kTileLds + 672 + kTaskCache * 4) * sizeof(double); }

bool dense_persist_supported(const DensePlan &p)
{
    return p.ok && p.nt * NB <= 2 * kTileLds && dense_persist_lds_bytes(p.slots) <= 160 * 1024 - 1024;
}

hipError_t launch_dense_persist(const DevWindow &w, unsigned epoch, hipStream_t s)
{
    hipLaunchKernelGGL(k_dense_persist, dim3(w.dense.G), dim3(kPT), dense_persist_lds_bytes(w.dense.slots), s, w, epoch);
    return hipGetLastError();
}

hipError_t configure_dense_persist()
{
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k_dense_persist), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
}

}  // namespace movba
