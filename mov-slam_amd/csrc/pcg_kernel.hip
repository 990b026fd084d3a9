// k_pcg_rows: the reduced-system solve of one LM trial in ONE workgroup, with the whole
// (block-sparse, symmetric) matrix S held on-chip for the duration of the solve.
//
// Replaces the LinearSolverCSparse Cholesky the reference selects at
// /root/reference/src/Optimizer.cc:535 by the damped block-Jacobi PCG BASELINE.json's
// north_star asks for.  S = Hpp + lambda I - sum_l B_il Dinv_l B_jl^T is at most a few
// hundred KB: too big for one lane, too small to be worth a grid (a grid barrier costs more
// than two whole CG iterations here), so the design goal is latency per CG iteration:
//   * a wave owns a run of block rows; the 6x6 blocks of those rows (both triangles, read
//     through the per-row gather lists) are dealt to the wave's lanes in PAIRS of the same row
//     and stay in VGPRs for the whole solve; they are assembled straight from the schur
//     work-item partials, S is never written to memory (only dense / very large windows
//     spill the tail of a list to an L2 copy);
//   * mat-vec = per lane y = B0 p_c0 + B1 p_c1 (p read once per block from LDS), the six sums
//     parked in a wave-private LDS strip and added up by the row's owner lane in list order:
//     no workgroup barrier inside the mat-vec;
//   * the 6 rows of a block live in one wave, so the block-Jacobi preconditioner needs only a
//     wave-local LDS exchange;
//   * preconditioner = block-Jacobi + an aggregate coarse level (two-level additive Schwarz): the block
//     rows of one wave form an aggregate with 12 coarse dofs (the 6 pose components constant over the aggregate
//     and varying linearly with the keyframe index); A_c^-1 (96 x 96, kept in LDS as fp32) comes from the
//     previous trial (built by the SECOND workgroup of that launch, coarse_level.h, beside the CG) and removes
//     the smooth drift / bending modes block-Jacobi cannot see: 869 -> 215 CG iterations per cfg3 window solve.
//     A window with at most one keyframe per wave builds the (then exact) inverse first and solves in ~3 iterations;
//   * single-reduction conjugate gradients (Chronopoulos & Gear): per iteration 2 workgroup barriers and ONE pair of
//     DPP wave reductions; each wave's coarse correction follows a recurrence, its dense product runs behind the
//     reduction barrier on the restricted mat-vec result that is published with the dot products.
// All reductions run in a fixed order: results are bit-reproducible run to run.
#include <hip/hip_runtime.h>

#include "coarse_level.h"
#include "device_math.h"
#include "device_types.h"
#include "handoff.h"
#include "kernels.h"

namespace movba {

#ifdef MOVBA_CLOCK_STAMP
#define SEG_STAMP(k) do { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); seg[k] += _t - seg_last; seg_last = _t; } while (0)
#else
#define SEG_STAMP(k) do { } while (0)
#endif
#ifdef MOVBA_CLOCK_STAMP
#define SETUP_STAMP(k) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); if (tid == 0) c->dbg_seg2[k] += _t - setup_last; setup_last = _t; } while (0)
#else
#define SETUP_STAMP(k) do { } while (0)
#endif

// wave-uniform scalars of the CG recurrences: moved to scalar registers (two VGPRs less per value) or left where they are
#ifdef MOVBA_PCG_NO_UNIFORM
#define PCG_UNI(x) (x)
#else
#define PCG_UNI(x) uniform_f64(x)
#endif

namespace {

constexpr int kT = kPcgRowsThreads;     // 512 = 8 waves, 2 per SIMD -> 256 VGPRs per lane
constexpr int kNW = kT / 64;
constexpr int kOwnBatch = 10;           // pair sums an owner lane loads per LDS round trip
constexpr int kPA = kCoarsePerAgg;       // coarse dofs per aggregate: 6 constant + 6 linear-in-keyframe-index modes
constexpr int kNC = kPA * kNW;          // coarse dofs: one aggregate per wave
static_assert(kNC == kCoarseDim, "coarse dimension");

// fixed-order sum of the kNW wave partials (a balanced tree: three dependent adds instead of seven)
__device__ __forceinline__ double sum_fixed(const double *red)
{
    static_assert(kNW == 8, "tree written for 8 waves");
    const double2 *r2 = reinterpret_cast<const double2 *>(red);
    const double2 a = r2[0], b = r2[1], c = r2[2], d = r2[3];
    return ((a.x + a.y) + (b.x + b.y)) + ((c.x + c.y) + (d.x + d.y));
}

}  // namespace

// OVERFLOW: some wave's gather list does not fit its 64 VGPR-resident pairs; the tail is multiplied from an L2 copy of S
// (a separate instantiation: its extra live values must not cost the common case registers)
// role 0: the solving workgroup, role 1: the coarse-level builder of the same launch
// PADDED (PcgParams::padded: no block row with more than kOwnBatch entry pairs, nothing overflows - cfg3 and everything smaller):
// the pair sums of the mat-vec are parked by ROW, each row with kOwnBatch slots of which the unused ones hold zeros for the
// whole solve, and the residual exchange of a wave goes through a strip of its own whose unowned places hold zeros: the owner
// sums and the restriction read their ten values without the forty v_cndmask per iteration that masked the over-read of the
// packed layout.  Same values added in the same order: the two layouts give the same bits (batched runs may mix them).
template <bool OVERFLOW, bool PADDED>
__device__ __forceinline__ void pcg_rows_body(const DevWindow &w, const PcgParams &pp, int trial, int role)
{
    static_assert(!(OVERFLOW && PADDED), "the padded layout holds the register-resident pairs only");
    extern __shared__ __attribute__((aligned(16))) double sm[];
    Ctrl *c = w.ctrl;
    const int tid = threadIdx.x, ln = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: row ranges and their predicates live in SGPRs
    // use_coarse == 2 (windows with at most one keyframe per wave: the coarse space is the whole space): the coarse level
    // is built FIRST, by this workgroup, from THIS trial's matrix, and the solve below converges in a couple of iterations
    const bool fresh = pp.use_coarse == 2;
    // ---- the solver proper: everything whose address follows from the kernel's arguments alone is REQUESTED first - its plans,
    // the rows of the gather lists, the previous trial's coarse inverse, the blocks the schur pass left for
    // this thread - before anything is waited for: the setup used to be a chain of five dependent round trips behind the
    // launch boundary (Ctrl -> plans -> item ranges -> partials), ~1 us each from a cold L2 ----
    const bool solver = role == 0 && !fresh;
    const int nf = w.nfree, n = 6 * nf;
    const int b0 = pp.wave_row0[wv], b1 = pp.wave_row0[wv + 1];      // wave wv owns block rows [b0, b1); lane ln < 6 (b1 - b0) owns scalar row 6 b0 + ln
    const int row = b0 * 6 + ln;
    const bool owner = ln < 6 * (b1 - b0);
    const int bi = owner ? row / 6 : 0, ba = owner ? row - bi * 6 : 0;
    const int ctrial = fresh ? trial : trial - 1;         // the trial whose coarse matrix preconditions this solve
    int cur = 0;
    double lambda = 0.0;
#ifdef MOVBA_CLOCK_STAMP
    const unsigned long long stamp_c0 = __builtin_amdgcn_s_memtime(), stamp_t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long setup_last = stamp_c0;
#endif
    const double *part = w.part;
    if (!solver) {
        // the coarse-level workgroup, and a solver that builds its coarse level first, read every partial of the schur pass
        if (c->done) return;
        cur = c->cur; lambda = c->lambda;
        if (role == 1) {            // second workgroup: coarse level of THIS trial's matrix, for the next trial
            coarse_build<kT, kNC, kPA>(w, pp, trial, lambda, sm, true);
            return;
        }
        coarse_build<kT, kNC, kPA>(w, pp, trial, lambda, sm, false);
        __syncthreads();
    }
    // ---- the requests, in the order their answers are needed (vector loads return in order).  One plan load per thread says
    // everything it needs of the window's structure (api.cpp, lay_out_rest); the rest follows from the kernel's arguments. ----
    int done_w = 0;
    if (solver) { done_w = c->done; cur = c->cur; lambda = c->lambda; }
    // The previous trial's coarse inverse (36 KB, fp32) goes from global memory STRAIGHT into its place in LDS
    // (global_load_lds_dwordx4: no register in between - the kernel is at its 256 - and no second pass), requested here with
    // everything else; whether it is valid (its tag) is looked at further down, the first read of it lies behind workgroup barriers.
    // (It used to be requested ~3 us into the kernel, behind the diagonal sums, through 18 registers, and stored to LDS after the
    //  block inverses.)
    int aci_tag = -1;
    if (pp.use_coarse) {
        typedef __attribute__((address_space(1))) const void *gptr_t;
        typedef __attribute__((address_space(3))) void *lptr_t;
        const int npad0 = (6 * w.nfree + 1) & ~1;
        float *acf0 = reinterpret_cast<float *>(sm + 2 * npad0 + 36 * w.nfree + 2 * kNW + 2);      // (= Acf of the carve below)
        aci_tag = w.aci_tag[max(ctrial, 0) & 1];
        const float *src = w.aci + (size_t)(max(ctrial, 0) & 1) * kNC * kNC;
        static_assert((kNC * kNC) % 256 == 0, "whole waves of 16-byte pieces");
#pragma unroll
        for (int u = 0; u * kT * 4 < kNC * kNC; ++u) {
            const int piece = tid + u * kT;                 // 16-byte piece; a wave's 64 pieces land behind one another
            if ((piece & ~63) * 4 < kNC * kNC)
                __builtin_amdgcn_global_load_lds((gptr_t)(src + 4 * piece), (lptr_t)(acf0 + 4 * (piece & ~63)), 16, 0, 0);
        }
    }
    const int4 *plan = reinterpret_cast<const int4 *>(w.lane_plan) + (size_t)tid * 3;
    const int4 pl0 = plan[0], pl1 = plan[1], plo = plan[2];
    const int oi0 = plo.x, oi1 = plo.y;                   // owner lanes: items of the diagonal pair (bi, bi): they carry b_p and B Dinv b_l
    const int rp_b0 = pp.wave_ent0[wv], rp_b1 = pp.wave_ent0[wv + 1], rp_nf = pp.nrowent, rp_bi = plo.z, rp_bi1 = plo.w;
    // the keyframe's first four diagonal records (DevWindow::rec_d: place s of keyframe bi at (bi rec_slots + s) 48; row ba = 8 doubles)
    constexpr int kRecFly = 4;
    double2 rv[kRecFly][4];
    const double2 *rec0 = reinterpret_cast<const double2 *>(w.rec_d + (size_t)bi * w.rec_slots * 48 + ba * 8);
#pragma unroll
    for (int u = 0; u < kRecFly; ++u) {
        const double2 *src = rec0 + (size_t)min(u, w.rec_slots - 1) * 24;
#pragma unroll
        for (int q = 0; q < 4; ++q) rv[u][q] = src[q];
    }
    double Bo[2][36];
    if (!OVERFLOW) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int q = 0; q < 36; ++q) Bo[k][q] = w.img_b[(size_t)(36 * k + q) * kT + tid];
    }
    if (solver && done_w) return;
    SETUP_STAMP(6);
    const int npad = (n + 1) & ~1;
    const int nrowent_all = rp_nf;

    // LDS carve (16-byte aligned pieces, no static LDS in front of it)
    double *p_lds = sm;                                   // n
    double *r_lds = p_lds + npad;                         // n: residual, exchanged inside a wave only
    double *minv = r_lds + npad;                          // nf x 36
    double *red0 = minv + 36 * nf;                        // kNW
    double *red1 = red0 + kNW;                            // kNW
    float *Acf = reinterpret_cast<float *>(red1 + kNW + 2); // kNC x kNC floats: inverse coarse matrix of the previous trial
    double *rcg = red1 + kNW + 2 + kNC * kNC / 2;         // kNC: restricted vector of every aggregate
    double *zstrip = rcg + kNC + 32 * wv;                 // 16 per wave: the wave's coarse correction z_c = A_c^-1 P^T r (kPA used)
    double *ustrip = zstrip + 16;                         // 16 per wave: A_c^-1 P^T s of the wave's aggregate
    double z_reg = 0.0, u_reg = 0.0;                      // ... both in registers since round 5 (lane 4 r: coarse row r of the wave's aggregate)
    double *ypart = rcg + kNC + 32 * kNW;                 // 6 doubles per gather-list PAIR (+ one dummy strip); PADDED: kOwnBatch per row
    constexpr int kYDummy = 5 * kOwnBatch + 6;            // PADDED: where the lanes without a pair write (stride kOwnBatch like the others)
    const int ypart_len = PADDED ? n * kOwnBatch + kYDummy + kOwnBatch : 6 * ((nrowent_all >> 1) + 1 + kOwnBatch);
    double *sdiag = ypart + ypart_len;                    // nf x 36: the damped diagonal blocks S_ii
    double *rsw = sdiag + 36 * nf + 64 * wv;              // 64 per wave: the wave's residual exchange (PADDED: zeros at unowned places)
    if (PADDED) {                                         // (ordered before the loop's first writes by the barriers in front of it)
        for (int k = tid; k < ypart_len; k += kT) ypart[k] = 0.0;
    }
    const int nrowent = nrowent_all;
    const int P0 = rp_b0 >> 1, P1 = rp_b1 >> 1;          // the wave's entry pairs
    const int own_p0 = rp_bi >> 1, own_p1 = rp_bi1 >> 1;
    const int own_cnt = own_p1 - own_p0;                  // pair sums of this lane's row (0 for non-owners)
    const int nb = b1 - b0;                               // block rows of this wave (wave-uniform)

    // ---- overflow only: materialise S in L2 for the list tails that do not fit in VGPRs ----
    if (OVERFLOW) {
        // oriented copies of the wave's tail entries (those beyond its 64 VGPR-resident pairs), summed straight from the
        // work-item partials, 8 elements per lane in flight: the mat-vec reads them with wide row-major loads
        const int e0 = 2 * (P0 + 64), ntail = max(2 * P1 - e0, 0) * 36;
        for (int base = ln; base < ntail; base += 8 * 64) {
            int q[8], src[8], hu[8], i0[8], i1[8], pr[8];
            double sv[8], hv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = min(base + 64 * u, ntail - 1);
                const int e = e0 + idx / 36;
                q[u] = idx - (idx / 36) * 36;
                const RowEnt re = w.row_ent[e];
                const int a = q[u] / 6, b = q[u] - a * 6;
                pr[u] = re.block;
                src[u] = re.transposed ? b * 6 + a : q[u];
                hu[u] = 42 + (a <= b ? ut6(a, b) : ut6(b, a));
                i0[u] = re.block >= 0 ? w.pair_item_start[re.block] : 0;
                i1[u] = re.block >= 0 ? w.pair_item_start[re.block + 1] : 0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const double *rec = part + (size_t)i0[u] * kPartStride;
                sv[u] = i1[u] > i0[u] ? rec[src[u]] : 0.0;
                hv[u] = (i1[u] > i0[u] && pr[u] < nf) ? rec[hu[u]] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                for (int itx = i0[u] + 1; itx < i1[u]; ++itx) {
                    const double *rec = part + (size_t)itx * kPartStride;
                    sv[u] += rec[src[u]];
                    if (pr[u] < nf) hv[u] += rec[hu[u]];
                }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + 64 * u;
                if (idx < ntail) {
                    const bool dg = pr[u] >= 0 && pr[u] < nf;
                    w.blocks_ov[(size_t)e0 * 36 + idx] = dg ? (hv[u] + (q[u] % 7 == 0 ? lambda : 0.0)) - sv[u] : -sv[u];
                }
            }
        }
        __syncthreads();
    }

    // ---- this lane's pair of oriented blocks, assembled straight from the partials into VGPRs ----
    // (the host-built lane plan says which blocks and which partial items: one dependent load level)
    int colo[2];
    const int my_pair = P0 + ln;
    const bool have_pair = my_pair < P1;
    int yslot = (have_pair ? my_pair : (nrowent >> 1)) * 6;     // lanes without a pair write the dummy strip
    const double *yown = ypart + own_p0 * 6 + ba;         // first pair sum of this lane's row
    if (PADDED) {
        // the pair's row and its place in the row: the wave's rows' list ranges sit in their owner lanes (lane 6 u: row b0 + u)
        yslot = n * kOwnBatch;                            // (no pair: the dummy strip)
#pragma unroll
        for (int u = 0; u < 10; ++u) {
            const int q0 = __builtin_amdgcn_readlane(rp_bi, 6 * u) >> 1, q1 = __builtin_amdgcn_readlane(rp_bi1, 6 * u) >> 1;
            if (u < nb && have_pair && my_pair >= q0 && my_pair < q1) yslot = (b0 + u) * 6 * kOwnBatch + (my_pair - q0);
        }
        yown = ypart + (owner ? row : 0) * kOwnBatch;
    }
    // ---- diagonal blocks and right-hand side, cooperatively: owner lane (bi, ba) sums ROW ba of S_ii = Hpp + lambda I -
    // sum B Dinv B^T, b_p and B Dinv b_l over the work items of pair (bi, bi), four items in flight: the cost does not
    // grow with the number of items a long diagonal pair is cut into ----
    // ---- diagonal blocks and right-hand side: owner lane (bi, ba) sums ROW ba of S_ii = Hpp - sum B Dinv B^T (+ lambda), of
    // B Dinv b_l and of b_p over the work items of pair (bi, bi).  The schur pass leaves every diagonal item once more in the
    // layout THIS loop reads (DevWindow::rec_d: per item and row 8 contiguous doubles = the row of Hpp - B Dinv B^T, then
    // B Dinv b_l and b_p): four 16-byte loads per item and lane, the six rows of a keyframe on one 384-byte run - until round 4
    // the lane gathered 14 scattered doubles per item from the 72-double partial (6 us per launch on the one CU's texture
    // addresser).  Items in order, four in flight. ----
    double r_r = 0.0;
    double mi[6] = { 1, 1, 1, 1, 1, 1 };                 // this lane's row of S_ii, then of its inverse
    auto wave_lds_sync0 = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    if (owner) {
        double d6[6] = { 0, 0, 0, 0, 0, 0 }, cc = 0.0, bb = 0.0;
        const int ni = oi1 - oi0;
        auto add = [&](const double2 (&v)[4], bool in) {
            d6[0] += in ? v[0].x : 0.0; d6[1] += in ? v[0].y : 0.0; d6[2] += in ? v[1].x : 0.0;
            d6[3] += in ? v[1].y : 0.0; d6[4] += in ? v[2].x : 0.0; d6[5] += in ? v[2].y : 0.0;
            cc += in ? v[3].x : 0.0; bb += in ? v[3].y : 0.0;
        };
#pragma unroll
        for (int u = 0; u < kRecFly; ++u) add(rv[u], u < ni);
        for (int i0 = kRecFly; i0 < ni; i0 += kRecFly) {        // (keyframes with more than 2 048 edges)
            double2 v[kRecFly][4];
#pragma unroll
            for (int u = 0; u < kRecFly; ++u) {
                const double2 *src = rec0 + (size_t)min(i0 + u, ni - 1) * 24;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[u][q] = src[q];
            }
#pragma unroll
            for (int u = 0; u < kRecFly; ++u) add(v[u], i0 + u < ni);
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) d6[q] += q == ba ? lambda : 0.0;
#pragma unroll
        for (int q = 0; q < 6; ++q) sdiag[bi * 36 + ba * 6 + q] = d6[q];
        r_r = bb - cc;
        w.bp[row] = bb;
#pragma unroll
        for (int q = 0; q < 6; ++q) mi[q] = d6[q];
    }
    SETUP_STAMP(7);
    // (coarse level: usable when the previous trial's launch left a valid inverse - never for the first trial; on its way into
    //  LDS since the top of the kernel)
    // ---- block-Jacobi preconditioner: S_ii^-1 by the block's six owner lanes, each holding a row: in-place Gauss-Jordan, the
    // scaled pivot row handed round through a wave-private LDS strip (alternating between two: no wait before the next
    // pivot's write), then symmetrised.  (Until round 4 ONE thread per keyframe factored its block on its own - a 600-flop
    // dependent chain with 50 of 512 threads at work, ~2.5 us in front of every solve.)  A pivot that is not positive fails the
    // solve like a failed factorisation. ----
    {
        bool bad = false;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            double *strip = ((k & 1) ? p_lds : r_lds) + (owner ? bi * 6 : 0);
            if (owner && ba == k) {
                const double p = mi[k];
                if (!(p > 0.0)) bad = true;
                double pinv = __builtin_amdgcn_rcp(p);
                pinv = pinv * (2.0 - p * pinv);
                pinv = pinv * (2.0 - p * pinv);
#pragma unroll
                for (int q = 0; q < 6; ++q) { mi[q] = q == k ? pinv : mi[q] * pinv; strip[q] = mi[q]; }
            }
            wave_lds_sync0();
            if (owner && ba != k) {
                const double2 *sr = reinterpret_cast<const double2 *>(strip);
                const double2 s0 = sr[0], s1 = sr[1], s2 = sr[2];
                const double srow[6] = { s0.x, s0.y, s1.x, s1.y, s2.x, s2.y };
                const double f = mi[k];
#pragma unroll
                for (int q = 0; q < 6; ++q) mi[q] = q == k ? -f * srow[k] : mi[q] - f * srow[q];
            }
        }
        // symmetrise: rows out, columns in
        if (owner) {
#pragma unroll
            for (int q = 0; q < 6; ++q) minv[bi * 36 + ba * 6 + q] = mi[q];
        }
        wave_lds_sync0();
        if (owner) {
            double col[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) col[q] = minv[bi * 36 + q * 6 + ba];
#pragma unroll
            for (int q = 0; q < 6; ++q) mi[q] = 0.5 * (mi[q] + col[q]);
        }
        wave_lds_sync0();
        if (owner) {
#pragma unroll
            for (int q = 0; q < 6; ++q) minv[bi * 36 + ba * 6 + q] = mi[q];
        }
        // (a wave's verdict in a slot of its own: read by everyone behind the workgroup barrier in front of the solve)
        const int wbad = __any(bad) ? 1 : 0;
        if (ln == 0) reinterpret_cast<int *>(red1)[wv] = wbad;
    }
    const bool coarse = pp.use_coarse && ctrial >= 0 && aci_tag == ctrial;
    SETUP_STAMP(0);
    // (a keyframe's diagonal block is read by a lane of the wave that owns its rows: no workgroup barrier between the two)
    wave_lds_sync0();

    // ---- this lane's pair of oriented blocks.  Off-diagonal ones: the schur pass leaves the block of a pair that is ONE work
    // item (almost all are) in both orientations where this kernel's lanes read it - DevWindow::img_b, element q of the block
    // of (thread, slot k) at ((36 k + q) 512 + thread): 72 coalesced 8-byte loads per lane instead of eighteen 16-byte gathers
    // per block from the item's partial (each lane its own three cache lines: 4.5 us per launch).  Pairs cut into several items
    // are still summed from the partials here; diagonal blocks come from LDS. ----
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        colo[k] = 0;
        const int4 pl = k ? pl1 : pl0;
        const bool valid = pl.x >= 0, tr = valid && ((pl.y >> 30) & 1) != 0;
        if (valid) colo[k] = pl.y & 0x3fffffff;
        const bool diag = valid && pl.x < nf, from_img = !OVERFLOW && valid && pl.x >= nf && pl.w - pl.z == 1;
        if (!OVERFLOW) {
            // (every lane read its image slot at the top of the kernel - the loads are coalesced whatever the lanes need; a slot
            //  nothing wrote is not used)
#pragma unroll
            for (int q = 0; q < 36; ++q) Bo[k][q] = from_img ? -Bo[k][q] : 0.0;
        } else {
#pragma unroll
            for (int q = 0; q < 36; ++q) Bo[k][q] = 0.0;
        }
        if (diag) {         // (symmetric: no orientation)
            const double2 *src = reinterpret_cast<const double2 *>(sdiag + pl.x * 36);
#pragma unroll
            for (int q = 0; q < 18; ++q) { const double2 v = src[q]; Bo[k][2 * q] = v.x; Bo[k][2 * q + 1] = v.y; }
        }
        if (__any(valid && !diag && !from_img)) {
            // pairs cut into several work items (and every off-diagonal block of a window whose lists overflow the registers):
            // S_block = - sum over the pair's work items of the 6x6 partial (wide, independent loads), then oriented
            double raw[36];
#pragma unroll
            for (int q = 0; q < 36; ++q) raw[q] = 0.0;
            const bool mine = valid && !diag && !from_img;
            for (int itx = mine ? pl.z : 0; itx < (mine ? pl.w : 0); ++itx) {
                const double2 *src = reinterpret_cast<const double2 *>(part + (size_t)itx * kPartStride);
#pragma unroll
                for (int q = 0; q < 18; ++q) { const double2 v = src[q]; raw[2 * q] -= v.x; raw[2 * q + 1] -= v.y; }
            }
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int q = 0; q < 6; ++q) Bo[k][a * 6 + q] = mine ? (tr ? raw[q * 6 + a] : raw[a * 6 + q]) : Bo[k][a * 6 + q];
        }
    }
    SETUP_STAMP(1);

    double x_r = 0.0, z_r = 0.0;
    // z = Minv r: a block's six rows sit in one wave, so its residuals are exchanged through LDS
    // without a workgroup barrier (LDS operations of one wave execute in order)
    const double2 *mrow = reinterpret_cast<const double2 *>(minv + (owner ? bi * 36 + ba * 6 : 0));
    const double2 *rblk = reinterpret_cast<const double2 *>(PADDED ? rsw + (owner ? (bi - b0) * 6 : 0) : r_lds + (owner ? bi * 6 : 0));
    auto wave_lds_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // Coarse level.  Coarse dof (g, d, a): aggregate g = the block rows of wave g, mode d (0 constant, 1 linear:
    // phi(i) = (i - c_g) / h_g), pose component a (same c_g, h_g as coarse_level.h).  Each wave keeps the coarse correction
    // of ITS aggregate, z_c = (A_c^-1 P^T r)[own 12 rows], in a 16-double LDS strip.  z_c is not recomputed from the
    // residual every iteration: r -= alpha s gives z_c -= alpha u_c with u_c = A_c^-1 P^T s, and s = w + beta s gives
    // u_c = A_c^-1 (P^T w) + beta u_c.  P^T w of every aggregate is published with the dot products, so the one dense
    // product per iteration, A_c^-1[own rows, :] (P^T w), runs right after the reduction barrier, beside the scalar
    // recurrences, and the preconditioner itself only reads two values of the strip.
    const double agg_c = uniform_f64(b0 + 0.5 * (nb - 1)), agg_ih = uniform_f64(1.0 / fmax(1.0, 0.5 * nb));
    const double agg_cc = uniform_f64(0.5 * (nb - 1));                  // the aggregate's centre, counted from its first row
    const double phi = owner ? (bi - agg_c) * agg_ih : 0.0;            // this lane's row in its aggregate's linear mode
    const int crow = min(ln >> 2, kPA - 1), cq = ln & 3;                // coarse product: 4 lanes per coarse row, 24 columns each
    // restriction of a vector held one value per owner lane: P^T v of this wave's aggregate -> rcg[wv * kPA ..]
    auto restrict_own = [&](double v_r) {
        if (PADDED) rsw[ln] = owner ? v_r : 0.0;
        else if (owner) r_lds[row] = v_r;
        wave_lds_sync();
        if (ln < kPA) {
            // rows b0 .. b0+9 at fixed offsets; the ones past the wave's last row are masked by a wave-uniform
            // predicate (PADDED: they hold zeros); lanes 0-5 sum them (constant modes), lanes 6-11 weight them by the linear mode
            const int a = ln < 6 ? ln : ln - 6;
            const double *rb = PADDED ? rsw + a : r_lds + b0 * 6 + a;
            double v[10];
#pragma unroll
            for (int u = 0; u < 10; ++u) v[u] = rb[6 * u];
            if (!PADDED) {
#pragma unroll
                for (int u = 0; u < 10; ++u) v[u] = (u < nb) ? v[u] : 0.0;
            }
            const double tc = (((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]))) + (v[8] + v[9]);
            // sum_u (u - cc) v_u / h = (sum_u u v_u - cc sum_u v_u) / h, with the row offsets u as constants
            const double ti = (((v[1] + 2.0 * v[2]) + (3.0 * v[3] + 4.0 * v[4])) + ((5.0 * v[5] + 6.0 * v[6]) + (7.0 * v[7] + 8.0 * v[8]))) + 9.0 * v[9];
            rcg[wv * kPA + ln] = ln < 6 ? tc : (ti - agg_cc * tc) * agg_ih;
        }
    };
    // The same in two halves for the iteration: the exchange and the ten loads by ALL lanes (the lanes that restrict nothing
    // read a place of the wave's own strip), THEN the wave's dot-product reductions - seven DPP links that used to queue up
    // behind the masked block's wait for its loads -, then the sums and the store by the twelve lanes that restrict.
    const int ra = ln < kPA ? (ln < 6 ? ln : ln - 6) : 0;
    auto restrict_issue = [&](double v_r, double (&v)[10]) {
        if (PADDED) rsw[ln] = owner ? v_r : 0.0;
        else if (owner) r_lds[row] = v_r;
        wave_lds_sync();
        const double *rb = PADDED ? rsw + ra : r_lds + b0 * 6 + ra;
#pragma unroll
        for (int u = 0; u < 10; ++u) v[u] = rb[6 * u];
    };
    auto restrict_finish = [&](double (&v)[10]) {
        if (ln < kPA) {
            if (!PADDED) {
#pragma unroll
                for (int u = 0; u < 10; ++u) v[u] = (u < nb) ? v[u] : 0.0;
            }
            const double tc = (((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]))) + (v[8] + v[9]);
            const double ti = (((v[1] + 2.0 * v[2]) + (3.0 * v[3] + 4.0 * v[4])) + ((5.0 * v[5] + 6.0 * v[6]) + (7.0 * v[7] + 8.0 * v[8]))) + 9.0 * v[9];
            rcg[wv * kPA + ln] = ln < 6 ? tc : (ti - agg_cc * tc) * agg_ih;
        }
    };
    // (A_c^-1 v_c)[own row crow], v_c = rcg (all aggregates, visible after a workgroup barrier): valid in lanes with cq == 0.
    // A_c^-1 sits in LDS rounded to fp32: as a preconditioner the inverse needs no more (same CG iteration counts), and
    // this product reads all 96 x 96 entries every iteration: in fp64 that was more LDS traffic than the rest of the
    // iteration together.  The vector side and the row sums stay fp64: z_c is carried by the recurrence z_c -= alpha u_c,
    // which is only consistent while u_c is an exact linear image of s (fp32 sums put 1e-7 |u_c| of noise into z_c, more
    // than z_c itself once the residual has dropped 7 digits: tried, the solve then ends 1e-8 off the oracle's poses).
    auto coarse_rows = [&]() {
        const float4 *arow = reinterpret_cast<const float4 *>(Acf + (wv * kPA + crow) * kNC + cq * 24);
        const double2 *rcv = reinterpret_cast<const double2 *>(rcg + cq * 24);
        double t0 = 0.0, t1 = 0.0;
#pragma unroll
        for (int q = 0; q < 6; q += 2) {
            const float4 a0 = arow[q], a1 = arow[q + 1];
            const double2 c0 = rcv[2 * q], c1 = rcv[2 * q + 1], c2 = rcv[2 * q + 2], c3 = rcv[2 * q + 3];
            t0 += ((double)a0.x * c0.x + (double)a0.y * c0.y) + ((double)a0.z * c1.x + (double)a0.w * c1.y);
            t1 += ((double)a1.x * c2.x + (double)a1.y * c2.y) + ((double)a1.z * c3.x + (double)a1.w * c3.y);
        }
        double t = t0 + t1;
        t += dpp_mov0<0xb1>(t);
        t += dpp_mov0<0x4e>(t);
        return t;
    };
    if (coarse) {
        restrict_own(r_r);
        u_reg = 0.0;
        __syncthreads();
        const double t = coarse_rows();
        z_reg = t;                                            // (valid in the lanes with cq == 0, lane 4 r for coarse row r)
        __syncthreads();                                      // rcg is rewritten inside the loop
    } else __syncthreads();                                   // (the waves' block-inverse verdicts are in)
    int anybad = 0;
#pragma unroll
    for (int k = 0; k < kNW; ++k) anybad |= reinterpret_cast<const int *>(red1)[k];
    auto precond = [&](double rv) {
        if (PADDED) rsw[ln] = owner ? rv : 0.0;
        else if (owner) r_lds[row] = rv;
        wave_lds_sync();
        const double2 m0 = mrow[0], m1 = mrow[1], m2 = mrow[2];
        const double2 r0 = rblk[0], r1 = rblk[1], r2 = rblk[2];
        double s = (m0.x * r0.x + m0.y * r0.y) + (m1.x * r1.x + m1.y * r1.y) + (m2.x * r2.x + m2.y * r2.y);
        // fresh mode: every aggregate is one keyframe, the coarse space is the whole space and A_c^-1 is S^-1 itself
        if (fresh && coarse) s = 0.0;
        if (coarse) {
            // z_c of the wave's aggregate lives in registers (lane 4 r: row r, where the coarse product leaves it); an owner
            // lane takes the two values of its pose component across the wave (ds_bpermute: the LDS crossbar, no memory)
            const double zc0 = __shfl(z_reg, 4 * ba), zc1 = __shfl(z_reg, 24 + 4 * ba);
            if (owner) s += zc0 + phi * zc1;
        }
        return owner ? s : 0.0;
    };
    // ---- conjugate gradients, single-reduction form (Chronopoulos & Gear): with z = Minv r and w = A z,
    //   gamma' = r.z, delta = w.z  (ONE reduction) -> beta = gamma'/gamma, alpha = gamma' / (delta - beta gamma'/alpha),
    //   p = z + beta p, s = w + beta s (= A p), x += alpha p, r -= alpha s.
    // Same iterates as textbook CG in exact arithmetic; two workgroup barriers and one reduction per iteration instead
    // of three and two.  The restricted residual P^T r follows r_c -= alpha P^T s with P^T s = P^T w + beta P^T s.
    double p_r = 0.0, s_r = 0.0;
    double inv_gamma = 1.0, inv_alpha = 0.0, alpha = 0.0, thresh = 0.0;
    bool fail = anybad != 0;
    bool first = true;
    int iters = 0;
    // How an iteration's checks came out (0 go on, 1 broken down, 2 converged, 3 zero right-hand side): DECIDED where the
    // reduction's sums arrive, TESTED at the top of the next iteration, before anything of it is applied - same iterates, same
    // iteration counts.  A vector compare that a scalar branch waits for costs ~60 cycles (profiles/r05_chain_probe.log), and
    // there were four of them in a row on the chain of every iteration; this way the branch finds its condition long computed.
    int stop = 0;

#ifdef MOVBA_CLOCK_STAMP
    unsigned long long seg[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, seg_last = __builtin_amdgcn_s_memtime();
#endif
    if (!fail) {
        for (iters = 1; iters <= pp.max_iters; ++iters) {
            if (stop) break;
            SEG_STAMP(7);
            // ---- x += alpha p, r -= alpha s, z = Minv r (wave-local) ----
            x_r += alpha * p_r;
            r_r -= alpha * s_r;
            if (coarse) {
                if ((iters & 63) == 0) {
                    // every 64 iterations z_c is recomputed from the residual itself, so that rounding drift of the
                    // recurrence cannot build up in long solves (ill-conditioned windows)
                    __syncthreads();                          // all reads of rcg from the previous iteration are done
                    restrict_own(r_r);
                    __syncthreads();
                    const double zc = coarse_rows();
                    z_reg = zc;
                    __syncthreads();                          // rcg is rewritten before the next reduction barrier
                } else z_reg -= alpha * u_reg;                // z_c -= alpha u_c, in the registers of the lanes that hold them
            }
            // (the preconditioner's exchange and loads in FRONT of that masked read-modify-write were tried: 62.9 - 63.7 us per
            //  launch against 62.2 - 62.7, same box)
            z_r = precond(r_r);
            if (owner) p_lds[row] = z_r;                      // the vector the mat-vec multiplies
            SEG_STAMP(0);
            __syncthreads();                                  // (A) z visible
            SEG_STAMP(1);
            // ---- y = B0 z_c0 + B1 z_c1 for this lane's pair, parked in the wave's strip of ypart ----
            // (branch-free: a lane without a pair multiplies zeros and writes the dummy strip)
            {
                const double2 *pa = reinterpret_cast<const double2 *>(p_lds + colo[0]);
                const double2 *pb = reinterpret_cast<const double2 *>(p_lds + colo[1]);
                const double2 a0 = pa[0], a1 = pa[1], a2 = pa[2], c0 = pb[0], c1 = pb[1], c2 = pb[2];
                double y[6];
#pragma unroll
                for (int a = 0; a < 6; ++a)
                    y[a] = Bo[0][a * 6] * a0.x + Bo[0][a * 6 + 1] * a0.y + Bo[0][a * 6 + 2] * a1.x + Bo[0][a * 6 + 3] * a1.y +
                           Bo[0][a * 6 + 4] * a2.x + Bo[0][a * 6 + 5] * a2.y +
                           Bo[1][a * 6] * c0.x + Bo[1][a * 6 + 1] * c0.y + Bo[1][a * 6 + 2] * c1.x + Bo[1][a * 6 + 3] * c1.y +
                           Bo[1][a * 6 + 4] * c2.x + Bo[1][a * 6 + 5] * c2.y;
                if (PADDED) {
                    double *yo = ypart + yslot;           // the pair's slot in each of its row's six runs
#pragma unroll
                    for (int a = 0; a < 6; ++a) yo[a * kOwnBatch] = y[a];
                } else {
                    double2 *yo = reinterpret_cast<double2 *>(ypart + yslot);
                    yo[0] = make_double2(y[0], y[1]); yo[1] = make_double2(y[2], y[3]); yo[2] = make_double2(y[4], y[5]);
                }
            }
            if (OVERFLOW) for (int pq = P0 + ln + 64; pq < P1; pq += 64) {          // overflow pairs: oriented blocks from the L2 copy
                const int c0 = w.row_ent[2 * pq].col * 6, c1 = w.row_ent[2 * pq + 1].col * 6;
                const double2 *B = reinterpret_cast<const double2 *>(w.blocks_ov + (size_t)(2 * pq) * 36);
                const double2 *pa = reinterpret_cast<const double2 *>(p_lds + c0);
                const double2 *pb = reinterpret_cast<const double2 *>(p_lds + c1);
                const double2 a0 = pa[0], a1 = pa[1], a2 = pa[2], c0v = pb[0], c1v = pb[1], c2v = pb[2];
                double y[6];
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    const double2 u0 = B[3 * a], u1 = B[3 * a + 1], u2 = B[3 * a + 2];
                    const double2 v0 = B[18 + 3 * a], v1 = B[18 + 3 * a + 1], v2 = B[18 + 3 * a + 2];
                    y[a] = (u0.x * a0.x + u0.y * a0.y + u1.x * a1.x + u1.y * a1.y + u2.x * a2.x + u2.y * a2.y) +
                           (v0.x * c0v.x + v0.y * c0v.y + v1.x * c1v.x + v1.y * c1v.y + v2.x * c2v.x + v2.y * c2v.y);
                }
                double2 *yo = reinterpret_cast<double2 *>(ypart + pq * 6);
                yo[0] = make_double2(y[0], y[1]); yo[1] = make_double2(y[2], y[3]); yo[2] = make_double2(y[4], y[5]);
            }
            SEG_STAMP(2);
            wave_lds_sync();
            // owner: add up its row's pair sums in list order; loads issued together, adds in order
            double w_r = 0.0;
            if (PADDED) {
                // the row's ten slots (zeros behind its last pair), five 16-byte loads, the same tree
                const double2 *yb = reinterpret_cast<const double2 *>(yown);
                const double2 v0 = yb[0], v1 = yb[1], v2 = yb[2], v3 = yb[3], v4 = yb[4];
                static_assert(kOwnBatch == 10, "tree written for 10");
                w_r += (((v0.x + v0.y) + (v1.x + v1.y)) + ((v2.x + v2.y) + (v3.x + v3.y))) + (v4.x + v4.y);
                w_r = owner ? w_r : 0.0;
            } else {
                // loads at fixed offsets from the row's first pair sum (no per-load address arithmetic); the slots past the
                // row's end hold other rows' sums (or the padding behind ypart) and are masked out
                const double *yb = yown;
                for (int e = 0; e < own_cnt; e += kOwnBatch, yb += 6 * kOwnBatch) {
                    double v[kOwnBatch];
#pragma unroll
                    for (int u = 0; u < kOwnBatch; ++u) v[u] = yb[6 * u];
#pragma unroll
                    for (int u = 0; u < kOwnBatch; ++u) v[u] = (e + u < own_cnt) ? v[u] : 0.0;
                    // fixed tree over the batch (depth 4 instead of a 10-long chain)
                    static_assert(kOwnBatch == 10, "tree written for 10");
                    w_r += (((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]))) + (v[8] + v[9]);
                }
            }
            SEG_STAMP(3);
            double rv10[10];
            if (coarse) restrict_issue(w_r, rv10);            // P^T w of this wave's aggregate, published with the dot-product partials
            {
                double g = owner ? r_r * z_r : 0.0, d = owner ? w_r * z_r : 0.0;
                wave_sum_dpp2(g, d);
                if (ln == 0) *reinterpret_cast<double2 *>(red0 + 2 * wv) = make_double2(g, d);      // (red0 / red1: one run of 2 kNW doubles)
            }
            if (coarse) restrict_finish(rv10);
            SEG_STAMP(4);
            __syncthreads();                                  // (B) r.z, w.z and P^T w visible; all reads of z done
            SEG_STAMP(5);
            // the eight waves' (r.z, w.z): lane k < 8 reads wave k's pair, three DPP steps add them in the fixed tree
            // ((0 + 1) + (2 + 3)) + ((4 + 5) + (6 + 7)) - ONE 16-byte LDS read per lane and six adds where every lane used to read
            // all sixteen values and add them up itself; requested ahead of the coarse product's reads (LDS answers in order: the
            // scalar recurrences below start while those are still in flight)
            double g, delta;
            {
                const double2 gd = *reinterpret_cast<const double2 *>(red0 + 2 * (ln & 7));
                double a = gd.x, b = gd.y;
                a += dpp_mov0<0xb1>(a); b += dpp_mov0<0xb1>(b);
                a += dpp_mov0<0x4e>(a); b += dpp_mov0<0x4e>(b);
                a += dpp_mov0<0x141>(a); b += dpp_mov0<0x141>(b);       // row_half_mirror: lanes 0 - 3 meet lanes 7 - 4
                g = PCG_UNI(a); delta = PCG_UNI(b);
            }
            // the dense coarse product of this iteration (independent of the scalar recurrences)
            const double yc = coarse ? coarse_rows() : 0.0;
#ifdef MOVBA_PCG_EARLY_CHECKS
            // (the checks where they stood until round 5, four compare -> branch hops on the iteration's chain: A/B builds)
            if (!isfinite(g) || !isfinite(delta)) { stop = 1; ++iters; break; }
            if (first) {
                thresh = PCG_UNI(pp.rel_tol * pp.rel_tol * g);
                if (!(g >= 0.0)) { stop = 1; ++iters; break; }
                if (g == 0.0) { stop = 3; ++iters; break; }
            }
            if (!first && g <= thresh) { stop = 2; ++iters; break; }
            const double beta = PCG_UNI(first ? 0.0 : g * inv_gamma);
            const double den = delta - beta * g * inv_alpha;          // = p.Ap of the new search direction
            if (!(den > 0.0)) { stop = 1; ++iters; break; }
#else
            if (first) thresh = PCG_UNI(pp.rel_tol * pp.rel_tol * g);
            const double beta = PCG_UNI(first ? 0.0 : g * inv_gamma);
            const double den = delta - beta * g * inv_alpha;          // = p.Ap of the new search direction
            {
                // (in the order the checks have always had: not finite, the first residual's sign, a zero right-hand side,
                //  converged, p.Ap not positive)
                const bool nonfin = !(isfinite(g) && isfinite(delta));
                const bool neg = first && !(g >= 0.0), zero = first && g == 0.0, conv = !first && g <= thresh, noden = !(den > 0.0);
                stop = (nonfin || neg) ? 1 : zero ? 3 : conv ? 2 : noden ? 1 : 0;
            }
#endif
            // (reciprocals by v_rcp_f64 + two Newton steps: the IEEE division sequence is ~25 dependent instructions, on the
            //  critical path of every iteration; the two are independent of each other and overlap)
            inv_gamma = PCG_UNI(fast_rcp(g));
            alpha = PCG_UNI(g * fast_rcp(den));
            inv_alpha = PCG_UNI(den * inv_gamma);
            first = false;
            p_r = z_r + beta * p_r;
            s_r = w_r + beta * s_r;
            if (coarse) {
                u_reg = yc + beta * u_reg;
                wave_lds_sync();                              // u_c is read by lanes 0-11 at the top of the next iteration
            }
            SEG_STAMP(6);
        }
    }
    if (stop) { iters = stop == 3 ? 0 : iters - 1; fail = stop == 1; }      // (the loop counted one more before it looked)
    const bool capped = iters > pp.max_iters;
    if (capped) iters = pp.max_iters;
    // (the current pose of this thread's keyframe, for the update at the very end: requested here, a cold round trip that runs
    //  beside the barriers and the reduction below instead of behind them)
    double T0[7] = { 0, 0, 0, 1, 0, 0, 0 };
    int h0 = -1;
    if (tid < w.NP) {
#pragma unroll
        for (int k = 0; k < 7; ++k) T0[k] = w.st[cur].pose[7 * tid + k];
        h0 = w.hidx[tid];
    }
    __syncthreads();
    // ---- no answer from the PCG (a diagonal block was not positive definite, the recurrence broke down, or the cap was
    // reached before the tolerance: a weakly constrained window, where an iterative solve departs from the exact step
    // anyway): park the solve (Ctrl::done = 2 turns every kernel queued behind into a no-op) and tell the host, which
    // queues the direct solver (dense_solve.hip) for this trial and for every later one.  fail / capped are workgroup-uniform.
    if (fail || capped) {
        if (tid == 0) {
            const int np = c->n_pause + 1;
            c->pcg_last_iters = iters;
            c->pcg_total_iters += iters;
            c->solver_mode = 1; c->direct_from = c->n_solves;
            c->n_pause = np;
            c->done = 2;
            __hip_atomic_store(&w.hstat->pause_seq, np, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }

    // ---- outputs: increment, pose part of computeScale(), trial poses (VertexSE3Expmap::oplusImpl) ----
    const double xv = owner ? x_r : 0.0;
    if (owner) { w.xp[row] = xv; p_lds[row] = xv; }
    {
        const double ps = wave_sum_dpp(owner ? xv * (lambda * xv + w.bp[row]) : 0.0);
        if (ln == 0) red0[wv] = ps;
    }
    __syncthreads();
    const double scs = sum_fixed(red0);
    const DevState &S0 = w.st[cur];
    const DevState &S1 = w.st[cur ^ 1];
    for (int i = tid; i < w.NP; i += kT) {
        double T[7], Tn[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) T[k] = i == tid ? T0[k] : S0.pose[7 * i + k];
        const int h = i == tid ? h0 : w.hidx[i];
        if (h >= 0) {
            double u[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) u[k] = p_lds[6 * h + k];
            se3_oplus(u, T, Tn);
        } else {
#pragma unroll
            for (int k = 0; k < 7; ++k) Tn[k] = T[k];
        }
        double R[9];
        quat_to_R(Tn, R);
#pragma unroll
        for (int k = 0; k < 7; ++k) S1.pose[7 * i + k] = Tn[k];
#pragma unroll
        for (int k = 0; k < 9; ++k) S1.Rt[12 * i + k] = R[k];
        S1.Rt[12 * i + 9] = Tn[4]; S1.Rt[12 * i + 10] = Tn[5]; S1.Rt[12 * i + 11] = Tn[6];
    }
#ifdef MOVBA_CLOCK_STAMP
    if (ln == 0) for (int k = 0; k < 8; ++k) c->dbg_wseg[wv][k] += seg[k];
#endif
    if (tid == 0) {
        w.scale_part[w.n_pt_blocks] = scs;
        c->pcg_fail = 0;
        c->pcg_last_iters = iters;
        c->pcg_total_iters += iters;
#ifdef MOVBA_CLOCK_STAMP
        c->dbg_cycles += __builtin_amdgcn_s_memtime() - stamp_c0;
        c->dbg_ticks += __builtin_amdgcn_s_memrealtime() - stamp_t0;
        for (int k = 0; k < 8; ++k) c->dbg_seg[k] += seg[k];
        (void)setup_last;
#endif
    }
}

template <bool OVERFLOW, bool PADDED>
__global__ __launch_bounds__(kT) void k_pcg_rows(DevWindow w, PcgParams pp, int trial) { pcg_rows_body<OVERFLOW, PADDED>(w, pp, trial, blockIdx.x); }

// batched: workgroups 2 i and 2 i + 1 are the solver and the coarse builder of window i (a window in fresh-coarse mode
// has no builder: its second workgroup returns at once)
template <bool OVERFLOW, bool PADDED>
__global__ __launch_bounds__(kT) void k_pcg_rows_b(BatchDev b, int trial)
{
    const int wi = blockIdx.x >> 1, role = blockIdx.x & 1;
    const PcgParams &pp = b.pps[wi];
    if (b.band_bw && b.band_bw[wi] >= 0) return;           // (this window's reduced solve is k_band_b's)
    if (role == 1 && pp.use_coarse != 1) return;
    pcg_rows_body<OVERFLOW, PADDED>(b.wins[wi], pp, trial, role);
}

static_assert(kNW == kPcgPlanWaves && kNC == kCoarseDim && kOwnBatch == kPcgPlanOwnBatch, "pcg_plan.cpp sizes the LDS carve of this kernel");

hipError_t launch_pcg_rows(const DevWindow &w, int nrowent, const PcgParams &pp, int trial, hipStream_t s)
{
    const dim3 g(pp.use_coarse == 1 ? 2 : 1), t(kT);
    const bool padded = pp.padded && !pp.overflow;
    const size_t lds = pcg_rows_lds_bytes(w.nfree, nrowent, padded);
    if (pp.overflow) hipLaunchKernelGGL((k_pcg_rows<true, false>), g, t, lds, s, w, pp, trial);
    else if (padded) hipLaunchKernelGGL((k_pcg_rows<false, true>), g, t, lds, s, w, pp, trial);
    else hipLaunchKernelGGL((k_pcg_rows<false, false>), g, t, lds, s, w, pp, trial);
    return hipGetLastError();
}

hipError_t launch_pcg_rows_batch(const BatchDev &b, bool overflow, bool padded, size_t lds, int trial, hipStream_t s)
{
    // (padded: EVERY window of the batch has the padded layout's plan; the two layouts give the same bits)
    if (overflow) hipLaunchKernelGGL((k_pcg_rows_b<true, false>), dim3(2 * b.n), dim3(kT), lds, s, b, trial);
    else if (padded) hipLaunchKernelGGL((k_pcg_rows_b<false, true>), dim3(2 * b.n), dim3(kT), lds, s, b, trial);
    else hipLaunchKernelGGL((k_pcg_rows_b<false, false>), dim3(2 * b.n), dim3(kT), lds, s, b, trial);
    return hipGetLastError();
}

hipError_t configure_pcg_rows()
{
    const void *fs[6] = { reinterpret_cast<const void *>(k_pcg_rows<false, false>), reinterpret_cast<const void *>(k_pcg_rows<false, true>),
                          reinterpret_cast<const void *>(k_pcg_rows<true, false>),
                          reinterpret_cast<const void *>(k_pcg_rows_b<false, false>), reinterpret_cast<const void *>(k_pcg_rows_b<false, true>),
                          reinterpret_cast<const void *>(k_pcg_rows_b<true, false>) };
    for (const void *f : fs) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace movba
