// k_band: the reduced-system solve of one LM trial as a banded block factorisation in ONE workgroup, the whole lower band of
// S = Hpp + lambda I - sum_l B_il Dinv_l B_jl^T held in LDS for the duration of the solve.
//
// The reference factors S exactly (g2o's LinearSolverCSparse, /root/reference/src/Optimizer.cc:535, run by optimize(10) at
// :754-755).  A local-BA window numbered along its covisibility graph (structure.cpp: covisibility_order) has a banded S:
// cfg3's 50 free keyframes couple over at most 9 neighbours either side, 50 x 10 blocks of 6 x 6 = 144 KB of the CU's 160 KB
// of LDS.  A banded factorisation fills the band and nothing else, so the solve needs no global memory between its first load
// and its last store, no second workgroup, no hand-off between workgroups (what makes the one-launch dense solver
// 109 us at this size) and no stopping rule: Cholesky S = L L^T in 6 x 6 blocks,
//     for k:  one sweep over the stacked rows [ D_k ; A_ik (i in the band below k) ; b_k^T ], a row per lane:
//                 L_kk L_kk^T = D_k,  L_ik = A_ik L_kk^-T,  y_k = L_kk^-1 b_k   (triangular solves, nothing is inverted);
//             A_ij -= L_ik L_jk^T (i >= j in the band);  b_i -= L_ik y_k,
// then the backward sweep L^T x = y by one wave, a triangular solve with L_kk^T per block.  Every sum runs in a fixed order:
// bit-reproducible run to run, solo or batched.  Round 4's version inverted the pivot blocks explicitly (block LDL^T), which is
// only as accurate as the blocks are conditioned, and needed a tuned threshold to hand weak windows to the dense solver;
// the Cholesky sweep is backward stable whatever the conditioning, as the reference's factorisation is.
// The only thing that can go wrong is a pivot that is not positive: NaN / inf in the normal equations fails the trial as a
// failed Cholesky factorisation does in g2o (the trial is rejected: Ctrl::pcg_fail, n_chol_fail); a finite non-positive pivot
// (S is positive definite in exact arithmetic: rounding in a window that is singular but for the LM damping) PARKS the solve
// like a PCG that gives up - the host queues the dense direct solver for this trial and stays with it.
#include <hip/hip_runtime.h>

#include "device_math.h"
#include "device_types.h"
#include "kernels.h"

namespace movba {

namespace {

constexpr int kBT = kBandThreads;       // 512
constexpr int kBW = kBT / 64;
constexpr int kSweepRows = 58;          // rows below the pivot block a wave takes in the factorisation's sweep (lanes 6 - 63)

__device__ __forceinline__ int band_off(int i, int j, int bw) { return (i * (bw + 1) + (j - i + bw)) * 36; }

__device__ __forceinline__ void band_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace

#ifdef MOVBA_CLOCK_STAMP
#define BAND_STAMP(k) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); if (tid == 0) c->dbg_seg2[k] += _t - stamp_last; stamp_last = _t; } while (0)
// fine stamps of ONE block step (k == 10), kept in registers and written once at the end; `dep`: a value of the chain the stamp
// has to stand behind (stamps are scalar instructions, the chain is vector ones)
#define BAND_FINE(i, dep) do { if (k == 10) { const int d_ = __builtin_amdgcn_readfirstlane(__double2loint(dep)); unsigned long long t_; \
    asm volatile("s_nop 7\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "s"(d_) : "memory"); fine[i] = t_; } } while (0)
#else
#define BAND_STAMP(k) do { } while (0)
#define BAND_FINE(i, dep) do { } while (0)
#endif

__device__ __forceinline__ void band_body(const DevWindow &w, int bw_arg)
{
    const int bw = bw_arg & 0xffff;                     // (from bit 16: the test build's park hook, api.cpp band_arg)
    extern __shared__ __attribute__((aligned(16))) double sm[];
    Ctrl *c = w.ctrl;
    const int tid = threadIdx.x, ln = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nf = w.nfree, n = 6 * nf, npad = (n + 1) & ~1, B1 = bw + 1;
    // Everything whose address follows from the kernel's arguments alone is REQUESTED here, in front of the first wait: the
    // controller's words, the first round of the diagonal records, the first round of the pairs' item ranges.  The assembly was
    // four dependent round trips behind the launch boundary (controller -> pair tables -> partials -> records, ~1.5 us each from
    // L2s that the schur pass filled on other XCDs): two now (these, then the partials).
    const int done0 = c->done, cur = c->cur;
    const double lambda = c->lambda;
    constexpr int kDFly = 5, kItemsFly = 4, kFly = 16;
    const int noff = w.npairs - nf;
    int d_ni[kDFly];
    double d_v[kDFly][kItemsFly];
#pragma unroll
    for (int u = 0; u < kDFly; ++u) {
        const int e = min(tid + u * kBT, nf * 48 - 1);
        const int h = e / 48, r = e - h * 48;
        d_ni[u] = w.pair_item_start[h + 1] - w.pair_item_start[h];
        const double *rec = w.rec_d + (size_t)h * w.rec_slots * 48 + r;
#pragma unroll
        for (int t = 0; t < kItemsFly; ++t) d_v[u][t] = rec[(size_t)min(t, w.rec_slots - 1) * 48];
    }
    // (the first kPre of a thread's sixteen elements: 113 off-diagonal pairs, every window of up to fifteen keyframes whole)
    constexpr int kPre = 8;
    int o_i[kPre], o_j[kPre], o_it0[kPre], o_it1[kPre];
#pragma unroll
    for (int u = 0; u < kPre; ++u) {
        const int e = max(min(tid + u * kBT, noff * 36 - 1), 0);
        const int p = nf + e / 36;
        const bool any = u * kBT < noff * 36;           // (workgroup-uniform: this round of elements exists)
        o_i[u] = any ? w.pair_i[p] : 0; o_j[u] = any ? w.pair_j[p] : 0;
        o_it0[u] = any ? w.pair_item_start[p] : 0; o_it1[u] = any ? w.pair_item_start[p + 1] : 0;
    }
    if (done0) return;
#ifdef MOVBA_CLOCK_STAMP
    unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
    unsigned long long fine[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#endif
    // LDS carve (pcg_plan.cpp: band_lds_bytes): the band, the right-hand side (then z), a second vector (b_p's partner, then x),
    // the step's panel times D, a strip for the reductions, the enumeration of the trailing blocks, two failure words
    double *Lb = sm;                                  // nf x (bw + 1) x 36: block (i, j), i - bw <= j <= i, at band_off(i, j)
    double *rhs = Lb + (size_t)nf * B1 * 36;          // n
    double *aux = rhs + npad;                         // n
    double *T = aux + npad;                           // bw x 36 + 6 (+ 2): the step's rows below the pivot and its right-hand side, times D
    double *gs = T + bw * 36 + 8;                     // 12
    int *tri = reinterpret_cast<int *>(gs + 12);      // bw (bw + 1) / 2 pairs (irel << 8 | jrel)
    int *failw = tri + ((bw * (bw + 1) / 2 + 1) & ~1);     // [0] a pivot was NaN / inf, [1] a pivot was not positive

    // ---- assembly ----
    for (int k = tid; k < nf * B1 * 18; k += kBT) reinterpret_cast<double2 *>(Lb)[k] = make_double2(0.0, 0.0);
    for (int pr = tid; pr < bw * (bw + 1) / 2; pr += kBT) {
        int irel = 0;
        while ((irel + 1) * (irel + 2) / 2 <= pr) ++irel;
        tri[pr] = (irel << 8) | (pr - irel * (irel + 1) / 2);
    }
    if (tid == 0) { failw[0] = 0; failw[1] = 0; }
    __syncthreads();
    // (the diagonal first: its records are in flight since the top of the kernel and leave their registers here)
    // diagonal blocks and right-hand side from the diagonal items' records (DevWindow::rec_d: per item and row a: the row of
    // Hpp - sum B Dinv B^T, then (sum B Dinv b_l)_a and (b_p)_a), items in order, four items and five elements in flight
    for (int e0 = tid; e0 < nf * 48; e0 += kBT * kDFly) {
        int ni[kDFly];
        double v[kDFly][kItemsFly];
        const bool first_round = e0 == tid;
#pragma unroll
        for (int u = 0; u < kDFly; ++u) {
            const int e = min(e0 + u * kBT, nf * 48 - 1);
            const int h = e / 48, r = e - h * 48;
            if (first_round) {
                ni[u] = d_ni[u];
#pragma unroll
                for (int t = 0; t < kItemsFly; ++t) v[u][t] = d_v[u][t];
                continue;
            }
            ni[u] = w.pair_item_start[h + 1] - w.pair_item_start[h];
            const double *rec = w.rec_d + (size_t)h * w.rec_slots * 48 + r;
#pragma unroll
            for (int t = 0; t < kItemsFly; ++t) v[u][t] = rec[(size_t)min(t, w.rec_slots - 1) * 48];
        }
#pragma unroll
        for (int u = 0; u < kDFly; ++u) {
            const int e = e0 + u * kBT;
            if (e >= nf * 48) continue;
            const int h = e / 48, r = e - h * 48, a = r >> 3, q = r & 7;
            double s = 0.0;
#pragma unroll
            for (int t = 0; t < kItemsFly; ++t) s += t < ni[u] ? v[u][t] : 0.0;
            const double *rec = w.rec_d + (size_t)h * w.rec_slots * 48 + r;
            for (int t = kItemsFly; t < ni[u]; ++t) s += rec[(size_t)t * 48];
            if (q < 6) Lb[band_off(h, h, bw) + a * 6 + q] = s + (q == a ? lambda : 0.0);
            else if (q == 6) aux[6 * h + a] = s;
            else rhs[6 * h + a] = s;
        }
    }
    // off-diagonal pairs (i < j): the lower block (j, i) = - (sum over the pair's work items of the 6 x 6 partial)^T, items in order
    // (sixteen elements per thread in flight through the two dependent load levels - pair -> its items' partials -: walked one
    //  element at a time the loop was 32 x 2 cold round trips, 100 us of a 160 us launch at cfg3)
    for (int e0 = tid; e0 < noff * 36; e0 += kBT * kFly) {
        int dst[kFly], src[kFly], it0[kFly], it1[kFly];
        const bool first_round = e0 == tid;             // (its pair tables were requested at the top of the kernel)
#pragma unroll
        for (int u = 0; u < kFly; ++u) {
            const int e = min(e0 + u * kBT, noff * 36 - 1);
            const int pr = e / 36, q = e - pr * 36, a = q / 6, b = q - a * 6, p = nf + pr;
            int i = 0, j = 0;
            it0[u] = 0; it1[u] = 0;
            if (first_round && u < kPre) { i = o_i[u]; j = o_j[u]; it0[u] = o_it0[u]; it1[u] = o_it1[u]; }
            else if (e0 - tid + u * kBT < noff * 36) {       // (workgroup-uniform; nothing is loaded for rounds past the end)
                i = w.pair_i[p]; j = w.pair_j[p]; it0[u] = w.pair_item_start[p]; it1[u] = w.pair_item_start[p + 1];
            }
            dst[u] = band_off(j, i, bw) + a * 6 + b;
            src[u] = b * 6 + a;
        }
        double v[kFly];
#pragma unroll
        for (int u = 0; u < kFly; ++u) v[u] = it1[u] > it0[u] ? w.part[(size_t)it0[u] * kPartStride + src[u]] : 0.0;
#pragma unroll
        for (int u = 0; u < kFly; ++u) {
            double s = 0.0 - v[u];
            for (int it = it0[u] + 1; it < it1[u]; ++it) s -= w.part[(size_t)it * kPartStride + src[u]];      // (pairs cut into several items)
            if (e0 + u * kBT < noff * 36) Lb[dst[u]] = s;
        }
    }
    __syncthreads();
    for (int k = tid; k < n; k += kBT) { const double bb = rhs[k]; w.bp[k] = bb; rhs[k] = bb - aux[k]; }
    __syncthreads();

    BAND_STAMP(0);
    // ---- factorisation S = L D L^T (L unit lower, D diagonal: Cholesky without its square roots), the right-hand side carried
    //      along as one more row (z = D^-1 L^-1 b) ----
    // Step k is ONE sweep over the stacked rows [ D_k ; A_(k+1)k ; ... ; A_(k+m)k ; b_k^T ], one row of six per lane: column by
    // column, pivot d = the diagonal element as updated so far (broadcast from its lane through v_readlane, like the pivot block's
    // other entries of the column), every row's entry divided by d - hardware reciprocal and ONE cubic step, 2^-22 -> 2^-66 - and
    // taken out of the row's later entries.  A lane keeps both forms of its row: divided (L: in place in the band) and as it
    // stood when its column came up (L D: into the panel T), because the trailing update A_ij -= L_ik D_k L_jk^T multiplies one by
    // the other.  No square root, no inverse of a pivot block: the dependent chain per pivot is a v_readlane hop, the
    // reciprocal, three fused multiply-adds and the update of the next pivot (~80 cycles: a dependent v_fma_f64 costs 4, a
    // v_readlane or v_rcp_f64 link ~20 more, profiles/r05_chain_probe.log), and the factorisation is backward stable like the
    // reference's Cholesky (src/Optimizer.cc:535): the only thing that can go wrong is a pivot that is not positive.
    // Where a block step's ~2 600 cycles go at cfg3's band of nine (stamps inside one step, profiles/r05_band_stamps_cfg3.log):
    // the sweeping wave's three 16-byte LDS loads ~210, the six columns ~480, its six 16-byte stores ~320 (a lone wave gets half
    // the LDS store rate and drains before the barrier), the trailing phase ~1 500 (18 wide loads, 9 + 9 eight-byte loads and
    // stores, 54 fp64 operations per thread on one wave per SIMD), the two barriers ~100 together.
    // What a thread touches in a step does not depend on the step but for a common offset (k (bw + 1) 36 doubles into the band,
    // 6 k into the right-hand side): decoded once.
    const int step_stride = B1 * 36;
    // the sweep: lanes 0 - 5 of every sweeping wave hold D_k's rows (identical arithmetic in every wave), lane 6 of wave 0 the
    // right-hand side, the other lanes the rows below, 58 per wave
    int s_off, s_irel, s_toff;                         // s_irel: -1 = row of D_k, -2 = right-hand side, >= 0: panel block; s_toff: the row's place in T
    if (ln < 6) { s_off = bw * 36 + ln * 6; s_irel = -1; s_toff = 0; }
    else {
        const int rr = kSweepRows * wv + ln - 6;
        if (rr == 0) { s_off = 0; s_irel = -2; s_toff = bw * 36; }
        else { const int irel = (rr - 1) / 6, a = (rr - 1) - irel * 6; s_irel = irel; s_off = ((1 + irel) * B1 + (bw - 1 - irel)) * 36 + a * 6; s_toff = (rr - 1) * 6; }
    }
    const int sw_first_irel = wv == 0 ? -1 : (kSweepRows * wv - 1) / 6;      // the first panel block a wave beyond the first has rows of
    // Trailing phase: a thread takes a 3 x 3 piece of a block below the pivot - three rows of L_ik, three rows of (L D)_jk, nine
    // dot products of six - or three components of a block of the right-hand side; up to three such pieces per thread.  (A third of
    // the LDS bytes of the element-by-element form, which re-read two 48-byte rows per element - and the same ~1 430 cycles per
    // step: 36 LDS instructions per thread at a wave per SIMD, each wave waiting out its own loads; 3 x 2 pieces over all eight
    // waves: 1 690.)
    // Pieces are enumerated by the block row below the pivot (irel), so that a short last band (irel >= m) just drops out.
    constexpr int kTr = 3;
    const int npair = bw * (bw + 1) / 2, nblk_items = npair * 4, nitems_full = nblk_items + bw * 2;
    const bool fast_tr = nitems_full <= kTr * kBT;
    int t_irel[kTr], t_A[kTr], t_B[kTr], t_dst[kTr];     // t_B < 0: the right-hand side (three components at t_dst)
#pragma unroll
    for (int u = 0; u < kTr; ++u) {
        const int idx = tid + u * kBT;
        t_irel[u] = 1 << 20; t_A[u] = 0; t_B[u] = 0; t_dst[u] = 0;
        if (idx < nblk_items) {
            const int pr = idx >> 2, a3 = (idx >> 1) & 1, b3 = idx & 1;
            const int ij = tri[pr], irel = ij >> 8, jrel = ij & 0xff;
            t_irel[u] = irel;
            t_A[u] = ((1 + irel) * B1 + (bw - 1 - irel)) * 36 + a3 * 18;          // rows 3 a3 ... of L_ik (relative to the step's base)
            t_B[u] = jrel * 36 + b3 * 18;                                         // rows 3 b3 ... of (L D)_jk in T
            t_dst[u] = ((1 + irel) * B1 + (jrel - irel + bw)) * 36 + a3 * 18 + b3 * 3;
        } else if (idx < nitems_full) {
            const int r = idx - nblk_items, irel = r >> 1, a3 = r & 1;
            t_irel[u] = irel; t_A[u] = ((1 + irel) * B1 + (bw - 1 - irel)) * 36 + a3 * 18; t_B[u] = -1; t_dst[u] = 6 * (1 + irel) + a3 * 3;
        }
    }
    // (conditions a scalar branch waits for are kept out of the step loop: a vector compare feeding a branch costs ~60 cycles,
    //  profiles/r05_chain_probe.log - the pivots' verdicts are collected in scalar masks and stored once, behind the loop; which
    //  rounds of pieces a wave has anything to do in is known per wave before the loop)
    int t_first[kTr];                                   // the smallest irel among the wave's pieces of round u
#pragma unroll
    for (int u = 0; u < kTr; ++u) {
        int mn = t_irel[u];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mn = min(mn, __shfl_xor(mn, o));
        t_first[u] = __builtin_amdgcn_readfirstlane(mn);
    }
    bool nonfinite = false, nonpos = false;
    for (int k = 0; k < nf; ++k) {
        const int m = min(bw, nf - 1 - k);
        double *Lk = Lb + (size_t)k * step_stride;        // what the decoded offsets are relative to
        double v[6];
        BAND_FINE(0, lambda);
        if (sw_first_irel < m) {
            const bool act = s_irel < m;
            double *row = s_irel == -2 ? rhs + 6 * k : Lk + s_off;
            if (!act) row = gs;                             // (a place nobody writes here)
            { const double2 *rp = reinterpret_cast<const double2 *>(row);
              const double2 r0 = rp[0], r1 = rp[1], r2 = rp[2];
              v[0] = r0.x; v[1] = r0.y; v[2] = r1.x; v[3] = r1.y; v[4] = r2.x; v[5] = r2.y; }
            BAND_FINE(1, v[5]);
            double wu[6];                                   // the row as it stood when its column came up: (L D)
#pragma unroll
            for (int kk = 0; kk < 6; ++kk) {
                const double d = readlane_f64(v[kk], kk);
                double cq[6];
#pragma unroll
                for (int q = kk + 1; q < 6; ++q) cq[q] = readlane_f64(v[kk], q);      // d l_qk of the pivot block's rows below
                nonfinite |= !isfinite(d);
                nonpos |= !(d > 0.0);
                const double r0 = __builtin_amdgcn_rcp(d);
                const double e = __builtin_fma(-d, r0, 1.0), t = __builtin_fma(e, e, e);
                const double g = v[kk] * r0;
                wu[kk] = v[kk];
                v[kk] = __builtin_fma(g, t, g);
#pragma unroll
                for (int q = kk + 1; q < 6; ++q) v[q] = __builtin_fma(-v[kk], cq[q], v[q]);
            }
            BAND_FINE(2, v[5]);
            if (act && ln >= 6) {
                double2 *wp = reinterpret_cast<double2 *>(row);
                wp[0] = make_double2(v[0], v[1]); wp[1] = make_double2(v[2], v[3]); wp[2] = make_double2(v[4], v[5]);
                double2 *tp = reinterpret_cast<double2 *>(T + s_toff);
                tp[0] = make_double2(wu[0], wu[1]); tp[1] = make_double2(wu[2], wu[3]); tp[2] = make_double2(wu[4], wu[5]);
            }
        }
        BAND_FINE(3, lambda);
        // L_kk goes over D_k only when every sweeping wave has read D_k (the trailing phase does not touch block (k, k))
        if (m > 0) __syncthreads();
        BAND_FINE(4, lambda);
        if (wv == 0 && ln < 6) {
            double2 *wp = reinterpret_cast<double2 *>(Lk + s_off);
            wp[0] = make_double2(v[0], v[1]); wp[1] = make_double2(v[2], v[3]); wp[2] = make_double2(v[4], v[5]);
        }
        if (m == 0) break;                                  // (the last step has nothing below it)
        BAND_STAMP(1);
        // trailing blocks A_ij -= L_ik (L D)_jk^T (i >= j below k in the band) and right-hand side b_i -= L_ik (D z)_k
        auto piece = [&](const double *A, const double *Bq, double *dst, bool is_rhs, bool live) {
            double2 ar[3][3];
#pragma unroll
            for (int ra = 0; ra < 3; ++ra) { const double2 *ap = reinterpret_cast<const double2 *>(A + 6 * ra); ar[ra][0] = ap[0]; ar[ra][1] = ap[1]; ar[ra][2] = ap[2]; }
            if (is_rhs) {
                const double2 *yp = reinterpret_cast<const double2 *>(Bq);
                const double2 y0 = yp[0], y1 = yp[1], y2 = yp[2];
                double old[3];
#pragma unroll
                for (int ra = 0; ra < 3; ++ra) old[ra] = dst[ra];
#pragma unroll
                for (int ra = 0; ra < 3; ++ra) {
                    const double sdot = ((ar[ra][0].x * y0.x + ar[ra][0].y * y0.y) + (ar[ra][1].x * y1.x + ar[ra][1].y * y1.y)) + (ar[ra][2].x * y2.x + ar[ra][2].y * y2.y);
                    if (live) dst[ra] = old[ra] - sdot;
                }
            } else {
                double2 br[3][3];
#pragma unroll
                for (int rb = 0; rb < 3; ++rb) { const double2 *bp = reinterpret_cast<const double2 *>(Bq + 6 * rb); br[rb][0] = bp[0]; br[rb][1] = bp[1]; br[rb][2] = bp[2]; }
                double old[3][3];
#pragma unroll
                for (int ra = 0; ra < 3; ++ra)
#pragma unroll
                    for (int rb = 0; rb < 3; ++rb) old[ra][rb] = dst[6 * ra + rb];
#pragma unroll
                for (int ra = 0; ra < 3; ++ra)
#pragma unroll
                    for (int rb = 0; rb < 3; ++rb) {
                        const double sdot = ((ar[ra][0].x * br[rb][0].x + ar[ra][0].y * br[rb][0].y) + (ar[ra][1].x * br[rb][1].x + ar[ra][1].y * br[rb][1].y)) +
                                            (ar[ra][2].x * br[rb][2].x + ar[ra][2].y * br[rb][2].y);
                        if (live) dst[6 * ra + rb] = old[ra][rb] - sdot;
                    }
            }
        };
        if (fast_tr) {
#pragma unroll
            for (int u = 0; u < kTr; ++u) {
                if (t_first[u] >= m) continue;                          // (nothing for this wave in this round)
                const bool live = t_irel[u] < m, is_rhs = t_B[u] < 0;
                const double *A = Lk + (live ? t_A[u] : 0);
                const double *Bq = is_rhs ? T + bw * 36 : T + (live ? t_B[u] : 0);
                double *dst = live ? (is_rhs ? rhs + 6 * k + t_dst[u] : Lk + t_dst[u]) : gs;
                if (is_rhs) piece(A, Bq, dst, true, live); else piece(A, Bq, dst, false, live);
            }
        } else for (int i0 = tid; i0 < nitems_full; i0 += kBT) {
            int irel, a3, b3 = 0, jrel = 0; bool is_rhs = false;
            if (i0 < nblk_items) { const int ij = tri[i0 >> 2]; irel = ij >> 8; jrel = ij & 0xff; a3 = (i0 >> 1) & 1; b3 = i0 & 1; }
            else { const int r = i0 - nblk_items; irel = r >> 1; a3 = r & 1; is_rhs = true; }
            if (irel >= m) continue;
            const double *A = Lb + band_off(k + 1 + irel, k, bw) + a3 * 18;
            if (is_rhs) piece(A, T + bw * 36, rhs + 6 * (k + 1 + irel) + a3 * 3, true, true);
            else piece(A, T + jrel * 36 + b3 * 18, Lb + band_off(k + 1 + irel, k + 1 + jrel, bw) + a3 * 18 + b3 * 3, false, true);
        }
        BAND_FINE(5, lambda);
        __syncthreads();
        BAND_FINE(6, lambda);
        BAND_STAMP(2);
    }
    if (tid == 0) { failw[0] = nonfinite; failw[1] = nonpos; }      // (wave 0 sweeps in every step)
    __syncthreads();
    const bool fail = failw[0] != 0;
    bool park = failw[1] != 0 && !fail;
#ifdef MOVBA_TEST_HOOKS
    if ((bw_arg >> 16) - 1 == c->n_solves) park = true;         // (test hook: this trial's factorisation is to hand the solve over)
#endif
    if (park) {
        // A pivot that is not positive where every input is finite: S is positive definite in exact arithmetic (Hpp + lambda I
        // minus a Schur complement), so the window is ill conditioned beyond what this ordering carries in fp64.  Park the solve
        // (Ctrl::done = 2 turns every kernel queued behind into a no-op) and tell the host, which queues the dense direct solver
        // for this trial and every later one: that factorisation decides whether the trial fails as a failed Cholesky does in g2o.
        if (tid == 0) {
            const int np = c->n_pause + 1;
            c->pcg_last_iters = 0;
            c->solver_mode = 1; c->direct_from = c->n_solves;
            c->n_pause = np;
            c->done = 2;
            __hip_atomic_store(&w.hstat->pause_seq, np, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    // ---- backward sweep L^T x = z (L unit lower) by one wave.  Block k's unknowns in lanes 0 - 5: what the blocks from k + 2 on
    //      contribute is in the right-hand side already (lanes 6 ... of earlier steps, through LDS, two steps and more ahead of its
    //      use); block k + 1's contribution is added HERE, from its x in scalar registers, and the six unknowns follow by
    //      back-substitution through v_readlane - the chain of a step holds no LDS round trip and no division.  (~970 cycles per
    //      step all the same, as with the contributions passed through LDS: six v_readlane hops with the multiply-adds between
    //      them are ~150 cycles; the rest is this one wave's eighteen LDS loads of the step ahead and its stores) ----
    if (wv == 0) {
        const int a6 = min(ln, 5);
        double xs[6] = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };    // x of block k + 1 (wave-uniform)
        // what a step reads of the factor does not depend on any x: requested ONE STEP AHEAD, in the shadow of the step's chain
        const int li = ln - 6, q0 = 2 + li / 6, qa = li - (li / 6) * 6;
        double lc[6], ln1[6], lq[6], s_cur;
        auto fetch = [&](int k, double (&c_)[6], double (&n_)[6], double (&q_)[6], double &s_) {
            const int mk = min(bw, k);
            const double *D = Lb + band_off(k, k, bw);
#pragma unroll
            for (int cc = 0; cc < 6; ++cc) c_[cc] = D[cc * 6 + a6];                    // column a6 of L_kk (entries below the diagonal are used)
            const bool below = k + 1 < nf;
            const double *L1 = Lb + band_off(below ? k + 1 : k, k, bw) + a6;           // column a6 of L_(k+1)k
#pragma unroll
            for (int cc = 0; cc < 6; ++cc) n_[cc] = below ? L1[cc * 6] : 0.0;
            const bool up = ln >= 6 && q0 <= mk;
            const double *Lq = Lb + band_off(k, up ? k - q0 : k, bw) + (up ? qa : 0);  // column qa of L_k(k-q0)
#pragma unroll
            for (int cc = 0; cc < 6; ++cc) q_[cc] = Lq[cc * 6];
            s_ = rhs[6 * k + a6];      // (complete but for block k + 1's contribution, which the step adds from registers)
        };
        fetch(nf - 1, lc, ln1, lq, s_cur);
        for (int k = nf - 1; k >= 0; --k) {
            const int mk = min(bw, k);
            double lc2[6], ln2[6], lq2[6], s_nxt = 0.0;
            if (k > 0) fetch(k - 1, lc2, ln2, lq2, s_nxt);
            // block k + 1's contribution, then the block's own back-substitution
            double s = s_cur - (((ln1[0] * xs[0] + ln1[1] * xs[1]) + (ln1[2] * xs[2] + ln1[3] * xs[3])) + (ln1[4] * xs[4] + ln1[5] * xs[5]));
            double x_own = 0.0;
#pragma unroll
            for (int cc = 5; cc >= 0; --cc) {
                xs[cc] = readlane_f64(s, cc);
                x_own = ln == cc ? xs[cc] : x_own;
                s = __builtin_fma(-lc[cc], xs[cc], s);
            }
            if (ln < 6) aux[6 * k + ln] = x_own;
            // lanes 6 ...: what x_k contributes to the blocks k - 2 ... k - mk, ADDED in LDS (ds_add_f64: nothing read back; a block
            // receives one addition per step, in step order); block k - 1 is served in registers by the next step
            if (ln >= 6 && q0 <= mk) {
                const double acc = ((lq[0] * xs[0] + lq[1] * xs[1]) + (lq[2] * xs[2] + lq[3] * xs[3])) + (lq[4] * xs[4] + lq[5] * xs[5]);
                atomicAdd(rhs + 6 * (k - q0) + qa, -acc);
            }
            for (int l = li + 58; 2 + l / 6 <= mk; l += 58) {        // (bands wider than ten blocks)
                if (ln < 6) break;
                const int q = 2 + l / 6, a = l - (l / 6) * 6;
                const double *L = Lb + band_off(k, k - q, bw) + a;
                double acc = L[0] * xs[0];
#pragma unroll
                for (int cc = 1; cc < 6; ++cc) acc += L[cc * 6] * xs[cc];
                atomicAdd(rhs + 6 * (k - q) + a, -acc);
            }
#pragma unroll
            for (int cc = 0; cc < 6; ++cc) { lc[cc] = lc2[cc]; ln1[cc] = ln2[cc]; lq[cc] = lq2[cc]; }
            s_cur = s_nxt;
            band_wave_sync();
        }
    }
    __syncthreads();

    BAND_STAMP(4);
    // ---- outputs: increment, pose part of computeScale(), trial poses (VertexSE3Expmap::oplusImpl): as the PCG's epilogue ----
    double ps = 0.0;
    for (int r = tid; r < n; r += kBT) {
        const double xv = fail ? 0.0 : aux[r];
        w.xp[r] = xv;
        aux[r] = xv;
        ps += xv * (lambda * xv + w.bp[r]);
    }
    ps = wave_sum_dpp(ps);
    double *red = gs;
    __syncthreads();                                     // (aux final for everyone; gs free)
    if (ln == 0) red[wv] = ps;
    __syncthreads();
    double scs = 0.0;
#pragma unroll
    for (int k = 0; k < kBW; ++k) scs += red[k];
    const DevState &S0 = w.st[cur];
    const DevState &S1 = w.st[cur ^ 1];
    for (int i = tid; i < w.NP; i += kBT) {
        double Tq[7], Tn[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) Tq[k] = S0.pose[7 * i + k];
        const int h = w.hidx[i];
        if (h >= 0 && !fail) {          // (a failed factorisation moves nothing: g2o returns from solve() before its update)
            double u[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) u[k] = aux[6 * h + k];
            se3_oplus(u, Tq, Tn);
        } else {
#pragma unroll
            for (int k = 0; k < 7; ++k) Tn[k] = Tq[k];
        }
        double R[9];
        quat_to_R(Tn, R);
#pragma unroll
        for (int k = 0; k < 7; ++k) S1.pose[7 * i + k] = Tn[k];
#pragma unroll
        for (int k = 0; k < 9; ++k) S1.Rt[12 * i + k] = R[k];
        S1.Rt[12 * i + 9] = Tn[4]; S1.Rt[12 * i + 10] = Tn[5]; S1.Rt[12 * i + 11] = Tn[6];
    }
    BAND_STAMP(5);
#ifdef MOVBA_CLOCK_STAMP
    if (tid == 0) for (int i = 0; i < 7; ++i) c->dbg_wseg[0][i] += i ? fine[i] - fine[i - 1] : 0;
#endif
    if (tid == 0) {
        w.scale_part[w.n_pt_blocks] = scs;
        c->pcg_fail = fail ? 1 : 0;
        c->pcg_last_iters = -2;                     // trace marker: this trial was solved by the banded factorisation
        c->n_band += 1;
        if (fail) c->n_chol_fail += 1;
    }
}

__global__ __launch_bounds__(kBT) void k_band(DevWindow w, int bw) { band_body(w, bw); }
// batched: workgroup i is window i of the group (a window solved by the PCG leaves at once)
__global__ __launch_bounds__(kBT) void k_band_b(BatchDev b, int)
{
    const int bw = b.band_bw[blockIdx.x];
    if (bw < 0) return;
    band_body(b.wins[blockIdx.x], bw);
}

hipError_t launch_band(const DevWindow &w, int bw, hipStream_t s)
{
    hipLaunchKernelGGL(k_band, dim3(1), dim3(kBT), band_lds_bytes(w.nfree, bw & 0xffff), s, w, bw);
    return hipGetLastError();
}

hipError_t launch_band_batch(const BatchDev &b, size_t lds, hipStream_t s)
{
    hipLaunchKernelGGL(k_band_b, dim3(b.n), dim3(kBT), lds, s, b, 0);
    return hipGetLastError();
}

hipError_t configure_band()
{
    const void *fs[2] = { reinterpret_cast<const void *>(k_band), reinterpret_cast<const void *>(k_band_b) };
    for (const void *f : fs) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace movba
