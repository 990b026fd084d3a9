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
#else
#define BAND_STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ void band_body(const DevWindow &w, int bw_arg)
{
    const int bw = bw_arg & 0xffff;                     // (from bit 16: the test build's park hook, api.cpp band_arg)
    extern __shared__ __attribute__((aligned(16))) double sm[];
    Ctrl *c = w.ctrl;
    if (c->done) return;
    const int tid = threadIdx.x, ln = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cur = c->cur;
    const double lambda = c->lambda;
    const int nf = w.nfree, n = 6 * nf, npad = (n + 1) & ~1, B1 = bw + 1;
#ifdef MOVBA_CLOCK_STAMP
    unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
#endif
    // LDS carve (pcg_plan.cpp: band_lds_bytes): the band, the right-hand side (then y), a second vector (b_p's partner, then x),
    // the reciprocals of the factor's diagonal, a strip for the reductions, the enumeration of the trailing blocks, two failure words
    double *Lb = sm;                                  // nf x (bw + 1) x 36: block (i, j), i - bw <= j <= i, at band_off(i, j)
    double *rhs = Lb + (size_t)nf * B1 * 36;          // n
    double *aux = rhs + npad;                         // n
    double *idg = aux + npad;                         // n: 1 / L_aa of every pivot block (what the backward sweep divides by)
    double *gs = idg + npad;                          // 12
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
    // off-diagonal pairs (i < j): the lower block (j, i) = - (sum over the pair's work items of the 6 x 6 partial)^T, items in order
    // (sixteen elements per thread in flight through the two dependent load levels - pair -> its items' partials -: walked one
    //  element at a time the loop was 32 x 2 cold round trips, 100 us of a 160 us launch at cfg3)
    const int noff = w.npairs - nf;
    constexpr int kFly = 16;
    for (int e0 = tid; e0 < noff * 36; e0 += kBT * kFly) {
        int dst[kFly], src[kFly], it0[kFly], it1[kFly];
#pragma unroll
        for (int u = 0; u < kFly; ++u) {
            const int e = min(e0 + u * kBT, noff * 36 - 1);
            const int pr = e / 36, q = e - pr * 36, a = q / 6, b = q - a * 6, p = nf + pr;
            const int i = w.pair_i[p], j = w.pair_j[p];
            it0[u] = w.pair_item_start[p]; it1[u] = w.pair_item_start[p + 1];
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
    // diagonal blocks and right-hand side from the diagonal items' records (DevWindow::rec_d: per item and row a: the row of
    // Hpp - sum B Dinv B^T, then (sum B Dinv b_l)_a and (b_p)_a), items in order, four items and five elements in flight
    constexpr int kDFly = 5, kItemsFly = 4;
    for (int e0 = tid; e0 < nf * 48; e0 += kBT * kDFly) {
        int ni[kDFly];
        double v[kDFly][kItemsFly];
#pragma unroll
        for (int u = 0; u < kDFly; ++u) {
            const int e = min(e0 + u * kBT, nf * 48 - 1);
            const int h = e / 48, r = e - h * 48;
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
    __syncthreads();
    for (int k = tid; k < n; k += kBT) { const double bb = rhs[k]; w.bp[k] = bb; rhs[k] = bb - aux[k]; }
    __syncthreads();

    BAND_STAMP(0);
    // ---- factorisation S = L L^T, the right-hand side carried along as one more row (y = L^-1 b) ----
    // Step k is ONE sweep over the stacked rows [ D_k ; A_(k+1)k ; ... ; A_(k+m)k ; b_k^T ], one row of six per lane: column by
    // column, pivot p = the diagonal element as updated so far, every row's entry scaled by 1 / sqrt(p) and taken out of the
    // row's later entries with the pivot block's own scaled entries (broadcast from lanes 0 - 5 through v_readlane).  Lanes
    // 0 - 5 come out holding L_kk, every other lane its row of L_ik = A_ik L_kk^-T - a triangular solve - or y_k: in place,
    // no inverse of a pivot block anywhere (backward stable like the reference's own Cholesky, src/Optimizer.cc:535).
    // What a thread touches in a step does not depend on the step but for a common offset (k (bw + 1) 36 doubles into the band,
    // 6 k into the right-hand side): decoded once.
    const int step_stride = B1 * 36;
    // the sweep: lanes 0 - 5 of every sweeping wave hold D_k's rows (identical arithmetic in every wave), lane 6 of wave 0 the
    // right-hand side, the other lanes the rows below, 58 per wave
    int s_off, s_irel;                                 // s_irel: -1 = row of D_k, -2 = right-hand side, >= 0: panel block
    if (ln < 6) { s_off = bw * 36 + ln * 6; s_irel = -1; }
    else {
        const int rr = kSweepRows * wv + ln - 6;
        if (rr == 0) { s_off = 0; s_irel = -2; }
        else { const int irel = (rr - 1) / 6, a = (rr - 1) - irel * 6; s_irel = irel; s_off = ((1 + irel) * B1 + (bw - 1 - irel)) * 36 + a * 6; }
    }
    const int sw_first_irel = wv == 0 ? -1 : (kSweepRows * wv - 1) / 6;      // the first panel block a wave beyond the first has rows of
    // Trailing phase: up to four elements per thread - (a, b) of block (irel, jrel) below the pivot, enumerated by irel so that
    // a short last band (irel >= m) just drops out - or, behind them, component a of the right-hand side of block irel.
    constexpr int kTr = 4;
    const int ntr_full = bw * (bw + 1) / 2 * 36;
    const bool fast_tr = ntr_full + bw * 6 <= kTr * kBT;
    int t_irel[kTr], t_T[kTr], t_A[kTr], t_dst[kTr];      // t_A / t_dst < 0: into the right-hand side (offset -1 - x)
#pragma unroll
    for (int u = 0; u < kTr; ++u) {
        const int idx = tid + u * kBT;
        t_irel[u] = 1 << 20; t_T[u] = 0; t_A[u] = 0; t_dst[u] = 0;
        if (idx < ntr_full) {
            const int pr = idx / 36, q = idx - pr * 36, a = q / 6, b = q - a * 6;
            const int ij = tri[pr], irel = ij >> 8, jrel = ij & 0xff;
            t_irel[u] = irel; t_T[u] = ((1 + irel) * B1 + (bw - 1 - irel)) * 36 + a * 6;
            t_A[u] = ((1 + jrel) * B1 + (bw - 1 - jrel)) * 36 + b * 6;
            t_dst[u] = ((1 + irel) * B1 + (jrel - irel + bw)) * 36 + q;
        } else if (idx < ntr_full + bw * 6) {
            const int r = idx - ntr_full, irel = r / 6, a = r - irel * 6;
            t_irel[u] = irel; t_T[u] = ((1 + irel) * B1 + (bw - 1 - irel)) * 36 + a * 6; t_A[u] = -1; t_dst[u] = -1 - (6 * (1 + irel) + a);
        }
    }
    for (int k = 0; k < nf; ++k) {
        const int m = min(bw, nf - 1 - k);
        double *Lk = Lb + (size_t)k * step_stride;        // what the decoded offsets are relative to
        double v[6];
        if (sw_first_irel < m) {
            const bool act = s_irel < m;
            double *row = s_irel == -2 ? rhs + 6 * k : Lk + s_off;
            if (!act) row = gs;                             // (a place nobody writes here)
            { const double2 *rp = reinterpret_cast<const double2 *>(row);
              const double2 r0 = rp[0], r1 = rp[1], r2 = rp[2];
              v[0] = r0.x; v[1] = r0.y; v[2] = r1.x; v[3] = r1.y; v[4] = r2.x; v[5] = r2.y; }
            bool nonfinite = false, nonpos = false;
            double ri_own = 0.0;
#pragma unroll
            for (int kk = 0; kk < 6; ++kk) {
                const double p = readlane_f64(v[kk], kk);
                if (!isfinite(p)) nonfinite = true;
                if (!(p > 0.0)) nonpos = true;
                // 1 / sqrt(p): v_rsq_f64 and two Newton steps y <- y (1.5 - (p / 2) y^2)
                const double hp = 0.5 * p;
                double y = __builtin_amdgcn_rsq(p);
                y = y * (1.5 - (hp * y) * y);
                y = y * (1.5 - (hp * y) * y);
                ri_own = ln == kk ? y : ri_own;
                v[kk] *= y;
#pragma unroll
                for (int q = kk + 1; q < 6; ++q) v[q] -= v[kk] * readlane_f64(v[kk], q);
            }
            if (act && ln >= 6) {
                double2 *wp = reinterpret_cast<double2 *>(row);
                wp[0] = make_double2(v[0], v[1]); wp[1] = make_double2(v[2], v[3]); wp[2] = make_double2(v[4], v[5]);
            }
            if (wv == 0) {
                if (ln < 6) idg[6 * k + ln] = ri_own;
                if (ln == 0) { if (nonfinite) failw[0] = 1; if (nonpos) failw[1] = 1; }
            }
        }
        // L_kk goes over D_k only when every sweeping wave has read D_k (the trailing phase does not touch block (k, k))
        if (m > 0) __syncthreads();
        if (wv == 0 && ln < 6) {
            double2 *wp = reinterpret_cast<double2 *>(Lk + s_off);
            wp[0] = make_double2(v[0], v[1]); wp[1] = make_double2(v[2], v[3]); wp[2] = make_double2(v[4], v[5]);
        }
        if (m == 0) break;                                  // (the last step has nothing below it)
        BAND_STAMP(1);
        // trailing blocks A_ij -= L_ik L_jk^T (i >= j below k in the band) and right-hand side b_i -= L_ik y_k
        const int ntr = m * (m + 1) / 2 * 36;
        if (fast_tr) {
            double2 t0[kTr], t1[kTr], t2[kTr], u0[kTr], u1[kTr], u2[kTr];
            double old[kTr];
            double *dst[kTr];
#pragma unroll
            for (int u = 0; u < kTr; ++u) {
                const bool act = t_irel[u] < m;
                const double2 *tp = reinterpret_cast<const double2 *>(Lk + (act ? t_T[u] : 0));
                const double2 *ap = reinterpret_cast<const double2 *>(t_A[u] < 0 ? rhs + 6 * k : Lk + (act ? t_A[u] : 0));
                dst[u] = t_dst[u] < 0 ? rhs + 6 * k + (-1 - t_dst[u]) : Lk + t_dst[u];
                if (!act) dst[u] = gs;                      // (a place nobody reads here)
                t0[u] = tp[0]; t1[u] = tp[1]; t2[u] = tp[2]; u0[u] = ap[0]; u1[u] = ap[1]; u2[u] = ap[2];
                old[u] = *dst[u];
            }
#pragma unroll
            for (int u = 0; u < kTr; ++u) {
                const double sdot = ((t0[u].x * u0[u].x + t0[u].y * u0[u].y) + (t1[u].x * u1[u].x + t1[u].y * u1[u].y)) + (t2[u].x * u2[u].x + t2[u].y * u2[u].y);
                if (t_irel[u] < m) *dst[u] = old[u] - sdot;
            }
        } else for (int i0 = tid; i0 < ntr + m * 6; i0 += kBT * kTr) {
            double2 t0[kTr], t1[kTr], t2[kTr], u0[kTr], u1[kTr], u2[kTr];
            double old[kTr];
            double *dst[kTr];
#pragma unroll
            for (int u = 0; u < kTr; ++u) {
                const int idx = min(i0 + u * kBT, ntr + m * 6 - 1);
                const double *Ti, *Aj;
                if (idx < ntr) {
                    const int pr = idx / 36, q = idx - pr * 36, a = q / 6, b = q - a * 6;
                    const int ij = tri[pr], irel = ij >> 8, jrel = ij & 0xff;
                    Ti = Lb + band_off(k + 1 + irel, k, bw) + a * 6;
                    Aj = Lb + band_off(k + 1 + jrel, k, bw) + b * 6;
                    dst[u] = Lb + band_off(k + 1 + irel, k + 1 + jrel, bw) + q;
                } else {
                    const int r = idx - ntr, irel = r / 6, a = r - irel * 6;
                    Ti = Lb + band_off(k + 1 + irel, k, bw) + a * 6;
                    Aj = rhs + 6 * k;
                    dst[u] = rhs + 6 * (k + 1 + irel) + a;
                }
                const double2 *tp = reinterpret_cast<const double2 *>(Ti), *ap = reinterpret_cast<const double2 *>(Aj);
                t0[u] = tp[0]; t1[u] = tp[1]; t2[u] = tp[2]; u0[u] = ap[0]; u1[u] = ap[1]; u2[u] = ap[2];
                old[u] = *dst[u];
            }
#pragma unroll
            for (int u = 0; u < kTr; ++u) {
                const double sdot = ((t0[u].x * u0[u].x + t0[u].y * u0[u].y) + (t1[u].x * u1[u].x + t1[u].y * u1[u].y)) + (t2[u].x * u2[u].x + t2[u].y * u2[u].y);
                if (i0 + u * kBT < ntr + m * 6) *dst[u] = old[u] - sdot;
            }
        }
        __syncthreads();
        BAND_STAMP(2);
    }
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
    // ---- backward sweep L^T x = y by one wave: x_k = L_kk^-T s_k by a triangular solve in lanes 0 - 5 (the partial sums of
    //      the later unknowns broadcast through v_readlane), then s_j -= L_kj^T x_k for the band above ----
    if (wv == 0) {
        const int a6 = min(ln, 5);
        for (int k = nf - 1; k >= 0; --k) {
            const int mk = min(bw, k);
            const double *D = Lb + band_off(k, k, bw);
            // (everything that does not depend on x_k is requested ahead of the chain)
            double lc[6];
#pragma unroll
            for (int cc = 0; cc < 6; ++cc) lc[cc] = D[cc * 6 + a6];                    // column a6 of L_kk
            const double idv = idg[6 * k + a6];
            double s = rhs[6 * k + a6];
            const int jrel0 = ln / 6, a0 = ln - jrel0 * 6;
            const bool up0 = ln < mk * 6;
            const double *L0 = Lb + band_off(k, up0 ? k - 1 - jrel0 : k, bw) + a0;
            double l0[6];
#pragma unroll
            for (int cc = 0; cc < 6; ++cc) l0[cc] = L0[cc * 6];
            double *y0 = rhs + 6 * (up0 ? k - 1 - jrel0 : k) + a0;
            const double y0v = *y0;
            double x[6], x_own = 0.0;
#pragma unroll
            for (int cc = 5; cc >= 0; --cc) {
                const double t = s * idv;
                x[cc] = readlane_f64(t, cc);
                x_own = ln == cc ? x[cc] : x_own;
                s -= lc[cc] * x[cc];
            }
            if (ln < 6) aux[6 * k + ln] = x_own;
            if (up0) {
                double acc = l0[0] * x[0];
#pragma unroll
                for (int cc = 1; cc < 6; ++cc) acc += l0[cc] * x[cc];
                *y0 = y0v - acc;
            }
            for (int l = ln + 64; l < mk * 6; l += 64) {
                const int jrel = l / 6, a = l - jrel * 6, j = k - 1 - jrel;
                const double *L = Lb + band_off(k, j, bw) + a;
                double acc = L[0] * x[0];
#pragma unroll
                for (int cc = 1; cc < 6; ++cc) acc += L[cc * 6] * x[cc];
                rhs[6 * j + a] -= acc;
            }
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
