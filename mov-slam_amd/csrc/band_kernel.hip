// k_band: the reduced-system solve of one LM trial as a banded block factorisation in ONE workgroup, the whole lower band of
// S = Hpp + lambda I - sum_l B_il Dinv_l B_jl^T held in LDS for the duration of the solve.
//
// The reference factors S exactly (g2o's LinearSolverCSparse, /root/reference/src/Optimizer.cc:535, run by optimize(10) at
// :754-755).  A local-BA window numbered along its covisibility graph (structure.cpp: covisibility_order) has a banded S:
// cfg3's 50 free keyframes couple over at most 9 neighbours either side, 50 x 10 blocks of 6 x 6 = 144 KB of the CU's 160 KB
// of LDS.  A banded factorisation fills the band and nothing else, so the solve needs no global memory between its first load
// and its last store, no second workgroup, no hand-off between workgroups (what makes the one-launch dense solver
// 109 us at this size) and no stopping rule: block LDL^T with 6 x 6 pivot blocks,
//     for k:  Dk^-1 (in place, Gauss-Jordan by six lanes of one wave);  T_i = A_ik Dk^-1 (i in the band below k);
//             A_ij -= T_i A_jk^T (i >= j in the band);  b_i -= T_i b_k;  A_ik <- T_i (= L_ik),
// the right-hand side carried along as one more column, then z_k = Dk^-1 y_k and the backward sweep x_j -= L_kj^T x_k by one
// wave.  Every sum runs in a fixed order: bit-reproducible run to run, solo or batched.
// A pivot block that is not positive definite fails the trial as a failed Cholesky factorisation does in g2o (the trial is
// rejected: Ctrl::pcg_fail, n_chol_fail).  The pivot blocks are inverted explicitly, which is only as accurate as they are
// well conditioned: a pivot that has lost more than five digits against its diagonal element of S (keyframes held by a
// handful of observations: the reduced matrix is singular but for the LM damping) PARKS the solve like a PCG that gives up -
// the host queues the dense direct solver (backward stable) for this trial and stays with it.
#include <hip/hip_runtime.h>

#include "device_math.h"
#include "device_types.h"
#include "kernels.h"

namespace movba {

namespace {

constexpr int kBT = kBandThreads;       // 512
constexpr int kBW = kBT / 64;
constexpr double kBandPivotTol = 1e-5;  // a pivot below this share of its diagonal element of S: the window goes to the dense direct solver

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

__device__ __forceinline__ void band_body(const DevWindow &w, int bw)
{
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
    // LDS carve: the band, the right-hand side (then y), a second vector (b_p's partner, then z and x), the panel, two strips
    // for the pivot block's elimination, the enumeration of the trailing blocks, a failure word
    double *Lb = sm;                                  // nf x (bw + 1) x 36: block (i, j), i - bw <= j <= i, at band_off(i, j)
    double *rhs = Lb + (size_t)nf * B1 * 36;          // n
    double *aux = rhs + npad;                         // n
    double *dg = aux + npad;                          // n: the diagonal of S as assembled (what the pivots are measured against)
    double *T = dg + npad;                            // bw x 36
    double *gs = T + bw * 36;                         // 12
    int *tri = reinterpret_cast<int *>(gs + 12);      // bw (bw + 1) / 2 pairs (irel << 8 | jrel)
    int *failw = tri + ((bw * (bw + 1) / 2 + 1) & ~1);     // [0] a pivot was not positive, [1] a pivot lost too many digits

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
            if (q < 6) { Lb[band_off(h, h, bw) + a * 6 + q] = s + (q == a ? lambda : 0.0); if (q == a) dg[6 * h + a] = s + lambda; }
            else if (q == 6) aux[6 * h + a] = s;
            else rhs[6 * h + a] = s;
        }
    }
    __syncthreads();
    for (int k = tid; k < n; k += kBT) { const double bb = rhs[k]; w.bp[k] = bb; rhs[k] = bb - aux[k]; }
    __syncthreads();

    BAND_STAMP(0);
    // ---- factorisation, the right-hand side carried along ----
    // What a thread touches in a step does not depend on the step but for a common offset (k (bw + 1) 36 doubles into the band,
    // 6 k into the right-hand side): decoded once.  Panel: element (a, b) of panel block `p_irel`.  Trailing phase: up to four
    // elements per thread - (a, b) of block (irel, jrel) below the pivot, enumerated by irel so that a short last band
    // (irel >= m) just drops out - or, behind them, component a of the right-hand side of block irel.
    const int step_stride = B1 * 36;
    const int p_irel = tid / 36;
    int p_A, p_D, p_Dc;
    { const int q = tid - p_irel * 36, a = q / 6, b = q - a * 6;
      p_A = ((1 + p_irel) * B1 + (bw - 1 - p_irel)) * 36 + a * 6; p_D = bw * 36 + b * 6; p_Dc = bw * 36 + b; }
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
            t_irel[u] = irel; t_T[u] = irel * 36 + a * 6;
            t_A[u] = ((1 + jrel) * B1 + (bw - 1 - jrel)) * 36 + b * 6;
            t_dst[u] = ((1 + irel) * B1 + (jrel - irel + bw)) * 36 + q;
        } else if (idx < ntr_full + bw * 6) {
            const int r = idx - ntr_full, irel = r / 6, a = r - irel * 6;
            t_irel[u] = irel; t_T[u] = irel * 36 + a * 6; t_A[u] = -1; t_dst[u] = -1 - (6 * (1 + irel) + a);
        }
    }
    for (int k = 0; k < nf; ++k) {
        const int m = min(bw, nf - 1 - k);
        double *D = Lb + band_off(k, k, bw);
        double *Lk = Lb + (size_t)k * step_stride;        // what the decoded offsets are relative to
        if (wv == 0) {
            // Dk^-1 in place: Gauss-Jordan with the block's rows in lanes 0 - 5 and the pivot row broadcast through v_readlane
            // (scalar operands of the other lanes' multiply-adds): no LDS round trip inside the elimination - through LDS strips,
            // as the PCG's block-Jacobi setup does it, a block took 2 960 cycles of this kernel's critical path, 50 times
            double mi[6] = { 1, 0, 0, 0, 0, 0 };
            const int lr = min(ln, 5);
#pragma unroll
            for (int q = 0; q < 6; ++q) mi[q] = D[lr * 6 + q];
            const double dref = dg[6 * k + lr] * kBandPivotTol;       // this lane's row: its pivot must keep that much of S's diagonal
            bool bad = false;
            double p_own = 1.0;                                       // this lane's own pivot (lanes 0 - 5), compared once behind the elimination
#pragma unroll
            for (int kk = 0; kk < 6; ++kk) {
                double r[6];
#pragma unroll
                for (int q = 0; q < 6; ++q) r[q] = readlane_f64(mi[q], kk);
                const double p = r[kk];
                if (!(p > 0.0) || !isfinite(p)) bad = true;
                p_own = ln == kk ? p : p_own;
                double pinv = __builtin_amdgcn_rcp(p);
                pinv = pinv * (2.0 - p * pinv);
                pinv = pinv * (2.0 - p * pinv);
                // row kk becomes the scaled pivot row (pivot -> 1 / p), every other row i: a_iq -= a_ik r_q / p (column kk -> - a_ik / p):
                // one form for all lanes, new = keep * old + coef * (r_q / p), with keep = 0, coef = 1 on the pivot's own lane
                const bool own = ln == kk;
                const double keep = own ? 0.0 : 1.0, coef = own ? 1.0 : -mi[kk];
#pragma unroll
                for (int q = 0; q < 6; ++q) mi[q] = q == kk ? coef * pinv : keep * mi[q] + coef * (r[q] * pinv);
            }
            if (ln < 6) {
#pragma unroll
                for (int q = 0; q < 6; ++q) D[ln * 6 + q] = mi[q];
            }
            if (bad && ln == 0) failw[0] = 1;
            if (__any(ln < 6 && p_own < dref) && ln == 0) failw[1] = 1;
        } else if (k > 0) {
            // the other waves meanwhile store the previous step's panel as the factor's blocks L(i, k - 1) (nobody reads column
            // k - 1 of the band any more until the backward sweep; the panel is rewritten behind the barrier below)
            const int mp = min(bw, nf - k);
            for (int idx = tid - 64; idx < mp * 36; idx += kBT - 64) {
                const int irel = idx / 36, q = idx - irel * 36;
                Lb[band_off(k + irel, k - 1, bw) + q] = T[idx];
            }
        }
        __syncthreads();
        BAND_STAMP(1);
        // panel: T_i = A_ik Dk^-1 (row b of the inverse stands for its column b: the block is symmetric)
        if (bw * 36 <= kBT) {
            if (p_irel < m) {
                const double2 *A = reinterpret_cast<const double2 *>(Lk + p_A);
                const double2 *Dr = reinterpret_cast<const double2 *>(Lk + p_D);
                const double *Dc = Lk + p_Dc;
                const double2 a0 = A[0], a1 = A[1], a2 = A[2], r0 = Dr[0], r1 = Dr[1], r2 = Dr[2];
                // (the elimination leaves Dk^-1 symmetric but for rounding; everything downstream uses ITS SYMMETRIC PART, row b and
                //  column b averaged where they are read: taken as it came, row b for column b here and rows in the sweep, a chain of 40
                //  keyframes held by tracks of three ended 4e-8 m from the oracle's poses instead of 2e-11)
                const double2 d0 = make_double2(0.5 * (r0.x + Dc[0]), 0.5 * (r0.y + Dc[6])), d1 = make_double2(0.5 * (r1.x + Dc[12]), 0.5 * (r1.y + Dc[18])),
                              d2 = make_double2(0.5 * (r2.x + Dc[24]), 0.5 * (r2.y + Dc[30]));
                T[tid] = ((a0.x * d0.x + a0.y * d0.y) + (a1.x * d1.x + a1.y * d1.y)) + (a2.x * d2.x + a2.y * d2.y);
            }
        } else for (int idx = tid; idx < m * 36; idx += kBT) {
            const int irel = idx / 36, q = idx - irel * 36, a = q / 6, b = q - a * 6;
            const double2 *A = reinterpret_cast<const double2 *>(Lb + band_off(k + 1 + irel, k, bw) + a * 6);
            const double2 *Dr = reinterpret_cast<const double2 *>(D + b * 6);
            const double *Dc = D + b;
            const double2 a0 = A[0], a1 = A[1], a2 = A[2], r0 = Dr[0], r1 = Dr[1], r2 = Dr[2];
            const double2 d0 = make_double2(0.5 * (r0.x + Dc[0]), 0.5 * (r0.y + Dc[6])), d1 = make_double2(0.5 * (r1.x + Dc[12]), 0.5 * (r1.y + Dc[18])),
                          d2 = make_double2(0.5 * (r2.x + Dc[24]), 0.5 * (r2.y + Dc[30]));
            T[idx] = ((a0.x * d0.x + a0.y * d0.y) + (a1.x * d1.x + a1.y * d1.y)) + (a2.x * d2.x + a2.y * d2.y);
        }
        __syncthreads();
        BAND_STAMP(2);
        // trailing blocks A_ij -= T_i A_jk^T (i >= j below k in the band) and right-hand side b_i -= T_i b_k
        const int ntr = m * (m + 1) / 2 * 36;
        if (fast_tr) {
            double2 t0[kTr], t1[kTr], t2[kTr], u0[kTr], u1[kTr], u2[kTr];
            double old[kTr];
            double *dst[kTr];
#pragma unroll
            for (int u = 0; u < kTr; ++u) {
                const bool act = t_irel[u] < m;
                const double2 *tp = reinterpret_cast<const double2 *>(T + (act ? t_T[u] : 0));
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
                    Ti = T + irel * 36 + a * 6;
                    Aj = Lb + band_off(k + 1 + jrel, k, bw) + b * 6;
                    dst[u] = Lb + band_off(k + 1 + irel, k + 1 + jrel, bw) + q;
                } else {
                    const int r = idx - ntr, irel = r / 6, a = r - irel * 6;
                    Ti = T + irel * 36 + a * 6;
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
        BAND_STAMP(3);
    }
    const bool fail = failw[0] != 0;
    if (failw[1] != 0 && !fail) {
        // ill conditioned beyond what explicit pivot-block inverses carry: park the solve (Ctrl::done = 2 turns every kernel queued
        // behind into a no-op) and tell the host, which queues the dense direct solver for this trial and every later one
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
    // (the last step has no panel: nothing left to store)
    // ---- z_k = Dk^-1 y_k, then the backward sweep by one wave: x_k final, z_j -= L_kj^T x_k for the band above ----
    for (int r = tid; r < n; r += kBT) {
        const int k = r / 6, a = r - k * 6;
        const double *D = Lb + band_off(k, k, bw) + a * 6, *Dc = Lb + band_off(k, k, bw) + a;
        const double *y = rhs + 6 * k;
        double s = (0.5 * (D[0] + Dc[0])) * y[0];
#pragma unroll
        for (int cc = 1; cc < 6; ++cc) s += (0.5 * (D[cc] + Dc[6 * cc])) * y[cc];
        aux[r] = s;
    }
    __syncthreads();
    if (wv == 0) {
        for (int k = nf - 1; k > 0; --k) {
            const int mk = min(bw, k);
            const double *xk = aux + 6 * k;
            for (int l = ln; l < mk * 6; l += 64) {
                const int jrel = l / 6, a = l - jrel * 6, j = k - 1 - jrel;
                const double *L = Lb + band_off(k, j, bw) + a;
                double s = L[0] * xk[0];
#pragma unroll
                for (int cc = 1; cc < 6; ++cc) s += L[cc * 6] * xk[cc];
                aux[6 * j + a] -= s;
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
        if (h >= 0) {
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
    hipLaunchKernelGGL(k_band, dim3(1), dim3(kBT), band_lds_bytes(w.nfree, bw), s, w, bw);
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
