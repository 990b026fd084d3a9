// Coarse level of the two-level PCG preconditioner, built OFF the critical path: the second workgroup of the
// k_pcg_rows launch runs coarse_build() while the first one runs the conjugate gradients.
//
// For LM trial t it assembles the reduced matrix S from the schur work-item partials, forms
// A_c = P^T S P over the keyframe aggregates (aggregate = the block rows one wave of k_pcg_rows
// owns; 12 coarse dofs each: the 6 pose components constant over the aggregate and the same 6 varying
// linearly with the keyframe index; 96 x 96), inverts it by Gauss-Jordan in LDS and leaves A_c^-1 in HBM.
// The linear modes matter: late LM iterations (small lambda) are dominated by smooth bending of the
// trajectory between the fixed keyframes, which piecewise-constant modes resolve poorly: cfg3 needs
// 397 CG iterations per window solve with the 6 constant modes alone and 247 with the 12.
// Its result preconditions trial t+1.  A preconditioner need not be exact, but a one-trial-old coarse matrix
// costs ~30 % more CG iterations than a fresh one (lambda drops to a third per good step and the weakly
// constrained modes follow it), so what is inverted is a PREDICTION of trial t+1's matrix from this build and
// the previous one (below): 215 iterations, against ~185 with fresh matrices and ~870 with block-Jacobi alone.
// Everything in fixed order, so the lagged preconditioner is as reproducible as the rest of the solve.
// (A separate kernel on a side stream did the same job at first: the two cross-stream event waits per
// trial cost ~10 us of idle time on the LM chain, rocprofv3 kernel trace.)
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "device_math.h"
#include "device_types.h"
#include "handoff.h"

namespace movba {

#ifndef MOVBA_COARSE_EXTRAPOLATE
#define MOVBA_COARSE_EXTRAPOLATE 1
#endif

// sm: >= kNC*kNC + 9*kNC + 8 doubles + 2 ints per coarse term (= gather-list entry) of LDS; 512 threads
template <int kT, int kNC, int kPA>
__device__ __forceinline__ void coarse_build(const DevWindow &w, const PcgParams &pp, int trial, double lambda, double *sm, bool extrapolate)
{
    double *Ac = sm;
    double *gj = Ac + kNC * kNC;
    int &s_bad = *reinterpret_cast<int *>(gj + 9 * kNC + 2);     // gj: 2 x 4 kNC snapshots + 2 pivots + kNC scaling, then the flag
    const int tid = threadIdx.x;
    const int nf = w.nfree;
#ifdef MOVBA_CLOCK_STAMP
    unsigned long long cst_last = __builtin_amdgcn_s_memtime();
#define COARSE_STAMP(k) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); if (tid == 0) w.ctrl->dbg_seg2[k] += _t - cst_last; cst_last = _t; } while (0)
#else
#define COARSE_STAMP(k) do { } while (0)
#endif
    // (the schur partials: plain loads; on the two-stream path the caller's waves have made their agent-scope acquire behind
    //  the pass's flags)
    const double *part = w.part, *part_ = part;
    double *blocks = w.blocks_c;
    if (tid == 0) s_bad = 0;
    for (int idx = tid; idx < kNC * kNC; idx += kT) Ac[idx] = 0.0;

    // ---- S blocks (upper triangle) from the partials, item order.  Off-diagonal pairs (one work item almost always):
    // 8 elements per thread in flight; diagonal pairs (cut into several finer items): 8 items of one element in flight ----
    const int ndiag = nf * 36;
    // off-diagonal pairs cut into several work items (more than kSchurChunk shared points: rare) are materialised too;
    // single-item pairs are read straight from their partial by the A_c pass
    for (int idx = tid; idx < w.n_multi * 36; idx += kT) {
        const int m = idx / 36, k = idx - m * 36, pr = w.multi_pairs[m];
        double sacc = 0.0;
        for (int itx = w.pair_item_start[pr]; itx < w.pair_item_start[pr + 1]; ++itx) sacc += part[(size_t)itx * kPartStride + k];
        blocks[pr * 36 + k] = -sacc;
    }
    for (int base = tid; base < ndiag; base += 4 * kT) {
        // four diagonal elements per thread at a time, up to 6 work items of each in flight
        int k[4], hu[4], i0[4], i1[4];
        double s[4] = { 0, 0, 0, 0 }, hp[4] = { 0, 0, 0, 0 };
        int nmax = 0;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int idx = min(base + m * kT, ndiag - 1);
            const int pr = idx / 36;
            k[m] = idx - pr * 36;
            const int a = k[m] / 6, b = k[m] - a * 6;
            hu[m] = 42 + (a <= b ? ut6(a, b) : ut6(b, a));
            i0[m] = w.pair_item_start[pr]; i1[m] = w.pair_item_start[pr + 1];
            nmax = max(nmax, i1[m] - i0[m]);
        }
        for (int o = 0; o < nmax; o += 6) {
            double sv[4][6], hv[4][6];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    const double *src = part + (size_t)min(i0[m] + o + u, i1[m] - 1) * kPartStride;
                    sv[m][u] = src[k[m]]; hv[m][u] = src[hu[m]];
                }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int u = 0; u < 6; ++u) { const bool in = i0[m] + o + u < i1[m]; s[m] += in ? sv[m][u] : 0.0; hp[m] += in ? hv[m][u] : 0.0; }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int idx = base + m * kT;
            if (idx < ndiag) blocks[idx] = (hp[m] + ((k[m] % 7 == 0) ? lambda : 0.0)) - s[m];     // k = 7 a: diagonal element
        }
    }
    __syncthreads();
    COARSE_STAMP(2);
    // ---- A_c = P^T S P.  Coarse dof (g, d, a): aggregate g, mode d (0: constant, 1: linear in the keyframe index,
    // phi_1(i) = (i - c_g) / h_g), pose component a.  One thread per (coarse block, a, b): it walks the block's fine terms
    // once, in list order, and accumulates the four mode combinations phi_d(i) phi_e(j) together. ----
    // centre and inverse half-width of every aggregate (the linear mode's phi), once, in LDS
    double *aggc = gj;                                      // 2 x (kNC / kPA) doubles; gj is not in use yet
    if (tid < kNC / kPA) {
        const int g0 = pp.wave_row0[tid], g1 = pp.wave_row0[tid + 1];
        aggc[tid] = g0 + 0.5 * (g1 - g0 - 1); aggc[kNC / kPA + tid] = 1.0 / fmax(1.0, 0.5 * (g1 - g0));
    }
    // the term lists themselves go to LDS first (one coalesced pass): the gathers below then depend on ONE global
    // round trip per batch instead of two
    int *tent = reinterpret_cast<int *>(gj + 9 * kNC + 8), *tij = tent + w.cblk_ptr[w.n_cblk];
    for (int q = tid; q < w.cblk_ptr[w.n_cblk]; q += kT) { tent[q] = w.cblk_ent[q]; tij[q] = w.cblk_ij[q]; }
    __syncthreads();
    // Thread (cb, a) owns row a of coarse block cb = (g, h) in all four mode combinations: it walks the block's term
    // list once (8 terms = 48 gathers in flight), so no thread pads its list to a longer neighbour's.  When there are
    // threads to spare (n_cblk * 12 <= 512: always with 8 aggregates) two threads share a row: the first takes the front
    // of the list (whole batches of 8), the second the rest, and the second's sums are added after the first's are stored.
    const int nw = w.n_cblk * 6;
    const bool split = 2 * nw <= kT;
    auto walk = [&](int cb, int a, int part, double (&s00)[6], double (&s01)[6], double (&s10)[6], double (&s11)[6]) {
        const int g = w.cblk_g[cb], h = w.cblk_h[cb];
        const double cg = aggc[g], ch = aggc[h], ig = aggc[kNC / kPA + g], ih = aggc[kNC / kPA + h];
        int t0 = w.cblk_ptr[cb], t1 = w.cblk_ptr[cb + 1];
        if (split) {
            const int n = t1 - t0, front = min(n, (((n + 1) >> 1) + 7) & ~7);
            if (part == 0) t1 = t0 + front; else t0 += front;
        }
#pragma unroll
        for (int m = 0; m < 6; ++m) s00[m] = s01[m] = s10[m] = s11[m] = 0.0;
        for (int t = t0; t < t1; t += 8) {
            int pk[8], ij[8];
            double v[8][6];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int tt = min(t + u, t1 - 1); pk[u] = tent[tt]; ij[u] = tij[tt]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = pk[u];
                const double *src = (e & 1) ? part_ + (size_t)(e >> 2) * kPartStride : blocks + (size_t)(e >> 2) * 36;
                // row a of the (possibly transposed) fine block
                const int o0 = (e & 2) ? a : a * 6, st = (e & 2) ? 6 : 1;
#pragma unroll
                for (int m = 0; m < 6; ++m) v[u][m] = src[o0 + st * m];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool in = t + u < t1;
                const double sgn = in ? ((pk[u] & 1) ? -1.0 : 1.0) : 0.0;
                const double pi = ((ij[u] >> 16) - cg) * ig, pj = ((ij[u] & 0xffff) - ch) * ih;
#pragma unroll
                for (int m = 0; m < 6; ++m) {
                    const double x = sgn * v[u][m];
                    s00[m] += x; s01[m] += pj * x; s10[m] += pi * x; s11[m] += pi * pj * x;
                }
            }
        }
        return Ac + (g * kPA + a) * kNC + h * kPA;
    };
    if (split) {
        const bool active = tid < 2 * nw;
        const int half = tid >= nw, wi = tid - half * nw;
        double s00[6], s01[6], s10[6], s11[6];
        double *dst = nullptr;
        if (active) dst = walk(wi / 6, wi - (wi / 6) * 6, half, s00, s01, s10, s11);
        if (active && !half) {
#pragma unroll
            for (int m = 0; m < 6; ++m) { dst[m] = s00[m]; dst[6 + m] = s01[m]; dst[6 * kNC + m] = s10[m]; dst[6 * kNC + 6 + m] = s11[m]; }
        }
        __syncthreads();
        if (active && half) {
#pragma unroll
            for (int m = 0; m < 6; ++m) { dst[m] += s00[m]; dst[6 + m] += s01[m]; dst[6 * kNC + m] += s10[m]; dst[6 * kNC + 6 + m] += s11[m]; }
        }
    } else {
        for (int wi = tid; wi < nw; wi += kT) {
            double s00[6], s01[6], s10[6], s11[6];
            double *dst = walk(wi / 6, wi - (wi / 6) * 6, 0, s00, s01, s10, s11);
#pragma unroll
            for (int m = 0; m < 6; ++m) { dst[m] = s00[m]; dst[6 + m] = s01[m]; dst[6 * kNC + m] = s10[m]; dst[6 * kNC + 6 + m] = s11[m]; }
        }
    }
    __syncthreads();
    // dofs without support (an aggregate with no rows; the linear modes of an aggregate with a single row): identity
    // rows keep A_c invertible, their restricted residual is always zero
    if (tid < kNC) {
        const int g = tid / kPA, d = (tid - g * kPA) / 6, nrows = pp.wave_row0[g + 1] - pp.wave_row0[g];
        if (nrows == 0 || (d == 1 && nrows < 2)) {
            for (int j = 0; j < kNC; ++j) { Ac[tid * kNC + j] = 0.0; Ac[j * kNC + tid] = 0.0; }
        }
    }
    __syncthreads();
    if (tid < kNC) {
        const int g = tid / kPA, d = (tid - g * kPA) / 6, nrows = pp.wave_row0[g + 1] - pp.wave_row0[g];
        if (nrows == 0 || (d == 1 && nrows < 2)) Ac[tid * kNC + tid] = 1.0;
    }
    __syncthreads();
    // ---- the inverse is for the NEXT trial, whose matrix is not this one: lambda will most likely be a third of today's
    // (OptimizationAlgorithmLevenberg's factor after a good step), and the weakly constrained modes of S scale with
    // lambda.  Model A_c(lambda) as affine through this build and the previous one and invert the prediction
    // A + gamma (A - A_prev), gamma = (lambda/3 - lambda) / (lambda - lambda_prev) clamped to [-1, 1/3]  (gamma < 0: a convex
    // combination; gamma = 1/3 stays positive definite while A > A_prev / 4, and a lost pivot only costs the coarse level
    // for one trial).  cfg3: 247 -> ~215 CG iterations per window solve.  Not in fresh mode (the matrix is this trial's). ----
    if (extrapolate) {
        double *prev = w.ac_prev;
        const double lam_prev = prev[kNC * kNC], trial_prev = prev[kNC * kNC + 1];
        double gamma = 0.0;
        if (trial >= 1 && trial_prev == (double)(trial - 1) && lambda != lam_prev && MOVBA_COARSE_EXTRAPOLATE) {
            gamma = (lambda * (1.0 / 3.0) - lambda) / (lambda - lam_prev);
            gamma = fmin(fmax(gamma, -1.0), 1.0 / 3.0);
            if (!isfinite(gamma)) gamma = 0.0;
        }
        constexpr int kPer = kNC * kNC / kT;
        static_assert(kPer * kT == kNC * kNC, "one pass");
        double pv[kPer];
#pragma unroll
        for (int u = 0; u < kPer; ++u) pv[u] = prev[tid + u * kT];
        __syncthreads();        // lambda / trial of the previous build have been read by everyone
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const double a = Ac[tid + u * kT];
            prev[tid + u * kT] = a;
            if (gamma != 0.0) Ac[tid + u * kT] = a + gamma * (a - pv[u]);    // (the first build of a solve finds stale memory in prev)
        }
        if (tid == 0) { prev[kNC * kNC] = lambda; prev[kNC * kNC + 1] = (double)trial; }
        __syncthreads();
    }

    COARSE_STAMP(3);
    // ---- Gauss-Jordan inverse of the symmetrically scaled matrix D A_c D (unit diagonal), the matrix held in REGISTERS:
    // thread (rg, cg) owns rows 3 rg .. 3 rg + 2 x columns 6 cg .. 6 cg + 5 (32 x 16 threads x 18 elements = 96 x 96).
    // TWO pivots per workgroup barrier (round 4; one until then: 975 cycles per pivot, 39 us per build, most of it the
    // barrier, the LDS round trip of the snapshot and its publication).  For the pair (k, k + 1) the snapshot holds, from
    // the state BEFORE pivot k: row k and column k published with r_k = p + 1 and c_k = p - 1 - which turns ONE formula,
    // a_ij -= c_i r_j / p, into the whole in-place step (pivot row -> r_j / p, pivot column -> -c_i / p, pivot -> 1 / p;
    // accurate because the scaled pivots are <= 1) - and row k + 1 and column k + 1 as they stand.  Every thread brings
    // those two forward over pivot k itself (the same formula applied to them: r'_j = r_j - c_(k+1) r^k_j / p, ...), which
    // also yields the second pivot, and applies both steps to its tile.  The groups of 6 pivots are unrolled so that the
    // tile rows / columns holding the next pair are compile-time indices.
    static_assert(kNC == 96 && kT == 512, "tile layout written for 96 x 96 on 512 threads");
    double *dsc = gj + 8 * kNC + 2;                         // kNC: 1 / sqrt(diagonal)
    double *pvs = gj + 8 * kNC;                             // first pivot of the pair, per snapshot buffer
    if (tid < kNC) { const double d = Ac[tid * kNC + tid]; dsc[tid] = (d > 0.0 && isfinite(d)) ? rsqrt(d) : 0.0; if (!(d > 0.0) || !isfinite(d)) s_bad = 1; }
    __syncthreads();
    const int rg = tid >> 4, cg = tid & 15;
    double t[3][6];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 6; ++q) t[r][q] = Ac[(3 * rg + r) * kNC + 6 * cg + q] * dsc[3 * rg + r] * dsc[6 * cg + q];
    // Groups of 6 pivots (aggregate g, mode d) without support are identity rows and columns: eliminating them changes
    // nothing, so they are skipped (small windows: fewer keyframes than 2 per wave have no linear modes at all).
    unsigned live = 0;
    for (int q = 0; q < kNC / 6; ++q) {
        const int nrows = pp.wave_row0[(q >> 1) + 1] - pp.wave_row0[q >> 1];
        if (!(nrows == 0 || ((q & 1) && nrows < 2))) live |= 1u << q;
    }
    __syncthreads();                                        // (aggc, which shares the snapshot area, has been read by everyone)
    // publishes the snapshot of the pair (6 gq + C, 6 gq + C + 1) from the tiles as they stand
    auto publish = [&](auto Ctag, int gq, int buf) {
        constexpr int C = decltype(Ctag)::value;
        constexpr int lr0 = C % 3, lr1 = (C + 1) % 3;      // tile rows of the two pivots (6 gq is a multiple of 3)
        const int rg0 = 2 * gq + C / 3, rg1 = 2 * gq + (C + 1) / 3;
        double *R1 = gj + buf * 4 * kNC, *C1 = R1 + kNC, *R2 = C1 + kNC, *C2 = R2 + kNC;
        if (rg == rg0) {
#pragma unroll
            for (int q = 0; q < 6; ++q) R1[6 * cg + q] = t[lr0][q] + ((cg == gq && q == C) ? 1.0 : 0.0);
        }
        if (rg == rg1) {
#pragma unroll
            for (int q = 0; q < 6; ++q) R2[6 * cg + q] = t[lr1][q];
        }
        if (cg == gq) {
#pragma unroll
            for (int r = 0; r < 3; ++r) { C1[3 * rg + r] = t[r][C] - ((rg == rg0 && r == lr0) ? 1.0 : 0.0); C2[3 * rg + r] = t[r][C + 1]; }
        }
        if (rg == rg0 && cg == gq) pvs[buf] = t[lr0][C];
    };
    int kk = live ? __builtin_ctz(live) : kNC / 6;
    if (kk < kNC / 6) publish(std::integral_constant<int, 0>{}, kk, 0);
    __syncthreads();
    bool bad = s_bad != 0;
    int buf = 0;
    // one pair: both pivots applied to the tile; false when a pivot is not positive (the coarse level is then unusable for one trial)
    auto pair_step = [&](auto Ctag, int gq) -> bool {
        constexpr int C = decltype(Ctag)::value;
        constexpr int lr1 = (C + 1) % 3;
        const int k = 6 * gq + C, rg1 = 2 * gq + (C + 1) / 3;
        const double *R1 = gj + buf * 4 * kNC, *C1 = R1 + kNC, *R2 = C1 + kNC, *C2 = R2 + kNC;
        const double p1 = pvs[buf];
        const double o12 = R1[k + 1], o21 = C1[k + 1], q22 = R2[k + 1];
        double r1[6], r2[6], c1[3], c2[3];
#pragma unroll
        for (int q = 0; q < 6; ++q) { r1[q] = R1[6 * cg + q]; r2[q] = R2[6 * cg + q]; }
#pragma unroll
        for (int r = 0; r < 3; ++r) { c1[r] = C1[3 * rg + r]; c2[r] = C2[3 * rg + r]; }
        if (!(p1 > 0.0) || !isfinite(p1)) return false;
        // reciprocals by v_rcp_f64 and two Newton steps (the scaled pivots are in (0, 1]; this is a preconditioner)
        double ip1 = __builtin_amdgcn_rcp(p1);
        ip1 = ip1 * (2.0 - p1 * ip1);
        ip1 = ip1 * (2.0 - p1 * ip1);
        const double s12 = o12 * ip1;
        const double p2 = q22 - o21 * s12;
        if (!(p2 > 0.0) || !isfinite(p2)) return false;
        double ip2 = __builtin_amdgcn_rcp(p2);
        ip2 = ip2 * (2.0 - p2 * ip2);
        ip2 = ip2 * (2.0 - p2 * ip2);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            r1[q] *= ip1;
            r2[q] = ((r2[q] - o21 * r1[q]) + ((cg == gq && q == C + 1) ? 1.0 : 0.0)) * ip2;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) c2[r] = (c2[r] - c1[r] * s12) - ((rg == rg1 && r == lr1) ? 1.0 : 0.0);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 6; ++q) { t[r][q] -= c1[r] * r1[q]; t[r][q] -= c2[r] * r2[q]; }
        return true;
    };
    while (kk < kNC / 6 && !bad) {
        const unsigned later = kk < 31 ? (live & ~((2u << kk) - 1u)) : 0u;
        const int next = later ? __builtin_ctz(later) : kNC / 6;
        if (!pair_step(std::integral_constant<int, 0>{}, kk)) { bad = true; break; }
        publish(std::integral_constant<int, 2>{}, kk, buf ^ 1);
        __syncthreads(); buf ^= 1;
        if (!pair_step(std::integral_constant<int, 2>{}, kk)) { bad = true; break; }
        publish(std::integral_constant<int, 4>{}, kk, buf ^ 1);
        __syncthreads(); buf ^= 1;
        if (!pair_step(std::integral_constant<int, 4>{}, kk)) { bad = true; break; }
        if (next < kNC / 6) publish(std::integral_constant<int, 0>{}, next, buf ^ 1);
        __syncthreads(); buf ^= 1;
        kk = next;
    }
    if (bad && tid == 0) s_bad = 1;
    __syncthreads();
    COARSE_STAMP(4);
    // ---- publish: A_c^-1 = D (D A_c D)^-1 D for trial+1 and its validity tag ----
    float *dst = w.aci + (size_t)(trial & 1) * kNC * kNC;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 6; ++q) dst[(3 * rg + r) * kNC + 6 * cg + q] = (float)(t[r][q] * dsc[3 * rg + r] * dsc[6 * cg + q]);
    __syncthreads();
    if (tid == 0) w.aci_tag[trial & 1] = s_bad ? -1 : trial;
    COARSE_STAMP(5);
}

}  // namespace movba
