// Coarse level of the two-level PCG preconditioner, built OFF the critical path: the second workgroup of the
// k_pcg_rows launch runs coarse_build() while the first one runs the conjugate gradients.
//
// For LM trial t it assembles the reduced matrix S from the schur work-item partials, forms
// A_c = P^T S P over the keyframe aggregates (aggregate = the block rows one wave of k_pcg_rows
// owns, 6 coarse dofs each, 48 x 48), inverts it by Gauss-Jordan in LDS and leaves A_c^-1 in HBM.
// Its result preconditions trial t+1 (a preconditioner need not be exact: a one-trial-old coarse
// inverse costs ~3 % more CG iterations than a fresh one, and block-Jacobi alone ~2.3x more).
// Everything in fixed order, so the lagged preconditioner is as reproducible as the rest of the solve.
// (A separate kernel on a side stream did the same job at first: the two cross-stream event waits per
// trial cost ~10 us of idle time on the LM chain, rocprofv3 kernel trace.)
#pragma once
#include <hip/hip_runtime.h>

#include "device_math.h"
#include "device_types.h"

namespace movba {

// sm: >= kNC*kNC + 4*kNC + 2 doubles of LDS; 512 threads
template <int kT, int kNC>
__device__ __forceinline__ void coarse_build(const DevWindow &w, const PcgParams &pp, int trial, double lambda, double *sm)
{
    constexpr int kNW = kT / 64;
    double *Ac = sm;
    double *gj = Ac + kNC * kNC;
    int &s_bad = *reinterpret_cast<int *>(gj + 4 * kNC);
    const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
    const int nf = w.nfree;
    const double *part = w.part;
    double *blocks = w.blocks_c;
    if (tid == 0) s_bad = 0;
    for (int idx = tid; idx < kNC * kNC; idx += kT) Ac[idx] = 0.0;

    // ---- S blocks (upper triangle) from the partials, item order; 4 elements per thread in flight ----
    const int total = w.npairs * 36;
    for (int base = tid; base < total; base += 4 * kT) {
        int k[4], i0[4], i1[4], pr[4];
        double s[4], hp[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = min(base + u * kT, total - 1);
            pr[u] = idx / 36; k[u] = idx - pr[u] * 36;
            i0[u] = w.pair_item_start[pr[u]]; i1[u] = w.pair_item_start[pr[u] + 1];
        }
        int hu[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int a = k[u] / 6, b = k[u] - a * 6;
            hu[u] = 42 + (a <= b ? ut6(a, b) : ut6(b, a));
            const bool any = i1[u] > i0[u];
            const double *src = part + (size_t)(any ? i0[u] : 0) * kPartStride;
            const double v0 = src[k[u]], v1 = src[hu[u]];
            s[u] = any ? v0 : 0.0; hp[u] = (any && pr[u] < nf) ? v1 : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            for (int itx = i0[u] + 1; itx < i1[u]; ++itx) {
                const double *src = part + (size_t)itx * kPartStride;
                s[u] += src[k[u]];
                if (pr[u] < nf) hp[u] += src[hu[u]];
            }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * kT;
            if (idx < total) {
                const int a = k[u] / 6, b = k[u] - a * 6;
                blocks[idx] = (pr[u] < nf) ? (hp[u] + (a == b ? lambda : 0.0)) - s[u] : -s[u];
            }
        }
    }
    __syncthreads();
    // ---- A_c = P^T S P: every coarse element is the fixed-order sum of its fine-block terms ----
    for (int idx = tid; idx < w.n_cblk * 36; idx += kT) {
        const int cb = idx / 36, k = idx - cb * 36, a = k / 6, b = k - a * 6;
        const int t0 = w.cblk_ptr[cb], t1 = w.cblk_ptr[cb + 1];
        double sum = 0.0;
        for (int t = t0; t < t1; t += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int pk = w.cblk_ent[min(t + u, t1 - 1)];
                v[u] = blocks[(size_t)(pk >> 1) * 36 + ((pk & 1) ? b * 6 + a : a * 6 + b)];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) sum += (t + u < t1) ? v[u] : 0.0;
        }
        Ac[(w.cblk_g[cb] * 6 + a) * kNC + w.cblk_h[cb] * 6 + b] = sum;
    }
    // aggregates without rows (fewer block rows than waves): identity rows keep A_c invertible
    if (tid < kNC && pp.wave_row0[tid / 6 + 1] == pp.wave_row0[tid / 6]) Ac[tid * kNC + tid] = 1.0;
    __syncthreads();

    // ---- in-place Gauss-Jordan inverse (SPD, no pivoting), one barrier per pivot ----
    for (int j = tid; j < kNC; j += kT) { gj[j] = Ac[j]; gj[kNC + j] = Ac[j * kNC]; }
    __syncthreads();
    bool bad = false;
    for (int k = 0; k < kNC; ++k) {
        const double *rk = gj + (k & 1) * 2 * kNC, *ck = rk + kNC;
        double *rn = gj + ((k + 1) & 1) * 2 * kNC, *cn = rn + kNC;
        const double piv = rk[k];
        if (!(piv > 0.0) || !isfinite(piv)) { bad = true; break; }
        const double pinv = 1.0 / piv;
        if (ln < kNC) {
            const int j = ln;
            const double rj = rk[j];
            double av[kNC / kNW], cv[kNC / kNW];
#pragma unroll
            for (int m = 0; m < kNC / kNW; ++m) { av[m] = Ac[(wv + kNW * m) * kNC + j]; cv[m] = ck[wv + kNW * m]; }
#pragma unroll
            for (int m = 0; m < kNC / kNW; ++m) {
                const int i = wv + kNW * m;
                double v = av[m] - cv[m] * rj * pinv;
                v = (j == k) ? -cv[m] * pinv : v;
                v = (i == k) ? ((j == k) ? pinv : rj * pinv) : v;
                Ac[i * kNC + j] = v;
                if (i == k + 1) rn[j] = v;
                if (j == k + 1) cn[i] = v;
            }
        }
        __syncthreads();
    }
    if (bad && tid == 0) s_bad = 1;
    __syncthreads();
    // ---- publish: A_c^-1 for trial+1 and its validity tag ----
    double *dst = w.aci + (size_t)(trial & 1) * kNC * kNC;
    for (int idx = tid; idx < kNC * kNC; idx += kT) dst[idx] = Ac[idx];
    __syncthreads();
    if (tid == 0) w.aci_tag[trial & 1] = s_bad ? -1 : trial;
}


}  // namespace movba
