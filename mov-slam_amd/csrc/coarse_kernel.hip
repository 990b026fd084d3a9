// k_coarse: coarse level of the two-level PCG preconditioner, built OFF the critical path.
//
// For LM trial t it assembles the reduced matrix S from the schur work-item partials, forms
// A_c = P^T S P over the keyframe aggregates (aggregate = the block rows one wave of k_pcg_rows
// owns, 6 coarse dofs each, 48 x 48), inverts it by Gauss-Jordan in LDS and leaves A_c^-1 in HBM.
// It runs on a side stream concurrently with k_pcg_rows(t); its result preconditions trial t+1
// (a preconditioner need not be exact: a one-trial-old coarse inverse costs ~3 % more CG iterations
// than a fresh one, and block-Jacobi alone ~2.3x more).  One workgroup; everything in fixed order,
// so the lagged preconditioner is as reproducible as the rest of the solve.
#include <hip/hip_runtime.h>

#include "device_math.h"
#include "device_types.h"
#include "kernels.h"

namespace movba {

namespace {
constexpr int kT = 512;
constexpr int kNW = kT / 64;
constexpr int kNC = 6 * (kPcgRowsThreads / 64);
}  // namespace

__global__ __launch_bounds__(kT) void k_coarse(DevWindow w, PcgParams pp, int trial)
{
    __shared__ __attribute__((aligned(16))) double Ac[kNC * kNC];
    __shared__ __attribute__((aligned(16))) double gj[4 * kNC];
    __shared__ int s_bad;
    const Ctrl *c = w.ctrl;
    if (c->done) return;
    const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
    const int nf = w.nfree;
    const double lambda = w.lam_snap[trial & 1];
    const double *part = w.part + (size_t)(trial & 1) * w.part_stride;
    double *blocks = w.blocks_c;
    if (tid == 0) s_bad = 0;
    for (int idx = tid; idx < kNC * kNC; idx += kT) Ac[idx] = 0.0;

    // ---- S blocks (upper triangle) from the partials, item order ----
    for (int idx = tid; idx < w.npairs * 36; idx += kT) {
        const int pr = idx / 36, k = idx - pr * 36;
        const int i0 = w.pair_item_start[pr], i1 = w.pair_item_start[pr + 1];
        double s = 0.0;
        for (int itx = i0; itx < i1; ++itx) s += part[(size_t)itx * kPartStride + k];
        double v = -s;
        if (pr < nf) {
            const int a = k / 6, b = k - a * 6;
            const int u = a <= b ? ut6(a, b) : ut6(b, a);
            double hpp = 0.0;
            for (int itx = i0; itx < i1; ++itx) hpp += part[(size_t)itx * kPartStride + 42 + u];
            v += hpp + (a == b ? lambda : 0.0);
        }
        blocks[idx] = v;
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

hipError_t launch_coarse(const DevWindow &w, const PcgParams &pp, int trial, hipStream_t s)
{
    hipLaunchKernelGGL(k_coarse, dim3(1), dim3(kT), 0, s, w, pp, trial);
    return hipGetLastError();
}

}  // namespace movba
