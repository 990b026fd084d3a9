// Direct solve of the reduced camera system: dense blocked Cholesky over several launches.
//
// The reference's step is an exact sparse Cholesky (LinearSolverCSparse, selected at
// /root/reference/src/Optimizer.cc:535; a failed factorisation rejects the LM trial).  The product's first choice for
// windows of up to 80 free keyframes is the on-chip PCG (pcg_kernel.hip); this file is the exact path behind it:
//   * every window with more free keyframes than k_pcg_rows holds on chip (any number: S lives in HBM / L2 here),
//   * every trial from the first one whose PCG broke down or ran into its iteration cap (weakly constrained windows,
//     where an iterative solve at any tolerance departs from the exact step).
// S = Hpp + lambda I - sum B Dinv B^T is laid out densely in 48 x 48 tiles (8 pose blocks; lower block triangle), with the
// right-hand side as one more block row, so that forward substitution is part of the factorisation (Cholesky of
// [[S, b], [b^T, .]] leaves L^-1 b in the last row).  One launch per block column j (right-looking with a one-column
// look-ahead by redundancy):
//     step(j):  tile (I, K), K > j   <- tile (I, K) - L(I, j-1) L(K, j-1)^T        (trailing update with the previous column)
//               tile (I, j)          <- the same update, then  L(I, j) = tile (I, j) L(j, j)^-T, where EVERY workgroup of
//                                       column j updates and factors the diagonal tile (j, j) for itself: no workgroup
//                                       waits for another inside a launch, and column j is final when the launch ends.
// The 48 x 48 x 48 tile products run on the fp64 matrix cores (v_mfma_f64_16x16x4_f64: the one GEMM-shaped piece of the
// path); the 48 sequential pivots of a column are one workgroup barrier each, with the panel in registers.
// Then one workgroup solves L^T x = y backwards and applies the pose increments (VertexSE3Expmap::oplusImpl).
// Fixed summation order everywhere: results are bit-reproducible run to run.
#include <hip/hip_runtime.h>

#include "device_math.h"
#include "device_types.h"
#include "kernels.h"
#include "dense_tile.h"

namespace movba {

using namespace dense;

// ---------------------------------------------------------------------------------------------------------------------
// k_dense_assemble: S (damped) and b_S from the schur work-item partials into the tile layout; one workgroup per tile.
// Same item order as the PCG's own assembly, so both solvers see the same matrix to the last bit.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dense_assemble(DevWindow w)
{
    const Ctrl *c = w.ctrl;
    if (c->done == 1) return;
    const DenseSys &ds = w.dense;
    const int tid = threadIdx.x;
    // tile (I, J) of the lower block triangle from the linear block index
    int I = (int)((sqrt(8.0 * (double)blockIdx.x + 1.0) - 1.0) * 0.5);
    while ((I + 1) * (I + 2) / 2 <= (int)blockIdx.x) ++I;
    while (I * (I + 1) / 2 > (int)blockIdx.x) --I;
    const int J = (int)blockIdx.x - I * (I + 1) / 2;
    if (J >= ds.ntile) return;                                  // the corner tile behind the right-hand side is never read
    if (blockIdx.x == 0 && tid == 0) *ds.fail = 0;
    const double lambda = c->lambda;
    const int nf = w.nfree, n = ds.n;
    const double *part = w.part;
    double *dst = ds.tiles + tile_off(I, J);
    if (I == ds.ntile) {
        // right-hand side b_S = b_p - sum B Dinv b_l in row 0, zeros below
        for (int e = tid; e < NB * NB; e += 256) {
            const int r = e / NB, cc = e - r * NB, gc = J * NB + cc;
            double v = 0.0;
            if (r == 0 && gc < n) {
                const int bj = gc / 6, a = gc - bj * 6;
                double bb = 0.0, cb = 0.0;
                for (int itx = w.pair_item_start[bj]; itx < w.pair_item_start[bj + 1]; ++itx) {
                    bb += part[(size_t)itx * kPartStride + 63 + a];
                    cb += part[(size_t)itx * kPartStride + 36 + a];
                }
                w.bp[gc] = bb;
                v = bb - cb;
            }
            dst[e] = v;
        }
        return;
    }
    // nine elements per thread, their dependent loads (pair id -> item range -> partials) issued level by level for all nine:
    // one element after the other the kernel was a chain of 27 memory round trips (24 us for 36 tiles)
    constexpr int kPer = NB * NB / 256;
    static_assert(kPer * 256 == NB * NB, "whole passes");
    int pr[kPer], kk[kPer], uu[kPer], i0[kPer], i1[kPer];
    double pad[kPer];
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
        const int e = tid + 256 * q, r = e / NB, cc = e - r * NB;
        const int gr = I * NB + r, gc = J * NB + cc;
        pr[q] = -1; kk[q] = 0; uu[q] = -1; pad[q] = 0.0;
        if (gr >= n || gc >= n) pad[q] = (gr == gc) ? 1.0 : 0.0;           // padding rows: identity
        else {
            const int bi = gr / 6, a = gr - bi * 6, bj = gc / 6, b = gc - bj * 6;
            // upper-triangle pair (lo <= hi) holds S_lo,hi row-major; the lower block is its transpose
            const int lo = bi < bj ? bi : bj, hi = bi < bj ? bj : bi;
            pr[q] = ds.pid[(size_t)lo * nf + hi];
            kk[q] = bi <= bj ? a * 6 + b : b * 6 + a;
            if (bi == bj) { uu[q] = 42 + (a <= b ? ut6(a, b) : ut6(b, a)); pad[q] = a == b ? lambda : 0.0; }
        }
    }
#pragma unroll
    for (int q = 0; q < kPer; ++q) { i0[q] = pr[q] >= 0 ? w.pair_item_start[pr[q]] : 0; i1[q] = pr[q] >= 0 ? w.pair_item_start[pr[q] + 1] : 0; }
    double sacc[kPer], hpp[kPer];
#pragma unroll
    for (int q = 0; q < kPer; ++q) {                // first item of every element together, the (rare) further items after
        const bool any = i1[q] > i0[q];
        sacc[q] = any ? part[(size_t)i0[q] * kPartStride + kk[q]] : 0.0;
        hpp[q] = (any && uu[q] >= 0) ? part[(size_t)i0[q] * kPartStride + uu[q]] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < kPer; ++q)
        for (int itx = i0[q] + 1; itx < i1[q]; ++itx) {
            sacc[q] += part[(size_t)itx * kPartStride + kk[q]];
            if (uu[q] >= 0) hpp[q] += part[(size_t)itx * kPartStride + uu[q]];
        }
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
        double v = pad[q];
        if (pr[q] >= 0) v = uu[q] >= 0 ? (hpp[q] + pad[q]) - sacc[q] : -sacc[q];
        dst[tid + 256 * q] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_chol_step: block column j (see the header).  Workgroups [0, nfin) finalise tiles (j + b, j); the rest apply the
// previous column to the trailing tiles (only launched for j > 0).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kStepThreads) void k_chol_step(DevWindow w, int j)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const Ctrl *c = w.ctrl;
    if (c->done == 1) return;
    const DenseSys &ds = w.dense;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = ds.ntile, m = nt - j, nfin = m + 1;
    double *As = sm, *Bs = sm + NB * LD;            // L(I, j-1), L(K, j-1)
    double *Ps = Bs + NB * LD;                      // finalising workgroups: the 96 x 48 panel [D; U]

    if ((int)blockIdx.x >= nfin) {
        // ---- trailing tile (I, K), K > j:  tile -= L(I, j-1) L(K, j-1)^T, C kept in registers in the MFMA layout ----
        int t = (int)blockIdx.x - nfin, cc = 1;
        while (t >= m - cc + 1) { t -= m - cc + 1; ++cc; }
        const int K = j + cc, I = K + t;
        load_tile(ds.tiles + tile_off(I, j - 1), As, tid);
        load_tile(ds.tiles + tile_off(K, j - 1), Bs, tid);
        double *C = ds.tiles + tile_off(I, K);
        dbl4 acc[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int q = wv + 4 * u;                   // MFMA tiles dealt round-robin to the four waves
            if (q < 9) {
                const int mt = q / 3, ntc = q - mt * 3;
                const double *cp = C + (mt * 16 + (lane >> 4)) * NB + ntc * 16 + (lane & 15);
                acc[u] = dbl4{ cp[0], cp[4 * NB], cp[8 * NB], cp[12 * NB] };
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int q = wv + 4 * u;
            if (q < 9) {
                const int mt = q / 3, ntc = q - mt * 3;
                acc[u] = tile_mfma(As, Bs, mt, ntc, lane, acc[u]);
                double *cp = C + (mt * 16 + (lane >> 4)) * NB + ntc * 16 + (lane & 15);
                cp[0] = acc[u].x; cp[4 * NB] = acc[u].y; cp[8 * NB] = acc[u].z; cp[12 * NB] = acc[u].w;
            }
        }
        return;
    }

    // ---- finalising workgroup of tile (I, j) ----
    const int I = j + (int)blockIdx.x;
    const bool diag = I == j;
    // panel rows 0..47 = D = tile (j, j), rows 48..95 = U = tile (I, j) (unused for the diagonal workgroup)
    load_tile(ds.tiles + tile_off(j, j), Ps, tid);
    if (!diag) load_tile(ds.tiles + tile_off(I, j), Ps + NB * LD, tid);
    if (j > 0) {
        load_tile(ds.tiles + tile_off(j, j - 1), Bs, tid);
        if (!diag) load_tile(ds.tiles + tile_off(I, j - 1), As, tid);
    }
    __syncthreads();
    if (j > 0) {
        // D -= L(j, j-1) L(j, j-1)^T  (MFMA tiles 0..8),  U -= L(I, j-1) L(j, j-1)^T  (tiles 9..17), dealt round-robin to the waves
        const int ntiles = diag ? 9 : 18;
        dbl4 acc[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int q = wv + 4 * u;
            if (q < ntiles) {
                const int half = q >= 9, qq = q - 9 * half, mt = qq / 3, ntc = qq - mt * 3;
                const double *pp = Ps + (half * NB + mt * 16 + (lane >> 4)) * LD + ntc * 16 + (lane & 15);
                acc[u] = dbl4{ pp[0], pp[4 * LD], pp[8 * LD], pp[12 * LD] };
                acc[u] = tile_mfma(half ? As : Bs, Bs, mt, ntc, lane, acc[u]);
            }
        }
        __syncthreads();                            // every wave has read its C tiles (and the operands) before anyone writes
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int q = wv + 4 * u;
            if (q < ntiles) {
                const int half = q >= 9, qq = q - 9 * half, mt = qq / 3, ntc = qq - mt * 3;
                double *pp = Ps + (half * NB + mt * 16 + (lane >> 4)) * LD + ntc * 16 + (lane & 15);
                pp[0] = acc[u].x; pp[4 * LD] = acc[u].y; pp[8 * LD] = acc[u].z; pp[12 * LD] = acc[u].w;
            }
        }
        __syncthreads();
    }
    // ---- factor D, then solve U L^T = U, each by ONE wave with a panel row per lane in registers (lane r < 48: row r).
    // Pivot k of the factorisation:  p_rc -= (p_rk / p_kk) p_ck  for c > k; column k of D (the p_ck) is published in an
    // LDS table, T[k][.], and read back as broadcasts: no workgroup barrier inside the sweep, against 48 barriers and a
    // chain of LDS round trips per pivot with the panel spread over four waves (40 us per launch).  The upper triangle
    // of D is carried along as its symmetric image (same formula), so nothing is masked.  Wave 1 then runs the rows of
    // U through the same table (no cross-lane traffic at all).  L = P / sqrt(pivot) at the end. ----
    double *T = As;                                 // 48 x 64 doubles: As and Bs (adjacent, 2 x 48 x 49) are free now
    double *pivb = As + NB * 64, *rinvb = pivb + 64;
    lds_vint *prog = (lds_vint *)(rinvb + 64);       // pivots published so far (NB + 1: column scales too)
    static_assert(NB * 64 + 130 <= 2 * NB * LD, "the column table fits the two operand tiles");
    __syncthreads();                                // every wave is done with As / Bs as MFMA operands
    if (tid == 0) *prog = 0;
    __syncthreads();
    if (wv == 0) {
        bool bad = false;
        const int r = lane < NB ? lane : NB - 1;    // lanes 48..63 shadow row 47 (their results are never stored)
        double d[NB];
#pragma unroll
        for (int c = 0; c < NB; ++c) d[c] = Ps[r * LD + c];
        DiagSweep<0>::run(d, T, rinvb, pivb, prog, lane, bad);  // pivots 0..47, fully unrolled (register indices are static)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // 1 / sqrt(pivot) of every column, once (lane c), then the rows back into the panel image for the coalesced store
        if (lane < NB) { const double p = pivb[lane]; pivb[lane] = (p > 0.0 && isfinite(p)) ? 1.0 / sqrt(p) : 1.0; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane == 0) *prog = NB + 1;
        if (lane < NB) {
#pragma unroll
            for (int c = 0; c < NB; ++c) Ps[lane * LD + c] = c <= lane ? d[c] * pivb[c] : 0.0;
        }
        if (bad && lane == 0) *ds.fail = 1;
    }
    if (wv == 1 && !diag) {                         // beside the factorisation, a few pivots behind it
        const int r = lane < NB ? lane : NB - 1;
        double u[NB];
#pragma unroll
        for (int c = 0; c < NB; ++c) u[c] = Ps[(NB + r) * LD + c];
        RowSweep<0>::run(u, T, rinvb, prog);
        for (int guard = 0; *prog <= NB && guard < (1 << 20); ++guard) __builtin_amdgcn_s_sleep(1);        // the column scales
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (lane < NB) {
#pragma unroll
            for (int c = 0; c < NB; ++c) Ps[(NB + lane) * LD + c] = u[c] * pivb[c];
        }
    }
    __syncthreads();
    // ---- the diagonal workgroup leaves L(j, j) in diagL, the others L(I, j) in place ----
    {
        double2 *out = reinterpret_cast<double2 *>(diag ? ds.diagL + (size_t)j * NB * NB : ds.tiles + tile_off(I, j));
        const double *src = diag ? Ps : Ps + NB * LD;
        for (int e = tid; e < NB * NB / 2; e += kStepThreads) {
            const int r = (2 * e) / NB, c = (2 * e) - r * NB;
            out[e] = make_double2(src[r * LD + c], src[r * LD + c + 1]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_dense_backsolve: one workgroup.  y = L^-1 b sits in row 0 of block row `ntile`; solves L^T x = y from the last
// block column up, then writes the increment, the pose part of computeScale() and the trial poses exactly like the PCG
// kernels' epilogue, and releases a parked solve (Ctrl::done 2 -> 0).
// ---------------------------------------------------------------------------------------------------------------------
// k_back_step: back substitution of block row I for large systems, one launch per block row from the last one up
// (right-looking: y_J -= L(I, J)^T x_I for every J < I, one workgroup each).  Every workgroup solves the 48 x 48 triangular
// system L(I, I)^T x_I = y_I for itself first (one wave, ~1 us), so none waits for another; workgroup I stores x_I.
// One workgroup streaming the whole factor (k_dense_backsolve's loop) takes 1.1 ms at 400 free keyframes; this takes 50
// launches of a few microseconds.
__global__ __launch_bounds__(256) void k_back_step(DevWindow w, int I)
{
    __shared__ double Ls[NB * LD], rdiag[NB], xs[NB], ps[5 * NB];
    const Ctrl *c = w.ctrl;
    if (c->done == 1) return;
    const DenseSys &ds = w.dense;
    const int tid = threadIdx.x, nt = ds.ntile, J = blockIdx.x;
    for (int e = tid; e < NB * NB; e += 256) { const int r = e / NB, q = e - r * NB; Ls[r * LD + q] = ds.diagL[(size_t)I * NB * NB + e]; }
    if (tid < NB) rdiag[tid] = fast_rcp(ds.diagL[(size_t)I * NB * NB + tid * NB + tid]);
    // this workgroup's tile of the factor is requested before the substitution, so that it arrives behind it
    const int g = tid / NB, cc = tid - g * NB;      // 5 row groups x 48 columns = 240 threads
    double lv[10];
    const bool upd = J < I && g < 5;
#pragma unroll
    for (int q = 0; q < 10; ++q) { const int r = g + 5 * q; lv[q] = (upd && r < NB) ? ds.tiles[tile_off(I, J) + r * NB + cc] : 0.0; }
    __syncthreads();
    if (tid < 64) {
        double sv = tid < NB ? ds.tiles[tile_off(nt, I) + tid] : 0.0;
        double lk = tid < NB ? Ls[(NB - 1) * LD + tid] : 0.0, rk = rdiag[NB - 1];
        for (int k = NB - 1; k >= 0; --k) {
            const double lcur = lk, rcur = rk;
            if (k > 0) { lk = tid < NB ? Ls[(k - 1) * LD + tid] : 0.0; rk = rdiag[k - 1]; }
            const double xk = readlane_f64(sv, k) * rcur;
            if (tid == k) sv = xk;
            else if (tid < k) sv -= lcur * xk;
        }
        if (tid < NB) { xs[tid] = sv; if (J == I) ds.xsol[I * NB + tid] = sv; }
    }
    __syncthreads();
    if (J == I) return;
    if (upd) {
        double sacc = 0.0;
#pragma unroll
        for (int q = 0; q < 10; ++q) { const int r = g + 5 * q; if (r < NB) sacc += lv[q] * xs[r]; }
        ps[g * NB + cc] = sacc;
    }
    __syncthreads();
    if (tid < NB) {
        double *y = ds.tiles + tile_off(nt, J);
        y[tid] -= (((ps[tid] + ps[NB + tid]) + ps[2 * NB + tid]) + ps[3 * NB + tid]) + ps[4 * NB + tid];
    }
}

// PRESOLVED: the solution was left in DenseSys::xsol by the k_back_step launches (large systems); otherwise this
// workgroup runs the whole back substitution itself.
template <bool PRESOLVED>
__global__ __launch_bounds__(kBackThreads) void k_dense_backsolve(DevWindow w)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    Ctrl *c = w.ctrl;
    if (c->done == 1) return;
    const DenseSys &ds = w.dense;
    const int tid = threadIdx.x;
    const int nt = ds.ntile, n = ds.n, npad = nt * NB;
    double *x = sm;                                 // npad
    double *Ls = x + npad;                          // NB x LD: diagonal factor of the current block column
    double *ps = Ls + NB * LD;                      // 21 x NB partial sums
    double *red = ps + 21 * NB;                     // 16
    double *rdiag = red + 16;                       // NB: reciprocals of the diagonal of the current factor
    constexpr int G = 21;                           // row groups of the column sums: 21 x 48 = 1008 threads
    const int g = tid / NB, cc = tid - g * NB;
    const bool fail = *ds.fail != 0;
    const double lambda = c->lambda;
    if (PRESOLVED) {
        for (int idx = tid; idx < npad; idx += kBackThreads) x[idx] = ds.xsol[idx];
        __syncthreads();
    }
    for (int J = PRESOLVED ? -1 : nt - 1; J >= 0; --J) {
        // diagonal factor -> LDS, partial sums of  sum_{I > J} L(I, J)^T x_I  over the row groups
        for (int e = tid; e < NB * NB; e += kBackThreads) { const int r = e / NB, q = e - r * NB; Ls[r * LD + q] = ds.diagL[(size_t)J * NB * NB + e]; }
        double s = 0.0;
        if (g < G) {
            const int rows = (nt - 1 - J) * NB;
            for (int rr = g; rr < rows; rr += G) {
                const int I = J + 1 + rr / NB, r = rr - (rr / NB) * NB;
                s += ds.tiles[tile_off(I, J) + r * NB + cc] * x[I * NB + r];
            }
            ps[g * NB + cc] = s;
        }
        if (tid >= kBackThreads - NB) { const int k = tid - (kBackThreads - NB); rdiag[k] = fast_rcp(ds.diagL[(size_t)J * NB * NB + k * NB + k]); }
        __syncthreads();
        if (tid < 64) {
            // wave 0: s_c = y_c - (sums in group order), then the 48-step back substitution of L(J, J)^T x = s
            double sv = 0.0;
            if (tid < NB) {
                double acc = 0.0;
                for (int q = 0; q < G; ++q) acc += ps[q * NB + tid];
                sv = ds.tiles[tile_off(nt, J) + tid] - acc;
            }
            // (row k of the factor and its reciprocal diagonal are fetched one step ahead: the chain per step is
            //  readlane -> multiply -> fused update, not an LDS round trip)
            double lk = tid < NB ? Ls[(NB - 1) * LD + tid] : 0.0, rk = rdiag[NB - 1];
            for (int k = NB - 1; k >= 0; --k) {
                const double lcur = lk, rcur = rk;
                if (k > 0) { lk = tid < NB ? Ls[(k - 1) * LD + tid] : 0.0; rk = rdiag[k - 1]; }
                const double xk = readlane_f64(sv, k) * rcur;
                if (tid == k) sv = xk;
                else if (tid < k) sv -= lcur * xk;
            }
            if (tid < NB) x[J * NB + tid] = sv;
        }
        __syncthreads();
    }
    // ---- outputs ----
    double sc = 0.0;
    for (int idx = tid; idx < n; idx += kBackThreads) {
        const double xv = fail ? 0.0 : x[idx];
        w.xp[idx] = xv;
        sc += xv * (lambda * xv + w.bp[idx]);
        x[idx] = xv;
    }
    const double scs = block_reduce<kBackThreads / 64, false>(sc, red);
    const int cur = c->cur;
    const DevState &S0 = w.st[cur];
    const DevState &S1 = w.st[cur ^ 1];
    for (int i = tid; i < w.NP; i += kBackThreads) {
        double T[7], Tn[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) T[k] = S0.pose[7 * i + k];
        const int h = w.hidx[i];
        if (h >= 0 && !fail) {          // (a failed factorisation moves nothing: g2o returns from solve() before its update)
            double u[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) u[k] = x[6 * h + k];
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
    if (tid == 0) {
        w.scale_part[w.n_pt_blocks] = scs;
        c->pcg_fail = fail ? 1 : 0;
        c->pcg_last_iters = -1;                     // trace marker: this trial was solved directly
        c->n_direct += 1;
        if (fail) c->n_chol_fail += 1;
        if (c->done == 2) c->done = 0;              // the solve was parked for this: the kernels behind run again
    }
}

// ---------------------------------------------------------------------------------------------------------------------
size_t dense_tiles_doubles(int nfree)
{
    const size_t nt = ((size_t)6 * nfree + NB - 1) / NB;
    return (nt + 1) * (nt + 2) / 2 * (size_t)(NB * NB);
}

int dense_ntile(int nfree) { return (6 * nfree + NB - 1) / NB; }

hipError_t launch_dense_solve(const DevWindow &w, hipStream_t s)
{
    const int nt = w.dense.ntile;
    hipLaunchKernelGGL(k_dense_assemble, dim3((nt + 1) * (nt + 2) / 2), dim3(256), 0, s, w);
    const size_t lds = (size_t)(4 * NB * LD) * sizeof(double);
    for (int j = 0; j < nt; ++j) {
        const int m = nt - j;
        int grid = m + 1;
        if (j > 0) for (int cc = 1; cc < m; ++cc) grid += m - cc + 1;
        hipLaunchKernelGGL(k_chol_step, dim3(grid), dim3(kStepThreads), lds, s, w, j);
    }
    const size_t lds_b = ((size_t)nt * NB + NB * LD + 21 * NB + 16 + NB) * sizeof(double);
    if (nt > kBackStepsFrom) {
        for (int I = nt - 1; I >= 0; --I) hipLaunchKernelGGL(k_back_step, dim3(I + 1), dim3(256), 0, s, w, I);
        hipLaunchKernelGGL(k_dense_backsolve<true>, dim3(1), dim3(kBackThreads), lds_b, s, w);
    } else hipLaunchKernelGGL(k_dense_backsolve<false>, dim3(1), dim3(kBackThreads), lds_b, s, w);
    return hipGetLastError();
}

hipError_t configure_dense_kernels()
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_chol_step), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_dense_backsolve<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k_dense_backsolve<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
}

}  // namespace movba
