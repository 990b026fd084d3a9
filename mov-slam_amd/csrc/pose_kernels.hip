// Pose-only optimisation kernel behind Optimizer::PoseOptimization
// (/root/reference/src/Optimizer.cc:397-459).  The reference delegates that call to OpenCV's
// solvePnPRansac (USAC_MAGSAC), whose arithmetic is neither in the reference repository nor
// available here; what the reference does define is the motion-only edge
// EdgeSE3ProjectXYZOnlyPose (include/OptimizableTypes.h:30-58, src/OptimizableTypes.cpp:54-69),
// so this kernel runs g2o-style Levenberg over that edge (dense 6x6 system) in rounds with
// outlier re-classification, the scheme BASELINE.json's cfg1 "g2o CPU path" names
// (DESIGN.md §2, row A9).  One workgroup; the whole LM loop stays inside one launch.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>

#include "movba.h"
#include "pose_kernels.h"

namespace movba {

namespace {

constexpr int kT = 256;
constexpr int kW = kT / 64;

__device__ __forceinline__ void q2R(const double q[7], double R[12])
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
    R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
    R[9] = q[4]; R[10] = q[5]; R[11] = q[6];
}

__device__ __forceinline__ void qnorm(double q[4])
{
    if (q[3] < 0.0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

__device__ void oplus(const double u[6], const double T[7], double out[7])
{
    const double wx = u[0], wy = u[1], wz = u[2];
    const double th2 = wx * wx + wy * wy + wz * wz, th = sqrt(th2);
    double a, b, c, d;
    if (th < 0.00001) { a = 1.0; b = 0.5; c = 0.5; d = 1.0 / 6.0; }
    else { a = sin(th) / th; b = (1.0 - cos(th)) / th2; c = b; d = (th - sin(th)) / (th2 * th); }
    const double Om[9] = { 0.0, -wz, wy, wz, 0.0, -wx, -wy, wx, 0.0 };
    const double Om2[9] = { wx * wx - th2, wx * wy, wx * wz, wy * wx, wy * wy - th2, wy * wz, wz * wx, wz * wy, wz * wz - th2 };
    double m[9], V[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const double I = (i % 4 == 0) ? 1.0 : 0.0;
        m[i] = I + a * Om[i] + b * Om2[i];
        V[i] = I + c * Om[i] + d * Om2[i];
    }
    double e[7];
    double t = m[0] + m[4] + m[8];
    if (t > 0.0) {
        t = sqrt(t + 1.0); e[3] = 0.5 * t; t = 0.5 / t;
        e[0] = (m[7] - m[5]) * t; e[1] = (m[2] - m[6]) * t; e[2] = (m[3] - m[1]) * t;
    } else if (m[0] >= m[4] && m[0] >= m[8]) {
        t = sqrt(m[0] - m[4] - m[8] + 1.0); e[0] = 0.5 * t; t = 0.5 / t;
        e[3] = (m[7] - m[5]) * t; e[1] = (m[3] + m[1]) * t; e[2] = (m[6] + m[2]) * t;
    } else if (m[4] > m[0] && m[4] >= m[8]) {
        t = sqrt(m[4] - m[8] - m[0] + 1.0); e[1] = 0.5 * t; t = 0.5 / t;
        e[3] = (m[2] - m[6]) * t; e[2] = (m[7] + m[5]) * t; e[0] = (m[1] + m[3]) * t;
    } else {
        t = sqrt(m[8] - m[0] - m[4] + 1.0); e[2] = 0.5 * t; t = 0.5 / t;
        e[3] = (m[3] - m[1]) * t; e[0] = (m[2] + m[6]) * t; e[1] = (m[5] + m[7]) * t;
    }
    e[4] = V[0] * u[3] + V[1] * u[4] + V[2] * u[5];
    e[5] = V[3] * u[3] + V[4] * u[4] + V[5] * u[5];
    e[6] = V[6] * u[3] + V[7] * u[4] + V[8] * u[5];
    qnorm(e);
    double r[4];
    r[3] = e[3] * T[3] - e[0] * T[0] - e[1] * T[1] - e[2] * T[2];
    r[0] = e[3] * T[0] + e[0] * T[3] + e[1] * T[2] - e[2] * T[1];
    r[1] = e[3] * T[1] + e[1] * T[3] + e[2] * T[0] - e[0] * T[2];
    r[2] = e[3] * T[2] + e[2] * T[3] + e[0] * T[1] - e[1] * T[0];
    const double *v = T + 4;
    double ux = e[1] * v[2] - e[2] * v[1], uy = e[2] * v[0] - e[0] * v[2], uz = e[0] * v[1] - e[1] * v[0];
    ux += ux; uy += uy; uz += uz;
    const double rx = v[0] + e[3] * ux + (e[1] * uz - e[2] * uy);
    const double ry = v[1] + e[3] * uy + (e[2] * ux - e[0] * uz);
    const double rz = v[2] + e[3] * uz + (e[0] * uy - e[1] * ux);
    qnorm(r);
    out[0] = r[0]; out[1] = r[1]; out[2] = r[2]; out[3] = r[3];
    out[4] = e[4] + rx; out[5] = e[5] + ry; out[6] = e[6] + rz;
}

// fixed-order reduction of NV values per thread; result in every thread
template <int NV>
__device__ __forceinline__ void reduce_all(double (&v)[NV], double *lds /* kW*NV */)
{
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double s = v[k];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        v[k] = s;
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) lds[wave * NV + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double s = lds[k];
#pragma unroll
        for (int q = 1; q < kW; ++q) s += lds[q * NV + k];
        v[k] = s;
    }
    __syncthreads();
}

// dense 6x6 LL^T solve, every thread redundantly (H upper-triangle packed 21)
__device__ bool solve6(const double Hu[21], double lambda, const double b[6], double x[6])
{
    double L[36];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int i = a <= c ? a : c, j = a <= c ? c : a;
            L[a * 6 + c] = Hu[i * 6 - i * (i - 1) / 2 + (j - i)] + (a == c ? lambda : 0.0);
        }
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = L[j * 6 + j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= L[j * 6 + k] * L[j * 6 + k];
        if (!(d > 0.0) || !isfinite(d)) ok = false;
        d = sqrt(d);
        L[j * 6 + j] = d;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double s = L[i * 6 + j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= L[i * 6 + k] * L[j * 6 + k];
            L[i * 6 + j] = s / d;
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= L[i * 6 + k] * x[k];
        x[i] = s / L[i * 6 + i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = x[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) s -= L[k * 6 + i] * x[k];
        x[i] = s / L[i * 6 + i];
    }
    return ok;
}

}  // namespace

// STAGED: the matches are read ONCE from where the host left them (its pinned staging buffer, across the bus) into LDS,
// every pass of the 4 x 10 iterations then reads LDS, and the results are written straight back to the pinned buffer: no
// copy engine on either side of the launch (small copies cost ~0.1 ms each, as much as the kernel itself).
template <bool STAGED>
__global__ __launch_bounds__(kT) void k_pose_opt(PoseDev p)
{
    __shared__ double lds[kW * 28];
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    const int tid = threadIdx.x;
    const double *Xw = p.Xw, *obs = p.obs, *isig = p.isig;
    uint8_t *level1 = p.level1;
    if (STAGED) {
        double *sX = dyn, *so = sX + 3 * p.n, *si = so + 2 * p.n;
        for (int i = tid; i < 3 * p.n; i += kT) sX[i] = p.Xw[i];
        for (int i = tid; i < 2 * p.n; i += kT) so[i] = p.obs[i];
        for (int i = tid; i < p.n; i += kT) si[i] = p.isig[i];
        Xw = sX; obs = so; isig = si;
        level1 = reinterpret_cast<uint8_t *>(si + p.n);
    }
    const double dsqr = p.huber_delta * p.huber_delta;
    double pose0[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) pose0[k] = p.pose0[k];
    qnorm(pose0);
    double pose[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) pose[k] = pose0[k];
    for (int i = tid; i < p.n; i += kT) level1[i] = 0;
    __syncthreads();

    // robust cost of the active matches at a pose
    auto cost = [&](const double T[7], bool robust) {
        double R[12];
        q2R(T, R);
        double F[1] = { 0.0 };
        for (int i = tid; i < p.n; i += kT) {
            if (level1[i]) continue;
            const double X0 = Xw[3 * i], X1 = Xw[3 * i + 1], X2 = Xw[3 * i + 2];
            const double x = R[0] * X0 + R[1] * X1 + R[2] * X2 + R[9];
            const double y = R[3] * X0 + R[4] * X1 + R[5] * X2 + R[10];
            const double z = R[6] * X0 + R[7] * X1 + R[8] * X2 + R[11];
            const double om = isig[i];
            const double e0 = obs[2 * i] - (p.fx * x / z + p.cx), e1 = obs[2 * i + 1] - (p.fy * y / z + p.cy);
            const double chi2 = e0 * (om * e0) + e1 * (om * e1);
            F[0] += (robust && p.huber_delta > 0.0 && !(chi2 <= dsqr)) ? 2.0 * sqrt(chi2) * p.huber_delta - dsqr : chi2;
        }
        reduce_all<1>(F, lds);
        return F[0];
    };

    int n_bad = 0;
    for (int round = 0; round < p.rounds; ++round) {
        const bool robust = round <= 2;
#pragma unroll
        for (int k = 0; k < 7; ++k) pose[k] = pose0[k];
        double cnt[1] = { 0.0 };
        for (int i = tid; i < p.n; i += kT) cnt[0] += level1[i] ? 0.0 : 1.0;
        reduce_all<1>(cnt, lds);
        bool ok = cnt[0] > 0.0;
        double lambda = 0.0, ni = 2.0;
        for (int it = 0; it < p.its && ok; ++it) {
            // computeActiveErrors + buildSystem in one pass
            double R[12];
            q2R(pose, R);
            double acc[28];
#pragma unroll
            for (int k = 0; k < 28; ++k) acc[k] = 0.0;
            for (int i = tid; i < p.n; i += kT) {
                if (level1[i]) continue;
                const double X0 = Xw[3 * i], X1 = Xw[3 * i + 1], X2 = Xw[3 * i + 2];
                const double x = R[0] * X0 + R[1] * X1 + R[2] * X2 + R[9];
                const double y = R[3] * X0 + R[4] * X1 + R[5] * X2 + R[10];
                const double z = R[6] * X0 + R[7] * X1 + R[8] * X2 + R[11];
                const double om = isig[i];
                const double e0 = obs[2 * i] - (p.fx * x / z + p.cx), e1 = obs[2 * i + 1] - (p.fy * y / z + p.cy);
                const double chi2 = e0 * (om * e0) + e1 * (om * e1);
                double rho0 = chi2, rho1 = 1.0;
                if (robust && p.huber_delta > 0.0 && !(chi2 <= dsqr)) {
                    const double sq = sqrt(chi2);
                    rho0 = 2.0 * sq * p.huber_delta - dsqr; rho1 = p.huber_delta / sq;
                }
                const double wg = rho1 * om, r0 = -wg * e0, r1 = -wg * e1;
                const double a00 = -(p.fx / z), a02 = p.fx * x / (z * z), a11 = -(p.fy / z), a12 = p.fy * y / (z * z);
                const double C0[6] = { a02 * y, a00 * z - a02 * x, -a00 * y, a00, 0.0, a02 };
                const double C1[6] = { -a11 * z + a12 * y, -a12 * x, a11 * x, 0.0, a11, a12 };
                int u = 0;
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    acc[21 + a] += C0[a] * r0 + C1[a] * r1;
#pragma unroll
                    for (int c = a; c < 6; ++c) acc[u++] += wg * (C0[a] * C0[c] + C1[a] * C1[c]);
                }
                acc[27] += rho0;
            }
            reduce_all<28>(acc, lds);
            double F0 = acc[27];
            if (it == 0) {
                double md = 0.0;
#pragma unroll
                for (int a = 0; a < 6; ++a) md = fmax(md, fabs(acc[a * 6 - a * (a - 1) / 2]));
                lambda = 1e-5 * md; ni = 2.0;
            }
            double rho = 0.0;
            int qmax = 0;
            do {
                double x[6] = { 0, 0, 0, 0, 0, 0 }, bk[7];
#pragma unroll
                for (int k = 0; k < 7; ++k) bk[k] = pose[k];
                const bool ok2 = solve6(acc, lambda, acc + 21, x);
                if (ok2) oplus(x, bk, pose);
                double F1 = cost(pose, robust);
                if (!ok2) F1 = DBL_MAX;
                double scale = 1e-3;
                if (ok2) {
                    double s = 0.0;
#pragma unroll
                    for (int a = 0; a < 6; ++a) s += x[a] * (lambda * x[a] + acc[21 + a]);
                    scale += s;
                }
                rho = (F0 - F1) / scale;
                if (rho > 0.0 && isfinite(F1)) {
                    double alpha = 1.0 - pow(2.0 * rho - 1.0, 3.0);
                    alpha = fmin(alpha, 2.0 / 3.0);
                    lambda *= fmax(1.0 / 3.0, alpha); ni = 2.0; F0 = F1;
                } else {
                    lambda *= ni; ni *= 2.0;
#pragma unroll
                    for (int k = 0; k < 7; ++k) pose[k] = bk[k];
                    if (!isfinite(lambda)) { qmax++; break; }
                }
                qmax++;
            } while (rho < 0.0 && qmax < 10);
            if (qmax == 10 || rho == 0.0 || !isfinite(lambda)) ok = false;
        }
        // classify every match at the round's final pose (mvbOutlier, Optimizer.cc:452-456)
        double R[12];
        q2R(pose, R);
        double nb[1] = { 0.0 };
        __syncthreads();
        for (int i = tid; i < p.n; i += kT) {
            const double X0 = Xw[3 * i], X1 = Xw[3 * i + 1], X2 = Xw[3 * i + 2];
            const double x = R[0] * X0 + R[1] * X1 + R[2] * X2 + R[9];
            const double y = R[3] * X0 + R[4] * X1 + R[5] * X2 + R[10];
            const double z = R[6] * X0 + R[7] * X1 + R[8] * X2 + R[11];
            const double om = isig[i];
            const double e0 = obs[2 * i] - (p.fx * x / z + p.cx), e1 = obs[2 * i + 1] - (p.fy * y / z + p.cy);
            const double chi2 = e0 * (om * e0) + e1 * (om * e1);
            const int bad = (chi2 > p.chi2_gate) || !(z > 0.0);
            p.chi2[i] = chi2;
            level1[i] = (uint8_t)bad;
            nb[0] += bad;
        }
        reduce_all<1>(nb, lds);
        n_bad = (int)nb[0];
        if (p.n - n_bad < 10) break;
    }
    if (STAGED) {
        __syncthreads();
        for (int i = tid; i < p.n; i += kT) p.level1[i] = level1[i];
    }
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k) p.pose_out[k] = pose[k];
        p.pose_out[7] = (double)(p.n - n_bad);
    }
}

size_t pose_opt_staged_lds_bytes(int n) { return (size_t)n * 6 * sizeof(double) + (((size_t)n + 15) & ~(size_t)15); }

hipError_t configure_pose_kernels()
{
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k_pose_opt<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
}

hipError_t launch_pose_opt(const PoseDev &p, bool staged, hipStream_t s)
{
    if (staged) hipLaunchKernelGGL(k_pose_opt<true>, dim3(1), dim3(kT), pose_opt_staged_lds_bytes(p.n), s, p);
    else hipLaunchKernelGGL(k_pose_opt<false>, dim3(1), dim3(kT), 0, s, p);
    return hipGetLastError();
}

}  // namespace movba
