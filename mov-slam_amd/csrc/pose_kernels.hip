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

#include "device_math.h"
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

// (SE3 update and quaternion helpers: device_math.h, shared with the local-BA kernels)
__device__ __forceinline__ void qnorm(double q[4]) { quat_normalize(q); }
__device__ __forceinline__ void oplus(const double u[6], const double T[7], double out[7]) { se3_oplus(u, T, out); }

// fixed-order reduction of NV values per thread; result in every thread (wave-uniform: scalar registers).  Inside a wave: two
// DPP steps sum each quad, the 16 quad sums of every value cross a wave-private LDS strip and lane k adds those of value k up in
// order (a butterfly of 64-bit shuffles per value costs ~500 cycles each: 28 of them were three quarters of an LM iteration);
// one or two values take the whole-wave DPP tree instead.  Then ONE workgroup barrier: lane k of EVERY wave adds the waves' sums
// of value k in wave order and the values go round by v_readlane - until round 5 every thread read all kW x NV sums back from
// LDS itself behind a second barrier (112 eight-byte loads of a lone wave per SIMD: 1 400 of the reduction's 3 600 cycles).
// The waves' sums alternate between two places (`flip`), so a wave that runs ahead into the next reduction cannot overwrite
// what a slower one still reads: it stops at that reduction's barrier first.
constexpr int kRedMax = 28;
constexpr int kRedLds = 2 * kW * kRedMax + kW * kRedMax * 16;
template <int NV>
__device__ __forceinline__ void reduce_all(double (&v)[NV], double *lds /* kRedLds */, int &flip)
{
    static_assert(NV <= kRedMax, "reduce_all: LDS strips sized for 28 values");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *cross = lds + flip * (kW * kRedMax);
    flip ^= 1;
    if (NV <= 2) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const double t = wave_sum_dpp(v[k]);
            if (lane == 0) cross[wave * NV + k] = t;
        }
    } else {
        double *strip = lds + 2 * kW * kRedMax + wave * (NV * 16);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            double t = v[k];
            t += dpp_mov0<0xb1>(t);
            t += dpp_mov0<0x4e>(t);
            if ((lane & 3) == 0) strip[k * 16 + (lane >> 2)] = t;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < NV) {
            const double2 *src = reinterpret_cast<const double2 *>(strip + lane * 16);
            double2 t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = src[q];
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q) { s += t[q].x; s += t[q].y; }
            cross[wave * NV + lane] = s;
        }
    }
    __syncthreads();
    const int kk = lane < NV ? lane : NV - 1;
    double w[kW];
#pragma unroll
    for (int q = 0; q < kW; ++q) w[q] = cross[q * NV + kk];
    double tot = w[0];
#pragma unroll
    for (int q = 1; q < kW; ++q) tot += w[q];
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = readlane_f64(tot, k);
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
    // (every thread runs this chain by itself with one wave per SIMD: nothing hides its latency, and an fp64 division or
    // square root is a ~30-instruction sequence of its own.  One reciprocal square root per pivot and multiplications
    // instead of 6 square roots and 27 divisions: the chain was a third of an LM iteration)
    bool ok = true;
    double inv[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = L[j * 6 + j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= L[j * 6 + k] * L[j * 6 + k];
        if (!(d > 0.0) || !isfinite(d)) ok = false;
        const double r = rsqrt(d);
        inv[j] = r;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double s = L[i * 6 + j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= L[i * 6 + k] * L[j * 6 + k];
            L[i * 6 + j] = s * r;
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= L[i * 6 + k] * x[k];
        x[i] = s * inv[i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = x[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) s -= L[k * 6 + i] * x[k];
        x[i] = s * inv[i];
    }
    return ok;
}

// ---------------------------------------------------------------------------------------------------------------------
// Hypothesis stage in front of the LM: RANSAC over minimal P3P solves, so that the result does not depend on the pose the
// caller's Frame happens to hold.  The reference's PoseOptimization IS a RANSAC-PnP (cv::solvePnPRansac with
// useExtrinsicGuess = false, /root/reference/src/Optimizer.cc:437; USAC_MAGSAC over P3P): it recovers from a stale
// Frame pose (mState == RECENTLY_LOST, Tracking.cc:806) and from match sets dominated by outliers.  OpenCV's arithmetic is
// not in the reference tree; this is the classical scheme it implements: H minimal samples (host-drawn, seeded), the
// Grunert three-point solution of each (Haralick et al. 1994, eq. for the quartic in v = s3 / s1), every candidate pose
// scored on all matches by its inlier count at the caller's reprojection threshold (ties: truncated cost), the best one
// starts the LM.
// ---------------------------------------------------------------------------------------------------------------------
struct Cplx { double re, im; };
__device__ __forceinline__ Cplx cmul(Cplx a, Cplx b) { return Cplx{ a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re }; }
__device__ __forceinline__ Cplx csub(Cplx a, Cplx b) { return Cplx{ a.re - b.re, a.im - b.im }; }
__device__ __forceinline__ Cplx cdiv(Cplx a, Cplx b)
{
    const double id = 1.0 / (b.re * b.re + b.im * b.im);
    return Cplx{ (a.re * b.re + a.im * b.im) * id, (a.im * b.re - a.re * b.im) * id };
}

// all four roots of z^4 + c3 z^3 + c2 z^2 + c1 z + c0 (Durand-Kerner from fixed starting points; sweeps until no root
// moves by more than 1e-15 of the root bound, 80 at most: a sweep is a chain of ~400 dependent fp64 instructions that
// nothing hides — one thread per hypothesis —, and 80 of them were a third of a PoseOptimization call; the usual quartic
// is through after 10 - 20)
__device__ void quartic_roots(double c3, double c2, double c1, double c0, Cplx z[4])
{
    // Fujiwara bound: every root lies within 2 max(|c3|, |c2|^1/2, |c1|^1/3, |c0|^1/4)
    const double rb = 2.0 * fmax(fmax(fabs(c3), sqrt(fabs(c2))), fmax(cbrt(fabs(c1)), sqrt(sqrt(fabs(c0))))) + 1e-300;
    const double r0 = 0.5 * rb;
    z[0] = Cplx{ r0 * 0.9210609940028851, r0 * 0.3894183423086505 };      // radius r0, angles 0.4 + k pi / 2
    z[1] = Cplx{ -z[0].im, z[0].re }; z[2] = Cplx{ -z[0].re, -z[0].im }; z[3] = Cplx{ z[0].im, -z[0].re };
    const double tol2 = (1e-15 * rb) * (1e-15 * rb);
    for (int it = 0; it < 80; ++it) {
        double moved = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const Cplx x = z[k];
            // p(x) by Horner
            Cplx pv = Cplx{ x.re + c3, x.im };
            pv = cmul(pv, x); pv.re += c2;
            pv = cmul(pv, x); pv.re += c1;
            pv = cmul(pv, x); pv.re += c0;
            Cplx den = Cplx{ 1.0, 0.0 };
#pragma unroll
            for (int j = 0; j < 4; ++j) if (j != k) den = cmul(den, csub(x, z[j]));
            if (den.re * den.re + den.im * den.im > 0.0) {
                const Cplx st = cdiv(pv, den);
                z[k] = csub(x, st);
                moved = fmax(moved, st.re * st.re + st.im * st.im);
            }
        }
        if (moved <= tol2) break;
    }
}

__device__ __forceinline__ void cross3(const double a[3], const double b[3], double o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}

// orthonormal frame of a point triple: e1 along P2 - P1, e3 normal to the triangle, e2 = e3 x e1; false when degenerate
__device__ bool triple_frame(const double P1[3], const double P2[3], const double P3[3], double F[9])
{
    double d1[3] = { P2[0] - P1[0], P2[1] - P1[1], P2[2] - P1[2] }, d2[3] = { P3[0] - P1[0], P3[1] - P1[1], P3[2] - P1[2] };
    const double n1 = sqrt(d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2]);
    if (!(n1 > 0.0)) return false;
    d1[0] /= n1; d1[1] /= n1; d1[2] /= n1;
    double e3[3];
    cross3(d1, d2, e3);
    const double n3 = sqrt(e3[0] * e3[0] + e3[1] * e3[1] + e3[2] * e3[2]);
    if (!(n3 > 1e-12 * n1)) return false;
    e3[0] /= n3; e3[1] /= n3; e3[2] /= n3;
    double e2[3];
    cross3(e3, d1, e2);
    // columns e1 e2 e3
    F[0] = d1[0]; F[3] = d1[1]; F[6] = d1[2]; F[1] = e2[0]; F[4] = e2[1]; F[7] = e2[2]; F[2] = e3[0]; F[5] = e3[1]; F[8] = e3[2];
    return true;
}

// Grunert's P3P: world points X[3][3], unit bearings j[3][3] -> up to 4 poses (R row-major, t) with X_cam = R X + t
__device__ int p3p_grunert(const double X[3][3], const double j[3][3], double Rs[4][9], double ts[4][3])
{
    auto d2 = [](const double *a, const double *b) { const double x = a[0] - b[0], y = a[1] - b[1], z = a[2] - b[2]; return x * x + y * y + z * z; };
    const double a2 = d2(X[1], X[2]), b2 = d2(X[0], X[2]), c2 = d2(X[0], X[1]);
    if (!(b2 > 0.0) || !(a2 > 0.0) || !(c2 > 0.0)) return 0;
    const double ca = j[1][0] * j[2][0] + j[1][1] * j[2][1] + j[1][2] * j[2][2];
    const double cb = j[0][0] * j[2][0] + j[0][1] * j[2][1] + j[0][2] * j[2][2];
    const double cg = j[0][0] * j[1][0] + j[0][1] * j[1][1] + j[0][2] * j[1][2];
    const double q = (a2 - c2) / b2, pp = (a2 + c2) / b2;
    const double A4 = (q - 1.0) * (q - 1.0) - 4.0 * c2 / b2 * ca * ca;
    const double A3 = 4.0 * (q * (1.0 - q) * cb - (1.0 - pp) * ca * cg + 2.0 * c2 / b2 * ca * ca * cb);
    const double A2 = 2.0 * (q * q - 1.0 + 2.0 * q * q * cb * cb + 2.0 * (b2 - c2) / b2 * ca * ca - 4.0 * pp * ca * cb * cg + 2.0 * (b2 - a2) / b2 * cg * cg);
    const double A1 = 4.0 * (-q * (1.0 + q) * cb + 2.0 * a2 / b2 * cg * cg * cb - (1.0 - pp) * ca * cg);
    const double A0 = (1.0 + q) * (1.0 + q) - 4.0 * a2 / b2 * cg * cg;
    const double mx = fmax(fmax(fabs(A4), fabs(A3)), fmax(fmax(fabs(A2), fabs(A1)), fabs(A0)));
    if (!(fabs(A4) > 1e-12 * mx) || !isfinite(mx)) return 0;
    const double c3 = A3 / A4, c2q = A2 / A4, c1 = A1 / A4, c0 = A0 / A4;
    Cplx z[4];
    quartic_roots(c3, c2q, c1, c0, z);
    double Fw[9];
    if (!triple_frame(X[0], X[1], X[2], Fw)) return 0;
    int ns = 0;
    for (int k = 0; k < 4; ++k) {
        if (!(fabs(z[k].im) <= 1e-6 * (1.0 + fabs(z[k].re)))) continue;
        double v = z[k].re;
        for (int it = 0; it < 2; ++it) {            // polish on the real quartic
            const double f = (((v + c3) * v + c2q) * v + c1) * v + c0, df = ((4.0 * v + 3.0 * c3) * v + 2.0 * c2q) * v + c1;
            if (df != 0.0) v -= f / df;
        }
        if (!(v > 0.0)) continue;
        const double den = 2.0 * (cg - v * ca);
        if (!(fabs(den) > 1e-12)) continue;
        const double u = ((q - 1.0) * v * v - 2.0 * q * cb * v + 1.0 + q) / den;
        if (!(u > 0.0)) continue;
        const double dd = 1.0 + v * v - 2.0 * v * cb;
        if (!(dd > 0.0)) continue;
        const double s1 = sqrt(b2 / dd), s2 = u * s1, s3 = v * s1;
        const double P1[3] = { s1 * j[0][0], s1 * j[0][1], s1 * j[0][2] }, P2[3] = { s2 * j[1][0], s2 * j[1][1], s2 * j[1][2] },
                     P3[3] = { s3 * j[2][0], s3 * j[2][1], s3 * j[2][2] };
        double Fc[9];
        if (!triple_frame(P1, P2, P3, Fc)) continue;
        // R = Fc Fw^T
        double *R = Rs[ns];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) R[r * 3 + c] = Fc[r * 3] * Fw[c * 3] + Fc[r * 3 + 1] * Fw[c * 3 + 1] + Fc[r * 3 + 2] * Fw[c * 3 + 2];
#pragma unroll
        for (int r = 0; r < 3; ++r) ts[ns][r] = P1[r] - (R[r * 3] * X[0][0] + R[r * 3 + 1] * X[0][1] + R[r * 3 + 2] * X[0][2]);
        ++ns;
    }
    return ns;
}

// rotation matrix (row-major) -> unit quaternion (x, y, z, w), w >= 0 (Eigen's conversion, as in oplus above)
__device__ void R2q(const double m[9], double q[4])
{
    double t = m[0] + m[4] + m[8];
    if (t > 0.0) {
        t = sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t;
        q[0] = (m[7] - m[5]) * t; q[1] = (m[2] - m[6]) * t; q[2] = (m[3] - m[1]) * t;
    } else if (m[0] >= m[4] && m[0] >= m[8]) {
        t = sqrt(m[0] - m[4] - m[8] + 1.0); q[0] = 0.5 * t; t = 0.5 / t;
        q[3] = (m[7] - m[5]) * t; q[1] = (m[3] + m[1]) * t; q[2] = (m[6] + m[2]) * t;
    } else if (m[4] > m[0] && m[4] >= m[8]) {
        t = sqrt(m[4] - m[8] - m[0] + 1.0); q[1] = 0.5 * t; t = 0.5 / t;
        q[3] = (m[2] - m[6]) * t; q[2] = (m[7] + m[5]) * t; q[0] = (m[1] + m[3]) * t;
    } else {
        t = sqrt(m[8] - m[0] - m[4] + 1.0); q[2] = 0.5 * t; t = 0.5 / t;
        q[3] = (m[3] - m[1]) * t; q[0] = (m[2] + m[6]) * t; q[1] = (m[5] + m[7]) * t;
    }
    qnorm(q);
}

}  // namespace

// minimal sample h -> up to four candidate poses (cand[(4 h + k) * 12]: R row-major, t) and their number (nsol[h])
__device__ void hyp_solve(const PoseDev &p, const double *Xw, const double *obs, int h, double *cand, int *nsol)
{
    double X[3][3], jb[3][3];
    bool okh = true;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const int i = p.samples[3 * h + m];
        okh &= i >= 0 && i < p.n;
        const int ii = okh ? i : 0;
        X[m][0] = Xw[3 * ii]; X[m][1] = Xw[3 * ii + 1]; X[m][2] = Xw[3 * ii + 2];
        const double bx = (obs[2 * ii] - p.cx) / p.fx, by = (obs[2 * ii + 1] - p.cy) / p.fy;
        const double nn = 1.0 / sqrt(bx * bx + by * by + 1.0);
        jb[m][0] = bx * nn; jb[m][1] = by * nn; jb[m][2] = nn;
    }
    double Rs[4][9], ts[4][3];
    const int ns = okh ? p3p_grunert(X, jb, Rs, ts) : 0;
    nsol[h] = ns;
    for (int k = 0; k < ns; ++k) {
        double *c = cand + ((size_t)h * 4 + k) * 12;
#pragma unroll
        for (int e = 0; e < 9; ++e) c[e] = Rs[k][e];
        c[9] = ts[k][0]; c[10] = ts[k][1]; c[11] = ts[k][2];
    }
}

// sigma-consensus++ (MAGSAC++: Barath, Noskova, Ivashechkin, Matas, CVPR 2020), what flag 38 = cv::USAC_MAGSAC of the reference's
// call scores models with (src/Optimizer.cc:437, Examples/Monocular/TartanAir.yaml:51): a residual is not classified at ONE
// threshold, its noise scale is marginalised over (0, sigma_max], sigma_max = tau / k, tau^2 = chi2_gate (the caller's
// reprojectionError), k^2 = 9.21034 (0.99 quantile of chi^2 with the residual's 2 degrees of freedom).  With x = r^2 / (2
// sigma_max^2), x_k = k^2 / 2 the paper's incomplete gamma functions are elementary for n = 2:
//     weight  w(r)   = sqrt(pi) (erfc(sqrt x) - erfc(sqrt x_k))                                             r <= tau, else 0
//     loss    rho(r) = sigma_max^2 / 2 (sqrt(pi) / 2 erf(sqrt x) - sqrt x exp(-x)) + r^2 / 4 w(r)           r <= tau, rho(tau) beyond
// (common factors dropped).  loss: rho / rho(tau) in [0, 1], 1 = outlier or behind the camera; weight: w / w(0).
struct Magsac {
    double s2, g_k, inv_rho_max, inv_w0;
    __device__ explicit Magsac(double gate)
    {
        const double sq_pi = 1.7724538509055160273, k2 = 9.210340371976184, xk = 0.5 * k2;
        s2 = gate / k2;
        g_k = sq_pi * erfc(sqrt(xk));
        inv_rho_max = 1.0 / (0.5 * s2 * (0.5 * sq_pi * erf(sqrt(xk)) - sqrt(xk) * exp(-xk)));
        inv_w0 = 1.0 / (sq_pi * (1.0 - erfc(sqrt(xk))));
    }
    __device__ void terms(double chi2, bool in_front, double gate, double &loss, double &weight) const
    {
        const double sq_pi = 1.7724538509055160273;
        if (!in_front || !(chi2 <= gate)) { loss = 1.0; weight = 0.0; return; }
        const double x = chi2 / (2.0 * s2), sx = sqrt(x);
        const double w = sq_pi * erfc(sx) - g_k;
        loss = (0.5 * s2 * (0.5 * sq_pi * erf(sx) - sx * exp(-x)) + 0.25 * chi2 * w) * inv_rho_max;
        weight = w > 0.0 ? w * inv_w0 : 0.0;
    }
};

// one wave scores candidate cidx on all matches: matches inside the caller's threshold, sigma-consensus++ loss (score[2 cidx], [2 cidx + 1])
__device__ void hyp_score(const PoseDev &p, const double *Xw, const double *obs, const double *isig, int cidx, const double *cand,
                          const int *nsol, double *score, int lane)
{
    double cnt = 0.0, cst = 0.0;
    if ((cidx & 3) < nsol[cidx >> 2]) {
        const Magsac ms(p.chi2_gate);
        const double *R = cand + (size_t)cidx * 12;
        for (int i = lane; i < p.n; i += 64) {
            const double X0 = Xw[3 * i], X1 = Xw[3 * i + 1], X2 = Xw[3 * i + 2];
            const double x = R[0] * X0 + R[1] * X1 + R[2] * X2 + R[9];
            const double y = R[3] * X0 + R[4] * X1 + R[5] * X2 + R[10];
            const double z = R[6] * X0 + R[7] * X1 + R[8] * X2 + R[11];
            const double om = isig[i];
            const double e0 = obs[2 * i] - (p.fx * x / z + p.cx), e1 = obs[2 * i + 1] - (p.fy * y / z + p.cy);
            const double chi2 = e0 * (om * e0) + e1 * (om * e1);
            const bool in = (z > 0.0) && (chi2 <= p.chi2_gate);
            double ls, wt;
            ms.terms(chi2, z > 0.0, p.chi2_gate, ls, wt);
            cnt += in ? 1.0 : 0.0; cst += ls;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { cnt += __shfl_xor(cnt, o, 64); cst += __shfl_xor(cst, o, 64); }
    } else { cnt = -1.0; cst = DBL_MAX; }
    if (lane == 0) { score[2 * cidx] = cnt; score[2 * cidx + 1] = cst; }
}

__device__ __forceinline__ void hyp_tables(const PoseDev &p, double *cand, double *&score, int *&nsol)
{
    score = cand + (size_t)p.n_hyp * 48;
    nsol = reinterpret_cast<int *>(score + (size_t)p.n_hyp * 8);
}

// The hypothesis stage over the whole chip: workgroup h solves sample h (one thread: a chain of dependent fp64 work) and
// its four waves score the four candidates.  Inside k_pose_opt's single workgroup the scoring alone — 4 n_hyp candidates x n
// matches x ~120 fp64 instructions on one CU — took 0.18 ms of a 0.42 ms call at 50 hypotheses and 500 matches.
__global__ __launch_bounds__(kT) void k_pose_hyp(PoseDev p)
{
    double *score; int *nsol;
    hyp_tables(p, p.cand, score, nsol);
    const int h = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) hyp_solve(p, p.Xw, p.obs, h, p.cand, nsol);
    __syncthreads();
    static_assert(kW == 4, "one wave per candidate of a sample");
    hyp_score(p, p.Xw, p.obs, p.isig, 4 * h + (tid >> 6), p.cand, nsol, score, tid & 63);
}

// STAGED: the matches are read ONCE from where the host left them (its pinned staging buffer, across the bus) into LDS,
// every pass of the 4 x 10 iterations then reads LDS, and the results are written straight back to the pinned buffer: no
// copy engine on either side of the launch (small copies cost ~0.1 ms each, as much as the kernel itself).
template <bool STAGED>
__global__ __launch_bounds__(kT) void k_pose_opt(PoseDev p)
{
    __shared__ __attribute__((aligned(16))) double lds[kRedLds];      // reduce_all: the waves' sums (two places), then a strip per wave
    int flip = 0;
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
        level1 = reinterpret_cast<uint8_t *>(si + 2 * p.n);     // (behind the n IRLS weights of the local-optimisation step)
    }
    const double dsqr = p.huber_delta * p.huber_delta;
    double pose0[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) pose0[k] = p.pose0[k];
    quat_normalize_exact(pose0);
    for (int i = tid; i < p.n; i += kT) level1[i] = 0;
    __syncthreads();
    bool have_hyp = false;
    if (p.n_hyp > 0) {
        // ---- hypothesis stage (see the header of the P3P block): thread h solves sample h, then every wave scores candidates ----
        // n_hyp x 4 x 12 poses, then x 2 scores: in LDS when this workgroup runs the stage itself and the matches are staged,
        // in device memory otherwise (hyp_done: k_pose_hyp has filled them)
        double *cand = (STAGED && !p.hyp_done) ? reinterpret_cast<double *>(level1 + (((size_t)p.n + 15) & ~(size_t)15)) : p.cand;
        double *score; int *nsol;
        hyp_tables(p, cand, score, nsol);
        if (!p.hyp_done) {
            for (int h = tid; h < p.n_hyp; h += kT) hyp_solve(p, Xw, obs, h, cand, nsol);
            __syncthreads();
            const int lane = tid & 63, wv = tid >> 6;
            for (int cidx = wv; cidx < 4 * p.n_hyp; cidx += kW) hyp_score(p, Xw, obs, isig, cidx, cand, nsol, score, lane);
        }
        __syncthreads();
        // best candidate: lowest sigma-consensus++ loss among those with at least 4 matches inside the threshold, then lowest
        // index; every thread scans the same table.
        // The samples are walked in the order a sequential RANSAC would draw them, with its stopping rule (cv::solvePnPRansac's
        // `confidence`, Optimizer.cc:437): after sample h, N = log(1 - confidence) / log(1 - w^3) with w the inlier ratio of the best
        // pose so far; the walk ends once h + 1 >= N — hypotheses behind that point were scored (all at once, on the grid) but are
        // not eligible, as they would never have been drawn.
        int best = -1, used = p.n_hyp; double bc = 3.5, bs = DBL_MAX;   // a pose needs at least 4 inliers to replace the caller's
        const bool stop_rule = p.confidence > 0.0 && p.confidence < 1.0;
        const double lconf = stop_rule ? log(1.0 - p.confidence) : 0.0;
        for (int h = 0; h < p.n_hyp; ++h) {
            for (int cidx = 4 * h; cidx < 4 * h + 4; ++cidx) {
                const double cnt = score[2 * cidx], cst = score[2 * cidx + 1];
                if (cnt > 3.5 && cst < bs) { best = cidx; bc = cnt; bs = cst; }
            }
            if (stop_rule && best >= 0) {
                const double wr = bc / (double)p.n, w3 = wr * wr * wr;
                const double need = w3 >= 1.0 ? 0.0 : lconf / log(1.0 - w3);
                if ((double)(h + 1) >= need) { used = h + 1; break; }
            }
        }
        if (best >= 0) {
            const double *R = cand + (size_t)best * 12;
            double Rm[9];
#pragma unroll
            for (int e = 0; e < 9; ++e) Rm[e] = R[e];
            R2q(Rm, pose0);
            pose0[4] = R[9]; pose0[5] = R[10]; pose0[6] = R[11];
        }
        have_hyp = best >= 0;
        if (tid == 0) {
#pragma unroll
            for (int k = 0; k < 7; ++k) p.pose_out[9 + k] = pose0[k];
        }
        if (tid == 0) { p.pose_out[8] = best >= 0 ? bc : 0.0; p.pose_out[17] = (double)used; p.pose_out[18] = 0.0; p.pose_out[19] = best >= 0 ? bc : 0.0; }
        __syncthreads();
    }
    double pose[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) pose[k] = pose0[k];

    double *wls = STAGED ? dyn + 6 * (size_t)p.n : p.chi2;       // (not staged: the chi2 output array, rewritten by every round below)
    const double *wuse = nullptr;                               // weights of the LM passes (the local-optimisation step), or none
    // robust cost of the active matches at a pose
    // The passes over the matches take a thread's matches kFly at a time: every load of the batch is requested before the first
    // is used, and a match that is switched off (level1) is not branched round but run with a harmless point and weight zero -
    // it adds exact zeros.  (One match at a time behind `if (level1[i]) continue`, every match paid its own LDS round trip and
    // a vector-compare-to-branch hop: ~1 050 cycles per match in the system pass for ~600 of arithmetic, profiles/r05_pose_stamps.log.)
    constexpr int kFly = 4;
    struct Batch { double X[kFly][3], o[kFly][2], om[kFly]; bool in[kFly]; };
    auto fetch = [&](int i0, Batch &b) {
        // (no branch and no wait between the loads: `iu < n && level1[i] == 0` written with && put a wait-and-branch
        //  in front of every match's loads)
        uint8_t lv[kFly];
        double wq[kFly];
#pragma unroll
        for (int u = 0; u < kFly; ++u) {
            const int i = min(i0 + u * kT, p.n - 1);
            lv[u] = level1[i];
            b.X[u][0] = Xw[3 * i]; b.X[u][1] = Xw[3 * i + 1]; b.X[u][2] = Xw[3 * i + 2];
            b.o[u][0] = obs[2 * i]; b.o[u][1] = obs[2 * i + 1];
            b.om[u] = isig[i];
        }
        if (wuse) {
#pragma unroll
            for (int u = 0; u < kFly; ++u) wq[u] = wuse[min(i0 + u * kT, p.n - 1)];
#pragma unroll
            for (int u = 0; u < kFly; ++u) b.om[u] *= wq[u];
        }
#pragma unroll
        for (int u = 0; u < kFly; ++u) b.in[u] = (i0 + u * kT < p.n) & (lv[u] == 0);
    };
    auto cost = [&](const double T[7], bool robust) {
        double R[12];
        q2R(T, R);
        double F[1] = { 0.0 };
        for (int i0 = tid; i0 < p.n; i0 += kFly * kT) {
            Batch b;
            fetch(i0, b);
            const int w0 = __builtin_amdgcn_readfirstlane(i0);          // (the wave's lowest index: places wholly past the end are skipped)
#pragma unroll
            for (int u = 0; u < kFly; ++u) {
                if (w0 + u * kT >= p.n) break;
                const double X0 = b.X[u][0], X1 = b.X[u][1], X2 = b.X[u][2];
                const bool in = b.in[u];
                const double x = in ? R[0] * X0 + R[1] * X1 + R[2] * X2 + R[9] : 0.0;
                const double y = in ? R[3] * X0 + R[4] * X1 + R[5] * X2 + R[10] : 0.0;
                const double z = in ? R[6] * X0 + R[7] * X1 + R[8] * X2 + R[11] : 1.0;
                const double om = in ? b.om[u] : 0.0;
                const double iz = fast_rcp(z);
                const double e0 = b.o[u][0] - (p.fx * x * iz + p.cx), e1 = b.o[u][1] - (p.fy * y * iz + p.cy);
                const double chi2 = e0 * (om * e0) + e1 * (om * e1);
                F[0] += (robust && p.huber_delta > 0.0 && !(chi2 <= dsqr)) ? 2.0 * sqrt(chi2) * p.huber_delta - dsqr : chi2;
            }
        }
        reduce_all<1>(F, lds, flip);
        return F[0];
    };

    int n_bad = 0, n_lm = 0;
#ifdef MOVBA_CLOCK_STAMP
    unsigned long long pseg[4] = { 0, 0, 0, 0 }, plast = 0;
#define POSE_STAMP0() do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(plast) :: "memory"); } while (0)
#define POSE_STAMP(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); pseg[k] += t_ - plast; plast = t_; } while (0)
#else
#define POSE_STAMP0() do { } while (0)
#define POSE_STAMP(k) do { } while (0)
#endif
    // the LM iterations of one round over the active matches (level1 == 0), from and into `pose`
    auto lm_round = [&](bool robust, int its) {
        double cnt[1] = { 0.0 };
        for (int i = tid; i < p.n; i += kT) cnt[0] += level1[i] ? 0.0 : 1.0;
        reduce_all<1>(cnt, lds, flip);
        bool ok = cnt[0] > 0.0;
        double lambda = 0.0, ni = 2.0;
        for (int it = 0; it < its && ok; ++it) {
            ++n_lm;
            POSE_STAMP0();
            // computeActiveErrors + buildSystem in one pass
            double R[12];
            q2R(pose, R);
            double acc[28];
#pragma unroll
            for (int k = 0; k < 28; ++k) acc[k] = 0.0;
            for (int i0 = tid; i0 < p.n; i0 += kFly * kT) {
              Batch b;
              fetch(i0, b);
              const int w0 = __builtin_amdgcn_readfirstlane(i0);
#pragma unroll
              for (int u = 0; u < kFly; ++u) {
                if (w0 + u * kT >= p.n) break;
                const double X0 = b.X[u][0], X1 = b.X[u][1], X2 = b.X[u][2];
                const bool in = b.in[u];
                const double x = in ? R[0] * X0 + R[1] * X1 + R[2] * X2 + R[9] : 0.0;
                const double y = in ? R[3] * X0 + R[4] * X1 + R[5] * X2 + R[10] : 0.0;
                const double z = in ? R[6] * X0 + R[7] * X1 + R[8] * X2 + R[11] : 1.0;
                const double om = in ? b.om[u] : 0.0;
                // (one reciprocal per match and pass instead of six divisions)
                const double iz = fast_rcp(z), uu = p.fx * x * iz, vv = p.fy * y * iz;
                const double e0 = b.o[u][0] - (uu + p.cx), e1 = b.o[u][1] - (vv + p.cy);
                const double chi2 = e0 * (om * e0) + e1 * (om * e1);
                double rho0 = chi2, rho1 = 1.0;
                if (robust && p.huber_delta > 0.0 && !(chi2 <= dsqr)) {
                    const double rs = rsqrt(chi2), sq = chi2 > DBL_MAX ? chi2 : chi2 * rs;       // (sqrt(inf) = inf, not inf * 0)
                    rho0 = 2.0 * sq * p.huber_delta - dsqr; rho1 = p.huber_delta * rs;
                }
                const double wg = rho1 * om, r0 = -wg * e0, r1 = -wg * e1;
                const double a00 = -(p.fx * iz), a02 = uu * iz, a11 = -(p.fy * iz), a12 = vv * iz;
                const double C0[6] = { a02 * y, a00 * z - a02 * x, -a00 * y, a00, 0.0, a02 };
                const double C1[6] = { -a11 * z + a12 * y, -a12 * x, a11 * x, 0.0, a11, a12 };
                // H += J^T (wg J), b += J^T r: the weight taken into one factor once (D = wg C), every entry two fused
                // multiply-adds, and the terms of the structural zeros C0[4] = C1[3] = 0 left out (they add exact zeros):
                // 50 operations per match where `acc += wg * (C0 C0 + C1 C1)` over all 21 + 6 entries took 75
                double D0[6], D1[6];
#pragma unroll
                for (int a = 0; a < 6; ++a) { D0[a] = a == 4 ? 0.0 : wg * C0[a]; D1[a] = a == 3 ? 0.0 : wg * C1[a]; }
                int ue = 0;
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    if (a != 4) acc[21 + a] = __builtin_fma(C0[a], r0, acc[21 + a]);
                    if (a != 3) acc[21 + a] = __builtin_fma(C1[a], r1, acc[21 + a]);
#pragma unroll
                    for (int c = a; c < 6; ++c, ++ue) {
                        if (a != 4 && c != 4) acc[ue] = __builtin_fma(D0[a], C0[c], acc[ue]);
                        if (a != 3 && c != 3) acc[ue] = __builtin_fma(D1[a], C1[c], acc[ue]);
                    }
                }
                acc[27] += rho0;
              }
            }
            POSE_STAMP(0);
            reduce_all<28>(acc, lds, flip);
            POSE_STAMP(1);
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
                POSE_STAMP(2);
                double F1 = cost(pose, robust);
                POSE_STAMP(3);
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
                    const double tr = 2.0 * rho - 1.0;
                    double alpha = 1.0 - tr * tr * tr;           // (pow(tmp, 3) in g2o)
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
    };
    // a pose at the hypothesis threshold: matches inside it and sigma-consensus++ loss (the scores of the hypothesis stage); mark:
    // also the matches outside (level1) and the IRLS weights of those inside (wls)
    auto score_pose = [&](const double T[7], double &cnt_o, double &cst_o, bool mark) {
        const Magsac ms(p.chi2_gate);
        double R[12];
        q2R(T, R);
        double sc[2] = { 0.0, 0.0 };
        for (int i = tid; i < p.n; i += kT) {
            const double X0 = Xw[3 * i], X1 = Xw[3 * i + 1], X2 = Xw[3 * i + 2];
            const double x = R[0] * X0 + R[1] * X1 + R[2] * X2 + R[9];
            const double y = R[3] * X0 + R[4] * X1 + R[5] * X2 + R[10];
            const double z = R[6] * X0 + R[7] * X1 + R[8] * X2 + R[11];
            const double om = isig[i];
            const double e0 = obs[2 * i] - (p.fx * x / z + p.cx), e1 = obs[2 * i + 1] - (p.fy * y / z + p.cy);
            const double chi2 = e0 * (om * e0) + e1 * (om * e1);
            const bool in = (z > 0.0) && (chi2 <= p.chi2_gate);
            double ls, wt;
            ms.terms(chi2, z > 0.0, p.chi2_gate, ls, wt);
            sc[0] += in ? 1.0 : 0.0; sc[1] += ls;
            if (mark) { level1[i] = in ? 0 : 1; wls[i] = wt; }
        }
        reduce_all<2>(sc, lds, flip);
        cnt_o = sc[0]; cst_o = sc[1];
    };
    // ---- local optimisation of the winning hypothesis (USAC's LO step; MAGSAC++'s model polishing is iteratively reweighted least
    // squares): LM on the matches inside the threshold, each weighted by its sigma-consensus weight at the winner, no robust
    // kernel; kept when the refit pose has a lower loss.  The four rounds below then start from that pose, over all matches. ----
    if (have_hyp && p.lo_its > 0) {
        double c0, s0, c1, s1;
        __syncthreads();
        score_pose(pose0, c0, s0, true);
        __syncthreads();
        wuse = wls;
        lm_round(false, p.lo_its);
        wuse = nullptr;
        score_pose(pose, c1, s1, false);
        const bool keep = c1 > 3.5 && s1 < s0;
        if (keep) {
#pragma unroll
            for (int k = 0; k < 7; ++k) pose0[k] = pose[k];
        }
        __syncthreads();
        for (int i = tid; i < p.n; i += kT) level1[i] = 0;
        if (tid == 0) {
            p.pose_out[18] = keep ? 1.0 : 0.0; p.pose_out[19] = keep ? c1 : c0;
#pragma unroll
            for (int k = 0; k < 7; ++k) p.pose_out[9 + k] = pose0[k];
        }
        __syncthreads();
    }
    for (int round = 0; round < p.rounds; ++round) {
        const bool robust = round <= 2;
#pragma unroll
        for (int k = 0; k < 7; ++k) pose[k] = pose0[k];
        lm_round(robust, p.its);
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
        reduce_all<1>(nb, lds, flip);
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
        p.pose_out[16] = (double)n_lm;
#ifdef MOVBA_CLOCK_STAMP
        for (int k = 0; k < 4; ++k) p.pose_out[20 + k] = (double)pseg[k];
#endif
    }
}

size_t pose_ransac_bytes(int n_hyp) { return n_hyp > 0 ? (size_t)n_hyp * (48 + 8) * sizeof(double) + (size_t)n_hyp * sizeof(int) + 16 : 0; }
size_t pose_opt_staged_lds_bytes(int n, int n_hyp) { return (size_t)n * 7 * sizeof(double) + (((size_t)n + 15) & ~(size_t)15) + pose_ransac_bytes(n_hyp); }

hipError_t configure_pose_kernels()
{
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k_pose_opt<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
}

hipError_t launch_pose_hyp(const PoseDev &p, hipStream_t s)
{
    if (p.n_hyp > 0) hipLaunchKernelGGL(k_pose_hyp, dim3(p.n_hyp), dim3(kT), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_pose_opt(const PoseDev &p, bool staged, hipStream_t s)
{
    if (staged) hipLaunchKernelGGL(k_pose_opt<true>, dim3(1), dim3(kT), pose_opt_staged_lds_bytes(p.n, p.hyp_done ? 0 : p.n_hyp), s, p);
    else hipLaunchKernelGGL(k_pose_opt<false>, dim3(1), dim3(kT), 0, s, p);
    return hipGetLastError();
}

}  // namespace movba
