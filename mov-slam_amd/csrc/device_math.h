// Small fp64 device helpers shared by the HIP kernels: SE3Quat arithmetic as g2o defines it
// (normalizeRotation, operator*, exp — SURVEY.md Appendix A.8), Eigen's quaternion <-> matrix
// conversions, the cofactor 3x3 inverse, and fixed-order wave / block reductions.
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>

namespace movba {

__device__ __forceinline__ void quat_to_R(const double q[4], double R[9])
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
    R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

// 1 / p by v_rcp_f64 and two Newton steps (~1 ulp; no special cases: p finite, non-zero, normal).  The IEEE division the
// compiler emits for `1.0 / p` is a ~25-instruction dependent sequence (scale, reciprocal, four refinements, fix-up);
// the per-edge and per-entry arithmetic of the point and schur kernels held three to six of them.
__device__ __forceinline__ double fast_rcp(double p)
{
    double r = __builtin_amdgcn_rcp(p);
    r = r * (2.0 - p * r);
    r = r * (2.0 - p * r);
    return r;
}

// ... and where a zero can come in (the depth of a map point lying exactly on a keyframe's z = 0 plane): 1 / 0 = inf like the
// IEEE division of Pinhole::project (src/CameraModels/Pinhole.cpp:36-43), so that the edge's chi2 — and with it the robust
// cost the LM decides on — is inf as in the reference, not NaN (the Newton step alone makes inf * (2 - 0 * inf) = NaN)
__device__ __forceinline__ double fast_rcp_zero_safe(double p)
{
    const double r0 = __builtin_amdgcn_rcp(p);
    double r = r0 * (2.0 - p * r0);
    r = r * (2.0 - p * r);
    return p == 0.0 ? r0 : r;
}

__device__ __forceinline__ void R_to_quat(const double m[9], double q[4])
{
    double t = m[0] + m[4] + m[8];
    if (t > 0.0) {
        const double rs = rsqrt(t + 1.0);
        q[3] = 0.5 * ((t + 1.0) * rs);
        t = 0.5 * rs;
        q[0] = (m[7] - m[5]) * t;
        q[1] = (m[2] - m[6]) * t;
        q[2] = (m[3] - m[1]) * t;
    } else {
        // i = argmax diagonal, written without dynamic register indexing
        if (m[0] >= m[4] && m[0] >= m[8]) {
            const double v = m[0] - m[4] - m[8] + 1.0, rs = rsqrt(v);
            q[0] = 0.5 * (v * rs); t = 0.5 * rs;
            q[3] = (m[7] - m[5]) * t; q[1] = (m[3] + m[1]) * t; q[2] = (m[6] + m[2]) * t;
        } else if (m[4] > m[0] && m[4] >= m[8]) {
            const double v = m[4] - m[8] - m[0] + 1.0, rs = rsqrt(v);
            q[1] = 0.5 * (v * rs); t = 0.5 * rs;
            q[3] = (m[2] - m[6]) * t; q[2] = (m[7] + m[5]) * t; q[0] = (m[1] + m[3]) * t;
        } else {
            const double v = m[8] - m[0] - m[4] + 1.0, rs = rsqrt(v);
            q[2] = 0.5 * (v * rs); t = 0.5 * rs;
            q[3] = (m[3] - m[1]) * t; q[0] = (m[2] + m[6]) * t; q[1] = (m[5] + m[7]) * t;
        }
    }
}

// SE3Quat::normalizeRotation with the reference's own operations (square root, four divisions): where a pose is taken over
// from the caller — a unit quaternion comes out bit for bit as it went in, so a keyframe nothing moves is returned unchanged
__device__ __forceinline__ void quat_normalize_exact(double q[4])
{
    if (q[3] < 0.0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

// the same inside the update chain (reciprocal square root and multiplications, ~1 ulp apart)
__device__ __forceinline__ void quat_normalize(double q[4])
{
    if (q[3] < 0.0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const double in = rsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] *= in; q[1] *= in; q[2] *= in; q[3] *= in;
}

__device__ __forceinline__ void quat_rotate(const double q[4], const double v[3], double o[3])
{
    double ux = q[1] * v[2] - q[2] * v[1], uy = q[2] * v[0] - q[0] * v[2], uz = q[0] * v[1] - q[1] * v[0];
    ux += ux; uy += uy; uz += uz;
    o[0] = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
    o[1] = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
    o[2] = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
}

// T <- exp(u) * T   (VertexSE3Expmap::oplusImpl; u = (omega, upsilon), rotation first)
__device__ inline void se3_oplus(const double u[6], const double T[7], double out[7])
{
    const double wx = u[0], wy = u[1], wz = u[2];
    const double th2 = wx * wx + wy * wy + wz * wz;
    double a, b, c, d;
    if (th2 < 1e-10) { a = 1.0; b = 0.5; c = 0.5; d = 1.0 / 6.0; }       // (theta < 0.00001, as g2o's SE3Quat::exp)
    else if (th2 < 0.25) {
        // sin t / t, (1 - cos t) / t^2 and (t - sin t) / t^3 as their power series in t^2, nine terms each (the next one is below
        // 3e-23 for t < 0.5: every LM step of a window or a frame that is being refined): three chains of eight fused
        // multiply-adds that run side by side, no square root, no reciprocal, no argument reduction.  sincos() + the reciprocal
        // were a chain of ~90 dependent instructions - the tail of every reduced solve's launch and of every pose-only LM trial,
        // where a lone wave pays ~35 cycles for each.  (The closed forms below agree with the series to rounding.)
        const double t = th2;
        a = 1.0 / 355687428096000.0; b = 1.0 / 6402373705728000.0; d = 1.0 / 121645100408832000.0;
        a = __builtin_fma(a, -t, 1.0 / 1307674368000.0); b = __builtin_fma(b, -t, 1.0 / 20922789888000.0); d = __builtin_fma(d, -t, 1.0 / 355687428096000.0);
        a = __builtin_fma(a, -t, 1.0 / 6227020800.0);    b = __builtin_fma(b, -t, 1.0 / 87178291200.0);    d = __builtin_fma(d, -t, 1.0 / 1307674368000.0);
        a = __builtin_fma(a, -t, 1.0 / 39916800.0);      b = __builtin_fma(b, -t, 1.0 / 479001600.0);      d = __builtin_fma(d, -t, 1.0 / 6227020800.0);
        a = __builtin_fma(a, -t, 1.0 / 362880.0);        b = __builtin_fma(b, -t, 1.0 / 3628800.0);        d = __builtin_fma(d, -t, 1.0 / 39916800.0);
        a = __builtin_fma(a, -t, 1.0 / 5040.0);          b = __builtin_fma(b, -t, 1.0 / 40320.0);          d = __builtin_fma(d, -t, 1.0 / 362880.0);
        a = __builtin_fma(a, -t, 1.0 / 120.0);           b = __builtin_fma(b, -t, 1.0 / 720.0);            d = __builtin_fma(d, -t, 1.0 / 5040.0);
        a = __builtin_fma(a, -t, 1.0 / 6.0);             b = __builtin_fma(b, -t, 1.0 / 24.0);             d = __builtin_fma(d, -t, 1.0 / 120.0);
        a = __builtin_fma(a, -t, 1.0);                   b = __builtin_fma(b, -t, 0.5);                    d = __builtin_fma(d, -t, 1.0 / 6.0);
        c = b;
    } else {
        const double th = sqrt(th2);
        double sn, cs;
        sincos(th, &sn, &cs);
        const double ith = fast_rcp(th), ith2 = ith * ith;
        a = sn * ith; b = (1.0 - cs) * ith2; c = b; d = (th - sn) * (ith2 * ith);
    }
    // Om = [w]x ; Om2 = w w^T - th2 I
    const double Om[9] = { 0.0, -wz, wy, wz, 0.0, -wx, -wy, wx, 0.0 };
    const double Om2[9] = { wx * wx - th2, wx * wy, wx * wz, wy * wx, wy * wy - th2, wy * wz, wz * wx, wz * wy, wz * wz - th2 };
    double R[9], V[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const double I = (i % 4 == 0) ? 1.0 : 0.0;
        R[i] = I + a * Om[i] + b * Om2[i];
        V[i] = I + c * Om[i] + d * Om2[i];
    }
    double e[7];
    R_to_quat(R, e);
    e[4] = V[0] * u[3] + V[1] * u[4] + V[2] * u[5];
    e[5] = V[3] * u[3] + V[4] * u[4] + V[5] * u[5];
    e[6] = V[6] * u[3] + V[7] * u[4] + V[8] * u[5];
    quat_normalize(e);
    // SE3Quat::operator*
    double r[4];
    r[3] = e[3] * T[3] - e[0] * T[0] - e[1] * T[1] - e[2] * T[2];
    r[0] = e[3] * T[0] + e[0] * T[3] + e[1] * T[2] - e[2] * T[1];
    r[1] = e[3] * T[1] + e[1] * T[3] + e[2] * T[0] - e[0] * T[2];
    r[2] = e[3] * T[2] + e[2] * T[3] + e[0] * T[1] - e[1] * T[0];
    double rt[3];
    quat_rotate(e, T + 4, rt);
    quat_normalize(r);
    out[0] = r[0]; out[1] = r[1]; out[2] = r[2]; out[3] = r[3];
    out[4] = e[4] + rt[0]; out[5] = e[5] + rt[1]; out[6] = e[6] + rt[2];
}

// inverse of the symmetric 3x3 (xx xy xz yy yz zz) by cofactors
__device__ __forceinline__ void inv3sym(const double A[6], double B[6])
{
    const double c00 = A[3] * A[5] - A[4] * A[4];
    const double c01 = A[4] * A[2] - A[1] * A[5];
    const double c02 = A[1] * A[4] - A[3] * A[2];
    const double id = fast_rcp(A[0] * c00 + A[1] * c01 + A[2] * c02);
    B[0] = c00 * id; B[1] = c01 * id; B[2] = c02 * id;
    B[3] = (A[0] * A[5] - A[2] * A[2]) * id;
    B[4] = (A[2] * A[1] - A[0] * A[4]) * id;
    B[5] = (A[0] * A[3] - A[1] * A[1]) * id;
}

// ---- DPP cross-lane moves (no LDS crossbar, unlike ds_bpermute-based __shfl) ----
// ctrl: quad_perm 0x00-0xff, row_shr:n 0x110+n, row_mirror 0x140, row_half_mirror 0x141,
//       row_bcast:15 0x142, row_bcast:31 0x143 (gfx9 encodings)
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ double dpp_mov0(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, BANK_MASK, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, BANK_MASK, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

// a wave-uniform double moved to SGPRs (frees two VGPRs per value; VALU ops take it as a scalar operand)
__device__ __forceinline__ double uniform_f64(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// Fixed-tree sum over the 64 lanes, result broadcast to every lane (wave-uniform).
__device__ __forceinline__ double wave_sum_dpp(double v)
{
    v += dpp_mov0<0xb1>(v);             // quad_perm [1,0,3,2]
    v += dpp_mov0<0x4e>(v);             // quad_perm [2,3,0,1]
    v += dpp_mov0<0x141>(v);            // row_half_mirror: sums of 8
    v += dpp_mov0<0x140>(v);            // row_mirror: every lane holds its row-of-16 sum
    v += dpp_mov0<0x142, 0xa>(v);       // row_bcast:15 into rows 1 and 3
    v += dpp_mov0<0x143, 0xc>(v);       // row_bcast:31 into rows 2 and 3
    return readlane_f64(v, 63);
}

// Two independent sums at once (their DPP steps interleave, hiding each other's latency); results in (a, b).
__device__ __forceinline__ void wave_sum_dpp2(double &a, double &b)
{
    a += dpp_mov0<0xb1>(a);             b += dpp_mov0<0xb1>(b);
    a += dpp_mov0<0x4e>(a);             b += dpp_mov0<0x4e>(b);
    a += dpp_mov0<0x141>(a);            b += dpp_mov0<0x141>(b);
    a += dpp_mov0<0x140>(a);            b += dpp_mov0<0x140>(b);
    a += dpp_mov0<0x142, 0xa>(a);       b += dpp_mov0<0x142, 0xa>(b);
    a += dpp_mov0<0x143, 0xc>(a);       b += dpp_mov0<0x143, 0xc>(b);
    a = readlane_f64(a, 63);            b = readlane_f64(b, 63);
}

// Sum over aligned groups of G adjacent lanes (G = 1, 2, 4 or 8), valid in every lane of the group.
__device__ __forceinline__ double group_sum_dpp(double v, int G)
{
    if (G >= 2) v += dpp_mov0<0xb1>(v);
    if (G >= 4) v += dpp_mov0<0x4e>(v);
    if (G >= 8) v += dpp_mov0<0x141>(v);
    return v;
}

// (DPP trees: a butterfly of 64-bit __shfl_xor costs two ds_bpermute through the LDS crossbar per step)
__device__ __forceinline__ double wave_sum(double v) { return wave_sum_dpp(v); }

__device__ __forceinline__ double wave_max(double v)
{
    v = fmax(v, dpp_mov0<0xb1>(v));
    v = fmax(v, dpp_mov0<0x4e>(v));
    v = fmax(v, dpp_mov0<0x141>(v));
    v = fmax(v, dpp_mov0<0x140>(v));
    // (the lanes the row broadcasts do not write receive 0: harmless for the maxima of magnitudes taken here, and lane 63,
    // which is read, is written by both)
    v = fmax(v, dpp_mov0<0x142, 0xa>(v));
    v = fmax(v, dpp_mov0<0x143, 0xc>(v));
    return readlane_f64(v, 63);
}

// deterministic block reduction (fixed order), result valid in every thread
template <int NWAVES, bool MAX>
__device__ __forceinline__ double block_reduce(double v, double *red /* NWAVES doubles in LDS */)
{
    v = MAX ? wave_max(v) : wave_sum(v);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    double s = red[0];
#pragma unroll
    for (int k = 1; k < NWAVES; ++k) s = MAX ? fmax(s, red[k]) : s + red[k];
    __syncthreads();
    return s;
}

// upper-triangle index of a symmetric 6x6, a <= b
__device__ __forceinline__ constexpr int ut6(int a, int b) { return a * 6 - a * (a - 1) / 2 + (b - a); }

}  // namespace movba
