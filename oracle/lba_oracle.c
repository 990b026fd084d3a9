/*
 * lba_oracle.c — CPU restatement of the reference's local-BA path (see lba_oracle.h:
 * TEST INFRASTRUCTURE ONLY, PARITY UNPINNED).
 *
 * Each function cites the reference file:line it follows, or, for the arithmetic that
 * the reference delegates to its un-vendored dependency g2o (master, unpinned,
 * reference Dockerfile:154), the g2o class whose published behaviour it restates.
 * Plain C99, no dependencies, single-threaded (g2o compiles its OpenMP pragmas out
 * by default and the reference adds none: SURVEY.md §8d).
 */
#include "lba_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* SE3Quat (g2o types/slam3d/se3quat.h): unit quaternion (x,y,z,w) + t       */
/* ------------------------------------------------------------------------- */

/* Eigen QuaternionBase::toRotationMatrix */
static void quat_to_R(const double q[4], double R[9])
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

/* Eigen quaternion from rotation matrix (internal::quaternionbase_assign_impl) */
static void R_to_quat(const double m[9], double q[4])
{
    double t = m[0] + m[4] + m[8];
    if (t > 0.0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m[7] - m[5]) * t;
        q[1] = (m[2] - m[6]) * t;
        q[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 3 + i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (m[k * 3 + j] - m[j * 3 + k]) * t;
        q[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        q[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
    }
}

/* SE3Quat::normalizeRotation: flip to w >= 0, then normalise */
void lba_oracle_se3_normalize(double qt[7])
{
    if (qt[3] < 0.0) { qt[0] = -qt[0]; qt[1] = -qt[1]; qt[2] = -qt[2]; qt[3] = -qt[3]; }
    const double n = sqrt(qt[0] * qt[0] + qt[1] * qt[1] + qt[2] * qt[2] + qt[3] * qt[3]);
    qt[0] /= n; qt[1] /= n; qt[2] /= n; qt[3] /= n;
}

/* Eigen QuaternionBase::_transformVector: v + w*uv + q.vec x uv, uv = 2 q.vec x v */
static void quat_rotate(const double q[4], const double v[3], double out[3])
{
    double uv[3] = { q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0] };
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    out[0] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
    out[1] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
    out[2] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}

/* SE3Quat::map(X) = _r * X + _t; the estimate is Tcw (src/Optimizer.cc:558-559) */
void lba_oracle_se3_map(const double qt[7], const double X[3], double Xc[3])
{
    quat_rotate(qt, X, Xc);
    Xc[0] += qt[4]; Xc[1] += qt[5]; Xc[2] += qt[6];
}

/* SE3Quat::operator*: r = a.r*b.r, t = a.t + a.r*b.t, then normalizeRotation */
void lba_oracle_se3_mul(const double a[7], const double b[7], double out[7])
{
    double r[7];
    r[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    r[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    r[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    r[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    double rt[3];
    quat_rotate(a, b + 4, rt);
    r[4] = a[4] + rt[0]; r[5] = a[5] + rt[1]; r[6] = a[6] + rt[2];
    lba_oracle_se3_normalize(r);
    memcpy(out, r, sizeof r);
}

static void mat3_mul(const double A[9], const double B[9], double C[9])
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C[i * 3 + j] = A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j] + A[i * 3 + 2] * B[2 * 3 + j];
}

/* SE3Quat::exp(update), update = (omega, upsilon): rotation first */
void lba_oracle_se3_exp(const double u[6], double out[7])
{
    const double wx = u[0], wy = u[1], wz = u[2];
    const double theta = sqrt(wx * wx + wy * wy + wz * wz);
    const double Om[9] = { 0.0, -wz, wy, wz, 0.0, -wx, -wy, wx, 0.0 };
    double Om2[9];
    mat3_mul(Om, Om, Om2);
    double R[9], V[9];
    double a, b, c, d;               /* R = I + a Om + b Om2 ; V = I + c Om + d Om2 */
    if (theta < 0.00001) {
        a = 1.0; b = 0.5; c = 0.5; d = 1.0 / 6.0;
    } else {
        a = sin(theta) / theta;
        b = (1.0 - cos(theta)) / (theta * theta);
        c = b;
        d = (theta - sin(theta)) / pow(theta, 3);
    }
    for (int i = 0; i < 9; ++i) {
        const double I = (i % 4 == 0) ? 1.0 : 0.0;
        R[i] = I + a * Om[i] + b * Om2[i];
        V[i] = I + c * Om[i] + d * Om2[i];
    }
    double r[7];
    R_to_quat(R, r);
    r[4] = V[0] * u[3] + V[1] * u[4] + V[2] * u[5];
    r[5] = V[3] * u[3] + V[4] * u[4] + V[5] * u[5];
    r[6] = V[6] * u[3] + V[7] * u[4] + V[8] * u[5];
    lba_oracle_se3_normalize(r);     /* SE3Quat(q, t) constructor normalises */
    memcpy(out, r, sizeof r);
}

/* ------------------------------------------------------------------------- */
/* Edge arithmetic                                                            */
/* ------------------------------------------------------------------------- */

/* EdgeSE3ProjectXYZ::computeError (include/OptimizableTypes.h:103-109) with
 * Pinhole::project (src/CameraModels/Pinhole.cpp:36-43): obs - (fx*x/z+cx, fy*y/z+cy) */
static inline void edge_error(const double Xc[3], const double obs[2], const double cam[4], double e[2])
{
    e[0] = obs[0] - (cam[0] * Xc[0] / Xc[2] + cam[2]);
    e[1] = obs[1] - (cam[1] * Xc[1] / Xc[2] + cam[3]);
}

/* EdgeSE3ProjectXYZ::linearizeOplus (src/OptimizableTypes.cpp:158-180) with
 * Pinhole::projectJac (src/CameraModels/Pinhole.cpp:77-88).
 * Jp = -Jpi * R  (2x3, vertex 0 = point); Jc = -Jpi * [ -[Xc]x | I ] (2x6, vertex 1 = pose) */
static inline void edge_jacobians(const double R[9], const double Xc[3], const double cam[4],
                                  double Jp[6], double Jc[12])
{
    const double x = Xc[0], y = Xc[1], z = Xc[2];
    /* -projectJac */
    const double a00 = -(cam[0] / z), a02 = -(-cam[0] * x / (z * z));
    const double a11 = -(cam[1] / z), a12 = -(-cam[1] * y / (z * z));
    for (int j = 0; j < 3; ++j) {
        Jp[j]     = a00 * R[0 * 3 + j] + a02 * R[2 * 3 + j];
        Jp[3 + j] = a11 * R[1 * 3 + j] + a12 * R[2 * 3 + j];
    }
    /* SE3deriv = [0 z -y 1 0 0; -z 0 x 0 1 0; y -x 0 0 0 1] */
    Jc[0] = a02 * y;             Jc[1] = a00 * z + a02 * (-x); Jc[2] = a00 * (-y);
    Jc[3] = a00;                 Jc[4] = 0.0;                  Jc[5] = a02;
    Jc[6] = a11 * (-z) + a12 * y; Jc[7] = a12 * (-x);          Jc[8] = a11 * x;
    Jc[9] = 0.0;                 Jc[10] = a11;                 Jc[11] = a12;
}

void lba_oracle_edge(const double qt[7], const double X[3], const double obs[2],
                     const double cam[4], double err[2], double Jp[6], double Jc[12])
{
    double Xc[3], R[9];
    lba_oracle_se3_map(qt, X, Xc);
    quat_to_R(qt, R);
    edge_error(Xc, obs, cam, err);
    edge_jacobians(R, Xc, cam, Jp, Jc);
}

/* g2o EdgeStereoSE3ProjectXYZ (types_six_dof_expmap; instantiated at src/Optimizer.cc:680): third residual
 * u_r - (fx x/z + cx - bf/z) and its Jacobian rows: row 2 of J_point = row 0 - bf R(2,:)/z^2, row 2 of
 * J_pose = row 0 + [-bf y/z^2, +bf x/z^2, 0, 0, 0, -bf/z^2]. */
static inline double stereo_error(const double Xc[3], double ur, const double cam[4], double bf)
{
    return ur - (cam[0] * Xc[0] / Xc[2] + cam[2] - bf / Xc[2]);
}

static inline void stereo_rows(const double R[9], const double Xc[3], double bf, const double Jp[6], const double Jc[12],
                               double Jp2[3], double Jc2[6])
{
    const double x = Xc[0], y = Xc[1], z = Xc[2], z2 = z * z;
    for (int j = 0; j < 3; ++j) Jp2[j] = Jp[j] - bf * R[6 + j] / z2;
    for (int j = 0; j < 6; ++j) Jc2[j] = Jc[j];
    Jc2[0] -= bf * y / z2; Jc2[1] += bf * x / z2; Jc2[5] -= bf / z2;
}

static inline int is_stereo(const lba_oracle_problem *pb, int e)
{
    return pb->obs_right && pb->obs_right[e] >= 0.0;
}

/* g2o RobustKernelHuber::robustify; delta = (double)sqrtf(5.0f) at src/Optimizer.cc:616, 660-662 */
void lba_oracle_huber(double e, double delta, double rho[3])
{
    const double dsqr = delta * delta;
    if (e <= dsqr) {
        rho[0] = e; rho[1] = 1.0; rho[2] = 0.0;
    } else {
        const double sqrte = sqrt(e);
        rho[0] = 2.0 * sqrte * delta - dsqr;
        rho[1] = delta / sqrte;
        rho[2] = -0.5 * rho[1] / e;
    }
}

/* Plain 3x3 inverse by cofactors (Eigen's fixed-size inverse, BlockSolver::solve Dinv) */
static void inv3(const double A[9], double B[9])
{
    const double c00 = A[4] * A[8] - A[5] * A[7];
    const double c01 = A[5] * A[6] - A[3] * A[8];
    const double c02 = A[3] * A[7] - A[4] * A[6];
    const double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
    const double id = 1.0 / det;
    B[0] = c00 * id; B[1] = (A[2] * A[7] - A[1] * A[8]) * id; B[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    B[3] = c01 * id; B[4] = (A[0] * A[8] - A[2] * A[6]) * id; B[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    B[6] = c02 * id; B[7] = (A[1] * A[6] - A[0] * A[7]) * id; B[8] = (A[0] * A[4] - A[1] * A[3]) * id;
}

/* Dense LL^T solve of the symmetric n x n system (stands in for LinearSolverCSparse at
 * src/Optimizer.cc:535: an exact fp64 Cholesky; ordering only changes round-off).
 * A is overwritten. Returns 0 when A is not positive definite. */
static int chol_solve(double *A, int n, const double *b, double *x)
{
    for (int j = 0; j < n; ++j) {
        double d = A[j * n + j];
        for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0.0) || !isfinite(d)) return 0;
        d = sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[i * n + j];
            const double *ri = A + i * n, *rj = A + j * n;
            for (int k = 0; k < j; ++k) s -= ri[k] * rj[k];
            A[i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= A[i * n + k] * x[k];
        x[i] = s / A[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = x[i];
        for (int k = i + 1; k < n; ++k) s -= A[k * n + i] * x[k];
        x[i] = s / A[i * n + i];
    }
    return 1;
}

/* ------------------------------------------------------------------------- */
/* Graph workspace (what g2o builds in initializeOptimization / buildStructure) */
/* ------------------------------------------------------------------------- */

typedef struct {
    const lba_oracle_problem *pb;
    int nfree;                 /* free AND active poses                               */
    int32_t *hidx;             /* pose -> hessian index or -1 (fixed or edge-less)    */
    int32_t *pt_start;         /* CSR over points: edges of point l                   */
    int32_t *pt_edges;
    uint8_t *pt_active;        /* point has >= 1 edge                                 */
    double cam[4];
    /* state */
    double *poses, *points;    /* current estimates                                   */
    double *poses_bk, *points_bk;
    double *err;               /* E x 2 stored _error                                 */
    /* system */
    double *Hpp, *bp;          /* nfree x 36, nfree x 6                               */
    double *Hll, *bl;          /* P x 9, P x 3                                        */
    double *Hpl;               /* E x 18 (6x3 row-major), valid where pose free        */
    double *S, *bS, *xp, *xl;  /* reduced system and increments                       */
    double *Dinv;              /* P x 9                                               */
} ws_t;

static void ws_free(ws_t *w)
{
    free(w->hidx); free(w->pt_start); free(w->pt_edges); free(w->pt_active);
    free(w->poses); free(w->points); free(w->poses_bk); free(w->points_bk); free(w->err);
    free(w->Hpp); free(w->bp); free(w->Hll); free(w->bl); free(w->Hpl);
    free(w->S); free(w->bS); free(w->xp); free(w->xl); free(w->Dinv);
}

/* SparseOptimizer::initializeOptimization: active vertices are those with
 * >= 1 edge; free non-marginalised vertices (poses) are indexed first in ascending id
 * — the caller supplies poses in ascending KeyFrame::mnId — then the points. */
static void ws_init(ws_t *w, const lba_oracle_problem *pb)
{
    memset(w, 0, sizeof *w);
    w->pb = pb;
    const int NP = pb->n_poses, P = pb->n_points, E = pb->n_edges;
    w->cam[0] = pb->fx; w->cam[1] = pb->fy; w->cam[2] = pb->cx; w->cam[3] = pb->cy;
    w->hidx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(NP > 0 ? NP : 1));
    uint8_t *pose_active = (uint8_t *)calloc((size_t)(NP > 0 ? NP : 1), 1);
    w->pt_active = (uint8_t *)calloc((size_t)(P > 0 ? P : 1), 1);
    w->pt_start = (int32_t *)calloc((size_t)P + 2, sizeof(int32_t));
    w->pt_edges = (int32_t *)malloc(sizeof(int32_t) * (size_t)(E > 0 ? E : 1));
    for (int e = 0; e < E; ++e) {
        pose_active[pb->edge_pose[e]] = 1;
        w->pt_active[pb->edge_point[e]] = 1;
        w->pt_start[pb->edge_point[e] + 2]++;
    }
    for (int l = 0; l < P; ++l) w->pt_start[l + 2] += w->pt_start[l + 1];
    for (int e = 0; e < E; ++e) w->pt_edges[w->pt_start[pb->edge_point[e] + 1]++] = e;
    int nf = 0;
    for (int i = 0; i < NP; ++i) w->hidx[i] = (!pb->pose_fixed[i] && pose_active[i]) ? nf++ : -1;
    w->nfree = nf;
    free(pose_active);

    w->poses = (double *)malloc(sizeof(double) * 7 * (size_t)(NP + 1));
    w->poses_bk = (double *)malloc(sizeof(double) * 7 * (size_t)(NP + 1));
    w->points = (double *)malloc(sizeof(double) * 3 * (size_t)(P + 1));
    w->points_bk = (double *)malloc(sizeof(double) * 3 * (size_t)(P + 1));
    memcpy(w->poses, pb->poses, sizeof(double) * 7 * (size_t)NP);
    memcpy(w->points, pb->points, sizeof(double) * 3 * (size_t)P);
    /* g2o::SE3Quat(q, t) normalises on construction (src/Optimizer.cc:559) */
    for (int i = 0; i < NP; ++i) lba_oracle_se3_normalize(w->poses + 7 * i);
    w->err = (double *)calloc(3 * (size_t)E + 3, sizeof(double));
    w->Hpp = (double *)calloc(36 * (size_t)nf + 36, sizeof(double));
    w->bp = (double *)calloc(6 * (size_t)nf + 6, sizeof(double));
    w->Hll = (double *)calloc(9 * (size_t)P + 9, sizeof(double));
    w->bl = (double *)calloc(3 * (size_t)P + 3, sizeof(double));
    w->Hpl = (double *)calloc(18 * (size_t)E + 18, sizeof(double));
    w->S = (double *)calloc((size_t)(6 * nf) * (size_t)(6 * nf) + 1, sizeof(double));
    w->bS = (double *)calloc(6 * (size_t)nf + 6, sizeof(double));
    w->xp = (double *)calloc(6 * (size_t)nf + 6, sizeof(double));
    w->xl = (double *)calloc(3 * (size_t)P + 3, sizeof(double));
    w->Dinv = (double *)calloc(9 * (size_t)P + 9, sizeof(double));
}

/* the camera of edge e: its keyframe's (src/Optimizer.cc:664, 690-695) */
static inline const double *edge_cam(const ws_t *w, int e)
{
    const lba_oracle_problem *pb = w->pb;
    return pb->cam_kf ? pb->cam_kf + 4 * (size_t)pb->edge_pose[e] : w->cam;
}
static inline double edge_bf(const lba_oracle_problem *pb, int e) { return pb->bf_kf ? pb->bf_kf[pb->edge_pose[e]] : pb->bf; }

/* computeActiveErrors + activeRobustChi2 */
static double ws_errors(ws_t *w)
{
    const lba_oracle_problem *pb = w->pb;
    double F = 0.0;
#ifdef _OPENMP          /* all-cores variant (liblba_oracle_omp.so, timing context only): the sum order differs */
#pragma omp parallel for reduction(+ : F) schedule(static)
#endif
    for (int e = 0; e < pb->n_edges; ++e) {
        double Xc[3];
        lba_oracle_se3_map(w->poses + 7 * pb->edge_pose[e], w->points + 3 * pb->edge_point[e], Xc);
        edge_error(Xc, pb->obs + 2 * e, edge_cam(w, e), w->err + 3 * e);
        double *er = w->err + 3 * e;
        er[2] = is_stereo(pb, e) ? stereo_error(Xc, pb->obs_right[e], edge_cam(w, e), edge_bf(pb, e)) : 0.0;
        const double chi2 = pb->inv_sigma2[e] * (er[0] * er[0] + er[1] * er[1] + er[2] * er[2]);
        if (pb->huber_delta > 0.0) {
            double rho[3];
            lba_oracle_huber(chi2, pb->huber_delta, rho);
            F += rho[0];
        } else {
            F += chi2;
        }
    }
    return F;
}

/* BlockSolver<6,3>::buildSystem: per edge linearizeOplus + constructQuadraticForm
 * (g2o BaseBinaryEdge), sequential in edge order. */
/* one edge of buildSystem; Hpp / bp are where the pose blocks accumulate (the workspace's, or a thread's own copy in the
 * OpenMP variant) */
static inline void ws_build_edge(ws_t *w, int e, double *Hpp, double *bp)
{
    const lba_oracle_problem *pb = w->pb;
    {
        const int ip = pb->edge_pose[e], l = pb->edge_point[e];
        const double *qt = w->poses + 7 * ip;
        double Xc[3], R[9], A[6], B[12];
        lba_oracle_se3_map(qt, w->points + 3 * l, Xc);
        quat_to_R(qt, R);
        edge_jacobians(R, Xc, edge_cam(w, e), A, B);
        double A2[3] = { 0, 0, 0 }, B2[6] = { 0, 0, 0, 0, 0, 0 };      /* third row: stereo edges only */
        if (is_stereo(pb, e)) stereo_rows(R, Xc, edge_bf(pb, e), A, B, A2, B2);
        const double *er = w->err + 3 * e;
        const double om = pb->inv_sigma2[e];
        double wgt = 1.0;
        if (pb->huber_delta > 0.0) {
            double rho[3];
            lba_oracle_huber(om * (er[0] * er[0] + er[1] * er[1] + er[2] * er[2]), pb->huber_delta, rho);
            wgt = rho[1];
        }
        const double wo = wgt * om;                      /* robustInformation = rho1 * Omega */
        const double r0 = -om * er[0] * wgt, r1 = -om * er[1] * wgt, r2 = -om * er[2] * wgt;   /* omega_r *= rho1 */
        /* point vertex (never fixed) */
        double *Hl = w->Hll + 9 * l, *b_l = w->bl + 3 * l;
        for (int a = 0; a < 3; ++a) {
            b_l[a] += A[a] * r0 + A[3 + a] * r1 + A2[a] * r2;
            for (int c = 0; c < 3; ++c) Hl[a * 3 + c] += wo * (A[a] * A[c] + A[3 + a] * A[3 + c] + A2[a] * A2[c]);
        }
        const int hi = w->hidx[ip];
        if (hi >= 0) {
            double *Hp = Hpp + 36 * hi, *b_p = bp + 6 * hi, *Hx = w->Hpl + 18 * e;
            for (int a = 0; a < 6; ++a) {
                b_p[a] += B[a] * r0 + B[6 + a] * r1 + B2[a] * r2;
                for (int c = 0; c < 6; ++c) Hp[a * 6 + c] += wo * (B[a] * B[c] + B[6 + a] * B[6 + c] + B2[a] * B2[c]);
                for (int c = 0; c < 3; ++c) Hx[a * 3 + c] = wo * (B[a] * A[c] + B[6 + a] * A[3 + c] + B2[a] * A2[c]);
            }
        }
    }
}

static void ws_build(ws_t *w)
{
    const lba_oracle_problem *pb = w->pb;
    const int nf = w->nfree, P = pb->n_points;
    memset(w->Hpp, 0, sizeof(double) * 36 * (size_t)nf);
    memset(w->bp, 0, sizeof(double) * 6 * (size_t)nf);
    memset(w->Hll, 0, sizeof(double) * 9 * (size_t)P);
    memset(w->bl, 0, sizeof(double) * 3 * (size_t)P);
#ifdef _OPENMP
    /* edge-parallel by map point (a point's block is touched by its own edges only); pose blocks go to per-thread copies */
#pragma omp parallel
    {
        double *tH = (double *)calloc(36 * (size_t)nf + 36, sizeof(double)), *tb = (double *)calloc(6 * (size_t)nf + 6, sizeof(double));
#pragma omp for schedule(static)
        for (int l = 0; l < P; ++l)
            for (int sidx = w->pt_start[l]; sidx < w->pt_start[l + 1]; ++sidx) ws_build_edge(w, w->pt_edges[sidx], tH, tb);
#pragma omp critical
        {
            for (int k = 0; k < 36 * nf; ++k) w->Hpp[k] += tH[k];
            for (int k = 0; k < 6 * nf; ++k) w->bp[k] += tb[k];
        }
        free(tH); free(tb);
    }
#else
    for (int e = 0; e < pb->n_edges; ++e) ws_build_edge(w, e, w->Hpp, w->bp);
#endif
}

static void ws_schur_point(ws_t *w, int l, double lambda, double *S, double *coef);
static int ws_solve_tail(ws_t *w, int keep_S);

/* BlockSolver::solve with Schur complement + exact reduced solve + back-substitution
 * (SURVEY A.7). lambda is added to every diagonal scalar of Hpp and Hll (setLambda). */
static int ws_solve(ws_t *w, double lambda, int keep_S)
{
    const lba_oracle_problem *pb = w->pb;
    const int nf = w->nfree, n = 6 * nf, P = pb->n_points;
    double *S = w->S;
    memset(S, 0, sizeof(double) * (size_t)n * (size_t)n);
    for (int i = 0; i < nf; ++i)
        for (int a = 0; a < 6; ++a)
            for (int c = 0; c < 6; ++c)
                S[(6 * i + a) * n + 6 * i + c] = w->Hpp[36 * i + a * 6 + c] + (a == c ? lambda : 0.0);
    double *coef = (double *)calloc((size_t)n + 1, sizeof(double));
#ifdef _OPENMP
#pragma omp parallel
    {
        /* per-thread copies of the reduced matrix and of B Dinv b_l, summed afterwards */
        double *tS = (double *)calloc((size_t)n * (size_t)n + 1, sizeof(double)), *tc = (double *)calloc((size_t)n + 1, sizeof(double));
#pragma omp for schedule(static)
        for (int l = 0; l < P; ++l) ws_schur_point(w, l, lambda, tS, tc);
#pragma omp critical
        {
            for (size_t k = 0; k < (size_t)n * (size_t)n; ++k) S[k] += tS[k];
            for (int k = 0; k < n; ++k) coef[k] += tc[k];
        }
        free(tS); free(tc);
    }
#else
    for (int l = 0; l < P; ++l) ws_schur_point(w, l, lambda, S, coef);
#endif
    for (int i = 0; i < n; ++i) w->bS[i] = w->bp[i] - coef[i];
    free(coef);
    return ws_solve_tail(w, keep_S);
}

/* Schur-complement contribution of map point l (and Dinv_l, kept for the back-substitution) */
static void ws_schur_point(ws_t *w, int l, double lambda, double *S, double *coef)
{
    const lba_oracle_problem *pb = w->pb;
    const int n = 6 * w->nfree;
    {
        if (!w->pt_active[l]) return;
        double D[9];
        memcpy(D, w->Hll + 9 * l, sizeof D);
        D[0] += lambda; D[4] += lambda; D[8] += lambda;
        double *Di = w->Dinv + 9 * l;
        inv3(D, Di);
        const double *b_l = w->bl + 3 * l;
        const double db[3] = { Di[0] * b_l[0] + Di[1] * b_l[1] + Di[2] * b_l[2],
                               Di[3] * b_l[0] + Di[4] * b_l[1] + Di[5] * b_l[2],
                               Di[6] * b_l[0] + Di[7] * b_l[1] + Di[8] * b_l[2] };
        for (int s = w->pt_start[l]; s < w->pt_start[l + 1]; ++s) {
            const int e1 = w->pt_edges[s], i1 = w->hidx[pb->edge_pose[e1]];
            if (i1 < 0) continue;
            const double *Bi = w->Hpl + 18 * e1;
            double BD[18];
            for (int a = 0; a < 6; ++a) {
                coef[6 * i1 + a] += Bi[a * 3] * db[0] + Bi[a * 3 + 1] * db[1] + Bi[a * 3 + 2] * db[2];
                for (int c = 0; c < 3; ++c)
                    BD[a * 3 + c] = Bi[a * 3] * Di[c] + Bi[a * 3 + 1] * Di[3 + c] + Bi[a * 3 + 2] * Di[6 + c];
            }
            for (int s2 = w->pt_start[l]; s2 < w->pt_start[l + 1]; ++s2) {
                const int e2 = w->pt_edges[s2], i2 = w->hidx[pb->edge_pose[e2]];
                if (i2 < 0) continue;
                /* g2o fills the upper block triangle; the mirrored lower blocks here make the
                 * dense symmetric matrix the Cholesky below factors */
                const double *Bj = w->Hpl + 18 * e2;
                for (int a = 0; a < 6; ++a)
                    for (int c = 0; c < 6; ++c)
                        S[(6 * i1 + a) * n + 6 * i2 + c] -=
                            BD[a * 3] * Bj[c * 3] + BD[a * 3 + 1] * Bj[c * 3 + 1] + BD[a * 3 + 2] * Bj[c * 3 + 2];
            }
        }
    }
}

/* exact reduced solve + back-substitution */
static int ws_solve_tail(ws_t *w, int keep_S)
{
    const lba_oracle_problem *pb = w->pb;
    const int nf = w->nfree, n = 6 * nf, P = pb->n_points;
    double *S = w->S;
    int ok = 1;
    if (n > 0) {
        double *Sw = S;
        if (keep_S) {
            Sw = (double *)malloc(sizeof(double) * (size_t)n * (size_t)n);
            memcpy(Sw, S, sizeof(double) * (size_t)n * (size_t)n);
        }
        ok = chol_solve(Sw, n, w->bS, w->xp);
        if (keep_S) free(Sw);
    }
    if (!ok) return 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int l = 0; l < P; ++l) {
        double *x = w->xl + 3 * l;
        x[0] = x[1] = x[2] = 0.0;
        if (!w->pt_active[l]) continue;
        double c[3] = { w->bl[3 * l], w->bl[3 * l + 1], w->bl[3 * l + 2] };
        for (int s = w->pt_start[l]; s < w->pt_start[l + 1]; ++s) {
            const int e1 = w->pt_edges[s], i1 = w->hidx[pb->edge_pose[e1]];
            if (i1 < 0) continue;
            const double *Bi = w->Hpl + 18 * e1, *xpi = w->xp + 6 * i1;
            for (int a = 0; a < 6; ++a)
                for (int k = 0; k < 3; ++k) c[k] -= Bi[a * 3 + k] * xpi[a];
        }
        const double *Di = w->Dinv + 9 * l;
        for (int k = 0; k < 3; ++k) x[k] = Di[k * 3] * c[0] + Di[k * 3 + 1] * c[1] + Di[k * 3 + 2] * c[2];
    }
    return 1;
}

/* SparseOptimizer::update: VertexSE3Expmap::oplusImpl  T <- exp(d) * T ;
 * VertexPointXYZ::oplusImpl  X <- X + d   */
static void ws_update(ws_t *w)
{
    const lba_oracle_problem *pb = w->pb;
    for (int i = 0; i < pb->n_poses; ++i) {
        const int hi = w->hidx[i];
        if (hi < 0) continue;
        double ex[7];
        lba_oracle_se3_exp(w->xp + 6 * hi, ex);
        lba_oracle_se3_mul(ex, w->poses + 7 * i, w->poses + 7 * i);
    }
    for (int l = 0; l < pb->n_points; ++l) {
        if (!w->pt_active[l]) continue;
        for (int k = 0; k < 3; ++k) w->points[3 * l + k] += w->xl[3 * l + k];
    }
}

int lba_oracle_linearize(const lba_oracle_problem *pb, double lambda,
                         double *Hpp, double *bp, double *Hll, double *bl,
                         double *S, double *bS, int32_t *free_index, double *F0)
{
    ws_t w;
    ws_init(&w, pb);
    const double F = ws_errors(&w);
    ws_build(&w);
    ws_solve(&w, lambda, 1);
    const int nf = w.nfree, n = 6 * nf;
    if (Hpp) memcpy(Hpp, w.Hpp, sizeof(double) * 36 * (size_t)nf);
    if (bp) memcpy(bp, w.bp, sizeof(double) * 6 * (size_t)nf);
    if (Hll) memcpy(Hll, w.Hll, sizeof(double) * 9 * (size_t)pb->n_points);
    if (bl) memcpy(bl, w.bl, sizeof(double) * 3 * (size_t)pb->n_points);
    if (S) memcpy(S, w.S, sizeof(double) * (size_t)n * (size_t)n);
    if (bS) memcpy(bS, w.bS, sizeof(double) * (size_t)n);
    if (free_index) memcpy(free_index, w.hidx, sizeof(int32_t) * (size_t)pb->n_poses);
    if (F0) *F0 = F;
    ws_free(&w);
    return nf;
}

static int stop_requested(const lba_oracle_problem *pb)
{
    return pb->stop ? (*pb->stop != 0) : 0;
}

/* optimizer.initializeOptimization(); optimizer.optimize(max_iters) (src/Optimizer.cc:754-755)
 * with OptimizationAlgorithmLevenberg::solve restated per SURVEY A.3-A.4, followed by
 * the outlier gate of src/Optimizer.cc:757-775. */
int lba_oracle_solve(const lba_oracle_problem *pb, lba_oracle_result *res)
{
    res->iters_done = 0; res->n_solves = 0; res->n_outliers = 0; res->n_trace = 0;
    res->lambda = 0.0; res->cost = 0.0; res->cost0 = 0.0; res->status = 0;
    const int NP = pb->n_poses, P = pb->n_points, E = pb->n_edges;

    /* early return before the solve (src/Optimizer.cc:749-751): nothing is written */
    if (stop_requested(pb)) {
        memcpy(res->poses, pb->poses, sizeof(double) * 7 * (size_t)NP);
        memcpy(res->points, pb->points, sizeof(double) * 3 * (size_t)P);
        for (int e = 0; e < E; ++e) { res->chi2[e] = 0.0; res->outlier[e] = 0; }
        res->status = 1;
        return 1;
    }

    ws_t w;
    ws_init(&w, pb);
    const int nvec_p = 6 * w.nfree;
    double lambda = 0.0, ni = 2.0, F_cur = 0.0;
    int ok = 1;
    const int max_trials = pb->max_trials > 0 ? pb->max_trials : 10;   /* _maxTrialsAfterFailure */
    int have_system = (w.nfree > 0 || P > 0) && E > 0;   /* optimize() returns -1 on an empty index map */

    if (!have_system) res->status = 3;
    for (int it = 0; have_system && it < pb->max_iters && !stop_requested(pb) && ok; ++it) {
        double F0 = ws_errors(&w);
        if (it == 0) res->cost0 = F0;
        ws_build(&w);
        if (it == 0) {
            /* computeLambdaInit: tau * max |H_jj| over all free active vertices, tau = 1e-5 */
            double md = 0.0;
            for (int i = 0; i < w.nfree; ++i)
                for (int a = 0; a < 6; ++a) md = fmax(fabs(w.Hpp[36 * i + a * 7]), md);
            for (int l = 0; l < P; ++l)
                if (w.pt_active[l])
                    for (int a = 0; a < 3; ++a) md = fmax(fabs(w.Hll[9 * l + a * 4]), md);
            lambda = 1e-5 * md;
            ni = 2.0;
        }
        double rho = 0.0;
        int qmax = 0;
        do {
            memcpy(w.poses_bk, w.poses, sizeof(double) * 7 * (size_t)NP);      /* push() */
            memcpy(w.points_bk, w.points, sizeof(double) * 3 * (size_t)P);
            const int ok2 = ws_solve(&w, lambda, 0);
            if (ok2) ws_update(&w);
            double F1 = ws_errors(&w);
            if (!ok2) F1 = DBL_MAX;
            /* computeScale: sum_j x_j (lambda x_j + b_j) over poses then points */
            double scale = 0.0;
            if (ok2) {
                for (int j = 0; j < nvec_p; ++j) scale += w.xp[j] * (lambda * w.xp[j] + w.bp[j]);
                for (int l = 0; l < P; ++l)
                    if (w.pt_active[l])
                        for (int k = 0; k < 3; ++k)
                            scale += w.xl[3 * l + k] * (lambda * w.xl[3 * l + k] + w.bl[3 * l + k]);
            }
            scale += 1e-3;
            rho = (F0 - F1) / scale;
            const int tr = res->n_trace;
            int accepted = 0, lambda_ok = 1;
            if (tr < LBA_ORACLE_MAX_TRACE) {
                res->tr_lambda[tr] = lambda; res->tr_f0[tr] = F0; res->tr_f1[tr] = F1; res->tr_rho[tr] = rho;
            }
            if (rho > 0.0 && isfinite(F1)) {
                double alpha = 1.0 - pow(2.0 * rho - 1.0, 3);
                alpha = fmin(alpha, 2.0 / 3.0);
                const double factor = fmax(1.0 / 3.0, alpha);
                lambda *= factor;
                ni = 2.0;
                F0 = F1;
                accepted = 1;                                                   /* discardTop() */
            } else {
                lambda *= ni;
                ni *= 2.0;
                memcpy(w.poses, w.poses_bk, sizeof(double) * 7 * (size_t)NP);   /* pop() */
                memcpy(w.points, w.points_bk, sizeof(double) * 3 * (size_t)P);
                lambda_ok = isfinite(lambda);
            }
            if (tr < LBA_ORACLE_MAX_TRACE) { res->tr_accept[tr] = accepted; res->n_trace = tr + 1; }
            res->n_solves++;
            qmax++;
            if (!lambda_ok) break;
        } while (rho < 0.0 && qmax < max_trials && !stop_requested(pb));
        F_cur = F0;
        res->iters_done = it + 1;
        if (qmax == max_trials || rho == 0.0 || !isfinite(lambda)) ok = 0;    /* Terminate */
    }

    /* A8: chi2 from the stored _error (stale after a rejected last trial, SURVEY A.4
     * quirk) or recomputed at the final state; depth at the final estimates. */
    if (!pb->stale_error_quirk) F_cur = ws_errors(&w);
    memcpy(res->poses, w.poses, sizeof(double) * 7 * (size_t)NP);
    memcpy(res->points, w.points, sizeof(double) * 3 * (size_t)P);
    for (int e = 0; e < E; ++e) {
        const double *er = w.err + 3 * e;
        const double chi2 = pb->inv_sigma2[e] * (er[0] * er[0] + er[1] * er[1] + er[2] * er[2]);
        double Xc[3];
        lba_oracle_se3_map(w.poses + 7 * pb->edge_pose[e], w.points + 3 * pb->edge_point[e], Xc);
        const int out = (chi2 > pb->chi2_gate) || !(Xc[2] > 0.0);
        res->chi2[e] = chi2;
        res->outlier[e] = (uint8_t)out;
        res->n_outliers += out;
    }
    res->lambda = lambda;
    res->cost = F_cur;
    ws_free(&w);
    return res->status;
}

/* ------------------------------------------------------------------------- */
/* Pose-only optimisation (A9')                                               */
/* ------------------------------------------------------------------------- */

/* EdgeSE3ProjectXYZOnlyPose::linearizeOplus (src/OptimizableTypes.cpp:54-69):
 * J = -projectJac(Xc) * [ -[Xc]x | I ]; error as include/OptimizableTypes.h:41-46 */
static double pose_errors(const lba_oracle_pose_problem *pb, const double pose[7], const uint8_t *level1,
                          int use_robust, double *err)
{
    const double cam[4] = { pb->fx, pb->fy, pb->cx, pb->cy };
    double F = 0.0;
    for (int i = 0; i < pb->n; ++i) {
        if (level1[i]) continue;
        double Xc[3];
        lba_oracle_se3_map(pose, pb->Xw + 3 * i, Xc);
        edge_error(Xc, pb->obs + 2 * i, cam, err + 2 * i);
        const double chi2 = pb->inv_sigma2[i] * (err[2 * i] * err[2 * i] + err[2 * i + 1] * err[2 * i + 1]);
        if (use_robust && pb->huber_delta > 0.0) {
            double rho[3];
            lba_oracle_huber(chi2, pb->huber_delta, rho);
            F += rho[0];
        } else {
            F += chi2;
        }
    }
    return F;
}

/* The LM iterations of one round of the pose-only optimisation over the active matches (level1[i] == 0), from and into `pose`
 * (g2o's OptimizationAlgorithmLevenberg on one VertexSE3Expmap with EdgeSE3ProjectXYZOnlyPose edges: SURVEY App. A.4). */
static void pose_lm_round(const lba_oracle_pose_problem *pb, double pose[7], const uint8_t *level1, int robust, int its, double *err)
{
    const int n = pb->n;
    const double cam[4] = { pb->fx, pb->fy, pb->cx, pb->cy };
    const double I9[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
        int n_act = 0;
    for (int i = 0; i < n; ++i) n_act += !level1[i];
    double lambda = 0.0, ni = 2.0;
    int ok = n_act > 0;
    for (int it = 0; it < its && ok; ++it) {
        double F0 = pose_errors(pb, pose, level1, robust, err);
        double H[36] = { 0 }, b[6] = { 0 };
        for (int i = 0; i < n; ++i) {
            if (level1[i]) continue;
            double Xc[3], Jp[6], Jc[12];
            lba_oracle_se3_map(pose, pb->Xw + 3 * i, Xc);
            edge_jacobians(I9, Xc, cam, Jp, Jc);
            const double om = pb->inv_sigma2[i];
            double wgt = 1.0;
            if (robust && pb->huber_delta > 0.0) {
                double rho[3];
                lba_oracle_huber(om * (err[2 * i] * err[2 * i] + err[2 * i + 1] * err[2 * i + 1]), pb->huber_delta, rho);
                wgt = rho[1];
            }
            const double wo = wgt * om, r0 = -om * err[2 * i] * wgt, r1 = -om * err[2 * i + 1] * wgt;
            for (int a = 0; a < 6; ++a) {
                b[a] += Jc[a] * r0 + Jc[6 + a] * r1;
                for (int c = 0; c < 6; ++c) H[a * 6 + c] += wo * (Jc[a] * Jc[c] + Jc[6 + a] * Jc[6 + c]);
            }
        }
        if (it == 0) {
            double md = 0.0;
            for (int a = 0; a < 6; ++a) md = fmax(fabs(H[a * 7]), md);
            lambda = 1e-5 * md;
            ni = 2.0;
        }
        double rho = 0.0;
        int qmax = 0;
        do {
            double bk[7], Hd[36], x[6];
            memcpy(bk, pose, sizeof bk);
            memcpy(Hd, H, sizeof Hd);
            for (int a = 0; a < 6; ++a) Hd[a * 7] += lambda;
            const int ok2 = chol_solve(Hd, 6, b, x);
            if (ok2) {
                double ex[7];
                lba_oracle_se3_exp(x, ex);
                lba_oracle_se3_mul(ex, pose, pose);
            }
            double F1 = pose_errors(pb, pose, level1, robust, err);
            if (!ok2) F1 = DBL_MAX;
            double scale = 0.0;
            if (ok2) for (int a = 0; a < 6; ++a) scale += x[a] * (lambda * x[a] + b[a]);
            scale += 1e-3;
            rho = (F0 - F1) / scale;
            if (rho > 0.0 && isfinite(F1)) {
                double alpha = 1.0 - pow(2.0 * rho - 1.0, 3);
                alpha = fmin(alpha, 2.0 / 3.0);
                lambda *= fmax(1.0 / 3.0, alpha);
                ni = 2.0;
                F0 = F1;
            } else {
                lambda *= ni;
                ni *= 2.0;
                memcpy(pose, bk, sizeof bk);
                if (!isfinite(lambda)) { qmax++; break; }
            }
            qmax++;
        } while (rho < 0.0 && qmax < 10);
        if (qmax == 10 || rho == 0.0 || !isfinite(lambda)) ok = 0;
    }
}

int lba_oracle_pose_opt(const lba_oracle_pose_problem *pb, double pose_out[7],
                        uint8_t *outlier, double *chi2_out)
{
    const int n = pb->n;
    const double cam[4] = { pb->fx, pb->fy, pb->cx, pb->cy };
    
    double *err = (double *)calloc(2 * (size_t)n + 2, sizeof(double));
    uint8_t *level1 = (uint8_t *)calloc((size_t)n + 1, 1);
    double pose[7], pose0[7];
    memcpy(pose0, pb->pose0, sizeof pose0);
    lba_oracle_se3_normalize(pose0);
    memcpy(pose, pose0, sizeof pose);
    for (int i = 0; i < n; ++i) outlier[i] = 0;
    int n_bad = 0;
    for (int round = 0; round < pb->rounds; ++round) {
        const int robust = (round <= 2) ? 1 : 0;           /* kernel dropped after round 2 */
        memcpy(pose, pose0, sizeof pose);                  /* restart from the initial estimate */
        pose_lm_round(pb, pose, level1, robust, pb->its_per_round, err);
        /* re-classify every correspondence at the round's final pose */
        n_bad = 0;
        for (int i = 0; i < n; ++i) {
            double Xc[3], e[2];
            lba_oracle_se3_map(pose, pb->Xw + 3 * i, Xc);
            edge_error(Xc, pb->obs + 2 * i, cam, e);
            const double c2 = pb->inv_sigma2[i] * (e[0] * e[0] + e[1] * e[1]);
            const int bad = (c2 > pb->chi2_gate) || !(Xc[2] > 0.0);
            chi2_out[i] = c2;
            outlier[i] = (uint8_t)bad;
            level1[i] = (uint8_t)bad;
            n_bad += bad;
        }
        if (n - n_bad < 10) break;
    }
    memcpy(pose_out, pose, sizeof pose);
    free(err); free(level1);
    return n - n_bad;
}

/* OpenMP build only (liblba_oracle_omp.so): number of threads of the all-cores timing variant; returns what is in force
 * (1 in the serial library). */
/* ---------------------------------------------------------------------------------------------
 * Hypothesis stage of PoseOptimization: RANSAC over minimal P3P solves (the classical scheme behind
 * cv::solvePnPRansac, which the reference calls with useExtrinsicGuess = false at src/Optimizer.cc:437;
 * OpenCV's own arithmetic is not in the reference tree).  Plain restatement of the same algorithm the
 * kernel runs: Grunert's three-point solution (Haralick et al. 1994: quartic in v = s3 / s1), every
 * candidate scored on all matches by its sigma-consensus++ loss (below), among those with at least 4 matches inside
 * the threshold; ties by index.
 * --------------------------------------------------------------------------------------------- */
typedef struct { double re, im; } cplx_t;
static cplx_t c_mul(cplx_t a, cplx_t b) { cplx_t r = { a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re }; return r; }
static cplx_t c_sub(cplx_t a, cplx_t b) { cplx_t r = { a.re - b.re, a.im - b.im }; return r; }
static cplx_t c_div(cplx_t a, cplx_t b)
{
    const double d = b.re * b.re + b.im * b.im;
    cplx_t r = { (a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d };
    return r;
}

static void quartic_roots(double c3, double c2, double c1, double c0, cplx_t z[4])
{
    const double rb = 2.0 * fmax(fmax(fabs(c3), sqrt(fabs(c2))), fmax(cbrt(fabs(c1)), sqrt(sqrt(fabs(c0))))) + 1e-300;
    const double r0 = 0.5 * rb;
    z[0].re = r0 * 0.9210609940028851; z[0].im = r0 * 0.3894183423086505;
    z[1].re = -z[0].im; z[1].im = z[0].re; z[2].re = -z[0].re; z[2].im = -z[0].im; z[3].re = z[0].im; z[3].im = -z[0].re;
    /* sweeps until no root moves by more than 1e-15 of the root bound (80 at most), like the kernel */
    const double tol2 = (1e-15 * rb) * (1e-15 * rb);
    for (int it = 0; it < 80; ++it) {
        double moved = 0.0;
        for (int k = 0; k < 4; ++k) {
            const cplx_t x = z[k];
            cplx_t pv = { x.re + c3, x.im };
            pv = c_mul(pv, x); pv.re += c2;
            pv = c_mul(pv, x); pv.re += c1;
            pv = c_mul(pv, x); pv.re += c0;
            cplx_t den = { 1.0, 0.0 };
            for (int j = 0; j < 4; ++j) if (j != k) den = c_mul(den, c_sub(x, z[j]));
            if (den.re * den.re + den.im * den.im > 0.0) {
                const cplx_t st = c_div(pv, den);
                z[k] = c_sub(x, st);
                moved = fmax(moved, st.re * st.re + st.im * st.im);
            }
        }
        if (moved <= tol2) break;
    }
}

static void cross3(const double a[3], const double b[3], double o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}

static int triple_frame(const double P1[3], const double P2[3], const double P3[3], double F[9])
{
    double d1[3] = { P2[0] - P1[0], P2[1] - P1[1], P2[2] - P1[2] }, d2[3] = { P3[0] - P1[0], P3[1] - P1[1], P3[2] - P1[2] };
    const double n1 = sqrt(d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2]);
    if (!(n1 > 0.0)) return 0;
    d1[0] /= n1; d1[1] /= n1; d1[2] /= n1;
    double e3[3], e2[3];
    cross3(d1, d2, e3);
    const double n3 = sqrt(e3[0] * e3[0] + e3[1] * e3[1] + e3[2] * e3[2]);
    if (!(n3 > 1e-12 * n1)) return 0;
    e3[0] /= n3; e3[1] /= n3; e3[2] /= n3;
    cross3(e3, d1, e2);
    F[0] = d1[0]; F[3] = d1[1]; F[6] = d1[2]; F[1] = e2[0]; F[4] = e2[1]; F[7] = e2[2]; F[2] = e3[0]; F[5] = e3[1]; F[8] = e3[2];
    return 1;
}

static double dist2(const double *a, const double *b)
{
    const double x = a[0] - b[0], y = a[1] - b[1], z = a[2] - b[2];
    return x * x + y * y + z * z;
}

static int p3p_grunert(double X[3][3], double j[3][3], double Rs[4][9], double ts[4][3])
{
    const double a2 = dist2(X[1], X[2]), b2 = dist2(X[0], X[2]), c2 = dist2(X[0], X[1]);
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
    cplx_t z[4];
    quartic_roots(c3, c2q, c1, c0, z);
    double Fw[9];
    if (!triple_frame(X[0], X[1], X[2], Fw)) return 0;
    int ns = 0;
    for (int k = 0; k < 4; ++k) {
        if (!(fabs(z[k].im) <= 1e-6 * (1.0 + fabs(z[k].re)))) continue;
        double v = z[k].re;
        for (int it = 0; it < 2; ++it) {
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
        double *R = Rs[ns];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) R[r * 3 + c] = Fc[r * 3] * Fw[c * 3] + Fc[r * 3 + 1] * Fw[c * 3 + 1] + Fc[r * 3 + 2] * Fw[c * 3 + 2];
        for (int r = 0; r < 3; ++r) ts[ns][r] = P1[r] - (R[r * 3] * X[0][0] + R[r * 3 + 1] * X[0][1] + R[r * 3 + 2] * X[0][2]);
        ++ns;
    }
    return ns;
}

/* ---------------------------------------------------------------------------------------------
 * sigma-consensus++ (MAGSAC++; D. Barath, J. Noskova, M. Ivashechkin, J. Matas, "MAGSAC++, a fast, reliable and accurate
 * robust estimator", CVPR 2020), what flag 38 = cv::USAC_MAGSAC of the reference's call (src/Optimizer.cc:437,
 * Examples/Monocular/TartanAir.yaml:51) scores models with; restated from the paper, OpenCV's source is not in the tree.
 * A residual r (here r^2 = the edge's chi2, 2 degrees of freedom) is not classified at one threshold: its noise scale sigma
 * is marginalised over (0, sigma_max], sigma_max = tau / k with tau the caller's threshold (reprojectionError, tau^2 =
 * chi2_gate) and k^2 = 9.21034 the 0.99 quantile of chi^2 with 2 degrees of freedom.  With x = r^2 / (2 sigma_max^2),
 * x_k = k^2 / 2 and n = 2 the paper's incomplete gamma functions are elementary:
 *     Gamma(1/2, x) = sqrt(pi) erfc(sqrt x),      gamma(3/2, x) = sqrt(pi) / 2 erf(sqrt x) - sqrt x exp(-x)
 *   weight (eq. 6; the IRLS weight of the local optimisation)   w(r)   = Gamma(1/2, x) - Gamma(1/2, x_k)         r <= tau, else 0
 *   loss   (eq. 9-10; the model quality is minus its sum)       rho(r) = sigma_max^2 / 2 gamma(3/2, x) + r^2 / 4 w(r)   r <= tau,
 *                                                                rho(tau) beyond (and for a point behind the camera)
 * both up to the common factor C(n) 2^((n+1)/2) / sigma_max, which cancels in every comparison; the loss is reported
 * divided by rho(tau): in [0, 1] per match, 1 = outlier.
 * --------------------------------------------------------------------------------------------- */
#define MAGSAC_K2 9.210340371976184          /* -2 ln(0.01) */
static void magsac_terms(double chi2, int in_front, double gate, double *loss, double *weight)
{
    const double sq_pi = 1.7724538509055160273, xk = 0.5 * MAGSAC_K2;
    const double s2 = gate / MAGSAC_K2;                      /* sigma_max^2 */
    const double g_k = sq_pi * erfc(sqrt(xk));
    const double rho_max = 0.5 * s2 * (0.5 * sq_pi * erf(sqrt(xk)) - sqrt(xk) * exp(-xk));
    if (!in_front || !(chi2 <= gate)) { *loss = 1.0; *weight = 0.0; return; }
    const double x = chi2 / (2.0 * s2), sx = sqrt(x);
    const double w = sq_pi * erfc(sx) - g_k;
    const double rho = 0.5 * s2 * (0.5 * sq_pi * erf(sx) - sx * exp(-x)) + 0.25 * chi2 * w;
    *loss = rho / rho_max; *weight = w > 0.0 ? w : 0.0;
}

/* tentative inliers of a pose at the threshold (count) and its MAGSAC++ loss; mark / wgt (optional): matches outside the
 * threshold, IRLS weights of those inside (largest = 1) */
static void pose_score(const lba_oracle_pose_problem *pb, const double pose[7], double *cnt_o, double *cst_o, uint8_t *mark, double *wgt)
{
    const double cam[4] = { pb->fx, pb->fy, pb->cx, pb->cy };
    double cnt = 0.0, cst = 0.0;
    for (int i = 0; i < pb->n; ++i) {
        double Xc[3], e[2];
        lba_oracle_se3_map(pose, pb->Xw + 3 * i, Xc);
        edge_error(Xc, pb->obs + 2 * i, cam, e);
        const double om = pb->inv_sigma2 ? pb->inv_sigma2[i] : 1.0;
        const double chi2 = om * (e[0] * e[0] + e[1] * e[1]);
        const int in = (Xc[2] > 0.0) && (chi2 <= pb->chi2_gate);
        double ls, wt;
        magsac_terms(chi2, Xc[2] > 0.0, pb->chi2_gate, &ls, &wt);
        cnt += in ? 1.0 : 0.0; cst += ls;
        if (mark) mark[i] = in ? 0 : 1;
        if (wgt) wgt[i] = wt / (1.7724538509055160273 * (1.0 - erfc(sqrt(0.5 * MAGSAC_K2))));      /* w(0) = sqrt(pi) (1 - erfc(sqrt x_k)) */
    }
    *cnt_o = cnt; *cst_o = cst;
}

/* The same with cv::solvePnPRansac's stopping rule and one local-optimisation step (Optimizer.cc:437: confidence 0.95,
 * flag 38 = USAC_MAGSAC, whose pipeline refits the best model on its inliers):
 *   confidence in (0, 1): the samples are walked in order; after sample h, N = log(1 - confidence) / log(1 - w^3) with w the
 *     inlier ratio of the best pose so far, and the walk ends once h + 1 >= N (info[0] = samples admitted);
 *   lo_its > 0: LM refit of the winner on the matches inside the threshold, each weighted by its sigma-consensus weight at
 *     the winner (one IRLS step of MAGSAC++'s model polishing), no robust kernel, lo_its iterations; kept when its loss is
 *     lower (info[1] = kept, info[2] = matches inside the threshold at the pose returned). */
int lba_oracle_pose_ransac_lo(const lba_oracle_pose_problem *pb, int n_hyp, const int32_t *samples, double confidence, int lo_its,
                              double pose_out[7], int32_t info[3])
{
    const int n = pb->n;
    double best_cnt = 3.5, best_cost = DBL_MAX, bestR[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, bestt[3] = { 0, 0, 0 };
    int have = 0, used = n_hyp;
    for (int h = 0; h < n_hyp; ++h) {
        double X[3][3], jb[3][3];
        int okh = 1;
        for (int m = 0; m < 3; ++m) {
            const int i = samples[3 * h + m];
            if (i < 0 || i >= n) { okh = 0; break; }
            for (int k = 0; k < 3; ++k) X[m][k] = pb->Xw[3 * i + k];
            const double bx = (pb->obs[2 * i] - pb->cx) / pb->fx, by = (pb->obs[2 * i + 1] - pb->cy) / pb->fy;
            const double nn = 1.0 / sqrt(bx * bx + by * by + 1.0);
            jb[m][0] = bx * nn; jb[m][1] = by * nn; jb[m][2] = nn;
        }
        double Rs[4][9], ts[4][3];
        const int ns = okh ? p3p_grunert(X, jb, Rs, ts) : 0;
        for (int k = 0; k < ns; ++k) {
            const double *R = Rs[k], *t = ts[k];
            double cnt = 0.0, cst = 0.0;
            for (int i = 0; i < n; ++i) {
                const double *Xi = pb->Xw + 3 * i;
                const double x = R[0] * Xi[0] + R[1] * Xi[1] + R[2] * Xi[2] + t[0];
                const double y = R[3] * Xi[0] + R[4] * Xi[1] + R[5] * Xi[2] + t[1];
                const double z = R[6] * Xi[0] + R[7] * Xi[1] + R[8] * Xi[2] + t[2];
                const double om = pb->inv_sigma2 ? pb->inv_sigma2[i] : 1.0;
                const double e0 = pb->obs[2 * i] - (pb->fx * x / z + pb->cx), e1 = pb->obs[2 * i + 1] - (pb->fy * y / z + pb->cy);
                const double chi2 = e0 * (om * e0) + e1 * (om * e1);
                const int in = (z > 0.0) && (chi2 <= pb->chi2_gate);
                double ls, wt;
                magsac_terms(chi2, z > 0.0, pb->chi2_gate, &ls, &wt);
                cnt += in ? 1.0 : 0.0; cst += ls;
            }
            /* the model of lowest sigma-consensus loss among those with at least 4 matches inside the threshold */
            if (cnt > 3.5 && cst < best_cost) {
                best_cnt = cnt; best_cost = cst; have = 1;
                memcpy(bestR, R, sizeof bestR); memcpy(bestt, t, sizeof bestt);
            }
        }
        if (confidence > 0.0 && confidence < 1.0 && have) {
            const double wr = best_cnt / (double)n, w3 = wr * wr * wr;
            const double need = w3 >= 1.0 ? 0.0 : log(1.0 - confidence) / log(1.0 - w3);
            if ((double)(h + 1) >= need) { used = h + 1; break; }
        }
    }
    if (info) { info[0] = used; info[1] = 0; info[2] = have ? (int32_t)best_cnt : 0; }
    memcpy(pose_out, pb->pose0, 7 * sizeof(double));
    lba_oracle_se3_normalize(pose_out);
    if (!have) return 0;
    R_to_quat(bestR, pose_out);
    pose_out[4] = bestt[0]; pose_out[5] = bestt[1]; pose_out[6] = bestt[2];
    lba_oracle_se3_normalize(pose_out);
    if (lo_its > 0) {
        uint8_t *mark = (uint8_t *)calloc((size_t)n + 1, 1);
        double *err = (double *)calloc(2 * (size_t)n + 2, sizeof(double));
        double c0, s0, c1, s1, refit[7];
        /* one iteratively-reweighted step: the matches inside the threshold, each with its sigma-consensus weight at the winner */
        double *wls = (double *)calloc((size_t)n + 1, sizeof(double));
        pose_score(pb, pose_out, &c0, &s0, mark, wls);
        for (int i = 0; i < n; ++i) wls[i] *= pb->inv_sigma2 ? pb->inv_sigma2[i] : 1.0;
        lba_oracle_pose_problem pw = *pb;
        pw.inv_sigma2 = wls;
        memcpy(refit, pose_out, sizeof refit);
        pose_lm_round(&pw, refit, mark, 0, lo_its, err);
        pose_score(pb, refit, &c1, &s1, NULL, NULL);
        free(wls);
        const int keep = c1 > 3.5 && s1 < s0;
        if (keep) memcpy(pose_out, refit, sizeof refit);
        if (info) { info[1] = keep; info[2] = (int32_t)(keep ? c1 : c0); }
        free(mark); free(err);
    }
    return (int)best_cnt;
}

int lba_oracle_pose_ransac(const lba_oracle_pose_problem *pb, int n_hyp, const int32_t *samples, double pose_out[7])
{
    return lba_oracle_pose_ransac_lo(pb, n_hyp, samples, 0.0, 0, pose_out, NULL);
}


int lba_oracle_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}
