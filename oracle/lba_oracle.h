/*
 * lba_oracle.h — CPU restatement of MoV-SLAM's local bundle adjustment path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or
 * call it, and there only as the checker / the timed CPU baseline.  The product
 * path (mov-slam_amd/csrc, libmovba.so) never links or calls into this directory.
 *
 * PARITY UNPINNED: the reference holds no tests, golden vectors or fixtures for this
 * path and cannot be built or run here (g2o / Sophus / Eigen /
 * OpenCV absent, prebuilt lib/ .so files are aarch64).  The arithmetic of the path lives in
 * g2o (unpinned `git clone` of master, reference Dockerfile:154), so this file restates
 * g2o's published algorithm (OptimizationAlgorithmLevenberg, BlockSolver<6,3> with
 * Schur complement, RobustKernelHuber, SE3Quat, VertexSE3Expmap) and anchors it on
 * the reference's own call sites:
 *   src/Optimizer.cc:532-545, 554-584, 623-672, 754-755, 757-804   (graph + solve + gate)
 *   include/OptimizableTypes.h:92-121, src/OptimizableTypes.cpp:158-180 (mono edge)
 *   include/OptimizableTypes.h:30-58,  src/OptimizableTypes.cpp:54-69   (pose-only edge)
 *   src/CameraModels/Pinhole.cpp:36-43, 77-88                     (project / projectJac)
 * It is pinned only by self-validation (tests/test_oracle_*.py: finite-difference
 * Jacobians, SE3 identities, an independent dense numpy LM in tests/golden/).
 */
#ifndef LBA_ORACLE_H
#define LBA_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int n_poses;                 /* K + F keyframe vertices                        */
    int n_points;                /* P map-point vertices                           */
    int n_edges;                 /* E monocular reprojection edges                 */
    const double *poses;         /* n_poses x 7: qx qy qz qw tx ty tz  (Tcw)       */
    const uint8_t *pose_fixed;   /* n_poses: 1 = fixed vertex                      */
    const double *points;        /* n_points x 3 world positions                   */
    const int32_t *edge_pose;    /* E: keyframe vertex index                       */
    const int32_t *edge_point;   /* E: map-point vertex index                      */
    const double *obs;           /* E x 2 pixel measurements                       */
    const double *inv_sigma2;    /* E: information = inv_sigma2 * I2                */
    double fx, fy, cx, cy;       /* pinhole parameters (float-rounded upstream)    */
    double huber_delta;          /* RobustKernelHuber delta; <= 0 disables kernel  */
    double chi2_gate;            /* outlier gate (5.0 in the reference)            */
    int max_iters;               /* optimize(N) outer iterations (10)              */
    int stale_error_quirk;       /* 1: chi2 from the edges' stored _error (g2o)    */
    int max_trials;              /* g2o maxTrialsAfterFailure property; 0 -> 10    */
    const volatile uint8_t *stop;/* forceStopFlag, may be NULL                     */
    /* stereo observations: g2o::EdgeStereoSE3ProjectXYZ, built at src/Optimizer.cc:673-705 when
     * mvuRight[idx] >= 0.  obs_right[e] >= 0: third measurement u_right, residual u_r - (u - bf/z);
     * < 0 (or obs_right == NULL): monocular edge. */
    const double *obs_right;     /* E or NULL                                      */
    double bf;                   /* KeyFrame::mbf                                  */
    /* intrinsics by keyframe: every edge carries ITS keyframe's camera (e->pCamera = pKFi->mpCamera, src/Optimizer.cc:664;
     * e->fx .. e->bf from pKFi, :690-695).  NULL: fx .. cy / bf above for every keyframe. */
    const double *cam_kf;        /* n_poses x 4 or NULL                            */
    const double *bf_kf;         /* n_poses or NULL                                */
} lba_oracle_problem;

#define LBA_ORACLE_MAX_TRACE 128

typedef struct {
    double *poses;               /* n_poses x 7 out                                */
    double *points;              /* n_points x 3 out                               */
    double *chi2;                /* E out                                          */
    uint8_t *outlier;            /* E out: chi2 > gate || depth <= 0               */
    int iters_done;              /* outer iterations whose solve() ran             */
    int n_solves;                /* linear solves (accepted + rejected trials)     */
    int n_outliers;
    double lambda;               /* final damping                                  */
    double cost;                 /* final robust cost F                            */
    double cost0;                /* initial robust cost                            */
    int status;                  /* 0 ok, 1 stopped before solve, 3 nothing to do  */
    /* per-trial trace */
    int n_trace;
    double tr_lambda[LBA_ORACLE_MAX_TRACE];
    double tr_f0[LBA_ORACLE_MAX_TRACE];
    double tr_f1[LBA_ORACLE_MAX_TRACE];
    double tr_rho[LBA_ORACLE_MAX_TRACE];
    int tr_accept[LBA_ORACLE_MAX_TRACE];
} lba_oracle_result;

/* Full local-BA solve: A3-A8 of SURVEY.md §8(a). Returns status. */
int lba_oracle_solve(const lba_oracle_problem *pb, lba_oracle_result *res);

/* One linearisation at the given state: assembles the reduced system for damping
 * `lambda` exactly as the solve does. Outputs (any may be NULL):
 *   Hpp   nfree x 36 (row-major 6x6 diagonal blocks, undamped)
 *   bp    nfree x 6
 *   Hll   P x 9 (undamped), bl P x 3
 *   S     (6 nfree)^2 row-major dense reduced matrix (damped, symmetric, full)
 *   bS    6 nfree
 *   free_index n_poses: hessian index of each pose or -1
 * Returns nfree. */
int lba_oracle_linearize(const lba_oracle_problem *pb, double lambda,
                         double *Hpp, double *bp, double *Hll, double *bl,
                         double *S, double *bS, int32_t *free_index, double *F0);

/* Building blocks exposed for unit tests. */
void lba_oracle_se3_exp(const double upd[6], double out_qt[7]);
void lba_oracle_se3_mul(const double a[7], const double b[7], double out[7]);
void lba_oracle_se3_map(const double qt[7], const double X[3], double Xc[3]);
void lba_oracle_se3_normalize(double qt[7]);
/* error (2), J_point (2x3 row-major), J_pose (2x6 row-major) of EdgeSE3ProjectXYZ */
void lba_oracle_edge(const double qt[7], const double X[3], const double obs[2],
                     const double cam[4], double err[2], double Jp[6], double Jc[12]);
void lba_oracle_huber(double chi2, double delta, double rho[3]);

/* Pose-only optimisation (A9'): EdgeSE3ProjectXYZOnlyPose + dense 6x6 LM.
 * rounds x its LM iterations, re-classifying outliers with chi2 > gate after each
 * round (robust kernel dropped from round index >= 2), restarting from pose0 each
 * round like ORB-SLAM-style motion-only BA.  Returns number of inliers. */
typedef struct {
    int n;                       /* 2D-3D correspondences                          */
    const double *Xw;            /* n x 3                                          */
    const double *obs;           /* n x 2                                          */
    const double *inv_sigma2;    /* n                                              */
    double fx, fy, cx, cy;
    double pose0[7];
    double huber_delta;
    double chi2_gate;
    int rounds;
    int its_per_round;
} lba_oracle_pose_problem;

int lba_oracle_pose_opt(const lba_oracle_pose_problem *pb, double pose_out[7],
                        uint8_t *outlier, double *chi2);

/* Hypothesis stage of PoseOptimization (RANSAC over P3P, see lba_oracle.c): best pose over n_hyp minimal samples
 * (n_hyp x 3 match indices), scored at pb->chi2_gate; pose_out = normalised pb->pose0 and 0 when no candidate has
 * >= 4 inliers.  Returns the inlier count of the returned pose. */
int lba_oracle_pose_ransac(const lba_oracle_pose_problem *pb, int n_hyp, const int32_t *samples, double pose_out[7]);
/* ... with the stopping rule of `confidence` and one local-optimisation step (lba_oracle.c); info: samples admitted, refit kept, inliers */
int lba_oracle_pose_ransac_lo(const lba_oracle_pose_problem *pb, int n_hyp, const int32_t *samples, double confidence, int lo_its,
                              double pose_out[7], int32_t info[3]);

/* threads of the OpenMP timing variant (liblba_oracle_omp.so); the serial library always answers 1 */
int lba_oracle_set_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
