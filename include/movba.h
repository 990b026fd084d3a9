/*
 * movba.h — C-ABI of the MI355X-native local bundle adjustment (libmovba.so).
 *
 * Drop-in boundary for MoV-SLAM's optimizer hot path.  The reference has no FFI: the
 * boundary is the static-method surface of class MOV_SLAM::Optimizer
 * (/root/reference/include/Optimizer.h:45-60).  An adapter with those exact signatures
 * (mov-slam_amd/host/Optimizer.cc) flattens the KeyFrame/MapPoint graph into the plain
 * arrays below and calls these entry points; everything from "arrays in" to
 * "poses / points / chi2 / outlier list out" — what the reference delegates to
 * g2o at src/Optimizer.cc:754-755 plus the gate at :757-804 — runs in hand-written
 * HIP kernels for gfx950.  No Eigen / Sophus / g2o / OpenCV / torch types cross this
 * boundary: plain pointers and sizes only.
 *
 * Threading: a handle is bound to one device + one stream and must be used by one
 * thread at a time; different handles are independent (LocalMapping thread and
 * Tracking thread each own one, reference src/System.cc:128-129).
 * Ownership: the caller owns every buffer; the library keeps no caller pointer
 * after a call returns (the stop flag is read only during the call).
 * Errors: int status, never throws / exits; on non-zero status nothing was written
 * to the result buffers except `status` (reference convention of silent early
 * returns, src/Optimizer.cc:525-529, 749-751).
 */
#ifndef MOVBA_H
#define MOVBA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOVBA_VERSION 5

/* status codes */
#define MOVBA_OK              0
#define MOVBA_STOPPED         1   /* *stop was set before the solve (Optimizer.cc:749-751)   */
#define MOVBA_NO_FIXED        2   /* no fixed keyframe vertex (Optimizer.cc:525-529)          */
#define MOVBA_EMPTY           3   /* nothing to optimise (no edges / no free vertex)          */
#define MOVBA_ERR_ARG        -1
#define MOVBA_ERR_HIP        -2   /* HIP runtime failure (no device, OOM, launch error)       */
#define MOVBA_ERR_STATE      -3   /* call order violated (run before upload, ...)             */
#define MOVBA_ERR_DEVICE_WAIT -4  /* a workgroup gave up a bounded wait inside a launch (20 ms) in the FIRST attempt of a run
                                     AND in its repeat on the paths that do not wait inside launches: never seen.  The usual
                                     outcome of a given-up wait is MOVBA_OK with n_sync_timeouts > 0 (movba_lba_run solves the
                                     window again, see movba_lba_result); results are never handed out from a failed attempt */
#define MOVBA_ERR_TOO_LARGE  -5   /* the window's reduced system exceeds what the direct solver can hold             */
/* Free keyframes per window: <= 80 the on-chip PCG (where the reduced matrix fits its registers), <= 432 the one-launch
 * direct solver, beyond that one launch per block column; the solution vector of the latter's back substitution lives in
 * LDS (48 doubles per 8 keyframes beside 28 KB of work space: 159 KB at 2 700 keyframes). */
#define MOVBA_MAX_FREE_KEYFRAMES 2700

/* flags */
#define MOVBA_FLAG_STALE_ERROR_QUIRK 1u  /* chi2 of a rejected last trial, as g2o leaves it */

typedef struct movba_handle movba_handle;

/* One flattened local-BA window.  Replaces the g2o graph the reference builds at
 * src/Optimizer.cc:532-747 (vertices :554-584, :623-632; mono edges :646-672). */
typedef struct {
    int32_t n_poses;            /* K + F keyframe vertices, ascending KeyFrame::mnId          */
    int32_t n_points;           /* P map-point vertices                                       */
    int32_t n_edges;            /* E monocular edges, caller order = vpEdgesMono order        */
    const double  *poses;       /* n_poses x 7: qx qy qz qw tx ty tz = Tcw (Optimizer.cc:559) */
    const uint8_t *pose_fixed;  /* n_poses: 1 = setFixed(true) (Optimizer.cc:561, 578)        */
    const double  *points;      /* n_points x 3 world position (Optimizer.cc:627)             */
    const int32_t *edge_pose;   /* E: index into poses   (vertex 1 of the edge, :655)         */
    const int32_t *edge_point;  /* E: index into points  (vertex 0 of the edge, :654)         */
    const double  *obs;         /* E x 2: mvKeysUn[idx].pt (Optimizer.cc:648-650)             */
    const double  *inv_sigma2;  /* E: information = inv_sigma2 * I2 (Optimizer.cc:656-657)    */
    double fx, fy, cx, cy;      /* GeometricCamera::getParameter(0..3), float -> double       */
    double huber_delta;         /* (double)sqrtf(5.0f) (Optimizer.cc:616); <= 0: no kernel    */
    double chi2_gate;           /* 5.0 (Optimizer.cc:52, 769)                                 */
    int32_t max_iters;          /* optimizer.optimize(10) (Optimizer.cc:755)                  */
    int32_t max_trials;         /* g2o maxTrialsAfterFailure (default-constructed LM, :539); 0 -> 10 */
    uint32_t flags;             /* MOVBA_FLAG_*                                               */
    const volatile uint8_t *stop; /* pbStopFlag (Optimizer.cc:544-545, 749), may be NULL      */
    /* stereo observations (g2o::EdgeStereoSE3ProjectXYZ, Optimizer.cc:673-705): obs_right[e] >= 0 is
     * mvuRight[idx] of a stereo observation (third residual u_r - (u - bf/z)); < 0 or NULL: monocular */
    const double  *obs_right;   /* E or NULL                                                  */
    double bf;                  /* KeyFrame::mbf (Optimizer.cc:695)                           */
    /* intrinsics by keyframe: the reference gives every edge the camera of ITS keyframe (e->pCamera = pKFi->mpCamera,
     * Optimizer.cc:664; e->fx .. e->bf = pKFi->fx .. pKFi->mbf, :690-695).  NULL (every shipped MoV-SLAM configuration has
     * one camera): fx, fy, cx, cy / bf above hold for every keyframe.  Either may be given without the other.        */
    const double  *cam_kf;      /* n_poses x 4 (fx fy cx cy) or NULL                          */
    const double  *bf_kf;       /* n_poses or NULL                                            */
} movba_lba_desc;

#define MOVBA_MAX_TRACE 128

typedef struct {
    double  *poses;             /* n_poses x 7 out (fixed vertices returned unchanged)        */
    double  *points;            /* n_points x 3 out                                           */
    double  *chi2;              /* E out, caller edge order: e->chi2() (Optimizer.cc:769)     */
    uint8_t *outlier;           /* E out: chi2 > gate || !isDepthPositive() (Optimizer.cc:769)*/
    int32_t status;
    int32_t iters_done;         /* outer LM iterations run                                    */
    int32_t n_solves;           /* linear solves = accepted + rejected trials                 */
    int32_t n_outliers;
    int32_t pcg_iters;          /* total PCG iterations over all solves                       */
    int32_t last_rejected;      /* 1 if the final trial was rejected                          */
    double lambda;              /* final damping                                              */
    double cost0, cost;         /* initial / final robust cost (activeRobustChi2)             */
    int32_t n_trace;            /* per-trial trace (min(n_solves, MOVBA_MAX_TRACE) entries)   */
    double  tr_lambda[MOVBA_MAX_TRACE];
    double  tr_f0[MOVBA_MAX_TRACE];
    double  tr_f1[MOVBA_MAX_TRACE];
    double  tr_rho[MOVBA_MAX_TRACE];
    int32_t tr_accept[MOVBA_MAX_TRACE];
    int32_t tr_pcg_iters[MOVBA_MAX_TRACE];   /* PCG iterations of the trial; -1: solved by the dense direct solver; -2: by the
                                                banded factorisation                                                          */
    /* Reduced solve (LinearSolverCSparse in the reference, Optimizer.cc:535): banded Cholesky in one workgroup for the
     * windows it solves fastest (up to ~24 free keyframes at a band of 9), on-chip PCG for windows of up to 80 free
     * keyframes whose reduced matrix fits the PCG workgroup's registers, dense Cholesky otherwise (more keyframes, or
     * denser covisibility) and from the first trial on which one of the other two handed the solve over. */
    int32_t n_direct;           /* trials solved by the dense direct solver                                    */
    int32_t direct_from;        /* n_solves at the switch to the dense direct solver (0: whole solve; > 0: the PCG gave up or
                                   the banded factorisation met a non-positive pivot on that trial), -1: never */
    int32_t n_chol_fail;        /* trials whose factorisation failed (dense solver: non-positive pivot; banded: NaN / inf in
                                   the normal equations): rejected, as g2o rejects a trial whose Cholesky fails */
    int32_t n_pcg_giveups;      /* 0 or 1: hand-overs of the solve to the dense direct solver - the PCG gave up (breakdown or
                                   iteration cap), or the banded factorisation met a finite non-positive pivot (the name is
                                   from when only the PCG could hand over)                                      */
    int32_t n_sync_timeouts;    /* waits inside a launch that a workgroup gave up (one-launch direct solver without all its
                                   workgroups resident): > 0 with status MOVBA_OK = the run was repeated with the multi-launch
                                   direct solver and THIS is its result (the reference never skips a solve,
                                   src/Optimizer.cc:535, 754)                                                                   */
    int32_t n_band;             /* trials solved by the single-workgroup banded factorisation (exact, one launch)  */
} movba_lba_result;

/* Solver options (all have defaults; pass NULL to movba_create for defaults). */
typedef struct {
    double pcg_rel_tol;         /* stop when sqrt(r.z / r0.z0) <= tol       (default 1e-10)   */
    int32_t pcg_max_iters;      /* per solve; reaching it hands the trial to the direct solver (default 200) */
    int32_t run_ahead;          /* trial sets the host keeps queued ahead    (default 2)       */
    int32_t profile;            /* bit k set: bracket launches of kernel class k (see movba_profile)
                                 * with HIP events on the handle's stream; 0x3f = all          */
    int32_t pcg_coarse;         /* 0 = default (two-level: block-Jacobi + aggregate coarse level),
                                 * -1 = block-Jacobi only                                       */
    int32_t host_wait;          /* how the calling thread waits for the device between queueing trials:
                                 * 0 = spin on the progress word (default, lowest latency), 1 = sched_yield() between
                                 * looks (for a LocalMapping thread that shares its core with Tracking; the upload's helper
                                 * thread then sleeps between uploads instead of staying awake for 4 ms after each)    */
    int32_t pcg_spill;          /* 1 = keep the PCG for windows whose reduced matrix does not fit the PCG workgroup's registers
                                 * (list tails read from an L2 copy every iteration); 0 = default: such windows take the
                                 * one-launch direct solver from the first trial                                        */
    int32_t solver;             /* 0 = default, 1 = the dense direct solver for every window, 2 = the banded factorisation in one workgroup
                                   where the window's band fits LDS (else as 0), 3 = never the banded factorisation (PCG where it fits,
                                   dense direct solver elsewhere)                                                                    */
    int32_t reorder;            /* 0 = default: free keyframes are renumbered by covisibility (reverse Cuthill-McKee on the pair
                                 * graph) when that shrinks the reduced matrix's envelope by a fifth or more; -1 = keep the
                                 * caller's order (KeyFrame::mnId order, as the reference numbers its vertices)             */
    int32_t pad_o;              /* (keeps the struct a multiple of 8 bytes; until ABI 4 the opt-in two-stream LM loop, removed in
                                 * round 5: measured 1 - 3 % slower on MI355X than the one-stream loop)                  */
} movba_options;

/* Per-kernel-class timing collected with HIP events on the handle's stream. */
#define MOVBA_NKERNELS 6
typedef struct {
    const char *name[MOVBA_NKERNELS];
    double      ms[MOVBA_NKERNELS];      /* summed device time                               */
    int64_t     launches[MOVBA_NKERNELS];
    double      upload_ms, structure_ms, download_ms;   /* host-side phases, wall clock     */
} movba_profile;

int  movba_version(void);
const char *movba_status_string(int status);

/* device: HIP device ordinal.  stream: a hipStream_t created by the caller on that device
 * (e.g. torch.cuda.current_stream().cuda_stream) or NULL for a private stream. */
int  movba_create(movba_handle **out, int device, void *stream, const movba_options *opt);
void movba_destroy(movba_handle *h);

/* Optimizer::LocalBundleAdjustment's solve (Optimizer.cc:754-755 + :757-775) in one call:
 * upload + run + download.  Also serves Optimizer::BundleAdjustment (Optimizer.cc:286-287). */
int  movba_lba_solve(movba_handle *h, const movba_lba_desc *desc, movba_lba_result *res);

/* The same in three phases, for callers that keep the window resident in HBM
 * (the window can be re-run after movba_lba_reset).  desc->stop is the one caller pointer the
 * phased API keeps: from movba_lba_upload until the next upload / solve on the handle, read by
 * every movba_lba_run in between; it must stay valid that long (or be NULL).  A raised flag
 * makes that run return MOVBA_STOPPED; the next run looks at the flag again. */
int  movba_lba_upload(movba_handle *h, const movba_lba_desc *desc);   /* host structure + H2D */
int  movba_lba_reset(movba_handle *h);                                /* restore uploaded state on device */
int  movba_lba_run(movba_handle *h);                                  /* LM loop on device; returns after stream sync */
int  movba_lba_download(movba_handle *h, movba_lba_result *res);      /* D2H + caller edge order */

/* movba_lba_run on the resident windows of n handles at once (multi-session serving: several independent windows on one
 * GPU): every kernel of an LM trial is one launch over the concatenated windows, with per-window LM state, so each window
 * takes exactly the steps of its solo run and returns bit-identical results; download each handle as usual.  The handles
 * must have been created on the same device and the same stream, and hold either stereo or monocular windows.  Windows
 * that end before the solve (MOVBA_EMPTY / MOVBA_NO_FIXED / stop flag up) report that through their own download.
 * The windows run in two groups half a trial out of phase; the second group's launches go to the device's COPY stream,
 * which every handle's upload also uses (one stream per device for all handles: each further stream of the process
 * competes for the few hardware queues).  Results never depend on it (tested with another handle uploading and solving
 * meanwhile), but an upload on ANOTHER handle of the device queues behind the running batch, and its copies between the
 * second group's kernels cost the batch its out-of-phase schedule for that trial: for the quoted batch throughput, upload
 * the windows first, then run.  Windows without an on-chip PCG are run one by one behind the batched launches. */
int  movba_lba_run_batch(movba_handle *const *handles, int32_t n);

/* Copy the optimised poses (n_poses x 7 f64) into a caller-owned DEVICE buffer on the
 * handle's stream — what the RCCL all-gather of independent windows sends. */
int  movba_lba_export_poses_device(movba_handle *h, void *dst_device, int64_t capacity_bytes);
/* The same without a copy of its own: register a DEVICE buffer once and every later movba_lba_run leaves
 * the optimised poses in it (written by the solve's last kernel; valid when movba_lba_run returns).
 * NULL unregisters.  The buffer must stay allocated while it is registered. */
int  movba_lba_set_pose_export(movba_handle *h, void *dst_device, int64_t capacity_bytes);

/* Pinned, device-visible host memory for result arrays (optional).  When `poses`, `points` or `chi2` of the
 * movba_lba_result handed to movba_lba_solve point into a block obtained here, the solve's last kernel writes those
 * results straight into them across the bus; otherwise they arrive in the handle's staging buffer and are copied out by
 * the calling thread.  (g2o's own results are in place too: vertex->estimate() is read after optimize(),
 * /root/reference/src/Optimizer.cc:822-838.)  Ordinary host memory for the CPU: read, write and keep it as long as
 * needed; release it with movba_host_free.  Returns NULL when no device is available or the allocation fails. */
void *movba_host_alloc(size_t bytes);
void  movba_host_free(void *p);

int  movba_get_profile(movba_handle *h, movba_profile *out);
int  movba_reset_profile(movba_handle *h);
int  movba_set_profile_mask(movba_handle *h, int32_t mask);

/* Host-only structure pass (no GPU needed): what movba_lba_upload derives from a window
 * before any H2D copy.  Exposed for the CPU test suite. */
typedef struct {
    int32_t n_free;             /* free AND active pose vertices (hessian blocks)            */
    int32_t n_pairs;            /* upper-triangle pose pairs sharing >= 1 point               */
    int64_t n_entries;          /* sum over points of d(d+1)/2 over free observers            */
    int32_t n_items;            /* schur work items (pair chunks)                             */
    int32_t max_degree;         /* max edges per point                                        */
    int32_t already_grouped;    /* 1 if caller edges were already grouped by point            */
    int32_t pcg_on_chip;        /* 1: reduced matrix stays in VGPRs during the PCG (k_pcg_rows) */
    int32_t pcg_overflow;       /* 1: some gather-list tails are read from an L2 copy          */
    int32_t pcg_max_wave_entries; /* largest number of gather entries dealt to one wave       */
    int32_t n_row_entries;      /* gather-list entries incl. padding                          */
    int32_t n_sched_slots;      /* slots of the schur launch schedule (8 XCD segments)        */
    int32_t sched_items;        /* work items found in the schedule (must equal n_items)      */
    int32_t sched_max_permille; /* heaviest XCD segment / mean segment weight, x1000          */
    int32_t slots_ok;           /* pose-major edge slots are a bijection onto 0..E_free-1     */
    int32_t reordered;          /* 1: the free keyframes were renumbered by covisibility (free_index is the new numbering) */
    int32_t pad_s;
} movba_structure_info;
int  movba_structure_probe(const movba_lba_desc *desc, movba_structure_info *info,
                           int32_t *edge_perm /* E or NULL */, int32_t *free_index /* n_poses or NULL */);

/* Host only (no device needed): the static schedule of the one-launch direct solver that stands in for the reference's
 * LinearSolverCSparse factorisation (Optimizer.cc:535) for a reduced system of n_block_cols 48-wide block columns — for
 * tests, which replay it against its own flags.  info[0..3] = supported, workgroups, tile slots per workgroup, tasks;
 * task_ptr (workgroups + 1) and tasks (8 int32 each: op, slot, I, K, k, 0, 0, 0) are filled up to their capacities. */
int  movba_dense_plan_probe(int32_t n_block_cols, int32_t max_groups, int32_t max_slots, int32_t info[4],
                            int32_t *task_ptr, int32_t task_ptr_cap, int32_t *tasks, int32_t tasks_cap);

/* Pose-only optimisation behind Optimizer::PoseOptimization (Optimizer.cc:397-459), over the
 * reference's EdgeSE3ProjectXYZOnlyPose (include/OptimizableTypes.h:30-58). */
typedef struct {
    int32_t n;                  /* 2D-3D matches (Frame::N non-null, Optimizer.cc:404-413)    */
    const double *Xw;           /* n x 3 MapPoint::GetWorldPos                                */
    const double *obs;          /* n x 2 mvKeys[i].pt                                         */
    const double *inv_sigma2;   /* n or NULL (= 1)                                            */
    double fx, fy, cx, cy;
    double pose0[7];            /* initial Tcw                                                */
    double huber_delta;         /* reprojection threshold in px                               */
    double chi2_gate;           /* threshold^2                                                */
    int32_t rounds;             /* 4                                                          */
    int32_t its_per_round;      /* 10                                                         */
    /* Hypothesis stage in front of the LM, so that the result does not depend on pose0 (the reference calls
     * cv::solvePnPRansac with useExtrinsicGuess = false, Optimizer.cc:437): ransac_iters minimal P3P samples
     * (iterationCount of PoseOptimization, 50 by default), drawn from ransac_seed; every candidate pose is scored on
     * all matches by MAGSAC++'s sigma-consensus loss (flag 38 = cv::USAC_MAGSAC, Optimizer.cc:437; sigma_max from
     * chi2_gate: pose_kernels.hip) and the one of lowest loss among those with at least 4 matches inside chi2_gate starts
     * the LM.  0: LM from pose0. */
    int32_t ransac_iters;       /* <= MOVBA_MAX_RANSAC_ITERS                                  */
    uint32_t ransac_seed;
    /* `confidence` of cv::solvePnPRansac (Optimizer.cc:437; 0.95 in TartanAir.yaml): the standard stopping rule
     * N = log(1 - confidence) / log(1 - w^3), w = inlier ratio of the best hypothesis so far.  The device scores all
     * ransac_iters samples at once; the rule decides which of them a sequential RANSAC over the same samples would have
     * drawn: only those are eligible (ransac_samples_used in the result).  <= 0 or >= 1: all samples are eligible.
     * lo_iters > 0: one local-optimisation step on the winner (USAC's LO): an LM refit weighted by the sigma-consensus
     * weights w(r) (lo_iters iterations), kept when its sigma-consensus loss is lower.
     * rounds x its_per_round: the motion-only LM with re-classification behind the hypothesis stage is ORB-SLAM's
     * PoseOptimization scheme (4 x 10); MoV-SLAM's own PoseOptimization (Optimizer.cc:397-459) takes solvePnPRansac's pose
     * as it is - rounds = 0 gives exactly that. */
    double  confidence;
    int32_t lo_iters;
    int32_t pad_p;
} movba_pose_desc;

#define MOVBA_MAX_RANSAC_ITERS 256

typedef struct {
    double   pose[7];
    uint8_t *outlier;           /* n out (Frame::mvbOutlier, Optimizer.cc:452-456)            */
    double  *chi2;              /* n out or NULL                                              */
    int32_t  n_inliers;         /* return value of PoseOptimization (Optimizer.cc:458)        */
    int32_t  status;
    int32_t  ransac_inliers;    /* inliers of the best hypothesis (0: stage off, or no candidate with >= 4)  */
    int32_t  lm_iters;          /* LM iterations run over the 4 rounds (g2o stops a round early when the cost stops changing) */
    double   ransac_pose[7];    /* the pose the LM started from                               */
    int32_t  ransac_samples_used;   /* minimal samples the stopping rule admitted (<= ransac_iters)                  */
    int32_t  lo_accepted;       /* 1: the local-optimisation refit replaced the winning hypothesis                   */
    int32_t  lo_inliers;        /* inliers of the pose the LM started from, at chi2_gate (after the LO step)          */
    int32_t  pad_q;
} movba_pose_result;

int  movba_pose_opt(movba_handle *h, const movba_pose_desc *desc, movba_pose_result *res);
/* The minimal samples movba_pose_opt draws for (n matches, n_hyp, seed): n_hyp x 3 distinct match indices.  Host only. */
int  movba_pose_ransac_samples(int32_t n, int32_t n_hyp, uint32_t seed, int32_t *out);

#ifdef __cplusplus
}
#endif
#endif /* MOVBA_H */
