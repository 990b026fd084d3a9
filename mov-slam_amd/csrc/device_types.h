// Device-side view of one uploaded local-BA window (all pointers are HBM addresses).
// Layout rationale in DESIGN.md §3: SoA fp64, edges grouped by map point, two state
// buffers (current linearisation point / trial state) so g2o's push()/pop()
// (OptimizationAlgorithmLevenberg::solve, called from /root/reference/src/Optimizer.cc:755)
// becomes an index flip.
#pragma once
#include <cstdint>

#include "dense_plan.h"
#include "structure.h"

namespace movba {

#ifndef MOVBA_PART_STRIDE
#define MOVBA_PART_STRIDE 72
#endif
// coarse level of the k_pcg_rows preconditioner: one aggregate per wave, 12 dofs each (6 constant + 6 linear-in-keyframe-index modes)
constexpr int kCoarsePerAgg = 12;
constexpr int kCoarseDim = kCoarsePerAgg * (512 / 64);
constexpr int kPartStride = MOVBA_PART_STRIDE;     // doubles per schur work-item partial
// partial layout: [0,36) sum of B_i Dinv B_j^T (6x6 row-major)
//                 [36,42) sum of B_i Dinv b_l           (diagonal pairs only)
//                 [42,63) upper triangle of Hpp_ii      (diagonal pairs only)
//                 [63,69) b_p,i                          (diagonal pairs only)
// packed off-diagonal schur entries: 22 + 22 bits of pose-major slots, 20 bits of map point (8 bytes instead of 12 per entry)
constexpr int kEntPackSlots = 1 << 22, kEntPackPoints = 1 << 20;
__host__ __device__ inline unsigned long long ent_pack(int si, int sj, int l) { return (unsigned long long)(unsigned)si | ((unsigned long long)(unsigned)sj << 22) | ((unsigned long long)(unsigned)l << 44); }

// bound of the deciding wave's wait for the back-substitution pass's records (10 ns ticks: 30 s, the host watchdog's order)
constexpr unsigned long long kDecideWaitTicks = 3000000000ull;

constexpr int kPointGroup = 8;      // lanes cooperating on one map point
#ifndef MOVBA_POINT_BLOCK
#define MOVBA_POINT_BLOCK 256
#endif
constexpr int kPointBlock = MOVBA_POINT_BLOCK;    // threads per block of the point kernels
constexpr int kPointsPerBlock = kPointBlock / kPointGroup;
constexpr int kPointRed = 2 * (kPointBlock / 64) > 8 ? 2 * (kPointBlock / 64) : 8;     // doubles of the point kernels' reduction strip
constexpr int kMaxTrace = MOVBA_MAX_TRACE;

struct DevState {
    double *pose;    // NP x 7  (qx qy qz qw tx ty tz)
    double *Rt;      // NP x 12 (R row-major, t)
    double *point;   // P x 3
    double *Hll;     // P x 6   (xx xy xz yy yz zz), undamped
    double *bl;      // P x 3
    double *Fpart;   // n_pt_blocks robust-cost partials of this state
    // per-edge records of the edges of FREE keyframes, POSE-major (DevWindow::slot), written by k_point for k_schur:
    double *erecA;   // E_free x 4: camera-frame point and weight (Xc.x Xc.y Xc.z, w = rho1 * inv_sigma2): everything both Jacobians need
};

// LM controller state, lives in HBM; every kernel reads it, k_decide/k_pcg/k_lambda_init write it.
struct Ctrl {
    double lambda, nu, F0, cost0;
    double tr_lambda[kMaxTrace], tr_f0[kMaxTrace], tr_f1[kMaxTrace], tr_rho[kMaxTrace];
    int32_t tr_accept[kMaxTrace], tr_pcg[kMaxTrace];
    int32_t it, qmax, cur, done;
    int32_t n_solves, last_rejected, iters_done, n_trace;
    int32_t pcg_fail, pcg_last_iters, pcg_total_iters, n_outliers;
    // reduced-solve bookkeeping: solver_mode 0 = PCG (k_pcg_rows), 1 = direct (dense Cholesky, dense_solve.hip); sticky from
    // the first trial whose PCG broke down or ran into its iteration cap.  done == 2 parks the solve until the host has
    // queued the direct kernels for that trial (HostStatus::pause_seq).
    int32_t solver_mode, n_pause, n_direct, n_chol_fail;
    int32_t direct_from;                // n_solves at the switch to the direct solver (-1: never)
    int32_t n_sync_timeouts;            // one-launch direct solver: solves in which a workgroup gave up waiting for another (dense_persist.hip)
    int32_t n_band;                     // trials solved by the single-workgroup banded factorisation (band_kernel.hip)
    int32_t pad_c[1];
    // diagnostic build only (-DMOVBA_CLOCK_STAMP): shader cycles / 100 MHz ticks spent in k_pcg_rows
    unsigned long long dbg_cycles, dbg_ticks, dbg_seg[8], dbg_seg2[8], dbg_wseg[8][8];
    unsigned long long dbg_xs[8];       // two-stream path, 10 ns ticks: [0] time of the schur pass's last item flag of the trial (atomic max), then sums over the
                                        // trials of: [1] last flag -> partials in registers, [2] -> CG starts, [3] -> CG ends, [4] -> done word stored; [5] trials
};

// Written by k_decide into pinned host memory so the host can keep the queue fed
// without a stream synchronise per trial.
struct HostStatus {
    // device -> host, ONE word so that one store (one PCIe write acknowledgement) publishes a consistent triple:
    // bits 0-23 trials_done, 24-47 completed outer LM iterations (the host does not queue trials past the last possible
    // one), bit 48 done
    volatile uint64_t progress;
    volatile int32_t stop;      // host -> device: forceStopFlag seen by the host poll
    // device -> host: number of times the solve has parked itself (Ctrl::done == 2) because the PCG failed and the direct
    // kernels for the trial must be queued; the host answers each increment once
    volatile int32_t pause_seq;
    static constexpr uint64_t pack(int trials_done, int it, int done)
    {
        return (uint64_t)(uint32_t)trials_done | ((uint64_t)(uint32_t)it << 24) | ((uint64_t)(done ? 1 : 0) << 48);
    }
};

// Dense form of the reduced system for the direct solver (dense_solve.hip): lower block triangle in tiles of kDenseNB x
// kDenseNB doubles (row-major inside a tile), tile (I, J), I >= J, at index I (I + 1) / 2 + J.  Row tile `ntile` carries
// the right-hand side in its first row (forward substitution by augmentation: it leaves the factorisation as L^-1 b).
constexpr int kDenseNB = 48;        // 8 pose blocks
struct DenseSys {
    double *tiles;
    double *diagL;              // ntile x NB x NB: Cholesky factors of the diagonal tiles
    double *xsol;               // ntile x NB: solution of the multi-launch back substitution (large systems)
    const int32_t *pid;         // nfree x nfree: pair id of block (i <= j) or -1
    const int32_t *prange;      // nfree x nfree x 2: first / one-past-last schur work item of block (i <= j), (0, 0) where there is no pair
                                // (pid and pair_item_start folded into one load level for the one-launch solver's assembly)
    int32_t *fail;              // != 0: a pivot was not positive (the trial is rejected like a failed CSparse factorisation)
    int32_t ntile, n;           // column tiles; unknowns (6 nfree)
    // one-launch form (dense_persist.hip): the static schedule (dense_plan.h), the hand-off flags (compared with the launch's
    // epoch), the two failure words (bad pivot, wait given up), the back substitution's per-tile contributions
    const int32_t *task_ptr;
    const DenseTask *tasks;
    unsigned *flags, *failw;
    double *contrib;            // ntile x ntile x NB
    unsigned *ctag;             // ntile x NB x 4 words: the sub-diagonal contributions c(J+1, J) as self-announcing 16-byte records
                                // (value, epoch, check word), behind the flags: zeroed with them
    unsigned long long *stamps; // diagnostic (MOVBA_DENSE_STAMPS=1): per task 3 readings of the 100 MHz clock (start, wait over, end), else null
    int32_t G, slots;           // workgroups of the launch, LDS tile slots per workgroup (G == 0: multi-launch path only)
};

struct DevWindow {
    int32_t NP, P, E, nfree, npairs, nitems, n_pt_blocks, max_iters;
    uint32_t flags;
    int32_t max_trials;
    double fx, fy, cx, cy, huber_delta, chi2_gate;
    // structure
    const int32_t *g_pose, *g_point, *pt_start, *perm, *hidx, *free_pose;
    const double *obs;      // E x 2 (grouped order)
    const double *isig;     // E
    const double *obs_r;    // E: right-image u of stereo observations, < 0 = monocular edge (stereo windows only)
    double *obs_pm;         // E_free x 2: the observations of the free keyframes' edges once more, pose-major (by slot), written by the
                            // first linearisation: the diagonal schur entries rebuild the weighted residual -w e from them and the
                            // edge record instead of reading a second per-edge record that every trial would have to write
    double *obsr_pm;        // E_free: right-image u by slot (stereo windows only)
    double bf;              // KeyFrame::mbf
    const double *kcam;     // per-keyframe intrinsics, NP x 8 (fx fy cx cy bf 0 0 0), or nullptr: fx .. cy / bf above hold for every keyframe
                            // (the reference gives every edge its keyframe's camera: src/Optimizer.cc:664, 690-695)
    int32_t stereo, pad2;   // window has >= 1 stereo edge: 3-row kernels
    const int32_t *slot;    // E: pose-major slot of a grouped edge (-1: edge of a fixed pose)
    // k_schur entry lists.  Diagonal pair of free pose h: its entry k IS pose-major slot k (the diagonal pairs come first
    // and list every edge of the pose in slot order), only the map point of a slot is stored.  Off-diagonal pairs: entry k
    // of the global numbering at k - n_diag in three arrays (slot of the edge of pose i, of pose j, map point): 12 bytes.
    const int32_t *ent_i, *ent_j, *ent_l, *slot_point;
    const unsigned long long *ent64;    // the off-diagonal entries packed (slot of i: bits 0-21, slot of j: 22-43, point: 44-63) when the window
                                        // allows (kEntPackSlots / kEntPackPoints), else null and the three arrays above hold them
    int32_t n_diag, pad5;   // entries of the diagonal pairs = edges of free poses
    const Item *items;
    const SchedItem *sched; // k_schur launch schedule: 8 x sched_per_xcd slots (structure.h)
    int32_t sched_per_xcd, pad3;
    const int32_t *pair_i, *pair_j, *pair_item_start, *row_ptr;
    const RowEnt *row_ent;
    // k_pcg_rows: per (wave, lane, slot) plan {pair id or -1, transposed, col*6, first item, end item} (host-built)
    const int32_t *lane_plan;   // kPcgRowsThreads x 3 x 4 int32: the thread's two blocks {pair id or -1, col * 6 | transposed << 30, first item, end item},
                                // then, for the owner lane of a scalar row, {first item, end item of its keyframe's diagonal pair, row_ptr of its block row, of the next}
    // coarse level of the PCG preconditioner
    int32_t n_agg, n_cblk;
    const int32_t *cblk_g, *cblk_h, *cblk_ptr, *cblk_ent, *cblk_ij;
    const int32_t *multi_pairs; // off-diagonal pairs cut into several work items
    int32_t n_multi, pad4;
    // state
    DevState st[2];
    const double *pose0, *point0;   // uploaded initial state (for reset)
    // reduced system
    double *part;       // nitems x kPartStride: k_schur work-item partials
    // what the schur pass leaves once more in the layout the on-chip PCG's setup reads (pcg_kernel.hip), beside `part`:
    double *rec_d;      // per free keyframe h and work item s of its diagonal pair, at (h rec_slots + s) 48: 6 rows x 8 doubles: row a of
                        // Hpp - sum B Dinv B^T (6), then (sum B Dinv b_l)_a and (b_p)_a
    int32_t rec_slots, pad7;    // most work items a diagonal pair of this window is cut into
    double *img_b;      // 72 x kPcgRowsThreads: element q (oriented) of the block that PCG thread t holds in slot k at ((36 k + q) 512 + t), for
                        // off-diagonal pairs that are one work item (SchedItem::dst_a / dst_b say where an item's block goes)
    double *blocks_ov;  // overflow windows only: oriented copies of the blocks of the gather-list tails (entry e at 36 e)
    double *blocks_c;   // npairs x 36: the coarse-level workgroup's own copy of S (coarse_level.h)
    float *aci;         // 2 x kCoarseDim x kCoarseDim: inverse coarse matrices (by trial parity), rounded to fp32 by the workgroup that
                        // builds them: the solver keeps them in LDS in that precision (pcg_kernel.hip) and takes in half the bytes
    int32_t *aci_tag;   // 2: trial that produced aci[parity], -1 = unusable
    double *ac_prev;    // kCoarseDim^2 + 2: coarse matrix of the previous build, then its lambda and its trial (coarse_level.h)
    double *blocks;     // npairs x 36 upper blocks of S (damped), diagonal pairs first
    double *bp;         // 6 nfree
    double *xp;         // 6 nfree
    double *scale_part; // n_pt_blocks + 1
    unsigned *dec_rec;  // 2 n_pt_blocks records of 16 bytes (handoff.h): every workgroup of the back-substitution pass hands its cost and scale
                        // partials to the pass's deciding workgroup as tagged records (tag = trial + 1)
    double *hmax_part;  // n_pt_blocks
    Ctrl *ctrl;
    HostStatus *hstat;  // device view of the pinned status block
    double *pose_export; // caller's registered device buffer for the final poses (NP x 7), or null; written by k_finalize
    Ctrl *ctrl_out;     // device view of the host's pinned copy of Ctrl: written by k_finalize (no copy engine at the end of a solve)
    // outputs (caller edge order)
    double *out_chi2;
    uint8_t *out_outlier;
    // direct solver
    DenseSys dense;
    int32_t direct_only, lds_poses;     // no on-chip PCG for this window (size); keyframe poses fit the point kernels' LDS staging
    // bound of every device-side wait of one workgroup for another, in ticks of the 100 MHz clock: 20 ms (a workgroup that gives up
    // marks the solve - Ctrl::n_sync_timeouts - and the host runs it again on the paths that wait for nothing; MOVBA_TEST_WAIT_TICKS
    // shortens the first attempt's bound so that tests can see that happen)
    unsigned long long wait_ticks;      // bound of the waits inside the one-launch direct solver (10 ns ticks; 20 ms)
};

// Device view of the structure pass (struct_kernels.hip)
struct StructDev {
    int32_t P, nfree, nchunks, NP;
    const int32_t *g_pose, *pt_start, *hidx;
    int32_t *cntw;              // nchunks x nfree^2: per-chunk counts, then their exclusive scan over the chunks
    int32_t *cnt;               // nfree^2: entries per pair bin
    int32_t *error;             // set when a keyframe observes a point twice
    int32_t *ent0;              // nfree^2: first off-diagonal entry of a pair bin (i < j; k_struct_counts_out -> fill)
    const int32_t *slot;        // E: pose-major slot of a grouped edge (fill)
    int32_t *ent_i, *ent_j, *ent_l;     // off-diagonal entry lists (fill), or ...
    unsigned long long *ent64;          // ... the packed form (non-null: used instead)
    const int32_t *abort;               // non-null: two words of the device grouping pass (BasicDev::info[0], [1]); either one set = the
                                        // points' ranges are not to be trusted (an index out of range, edges not grouped): the pass leaves at once
};

// The grouping pass on the device (struct_kernels.hip: k_basic_hist, k_basic_index, k_basic_scan): what structure.cpp's
// build_basic derives on the host in one pass over the caller's edges - validation, the points' edge ranges, edges per
// keyframe, hessian indices, first pose-major slots, every edge's rank among its keyframe's edges - from the caller's index
// arrays where the upload's first copy put them.
constexpr int kBasicBlock = 256;        // edges per workgroup of k_basic_hist (and of k_slot_point, which completes the slots)
struct BasicDev {
    int32_t E, P, NP, nblk;
    const int32_t *edge_pose, *edge_point;      // the caller's index arrays in the arena (the upload's first copies)
    const uint8_t *pose_fixed;                  // (host memory: the staging buffer is mapped)
    int32_t *pt_start;          // P + 1
    int32_t *rank;              // E: the edge's rank among the edges of its keyframe INSIDE its workgroup's 256 edges
    int32_t *H;                 // nblk x NP: edges of keyframe k in workgroup b, then (k_basic_scan) in the workgroups before b
    int32_t *pose_edges;        // NP (zeroed): edges per keyframe
    int32_t *hidx, *base, *free_pose;           // NP, NP + 1, <= NP
    int32_t *info;              // kBasicInfo words (zeroed): [0] an index out of range, [1] edges not grouped by point,
                                // [2] free keyframes with edges, [3] their edges, [4] fixed keyframes, [5] workgroups of k_basic_hist done
};
constexpr int kBasicInfo = 8;

// k_ingest (struct_kernels.hip): arrays of the caller that lie in mapped host memory, read across the bus by the kernel itself
// and left in the arena.  Up to eight pieces per launch, in order; `counter` is raised once by every workgroup when it is
// through, and a launch may hold its reads back until the counter has reached `wait_for` (the launch in front of it is through:
// the bus is the bottleneck, and whoever the pair structure waits for gets it first).
struct IngestSeg { const void *src; void *dst; unsigned long long bytes; };
// `zero` / `zero_words`: a piece of device memory the launch clears first (the grouping pass's counters: a hipMemsetAsync in
// front of the launch cost the stream ~6 us and the calling thread another launch call)
struct IngestArgs { IngestSeg seg[8]; int32_t nseg, pad_i; unsigned *counter; unsigned wait_for; unsigned zero_words; unsigned *zero; };

struct PcgParams {
    double rel_tol;
    int32_t max_iters;
    int32_t wave_row0[17];      // k_pcg_rows: wave wv owns block rows [wave_row0[wv], wave_row0[wv+1])
    int32_t overflow;           // k_pcg_rows: some wave has more gather entries than fit in VGPRs
    int32_t use_coarse;         // k_pcg_rows: add the aggregate coarse-level correction to block-Jacobi (1: lagged, 2: fresh)
    int32_t wave_ent0[17];      // row_ptr[wave_row0[wv]]: first gather-list entry of the wave's rows (so that the kernel's setup needs no
    int32_t nrowent;            // load for them), and row_ptr[nfree]
    int32_t padded;             // k_pcg_rows: no block row has more than kPcgPlanOwnBatch entry pairs and nothing overflows: the pair
                                // sums of the mat-vec sit in zero-padded slots by row (pcg_kernel.hip, PADDED), no masks in the loop
    int32_t pad_pp;
};

// Batched launches over n resident windows (movba_lba_run_batch): device arrays of the windows' views and PCG plans, and
// per kernel the prefix of the windows' block counts (n + 1 entries each).
struct BatchDev {
    const DevWindow *wins;
    const PcgParams *pps;
    const int32_t *blk_point, *blk_schur, *blk_final, *blk_init;
    const int32_t *band_bw;     // per window: half bandwidth of its reduced matrix in blocks (batches solved by k_band_b), or null
    int32_t n, pad;
};

}  // namespace movba
