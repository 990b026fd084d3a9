// Hand-written HIP kernels (gfx950 / CDNA4, wave64, fp64) for the local-BA hot path.
//
// What they replace: the work g2o does inside optimizer.optimize(10)
// (/root/reference/src/Optimizer.cc:754-755) with the reference's own edge callbacks
//   EdgeSE3ProjectXYZ::computeError   include/OptimizableTypes.h:103-109
//   EdgeSE3ProjectXYZ::linearizeOplus src/OptimizableTypes.cpp:158-180
//   Pinhole::project / projectJac     src/CameraModels/Pinhole.cpp:36-43, 77-88
// and the outlier gate of src/Optimizer.cc:757-775.
//
// Kernel chain per LM trial (all on one stream, LM control stays on the device):
//   k_schur  (one wave per pose-pair chunk)  -> partial 6x6 blocks of the reduced system
//   k_pcg    (one workgroup)                 -> assemble S, block-Jacobi PCG, trial poses
//   k_point<true> (8 lanes per map point)    -> back-substitution, trial points, errors and
//                                               the linearisation (Hll, bl, weights) at the trial state; its last
//                                               workgroup to arrive: gain ratio, accept/reject, lambda schedule
// None of this is GEMM-shaped (block-sparse 6x3 / 3x3 / 6x6 products over a point
// graph), so there is no MFMA: the kernels are gather/stream kernels bound by HBM/L2
// traffic and launch latency.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>

#include "device_math.h"
#include "device_types.h"
#include "handoff.h"
#include "kernels.h"

namespace movba {

// --------------------------------------------------------------------------------
// k_init_pose: uploaded poses -> state 0 (normalised like SE3Quat's constructor), Rt cache; uploaded points -> state 0
// (in the same launch: a separate device-to-device copy costs a launch gap of its own at the start of every solve)
// --------------------------------------------------------------------------------
__device__ __forceinline__ void init_pose_body(const DevWindow &w, int bid, int nblk)
{
    const int i = bid * blockDim.x + threadIdx.x;
    for (int k = i; k < 8 * w.n_pt_blocks; k += nblk * blockDim.x) w.dec_rec[k] = 0u;  // the point pass's hand-off records: no tag of an earlier run may fit
    {
        const double2 *src = reinterpret_cast<const double2 *>(w.point0);
        double2 *dst = reinterpret_cast<double2 *>(w.st[0].point);
        const int n2 = (3 * w.P) >> 1;
        for (int k = i; k < n2; k += nblk * blockDim.x) dst[k] = src[k];
        if (i == 0 && ((3 * w.P) & 1)) w.st[0].point[3 * w.P - 1] = w.point0[3 * w.P - 1];
    }
    if (i == 0) {
        Ctrl *c = w.ctrl;
        c->lambda = 0.0; c->nu = 2.0; c->F0 = 0.0; c->cost0 = 0.0;
        c->it = 0; c->qmax = 0; c->cur = 0; c->done = (w.max_iters <= 0) ? 1 : 0;
        c->n_solves = 0; c->last_rejected = 0; c->iters_done = 0; c->n_trace = 0;
        c->pcg_fail = 0; c->pcg_last_iters = 0; c->pcg_total_iters = 0; c->n_outliers = 0;
        // (solver_mode / direct_from: k_lambda_init - this kernel may run before the upload has chosen the reduced solver)
        c->n_pause = 0; c->n_direct = 0; c->n_chol_fail = 0; c->n_band = 0; c->n_sync_timeouts = 0;
        w.aci_tag[0] = -1; w.aci_tag[1] = -1;
        w.ac_prev[kCoarseDim * kCoarseDim + 1] = -1.0;
        c->dbg_cycles = 0; c->dbg_ticks = 0;
        for (int k = 0; k < 8; ++k) { c->dbg_seg[k] = 0; c->dbg_seg2[k] = 0; for (int q = 0; q < 8; ++q) c->dbg_wseg[k][q] = 0; }
    }
    if (i >= w.NP) return;
    double q[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) q[k] = w.pose0[7 * i + k];
    quat_normalize_exact(q);
    double R[9];
    quat_to_R(q, R);
#pragma unroll
    for (int k = 0; k < 7; ++k) w.st[0].pose[7 * i + k] = q[k];
#pragma unroll
    for (int k = 0; k < 9; ++k) w.st[0].Rt[12 * i + k] = R[k];
    w.st[0].Rt[12 * i + 9] = q[4]; w.st[0].Rt[12 * i + 10] = q[5]; w.st[0].Rt[12 * i + 11] = q[6];
}

// The camera of an edge is its keyframe's (src/Optimizer.cc:664: e->pCamera = pKFi->mpCamera; :690-695: e->fx .. e->bf from
// pKFi): one set of intrinsics for the whole window in every shipped MoV-SLAM configuration (DevWindow::fx ..), a table by
// keyframe when the caller hands one (movba_lba_desc::cam_kf / bf_kf).  The choice is a wave-uniform branch.
struct Cam { double fx, fy, cx, cy, bf; };
// PERKF = false: the window's one camera, straight from the kernel's scalar arguments (the instantiations every shipped
// configuration runs: carried as a runtime choice the table's values took vector registers in k_schur, which has none to
// spare - 17 -> 22 us per launch at cfg3); PERKF = true: row ip of DevWindow::kcam
template <bool PERKF>
__device__ __forceinline__ Cam cam_of(const DevWindow &w, int ip)
{
    if (!PERKF) return Cam{ w.fx, w.fy, w.cx, w.cy, w.bf };
    const double *k = w.kcam + 8 * (size_t)ip;
    return Cam{ k[0], k[1], k[2], k[3], k[4] };
}
__device__ __forceinline__ Cam cam_of_rt(const DevWindow &w, int ip) { return w.kcam ? cam_of<true>(w, ip) : cam_of<false>(w, ip); }

__global__ void k_init_pose(DevWindow w) { init_pose_body(w, blockIdx.x, gridDim.x); }

// --------------------------------------------------------------------------------
// decide_body: one wave - wave 0 of the extra workgroup of the back-substitution pass (block n_pt_blocks of point_body).
// The accept / reject logic and lambda schedule of OptimizationAlgorithmLevenberg::solve plus the loop conditions of
// SparseOptimizer::optimize, restated as a state machine that advances by one trial per
// point pass.  Publishes progress to pinned host memory.
// (A launch of its own until round 4: 4.6 us per trial plus a launch boundary.  Now the pass's workgroups hand their cost
// and scale partials to this wave INSIDE the launch, as tagged 16-byte records (handoff.h) which its lanes poll: the
// decision is taken ~1 us behind the last workgroup's partials, and the launch ends with it.)
// --------------------------------------------------------------------------------
__device__ __forceinline__ void decide_body(const DevWindow &w, int cur)
{
    Ctrl *c = w.ctrl;
    const int lane = threadIdx.x;
    // Lane-strided partial sums, records polled 10 blocks deep per lane (10 x 64 covers cfg3's 625 blocks in one round),
    // added in a fixed order once a round is complete.  None of the producers waits for anything, and this wave's workgroup is
    // the LAST of the grid (every producer was dispatched before it), so the wait cannot deadlock; it can only be slow - another
    // process time-slicing the GPU, a preempted queue, a very long batched pass - and a slow pass must not cost the caller the
    // solve: the bound is of the host watchdog's order (kDecideWaitTicks, 30 s), there for a device that has stopped altogether.
    constexpr int kDeep = 10;
    const unsigned tag = (unsigned)c->n_solves + 1u;
    // What the decision reads besides the partials is requested HERE, ahead of the wait: the caller's stop flag sits in host
    // memory (one PCIe round trip, ~1.5 us) and the controller's words are cold lines of L2 - behind the wait each of them was a
    // dependent round trip on the path of every trial.
    const int stop = __hip_atomic_load(&w.hstat->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const double F0 = c->F0, lambda0 = c->lambda, nu0 = c->nu;
    const int tr = c->n_trace, qmax0 = c->qmax, it0 = c->it, nsolves0 = c->n_solves;
    // (what the solver left for this wave - the pose part of the scale, its failure flag, its iteration count: final before
    //  this launch started: the solve ran ahead of it on the same stream)
    const double scale_pose = w.scale_part[w.n_pt_blocks];
    const int pcg_fail = c->pcg_fail, pcg_iters = c->pcg_last_iters;
    const __amdgpu_buffer_rsrc_t rr = hx_rsrc(w.dec_rec, 32u * (unsigned)w.n_pt_blocks);
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
    double F1 = 0.0, scale = 0.0;
    bool good = true;
    for (int base = 0; base < w.n_pt_blocks && good; base += 64 * kDeep) {
        const int k0 = base + lane;
        double f1[kDeep], sv[kDeep];
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int u = 0; u < kDeep; ++u) {
                const unsigned k = (unsigned)min(k0 + 64 * u, w.n_pt_blocks - 1);
                ok &= hx_ld_tagged(rr, 2u * k, tag, f1[u]);
                ok &= hx_ld_tagged(rr, 2u * k + 1u, tag, sv[u]);
            }
            if (__all(ok)) break;
            if (__builtin_amdgcn_s_memrealtime() - t_start > kDecideWaitTicks) { good = false; break; }
            __builtin_amdgcn_s_sleep(2);
        }
#pragma unroll
        for (int u = 0; u < kDeep; ++u) {
            const bool in = k0 + 64 * u < w.n_pt_blocks;
            F1 += in ? f1[u] : 0.0; scale += in ? sv[u] : 0.0;
        }
    }
    if (!good) {        // (never seen; the download then reports MOVBA_ERR_DEVICE_WAIT through n_sync_timeouts)
        if (lane == 0) {
            c->n_sync_timeouts += 1; c->done = 1;
            __hip_atomic_store(&w.hstat->progress, HostStatus::pack(c->n_solves, c->it, 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    F1 = wave_sum(F1);
    scale = wave_sum(scale);
    if (lane != 0) return;
    scale += scale_pose;
    // g2o on a failed solve: tempChi = DBL_MAX (std::numeric_limits<double>::max(): FINITE), scale = 1e-3, nothing updated.  With a
    // finite current cost rho is hugely negative and the trial is rejected; with an infinite one (a point on a keyframe's z = 0
    // plane) rho = +inf and the trial counts as ACCEPTED - of a state that has not moved, whose cost the next iteration's
    // computeActiveErrors() finds again: F1_eval, what this pass has just summed over the unmoved trial state.
    const double F1_eval = F1;
    if (pcg_fail) { F1 = DBL_MAX; scale = 0.0; }
    scale += 1e-3;
    const double rho = (F0 - F1) / scale;
    if (tr < kMaxTrace) {
        c->tr_lambda[tr] = lambda0; c->tr_f0[tr] = F0; c->tr_f1[tr] = F1; c->tr_rho[tr] = rho;
        c->tr_pcg[tr] = pcg_iters;
    }
    bool lambda_ok = true;
    int accepted = 0;
    if (rho > 0.0 && isfinite(F1)) {
        const double t = 2.0 * rho - 1.0;
        double alpha = 1.0 - t * t * t;                // (pow(tmp, 3) in g2o: a library call of ~100 instructions here)
        alpha = fmin(alpha, 2.0 / 3.0);
        c->lambda = lambda0 * fmax(1.0 / 3.0, alpha);
        c->nu = 2.0;
        c->F0 = pcg_fail ? F1_eval : F1;
        c->cur = cur ^ 1;                    // discardTop(): the trial state becomes current
        c->last_rejected = 0;
        accepted = 1;
    } else {
        const double lam = lambda0 * nu0;
        c->lambda = lam;
        c->nu = nu0 * 2.0;
        c->last_rejected = 1;                // pop(): keep the current state
        lambda_ok = isfinite(lam);
    }
    if (tr < kMaxTrace) { c->tr_accept[tr] = accepted; c->n_trace = tr + 1; }
    const int nsolves = nsolves0 + 1, qmax = qmax0 + 1;
    int it = it0, qnext = qmax;
    c->n_solves = nsolves;
    const bool more_trials = lambda_ok && (rho < 0.0) && (qmax < w.max_trials) && !stop;
    int done = 0;
    if (!more_trials) {
        c->iters_done = it0 + 1;
        if (qmax == w.max_trials || rho == 0.0 || !lambda_ok) done = 1;       // Terminate
        else {
            it = it0 + 1;
            c->it = it;
            qnext = 0;
            if (it >= w.max_iters || stop) done = 1;
        }
    }
    c->qmax = qnext;
    c->done = done;
    // one word, one store: the host sees a consistent (trials_done, it, done), and the launch ends behind ONE write
    // acknowledgement from host memory instead of a chain of them
    __hip_atomic_store(&w.hstat->progress, HostStatus::pack(nsolves, it, done), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// --------------------------------------------------------------------------------
// k_point<BACKSUB>: 8 lanes per map point, edges of a point are contiguous.
//   BACKSUB=false : evaluate errors + linearise (Hll, bl, per-edge Xc/weight) at state cur
//   BACKSUB=true  : x_l = Dinv (b_l - sum_i B_il^T xp_i)  (BlockSolver::solve back-substitution),
//                   trial point X + x_l, then the same evaluation at the trial state cur^1
// Pose rotations/translations and the pose increments are staged in LDS.
// --------------------------------------------------------------------------------
// LDSP: the keyframe rotations (and, for BACKSUB, the pose increments and hessian indices) are staged in LDS; windows
// with more keyframes than fit (~850) read them through L2 instead (same arithmetic, same results).
template <bool BACKSUB, bool STEREO, bool LDSP, bool PERKF = false>
__device__ __forceinline__ void point_body(const DevWindow &w, int bid)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const Ctrl *c = w.ctrl;
#ifdef MOVBA_CLOCK_STAMP
    unsigned long long pst[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    pst[0] = __builtin_amdgcn_s_memrealtime();
#define PSTAMP(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); pst[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PSTAMP(k) do { } while (0)
#endif
    // (the map point's edge range does not depend on the LM state: requested together with the controller's words, one cold
    //  round trip instead of two in a row)
    const int sub = threadIdx.x & (kPointGroup - 1);
    const int l = bid * kPointsPerBlock + (threadIdx.x / kPointGroup);
    const bool valid = l < w.P && bid < w.n_pt_blocks;
    int begin = 0, end = 0;
    if (valid) { begin = w.pt_start[l]; end = w.pt_start[l + 1]; }
    if (c->done) return;
    const int cur = c->cur;
    if (bid >= w.n_pt_blocks) {         // the pass's extra workgroup: its first wave takes the LM decision (back-substitution passes only)
#ifdef MOVBA_CLOCK_STAMP
        const int ns0 = c->n_solves;
#endif
        if (BACKSUB && bid == w.n_pt_blocks && threadIdx.x < 64) decide_body(w, cur);
#ifdef MOVBA_CLOCK_STAMP
        if (BACKSUB && threadIdx.x == 0 && ns0 == 3 && 20000 + 8 * ((size_t)bid + 1) <= (size_t)w.E) {      // (stamps live in the chi2 array: windows large enough only)
            unsigned long long *dbg = reinterpret_cast<unsigned long long *>(w.out_chi2) + 20000 + 8 * (size_t)bid;
            dbg[0] = pst[0]; dbg[1] = __builtin_amdgcn_s_memrealtime(); dbg[7] = 2;
        }
#endif
        return;
    }
    PSTAMP(1);
    const int dst = BACKSUB ? (cur ^ 1) : cur;
    const double lambda = c->lambda;
    const DevState &S0 = w.st[cur];
    const DevState &S1 = w.st[dst];

    double *sRt = sm;                               // NP x 12 at dst
    double *sR0 = sm + (LDSP ? 12 * w.NP : 0);      // NP x 12 at cur  (BACKSUB)
    double *sxp = sR0 + ((BACKSUB && LDSP) ? 12 * w.NP : 0);  // nfree x 6       (BACKSUB)
    double *red = sxp + ((BACKSUB && LDSP) ? 6 * w.nfree : 0);// kPointRed
    int *shidx = reinterpret_cast<int *>(red + kPointRed);  // NP              (BACKSUB)
    // where the pose data is read from: the LDS images, or the state buffers themselves
    const double *pRt = LDSP ? sRt : S1.Rt;
    const double *pR0 = LDSP ? sR0 : S0.Rt;
    const double *pxp = LDSP ? sxp : w.xp;
    const int *phidx = LDSP ? shidx : w.hidx;

    // ---- everything this lane needs from HBM is requested before the LDS staging barrier, so that the
    //      point / edge gathers and the pose staging overlap instead of queueing behind each other ----
    double X[3] = { 0, 0, 0 };
    double Hc[6] = { 1, 0, 0, 1, 0, 1 }, bc[3] = { 0, 0, 0 };
    if (valid) {
        X[0] = S0.point[3 * l]; X[1] = S0.point[3 * l + 1]; X[2] = S0.point[3 * l + 2];
        if (BACKSUB) {
            const double2 *hp = reinterpret_cast<const double2 *>(S0.Hll + 6 * (size_t)l);      // 48-byte records: 16-byte aligned
            const double2 q0 = hp[0], q1 = hp[1], q2 = hp[2];
            Hc[0] = q0.x; Hc[1] = q0.y; Hc[2] = q1.x; Hc[3] = q1.y; Hc[4] = q2.x; Hc[5] = q2.y;
            bc[0] = S0.bl[3 * l]; bc[1] = S0.bl[3 * l + 1]; bc[2] = S0.bl[3 * l + 2];
        }
    }
    // the lane's first two edges (a point with more than 16 observations takes the loop further down)
    constexpr int kPre = 2;
    int pg[kPre], pip[kPre], psl[kPre];
    double2 pob[kPre];
    double pom[kPre], pur[kPre];
#pragma unroll
    for (int k = 0; k < kPre; ++k) {
        pg[k] = begin + sub + kPointGroup * k;
        const int g = min(pg[k], max(end - 1, 0));
        const bool in = pg[k] < end;
        pip[k] = in ? w.g_pose[g] : 0;
        psl[k] = in ? w.slot[g] : -1;
        pob[k] = in ? *reinterpret_cast<const double2 *>(w.obs + 2 * g) : make_double2(0, 0);
        pom[k] = in ? w.isig[g] : 0.0;
        pur[k] = (STEREO && in) ? w.obs_r[g] : -1.0;
    }

    // Staging of the keyframes' data, the usual window (up to 85 keyframes): every thread requests its pieces of ALL the
    // images - 16 bytes each, two per image - before it stores the first one.  (Loop after loop, as below for larger windows,
    // the passes were nine dependent L2 round trips per thread: 3 us of a 12 us pass at cfg3.)
    const bool stage_at_once = LDSP && BACKSUB && 12 * w.NP <= 4 * kPointBlock && 6 * w.nfree <= 4 * kPointBlock;
    if (stage_at_once) {
        const int n2 = 6 * w.NP, x2 = 3 * w.nfree;            // 16-byte pieces of a rotation image / of the increments
        const double2 *g0 = reinterpret_cast<const double2 *>(S0.Rt), *g1 = reinterpret_cast<const double2 *>(S1.Rt), *gx = reinterpret_cast<const double2 *>(w.xp);
        double2 a[2], b[2], x[2];
        int hh = -1;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = threadIdx.x + kPointBlock * u;
            a[u] = k < n2 ? g0[k] : make_double2(0.0, 0.0);
            b[u] = k < n2 ? g1[k] : make_double2(0.0, 0.0);
            x[u] = k < x2 ? gx[k] : make_double2(0.0, 0.0);
        }
        if ((int)threadIdx.x < w.NP) hh = w.hidx[threadIdx.x];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = threadIdx.x + kPointBlock * u;
            if (k < n2) { reinterpret_cast<double2 *>(sR0)[k] = a[u]; reinterpret_cast<double2 *>(sRt)[k] = b[u]; }
            if (k < x2) reinterpret_cast<double2 *>(sxp)[k] = x[u];
        }
        if ((int)threadIdx.x < w.NP) shidx[threadIdx.x] = hh;
        __syncthreads();
    } else if (BACKSUB) {
        if (LDSP) {
            for (int k = threadIdx.x; k < 12 * w.NP; k += kPointBlock) sR0[k] = S0.Rt[k];
            for (int k = threadIdx.x; k < w.NP; k += kPointBlock) shidx[k] = w.hidx[k];
        }
        if (LDSP) {
            for (int k = threadIdx.x; k < 12 * w.NP; k += kPointBlock) sRt[k] = S1.Rt[k];
            for (int k = threadIdx.x; k < 6 * w.nfree; k += kPointBlock) sxp[k] = w.xp[k];
            __syncthreads();
        }
    } else if (LDSP) {
        if (12 * w.NP <= 4 * kPointBlock) {                   // (both pieces of the image in flight at once, as above)
            const int n2 = 6 * w.NP;
            const double2 *g1 = reinterpret_cast<const double2 *>(S1.Rt);
            double2 b[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) { const int k = threadIdx.x + kPointBlock * u; b[u] = k < n2 ? g1[k] : make_double2(0.0, 0.0); }
#pragma unroll
            for (int u = 0; u < 2; ++u) { const int k = threadIdx.x + kPointBlock * u; if (k < n2) reinterpret_cast<double2 *>(sRt)[k] = b[u]; }
        } else {
            for (int k = threadIdx.x; k < 12 * w.NP; k += kPointBlock) sRt[k] = S1.Rt[k];
        }
        __syncthreads();
    }

    PSTAMP(2);
    double scale = 0.0;
    if (BACKSUB) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
        // x_l's right-hand side: sum over the free observers of B_il^T xp_i, rebuilt from the cached (Xc, w)
        // (the camera-frame point and the robust weight of the edge at the CURRENT state are recomputed from the staged pose,
        //  the point and the observation — the same expressions that produced Hll / bl — instead of being read back from a
        //  per-edge record: this kernel is bound by memory, not arithmetic)
        const double dsq0 = w.huber_delta * w.huber_delta;
        auto back_edge = [&](int g, int ip, const double2 &ob, double om, double ur) {
            const int h = phidx[ip];
            if (h < 0) return;
            const double *R = pR0 + 12 * ip;
            const double x = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + R[9];
            const double y = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + R[10];
            const double z = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + R[11];
            // (one reciprocal per edge and multiplications: an fp64 division is a ~25-instruction sequence, and the projection
            //  and its Jacobian would take six of them)
            const Cam cm = cam_of<PERKF>(w, ip);
            const double iz = fast_rcp_zero_safe(z), u = cm.fx * x * iz, v = cm.fy * y * iz;
            const double e0 = ob.x - (u + cm.cx);
            const double e1 = ob.y - (v + cm.cy);
            double chi2 = e0 * (om * e0) + e1 * (om * e1);
            if (STEREO && ur >= 0.0) { const double e2 = ur - (u + cm.cx - cm.bf * iz); chi2 += e2 * (om * e2); }
            double rho1 = 1.0;
            if (w.huber_delta > 0.0 && !(chi2 <= dsq0)) rho1 = w.huber_delta * rsqrt(chi2);
            const double wg = rho1 * om;
            const double a00 = -(cm.fx * iz), a02 = u * iz;
            const double a11 = -(cm.fy * iz), a12 = v * iz;
            const double *xp = pxp + 6 * h;
            // t = J_c xp  (rows of -Jpi [ -[Xc]x | I ])
            const double t0 = (a02 * y) * xp[0] + (a00 * z - a02 * x) * xp[1] + (-a00 * y) * xp[2] + a00 * xp[3] + a02 * xp[5];
            const double t1 = (-a11 * z + a12 * y) * xp[0] + (-a12 * x) * xp[1] + (a11 * x) * xp[2] + a11 * xp[4] + a12 * xp[5];
            const double g0 = wg * t0, g1 = wg * t1;
            // J_p = -Jpi R : rows p0 = a00 R0 + a02 R2, p1 = a11 R1 + a12 R2
            a0 += (a00 * R[0] + a02 * R[6]) * g0 + (a11 * R[3] + a12 * R[6]) * g1;
            a1 += (a00 * R[1] + a02 * R[7]) * g0 + (a11 * R[4] + a12 * R[7]) * g1;
            a2 += (a00 * R[2] + a02 * R[8]) * g0 + (a11 * R[5] + a12 * R[8]) * g1;
            if (STEREO && ur >= 0.0) {
                // stereo row (g2o::EdgeStereoSE3ProjectXYZ): like row 0 with a02 -> a02 - bf/z^2
                const double c02 = a02 - cm.bf * iz * iz;
                const double t2 = (c02 * y) * xp[0] + (a00 * z - c02 * x) * xp[1] + (-a00 * y) * xp[2] + a00 * xp[3] + c02 * xp[5];
                const double g2 = wg * t2;
                a0 += (a00 * R[0] + c02 * R[6]) * g2; a1 += (a00 * R[1] + c02 * R[7]) * g2; a2 += (a00 * R[2] + c02 * R[8]) * g2;
            }
            (void)g;
        };
#pragma unroll
        for (int k = 0; k < kPre; ++k)
            if (pg[k] < end) back_edge(pg[k], pip[k], pob[k], pom[k], pur[k]);
        for (int g = begin + sub + kPointGroup * kPre; g < end; g += kPointGroup)
            back_edge(g, w.g_pose[g], *reinterpret_cast<const double2 *>(w.obs + 2 * g), w.isig[g], STEREO ? w.obs_r[g] : -1.0);
        // (DPP moves inside the group of 8 lanes: no LDS crossbar, unlike the ds_bpermute behind __shfl_xor)
        a0 = group_sum_dpp(a0, kPointGroup); a1 = group_sum_dpp(a1, kPointGroup); a2 = group_sum_dpp(a2, kPointGroup);
        if (valid && end > begin) {
            double H[6], D[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) H[k] = Hc[k];
            H[0] += lambda; H[3] += lambda; H[5] += lambda;
            inv3sym(H, D);
            const double b0 = bc[0], b1 = bc[1], b2 = bc[2];
            const double c0 = b0 - a0, c1 = b1 - a1, c2 = b2 - a2;
            const double x0 = D[0] * c0 + D[1] * c1 + D[2] * c2;
            const double x1 = D[1] * c0 + D[3] * c1 + D[4] * c2;
            const double x2 = D[2] * c0 + D[4] * c1 + D[5] * c2;
            // A reduced solve that failed (Ctrl::pcg_fail: a Cholesky factorisation that met a non-positive pivot, NaN in the
            // normal equations) moves NOTHING: g2o's solver returns false before any vertex is updated
            // (OptimizationAlgorithmLevenberg::solve: `ok2 = _solver->solve(); if (ok2) update`), so the trial state is the current
            // state, evaluated once more below - which is what the next iteration's fresh cost is taken from when the decision
            // counts such a trial as accepted (decide_body).
            if (!c->pcg_fail) {
                X[0] += x0; X[1] += x1; X[2] += x2;
                if (sub == 0) scale = x0 * (lambda * x0 + b0) + x1 * (lambda * x1 + b1) + x2 * (lambda * x2 + b2);
            }
        }
    }

    PSTAMP(3);
    // ---- evaluate at the destination state ----
    double h0 = 0, h1 = 0, h2 = 0, h3 = 0, h4 = 0, h5 = 0, v0 = 0, v1 = 0, v2 = 0, F = 0.0;
    const double dsqr = w.huber_delta * w.huber_delta;
    auto eval_edge = [&](int g, int ip, int sl, const double2 &ob, double om, double ur) {
        const double *R = pRt + 12 * ip;
        const double x = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + R[9];
        const double y = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + R[10];
        const double z = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + R[11];
        const Cam cm = cam_of<PERKF>(w, ip);
        const double iz = fast_rcp_zero_safe(z), u = cm.fx * x * iz, v = cm.fy * y * iz;      // (one reciprocal per edge: see back_edge)
        const double e0 = ob.x - (u + cm.cx);
        const double e1 = ob.y - (v + cm.cy);
        double chi2 = e0 * (om * e0) + e1 * (om * e1);
        bool st = false;
        double e2 = 0.0;
        if (STEREO) {
            st = ur >= 0.0;
            if (st) { e2 = ur - (u + cm.cx - cm.bf * iz); chi2 += e2 * (om * e2); }
        }
        double rho0 = chi2, rho1 = 1.0;
        if (w.huber_delta > 0.0 && !(chi2 <= dsqr)) {
            // (sqrt(chi2) = chi2 / sqrt(chi2): for chi2 = inf - a point on the keyframe's z = 0 plane - the product is inf * 0;
            //  g2o's RobustKernelHuber takes sqrt(inf) = inf there, an infinite cost, not NaN)
            const double rs = rsqrt(chi2), sq = chi2 > DBL_MAX ? chi2 : chi2 * rs;
            rho0 = 2.0 * sq * w.huber_delta - dsqr;
            rho1 = w.huber_delta * rs;
        }
        const double wg = rho1 * om;
        const double r0 = -wg * e0, r1 = -wg * e1;
        if (sl >= 0) {      // pose-major record for the schur pass (edges of free keyframes): (Xc, w)
            *reinterpret_cast<double4 *>(S1.erecA + 4 * (size_t)sl) = make_double4(x, y, z, wg);
            if (!BACKSUB) {     // (once per run: the observation by slot, from which the schur pass rebuilds the residual)
                *reinterpret_cast<double2 *>(w.obs_pm + 2 * (size_t)sl) = ob;
                if (STEREO) w.obsr_pm[sl] = ur;
            }
        }
        F += rho0;
        const double a00 = -(cm.fx * iz), a02 = u * iz;
        const double a11 = -(cm.fy * iz), a12 = v * iz;
        const double p00 = a00 * R[0] + a02 * R[6], p01 = a00 * R[1] + a02 * R[7], p02 = a00 * R[2] + a02 * R[8];
        const double p10 = a11 * R[3] + a12 * R[6], p11 = a11 * R[4] + a12 * R[7], p12 = a11 * R[5] + a12 * R[8];
        h0 += wg * (p00 * p00 + p10 * p10); h1 += wg * (p00 * p01 + p10 * p11); h2 += wg * (p00 * p02 + p10 * p12);
        h3 += wg * (p01 * p01 + p11 * p11); h4 += wg * (p01 * p02 + p11 * p12); h5 += wg * (p02 * p02 + p12 * p12);
        v0 += p00 * r0 + p10 * r1; v1 += p01 * r0 + p11 * r1; v2 += p02 * r0 + p12 * r1;
        if (STEREO) {
            const double r2 = -wg * e2;
            if (st) {
                const double c02 = a02 - cm.bf * iz * iz;
                const double p20 = a00 * R[0] + c02 * R[6], p21 = a00 * R[1] + c02 * R[7], p22 = a00 * R[2] + c02 * R[8];
                h0 += wg * p20 * p20; h1 += wg * p20 * p21; h2 += wg * p20 * p22;
                h3 += wg * p21 * p21; h4 += wg * p21 * p22; h5 += wg * p22 * p22;
                v0 += p20 * r2; v1 += p21 * r2; v2 += p22 * r2;
            }
        }
    };
#pragma unroll
    for (int k = 0; k < kPre; ++k)
        if (pg[k] < end) eval_edge(pg[k], pip[k], psl[k], pob[k], pom[k], pur[k]);
    for (int g = begin + sub + kPointGroup * kPre; g < end; g += kPointGroup)
        eval_edge(g, w.g_pose[g], w.slot[g], *reinterpret_cast<const double2 *>(w.obs + 2 * g), w.isig[g], STEREO ? w.obs_r[g] : -1.0);
    h0 = group_sum_dpp(h0, kPointGroup); h1 = group_sum_dpp(h1, kPointGroup); h2 = group_sum_dpp(h2, kPointGroup);
    h3 = group_sum_dpp(h3, kPointGroup); h4 = group_sum_dpp(h4, kPointGroup); h5 = group_sum_dpp(h5, kPointGroup);
    v0 = group_sum_dpp(v0, kPointGroup); v1 = group_sum_dpp(v1, kPointGroup); v2 = group_sum_dpp(v2, kPointGroup);
    double hmax = 0.0;
    if (valid && sub == 0) {
        double *Hd = S1.Hll + 6 * l;
        Hd[0] = h0; Hd[1] = h1; Hd[2] = h2; Hd[3] = h3; Hd[4] = h4; Hd[5] = h5;
        S1.bl[3 * l] = v0; S1.bl[3 * l + 1] = v1; S1.bl[3 * l + 2] = v2;
        if (BACKSUB) { S1.point[3 * l] = X[0]; S1.point[3 * l + 1] = X[1]; S1.point[3 * l + 2] = X[2]; }
        if (end > begin) hmax = fmax(fabs(h0), fmax(fabs(h3), fabs(h5)));
    }
    PSTAMP(4);
    if (BACKSUB) {
        // cost and scale partials of the workgroup behind ONE barrier (the sums in block_reduce's order)
        constexpr int NWV = kPointBlock / 64;
        static_assert(2 * NWV <= kPointRed, "red holds kPointRed doubles (point_lds_bytes)");
        F = wave_sum(F); scale = wave_sum(scale);
        if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = F; red[NWV + (threadIdx.x >> 6)] = scale; }
        __syncthreads();
        // the LM decision needs every workgroup's partials: handed to the pass's deciding wave (decide_body) as two tagged
        // records, one 16-byte write-through store each, nothing to wait for
        if (threadIdx.x == 0) {
            double Fsum = red[0], ssum = red[NWV];
#pragma unroll
            for (int k = 1; k < NWV; ++k) { Fsum += red[k]; ssum += red[NWV + k]; }
            const unsigned tag = (unsigned)c->n_solves + 1u;
            const __amdgpu_buffer_rsrc_t rr = hx_rsrc(w.dec_rec, 32u * (unsigned)w.n_pt_blocks);
            hx_st_tagged(rr, 2u * (unsigned)bid, Fsum, tag); hx_st_tagged(rr, 2u * (unsigned)bid + 1u, ssum, tag);
#ifdef MOVBA_CLOCK_STAMP
            if (c->n_solves == 3 && 20000 + 8 * ((size_t)bid + 1) <= (size_t)w.E) {
                PSTAMP(5);
                unsigned long long *dbg = reinterpret_cast<unsigned long long *>(w.out_chi2) + 20000 + 8 * (size_t)bid;
                for (int k = 0; k < 6; ++k) dbg[k] = pst[k];
                dbg[6] = ((unsigned long long)__builtin_amdgcn_s_getreg(63492)) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
                dbg[7] = 1;
            }
#endif
        }
    } else {
        const double Fsum = block_reduce<kPointBlock / 64, false>(F, red);
        const double m = block_reduce<kPointBlock / 64, true>(hmax, red);
        if (threadIdx.x == 0) { w.hmax_part[bid] = m; S1.Fpart[bid] = Fsum; }
    }
}

template <bool BACKSUB, bool STEREO, bool LDSP>
__global__ __launch_bounds__(kPointBlock) void k_point(DevWindow w) { point_body<BACKSUB, STEREO, LDSP>(w, blockIdx.x); }
// intrinsics by keyframe (DevWindow::kcam): the variant that reads the keyframes' data through L2
template <bool BACKSUB, bool STEREO>
__global__ __launch_bounds__(kPointBlock) void k_point_kf(DevWindow w) { point_body<BACKSUB, STEREO, false, true>(w, blockIdx.x); }

// Batched launches (movba_lba_run_batch): one grid over the concatenated windows; `pre` is the prefix of the windows'
// block counts for this kernel.  Every window runs exactly the code of its solo launch, so results are bit-identical.
__device__ __forceinline__ int batch_window(const int32_t *pre, int n, int b)
{
    int lo = 0, hi = n - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (pre[mid] <= b) lo = mid; else hi = mid - 1; }
    return __builtin_amdgcn_readfirstlane(lo);
}

template <bool BACKSUB, bool STEREO, bool LDSP>
__global__ __launch_bounds__(kPointBlock) void k_point_b(BatchDev b)
{
    const int wi = batch_window(b.blk_point, b.n, blockIdx.x);
    point_body<BACKSUB, STEREO, LDSP>(b.wins[wi], blockIdx.x - b.blk_point[wi]);
}

__global__ void k_init_pose_b(BatchDev b)
{
    const int wi = batch_window(b.blk_init, b.n, blockIdx.x);
    init_pose_body(b.wins[wi], blockIdx.x - b.blk_init[wi], b.blk_init[wi + 1] - b.blk_init[wi]);
}

// partial-slot of each of the 54 sums of a diagonal work item (layout in device_types.h) and, for the
// 21 upper-triangle elements of the 6x6 block, the slot of the mirrored element (-1 on the diagonal)
__device__ __constant__ int8_t kDiagMap[54] = {0, 1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 14, 15, 16, 17, 21, 22, 23, 28, 29, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63, 64, 65, 66, 67, 68};
__device__ __constant__ int8_t kDiagMirror[21] = {-1, 6, 12, 18, 24, 30, -1, 13, 19, 25, 31, -1, 20, 26, 32, -1, 27, 33, -1, 34, -1};

// Fixed-order reduction of NV per-lane values over the 64 lanes of a wave: two DPP steps sum each quad,
// the 16 quad sums of every value go through a wave-private LDS strip ([NV][16] doubles) and lane k < NV
// adds them up in order (returned in lane k).  ~2k cycles for NV = 54, against ~30k for NV butterfly
// reductions built on ds_bpermute shuffles.
template <int NV>
__device__ __forceinline__ double wave_reduce(double (&v)[NV], double *strip, int lane)
{
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
    double s = 0.0;
    if (lane < NV) {
        const double2 *src = reinterpret_cast<const double2 *>(strip + lane * 16);
#pragma unroll
        for (int q = 0; q < 8; ++q) { const double2 t = src[q]; s += t.x; s += t.y; }
    }
    return s;                                   // lane k < NV holds the wave's sum of value k
}

// --------------------------------------------------------------------------------
// k_schur: one workgroup (kSchurWaves waves) per work item (a chunk of up to 2048 of one pose pair's shared points).
// Per entry (edge of pose i, edge of pose j, both on point l) a lane rebuilds the
// Jacobians from the cached camera-frame point and adds
//     B_il Dinv_l B_jl^T = w_i w_j  Jc_i^T ( Jp_i Dinv_l Jp_j^T ) Jc_j          (6x6)
// (BlockSolver<6,3>::solve, Schur step, without ever storing the 6x3 Hpl blocks);
// diagonal pairs also accumulate Hpp_ii, b_p,i (BaseBinaryEdge::constructQuadraticForm)
// and B_il Dinv_l b_l.  Wave-level shuffle reduction, one 72-double partial per item.
// mode 1 = diagonal pairs only, Hpp only (used once to seed lambda).
// --------------------------------------------------------------------------------
// Jacobian rows of one edge from its cached camera-frame point: P = J_point rows (-Jpi R), C = J_pose rows
// (-Jpi [ -[Xc]x | I ]).  NR = 2: monocular edge (src/OptimizableTypes.cpp:158-180); NR = 3 adds the stereo row of
// g2o::EdgeStereoSE3ProjectXYZ (built at src/Optimizer.cc:673-705), zeroed for the monocular edges of a mixed window.
// (A division-free variant on normalised records (x / z, y / z, 1 / z) with the structural zeros of C skipped was 20 %
//  SLOWER, solo and batched: the kernel then needs all 256 registers and the compiler schedules its gathers worse.)
template <int NR>
__device__ __forceinline__ void edge_rows(const Cam &cm, double x, double y, double z, const double R[9], bool stereo,
                                          double (&P)[NR][3], double (&C)[NR][6])
{
    const double iz = fast_rcp(z);
    const double a00 = -cm.fx * iz, a02 = cm.fx * x * iz * iz, a11 = -cm.fy * iz, a12 = cm.fy * y * iz * iz;
#pragma unroll
    for (int q = 0; q < 3; ++q) { P[0][q] = a00 * R[q] + a02 * R[6 + q]; P[1][q] = a11 * R[3 + q] + a12 * R[6 + q]; }
    C[0][0] = a02 * y; C[0][1] = a00 * z - a02 * x; C[0][2] = -a00 * y; C[0][3] = a00; C[0][4] = 0.0; C[0][5] = a02;
    C[1][0] = -a11 * z + a12 * y; C[1][1] = -a12 * x; C[1][2] = a11 * x; C[1][3] = 0.0; C[1][4] = a11; C[1][5] = a12;
    if (NR == 3) {
        const double m = stereo ? 1.0 : 0.0;
        const double c00 = m * a00, c02 = m * (a02 - cm.bf * iz * iz);
#pragma unroll
        for (int q = 0; q < 3; ++q) P[NR - 1][q] = c00 * R[q] + c02 * R[6 + q];
        C[NR - 1][0] = c02 * y; C[NR - 1][1] = c00 * z - c02 * x; C[NR - 1][2] = -c00 * y;
        C[NR - 1][3] = c00; C[NR - 1][4] = 0.0; C[NR - 1][5] = c02;
    }
}

// accumulates one diagonal-pair entry: ha += w C^T C (Hpp), ba += C^T (-w e) (b_p) and, unless HPP_ONLY, the Schur
// terms sa += B Dinv B^T, ca += B Dinv b_l.  `wg` = 0 masks the entry out (all its contributions are exact zeros).
template <int NR, bool HPP_ONLY>
__device__ __forceinline__ void schur_diag_entry(const Cam &cm, const double4 &rc, double wg, const double (&rv)[NR], bool st,
                                                 const double (&Ri)[9], const double2 (&h)[3], const double (&bl)[3], double lambda,
                                                 double (&sa)[21], double (&ha)[21], double (&ca)[6], double (&ba)[6])
{
    double P[NR][3], C[NR][6];
    edge_rows<NR>(cm, rc.x, rc.y, rc.z, Ri, st, P, C);
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int m = 0; m < NR; ++m) ba[a] += C[m][a] * rv[m];
#pragma unroll
        for (int b = a; b < 6; ++b) {
            double t = 0.0;
#pragma unroll
            for (int m = 0; m < NR; ++m) t += C[m][a] * C[m][b];
            ha[ut6(a, b)] += wg * t;
        }
    }
    if (HPP_ONLY) return;
    double H[6], D[6];
    H[0] = h[0].x + lambda; H[1] = h[0].y; H[2] = h[1].x; H[3] = h[1].y + lambda; H[4] = h[2].x; H[5] = h[2].y + lambda;
    inv3sym(H, D);
    double T[NR][3], M[NR][NR], pv[NR];
    const double w2 = wg * wg;
#pragma unroll
    for (int m = 0; m < NR; ++m) {
        T[m][0] = P[m][0] * D[0] + P[m][1] * D[1] + P[m][2] * D[2];
        T[m][1] = P[m][0] * D[1] + P[m][1] * D[3] + P[m][2] * D[4];
        T[m][2] = P[m][0] * D[2] + P[m][1] * D[4] + P[m][2] * D[5];
        pv[m] = wg * (T[m][0] * bl[0] + T[m][1] * bl[1] + T[m][2] * bl[2]);
    }
#pragma unroll
    for (int m = 0; m < NR; ++m)
#pragma unroll
        for (int q = 0; q < NR; ++q) M[m][q] = w2 * (T[m][0] * P[q][0] + T[m][1] * P[q][1] + T[m][2] * P[q][2]);
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        double u[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            u[q] = 0.0;
#pragma unroll
            for (int m = 0; m < NR; ++m) u[q] += C[m][a] * M[m][q];
        }
#pragma unroll
        for (int m = 0; m < NR; ++m) ca[a] += C[m][a] * pv[m];
#pragma unroll
        for (int b = a; b < 6; ++b) {
#pragma unroll
            for (int q = 0; q < NR; ++q) sa[ut6(a, b)] += u[q] * C[q][b];
        }
    }
}

// accumulates one off-diagonal entry: acc += w_i w_j Jc_i^T (Jp_i Dinv Jp_j^T) Jc_j; `ww` = 0 masks the entry out
template <int NR>
__device__ __forceinline__ void schur_offdiag_entry(const Cam &cmi, const Cam &cmj, const double4 &ri, const double4 &rj, double ww, bool sti, bool stj,
                                                    const double (&Ri)[9], const double (&Rj)[9], const double2 (&h)[3], double lambda,
                                                    double (&acc)[36])
{
    double H[6], D[6];
    H[0] = h[0].x + lambda; H[1] = h[0].y; H[2] = h[1].x; H[3] = h[1].y + lambda; H[4] = h[2].x; H[5] = h[2].y + lambda;
    inv3sym(H, D);
    double P[NR][3], C[NR][6], Q[NR][3], Ec[NR][6];
    edge_rows<NR>(cmi, ri.x, ri.y, ri.z, Ri, sti, P, C);
    edge_rows<NR>(cmj, rj.x, rj.y, rj.z, Rj, stj, Q, Ec);
    double M[NR][NR];
#pragma unroll
    for (int m = 0; m < NR; ++m) {
        const double t0 = P[m][0] * D[0] + P[m][1] * D[1] + P[m][2] * D[2];
        const double t1 = P[m][0] * D[1] + P[m][1] * D[3] + P[m][2] * D[4];
        const double t2 = P[m][0] * D[2] + P[m][1] * D[4] + P[m][2] * D[5];
#pragma unroll
        for (int q = 0; q < NR; ++q) M[m][q] = ww * (t0 * Q[q][0] + t1 * Q[q][1] + t2 * Q[q][2]);
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        double u[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            u[q] = 0.0;
#pragma unroll
            for (int m = 0; m < NR; ++m) u[q] += C[m][a] * M[m][q];
        }
#pragma unroll
        for (int b = 0; b < 6; ++b) {
#pragma unroll
            for (int q = 0; q < NR; ++q) acc[a * 6 + b] += u[q] * Ec[q][b];
        }
    }
}

// (one entry per lane in flight, diagonal and off-diagonal items alike: 196 registers per lane instead of 232 and, with the
//  small items sharing workgroups - structure.cpp - a shorter pass than two in flight: 14.0 against 15.1 us at cfg3, batched
//  2.09 against 2.17 ms per eight windows; round 4, builds with -DMOVBA_SCHUR_BD / _BO)
#ifndef MOVBA_SCHUR_BD
#define MOVBA_SCHUR_BD 1
#endif
#ifndef MOVBA_SCHUR_BO
#define MOVBA_SCHUR_BO 1
#endif
constexpr int kSchurBatchDiag = MOVBA_SCHUR_BD;     // entries per lane whose gathers are in flight together (diagonal items)
constexpr int kSchurBatchOff = MOVBA_SCHUR_BO;      // same, off-diagonal items

// HPP_ONLY: diagonal pairs only, Hpp and b_p only (one launch per solve, seeds lambda)
template <int NR, bool HPP_ONLY, bool PERKF = false>
__device__ __forceinline__ void schur_body(const DevWindow &w, int bid)
{
#ifdef MOVBA_CLOCK_STAMP
    unsigned long long wst[5];
    wst[0] = __builtin_amdgcn_s_memrealtime();
#define WSTAMP(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); wst[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define WSTAMP(k) do { } while (0)
#endif
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);       // wave-uniform: item, poses and rotations live in SGPRs
    // XCD-aware launch schedule (structure.cpp): workgroups b, b+8, ... share an XCD (and its L2) and take the slots of
    // that XCD's segment in order.  The slot is fetched together with the LM state (one scalar round trip).
    static_assert(kSchurWaves == 4, "the schedule deals wave slots in workgroups of four (structure.cpp)");
    const int wg = (bid & 7) * (w.sched_per_xcd / kSchurWaves) + (bid >> 3);
    const SchedItem it = w.sched[wg * kSchurWaves + wv];    // this wave's slot: its share of an item one, two or four waves take
    const Ctrl *c = w.ctrl;
    if (c->done) return;
    const int sub = it.sub & 0xff, nsub = it.sub >> 8;
    __shared__ __attribute__((aligned(16))) double strips[kSchurWaves][54 * 16];
    __shared__ double wsum[kSchurWaves][64];
    double *strip = strips[wv];
    bool active = it.tag >= 0;
    const int item = active ? (it.tag >> 1) : 0;
    const bool is_diag = active && (it.tag & 1);
    if (HPP_ONLY && !is_diag) active = false;
    const int wbeg = it.begin, wend = active ? it.end : it.begin;
    const int cur = c->cur;
    const double lambda = c->lambda;
    const DevState &S0 = w.st[cur];
    const int ip = it.pose_i, jp = it.pose_j;
    double Ri[9], Rj[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) { Ri[k] = S0.Rt[12 * ip + k]; Rj[k] = S0.Rt[12 * jp + k]; }
    const Cam cmi = cam_of<PERKF>(w, ip), cmj = cam_of<PERKF>(w, jp);
    double *out = w.part + (size_t)item * kPartStride;
    WSTAMP(1);

    // Every batch issues the gathers of B entries per lane level by level (entry -> edge records -> point block) before
    // any arithmetic: the loop is a chain of dependent L2 / fabric round trips, not bandwidth.
    if (is_diag) {
        constexpr int B = kSchurBatchDiag;
        double sa[21], ha[21], ca[6], ba[6];
#pragma unroll
        for (int k = 0; k < 21; ++k) { sa[k] = 0.0; ha[k] = 0.0; }
#pragma unroll
        for (int k = 0; k < 6; ++k) { ca[k] = 0.0; ba[k] = 0.0; }
        for (int base = wbeg; base < wend; base += 64 * B) {
            Int4 en[B]; bool ok[B];
            // the diagonal pair of keyframe i lists ALL its edges in map-point order and the diagonal pairs come first: entry k
            // IS pose-major slot k (no entry list to read: the record loads do not wait for one), its map point comes from
            // slot_point; the record loads of a wave are contiguous
#pragma unroll
            for (int u = 0; u < B; ++u) { const int kk = base + lane + 64 * u; ok[u] = kk < wend; const int k = min(kk, wend - 1); en[u] = Int4{ k, k, w.slot_point[k], 0 }; }
            double4 rc[B]; double2 ob[B]; double ur[B]; double2 h[B][3]; double bl[B][3];
#pragma unroll
            for (int u = 0; u < B; ++u) {
                rc[u] = *reinterpret_cast<const double4 *>(S0.erecA + 4 * (size_t)en[u].x);
                ob[u] = *reinterpret_cast<const double2 *>(w.obs_pm + 2 * (size_t)en[u].x);
                ur[u] = NR == 3 ? w.obsr_pm[en[u].x] : -1.0;
                if (HPP_ONLY) { h[u][0] = h[u][1] = h[u][2] = make_double2(0.0, 0.0); bl[u][0] = bl[u][1] = bl[u][2] = 0.0; continue; }
                const int l = en[u].z;
                const double2 *hp = reinterpret_cast<const double2 *>(S0.Hll + 6 * l);      // 48-byte records: 16-byte aligned
                h[u][0] = hp[0]; h[u][1] = hp[1]; h[u][2] = hp[2];
                bl[u][0] = S0.bl[3 * l]; bl[u][1] = S0.bl[3 * l + 1]; bl[u][2] = S0.bl[3 * l + 2];
            }
#pragma unroll
            for (int u = 0; u < B; ++u) {
                const double m = ok[u] ? 1.0 : 0.0;
                const bool st = NR == 3 && ur[u] >= 0.0;
                // -w e, e = observation - projection of the recorded camera-frame point (include/OptimizableTypes.h: computeError)
                const double iz = fast_rcp(rc[u].z), mw = -(m * rc[u].w);
                double rv[NR];
                rv[0] = mw * (ob[u].x - (cmi.fx * rc[u].x * iz + cmi.cx)); rv[1] = mw * (ob[u].y - (cmi.fy * rc[u].y * iz + cmi.cy));
                if (NR == 3) rv[NR - 1] = st ? mw * (ur[u] - (cmi.fx * rc[u].x * iz + cmi.cx - cmi.bf * iz)) : 0.0;
                schur_diag_entry<NR, HPP_ONLY>(cmi, rc[u], m * rc[u].w, rv, st, Ri, h[u], bl[u], lambda, sa, ha, ca, ba);
            }
        }
        // 54 sums: [0,21) upper triangle of sum B Dinv B^T, [21,27) B Dinv b_l, [27,48) upper Hpp, [48,54) b_p
        double all[54];
#pragma unroll
        for (int k = 0; k < 21; ++k) { all[k] = sa[k]; all[27 + k] = ha[k]; }
#pragma unroll
        for (int k = 0; k < 6; ++k) { all[21 + k] = ca[k]; all[48 + k] = ba[k]; }
        WSTAMP(2);
        wsum[wv][lane] = wave_reduce<54>(all, strip, lane);
    } else {
        constexpr int B = kSchurBatchOff;
        double acc[36];
#pragma unroll
        for (int k = 0; k < 36; ++k) acc[k] = 0.0;
        for (int base = wbeg; base < wend; base += 64 * B) {
            Int4 en[B]; bool ok[B];
#pragma unroll
            for (int u = 0; u < B; ++u) {
                const int kk = base + lane + 64 * u; ok[u] = kk < wend;
                const int k = min(kk, wend - 1) - w.n_diag;
                if (w.ent64) { const unsigned long long e = w.ent64[k]; en[u] = Int4{ (int)(e & 0x3fffffu), (int)((e >> 22) & 0x3fffffu), (int)(e >> 44), 0 }; }
                else en[u] = Int4{ w.ent_i[k], w.ent_j[k], w.ent_l[k], 0 };
            }
            // edges of keyframe i (and of j) shared with the other one, in map-point order: ascending pose-major records
            double4 ri[B], rj[B]; double2 h[B][3]; bool sti[B], stj[B];
#pragma unroll
            for (int u = 0; u < B; ++u) {
                ri[u] = *reinterpret_cast<const double4 *>(S0.erecA + 4 * (size_t)en[u].x);
                rj[u] = *reinterpret_cast<const double4 *>(S0.erecA + 4 * (size_t)en[u].y);
                sti[u] = false; stj[u] = false;
                if (NR == 3) { sti[u] = w.obsr_pm[en[u].x] >= 0.0; stj[u] = w.obsr_pm[en[u].y] >= 0.0; }
                const double2 *hp = reinterpret_cast<const double2 *>(S0.Hll + 6 * en[u].z);
                h[u][0] = hp[0]; h[u][1] = hp[1]; h[u][2] = hp[2];
            }
#pragma unroll
            for (int u = 0; u < B; ++u)
                schur_offdiag_entry<NR>(cmi, cmj, ri[u], rj[u], ok[u] ? ri[u].w * rj[u].w : 0.0, sti[u], stj[u], Ri, Rj, h[u], lambda, acc);
        }
        WSTAMP(2);
        wsum[wv][lane] = wave_reduce<36>(acc, strip, lane);
    }
    WSTAMP(3);
    __syncthreads();
    if (active && sub == 0) {
        double t = wsum[wv][lane];
        for (int q = 1; q < nsub; ++q) t += wsum[wv + q][lane];
        if (is_diag) {
            if (lane < 54) {
                *(out + kDiagMap[lane]) = t;
                // the lane holding upper element (a,b) also fills its mirror (b,a) of the 6x6 block
                if (lane < 21 && kDiagMirror[lane] >= 0) *(out + kDiagMirror[lane]) = t;
            }
            // ... and once more as the on-chip PCG's setup reads it (DevWindow::rec_d): per row a of the keyframe's block the six
            // values of Hpp - sum B Dinv B^T, then (sum B Dinv b_l)_a and (b_p)_a.  The 54 sums sit one per lane ([0,21) upper
            // B Dinv B^T, [21,27) B Dinv b_l, [27,48) upper Hpp, [48,54) b_p): exchanged through the wave's LDS strip.
            wsum[wv][lane] = t;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < 48) {
                double *rec = w.rec_d + (size_t)it.dst_a * 48;        // (diagonal items: dst_a = keyframe * rec_slots + the item's place in its pair)
                if (lane < 36) {
                    const int a = lane / 6, q = lane - a * 6, u = a <= q ? ut6(a, q) : ut6(q, a);
                    *(rec + a * 8 + q) = wsum[wv][27 + u] - wsum[wv][u];
                } else if (lane < 42) *(rec + (lane - 36) * 8 + 6) = wsum[wv][21 + (lane - 36)];
                else *(rec + (lane - 42) * 8 + 7) = wsum[wv][48 + (lane - 42)];
            }
        } else if (lane < 36) {
            *(out + lane) = t;
            // ... and where the lanes of the on-chip PCG that hold the block read it (DevWindow::img_b), as stored and transposed
            if (it.dst_a >= 0) *(w.img_b + it.dst_a + (size_t)lane * kPcgRowsThreads) = t;
            if (it.dst_b >= 0) *(w.img_b + it.dst_b + (size_t)((lane % 6) * 6 + lane / 6) * kPcgRowsThreads) = t;
        }
    }
#ifdef MOVBA_CLOCK_STAMP
    WSTAMP(4);
    if (!HPP_ONLY && lane == 0 && c->n_solves == 3 && 8 * ((size_t)(bid * kSchurWaves + wv) + 1) <= (size_t)w.E) {        // one launch: per-wave stamps (100 MHz ticks), read back through out_chi2
        unsigned long long *dbg = reinterpret_cast<unsigned long long *>(w.out_chi2) + 8 * (size_t)(bid * kSchurWaves + wv);
        for (int k = 0; k < 5; ++k) dbg[k] = wst[k];
        dbg[5] = (unsigned long long)(wend - wbeg);
        dbg[6] = (unsigned long long)is_diag | ((unsigned long long)__builtin_amdgcn_s_getreg(63492) << 8) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 40);
        dbg[7] = 1;
    }
#endif
}

#ifndef MOVBA_SCHUR_MINWAVES
#define MOVBA_SCHUR_MINWAVES 1
#endif
template <int NR, bool HPP_ONLY>
__global__ __launch_bounds__(kSchurWaves * 64, MOVBA_SCHUR_MINWAVES) void k_schur(DevWindow w) { schur_body<NR, HPP_ONLY>(w, blockIdx.x); }

// (a window with intrinsics by keyframe: solved on its own, never in a batch - api.cpp)
template <int NR, bool HPP_ONLY>
__global__ __launch_bounds__(kSchurWaves * 64) void k_schur_kf(DevWindow w) { schur_body<NR, HPP_ONLY, true>(w, blockIdx.x); }

template <int NR, bool HPP_ONLY>
__global__ __launch_bounds__(kSchurWaves * 64) void k_schur_b(BatchDev b)
{
    const int wi = batch_window(b.blk_schur, b.n, blockIdx.x);
    schur_body<NR, HPP_ONLY>(b.wins[wi], blockIdx.x - b.blk_schur[wi]);
}

// --------------------------------------------------------------------------------
// k_lambda_init: OptimizationAlgorithmLevenberg::computeLambdaInit — tau * max |H_jj| over
// the free pose and point diagonals (tau = 1e-5), and the initial robust cost F0.
// --------------------------------------------------------------------------------
__device__ __forceinline__ void lambda_init_body(const DevWindow &w)
{
    Ctrl *c = w.ctrl;
    const int lane = threadIdx.x;
    // which reduced solver the window starts on (no on-chip PCG: the direct solver from the first trial)
    if (lane == 0) { c->solver_mode = w.direct_only ? 1 : 0; c->direct_from = w.direct_only ? 0 : -1; }
    if (c->done) return;
    double m = 0.0, F = 0.0;
    // fixed-order cost sum: lane-strided partials (loads 8 deep), then a fixed butterfly
    for (int k0 = lane; k0 < w.n_pt_blocks; k0 += 64 * 8) {
        double fv[8], mv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = min(k0 + 64 * u, w.n_pt_blocks - 1);
            fv[u] = w.st[0].Fpart[k]; mv[u] = w.hmax_part[k];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool in = k0 + 64 * u < w.n_pt_blocks;
            F += in ? fv[u] : 0.0; m = fmax(m, in ? mv[u] : 0.0);
        }
    }
    // diagonal of Hpp: element (a, a) of keyframe i = the sum over the keyframe's diagonal work items, in item order, of what
    // the Hpp-only schur pass left in DevWindow::rec_d (row a, place a: Hpp - 0).  One (keyframe, a) per lane and round, five
    // rounds and four items each in flight at once (until round 4: a lane per keyframe walking items and elements one dependent
    // load at a time, ~7 us of the solve's setup).
    constexpr int kRounds = 5, kItemsFly = 4;
    for (int t0 = lane; t0 < 6 * w.nfree; t0 += 64 * kRounds) {
        int ni[kRounds];
        double v[kRounds][kItemsFly];
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {
            const int t = min(t0 + 64 * r, 6 * w.nfree - 1), i = t / 6, a = t - 6 * i;
            ni[r] = w.pair_item_start[i + 1] - w.pair_item_start[i];
            const double *rec = w.rec_d + (size_t)i * w.rec_slots * 48 + a * 9;
#pragma unroll
            for (int u = 0; u < kItemsFly; ++u) v[r][u] = rec[(size_t)min(u, w.rec_slots - 1) * 48];
        }
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {
            const int t = min(t0 + 64 * r, 6 * w.nfree - 1), i = t / 6, a = t - 6 * i;
            double s = 0.0;
#pragma unroll
            for (int u = 0; u < kItemsFly; ++u) s += u < ni[r] ? v[r][u] : 0.0;
            const double *rec = w.rec_d + (size_t)i * w.rec_slots * 48 + a * 9;
            for (int u = kItemsFly; u < ni[r]; ++u) s += rec[(size_t)u * 48];       // (keyframes with more than 8 192 edges)
            m = fmax(m, t0 + 64 * r < 6 * w.nfree ? fabs(s) : 0.0);
        }
    }
    m = wave_max(m);
    F = wave_sum(F);
    if (lane == 0) {
        c->lambda = 1e-5 * m;
        c->nu = 2.0;
        c->F0 = F;
        c->cost0 = F;
    }
}

__global__ __launch_bounds__(64) void k_lambda_init(DevWindow w) { lambda_init_body(w); }
__global__ __launch_bounds__(64) void k_lambda_init_b(BatchDev b) { lambda_init_body(b.wins[blockIdx.x]); }


// --------------------------------------------------------------------------------
// k_finalize: chi2 / outlier flags in caller edge order (src/Optimizer.cc:757-775).
// With the stale-error quirk the chi2 of a rejected last trial is reported, as g2o leaves
// it in the edges; the depth test always uses the final estimates.
// --------------------------------------------------------------------------------
__device__ __forceinline__ void finalize_body(const DevWindow &w, int bid, int nblk)
{
    const Ctrl *c = w.ctrl;
    const int g = bid * blockDim.x + threadIdx.x;
    int bad = 0;
    if (g < w.E) {
        const int cur = c->cur;
        const int sel = ((w.flags & MOVBA_FLAG_STALE_ERROR_QUIRK) && c->last_rejected) ? (cur ^ 1) : cur;
        // e->chi2() of the edge at state `sel`, recomputed here once instead of being stored by every trial's point pass
        // (the trial state of a rejected last trial is still in the other buffer: poses from the solver's epilogue, points
        // from the back-substitution)
        double chi2;
        {
            const double *Rs = w.st[sel].Rt + 12 * w.g_pose[g];
            const double *Xs = w.st[sel].point + 3 * w.g_point[g];
            const double x = Rs[0] * Xs[0] + Rs[1] * Xs[1] + Rs[2] * Xs[2] + Rs[9];
            const double y = Rs[3] * Xs[0] + Rs[4] * Xs[1] + Rs[5] * Xs[2] + Rs[10];
            const double z = Rs[6] * Xs[0] + Rs[7] * Xs[1] + Rs[8] * Xs[2] + Rs[11];
            const double om = w.isig[g];
            const Cam cm = cam_of_rt(w, w.g_pose[g]);
            const double e0 = w.obs[2 * g] - (cm.fx * x / z + cm.cx), e1 = w.obs[2 * g + 1] - (cm.fy * y / z + cm.cy);
            chi2 = e0 * (om * e0) + e1 * (om * e1);
            if (w.stereo) {
                const double ur = w.obs_r[g];
                if (ur >= 0.0) { const double e2 = ur - (cm.fx * x / z + cm.cx - cm.bf / z); chi2 += e2 * (om * e2); }
            }
        }
        // isDepthPositive() at the final estimates (include/OptimizableTypes.h:111-116)
        const double *Rz = w.st[cur].Rt + 12 * w.g_pose[g];
        const double *Xf = w.st[cur].point + 3 * w.g_point[g];
        const double zc = Rz[6] * Xf[0] + Rz[7] * Xf[1] + Rz[8] * Xf[2] + Rz[11];
        bad = (chi2 > w.chi2_gate) || !(zc > 0.0);
        const int e = w.perm ? w.perm[g] : g;          // null: the caller's edges were already grouped by map point
#ifdef MOVBA_CLOCK_STAMP
        if (e >= 32768)
#endif
        w.out_chi2[e] = chi2;
        w.out_outlier[e] = (uint8_t)bad;
    }
    // (the outlier count is taken on the host from the downloaded flags: one contended atomic per wave
    // made this kernel four times longer than its memory traffic)
    // the controller state goes to the host's pinned copy from here: a separate small device-to-host copy behind the last
    // kernel costs more than these few hundred stores across the bus
    if (bid == 0) {
        static_assert(sizeof(Ctrl) % 8 == 0, "copied as 64-bit words");
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(c);
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(w.ctrl_out);
        for (int k = threadIdx.x; k < (int)(sizeof(Ctrl) / 8); k += blockDim.x) dst[k] = src[k];
    }
    // final poses into the caller's registered device buffer (movba_lba_set_pose_export: what the all-gather sends)
    if (w.pose_export && bid == nblk - 1) {
        const double *src = w.st[c->cur].pose;
        for (int k = threadIdx.x; k < 7 * w.NP; k += blockDim.x) w.pose_export[k] = src[k];
    }
}

__global__ __launch_bounds__(256) void k_finalize(DevWindow w) { finalize_body(w, blockIdx.x, gridDim.x); }
__global__ __launch_bounds__(256) void k_finalize_b(BatchDev b)
{
    const int wi = batch_window(b.blk_final, b.n, blockIdx.x);
    finalize_body(b.wins[wi], blockIdx.x - b.blk_final[wi], b.blk_final[wi + 1] - b.blk_final[wi]);
}

// --------------------------------------------------------------------------------
// k_export: results -> the host's pinned staging buffer, written across the bus by the kernel itself.  Four separate
// device-to-host copies cost ~0.1 ms EACH when they are small (cfg2: 0.45 ms per download), one launch costs ~10 us.
// --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_export(DevWindow w, ExportDst d)
{
    const Ctrl *c = w.ctrl;
    const int cur = c->cur;
    const unsigned long long *pose = reinterpret_cast<const unsigned long long *>(w.st[cur].pose);
    const unsigned long long *point = reinterpret_cast<const unsigned long long *>(w.st[cur].point);
    const unsigned long long *chi2 = reinterpret_cast<const unsigned long long *>(w.out_chi2);
    const unsigned long long *outl = reinterpret_cast<const unsigned long long *>(w.out_outlier);     // (allocations are padded to 256 bytes)
    const long n0 = d.poses ? 7L * w.NP : 0, n1 = n0 + (d.points ? 3L * w.P : 0), n2 = n1 + (d.chi2 ? (long)w.E : 0), n3 = n2 + ((long)w.E + 7) / 8;
    for (long k = blockIdx.x * (long)blockDim.x + threadIdx.x; k < n3; k += (long)gridDim.x * blockDim.x) {
        if (k < n0) d.poses[k] = pose[k];
        else if (k < n1) d.points[k - n0] = point[k - n0];
        else if (k < n2) d.chi2[k - n1] = chi2[k - n1];
        else d.outlier[k - n2] = outl[k - n2];
    }
}

// --------------------------------------------------------------------------------
// launch wrappers
// --------------------------------------------------------------------------------
static inline size_t point_lds_bytes(const DevWindow &w, bool backsub)
{
    if (!w.lds_poses) return kPointRed * sizeof(double) + 16;
    size_t d = 12 * (size_t)w.NP + (backsub ? 12 * (size_t)w.NP + 6 * (size_t)w.nfree : 0) + kPointRed;
    return d * sizeof(double) + (backsub ? sizeof(int) * (size_t)w.NP : 0) + 16;
}

// the largest LDS image the point kernels would stage for this window (decides DevWindow::lds_poses)
size_t point_lds_need(int NP, int nfree)
{
    return (24 * (size_t)NP + 6 * (size_t)nfree + kPointRed) * sizeof(double) + sizeof(int) * (size_t)NP + 16;
}

hipError_t launch_init(const DevWindow &w, hipStream_t s)
{
    const int work = w.NP > (3 * w.P) / 2 ? w.NP : (3 * w.P) / 2;
    int nb = (work + 255) / 256;
    nb = nb < 1 ? 1 : (nb > 1024 ? 1024 : nb);
    hipLaunchKernelGGL(k_init_pose, dim3(nb), dim3(256), 0, s, w);
    return hipGetLastError();
}

hipError_t launch_linearize(const DevWindow &w, hipStream_t s)
{
    if (w.kcam) {
        if (w.stereo) hipLaunchKernelGGL((k_point_kf<false, true>), dim3(w.n_pt_blocks), dim3(kPointBlock), point_lds_bytes_for(w, false, false), s, w);
        else hipLaunchKernelGGL((k_point_kf<false, false>), dim3(w.n_pt_blocks), dim3(kPointBlock), point_lds_bytes_for(w, false, false), s, w);
    } else if (!w.lds_poses) {
        if (w.stereo) hipLaunchKernelGGL((k_point<false, true, false>), dim3(w.n_pt_blocks), dim3(kPointBlock), point_lds_bytes(w, false), s, w);
        else hipLaunchKernelGGL((k_point<false, false, false>), dim3(w.n_pt_blocks), dim3(kPointBlock), point_lds_bytes(w, false), s, w);
    } else if (w.stereo) hipLaunchKernelGGL((k_point<false, true, true>), dim3(w.n_pt_blocks), dim3(kPointBlock), point_lds_bytes(w, false), s, w);
    else hipLaunchKernelGGL((k_point<false, false, true>), dim3(w.n_pt_blocks), dim3(kPointBlock), point_lds_bytes(w, false), s, w);
    return hipGetLastError();
}

hipError_t launch_schur(const DevWindow &w, int mode, hipStream_t s)
{
    const int nblk = 8 * (w.sched_per_xcd / kSchurWaves);     // four wave slots per workgroup; multiple of 8: a contiguous run of slots per XCD
    if (w.kcam) {
        if (mode == 1) {
            if (w.stereo) hipLaunchKernelGGL((k_schur_kf<3, true>), dim3(nblk), dim3(kSchurWaves * 64), 0, s, w);
            else hipLaunchKernelGGL((k_schur_kf<2, true>), dim3(nblk), dim3(kSchurWaves * 64), 0, s, w);
        } else {
            if (w.stereo) hipLaunchKernelGGL((k_schur_kf<3, false>), dim3(nblk), dim3(kSchurWaves * 64), 0, s, w);
            else hipLaunchKernelGGL((k_schur_kf<2, false>), dim3(nblk), dim3(kSchurWaves * 64), 0, s, w);
        }
    } else if (mode == 1) {
        if (w.stereo) hipLaunchKernelGGL((k_schur<3, true>), dim3(nblk), dim3(kSchurWaves * 64), 0, s, w);
        else hipLaunchKernelGGL((k_schur<2, true>), dim3(nblk), dim3(kSchurWaves * 64), 0, s, w);
    } else {
        if (w.stereo) hipLaunchKernelGGL((k_schur<3, false>), dim3(nblk), dim3(kSchurWaves * 64), 0, s, w);
        else hipLaunchKernelGGL((k_schur<2, false>), dim3(nblk), dim3(kSchurWaves * 64), 0, s, w);
    }
    return hipGetLastError();
}

hipError_t launch_lambda_init(const DevWindow &w, hipStream_t s)
{
    hipLaunchKernelGGL(k_lambda_init, dim3(1), dim3(64), 0, s, w);
    return hipGetLastError();
}

hipError_t launch_backsub(const DevWindow &w, hipStream_t s)
{
    // (one workgroup more than the points need: its first wave takes the LM decision, decide_body)
    if (w.kcam) {
        if (w.stereo) hipLaunchKernelGGL((k_point_kf<true, true>), dim3(w.n_pt_blocks + 1), dim3(kPointBlock), point_lds_bytes_for(w, true, false), s, w);
        else hipLaunchKernelGGL((k_point_kf<true, false>), dim3(w.n_pt_blocks + 1), dim3(kPointBlock), point_lds_bytes_for(w, true, false), s, w);
    } else if (!w.lds_poses) {
        if (w.stereo) hipLaunchKernelGGL((k_point<true, true, false>), dim3(w.n_pt_blocks + 1), dim3(kPointBlock), point_lds_bytes(w, true), s, w);
        else hipLaunchKernelGGL((k_point<true, false, false>), dim3(w.n_pt_blocks + 1), dim3(kPointBlock), point_lds_bytes(w, true), s, w);
    } else if (w.stereo) hipLaunchKernelGGL((k_point<true, true, true>), dim3(w.n_pt_blocks + 1), dim3(kPointBlock), point_lds_bytes(w, true), s, w);
    else hipLaunchKernelGGL((k_point<true, false, true>), dim3(w.n_pt_blocks + 1), dim3(kPointBlock), point_lds_bytes(w, true), s, w);
    return hipGetLastError();
}

hipError_t launch_finalize(const DevWindow &w, hipStream_t s)
{
    hipLaunchKernelGGL(k_finalize, dim3((w.E + 255) / 256), dim3(256), 0, s, w);
    return hipGetLastError();
}

hipError_t launch_export(const DevWindow &w, const ExportDst &d, hipStream_t s)
{
    const long words = 7L * w.NP + 3L * w.P + w.E + (w.E + 7) / 8;
    long nb = (words + 255) / 256;
    nb = nb < 1 ? 1 : (nb > 1024 ? 1024 : nb);
    hipLaunchKernelGGL(k_export, dim3((unsigned)nb), dim3(256), 0, s, w, d);
    return hipGetLastError();
}

// ---- batched launches (movba_lba_run_batch).  stereo / lds_poses are properties of the whole batch (api.cpp checks) ----
hipError_t launch_init_batch(const BatchDev &b, int nblk, hipStream_t s)
{
    hipLaunchKernelGGL(k_init_pose_b, dim3(nblk), dim3(256), 0, s, b);
    return hipGetLastError();
}

hipError_t launch_point_batch(const BatchDev &b, int nblk, bool backsub, bool stereo, bool ldsp, size_t lds, hipStream_t s)
{
    const dim3 g(nblk), t(kPointBlock);
#define MOVBA_PB(B, S, L) hipLaunchKernelGGL((k_point_b<B, S, L>), g, t, lds, s, b)
    if (backsub) {
        if (stereo) { if (ldsp) MOVBA_PB(true, true, true); else MOVBA_PB(true, true, false); }
        else { if (ldsp) MOVBA_PB(true, false, true); else MOVBA_PB(true, false, false); }
    } else {
        if (stereo) { if (ldsp) MOVBA_PB(false, true, true); else MOVBA_PB(false, true, false); }
        else { if (ldsp) MOVBA_PB(false, false, true); else MOVBA_PB(false, false, false); }
    }
#undef MOVBA_PB
    return hipGetLastError();
}

size_t point_lds_bytes_for(const DevWindow &w, bool backsub, bool ldsp)
{
    DevWindow v = w; v.lds_poses = ldsp ? 1 : 0;
    return point_lds_bytes(v, backsub);
}

int schur_blocks(const DevWindow &w) { return 8 * (w.sched_per_xcd / kSchurWaves); }

hipError_t launch_schur_batch(const BatchDev &b, int nblk, int mode, bool stereo, hipStream_t s)
{
    const dim3 g(nblk), t(kSchurWaves * 64);
    if (mode == 1) {
        if (stereo) hipLaunchKernelGGL((k_schur_b<3, true>), g, t, 0, s, b);
        else hipLaunchKernelGGL((k_schur_b<2, true>), g, t, 0, s, b);
    } else {
        if (stereo) hipLaunchKernelGGL((k_schur_b<3, false>), g, t, 0, s, b);
        else hipLaunchKernelGGL((k_schur_b<2, false>), g, t, 0, s, b);
    }
    return hipGetLastError();
}

hipError_t launch_lambda_init_batch(const BatchDev &b, hipStream_t s)
{
    hipLaunchKernelGGL(k_lambda_init_b, dim3(b.n), dim3(64), 0, s, b);
    return hipGetLastError();
}

hipError_t launch_finalize_batch(const BatchDev &b, int nblk, hipStream_t s)
{
    hipLaunchKernelGGL(k_finalize_b, dim3(nblk), dim3(256), 0, s, b);
    return hipGetLastError();
}

hipError_t configure_kernels(int nfree_max_lds_bytes)
{
    (void)nfree_max_lds_bytes;
    // allow the point kernels to use more than the default 64 KiB of LDS
    const void *pk[8] = { reinterpret_cast<const void *>(k_point<true, false, true>), reinterpret_cast<const void *>(k_point<false, false, true>),
                          reinterpret_cast<const void *>(k_point<true, true, true>), reinterpret_cast<const void *>(k_point<false, true, true>),
                          reinterpret_cast<const void *>(k_point_b<true, false, true>), reinterpret_cast<const void *>(k_point_b<false, false, true>),
                          reinterpret_cast<const void *>(k_point_b<true, true, true>), reinterpret_cast<const void *>(k_point_b<false, true, true>) };
    for (const void *f : pk) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace movba
