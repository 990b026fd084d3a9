// C-ABI of libmovba.so (include/movba.h): handle, HBM arena, upload / run / download and
// the host side of the LM loop.  The host only feeds the stream: every numerical decision
// (gain ratio, accept/reject, lambda) is taken on the device by k_decide, which publishes
// its progress in pinned host memory so the host can stay a bounded number of trial sets
// ahead without a stream synchronise per trial.
//
// Replaces, behind Optimizer::LocalBundleAdjustment (/root/reference/src/Optimizer.cc:461-841),
// the g2o objects set up at :532-545 and driven at :754-755, and the gate at :757-775.
#include <hip/hip_runtime.h>
#include <sched.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "device_types.h"
#include "kernels.h"
#include "pose_kernels.h"
#include "movba.h"
#include "structure.h"

using namespace movba;

namespace {

constexpr int kPhaseEvents = 16;
constexpr int kMaxGroups = 4;
enum KernelClass { KC_SCHUR = 0, KC_PCG, KC_BACKSUB, KC_DECIDE, KC_SETUP, KC_FINALIZE };
const char *kKernelNames[MOVBA_NKERNELS] = { "k_schur", "k_pcg", "k_point<backsub>", "k_decide", "setup(init+linearize+lambda)", "k_finalize" };

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct EventPair { hipEvent_t a, b; int cls; };

// One helper thread per handle: copies the caller's big arrays into the pinned staging buffer while the calling thread
// runs the grouping / validation pass over the edges (both are memory-bound single-thread loops of ~0.1 ms at cfg3).
// Sleeps on a condition variable between uploads.
struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> job;
    int state = 0;              // 0 idle, 1 job posted, 2 job done
    bool quit = false;
    std::atomic<int> posted{0}; // set with state = 1: what the thread polls while it stays awake between jobs
    int spin_ms = 4;            // how long it stays awake after a job (0 with movba_options::host_wait = 1: it sleeps at once)
    void post(std::function<void()> j)
    {
        if (!th.joinable())
            th = std::thread([this] {
                std::unique_lock<std::mutex> lk(m);
                for (;;) {
                    // stay awake for a moment after a job: back-to-back uploads find the thread running instead of paying a
                    // futex wake-up (tens to hundreds of microseconds) for 70 us of copying
                    if (state != 1 && !quit) {
                        lk.unlock();
                        const auto t_spin = std::chrono::steady_clock::now();
                        while (posted.load(std::memory_order_acquire) == 0 &&
                               std::chrono::steady_clock::now() - t_spin < std::chrono::milliseconds(spin_ms)) {
#if defined(__x86_64__)
                            __builtin_ia32_pause();
#endif
                        }
                        lk.lock();
                    }
                    cv.wait(lk, [&] { return state == 1 || quit; });
                    posted.store(0, std::memory_order_relaxed);
                    if (quit) return;
                    std::function<void()> jb = std::move(job);
                    lk.unlock();
                    jb();
                    lk.lock();
                    state = 2;
                    cv.notify_all();
                }
            });
        { std::lock_guard<std::mutex> lk(m); job = std::move(j); state = 1; posted.store(1, std::memory_order_release); }
        cv.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return state != 1; });
        state = 0;
    }
    ~Worker()
    {
        if (th.joinable()) {
            { std::lock_guard<std::mutex> lk(m); quit = true; }
            cv.notify_all();
            th.join();
        }
    }
};

}  // namespace

// Diagnostic switches of the PROCESS, read from the environment once, at their first use (never on the path of a solve):
//   MOVBA_WATCHDOG_MS        host gives up a device that makes no progress for that long (default 60 000)
//   MOVBA_TIME_UPLOAD / MOVBA_TIME_SOLVE   print the phases of every upload / solve call to stderr
//   MOVBA_DENSE_STAMPS       the one-launch direct solver records its tasks' clock stamps
//   MOVBA_DENSE_MULTILAUNCH  the direct solver one launch per block column (dense_solve.hip) for every window
//   MOVBA_BAND = 0 / 1       banded factorisation never / wherever its band fits (handles made with solver = 0)
//   MOVBA_BATCH_GROUPS       number of stream groups of a batched run
// Everything a TEST switches (structure pass on the host, entry formats, a late helper thread, short device-side waits, a
// smaller device, a forced park of k_band) is a per-handle hook of the test build only: -DMOVBA_TEST_HOOKS, libmovba_hooks.so,
// movba_test_hook().  The product library has neither the symbol nor the branches.
struct ProcessSwitches {
    double watchdog_ms = 60000.0;
    bool time_upload = false, time_solve = false, dense_stamps = false, dense_multilaunch = false;
    int band = -1, batch_groups = 0;
};
static const ProcessSwitches &process_switches()
{
    static const ProcessSwitches sw = [] {
        ProcessSwitches v;
        if (const char *e = std::getenv("MOVBA_WATCHDOG_MS")) { const double x = std::atof(e); if (x > 0.0) v.watchdog_ms = x; }
        v.time_upload = std::getenv("MOVBA_TIME_UPLOAD") != nullptr; v.time_solve = std::getenv("MOVBA_TIME_SOLVE") != nullptr;
        v.dense_stamps = std::getenv("MOVBA_DENSE_STAMPS") != nullptr; v.dense_multilaunch = std::getenv("MOVBA_DENSE_MULTILAUNCH") != nullptr;
        if (const char *e = std::getenv("MOVBA_BAND")) v.band = std::atoi(e);
        if (const char *e = std::getenv("MOVBA_BATCH_GROUPS")) v.batch_groups = std::atoi(e);
        return v;
    }();
    return sw;
}

struct TestHooks {
    int host_structure = 0;         // structure pass on the host even where the device would build it
    int host_grouping = 0;          // grouping pass (build_basic) on the calling thread even where the device would run it
    int entries_unpacked = 0;       // 12-byte schur entries where the 8-byte packed form would do
    int no_sorted_structure = 0;    // beyond the pair-bin masks: host structure pass instead of the sort-based device pass
    int pcg_packed = 0;             // packed layout of the PCG's pair sums where the padded one would do
    int helper_delay_us = 0;        // the upload's helper thread starts that much later
    long long wait_ticks = -1;      // >= 0: bound of the in-launch waits of a run's FIRST attempt (10 ns ticks)
    int band_park_trial = -1;       // >= 0: k_band treats that trial's factorisation as one that met a non-positive pivot
};
#ifdef MOVBA_TEST_HOOKS
#define HOOK(h, field) ((h)->hooks.field)
#else
static constexpr TestHooks kNoHooks{};
#define HOOK(h, field) (kNoHooks.field)
#endif

struct movba_handle {
    int device = 0;
    int device_cus = 256;               // compute units of the device (or of this process's partition of it): bounds the one-launch direct solver's workgroups
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t copy_stream = nullptr;  // H2D of the caller's arrays, issued by the helper thread of movba_lba_upload (shared by the
                                        // handles of a device: every extra stream of the process competes for the few hardware queues,
                                        // and two streams of a batched run that land on one queue run in turns)
    hipEvent_t copy_event = nullptr;
#ifdef MOVBA_TEST_HOOKS
    TestHooks hooks;                    // (test build only: movba_test_hook)
#endif
    unsigned *ingest_counter = nullptr; // device word: workgroups of k_ingest that are through (IngestArgs::counter), never reset
    unsigned ingest_expect = 0;         // its value once every launch queued so far is through
    bool early_setup = false;           // the upload has queued k_init_pose and the first linearisation itself (behind the edge data, in
                                        // the shadow of its own pair layout): the next run starts with the Hpp pass
    int sync_retries = 0;               // > 0: this run's first attempt gave up that many in-launch waits and was repeated on the paths without any
    hipEvent_t edgeb_event = nullptr;   // the derived edge arrays sent early on the copy stream have arrived; device grouping pass: the
                                        // index arrays have been read out of host memory (the big arrays' DMA starts behind it)
    uint64_t count_seq = 0;             // uploads that went through the device structure pass (what the host polls for in the counts buffer)
    movba_options opt{};
    // device arena
    char *arena = nullptr;
    size_t arena_cap = 0;
    uint64_t arena_gen = 0;             // bumped by every (re)allocation: hipFree + hipMalloc may return the same address
    // pinned staging
    char *stage = nullptr;
    size_t stage_cap = 0;
    HostStatus *hstat = nullptr;        // pinned, mapped
    HostStatus *hstat_dev = nullptr;
    Ctrl *ctrl_host = nullptr;          // pinned copy of the device Ctrl
    Ctrl *ctrl_host_dev = nullptr;      // its device view (written by k_finalize)
    double *pose_export = nullptr;      // registered device buffer for the final poses
    int64_t pose_export_cap = 0;
    char *stage_dev = nullptr;          // device view of the pinned staging buffer (written by k_export)
    // current window
    bool uploaded = false, ran = false;
    bool export_in_run = false;         // this run's results were written to the staging buffer behind its last kernel
    bool export_hint = false;           // set by movba_lba_solve around its run: a download follows at once
    // movba_lba_solve: result arrays of the caller that lie in movba_host_alloc memory (poses, points, chi2): device view
    // the export kernel writes to, and the host pointer it stands for (download skips the copy-out of exactly that array)
    unsigned long long *user_dst[3] = {nullptr, nullptr, nullptr};
    const void *user_host[3] = {nullptr, nullptr, nullptr};
    bool exported[3] = {true, true, true};      // which of poses / points / chi2 the run's export wrote (the caller asked for)
    Structure st;
    DevWindow win{};
    size_t h2d_bytes = 0;
    const volatile uint8_t *stop = nullptr;
    int early_status = MOVBA_OK;        // decided at upload (MOVBA_EMPTY / MOVBA_NO_FIXED): holds for every run of the window
    int run_status = MOVBA_OK;          // decided per run (MOVBA_STOPPED when the flag was up before the solve)
    PcgParams pp{};
    bool rows_kernel = false;
    bool band = false;                  // this window's reduced system is solved by the single-workgroup banded factorisation (band_kernel.hip)
    int band_bw = 0;                    // its half bandwidth in blocks
    DensePlan dplan;                    // static schedule of the one-launch direct solver (kept across uploads of the same size)
    int dplan_nt = -1;
    bool dense_flags_clean = false;     // the window's hand-off flags have been zeroed since its upload (done before the first direct launch)
    unsigned dense_epoch = 0;           // direct launches on this window so far: the value a flag of the current launch carries
    char *scratch = nullptr;            // structure-pass temporaries (struct_kernels.hip)
    size_t scratch_cap = 0;
    char *scratch2 = nullptr;           // ... of the sort-based fill (struct_sort.hip): keys, values, rocPRIM's temporary storage
    size_t scratch2_cap = 0;
    // movba_lba_run_batch (kept by the first handle of a batch): device views, PCG plans and block prefixes of the windows
    char *batch_host = nullptr, *batch_dev = nullptr;
    size_t batch_cap = 0;
    hipStream_t batch_streams[kMaxGroups] = {};   // extra streams of a batched run (groups of windows run out of phase); [0] unused
    hipEvent_t batch_ev[kMaxGroups + 1] = {};     // [0]: fork from the callers' stream, [g]: join of group g
    hipEvent_t batch_phase_ev[kMaxGroups][kPhaseEvents] = {};   // ring: end of group g's schur launch of trial t (t mod 16)
    // pose-only scratch
    char *pose_arena = nullptr;
    size_t pose_cap = 0;
    Worker packer;                      // helper thread of movba_lba_upload
    // profiling
    std::vector<EventPair> ev_used;
    std::vector<hipEvent_t> ev_pool;
    movba_profile prof{};
};

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            std::fprintf(stderr, "libmovba: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return MOVBA_ERR_HIP;                                                             \
        }                                                                                     \
    } while (0)

namespace {

// One launch of the one-launch direct solver at a time per device.  Its workgroups wait for one another, which is safe while all
// of them are resident (<= 248 of the 256 CUs); two such launches dispatched at the same moment from two streams could each get
// part of the chip and wait for workgroups that cannot start until the other gives way - until the 20 ms clock ends both with
// MOVBA_ERR_DEVICE_WAIT.  As long as ONE stream uses the solver on a device (MoV-SLAM: the LocalMapping thread) nothing is
// added; from the moment a second stream does, every launch waits for the one before it through an event.
struct DenseGate {
    std::mutex mu;
    hipStream_t only_stream = nullptr;      // single mode: the one stream that has launched the solver on this device
    bool multi = false;
    hipEvent_t ev = nullptr;                // multi mode: completion of the latest launch, whichever stream it was on
    bool ev_recorded = false;
};
DenseGate &dense_gate(int device) { static DenseGate gates[32]; return gates[device & 31]; }

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// one copy stream per device for all handles (created on first use, kept for the life of the process)
hipStream_t shared_copy_stream(int device)
{
    static std::mutex mu;
    static std::vector<hipStream_t> streams;
    std::lock_guard<std::mutex> lk(mu);
    if ((int)streams.size() <= device) streams.resize(device + 1, nullptr);
    if (!streams[device] && hipStreamCreateWithFlags(&streams[device], hipStreamNonBlocking) != hipSuccess) streams[device] = nullptr;
    return streams[device];
}

// between two looks at the device's progress word: spin (lowest latency: the LM chain is ~100 us per trial), or give the
// core away (movba_options::host_wait = 1: the LocalMapping thread then does not starve a Tracking thread it shares a core with)
inline void host_relax(int mode)
{
    if (mode == 1) { sched_yield(); return; }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
}

// Words the device writes into pinned host memory while the host polls them (k_decide's progress word, the PCG's park
// counter, the structure pass's sequence number): read with acquire loads — what is published before them (the counts, the
// controller's copy) is read after them — and written from the host with release stores.
inline uint64_t rd_progress(const HostStatus *hs) { return __atomic_load_n(&hs->progress, __ATOMIC_ACQUIRE); }
inline int32_t rd_pause(const HostStatus *hs) { return __atomic_load_n(&hs->pause_seq, __ATOMIC_ACQUIRE); }
inline void wr_stop(HostStatus *hs, int32_t v) { __atomic_store_n(&hs->stop, v, __ATOMIC_RELEASE); }
inline bool caller_stop(const volatile uint8_t *p) { return p && __atomic_load_n(p, __ATOMIC_RELAXED) != 0; }
// how long the device may go without ANY progress (a changed progress word, or new work queued) before the host gives up
inline double watchdog_ms() { return process_switches().watchdog_ms; }

struct Carver {
    size_t off = 0;
    template <typename T> size_t take(size_t count)
    {
        const size_t o = off;
        off = align_up(off + count * sizeof(T), 256);
        return o;
    }
};

hipEvent_t get_event(movba_handle *h)
{
    if (!h->ev_pool.empty()) { hipEvent_t e = h->ev_pool.back(); h->ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

struct ScopedEvents {
    movba_handle *h; EventPair p{}; bool on; hipStream_t st;
    ScopedEvents(movba_handle *h_, int cls, hipStream_t st_ = nullptr) : h(h_), on(((h_->opt.profile >> cls) & 1) != 0), st(st_ ? st_ : h_->stream)
    {
        if (!on) return;
        p.a = get_event(h); p.b = get_event(h); p.cls = cls;
        if (!p.a || !p.b) { on = false; return; }
        (void)hipEventRecord(p.a, st);
    }
    ~ScopedEvents()
    {
        if (!on) return;
        (void)hipEventRecord(p.b, st);
        h->ev_used.push_back(p);
    }
};

void harvest_events(movba_handle *h)
{
    for (const EventPair &p : h->ev_used) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            h->prof.ms[p.cls] += ms;
            h->prof.launches[p.cls] += 1;
        }
        h->ev_pool.push_back(p.a);
        h->ev_pool.push_back(p.b);
    }
    h->ev_used.clear();
}

int ensure_arena(movba_handle *h, size_t bytes)
{
    if (bytes <= h->arena_cap) return MOVBA_OK;
    if (h->arena) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (h->copy_stream) HIP_TRY(hipStreamSynchronize(h->copy_stream));
        HIP_TRY(hipFree(h->arena)); h->arena = nullptr; h->arena_cap = 0;
    }
    const size_t cap = align_up(bytes + bytes / 4, 1 << 20);
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->arena), cap));
    h->arena_cap = cap;
    h->arena_gen += 1;
    return MOVBA_OK;
}

int ensure_stage(movba_handle *h, size_t bytes)
{
    if (bytes <= h->stage_cap) return MOVBA_OK;
    if (h->stage) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (h->copy_stream) HIP_TRY(hipStreamSynchronize(h->copy_stream));
        HIP_TRY(hipHostFree(h->stage)); h->stage = nullptr; h->stage_cap = 0;
    }
    const size_t cap = align_up(bytes + bytes / 4, 1 << 20);
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&h->stage), cap, hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&h->stage_dev), h->stage, 0));
    h->stage_cap = cap;
    return MOVBA_OK;
}

// blocks handed out by movba_host_alloc: host address range -> device view (process-wide, a handful of entries)
struct HostBlock { char *host; size_t bytes; char *dev; };
std::mutex g_host_blocks_mu;
std::vector<HostBlock> g_host_blocks;

// device view of [p, p + bytes) if it lies inside a movba_host_alloc block, else nullptr
unsigned long long *host_block_view(const void *p, size_t bytes)
{
    if (!p) return nullptr;
    const char *c = static_cast<const char *>(p);
    std::lock_guard<std::mutex> lk(g_host_blocks_mu);
    for (const HostBlock &b : g_host_blocks)
        if (c >= b.host && c + bytes <= b.host + b.bytes) return reinterpret_cast<unsigned long long *>(b.dev + (c - b.host));
    return nullptr;
}

// where k_export leaves a window's results in the pinned staging buffer
struct ExportLayout { size_t o_pose, o_pt, o_chi, o_out, end; };
ExportLayout export_layout(const DevWindow &w)
{
    ExportLayout L;
    L.o_pose = 0;
    L.o_pt = align_up(sizeof(double) * 7 * (size_t)w.NP, 256);
    L.o_chi = L.o_pt + align_up(sizeof(double) * 3 * (size_t)w.P, 256);
    L.o_out = L.o_chi + align_up(sizeof(double) * (size_t)w.E, 256);
    L.end = L.o_out + align_up((size_t)w.E + 8, 256);
    return L;
}

ExportDst export_dst(char *stage_dev, const ExportLayout &L, bool poses, bool points, bool chi2)
{
    ExportDst dst;
    dst.poses = poses ? reinterpret_cast<unsigned long long *>(stage_dev + L.o_pose) : nullptr;
    dst.points = points ? reinterpret_cast<unsigned long long *>(stage_dev + L.o_pt) : nullptr;
    dst.chi2 = chi2 ? reinterpret_cast<unsigned long long *>(stage_dev + L.o_chi) : nullptr;
    dst.outlier = reinterpret_cast<unsigned long long *>(stage_dev + L.o_out);
    return dst;
}

}  // namespace

extern "C" {

int movba_version(void) { return MOVBA_VERSION; }

#ifdef MOVBA_TEST_HOOKS
// Test build only (libmovba_hooks.so; declared by the tests themselves, not by include/movba.h): sets one hook of a handle.
// "device_cus" plans the one-launch direct solver for a device with fewer compute units than this one has.
int movba_test_hook(movba_handle *h, const char *name, long long value)
{
    if (!h || !name) return MOVBA_ERR_ARG;
    const std::string n(name);
    if (n == "host_structure") h->hooks.host_structure = (int)value;
    else if (n == "host_grouping") h->hooks.host_grouping = (int)value;
    else if (n == "entries_unpacked") h->hooks.entries_unpacked = (int)value;
    else if (n == "no_sorted_structure") h->hooks.no_sorted_structure = (int)value;
    else if (n == "pcg_packed") h->hooks.pcg_packed = (int)value;
    else if (n == "helper_delay_us") h->hooks.helper_delay_us = (int)value;
    else if (n == "wait_ticks") h->hooks.wait_ticks = value;
    else if (n == "band_park_trial") h->hooks.band_park_trial = (int)value;
    else if (n == "device_cus") { if (value > 0 && value < h->device_cus) h->device_cus = (int)value; }
    else return MOVBA_ERR_ARG;
    return MOVBA_OK;
}
#endif

const char *movba_status_string(int s)
{
    switch (s) {
    case MOVBA_OK: return "ok";
    case MOVBA_STOPPED: return "stopped before solve";
    case MOVBA_NO_FIXED: return "no fixed keyframe";
    case MOVBA_EMPTY: return "nothing to optimise";
    case MOVBA_ERR_ARG: return "invalid argument";
    case MOVBA_ERR_HIP: return "HIP runtime error";
    case MOVBA_ERR_STATE: return "invalid call order";
    case MOVBA_ERR_DEVICE_WAIT: return "direct solver: a workgroup gave up waiting for another";
    case MOVBA_ERR_TOO_LARGE: return "reduced system too large for the direct solver";
    default: return "unknown";
    }
}

int movba_create(movba_handle **out, int device, void *stream, const movba_options *opt)
{
    if (!out) return MOVBA_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        std::fprintf(stderr, "libmovba: no HIP device available — the local-BA path has no CPU fallback\n");
        return MOVBA_ERR_HIP;
    }
    if (device < 0 || device >= ndev) return MOVBA_ERR_ARG;
    movba_handle *h = new (std::nothrow) movba_handle();
    if (!h) return MOVBA_ERR_HIP;
    h->device = device;
    h->opt.pcg_rel_tol = 1e-10; h->opt.pcg_max_iters = 0; h->opt.run_ahead = 2; h->opt.profile = 0; h->opt.pcg_coarse = 1; h->opt.host_wait = 0; h->opt.pcg_spill = 0; h->opt.solver = 0; h->opt.reorder = 0;
    if (opt) {
        if (opt->pcg_rel_tol > 0) h->opt.pcg_rel_tol = opt->pcg_rel_tol;
        if (opt->pcg_max_iters > 0) h->opt.pcg_max_iters = opt->pcg_max_iters;
        if (opt->run_ahead > 0) h->opt.run_ahead = opt->run_ahead;
        h->opt.profile = opt->profile;
        if (opt->pcg_coarse < 0) h->opt.pcg_coarse = 0;
        h->opt.host_wait = opt->host_wait == 1 ? 1 : 0;
        h->opt.pcg_spill = opt->pcg_spill == 1 ? 1 : 0;
        h->opt.solver = (opt->solver >= 1 && opt->solver <= 3) ? opt->solver : 0;
        h->opt.reorder = opt->reorder == -1 ? -1 : 0;
    }
    if (h->opt.host_wait == 1) h->packer.spin_ms = 0;      // (a caller that asks for yielding waits does not want a spinning helper either)
    for (int k = 0; k < MOVBA_NKERNELS; ++k) h->prof.name[k] = kKernelNames[k];
    if (hipSetDevice(device) != hipSuccess) { delete h; return MOVBA_ERR_HIP; }
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) h->device_cus = cus; }
    if (stream) { h->stream = static_cast<hipStream_t>(stream); }
    else {
        if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return MOVBA_ERR_HIP; }
        h->own_stream = true;
    }
    if ((h->copy_stream = shared_copy_stream(device)) == nullptr ||
        hipEventCreateWithFlags(&h->copy_event, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->edgeb_event, hipEventDisableTiming) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&h->ingest_counter), 256) != hipSuccess || hipMemset(h->ingest_counter, 0, 256) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **>(&h->hstat), sizeof(HostStatus), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer(reinterpret_cast<void **>(&h->hstat_dev), h->hstat, 0) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **>(&h->ctrl_host), sizeof(Ctrl), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer(reinterpret_cast<void **>(&h->ctrl_host_dev), h->ctrl_host, 0) != hipSuccess ||
        configure_kernels(0) != hipSuccess || configure_pcg_rows() != hipSuccess || configure_band() != hipSuccess || configure_struct_kernels() != hipSuccess ||
        configure_dense_kernels() != hipSuccess || configure_dense_persist() != hipSuccess ||
        configure_pose_kernels() != hipSuccess) {
        movba_destroy(h);
        return MOVBA_ERR_HIP;
    }
    std::memset((void *)h->hstat, 0, sizeof(HostStatus));
    *out = h;
    return MOVBA_OK;
}

void movba_destroy(movba_handle *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);        // shared: stays
    if (h->copy_event) (void)hipEventDestroy(h->copy_event);
    if (h->edgeb_event) (void)hipEventDestroy(h->edgeb_event);
    if (h->ingest_counter) (void)hipFree(h->ingest_counter);
    harvest_events(h);
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
    if (h->arena) (void)hipFree(h->arena);
    if (h->pose_arena) (void)hipFree(h->pose_arena);
    for (hipStream_t st : h->batch_streams) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (hipEvent_t e : h->batch_ev) if (e) (void)hipEventDestroy(e);
    for (auto &ring : h->batch_phase_ev) for (hipEvent_t e : ring) if (e) (void)hipEventDestroy(e);
    if (h->batch_dev) (void)hipFree(h->batch_dev);
    if (h->batch_host) (void)hipHostFree(h->batch_host);
    if (h->scratch) (void)hipFree(h->scratch);
    if (h->scratch2) (void)hipFree(h->scratch2);
    if (h->stage) (void)hipHostFree(h->stage);
    if (h->hstat) (void)hipHostFree((void *)h->hstat);
    if (h->ctrl_host) (void)hipHostFree(h->ctrl_host);
    {
        DenseGate &g = dense_gate(h->device);       // (queue_direct: the gate must not keep a stream that is about to go)
        std::lock_guard<std::mutex> lk(g.mu);
        if (g.only_stream == h->stream) g.only_stream = nullptr;       // (drained above: nothing of this stream is in flight)
    }
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int movba_structure_probe(const movba_lba_desc *desc, movba_structure_info *info, int32_t *edge_perm, int32_t *free_index)
{
    if (!desc || !info) return MOVBA_ERR_ARG;
    Structure s;
    const int rc = build_structure(*desc, s);
    if (rc < 0) return rc;
    if (s.nfree > MOVBA_MAX_FREE_KEYFRAMES) return MOVBA_ERR_TOO_LARGE;
    info->n_free = s.nfree; info->n_pairs = s.npairs; info->n_entries = s.nentries; info->n_items = s.nitems;
    info->max_degree = s.max_degree; info->already_grouped = s.already_grouped ? 1 : 0; info->reordered = s.reordered ? 1 : 0; info->pad_s = 0;
    info->pcg_on_chip = 0; info->pcg_overflow = 0; info->pcg_max_wave_entries = 0; info->n_row_entries = (int32_t)s.row_ent.size();
    if (rc == MOVBA_OK && s.nfree > 0) {
        PcgParams pp{};
        if (pcg_rows_supported(s.nfree, s.row_ptr.data(), &pp)) {
            info->pcg_on_chip = 1; info->pcg_overflow = pp.overflow;
            for (int wv = 0; wv < kPcgRowsThreads / 64; ++wv)
                info->pcg_max_wave_entries = std::max(info->pcg_max_wave_entries, s.row_ptr[pp.wave_row0[wv + 1]] - s.row_ptr[pp.wave_row0[wv]]);
        }
    }
    {   // launch schedule and pose-major slots: self-checks reported to the caller (CPU tests)
        info->n_sched_slots = (int32_t)s.sched.size(); info->sched_items = 0; info->sched_max_permille = 0; info->slots_ok = 1;
        // (every item once: the wave slots that share it lie side by side in ONE workgroup, their places 0 .. n - 1 in order, and
        //  their entry ranges cut the item's range without gap or overlap)
        std::vector<uint8_t> seen(s.nitems > 0 ? s.nitems : 1, 0);
        int64_t segw[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, tot = 0;
        for (size_t k = 0; k < s.sched.size(); ++k) {
            const SchedItem &it = s.sched[k];
            if (it.tag < 0 || (it.sub & 0xff) != 0) continue;
            const int item = it.tag >> 1, nw = it.sub >> 8;
            bool ok = item < s.nitems && !seen[item] && nw >= 1 && nw <= 4 && (k & 3) + (size_t)nw <= 4 && s.items[item].begin == it.begin;
            int32_t at = it.begin;
            for (int q = 0; ok && q < nw; ++q) {
                const SchedItem &w = s.sched[k + q];
                ok = w.tag == it.tag && w.sub == (q | (nw << 8)) && w.begin == at && w.end >= w.begin;
                at = w.end;
            }
            if (!ok || at != s.items[item].end) { info->sched_items = -1; break; }
            seen[item] = 1; info->sched_items++;
            const int64_t wgt = (int64_t)(s.items[item].end - s.items[item].begin) * ((it.tag & 1) ? 3 : 2) + 128;
            if (s.sched_per_xcd > 0) segw[k / s.sched_per_xcd] += wgt;
            tot += wgt;
        }
        if (tot > 0) { int64_t mx = 0; for (int64_t v : segw) mx = std::max(mx, v); info->sched_max_permille = (int32_t)(mx * 8000 / tot); }
        int64_t nfree_edges = 0;
        for (int32_t v : s.slot) nfree_edges += v >= 0;
        std::vector<uint8_t> hit((size_t)nfree_edges + 1, 0);
        for (int32_t v : s.slot) if (v >= 0) { if (v >= nfree_edges || hit[v]) { info->slots_ok = 0; break; } hit[v] = 1; }
    }
    if (edge_perm) {
        if (s.perm.empty()) for (int e = 0; e < s.E; ++e) edge_perm[e] = e;       // already grouped: identity
        else std::memcpy(edge_perm, s.perm.data(), sizeof(int32_t) * s.perm.size());
    }
    if (free_index) std::memcpy(free_index, s.hidx.data(), sizeof(int32_t) * s.hidx.size());
    return rc;
}

int movba_dense_plan_probe(int32_t n_block_cols, int32_t max_groups, int32_t max_slots, int32_t info[4], int32_t *task_ptr, int32_t task_ptr_cap,
                           int32_t *tasks, int32_t tasks_cap)
{
    if (!info || n_block_cols < 1) return MOVBA_ERR_ARG;
    DensePlan p;
    build_dense_plan(n_block_cols, p, max_groups > 0 ? max_groups : kDenseMaxGroups, max_slots > 0 ? max_slots : kDenseMaxSlots);
    info[0] = p.ok ? 1 : 0; info[1] = p.G; info[2] = p.slots; info[3] = (int32_t)p.tasks.size();
    if (p.ok && task_ptr) for (int g = 0; g <= p.G && g < task_ptr_cap; ++g) task_ptr[g] = p.task_ptr[g];
    if (p.ok && tasks) std::memcpy(tasks, p.tasks.data(), sizeof(DenseTask) * std::min<size_t>(p.tasks.size(), (size_t)std::max(tasks_cap, 0)));
    return MOVBA_OK;
}

// =====================================================================================================================
// movba_lba_upload, in phases.  One Upload object lives for the duration of one call; its members say who owns what.
//
//   threads     the CALLER's thread runs every phase below; the handle's HELPER thread (movba_handle::packer) runs exactly
//               one closure per call, posted by post_helper(), which copies the caller's big arrays into the staging buffer
//               and sends them to the arena on the copy stream.
//   hand-off    HelperHandOff: the two things the helper produces for the caller (idx_ready, copy_err) and the things
//               fixed before it is posted.  Everything else in Upload belongs to the caller's thread alone.
//   staging     the pinned buffer is laid out like the edge region of the arena (EdgeLayout): the helper writes
//               [gpose, gpoint) index copies and [raw_begin, grouped_end) there, the caller everything else - and the
//               helper's parts only after join_helper() (+ copy_event, when they are to be rewritten).
//   streams     h->stream: structure kernels, derived arrays, pair region, and later the solve.  h->copy_stream: the
//               caller's arrays (helper) and the early copy of the derived edge arrays.
//   events      copy_event  (copy stream -> stream): the caller's arrays have arrived; recorded by the helper, waited for
//                           by the stream before the solve's first kernel (send_pairs) and by the host before the staging
//                           copy of those arrays is rewritten (ungrouped windows);
//               edgeb_event (copy stream -> stream): the derived edge arrays sent early have arrived; waited for by the
//                           stream ahead of the slot-completion / fill kernels (queue_edge_b).
//   arena       may be reallocated by ensure_arena() in lay_out_rest(): its generation (arena_gen) at the time something
//               was queued tells whether that something has to be queued again.
// =====================================================================================================================
namespace {

// memcpy that takes an empty source (a vector without storage has a null data()): copying nothing from nowhere is undefined
// behaviour for memcpy itself (found by UBSan over the host build, tests/hipstub)
inline void put(void *dst, const void *src, size_t bytes) { if (bytes) std::memcpy(dst, src, bytes); }

// byte offsets of the edge region's arrays, the same in the arena and in the staging buffer; fixed by the caller's counts
struct EdgeLayout {
    // (what the device structure pass reads comes first: it is copied ahead of the rest)
    size_t gpose = 0, ptstart = 0, hidx = 0;
    size_t a_end = 0;                   // end of that first part
    size_t gpoint = 0, free_pose = 0, slot = 0;
    size_t base = 0;                    // first pose-major slot of every keyframe (slots are completed on the device)
    // (the caller's own arrays, contiguous: they cross the bus on the copy stream, straight from the helper thread)
    size_t raw_begin = 0, obs = 0, isig = 0, obsr = 0, pose0 = 0, point0 = 0, kcam = 0;
    size_t grouped_end = 0;             // end of the region when the caller's edges come grouped by map point
    size_t perm = 0;                    // only travels when they do not
    size_t max_end = 0;
    bool has_kcam = false;              // intrinsics by keyframe (src/Optimizer.cc:664, 690-695)
};

constexpr int kSortedMaxPoses = 1024;   // windows the sort-based device structure pass takes (pair counts back: 4 NP^2 bytes of pinned memory)

// (below this many edges the host builds grouping, pair counts and entry lists itself: its passes are a few microseconds then,
//  less than the device's chain of launches and the two trips across the bus - 0.43 against 0.47 ms per call at 693 edges,
//  0.54 against 0.52 at 7 102, scripts/small_paths.py)
constexpr int kDeviceStructureMinEdges = 2048;

struct HelperHandOff {
    // helper -> caller
    std::atomic<int> idx_ready{0};      // the caller's index arrays are in the staging buffer (release / acquire): what the
                                        // structure pass on the device waits for

    hipError_t copy_err = hipSuccess;   // read by the caller only after Worker::wait()
    hipError_t idx_err = hipSuccess;    // ... this one behind idx_ready (release / acquire)
    // fixed before the helper is posted, read by both
    char *arena = nullptr;              // the arena the helper sends to ...
    uint64_t arena_gen = 0;             // ... and its generation: a reallocation later on is told by it
    // caller only
    bool joined = false;
};

struct Upload {
    movba_handle *const h;
    const movba_lba_desc *const d;
    const int NP, P, E;
    Carver c;                           // the arena's layout, carved phase by phase
    EdgeLayout L;
    HelperHandOff ho;
    char *sg = nullptr;                 // staging buffer
    size_t misc_bytes = 0;              // its tail: counts back / pair ids out (device structure pass)
    bool stereo = false;
    bool done = false;                  // the call is complete (early status): run() returns rc as it stands
    // --- grouping ---
    bool rank_mode = true;              // the staging buffer's slot array holds ranks; pose_slot0 the keyframes' first slots
    int nf = 0, nb = 0, nbins = 0;
    size_t edge_bytes = 0;
    // --- edge copies ---
    uint64_t arena_gen_at_edge_copy = 0;
    bool edge_b_early = false, edge_b_stale = false, edge_b_queued = false;
    double upload_host_ms = 0.0;
    // --- grouping pass on the device (group_on_device) ---
    int nf_expect = 0;                  // non-fixed keyframes: the free keyframes of the window unless one of them has no edge
    bool raw_synced = false;
    bool direct_raw = false;            // the caller's big arrays lie in movba_host_alloc memory: they cross the bus from where they are
    bool dev_first = false;             // validation, point ranges, hessian indices and slots are the device's work: no pass over the edges here
    BasicDev bd{};
    size_t so_cnt = 0, so_err = 0, so_ent0 = 0, so_cntw = 0, so_pe = 0, so_info = 0, so_fixed = 0, so_H = 0;
    volatile int32_t *misc_seq = nullptr;
    int32_t seq = 0;
    // --- structure ---
    StructDev sd{};
    bool dev_structure = false, ent_packed = false, filled_early = false;
    bool scan_pending = false;          // k_struct_scan of this upload is still to be launched (in front of the fill)
    bool sorted_structure = false;      // the device pass of struct_sort.hip (beyond k_struct_pairs' 80 free keyframes)
    size_t s2_off = 0, s2_keys_in = 0, s2_keys_out = 0, s2_vals_in = 0, s2_tmp = 0, s2_tmp_bytes = 0, so_cntpt = 0;
    uint64_t fill_gen = 0;
    size_t noff = 0, o_ent = 0, o_slotpt = 0;
    // --- pair region / device-only region ---
    size_t pair_begin = 0, hole = 0, h2d = 0, total = 0;
    size_t o_items = 0, o_sched = 0, o_pi = 0, o_pj = 0, o_pis = 0, o_rowptr = 0, o_rowent = 0, o_plan = 0;
    size_t o_cg = 0, o_ch = 0, o_cp = 0, o_ce = 0, o_cij = 0, o_multi = 0, o_pid = 0, o_prange = 0, o_dtp = 0, o_dtk = 0;
    size_t o_st[2][11] = {};
    size_t o_obspm = 0, o_obsrpm = 0, o_part = 0, o_blocks = 0, o_blocks_ov = 0, o_blocks_c = 0, o_aci = 0, o_acitag = 0;
    size_t o_bp = 0, o_xp = 0, o_scale = 0, o_hmax = 0, o_tick = 0, o_recd = 0, o_imgb = 0, o_ctrl = 0, o_chi2 = 0, o_outl = 0;
    size_t o_dtiles = 0, o_ddiag = 0, o_dfail = 0, o_dx = 0, o_dflags = 0, o_dcontrib = 0, o_dstamps = 0;
    std::vector<int32_t> lane_plan;
    int rec_slots = 1;
    size_t ncb = 0;
    int ntile = 0;
    bool dense_one = false, dense_stamps = false;
    // --- timing ---
    double t0 = 0.0, lap_t = 0.0;
    bool lap_on = false;

    Upload(movba_handle *h_, const movba_lba_desc *d_) : h(h_), d(d_), NP(d_->n_poses), P(d_->n_points), E(d_->n_edges) {}
    // (every way out waits for the helper first: it reads the caller's arrays and writes to this object)
    // (... and in direct mode the copy engine reads the caller's own arrays: through with them before the caller has them back)
    ~Upload() { h->packer.wait(); if (direct_raw && !raw_synced) (void)hipStreamSynchronize(h->copy_stream); }

    const Structure &s() const { return h->st; }
    void lap(const char *what)
    {
        if (!lap_on) return;
        const double t = now_ms();
        std::fprintf(stderr, "libmovba[upload]: %-28s %.3f ms\n", what, t - lap_t);
        lap_t = t;
    }
    int join_helper()
    {
        if (ho.joined) return MOVBA_OK;
        h->packer.wait();
        ho.joined = true;
        if (ho.copy_err != hipSuccess) { std::fprintf(stderr, "libmovba: upload copy failed: %s\n", hipGetErrorString(ho.copy_err)); return MOVBA_ERR_HIP; }
        return MOVBA_OK;
    }

    int run(bool allow_dev_first);
    // phases, in the order run() takes them
    int begin();                        // arguments, edge layout, buffers, streams drained
    void post_helper();                 // the caller's arrays: staging buffer + copy stream, on the helper thread
    int group();                        // build_basic (grouping / validation); the early ways out
    bool dev_first_eligible() const;
    int group_on_device();              // ... the same on the device, behind the index arrays' way into the staging buffer
    void carve_state();                 // device-only arrays whose size follows from the caller's counts alone
    bool state_carved = false;
    int carve_scratch(bool basic);
    int launch_counts();
    int after_counts();
    int pack_derived();                 // derived edge arrays into the staging buffer
    int send_edge_a();                  // what the device structure pass reads -> stream; the rest early -> copy stream
    int structure_on_host();
    int structure_on_device();
    int structure_on_device_sorted();
    void choose_solver();
    int lay_out_rest();                 // pair region + device-only region; arena / staging buffer sized
    void pack_pairs();
    int send_pairs();
    void device_view();
    // pieces several phases share
    void pack_a(bool raw_too);          // what the device structure pass reads (first part of the edge region) ...
    void pack_b(bool raw_too);          // ... and the rest of the derived arrays
    void pack_edges(bool raw_too) { pack_a(raw_too); pack_b(raw_too); }
    int queue_edge_b();
    int launch_fill();
    int launch_slotpt();
    size_t ent_words() const { return (ent_packed ? 2 : 3) * noff + 4; }       // int32 words of the entry region
    char *sp(size_t o) const { return sg + (o - hole); }                        // staging address of a pair-region offset
};

int Upload::begin()
{
    HIP_TRY(hipSetDevice(h->device));
    h->uploaded = false; h->ran = false; h->early_status = MOVBA_OK; h->early_setup = false;
    t0 = lap_t = now_ms();
    lap_on = process_switches().time_upload;
    // ---- edge region of the arena, laid out from the caller's counts alone so that the helper thread can start copying the
    // caller's big arrays (observations, information, initial estimates: 3/4 of the region) into the pinned staging buffer
    // while this thread runs the grouping / validation pass.  Its H2D copies are queued as soon as it is packed, so that
    // the transfer runs while the pair structure is still being worked out ----
    if (NP < 0 || P < 0 || E < 0) return MOVBA_ERR_ARG;
    if ((NP && (!d->poses || !d->pose_fixed)) || (P && !d->points)) return MOVBA_ERR_ARG;
    if (E && (!d->edge_pose || !d->edge_point || !d->obs || !d->inv_sigma2)) return MOVBA_ERR_ARG;
    L.gpose = c.take<int32_t>(E); L.ptstart = c.take<int32_t>(P + 1); L.hidx = c.take<int32_t>(NP);
    L.a_end = c.off;
    L.gpoint = c.take<int32_t>(E);
    L.free_pose = c.take<int32_t>(NP + 1);
    L.slot = c.take<int32_t>(E);
    L.base = c.take<int32_t>(NP + 1);
    L.raw_begin = c.off;
    L.obs = c.take<double>(2 * (size_t)E); L.isig = c.take<double>(E);
    L.obsr = c.take<double>(d->obs_right ? E : 0);
    L.pose0 = c.take<double>(7 * (size_t)NP); L.point0 = c.take<double>(3 * (size_t)P);
    L.has_kcam = d->cam_kf || d->bf_kf;
    L.kcam = c.take<double>(L.has_kcam ? 8 * (size_t)NP : 0);
    L.grouped_end = c.off;
    L.perm = c.take<int32_t>(E);
    L.max_end = c.off;
    // (the device structure passes hand the pair counts back through the tail of the staging buffer: up to kSortedMaxPoses keyframes)
    const size_t nf_dev = (size_t)(NP <= kSortedMaxPoses ? NP : 80);
    misc_bytes = (nf_dev * nf_dev + 8) * sizeof(int32_t) * 2 + 4096 + sizeof(int32_t) * ((size_t)std::min(NP, 1024) + kBasicInfo + 8);
    int rc = ensure_stage(h, L.max_end + misc_bytes); if (rc) return rc;
    // first sizing of the arena: room for the states and pair lists too, so that it is not reallocated a moment later
    if (L.max_end > h->arena_cap) { rc = ensure_arena(h, 10 * L.max_end); if (rc) return rc; }
    HIP_TRY(hipStreamSynchronize(h->stream));     // staging buffer may still be in flight from a previous call
    HIP_TRY(hipStreamSynchronize(h->copy_stream));
    sg = h->stage;
    // (a window is a stereo window when any observation carries a right-image coordinate; looked up on this thread — the first
    //  stereo observation ends the scan — so that nothing the layout below depends on is produced by the helper thread)
    if (d->obs_right) for (int e = 0; e < E && !stereo; ++e) stereo = d->obs_right[e] >= 0.0;
    return MOVBA_OK;
}

// helper: straight copies of the caller's arrays (valid as they are when the edges come grouped by map point, the
// reference's own order; an ungrouped window has them permuted again by pack_b) on their way to the arena
void Upload::post_helper()
{
    ho.arena = h->arena;
    ho.arena_gen = h->arena_gen;
    if (direct_raw) {
        // Direct mode: every array of the caller lies in movba_host_alloc memory (pinned, mapped): nothing is staged.  The index
        // arrays, which the pair structure waits for, are read across the bus by a kernel on the handle's stream (k_ingest,
        // group_on_device: no copy command, no event between it and the grouping kernel); estimates, observations and
        // information go through the copy engine, queued here by the helper thread at once.  (A second k_ingest launch on the
        // copy stream was tried for them: the two streams share a hardware queue, and every kernel of the structure chain
        // queued behind it waited for its 60 us; the copy engine's commands cost ~8 us of latency each but run beside anything.)
        movba_handle *const hh = h;
        const movba_lba_desc *const dd = d;
        const EdgeLayout lay = L;
        char *const stage = sg;
        const int np = NP, p = P, e = E;
        HelperHandOff *const out = &ho;
        h->packer.post([=]() {
            hipError_t err = hipSetDevice(hh->device);
            // (The bus is shared: beside these 3.5 MB the index arrays' k_ingest takes 27 us for its 0.9 MB instead of 19 - 22.
            //  Holding the copy commands back behind that launch was tried: the last, small copy of the chain is a blit kernel,
            //  which then queued up behind the structure kernels of the handle's stream, and the solve's first kernels - which
            //  wait for it - started 25 us later: 1.107 ms per call against 1.09.  With that small copy sent first and only the
            //  large commands held back: the copy engine's chain - three commands of ~8 us latency each and 78 us of transfer -
            //  then ends ~25 us behind the structure kernels instead of ahead of them, and the solve's first kernels wait for IT:
            //  1.09 - 1.10 ms.  The bus time of the upload, ~85 us for 4.4 MB, has to start at once.)
            auto dma = [&](size_t to, const void *from, size_t bytes) {
                if (err == hipSuccess && bytes) err = hipMemcpyAsync(out->arena + to, from, bytes, hipMemcpyHostToDevice, hh->copy_stream);
            };
            dma(lay.obs, dd->obs, sizeof(double) * 2 * (size_t)e);
            dma(lay.isig, dd->inv_sigma2, sizeof(double) * (size_t)e);
            if (dd->obs_right) dma(lay.obsr, dd->obs_right, sizeof(double) * (size_t)e);
            dma(lay.point0, dd->points, sizeof(double) * 3 * (size_t)p);
            dma(lay.pose0, dd->poses, sizeof(double) * 7 * (size_t)np);
            if (lay.has_kcam) {
                double *kc = reinterpret_cast<double *>(stage + lay.kcam);
                for (int i = 0; i < np; ++i) {
                    const double *ck = dd->cam_kf ? dd->cam_kf + 4 * (size_t)i : &dd->fx;      // (fx, fy, cx, cy are contiguous in the descriptor)
                    kc[8 * i] = ck[0]; kc[8 * i + 1] = ck[1]; kc[8 * i + 2] = ck[2]; kc[8 * i + 3] = ck[3];
                    kc[8 * i + 4] = dd->bf_kf ? dd->bf_kf[i] : dd->bf; kc[8 * i + 5] = kc[8 * i + 6] = kc[8 * i + 7] = 0.0;
                }
                dma(lay.kcam, stage + lay.kcam, sizeof(double) * 8 * (size_t)np);
            }
            if (err == hipSuccess) err = hipEventRecord(hh->copy_event, hh->copy_stream);
            out->copy_err = err;
        });
        ho.idx_ready.store(1, std::memory_order_release);
        return;
    }
    // (test hook helper_delay_us: the helper starts that much later: whatever this thread takes from the helper
    //  without waiting for it shows up as a wrong result instead of hiding behind the usual timing)
    const int helper_delay_us = HOOK(h, helper_delay_us);
    // (by value: the layout, the buffers and the handle's streams; by reference: the hand-off object alone)
    const EdgeLayout lay = L;
    char *const stage = sg;
    movba_handle *const hh = h;
    const movba_lba_desc *const dd = d;
    const int np = NP, p = P, e = E;
    HelperHandOff *const out = &ho;
    const bool send_idx = dev_first;        // (device grouping pass: the index arrays cross first, on the copy stream, behind edgeb_event)
    h->packer.post([=]() {
        if (helper_delay_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(helper_delay_us));
        // (the index arrays as they are: right when the edges come grouped by point, overwritten by pack_a otherwise)
        std::memcpy(stage + lay.gpose, dd->edge_pose, sizeof(int32_t) * (size_t)e);
        std::memcpy(stage + lay.gpoint, dd->edge_point, sizeof(int32_t) * (size_t)e);
        if (send_idx) {
            hipError_t e0 = hipSetDevice(hh->device);
            if (e0 == hipSuccess) e0 = hipMemcpyAsync(out->arena + lay.gpose, stage + lay.gpose, sizeof(int32_t) * (size_t)e, hipMemcpyHostToDevice, hh->copy_stream);
            if (e0 == hipSuccess) e0 = hipMemcpyAsync(out->arena + lay.gpoint, stage + lay.gpoint, sizeof(int32_t) * (size_t)e, hipMemcpyHostToDevice, hh->copy_stream);
            if (e0 == hipSuccess) e0 = hipEventRecord(hh->edgeb_event, hh->copy_stream);
            out->idx_err = e0;
        }
        out->idx_ready.store(1, std::memory_order_release);
        // ... each part straight on to the device on the copy stream while the next one is being staged: most of the upload
        // is across the bus before the calling thread has finished its pass over the edges (the solve's first kernels wait
        // for copy_event, nothing else does)
        hipError_t err = hipSetDevice(hh->device);
        auto send = [&](size_t from, size_t to) {
            if (err == hipSuccess && to > from)
                err = hipMemcpyAsync(out->arena + from, stage + from, to - from, hipMemcpyHostToDevice, hh->copy_stream);
        };
        std::memcpy(stage + lay.obs, dd->obs, sizeof(double) * 2 * (size_t)e);
        send(lay.obs, lay.isig);
        std::memcpy(stage + lay.isig, dd->inv_sigma2, sizeof(double) * (size_t)e);
        if (dd->obs_right) std::memcpy(stage + lay.obsr, dd->obs_right, sizeof(double) * (size_t)e);
        std::memcpy(stage + lay.pose0, dd->poses, sizeof(double) * 7 * (size_t)np);
        std::memcpy(stage + lay.point0, dd->points, sizeof(double) * 3 * (size_t)p);
        if (lay.has_kcam) {
            double *kc = reinterpret_cast<double *>(stage + lay.kcam);
            for (int i = 0; i < np; ++i) {
                const double *ck = dd->cam_kf ? dd->cam_kf + 4 * (size_t)i : &dd->fx;      // (fx, fy, cx, cy are contiguous in the descriptor)
                kc[8 * i] = ck[0]; kc[8 * i + 1] = ck[1]; kc[8 * i + 2] = ck[2]; kc[8 * i + 3] = ck[3];
                kc[8 * i + 4] = dd->bf_kf ? dd->bf_kf[i] : dd->bf; kc[8 * i + 5] = kc[8 * i + 6] = kc[8 * i + 7] = 0.0;
            }
        }
        send(lay.isig, lay.grouped_end);
        if (err == hipSuccess) err = hipEventRecord(hh->copy_event, hh->copy_stream);
        out->copy_err = err;
    });
}

int Upload::group()
{
    // (the pose-major slots are left to the device when the edges come grouped by point: the pass leaves each edge's rank
    // among its keyframe's edges where the slots go)
    h->st.no_reorder = h->opt.reorder == -1;
    const int rc = build_basic(*d, h->st, reinterpret_cast<int32_t *>(sg + L.slot));
    rank_mode = true;
    lap("build_basic");
    if (rc < 0) { (void)join_helper(); (void)hipStreamSynchronize(h->copy_stream); return rc; }
    h->stop = d->stop;
    if (rc == MOVBA_EMPTY || s().P == 0) { h->early_status = MOVBA_EMPTY; }
    else if (s().n_fixed == 0) { h->early_status = MOVBA_NO_FIXED; }
    if (h->early_status != MOVBA_OK) {
        const int rw = join_helper(); if (rw) return rw;
        h->prof.structure_ms += now_ms() - t0; h->uploaded = true; done = true; return MOVBA_OK;
    }
    nf = s().nfree;
    // beyond the one-launch direct solver (dense_plan.h) the multi-launch one holds the solution vector in LDS and the lower
    // block triangle in HBM: refused by name past that, instead of failing in a launch
    if (nf > MOVBA_MAX_FREE_KEYFRAMES) {
        (void)join_helper();
        std::fprintf(stderr, "libmovba: %d free keyframes: the reduced system exceeds the direct solver's capacity (%d)\n", nf, MOVBA_MAX_FREE_KEYFRAMES);
        return MOVBA_ERR_TOO_LARGE;
    }
    nb = (P + kPointsPerBlock - 1) / kPointsPerBlock;
    edge_bytes = s().already_grouped ? L.grouped_end : L.max_end;
    nbins = nf * nf;
    return MOVBA_OK;
}

// States, per-edge records, cost partials, controller, results: sizes from NP, P, E (and the free keyframes' number) alone, so
// that a device grouping pass can carve them BEFORE the pair structure exists and start the solve's first kernels on them.
// (pose-major arrays are sized for E edges of free keyframes, their upper bound)
void Upload::carve_state()
{
    for (int b = 0; b < 2; ++b) {
        o_st[b][0] = c.take<double>(7 * (size_t)NP); o_st[b][1] = c.take<double>(12 * (size_t)NP);
        o_st[b][2] = c.take<double>(3 * (size_t)P);  o_st[b][3] = c.take<double>(6 * (size_t)P);
        o_st[b][4] = c.take<double>(3 * (size_t)P);  o_st[b][5] = c.take<double>(4 * (size_t)E);       // erecA
        o_st[b][6] = 0;  o_st[b][7] = 0;
        o_st[b][8] = c.take<double>(nb);
        o_st[b][9] = 0;
        o_st[b][10] = 0;
    }
    o_obspm = c.take<double>(2 * (size_t)E + 2); o_obsrpm = c.take<double>(stereo ? (size_t)E + 1 : 1);
    o_aci = c.take<double>(3 * kCoarseDim * kCoarseDim + 2); o_acitag = c.take<int32_t>(2);
    o_bp = c.take<double>(6 * (size_t)nf + 1); o_xp = c.take<double>(6 * (size_t)nf + 1);
    o_scale = c.take<double>(nb + 1); o_hmax = c.take<double>(nb); o_tick = c.take<uint32_t>(8 * (size_t)nb + 8);
    o_ctrl = c.take<Ctrl>(1); o_chi2 = c.take<double>(E); o_outl = c.take<uint8_t>(E);
    state_carved = true;
}

// The grouping pass on the device: wanted where the device also builds the pair structure through its pair-bin masks (up to 80
// free keyframes: every window MoV-SLAM's local mapping produces) and where the host can tell from the keyframes' flags
// alone how many free keyframes there are.
bool Upload::dev_first_eligible() const
{
    if (HOOK(h, host_structure) || HOOK(h, host_grouping)) return false;
    if (E < kDeviceStructureMinEdges) return false;
    if (E <= 0 || P <= 0 || NP <= 0 || NP > 1024) return false;
    int nfix = 0;
    for (int i = 0; i < NP; ++i) nfix += d->pose_fixed[i] != 0;
    const int nfm = NP - nfix;
    return nfix > 0 && nfm > 0 && nfm <= 80 && struct_lds_fits(nfm, NP);
}

// What build_basic derives in one pass over the caller's edges on this thread (0.11 ms at cfg3, a tenth of the whole call, with
// the device idle but for the copies) is the device's work here: it validates the index arrays, finds the points' ranges,
// counts the edges per keyframe, numbers the free keyframes and ranks every edge among its keyframe's edges (k_basic_hist,
// k_basic_index), then counts the pair bins as before; this thread waits ONCE, for the pair counts and the edges per keyframe
// together, and rebuilds its own small tables (hessian indices, free poses, first slots) from the latter.  A window the pass
// cannot take as it is - edges not grouped by point, a free keyframe nobody observes - is handed back to the host pass.
constexpr int kRetryClassic = 1 << 20;
int Upload::group_on_device()
{
    Structure &st = h->st;
    reset_structure(st, NP, P, E);
    st.no_reorder = h->opt.reorder == -1;
    for (nf_expect = 0, nf = 0; nf < NP; ++nf) nf_expect += d->pose_fixed[nf] == 0;
    nf = nf_expect; nbins = nf * nf;
    nb = (P + kPointsPerBlock - 1) / kPointsPerBlock;
    edge_bytes = L.grouped_end;
    carve_state();
    int rc = carve_scratch(true); if (rc) return rc;
    char *sa = h->scratch, *misc = sg + h->stage_cap - misc_bytes;
    // (the keyframes' flags are read out of host memory - the staging buffer is mapped -, through the place of the hessian
    //  indices, which the device makes itself here)
    std::memcpy(sg + L.hidx, d->pose_fixed, (size_t)NP);
    // bin totals, error word, edges per keyframe, info words: cleared by the ingest launch itself (direct mode)
    const size_t zero_bytes = so_info + sizeof(int32_t) * kBasicInfo - so_cnt;
    if (!direct_raw) HIP_TRY(hipMemsetAsync(sa + so_cnt, 0, zero_bytes, h->stream));
    bd = BasicDev{};
    bd.E = E; bd.P = P; bd.NP = NP; bd.nblk = (E + kBasicBlock - 1) / kBasicBlock;
    if (direct_raw) {
        // The caller's index arrays out of its own pinned memory, by kernel (struct_kernels.hip: k_ingest), on this stream, in
        // front of the grouping kernel that reads them.
        auto view = [](const void *p, size_t bytes) { return static_cast<const void *>(host_block_view(p, bytes)); };
        IngestArgs ia{};
        ia.seg[0] = IngestSeg{ view(d->edge_pose, sizeof(int32_t) * (size_t)E), h->arena + L.gpose, sizeof(int32_t) * (size_t)E };
        ia.seg[1] = IngestSeg{ view(d->edge_point, sizeof(int32_t) * (size_t)E), h->arena + L.gpoint, sizeof(int32_t) * (size_t)E };
        // (the keyframes' flags with them: four bytes at a time out of the staging buffer's copy, which is padded)
        ia.seg[2] = IngestSeg{ h->stage_dev + L.hidx, sa + so_fixed, ((size_t)NP + 3) & ~(size_t)3 };
        ia.nseg = 3; ia.counter = h->ingest_counter; ia.wait_for = 0;
        ia.zero = reinterpret_cast<unsigned *>(sa + so_cnt); ia.zero_words = (unsigned)(zero_bytes / 4);
        HIP_TRY(launch_ingest(ia, h->stream));
        h->ingest_expect += (unsigned)ingest_workgroups();

    } else {
        // the index arrays are on their way on the copy stream (post_helper): the grouping kernel starts behind their event
        while (ho.idx_ready.load(std::memory_order_acquire) == 0) host_relax(h->opt.host_wait);
        if (ho.idx_err != hipSuccess) { std::fprintf(stderr, "libmovba: upload copy failed: %s\n", hipGetErrorString(ho.idx_err)); return MOVBA_ERR_HIP; }
        HIP_TRY(hipStreamWaitEvent(h->stream, h->edgeb_event, 0));
    }
    arena_gen_at_edge_copy = ho.arena_gen;
    bd.edge_pose = reinterpret_cast<const int32_t *>(h->arena + L.gpose); bd.edge_point = reinterpret_cast<const int32_t *>(h->arena + L.gpoint);
    // (staged mode: the flags are read out of the mapped staging buffer; direct mode: k_ingest has brought them along)
    bd.pose_fixed = direct_raw ? reinterpret_cast<const uint8_t *>(sa + so_fixed) : reinterpret_cast<const uint8_t *>(h->stage_dev + L.hidx);
    bd.pt_start = reinterpret_cast<int32_t *>(h->arena + L.ptstart); bd.rank = reinterpret_cast<int32_t *>(h->arena + L.slot);
    bd.H = reinterpret_cast<int32_t *>(sa + so_H); bd.pose_edges = reinterpret_cast<int32_t *>(sa + so_pe);
    bd.hidx = reinterpret_cast<int32_t *>(h->arena + L.hidx); bd.base = reinterpret_cast<int32_t *>(h->arena + L.base);
    bd.free_pose = reinterpret_cast<int32_t *>(h->arena + L.free_pose); bd.info = reinterpret_cast<int32_t *>(sa + so_info);
    HIP_TRY(launch_basic(bd, h->stream));
    rc = launch_counts(); if (rc) return rc;
    lap("grouping + count launches");
    {
        const double t_wait = now_ms();
        while (__atomic_load_n(misc_seq, __ATOMIC_ACQUIRE) != seq) {
            host_relax(h->opt.host_wait);
            if (now_ms() - t_wait > 10000.0) { HIP_TRY(hipStreamSynchronize(h->stream)); if (__atomic_load_n(misc_seq, __ATOMIC_ACQUIRE) != seq) return MOVBA_ERR_HIP; }
        }
    }
    const int32_t *info = reinterpret_cast<const int32_t *>(misc) + nbins + 2, *pe = info + kBasicInfo;
    if (info[0]) { (void)join_helper(); (void)hipStreamSynchronize(h->copy_stream); return MOVBA_ERR_ARG; }      // an index out of range
    if (info[1] || info[2] != nf_expect) return kRetryClassic;
    st.pose_edges.assign(pe, pe + NP); st.pose_edges.push_back(0);
    index_poses(d->pose_fixed, st);
    if (st.nfree != nf_expect || st.E_free != info[3]) return MOVBA_ERR_HIP;      // (the device and this thread number the same keyframes)
    st.already_grouped = true; st.perm.clear(); st.gp = d->edge_pose; st.gl = d->edge_point;
    st.pt_start.clear();        // (the points' ranges exist on the device only)
    rank_mode = true;
    h->stop = d->stop;
    lap("wait for the grouping pass and the pair counts");
    return MOVBA_OK;
}

void Upload::pack_a(bool raw_too)
{
    if (!s().already_grouped || raw_too) {         // (grouped order: the helper thread copied the caller's index arrays)
        put(sg + L.gpose, s().gp, sizeof(int32_t) * E);
        put(sg + L.gpoint, s().gl, sizeof(int32_t) * E);
    }
    put(sg + L.ptstart, s().pt_start.data(), sizeof(int32_t) * (P + 1));
    put(sg + L.hidx, s().hidx.data(), sizeof(int32_t) * NP);
}

void Upload::pack_b(bool raw_too)
{
    if (!s().already_grouped) put(sg + L.perm, s().perm.data(), sizeof(int32_t) * E);
    if (!rank_mode) put(sg + L.slot, s().slot.data(), sizeof(int32_t) * E);
    else put(sg + L.base, s().pose_slot0.data(), sizeof(int32_t) * NP);
    put(sg + L.free_pose, s().free_pose.data(), sizeof(int32_t) * nf);
    double *obs = reinterpret_cast<double *>(sg + L.obs), *isg = reinterpret_cast<double *>(sg + L.isig);
    double *obr = reinterpret_cast<double *>(sg + L.obsr);
    if (!s().already_grouped) {       // the helper's straight copies are in caller order: permute into grouped order
        for (int g = 0; g < E; ++g) {
            const int e = s().perm[g];
            obs[2 * g] = d->obs[2 * e]; obs[2 * g + 1] = d->obs[2 * e + 1]; isg[g] = d->inv_sigma2[e];
        }
        if (d->obs_right) for (int g = 0; g < E; ++g) obr[g] = d->obs_right[s().perm[g]];
    } else if (raw_too) {
        put(obs, d->obs, sizeof(double) * 2 * (size_t)E);
        put(isg, d->inv_sigma2, sizeof(double) * (size_t)E);
        if (d->obs_right) put(obr, d->obs_right, sizeof(double) * (size_t)E);
    }
    if (raw_too) {
        put(sg + L.pose0, d->poses, sizeof(double) * 7 * (size_t)NP);
        put(sg + L.point0, d->points, sizeof(double) * 3 * (size_t)P);
    }
}

int Upload::pack_derived()
{
    if (!s().already_grouped) {
        // (rare: the helper's straight copies get permuted below, so it has to be through with them)
        const int rw = join_helper(); if (rw) return rw;
        // ... and so do its transfers out of the staging buffer (found by ThreadSanitizer over the fake device, tests/hipstub: the
        // copy engine was still reading the caller-order observations while they were being permuted; harmless for the result —
        // the permuted region is sent again behind that copy — but a torn first copy is nothing to rely on)
        HIP_TRY(hipEventSynchronize(h->copy_event));
        build_slots(h->st); rank_mode = false;
        pack_edges(false);
    } else {
        while (ho.idx_ready.load(std::memory_order_acquire) == 0) host_relax(h->opt.host_wait);
        pack_a(false);
        pack_b(false);      // (ranks where the slots go, the keyframes' first slots, the free keyframes)
    }
    lap("pack derived arrays");
    return MOVBA_OK;
}

int Upload::send_edge_a()
{
    const double t_up0 = now_ms();
    HIP_TRY(hipMemcpyAsync(h->arena, sg, L.a_end, hipMemcpyHostToDevice, h->stream));
    arena_gen_at_edge_copy = ho.arena_gen;
    // grouped edges: the rest of the derived arrays (point ids, ranks / slots, first slots) leaves at once on the copy
    // stream, beside the structure kernels of this stream; what needs it (slot completion, fill) waits for edgeb_event
    // (not where the host goes on to build the pair structure itself - windows below kDeviceStructureMinEdges -: it packs its
    //  slots over the ranks in the staging buffer, which this copy would still be reading; that part then travels once, later)
    if (s().already_grouped && h->arena_gen == ho.arena_gen && E >= kDeviceStructureMinEdges && !HOOK(h, host_structure)) {
        HIP_TRY(hipMemcpyAsync(h->arena + L.a_end, sg + L.a_end, L.raw_begin - L.a_end, hipMemcpyHostToDevice, h->copy_stream));
        HIP_TRY(hipEventRecord(h->edgeb_event, h->copy_stream));
        edge_b_early = true;
    }
    upload_host_ms = now_ms() - t_up0;
    return MOVBA_OK;
}

int Upload::queue_edge_b()
{
    if (dev_first) { edge_b_queued = true; return MOVBA_OK; }       // (point ids, ranks, first slots: all made on the device)
    if (edge_b_early) HIP_TRY(hipStreamWaitEvent(h->stream, h->edgeb_event, 0));
    if (!edge_b_early || edge_b_stale) {
        // (not sent yet, or packed again since: host-built slots instead of ranks)
        HIP_TRY(hipMemcpyAsync(h->arena + L.a_end, sg + L.a_end, L.raw_begin - L.a_end, hipMemcpyHostToDevice, h->stream));
    }
    if (!s().already_grouped) {
        // the helper's straight copies were permuted again by pack_edges: that part travels once more, behind the first copy
        { const int rw = join_helper(); if (rw) return rw; }      // (its copy_event must have been recorded)
        HIP_TRY(hipStreamWaitEvent(h->stream, h->copy_event, 0));
        HIP_TRY(hipMemcpyAsync(h->arena + L.raw_begin, sg + L.raw_begin, edge_bytes - L.raw_begin, hipMemcpyHostToDevice, h->stream));
    }
    edge_b_queued = true;
    return MOVBA_OK;
}

int Upload::launch_fill()
{
    // (the scan of the per-chunk counts the fill reads, where launch_counts left it to whoever launches the fill)
    if (scan_pending) { HIP_TRY(launch_struct_scan(sd, h->stream)); scan_pending = false; }
    if (sorted_structure) {
        sd.ent64 = reinterpret_cast<unsigned long long *>(h->arena + o_ent);
        sd.g_pose = reinterpret_cast<int32_t *>(h->arena + L.gpose); sd.pt_start = reinterpret_cast<int32_t *>(h->arena + L.ptstart);
        sd.hidx = reinterpret_cast<int32_t *>(h->arena + L.hidx);
        sd.slot = reinterpret_cast<const int32_t *>(h->arena + L.slot);
        char *s2 = h->scratch2;
        HIP_TRY(launch_sorted_fill(sd, reinterpret_cast<const int32_t *>(h->scratch + so_cntpt), reinterpret_cast<int32_t *>(s2 + s2_off),
                                   reinterpret_cast<unsigned *>(s2 + s2_keys_in), reinterpret_cast<unsigned *>(s2 + s2_keys_out),
                                   reinterpret_cast<unsigned long long *>(s2 + s2_vals_in), s2 + s2_tmp, s2_tmp_bytes, (long long)noff, h->stream));
        return MOVBA_OK;
    }
    int32_t *ed = reinterpret_cast<int32_t *>(h->arena + o_ent);
    sd.ent_i = ed; sd.ent_j = ed + noff; sd.ent_l = ed + 2 * noff;
    sd.ent64 = ent_packed ? reinterpret_cast<unsigned long long *>(h->arena + o_ent) : nullptr;
    sd.g_pose = reinterpret_cast<int32_t *>(h->arena + L.gpose); sd.pt_start = reinterpret_cast<int32_t *>(h->arena + L.ptstart);
    sd.hidx = reinterpret_cast<int32_t *>(h->arena + L.hidx);
    sd.slot = reinterpret_cast<const int32_t *>(h->arena + L.slot);
    HIP_TRY(launch_struct_fill(sd, h->stream));
    return MOVBA_OK;
}

int Upload::launch_slotpt()
{
    HIP_TRY(launch_slot_point(reinterpret_cast<int32_t *>(h->arena + L.slot), reinterpret_cast<const int32_t *>(h->arena + L.gpose),
                              rank_mode ? reinterpret_cast<const int32_t *>(h->arena + L.base) : nullptr,
                              reinterpret_cast<const int32_t *>(h->arena + L.gpoint), reinterpret_cast<int32_t *>(h->arena + o_slotpt), E,
                              dev_first ? reinterpret_cast<const int32_t *>(h->scratch + so_H) : nullptr, NP, h->stream));
    return MOVBA_OK;
}

// the per-pair entry lists built on the host (ungrouped edges, more than 80 free keyframes, or a pair-bin mask beyond LDS)
int Upload::structure_on_host()
{
    const int rc = build_structure(*d, h->st);         // (runs build_basic again, with the slots this time)
    if (rc < 0) return rc;
    rank_mode = false;
    pack_b(false); edge_b_stale = true;      // (slots instead of ranks in the staging buffer now)
    noff = (size_t)(s().nentries - s().E_free);
    o_slotpt = c.take<int32_t>((size_t)s().E_free + 1);
    return MOVBA_OK;
}

// scratch of the device structure pass (and of the device grouping pass in front of it)
int Upload::carve_scratch(bool basic)
{
    const int nchunks = (P + 63) / 64;
    Carver sc;
    so_cnt = sc.take<int32_t>(nbins); so_err = sc.take<int32_t>(4);
    so_pe = sc.take<int32_t>(basic ? NP : 0); so_info = sc.take<int32_t>(basic ? kBasicInfo : 0);     // (zeroed together with the two above)
    so_ent0 = sc.take<int32_t>(nbins);
    so_cntw = sc.take<int32_t>((size_t)nbins * nchunks);
    so_fixed = sc.take<uint8_t>(basic ? (size_t)NP + 4 : 0);
    so_H = sc.take<int32_t>(basic ? (size_t)((E + kBasicBlock - 1) / kBasicBlock) * NP : 0);
    if (sc.off > h->scratch_cap) {
        if (h->scratch) { HIP_TRY(hipStreamSynchronize(h->stream)); HIP_TRY(hipFree(h->scratch)); h->scratch = nullptr; h->scratch_cap = 0; }
        const size_t cap = align_up(sc.off + sc.off / 4, 1 << 20);
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->scratch), cap));
        h->scratch_cap = cap;
    }
    return MOVBA_OK;
}

// count launches of the device structure pass; the bins' totals (and, after a device grouping pass, its results) come back
// through the tail of the staging buffer, behind a sequence number
int Upload::launch_counts()
{
    const int nchunks = (P + 63) / 64;
    char *sa = h->scratch, *misc = sg + h->stage_cap - misc_bytes;    // tail of the staging buffer: the pair region is packed in front of it
    sd.P = P; sd.nfree = nf; sd.nchunks = nchunks; sd.NP = NP;
    // grouped edges, point ranges and hessian indices are read where the edge copy (or the device grouping pass) put them
    sd.g_pose = reinterpret_cast<int32_t *>(h->arena + L.gpose); sd.pt_start = reinterpret_cast<int32_t *>(h->arena + L.ptstart);
    sd.hidx = reinterpret_cast<int32_t *>(h->arena + L.hidx);
    sd.cntw = reinterpret_cast<int32_t *>(sa + so_cntw); sd.cnt = reinterpret_cast<int32_t *>(sa + so_cnt);
    sd.error = reinterpret_cast<int32_t *>(sa + so_err);
    sd.ent0 = reinterpret_cast<int32_t *>(sa + so_ent0);
    sd.abort = dev_first ? reinterpret_cast<const int32_t *>(sa + so_info) : nullptr;
    HIP_TRY(launch_struct_count(sd, h->stream));
    // (misc: nbins totals, the error word, the sequence number of this upload[, kBasicInfo words, NP edges per keyframe])
    misc_seq = reinterpret_cast<volatile int32_t *>(misc) + nbins + 1;
    seq = (int32_t)(++h->count_seq & 0x7fffffff);
    __atomic_store_n(misc_seq, seq - 1, __ATOMIC_RELAXED);
    HIP_TRY(launch_struct_counts_out(sd, reinterpret_cast<int32_t *>(h->stage_dev + (misc - sg)), seq, h->stream,
                                     dev_first ? reinterpret_cast<const int32_t *>(sa + so_pe) : nullptr, dev_first ? reinterpret_cast<const int32_t *>(sa + so_info) : nullptr));
    // (the scan over the chunks is what the FILL needs, not the host: in direct mode the helper thread launches it with the fill)
    scan_pending = dev_first && direct_raw && h->opt.profile == 0;
    if (!scan_pending) HIP_TRY(launch_struct_scan(sd, h->stream));
    return MOVBA_OK;
}

// ... counted and filled on the GPU (struct_kernels.hip): the reference's own edge order, up to 80 free keyframes
int Upload::structure_on_device()
{
    int rc = carve_scratch(false); if (rc) return rc;
    HIP_TRY(hipMemsetAsync(h->scratch + so_cnt, 0, so_err + 16 - so_cnt, h->stream));        // bin totals and the error word
    rc = launch_counts(); if (rc) return rc;
    lap("edge H2D + count launches");
    return after_counts();
}

int Upload::after_counts()
{
    char *sa = h->scratch, *misc = sg + h->stage_cap - misc_bytes;
    if (!dev_first) {
        const double t_wait = now_ms();
        while (__atomic_load_n(misc_seq, __ATOMIC_ACQUIRE) != seq) {
            host_relax(h->opt.host_wait);
            if (now_ms() - t_wait > 10000.0) { HIP_TRY(hipStreamSynchronize(h->stream)); if (__atomic_load_n(misc_seq, __ATOMIC_ACQUIRE) != seq) return MOVBA_ERR_HIP; }
        }
    }
    if (reinterpret_cast<const int32_t *>(misc)[nbins] != 0) return MOVBA_ERR_ARG;     // duplicate observation
    lap("wait for the pair counts");
    // covisibility ordering (structure.h): a window whose keyframe ids do not follow its covisibility graph is renumbered
    // here, from the counts: hessian indices, free-pose list and first slots are sent again (a few hundred bytes) and the
    // count / scan kernels run once more in the new numbering (the host permutes its copy of the counts itself)
    if (!h->st.no_reorder) {
        std::vector<int32_t> new_of_old;
        if (covisibility_order(nf, reinterpret_cast<const int32_t *>(misc), new_of_old)) {
            apply_pose_order(h->st, new_of_old, reinterpret_cast<int32_t *>(misc));
            h->st.reordered = true;
            std::memcpy(sg + L.hidx, s().hidx.data(), sizeof(int32_t) * NP);
            std::memcpy(sg + L.base, s().pose_slot0.data(), sizeof(int32_t) * NP);
            std::memcpy(sg + L.free_pose, s().free_pose.data(), sizeof(int32_t) * nf);
            HIP_TRY(hipMemcpyAsync(h->arena + L.hidx, sg + L.hidx, sizeof(int32_t) * NP, hipMemcpyHostToDevice, h->stream));
            if (dev_first) {
                // (the device made these itself in the caller's numbering: only the three renumbered tables travel)
                HIP_TRY(hipMemcpyAsync(h->arena + L.base, sg + L.base, sizeof(int32_t) * NP, hipMemcpyHostToDevice, h->stream));
                HIP_TRY(hipMemcpyAsync(h->arena + L.free_pose, sg + L.free_pose, sizeof(int32_t) * nf, hipMemcpyHostToDevice, h->stream));
            } else edge_b_stale = true;
            HIP_TRY(hipMemsetAsync(sa + so_cnt, 0, so_err + 16 - so_cnt, h->stream));
            HIP_TRY(launch_struct_count(sd, h->stream));
            HIP_TRY(launch_struct_counts_out(sd, nullptr, 0, h->stream));
            HIP_TRY(launch_struct_scan(sd, h->stream));
            scan_pending = false;
            lap("covisibility reorder + recount");
        }
    }
    // slots, point ids, observations and initial estimates cross the bus, then the entry lists are filled, while the
    // host lays out the pairs
    { const int rq = queue_edge_b(); if (rq) return rq; }
    {
        const int32_t *cnt = reinterpret_cast<const int32_t *>(misc);
        int64_t n = 0;
        for (int i = 0; i < nf; ++i) for (int j = i + 1; j < nf; ++j) n += cnt[(size_t)i * nf + j];
        if (n > (int64_t)0x7fffffff / 4) return MOVBA_ERR_ARG;
        noff = (size_t)n;
    }
    o_ent = c.take<int32_t>(ent_words());
    o_slotpt = c.take<int32_t>((size_t)s().E_free + 1);
    if (c.off <= h->arena_cap && h->arena_gen == ho.arena_gen) {
        filled_early = true; fill_gen = h->arena_gen;
        if (dev_first && direct_raw && h->opt.profile == 0) {
            // The slots' completion, the fill of the entry lists and the solve's first two kernels - state 0 from the uploaded
            // estimates, the first linearisation: they need the edge data, the slots and the state arrays, none of which depends
            // on the pair structure - are queued by the HELPER thread (idle in direct mode) while this thread lays out the pairs:
            // four launches and an event wait are ~25 us of API calls that would otherwise stand in front of finish_pairs, and
            // the kernels run in the shadow of the pair layout; movba_lba_run then starts with the Hpp pass.  Whatever order the
            // two threads' commands reach the stream in, each of this thread's (the pair region's copy) is independent of the
            // helper's; send_pairs joins the helper before the run can queue anything behind them.
            { const int rw = join_helper(); if (rw) return rw; }      // (its DMA commands are queued: ~30 us into the call)
            device_view();
            const DevWindow wv = h->win;
            movba_handle *const hh = h;
            Upload *const self = this;
            HelperHandOff *const out = &ho;
            const bool laps = lap_on;
            const double tz = t0;
            h->packer.post([=]() {
                double tl[6]; tl[0] = now_ms();
                hipError_t err = hipSetDevice(hh->device);
                int rq = MOVBA_OK;
                if (err == hipSuccess) rq = self->launch_slotpt();
                tl[1] = now_ms();
                if (err == hipSuccess && rq == MOVBA_OK) rq = self->launch_fill();
                tl[2] = now_ms();
                if (err == hipSuccess && rq == MOVBA_OK) err = hipStreamWaitEvent(hh->stream, hh->copy_event, 0);
                tl[3] = now_ms();
                if (err == hipSuccess && rq == MOVBA_OK) err = launch_init(wv, hh->stream);
                tl[4] = now_ms();
                if (err == hipSuccess && rq == MOVBA_OK) err = launch_linearize(wv, hh->stream);
                tl[5] = now_ms();
                out->copy_err = (err == hipSuccess && rq != MOVBA_OK) ? hipErrorUnknown : err;
                if (laps) std::fprintf(stderr, "libmovba[upload]: helper's launches (ms into the call): start %.3f, slots %.3f, scan + fill %.3f, event wait %.3f, init %.3f, linearise %.3f\n",
                                       tl[0] - tz, tl[1] - tz, tl[2] - tz, tl[3] - tz, tl[4] - tz, tl[5] - tz);
            });
            ho.joined = false;                              // (send_pairs waits for it)
            h->early_setup = true;
        } else {
            int rq = launch_slotpt(); if (rq) return rq;         // (completes the slots the fill reads)
            rq = launch_fill(); if (rq) return rq;
        }
    }
    lap("edge B H2D + fill kernel (queued)");
    const int rc = finish_pairs(h->st, reinterpret_cast<const int32_t *>(misc));
    lap("finish_pairs");
    if (rc < 0) return rc;
    if ((size_t)(s().nentries - s().E_free) != noff) return MOVBA_ERR_ARG;
    return MOVBA_OK;
}

// ... beyond k_struct_pairs' reach (more than 80 free keyframes, or a pair-bin mask that does not fit LDS): counted by atomics
// and filled by a stable sort of the points' couples (struct_sort.hip); the same hand-offs with the host as above
int Upload::structure_on_device_sorted()
{
    Carver sc;
    const size_t so_cnt = sc.take<int32_t>(nbins), so_err = sc.take<int32_t>(4);
    const size_t so_ent0 = sc.take<int32_t>(nbins);
    so_cntpt = sc.take<int32_t>((size_t)P + 1);
    if (sc.off > h->scratch_cap) {
        if (h->scratch) { HIP_TRY(hipStreamSynchronize(h->stream)); HIP_TRY(hipFree(h->scratch)); h->scratch = nullptr; h->scratch_cap = 0; }
        const size_t cap = align_up(sc.off + sc.off / 4, 1 << 20);
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->scratch), cap));
        h->scratch_cap = cap;
    }
    char *sa = h->scratch, *misc = sg + h->stage_cap - misc_bytes;
    HIP_TRY(hipMemsetAsync(sa + so_cnt, 0, so_err + 16 - so_cnt, h->stream));        // bin totals and the error word
    sd = StructDev{};
    sd.P = s().P; sd.nfree = nf; sd.nchunks = 0; sd.NP = NP;
    sd.g_pose = reinterpret_cast<int32_t *>(h->arena + L.gpose); sd.pt_start = reinterpret_cast<int32_t *>(h->arena + L.ptstart);
    sd.hidx = reinterpret_cast<int32_t *>(h->arena + L.hidx);
    sd.cnt = reinterpret_cast<int32_t *>(sa + so_cnt); sd.error = reinterpret_cast<int32_t *>(sa + so_err);
    sd.ent0 = reinterpret_cast<int32_t *>(sa + so_ent0);
    HIP_TRY(launch_couple_count(sd, reinterpret_cast<int32_t *>(sa + so_cntpt), h->stream));
    volatile int32_t *misc_seq = reinterpret_cast<volatile int32_t *>(misc) + nbins + 1;
    const int32_t seq = (int32_t)(++h->count_seq & 0x7fffffff);
    __atomic_store_n(misc_seq, seq - 1, __ATOMIC_RELAXED);
    HIP_TRY(launch_struct_counts_out(sd, reinterpret_cast<int32_t *>(h->stage_dev + (misc - sg)), seq, h->stream));
    lap("edge H2D + count launches");
    {
        const double t_wait = now_ms();
        while (__atomic_load_n(misc_seq, __ATOMIC_ACQUIRE) != seq) {
            host_relax(h->opt.host_wait);
            if (now_ms() - t_wait > 10000.0) { HIP_TRY(hipStreamSynchronize(h->stream)); if (__atomic_load_n(misc_seq, __ATOMIC_ACQUIRE) != seq) return MOVBA_ERR_HIP; }
        }
    }
    if (reinterpret_cast<const int32_t *>(misc)[nbins] != 0) return MOVBA_ERR_ARG;     // duplicate observation
    lap("wait for the pair counts");
    { const int rq = queue_edge_b(); if (rq) return rq; }
    {
        const int32_t *cnt = reinterpret_cast<const int32_t *>(misc);
        int64_t n = 0;
        for (int i = 0; i < nf; ++i) for (int j = i + 1; j < nf; ++j) n += cnt[(size_t)i * nf + j];
        if (n > (int64_t)0x7fffffff / 4) return MOVBA_ERR_ARG;
        noff = (size_t)n;
    }
    o_ent = c.take<int32_t>(ent_words());
    o_slotpt = c.take<int32_t>((size_t)s().E_free + 1);
    // keys, values and rocPRIM's temporary storage of the fill
    {
        Carver s2;
        s2_off = s2.take<int32_t>((size_t)P + 1);
        s2_keys_in = s2.take<unsigned>(noff + 1); s2_keys_out = s2.take<unsigned>(noff + 1);
        s2_vals_in = s2.take<unsigned long long>(noff + 1);
        s2_tmp_bytes = sorted_fill_temp_bytes(P, (long long)noff, nf);
        s2_tmp = s2.take<char>(s2_tmp_bytes);
        if (s2.off > h->scratch2_cap) {
            if (h->scratch2) { HIP_TRY(hipStreamSynchronize(h->stream)); HIP_TRY(hipFree(h->scratch2)); h->scratch2 = nullptr; h->scratch2_cap = 0; }
            const size_t cap = align_up(s2.off + s2.off / 4, 1 << 20);
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->scratch2), cap));
            h->scratch2_cap = cap;
        }
    }
    sorted_structure = true;
    if (c.off <= h->arena_cap && h->arena_gen == ho.arena_gen) {
        int rq = launch_slotpt(); if (rq) return rq;         // (completes the slots the fill reads)
        rq = launch_fill(); if (rq) return rq;
        filled_early = true; fill_gen = h->arena_gen;
    }
    lap("edge B H2D + fill kernels (queued)");
    const int rc = finish_pairs(h->st, reinterpret_cast<const int32_t *>(misc));
    lap("finish_pairs");
    if (rc < 0) return rc;
    if ((size_t)(s().nentries - s().E_free) != noff) return MOVBA_ERR_ARG;
    return MOVBA_OK;
}

void Upload::choose_solver()
{
    h->pp = PcgParams{};
    h->rows_kernel = pcg_rows_supported(s().nfree, s().row_ptr.data(), &h->pp);
    // (test switch: the packed layout of the mat-vec's pair sums where the padded one would do - same bits, tests/test_gpu_parity.py)
    if (HOOK(h, pcg_packed)) h->pp.padded = 0;
    // A reduced matrix beyond the PCG workgroup's registers (dense covisibility: every keyframe pair shares points, as in the
    // reference's own windows, KeyFrame.cc:227-231; or simply more keyframes) is not iterated over from L2: the one-launch
    // direct solver takes the window from the first trial, whatever its pattern.  Measured (profiles/r03zd_solver_switch.log,
    // solve kernels per window solve): 50 keyframes, 900 gather entries, all in registers: PCG 0.71 ms, direct 1.30; 56
    // keyframes, 1 020 entries, the first to overflow: PCG 1.35, direct 1.26; 80 keyframes, 1 500 entries: 2.21 / 1.74; 50
    // keyframes with tracks of up to 20, 1 600 entries: 1.88 / 1.38 - the PCG's cost doubles the moment it spills, so that is
    // where the switch sits.
    // (movba_options::pcg_spill = 1 keeps the spilling PCG whatever the size; ::solver = 1 takes every window direct.)
    {
        const bool over = h->rows_kernel && h->pp.overflow;
        if (h->rows_kernel && ((over && !h->opt.pcg_spill) || h->opt.solver == 1)) h->rows_kernel = false;
    }
    // The banded factorisation in one workgroup (band_kernel.hip): every window whose band - in the numbering the upload has
    // settled on - fits one CU's LDS, the PCG's windows and the dense small ones of the direct solver alike.
    {
        int bw = 0;
        for (int p = nf; p < s().npairs; ++p) bw = std::max(bw, (int)(s().pair_j[p] - s().pair_i[p]));
        h->band_bw = bw;
        // ... where it is the faster of the two.  Its cost (tests/dev/band_scan.py, profiles/r04_band_scan.log; cycles from the
        // stamp build): per block step ~1 970 for the pivot block, ~780 per round of the panel, ~420 per round of 512 trailing
        // elements - with the AVERAGE number of blocks below a pivot, min(bw, nfree - 1 - k) over the steps -, plus assembly,
        // sweeps and epilogue; against the PCG's ~130 000 cycles whatever the size.  At a band of 9: 8 keyframes 0.49 ms per
        // resident window solve against 0.78, 16: 0.63 / 0.78, 24: 0.78 / 0.80, from 28 on the PCG wins (0.88 / 0.87; 40: 1.12 / 0.88).
        // MOVBA_BAND=0 / 1 (or movba_options::solver = 3 / 2) switch the choice off / force it.
        const int band_env = process_switches().band;
        double m_sum = 0.0;
        for (int k = 0; k < nf; ++k) m_sum += std::min(bw, nf - 1 - k);
        const double m_avg = nf > 0 ? m_sum / nf : 0.0;
        const double rt = std::ceil((m_avg * (m_avg + 1.0) * 18.0 + 6.0 * m_avg) / 512.0), rp = std::max(1.0, std::ceil(m_avg * 36.0 / 512.0));
        const double est = nf * (1970.0 + 780.0 * rp + 420.0 * rt) + 54.0 * nf * (bw + 1) + 360.0 * nf + 5000.0;
        // (the environment variable speaks for handles made with solver = 0 only)
        const bool forced = h->opt.solver == 2 || (h->opt.solver == 0 && band_env == 1);
        const bool off = h->opt.solver == 3 || h->opt.solver == 1 || (h->opt.solver == 0 && band_env == 0);
        const bool want = forced || (h->rows_kernel && est <= 130000.0);
        h->band = want && !off && band_supported(nf, bw);
    }
    if (h->rows_kernel && !h->band) build_coarse(h->st, h->pp.wave_row0, kPcgRowsThreads / 64);
    lap("pcg plan + coarse lists");
}

int Upload::lay_out_rest()
{
    // ---- pair region (second H2D copy): packed in the staging buffer right behind the edge region, `hole` bytes before
    // its place in the arena (the device-only arrays carved above sit in between) ----
    pair_begin = c.off; hole = pair_begin - L.max_end;
    o_items = c.take<Item>((size_t)s().nitems + 1); o_sched = c.take<SchedItem>(s().sched.size() + 1);
    o_pi = c.take<int32_t>(s().npairs + 1); o_pj = c.take<int32_t>(s().npairs + 1); o_pis = c.take<int32_t>(s().npairs + 1);
    o_rowptr = c.take<int32_t>(nf + 1); o_rowent = c.take<RowEnt>(s().row_ent.size() + 1);
    lane_plan.clear();
    if (h->rows_kernel) {
        // which two oriented blocks every lane of k_pcg_rows holds, and where their partial items are
        lane_plan.assign((size_t)kPcgRowsThreads * 12, -1);
        for (int wv = 0; wv < kPcgRowsThreads / 64; ++wv) {
            const int r0 = h->pp.wave_row0[wv], r1 = h->pp.wave_row0[wv + 1];
            const int P0 = s().row_ptr[r0] >> 1, P1 = s().row_ptr[r1] >> 1;
            for (int ln = 0; ln < 64 && P0 + ln < P1; ++ln)
                for (int k = 0; k < 2; ++k) {
                    const RowEnt &re = s().row_ent[2 * (P0 + ln) + k];
                    int32_t *pl = &lane_plan[((size_t)(wv * 64 + ln) * 3 + k) * 4];
                    if (re.block < 0) continue;
                    pl[0] = re.block; pl[1] = (re.col * 6) | (re.transposed ? (1 << 30) : 0);
                    pl[2] = s().pair_item_start[re.block]; pl[3] = s().pair_item_start[re.block + 1];
                }
            for (int ln = 0; ln < 64; ++ln) {               // owner lanes: what they need of their keyframe
                int32_t *pl = &lane_plan[((size_t)(wv * 64 + ln) * 3 + 2) * 4];
                pl[0] = pl[1] = pl[2] = pl[3] = 0;
                if (ln >= 6 * (r1 - r0)) continue;
                const int bi = r0 + ln / 6;
                pl[0] = s().pair_item_start[bi]; pl[1] = s().pair_item_start[bi + 1]; pl[2] = s().row_ptr[bi]; pl[3] = s().row_ptr[bi + 1];
            }
        }
        for (int wv = 0; wv <= kPcgRowsThreads / 64; ++wv) h->pp.wave_ent0[wv] = s().row_ptr[h->pp.wave_row0[wv]];
        h->pp.nrowent = s().row_ptr[nf];
    }
    // the diagonal items' records (DevWindow::rec_d): by keyframe and place in the pair
    rec_slots = 1;
    for (int hh = 0; hh < nf; ++hh) rec_slots = std::max(rec_slots, (int)(s().pair_item_start[hh + 1] - s().pair_item_start[hh]));
    // where the schur pass leaves the block of every single-item off-diagonal pair for those lanes (DevWindow::img_b)
    for (SchedItem &si : h->st.sched) {
        si.dst_a = -1; si.dst_b = -1;
        if (si.tag >= 0 && (si.tag & 1)) { const Item &it = s().items[(size_t)(si.tag >> 1)]; si.dst_a = it.pair * rec_slots + ((si.tag >> 1) - s().pair_item_start[it.pair]); }
    }
    if (h->rows_kernel && !h->pp.overflow) {
        std::vector<int32_t> slot_of_item((size_t)s().nitems, -1);
        for (size_t q = 0; q < s().sched.size(); ++q)      // (the slot of the wave that stores the item: place 0 among the item's waves)
            if (s().sched[q].tag >= 0 && (s().sched[q].sub & 0xff) == 0) slot_of_item[(size_t)(s().sched[q].tag >> 1)] = (int32_t)q;
        for (int t = 0; t < kPcgRowsThreads; ++t)
            for (int k = 0; k < 2; ++k) {
                const int32_t *pl = &lane_plan[((size_t)t * 3 + k) * 4];
                if (pl[0] < nf || pl[3] - pl[2] != 1) continue;         // no block, a diagonal one, or a pair cut into several items
                SchedItem &si = h->st.sched[(size_t)slot_of_item[(size_t)pl[2]]];
                const int32_t dst = 36 * k * kPcgRowsThreads + t;
                if ((pl[1] >> 30) & 1) si.dst_b = dst; else si.dst_a = dst;
            }
    }
    ncb = s().cblk_g.size();
    o_plan = c.take<int32_t>(lane_plan.size() + 4);
    o_cg = c.take<int32_t>(ncb + 1); o_ch = c.take<int32_t>(ncb + 1); o_cp = c.take<int32_t>(ncb + 2); o_ce = c.take<int32_t>(s().cblk_ent.size() + 1);
    o_cij = c.take<int32_t>(s().cblk_ij.size() + 1); o_multi = c.take<int32_t>(s().multi_pairs.size() + 1);
    o_pid = c.take<int32_t>((size_t)nf * nf + 1);                                // block -> pair map of the direct solver's assembly
    // one-launch direct solver: the static schedule depends on the number of block columns only (rebuilt when that changes)
    ntile = dense_ntile(nf);
    const bool dense_multi = process_switches().dense_multilaunch;
    // every workgroup of the one-launch solver must be resident while it runs: no more of them than the device (a partition
    // of an MI355X in CPX mode shows 32 compute units) has to give, a thirty-second held back as on the whole chip (248 of
    // 256); a plan that then needs more tiles per workgroup than fit LDS falls to the multi-launch solver
    const int dense_groups = std::min(kDenseMaxGroups, h->device_cus - std::max(1, h->device_cus / 32));
    if (h->dplan_nt != ntile) { build_dense_plan(ntile, h->dplan, std::max(dense_groups, 1)); h->dplan_nt = ntile; }
    dense_one = !dense_multi && dense_groups >= 8 && dense_persist_supported(h->dplan);
    o_dtp = c.take<int32_t>(dense_one ? h->dplan.task_ptr.size() : 1); o_dtk = c.take<DenseTask>(dense_one ? h->dplan.tasks.size() : 1);
    o_prange = c.take<int32_t>(dense_one ? 2 * (size_t)nf * nf : 1);
    if (!dev_structure) o_ent = c.take<int32_t>(ent_words());       // host-built entry lists (off-diagonal; the diagonal ones are their slot) travel with the pair region
    h2d = c.off;
    // ---- device-only region (what does not depend on the pair structure: carve_state) ----
    if (!state_carved) carve_state();
    const size_t part_stride = ((size_t)s().nitems * kPartStride + 31) / 32 * 32;
    o_part = c.take<double>(part_stride + 1); o_blocks = c.take<double>((size_t)s().npairs * 36 + 1);
    o_recd = c.take<double>((size_t)nf * rec_slots * 48 + 2); o_imgb = c.take<double>((size_t)72 * kPcgRowsThreads);
    o_blocks_ov = c.take<double>(h->rows_kernel && h->pp.overflow ? s().row_ent.size() * 36 + 2 : 2);
    o_blocks_c = c.take<double>((size_t)s().npairs * 36 + 1);
    // direct solver (dense_solve.hip): tiles of the lower block triangle + right-hand side row, diagonal factors, failure flag
    o_dtiles = c.take<double>(dense_tiles_doubles(nf)); o_ddiag = c.take<double>((size_t)ntile * kDenseNB * kDenseNB + 1); o_dfail = c.take<int32_t>(4);
    o_dx = c.take<double>((size_t)ntile * kDenseNB + 1);
    o_dflags = c.take<uint32_t>(dense_one ? (size_t)dense_flag_words(ntile) : 8);
    o_dcontrib = c.take<double>(dense_one ? (size_t)ntile * ntile * kDenseNB : 1);
    dense_stamps = process_switches().dense_stamps;
    o_dstamps = c.take<unsigned long long>(dense_one && dense_stamps ? 6 * h->dplan.tasks.size() : 1);
    total = c.off;

    // (a reallocation of the arena or of the staging buffer below must find the helper thread through with both: it reads
    //  the caller's arrays into the staging buffer and sends them to the arena it was given at the start)
    if (total > h->arena_cap || h2d - hole + misc_bytes > h->stage_cap) { const int rw = join_helper(); if (rw) return rw; }
    int rc = ensure_arena(h, total); if (rc) return rc;
    if (dev_first && (h->arena_gen != arena_gen_at_edge_copy || h2d - hole + misc_bytes > h->stage_cap)) return kRetryClassic;   // (tables the device made are gone with the old arena)
    if (h->arena_gen != arena_gen_at_edge_copy) {
        // the arena was reallocated (told by its generation: the new allocation may sit at the old address): queue the
        // edge region again (the staging copy is intact); the fill below then runs on the new arena
        HIP_TRY(hipMemcpyAsync(h->arena, sg, edge_bytes, hipMemcpyHostToDevice, h->stream));
    }
    if (h2d - hole + misc_bytes > h->stage_cap) {
        // (rare: huge host-built entry lists) a bigger staging buffer: ensure_stage drains the stream first, so the edge copy
        // has landed; the edge region is packed again only to keep the buffer self-consistent
        rc = ensure_stage(h, h2d - hole + misc_bytes); if (rc) return rc;
        sg = h->stage;
        pack_edges(true);
    }
    return MOVBA_OK;
}

void Upload::pack_pairs()
{
    if (!dev_structure && noff) {
        int32_t *eh = reinterpret_cast<int32_t *>(sp(o_ent));
        if (ent_packed) {
            unsigned long long *e64 = reinterpret_cast<unsigned long long *>(eh);
            for (size_t k = 0; k < noff; ++k) e64[k] = ent_pack(s().ent_i[k], s().ent_j[k], s().ent_l[k]);
        } else {
            put(eh, s().ent_i.data(), sizeof(int32_t) * noff); put(eh + noff, s().ent_j.data(), sizeof(int32_t) * noff);
            put(eh + 2 * noff, s().ent_l.data(), sizeof(int32_t) * noff);
        }
    }
    put(sp(o_items), s().items.data(), sizeof(Item) * (size_t)s().nitems);
    put(sp(o_sched), s().sched.data(), sizeof(SchedItem) * s().sched.size());
    put(sp(o_pi), s().pair_i.data(), sizeof(int32_t) * s().npairs);
    put(sp(o_pj), s().pair_j.data(), sizeof(int32_t) * s().npairs);
    put(sp(o_pis), s().pair_item_start.data(), sizeof(int32_t) * (s().npairs + 1));
    put(sp(o_rowptr), s().row_ptr.data(), sizeof(int32_t) * (nf + 1));
    put(sp(o_rowent), s().row_ent.data(), sizeof(RowEnt) * s().row_ent.size());
    if (!lane_plan.empty()) put(sp(o_plan), lane_plan.data(), sizeof(int32_t) * lane_plan.size());
    put(sp(o_cg), s().cblk_g.data(), sizeof(int32_t) * ncb);
    put(sp(o_ch), s().cblk_h.data(), sizeof(int32_t) * ncb);
    put(sp(o_cp), s().cblk_ptr.data(), sizeof(int32_t) * s().cblk_ptr.size());
    put(sp(o_ce), s().cblk_ent.data(), sizeof(int32_t) * s().cblk_ent.size());
    put(sp(o_cij), s().cblk_ij.data(), sizeof(int32_t) * s().cblk_ij.size());
    put(sp(o_multi), s().multi_pairs.data(), sizeof(int32_t) * s().multi_pairs.size());
    put(sp(o_pid), s().pid.data(), sizeof(int32_t) * (size_t)nf * nf);
    if (dense_one) {
        put(sp(o_dtp), h->dplan.task_ptr.data(), sizeof(int32_t) * h->dplan.task_ptr.size());
        put(sp(o_dtk), h->dplan.tasks.data(), sizeof(DenseTask) * h->dplan.tasks.size());
        int32_t *pr = reinterpret_cast<int32_t *>(sp(o_prange));
        for (size_t q = 0; q < (size_t)nf * nf; ++q) {
            const int32_t pair = s().pid[q];
            pr[2 * q] = pair >= 0 ? s().pair_item_start[pair] : 0;
            pr[2 * q + 1] = pair >= 0 ? s().pair_item_start[pair + 1] : 0;
        }
    }
    lap("carve + pack pair region");
}

int Upload::send_pairs()
{
    const double t2 = now_ms();
    h->prof.structure_ms += (t2 - t0) - upload_host_ms;
    // The pair region (~100 KB at cfg3) stands between the last structure kernel and the solve's first pass over the pairs: a
    // copy command costs it ~10 us to start and ~9 us to be seen finished by the kernel behind it; read out of the (mapped)
    // staging buffer by k_ingest it is one more kernel in the chain.  Large regions (host-built entry lists) take the copy engine.
    if (h2d - pair_begin <= (size_t)1 << 20) {
        IngestArgs ia{};
        ia.seg[0] = IngestSeg{ h->stage_dev + L.max_end, h->arena + pair_begin, ((h2d - pair_begin) + 3) & ~(size_t)3 };
        ia.nseg = 1; ia.counter = nullptr; ia.wait_for = 0;
        HIP_TRY(launch_ingest(ia, h->stream));
    } else
        HIP_TRY(hipMemcpyAsync(h->arena + pair_begin, sg + L.max_end, h2d - pair_begin, hipMemcpyHostToDevice, h->stream));
    if (!(filled_early && fill_gen == h->arena_gen)) { const int rq = launch_slotpt(); if (rq) return rq; }
    if (dev_structure && !(filled_early && fill_gen == h->arena_gen)) { const int rq = launch_fill(); if (rq) return rq; }
    // the solve's kernels start behind the caller's arrays on the copy stream (the structure pass above did not need them)
    { const int rw = join_helper(); if (rw) return rw; }          // (the helper has recorded copy_event by now)
    HIP_TRY(hipStreamWaitEvent(h->stream, h->copy_event, 0));
    // (arrays the copy engine reads out of the caller's own memory: through before the caller has them back)
    if (direct_raw) { HIP_TRY(hipEventSynchronize(h->copy_event)); raw_synced = true; }
    // no synchronise: the solve's kernels queue on the same stream behind these transfers, and the caller's buffers were
    // copied to the staging buffer already (the next upload synchronises before it refills it)
    lap("pair H2D (queued)");
    h->prof.upload_ms += now_ms() - t2 + upload_host_ms;
    h->h2d_bytes = h2d;
    return MOVBA_OK;
}

void Upload::device_view()
{
    DevWindow &w = h->win;
    char *a = h->arena;
    w = DevWindow{};
    w.NP = NP; w.P = P; w.E = E; w.nfree = nf; w.npairs = s().npairs; w.nitems = s().nitems; w.n_pt_blocks = nb;
    w.max_iters = d->max_iters; w.flags = d->flags; w.max_trials = d->max_trials > 0 ? d->max_trials : 10;
    w.fx = d->fx; w.fy = d->fy; w.cx = d->cx; w.cy = d->cy; w.huber_delta = d->huber_delta; w.chi2_gate = d->chi2_gate;
    w.g_pose = reinterpret_cast<int32_t *>(a + L.gpose); w.g_point = reinterpret_cast<int32_t *>(a + L.gpoint);
    w.pt_start = reinterpret_cast<int32_t *>(a + L.ptstart); w.perm = s().already_grouped ? nullptr : reinterpret_cast<int32_t *>(a + L.perm);
    w.hidx = reinterpret_cast<int32_t *>(a + L.hidx); w.free_pose = reinterpret_cast<int32_t *>(a + L.free_pose);
    w.obs = reinterpret_cast<double *>(a + L.obs); w.isig = reinterpret_cast<double *>(a + L.isig);
    w.obs_r = d->obs_right ? reinterpret_cast<double *>(a + L.obsr) : nullptr; w.bf = d->bf; w.stereo = stereo ? 1 : 0;
    w.kcam = L.has_kcam ? reinterpret_cast<const double *>(a + L.kcam) : nullptr;
    w.slot = reinterpret_cast<int32_t *>(a + L.slot);
    w.obs_pm = reinterpret_cast<double *>(a + o_obspm); w.obsr_pm = reinterpret_cast<double *>(a + o_obsrpm);
    {
        const int32_t *ed = reinterpret_cast<const int32_t *>(a + o_ent);
        w.ent_i = ed; w.ent_j = ed + noff; w.ent_l = ed + 2 * noff;
        w.ent64 = ent_packed ? reinterpret_cast<const unsigned long long *>(a + o_ent) : nullptr;
        w.slot_point = reinterpret_cast<const int32_t *>(a + o_slotpt); w.n_diag = s().E_free;
    }
    w.items = reinterpret_cast<Item *>(a + o_items);
    w.sched = reinterpret_cast<SchedItem *>(a + o_sched); w.sched_per_xcd = s().sched_per_xcd;
    w.pair_i = reinterpret_cast<int32_t *>(a + o_pi); w.pair_j = reinterpret_cast<int32_t *>(a + o_pj);
    w.pair_item_start = reinterpret_cast<int32_t *>(a + o_pis); w.row_ptr = reinterpret_cast<int32_t *>(a + o_rowptr);
    w.row_ent = reinterpret_cast<RowEnt *>(a + o_rowent);
    w.lane_plan = reinterpret_cast<int32_t *>(a + o_plan);
    w.n_agg = s().n_agg; w.n_cblk = (int32_t)ncb;
    w.cblk_g = reinterpret_cast<int32_t *>(a + o_cg); w.cblk_h = reinterpret_cast<int32_t *>(a + o_ch);
    w.cblk_ptr = reinterpret_cast<int32_t *>(a + o_cp); w.cblk_ent = reinterpret_cast<int32_t *>(a + o_ce);
    w.cblk_ij = reinterpret_cast<int32_t *>(a + o_cij);
    w.multi_pairs = reinterpret_cast<int32_t *>(a + o_multi); w.n_multi = (int32_t)s().multi_pairs.size();
    w.pose0 = reinterpret_cast<double *>(a + L.pose0); w.point0 = reinterpret_cast<double *>(a + L.point0);
    for (int b = 0; b < 2; ++b) {
        DevState &S = w.st[b];
        S.pose = reinterpret_cast<double *>(a + o_st[b][0]); S.Rt = reinterpret_cast<double *>(a + o_st[b][1]);
        S.point = reinterpret_cast<double *>(a + o_st[b][2]); S.Hll = reinterpret_cast<double *>(a + o_st[b][3]);
        S.bl = reinterpret_cast<double *>(a + o_st[b][4]); S.erecA = reinterpret_cast<double *>(a + o_st[b][5]);
        S.Fpart = reinterpret_cast<double *>(a + o_st[b][8]);
    }
    w.part = reinterpret_cast<double *>(a + o_part); w.blocks = reinterpret_cast<double *>(a + o_blocks);
    w.rec_d = reinterpret_cast<double *>(a + o_recd); w.img_b = reinterpret_cast<double *>(a + o_imgb); w.rec_slots = rec_slots;
    w.blocks_c = reinterpret_cast<double *>(a + o_blocks_c); w.blocks_ov = reinterpret_cast<double *>(a + o_blocks_ov);
    w.aci = reinterpret_cast<float *>(a + o_aci); w.ac_prev = reinterpret_cast<double *>(a + o_aci) + 2 * kCoarseDim * kCoarseDim; w.aci_tag = reinterpret_cast<int32_t *>(a + o_acitag);
    w.bp = reinterpret_cast<double *>(a + o_bp); w.xp = reinterpret_cast<double *>(a + o_xp);
    w.scale_part = reinterpret_cast<double *>(a + o_scale); w.hmax_part = reinterpret_cast<double *>(a + o_hmax);
    w.dec_rec = reinterpret_cast<unsigned *>(a + o_tick);
    w.ctrl = reinterpret_cast<Ctrl *>(a + o_ctrl); w.hstat = h->hstat_dev; w.ctrl_out = h->ctrl_host_dev;
    w.out_chi2 = reinterpret_cast<double *>(a + o_chi2); w.out_outlier = reinterpret_cast<uint8_t *>(a + o_outl);
    w.dense.tiles = reinterpret_cast<double *>(a + o_dtiles); w.dense.diagL = reinterpret_cast<double *>(a + o_ddiag);
    w.dense.pid = reinterpret_cast<const int32_t *>(a + o_pid); w.dense.fail = reinterpret_cast<int32_t *>(a + o_dfail);
    w.dense.prange = dense_one ? reinterpret_cast<const int32_t *>(a + o_prange) : nullptr;
    w.dense.ntile = ntile; w.dense.n = 6 * nf; w.dense.xsol = reinterpret_cast<double *>(a + o_dx);
    w.dense.task_ptr = reinterpret_cast<const int32_t *>(a + o_dtp); w.dense.tasks = reinterpret_cast<const DenseTask *>(a + o_dtk);
    w.dense.flags = reinterpret_cast<unsigned *>(a + o_dflags); w.dense.failw = w.dense.flags + dense_flag_count(ntile);
    w.dense.ctag = w.dense.flags + dense_ctag_word(ntile);
    w.dense.contrib = reinterpret_cast<double *>(a + o_dcontrib);
    w.dense.stamps = dense_one && dense_stamps ? reinterpret_cast<unsigned long long *>(a + o_dstamps) : nullptr;
    w.dense.G = dense_one ? h->dplan.G : 0; w.dense.slots = dense_one ? h->dplan.slots : 0;
    h->dense_flags_clean = false; h->dense_epoch = 0;
    w.direct_only = h->rows_kernel ? 0 : 1;
    w.wait_ticks = 2000000ull;
    w.lds_poses = point_lds_need(NP, nf) <= kPointLdsLimit ? 1 : 0;
}

int Upload::run(bool allow_dev_first)
{
    int rc = begin(); if (rc) return rc;
    dev_first = allow_dev_first && dev_first_eligible();
    if (dev_first) {
        // arrays of the caller that lie in movba_host_alloc memory (pinned, mapped) are read by the device where they are
        direct_raw = host_block_view(d->edge_pose, sizeof(int32_t) * (size_t)E) && host_block_view(d->edge_point, sizeof(int32_t) * (size_t)E) && host_block_view(d->obs, sizeof(double) * 2 * (size_t)E) && host_block_view(d->inv_sigma2, sizeof(double) * (size_t)E) &&
                     host_block_view(d->poses, sizeof(double) * 7 * (size_t)NP) && host_block_view(d->points, sizeof(double) * 3 * (size_t)P) &&
                     (!d->obs_right || host_block_view(d->obs_right, sizeof(double) * (size_t)E));
    }
    post_helper();
    if (dev_first) {
        rc = group_on_device(); if (rc) return rc;
        ent_packed = s().E_free < kEntPackSlots && P < kEntPackPoints && !HOOK(h, entries_unpacked);
        dev_structure = true;
        rc = after_counts(); if (rc) return rc;
        choose_solver();
        rc = lay_out_rest(); if (rc) return rc;
        pack_pairs();
        rc = send_pairs(); if (rc) return rc;
        device_view();
        h->uploaded = true;
        return MOVBA_OK;
    }
    rc = group(); if (rc || done) return rc;
    rc = pack_derived(); if (rc) return rc;
    rc = send_edge_a(); if (rc) return rc;
    // The per-pair entry lists are counted and filled on the GPU (struct_kernels.hip) when the caller's edges are
    // already grouped by map point (the reference's own order) and the pair-bin masks fit in LDS; otherwise on the host.
    // (on the device: up to 80 free keyframes, and as many keyframes in all as the kernels' LDS image has room for)
    const bool on_device = s().already_grouped && s().nfree > 0 && !HOOK(h, host_structure) && E >= kDeviceStructureMinEdges;
    const bool masks_fit = s().nfree <= 80 && struct_lds_fits(s().nfree, NP);
    // entry lists and slot -> point map: device-only, carved ahead of the pair region so that the fill kernel can be
    // launched before the pair region is laid out (host-built entry lists travel inside the pair region instead)
    // 8-byte packed entries when slots and point ids fit (any realistic window; the test hook entries_unpacked keeps the 12-byte form)
    ent_packed = s().E_free < kEntPackSlots && P < kEntPackPoints && !HOOK(h, entries_unpacked);
    // (beyond the pair-bin masks: the sort-based pass, for packed entries and up to kSortedMaxPoses keyframes)
    const bool sorted = on_device && !masks_fit && ent_packed && NP <= kSortedMaxPoses && !HOOK(h, no_sorted_structure);
    dev_structure = on_device && (masks_fit || sorted);
    rc = !dev_structure ? structure_on_host() : (masks_fit ? structure_on_device() : structure_on_device_sorted()); if (rc) return rc;
    if (!edge_b_queued) { rc = queue_edge_b(); if (rc) return rc; }
    choose_solver();
    rc = lay_out_rest(); if (rc) return rc;
    pack_pairs();
    rc = send_pairs(); if (rc) return rc;
    device_view();
    h->uploaded = true;
    return MOVBA_OK;
}

}  // namespace

int movba_lba_upload(movba_handle *h, const movba_lba_desc *d)
{
    if (!h || !d) return MOVBA_ERR_ARG;
    {
        Upload u(h, d);
        const int rc = u.run(true);
        if (rc != kRetryClassic) return rc;
    }
    // (edges not grouped by point, a free keyframe without an edge, an arena that had to grow under the device's own tables:
    //  once more with the grouping pass on this thread)
    Upload u(h, d);
    return u.run(false);
}

int movba_lba_reset(movba_handle *h)
{
    if (!h) return MOVBA_ERR_ARG;
    if (!h->uploaded) return MOVBA_ERR_STATE;
    h->ran = false;
    return MOVBA_OK;
}

}  // extern "C"

namespace {

// PCG parameters of the handle's resident window for a run
PcgParams run_pcg_params(const movba_handle *h)
{
    PcgParams pp = h->pp;
    pp.rel_tol = h->opt.pcg_rel_tol;
    // PCG cap: past ~200 iterations the direct solver is cheaper than carrying on, and a system that slow to converge is
    // one whose iterative answer would depart from the exact step anyway (see the park in k_pcg_rows)
    pp.max_iters = h->opt.pcg_max_iters > 0 ? h->opt.pcg_max_iters : 200;
    // 1: coarse level built beside the solve, one trial old; 2: small window (<= one keyframe per wave), built first and fresh
    bool one_row_per_wave = true;
    for (int wv = 0; wv < kPcgRowsThreads / 64; ++wv) one_row_per_wave &= pp.wave_row0[wv + 1] - pp.wave_row0[wv] <= 1;
    pp.use_coarse = (h->opt.pcg_coarse && h->rows_kernel) ? (one_row_per_wave ? 2 : 1) : 0;
    return pp;
}

// the direct solve of the current trial: one launch when the window's schedule fits (dense_persist.hip), else the launch per
// block column of dense_solve.hip.  The hand-off flags are zeroed once per uploaded window; every launch has its own epoch.
hipError_t queue_direct(movba_handle *h)
{
    const DevWindow &w = h->win;
    if (w.dense.G <= 0) return launch_dense_solve(w, h->stream);
    if (!h->dense_flags_clean) {
        const hipError_t e = hipMemsetAsync(w.dense.flags, 0, sizeof(uint32_t) * (size_t)dense_flag_words(w.dense.ntile), h->stream);
        if (e != hipSuccess) return e;
        h->dense_flags_clean = true; h->dense_epoch = 0;
    }
    hipError_t el;
    {
        DenseGate &g = dense_gate(h->device);
        std::lock_guard<std::mutex> lk(g.mu);
        if (!g.multi && g.only_stream && g.only_stream != h->stream) {
            // a second stream: what the first one has in flight is waited for once, on the host; the event chain takes over
            hipError_t e = hipStreamSynchronize(g.only_stream);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&g.ev, hipEventDisableTiming);
            if (e != hipSuccess) return e;
            g.multi = true; g.ev_recorded = false;
        }
        if (g.multi && g.ev_recorded) { const hipError_t e = hipStreamWaitEvent(h->stream, g.ev, 0); if (e != hipSuccess) return e; }
        el = launch_dense_persist(w, ++h->dense_epoch, h->stream);
        if (g.multi) { if (el == hipSuccess) { el = hipEventRecord(g.ev, h->stream); g.ev_recorded = el == hipSuccess; } }
        else g.only_stream = h->stream;
    }
    if (w.dense.stamps && el == hipSuccess && h->dense_epoch == 3) {
        // diagnostic (MOVBA_DENSE_STAMPS=1): the third direct launch of a window, task by task, in 10 ns ticks from the first start
        const size_t nt_ = h->dplan.tasks.size();
        std::vector<unsigned long long> st(6 * nt_);
        if (hipStreamSynchronize(h->stream) == hipSuccess && hipMemcpy(st.data(), w.dense.stamps, sizeof(unsigned long long) * 6 * nt_, hipMemcpyDeviceToHost) == hipSuccess) {
            unsigned long long t0 = ~0ull;
            for (size_t k = 0; k < nt_; ++k) if (st[6 * k] && st[6 * k] < t0) t0 = st[6 * k];
            static const char *names[] = { "ASM", "UPD", "DIAG", "OFF", "RHS", "BSX", "BSC", "EPI", "RUP", "UPD2", "COL" };
            for (int g = 0; g < h->dplan.G; ++g)
                for (int t = h->dplan.task_ptr[g]; t < h->dplan.task_ptr[g + 1]; ++t) {
                    const DenseTask &tk = h->dplan.tasks[t];
                    const unsigned long long *q = &st[6 * (size_t)t];
                    std::fprintf(stderr, "libmovba[dense]: wg %3d %-4s (%2d,%2d) k=%2d  start %7.2f us  wait %6.2f  work %6.2f", g, names[tk.op], tk.I, tk.K, tk.k,
                                 0.01 * (double)(q[0] - t0), q[1] ? 0.01 * (double)(q[1] - q[0]) : 0.0, 0.01 * (double)(q[5] - (q[1] ? q[1] : q[0])));
                    if (q[2] && q[3]) std::fprintf(stderr, "   [fetch %5.2f  sweep %5.2f  publish %5.2f]", 0.01 * (double)(q[2] - q[1]), 0.01 * (double)(q[3] - q[2]), 0.01 * (double)(q[5] - q[3]));
                    std::fprintf(stderr, "\n");
                }
        }
    }
    return el;
}

// k_band's second argument: the half bandwidth, and in the test build the trial whose factorisation is to park (+ 1, from bit 16)
inline int band_arg(const movba_handle *h) { return h->band_bw | ((HOOK(h, band_park_trial) + 1) << 16); }

// The LM trial loop of the handle's window on its own stream, from the state the device is in: a fresh window (after the
// setup launches), or one that parked itself during a batched run (its pause is then the first thing answered).
int lm_loop(movba_handle *h, bool parked)
{
    const DevWindow &w = h->win;
    hipStream_t s = h->stream;
    PcgParams pp = run_pcg_params(h);
    const int nrowent = (int)h->st.row_ent.size();
    // the reduced solve of a trial: on-chip PCG, or (larger windows, and from the first PCG failure on) the direct solver
    const bool band = h->band;                      // (an exact solve in one launch; it parks the solve only when a pivot comes out non-positive)
    bool direct = !h->rows_kernel && !band;
    int pauses_seen = 0;
    if (!parked) __atomic_store_n(&h->hstat->pause_seq, 0, __ATOMIC_RELAXED);
    const int max_trials = (w.max_iters > 0 ? w.max_iters : 0) * w.max_trials;
    double t_progress = now_ms();                   // last time the device's progress word changed or work was queued
    uint64_t last_pg = ~0ull;
    // k_finalize + the Ctrl read-back are queued speculatively behind a trial that is likely the last one, so that the
    // end of the solve does not wait for a host round trip; a later trial simply queues them again
    int t = 0, final_after = -1;
    auto queue_finalize = [&]() -> int {
        { ScopedEvents ev(h, KC_FINALIZE); HIP_TRY(launch_finalize(w, s)); }        // (also writes Ctrl to h->ctrl_host)
        // ... and the results go across the bus into the staging buffer right behind it: movba_lba_download then finds them
        // there instead of paying a launch and a stream synchronise of its own
        if (h->export_in_run) {
            ExportDst dst = export_dst(h->stage_dev, export_layout(w), h->exported[0], h->exported[1], h->exported[2]);
            if (h->user_dst[0]) dst.poses = h->user_dst[0];
            if (h->user_dst[1]) dst.points = h->user_dst[1];
            if (h->user_dst[2]) dst.chi2 = h->user_dst[2];
            HIP_TRY(launch_export(w, dst, s));
        }
        final_after = t;
        return MOVBA_OK;
    };
    auto queue_solve = [&]() -> int {
        if (band && !direct) { ScopedEvents ev(h, KC_PCG); HIP_TRY(launch_band(w, band_arg(h), s)); }      // (direct: the factorisation parked the solve)
        else if (direct) { ScopedEvents ev(h, KC_PCG); HIP_TRY(queue_direct(h)); }
        else { ScopedEvents ev(h, KC_PCG); HIP_TRY(launch_pcg_rows(w, nrowent, pp, t, s)); }
        return MOVBA_OK;
    };
    auto queue_tail = [&]() -> int {
        // (the pass's extra workgroup takes the LM decision: no launch of its own)
        { ScopedEvents ev(h, KC_BACKSUB); HIP_TRY(launch_backsub(w, s)); }
        return MOVBA_OK;
    };
    // The device parked the solve (k_pcg_rows gave up on trial `td`): every trial set queued behind has turned into no-ops.
    // Queue the direct solver for that trial (its schur partials are still in place) and carry on in direct mode.
    auto answer_pause = [&]() -> int {
        pauses_seen = rd_pause(h->hstat);
        direct = true;
        t = (int)(rd_progress(h->hstat) & 0xffffff);
        int rq = queue_solve(); if (rq != MOVBA_OK) return rq;
        rq = queue_tail(); if (rq != MOVBA_OK) return rq;
        ++t;
        final_after = -1;
        return MOVBA_OK;
    };
    for (;;) {
        for (; t < max_trials; ++t) {
            // stay at most run_ahead trial sets ahead of the device, and never further than the outer iterations that are
            // left: with R iterations to go at most R more trials run unless one is rejected, so the sets queued beyond that
            // would almost always be no-op launches (~5 us each) at the end of the solve
            // (k_decide publishes trials_done, it and done as one word)
            bool finished = false, paused = false;
            for (;;) {
                const uint64_t pg = rd_progress(h->hstat);
                if (pg != last_pg) { last_pg = pg; t_progress = now_ms(); }
                const int td = (int)(pg & 0xffffff), it_done = (int)((pg >> 24) & 0xffffff);
                if ((pg >> 48) & 1) { finished = true; break; }
                if (rd_pause(h->hstat) != pauses_seen) { paused = true; break; }
                const int left = w.max_iters - it_done;
                const int limit = left < h->opt.run_ahead ? (left > 1 ? left : 1) : h->opt.run_ahead;
                if (t - td < limit) break;
                if (final_after != t && t - td < h->opt.run_ahead) { const int rq = queue_finalize(); if (rq != MOVBA_OK) return rq; }
                if (caller_stop(h->stop)) wr_stop(h->hstat, 1);
                if (now_ms() - t_progress > watchdog_ms()) {
                    // raise the device-side stop flag on the way out: whatever is still queued on the stream turns into
                    // no-op launches as soon as a k_decide sees it, and the window has to be uploaded again
                    std::fprintf(stderr, "libmovba: device made no progress for %.0f ms, giving up\n", watchdog_ms());
                    wr_stop(h->hstat, 1); h->uploaded = false;
                    (void)hipStreamSynchronize(s);          // nothing of this solve is left queued when the caller gets the error
                    return MOVBA_ERR_HIP;
                }
                host_relax(h->opt.host_wait);
            }
            if (finished) break;
            if (paused) { const int rq = answer_pause(); if (rq != MOVBA_OK) return rq; --t; continue; }
            if (caller_stop(h->stop)) wr_stop(h->hstat, 1);
            t_progress = now_ms();                          // (new work queued counts as progress)
            if (w.nitems > 0) { ScopedEvents ev(h, KC_SCHUR); HIP_TRY(launch_schur(w, 0, s)); }
            { const int rq = queue_solve(); if (rq != MOVBA_OK) return rq; }
            { const int rq = queue_tail(); if (rq != MOVBA_OK) return rq; }
        }
        if (final_after != t) { const int rq = queue_finalize(); if (rq != MOVBA_OK) return rq; }
        HIP_TRY(hipStreamSynchronize(s));
        // a park that happened behind the last queued set is only seen now
        if (rd_pause(h->hstat) == pauses_seen) break;
        const int rq = answer_pause(); if (rq != MOVBA_OK) return rq;
    }
    harvest_events(h);
    return MOVBA_OK;
}

}  // namespace

extern "C" {

int movba_lba_run(movba_handle *h)
{
    if (!h) return MOVBA_ERR_ARG;
    if (!h->uploaded) return MOVBA_ERR_STATE;
    HIP_TRY(hipSetDevice(h->device));
    h->ran = false; h->run_status = MOVBA_OK;
    if (h->early_status != MOVBA_OK) { h->ran = true; return h->early_status; }
    // early return before the solve (src/Optimizer.cc:749-751); not sticky: the next run looks at the flag again
    if (caller_stop(h->stop)) { h->run_status = MOVBA_STOPPED; h->ran = true; return MOVBA_STOPPED; }
    // (a registered export buffer too small for this window is ignored rather than overrun)
    h->win.pose_export = (h->pose_export && h->pose_export_cap >= (int64_t)sizeof(double) * 7 * h->win.NP) ? h->pose_export : nullptr;
    const DevWindow &w = h->win;
    hipStream_t s = h->stream;
    // (the staging buffer is free once the upload's copies, queued ahead of every kernel of the run, have left it; it is as
    // large as the upload needed, which is more than the results take)
    h->export_in_run = h->export_hint && export_layout(w).end <= h->stage_cap;
    if (!h->export_in_run) for (int k = 0; k < 3; ++k) { h->user_dst[k] = nullptr; h->user_host[k] = nullptr; }

    // One kernel waits for other workgroups INSIDE a launch: the one-launch direct solver (dense_persist.hip).  Its waits are
    // bounded (DevWindow::wait_ticks, 20 ms) and cannot deadlock on an otherwise idle device, but its workgroups are not
    // GUARANTEED their residency either: another process on the GPU, a CU-masked or partitioned device, any other kernel
    // holding the CUs.  A solve in which a wait was given up (Ctrl::n_sync_timeouts) is therefore run AGAIN from the uploaded
    // state on the path that waits for nothing - the direct solver one launch per block column (dense_solve.hip) - instead
    // of handing the caller an error: the reference never skips a solve for such a reason (src/Optimizer.cc:535).
    // movba_lba_result::n_sync_timeouts reports that it happened.  (The deciding wave of the back-substitution pass waits
    // too, for producers that wait for nothing themselves: that wait has the host watchdog's bound, kernels.hip.)
    h->sync_retries = 0;
    // per-attempt values of the window descriptor, put back on every way out of the loop (an error return included)
    struct Restore {
        DevWindow &w; const int32_t G; const unsigned long long ticks;
        ~Restore() { w.dense.G = G; w.wait_ticks = ticks; }
    } restore{ h->win, h->win.dense.G, h->win.wait_ticks };
    for (int attempt = 0;; ++attempt) {
        const bool careful = attempt > 0;
        h->win.wait_ticks = (!careful && HOOK(h, wait_ticks) >= 0) ? (unsigned long long)HOOK(h, wait_ticks) : restore.ticks;
        h->win.dense.G = careful ? 0 : restore.G;
        __atomic_store_n(&h->hstat->progress, (uint64_t)0, __ATOMIC_RELAXED); wr_stop(h->hstat, 0);
        {   // state 0 from the uploaded estimates, first linearisation, lambda_0 and F0
            ScopedEvents ev(h, KC_SETUP);
            if (!h->early_setup) {          // (else: queued by the upload already, behind the edge data)
                HIP_TRY(launch_init(w, s));
                HIP_TRY(launch_linearize(w, s));
            }
            h->early_setup = false;
            if (w.nitems > 0) HIP_TRY(launch_schur(w, 1, s));
            HIP_TRY(launch_lambda_init(w, s));
        }
        const int rl = lm_loop(h, false);
        if (rl != MOVBA_OK) return rl;
        if (h->ctrl_host->n_sync_timeouts > 0 && !careful) {
            std::fprintf(stderr, "libmovba: a workgroup gave up waiting for another in %d launch(es) of this solve: running it again, the direct solver launch by launch\n",
                         h->ctrl_host->n_sync_timeouts);
            h->sync_retries = h->ctrl_host->n_sync_timeouts;
            h->dense_flags_clean = false;           // (hand-off flags of the abandoned launches: zeroed again before the next one-launch solve)
            continue;
        }
        break;
    }
#ifdef MOVBA_CLOCK_STAMP
    std::fprintf(stderr, "libmovba[stamp]: k_pcg_rows %llu shader cycles in %llu x 10 ns -> %.3f GHz\n", h->ctrl_host->dbg_cycles,
                 h->ctrl_host->dbg_ticks, h->ctrl_host->dbg_ticks ? 0.1 * (double)h->ctrl_host->dbg_cycles / (double)h->ctrl_host->dbg_ticks : 0.0);
    std::fprintf(stderr, "libmovba[stamp]: per-iteration segments (wave 0, cycles):");
    for (int k = 0; k < 8; ++k) std::fprintf(stderr, " s%d=%.0f", k, (double)h->ctrl_host->dbg_seg[k] / (h->ctrl_host->pcg_total_iters ? h->ctrl_host->pcg_total_iters : 1));
    for (int wv = 0; wv < 8; ++wv) {
        std::fprintf(stderr, "\nlibmovba[stamp]:   wave %d:", wv);
        for (int k = 0; k < 8; ++k) std::fprintf(stderr, " s%d=%.0f", k, (double)h->ctrl_host->dbg_wseg[wv][k] / (h->ctrl_host->pcg_total_iters ? h->ctrl_host->pcg_total_iters : 1));
    }
    std::fprintf(stderr, "\nlibmovba[stamp]: setup phases per launch (cycles):");
    for (int k = 0; k < 8; ++k) std::fprintf(stderr, " p%d=%.0f", k, (double)h->ctrl_host->dbg_seg2[k] / (h->ctrl_host->n_solves ? h->ctrl_host->n_solves : 1));
    std::fprintf(stderr, "\n");
#endif
    h->ran = true;
    return MOVBA_OK;
}

// Optimizer::LocalBundleAdjustment's solve on n resident windows at once (multi-session serving; BASELINE cfg5's windows when
// they share a GPU): every kernel of a trial is ONE launch over the concatenated windows — the point and schur grids of a
// single 50-keyframe window leave most of the chip idle, and its PCG occupies 2 of 256 CUs — with per-window LM state, so
// each window takes exactly the steps (and produces exactly the bits) of its solo movba_lba_run.
int movba_lba_run_batch(movba_handle *const *hs, int32_t n)
{
    if (!hs || n < 1) return MOVBA_ERR_ARG;
    for (int i = 0; i < n; ++i) {
        if (!hs[i]) return MOVBA_ERR_ARG;
        if (!hs[i]->uploaded) return MOVBA_ERR_STATE;
        if (hs[i]->device != hs[0]->device || hs[i]->stream != hs[0]->stream) {
            std::fprintf(stderr, "libmovba: the handles of a batch must share one device and one stream\n");
            return MOVBA_ERR_ARG;
        }
        for (int k = 0; k < i; ++k) if (hs[k] == hs[i]) return MOVBA_ERR_ARG;
    }
    movba_handle *h0 = hs[0];
    HIP_TRY(hipSetDevice(h0->device));
    hipStream_t s = h0->stream;
    // windows that return before the solve (nothing to do, no fixed keyframe, stop flag up) and windows without an on-chip
    // PCG (they take the direct solver's per-window launches) stay out of the batched launches
    std::vector<movba_handle *> act, solo;
    for (int i = 0; i < n; ++i) {
        movba_handle *h = hs[i];
        h->ran = false; h->run_status = MOVBA_OK; h->export_in_run = false; h->early_setup = false;      // (the batch queues every window's setup itself)
        if (h->early_status != MOVBA_OK) { h->ran = true; continue; }
        if (caller_stop(h->stop)) { h->run_status = MOVBA_STOPPED; h->ran = true; continue; }
        if (!h->rows_kernel || h->win.kcam) { solo.push_back(h); continue; }      // (direct-solver windows and windows with intrinsics by keyframe run on their own)
        act.push_back(h);
    }
    const int na = (int)act.size();
    if (na > 0) {
        bool stereo = act[0]->win.stereo != 0, ldsp = true, overflow = false, padded = true;
        for (movba_handle *h : act) {
            if ((h->win.stereo != 0) != stereo) {
                std::fprintf(stderr, "libmovba: a batch holds either stereo or monocular windows, not both\n");
                return MOVBA_ERR_ARG;
            }
            ldsp &= h->win.lds_poses != 0; overflow |= h->pp.overflow != 0; padded &= h->pp.padded != 0 && !h->pp.overflow;
        }
        // Groups of windows on streams of their own, out of phase: a group's PCG launch keeps 2 CUs per window busy for most
        // of a trial while its point / schur launches fill the chip for the rest, so the PCG of one group runs beside the
        // streaming kernels of the others.  Each window's own kernels still run in its solo order on one stream.
        // (two by default: with more streams than hardware queues left to the process the groups fall back into lockstep)
        int ngroups = na >= 2 ? 2 : 1;
        if (process_switches().batch_groups > 0) ngroups = std::max(1, std::min(std::min(kMaxGroups, na), process_switches().batch_groups));
        if (ngroups > 1 && !h0->batch_ev[0]) {
            for (hipEvent_t &e : h0->batch_ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            for (auto &ring : h0->batch_phase_ev) for (hipEvent_t &e : ring) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        // the second group runs on the device's shared copy stream (idle during a run): every stream the process creates
        // competes for the few hardware queues, and two groups that land on one queue run in turns (measured: 2.3 -> 3.7 ms)
        for (int g = 2; g < ngroups; ++g)
            if (!h0->batch_streams[g]) HIP_TRY(hipStreamCreateWithFlags(&h0->batch_streams[g], hipStreamNonBlocking));
        struct Group {
            std::vector<movba_handle *> hs;
            hipStream_t s = nullptr;
            BatchDev b{};
            int nb_point = 0, nb_schur = 0, nb_final = 0, nb_init = 0, max_trials = 0, max_iters = 0;
            size_t lds_lin = 0, lds_back = 0, lds_pcg = 0, lds_band = 0;
            bool any_band = false, any_pcg = false;     // windows solved by k_band_b / by k_pcg_rows_b (each window as in its solo run)
            int t = 0, final_after = -1;
            bool finished = false;
        } grp[kMaxGroups];
        for (int i = 0; i < na; ++i) grp[(int)((int64_t)i * ngroups / na)].hs.push_back(act[i]);
        grp[0].s = s;
        for (int g = 1; g < ngroups; ++g) grp[g].s = g == 1 ? h0->copy_stream : h0->batch_streams[g];
        // ---- device views, PCG plans and block prefixes of both groups in one buffer ----
        Carver c;
        size_t o_win[kMaxGroups], o_pp[kMaxGroups], o_bp[kMaxGroups], o_bs[kMaxGroups], o_bf[kMaxGroups], o_bi[kMaxGroups], o_bw[kMaxGroups];
        for (int g = 0; g < ngroups; ++g) {
            const size_t m = grp[g].hs.size();
            o_win[g] = c.take<DevWindow>(m); o_pp[g] = c.take<PcgParams>(m);
            o_bp[g] = c.take<int32_t>(m + 1); o_bs[g] = c.take<int32_t>(m + 1); o_bf[g] = c.take<int32_t>(m + 1); o_bi[g] = c.take<int32_t>(m + 1);
            o_bw[g] = c.take<int32_t>(m + 1);
        }
        if (c.off > h0->batch_cap) {
            HIP_TRY(hipStreamSynchronize(s));
            if (h0->batch_dev) { HIP_TRY(hipFree(h0->batch_dev)); h0->batch_dev = nullptr; }
            if (h0->batch_host) { HIP_TRY(hipHostFree(h0->batch_host)); h0->batch_host = nullptr; }
            h0->batch_cap = 0;
            const size_t cap = align_up(2 * c.off, 1 << 16);
            HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&h0->batch_host), cap, hipHostMallocDefault));
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h0->batch_dev), cap));
            h0->batch_cap = cap;
        }
        HIP_TRY(hipStreamSynchronize(s));           // the previous batch's H2D copy of this buffer has landed
        char *bh = h0->batch_host, *bd = h0->batch_dev;
        const int run_ahead = h0->opt.run_ahead;
        for (int g = 0; g < ngroups; ++g) {
            Group &G = grp[g];
            const int m = (int)G.hs.size();
            DevWindow *wins = reinterpret_cast<DevWindow *>(bh + o_win[g]);
            PcgParams *pps = reinterpret_cast<PcgParams *>(bh + o_pp[g]);
            int32_t *bp = reinterpret_cast<int32_t *>(bh + o_bp[g]), *bs = reinterpret_cast<int32_t *>(bh + o_bs[g]);
            int32_t *bf = reinterpret_cast<int32_t *>(bh + o_bf[g]), *bi = reinterpret_cast<int32_t *>(bh + o_bi[g]);
            int32_t *bwv = reinterpret_cast<int32_t *>(bh + o_bw[g]);
            bp[0] = bs[0] = bf[0] = bi[0] = 0;
            for (int i = 0; i < m; ++i) {
                movba_handle *h = G.hs[i];
                h->win.pose_export = (h->pose_export && h->pose_export_cap >= (int64_t)sizeof(double) * 7 * h->win.NP) ? h->pose_export : nullptr;
                __atomic_store_n(&h->hstat->progress, (uint64_t)0, __ATOMIC_RELAXED); wr_stop(h->hstat, 0); __atomic_store_n(&h->hstat->pause_seq, 0, __ATOMIC_RELAXED);
                wins[i] = h->win; wins[i].lds_poses = ldsp ? 1 : 0;
                pps[i] = run_pcg_params(h);
                bwv[i] = h->band ? band_arg(h) : -1;
                if (h->band) { G.any_band = true; G.lds_band = std::max(G.lds_band, band_lds_bytes(h->win.nfree, h->band_bw)); } else G.any_pcg = true;
                const DevWindow &w = h->win;
                bp[i + 1] = bp[i] + w.n_pt_blocks + 1;          // (+ the deciding workgroup of the window's back-substitution pass)
                bs[i + 1] = bs[i] + (w.nitems > 0 ? schur_blocks(w) : 0);
                bf[i + 1] = bf[i] + (w.E + 255) / 256;
                const int work = w.NP > (3 * w.P) / 2 ? w.NP : (3 * w.P) / 2;
                int nb = (work + 255) / 256; nb = nb < 1 ? 1 : (nb > 1024 ? 1024 : nb);
                bi[i + 1] = bi[i] + nb;
                G.lds_lin = std::max(G.lds_lin, point_lds_bytes_for(w, false, ldsp)); G.lds_back = std::max(G.lds_back, point_lds_bytes_for(w, true, ldsp));
                G.lds_pcg = std::max(G.lds_pcg, pcg_rows_lds_bytes(w.nfree, (int)h->st.row_ent.size(), padded && !overflow));
                G.max_trials = std::max(G.max_trials, (w.max_iters > 0 ? w.max_iters : 0) * w.max_trials);
                G.max_iters = std::max(G.max_iters, w.max_iters);
            }
            G.nb_point = bp[m]; G.nb_schur = bs[m]; G.nb_final = bf[m]; G.nb_init = bi[m];
            G.b.wins = reinterpret_cast<const DevWindow *>(bd + o_win[g]); G.b.pps = reinterpret_cast<const PcgParams *>(bd + o_pp[g]);
            G.b.blk_point = reinterpret_cast<const int32_t *>(bd + o_bp[g]); G.b.blk_schur = reinterpret_cast<const int32_t *>(bd + o_bs[g]);
            G.b.blk_final = reinterpret_cast<const int32_t *>(bd + o_bf[g]); G.b.blk_init = reinterpret_cast<const int32_t *>(bd + o_bi[g]);
            G.b.band_bw = reinterpret_cast<const int32_t *>(bd + o_bw[g]);
            G.b.n = m;
        }
        HIP_TRY(hipMemcpyAsync(bd, bh, c.off, hipMemcpyHostToDevice, s));
        if (ngroups > 1) {                          // the other streams start behind the uploads and this copy
            HIP_TRY(hipEventRecord(h0->batch_ev[0], s));
            for (int g = 1; g < ngroups; ++g) HIP_TRY(hipStreamWaitEvent(grp[g].s, h0->batch_ev[0], 0));
        }
        // ---- setup: state 0, first linearisation, lambda_0 and F0 of every window ----
        for (int g = 0; g < ngroups; ++g) {
            Group &G = grp[g];
            HIP_TRY(launch_init_batch(G.b, G.nb_init, G.s));
            HIP_TRY(launch_point_batch(G.b, G.nb_point, false, stereo, ldsp, G.lds_lin, G.s));
            if (G.nb_schur > 0) HIP_TRY(launch_schur_batch(G.b, G.nb_schur, 1, stereo, G.s));
            HIP_TRY(launch_lambda_init_batch(G.b, G.s));
        }
        // ---- trial sets: each group a bounded number of sets ahead of its slowest window still running ----
        double t_progress = now_ms();               // last time some window's progress word changed or work was queued
        uint64_t last_sum = ~0ull;
        for (;;) {
            bool all_finished = true;
            uint64_t pg_sum = 0;
            for (int g = 0; g < ngroups; ++g) {
                Group &G = grp[g];
                if (G.finished) continue;
                int td_min = 1 << 30, it_min = 1 << 30, running = 0;
                for (movba_handle *h : G.hs) {
                    const uint64_t pg = rd_progress(h->hstat);
                    pg_sum += pg;
                    if (((pg >> 48) & 1) || rd_pause(h->hstat) != 0) continue;      // done, or parked for the direct solver
                    ++running;
                    td_min = std::min(td_min, (int)(pg & 0xffffff)); it_min = std::min(it_min, (int)((pg >> 24) & 0xffffff));
                    if (caller_stop(h->stop)) wr_stop(h->hstat, 1);
                }
                if (running == 0 || G.t >= G.max_trials) {
                    if (G.final_after != G.t) { HIP_TRY(launch_finalize_batch(G.b, G.nb_final, G.s)); G.final_after = G.t; }
                    G.finished = true;
                    continue;
                }
                all_finished = false;
                const int left = G.max_iters - it_min;
                const int limit = left < run_ahead ? (left > 1 ? left : 1) : run_ahead;
                if (G.t - td_min >= limit) {
                    if (G.final_after != G.t && G.t - td_min < run_ahead) { HIP_TRY(launch_finalize_batch(G.b, G.nb_final, G.s)); G.final_after = G.t; }
                    continue;
                }
                // keep the groups out of phase: group g's schur launch of trial t starts when group g-1's has ended (otherwise
                // the streams drift into lockstep, all in their PCG at once with the chip idle: rocprofv3 trace); group g-1's
                // launch of that trial must therefore be queued first
                if (g > 0 && !grp[g - 1].finished && grp[g - 1].t <= G.t) continue;
                if (g > 0 && grp[g - 1].t > G.t) HIP_TRY(hipStreamWaitEvent(G.s, h0->batch_phase_ev[g - 1][G.t % kPhaseEvents], 0));
                if (G.nb_schur > 0) HIP_TRY(launch_schur_batch(G.b, G.nb_schur, 0, stereo, G.s));
                if (g + 1 < ngroups) HIP_TRY(hipEventRecord(h0->batch_phase_ev[g][G.t % kPhaseEvents], G.s));
                if (G.any_pcg) HIP_TRY(launch_pcg_rows_batch(G.b, overflow, padded && !overflow, G.lds_pcg, G.t, G.s));
                if (G.any_band) HIP_TRY(launch_band_batch(G.b, G.lds_band, G.s));
                HIP_TRY(launch_point_batch(G.b, G.nb_point, true, stereo, ldsp, G.lds_back, G.s));
                G.t += 1;
                t_progress = now_ms();
            }
            if (all_finished) break;
            if (pg_sum != last_sum) { last_sum = pg_sum; t_progress = now_ms(); }
            if (now_ms() - t_progress > watchdog_ms()) {
                std::fprintf(stderr, "libmovba: device made no progress for %.0f ms, giving up\n", watchdog_ms());
                for (movba_handle *h : act) { wr_stop(h->hstat, 1); h->uploaded = false; }
                for (int g = 0; g < ngroups; ++g) (void)hipStreamSynchronize(grp[g].s);
                return MOVBA_ERR_HIP;
            }
            host_relax(h0->opt.host_wait);
        }
        for (int g = 1; g < ngroups; ++g) {         // the callers' stream ends behind the others
            HIP_TRY(hipEventRecord(h0->batch_ev[g], grp[g].s));
            HIP_TRY(hipStreamWaitEvent(s, h0->batch_ev[g], 0));
        }
        HIP_TRY(hipStreamSynchronize(s));           // (every group was finalised behind its last trial set, in stream order)
        // windows whose PCG gave up parked themselves: each finishes on the direct solver from where it stands
        for (movba_handle *h : act) {
            if (rd_pause(h->hstat) != 0 && !((rd_progress(h->hstat) >> 48) & 1)) {
                const int rl = lm_loop(h, true);
                if (rl != MOVBA_OK) return rl;
            }
            h->ran = true;
        }
    }
    for (movba_handle *h : solo) { const int rc = movba_lba_run(h); if (rc < 0) return rc; }
    return MOVBA_OK;
}

int movba_lba_download(movba_handle *h, movba_lba_result *res)
{
    if (!h || !res) return MOVBA_ERR_ARG;
    if (!h->uploaded || !h->ran) return MOVBA_ERR_STATE;
    HIP_TRY(hipSetDevice(h->device));
    const int pre = h->early_status != MOVBA_OK ? h->early_status : h->run_status;
    res->status = pre;
    res->iters_done = 0; res->n_solves = 0; res->n_outliers = 0; res->pcg_iters = 0; res->last_rejected = 0;
    res->lambda = 0; res->cost0 = 0; res->cost = 0; res->n_trace = 0;
    res->n_direct = 0; res->direct_from = -1; res->n_chol_fail = 0; res->n_pcg_giveups = 0; res->n_sync_timeouts = 0; res->n_band = 0;
    if (pre != MOVBA_OK) return pre;
    const double t0 = now_ms();
    const DevWindow &w = h->win;
    const Ctrl &c = *h->ctrl_host;
    const size_t nb_pose = sizeof(double) * 7 * (size_t)w.NP, nb_pt = sizeof(double) * 3 * (size_t)w.P, nb_chi = sizeof(double) * (size_t)w.E;
    char *sg = h->stage;
    const ExportLayout L = export_layout(w);
    const size_t o_pose = L.o_pose, o_pt = L.o_pt, o_chi = L.o_chi, o_out = L.o_out;
    // results a movba_lba_solve exported straight into the caller's own (movba_host_alloc) arrays are not in the staging
    // buffer: a download into other arrays exports again
    if (h->export_in_run) {
        void *const arr[3] = { res->poses, res->points, res->chi2 };
        for (int k = 0; k < 3; ++k) if (arr[k] && (!h->exported[k] || (h->user_host[k] && h->user_host[k] != arr[k]))) h->export_in_run = false;
    }
    if (!h->export_in_run) {
        for (int k = 0; k < 3; ++k) h->user_host[k] = nullptr;
        int rs = ensure_stage(h, L.end); if (rs) return rs;
        sg = h->stage;
        HIP_TRY(launch_export(w, export_dst(h->stage_dev, L, res->poses != nullptr, res->points != nullptr, res->chi2 != nullptr), h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    // out of the pinned buffer into the caller's arrays (handing half of it to the helper thread saved 20 us when the
    // thread was awake and cost 0.4 ms when it had to be woken: not worth it for a copy the caller waits on)
    // (arrays in movba_host_alloc memory were written by the export kernel itself)
    const bool in_place = h->export_in_run;
    if (res->chi2 && !(in_place && h->user_host[2] == res->chi2)) std::memcpy(res->chi2, sg + o_chi, nb_chi);
    if (res->poses && !(in_place && h->user_host[0] == res->poses)) std::memcpy(res->poses, sg + o_pose, nb_pose);
    if (res->points && !(in_place && h->user_host[1] == res->points)) std::memcpy(res->points, sg + o_pt, nb_pt);
    if (res->outlier) std::memcpy(res->outlier, sg + o_out, (size_t)w.E);
    int n_out = 0;
    {
        // (flags are 0 / 1 bytes: eight at a time, summed by one multiplication)
        const uint8_t *of = reinterpret_cast<const uint8_t *>(sg + o_out);
        uint64_t acc = 0;
        int e = 0;
        for (; e + 8 <= w.E; e += 8) { uint64_t v; std::memcpy(&v, of + e, 8); acc += (v * 0x0101010101010101ull) >> 56; }
        for (; e < w.E; ++e) acc += of[e] != 0;
        n_out = (int)acc;
    }
    res->iters_done = c.iters_done; res->n_solves = c.n_solves; res->n_outliers = n_out;
    res->pcg_iters = c.pcg_total_iters; res->last_rejected = c.last_rejected;
    res->n_direct = c.n_direct; res->direct_from = c.direct_from; res->n_chol_fail = c.n_chol_fail; res->n_pcg_giveups = c.n_pause;
    res->n_band = c.n_band;
    res->lambda = c.lambda; res->cost0 = c.cost0; res->cost = c.F0;
    res->n_trace = c.n_trace;
    for (int k = 0; k < c.n_trace && k < MOVBA_MAX_TRACE; ++k) {
        res->tr_lambda[k] = c.tr_lambda[k]; res->tr_f0[k] = c.tr_f0[k]; res->tr_f1[k] = c.tr_f1[k];
        res->tr_rho[k] = c.tr_rho[k]; res->tr_accept[k] = c.tr_accept[k]; res->tr_pcg_iters[k] = c.tr_pcg[k];
    }
    res->n_sync_timeouts = c.n_sync_timeouts + h->sync_retries;
    h->prof.download_ms += now_ms() - t0;
    if (c.n_sync_timeouts > 0) {        // (only a batched run, which has no second attempt, or a second attempt that met a wait of the decide wave)
        // (the trials concerned were rejected like failed factorisations, so the state is a valid LM state — but not the one the
        //  reference's exact solver would have reached: never handed out as a success)
        std::fprintf(stderr, "libmovba: the one-launch direct solver gave up waiting between workgroups in %d solve(s): results discarded\n", c.n_sync_timeouts);
        res->status = MOVBA_ERR_DEVICE_WAIT;
        return MOVBA_ERR_DEVICE_WAIT;
    }
    return MOVBA_OK;
}

int movba_lba_solve(movba_handle *h, const movba_lba_desc *desc, movba_lba_result *res)
{
    if (!h || !desc || !res) return MOVBA_ERR_ARG;
    res->status = MOVBA_ERR_ARG;
    const bool lap_on = process_switches().time_solve;
    const double t_s0 = lap_on ? now_ms() : 0.0;
    int rc = movba_lba_upload(h, desc);
    const double t_s1 = lap_on ? now_ms() : 0.0;
    if (rc != MOVBA_OK) { res->status = rc; return rc; }
    h->export_hint = true;
    {
        const DevWindow &w = h->win;
        void *const arr[3] = { res->poses, res->points, res->chi2 };
        const size_t nb[3] = { sizeof(double) * 7 * (size_t)w.NP, sizeof(double) * 3 * (size_t)w.P, sizeof(double) * (size_t)w.E };
        for (int k = 0; k < 3; ++k) {
            h->user_dst[k] = host_block_view(arr[k], nb[k]); h->user_host[k] = h->user_dst[k] ? arr[k] : nullptr;
            h->exported[k] = arr[k] != nullptr;          // (an array the caller does not ask for does not cross the bus)
        }
    }
    rc = movba_lba_run(h);
    h->export_hint = false;
    for (int k = 0; k < 3; ++k) h->user_dst[k] = nullptr;      // (user_host stays for the download below)
    if (rc < 0) { res->status = rc; h->stop = nullptr; return rc; }
    const double t_s2 = lap_on ? now_ms() : 0.0;
    rc = movba_lba_download(h, res);
    if (lap_on) std::fprintf(stderr, "libmovba[solve]: upload %.3f  run %.3f  download %.3f ms\n", t_s1 - t_s0, t_s2 - t_s1, now_ms() - t_s2);
    h->stop = nullptr;      // keep no caller pointer after the call returns
    res->status = rc;
    return rc;
}

int movba_lba_export_poses_device(movba_handle *h, void *dst, int64_t cap)
{
    if (!h || !dst) return MOVBA_ERR_ARG;
    if (!h->uploaded || !h->ran || h->early_status != MOVBA_OK || h->run_status != MOVBA_OK) return MOVBA_ERR_STATE;
    const DevWindow &w = h->win;
    const size_t nb = sizeof(double) * 7 * (size_t)w.NP;
    if (cap < (int64_t)nb) return MOVBA_ERR_ARG;
    HIP_TRY(hipMemcpyAsync(dst, w.st[h->ctrl_host->cur].pose, nb, hipMemcpyDeviceToDevice, h->stream));
    return MOVBA_OK;
}

int movba_lba_set_pose_export(movba_handle *h, void *dst, int64_t cap)
{
    if (!h || (dst && cap <= 0)) return MOVBA_ERR_ARG;
    h->pose_export = static_cast<double *>(dst);
    h->pose_export_cap = dst ? cap : 0;
    return MOVBA_OK;
}

void *movba_host_alloc(size_t bytes)
{
    if (bytes == 0) return nullptr;
    void *p = nullptr, *d = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(p); return nullptr; }
    std::lock_guard<std::mutex> lk(g_host_blocks_mu);
    g_host_blocks.push_back(HostBlock{ static_cast<char *>(p), bytes, static_cast<char *>(d) });
    return p;
}

void movba_host_free(void *p)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_host_blocks_mu);
        for (size_t k = 0; k < g_host_blocks.size(); ++k)
            if (g_host_blocks[k].host == p) { g_host_blocks.erase(g_host_blocks.begin() + (long)k); break; }
    }
    (void)hipDeviceSynchronize();           // a kernel still writing results into the block
    (void)hipHostFree(p);
}

int movba_get_profile(movba_handle *h, movba_profile *out)
{
    if (!h || !out) return MOVBA_ERR_ARG;
    *out = h->prof;
    return MOVBA_OK;
}

int movba_set_profile_mask(movba_handle *h, int32_t mask)
{
    if (!h) return MOVBA_ERR_ARG;
    h->opt.profile = mask;
    return MOVBA_OK;
}

int movba_reset_profile(movba_handle *h)
{
    if (!h) return MOVBA_ERR_ARG;
    for (int k = 0; k < MOVBA_NKERNELS; ++k) { h->prof.ms[k] = 0; h->prof.launches[k] = 0; }
    h->prof.upload_ms = h->prof.structure_ms = h->prof.download_ms = 0;
    return MOVBA_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------
// Optimizer::PoseOptimization (/root/reference/src/Optimizer.cc:397-459)
// ---------------------------------------------------------------------------------------
// minimal samples of the hypothesis stage: n_hyp triples of distinct match indices from a xorshift32 stream (the same
// function feeds the oracle in the tests, so both sides score the same hypotheses)
extern "C" int movba_pose_ransac_samples(int32_t n, int32_t n_hyp, uint32_t seed, int32_t *out)
{
    if (n < 3 || n_hyp < 0 || !out) return MOVBA_ERR_ARG;
    uint32_t x = seed ? seed : 0x9E3779B9u;
    auto next = [&]() { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; };
    for (int h = 0; h < n_hyp; ++h) {
        int32_t a = (int32_t)(next() % (uint32_t)n), b, c;
        do { b = (int32_t)(next() % (uint32_t)n); } while (b == a);
        do { c = (int32_t)(next() % (uint32_t)n); } while (c == a || c == b);
        out[3 * h] = a; out[3 * h + 1] = b; out[3 * h + 2] = c;
    }
    return MOVBA_OK;
}

extern "C" int movba_pose_opt(movba_handle *h, const movba_pose_desc *d, movba_pose_result *res)
{
    if (!h || !d || !res) return MOVBA_ERR_ARG;
    res->status = MOVBA_ERR_ARG; res->n_inliers = 0; res->ransac_inliers = 0; res->lm_iters = 0;
    res->ransac_samples_used = 0; res->lo_accepted = 0; res->lo_inliers = 0; res->pad_q = 0;
    const int n = d->n;
    const int n_hyp = d->ransac_iters > 0 ? std::min(d->ransac_iters, (int32_t)MOVBA_MAX_RANSAC_ITERS) : 0;
    if (n < 0 || (n && (!d->Xw || !d->obs)) || d->rounds < 1 || d->its_per_round < 1) return MOVBA_ERR_ARG;
    for (int k = 0; k < 7; ++k) res->pose[k] = d->pose0[k];
    // fewer than 4 matches: the reference returns 0 without touching the frame (Optimizer.cc:415-418)
    if (n < 4) { res->status = MOVBA_EMPTY; return MOVBA_EMPTY; }
    HIP_TRY(hipSetDevice(h->device));
    Carver c;
    const size_t o_X = c.take<double>(3 * (size_t)n), o_obs = c.take<double>(2 * (size_t)n), o_is = c.take<double>(n);
    const size_t o_samp = c.take<int32_t>(3 * (size_t)n_hyp + 1);
    const size_t h2d = c.off;
    const size_t o_chi = c.take<double>(n), o_pose = c.take<double>(24), o_lvl = c.take<uint8_t>(n);
    const size_t d2h_end = c.off;
    const size_t o_cand = c.take<uint8_t>(pose_ransac_bytes(n_hyp) + 16);
    const size_t total = c.off;
    // The hypothesis stage runs as a grid of its own over the whole chip (k_pose_hyp, one workgroup per sample) on a device
    // copy of the matches; the LM kernel then only picks the best candidate.  The LM keeps the matches in LDS when they fit
    // (staged), reading them once from the device copy (hypothesis stage on) or straight from the pinned buffer (off), and
    // writes its results back into the pinned buffer itself.
    const bool grid_hyp = n_hyp > 0;
    const bool staged = pose_opt_staged_lds_bytes(n, 0) <= 144 * 1024;
    const bool need_arena = !staged || grid_hyp;
    if (need_arena && total > h->pose_cap) {
        if (h->pose_arena) { HIP_TRY(hipStreamSynchronize(h->stream)); HIP_TRY(hipFree(h->pose_arena)); h->pose_arena = nullptr; h->pose_cap = 0; }
        const size_t cap = align_up(2 * total, 1 << 16);
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->pose_arena), cap));
        h->pose_cap = cap;
    }
    int rc = ensure_stage(h, total); if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    // (a window uploaded on this handle and not run yet: its arrays may still be crossing the bus out of the staging buffer)
    HIP_TRY(hipEventSynchronize(h->copy_event));
    h->export_in_run = false;        // (results a run may have left in the staging buffer are overwritten here: download exports again)
    char *sg = h->stage;
    std::memcpy(sg + o_X, d->Xw, sizeof(double) * 3 * (size_t)n);
    std::memcpy(sg + o_obs, d->obs, sizeof(double) * 2 * (size_t)n);
    double *isg = reinterpret_cast<double *>(sg + o_is);
    for (int i = 0; i < n; ++i) isg[i] = d->inv_sigma2 ? d->inv_sigma2[i] : 1.0;
    if (n_hyp > 0) (void)movba_pose_ransac_samples(n, n_hyp, d->ransac_seed, reinterpret_cast<int32_t *>(sg + o_samp));
    if (need_arena) HIP_TRY(hipMemcpyAsync(h->pose_arena, sg, h2d, hipMemcpyHostToDevice, h->stream));
    PoseDev p{};
    p.n = n; p.rounds = d->rounds; p.its = d->its_per_round; p.n_hyp = n_hyp; p.hyp_done = 0;
    p.confidence = d->confidence; p.lo_its = n_hyp > 0 && d->lo_iters > 0 ? d->lo_iters : 0;
    p.fx = d->fx; p.fy = d->fy; p.cx = d->cx; p.cy = d->cy; p.huber_delta = d->huber_delta; p.chi2_gate = d->chi2_gate;
    for (int k = 0; k < 7; ++k) p.pose0[k] = d->pose0[k];
    char *in = need_arena ? h->pose_arena : h->stage_dev;          // where the kernels read the matches
    char *out = staged ? h->stage_dev : h->pose_arena;              // where the LM kernel leaves its results
    p.Xw = reinterpret_cast<double *>(in + o_X); p.obs = reinterpret_cast<double *>(in + o_obs); p.isig = reinterpret_cast<double *>(in + o_is);
    p.samples = reinterpret_cast<const int32_t *>(in + o_samp);
    p.chi2 = reinterpret_cast<double *>(out + o_chi); p.pose_out = reinterpret_cast<double *>(out + o_pose); p.level1 = reinterpret_cast<uint8_t *>(out + o_lvl);
    p.cand = need_arena ? reinterpret_cast<double *>(h->pose_arena + o_cand) : nullptr;
    if (grid_hyp) {
        HIP_TRY(launch_pose_hyp(p, h->stream));
        p.hyp_done = 1;
    }
    HIP_TRY(launch_pose_opt(p, staged, h->stream));
    if (!staged) HIP_TRY(hipMemcpyAsync(sg + o_chi, h->pose_arena + o_chi, d2h_end - o_chi, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const double *po = reinterpret_cast<const double *>(sg + o_pose);
    for (int k = 0; k < 7; ++k) res->pose[k] = po[k];
    res->n_inliers = (int32_t)po[7];
    res->ransac_inliers = n_hyp > 0 ? (int32_t)po[8] : 0;
    res->lm_iters = (int32_t)po[16];
#ifdef MOVBA_CLOCK_STAMP
    std::fprintf(stderr, "libmovba[stamp]: k_pose_opt, cycles per LM iteration (%d): system pass %.0f, reduction of 28 %.0f, solve + update %.0f, cost pass + reduction %.0f\n",
                 res->lm_iters, po[20] / res->lm_iters, po[21] / res->lm_iters, po[22] / res->lm_iters, po[23] / res->lm_iters);
#endif
    res->ransac_samples_used = n_hyp > 0 ? (int32_t)po[17] : 0; res->lo_accepted = n_hyp > 0 ? (int32_t)po[18] : 0; res->lo_inliers = n_hyp > 0 ? (int32_t)po[19] : 0;
    for (int k = 0; k < 7; ++k) res->ransac_pose[k] = n_hyp > 0 ? po[9 + k] : d->pose0[k];
    if (res->outlier) std::memcpy(res->outlier, sg + o_lvl, (size_t)n);
    if (res->chi2) std::memcpy(res->chi2, sg + o_chi, sizeof(double) * (size_t)n);
    res->status = MOVBA_OK;
    return MOVBA_OK;
}
