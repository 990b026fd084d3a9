// The "device" of the CPU-only concurrency tests: every launch_* wrapper of mov-slam_amd/csrc/kernels.h as a closure on the
// fake stream (fake_hip.cpp).  The closures READ what the real kernels read and WRITE what they write (checksums and
// stand-in values, no bundle adjustment), follow the LM controller's protocol with the host (Ctrl::done, the progress word,
// the park counter, the stop flag) and can be told to misbehave (fake_set_mode): park the solve for the direct solver at a
// given trial, or stop making progress (a hung device) — so that ThreadSanitizer and the tests see the host's state machine
// under every hand-off it has.  Test infrastructure only.
#include <hip/hip_runtime.h>

#include <atomic>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "device_types.h"
#include "kernels.h"
#include "pose_kernels.h"

namespace {
std::atomic<int> g_park_trial{-1}, g_stall_after{-1};
std::atomic<long> g_sink{0};

long sum_bytes(const void *p, size_t n)
{
    const unsigned char *c = static_cast<const unsigned char *>(p);
    long s = 0;
    for (size_t k = 0; k < n; k += 64) s += c[k];
    if (n) s += c[n - 1];
    return s;
}
}  // namespace

extern "C" void fake_set_mode(int park_trial, int stall_after)
{
    g_park_trial.store(park_trial); g_stall_after.store(stall_after);
}

namespace movba {

hipError_t configure_kernels(int) { return hipSuccess; }
hipError_t configure_pcg_rows() { return hipSuccess; }
hipError_t configure_struct_kernels() { return hipSuccess; }
hipError_t configure_dense_kernels() { return hipSuccess; }
hipError_t configure_dense_persist() { return hipSuccess; }
hipError_t configure_pose_kernels() { return hipSuccess; }
bool struct_lds_fits(int, int) { return true; }
size_t point_lds_need(int NP, int) { return (size_t)NP * 96; }
size_t point_lds_bytes_for(const DevWindow &w, bool, bool) { return (size_t)w.NP * 96; }
int schur_blocks(const DevWindow &w) { return 8 * (w.sched_per_xcd / kSchurWaves); }
size_t dense_tiles_doubles(int nfree) { const size_t nt = ((size_t)6 * nfree + kDenseNB - 1) / kDenseNB; return (nt + 1) * (nt + 2) / 2 * (size_t)(kDenseNB * kDenseNB); }
int dense_ntile(int nfree) { return (6 * nfree + kDenseNB - 1) / kDenseNB; }
bool dense_persist_supported(const DensePlan &p) { return p.ok; }
size_t pose_ransac_bytes(int n_hyp) { return (size_t)n_hyp * 64; }
size_t pose_opt_staged_lds_bytes(int n, int) { return (size_t)n * 48; }

// ---- structure pass: the pair counts for real (the host lays the window out from them), the rest reads its inputs ----
hipError_t launch_struct_count(const StructDev &sd, hipStream_t s)
{
    fake_enqueue(s, [sd] {
        const int nf = sd.nfree;
        std::vector<int> hs;
        for (int l = 0; l < sd.P; ++l) {
            hs.clear();
            for (int e = sd.pt_start[l]; e < sd.pt_start[l + 1]; ++e) { const int h = sd.hidx[sd.g_pose[e]]; if (h >= 0) hs.push_back(h); }
            for (size_t a = 0; a < hs.size(); ++a)
                for (size_t b = a; b < hs.size(); ++b) {
                    if (a != b && hs[a] == hs[b]) { *sd.error = 1; continue; }
                    const int lo = hs[a] < hs[b] ? hs[a] : hs[b], hi = hs[a] < hs[b] ? hs[b] : hs[a];
                    sd.cnt[lo * nf + hi] += 1;
                }
        }
    });
    return hipSuccess;
}
// (the sort-based pass of struct_sort.hip: the same counts, plus the couples per point; the fill touches its inputs)
hipError_t launch_couple_count(const StructDev &sd, int32_t *cnt_pt, hipStream_t s)
{
    fake_enqueue(s, [sd, cnt_pt] {
        const int nf = sd.nfree;
        std::vector<int> hs;
        for (int l = 0; l < sd.P; ++l) {
            hs.clear();
            for (int e = sd.pt_start[l]; e < sd.pt_start[l + 1]; ++e) { const int h = sd.hidx[sd.g_pose[e]]; if (h >= 0) hs.push_back(h); }
            for (size_t a = 0; a < hs.size(); ++a)
                for (size_t b = a; b < hs.size(); ++b) {
                    if (a != b && hs[a] == hs[b]) { *sd.error = 1; continue; }
                    const int lo = hs[a] < hs[b] ? hs[a] : hs[b], hi = hs[a] < hs[b] ? hs[b] : hs[a];
                    sd.cnt[(size_t)lo * nf + hi] += 1;
                }
            cnt_pt[l] = (int32_t)(hs.size() * (hs.size() - (hs.empty() ? 0 : 1)) / 2);
        }
    });
    return hipSuccess;
}
size_t sorted_fill_temp_bytes(int P, long long noff, int) { return (size_t)P * 4 + (size_t)noff * 16 + 256; }
hipError_t launch_sorted_fill(const StructDev &sd, const int32_t *cnt_pt, int32_t *off, unsigned *keys_in, unsigned *keys_out,
                              unsigned long long *vals_in, void *tmp, size_t tmp_bytes, long long noff, hipStream_t s)
{
    fake_enqueue(s, [=] {
        if (noff <= 0) return;
        const int E = sd.pt_start[sd.P];
        g_sink += sum_bytes(sd.slot, sizeof(int32_t) * E) + sum_bytes(sd.g_pose, sizeof(int32_t) * E) + sum_bytes(sd.hidx, sizeof(int32_t) * sd.NP) + sum_bytes(cnt_pt, 4 * (size_t)sd.P);
        off[0] = 0; keys_in[noff - 1] = 0; keys_out[noff - 1] = 0; vals_in[noff - 1] = 0;
        if (tmp_bytes) static_cast<char *>(tmp)[tmp_bytes - 1] = 0;
        sd.ent64[0] = 1; sd.ent64[noff - 1] = 1;
    });
    return hipSuccess;
}

hipError_t launch_struct_counts_out(const StructDev &sd, int32_t *host_cnt, int seq, hipStream_t s, const int32_t *basic_pe, const int32_t *basic_info)
{
    fake_enqueue(s, [sd, host_cnt, seq, basic_pe, basic_info] {
        const int nbins = sd.nfree * sd.nfree;
        if (host_cnt) {
            for (int b = 0; b < nbins; ++b) host_cnt[b] = sd.cnt[b];
            host_cnt[nbins] = *sd.error;
            if (basic_pe) {
                for (int k = 0; k < kBasicInfo; ++k) host_cnt[nbins + 2 + k] = basic_info[k];
                for (int k = 0; k < sd.NP; ++k) host_cnt[nbins + 2 + kBasicInfo + k] = basic_pe[k];
            }
            __atomic_store_n(host_cnt + nbins + 1, seq, __ATOMIC_RELEASE);
        }
        int run = 0;
        for (int b = 0; b < nbins; ++b) { sd.ent0[b] = run; if (b / sd.nfree < b % sd.nfree) run += sd.cnt[b]; }
    });
    return hipSuccess;
}
hipError_t launch_struct_scan(const StructDev &sd, hipStream_t s)
{
    fake_enqueue(s, [sd] { g_sink += sum_bytes(sd.cnt, sizeof(int32_t) * sd.nfree * sd.nfree); sd.cntw[0] = 0; });
    return hipSuccess;
}
hipError_t launch_struct_fill(const StructDev &sd, hipStream_t s)
{
    fake_enqueue(s, [sd] {
        const int E = sd.pt_start[sd.P];
        g_sink += sum_bytes(sd.slot, sizeof(int32_t) * E) + sum_bytes(sd.g_pose, sizeof(int32_t) * E) + sum_bytes(sd.hidx, sizeof(int32_t) * sd.NP) + sum_bytes(sd.ent0, 4);
        if (sd.ent64) sd.ent64[0] = 1; else sd.ent_i[0] = 1;
    });
    return hipSuccess;
}
hipError_t launch_slot_point(int32_t *slot, const int32_t *g_pose, const int32_t *base, const int32_t *g_point, int32_t *slot_point, int E, const int32_t *hx, int NP, hipStream_t s)
{
    fake_enqueue(s, [=] {
        for (int g = 0; g < E; ++g) {
            int sl = slot[g];
            if (base) {
                const int b = base[g_pose[g]];
                if (hx && b >= 0) sl += hx[(size_t)(g / kBasicBlock) * NP + g_pose[g]];
                sl = b >= 0 ? b + sl : -1; slot[g] = sl;
            }
            if (sl >= 0) slot_point[sl] = g_point[g];
        }
    });
    return hipSuccess;
}
hipError_t launch_ingest(const IngestArgs &a, hipStream_t s)
{
    fake_enqueue(s, [a] {
        if (a.zero_words) std::memset(a.zero, 0, 4 * (size_t)a.zero_words);
        for (int q = 0; q < a.nseg; ++q) std::memcpy(a.seg[q].dst, a.seg[q].src, a.seg[q].bytes);
        if (a.counter) __atomic_fetch_add(a.counter, (unsigned)ingest_workgroups(), __ATOMIC_RELAXED);
    });
    return hipSuccess;
}
int ingest_workgroups() { return 64; }
// the grouping pass of the device (k_basic_hist, k_basic_index), word for word what the kernels leave behind
hipError_t launch_basic(const BasicDev &bd, hipStream_t s)
{
    fake_enqueue(s, [bd] {
        std::vector<int> seen((size_t)bd.NP);
        for (int b = 0; b < bd.nblk; ++b) {
            std::fill(seen.begin(), seen.end(), 0);
            for (int e = b * kBasicBlock; e < std::min(bd.E, (b + 1) * kBasicBlock); ++e) {
                const int kf = bd.edge_pose[e], l = bd.edge_point[e], lp = e > 0 ? bd.edge_point[e - 1] : -1;
                if ((unsigned)kf >= (unsigned)bd.NP || (unsigned)l >= (unsigned)bd.P) { bd.info[0] = 1; bd.rank[e] = 0; continue; }
                if (l < lp) bd.info[1] = 1; else for (int q = std::max(lp, -1) + 1; q <= l; ++q) bd.pt_start[q] = e;
                if (e == bd.E - 1) for (int q = l + 1; q <= bd.P; ++q) bd.pt_start[q] = bd.E;
                bd.rank[e] = seen[kf]++;
            }
            for (int k = 0; k < bd.NP; ++k) { bd.H[(size_t)b * bd.NP + k] = seen[k]; bd.pose_edges[k] += seen[k]; }
        }
        for (int k = 0; k < bd.NP; ++k) { int run = 0; for (int b = 0; b < bd.nblk; ++b) { const int v = bd.H[(size_t)b * bd.NP + k]; bd.H[(size_t)b * bd.NP + k] = run; run += v; } }
        int nf = 0, run = 0, nfix = 0;
        for (int i = 0; i < bd.NP; ++i) {
            bd.hidx[i] = -1; bd.base[i] = -1;
            if (bd.pose_fixed[i]) { ++nfix; continue; }
            if (bd.pose_edges[i] > 0) { bd.hidx[i] = nf; bd.free_pose[nf++] = i; bd.base[i] = run; run += bd.pose_edges[i]; }
        }
        bd.base[bd.NP] = -1; bd.info[2] = nf; bd.info[3] = run; bd.info[4] = nfix;
    });
    return hipSuccess;
}

// ---- the LM controller's protocol ----
namespace {
int rd_done(const Ctrl *c) { return __atomic_load_n(&c->done, __ATOMIC_ACQUIRE); }
void wr_done(Ctrl *c, int v) { __atomic_store_n(&c->done, v, __ATOMIC_RELEASE); }

void fake_init(const DevWindow &w)
{
    // reads everything the upload sent (the caller's arrays crossed on the copy stream: the solve waits for their event)
    g_sink += sum_bytes(w.pose0, 56 * (size_t)w.NP) + sum_bytes(w.point0, 24 * (size_t)w.P) + sum_bytes(w.obs, 16 * (size_t)w.E) + sum_bytes(w.isig, 8 * (size_t)w.E) +
              sum_bytes(w.g_pose, 4 * (size_t)w.E) + sum_bytes(w.g_point, 4 * (size_t)w.E) + sum_bytes(w.slot, 4 * (size_t)w.E) + sum_bytes(w.items, sizeof(Item) * (size_t)w.nitems) +
              sum_bytes(w.sched, 32) + sum_bytes(w.row_ptr, 4 * ((size_t)w.nfree + 1)) + sum_bytes(w.hidx, 4 * (size_t)w.NP);
    if (w.obs_r) g_sink += sum_bytes(w.obs_r, 8 * (size_t)w.E);
    if (w.dense.G > 0) g_sink += sum_bytes(w.dense.tasks, 32) + sum_bytes(w.dense.task_ptr, 4 * ((size_t)w.dense.G + 1));
    std::memcpy(w.st[0].pose, w.pose0, 56 * (size_t)w.NP);
    std::memcpy(w.st[0].point, w.point0, 24 * (size_t)w.P);
    std::memset(w.dec_rec, 0, sizeof(unsigned) * 8 * (size_t)w.n_pt_blocks);
    Ctrl *c = w.ctrl;
    std::memset(c, 0, sizeof(Ctrl));
    c->nu = 2.0; wr_done(c, (w.max_iters <= 0) ? 1 : 0);
}
void fake_decide(const DevWindow &w)
{
    Ctrl *c = w.ctrl;
    if (__atomic_load_n(&c->done, __ATOMIC_ACQUIRE)) return;
    const int stall = g_stall_after.load();
    if (stall >= 0 && c->n_solves >= stall) return;              // a hung device: no progress, no completion
    c->n_solves += 1; c->it += 1; c->iters_done = c->it;
    if (c->n_trace < kMaxTrace) { c->tr_accept[c->n_trace] = 1; c->tr_pcg[c->n_trace] = c->pcg_last_iters; c->n_trace += 1; }
    const bool fin = c->it >= w.max_iters || __atomic_load_n(&w.hstat->stop, __ATOMIC_ACQUIRE);
    if (fin) __atomic_store_n(&c->done, 1, __ATOMIC_RELEASE);
    __atomic_store_n(&w.hstat->progress, HostStatus::pack(c->n_solves, c->it, fin), __ATOMIC_RELEASE);
}
void fake_schur(const DevWindow &w)
{
    g_sink += (w.ent64 ? (long)w.ent64[0] : (long)w.ent_i[0]) + w.slot_point[0];
}

// a reduced solve that hands the solve over in the trial the driver asked for: the PCG that gives up, the banded factorisation
// that meets a non-positive pivot
bool fake_park(const DevWindow &w)
{
    Ctrl *c = w.ctrl;
    if (g_park_trial.load() >= 0 && c->n_solves >= g_park_trial.load() && c->solver_mode == 0) {
        c->solver_mode = 1; c->direct_from = c->n_solves; c->n_pause += 1; wr_done(c, 2);
        __atomic_store_n(&w.hstat->pause_seq, c->n_pause, __ATOMIC_RELEASE);
        return true;
    }
    return false;
}
void fake_pcg(const DevWindow &w)
{
    Ctrl *c = w.ctrl;
    if (rd_done(c)) return;
    if (fake_park(w)) return;
    c->pcg_last_iters = 7; c->pcg_total_iters += 7;
}

// the back-substitution pass and, in its last workgroup, the LM decision
void fake_backsub(const DevWindow &w)
{
    if (rd_done(w.ctrl)) return;
    g_sink += w.dec_rec[0];
    fake_decide(w);
}
// What the one-launch direct solver (dense_persist.hip) indexes with values the HOST built - the item-range table prange read
// by assemble_tile, the work items' partials behind it, every task's tile / operand tiles / flags, the tagged-record area - walked
// here with the same index arithmetic, every index checked against the size the upload carved and the last element of every
// range touched (AddressSanitizer sees the rest).  An assembly variant of round 3 faulted on the GPU and its cause was never
// found: whatever such a variant gets wrong on the device, a table that points outside its arrays is caught here on the CPU.
long walk_direct_tables(const DevWindow &w)
{
    long sink = 0;
    const int nf = w.nfree, nt = w.dense.ntile, NB = kDenseNB;
    auto need = [](bool ok, const char *what) { if (!ok) { std::fprintf(stderr, "fake device: direct-solver table out of range: %s\n", what); std::abort(); } };
    need(nt == (6 * nf + NB - 1) / NB && w.dense.n == 6 * nf, "tile count");
    for (int lo = 0; lo < nf; ++lo)
        for (int hi = lo; hi < nf; ++hi) {
            const int32_t i0 = w.dense.prange[2 * ((size_t)lo * nf + hi)], i1 = w.dense.prange[2 * ((size_t)lo * nf + hi) + 1];
            need(0 <= i0 && i0 <= i1 && i1 <= w.nitems, "prange");
            need(lo != hi || i1 > i0, "a diagonal pair without work items");
            if (i1 > i0) sink += (long)w.part[(size_t)(i1 - 1) * kPartStride + kPartStride - 1];
        }
    // the staging pass of a diagonal tile reads the records of its eight diagonal pairs as ONE run: they must be contiguous
    for (int K = 0; K < nt; ++K) {
        const int b0 = K * (NB / 6), b1 = std::min(b0 + NB / 6, nf);
        for (int b = b0; b + 1 < b1; ++b)
            need(w.dense.prange[2 * ((size_t)b * nf + b) + 1] == w.dense.prange[2 * ((size_t)(b + 1) * nf + b + 1)], "diagonal pairs' items not contiguous");
    }
    const size_t tile_doubles = (size_t)NB * NB, ntiles = (size_t)(nt + 1) * (nt + 2) / 2;
    const int nflag = dense_flag_count(nt);
    auto tile_last = [&](int I, int K) { need(0 <= K && K <= I && I <= nt, "tile index"); const size_t t = (size_t)I * (I + 1) / 2 + K; need(t < ntiles, "tile offset"); return (long)w.dense.tiles[t * tile_doubles + tile_doubles - 1]; };
    need(w.dense.task_ptr[0] == 0, "task list start");
    for (int g = 0; g < w.dense.G; ++g) {
        need(w.dense.task_ptr[g] <= w.dense.task_ptr[g + 1], "task list order");
        for (int t = w.dense.task_ptr[g]; t < w.dense.task_ptr[g + 1]; ++t) {
            const DenseTask &tk = w.dense.tasks[t];
            need(tk.op >= DT_ASM && tk.op <= DT_COL, "task op");
            need(tk.slot >= -1 && tk.slot < w.dense.slots, "task slot");
            if (tk.op == DT_EPI) continue;
            need(tk.I >= 0 && tk.I < nt && tk.K >= 0 && tk.K <= tk.I, "task tile");
            sink += tile_last(tk.I, tk.K);
            if (tk.op == DT_UPD || tk.op == DT_UPD2 || tk.op == DT_RUP) {
                need(tk.k >= 0 && tk.k < tk.K + (tk.op == DT_RUP ? 1 : 0), "task block column");
                sink += tile_last(tk.I, tk.k) + tile_last(tk.K, tk.k);
                need(dense_flag_F(nt, tk.I, tk.k) < nflag && dense_flag_F(nt, tk.K, tk.k) < nflag, "flag index");
            }
            need(dense_flag_FC(nt, tk.I, tk.K) < nflag && dense_flag_PD(nt, tk.K) < nflag && dense_flag_FX(nt, tk.I) < nflag, "flag index");
        }
    }
    sink += (long)w.dense.flags[dense_flag_words(nt) - 1] + (long)w.dense.contrib[(size_t)nt * nt * NB - 1] + (long)w.dense.failw[1];
    return sink;
}

void fake_direct(const DevWindow &w)
{
    Ctrl *c = w.ctrl;
    if (rd_done(c) == 1) return;
    if (w.dense.G > 0) g_sink += sum_bytes(w.dense.flags, 16) + walk_direct_tables(w);
    c->pcg_last_iters = -1; c->n_direct += 1;
    if (rd_done(c) == 2) wr_done(c, 0);
}
void fake_finalize(const DevWindow &w)
{
    for (int e = 0; e < w.E; ++e) { w.out_chi2[e] = 1.0; w.out_outlier[e] = 0; }
    if (w.pose_export) std::memcpy(w.pose_export, w.st[0].pose, 56 * (size_t)w.NP);
    std::memcpy(w.ctrl_out, w.ctrl, sizeof(Ctrl));
    if (__atomic_load_n(&w.ctrl->done, __ATOMIC_ACQUIRE) == 1) __atomic_store_n(&w.hstat->progress, HostStatus::pack(w.ctrl->n_solves, w.ctrl->it, 1), __ATOMIC_RELEASE);
}
}  // namespace

hipError_t launch_init(const DevWindow &w, hipStream_t s) { fake_enqueue(s, [w] { fake_init(w); }); return hipSuccess; }
hipError_t launch_linearize(const DevWindow &w, hipStream_t s) { fake_enqueue(s, [w] { g_sink += rd_done(w.ctrl); }); return hipSuccess; }
hipError_t launch_schur(const DevWindow &w, int, hipStream_t s) { fake_enqueue(s, [w] { fake_schur(w); }); return hipSuccess; }
hipError_t launch_lambda_init(const DevWindow &w, hipStream_t s) { fake_enqueue(s, [w] { w.ctrl->solver_mode = w.direct_only ? 1 : 0; w.ctrl->direct_from = w.direct_only ? 0 : -1; w.ctrl->lambda = 1e-3; }); return hipSuccess; }
// (the back-substitution pass takes the LM decision in its last workgroup: one launch)
hipError_t launch_backsub(const DevWindow &w, hipStream_t s) { fake_enqueue(s, [w] { fake_backsub(w); }); return hipSuccess; }
hipError_t launch_finalize(const DevWindow &w, hipStream_t s) { fake_enqueue(s, [w] { fake_finalize(w); }); return hipSuccess; }
hipError_t launch_export(const DevWindow &w, const ExportDst &d, hipStream_t s)
{
    fake_enqueue(s, [w, d] {
        if (d.poses) std::memcpy(d.poses, w.st[0].pose, 56 * (size_t)w.NP);
        if (d.points) std::memcpy(d.points, w.st[0].point, 24 * (size_t)w.P);
        if (d.chi2) std::memcpy(d.chi2, w.out_chi2, 8 * (size_t)w.E);
        std::memcpy(d.outlier, w.out_outlier, (size_t)w.E);
    });
    return hipSuccess;
}
hipError_t launch_pcg_rows(const DevWindow &w, int, const PcgParams &, int, hipStream_t s) { fake_enqueue(s, [w] { fake_pcg(w); }); return hipSuccess; }
hipError_t launch_dense_solve(const DevWindow &w, hipStream_t s) { fake_enqueue(s, [w] { fake_direct(w); }); return hipSuccess; }
hipError_t launch_dense_persist(const DevWindow &w, unsigned, hipStream_t s) { fake_enqueue(s, [w] { fake_direct(w); }); return hipSuccess; }

// ---- batched launches: the same closures over the windows of the batch (the device arrays are read where the host put them) ----
hipError_t launch_init_batch(const BatchDev &b, int, hipStream_t s) { fake_enqueue(s, [b] { for (int i = 0; i < b.n; ++i) fake_init(b.wins[i]); }); return hipSuccess; }
hipError_t launch_point_batch(const BatchDev &b, int, bool backsub, bool, bool, size_t, hipStream_t s)
{
    fake_enqueue(s, [b, backsub] { for (int i = 0; i < b.n; ++i) { g_sink += b.blk_point[i]; if (backsub) fake_backsub(b.wins[i]); } });
    return hipSuccess;
}
hipError_t launch_schur_batch(const BatchDev &b, int, int, bool, hipStream_t s) { fake_enqueue(s, [b] { for (int i = 0; i < b.n; ++i) { g_sink += b.blk_schur[i]; fake_schur(b.wins[i]); } }); return hipSuccess; }
hipError_t launch_lambda_init_batch(const BatchDev &b, hipStream_t s) { fake_enqueue(s, [b] { for (int i = 0; i < b.n; ++i) b.wins[i].ctrl->lambda = 1e-3; }); return hipSuccess; }
hipError_t launch_finalize_batch(const BatchDev &b, int, hipStream_t s) { fake_enqueue(s, [b] { for (int i = 0; i < b.n; ++i) fake_finalize(b.wins[i]); }); return hipSuccess; }
// the banded factorisation: an exact solve in one launch - nothing parks, nothing waits
static void fake_band(const DevWindow &w) { Ctrl *c = w.ctrl; if (rd_done(c)) return; if (fake_park(w)) return; c->pcg_last_iters = -2; c->n_band += 1; }
hipError_t launch_band(const DevWindow &w, int, hipStream_t s) { fake_enqueue(s, [w] { fake_band(w); }); return hipSuccess; }
hipError_t launch_band_batch(const BatchDev &b, size_t, hipStream_t s) { fake_enqueue(s, [b] { for (int i = 0; i < b.n; ++i) if (b.band_bw[i] >= 0) fake_band(b.wins[i]); }); return hipSuccess; }
hipError_t configure_band() { return hipSuccess; }
hipError_t launch_pcg_rows_batch(const BatchDev &b, bool, bool, size_t, int, hipStream_t s) { fake_enqueue(s, [b] { for (int i = 0; i < b.n; ++i) { if (b.band_bw && b.band_bw[i] >= 0) continue; g_sink += b.pps[i].max_iters; fake_pcg(b.wins[i]); } }); return hipSuccess; }

// ---- pose-only optimisation: echoes the start pose, every match an inlier ----
hipError_t launch_pose_hyp(const PoseDev &p, hipStream_t s) { fake_enqueue(s, [p] { g_sink += sum_bytes(p.Xw, 24 * (size_t)p.n) + sum_bytes(p.samples, 12 * (size_t)p.n_hyp); }); return hipSuccess; }
hipError_t launch_pose_opt(const PoseDev &p, bool, hipStream_t s)
{
    fake_enqueue(s, [p] {
        g_sink += sum_bytes(p.obs, 16 * (size_t)p.n) + sum_bytes(p.isig, 8 * (size_t)p.n);
        for (int k = 0; k < 7; ++k) { p.pose_out[k] = p.pose0[k]; p.pose_out[9 + k] = p.pose0[k]; }
        p.pose_out[7] = p.n; p.pose_out[8] = p.n; p.pose_out[16] = 1;
        for (int i = 0; i < p.n; ++i) { p.chi2[i] = 0.5; p.level1[i] = 0; }
    });
    return hipSuccess;
}

}  // namespace movba
