// Drives libmovba's HOST side (mov-slam_amd/csrc/api.cpp: upload with its helper thread and copy stream, the LM loop's polling
// of the device's progress, park / resume for the direct solver, batched runs, stop flag, watchdog) against the fake device of
// this directory, from several threads, under ThreadSanitizer.  Exit code 0 and the last line "HOST-TSAN OK" = every scenario
// ran to its expected end; ThreadSanitizer reports (if any) go to stderr and make the exit code non-zero (halt_on_error).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

#include "movba.h"

extern "C" void fake_set_mode(int park_trial, int stall_after);

namespace {

struct Win {
    std::vector<double> poses, points, obs, isig, obs_right;
    std::vector<uint8_t> fixed;
    std::vector<int32_t> ep, el;
    movba_lba_desc d{};
    std::vector<double> out_poses, out_points, out_chi2;
    std::vector<uint8_t> out_outlier;
    movba_lba_result r{};
};

void make(Win &w, int K, int F, int P, unsigned seed, bool shuffle_edges, bool stereo)
{
    std::mt19937 rng(seed);
    const int NP = K + F;
    w.poses.assign(7 * (size_t)NP, 0.0); w.fixed.assign(NP, 0); w.points.assign(3 * (size_t)P, 1.0);
    for (int i = 0; i < NP; ++i) { w.poses[7 * i + 3] = 1.0; w.poses[7 * i + 4] = 0.3 * i; w.fixed[i] = i < F; }
    w.ep.clear(); w.el.clear();
    for (int l = 0; l < P; ++l) {
        const int run = 2 + (int)(rng() % 4), first = (int)(rng() % (unsigned)(NP - run + 1));
        for (int k = first; k < first + run; ++k) { w.ep.push_back(k); w.el.push_back(l); }
    }
    const size_t E = w.ep.size();
    if (shuffle_edges)
        for (size_t e = E - 1; e > 0; --e) { const size_t j = rng() % (e + 1); std::swap(w.ep[e], w.ep[j]); std::swap(w.el[e], w.el[j]); }
    w.obs.assign(2 * E, 100.0); w.isig.assign(E, 1.0);
    w.obs_right.clear();
    if (stereo) { w.obs_right.assign(E, -1.0); for (size_t e = 0; e < E; e += 3) w.obs_right[e] = 90.0; }
    w.d = movba_lba_desc{};
    w.d.n_poses = NP; w.d.n_points = P; w.d.n_edges = (int32_t)E;
    w.d.poses = w.poses.data(); w.d.pose_fixed = w.fixed.data(); w.d.points = w.points.data();
    w.d.edge_pose = w.ep.data(); w.d.edge_point = w.el.data(); w.d.obs = w.obs.data(); w.d.inv_sigma2 = w.isig.data();
    w.d.fx = w.d.fy = 320; w.d.cx = 320; w.d.cy = 240; w.d.huber_delta = 2.236; w.d.chi2_gate = 5.0; w.d.max_iters = 10; w.d.flags = MOVBA_FLAG_STALE_ERROR_QUIRK;
    if (stereo) { w.d.obs_right = w.obs_right.data(); w.d.bf = 40.0; }
    w.out_poses.assign(7 * (size_t)NP, 0.0); w.out_points.assign(3 * (size_t)P, 0.0); w.out_chi2.assign(E, 0.0); w.out_outlier.assign(E, 9);
    w.r = movba_lba_result{};
    w.r.poses = w.out_poses.data(); w.r.points = w.out_points.data(); w.r.chi2 = w.out_chi2.data(); w.r.outlier = w.out_outlier.data();
}

int fails = 0;
#define EXPECT(c) do { if (!(c)) { std::fprintf(stderr, "EXPECT failed at line %d: %s\n", __LINE__, #c); ++fails; } } while (0)

void check_solved(const Win &w, int rc)
{
    EXPECT(rc == MOVBA_OK);
    EXPECT(w.r.n_solves == 10 && w.r.iters_done == 10);
    EXPECT(w.out_poses[3] == 1.0 && w.out_points[0] == 1.0 && w.out_outlier[0] == 0 && w.out_chi2[0] == 1.0);     // what the fake device exported
}

}  // namespace

int main()
{
    setenv("MOVBA_WATCHDOG_MS", "200", 1);
    // ---- 1. two threads, a handle each (they share the device's copy stream), windows of changing size, order and kind ----
    {
        auto worker = [](unsigned seed) {
            movba_handle *h = nullptr;
            EXPECT(movba_create(&h, 0, nullptr, nullptr) == MOVBA_OK);
            Win w;
            for (int it = 0; it < 12; ++it) {
                make(w, 6 + 5 * (it % 5), 2, 300 + 400 * (it % 4), seed + it, it % 4 == 3, it % 3 == 1);
                check_solved(w, movba_lba_solve(h, &w.d, &w.r));
                if (it % 4 == 0) {          // the phased API, a PoseOptimization between upload and run, two runs of one upload
                    EXPECT(movba_lba_upload(h, &w.d) == MOVBA_OK);
                    double X[12] = { 0, 0, 5, 1, 0, 5, 0, 1, 5, 1, 1, 6 }, o[8] = { 1, 2, 3, 4, 5, 6, 7, 8 };
                    movba_pose_desc pd{}; movba_pose_result pr{};
                    pd.n = 4; pd.Xw = X; pd.obs = o; pd.fx = pd.fy = 320; pd.cx = 320; pd.cy = 240; pd.pose0[3] = 1.0; pd.huber_delta = 2.2; pd.chi2_gate = 5.0;
                    pd.rounds = 4; pd.its_per_round = 10; pd.ransac_iters = it ? 8 : 0; pd.ransac_seed = 3;
                    EXPECT(movba_pose_opt(h, &pd, &pr) == MOVBA_OK && pr.n_inliers == 4);
                    EXPECT(movba_lba_run(h) == MOVBA_OK && movba_lba_run(h) == MOVBA_OK);
                    check_solved(w, movba_lba_download(h, &w.r));
                }
            }
            movba_destroy(h);
        };
        std::thread a(worker, 100u), b(worker, 900u);
        a.join(); b.join();
    }
    // ---- 2. a batch of four handles on one stream, while a fifth handle uploads and solves from another thread ----
    {
        hipStream_t st = nullptr;
        (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        movba_handle *hs[4] = {};
        Win ws[4];
        for (int i = 0; i < 4; ++i) {
            EXPECT(movba_create(&hs[i], 0, st, nullptr) == MOVBA_OK);
            make(ws[i], 8 + 3 * i, 2, 500 + 100 * i, 40 + i, false, false);
            EXPECT(movba_lba_upload(hs[i], &ws[i].d) == MOVBA_OK);
        }
        std::thread other([] {
            movba_handle *h = nullptr;
            EXPECT(movba_create(&h, 0, nullptr, nullptr) == MOVBA_OK);
            Win w;
            for (int it = 0; it < 6; ++it) { make(w, 9, 2, 700, 70 + it, false, false); check_solved(w, movba_lba_solve(h, &w.d, &w.r)); }
            movba_destroy(h);
        });
        for (int rep = 0; rep < 3; ++rep) {
            EXPECT(movba_lba_run_batch(hs, 4) == MOVBA_OK);
            for (int i = 0; i < 4; ++i) check_solved(ws[i], movba_lba_download(hs[i], &ws[i].r));
        }
        other.join();
        for (int i = 0; i < 4; ++i) movba_destroy(hs[i]);
        (void)hipStreamDestroy(st);
    }
    // ---- 3. the PCG parks the solve in trial 2: the host queues the direct solver for that trial and stays with it ----
    {
        fake_set_mode(2, -1);
        movba_options opt{};
        opt.solver = 3;                         // (never the banded factorisation: a 12-keyframe window would not reach the PCG)
        movba_handle *h = nullptr;
        EXPECT(movba_create(&h, 0, nullptr, &opt) == MOVBA_OK);
        Win w; make(w, 12, 2, 800, 7, false, false);
        check_solved(w, movba_lba_solve(h, &w.d, &w.r));
        EXPECT(w.r.n_pcg_giveups == 1 && w.r.direct_from == 2 && w.r.n_direct == 8);
        fake_set_mode(-1, -1);
        check_solved(w, movba_lba_solve(h, &w.d, &w.r));
        EXPECT(w.r.n_pcg_giveups == 0 && w.r.n_direct == 0);
        movba_destroy(h);
    }
    // ---- 4. Optimizer::LocalBundleAdjustment's pbStopFlag raised by another thread while the solve runs ----
    {
        movba_handle *h = nullptr;
        EXPECT(movba_create(&h, 0, nullptr, nullptr) == MOVBA_OK);
        Win w; make(w, 30, 2, 4000, 9, false, false);
        w.d.max_iters = 100000;                          // (only the flag ends this solve)
        static volatile uint8_t stop;
        __atomic_store_n(&stop, (uint8_t)0, __ATOMIC_RELAXED);
        w.d.stop = &stop;
        EXPECT(movba_lba_upload(h, &w.d) == MOVBA_OK);     // (uploaded first: the flag must go up DURING the run, not in front of it)
        std::thread raiser([] { std::this_thread::sleep_for(std::chrono::milliseconds(30)); __atomic_store_n(&stop, (uint8_t)1, __ATOMIC_RELAXED); });
        int rc = movba_lba_run(h);
        raiser.join();
        if (rc == MOVBA_OK) rc = movba_lba_download(h, &w.r);
        EXPECT(rc == MOVBA_OK && w.r.n_solves >= 1 && w.r.n_solves < 100000);
        EXPECT(movba_lba_solve(h, &w.d, &w.r) == MOVBA_STOPPED);      // the flag is still up: the reference's early return
        movba_destroy(h);
    }
    // ---- 5. a device that stops making progress: the watchdog (200 ms here) gives the call back with an error, nothing of
    //         the solve is left queued, and the handle serves the next window ----
    {
        movba_handle *h = nullptr;
        EXPECT(movba_create(&h, 0, nullptr, nullptr) == MOVBA_OK);
        Win w; make(w, 10, 2, 600, 11, false, false);
        fake_set_mode(-1, 3);
        const auto t0 = std::chrono::steady_clock::now();
        EXPECT(movba_lba_solve(h, &w.d, &w.r) == MOVBA_ERR_HIP);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        EXPECT(ms > 150.0 && ms < 5000.0);
        EXPECT(movba_lba_run(h) == MOVBA_ERR_STATE);       // the window has to be uploaded again
        fake_set_mode(-1, -1);
        check_solved(w, movba_lba_solve(h, &w.d, &w.r));
        movba_destroy(h);
    }
    // ---- 6. windows beyond the on-chip PCG: the sort-based structure pass (more than 80 free keyframes, grouped edges), the host
    //         pass (ungrouped edges), intrinsics by keyframe; two handles with direct-solver windows from two threads (the gate
    //         that keeps their launches apart switches to its event chain) ----
    {
        auto worker = [](unsigned seed) {
            movba_handle *h = nullptr;
            EXPECT(movba_create(&h, 0, nullptr, nullptr) == MOVBA_OK);
            Win w;
            std::vector<double> cams, bfs;
            for (int it = 0; it < 4; ++it) {
                make(w, 95 + 20 * it, 3, 1500 + 300 * it, seed + it, it == 2, it == 1);
                if (it == 3) {
                    cams.assign(4 * (size_t)w.d.n_poses, 320.0); bfs.assign(w.d.n_poses, 40.0);
                    for (int i = 0; i < w.d.n_poses; i += 2) cams[4 * i] = 400.0;
                    w.d.cam_kf = cams.data(); w.d.bf_kf = bfs.data();
                }
                check_solved(w, movba_lba_solve(h, &w.d, &w.r));
                EXPECT(w.r.n_direct == w.r.n_solves && w.r.n_sync_timeouts == 0);
            }
            movba_destroy(h);
        };
        std::thread a(worker, 300u), b(worker, 700u);
        a.join(); b.join();
    }
    // ---- 7. a banded factorisation that meets a non-positive pivot in trial 3: the solve parks, the host queues the dense direct
    //         solver for that trial and stays with it; the next window is solved by k_band again ----
    {
        movba_options opt{};
        opt.solver = 2;
        movba_handle *h = nullptr;
        EXPECT(movba_create(&h, 0, nullptr, &opt) == MOVBA_OK);
        Win w;
        make(w, 14, 2, 900, 77, false, false);
        fake_set_mode(3, -1);
        check_solved(w, movba_lba_solve(h, &w.d, &w.r));
        EXPECT(w.r.n_pcg_giveups == 1 && w.r.direct_from == 3 && w.r.n_direct == 7 && w.r.n_band == 3);
        fake_set_mode(-1, -1);
        check_solved(w, movba_lba_solve(h, &w.d, &w.r));
        EXPECT(w.r.n_pcg_giveups == 0 && w.r.n_direct == 0 && w.r.n_band == w.r.n_solves);
        EXPECT(movba_lba_upload(h, &w.d) == MOVBA_OK);
        EXPECT(movba_lba_run(h) == MOVBA_OK && movba_lba_run(h) == MOVBA_OK);
        check_solved(w, movba_lba_download(h, &w.r));
        movba_destroy(h);
    }
    // ---- 8. the banded factorisation in one workgroup (small windows by default, any window whose band fits on request): one
    //         solve launch per trial, nothing parks; alone, and as part of a batch beside windows of the PCG ----
    {
        movba_options opt{};
        opt.solver = 2;
        movba_handle *h = nullptr;
        EXPECT(movba_create(&h, 0, nullptr, &opt) == MOVBA_OK);
        Win w;
        for (int it = 0; it < 4; ++it) {
            make(w, 8 + 9 * it, 2, 400 + 500 * it, 800 + it, false, it % 2 == 1);
            check_solved(w, movba_lba_solve(h, &w.d, &w.r));
            EXPECT(w.r.n_band == w.r.n_solves && w.r.n_direct == 0 && w.r.n_pcg_giveups == 0);
        }
        movba_destroy(h);
        movba_handle *hs[3] = { nullptr, nullptr, nullptr };
        Win ws[3];
        hipStream_t st = nullptr;
        EXPECT(hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess);
        for (int i = 0; i < 3; ++i) {
            EXPECT(movba_create(&hs[i], 0, st, nullptr) == MOVBA_OK);
            make(ws[i], i == 1 ? 40 : 9 + i, 2, i == 1 ? 3000 : 500, 900 + i, false, false);       // (default choice: band, PCG, band)
            EXPECT(movba_lba_upload(hs[i], &ws[i].d) == MOVBA_OK);
        }
        EXPECT(movba_lba_run_batch(hs, 3) == MOVBA_OK);
        for (int i = 0; i < 3; ++i) {
            check_solved(ws[i], movba_lba_download(hs[i], &ws[i].r));
            EXPECT((ws[i].r.n_band > 0) == (i != 1));
            movba_destroy(hs[i]);
        }
        (void)hipStreamDestroy(st);
    }
    // ---- 9. a caller whose flattened window lies in movba_host_alloc memory (the adapter): the device reads the index arrays where
    //         they are, the copy engine the rest, nothing is staged, and the upload itself queues the solve's first two kernels;
    //         window after window on one handle, two such threads at once, then the same windows from ordinary memory again ----
    {
        auto worker = [&](unsigned seed0) {
            movba_handle *h = nullptr;
            EXPECT(movba_create(&h, 0, nullptr, nullptr) == MOVBA_OK);
            Win w;
            for (int it = 0; it < 6; ++it) {
                make(w, 9 + 7 * it, 2, 600 + 250 * it, seed0 + it, false, it % 3 == 2);
                const size_t E = w.ep.size();
                struct Pin { void *p; } pins[7];
                auto pin = [&](const void *src, size_t bytes, int k) { pins[k].p = movba_host_alloc(bytes); EXPECT(pins[k].p != nullptr); std::memcpy(pins[k].p, src, bytes); return pins[k].p; };
                w.d.poses = static_cast<double *>(pin(w.poses.data(), 8 * w.poses.size(), 0)); w.d.points = static_cast<double *>(pin(w.points.data(), 8 * w.points.size(), 1));
                w.d.edge_pose = static_cast<int32_t *>(pin(w.ep.data(), 4 * E, 2)); w.d.edge_point = static_cast<int32_t *>(pin(w.el.data(), 4 * E, 3));
                w.d.obs = static_cast<double *>(pin(w.obs.data(), 16 * E, 4)); w.d.inv_sigma2 = static_cast<double *>(pin(w.isig.data(), 8 * E, 5));
                pins[6].p = nullptr;
                if (w.d.obs_right) w.d.obs_right = static_cast<double *>(pin(w.obs_right.data(), 8 * E, 6));
                check_solved(w, movba_lba_solve(h, &w.d, &w.r));
                // the arrays are the caller's again: scribbled over, then the phased calls on a fresh copy
                std::memset(const_cast<double *>(w.d.obs), 0x7f, 16 * E); std::memset(const_cast<int32_t *>(w.d.edge_pose), 0x7f, 4 * E);
                std::memcpy(const_cast<double *>(w.d.obs), w.obs.data(), 16 * E); std::memcpy(const_cast<int32_t *>(w.d.edge_pose), w.ep.data(), 4 * E);
                EXPECT(movba_lba_upload(h, &w.d) == MOVBA_OK);
                EXPECT(movba_lba_run(h) == MOVBA_OK && movba_lba_run(h) == MOVBA_OK);
                check_solved(w, movba_lba_download(h, &w.r));
                for (Pin &q : pins) if (q.p) movba_host_free(q.p);
                // ... and from the caller's ordinary vectors
                w.d.poses = w.poses.data(); w.d.points = w.points.data(); w.d.edge_pose = w.ep.data(); w.d.edge_point = w.el.data();
                w.d.obs = w.obs.data(); w.d.inv_sigma2 = w.isig.data(); if (w.d.obs_right) w.d.obs_right = w.obs_right.data();
                check_solved(w, movba_lba_solve(h, &w.d, &w.r));
            }
            movba_destroy(h);
        };
        std::thread a(worker, 1300u), b(worker, 1700u);
        a.join(); b.join();
    }
    if (fails) { std::fprintf(stderr, "HOST-TSAN FAILED: %d expectation(s)\n", fails); return 1; }
    std::printf("HOST-TSAN OK\n");
    return 0;
}
