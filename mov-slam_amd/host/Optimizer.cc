// MOV_SLAM::Optimizer on MI355X: the host adapter between the reference's map objects and
// the C-ABI of libmovba.so (include/movba.h).
//
// It re-implements, against the reference's own KeyFrame / MapPoint / Map / Frame accessors,
//   * the window selection of Optimizer::LocalBundleAdjustment      (/root/reference/src/Optimizer.cc:464-529),
//   * a flattening pass that replaces the g2o graph construction     (:532-747),
//   * the outlier erase + write-back under Map::mMutexMapUpdate      (:757-840),
//   * BundleAdjustment / GlobalBundleAdjustemnt with their selection and write-back rules (:61-395),
//   * PoseOptimization's gather / result plumbing                    (:397-459),
// and hands the numerical work (what the reference delegates to g2o at :754-755 and, for
// PoseOptimization, to cv::solvePnPRansac at :437) to the GPU.  There is no CPU fallback: when
// the library cannot create a device handle the calls return without touching the map, exactly
// like the reference's own silent early returns (:525-529, :749-751).
//
// Built inside MoV-SLAM it compiles against the real headers; here it is compile-checked and
// tested against mov-slam_amd/host/mock/ (same member names, minimal types).
#include "Optimizer.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <map>
#include <mutex>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "movba.h"

namespace MOV_SLAM
{
    namespace
    {
        const float delta = 5.0f;      // chi2 gate and squared Huber threshold, as the file-static at Optimizer.cc:52
        const uint32_t kPoseRansacSeed = 20221105u;              // fixed: PoseOptimization is reproducible call to call

        // One device handle per calling thread: LocalMapping's thread runs LocalBundleAdjustment while the
        // tracking thread runs PoseOptimization (System.cc:128-129), and a handle is single-threaded.
        struct ThreadHandle
        {
            movba_handle *h = nullptr;
            bool tried = false;
            ~ThreadHandle() { if (h) movba_destroy(h); }
            movba_handle *get()
            {
                if (!tried)
                {
                    tried = true;
                    if (movba_create(&h, 0, nullptr, nullptr) != MOVBA_OK)
                    {
                        h = nullptr;
                        std::fprintf(stderr, "MOV_SLAM::Optimizer: libmovba could not open HIP device 0 — optimisation calls will be skipped\n");
                    }
                }
                return h;
            }
        };
        thread_local ThreadHandle tls_handle;

        // KeyFrame* -> vertex index, open addressing (one lookup per observation: std::unordered_map's node chasing was a
        // third of the extraction time on a 50-keyframe window)
        struct KfIndex
        {
            std::vector<KeyFrame *> key;
            std::vector<int32_t> val;
            std::vector<uint8_t> usable;        // per vertex: !isBad() && same map (see build)
            // per vertex, read once per window instead of once per observation: where the keyframe's (immutable) keypoint
            // arrays start, its camera's four parameters (GeometricCamera::getParameter is a virtual call) and its baseline
            struct PerKf { const cv::KeyPoint *keys; const float *uright; const float *inv_sigma2; double cam[4]; double bf; };
            std::vector<PerKf> kf;
            size_t mask = 0;
            static size_t hash(const KeyFrame *p) { return (size_t)((reinterpret_cast<uintptr_t>(p) >> 4) * 0x9E3779B97F4A7C15ull >> 20); }
            // `pCurrentMap` non-null: a keyframe of another map yields no edges (Optimizer.cc:646).  The edge loop asks
            // KeyFrame::isBad() and GetMap() — a mutex each in MoV-SLAM's classes — once per window keyframe here instead
            // of once per observation (the reference asks per observation, :646, :155: 2 x 110 000 locks at cfg3)
            void build(const std::vector<KeyFrame *> &kfs, Map *pCurrentMap)
            {
                usable.resize(kfs.size());
                for (size_t i = 0; i < kfs.size(); ++i) usable[i] = (!kfs[i]->isBad() && (!pCurrentMap || kfs[i]->GetMap() == pCurrentMap)) ? 1 : 0;
                kf.resize(kfs.size());
                for (size_t i = 0; i < kfs.size(); ++i)
                {
                    const KeyFrame *k = kfs[i];
                    kf[i].keys = k->mvKeysUn.data(); kf[i].uright = k->mvuRight.data(); kf[i].inv_sigma2 = k->mvInvLevelSigma2.data();
                    for (int q = 0; q < 4; ++q) kf[i].cam[q] = k->mpCamera ? (double)k->mpCamera->getParameter(q) : 0.0;
                    kf[i].bf = k->mbf;
                }
                size_t cap = 16;
                while (cap < 2 * kfs.size() + 2) cap <<= 1;
                key.assign(cap, nullptr); val.assign(cap, -1); mask = cap - 1;
                for (size_t i = 0; i < kfs.size(); ++i)
                {
                    size_t h = hash(kfs[i]) & mask;
                    while (key[h]) h = (h + 1) & mask;
                    key[h] = kfs[i]; val[h] = (int32_t)i;
                }
            }
            int32_t find(const KeyFrame *p) const       // -1: no vertex
            {
                size_t h = hash(p) & mask;
                while (key[h]) { if (key[h] == p) return val[h]; h = (h + 1) & mask; }
                return -1;
            }
        };

        // One entry of a MapPoint's observation map, copied ONCE per point and call (MapPoint::GetObservations() hands out a
        // std::map copy under the point's mutex, MapPoint.cc:211-215: the reference takes two per point before the solve and
        // UpdateNormalAndDepth a third one after it).
        struct ObsRef
        {
            KeyFrame *kf;
            int left, right;
            int32_t vertex;                         // index of the keyframe's vertex in the window (-1: none), filled by push_point
        };

        // The window's camera: ONE pinhole (and one stereo baseline) in every shipped MoV-SLAM configuration; a window whose
        // keyframes disagree is handed over with intrinsics by keyframe (see emit_point, solve).
        struct CamState
        {
            double cam[4] = {0, 0, 0, 0};
            bool cam_set = false;
            bool any_stereo = false;
            double bf = 0.0;
            bool cam_mixed = false;                 // keyframes with different intrinsics / baselines in one window
            void reset() { cam_set = false; any_stereo = false; bf = 0.0; cam_mixed = false; }
        };

        // The arrays the window is flattened INTO live in movba_host_alloc memory (pinned, device-visible) wherever the library
        // can provide it: the device then reads the index arrays where they lie and the copy engine takes observations, information
        // and estimates straight out of them - nothing is staged in between (movba.h, movba_lba_desc).  Allocations happen when a
        // vector grows, i.e. during the first windows of a session; ordinary memory when the library has none to give.
        template <class T> struct PinnedAlloc
        {
            typedef T value_type;
            PinnedAlloc() = default;
            template <class U> PinnedAlloc(const PinnedAlloc<U> &) {}
            T *allocate(size_t n)
            {
                // (a small header in front of the block says which allocator it came from)
                const size_t bytes = n * sizeof(T) + 16;
                char *p = static_cast<char *>(movba_host_alloc(bytes));
                const bool pinned = p != nullptr;
                if (!p) p = static_cast<char *>(::operator new(bytes));
                *reinterpret_cast<uint64_t *>(p) = pinned ? 1u : 0u;
                return reinterpret_cast<T *>(p + 16);
            }
            void deallocate(T *q, size_t)
            {
                char *p = reinterpret_cast<char *>(q) - 16;
                if (*reinterpret_cast<uint64_t *>(p)) movba_host_free(p); else ::operator delete(p);
            }
            template <class U> bool operator==(const PinnedAlloc<U> &) const { return true; }
            template <class U> bool operator!=(const PinnedAlloc<U> &) const { return false; }
        };
        // (-DMOVBA_ADAPTER_PLAIN_VECTORS: ordinary memory, for same-box comparisons - the library then stages the arrays once
        //  more: 1.13 against 1.05 ms per solve call at cfg3, extraction and write-back unchanged)
#ifdef MOVBA_ADAPTER_PLAIN_VECTORS
        template <class T> using pinned_vector = std::vector<T>;
#else
        template <class T> using pinned_vector = std::vector<T, PinnedAlloc<T>>;
#endif

        // Flattened window in the layout of movba_lba_desc, plus the bookkeeping to write results back.
        // One instance per calling thread, reused from call to call (the vectors keep their capacity).
        struct Flat : CamState
        {
            std::vector<double> cam_kf, bf_kf;      // intrinsics / baselines by vertex, filled only for a mixed-camera window
            std::vector<ObsRef> obs_all;            // observation lists of the local map points, back to back (map order)
            std::vector<size_t> obs_start;          // first observation of local map point k; one more entry at the end
            std::vector<int32_t> point_edge0;       // first edge of problem point k (its edges are contiguous, in observation order); +1 entry
            void clear()
            {
                obs_all.clear(); obs_start.clear(); point_edge0.clear();
                kfs.clear(); fixed.clear(); poses.clear(); mps.clear(); points.clear(); edge_pose.clear(); edge_point.clear();
                obs.clear(); inv_sigma2.clear(); edge_kf.clear(); edge_mp.clear(); obs_right.clear();
                reset();
            }
            void resize_edges(size_t n)
            {
                edge_pose.resize(n); edge_point.resize(n); obs.resize(2 * n); inv_sigma2.resize(n);
                edge_kf.resize(n); edge_mp.resize(n); obs_right.resize(n);
            }
            std::vector<KeyFrame *> kfs;            // vertex order: ascending mnId (g2o's hessian order)
            pinned_vector<uint8_t> fixed;
            pinned_vector<double> poses;
            std::vector<MapPoint *> mps;
            pinned_vector<double> points;
            pinned_vector<int32_t> edge_pose, edge_point;
            pinned_vector<double> obs, inv_sigma2;
            std::vector<KeyFrame *> edge_kf;        // vpEdgeKFMono
            std::vector<MapPoint *> edge_mp;        // vpMapPointEdgeMono
            pinned_vector<double> obs_right;        // mvuRight of stereo observations, -1 for monocular ones
        };

        void push_pose(Flat &f, KeyFrame *pKF, bool isFixed)
        {
            const Sophus::SE3<float> Tcw = pKF->GetPose();
            f.kfs.push_back(pKF);
            f.fixed.push_back(isFixed ? 1 : 0);
            // Tcw.unit_quaternion().cast<double>(), Tcw.translation().cast<double>()  (Optimizer.cc:559)
            f.poses.push_back(Tcw.unit_quaternion().x()); f.poses.push_back(Tcw.unit_quaternion().y());
            f.poses.push_back(Tcw.unit_quaternion().z()); f.poses.push_back(Tcw.unit_quaternion().w());
            f.poses.push_back(Tcw.translation()(0)); f.poses.push_back(Tcw.translation()(1)); f.poses.push_back(Tcw.translation()(2));
        }

        // Vertices in ascending KeyFrame::mnId: g2o numbers the free poses in vertex-id order
        // (SparseOptimizer::initializeOptimization), and the ids are the mnIds (Optimizer.cc:560, 577).
        void sort_poses(Flat &f)
        {
            std::vector<size_t> order(f.kfs.size());
            for (size_t i = 0; i < order.size(); ++i) order[i] = i;
            std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return f.kfs[a]->mnId < f.kfs[b]->mnId; });
            // (permuted through ordinary temporaries and written back: the window's own arrays keep their pinned storage)
            std::vector<KeyFrame *> kfs; std::vector<uint8_t> fixed; std::vector<double> poses;
            for (size_t k : order)
            {
                kfs.push_back(f.kfs[k]); fixed.push_back(f.fixed[k]);
                for (int q = 0; q < 7; ++q) poses.push_back(f.poses[7 * k + q]);
            }
            f.kfs.swap(kfs);
            std::copy(fixed.begin(), fixed.end(), f.fixed.begin()); std::copy(poses.begin(), poses.end(), f.poses.begin());
        }

        // The edges of one MapPoint vertex (Optimizer.cc:623-705 / :142-190) from the point's observation list, written to
        // edge slots e0, e0 + 1, ... of `f` (sized by the caller: at most one edge per observation) as edges of problem point
        // `pid`.  Returns the number of edges written.
        // The vertices of a run of observations, looked up ahead of emit_point (resolved = true there), and the two cache lines
        // each edge will read - its keypoint and its right-image coordinate, at random places of the keyframes' arrays -
        // requested while the edges of the points before are written (LocalBundleAdjustment: two points ahead).
        void resolve_ahead(ObsRef *ob, ObsRef *ob_end, const KfIndex &kfIndex)
        {
            for (; ob != ob_end; ++ob)
            {
                const int32_t vertex = kfIndex.find(ob->kf);
                ob->vertex = vertex;
                if (vertex < 0 || ob->left < 0) continue;
                __builtin_prefetch(kfIndex.kf[vertex].keys + ob->left);
                __builtin_prefetch(kfIndex.kf[vertex].uright + ob->left);
            }
        }

        int emit_point(Flat &f, CamState &cs, size_t e0, MapPoint *pMP, int32_t pid, ObsRef *ob, ObsRef *ob_end, const KfIndex &kfIndex,
                       Map *pCurrentMap, bool requireSameMap, bool resolved = false)
        {
            (void)pCurrentMap; (void)requireSameMap;             // (folded into kfIndex.usable by its build)
            size_t e = e0;
            for (; ob != ob_end; ++ob)
            {
                KeyFrame *pKFi = ob->kf;
                const int32_t vertex = resolved ? ob->vertex : kfIndex.find(pKFi);
                ob->vertex = vertex;
                if (vertex < 0)
                    continue;                                   // observer without a vertex
                if (!kfIndex.usable[vertex])
                    continue;                                   // pKFi->isBad() || pKFi->GetMap() != pCurrentMap, asked once per keyframe
                const int leftIndex = ob->left;
                if (leftIndex == -1)
                    continue;
                const KfIndex::PerKf &pk = kfIndex.kf[vertex];
                const cv::KeyPoint &kpUn = pk.keys[leftIndex];   // pKFi->mvKeysUn[leftIndex]
                // stereo observation (Optimizer.cc:673-705): third measurement kp_ur = mvuRight[idx], e->bf = pKFi->mbf
                const float kp_ur = pk.uright[leftIndex];
                if (kp_ur >= 0)
                {
                    if (cs.any_stereo && cs.bf != pk.bf) cs.cam_mixed = true;
                    cs.any_stereo = true;
                    cs.bf = pk.bf;
                }
                f.obs_right[e] = kp_ur >= 0 ? (double)kp_ur : -1.0;
                // e->pCamera = pKFi->mpCamera per edge (Optimizer.cc:664): ONE pinhole for the window is what every MoV-SLAM
                // configuration has (one camera, Tracking.cc builds a single mpCamera) and what the descriptor's scalars say;
                // a window whose keyframes disagree goes over with movba_lba_desc::cam_kf / bf_kf (solve() below)
                if (!cs.cam_set)
                {
                    for (int k = 0; k < 4; ++k) cs.cam[k] = pk.cam[k];
                    cs.cam_set = true;
                }
                else if ((cs.cam[0] != pk.cam[0]) | (cs.cam[1] != pk.cam[1]) | (cs.cam[2] != pk.cam[2]) | (cs.cam[3] != pk.cam[3]))
                    cs.cam_mixed = true;
                f.edge_pose[e] = vertex;
                f.edge_point[e] = pid;
                f.obs[2 * e] = kpUn.pt.x; f.obs[2 * e + 1] = kpUn.pt.y;
                f.inv_sigma2[e] = pk.inv_sigma2[kpUn.octave];
                f.edge_kf[e] = pKFi;
                f.edge_mp[e] = pMP;
                ++e;
            }
            return (int)(e - e0);
        }

        // One MapPoint vertex and its edges appended to `f`.  Returns the number of edges added (none: no vertex either).
        int push_point(Flat &f, MapPoint *pMP, ObsRef *ob, ObsRef *ob_end, const KfIndex &kfIndex, Map *pCurrentMap, bool requireSameMap)
        {
            const Eigen::Vector3f wp = pMP->GetWorldPos();
            const size_t e0 = f.edge_pose.size();
            f.resize_edges(e0 + (size_t)(ob_end - ob));
            const int nEdges = emit_point(f, f, e0, pMP, (int32_t)f.mps.size(), ob, ob_end, kfIndex, pCurrentMap, requireSameMap);
            f.resize_edges(e0 + (size_t)nEdges);
            if (nEdges > 0)
            {
                f.point_edge0.push_back((int32_t)e0);
                f.mps.push_back(pMP);
                f.points.push_back(wp(0)); f.points.push_back(wp(1)); f.points.push_back(wp(2));
            }
            return nEdges;
        }

        // the same from a fresh copy of the point's observations (BundleAdjustment: one copy per point)
        int push_point(Flat &f, MapPoint *pMP, const KfIndex &kfIndex, Map *pCurrentMap, bool requireSameMap)
        {
            const std::map<KeyFrame *, std::tuple<int, int>> observations = pMP->GetObservations();
            const size_t o0 = f.obs_all.size();
            for (std::map<KeyFrame *, std::tuple<int, int>>::const_iterator mit = observations.begin(); mit != observations.end(); ++mit)
                f.obs_all.push_back(ObsRef{mit->first, std::get<0>(mit->second), std::get<1>(mit->second), -1});
            const int n = push_point(f, pMP, f.obs_all.data() + o0, f.obs_all.data() + f.obs_all.size(), kfIndex, pCurrentMap, requireSameMap);
            f.obs_all.resize(o0);
            return n;
        }

        double now_ms()
        {
            return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
        }
        // host-side phases of the last LocalBundleAdjustment call of this thread: extraction (selection + flattening), solve
        // call, write-back (ms); read through movba_adapter_last_timing()
        thread_local double tls_timing[3] = {0, 0, 0};
        // what became of this thread's last optimisation call: MOVBA_OK, one of the reference's own silent returns (> 0:
        // stopped, nothing to optimise, no fixed keyframe), or a library error (< 0: the map was left untouched although the
        // reference would have optimised it); library errors are also counted process-wide (movba_adapter_error_count())
        thread_local int tls_last_status = 0;
        std::atomic<long> g_library_errors{0};
        void note_status(const char *who, int status)
        {
            tls_last_status = status;
            if (status < 0)
            {
                g_library_errors.fetch_add(1, std::memory_order_relaxed);
                std::fprintf(stderr, "MOV_SLAM::Optimizer::%s: libmovba returned %d (%s): optimisation SKIPPED, map left as it was\n", who, status, movba_status_string(status));
            }
        }

        // $MOVBA_DUMP_DIR/lba_<n>.mbw: the flattened window, for replay on a GPU box without the reference's
        // stack (layout in mov-slam_amd/movba/capture.py)
        void dump_window(const movba_lba_desc &d)
        {
            const char *dir = std::getenv("MOVBA_DUMP_DIR");
            if (!dir || !*dir) return;
            static std::atomic<int> counter{0};
            char path[1024];
            std::snprintf(path, sizeof path, "%s/lba_%06d.mbw", dir, counter.fetch_add(1));
            FILE *f = std::fopen(path, "wb");
            if (!f) return;
            const uint32_t ver = 1, flags = d.flags | (d.obs_right ? 0x80000000u : 0u);   // top bit: stereo trailer follows
            const int32_t hd[4] = {d.n_poses, d.n_points, d.n_edges, d.max_iters};
            const double dd[6] = {d.fx, d.fy, d.cx, d.cy, d.huber_delta, d.chi2_gate};
            const uint8_t pad[8] = {0};
            std::fwrite("MOVBAWIN", 1, 8, f);
            std::fwrite(&ver, 4, 1, f); std::fwrite(hd, 4, 4, f); std::fwrite(&flags, 4, 1, f); std::fwrite(dd, 8, 6, f);
            std::fwrite(d.pose_fixed, 1, d.n_poses, f); std::fwrite(pad, 1, (8 - d.n_poses % 8) % 8, f);
            std::fwrite(d.poses, 8, 7 * (size_t)d.n_poses, f); std::fwrite(d.points, 8, 3 * (size_t)d.n_points, f);
            std::fwrite(d.edge_pose, 4, d.n_edges, f); std::fwrite(d.edge_point, 4, d.n_edges, f);
            std::fwrite(d.obs, 8, 2 * (size_t)d.n_edges, f); std::fwrite(d.inv_sigma2, 8, d.n_edges, f);
            if (d.obs_right) { std::fwrite(&d.bf, 8, 1, f); std::fwrite(d.obs_right, 8, d.n_edges, f); }
            std::fclose(f);
        }

        // double array in movba_host_alloc memory (pinned, device-visible: the solve's last kernel writes the results into
        // it across the bus, nothing is copied out afterwards); plain new[] when the library cannot provide it
        struct PinnedArray
        {
            double *p = nullptr;
            size_t cap = 0, n = 0;
            bool pinned = false;
            ~PinnedArray() { release(); }
            void release()
            {
                if (p) { if (pinned) movba_host_free(p); else delete[] p; }
                p = nullptr; cap = 0;
            }
            void resize(size_t count)
            {
                if (count > cap)
                {
                    release();
                    const size_t c = count + count / 4 + 16;
                    p = static_cast<double *>(movba_host_alloc(c * sizeof(double)));
                    pinned = p != nullptr;
                    if (!p) p = new double[c];
                    cap = c;
                }
                n = count;
            }
            double *data() { return p; }
            const double &operator[](size_t k) const { return p[k]; }
        };

        struct Solved
        {
            PinnedArray poses, points;
            std::vector<uint8_t> outlier;
            int status = MOVBA_ERR_HIP;
        };

        thread_local Flat tls_flat;
        thread_local std::vector<size_t> tls_point_local;
        thread_local std::vector<int32_t> tls_nedge;
        thread_local Solved tls_solved;

        Solved &solve(Flat &f, int nIterations, bool bRobust, bool *pbStopFlag)
        {
            Solved &s = tls_solved;                              // result buffers reused from call to call
            s.status = MOVBA_ERR_HIP;
            movba_handle *h = tls_handle.get();
            if (!h) return s;
            movba_lba_desc d{};
            d.n_poses = (int32_t)f.kfs.size(); d.n_points = (int32_t)f.mps.size(); d.n_edges = (int32_t)f.edge_pose.size();
            d.poses = f.poses.data(); d.pose_fixed = f.fixed.data(); d.points = f.points.data();
            d.edge_pose = f.edge_pose.data(); d.edge_point = f.edge_point.data();
            d.obs = f.obs.data(); d.inv_sigma2 = f.inv_sigma2.data();
            d.obs_right = f.any_stereo ? f.obs_right.data() : nullptr;
            d.bf = f.bf;
            d.fx = f.cam[0]; d.fy = f.cam[1]; d.cx = f.cam[2]; d.cy = f.cam[3];
            if (f.cam_mixed)
            {
                // every edge its keyframe's camera and baseline (Optimizer.cc:664, 690-695): tables by vertex
                f.cam_kf.resize(4 * f.kfs.size()); f.bf_kf.resize(f.kfs.size());
                for (size_t i = 0; i < f.kfs.size(); ++i)
                {
                    for (int k = 0; k < 4; ++k) f.cam_kf[4 * i + k] = f.kfs[i]->mpCamera->getParameter(k);
                    f.bf_kf[i] = f.kfs[i]->mbf;
                }
                d.cam_kf = f.cam_kf.data(); d.bf_kf = f.bf_kf.data();
            }
            const float thHuber = std::sqrt(delta);              // const float thHuberMono = sqrt(delta)  (Optimizer.cc:616)
            d.huber_delta = bRobust ? (double)thHuber : 0.0;
            d.chi2_gate = delta;
            d.max_iters = nIterations;
            d.max_trials = 0;
            d.flags = MOVBA_FLAG_STALE_ERROR_QUIRK;
            d.stop = reinterpret_cast<const volatile uint8_t *>(pbStopFlag);
            s.poses.resize(f.poses.size()); s.points.resize(f.points.size());
            s.outlier.resize(f.edge_pose.size());
            movba_lba_result r{};
            r.poses = s.poses.data(); r.points = s.points.data(); r.outlier = s.outlier.data();
            r.chi2 = nullptr;                                    // (the gate's verdict is all Optimizer.cc:757-775 uses: not asked for, not transferred)
            dump_window(d);
            s.status = movba_lba_solve(h, &d, &r);
            return s;
        }

        Sophus::SE3f pose_to_se3f(const double *p)
        {
            // Sophus::SE3f(SE3quat.rotation().cast<float>(), SE3quat.translation().cast<float>())  (Optimizer.cc:827)
            return Sophus::SE3f(Eigen::Quaternionf((float)p[3], (float)p[0], (float)p[1], (float)p[2]),
                                Eigen::Vector3f((float)p[4], (float)p[5], (float)p[6]));
        }
    } // namespace

    double adapter_timing(int k) { return tls_timing[k]; }
    int adapter_last_status() { return tls_last_status; }
    long adapter_error_count() { return g_library_errors.load(std::memory_order_relaxed); }

    void Optimizer::GlobalBundleAdjustemnt(Map *pMap, int nIterations, bool *pbStopFlag, const unsigned long nLoopKF, const bool bRobust)
    {
        std::vector<KeyFrame *> vpKFs = pMap->GetAllKeyFrames();
        std::vector<MapPoint *> vpMP = pMap->GetAllMapPoints();
        BundleAdjustment(vpKFs, vpMP, nIterations, pbStopFlag, nLoopKF, bRobust);
    }

    void Optimizer::BundleAdjustment(const std::vector<KeyFrame *> &vpKFs, const std::vector<MapPoint *> &vpMP,
                                     int nIterations, bool *pbStopFlag, const unsigned long nLoopKF, const bool bRobust)
    {
        if (vpKFs.empty())
            return;
        Map *pMap = vpKFs[0]->GetMap();
        Flat &f = tls_flat;
        f.clear();
        for (size_t i = 0; i < vpKFs.size(); i++)
        {
            KeyFrame *pKF = vpKFs[i];
            if (pKF->isBad())
                continue;
            push_pose(f, pKF, pKF->mnId == pMap->GetInitKFid());
        }
        sort_poses(f);
        KfIndex kfIndex;
        kfIndex.build(f.kfs, nullptr);                  // (BundleAdjustment takes keyframes of any map: requireSameMap = false)

        // MapPoints without any edge are left out of the problem (vbNotIncludedMP, Optimizer.cc:273-281)
        std::vector<bool> vbNotIncludedMP(vpMP.size(), true);
        for (size_t i = 0; i < vpMP.size(); i++)
        {
            MapPoint *pMP = vpMP[i];
            if (pMP->isBad())
                continue;
            vbNotIncludedMP[i] = push_point(f, pMP, kfIndex, pMap, false) == 0;
        }
        if (f.edge_pose.empty())
            return;

        const Solved &s = solve(f, nIterations, bRobust, pbStopFlag);
        note_status("BundleAdjustment", s.status);
        if (s.status != MOVBA_OK)
            return;

        const bool direct = (nLoopKF == pMap->GetOriginKF()->mnId);
        for (size_t i = 0; i < f.kfs.size(); ++i)
        {
            KeyFrame *pKF = f.kfs[i];
            if (direct)
                pKF->SetPose(pose_to_se3f(&s.poses[7 * i]));
            else
            {
                pKF->mTcwGBA = pose_to_se3f(&s.poses[7 * i]);
                pKF->mnBAGlobalForKF = nLoopKF;
            }
        }
        for (size_t k = 0; k < f.mps.size(); ++k)
        {
            MapPoint *pMP = f.mps[k];
            if (pMP->isBad())
                continue;
            const Eigen::Vector3f X((float)s.points[3 * k], (float)s.points[3 * k + 1], (float)s.points[3 * k + 2]);
            if (direct)
            {
                pMP->SetWorldPos(X);
                pMP->UpdateNormalAndDepth();
            }
            else
            {
                pMP->mPosGBA = X;
                pMP->mnBAGlobalForKF = nLoopKF;
            }
        }
    }

    void Optimizer::LocalBundleAdjustment(KeyFrame *pKF, bool *pbStopFlag, Map *pMap, int &num_fixedKF, int &num_OptKF, int &num_MPs, int &num_edges)
    {
        (void)num_MPs;                                          // never assigned by the reference either
        const double t_begin = now_ms();
        tls_timing[0] = tls_timing[1] = tls_timing[2] = 0.0;
        // ---- local keyframes: pKF and its covisible keyframes (Optimizer.cc:464-477) ----
        std::vector<KeyFrame *> lLocalKeyFrames;
        lLocalKeyFrames.push_back(pKF);
        pKF->mnBALocalForKF = pKF->mnId;
        Map *pCurrentMap = pKF->GetMap();
        const std::vector<KeyFrame *> vNeighKFs = pKF->GetVectorCovisibleKeyFrames();
        for (size_t i = 0; i < vNeighKFs.size(); i++)
        {
            KeyFrame *pKFi = vNeighKFs[i];
            pKFi->mnBALocalForKF = pKF->mnId;
            if (!pKFi->isBad() && pKFi->GetMap() == pCurrentMap)
                lLocalKeyFrames.push_back(pKFi);
        }

        // ---- local map points seen by them (Optimizer.cc:479-504) ----
        num_fixedKF = 0;
        std::vector<MapPoint *> lLocalMapPoints;
        for (KeyFrame *pKFi : lLocalKeyFrames)
        {
            if (pKFi->mnId == pMap->GetInitKFid())
                num_fixedKF = 1;
            const std::vector<MapPoint *> vpMPs = pKFi->GetMapPointMatches();
            const size_t nMatches = vpMPs.size();
            for (size_t im = 0; im < nMatches; ++im)
            {
                MapPoint *pMP = vpMPs[im];
                if (im + 8 < nMatches && vpMPs[im + 8]) __builtin_prefetch(&vpMPs[im + 8]->mnBALocalForKF);      // (the points sit all over the heap)
                // (the reference's conjunction, Optimizer.cc:489-499, with the plain member test first: a point seen by five
                //  local keyframes is met five times, and MapPoint::isBad() / GetMap() take three mutexes between them)
                if (pMP && pMP->mnBALocalForKF != pKF->mnId && !pMP->isBad() && pMP->GetMap() == pCurrentMap)
                {
                    lLocalMapPoints.push_back(pMP);
                    pMP->mnBALocalForKF = pKF->mnId;
                }
            }
        }

        static const bool lapon = std::getenv("MOVBA_ADAPTER_LAPS") != nullptr;
        double lap_t = now_ms();
        auto lap = [&](const char *what) { if (lapon) { const double t = now_ms(); std::fprintf(stderr, "adapter lap: %-24s %.3f ms\n", what, t - lap_t); lap_t = t; } };
        lap("selection");
        // ---- ONE copy of every local point's observation map, shared by the fixed-keyframe pass, the edge pass and the
        //      normal / depth update after the solve (the reference copies the std::map three times per point) ----
        Flat &f = tls_flat;
        f.clear();
        const size_t nLocal = lLocalMapPoints.size();
        f.obs_start.assign(nLocal + 1, 0);
        for (size_t lp = 0; lp < nLocal; ++lp)
        {
#ifdef MOVBA_MAPPOINT_HAS_FOR_EACH_OBSERVATION
            // (the optional accessor of INTEGRATION.md 1.3: the same entries in the same order, read under the point's lock
            //  without the std::map copy and its node allocations — 1.8 of the 2.9 ms of this phase at 20 000 points)
            lLocalMapPoints[lp]->ForEachObservation([&](KeyFrame *pKFo, int left, int right) { f.obs_all.push_back(ObsRef{pKFo, left, right, -1}); });
#else
            const std::map<KeyFrame *, std::tuple<int, int>> observations = lLocalMapPoints[lp]->GetObservations();
            for (std::map<KeyFrame *, std::tuple<int, int>>::const_iterator mit = observations.begin(); mit != observations.end(); ++mit)
                f.obs_all.push_back(ObsRef{mit->first, std::get<0>(mit->second), std::get<1>(mit->second), -1});
#endif
            f.obs_start[lp + 1] = f.obs_all.size();
        }
        lap("observation copies");
        // ---- fixed keyframes: other observers of the local points (Optimizer.cc:506-523) ----
        std::vector<KeyFrame *> lFixedCameras;
        for (const ObsRef &ob : f.obs_all)
        {
            KeyFrame *pKFi = ob.kf;
            if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId)
            {
                pKFi->mnBAFixedForKF = pKF->mnId;
                if (!pKFi->isBad() && pKFi->GetMap() == pCurrentMap)
                    lFixedCameras.push_back(pKFi);
            }
        }
        num_fixedKF = (int)lFixedCameras.size() + num_fixedKF;
        if (num_fixedKF == 0)
            return;                                             // "LBA aborted" (Optimizer.cc:525-529)

        lap("fixed keyframes");
        // ---- flatten (replaces the g2o vertex / edge construction, Optimizer.cc:532-747) ----
        pCurrentMap->msOptKFs.clear();
        pCurrentMap->msFixedKFs.clear();
        for (KeyFrame *pKFi : lLocalKeyFrames)
        {
            push_pose(f, pKFi, pKFi->mnId == pMap->GetInitKFid());
            pCurrentMap->msOptKFs.insert(pKFi->mnId);
        }
        num_OptKF = (int)lLocalKeyFrames.size();
        for (KeyFrame *pKFi : lFixedCameras)
        {
            push_pose(f, pKFi, true);
            pCurrentMap->msFixedKFs.insert(pKFi->mnId);
        }
        sort_poses(f);
        KfIndex kfIndex;
        kfIndex.build(f.kfs, pCurrentMap);

        lap("poses");
        // every local point writes its vertex at its own index and its edges where its observations start (at most one
        // edge per observation): plain indexed stores into arrays sized once.  Almost always every observation becomes an
        // edge (all observers are local or fixed keyframes of this window) and the arrays are final as they are; otherwise
        // the gaps are closed below.
        int nEdges = 0;
        std::vector<MapPoint *> edgeless;                        // vertices g2o would keep but never move
        std::vector<size_t> &point_local = tls_point_local;      // problem point k -> its index among the local map points
        std::vector<int32_t> &nedge = tls_nedge;
        const size_t nObsAll = f.obs_all.size();
        f.resize_edges(nObsAll);
        f.mps.assign(nLocal, nullptr); f.points.resize(3 * nLocal); nedge.resize(nLocal);
        constexpr size_t kAhead = 2;                              // points whose observations are resolved and prefetched ahead
        resolve_ahead(f.obs_all.data(), f.obs_all.data() + f.obs_start[std::min(kAhead, nLocal)], kfIndex);
        for (size_t lp = 0; lp < nLocal; ++lp)
        {
            MapPoint *pMP = lLocalMapPoints[lp];
            if (lp + kAhead < nLocal)
            {
                resolve_ahead(f.obs_all.data() + f.obs_start[lp + kAhead], f.obs_all.data() + f.obs_start[lp + kAhead + 1], kfIndex);
                __builtin_prefetch(lLocalMapPoints[lp + kAhead]);
            }
            const Eigen::Vector3f wp = pMP->GetWorldPos();
            nedge[lp] = emit_point(f, f, f.obs_start[lp], pMP, (int32_t)lp, f.obs_all.data() + f.obs_start[lp],
                                   f.obs_all.data() + f.obs_start[lp + 1], kfIndex, pCurrentMap, true, true);
            f.mps[lp] = pMP;
            f.points[3 * lp] = wp(0); f.points[3 * lp + 1] = wp(1); f.points[3 * lp + 2] = wp(2);
        }
        bool dense = true;
        for (size_t lp = 0; lp < nLocal; ++lp)
        {
            nEdges += nedge[lp];
            dense &= nedge[lp] > 0 && (size_t)nedge[lp] == f.obs_start[lp + 1] - f.obs_start[lp];
        }
        point_local.resize(nLocal);
        f.point_edge0.resize(nLocal + 1);
        if (dense)
        {
            for (size_t lp = 0; lp < nLocal; ++lp) { point_local[lp] = lp; f.point_edge0[lp] = (int32_t)f.obs_start[lp]; }
            f.point_edge0[nLocal] = (int32_t)nObsAll;
        }
        else
        {
            size_t eo = 0, k = 0;
            for (size_t lp = 0; lp < nLocal; ++lp)
            {
                const size_t n = (size_t)nedge[lp], src = f.obs_start[lp];
                if (n == 0) { edgeless.push_back(lLocalMapPoints[lp]); continue; }
                for (size_t q = 0; q < n; ++q)
                {
                    f.edge_pose[eo + q] = f.edge_pose[src + q]; f.edge_point[eo + q] = (int32_t)k;
                    f.obs[2 * (eo + q)] = f.obs[2 * (src + q)]; f.obs[2 * (eo + q) + 1] = f.obs[2 * (src + q) + 1];
                    f.inv_sigma2[eo + q] = f.inv_sigma2[src + q]; f.edge_kf[eo + q] = f.edge_kf[src + q];
                    f.edge_mp[eo + q] = f.edge_mp[src + q]; f.obs_right[eo + q] = f.obs_right[src + q];
                }
                f.mps[k] = f.mps[lp];
                for (int q = 0; q < 3; ++q) f.points[3 * k + q] = f.points[3 * lp + q];
                f.point_edge0[k] = (int32_t)eo;
                point_local[k] = lp;
                eo += n; ++k;
            }
            f.resize_edges(eo);
            f.mps.resize(k); f.points.resize(3 * k); point_local.resize(k);
            f.point_edge0.resize(k + 1);
            f.point_edge0[k] = (int32_t)eo;
        }
        num_edges = nEdges;
        lap("points and edges");

        if (pbStopFlag)
            if (*pbStopFlag)
                return;                                         // Optimizer.cc:749-751

        // ---- solve on the GPU: optimizer.initializeOptimization(); optimizer.optimize(10) (Optimizer.cc:754-755) ----
        const double t_solve0 = now_ms();
        tls_timing[0] = t_solve0 - t_begin;
        const Solved &s = solve(f, 10, true, pbStopFlag);
        const double t_solve1 = now_ms();
        tls_timing[1] = t_solve1 - t_solve0;
        note_status("LocalBundleAdjustment", s.status);
        if (s.status != MOVBA_OK)
            return;                                             // stopped / nothing to do / library error: map untouched

        // ---- inlier check (Optimizer.cc:757-803): the monocular edges in their order, THEN the stereo edges in theirs — the
        // reference keeps them in separate vectors (vpEdgesMono / vpEdgesStereo), so in a mixed window every monocular outlier
        // is erased before the first stereo one ----
        std::vector<std::pair<KeyFrame *, MapPoint *>> vToErase;
        for (int pass = 0; pass < (f.any_stereo ? 2 : 1); ++pass)
            for (size_t i = 0; i < f.edge_kf.size(); i++)
            {
                if (!s.outlier[i] || (f.any_stereo && (f.obs_right[i] >= 0.0) != (pass == 1)))
                    continue;
                MapPoint *pMP = f.edge_mp[i];
                if (pMP->isBad())
                    continue;
                vToErase.push_back(std::make_pair(f.edge_kf[i], pMP));
            }

        lap_t = now_ms();
        lap("inlier check");
        // ---- write-back under the map mutex (Optimizer.cc:807-840) ----
        std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);
        for (size_t i = 0; i < vToErase.size(); i++)
        {
            KeyFrame *pKFi = vToErase[i].first;
            MapPoint *pMPi = vToErase[i].second;
            pKFi->EraseMapPointMatch(pMPi);                     // before EraseObservation: it needs the index
            pMPi->EraseObservation(pKFi);
        }
        lap("erasures");
        for (KeyFrame *pKFi : lLocalKeyFrames)                  // local keyframes only, the fixed init KF included
            pKFi->SetPose(pose_to_se3f(&s.poses[7 * kfIndex.find(pKFi)]));
#ifdef MOVBA_MAPPOINT_HAS_SET_DISTANCES
        // MapPoint::UpdateNormalAndDepth (MapPoint.cc:362-435) copies the observation map once more and locks every observer
        // for its camera centre: with P points that is as expensive as the solve.  Its arithmetic needs only what is already
        // here: the observation lists copied before the solve (minus the pairs just erased), the camera centres of the
        // window's keyframes (read ONCE each, after SetPose) and the new positions; the results are stored through
        // SetNormalVector / SetMinMaxDistance.  Same float operations in the same order as the reference.  A point with an
        // observer outside the window or a right-camera observation falls back to UpdateNormalAndDepth itself.
        static const bool reference_normals = std::getenv("MOVBA_ADAPTER_REFERENCE_NORMALS") != nullptr;
        std::vector<Eigen::Vector3f> Ow;
        if (!reference_normals)
        {
            Ow.resize(f.kfs.size());
            for (size_t i = 0; i < f.kfs.size(); ++i) Ow[i] = f.kfs[i]->GetCameraCenter();
        }
#endif
        for (size_t k = 0; k < f.mps.size(); ++k)
        {
            MapPoint *pMP = f.mps[k];
            if (k + 4 < f.mps.size())                           // (the points sit all over the heap)
            {
                const char *nx = reinterpret_cast<const char *>(f.mps[k + 4]);
                __builtin_prefetch(nx); __builtin_prefetch(nx + 64); __builtin_prefetch(nx + 128);
            }
            const Eigen::Vector3f Pos((float)s.points[3 * k], (float)s.points[3 * k + 1], (float)s.points[3 * k + 2]);
            pMP->SetWorldPos(Pos);
#ifdef MOVBA_MAPPOINT_HAS_SET_DISTANCES
            if (!reference_normals)
            {
                if (pMP->isBad())
                    continue;                                   // UpdateNormalAndDepth returns at once for a bad point
                const size_t lp = point_local[k];
                const ObsRef *ob = f.obs_all.data() + f.obs_start[lp], *ob_end = f.obs_all.data() + f.obs_start[lp + 1];
                int32_t e = f.point_edge0[k];
                const int32_t e_end = f.point_edge0[k + 1];
                KeyFrame *pRefKF = pMP->GetReferenceKeyFrame();  // after the erasures (EraseObservation may move it)
                Eigen::Vector3f normal;
                normal.setZero();
                int n = 0, refLeft = -1, refIdx = -1, nobs_expected = 0;
                bool fallback = false, any = false;
                for (; ob != ob_end && !fallback; ++ob)
                {
                    // the observation became edge e iff its keyframe is the edge's (edges follow the observation order)
                    bool erased = false;
                    int32_t eo = -1;
                    if (e < e_end && f.edge_kf[e] == ob->kf)
                    {
                        eo = e;
                        erased = s.outlier[e] != 0;
                        ++e;
                    }
                    if (erased)
                        continue;                               // EraseObservation removed it before the update
                    any = true;
                    // (what MapPoint::AddObservation counted for this observation, MapPoint.cc:162-165: stereo twice; an edge's
                    //  mvuRight[left] was read when the edge was written)
                    const bool stereo_obs = eo >= 0 ? f.obs_right[eo] >= 0.0 : (ob->left != -1 && ob->kf->mvuRight[ob->left] >= 0);
                    nobs_expected += stereo_obs ? 2 : 1;
                    if (ob->vertex < 0 || ob->right != -1) { fallback = true; break; }
                    if (ob->left != -1)
                    {
                        const Eigen::Vector3f normali = Pos - Ow[ob->vertex];
                        normal = normal + normali / normali.norm();
                        n++;
                    }
                    if (ob->kf == pRefKF) { refLeft = ob->left; refIdx = ob->vertex; }
                }
                if (!fallback && !any)
                    continue;                                   // observations.empty(): nothing is updated
                // the observation list was copied BEFORE the solve: if another thread has added or erased an observation of
                // this point since (Tracking / LoopClosing hold no map mutex for that), MapPoint::Observations() no longer
                // matches the snapshot minus this call's erasures, and the reference's own update re-reads the point
                if (!fallback && pMP->Observations() != nobs_expected)
                    fallback = true;
                if (fallback || refIdx < 0 || refLeft < 0 || pRefKF->NLeft != -1 || n == 0)
                {
                    pMP->UpdateNormalAndDepth();
                    continue;
                }
                const Eigen::Vector3f PC = Pos - Ow[refIdx];
                const float dist = PC.norm();
                const int level = pRefKF->mvKeysUn[refLeft].octave;
                const float levelScaleFactor = pRefKF->mvScaleFactors[level];
                const int nLevels = pRefKF->mnScaleLevels;
                const float maxDistance = dist * levelScaleFactor;
                pMP->SetMinMaxDistance(maxDistance / pRefKF->mvScaleFactors[nLevels - 1], maxDistance);
                pMP->SetNormalVector(normal / n);
                continue;
            }
#endif
            pMP->UpdateNormalAndDepth();
        }
        lap("positions, normals, depth");
        for (MapPoint *pMP : edgeless)                          // estimate unchanged; the reference still casts and updates
        {
            pMP->SetWorldPos(pMP->GetWorldPos());
            pMP->UpdateNormalAndDepth();
        }
        pMap->IncreaseChangeIndex();
        tls_timing[2] = now_ms() - t_solve1;
    }

    int Optimizer::PoseOptimization(Frame *pFrame, const bool isLost, const int iterationCount, const double reprojectionError,
                                    const double reprojectErrorLost, const double confidence, const int algorithm)
    {
        // `confidence` is cv::solvePnPRansac's stopping rule: the hypothesis stage scores all iterationCount minimal samples at
        // once, and only those a sequential RANSAC over the same samples would have drawn before stopping are eligible.
        // `algorithm` picks OpenCV's sampler / scorer; the reference's configurations pass 38 = cv::USAC_MAGSAC (TartanAir.yaml:51):
        // the stage scores every hypothesis by its sigma-consensus++ loss (MAGSAC++, restated from the paper: pose_kernels.hip),
        // followed - like the USAC pipeline - by one sigma-consensus-weighted local optimisation of the winner and the final refit
        // (the four LM rounds).  Other flag values run the same pipeline: the flag itself is not interpreted.
        (void)algorithm;
        // ---- gather 3D-2D matches (Optimizer.cc:404-413) ----
        std::vector<double> Xw, obs;
        std::vector<int> indx;
        for (size_t i = 0; i < pFrame->mvpMapPoints.size(); i++)
        {
            if (pFrame->mvpMapPoints[i])
            {
                const Eigen::Vector3f wp = pFrame->mvpMapPoints[i]->GetWorldPos();
                Xw.push_back(wp(0)); Xw.push_back(wp(1)); Xw.push_back(wp(2));
                obs.push_back(pFrame->mvKeys[i].pt.x); obs.push_back(pFrame->mvKeys[i].pt.y);
                indx.push_back((int)i);
            }
        }
        if (indx.size() < 4)
            return 0;                                           // Optimizer.cc:415-418
        movba_handle *h = tls_handle.get();
        if (!h)
            return 0;

        float repError = reprojectionError;
        if (isLost)
            repError = reprojectErrorLost;

        movba_pose_desc d{};
        d.n = (int32_t)indx.size(); d.Xw = Xw.data(); d.obs = obs.data(); d.inv_sigma2 = nullptr;
        d.fx = pFrame->mpCamera->getParameter(0); d.fy = pFrame->mpCamera->getParameter(1);
        d.cx = pFrame->mpCamera->getParameter(2); d.cy = pFrame->mpCamera->getParameter(3);
        const Sophus::SE3<float> Tcw = pFrame->GetPose();       // the pose Tracking set before the call (Tracking.cc:806, 890)
        d.pose0[0] = Tcw.unit_quaternion().x(); d.pose0[1] = Tcw.unit_quaternion().y(); d.pose0[2] = Tcw.unit_quaternion().z();
        d.pose0[3] = Tcw.unit_quaternion().w();
        d.pose0[4] = Tcw.translation()(0); d.pose0[5] = Tcw.translation()(1); d.pose0[6] = Tcw.translation()(2);
        d.huber_delta = repError;                               // pixels
        d.chi2_gate = (double)repError * (double)repError;
        d.rounds = 4;
        d.its_per_round = std::max(1, std::min(10, iterationCount / 4));
        // cv::solvePnPRansac(..., useExtrinsicGuess = false, iterationCount, repError, ...) (Optimizer.cc:437): the result must
        // not depend on the pose the Frame holds (stale after tracking loss, Tracking.cc:806): iterationCount P3P hypotheses
        // scored on the GPU, the best one starts the LM
        d.ransac_iters = std::max(0, iterationCount);
        d.ransac_seed = kPoseRansacSeed;
        d.confidence = confidence;
        d.lo_iters = 10;
        std::vector<uint8_t> outl(indx.size(), 1);
        movba_pose_result r{};
        r.outlier = outl.data(); r.chi2 = nullptr;
        if (movba_pose_opt(h, &d, &r) != MOVBA_OK)
            return 0;                                           // like the empty-R early return, Optimizer.cc:442-445

        pFrame->SetPose(pose_to_se3f(r.pose));
        pFrame->mvbOutlier = std::vector<bool>(pFrame->N, true);
        for (size_t k = 0; k < indx.size(); ++k)
            if (!outl[k])
                pFrame->mvbOutlier[indx[k]] = false;
        return r.n_inliers;
    }

    void Optimizer::InertialOptimization(Map *pMap, Eigen::Matrix3d &Rwg, double &scale)
    {
        // Dead in this fork: reachable only through LocalMapping::ScaleRefinement (LocalMapping.cc:833), which
        // nothing calls, and LocalMapping is built with bInertial=false (System.cc:128).  Kept for the link.
        (void)pMap; (void)Rwg; (void)scale;
    }

} // namespace MOV_SLAM

// host-side phases of the calling thread's last LocalBundleAdjustment: extraction (window selection + flattening), the
// movba_lba_solve call, write-back under the map mutex, in ms ("timed separately and reported")
extern "C" void movba_adapter_last_timing(double out[3])
{
    for (int k = 0; k < 3; ++k) out[k] = MOV_SLAM::adapter_timing(k);
}

// What became of the calling thread's last BundleAdjustment / LocalBundleAdjustment call: 0 = optimised, > 0 = one of the
// reference's own silent returns (MOVBA_STOPPED, MOVBA_NO_FIXED, MOVBA_EMPTY), < 0 = libmovba error: optimisation skipped
// where the reference would have run it.  movba_adapter_error_count(): such skips in this process so far.
extern "C" int movba_adapter_last_status(void) { return MOV_SLAM::adapter_last_status(); }
extern "C" long movba_adapter_error_count(void) { return MOV_SLAM::adapter_error_count(); }
