// Exercises the Optimizer adapter (Optimizer.cc) against the mock map classes: builds a map from a
// flattened window file written by tests/test_gpu_adapter.py, calls the reference-signature entry
// points, and dumps what the map received.  Needs a GPU at run time (no CPU fallback).
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <vector>

#include "Optimizer.h"

using namespace MOV_SLAM;

extern "C" void movba_adapter_last_timing(double out[3]);
extern "C" int movba_adapter_last_status(void);
extern "C" long movba_adapter_error_count(void);

// KeyFrame's keypoints, right coordinates and baseline are const members in the reference (KeyFrame.h:315-324), filled by its
// constructors and, when a keyframe is deserialised, through const_cast (KeyFrame.h:68-88): the test map is filled the same way
template <typename T> static T &fill(const T &x) { return const_cast<T &>(x); }

template <typename T> static std::vector<T> rd(FILE *f, size_t n)
{
    std::vector<T> v(n);
    if (n && fread(v.data(), sizeof(T), n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(2); }
    return v;
}

static int run_lba(const char *in, const char *out, bool global)
{
    FILE *f = fopen(in, "rb");
    if (!f) return 2;
    const std::vector<int32_t> hd = rd<int32_t>(f, 4);
    const int NP = hd[0], P = hd[1], E = hd[2];
    const int bad_kf = hd[3] - 1;                  // (4th header word: 1 + index of a keyframe to flag bad, 0 = none)
    const std::vector<uint8_t> fixed = rd<uint8_t>(f, NP);
    const std::vector<double> poses = rd<double>(f, 7 * (size_t)NP), points = rd<double>(f, 3 * (size_t)P);
    const std::vector<int32_t> ep = rd<int32_t>(f, E), el = rd<int32_t>(f, E);
    const std::vector<double> obs = rd<double>(f, 2 * (size_t)E);
    // optional stereo trailer: bf, then mvuRight per edge (< 0 = monocular)
    double bf = 0.0;
    std::vector<double> obs_right;
    if (fread(&bf, sizeof(double), 1, f) == 1) obs_right = rd<double>(f, E);
    // optional camera trailer (behind the stereo one): the word 0x43414d31, then fx fy cx cy per keyframe and mbf per keyframe -
    // every keyframe then has a GeometricCamera of its own (e->pCamera = pKFi->mpCamera, Optimizer.cc:664)
    std::vector<double> cam_kf, bf_kf;
    int32_t magic = 0;
    if (fread(&magic, sizeof(int32_t), 1, f) == 1 && magic == 0x43414d31) { cam_kf = rd<double>(f, 4 * (size_t)NP); bf_kf = rd<double>(f, NP); }
    fclose(f);

    // $MOVBA_ADAPTER_REPS: the call repeated on a freshly built copy of the map; the LAST repetition is written out (its
    // timing is the steady state of a running system: the adapter's per-thread buffers, the handle's arena and staging
    // buffer and the crew's threads exist from the first call on)
    const int reps = std::getenv("MOVBA_ADAPTER_REPS") ? std::max(1, std::atoi(std::getenv("MOVBA_ADAPTER_REPS"))) : 1;
    for (int rep = 0; rep < reps; ++rep) {
    Map map;
    GeometricCamera cam({320.f, 320.f, 320.f, 240.f});
    std::vector<GeometricCamera> cams;             // (one per keyframe when the window file carries a camera trailer)
    for (int i = 0; i < NP && !cam_kf.empty(); ++i)
        cams.emplace_back(std::vector<float>{(float)cam_kf[4 * i], (float)cam_kf[4 * i + 1], (float)cam_kf[4 * i + 2], (float)cam_kf[4 * i + 3]});
    std::vector<KeyFrame> kfs(NP);                 // contiguous: pointer order == id order == std::map order
    std::vector<MapPoint> mps(P);
    for (int i = 0; i < NP; ++i) {
        KeyFrame &k = kfs[i];
        k.mnId = 100 + i; k.mpMap = &map; k.mpCamera = cams.empty() ? &cam : &cams[i]; fill(k.mbf) = (float)(bf_kf.empty() ? bf : bf_kf[i]);
        k.mTcw = Sophus::SE3f(Eigen::Quaternionf((float)poses[7 * i + 3], (float)poses[7 * i], (float)poses[7 * i + 1], (float)poses[7 * i + 2]),
                              Eigen::Vector3f((float)poses[7 * i + 4], (float)poses[7 * i + 5], (float)poses[7 * i + 6]));
        map.mvKFs.push_back(&k);
    }
    if (bad_kf >= 0 && bad_kf < NP) kfs[bad_kf].mbBad = true;
    for (int l = 0; l < P; ++l) {
        mps[l].mnId = 5000 + l; mps[l].mpMap = &map;
        mps[l].mWorldPos = Eigen::Vector3f((float)points[3 * l], (float)points[3 * l + 1], (float)points[3 * l + 2]);
        map.mvMPs.push_back(&mps[l]);
    }
    for (int e = 0; e < E; ++e) {
        KeyFrame &k = kfs[ep[e]];
        cv::KeyPoint kp; kp.pt.x = (float)obs[2 * e]; kp.pt.y = (float)obs[2 * e + 1]; kp.octave = 0;
        const int idx = (int)k.mvKeysUn.size();
        fill(k.mvKeysUn).push_back(kp); fill(k.mvuRight).push_back(obs_right.empty() ? -1.f : (float)obs_right[e]); k.mvpMapPoints.push_back(&mps[el[e]]);
        mps[el[e]].AddObservation(&k, idx);               // (nObs as MapPoint.cc:139-169 counts it: a stereo observation twice)
        if (!mps[el[e]].mpRefKF) mps[el[e]].mpRefKF = &k;           // the keyframe that created the point
    }
    // the newest free keyframe is the one LocalMapping passes in; every other free keyframe is covisible
    KeyFrame *pKF = nullptr;
    for (int i = NP - 1; i >= 0; --i) if (!fixed[i]) { pKF = &kfs[i]; break; }
    for (int i = 0; i < NP; ++i) if (!fixed[i] && &kfs[i] != pKF) pKF->mvCovisible.push_back(&kfs[i]);
    // global BA fixes the init keyframe only; local BA must not see it among the local ones
    map.mnInitKFid = global ? kfs[0].mnId : 1;
    map.mpOriginKF = &kfs[0];

    int num_fixedKF = 0, num_OptKF = 0, num_MPs = 0, num_edges = 0;
    bool stop = false;
    MapPoint::nObservationCopies() = 0;
    if (global) Optimizer::GlobalBundleAdjustemnt(&map, 10, &stop, kfs[0].mnId, true);
    else Optimizer::LocalBundleAdjustment(pKF, &stop, &map, num_fixedKF, num_OptKF, num_MPs, num_edges);

    if (std::getenv("MOVBA_ADAPTER_TIMING")) {
        double tmr[3] = { 0, 0, 0 };
        if (!global) movba_adapter_last_timing(tmr);
        std::fprintf(stderr, "adapter[rep %d]: extraction %.3f ms, solve call %.3f ms, write-back %.3f ms (NP=%d P=%d E=%d)\n", rep, tmr[0], tmr[1], tmr[2], NP, P, E);
    }
    if (rep + 1 < reps) continue;
    // the (keyframe, map point) pairs EraseObservation removed, per point in call order (a point that went bad on the way —
    // nObs <= 2, SetBadFlag — has lost its remaining observations without further calls)
    std::vector<int32_t> erased;
    for (int l = 0; l < P; ++l)
        for (KeyFrame *k : mps[l].vErasedBy) { erased.push_back((int32_t)(k - kfs.data())); erased.push_back(l); }
    FILE *o = fopen(out, "wb");
    const int32_t oh[5] = { num_fixedKF, num_OptKF, num_edges, (int32_t)(erased.size() / 2), map.mnChangeIdx };
    fwrite(oh, sizeof(int32_t), 5, o);
    for (int i = 0; i < NP; ++i) {
        const Sophus::SE3f T = kfs[i].GetPose();
        const float v[7] = { T.unit_quaternion().x(), T.unit_quaternion().y(), T.unit_quaternion().z(), T.unit_quaternion().w(),
                             T.translation()(0), T.translation()(1), T.translation()(2) };
        fwrite(v, sizeof(float), 7, o);
    }
    for (int l = 0; l < P; ++l) { const Eigen::Vector3f X = mps[l].GetWorldPos(); fwrite(X.v, sizeof(float), 3, o); }
    if (!erased.empty()) fwrite(erased.data(), sizeof(int32_t), erased.size(), o);
    int nposes = 0, nnorm = 0;
    for (int i = 0; i < NP; ++i) nposes += kfs[i].nPoseSets;
    for (int l = 0; l < P; ++l) nnorm += mps[l].nNormalUpdates;
    const int32_t tail[2] = { nposes, nnorm };
    fwrite(tail, sizeof(int32_t), 2, o);
    // what UpdateNormalAndDepth (or the adapter's own normal / depth pass) left in the points, and the accessor traffic
    int ncenter = 0;
    for (int i = 0; i < NP; ++i) ncenter += kfs[i].nCenterReads;
    const int32_t counts[2] = { (int32_t)MapPoint::nObservationCopies(), ncenter };
    fwrite(counts, sizeof(int32_t), 2, o);
    for (int l = 0; l < P; ++l) {
        const float v[5] = { mps[l].mNormalVector(0), mps[l].mNormalVector(1), mps[l].mNormalVector(2), mps[l].mfMinDistance, mps[l].mfMaxDistance };
        fwrite(v, sizeof(float), 5, o);
    }
    double tm[3] = { 0, 0, 0 };
    if (!global) movba_adapter_last_timing(tm);
    fwrite(tm, sizeof(double), 3, o);
    // per point: bad flag, nObs, observations left, index of the reference keyframe; per keyframe: matches still set; then the
    // adapter's status plumbing
    for (int l = 0; l < P; ++l) {
        const int32_t v[4] = { mps[l].mbBad ? 1 : 0, mps[l].nObs, (int32_t)mps[l].mObservations.size(), mps[l].mpRefKF ? (int32_t)(mps[l].mpRefKF - kfs.data()) : -1 };
        fwrite(v, sizeof(int32_t), 4, o);
    }
    for (int i = 0; i < NP; ++i) {
        int32_t live = 0;
        for (MapPoint *m : kfs[i].mvpMapPoints) live += m != nullptr;
        fwrite(&live, sizeof(int32_t), 1, o);
    }
    const int32_t st[3] = { movba_adapter_last_status(), (int32_t)movba_adapter_error_count(), (int32_t)map.mspErased.size() };
    fwrite(st, sizeof(int32_t), 3, o);
    fclose(o);
    }
    return 0;
}

static int run_pose(const char *in, const char *out)
{
    FILE *f = fopen(in, "rb");
    if (!f) return 2;
    const std::vector<int32_t> hd = rd<int32_t>(f, 2);
    const int n = hd[0], isLost = hd[1];
    const std::vector<double> Xw = rd<double>(f, 3 * (size_t)n), obs = rd<double>(f, 2 * (size_t)n), pose0 = rd<double>(f, 7);
    fclose(f);
    Map map;
    GeometricCamera cam({320.f, 320.f, 320.f, 240.f});
    std::vector<MapPoint> mps(n);
    Frame fr;
    fr.N = n + 3; fr.mpCamera = &cam;
    fr.mvKeys.resize(fr.N); fr.mvpMapPoints.assign(fr.N, nullptr); fr.mvbOutlier.assign(fr.N, false);
    for (int i = 0; i < n; ++i) {
        mps[i].mWorldPos = Eigen::Vector3f((float)Xw[3 * i], (float)Xw[3 * i + 1], (float)Xw[3 * i + 2]);
        fr.mvpMapPoints[i] = &mps[i];
        fr.mvKeys[i].pt.x = (float)obs[2 * i]; fr.mvKeys[i].pt.y = (float)obs[2 * i + 1];
    }
    fr.mTcw = Sophus::SE3f(Eigen::Quaternionf((float)pose0[3], (float)pose0[0], (float)pose0[1], (float)pose0[2]),
                           Eigen::Vector3f((float)pose0[4], (float)pose0[5], (float)pose0[6]));
    const int ninl = Optimizer::PoseOptimization(&fr, isLost != 0);
    FILE *o = fopen(out, "wb");
    const int32_t oh[1] = { ninl };
    fwrite(oh, sizeof(int32_t), 1, o);
    const Sophus::SE3f T = fr.GetPose();
    const float v[7] = { T.unit_quaternion().x(), T.unit_quaternion().y(), T.unit_quaternion().z(), T.unit_quaternion().w(),
                         T.translation()(0), T.translation()(1), T.translation()(2) };
    fwrite(v, sizeof(float), 7, o);
    std::vector<uint8_t> ob(fr.N);
    for (int i = 0; i < fr.N; ++i) ob[i] = fr.mvbOutlier[i] ? 1 : 0;
    fwrite(ob.data(), 1, ob.size(), o);
    fclose(o);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc != 4) { std::fprintf(stderr, "usage: adapter_test lba|gba|pose in.bin out.bin\n"); return 2; }
    if (!std::strcmp(argv[1], "lba")) return run_lba(argv[2], argv[3], false);
    if (!std::strcmp(argv[1], "gba")) return run_lba(argv[2], argv[3], true);
    if (!std::strcmp(argv[1], "pose")) return run_pose(argv[2], argv[3]);
    return 2;
}
