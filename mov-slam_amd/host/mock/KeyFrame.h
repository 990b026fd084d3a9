// Mock of include/KeyFrame.h (:154, :172, :198, :201, :218, :233, :254, :270-271, :293, :298, :323-324, :340, :441; mbf :312).
#pragma once
#include <mutex>
#include <vector>
#include "mock_math.h"
#include "GeometricCamera.h"
#include "Map.h"
#include "MapPoint.h"
namespace MOV_SLAM {
class KeyFrame {
public:
    // (locks as in KeyFrame.cc)
    Sophus::SE3f GetPose() { std::unique_lock<std::mutex> lock(mMutexPose); return mTcw; }
    void SetPose(const Sophus::SE3f &T) { std::unique_lock<std::mutex> lock(mMutexPose); mTcw = T; ++nPoseSets; }
    // KeyFrame.h:155: mOw = Twc.translation() = -Rcw^T tcw, in float like Sophus::SE3f::inverse()
    Eigen::Vector3f GetCameraCenter() {
        std::unique_lock<std::mutex> lock(mMutexPose);
        ++nCenterReads;
        const float x = mTcw.q.qx, y = mTcw.q.qy, z = mTcw.q.qz, w = mTcw.q.qw;
        const float R[9] = { 1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                             2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                             2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y) };
        const Eigen::Vector3f &t = mTcw.t;
        return Eigen::Vector3f(-(R[0] * t(0) + R[3] * t(1) + R[6] * t(2)), -(R[1] * t(0) + R[4] * t(1) + R[7] * t(2)),
                               -(R[2] * t(0) + R[5] * t(1) + R[8] * t(2)));
    }
    Eigen::Vector3f GetRightCameraCenter() { return GetCameraCenter(); }
    const int NLeft = -1, NRight = -1;                       // KeyFrame.h:452 (no fisheye rig in this fork)
    const int mnScaleLevels = 8;                             // KeyFrame.h:335
    const std::vector<float> mvScaleFactors = std::vector<float>{1.f, 1.2f, 1.44f, 1.728f, 2.0736f, 2.48832f, 2.985984f, 3.5831808f};
    const std::vector<cv::KeyPoint> mvKeys, mvKeysRight;     // KeyFrame.h:322, :450 (const: the test plumbing fills them as the reference's deserialiser does, through const_cast)
    int nCenterReads = 0;
    std::vector<KeyFrame *> GetVectorCovisibleKeyFrames() { std::unique_lock<std::mutex> lock(mMutexConnections); return mvCovisible; }
    std::vector<MapPoint *> GetMapPointMatches() { std::unique_lock<std::mutex> lock(mMutexFeatures); return mvpMapPoints; }
    // KeyFrame.cc:298-306: the indices come from the point's own observation entry (not a search through the matches)
    void EraseMapPointMatch(MapPoint *pMP) {
        const std::tuple<int, int> indexes = pMP->GetIndexInKeyFrame(this);
        const int leftIndex = std::get<0>(indexes), rightIndex = std::get<1>(indexes);
        if (leftIndex != -1) mvpMapPoints[leftIndex] = nullptr;
        if (rightIndex != -1) mvpMapPoints[rightIndex] = nullptr;
    }
    void EraseMapPointMatch(const int &idx) { std::unique_lock<std::mutex> lock(mMutexFeatures); mvpMapPoints[idx] = nullptr; }   // KeyFrame.cc:292-296
    bool isBad() { std::unique_lock<std::mutex> lock(mMutexConnections); return mbBad; }
    Map *GetMap() { std::unique_lock<std::mutex> lock(mMutexMap); return mpMap; }
    long unsigned int mnId = 0, mnBALocalForKF = ~0ul, mnBAFixedForKF = ~0ul, mnBAGlobalForKF = 0;
    Sophus::SE3f mTcwGBA;
    const std::vector<cv::KeyPoint> mvKeysUn;
    const std::vector<float> mvuRight;
    const float mbf = 0.f;
    const std::vector<float> mvInvLevelSigma2 = std::vector<float>(8, 1.0f);
    GeometricCamera *mpCamera = nullptr, *mpCamera2 = nullptr;
    std::mutex mMutexPose, mMutexConnections, mMutexFeatures, mMutexMap;
    // test plumbing
    Sophus::SE3f mTcw; std::vector<KeyFrame *> mvCovisible; std::vector<MapPoint *> mvpMapPoints;
    bool mbBad = false; Map *mpMap = nullptr; int nPoseSets = 0;
};
inline void MapPoint::AddObservation(KeyFrame *pKF, int idx)
{
    std::unique_lock<std::mutex> lock(mMutexFeatures);
    std::tuple<int, int> indexes = mObservations.count(pKF) ? mObservations[pKF] : std::tuple<int, int>(-1, -1);
    if (pKF->NLeft != -1 && idx >= pKF->NLeft) std::get<1>(indexes) = idx; else std::get<0>(indexes) = idx;
    mObservations[pKF] = indexes;
    if (!pKF->mpCamera2 && pKF->mvuRight[idx] >= 0) nObs += 2; else nObs++;
}
inline void MapPoint::EraseObservation(KeyFrame *pKF)
{
    bool bBad = false;
    {
        std::unique_lock<std::mutex> lock(mMutexFeatures);
        if (mObservations.count(pKF)) {
            const std::tuple<int, int> indexes = mObservations[pKF];
            const int leftIndex = std::get<0>(indexes), rightIndex = std::get<1>(indexes);
            if (leftIndex != -1) { if (!pKF->mpCamera2 && pKF->mvuRight[leftIndex] >= 0) nObs -= 2; else nObs--; }
            if (rightIndex != -1) nObs--;
            mObservations.erase(pKF); ++nErased; vErasedBy.push_back(pKF);
            if (mpRefKF == pKF && !mObservations.empty()) mpRefKF = mObservations.begin()->first;   // (the reference dereferences begin() of an empty map here)
            if (nObs <= 2) bBad = true;
        }
    }
    if (bBad) SetBadFlag();
}
inline void MapPoint::SetBadFlag()
{
    std::map<KeyFrame *, std::tuple<int, int>> obs;
    {
        std::unique_lock<std::mutex> lock1(mMutexFeatures);
        std::unique_lock<std::mutex> lock2(mMutexPos);
        mbBad = true;
        obs = mObservations;
        mObservations.clear();
    }
    for (auto mit = obs.begin(); mit != obs.end(); ++mit) {
        KeyFrame *pKF = mit->first;
        const int leftIndex = std::get<0>(mit->second), rightIndex = std::get<1>(mit->second);
        if (leftIndex != -1) pKF->EraseMapPointMatch(leftIndex);
        if (rightIndex != -1) pKF->EraseMapPointMatch(rightIndex);
    }
    mpMap->EraseMapPoint(this);
}
// MapPoint::UpdateNormalAndDepth as the reference computes it (MapPoint.cc:362-435): the mean of the unit viewing rays of
// all observers, and the scale-invariance distances from the reference keyframe.  Defined here because it needs KeyFrame.
inline void MapPoint::UpdateNormalAndDepth()
{
    ++nNormalUpdates;
    std::map<KeyFrame *, std::tuple<int, int>> observations;
    KeyFrame *pRefKF;
    Eigen::Vector3f Pos;
    {
        std::unique_lock<std::mutex> lock1(mMutexFeatures);       // MapPoint.cc:367-375
        std::unique_lock<std::mutex> lock2(mMutexPos);
        if (mbBad) return;
        observations = mObservations;
        pRefKF = mpRefKF;
        Pos = mWorldPos;
    }
    if (observations.empty()) return;
    Eigen::Vector3f normal; normal.setZero();
    int n = 0;
    for (auto mit = observations.begin(); mit != observations.end(); ++mit) {
        KeyFrame *pKF = mit->first;
        const int leftIndex = std::get<0>(mit->second), rightIndex = std::get<1>(mit->second);
        if (leftIndex != -1) { const Eigen::Vector3f normali = Pos - pKF->GetCameraCenter(); normal = normal + normali / normali.norm(); n++; }
        if (rightIndex != -1) { const Eigen::Vector3f normali = Pos - pKF->GetRightCameraCenter(); normal = normal + normali / normali.norm(); n++; }
    }
    const Eigen::Vector3f PC = Pos - pRefKF->GetCameraCenter();
    const float dist = PC.norm();
    const std::tuple<int, int> indexes = observations.count(pRefKF) ? observations.at(pRefKF) : std::make_tuple(0, -1);
    const int leftIndex = std::get<0>(indexes);
    const int level = pRefKF->mvKeysUn[leftIndex].octave;                    // NLeft == -1
    const float levelScaleFactor = pRefKF->mvScaleFactors[level];
    const int nLevels = pRefKF->mnScaleLevels;
    std::unique_lock<std::mutex> lock3(mMutexPos);                  // :429-434
    mfMaxDistance = dist * levelScaleFactor;
    mfMinDistance = mfMaxDistance / pRefKF->mvScaleFactors[nLevels - 1];
    mNormalVector = normal / n;
}
}  // namespace MOV_SLAM
