// Mock of include/MapPoint.h (:81-82, :89, :93, :99, :114, :121, :130, :150, :157-158).
#pragma once
#include <map>
#include <mutex>
#include <tuple>
#include <vector>
#include "mock_math.h"
#include "Map.h"
namespace MOV_SLAM {
class KeyFrame;
class MapPoint {
public:
    // (the accessors lock what the reference's lock, MapPoint.cc: the adapter's host-side timings are only meaningful with them)
    void SetWorldPos(const Eigen::Vector3f &p) { std::unique_lock<std::mutex> lock2(mGlobalMutex()); std::unique_lock<std::mutex> lock(mMutexPos); mWorldPos = p; }   // :113-118
    Eigen::Vector3f GetWorldPos() { std::unique_lock<std::mutex> lock(mMutexPos); return mWorldPos; }
    std::map<KeyFrame *, std::tuple<int, int>> GetObservations() { std::unique_lock<std::mutex> lock(mMutexFeatures); ++nObservationCopies(); return mObservations; }
    static long &nObservationCopies() { static long n = 0; return n; }      // test plumbing: std::map copies handed out
#ifdef MOVBA_MAPPOINT_HAS_FOR_EACH_OBSERVATION
    // the second accessor INTEGRATION.md offers MoV-SLAM's MapPoint.h: the observations visited in map order under the
    // point's lock, without the std::map copy (and its node allocations) GetObservations() hands out
    template <class F> void ForEachObservation(F &&f) {
        std::unique_lock<std::mutex> lock(mMutexFeatures);
        ++nObservationVisits();
        for (const auto &o : mObservations) f(o.first, std::get<0>(o.second), std::get<1>(o.second));
    }
#endif
    static long &nObservationVisits() { static long n = 0; return n; }       // test plumbing
    // MapPoint.cc:347-354
    std::tuple<int, int> GetIndexInKeyFrame(KeyFrame *pKF) {
        std::unique_lock<std::mutex> lock(mMutexFeatures);
        const auto it = mObservations.find(pKF);
        return it != mObservations.end() ? it->second : std::tuple<int, int>(-1, -1);
    }
    KeyFrame *GetReferenceKeyFrame() { std::unique_lock<std::mutex> lock(mMutexFeatures); return mpRefKF; }                     // MapPoint.h:87
    Eigen::Vector3f GetNormal() { std::unique_lock<std::mutex> lock(mMutexPos); return mNormalVector; }
    void SetNormalVector(const Eigen::Vector3f &n) { std::unique_lock<std::mutex> lock3(mMutexPos); mNormalVector = n; }     // MapPoint.h:85
#ifdef MOVBA_MAPPOINT_HAS_SET_DISTANCES
    // the accessor INTEGRATION.md asks MoV-SLAM's MapPoint.h to gain (the reference can only set these inside UpdateNormalAndDepth)
    void SetMinMaxDistance(float mn, float mx) { std::unique_lock<std::mutex> lock(mMutexPos); mfMinDistance = mn; mfMaxDistance = mx; }
#endif
    void UpdateNormalAndDepth();                                             // MapPoint.cc:362-435, restated on the mock types below
    // MapPoint.cc:139-169: nObs counts a stereo observation (mvuRight >= 0, no second camera) twice
    void AddObservation(KeyFrame *pKF, int idx);
    int Observations() { std::unique_lock<std::mutex> lock(mMutexFeatures); return nObs; }                                     // MapPoint.cc:224-228
    // MapPoint.cc:171-209: the reference keyframe moves on when its observation goes; nObs <= 2 afterwards: SetBadFlag()
    void EraseObservation(KeyFrame *pKF);
    // MapPoint.cc:230-258: the point leaves the map: observations cleared, its slot nulled in every observer
    void SetBadFlag();
    bool isBad() { std::unique_lock<std::mutex> lock1(mMutexFeatures, std::defer_lock); std::unique_lock<std::mutex> lock2(mMutexPos, std::defer_lock); std::lock(lock1, lock2); return mbBad; }   // :314-320
    Map *GetMap() { std::unique_lock<std::mutex> lock(mMutexMap); return mpMap; }
    long unsigned int mnId = 0, mnBALocalForKF = ~0ul, mnBAGlobalForKF = 0;
    Eigen::Vector3f mPosGBA;
    std::mutex mMutexPos, mMutexFeatures, mMutexMap;
    static std::mutex &mGlobalMutex() { static std::mutex m; return m; }
    // test plumbing
    Eigen::Vector3f mWorldPos; std::map<KeyFrame *, std::tuple<int, int>> mObservations;
    bool mbBad = false; Map *mpMap = nullptr; int nErased = 0, nNormalUpdates = 0, nObs = 0;
    std::vector<KeyFrame *> vErasedBy;      // EraseObservation calls that found their observation, in call order
    KeyFrame *mpRefKF = nullptr; Eigen::Vector3f mNormalVector; float mfMinDistance = 0.f, mfMaxDistance = 0.f;
};
}  // namespace MOV_SLAM
