// Mock of include/Map.h with the members the Optimizer touches (:80-81, :89, :93, :107, :137, :151-152).
#pragma once
#include <mutex>
#include <set>
#include <vector>
namespace MOV_SLAM {
class KeyFrame;
class MapPoint;
class Map {
public:
    std::vector<KeyFrame *> GetAllKeyFrames() { return mvKFs; }
    std::vector<MapPoint *> GetAllMapPoints() { return mvMPs; }
    long unsigned int GetInitKFid() { return mnInitKFid; }
    KeyFrame *GetOriginKF() { return mpOriginKF; }
    void IncreaseChangeIndex() { ++mnChangeIdx; }
    void EraseMapPoint(MapPoint *pMP) { std::unique_lock<std::mutex> lock(mMutexMap); mspErased.insert(pMP); }      // Map.cc: the point leaves mspMapPoints
    std::mutex mMutexMap; std::set<MapPoint *> mspErased;
    std::mutex mMutexMapUpdate;
    std::set<long unsigned int> msOptKFs, msFixedKFs;
    // test plumbing
    std::vector<KeyFrame *> mvKFs; std::vector<MapPoint *> mvMPs;
    long unsigned int mnInitKFid = 0; KeyFrame *mpOriginKF = nullptr; int mnChangeIdx = 0;
};
}  // namespace MOV_SLAM
