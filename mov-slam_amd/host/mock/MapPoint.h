// Mock of include/MapPoint.h (:81-82, :89, :93, :99, :114, :121, :130, :150, :157-158).
#pragma once
#include <map>
#include <tuple>
#include "mock_math.h"
#include "Map.h"
namespace MOV_SLAM {
class KeyFrame;
class MapPoint {
public:
    void SetWorldPos(const Eigen::Vector3f &p) { mWorldPos = p; }
    Eigen::Vector3f GetWorldPos() { return mWorldPos; }
    std::map<KeyFrame *, std::tuple<int, int>> GetObservations() { return mObservations; }
    void EraseObservation(KeyFrame *pKF) { mObservations.erase(pKF); ++nErased; }
    bool isBad() { return mbBad; }
    void UpdateNormalAndDepth() { ++nNormalUpdates; }
    Map *GetMap() { return mpMap; }
    long unsigned int mnId = 0, mnBALocalForKF = ~0ul, mnBAGlobalForKF = 0;
    Eigen::Vector3f mPosGBA;
    // test plumbing
    Eigen::Vector3f mWorldPos; std::map<KeyFrame *, std::tuple<int, int>> mObservations;
    bool mbBad = false; Map *mpMap = nullptr; int nErased = 0, nNormalUpdates = 0;
};
}  // namespace MOV_SLAM
