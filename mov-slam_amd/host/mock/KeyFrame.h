// Mock of include/KeyFrame.h (:154, :172, :198, :201, :218, :233, :254, :270-271, :293, :298, :323-324, :340, :441; mbf :312).
#pragma once
#include <vector>
#include "mock_math.h"
#include "GeometricCamera.h"
#include "Map.h"
#include "MapPoint.h"
namespace MOV_SLAM {
class KeyFrame {
public:
    Sophus::SE3f GetPose() { return mTcw; }
    void SetPose(const Sophus::SE3f &T) { mTcw = T; ++nPoseSets; }
    std::vector<KeyFrame *> GetVectorCovisibleKeyFrames() { return mvCovisible; }
    std::vector<MapPoint *> GetMapPointMatches() { return mvpMapPoints; }
    void EraseMapPointMatch(MapPoint *pMP) {
        for (auto &p : mvpMapPoints) if (p == pMP) p = nullptr;
    }
    bool isBad() { return mbBad; }
    Map *GetMap() { return mpMap; }
    long unsigned int mnId = 0, mnBALocalForKF = ~0ul, mnBAFixedForKF = ~0ul, mnBAGlobalForKF = 0;
    Sophus::SE3f mTcwGBA;
    std::vector<cv::KeyPoint> mvKeysUn;
    std::vector<float> mvuRight;
    float mbf = 0.f;
    std::vector<float> mvInvLevelSigma2 = std::vector<float>(8, 1.0f);
    GeometricCamera *mpCamera = nullptr, *mpCamera2 = nullptr;
    // test plumbing
    Sophus::SE3f mTcw; std::vector<KeyFrame *> mvCovisible; std::vector<MapPoint *> mvpMapPoints;
    bool mbBad = false; Map *mpMap = nullptr; int nPoseSets = 0;
};
}  // namespace MOV_SLAM
