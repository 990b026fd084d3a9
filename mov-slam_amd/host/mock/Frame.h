// Mock of include/Frame.h (:184, :230, :311, :319, :327, :337, :411).
#pragma once
#include <vector>
#include "mock_math.h"
#include "GeometricCamera.h"
#include "MapPoint.h"
namespace MOV_SLAM {
class Frame {
public:
    void SetPose(const Sophus::SE3<float> &Tcw) { mTcw = Tcw; }
    Sophus::SE3<float> GetPose() const { return mTcw; }
    int N = 0;
    std::vector<cv::KeyPoint> mvKeys;
    std::vector<MapPoint *> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    GeometricCamera *mpCamera = nullptr;
    Sophus::SE3f mTcw;
};
}  // namespace MOV_SLAM
