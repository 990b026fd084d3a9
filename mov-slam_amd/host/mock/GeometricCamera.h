// Mock of include/CameraModels/GeometricCamera.h: float parameters [fx, fy, cx, cy] (:81, :101).
#pragma once
#include <vector>
namespace MOV_SLAM {
class GeometricCamera {
public:
    explicit GeometricCamera(const std::vector<float> &p) : mvParameters(p) {}
    float getParameter(const int i) { return mvParameters[i]; }
protected:
    std::vector<float> mvParameters;
};
}  // namespace MOV_SLAM
