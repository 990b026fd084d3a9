// Drop-in replacement for MoV-SLAM's include/Optimizer.h (/root/reference/include/Optimizer.h:45-60).
//
// The reference has no plugin/FFI layer: Tracking.cc (:688, :808, :902), LocalMapping.cc (:86, :833)
// and System.cc (:167) call these five static methods, resolved at link time inside
// libMOV_SLAM.so.  This header keeps their names, argument lists and default values, so those
// callers compile unchanged; the implementation (Optimizer.cc next to this file) flattens the
// KeyFrame / MapPoint graph and runs the solve on the GPU through the C-ABI in include/movba.h.
//
// Differences from the reference header, both invisible to the callers:
//   * the g2o solver headers the reference header pulled in (:27-36) are included only where they exist
//     (__has_include: inside the reference tree they do, so whoever relied on getting them through
//     this header still does; in this repository they do not, and nothing here needs them: the
//     arithmetic they provided lives in libmovba.so);
//   * the KeyFrameAndPose typedef (it names g2o::Sim3 and is used nowhere in the reference's
//     sources) is declared when the sim3 header was found or MOVBA_HAVE_G2O_SIM3 is defined.
#ifndef OPTIMIZER_H
#define OPTIMIZER_H

#include "Map.h"
#include "MapPoint.h"
#include "KeyFrame.h"
#include "Frame.h"

#include <math.h>

#include <map>
#include <set>
#include <utility>
#include <vector>

#if defined(__has_include)
#if __has_include("g2o/types/sim3/types_seven_dof_expmap.h") && __has_include("g2o/core/block_solver.h")
#include "g2o/types/sim3/types_seven_dof_expmap.h"
#include "g2o/core/sparse_block_matrix.h"
#include "g2o/core/block_solver.h"
#include "g2o/core/optimization_algorithm_levenberg.h"
#include "g2o/core/optimization_algorithm_gauss_newton.h"
#include "g2o/solvers/eigen/linear_solver_eigen.h"
#include "g2o/types/sba/types_six_dof_expmap.h"
#include "g2o/core/robust_kernel_impl.h"
#include "g2o/solvers/dense/linear_solver_dense.h"
#ifndef MOVBA_HAVE_G2O_SIM3
#define MOVBA_HAVE_G2O_SIM3 1
#endif
#endif
#endif
#ifdef MOVBA_HAVE_G2O_SIM3
#include "g2o/types/sim3/types_seven_dof_expmap.h"
#endif

#ifndef EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#endif

namespace MOV_SLAM
{
    typedef std::pair<std::set<KeyFrame *>, int> ConsistentGroup;
#ifdef MOVBA_HAVE_G2O_SIM3
    typedef std::map<KeyFrame *, g2o::Sim3, std::less<KeyFrame *>,
                     Eigen::aligned_allocator<std::pair<KeyFrame *const, g2o::Sim3>>>
        KeyFrameAndPose;
#endif

    class Optimizer
    {
    public:
        void static BundleAdjustment(const std::vector<KeyFrame *> &vpKF, const std::vector<MapPoint *> &vpMP,
                                     int nIterations = 5, bool *pbStopFlag = NULL, const unsigned long nLoopKF = 0,
                                     const bool bRobust = true);
        void static GlobalBundleAdjustemnt(Map *pMap, int nIterations = 5, bool *pbStopFlag = NULL,
                                           const unsigned long nLoopKF = 0, const bool bRobust = true);
        void static LocalBundleAdjustment(KeyFrame *pKF, bool *pbStopFlag, Map *pMap, int &num_fixedKF, int &num_OptKF, int &num_MPs, int &num_edges);

        static int PoseOptimization(Frame *pFrame, const bool isLost, const int iterationCount = 50, const double reprojectionError = 5.0, const double reprojectErrorLost = 8.0, const double confidence = 0.95, const int algorithm = 38);

        void static InertialOptimization(Map *pMap, Eigen::Matrix3d &Rwg, double &scale);

        EIGEN_MAKE_ALIGNED_OPERATOR_NEW;
    };

} // namespace MOV_SLAM

#endif // OPTIMIZER_H
