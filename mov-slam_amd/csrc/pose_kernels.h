// Device view + launch wrapper of the pose-only optimisation kernel (pose_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace movba {

struct PoseDev {
    int32_t n, rounds, its, n_hyp;     // n_hyp > 0: RANSAC / P3P hypothesis stage in front of the LM (samples: n_hyp x 3 match indices)
    int32_t hyp_done, pad;             // the candidates were solved and scored by k_pose_hyp already (tables in `cand`)
    double fx, fy, cx, cy, huber_delta, chi2_gate;
    double pose0[7];
    double confidence;      // stopping rule of the hypothesis stage (<= 0 or >= 1: every sample is eligible)
    int32_t lo_its, pad3;   // local-optimisation step on the winning hypothesis: LM iterations of the refit on its inliers (0: off)
    const double *Xw;       // n x 3
    const double *obs;      // n x 2
    const double *isig;     // n
    double *chi2;           // n out
    uint8_t *level1;        // n out: outlier flags
    double *pose_out;       // 20: pose (7) + inlier count | inliers of the best hypothesis, its pose (7) | LM iterations run | samples the
                            // stopping rule admitted | LO refit kept (0 / 1) | inliers of the LM's start pose
    const int32_t *samples; // n_hyp x 3
    double *cand;           // device scratch for the candidate poses and scores (pose_ransac_bytes); unused when the stage runs staged in k_pose_opt
};

// staged: inputs and outputs are device views of pinned host memory, the kernel keeps the matches in LDS
// (pose_opt_staged_lds_bytes(n) must fit the 150 KB granted by configure_pose_kernels())
size_t pose_opt_staged_lds_bytes(int n, int n_hyp);
size_t pose_ransac_bytes(int n_hyp);
hipError_t configure_pose_kernels();
hipError_t launch_pose_hyp(const PoseDev &p, hipStream_t s);      // the hypothesis stage as a grid of its own (inputs in device memory)
hipError_t launch_pose_opt(const PoseDev &p, bool staged, hipStream_t s);

}  // namespace movba
