// Host-side launcher of the marker finalisation / pose / observation kernel (pose.hip).
#pragma once
#include "common.h"

namespace aslam {

// cv::cornerSubPix on the kept markers before the pose solve (DetectorParameters::doCornerRefinement; off in the reference)
struct RefineCfg {
    int on, win, max_iters, rows, cols;
    double eps2;                      // cornerRefinementMinAccuracy squared
    const float* mask;                // (2 win + 1)^2 window weights, built on the host with the same expf as the oracle
    const uint8_t* gray;              // tight gray planes of the call's frames
};

void launch_pose(hipStream_t st, int nframes, const FinalCand* finals, const unsigned* n_final, Marker* markers,
                 unsigned* n_markers, ObsRaw* obs, const CamParams& cam, const SlamParams& sp, Counters* ctr, const RefineCfg& rf);

} // namespace aslam
