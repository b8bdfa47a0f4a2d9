// Host-side launcher of the marker finalisation / pose / observation kernel (pose.hip).
#pragma once
#include "common.h"

namespace aslam {

void launch_pose(hipStream_t st, int nframes, const FinalCand* finals, const unsigned* n_final, Marker* markers,
                 unsigned* n_markers, ObsRaw* obs, const CamParams& cam, const SlamParams& sp, Counters* ctr);

} // namespace aslam
