// Host-side launchers of the detection kernels (detect.hip).
#pragma once
#include "common.h"

namespace aslam {

void launch_threshold(hipStream_t st, const uint8_t* in, int channels, size_t frame_stride, size_t row_step, int nframes,
                      uint8_t* gray, uint8_t* nbr, const DetectCfg& cfg, unsigned long long* starts, Counters* ctr);
void launch_trace(hipStream_t st, int nwaves, const uint8_t* nbr, const DetectCfg& cfg, const unsigned long long* starts,
                  Counters* ctr, ContourRec* contours, unsigned* points);
void launch_quads(hipStream_t st, int nwaves, const DetectCfg& cfg, Counters* ctr, const ContourRec* contours,
                  const unsigned* points, CandRec* cands, unsigned* n_cand);
void launch_assemble(hipStream_t st, int nframes, const DetectCfg& cfg, Counters* ctr, const CandRec* cands,
                     const unsigned* n_cand, FinalCand* finals, unsigned* n_final, IdentWork* work);
void launch_identify(hipStream_t st, int nwaves, const DetectCfg& cfg, Counters* ctr, const uint8_t* gray,
                     FinalCand* finals, const IdentWork* work, const unsigned long long* dict_codes);

} // namespace aslam
