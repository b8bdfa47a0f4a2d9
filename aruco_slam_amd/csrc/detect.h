// Host-side launchers of the detection kernels (detect.hip).
#pragma once
#include "common.h"

namespace aslam {

void launch_threshold(hipStream_t st, const uint8_t* in, int channels, size_t frame_stride, size_t row_step, int nframes,
                      uint8_t* gray, uint8_t* nbr, const DetectCfg& cfg, unsigned* starts, unsigned* n_starts, unsigned* nodeplane,
                      Counters* ctr);
void launch_clear_counts(hipStream_t st, int nframes, Counters* ctr, unsigned* n_starts, unsigned* n_contours, unsigned* n_points, unsigned* n_write,
                         unsigned* n_cand);
void launch_prefix(hipStream_t st, int nframes, const unsigned* counts, unsigned cap, unsigned per_ticket, unsigned* pre);
void launch_seg(hipStream_t st, int nwaves, const uint8_t* nbr, const DetectCfg& cfg, int nframes, const unsigned* starts,
                const unsigned* n_starts, const unsigned* nodeplane, const unsigned* pre, Counters* ctr, NodeRec* nodes);
void launch_link(hipStream_t st, const DetectCfg& cfg, int nframes, const unsigned* n_starts, Counters* ctr, const NodeRec* nodes,
                 unsigned* link_todo, ContourRec* contours, unsigned* n_contours, unsigned* n_points, WriteRec* wlist, unsigned* n_write);
void launch_trace_write(hipStream_t st, int nblocks, const uint8_t* nbr, const DetectCfg& cfg, int nframes, const unsigned* pre,
                        Counters* ctr, const ContourRec* contours, const WriteRec* wlist, const unsigned* n_write, unsigned* points);
void launch_quads(hipStream_t st, int nwaves, const DetectCfg& cfg, int nframes, Counters* ctr, const ContourRec* contours,
                  const unsigned* n_contours, const unsigned* pre, const unsigned* points, CandRec* cands, unsigned* n_cand);
void launch_assemble(hipStream_t st, int nframes, const DetectCfg& cfg, Counters* ctr, const CandRec* cands,
                     const unsigned* n_cand, FinalCand* finals, unsigned* n_final, IdentWork* work);
void launch_identify(hipStream_t st, int nwaves, const DetectCfg& cfg, Counters* ctr, const uint8_t* gray,
                     FinalCand* finals, const IdentWork* work, const unsigned long long* dict_codes);
int max_frames_per_call();

} // namespace aslam
