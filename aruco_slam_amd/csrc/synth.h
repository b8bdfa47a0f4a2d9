// Synthetic frame renderer (synth.hip) — input generation only.
#pragma once
#include "common.h"

namespace aslam {

struct SynthMarker {
    double Hinv[9];              // normalised image coords (x, y, 1) -> marker plane (X, Y, W)
    int bbox[4];                 // x0, y0, x1, y1 (inclusive) of the quiet-zone square in pixels
    unsigned long long bits[2];  // (markerSize+2)^2 cells row-major, 1 = white
};

void launch_render(hipStream_t st, uint8_t* out, int rows, int cols, double fx, double fy, double cx, double cy, int n_markers,
                   const SynthMarker* mk, int nc, double marker_length, int background, int noise_amp, unsigned seed, int ss);

} // namespace aslam
