// Deterministic synthetic frame renderer (input generation for tests and bench.py — NOT part of the hot
// path and not a restatement of anything in the reference).  Renders planar ArUco markers with a one-cell
// white quiet zone on a flat background through an ideal pinhole camera, box-filter anti-aliased.
// Only +,-,*,/ on doubles and integers are used per sample, so the image is bit-identical wherever it runs.
#include "common.h"
#include "synth.h"

namespace aslam {

__global__ __launch_bounds__(256) void k_render(uint8_t* __restrict__ out, int rows, int cols, double fx, double fy, double cx,
                                                double cy, int n_markers, const SynthMarker* __restrict__ mk, int nc,
                                                double marker_length, int background, int noise_amp, unsigned seed, int ss) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x < cols && y < rows) {
        const double half = marker_length * 0.5;
        const double cell = marker_length / nc;
        const double ext = half + cell;
        const int nsamp = ss * ss;
        int acc = background * nsamp;
        for (int m = 0; m < n_markers; m++) {
            const SynthMarker& M = mk[m];
            if (x < M.bbox[0] || x > M.bbox[2] || y < M.bbox[1] || y > M.bbox[3]) continue;
            for (int sy = 0; sy < ss; sy++)
                for (int sx = 0; sx < ss; sx++) {
                    double u = x + (sx + 0.5) / ss - 0.5, v = y + (sy + 0.5) / ss - 0.5;
                    double xn = (u - cx) / fx, yn = (v - cy) / fy;
                    double W = M.Hinv[6] * xn + M.Hinv[7] * yn + M.Hinv[8];
                    if (W == 0.0) continue;
                    double X = (M.Hinv[0] * xn + M.Hinv[1] * yn + M.Hinv[2]) / W;
                    double Y = (M.Hinv[3] * xn + M.Hinv[4] * yn + M.Hinv[5]) / W;
                    if (X < -ext || X > ext || Y < -ext || Y > ext) continue;
                    int val = 255;
                    if (X > -half && X < half && Y > -half && Y < half) {
                        int c = (int)((X + half) / cell), r = (int)((half - Y) / cell);
                        c = min(max(c, 0), nc - 1);
                        r = min(max(r, 0), nc - 1);
                        int bit = r * nc + c;
                        unsigned long long w = bit < 64 ? M.bits[0] >> bit : M.bits[1] >> (bit - 64);
                        val = (w & 1ull) ? 255 : 0;
                    }
                    acc += val - background;
                }
        }
        int v = (acc + nsamp / 2) / nsamp;
        if (noise_amp > 0) {
            unsigned h = seed * 0x9E3779B1u + (unsigned)(y * cols + x) * 0x85EBCA6Bu;
            h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
            v += (int)(h % (unsigned)(2 * noise_amp + 1)) - noise_amp;
        }
        out[(size_t)y * cols + x] = (uint8_t)min(max(v, 0), 255);
    }
}

void launch_render(hipStream_t st, uint8_t* out, int rows, int cols, double fx, double fy, double cx, double cy, int n_markers,
                   const SynthMarker* mk, int nc, double marker_length, int background, int noise_amp, unsigned seed, int ss) {
    dim3 grid((cols + 63) / 64, (rows + 3) / 4);
    hipLaunchKernelGGL(k_render, grid, dim3(256), 0, st, out, rows, cols, fx, fy, cx, cy, n_markers, mk, nc, marker_length,
                       background, noise_amp, seed, ss);
}

} // namespace aslam
