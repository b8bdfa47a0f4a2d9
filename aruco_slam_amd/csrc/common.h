// Shared host/device definitions for the MI355X (gfx950) ArUco EKF-SLAM hot path.
// Replaces the role of OpenCV/Eigen in the reference's ArucoSlam (src/aruco_slam.cpp:21-74,
// 76-263, 307-376); see DESIGN.md for the data layout in HBM and the kernel list.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstddef>

// Workgroup barrier that orders LDS traffic only: `s_waitcnt lgkmcnt(0)` + `s_barrier`.  __syncthreads() also waits for every
// outstanding global store (vmcnt(0)), which a kernel that streams a log to HBM between its LDS phases must not pay per barrier.
// (The CPU emulation used by the tests defines it as its own barrier.)
#ifndef ASLAM_LDS_BARRIER
#define ASLAM_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

namespace aslam {

constexpr int kScales = 3;        // adaptive-threshold windows 3, 13, 23 (DetectorParameters defaults)
constexpr int kCandMax = 2048;    // quad candidates kept per frame
constexpr int kMarkerMax = 128;   // identified markers / observations per frame
constexpr int kDictMaxCells = 9;  // markerSize + 2 border cells <= 9 (7x7 dictionaries)
constexpr int kCellPx = 8;        // largest perspectiveRemovePixelPerCell (= the 3.2.0 default; 4 from 3.3)

// overflow / error bits reported by the device in Counters::overflow
enum : unsigned {
    kOvfStarts = 1u, kOvfContours = 2u, kOvfPoints = 4u, kOvfCands = 8u, kOvfMarkers = 16u, kOvfLandmarks = 32u, kOvfUpdates = 64u,
};

// Everything the detector kernels need to know about one batch (host-filled, passed by value).
struct DetectCfg {
    int rows, cols, pitch;            // pitch of the neighbour-mask planes (multiple of 64)
    int win_r[kScales];               // box radius per scale: 1, 6, 11 by default
    int n_scales;                     // threshold windows in use (1..kScales)
    int thresh_c;                     // floor(adaptiveThreshConstant) = 7
    int min_perim, max_perim;         // contour point-count bounds
    double approx_rate;               // polygonalApproxAccuracyRate (0.05)
    double min_corner_rate;           // minCornerDistanceRate (0.05)
    double min_marker_dist_rate;      // minMarkerDistanceRate (0.05)
    int min_border_dist;              // minDistanceToBorder (3)
    int marker_size;                  // dictionary markerSize (5)
    int border_bits;                  // markerBorderBits (1)
    int cell_px;                      // perspectiveRemovePixelPerCell (<= kCellPx)
    int cell_margin;                  // int(0.13 * cellSize) = 1
    int max_border_err;               // int(ms*ms*0.35)
    int max_corr;                     // int(maxCorrectionBits * errorCorrectionRate)
    int n_dict;                       // markers in the dictionary
    double min_otsu_std;              // 5.0
    unsigned cap_starts, cap_contours, cap_points, cap_ckpt;   // per frame
    int ckpt_per_walk;                // checkpoints one border walk can leave: max_perim / kCkptStride + 2
};

struct Counters {                 // per-call scalars (work-queue heads, overflow mask); list sizes live in per-frame arrays
    unsigned q_trace, q_quads, n_ident, q_ident, q_write, overflow, pad[2];
};
constexpr int kCounterHeads = 5;  // leading words reset before every detection call (the overflow mask is sticky)

struct ContourRec {
    unsigned frame, scale, key, n, off;
    short sx, sy;                 // the start state the sequential scan would have used (point 0 of the contour)
    int s0;
    unsigned ck_off;              // first of its ceil(n / kCkptStride) checkpoints in the frame's checkpoint list
    int kpos;                     // step of the closing walk at which it stood on (sx, sy, s0): point i = walk step kpos + i (mod n)
};

constexpr int kCkptStride = 64;   // border-walk steps between two checkpoints (= steps one lane of k_trace_write replays)
struct CkptRec {                  // walk state every kCkptStride steps of a kept contour: x[0:12) y[12:24) s[24:27), and its contour
    unsigned state, ci;
};

struct CandRec {                  // quad that passed _findMarkerContours
    short x[4], y[4];
    unsigned n;                   // contour point count ("perimeter" in 3.2.0)
    unsigned ordkey;              // scale * 2^22 + (2^22 - 1 - key): ascending = OpenCV candidate order
};

struct FinalCand {                // candidate after _reorderCandidatesCorners + _filterTooCloseCandidates
    float c[8];
    int n;
    int id;                       // >= 0 once identified, -1 rejected
    int pad[2];
};

struct IdentWork { unsigned frame, idx; };

struct Marker {                   // one detection: what detectMarkers + estimatePoseSingleMarkers return
    int id;
    int pad;
    float c[8];
    double rvec[3], tvec[3];
};

struct ObsRaw {                   // getObservations loop body, before landmark lookup (aruco_slam.cpp:325-369)
    int id;
    int valid;                    // 0 = dropped by the range / covariance gates
    double x, y, th;
    double r[3];                  // diagonal of observe_covariance_
};

struct CamParams {
    double fx, fy, cx, cy;
    double k[5];
    int nD;
    int pad;
};

struct SlamParams {
    double Q_k, R_x, R_y, R_theta, kl, kr, b, marker_length;
    double r2c_tx, r2c_ty;
    float useful_distance_threshold;
    int pad;
};

// Neighbour-mask planes are stored in 8x8-pixel tiles (one 64-byte line per tile) so that a border walk, which moves one
// pixel at a time in any direction, stays on the same cache line for several steps.  pitch is a multiple of 64.
__host__ __device__ inline size_t nbr_index(int x, int y, int pitch) {
    // 32-bit arithmetic: a plane is rows x pitch < 2^32 bytes (the border walks pay for every instruction of a step)
    return (size_t)((((unsigned)(y >> 3) * (unsigned)(pitch >> 3) + (unsigned)(x >> 3)) << 6) + (unsigned)((y & 7) << 3) + (unsigned)(x & 7));
}
__host__ __device__ inline size_t nbr_plane_bytes(int rows, int pitch) { return (size_t)((rows + 7) & ~7) * (size_t)pitch; }

// start-list entry: x[0:12) y[12:24) scale[24:26) type[26] frame[32:48)
__host__ __device__ inline unsigned long long pack_start(unsigned x, unsigned y, unsigned scale, unsigned type, unsigned frame) {
    return (unsigned long long)x | ((unsigned long long)y << 12) | ((unsigned long long)scale << 24) |
           ((unsigned long long)type << 26) | ((unsigned long long)frame << 32);
}

} // namespace aslam
