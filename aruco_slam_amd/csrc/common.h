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

// Wait of one wave for a value another wave of the SAME workgroup writes to LDS (both resident by construction, the writer never waits
// for the reader): polls with s_sleep, gives up after a bounded number of polls rather than hang the device.  (The CPU emulation runs
// the lanes of a workgroup as coroutines and yields instead.)
#ifndef ASLAM_SPIN_UNTIL
#define ASLAM_SPIN_UNTIL(cond) do { for (int spin_ = 0; !(cond) && spin_ < (1 << 22); spin_++) __builtin_amdgcn_s_sleep(1); } while (0)
#endif

// The dynamic LDS allocation of a launch (the CPU emulation gives every workgroup a fixed array).
#ifndef ASLAM_DYN_LDS
#define ASLAM_DYN_LDS(name) extern __shared__ __align__(16) unsigned char name[]
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
    int cut_mask;                     // pitch of the cut lattice - 1 (set per call)
    unsigned cap_starts, cap_contours, cap_points, cap_write;   // per frame (cap_starts: border nodes = start candidates + cut states; cap_write: write tickets)
};

struct Counters {                 // per-call scalars (work-queue heads, overflow mask); list sizes live in per-frame arrays
    unsigned q_trace, q_quads, n_ident, q_ident, q_write, overflow, pad[2];
};
constexpr int kCounterHeads = 5;  // leading words reset before every detection call (the overflow mask is sticky)

struct ContourRec {
    unsigned frame, scale, key, n, off;
    short sx, sy;                 // the start state the sequential scan would have used (point 0 of the contour)
    int s0;
    unsigned pad[2];
};

// Border nodes.  A border (outer or hole) is a cycle of (pixel, back-direction) states under the border-following step.  The
// nodes of a frame are the states that cut those cycles into short segments, all decidable from the 3x3 neighbourhood:
//   * the start candidates of the sequential raster scan (outer type: foreground with W / NW / N / NE background; hole type:
//     foreground with E background and NE foreground, i.e. the pixel left of a background pixel whose W and N are foreground),
//   * cut states: any state on a pixel of the cut lattice (x or y a multiple of its pitch, DetectCfg::cut_mask + 1) whose first examined neighbour
//     (direction s + 1) is background - such a state hugs a background pixel, so its cycle is a real border.
// k_threshold lists them (packed as below), k_seg walks each node's segment to the next node, k_link follows the node cycles
// (dozens of hops instead of thousands of pixel steps) to elect the canonical start and to cut the kept borders into write
// tickets, and k_trace_write replays those in parallel.
constexpr int kCutGrid = 64;      // pitch of the cut lattice of a batch: a power of two that divides the 64-pixel tile width of k_threshold and is a multiple of 4
constexpr int kCutGridSingle = 32; // ... of a call of one frame: the longest segment sets its latency, and the node count does not matter there
constexpr unsigned kNodeCut = 0u, kNodeOuter = 1u, kNodeHole = 2u, kNodeInvalid = 3u;   // type field
constexpr unsigned kNone = 0xFFFFFFFFu;
// x[0:12) y[12:24) s[24:27) scale[27:29) type[29:31)
__host__ __device__ inline unsigned pack_node(unsigned x, unsigned y, unsigned s, unsigned scale, unsigned type) {
    return x | (y << 12) | (s << 24) | (scale << 27) | (type << 29);
}
constexpr unsigned kNodeStateMask = 0x1FFFFFFFu;   // x, y, s, scale: identifies the state
constexpr unsigned kNodePixelMask = 0x18FFFFFFu;   // x, y, scale: identifies the pixel
struct NodeRec {                  // one segment: from this node's state to the next node's on the same border
    unsigned state;               // pack_node
    unsigned nxt;                 // index of the next node in the frame's list, kNone when the walk was cut (longer than max_perim)
    unsigned len;                 // border-following steps to it
    int area;                     // shoelace partial sum over those steps
};
constexpr int kWriteChunk = 64;   // points per write ticket (k_link; k_link_serial: whole segments, at least this many points)
struct WriteRec {                 // one write ticket: from `state` skip cnt >> 16 steps, then the next cnt & 0xFFFF points go to offset `rel` (cyclic) of contour `ci`
    unsigned state, ci, rel, cnt;
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


} // namespace aslam
