// C-ABI of libaruco_slam_hip.so (include/aruco_slam_hip.h): context, device buffers, stream orchestration.
// Host logic only; all arithmetic of the hot path runs in the kernels of detect.hip / pose.hip / ekf.hip.
#include "common.h"
#include "detect.h"
#include "pose.h"
#include "ekf.h"
#include "synth.h"
#include "../../include/aruco_slam_hip.h"
#include <string>
#include <vector>
#include <queue>
#include <cfloat>
#include <fstream>
#include <sstream>
#include <cctype>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <dlfcn.h>
#include <cstdio>
#include <cmath>
#include <algorithm>
#include <chrono>
#include <iterator>

using namespace aslam;

namespace {

constexpr int kWinWidenFrames = 16;     // a window of at least this many frames is not widened to the next image size (64 -> 128 -> 192)
constexpr int kWinChainFrames = 8;      // frames per chain kernel of a window (its log is replayed meanwhile); <= kWinPieceMax

enum ProfId { P_THRESH, P_SEG, P_LINK, P_WRITE, P_QUADS, P_ASSEMBLE, P_IDENTIFY, P_POSE, P_EKF_PLAN, P_EKF_GATHER, P_EKF_SMALL,
              P_EKF_T, P_EKF_UPDATE, P_EKF_MID, P_EKF_APPLY, P_EKF_MID64, P_EKF_WIN_CHAIN, P_EKF_WIN_SCAN, P_EKF_WIN_FLUSH, P_EKF_WIN_NEXT, P_COUNT };
const char* kProfNames[P_COUNT] = {"k_threshold", "k_seg", "k_link", "k_trace_write", "k_quads", "k_assemble", "k_identify", "k_pose",
                                   "k_ekf_plan", "k_ekf_gather", "k_ekf_small", "k_ekf_T", "k_ekf_update_mfma", "k_ekf_mid", "k_ekf_apply",
                                   "k_ekf_mid64", "k_ekf_win_step", "k_ekf_win_drain", "k_ekf_win_flush", "k_ekf_win_next"};

struct ProfSpan { int id; hipEvent_t a, b; hipStream_t st; };

} // namespace

struct aslam_ctx {
    aslam_init init{};
    hipStream_t stream = nullptr;         // detection + pose (batched over frames)
    hipStream_t stream_part = nullptr;    // detection beside an EKF chain: CU-masked so that part of every XCD stays free for the chain
    hipStream_t last_detect = nullptr;    // stream of the most recent detection (ordering when it changes)
    hipEvent_t ev_export[2] = {nullptr, nullptr};   // aslam_export_map_async: the records of buffer 0 / 1 are in place
    hipStream_t stream_copy = nullptr;    // host-fed stream: uploads from the pinned ring
    hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_det[2] = {nullptr, nullptr};   // per ring half: upload done / detection done with the slots
    bool ev_det_set[2] = {false, false}, ev_up_set[2] = {false, false};
    uint8_t* h_ring = nullptr;            // two half batches of page-locked frames
    std::vector<double> ring_enc;         // encoder samples of the half being filled
    int ring_H = 0, ring_half = 0, ring_fill = 0;
    bool ring_acquired = false;
    hipStream_t stream_ekf = nullptr;     // EKF chain (sequential over frames); overlaps the next batch's detection
    hipEvent_t ev_detect = nullptr, ev_ekf = nullptr;
    int ekf_first = 0, ekf_count = 0;     // slots the in-flight EKF chain still reads
    std::string err;
    bool have_cam = false;
    CamParams cam{};
    SlamParams sp{};

    // staged batch
    int rows = 0, cols = 0, channels = 0;
    int max_batch = 0, nwaves = 0;
    DetectCfg cfg{};
    int last_first = 0, last_count = 0;

    uint8_t* d_in = nullptr;
    uint8_t* d_gray = nullptr;
    uint8_t* d_nbr = nullptr;
    unsigned* d_starts = nullptr;          // per frame: cap_starts entries
    unsigned* d_nstarts = nullptr;
    Counters* d_ctr = nullptr;
    ContourRec* d_contours = nullptr;      // per frame: cap_contours records
    unsigned* d_ncontours = nullptr;
    unsigned* d_points = nullptr;          // per frame: cap_points packed (x, y)
    unsigned* d_npoints = nullptr;
    unsigned* d_nodeplane = nullptr;       // per frame and scale: index of a pixel's first border node (tiled like the mask planes; only entries of pixels that carry nodes are ever written or read)
    NodeRec* d_nodes = nullptr;            // per frame: cap_starts segments (k_seg)
    WriteRec* d_wlist = nullptr;           // per frame: cap_write write tickets (k_link)
    unsigned* d_nwrite = nullptr;
    unsigned* d_link_todo = nullptr;       // per frame: left to k_link_serial
    unsigned* d_pre_write = nullptr;
    unsigned* d_pre_trace = nullptr;       // ticket ranges of the work-queue kernels
    unsigned* d_pre_quads = nullptr;
    CandRec* d_cands = nullptr;
    unsigned* d_ncand = nullptr;
    FinalCand* d_finals = nullptr;
    unsigned* d_nfinal = nullptr;
    IdentWork* d_work = nullptr;
    unsigned long long* d_dict = nullptr;
    Marker* d_markers = nullptr;
    unsigned* d_nmarkers = nullptr;
    ObsRaw* d_obs = nullptr;
    double* d_enc = nullptr;          // per slot: wl, wr, dt
    std::vector<double> enc_host;
    SynthMarker* d_synth = nullptr;
    size_t in_frame_bytes = 0, pitch = 0;
    int dict_ms = 5, dict_n = 1024, dict_maxcorr = 0;
    float* d_refine_mask = nullptr;       // cornerSubPix window weights (15 x 15 at most)
    aslam_detector_params dp{};           // cv::aruco::DetectorParameters in force (defaults of 3.2.0 unless aslam_set_detector_params)
    std::vector<unsigned long long> dict_cells;   // per id: (ms+2)^2 cell image incl. border (for the renderer)

    EkfState ekf{};
    double last_time = 0;
    bool is_init = false;

    // windowed EKF (ekf_window.hip): the observations of a batch come back to the host, which cuts the batch into runs of
    // frames that fuse the same landmarks; the batch's EKF work is enqueued one call later (or at the next synchronisation),
    // behind the NEXT batch's detection, so that the detection stream never waits for the host
    bool win_enabled = true;
    int win_piece = kWinChainFrames;      // frames per chain piece (ASLAM_WIN_PIECE, 1..kWinChainFrames: a test knob)
    bool win_no_early = false;            // ASLAM_WIN_NO_EARLY: every window waits for its own flush (test / comparison knob)
    struct Pending { bool active = false; int first = 0, count = 0, ev = 0; } pend;
    hipEvent_t ev_obs[2] = {nullptr, nullptr}, ev_idx = nullptr;
    hipStream_t stream_win = nullptr;     // scan / flush of the windows, beside the chain on stream_ekf
    hipEvent_t ev_win[64] = {};
    unsigned win_count = 0;               // windows enqueued so far (parity = which hand-over image a window uses)
    unsigned ev_win_next = 0;
    int ev_obs_next = 0;
    bool ev_idx_set = false;
    ObsRaw* h_obs = nullptr;              // pinned: max_batch x kMarkerMax
    unsigned* h_nm = nullptr;             // pinned: max_batch
    WinFrame* h_win_frames = nullptr;     // pinned: max_batch frame plans (uploaded to ekf.d_win_frames per batch)
    std::vector<int> m_id2idx;            // host mirror of the id -> landmark table, valid unless mirror_dirty
    int m_L = 0;
    struct HostLast { int id; double z[3]; };
    std::vector<HostLast> m_last;         // host mirror of last_observed_marker_ (NaN z = unset)
    bool mirror_dirty = true;             // the device planned frames the host could not follow: read the tables back before planning
    int ekf_lo = 0, ekf_hi = 0;           // union of the slot ranges of EKF work enqueued since the last wait on ev_ekf
    double last_timing[6] = {0, 0, 0, 0, 0, 0};   // aslam_add_image, host clock, microseconds: upload, detection enqueue, EKF enqueue, wait, read-back / checks, total
    long long plan_stats[4] = {0, 0, 0, 0};   // frames inside windows, frames on the per-frame chain, windows, frames left to the device's own plan

    // map gather over RCCL without torch (aslam_comm_*): librccl is dlopen'ed on first use
    void* comm = nullptr;
    int comm_world = 0, comm_rank = 0;
    uint8_t* d_gather = nullptr;          // world x max_landmarks records

    bool prof_on = false;
    std::vector<ProfSpan> spans;
    double prof_empty_ms = 0.0;           // what an event pair around NOTHING measures on the EKF stream (subtracted per span)
    int prof_calls[P_COUNT] = {0};
    double prof_ms[P_COUNT] = {0};
};

extern "C" int finalize_pending(aslam_ctx* c);      // defined with run_staged below (internal, not part of the C-ABI header)

namespace {

int fail(aslam_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

#define HIP_TRY(c, expr)                                                                                  \
    do {                                                                                                  \
        hipError_t e__ = (expr);                                                                          \
        if (e__ != hipSuccess) return fail((c), ASLAM_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)

template <class T> hipError_t dalloc(T** p, size_t count) { return hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)); }

// DICT_ARUCO_ORIGINAL from first principles: 5 rows x 2 id bits (MSB first) through the words
// 10000 / 10111 / 01001 / 01110 (original ArUco library).  parameters.yaml:16 selects enum 16.
void make_dict_aruco_original(std::vector<unsigned long long>& codes, std::vector<unsigned long long>& cells) {
    static const int words[4] = {0x10, 0x17, 0x09, 0x0e};
    const int n = 5;
    codes.resize(1024 * 4);
    cells.resize(1024 * 2);
    for (int id = 0; id < 1024; id++) {
        int B[5][5];
        for (int y = 0; y < n; y++) {
            int val = words[(id >> (2 * (4 - y))) & 3];
            for (int x = 0; x < n; x++) B[y][x] = (val >> (4 - x)) & 1;
        }
        unsigned long long c[4] = {0, 0, 0, 0};
        for (int row = 0; row < n; row++)
            for (int col = 0; col < n; col++) {
                c[0] = (c[0] << 1) | (unsigned)B[row][col];
                c[1] = (c[1] << 1) | (unsigned)B[col][n - 1 - row];
                c[2] = (c[2] << 1) | (unsigned)B[n - 1 - row][n - 1 - col];
                c[3] = (c[3] << 1) | (unsigned)B[n - 1 - col][row];
            }
        for (int r = 0; r < 4; r++) codes[(size_t)id * 4 + r] = c[r];
        unsigned long long lo = 0;     // 7x7 cells incl. black border, bit index = r*7 + c
        for (int row = 0; row < n; row++)
            for (int col = 0; col < n; col++)
                if (B[row][col]) lo |= 1ull << ((row + 1) * 7 + col + 1);
        cells[(size_t)id * 2] = lo;
        cells[(size_t)id * 2 + 1] = 0;
    }
}

void prof_begin(aslam_ctx* c, int id, hipStream_t st) {
    if (!c->prof_on) return;
    ProfSpan s;
    s.id = id;
    s.st = st;
    hipEventCreate(&s.a);
    hipEventCreate(&s.b);
    hipEventRecord(s.a, st);
    c->spans.push_back(s);
}
void prof_end(aslam_ctx* c) {
    if (!c->prof_on) return;
    hipEventRecord(c->spans.back().b, c->spans.back().st);
}
void prof_collect(aslam_ctx* c) {
    for (ProfSpan& s : c->spans) {
        float ms = 0;
        hipEventSynchronize(s.b);
        hipEventElapsedTime(&ms, s.a, s.b);
        c->prof_calls[s.id]++;
        c->prof_ms[s.id] += ms;
        hipEventDestroy(s.a);
        hipEventDestroy(s.b);
    }
    c->spans.clear();
}

// codes (4 rotations per id, row-major bit string, first bit most significant - the order k_identify assembles the sampled
// bits in) and the (ms+2)^2 cell image incl. the black border (renderer) from bits[n][ms*ms], 1 = white
void make_dict_from_bits(int n, int nm, const uint8_t* bits, std::vector<unsigned long long>& codes, std::vector<unsigned long long>& cells) {
    codes.assign((size_t)nm * 4, 0ull);
    cells.assign((size_t)nm * 2, 0ull);
    const int nc = n + 2;
    for (int id = 0; id < nm; id++) {
        const uint8_t* B = bits + (size_t)id * n * n;
        unsigned long long c[4] = {0, 0, 0, 0};
        for (int row = 0; row < n; row++)
            for (int col = 0; col < n; col++) {
                c[0] = (c[0] << 1) | (unsigned)(B[row * n + col] & 1);
                c[1] = (c[1] << 1) | (unsigned)(B[col * n + (n - 1 - row)] & 1);
                c[2] = (c[2] << 1) | (unsigned)(B[(n - 1 - row) * n + (n - 1 - col)] & 1);
                c[3] = (c[3] << 1) | (unsigned)(B[(n - 1 - col) * n + row] & 1);
                if (B[row * n + col] & 1) {
                    const int bit = (row + 1) * nc + col + 1;
                    cells[(size_t)id * 2 + (bit >> 6)] |= 1ull << (bit & 63);
                }
            }
        for (int r = 0; r < 4; r++) codes[(size_t)id * 4 + r] = c[r];
    }
}

int sync_streams(aslam_ctx* c);

int configure_frames(aslam_ctx* c, int rows, int cols, int channels) {
    if (rows <= 0 || cols <= 0 || rows > c->init.max_rows || cols > c->init.max_cols || rows > 4095 || cols > 4095)
        return fail(c, ASLAM_E_INVALID, "frame size outside [1, max_rows x max_cols]");
    if (channels != 1 && channels != 3) return fail(c, ASLAM_E_INVALID, "channels must be 1 (gray) or 3 (bgr8)");
    c->rows = rows; c->cols = cols; c->channels = channels;
    DetectCfg& g = c->cfg;
    g.rows = rows; g.cols = cols;
    g.pitch = (cols + 63) / 64 * 64;
    // cv::aruco::DetectorParameters: OpenCV 3.2.0 defaults (the reference passes none, aruco_slam.cpp:313) unless replaced
    const aslam_detector_params& dp = c->dp;
    {
        // aruco.cpp::_detectInitialCandidates: nScales = (max - min) / step + 1 windows min + i step, even sizes bumped to odd
        const int ns = (dp.adaptiveThreshWinSizeMax - dp.adaptiveThreshWinSizeMin) / dp.adaptiveThreshWinSizeStep + 1;
        g.n_scales = ns;
        for (int i = 0; i < kScales; i++) {
            int win = dp.adaptiveThreshWinSizeMin + i * dp.adaptiveThreshWinSizeStep;
            if (win % 2 == 0) win++;
            g.win_r[i] = i < ns ? win / 2 : 0;
        }
    }
    g.thresh_c = (int)std::floor(dp.adaptiveThreshConstant);             // THRESH_BINARY_INV: src - mean <= -floor(C)
    g.min_perim = (int)(unsigned)(dp.minMarkerPerimeterRate * std::max(cols, rows));
    g.max_perim = (int)(unsigned)(dp.maxMarkerPerimeterRate * std::max(cols, rows));
    g.approx_rate = dp.polygonalApproxAccuracyRate;
    g.min_corner_rate = dp.minCornerDistanceRate;
    g.min_marker_dist_rate = dp.minMarkerDistanceRate;
    g.min_border_dist = dp.minDistanceToBorder;
    g.marker_size = c->dict_ms;
    g.border_bits = dp.markerBorderBits;
    g.cell_px = dp.perspectiveRemovePixelPerCell;
    g.cell_margin = (int)(dp.perspectiveRemoveIgnoredMarginPerCell * g.cell_px);
    g.max_border_err = (int)(c->dict_ms * c->dict_ms * dp.maxErroneousBitsInBorderRate);
    g.max_corr = (int)((double)c->dict_maxcorr * dp.errorCorrectionRate);
    g.n_dict = c->dict_n;
    g.min_otsu_std = dp.minOtsuStdDev;
    g.cap_starts = c->init.cap_starts_per_frame;
    g.cap_contours = c->init.cap_contours_per_frame;
    g.cap_points = c->init.cap_points_per_frame;
    g.cap_write = g.cap_points / kWriteChunk + g.cap_contours;
    return ASLAM_OK;
}

int check_slot_range(aslam_ctx* c, int first, int count) {
    if (first < 0 || count <= 0 || first + count > c->max_batch) return fail(c, ASLAM_E_INVALID, "slot range outside [0, max_batch)");
    return ASLAM_OK;
}

// Before the host rewrites the inputs of slots [slot0, slot0 + n) (frames, encoder samples, injected observations): work that
// still reads them must have finished.  A batch whose EKF work is still pending (run_staged defers it by one call) reads its
// encoder samples and observations only when it is finalised, and the chain kernels read d_enc when they run.
int quiesce_slots(aslam_ctx* c, int slot0, int n) {
    if (c->pend.active && slot0 < c->pend.first + c->pend.count && c->pend.first < slot0 + n) {
        int r = finalize_pending(c);
        if (r) return r;
    }
    if (c->ekf_count > 0 && slot0 < c->ekf_hi && c->ekf_lo < slot0 + n) {
        HIP_TRY(c, hipEventSynchronize(c->ev_ekf));
        c->ekf_count = 0;
    }
    if (c->last_detect && slot0 < c->last_first + c->last_count && c->last_first < slot0 + n) HIP_TRY(c, hipEventSynchronize(c->ev_detect));
    return ASLAM_OK;
}

// detection + pose for `count` staged frames starting at slot `first` (asynchronous on the stream)
int run_detect(aslam_ctx* c, int first, int count, bool beside_ekf = false, hipEvent_t wait_before = nullptr) {
    if (c->rows == 0) return fail(c, ASLAM_E_STATE, "no frames staged");
    if (!c->have_cam) return fail(c, ASLAM_E_STATE, "camera parameters not set (aslam_set_camera)");
    // the CU-masked stream only pays off while an EKF chain is actually in flight beside this detection; the first batch after
    // a synchronisation gets the whole GPU
    hipStream_t st = (beside_ekf && c->stream_part && (c->ekf_count > 0 || c->pend.active)) ? c->stream_part : c->stream;
    if (c->last_detect && c->last_detect != st) HIP_TRY(c, hipStreamWaitEvent(st, c->ev_detect, 0));   // slots / work lists are shared
    c->last_detect = st;
    if (wait_before) HIP_TRY(c, hipStreamWaitEvent(st, wait_before, 0));          // frames still in flight on the copy stream
    DetectCfg g = c->cfg;
    // one frame (the drop-in call): a finer cut lattice - its longest segment sets the latency of k_seg / k_trace_write, and the extra nodes
    // (x 1.6) still fit k_link's LDS image for frames up to a megapixel or so
    g.cut_mask = (count == 1 && (size_t)g.rows * g.cols <= (size_t)1200 * 1000 ? kCutGridSingle : kCutGrid) - 1;
    const size_t frame_px = (size_t)g.rows * g.cols;
    const bool alias_gray = c->channels == 1;              // staged gray frames are tight: the detector reads them in place
    c->last_first = first;
    c->last_count = count;
    if (c->ekf_count > 0 && first < c->ekf_hi && c->ekf_lo < first + count) {
        HIP_TRY(c, hipStreamWaitEvent(st, c->ev_ekf, 0));   // EKF work in flight still reads observations of these slots
        c->ekf_count = 0;
    }
    // frames per launch of the detection kernels: everything of the call at once (the work queues balance it) unless
    // ASLAM_DETECT_CHUNK asks for smaller sub-batches (an experiment knob: mask planes of fewer frames stay cache-resident
    // between k_threshold and k_trace, at the price of one longest-walk tail per sub-batch)
    static const int chunk_env = [] { const char* e = std::getenv("ASLAM_DETECT_CHUNK"); return e ? std::max(1, std::atoi(e)) : 0; }();
    const int chunk = chunk_env > 0 ? std::min(chunk_env, max_frames_per_call()) : max_frames_per_call();
    for (int f0 = first; f0 < first + count; f0 += chunk) {
        const int nf = std::min(chunk, first + count - f0);
        // queue heads, work count and the per-frame list sizes of these frames, in one launch (the overflow mask is sticky)
        launch_clear_counts(st, nf, c->d_ctr, c->d_nstarts + f0, c->d_ncontours + f0, c->d_npoints + f0, c->d_nwrite + f0, c->d_ncand + f0);
        const uint8_t* in = c->d_in + (size_t)f0 * c->in_frame_bytes;
        uint8_t* nbr = c->d_nbr + (size_t)f0 * kScales * nbr_plane_bytes(g.rows, g.pitch);
        const uint8_t* gray = alias_gray ? in : c->d_gray + (size_t)f0 * frame_px;
        unsigned* starts = c->d_starts + (size_t)f0 * g.cap_starts;
        ContourRec* contours = c->d_contours + (size_t)f0 * g.cap_contours;
        unsigned* points = c->d_points + (size_t)f0 * g.cap_points;
        prof_begin(c, P_THRESH, st);
        launch_threshold(st, in, c->channels, c->in_frame_bytes, (size_t)g.cols * c->channels, nf,
                         alias_gray ? nullptr : c->d_gray + (size_t)f0 * frame_px, nbr, g, starts, c->d_nstarts + f0,
                         c->d_nodeplane + (size_t)f0 * kScales * nbr_plane_bytes(g.rows, g.pitch), c->d_ctr);
        prof_end(c);
        prof_begin(c, P_SEG, st);
        launch_prefix(st, nf, c->d_nstarts + f0, g.cap_starts, 1u, c->d_pre_trace);
        NodeRec* nodes = c->d_nodes + (size_t)f0 * g.cap_starts;
        WriteRec* wlist = c->d_wlist + (size_t)f0 * g.cap_write;
        launch_seg(st, c->nwaves, nbr, g, nf, starts, c->d_nstarts + f0, c->d_nodeplane + (size_t)f0 * kScales * nbr_plane_bytes(g.rows, g.pitch),
                   c->d_pre_trace, c->d_ctr, nodes);
        prof_end(c);
        prof_begin(c, P_LINK, st);
        launch_link(st, g, nf, c->d_nstarts + f0, c->d_ctr, nodes, c->d_link_todo + f0, contours, c->d_ncontours + f0, c->d_npoints + f0, wlist,
                    c->d_nwrite + f0);
        prof_end(c);
        prof_begin(c, P_WRITE, st);
        launch_prefix(st, nf, c->d_nwrite + f0, g.cap_write, 1u, c->d_pre_write);
        launch_trace_write(st, std::max(64, c->nwaves / 4), nbr, g, nf, c->d_pre_write, c->d_ctr, contours, wlist, c->d_nwrite + f0, points);
        prof_end(c);
        prof_begin(c, P_QUADS, st);
        launch_prefix(st, nf, c->d_ncontours + f0, g.cap_contours, 1u, c->d_pre_quads);
        launch_quads(st, c->nwaves, g, nf, c->d_ctr, contours, c->d_ncontours + f0, c->d_pre_quads, points,
                     c->d_cands + (size_t)f0 * kCandMax, c->d_ncand + f0);
        prof_end(c);
        prof_begin(c, P_ASSEMBLE, st);
        launch_assemble(st, nf, g, c->d_ctr, c->d_cands + (size_t)f0 * kCandMax, c->d_ncand + f0,
                        c->d_finals + (size_t)f0 * kCandMax, c->d_nfinal + f0, c->d_work);
        prof_end(c);
        prof_begin(c, P_IDENTIFY, st);
        launch_identify(st, c->nwaves, g, c->d_ctr, gray, c->d_finals + (size_t)f0 * kCandMax, c->d_work, c->d_dict);
        prof_end(c);
        prof_begin(c, P_POSE, st);
        launch_pose(st, nf, c->d_finals + (size_t)f0 * kCandMax, c->d_nfinal + f0, c->d_markers + (size_t)f0 * kMarkerMax,
                    c->d_nmarkers + f0, c->d_obs + (size_t)f0 * kMarkerMax, c->cam, c->sp, c->d_ctr,
                    RefineCfg{c->dp.doCornerRefinement ? 1 : 0, c->dp.cornerRefinementWinSize, std::min(std::max(c->dp.cornerRefinementMaxIterations, 1), 100),
                              g.rows, g.cols, std::max(c->dp.cornerRefinementMinAccuracy, 0.0) * std::max(c->dp.cornerRefinementMinAccuracy, 0.0),
                              c->d_refine_mask, gray});
        prof_end(c);
    }
    HIP_TRY(c, hipEventRecord(c->ev_detect, st));
    HIP_TRY(c, hipGetLastError());
    return ASLAM_OK;
}

int run_ekf_frame(aslam_ctx* c, int slot, double wl, double wr, double dt, bool do_predict) {
    hipStream_t st = c->stream_ekf;
    const bool fast = c->init.max_updates_per_frame <= ekf_fast_max_updates();
    prof_begin(c, P_EKF_PLAN, st);
    launch_ekf_plan(st, c->ekf, c->sp, wl, wr, dt, do_predict ? 1 : 0, c->d_obs + (size_t)slot * kMarkerMax, c->d_nmarkers + slot, c->d_ctr,
                    c->init.max_updates_per_frame, slot);
    prof_end(c);
    if (fast) {
        prof_begin(c, P_EKF_MID, st);
        launch_ekf_mid(st, c->ekf);
        prof_end(c);
        prof_begin(c, P_EKF_APPLY, st);
        launch_ekf_apply(st, c->ekf);
        prof_end(c);
    } else if (c->init.max_updates_per_frame <= ekf_mid_max_updates()) {
        prof_begin(c, P_EKF_MID64, st);
        launch_ekf_mid64(st, c->ekf);
        prof_end(c);
        prof_begin(c, P_EKF_T, st);
        launch_ekf_T(st, c->ekf);
        prof_end(c);
        prof_begin(c, P_EKF_UPDATE, st);
        launch_ekf_update_mfma(st, c->ekf);
        prof_end(c);
    } else {
        prof_begin(c, P_EKF_GATHER, st);
        launch_ekf_gather(st, c->ekf);
        prof_end(c);
        prof_begin(c, P_EKF_SMALL, st);
        launch_ekf_small(st, c->ekf);
        prof_end(c);
        prof_begin(c, P_EKF_T, st);
        launch_ekf_T(st, c->ekf);
        prof_end(c);
        prof_begin(c, P_EKF_UPDATE, st);
        launch_ekf_update_mfma(st, c->ekf);
        prof_end(c);
    }
    HIP_TRY(c, hipGetLastError());
    return ASLAM_OK;
}

int sync_streams(aslam_ctx* c) {
    { int r = finalize_pending(c); if (r) return r; }
    if (c->stream_copy) HIP_TRY(c, hipStreamSynchronize(c->stream_copy));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->stream_part) HIP_TRY(c, hipStreamSynchronize(c->stream_part));
    HIP_TRY(c, hipStreamSynchronize(c->stream_ekf));
    if (c->stream_win) HIP_TRY(c, hipStreamSynchronize(c->stream_win));
    c->ekf_count = 0;
    return ASLAM_OK;
}

int install_dictionary(aslam_ctx* c, int ms, int n, int maxcorr, const uint8_t* bits) {
    if (ms < 3 || ms + 2 > kDictMaxCells || n < 1 || n > kIdTableSize || maxcorr < 0 || !bits)
        return fail(c, ASLAM_E_INVALID, "dictionary: marker size 3..7, 1..1024 markers (the id -> landmark table), maxCorrectionBits >= 0");
    std::vector<unsigned long long> codes;
    make_dict_from_bits(ms, n, bits, codes, c->dict_cells);
    int r = sync_streams(c);
    if (r) return r;
    unsigned long long* d_new = nullptr;
    HIP_TRY(c, dalloc(&d_new, codes.size()));
    HIP_TRY(c, hipMemcpy(d_new, codes.data(), codes.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
    hipFree(c->d_dict);
    c->d_dict = d_new;
    c->dict_ms = ms; c->dict_n = n; c->dict_maxcorr = maxcorr;
    if (c->rows > 0) return configure_frames(c, c->rows, c->cols, c->channels);      // marker size / error budgets follow
    return ASLAM_OK;
}

int sync_and_check(aslam_ctx* c) {
    int rs = sync_streams(c);
    if (rs) return rs;
    prof_collect(c);
    Counters h{};
    HIP_TRY(c, hipMemcpy(&h, c->d_ctr, sizeof(h), hipMemcpyDeviceToHost));
    if (h.overflow) {
        unsigned zero = 0;
        hipMemcpy(&c->d_ctr->overflow, &zero, sizeof(unsigned), hipMemcpyHostToDevice);
        char buf[256];
        snprintf(buf, sizeof(buf), "device list overflow (mask 0x%x: 1 starts, 2 contours, 4 points, 8 candidates, 16 markers, 32 landmarks, 64 fused updates > max_updates_per_frame)", h.overflow);
        return fail(c, ASLAM_E_CAPACITY, buf);
    }
    return ASLAM_OK;
}

} // namespace

extern "C" {

void aslam_default_init(aslam_init* i) {
    std::memset(i, 0, sizeof(*i));
    i->Q_k = 0.01; i->R_x = 100; i->R_y = 100; i->R_theta = 10;          // parameters.yaml:5-8
    i->kl = 0.05; i->kr = 0.05; i->b = 0.09;                              // parameters.yaml:11-13
    i->marker_length = 0.27; i->markers_dictionary = 16;                  // parameters.yaml:16-17
    i->useful_distance_threshold = 3.0f;                                  // aruco_slam.h:58
    i->r2c_q[3] = 1.0;
    i->device_id = 0;
    i->max_landmarks = 256;
    i->max_rows = 720; i->max_cols = 1280;
    i->max_batch = 1;
    i->persistent_waves = 0;
    i->ekf_reserved_cus_per_xcd = 0;
    i->max_updates_per_frame = 24;
    i->cap_starts_per_frame = 0; i->cap_contours_per_frame = 0; i->cap_points_per_frame = 0;
}

int aslam_create(const aslam_init* init, aslam_ctx** out) {
    if (!init || !out) return ASLAM_E_INVALID;
    *out = nullptr;
    if (init->markers_dictionary != 16) return ASLAM_E_INVALID;   // only DICT_ARUCO_ORIGINAL can be generated offline
    if (init->max_rows <= 0 || init->max_cols <= 0 || init->max_batch <= 0 || init->max_landmarks <= 0) return ASLAM_E_INVALID;
    if (init->max_updates_per_frame <= 0 || init->max_updates_per_frame > kMarkerMax) return ASLAM_E_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || init->device_id >= ndev) return ASLAM_E_NO_DEVICE;
    if (hipSetDevice(init->device_id) != hipSuccess) return ASLAM_E_NO_DEVICE;
    aslam_ctx* c = new aslam_ctx();
    c->init = *init;
    // default capacity of the start-candidate list: textured / noisy frames produce a candidate every few pixels
    if (c->init.cap_starts_per_frame == 0)
        c->init.cap_starts_per_frame = std::max(1u << 16, (unsigned)((size_t)init->max_rows * init->max_cols / 2));
    if (c->init.cap_contours_per_frame == 0) c->init.cap_contours_per_frame = 1u << 12;
    if (c->init.cap_points_per_frame == 0) c->init.cap_points_per_frame = 1u << 19;
    c->max_batch = init->max_batch;
    c->nwaves = init->persistent_waves > 0 ? init->persistent_waves : 4096;
    if (init->persistent_waves <= 0) {                       // tuning knob for callers that cannot reach aslam_init (the class adapter)
        const char* e = std::getenv("ASLAM_PERSISTENT_WAVES");
        if (e && std::atoi(e) > 0) c->nwaves = std::atoi(e);
    }
    c->sp.Q_k = init->Q_k; c->sp.R_x = init->R_x; c->sp.R_y = init->R_y; c->sp.R_theta = init->R_theta;
    c->sp.kl = init->kl; c->sp.kr = init->kr; c->sp.b = init->b; c->sp.marker_length = init->marker_length;
    c->sp.r2c_tx = init->r2c_t[0]; c->sp.r2c_ty = init->r2c_t[1];
    c->sp.useful_distance_threshold = init->useful_distance_threshold;

    c->win_enabled = std::getenv("ASLAM_NO_WINDOWS") == nullptr;
    c->win_no_early = std::getenv("ASLAM_WIN_NO_EARLY") != nullptr;
    if (const char* e = std::getenv("ASLAM_WIN_PIECE")) c->win_piece = std::min(kWinChainFrames, std::max(1, std::atoi(e)));
    const int B = c->max_batch;
    const size_t px = (size_t)init->max_rows * init->max_cols;
    const size_t pitch = ((size_t)init->max_cols + 63) / 64 * 64;
    int prio_lo = 0, prio_hi = 0;
    hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);      // the latency-bound EKF chain outranks the batched detection
    bool ok = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_lo) == hipSuccess;
    ok = ok && hipStreamCreateWithPriority(&c->stream_ekf, hipStreamNonBlocking, prio_hi) == hipSuccess;
    {
        // Detection that runs beside an EKF chain is confined to part of every XCD (CU-masked stream), so that the chain's
        // small dependent kernels always find idle CUs.  Mask bit i is CU i / nxcd of XCD i % nxcd (MI355X: 256 CUs = 8 XCDs
        // x 32); the layout is only assumed on the device it was measured on (gfx950 with 256 CUs), elsewhere no mask is used.
        hipDeviceProp_t prop{};
        const bool known = hipGetDeviceProperties(&prop, init->device_id) == hipSuccess && prop.multiProcessorCount == 256 &&
                           std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
        const int nxcd = 8, per_xcd = 32;
        // windowed EKF: one chain workgroup, 16 scan workgroups and a tile pass per window instead of three kernels per frame
        const int res = init->ekf_reserved_cus_per_xcd == 0 ? (c->win_enabled ? 4 : 16) : init->ekf_reserved_cus_per_xcd;
        if (known && res > 0 && res < per_xcd) {
            uint32_t mask[8];
            for (int w = 0; w < 8; w++) { mask[w] = 0; for (int b = 0; b < 32; b++) if ((w * 32 + b) / nxcd >= res) mask[w] |= 1u << b; }
            if (hipExtStreamCreateWithCUMask(&c->stream_part, 8, mask) != hipSuccess) { c->stream_part = nullptr; (void)hipGetLastError(); }
        }
    }
    ok = ok && hipEventCreateWithFlags(&c->ev_detect, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->ev_ekf, hipEventDisableTiming) == hipSuccess;
    ok = ok && dalloc(&c->d_in, px * 3 * B) == hipSuccess;
    ok = ok && dalloc(&c->d_gray, px * B) == hipSuccess;
    ok = ok && dalloc(&c->d_nbr, (size_t)kScales * nbr_plane_bytes(init->max_rows, (int)pitch) * B) == hipSuccess;
    ok = ok && dalloc(&c->d_starts, (size_t)c->init.cap_starts_per_frame * B) == hipSuccess;
    ok = ok && dalloc(&c->d_nstarts, B) == hipSuccess;
    ok = ok && dalloc(&c->d_ctr, 1) == hipSuccess;
    ok = ok && dalloc(&c->d_contours, (size_t)c->init.cap_contours_per_frame * B) == hipSuccess;
    ok = ok && dalloc(&c->d_ncontours, B) == hipSuccess;
    ok = ok && dalloc(&c->d_points, (size_t)c->init.cap_points_per_frame * B) == hipSuccess;
    ok = ok && dalloc(&c->d_npoints, B) == hipSuccess;
    ok = ok && dalloc(&c->d_nodeplane, (size_t)kScales * nbr_plane_bytes(init->max_rows, (int)pitch) * B) == hipSuccess;
    ok = ok && dalloc(&c->d_nodes, (size_t)c->init.cap_starts_per_frame * B) == hipSuccess;
    ok = ok && dalloc(&c->d_wlist, ((size_t)c->init.cap_points_per_frame / kWriteChunk + c->init.cap_contours_per_frame) * B) == hipSuccess;
    ok = ok && dalloc(&c->d_nwrite, B) == hipSuccess;
    ok = ok && dalloc(&c->d_link_todo, B) == hipSuccess;
    ok = ok && dalloc(&c->d_pre_write, (size_t)max_frames_per_call() + 1) == hipSuccess;
    ok = ok && dalloc(&c->d_refine_mask, 15 * 15) == hipSuccess;
    ok = ok && dalloc(&c->d_pre_trace, (size_t)max_frames_per_call() + 1) == hipSuccess;
    ok = ok && dalloc(&c->d_pre_quads, (size_t)max_frames_per_call() + 1) == hipSuccess;
    ok = ok && dalloc(&c->d_cands, (size_t)kCandMax * B) == hipSuccess;
    ok = ok && dalloc(&c->d_ncand, B) == hipSuccess;
    ok = ok && dalloc(&c->d_finals, (size_t)kCandMax * B) == hipSuccess;
    ok = ok && dalloc(&c->d_nfinal, B) == hipSuccess;
    ok = ok && dalloc(&c->d_work, (size_t)kCandMax * B) == hipSuccess;
    ok = ok && dalloc(&c->d_markers, (size_t)kMarkerMax * B) == hipSuccess;
    ok = ok && dalloc(&c->d_nmarkers, B) == hipSuccess;
    ok = ok && dalloc(&c->d_obs, (size_t)kMarkerMax * B) == hipSuccess;
    ok = ok && dalloc(&c->d_enc, (size_t)3 * B) == hipSuccess;
    ok = ok && dalloc(&c->d_synth, 256) == hipSuccess;
    std::vector<unsigned long long> codes;
    aslam_default_detector_params(&c->dp);
    make_dict_aruco_original(codes, c->dict_cells);
    ok = ok && dalloc(&c->d_dict, codes.size()) == hipSuccess;
    ok = ok && hipMemcpy(c->d_dict, codes.data(), codes.size() * sizeof(unsigned long long), hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemset(c->d_ctr, 0, sizeof(Counters)) == hipSuccess;
    ok = ok && hipMemset(c->d_nstarts, 0, sizeof(unsigned) * B) == hipSuccess;
    ok = ok && hipMemset(c->d_ncontours, 0, sizeof(unsigned) * B) == hipSuccess;
    ok = ok && hipMemset(c->d_npoints, 0, sizeof(unsigned) * B) == hipSuccess;
    ok = ok && hipMemset(c->d_nmarkers, 0, sizeof(unsigned) * B) == hipSuccess;
    ok = ok && hipMemset(c->d_nfinal, 0, sizeof(unsigned) * B) == hipSuccess;
    ok = ok && hipMemset(c->d_ncand, 0, sizeof(unsigned) * B) == hipSuccess;
    ok = ok && hipMemset(c->d_enc, 0, sizeof(double) * 3 * B) == hipSuccess;
    ok = ok && ekf_alloc(c->ekf, init->max_landmarks, init->max_batch, init->max_updates_per_frame) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->ev_obs[0], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->ev_obs[1], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->ev_idx, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipStreamCreateWithPriority(&c->stream_win, hipStreamNonBlocking, prio_hi) == hipSuccess;
    for (int i = 0; i < 64; i++) ok = ok && hipEventCreateWithFlags(&c->ev_win[i], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipHostMalloc(reinterpret_cast<void**>(&c->h_obs), (size_t)B * kMarkerMax * sizeof(ObsRaw), hipHostMallocDefault) == hipSuccess;
    ok = ok && hipHostMalloc(reinterpret_cast<void**>(&c->h_nm), (size_t)B * sizeof(unsigned), hipHostMallocDefault) == hipSuccess;
    ok = ok && hipHostMalloc(reinterpret_cast<void**>(&c->h_win_frames), (size_t)B * sizeof(WinFrame), hipHostMallocDefault) == hipSuccess;
    if (!ok) { aslam_destroy(c); return ASLAM_E_NO_DEVICE; }
    *out = c;
    return ASLAM_OK;
}

void aslam_destroy(aslam_ctx* c) {
    if (!c) return;
    c->pend.active = false;
    aslam_comm_destroy(c);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->stream_part) hipStreamSynchronize(c->stream_part);
    if (c->stream_ekf) hipStreamSynchronize(c->stream_ekf);
    prof_collect(c);
    hipFree(c->d_in); hipFree(c->d_gray); hipFree(c->d_nbr); hipFree(c->d_starts); hipFree(c->d_ctr);
    hipFree(c->d_refine_mask);
    hipFree(c->d_nodeplane); hipFree(c->d_nodes); hipFree(c->d_wlist); hipFree(c->d_nwrite); hipFree(c->d_link_todo); hipFree(c->d_pre_write);
    hipFree(c->d_nstarts); hipFree(c->d_ncontours); hipFree(c->d_npoints); hipFree(c->d_pre_trace); hipFree(c->d_pre_quads);
    hipFree(c->d_contours); hipFree(c->d_points); hipFree(c->d_cands); hipFree(c->d_ncand); hipFree(c->d_finals);
    hipFree(c->d_nfinal); hipFree(c->d_work); hipFree(c->d_dict); hipFree(c->d_markers); hipFree(c->d_nmarkers);
    hipFree(c->d_obs); hipFree(c->d_enc); hipFree(c->d_synth);
    ekf_free(c->ekf);
    if (c->ev_detect) hipEventDestroy(c->ev_detect);
    if (c->ev_ekf) hipEventDestroy(c->ev_ekf);
    if (c->stream_copy) { hipStreamSynchronize(c->stream_copy); hipStreamDestroy(c->stream_copy); }
    for (int h = 0; h < 2; h++) { if (c->ev_up[h]) hipEventDestroy(c->ev_up[h]); if (c->ev_det[h]) hipEventDestroy(c->ev_det[h]); }
    for (int h = 0; h < 2; h++) if (c->ev_export[h]) hipEventDestroy(c->ev_export[h]);
    if (c->h_ring) hipHostFree(c->h_ring);
    if (c->h_obs) hipHostFree(c->h_obs);
    if (c->h_nm) hipHostFree(c->h_nm);
    if (c->h_win_frames) hipHostFree(c->h_win_frames);
    for (int h = 0; h < 2; h++) if (c->ev_obs[h]) hipEventDestroy(c->ev_obs[h]);
    if (c->ev_idx) hipEventDestroy(c->ev_idx);
    for (int i = 0; i < 64; i++) if (c->ev_win[i]) hipEventDestroy(c->ev_win[i]);
    if (c->stream_win) { hipStreamSynchronize(c->stream_win); hipStreamDestroy(c->stream_win); }
    if (c->stream) hipStreamDestroy(c->stream);
    if (c->stream_part) hipStreamDestroy(c->stream_part);
    if (c->stream_ekf) hipStreamDestroy(c->stream_ekf);
    delete c;
}

const char* aslam_last_error(const aslam_ctx* c) { return c ? c->err.c_str() : "null context"; }

int aslam_set_camera(aslam_ctx* c, const double K[9], const double* D, int nD) {
    if (!c || !K || nD < 0 || (nD > 0 && !D)) return fail(c, ASLAM_E_INVALID, "bad camera arguments");
    c->cam.fx = K[0]; c->cam.fy = K[4]; c->cam.cx = K[2]; c->cam.cy = K[5];
    c->cam.nD = std::min(nD, 5);
    for (int i = 0; i < 5; i++) c->cam.k[i] = i < c->cam.nD ? D[i] : 0.0;
    c->have_cam = true;
    return ASLAM_OK;
}

int aslam_stage_frames(aslam_ctx* c, int slot0, const uint8_t* frames, int nframes, int rows, int cols, int channels,
                       size_t step, size_t frame_stride) {
    if (!c || !frames) return fail(c, ASLAM_E_INVALID, "null argument");
    int r = check_slot_range(c, slot0, nframes);
    if (r) return r;
    if (rows != c->rows || cols != c->cols || channels != c->channels) {
        r = configure_frames(c, rows, cols, channels);
        if (r) return r;
    }
    if (step < (size_t)cols * channels) return fail(c, ASLAM_E_INVALID, "step smaller than a row");
    r = quiesce_slots(c, slot0, nframes);
    if (r) return r;
    c->in_frame_bytes = (size_t)rows * cols * channels;
    const bool tight = step == (size_t)cols * channels;        // (a plain copy for tight rows: the 2-D path goes row by row for pageable memory)
    if (tight && (nframes == 1 || frame_stride == c->in_frame_bytes)) {
        HIP_TRY(c, hipMemcpyAsync(c->d_in + (size_t)slot0 * c->in_frame_bytes, frames, c->in_frame_bytes * nframes, hipMemcpyHostToDevice, c->stream));
    } else {
        for (int f = 0; f < nframes; f++) {
            if (tight)
                HIP_TRY(c, hipMemcpyAsync(c->d_in + (size_t)(slot0 + f) * c->in_frame_bytes, frames + (size_t)f * frame_stride, c->in_frame_bytes,
                                          hipMemcpyHostToDevice, c->stream));
            else
                HIP_TRY(c, hipMemcpy2DAsync(c->d_in + (size_t)(slot0 + f) * c->in_frame_bytes, (size_t)cols * channels,
                                            frames + (size_t)f * frame_stride, step, (size_t)cols * channels, rows,
                                            hipMemcpyHostToDevice, c->stream));
        }
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));      // px is borrowed for the call only
    return ASLAM_OK;
}

int aslam_stage_encoders(aslam_ctx* c, int slot0, int n, const double* wl, const double* wr, const double* dt) {
    if (!c || !wl || !wr || !dt) return fail(c, ASLAM_E_INVALID, "null argument");
    int r = check_slot_range(c, slot0, n);
    if (r) return r;
    r = quiesce_slots(c, slot0, n);              // a pending / in-flight batch still reads the samples in these slots
    if (r) return r;
    std::vector<double> h((size_t)3 * n);
    for (int i = 0; i < n; i++) { h[3 * i] = wl[i]; h[3 * i + 1] = wr[i]; h[3 * i + 2] = dt[i]; }
    HIP_TRY(c, hipMemcpy(c->d_enc + (size_t)3 * slot0, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
    c->enc_host.resize((size_t)3 * c->max_batch);
    std::memcpy(&c->enc_host[(size_t)3 * slot0], h.data(), h.size() * sizeof(double));
    return ASLAM_OK;
}

void note_ekf_range(aslam_ctx* c, int first, int count) {
    if (c->ekf_count == 0) { c->ekf_lo = first; c->ekf_hi = first + count; }
    else { c->ekf_lo = std::min(c->ekf_lo, first); c->ekf_hi = std::max(c->ekf_hi, first + count); }
    c->ekf_first = first;
    c->ekf_count = count;
}

int read_mirror(aslam_ctx* c) {
    HIP_TRY(c, hipStreamSynchronize(c->stream_ekf));
    c->m_id2idx.resize(kIdTableSize);
    HIP_TRY(c, hipMemcpy(c->m_id2idx.data(), c->ekf.d_id2idx, sizeof(int) * kIdTableSize, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(&c->m_L, c->ekf.d_L, sizeof(int), hipMemcpyDeviceToHost));
    int nl = 0;
    HIP_TRY(c, hipMemcpy(&nl, c->ekf.d_nlast, sizeof(int), hipMemcpyDeviceToHost));
    nl = std::min(std::max(nl, 0), (int)kMarkerMax);
    std::vector<LastObs> h(std::max(nl, 1));
    if (nl) HIP_TRY(c, hipMemcpy(h.data(), c->ekf.d_last, sizeof(LastObs) * nl, hipMemcpyDeviceToHost));
    c->m_last.resize(nl);
    for (int i = 0; i < nl; i++) { c->m_last[i].id = h[i].id; for (int k = 0; k < 3; k++) c->m_last[i].z[k] = h[i].z[k]; }
    c->mirror_dirty = false;
    return ASLAM_OK;
}

// The EKF work of the pending batch: the host follows the reference's bookkeeping (checkLandmark, queue order, the
// "stationary" test, aruco_slam.cpp:92-95, 192-198, 423-435) on the observations read back from the device and cuts the batch
// into windows = runs of frames whose fused landmarks fit into one set S (ekf_window.hip); every other frame, and every frame
// the host cannot decide (a marker id it does not know: a new landmark; one id twice), takes the per-frame chain, planned on the
// device.  A window frame may fuse any subset of S and may drop "stationary" observations; S is the union over the window.
int finalize_pending(aslam_ctx* c) {
    if (!c->pend.active) return ASLAM_OK;
    const aslam_ctx::Pending p = c->pend;
    c->pend.active = false;
    HIP_TRY(c, hipEventSynchronize(c->ev_obs[p.ev]));
    if (c->mirror_dirty) { int r = read_mirror(c); if (r) return r; }
    if (c->ev_idx_set) HIP_TRY(c, hipEventSynchronize(c->ev_idx));          // the previous batch's upload has left the pinned rows
    HIP_TRY(c, hipStreamWaitEvent(c->stream_ekf, c->ev_obs[p.ev], 0));
    struct FramePlan { std::vector<int> corr_idx, corr_det, pop_idx, pop_det, pop_act; int n_markers = 0; };
    struct Op { int frame; bool predict; int K; std::vector<int> S; };      // K == 0: per-frame chain for `frame`; else a window of K frames on S
    std::vector<Op> ops;
    std::vector<FramePlan> plans(p.count);
    // largest set S: contexts configured for few corrections per frame keep SP <= 128 (their chain steps stay cheap)
    const int s_cap = c->ekf.win_sp_max >= 192 ? kWinSMax : 41;
    int w_first = -1, w_K = 0;
    std::vector<int> w_S;                                                   // sorted landmark indices of the open window
    bool device_plans = false;
    auto close_window = [&]() {
        if (w_K == 0) return;
        Op o{};
        o.frame = w_first; o.predict = true;
        if (w_K == 1) o.K = 0;                                              // a lone frame: the per-frame chain is as good
        else { o.K = w_K; o.S = w_S; }
        ops.push_back(o);
        w_K = 0; w_S.clear();
    };
    for (int f = p.first; f < p.first + p.count; f++) {
        const bool predict = c->is_init;            // addEncoder semantics (aruco_slam.cpp:24-29): the very first sample only arms the filter
        c->is_init = true;
        const ObsRaw* ob = c->h_obs + (size_t)f * kMarkerMax;
        const int nM = (int)std::min(c->h_nm[f], (unsigned)kMarkerMax);
        bool clean = !device_plans;
        std::vector<std::pair<int, int>> pop;       // (landmark index, detection index) of the observations that pass the gates, pop order
        int n_new = 0;
        if (clean) {
            // obs_.push(ob) in detection order into the reference's own priority queue (aruco_slam.h:85-88, aruco_slam.cpp:369-373):
            // new markers (index -1) pop first, in libstdc++ heap order, and take the next landmark indices in that order (:256)
            struct QItem { int idx, det; bool operator<(const QItem& o) const { return idx > o.idx; } };
            std::priority_queue<QItem> q;
            std::vector<int> seen;
            for (int i = 0; i < nM && clean; i++) {
                if (!ob[i].valid) continue;
                const int id = ob[i].id;
                if (id < 0 || id >= kIdTableSize || std::find(seen.begin(), seen.end(), id) != seen.end()) { clean = false; break; }   // one id twice (Q10): device
                seen.push_back(id);
                q.push(QItem{c->m_id2idx[id], i});
            }
            while (clean && !q.empty()) {
                QItem it = q.top();
                q.pop();
                if (it.idx < 0) {
                    if (c->m_L >= c->ekf.max_landmarks) { clean = false; break; }        // capacity: reported by the device
                    it.idx = c->m_L++;
                    c->m_id2idx[ob[it.det].id] = it.idx;
                    n_new++;
                }
                pop.push_back({it.idx, it.det});
            }
        }
        if (!clean) {
            if (std::getenv("ASLAM_DEBUG_PLAN")) std::fprintf(stderr, "plan frame %d: nM %d left to the device\n", f, nM);
            close_window();
            device_plans = true;
            c->plan_stats[3]++;
            c->mirror_dirty = true;
            Op o{};
            o.frame = f; o.predict = predict; o.K = 0;
            ops.push_back(o);
            continue;
        }
        // pop order = ascending landmark index (aruco_slam.h:85-88); "stationary" test against last frame's list (:192-198)
        FramePlan& fp = plans[f - p.first];
        fp.n_markers = nM;
        int n_stationary = 0;
        std::vector<aslam_ctx::HostLast> nlast(pop.size());
        for (size_t j = 0; j < pop.size(); j++) {
            const ObsRaw& o = ob[pop[j].second];
            bool stationary = false;
            if ((int)j < n_new) {                                            // augment branch: last_observation_ stays unset (Q3)
                nlast[j].id = o.id;
                nlast[j].z[0] = nlast[j].z[1] = nlast[j].z[2] = std::nan("");
                continue;
            }
            for (const aslam_ctx::HostLast& l : c->m_last)
                if (l.id == o.id) {                                          // std::find: first with the same id
                    const double d0 = l.z[0] - o.x, d1 = l.z[1] - o.y, d2 = l.z[2] - o.th;
                    stationary = std::sqrt(d0 * d0 + d1 * d1 + d2 * d2) < 0.01;     // NaN compares false
                    break;
                }
            n_stationary += stationary ? 1 : 0;
            nlast[j].id = o.id;
            if (stationary) { nlast[j].z[0] = nlast[j].z[1] = nlast[j].z[2] = std::nan(""); }
            else { nlast[j].z[0] = o.x; nlast[j].z[1] = o.y; nlast[j].z[2] = o.th; }
            fp.pop_idx.push_back(pop[j].first); fp.pop_det.push_back(pop[j].second); fp.pop_act.push_back(stationary ? 2 : 1);
            if (!stationary) { fp.corr_idx.push_back(pop[j].first); fp.corr_det.push_back(pop[j].second); }
        }
        c->m_last.swap(nlast);
        const int m = (int)fp.corr_idx.size();
        const bool eligible = predict && n_new == 0 && m <= kWinCorrMax && (int)pop.size() <= 64 && m <= s_cap && m <= c->init.max_updates_per_frame;
        if (std::getenv("ASLAM_DEBUG_PLAN"))
            std::fprintf(stderr, "plan frame %d: nM %d m %d new %d stationary %d predict %d eligible %d (window K %d, |S| %zu)\n", f, nM, m, n_new, n_stationary,
                         (int)predict, (int)eligible, w_K, w_S.size());
        if (!eligible) {
            close_window();
            Op o{};
            o.frame = f; o.predict = predict; o.K = 0;
            ops.push_back(o);
            continue;
        }
        // does the frame fit into the open window?  (its landmarks are ascending, like S)
        std::vector<int> un;
        if (w_K > 0) std::set_union(w_S.begin(), w_S.end(), fp.corr_idx.begin(), fp.corr_idx.end(), std::back_inserter(un));
        // a wider image makes every step of the window dearer: a window that is already long is closed rather than widened
        const bool widens = w_K >= kWinWidenFrames && ekf_win_tiles((int)un.size()) > ekf_win_tiles((int)w_S.size());
        if (w_K == 0 || w_K >= kWinFrames || (int)un.size() > s_cap || f != w_first + w_K || widens) {
            close_window();
            w_first = f; w_K = 0;
            un = fp.corr_idx;
        }
        w_S.swap(un);
        w_K++;
    }
    close_window();
    // the plan of every window frame in the device's format (positions within the final S of its window), one upload per batch
    bool any_window = false;
    for (const Op& o : ops) {
        if (o.K == 0) continue;
        any_window = true;
        for (int k = 0; k < o.K; k++) {
            const FramePlan& fp = plans[o.frame + k - p.first];
            WinFrame& wf = c->h_win_frames[o.frame + k];
            wf.m = (int)fp.corr_idx.size(); wf.npop = (int)fp.pop_idx.size(); wf.n_markers = fp.n_markers; wf.pad = 0;
            for (int a = 0; a < wf.m; a++) {
                wf.cdet[a] = (unsigned char)fp.corr_det[a];
                wf.cpos[a] = (unsigned char)(std::lower_bound(o.S.begin(), o.S.end(), fp.corr_idx[a]) - o.S.begin());
            }
            for (int i = 0; i < wf.npop; i++) {
                wf.pdet[i] = (unsigned char)fp.pop_det[i]; wf.pact[i] = (unsigned char)fp.pop_act[i]; wf.pidx[i] = (short)fp.pop_idx[i];
            }
        }
    }
    if (any_window) {
        HIP_TRY(c, hipMemcpyAsync(c->ekf.d_win_frames + p.first, c->h_win_frames + p.first, (size_t)p.count * sizeof(WinFrame),
                                  hipMemcpyHostToDevice, c->stream_ekf));
        HIP_TRY(c, hipEventRecord(c->ev_idx, c->stream_ekf));
        c->ev_idx_set = true;
    }
    for (const Op& o : ops) {
        if (o.K == 0) c->plan_stats[1]++;
        else { c->plan_stats[0] += o.K; c->plan_stats[2]++; }
    }
    // Streams: sa = the EKF stream (chain pieces with their replay, per-frame chains), sb = the window's gather and its flush.  A window that is followed by another window does not wait for its own flush: the next window's P and mu_S are
    // derived from this window's small results (launch_ekf_win_next, on sa) while the pass over Sigma runs on sb; only the next
    // window's LAST piece (which writes mu back) and whatever is not a window wait for the flush.
    hipStream_t sa = c->stream_ekf, sb = c->stream_win;
    auto new_event = [&]() { return c->ev_win[c->ev_win_next++ & 63]; };
    struct { bool active = false; WinDesc wd{}; } pf;               // the previous window's flush, not yet enqueued
    auto flush_now = [&](const WinDesc& fwd) -> int {                                    // classic order: flush, then the EKF stream goes on
        hipEvent_t evr = new_event();
        HIP_TRY(c, hipEventRecord(evr, sa));                         // the window's replay (Lambda, Psi, psi) is complete
        HIP_TRY(c, hipStreamWaitEvent(sb, evr, 0));
        prof_begin(c, P_EKF_WIN_FLUSH, sb);
        launch_ekf_win_flush(sb, c->ekf, fwd);
        prof_end(c);
        hipEvent_t ev = new_event();
        HIP_TRY(c, hipEventRecord(ev, sb));
        HIP_TRY(c, hipStreamWaitEvent(sa, ev, 0));                   // whatever follows on the EKF stream sees the flushed Sigma
        return ASLAM_OK;
    };
    for (size_t oi = 0; oi < ops.size(); oi++) {
        const Op& o = ops[oi];
        if (o.K == 0) {
            if (pf.active) { int r = flush_now(pf.wd); if (r) return r; pf.active = false; }
            const double* e = &c->enc_host[(size_t)3 * o.frame];
            int r = run_ekf_frame(c, o.frame, e[0], e[1], e[2], o.predict);
            if (r) return r;
            continue;
        }
        // One window = K frames on the set S.  Its chain is cut into pieces of a few frames on sa; the replay of each piece's
        // log goes to sb, so that only the last piece's replay is not hidden behind the chain.
        WinDesc wd{};
        wd.nS = (int)o.S.size();
        wd.T = ekf_win_tiles(wd.nS);
        wd.wpar = (int)(c->win_count++ & 1u);
        for (int a = 0; a < wd.nS; a++) wd.li[a] = (short)(3 + 3 * o.S[a]);
        hipEvent_t ev_prev_flush = nullptr;
        if (pf.active) {
            prof_begin(c, P_EKF_WIN_NEXT, sa);                       // (the previous window's Lambda, Psi, psi are complete: same stream)
            launch_ekf_win_next(sa, c->ekf, pf.wd, wd);
            prof_end(c);
            hipEvent_t ev_next = new_event();
            HIP_TRY(c, hipEventRecord(ev_next, sa));
            HIP_TRY(c, hipStreamWaitEvent(sb, ev_next, 0));          // the flush rewrites what launch_ekf_win_next reads (Sigma, mu, Y_0)
            prof_begin(c, P_EKF_WIN_FLUSH, sb);
            launch_ekf_win_flush(sb, c->ekf, pf.wd);
            prof_end(c);
            ev_prev_flush = new_event();
            HIP_TRY(c, hipEventRecord(ev_prev_flush, sb));
            pf.active = false;
            wd.from_image = 1;
        } else {
            hipEvent_t ev = new_event();
            HIP_TRY(c, hipEventRecord(ev, sa));
            HIP_TRY(c, hipStreamWaitEvent(sb, ev, 0));               // Sigma as the EKF stream leaves it
        }
        launch_ekf_win_gather(sb, c->ekf, wd);                       // Y_0 of this window (behind the previous window's flush: same stream)
        // The pieces of the window, back to back on sa: launch i carries the chain of piece i, the replay of piece i - 1 and the Psi
        // product of piece i - 2 (ekf_window.hip: k_ekf_win_step), so nothing but the stream orders them; two more launches drain
        // the replay.  What follows the window waits for that drain: the pieces shrink towards the end (.., 4, 2, 1, 1 frames).
        struct Piece { WinDesc sub; int nsteps; };
        std::vector<Piece> pieces;
        {
            // piece sizes: the replay of piece i - 1 runs in the launch of piece i and takes about half as long per frame as the
            // chain, so a piece may be at most twice as long as the one that follows it; from the window's end: 1, 1, 2, 4, 8, 8, ..
            std::vector<int> sizes;
            {
                int left = o.K, nxt = 1, count = 0;
                while (left > 0) {
                    int kn = std::min(std::min(nxt, c->win_piece), left);
                    sizes.push_back(kn);
                    left -= kn;
                    if (++count >= 2) nxt = std::min(2 * nxt, c->win_piece);       // 1, 1, 2, 4, ...
                }
                std::reverse(sizes.begin(), sizes.end());
            }
            int piece = 0, log0 = 0, k0 = 0;
            for (int kn : sizes) {
                Piece pc;
                pc.sub = wd;
                pc.sub.first_slot = o.frame + k0;
                pc.sub.K = kn;
                pc.sub.piece = piece++;
                pc.sub.log0 = log0;
                pc.sub.last = k0 + kn == o.K ? 1 : 0;
                pc.nsteps = 0;
                for (int k = 0; k < kn; k++) pc.nsteps += 1 + c->h_win_frames[pc.sub.first_slot + k].m;
                log0 += pc.nsteps;
                k0 += kn;
                pieces.push_back(pc);
            }
        }
        const int P = (int)pieces.size();
        for (int i = 0; i < P + 2; i++) {
            WinDesc cw = wd;
            cw.K = 0;
            if (i < P) cw = pieces[i].sub;
            const Piece* ps = (i >= 1 && i <= P) ? &pieces[i - 1] : nullptr;
            const Piece* pq = (i >= 2) ? &pieces[i - 2] : nullptr;
            if (i < P && cw.last && ev_prev_flush) HIP_TRY(c, hipStreamWaitEvent(sa, ev_prev_flush, 0));   // mu_S goes back into the state: after the previous flush's mu_R pass
            prof_begin(c, i < P ? P_EKF_WIN_CHAIN : P_EKF_WIN_SCAN, sa);
            launch_ekf_win_step(sa, c->ekf, c->sp, cw, c->d_obs, c->d_enc, ps ? ps->sub.piece : 0, ps ? ps->sub.log0 : 0, ps ? ps->nsteps : 0,
                                pq ? pq->sub.piece : 0, pq ? pq->sub.log0 : 0, pq ? pq->nsteps : 0);
            prof_end(c);
        }
        HIP_TRY(c, hipGetLastError());
        const bool next_is_window = oi + 1 < ops.size() && ops[oi + 1].K > 0 && !c->win_no_early;
        if (next_is_window) {
            pf.active = true; pf.wd = wd;
        } else {
            int r = flush_now(wd);
            if (r) return r;
        }
    }
    if (pf.active) { int r = flush_now(pf.wd); if (r) return r; }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev_ekf, c->stream_ekf));
    note_ekf_range(c, p.first, p.count);
    return ASLAM_OK;
}

int run_staged(aslam_ctx* c, int first, int count, int with_ekf, hipEvent_t wait_before) {
    int r = check_slot_range(c, first, count);
    if (r) return r;
    if (with_ekf && c->enc_host.size() < (size_t)3 * (first + count)) return fail(c, ASLAM_E_STATE, "encoders not staged");
    if (c->pend.active && first < c->pend.first + c->pend.count && c->pend.first < first + count) {
        r = finalize_pending(c);               // the pending batch still needs the observations in these slots
        if (r) return r;
    }
    if (with_ekf != 2) {                       // 2 = EKF only, on observations already present in the slots (tests)
        r = run_detect(c, first, count, with_ekf == 1, wait_before);
        if (r) return r;
    } else {
        if (c->last_detect && c->last_detect != c->stream) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_detect, 0));
        c->last_detect = c->stream;
        HIP_TRY(c, hipEventRecord(c->ev_detect, c->stream));
    }
    if (!with_ekf) return ASLAM_OK;
    if (!c->win_enabled) {                     // every frame on the per-frame chain, enqueued at once
        r = finalize_pending(c);
        if (r) return r;
        HIP_TRY(c, hipStreamWaitEvent(c->stream_ekf, c->ev_detect, 0));
        for (int i = 0; i < count; i++) {
            const double* e = &c->enc_host[(size_t)3 * (first + i)];
            bool predict = c->is_init;         // addEncoder semantics (aruco_slam.cpp:24-29): the very first sample only arms the filter
            c->is_init = true;
            r = run_ekf_frame(c, first + i, e[0], e[1], e[2], predict);
            if (r) return r;
            c->plan_stats[1]++;
        }
        HIP_TRY(c, hipEventRecord(c->ev_ekf, c->stream_ekf));
        note_ekf_range(c, first, count);
        return ASLAM_OK;
    }
    // observations of the batch to the host, behind its detection; the batch's EKF work is enqueued by finalize_pending
    hipStream_t st = c->last_detect;
    const int e = c->ev_obs_next;
    c->ev_obs_next ^= 1;
    HIP_TRY(c, hipMemcpyAsync(c->h_obs + (size_t)first * kMarkerMax, c->d_obs + (size_t)first * kMarkerMax, (size_t)count * kMarkerMax * sizeof(ObsRaw),
                              hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(c->h_nm + first, c->d_nmarkers + first, (size_t)count * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipEventRecord(c->ev_obs[e], st));
    r = finalize_pending(c);                   // the PREVIOUS batch: its detection has long finished, this one's is already queued
    if (r) return r;
    c->pend.active = true; c->pend.first = first; c->pend.count = count; c->pend.ev = e;
    return ASLAM_OK;
}

int aslam_run_staged(aslam_ctx* c, int first, int count, int with_ekf) {
    if (!c) return ASLAM_E_INVALID;
    return run_staged(c, first, count, with_ekf, nullptr);
}

// ---- host-fed stream: pinned ring, asynchronous upload (include/aruco_slam_hip.h) --------------------------------------
int ring_submit(aslam_ctx* c) {
    const int n = c->ring_fill, h = c->ring_half, H = c->ring_H;
    if (n == 0) return ASLAM_OK;
    const size_t fb = c->in_frame_bytes;
    const int slot0 = h * H;
    if (c->ev_det_set[h]) HIP_TRY(c, hipStreamWaitEvent(c->stream_copy, c->ev_det[h], 0));     // the detector still reads these slots
    if (c->pend.active && slot0 < c->pend.first + c->pend.count && c->pend.first < slot0 + n) { int rp = finalize_pending(c); if (rp) return rp; }
    if (c->ekf_count > 0 && slot0 < c->ekf_hi && c->ekf_lo < slot0 + n)
        HIP_TRY(c, hipStreamWaitEvent(c->stream_copy, c->ev_ekf, 0));                          // EKF work in flight still reads these slots' encoder samples
    // the encoder samples go to the device as well: the window chain reads them there (ekf_window.hip)
    HIP_TRY(c, hipMemcpyAsync(c->d_enc + (size_t)3 * slot0, c->ring_enc.data(), (size_t)3 * n * sizeof(double), hipMemcpyHostToDevice, c->stream_copy));
    HIP_TRY(c, hipMemcpyAsync(c->d_in + (size_t)slot0 * fb, c->h_ring + (size_t)slot0 * fb, (size_t)n * fb, hipMemcpyHostToDevice, c->stream_copy));
    HIP_TRY(c, hipEventRecord(c->ev_up[h], c->stream_copy));
    c->ev_up_set[h] = true;
    c->enc_host.resize((size_t)3 * c->max_batch);
    std::memcpy(&c->enc_host[(size_t)3 * slot0], c->ring_enc.data(), (size_t)3 * n * sizeof(double));
    int r = run_staged(c, slot0, n, 1, c->ev_up[h]);
    if (r) return r;
    HIP_TRY(c, hipEventRecord(c->ev_det[h], c->last_detect));
    c->ev_det_set[h] = true;
    c->ring_half ^= 1;
    c->ring_fill = 0;
    // the other half is filled next: its previous upload must have left the pinned memory
    if (c->ev_up_set[c->ring_half]) HIP_TRY(c, hipEventSynchronize(c->ev_up[c->ring_half]));
    return ASLAM_OK;
}

int aslam_stream_open(aslam_ctx* c, int rows, int cols, int channels, int frames_per_submit) {
    if (!c) return ASLAM_E_INVALID;
    if (frames_per_submit < 1 || 2 * frames_per_submit > c->max_batch) return fail(c, ASLAM_E_INVALID, "frames_per_submit must be in [1, max_batch / 2]");
    int r = sync_streams(c);
    if (r) return r;
    r = configure_frames(c, rows, cols, channels);
    if (r) return r;
    c->in_frame_bytes = (size_t)rows * cols * channels;
    if (c->h_ring) { hipHostFree(c->h_ring); c->h_ring = nullptr; }
    HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_ring), (size_t)2 * frames_per_submit * c->in_frame_bytes, hipHostMallocDefault));
    if (!c->stream_copy) {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->stream_copy, hipStreamNonBlocking));
        for (int h = 0; h < 2; h++) {
            HIP_TRY(c, hipEventCreateWithFlags(&c->ev_up[h], hipEventDisableTiming));
            HIP_TRY(c, hipEventCreateWithFlags(&c->ev_det[h], hipEventDisableTiming));
        }
    }
    c->ev_det_set[0] = c->ev_det_set[1] = c->ev_up_set[0] = c->ev_up_set[1] = false;
    c->ring_H = frames_per_submit; c->ring_half = 0; c->ring_fill = 0; c->ring_acquired = false;
    c->ring_enc.assign((size_t)3 * frames_per_submit, 0.0);
    return ASLAM_OK;
}

int aslam_stream_acquire(aslam_ctx* c, uint8_t** px, size_t* step) {
    if (!c || !px) return ASLAM_E_INVALID;
    if (!c->h_ring) return fail(c, ASLAM_E_STATE, "aslam_stream_open first");
    if (!c->have_cam) return fail(c, ASLAM_E_STATE, "camera parameters not set (aslam_set_camera)");
    *px = c->h_ring + ((size_t)c->ring_half * c->ring_H + c->ring_fill) * c->in_frame_bytes;
    if (step) *step = (size_t)c->cols * c->channels;
    c->ring_acquired = true;
    return ASLAM_OK;
}

int aslam_stream_commit(aslam_ctx* c, double wl, double wr, double dt) {
    if (!c) return ASLAM_E_INVALID;
    if (!c->ring_acquired) return fail(c, ASLAM_E_STATE, "aslam_stream_acquire first");
    c->ring_acquired = false;
    double* e = &c->ring_enc[(size_t)3 * c->ring_fill];
    e[0] = wl; e[1] = wr; e[2] = dt;
    if (++c->ring_fill == c->ring_H) return ring_submit(c);
    return ASLAM_OK;
}

int aslam_stream_push(aslam_ctx* c, const uint8_t* px, size_t step, double wl, double wr, double dt) {
    if (!c || !px) return ASLAM_E_INVALID;
    uint8_t* dst = nullptr;
    size_t dstep = 0;
    int r = aslam_stream_acquire(c, &dst, &dstep);
    if (r) return r;
    if (step < dstep) { c->ring_acquired = false; return fail(c, ASLAM_E_INVALID, "step smaller than a row"); }
    if (step == dstep) std::memcpy(dst, px, (size_t)c->rows * dstep);
    else for (int y = 0; y < c->rows; y++) std::memcpy(dst + (size_t)y * dstep, px + (size_t)y * step, dstep);
    return aslam_stream_commit(c, wl, wr, dt);
}

int aslam_stream_flush(aslam_ctx* c) {
    if (!c) return ASLAM_E_INVALID;
    if (!c->h_ring) return fail(c, ASLAM_E_STATE, "aslam_stream_open first");
    int r = ring_submit(c);
    if (r) return r;
    return sync_and_check(c);
}

int aslam_sync(aslam_ctx* c) {
    if (!c) return ASLAM_E_INVALID;
    return sync_and_check(c);
}

void aslam_default_detector_params(aslam_detector_params* p) {
    if (!p) return;
    p->adaptiveThreshWinSizeMin = 3; p->adaptiveThreshWinSizeMax = 23; p->adaptiveThreshWinSizeStep = 10;
    p->adaptiveThreshConstant = 7;
    p->minMarkerPerimeterRate = 0.03; p->maxMarkerPerimeterRate = 4.0;
    p->polygonalApproxAccuracyRate = 0.05;            // 3.2.0 (0.03 from 3.3)
    p->minCornerDistanceRate = 0.05;
    p->minDistanceToBorder = 3;
    p->minMarkerDistanceRate = 0.05;
    p->doCornerRefinement = 0; p->cornerRefinementWinSize = 5; p->cornerRefinementMaxIterations = 30; p->cornerRefinementMinAccuracy = 0.1;
    p->markerBorderBits = 1;
    p->perspectiveRemovePixelPerCell = 8;             // 3.2.0 (4 from 3.3)
    p->perspectiveRemoveIgnoredMarginPerCell = 0.13;
    p->maxErroneousBitsInBorderRate = 0.35;
    p->minOtsuStdDev = 5.0;
    p->errorCorrectionRate = 0.6;
}

int aslam_set_detector_params(aslam_ctx* c, const aslam_detector_params* p) {
    if (!c || !p) return ASLAM_E_INVALID;
    {
        // the LDS tile carries a 12-pixel halo and three mask planes: up to 3 windows of at most 23 pixels
        const int step = p->adaptiveThreshWinSizeStep;
        const int ns = step > 0 && p->adaptiveThreshWinSizeMax >= p->adaptiveThreshWinSizeMin
                           ? (p->adaptiveThreshWinSizeMax - p->adaptiveThreshWinSizeMin) / step + 1 : 0;
        int last = p->adaptiveThreshWinSizeMin + (ns - 1) * step;
        if (last % 2 == 0) last++;
        if (ns < 1 || ns > kScales || p->adaptiveThreshWinSizeMin < 3 || last > 23)
            return fail(c, ASLAM_E_INVALID, "adaptive threshold windows: 1..3 sizes (min + i step) between 3 and 23 pixels");
    }
    if (p->perspectiveRemovePixelPerCell < 2 || p->perspectiveRemovePixelPerCell > kCellPx ||
        p->perspectiveRemovePixelPerCell - 2 * (int)(p->perspectiveRemoveIgnoredMarginPerCell * p->perspectiveRemovePixelPerCell) < 1)
        return fail(c, ASLAM_E_INVALID, "perspectiveRemovePixelPerCell: 2..8, with at least one pixel left inside the margin");
    if (p->markerBorderBits != 1) return fail(c, ASLAM_E_INVALID, "markerBorderBits is compiled in as 1");
    if (p->doCornerRefinement && (p->cornerRefinementWinSize < 1 || p->cornerRefinementWinSize > 7 || p->cornerRefinementMaxIterations < 1 ||
                                  !(p->cornerRefinementMinAccuracy > 0)))
        return fail(c, ASLAM_E_INVALID, "corner refinement: window 1..7, at least one iteration, positive accuracy");
    if (!(p->maxMarkerPerimeterRate > 0) || !(p->minMarkerPerimeterRate > 0) || p->minMarkerPerimeterRate > p->maxMarkerPerimeterRate ||
        p->maxMarkerPerimeterRate * std::max(c->init.max_rows, c->init.max_cols) > 65534.0)
        return fail(c, ASLAM_E_INVALID, "0 < minMarkerPerimeterRate <= maxMarkerPerimeterRate, and the longest kept border (rate x larger frame side) <= 65534 points");
    if (!(p->polygonalApproxAccuracyRate > 0) || p->minCornerDistanceRate < 0 || p->minMarkerDistanceRate < 0 || p->minDistanceToBorder < 0 ||
        p->adaptiveThreshConstant < 0 || p->adaptiveThreshConstant > 255 || p->perspectiveRemoveIgnoredMarginPerCell < 0 ||
        p->perspectiveRemoveIgnoredMarginPerCell >= 0.5 || p->maxErroneousBitsInBorderRate < 0 || p->errorCorrectionRate < 0 || p->minOtsuStdDev < 0)
        return fail(c, ASLAM_E_INVALID, "detector parameter out of range");
    int r = sync_streams(c);
    if (r) return r;
    c->dp = *p;
    if (p->doCornerRefinement) {
        // cornerSubPix's separable window (cornersubpix.cpp): float arithmetic on the host, the same libm as the CPU restatement
        const int win = p->cornerRefinementWinSize, w = 2 * win + 1;
        std::vector<float> mask((size_t)w * w);
        for (int i = 0; i < w; i++) {
            float y = (float)(i - win) / win;
            float vy = std::exp(-y * y);
            for (int j = 0; j < w; j++) {
                float x = (float)(j - win) / win;
                mask[(size_t)i * w + j] = (float)(vy * std::exp(-x * x));
            }
        }
        HIP_TRY(c, hipMemcpy(c->d_refine_mask, mask.data(), mask.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    if (c->rows > 0) return configure_frames(c, c->rows, c->cols, c->channels);
    return ASLAM_OK;
}

int aslam_set_dictionary(aslam_ctx* c, int marker_size, int n_markers, int max_correction_bits, const uint8_t* bits) {
    if (!c) return ASLAM_E_INVALID;
    return install_dictionary(c, marker_size, n_markers, max_correction_bits, bits);
}

int aslam_set_dictionary_bytes(aslam_ctx* c, int marker_size, int n_markers, int max_correction_bits, const uint8_t* bytes_list) {
    if (!c) return ASLAM_E_INVALID;
    if (marker_size < 3 || marker_size + 2 > kDictMaxCells || n_markers < 1 || !bytes_list)
        return fail(c, ASLAM_E_INVALID, "dictionary: marker size 3..7, at least one marker");
    // cv::aruco::Dictionary::bytesList: n rows x nbytes columns x 4 channels (rotations); channel 0 = the unrotated marker,
    // bits shifted in first-to-last, so a trailing partial byte holds its bits right-aligned (getByteListFromBits)
    const int nb = marker_size * marker_size, nbytes = (nb + 7) / 8, rem = nb % 8;
    std::vector<uint8_t> bits((size_t)n_markers * nb);
    for (int id = 0; id < n_markers; id++)
        for (int j = 0; j < nb; j++) {
            const int b = j / 8, p = j % 8;
            const unsigned v = bytes_list[((size_t)id * nbytes + b) * 4];
            const int width = (b == nbytes - 1 && rem) ? rem : 8;
            bits[(size_t)id * nb + j] = (uint8_t)((v >> (width - 1 - p)) & 1u);
        }
    return install_dictionary(c, marker_size, n_markers, max_correction_bits, bits.data());
}

int aslam_add_encoder(aslam_ctx* c, double wl, double wr, double t_now) {
    if (!c) return ASLAM_E_INVALID;
    { int rp = finalize_pending(c); if (rp) return rp; }     // a pending batch arms the filter itself (is_init is set there)
    if (!c->is_init) {                       // aruco_slam.cpp:24-29
        c->last_time = t_now;
        c->is_init = true;
        return ASLAM_OK;
    }
    double dt = t_now - c->last_time;        // aruco_slam.cpp:31-32
    c->last_time = t_now;
    launch_ekf_predict_only(c->stream_ekf, c->ekf, c->sp, wl, wr, dt);
    HIP_TRY(c, hipGetLastError());
    return ASLAM_OK;
}

int aslam_add_image(aslam_ctx* c, const uint8_t* px, int rows, int cols, int channels, size_t step) {
    if (!c || !px) return fail(c, ASLAM_E_INVALID, "null argument");
    int r = finalize_pending(c);
    if (r) return r;
    if (!c->is_init) return ASLAM_OK;        // aruco_slam.cpp:84-85: nothing happens before the first encoder message
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    c->mirror_dirty = true;                  // planned on the device: the host's copy of the tables is stale afterwards
    r = aslam_stage_frames(c, 0, px, 1, rows, cols, channels, step, 0);
    if (r) return r;
    const auto t1 = clk::now();
    r = run_detect(c, 0, 1);
    if (r) return r;
    const auto t2 = clk::now();
    HIP_TRY(c, hipStreamWaitEvent(c->stream_ekf, c->ev_detect, 0));
    r = run_ekf_frame(c, 0, 0, 0, 0, false);
    if (r) return r;
    const auto t3 = clk::now();
    r = sync_streams(c);
    const auto t4 = clk::now();
    if (!r) r = sync_and_check(c);
    const auto t5 = clk::now();
    auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    c->last_timing[0] = us(t0, t1); c->last_timing[1] = us(t1, t2); c->last_timing[2] = us(t2, t3); c->last_timing[3] = us(t3, t4);
    c->last_timing[4] = us(t4, t5); c->last_timing[5] = us(t0, t5);
    return r;
}

int aslam_get_last_timing(aslam_ctx* c, double out[6]) {
    if (!c || !out) return ASLAM_E_INVALID;
    for (int i = 0; i < 6; i++) out[i] = c->last_timing[i];
    return ASLAM_OK;
}

int aslam_get_state(aslam_ctx* c, int* N, double* mu, double* sigma) {
    if (!c || !N) return fail(c, ASLAM_E_INVALID, "null argument");
    { int rs = sync_streams(c); if (rs) return rs; }
    int L = 0;
    HIP_TRY(c, hipMemcpy(&L, c->ekf.d_L, sizeof(int), hipMemcpyDeviceToHost));
    const int n = 3 + 3 * L;
    *N = n;
    if (mu) HIP_TRY(c, hipMemcpy(mu, c->ekf.d_mu, sizeof(double) * n, hipMemcpyDeviceToHost));
    if (sigma) {
        // device layout: column-major with leading dimension ld = N_max; pack to ld = N
        std::vector<double> tmp((size_t)c->ekf.ld * n);
        HIP_TRY(c, hipMemcpy(tmp.data(), c->ekf.d_sigma, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int col = 0; col < n; col++) std::memcpy(sigma + (size_t)col * n, &tmp[(size_t)col * c->ekf.ld], sizeof(double) * n);
    }
    return ASLAM_OK;
}

int aslam_set_state(aslam_ctx* c, int N, const double* mu, const double* sigma, const int* landmark_ids) {
    if (!c || !mu || !sigma || N < 3 || (N - 3) % 3 != 0) return fail(c, ASLAM_E_INVALID, "bad state");
    const int L = (N - 3) / 3;
    if (L > c->ekf.max_landmarks) return fail(c, ASLAM_E_CAPACITY, "state larger than max_landmarks");
    if (L > 0 && !landmark_ids) return fail(c, ASLAM_E_INVALID, "landmark ids required");
    { int rs = sync_streams(c); if (rs) return rs; }
    std::vector<double> tmp((size_t)c->ekf.ld * c->ekf.ld, 0.0);
    for (int col = 0; col < N; col++) std::memcpy(&tmp[(size_t)col * c->ekf.ld], sigma + (size_t)col * N, sizeof(double) * N);
    HIP_TRY(c, hipMemcpy(c->ekf.d_sigma, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->ekf.d_mu, mu, sizeof(double) * N, hipMemcpyHostToDevice));
    std::vector<int> id2idx(kIdTableSize, -1), idx2id(c->ekf.max_landmarks, -1);
    for (int i = 0; i < L; i++) {
        if (landmark_ids[i] < 0 || landmark_ids[i] >= kIdTableSize) return fail(c, ASLAM_E_INVALID, "landmark id out of range");
        if (id2idx[landmark_ids[i]] < 0) id2idx[landmark_ids[i]] = i;
        idx2id[i] = landmark_ids[i];
    }
    HIP_TRY(c, hipMemcpy(c->ekf.d_id2idx, id2idx.data(), sizeof(int) * kIdTableSize, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->ekf.d_idx2id, idx2id.data(), sizeof(int) * c->ekf.max_landmarks, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->ekf.d_L, &L, sizeof(int), hipMemcpyHostToDevice));
    int zero = 0;
    HIP_TRY(c, hipMemcpy(c->ekf.d_nlast, &zero, sizeof(int), hipMemcpyHostToDevice));
    c->mirror_dirty = true;
    return ASLAM_OK;
}

// ---- host-side result surface (what the node publishes; pure host arithmetic on a few values read back) ---------------
namespace {
// tf2::Quaternion::setRPY(roll, pitch, yaw) -> (x, y, z, w)
void quat_from_rpy(double roll, double pitch, double yaw, double q[4]) {
    const double hy = yaw * 0.5, hp = pitch * 0.5, hr = roll * 0.5;
    const double cy = std::cos(hy), sy = std::sin(hy), cp = std::cos(hp), sp = std::sin(hp), cr = std::cos(hr), sr = std::sin(hr);
    q[0] = sr * cp * cy - cr * sp * sy;
    q[1] = cr * sp * cy + sr * cp * sy;
    q[2] = cr * cp * sy - sr * sp * cy;
    q[3] = cr * cp * cy + sr * sp * sy;
}
// cv::Rodrigues (vector -> matrix), row-major
void rodrigues_host(const double r[3], double R[9]) {
    const double th = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (th < DBL_EPSILON) { for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0; return; }
    const double c = std::cos(th), s = std::sin(th), c1 = 1.0 - c, it = 1.0 / th;
    const double x = r[0] * it, y = r[1] * it, z = r[2] * it;
    R[0] = c + c1 * x * x;     R[1] = c1 * x * y - s * z; R[2] = c1 * x * z + s * y;
    R[3] = c1 * x * y + s * z; R[4] = c + c1 * y * y;     R[5] = c1 * y * z - s * x;
    R[6] = c1 * x * z - s * y; R[7] = c1 * y * z + s * x; R[8] = c + c1 * z * z;
}
// tf2::Matrix3x3::getRotation
void quat_from_matrix(const double m[9], double q[4]) {
    const double trace = m[0] + m[4] + m[8];
    if (trace > 0.0) {
        double s = std::sqrt(trace + 1.0);
        q[3] = s * 0.5;
        s = 0.5 / s;
        q[0] = (m[7] - m[5]) * s; q[1] = (m[2] - m[6]) * s; q[2] = (m[3] - m[1]) * s;
    } else {
        const int i = m[0] < m[4] ? (m[4] < m[8] ? 2 : 1) : (m[0] < m[8] ? 2 : 0);
        const int j = (i + 1) % 3, k = (i + 2) % 3;
        double s = std::sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
        q[i] = s * 0.5;
        s = 0.5 / s;
        q[3] = (m[k * 3 + j] - m[j * 3 + k]) * s;
        q[j] = (m[j * 3 + i] + m[i * 3 + j]) * s;
        q[k] = (m[k * 3 + i] + m[i * 3 + k]) * s;
    }
}
void quat_mul(const double a[4], const double b[4], double o[4]) {          // Hamilton product, (x, y, z, w)
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
void quat_rotate(const double q[4], const double v[3], double o[3]) {        // tf2::Matrix3x3(q) * v
    const double d = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3], s = 2.0 / d;
    const double xs = q[0] * s, ys = q[1] * s, zs = q[2] * s;
    const double wx = q[3] * xs, wy = q[3] * ys, wz = q[3] * zs, xx = q[0] * xs, xy = q[0] * ys, xz = q[0] * zs, yy = q[1] * ys, yz = q[1] * zs, zz = q[2] * zs;
    o[0] = (1.0 - (yy + zz)) * v[0] + (xy - wz) * v[1] + (xz + wy) * v[2];
    o[1] = (xy + wz) * v[0] + (1.0 - (xx + zz)) * v[1] + (yz - wx) * v[2];
    o[2] = (xz - wy) * v[0] + (yz + wx) * v[1] + (1.0 - (xx + yy)) * v[2];
}
void fill_marker(aslam_marker_msg& m, int id, double length, double x, double y, double z, const double q[4], float r, float g, float b,
                 float a, double lifetime) {                                 // ArucoSlam::GenerateMarker, aruco_slam.cpp:289-305
    m.id = id;
    m.scale[0] = length; m.scale[1] = length; m.scale[2] = 0.01;
    m.color[0] = r; m.color[1] = g; m.color[2] = b; m.color[3] = a;
    m.position[0] = x; m.position[1] = y; m.position[2] = z;
    for (int k = 0; k < 4; k++) m.orientation[k] = q[k];
    m.lifetime_sec = lifetime;
}
}  // namespace

int aslam_get_pose_msg(aslam_ctx* c, aslam_pose_msg* out) {                   // ArucoSlam::toRosPose, aruco_slam.cpp:378-410
    if (!c || !out) return fail(c, ASLAM_E_INVALID, "null argument");
    { int rs = sync_streams(c); if (rs) return rs; }
    double mu[3], S[9];
    HIP_TRY(c, hipMemcpy(mu, c->ekf.d_mu, sizeof(mu), hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy2D(S, 3 * sizeof(double), c->ekf.d_sigma, (size_t)c->ekf.ld * sizeof(double), 3 * sizeof(double), 3, hipMemcpyDeviceToHost));
    // S[col * 3 + row]
    out->position[0] = mu[0]; out->position[1] = mu[1]; out->position[2] = 0.1;
    quat_from_rpy(0, 0, mu[2], out->orientation);
    for (int i = 0; i < 36; i++) out->covariance[i] = 0.0;
    static const int at[3] = {0, 1, 5};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) out->covariance[at[i] * 6 + at[j]] = S[j * 3 + i];     // sigma_(i, j)
    return ASLAM_OK;
}

int aslam_get_map_markers(aslam_ctx* c, int max, int* n, aslam_marker_msg* out) {   // detected_map_, aruco_slam.cpp:265-281
    if (!c || !n) return fail(c, ASLAM_E_INVALID, "null argument");
    { int rs = sync_streams(c); if (rs) return rs; }
    int L = 0;
    HIP_TRY(c, hipMemcpy(&L, c->ekf.d_L, sizeof(int), hipMemcpyDeviceToHost));
    *n = L;
    if (!out || L == 0) return ASLAM_OK;
    std::vector<double> mu((size_t)3 + 3 * L);
    HIP_TRY(c, hipMemcpy(mu.data(), c->ekf.d_mu, mu.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int i = 0; i < L && i < max; i++) {
        double q[4];
        quat_from_rpy(0, 1.5708, mu[3 * i + 5], q);
        fill_marker(out[i], i, c->sp.marker_length, mu[3 * i + 3], mu[3 * i + 4], 0.3, q, 1.f, 0.5f, 1.f, 0.5f, 0.0);
    }
    return ASLAM_OK;
}

int aslam_get_detected_markers(aslam_ctx* c, int max, int* n, aslam_marker_msg* out) {   // detected_markers_, aruco_slam.cpp:325-347
    if (!c || !n) return fail(c, ASLAM_E_INVALID, "null argument");
    int M = 0;
    int r = aslam_get_detections(c, &M, nullptr, nullptr, nullptr, nullptr);
    if (r) return r;
    std::vector<int> ids(M);
    std::vector<double> rv((size_t)3 * M), tv((size_t)3 * M);
    if (M) { r = aslam_get_detections(c, &M, ids.data(), nullptr, rv.data(), tv.data()); if (r) return r; }
    int k = 0;
    for (int i = 0; i < M; i++) {
        const double* t = &tv[(size_t)3 * i];
        // the range gate precedes the visualisation (aruco_slam.cpp:327-333): float norm against the float threshold
        if ((float)std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]) > c->sp.useful_distance_threshold) continue;
        if (out && k < max) {
            double R[9], q[4], qo[4], p[3];
            rodrigues_host(&rv[(size_t)3 * i], R);
            quat_from_matrix(R, q);                                    // fillTransform + getRotation
            quat_rotate(c->init.r2c_q, t, p);                           // tf2::doTransform(pose, pose, transformStamped_r2c_)
            for (int a = 0; a < 3; a++) p[a] += c->init.r2c_t[a];
            quat_mul(c->init.r2c_q, q, qo);
            fill_marker(out[k], ids[i], c->sp.marker_length, p[0], p[1], p[2], qo, 1.f, 0.f, 0.f, 1.f, 0.1);
        }
        k++;
    }
    *n = k;
    return ASLAM_OK;
}

// markered_img_ (aruco_slam.cpp:318-319): cv::aruco::drawDetectedMarkers(img.clone(), corners, ids) with its default colours on a
// bgr8 buffer - the quad outline in (0, 255, 0), a 7 x 7 square outline in (0, 0, 255) around corner 0 (the marker's own
// top-left).  Host drawing from the last frame's detections; the "id=N" text of the original is not rendered (no font here).
namespace {
void put_px(uint8_t* img, int rows, int cols, size_t step, int x, int y, uint8_t b, uint8_t g, uint8_t r) {
    if (x < 0 || y < 0 || x >= cols || y >= rows) return;
    uint8_t* p = img + (size_t)y * step + (size_t)x * 3;
    p[0] = b; p[1] = g; p[2] = r;
}
void draw_line(uint8_t* img, int rows, int cols, size_t step, int x0, int y0, int x1, int y1, uint8_t b, uint8_t g, uint8_t r) {
    const int dx = std::abs(x1 - x0), dy = -std::abs(y1 - y0), sx = x0 < x1 ? 1 : -1, sy = y0 < y1 ? 1 : -1;
    int err = dx + dy;
    for (;;) {                                                    // 8-connected Bresenham
        put_px(img, rows, cols, step, x0, y0, b, g, r);
        if (x0 == x1 && y0 == y1) break;
        const int e2 = 2 * err;
        if (e2 >= dy) { err += dy; x0 += sx; }
        if (e2 <= dx) { err += dx; y0 += sy; }
    }
}
}  // namespace

int aslam_draw_detected_markers(aslam_ctx* c, uint8_t* bgr, int rows, int cols, size_t step) {
    if (!c || !bgr || rows <= 0 || cols <= 0 || step < (size_t)cols * 3) return fail(c, ASLAM_E_INVALID, "bad arguments");
    int M = 0;
    int r = aslam_get_detections(c, &M, nullptr, nullptr, nullptr, nullptr);
    if (r) return r;
    std::vector<float> corners((size_t)std::max(M, 1) * 8);
    if (M) { r = aslam_get_detections(c, &M, nullptr, corners.data(), nullptr, nullptr); if (r) return r; }
    for (int i = 0; i < M; i++) {
        const float* q = &corners[(size_t)i * 8];
        int px[4], py[4];
        for (int k = 0; k < 4; k++) { px[k] = (int)std::lrintf(q[2 * k]); py[k] = (int)std::lrintf(q[2 * k + 1]); }
        for (int k = 0; k < 4; k++) draw_line(bgr, rows, cols, step, px[k], py[k], px[(k + 1) & 3], py[(k + 1) & 3], 0, 255, 0);
        const int x0 = px[0] - 3, y0 = py[0] - 3, x1 = px[0] + 3, y1 = py[0] + 3;
        draw_line(bgr, rows, cols, step, x0, y0, x1, y0, 0, 0, 255); draw_line(bgr, rows, cols, step, x1, y0, x1, y1, 0, 0, 255);
        draw_line(bgr, rows, cols, step, x1, y1, x0, y1, 0, 0, 255); draw_line(bgr, rows, cols, step, x0, y1, x0, y0, 0, 0, 255);
    }
    return ASLAM_OK;
}

// MapLoader::loadMap (map_loader.cpp:7-81) as plain data: the ground-truth map file behind the latched `real_map` topic.
// One marker per line "id length x y [z [roll [pitch [yaw]]]]"; '#' starts a comment line, blank lines are skipped, a line
// that starts with anything else than a digit aborts the whole load with an EMPTY result; a line with fewer than four
// fields is skipped.  The optional fields reproduce the loader's crossed fallbacks: a missing roll zeroes YAW, a missing yaw
// zeroes ROLL (each leaving the other at whatever the previous line left there; 0 at the start).  Orientation =
// setRPY(roll, pitch, yaw); CUBE (length, length, 0.01), rgba (1, 1, 1, 0.5), frame "world", lifetime 0
// (MapLoader::generateMarker, map_loader.cpp:96-118).  Needs no device: ctx may be NULL.
int aslam_load_map_txt(aslam_ctx* c, const char* path, int max, int* n, aslam_marker_msg* out) {
    if (!path || !n) return c ? fail(c, ASLAM_E_INVALID, "null argument") : ASLAM_E_INVALID;
    *n = 0;
    std::ifstream f(path);
    if (!f.good()) return c ? fail(c, ASLAM_E_INVALID, std::string("cannot read ") + path) : ASLAM_E_INVALID;
    std::string line;
    int count = 0;
    int id = 0;
    double length = 0, x = 0, y = 0, z = 0, yaw = 0, pitch = 0, roll = 0;       // deliberately outside the loop: see above
    while (std::getline(f, line)) {
        std::istringstream s(line);
        char first = 0;
        if (!(s >> first)) continue;                          // blank line
        if (first == '#') continue;
        if (!isdigit((unsigned char)first)) { *n = 0; return ASLAM_OK; }       // "Malformed input": the map is cleared
        s.putback(first);
        if (!(s >> id >> length >> x >> y)) continue;
        if (!(s >> z)) z = 0;
        if (!(s >> roll)) yaw = 0;
        if (!(s >> pitch)) pitch = 0;
        if (!(s >> yaw)) roll = 0;
        if (out && count < max) {
            double q[4];
            quat_from_rpy(roll, pitch, yaw, q);
            fill_marker(out[count], id, length, x, y, z, q, 1.f, 1.f, 1.f, 0.5f, 0.0);
        }
        count++;
    }
    *n = count;
    return ASLAM_OK;
}

// ---- persistence (no counterpart in the reference: warm starts of large maps) ----------------------------
int aslam_save_state(aslam_ctx* c, const char* path) {
    if (!c || !path) return fail(c, ASLAM_E_INVALID, "null argument");
    int N = 0;
    int r = aslam_get_state(c, &N, nullptr, nullptr);
    if (r) return r;
    const int L = (N - 3) / 3;
    std::vector<double> mu(N), S((size_t)N * N);
    std::vector<int> ids(std::max(L, 1));
    r = aslam_get_state(c, &N, mu.data(), S.data());
    if (r) return r;
    int L2 = 0;
    r = aslam_get_landmark_ids(c, &L2, ids.data());
    if (r) return r;
    FILE* f = std::fopen(path, "wb");
    if (!f) return fail(c, ASLAM_E_INVALID, std::string("cannot write ") + path);
    const char magic[8] = {'A', 'S', 'L', 'A', 'M', 'S', 'T', '1'};
    const int hdr[2] = {N, c->is_init ? 1 : 0};
    bool ok = std::fwrite(magic, 1, 8, f) == 8 && std::fwrite(hdr, sizeof(int), 2, f) == 2 &&
              std::fwrite(mu.data(), sizeof(double), mu.size(), f) == mu.size() &&
              std::fwrite(S.data(), sizeof(double), S.size(), f) == S.size() &&
              std::fwrite(ids.data(), sizeof(int), (size_t)L, f) == (size_t)L;
    ok = (std::fclose(f) == 0) && ok;
    return ok ? ASLAM_OK : fail(c, ASLAM_E_INVALID, std::string("short write to ") + path);
}

int aslam_load_state(aslam_ctx* c, const char* path) {
    if (!c || !path) return fail(c, ASLAM_E_INVALID, "null argument");
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(c, ASLAM_E_INVALID, std::string("cannot read ") + path);
    char magic[8];
    int hdr[2] = {0, 0};
    bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "ASLAMST1", 8) == 0 && std::fread(hdr, sizeof(int), 2, f) == 2;
    const int N = hdr[0];
    ok = ok && N >= 3 && (N - 3) % 3 == 0 && (N - 3) / 3 <= c->ekf.max_landmarks;
    std::vector<double> mu, S;
    std::vector<int> ids;
    if (ok) {
        const int L = (N - 3) / 3;
        mu.resize(N); S.resize((size_t)N * N); ids.resize(std::max(L, 1));
        ok = std::fread(mu.data(), sizeof(double), mu.size(), f) == mu.size() && std::fread(S.data(), sizeof(double), S.size(), f) == S.size() &&
             std::fread(ids.data(), sizeof(int), (size_t)L, f) == (size_t)L;
    }
    std::fclose(f);
    if (!ok) return fail(c, ASLAM_E_INVALID, std::string("not a state file that fits this context: ") + path);
    int r = aslam_set_state(c, N, mu.data(), S.data(), ids.data());
    if (r) return r;
    c->is_init = hdr[1] != 0;
    return ASLAM_OK;
}

int aslam_get_slot_detections(aslam_ctx* c, int slot, int* M, int* ids, float* corners, double* rvecs, double* tvecs) {
    if (!c || !M) return fail(c, ASLAM_E_INVALID, "null argument");
    int r = check_slot_range(c, slot, 1);
    if (r) return r;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    unsigned n = 0;
    HIP_TRY(c, hipMemcpy(&n, c->d_nmarkers + slot, sizeof(unsigned), hipMemcpyDeviceToHost));
    n = std::min(n, (unsigned)kMarkerMax);
    std::vector<Marker> h(n);
    if (n) HIP_TRY(c, hipMemcpy(h.data(), c->d_markers + (size_t)slot * kMarkerMax, n * sizeof(Marker), hipMemcpyDeviceToHost));
    *M = (int)n;
    for (unsigned i = 0; i < n; i++) {
        if (ids) ids[i] = h[i].id;
        if (corners) std::memcpy(corners + 8 * i, h[i].c, sizeof(float) * 8);
        if (rvecs) std::memcpy(rvecs + 3 * i, h[i].rvec, sizeof(double) * 3);
        if (tvecs) std::memcpy(tvecs + 3 * i, h[i].tvec, sizeof(double) * 3);
    }
    return ASLAM_OK;
}

int aslam_get_detections(aslam_ctx* c, int* M, int* ids, float* corners, double* rvecs, double* tvecs) {
    if (!c) return ASLAM_E_INVALID;
    return aslam_get_slot_detections(c, c->last_first + std::max(c->last_count, 1) - 1, M, ids, corners, rvecs, tvecs);
}

int aslam_get_slot_raw_observations(aslam_ctx* c, int slot, int* n_out, int* ids, int* valid, double* xyth, double* Rdiag) {
    if (!c || !n_out) return fail(c, ASLAM_E_INVALID, "null argument");
    int r = check_slot_range(c, slot, 1);
    if (r) return r;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    unsigned n = 0;
    HIP_TRY(c, hipMemcpy(&n, c->d_nmarkers + slot, sizeof(unsigned), hipMemcpyDeviceToHost));
    n = std::min(n, (unsigned)kMarkerMax);
    std::vector<ObsRaw> h(n);
    if (n) HIP_TRY(c, hipMemcpy(h.data(), c->d_obs + (size_t)slot * kMarkerMax, n * sizeof(ObsRaw), hipMemcpyDeviceToHost));
    *n_out = (int)n;
    for (unsigned i = 0; i < n; i++) {
        if (ids) ids[i] = h[i].id;
        if (valid) valid[i] = h[i].valid;
        if (xyth) { xyth[3 * i] = h[i].x; xyth[3 * i + 1] = h[i].y; xyth[3 * i + 2] = h[i].th; }
        if (Rdiag) std::memcpy(Rdiag + 3 * i, h[i].r, sizeof(double) * 3);
    }
    return ASLAM_OK;
}

int aslam_get_observations(aslam_ctx* c, int* n_out, int* ids, int* idx, int* action, double* xyth, double* Rdiag) {
    if (!c || !n_out) return fail(c, ASLAM_E_INVALID, "null argument");
    { int rs = sync_streams(c); if (rs) return rs; }
    int n = 0;
    HIP_TRY(c, hipMemcpy(&n, c->ekf.d_npop, sizeof(int), hipMemcpyDeviceToHost));
    std::vector<PopRec> h(n);
    if (n) HIP_TRY(c, hipMemcpy(h.data(), c->ekf.d_pop, n * sizeof(PopRec), hipMemcpyDeviceToHost));
    *n_out = n;
    for (int i = 0; i < n; i++) {
        if (ids) ids[i] = h[i].id;
        if (idx) idx[i] = h[i].index;
        if (action) action[i] = h[i].action;
        if (xyth) { xyth[3 * i] = h[i].z[0]; xyth[3 * i + 1] = h[i].z[1]; xyth[3 * i + 2] = h[i].z[2]; }
        if (Rdiag) std::memcpy(Rdiag + 3 * i, h[i].r, sizeof(double) * 3);
    }
    return ASLAM_OK;
}

int aslam_get_slot_ekf_stats(aslam_ctx* c, int first, int count, int* stats) {
    if (!c || !stats) return fail(c, ASLAM_E_INVALID, "null argument");
    int r = check_slot_range(c, first, count);
    if (r) return r;
    { int rs = sync_streams(c); if (rs) return rs; }
    HIP_TRY(c, hipMemcpy(stats, c->ekf.d_slot_stat + (size_t)4 * first, sizeof(int) * 4 * count, hipMemcpyDeviceToHost));
    return ASLAM_OK;
}

int aslam_get_landmark_ids(aslam_ctx* c, int* L, int* ids) {
    if (!c || !L) return fail(c, ASLAM_E_INVALID, "null argument");
    { int rs = sync_streams(c); if (rs) return rs; }
    int n = 0;
    HIP_TRY(c, hipMemcpy(&n, c->ekf.d_L, sizeof(int), hipMemcpyDeviceToHost));
    *L = n;
    if (ids && n) HIP_TRY(c, hipMemcpy(ids, c->ekf.d_idx2id, sizeof(int) * n, hipMemcpyDeviceToHost));
    return ASLAM_OK;
}

int aslam_detect_batch(aslam_ctx* c, const uint8_t* frames, int nframes, int rows, int cols, int channels, size_t step,
                       size_t frame_stride, int max_per_frame, int* counts, int* ids, float* corners, double* rvecs, double* tvecs) {
    if (!c || !frames || !counts || nframes <= 0 || max_per_frame <= 0) return fail(c, ASLAM_E_INVALID, "bad arguments");
    for (int f0 = 0; f0 < nframes; f0 += c->max_batch) {
        int nb = std::min(c->max_batch, nframes - f0);
        int r = aslam_stage_frames(c, 0, frames + (size_t)f0 * frame_stride, nb, rows, cols, channels, step, frame_stride);
        if (r) return r;
        r = run_detect(c, 0, nb);
        if (r) return r;
        r = sync_and_check(c);
        if (r) return r;
        for (int i = 0; i < nb; i++) {
            int M = 0;
            std::vector<int> hid(kMarkerMax);
            std::vector<float> hc((size_t)kMarkerMax * 8);
            std::vector<double> hr((size_t)kMarkerMax * 3), ht((size_t)kMarkerMax * 3);
            r = aslam_get_slot_detections(c, i, &M, hid.data(), hc.data(), hr.data(), ht.data());
            if (r) return r;
            const int f = f0 + i;
            counts[f] = M;
            const int m = std::min(M, max_per_frame);
            if (ids) std::memcpy(ids + (size_t)f * max_per_frame, hid.data(), sizeof(int) * m);
            if (corners) std::memcpy(corners + (size_t)f * max_per_frame * 8, hc.data(), sizeof(float) * 8 * m);
            if (rvecs) std::memcpy(rvecs + (size_t)f * max_per_frame * 3, hr.data(), sizeof(double) * 3 * m);
            if (tvecs) std::memcpy(tvecs + (size_t)f * max_per_frame * 3, ht.data(), sizeof(double) * 3 * m);
        }
    }
    return ASLAM_OK;
}

int aslam_export_map(aslam_ctx* c, void* dst, int dst_is_device) {
    if (!c || !dst) return fail(c, ASLAM_E_INVALID, "null argument");
    { int rp = finalize_pending(c); if (rp) return rp; }
    launch_ekf_export_map(c->stream_ekf, c->ekf);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(dst, c->ekf.d_maprec, (size_t)ASLAM_MAP_RECORD_BYTES * c->ekf.max_landmarks,
                              dst_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream_ekf));
    HIP_TRY(c, hipStreamSynchronize(c->stream_ekf));
    return ASLAM_OK;
}

int aslam_export_map_async(aslam_ctx* c, void* d_dst, int buffer) {
    if (!c || !d_dst || buffer < 0 || buffer > 1) return fail(c, ASLAM_E_INVALID, "bad arguments");
    if (!c->ev_export[buffer]) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_export[buffer], hipEventDisableTiming));
    { int rp = finalize_pending(c); if (rp) return rp; }
    launch_ekf_export_map(c->stream_ekf, c->ekf);              // ordered after the EKF steps enqueued so far
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(d_dst, c->ekf.d_maprec, (size_t)ASLAM_MAP_RECORD_BYTES * c->ekf.max_landmarks, hipMemcpyDeviceToDevice, c->stream_ekf));
    HIP_TRY(c, hipEventRecord(c->ev_export[buffer], c->stream_ekf));
    return ASLAM_OK;
}

int aslam_export_wait(aslam_ctx* c, int buffer) {
    if (!c || buffer < 0 || buffer > 1) return fail(c, ASLAM_E_INVALID, "bad arguments");
    if (c->ev_export[buffer]) HIP_TRY(c, hipEventSynchronize(c->ev_export[buffer]));
    return ASLAM_OK;
}

// ---- landmark-map gather over RCCL, reachable from C / C++ (the Python side can use torch.distributed instead) ----
namespace {
struct NcclUid { char internal[128]; };                                    // ncclUniqueId (rccl.h): 128 opaque bytes
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(NcclUid*) = nullptr;
    int (*CommInitRank)(void**, int, NcclUid, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl* rccl() {
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names) { r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (r.h) break; }      // a copy the process already uses (torch's)
        if (!r.h) for (const char* n : names) { r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.h) break; }
        if (r.h) {
            r.GetUniqueId = reinterpret_cast<int (*)(NcclUid*)>(dlsym(r.h, "ncclGetUniqueId"));
            r.CommInitRank = reinterpret_cast<int (*)(void**, int, NcclUid, int)>(dlsym(r.h, "ncclCommInitRank"));
            r.AllGather = reinterpret_cast<int (*)(const void*, void*, size_t, int, void*, hipStream_t)>(dlsym(r.h, "ncclAllGather"));
            r.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(r.h, "ncclCommDestroy"));
            r.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(r.h, "ncclGetErrorString"));
            if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy) r.h = nullptr;
        }
    }
    return r.h ? &r : nullptr;
}
int rccl_fail(aslam_ctx* c, const char* what, int rc) {
    Rccl* r = rccl();
    return fail(c, ASLAM_E_HIP, std::string(what) + ": " + (r && r->GetErrorString ? r->GetErrorString(rc) : "rccl error"));
}
}  // namespace

int aslam_comm_get_unique_id(void* id) {
    Rccl* r = rccl();
    if (!id) return ASLAM_E_INVALID;
    if (!r) return ASLAM_E_STATE;
    return r->GetUniqueId(static_cast<NcclUid*>(id)) == 0 ? ASLAM_OK : ASLAM_E_HIP;
}

int aslam_comm_create(aslam_ctx* c, const void* id, int world, int rank) {
    if (!c || !id || world < 1 || rank < 0 || rank >= world) return fail(c, ASLAM_E_INVALID, "bad arguments");
    Rccl* r = rccl();
    if (!r) return fail(c, ASLAM_E_STATE, "librccl.so not found (the map gather needs RCCL)");
    if (c->comm) return fail(c, ASLAM_E_STATE, "communicator already created");
    HIP_TRY(c, hipSetDevice(c->init.device_id));
    NcclUid uid;
    std::memcpy(&uid, id, sizeof(uid));
    int rc = r->CommInitRank(&c->comm, world, uid, rank);
    if (rc != 0) { c->comm = nullptr; return rccl_fail(c, "ncclCommInitRank", rc); }
    c->comm_world = world; c->comm_rank = rank;
    HIP_TRY(c, dalloc(&c->d_gather, (size_t)world * c->ekf.max_landmarks * ASLAM_MAP_RECORD_BYTES));
    return ASLAM_OK;
}

int aslam_comm_gather_maps(aslam_ctx* c, void* dst, int dst_is_device) {
    if (!c || !dst) return fail(c, ASLAM_E_INVALID, "null argument");
    if (!c->comm) return fail(c, ASLAM_E_STATE, "aslam_comm_create first");
    Rccl* r = rccl();
    const size_t nb = (size_t)ASLAM_MAP_RECORD_BYTES * c->ekf.max_landmarks;
    { int rp = finalize_pending(c); if (rp) return rp; }
    launch_ekf_export_map(c->stream_ekf, c->ekf);                  // ordered after the EKF steps enqueued so far
    HIP_TRY(c, hipGetLastError());
    int rc = r->AllGather(c->ekf.d_maprec, c->d_gather, nb, /* ncclInt8 */ 0, c->comm, c->stream_ekf);
    if (rc != 0) return rccl_fail(c, "ncclAllGather", rc);
    HIP_TRY(c, hipMemcpyAsync(dst, c->d_gather, nb * c->comm_world, dst_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream_ekf));
    HIP_TRY(c, hipStreamSynchronize(c->stream_ekf));
    return ASLAM_OK;
}

int aslam_comm_destroy(aslam_ctx* c) {
    if (!c) return ASLAM_E_INVALID;
    if (c->comm) {
        hipStreamSynchronize(c->stream_ekf);
        Rccl* r = rccl();
        if (r) r->CommDestroy(c->comm);
        c->comm = nullptr;
    }
    if (c->d_gather) { hipFree(c->d_gather); c->d_gather = nullptr; }
    return ASLAM_OK;
}

// ---- instrumentation -----------------------------------------------------------------------------------
int aslam_debug_get_nbr(aslam_ctx* c, int slot, int scale, uint8_t* out) {
    if (!c || !out || scale < 0 || scale >= kScales) return fail(c, ASLAM_E_INVALID, "bad arguments");
    int r = check_slot_range(c, slot, 1);
    if (r) return r;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const DetectCfg& g = c->cfg;
    const size_t pb = nbr_plane_bytes(g.rows, g.pitch);
    std::vector<uint8_t> tiled(pb);
    HIP_TRY(c, hipMemcpy(tiled.data(), c->d_nbr + ((size_t)slot * kScales + scale) * pb, pb, hipMemcpyDeviceToHost));
    for (int y = 0; y < g.rows; y++)
        for (int x = 0; x < g.cols; x++) out[(size_t)y * g.cols + x] = tiled[nbr_index(x, y, g.pitch)];     // un-tile for the caller
    return ASLAM_OK;
}

// contours of one (slot, scale) of the last detection that covered the slot, sorted into OpenCV order (descending key)
int aslam_debug_get_contours(aslam_ctx* c, int slot, int scale, int max_contours, long long max_points, int* n_contours,
                             int* sizes, int* keys, int* points_xy, long long* n_points) {
    if (!c || !n_contours) return fail(c, ASLAM_E_INVALID, "bad arguments");
    int rr = check_slot_range(c, slot, 1);
    if (rr) return rr;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    unsigned nc = 0;
    HIP_TRY(c, hipMemcpy(&nc, c->d_ncontours + slot, sizeof(unsigned), hipMemcpyDeviceToHost));
    nc = std::min(nc, c->cfg.cap_contours);
    std::vector<ContourRec> recs(nc);
    if (nc) HIP_TRY(c, hipMemcpy(recs.data(), c->d_contours + (size_t)slot * c->cfg.cap_contours, nc * sizeof(ContourRec), hipMemcpyDeviceToHost));
    std::vector<ContourRec> sel;
    for (auto& r : recs) if (r.scale == (unsigned)scale) sel.push_back(r);
    std::sort(sel.begin(), sel.end(), [](const ContourRec& a, const ContourRec& b) { return a.key > b.key; });
    long long tot = 0;
    int n = 0;
    std::vector<unsigned> pts;
    for (auto& r : sel) {
        if (n >= max_contours || tot + (long long)r.n > max_points) return fail(c, ASLAM_E_CAPACITY, "debug buffer too small");
        pts.resize(r.n);
        if (r.n) HIP_TRY(c, hipMemcpy(pts.data(), c->d_points + (size_t)slot * c->cfg.cap_points + r.off, r.n * sizeof(unsigned), hipMemcpyDeviceToHost));
        if (sizes) sizes[n] = (int)r.n;
        if (keys) keys[n] = (int)r.key;
        if (points_xy)
            for (unsigned i = 0; i < r.n; i++) {
                points_xy[2 * (tot + i)] = (int)(short)(pts[i] & 0xFFFFu);
                points_xy[2 * (tot + i) + 1] = (int)(short)(pts[i] >> 16);
            }
        tot += r.n;
        n++;
    }
    *n_contours = n;
    if (n_points) *n_points = tot;
    return ASLAM_OK;
}

int aslam_debug_get_candidates(aslam_ctx* c, int slot, int stage, int max, int* n_out, float* corners, int* sizes, int* ids) {
    if (!c || !n_out) return fail(c, ASLAM_E_INVALID, "bad arguments");
    int r = check_slot_range(c, slot, 1);
    if (r) return r;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (stage == 0) {
        unsigned n = 0;
        HIP_TRY(c, hipMemcpy(&n, c->d_ncand + slot, sizeof(unsigned), hipMemcpyDeviceToHost));
        n = std::min(n, (unsigned)kCandMax);
        std::vector<CandRec> h(n);
        if (n) HIP_TRY(c, hipMemcpy(h.data(), c->d_cands + (size_t)slot * kCandMax, n * sizeof(CandRec), hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end(), [](const CandRec& a, const CandRec& b) { return a.ordkey < b.ordkey; });
        *n_out = (int)n;
        for (unsigned i = 0; i < n && (int)i < max; i++) {
            if (corners) for (int k = 0; k < 4; k++) { corners[8 * i + 2 * k] = h[i].x[k]; corners[8 * i + 2 * k + 1] = h[i].y[k]; }
            if (sizes) sizes[i] = (int)h[i].n;
            if (ids) ids[i] = -1;
        }
    } else {
        unsigned n = 0;
        HIP_TRY(c, hipMemcpy(&n, c->d_nfinal + slot, sizeof(unsigned), hipMemcpyDeviceToHost));
        n = std::min(n, (unsigned)kCandMax);
        std::vector<FinalCand> h(n);
        if (n) HIP_TRY(c, hipMemcpy(h.data(), c->d_finals + (size_t)slot * kCandMax, n * sizeof(FinalCand), hipMemcpyDeviceToHost));
        *n_out = (int)n;
        for (unsigned i = 0; i < n && (int)i < max; i++) {
            if (corners) std::memcpy(corners + 8 * i, h[i].c, sizeof(float) * 8);
            if (sizes) sizes[i] = h[i].n;
            if (ids) ids[i] = h[i].id;
        }
    }
    return ASLAM_OK;
}

int aslam_debug_inject_observations(aslam_ctx* c, int slot, int n, const int* ids, const int* valid, const double* xyth,
                                    const double* Rdiag) {
    if (!c || n < 0 || n > kMarkerMax || (n && (!ids || !valid || !xyth || !Rdiag))) return fail(c, ASLAM_E_INVALID, "bad arguments");
    int r = check_slot_range(c, slot, 1);
    if (r) return r;
    r = sync_streams(c);
    if (r) return r;
    std::vector<ObsRaw> h(n);
    for (int i = 0; i < n; i++) {
        h[i].id = ids[i]; h[i].valid = valid[i];
        h[i].x = xyth[3 * i]; h[i].y = xyth[3 * i + 1]; h[i].th = xyth[3 * i + 2];
        h[i].r[0] = Rdiag[3 * i]; h[i].r[1] = Rdiag[3 * i + 1]; h[i].r[2] = Rdiag[3 * i + 2];
    }
    unsigned un = (unsigned)n;
    if (n) HIP_TRY(c, hipMemcpy(c->d_obs + (size_t)slot * kMarkerMax, h.data(), n * sizeof(ObsRaw), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_nmarkers + slot, &un, sizeof(unsigned), hipMemcpyHostToDevice));
    return ASLAM_OK;
}

int aslam_debug_get_frame_counts(aslam_ctx* c, int slot, unsigned* out /* 6 */) {
    if (!c || !out || slot < 0 || slot >= c->max_batch) return ASLAM_E_INVALID;
    const unsigned* src[6] = {c->d_nstarts, c->d_ncontours, c->d_npoints, c->d_nwrite, c->d_ncand, c->d_link_todo};
    for (int i = 0; i < 6; i++)
        if (hipMemcpy(out + i, src[i] + slot, sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) return ASLAM_E_HIP;
    return ASLAM_OK;
}
int aslam_debug_get_counters(aslam_ctx* c, unsigned* out) {
    return hipMemcpy(out, c->d_ctr, 32, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
}
int aslam_profile_enable(aslam_ctx* c, int on) {
    if (!c) return ASLAM_E_INVALID;
    c->prof_on = on != 0;
    if (c->prof_on && c->prof_empty_ms == 0.0) {
        // calibrate the event pair itself: 64 empty spans on the (idle) EKF stream, median
        int r = sync_streams(c);
        if (r) return r;
        std::vector<float> v;
        for (int k = 0; k < 64; k++) {
            hipEvent_t a, b;
            hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a, c->stream_ekf);
            hipEventRecord(b, c->stream_ekf);
            hipEventSynchronize(b);
            float ms = 0;
            if (hipEventElapsedTime(&ms, a, b) == hipSuccess) v.push_back(ms);
            hipEventDestroy(a); hipEventDestroy(b);
        }
        if (!v.empty()) { std::sort(v.begin(), v.end()); c->prof_empty_ms = v[v.size() / 2]; }
    }
    return ASLAM_OK;
}
int aslam_get_plan_stats(aslam_ctx* c, long long out[4]) {
    if (!c || !out) return ASLAM_E_INVALID;
    int r = finalize_pending(c);
    if (r) return r;
    for (int i = 0; i < 4; i++) out[i] = c->plan_stats[i];
    return ASLAM_OK;
}
int aslam_profile_reset(aslam_ctx* c) {
    if (!c) return ASLAM_E_INVALID;
    sync_streams(c);
    prof_collect(c);
    for (int i = 0; i < 4; i++) c->plan_stats[i] = 0;
    for (int i = 0; i < P_COUNT; i++) { c->prof_calls[i] = 0; c->prof_ms[i] = 0; }
    return ASLAM_OK;
}
int aslam_profile_get(aslam_ctx* c, int max, const char** names, int* calls, double* total_ms) {
    if (!c) return ASLAM_E_INVALID;
    sync_streams(c);
    prof_collect(c);
    int n = std::min(max, (int)P_COUNT);
    for (int i = 0; i < n; i++) {
        if (names) names[i] = kProfNames[i];
        if (calls) calls[i] = c->prof_calls[i];
        // a start/stop event pair measures a few microseconds even around nothing; that calibrated amount is taken off every
        // span so that the averages agree with rocprofv3's kernel durations
        if (total_ms) total_ms[i] = std::max(0.0, c->prof_ms[i] - c->prof_calls[i] * c->prof_empty_ms);
    }
    return P_COUNT;
}

int aslam_synth_render(aslam_ctx* c, int slot, int rows, int cols, const double K[9], int n_markers, const int* ids,
                       const double* poses, double marker_length, int background, int noise_amp, unsigned seed, int ss,
                       uint8_t* out_host) {
    if (!c || !K || n_markers < 0 || n_markers > 256 || (n_markers && (!ids || !poses)) || ss < 1 || ss > 8)
        return fail(c, ASLAM_E_INVALID, "bad arguments");
    int r = check_slot_range(c, slot, 1);
    if (r) return r;
    if (rows != c->rows || cols != c->cols || c->channels != 1) {
        r = configure_frames(c, rows, cols, 1);
        if (r) return r;
    }
    c->in_frame_bytes = (size_t)rows * cols;
    r = quiesce_slots(c, slot, 1);
    if (r) return r;
    const int nc = c->dict_ms + 2;
    std::vector<SynthMarker> mk(n_markers);
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    for (int m = 0; m < n_markers; m++) {
        if (ids[m] < 0 || ids[m] >= c->dict_n) return fail(c, ASLAM_E_INVALID, "marker id outside the dictionary");
        const double* P = poses + 12 * m;      // R row-major then t
        double H[9] = {P[0], P[1], P[9], P[3], P[4], P[10], P[6], P[7], P[11]};    // [r1 r2 t]
        double det = H[0] * (H[4] * H[8] - H[5] * H[7]) - H[1] * (H[3] * H[8] - H[5] * H[6]) + H[2] * (H[3] * H[7] - H[4] * H[6]);
        if (det == 0) return fail(c, ASLAM_E_INVALID, "degenerate marker pose");
        double d = 1.0 / det;
        SynthMarker& S = mk[m];
        S.Hinv[0] = (H[4] * H[8] - H[5] * H[7]) * d; S.Hinv[1] = (H[2] * H[7] - H[1] * H[8]) * d; S.Hinv[2] = (H[1] * H[5] - H[2] * H[4]) * d;
        S.Hinv[3] = (H[5] * H[6] - H[3] * H[8]) * d; S.Hinv[4] = (H[0] * H[8] - H[2] * H[6]) * d; S.Hinv[5] = (H[2] * H[3] - H[0] * H[5]) * d;
        S.Hinv[6] = (H[3] * H[7] - H[4] * H[6]) * d; S.Hinv[7] = (H[1] * H[6] - H[0] * H[7]) * d; S.Hinv[8] = (H[0] * H[4] - H[1] * H[3]) * d;
        const double ext = marker_length * 0.5 + marker_length / nc;
        double x0 = 1e30, y0 = 1e30, x1 = -1e30, y1 = -1e30;
        for (int k = 0; k < 4; k++) {
            double X = (k == 0 || k == 3) ? -ext : ext, Y = (k < 2) ? ext : -ext;
            double px = H[0] * X + H[1] * Y + H[2], py = H[3] * X + H[4] * Y + H[5], pw = H[6] * X + H[7] * Y + H[8];
            double u = fx * px / pw + cx, v = fy * py / pw + cy;
            x0 = std::min(x0, u); x1 = std::max(x1, u); y0 = std::min(y0, v); y1 = std::max(y1, v);
        }
        S.bbox[0] = (int)std::floor(x0) - 2; S.bbox[1] = (int)std::floor(y0) - 2;
        S.bbox[2] = (int)std::ceil(x1) + 2;  S.bbox[3] = (int)std::ceil(y1) + 2;
        S.bits[0] = c->dict_cells[(size_t)ids[m] * 2];
        S.bits[1] = c->dict_cells[(size_t)ids[m] * 2 + 1];
    }
    if (n_markers) HIP_TRY(c, hipMemcpyAsync(c->d_synth, mk.data(), mk.size() * sizeof(SynthMarker), hipMemcpyHostToDevice, c->stream));
    uint8_t* dst = c->d_in + (size_t)slot * c->in_frame_bytes;
    launch_render(c->stream, dst, rows, cols, fx, fy, cx, cy, n_markers, c->d_synth, nc, marker_length, background, noise_amp, seed, ss);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (out_host) HIP_TRY(c, hipMemcpy(out_host, dst, (size_t)rows * cols, hipMemcpyDeviceToHost));
    return ASLAM_OK;
}

} // extern "C"
