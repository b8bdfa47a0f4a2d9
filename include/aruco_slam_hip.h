/* aruco_slam_hip.h — C-ABI of libaruco_slam_hip.so, the MI355X (gfx950) drop-in for the hot path of
 * gitAugust/Aruco_Slam: per-frame ArUco detect + per-marker pose + SE(2) EKF-SLAM predict/update.
 *
 * The reference has no FFI layer; its boundary is the C++ class `ArucoSlam`
 * (include/aruco_slam/aruco_slam.h:101-193) whose headers drag in Eigen, OpenCV and ROS.  This header is
 * the POD-only surface a maintainer binds instead; include/aruco_slam/aruco_slam.h wraps it in the
 * reference's own class (same names and signatures).  Every entry point cites the reference interface it replaces.
 *
 * Conventions: every function returns 0 on success and a negative ASLAM_E_* code on failure (never
 * throws); aslam_last_error() gives the text.  A context is bound to one HIP device (it owns a few HIP streams there:
 * detection, EKF chain, uploads) and is NOT thread-safe (the reference relies on the single-threaded ROS spinner, aruco_slam_node.cpp:79).
 * Pointers are plain host pointers unless the parameter name starts with d_.  There is no CPU fallback: on a
 * machine without a usable gfx950 device aslam_create fails with ASLAM_E_NO_DEVICE.
 */
#ifndef ARUCO_SLAM_HIP_H
#define ARUCO_SLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct aslam_ctx aslam_ctx;

enum {
    ASLAM_OK = 0,
    ASLAM_E_INVALID = -1,     /* bad argument */
    ASLAM_E_NO_DEVICE = -2,   /* no HIP device / HIP runtime error at start-up */
    ASLAM_E_HIP = -3,         /* HIP runtime error during a call */
    ASLAM_E_CAPACITY = -4,    /* a device-side list overflowed (see aslam_last_error) */
    ASLAM_E_STATE = -5        /* call out of order (e.g. image before camera parameters) */
};

/* Mirrors struct ArucoSlamIniteData (aruco_slam.h:40-60) plus device/capacity settings.
 * r2c_* is transformStamped_r2c (base_link <- camera_frame_optical); only translation x,y enter the
 * arithmetic (aruco_slam.cpp:359-360), the rest is carried for the visualisation getters. */
typedef struct {
    double Q_k, R_x, R_y, R_theta;       /* parameters.yaml:5-8 */
    double kl, kr, b;                    /* parameters.yaml:11-13 */
    double marker_length;                /* parameters.yaml:17 */
    int    markers_dictionary;           /* parameters.yaml:16 ; 16 = DICT_ARUCO_ORIGINAL */
    float  useful_distance_threshold;    /* aruco_slam.h:58, default 3 (the YAML key never takes effect) */
    double r2c_t[3], r2c_q[4];
    int    device_id;
    int    max_landmarks;                /* capacity of the EKF state: N_max = 3 + 3*max_landmarks */
    int    max_rows, max_cols;           /* largest frame that will be handed over */
    int    max_batch;                    /* frames staged / processed per call of the *_staged functions */
    int    persistent_waves;             /* wavefronts of the work-queue kernels; 0 = default (4096) */
    int    max_updates_per_frame;        /* EKF corrections fused per frame; <= 24 selects the 3-kernel fast chain,
                                            larger values (up to 128) the general 5-kernel chain; exceeding it at run
                                            time is reported as ASLAM_E_CAPACITY */
    unsigned cap_starts_per_frame;       /* 0 = defaults */
    unsigned cap_contours_per_frame;
    unsigned cap_points_per_frame;
    int    ekf_reserved_cus_per_xcd;     /* CUs of every XCD (32 each) kept free of batched detection while an EKF chain runs
                                            beside it (CU-masked detection stream); 0 = default (16), negative = no masking */
} aslam_init;

/* fills *init with the reference's shipped parameters.yaml values and sane capacities */
void aslam_default_init(aslam_init* init);

/* ArucoSlam::ArucoSlam(const ArucoSlamIniteData&)  — aruco_slam.h:109, aruco_slam.cpp:3-19 */
int  aslam_create(const aslam_init* init, aslam_ctx** out);
void aslam_destroy(aslam_ctx* ctx);
const char* aslam_last_error(const aslam_ctx* ctx);

/* ArucoSlam::setCameraParameters(pair<cv::Mat K, cv::Mat D>) — aruco_slam.h:129-133; K row-major 3x3, D n x 1 */
int aslam_set_camera(aslam_ctx* ctx, const double K[9], const double* D, int nD);

/* cv::aruco::DetectorParameters (OpenCV 3.2.0 field names and defaults).  The reference passes none (aruco_slam.cpp:313 uses
 * DetectorParameters::create()), so the defaults are what parity is checked with; this is the knob a maintainer gets
 * instead of the cv::Ptr.  Limits of the build, refused (never ignored) when exceeded: at most 3 threshold windows of 3..23
 * pixels (LDS tile halo, three mask planes), perspectiveRemovePixelPerCell 2..8, markerBorderBits 1, maxMarkerPerimeterRate x
 * the larger side of the frame at most 65534 points (16-bit distances along a border).  doCornerRefinement (off in the reference) runs cv::cornerSubPix on the kept markers, one lane per corner,
 * with cornerRefinementWinSize 1..7. */
typedef struct {
    int    adaptiveThreshWinSizeMin, adaptiveThreshWinSizeMax, adaptiveThreshWinSizeStep;
    double adaptiveThreshConstant;
    double minMarkerPerimeterRate, maxMarkerPerimeterRate;
    double polygonalApproxAccuracyRate;
    double minCornerDistanceRate;
    int    minDistanceToBorder;
    double minMarkerDistanceRate;
    int    doCornerRefinement;
    int    cornerRefinementWinSize, cornerRefinementMaxIterations;
    double cornerRefinementMinAccuracy;
    int    markerBorderBits;
    int    perspectiveRemovePixelPerCell;
    double perspectiveRemoveIgnoredMarginPerCell;
    double maxErroneousBitsInBorderRate;
    double minOtsuStdDev;
    double errorCorrectionRate;
} aslam_detector_params;
void aslam_default_detector_params(aslam_detector_params* p);
int  aslam_set_detector_params(aslam_ctx* ctx, const aslam_detector_params* p);

/* dictionary_ = cv::aruco::getPredefinedDictionary(markers_dictionary) — aruco_slam.cpp:11-12, aruco_slam.h:176.  Only
 * DICT_ARUCO_ORIGINAL (16, the shipped parameters.yaml value) can be generated without OpenCV's tables, so every other
 * dictionary is handed over as data after aslam_create: either bits[n_markers][marker_size^2] (row-major, 1 = white, what
 * Dictionary::getBitsFromByteList returns) or OpenCV's own Dictionary::bytesList buffer (n rows x ceil(ms^2/8) columns x
 * 4 channels, e.g. dictionary->bytesList.data) with dictionary->markerSize / maxCorrectionBits.  marker_size 3..7. */
int aslam_set_dictionary(aslam_ctx* ctx, int marker_size, int n_markers, int max_correction_bits, const uint8_t* bits);
int aslam_set_dictionary_bytes(aslam_ctx* ctx, int marker_size, int n_markers, int max_correction_bits,
                               const uint8_t* bytes_list);

/* ArucoSlam::addEncoder(wl, wr) — aruco_slam.h:116, aruco_slam.cpp:21-74.  t_now_sec stands in for the two
 * ros::Time::now() calls (:26,:31-32); the adapter passes ros::Time::now().toSec(). */
int aslam_add_encoder(aslam_ctx* ctx, double wl, double wr, double t_now_sec);

/* ArucoSlam::addImage(const cv::Mat&) — aruco_slam.h:122, aruco_slam.cpp:76-287.  px is borrowed for the
 * call only (cv_bridge::toCvShare aliasing, aruco_slam_node.cpp:93); channels 1 (gray) or 3 (bgr8). */
int aslam_add_image(aslam_ctx* ctx, const uint8_t* px, int rows, int cols, int channels, size_t step_bytes);
/* host-clock breakdown of the last aslam_add_image in microseconds: upload of the borrowed pixels, enqueueing detection + pose,
 * enqueueing the EKF step, waiting for the device, read-back of the overflow flags, total (instrumentation; no reference counterpart) */
int aslam_get_last_timing(aslam_ctx* ctx, double out[6]);

/* mu_ / sigma_ (aruco_slam.h:182-183).  sigma is written column-major with leading dimension N, exactly
 * Eigen::MatrixXd's layout.  Pass NULL for mu/sigma to query N only. */
int aslam_get_state(aslam_ctx* ctx, int* N, double* mu, double* sigma);
/* overwrite the state (no reference counterpart: used by tests and warm starts) */
int aslam_set_state(aslam_ctx* ctx, int N, const double* mu, const double* sigma, const int* landmark_ids);

/* marker_corners / IDs / rvs / tvs of getObservations (aruco_slam.cpp:309-314) for the last frame of the last
 * call; corners M*8 floats (x0,y0,...,x3,y3), rvecs/tvecs M*3 doubles.  Arrays may be NULL. */
int aslam_get_detections(aslam_ctx* ctx, int* M, int* ids, float* corners, double* rvecs, double* tvecs);
/* the observations popped from obs_ in pop order (aruco_slam.cpp:92-95) for the last frame: id, landmark index
 * at push time (-1 new), action (0 augment, 1 update, 2 stationary no-op), (x,y,theta), diag(R). */
int aslam_get_observations(aslam_ctx* ctx, int* n, int* ids, int* idx, int* action, double* xyth, double* Rdiag);
/* aruco_id_map inverted: ids[i] = marker id of landmark index i (aruco_slam.h:164) */
int aslam_get_landmark_ids(aslam_ctx* ctx, int* L, int* ids);

/* ---- device-resident stream API (throughput path: frames staged once in HBM) -------------------------
 * aslam_stage_frames uploads nframes tightly packed frames into slots [slot0, slot0+nframes);
 * aslam_stage_encoders stores, per slot, the encoder sample (wl, wr, dt) that precedes that frame;
 * aslam_run_staged(first, count, with_ekf) then runs (with_ekf: 0 detection+pose only, 1 full path, 2 EKF steps only), entirely on the device and asynchronously on the context's
 * stream: detection + pose for all `count` frames batched, followed (with_ekf != 0) by `count` sequential
 * addEncoder(dt) + addImage EKF steps.  aslam_sync waits and reports device-side overflow.
 * Re-staging rule: aslam_run_staged returns before the batch has run (its EKF work is even enqueued one call
 * later, once its observations have reached the host).  aslam_stage_frames / aslam_stage_encoders on slots a
 * submitted batch still reads first finish that batch's use of them (they enqueue the deferred EKF work and
 * wait for it), so the new samples can never reach the old batch; slots outside the range are not waited for. */
int aslam_stage_frames(aslam_ctx* ctx, int slot0, const uint8_t* frames, int nframes, int rows, int cols,
                       int channels, size_t step_bytes, size_t frame_stride_bytes);
int aslam_stage_encoders(aslam_ctx* ctx, int slot0, int n, const double* wl, const double* wr, const double* dt);
int aslam_run_staged(aslam_ctx* ctx, int first, int count, int with_ekf);
int aslam_sync(aslam_ctx* ctx);
/* per-slot results of the last aslam_run_staged */
int aslam_get_slot_detections(aslam_ctx* ctx, int slot, int* M, int* ids, float* corners, double* rvecs, double* tvecs);
int aslam_get_slot_raw_observations(aslam_ctx* ctx, int slot, int* n, int* ids, int* valid, double* xyth, double* Rdiag);
/* what the EKF step of each slot in [first, first + count) did, 4 ints per slot: markers detected, landmarks appended
 * (aruco_slam.cpp:208-260), corrections fused (:108-207) and "stationary" no-ops (:192-198).  Lets a caller assert per frame
 * that no observation was lost to the range / covariance gates (aruco_slam.cpp:327-333, :367-368). */
int aslam_get_slot_ekf_stats(aslam_ctx* ctx, int first, int count, int* stats);

/* ---- what the node publishes (aruco_slam_node.cpp:99-118), as plain data for the adapter to wrap in ROS messages ----
 * aslam_get_pose_msg = ArucoSlam::toRosPose (aruco_slam.cpp:378-410): frame "world", z = 0.1, yaw-only quaternion
 * (x, y, z, w) and the 6x6 row-major covariance with sigma_(0..2, 0..2) scattered to rows/columns 0, 1, 5.
 * aslam_get_map_markers = detected_map_ (aruco_slam.cpp:265-281): frame "world", CUBE (L, L, 0.01), rgba (1, .5, 1, .5),
 * z = 0.3, setRPY(0, 1.5708, theta), lifetime 0; id = landmark index.
 * aslam_get_detected_markers = detected_markers_ of the last frame (aruco_slam.cpp:325-347): markers inside the range gate,
 * frame "base_link", rgba (1, 0, 0, 1), pose = transformStamped_r2c applied to (rvec, tvec), lifetime 0.1 s.
 * *n receives the number available; at most max are written. */
typedef struct { double position[3]; double orientation[4]; double covariance[36]; } aslam_pose_msg;
typedef struct { int id; int pad; double scale[3]; float color[4]; double position[3]; double orientation[4]; double lifetime_sec; } aslam_marker_msg;
int aslam_get_pose_msg(aslam_ctx* ctx, aslam_pose_msg* out);
int aslam_get_map_markers(aslam_ctx* ctx, int max, int* n, aslam_marker_msg* out);
int aslam_get_detected_markers(aslam_ctx* ctx, int max, int* n, aslam_marker_msg* out);

/* markered_img_ = img.clone(); cv::aruco::drawDetectedMarkers(markered_img_, marker_corners, IDs) (aruco_slam.cpp:318-319, returned
 * by getMarkedImg(), aruco_slam.h:152): draws the last frame's detections into the caller's bgr8 buffer - quad outline (0,255,0),
 * 7x7 square outline (0,0,255) around corner 0.  Host code; the "id=N" label of the original is not rendered. */
int aslam_draw_detected_markers(aslam_ctx* ctx, uint8_t* bgr, int rows, int cols, size_t step_bytes);

/* MapLoader::loadMap (map_loader.cpp:7-118): the ground-truth map file "id length x y [z [roll [pitch [yaw]]]]" behind the
 * latched real_map topic, parsed with the loader's own rules ('#' comments, malformed line => empty map, short line skipped,
 * the crossed roll / yaw fallbacks) into marker messages (frame "world", rgba (1,1,1,.5)).  Host only; ctx may be NULL. */
int aslam_load_map_txt(aslam_ctx* ctx, const char* path, int max, int* n, aslam_marker_msg* out);

/* filter state (mu, sigma, landmark ids, armed flag) to / from a file; no counterpart in the reference (warm starts) */
int aslam_save_state(aslam_ctx* ctx, const char* path);
int aslam_load_state(aslam_ctx* ctx, const char* path);

/* ---- host-fed stream: pinned ring + asynchronous upload (the input step before the path, aruco_slam_node.cpp:85-96) ----
 * aslam_stream_open page-locks a ring of two half batches of frames_per_submit frames (<= max_batch / 2).
 * aslam_stream_push copies one frame and the encoder sample (wl, wr, dt) that precedes it into the ring (px is borrowed
 * for the call only); aslam_stream_acquire / aslam_stream_commit hand the pinned slot to the producer instead (no host
 * copy).  Every frames_per_submit frames the half is uploaded on a copy stream and its detection + EKF steps are
 * enqueued, so the upload of one half overlaps the processing of the other; the calls only block when the ring is full.
 * aslam_stream_flush submits what is pending, waits and reports device-side overflow; then the getters are valid. */
int aslam_stream_open(aslam_ctx* ctx, int rows, int cols, int channels, int frames_per_submit);
int aslam_stream_push(aslam_ctx* ctx, const uint8_t* px, size_t step_bytes, double wl, double wr, double dt);
int aslam_stream_acquire(aslam_ctx* ctx, uint8_t** px, size_t* step_bytes);
int aslam_stream_commit(aslam_ctx* ctx, double wl, double wr, double dt);
int aslam_stream_flush(aslam_ctx* ctx);

/* cv::aruco::detectMarkers + estimatePoseSingleMarkers on a batch of independent host frames, no EKF
 * (BASELINE config 5).  counts[nframes]; ids/corners/rvecs/tvecs hold max_per_frame entries per frame. */
int aslam_detect_batch(aslam_ctx* ctx, const uint8_t* frames, int nframes, int rows, int cols, int channels,
                       size_t step_bytes, size_t frame_stride_bytes, int max_per_frame, int* counts, int* ids,
                       float* corners, double* rvecs, double* tvecs);

/* ---- landmark-map record for the multi-GPU gather ---------------------------------------
 * Fixed-size record per landmark: { int32 id, int32 index, f64 x, y, theta, f64 Sigma_ll[9] } = 104 bytes.
 * Writes max_landmarks records (unused ones have id = -1) to a host or device buffer. */
int aslam_export_map(aslam_ctx* ctx, void* dst, int dst_is_device);
/* the same without stalling the pipeline: the export is enqueued behind the EKF steps submitted so far and lands in the
 * device buffer d_dst; aslam_export_wait(buffer) blocks until that particular export (buffer 0 or 1) is complete, so a caller
 * can gather step k-1's map while step k is already running on the GPU */
int aslam_export_map_async(aslam_ctx* ctx, void* d_dst, int buffer);
int aslam_export_wait(aslam_ctx* ctx, int buffer);
#define ASLAM_MAP_RECORD_BYTES 104

/* The gather itself for a C / C++ node (one process and one context per GPU): RCCL's all-gather over xGMI straight from the
 * device buffers, librccl.so dlopen'ed on first use (no link-time dependency; a process that already uses RCCL - e.g. through
 * torch - shares its copy).  Rank 0 calls aslam_comm_get_unique_id and hands the 128 bytes to the other ranks by any
 * out-of-band means (ROS parameter, file, socket); every rank then calls aslam_comm_create(ctx, id, world, rank).
 * aslam_comm_gather_maps exports this rank's map behind the EKF steps enqueued so far, all-gathers, and writes
 * world x max_landmarks records (rank-major) to dst (host, or device if dst_is_device).  Read-only: nothing is fused back. */
#define ASLAM_COMM_ID_BYTES 128
int aslam_comm_get_unique_id(void* id /* ASLAM_COMM_ID_BYTES */);
int aslam_comm_create(aslam_ctx* ctx, const void* id, int world, int rank);
int aslam_comm_gather_maps(aslam_ctx* ctx, void* dst, int dst_is_device);
int aslam_comm_destroy(aslam_ctx* ctx);

/* ---- instrumentation ---------------------------------------------------------------------------------
 * Stage taps used by the parity tests (tests/): what each detector stage produced for a staged slot. */
int aslam_debug_get_nbr(aslam_ctx* ctx, int slot, int scale, uint8_t* out /* rows*cols */);
/* list sizes of one slot after its last detection pass: border nodes, kept contours, contour points, write tickets, quad candidates, and
 * whether the frame's node cycles went through the serial fallback kernel (more nodes / borders than the LDS image holds) */
int aslam_debug_get_frame_counts(aslam_ctx* ctx, int slot, unsigned out[6]);
int aslam_debug_get_contours(aslam_ctx* ctx, int slot, int scale, int max_contours, long long max_points,
                             int* n_contours, int* sizes, int* keys, int* points_xy, long long* n_points);
int aslam_debug_get_candidates(aslam_ctx* ctx, int slot, int stage /*0 quads (unordered), 2 final*/, int max,
                               int* n, float* corners, int* sizes, int* ids);
/* overwrite a slot's per-marker observations (id, passed-the-gates flag, (x,y,theta), diag R); together with
 * aslam_run_staged(..., with_ekf = 2) = "EKF steps only" this replays recorded observation sequences through the
 * device EKF without the detector (tests/test_ekf_golden.py). */
int aslam_debug_inject_observations(aslam_ctx* ctx, int slot, int n, const int* ids, const int* valid, const double* xyth,
                                    const double* Rdiag);
/* HIP-event timing of each kernel family on the context's stream, accumulated since the last reset:
 * names[i] (static strings), calls[i], total_ms[i]; returns the number of entries. */
int aslam_profile_enable(aslam_ctx* ctx, int on);
int aslam_profile_reset(aslam_ctx* ctx);
int aslam_profile_get(aslam_ctx* ctx, int max, const char** names, int* calls, double* total_ms);
/* how the frames submitted with an EKF step since the last aslam_profile_reset were scheduled: out[0] frames fused inside
 * windows (ekf_window.hip), out[1] frames on the per-frame chain, out[2] windows formed, out[3] frames whose bookkeeping was left
 * to the device (a new landmark, one id twice) - the windowed path is an optimisation, never a different result. */
int aslam_get_plan_stats(aslam_ctx* ctx, long long out[4]);

/* deterministic synthetic frame renderer (input generation for tests and bench; not on the hot path).
 * markers: per marker 12 doubles = rotation (row-major 3x3, marker->camera) then translation;
 * ids: dictionary id per marker.  Writes gray frames (rows*cols) into staged slots (on_device=1) or to out. */
int aslam_synth_render(aslam_ctx* ctx, int slot, int rows, int cols, const double K[9], int n_markers, const int* ids,
                       const double* poses, double marker_length, int background, int noise_amp, unsigned seed,
                       int supersample, uint8_t* out_host /* may be NULL */);

#ifdef __cplusplus
}
#endif
#endif
