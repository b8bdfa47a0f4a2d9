// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement (plain sequential C++17, no third-party code) of the hot path of
// gitAugust/Aruco_Slam:  detectMarkers -> estimatePoseSingleMarkers -> observation model ->
// SE(2) EKF predict / update / augment  (reference src/aruco_slam.cpp:21-74, 76-263, 307-376,
// 412-471; types include/aruco_slam/aruco_slam.h:40-94,164-191).
//
// PARITY STATUS: **parity unpinned**.  The reference ships no tests, no golden vectors and no
// fixtures, and it cannot be compiled here (it needs ROS, OpenCV 3.x + contrib
// and Eigen, none of which exist in this image).  The EKF / observation part is a literal
// restatement of the reference's own source; the detector / PnP part restates the *published
// algorithm* of the third-party code the reference calls (OpenCV 3.2.0 core/imgproc/calib3d +
// opencv_contrib aruco, the de-facto pin via ROS melodic, .vscode/c_cpp_properties.json:10) from
// its call sites aruco_slam.cpp:11,313,314,354,441.  The restatement is pinned only by hand-derived
// known-answer tests (tests/test_oracle_*.py) and by a second, independently written numpy
// transcription of the EKF (oracle/ekf_literal.py).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may include, link or call
// anything in this directory.  The product (aruco_slam_amd/) never does.
#pragma once
#include <cstdint>
#include <cstddef>
#include <vector>
#include <map>
#include <queue>

namespace oracle {

// ----------------------------------------------------------------------------------------------
// Detector parameters: cv::aruco::DetectorParameters defaults of OpenCV 3.2.0 (the reference calls
// detectMarkers with the library defaults, aruco_slam.cpp:313).
struct DetectorParams {
    int    adaptiveThreshWinSizeMin   = 3;
    int    adaptiveThreshWinSizeMax   = 23;
    int    adaptiveThreshWinSizeStep  = 10;
    double adaptiveThreshConstant     = 7;
    double minMarkerPerimeterRate     = 0.03;
    double maxMarkerPerimeterRate     = 4.0;
    double polygonalApproxAccuracyRate = 0.05;   // 3.2.0 (0.03 from 3.3)
    double minCornerDistanceRate      = 0.05;
    int    minDistanceToBorder        = 3;
    double minMarkerDistanceRate      = 0.05;
    int    markerBorderBits           = 1;
    int    perspectiveRemovePixelPerCell = 8;    // 3.2.0 (4 from 3.3)
    double perspectiveRemoveIgnoredMarginPerCell = 0.13;
    double maxErroneousBitsInBorderRate = 0.35;
    double minOtsuStdDev              = 5.0;
    double errorCorrectionRate        = 0.6;
    bool   doCornerRefinement         = false;  // off in the reference (default parameters)
    int    cornerRefinementWinSize    = 5;
    int    cornerRefinementMaxIterations = 30;
    double cornerRefinementMinAccuracy = 0.1;
};

struct Pt  { int x, y; };
struct Pt2f { float x, y; };

struct Contour {
    std::vector<Pt> pts;
    int  is_hole;     // 1 if found as a hole border
    int  key;         // raster position (y*cols + x) of the scan pixel at which it was discovered
};

struct Candidate {
    Pt2f c[4];
    int  contour_size;
    int  scale;       // index of the adaptive-threshold window (0,1,2)
    int  key;         // discovery key of its contour
};

struct Dictionary {
    int markerSize = 0;
    int maxCorrectionBits = 0;
    int nMarkers = 0;
    int nbytes = 0;                       // bytes per rotation
    std::vector<uint8_t> bytesList;       // nMarkers * 4 rotations * nbytes
};

struct Detection {
    int  id;
    Pt2f c[4];
};

// -- image stages -------------------------------------------------------------------------------
void bgr_to_gray(const uint8_t* bgr, int rows, int cols, size_t step, uint8_t* gray);
void box_mean_u8(const uint8_t* gray, int rows, int cols, int k, uint8_t* mean);
void adaptive_threshold_mean_inv(const uint8_t* gray, int rows, int cols, int k, double C, uint8_t* out);
void find_contours_list_none(const uint8_t* bin, int rows, int cols, std::vector<Contour>& out);
void approx_poly_dp_closed(const std::vector<Pt>& src, double eps, std::vector<Pt>& dst);
bool is_contour_convex(const std::vector<Pt>& p);
void find_marker_contours(const uint8_t* thresh, int rows, int cols, const DetectorParams& p, int scale,
                          std::vector<Candidate>& out);
void detect_initial_candidates(const uint8_t* gray, int rows, int cols, const DetectorParams& p,
                               std::vector<Candidate>& out);
void reorder_candidate_corners(std::vector<Candidate>& c);
void filter_too_close_candidates(const std::vector<Candidate>& in, std::vector<Candidate>& out, double rate);
void get_perspective_transform(const Pt2f src[4], const Pt2f dst[4], double M[9]);
void warp_perspective_nearest(const uint8_t* gray, int rows, int cols, const double M[9], int dsize, uint8_t* dst);
int  otsu_threshold(const uint8_t* img, int n);
// returns false if the candidate bits could not be extracted; bits is (markerSize+2b)^2 row-major 0/1
void extract_bits(const uint8_t* gray, int rows, int cols, const Pt2f corners[4], int markerSize,
                  const DetectorParams& p, std::vector<uint8_t>& bits);
Dictionary make_dict_aruco_original();
void get_rect_sub_pix_8u32f(const uint8_t* src, int rows, int cols, int win_w, int win_h, float cx, float cy, float* dst);
void corner_sub_pix(const uint8_t* gray, int rows, int cols, Pt2f* corners, int count, int win, int maxCount, double epsilon);
Dictionary make_dict_from_bits(int markerSize, int nMarkers, int maxCorrectionBits, const uint8_t* bits);
bool dictionary_identify(const Dictionary& d, const uint8_t* onlyBits, int& idx, int& rotation, double rate);
bool identify_one_candidate(const Dictionary& d, const uint8_t* gray, int rows, int cols, Pt2f corners[4],
                            int& id, const DetectorParams& p);
void filter_detected_markers(std::vector<Detection>& d);
// full cv::aruco::detectMarkers (channels = 1 gray or 3 BGR)
void detect_markers(const uint8_t* img, int rows, int cols, int channels, size_t step, const Dictionary& dict,
                    const DetectorParams& p, std::vector<Detection>& out,
                    std::vector<Candidate>* cand_after_filter = nullptr);

// -- pose ---------------------------------------------------------------------------------------
struct Camera { double K[9]; double D[5]; int nD; };
void rodrigues_vec_to_mat(const double r[3], double R[9], double J[27] /*may be null: dR/dr, 3x9*/);
void rodrigues_mat_to_vec(const double R[9], double r[3]);
void project_points(const double obj[][3], int n, const double r[3], const double t[3], const Camera& cam,
                    double out[][2], double* dpdr /*2n x 3 or null*/, double* dpdt /*2n x 3 or null*/);
void undistort_points(const double in[][2], int n, const Camera& cam, double out[][2]);
bool find_homography4(const float src[4][2], const float dst[4][2], double H[9]);
// cv::solvePnP(SOLVEPNP_ITERATIVE) on the 4 coplanar marker corners
void solve_pnp_marker(const Pt2f corners[4], float markerLength, const Camera& cam, double rvec[3], double tvec[3],
                      int* iters_out = nullptr);

// -- SLAM ---------------------------------------------------------------------------------------
struct SlamParams {
    double Q_k = 0.01, R_x = 100, R_y = 100, R_theta = 10;
    double kl = 0.05, kr = 0.05, b = 0.09;
    double marker_length = 0.27;
    float  useful_distance_threshold = 3.0f;
    double r2c_tx = 0, r2c_ty = 0;
};

struct Observation {      // ArucoMarker, aruco_slam.h:67-94
    int id, index;
    double x, y, theta;
    double R[9];          // observe_covariance_ (row-major; diagonal)
    double last_obs[3];   // last_observation_; NaN = never set (quirk Q2/Q3 pinned as "never matches")
    friend bool operator<(const Observation& a, const Observation& b) { return a.index > b.index; }
};

struct FrameLog {         // what one addImage did (for parity tests)
    std::vector<Detection> detections;
    std::vector<double>    rvecs, tvecs;           // 3 per detection
    std::vector<Observation> popped;               // in pop order
    std::vector<int>       action;                 // per popped obs: 0 augment, 1 update, 2 stationary no-op
};

class Slam {
public:
    explicit Slam(const SlamParams& p);
    void setCamera(const Camera& c) { cam_ = c; }
    void addEncoder(double wl, double wr, double t_now);
    // dense literal EKF (as the reference executes it) when literal=true, rank-3 form otherwise
    void addImage(const uint8_t* img, int rows, int cols, int channels, size_t step, FrameLog* log = nullptr);
    // EKF part only, from already-computed marker poses (used to isolate stages in tests)
    void addObservationsFromPoses(const std::vector<Detection>& det, const std::vector<double>& rvecs,
                                  const std::vector<double>& tvecs, FrameLog* log = nullptr);
    bool makeObservation(const Detection& d, const double rvec[3], const double tvec[3], Observation& ob) const;

    std::vector<double> mu;        // N
    std::vector<double> sigma;     // N*N column-major (Eigen default), ld = N
    std::map<int,int> aruco_id_map;
    std::vector<Observation> last_observed;
    bool is_init = false;
    double last_time = 0;
    bool literal = true;
    SlamParams P;
    Camera cam_{};
    Dictionary dict;
    DetectorParams dp;
private:
    void processQueue(std::priority_queue<Observation>& obs, FrameLog* log);
};

void norm_angle(double& a);
void calculate_covariance(const SlamParams& P, const Camera& cam, const double tvec[3], const double rvec[3],
                          const Pt2f corners[4], double cov[9]);

} // namespace oracle
