// ORACLE (test infrastructure only — see oracle.h).  Plain C entry points so that tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg can drive the CPU restatement via ctypes.
#include "oracle.h"
#include <cstring>
#include <cmath>
#include <algorithm>

using namespace oracle;

namespace {
struct SlamHandle {
    Slam slam;
    FrameLog log;
    explicit SlamHandle(const SlamParams& p) : slam(p) {}
};
Camera make_cam(const double K[9], const double* D, int nD) {
    Camera c;
    std::memcpy(c.K, K, sizeof(c.K));
    c.nD = std::min(nD, 5);
    for (int i = 0; i < 5; i++) c.D[i] = (i < c.nD && D) ? D[i] : 0.0;
    return c;
}
Dictionary& dict() { static Dictionary d = make_dict_aruco_original(); return d; }
DetectorParams& params() { static DetectorParams p; return p; }
// 20 values in the order of struct DetectorParams (oracle.h); NULL = the OpenCV 3.2.0 defaults
DetectorParams params_from(const double* v) {
    DetectorParams p;
    if (!v) return p;
    p.adaptiveThreshWinSizeMin = (int)v[0]; p.adaptiveThreshWinSizeMax = (int)v[1]; p.adaptiveThreshWinSizeStep = (int)v[2];
    p.adaptiveThreshConstant = v[3]; p.minMarkerPerimeterRate = v[4]; p.maxMarkerPerimeterRate = v[5];
    p.polygonalApproxAccuracyRate = v[6]; p.minCornerDistanceRate = v[7]; p.minDistanceToBorder = (int)v[8];
    p.minMarkerDistanceRate = v[9]; p.markerBorderBits = (int)v[10]; p.perspectiveRemovePixelPerCell = (int)v[11];
    p.perspectiveRemoveIgnoredMarginPerCell = v[12]; p.maxErroneousBitsInBorderRate = v[13]; p.minOtsuStdDev = v[14];
    p.errorCorrectionRate = v[15];
    p.doCornerRefinement = v[16] != 0; p.cornerRefinementWinSize = (int)v[17]; p.cornerRefinementMaxIterations = (int)v[18];
    p.cornerRefinementMinAccuracy = v[19];
    return p;
}
}

extern "C" {

double orc_norm_angle(double a) { norm_angle(a); return a; }

void orc_bgr2gray(const uint8_t* bgr, int rows, int cols, size_t step, uint8_t* gray) { bgr_to_gray(bgr, rows, cols, step, gray); }

void orc_box_mean(const uint8_t* gray, int rows, int cols, int k, uint8_t* out) { box_mean_u8(gray, rows, cols, k, out); }

void orc_threshold(const uint8_t* gray, int rows, int cols, int k, double C, uint8_t* out) {
    adaptive_threshold_mean_inv(gray, rows, cols, k, C, out);
}

// contours in OpenCV output order (reverse discovery).  points_xy holds 2 ints per point.
int orc_find_contours(const uint8_t* bin, int rows, int cols, int max_contours, long long max_points,
                      int* sizes, int* keys, int* is_hole, int* points_xy, long long* total_points) {
    std::vector<Contour> cs;
    find_contours_list_none(bin, rows, cols, cs);
    long long tot = 0;
    int n = 0;
    for (const Contour& c : cs) {
        if (n >= max_contours) return -1;
        if (tot + (long long)c.pts.size() > max_points) return -2;
        sizes[n] = (int)c.pts.size(); keys[n] = c.key; is_hole[n] = c.is_hole;
        for (const Pt& p : c.pts) { points_xy[2 * tot] = p.x; points_xy[2 * tot + 1] = p.y; tot++; }
        n++;
    }
    *total_points = tot;
    return n;
}

int orc_approx_poly(const int* pts_xy, int n, double eps, int* out_xy, int max_out) {
    std::vector<Pt> src(n), dst;
    for (int i = 0; i < n; i++) src[i] = Pt{pts_xy[2 * i], pts_xy[2 * i + 1]};
    approx_poly_dp_closed(src, eps, dst);
    for (int i = 0; i < (int)dst.size() && i < max_out; i++) { out_xy[2 * i] = dst[i].x; out_xy[2 * i + 1] = dst[i].y; }
    return (int)dst.size();
}

static int export_candidates(const std::vector<Candidate>& c, int max, float* corners, int* sizes, int* scales, int* keys) {
    int n = (int)std::min<size_t>(c.size(), (size_t)max);
    for (int i = 0; i < n; i++) {
        for (int k = 0; k < 4; k++) { corners[8 * i + 2 * k] = c[i].c[k].x; corners[8 * i + 2 * k + 1] = c[i].c[k].y; }
        sizes[i] = c[i].contour_size; scales[i] = c[i].scale; keys[i] = c[i].key;
    }
    return (int)c.size();
}

// stage = 0: _detectInitialCandidates; 1: + _reorderCandidatesCorners; 2: + _filterTooCloseCandidates
int orc_candidates(const uint8_t* gray, int rows, int cols, int stage, int max, float* corners, int* sizes, int* scales, int* keys) {
    const DetectorParams P = params();
    std::vector<Candidate> a, b;
    detect_initial_candidates(gray, rows, cols, P, a);
    if (stage >= 1) reorder_candidate_corners(a);
    if (stage >= 2) { filter_too_close_candidates(a, b, P.minMarkerDistanceRate); a.swap(b); }
    return export_candidates(a, max, corners, sizes, scales, keys);
}

void orc_perspective_transform(const float src[8], const float dst[8], double M[9]) {
    Pt2f s[4], d[4];
    for (int i = 0; i < 4; i++) { s[i] = Pt2f{src[2 * i], src[2 * i + 1]}; d[i] = Pt2f{dst[2 * i], dst[2 * i + 1]}; }
    get_perspective_transform(s, d, M);
}

void orc_extract_bits(const uint8_t* gray, int rows, int cols, const float corners[8], uint8_t* bits /*49*/) {
    const DetectorParams P = params();
    Pt2f c[4];
    for (int i = 0; i < 4; i++) c[i] = Pt2f{corners[2 * i], corners[2 * i + 1]};
    std::vector<uint8_t> b;
    extract_bits(gray, rows, cols, c, dict().markerSize, P, b);
    std::memcpy(bits, b.data(), b.size());
}

int orc_identify(const uint8_t* gray, int rows, int cols, float corners[8], int* id) {
    const DetectorParams P = params();
    Pt2f c[4];
    for (int i = 0; i < 4; i++) c[i] = Pt2f{corners[2 * i], corners[2 * i + 1]};
    bool ok = identify_one_candidate(dict(), gray, rows, cols, c, *id, P);
    for (int i = 0; i < 4; i++) { corners[2 * i] = c[i].x; corners[2 * i + 1] = c[i].y; }
    return ok ? 1 : 0;
}

// replace the dictionary used by the free functions below (Slam objects carry their own, see orc_slam_set_dictionary)
void orc_set_dictionary(int ms, int n, int maxcorr, const uint8_t* bits) { dict() = bits ? make_dict_from_bits(ms, n, maxcorr, bits) : make_dict_aruco_original(); }

void orc_dict_bytes(uint8_t* out /*1024*16*/) { std::memcpy(out, dict().bytesList.data(), dict().bytesList.size()); }

// marker bits (5x5, 1 = white) of DICT_ARUCO_ORIGINAL id
void orc_dict_bits(int id, uint8_t* bits25) {
    static const int words[4] = {0x10, 0x17, 0x09, 0x0e};
    for (int y = 0; y < 5; y++) {
        int val = words[(id >> (2 * (4 - y))) & 3];
        for (int x = 0; x < 5; x++) bits25[y * 5 + x] = (val >> (4 - x)) & 1;
    }
}

int orc_detect(const uint8_t* img, int rows, int cols, int channels, size_t step, int max, int* ids, float* corners) {
    const DetectorParams P = params();
    std::vector<Detection> det;
    detect_markers(img, rows, cols, channels, step, dict(), P, det);
    int n = (int)std::min<size_t>(det.size(), (size_t)max);
    for (int i = 0; i < n; i++) {
        ids[i] = det[i].id;
        for (int k = 0; k < 4; k++) { corners[8 * i + 2 * k] = det[i].c[k].x; corners[8 * i + 2 * k + 1] = det[i].c[k].y; }
    }
    return (int)det.size();
}

void orc_rodrigues(const double r[3], double R[9], double* J27) { rodrigues_vec_to_mat(r, R, J27); }
void orc_rodrigues_inv(const double R[9], double r[3]) { rodrigues_mat_to_vec(R, r); }

void orc_project_points(const double* obj3, int n, const double r[3], const double t[3], const double K[9],
                        const double* D, int nD, double* out2, double* dpdr, double* dpdt) {
    Camera c = make_cam(K, D, nD);
    project_points(reinterpret_cast<const double(*)[3]>(obj3), n, r, t, c, reinterpret_cast<double(*)[2]>(out2), dpdr, dpdt);
}

void orc_solve_pnp(const float corners[8], float L, const double K[9], const double* D, int nD, double rvec[3],
                   double tvec[3], int* iters) {
    Camera c = make_cam(K, D, nD);
    Pt2f cc[4];
    for (int i = 0; i < 4; i++) cc[i] = Pt2f{corners[2 * i], corners[2 * i + 1]};
    solve_pnp_marker(cc, L, c, rvec, tvec, iters);
}

// pop order of std::priority_queue<ArucoMarker> for the given push sequence of aruco_index_ values
void orc_heap_order(int n, const int* indices, int* pop_order) {
    std::priority_queue<Observation> q;
    for (int i = 0; i < n; i++) { Observation o{}; o.id = i; o.index = indices[i]; q.push(o); }
    for (int i = 0; i < n; i++) { pop_order[i] = q.top().id; q.pop(); }
}

// params: Q_k,R_x,R_y,R_theta,kl,kr,b,marker_length,r2c_tx,r2c_ty
void* orc_slam_create(const double params[10], float useful_distance_threshold, int literal) {
    SlamParams p;
    p.Q_k = params[0]; p.R_x = params[1]; p.R_y = params[2]; p.R_theta = params[3];
    p.kl = params[4]; p.kr = params[5]; p.b = params[6]; p.marker_length = params[7];
    p.r2c_tx = params[8]; p.r2c_ty = params[9];
    p.useful_distance_threshold = useful_distance_threshold;
    SlamHandle* h = new SlamHandle(p);
    h->slam.literal = literal != 0;
    return h;
}
void orc_slam_destroy(void* h) { delete static_cast<SlamHandle*>(h); }
void orc_slam_set_camera(void* h, const double K[9], const double* D, int nD) { static_cast<SlamHandle*>(h)->slam.setCamera(make_cam(K, D, nD)); }
void orc_corner_sub_pix(const uint8_t* gray, int rows, int cols, float* corners_xy, int count, int win, int max_iter, double eps) {
    std::vector<Pt2f> c(count);
    for (int i = 0; i < count; i++) c[i] = Pt2f{corners_xy[2 * i], corners_xy[2 * i + 1]};
    corner_sub_pix(gray, rows, cols, c.data(), count, win, max_iter, eps);
    for (int i = 0; i < count; i++) { corners_xy[2 * i] = c[i].x; corners_xy[2 * i + 1] = c[i].y; }
}
void orc_set_detector_params(const double* v) { params() = params_from(v); }
void orc_slam_set_detector_params(void* h, const double* v) { static_cast<SlamHandle*>(h)->slam.dp = params_from(v); }
void orc_slam_set_dictionary(void* h, int ms, int n, int maxcorr, const uint8_t* bits) {
    static_cast<SlamHandle*>(h)->slam.dict = make_dict_from_bits(ms, n, maxcorr, bits);
}
void orc_slam_add_encoder(void* h, double wl, double wr, double t) { static_cast<SlamHandle*>(h)->slam.addEncoder(wl, wr, t); }
void orc_slam_add_image(void* h, const uint8_t* img, int rows, int cols, int channels, size_t step) {
    SlamHandle* s = static_cast<SlamHandle*>(h);
    s->log = FrameLog();
    s->slam.addImage(img, rows, cols, channels, step, &s->log);
}
void orc_slam_add_poses(void* h, int n, const int* ids, const float* corners, const double* rvecs, const double* tvecs) {
    SlamHandle* s = static_cast<SlamHandle*>(h);
    std::vector<Detection> det(n);
    for (int i = 0; i < n; i++) {
        det[i].id = ids[i];
        for (int k = 0; k < 4; k++) det[i].c[k] = Pt2f{corners[8 * i + 2 * k], corners[8 * i + 2 * k + 1]};
    }
    std::vector<double> rv(rvecs, rvecs + 3 * n), tv(tvecs, tvecs + 3 * n);
    s->log = FrameLog();
    s->log.detections = det; s->log.rvecs = rv; s->log.tvecs = tv;
    s->slam.addObservationsFromPoses(det, rv, tv, &s->log);
}
int orc_slam_state_size(void* h) { return (int)static_cast<SlamHandle*>(h)->slam.mu.size(); }
void orc_slam_get_state(void* h, double* mu, double* sigma) {
    Slam& s = static_cast<SlamHandle*>(h)->slam;
    std::memcpy(mu, s.mu.data(), s.mu.size() * sizeof(double));
    std::memcpy(sigma, s.sigma.data(), s.sigma.size() * sizeof(double));
}
// overwrite the state (to start a test from a given mu/Sigma/map)
void orc_slam_set_state(void* h, int N, const double* mu, const double* sigma, const int* landmark_ids) {
    Slam& s = static_cast<SlamHandle*>(h)->slam;
    s.mu.assign(mu, mu + N);
    s.sigma.assign(sigma, sigma + (size_t)N * N);
    s.aruco_id_map.clear();
    for (int i = 0; i < (N - 3) / 3; i++) s.aruco_id_map.insert({landmark_ids[i], i});
    s.last_observed.clear();
}
int orc_slam_landmark_ids(void* h, int* ids) {
    Slam& s = static_cast<SlamHandle*>(h)->slam;
    int L = ((int)s.mu.size() - 3) / 3;
    for (int i = 0; i < L; i++) ids[i] = -1;
    // index -> id (first insertion wins; duplicates leave later indices unnamed = -1)
    for (auto& kv : s.aruco_id_map) if (kv.second < L) ids[kv.second] = kv.first;
    return L;
}
int orc_slam_log_detections(void* h, int max, int* ids, float* corners, double* rvecs, double* tvecs) {
    FrameLog& l = static_cast<SlamHandle*>(h)->log;
    int n = (int)std::min<size_t>(l.detections.size(), (size_t)max);
    for (int i = 0; i < n; i++) {
        ids[i] = l.detections[i].id;
        for (int k = 0; k < 4; k++) { corners[8 * i + 2 * k] = l.detections[i].c[k].x; corners[8 * i + 2 * k + 1] = l.detections[i].c[k].y; }
        for (int k = 0; k < 3; k++) { rvecs[3 * i + k] = l.rvecs[3 * i + k]; tvecs[3 * i + k] = l.tvecs[3 * i + k]; }
    }
    return (int)l.detections.size();
}
int orc_slam_log_observations(void* h, int max, int* ids, int* idx, int* action, double* xyth, double* R) {
    FrameLog& l = static_cast<SlamHandle*>(h)->log;
    int n = (int)std::min<size_t>(l.popped.size(), (size_t)max);
    for (int i = 0; i < n; i++) {
        ids[i] = l.popped[i].id; idx[i] = l.popped[i].index; action[i] = l.action[i];
        xyth[3 * i] = l.popped[i].x; xyth[3 * i + 1] = l.popped[i].y; xyth[3 * i + 2] = l.popped[i].theta;
        std::memcpy(&R[9 * i], l.popped[i].R, 9 * sizeof(double));
    }
    return (int)l.popped.size();
}

} // extern "C"
