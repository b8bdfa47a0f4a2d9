// ORACLE (test infrastructure only — see oracle.h).  Literal sequential restatement of the
// reference's own EKF-SLAM arithmetic:
//   addEncoder            src/aruco_slam.cpp:21-74
//   addImage              src/aruco_slam.cpp:76-263   (update :108-207, augment :208-260)
//   getObservations loop  src/aruco_slam.cpp:325-374
//   normAngle             src/aruco_slam.cpp:412-421
//   checkLandmark         src/aruco_slam.cpp:423-435
//   CalculateCovariance   src/aruco_slam.cpp:437-471
// including the quirk ledger of SURVEY.md §7 (Q1..Q14).  `literal = true` executes the dense
// O(N^3) products exactly as the reference writes them; `literal = false` uses the algebraically
// identical rank-3 form (the fair algorithmic CPU baseline).
#include "oracle.h"
#include <cmath>
#include <cstring>
#include <algorithm>
#include <limits>

namespace oracle {

void norm_angle(double& angle) {          // aruco_slam.cpp:412-421 — wraps ONCE (Q6)
    const double PI = 3.14159265358979323846;
    const double Two_PI = 2.0 * PI;
    if (angle >= PI) angle -= Two_PI;
    if (angle < -PI) angle += Two_PI;
}

static void inv3_lu(const double A[9], double out[9]) {   // Eigen dynamic .inverse(): partial-pivot LU (Q13)
    double a[3][6];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { a[i][j] = A[i * 3 + j]; a[i][3 + j] = (i == j) ? 1. : 0.; }
    for (int col = 0; col < 3; col++) {
        int piv = col;
        double best = std::fabs(a[col][col]);
        for (int r = col + 1; r < 3; r++)
            if (std::fabs(a[r][col]) > best) { best = std::fabs(a[r][col]); piv = r; }
        if (piv != col)
            for (int c = 0; c < 6; c++) std::swap(a[piv][c], a[col][c]);
        for (int r = col + 1; r < 3; r++) {
            double f = a[r][col] / a[col][col];
            for (int c = col; c < 6; c++) a[r][c] -= f * a[col][c];
        }
    }
    for (int j = 0; j < 3; j++)
        for (int i = 2; i >= 0; i--) {
            double s = a[i][3 + j];
            for (int c = i + 1; c < 3; c++) s -= a[i][c] * out[c * 3 + j];
            out[i * 3 + j] = s / a[i][i];
        }
}

// aruco_slam.cpp:437-471
void calculate_covariance(const SlamParams& P, const Camera& cam, const double tvec[3], const double rvec[3],
                          const Pt2f corners[4], double cov[9]) {
    const float hl = (float)(P.marker_length / 2.f);            // objectPoints_, aruco_slam.h:189
    const double obj[4][3] = {{-hl, hl, 0}, {hl, hl, 0}, {hl, -hl, 0}, {-hl, -hl, 0}};
    double proj[4][2];
    project_points(obj, 4, rvec, tvec, cam, proj, nullptr, nullptr);
    auto dist = [](double x1, double y1, double x2, double y2) {
        double dx = x1 - x2, dy = y1 - y2;
        return std::sqrt(dx * dx + dy * dy);
    };
    double totalError = 0.0;
    for (int i = 0; i < 4; i++) {
        float px = (float)proj[i][0], py = (float)proj[i][1];  // projectedPoints is vector<Point2f>
        double e = dist(corners[i].x, corners[i].y, px, py);
        totalError += e * e;
    }
    double rmserror = totalError / 4.0;
    double nt = std::sqrt(tvec[0] * tvec[0] + tvec[1] * tvec[1] + tvec[2] * tvec[2]);
    double object_error = (rmserror / dist(corners[0].x, corners[0].y, corners[2].x, corners[2].y)) * (nt / P.marker_length);
    std::memset(cov, 0, 9 * sizeof(double));
    cov[0] = object_error * P.R_x + 1e-2;
    cov[4] = object_error * P.R_y + 1e-2;
    cov[8] = object_error * P.R_theta + 1e-3;
}

Slam::Slam(const SlamParams& p) : P(p) {      // aruco_slam.cpp:3-19
    mu.assign(3, 0.0);
    sigma.assign(9, 0.0);
    dict = make_dict_aruco_original();
    for (int i = 0; i < 9; i++) cam_.K[i] = 0;
    cam_.nD = 0;
}

// aruco_slam.cpp:21-74.  `t_now` replaces ros::Time::now() (Q14).
void Slam::addEncoder(double wl, double wr, double t_now) {
    if (!is_init) { last_time = t_now; is_init = true; return; }
    double dt = t_now - last_time;
    last_time = t_now;

    double delta_enl = dt * wl, delta_enr = dt * wr;
    double delta_sl = P.kl * delta_enl, delta_sr = P.kr * delta_enr;
    double l_ = 2 * P.b;
    double delta_theta = (delta_sr - delta_sl) / l_;
    double delta_s = 0.5 * (delta_sr + delta_sl);

    double tmp_th = mu[2] + 0.5 * delta_theta;
    double cos_tmp_th = std::cos(tmp_th), sin_tmp_th = std::sin(tmp_th);
    mu[0] += delta_s * cos_tmp_th;
    mu[1] += delta_s * sin_tmp_th;
    mu[2] += delta_theta;
    norm_angle(mu[2]);

    double H[9] = {1.0, 0.0, -delta_s * sin_tmp_th, 0.0, 1.0, delta_s * cos_tmp_th, 0.0, 0.0, 1.0};
    double wkh[6] = {cos_tmp_th, cos_tmp_th, sin_tmp_th, sin_tmp_th, 1 / P.b, -1 / P.b};   // 3x2 row-major
    for (double& v : wkh) v = (0.5 * P.kl * dt) * v;                                      // kl for both wheels (Q7)
    double su[2] = {P.Q_k * std::fabs(wl), P.Q_k * std::fabs(wr)};
    double Qk[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Qk[i * 3 + j] = wkh[i * 2] * su[0] * wkh[j * 2] + wkh[i * 2 + 1] * su[1] * wkh[j * 2 + 1];

    const int N = (int)mu.size();
    auto S = [&](int r, int c) -> double& { return sigma[(size_t)c * N + r]; };
    if (literal) {
        // sigma_ = Hx * sigma_ * Hx^T + F Qk F^T with a dense N x N Hx (aruco_slam.cpp:64-73)
        std::vector<double> Hx((size_t)N * N, 0.0), T((size_t)N * N, 0.0), O((size_t)N * N, 0.0);
        for (int i = 0; i < N; i++) Hx[(size_t)i * N + i] = 1.0;               // row-major Hx
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Hx[(size_t)i * N + j] = H[i * 3 + j];
        for (int i = 0; i < N; i++)                                            // T = Hx * sigma
            for (int k = 0; k < N; k++) {
                double h = Hx[(size_t)i * N + k];
                if (h == 0.0) continue;                                        // adding an exact zero changes nothing
                for (int j = 0; j < N; j++) T[(size_t)j * N + i] += h * S(k, j);
            }
        for (int j = 0; j < N; j++)                                            // O = T * Hx^T
            for (int k = 0; k < N; k++) {
                double h = Hx[(size_t)j * N + k];
                if (h == 0.0) continue;
                for (int i = 0; i < N; i++) O[(size_t)j * N + i] += T[(size_t)k * N + i] * h;
            }
        sigma.swap(O);
    } else {
        // rows 0..2 <- H * rows 0..2 ; then cols 0..2 <- cols 0..2 * H^T
        for (int j = 0; j < N; j++) {
            double a = S(0, j), b = S(1, j), c = S(2, j);
            for (int i = 0; i < 3; i++) S(i, j) = H[i * 3] * a + H[i * 3 + 1] * b + H[i * 3 + 2] * c;
        }
        for (int i = 0; i < N; i++) {
            double a = S(i, 0), b = S(i, 1), c = S(i, 2);
            for (int j = 0; j < 3; j++) S(i, j) = a * H[j * 3] + b * H[j * 3 + 1] + c * H[j * 3 + 2];
        }
    }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) S(i, j) += Qk[i * 3 + j];
}

// One iteration of the detection loop of getObservations (aruco_slam.cpp:325-374); false = skipped.
bool Slam::makeObservation(const Detection& d, const double rvec[3], const double tvec[3], Observation& ob) const {
    float dist = (float)std::sqrt(tvec[0] * tvec[0] + tvec[1] * tvec[1] + tvec[2] * tvec[2]);
    if (dist > P.useful_distance_threshold) return false;                      // Q11
    double R[9];
    rodrigues_vec_to_mat(rvec, R, nullptr);
    double x = tvec[2] + P.r2c_tx;                                             // Q12
    double y = -tvec[0] + P.r2c_ty;
    double theta = std::atan2(-R[2], R[8]);
    norm_angle(theta);
    double cov[9];
    calculate_covariance(P, cam_, tvec, rvec, d.c, cov);
    double fro = 0;
    for (int i = 0; i < 9; i++) fro += cov[i] * cov[i];
    if (std::sqrt(fro) > 1) return false;
    ob.id = d.id; ob.x = x; ob.y = y; ob.theta = theta;
    std::memcpy(ob.R, cov, sizeof(cov));
    ob.last_obs[0] = ob.last_obs[1] = ob.last_obs[2] = std::numeric_limits<double>::quiet_NaN();   // Q2/Q3
    auto it = aruco_id_map.find(d.id);                                         // checkLandmark
    ob.index = (it != aruco_id_map.end()) ? it->second : -1;
    return true;
}

void Slam::addObservationsFromPoses(const std::vector<Detection>& det, const std::vector<double>& rvecs,
                                    const std::vector<double>& tvecs, FrameLog* log) {
    if (!is_init) return;                                                      // aruco_slam.cpp:84-85
    std::priority_queue<Observation> obs;
    for (size_t i = 0; i < det.size(); i++) {
        Observation ob;
        if (makeObservation(det[i], &rvecs[3 * i], &tvecs[3 * i], ob)) obs.push(ob);
    }
    processQueue(obs, log);
}

void Slam::addImage(const uint8_t* img, int rows, int cols, int channels, size_t step, FrameLog* log) {
    if (!is_init) return;
    std::vector<Detection> det;
    detect_markers(img, rows, cols, channels, step, dict, dp, det);
    std::vector<double> rv(3 * det.size()), tv(3 * det.size());
    for (size_t i = 0; i < det.size(); i++)
        solve_pnp_marker(det[i].c, (float)P.marker_length, cam_, &rv[3 * i], &tv[3 * i]);
    if (log) { log->detections = det; log->rvecs = rv; log->tvecs = tv; }
    addObservationsFromPoses(det, rv, tv, log);
}

// aruco_slam.cpp:88-263
void Slam::processQueue(std::priority_queue<Observation>& obs, FrameLog* log) {
    const std::vector<double> mu0 = mu;                                        // frozen copy (Q1)
    std::vector<Observation> observed;
    if (log) { log->popped.clear(); log->action.clear(); }
    while (!obs.empty()) {
        Observation ob = obs.top();
        obs.pop();
        const double* Rk = ob.R;
        int action;
        if (ob.index >= 0) {
            const int N = (int)mu.size();
            const int li = 3 + 3 * ob.index;
            auto S = [&](int r, int c) -> double& { return sigma[(size_t)c * N + r]; };
            double mx = mu0[li], my = mu0[li + 1], mth = mu0[li + 2];
            double x = mu0[0], y = mu0[1], theta = mu0[2];
            double sintheta = std::sin(theta), costheta = std::cos(theta);
            double gdx = mx - x, gdy = my - y, gdth = mth - theta;
            norm_angle(gdth);
            double z_hat[3] = {gdx * costheta + gdy * sintheta, -gdx * sintheta + gdy * costheta, gdth};
            double z[3] = {ob.x, ob.y, ob.theta};
            double ze[3] = {z[0] - z_hat[0], z[1] - z_hat[1], z[2] - z_hat[2]};
            norm_angle(ze[2]);
            double Gxm[18] = {-costheta, -sintheta, -gdx * sintheta + gdy * costheta, costheta, sintheta, 0,
                              sintheta, -costheta, -gdx * costheta - gdy * sintheta, -sintheta, costheta, 0,
                              0, 0, -1, 0, 0, 1};
            const int cols6[6] = {0, 1, 2, li, li + 1, li + 2};
            // PH = sigma * Gx^T (N x 3), HP = Gx * sigma (3 x N)
            std::vector<double> PH((size_t)N * 3), HP((size_t)3 * N);
            for (int r = 0; r < N; r++)
                for (int a = 0; a < 3; a++) {
                    double s = 0;
                    for (int k = 0; k < 6; k++) s += S(r, cols6[k]) * Gxm[a * 6 + k];
                    PH[(size_t)r * 3 + a] = s;
                }
            for (int a = 0; a < 3; a++)
                for (int c = 0; c < N; c++) {
                    double s = 0;
                    for (int k = 0; k < 6; k++) s += Gxm[a * 6 + k] * S(cols6[k], c);
                    HP[(size_t)a * N + c] = s;
                }
            double Sm[9], Si[9];
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) {
                    double s = 0;
                    for (int k = 0; k < 6; k++) s += HP[(size_t)a * N + cols6[k]] * Gxm[b * 6 + k];
                    Sm[a * 3 + b] = s + Rk[a * 3 + b];
                }
            inv3_lu(Sm, Si);
            std::vector<double> K((size_t)N * 3);
            for (int r = 0; r < N; r++)
                for (int b = 0; b < 3; b++)
                    K[(size_t)r * 3 + b] = PH[(size_t)r * 3] * Si[b] + PH[(size_t)r * 3 + 1] * Si[3 + b] + PH[(size_t)r * 3 + 2] * Si[6 + b];

            // "stationary" test against the previous frame (aruco_slam.cpp:192-198) — a no-op (Q2)
            auto last = std::find_if(last_observed.begin(), last_observed.end(),
                                     [&](const Observation& o) { return o.id == ob.id; });
            bool stationary = false;
            if (last != last_observed.end()) {
                double d0 = last->last_obs[0] - z[0], d1 = last->last_obs[1] - z[1], d2 = last->last_obs[2] - z[2];
                stationary = std::sqrt(d0 * d0 + d1 * d1 + d2 * d2) < 0.01;    // NaN compares false
            }
            if (stationary) {
                action = 2;
            } else {
                action = 1;
                ob.last_obs[0] = z[0]; ob.last_obs[1] = z[1]; ob.last_obs[2] = z[2];
                for (int r = 0; r < N; r++)
                    mu[r] += K[(size_t)r * 3] * ze[0] + K[(size_t)r * 3 + 1] * ze[1] + K[(size_t)r * 3 + 2] * ze[2];
                if (literal) {
                    // sigma_ = (I - K*Gx) * sigma_   (dense, aruco_slam.cpp:204)
                    std::vector<double> A((size_t)N * N, 0.0), O((size_t)N * N, 0.0);   // A row-major
                    for (int r = 0; r < N; r++) {
                        A[(size_t)r * N + r] = 1.0;
                        for (int k = 0; k < 6; k++) {
                            double g = K[(size_t)r * 3] * Gxm[k] + K[(size_t)r * 3 + 1] * Gxm[6 + k] + K[(size_t)r * 3 + 2] * Gxm[12 + k];
                            A[(size_t)r * N + cols6[k]] -= g;
                        }
                    }
                    for (int r = 0; r < N; r++)
                        for (int k = 0; k < N; k++) {
                            double a = A[(size_t)r * N + k];
                            if (a == 0.0) continue;
                            for (int c = 0; c < N; c++) O[(size_t)c * N + r] += a * S(k, c);
                        }
                    sigma.swap(O);
                } else {
                    for (int c = 0; c < N; c++)
                        for (int r = 0; r < N; r++)
                            S(r, c) -= K[(size_t)r * 3] * HP[c] + K[(size_t)r * 3 + 1] * HP[(size_t)N + c] + K[(size_t)r * 3 + 2] * HP[(size_t)2 * N + c];
                }
            }
        } else {
            action = 0;
            float sinth = (float)std::sin(mu0[2]);                              // float trig (Q4)
            float costh = (float)std::cos(mu0[2]);
            const int N = (int)mu.size();
            double map_x = mu0[0] + costh * ob.x - sinth * ob.y;
            double map_y = mu0[1] + sinth * ob.x + costh * ob.y;
            double map_theta = mu0[2] + ob.theta;
            norm_angle(map_theta);
            double deltax = map_x - mu0[0], deltay = map_y - mu0[1];
            double Gsk[9] = {-costh, -sinth, -sinth * deltax + costh * deltay,
                             sinth, -costh, -deltax * costh - deltay * sinth,
                             0, 0, -1};
            double Gmi[9] = {costh, sinth, 0, -sinth, costh, 0, 0, 0, 1};
            auto S = [&](int r, int c) -> double& { return sigma[(size_t)c * N + r]; };
            double ss[9];
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) ss[i * 3 + j] = S(i, j);
            // sigma_mm = Gmi * (Gsk*sigma_s*Gsk^T + Rk)^T * Gmi^T      (Q5, literal)
            double T1[9], T2[9], T3[9], smm[9];
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
                double s = 0; for (int k = 0; k < 3; k++) s += Gsk[i * 3 + k] * ss[k * 3 + j]; T1[i * 3 + j] = s; }
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
                double s = 0; for (int k = 0; k < 3; k++) s += T1[i * 3 + k] * Gsk[j * 3 + k]; T2[i * 3 + j] = s + Rk[i * 3 + j]; }
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
                double s = 0; for (int k = 0; k < 3; k++) s += Gmi[i * 3 + k] * T2[j * 3 + k]; T3[i * 3 + j] = s; }   // Gmi * T2^T
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
                double s = 0; for (int k = 0; k < 3; k++) s += T3[i * 3 + k] * Gmi[j * 3 + k]; smm[i * 3 + j] = s; }
            // sigma_mx = -Gmi * Gsk * sigma_.topRows(3)
            double G[9];
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
                double s = 0; for (int k = 0; k < 3; k++) s += (-Gmi[i * 3 + k]) * Gsk[k * 3 + j]; G[i * 3 + j] = s; }
            std::vector<double> smx((size_t)3 * N);
            for (int i = 0; i < 3; i++)
                for (int c = 0; c < N; c++) smx[(size_t)i * N + c] = G[i * 3] * S(0, c) + G[i * 3 + 1] * S(1, c) + G[i * 3 + 2] * S(2, c);
            const int N2 = N + 3;
            std::vector<double> ns((size_t)N2 * N2, 0.0);
            for (int c = 0; c < N; c++) for (int r = 0; r < N; r++) ns[(size_t)c * N2 + r] = S(r, c);
            for (int i = 0; i < 3; i++)
                for (int c = 0; c < N; c++) {
                    ns[(size_t)c * N2 + N + i] = smx[(size_t)i * N + c];        // bottom-left
                    ns[(size_t)(N + i) * N2 + c] = smx[(size_t)i * N + c];      // top-right (transpose)
                }
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) ns[(size_t)(N + j) * N2 + N + i] = smm[i * 3 + j];
            sigma.swap(ns);
            mu.push_back(map_x); mu.push_back(map_y); mu.push_back(map_theta);
            aruco_id_map.insert({ob.id, ((int)mu.size() - 3) / 3 - 1});         // keeps the first (Q10)
        }
        observed.push_back(ob);
        if (log) { log->popped.push_back(ob); log->action.push_back(action); }
    }
    last_observed = observed;
}

} // namespace oracle
