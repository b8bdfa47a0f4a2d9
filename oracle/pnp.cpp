// ORACLE (test infrastructure only — see oracle.h).  CPU restatement of
// cv::aruco::estimatePoseSingleMarkers -> cv::solvePnP(SOLVEPNP_ITERATIVE) (reference call site
// src/aruco_slam.cpp:314), cv::Rodrigues (:354, :478) and cv::projectPoints (:441).  The code lives
// in third-party OpenCV 3.2.0 calib3d (cvFindExtrinsicCameraParams2, cvRodrigues2,
// cvProjectPoints2, cvUndistortPoints, CvLevMarq), absent here; this follows its published algorithm:
// undistort -> planar branch (4-point homography on the float-converted points) -> R from h1,h2,h1xh2
// re-orthogonalised -> Levenberg-Marquardt, 6 params / 8 residuals, <= 20 iterations, eps FLT_EPSILON.
// Where OpenCV uses an SVD (3x3 polar factor, 6x6 normal equations, homography null vector) this spec
// fixes a closed/elimination form with the same solution up to rounding; poses are compared at 1e-4.
#include "oracle.h"
#include <cmath>
#include <cfloat>
#include <cstring>
#include <algorithm>

namespace oracle {

static void solve_ge(int n, double* A /*n*n row-major, destroyed*/, double* b, double* x) {
    for (int col = 0; col < n; col++) {
        int piv = col;
        double best = std::fabs(A[col * n + col]);
        for (int r = col + 1; r < n; r++)
            if (std::fabs(A[r * n + col]) > best) { best = std::fabs(A[r * n + col]); piv = r; }
        if (piv != col) {
            for (int c = 0; c < n; c++) std::swap(A[piv * n + c], A[col * n + c]);
            std::swap(b[piv], b[col]);
        }
        for (int r = col + 1; r < n; r++) {
            double f = A[r * n + col] / A[col * n + col];
            for (int c = col; c < n; c++) A[r * n + c] -= f * A[col * n + c];
            b[r] -= f * b[col];
        }
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = b[i];
        for (int c = i + 1; c < n; c++) s -= A[i * n + c] * x[c];
        x[i] = s / A[i * n + i];
    }
}

static void mat3_mul(const double A[9], const double B[9], double C[9]) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}

static double det3(const double m[9]) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// cvRodrigues2, vector -> matrix, with d R / d r (3 x 9, row i = derivative wrt r_i)
void rodrigues_vec_to_mat(const double rv[3], double R[9], double J[27]) {
    double rx = rv[0], ry = rv[1], rz = rv[2];
    double theta = std::sqrt(rx * rx + ry * ry + rz * rz);
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1. : 0.;
        if (J) {
            std::memset(J, 0, sizeof(double) * 27);
            J[5] = J[15] = J[19] = -1;
            J[7] = J[11] = J[21] = 1;
        }
        return;
    }
    double c = std::cos(theta), s = std::sin(theta), c1 = 1. - c, itheta = theta ? 1. / theta : 0.;
    rx *= itheta; ry *= itheta; rz *= itheta;
    double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    static const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * r_x[k];
    if (J) {
        double drrt[27] = {rx + rx, ry, rz, ry, 0, 0, rz, 0, 0,
                           0, rx, 0, rx, ry + ry, rz, 0, rz, 0,
                           0, 0, rx, 0, 0, ry, rx, ry, rz + rz};
        static const double d_r_x_[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0,
                                          0, 0, 1, 0, 0, 0, -1, 0, 0,
                                          0, -1, 0, 1, 0, 0, 0, 0, 0};
        for (int i = 0; i < 3; i++) {
            double ri = i == 0 ? rx : i == 1 ? ry : rz;
            double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta;
            double a3 = (c - s * itheta) * ri, a4 = s * itheta;
            for (int k = 0; k < 9; k++)
                J[i * 9 + k] = a0 * I[k] + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * r_x[k] + a4 * d_r_x_[i * 9 + k];
        }
    }
}

// cvRodrigues2, matrix -> vector.  OpenCV first replaces R by U*Vt of its SVD (the orthogonal polar
// factor); this spec computes the same factor with the Newton iteration X <- (X + X^-T)/2.
void rodrigues_mat_to_vec(const double Rin[9], double r[3]) {
    double R[9];
    std::memcpy(R, Rin, sizeof(R));
    for (int it = 0; it < 30; it++) {
        double d = det3(R);
        if (d == 0.) break;
        double id = 1. / d;
        // inverse transpose = cofactor matrix / det
        double C[9] = {(R[4] * R[8] - R[5] * R[7]) * id, (R[5] * R[6] - R[3] * R[8]) * id, (R[3] * R[7] - R[4] * R[6]) * id,
                       (R[2] * R[7] - R[1] * R[8]) * id, (R[0] * R[8] - R[2] * R[6]) * id, (R[1] * R[6] - R[0] * R[7]) * id,
                       (R[1] * R[5] - R[2] * R[4]) * id, (R[2] * R[3] - R[0] * R[5]) * id, (R[0] * R[4] - R[1] * R[3]) * id};
        double delta = 0;
        for (int k = 0; k < 9; k++) {
            double n = 0.5 * (R[k] + C[k]);
            delta = std::max(delta, std::fabs(n - R[k]));
            R[k] = n;
        }
        if (delta < 1e-15) break;
    }
    double x = R[7] - R[5], y = R[2] - R[6], z = R[3] - R[1];
    double s = std::sqrt((x * x + y * y + z * z) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = std::acos(c);
    if (s < 1e-5) {
        if (c > 0) { r[0] = r[1] = r[2] = 0; return; }
        double t;
        t = (R[0] + 1) * 0.5; x = std::sqrt(std::max(t, 0.));
        t = (R[4] + 1) * 0.5; y = std::sqrt(std::max(t, 0.)) * (R[1] < 0 ? -1. : 1.);
        t = (R[8] + 1) * 0.5; z = std::sqrt(std::max(t, 0.)) * (R[2] < 0 ? -1. : 1.);
        if (std::fabs(x) < std::fabs(y) && std::fabs(x) < std::fabs(z) && (R[5] > 0) != (y * z > 0)) z = -z;
        theta /= std::sqrt(x * x + y * y + z * z);
        r[0] = x * theta; r[1] = y * theta; r[2] = z * theta;
    } else {
        double vth = 1 / (2 * s);
        vth *= theta;
        r[0] = x * vth; r[1] = y * vth; r[2] = z * vth;
    }
}

// cvProjectPoints2 with the 5-coefficient plumb_bob model (k1,k2,p1,p2,k3), optional Jacobians.
void project_points(const double obj[][3], int n, const double rv[3], const double t[3], const Camera& cam,
                    double out[][2], double* dpdr, double* dpdt) {
    double R[9], dRdr[27];
    rodrigues_vec_to_mat(rv, R, dpdr ? dRdr : nullptr);
    double k[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < cam.nD && i < 5; i++) k[i] = cam.D[i];
    const double fx = cam.K[0], fy = cam.K[4], cx = cam.K[2], cy = cam.K[5];
    for (int i = 0; i < n; i++) {
        double X = obj[i][0], Y = obj[i][1], Z = obj[i][2];
        double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
        double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
        double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
        z = z ? 1. / z : 1;
        x *= z; y *= z;
        double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
        double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
        double cdist = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6;
        double xd = x * cdist + k[2] * a1 + k[3] * a2;
        double yd = y * cdist + k[2] * a3 + k[3] * a1;
        out[i][0] = xd * fx + cx;
        out[i][1] = yd * fy + cy;
        if (dpdt) {
            double dxdt[3] = {z, 0, -x * z}, dydt[3] = {0, z, -y * z};
            for (int j = 0; j < 3; j++) {
                double dr2dt = 2 * x * dxdt[j] + 2 * y * dydt[j];
                double dcdist_dt = k[0] * dr2dt + 2 * k[1] * r2 * dr2dt + 3 * k[4] * r4 * dr2dt;
                double da1dt = 2 * (x * dydt[j] + y * dxdt[j]);
                double dmxdt = dxdt[j] * cdist + x * dcdist_dt + k[2] * da1dt + k[3] * (dr2dt + 4 * x * dxdt[j]);
                double dmydt = dydt[j] * cdist + y * dcdist_dt + k[2] * (dr2dt + 4 * y * dydt[j]) + k[3] * da1dt;
                dpdt[(2 * i) * 3 + j] = fx * dmxdt;
                dpdt[(2 * i + 1) * 3 + j] = fy * dmydt;
            }
        }
        if (dpdr) {
            double dx0dr[3] = {X * dRdr[0] + Y * dRdr[1] + Z * dRdr[2], X * dRdr[9] + Y * dRdr[10] + Z * dRdr[11],
                               X * dRdr[18] + Y * dRdr[19] + Z * dRdr[20]};
            double dy0dr[3] = {X * dRdr[3] + Y * dRdr[4] + Z * dRdr[5], X * dRdr[12] + Y * dRdr[13] + Z * dRdr[14],
                               X * dRdr[21] + Y * dRdr[22] + Z * dRdr[23]};
            double dz0dr[3] = {X * dRdr[6] + Y * dRdr[7] + Z * dRdr[8], X * dRdr[15] + Y * dRdr[16] + Z * dRdr[17],
                               X * dRdr[24] + Y * dRdr[25] + Z * dRdr[26]};
            for (int j = 0; j < 3; j++) {
                double dxdr = z * (dx0dr[j] - x * dz0dr[j]);
                double dydr = z * (dy0dr[j] - y * dz0dr[j]);
                double dr2dr = 2 * x * dxdr + 2 * y * dydr;
                double dcdist_dr = (k[0] + 2 * k[1] * r2 + 3 * k[4] * r4) * dr2dr;
                double da1dr = 2 * (x * dydr + y * dxdr);
                double dmxdr = dxdr * cdist + x * dcdist_dr + k[2] * da1dr + k[3] * (dr2dr + 4 * x * dxdr);
                double dmydr = dydr * cdist + y * dcdist_dr + k[2] * (dr2dr + 4 * y * dydr) + k[3] * da1dr;
                dpdr[(2 * i) * 3 + j] = fx * dmxdr;
                dpdr[(2 * i + 1) * 3 + j] = fy * dmydr;
            }
        }
    }
}

// cvUndistortPoints (R = I, no P): normalised coordinates, 5 fixed-point iterations when a
// distortion vector is supplied (the node always passes cinfo->D, aruco_slam_node.cpp:125).
void undistort_points(const double in[][2], int n, const Camera& cam, double out[][2]) {
    double k[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < cam.nD && i < 5; i++) k[i] = cam.D[i];
    const int iters = cam.nD > 0 ? 5 : 0;
    const double ifx = 1. / cam.K[0], ify = 1. / cam.K[4], cx = cam.K[2], cy = cam.K[5];
    for (int i = 0; i < n; i++) {
        double x, y, x0, y0;
        x0 = x = (in[i][0] - cx) * ifx;
        y0 = y = (in[i][1] - cy) * ify;
        for (int j = 0; j < iters; j++) {
            double r2 = x * x + y * y;
            double icdist = 1. / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
            double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        out[i][0] = x;
        out[i][1] = y;
    }
}

// cv::findHomography(method 0) for exactly 4 correspondences: points are float (findHomography converts
// to CV_32F), Hartley-normalised exactly as HomographyEstimatorCallback::runKernel; the exact
// 4-point solution is obtained by elimination instead of the 9x9 eigen-decomposition.
bool find_homography4(const float M[4][2], const float m[4][2], double H[9]) {
    const int count = 4;
    double cMx = 0, cMy = 0, cmx = 0, cmy = 0, sMx = 0, sMy = 0, smx = 0, smy = 0;
    for (int i = 0; i < count; i++) { cmx += m[i][0]; cmy += m[i][1]; cMx += M[i][0]; cMy += M[i][1]; }
    cmx /= count; cmy /= count; cMx /= count; cMy /= count;
    for (int i = 0; i < count; i++) {
        smx += std::fabs(m[i][0] - cmx); smy += std::fabs(m[i][1] - cmy);
        sMx += std::fabs(M[i][0] - cMx); sMy += std::fabs(M[i][1] - cMy);
    }
    if (std::fabs(smx) < DBL_EPSILON || std::fabs(smy) < DBL_EPSILON || std::fabs(sMx) < DBL_EPSILON || std::fabs(sMy) < DBL_EPSILON)
        return false;
    smx = count / smx; smy = count / smy; sMx = count / sMx; sMy = count / sMy;
    double invHnorm[9] = {1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1};
    double Hnorm2[9] = {sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1};
    double A[64], b[8], h[8];
    for (int i = 0; i < count; i++) {
        double x = (m[i][0] - cmx) * smx, y = (m[i][1] - cmy) * smy;
        double X = (M[i][0] - cMx) * sMx, Y = (M[i][1] - cMy) * sMy;
        double* r0 = &A[(2 * i) * 8];
        double* r1 = &A[(2 * i + 1) * 8];
        r0[0] = X; r0[1] = Y; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -x * X; r0[7] = -x * Y; b[2 * i] = x;
        r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = X; r1[4] = Y; r1[5] = 1; r1[6] = -y * X; r1[7] = -y * Y; b[2 * i + 1] = y;
    }
    solve_ge(8, A, b, h);
    double H0[9] = {h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], 1.}, T[9];
    mat3_mul(invHnorm, H0, T);
    mat3_mul(T, Hnorm2, H);
    double s = 1. / H[8];
    for (int i = 0; i < 9; i++) H[i] *= s;
    for (int i = 0; i < 9; i++) if (!std::isfinite(H[i])) return false;
    return true;
}

// cvFindExtrinsicCameraParams2 (useExtrinsicGuess = false) specialised to the 4 marker corners, whose
// centroid is the origin and which span the z = 0 plane, so the planar branch applies with
// R_transform = I, T_transform = 0 (aruco.cpp::_getSingleMarkerObjectPoints gives the object points).
void solve_pnp_marker(const Pt2f corners[4], float markerLength, const Camera& cam, double rvec[3], double tvec[3],
                      int* iters_out) {
    const float hl = markerLength / 2.f;
    const float objf[4][3] = {{-hl, hl, 0}, {hl, hl, 0}, {hl, -hl, 0}, {-hl, -hl, 0}};
    double obj[4][3], m[4][2], mn[4][2];
    for (int i = 0; i < 4; i++) {
        for (int k = 0; k < 3; k++) obj[i][k] = objf[i][k];
        m[i][0] = corners[i].x;
        m[i][1] = corners[i].y;
    }
    undistort_points(m, 4, cam, mn);

    double param[6];
    {
        float Mxy[4][2], mnf[4][2];
        for (int i = 0; i < 4; i++) {
            Mxy[i][0] = (float)obj[i][0]; Mxy[i][1] = (float)obj[i][1];
            mnf[i][0] = (float)mn[i][0];  mnf[i][1] = (float)mn[i][1];
        }
        double h[9], R[9];
        if (find_homography4(Mxy, mnf, h)) {
            double h1n = std::sqrt(h[0] * h[0] + h[3] * h[3] + h[6] * h[6]);
            double h2n = std::sqrt(h[1] * h[1] + h[4] * h[4] + h[7] * h[7]);
            double s1 = 1. / std::max(h1n, DBL_EPSILON), s2 = 1. / std::max(h2n, DBL_EPSILON);
            double st = 2. / std::max(h1n + h2n, DBL_EPSILON);
            double t0 = h[2] * st, t1 = h[5] * st, t2 = h[8] * st;
            h[0] *= s1; h[3] *= s1; h[6] *= s1;
            h[1] *= s2; h[4] *= s2; h[7] *= s2;
            h[2] = h[3] * h[7] - h[6] * h[4];
            h[5] = h[6] * h[1] - h[0] * h[7];
            h[8] = h[0] * h[4] - h[3] * h[1];
            double r[3];
            rodrigues_mat_to_vec(h, r);
            rodrigues_vec_to_mat(r, R, nullptr);
            param[3] = t0; param[4] = t1; param[5] = t2;     // + matH * T_transform (= 0)
        } else {
            for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1. : 0.;
            param[3] = param[4] = param[5] = 0;
        }
        rodrigues_mat_to_vec(R, param);
    }

    // CvLevMarq(6, 8, {EPS+ITER, 20, FLT_EPSILON}, completeSymm) — modules/calib3d/src/compat_ptsetreg.cpp
    const int max_iter = 20;
    const double epsilon = FLT_EPSILON;
    double prevParam[6], J[8 * 6], err[8], JtJ[36], JtErr[6];
    double proj[4][2], dpdr[24], dpdt[24];
    double prevErrNorm = DBL_MAX, errNorm = DBL_MAX;
    int lambdaLg10 = -3, iters = 0;
    const double LOG10 = std::log(10.);

    auto eval = [&](bool withJ) {
        project_points(obj, 4, param, param + 3, cam, proj, withJ ? dpdr : nullptr, withJ ? dpdt : nullptr);
        for (int i = 0; i < 4; i++) { err[2 * i] = proj[i][0] - m[i][0]; err[2 * i + 1] = proj[i][1] - m[i][1]; }
        if (withJ)
            for (int r = 0; r < 8; r++)
                for (int c = 0; c < 3; c++) { J[r * 6 + c] = dpdr[r * 3 + c]; J[r * 6 + 3 + c] = dpdt[r * 3 + c]; }
    };
    auto norm8 = [&]() { double s = 0; for (int i = 0; i < 8; i++) s += err[i] * err[i]; return std::sqrt(s); };
    auto step = [&]() {
        double lambda = std::exp(lambdaLg10 * LOG10);
        double A[36], b[6], x[6];
        std::memcpy(A, JtJ, sizeof(A));
        std::memcpy(b, JtErr, sizeof(b));
        for (int i = 0; i < 6; i++) A[i * 6 + i] *= 1. + lambda;
        solve_ge(6, A, b, x);
        for (int i = 0; i < 6; i++) param[i] = prevParam[i] - x[i];
    };

    eval(true);                                    // STARTED -> CALC_J
    for (;;) {
        // CALC_J
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) {
                double s = 0;
                for (int r = 0; r < 8; r++) s += J[r * 6 + i] * J[r * 6 + j];
                JtJ[i * 6 + j] = s;
            }
        for (int i = 0; i < 6; i++) {
            double s = 0;
            for (int r = 0; r < 8; r++) s += J[r * 6 + i] * err[r];
            JtErr[i] = s;
        }
        std::memcpy(prevParam, param, sizeof(prevParam));
        step();
        if (iters == 0) prevErrNorm = norm8();
        eval(false);
        // CHECK_ERR
        bool done = false;
        for (;;) {
            errNorm = norm8();
            if (errNorm > prevErrNorm) {
                if (++lambdaLg10 <= 16) { step(); eval(false); continue; }
            }
            lambdaLg10 = std::max(lambdaLg10 - 1, -16);
            double dn = 0, pn = 0;
            for (int i = 0; i < 6; i++) { dn += (param[i] - prevParam[i]) * (param[i] - prevParam[i]); pn += prevParam[i] * prevParam[i]; }
            if (++iters >= max_iter || std::sqrt(dn) / std::sqrt(pn) < epsilon) { done = true; break; }
            prevErrNorm = errNorm;
            eval(true);
            break;
        }
        if (done) break;
    }
    for (int i = 0; i < 3; i++) { rvec[i] = param[i]; tvec[i] = param[3 + i]; }
    if (iters_out) *iters_out = iters;
}

} // namespace oracle
