// Per-frame marker finalisation and pose on gfx950.  Replaces, for every frame of the batch:
//   _filterDetectedMarkers (tail of cv::aruco::detectMarkers, aruco_slam.cpp:313),
//   cv::aruco::estimatePoseSingleMarkers -> solvePnP(SOLVEPNP_ITERATIVE) per marker (aruco_slam.cpp:314),
//   the per-detection body of ArucoSlam::getObservations (aruco_slam.cpp:325-369): range gate, Rodrigues,
//   (x, y, theta) observation, CalculateCovariance (aruco_slam.cpp:437-471) and the covariance gate.
// One workgroup per frame; the pose solve is one lane per marker (6-parameter Levenberg-Marquardt on 8
// residuals in fp64 — a latency-bound scalar chain; frames of the batch supply the parallelism).
#include "common.h"
#include "pose.h"
#include <cfloat>

namespace aslam {

// ---- small dense helpers (fp64, per lane) --------------------------------------------------------------
template <int N> __device__ __forceinline__ void solve_pp(double* A, double* b, double* x) {   // Gaussian elimination, partial pivoting
    // Every index below is a compile-time constant once the loops are unrolled - the pivot row is exchanged by selects over the
    // candidate rows, not through a run-time index - so that A, b and x live in registers (a run-time row index sends the whole
    // matrix to scratch memory, and the LM iterations of k_pose are one long chain through it).
#pragma unroll
    for (int col = 0; col < N; col++) {
        int piv = col;
        double best = fabs(A[col * N + col]);
#pragma unroll
        for (int r = col + 1; r < N; r++) {
            double v = fabs(A[r * N + col]);
            if (v > best) { best = v; piv = r; }
        }
#pragma unroll
        for (int r = col + 1; r < N; r++) {
            const bool sw = piv == r;                            // true for at most one r
#pragma unroll
            for (int c = 0; c < N; c++) {
                const double u = A[col * N + c], w = A[r * N + c];
                A[col * N + c] = sw ? w : u;
                A[r * N + c] = sw ? u : w;
            }
            const double u = b[col], w = b[r];
            b[col] = sw ? w : u;
            b[r] = sw ? u : w;
        }
#pragma unroll
        for (int r = col + 1; r < N; r++) {
            double f = A[r * N + col] / A[col * N + col];
#pragma unroll
            for (int c = col; c < N; c++) A[r * N + c] -= f * A[col * N + c];
            b[r] -= f * b[col];
        }
    }
#pragma unroll
    for (int i = N - 1; i >= 0; i--) {
        double s = b[i];
#pragma unroll
        for (int c = i + 1; c < N; c++) s -= A[i * N + c] * x[c];
        x[i] = s / A[i * N + i];
    }
}

__device__ void mul33(const double* A, const double* B, double* C) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}

// Rodrigues vector -> matrix, optional dR/dr (3 x 9)
__device__ void rodrigues_fwd(const double* rv, double* R, double* J) {
    double rx = rv[0], ry = rv[1], rz = rv[2];
    double theta = sqrt(rx * rx + ry * ry + rz * rz);
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1. : 0.;
        if (J) {
            for (int i = 0; i < 27; i++) J[i] = 0;
            J[5] = J[15] = J[19] = -1;
            J[7] = J[11] = J[21] = 1;
        }
        return;
    }
    double c = cos(theta), s = sin(theta), c1 = 1. - c, it = 1. / theta;
    rx *= it; ry *= it; rz *= it;
    double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    double rX[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    for (int k = 0; k < 9; k++) R[k] = c * ((k % 4 == 0) ? 1. : 0.) + c1 * rrt[k] + s * rX[k];
    if (J) {
        double drrt[27] = {rx + rx, ry, rz, ry, 0, 0, rz, 0, 0, 0, rx, 0, rx, ry + ry, rz, 0, rz, 0, 0, 0, rx, 0, 0, ry, rx, ry, rz + rz};
        const double drX[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 1, 0, 0, 0, -1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0};
        for (int i = 0; i < 3; i++) {
            double ri = i == 0 ? rx : i == 1 ? ry : rz;
            double a0 = -s * ri, a1 = (s - 2 * c1 * it) * ri, a2 = c1 * it, a3 = (c - s * it) * ri, a4 = s * it;
            for (int k = 0; k < 9; k++)
                J[i * 9 + k] = a0 * ((k % 4 == 0) ? 1. : 0.) + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * rX[k] + a4 * drX[i * 9 + k];
        }
    }
}

// Rodrigues matrix -> vector; the input is first replaced by its orthogonal polar factor (what U*Vt of the SVD is)
__device__ void rodrigues_inv(const double* Rin, double* r) {
    double R[9];
    for (int i = 0; i < 9; i++) R[i] = Rin[i];
    for (int it = 0; it < 30; it++) {
        double d = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
        if (d == 0.) break;
        double id = 1. / d;
        double C[9] = {(R[4] * R[8] - R[5] * R[7]) * id, (R[5] * R[6] - R[3] * R[8]) * id, (R[3] * R[7] - R[4] * R[6]) * id,
                       (R[2] * R[7] - R[1] * R[8]) * id, (R[0] * R[8] - R[2] * R[6]) * id, (R[1] * R[6] - R[0] * R[7]) * id,
                       (R[1] * R[5] - R[2] * R[4]) * id, (R[2] * R[3] - R[0] * R[5]) * id, (R[0] * R[4] - R[1] * R[3]) * id};
        double delta = 0;
        for (int k = 0; k < 9; k++) {
            double n = 0.5 * (R[k] + C[k]);
            delta = fmax(delta, fabs(n - R[k]));
            R[k] = n;
        }
        if (delta < 1e-15) break;
    }
    double x = R[7] - R[5], y = R[2] - R[6], z = R[3] - R[1];
    double s = sqrt((x * x + y * y + z * z) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0) { r[0] = r[1] = r[2] = 0; return; }
        double t;
        t = (R[0] + 1) * 0.5; x = sqrt(fmax(t, 0.));
        t = (R[4] + 1) * 0.5; y = sqrt(fmax(t, 0.)) * (R[1] < 0 ? -1. : 1.);
        t = (R[8] + 1) * 0.5; z = sqrt(fmax(t, 0.)) * (R[2] < 0 ? -1. : 1.);
        if (fabs(x) < fabs(y) && fabs(x) < fabs(z) && (R[5] > 0) != (y * z > 0)) z = -z;
        theta /= sqrt(x * x + y * y + z * z);
        r[0] = x * theta; r[1] = y * theta; r[2] = z * theta;
    } else {
        double vth = 1 / (2 * s);
        vth *= theta;
        r[0] = x * vth; r[1] = y * vth; r[2] = z * vth;
    }
}

// pinhole + plumb_bob projection of the 4 marker corners; J (8 x 6, [d/dr | d/dt]) optional
__device__ void project4(const double* p /*r,t*/, double hl, const CamParams& cam, double* out /*8*/, double* J) {
    double R[9], dR[27];
    rodrigues_fwd(p, R, J ? dR : nullptr);
    const double* k = cam.k;
    for (int i = 0; i < 4; i++) {
        const double X = (i == 0 || i == 3) ? -hl : hl, Y = (i < 2) ? hl : -hl;     // Z = 0
        double x = R[0] * X + R[1] * Y + p[3];
        double y = R[3] * X + R[4] * Y + p[4];
        double z = R[6] * X + R[7] * Y + p[5];
        z = z ? 1. / z : 1;
        x *= z; y *= z;
        double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
        double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
        double cd = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6;
        out[2 * i] = (x * cd + k[2] * a1 + k[3] * a2) * cam.fx + cam.cx;
        out[2 * i + 1] = (y * cd + k[2] * a3 + k[3] * a1) * cam.fy + cam.cy;
        if (J) {
            for (int j = 0; j < 6; j++) {
                double dxd, dyd;      // derivative of the normalised (x, y)
                if (j < 3) {
                    double dx0 = X * dR[j * 9] + Y * dR[j * 9 + 1];
                    double dy0 = X * dR[j * 9 + 3] + Y * dR[j * 9 + 4];
                    double dz0 = X * dR[j * 9 + 6] + Y * dR[j * 9 + 7];
                    dxd = z * (dx0 - x * dz0);
                    dyd = z * (dy0 - y * dz0);
                } else {
                    dxd = j == 3 ? z : (j == 4 ? 0. : -x * z);
                    dyd = j == 3 ? 0. : (j == 4 ? z : -y * z);
                }
                double dr2 = 2 * x * dxd + 2 * y * dyd;
                double dcd = (k[0] + 2 * k[1] * r2 + 3 * k[4] * r4) * dr2;
                double da1 = 2 * (x * dyd + y * dxd);
                double dmx = dxd * cd + x * dcd + k[2] * da1 + k[3] * (dr2 + 4 * x * dxd);
                double dmy = dyd * cd + y * dcd + k[2] * (dr2 + 4 * y * dyd) + k[3] * da1;
                J[(2 * i) * 6 + j] = cam.fx * dmx;
                J[(2 * i + 1) * 6 + j] = cam.fy * dmy;
            }
        }
    }
}

// solvePnP(ITERATIVE) for one marker: undistort -> 4-point homography (float inputs, Hartley-normalised)
// -> R from h1, h2, h1 x h2 -> LM (<= 20 iterations, eps FLT_EPSILON, lambda = 10^k starting at k = -3)
__device__ void solve_marker_pose(const float* c8, float markerLength, const CamParams& cam, double* rvec, double* tvec) {
    const float hlf = markerLength / 2.f;
    const double hl = (double)hlf;
    double m[8];
    for (int i = 0; i < 8; i++) m[i] = c8[i];
    double param[6];
    {
        // cvUndistortPoints: 5 fixed-point iterations when a distortion vector is given
        float mn[8];
        const double ifx = 1. / cam.fx, ify = 1. / cam.fy;
        const int iters = cam.nD > 0 ? 5 : 0;
        for (int i = 0; i < 4; i++) {
            double x0 = (m[2 * i] - cam.cx) * ifx, y0 = (m[2 * i + 1] - cam.cy) * ify, x = x0, y = y0;
            for (int j = 0; j < iters; j++) {
                double r2 = x * x + y * y;
                double icd = 1. / (1 + ((cam.k[4] * r2 + cam.k[1]) * r2 + cam.k[0]) * r2);
                double dX = 2 * cam.k[2] * x * y + cam.k[3] * (r2 + 2 * x * x);
                double dY = cam.k[2] * (r2 + 2 * y * y) + 2 * cam.k[3] * x * y;
                x = (x0 - dX) * icd;
                y = (y0 - dY) * icd;
            }
            mn[2 * i] = (float)x;
            mn[2 * i + 1] = (float)y;
        }
        // object plane points (float) -> normalised image points (float)
        const float MX[4] = {-hlf, hlf, hlf, -hlf}, MY[4] = {hlf, hlf, -hlf, -hlf};
        double cMx = 0, cMy = 0, cmx = 0, cmy = 0, sMx = 0, sMy = 0, smx = 0, smy = 0;
        for (int i = 0; i < 4; i++) { cmx += mn[2 * i]; cmy += mn[2 * i + 1]; cMx += MX[i]; cMy += MY[i]; }
        cmx /= 4; cmy /= 4; cMx /= 4; cMy /= 4;
        for (int i = 0; i < 4; i++) {
            smx += fabs(mn[2 * i] - cmx); smy += fabs(mn[2 * i + 1] - cmy);
            sMx += fabs(MX[i] - cMx); sMy += fabs(MY[i] - cMy);
        }
        bool okH = !(fabs(smx) < DBL_EPSILON || fabs(smy) < DBL_EPSILON || fabs(sMx) < DBL_EPSILON || fabs(sMy) < DBL_EPSILON);
        double H[9];
        if (okH) {
            smx = 4 / smx; smy = 4 / smy; sMx = 4 / sMx; sMy = 4 / sMy;
            double A[64], b[8], h[8];
            for (int i = 0; i < 4; i++) {
                double x = (mn[2 * i] - cmx) * smx, y = (mn[2 * i + 1] - cmy) * smy;
                double X = (MX[i] - cMx) * sMx, Y = (MY[i] - cMy) * sMy;
                double* r0 = &A[(2 * i) * 8];
                double* r1 = &A[(2 * i + 1) * 8];
                r0[0] = X; r0[1] = Y; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -x * X; r0[7] = -x * Y; b[2 * i] = x;
                r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = X; r1[4] = Y; r1[5] = 1; r1[6] = -y * X; r1[7] = -y * Y; b[2 * i + 1] = y;
            }
            solve_pp<8>(A, b, h);
            double invHn[9] = {1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1};
            double Hn2[9] = {sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1};
            double H0[9] = {h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], 1.}, T[9];
            mul33(invHn, H0, T);
            mul33(T, Hn2, H);
            double sc = 1. / H[8];
            for (int i = 0; i < 9; i++) { H[i] *= sc; if (!isfinite(H[i])) okH = false; }
        }
        double R[9];
        if (okH) {
            double h1n = sqrt(H[0] * H[0] + H[3] * H[3] + H[6] * H[6]);
            double h2n = sqrt(H[1] * H[1] + H[4] * H[4] + H[7] * H[7]);
            double s1 = 1. / fmax(h1n, DBL_EPSILON), s2 = 1. / fmax(h2n, DBL_EPSILON), st = 2. / fmax(h1n + h2n, DBL_EPSILON);
            param[3] = H[2] * st; param[4] = H[5] * st; param[5] = H[8] * st;
            H[0] *= s1; H[3] *= s1; H[6] *= s1;
            H[1] *= s2; H[4] *= s2; H[7] *= s2;
            H[2] = H[3] * H[7] - H[6] * H[4];
            H[5] = H[6] * H[1] - H[0] * H[7];
            H[8] = H[0] * H[4] - H[3] * H[1];
            double r[3];
            rodrigues_inv(H, r);
            rodrigues_fwd(r, R, nullptr);
        } else {
            for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1. : 0.;
            param[3] = param[4] = param[5] = 0;
        }
        rodrigues_inv(R, param);
    }

    // CvLevMarq state machine
    double prevParam[6], J[48], err[8], JtJ[36], JtErr[6], proj[8];
    double prevErrNorm = DBL_MAX, errNorm;
    int lambdaLg10 = -3, iters = 0;
    const double LOG10 = log(10.);
    project4(param, hl, cam, proj, J);
    for (int i = 0; i < 8; i++) err[i] = proj[i] - m[i];
    for (;;) {
        for (int i = 0; i < 6; i++) {
            for (int j = 0; j < 6; j++) {
                double s = 0;
                for (int r = 0; r < 8; r++) s += J[r * 6 + i] * J[r * 6 + j];
                JtJ[i * 6 + j] = s;
            }
            double s = 0;
            for (int r = 0; r < 8; r++) s += J[r * 6 + i] * err[r];
            JtErr[i] = s;
            prevParam[i] = param[i];
        }
        if (iters == 0) { double s = 0; for (int i = 0; i < 8; i++) s += err[i] * err[i]; prevErrNorm = sqrt(s); }
        bool done = false;
        for (;;) {
            {   // step(): solve (JtJ with diagonal * (1 + lambda)) x = JtErr ; param = prevParam - x
                double lambda = exp(lambdaLg10 * LOG10);
                double A[36], b[6], x[6];
                for (int i = 0; i < 36; i++) A[i] = JtJ[i];
                for (int i = 0; i < 6; i++) { A[i * 6 + i] *= 1. + lambda; b[i] = JtErr[i]; }
                solve_pp<6>(A, b, x);
                for (int i = 0; i < 6; i++) param[i] = prevParam[i] - x[i];
            }
            project4(param, hl, cam, proj, nullptr);
            double s = 0;
            for (int i = 0; i < 8; i++) { err[i] = proj[i] - m[i]; s += err[i] * err[i]; }
            errNorm = sqrt(s);
            if (errNorm > prevErrNorm && ++lambdaLg10 <= 16) continue;
            lambdaLg10 = max(lambdaLg10 - 1, -16);
            double dn = 0, pn = 0;
            for (int i = 0; i < 6; i++) { double d = param[i] - prevParam[i]; dn += d * d; pn += prevParam[i] * prevParam[i]; }
            if (++iters >= 20 || sqrt(dn) / sqrt(pn) < (double)FLT_EPSILON) done = true;
            break;
        }
        if (done) break;
        prevErrNorm = errNorm;
        project4(param, hl, cam, proj, J);
        for (int i = 0; i < 8; i++) err[i] = proj[i] - m[i];
    }
    for (int i = 0; i < 3; i++) { rvec[i] = param[i]; tvec[i] = param[3 + i]; }
}

__device__ __forceinline__ void wrap_once(double& a) {      // ArucoSlam::normAngle, aruco_slam.cpp:412-421
    const double PI = 3.14159265358979323846;
    if (a >= PI) a -= 2.0 * PI;
    if (a < -PI) a += 2.0 * PI;
}

// cv::pointPolygonTest(measureDist = false), float contour of 4 points
__device__ int point_in_quad(const float* q, float px, float py) {
    int counter = 0;
    float vx = q[6], vy = q[7], v0x, v0y;
    for (int i = 0; i < 4; i++) {
        v0x = vx; v0y = vy;
        vx = q[2 * i]; vy = q[2 * i + 1];
        if ((v0y <= py && vy <= py) || (v0y > py && vy > py) || (v0x < px && vx < px)) {
            if (py == vy && (px == vx || (py == v0y && ((v0x <= px && px <= vx) || (vx <= px && px <= v0x))))) return 0;
            continue;
        }
        double dist = (double)(py - v0y) * (vx - v0x) - (double)(px - v0x) * (vy - v0y);
        if (dist == 0) return 0;
        if (vy < v0y) dist = -dist;
        counter += dist > 0;
    }
    return counter % 2 == 0 ? -1 : 1;
}

// ---- cv::cornerSubPix (imgproc/cornersubpix.cpp, 3.2.0) for one corner, one lane -------------------------------------
// Optional stage (DetectorParameters::doCornerRefinement, off in the reference).  The (win+2)^2 patch that OpenCV samples
// into a buffer with getRectSubPix is evaluated on the fly, element by element with the same float expressions (interior
// fast path: dst[j] = prev + t, prev = (float)(t * (1-a)/a); clipped path: four-tap bilinear with replicated borders), and
// the normal equations are accumulated in the same order, so the result is bit-identical to the scalar restatement.
struct SubPix {
    const uint8_t* gray;
    int rows, cols, ipx, ipy;
    bool interior;
    float a, b, a11, a12, a21, a22, b1, b2;
    double s;
};
__device__ __forceinline__ float subpix_at(const SubPix& P, int i, int j) {         // element (row i, column j) of the patch
    if (P.interior) {
        const uint8_t* p = P.gray + (size_t)(P.ipy + i) * P.cols + P.ipx;
        const float t = P.a12 * p[j + 1] + P.a22 * p[j + 1 + P.cols];
        float prev;
        if (j == 0) prev = (1 - P.a) * (P.b1 * p[0] + P.b2 * p[P.cols]);
        else { const float tp = P.a12 * p[j] + P.a22 * p[j + P.cols]; prev = (float)(tp * P.s); }
        return prev + t;
    }
    const int y0 = min(max(P.ipy + i, 0), P.rows - 1), y1 = min(max(P.ipy + i + 1, 0), P.rows - 1);
    const int x0 = min(max(P.ipx + j, 0), P.cols - 1), x1 = min(max(P.ipx + j + 1, 0), P.cols - 1);
    return P.gray[(size_t)y0 * P.cols + x0] * P.a11 + P.gray[(size_t)y0 * P.cols + x1] * P.a12 + P.gray[(size_t)y1 * P.cols + x0] * P.a21 +
           P.gray[(size_t)y1 * P.cols + x1] * P.a22;
}
__device__ void corner_sub_pix_one(const uint8_t* gray, int rows, int cols, const float* __restrict__ mask, int win, int max_iters,
                                   double eps2, float& x, float& y) {
    const int win_w = 2 * win + 1, pw = win_w + 2;
    const float cTx = x, cTy = y;
    float cIx = x, cIy = y;
    int iter = 0;
    double err = 0;
    do {
        SubPix P;
        P.gray = gray; P.rows = rows; P.cols = cols;
        const float centerx = cIx - (pw - 1) * 0.5f, centery = cIy - (pw - 1) * 0.5f;
        P.ipx = (int)floorf(centerx); P.ipy = (int)floorf(centery);
        P.interior = 0 <= P.ipx && P.ipx + pw < cols && 0 <= P.ipy && P.ipy + pw < rows;
        P.a = centerx - P.ipx; P.b = centery - P.ipy;
        if (P.interior) {
            P.a = P.a > 0.0001f ? P.a : 0.0001f;
            P.a12 = P.a * (1.f - P.b); P.a22 = P.a * P.b; P.b1 = 1.f - P.b; P.b2 = P.b;
            P.s = (1. - P.a) / P.a;
            P.a11 = P.a21 = 0.f;
        } else {
            P.a11 = (1.f - P.a) * (1.f - P.b); P.a12 = P.a * (1.f - P.b); P.a21 = (1.f - P.a) * P.b; P.a22 = P.a * P.b;
            P.b1 = P.b2 = 0.f; P.s = 0.0;
        }
        double a = 0, b = 0, c = 0, bb1 = 0, bb2 = 0;
        for (int i = 0, k = 0; i < win_w; i++) {
            const double py = i - win;
            for (int j = 0; j < win_w; j++, k++) {
                const double m = mask[k];
                const double tgx = subpix_at(P, i + 1, j + 2) - subpix_at(P, i + 1, j);
                const double tgy = subpix_at(P, i + 2, j + 1) - subpix_at(P, i, j + 1);
                const double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m;
                const double px = j - win;
                a += gxx; b += gxy; c += gyy;
                bb1 += gxx * px + gxy * py;
                bb2 += gxy * px + gyy * py;
            }
        }
        const double det = a * c - b * b;
        if (fabs(det) <= DBL_EPSILON * DBL_EPSILON) break;
        const double scale = 1.0 / det;
        const float nx = (float)(cIx + c * scale * bb1 - b * scale * bb2);
        const float ny = (float)(cIy - b * scale * bb1 + a * scale * bb2);
        err = (nx - cIx) * (nx - cIx) + (ny - cIy) * (ny - cIy);
        cIx = nx; cIy = ny;
        if (cIx < 0 || cIx >= cols || cIy < 0 || cIy >= rows) break;
    } while (++iter < max_iters && err > eps2);
    if (fabsf(cIx - cTx) > win || fabsf(cIy - cTy) > win) { cIx = cTx; cIy = cTy; }      // poor convergence: keep the initial point
    x = cIx; y = cIy;
}

__global__ __launch_bounds__(128) void k_pose(const FinalCand* __restrict__ finals, const unsigned* __restrict__ n_final,
                                              Marker* __restrict__ markers, unsigned* __restrict__ n_markers,
                                              ObsRaw* __restrict__ obs, CamParams cam, SlamParams sp, Counters* ctr, RefineCfg rf) {
    __shared__ float sC[kMarkerMax][8];
    __shared__ int sId[kMarkerMax];
    __shared__ unsigned char sRem[kMarkerMax];
    __shared__ int sOut[kMarkerMax];
    __shared__ int sN, sM;
    const int tid = threadIdx.x;
    const int f = blockIdx.x;
    const FinalCand* fin = finals + (size_t)f * kCandMax;
    const int nF = (int)min(n_final[f], (unsigned)kCandMax);

    // the identified candidates, in candidate order (an ordered compaction: ballot ranks inside a wave, the first wave's count
    // for the second - one thread walking the list would pay one dependent global load per candidate)
    __shared__ int sCnt[2];
    int kbase = 0;
    for (int i0 = 0; i0 < nF; i0 += 128) {                     // uniform
        const int i = i0 + tid;
        const int id = i < nF ? fin[i].id : -1;
        const unsigned long long has = __ballot(id >= 0);
        if ((tid & 63) == 0) sCnt[tid >> 6] = __popcll(has);
        __syncthreads();
        const int k = kbase + (tid >= 64 ? sCnt[0] : 0) + __popcll(has & ((1ull << (tid & 63)) - 1ull));
        if (id >= 0) {
            if (k < kMarkerMax) {
                sId[k] = id;
                // _identifyOneCandidate: std::rotate(begin, begin + 4 - rot, end) -> new[j] = old[(j + 4 - rot) % 4]
                const int rot = fin[i].pad[0];
                for (int j = 0; j < 4; j++) {
                    const int sidx = (j + 4 - rot) & 3;
                    sC[k][2 * j] = fin[i].c[2 * sidx];
                    sC[k][2 * j + 1] = fin[i].c[2 * sidx + 1];
                }
            } else {
                atomicOr(&ctr->overflow, (unsigned)kOvfMarkers);
            }
        }
        kbase += sCnt[0] + sCnt[1];
        __syncthreads();                                        // before the next chunk's counts
    }
    if (tid == 0) sN = min(kbase, kMarkerMax);
    for (int i = tid; i < kMarkerMax; i += 128) sRem[i] = 0;
    __syncthreads();
    const int n = sN;
    // _filterDetectedMarkers: same id and one quad inside the other -> drop the inner one
    for (int p = tid; p < n * n; p += 128) {
        int i = p / n, j = p - i * n;
        if (j > i && sId[i] == sId[j]) {
            bool inside = true;
            for (int q = 0; q < 4 && inside; q++)
                if (point_in_quad(sC[i], sC[j][2 * q], sC[j][2 * q + 1]) < 0) inside = false;
            if (inside) {
                sRem[j] = 1;
            } else {
                inside = true;
                for (int q = 0; q < 4 && inside; q++)
                    if (point_in_quad(sC[j], sC[i][2 * q], sC[i][2 * q + 1]) < 0) inside = false;
                if (inside) sRem[i] = 1;
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        int k = 0;
        for (int i = 0; i < n; i++) if (!sRem[i]) sOut[k++] = i;
        sM = k;
        n_markers[f] = (unsigned)k;
    }
    __syncthreads();
    const int M = sM;
    if (rf.on) {                                               // uniform: cornerSubPix on the surviving markers, one lane per corner
        const uint8_t* gray = rf.gray + (size_t)f * rf.rows * rf.cols;
        for (int e = tid; e < 4 * M; e += 128) {
            const int i = sOut[e >> 2], q = e & 3;
            corner_sub_pix_one(gray, rf.rows, rf.cols, rf.mask, rf.win, rf.max_iters, rf.eps2, sC[i][2 * q], sC[i][2 * q + 1]);
        }
        __syncthreads();
    }
    for (int k = tid; k < M; k += 128) {
        const int i = sOut[k];
        Marker mk;
        mk.id = sId[i];
        mk.pad = 0;
        for (int j = 0; j < 8; j++) mk.c[j] = sC[i][j];
        solve_marker_pose(mk.c, (float)sp.marker_length, cam, mk.rvec, mk.tvec);
        markers[(size_t)f * kMarkerMax + k] = mk;

        // getObservations loop body (aruco_slam.cpp:325-369)
        ObsRaw o;
        o.id = mk.id;
        o.valid = 1;
        const double* t = mk.tvec;
        const double nt = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
        float dist = (float)nt;
        if (dist > sp.useful_distance_threshold) o.valid = 0;                    // aruco_slam.cpp:327-333
        double R[9];
        rodrigues_fwd(mk.rvec, R, nullptr);
        o.x = t[2] + sp.r2c_tx;                                                  // aruco_slam.cpp:359
        o.y = -t[0] + sp.r2c_ty;                                                 // aruco_slam.cpp:360
        o.th = atan2(-R[2], R[8]);                                               // aruco_slam.cpp:361
        wrap_once(o.th);
        // CalculateCovariance (aruco_slam.cpp:437-471): reprojection error of the 4 corners (points stored as float)
        double par[6] = {mk.rvec[0], mk.rvec[1], mk.rvec[2], t[0], t[1], t[2]};
        double proj[8];
        const double hl = (double)(float)(sp.marker_length / 2.f);
        project4(par, hl, cam, proj, nullptr);
        double total = 0;
        for (int q = 0; q < 4; q++) {
            double dx = (double)mk.c[2 * q] - (double)(float)proj[2 * q], dy = (double)mk.c[2 * q + 1] - (double)(float)proj[2 * q + 1];
            double e = sqrt(dx * dx + dy * dy);
            total += e * e;
        }
        double rms = total / 4.0;
        double ddx = (double)mk.c[0] - (double)mk.c[4], ddy = (double)mk.c[1] - (double)mk.c[5];
        double object_error = (rms / sqrt(ddx * ddx + ddy * ddy)) * (nt / sp.marker_length);
        o.r[0] = object_error * sp.R_x + 1e-2;
        o.r[1] = object_error * sp.R_y + 1e-2;
        o.r[2] = object_error * sp.R_theta + 1e-3;
        if (sqrt(o.r[0] * o.r[0] + o.r[1] * o.r[1] + o.r[2] * o.r[2]) > 1) o.valid = 0;   // aruco_slam.cpp:367
        obs[(size_t)f * kMarkerMax + k] = o;
    }
}

void launch_pose(hipStream_t st, int nframes, const FinalCand* finals, const unsigned* n_final, Marker* markers,
                 unsigned* n_markers, ObsRaw* obs, const CamParams& cam, const SlamParams& sp, Counters* ctr, const RefineCfg& rf) {
    hipLaunchKernelGGL(k_pose, dim3(nframes), dim3(128), 0, st, finals, n_final, markers, n_markers, obs, cam, sp, ctr, rf);
}

} // namespace aslam
