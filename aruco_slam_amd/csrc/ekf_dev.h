// Small device helpers shared by the EKF kernels (ekf.hip, ekf_window.hip).
#pragma once
#include "common.h"
#include <cmath>

// hardware reciprocal estimate (v_rcp_f64); the CPU emulation of the tests defines it as a plain division
#ifndef ASLAM_RCP_ESTIMATE
#define ASLAM_RCP_ESTIMATE(x) __builtin_amdgcn_rcp(x)
#endif

namespace aslam {

// value of lane `src` (wave-uniform) in every lane, through two v_readlane - no LDS traffic, unlike __shfl's ds_bpermute
#ifndef ASLAM_WAVE_BCAST
#define ASLAM_WAVE_BCAST(v, src) aslam_wave_bcast(v, src)
__device__ __forceinline__ double aslam_wave_bcast(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
#endif
__device__ __forceinline__ double aslam_rcp_estimate(double x) { return ASLAM_RCP_ESTIMATE(x); }

typedef double v4d __attribute__((vector_size(4 * sizeof(double))));   // accumulator of v_mfma_f64_16x16x4_f64

__device__ __forceinline__ void wrap1(double& a) {      // ArucoSlam::normAngle (aruco_slam.cpp:412-421): wraps once
    const double PI = 3.14159265358979323846;
    if (a >= PI) a -= 2.0 * PI;
    if (a < -PI) a += 2.0 * PI;
}

__device__ __forceinline__ void inv3_reg(const double* P, double* o) {   // 3x3 inverse (cofactors), row-major
    const double a = P[0], b = P[1], c = P[2], d = P[3], e = P[4], f = P[5], g = P[6], h = P[7], i = P[8];
    const double A = e * i - f * h, B = f * g - d * i, C = d * h - e * g;
    const double id = 1.0 / (a * A + b * B + c * C);
    o[0] = A * id; o[1] = (c * h - b * i) * id; o[2] = (b * f - c * e) * id;
    o[3] = B * id; o[4] = (a * i - c * g) * id; o[5] = (c * d - a * f) * id;
    o[6] = C * id; o[7] = (b * g - a * h) * id; o[8] = (a * e - b * d) * id;
}
// the same with the reciprocal of the determinant by Newton's iteration on a hardware estimate (full double precision to a
// couple of ulps; the IEEE division sequence is four times as long and sits on the Gauss-Jordan sweep's critical path)
__device__ __forceinline__ void inv3_fast(const double* P, double* o) {
    const double a = P[0], b = P[1], c = P[2], d = P[3], e = P[4], f = P[5], g = P[6], h = P[7], i = P[8];
    const double A = fma(e, i, -f * h), B = fma(f, g, -d * i), C = fma(d, h, -e * g);
    const double det = fma(a, A, fma(b, B, c * C));
    double id = aslam_rcp_estimate(det);
    id = fma(fma(-det, id, 1.0), id, id);
    id = fma(fma(-det, id, 1.0), id, id);
    o[0] = A * id; o[1] = fma(c, h, -b * i) * id; o[2] = fma(b, f, -c * e) * id;
    o[3] = B * id; o[4] = fma(a, i, -c * g) * id; o[5] = fma(c, d, -a * f) * id;
    o[6] = C * id; o[7] = fma(b, g, -a * h) * id; o[8] = fma(a, e, -b * d) * id;
}
__device__ __forceinline__ void mul3(const double* X, const double* Y, double* Z) {   // Z = X * Y (3x3 row-major)
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Z[i * 3 + j] = X[i * 3] * Y[j] + X[i * 3 + 1] * Y[3 + j] + X[i * 3 + 2] * Y[6 + j];
}

} // namespace aslam
