#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "../../aruco_slam_amd/csrc/ekf_dev.h"
using namespace aslam;
constexpr int WS = 66, WIMG = 64 * WS;
__device__ __forceinline__ int gpix(int r, int c) { return c * WS + (r & 48) + 4 * (r & 3) + ((r >> 2) & 3); }
__device__ __forceinline__ void gj_publish_rows(const v4d (&ga)[4], int p, int gw, int lk, int li, double* rows) {
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const int r = p + q;
        if ((r >> 4) != gw) continue;
        const int reg = (r >> 2) & 3;
        const bool mine = lk == (r & 3);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const double v = reg == 0 ? ga[t][0] : reg == 1 ? ga[t][1] : reg == 2 ? ga[t][2] : ga[t][3];
            if (mine) rows[q * 64 + 16 * t + li] = v;
        }
    }
}
// FLAGS: 1 = skip inverse math, 2 = skip publish, 4 = skip mfma, 8 = skip prep entirely, 16 = no barriers
template <int R16> __device__ __forceinline__ void gj_publish_static(const v4d (&ga)[4], int p, int gw, int lk, int li, double* rows) {
#pragma unroll
    for (int q = 0; q < 3; q++) {
        constexpr int dummy = 0;
        const int rl = (R16 + q) & 15;
        if (gw == ((p + q) >> 4) && lk == (rl & 3)) {
            double* d = rows + q * 64 + li;
            d[0] = ga[0][rl >> 2]; d[16] = ga[1][rl >> 2]; d[32] = ga[2][rl >> 2]; d[48] = ga[3][rl >> 2];
        }
    }
}
__device__ __forceinline__ void gj_publish_dispatch(const v4d (&ga)[4], int p, int gw, int lk, int li, double* rows) {
    switch (p & 15) {
#define C(i) case i: gj_publish_static<i>(ga, p, gw, lk, li, rows); break;
    C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15)
#undef C
    }
}
template <int F>
__global__ __launch_bounds__(512) void k(const double* A, double* G, long long* out, int m, int reps) {
    __shared__ __align__(16) double sG[WIMG];
    __shared__ double sGY[2][4][WS];
    __shared__ double sRow[2][4][64];
    __shared__ double sPub[2][3][64];
    __shared__ double sZe[64], sNu[64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int n3 = 3 * m;
    for (int e = tid; e < WIMG; e += 512) sG[e] = 0.0;
    if (tid < 64) { sGY[0][3][tid] = 0; sGY[1][3][tid] = 0; sRow[0][3][tid] = 0; sRow[1][3][tid] = 0; sZe[tid] = 0.01 * tid; }
    __syncthreads();
    long long t0 = 0, t1 = 0;
    for (int rep = 0; rep < reps; rep++) {
    for (int e = tid; e < 64 * 64; e += 512) sG[gpix(e >> 6, e & 63)] = A[e];
    __syncthreads();
    if (rep == reps - 1) t0 = clock64();
        const int gw = wave & 3;
        const int grow = 16 * gw + li;
        double nu = (wave == 4 && lane < n3) ? sZe[lane] : 0.0;
        const double ze = nu;
        double Rp0 = 0, Rp1 = 0, Rp2 = 0, Yp0 = 0, Yp1 = 0, Yp2 = 0;      // prep wave: rows and Y~ of the previous pivot
        v4d ga[4];
        if (wave < 4) {
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const double2 lo = *reinterpret_cast<const double2*>(&sG[(16 * t + li) * WS + 16 * gw + 4 * lk]);
                const double2 hi = *reinterpret_cast<const double2*>(&sG[(16 * t + li) * WS + 16 * gw + 4 * lk + 2]);
                ga[t][0] = lo.x; ga[t][1] = lo.y; ga[t][2] = hi.x; ga[t][3] = hi.y;
            }
            gj_publish_rows(ga, 0, gw, lk, li, &sPub[0][0][0]);
            if (m > 1) gj_publish_rows(ga, 3, gw, lk, li, &sPub[1][0][0]);
        }
        ASLAM_LDS_BARRIER();
        for (int j = -1; j < m; j++) {
            // phase j: workers apply step j and publish the rows of pivot j + 2; wave 4 prepares pivot j + 1
            long long ta = 0, tb = 0, tc = 0, td = 0;
            if (j == 5 && rep == reps - 1) ta = clock64();
            if (wave == 4) {
                if (j + 1 < m && !((F & 8) && j >= 0)) {
                    const int jb = (j + 1) & 1, p = 3 * (j + 1);
                    double R0 = sPub[jb][0][lane], R1 = sPub[jb][1][lane], R2 = sPub[jb][2][lane];     // rows of pivot j + 1 as of step j - 1
                    if (j >= 0) {
                        // step j's rank-3 correction of these rows: C~_j[p + q][k] = R_j[k][p + q] (symmetry), Y~_j from the registers
#pragma unroll
                        for (int q = 0; q < 3; q++) {
                            const double c0 = ASLAM_WAVE_BCAST(Rp0, p + q), c1 = ASLAM_WAVE_BCAST(Rp1, p + q), c2 = ASLAM_WAVE_BCAST(Rp2, p + q);
                            double& R = q == 0 ? R0 : q == 1 ? R1 : R2;
                            R = fma(-c2, Yp2, fma(-c1, Yp1, fma(-c0, Yp0, R)));
                        }
                    }
                    double Sm[9], Si[9];
#pragma unroll
                    for (int c = 0; c < 3; c++) { Sm[c] = ASLAM_WAVE_BCAST(R0, p + c); Sm[3 + c] = ASLAM_WAVE_BCAST(R1, p + c); Sm[6 + c] = ASLAM_WAVE_BCAST(R2, p + c); }
                    if (F & 1) { for (int q = 0; q < 9; q++) Si[q] = Sm[q] * 0.01; } else inv3_fast(Sm, Si);
                    const double r0 = R0 + (lane == p ? 1.0 : 0.0), r1 = R1 + (lane == p + 1 ? 1.0 : 0.0), r2 = R2 + (lane == p + 2 ? 1.0 : 0.0);
                    Yp0 = fma(Si[2], r2, fma(Si[1], r1, Si[0] * r0));
                    Yp1 = fma(Si[5], r2, fma(Si[4], r1, Si[3] * r0));
                    Yp2 = fma(Si[8], r2, fma(Si[7], r1, Si[6] * r0));
                    sGY[jb][0][lane] = Yp0; sGY[jb][1][lane] = Yp1; sGY[jb][2][lane] = Yp2;
                    sRow[jb][0][lane] = R0; sRow[jb][1][lane] = R1; sRow[jb][2][lane] = R2;
                    Rp0 = R0; Rp1 = R1; Rp2 = R2;
                    const double z0 = ASLAM_WAVE_BCAST(ze, p), z1 = ASLAM_WAVE_BCAST(ze, p + 1), z2 = ASLAM_WAVE_BCAST(ze, p + 2);
                    const double u0 = fma(Si[2], z2, fma(Si[1], z1, Si[0] * z0)), u1 = fma(Si[5], z2, fma(Si[4], z1, Si[3] * z0)), u2 = fma(Si[8], z2, fma(Si[7], z1, Si[6] * z0));
                    if (lane >= p + 3 && lane < n3) nu += R0 * u0 + R1 * u1 + R2 * u2;
                }
            } else if (wave < 4 && j >= 0) {
                const int cb = j & 1, p0 = 3 * j;
                const double rr = sRow[cb][lk][grow];
                double af = grow < p0 ? rr : -rr;
                if (lk < 3 && grow == p0 + lk) af += 1.0;
#pragma unroll
                for (int t = 0; t < 4; t++) ga[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, sGY[cb][lk][16 * t + li], ga[t], 0, 0, 0);
                if (j == 5 && rep == reps - 1) tc = clock64();
                if (j + 2 < m && !(F & 2)) gj_publish_dispatch(ga, p0 + 6, gw, lk, li, &sPub[cb][0][0]);
            }
            if (j == 5 && rep == reps - 1) tb = clock64();
            ASLAM_LDS_BARRIER();
            if (j == 5 && rep == reps - 1 && lane == 0) { td = clock64(); out[8 + wave * 4] = tb - ta; out[9 + wave * 4] = td - tb; out[10 + wave * 4] = tc - ta; }
        }
        if (wave < 4) {
#pragma unroll
            for (int t = 0; t < 4; t++) {
                double2 lo, hi;
                lo.x = ga[t][0]; lo.y = ga[t][1]; hi.x = ga[t][2]; hi.y = ga[t][3];
                *reinterpret_cast<double2*>(&sG[(16 * t + li) * WS + 16 * gw + 4 * lk]) = lo;
                *reinterpret_cast<double2*>(&sG[(16 * t + li) * WS + 16 * gw + 4 * lk + 2]) = hi;
            }
        } else if (wave == 4) sNu[lane] = nu;
        ASLAM_LDS_BARRIER();
    if (rep == reps - 1) t1 = clock64();
    }
    for (int e = tid; e < 64 * 64; e += 512) G[e] = sG[gpix(e >> 6, e & 63)];
    if (tid == 0) out[0] = t1 - t0;
}
template <int F> void run(const double* dA, double* dG, long long* dO, int m, const std::vector<double>& A) {
    hipLaunchKernelGGL(k<F>, dim3(1), dim3(512), 0, 0, dA, dG, dO, m, 3);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, dO, 8, hipMemcpyDeviceToHost);
    long long o[48]; hipMemcpy(o, dO, sizeof o, hipMemcpyDeviceToHost);
    for (int w = 0; w < 5; w++) printf("  wave %d: work %lld (mfma issued at %lld) barrier wait %lld\n", w, o[8 + w * 4], o[10 + w * 4], o[9 + w * 4]);
    std::vector<double> G(4096); hipMemcpy(G.data(), dG, 4096 * 8, hipMemcpyDeviceToHost);
    // check G A = I
    double err = 0; int n = 3 * m;
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) { double s = 0; for (int q = 0; q < n; q++) s += G[i * 64 + q] * A[q * 64 + j]; err = fmax(err, fabs(s - (i == j))); }
    printf("flags %2d: %lld cycles total, %.0f per step, |GA-I| %.2e\n", F, c, (double)c / m, err);
}
int main() {
    int m = 20, n = 60;
    std::vector<double> B(64 * 64, 0), A(64 * 64, 0);
    srand(1);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) B[i * 64 + j] = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) { double s = 0; for (int q = 0; q < n; q++) s += B[i * 64 + q] * B[j * 64 + q]; A[i * 64 + j] = s + (i == j ? 5.0 : 0); }
    double *dA, *dG; long long* dO; hipMalloc(&dA, 4096 * 8); hipMalloc(&dG, 4096 * 8); hipMalloc(&dO, 1024);
    hipMemcpy(dA, A.data(), 4096 * 8, hipMemcpyHostToDevice);
    run<0>(dA, dG, dO, m, A); run<8>(dA, dG, dO, m, A);
    return 0;
}
