// Windowed EKF: a run of K consecutive frames that fuse the SAME m landmarks (no new landmark, no "stationary" no-op,
// m <= kWinM) is processed without streaming the N x N covariance once per frame.
//
// Split the state into S (robot pose + the m observed landmarks, s = 3 + 3m <= 63) and R (everything else):
//     Sigma = [ P  Y ]      P = Sigma[S,S]   Y = Sigma[S,R]
//             [ X  Z ]      X = Sigma[R,S]   Z = Sigma[R,R]
// Every predict (aruco_slam.cpp:21-74: D = blkdiag(H3, I) on S, process noise on the pose block) and every correction
// (aruco_slam.cpp:108-207 fused per frame: H = [H_S 0], G = (H_S P' H_S^T + R)^-1, W = P' H_S^T, V = H_S P') touches R only
// through X and Y, linearly:
//     P+ = P' - W G V                     Y+ = (D Y) - W G (H_S D Y)                 X+ = (X D^T) - (X D^T H_S^T) G V
//     Z+ = Z - (X D^T H_S^T) G (H_S D Y)  mu_R+ = mu_R + (X D^T H_S^T) g
// hence, over the whole window, with s x s accumulators Lambda, Gamma, Psi and a vector psi
//     Y_K = Lambda Y_0      X_K = X_0 Gamma      Z_K = Z_0 - X_0 Psi Y_0      mu_R,K = mu_R,0 + X_0 psi
//     Lambda+ = D Lambda - W G B       Gamma+ = Gamma D^T - A G V       Psi+ = Psi + A G B      psi+ = psi + A g
//     with  B = H_S D Lambda  (3m x s),   A = Gamma D^T H_S^T  (s x 3m).
// This is the reference's arithmetic regrouped: the same identity that fuses one frame's M corrections (ekf.hip) applied across
// frames.  Nothing is approximated; the one property used beyond the identity is that the S x S block P is symmetric to
// rounding (it is in the reference: aruco_slam.cpp:73, 204 keep Sigma = Sigma^T up to the last bit or two), so that inside the
// chain kernel V = H_S P' is read as W^T and the innovation matrix A = H_S W + R is treated as symmetric.  X / Gamma and
// Y / Lambda are NOT assumed to be transposes of each other.  Three kernels per window:
//   k_ekf_win_chain   one workgroup runs the K frames on P and mu_S held in LDS (predict, records, innovation matrix,
//                     block Gauss-Jordan, P update on the f64 matrix cores) and logs G, W, g, H3 and the Jacobians of every
//                     frame; further workgroups copy X_0 (columns S of Sigma) and Y_0 (rows S) aside meanwhile;
//   k_ekf_win_scan    4 x 4 workgroups replay the log: workgroup (i, j) carries 16 columns of Lambda, 16 rows of Gamma and
//                     the 16 x 16 block of Psi they determine (columns of Lambda and rows of Gamma evolve independently);
//   k_ekf_win_flush   one workgroup per 64 x 64 tile of Sigma: Z tile -= X_0 (Psi Y_0), rows / columns of S replaced by
//                     Lambda Y_0 / X_0 Gamma / P_K, mu_R += X_0 psi - the ONE pass over Sigma per window.
#include "common.h"
#include "ekf.h"
#include "ekf_dev.h"
#include <cmath>

namespace aslam {

constexpr int WS = 66;                        // row stride (doubles) of a 64 x 64 image: conflict-free MFMA A-operand reads from LDS
constexpr int WIMG = 64 * WS;                 // doubles per image
constexpr int WLOG_G = 0, WLOG_W = WIMG, WLOG_g = 2 * WIMG, WLOG_H3 = WLOG_g + 64, WLOG_HREC = WLOG_H3 + 16;
constexpr int WLOG_STRIDE = WLOG_HREC + kWinM * 18;     // doubles per logged frame
// layout of d_win_small: TWO sets of the scan's accumulators (Lambda | Gamma | Psi | psi), then P.  A continuation piece of the scan
// reads the set the previous piece wrote and writes the other one: its 4 x 4 workgroups share Lambda's column blocks, Gamma's row
// blocks and psi, and a workgroup that finishes early must not overwrite what a workgroup that starts late still has to read.
constexpr int WSM_SET = 3 * WIMG + 64;
constexpr int WSM_LAM = 0, WSM_GAM = WIMG, WSM_PSI = 2 * WIMG, WSM_psi = 3 * WIMG, WSM_P = 2 * WSM_SET;
constexpr int WCT = 512;                      // threads of the chain workgroup

// development aid (make FLAGS+=-DASLAM_WIN_STAMPS): cycle stamps of the chain's phases, printed for one frame
#ifdef ASLAM_WIN_STAMPS
#define WIN_STAMP(i) do { if (tid == 0 && k == 3) stamps[i] = clock64(); } while (0)
#ifdef ASLAM_GJ_STAMPS
#define GJ_STAMP(i) do { if ((tid & 63) == 0 && k == 3 && j == 5) gst[i] = clock64(); } while (0)
#else
#define GJ_STAMP(i) do { } while (0)
#endif
#else
#define WIN_STAMP(i) do { } while (0)
#define GJ_STAMP(i) do { } while (0)
#endif

size_t ekf_win_log_doubles() { return (size_t)WLOG_STRIDE * kWinFrames + 512; }   // + slack: the scan stages whole 16-byte x 256-thread passes
size_t ekf_win_small_doubles() { return (size_t)2 * WSM_SET + WIMG; }

// Gauss-Jordan image layout: element (r, c) of the 64 x 64 image.  Column-major with the rows of every 16-row tile regrouped so
// that the four rows one lane holds of an MFMA accumulator tile (r = 16 g + lk + 4 reg) are neighbours: 16-byte LDS accesses.
__device__ __forceinline__ int gpix(int r, int c) { return c * WS + (r & 48) + 4 * (r & 3) + ((r >> 2) & 3); }

// rows p .. p + 2 of the image leave the accumulators of the working wave(s) that own them (row r = 16 g + lk + 4 reg of tile
// row g): lanes with lk == r & 3 write their register reg = (r >> 2) & 3 of each of the four column tiles.  The register number
// must be static (a select chain over the accumulators costs more than the rest of the step): one instance per p mod 16.
template <int P16> __device__ __forceinline__ void gj_publish_rows_at(const v4d (&ga)[4], int p, int gw, int lk, int li, double* rows) {
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const int rl = (P16 + q) & 15;
        if (gw == ((p + q) >> 4) && lk == (rl & 3)) {
            double* d = rows + q * 64 + li;
            d[0] = ga[0][rl >> 2]; d[16] = ga[1][rl >> 2]; d[32] = ga[2][rl >> 2]; d[48] = ga[3][rl >> 2];
        }
    }
}
__device__ __forceinline__ void gj_publish_rows(const v4d (&ga)[4], int p, int gw, int lk, int li, double* rows) {
    switch (p & 15) {
#define ASLAM_GJ_CASE(i) case i: gj_publish_rows_at<i>(ga, p, gw, lk, li, rows); break;
        ASLAM_GJ_CASE(0) ASLAM_GJ_CASE(1) ASLAM_GJ_CASE(2) ASLAM_GJ_CASE(3) ASLAM_GJ_CASE(4) ASLAM_GJ_CASE(5) ASLAM_GJ_CASE(6) ASLAM_GJ_CASE(7)
        ASLAM_GJ_CASE(8) ASLAM_GJ_CASE(9) ASLAM_GJ_CASE(10) ASLAM_GJ_CASE(11) ASLAM_GJ_CASE(12) ASLAM_GJ_CASE(13) ASLAM_GJ_CASE(14) ASLAM_GJ_CASE(15)
#undef ASLAM_GJ_CASE
    }
}

__device__ __forceinline__ int win_state_index(const WinDesc& wd, int p) {      // state offset of position p of S
    return p < 3 ? p : wd.li[(p - 3) / 3] + (p - 3) % 3;
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WCT) void k_ekf_win_chain(EkfState E, SlamParams sp, WinDesc wd, const ObsRaw* __restrict__ obs,
                                                      const unsigned* __restrict__ n_markers, const double* __restrict__ enc,
                                                      const unsigned char* __restrict__ obs_idx) {
    __shared__ __align__(16) double sP[WIMG];
    __shared__ __align__(16) double sW[WIMG];
    __shared__ __align__(16) double sV[WIMG];          // J = G V
    __shared__ __align__(16) double sG[WIMG];
    __shared__ double sMu[64], sZe[64], sNu[64];
    __shared__ double sHr[kWinM][9], sHl[kWinM][9], sRd[kWinM][3];
    __shared__ double sH3[9], sQ[9], sPose[5];
    __shared__ double sGY[2][4][WS];                   // Gauss-Jordan: a step's B operand Y~ (3 x 64, 4th depth row zero), double buffered
    __shared__ double sRow[2][4][64];                  // ... and its pivot rows (the column operand by symmetry; 4th row zero)
    __shared__ double sPub[2][3][64];                  // rows of the pivot after next, as published by the workers that own them
    __shared__ int sS[64];
    const int tid = threadIdx.x;
    const int m = wd.m, s = wd.s, n3 = 3 * m;
    const int ld = E.ld;

    if (blockIdx.x > 0) {
        if (wd.cont) return;                                       // a run's later chains: X_0, Y_0 are those of its first one
        // ---- X_0^T (row p = column S_p of Sigma) -> d_Wt, Y_0 (row p = row S_p of Sigma) -> d_V, S-position table ----
        const int N = 3 + 3 * (*E.d_L);
        for (int t = (blockIdx.x - 1) * WCT + tid; t < N; t += (gridDim.x - 1) * WCT) {
            int pos = -1;
            if (t < 3) pos = t;
            else {
                const int base = (t - 3) / 3 * 3 + 3;
                for (int a = 0; a < m; a++) if (wd.li[a] == base) pos = 3 + 3 * a + (t - base);
            }
            E.d_win_sidx[t] = pos;
            for (int p = 0; p < s; p++) {
                const int Sp = win_state_index(wd, p);
                E.d_Wt[(size_t)p * ld + t] = E.d_sigma[(size_t)Sp * ld + t];
                E.d_V[(size_t)p * ld + t] = E.d_sigma[(size_t)t * ld + Sp];
            }
        }
        return;
    }

    // ---- workgroup 0: P = Sigma[S,S] and mu_S into LDS (zero padded to 64) ----
    if (tid < 64) {
        sS[tid] = tid < s ? win_state_index(wd, tid) : 0; sMu[tid] = 0.0; sZe[tid] = 0.0; sNu[tid] = 0.0;
        sGY[0][3][tid] = 0.0; sGY[1][3][tid] = 0.0; sRow[0][3][tid] = 0.0; sRow[1][3][tid] = 0.0;
    }
    for (int e = tid; e < WIMG; e += WCT) { sP[e] = 0.0; sW[e] = 0.0; sV[e] = 0.0; sG[e] = 0.0; }
    __syncthreads();
    if (tid < s) sMu[tid] = E.d_mu[sS[tid]];
    if (wd.cont) {                                                 // the run goes on: P as the previous chain left it
        for (int e = tid; e < WIMG; e += WCT) sP[e] = E.d_win_small[WSM_P + e];
    } else {
        for (int e = tid; e < s * s; e += WCT) {
            const int q = e / s, p = e - q * s;                    // column q, row p: consecutive threads walk down a column
            sP[p * WS + q] = E.d_sigma[(size_t)sS[q] * ld + sS[p]];
        }
    }
    const int wave = tid >> 6, lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int bj = tid % m, bi = tid / m;                          // Gauss-Jordan block owned by this thread
    const bool act = bi < m;
    // prefetch of the first frame's inputs
    ObsRaw myObs{};
    if (tid < m) myObs = obs[(size_t)wd.first_slot * kMarkerMax + obs_idx[tid]];
    double e_wl = 0, e_wr = 0, e_dt = 0;
    if (tid == 0) { const double* e = enc + (size_t)3 * wd.first_slot; e_wl = e[0]; e_wr = e[1]; e_dt = e[2]; }
    __syncthreads();

#ifdef ASLAM_WIN_STAMPS
    long long stamps[12] = {0};
    long long gst[10] = {0};
#endif
    for (int k = 0; k < wd.K; k++) {
        const int slot = wd.first_slot + k;
        double* log = E.d_win_log + (size_t)(wd.log0 + k) * WLOG_STRIDE;
        WIN_STAMP(0);
        // ---- 1. predict (aruco_slam.cpp:35-73): pose, H3, Qk ----
        if (tid == 0) {
            const double delta_sl = sp.kl * (e_dt * e_wl), delta_sr = sp.kr * (e_dt * e_wr);
            const double delta_theta = (delta_sr - delta_sl) / (2 * sp.b);
            const double delta_s = 0.5 * (delta_sr + delta_sl);
            const double m0 = sMu[0], m1 = sMu[1], m2 = sMu[2];
            double c, sn;
            sincos(m2 + 0.5 * delta_theta, &sn, &c);
            double th = m2 + delta_theta;
            wrap1(th);
            sMu[0] = m0 + delta_s * c; sMu[1] = m1 + delta_s * sn; sMu[2] = th;
            sH3[0] = 1.0; sH3[1] = 0.0; sH3[2] = -delta_s * sn;
            sH3[3] = 0.0; sH3[4] = 1.0; sH3[5] = delta_s * c;
            sH3[6] = 0.0; sH3[7] = 0.0; sH3[8] = 1.0;
            const double f = 0.5 * sp.kl * e_dt;                    // kl for BOTH wheels (quirk Q7)
            const double wkh[6] = {f * c, f * c, f * sn, f * sn, f * (1 / sp.b), f * (-1 / sp.b)};
            const double su0 = sp.Q_k * fabs(e_wl), su1 = sp.Q_k * fabs(e_wr);
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) sQ[i * 3 + j] = wkh[i * 2] * su0 * wkh[j * 2] + wkh[i * 2 + 1] * su1 * wkh[j * 2 + 1];
            sPose[0] = sMu[0]; sPose[1] = sMu[1]; sPose[2] = th;
            sincos(th, &sPose[3], &sPose[4]);
            for (int i = 0; i < 9; i++) log[WLOG_H3 + i] = sH3[i];
        }
        ASLAM_LDS_BARRIER();
        if (tid < s) {                                              // rows 0..2 <- H3 * rows 0..2 (every column)
            const double a = sP[tid], b = sP[WS + tid], c = sP[2 * WS + tid];
            sP[tid] = sH3[0] * a + sH3[1] * b + sH3[2] * c;
            sP[WS + tid] = sH3[3] * a + sH3[4] * b + sH3[5] * c;
            sP[2 * WS + tid] = sH3[6] * a + sH3[7] * b + sH3[8] * c;
        }
        ASLAM_LDS_BARRIER();
        if (tid < s) {                                              // columns 0..2 <- columns 0..2 * H3^T (every row), + Qk on the pose block
            double* row = sP + tid * WS;
            const double a = row[0], b = row[1], c = row[2];
            double v0 = a * sH3[0] + b * sH3[1] + c * sH3[2], v1 = a * sH3[3] + b * sH3[4] + c * sH3[5], v2 = a * sH3[6] + b * sH3[7] + c * sH3[8];
            if (tid < 3) { v0 += sQ[tid * 3]; v1 += sQ[tid * 3 + 1]; v2 += sQ[tid * 3 + 2]; }
            row[0] = v0; row[1] = v1; row[2] = v2;
        }
        WIN_STAMP(1);
        // ---- 2. records of the m corrections (aruco_slam.cpp:119-143), linearised at the frozen mean ----
        if (tid < m) {
            const int a = tid;
            const double mu0x = sPose[0], mu0y = sPose[1], mu0t = sPose[2], sintheta = sPose[3], costheta = sPose[4];
            const double mx = sMu[3 + 3 * a], my = sMu[4 + 3 * a], mth = sMu[5 + 3 * a];
            double gdx = mx - mu0x, gdy = my - mu0y, gdth = mth - mu0t;
            wrap1(gdth);
            const double zh0 = gdx * costheta + gdy * sintheta, zh1 = -gdx * sintheta + gdy * costheta;
            double z2 = myObs.th - gdth;
            wrap1(z2);
            sZe[3 * a] = myObs.x - zh0; sZe[3 * a + 1] = myObs.y - zh1; sZe[3 * a + 2] = z2;
            sNu[3 * a] = sZe[3 * a]; sNu[3 * a + 1] = sZe[3 * a + 1]; sNu[3 * a + 2] = sZe[3 * a + 2];
            const double G[18] = {-costheta, -sintheta, -gdx * sintheta + gdy * costheta, costheta, sintheta, 0,
                                  sintheta, -costheta, -gdx * costheta - gdy * sintheta, -sintheta, costheta, 0,
                                  0, 0, -1, 0, 0, 1};
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) { sHr[a][r * 3 + c] = G[r * 6 + c]; sHl[a][r * 3 + c] = G[r * 6 + 3 + c]; }
            sRd[a][0] = myObs.r[0]; sRd[a][1] = myObs.r[1]; sRd[a][2] = myObs.r[2];
            for (int q = 0; q < 18; q++) log[WLOG_HREC + a * 18 + q] = G[q];
            if (k == wd.K - 1) {                                    // what the frame leaves behind for whatever follows the window
                PopRec pr;
                pr.id = myObs.id; pr.index = (wd.li[a] - 3) / 3; pr.action = 1; pr.pad = 0;
                pr.z[0] = myObs.x; pr.z[1] = myObs.y; pr.z[2] = myObs.th;
                pr.r[0] = myObs.r[0]; pr.r[1] = myObs.r[1]; pr.r[2] = myObs.r[2];
                E.d_pop[a] = pr;
                LastObs lo;
                lo.id = myObs.id; lo.pad = 0; lo.z[0] = myObs.x; lo.z[1] = myObs.y; lo.z[2] = myObs.th;   // update branch (aruco_slam.cpp:202)
                E.d_last[a] = lo;
            }
        }
        if (tid == 0 && slot < E.max_slots) {
            int* st = E.d_slot_stat + 4 * slot;
            st[0] = (int)min(n_markers[slot], (unsigned)kMarkerMax); st[1] = 0; st[2] = m; st[3] = 0;
        }
        ASLAM_LDS_BARRIER();
        // the next frame's inputs are fetched while this one is solved
        if (k + 1 < wd.K) {
            if (tid < m) myObs = obs[(size_t)(slot + 1) * kMarkerMax + obs_idx[(size_t)(k + 1) * kWinM + tid]];
            if (tid == 0) { const double* e = enc + (size_t)3 * (slot + 1); e_wl = e[0]; e_wr = e[1]; e_dt = e[2]; }
        }
        WIN_STAMP(2);
        // ---- 3. W = P' H^T (s x 3m); 3x3 block (i, a): block row i of S, correction a ----
        if (tid < (m + 1) * m) {
            const int i = tid / m, a = tid - i * m;
            double Pa[9], Pb[9];
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    Pa[r * 3 + c] = sP[(3 * i + r) * WS + c];                    // P[i, 0]
                    Pb[r * 3 + c] = sP[(3 * i + r) * WS + 3 + 3 * a + c];        // P[i, 1 + a]
                }
            const double* Hr = sHr[a];
            const double* Hl = sHl[a];
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int c = 0; c < 3; c++)
                    sW[(3 * i + r) * WS + 3 * a + c] = (Pa[r * 3] * Hr[c * 3] + Pa[r * 3 + 1] * Hr[c * 3 + 1] + Pa[r * 3 + 2] * Hr[c * 3 + 2]) +
                                                       (Pb[r * 3] * Hl[c * 3] + Pb[r * 3 + 1] * Hl[c * 3 + 1] + Pb[r * 3 + 2] * Hl[c * 3 + 2]);
        }
        ASLAM_LDS_BARRIER();
        WIN_STAMP(3);
        // ---- 4. innovation matrix A = H W + R (aruco_slam.cpp:146): thread (bi, bj) forms its 3x3 block into the G image ----
        if (act) {
            double W0[9], W1[9];
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int c = 0; c < 3; c++) { W0[r * 3 + c] = sW[r * WS + 3 * bj + c]; W1[r * 3 + c] = sW[(3 + 3 * bi + r) * WS + 3 * bj + c]; }
            const double* Hr = sHr[bi];
            const double* Hl = sHl[bi];
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    double v = (Hr[r * 3] * W0[c] + Hr[r * 3 + 1] * W0[3 + c] + Hr[r * 3 + 2] * W0[6 + c]) +
                               (Hl[r * 3] * W1[c] + Hl[r * 3 + 1] * W1[3 + c] + Hl[r * 3 + 2] * W1[6 + c]);
                    if (bi == bj && r == c) v += sRd[bi][r];
                    sG[gpix(3 * bi + r, 3 * bj + c)] = v;
                }
        }
        ASLAM_LDS_BARRIER();
        WIN_STAMP(4);
        // ---- block Gauss-Jordan on the f64 matrix cores.  Waves 0..3 ("workers") each keep one tile row (16 rows x 64 columns)
        // of the 64 x 64 image (A, zero padded) in their accumulators for the whole sweep.  Step j with pivot rows / columns
        // p = 3 j .. 3 j + 2, S = A[p,p] (the reference's S_j = H_j Sigma_{j-1} H_j^T + R_j), C = A[:,p], R = A[p,:] is ONE
        // rank-3 product
        //     A <- A - C~ Y~ ,   C~ = C with rows p replaced by S - I ,   Y~ = S^-1 R with columns p replaced by I + S^-1 ,
        // which leaves S^-1 in the pivot block, S^-1 R in the pivot rows, -C S^-1 in the pivot columns (= -(H_r K_j), whose
        // product with ze_j the pseudo-innovation nu_r collects, quirk Q1) and the Schur update everywhere else.
        // A step is a chain of dependent LDS round trips and cross-lane moves, not arithmetic (measured), so
        //  * only the three pivot ROWS ever leave the accumulators: the partially inverted image stays symmetric up to the sign
        //    of the pivoted/unpivoted cross blocks (A is symmetric to rounding), so C[r][k] = +-R[k][r] and the column operand
        //    is read from the same three rows;
        //  * wave 4 prepares pivot j + 1 WHILE the workers apply step j: the workers publish the rows of pivot j + 2 as they
        //    stand after step j; one phase later wave 4 applies step j + 1's rank-3 correction to those three rows itself, from
        //    the rows and Y~ of pivot j + 1 it still holds in registers (lane = column; uniform values by v_readlane, no LDS),
        //    inverts S and hands Y~ and the corrected rows to the workers.  One barrier per step.
        const int tr = wave >> 1, tc0 = 2 * (wave & 1);           // tile ownership of the two products below (all eight waves)
        const int kd = (n3 + 3) & ~3;                               // their depth: 3m, whole MFMA steps (rows / columns >= 3m of G, J are zero)
        const int gw = wave & 3;                                    // tile row of the working waves
        const int grow = 16 * gw + li;                              // this lane's operand row
        double nu = (wave == 4 && lane < n3) ? sZe[lane] : 0.0;     // wave 4: innovation / pseudo-innovation of row `lane`
        const double ze = nu;
        double Rp0 = 0.0, Rp1 = 0.0, Rp2 = 0.0, Yp0 = 0.0, Yp1 = 0.0, Yp2 = 0.0;      // wave 4: rows and Y~ of the pivot prepared last
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
            // phase j: the workers apply step j and publish the rows of pivot j + 2; wave 4 prepares pivot j + 1
            GJ_STAMP(0);
            if (wave == 4) {
                if (j + 1 < m) {
                    const int jb = (j + 1) & 1, p = 3 * (j + 1);
                    double R0 = sPub[jb][0][lane], R1 = sPub[jb][1][lane], R2 = sPub[jb][2][lane];     // rows of pivot j + 1 as of step j - 1
                    if (j >= 0) {
                        // step j's correction of these rows: C~_j[p + q][k] = R_j[k][p + q] (rows behind pivot j), Y~_j from the registers
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
                    inv3_fast(Sm, Si);
                    const double r0 = R0 + (lane == p ? 1.0 : 0.0), r1 = R1 + (lane == p + 1 ? 1.0 : 0.0), r2 = R2 + (lane == p + 2 ? 1.0 : 0.0);   // R~
                    Yp0 = fma(Si[2], r2, fma(Si[1], r1, Si[0] * r0));
                    Yp1 = fma(Si[5], r2, fma(Si[4], r1, Si[3] * r0));
                    Yp2 = fma(Si[8], r2, fma(Si[7], r1, Si[6] * r0));
                    sGY[jb][0][lane] = Yp0; sGY[jb][1][lane] = Yp1; sGY[jb][2][lane] = Yp2;
                    sRow[jb][0][lane] = R0; sRow[jb][1][lane] = R1; sRow[jb][2][lane] = R2;
                    Rp0 = R0; Rp1 = R1; Rp2 = R2;
                    // nu_r += (C_r S^-1) ze_p for the rows behind the pivot: C[r][k] = R[k][r] there, u = S^-1 ze_p
                    const double z0 = ASLAM_WAVE_BCAST(ze, p), z1 = ASLAM_WAVE_BCAST(ze, p + 1), z2 = ASLAM_WAVE_BCAST(ze, p + 2);
                    const double u0 = fma(Si[2], z2, fma(Si[1], z1, Si[0] * z0)), u1 = fma(Si[5], z2, fma(Si[4], z1, Si[3] * z0)), u2 = fma(Si[8], z2, fma(Si[7], z1, Si[6] * z0));
                    if (lane >= p + 3 && lane < n3) nu += R0 * u0 + R1 * u1 + R2 * u2;
                }
            } else if (wave < 4 && j >= 0) {
                // ---- apply step j: A <- A - C~ Y~ on this wave's tile row.  A operand C~[row][k = lk] from the pivot rows: rows
                // already pivoted carry the opposite sign, the pivot rows themselves S - I; depth 3 is the zero row ----
                const int cb = j & 1, p0 = 3 * j;
                const double rr = sRow[cb][lk][grow];
                double af = grow < p0 ? rr : -rr;                   // = -C~
                if (lk < 3 && grow == p0 + lk) af += 1.0;
#pragma unroll
                for (int t = 0; t < 4; t++) ga[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, sGY[cb][lk][16 * t + li], ga[t], 0, 0, 0);
                if (j + 2 < m) gj_publish_rows(ga, p0 + 6, gw, lk, li, &sPub[cb][0][0]);
            }
            GJ_STAMP(1);
            ASLAM_LDS_BARRIER();
            GJ_STAMP(2);
        }
        WIN_STAMP(5);
        // G = A^-1 into its image (gpix layout: a lane's four rows of a tile are neighbours); nu beside V (it rides the
        // product J = G V as column 63: g = G nu)
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
        // ---- 5. J = G V on the f64 matrix cores, same tile ownership.  V = H P' is W^T (P' is symmetric to rounding, like A above),
        //         so the B operand is read from W's image transposed; column 63 (padding, s <= 63) is nu, so column 63 of J is g ----
        {
            v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
            const bool gcol = 16 * tc0 + 16 + li == 63;
            for (int p0 = 0; p0 < kd; p0 += 4) {
                const double a = sG[gpix(16 * tr + li, p0 + lk)];
                const double b0 = sW[(16 * tc0 + li) * WS + p0 + lk], b1 = gcol ? sNu[p0 + lk] : sW[(16 * tc0 + 16 + li) * WS + p0 + lk];
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, acc1, 0, 0, 0);
            }
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                sV[(16 * tr + lk + 4 * reg) * WS + 16 * tc0 + li] = acc0[reg];
                sV[(16 * tr + lk + 4 * reg) * WS + 16 * tc0 + 16 + li] = acc1[reg];
            }
        }
        ASLAM_LDS_BARRIER();
        if (tid < 64) log[WLOG_g + tid] = sV[tid * WS + 63];
        WIN_STAMP(8);
        // ---- 7. P <- P' - W J (aruco_slam.cpp:204 regrouped), mu_S += W g (:203) ----
        {
            v4d acc0, acc1;
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                acc0[reg] = sP[(16 * tr + lk + 4 * reg) * WS + 16 * tc0 + li];
                acc1[reg] = sP[(16 * tr + lk + 4 * reg) * WS + 16 * tc0 + 16 + li];
            }
            for (int p0 = 0; p0 < kd; p0 += 4) {
                const double a = -sW[(16 * tr + li) * WS + p0 + lk];
                const double b0 = sV[(p0 + lk) * WS + 16 * tc0 + li], b1 = sV[(p0 + lk) * WS + 16 * tc0 + 16 + li];
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, acc1, 0, 0, 0);
            }
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                sP[(16 * tr + lk + 4 * reg) * WS + 16 * tc0 + li] = acc0[reg];
                sP[(16 * tr + lk + 4 * reg) * WS + 16 * tc0 + 16 + li] = acc1[reg];
            }
        }
        ASLAM_LDS_BARRIER();
        if (tid < 64) {                                             // column 63 of the product is -(W g): mu_S += W g (aruco_slam.cpp:203)
            sMu[tid] -= sP[tid * WS + 63];
            sP[tid * WS + 63] = 0.0;
        }
        WIN_STAMP(9);
        // log G and W (both stay untouched until the next frame's steps 3 / 4, behind a barrier)
        for (int e = tid; e < WIMG / 2; e += WCT) reinterpret_cast<double2*>(log + WLOG_W)[e] = reinterpret_cast<const double2*>(sW)[e];
        for (int e = tid; e < 64 * 64; e += WCT) log[WLOG_G + (e >> 6) * WS + (e & 63)] = sG[gpix(e >> 6, e & 63)];      // G leaves in row-major order
        ASLAM_LDS_BARRIER();
        // (J's rows >= 3m and columns >= s are exact zeros - G's are - so the image V is rebuilt into next frame needs no clearing)
        WIN_STAMP(10);
    }
#ifdef ASLAM_WIN_STAMPS
#ifdef ASLAM_GJ_STAMPS
    if ((tid & 63) == 0 && wd.K > 3) {
        printf("tid %d gj step: work %lld barrier %lld\n", tid, gst[1] - gst[0], gst[2] - gst[1]);
    }
#endif
    if (tid == 0 && wd.K > 3) {
        if (false) printf("gj step: loads+S %lld inv %lld frags+mfma %lld writeback %lld copy %lld barrier %lld | step %lld\n", gst[1] - gst[0], gst[2] - gst[1], gst[3] - gst[2],
               gst[4] - gst[3], gst[5] - gst[4], gst[6] - gst[5], gst[6] - gst[0]);
        printf("chain m=%d cycles: predict %lld records %lld barrier+prefetch %lld WV %lld A %lld GJ %lld G,nu+J %lld P,mu %lld logGW %lld | frame %lld\n", m,
               stamps[1] - stamps[0], stamps[2] - stamps[1], 0LL, stamps[3] - stamps[2], stamps[4] - stamps[3], stamps[5] - stamps[4],
               stamps[8] - stamps[5], stamps[9] - stamps[8], stamps[10] - stamps[9], stamps[10] - stamps[0]);
    }
#endif
    // ---- the window's result on S: P_K for the flush, mu_S in place; bookkeeping of the last frame ----
    double* small = E.d_win_small;
    for (int e = tid; e < WIMG; e += WCT) small[WSM_P + e] = sP[e];
    if (tid < s) E.d_mu[sS[tid]] = sMu[tid];
    if (tid == 0) { *E.d_nlast = m; *E.d_npop = m; *E.d_m = m; }
}

// ---------------------------------------------------------------------------------------------------------------------
// Replay of the log: workgroup (x = j, y = i) carries Lambda[:, 16j .. 16j+15], Gamma[16i .. 16i+15, :], Psi block (i, j).
// The next frame's record (G, W images: 68 KB) is fetched into registers while the current one is multiplied.
constexpr int WPF = (WLOG_STRIDE / 2 + 255) / 256;   // 16-byte loads per thread and frame (26): the whole record, unconditionally
constexpr int WREC = WPF * 256 * 2;                  // doubles staged per frame (a little past the record: the log has slack)

__global__ __launch_bounds__(256) void k_ekf_win_scan(EkfState E, WinDesc wd) {
    __shared__ __align__(16) double sRec[WREC];          // one logged frame: G | W | g | H3 | Jacobians
    __shared__ double sLam[64][17];            // Lambda columns (s x 16)
    __shared__ double sGam[16][WS];            // Gamma rows (16 x s)
    __shared__ double sB[64][17], sGB[64][17]; // B = H D Lambda (3m x 16), G B
    __shared__ double sA[16][WS], sAG[16][WS]; // A = Gamma D^T H^T (16 x 3m), A G
    __shared__ double sPsi[16][17], spsi[16];
    const double* sG = sRec + WLOG_G;
    const double* sW = sRec + WLOG_W;
    const double* sgv = sRec + WLOG_g;
    const double* sH3 = sRec + WLOG_H3;
    const double (*sHrec)[18] = reinterpret_cast<const double (*)[18]>(sRec + WLOG_HREC);
    double (*sPsiPart)[16][16] = reinterpret_cast<double (*)[16][16]>(&sAG[0][0]);   // A G is dead once Gamma has been updated
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int bj = blockIdx.x, bi = blockIdx.y;
    const int m = wd.m, n3 = 3 * m;
    // wd.cont = index of this piece within its run (0 = first): piece p reads set (p - 1) & 1 and writes set p & 1
    const double* rsmall = E.d_win_small + ((wd.cont - 1) & 1) * WSM_SET;
    double* small = E.d_win_small + (wd.cont & 1) * WSM_SET;
    if (wd.cont) {                                                  // the run goes on: the accumulators as the previous scan left them
        for (int e = tid; e < 64 * 16; e += 256) { const int r = e >> 4, c = e & 15; sLam[r][c] = rsmall[WSM_LAM + r * WS + 16 * bj + c]; }
        for (int e = tid; e < 16 * 64; e += 256) { const int r = e >> 6, c = e & 63; sGam[r][c] = rsmall[WSM_GAM + (16 * bi + r) * WS + c]; }
        { const int r = tid >> 4, c = tid & 15; sPsi[r][c] = rsmall[WSM_PSI + (16 * bi + r) * WS + 16 * bj + c]; }
        if (tid < 16) spsi[tid] = rsmall[WSM_psi + 16 * bi + tid];
    } else {                                                        // Lambda = Gamma = I, Psi = 0, psi = 0
        for (int e = tid; e < 64 * 16; e += 256) { const int r = e >> 4, c = e & 15; sLam[r][c] = (r == 16 * bj + c && r < wd.s) ? 1.0 : 0.0; }
        for (int e = tid; e < 16 * 64; e += 256) { const int r = e >> 6, c = e & 63; sGam[r][c] = (c == 16 * bi + r && c < wd.s) ? 1.0 : 0.0; }
        for (int e = tid; e < 16 * 16; e += 256) sPsi[e >> 4][e & 15] = 0.0;
        if (tid < 16) spsi[tid] = 0.0;
    }
    for (int e = tid; e < 64 * 17; e += 256) { (&sB[0][0])[e] = 0.0; (&sGB[0][0])[e] = 0.0; }
    for (int e = tid; e < 16 * WS; e += 256) { (&sA[0][0])[e] = 0.0; (&sAG[0][0])[e] = 0.0; }
    double pfx[WPF], pfy[WPF];
    {
        const double2* rec = reinterpret_cast<const double2*>(E.d_win_log + (size_t)wd.log0 * WLOG_STRIDE) + tid;
#pragma unroll
        for (int q = 0; q < WPF; q++) { const double2 v = rec[256 * q]; pfx[q] = v.x; pfy[q] = v.y; }
    }
    __syncthreads();
    for (int k = 0; k < wd.K; k++) {
#pragma unroll
        for (int q = 0; q < WPF; q++) { double2 v; v.x = pfx[q]; v.y = pfy[q]; reinterpret_cast<double2*>(sRec)[tid + 256 * q] = v; }
        ASLAM_LDS_BARRIER();
        if (k + 1 < wd.K) {                                         // in flight while this frame is multiplied
            const double2* rec = reinterpret_cast<const double2*>(E.d_win_log + (size_t)(wd.log0 + k + 1) * WLOG_STRIDE) + tid;
#pragma unroll
            for (int q = 0; q < WPF; q++) { const double2 v = rec[256 * q]; pfx[q] = v.x; pfy[q] = v.y; }
        }
        // D Lambda (rows 0..2) and Gamma D^T (columns 0..2)
        if (tid < 16) {
            const double a = sLam[0][tid], b = sLam[1][tid], c = sLam[2][tid];
            sLam[0][tid] = sH3[0] * a + sH3[1] * b + sH3[2] * c;
            sLam[1][tid] = sH3[3] * a + sH3[4] * b + sH3[5] * c;
            sLam[2][tid] = sH3[6] * a + sH3[7] * b + sH3[8] * c;
        } else if (tid < 32) {
            const int r = tid - 16;
            const double a = sGam[r][0], b = sGam[r][1], c = sGam[r][2];
            sGam[r][0] = a * sH3[0] + b * sH3[1] + c * sH3[2];
            sGam[r][1] = a * sH3[3] + b * sH3[4] + c * sH3[5];
            sGam[r][2] = a * sH3[6] + b * sH3[7] + c * sH3[8];
        }
        ASLAM_LDS_BARRIER();
        // B[3a+r][c] = Hr_a[r,:] Lambda'[0:3, c] + Hl_a[r,:] Lambda'[3+3a .. , c];   A[r'][3a+r] = Gamma'[r', 0:3] Hr_a[r,:]^T + Gamma'[r', 3+3a ..] Hl_a[r,:]^T
        for (int e = tid; e < n3 * 16; e += 256) {
            const int row = e >> 4, c = e & 15, a = row / 3, r = row - 3 * a;
            const double* h = &sHrec[a][r * 6];
            sB[row][c] = (h[0] * sLam[0][c] + h[1] * sLam[1][c] + h[2] * sLam[2][c]) +
                         (h[3] * sLam[3 + 3 * a][c] + h[4] * sLam[4 + 3 * a][c] + h[5] * sLam[5 + 3 * a][c]);
        }
        for (int e = tid; e < 16 * n3; e += 256) {
            const int rp = e / n3, col = e - rp * n3, a = col / 3, r = col - 3 * a;
            const double* h = &sHrec[a][r * 6];
            const double* g = sGam[rp];
            sA[rp][col] = (g[0] * h[0] + g[1] * h[1] + g[2] * h[2]) + (g[3 + 3 * a] * h[3] + g[4 + 3 * a] * h[4] + g[5 + 3 * a] * h[5]);
        }
        ASLAM_LDS_BARRIER();
        // GB = G B (wave w: tile row w), AG = A G (wave w: tile column w)
        {
            v4d accB = {0.0, 0.0, 0.0, 0.0}, accA = {0.0, 0.0, 0.0, 0.0};
            for (int p0 = 0; p0 < 64; p0 += 4) {
                const double a1 = sG[(16 * wave + li) * WS + p0 + lk], b1 = sB[p0 + lk][li];
                accB = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, accB, 0, 0, 0);
                const double a2 = sA[li][p0 + lk], b2 = sG[(p0 + lk) * WS + 16 * wave + li];
                accA = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, accA, 0, 0, 0);
            }
#pragma unroll
            for (int reg = 0; reg < 4; reg++) { sGB[16 * wave + lk + 4 * reg][li] = accB[reg]; sAG[lk + 4 * reg][16 * wave + li] = accA[reg]; }
        }
        ASLAM_LDS_BARRIER();
        // Lambda -= W GB (wave w: tile row w), Gamma -= AG V = AG W^T (wave w: tile column w), Psi += A GB (depth split over the waves)
        {
            v4d accL, accG, accP = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int reg = 0; reg < 4; reg++) { accL[reg] = sLam[16 * wave + lk + 4 * reg][li]; accG[reg] = sGam[lk + 4 * reg][16 * wave + li]; }
            for (int p0 = 0; p0 < 64; p0 += 4) {
                const double a1 = -sW[(16 * wave + li) * WS + p0 + lk], b1 = sGB[p0 + lk][li];
                accL = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, accL, 0, 0, 0);
                const double a2 = -sAG[li][p0 + lk], b2 = sW[(16 * wave + li) * WS + p0 + lk];      // V = W^T
                accG = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, accG, 0, 0, 0);
            }
            for (int p0 = 16 * wave; p0 < 16 * wave + 16; p0 += 4) {
                const double a3 = sA[li][p0 + lk], b3 = sGB[p0 + lk][li];
                accP = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b3, accP, 0, 0, 0);
            }
            double ps = 0;                                          // psi += A g
            if (tid < 16) for (int c = 0; c < n3; c++) ps += sA[tid][c] * sgv[c];
            ASLAM_LDS_BARRIER();                                    // every wave is done with A G (its LDS doubles as the Psi partials)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                sLam[16 * wave + lk + 4 * reg][li] = accL[reg];
                sGam[lk + 4 * reg][16 * wave + li] = accG[reg];
                sPsiPart[wave][lk + 4 * reg][li] = accP[reg];
            }
            if (tid < 16) spsi[tid] += ps;
        }
        ASLAM_LDS_BARRIER();
        { const int r = tid >> 4, c = tid & 15; sPsi[r][c] += (sPsiPart[0][r][c] + sPsiPart[1][r][c]) + (sPsiPart[2][r][c] + sPsiPart[3][r][c]); }
        ASLAM_LDS_BARRIER();
    }
    if (bi == 0) for (int e = tid; e < 64 * 16; e += 256) { const int r = e >> 4, c = e & 15; small[WSM_LAM + r * WS + 16 * bj + c] = sLam[r][c]; }
    if (bj == 0) for (int e = tid; e < 16 * 64; e += 256) { const int r = e >> 6, c = e & 63; small[WSM_GAM + (16 * bi + r) * WS + c] = sGam[r][c]; }
    { const int r = tid >> 4, c = tid & 15; small[WSM_PSI + (16 * bi + r) * WS + 16 * bj + c] = sPsi[r][c]; }
    if (bj == 0 && tid < 16) small[WSM_psi + 16 * bi + tid] = spsi[tid];
}

// ---------------------------------------------------------------------------------------------------------------------
// The one pass over Sigma per window.  Tile (r0, c0), 64 x 64: T = Psi Y_0tile, Sigma_tile -= X_0tile T (formed transposed so
// that the read-modify-write of the column-major Sigma is coalesced, as k_ekf_apply does), then the rows / columns that belong to
// S are replaced: row S_p <- (Lambda Y_0)[p, :], column S_q <- (X_0 Gamma)[:, q], (S_p, S_q) <- P_K[p][q]; mu_R += X_0 psi.
__global__ __launch_bounds__(256) void k_ekf_win_flush(EkfState E, WinDesc wd) {
    __shared__ __align__(16) double sM[WIMG];          // Psi, later Lambda / Gamma
    __shared__ double sY[64][64];                      // Y_0 tile: [p][column]
    __shared__ double sX[64][64];                      // X_0^T tile: [p][row]
    __shared__ double sT[64][64];                      // product tile
    __shared__ int sRowPos[64], sColPos[64], sAnyRow, sAnyCol;
    const int ld = E.ld;
    const int N = 3 + 3 * (*E.d_L);
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    if (r0 >= N || c0 >= N) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int s = wd.s;
    const double* small = E.d_win_small;                          // P_K
    const double* acc_set = E.d_win_small + (wd.cont & 1) * WSM_SET;   // the accumulators as the run's LAST scan piece (index wd.cont) left them
    double sig[4][4];
#pragma unroll
    for (int ri = 0; ri < 4; ri++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int c = c0 + 16 * wave + lk + 4 * reg, r = r0 + 16 * ri + li;
            sig[ri][reg] = (r < N && c < N) ? E.d_sigma[(size_t)c * ld + r] : 0.0;
        }
    for (int e = tid; e < WIMG; e += 256) sM[e] = acc_set[WSM_PSI + e];
    for (int e = tid; e < 64 * 64; e += 256) {
        const int p = e >> 6, x = e & 63;
        sY[p][x] = (p < s && c0 + x < N) ? E.d_V[(size_t)p * ld + c0 + x] : 0.0;
        sX[p][x] = (p < s && r0 + x < N) ? E.d_Wt[(size_t)p * ld + r0 + x] : 0.0;
    }
    if (tid < 64) {
        sRowPos[tid] = r0 + tid < N ? E.d_win_sidx[r0 + tid] : -1;
        sColPos[tid] = c0 + tid < N ? E.d_win_sidx[c0 + tid] : -1;
    }
    if (tid == 0) { sAnyRow = 0; sAnyCol = 0; }
    __syncthreads();
    if (tid < 64) { if (sRowPos[tid] >= 0) sAnyRow = 1; if (sColPos[tid] >= 0) sAnyCol = 1; }
    // T = Psi Y_0tile: wave w owns columns 16w .. 16w+15 and all four 16-row tiles
    {
        v4d acc[4];
#pragma unroll
        for (int qi = 0; qi < 4; qi++) acc[qi] = v4d{0.0, 0.0, 0.0, 0.0};
        for (int p0 = 0; p0 < 64; p0 += 4) {
            const double b = sY[p0 + lk][16 * wave + li];
#pragma unroll
            for (int qi = 0; qi < 4; qi++) acc[qi] = __builtin_amdgcn_mfma_f64_16x16x4f64(sM[(16 * qi + li) * WS + p0 + lk], b, acc[qi], 0, 0, 0);
        }
#pragma unroll
        for (int qi = 0; qi < 4; qi++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) sT[16 * qi + lk + 4 * reg][16 * wave + li] = acc[qi][reg];
    }
    __syncthreads();
    const bool anyRow = sAnyRow != 0, anyCol = sAnyCol != 0;       // uniform
    // Sigma tile (transposed product): D'[c][r] = sum_p T[p][c] X_0^T[p][r]
    {
        v4d acc[4];
#pragma unroll
        for (int ri = 0; ri < 4; ri++) acc[ri] = v4d{0.0, 0.0, 0.0, 0.0};
        for (int p0 = 0; p0 < 64; p0 += 4) {
            const double a = sT[p0 + lk][16 * wave + li];
#pragma unroll
            for (int ri = 0; ri < 4; ri++) acc[ri] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sX[p0 + lk][16 * ri + li], acc[ri], 0, 0, 0);
        }
#pragma unroll
        for (int ri = 0; ri < 4; ri++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) sig[ri][reg] -= acc[ri][reg];
    }
    if (anyRow) {
        // rows of S: (Lambda Y_0)[p][c]
        __syncthreads();
        for (int e = tid; e < WIMG; e += 256) sM[e] = acc_set[WSM_LAM + e];
        __syncthreads();
        v4d acc[4];
#pragma unroll
        for (int qi = 0; qi < 4; qi++) acc[qi] = v4d{0.0, 0.0, 0.0, 0.0};
        for (int p0 = 0; p0 < 64; p0 += 4) {
            const double b = sY[p0 + lk][16 * wave + li];
#pragma unroll
            for (int qi = 0; qi < 4; qi++) acc[qi] = __builtin_amdgcn_mfma_f64_16x16x4f64(sM[(16 * qi + li) * WS + p0 + lk], b, acc[qi], 0, 0, 0);
        }
#pragma unroll
        for (int qi = 0; qi < 4; qi++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) sT[16 * qi + lk + 4 * reg][16 * wave + li] = acc[qi][reg];     // sT[p][column]
        __syncthreads();
#pragma unroll
        for (int ri = 0; ri < 4; ri++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int cl = 16 * wave + lk + 4 * reg, rl = 16 * ri + li;
                const int p = sRowPos[rl];
                if (p >= 0) sig[ri][reg] = sT[p][cl];
            }
    }
    if (anyCol) {
        // columns of S: (X_0 Gamma)[r][q] = sum_p X_0^T[p][r] Gamma[p][q]; formed as D[q][r] with A[i = q][k = p] = Gamma[p][q]
        __syncthreads();
        for (int e = tid; e < WIMG; e += 256) sM[e] = acc_set[WSM_GAM + e];
        __syncthreads();
        v4d acc[4];
#pragma unroll
        for (int qi = 0; qi < 4; qi++) acc[qi] = v4d{0.0, 0.0, 0.0, 0.0};
        for (int p0 = 0; p0 < 64; p0 += 4) {
            const double b = sX[p0 + lk][16 * wave + li];                                                   // B[k = p][j = r]
#pragma unroll
            for (int qi = 0; qi < 4; qi++) acc[qi] = __builtin_amdgcn_mfma_f64_16x16x4f64(sM[(p0 + lk) * WS + 16 * qi + li], b, acc[qi], 0, 0, 0);
        }
#pragma unroll
        for (int qi = 0; qi < 4; qi++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) sT[16 * qi + lk + 4 * reg][16 * wave + li] = acc[qi][reg];     // sT[q][row]
        __syncthreads();
#pragma unroll
        for (int ri = 0; ri < 4; ri++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int cl = 16 * wave + lk + 4 * reg, rl = 16 * ri + li;
                const int q = sColPos[cl];
                if (q >= 0) {
                    const int p = sRowPos[rl];
                    sig[ri][reg] = p >= 0 ? small[WSM_P + p * WS + q] : sT[q][rl];
                }
            }
    }
#pragma unroll
    for (int ri = 0; ri < 4; ri++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int c = c0 + 16 * wave + lk + 4 * reg, r = r0 + 16 * ri + li;
            if (r < N && c < N) E.d_sigma[(size_t)c * ld + r] = sig[ri][reg];
        }
    if (blockIdx.y == 0 && tid < 64 && r0 + tid < N && sRowPos[tid] < 0) {
        double acc = 0;
        for (int p = 0; p < s; p++) acc += sX[p][tid] * acc_set[WSM_psi + p];
        E.d_mu[r0 + tid] += acc;                                   // mu_R += X_0 psi
    }
}

// ---- host side --------------------------------------------------------------------------------------------------------
void launch_ekf_win_chain(hipStream_t st, const EkfState& E, const SlamParams& sp, const WinDesc& wd, const ObsRaw* obs,
                          const unsigned* n_markers, const double* enc, const unsigned char* d_obs_idx) {
    const int ngather = (E.ld + WCT - 1) / WCT;
    hipLaunchKernelGGL(k_ekf_win_chain, dim3(1 + ngather), dim3(WCT), 0, st, E, sp, wd, obs, n_markers, enc, d_obs_idx);
}
void launch_ekf_win_scan(hipStream_t st, const EkfState& E, const WinDesc& wd) {
    hipLaunchKernelGGL(k_ekf_win_scan, dim3(4, 4), dim3(256), 0, st, E, wd);
}
void launch_ekf_win_flush(hipStream_t st, const EkfState& E, const WinDesc& wd) {
    const int t = (E.ld + 63) / 64;
    hipLaunchKernelGGL(k_ekf_win_flush, dim3(t, t), dim3(256), 0, st, E, wd);
}

} // namespace aslam
