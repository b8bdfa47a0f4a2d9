// Windowed EKF: a run of consecutive frames whose fused landmarks all lie in one set S (no new landmark; at most kWinSMax
// landmarks in the union) is processed on the S x S block of the covariance only; the rest of Sigma follows ONCE per window.
//
// Split the state into S (robot pose + the landmarks of the set, s = 3 + 3 nS, padded to SP = 16 T) and R (everything else):
//     Sigma = [ P   Y ]      P = Sigma[S,S]   Y = Sigma[S,R]   Z = Sigma[R,R]      (Sigma is symmetric to rounding: aruco_slam.cpp:73, 204)
//             [ Y^T Z ]
// The reference's filter is sequential: one predict per encoder sample (aruco_slam.cpp:21-74: Sigma <- D Sigma D^T + Q, D = identity
// except its pose block) and, per popped observation j of a frame, K_j = Sigma H_j^T (H_j Sigma H_j^T + R_j)^-1 with the LIVE Sigma
// and the innovation at the frame's FROZEN mean, mu += K_j ze_j, Sigma <- (I - K_j H_j) Sigma (aruco_slam.cpp:88, 108-207).  H_j has
// only the pose block and the block of landmark j, so restricted to S every one of these steps is a rank-3 (predict: rank-4)
// correction of P,
//     correction:  c = H_j P (3 x s),  S_j = c H_j^T + R_j,  Kt = S_j^-1 c (= K_j^T),  P <- P - Kt^T c,  mu_S += Kt^T ze_j
//     predict:     P <- P + u r2 + (r2^T + P22 u) u^T + Q     (u = the third column of D - I, r2 = row 2 of P; Q = F Qk F^T has rank 2)
// and touches R only linearly through Y:  Y <- D Y,  Y <- Y - Kt^T (H_j Y),  Z <- Z - (H_j Y)^T S_j^-1 (H_j Y),
// mu_R <- mu_R + (H_j Y)^T S_j^-1 ze_j.  Hence, over the whole window, with an s x s accumulator Lambda (= I at the start),
//     Y_K = Lambda Y_0      Z_K = Z_0 - Y_0^T Psi Y_0      mu_R,K = mu_R,0 + Y_0^T psi
//     per step:  t = H_j Lambda,  u = S_j^-1 t,  Lambda <- Lambda - Kt^T t  (predict: Lambda <- D Lambda),  Psi += t^T u,  psi += t^T S_j^-1 ze_j
// This is the reference's own arithmetic, step for step, on the rows and columns it can change; nothing is approximated.  The one
// property used is the symmetry of Sigma (H P read as (P H^T)^T, X = Y^T).  Frames of a window may fuse ANY subset of S, in the
// reference's pop order (ascending landmark index = ascending position in S), and may drop "stationary" observations
// (aruco_slam.cpp:192-198: a no-op): a window ends only when a frame brings a landmark that does not fit into S, or a new one.
//
// Kernels (per window; a window's frames are cut into pieces of a few frames):
//   k_ekf_win_step    one launch per piece, three roles by workgroup:
//     chain   (workgroup 0) walks the steps of piece i.  P lives in the f64 matrix-core accumulators of the worker waves for the
//             whole piece (wave w: RW tile rows of T 16 x 16 tiles); a step is one v_mfma_f64_16x16x4_f64 per tile (depth 3 or 4).
//             A separate "prepare" wave (lane = column) runs one step AHEAD: the workers publish the six rows (pose + landmark) of
//             step j + 2 as they stand after step j, the prepare wave applies step j + 1's correction to them itself from the
//             operands of the previous step, forms c, S, S^-1, Kt and hands the operands to the workers: one barrier per step.  It
//             also logs -Kt, S^-1, ze and the Jacobian scalars;
//     replay  (SP / 8 workgroups) replays the log of piece i - 1, each on its own 8 columns of Lambda (all rows; in LDS) and its
//             part of psi, and logs t and u;
//     Psi     (T workgroups) adds the t^T u of piece i - 2 to Psi on the matrix cores.
//             The three depend on each other only through the previous launch: the stream orders them, no events between pieces.
//   k_ekf_win_gather  Y_0 = rows S of Sigma (second stream, behind the previous window's flush)
//   k_ekf_win_thin    [Psi; Lambda] Y_0 as one tiled product (U = Psi Y_0, Y_K = Lambda Y_0), mu_R += Y_0^T psi;
//   k_ekf_update_mfma (ekf.hip) Sigma -= Y_0^T U: the ONE pass over Sigma per window;  k_ekf_win_fix writes rows / columns S and P_K.
//   k_ekf_win_next_*  the next window's P and mu_S from this window's small results, before its flush has run.
#include "common.h"
#include "ekf.h"
#include "ekf_dev.h"
#include <cmath>
#include <cstdlib>
#include <algorithm>

namespace aslam {

constexpr int WBW = 8;                        // columns of Lambda per scan workgroup

// development aid (make FLAGS+=-DASLAM_WIN_STAMPS): cycle stamps of the prepare wave's step phases and of the workers, summed over a piece
#ifdef ASLAM_WIN_STAMPS
#define WSTAMP(i) do { const long long t_ = clock64(); stamp_acc[i] += t_ - stamp_last; stamp_last = t_; } while (0)
#else
#define WSTAMP(i) do { } while (0)
#endif

__host__ __device__ inline int win_log_stride(int T) { return 3 * 16 * T + kWinHdr; }
__host__ __device__ inline int win_tlog_stride(int T) { return 8 * 16 * T; }
// header of a logged step (doubles after the three operand rows)
enum { WH_TYPE = 0, WH_POS = 1, WH_SI = 2, WH_ZE = 11, WH_C = 14, WH_S = 15, WH_G02 = 16, WH_G12 = 17, WH_A = 18, WH_B = 19 };
// d_win_small: images of the window, each SPm x SPm at most (SPm = E.win_sp_max), stored with the window's own row stride SP
__host__ __device__ inline size_t wsm_P(int SPm, int par) { return (size_t)par * SPm * SPm; }             // P image of window parity par
__host__ __device__ inline size_t wsm_LAM(int SPm, int par) { return (size_t)(2 + par) * SPm * SPm; }      // Lambda / Psi / psi: also per parity (the flush of a
__host__ __device__ inline size_t wsm_PSI(int SPm, int par) { return (size_t)(4 + par) * SPm * SPm; }      // window reads them while the next window's replay writes its own)
__host__ __device__ inline size_t wsm_psi(int SPm, int par) { return (size_t)6 * SPm * SPm + (size_t)par * SPm; }
__host__ __device__ inline size_t wsm_MU(int SPm, int par) { return (size_t)6 * SPm * SPm + (size_t)(2 + par) * SPm; }   // mu_S image
__host__ __device__ inline size_t wsm_doubles(int SPm) { return (size_t)6 * SPm * SPm + (size_t)4 * SPm; }

int ekf_win_tiles(int nS) { return nS <= 20 ? 4 : nS <= 41 ? 8 : 12; }

__device__ __forceinline__ int win_state_index(const WinDesc& wd, int p) {      // state offset of position p of S
    return p < 3 ? p : wd.li[(p - 3) / 3] + (p - 3) % 3;
}

// value of element `idx` (0 .. 64 NC - 1) of a per-lane array v[NC] (element idx lives in lane idx & 63 of v[idx >> 6]); idx is
// wave-uniform.  Every lane of the wave must call it.
template <int NC> __device__ __forceinline__ double bcast_at(const double (&v)[NC], int idx) {
    const int ch = idx >> 6, ln = idx & 63;
    double r = ASLAM_WAVE_BCAST(v[0], ln);
    if (NC > 1) { const double r1 = ASLAM_WAVE_BCAST(v[NC > 1 ? 1 : 0], ln); r = ch == 1 ? r1 : r; }
    if (NC > 2) { const double r2 = ASLAM_WAVE_BCAST(v[NC > 2 ? 2 : 0], ln); r = ch == 2 ? r2 : r; }
    return r;
}

// the same with the chunk selected by uniform conditional moves in front of ONE cross-lane read (idx wave-uniform, every lane calls)
template <int NC> __device__ __forceinline__ double bcast_sel(const double (&v)[NC], int idx) {
    const int ch = idx >> 6;
    double x = v[0];
    if (NC > 1) x = ch == 1 ? v[NC > 1 ? 1 : 0] : x;
    if (NC > 2) x = ch == 2 ? v[NC > 2 ? 2 : 0] : x;
    return ASLAM_WAVE_BCAST(x, idx & 63);
}

// Rows p .. p + 2 of the image leave the accumulators of the worker wave(s) that own them (row r = 16 g + lk + 4 reg of tile row
// g): lanes with lk == r & 3 write their register reg = (r >> 2) & 3 of each of the T column tiles.  The register number must
// be static (a select chain over the accumulators costs more than the rest of the step): one instance per p mod 16.
template <int P16, int T, int RW>
__device__ __forceinline__ void win_publish_at(const v4d (&acc)[RW][T], int p, int wv, int lk, int li, double* rows, int spp) {
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const int rl = (P16 + q) & 15;
        const int g = (p + q) >> 4;
        double* d = rows + q * spp + li;
#pragma unroll
        for (int rr = 0; rr < RW; rr++)
            if (g == wv * RW + rr && lk == (rl & 3)) {
#pragma unroll
                for (int t = 0; t < T; t++) d[16 * t] = acc[rr][t][rl >> 2];
            }
    }
}
template <int T, int RW>
__device__ __forceinline__ void win_publish(const v4d (&acc)[RW][T], int p, int wv, int lk, int li, double* rows, int spp) {
    switch (p & 15) {
#define ASLAM_WP_CASE(i) case i: win_publish_at<i, T, RW>(acc, p, wv, lk, li, rows, spp); break;
        ASLAM_WP_CASE(0) ASLAM_WP_CASE(1) ASLAM_WP_CASE(2) ASLAM_WP_CASE(3) ASLAM_WP_CASE(4) ASLAM_WP_CASE(5) ASLAM_WP_CASE(6) ASLAM_WP_CASE(7)
        ASLAM_WP_CASE(8) ASLAM_WP_CASE(9) ASLAM_WP_CASE(10) ASLAM_WP_CASE(11) ASLAM_WP_CASE(12) ASLAM_WP_CASE(13) ASLAM_WP_CASE(14) ASLAM_WP_CASE(15)
#undef ASLAM_WP_CASE
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// T tiles per side, RW tile rows per worker wave: ceil(T / RW) worker waves + 1 prepare wave.  With T = 4 (2 x 2 rows on two
// workers) the prepare wave - the kernel's critical path - has a SIMD to itself (measured: -11 % per step); at T = 8 the same idea
// (3 + 3 + 2 rows on three workers) makes the workers the bottleneck (measured: +12 %), so it keeps one worker per SIMD.
template <int T> struct WinChainLds {
    static constexpr int SP = 16 * T, SPP = SP + 16, NSMAX = kWinPieceMax * 64;
    double sA[2][4][SPP];                  // a step's A operand rows  Aop[k][row]   (P += Aop^T Bop)
    double sB[2][4][SPP];                  // ... and B operand rows   Bop[k][column]
    double sPub[2][6][SPP];                // rows 0..2 (pose) and the landmark rows of the step after next
    double sMu[SPP];                       // prepare wave's scratch: mu_S by position
    int sS[SP];
    int sOff[kWinPieceMax + 1];
    unsigned char sPos[NSMAX], sIdx[NSMAX], sFrm[NSMAX];   // per step: landmark position (255 = predict), correction index, frame
};
template <int T, int RW>
__device__ void win_chain_role(const EkfState& E, const SlamParams& sp, const WinDesc& wd, const ObsRaw* __restrict__ obs,
                               const double* __restrict__ enc, unsigned char* smem) {
    constexpr int SP = 16 * T, SPP = SP + 16, NC = SP / 64;       // SPP: operand rows lk and lk + 1 fall on opposite halves of the bank row
    constexpr int NWK = (T + RW - 1) / RW, NT = (NWK + 1) * 64;
    WinChainLds<T>& L = *reinterpret_cast<WinChainLds<T>*>(smem);
    auto& sA = L.sA; auto& sB = L.sB; auto& sPub = L.sPub; auto& sMu = L.sMu; auto& sS = L.sS; auto& sOff = L.sOff;
    auto& sPos = L.sPos; auto& sIdx = L.sIdx; auto& sFrm = L.sFrm;
    if (threadIdx.x >= NT) return;                                  // (the launch's block is sized for its widest role)
    const int tid = threadIdx.x;
    const int nS = wd.nS, s = 3 + 3 * nS;
    const int ld = E.ld;
    const WinFrame* __restrict__ frames = E.d_win_frames;

    const int wave = tid >> 6, lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    for (int e = tid; e < SP; e += NT) sS[e] = e < s ? win_state_index(wd, e) : 0;
    for (int e = tid; e < 2 * 4 * SPP; e += NT) { (&sA[0][0][0])[e] = 0.0; (&sB[0][0][0])[e] = 0.0; }
    for (int e = tid; e < 2 * 6 * SPP; e += NT) (&sPub[0][0][0])[e] = 0.0;
    // steps per frame (1 predict + m corrections), every frame's count loaded by its own thread (one thread walking the plan would
    // pay one dependent global load per frame), then the running sum
    if (tid < wd.K) sOff[tid + 1] = 1 + frames[wd.first_slot + tid].m;
    if (tid == 0) sOff[0] = 0;
    __syncthreads();
    if (tid == 0)
        for (int k = 0; k < wd.K; k++) sOff[k + 1] += sOff[k];
    __syncthreads();
    const int NS = sOff[wd.K];                                    // steps of the piece: per frame one predict + m corrections
    for (int k = 0; k < wd.K; k++) {
        const WinFrame& fr = frames[wd.first_slot + k];
        for (int a = tid; a <= fr.m; a += NT) {
            const int st = sOff[k] + a;
            sPos[st] = a == 0 ? 255 : fr.cpos[a - 1];
            sIdx[st] = a == 0 ? 0 : (unsigned char)(a - 1);
            sFrm[st] = (unsigned char)k;
        }
    }
    __syncthreads();

    if (wave < NWK) {
        // =================================== worker waves: P in the accumulators ===================================
        v4d acc[RW][T];
        const double* Pimg = E.d_win_small + wsm_P(E.win_sp_max, wd.wpar);
        const bool from_img = wd.piece != 0 || wd.from_image != 0;
#pragma unroll
        for (int rr = 0; rr < RW; rr++)
#pragma unroll
            for (int t = 0; t < T; t++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int r = 16 * (wave * RW + rr) + lk + 4 * reg, c = 16 * t + li;
                    double v = 0.0;
                    if (r < SP) {                                                  // (the last worker may own fewer than RW tile rows)
                        if (from_img) v = Pimg[(size_t)r * SP + c];                // the window goes on (or was prepared ahead): P from the image
                        else if (r < s && c < s) v = E.d_sigma[(size_t)sS[c] * ld + sS[r]];
                    }
                    acc[rr][t][reg] = v;
                }
        // rows of step 0 (pose rows only: a predict) and of step 1, as they stand before any step
        if (wave == 0) {
#pragma unroll
            for (int q = 0; q < 3; q++)
                if (lk == q) {
#pragma unroll
                    for (int t = 0; t < T; t++) { sPub[0][q][16 * t + li] = acc[0][t][0]; sPub[1][q][16 * t + li] = acc[0][t][0]; }
                }
        }
        if (NS > 1 && sPos[1] != 255) win_publish<T, RW>(acc, 3 + 3 * sPos[1], wave, lk, li, &sPub[1][3][0], SPP);
        ASLAM_LDS_BARRIER();
#ifdef ASLAM_WIN_STAMPS
        long long stamp_acc[4] = {0, 0, 0, 0}, stamp_last = clock64();
#endif
        for (int j = -1; j < NS; j++) {
            WSTAMP(0);
            if (j >= 0) {
                const int cb = j & 1;
                double b[T];
#pragma unroll
                for (int t = 0; t < T; t++) b[t] = sB[cb][lk][16 * t + li];
#pragma unroll
                for (int rr = 0; rr < RW; rr++) {
                    if (T % RW != 0 && wave * RW + rr >= T) break;
                    const double a = sA[cb][lk][16 * (wave * RW + rr) + li];
#pragma unroll
                    for (int t = 0; t < T; t++) acc[rr][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[t], acc[rr][t], 0, 0, 0);
                }
                if (j + 2 < NS) {                                   // rows of step j + 2 as they stand after step j
                    if (wave == 0) {
#pragma unroll
                        for (int q = 0; q < 3; q++)
                            if (lk == q) {
#pragma unroll
                                for (int t = 0; t < T; t++) sPub[cb][q][16 * t + li] = acc[0][t][0];
                            }
                    }
                    const int pn = sPos[j + 2];
                    if (pn != 255) win_publish<T, RW>(acc, 3 + 3 * pn, wave, lk, li, &sPub[cb][3][0], SPP);
                }
            }
            WSTAMP(1);
            ASLAM_LDS_BARRIER();
        }
#ifdef ASLAM_WIN_STAMPS
        if (lane == 0 && wd.piece == 1) printf("worker %d T %d steps %d: barrier-wait %lld work %lld cycles per step\n", wave, T, NS, stamp_acc[0] / (NS + 1), stamp_acc[1] / (NS + 1));
#endif
        // P_K for the next piece / the flush
        double* Pout = E.d_win_small + wsm_P(E.win_sp_max, wd.wpar);
#pragma unroll
        for (int rr = 0; rr < RW; rr++)
#pragma unroll
            for (int t = 0; t < T; t++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++)
                    if (T % RW == 0 || wave * RW + rr < T) Pout[(size_t)(16 * (wave * RW + rr) + lk + 4 * reg) * SP + 16 * t + li] = acc[rr][t][reg];
        return;
    }

    // ======================================= prepare wave: lane = column (NC chunks of 64) =======================================
    // What a step needs from another column (the previous step's A operand at the six row indices, c at the six special columns)
    // is wave-uniform and is read back from LDS as a broadcast - the operands live there anyway (measured: cheaper than v_readlane
    // plus chunk selection).
#ifndef ASLAM_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3);                                  // the critical path of the kernel: wins the issue slot over the worker wave it shares a SIMD with
#endif
    const double kl = sp.kl, kr = sp.kr, inv2b = 1.0 / (2 * sp.b), invb = 1 / sp.b, Qk = sp.Q_k;
    double* const logbase = E.d_win_log + (size_t)wd.log0 * win_log_stride(T);
    double* const muimg = E.d_win_small + wsm_MU(E.win_sp_max, wd.wpar);
    double mu[NC], pB[4][NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int col = lane + 64 * c;
        // mu_S travels between the pieces of a window in its image; only the window's last piece puts it back into the state
        mu[c] = col >= s ? 0.0 : (wd.piece != 0 || wd.from_image != 0) ? muimg[col] : E.d_mu[sS[col]];
#pragma unroll
        for (int k = 0; k < 4; k++) pB[k][c] = 0.0;
    }
    // per-frame records, lane a = correction a of the frame (aruco_slam.cpp:119-143 at the frozen mean)
    double rze0 = 0, rze1 = 0, rze2 = 0, rR0 = 0, rR1 = 0, rR2 = 0, rg02 = 0, rg12 = 0;
    double cth = 1.0, sth = 0.0;
    bool prev_predict = false;
    bool dirty0 = false, dirty1 = false;                           // operand buffer 0 / 1 holds a predict's fourth depth row
    // the first frame's inputs
    ObsRaw nObs{};
    if (lane < sOff[1] - 1) nObs = obs[(size_t)wd.first_slot * kMarkerMax + frames[wd.first_slot].cdet[lane]];
    double e_wl, e_wr, e_dt;
    { const double* e = enc + (size_t)3 * wd.first_slot; e_wl = e[0]; e_wr = e[1]; e_dt = e[2]; }
    ASLAM_LDS_BARRIER();                                           // (pairs with the workers' barrier after their first publish)
#ifdef ASLAM_WIN_STAMPS
    long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last = clock64();
    int n_pred = 0;
#endif
    for (int j = -1; j < NS; j++) {
        const int n = j + 1;                                       // the step prepared in this phase
        WSTAMP(0);
        if (n < NS) {
            const int nb = n & 1, pb = j & 1;
            const int pos = sPos[n];
            const bool is_predict = pos == 255;
            const int lrow = is_predict ? 0 : 3 + 3 * pos;         // first landmark row (a predict has none: copies of the pose rows)
            double r[6][NC];
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int c = 0; c < NC; c++) r[i][c] = sPub[nb][i][lane + 64 * c];
            if (j >= 0) {
                // step j's correction of these rows: P[R][col] += sum_k Aop_j[k][R] Bop_j[k][col]
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    if (kk == 3 && !prev_predict) break;            // a correction has depth 3
                    const double f0 = sA[pb][kk][0], f1 = sA[pb][kk][1], f2 = sA[pb][kk][2];      // (LDS broadcast reads: measured faster than v_readlane)
                    const double f3 = sA[pb][kk][lrow], f4 = sA[pb][kk][lrow + 1], f5 = sA[pb][kk][lrow + 2];
#pragma unroll
                    for (int c = 0; c < NC; c++) {
                        r[0][c] = fma(f0, pB[kk][c], r[0][c]); r[1][c] = fma(f1, pB[kk][c], r[1][c]); r[2][c] = fma(f2, pB[kk][c], r[2][c]);
                        r[3][c] = fma(f3, pB[kk][c], r[3][c]); r[4][c] = fma(f4, pB[kk][c], r[4][c]); r[5][c] = fma(f5, pB[kk][c], r[5][c]);
                    }
                }
            }
            WSTAMP(1);
            double* log = logbase + (size_t)n * win_log_stride(T);
            double* hdr = log + 3 * SP;
            double A[4][NC], B[4][NC];
            if (is_predict) {
#ifdef ASLAM_WIN_STAMPS
                n_pred++;
#endif
                const int k = sFrm[n];
                const int slot = wd.first_slot + k;
                const int fm = sOff[k + 1] - sOff[k] - 1;           // corrections of the frame
                // ---- predict (aruco_slam.cpp:35-73) with the final mean of the previous frame ----
                const double delta_sl = kl * (e_dt * e_wl), delta_sr = kr * (e_dt * e_wr);
                const double delta_theta = (delta_sr - delta_sl) * inv2b;
                const double delta_s = 0.5 * (delta_sr + delta_sl);
#pragma unroll
                for (int c = 0; c < NC; c++) sMu[lane + 64 * c] = mu[c];
                __builtin_amdgcn_wave_barrier();
                const double m0 = sMu[0], m1 = sMu[1], m2 = sMu[2];
                double th = m2 + delta_theta;
                wrap1(th);
                // the two sincos of the frame in one call: even lanes the mid-step heading, odd lanes the new heading
                double sv, cv;
                sincos((lane & 1) ? th : m2 + 0.5 * delta_theta, &sv, &cv);
                const double cm = ASLAM_WAVE_BCAST(cv, 0), sm = ASLAM_WAVE_BCAST(sv, 0);
                cth = ASLAM_WAVE_BCAST(cv, 1); sth = ASLAM_WAVE_BCAST(sv, 1);
                const double ua = -delta_s * sm, ub = delta_s * cm;                 // H3 = I + [ua ub 0]^T e2^T
                const double f = 0.5 * kl * e_dt;                                    // kl for BOTH wheels (quirk Q7)
                const double su0 = Qk * fabs(e_wl), su1 = Qk * fabs(e_wr);
                const double P22 = ASLAM_WAVE_BCAST(r[2][0], 2);
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const int col = lane + 64 * c;
                    const double u = col == 0 ? ua : col == 1 ? ub : 0.0;
                    const double w0 = col == 0 ? f * cm : col == 1 ? f * sm : col == 2 ? f * invb : 0.0;        // wkh column 0
                    const double w1 = col == 0 ? f * cm : col == 1 ? f * sm : col == 2 ? f * -invb : 0.0;       // wkh column 1
                    A[0][c] = u;                       B[0][c] = r[2][c];
                    A[1][c] = fma(P22, u, r[2][c]);    B[1][c] = u;
                    A[2][c] = su0 * w0;                B[2][c] = w0;
                    A[3][c] = su1 * w1;                B[3][c] = w1;
                    sA[nb][3][col] = A[3][c]; sB[nb][3][col] = B[3][c];
                }
                if (nb) dirty1 = true; else dirty0 = true;
                const double np0 = m0 + delta_s * cm, np1 = m1 + delta_s * sm;
                if (lane == 0) mu[0] = np0;
                if (lane == 1) mu[0] = np1;
                if (lane == 2) mu[0] = th;
                // ---- the frame's records at the frozen mean (pose just predicted, landmarks as the previous frame left them) ----
                if (lane < fm) {
                    const int q = 3 + 3 * sPos[sOff[k] + 1 + lane];
                    const double mx = sMu[q], my = sMu[q + 1], mth = sMu[q + 2];
                    const double gdx = mx - np0, gdy = my - np1;
                    double gdth = mth - th;
                    wrap1(gdth);
                    const double zh0 = gdx * cth + gdy * sth, zh1 = -gdx * sth + gdy * cth;
                    double z2 = nObs.th - gdth;
                    wrap1(z2);
                    rze0 = nObs.x - zh0; rze1 = nObs.y - zh1; rze2 = z2;
                    rg02 = -gdx * sth + gdy * cth; rg12 = -gdx * cth - gdy * sth;
                    rR0 = nObs.r[0]; rR1 = nObs.r[1]; rR2 = nObs.r[2];
                }
                __builtin_amdgcn_wave_barrier();
                const WinFrame& fr = frames[slot];
                if (lane == 0 && slot < E.max_slots) {
                    int* st = E.d_slot_stat + 4 * slot;
                    st[0] = fr.n_markers; st[1] = 0; st[2] = fm; st[3] = fr.npop - fm;
                }
                if (wd.last && k == wd.K - 1) {
                    // what the window's last frame leaves behind for whatever follows: pop list, last_observed_marker_ (aruco_slam.cpp:202, 263)
                    const int npop = fr.npop;
                    if (lane < npop) {
                        const ObsRaw o = obs[(size_t)slot * kMarkerMax + fr.pdet[lane]];
                        const bool upd = fr.pact[lane] == 1;
                        PopRec pr;
                        pr.id = o.id; pr.index = fr.pidx[lane]; pr.action = upd ? 1 : 2; pr.pad = 0;
                        pr.z[0] = o.x; pr.z[1] = o.y; pr.z[2] = o.th;
                        pr.r[0] = o.r[0]; pr.r[1] = o.r[1]; pr.r[2] = o.r[2];
                        E.d_pop[lane] = pr;
                        LastObs lo;
                        lo.id = o.id; lo.pad = 0;
                        const double nanv = __builtin_nan("");
                        lo.z[0] = upd ? o.x : nanv; lo.z[1] = upd ? o.y : nanv; lo.z[2] = upd ? o.th : nanv;   // stationary: last_observation_ stays unset
                        E.d_last[lane] = lo;
                    }
                    if (lane == 0) { *E.d_nlast = npop; *E.d_npop = npop; *E.d_m = fm; }
                }
                // the next frame's inputs are fetched while this one is solved
                if (k + 1 < wd.K) {
                    const int fmn = sOff[k + 2] - sOff[k + 1] - 1;
                    if (lane < fmn) nObs = obs[(size_t)(slot + 1) * kMarkerMax + frames[slot + 1].cdet[lane]];
                    const double* e = enc + (size_t)3 * (slot + 1);
                    e_wl = e[0]; e_wr = e[1]; e_dt = e[2];
                }
                // header of the logged step: type 0, D's two entries, the frame's cos / sin (the corrections of the frame use them)
                if (lane == 0) { hdr[WH_TYPE] = 0.0; hdr[WH_POS] = -1.0; hdr[WH_A] = ua; hdr[WH_B] = ub; hdr[WH_C] = cth; hdr[WH_S] = sth; }
                WSTAMP(2);
            } else {
                // ---- correction a of the frame: c = H P, S = c H^T + R, Kt = S^-1 c ----
                const int a = sIdx[n];
                const double ze0 = ASLAM_WAVE_BCAST(rze0, a), ze1 = ASLAM_WAVE_BCAST(rze1, a), ze2 = ASLAM_WAVE_BCAST(rze2, a);
                const double R0 = ASLAM_WAVE_BCAST(rR0, a), R1 = ASLAM_WAVE_BCAST(rR1, a), R2 = ASLAM_WAVE_BCAST(rR2, a);
                const double g02 = ASLAM_WAVE_BCAST(rg02, a), g12 = ASLAM_WAVE_BCAST(rg12, a);
                // Gxm = [ -c -s g02  c  s 0 ;  s -c g12 -s  c 0 ;  0 0 -1  0 0 1 ]   (aruco_slam.cpp:140-143)
                double cc[3][NC];
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    cc[0][c] = (-cth * r[0][c] - sth * r[1][c] + g02 * r[2][c]) + (cth * r[3][c] + sth * r[4][c]);
                    cc[1][c] = (sth * r[0][c] - cth * r[1][c] + g12 * r[2][c]) + (-sth * r[3][c] + cth * r[4][c]);
                    cc[2][c] = r[5][c] - r[2][c];
                    sB[nb][0][lane + 64 * c] = cc[0][c]; sB[nb][1][lane + 64 * c] = cc[1][c]; sB[nb][2][lane + 64 * c] = cc[2][c];   // (the B operand, in place)
                }
                __builtin_amdgcn_wave_barrier();
                WSTAMP(3);
                double Sm[9], Si[9];
#pragma unroll
                for (int kk = 0; kk < 3; kk++) {
                    const double p0 = sB[nb][kk][0], p1 = sB[nb][kk][1], p2 = sB[nb][kk][2];
                    const double l0 = sB[nb][kk][lrow], l1 = sB[nb][kk][lrow + 1], l2 = sB[nb][kk][lrow + 2];
                    Sm[kk * 3 + 0] = (-cth * p0 - sth * p1 + g02 * p2) + (cth * l0 + sth * l1);
                    Sm[kk * 3 + 1] = (sth * p0 - cth * p1 + g12 * p2) + (-sth * l0 + cth * l1);
                    Sm[kk * 3 + 2] = l2 - p2;
                }
                Sm[0] += R0; Sm[4] += R1; Sm[8] += R2;
                inv3_fast(Sm, Si);
                WSTAMP(4);
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    // K = (P H^T) S^-1, (P H^T) = c^T:  Kt[k][col] = sum_k' c[k'][col] Si[k'][k];  the A operand is -Kt
                    const double k0 = cc[0][c] * Si[0] + cc[1][c] * Si[3] + cc[2][c] * Si[6];
                    const double k1 = cc[0][c] * Si[1] + cc[1][c] * Si[4] + cc[2][c] * Si[7];
                    const double k2 = cc[0][c] * Si[2] + cc[1][c] * Si[5] + cc[2][c] * Si[8];
                    mu[c] += k0 * ze0 + k1 * ze1 + k2 * ze2;                       // mu_ += K ze (aruco_slam.cpp:203)
                    A[0][c] = -k0; A[1][c] = -k1; A[2][c] = -k2; A[3][c] = 0.0;
                    B[0][c] = cc[0][c]; B[1][c] = cc[1][c]; B[2][c] = cc[2][c]; B[3][c] = 0.0;
                }
                if (nb ? dirty1 : dirty0) {                         // the buffer last held a predict's fourth depth row
#pragma unroll
                    for (int c = 0; c < NC; c++) { sA[nb][3][lane + 64 * c] = 0.0; sB[nb][3][lane + 64 * c] = 0.0; }
                    if (nb) dirty1 = false; else dirty0 = false;
                }
                // header of the logged step, stored by the lanes that hold the values
                if (lane == 0) {
                    hdr[WH_TYPE] = 1.0; hdr[WH_POS] = (double)pos;
#pragma unroll
                    for (int q = 0; q < 9; q++) hdr[WH_SI + q] = Si[q];
                }
                if (lane == a) { hdr[WH_ZE] = rze0; hdr[WH_ZE + 1] = rze1; hdr[WH_ZE + 2] = rze2; hdr[WH_G02] = rg02; hdr[WH_G12] = rg12; }
                WSTAMP(5);
            }
            // operands to the workers, the log, and this wave's own copy
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const int col = lane + 64 * c;
                sA[nb][0][col] = A[0][c]; sA[nb][1][col] = A[1][c]; sA[nb][2][col] = A[2][c];
                if (is_predict) { sB[nb][0][col] = B[0][c]; sB[nb][1][col] = B[1][c]; sB[nb][2][col] = B[2][c]; }
                log[col] = A[0][c]; log[SP + col] = A[1][c]; log[2 * SP + col] = A[2][c];
#pragma unroll
                for (int kk = 0; kk < 4; kk++) pB[kk][c] = B[kk][c];
            }
            prev_predict = is_predict;
            WSTAMP(6);
        }
        ASLAM_LDS_BARRIER();
    }
#ifdef ASLAM_WIN_STAMPS
    if (lane == 0 && wd.piece == 1) {
        const int nc = NS - n_pred;
        printf("prepare T %d steps %d (%d predict): barrier %lld | rows+correct %lld | predict path %lld per predict | c %lld S+inv %lld Kt+hdr %lld per correction | operands+log %lld per step\n",
               T, NS, n_pred, stamp_acc[0] / (NS + 1), stamp_acc[1] / NS, stamp_acc[2] / (n_pred ? n_pred : 1), stamp_acc[3] / (nc ? nc : 1), stamp_acc[4] / (nc ? nc : 1),
               stamp_acc[5] / (nc ? nc : 1), stamp_acc[6] / NS);
    }
#endif
    // ---- mu_S for the next piece; back into the state at the window's end ----
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int col = lane + 64 * c;
        if (col < s) { muimg[col] = mu[c]; if (wd.last) E.d_mu[sS[col]] = mu[c]; }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Replay of a piece's log on Lambda: workgroup b carries columns WBW b .. WBW b + WBW - 1 (all SP rows; in LDS) and its entries of psi.
// Per step: t = H Lambda (3 x WBW), Lambda += Aop^T t, psi += t^T (S^-1 ze); t and u = S^-1 t are logged for the Psi product.
// Two LDS barriers per step.  Wave 0 forms t (and then u, for the log) and is the only wave that stores, waves 1..3 are the only
// ones that load (the next step's record, one step ahead): no wave ever waits for its own stores to be acknowledged.
struct WinReplay { int piece, log0, nsteps, wpar; };              // the piece a replay role works on
template <int T> struct WinScanLds {
    static constexpr int SP = 16 * T, REC = 3 * SP + kWinHdr;
    double sLam[SP][WBW + 1];
    double sRec[2][REC];
    double sT[2][4][WBW];                  // t (row 3 stays zero)
};
template <int T>
__device__ void win_scan_role(const EkfState& E, const WinReplay& wd, int b, unsigned char* smem) {
    constexpr int SP = 16 * T, EPT = SP * WBW / 256, REC = 3 * SP + kWinHdr;
    constexpr int RPT = (REC + 191) / 192;                         // record doubles per loading thread
    WinScanLds<T>& L = *reinterpret_cast<WinScanLds<T>*>(smem);
    auto& sLam = L.sLam; auto& sRec = L.sRec; auto& sT = L.sT;
    if (threadIdx.x >= 256) return;
    const int nsteps = wd.nsteps;
    const int tid = threadIdx.x;
    const int ls = win_log_stride(T), ts = win_tlog_stride(T);
    double* Lam = E.d_win_small + wsm_LAM(E.win_sp_max, wd.wpar);
    double* psi = E.d_win_small + wsm_psi(E.win_sp_max, wd.wpar);
    for (int e = tid; e < SP * WBW; e += 256) {
        const int r = e / WBW, c = e % WBW;
        sLam[r][c] = wd.piece ? Lam[(size_t)r * SP + WBW * b + c] : (r == WBW * b + c ? 1.0 : 0.0);
    }
    double ps = (tid >= 4 * WBW && tid < 5 * WBW && wd.piece) ? psi[WBW * b + tid - 4 * WBW] : 0.0;     // (the thread that accumulates entry c: below)
    double cth = 1.0, sth = 0.0;
    const double* logp = E.d_win_log + (size_t)wd.log0 * ls;
    double* tlog = E.d_win_tlog + (size_t)wd.log0 * ts + WBW * b;
    const int lt = tid - 64;                                       // loading thread index (waves 1..3)
    // the log was written moments ago by another CU: a record takes longer to arrive than a step lasts, so the loading waves keep
    // PFD records in flight (registers), one per step of the unrolled loop
    constexpr int PFD = 3;
    double pf[PFD][RPT];
#pragma unroll
    for (int u = 0; u < PFD; u++)
#pragma unroll
        for (int q = 0; q < RPT; q++) { const int e = lt + 192 * q; pf[u][q] = (lt >= 0 && e < REC && u < nsteps) ? logp[(size_t)u * ls + e] : 0.0; }
    for (int e = tid; e < 2 * 4 * WBW; e += 256) (&sT[0][0][0])[e] = 0.0;
    __syncthreads();
    for (int n0 = 0; n0 < nsteps; n0 += PFD) {
#pragma unroll
      for (int u = 0; u < PFD; u++) {
        const int n = n0 + u;
        if (n >= nsteps) break;
        double* rec = sRec[n & 1];
        double (*tt)[WBW] = sT[n & 1];
        if (lt >= 0) {
#pragma unroll
            for (int q = 0; q < RPT; q++) { const int e = lt + 192 * q; if (e < REC) rec[e] = pf[u][q]; }
            if (n + PFD < nsteps) {                                 // in flight while the next PFD steps are applied
                const double* nx = logp + (size_t)(n + PFD) * ls;
#pragma unroll
                for (int q = 0; q < RPT; q++) { const int e = lt + 192 * q; pf[u][q] = e < REC ? nx[e] : 0.0; }
            }
        }
        ASLAM_LDS_BARRIER();
        const double* hd = rec + 3 * SP;
        const bool is_predict = hd[WH_TYPE] == 0.0;
        if (tid < WBW) {
            const int c = tid;
            if (is_predict) {
                // Lambda <- D Lambda (rows 0, 1 += (a, b) row 2); nothing for Psi / psi
                cth = hd[WH_C]; sth = hd[WH_S];                     // the frame's cos / sin (every piece starts with a predict)
                const double r2 = sLam[2][c];
                sLam[0][c] += hd[WH_A] * r2; sLam[1][c] += hd[WH_B] * r2;
                tt[0][c] = 0.0; tt[1][c] = 0.0; tt[2][c] = 0.0;
            } else {
                const int lrow = 3 + 3 * (int)hd[WH_POS];
                const double g02 = hd[WH_G02], g12 = hd[WH_G12];
                const double r0 = sLam[0][c], r1 = sLam[1][c], r2 = sLam[2][c], l0 = sLam[lrow][c], l1 = sLam[lrow + 1][c], l2 = sLam[lrow + 2][c];
                tt[0][c] = (-cth * r0 - sth * r1 + g02 * r2) + (cth * l0 + sth * l1);
                tt[1][c] = (sth * r0 - cth * r1 + g12 * r2) + (-sth * l0 + cth * l1);
                tt[2][c] = l2 - r2;
            }
        }
        ASLAM_LDS_BARRIER();
        if (tid < 8 * WBW) {
            // the step's log rows of this block (wave 0): t (rows 0..2, row 3 zero), u = S^-1 t (rows 4..6, row 7 zero); psi += t^T (S^-1 ze)
            const int row = tid / WBW, c = tid % WBW, k = row & 3;
            const double t0 = tt[0][c], t1 = tt[1][c], t2 = tt[2][c];
            double v = 0.0;
            if (!is_predict && k < 3) {
                const double* Si = hd + WH_SI + 3 * k;
                v = row < 4 ? tt[k][c] : Si[0] * t0 + Si[1] * t1 + Si[2] * t2;     // (Z <- Z - (H Y)^T S^-1 (H Y))
                if (row == 4) {                                     // thread (4, c): the whole of psi's entry c
                    const double* S9 = hd + WH_SI;
                    const double z0 = hd[WH_ZE], z1 = hd[WH_ZE + 1], z2 = hd[WH_ZE + 2];
                    const double w0 = S9[0] * z0 + S9[1] * z1 + S9[2] * z2, w1 = S9[3] * z0 + S9[4] * z1 + S9[5] * z2, w2 = S9[6] * z0 + S9[7] * z1 + S9[8] * z2;
                    ps += t0 * w0 + t1 * w1 + t2 * w2;
                }
            }
            tlog[(size_t)n * ts + row * SP + c] = v;
        }
        if (!is_predict) {
            // Lambda[r][c] += sum_k Aop[k][r] t[k][c]      (c is the same for all of a thread's entries)
            const int c = tid % WBW;
            const double t0 = tt[0][c], t1 = tt[1][c], t2 = tt[2][c];
#pragma unroll
            for (int q = 0; q < EPT; q++) {
                const int r = (tid + 256 * q) / WBW;
                sLam[r][c] += rec[r] * t0 + rec[SP + r] * t1 + rec[2 * SP + r] * t2;
            }
        }
      }
    }
    __syncthreads();
    for (int e = tid; e < SP * WBW; e += 256) { const int r = e / WBW, c = e % WBW; Lam[(size_t)r * SP + WBW * b + c] = sLam[r][c]; }
    // psi's entry c was accumulated by thread (row 4, c) = tid 4 WBW + c
    if (tid >= 4 * WBW && tid < 5 * WBW) psi[WBW * b + tid - 4 * WBW] = ps;
}

// Psi (+)= sum over the piece's steps of t^T u on the f64 matrix cores: workgroup = tile row, wave w = tile columns w, w + 4, ...
// The operands come straight from the t / u log (L2): four steps are fetched ahead of the four products.
template <int T>
__device__ void win_psi_role(const EkfState& E, const WinReplay& wd, int tr) {
    constexpr int SP = 16 * T, TW = (T + 3) / 4, UN = 4;
    if (threadIdx.x >= 256) return;
    const int nsteps = wd.nsteps;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int ts = win_tlog_stride(T);
    double* Psi = E.d_win_small + wsm_PSI(E.win_sp_max, wd.wpar);
    const double* tlog = E.d_win_tlog + (size_t)wd.log0 * ts;
    v4d acc[TW];
#pragma unroll
    for (int q = 0; q < TW; q++) {
        const int tc = wave + 4 * q;
#pragma unroll
        for (int reg = 0; reg < 4; reg++)
            acc[q][reg] = (wd.piece && tc < T) ? Psi[(size_t)(16 * tr + lk + 4 * reg) * SP + 16 * tc + li] : 0.0;
    }
    // A[i][k] = t[k][16 tr + i] (lane k * 16 + i), B[k][j] = u[k][16 tc + j]
    for (int n0 = 0; n0 < nsteps; n0 += UN) {
        double a[UN], bv[UN][TW];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const bool ok = n0 + u < nsteps;
            const double* st = tlog + (size_t)(ok ? n0 + u : n0) * ts;
            a[u] = ok ? st[lk * SP + 16 * tr + li] : 0.0;
#pragma unroll
            for (int q = 0; q < TW; q++) { const int tc = wave + 4 * q; bv[u][q] = tc < T ? st[(4 + lk) * SP + 16 * tc + li] : 0.0; }
        }
#pragma unroll
        for (int u = 0; u < UN; u++)
#pragma unroll
            for (int q = 0; q < TW; q++) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], bv[u][q], acc[q], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < TW; q++) {
        const int tc = wave + 4 * q;
        if (tc < T) {
#pragma unroll
            for (int reg = 0; reg < 4; reg++) Psi[(size_t)(16 * tr + lk + 4 * reg) * SP + 16 * tc + li] = acc[q][reg];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// One launch = the chain of piece i (workgroup 0), the replay of piece i - 1 on Lambda (the next SP / 8 workgroups) and the Psi
// product of piece i - 2 (the last T workgroups).  The three depend on each other only through the PREVIOUS launch (the log of
// piece i - 1 is complete when this launch starts: same stream), so no events are needed between the pieces of a window and the
// replay is hidden behind the chain: with one event per piece the EKF alone ran 14 % slower (cfg2).
template <int T, int RW>
__global__ __launch_bounds__(((T + RW - 1) / RW + 1) * 64 > 256 ? ((T + RW - 1) / RW + 1) * 64 : 256)
void k_ekf_win_step(EkfState E, SlamParams sp, WinDesc wd, WinReplay rs, WinReplay rq, const ObsRaw* __restrict__ obs, const double* __restrict__ enc) {
    constexpr size_t kLds = sizeof(WinChainLds<T>) > sizeof(WinScanLds<T>) ? sizeof(WinChainLds<T>) : sizeof(WinScanLds<T>);
    __shared__ __align__(16) unsigned char smem[kLds];
    constexpr int NSCAN = 16 * T / WBW;
    const int bx = blockIdx.x;
    if (bx == 0) { if (wd.K > 0) win_chain_role<T, RW>(E, sp, wd, obs, enc, smem); }
    else if (bx <= NSCAN) { if (rs.nsteps > 0) win_scan_role<T>(E, rs, bx - 1, smem); }
    else if (rq.nsteps > 0) win_psi_role<T>(E, rq, bx - 1 - NSCAN);
}

// ---------------------------------------------------------------------------------------------------------------------
// U = Psi Y_0 -> d_T and Y_K = Lambda Y_0 -> d_V (both SP x N) as ONE product [Psi; Lambda] (2 SP x SP) . Y_0 (SP x N) on the matrix
// cores: workgroup (x, y) = 64 columns of Sigma x 64 rows of the stacked matrix, depth in chunks of 64 staged through LDS (both
// operands, coalesced), wave w = 16 of the columns x all 64 rows.  Workgroups with y = 0 also add Y_0^T psi to their 64 entries
// of mu_R.
template <int T>
__global__ __launch_bounds__(256) void k_ekf_win_thin(EkfState E, WinDesc wd) {
    constexpr int SP = 16 * T, YS = 66;
    __shared__ double sY[64 * YS];                                  // Y_0 chunk: [depth][column]
    __shared__ double sM[64 * YS];                                  // Psi / Lambda chunk: [row][depth]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int ld = E.ld;
    const int N = 3 + 3 * (*E.d_L);
    const int c0 = blockIdx.x * 64;
    if (c0 >= N) return;
    const int r0 = blockIdx.y * 64;                                 // row of the stacked matrix
    const bool lam = r0 >= SP;
    const int s = 3 + 3 * wd.nS;
    const double* M = E.d_win_small + (lam ? wsm_LAM(E.win_sp_max, wd.wpar) : wsm_PSI(E.win_sp_max, wd.wpar)) + (size_t)(lam ? r0 - SP : r0) * SP;
    const double* psi = E.d_win_small + wsm_psi(E.win_sp_max, wd.wpar);
    double* out = lam ? E.d_V : E.d_T;
    const int orow = lam ? r0 - SP : r0;
    v4d acc[4];
#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] = v4d{0.0, 0.0, 0.0, 0.0};
    double macc = 0.0;
    for (int ch = 0; ch < SP / 64; ch++) {
        if (ch) __syncthreads();
        for (int e = tid; e < 64 * 64; e += 256) {
            const int p = e >> 6, x = e & 63;
            sY[p * YS + x] = c0 + x < N ? E.d_Wt[(size_t)(64 * ch + p) * ld + c0 + x] : 0.0;
            sM[p * YS + x] = M[(size_t)p * SP + 64 * ch + x];
        }
        __syncthreads();
        for (int p0 = 0; p0 < 64; p0 += 4) {
            const double bv = sY[(p0 + lk) * YS + 16 * wave + li];
#pragma unroll
            for (int q = 0; q < 4; q++) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(sM[(16 * q + li) * YS + p0 + lk], bv, acc[q], 0, 0, 0);
        }
        if (blockIdx.y == 0 && tid < 64)
            for (int p = 0; p < 64 && 64 * ch + p < s; p++) macc += sY[p * YS + tid] * psi[64 * ch + p];
    }
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int p = orow + 16 * q + lk + 4 * reg, c = c0 + 16 * wave + li;
            if (c < N) out[(size_t)p * ld + c] = acc[q][reg];
        }
    if (blockIdx.y == 0 && tid < 64 && c0 + tid < N && E.d_win_sidx[c0 + tid] < 0) E.d_mu[c0 + tid] += macc;      // mu_R += Y_0^T psi
}

// Rows and columns S of Sigma after the Z pass: row S_p <- Y_K[p][:], column S_p <- the same (symmetry), (S_p, S_q) <- P_K[p][q].
__global__ __launch_bounds__(256) void k_ekf_win_fix(EkfState E, WinDesc wd, int SP) {
    const int ld = E.ld;
    const int N = 3 + 3 * (*E.d_L);
    const int s = 3 + 3 * wd.nS;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= N) return;
    const int tp = E.d_win_sidx[t];
    const double* Pimg = E.d_win_small + wsm_P(E.win_sp_max, wd.wpar);
    for (int p = blockIdx.y; p < s; p += gridDim.y) {
        const int Sp = win_state_index(wd, p);
        const double v = tp >= 0 ? Pimg[(size_t)p * SP + tp] : E.d_V[(size_t)p * ld + t];
        E.d_sigma[(size_t)Sp * ld + t] = v;                        // column S_p, row t (coalesced)
        if (tp < 0) E.d_sigma[(size_t)t * ld + Sp] = v;            // row S_p, column t
    }
}

// Y_0 (row p = row S_p of Sigma = its column S_p) -> d_Wt (rows s .. SP - 1 zero), position table of S.
__global__ __launch_bounds__(256) void k_ekf_win_gather(EkfState E, WinDesc wd) {
    const int ld = E.ld, nS = wd.nS, s = 3 + 3 * nS, SP = 16 * wd.T;
    const int N = 3 + 3 * (*E.d_L);
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= N) return;
    int pos = -1;
    if (t < 3) pos = t;
    else {
        const int base = (t - 3) / 3 * 3 + 3;
        for (int a = 0; a < nS; a++) if (wd.li[a] == base) pos = 3 + 3 * a + (t - base);
    }
    if (blockIdx.y == 0) E.d_win_sidx[t] = pos;
    for (int p = blockIdx.y; p < SP; p += gridDim.y) E.d_Wt[(size_t)p * ld + t] = p < s ? E.d_sigma[(size_t)win_state_index(wd, p) * ld + t] : 0.0;
}

// ---- early start of the next window -----------------------------------------------------------------------------------
// The chain of the window that follows needs only P' = Sigma[S', S'] and mu_S' as they will stand AFTER the previous window's
// flush; both follow from the previous window's small results without the pass over Sigma:
//     (a, b in S)      P_K                 (a in S, b not)   (Lambda Y_0)[a][b]            (neither)   Sigma_old[a][b] - (Y_0^T Psi Y_0)[a][b]
//     mu: in S as the chain left it, otherwise mu_old + Y_0^T psi.
// k_ekf_win_next_gather collects Y_0's columns S' (Vg, zero where the entry is in S), Sigma_old[S', S'] and mu_old[S'] into
// small dense buffers with the index tables a miniature EkfState view needs; k_ekf_win_thin and k_ekf_update_mfma then run on that
// view (N := s'), and k_ekf_win_next_fix assembles the image the chain loads.  The flush of the previous window runs meanwhile.
__global__ __launch_bounds__(256) void k_ekf_win_next_gather(EkfState E, WinDesc pv, WinDesc nx) {
    const int ld = E.ld, SPm = E.win_sp_max;
    const int s2 = 3 + 3 * nx.nS, SPp = 16 * pv.T;
    double* Vg = E.d_win_next;                                      // [p][a'] row stride SPm
    double* Ptmp = E.d_win_next + (size_t)3 * SPm * SPm;            // column-major, ld SPm
    double* mu2 = E.d_win_small + wsm_MU(SPm, nx.wpar);
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a >= s2) return;
    const int ia = win_state_index(nx, a);
    const int pa = E.d_win_sidx[ia];
    if (blockIdx.y == 0) {
        E.d_win_next_idx[a] = pa;
        if (a == 0) E.d_win_next_idx[SPm] = nx.nS;
        mu2[a] = E.d_mu[ia];
    }
    for (int p = blockIdx.y; p < SPp; p += gridDim.y) Vg[(size_t)p * SPm + a] = pa < 0 ? E.d_Wt[(size_t)p * ld + ia] : 0.0;
    for (int b = blockIdx.y; b < s2; b += gridDim.y) Ptmp[(size_t)b * SPm + a] = E.d_sigma[(size_t)win_state_index(nx, b) * ld + ia];
}
__global__ __launch_bounds__(256) void k_ekf_win_next_fix(EkfState E, WinDesc pv, WinDesc nx) {
    const int SPm = E.win_sp_max, SPp = 16 * pv.T, SPn = 16 * nx.T;
    const int s2 = 3 + 3 * nx.nS;
    const double* Lg = E.d_win_next + (size_t)2 * SPm * SPm;        // (Lambda Vg)[p][a']
    const double* Ptmp = E.d_win_next + (size_t)3 * SPm * SPm;
    const double* Pprev = E.d_win_small + wsm_P(SPm, pv.wpar);
    double* Pout = E.d_win_small + wsm_P(SPm, nx.wpar);
    const int b = blockIdx.x * 256 + threadIdx.x;                   // column of the image (fastest)
    if (b >= SPn) return;
    const int pb = b < s2 ? E.d_win_next_idx[b] : -1;
    for (int a = blockIdx.y; a < SPn; a += gridDim.y) {
        double v = 0.0;
        if (a < s2 && b < s2) {
            const int pa = E.d_win_next_idx[a];
            v = pa >= 0 && pb >= 0 ? Pprev[(size_t)pa * SPp + pb] : pa >= 0 ? Lg[(size_t)pa * SPm + b] : pb >= 0 ? Lg[(size_t)pb * SPm + a] : Ptmp[(size_t)b * SPm + a];
        }
        Pout[(size_t)a * SPn + b] = v;
    }
}

// ---- host side --------------------------------------------------------------------------------------------------------
// Every workgroup of a step launch asks for more than half of a CU's LDS (an unused dynamic allocation on top of the static one), so
// that no second workgroup - a replay workgroup of the same launch, or anything else - is placed on the chain workgroup's CU and
// competes with the prepare wave for issue slots and LDS bandwidth (ASLAM_WIN_SHARE_CU: off, for comparison).
template <int T, class K> static void launch_step_kernel(K kernel, hipStream_t st, int nb, int nt, size_t static_lds, const EkfState& E, const SlamParams& sp,
                                                const WinDesc& wd, const WinReplay& rs, const WinReplay& rq, const ObsRaw* obs, const double* enc) {
    static const bool share = std::getenv("ASLAM_WIN_SHARE_CU") != nullptr;
    const size_t dyn = share ? 0 : (size_t)84 * 1024 - std::min(static_lds, (size_t)20 * 1024);      // static + dynamic > 80 KB of the 160 KB
    static bool attr_done = false;
    if (!attr_done && dyn > 0) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        attr_done = true;
    }
    hipLaunchKernelGGL(kernel, dim3(nb), dim3(nt), dyn, st, E, sp, wd, rs, rq, obs, enc);
}
void launch_ekf_win_step(hipStream_t st, const EkfState& E, const SlamParams& sp, const WinDesc& wd, const ObsRaw* obs, const double* enc,
                         int s_piece, int s_log0, int s_nsteps, int q_piece, int q_log0, int q_nsteps) {
    const WinReplay rs{s_piece, s_log0, s_nsteps, wd.wpar}, rq{q_piece, q_log0, q_nsteps, wd.wpar};
    const int nb = 1 + 16 * wd.T / WBW + wd.T;
    if (wd.T == 4) launch_step_kernel<4>(k_ekf_win_step<4, 2>, st, nb, 256, sizeof(WinChainLds<4>), E, sp, wd, rs, rq, obs, enc);
    else if (wd.T == 8) launch_step_kernel<8>(k_ekf_win_step<8, 2>, st, nb, 320, sizeof(WinChainLds<8>), E, sp, wd, rs, rq, obs, enc);   // (3 + 3 + 2 rows on three workers: measured slower)
    else launch_step_kernel<12>(k_ekf_win_step<12, 2>, st, nb, 448, sizeof(WinChainLds<12>), E, sp, wd, rs, rq, obs, enc);
}
void launch_ekf_win_gather(hipStream_t st, const EkfState& E, const WinDesc& wd) {
    hipLaunchKernelGGL(k_ekf_win_gather, dim3((E.ld + 255) / 256, 16), dim3(256), 0, st, E, wd);       // y: rows of Y_0 in turn (one load in flight per thread otherwise)
}
static void launch_thin(hipStream_t st, const EkfState& E, const WinDesc& wd, int ncols) {
    const int SP = 16 * wd.T, nb = (ncols + 63) / 64;
    if (wd.T == 4) hipLaunchKernelGGL(k_ekf_win_thin<4>, dim3(nb, 2 * SP / 64), dim3(256), 0, st, E, wd);
    else if (wd.T == 8) hipLaunchKernelGGL(k_ekf_win_thin<8>, dim3(nb, 2 * SP / 64), dim3(256), 0, st, E, wd);
    else hipLaunchKernelGGL(k_ekf_win_thin<12>, dim3(nb, 2 * SP / 64), dim3(256), 0, st, E, wd);
}
void launch_ekf_win_next(hipStream_t st, const EkfState& E, const WinDesc& pv, const WinDesc& nx) {
    const int SPm = E.win_sp_max, s2 = 3 + 3 * nx.nS, SPn = 16 * nx.T;
    hipLaunchKernelGGL(k_ekf_win_next_gather, dim3((s2 + 255) / 256, 64), dim3(256), 0, st, E, pv, nx);
    // the miniature state the previous window's thin products and the Sigma pass run on: N := s', Sigma := Sigma_old[S', S']
    EkfState E2 = E;
    E2.ld = SPm;
    E2.d_sigma = E.d_win_next + (size_t)3 * SPm * SPm;
    E2.d_Wt = E.d_win_next;                                         // Vg
    E2.d_T = E.d_win_next + (size_t)1 * SPm * SPm;                  // Psi Vg
    E2.d_V = E.d_win_next + (size_t)2 * SPm * SPm;                  // Lambda Vg
    E2.d_mu = E.d_win_small + wsm_MU(SPm, nx.wpar);
    E2.d_win_sidx = E.d_win_next_idx;
    E2.d_L = E.d_win_next_idx + SPm;
    launch_thin(st, E2, pv, s2);
    launch_ekf_update_mfma(st, E2, 16 * pv.T);
    hipLaunchKernelGGL(k_ekf_win_next_fix, dim3((SPn + 255) / 256, 16), dim3(256), 0, st, E, pv, nx);
}
void launch_ekf_win_flush(hipStream_t st, const EkfState& E, const WinDesc& wd) {
    const int SP = 16 * wd.T;
    launch_thin(st, E, wd, E.ld);
    launch_ekf_update_mfma(st, E, SP);                            // Sigma -= Y_0^T U (d_Wt = Y_0, d_T = U), rows / columns S included
    hipLaunchKernelGGL(k_ekf_win_fix, dim3((E.ld + 255) / 256, 16), dim3(256), 0, st, E, wd, SP);
}

} // namespace aslam
