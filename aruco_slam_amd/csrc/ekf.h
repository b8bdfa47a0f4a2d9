// EKF-SLAM state resident in HBM and the launchers of the EKF kernels (ekf.hip).
// Replaces Eigen's role in ArucoSlam::addEncoder / addImage (src/aruco_slam.cpp:21-74, 88-263).
#pragma once
#include "common.h"

namespace aslam {

constexpr int kIdTableSize = 1024;     // marker id -> landmark index (std::map<int,int> aruco_id_map, aruco_slam.h:164)

struct PopRec {                        // one popped observation (aruco_slam.cpp:92-95) and what was done with it
    int id, index, action, pad;        // action: 0 augment, 1 update, 2 stationary no-op
    double z[3];
    double r[3];
};
struct LastObs {                       // last_observed_marker_ entry (aruco_slam.h:188): id + last_observation_
    int id, pad;
    double z[3];                       // NaN = never set (quirk Q2/Q3: never matches)
};
struct UpdRec {                        // one fused EKF correction
    int li, pad;                       // state offset of the landmark: 3 + 3*index
    double Gxm[18];                    // 3 x 6 Jacobian block (aruco_slam.cpp:140-143)
    double ze[3];                      // innovation (aruco_slam.cpp:137-138)
    double r[3];                       // diag of Rk
};
struct MapRecord {                     // 104-byte landmark record gathered across GPUs
    int id, index;
    double x, y, theta;
    double S[9];
};

// ---- windowed EKF (ekf_window.hip): runs of frames whose fused landmarks stay inside one set S --------------------------
constexpr int kWinFrames = 64;         // frames per window at most
constexpr int kWinPieceMax = 16;       // frames per chain piece at most (its step table lives in LDS)
constexpr int kWinCorrMax = 63;        // corrections fused per window frame at most (one lane of the prepare wave each)
constexpr int kWinSMax = 63;           // landmarks in a window's set S at most: 3 + 3 * 63 = 192 = 12 MFMA tiles
constexpr int kWinHdr = 24;            // doubles of scalar header per logged step (after the 3 operand rows)
struct WinFrame {                      // host-planned bookkeeping of one window frame (aruco_slam.cpp:92-95, 192-198, 423-435 replayed on the host)
    int m;                             // corrections fused, in pop order = ascending landmark index (aruco_slam.h:85-88)
    int npop;                          // popped observations: the m corrections + the "stationary" no-ops
    int n_markers;                     // detections of the frame (statistics)
    int pad;
    unsigned char cdet[64];            // correction a: index of its detection in the frame's observation list
    unsigned char cpos[64];            // ... and the position of its landmark in S (rows 3 + 3 pos .. 5 + 3 pos of the S block)
    unsigned char pdet[64];            // popped observation i: detection index
    unsigned char pact[64];            // ... 1 = update, 2 = stationary no-op
    short pidx[64];                    // ... landmark index
};
struct WinDesc {                       // one chain piece / one window (kernel argument)
    int first_slot, K;                 // slots first_slot .. first_slot + K - 1
    int nS, T;                         // landmarks in S; MFMA tiles per side: SP = 16 T >= s = 3 + 3 nS  (T = 4, 8 or 12)
    int piece, log0;                   // index of the piece within its window; steps logged by the earlier pieces
    int last;                          // 1 = the window's last piece: mu_S goes back to the state, the last frame's pop list / last-observation list are left behind
    int wpar;                          // parity of the window within its batch: which of the two P / mu hand-over images it uses
    int from_image, pad;               // first piece of a window whose P and mu_S were prepared in the image (k_ekf_win_next_*) instead of read from Sigma / mu
    short li[kWinSMax + 1];            // state offset 3 + 3 index of every landmark of S, ascending
};

struct EkfState {
    int max_landmarks, ld;             // ld = 3 + 3*max_landmarks: leading dimension of sigma (column-major)
    double* d_mu;
    double* d_sigma;
    int* d_L;                          // landmarks in the map (N = 3 + 3 L)
    int* d_id2idx;
    int* d_idx2id;
    LastObs* d_last;
    LastObs* d_lastNext;               // staging of the next frame's list while the current one is still read
    int* d_nlast;
    PopRec* d_pop;
    int* d_npop;
    UpdRec* d_upd;
    int* d_m;                          // fused updates this frame
    double *d_V, *d_Wt, *d_T;          // 3m x ld each, row k contiguous
    double *d_Sv, *d_Sw, *d_alpha, *d_gamma, *d_G, *d_g;
    MapRecord* d_maprec;
    double* d_win_log;                 // per window step: operand rows -K^T (3 x SP) + header (ekf_window.hip)
    double* d_win_tlog;                // per window step: t = H Lambda and u = S^-1 t (4 x SP each, 4th row zero)
    double* d_win_small;               // two P images and mu_S images (hand-over between the pieces of a window), Lambda, Psi images (SP x SP) and psi
    double* d_win_next;                // early start of the next window: Y_0 columns S' (Vg), Psi Vg, Lambda Vg, Sigma[S',S'] (each SPm x SPm)
    int* d_win_next_idx;               // ... position in the previous S of every entry of S' (or -1), and nS' (at [SPm])
    int* d_win_sidx;                   // per state index: position in S or -1
    WinFrame* d_win_frames;            // per staged slot: the host's plan of the frame
    int win_sp_max, win_steps_max;     // capacity: largest SP and most steps (frames + corrections) per window
    int* d_slot_stat;                  // per staged slot, written by k_ekf_plan: detections, augments, fused updates, stationary no-ops
    int max_slots;
};

hipError_t ekf_alloc(EkfState& E, int max_landmarks, int max_slots, int max_updates_per_frame);
void ekf_free(EkfState& E);
void launch_ekf_predict_only(hipStream_t st, const EkfState& E, const SlamParams& sp, double wl, double wr, double dt);
void launch_ekf_plan(hipStream_t st, const EkfState& E, const SlamParams& sp, double wl, double wr, double dt, int do_predict,
                     const ObsRaw* obs, const unsigned* n_markers, Counters* ctr, int max_m, int slot);
void launch_ekf_mid(hipStream_t st, const EkfState& E);
void launch_ekf_apply(hipStream_t st, const EkfState& E);
int ekf_fast_max_updates();
int ekf_mid_max_updates();
void launch_ekf_mid64(hipStream_t st, const EkfState& E);
void launch_ekf_update_mfma(hipStream_t st, const EkfState& E, int depth = -1);   // depth >= 0: rows of d_T / d_Wt to contract instead of 3 * *d_m
void launch_ekf_gather(hipStream_t st, const EkfState& E);
void launch_ekf_small(hipStream_t st, const EkfState& E);
void launch_ekf_T(hipStream_t st, const EkfState& E);
void launch_ekf_export_map(hipStream_t st, const EkfState& E);
int ekf_win_tiles(int nS);             // T for a set of nS landmarks (4, 8 or 12)
// one launch of a window: the chain of piece wd (wd.K == 0: none), the replay (scan) of piece s_*, the Psi product of piece q_*
// (nsteps == 0: none); obs / enc: the context's per-slot arrays
void launch_ekf_win_step(hipStream_t st, const EkfState& E, const SlamParams& sp, const WinDesc& wd, const ObsRaw* obs, const double* enc,
                         int s_piece, int s_log0, int s_nsteps, int q_piece, int q_log0, int q_nsteps);
void launch_ekf_win_gather(hipStream_t st, const EkfState& E, const WinDesc& wd);               // Y_0 = rows S of Sigma, position table
// P and mu_S of the NEXT window (set nx) from the previous window's (pv) P_K, Lambda, Psi, psi, Y_0 and the not yet flushed Sigma / mu
void launch_ekf_win_next(hipStream_t st, const EkfState& E, const WinDesc& pv, const WinDesc& nx);
void launch_ekf_win_flush(hipStream_t st, const EkfState& E, const WinDesc& wd);                 // thin products, Sigma pass, rows / columns of S

} // namespace aslam
