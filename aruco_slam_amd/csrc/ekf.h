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

// ---- windowed EKF (ekf_window.hip): runs of frames that fuse the same landmarks --------------------------------------
constexpr int kWinM = 20;              // landmarks per window frame at most (s = 3 + 3 m <= 63 fits 64 x 64 images)
constexpr int kWinFrames = 64;         // frames per window at most
struct WinDesc {                       // one window (kernel argument)
    int first_slot, K, m, s;           // slots first_slot .. first_slot + K - 1; s = 3 + 3 m
    int cont, log0;                    // a run (frames on the same landmarks) is cut into chains of a few frames: cont = index of the
                                       // piece within its run (0 = first; for the flush: of the LAST piece), log0 = frames already logged
    int li[kWinM];                     // state offset 3 + 3 index of every landmark, ascending (= pop order, aruco_slam.h:85-88)
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
    double* d_win_log;                 // per window frame: G, W, V images, g, H3, Jacobians (ekf_window.hip)
    double* d_win_small;               // Lambda, Gamma, Psi, P_K images and psi
    int* d_win_sidx;                   // per state index: position in S or -1
    int* d_slot_stat;                  // per staged slot, written by k_ekf_plan: detections, augments, fused updates, stationary no-ops
    int max_slots;
};

hipError_t ekf_alloc(EkfState& E, int max_landmarks, int max_slots);
void ekf_free(EkfState& E);
void launch_ekf_predict_only(hipStream_t st, const EkfState& E, const SlamParams& sp, double wl, double wr, double dt);
void launch_ekf_plan(hipStream_t st, const EkfState& E, const SlamParams& sp, double wl, double wr, double dt, int do_predict,
                     const ObsRaw* obs, const unsigned* n_markers, Counters* ctr, int max_m, int slot);
void launch_ekf_mid(hipStream_t st, const EkfState& E);
void launch_ekf_apply(hipStream_t st, const EkfState& E);
int ekf_fast_max_updates();
int ekf_mid_max_updates();
void launch_ekf_mid64(hipStream_t st, const EkfState& E);
void launch_ekf_update_mfma(hipStream_t st, const EkfState& E);
void launch_ekf_gather(hipStream_t st, const EkfState& E);
void launch_ekf_small(hipStream_t st, const EkfState& E);
void launch_ekf_T(hipStream_t st, const EkfState& E);
void launch_ekf_export_map(hipStream_t st, const EkfState& E);
size_t ekf_win_log_doubles();
size_t ekf_win_small_doubles();
// obs / n_markers / enc: the context's per-slot arrays; d_obs_idx: K x kWinM bytes, detection index of the j-th popped observation
void launch_ekf_win_chain(hipStream_t st, const EkfState& E, const SlamParams& sp, const WinDesc& wd, const ObsRaw* obs,
                          const unsigned* n_markers, const double* enc, const unsigned char* d_obs_idx);
void launch_ekf_win_scan(hipStream_t st, const EkfState& E, const WinDesc& wd);
void launch_ekf_win_flush(hipStream_t st, const EkfState& E, const WinDesc& wd);

} // namespace aslam
