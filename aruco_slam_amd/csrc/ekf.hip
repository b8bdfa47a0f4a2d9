// SE(2) EKF-SLAM on gfx950, state resident in HBM (mu: N doubles, Sigma: N x N doubles column-major with a
// fixed leading dimension so the map grows in place).  Replaces the Eigen arithmetic of
//   ArucoSlam::addEncoder  (src/aruco_slam.cpp:21-74)    -> predict part of k_ekf_plan / k_ekf_predict
//   ArucoSlam::addImage    (src/aruco_slam.cpp:88-263)   -> k_ekf_plan (queue order, augment, update plan)
//                                                           + k_ekf_gather / k_ekf_small / k_ekf_T / k_ekf_update
// The reference applies the M updates of a frame one after another with dense N x N x N products.  Every
// update is linearised at the same frozen pre-frame mean (aruco_slam.cpp:88), so the M sequential rank-3
// corrections compose EXACTLY into one rank-3M correction
//     Sigma <- Sigma0 - W G V ,   mu <- mu + W g ,
// with V = H Sigma0 (3M x N, from rows), W = Sigma0 H^T (N x 3M, from columns) and small 3M x 3M factors G, g
// obtained by replaying the sequential recurrences on 3x3 blocks (k_ekf_small).  Sigma is then streamed once
// per frame (k_ekf_update) instead of 2M times.  Observations that take the reference's "stationary" no-op
// branch (aruco_slam.cpp:192-198) are excluded; new landmarks are appended first (they pop first).
#include "common.h"
#include "ekf.h"
#include "ekf_dev.h"
#include <cmath>
#include <algorithm>

namespace aslam {

__device__ void inv3_pp(const double* A, double* out) {   // 3x3 inverse by partial-pivot LU (Eigen dynamic .inverse())
    double a[3][6];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { a[i][j] = A[i * 3 + j]; a[i][3 + j] = (i == j) ? 1. : 0.; }
    for (int col = 0; col < 3; col++) {
        int piv = col;
        double best = fabs(a[col][col]);
        for (int r = col + 1; r < 3; r++)
            if (fabs(a[r][col]) > best) { best = fabs(a[r][col]); piv = r; }
        if (piv != col)
            for (int c = 0; c < 6; c++) { double t = a[piv][c]; a[piv][c] = a[col][c]; a[col][c] = t; }
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

// ---- predict (aruco_slam.cpp:35-73), executed by one workgroup -------------------------------------------
// Sigma <- Hx Sigma Hx^T + F Qk F^T with Hx = identity except its 3x3 corner.  Rows 0..2 are multiplied by H from the left,
// columns 0..2 by H^T from the right; outside the 3x3 corner the two touch disjoint entries, so one pass does both
// (thread t >= 3: entries (0..2, t) and (t, 0..2); thread 0: the corner (H S3) H^T + Qk).  sMu receives the new pose.
__device__ void predict_block(const EkfState& E, const SlamParams& sp, double wl, double wr, double dt, int N,
                              double* sH /*shared 9*/, double* sQ /*shared 9*/, double* sMu /*shared 5: pose, then sin / cos of the new heading*/) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int ld = E.ld;
    // every load of the pass is issued before the first dependent instruction: the three pose rows / columns of this
    // thread's landmark columns (up to kPredCols of them in registers) and, on thread 0, the pose block itself
    constexpr int kPredCols = 4;
    double ca[kPredCols], cb[kPredCols], cc[kPredCols], ra[kPredCols], rb[kPredCols], rc[kPredCols];
#pragma unroll
    for (int k = 0; k < kPredCols; k++) {
        const int t = 3 + tid + k * nt;
        if (t < N) {
            const double* col = E.d_sigma + (size_t)t * ld;
            ca[k] = col[0]; cb[k] = col[1]; cc[k] = col[2];
            ra[k] = E.d_sigma[t]; rb[k] = E.d_sigma[(size_t)ld + t]; rc[k] = E.d_sigma[(size_t)2 * ld + t];
        }
    }
    double S[9];
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) S[i * 3 + j] = E.d_sigma[(size_t)j * ld + i];
    }
    double pc = 0.0, ps = 0.0;                                    // thread 0: cos / sin of the mid-step heading, kept for Qk
    if (tid == 0) {
        double delta_enl = dt * wl, delta_enr = dt * wr;
        double delta_sl = sp.kl * delta_enl, delta_sr = sp.kr * delta_enr;
        double l_ = 2 * sp.b;
        double delta_theta = (delta_sr - delta_sl) / l_;
        double delta_s = 0.5 * (delta_sr + delta_sl);
        const double m0 = E.d_mu[0], m1 = E.d_mu[1], m2 = E.d_mu[2];
        double tmp_th = m2 + 0.5 * delta_theta;
        double c, s;
        sincos(tmp_th, &s, &c);
        pc = c; ps = s;
        double th = m2 + delta_theta;
        wrap1(th);
        sMu[0] = m0 + delta_s * c; sMu[1] = m1 + delta_s * s; sMu[2] = th;
        sH[0] = 1.0; sH[1] = 0.0; sH[2] = -delta_s * s;
        sH[3] = 0.0; sH[4] = 1.0; sH[5] = delta_s * c;
        sH[6] = 0.0; sH[7] = 0.0; sH[8] = 1.0;
    }
    __syncthreads();                                              // sH / sMu: all the other threads need
    if (tid == 0) {
        // off the other threads' path: the new pose, the process noise and the pose block H S H^T + Q (only thread 0 ever
        // touches these entries; the callers' later barriers order them before anything reads Sigma again)
        E.d_mu[0] = sMu[0]; E.d_mu[1] = sMu[1]; E.d_mu[2] = sMu[2];
        sincos(sMu[2], &sMu[3], &sMu[4]);                        // for the update records (same value every thread would compute)
        const double c = pc, s = ps;
        const double f = 0.5 * sp.kl * dt;                       // kl for BOTH wheels (quirk Q7)
        double wkh[6] = {f * c, f * c, f * s, f * s, f * (1 / sp.b), f * (-1 / sp.b)};
        double su0 = sp.Q_k * fabs(wl), su1 = sp.Q_k * fabs(wr);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) sQ[i * 3 + j] = wkh[i * 2] * su0 * wkh[j * 2] + wkh[i * 2 + 1] * su1 * wkh[j * 2 + 1];
        double T[9];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) T[i * 3 + j] = sH[i * 3] * S[j] + sH[i * 3 + 1] * S[3 + j] + sH[i * 3 + 2] * S[6 + j];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                E.d_sigma[(size_t)j * ld + i] = (T[i * 3] * sH[j * 3] + T[i * 3 + 1] * sH[j * 3 + 1] + T[i * 3 + 2] * sH[j * 3 + 2]) + sQ[i * 3 + j];
    }
#pragma unroll
    for (int k = 0; k < kPredCols; k++) {
        const int t = 3 + tid + k * nt;
        if (t < N) {
            double* col = E.d_sigma + (size_t)t * ld;
            col[0] = sH[0] * ca[k] + sH[1] * cb[k] + sH[2] * cc[k];
            col[1] = sH[3] * ca[k] + sH[4] * cb[k] + sH[5] * cc[k];
            col[2] = sH[6] * ca[k] + sH[7] * cb[k] + sH[8] * cc[k];
            E.d_sigma[t] = ra[k] * sH[0] + rb[k] * sH[1] + rc[k] * sH[2];
            E.d_sigma[(size_t)ld + t] = ra[k] * sH[3] + rb[k] * sH[4] + rc[k] * sH[5];
            E.d_sigma[(size_t)2 * ld + t] = ra[k] * sH[6] + rb[k] * sH[7] + rc[k] * sH[8];
        }
    }
    for (int t = 3 + tid + kPredCols * nt; t < N; t += nt) {         // larger maps: the remaining columns
        double* col = E.d_sigma + (size_t)t * ld;
        const double a = col[0], b = col[1], c = col[2];
        const double xa = E.d_sigma[t], xb = E.d_sigma[(size_t)ld + t], xc = E.d_sigma[(size_t)2 * ld + t];
        col[0] = sH[0] * a + sH[1] * b + sH[2] * c;
        col[1] = sH[3] * a + sH[4] * b + sH[5] * c;
        col[2] = sH[6] * a + sH[7] * b + sH[8] * c;
        E.d_sigma[t] = xa * sH[0] + xb * sH[1] + xc * sH[2];
        E.d_sigma[(size_t)ld + t] = xa * sH[3] + xb * sH[4] + xc * sH[5];
        E.d_sigma[(size_t)2 * ld + t] = xa * sH[6] + xb * sH[7] + xc * sH[8];
    }
    // no barrier here: nothing below reads these entries of Sigma from another thread before the caller's next barriers
}

__global__ __launch_bounds__(256) void k_ekf_predict(EkfState E, SlamParams sp, double wl, double wr, double dt) {
    __shared__ double sH[9], sQ[9], sMu[5];
    const int N = 3 + 3 * (*E.d_L);
    predict_block(E, sp, wl, wr, dt, N, sH, sQ, sMu);
}

// ---- plan: predict + queue order + augment + update plan (one workgroup) -----------------------------------
__global__ __launch_bounds__(256) void k_ekf_plan(EkfState E, SlamParams sp, double wl, double wr, double dt, int do_predict,
                                                  const ObsRaw* __restrict__ obs, const unsigned* __restrict__ n_markers,
                                                  Counters* ctr, int max_m, int slot) {
    __shared__ double sH[9], sQ[9], sMu[5];
    __shared__ ObsRaw sObs[kMarkerMax];
    __shared__ LastObs sLast[kMarkerMax];
    __shared__ int sHeap[kMarkerMax];          // heap of observation slots
    __shared__ int sIndex[kMarkerMax];         // aruco_index_ per slot (-1 new, -2 dropped by the gates)
    __shared__ int sOrder[kMarkerMax];         // pop order
    __shared__ int sAction[kMarkerMax];        // per popped observation
    __shared__ int sUpdPos[kMarkerMax];        // position in the fused update list (-1 = none)
    __shared__ int sNPop, sL, sM, sNNew, sDup, sWaveCnt[2], sStatCnt[2], sCntNew[2], sCntPop[2];
    __shared__ double sLm[kMarkerMax][3];       // landmark mean per observation slot
    __shared__ double sG[9], sMM[9], sNew[3];
    __shared__ int sDoAug;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int ld = E.ld;

    // every independent global load is issued before the first dependent one: counts, this thread's observation and
    // previous-frame entry, the old pose
    const int L0 = *E.d_L;
    const int nM = (int)min(*n_markers, (unsigned)kMarkerMax);
    const int nl = min(*E.d_nlast, kMarkerMax);
    ObsRaw myObs;
    LastObs myLast;
    if (tid < kMarkerMax) { myObs = obs[tid]; myLast = E.d_last[tid]; }          // slots beyond nM / nl are never used
    double mu0x = E.d_mu[0], mu0y = E.d_mu[1], mu0t = E.d_mu[2];
    int myIndex = -2;                                                            // checkLandmark (aruco_slam.cpp:423-435), in flight during the predict
    if (tid < nM && myObs.valid) myIndex = (myObs.id >= 0 && myObs.id < kIdTableSize) ? E.d_id2idx[myObs.id] : -1;
    double lmx = 0, lmy = 0, lmt = 0;                                            // my landmark's mean (untouched by the predict)
    if (myIndex >= 0) { const double* lm = E.d_mu + 3 + 3 * myIndex; lmx = lm[0]; lmy = lm[1]; lmt = lm[2]; }
    if (do_predict) {
        predict_block(E, sp, wl, wr, dt, 3 + 3 * L0, sH, sQ, sMu);
        mu0x = sMu[0]; mu0y = sMu[1]; mu0t = sMu[2];                             // frozen pre-frame robot pose (Q1)
    } else if (tid == 0) {
        sincos(mu0t, &sMu[3], &sMu[4]);
    }
    if (tid < nM) {
        sObs[tid] = myObs;
        sIndex[tid] = myIndex;
        sLm[tid][0] = lmx; sLm[tid][1] = lmy; sLm[tid][2] = lmt;
    }
    if (tid < nl) sLast[tid] = myLast;
    {
        // how many observations are new / enter the queue: one ballot per wavefront (observations sit in threads 0..127)
        const unsigned long long bNew = __ballot(tid < nM && myIndex == -1), bPop = __ballot(tid < nM && myIndex != -2);
        if (tid < kMarkerMax && (tid & 63) == 0) { sCntNew[tid >> 6] = __popcll(bNew); sCntPop[tid >> 6] = __popcll(bPop); }
    }
    if (tid == 0) sDup = 0;
    __syncthreads();
    if (tid == 0) { sNNew = sCntNew[0] + sCntNew[1]; sNPop = sCntPop[0] + sCntPop[1]; }     // read after the barriers below
    if (sCntNew[0] + sCntNew[1] == 0) {
        // Steady state (every marker already mapped): all keys of the priority queue are distinct unless one id was
        // detected twice, so the pop order is simply ascending landmark index -> rank in parallel.
        for (int i = tid; i < nM; i += nt) {
            const int ki = sIndex[i];
            if (ki < 0) continue;
            int rank = 0;
            for (int j = 0; j < nM; j++) {
                const int kj = sIndex[j];
                rank += (kj >= 0 && kj < ki);
                if (j != i && kj == ki) sDup = 1;
            }
            sOrder[rank] = i;
        }
    }
    __syncthreads();
    if (tid == 0 && (sNNew > 0 || sDup)) {
        // obs_.push(ob) in detection order (aruco_slam.cpp:369-373): libstdc++ std::priority_queue = push_heap
        // with operator< inverted on aruco_index_ (aruco_slam.h:85-88): new markers (-1) first, then ascending index;
        // equal keys come out in heap order, which is reproduced by running the very same push_heap / pop_heap steps.
        int len = 0, nnew = 0;
        for (int i = 0; i < nM; i++) {
            if (sIndex[i] == -2) continue;
            nnew += sIndex[i] < 0;
            int hole = len++, value = i;
            int parent = (hole - 1) / 2;
            while (hole > 0 && sIndex[sHeap[parent]] > sIndex[value]) {            // comp(parent, value) = parent < value
                sHeap[hole] = sHeap[parent];
                hole = parent;
                parent = (hole - 1) / 2;
            }
            sHeap[hole] = value;
        }
        int np = 0;
        while (len > 0) {
            sOrder[np++] = sHeap[0];                                               // top()
            if (len > 1) {                                                         // pop_heap -> __adjust_heap
                int value = sHeap[len - 1];
                sHeap[len - 1] = sHeap[0];
                int n = len - 1, hole = 0, second = 0;
                while (second < (n - 1) / 2) {
                    second = 2 * (second + 1);
                    if (sIndex[sHeap[second]] > sIndex[sHeap[second - 1]]) second--;
                    sHeap[hole] = sHeap[second];
                    hole = second;
                }
                if ((n & 1) == 0 && second == (n - 2) / 2) {
                    second = 2 * (second + 1);
                    sHeap[hole] = sHeap[second - 1];
                    hole = second - 1;
                }
                int parent = (hole - 1) / 2;
                while (hole > 0 && sIndex[sHeap[parent]] > sIndex[value]) {
                    sHeap[hole] = sHeap[parent];
                    hole = parent;
                    parent = (hole - 1) / 2;
                }
                sHeap[hole] = value;
            }
            len--;
        }
        sNPop = np;
        sNNew = nnew;
    }
    if (tid == 0) { sL = L0; sM = 0; }
    __syncthreads();
    const int np = sNPop;
    const int nnew = sNNew;

    // ---- new landmarks first (they pop first): strictly sequential (aruco_slam.cpp:208-260) ----
    for (int q = 0; q < nnew; q++) {
        const int slot = sOrder[q];
        if (tid == 0) {
            const ObsRaw o = sObs[slot];
            sAction[q] = 0;
            if (sL >= E.max_landmarks) {
                atomicOr(&ctr->overflow, (unsigned)kOvfLandmarks);
                sDoAug = 0;
            } else {
                sDoAug = 1;
                float sinth = (float)sin(mu0t);                               // float trig (quirk Q4)
                float costh = (float)cos(mu0t);
                double map_x = mu0x + costh * o.x - sinth * o.y;
                double map_y = mu0y + sinth * o.x + costh * o.y;
                double map_theta = mu0t + o.th;
                wrap1(map_theta);
                sNew[0] = map_x; sNew[1] = map_y; sNew[2] = map_theta;
                double deltax = map_x - mu0x, deltay = map_y - mu0y;
                double Gsk[9] = {-costh, -sinth, -sinth * deltax + costh * deltay,
                                 sinth, -costh, -deltax * costh - deltay * sinth,
                                 0, 0, -1};
                double Gmi[9] = {costh, sinth, 0, -sinth, costh, 0, 0, 0, 1};
                double ss[9], T1[9], T2[9], T3[9];
                for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) ss[i * 3 + j] = E.d_sigma[(size_t)j * ld + i];
                // sigma_mm = Gmi * (Gsk*sigma_s*Gsk^T + Rk)^T * Gmi^T   (quirk Q5, literal)
                for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
                    double s = 0; for (int k = 0; k < 3; k++) s += Gsk[i * 3 + k] * ss[k * 3 + j]; T1[i * 3 + j] = s; }
                for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
                    double s = 0; for (int k = 0; k < 3; k++) s += T1[i * 3 + k] * Gsk[j * 3 + k];
                    T2[i * 3 + j] = s + (i == j ? o.r[i] : 0.0); }
                for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
                    double s = 0; for (int k = 0; k < 3; k++) s += Gmi[i * 3 + k] * T2[j * 3 + k]; T3[i * 3 + j] = s; }
                for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
                    double s = 0; for (int k = 0; k < 3; k++) s += T3[i * 3 + k] * Gmi[j * 3 + k]; sMM[i * 3 + j] = s; }
                // sigma_mx = (-Gmi * Gsk) * sigma_.topRows(3)
                for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
                    double s = 0; for (int k = 0; k < 3; k++) s += (-Gmi[i * 3 + k]) * Gsk[k * 3 + j]; sG[i * 3 + j] = s; }
            }
        }
        __syncthreads();
        if (sDoAug) {
            const int N = 3 + 3 * sL;
            for (int c = tid; c < N; c += nt) {
                const double* col = E.d_sigma + (size_t)c * ld;
                double a = col[0], b = col[1], d = col[2];
                for (int i = 0; i < 3; i++) {
                    double v = sG[i * 3] * a + sG[i * 3 + 1] * b + sG[i * 3 + 2] * d;
                    E.d_sigma[(size_t)c * ld + N + i] = v;            // bottom-left block: sigma_mx
                    E.d_sigma[(size_t)(N + i) * ld + c] = v;          // top-right block: sigma_mx^T
                }
            }
            __syncthreads();
            if (tid == 0) {
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 3; j++) E.d_sigma[(size_t)(N + j) * ld + N + i] = sMM[i * 3 + j];
                E.d_mu[N] = sNew[0]; E.d_mu[N + 1] = sNew[1]; E.d_mu[N + 2] = sNew[2];
                const int id = sObs[slot].id;
                if (id >= 0 && id < kIdTableSize && E.d_id2idx[id] < 0) E.d_id2idx[id] = sL;   // map::insert keeps the first (Q10)
                E.d_idx2id[sL] = id;
                sL = sL + 1;
            }
        }
        __syncthreads();
    }

    // ---- already mapped (aruco_slam.cpp:108-207): every record is independent of the others (frozen mean, Q1) ----
    for (int q = nnew + tid; q < np; q += nt) {
        const ObsRaw o = sObs[sOrder[q]];
        // "stationary" test against the previous frame (aruco_slam.cpp:192-198): a no-op branch (quirk Q2)
        bool stationary = false;
        for (int k = 0; k < nl; k++)
            if (sLast[k].id == o.id) {                                     // std::find: first with the same id
                double d0 = sLast[k].z[0] - o.x, d1 = sLast[k].z[1] - o.y, d2 = sLast[k].z[2] - o.th;
                stationary = sqrt(d0 * d0 + d1 * d1 + d2 * d2) < 0.01;     // NaN compares false (Q2/Q3)
                break;
            }
        sAction[q] = stationary ? 2 : 1;
    }
    __syncthreads();
    {
        // position of every update in the fused list = number of updates popped before it (np <= kMarkerMax = 2 wavefronts)
        const bool upd = tid < np && tid >= nnew && sAction[tid] == 1;
        const unsigned long long bal = __ballot(upd);
        const unsigned long long balStat = __ballot(tid < np && tid >= nnew && sAction[tid] == 2);
        if (tid < kMarkerMax && (tid & 63) == 0) { sWaveCnt[tid >> 6] = __popcll(bal); sStatCnt[tid >> 6] = __popcll(balStat); }
        __syncthreads();
        const int before = (tid >= 64 ? sWaveCnt[0] : 0) + __popcll(bal & ((1ull << (tid & 63)) - 1ull));
        const int m = sWaveCnt[0] + sWaveCnt[1];
        if (tid < np) sUpdPos[tid] = (upd && m <= max_m) ? before : -1;
        if (tid == 0) {
            if (m > max_m) atomicOr(&ctr->overflow, (unsigned)kOvfUpdates);   // more fused updates than the configured chain handles: reported, never silent
            sM = m <= max_m ? m : 0;
        }
    }
    __syncthreads();
    for (int q = tid; q < np; q += nt) {
        const int slot = sOrder[q];
        const ObsRaw o = sObs[slot];
        const int index = sIndex[slot];
        PopRec pr;
        pr.id = o.id; pr.index = index; pr.action = sAction[q]; pr.pad = 0;
        pr.z[0] = o.x; pr.z[1] = o.y; pr.z[2] = o.th;
        pr.r[0] = o.r[0]; pr.r[1] = o.r[1]; pr.r[2] = o.r[2];
        E.d_pop[q] = pr;
        // last_observed_marker_ = observed_marker (aruco_slam.cpp:263): last_observation_ is only ever set in the
        // update branch (:202); everywhere else it stays unset -> NaN sentinel
        LastObs lo;
        lo.id = o.id; lo.pad = 0;
        if (sAction[q] == 1) { lo.z[0] = o.x; lo.z[1] = o.y; lo.z[2] = o.th; }
        else { lo.z[0] = lo.z[1] = lo.z[2] = nan(""); }
        E.d_last[q] = lo;            // every thread finished reading the previous list (LDS copy) before the barrier above
        const int up = sUpdPos[q];
        if (up >= 0) {
            const int li = 3 + 3 * index;
            const double mx = sLm[slot][0], my = sLm[slot][1], mth = sLm[slot][2];   // mu_[li..li+2], loaded at kernel start
            const double sintheta = sMu[3], costheta = sMu[4];                   // sin / cos of mu0t, thread 0 (barriers in between)
            double gdx = mx - mu0x, gdy = my - mu0y, gdth = mth - mu0t;
            wrap1(gdth);
            double zh0 = gdx * costheta + gdy * sintheta, zh1 = -gdx * sintheta + gdy * costheta;
            UpdRec u;
            u.li = li; u.pad = 0;
            u.ze[0] = o.x - zh0; u.ze[1] = o.y - zh1; u.ze[2] = o.th - gdth;
            wrap1(u.ze[2]);
            const double G[18] = {-costheta, -sintheta, -gdx * sintheta + gdy * costheta, costheta, sintheta, 0,
                                  sintheta, -costheta, -gdx * costheta - gdy * sintheta, -sintheta, costheta, 0,
                                  0, 0, -1, 0, 0, 1};
            for (int k = 0; k < 18; k++) u.Gxm[k] = G[k];
            u.r[0] = o.r[0]; u.r[1] = o.r[1]; u.r[2] = o.r[2];
            E.d_upd[up] = u;
        }
    }
    if (tid == 0) {
        *E.d_nlast = np;
        *E.d_npop = np;
        *E.d_L = sL;
        *E.d_m = sM;
        if (slot >= 0 && slot < E.max_slots) {          // what this frame did (aslam_get_slot_ekf_stats): detections, augments, fused updates, no-ops
            int* st = E.d_slot_stat + 4 * slot;
            st[0] = nM; st[1] = sL - L0; st[2] = sM; st[3] = sStatCnt[0] + sStatCnt[1];
        }
    }
}

// ---- gather: V = H Sigma0 (rows), W = Sigma0 H^T (columns) for state column / row t, updates k = slice, slice + nslices, ...
__device__ __forceinline__ void gather_vw(const EkfState& E, int t, int N, int m, int slice, int nslices) {
    const int ld = E.ld;
    if (t >= N) return;
    const double* col = E.d_sigma + (size_t)t * ld;           // column t: Sigma(:, t)
    const double c0 = col[0], c1 = col[1], c2 = col[2];
    const double r0 = E.d_sigma[t], r1 = E.d_sigma[(size_t)ld + t], r2 = E.d_sigma[(size_t)2 * ld + t];   // Sigma(t, 0..2)
    for (int k = slice; k < m; k += nslices) {
        const UpdRec& u = E.d_upd[k];
        const int li = u.li;
        const double l0 = col[li], l1 = col[li + 1], l2 = col[li + 2];
        const double q0 = E.d_sigma[(size_t)li * ld + t], q1 = E.d_sigma[(size_t)(li + 1) * ld + t],
                     q2 = E.d_sigma[(size_t)(li + 2) * ld + t];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const double* g = &u.Gxm[a * 6];
            E.d_V[(size_t)(3 * k + a) * ld + t] = g[0] * c0 + g[1] * c1 + g[2] * c2 + g[3] * l0 + g[4] * l1 + g[5] * l2;
            E.d_Wt[(size_t)(3 * k + a) * ld + t] = r0 * g[0] + r1 * g[1] + r2 * g[2] + q0 * g[3] + q1 * g[4] + q2 * g[5];
        }
    }
}

// grid (columns / 256, update slices)
__global__ __launch_bounds__(256) void k_ekf_gather(EkfState E) {
    gather_vw(E, blockIdx.x * 256 + threadIdx.x, 3 + 3 * (*E.d_L), *E.d_m, blockIdx.y, gridDim.y);
}

// ---- small: replay the sequential recurrences on 3x3 blocks -> G (3m x 3m), g (3m) -------------------------
// With HP_i = sum_l alpha_il V_l and PH_i = sum_l W_l beta_li (alpha block lower-, beta block upper-triangular,
// identity diagonals), S_i = HP_i H_i^T + R_i, K_i = PH_i S_i^-1:
//   C_ji = HP_j H_i^T = sum_{l<=j} alpha_jl Sv_li        D_ij = H_i K_j = sum_{l<=j} Sw_il gamma_lj
//   alpha_i. = e_i - sum_{j<i} D_ij alpha_j.             beta_.i = e_i - sum_{j<i} gamma_.j C_ji
//   gamma_.i = beta_.i S_i^-1                            G = sum_i gamma_.i alpha_i.      g = sum_i gamma_.i ze_i
// where Sv_li = V_l H_i^T and Sw_il = H_i W_l are the 3x3 blocks of H Sigma0 H^T taken from rows / columns.
__device__ void ekf_small_general(const EkfState& E, double* scratch /* >= 2*9*kMarkerMax + 9*kMarkerMax + 9 doubles of LDS */) {
    double* sC = scratch;
    double* sD = scratch + kMarkerMax * 9;
    double* sBeta = scratch + 2 * kMarkerMax * 9;
    double* sSinv = scratch + 3 * kMarkerMax * 9;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int m = *E.d_m;
    const int n3 = 3 * m;
    const int ld = E.ld;
    double* Sv = E.d_Sv; double* Sw = E.d_Sw; double* al = E.d_alpha; double* ga = E.d_gamma;

    // Sv[(3l+a)][(3i+b)] = (V_l H_i^T)[a][b] ;  Sw[(3i+a)][(3l+b)] = (H_i W_l)[a][b]
    for (int p = tid; p < n3 * n3; p += nt) {
        const int rr = p / n3, cc = p - rr * n3;
        {
            const int i = cc / 3, b = cc - 3 * i;
            const UpdRec& u = E.d_upd[i];
            const double* v = E.d_V + (size_t)rr * ld;
            const double* g = &u.Gxm[b * 6];
            Sv[p] = v[0] * g[0] + v[1] * g[1] + v[2] * g[2] + v[u.li] * g[3] + v[u.li + 1] * g[4] + v[u.li + 2] * g[5];
        }
        {
            const int i = rr / 3, a = rr - 3 * i;
            const UpdRec& u = E.d_upd[i];
            const double* w = E.d_Wt + (size_t)cc * ld;
            const double* g = &u.Gxm[a * 6];
            Sw[p] = g[0] * w[0] + g[1] * w[1] + g[2] * w[2] + g[3] * w[u.li] + g[4] * w[u.li + 1] + g[5] * w[u.li + 2];
        }
        al[p] = 0.0;
        ga[p] = 0.0;
    }
    __syncthreads();

    for (int i = 0; i < m; i++) {
        // (a) C_ji and (b) D_ij for all j < i
        for (int p = tid; p < 9 * i; p += nt) {
            const int j = p / 9, ab = p - 9 * j, a = ab / 3, b = ab - 3 * a;
            double c = 0, d = 0;
            for (int q = 0; q < 3 * j + 3; q++) {
                c += al[(size_t)(3 * j + a) * n3 + q] * Sv[(size_t)q * n3 + 3 * i + b];
                d += Sw[(size_t)(3 * i + a) * n3 + q] * ga[(size_t)q * n3 + 3 * j + b];
            }
            sC[j * 9 + ab] = c;
            sD[j * 9 + ab] = d;
        }
        __syncthreads();
        // (c) alpha row block i, (d) beta column block i
        const int w3 = 3 * i + 3;
        for (int p = tid; p < 3 * w3; p += nt) {
            const int a = p / w3, c = p - a * w3;
            double s = (c == 3 * i + a) ? 1.0 : 0.0;
            for (int j = c / 3; j < i; j++)
                for (int k = 0; k < 3; k++) s -= sD[j * 9 + a * 3 + k] * al[(size_t)(3 * j + k) * n3 + c];
            al[(size_t)(3 * i + a) * n3 + c] = s;
            // beta[r = c][b = a]
            double t = (c == 3 * i + a) ? 1.0 : 0.0;
            for (int j = c / 3; j < i; j++)
                for (int k = 0; k < 3; k++) t -= ga[(size_t)c * n3 + 3 * j + k] * sC[j * 9 + k * 3 + a];
            sBeta[c * 3 + a] = t;
        }
        __syncthreads();
        // (e) S_i = sum_c alpha[(3i+a)][c] Sv[c][(3i+b)] + R_i ; inverse
        if (tid == 0) {
            double S[9];
            const UpdRec& u = E.d_upd[i];
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) {
                    double s = 0;
                    for (int c = 0; c < w3; c++) s += al[(size_t)(3 * i + a) * n3 + c] * Sv[(size_t)c * n3 + 3 * i + b];
                    S[a * 3 + b] = s + (a == b ? u.r[a] : 0.0);
                }
            inv3_pp(S, sSinv);
        }
        __syncthreads();
        // (f) gamma column block i = beta_.i S_i^-1
        for (int p = tid; p < 3 * w3; p += nt) {
            const int r = p / 3, b = p - 3 * r;
            ga[(size_t)r * n3 + 3 * i + b] = sBeta[r * 3] * sSinv[b] + sBeta[r * 3 + 1] * sSinv[3 + b] + sBeta[r * 3 + 2] * sSinv[6 + b];
        }
        __syncthreads();
    }
    // G = gamma * alpha ; g = gamma * ze
    for (int p = tid; p < n3 * n3; p += nt) {
        const int r = p / n3, c = p - r * n3;
        double s = 0;
        for (int q = (r > c ? r : c) / 3 * 3; q < n3; q++) s += ga[(size_t)r * n3 + q] * al[(size_t)q * n3 + c];
        E.d_G[p] = s;
    }
    for (int r = tid; r < n3; r += nt) {
        double s = 0;
        for (int q = r / 3 * 3; q < n3; q++) s += ga[(size_t)r * n3 + q] * E.d_upd[q / 3].ze[q % 3];
        E.d_g[r] = s;
    }
}

// ---- small (fast path, 3m <= kSmallMax): the innovation matrix in LDS --------------------------------------
// In exact arithmetic the M sequential rank-3 corrections equal ONE batch correction with A = H Sigma0 H^T + blockdiag(R):
// G = A^-1 (the matrix-inversion lemma needs no symmetry).  The reference, however, feeds every update the innovation
// computed at the frozen pre-frame mean (quirk Q1) instead of the running one, i.e. the batch filter sees the pseudo
// innovations nu_i = ze_i + sum_{j<i} (H_i K_j) ze_j.  Both come out of ONE block Gauss-Jordan sweep with 3x3 pivots,
// which is the reference's own recursion: at step i the pivot block IS S_i = H_i Sigma_{i-1} H_i^T + R_i
// (aruco_slam.cpp:146) and the block multiplier of a later row block r IS H_r K_i.  A^-1 is formed in LDS, ping-pong
// between two images (one barrier per pivot block).
constexpr int kSmallMax = 96;            // 3m <= 96 (m <= 32 fused updates) runs out of LDS; larger frames use the general path
constexpr int SMT = 768;                 // threads: 96 columns x 8 row groups
constexpr int SMR = 12;                  // rows per thread = kSmallMax / 8

__device__ __forceinline__ void inv3_cof(const double* P, int n3, double* o) {   // 3x3 inverse (cofactors); P has row stride n3
    const double a = P[0], b = P[1], c = P[2], d = P[n3], e = P[n3 + 1], f = P[n3 + 2], g = P[2 * n3], h = P[2 * n3 + 1], i = P[2 * n3 + 2];
    const double A = e * i - f * h, B = f * g - d * i, C = d * h - e * g;
    const double id = 1.0 / (a * A + b * B + c * C);
    o[0] = A * id; o[1] = (c * h - b * i) * id; o[2] = (b * f - c * e) * id;
    o[3] = B * id; o[4] = (a * i - c * g) * id; o[5] = (c * d - a * f) * id;
    o[6] = C * id; o[7] = (b * g - a * h) * id; o[8] = (a * e - b * d) * id;
}

__global__ __launch_bounds__(768) void k_ekf_small(EkfState E) {
    __shared__ double sA0[kSmallMax * kSmallMax];     // ping
    __shared__ double sA1[kSmallMax * kSmallMax];     // pong
    __shared__ double sZe[kSmallMax], sNu[kSmallMax];
    const int tid = threadIdx.x;
    const int m = *E.d_m;
    const int n3 = 3 * m;
    const int ld = E.ld;
    const int tc = tid % kSmallMax, tr = tid / kSmallMax;     // column, row group (0..7); rows r = tr + 8*i
    if (n3 > kSmallMax) {                              // uniform branch
        ekf_small_general(E, sA0);
    } else if (m > 0) {
        // A[(3i+a)][(3j+b)] = (V_i H_j^T)[a][b] + delta_ij R_i[a][b]   (aruco_slam.cpp:146: (Gx*sigma_)*Gx^T + Rk)
        if (tc < n3) {
            const int j = tc / 3, bq = tc - 3 * j;
            const UpdRec& u = E.d_upd[j];
            const int li = u.li;
            const double g0 = u.Gxm[bq * 6], g1 = u.Gxm[bq * 6 + 1], g2 = u.Gxm[bq * 6 + 2], g3 = u.Gxm[bq * 6 + 3],
                         g4 = u.Gxm[bq * 6 + 4], g5 = u.Gxm[bq * 6 + 5];
            const double rdiag = u.r[bq];
            double v0[SMR], v1[SMR], v2[SMR], v3[SMR], v4[SMR], v5[SMR];
#pragma unroll
            for (int i = 0; i < SMR; i++) {                  // issue every load before the first use
                const int r = tr + 8 * i;
                const double* v = E.d_V + (size_t)(r < n3 ? r : 0) * ld;
                v0[i] = v[0]; v1[i] = v[1]; v2[i] = v[2]; v3[i] = v[li]; v4[i] = v[li + 1]; v5[i] = v[li + 2];
            }
#pragma unroll
            for (int i = 0; i < SMR; i++) {
                const int r = tr + 8 * i;
                if (r < n3) {
                    double a = v0[i] * g0 + v1[i] * g1 + v2[i] * g2 + v3[i] * g3 + v4[i] * g4 + v5[i] * g5;
                    if (r == tc) a += rdiag;
                    sA0[r * n3 + tc] = a;
                }
            }
        }
        if (tid < n3) { const double z = E.d_upd[tid / 3].ze[tid % 3]; sZe[tid] = z; sNu[tid] = z; }
        __syncthreads();
        double* cur = sA0;
        double* nxt = sA1;
        for (int ib = 0; ib < m; ib++) {
            const int k0 = 3 * ib;
            if (tc < n3) {
                double Pi[9];
                inv3_cof(cur + k0 * n3 + k0, n3, Pi);                         // S_i^-1 (every thread, from LDS broadcast reads)
                const bool cin = tc >= k0 && tc < k0 + 3;
                // pivot rows at this column, with the pivot block column replaced by the identity
                const double p0 = cin ? (tc == k0 ? 1.0 : 0.0) : cur[k0 * n3 + tc];
                const double p1 = cin ? (tc == k0 + 1 ? 1.0 : 0.0) : cur[(k0 + 1) * n3 + tc];
                const double p2 = cin ? (tc == k0 + 2 ? 1.0 : 0.0) : cur[(k0 + 2) * n3 + tc];
                const double y0 = Pi[0] * p0 + Pi[1] * p1 + Pi[2] * p2;       // (S_i^-1 * pivot rows)[., tc]
                const double y1 = Pi[3] * p0 + Pi[4] * p1 + Pi[5] * p2;
                const double y2 = Pi[6] * p0 + Pi[7] * p1 + Pi[8] * p2;
                const double z0 = sZe[k0], z1 = sZe[k0 + 1], z2 = sZe[k0 + 2];
#pragma unroll
                for (int i = 0; i < SMR; i++) {
                    const int r = tr + 8 * i;
                    if (r < n3) {
                        if (r >= k0 && r < k0 + 3) {
                            nxt[r * n3 + tc] = r == k0 ? y0 : (r == k0 + 1 ? y1 : y2);
                        } else {
                            const double f0 = cur[r * n3 + k0], f1 = cur[r * n3 + k0 + 1], f2 = cur[r * n3 + k0 + 2];
                            const double old = cin ? 0.0 : cur[r * n3 + tc];
                            nxt[r * n3 + tc] = old - (f0 * y0 + f1 * y1 + f2 * y2);
                            if (tc == k0 && r > k0 + 2) {
                                // block multiplier H_r K_i = F_r S_i^-1 : nu_r += (H_r K_i) ze_i
                                const double m0 = f0 * Pi[0] + f1 * Pi[3] + f2 * Pi[6];
                                const double m1 = f0 * Pi[1] + f1 * Pi[4] + f2 * Pi[7];
                                const double m2 = f0 * Pi[2] + f1 * Pi[5] + f2 * Pi[8];
                                sNu[r] += m0 * z0 + m1 * z1 + m2 * z2;
                            }
                        }
                    }
                }
            }
            __syncthreads();
            double* t = cur; cur = nxt; nxt = t;
        }
        for (int p = tid; p < n3 * n3; p += SMT) E.d_G[p] = cur[p];
        if (tid < n3) {
            double s = 0;
            for (int c = 0; c < n3; c++) s += cur[tid * n3 + c] * sNu[c];
            E.d_g[tid] = s;
        }
    }
}

// ---- T = G V (3m x N) on the f64 matrix cores, and mu += W g -----------------------------------------------------
// Workgroup (x, y) = 32 columns x 64 rows of T; wavefront w = 16 of those rows x 32 columns = two v_mfma_f64_16x16x4_f64
// tiles.  Operands come straight from L2 in MFMA layout (A[i][k] = G[q0 + i][p0 + k] in lane k*16+i, B[k][j] = V[p0 + k][c0 + j]
// in lane k*16+j): G is 3m x 3m (180 KB at m = 50) and V a 3m x 32 panel, both far below the cache sizes, and 94 x 3
// workgroups cover the GPU where 47 scalar ones did not.  Workgroups with y = 0 also add W g to their 32 entries of mu.
constexpr int TC = 32, TR = 64;

__global__ __launch_bounds__(256) void k_ekf_T(EkfState E) {
    __shared__ double sMu[8][TC];
    const int m = *E.d_m;
    const int n3 = 3 * m;
    const int N = 3 + 3 * (*E.d_L);
    const int ld = E.ld;
    const int c0 = blockIdx.x * TC, q0 = blockIdx.y * TR;
    if (m <= 0 || c0 >= N || q0 >= n3) return;                      // uniform per workgroup
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    if (blockIdx.y == 0) {
        // mu += W g: 8 partial sums per column (strided over p), combined in a fixed order
        const int x = tid & 31, pg = tid >> 5;
        double s = 0;
        if (c0 + x < N)
            for (int p = pg; p < n3; p += 8) s += E.d_Wt[(size_t)p * ld + c0 + x] * E.d_g[p];
        sMu[pg][x] = s;
    }
    const int qrow = q0 + 16 * wave + li;                           // A operand row of this lane
    const bool qok = qrow < n3;
    const bool c0ok = c0 + li < N, c1ok = c0 + 16 + li < N;
    v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    const double* gp = E.d_G + (size_t)(qok ? qrow : 0) * n3;
    // software pipeline: the operands of step p0 + 4 are in flight while step p0 multiplies
    double a = (qok && lk < n3) ? gp[lk] : 0.0;
    double b0 = (c0ok && lk < n3) ? E.d_V[(size_t)lk * ld + c0 + li] : 0.0;
    double b1 = (c1ok && lk < n3) ? E.d_V[(size_t)lk * ld + c0 + 16 + li] : 0.0;
    for (int p0 = 0; p0 < n3; p0 += 4) {
        const int pn = p0 + 4 + lk;
        const double an = (qok && pn < n3) ? gp[pn] : 0.0;
        const double b0n = (c0ok && pn < n3) ? E.d_V[(size_t)pn * ld + c0 + li] : 0.0;
        const double b1n = (c1ok && pn < n3) ? E.d_V[(size_t)pn * ld + c0 + 16 + li] : 0.0;
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, acc1, 0, 0, 0);
        a = an; b0 = b0n; b1 = b1n;
    }
    // D rows (lane>>4) + 4*reg = row of T within the wave's 16, column lane&15
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
        const int q = q0 + 16 * wave + lk + 4 * reg;
        if (q < n3) {
            if (c0ok) E.d_T[(size_t)q * ld + c0 + li] = acc0[reg];
            if (c1ok) E.d_T[(size_t)q * ld + c0 + 16 + li] = acc1[reg];
        }
    }
    if (blockIdx.y == 0) {
        __syncthreads();
        if (tid < TC && c0 + tid < N)
            E.d_mu[c0 + tid] += ((sMu[0][tid] + sMu[1][tid]) + (sMu[2][tid] + sMu[3][tid])) +
                                ((sMu[4][tid] + sMu[5][tid]) + (sMu[6][tid] + sMu[7][tid]));   // mu_ += sum_i K_i ze_i (aruco_slam.cpp:203)
    }
}

// =============================================================================================================
// Fast chain (frames with at most kFastM fused updates): plan -> k_ekf_mid -> k_ekf_apply.
//   k_ekf_mid   workgroup 0 forms the innovation matrix from the 3+3m observed rows/columns of Sigma0 only
//               (A = H Sigma0[S,S] H^T + R needs nothing else), inverts it by the block Gauss-Jordan sweep above and
//               emits G, g; the other workgroups gather V = H Sigma0 and W^T concurrently (independent of G).
//   k_ekf_apply one workgroup per 64x64 tile of Sigma: T_tile = G V_tile in LDS, Sigma_tile -= W_tile^T... T_tile, and
//               mu += W g on the tile column 0.  Sigma is read once and written once per frame.
// =============================================================================================================
constexpr int kFastM = 24;               // fused updates per frame handled by the fast chain
constexpr int kFastN3 = 3 * kFastM;      // 72
constexpr int MIDT = 576;                // kFastM x kFastM: one thread per 3x3 block of the innovation matrix

// <= 128 VGPRs (4 waves per SIMD) so that the workgroup always finds room beside the persistent detection waves of the other stream
__global__ __launch_bounds__(576, 4) void k_ekf_mid(EkfState E) {
    __shared__ __align__(16) double sCol[2][kFastM][10];   // pivot column blocks (bi, ib); rows padded to 80 B for 128-bit LDS reads
    __shared__ __align__(16) double sRow[2][kFastM][10];   // pivot row blocks (ib, bj), unscaled
    __shared__ __align__(16) double sPinv[2][10];          // S_ib^-1 (from the owner of the pivot block, one step ahead)
    __shared__ double sZe[kFastN3], sNu[kFastN3];
    __shared__ double sPart[kFastM][kFastM][3];
    const int tid = threadIdx.x;
    // workgroup 0 stages the first kFastM update records in LDS with loads issued TOGETHER with the load of m (they do not
    // depend on it; records past m are stale and never used): one global round trip less on the dependent chain
    constexpr int kUpdDoubles = (int)(sizeof(UpdRec) / sizeof(double));
    __shared__ double sUpdRaw[kFastM * kUpdDoubles];
    double u0 = 0.0, u1 = 0.0;
    if (blockIdx.x == 0) {
        const double* raw = reinterpret_cast<const double*>(E.d_upd);
        u0 = raw[tid];
        if (tid + MIDT < kFastM * kUpdDoubles) u1 = raw[tid + MIDT];
    }
    const int m = *E.d_m;
    const int ld = E.ld;
    if (blockIdx.x == 0) {
        sUpdRaw[tid] = u0;
        if (tid + MIDT < kFastM * kUpdDoubles) sUpdRaw[tid + MIDT] = u1;
    }
    if (m <= 0 || m > kFastM) return;                  // uniform (m > kFastM is reported by k_ekf_plan)
    const int n3 = 3 * m;
    if (blockIdx.x > 0) {
        // ---- gather: V = H Sigma0 (rows), W = Sigma0 H^T (columns) ----
        const int N = 3 + 3 * (*E.d_L);
        const int ncg = (ld + MIDT - 1) / MIDT;
        const int gb = blockIdx.x - 1;
        gather_vw(E, (gb % ncg) * MIDT + tid, N, m, gb / ncg, (gridDim.x - 1) / ncg);
        return;
    }
    // ---- workgroup 0: thread (bi, bj) owns the 3x3 block (bi, bj) of A = H Sigma0 H^T + R in registers ----
    const int bj = tid % m, bi = tid / m;              // dense m x m mapping: the active threads fill whole waves
    const bool act = bi < m && bj < m;
    double A[9];
    __syncthreads();                                   // sUpdRaw complete
    const UpdRec* sUpd = reinterpret_cast<const UpdRec*>(sUpdRaw);
    if (tid < n3) { const double z = sUpd[tid / 3].ze[tid % 3]; sZe[tid] = z; sNu[tid] = z; }
    if (act) {
        const UpdRec& ui = sUpd[bi];
        const UpdRec& uj = sUpd[bj];
        const int li = ui.li, lj = uj.li;
        // the 6x6 block Sigma0[c6_i, c6_j] (c6 = robot triple + landmark triple), every load issued before the first use
        double S[36];
#pragma unroll
        for (int p = 0; p < 6; p++)
#pragma unroll
            for (int q = 0; q < 6; q++) {
                const int r = p < 3 ? p : li + p - 3, c = q < 3 ? q : lj + q - 3;
                S[p * 6 + q] = E.d_sigma[(size_t)c * ld + r];
            }
        double Gi[18], Gj[18];
#pragma unroll
        for (int k = 0; k < 18; k++) { Gi[k] = ui.Gxm[k]; Gj[k] = uj.Gxm[k]; }
        // (Gx * sigma_) * Gx^T + Rk   (aruco_slam.cpp:146)
        double HP[18];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int q = 0; q < 6; q++) {
                double s = 0;
#pragma unroll
                for (int p = 0; p < 6; p++) s += Gi[a * 6 + p] * S[p * 6 + q];
                HP[a * 6 + q] = s;
            }
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) {
                double s = 0;
#pragma unroll
                for (int q = 0; q < 6; q++) s += HP[a * 6 + q] * Gj[b * 6 + q];
                A[a * 3 + b] = s;
            }
        if (bi == bj) { A[0] += ui.r[0]; A[4] += ui.r[1]; A[8] += ui.r[2]; }
        if (bj == 0) { for (int k = 0; k < 9; k++) sCol[0][bi][k] = A[k]; }
        if (bi == 0) { for (int k = 0; k < 9; k++) sRow[0][bj][k] = A[k]; }
        if (bi == 0 && bj == 0) {
            double Pn[9];
            inv3_reg(A, Pn);
            for (int k = 0; k < 9; k++) sPinv[0][k] = Pn[k];
        }
    }
    __syncthreads();
    // Block Gauss-Jordan with 3x3 pivots: pivot block ib IS S_ib = H_ib Sigma_{ib-1} H_ib^T + R_ib, and the block multiplier
    // F S_ib^-1 of a later row block IS H_bi K_ib (the reference's own recursion, aruco_slam.cpp:146,204).
    // One barrier per step: a step publishes the NEXT pivot's row blocks R_bj = A(ib+1, bj), column blocks F_bi = A(bi, ib+1)
    // and inverse (from the block's owner); in the next step every thread scales its own column's row block itself,
    // Y_bj = S^-1 R_bj (27 flops, redundant down each column) instead of waiting for the pivot row to do it.
    for (int ib = 0; ib < m; ib++) {
        const int cb = ib & 1;
        if (act) {
            double Pi[9], Y[9];
#pragma unroll
            for (int k = 0; k < 9; k++) Pi[k] = sPinv[cb][k];
            if (bj == ib) {
#pragma unroll
                for (int k = 0; k < 9; k++) Y[k] = Pi[k];                    // pivot column: Y is S^-1 itself
            } else {
                double R[9];
#pragma unroll
                for (int k = 0; k < 9; k++) R[k] = sRow[cb][bj][k];
                mul3(Pi, R, Y);
            }
            if (bi == ib) {
#pragma unroll
                for (int k = 0; k < 9; k++) A[k] = Y[k];                     // pivot row: S^-1 A(ib, bj), S^-1 on the diagonal
            } else {
                double F[9];
#pragma unroll
                for (int k = 0; k < 9; k++) F[k] = sCol[cb][bi][k];
                if (bj == ib) {
#pragma unroll
                    for (int k = 0; k < 9; k++) A[k] = 0.0;                  // pivot column: becomes 0 - F S^-1
                }
#pragma unroll
                for (int i = 0; i < 3; i++)
#pragma unroll
                    for (int j = 0; j < 3; j++)
                        A[i * 3 + j] = fma(-F[i * 3 + 2], Y[6 + j], fma(-F[i * 3 + 1], Y[3 + j], fma(-F[i * 3], Y[j], A[i * 3 + j])));
                if (bj == ib && bi > ib) {
                    const double z0 = sZe[3 * ib], z1 = sZe[3 * ib + 1], z2 = sZe[3 * ib + 2];
#pragma unroll
                    for (int a = 0; a < 3; a++) sNu[3 * bi + a] -= A[a * 3] * z0 + A[a * 3 + 1] * z1 + A[a * 3 + 2] * z2;   // nu += (H K) ze, H K = -A
                }
            }
            // what step ib + 1 needs
            if (bi == ib + 1) { for (int k = 0; k < 9; k++) sRow[cb ^ 1][bj][k] = A[k]; }
            if (bj == ib + 1) { for (int k = 0; k < 9; k++) sCol[cb ^ 1][bi][k] = A[k]; }
            if (bi == ib + 1 && bj == ib + 1) {
                double Pn[9];
                inv3_reg(A, Pn);
#pragma unroll
                for (int k = 0; k < 9; k++) sPinv[cb ^ 1][k] = Pn[k];
            }
        }
        __syncthreads();
    }
    // G = A^-1 (block (bi, bj) from its owner), g = G nu
    if (act) {
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) E.d_G[(size_t)(3 * bi + a) * kFastN3 + 3 * bj + b] = A[a * 3 + b];   // fixed row stride (k_ekf_apply loads before it knows m)
        const double n0 = sNu[3 * bj], n1 = sNu[3 * bj + 1], n2 = sNu[3 * bj + 2];
#pragma unroll
        for (int a = 0; a < 3; a++) sPart[bi][bj][a] = A[a * 3] * n0 + A[a * 3 + 1] * n1 + A[a * 3 + 2] * n2;
    }
    __syncthreads();
    if (tid < 4 * n3) {                                // four lanes per entry of g: partial sums over j = q, q + 4, ..., then two shuffles
        const int r = tid >> 2, q = tid & 3;
        const int i = r / 3, a = r - 3 * i;
        double s = 0;
        for (int j = q; j < m; j += 4) s += sPart[i][j][a];
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if (q == 0) E.d_g[r] = s;
    }
}

constexpr int APK = kFastN3;             // padded depth of the LDS images (72)


// f64 matrix cores (v_mfma_f64_16x16x4_f64) for both products of the tile.  Operand layout (cdna_hip_programming.md §3):
// A[i][k] in lane k*16+i, B[k][j] in lane k*16+j, D rows (lane>>4)+4*reg, column lane&15.  The Sigma tile is formed
// transposed, D'[c][r] = sum_p T[p][c] W^T[p][r], so each accumulator register covers 16 consecutive rows r of one
// column c and the read-modify-write of the column-major Sigma is coalesced.
__global__ __launch_bounds__(256) void k_ekf_apply(EkfState E) {
    __shared__ double sGt[APK * APK + 16];    // G transposed: sGt[p][q], row stride APK, zero padded (+16: the 5th q-tile reads 8 past)
    __shared__ double sVW[APK][64];           // V tile, later the W^T tile
    __shared__ double sT[APK + 8][64];        // T tile = G V tile (rows 72..79 belong to the padded 5th q-tile)
    __shared__ double sg[APK];
    const int ld = E.ld;
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;

    // Every global load of the workgroup is issued up front (Sigma tile, G, V tile, W^T tile: ~75 loads per thread in
    // flight) so the kernel pays the dependent-launch memory latency once, not once per staging loop.
    // Sigma tile in the accumulator layout of the update: wave = column tile, ri = row tile, reg
    double sig[4][4];
#pragma unroll
    for (int ri = 0; ri < 4; ri++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int c = c0 + 16 * wave + lk + 4 * reg, r = r0 + 16 * ri + li;
            sig[ri][reg] = (r < ld && c < ld) ? E.d_sigma[(size_t)c * ld + r] : 0.0;
        }
    // None of these addresses depends on m or N (G has the fixed row stride APK, V / W^T rows the stride ld), so they are all
    // in flight together with the loads of m and L; what lies beyond n3 / N is stale and is masked when it is staged.
    constexpr int NG = (APK * APK + 255) / 256;      // 21
    constexpr int NV = APK * 64 / 256;               // 18
    double tg[NG], tv[NV], tw[NV];
#pragma unroll
    for (int k = 0; k < NG; k++) {
        const int i = tid + 256 * k;
        tg[k] = (i < APK * APK) ? E.d_G[i] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < NV; k++) {
        const int i = tid + 256 * k;
        const int pq = i >> 6, xx = i & 63;
        tv[k] = (c0 + xx < ld) ? E.d_V[(size_t)pq * ld + c0 + xx] : 0.0;
        tw[k] = (r0 + xx < ld) ? E.d_Wt[(size_t)pq * ld + r0 + xx] : 0.0;
    }
    const double graw = (tid < APK) ? E.d_g[tid] : 0.0;
    const int m = *E.d_m;
    const int N = 3 + 3 * (*E.d_L);
    if (m <= 0 || m > kFastM || r0 >= N || c0 >= N) return;     // uniform
    const int n3 = 3 * m;
#pragma unroll
    for (int ri = 0; ri < 4; ri++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int c = c0 + 16 * wave + lk + 4 * reg, r = r0 + 16 * ri + li;
            if (!(r < N && c < N)) sig[ri][reg] = 0.0;
        }
#pragma unroll
    for (int k = 0; k < NG; k++) {
        const int i = tid + 256 * k;
        const int q = i / APK, pq = i - q * APK;
        if (i < APK * APK) sGt[pq * APK + q] = (q < n3 && pq < n3) ? tg[k] : 0.0;
    }
    if (tid < 16) sGt[APK * APK + tid] = 0.0;
#pragma unroll
    for (int k = 0; k < NV; k++) {
        const int i = tid + 256 * k;
        const int pq = i >> 6, xx = i & 63;
        if (!(pq < n3 && c0 + xx < N)) tv[k] = 0.0;
        if (!(pq < n3 && r0 + xx < N)) tw[k] = 0.0;
        sVW[pq][xx] = tv[k];
    }
    if (tid < APK) sg[tid] = tid < n3 ? graw : 0.0;
    __syncthreads();

    // T tile (80 x 64, rows >= 72 unused): wave w owns columns 16w .. 16w+15 and all five 16-row tiles
    {
        v4d acc[5];
#pragma unroll
        for (int qi = 0; qi < 5; qi++) acc[qi] = v4d{0.0, 0.0, 0.0, 0.0};
        for (int p0 = 0; p0 < APK; p0 += 4) {
            const double bfrag = sVW[p0 + lk][16 * wave + li];                       // B[k = p][j = x] = V[p][x]
#pragma unroll
            for (int qi = 0; qi < 5; qi++) {
                const double afrag = sGt[(p0 + lk) * APK + 16 * qi + li];           // A[i = q][k = p] = G[q][p]
                acc[qi] = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag, bfrag, acc[qi], 0, 0, 0);
            }
        }
#pragma unroll
        for (int qi = 0; qi < 5; qi++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) sT[16 * qi + lk + 4 * reg][16 * wave + li] = acc[qi][reg];
    }
    __syncthreads();
    // W^T tile (already in registers) replaces the V tile
#pragma unroll
    for (int k = 0; k < NV; k++) { const int i = tid + 256 * k; sVW[i >> 6][i & 63] = tw[k]; }
    __syncthreads();
    // Sigma tile, transposed product: wave w owns the 16 columns c0 + 16w .. and all four 16-row tiles
    v4d acc[4];
#pragma unroll
    for (int ri = 0; ri < 4; ri++) acc[ri] = v4d{0.0, 0.0, 0.0, 0.0};
    for (int p0 = 0; p0 < APK; p0 += 4) {
        const double afrag = sT[p0 + lk][16 * wave + li];                            // A[i = c][k = p] = T[p][c]
#pragma unroll
        for (int ri = 0; ri < 4; ri++) {
            const double bfrag = sVW[p0 + lk][16 * ri + li];                         // B[k = p][j = r] = W^T[p][r]
            acc[ri] = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag, bfrag, acc[ri], 0, 0, 0);
        }
    }
#pragma unroll
    for (int ri = 0; ri < 4; ri++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int c = c0 + 16 * wave + lk + 4 * reg, r = r0 + 16 * ri + li;
            if (r < N && c < N) E.d_sigma[(size_t)c * ld + r] = sig[ri][reg] - acc[ri][reg];
        }
    if (blockIdx.y == 0 && tid < 64 && r0 + tid < N) {
        double s0 = 0, s1 = 0;
        for (int pq = 0; pq < n3; pq += 2) { s0 += sVW[pq][tid] * sg[pq]; s1 += sVW[pq + 1][tid] * sg[pq + 1]; }
        E.d_mu[r0 + tid] += s0 + s1;                              // mu_ += sum_i K_i ze_i (aruco_slam.cpp:203)
    }
}

// =============================================================================================================
// Medium chain (25 .. 64 fused updates per frame, e.g. the 50-marker / 1000-landmark configuration):
//   plan -> k_ekf_mid64 -> k_ekf_T -> k_ekf_update_mfma.
// k_ekf_mid64 is the register-resident block Gauss-Jordan of k_ekf_mid with up to 2 x 2 blocks per thread (32 x 32
// threads); k_ekf_update_mfma is the rank-3M covariance correction on the f64 matrix cores: at N = 3003, M = 50 it IS a
// dense contraction (2.7 GFLOP and 144 MB of Sigma traffic per frame).
// =============================================================================================================
constexpr int kMidM = 64;

constexpr int M64T = 512;                // threads of k_ekf_mid64: 8 wavefronts, 2 per SIMD -> 256 VGPRs each, no spills
constexpr int M64B = (kMidM * kMidM + M64T - 1) / M64T;   // 3x3 blocks per thread (8)

// The fast chain's Gauss-Jordan with up to M64B blocks per thread, dense block index e = tid + M64T k -> (e / m, e % m).
__global__ __launch_bounds__(M64T) void k_ekf_mid64(EkfState E) {
    __shared__ __align__(16) double sCol[2][kMidM][10];
    __shared__ __align__(16) double sY[kMidM][10];
    __shared__ double sPinv[9];
    __shared__ double sZe[3 * kMidM], sNu[3 * kMidM];
    __shared__ double sPart[kMidM][kMidM][3];
    const int tid = threadIdx.x;
    const int m = *E.d_m;
    const int ld = E.ld;
    if (m <= 0 || m > kMidM) return;                   // uniform
    const int n3 = 3 * m;
    if (blockIdx.x > 0) {
        // ---- gather: V = H Sigma0 (rows), W = Sigma0 H^T (columns) ----
        const int N = 3 + 3 * (*E.d_L);
        const int ncg = (ld + M64T - 1) / M64T;
        const int gb = blockIdx.x - 1;
        gather_vw(E, (gb % ncg) * M64T + tid, N, m, gb / ncg, (gridDim.x - 1) / ncg);
        return;
    }
    // ---- workgroup 0: innovation matrix A = H Sigma0 H^T + R, 3x3 blocks in registers ----
    const int nblk = m * m;
    double A[M64B][9];
    int bij[M64B];                                     // (bi << 8) | bj of this thread's blocks, -1 = none (the division is done once)
#pragma unroll
    for (int k = 0; k < M64B; k++) {
        const int e = tid + M64T * k;
        const int bi = e / m;
        bij[k] = e < nblk ? ((bi << 8) | (e - bi * m)) : -1;
    }
    for (int r = tid; r < n3; r += M64T) { const double z = E.d_upd[r / 3].ze[r % 3]; sZe[r] = z; sNu[r] = z; }
#pragma unroll
    for (int k = 0; k < M64B; k++) {
        if (bij[k] >= 0) {
            const int bi = bij[k] >> 8, bj = bij[k] & 255;
            const UpdRec& ui = E.d_upd[bi];
            const UpdRec& uj = E.d_upd[bj];
            const int li = ui.li, lj = uj.li;
            double S[36];
#pragma unroll
            for (int p = 0; p < 6; p++)
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    const int r = p < 3 ? p : li + p - 3, c = q < 3 ? q : lj + q - 3;
                    S[p * 6 + q] = E.d_sigma[(size_t)c * ld + r];
                }
            double HP[18];
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    double acc = 0;
#pragma unroll
                    for (int p = 0; p < 6; p++) acc += ui.Gxm[a * 6 + p] * S[p * 6 + q];
                    HP[a * 6 + q] = acc;
                }
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    double acc = 0;
#pragma unroll
                    for (int q = 0; q < 6; q++) acc += HP[a * 6 + q] * uj.Gxm[c * 6 + q];
                    A[k][a * 3 + c] = acc;
                }
            if (bi == bj) { A[k][0] += ui.r[0]; A[k][4] += ui.r[1]; A[k][8] += ui.r[2]; }
            if (bj == 0) { for (int q = 0; q < 9; q++) sCol[0][bi][q] = A[k][q]; }
            if (bij[k] == 0) {
                double Pn[9];
                inv3_reg(A[k], Pn);
                for (int q = 0; q < 9; q++) sPinv[q] = Pn[q];
            }
        }
    }
    __syncthreads();
    // block Gauss-Jordan with 3x3 pivots (see k_ekf_mid)
    for (int ib = 0; ib < m; ib++) {
        const int cb = ib & 1;
        // phase 1: the pivot row: Y_bj = S_ib^-1 * A(ib, bj)  (S_ib^-1 itself at bj == ib); the pivot column blocks are zeroed
#pragma unroll
        for (int k = 0; k < M64B; k++) {
            if (bij[k] >= 0) {
                const int bi = bij[k] >> 8, bj = bij[k] & 255;
                if (bi == ib) {
                    double Pi[9];
#pragma unroll
                    for (int q = 0; q < 9; q++) Pi[q] = sPinv[q];
                    if (bj == ib) {
#pragma unroll
                        for (int q = 0; q < 9; q++) A[k][q] = Pi[q];
                    } else {
                        double Y[9];
                        mul3(Pi, A[k], Y);
#pragma unroll
                        for (int q = 0; q < 9; q++) A[k][q] = Y[q];
                    }
#pragma unroll
                    for (int q = 0; q < 9; q++) sY[bj][q] = A[k][q];
                    if (bj == ib + 1) { for (int q = 0; q < 9; q++) sCol[cb ^ 1][bi][q] = A[k][q]; }
                } else if (bj == ib) {
#pragma unroll
                    for (int q = 0; q < 9; q++) A[k][q] = 0.0;
                }
            }
        }
        __syncthreads();
        // phase 2: every other block: A(bi, bj) -= F * Y_bj, three fused multiply-adds per element
#pragma unroll
        for (int k = 0; k < M64B; k++) {
            if (bij[k] >= 0) {
                const int bi = bij[k] >> 8, bj = bij[k] & 255;
                if (bi != ib) {
                    double F[9], Y[9];
#pragma unroll
                    for (int q = 0; q < 9; q++) { F[q] = sCol[cb][bi][q]; Y[q] = sY[bj][q]; }
#pragma unroll
                    for (int i = 0; i < 3; i++)
#pragma unroll
                        for (int j = 0; j < 3; j++)
                            A[k][i * 3 + j] = fma(-F[i * 3 + 2], Y[6 + j], fma(-F[i * 3 + 1], Y[3 + j], fma(-F[i * 3], Y[j], A[k][i * 3 + j])));
                    if (bj == ib && bi > ib) {
                        const double z0 = sZe[3 * ib], z1 = sZe[3 * ib + 1], z2 = sZe[3 * ib + 2];
#pragma unroll
                        for (int a = 0; a < 3; a++) sNu[3 * bi + a] -= A[k][a * 3] * z0 + A[k][a * 3 + 1] * z1 + A[k][a * 3 + 2] * z2;   // nu += (H K) ze, H K = -A
                    }
                    if (bj == ib + 1) { for (int q = 0; q < 9; q++) sCol[cb ^ 1][bi][q] = A[k][q]; }
                    if (bi == ib + 1 && bj == ib + 1) {
                        double Pn[9];
                        inv3_reg(A[k], Pn);
#pragma unroll
                        for (int q = 0; q < 9; q++) sPinv[q] = Pn[q];
                    }
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < M64B; k++) {
        if (bij[k] >= 0) {
            const int bi = bij[k] >> 8, bj = bij[k] & 255;
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int c = 0; c < 3; c++) E.d_G[(size_t)(3 * bi + a) * n3 + 3 * bj + c] = A[k][a * 3 + c];
            const double n0 = sNu[3 * bj], n1 = sNu[3 * bj + 1], n2 = sNu[3 * bj + 2];
#pragma unroll
            for (int a = 0; a < 3; a++) sPart[bi][bj][a] = A[k][a * 3] * n0 + A[k][a * 3 + 1] * n1 + A[k][a * 3 + 2] * n2;
        }
    }
    __syncthreads();
    for (int r = tid; r < n3; r += M64T) {
        const int i = r / 3, a = r - 3 * i;
        double acc = 0;
        for (int j = 0; j < m; j++) acc += sPart[i][j][a];
        E.d_g[r] = acc;
    }
}

// Sigma <- Sigma - W T on the f64 matrix cores.  One wavefront owns a 64 x 64 tile of Sigma as 4 x 4 tiles of
// v_mfma_f64_16x16x4_f64; the tile is formed transposed, D'[c][r] = sum_p T[p][c] W^T[p][r], so that every accumulator
// register maps to 16 consecutive rows r of one column c: the read-modify-write of the column-major Sigma is coalesced.
// Operand layout (cdna_hip_programming.md §3): A[i][k] in lane k*16+i, B[k][j] in lane k*16+j, D rows (lane>>4)+4*reg, col lane&15.

constexpr int MUK = 8;                   // depth rows of T / W^T staged per chunk
constexpr int MUR = 64;                  // rows of Sigma per workgroup

// Workgroup = 2 wavefronts, tile = 64 rows x (2 x 16 WCT) columns, WCT = 4 or 5 MFMA column tiles per wavefront.  The host picks
// the width that lets ALL workgroups be resident at once (4 per CU: 2 wavefronts per SIMD at this register footprint): at
// N = 3003, 128-wide tiles make 1128 workgroups = a second, nearly empty round, 160-wide ones 893 <= 1024.  The Sigma tile is
// loaded straight into the accumulators before the depth loop (its latency hides behind the first chunks) and the product
// is subtracted by negating one operand: the epilogue is a pure store.
template <int WCT>
__global__ __launch_bounds__(128) void k_ekf_update_mfma(EkfState E, int depth) {
    constexpr int MUC = 2 * 16 * WCT;                              // columns per workgroup
    __shared__ double sT[2][MUK][MUC];
    __shared__ double sW[2][MUK][MUR];
    const int n3 = depth >= 0 ? depth : 3 * (*E.d_m);             // rows of d_T / d_Wt contracted (the window flush passes its own)
    if (n3 <= 0) return;
    const int N = 3 + 3 * (*E.d_L);
    const int ld = E.ld;
    const int cb0 = blockIdx.y * MUC, rb0 = blockIdx.x * MUR;     // workgroup tile: columns cb0.., rows rb0.. of Sigma
    if (cb0 >= N || rb0 >= N) return;                              // uniform per workgroup
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int wc = wave * 16 * WCT;                                // this wavefront's columns (all 64 rows)
    const int li = lane & 15, lk = lane >> 4;

    // chunk loader: MUK x (MUC + 64) doubles, issued in bulk: T rows by (tid, tid + 128), W^T rows split by wavefront
    constexpr int NTL = (MUC + 127) / 128;                         // T columns per thread (1 or 2)
    const int wx = tid & 63, wp = tid >> 6;
    double pt[MUK][NTL], pw[MUK / 2];
    const int nchunks = (n3 + MUK - 1) / MUK;
#pragma unroll
    for (int k = 0; k < MUK; k++)
#pragma unroll
        for (int q = 0; q < NTL; q++) {
            const int x = tid + 128 * q;
            pt[k][q] = (x < MUC && k < n3 && cb0 + x < N) ? E.d_T[(size_t)k * ld + cb0 + x] : 0.0;
        }
#pragma unroll
    for (int k = 0; k < MUK / 2; k++) { const int p = wp + 2 * k; pw[k] = (p < n3 && rb0 + wx < N) ? E.d_Wt[(size_t)p * ld + rb0 + wx] : 0.0; }

    v4d acc[WCT][4];                                               // D'[c][r]: register reg of acc[ci][ri] = Sigma(r = 16 ri + li, c = 16 ci + lk + 4 reg)
#pragma unroll
    for (int ci = 0; ci < WCT; ci++)
#pragma unroll
        for (int ri = 0; ri < 4; ri++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int c = cb0 + wc + 16 * ci + lk + 4 * reg, r = rb0 + 16 * ri + li;
                acc[ci][ri][reg] = (c < N && r < N) ? E.d_sigma[(size_t)c * ld + r] : 0.0;
            }

    for (int ch = 0; ch < nchunks; ch++) {
        const int buf = ch & 1;
#pragma unroll
        for (int k = 0; k < MUK; k++)
#pragma unroll
            for (int q = 0; q < NTL; q++) { const int x = tid + 128 * q; if (x < MUC) sT[buf][k][x] = pt[k][q]; }
#pragma unroll
        for (int k = 0; k < MUK / 2; k++) sW[buf][wp + 2 * k][wx] = pw[k];
        __syncthreads();
        if (ch + 1 < nchunks) {                                    // prefetch the next chunk while this one is multiplied
            const int pb = (ch + 1) * MUK;
#pragma unroll
            for (int k = 0; k < MUK; k++)
#pragma unroll
                for (int q = 0; q < NTL; q++) {
                    const int x = tid + 128 * q;
                    pt[k][q] = (x < MUC && pb + k < n3 && cb0 + x < N) ? E.d_T[(size_t)(pb + k) * ld + cb0 + x] : 0.0;
                }
#pragma unroll
            for (int k = 0; k < MUK / 2; k++) { const int p = pb + wp + 2 * k; pw[k] = (p < n3 && rb0 + wx < N) ? E.d_Wt[(size_t)p * ld + rb0 + wx] : 0.0; }
        }
#pragma unroll
        for (int kk = 0; kk < MUK / 4; kk++) {
            double a[WCT], b[4];
#pragma unroll
            for (int q = 0; q < WCT; q++) a[q] = -sT[buf][kk * 4 + lk][wc + 16 * q + li];
#pragma unroll
            for (int q = 0; q < 4; q++) b[q] = sW[buf][kk * 4 + lk][16 * q + li];
#pragma unroll
            for (int ci = 0; ci < WCT; ci++)
#pragma unroll
                for (int ri = 0; ri < 4; ri++) acc[ci][ri] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ci], b[ri], acc[ci][ri], 0, 0, 0);
        }
        // the other buffer is rewritten next iteration: every wave has finished reading it one iteration ago (barrier above)
    }
#pragma unroll
    for (int ci = 0; ci < WCT; ci++)
#pragma unroll
        for (int ri = 0; ri < 4; ri++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int c = cb0 + wc + 16 * ci + lk + 4 * reg, r = rb0 + 16 * ri + li;
                if (c < N && r < N) E.d_sigma[(size_t)c * ld + r] = acc[ci][ri][reg];
            }
}

__global__ __launch_bounds__(256) void k_ekf_export_map(EkfState E) {
    const int L = *E.d_L;
    const int ld = E.ld;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < E.max_landmarks; i += gridDim.x * 256) {
        MapRecord r;
        if (i < L) {
            const int li = 3 + 3 * i;
            r.id = E.d_idx2id[i]; r.index = i;
            r.x = E.d_mu[li]; r.y = E.d_mu[li + 1]; r.theta = E.d_mu[li + 2];
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) r.S[a * 3 + b] = E.d_sigma[(size_t)(li + b) * ld + li + a];
        } else {
            r.id = -1; r.index = -1; r.x = r.y = r.theta = 0;
            for (int k = 0; k < 9; k++) r.S[k] = 0;
        }
        E.d_maprec[i] = r;
    }
}

// ---- host side -------------------------------------------------------------------------------------------
template <class T> static hipError_t dalloc(T** p, size_t count) { return hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)); }

hipError_t ekf_alloc(EkfState& E, int max_landmarks, int max_slots, int max_updates_per_frame) {
    E = EkfState{};
    E.max_landmarks = max_landmarks;
    E.max_slots = max_slots;
    E.ld = 3 + 3 * max_landmarks;
    const size_t ld = (size_t)E.ld, n3 = 3 * (size_t)kMarkerMax;
    hipError_t e;
#define A(x) if ((e = (x)) != hipSuccess) return e
    A(dalloc(&E.d_mu, ld));
    A(dalloc(&E.d_sigma, ld * ld));
    A(dalloc(&E.d_L, 1));
    A(dalloc(&E.d_id2idx, kIdTableSize));
    A(dalloc(&E.d_idx2id, (size_t)max_landmarks));
    A(dalloc(&E.d_last, kMarkerMax));
    A(dalloc(&E.d_lastNext, kMarkerMax));
    A(dalloc(&E.d_nlast, 1));
    A(dalloc(&E.d_pop, kMarkerMax));
    A(dalloc(&E.d_npop, 1));
    A(dalloc(&E.d_upd, kMarkerMax));
    A(dalloc(&E.d_m, 1));
    A(dalloc(&E.d_V, n3 * ld));
    A(dalloc(&E.d_Wt, n3 * ld));
    A(dalloc(&E.d_T, n3 * ld));
    A(dalloc(&E.d_Sv, n3 * n3));
    A(dalloc(&E.d_Sw, n3 * n3));
    A(dalloc(&E.d_alpha, n3 * n3));
    A(dalloc(&E.d_gamma, n3 * n3));
    A(dalloc(&E.d_G, n3 * n3));
    A(dalloc(&E.d_g, n3));
    A(dalloc(&E.d_maprec, (size_t)max_landmarks));
    // windowed EKF: contexts configured for <= 24 corrections per frame keep sets of <= 41 landmarks (SP <= 128)
    E.win_sp_max = max_updates_per_frame <= 24 ? 128 : 192;
    E.win_steps_max = kWinFrames * (1 + std::min(max_updates_per_frame, kWinCorrMax));
    A(dalloc(&E.d_win_log, (size_t)E.win_steps_max * (3 * E.win_sp_max + kWinHdr) + 512));
    A(dalloc(&E.d_win_tlog, (size_t)E.win_steps_max * 8 * E.win_sp_max));
    A(dalloc(&E.d_win_small, (size_t)6 * E.win_sp_max * E.win_sp_max + 4 * E.win_sp_max));
    A(dalloc(&E.d_win_next, (size_t)4 * E.win_sp_max * E.win_sp_max));
    A(dalloc(&E.d_win_next_idx, (size_t)E.win_sp_max + 4));
    A(dalloc(&E.d_win_sidx, ld));
    A(dalloc(&E.d_win_frames, (size_t)max_slots));
    A(dalloc(&E.d_slot_stat, (size_t)4 * max_slots));
    A(hipMemset(E.d_slot_stat, 0, (size_t)4 * max_slots * sizeof(int)));
    // ArucoSlam::ArucoSlam (aruco_slam.cpp:13-18): mu = 0 (3), sigma = 0 (3x3), empty map
    A(hipMemset(E.d_mu, 0, ld * sizeof(double)));
    A(hipMemset(E.d_sigma, 0, ld * ld * sizeof(double)));
    A(hipMemset(E.d_L, 0, sizeof(int)));
    A(hipMemset(E.d_id2idx, 0xFF, kIdTableSize * sizeof(int)));
    A(hipMemset(E.d_idx2id, 0xFF, (size_t)max_landmarks * sizeof(int)));
    A(hipMemset(E.d_nlast, 0, sizeof(int)));
    A(hipMemset(E.d_npop, 0, sizeof(int)));
    A(hipMemset(E.d_m, 0, sizeof(int)));
#undef A
    return hipSuccess;
}

void ekf_free(EkfState& E) {
    hipFree(E.d_mu); hipFree(E.d_sigma); hipFree(E.d_L); hipFree(E.d_id2idx); hipFree(E.d_idx2id); hipFree(E.d_last); hipFree(E.d_lastNext);
    hipFree(E.d_nlast); hipFree(E.d_pop); hipFree(E.d_npop); hipFree(E.d_upd); hipFree(E.d_m); hipFree(E.d_V); hipFree(E.d_Wt);
    hipFree(E.d_T); hipFree(E.d_Sv); hipFree(E.d_Sw); hipFree(E.d_alpha); hipFree(E.d_gamma); hipFree(E.d_G); hipFree(E.d_g);
    hipFree(E.d_maprec); hipFree(E.d_slot_stat); hipFree(E.d_win_log); hipFree(E.d_win_tlog); hipFree(E.d_win_small); hipFree(E.d_win_next); hipFree(E.d_win_next_idx); hipFree(E.d_win_sidx); hipFree(E.d_win_frames);
    E = EkfState{};
}

void launch_ekf_predict_only(hipStream_t st, const EkfState& E, const SlamParams& sp, double wl, double wr, double dt) {
    hipLaunchKernelGGL(k_ekf_predict, dim3(1), dim3(256), 0, st, E, sp, wl, wr, dt);
}
void launch_ekf_plan(hipStream_t st, const EkfState& E, const SlamParams& sp, double wl, double wr, double dt, int do_predict,
                     const ObsRaw* obs, const unsigned* n_markers, Counters* ctr, int max_m, int slot) {
    hipLaunchKernelGGL(k_ekf_plan, dim3(1), dim3(256), 0, st, E, sp, wl, wr, dt, do_predict, obs, n_markers, ctr, max_m, slot);
}
void launch_ekf_gather(hipStream_t st, const EkfState& E) {
    hipLaunchKernelGGL(k_ekf_gather, dim3((E.ld + 255) / 256, 32), dim3(256), 0, st, E);
}
void launch_ekf_small(hipStream_t st, const EkfState& E) {
    hipLaunchKernelGGL(k_ekf_small, dim3(1), dim3(SMT), 0, st, E);
}
void launch_ekf_T(hipStream_t st, const EkfState& E) {
    hipLaunchKernelGGL(k_ekf_T, dim3((E.ld + TC - 1) / TC, (3 * kMarkerMax + TR - 1) / TR), dim3(256), 0, st, E);
}
void launch_ekf_mid(hipStream_t st, const EkfState& E) {
    const int ncg = (E.ld + MIDT - 1) / MIDT;
    hipLaunchKernelGGL(k_ekf_mid, dim3(1 + ncg * 12), dim3(MIDT), 0, st, E);
}
void launch_ekf_apply(hipStream_t st, const EkfState& E) {
    const int t = (E.ld + 63) / 64;
    hipLaunchKernelGGL(k_ekf_apply, dim3(t, t), dim3(256), 0, st, E);
}
int ekf_fast_max_updates() { return kFastM; }
int ekf_mid_max_updates() { return kMidM; }
void launch_ekf_mid64(hipStream_t st, const EkfState& E) {
    const int ncg = (E.ld + M64T - 1) / M64T;
    hipLaunchKernelGGL(k_ekf_mid64, dim3(1 + ncg * 16), dim3(M64T), 0, st, E);
}
void launch_ekf_update_mfma(hipStream_t st, const EkfState& E, int depth) {
    // the narrowest tile whose workgroups all fit on the device at once (256 CUs x 4); N_max stands in for the current N
    const int rt = (E.ld + MUR - 1) / MUR;
    if (rt * ((E.ld + 127) / 128) <= 1024 || rt * ((E.ld + 159) / 160) > 1024)
        hipLaunchKernelGGL(k_ekf_update_mfma<4>, dim3(rt, (E.ld + 127) / 128), dim3(128), 0, st, E, depth);
    else
        hipLaunchKernelGGL(k_ekf_update_mfma<5>, dim3(rt, (E.ld + 159) / 160), dim3(128), 0, st, E, depth);
}
void launch_ekf_export_map(hipStream_t st, const EkfState& E) {
    hipLaunchKernelGGL(k_ekf_export_map, dim3((E.max_landmarks + 255) / 256), dim3(256), 0, st, E);
}

} // namespace aslam
