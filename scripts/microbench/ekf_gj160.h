// Block Gauss-Jordan inverse of a symmetric positive definite 3m x 3m matrix (3m <= 160) on the f64 matrix cores of ONE
// workgroup: the medium EKF chain's innovation matrix A = H Sigma0 H^T + R (aruco_slam.cpp:146 fused over the frame's m
// corrections, m = 25 .. 53), with the pseudo-innovation nu (quirk Q1) collected on the way and g = G nu at the end.
// Same scheme as the window chain's sweep (ekf_window.hip), sized up:
//   * step j (pivot rows / columns p = 3j .. 3j + 2, S = A[p,p] = the reference's S_j) is ONE rank-3 product A <- A - C~ Y~,
//     C~ = C with S - I in the pivot rows, Y~ = S^-1 R with the pivot columns replaced by I + S^-1;
//   * the partially inverted image stays symmetric up to the sign of the pivoted / unpivoted cross blocks (A is symmetric to
//     rounding), so only the tiles ON OR ABOVE THE DIAGONAL are kept: 55 of the 10 x 10 MFMA tiles of 16 x 16, in the
//     accumulators of 4 worker waves for the whole sweep (the image does not fit LDS - 205 KB - and does not have to); tile
//     u of the upper triangle (row-major) belongs to wave u % 4, which puts 14 tiles on every SIMD's matrix pipe;
//   * only the three pivot ROWS ever leave the accumulators (their part left of the diagonal comes out of the mirrored tiles
//     as columns, sign by pivoted / unpivoted), and the column operand C~ is read from the same rows;
//   * a fifth wave prepares pivot j + 1 while the workers apply step j: the owners publish the rows of pivot j + 2 as they
//     stand after step j; one phase later the prepare wave applies step j + 1's correction to them from the rows and Y~ it
//     still holds in registers (lane = column, three columns per lane; uniform values by v_readlane), inverts S and hands
//     Y~ and the corrected rows over.  One barrier per step.  320 threads = 2 waves on one SIMD, 1 on the others: 256 VGPRs.
#pragma once
#include "../../aruco_slam_amd/csrc/ekf_dev.h"

namespace aslam {

constexpr int GJ160_THREADS = 320;                       // 4 worker waves + 1 prepare wave
constexpr int GJ160_N = 160;                             // largest 3m
constexpr int GJ160_TPW = 14;                            // upper-triangle tiles per worker wave (55 / 4, rounded up)
constexpr int GJ160_GW = 168;                            // row stride of Y~ and of the pivot rows in LDS: the four depth rows a wave reads land in different banks
constexpr int GJ160_SW = 164;                            // row stride of the published tile rows (the transposed writes spread over the banks)
constexpr int GJ160_OFF_GY = 0;                                      // Y~        [2][4][GW]   (4th depth row zero)
constexpr int GJ160_OFF_ROW = GJ160_OFF_GY + 2 * 4 * GJ160_GW;      // rows      [2][4][GW]   (4th row zero)
constexpr int GJ160_OFF_PUB = GJ160_OFF_ROW + 2 * 4 * GJ160_GW;     // published tile rows [2][32][SW]
constexpr int GJ160_OFF_NU = GJ160_OFF_PUB + 2 * 32 * GJ160_SW;     // nu        [160]
constexpr int GJ160_OFF_PART = GJ160_OFF_NU + GJ160_N;              // partial sums of g [160][10]
constexpr int GJ160_LDS_DOUBLES = GJ160_OFF_PART + GJ160_N * 10;

struct Gj160Tiles { int tr[GJ160_TPW], tc[GJ160_TPW]; };             // this wave's tiles (wave-uniform), tr = -1: none

// Rows p .. p + 2 of the image leave the accumulators: the (at most two) tile rows they lie in are written out whole - 16 rows each,
// register numbers static, so no select chain over the accumulators and no per-pivot code (a switch over p mod 16 with the
// tile loop unrolled inside thrashes the instruction cache: 16 000 cycles per step) - and the prepare wave picks its three rows
// by address.  Columns right of (and in) the diagonal tile come from the tiles of tile row X; columns left of it from the tiles
// of tile COLUMN X, transposed, sign flipped where the column index is already pivoted (< piv).
__device__ __forceinline__ void gj160_publish(const v4d (&ga)[GJ160_TPW], const Gj160Tiles& tl, int p, int piv, int lk, int li, double* stage) {
    const int trA = p >> 4, trB = (p + 2) >> 4;
#pragma unroll
    for (int k = 0; k < GJ160_TPW; k++) {
        const int tr = tl.tr[k], tc = tl.tc[k];
        if (tr < 0) continue;
        if (tr == trA || tr == trB) {
            double* dst = stage + ((tr == trA ? 0 : 16) + lk) * GJ160_SW + 16 * tc + li;
#pragma unroll
            for (int reg = 0; reg < 4; reg++) dst[4 * reg * GJ160_SW] = ga[k][reg];
        }
        if (tr != tc && (tc == trA || tc == trB)) {
            double* dst = stage + ((tc == trA ? 0 : 16) + li) * GJ160_SW + 16 * tr + lk;
#pragma unroll
            for (int reg = 0; reg < 4; reg++) dst[4 * reg] = (16 * tr + lk + 4 * reg) < piv ? -ga[k][reg] : ga[k][reg];
        }
    }
}

// value of column / row `idx` (wave-uniform) of a quantity the prepare wave holds as three registers per lane (idx = lane + 64 slot)
__device__ __forceinline__ double gj160_pick(const double (&a)[3], int idx) {
    const int slot = idx >> 6;
    const double v = slot == 0 ? a[0] : slot == 1 ? a[1] : a[2];
    return ASLAM_WAVE_BCAST(v, idx & 63);
}

// G: in  the matrix A, row-major with stride n3 (global memory; only the elements on or above the diagonal are read; written
//        by this workgroup before the call and made visible by a barrier), out its inverse (both triangles);
// ze: the 3m innovations at the frozen mean; gout: g = G nu.  All GJ160_THREADS threads of the workgroup call.
__device__ __forceinline__ void gj160_sweep(double* __restrict__ G, int n3, int m, const double* __restrict__ ze_in, double* __restrict__ gout,
                                            double* smem) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int T = (n3 + 15) >> 4;                                   // tiles per side (<= 10)
    double* sGY = smem + GJ160_OFF_GY;
    double* sRow = smem + GJ160_OFF_ROW;
    double* sPub = smem + GJ160_OFF_PUB;
    double* sNu = smem + GJ160_OFF_NU;
    double* sPart = smem + GJ160_OFF_PART;
    for (int e = tid; e < GJ160_GW; e += GJ160_THREADS) { sGY[3 * GJ160_GW + e] = 0.0; sGY[7 * GJ160_GW + e] = 0.0; }
    for (int e = tid; e < GJ160_GW; e += GJ160_THREADS) { sRow[3 * GJ160_GW + e] = 0.0; sRow[7 * GJ160_GW + e] = 0.0; }
    for (int e = tid; e < 2 * 32 * GJ160_SW; e += GJ160_THREADS) sPub[e] = 0.0;

    // this wave's tiles: tile u of the upper triangle of the T x T tile grid (row-major) belongs to wave u % 4, slot u / 4
    Gj160Tiles tl;
    {
        const int uw = __builtin_amdgcn_readfirstlane(wave);       // scalar: the tile coordinates below stay in SGPRs
#pragma unroll
        for (int k = 0; k < GJ160_TPW; k++) {
            const int u = uw + 4 * k;
            int tr = -1, tc = -1, acc = 0;
#pragma unroll
            for (int t = 0; t < 10; t++) {
                const int len = T - t;
                if (len > 0 && tr < 0 && uw < 4) {
                    if (u < acc + len) { tr = t; tc = t + (u - acc); }
                    acc += len;
                }
            }
            tl.tr[k] = tr; tl.tc[k] = tc;
        }
    }
    v4d ga[GJ160_TPW];
    int rowk[GJ160_TPW], colk[GJ160_TPW];                          // this lane's operand row / column of every tile (unused slots: tile (0, 0))
#pragma unroll
    for (int k = 0; k < GJ160_TPW; k++) { rowk[k] = 16 * max(tl.tr[k], 0) + li; colk[k] = 16 * max(tl.tc[k], 0) + li; }
    double nu[3] = {0.0, 0.0, 0.0}, ze[3] = {0.0, 0.0, 0.0};
    if (wave < 4) {
#pragma unroll
        for (int k = 0; k < GJ160_TPW; k++) {
            const int c = 16 * tl.tc[k] + li;
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int r = 16 * tl.tr[k] + lk + 4 * reg;
                double v = 0.0;
                if (tl.tr[k] >= 0 && r < n3 && c < n3) v = r <= c ? G[(size_t)r * n3 + c] : G[(size_t)c * n3 + r];   // diagonal tiles: mirrored fill
                ga[k][reg] = v;
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < 3; s++) { const int r = lane + 64 * s; ze[s] = r < n3 ? ze_in[r] : 0.0; nu[s] = ze[s]; }
    }
    __syncthreads();                                                // the zeroed LDS rows; every tile has been read before G is rewritten
    if (wave < 4) {
        gj160_publish(ga, tl, 0, 0, lk, li, sPub);
        if (m > 1) gj160_publish(ga, tl, 3, 0, lk, li, sPub + 32 * GJ160_SW);
    }
    ASLAM_LDS_BARRIER();
    for (int j = -1; j < m; j++) {
        // phase j: the workers apply step j and publish the rows of pivot j + 2; the prepare wave forms the operands of pivot j + 1
        if (wave == 4) {
            if (j + 1 < m) {
                const int jb = (j + 1) & 1, p = 3 * (j + 1);
                double R[3][3];
#pragma unroll
                for (int q = 0; q < 3; q++)
#pragma unroll
                    for (int s = 0; s < 3; s++) {
                        const int srow = (((p + q) >> 4) == (p >> 4) ? 0 : 16) + ((p + q) & 15);
                        R[q][s] = (s < 2 || lane < GJ160_N - 128) ? sPub[(jb * 32 + srow) * GJ160_SW + lane + 64 * s] : 0.0;
                    }
                if (j >= 0) {
                    // step j's correction of these rows: C~_j[p + q][k] = R_j[k][p + q] (rows behind pivot j); R_j and Y~_j are what this
                    // wave handed over one phase ago (read back from LDS: keeping them in registers would cost every wave 36 VGPRs)
                    double Rp[3][3], Yp[3][3];
#pragma unroll
                    for (int k = 0; k < 3; k++)
#pragma unroll
                        for (int s = 0; s < 3; s++) {
                            const bool in = s < 2 || lane < GJ160_N - 128;
                            Rp[k][s] = in ? sRow[((jb ^ 1) * 4 + k) * GJ160_GW + lane + 64 * s] : 0.0;
                            Yp[k][s] = in ? sGY[((jb ^ 1) * 4 + k) * GJ160_GW + lane + 64 * s] : 0.0;
                        }
#pragma unroll
                    for (int q = 0; q < 3; q++) {
                        const double c0 = gj160_pick(Rp[0], p + q), c1 = gj160_pick(Rp[1], p + q), c2 = gj160_pick(Rp[2], p + q);
#pragma unroll
                        for (int s = 0; s < 3; s++) R[q][s] = fma(-c2, Yp[2][s], fma(-c1, Yp[1][s], fma(-c0, Yp[0][s], R[q][s])));
                    }
                }
                double Sm[9], Si[9];
#pragma unroll
                for (int q = 0; q < 3; q++)
#pragma unroll
                    for (int c = 0; c < 3; c++) Sm[q * 3 + c] = gj160_pick(R[q], p + c);
                inv3_fast(Sm, Si);
#pragma unroll
                for (int s = 0; s < 3; s++) {
                    const int c = lane + 64 * s;
                    const double r0 = R[0][s] + (c == p ? 1.0 : 0.0), r1 = R[1][s] + (c == p + 1 ? 1.0 : 0.0), r2 = R[2][s] + (c == p + 2 ? 1.0 : 0.0);   // R~
                    double Yn[3];
                    Yn[0] = fma(Si[2], r2, fma(Si[1], r1, Si[0] * r0));
                    Yn[1] = fma(Si[5], r2, fma(Si[4], r1, Si[3] * r0));
                    Yn[2] = fma(Si[8], r2, fma(Si[7], r1, Si[6] * r0));
                    if (c < GJ160_N) {
#pragma unroll
                        for (int k = 0; k < 3; k++) { sGY[(jb * 4 + k) * GJ160_GW + c] = Yn[k]; sRow[(jb * 4 + k) * GJ160_GW + c] = R[k][s]; }
                    }
                }
                // nu_r += (C_r S^-1) ze_p for the rows behind the pivot: C[r][k] = R[k][r] there, u = S^-1 ze_p
                const double z0 = gj160_pick(ze, p), z1 = gj160_pick(ze, p + 1), z2 = gj160_pick(ze, p + 2);
                const double u0 = fma(Si[2], z2, fma(Si[1], z1, Si[0] * z0)), u1 = fma(Si[5], z2, fma(Si[4], z1, Si[3] * z0)), u2 = fma(Si[8], z2, fma(Si[7], z1, Si[6] * z0));
#pragma unroll
                for (int s = 0; s < 3; s++) {
                    const int r = lane + 64 * s;
                    if (r >= p + 3 && r < n3) nu[s] += R[0][s] * u0 + R[1][s] * u1 + R[2][s] * u2;
                }
            }
        } else if (j >= 0) {
            // ---- apply step j on this wave's tiles.  A operand -C~[row][k = lk] from the pivot rows: rows already pivoted carry the
            // opposite sign, the pivot rows themselves S - I; depth 3 is the zero row ----
            const int cb = j & 1, p0 = 3 * j;
            // (no branch per tile: an unused slot multiplies tile (0, 0)'s operands into an accumulator nobody reads, and the
            // operand loads of all tiles are in flight before the first product is issued)
            double af[GJ160_TPW], bf[GJ160_TPW];
            const double* rowbase = sRow + (cb * 4 + lk) * GJ160_GW;
            const double* ybase = sGY + (cb * 4 + lk) * GJ160_GW;
#pragma unroll
            for (int k = 0; k < GJ160_TPW; k++) {
                const double rr = rowbase[rowk[k]];
                af[k] = rowk[k] < p0 ? rr : -rr;
                if (lk < 3 && rowk[k] == p0 + lk) af[k] += 1.0;
                bf[k] = ybase[colk[k]];
            }
#pragma unroll
            for (int k = 0; k < GJ160_TPW; k++) ga[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[k], bf[k], ga[k], 0, 0, 0);
            if (j + 2 < m) gj160_publish(ga, tl, p0 + 6, p0 + 3, lk, li, sPub + (cb * 32) * GJ160_SW);
        }
        ASLAM_LDS_BARRIER();
    }
    // G = A^-1 back to memory (both triangles); g = G nu from per-tile partial sums in a fixed order
    if (wave == 4) {
#pragma unroll
        for (int s = 0; s < 3; s++) { const int r = lane + 64 * s; if (r < GJ160_N) sNu[r] = nu[s]; }
    }
    ASLAM_LDS_BARRIER();
    if (wave < 4) {
#pragma unroll
        for (int k = 0; k < GJ160_TPW; k++) {
            if (tl.tr[k] >= 0) {                                    // wave-uniform
                const int tr = tl.tr[k], tc = tl.tc[k], c = 16 * tc + li;
                const double nvc = sNu[c];
                double colsum = 0.0;
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int r = 16 * tr + lk + 4 * reg;
                    const double gv = ga[k][reg];
                    if (r < n3 && c < n3) {
                        G[(size_t)r * n3 + c] = gv;
                        if (tr != tc) G[(size_t)c * n3 + r] = gv;
                    }
                    double part = gv * nvc;                         // row r of G times nu, this tile's 16 columns
                    part += __shfl_xor(part, 1); part += __shfl_xor(part, 2); part += __shfl_xor(part, 4); part += __shfl_xor(part, 8);
                    if (li == 0) sPart[r * 10 + tc] = part;
                    colsum += gv * sNu[r];                          // mirrored: row c of G times nu, this tile's 16 rows
                }
                if (tr != tc) {
                    colsum += __shfl_xor(colsum, 16); colsum += __shfl_xor(colsum, 32);
                    if (lk == 0) sPart[c * 10 + tr] = colsum;
                }
            }
        }
    }
    ASLAM_LDS_BARRIER();
    for (int r = tid; r < n3; r += GJ160_THREADS) {
        double acc = 0.0;
        for (int t = 0; t < T; t++) acc += sPart[r * 10 + t];
        gout[r] = acc;
    }
}

}  // namespace aslam
