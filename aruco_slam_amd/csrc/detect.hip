// Marker detection on gfx950: the device-side replacement for cv::aruco::detectMarkers as the reference
// calls it (src/aruco_slam.cpp:313, default DetectorParameters).  Stages and what they replace:
//
//   k_threshold  BGR->gray + 3x adaptiveThreshold(MEAN_C, BINARY_INV, win 3/13/23, C=7) in ONE pass over the
//                frame: LDS tile with a 12-px halo, LDS integral image, exact integer box means; emits, per
//                scale, an 8-neighbour occupancy byte per pixel and the list of border nodes (start candidates + cut states).
//   k_seg, k_link, k_trace_write
//                Suzuki-Abe border following (findContours RETR_LIST/CHAIN_APPROX_NONE) without a sequential raster scan and
//                without long walks: every border is a cycle of (pixel, back-direction) states; k_threshold lists the nodes that
//                cut the cycles into short segments (common.h), k_seg walks each segment (one lane per node), k_link resolves the
//                node cycles of a frame by pointer jumping in LDS - the canonical start (the state the sequential scan would
//                have started from) emits the contour - and k_trace_write replays the segments of the kept contours, 64 points
//                per lane.
//   k_quads      approxPolyDP + the quad tests of _findMarkerContours, one wavefront per contour.
//   k_assemble   candidate ordering (scale, reverse discovery), _reorderCandidatesCorners,
//                _filterTooCloseCandidates; one workgroup per frame.
//   k_identify   _extractBits (perspective warp, Otsu) + border check + Dictionary::identify; one wavefront
//                per candidate.
//
// All integer / index results are bit-identical to the CPU oracle (oracle/detect.cpp); the fp64 sequences that
// feed discrete decisions are written in the same operation order and the library is built with
// -ffp-contract=off.
#include "common.h"
#include "detect.h"
#include <cfloat>
#include <climits>
#include <cstdlib>
#include <algorithm>

namespace aslam {

// direction d = 0..7: E NE N NW W SW S SE (counter-clockwise on the screen)
__device__ __forceinline__ int dir_dx(int s) { return (int)((0x901Au >> (2 * s)) & 3u) - 1; }   // {1,1,0,-1,-1,-1,0,1}+1 packed
__device__ __forceinline__ int dir_dy(int s) { return (int)((0xA901u >> (2 * s)) & 3u) - 1; }   // {0,-1,-1,-1,0,1,1,1}+1 packed
// first foreground neighbour clockwise from W for a pixel whose NE,N,NW,W are background: E, SE, S, SW
__device__ __forceinline__ int first_outer(unsigned m) { return (m & 1u) ? 0 : (m & 128u) ? 7 : (m & 64u) ? 6 : 5; }
// first foreground neighbour clockwise from E: SE, S, SW, W, NW, N, NE = highest set bit among bits 7..1
__device__ __forceinline__ int first_hole(unsigned m) { return 31 - __clz((int)(m & 0xFEu)); }


// top bit of every byte of x that is zero (exact: no borrow between bytes)
__device__ __forceinline__ unsigned byte_is_zero(unsigned x) { return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u; }

// ------------------------------------------------------------------------------------------------
// k_threshold
// ------------------------------------------------------------------------------------------------
constexpr int TW = 64, TH = 32, HALO = 12;
constexpr int LW = TW + 2 * HALO;   // 88
constexpr int LH = TH + 2 * HALO;   // 56
constexpr int kBlockStarts = 1024;  // start candidates staged per tile before one reservation in the frame's list

// adaptiveThreshold(MEAN_C, BINARY_INV): foreground iff  v - mean <= -C  with  mean = round(sum / k^2) = floor((2 sum + k^2) / (2 k^2))
// (k^2 odd: no tie).  For integers  floor(a / b) >= t  <=>  a >= b t  (b > 0), so the decision needs no division:
//     mean >= v + C   <=>   2 sum + k^2 >= 2 k^2 (v + C)   <=>   2 sum >= k^2 (2 (v + C) - 1).
template <int R> __device__ __forceinline__ bool box_is_fg(const unsigned (*I)[LW + 1], int ly, int lx, int v_plus_c) {
    constexpr int k2 = (2 * R + 1) * (2 * R + 1);
    const int sum = (int)(I[ly + R + 1][lx + R + 1] - I[ly - R][lx + R + 1] - I[ly + R + 1][lx - R] + I[ly - R][lx - R]);
    return 2 * sum >= k2 * (2 * v_plus_c - 1);
}
// runtime window radius (DetectorParameters other than the default 3 / 13 / 23): the same exact test
__device__ __forceinline__ bool box_is_fg_r(const unsigned (*I)[LW + 1], int ly, int lx, int R, int v_plus_c) {
    const int k2 = (2 * R + 1) * (2 * R + 1);
    const int sum = (int)(I[ly + R + 1][lx + R + 1] - I[ly - R][lx + R + 1] - I[ly + R + 1][lx - R] + I[ly - R][lx - R]);
    return 2 * sum >= k2 * (2 * v_plus_c - 1);
}

#ifdef ASLAM_THR_STAMPS
#define THR_STAMP(i) do { if (tid == 0 && blockIdx.x == 4000) tst[i] = clock64(); } while (0)
#else
#define THR_STAMP(i) do { } while (0)
#endif
template <bool kDefaultWindows>
__global__ __launch_bounds__(256) void k_threshold(const uint8_t* __restrict__ in, int channels, size_t in_frame_stride,
                                                   size_t in_row_step, uint8_t* __restrict__ gray_out,
                                                   uint8_t* __restrict__ nbr, DetectCfg cfg,
                                                   unsigned* __restrict__ starts, unsigned* __restrict__ n_starts,
                                                   unsigned* __restrict__ nodeplane, Counters* ctr, int nframes) {
    __shared__ uint8_t g[LH][LW + 4];                          // (row pitch 23 words, odd: the row sums below read it with lane = row)
    __shared__ unsigned I[LH + 1][LW + 1];
    __shared__ unsigned long long sRow[kScales][TH + 2];     // threshold decisions of ring row by, columns x0 - 1 .. x0 + 62 (bit = column)
    __shared__ unsigned long long sRing[kScales][2];         // ... of columns x0 + 63 and x0 + 64 (bit = ring row)
    __shared__ unsigned sLutRow[3][64];            // six bits of the row above / at / below four pixels -> its bits of their four neighbour masks
    __shared__ unsigned sStart[kBlockStarts];      // pack_node(x, y, s, scale, type)
    __shared__ unsigned sNStart, sBase;

    const int tid = threadIdx.x;
    // XCD-aware tile order: consecutive workgroup ids are dealt round-robin to the 8 XCDs, each with its own L2.  Give every
    // XCD one contiguous run of tiles (raster order inside a frame, frames in sequence) so that the 12-px halo a tile shares
    // with its neighbours is re-read from the SAME L2 instead of being fetched from HBM once per XCD.
    const int rows = cfg.rows, cols = cfg.cols;
    const unsigned gx = (unsigned)(cols + TW - 1) / TW, gy = (unsigned)(rows + TH - 1) / TH;
    const unsigned per_xcd = gridDim.x >> 3;
    const unsigned logical = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (logical >= gx * gy * (unsigned)nframes) return;            // padding of the grid to a multiple of 8
    const int b = (int)(logical / (gx * gy));
    const unsigned rem = logical - (unsigned)b * gx * gy;
    const int x0 = (int)(rem % gx) * TW, y0 = (int)(rem / gx) * TH;
    const uint8_t* src = in + (size_t)b * in_frame_stride;
#ifdef ASLAM_THR_STAMPS
    long long tst[8] = {0};
#endif
    if (tid == 0) sNStart = 0;
    THR_STAMP(0);

    // 1. gray tile with replicated border (BORDER_REPLICATE of the box filter)
    if (channels == 1 && (in_row_step & 3) == 0 && ((size_t)src & 3) == 0) {
        // 4 pixels per load where the dword lies inside the row; LW = 88 = 22 dwords per tile row.  All of a thread's loads are
        // issued before the first of them is stored: one memory latency per tile, not one per pass
        constexpr int kPasses = (LH * (LW / 4) + 255) / 256;
        unsigned v[kPasses];
#pragma unroll
        for (int p = 0; p < kPasses; p++) {
            const int i = tid + 256 * p;
            v[p] = 0;
            if (i < LH * (LW / 4)) {
                const int ly = i / (LW / 4), q = i - ly * (LW / 4);
                const int gy = min(max(y0 + ly - HALO, 0), rows - 1);
                const int gx = x0 + 4 * q - HALO;
                const uint8_t* rowp = src + (size_t)gy * in_row_step;
                if (gx >= 0 && gx + 3 < cols) {
                    v[p] = *reinterpret_cast<const unsigned*>(rowp + gx);
                } else {
                    for (int k = 0; k < 4; k++) v[p] |= (unsigned)rowp[min(max(gx + k, 0), cols - 1)] << (8 * k);
                }
            }
        }
#pragma unroll
        for (int p = 0; p < kPasses; p++) {
            const int i = tid + 256 * p;
            if (i < LH * (LW / 4)) {
                const int ly = i / (LW / 4), q = i - ly * (LW / 4);
                *reinterpret_cast<unsigned*>(&g[ly][4 * q]) = v[p];
            }
        }
    } else {
        for (int i = tid; i < LH * LW; i += 256) {
            int ly = i / LW, lx = i - ly * LW;
            int gx = min(max(x0 + lx - HALO, 0), cols - 1);
            int gy = min(max(y0 + ly - HALO, 0), rows - 1);
            const uint8_t* p = src + (size_t)gy * in_row_step + (size_t)gx * channels;
            unsigned v;
            if (channels == 3) v = (p[0] * 1868u + p[1] * 9617u + p[2] * 4899u + 8192u) >> 14;
            else v = p[0];
            g[ly][lx] = (uint8_t)v;
        }
    }
    __syncthreads();
    THR_STAMP(1);

    // tight gray plane for the bit-extraction stage (skipped when the input already is one)
    if (gray_out != nullptr) {
        for (int i = tid; i < TH * TW; i += 256) {
            int ty = i / TW, tx = i - ty * TW;
            int gx = x0 + tx, gy = y0 + ty;
            if (gx < cols && gy < rows) gray_out[((size_t)b * rows + gy) * cols + gx] = g[ty + HALO][tx + HALO];
        }
    }

    // 2. integral image I[y+1][x+1] = sum_{y'<=y, x'<=x} g.  Row sums: wave q takes the q-th quarter of the columns (22), lane = row, so
    //    that the 64 lanes of an access stand in 64 different rows: the row pitch of I (89 words, odd) spreads them over all LDS banks
    //    (with a wave spanning 16 rows x 4 quarters, half of the kernel's LDS cycles were bank conflicts).  Column sums: lane = column.
    {
        const int q = tid >> 6, r = tid & 63;
        if (r < LH) {
            unsigned s = 0;
            for (int k = 0; k < LW / 4; k++) { s += g[r][q * (LW / 4) + k]; I[r + 1][q * (LW / 4) + k + 1] = s; }
        }
        if (tid < LW + 1) I[0][tid] = 0;
        if (tid < LH) I[tid + 1][0] = 0;
        __syncthreads();
        unsigned add = 0;
        if (r < LH)
            for (int qq = 0; qq < q; qq++) add += I[r + 1][(qq + 1) * (LW / 4)];     // totals of the preceding quarters (still local sums)
        __syncthreads();
        if (r < LH && q > 0)
            for (int k = 0; k < LW / 4; k++) I[r + 1][q * (LW / 4) + k + 1] += add;
    }
    __syncthreads();
    {
        const int h = tid >> 7, c = tid & 127;                     // two waves per half of the rows: columns 0..63 and 64..87
        if (c < LW) {
            unsigned s = 0;
            for (int k = 0; k < LH / 2; k++) { s += I[h * (LH / 2) + k + 1][c + 1]; I[h * (LH / 2) + k + 1][c + 1] = s; }
        }
        __syncthreads();
        if (c < LW && h == 1) {
            const unsigned add = I[LH / 2][c + 1];
            for (int k = 0; k < LH / 2; k++) I[LH / 2 + k + 1][c + 1] += add;
        }
    }
    __syncthreads();
    THR_STAMP(2);

    // 3. three thresholds (windows 3 / 13 / 23) for the tile plus a 1-px ring (needed by the neighbour masks), kept as BIT ROWS:
    //    a wave thresholds one 64-pixel row per pass and the row's 64 decisions per scale are one ballot.  Ring coordinates:
    //    row by = 0 .. TH + 1 is image row y0 + by - 1, bit e = 0 .. TW + 1 of a row is image column x0 + e - 1; bits 0 .. 63 live in
    //    sRow[.][by][0] (columns x0 - 1 .. x0 + 62), the last two columns in sRing (one ballot over the rows each).
    auto thr_bits = [&](int by, int bx) -> unsigned {
        const int gx = x0 + bx - 1, gy = y0 + by - 1;
        unsigned bits = 0;
        if (gx >= 0 && gx < cols && gy >= 0 && gy < rows) {
            const int lx = bx - 1 + HALO, ly = by - 1 + HALO;
            const int vc = (int)g[ly][lx] + cfg.thresh_c;
            if (kDefaultWindows) {
                if (box_is_fg<1>(I, ly, lx, vc)) bits |= 1u;
                if (box_is_fg<6>(I, ly, lx, vc)) bits |= 2u;
                if (box_is_fg<11>(I, ly, lx, vc)) bits |= 4u;
            } else {
#pragma unroll
                for (int s = 0; s < kScales; s++)
                    if (s < cfg.n_scales && box_is_fg_r(I, ly, lx, cfg.win_r[s], vc)) bits |= 1u << s;
            }
        }
        return bits;
    };
    {
        const int wave = tid >> 6, lane = tid & 63;
        for (int by = wave; by < TH + 2; by += 4) {
            const unsigned bits = thr_bits(by, lane);
#pragma unroll
            for (int s = 0; s < kScales; s++) {
                const unsigned long long row = __ballot((bits >> s) & 1u);
                if (lane == 0) sRow[s][by] = row;
            }
        }
        if (wave < 2) {                                             // columns TW and TW + 1 of the ring: lane = row
            const unsigned bits = lane < TH + 2 ? thr_bits(lane, TW + wave) : 0u;
#pragma unroll
            for (int s = 0; s < kScales; s++) {
                const unsigned long long col = __ballot((bits >> s) & 1u);
                if (lane == 0) sRing[s][wave] = col;
            }
        }
        // neighbour masks of four pixels of a row at once: bit d of a mask = the neighbour in direction d is foreground (0=E 1=NE 2=N
        // 3=NW 4=W 5=SW 6=S 7=SE).  Bits b0..b5 of a bit row are the columns x - 1 .. x + 4 of the four pixels x .. x + 3: the row above
        // gives NW / N / NE, the pixels' own row W / E, the row below SW / S / SE - one table per row, indexed by the six bits, holding
        // that row's share of the four mask bytes.
        if (tid < 3 * 64) {
            const unsigned r = tid >> 6, v = tid & 63u;
            unsigned w = 0;
            for (unsigned j = 0; j < 4; j++) {
                const unsigned a = (v >> j) & 1u, b = (v >> (j + 1)) & 1u, c = (v >> (j + 2)) & 1u;   // columns x_j - 1, x_j, x_j + 1
                const unsigned m = r == 0 ? (c << 1) | (b << 2) | (a << 3) : r == 1 ? c | (a << 4) : (a << 5) | (b << 6) | (c << 7);
                w |= m << (8 * j);
            }
            sLutRow[r][v] = w;
        }
    }
    __syncthreads();
    THR_STAMP(3);

    // 4. neighbour masks and border nodes (common.h: start candidates + cut states; staged in LDS, one reservation per tile in the
    //    frame's list): a thread takes four pixels of a row; six consecutive bits of each of the three bit rows around them index the tables above
    for (int u = tid; u < TH * TW / 4; u += 256) {
        const int ty = u / (TW / 4), tx4 = (u - ty * (TW / 4)) * 4;
        const int gy = y0 + ty;
        const bool row_on_grid = (gy & cfg.cut_mask) == 0;
        // which of the thread's four pixels lie on the cut lattice (its pitch divides the tile width and is a multiple of 4: x0 + tx4 is on it
        // iff tx4 is); pixels beyond the frame are never foreground
        const unsigned grid4 = row_on_grid ? 15u : ((tx4 & cfg.cut_mask) == 0 ? 1u : 0u);
        unsigned out[kScales] = {0, 0, 0};
        unsigned emit = 0;                                          // bit 4 s + j: pixel j of scale s may carry nodes
#pragma unroll
        for (int s = 0; s < kScales; s++) {
            unsigned six[3];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const int by = ty + r;
                unsigned v = (unsigned)(sRow[s][by] >> tx4) & 63u;                       // ring bits tx4 .. tx4 + 5
                if (tx4 == TW - 4) v |= ((unsigned)(sRing[s][0] >> by) & 1u) << 4 | ((unsigned)(sRing[s][1] >> by) & 1u) << 5;
                six[r] = v;
            }
            const unsigned m4 = sLutRow[0][six[0]] | sLutRow[1][six[1]] | sLutRow[2][six[2]];   // the four masks, one byte each
            out[s] = m4;
            // start candidates, byte-parallel (flags in the top bit of each byte): foreground and either outer type - not isolated, NE / N / NW
            // / W background - or hole type - E background, NE foreground
            const unsigned fgb = (((six[1] >> 1) & 15u) * 0x00204081u & 0x01010101u) << 7;
            const unsigned hole_x = (m4 & 0x03030303u) ^ 0x02020202u;
            const unsigned cand = fgb & ((byte_is_zero(m4 & 0x1E1E1E1Eu) & ~byte_is_zero(m4)) | byte_is_zero(hole_x));
            emit |= (((cand >> 7) | (cand >> 14) | (cand >> 21) | (cand >> 28)) & 15u) << (4 * s);
            emit |= (((six[1] >> 1) & grid4) & 15u) << (4 * s);     // foreground pixels on the cut lattice (bits 1..4 of the centre row)
        }
        for (unsigned rest = emit; rest; rest &= rest - 1u) {       // rare: a few pixels per tile
            const int bit = __ffs((int)rest) - 1, s = bit >> 2, j = bit & 3;
            const unsigned gx = (unsigned)(x0 + tx4 + j);
            const unsigned m = (out[s] >> (8 * j)) & 0xFFu;         // the pixel is foreground (both bits of the table require it)
            const bool outer = m != 0 && (m & 0x1Eu) == 0, hole = (m & 3u) == 2u;
            const unsigned s0 = outer ? (unsigned)first_outer(m) : (unsigned)first_hole(m);
            unsigned cut = 0;
            if (row_on_grid || (gx & (unsigned)cfg.cut_mask) == 0) cut = m & ~((m >> 1) | (m << 7)) & 0xFFu;   // neighbour s foreground, neighbour s + 1 background
            if (outer || hole) cut &= ~(1u << s0);                  // the candidate's own state is listed once
            const unsigned cnt = (unsigned)__popc(cut) + ((outer || hole) ? 1u : 0u);
            if (cnt == 0) continue;
            // the nodes of one pixel are consecutive in the frame's list (nodeplane holds the index of the first)
            const unsigned k = atomicAdd(&sNStart, cnt);
            const bool staged = k + cnt <= (unsigned)kBlockStarts;
            unsigned kk = 0;
            bool fits = true;
            if (!staged) {                                          // pathological tile (> 1024 nodes): go to the list directly
                for (unsigned e = k; e < (unsigned)kBlockStarts; e++) sStart[e] = kNone;      // staged slots this pixel does not use
                kk = atomicAdd(&n_starts[b], cnt);
                fits = kk + cnt <= cfg.cap_starts;
                if (fits) nodeplane[((size_t)b * kScales + s) * nbr_plane_bytes(rows, cfg.pitch) + nbr_index((int)gx, gy, cfg.pitch)] = kk;
                else atomicOr(&ctr->overflow, (unsigned)kOvfStarts);
            }
            unsigned e = 0;
            auto put = [&](unsigned v) {
                if (staged) sStart[k + e] = v;
                else if (fits) starts[(size_t)b * cfg.cap_starts + kk + e] = v;
                e++;
            };
            if (outer || hole) put(pack_node(gx, (unsigned)gy, s0, (unsigned)s, outer ? kNodeOuter : kNodeHole));
            for (unsigned c2 = cut; c2; c2 &= c2 - 1u) put(pack_node(gx, (unsigned)gy, (unsigned)(__ffs((int)c2) - 1), (unsigned)s, kNodeCut));
        }
        if (gy < rows) {
#pragma unroll
            for (int s = 0; s < kScales; s++) {
                unsigned* dst = reinterpret_cast<unsigned*>(nbr + ((size_t)b * kScales + s) * nbr_plane_bytes(rows, cfg.pitch) +
                                                            nbr_index(x0 + tx4, gy, cfg.pitch));
                *dst = out[s];
            }
        }
    }
    __syncthreads();
    THR_STAMP(4);
    const unsigned ns = min(sNStart, (unsigned)kBlockStarts);
    if (tid == 0 && ns > 0) sBase = atomicAdd(&n_starts[b], ns);
    __syncthreads();
    for (unsigned i = tid; i < ns; i += 256) {
        const unsigned k = sBase + i, e = sStart[i];
        if (k < cfg.cap_starts) {
            starts[(size_t)b * cfg.cap_starts + k] = e;
            if (e != kNone && (i == 0 || ((sStart[i - 1] ^ e) & kNodePixelMask) != 0))        // first node of its pixel
                nodeplane[((size_t)b * kScales + ((e >> 27) & 3u)) * nbr_plane_bytes(rows, cfg.pitch) +
                          nbr_index((int)(e & 0xFFFu), (int)((e >> 12) & 0xFFFu), cfg.pitch)] = k;
        } else {
            atomicOr(&ctr->overflow, (unsigned)kOvfStarts);
        }
    }
#ifdef ASLAM_THR_STAMPS
    THR_STAMP(5);
    if (tid == 0 && blockIdx.x == 4000) printf("thr tile: load %lld integral %lld threshold %lld masks %lld flush %lld\n", tst[1] - tst[0], tst[2] - tst[1], tst[3] - tst[2], tst[4] - tst[3], tst[5] - tst[4]);
#endif
}

// k_clear_counts: the work-queue heads and the per-frame list sizes of a detection call (one launch instead of six fills)
__global__ __launch_bounds__(256) void k_clear_counts(int nframes, Counters* ctr, unsigned* a0, unsigned* a1, unsigned* a2, unsigned* a3, unsigned* a4) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < kCounterHeads) reinterpret_cast<unsigned*>(ctr)[i] = 0u;
    if (i < nframes) { a0[i] = 0u; a1[i] = 0u; a2[i] = 0u; a3[i] = 0u; a4[i] = 0u; }
}

// ------------------------------------------------------------------------------------------------
// k_prefix : per-frame work counts -> ticket ranges of the two work-queue kernels (one workgroup)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prefix(int nframes, const unsigned* __restrict__ counts, unsigned cap, unsigned per_ticket,
                                                unsigned* __restrict__ pre /* nframes + 1 */) {
    __shared__ unsigned sPart[256];
    const int tid = threadIdx.x;
    const int per = (nframes + 255) / 256;
    unsigned s = 0;
    for (int k = 0; k < per; k++) {
        int f = tid * per + k;
        if (f < nframes) s += (min(counts[f], cap) + per_ticket - 1) / per_ticket;
    }
    sPart[tid] = s;
    __syncthreads();
    if (tid == 0) {
        unsigned acc = 0;
        for (int i = 0; i < 256; i++) { unsigned v = sPart[i]; sPart[i] = acc; acc += v; }
        pre[nframes] = acc;
    }
    __syncthreads();
    unsigned acc = sPart[tid];
    for (int k = 0; k < per; k++) {
        int f = tid * per + k;
        if (f < nframes) { pre[f] = acc; acc += (min(counts[f], cap) + per_ticket - 1) / per_ticket; }
    }
}

// ticket -> frame: largest f with pre[f] <= ticket (pre is non-decreasing, staged in LDS by the caller)
__device__ __forceinline__ int ticket_frame(const unsigned* sPre, int nframes, unsigned ticket) {
    int lo = 0, hi = nframes;                 // invariant: pre[lo] <= ticket < pre[hi]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (sPre[mid] <= ticket) lo = mid; else hi = mid;
    }
    return lo;
}

constexpr int kMaxFramesPerCall = 1024;

// The ticket ranges of a work-queue kernel into LDS: the prefix k_prefix made - or, for a call of one frame, straight from the
// frame's count (the host then skips the k_prefix launch: the single-frame call is a chain of small launches)
__device__ __forceinline__ void load_ticket_ranges(unsigned* sPre, const unsigned* __restrict__ pre, const unsigned* __restrict__ counts,
                                                   unsigned cap, int nframes, int tid, int nthreads) {
    if (nframes == 1) {
        if (tid == 0) { sPre[0] = 0u; sPre[1] = min(counts[0], cap); }
    } else {
        for (int i = tid; i <= nframes; i += nthreads) sPre[i] = pre[i];
    }
}

// ------------------------------------------------------------------------------------------------
// k_seg / k_link / k_trace_write : Suzuki-Abe border following without the sequential raster scan, in segments
// ------------------------------------------------------------------------------------------------
struct Walk {
    int x, y, s;
};
// one border-following step: first foreground neighbour counter-clockwise after the back direction
__device__ __forceinline__ void walk_step(Walk& w, unsigned m) {
    unsigned r = ((m | (m << 8)) >> (w.s + 1)) & 0xFFu;
    int k = __ffs((int)r) - 1;
    int s2 = (w.s + 1 + k) & 7;
    w.x += dir_dx(s2);
    w.y += dir_dy(s2);
    w.s = (s2 + 4) & 7;
}

constexpr int kTraceChunk = 64;      // tickets per grab: one per lane
constexpr int kSegRepsBusy = 32, kSegRepsRefill = 8;   // steps per round of k_seg when nothing can be picked up / when idle lanes wait for a refill (measured: 8 / 2 0.672 ms, 16 / 4 0.648, 32 / 8 0.627 per 320 frames)

// A border's canonical start - the state the sequential raster scan starts it from - is the start candidate of the border's own
// type (outer / hole) with the smallest raster key on the cycle: the scan meets the border's topmost-leftmost pixel (outer) or the
// hole's topmost-leftmost pixel (hole) first, and every other candidate of that type on the cycle lies on the same border, later
// in raster order.  Candidates of the other type on the cycle (a hole-type state that hugs the outer background, say) never matter.

// One border-following step as a table: (neighbour mask m, back direction s) -> dx + 1 | dy + 1 << 2 | new s << 4 | "this state is an
// outer-type start candidate" << 7 | "... a hole-type one" << 8 | "the first examined neighbour s + 1 is background: a cut state where
// the pixel lies on the cut lattice" << 9.  A walk is one dependent chain, so the ~35 instructions of walk_step become one LDS read.
__device__ __forceinline__ void build_step_table(unsigned short* sStep, int tid, int nthreads) {
    for (int i = tid; i < 256 * 8; i += nthreads) {
        const unsigned m = (unsigned)i >> 3;
        const int sd = i & 7;
        Walk t{0, 0, sd};
        if (m != 0) walk_step(t, m);
        const bool co = m != 0 && (m & 0x1Eu) == 0 && sd == first_outer(m);
        const bool ch = (m & 3u) == 2u && sd == first_hole(m);
        const bool cut = ((m >> sd) & 1u) != 0 && ((m >> ((sd + 1) & 7)) & 1u) == 0;
        sStep[i] = (unsigned short)((unsigned)(t.x + 1) | ((unsigned)(t.y + 1) << 2) | ((unsigned)t.s << 4) | (co ? 0x80u : 0u) | (ch ? 0x100u : 0u) |
                                    (cut ? 0x200u : 0u));
    }
}
__device__ __forceinline__ bool on_cut_grid(int x, int y, int mask) { return ((x & mask) == 0) || ((y & mask) == 0); }

// k_seg: work-queue kernel, one lane per node (ticket = node of the whole call; `pre` = per-frame prefix of the node counts).
// Every lane is a small state machine (idle -> walk -> idle); idle lanes are refilled from the queue with one atomic per 64
// tickets.  A lane walks from its node's state until it stands on the next node of the border, finds that node's index through the
// node plane, and records (next, steps, shoelace partial sum).
__global__ __launch_bounds__(64) void k_seg(const uint8_t* __restrict__ nbr, DetectCfg cfg, int nframes,
                                            const unsigned* __restrict__ starts, const unsigned* __restrict__ n_starts,
                                            const unsigned* __restrict__ nodeplane,
                                            const unsigned* __restrict__ pre, Counters* ctr,
                                            NodeRec* __restrict__ nodes) {
    __shared__ unsigned sPre[kMaxFramesPerCall + 1];
    __shared__ unsigned short sStep[256 * 8];
    const int lane = threadIdx.x & 63;
    build_step_table(sStep, lane, 64);
    load_ticket_ranges(sPre, pre, n_starts, cfg.cap_starts, nframes, lane, 64);
    __syncthreads();
    const unsigned total = sPre[nframes];
    const int pitch = cfg.pitch;
    const size_t plane_bytes = nbr_plane_bytes(cfg.rows, pitch);

    int mode = 0;                       // 0 idle, 1 walking
    bool drained = false;               // the queue is empty (wave-uniform)
    unsigned lo = 0, hi = 0;            // the wave's private ticket range (wave-uniform)
    const uint8_t* plane = nbr;
    const unsigned* idplane = nodeplane;
    Walk w{0, 0, 0};
    int n = 0, f = 0, area = 0;         // twice the signed area (shoelace); |step term| <= 4095, <= max_perim steps: fits 32 bits
    unsigned self = 0, state0 = 0;

    for (;;) {
        const unsigned long long idle = __ballot(mode == 0);
        if (idle != 0ull && !drained) {
            if (lo == hi) {                                       // wave-uniform
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(&ctr->q_trace, (unsigned)kTraceChunk);
                base = __shfl(base, 0);
                lo = base;
                hi = min(base + (unsigned)kTraceChunk, total);
                if (lo >= total) { drained = true; lo = hi = 0; }
            }
            const unsigned take = min((unsigned)__popcll(idle), hi - lo);
            const unsigned rank = (unsigned)__popcll(idle & ((1ull << lane) - 1ull));
            if (mode == 0 && rank < take) {
                const unsigned ticket = lo + rank;
                if (!(ticket >= sPre[f] && ticket < sPre[f + 1])) f = ticket_frame(sPre, nframes, ticket);
                self = ticket - sPre[f];
                const unsigned e = starts[(size_t)f * cfg.cap_starts + self];
                const unsigned type = (e >> 29) & 3u;
                if (type == kNodeInvalid) {                        // a staged slot k_threshold left unused
                    nodes[(size_t)f * cfg.cap_starts + self] = NodeRec{kNone, kNone, 0u, 0};
                } else {
                    const int x = (int)(e & 0xFFFu), y = (int)((e >> 12) & 0xFFFu);
                    const unsigned sc = (e >> 27) & 3u;
                    plane = nbr + ((size_t)f * kScales + sc) * plane_bytes;
                    idplane = nodeplane + ((size_t)f * kScales + sc) * plane_bytes;
                    state0 = e;
                    w = Walk{x, y, (int)((e >> 24) & 7u)};
                    area = 0;
                    n = 0;
                    mode = 1;
                }
            }
            lo += take;
        }
        if (__ballot(mode != 0) == 0ull) {
            if (drained) break;
            continue;
        }
        // several steps per round when nothing can be picked up anyway
        const int reps = (idle == 0ull || drained) ? kSegRepsBusy : kSegRepsRefill;      // wave-uniform
        for (int rep = 0; rep < reps; rep++) {
            if (mode == 1) {
                const unsigned m = plane[nbr_index(w.x, w.y, pitch)];
                const unsigned st = sStep[(m << 3) | (unsigned)w.s];
                const bool arrived = n > 0 && ((st & 0x180u) != 0 || ((st & 0x200u) != 0 && on_cut_grid(w.x, w.y, cfg.cut_mask)));
                if (arrived) {
                    // the node standing on this state: the nodes of a pixel are consecutive in the frame's list
                    const unsigned want = pack_node((unsigned)w.x, (unsigned)w.y, (unsigned)w.s, (state0 >> 27) & 3u, 0u);
                    const unsigned cnt = min(n_starts[f], cfg.cap_starts);
                    unsigned id = idplane[nbr_index(w.x, w.y, pitch)];
                    unsigned found = kNone;
                    for (int t = 0; t < 9 && id < cnt; t++, id++) {
                        const unsigned e2 = starts[(size_t)f * cfg.cap_starts + id];
                        if (((e2 ^ want) & kNodePixelMask) != 0) break;
                        if (((e2 ^ want) & kNodeStateMask) == 0) { found = id; break; }
                    }
                    nodes[(size_t)f * cfg.cap_starts + self] = NodeRec{state0, found, (unsigned)n, area};
                    mode = 0;
                } else {
                    const int ddx = (int)(st & 3u) - 1, ddy = (int)((st >> 2) & 3u) - 1;
                    area += w.x * ddy - ddx * w.y;                 // = px * (py + dy) - (px + dx) * py, unit steps in small integers
                    w.x += ddx; w.y += ddy; w.s = (int)((st >> 4) & 7u);
                    n++;
                    if (n > cfg.max_perim) {                       // no kept border is that long: cut
                        nodes[(size_t)f * cfg.cap_starts + self] = NodeRec{state0, kNone, (unsigned)n, area};
                            mode = 0;
                    }
                }
            }
            if (__ballot(mode != 0) == 0ull) break;
        }
    }
}

// k_link: the node cycles of one frame, resolved by one workgroup in LDS with pointer jumping (no lane ever walks a cycle):
//   1. election: every node learns the smallest node index of its cycle (its leader) - (leader, pointer) pairs in one 32-bit
//      word, pointer doubling for ceil(log2(nodes)) + 1 rounds; a cut segment poisons everything that leads into it;
//   2. ranking: every node learns its distance to the leader along the border ((distance, pointer) pairs, doubling until every
//      pointer stands on the leader); the leader's own distance is the border's length n;
//   3. borders with min_perim <= n <= max_perim get a slot; all their nodes add their shoelace sums into it and the start
//      candidates their keys, per type (LDS atomics);
//   4. the first candidate of the border's own type (outer / hole, by the sign of the area) - the start the sequential scan would
//      have used - emits the contour;
//   5. every node of an emitted border writes the tickets of the 64-point blocks of the contour that begin inside its segment.
// A frame with more nodes than the LDS image holds, or more borders than slots, is left to k_link_serial (flag in link_todo).
__device__ __forceinline__ int node_key(unsigned state, int cols) {
    return (int)((state >> 12) & 0xFFFu) * cols + (int)(state & 0xFFFu) + (((state >> 29) & 3u) == kNodeHole ? 1 : 0);
}
constexpr int kLinkThreads = 1024;
constexpr unsigned kLinkLdsNodes = 32000;        // x (4 B + 1 bit) = 129 KB; node indices < 0x8000
constexpr unsigned kLinkSlots = 1536;            // x 16 B = 24 KB
struct LinkSlot {                                // one kept border; the fields change their meaning from phase to phase:
    int a;                                       //   shoelace sum                    -> contour index (kNone: none emitted)
    int b;                                       //   smallest outer-type key         -> canonical key | hole << 31 (-1: none)
    int c;                                       //   smallest hole-type key          -> first write ticket
    unsigned d;                                  //   border length n (low 16 bits)   -> | position of the canonical start << 16
};
__global__ __launch_bounds__(kLinkThreads) void k_link(DetectCfg cfg, const unsigned* __restrict__ n_starts, Counters* ctr,
                                                       const NodeRec* __restrict__ nodes, unsigned lds_nodes, unsigned* __restrict__ link_todo,
                                                       ContourRec* __restrict__ contours, unsigned* __restrict__ n_contours,
                                                       unsigned* __restrict__ n_points, WriteRec* __restrict__ wlist, unsigned* __restrict__ n_write) {
    ASLAM_DYN_LDS(dyn_lds);
    __shared__ unsigned sNSlots, sActive;
    const int tid = threadIdx.x;
    const int f = blockIdx.x;
    const int cols = cfg.cols;
    const unsigned nn = min(n_starts[f], cfg.cap_starts);
    const NodeRec* gn = nodes + (size_t)f * cfg.cap_starts;
    if (nn > lds_nodes) {                                      // uniform
        if (tid == 0) link_todo[f] = 1u;
        return;
    }
    // 4 B + 1 bit of LDS per node (next and steps are read from the node records again where a phase starts from them).  A word whose
    // top bit is set belongs to a leader (after phase 3: n | (0x8000 | slot) << 16, slot 0x7FFF = none) or to a poisoned node (0xFFFF0000);
    // node indices stay below 0x8000.
    unsigned* sPair = reinterpret_cast<unsigned*>(dyn_lds);                     // election: leader + 1 | pointer << 16; ranking: distance | pointer << 16
    LinkSlot* sSlot = reinterpret_cast<LinkSlot*>(sPair + lds_nodes);
    unsigned* sIsLeader = reinterpret_cast<unsigned*>(sSlot + kLinkSlots);      // one bit per node
    if (tid == 0) { sNSlots = 0; sActive = 0; }
    for (unsigned i = tid; i < (nn + 31u) / 32u; i += kLinkThreads) sIsLeader[i] = 0u;

    // ---- 1. election ----
    for (unsigned i = tid; i < nn; i += kLinkThreads) {
        const NodeRec r = gn[i];
        const bool dead = r.state == kNone || r.nxt == kNone || r.nxt >= nn;
        sPair[i] = dead ? (0u | (i << 16)) : ((i + 1u) | (r.nxt << 16));
    }
    __syncthreads();
    int rounds = 1;
    while ((1u << rounds) < nn) rounds++;
    for (int r = 0; r <= rounds; r++) {
        for (unsigned i = tid; i < nn; i += kLinkThreads) {
            const unsigned w = sPair[i], wp = sPair[w >> 16];
            sPair[i] = min(w & 0xFFFFu, wp & 0xFFFFu) | (wp & 0xFFFF0000u);
        }
        __syncthreads();
    }
    // ---- 2. ranking ----
    for (unsigned i = tid; i < nn; i += kLinkThreads) {
        const unsigned lead = sPair[i] & 0xFFFFu;
        unsigned w = 0xFFFF0000u;                              // poisoned
        if (lead != 0u) {
            const NodeRec r = gn[i];
            w = min(r.len, 0xFFFFu) | (r.nxt << 16);           // (steps, next): a live node's next is valid
            if (lead == i + 1u) atomicOr(&sIsLeader[i >> 5], 1u << (i & 31u));
        }
        sPair[i] = w;                                          // (every lane reads its own word only in this loop)
    }
    __syncthreads();
    for (int r = 0; r <= rounds + 1; r++) {
        bool active = false;
        for (unsigned i = tid; i < nn; i += kLinkThreads) {
            const unsigned w = sPair[i], p = w >> 16;
            if (p >= 0x8000u || ((sIsLeader[p >> 5] >> (p & 31u)) & 1u)) continue;     // poisoned, or the pointer stands on the leader
            const unsigned wp = sPair[p];
            sPair[i] = min((w & 0xFFFFu) + (wp & 0xFFFFu), 0xFFFFu) | (wp & 0xFFFF0000u);
            active = true;
        }
        if (active) sActive = (unsigned)r + 1u;                // (a round number: never reset, so no write races with a reset)
        __syncthreads();
        const bool any = sActive == (unsigned)r + 1u;
        __syncthreads();
        if (!any) break;                                       // uniform
    }
    // ---- 3. slots and sums ----
    for (unsigned i = tid; i < nn; i += kLinkThreads) {
        if (!((sIsLeader[i >> 5] >> (i & 31u)) & 1u)) continue;                        // leaders only
        const unsigned w = sPair[i], n = w & 0xFFFFu;          // all the way round
        unsigned v = 0x7FFFu;
        if ((int)n >= cfg.min_perim && (int)n <= cfg.max_perim && (w >> 16) == i) {
            const unsigned slot = atomicAdd(&sNSlots, 1u);
            if (slot < kLinkSlots) {
                sSlot[slot] = LinkSlot{0, INT_MAX, INT_MAX, n};
                v = slot;
            }
        }
        sPair[i] = n | ((0x8000u | v) << 16);                  // (no other lane reads a leader's word in this loop)
    }
    __syncthreads();
    if (sNSlots > kLinkSlots) {                                // uniform; nothing has left the workgroup yet
        if (tid == 0) link_todo[f] = 1u;
        return;
    }
    if (tid == 0) link_todo[f] = 0u;
    auto slot_of = [&](unsigned i) -> int {                    // the slot of node i's border, -1 without one
        unsigned w = sPair[i];
        if (!(w >> 31)) w = sPair[w >> 16];                    // the leader's word (a pointer that never reached a leader cannot be: the rounds suffice)
        const unsigned v = (w >> 16) & 0x7FFFu;
        return (w >> 31) && v != 0x7FFFu ? (int)v : -1;
    };
    for (unsigned i = tid; i < nn; i += kLinkThreads) {
        const int sl = slot_of(i);
        if (sl < 0) continue;
        const NodeRec r = gn[i];
        atomicAdd(&sSlot[sl].a, r.area);
        const unsigned type = (r.state >> 29) & 3u;
        if (type == kNodeOuter) atomicMin(&sSlot[sl].b, node_key(r.state, cols));
        if (type == kNodeHole) atomicMin(&sSlot[sl].c, node_key(r.state, cols));
    }
    __syncthreads();
    // the leader turns (area, smallest outer key, smallest hole key) into the border's type and its canonical key
    for (unsigned i = tid; i < nn; i += kLinkThreads) {
        if (!((sIsLeader[i >> 5] >> (i & 31u)) & 1u)) continue;
        const unsigned v = (sPair[i] >> 16) & 0x7FFFu;
        if (v == 0x7FFFu) continue;
        LinkSlot& s = sSlot[v];
        const bool is_hole = s.a > 0;                          // outer borders run counter-clockwise on screen
        const int ckey = is_hole ? s.c : s.b;
        s.b = ckey == INT_MAX ? -1 : (int)((unsigned)ckey | (is_hole ? 0x80000000u : 0u));
        s.a = (int)kNone;                                      // no contour (yet)
    }
    __syncthreads();
    // ---- 4. the canonical start emits the contour ----
    for (unsigned i = tid; i < nn; i += kLinkThreads) {
        const int sl = slot_of(i);
        if (sl < 0) continue;
        const unsigned want = (unsigned)sSlot[sl].b;
        if (want == 0xFFFFFFFFu) continue;                     // no start candidate of the border's own type
        const NodeRec r = gn[i];
        const unsigned type = (r.state >> 29) & 3u;
        const bool is_hole = (want >> 31) != 0;
        const int ckey = (int)(want & 0x7FFFFFFFu);
        if (type != (is_hole ? kNodeHole : kNodeOuter) || node_key(r.state, cols) != ckey) continue;
        const unsigned n = sSlot[sl].d & 0xFFFFu, wcap = (n + 63u) / 64u;
        const unsigned ci = atomicAdd(&n_contours[f], 1u);
        const unsigned off = atomicAdd(&n_points[f], n);
        const unsigned wbase = atomicAdd(&n_write[f], wcap);
        const unsigned sc = (r.state >> 27) & 3u;
        const int cx = ckey % cols - (is_hole ? 1 : 0), cy = ckey / cols;
        if (ci >= cfg.cap_contours) {
            atomicOr(&ctr->overflow, (unsigned)kOvfContours);
        } else if ((unsigned long long)off + n > cfg.cap_points || (unsigned long long)wbase + wcap > cfg.cap_write) {
            atomicOr(&ctr->overflow, (unsigned)kOvfPoints);
            contours[(size_t)f * cfg.cap_contours + ci] = ContourRec{(unsigned)f, sc, (unsigned)ckey, 0u, 0u, (short)cx, (short)cy, 0, {0u, 0u}};
        } else {
            contours[(size_t)f * cfg.cap_contours + ci] = ContourRec{(unsigned)f, sc, (unsigned)ckey, n, off, (short)cx, (short)cy,
                                                                    (int)((r.state >> 24) & 7u), {0u, 0u}};
            const unsigned dist = sPair[i] & 0xFFFFu;          // to the leader; the leader's own is n
            sSlot[sl].d = n | ((n - dist) << 16);              // (the other lanes of this phase read the low half only: it does not change)
            sSlot[sl].c = (int)wbase;
            sSlot[sl].a = (int)ci;
        }
    }
    __syncthreads();
    // ---- 5. write tickets: the 64-point blocks of the contour that begin inside this node's segment ----
    for (unsigned i = tid; i < nn; i += kLinkThreads) {
        const int sl = slot_of(i);
        if (sl < 0) continue;
        const LinkSlot s = sSlot[sl];
        const unsigned ci = (unsigned)s.a;
        if (ci == kNone) continue;
        const NodeRec r = gn[i];
        const unsigned n = s.d & 0xFFFFu, pos_canon = s.d >> 16, len = r.len;
        unsigned rel = (n - (sPair[i] & 0xFFFFu)) + n - pos_canon;            // position after the canonical start, mod n
        while (rel >= n) rel -= n;
        const unsigned state = r.state;
        WriteRec* wl = wlist + (size_t)f * cfg.cap_write + (unsigned)s.c;
        for (unsigned b = (rel + 63u) / 64u; 64u * b < min(rel + len, n); b++)
            wl[b] = WriteRec{state, ci, 64u * b, min(64u, n - 64u * b) | ((64u * b - rel) << 16)};
    }
}

// k_link_serial: the frames k_link left (link_todo), through global memory, one workgroup per frame.  One lane per node; only
// start candidates do anything: such a lane hops from node to node along its border, one hop per round, while the idle lanes of
// its wave are refilled from the frame's ticket counter.  A candidate with a smaller key belongs to a lane that will do (or hand
// on) this border: stop.  The lane that gets back to its own node has the smallest key of the border; it emits the contour from
// the first candidate of the border's own type, then goes round once more and cuts the border into write tickets: runs of whole
// segments of at least kWriteChunk points each.
__global__ __launch_bounds__(512) void k_link_serial(DetectCfg cfg, const unsigned* __restrict__ n_starts, Counters* ctr,
                                                     const NodeRec* __restrict__ nodes, const unsigned* __restrict__ link_todo,
                                                     ContourRec* __restrict__ contours, unsigned* __restrict__ n_contours,
                                                     unsigned* __restrict__ n_points, WriteRec* __restrict__ wlist, unsigned* __restrict__ n_write) {
    __shared__ unsigned sNext;
    const int tid = threadIdx.x, lane = tid & 63;
    const int f = blockIdx.x;
    if (link_todo[f] == 0u) return;                            // uniform
    const int cols = cfg.cols;
    const unsigned nn = min(n_starts[f], cfg.cap_starts);
    const NodeRec* fn = nodes + (size_t)f * cfg.cap_starts;
    if (tid == 0) sNext = 0;
    __syncthreads();

    int mode = 0;                       // 0 idle, 1 first lap (election), 2 second lap (write tickets)
    bool drained = false;
    unsigned lo = 0, hi = 0;
    unsigned self = 0, n = 0, hops = 0;
    int area = 0, key0 = 0;
    int kmin_outer = INT_MAX, kmin_hole = INT_MAX;
    unsigned kpos_outer = 0, kpos_hole = 0;
    NodeRec cur{0u, 0u, 0u, 0};
    unsigned ci = 0, off = 0, kpos = 0, id = 0, pos = 0, wbase = 0, wcount = 0, wcap = 0, chunk_state = 0, chunk_rel = 0, chunk_len = 0, sc = 0;
    int ckey = 0, s_canon = 0;
    bool is_hole = false;

    for (;;) {
        const unsigned long long idle = __ballot(mode == 0);
        if (idle != 0ull && !drained) {
            if (lo == hi) {                                       // wave-uniform
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(&sNext, (unsigned)kTraceChunk);
                base = __shfl(base, 0);
                lo = base;
                hi = min(base + (unsigned)kTraceChunk, nn);
                if (lo >= nn) { drained = true; lo = hi = 0; }
            }
            const unsigned take = min((unsigned)__popcll(idle), hi - lo);
            const unsigned rank = (unsigned)__popcll(idle & ((1ull << lane) - 1ull));
            if (mode == 0 && rank < take) {
                self = lo + rank;
                cur = fn[self];
                const unsigned type = (cur.state >> 29) & 3u;
                if (cur.state != kNone && (type == kNodeOuter || type == kNodeHole)) {
                    key0 = node_key(cur.state, cols);
                    kmin_outer = type == kNodeOuter ? key0 : INT_MAX;
                    kmin_hole = type == kNodeHole ? key0 : INT_MAX;
                    kpos_outer = kpos_hole = 0;
                    sc = (cur.state >> 27) & 3u;
                    n = 0; area = 0; hops = 0;
                    mode = 1;
                }
            }
            lo += take;
        }
        if (__ballot(mode != 0) == 0ull) {
            if (drained) break;
            continue;
        }
        if (mode == 1) {
            n += cur.len;
            area += cur.area;
            const unsigned nx = cur.nxt;
            if (nx == self) {
                // closed: this lane holds the smallest surviving key of the border
                mode = 0;
                is_hole = area > 0;                                // outer borders run counter-clockwise on screen
                ckey = is_hole ? kmin_hole : kmin_outer;
                if (ckey != INT_MAX && (int)n >= cfg.min_perim && (int)n <= cfg.max_perim) {
                    kpos = is_hole ? kpos_hole : kpos_outer;
                    wcap = n / (unsigned)kWriteChunk + 1u;         // every ticket but the last covers >= kWriteChunk points
                    ci = atomicAdd(&n_contours[f], 1u);
                    off = atomicAdd(&n_points[f], n);
                    wbase = atomicAdd(&n_write[f], wcap);
                    const int cx = ckey % cols - (is_hole ? 1 : 0), cy = ckey / cols;
                    if (ci >= cfg.cap_contours) {
                        atomicOr(&ctr->overflow, (unsigned)kOvfContours);
                    } else if ((unsigned long long)off + n > cfg.cap_points || (unsigned long long)wbase + wcap > cfg.cap_write) {
                        atomicOr(&ctr->overflow, (unsigned)kOvfPoints);
                        contours[(size_t)f * cfg.cap_contours + ci] = ContourRec{(unsigned)f, sc, (unsigned)ckey, 0u, 0u, (short)cx, (short)cy, 0, {0u, 0u}};
                    } else {
                        id = self; pos = 0; hops = 0; wcount = 0; chunk_len = 0; s_canon = 0;
                        mode = 2;
                    }
                }
            } else if (nx == kNone || n > (unsigned)cfg.max_perim || ++hops > nn) {
                mode = 0;
            } else {
                cur = fn[nx];
                const unsigned t2 = (cur.state >> 29) & 3u;
                if (t2 == kNodeOuter || t2 == kNodeHole) {
                    const int key = node_key(cur.state, cols);
                    if (key < key0) mode = 0;                       // the candidate with the smaller key owns this border
                    if (t2 == kNodeHole) { if (key < kmin_hole) { kmin_hole = key; kpos_hole = n; } }
                    else if (key < kmin_outer) { kmin_outer = key; kpos_outer = n; }
                }
            }
        } else if (mode == 2) {
            // point 0 of the contour = the canonical start = walk step kpos of this lane
            const NodeRec r = fn[id];
            unsigned rel = pos + n - kpos;
            if (rel >= n) rel -= n;
            if (rel == 0) s_canon = (int)((r.state >> 24) & 7u);
            if (chunk_len == 0) { chunk_state = r.state; chunk_rel = rel; }
            chunk_len += r.len;
            pos += r.len;
            id = r.nxt;
            const bool done = id == self || ++hops > nn;
            WriteRec* wl = wlist + (size_t)f * cfg.cap_write + wbase;
            if (chunk_len >= (unsigned)kWriteChunk || done) {
                if (wcount < wcap) wl[wcount++] = WriteRec{chunk_state, ci, chunk_rel, chunk_len};   // (no skip)
                chunk_len = 0;
            }
            if (done) {
                for (; wcount < wcap; wcount++) wl[wcount] = WriteRec{0u, kNone, 0u, 0u};     // reserved, not needed
                contours[(size_t)f * cfg.cap_contours + ci] = ContourRec{(unsigned)f, sc, (unsigned)ckey, n, off,
                                                                        (short)(ckey % cols - (is_hole ? 1 : 0)), (short)(ckey / cols), s_canon, {0u, 0u}};
                mode = 0;
            }
        }
    }
}

// k_trace_write: the points of every kept contour.  Ticket = write ticket of k_link (`pre` = per-frame prefix of their counts): a
// lane replays its run of segments, 64 steps per round, into LDS, and the wave writes run after run with coalesced stores.
__global__ __launch_bounds__(256) void k_trace_write(const uint8_t* __restrict__ nbr, DetectCfg cfg, int nframes,
                                                     const unsigned* __restrict__ pre, Counters* ctr,
                                                     const ContourRec* __restrict__ contours, const WriteRec* __restrict__ wlist,
                                                     const unsigned* __restrict__ n_write, unsigned* __restrict__ points) {
    __shared__ unsigned sPre[kMaxFramesPerCall + 1];
    __shared__ unsigned sPts[4][64][64];
    __shared__ unsigned short sStep[256 * 8];
    __shared__ unsigned sBase;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    build_step_table(sStep, tid, 256);
    load_ticket_ranges(sPre, pre, n_write, cfg.cap_write, nframes, tid, 256);
    __syncthreads();
    const unsigned total = sPre[nframes];
    const int pitch = cfg.pitch;
    for (;;) {
        if (tid == 0) sBase = atomicAdd(&ctr->q_write, 256u);
        __syncthreads();
        const unsigned base = sBase;
        __syncthreads();
        if (base >= total) break;                              // uniform
        const unsigned ticket = base + tid;
        unsigned* dst = points;
        const uint8_t* plane = nbr;
        int n = 1, idx0 = 0, left = 0;
        Walk w{0, 0, 0};
        if (ticket < total) {
            const int f = ticket_frame(sPre, nframes, ticket);
            const WriteRec wr = wlist[(size_t)f * cfg.cap_write + (ticket - sPre[f])];
            if (wr.ci != kNone) {
                const ContourRec rec = contours[(size_t)f * cfg.cap_contours + wr.ci];
                plane = nbr + ((size_t)f * kScales + rec.scale) * nbr_plane_bytes(cfg.rows, pitch);
                dst = points + (size_t)f * cfg.cap_points + rec.off;
                n = (int)rec.n;
                idx0 = (int)wr.rel;
                left = (int)(wr.cnt & 0xFFFFu);
                w = Walk{(int)(wr.state & 0xFFFu), (int)((wr.state >> 12) & 0xFFFu), (int)((wr.state >> 24) & 7u)};
                for (int t = (int)(wr.cnt >> 16); t > 0; t--) {    // the block begins this many steps into the segment
                    const unsigned st = sStep[((unsigned)plane[nbr_index(w.x, w.y, pitch)] << 3) | (unsigned)w.s];
                    w.x += (int)(st & 3u) - 1; w.y += (int)((st >> 2) & 3u) - 1; w.s = (int)((st >> 4) & 7u);
                }
            }
        }
        const unsigned long long dptr = (unsigned long long)dst;
        while (__ballot(left > 0) != 0ull) {                    // wave-uniform
            // every lane replays (up to) 64 steps of its run into LDS (skewed so that the transposed read below is conflict-free) ...
            const int cnt = min(left, 64);
            for (int t = 0; t < cnt; t++) {
                sPts[wave][lane][(t + lane) & 63] = ((unsigned)w.x & 0xFFFFu) | ((unsigned)w.y << 16);
                const unsigned st = sStep[((unsigned)plane[nbr_index(w.x, w.y, pitch)] << 3) | (unsigned)w.s];
                w.x += (int)(st & 3u) - 1; w.y += (int)((st >> 2) & 3u) - 1; w.s = (int)((st >> 4) & 7u);
            }
            // ... and the wave writes run after run, 64 consecutive points (256 B) per store
            for (int sgm = 0; sgm < 64; sgm++) {
                const int scnt = __shfl(cnt, sgm);
                if (scnt == 0) continue;                       // wave-uniform
                const int sn = __shfl(n, sgm), sidx = __shfl(idx0, sgm);
                const unsigned lo32 = __shfl((unsigned)dptr, sgm), hi32 = __shfl((unsigned)(dptr >> 32), sgm);
                unsigned* sdst = (unsigned*)(((unsigned long long)hi32 << 32) | lo32);
                if (lane < scnt) {
                    int o = sidx + lane;
                    if (o >= sn) o -= sn;
                    sdst[o] = sPts[wave][sgm][(lane + sgm) & 63];
                }
            }
            idx0 += cnt;
            if (idx0 >= n) idx0 -= n;
            left -= cnt;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_quads : approxPolyDP (closed) + quad tests, one WAVEFRONT per contour
// ------------------------------------------------------------------------------------------------
struct IPt { int x, y; };
constexpr int kQuadLdsPts = 1536;     // 6 KB per wavefront (with the rest: 15 wavefronts per CU)
constexpr int kQuadGrab = 4;          // contours per grab of k_quads

// wave-wide "first maximum": the sequential scans use a strict '>' so the earliest position attaining the maximum wins;
// distances here are integers held exactly in doubles, so the parallel reduction is bit-identical to the scan.
__device__ __forceinline__ void wave_first_max(double& d, int& pos) {
    for (int o = 32; o > 0; o >>= 1) {
        double od = __shfl_xor(d, o);
        int op = __shfl_xor(pos, o);
        if (od > d || (od == d && op < pos)) { d = od; pos = op; }
    }
}
// The same for distances that are non-negative integers below 2^32 held exactly in doubles (approxPolyDP: squared distances and
// cross products of pixel coordinates, < 2^26) and positions >= 0: one 64-bit key (distance, ~position) per lane, two thirds of the
// lane-to-lane traffic of the general form.
__device__ __forceinline__ void wave_first_max_exact(double& d, int& pos) {
    unsigned long long key = ((unsigned long long)(unsigned)d << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)pos);
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long ok = __shfl_xor(key, o);
        key = ok > key ? ok : key;
    }
    d = (double)(unsigned)(key >> 32);
    pos = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
}

// Returns the number of vertices (<= 8) written to out, or -1 when the result cannot have 4 vertices.
// Early exit rule: every stack slice yields at least one vertex and the final clean-up removes at most
// floor(count/2) of them, so new_count + stack > 8 can never end at 4.  All lanes run the same control flow.
#ifdef ASLAM_QUADS_STAMPS
#define QST(i) do { const long long t_ = clock64(); qst[i] += t_ - qlast; qlast = t_; } while (0)
#else
#define QST(i) do { } while (0)
#endif
// kLds: the points come from the wavefront's LDS copy `spts` (plain DS reads; a generic pointer into LDS is never formed), else from
// global memory
template <bool kLds>
__device__ int approx_poly_closed_body(const unsigned* __restrict__ gsrc, const unsigned* spts, int count, double eps, IPt* out, int lane
#ifdef ASLAM_QUADS_STAMPS
                                       , long long* qst, long long& qlast
#endif
                                       ) {
    auto P = [&](int i) -> unsigned { return kLds ? spts[i] : gsrc[i]; };
    auto ld = [&](int i) -> IPt { const unsigned v = P(i); return IPt{(int)(short)(v & 0xFFFFu), (int)(short)(v >> 16)}; };
    struct Range { int start, end; };
    // wave-uniform work arrays: one copy per wavefront in LDS (every lane stores the same value) instead of per-lane scratch
    __shared__ Range stack[10];
    __shared__ IPt dst[9];
    int top = 0;
    Range slice{0, 0}, right_slice{0, 0};
    IPt start_pt{-1000000, -1000000}, end_pt{0, 0}, pt{0, 0};
    int pos = 0, new_count = 0;
    bool le_eps = false;
    eps *= eps;

    right_slice.start = 0;
    for (int it = 0; it < 3; it++) {
        pos = (pos + right_slice.start) % count;             // index of this iteration's start point
        start_pt = ld(pos);
        double best = 0.0;
        int bestj = 0x7fffffff;
        // (four loads in flight per lane: the points were written by another kernel, every load is a miss of this CU's caches)
        for (int j0 = 1 + lane; j0 < count; j0 += 4 * 64) {
            unsigned v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int j = j0 + 64 * u;
                int idx = pos + j;
                if (idx >= count) idx -= count;
                v[u] = j < count ? P(idx) : 0u;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int j = j0 + 64 * u;
                if (j < count) {
                    const double dx = (int)(short)(v[u] & 0xFFFFu) - start_pt.x, dy = (int)(short)(v[u] >> 16) - start_pt.y;
                    const double dist = dx * dx + dy * dy;
                    if (dist > best) { best = dist; bestj = j; }
                }
            }
        }
        QST(0);
        wave_first_max_exact(best, bestj);
        QST(1);
        if (best > 0.0) right_slice.start = bestj;          // unchanged when no point is farther than 0 (as in the scan)
        le_eps = best <= eps;
        // the scan reads `count` points in all, so `pos` is back at the start index
    }
    if (!le_eps) {
        right_slice.end = slice.start = pos % count;
        slice.end = right_slice.start = (right_slice.start + slice.start) % count;
        stack[top++] = right_slice;
        stack[top++] = slice;
    } else {
        dst[new_count++] = start_pt;
    }
    while (top > 0) {
        slice = stack[--top];
        end_pt = ld(slice.end);
        start_pt = ld(slice.start);
        int first = slice.start + 1;
        if (first >= count) first = 0;
        if (first != slice.end) {
            int len = slice.end - first;                      // interior points first .. end-1 (cyclic)
            if (len < 0) len += count;
            double dx = end_pt.x - start_pt.x, dy = end_pt.y - start_pt.y;
            double best = 0.0;
            int bestt = 0x7fffffff;
            QST(2);
            for (int t0 = lane; t0 < len; t0 += 4 * 64) {
                unsigned v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int t = t0 + 64 * u;
                    int idx = first + t;
                    if (idx >= count) idx -= count;
                    v[u] = t < len ? P(idx) : 0u;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int t = t0 + 64 * u;
                    if (t < len) {
                        const int px = (int)(short)(v[u] & 0xFFFFu), py = (int)(short)(v[u] >> 16);
                        const double dist = fabs((py - start_pt.y) * dx - (px - start_pt.x) * dy);
                        if (dist > best) { best = dist; bestt = t; }
                    }
                }
            }
            QST(3);
            wave_first_max_exact(best, bestt);
            QST(1);
            if (best > 0.0) { int idx = first + bestt; if (idx >= count) idx -= count; right_slice.start = idx; }
            le_eps = best * best <= eps * (dx * dx + dy * dy);
        } else {
            le_eps = true;
        }
        if (le_eps) {
            dst[new_count++] = start_pt;
        } else {
            right_slice.end = slice.end;
            slice.end = right_slice.start;
            stack[top++] = right_slice;
            stack[top++] = slice;
        }
        if (new_count + top > 8) return -1;
    }
    // final clean-up
    QST(2);
    int count2 = new_count;
    pos = count2 - 1;
    start_pt = dst[pos]; if (++pos >= count2) pos = 0;
    int wpos = pos;
    pt = dst[pos]; if (++pos >= count2) pos = 0;
    for (int i = 0; i < count2 && new_count > 2; i++) {
        end_pt = dst[pos]; if (++pos >= count2) pos = 0;
        double dx = end_pt.x - start_pt.x, dy = end_pt.y - start_pt.y;
        double dist = fabs((pt.x - start_pt.x) * dy - (pt.y - start_pt.y) * dx);
        double sip = (double)(pt.x - start_pt.x) * (end_pt.x - pt.x) + (double)(pt.y - start_pt.y) * (end_pt.y - pt.y);
        if (dist * dist <= 0.5 * eps * (dx * dx + dy * dy) && dx != 0 && dy != 0 && sip >= 0) {
            new_count--;
            dst[wpos] = start_pt = end_pt;
            if (++wpos >= count2) wpos = 0;
            pt = dst[pos]; if (++pos >= count2) pos = 0;
            i++;
            continue;
        }
        dst[wpos] = start_pt = pt;
        if (++wpos >= count2) wpos = 0;
        pt = end_pt;
    }
    for (int i = 0; i < new_count; i++) out[i] = dst[i];
    QST(4);
    return new_count;
}

// Returns the number of vertices (<= 8) written to out, or -1 when the result cannot have 4 vertices (see the body).
// The points were written by another kernel: every access would be a miss of this CU's caches, and the passes are chains of
// dependent accesses (slice end points, then the slice).  A contour of up to kQuadLdsPts points is read once, all loads in flight
// together, into LDS.
__device__ int approx_poly_closed_wave(const unsigned* __restrict__ src, int count, double eps, IPt* out, int lane
#ifdef ASLAM_QUADS_STAMPS
                                       , long long* qst, long long& qlast
#endif
                                       ) {
    __shared__ unsigned spts[kQuadLdsPts];
    if (count <= kQuadLdsPts) {                                 // wave-uniform
        __syncthreads();                                        // (the workgroup is this one wavefront) the previous contour's reads are done
        for (int i0 = lane; i0 < count; i0 += 8 * 64) {
            unsigned v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = src[min(i0 + 64 * u, count - 1)];      // (unconditional: a guarded load is a branch with its own wait)
#pragma unroll
            for (int u = 0; u < 8; u++) if (i0 + 64 * u < count) spts[i0 + 64 * u] = v[u];
        }
        __syncthreads();
#ifdef ASLAM_QUADS_STAMPS
        return approx_poly_closed_body<true>(src, spts, count, eps, out, lane, qst, qlast);
#else
        return approx_poly_closed_body<true>(src, spts, count, eps, out, lane);
#endif
    }
#ifdef ASLAM_QUADS_STAMPS
    return approx_poly_closed_body<false>(src, spts, count, eps, out, lane, qst, qlast);
#else
    return approx_poly_closed_body<false>(src, spts, count, eps, out, lane);
#endif
}

__device__ __forceinline__ bool quad_is_convex(const IPt* p) {
    const int n = 4;
    IPt prev_pt = p[2], cur_pt = p[3];
    int dx0 = cur_pt.x - prev_pt.x, dy0 = cur_pt.y - prev_pt.y;
    int orientation = 0;
    for (int i = 0; i < n; i++) {
        prev_pt = cur_pt;
        cur_pt = p[i];
        int dx = cur_pt.x - prev_pt.x, dy = cur_pt.y - prev_pt.y;
        int dxdy0 = dx * dy0, dydx0 = dy * dx0;
        orientation |= (dydx0 > dxdy0) ? 1 : ((dydx0 < dxdy0) ? 2 : 3);
        if (orientation == 3) return false;
        dx0 = dx;
        dy0 = dy;
    }
    return true;
}

__global__ __launch_bounds__(64) void k_quads(DetectCfg cfg, int nframes, Counters* ctr, const ContourRec* __restrict__ contours,
                                              const unsigned* __restrict__ n_contours, const unsigned* __restrict__ pre,
                                              const unsigned* __restrict__ points, CandRec* __restrict__ cands,
                                              unsigned* __restrict__ n_cand) {
    __shared__ unsigned sPre[kMaxFramesPerCall + 1];
    const int lane = threadIdx.x & 63;
    load_ticket_ranges(sPre, pre, n_contours, cfg.cap_contours, nframes, lane, 64);
    __syncthreads();
    const unsigned total = sPre[nframes];
#ifdef ASLAM_QUADS_STAMPS
    long long qst[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long qlast = clock64();
    int qn = 0;
#endif
    // kQuadGrab consecutive contours per grab: one atomic and one round of record loads (all in flight together) for the lot -
    // unless there are too few contours to go round (a single frame): then one at a time, on as many waves as possible
    const unsigned grab = total > 2u * (unsigned)kQuadGrab * gridDim.x ? (unsigned)kQuadGrab : 1u;
    for (;;) {
        unsigned base = 0;
        QST(5);
        if (lane == 0) base = atomicAdd(&ctr->q_quads, grab);
        base = __shfl(base, 0);
        if (base >= total) break;
        const unsigned limit = min(total, base + grab);
        ContourRec recs[kQuadGrab];
        int fs[kQuadGrab];
#pragma unroll
        for (int u = 0; u < kQuadGrab; u++) {
            const unsigned t = min(base + (unsigned)u, total - 1u);
            fs[u] = ticket_frame(sPre, nframes, t);
            recs[u] = contours[(size_t)fs[u] * cfg.cap_contours + (t - sPre[fs[u]])];
        }
        QST(6);
#pragma unroll
        for (int u = 0; u < kQuadGrab; u++) {
        const ContourRec rec = recs[u];
        const int f = fs[u];
        if (base + (unsigned)u < limit && rec.n > 0) {         // wave-uniform
            __shared__ IPt q[8];                                // wave-uniform, like the work arrays of approx_poly_closed_wave
#ifdef ASLAM_QUADS_STAMPS
            qn++;
            int nv = approx_poly_closed_wave(points + (size_t)f * cfg.cap_points + rec.off, (int)rec.n,
                                             (double)rec.n * cfg.approx_rate, q, lane, qst, qlast);
#else
            int nv = approx_poly_closed_wave(points + (size_t)f * cfg.cap_points + rec.off, (int)rec.n,
                                             (double)rec.n * cfg.approx_rate, q, lane);
#endif
            bool ok = nv == 4 && quad_is_convex(q);
            if (ok) {
                int mx = max(cfg.cols, cfg.rows);
                double minDistSq = (double)mx * mx;
                for (int j = 0; j < 4; j++) {
                    double ddx = (double)(q[j].x - q[(j + 1) & 3].x), ddy = (double)(q[j].y - q[(j + 1) & 3].y);
                    double d = ddx * ddx + ddy * ddy;
                    minDistSq = fmin(minDistSq, d);
                }
                double minCornerDistancePixels = (double)rec.n * cfg.min_corner_rate;
                if (minDistSq < minCornerDistancePixels * minCornerDistancePixels) ok = false;
                for (int j = 0; j < 4; j++)
                    if (q[j].x < cfg.min_border_dist || q[j].y < cfg.min_border_dist ||
                        q[j].x > cfg.cols - 1 - cfg.min_border_dist || q[j].y > cfg.rows - 1 - cfg.min_border_dist)
                        ok = false;
            }
            if (ok && lane == 0) {
                unsigned k = atomicAdd(&n_cand[f], 1u);
                if (k < (unsigned)kCandMax) {
                    CandRec c;
                    for (int j = 0; j < 4; j++) { c.x[j] = (short)q[j].x; c.y[j] = (short)q[j].y; }
                    c.n = rec.n;
                    c.ordkey = rec.scale * (1u << 22) + ((1u << 22) - 1u - rec.key);
                    cands[(size_t)f * kCandMax + k] = c;
                } else {
                    atomicOr(&ctr->overflow, (unsigned)kOvfCands);
                }
            }
            QST(7);
        }
        }
    }
#ifdef ASLAM_QUADS_STAMPS
    if (lane == 0 && blockIdx.x < 3 && nframes >= 32)
        printf("quads wave %d: %d contours; cycles: initial loops %lld, reductions %lld, slice overhead %lld, slice loops %lld, clean-up %lld, ticket %lld, record %lld, tests+emit %lld\n",
               (int)blockIdx.x, qn, qst[0], qst[1], qst[2], qst[3], qst[4], qst[5], qst[6], qst[7]);
#endif
}

// ------------------------------------------------------------------------------------------------
// k_assemble : one workgroup per frame
// ------------------------------------------------------------------------------------------------
constexpr int kPairMax = 4096;     // near pairs per frame (_filterTooCloseCandidates)
constexpr int AT = 1024;           // threads of k_assemble

__global__ __launch_bounds__(1024) void k_assemble(DetectCfg cfg, Counters* ctr, const CandRec* __restrict__ cands,
                                                   const unsigned* __restrict__ n_cand, FinalCand* __restrict__ finals,
                                                   unsigned* __restrict__ n_final, IdentWork* __restrict__ work) {
    __shared__ CandRec sIn[kCandMax];
    __shared__ CandRec sC[kCandMax];
    __shared__ unsigned sPair[kPairMax];           // (i << 16) | j, unordered
    __shared__ unsigned sPairSorted[kPairMax];     // lexicographic (i, j) order
    __shared__ unsigned char removed[kCandMax];
    __shared__ int outPos[kCandMax];
    __shared__ unsigned sNPair, sWorkBase;
    const int tid = threadIdx.x;
    const int f = blockIdx.x;
    const int C = (int)min(n_cand[f], (unsigned)kCandMax);

    for (int i = tid; i < C; i += AT) sIn[i] = cands[(size_t)f * kCandMax + i];
    for (int i = tid; i < kCandMax; i += AT) removed[i] = 0;
    if (tid == 0) sNPair = 0;
    __syncthreads();
    // rank sort by ordkey (unique inside a frame): OpenCV order = scale ascending, reverse discovery
    for (int i = tid; i < C; i += AT) {
        unsigned k = sIn[i].ordkey;
        int rank = 0;
        for (int j = 0; j < C; j++) rank += sIn[j].ordkey < k;
        CandRec c = sIn[i];
        // _reorderCandidatesCorners
        double dx1 = (double)c.x[1] - c.x[0], dy1 = (double)c.y[1] - c.y[0];
        double dx2 = (double)c.x[2] - c.x[0], dy2 = (double)c.y[2] - c.y[0];
        double cross = (dx1 * dy2) - (dy1 * dx2);
        if (cross < 0.0) {
            short tx = c.x[1], ty = c.y[1];
            c.x[1] = c.x[3]; c.y[1] = c.y[3];
            c.x[3] = tx; c.y[3] = ty;
        }
        sC[rank] = c;
    }
    __syncthreads();
    // _filterTooCloseCandidates, part 1: near pairs (i < j).  A cheap exact pre-test skips far pairs: the mean
    // squared corner distance of any cyclic matching is >= the squared distance between the corner centroids.
    // every unordered pair exactly once: candidate a against the next (C - 1) / 2 candidates in cyclic order (and, for even C, the
    // first half against the one opposite); the quotient p / half by a float reciprocal, corrected to the exact value
    const int half = (C - 1) / 2, extra = (C & 1) == 0 ? C / 2 : 0;
    const float inv_half = half > 0 ? 1.0f / (float)half : 0.f;
    for (int p = tid; p < C * half + extra; p += AT) {
        int a, k;
        if (p < C * half) {
            a = (int)(((float)p + 0.5f) * inv_half);
            if (a * half > p) a--;
            if ((a + 1) * half <= p) a++;
            k = p - a * half + 1;                                   // 1 .. half
        } else {
            a = p - C * half;                                       // 0 .. C / 2 - 1
            k = C / 2;
        }
        int b2 = a + k;
        if (b2 >= C) b2 -= C;
        const int i = min(a, b2), j = max(a, b2);
        {
            const CandRec& ca = sC[i];
            const CandRec& bq = sC[j];
            int minimumPerimeter = (int)min(ca.n, bq.n);
            double minMarkerDistancePixels = (double)minimumPerimeter * cfg.min_marker_dist_rate;
            double thr = minMarkerDistancePixels * minMarkerDistancePixels;
            int sax = ca.x[0] + ca.x[1] + ca.x[2] + ca.x[3], say = ca.y[0] + ca.y[1] + ca.y[2] + ca.y[3];
            int sbx = bq.x[0] + bq.x[1] + bq.x[2] + bq.x[3], sby = bq.y[0] + bq.y[1] + bq.y[2] + bq.y[3];
            double cdx = (double)(sax - sbx) * 0.25, cdy = (double)(say - sby) * 0.25;
            bool near = false;
            if (cdx * cdx + cdy * cdy < thr + 1.0) {
                for (int fc = 0; fc < 4 && !near; fc++) {
                    double distSq = 0;
                    for (int c = 0; c < 4; c++) {
                        int modC = (c + fc) & 3;
                        double ddx = (double)(ca.x[modC] - bq.x[c]), ddy = (double)(ca.y[modC] - bq.y[c]);
                        distSq += ddx * ddx + ddy * ddy;
                    }
                    distSq /= 4.;
                    if (distSq < thr) near = true;
                }
            }
            if (near) {
                unsigned k = atomicAdd(&sNPair, 1u);
                if (k < (unsigned)kPairMax) sPair[k] = ((unsigned)i << 16) | (unsigned)j;
                else atomicOr(&ctr->overflow, (unsigned)kOvfCands);
            }
        }
    }
    __syncthreads();
    const int P = (int)min(sNPair, (unsigned)kPairMax);
    for (int i = tid; i < P; i += AT) {
        unsigned k = sPair[i];
        int rank = 0;
        for (int j = 0; j < P; j++) rank += sPair[j] < k;
        sPairSorted[rank] = k;
    }
    __syncthreads();
    // part 2: sequential marking in pair order (depends on earlier removals).  Which of the two a pair would remove does not depend
    // on the marks: all threads work that out first, so that the one marking thread has a single LDS word to fetch per pair (its
    // address does not depend on the marks either: the fetches run ahead of the marking)
    for (int q = tid; q < P; q += AT) {
        const unsigned k = sPairSorted[q];
        const unsigned i = k >> 16, j = k & 0xFFFFu;
        sPair[q] = k | (sC[i].n > sC[j].n ? 0x80000000u : 0u);      // top bit: the pair removes j (candidate indices are below 2^11)
    }
    __syncthreads();
    if (tid == 0) {
        for (int q = 0; q < P; q++) {
            const unsigned k = sPair[q];
            const int i = (int)((k >> 16) & 0x7FFFu), j = (int)(k & 0xFFFFu);
            if (removed[i] || removed[j]) continue;
            removed[(k >> 31) ? j : i] = 1;
        }
    }
    __syncthreads();
    // positions of the survivors: an ordered compaction by all threads (ballot ranks, wave offsets through LDS)
    {
        __shared__ int sWaveCnt[AT / 64];
        int kbase = 0;
        for (int i0 = 0; i0 < C; i0 += AT) {                        // uniform
            const int i = i0 + tid;
            const bool keep = i < C && !removed[i];
            const unsigned long long m = __ballot(keep);
            if ((tid & 63) == 0) sWaveCnt[tid >> 6] = __popcll(m);
            __syncthreads();
            int before = 0, total = 0;
            for (int w = 0; w < AT / 64; w++) { const int c = sWaveCnt[w]; if (w < (tid >> 6)) before += c; total += c; }
            if (i < C) outPos[i] = keep ? kbase + before + __popcll(m & ((1ull << (tid & 63)) - 1ull)) : -1;
            kbase += total;
            __syncthreads();
        }
        if (tid == 0) {
            n_final[f] = (unsigned)kbase;
            sWorkBase = kbase > 0 ? atomicAdd(&ctr->n_ident, (unsigned)kbase) : 0u;
        }
    }
    __syncthreads();
    for (int i = tid; i < C; i += AT) {
        int k = outPos[i];
        if (k >= 0) {
            FinalCand fc;
            for (int j = 0; j < 4; j++) { fc.c[2 * j] = (float)sC[i].x[j]; fc.c[2 * j + 1] = (float)sC[i].y[j]; }
            fc.n = (int)sC[i].n;
            fc.id = -1;
            fc.pad[0] = fc.pad[1] = 0;
            finals[(size_t)f * kCandMax + k] = fc;
            work[sWorkBase + k] = IdentWork{(unsigned)f, (unsigned)k};
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_identify : one wavefront per candidate
// ------------------------------------------------------------------------------------------------
constexpr int kWarpMax = kDictMaxCells * kCellPx;   // 72

constexpr int kWarpUnroll = 25;        // pixels of the warped marker image per lane and round (a 7 x 7-cell marker at 8 px per cell: 49 per lane, two rounds)
__global__ __launch_bounds__(64) void k_identify(DetectCfg cfg, Counters* ctr, const uint8_t* __restrict__ gray,
                                                 FinalCand* __restrict__ finals, const IdentWork* __restrict__ work,
                                                 const unsigned long long* __restrict__ dict_codes) {
    __shared__ double sA[8][8];
    __shared__ double sB[8];
    __shared__ double sM[9];
    __shared__ uint8_t img[kWarpMax * kWarpMax];
    __shared__ int hist[256];
    __shared__ int sDecision[2];        // [0]: 0 = otsu, 1 = all zero bits, 2 = all one bits ; [1]: otsu threshold
    __shared__ double sOtsuA[256], sOtsuB[256], sMu;   // per bin: p_i, i p_i; then q1 (-1: skipped), mu1
    __shared__ int sOtsuRange[2];
    const int lane = threadIdx.x & 63;
    const int rows = cfg.rows, cols = cfg.cols;
    const int ms = cfg.marker_size, bb = cfg.border_bits;
    const int nc = ms + 2 * bb;                 // cells per side
    const int cell = cfg.cell_px;
    const int S = nc * cell;                    // warped image side
    const unsigned n_work = ctr->n_ident;
#ifdef ASLAM_IDENT_STAMPS
    long long ist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long ilast = clock64();
    int inum = 0;
#define IST(i) do { const long long t_ = clock64(); ist[i] += t_ - ilast; ilast = t_; } while (0)
#else
#define IST(i) do { } while (0)
#endif

    for (;;) {
        unsigned wi = 0;
        IST(7);
        if (lane == 0) wi = atomicAdd(&ctr->q_ident, 1u);
        wi = __shfl(wi, 0);
        if (wi >= n_work) break;
        const IdentWork wk = work[wi];
        FinalCand* fc = &finals[(size_t)wk.frame * kCandMax + wk.idx];
        const uint8_t* gimg = gray + (size_t)wk.frame * rows * cols;

        for (int i = lane; i < 256; i += 64) hist[i] = 0;
        IST(0);
        {
            // cv::getPerspectiveTransform(corners -> (0,0),(S-1,0),(S-1,S-1),(0,S-1)): the 8 x 8 elimination with partial pivoting, one
            // matrix element per lane (row er = lane / 8, column ec = lane % 8; every lane of a row carries the row's right-hand side).
            // Every element goes through exactly the operations of the sequential elimination (quotient, product, difference; first
            // maximal pivot), so the result is bit-identical to it - only the 58 k cycles of dependent LDS traffic on one lane are gone.
            const int er = lane >> 3, ec = lane & 7, ei = er & 3;
            const float dstx = (ei == 1 || ei == 2) ? (float)S - 1 : 0.f;
            const float dsty = ei >= 2 ? (float)S - 1 : 0.f;
            const float sx = fc->c[2 * ei], sy = fc->c[2 * ei + 1];
            const float dst = er < 4 ? dstx : dsty;
            double ea;
            if (ec == 6) ea = -(double)sx * dst;
            else if (ec == 7) ea = -(double)sy * dst;
            else {
                const int k = er < 4 ? ec : ec - 3;                  // rows 0..3: (sx, sy, 1) in columns 0..2; rows 4..7: in columns 3..5
                ea = k == 0 ? (double)sx : k == 1 ? (double)sy : k == 2 ? 1.0 : 0.0;
            }
            double eb = dst;
            // rows are exchanged through LDS (the workgroup is one wavefront): per column one write of the matrix, then broadcast reads of
            // the pivot column, the pivot row and the right-hand side - a quarter of the LDS-pipe operations of lane-to-lane shuffles
            for (int col = 0; col < 8; col++) {
                __syncthreads();                                     // the previous column's reads are done
                sA[er][ec] = ea;
                if (ec == 0) sB[er] = eb;
                __syncthreads();
                int piv = col;
                double best = fabs(sA[col][col]);
                for (int r = col + 1; r < 8; r++) {
                    const double v = fabs(sA[r][col]);
                    if (v > best) { best = v; piv = r; }
                }
                // after the exchange row `col` holds what row `piv` held, and the other way round
                if (er == col) { ea = sA[piv][ec]; eb = sB[piv]; }
                else if (er == piv) { ea = sA[col][ec]; eb = sB[col]; }
                const double pv = sA[piv][col];
                const double mine = er == col ? pv : er == piv ? sA[col][col] : sA[er][col];
                const double rowc = sA[piv][ec], brow = sB[piv];
                if (er > col) {
                    const double fct = mine / pv;
                    if (ec >= col) ea -= fct * rowc;
                    eb -= fct * brow;
                }
            }
            __syncthreads();
            sA[er][ec] = ea;
            if (ec == 0) sB[er] = eb;
        }
        __syncthreads();
        IST(1);
        if (lane == 0) {
            double x[8];
            for (int i = 7; i >= 0; i--) {
                double s = sB[i];
                for (int c = i + 1; c < 8; c++) s -= sA[i][c] * x[c];
                x[i] = s / sA[i][i];
            }
            // cv::invert (3x3 cofactor form) for warpPerspective without WARP_INVERSE_MAP
            double m0 = x[0], m1 = x[1], m2 = x[2], m3 = x[3], m4 = x[4], m5 = x[5], m6 = x[6], m7 = x[7], m8 = 1.0;
            double det = m0 * (m4 * m8 - m5 * m7) - m1 * (m3 * m8 - m5 * m6) + m2 * (m3 * m7 - m4 * m6);
            if (det != 0.) {
                double d = 1. / det;
                sM[0] = (m4 * m8 - m5 * m7) * d;
                sM[1] = (m2 * m7 - m1 * m8) * d;
                sM[2] = (m1 * m5 - m2 * m4) * d;
                sM[3] = (m5 * m6 - m3 * m8) * d;
                sM[4] = (m0 * m8 - m2 * m6) * d;
                sM[5] = (m2 * m3 - m0 * m5) * d;
                sM[6] = (m3 * m7 - m4 * m6) * d;
                sM[7] = (m1 * m6 - m0 * m7) * d;
                sM[8] = (m0 * m4 - m1 * m3) * d;
            } else {
                for (int i = 0; i < 9; i++) sM[i] = 0;
            }
        }
        __syncthreads();

        IST(2);
        // warpPerspective(INTER_NEAREST), histogram, inner-region moments
        const int lo = cell / 2, hi = S - cell / 2;
        long long sum = 0, sq = 0;
        // (kWarpUnroll pixels per round: their gray loads are issued together, at clamped addresses - a guarded load is a branch with
        //  its own wait, and the loads of a lane's ~50 pixels would queue up behind each other)
        // pixel p = lane + 64 k of the warped image; (x, y) advanced without divisions, once for the addresses and once for the use
        const int step_y = 64 / S, step_x = 64 - step_y * S;
        int py = lane / S, px = lane - py * S;
        int qy = py, qx = px;
        for (int p0 = lane; p0 < S * S; p0 += 64 * kWarpUnroll) {
            int vv[kWarpUnroll];                                    // gray value, -1 outside the frame
#pragma unroll
            for (int u = 0; u < kWarpUnroll; u++) {
                const int x = px, y = min(py, S - 1);                // (beyond the image: any valid pixel, the value is not used)
                px += step_x; py += step_y;
                if (px >= S) { px -= S; py++; }
                double X0 = sM[1] * y + sM[2], Y0 = sM[4] * y + sM[5], W0 = sM[7] * y + sM[8];
                double W = W0 + sM[6] * x;
                W = W ? 1. / W : 0;
                double fX = fmax((double)INT_MIN, fmin((double)INT_MAX, (X0 + sM[0] * x) * W));
                double fY = fmax((double)INT_MIN, fmin((double)INT_MAX, (Y0 + sM[3] * x) * W));
                const int X = (int)rint(fX), Y = (int)rint(fY);      // clamped to the int range above: the 32-bit conversion is exact
                const bool inside = X >= 0 && X < cols && Y >= 0 && Y < rows;
                const int g = gimg[inside ? (size_t)Y * cols + X : (size_t)0];
                vv[u] = inside ? g : -1;
            }
#pragma unroll
            for (int u = 0; u < kWarpUnroll; u++) {
                const int p = p0 + 64 * u;
                const int x = qx, y = qy;
                qx += step_x; qy += step_y;
                if (qx >= S) { qx -= S; qy++; }
                if (p < S * S) {
                    const int v = max(vv[u], 0);
                    img[p] = (uint8_t)v;
                    atomicAdd(&hist[v], 1);
                    if (x >= lo && x < hi && y >= lo && y < hi) { sum += v; sq += v * v; }
                }
            }
        }
        IST(3);
        for (int o = 32; o > 0; o >>= 1) { sum += __shfl_down(sum, o); sq += __shfl_down(sq, o); }
        __syncthreads();
        // getThreshVal_Otsu_8u over the whole warped image.  Only (q1, mu1) are carried from bin to bin: one lane runs that recurrence
        // - every operation of the sequential loop, bins before the first and after the last occupied one leave nothing behind - and all
        // lanes then evaluate sigma for their bins from the stored (q1, mu1); first maximum as in the scan.
        {
            const int N = S * S;
            const double sc = 1. / N;
            long long isum = 0;                                     // sum of i * hist[i]: integers, exact in any order
            unsigned long long occupied[4];
            for (int k = 0; k < 4; k++) {
                const int i = lane + 64 * k;
                const int h = hist[i];
                const double p_i = h * sc;
                sOtsuA[i] = p_i;
                sOtsuB[i] = i * p_i;
                isum += (long long)i * h;
                occupied[k] = __ballot(h != 0);
            }
            for (int o = 32; o > 0; o >>= 1) isum += __shfl_down(isum, o);
            __syncthreads();
            if (lane == 0) {
                const double scale = 1.0 / ((double)(hi - lo) * (hi - lo));
                const double mean = sum * scale;
                const double var = fmax(sq * scale - mean * mean, 0.);
                const double stddev = sqrt(var);
                if (stddev < cfg.min_otsu_std) {
                    sDecision[0] = mean > 127 ? 2 : 1;
                    sDecision[1] = 0;
                } else {
                    sDecision[0] = 0;
                    int first = 256, last = -1;
                    for (int k = 0; k < 4; k++)
                        if (occupied[k]) { first = min(first, 64 * k + __ffsll((long long)occupied[k]) - 1); last = 64 * k + 63 - __clzll((long long)occupied[k]); }
                    sOtsuRange[0] = first; sOtsuRange[1] = last;
                    sMu = (double)isum * sc;
                    double mu1 = 0, q1 = 0;
                    for (int i = first; i <= last; i++) {
                        const double p_i = sOtsuA[i], ip_i = sOtsuB[i];
                        mu1 *= q1;
                        q1 += p_i;
                        const double q2 = 1. - q1;
                        if (fmin(q1, q2) < FLT_EPSILON || fmax(q1, q2) > 1. - FLT_EPSILON) {
                            sOtsuA[i] = -1.;                         // no sigma for this bin
                        } else {
                            mu1 = (mu1 + ip_i) / q1;
                            sOtsuA[i] = q1;
                            sOtsuB[i] = mu1;
                        }
                    }
                }
            }
            __syncthreads();
            if (sDecision[0] == 0) {                                 // uniform
                const int first = sOtsuRange[0], last = sOtsuRange[1];
                const double mu = sMu;
                double best = 0.0;
                int besti = 0x7fffffff;
                for (int k = 0; k < 4; k++) {
                    const int i = lane + 64 * k;
                    if (i >= first && i <= last) {
                        const double q1 = sOtsuA[i], mu1 = sOtsuB[i];
                        if (q1 >= 0.) {
                            const double q2 = 1. - q1;
                            const double mu2 = (mu - q1 * mu1) / q2;
                            const double sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
                            if (sigma > best) { best = sigma; besti = i; }
                        }
                    }
                }
                wave_first_max(best, besti);
                if (lane == 0) sDecision[1] = best > 0.0 ? besti : 0;
            }
        }
        __syncthreads();
        IST(4);
        // cell votes: up to 81 cells, lanes take cells lane and lane + 64
        unsigned long long bitsLo = 0, bitsHi = 0;       // cell index c -> bit c (lo) / c - 64 (hi)
        {
            const int dec = sDecision[0], T = sDecision[1];
            const int wcell = cell - 2 * cfg.cell_margin;
            for (int half = 0; half < 2; half++) {
                int c = lane + 64 * half;
                int bit = 0;
                if (c < nc * nc) {
                    if (dec == 2) bit = 1;
                    else if (dec == 0) {
                        int cy = c / nc, cx = c - cy * nc;
                        int Xs = cx * cell + cfg.cell_margin, Ys = cy * cell + cfg.cell_margin;
                        int nz = 0;
                        for (int yy = 0; yy < wcell; yy++)
                            for (int xx = 0; xx < wcell; xx++) nz += img[(Ys + yy) * S + Xs + xx] > T;
                        bit = nz > (wcell * wcell) / 2;
                    }
                }
                unsigned long long bm = __ballot(bit);
                if (half == 0) bitsLo = bm; else bitsHi = bm;
            }
        }
        auto cell = [&](int cy, int cx) -> int {
            int c = cy * nc + cx;
            return c < 64 ? (int)((bitsLo >> c) & 1ull) : (int)((bitsHi >> (c - 64)) & 1ull);
        };
        IST(5);
        // _getBorderErrors
        int borderErr = 0;
        for (int y = 0; y < nc; y++)
            for (int k = 0; k < bb; k++) { borderErr += cell(y, k); borderErr += cell(y, nc - 1 - k); }
        for (int x = bb; x < nc - bb; x++)
            for (int k = 0; k < bb; k++) { borderErr += cell(k, x); borderErr += cell(nc - 1 - k, x); }
        int id = -1, rot = 0;
        if (borderErr <= cfg.max_border_err) {
            // inner bits, row-major MSB first (rotation 0 of Dictionary::getByteListFromBits)
            unsigned long long code = 0;
            for (int r = 0; r < ms; r++)
                for (int c = 0; c < ms; c++) code = (code << 1) | (unsigned long long)cell(r + bb, c + bb);
            // Dictionary::identify: first marker whose best rotation is within the correction budget
            int bestM = INT_MAX, bestR = 0;
            for (int m = lane; m < cfg.n_dict; m += 64) {
                int minD = ms * ms + 1, minR = -1;
                for (int r = 0; r < 4; r++) {
                    int h = __popcll(dict_codes[(size_t)m * 4 + r] ^ code);
                    if (h < minD) { minD = h; minR = r; }
                }
                if (minD <= cfg.max_corr && m < bestM) { bestM = m; bestR = minR; }
            }
            for (int o = 32; o > 0; o >>= 1) {
                int om = __shfl_down(bestM, o), orr = __shfl_down(bestR, o);
                if (om < bestM) { bestM = om; bestR = orr; }
            }
            bestM = __shfl(bestM, 0);
            bestR = __shfl(bestR, 0);
            if (bestM != INT_MAX) { id = bestM; rot = bestR; }
        } else {
            // keep the wave convergent: the shuffles above are executed by all lanes or by none
        }
        if (lane == 0) {
            fc->pad[0] = rot;        // corner rotation is applied when the marker list is built (k_pose)
            fc->id = id;
        }
        __syncthreads();
        IST(6);
#ifdef ASLAM_IDENT_STAMPS
        inum++;
#endif
    }
#ifdef ASLAM_IDENT_STAMPS
    if (lane == 0 && blockIdx.x < 3 && n_work > 1000)
        printf("identify wave %d: %d candidates; cycles: fetch %lld, elimination %lld, back-substitution %lld, warp %lld, moments+otsu %lld, votes %lld, border+dictionary %lld, ticket %lld\n",
               (int)blockIdx.x, inum, ist[0], ist[1], ist[2], ist[3], ist[4], ist[5], ist[6], ist[7]);
#endif
}

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
void launch_threshold(hipStream_t st, const uint8_t* in, int channels, size_t frame_stride, size_t row_step, int nframes,
                      uint8_t* gray, uint8_t* nbr, const DetectCfg& cfg, unsigned* starts, unsigned* n_starts, unsigned* nodeplane,
                      Counters* ctr) {
    const unsigned total = (unsigned)((cfg.cols + TW - 1) / TW) * (unsigned)((cfg.rows + TH - 1) / TH) * (unsigned)nframes;
    const bool def = cfg.n_scales == 3 && cfg.win_r[0] == 1 && cfg.win_r[1] == 6 && cfg.win_r[2] == 11;
    if (def)
        hipLaunchKernelGGL(k_threshold<true>, dim3((total + 7u) / 8u * 8u), dim3(256), 0, st, in, channels, frame_stride, row_step, gray, nbr, cfg,
                           starts, n_starts, nodeplane, ctr, nframes);
    else
        hipLaunchKernelGGL(k_threshold<false>, dim3((total + 7u) / 8u * 8u), dim3(256), 0, st, in, channels, frame_stride, row_step, gray, nbr, cfg,
                           starts, n_starts, nodeplane, ctr, nframes);
}
void launch_clear_counts(hipStream_t st, int nframes, Counters* ctr, unsigned* n_starts, unsigned* n_contours, unsigned* n_points, unsigned* n_write,
                         unsigned* n_cand) {
    hipLaunchKernelGGL(k_clear_counts, dim3((std::max(nframes, kCounterHeads) + 255) / 256), dim3(256), 0, st, nframes, ctr, n_starts, n_contours, n_points,
                       n_write, n_cand);
}
void launch_prefix(hipStream_t st, int nframes, const unsigned* counts, unsigned cap, unsigned per_ticket, unsigned* pre) {
    if (nframes == 1 && per_ticket == 1u) return;              // the consumers take a single frame's range from its count (load_ticket_ranges)
    hipLaunchKernelGGL(k_prefix, dim3(1), dim3(256), 0, st, nframes, counts, cap, per_ticket, pre);
}
void launch_seg(hipStream_t st, int nwaves, const uint8_t* nbr, const DetectCfg& cfg, int nframes, const unsigned* starts,
                const unsigned* n_starts, const unsigned* nodeplane, const unsigned* pre, Counters* ctr, NodeRec* nodes) {
    hipLaunchKernelGGL(k_seg, dim3(nwaves), dim3(64), 0, st, nbr, cfg, nframes, starts, n_starts, nodeplane, pre, ctr, nodes);
}
void launch_link(hipStream_t st, const DetectCfg& cfg, int nframes, const unsigned* n_starts, Counters* ctr, const NodeRec* nodes,
                 unsigned* link_todo, ContourRec* contours, unsigned* n_contours, unsigned* n_points, WriteRec* wlist, unsigned* n_write) {
    // ASLAM_LINK_LDS_NODES: nodes of a frame the LDS image holds (a test knob: 0 sends every frame through k_link_serial)
    const char* env = std::getenv("ASLAM_LINK_LDS_NODES");
    const unsigned lds_nodes = env ? std::min((unsigned)std::atoi(env), kLinkLdsNodes) : kLinkLdsNodes;
    const size_t dyn = (size_t)kLinkLdsNodes * 4u + sizeof(LinkSlot) * kLinkSlots + (kLinkLdsNodes + 31u) / 32u * 4u;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_link), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        attr_done = true;
    }
    hipLaunchKernelGGL(k_link, dim3(nframes), dim3(kLinkThreads), dyn, st, cfg, n_starts, ctr, nodes, lds_nodes, link_todo, contours, n_contours,
                       n_points, wlist, n_write);
    hipLaunchKernelGGL(k_link_serial, dim3(nframes), dim3(512), 0, st, cfg, n_starts, ctr, nodes, link_todo, contours, n_contours, n_points, wlist, n_write);
}
void launch_trace_write(hipStream_t st, int nblocks, const uint8_t* nbr, const DetectCfg& cfg, int nframes, const unsigned* pre,
                        Counters* ctr, const ContourRec* contours, const WriteRec* wlist, const unsigned* n_write, unsigned* points) {
    hipLaunchKernelGGL(k_trace_write, dim3(nblocks), dim3(256), 0, st, nbr, cfg, nframes, pre, ctr, contours, wlist, n_write, points);
}
void launch_quads(hipStream_t st, int nwaves, const DetectCfg& cfg, int nframes, Counters* ctr, const ContourRec* contours,
                  const unsigned* n_contours, const unsigned* pre, const unsigned* points, CandRec* cands, unsigned* n_cand) {
    hipLaunchKernelGGL(k_quads, dim3(nwaves), dim3(64), 0, st, cfg, nframes, ctr, contours, n_contours, pre, points, cands, n_cand);
}
void launch_assemble(hipStream_t st, int nframes, const DetectCfg& cfg, Counters* ctr, const CandRec* cands,
                     const unsigned* n_cand, FinalCand* finals, unsigned* n_final, IdentWork* work) {
    hipLaunchKernelGGL(k_assemble, dim3(nframes), dim3(AT), 0, st, cfg, ctr, cands, n_cand, finals, n_final, work);
}
void launch_identify(hipStream_t st, int nwaves, const DetectCfg& cfg, Counters* ctr, const uint8_t* gray,
                     FinalCand* finals, const IdentWork* work, const unsigned long long* dict_codes) {
    hipLaunchKernelGGL(k_identify, dim3(nwaves), dim3(64), 0, st, cfg, ctr, gray, finals, work, dict_codes);
}
int max_frames_per_call() { return kMaxFramesPerCall; }

} // namespace aslam
