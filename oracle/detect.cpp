// ORACLE (test infrastructure only — see oracle.h).  Sequential CPU restatement of
// cv::aruco::detectMarkers as called by the reference at src/aruco_slam.cpp:313 with the library
// default DetectorParameters.  The algorithm lives in third-party OpenCV 3.2.0 (+contrib aruco),
// which is not vendored by the reference and absent from this image; each function below names
// the OpenCV routine whose published behaviour it restates.
#include "oracle.h"
#include <cmath>
#include <cstring>
#include <algorithm>
#include <cfloat>
#include <climits>

namespace oracle {

// cv::cvtColor(BGR2GRAY), 8-bit fixed point: yuv_shift = 14, B2Y=1868, G2Y=9617, R2Y=4899.
// (aruco.cpp::_convertToGrey; call path aruco_slam_node.cpp:93 -> aruco_slam.cpp:313)
void bgr_to_gray(const uint8_t* bgr, int rows, int cols, size_t step, uint8_t* gray) {
    for (int y = 0; y < rows; y++) {
        const uint8_t* s = bgr + (size_t)y * step;
        for (int x = 0; x < cols; x++)
            gray[(size_t)y * cols + x] =
                (uint8_t)((s[3 * x] * 1868 + s[3 * x + 1] * 9617 + s[3 * x + 2] * 4899 + (1 << 13)) >> 14);
    }
}

// cv::boxFilter(8u->8u, normalize=true, BORDER_REPLICATE) for odd k.  The mean is
// saturate_cast<uchar>(sum * (1/k^2)) with round-to-nearest; because k^2 is odd a tie can never occur,
// so the rounding is expressed in exact integer arithmetic.
void box_mean_u8(const uint8_t* gray, int rows, int cols, int k, uint8_t* mean) {
    const int r = k / 2, k2 = k * k;
    std::vector<int> hs((size_t)rows * cols);
    for (int y = 0; y < rows; y++) {
        const uint8_t* g = gray + (size_t)y * cols;
        int s = 0;
        for (int d = -r; d <= r; d++) s += g[std::min(std::max(d, 0), cols - 1)];
        hs[(size_t)y * cols] = s;
        for (int x = 1; x < cols; x++) {
            s += g[std::min(x + r, cols - 1)] - g[std::max(x - r - 1, 0)];
            hs[(size_t)y * cols + x] = s;
        }
    }
    std::vector<int> col(cols);
    for (int x = 0; x < cols; x++) {
        int s = 0;
        for (int d = -r; d <= r; d++) s += hs[(size_t)std::min(std::max(d, 0), rows - 1) * cols + x];
        col[x] = s;
    }
    for (int y = 0; y < rows; y++) {
        if (y > 0)
            for (int x = 0; x < cols; x++)
                col[x] += hs[(size_t)std::min(y + r, rows - 1) * cols + x] - hs[(size_t)std::max(y - r - 1, 0) * cols + x];
        for (int x = 0; x < cols; x++) {
            int m = (2 * col[x] + k2) / (2 * k2);
            mean[(size_t)y * cols + x] = (uint8_t)std::min(m, 255);
        }
    }
}

// cv::adaptiveThreshold(ADAPTIVE_THRESH_MEAN_C, THRESH_BINARY_INV, k, C): dst = (src - mean <= -floor(C)) ? 255 : 0
// (aruco.cpp::_threshold)
void adaptive_threshold_mean_inv(const uint8_t* gray, int rows, int cols, int k, double C, uint8_t* out) {
    std::vector<uint8_t> mean((size_t)rows * cols);
    box_mean_u8(gray, rows, cols, k, mean.data());
    const int idelta = (int)std::floor(C);
    for (size_t i = 0; i < (size_t)rows * cols; i++)
        out[i] = ((int)gray[i] - (int)mean[i] <= -idelta) ? 255 : 0;
}

// ---------------------------------------------------------------------------------------------
// cv::findContours(RETR_LIST, CHAIN_APPROX_NONE): Suzuki-Abe border following
// (imgproc/contours.cpp: cvStartFindContours / cvFindNextContour / icvFetchContour), OpenCV 3.2:
// the source is copied into a 1-px zero frame, marks are 2 / -126 (nbd | -128), direction codes
// 0=E 1=NE 2=N 3=NW 4=W 5=SW 6=S 7=SE, contours returned in reverse order of discovery.
static const int kDx[8] = {1, 1, 0, -1, -1, -1, 0, 1};
static const int kDy[8] = {0, -1, -1, -1, 0, 1, 1, 1};

static void fetch_contour(signed char* ptr, int step, Pt pt, bool is_hole, std::vector<Pt>& out) {
    const signed char nbd = 2;
    int deltas[16] = {1, -step + 1, -step, -step - 1, -1, step - 1, step, step + 1,
                      1, -step + 1, -step, -step - 1, -1, step - 1, step, step + 1};
    signed char *i0 = ptr, *i1, *i3, *i4 = nullptr;
    int s, s_end;
    s_end = s = is_hole ? 0 : 4;
    do {
        s = (s - 1) & 7;
        i1 = i0 + deltas[s];
        if (*i1 != 0) break;
    } while (s != s_end);

    if (s == s_end) {               // single pixel domain
        *i0 = (signed char)(nbd | -128);
        out.push_back(pt);
        return;
    }
    i3 = i0;
    for (;;) {
        s_end = s;
        for (;;) {
            i4 = i3 + deltas[++s];
            if (*i4 != 0) break;
        }
        s &= 7;
        if ((unsigned)(s - 1) < (unsigned)s_end)
            *i3 = (signed char)(nbd | -128);       // "right" bound
        else if (*i3 == 1)
            *i3 = nbd;
        out.push_back(pt);
        pt.x += kDx[s];
        pt.y += kDy[s];
        if (i4 == i0 && i3 == i1) break;
        i3 = i4;
        s = (s + 4) & 7;
    }
}

void find_contours_list_none(const uint8_t* bin, int rows, int cols, std::vector<Contour>& out) {
    out.clear();
    const int W = cols + 2, H = rows + 2;
    std::vector<signed char> img((size_t)W * H, 0);
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++)
            img[(size_t)(y + 1) * W + x + 1] = bin[(size_t)y * cols + x] ? 1 : 0;

    std::vector<Contour> found;
    for (int y = 1; y < H - 1; y++) {
        signed char* row = &img[(size_t)y * W];
        int prev = 0;
        for (int x = 1; x < W - 1; x++) {
            int p = row[x];
            if (p == prev) continue;
            bool is_hole = false;
            if (!(prev == 0 && p == 1)) {
                if (p != 0 || prev < 1) { prev = p; continue; }
                is_hole = true;
            }
            int ox = x - (is_hole ? 1 : 0);
            found.emplace_back();
            Contour& c = found.back();
            c.is_hole = is_hole ? 1 : 0;
            c.key = (y - 1) * cols + (x - 1);
            fetch_contour(&row[ox], W, Pt{ox - 1, y - 1}, is_hole, c.pts);
            p = row[x];
            prev = p;
        }
    }
    // cvInsertNodeIntoTree prepends each new contour: the list comes back in reverse discovery order
    out.assign(found.rbegin(), found.rend());
}

// cv::approxPolyDP for a closed integer contour (imgproc/approx.cpp::approxPolyDP_<int>).
void approx_poly_dp_closed(const std::vector<Pt>& src, double eps, std::vector<Pt>& dstv) {
    struct Range { int start, end; };
    dstv.clear();
    int count = (int)src.size();
    if (count == 0) return;
    std::vector<Pt> dst((size_t)count);
    std::vector<Range> stack;
    const int init_iters = 3;
    Range slice{0, 0}, right_slice{0, 0};
    Pt start_pt{-1000000, -1000000}, end_pt{0, 0}, pt{0, 0};
    int i = 0, j, pos = 0, wpos, new_count = 0;
    bool le_eps = false;

    auto READ_PT = [&](Pt& p, int& ps) { p = src[ps]; if (++ps >= count) ps = 0; };
    auto READ_DST_PT = [&](Pt& p, int& ps) { p = dst[ps]; if (++ps >= count) ps = 0; };

    eps *= eps;

    // 1. approximately two farthest points of the contour
    right_slice.start = 0;
    for (i = 0; i < init_iters; i++) {
        double dist, max_dist = 0;
        pos = (pos + right_slice.start) % count;
        READ_PT(start_pt, pos);
        for (j = 1; j < count; j++) {
            double dx, dy;
            READ_PT(pt, pos);
            dx = pt.x - start_pt.x;
            dy = pt.y - start_pt.y;
            dist = dx * dx + dy * dy;
            if (dist > max_dist) { max_dist = dist; right_slice.start = j; }
        }
        le_eps = max_dist <= eps;
    }
    // 2. initialise the stack
    if (!le_eps) {
        right_slice.end = slice.start = pos % count;
        slice.end = right_slice.start = (right_slice.start + slice.start) % count;
        stack.push_back(right_slice);
        stack.push_back(slice);
    } else {
        dst[new_count++] = start_pt;
    }
    // 3. recursive process
    while (!stack.empty()) {
        slice = stack.back();
        stack.pop_back();
        end_pt = src[slice.end];
        pos = slice.start;
        READ_PT(start_pt, pos);
        if (pos != slice.end) {
            double dx, dy, dist, max_dist = 0;
            dx = end_pt.x - start_pt.x;
            dy = end_pt.y - start_pt.y;
            while (pos != slice.end) {
                READ_PT(pt, pos);
                dist = std::fabs((pt.y - start_pt.y) * dx - (pt.x - start_pt.x) * dy);
                if (dist > max_dist) { max_dist = dist; right_slice.start = (pos + count - 1) % count; }
            }
            le_eps = max_dist * max_dist <= eps * (dx * dx + dy * dy);
        } else {
            le_eps = true;
            start_pt = src[slice.start];
        }
        if (le_eps) {
            dst[new_count++] = start_pt;
        } else {
            right_slice.end = slice.end;
            slice.end = right_slice.start;
            stack.push_back(right_slice);
            stack.push_back(slice);
        }
    }
    // final clean-up: remove extra points on [almost] straight lines
    count = new_count;
    pos = count - 1;
    READ_DST_PT(start_pt, pos);
    wpos = pos;
    READ_DST_PT(pt, pos);
    for (i = 0; i < count && new_count > 2; i++) {
        double dx, dy, dist, successive_inner_product;
        READ_DST_PT(end_pt, pos);
        dx = end_pt.x - start_pt.x;
        dy = end_pt.y - start_pt.y;
        dist = std::fabs((pt.x - start_pt.x) * dy - (pt.y - start_pt.y) * dx);
        successive_inner_product = (double)(pt.x - start_pt.x) * (end_pt.x - pt.x) +
                                   (double)(pt.y - start_pt.y) * (end_pt.y - pt.y);
        if (dist * dist <= 0.5 * eps * (dx * dx + dy * dy) && dx != 0 && dy != 0 && successive_inner_product >= 0) {
            new_count--;
            dst[wpos] = start_pt = end_pt;
            if (++wpos >= count) wpos = 0;
            READ_DST_PT(pt, pos);
            i++;
            continue;
        }
        dst[wpos] = start_pt = pt;
        if (++wpos >= count) wpos = 0;
        pt = end_pt;
    }
    dstv.assign(dst.begin(), dst.begin() + new_count);
}

// cv::isContourConvex for integer points (imgproc/convhull.cpp::isContourConvex_<int>)
bool is_contour_convex(const std::vector<Pt>& p) {
    int n = (int)p.size();
    Pt prev_pt = p[(n - 2 + n) % n];
    Pt cur_pt = p[n - 1];
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

// aruco.cpp::_findMarkerContours
void find_marker_contours(const uint8_t* thresh, int rows, int cols, const DetectorParams& P, int scale,
                          std::vector<Candidate>& out) {
    unsigned minPerimeterPixels = (unsigned)(P.minMarkerPerimeterRate * std::max(cols, rows));
    unsigned maxPerimeterPixels = (unsigned)(P.maxMarkerPerimeterRate * std::max(cols, rows));
    std::vector<Contour> contours;
    find_contours_list_none(thresh, rows, cols, contours);
    for (const Contour& c : contours) {
        if (c.pts.size() < minPerimeterPixels || c.pts.size() > maxPerimeterPixels) continue;
        std::vector<Pt> approx;
        approx_poly_dp_closed(c.pts, double(c.pts.size()) * P.polygonalApproxAccuracyRate, approx);
        if (approx.size() != 4 || !is_contour_convex(approx)) continue;
        double minDistSq = (double)std::max(cols, rows) * std::max(cols, rows);
        for (int j = 0; j < 4; j++) {
            double d = (double)(approx[j].x - approx[(j + 1) % 4].x) * (double)(approx[j].x - approx[(j + 1) % 4].x) +
                       (double)(approx[j].y - approx[(j + 1) % 4].y) * (double)(approx[j].y - approx[(j + 1) % 4].y);
            minDistSq = std::min(minDistSq, d);
        }
        double minCornerDistancePixels = double(c.pts.size()) * P.minCornerDistanceRate;
        if (minDistSq < minCornerDistancePixels * minCornerDistancePixels) continue;
        bool tooNearBorder = false;
        for (int j = 0; j < 4; j++)
            if (approx[j].x < P.minDistanceToBorder || approx[j].y < P.minDistanceToBorder ||
                approx[j].x > cols - 1 - P.minDistanceToBorder || approx[j].y > rows - 1 - P.minDistanceToBorder)
                tooNearBorder = true;
        if (tooNearBorder) continue;
        Candidate cand;
        for (int j = 0; j < 4; j++) cand.c[j] = Pt2f{(float)approx[j].x, (float)approx[j].y};
        cand.contour_size = (int)c.pts.size();
        cand.scale = scale;
        cand.key = c.key;
        out.push_back(cand);
    }
}

// aruco.cpp::_detectInitialCandidates: one adaptive threshold + contour pass per window size,
// candidates joined in window order.
void detect_initial_candidates(const uint8_t* gray, int rows, int cols, const DetectorParams& P,
                               std::vector<Candidate>& out) {
    out.clear();
    int nScales = (P.adaptiveThreshWinSizeMax - P.adaptiveThreshWinSizeMin) / P.adaptiveThreshWinSizeStep + 1;
    std::vector<uint8_t> thresh((size_t)rows * cols);
    for (int i = 0; i < nScales; i++) {
        int win = P.adaptiveThreshWinSizeMin + i * P.adaptiveThreshWinSizeStep;
        if (win % 2 == 0) win++;
        adaptive_threshold_mean_inv(gray, rows, cols, win, P.adaptiveThreshConstant, thresh.data());
        find_marker_contours(thresh.data(), rows, cols, P, i, out);
    }
}

// aruco.cpp::_reorderCandidatesCorners
void reorder_candidate_corners(std::vector<Candidate>& cs) {
    for (Candidate& c : cs) {
        double dx1 = c.c[1].x - c.c[0].x, dy1 = c.c[1].y - c.c[0].y;
        double dx2 = c.c[2].x - c.c[0].x, dy2 = c.c[2].y - c.c[0].y;
        double cross = (dx1 * dy2) - (dy1 * dx2);
        if (cross < 0.0) std::swap(c.c[1], c.c[3]);
    }
}

// aruco.cpp::_filterTooCloseCandidates (3.2.0: "perimeter" is the contour point count)
void filter_too_close_candidates(const std::vector<Candidate>& in, std::vector<Candidate>& out, double rate) {
    std::vector<std::pair<int, int>> near;
    for (unsigned i = 0; i < in.size(); i++) {
        for (unsigned j = i + 1; j < in.size(); j++) {
            int minimumPerimeter = std::min(in[i].contour_size, in[j].contour_size);
            for (int fc = 0; fc < 4; fc++) {
                double distSq = 0;
                for (int c = 0; c < 4; c++) {
                    int modC = (c + fc) % 4;
                    distSq += (in[i].c[modC].x - in[j].c[c].x) * (in[i].c[modC].x - in[j].c[c].x) +
                              (in[i].c[modC].y - in[j].c[c].y) * (in[i].c[modC].y - in[j].c[c].y);
                }
                distSq /= 4.;
                double minMarkerDistancePixels = double(minimumPerimeter) * rate;
                if (distSq < minMarkerDistancePixels * minMarkerDistancePixels) {
                    near.push_back({(int)i, (int)j});
                    break;
                }
            }
        }
    }
    std::vector<bool> toRemove(in.size(), false);
    for (auto& pr : near) {
        if (toRemove[pr.first] || toRemove[pr.second]) continue;
        size_t p1 = in[pr.first].contour_size, p2 = in[pr.second].contour_size;
        if (p1 > p2) toRemove[pr.second] = true;
        else toRemove[pr.first] = true;
    }
    out.clear();
    for (unsigned i = 0; i < in.size(); i++)
        if (!toRemove[i]) out.push_back(in[i]);
}

// cv::getPerspectiveTransform: 8x8 linear system.  OpenCV 3.2 solves it with DECOMP_SVD; the SVD
// code is not reproducible offline, so this spec fixes Gaussian elimination with partial pivoting
// (first maximal pivot), row operations in the order written.  Same solution up to rounding.
void get_perspective_transform(const Pt2f src[4], const Pt2f dst[4], double M[9]) {
    double a[8][8], b[8];
    for (int i = 0; i < 4; ++i) {
        a[i][0] = a[i + 4][3] = src[i].x;
        a[i][1] = a[i + 4][4] = src[i].y;
        a[i][2] = a[i + 4][5] = 1;
        a[i][3] = a[i][4] = a[i][5] = a[i + 4][0] = a[i + 4][1] = a[i + 4][2] = 0;
        a[i][6] = -(double)src[i].x * dst[i].x;
        a[i][7] = -(double)src[i].y * dst[i].x;
        a[i + 4][6] = -(double)src[i].x * dst[i].y;
        a[i + 4][7] = -(double)src[i].y * dst[i].y;
        b[i] = dst[i].x;
        b[i + 4] = dst[i].y;
    }
    for (int col = 0; col < 8; col++) {
        int piv = col;
        double best = std::fabs(a[col][col]);
        for (int r = col + 1; r < 8; r++)
            if (std::fabs(a[r][col]) > best) { best = std::fabs(a[r][col]); piv = r; }
        if (piv != col) {
            for (int c = 0; c < 8; c++) std::swap(a[piv][c], a[col][c]);
            std::swap(b[piv], b[col]);
        }
        for (int r = col + 1; r < 8; r++) {
            double f = a[r][col] / a[col][col];
            for (int c = col; c < 8; c++) a[r][c] -= f * a[col][c];
            b[r] -= f * b[col];
        }
    }
    double x[8];
    for (int i = 7; i >= 0; i--) {
        double s = b[i];
        for (int c = i + 1; c < 8; c++) s -= a[i][c] * x[c];
        x[i] = s / a[i][i];
    }
    for (int i = 0; i < 8; i++) M[i] = x[i];
    M[8] = 1.0;
}

// cv::warpPerspective(INTER_NEAREST, BORDER_CONSTANT 0): M is inverted with the closed 3x3 cofactor
// formula (cv::invert, n == 3), source position = cvRound((M0*x + M1*y + M2) / w) (round half even).
void warp_perspective_nearest(const uint8_t* gray, int rows, int cols, const double Min[9], int dsize, uint8_t* dst) {
    const double* m = Min;
    double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    double M[9];
    if (det != 0.) {
        double d = 1. / det;
        M[0] = (m[4] * m[8] - m[5] * m[7]) * d;
        M[1] = (m[2] * m[7] - m[1] * m[8]) * d;
        M[2] = (m[1] * m[5] - m[2] * m[4]) * d;
        M[3] = (m[5] * m[6] - m[3] * m[8]) * d;
        M[4] = (m[0] * m[8] - m[2] * m[6]) * d;
        M[5] = (m[2] * m[3] - m[0] * m[5]) * d;
        M[6] = (m[3] * m[7] - m[4] * m[6]) * d;
        M[7] = (m[1] * m[6] - m[0] * m[7]) * d;
        M[8] = (m[0] * m[4] - m[1] * m[3]) * d;
    } else {
        for (int i = 0; i < 9; i++) M[i] = 0;
    }
    for (int y = 0; y < dsize; y++) {
        double X0 = M[1] * y + M[2], Y0 = M[4] * y + M[5], W0 = M[7] * y + M[8];
        for (int x = 0; x < dsize; x++) {
            double W = W0 + M[6] * x;
            W = W ? 1. / W : 0;
            double fX = std::max((double)INT_MIN, std::min((double)INT_MAX, (X0 + M[0] * x) * W));
            double fY = std::max((double)INT_MIN, std::min((double)INT_MAX, (Y0 + M[3] * x) * W));
            long X = std::lrint(fX), Y = std::lrint(fY);
            uint8_t v = 0;
            if (X >= 0 && X < cols && Y >= 0 && Y < rows) v = gray[(size_t)Y * cols + X];
            dst[y * dsize + x] = v;
        }
    }
}

// imgproc/thresh.cpp::getThreshVal_Otsu_8u
int otsu_threshold(const uint8_t* img, int n) {
    const int N = 256;
    int h[N] = {0};
    for (int i = 0; i < n; i++) h[img[i]]++;
    double mu = 0, scale = 1. / n;
    for (int i = 0; i < N; i++) mu += i * (double)h[i];
    mu *= scale;
    double mu1 = 0, q1 = 0, max_sigma = 0, max_val = 0;
    for (int i = 0; i < N; i++) {
        double p_i, q2, mu2, sigma;
        p_i = h[i] * scale;
        mu1 *= q1;
        q1 += p_i;
        q2 = 1. - q1;
        if (std::min(q1, q2) < FLT_EPSILON || std::max(q1, q2) > 1. - FLT_EPSILON) continue;
        mu1 = (mu1 + i * p_i) / q1;
        mu2 = (mu - q1 * mu1) / q2;
        sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
        if (sigma > max_sigma) { max_sigma = sigma; max_val = i; }
    }
    return (int)max_val;
}

// aruco.cpp::_extractBits
void extract_bits(const uint8_t* gray, int rows, int cols, const Pt2f corners[4], int markerSize,
                  const DetectorParams& P, std::vector<uint8_t>& bits) {
    const int cellSize = P.perspectiveRemovePixelPerCell;
    const int sizeWB = markerSize + 2 * P.markerBorderBits;
    const int cellMarginPixels = int(P.perspectiveRemoveIgnoredMarginPerCell * cellSize);
    const int S = sizeWB * cellSize;
    Pt2f dstc[4] = {{0, 0}, {(float)S - 1, 0}, {(float)S - 1, (float)S - 1}, {0, (float)S - 1}};
    double M[9];
    get_perspective_transform(corners, dstc, M);
    std::vector<uint8_t> img((size_t)S * S);
    warp_perspective_nearest(gray, rows, cols, M, S, img.data());
    bits.assign((size_t)sizeWB * sizeWB, 0);

    // meanStdDev of the inner region (half a cell removed on every side)
    const int lo = cellSize / 2, hi = S - cellSize / 2;
    long long sum = 0, sq = 0;
    for (int y = lo; y < hi; y++)
        for (int x = lo; x < hi; x++) { int v = img[y * S + x]; sum += v; sq += v * v; }
    const double scale = 1.0 / ((double)(hi - lo) * (hi - lo));
    const double mean = sum * scale;
    const double var = std::max(sq * scale - mean * mean, 0.);
    const double stddev = std::sqrt(var);
    if (stddev < P.minOtsuStdDev) {
        if (mean > 127) std::fill(bits.begin(), bits.end(), 1);
        return;
    }
    const int T = otsu_threshold(img.data(), S * S);
    for (int y = 0; y < sizeWB; y++)
        for (int x = 0; x < sizeWB; x++) {
            int Xs = x * cellSize + cellMarginPixels, Ys = y * cellSize + cellMarginPixels;
            int w = cellSize - 2 * cellMarginPixels;
            int nz = 0;
            for (int yy = 0; yy < w; yy++)
                for (int xx = 0; xx < w; xx++) nz += img[(Ys + yy) * S + Xs + xx] > T;
            if (nz > (w * w) / 2) bits[y * sizeWB + x] = 1;
        }
}

// aruco/dictionary.cpp::getByteListFromBits
static void byte_list_from_bits(const uint8_t* bits, int n, int nbytes, uint8_t* out /*4*nbytes*/) {
    std::memset(out, 0, 4 * nbytes);
    int currentBit = 0, currentByte = 0;
    uint8_t *rot0 = out, *rot1 = out + nbytes, *rot2 = out + 2 * nbytes, *rot3 = out + 3 * nbytes;
    for (int row = 0; row < n; row++)
        for (int col = 0; col < n; col++) {
            rot0[currentByte] <<= 1; rot1[currentByte] <<= 1; rot2[currentByte] <<= 1; rot3[currentByte] <<= 1;
            rot0[currentByte] |= bits[row * n + col];
            rot1[currentByte] |= bits[col * n + (n - 1 - row)];
            rot2[currentByte] |= bits[(n - 1 - row) * n + (n - 1 - col)];
            rot3[currentByte] |= bits[(n - 1 - col) * n + row];
            if (++currentBit == 8) { currentBit = 0; currentByte++; }
        }
}

// DICT_ARUCO_ORIGINAL (enum 16, parameters.yaml:16): the original ArUco 5x5 code.  Each row carries
// two id bits (MSB first) through the words {10000, 10111, 01001, 01110}; 1024 markers,
// maxCorrectionBits = 0.  Generated from first principles (OpenCV's table itself is not available).
Dictionary make_dict_aruco_original() {
    Dictionary d;
    d.markerSize = 5; d.maxCorrectionBits = 0; d.nMarkers = 1024; d.nbytes = (25 + 7) / 8;
    d.bytesList.resize((size_t)d.nMarkers * 4 * d.nbytes);
    static const int words[4] = {0x10, 0x17, 0x09, 0x0e};
    for (int id = 0; id < 1024; id++) {
        uint8_t bits[25];
        for (int y = 0; y < 5; y++) {
            int val = words[(id >> (2 * (4 - y))) & 3];
            for (int x = 0; x < 5; x++) bits[y * 5 + x] = (val >> (4 - x)) & 1;
        }
        byte_list_from_bits(bits, 5, d.nbytes, &d.bytesList[(size_t)id * 4 * d.nbytes]);
    }
    return d;
}

// A dictionary handed over as data: bits[n][ms*ms] row-major, 1 = white (what cv::aruco::Dictionary::getBitsFromByteList
// returns per marker).  The rotations are generated here exactly as Dictionary::getByteListFromBits does.
Dictionary make_dict_from_bits(int markerSize, int nMarkers, int maxCorrectionBits, const uint8_t* bits) {
    Dictionary d;
    d.markerSize = markerSize; d.maxCorrectionBits = maxCorrectionBits; d.nMarkers = nMarkers;
    d.nbytes = (markerSize * markerSize + 7) / 8;
    d.bytesList.resize((size_t)nMarkers * 4 * d.nbytes);
    for (int id = 0; id < nMarkers; id++)
        byte_list_from_bits(bits + (size_t)id * markerSize * markerSize, markerSize, d.nbytes, &d.bytesList[(size_t)id * 4 * d.nbytes]);
    return d;
}

// aruco/dictionary.cpp::Dictionary::identify
bool dictionary_identify(const Dictionary& d, const uint8_t* onlyBits, int& idx, int& rotation, double rate) {
    int maxCorrectionRecalculed = int(double(d.maxCorrectionBits) * rate);
    std::vector<uint8_t> cand(4 * d.nbytes);
    byte_list_from_bits(onlyBits, d.markerSize, d.nbytes, cand.data());
    idx = -1;
    for (int m = 0; m < d.nMarkers; m++) {
        int currentMinDistance = d.markerSize * d.markerSize + 1;
        int currentRotation = -1;
        for (unsigned r = 0; r < 4; r++) {
            const uint8_t* a = &d.bytesList[((size_t)m * 4 + r) * d.nbytes];
            int ham = 0;
            for (int k = 0; k < d.nbytes; k++) ham += __builtin_popcount((unsigned)(a[k] ^ cand[k]));
            if (ham < currentMinDistance) { currentMinDistance = ham; currentRotation = (int)r; }
        }
        if (currentMinDistance <= maxCorrectionRecalculed) { idx = m; rotation = currentRotation; break; }
    }
    return idx != -1;
}

// aruco.cpp::_getBorderErrors
static int border_errors(const uint8_t* bits, int markerSize, int borderSize) {
    int n = markerSize + 2 * borderSize, total = 0;
    for (int y = 0; y < n; y++)
        for (int k = 0; k < borderSize; k++) {
            if (bits[y * n + k] != 0) total++;
            if (bits[y * n + n - 1 - k] != 0) total++;
        }
    for (int x = borderSize; x < n - borderSize; x++)
        for (int k = 0; k < borderSize; k++) {
            if (bits[k * n + x] != 0) total++;
            if (bits[(n - 1 - k) * n + x] != 0) total++;
        }
    return total;
}

// aruco.cpp::_identifyOneCandidate
bool identify_one_candidate(const Dictionary& d, const uint8_t* gray, int rows, int cols, Pt2f corners[4],
                            int& id, const DetectorParams& P) {
    std::vector<uint8_t> bits;
    extract_bits(gray, rows, cols, corners, d.markerSize, P, bits);
    int maximumErrorsInBorder = int(d.markerSize * d.markerSize * P.maxErroneousBitsInBorderRate);
    if (border_errors(bits.data(), d.markerSize, P.markerBorderBits) > maximumErrorsInBorder) return false;
    int n = d.markerSize + 2 * P.markerBorderBits, b = P.markerBorderBits;
    std::vector<uint8_t> only((size_t)d.markerSize * d.markerSize);
    for (int y = 0; y < d.markerSize; y++)
        for (int x = 0; x < d.markerSize; x++) only[y * d.markerSize + x] = bits[(y + b) * n + x + b];
    int rotation;
    if (!dictionary_identify(d, only.data(), id, rotation, P.errorCorrectionRate)) return false;
    if (rotation != 0) std::rotate(corners, corners + 4 - rotation, corners + 4);
    return true;
}

// cv::pointPolygonTest(measureDist=false) for a float contour (imgproc/geometry.cpp)
static int point_polygon_test(const Pt2f* cnt, int total, Pt2f pt) {
    int counter = 0;
    Pt2f v = cnt[total - 1], v0;
    for (int i = 0; i < total; i++) {
        v0 = v;
        v = cnt[i];
        if ((v0.y <= pt.y && v.y <= pt.y) || (v0.y > pt.y && v.y > pt.y) || (v0.x < pt.x && v.x < pt.x)) {
            if (pt.y == v.y && (pt.x == v.x || (pt.y == v0.y && ((v0.x <= pt.x && pt.x <= v.x) || (v.x <= pt.x && pt.x <= v0.x)))))
                return 0;
            continue;
        }
        double dist = (double)(pt.y - v0.y) * (v.x - v0.x) - (double)(pt.x - v0.x) * (v.y - v0.y);
        if (dist == 0) return 0;
        if (v.y < v0.y) dist = -dist;
        counter += dist > 0;
    }
    return counter % 2 == 0 ? -1 : 1;
}

// aruco.cpp::_filterDetectedMarkers
void filter_detected_markers(std::vector<Detection>& d) {
    if (d.empty()) return;
    std::vector<bool> toRemove(d.size(), false);
    for (unsigned i = 0; i + 1 < d.size(); i++)
        for (unsigned j = i + 1; j < d.size(); j++) {
            if (d[i].id != d[j].id) continue;
            bool inside = true;
            for (unsigned p = 0; p < 4; p++)
                if (point_polygon_test(d[i].c, 4, d[j].c[p]) < 0) { inside = false; break; }
            if (inside) { toRemove[j] = true; continue; }
            inside = true;
            for (unsigned p = 0; p < 4; p++)
                if (point_polygon_test(d[j].c, 4, d[i].c[p]) < 0) { inside = false; break; }
            if (inside) { toRemove[i] = true; continue; }
        }
    std::vector<Detection> out;
    for (unsigned i = 0; i < d.size(); i++)
        if (!toRemove[i]) out.push_back(d[i]);
    d.swap(out);
}

// cv::aruco::detectMarkers (aruco.cpp), corner refinement off (default)
void detect_markers(const uint8_t* img, int rows, int cols, int channels, size_t step, const Dictionary& dict,
                    const DetectorParams& P, std::vector<Detection>& out, std::vector<Candidate>* cand_after_filter) {
    std::vector<uint8_t> gray((size_t)rows * cols);
    if (channels == 3) bgr_to_gray(img, rows, cols, step, gray.data());
    else for (int y = 0; y < rows; y++) std::memcpy(&gray[(size_t)y * cols], img + (size_t)y * step, cols);

    std::vector<Candidate> initial, cands;
    detect_initial_candidates(gray.data(), rows, cols, P, initial);
    reorder_candidate_corners(initial);
    filter_too_close_candidates(initial, cands, P.minMarkerDistanceRate);
    if (cand_after_filter) *cand_after_filter = cands;

    out.clear();
    for (Candidate& c : cands) {
        int id;
        Pt2f corners[4] = {c.c[0], c.c[1], c.c[2], c.c[3]};
        if (identify_one_candidate(dict, gray.data(), rows, cols, corners, id, P)) {
            Detection d;
            d.id = id;
            for (int k = 0; k < 4; k++) d.c[k] = corners[k];
            out.push_back(d);
        }
    }
    filter_detected_markers(out);
    // aruco.cpp::detectMarkers: "if (params->doCornerRefinement)" -> cornerSubPix on the grey image, per marker, after filtering
    if (P.doCornerRefinement)
        for (Detection& d : out)
            corner_sub_pix(gray.data(), rows, cols, d.c, 4, P.cornerRefinementWinSize, P.cornerRefinementMaxIterations, P.cornerRefinementMinAccuracy);
}

// imgproc/samplers.cpp::getRectSubPix for CV_8UC1 -> CV_32FC1 (3.2.0): bilinear patch of win_w x win_h centred on `center`.
// Interior fast path: dst[j] = prev + t with prev carried as (float)(t * (1-a)/a); otherwise the generic path on a rectangle
// clipped to the image with replicated borders.
void get_rect_sub_pix_8u32f(const uint8_t* src, int rows, int cols, int win_w, int win_h, float cx, float cy, float* dst) {
    float centerx = cx - (win_w - 1) * 0.5f, centery = cy - (win_h - 1) * 0.5f;
    int ipx = (int)std::floor(centerx), ipy = (int)std::floor(centery);
    if (0 <= ipx && ipx + win_w < cols && 0 <= ipy && ipy + win_h < rows) {
        float a = centerx - ipx, b = centery - ipy;
        a = a > 0.0001f ? a : 0.0001f;
        float a12 = a * (1.f - b), a22 = a * b, b1 = 1.f - b, b2 = b;
        double s = (1. - a) / a;
        const uint8_t* p = src + (size_t)ipy * cols + ipx;
        for (int i = 0; i < win_h; i++, p += cols, dst += win_w) {
            float prev = (1 - a) * (b1 * p[0] + b2 * p[cols]);
            for (int j = 0; j < win_w; j++) {
                float t = a12 * p[j + 1] + a22 * p[j + 1 + cols];
                dst[j] = prev + t;
                prev = (float)(t * s);
            }
        }
        return;
    }
    // getRectSubPix_Cn_<uchar, float, float, nop, nop> with adjustRect: replicate the border
    float a = centerx - ipx, b = centery - ipy;
    float a11 = (1.f - a) * (1.f - b), a12 = a * (1.f - b), a21 = (1.f - a) * b, a22 = a * b;
    for (int i = 0; i < win_h; i++)
        for (int j = 0; j < win_w; j++) {
            int y0 = std::min(std::max(ipy + i, 0), rows - 1), y1 = std::min(std::max(ipy + i + 1, 0), rows - 1);
            int x0 = std::min(std::max(ipx + j, 0), cols - 1), x1 = std::min(std::max(ipx + j + 1, 0), cols - 1);
            dst[i * win_w + j] = src[(size_t)y0 * cols + x0] * a11 + src[(size_t)y0 * cols + x1] * a12 + src[(size_t)y1 * cols + x0] * a21 +
                                 src[(size_t)y1 * cols + x1] * a22;
        }
}

// imgproc/cornersubpix.cpp::cornerSubPix (3.2.0), zeroZone = (-1,-1), criteria = MAX_ITER | EPS
void corner_sub_pix(const uint8_t* gray, int rows, int cols, Pt2f* corners, int count, int win, int maxCount, double epsilon) {
    const int MAX_ITERS = 100;
    const int win_w = win * 2 + 1, win_h = win * 2 + 1;
    const int max_iters = std::min(std::max(maxCount, 1), MAX_ITERS);
    double eps = std::max(epsilon, 0.);
    eps *= eps;
    std::vector<float> mask((size_t)win_w * win_h), buf((size_t)(win_w + 2) * (win_h + 2));
    for (int i = 0; i < win_h; i++) {
        float y = (float)(i - win) / win;
        float vy = std::exp(-y * y);
        for (int j = 0; j < win_w; j++) {
            float x = (float)(j - win) / win;
            mask[i * win_w + j] = (float)(vy * std::exp(-x * x));
        }
    }
    for (int pt = 0; pt < count; pt++) {
        const Pt2f cT = corners[pt];
        Pt2f cI = cT;
        int iter = 0;
        double err = 0;
        do {
            Pt2f cI2;
            double a = 0, b = 0, c = 0, bb1 = 0, bb2 = 0;
            get_rect_sub_pix_8u32f(gray, rows, cols, win_w + 2, win_h + 2, cI.x, cI.y, buf.data());
            const float* subpix = &buf[(size_t)(win_w + 2) + 1];
            for (int i = 0, k = 0; i < win_h; i++, subpix += win_w + 2) {
                double py = i - win;
                for (int j = 0; j < win_w; j++, k++) {
                    double m = mask[k];
                    double tgx = subpix[j + 1] - subpix[j - 1];
                    double tgy = subpix[j + win_w + 2] - subpix[j - win_w - 2];
                    double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m;
                    double px = j - win;
                    a += gxx; b += gxy; c += gyy;
                    bb1 += gxx * px + gxy * py;
                    bb2 += gxy * px + gyy * py;
                }
            }
            double det = a * c - b * b;
            if (std::fabs(det) <= DBL_EPSILON * DBL_EPSILON) break;
            double scale = 1.0 / det;
            cI2.x = (float)(cI.x + c * scale * bb1 - b * scale * bb2);
            cI2.y = (float)(cI.y - b * scale * bb1 + a * scale * bb2);
            err = (cI2.x - cI.x) * (cI2.x - cI.x) + (cI2.y - cI.y) * (cI2.y - cI.y);
            cI = cI2;
            if (cI.x < 0 || cI.x >= cols || cI.y < 0 || cI.y >= rows) break;
        } while (++iter < max_iters && err > eps);
        if (std::fabs(cI.x - cT.x) > win || std::fabs(cI.y - cT.y) > win) cI = cT;     // poor convergence: keep the initial point
        corners[pt] = cI;
    }
}

} // namespace oracle
