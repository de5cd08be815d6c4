// TEST INFRASTRUCTURE — NOT PRODUCT CODE.  PARITY UNPINNED (see below).
//
// CPU restatement of the OpenCV 3.4 leaf functions the reference's per-frame front
// end calls.  OpenCV is a third-party dependency that is NOT
// under /root/reference and is not installed in this image (no headers, no libs, no
// cv2), and the reference pins no version (CMakeLists.txt:31-38 accepts >=3.0;
// the committed Thirdparty/DBoW2/lib/libDBoW2.so links libopencv_world.so.3.4).
// Every function here is therefore restated from the *published algorithm* of
// OpenCV 3.4.x's non-IPP, non-OpenCL CPU path and anchored on the reference's call
// sites.  The reference ships no tests, golden vectors or sample images for this
// path, so nothing in it pins these semantics: PARITY UNPINNED.  The frozen choices
// are listed in DESIGN.md ("Oracle spec").
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.
#pragma once
#include <cstdint>
#include <cmath>
#include <cfloat>
#include <cstring>
#include <vector>
#include <algorithm>

namespace cvl {

// cvRound: SSE2 cvtsd2si / lrint => round-half-to-even under the default FP mode.
// Call sites: ORBextractor.cc:81,115,119-120,442,456-460,1112.
static inline int cvRound(double v) { return (int)lrint(v); }
static inline int cvRound(float v) { return (int)lrintf(v); }
static inline int cvFloor(double v) { int i = (int)v; return i - (i > v); }
static inline int cvCeil(double v) { int i = (int)v; return i + (i < v); }

// BORDER_REFLECT_101 index map: gfedcb|abcdefgh|gfedcba  (copyMakeBorder, ORBextractor.cc:1122-1128)
static inline int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    }
    return p;
}

// cv::resize(src, dst, sz, 0, 0, INTER_LINEAR) for CV_8UC1 (ORBextractor.cc:1120).
// Fixed-point path: INTER_RESIZE_COEF_BITS = 11, HResizeLinear<uchar,int,short,2048>,
// VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>> (8u specialisation).
static inline void resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride,
                                    uint8_t* dst, int dw, int dh, int dstride)
{
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> ialpha(2 * dw), ibeta(2 * dh);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cvFloor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        float c0 = 1.f - fx, c1 = fx;
        ialpha[2 * dx] = (short)cvRound(c0 * 2048);
        ialpha[2 * dx + 1] = (short)cvRound(c1 * 2048);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cvFloor(fy);
        fy -= sy;
        yofs[dy] = sy;
        float c0 = 1.f - fy, c1 = fy;
        ibeta[2 * dy] = (short)cvRound(c0 * 2048);
        ibeta[2 * dy + 1] = (short)cvRound(c1 * 2048);
    }
    std::vector<int> row0(dw), row1(dw);
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = yofs[dy];
        int r0 = std::min(std::max(sy0, 0), sh - 1);       // clip(sy0 + 0, 0, sh)
        int r1 = std::min(std::max(sy0 + 1, 0), sh - 1);   // clip(sy0 + 1, 0, sh)
        const uint8_t* S0 = src + (size_t)r0 * sstride;
        const uint8_t* S1 = src + (size_t)r1 * sstride;
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx];
            int sx1 = std::min(sx + 1, sw - 1);
            int a0 = ialpha[2 * dx], a1 = ialpha[2 * dx + 1];
            row0[dx] = S0[sx] * a0 + S0[sx1] * a1;
            row1[dx] = S1[sx] * a0 + S1[sx1] * a1;
        }
        int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
        uint8_t* D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; dx++)
            D[dx] = (uint8_t)((((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2);
    }
}

// cv::FAST(img, kps, threshold, true)  ==  FAST_t<16> (FAST-9/16) with 3x3 NMS.
// Call sites: ORBextractor.cc:809,814.  Structure follows OpenCV's row-buffer
// formulation on purpose (the HIP path uses a different, threshold-free score-map
// formulation; agreement of the two is part of what the parity tests show).
struct FastPt { int x, y, score; };

static inline int fast_corner_score16(const uint8_t* ptr, const int pixel[25], int threshold)
{
    const int K = 8, N = K * 3 + 1;
    int k, v = ptr[0];
    short d[N];
    for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
    int a0 = threshold;
    for (k = 0; k < 16; k += 2) {
        int a = std::min((int)d[k + 1], (int)d[k + 2]);
        a = std::min(a, (int)d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, (int)d[k + 4]);
        a = std::min(a, (int)d[k + 5]);
        a = std::min(a, (int)d[k + 6]);
        a = std::min(a, (int)d[k + 7]);
        a = std::min(a, (int)d[k + 8]);
        a0 = std::max(a0, std::min(a, (int)d[k]));
        a0 = std::max(a0, std::min(a, (int)d[k + 9]));
    }
    int b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = std::max((int)d[k + 1], (int)d[k + 2]);
        b = std::max(b, (int)d[k + 3]);
        b = std::max(b, (int)d[k + 4]);
        b = std::max(b, (int)d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, (int)d[k + 6]);
        b = std::max(b, (int)d[k + 7]);
        b = std::max(b, (int)d[k + 8]);
        b0 = std::min(b0, std::max(b, (int)d[k]));
        b0 = std::min(b0, std::max(b, (int)d[k + 9]));
    }
    return -b0 - 1;
}

static inline void fast9_16_nms(const uint8_t* img, int cols, int rows, int step, int threshold,
                                std::vector<FastPt>& out)
{
    static const int offs[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
                                    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};
    const int K = 8, N = 16 + K + 1;
    int pixel[25];
    for (int k = 0; k < 16; k++) pixel[k] = offs[k][0] + offs[k][1] * step;
    for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
    out.clear();
    threshold = std::min(std::max(threshold, 0), 255);
    uint8_t tab[512];
    for (int i = -255; i <= 255; i++) tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);
    if (cols < 7 || rows < 7) return;
    std::vector<uint8_t> sbuf((size_t)cols * 3, 0);
    std::vector<int> cbuf((size_t)(cols + 1) * 3, 0);
    uint8_t* buf[3] = {sbuf.data(), sbuf.data() + cols, sbuf.data() + 2 * cols};
    int* cpbuf[3] = {cbuf.data() + 1, cbuf.data() + 1 + (cols + 1), cbuf.data() + 1 + 2 * (cols + 1)};
    for (int i = 3; i < rows - 2; i++) {
        const uint8_t* ptr = img + (size_t)i * step + 3;
        uint8_t* curr = buf[(i - 3) % 3];
        int* cornerpos = cpbuf[(i - 3) % 3];
        memset(curr, 0, cols);
        int ncorners = 0;
        if (i < rows - 3) {
            for (int j = 3; j < cols - 3; j++, ptr++) {
                int v = ptr[0];
                const uint8_t* t = &tab[0] - v + 255;
                int d = t[ptr[pixel[0]]] | t[ptr[pixel[8]]];
                if (d == 0) continue;
                d &= t[ptr[pixel[2]]] | t[ptr[pixel[10]]];
                d &= t[ptr[pixel[4]]] | t[ptr[pixel[12]]];
                d &= t[ptr[pixel[6]]] | t[ptr[pixel[14]]];
                if (d == 0) continue;
                d &= t[ptr[pixel[1]]] | t[ptr[pixel[9]]];
                d &= t[ptr[pixel[3]]] | t[ptr[pixel[11]]];
                d &= t[ptr[pixel[5]]] | t[ptr[pixel[13]]];
                d &= t[ptr[pixel[7]]] | t[ptr[pixel[15]]];
                if (d & 1) {
                    int vt = v - threshold, count = 0;
                    for (int k = 0; k < N; k++) {
                        int x = ptr[pixel[k]];
                        if (x < vt) {
                            if (++count > K) {
                                cornerpos[ncorners++] = j;
                                curr[j] = (uint8_t)fast_corner_score16(ptr, pixel, threshold);
                                break;
                            }
                        } else count = 0;
                    }
                }
                if (d & 2) {
                    int vt = v + threshold, count = 0;
                    for (int k = 0; k < N; k++) {
                        int x = ptr[pixel[k]];
                        if (x > vt) {
                            if (++count > K) {
                                cornerpos[ncorners++] = j;
                                curr[j] = (uint8_t)fast_corner_score16(ptr, pixel, threshold);
                                break;
                            }
                        } else count = 0;
                    }
                }
            }
        }
        cornerpos[-1] = ncorners;
        if (i == 3) continue;
        const uint8_t* prev = buf[(i - 4 + 3) % 3];
        const uint8_t* pprev = buf[(i - 5 + 3) % 3];
        cornerpos = cpbuf[(i - 4 + 3) % 3];
        ncorners = cornerpos[-1];
        for (int k = 0; k < ncorners; k++) {
            int j = cornerpos[k];
            int score = prev[j];
            if (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j] &&
                score > pprev[j + 1] && score > curr[j - 1] && score > curr[j] && score > curr[j + 1])
                out.push_back(FastPt{j, i - 1, score});
        }
    }
}

// cv::fastAtan2(y, x) (degrees, [0,360)); scalar path of mathfuncs_core, no FMA.
// Call site: ORBextractor.cc:103.
static inline float fastAtan2(float y, float x)
{
    static const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
    static const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
    static const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
    static const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
    float ax = std::abs(x), ay = std::abs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// cv::GaussianBlur(img, img, Size(7,7), 2, 2, BORDER_REFLECT_101) on a continuous
// CV_8UC1 image (ORBextractor.cc:1085-1086).  OpenCV >=3.4.2 non-IPP builds take the
// bit-exact fixed-point path: 8.8 taps (ufixedpoint16), exact u16 horizontal sums,
// 16.16 vertical sums, result = (sum + 0x8000) >> 16.  The 7 taps are a SPEC
// PARAMETER (default {18,34,48,56,48,34,18}/256: sigma=2 taps rounded with error
// carry so that they sum to 256); both passes are exact, so the result equals the
// non-separable integer sum.
static inline void gaussian7x7_fixed(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride,
                                     const uint16_t taps[7])
{
    std::vector<uint32_t> hb((size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint32_t s = 0;
            for (int k = -3; k <= 3; k++) s += (uint32_t)taps[k + 3] * src[(size_t)y * sstride + reflect101(x + k, w)];
            hb[(size_t)y * w + x] = s;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint32_t s = 0;
            for (int k = -3; k <= 3; k++) s += (uint32_t)taps[k + 3] * hb[(size_t)reflect101(y + k, h) * w + x];
            uint32_t v = (s + 0x8000u) >> 16;
            dst[(size_t)y * dstride + x] = (uint8_t)(v > 255 ? 255 : v);
        }
}

// cvtColor(*2GRAY) for 8-bit: (R*4899 + G*9617 + B*1868 + (1<<13)) >> 14
// (Tracking.cc:175-200,256-269).  `rgb_order` = Camera.RGB (1: R first, 0: B first).
static inline void cvt_gray_u8(const uint8_t* src, int w, int h, int sstride, int channels, int rgb_order,
                               uint8_t* dst, int dstride)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t* p = src + (size_t)y * sstride + (size_t)x * channels;
            int r = rgb_order ? p[0] : p[2], g = p[1], b = rgb_order ? p[2] : p[0];
            dst[(size_t)y * dstride + x] = (uint8_t)((r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14);
        }
}

} // namespace cvl
