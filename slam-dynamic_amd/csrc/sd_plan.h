// Host-side geometry plan of the extractor: everything ORBextractor derives from its
// constructor arguments and from the image size, computed once per (extractor, WxH) with
// exactly the reference's float/double expression order, then uploaded as small tables.
// Reference: src/ORBextractor.cc:410-470 (ctor), :765-800 (cell grid), :541-563 (initial
// quadtree nodes), :1107-1113 (level sizes); OpenCV resize coefficient tables (Appendix A).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#define SD_EDGE 19          // EDGE_THRESHOLD, ORBextractor.cc:74
#define SD_XOFF 32          // column of interior x=0 inside a padded row (32-B aligned interior)
#define SD_HALF_PATCH 15    // HALF_PATCH_SIZE, ORBextractor.cc:73
#define SD_PATCH 31         // PATCH_SIZE, ORBextractor.cc:72
#define SD_MAX_LEVELS 12
#define SD_TH_HIGH 100      // ORBmatcher.cc:37
#define SD_TH_LOW 50        // ORBmatcher.cc:38

struct SdLevel {            // POD, copied to the device by value
    int W, H;               // interior size of mvImagePyramid[level]
    int stride;             // padded row stride (bytes)
    int pyrOffset;          // byte offset of the padded plane inside one image's pyramid block
    int blurStride, blurOffset;
    int minBX, minBY, maxBX, maxBY;
    int nCols, nRows, wCell, hCell;
    int cell0, nCells;      // range in the flat cell table
    int quota;              // mnFeaturesPerLevel[level]
    int nIni;               // initial quadtree nodes
    float hX;
    int candOffset, candCap;   // per-level slice of the candidate arrays
    int kpOffset, kpCap;       // per-level slice of the per-level keypoint scratch
    int tabOffset;             // offset of this level's resize tables (x tables then y tables)
    float scale, invScale;
    float sizeF;               // (float)(int)(PATCH_SIZE*scale)
    int maxNodes;              // node-array capacity needed by the quadtree kernel
};

struct SdCell {             // one FAST cell window (ORBextractor.cc:789-816)
    short level, pad;
    short x0, y0, x1, y1;   // window [x0,x1) x [y0,y1) in level-interior coordinates
    short jw, ih;           // j*wCell, i*hCell: shift applied at :822-823
    int listOffset;         // offset of this cell's slots in the per-image cell-list array
    int cap;                // max NMS survivors
};

static inline int sd_cvRound(double v) { return (int)lrint(v); }
static inline int sd_cvRoundf(float v) { return (int)lrintf(v); }

struct SdParams {
    int nfeatures; double scaleFactor; int nlevels, iniTh, minTh;
    float scale[SD_MAX_LEVELS], inv[SD_MAX_LEVELS], sigma2[SD_MAX_LEVELS], invSigma2[SD_MAX_LEVELS];
    int quota[SD_MAX_LEVELS];
    int umax[16];
    uint16_t blurTaps[7];
};

// ORBextractor::ORBextractor, src/ORBextractor.cc:410-470
static inline void sd_params_init(SdParams& p, int nfeatures, float scaleFactor, int nlevels, int ini, int mn)
{
    p.nfeatures = nfeatures; p.scaleFactor = scaleFactor; p.nlevels = nlevels; p.iniTh = ini; p.minTh = mn;
    p.scale[0] = 1.0f; p.sigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        p.scale[i] = (float)(p.scale[i - 1] * p.scaleFactor);
        p.sigma2[i] = p.scale[i] * p.scale[i];
    }
    for (int i = 0; i < nlevels; i++) { p.inv[i] = 1.0f / p.scale[i]; p.invSigma2[i] = 1.0f / p.sigma2[i]; }
    float factor = (float)(1.0f / p.scaleFactor);
    float nDesired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) {
        p.quota[l] = sd_cvRoundf(nDesired);
        sum += p.quota[l];
        nDesired *= factor;
    }
    p.quota[nlevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
    int v, v0;
    int vmax = (int)floor(SD_HALF_PATCH * sqrt(2.f) / 2 + 1);
    int vmin = (int)ceil(SD_HALF_PATCH * sqrt(2.f) / 2);
    const double hp2 = SD_HALF_PATCH * SD_HALF_PATCH;
    for (v = 0; v <= vmax; ++v) p.umax[v] = sd_cvRound(sqrt(hp2 - v * v));
    for (v = SD_HALF_PATCH, v0 = 0; v >= vmin; --v) {
        while (p.umax[v0] == p.umax[v0 + 1]) ++v0;
        p.umax[v] = v0;
        ++v0;
    }
    const uint16_t taps[7] = {18, 34, 48, 56, 48, 34, 18};
    memcpy(p.blurTaps, taps, sizeof(taps));
}

struct SdPlan {
    int W = 0, H = 0, nlevels = 0;
    SdLevel lv[SD_MAX_LEVELS];
    std::vector<SdCell> cells;
    // resize tables for levels >= 1, concatenated: per level W column entries {xofs,a0,a1,0} + H row entries {yofs,b0,b1,0}
    std::vector<int16_t> tabs;
    size_t pyrImageBytes = 0, blurImageBytes = 0;
    int cellListCap = 0;     // slots per image in the cell-list array
    int candCapTotal = 0;    // == cellListCap (compact candidates)
    int kpCapLevels = 0;     // sum of kpCap over levels
    int kpCap = 0;           // capacity of the final per-image keypoint array
    int kpMax = 0;           // most keypoints an image can produce (sum of the per-level capacities, without the slice padding)
    int maxNodesAll = 0;
    int maxWin = 0;          // largest FAST window side over all cells
    std::string error;
};

static inline int sd_align(int v, int a) { return (v + a - 1) / a * a; }

// Returns false (plan.error set) when the geometry is outside what the kernels support.
static inline bool sd_plan_build(SdPlan& P, const SdParams& prm, int W, int H, int minKpCap = 0)
{
    P = SdPlan();
    P.W = W; P.H = H; P.nlevels = prm.nlevels;
    if (W > 4095 || H > 4095) { P.error = "image larger than 4095 px"; return false; }
    size_t pyrOff = 0, blurOff = 0;
    int cellTotal = 0, listOff = 0, candOff = 0, kpOff = 0, tabOff = 0;
    for (int l = 0; l < prm.nlevels; l++) {
        SdLevel& g = P.lv[l];
        memset(&g, 0, sizeof(g));
        g.scale = prm.scale[l]; g.invScale = prm.inv[l];
        g.W = sd_cvRoundf((float)W * g.invScale);     // ORBextractor.cc:1112
        g.H = sd_cvRoundf((float)H * g.invScale);
        g.sizeF = (float)(int)(SD_PATCH * g.scale);   // :837,846
        g.quota = prm.quota[l];
        if (g.W < 2 * SD_EDGE + 2 || g.H < 2 * SD_EDGE + 2) {
            P.error = "pyramid level " + std::to_string(l) + " smaller than the 19-px border allows";
            return false;
        }
        g.stride = sd_align(SD_XOFF + g.W + SD_EDGE + 4, 64);
        g.pyrOffset = (int)pyrOff;
        pyrOff += (size_t)g.stride * (g.H + 2 * SD_EDGE);
        g.blurStride = sd_align(g.W, 64);
        g.blurOffset = (int)blurOff;
        blurOff += (size_t)g.blurStride * g.H;
        // cell grid, ORBextractor.cc:773-787
        g.minBX = SD_EDGE - 3; g.minBY = g.minBX;
        g.maxBX = g.W - SD_EDGE + 3; g.maxBY = g.H - SD_EDGE + 3;
        const float width = (float)(g.maxBX - g.minBX), height = (float)(g.maxBY - g.minBY);
        g.nCols = (int)(width / 30.f); g.nRows = (int)(height / 30.f);
        if (g.nCols < 1 || g.nRows < 1) { P.error = "level " + std::to_string(l) + " has no FAST cell"; return false; }
        g.wCell = (int)ceilf(width / g.nCols); g.hCell = (int)ceilf(height / g.nRows);
        if (g.wCell + 6 > 70 || g.hCell + 6 > 70) { P.error = "FAST cell window larger than 70 px"; return false; }   // cannot happen: wCell <= 59 (k_fast_cells holds 70)
        g.cell0 = cellTotal;
        g.candOffset = candOff;
        for (int i = 0; i < g.nRows; i++) {
            const float iniY = (float)(g.minBY + i * g.hCell);
            float maxY = iniY + g.hCell + 6;
            if (iniY >= g.maxBY - 3) continue;
            if (maxY > g.maxBY) maxY = (float)g.maxBY;
            for (int j = 0; j < g.nCols; j++) {
                const float iniX = (float)(g.minBX + j * g.wCell);
                float maxX = iniX + g.wCell + 6;
                if (iniX >= g.maxBX - 6) continue;
                if (maxX > g.maxBX) maxX = (float)g.maxBX;
                SdCell c;
                c.level = (short)l; c.pad = 0;
                c.x0 = (short)(int)iniX; c.x1 = (short)(int)maxX; c.y0 = (short)(int)iniY; c.y1 = (short)(int)maxY;
                c.jw = (short)(j * g.wCell); c.ih = (short)(i * g.hCell);
                int sw = c.x1 - c.x0 - 6, sh = c.y1 - c.y0 - 6;   // scanned area (FAST skips a 3-px frame)
                c.cap = (sw > 0 && sh > 0) ? ((sw + 1) / 2) * ((sh + 1) / 2) : 0;
                c.listOffset = listOff;
                if (c.x1 - c.x0 > P.maxWin) P.maxWin = c.x1 - c.x0;
                if (c.y1 - c.y0 > P.maxWin) P.maxWin = c.y1 - c.y0;
                listOff += c.cap;
                P.cells.push_back(c);
                cellTotal++;
            }
        }
        g.nCells = cellTotal - g.cell0;
        if (g.nCells > 4096) { P.error = "more than 4096 FAST cells in a level"; return false; }
        g.candCap = listOff - candOff;
        candOff = listOff;
        if (g.candCap >= (1 << 24)) { P.error = "too many candidate slots"; return false; }
        // initial quadtree nodes, ORBextractor.cc:543-545
        g.nIni = (int)roundf(width / height);
        if (g.nIni < 1) { P.error = "portrait level (width/height rounds to 0): undefined in the reference"; return false; }
        g.hX = width / g.nIni;
        int bound = g.quota > g.nIni ? g.quota : g.nIni;
        g.maxNodes = 4 * bound + 16;
        g.kpCap = (g.quota + 3 > 4 * g.nIni ? g.quota + 3 : 4 * g.nIni) + 1;
        g.kpOffset = kpOff;
        P.kpMax += g.kpCap;
        kpOff += (g.kpCap + 7) & ~7;      // slices start on multiples of 8: the slots of one k_orient / k_describe workgroup share a level
        if (g.maxNodes > P.maxNodesAll) P.maxNodesAll = g.maxNodes;
        // resize coefficient tables (OpenCV resize INTER_LINEAR, 8u)
        g.tabOffset = tabOff;
        if (l > 0) {
            const SdLevel& s = P.lv[l - 1];
            const double scale_x = 1. / ((double)g.W / s.W), scale_y = 1. / ((double)g.H / s.H);
            // layout per level: W entries {xofs, a0, a1, 0} then H entries {yofs, b0, b1, 0}; tabOffset counts ENTRIES (4 x int16)
            size_t base = P.tabs.size();
            P.tabs.resize(base + 4 * ((size_t)g.W + (size_t)g.H));
            int16_t* ct = &P.tabs[base]; int16_t* rt = ct + 4 * (size_t)g.W;
            for (int dx = 0; dx < g.W; dx++) {
                float fx = (float)((dx + 0.5) * scale_x - 0.5);
                int sx = (int)floorf(fx);
                fx -= sx;
                if (sx < 0) { fx = 0; sx = 0; }
                if (sx >= s.W - 1) { fx = 0; sx = s.W - 1; }
                float c0 = 1.f - fx, c1 = fx;
                ct[4 * dx] = (int16_t)sx; ct[4 * dx + 1] = (int16_t)sd_cvRoundf(c0 * 2048);
                ct[4 * dx + 2] = (int16_t)sd_cvRoundf(c1 * 2048); ct[4 * dx + 3] = 0;
            }
            for (int dy = 0; dy < g.H; dy++) {
                float fy = (float)((dy + 0.5) * scale_y - 0.5);
                int sy = (int)floorf(fy);
                fy -= sy;
                float c0 = 1.f - fy, c1 = fy;
                rt[4 * dy] = (int16_t)sy; rt[4 * dy + 1] = (int16_t)sd_cvRoundf(c0 * 2048);
                rt[4 * dy + 2] = (int16_t)sd_cvRoundf(c1 * 2048); rt[4 * dy + 3] = 0;
            }
            tabOff = (int)(P.tabs.size() / 4);
        }
    }
    P.pyrImageBytes = (pyrOff + 255) / 256 * 256;
    P.blurImageBytes = (blurOff + 255) / 256 * 256;
    P.cellListCap = listOff;
    P.candCapTotal = listOff;
    P.kpCapLevels = kpOff;
    P.kpCap = kpOff > minKpCap ? kpOff : sd_align(minKpCap, 8);      // row stride of the per-image result arrays (a tracker with a second, larger extractor shares slots)
    if (P.tabs.empty()) P.tabs.resize(4, 0);
    return true;
}
