// TEST INFRASTRUCTURE — NOT PRODUCT CODE.  PARITY UNPINNED.
//
// CPU restatement ("oracle") of the reference's per-frame front end
// (li-guihai/slam-dynamic, an ORB-SLAM2 fork).  Every function cites the reference
// file:line it follows.  The OpenCV leaves it calls are restated in cv_leaves.h.
// The reference has no tests / golden vectors / fixtures for this path and cannot be
// compiled here (OpenCV, Eigen, PCL, Pangolin absent), so this oracle is pinned only by
// (a) the constants the reference source holds (sampling pattern, thresholds, derived
// umax / per-level quotas; tests/test_oracle_constants.py) and (b) its own committed
// regression fixtures under tests/golden/.  => "parity unpinned" (DESIGN.md).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
//
// Written as C-style C++ (std::vector / std::list / std::sort only) so that the
// quadtree keeps the reference's std::list push_front / erase semantics literally.
#include "cv_leaves.h"
#include <list>
#include <utility>
#include <cstdio>
#include <climits>
#include <cstddef>
#include <map>
#include <fstream>
#include <sstream>
#include <string>
using std::ptrdiff_t;

using namespace cvl;

namespace {

const int PATCH_SIZE = 31;        // ORBextractor.cc:72
const int HALF_PATCH_SIZE = 15;   // :73
const int EDGE_THRESHOLD = 19;    // :74

const signed char bit_pattern_31[256 * 4] = {
#include "orb_pattern.inc"
};

struct KeyPoint {   // same 28-byte layout as cv::KeyPoint
    float x, y, size, angle, response;
    int octave, class_id;
};

struct Plane {      // one pyramid level: interior WxH inside a buffer padded by EDGE_THRESHOLD
    int W = 0, H = 0, stride = 0;
    std::vector<uint8_t> buf;
    uint8_t* interior() { return buf.data() + (size_t)EDGE_THRESHOLD * stride + EDGE_THRESHOLD; }
    const uint8_t* interior() const { return buf.data() + (size_t)EDGE_THRESHOLD * stride + EDGE_THRESHOLD; }
};

struct ExtractorNode {            // ORBextractor.h:31-43
    std::vector<KeyPoint> vKeys;
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
    std::list<ExtractorNode>::iterator lit;
    bool bNoMore = false;
    long seq = 0;   // creation sequence: deterministic stand-in for the node *address* the
                    // reference sorts by at ORBextractor.cc:684 (DESIGN.md "Oracle spec" Q1)
    void DivideNode(ExtractorNode& n1, ExtractorNode& n2, ExtractorNode& n3, ExtractorNode& n4);
};

// ORBextractor.cc:481-537
void ExtractorNode::DivideNode(ExtractorNode& n1, ExtractorNode& n2, ExtractorNode& n3, ExtractorNode& n4)
{
    const int halfX = (int)ceilf(static_cast<float>(URx - ULx) / 2);
    const int halfY = (int)ceilf(static_cast<float>(BRy - ULy) / 2);
    n1.ULx = ULx; n1.ULy = ULy;
    n1.URx = ULx + halfX; n1.URy = ULy;
    n1.BLx = ULx; n1.BLy = ULy + halfY;
    n1.BRx = ULx + halfX; n1.BRy = ULy + halfY;
    n2.ULx = n1.URx; n2.ULy = n1.URy;
    n2.URx = URx; n2.URy = URy;
    n2.BLx = n1.BRx; n2.BLy = n1.BRy;
    n2.BRx = URx; n2.BRy = ULy + halfY;
    n3.ULx = n1.BLx; n3.ULy = n1.BLy;
    n3.URx = n1.BRx; n3.URy = n1.BRy;
    n3.BLx = BLx; n3.BLy = BLy;
    n3.BRx = n1.BRx; n3.BRy = BLy;
    n4.ULx = n3.URx; n4.ULy = n3.URy;
    n4.URx = n2.BRx; n4.URy = n2.BRy;
    n4.BLx = n3.BRx; n4.BLy = n3.BRy;
    n4.BRx = BRx; n4.BRy = BRy;
    for (size_t i = 0; i < vKeys.size(); i++) {
        const KeyPoint& kp = vKeys[i];
        if (kp.x < n1.URx) {
            if (kp.y < n1.BRy) n1.vKeys.push_back(kp);
            else n3.vKeys.push_back(kp);
        } else if (kp.y < n1.BRy) n2.vKeys.push_back(kp);
        else n4.vKeys.push_back(kp);
    }
    if (n1.vKeys.size() == 1) n1.bNoMore = true;
    if (n2.vKeys.size() == 1) n2.bNoMore = true;
    if (n3.vKeys.size() == 1) n3.bNoMore = true;
    if (n4.vKeys.size() == 1) n4.bNoMore = true;
}

struct Extractor {
    int nfeatures; double scaleFactor; int nlevels, iniThFAST, minThFAST;
    std::vector<int> mnFeaturesPerLevel, umax;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<Plane> mvImagePyramid;
    std::vector<Plane> mvBlurred;   // the per-level "workingMat" (interior only), kept for tests
    uint16_t blurTaps[7] = {18, 34, 48, 56, 48, 34, 18};
    long seqCounter = 0;

    // ORBextractor.cc:410-470
    Extractor(int _nfeatures, float _scaleFactor, int _nlevels, int _ini, int _min)
        : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_ini), minThFAST(_min)
    {
        mvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels);
        mvScaleFactor[0] = 1.0f; mvLevelSigma2[0] = 1.0f;
        for (int i = 1; i < nlevels; i++) {
            mvScaleFactor[i] = (float)(mvScaleFactor[i - 1] * scaleFactor);   // float*double -> double -> float
            mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
        }
        mvInvScaleFactor.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
        for (int i = 0; i < nlevels; i++) {
            mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
            mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
        }
        mvImagePyramid.resize(nlevels); mvBlurred.resize(nlevels);
        mnFeaturesPerLevel.resize(nlevels);
        float factor = (float)(1.0f / scaleFactor);
        float nDesiredFeaturesPerScale = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
        int sumFeatures = 0;
        for (int level = 0; level < nlevels - 1; level++) {
            mnFeaturesPerLevel[level] = cvRound(nDesiredFeaturesPerScale);
            sumFeatures += mnFeaturesPerLevel[level];
            nDesiredFeaturesPerScale *= factor;
        }
        mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sumFeatures, 0);
        umax.resize(HALF_PATCH_SIZE + 1);
        int v, v0, vmax = cvFloor(HALF_PATCH_SIZE * sqrt(2.f) / 2 + 1);
        int vmin = cvCeil(HALF_PATCH_SIZE * sqrt(2.f) / 2);
        const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
        for (v = 0; v <= vmax; ++v) umax[v] = cvRound(sqrt(hp2 - v * v));
        for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
            while (umax[v0] == umax[v0 + 1]) ++v0;
            umax[v] = v0;
            ++v0;
        }
    }

    // ORBextractor.cc:1107-1132
    void ComputePyramid(const uint8_t* image, int cols, int rows, int step)
    {
        for (int level = 0; level < nlevels; ++level) {
            float scale = mvInvScaleFactor[level];
            int sw = cvRound((float)cols * scale), sh = cvRound((float)rows * scale);
            Plane& P = mvImagePyramid[level];
            P.W = sw; P.H = sh; P.stride = sw + EDGE_THRESHOLD * 2;
            P.buf.assign((size_t)P.stride * (sh + EDGE_THRESHOLD * 2), 0);
            if (level != 0) {
                const Plane& S = mvImagePyramid[level - 1];
                resize_linear_u8(S.interior(), S.W, S.H, S.stride, P.interior(), sw, sh, P.stride);
            } else {
                for (int y = 0; y < rows; y++) memcpy(P.interior() + (size_t)y * P.stride, image + (size_t)y * step, cols);
            }
            // copyMakeBorder(..., BORDER_REFLECT_101 [+BORDER_ISOLATED])
            uint8_t* I = P.interior();
            for (int y = -EDGE_THRESHOLD; y < sh + EDGE_THRESHOLD; y++) {
                int sy = reflect101(y, sh);
                for (int x = -EDGE_THRESHOLD; x < sw + EDGE_THRESHOLD; x++) {
                    if (y >= 0 && y < sh && x >= 0 && x < sw) continue;
                    I[(ptrdiff_t)y * P.stride + x] = I[(ptrdiff_t)sy * P.stride + reflect101(x, sw)];
                }
            }
        }
    }

    // ORBextractor.cc:539-763
    std::vector<KeyPoint> DistributeOctTree(const std::vector<KeyPoint>& vToDistributeKeys, const int minX, const int maxX,
                                            const int minY, const int maxY, const int N, const int /*level*/)
    {
        std::vector<KeyPoint> vResultKeys;
        const int nIni = (int)roundf(static_cast<float>(maxX - minX) / (maxY - minY));
        if (nIni < 1) return vResultKeys;   // reference: UB (portrait images); spec: unsupported, empty
        const float hX = static_cast<float>(maxX - minX) / nIni;
        std::list<ExtractorNode> lNodes;
        std::vector<ExtractorNode*> vpIniNodes(nIni);
        for (int i = 0; i < nIni; i++) {
            ExtractorNode ni;
            ni.ULx = (int)(hX * static_cast<float>(i)); ni.ULy = 0;
            ni.URx = (int)(hX * static_cast<float>(i + 1)); ni.URy = 0;
            ni.BLx = ni.ULx; ni.BLy = maxY - minY;
            ni.BRx = ni.URx; ni.BRy = maxY - minY;
            ni.seq = seqCounter++;
            lNodes.push_back(ni);
            vpIniNodes[i] = &lNodes.back();
        }
        for (size_t i = 0; i < vToDistributeKeys.size(); i++) {
            const KeyPoint& kp = vToDistributeKeys[i];
            vpIniNodes[(int)(kp.x / hX)]->vKeys.push_back(kp);
        }
        std::list<ExtractorNode>::iterator lit = lNodes.begin();
        while (lit != lNodes.end()) {
            if (lit->vKeys.size() == 1) { lit->bNoMore = true; lit++; }
            else if (lit->vKeys.empty()) lit = lNodes.erase(lit);
            else lit++;
        }
        bool bFinish = false;
        std::vector<std::pair<int, ExtractorNode*> > vSizeAndPointerToNode;
        auto push_child = [&](ExtractorNode& n, int& nToExpand) {
            if (n.vKeys.size() > 0) {
                n.seq = seqCounter++;
                lNodes.push_front(n);
                if (n.vKeys.size() > 1) {
                    nToExpand++;
                    vSizeAndPointerToNode.push_back(std::make_pair((int)n.vKeys.size(), &lNodes.front()));
                    lNodes.front().lit = lNodes.begin();
                }
            }
        };
        // (size, address) ordering of ORBextractor.cc:684 with address := creation sequence
        auto lessSizeSeq = [](const std::pair<int, ExtractorNode*>& a, const std::pair<int, ExtractorNode*>& b) {
            if (a.first != b.first) return a.first < b.first;
            return a.second->seq < b.second->seq;
        };
        while (!bFinish) {
            int prevSize = (int)lNodes.size();
            lit = lNodes.begin();
            int nToExpand = 0;
            vSizeAndPointerToNode.clear();
            while (lit != lNodes.end()) {
                if (lit->bNoMore) { lit++; continue; }
                ExtractorNode n1, n2, n3, n4;
                lit->DivideNode(n1, n2, n3, n4);
                push_child(n1, nToExpand); push_child(n2, nToExpand);
                push_child(n3, nToExpand); push_child(n4, nToExpand);
                lit = lNodes.erase(lit);
            }
            if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) {
                bFinish = true;
            } else if (((int)lNodes.size() + nToExpand * 3) > N) {
                while (!bFinish) {
                    prevSize = (int)lNodes.size();
                    std::vector<std::pair<int, ExtractorNode*> > vPrev = vSizeAndPointerToNode;
                    vSizeAndPointerToNode.clear();
                    std::sort(vPrev.begin(), vPrev.end(), lessSizeSeq);
                    for (int j = (int)vPrev.size() - 1; j >= 0; j--) {
                        ExtractorNode n1, n2, n3, n4;
                        vPrev[j].second->DivideNode(n1, n2, n3, n4);
                        int dummy = 0;
                        push_child(n1, dummy); push_child(n2, dummy);
                        push_child(n3, dummy); push_child(n4, dummy);
                        lNodes.erase(vPrev[j].second->lit);
                        if ((int)lNodes.size() >= N) break;
                    }
                    if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) bFinish = true;
                }
            }
        }
        vResultKeys.reserve(nfeatures);
        for (std::list<ExtractorNode>::iterator it = lNodes.begin(); it != lNodes.end(); it++) {
            std::vector<KeyPoint>& vNodeKeys = it->vKeys;
            KeyPoint* pKP = &vNodeKeys[0];
            float maxResponse = pKP->response;
            for (size_t k = 1; k < vNodeKeys.size(); k++)
                if (vNodeKeys[k].response > maxResponse) { pKP = &vNodeKeys[k]; maxResponse = vNodeKeys[k].response; }
            vResultKeys.push_back(*pKP);
        }
        return vResultKeys;
    }

    // ORBextractor.cc:77-104
    float IC_Angle(const Plane& P, float ptx, float pty)
    {
        int m_01 = 0, m_10 = 0;
        const uint8_t* center = P.interior() + (ptrdiff_t)cvRound(pty) * P.stride + cvRound(ptx);
        for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
        int step = P.stride;
        for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
            int v_sum = 0;
            int d = umax[v];
            for (int u = -d; u <= d; ++u) {
                int val_plus = center[u + v * step], val_minus = center[u - v * step];
                v_sum += (val_plus - val_minus);
                m_10 += u * (val_plus + val_minus);
            }
            m_01 += v * v_sum;
        }
        return fastAtan2((float)m_01, (float)m_10);
    }

    // ORBextractor.cc:765-853; also fills per-level candidate counts for tests
    std::vector<int> candCount;
    void ComputeKeyPointsOctTree(std::vector<std::vector<KeyPoint> >& allKeypoints)
    {
        allKeypoints.resize(nlevels);
        candCount.assign(nlevels, 0);
        const float W = 30;
        for (int level = 0; level < nlevels; ++level) {
            const Plane& P = mvImagePyramid[level];
            const int minBorderX = EDGE_THRESHOLD - 3;
            const int minBorderY = minBorderX;
            const int maxBorderX = P.W - EDGE_THRESHOLD + 3;
            const int maxBorderY = P.H - EDGE_THRESHOLD + 3;
            std::vector<KeyPoint> vToDistributeKeys;
            vToDistributeKeys.reserve(nfeatures * 10);
            const float width = (float)(maxBorderX - minBorderX);
            const float height = (float)(maxBorderY - minBorderY);
            const int nCols = (int)(width / W);
            const int nRows = (int)(height / W);
            if (nCols < 1 || nRows < 1) { allKeypoints[level].clear(); continue; }   // reference: div by zero
            const int wCell = (int)ceilf(width / nCols);
            const int hCell = (int)ceilf(height / nRows);
            std::vector<FastPt> cell;
            for (int i = 0; i < nRows; i++) {
                const float iniY = (float)(minBorderY + i * hCell);
                float maxY = iniY + hCell + 6;
                if (iniY >= maxBorderY - 3) continue;
                if (maxY > maxBorderY) maxY = (float)maxBorderY;
                for (int j = 0; j < nCols; j++) {
                    const float iniX = (float)(minBorderX + j * wCell);
                    float maxX = iniX + wCell + 6;
                    if (iniX >= maxBorderX - 6) continue;
                    if (maxX > maxBorderX) maxX = (float)maxBorderX;
                    const int x0 = (int)iniX, x1 = (int)maxX, y0 = (int)iniY, y1 = (int)maxY;
                    const uint8_t* sub = P.interior() + (ptrdiff_t)y0 * P.stride + x0;
                    fast9_16_nms(sub, x1 - x0, y1 - y0, P.stride, iniThFAST, cell);
                    if (cell.empty()) fast9_16_nms(sub, x1 - x0, y1 - y0, P.stride, minThFAST, cell);
                    for (size_t k = 0; k < cell.size(); k++) {
                        KeyPoint kp;
                        kp.x = (float)cell[k].x; kp.y = (float)cell[k].y;
                        kp.size = 7.f; kp.angle = -1; kp.response = (float)cell[k].score;
                        kp.octave = 0; kp.class_id = -1;
                        kp.x += j * wCell; kp.y += i * hCell;
                        vToDistributeKeys.push_back(kp);
                    }
                }
            }
            candCount[level] = (int)vToDistributeKeys.size();
            std::vector<KeyPoint>& keypoints = allKeypoints[level];
            keypoints = DistributeOctTree(vToDistributeKeys, minBorderX, maxBorderX, minBorderY, maxBorderY,
                                          mnFeaturesPerLevel[level], level);
            const int scaledPatchSize = (int)(PATCH_SIZE * mvScaleFactor[level]);
            for (size_t i = 0; i < keypoints.size(); i++) {
                keypoints[i].x += minBorderX;
                keypoints[i].y += minBorderY;
                keypoints[i].octave = level;
                keypoints[i].size = (float)scaledPatchSize;
            }
        }
        for (int level = 0; level < nlevels; ++level)
            for (size_t i = 0; i < allKeypoints[level].size(); i++)
                allKeypoints[level][i].angle = IC_Angle(mvImagePyramid[level], allKeypoints[level][i].x, allKeypoints[level][i].y);
    }

    // ORBextractor.cc:107-147.  a,b := correctly rounded f32 cos/sin of the f32 angle
    // (DESIGN.md "Oracle spec" Q3); products/sums in f32 without contraction.
    void computeOrbDescriptor(const KeyPoint& kpt, const uint8_t* img, int step, uint8_t* desc)
    {
        const float factorPI = (float)(M_PI / 180.f);
        float angle = (float)kpt.angle * factorPI;
        float a = (float)cos((double)angle), b = (float)sin((double)angle);
        const uint8_t* center = img + (ptrdiff_t)cvRound(kpt.y) * step + cvRound(kpt.x);
        const signed char* pattern = bit_pattern_31;
        for (int i = 0; i < 32; ++i, pattern += 32) {
            int val = 0;
            for (int k = 0; k < 8; k++) {
                const float x0 = (float)pattern[4 * k], y0 = (float)pattern[4 * k + 1];
                const float x1 = (float)pattern[4 * k + 2], y1 = (float)pattern[4 * k + 3];
                // built with -ffp-contract=off: two roundings per product-sum, as on an SSE2 build
                int t0 = center[cvRound(x0 * b + y0 * a) * step + cvRound(x0 * a - y0 * b)];
                int t1 = center[cvRound(x1 * b + y1 * a) * step + cvRound(x1 * a - y1 * b)];
                val |= (t0 < t1) << k;
            }
            desc[i] = (uint8_t)val;
        }
    }

    // ORBextractor.cc:1043-1105
    void run(const uint8_t* image, int cols, int rows, int step, std::vector<KeyPoint>& _keypoints,
             std::vector<uint8_t>& descriptors, std::vector<int>& perLevel)
    {
        _keypoints.clear(); descriptors.clear(); perLevel.assign(nlevels, 0);
        if (!image || cols <= 0 || rows <= 0) return;
        ComputePyramid(image, cols, rows, step);
        std::vector<std::vector<KeyPoint> > allKeypoints;
        ComputeKeyPointsOctTree(allKeypoints);
        int nkeypoints = 0;
        for (int level = 0; level < nlevels; ++level) nkeypoints += (int)allKeypoints[level].size();
        descriptors.assign((size_t)nkeypoints * 32, 0);
        _keypoints.reserve(nkeypoints);
        int offset = 0;
        for (int level = 0; level < nlevels; ++level) {
            std::vector<KeyPoint>& keypoints = allKeypoints[level];
            int nkeypointsLevel = (int)keypoints.size();
            perLevel[level] = nkeypointsLevel;
            const Plane& P = mvImagePyramid[level];
            Plane& B = mvBlurred[level];
            B.W = P.W; B.H = P.H; B.stride = P.W;
            B.buf.assign((size_t)P.W * P.H, 0);
            // workingMat = level.clone(); GaussianBlur(7x7, 2, 2, REFLECT_101).  The reference
            // skips the blur when the level has no keypoints; the plane is only an intermediate.
            gaussian7x7_fixed(P.interior(), P.W, P.H, P.stride, B.buf.data(), B.stride, blurTaps);
            if (nkeypointsLevel == 0) continue;
            for (int i = 0; i < nkeypointsLevel; i++)
                computeOrbDescriptor(keypoints[i], B.buf.data(), B.stride, &descriptors[(size_t)(offset + i) * 32]);
            offset += nkeypointsLevel;
            if (level != 0) {
                float scale = mvScaleFactor[level];
                for (int i = 0; i < nkeypointsLevel; i++) { keypoints[i].x *= scale; keypoints[i].y *= scale; }
            }
            _keypoints.insert(_keypoints.end(), keypoints.begin(), keypoints.end());
        }
    }
};

// ORBmatcher.cc:1804-1820
inline int DescriptorDistance(const uint8_t* a, const uint8_t* b)
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4); memcpy(&pb, b + 4 * i, 4);
        unsigned int v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

const int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;   // ORBmatcher.cc:37-39
const int FRAME_GRID_ROWS = 48, FRAME_GRID_COLS = 64;      // Frame.h:39-40

} // namespace

// ---------------------------------------------------------------------------------
// C entry points (ctypes).  All buffers are caller-owned.
// ---------------------------------------------------------------------------------
extern "C" {

void* orc_extractor_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
{
    return new Extractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST);
}
void orc_extractor_destroy(void* h) { delete (Extractor*)h; }
void orc_extractor_set_blur_taps(void* h, const uint16_t* taps) { memcpy(((Extractor*)h)->blurTaps, taps, 14); }

// tables: scale[nlevels], inv[nlevels], sigma2, invsigma2, quota[nlevels], umax[16]
void orc_extractor_tables(void* h, float* scale, float* inv, float* sigma2, float* invsigma2, int* quota, int* umax)
{
    Extractor* e = (Extractor*)h;
    for (int i = 0; i < e->nlevels; i++) {
        scale[i] = e->mvScaleFactor[i]; inv[i] = e->mvInvScaleFactor[i];
        sigma2[i] = e->mvLevelSigma2[i]; invsigma2[i] = e->mvInvLevelSigma2[i];
        quota[i] = e->mnFeaturesPerLevel[i];
    }
    for (int i = 0; i < 16; i++) umax[i] = e->umax[i];
}

// ORBextractor::operator() — returns number of keypoints (<= cap), kp as 28-byte records.
int orc_extract(void* h, const uint8_t* gray, int w, int hgt, int stride, void* kp_out, uint8_t* desc_out, int cap,
                int* per_level, int* cand_per_level)
{
    Extractor* e = (Extractor*)h;
    std::vector<KeyPoint> kps; std::vector<uint8_t> desc; std::vector<int> pl;
    e->run(gray, w, hgt, stride, kps, desc, pl);
    int n = (int)kps.size();
    if (n > cap) return -n;
    if (n) { memcpy(kp_out, kps.data(), (size_t)n * sizeof(KeyPoint)); memcpy(desc_out, desc.data(), (size_t)n * 32); }
    if (per_level) for (int i = 0; i < e->nlevels; i++) per_level[i] = pl[i];
    if (cand_per_level) for (int i = 0; i < e->nlevels; i++) cand_per_level[i] = e->candCount[i];
    return n;
}

// mvImagePyramid[level] geometry + copy-out (padded plane, stride = W+38) and blurred plane.
void orc_pyramid_dims(void* h, int level, int* W, int* H)
{
    Extractor* e = (Extractor*)h; *W = e->mvImagePyramid[level].W; *H = e->mvImagePyramid[level].H;
}
void orc_pyramid_copy(void* h, int level, uint8_t* padded_out)
{
    Extractor* e = (Extractor*)h; const Plane& P = e->mvImagePyramid[level];
    memcpy(padded_out, P.buf.data(), P.buf.size());
}
void orc_blurred_copy(void* h, int level, uint8_t* out)
{
    Extractor* e = (Extractor*)h; const Plane& P = e->mvBlurred[level];
    memcpy(out, P.buf.data(), P.buf.size());
}

int orc_descriptor_distance(const uint8_t* a, const uint8_t* b) { return DescriptorDistance(a, b); }

// Leaves exposed for unit tests
void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride)
{
    resize_linear_u8(src, sw, sh, sstride, dst, dw, dh, dstride);
}
int orc_fast(const uint8_t* img, int cols, int rows, int step, int threshold, int* xys, int cap)
{
    std::vector<FastPt> v; fast9_16_nms(img, cols, rows, step, threshold, v);
    int n = (int)v.size();
    for (int i = 0; i < n && i < cap; i++) { xys[3 * i] = v[i].x; xys[3 * i + 1] = v[i].y; xys[3 * i + 2] = v[i].score; }
    return n;
}
float orc_fast_atan2(float y, float x) { return fastAtan2(y, x); }
void orc_gaussian7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride, const uint16_t* taps)
{
    gaussian7x7_fixed(src, w, h, sstride, dst, dstride, taps);
}
void orc_cvt_gray(const uint8_t* src, int w, int h, int sstride, int channels, int rgb_order, uint8_t* dst, int dstride)
{
    cvt_gray_u8(src, w, h, sstride, channels, rgb_order, dst, dstride);
}

// Tracking.cc:271-272 — imDepth.convertTo(CV_32F, mDepthMapFactor) for a 16-bit depth image.
void orc_depth_to_f32(const uint16_t* src, int w, int h, int sstride_elems, float factor, float* dst)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) dst[(size_t)y * w + x] = (float)src[(size_t)y * sstride_elems + x] * factor;
}

// Frame::ComputeStereoFromRGBD (Frame.cc:1051-1072); mvKeysUn == mvKeys (k1 == 0, Frame.cc:814-818)
void orc_stereo_from_rgbd(const void* kps_, int N, const float* depth, int w, int /*h*/, float mbf, float* uRight,
                          float* mvDepth)
{
    const KeyPoint* kps = (const KeyPoint*)kps_;
    for (int i = 0; i < N; i++) {
        uRight[i] = -1; mvDepth[i] = -1;
        const float v = kps[i].y, u = kps[i].x;
        const float d = depth[(size_t)(int)v * w + (int)u];
        if (d > 0) { mvDepth[i] = d; uRight[i] = kps[i].x - mbf / d; }
    }
}

// Frame::ComputeStereoMatches (Frame.cc:874-1048).  hL/hR: extractors that processed the
// left/right image (their mvImagePyramid is read at :971,:988).  Returns #matched before
// the median filter; bestDist_out (optional) gets the SAD distance per left kp (-1 none).
int orc_stereo_matches(void* hL, void* hR, const void* kL_, const uint8_t* dL, int N, const void* kR_, const uint8_t* dR,
                       int Nr, float mbf, float fx, float* mvuRight, float* mvDepth, int* bestDist_out)
{
    Extractor* eL = (Extractor*)hL; Extractor* eR = (Extractor*)hR;
    const KeyPoint* mvKeys = (const KeyPoint*)kL_; const KeyPoint* mvKeysRight = (const KeyPoint*)kR_;
    for (int i = 0; i < N; i++) { mvuRight[i] = -1.0f; mvDepth[i] = -1.0f; if (bestDist_out) bestDist_out[i] = -1; }
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
    const int nRows = eL->mvImagePyramid[0].H;
    std::vector<std::vector<size_t> > vRowIndices(nRows);
    const std::vector<float>& mvScaleFactors = eL->mvScaleFactor;
    const std::vector<float>& mvInvScaleFactors = eL->mvInvScaleFactor;
    for (int iR = 0; iR < Nr; iR++) {
        const KeyPoint& kp = mvKeysRight[iR];
        const float kpY = kp.y;
        const float r = 2.0f * mvScaleFactors[kp.octave];
        const int maxr = (int)ceilf(kpY + r);
        const int minr = (int)floorf(kpY - r);
        for (int yi = minr; yi <= maxr; yi++)
            if (yi >= 0 && yi < nRows) vRowIndices[yi].push_back(iR);   // reference: unchecked (quirk C10)
    }
    const float mb = mbf / fx;
    const float minZ = mb;
    const float minD = 0;
    const float maxD = mbf / minZ;
    std::vector<std::pair<int, int> > vDistIdx;
    for (int iL = 0; iL < N; iL++) {
        const KeyPoint& kpL = mvKeys[iL];
        const int levelL = kpL.octave;
        const float vL = kpL.y, uL = kpL.x;
        const std::vector<size_t>& vCandidates = vRowIndices[(size_t)vL];
        if (vCandidates.empty()) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = TH_HIGH;
        size_t bestIdxR = 0;
        const uint8_t* dl = dL + (size_t)iL * 32;
        for (size_t iC = 0; iC < vCandidates.size(); iC++) {
            const size_t iR = vCandidates[iC];
            const KeyPoint& kpR = mvKeysRight[iR];
            if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
            const float uR = kpR.x;
            if (uR >= minU && uR <= maxU) {
                const int dist = DescriptorDistance(dl, dR + iR * 32);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        if (bestDist < thOrbDist) {
            const float uR0 = mvKeysRight[bestIdxR].x;
            const float scaleFactor = mvInvScaleFactors[kpL.octave];
            const float scaleduL = roundf(kpL.x * scaleFactor);
            const float scaledvL = roundf(kpL.y * scaleFactor);
            const float scaleduR0 = roundf(uR0 * scaleFactor);
            const int w = 5;
            const Plane& PL = eL->mvImagePyramid[kpL.octave];
            const Plane& PR = eR->mvImagePyramid[kpL.octave];
            const uint8_t* IL = PL.interior() + (ptrdiff_t)((int)(scaledvL - w)) * PL.stride + (int)(scaleduL - w);
            const float cL = (float)IL[w * PL.stride + w];
            int bestDistS = INT_MAX;
            int bestincR = 0;
            const int L = 5;
            float vDists[2 * 5 + 1];
            const float iniu = scaleduR0 + L - w;
            const float endu = scaleduR0 + L + w + 1;
            if (iniu < 0 || endu >= PR.W) continue;
            for (int incR = -L; incR <= +L; incR++) {
                const uint8_t* IR = PR.interior() + (ptrdiff_t)((int)(scaledvL - w)) * PR.stride + (int)(scaleduR0 + incR - w);
                const float cR = (float)IR[w * PR.stride + w];
                double s = 0;   // cv::norm(IL, IR, NORM_L1) accumulates |a-b| of f32 values in double
                for (int yy = 0; yy < 2 * w + 1; yy++)
                    for (int xx = 0; xx < 2 * w + 1; xx++) {
                        float a = (float)IL[yy * PL.stride + xx] - cL, b = (float)IR[yy * PR.stride + xx] - cR;
                        s += std::abs(a - b);
                    }
                float dist = (float)s;
                if (dist < bestDistS) { bestDistS = (int)dist; bestincR = incR; }
                vDists[L + incR] = dist;
            }
            if (bestincR == -L || bestincR == L) continue;
            const float dist1 = vDists[L + bestincR - 1];
            const float dist2 = vDists[L + bestincR];
            const float dist3 = vDists[L + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = mvScaleFactors[kpL.octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
            float disparity = (uL - bestuR);
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) { disparity = 0.01; bestuR = uL - 0.01; }
                mvDepth[iL] = mbf / disparity;
                mvuRight[iL] = bestuR;
                vDistIdx.push_back(std::pair<int, int>(bestDistS, iL));
                if (bestDist_out) bestDist_out[iL] = bestDistS;
            }
        }
    }
    int nm = (int)vDistIdx.size();
    if (vDistIdx.empty()) return 0;   // reference: UB at Frame.cc:1035 (quirk C10)
    std::sort(vDistIdx.begin(), vDistIdx.end());
    const float median = (float)vDistIdx[vDistIdx.size() / 2].first;
    const float thDist = 1.5f * 1.4f * median;
    for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
        if (vDistIdx[i].first < thDist) break;
        mvuRight[vDistIdx[i].second] = -1;
        mvDepth[vDistIdx[i].second] = -1;
    }
    return nm;
}

#include "frame_oracle.inc"
#include "cull_oracle.inc"
#include "bow_oracle.inc"
#include "motion_oracle.inc"

} // extern "C"
