// C-ABI implementation of include/sd_frontend.h: host orchestration of the HIP kernels.
// There is no CPU fallback anywhere in this file: without a usable HIP device every compute
// entry point returns SD_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstddef>
#include <cmath>
#include <string>
#include <vector>
#include "k_extract.h"
#include "k_frame.h"
#include "k_fast.h"
#include "k_motion.h"
#include "k_cull.h"
#include "k_tracker.h"
#include "k_cloud.h"
#include "sd_common.h"
#include "sd_vocab.h"

static thread_local std::string g_err;
int sd_set_err(int code, const std::string& msg) { g_err = msg; return code; }

struct sd_extractor {
    SdParams prm;
};

enum KernelId { K_PYR0, K_PYR, K_FAST, K_QTREE, K_ORIENT, K_BLUR, K_DESC, K_STEREO, K_STEREO_F, K_RGBD, K_GRID, K_UNPROJ, K_PROJ_A, K_PROJ_B, K_BOXSEP, K_SEPARATE, K_UPDATE, K_LOCAL_A, K_LOCAL_B, K_BOW_T, K_BOW_F, K_BOW_S, K_MOTION_P, K_MOTION_H, K_MOTION_S, K_COUNT };
static const char* kKernelNames[K_COUNT] = {"k_pyr_level0", "k_pyr_level", "k_fast_cells", "k_quadtree", "k_orient",
                                            "k_blur", "k_describe", "k_stereo_match", "k_stereo_filter", "k_rgbd",
                                            "k_grid_cells", "k_unproject", "k_proj_candidates", "k_proj_resolve",
                                            "k_box_separate", "k_separate", "k_update_frame", "k_local_candidates", "k_local_resolve", "k_bow_transform", "k_bow_finalize", "k_search_by_bow", "k_motion_prepare", "k_motion_hyp", "k_motion_select"};

#define SD_PT_TW 256
#define SD_PT_TH 16
struct SdPyrTiles { int tilesX = 0, tilesY = 0, srcRowBytes = 0, srcRowsMax = 0, extOff = 0; size_t lds = 0; bool use = false; };

struct sd_batch {
    std::vector<SdPyrTiles> pyrTiles;      // per level: tile grid of k_pyr_level_tiles
    int* d_pyrExt = nullptr;
    int* d_blurTiles = nullptr; int nBlurTiles = 0;      // k_blur_wide: tile list of one image
    sd_extractor* ex = nullptr;
    SdPlan plan;
    SdDevPlan hplan;
    int maxImages = 0;
    int nExtracted = 0;       // images processed by the last extract
    std::vector<uint8_t> slotValid;   // slot holds frame results (extracted or carried over)
    int2* d_pairIdx = nullptr;
    int nStereo = 0;
    hipStream_t stream = nullptr;
    hipStream_t lastStream = nullptr;
    // device buffers
    SdDevPlan* d_plan = nullptr;
    SdCell* d_cells = nullptr;
    SdFastCell* d_fcells = nullptr; int fastListCap = 0, fastMaxCap = 0;   // k_fast_cells_staged: per-cell descriptors, list sizes
    int16_t* d_tabs = nullptr;
    uint8_t* d_pyr = nullptr;
    uint8_t* d_blur = nullptr;
    uint32_t* d_cellList = nullptr;
    int* d_cellCount = nullptr;
    uint32_t* d_cand = nullptr;
    uint16_t* d_nodeOf = nullptr;
    int* d_lvlCount = nullptr;
    int* d_candCount = nullptr;
    uint32_t* d_lvlKp = nullptr;
    float2* d_rot = nullptr;
    sd_keypoint* d_kp = nullptr;
    sd_keypoint* d_kpUn = nullptr; sd_keypoint* d_kpDUn = nullptr; SdDistortion dist; bool hasDist = false; int* d_unSlots = nullptr;   // mvKeysUn / mvdynKeysUn (sd_batch_set_distortion)
    uint8_t* d_desc = nullptr;
    int* d_count = nullptr;
    int* d_err = nullptr;
    bool cullOk = true;              // the per-key-point LDS tables of k_box_separate / k_separate fit this workspace's capacity
    float* d_uright = nullptr;
    float* d_depth = nullptr;
    int* d_sad = nullptr;
    unsigned short* d_rowIdx = nullptr;   // right keypoints bucketed by row (stereo)
    int* d_rowStart = nullptr;
    short* d_cellOf = nullptr;      // grid cell of every keypoint
    unsigned short* d_sortedIdx = nullptr;   // keypoint indices sorted by (cell, index)
    unsigned short* d_cellStart = nullptr;   // [maxImages][3072 + 8]
    int gridSortN = 0;
    float* d_xw = nullptr;          // map-point world positions [maxImages][cap][3]
    uint8_t* d_flags = nullptr;     // bit0: has map point (not outlier); bit1: Observations() > 0
    unsigned short* d_pcand = nullptr;
    uint8_t* d_pncand = nullptr;
    // bag of words (per image): per-feature word / weight / node, FeatureVector (sorted) + runs, BowVector
    unsigned* d_bowWordF = nullptr; double* d_bowWF = nullptr; unsigned* d_bowNidF = nullptr; unsigned* d_fvNode = nullptr; unsigned* d_fvFeat = nullptr;
    int* d_fvRunStart = nullptr; unsigned* d_fvRunNode = nullptr; unsigned* d_bowWord = nullptr; double* d_bowVal = nullptr; int* d_bowMeta = nullptr;
    int* d_bowImg = nullptr; std::vector<uint8_t> bowValid;
    float* d_moPts = nullptr; SdMotionNorm* d_moNorm = nullptr; int* d_moCounts = nullptr; double* d_moModels = nullptr; uint8_t* d_moMaskH = nullptr; uint8_t* d_moMaskF = nullptr;
    SdMotionResult* d_moRes = nullptr; int nMotion = 0;      // TrackHomo model fit
    unsigned* d_lmCand = nullptr; uint8_t* d_lmN = nullptr; uint8_t* d_lmOvf = nullptr; int* d_lmIdx = nullptr; int lmCap = 0;   // local-map search scratch
    int* d_match = nullptr;
    int* d_pairs = nullptr;
    int* d_npairs = nullptr;
    int* d_nmatch = nullptr;
    float* d_pose = nullptr;        // staging for host poses: [2][maxImages][16]
    int nPairs = 0;
    int dlPairs = 0;          // pairs sd_batch_download_matches may read (the tracker also keeps pairs at [n_lanes, 2 * n_lanes))
    // dynamic-object cull
    SdFrameBoxes* d_fb = nullptr;
    SdFrameBoxes* d_fbStage = nullptr;     // one upload per sd_batch_first_separate call
    int* d_boxItems = nullptr;
    sd_keypoint* d_kpT = nullptr; uint8_t* d_descT = nullptr; float* d_urT = nullptr; float* d_depT = nullptr;
    sd_keypoint* d_kpD = nullptr; uint8_t* d_descD = nullptr; float* d_urD = nullptr; float* d_depD = nullptr;
    int* d_slots = nullptr;
    float* d_HorF = nullptr; int* d_sepFlag = nullptr; int* d_lastIdx = nullptr; int* d_lastStatus = nullptr; int* d_nLast = nullptr;
    int* d_dynStart = nullptr; int* d_dynStatus = nullptr; int* d_sepMatches = nullptr; int* d_sepRet = nullptr;
    int2* d_sepPairs = nullptr;
    int2* d_copyPairs = nullptr;
    unsigned long long* d_cloudBits = nullptr; int* d_cloudRows = nullptr; double* d_cloudT = nullptr; int* d_cloudSlots = nullptr; size_t cloudCap = 0;
    int itemsCap = 0;
    int nSepPairs = 0;
    const int* sepActive = nullptr;          // active mask of the last separate (tracker mode), applied by update_frame too
    std::vector<SdFrameBoxes> hostBoxes;     // staging that must outlive the async uploads
    std::vector<int2> hostPairs, hostCopyPairs;
    uint8_t* d_stage = nullptr;    // staging for host-image uploads
    size_t stageBytes = 0;
    int qtMN = 0, qtSortP = 0;
    size_t qtLds = 0;
    // profiling
    bool profiling = false;
    struct Rec { hipEvent_t a, b; int kid; };
    std::vector<Rec> pending;
    std::vector<hipEvent_t> pool;
    double totalMs[K_COUNT] = {0};
    int64_t launches[K_COUNT] = {0};
};

// mvKeysUn of the batch: the key points themselves unless a distortion was set (Frame.cc:814-818)
#define KPUN(b) ((b)->hasDist ? (b)->d_kpUn : (b)->d_kp)
#define KPDUN(b) ((b)->hasDist ? (b)->d_kpDUn : (b)->d_kpD)

extern "C" {

int sd_version(void) { return 100; }

const char* sd_status_string(int s)
{
    switch (s) {
    case SD_OK: return "ok";
    case SD_ERR_INVALID: return "invalid argument";
    case SD_ERR_NO_DEVICE: return "no HIP device";
    case SD_ERR_HIP: return "HIP error";
    case SD_ERR_CAPACITY: return "buffer too small";
    case SD_ERR_UNSUPPORTED: return "unsupported geometry";
    case SD_ERR_STATE: return "call sequence error";
    default: return "unknown";
    }
}

const char* sd_last_error(void) { return g_err.c_str(); }

int sd_device_count(int* n)
{
    if (!n) return SD_ERR_INVALID;
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess || c <= 0) { *n = 0; return set_err(SD_ERR_NO_DEVICE, "hipGetDeviceCount: no device"); }
    *n = c;
    return SD_OK;
}

int sd_extractor_create(sd_extractor** out, int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
{
    if (!out) return SD_ERR_INVALID;
    *out = nullptr;
    if (nfeatures < 1 || nlevels < 1 || nlevels > SD_MAX_LEVELS || !(scaleFactor > 1.0f) || iniThFAST < 1 ||
        iniThFAST > 255 || minThFAST < 1 || minThFAST > 255)
        return set_err(SD_ERR_INVALID, "extractor parameters out of range");
    sd_extractor* ex = new sd_extractor();
    sd_params_init(ex->prm, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST);
    *out = ex;
    return SD_OK;
}

int sd_extractor_destroy(sd_extractor* ex) { delete ex; return SD_OK; }

int sd_extractor_set_blur_taps(sd_extractor* ex, const uint16_t taps[7])
{
    if (!ex || !taps) return SD_ERR_INVALID;
    unsigned sum = 0;
    for (int i = 0; i < 7; i++) sum += taps[i];
    if (sum > 257) return set_err(SD_ERR_INVALID, "blur taps must sum to <= 257 (8.8 fixed point)");
    for (int i = 0; i < 7; i++) if (taps[i] > 255) return set_err(SD_ERR_INVALID, "each blur tap must be <= 255");
    memcpy(ex->prm.blurTaps, taps, 14);
    return SD_OK;
}

int sd_extractor_levels(const sd_extractor* ex, int* nlevels, float* scale_factor)
{
    if (!ex) return SD_ERR_INVALID;
    if (nlevels) *nlevels = ex->prm.nlevels;
    if (scale_factor) *scale_factor = (float)ex->prm.scaleFactor;
    return SD_OK;
}

int sd_extractor_tables(const sd_extractor* ex, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                        int32_t* quota, int32_t* umax)
{
    if (!ex) return SD_ERR_INVALID;
    const SdParams& p = ex->prm;
    for (int i = 0; i < p.nlevels; i++) {
        if (scale) scale[i] = p.scale[i];
        if (inv_scale) inv_scale[i] = p.inv[i];
        if (sigma2) sigma2[i] = p.sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = p.invSigma2[i];
        if (quota) quota[i] = p.quota[i];
    }
    if (umax) for (int i = 0; i < 16; i++) umax[i] = p.umax[i];
    return SD_OK;
}

int sd_extractor_level_size(const sd_extractor* ex, int width, int height, int level, int* lw, int* lh)
{
    if (!ex || level < 0 || level >= ex->prm.nlevels || width < 1 || height < 1) return SD_ERR_INVALID;
    if (lw) *lw = sd_cvRoundf((float)width * ex->prm.inv[level]);
    if (lh) *lh = sd_cvRoundf((float)height * ex->prm.inv[level]);
    return SD_OK;
}

static void batch_free(sd_batch* b)
{
    if (!b) return;
    void* ptrs[] = {b->d_plan, b->d_cells, b->d_fcells, b->d_blurTiles, b->d_tabs, b->d_pyr, b->d_blur, b->d_cellList, b->d_cellCount, b->d_cand,
                    b->d_nodeOf, b->d_lvlCount, b->d_candCount, b->d_lvlKp, b->d_rot, b->d_kp, b->d_desc, b->d_count,
                    b->d_err, b->d_uright, b->d_depth, b->d_sad, b->d_stage, b->d_cellOf, b->d_xw, b->d_flags,
                    b->d_pcand, b->d_pncand, b->d_match, b->d_pairs, b->d_npairs, b->d_nmatch, b->d_pose, b->d_pairIdx, b->d_sortedIdx, b->d_cellStart,
                    b->d_fb, b->d_fbStage, b->d_boxItems, b->d_kpT, b->d_descT, b->d_urT, b->d_depT, b->d_slots, b->d_HorF, b->d_sepFlag,
                    b->d_lastIdx, b->d_lastStatus, b->d_nLast, b->d_dynStart, b->d_dynStatus, b->d_sepMatches, b->d_sepRet,
                    b->d_sepPairs, b->d_kpD, b->d_descD, b->d_urD, b->d_depD, b->d_rowIdx, b->d_rowStart,
                    b->d_lmCand, b->d_lmN, b->d_lmOvf, b->d_lmIdx, b->d_bowWordF, b->d_bowWF, b->d_bowNidF, b->d_fvNode, b->d_fvFeat,
                    b->d_fvRunStart, b->d_fvRunNode, b->d_bowWord, b->d_bowVal, b->d_bowMeta, b->d_bowImg,
                    b->d_moPts, b->d_moNorm, b->d_moCounts, b->d_moModels, b->d_moMaskH, b->d_moMaskF, b->d_moRes, b->d_pyrExt, b->d_copyPairs, b->d_kpUn, b->d_kpDUn, b->d_unSlots, b->d_cloudBits, b->d_cloudRows, b->d_cloudT, b->d_cloudSlots};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (auto& r : b->pending) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : b->pool) (void)hipEventDestroy(e);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
}

static int batch_create_impl(sd_batch** out, sd_extractor* ex, int width, int height, int max_images, int minKpCap);
int sd_batch_create(sd_batch** out, sd_extractor* ex, int width, int height, int max_images) { return batch_create_impl(out, ex, width, height, max_images, 0); }

// minKpCap: row stride of the per-image result arrays at least this (a tracker whose lanes switch between two extractors)
static int batch_create_impl(sd_batch** out, sd_extractor* ex, int width, int height, int max_images, int minKpCap)
{
    if (!out) return SD_ERR_INVALID;
    *out = nullptr;
    if (!ex || width < 1 || height < 1 || max_images < 1) return set_err(SD_ERR_INVALID, "bad batch arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return set_err(SD_ERR_NO_DEVICE, "no HIP device: the front end has no CPU fallback");
    sd_batch* b = new sd_batch();
    b->ex = ex;
    b->maxImages = max_images;
    if (!sd_plan_build(b->plan, ex->prm, width, height, minKpCap)) {
        std::string e = b->plan.error;
        delete b;
        return set_err(SD_ERR_UNSUPPORTED, e);
    }
    const SdPlan& P = b->plan;
    SdDevPlan& D = b->hplan;
    memset(&D, 0, sizeof(D));
    for (int l = 0; l < P.nlevels; l++) D.lv[l] = P.lv[l];
    D.nlevels = P.nlevels; D.iniTh = ex->prm.iniTh; D.minTh = ex->prm.minTh;
    D.cellTotal = (int)P.cells.size(); D.cellListCap = P.cellListCap;
    D.kpCapLevels = P.kpCapLevels; D.kpCap = P.kpCap;
    D.pyrImageBytes = P.pyrImageBytes; D.blurImageBytes = P.blurImageBytes;
    for (int i = 0; i < 16; i++) D.umax[i] = ex->prm.umax[i];
    for (int i = 0; i < 7; i++) D.taps[i] = ex->prm.blurTaps[i];
    // quadtree LDS sizing: list capacity L (quota + 3, the initial nodes, or the cell count of a level)
    int MN = 0;
    for (int l = 0; l < P.nlevels; l++) {
        MN = std::max(MN, P.lv[l].kpCap + 8);
        MN = std::max(MN, P.lv[l].nCells + 16);
    }
    MN = (MN + 7) & ~7;
    int sortP = 1;
    while (sortP < MN) sortP <<= 1;
    size_t lds = (size_t)sortP * 8 + (size_t)MN * (8 + 8 + 4 * 4 + 16 * 2 + 2 * 3) + 64;
    if (lds > 160 * 1024 - 256 || MN > 30000)
        { delete b; return set_err(SD_ERR_UNSUPPORTED, "per-level feature quota too large for the LDS quadtree (nfeatures too high)"); }
    b->qtMN = MN; b->qtSortP = sortP; b->qtLds = lds;

#define ALLOC(ptr, bytes)                                                                     \
    do {                                                                                      \
        hipError_t e_ = hipMalloc((void**)&(ptr), (bytes));                                   \
        if (e_ != hipSuccess) {                                                               \
            std::string m = std::string("hipMalloc(" #ptr "): ") + hipGetErrorString(e_);     \
            batch_free(b);                                                                    \
            return set_err(SD_ERR_HIP, m);                                                    \
        }                                                                                     \
    } while (0)
    const size_t nI = (size_t)max_images;
    ALLOC(b->d_plan, sizeof(SdDevPlan));
    ALLOC(b->d_cells, sizeof(SdCell) * P.cells.size());
    ALLOC(b->d_fcells, sizeof(SdFastCell) * P.cells.size());
    ALLOC(b->d_tabs, sizeof(int16_t) * P.tabs.size());
    ALLOC(b->d_pyr, nI * P.pyrImageBytes + 4096);
    ALLOC(b->d_blur, nI * P.blurImageBytes + 4096);
    ALLOC(b->d_cellList, nI * P.cellListCap * 4 + 64);
    ALLOC(b->d_cellCount, nI * P.cells.size() * 4);
    ALLOC(b->d_cand, nI * P.cellListCap * 4 + 64);
    ALLOC(b->d_nodeOf, nI * P.cellListCap * 2 + 64);
    ALLOC(b->d_lvlCount, nI * P.nlevels * 4 + SD_MAX_LEVELS * 4);      // + padding: sd_level_counts loads SD_MAX_LEVELS entries
    ALLOC(b->d_candCount, nI * P.nlevels * 4);
    ALLOC(b->d_lvlKp, nI * P.kpCapLevels * 4);
    ALLOC(b->d_rot, nI * P.kpCapLevels * sizeof(float2));
    ALLOC(b->d_kp, nI * P.kpCap * sizeof(sd_keypoint));
    ALLOC(b->d_desc, nI * P.kpCap * 32);
    ALLOC(b->d_count, nI * 4);
    ALLOC(b->d_err, 4);
    ALLOC(b->d_uright, nI * P.kpCap * 4);
    ALLOC(b->d_depth, nI * P.kpCap * 4);
    ALLOC(b->d_sad, nI * P.kpCap * 4);
    ALLOC(b->d_rowIdx, nI * P.kpCap * 2);
    ALLOC(b->d_rowStart, nI * (size_t)(P.lv[0].H + 8) * 4);
    ALLOC(b->d_cellOf, nI * P.kpCap * 2);
    ALLOC(b->d_sortedIdx, nI * P.kpCap * 2);
    ALLOC(b->d_cellStart, nI * (SD_GRID_CELLS + 8) * 2);
    { int sn = 1; while (sn < P.kpCap) sn <<= 1; b->gridSortN = sn; }
    ALLOC(b->d_xw, nI * P.kpCap * 12);
    ALLOC(b->d_flags, nI * P.kpCap);
    ALLOC(b->d_pcand, nI * P.kpCap * SD_PROJ_K * 2);
    ALLOC(b->d_pncand, nI * P.kpCap);
    ALLOC(b->d_match, nI * P.kpCap * 4);
    ALLOC(b->d_pairs, nI * P.kpCap * 8);
    ALLOC(b->d_npairs, nI * 4);
    ALLOC(b->d_nmatch, nI * 4);
    ALLOC(b->d_pose, nI * 2 * 16 * 4);
    b->itemsCap = 2 * P.kpCap;
    ALLOC(b->d_fb, nI * sizeof(SdFrameBoxes));
    ALLOC(b->d_fbStage, nI * sizeof(SdFrameBoxes));
    ALLOC(b->d_boxItems, nI * b->itemsCap * 4);
    ALLOC(b->d_kpT, nI * P.kpCap * sizeof(sd_keypoint));
    ALLOC(b->d_descT, nI * P.kpCap * 32);
    ALLOC(b->d_urT, nI * P.kpCap * 4);
    ALLOC(b->d_depT, nI * P.kpCap * 4);
    ALLOC(b->d_kpD, nI * P.kpCap * sizeof(sd_keypoint));
    ALLOC(b->d_descD, nI * P.kpCap * 32);
    ALLOC(b->d_urD, nI * P.kpCap * 4);
    ALLOC(b->d_depD, nI * P.kpCap * 4);
    ALLOC(b->d_slots, nI * 4);
    ALLOC(b->d_HorF, nI * 9 * 4);
    ALLOC(b->d_sepFlag, nI * 4);
    ALLOC(b->d_lastIdx, nI * SD_MAXB * 4);
    ALLOC(b->d_lastStatus, nI * SD_MAXB * 4);
    ALLOC(b->d_nLast, nI * 4);
    ALLOC(b->d_dynStart, nI * (SD_MAXB + 1) * 4);
    ALLOC(b->d_dynStatus, nI * b->itemsCap * 4);
    ALLOC(b->d_sepMatches, nI * b->itemsCap * 8);
    ALLOC(b->d_sepRet, nI * 4);
    ALLOC(b->d_sepPairs, nI * sizeof(int2));
    ALLOC(b->d_copyPairs, nI * sizeof(int2));
    ALLOC(b->d_pairIdx, nI * sizeof(int2));
    b->slotValid.assign(nI, 0);
#undef ALLOC
    hipError_t e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMemcpy(b->d_plan, &D, sizeof(D), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(b->d_cells, P.cells.data(), sizeof(SdCell) * P.cells.size(), hipMemcpyHostToDevice);
    {
        std::vector<SdFastCell> fc(P.cells.size());
        for (size_t i = 0; i < P.cells.size(); i++) {
            const SdCell& c = P.cells[i];
            const SdLevel& g = P.lv[c.level];
            fc[i].srcOff = (uint32_t)(g.pyrOffset + (SD_EDGE + c.y0) * g.stride + SD_XOFF + c.x0 - 1);
            fc[i].stride = g.stride;
            fc[i].ww = (short)(c.x1 - c.x0); fc[i].wh = (short)(c.y1 - c.y0);
            fc[i].jw = c.jw; fc[i].ih = c.ih; fc[i].listOffset = c.listOffset; fc[i].cap = c.cap;
            b->fastListCap = std::max(b->fastListCap, (c.x1 - c.x0 - 6) * (c.y1 - c.y0 - 6));      // largest scanned area of a cell
            b->fastMaxCap = std::max(b->fastMaxCap, c.cap);
        }
        b->fastListCap = (b->fastListCap + 7) & ~7;
        if (e == hipSuccess) e = hipMemcpy(b->d_fcells, fc.data(), sizeof(SdFastCell) * fc.size(), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMemcpy(b->d_tabs, P.tabs.data(), sizeof(int16_t) * P.tabs.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(b->d_err, 0, 4);
    if (e == hipSuccess) e = hipMemset(b->d_fb, 0, nI * sizeof(SdFrameBoxes));
    // the dynamic-object kernels keep per-key-point tables in LDS: sized by this workspace's capacity, the limit only ever raised (workspaces of
    // different capacities share the kernels: a monocular tracker's 2 x nFeatures initialisation workspace beside the regular one)
    {
        static int ldsSeparate = 0, ldsBoxSeparate = 0;
        const int needSep = (int)sd_separate_lds(P.kpCap), needBox = (int)sd_box_separate_lds(P.kpCap);
        b->cullOk = needSep <= 160 * 1024 && needBox <= 160 * 1024;       // otherwise sd_batch_first_separate / sd_batch_separate refuse (extraction and matching are not affected)
        if (b->cullOk && e == hipSuccess && needSep > ldsSeparate) { e = hipFuncSetAttribute((const void*)k_separate, hipFuncAttributeMaxDynamicSharedMemorySize, needSep); ldsSeparate = needSep; }
        if (b->cullOk && e == hipSuccess && needBox > ldsBoxSeparate && needBox > 64 * 1024) { e = hipFuncSetAttribute((const void*)k_box_separate, hipFuncAttributeMaxDynamicSharedMemorySize, needBox); ldsBoxSeparate = needBox; }
    }
    if (e == hipSuccess) e = hipMemset(b->d_count, 0, nI * 4);
    if (e == hipSuccess) e = hipMemset(b->d_lvlCount, 0, nI * P.nlevels * 4);
    if (e == hipSuccess && lds > 64 * 1024)
        e = hipFuncSetAttribute((const void*)k_quadtree, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
        std::string m = std::string("batch setup: ") + hipGetErrorString(e);
        batch_free(b);
        return set_err(SD_ERR_HIP, m);
    }
    b->lastStream = b->stream;
    *out = b;
    return SD_OK;
}

int sd_batch_destroy(sd_batch* b)
{
    if (b) { (void)hipDeviceSynchronize(); batch_free(b); }
    return SD_OK;
}

int sd_batch_kp_capacity(const sd_batch* b, int* cap)
{
    if (!b || !cap) return SD_ERR_INVALID;
    *cap = b->plan.kpCap;
    return SD_OK;
}

// ---- profiling helpers: hipEvents on the stream the kernel is launched on
static hipEvent_t get_event(sd_batch* b)
{
    if (!b->pool.empty()) { hipEvent_t e = b->pool.back(); b->pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
struct ProfScope {
    sd_batch* b; hipStream_t s; int kid; hipEvent_t a, e;
    ProfScope(sd_batch* b_, hipStream_t s_, int kid_) : b(b_), s(s_), kid(kid_)
    {
        if (b->profiling) { a = get_event(b); e = get_event(b); (void)hipEventRecord(a, s); }
    }
    ~ProfScope()
    {
        if (b->profiling) { (void)hipEventRecord(e, s); b->pending.push_back({a, e, kid}); }
    }
};
static void drain_profile(sd_batch* b)
{
    for (auto& r : b->pending) {
        float ms = 0;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            b->totalMs[r.kid] += ms;
            b->launches[r.kid]++;
        }
        b->pool.push_back(r.a); b->pool.push_back(r.b);
    }
    b->pending.clear();
}

static bool slot_ok(const sd_batch* b, int image) { return image >= 0 && image < b->maxImages && b->slotValid[image]; }


static int extract_impl(sd_batch* b, const uint8_t* d_gray, size_t stride, size_t image_pitch, int n_images, void* stream_, int colorMode);

int sd_batch_extract_device(sd_batch* b, const uint8_t* d_gray, size_t stride, size_t image_pitch, int n_images,
                            void* stream_)
{
    return extract_impl(b, d_gray, stride, image_pitch, n_images, stream_, 0);
}

int sd_batch_extract_color_device(sd_batch* b, const uint8_t* d_src, size_t stride, size_t image_pitch, int rgb_order, int n_images,
                                  void* stream_)
{
    return extract_impl(b, d_src, stride, image_pitch, n_images, stream_, rgb_order ? 2 : 1);
}

int sd_batch_extract_pixels_device(sd_batch* b, const uint8_t* d_src, size_t stride, size_t image_pitch, int channels, int rgb_order,
                                   int n_images, void* stream_)
{
    if (channels != 1 && channels != 3 && channels != 4) return set_err(SD_ERR_INVALID, "images must have 1, 3 or 4 channels (Tracking.cc:175-200)");
    return extract_impl(b, d_src, stride, image_pitch, n_images, stream_, channels == 1 ? 0 : (channels == 3 ? 1 : 3) + (rgb_order ? 1 : 0));
}

// colorMode 0: 8-bit gray input; 1 / 2: 3-channel BGR / RGB, 3 / 4: 4-channel BGRA / RGBA input converted on the way into level 0
static int extract_impl(sd_batch* b, const uint8_t* d_gray, size_t stride, size_t image_pitch, int n_images, void* stream_, int colorMode)
{
    if (!b || n_images < 0 || n_images > b->maxImages) return set_err(SD_ERR_INVALID, "bad extract arguments");
    hipStream_t s = stream_ ? (hipStream_t)stream_ : b->stream;
    b->lastStream = s;
    b->nExtracted = 0;
    b->nStereo = 0;
    if (n_images == 0) return SD_OK;
    if (!d_gray) return set_err(SD_ERR_INVALID, "null image pointer");
    const SdPlan& P = b->plan;
    const int bpp = colorMode == 0 ? 1 : (colorMode <= 2 ? 3 : 4), rgbOrder = (colorMode == 2 || colorMode == 4) ? 1 : 0;
    if (stride < (size_t)P.W * bpp) return set_err(SD_ERR_INVALID, "stride smaller than width");
    const int nl = P.nlevels;
    {
        ProfScope ps(b, s, K_PYR0);
        const SdLevel& g = P.lv[0];
        {
            // interior 16-byte groups flattened over (row, group) so that every wave is full + a small kernel for the groups on the frame
            const int gpr = g.W / 16;                                                    // groups fully inside the interior
            const int lastGroup = (g.W + SD_EDGE - 1) / 16;                              // group of the last frame column
            const int tail = lastGroup - gpr + 1, rows = g.H + 2 * SD_EDGE;
            const dim3 gi((unsigned)((size_t)rows * gpr + 255) / 256, n_images), gf((unsigned)(rows * (2 + tail) + 255) / 256, n_images);
            if (gpr > 0) {
                const uint32_t gprInv = 0xFFFFFFFFu / (uint32_t)gpr + 1u;
                if (bpp == 4) hipLaunchKernelGGL(k_pyr_level0_rgba, gi, dim3(256), 0, s, d_gray, stride, image_pitch, rgbOrder, b->d_pyr, b->d_plan, gpr, gprInv);
                else if (bpp == 3) hipLaunchKernelGGL(k_pyr_level0_rgb, gi, dim3(256), 0, s, d_gray, stride, image_pitch, rgbOrder, b->d_pyr, b->d_plan, gpr, gprInv);
                else hipLaunchKernelGGL(k_pyr_level0_gray, gi, dim3(256), 0, s, d_gray, stride, image_pitch, b->d_pyr, b->d_plan, gpr, gprInv);
            }
            if (colorMode) hipLaunchKernelGGL(k_pyr_level0_rgb_frame, gf, dim3(256), 0, s, d_gray, stride, image_pitch, rgbOrder, b->d_pyr, b->d_plan, gpr, tail, bpp);
            else hipLaunchKernelGGL(k_pyr_level0_gray_frame, gf, dim3(256), 0, s, d_gray, stride, image_pitch, b->d_pyr, b->d_plan, gpr, tail);
        }
    }
    LAUNCH_CHECK("k_pyr_level0");
    if (b->pyrTiles.empty()) {          // once per batch: LDS extents of k_pyr_level_tiles per level, or the per-thread kernel as a fallback
        b->pyrTiles.assign(nl, SdPyrTiles());
        auto refl = [](int p, int len) { if (p < 0) p = -p; if (p >= len) p = 2 * (len - 1) - p; return p; };
        std::vector<int> ext;             // per level: source column origin of every tile column, (row origin, rows) of every tile row
        for (int l = 1; l < nl; l++) {
            const SdLevel& g = P.lv[l];
            const int16_t* ct = &P.tabs[4 * (size_t)g.tabOffset];
            const int16_t* rt = ct + 4 * (size_t)g.W;
            const int PW = g.W + 2 * SD_EDGE, HPl = g.H + 2 * SD_EDGE, sH = P.lv[l - 1].H;
            SdPyrTiles t;
            t.tilesX = (PW + SD_PT_XSHIFT + SD_PT_TW - 1) / SD_PT_TW; t.tilesY = (HPl + SD_PT_TH - 1) / SD_PT_TH;
            t.extOff = (int)ext.size();
            int spanX = 0, spanY = 0;
            for (int tx = 0; tx < t.tilesX; tx++) {
                int lo = 1 << 30, hi = -1;
                for (int k = 0; k < SD_PT_TW; k++) {
                    const int sx = ct[4 * refl(std::min(std::max(tx * SD_PT_TW - SD_PT_XSHIFT + k, 0), PW - 1) - SD_EDGE, g.W)];
                    lo = std::min(lo, sx); hi = std::max(hi, sx);
                }
                spanX = std::max(spanX, hi - lo + 2);
                ext.push_back(lo);
            }
            for (int ty = 0; ty < t.tilesY; ty++) {
                int lo = 1 << 30, hi = -1;
                for (int k = 0; k < SD_PT_TH; k++) {
                    const int sy = rt[4 * refl(std::min(ty * SD_PT_TH + k, HPl - 1) - SD_EDGE, g.H)];
                    lo = std::min(lo, std::min(std::max(sy, 0), sH - 1)); hi = std::max(hi, std::min(std::max(sy + 1, 0), sH - 1));
                }
                spanY = std::max(spanY, hi - lo + 1);
                ext.push_back(lo); ext.push_back(hi - lo + 1);
            }
            t.srcRowBytes = (spanX + 16 + 15) & ~15; t.srcRowsMax = spanY;
            t.lds = (size_t)t.srcRowsMax * t.srcRowBytes + (size_t)t.srcRowsMax * SD_PT_TW * 2;
            t.use = t.lds <= 48 * 1024 && g.W >= 40 && g.H >= 40;
            b->pyrTiles[l] = t;
        }
        if (!ext.empty()) {
            HIPCHK(hipMalloc((void**)&b->d_pyrExt, ext.size() * 4));
            HIPCHK(hipMemcpy(b->d_pyrExt, ext.data(), ext.size() * 4, hipMemcpyHostToDevice));
        }
    }
    for (int l = 1; l < nl; l++) {
        ProfScope ps(b, s, K_PYR);
        const SdLevel& g = P.lv[l];
        const SdPyrTiles& t = b->pyrTiles[l];
        if (t.use) {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pyr_level_tiles<SD_PT_TW, SD_PT_TH>), dim3(t.tilesX, t.tilesY, n_images), dim3(256), t.lds, s, b->d_pyr,
                               (const short4*)b->d_tabs, b->d_plan, l, t.srcRowBytes, t.srcRowsMax, b->d_pyrExt + t.extOff);
        } else {                         // resize ratios too large for the LDS tile: one thread per 4 pixels x 4 rows
            dim3 blk(64, 4), grd(((g.W + 39 + 3) / 4 + 63) / 64, (g.H + 2 * SD_EDGE + 4 * SD_PYR_ROWS - 1) / (4 * SD_PYR_ROWS), n_images);
            hipLaunchKernelGGL(k_pyr_level, grd, blk, 0, s, b->d_pyr, (const short4*)b->d_tabs, b->d_plan, l);
        }
    }
    LAUNCH_CHECK("k_pyr_level");
    {
        // (Running the blur on a side stream was measured twice: forked after the pyramid (beside FAST) no gain; forked after
        // FAST so that it runs beside the quadtree both kernels stretch (0.45 + 0.35 ms -> 0.73 ms together): +0.8 % frames/s,
        // not worth a second stream and overlapped per-kernel timings — everything stays on one stream.)
        ProfScope ps(b, s, K_BLUR);
        if (!b->d_blurTiles) {                                // once per batch: the non-empty 128 x SD_BLUR_TR tiles of one image, level by level
            std::vector<int> t;
            for (int l = 0; l < nl; l++)
                for (int ty = 0; ty * SD_BLUR_TR < P.lv[l].H; ty++)
                    for (int tx = 0; tx * 128 < P.lv[l].W; tx++) t.push_back(l | (tx << 8) | (ty << 16));
            b->nBlurTiles = (int)t.size();
            HIPCHK(hipMalloc((void**)&b->d_blurTiles, t.size() * 4));
            HIPCHK(hipMemcpy(b->d_blurTiles, t.data(), t.size() * 4, hipMemcpyHostToDevice));
        }
        dim3 grd((unsigned)b->nBlurTiles * (unsigned)((n_images + 7) / 8 * 8));
        unsigned tapSum = 0;
        for (int i = 0; i < 7; i++) tapSum += b->hplan.taps[i];
        if (tapSum <= 256) hipLaunchKernelGGL(k_blur_wide<false>, grd, dim3(256), 0, s, b->d_pyr, b->d_blur, b->d_plan, b->d_blurTiles, b->nBlurTiles, n_images);
        else hipLaunchKernelGGL(k_blur_wide<true>, grd, dim3(256), 0, s, b->d_pyr, b->d_blur, b->d_plan, b->d_blurTiles, b->nBlurTiles, n_images);
    }
    LAUNCH_CHECK("k_blur");
    {
        ProfScope ps(b, s, K_FAST);
        dim3 grd((unsigned)P.cells.size(), n_images);
        if (P.maxWin <= SD_FS_MAXWIN)
        {
            const int listCap = b->fastListCap;
            const size_t lds = (size_t)listCap * 4 + ((size_t)b->fastMaxCap + 4) * 4;
            // 128-thread workgroups: measured best (64: 0.66 ms, 128: 0.49 ms, 256: 0.65 ms per 128-image launch)
            // grid = (8 x cells, image groups): workgroup -> (cell, XCD, image group) -- see the kernel
            SdFastArgs fa;
            fa.pyrImageBytes = P.pyrImageBytes; fa.cellTotal = (int)P.cells.size(); fa.nImages = n_images; fa.listCap = listCap;
            fa.minTh = b->hplan.minTh; fa.iniTh = b->hplan.iniTh; fa.cellListCap = P.cellListCap;
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fast_cells_staged<128>), dim3((unsigned)P.cells.size() * 8u, (unsigned)((n_images + 7) / 8)), dim3(128),
                               lds, s, b->d_pyr, b->d_fcells, b->d_cellList, b->d_cellCount, fa);
        }
        else
            hipLaunchKernelGGL(k_fast_cells, grd, dim3(256), 0, s, b->d_pyr, b->d_cells, b->d_cellList, b->d_cellCount, b->d_plan);
    }
    LAUNCH_CHECK("k_fast_cells");
    {
        ProfScope ps(b, s, K_QTREE);
        dim3 grd(n_images, nl);
        hipLaunchKernelGGL(k_quadtree, grd, dim3(256), b->qtLds, s, b->d_cellList, b->d_cellCount, b->d_cells, b->d_cand,
                           b->d_nodeOf, b->d_lvlCount, b->d_candCount, b->d_lvlKp, b->d_err, b->d_plan, b->qtMN, b->qtSortP);
    }
    LAUNCH_CHECK("k_quadtree");
    {
        ProfScope ps(b, s, K_ORIENT);
        const int gpi = (P.kpCapLevels + 7) / 8, n8 = (n_images + 7) / 8 * 8;
        hipLaunchKernelGGL(k_orient, dim3((unsigned)gpi * n8), dim3(256), 0, s, b->d_pyr, b->d_lvlKp, b->d_lvlCount, b->d_kp, b->d_rot, b->d_count,
                           b->d_plan, n_images, gpi);
    }
    LAUNCH_CHECK("k_orient");
    {
        ProfScope ps(b, s, K_DESC);
        const int gpi = (P.kpCapLevels + 4 * SD_DP_KPW - 1) / (4 * SD_DP_KPW), n8 = (n_images + 7) / 8 * 8;
        hipLaunchKernelGGL(k_describe, dim3((unsigned)gpi * n8), dim3(256), 0, s, b->d_blur, b->d_lvlKp, b->d_lvlCount, b->d_rot, b->d_desc, b->d_plan,
                           n_images, gpi);
    }
    LAUNCH_CHECK("k_describe");
    b->nExtracted = n_images;
    for (int i = 0; i < n_images; i++) b->slotValid[i] = 1;
    return SD_OK;
}

int sd_batch_sync(sd_batch* b)
{
    if (!b) return SD_ERR_INVALID;
    HIPCHK(hipStreamSynchronize(b->lastStream));
    drain_profile(b);
    int err = 0;
    HIPCHK(hipMemcpy(&err, b->d_err, 4, hipMemcpyDeviceToHost));
    if (err) {
        (void)hipMemset(b->d_err, 0, 4);
        return set_err(SD_ERR_UNSUPPORTED, "device capacity exceeded: 1|2 quadtree nodes, 4 projection candidates, 8|16|32 box tables, 64 local-map candidates (flag " + std::to_string(err) + ")");
    }
    return SD_OK;
}

int sd_batch_extract_host(sd_batch* b, const uint8_t* gray, size_t stride, size_t image_pitch, int n_images)
{
    if (!b || n_images < 0 || n_images > b->maxImages) return set_err(SD_ERR_INVALID, "bad extract arguments");
    if (!gray || n_images == 0) { b->nExtracted = 0; return SD_OK; }   // empty image: silent return
    const SdPlan& P = b->plan;
    const size_t tight = (size_t)P.W * P.H;
    const size_t need = tight * n_images;
    if (b->stageBytes < need) {
        if (b->d_stage) (void)hipFree(b->d_stage);
        b->d_stage = nullptr; b->stageBytes = 0;
        HIPCHK(hipMalloc((void**)&b->d_stage, need));
        b->stageBytes = need;
    }
    for (int i = 0; i < n_images; i++)
        HIPCHK(hipMemcpy2DAsync(b->d_stage + tight * i, P.W, gray + image_pitch * i, stride, P.W, P.H,
                                hipMemcpyHostToDevice, b->stream));
    int rc = sd_batch_extract_device(b, b->d_stage, P.W, tight, n_images, b->stream);
    if (rc != SD_OK) return rc;
    return sd_batch_sync(b);
}

int sd_batch_results_device(sd_batch* b, sd_keypoint** d_kp, uint8_t** d_desc, int32_t** d_count, int* cap)
{
    if (!b) return SD_ERR_INVALID;
    if (d_kp) *d_kp = b->d_kp;
    if (d_desc) *d_desc = b->d_desc;
    if (d_count) *d_count = b->d_count;
    if (cap) *cap = b->plan.kpCap;
    return SD_OK;
}

int sd_batch_counts(sd_batch* b, int32_t* counts, int n_images)
{
    if (!b || !counts || n_images < 0 || n_images > b->maxImages) return SD_ERR_INVALID;
    int rc = sd_batch_sync(b);
    if (rc != SD_OK) return rc;
    for (int i = 0; i < n_images; i++) counts[i] = 0;
    int n = std::min(n_images, b->nExtracted);
    if (n > 0) HIPCHK(hipMemcpy(counts, b->d_count, (size_t)n * 4, hipMemcpyDeviceToHost));
    return SD_OK;
}

int sd_batch_download(sd_batch* b, int image, sd_keypoint* kp, uint8_t* desc, int cap, int* n, int32_t* per_level)
{
    if (!b || !n || image < 0 || image >= b->maxImages) return SD_ERR_INVALID;
    int rc = sd_batch_sync(b);
    if (rc != SD_OK) return rc;
    *n = 0;
    if (per_level) for (int l = 0; l < b->plan.nlevels; l++) per_level[l] = 0;
    if (!slot_ok(b, image)) return SD_OK;
    int cnt = 0;
    HIPCHK(hipMemcpy(&cnt, b->d_count + image, 4, hipMemcpyDeviceToHost));
    if (cnt > cap) { *n = cnt; return set_err(SD_ERR_CAPACITY, "keypoint buffer too small"); }
    if (cnt > 0) {
        if (kp) HIPCHK(hipMemcpy(kp, b->d_kp + (size_t)image * b->plan.kpCap, (size_t)cnt * sizeof(sd_keypoint), hipMemcpyDeviceToHost));
        if (desc) HIPCHK(hipMemcpy(desc, b->d_desc + (size_t)image * b->plan.kpCap * 32, (size_t)cnt * 32, hipMemcpyDeviceToHost));
    }
    if (per_level)
        HIPCHK(hipMemcpy(per_level, b->d_lvlCount + (size_t)image * b->plan.nlevels, (size_t)b->plan.nlevels * 4, hipMemcpyDeviceToHost));
    *n = cnt;
    return SD_OK;
}

int sd_batch_pyramid_level(sd_batch* b, int image, int level, const uint8_t** d_interior, int* w, int* h, size_t* stride)
{
    if (!b || image < 0 || image >= b->maxImages || level < 0 || level >= b->plan.nlevels) return SD_ERR_INVALID;
    const SdLevel& g = b->plan.lv[level];
    if (d_interior) *d_interior = b->d_pyr + (size_t)image * b->plan.pyrImageBytes + g.pyrOffset + (size_t)SD_EDGE * g.stride + SD_XOFF;
    if (w) *w = g.W;
    if (h) *h = g.H;
    if (stride) *stride = (size_t)g.stride;
    return SD_OK;
}

int sd_batch_download_pyramid(sd_batch* b, int image, int level, uint8_t* padded_out)
{
    if (!b || !padded_out || image < 0 || image >= b->maxImages || level < 0 || level >= b->plan.nlevels) return SD_ERR_INVALID;
    int rc = sd_batch_sync(b);
    if (rc != SD_OK) return rc;
    const SdLevel& g = b->plan.lv[level];
    const uint8_t* src = b->d_pyr + (size_t)image * b->plan.pyrImageBytes + g.pyrOffset + (SD_XOFF - SD_EDGE);
    HIPCHK(hipMemcpy2D(padded_out, g.W + 2 * SD_EDGE, src, g.stride, g.W + 2 * SD_EDGE, g.H + 2 * SD_EDGE, hipMemcpyDeviceToHost));
    return SD_OK;
}

int sd_batch_download_blurred(sd_batch* b, int image, int level, uint8_t* out)
{
    if (!b || !out || image < 0 || image >= b->maxImages || level < 0 || level >= b->plan.nlevels) return SD_ERR_INVALID;
    int rc = sd_batch_sync(b);
    if (rc != SD_OK) return rc;
    const SdLevel& g = b->plan.lv[level];
    const uint8_t* src = b->d_blur + (size_t)image * b->plan.blurImageBytes + g.blurOffset;
    HIPCHK(hipMemcpy2D(out, g.W, src, g.blurStride, g.W, g.H, hipMemcpyDeviceToHost));
    return SD_OK;
}

int sd_batch_candidate_counts(sd_batch* b, int image, int32_t* per_level)
{
    if (!b || !per_level || image < 0 || image >= b->maxImages) return SD_ERR_INVALID;
    int rc = sd_batch_sync(b);
    if (rc != SD_OK) return rc;
    HIPCHK(hipMemcpy(per_level, b->d_candCount + (size_t)image * b->plan.nlevels, (size_t)b->plan.nlevels * 4, hipMemcpyDeviceToHost));
    return SD_OK;
}

// ---------------------------------------------------------------- stereo / RGB-D
int sd_batch_stereo_match(sd_batch* b, int n_frames, float mbf, float fx, void* stream_)
{
    if (!b || n_frames < 0) return SD_ERR_INVALID;
    if (2 * n_frames > b->nExtracted) return set_err(SD_ERR_STATE, "stereo_match needs 2*n_frames extracted images (L,R interleaved)");
    if (!(fx > 0) || !(mbf > 0)) return set_err(SD_ERR_INVALID, "mbf and fx must be positive");
    if (b->plan.kpCap > 65535) return set_err(SD_ERR_UNSUPPORTED, "more than 65535 keypoints per image");
    hipStream_t s = stream_ ? (hipStream_t)stream_ : b->lastStream;
    b->lastStream = s;
    if (n_frames == 0) return SD_OK;
    const int H0 = b->plan.lv[0].H;
    // a right keypoint's band is floor(y - r) .. ceil(y + r), r = 2*scale[octave]: +2 covers floor/ceil and (int)y
    const int bandR = (int)ceilf(2.0f * b->plan.lv[b->plan.nlevels - 1].scale) + 2;
    {
        ProfScope ps(b, s, K_STEREO);
        const size_t lds = (size_t)(2 * H0 + 16) * 4;
        hipLaunchKernelGGL(k_row_sort, dim3(n_frames, 2), dim3(256), lds, s, b->d_kp, b->d_count, b->d_rowIdx, b->d_rowStart,
                           b->plan.kpCap, H0);
    }
    LAUNCH_CHECK("k_row_sort");
    {
        ProfScope ps(b, s, K_STEREO);
        if (SD_SR_ROWS + 2 * bandR + 2 > SD_SR_RS) return set_err(SD_ERR_UNSUPPORTED, "stereo row band larger than the staged row table (too many pyramid levels)");
        const int chunks = (H0 + SD_SR_ROWS - 1) / SD_SR_ROWS;      // SD_SR_ROWS image rows of one frame per workgroup, XCD-aware 1-D order
        hipLaunchKernelGGL(k_stereo_match, dim3((unsigned)chunks * (unsigned)((n_frames + 7) / 8 * 8)), dim3(256), 0, s, b->d_kp, b->d_desc, b->d_count, b->d_pyr, b->d_uright,
                           b->d_depth, b->d_sad, b->d_rowIdx, b->d_rowStart, bandR, b->d_plan, mbf, fx, n_frames, chunks);
    }
    LAUNCH_CHECK("k_stereo_match");
    {
        ProfScope ps(b, s, K_STEREO_F);
        hipLaunchKernelGGL(k_stereo_filter, dim3(n_frames), dim3(256), 0, s, b->d_count, b->d_uright, b->d_depth, b->d_sad, b->d_plan);
    }
    LAUNCH_CHECK("k_stereo_filter");
    b->nStereo = n_frames;
    return SD_OK;
}

int sd_batch_stereo_device(sd_batch* b, float** d_uright, float** d_depth, int* cap)
{
    if (!b) return SD_ERR_INVALID;
    if (d_uright) *d_uright = b->d_uright;
    if (d_depth) *d_depth = b->d_depth;
    if (cap) *cap = b->plan.kpCap;
    return SD_OK;
}

int sd_batch_download_stereo(sd_batch* b, int frame, float* uright, float* depth, int32_t* sad_dist, int cap)
{
    if (!b || frame < 0 || frame >= b->nStereo) return SD_ERR_INVALID;
    int rc = sd_batch_sync(b);
    if (rc != SD_OK) return rc;
    int cnt = 0;
    HIPCHK(hipMemcpy(&cnt, b->d_count + 2 * frame, 4, hipMemcpyDeviceToHost));
    if (cnt > cap) return set_err(SD_ERR_CAPACITY, "stereo buffer too small");
    const size_t off = (size_t)(2 * frame) * b->plan.kpCap;
    if (cnt > 0) {
        if (uright) HIPCHK(hipMemcpy(uright, b->d_uright + off, (size_t)cnt * 4, hipMemcpyDeviceToHost));
        if (depth) HIPCHK(hipMemcpy(depth, b->d_depth + off, (size_t)cnt * 4, hipMemcpyDeviceToHost));
        if (sad_dist) HIPCHK(hipMemcpy(sad_dist, b->d_sad + off, (size_t)cnt * 4, hipMemcpyDeviceToHost));
    }
    return SD_OK;
}

int sd_batch_rgbd_from_u16(sd_batch* b, const uint16_t* d_depth, size_t stride_elems, size_t image_pitch_elems,
                           int n_images, float depth_factor, float mbf, void* stream_)
{
    if (!b || !d_depth || n_images < 0) return SD_ERR_INVALID;
    if (n_images > b->nExtracted) return set_err(SD_ERR_STATE, "rgbd lookup needs extracted images");
    hipStream_t s = stream_ ? (hipStream_t)stream_ : b->lastStream;
    b->lastStream = s;
    if (n_images == 0) return SD_OK;
    {
        ProfScope ps(b, s, K_RGBD);
        dim3 grd((b->plan.kpCap + 255) / 256, n_images);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rgbd<uint16_t>), grd, dim3(256), 0, s, b->d_kp, KPUN(b), b->d_count, d_depth, stride_elems,
                           image_pitch_elems, depth_factor, mbf, b->d_uright, b->d_depth, b->d_plan);
    }
    LAUNCH_CHECK("k_rgbd");
    return SD_OK;
}

int sd_batch_rgbd_from_f32(sd_batch* b, const float* d_depth, size_t stride_elems, size_t image_pitch_elems, int n_images,
                           float mbf, void* stream_)
{
    return sd_batch_rgbd_from_f32_scaled(b, d_depth, stride_elems, image_pitch_elems, n_images, 1.0f, mbf, stream_);
}

int sd_batch_rgbd_from_f32_scaled(sd_batch* b, const float* d_depth, size_t stride_elems, size_t image_pitch_elems, int n_images,
                                  float depth_factor, float mbf, void* stream_)
{
    if (!b || !d_depth || n_images < 0) return SD_ERR_INVALID;
    if (n_images > b->nExtracted) return set_err(SD_ERR_STATE, "rgbd lookup needs extracted images");
    hipStream_t s = stream_ ? (hipStream_t)stream_ : b->lastStream;
    b->lastStream = s;
    if (n_images == 0) return SD_OK;
    {
        ProfScope ps(b, s, K_RGBD);
        dim3 grd((b->plan.kpCap + 255) / 256, n_images);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rgbd<float>), grd, dim3(256), 0, s, b->d_kp, KPUN(b), b->d_count, d_depth, stride_elems,
                           image_pitch_elems, depth_factor, mbf, b->d_uright, b->d_depth, b->d_plan);
    }
    LAUNCH_CHECK("k_rgbd");
    return SD_OK;
}

int sd_batch_download_rgbd(sd_batch* b, int image, float* uright, float* depth, int cap)
{
    if (!b || !slot_ok(b, image)) return SD_ERR_INVALID;
    int rc = sd_batch_sync(b);
    if (rc != SD_OK) return rc;
    int cnt = 0;
    HIPCHK(hipMemcpy(&cnt, b->d_count + image, 4, hipMemcpyDeviceToHost));
    if (cnt > cap) return set_err(SD_ERR_CAPACITY, "rgbd buffer too small");
    const size_t off = (size_t)image * b->plan.kpCap;
    if (cnt > 0) {
        if (uright) HIPCHK(hipMemcpy(uright, b->d_uright + off, (size_t)cnt * 4, hipMemcpyDeviceToHost));
        if (depth) HIPCHK(hipMemcpy(depth, b->d_depth + off, (size_t)cnt * 4, hipMemcpyDeviceToHost));
    }
    return SD_OK;
}

// ---------------------------------------------------------------- preprocessing / Hamming
int sd_cvt_gray_device(const uint8_t* d_src, int width, int height, size_t src_stride, size_t src_pitch, int channels,
                       int rgb_order, uint8_t* d_dst, size_t dst_stride, size_t dst_pitch, int n_images, void* stream)
{
    if (!d_src || !d_dst || width < 1 || height < 1 || n_images < 0 || (channels != 3 && channels != 4)) return SD_ERR_INVALID;
    if (n_images == 0) return SD_OK;
    dim3 blk(64, 4), grd(((width + 3) / 4 + 63) / 64, (height + 3) / 4, n_images);
    if (channels == 3)
        hipLaunchKernelGGL(k_cvt_gray3_wide, dim3(((width + 15) / 16 + 63) / 64, (height + 3) / 4, n_images), blk, 0, (hipStream_t)stream, d_src, width, height, src_stride, src_pitch,
                           rgb_order, d_dst, dst_stride, dst_pitch);
    else
        hipLaunchKernelGGL(k_cvt_gray, grd, blk, 0, (hipStream_t)stream, d_src, width, height, src_stride, src_pitch, channels,
                           rgb_order, d_dst, dst_stride, dst_pitch);
    LAUNCH_CHECK("k_cvt_gray");
    return SD_OK;
}

int sd_depth_to_f32_device(const uint16_t* d_src, int width, int height, size_t src_stride_elems, float factor, float* d_dst,
                           int n_images, size_t src_pitch_elems, void* stream)
{
    if (!d_src || !d_dst || width < 1 || height < 1 || n_images < 0) return SD_ERR_INVALID;
    if (n_images == 0) return SD_OK;
    dim3 blk(64, 4), grd((width + 63) / 64, (height + 3) / 4, n_images);
    hipLaunchKernelGGL(k_depth_to_f32, grd, blk, 0, (hipStream_t)stream, d_src, width, height, src_stride_elems, src_pitch_elems,
                       factor, d_dst);
    LAUNCH_CHECK("k_depth_to_f32");
    return SD_OK;
}

int sd_descriptor_distance(const uint8_t a[32], const uint8_t b[32])
{
    int d = 0;
    for (int i = 0; i < 4; i++) {
        uint64_t x, y;
        memcpy(&x, a + 8 * i, 8); memcpy(&y, b + 8 * i, 8);
        d += __builtin_popcountll(x ^ y);
    }
    return d;
}

int sd_hamming_matrix_device(const uint8_t* d_a, int na, const uint8_t* d_b, int nb, uint16_t* d_out, void* stream)
{
    if (!d_a || !d_b || !d_out || na < 0 || nb < 0) return SD_ERR_INVALID;
    if (na == 0 || nb == 0) return SD_OK;
    dim3 blk(64, 4), grd((nb + 63) / 64, (na + 3) / 4);
    hipLaunchKernelGGL(k_hamming_matrix, grd, blk, 0, (hipStream_t)stream, d_a, na, d_b, nb, d_out);
    LAUNCH_CHECK("k_hamming_matrix");
    return SD_OK;
}


// ---------------------------------------------------------------- grid / unproject / projection matcher
struct SdCopySegs { const char* src[24]; char* dst[24]; unsigned bytes[24]; int n; };
__global__ void __launch_bounds__(256) k_copy_segments(SdCopySegs S)
{
    const int seg = blockIdx.y;
    const unsigned n = S.bytes[seg];
    const char* s = S.src[seg]; char* d = S.dst[seg];
    const bool aligned = ((((size_t)s) | ((size_t)d)) & 3) == 0;
    const unsigned words = aligned ? n >> 2 : 0;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < words; i += gridDim.x * 256) ((uint32_t*)d)[i] = ((const uint32_t*)s)[i];
    for (unsigned i = (words << 2) + blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) d[i] = s[i];
}

static int cam_ok(const sd_camera* c)
{
    return c && c->fx > 0 && c->fy > 0 && c->mnMaxX > c->mnMinX && c->mnMaxY > c->mnMinY;
}
static SdCamera to_cam(const sd_camera* c)
{
    SdCamera k = {c->fx, c->fy, c->cx, c->cy, c->mbf, c->mb, c->mnMinX, c->mnMaxX, c->mnMinY, c->mnMaxY};
    return k;
}

static int assign_grid_impl(sd_batch* b, int n_images, int image_step, const sd_camera* cam, void* stream_);
int sd_batch_assign_grid(sd_batch* b, int n_images, const sd_camera* cam, void* stream_) { return assign_grid_impl(b, n_images, 1, cam, stream_); }

// slots 0, image_step, 2 * image_step, ... (n_images of them): the tracker grids the left images only
static int assign_grid_impl(sd_batch* b, int n_images, int image_step, const sd_camera* cam, void* stream_)
{
    if (!b || n_images < 0 || image_step < 1 || !cam_ok(cam)) return set_err(SD_ERR_INVALID, "bad grid arguments");
    if ((n_images - 1) * image_step + 1 > b->nExtracted) return set_err(SD_ERR_STATE, "grid needs extracted images");
    hipStream_t s = stream_ ? (hipStream_t)stream_ : b->lastStream;
    b->lastStream = s;
    if (n_images == 0) return SD_OK;
    {
        ProfScope ps(b, s, K_GRID);
        dim3 grd((b->plan.kpCap + 255) / 256, n_images);
        hipLaunchKernelGGL(k_grid_cells, grd, dim3(256), 0, s, KPUN(b), b->d_count, b->d_cellOf, to_cam(cam), b->plan.kpCap, image_step);
    }
    LAUNCH_CHECK("k_grid_cells");
    const size_t gridLds = (size_t)(SD_GRID_CELLS + 8) * 4 + (size_t)SD_GRID_CELLS * 4 + (size_t)b->plan.kpCap * 2 + 16;
    if (gridLds > 64 * 1024) return set_err(SD_ERR_UNSUPPORTED, "too many keypoints per image for the grid sort");
    {
        ProfScope ps(b, s, K_GRID);
        hipLaunchKernelGGL(k_grid_sort, dim3(n_images), dim3(256), gridLds, s, b->d_cellOf, b->d_count, b->d_sortedIdx,
                           b->d_cellStart, b->plan.kpCap, image_step);
    }
    LAUNCH_CHECK("k_grid_sort");
    return SD_OK;
}

int sd_batch_download_grid(sd_batch* b, int image, int16_t* cell, int cap)
{
    if (!b || !cell || !slot_ok(b, image)) return SD_ERR_INVALID;
    int rc = sd_batch_sync(b);
    if (rc != SD_OK) return rc;
    int cnt = 0;
    HIPCHK(hipMemcpy(&cnt, b->d_count + image, 4, hipMemcpyDeviceToHost));
    if (cnt > cap) return set_err(SD_ERR_CAPACITY, "grid buffer too small");
    if (cnt > 0) HIPCHK(hipMemcpy(cell, b->d_cellOf + (size_t)image * b->plan.kpCap, (size_t)cnt * 2, hipMemcpyDeviceToHost));
    return SD_OK;
}

int sd_batch_unproject(sd_batch* b, int first_image, int image_step, int n_frames, const sd_camera* cam, const float* Twc_host,
                       void* stream_)
{
    if (!b || n_frames < 0 || first_image != 0 || image_step < 1 || !cam_ok(cam) || !Twc_host)
        return set_err(SD_ERR_INVALID, "bad unproject arguments");
    if (n_frames > 0 && !slot_ok(b, (n_frames - 1) * image_step)) return set_err(SD_ERR_STATE, "unproject needs extracted images");
    hipStream_t s = stream_ ? (hipStream_t)stream_ : b->lastStream;
    b->lastStream = s;
    if (n_frames == 0) return SD_OK;
    HIPCHK(hipMemcpyAsync(b->d_pose, Twc_host, (size_t)n_frames * 64, hipMemcpyHostToDevice, s));
    {
        ProfScope ps(b, s, K_UNPROJ);
        dim3 grd((b->plan.kpCap + 255) / 256, n_frames);
        hipLaunchKernelGGL(k_unproject, grd, dim3(256), 0, s, KPUN(b), b->d_count, b->d_depth, b->d_pose, b->d_xw, b->d_flags,
                           to_cam(cam), b->plan.kpCap, image_step);
    }
    LAUNCH_CHECK("k_unproject");
    return SD_OK;
}

int sd_batch_mappoints_device(sd_batch* b, float** d_xw, uint8_t** d_flags, int* cap)
{
    if (!b) return SD_ERR_INVALID;
    if (d_xw) *d_xw = b->d_xw;
    if (d_flags) *d_flags = b->d_flags;
    if (cap) *cap = b->plan.kpCap;
    return SD_OK;
}

int sd_batch_set_mappoints(sd_batch* b, int image, const float* xw, const uint8_t* flags, int n)
{
    if (!b || image < 0 || image >= b->maxImages || n < 0 || n > b->plan.kpCap || !xw || !flags) return SD_ERR_INVALID;
    int rc = sd_batch_sync(b);
    if (rc != SD_OK) return rc;
    if (n > 0) {
        HIPCHK(hipMemcpy(b->d_xw + (size_t)image * b->plan.kpCap * 3, xw, (size_t)n * 12, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(b->d_flags + (size_t)image * b->plan.kpCap, flags, (size_t)n, hipMemcpyHostToDevice));
    }
    return SD_OK;
}

int sd_batch_download_mappoints(sd_batch* b, int image, float* xw, uint8_t* flags, int cap)
{
    if (!b || !slot_ok(b, image)) return SD_ERR_INVALID;
    int rc = sd_batch_sync(b);
    if (rc != SD_OK) return rc;
    int cnt = 0;
    HIPCHK(hipMemcpy(&cnt, b->d_count + image, 4, hipMemcpyDeviceToHost));
    if (cnt > cap) return set_err(SD_ERR_CAPACITY, "map point buffer too small");
    if (cnt > 0) {
        if (xw) HIPCHK(hipMemcpy(xw, b->d_xw + (size_t)image * b->plan.kpCap * 3, (size_t)cnt * 12, hipMemcpyDeviceToHost));
        if (flags) HIPCHK(hipMemcpy(flags, b->d_flags + (size_t)image * b->plan.kpCap, (size_t)cnt, hipMemcpyDeviceToHost));
    }
    return SD_OK;
}

// pairBase: first pair index of the per-pair arrays this call uses (the tracker keeps TrackHomo's pairs at [0, S) and
// TrackWithMotionModel's at [S, 2S)); d_active / redoBelow: see SdProjArgs; upload: 0 = index and pose arrays of a preceding
// call are reused (the 2*th retry).
static int search_by_projection_impl(sd_batch* b, int pairBase, int n_pairs, const int32_t* cur_index, const int32_t* last_index,
                                     const float* Tcw_host, const float* Tlw_host, const sd_camera* cam, float th, int bMono,
                                     int checkOrientation, const uint8_t* d_occupied, const uint8_t* d_mp_desc, void* stream_,
                                     const int* d_active, int redoBelow, int upload);

int sd_batch_search_by_projection(sd_batch* b, int n_pairs, const int32_t* cur_index, const int32_t* last_index,
                                  const float* Tcw_host, const float* Tlw_host, const sd_camera* cam, float th, int bMono,
                                  int checkOrientation, const uint8_t* d_occupied, const uint8_t* d_mp_desc, void* stream_)
{
    return search_by_projection_impl(b, 0, n_pairs, cur_index, last_index, Tcw_host, Tlw_host, cam, th, bMono, checkOrientation,
                                     d_occupied, d_mp_desc, stream_, nullptr, 0, 1);
}

static int search_by_projection_impl(sd_batch* b, int pairBase, int n_pairs, const int32_t* cur_index, const int32_t* last_index,
                                     const float* Tcw_host, const float* Tlw_host, const sd_camera* cam, float th, int bMono,
                                     int checkOrientation, const uint8_t* d_occupied, const uint8_t* d_mp_desc, void* stream_,
                                     const int* d_active, int redoBelow, int upload)
{
    if (!b || n_pairs < 0 || pairBase < 0 || pairBase + n_pairs > b->maxImages || !cam_ok(cam) || !Tcw_host || !Tlw_host || !(th > 0) ||
        (n_pairs > 0 && (!cur_index || !last_index)))
        return set_err(SD_ERR_INVALID, "bad search_by_projection arguments");
    std::vector<int2> idx(n_pairs);
    for (int p = 0; p < n_pairs; p++) {
        if (!slot_ok(b, cur_index[p]) || !slot_ok(b, last_index[p]))
            return set_err(SD_ERR_STATE, "search_by_projection: frame slot holds no results");
        idx[p] = make_int2(cur_index[p], last_index[p]);
    }
    if (b->plan.kpCap > 65535) return set_err(SD_ERR_UNSUPPORTED, "more than 65535 keypoints per image");
    hipStream_t s = stream_ ? (hipStream_t)stream_ : b->lastStream;
    b->lastStream = s;
    if (pairBase == 0) b->nPairs = 0;
    if (n_pairs == 0) return SD_OK;
    float* dTc = b->d_pose + (size_t)pairBase * 16;
    float* dTl = b->d_pose + ((size_t)b->maxImages + pairBase) * 16;
    int2* dIdx = b->d_pairIdx + pairBase;
    if (upload) {
        HIPCHK(hipMemcpyAsync(dTc, Tcw_host, (size_t)n_pairs * 64, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(dTl, Tlw_host, (size_t)n_pairs * 64, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(dIdx, idx.data(), (size_t)n_pairs * sizeof(int2), hipMemcpyHostToDevice, s));
    }
    const int cap = b->plan.kpCap;
    const size_t pOff = (size_t)pairBase * cap;
    {
        ProfScope ps(b, s, K_PROJ_A);
        dim3 grd((cap + 15) / 16, n_pairs);                 // 16 points per workgroup: four per wave, 16 lanes each
        SdProjArgs pa;
        pa.kp = KPUN(b); pa.desc = b->d_desc; pa.uRight = b->d_uright; pa.count = b->d_count; pa.cellOf = b->d_cellOf;
        pa.sortedIdx = b->d_sortedIdx; pa.cellStart = b->d_cellStart; pa.xw = b->d_xw; pa.flags = b->d_flags;
        pa.dmp = d_mp_desc ? d_mp_desc : b->d_desc; pa.Tcw = dTc; pa.Tlw = dTl; pa.cand = b->d_pcand + pOff * SD_PROJ_K; pa.ncand = b->d_pncand + pOff;
        pa.errFlag = b->d_err; pa.P = b->d_plan; pa.cam = to_cam(cam); pa.th = th; pa.bMono = bMono; pa.pairIdx = dIdx;
        pa.active = d_active; pa.redoNmatch = b->d_nmatch + pairBase; pa.redoBelow = redoBelow;
        hipLaunchKernelGGL(k_proj_candidates, grd, dim3(256), 0, s, pa);
    }
    LAUNCH_CHECK("k_proj_candidates");
    {
        ProfScope ps(b, s, K_PROJ_B);
        const size_t capA = (size_t)((cap + 15) & ~15);
        size_t lds = capA * (4 + 4 + 4 + 2 + 1 + 1 + 1) + 16;
        if (lds > 160 * 1024 - 256) return set_err(SD_ERR_UNSUPPORTED, "too many keypoints per image for the projection matcher's LDS tables");
        if (lds > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void*)k_proj_resolve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_proj_resolve, dim3(n_pairs), dim3(64), lds, s, KPUN(b), b->d_count, b->d_flags, b->d_pcand + pOff * SD_PROJ_K, b->d_pncand + pOff,
                           d_occupied, b->d_match + pOff, b->d_pairs + pOff * 2, b->d_npairs + pairBase, b->d_nmatch + pairBase, b->d_plan, checkOrientation,
                           dIdx, d_active, redoBelow, b->d_err);
    }
    LAUNCH_CHECK("k_proj_resolve");
    if (pairBase == 0) b->nPairs = n_pairs;
    b->dlPairs = std::max(pairBase + n_pairs, pairBase ? b->dlPairs : 0);
    return SD_OK;
}

// Tracking::SearchLocalPoints (Tracking.cc:2014-2064): Frame::isInFrustum for every local map point, then
// ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th) (ORBmatcher.cc:45-129).
int sd_batch_search_local_map(sd_batch* b, int n_frames, const int32_t* frame_index, const int32_t* point_offset,
                              const sd_map_point* d_points, const uint8_t* d_point_desc, const float* Tcw_host,
                              const sd_camera* cam, float th, float nnratio, float viewing_cos_limit,
                              const uint8_t* d_occupied, sd_track_info* d_track, int32_t* d_point_match,
                              int32_t* d_kp_match, int32_t* d_nmatches, void* stream_)
{
    if (!b || n_frames < 0 || n_frames > b->maxImages || !cam_ok(cam) || !(th > 0) ||
        (n_frames > 0 && (!frame_index || !point_offset || !Tcw_host || !d_track || !d_point_match || !d_kp_match || !d_nmatches)))
        return set_err(SD_ERR_INVALID, "bad search_local_map arguments");
    if (n_frames == 0) return SD_OK;
    int maxM = 0;
    for (int f = 0; f < n_frames; f++) {
        if (!slot_ok(b, frame_index[f])) return set_err(SD_ERR_STATE, "search_local_map: frame slot holds no results");
        const int M = point_offset[f + 1] - point_offset[f];
        if (M < 0 || point_offset[0] != 0) return set_err(SD_ERR_INVALID, "search_local_map: point offsets must start at 0 and ascend");
        maxM = std::max(maxM, M);
    }
    const int total = point_offset[n_frames];
    if (total > 0 && (!d_points || !d_point_desc)) return set_err(SD_ERR_INVALID, "search_local_map: no map points given");
    if (b->plan.kpCap > 65535) return set_err(SD_ERR_UNSUPPORTED, "more than 65535 keypoints per image");
    hipStream_t s = stream_ ? (hipStream_t)stream_ : b->lastStream;
    b->lastStream = s;
    if (total > b->lmCap) {                       // candidate scratch grows with the largest local map seen
        HIPCHK(hipStreamSynchronize(s));
        if (b->d_lmCand) { (void)hipFree(b->d_lmCand); (void)hipFree(b->d_lmN); (void)hipFree(b->d_lmOvf); b->d_lmCand = nullptr; b->d_lmN = b->d_lmOvf = nullptr; }
        const int want = std::max(total, 2 * b->lmCap);
        HIPCHK(hipMalloc((void**)&b->d_lmCand, (size_t)want * SD_PROJ_K * 4));
        HIPCHK(hipMalloc((void**)&b->d_lmN, (size_t)want));
        HIPCHK(hipMalloc((void**)&b->d_lmOvf, (size_t)want));
        b->lmCap = want;
    }
    if (!b->d_lmIdx) HIPCHK(hipMalloc((void**)&b->d_lmIdx, (size_t)(2 * b->maxImages + 2) * sizeof(int)));
    int* dFrameOf = b->d_lmIdx;
    int* dOff = b->d_lmIdx + b->maxImages;
    HIPCHK(hipMemcpyAsync(dFrameOf, frame_index, (size_t)n_frames * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dOff, point_offset, (size_t)(n_frames + 1) * 4, hipMemcpyHostToDevice, s));
    float* dT = b->d_pose;
    HIPCHK(hipMemcpyAsync(dT, Tcw_host, (size_t)n_frames * 64, hipMemcpyHostToDevice, s));
    const int cap = b->plan.kpCap;
    if (maxM > 0) {
        ProfScope ps(b, s, K_LOCAL_A);
        hipLaunchKernelGGL(k_local_candidates, dim3((maxM + 3) / 4, n_frames), dim3(256), 0, s, KPUN(b), b->d_desc, b->d_uright, b->d_count,
                           b->d_cellOf, b->d_sortedIdx, b->d_cellStart, (const SdMapPoint*)d_points, d_point_desc, dFrameOf, dOff, dT,
                           (SdTrack*)d_track, b->d_lmCand, b->d_lmN, b->d_lmOvf, b->d_plan, to_cam(cam), th, viewing_cos_limit);
        LAUNCH_CHECK("k_local_candidates");
    }
    {
        ProfScope ps(b, s, K_LOCAL_B);
        const size_t lds = (size_t)cap * 4 + cap + 16;
        hipLaunchKernelGGL(k_local_resolve, dim3(n_frames), dim3(64), lds, s, KPUN(b), b->d_count, (const SdMapPoint*)d_points, dFrameOf, dOff,
                           b->d_lmCand, b->d_lmN, b->d_lmOvf, d_occupied, d_point_match, d_kp_match, d_nmatches, b->d_err, b->d_plan, nnratio);
        LAUNCH_CHECK("k_local_resolve");
    }
    return SD_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Vocabulary (Thirdparty/DBoW2 TemplatedVocabulary) and the bag-of-words matcher
// ---------------------------------------------------------------------------------------------------------
static int require_device()
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return set_err(SD_ERR_NO_DEVICE, "no HIP device: the vocabulary lives in HBM, there is no CPU fallback");
    return SD_OK;
}

static int vocab_from_lines(sd_vocab** out, const SdVocabLines& lines)
{
    std::vector<uint8_t> blob;
    if (!vocab_pack(lines, blob)) return set_err(SD_ERR_INVALID, "vocabulary: a node names a parent that does not precede it");
    sd_vocab* v = new sd_vocab();
    memcpy(&v->h, blob.data(), sizeof(SdVocabHeader));
    if (hipMalloc(&v->d_blob, blob.size()) != hipSuccess) { delete v; return set_err(SD_ERR_HIP, "hipMalloc(vocabulary)"); }
    v->owned = true;
    if (hipMemcpy(v->d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(v->d_blob); delete v; return set_err(SD_ERR_HIP, "hipMemcpy(vocabulary)"); }
    vocab_bind(v);
    *out = v;
    return SD_OK;
}

int sd_vocab_load_text(sd_vocab** out, const char* path)
{
    if (!out || !path) return set_err(SD_ERR_INVALID, "null argument");
    *out = nullptr;
    int rc = require_device();
    if (rc != SD_OK) return rc;
    SdVocabLines lines;
    std::string err;
    if (!vocab_parse_text(path, lines, err)) return set_err(SD_ERR_INVALID, err);
    return vocab_from_lines(out, lines);
}

int sd_vocab_from_nodes(sd_vocab** out, int k, int L, int scoring, int weighting, int n_lines, const int32_t* parent,
                        const uint8_t* is_leaf, const uint8_t* desc, const double* weight)
{
    if (!out || n_lines < 0 || (n_lines > 0 && (!parent || !is_leaf || !desc || !weight))) return set_err(SD_ERR_INVALID, "null argument");
    *out = nullptr;
    if (k < 0 || k > 20 || L < 1 || L > 10 || scoring < 0 || scoring > 5 || weighting < 0 || weighting > 3)
        return set_err(SD_ERR_INVALID, "Vocabulary loading failure: This is not a correct text file!");
    int rc = require_device();
    if (rc != SD_OK) return rc;
    SdVocabLines lines;
    lines.k = k; lines.L = L; lines.scoring = scoring; lines.weighting = weighting;
    lines.parent.assign(parent, parent + n_lines); lines.isLeaf.assign(is_leaf, is_leaf + n_lines);
    lines.desc.assign(desc, desc + (size_t)n_lines * 32); lines.weight.assign(weight, weight + n_lines);
    return vocab_from_lines(out, lines);
}

int sd_vocab_from_packed_device(sd_vocab** out, void* d_blob, size_t bytes)
{
    if (!out || !d_blob || bytes < sizeof(SdVocabHeader)) return set_err(SD_ERR_INVALID, "null argument");
    *out = nullptr;
    int rc = require_device();
    if (rc != SD_OK) return rc;
    SdVocabHeader h;
    HIPCHK(hipMemcpy(&h, d_blob, sizeof(h), hipMemcpyDeviceToHost));
    SdVocabHeader chk = h;
    if (h.magic != SD_VOCAB_MAGIC || h.version != 1 || vocab_layout(chk) != h.totalBytes || h.totalBytes > bytes ||
        chk.offWordId != h.offWordId || chk.offDesc != h.offDesc)
        return set_err(SD_ERR_INVALID, "not a packed vocabulary");
    sd_vocab* v = new sd_vocab();
    v->h = h; v->d_blob = d_blob; v->owned = false;
    vocab_bind(v);
    *out = v;
    return SD_OK;
}

void sd_vocab_destroy(sd_vocab* v)
{
    if (!v) return;
    if (v->owned && v->d_blob) (void)hipFree(v->d_blob);
    delete v;
}

int sd_vocab_info(const sd_vocab* v, int* k, int* L, int* scoring, int* weighting, int* n_nodes, int* n_words)
{
    if (!v) return SD_ERR_INVALID;
    if (k) *k = (int)v->h.k; if (L) *L = (int)v->h.L; if (scoring) *scoring = (int)v->h.scoring; if (weighting) *weighting = (int)v->h.weighting;
    if (n_nodes) *n_nodes = (int)v->h.nNodes; if (n_words) *n_words = (int)v->h.nWords;
    return SD_OK;
}

int sd_vocab_packed_device(sd_vocab* v, void** d_blob, size_t* bytes)
{
    if (!v) return SD_ERR_INVALID;
    if (d_blob) *d_blob = v->d_blob;
    if (bytes) *bytes = (size_t)v->h.totalBytes;
    return SD_OK;
}

size_t sd_vocab_packed_bytes(int n_nodes)
{
    SdVocabHeader h = {};
    h.nNodes = (uint32_t)(n_nodes < 0 ? 0 : n_nodes);
    return vocab_layout(h);
}

int sd_vocab_download_nodes(const sd_vocab* v, int32_t* parent, int32_t* n_children, int32_t* word_id, uint8_t* desc, double* weight)
{
    if (!v) return SD_ERR_INVALID;
    const int n = (int)v->h.nNodes;
    const uint8_t* p = (const uint8_t*)v->d_blob;
    if (parent) HIPCHK(hipMemcpy(parent, p + v->h.offParent, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (word_id) HIPCHK(hipMemcpy(word_id, p + v->h.offWordId, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (desc) HIPCHK(hipMemcpy(desc, p + v->h.offDesc, (size_t)n * 32, hipMemcpyDeviceToHost));
    if (weight) HIPCHK(hipMemcpy(weight, p + v->h.offWeight, (size_t)n * 8, hipMemcpyDeviceToHost));
    if (n_children) {
        std::vector<int> cs(n + 1);
        HIPCHK(hipMemcpy(cs.data(), p + v->h.offChildStart, (size_t)(n + 1) * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; i++) n_children[i] = cs[i + 1] - cs[i];
    }
    return SD_OK;
}

static int bow_alloc(sd_batch* b)
{
    if (b->d_bowWordF) return SD_OK;
    const size_t nI = b->maxImages, cap = b->plan.kpCap;
    HIPCHK(hipMalloc((void**)&b->d_bowWordF, nI * cap * 4)); HIPCHK(hipMalloc((void**)&b->d_bowWF, nI * cap * 8));
    HIPCHK(hipMalloc((void**)&b->d_bowNidF, nI * cap * 4)); HIPCHK(hipMalloc((void**)&b->d_fvNode, nI * cap * 4));
    HIPCHK(hipMalloc((void**)&b->d_fvFeat, nI * cap * 4)); HIPCHK(hipMalloc((void**)&b->d_fvRunStart, nI * (cap + 1) * 4));
    HIPCHK(hipMalloc((void**)&b->d_fvRunNode, nI * cap * 4)); HIPCHK(hipMalloc((void**)&b->d_bowWord, nI * cap * 4));
    HIPCHK(hipMalloc((void**)&b->d_bowVal, nI * cap * 8)); HIPCHK(hipMalloc((void**)&b->d_bowMeta, nI * 4 * 4));
    HIPCHK(hipMalloc((void**)&b->d_bowImg, nI * 4));
    HIPCHK(hipMemset(b->d_bowMeta, 0, nI * 16));
    b->bowValid.assign(nI, 0);
    return SD_OK;
}

// Frame::ComputeBoW (src/Frame.cc:803-810) for the listed image slots.
int sd_batch_compute_bow(sd_batch* b, const sd_vocab* v, int n_images, const int32_t* image_index, int levelsup, void* stream_)
{
    if (!b || !v || n_images < 0 || n_images > b->maxImages || (n_images > 0 && !image_index) || levelsup < 0)
        return set_err(SD_ERR_INVALID, "bad compute_bow arguments");
    if (v->h.nNodes <= 1) return set_err(SD_ERR_INVALID, "compute_bow: empty vocabulary");
    for (int i = 0; i < n_images; i++) if (!slot_ok(b, image_index[i])) return set_err(SD_ERR_STATE, "compute_bow: image slot holds no results");
    const int cap = b->plan.kpCap;
    int sortN = 256;                     // LDS capacity of k_bow_finalize's sort: its per-image sort size starts at 256 too
    while (sortN < cap) sortN <<= 1;
    if (sortN > 8192) return set_err(SD_ERR_UNSUPPORTED, "compute_bow: more than 8192 keypoints per image");
    int rc = bow_alloc(b);
    if (rc != SD_OK) return rc;
    hipStream_t s = stream_ ? (hipStream_t)stream_ : b->lastStream;
    b->lastStream = s;
    if (n_images == 0) return SD_OK;
    HIPCHK(hipMemcpyAsync(b->d_bowImg, image_index, (size_t)n_images * 4, hipMemcpyHostToDevice, s));
    {
        ProfScope ps(b, s, K_BOW_T);
        hipLaunchKernelGGL(k_bow_transform, dim3((cap + 15) / 16, n_images), dim3(256), 0, s, b->d_desc, b->d_count, b->d_bowImg, v->dev, levelsup, cap,
                           b->d_bowWordF, b->d_bowWF, b->d_bowNidF);
        LAUNCH_CHECK("k_bow_transform");
    }
    {
        ProfScope ps(b, s, K_BOW_F);
        const size_t lds = (size_t)sortN * 8 + 257 * 4 + 16;
        hipLaunchKernelGGL(k_bow_finalize, dim3(n_images), dim3(256), lds, s, b->d_count, b->d_bowImg, b->d_bowWordF, b->d_bowWF, b->d_bowNidF, cap, sortN,
                           (int)v->h.scoring, (int)v->h.weighting, b->d_fvNode, b->d_fvFeat, b->d_fvRunStart, b->d_fvRunNode, b->d_bowWord,
                           b->d_bowVal, b->d_bowMeta);
        LAUNCH_CHECK("k_bow_finalize");
    }
    for (int i = 0; i < n_images; i++) b->bowValid[image_index[i]] = 1;
    return SD_OK;
}

int sd_batch_bow_device(sd_batch* b, uint32_t** d_bow_word, double** d_bow_value, uint32_t** d_fv_node, uint32_t** d_fv_feature,
                        int32_t** d_meta, int* cap)
{
    if (!b) return SD_ERR_INVALID;
    int rc = bow_alloc(b);
    if (rc != SD_OK) return rc;
    if (d_bow_word) *d_bow_word = b->d_bowWord; if (d_bow_value) *d_bow_value = b->d_bowVal;
    if (d_fv_node) *d_fv_node = b->d_fvNode; if (d_fv_feature) *d_fv_feature = b->d_fvFeat;
    if (d_meta) *d_meta = b->d_bowMeta; if (cap) *cap = b->plan.kpCap;
    return SD_OK;
}

int sd_batch_download_bow(sd_batch* b, int image, uint32_t* bow_word, double* bow_value, int* n_words, uint32_t* fv_node,
                          uint32_t* fv_feature, int* n_features, uint32_t* feature_word, double* feature_weight, uint32_t* feature_node, int cap)
{
    if (!b || image < 0 || image >= b->maxImages) return SD_ERR_INVALID;
    int rc = sd_batch_sync(b);
    if (rc != SD_OK) return rc;
    if (!b->d_bowWordF || !b->bowValid[image]) return set_err(SD_ERR_STATE, "download_bow: no bag of words computed for this slot");
    int meta[4];
    HIPCHK(hipMemcpy(meta, b->d_bowMeta + image * 4, 16, hipMemcpyDeviceToHost));
    int cnt = 0;
    HIPCHK(hipMemcpy(&cnt, b->d_count + image, 4, hipMemcpyDeviceToHost));
    if (cnt > cap) return set_err(SD_ERR_CAPACITY, "bow buffer too small");
    const size_t off = (size_t)image * b->plan.kpCap;
    if (n_words) *n_words = meta[2];
    if (n_features) *n_features = meta[0];
    if (meta[2] > 0) {
        if (bow_word) HIPCHK(hipMemcpy(bow_word, b->d_bowWord + off, (size_t)meta[2] * 4, hipMemcpyDeviceToHost));
        if (bow_value) HIPCHK(hipMemcpy(bow_value, b->d_bowVal + off, (size_t)meta[2] * 8, hipMemcpyDeviceToHost));
    }
    if (meta[0] > 0) {
        if (fv_node) HIPCHK(hipMemcpy(fv_node, b->d_fvNode + off, (size_t)meta[0] * 4, hipMemcpyDeviceToHost));
        if (fv_feature) HIPCHK(hipMemcpy(fv_feature, b->d_fvFeat + off, (size_t)meta[0] * 4, hipMemcpyDeviceToHost));
    }
    if (cnt > 0) {
        if (feature_word) HIPCHK(hipMemcpy(feature_word, b->d_bowWordF + off, (size_t)cnt * 4, hipMemcpyDeviceToHost));
        if (feature_weight) HIPCHK(hipMemcpy(feature_weight, b->d_bowWF + off, (size_t)cnt * 8, hipMemcpyDeviceToHost));
        if (feature_node) HIPCHK(hipMemcpy(feature_node, b->d_bowNidF + off, (size_t)cnt * 4, hipMemcpyDeviceToHost));
    }
    return SD_OK;
}

// ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches) (src/ORBmatcher.cc:159-288)
int sd_batch_search_by_bow(sd_batch* b, int n_pairs, const int32_t* kf_index, const int32_t* frame_index, const uint8_t* d_kf_valid,
                           float nnratio, int checkOrientation, void* stream_)
{
    if (!b || n_pairs < 0 || n_pairs > b->maxImages || (n_pairs > 0 && (!kf_index || !frame_index)))
        return set_err(SD_ERR_INVALID, "bad search_by_bow arguments");
    std::vector<int2> idx(n_pairs);
    for (int p = 0; p < n_pairs; p++) {
        if (!slot_ok(b, kf_index[p]) || !slot_ok(b, frame_index[p])) return set_err(SD_ERR_STATE, "search_by_bow: frame slot holds no results");
        if (!b->d_bowWordF || !b->bowValid[kf_index[p]] || !b->bowValid[frame_index[p]])
            return set_err(SD_ERR_STATE, "search_by_bow: sd_batch_compute_bow has not run on these slots");
        idx[p] = make_int2(kf_index[p], frame_index[p]);
    }
    hipStream_t s = stream_ ? (hipStream_t)stream_ : b->lastStream;
    b->lastStream = s;
    b->nPairs = 0; b->dlPairs = 0;
    if (n_pairs == 0) return SD_OK;
    HIPCHK(hipMemcpyAsync(b->d_pairIdx, idx.data(), (size_t)n_pairs * sizeof(int2), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(b->d_npairs, 0, (size_t)n_pairs * 4, s));
    const int cap = b->plan.kpCap;
    {
        ProfScope ps(b, s, K_BOW_S);
        const size_t lds = (size_t)cap * 4 + cap + 16;
        hipLaunchKernelGGL(k_search_by_bow, dim3(n_pairs), dim3(64 * SD_BOW_WAVES), lds, s, b->d_kp, b->d_desc, b->d_count, b->d_fvFeat, b->d_fvRunStart, b->d_fvRunNode,
                           b->d_bowMeta, d_kf_valid, b->d_pairIdx, cap, nnratio, checkOrientation, b->d_match, b->d_nmatch);
        LAUNCH_CHECK("k_search_by_bow");
    }
    b->nPairs = n_pairs; b->dlPairs = n_pairs;
    return SD_OK;
}

// The model fit of Tracking::TrackHomo (src/Tracking.cc:1026-1075) for every pair of the preceding
// sd_batch_search_by_projection: points_last / points_current -> H, F, inlier masks, the choice 1 (H) / 2 (F) / 0.
static int estimate_motion_impl(sd_batch* b, int n_pairs, void* stream_, const int* d_active, int minMatches);
int sd_batch_estimate_motion(sd_batch* b, void* stream_) { return estimate_motion_impl(b, b ? b->nPairs : 0, stream_, nullptr, 0); }

// d_active: see SdProjArgs; minMatches > 0: a pair whose matcher returned fewer matches gets flag 0 (TrackHomo's `nmatches<20`)
static int estimate_motion_impl(sd_batch* b, int n_pairs, void* stream_, const int* d_active, int minMatches)
{
    if (!b) return set_err(SD_ERR_INVALID, "null batch");
    b->nMotion = 0;
    if (n_pairs <= 0) return set_err(SD_ERR_STATE, "estimate_motion: no preceding sd_batch_search_by_projection");
    hipStream_t s = stream_ ? (hipStream_t)stream_ : b->lastStream;
    b->lastStream = s;
    const size_t nI = b->maxImages, cap = b->plan.kpCap;
    if (!b->d_moPts) {
        HIPCHK(hipMalloc((void**)&b->d_moPts, nI * cap * 16)); HIPCHK(hipMalloc((void**)&b->d_moNorm, nI * sizeof(SdMotionNorm)));
        HIPCHK(hipMalloc((void**)&b->d_moCounts, nI * SD_MOTION_K * 4)); HIPCHK(hipMalloc((void**)&b->d_moModels, nI * SD_MOTION_K * 72)); HIPCHK(hipMalloc((void**)&b->d_moMaskH, nI * cap));
        HIPCHK(hipMalloc((void**)&b->d_moMaskF, nI * cap)); HIPCHK(hipMalloc((void**)&b->d_moRes, nI * sizeof(SdMotionResult)));
    }
    {
        ProfScope ps(b, s, K_MOTION_P);
        hipLaunchKernelGGL(k_motion_prepare, dim3(n_pairs), dim3(256), cap * 16, s, KPUN(b), b->d_pairs, b->d_npairs, b->d_pairIdx, (int)cap, b->d_moPts, b->d_moNorm,
                           d_active, minMatches > 0 ? (const int*)b->d_nmatch : (const int*)nullptr, minMatches);
        LAUNCH_CHECK("k_motion_prepare");
    }
    {
        ProfScope ps(b, s, K_MOTION_H);
        for (int stage = 0; stage < 2; stage++) {
            const int nh = stage ? SD_MOTION_N1 : SD_MOTION_N0;
            hipLaunchKernelGGL(k_motion_models, dim3(nh / 64, n_pairs), dim3(64), 0, s, b->d_moPts, b->d_moNorm, (int)cap, b->d_moCounts, b->d_moModels, d_active, stage);
            LAUNCH_CHECK("k_motion_models");
            hipLaunchKernelGGL(k_motion_count, dim3(nh / SD_MOTION_HB, n_pairs), dim3(256), 0, s, b->d_moPts, b->d_moNorm, (int)cap, b->d_moCounts, b->d_moModels, d_active, stage);
            LAUNCH_CHECK("k_motion_count");
        }
    }
    {
        ProfScope ps(b, s, K_MOTION_S);
        hipLaunchKernelGGL(k_motion_select, dim3(n_pairs), dim3(256), 0, s, b->d_moPts, b->d_moNorm, b->d_moCounts, (int)cap, b->d_moMaskH, b->d_moMaskF, b->d_moRes, d_active);
        LAUNCH_CHECK("k_motion_select");
    }
    b->nMotion = n_pairs;
    return SD_OK;
}

int sd_batch_download_motion(sd_batch* b, int pair, double* H, double* F, uint8_t* mask_h, uint8_t* mask_f, int cap, int* n_points,
                             int* n_h, int* n_f, float* HorF, int* flag)
{
    if (!b || pair < 0 || pair >= b->nMotion) return SD_ERR_INVALID;
    int rc = sd_batch_sync(b);
    if (rc != SD_OK) return rc;
    SdMotionResult r;
    HIPCHK(hipMemcpy(&r, b->d_moRes + pair, sizeof(r), hipMemcpyDeviceToHost));
    int np = 0;
    HIPCHK(hipMemcpy(&np, b->d_npairs + pair, 4, hipMemcpyDeviceToHost));
    if (np > cap && (mask_h || mask_f)) return set_err(SD_ERR_CAPACITY, "mask buffer too small");
    if (H) memcpy(H, r.H, sizeof(r.H)); if (F) memcpy(F, r.F, sizeof(r.F)); if (HorF) memcpy(HorF, r.HorF, sizeof(r.HorF));
    if (n_h) *n_h = r.nH; if (n_f) *n_f = r.nF; if (flag) *flag = r.flag; if (n_points) *n_points = np;
    if (np > 0) {
        if (mask_h) HIPCHK(hipMemcpy(mask_h, b->d_moMaskH + (size_t)pair * b->plan.kpCap, np, hipMemcpyDeviceToHost));
        if (mask_f) HIPCHK(hipMemcpy(mask_f, b->d_moMaskF + (size_t)pair * b->plan.kpCap, np, hipMemcpyDeviceToHost));
    }
    return SD_OK;
}

// Frame::UndistortKeyPoints / ComputeImageBounds (src/Frame.cc:812-872) for cameras with distortion.
static SdDistortion to_distortion(const float* K4, const float* dist5)
{
    SdDistortion D;
    D.fx = K4[0]; D.fy = K4[1]; D.cx = K4[2]; D.cy = K4[3];
    D.k1 = dist5[0]; D.k2 = dist5[1]; D.p1 = dist5[2]; D.p2 = dist5[3]; D.k3 = dist5[4];
    return D;
}

int sd_undistort_points_device(const float* d_pts, int n, const float* K4, const float* dist5, float* d_out, void* stream)
{
    if (n < 0 || !K4 || !dist5 || (n > 0 && (!d_pts || !d_out))) return set_err(SD_ERR_INVALID, "bad undistort arguments");
    if (n == 0) return SD_OK;
    hipLaunchKernelGGL(k_undistort_points, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_pts, n, to_distortion(K4, dist5), d_out);
    LAUNCH_CHECK("k_undistort_points");
    return SD_OK;
}

int sd_batch_undistort_keypoints(sd_batch* b, int n_images, const float* K4, const float* dist5, sd_keypoint* d_keys_un, void* stream_)
{
    if (!b || n_images < 0 || n_images > b->nExtracted || !K4 || !dist5 || !d_keys_un) return set_err(SD_ERR_INVALID, "bad undistort_keypoints arguments");
    if (n_images == 0) return SD_OK;
    hipStream_t s = stream_ ? (hipStream_t)stream_ : b->lastStream;
    b->lastStream = s;
    const int cap = b->plan.kpCap;
    hipLaunchKernelGGL(k_undistort_keypoints, dim3((cap + 255) / 256, n_images), dim3(256), 0, s, b->d_kp, b->d_count, cap, to_distortion(K4, dist5),
                       dist5[0] == 0.0f ? 1 : 0, d_keys_un);                      // mDistCoef.at<float>(0) == 0.0 -> mvKeysUn = mvKeys
    LAUNCH_CHECK("k_undistort_keypoints");
    return SD_OK;
}

// mvKeysUn as a second key-point array of the batch: with a distortion set, grid, projection / local-map matchers, the RGB-D
// right coordinate, UnprojectStereo, the motion fit's point pairs and classifyH / classifyF read the undistorted key points, exactly
// where the reference reads mvKeysUn; box membership, the depth lookup and the stereo matcher keep reading mvKeys.
int sd_batch_set_distortion(sd_batch* b, const float* K4, const float* dist5)
{
    if (!b || !K4 || !dist5) return SD_ERR_INVALID;
    if (dist5[0] == 0.0f) { b->hasDist = false; return SD_OK; }       // mDistCoef.at<float>(0) == 0.0 -> mvKeysUn = mvKeys (Frame.cc:814-818)
    const size_t nI = b->maxImages, cap = b->plan.kpCap;
    if (!b->d_kpUn) {
        HIPCHK(hipMalloc((void**)&b->d_kpUn, nI * cap * sizeof(sd_keypoint)));
        HIPCHK(hipMalloc((void**)&b->d_kpDUn, nI * cap * sizeof(sd_keypoint)));
        HIPCHK(hipMalloc((void**)&b->d_unSlots, nI * 4));
    }
    b->dist = to_distortion(K4, dist5);
    b->hasDist = true;
    return SD_OK;
}

int sd_batch_undistort(sd_batch* b, int n_slots, const int32_t* slots, void* stream_)
{
    if (!b || n_slots < 0 || n_slots > b->maxImages || (n_slots > 0 && !slots)) return set_err(SD_ERR_INVALID, "bad undistort arguments");
    hipStream_t s = stream_ ? (hipStream_t)stream_ : b->lastStream;
    b->lastStream = s;
    if (!b->hasDist || n_slots == 0) return SD_OK;
    for (int i = 0; i < n_slots; i++) if (slots[i] < 0 || slots[i] >= b->maxImages) return set_err(SD_ERR_INVALID, "bad undistort slot");
    HIPCHK(hipMemcpyAsync(b->d_unSlots, slots, (size_t)n_slots * 4, hipMemcpyHostToDevice, s));
    const int cap = b->plan.kpCap;
    hipLaunchKernelGGL(k_undistort_slots, dim3((cap + 255) / 256, n_slots, 2), dim3(256), 0, s, b->d_kp, b->d_kpD, b->d_count, &b->d_fb[0].nDyn,
                       (int)(sizeof(SdFrameBoxes) / 4), b->d_unSlots, cap, b->dist, b->d_kpUn, b->d_kpDUn);
    LAUNCH_CHECK("k_undistort_slots");
    return SD_OK;
}

int sd_batch_download_keys_un(sd_batch* b, int image, sd_keypoint* kp, int cap, int* n)
{
    if (!b || !n || !slot_ok(b, image)) return SD_ERR_INVALID;
    int rc = sd_batch_sync(b);
    if (rc != SD_OK) return rc;
    int cnt = 0;
    HIPCHK(hipMemcpy(&cnt, b->d_count + image, 4, hipMemcpyDeviceToHost));
    *n = cnt;
    if (cnt > cap) return set_err(SD_ERR_CAPACITY, "keypoint buffer too small");
    if (cnt > 0 && kp) HIPCHK(hipMemcpy(kp, KPUN(b) + (size_t)image * b->plan.kpCap, (size_t)cnt * sizeof(sd_keypoint), hipMemcpyDeviceToHost));
    return SD_OK;
}

int sd_batch_download_dynamic_keys_un(sd_batch* b, int slot, sd_keypoint* kp, int cap, int* n)
{
    if (!b || !n || !slot_ok(b, slot)) return SD_ERR_INVALID;
    int rc = sd_batch_sync(b);
    if (rc != SD_OK) return rc;
    SdFrameBoxes h;
    HIPCHK(hipMemcpy(&h, b->d_fb + slot, sizeof(h), hipMemcpyDeviceToHost));
    *n = h.nDyn;
    if (h.nDyn > cap) return set_err(SD_ERR_CAPACITY, "dynamic keypoint buffer too small");
    if (h.nDyn > 0 && kp) HIPCHK(hipMemcpy(kp, KPDUN(b) + (size_t)slot * b->plan.kpCap, (size_t)h.nDyn * sizeof(sd_keypoint), hipMemcpyDeviceToHost));
    return SD_OK;
}

// Four corner points, once per camera: host arithmetic (double, the same expression order as the kernel).
int sd_image_bounds(int cols, int rows, const float* K4, const float* dist5, float* bounds4)
{
    if (!K4 || !dist5 || !bounds4 || cols < 1 || rows < 1) return SD_ERR_INVALID;
    if (dist5[0] != 0.0f) {
        const double fx = K4[0], fy = K4[1], cx = K4[2], cy = K4[3], k1 = dist5[0], k2 = dist5[1], p1 = dist5[2], p2 = dist5[3], k3 = dist5[4];
        const float corners[8] = {0.f, 0.f, (float)cols, 0.f, 0.f, (float)rows, (float)cols, (float)rows};
        float u[8];
        const double ifx = 1. / fx, ify = 1. / fy;
        for (int i = 0; i < 4; i++) {
            double x = corners[2 * i], y = corners[2 * i + 1];
            x = (x - cx) * ifx; y = (y - cy) * ify;
            const double x0 = x, y0 = y;
            for (int j = 0; j < 5; j++) {
                const double r2 = x * x + y * y;
                const double icdist = (1 + ((0. * r2 + 0.) * r2 + 0.) * r2) / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
                const double deltaX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
                const double deltaY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
                x = (x0 - deltaX) * icdist;
                y = (y0 - deltaY) * icdist;
            }
            u[2 * i] = (float)(x * fx + cx); u[2 * i + 1] = (float)(y * fy + cy);
        }
        bounds4[0] = std::min(u[0], u[4]); bounds4[1] = std::max(u[2], u[6]);
        bounds4[2] = std::min(u[1], u[3]); bounds4[3] = std::max(u[5], u[7]);
    } else {
        bounds4[0] = 0.0f; bounds4[1] = (float)cols; bounds4[2] = 0.0f; bounds4[3] = (float)rows;
    }
    return SD_OK;
}

// Frame copy (mLastFrame = Frame(mCurrentFrame), Tracking.cc; Frame.cc:39-63): keypoints, descriptors,
// stereo coordinates, grid cells and the map-point table of slot `src` into slot `dst`.
int sd_batch_copy_frame(sd_batch* b, int src, int dst, void* stream_)
{
    if (!b || !slot_ok(b, src) || dst < 0 || dst >= b->maxImages || src == dst) return set_err(SD_ERR_INVALID, "bad copy_frame slots");
    hipStream_t s = stream_ ? (hipStream_t)stream_ : b->lastStream;
    b->lastStream = s;
    const size_t cap = b->plan.kpCap;
#define CP(ptr, elemBytes) do { if (nseg < 24) { segs.src[nseg] = (const char*)(ptr) + src * cap * (elemBytes); segs.dst[nseg] = (char*)(ptr) + dst * cap * (elemBytes); segs.bytes[nseg] = (unsigned)(cap * (elemBytes)); nseg++; } } while (0)
    SdCopySegs segs;
    int nseg = 0;
    CP(b->d_kp, sizeof(sd_keypoint)); CP(b->d_desc, 32); CP(b->d_uright, 4); CP(b->d_depth, 4); CP(b->d_sad, 4);
    CP(b->d_cellOf, 2); CP(b->d_xw, 12); CP(b->d_flags, 1); CP(b->d_sortedIdx, 2);
    CP(b->d_kpD, sizeof(sd_keypoint)); CP(b->d_descD, 32); CP(b->d_urD, 4); CP(b->d_depD, 4);
    if (b->hasDist) { CP(b->d_kpUn, sizeof(sd_keypoint)); CP(b->d_kpDUn, sizeof(sd_keypoint)); }
#undef CP
#define CPX(ptr, elems, elemBytes) do { if (nseg < 24) { segs.src[nseg] = (const char*)((ptr) + (size_t)src * (elems)); segs.dst[nseg] = (char*)((ptr) + (size_t)dst * (elems)); segs.bytes[nseg] = (unsigned)((elems) * (elemBytes)); nseg++; } } while (0)
    CPX(b->d_cellStart, SD_GRID_CELLS + 8, 2);
    CPX(b->d_count, 1, 4);
    CPX(b->d_lvlCount, b->plan.nlevels, 4);
    CPX(b->d_fb, 1, sizeof(SdFrameBoxes));
    CPX(b->d_boxItems, b->itemsCap, 4);
#undef CPX
    segs.n = nseg;
    hipLaunchKernelGGL(k_copy_segments, dim3(32, nseg), dim3(256), 0, s, segs);      // one launch instead of 19 small copies
    LAUNCH_CHECK("k_copy_segments");
    b->slotValid[dst] = 1;
    if (!b->bowValid.empty()) b->bowValid[dst] = 0;      // mBowVec / mFeatVec are recomputed on demand (ComputeBoW's `if(mBowVec.empty())`)
    return SD_OK;
}

int sd_batch_matches_device(sd_batch* b, int32_t** d_match, int32_t** d_pairs, int32_t** d_npairs, int32_t** d_nmatches, int* cap)
{
    if (!b) return SD_ERR_INVALID;
    if (d_match) *d_match = b->d_match;
    if (d_pairs) *d_pairs = b->d_pairs;
    if (d_npairs) *d_npairs = b->d_npairs;
    if (d_nmatches) *d_nmatches = b->d_nmatch;
    if (cap) *cap = b->plan.kpCap;
    return SD_OK;
}

int sd_batch_download_matches(sd_batch* b, int pair, int32_t* match, int32_t* pairs, int cap, int* npairs, int* nmatches)
{
    if (!b || pair < 0 || pair >= b->dlPairs) return SD_ERR_INVALID;
    int rc = sd_batch_sync(b);
    if (rc != SD_OK) return rc;
    if (cap < b->plan.kpCap) return set_err(SD_ERR_CAPACITY, "match buffers need kp_capacity entries");
    int np = 0, nm = 0;
    HIPCHK(hipMemcpy(&np, b->d_npairs + pair, 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&nm, b->d_nmatch + pair, 4, hipMemcpyDeviceToHost));
    if (match) HIPCHK(hipMemcpy(match, b->d_match + (size_t)pair * b->plan.kpCap, (size_t)b->plan.kpCap * 4, hipMemcpyDeviceToHost));
    if (pairs && np > 0) HIPCHK(hipMemcpy(pairs, b->d_pairs + (size_t)pair * b->plan.kpCap * 2, (size_t)np * 8, hipMemcpyDeviceToHost));
    if (npairs) *npairs = np;
    if (nmatches) *nmatches = nm;
    return SD_OK;
}


// ---------------------------------------------------------------- dynamic-object cull
static SdCullPtrs cull_ptrs(sd_batch* b)
{
    SdCullPtrs A;
    A.kp = b->d_kp; A.desc = b->d_desc; A.uright = b->d_uright; A.depth = b->d_depth; A.count = b->d_count;
    A.kpT = b->d_kpT; A.descT = b->d_descT; A.urT = b->d_urT; A.depT = b->d_depT;
    A.kpD = b->d_kpD; A.descD = b->d_descD; A.urD = b->d_urD; A.depD = b->d_depD; A.kpDUn = KPDUN(b);
    A.fb = b->d_fb; A.boxItems = b->d_boxItems; A.cap = b->plan.kpCap; A.itemsCap = b->itemsCap; A.errFlag = b->d_err;
    return A;
}

namespace {
struct HRect { double x, y, w, h; bool empty() const { return w <= 0 || h <= 0; } double area() const { return w * h; } };
inline HRect hrect_and(HRect a, const HRect& b)
{
    const double x1 = a.x > b.x ? a.x : b.x, y1 = a.y > b.y ? a.y : b.y;
    const double x2 = a.x + a.w < b.x + b.w ? a.x + a.w : b.x + b.w, y2 = a.y + a.h < b.y + b.h ? a.y + a.h : b.y + b.h;
    HRect r = {x1, y1, x2 - x1, y2 - y1};
    if (r.w <= 0 || r.h <= 0) r = HRect{0, 0, 0, 0};
    return r;
}
inline HRect hrect_or(HRect a, const HRect& b)
{
    if (a.empty()) return b;
    if (b.empty()) return a;
    const double x1 = a.x < b.x ? a.x : b.x, y1 = a.y < b.y ? a.y : b.y;
    const double x2 = a.x + a.w > b.x + b.w ? a.x + a.w : b.x + b.w, y2 = a.y + a.h > b.y + b.h ? a.y + a.h : b.y + b.h;
    return HRect{x1, y1, x2 - x1, y2 - y1};
}
}  // namespace

// Frame::boxTrack (src/Frame.cc:481-552) — host code: a handful of boxes per frame, f64 arithmetic.
int sd_box_track(double* boxes, int n_box, int cap, const double* last_objects, int n_last, const int32_t* last_box_idx,
                 const uint8_t* last_omit, const double* last_velocity, int img_cols, int img_rows, int32_t* box_idx,
                 uint8_t* omit, double* velocity, int* n_out)
{
    if (!boxes || n_box < 0 || cap < n_box || n_last < 0 || !box_idx || !omit || !velocity || !n_out ||
        (n_last > 0 && (!last_objects || !last_box_idx || !last_omit || !last_velocity)))
        return set_err(SD_ERR_INVALID, "bad box_track arguments");
    std::vector<HRect> bx(n_box);
    for (int i = 0; i < n_box; i++) bx[i] = HRect{boxes[4 * i], boxes[4 * i + 1], boxes[4 * i + 2], boxes[4 * i + 3]};
    std::vector<int> idx(n_box, -1);
    std::vector<uint8_t> om(n_box, 0);
    std::vector<double> vel(2 * (size_t)n_box, 0.0);
    if (n_last > 0) {
        std::vector<HRect> lo(n_last);
        for (int i = 0; i < n_last; i++) lo[i] = HRect{last_objects[4 * i], last_objects[4 * i + 1], last_objects[4 * i + 2], last_objects[4 * i + 3]};
        for (int i = 0; i < n_last; i++) {           // greedy 1 - IoU association, later previous boxes may overwrite
            double minCost = 1;
            int best = -1;
            for (int j = 0; j < n_box; j++) {
                const double cost = 1 - hrect_and(lo[i], bx[j]).area() / hrect_or(lo[i], bx[j]).area();
                if (cost < minCost) { minCost = cost; best = j; }
            }
            if (best != -1 && !last_omit[i]) {
                idx[best] = last_box_idx[i];
                vel[2 * best] = bx[best].x + bx[best].w / 2 - lo[i].x - lo[i].w / 2;
                vel[2 * best + 1] = bx[best].y + bx[best].h / 2 - lo[i].y - lo[i].h / 2;
            }
        }
        for (int i = 0; i < n_last; i++) {           // re-inject an unmatched previous box once
            if (last_omit[i]) continue;
            bool found = false;
            for (size_t k = 0; k < idx.size(); k++) found |= idx[k] == last_box_idx[i];
            if (found) continue;
            const float cx = (float)(lo[i].x + lo[i].w / 2 + last_velocity[2 * i]);
            const float cy = (float)(lo[i].y + lo[i].h / 2 + last_velocity[2 * i + 1]);
            if (0.f <= cx && cx < (float)img_cols && 0.f <= cy && cy < (float)img_rows) {
                bx.push_back(HRect{lo[i].x + last_velocity[2 * i], lo[i].y + last_velocity[2 * i + 1], lo[i].w, lo[i].h});
                idx.push_back(last_box_idx[i]);
                om.push_back(1);
                vel.push_back(last_velocity[2 * i]); vel.push_back(last_velocity[2 * i + 1]);
            }
        }
        for (int i = 0; i < n_box; i++) {            // new ids for unmatched current boxes
            if (idx[i] != -1) continue;
            int mx = idx[0];
            for (size_t k = 1; k < idx.size(); k++) mx = idx[k] > mx ? idx[k] : mx;
            idx[i] = mx + 1;
        }
    } else {
        for (int i = 0; i < n_box; i++) idx[i] = i;
    }
    const int n = (int)bx.size();
    if (n > cap) return set_err(SD_ERR_CAPACITY, "box buffer too small for the re-injected boxes");
    for (int i = 0; i < n; i++) {
        boxes[4 * i] = bx[i].x; boxes[4 * i + 1] = bx[i].y; boxes[4 * i + 2] = bx[i].w; boxes[4 * i + 3] = bx[i].h;
        box_idx[i] = idx[i]; omit[i] = om[i]; velocity[2 * i] = vel[2 * i]; velocity[2 * i + 1] = vel[2 * i + 1];
    }
    *n_out = n;
    return SD_OK;
}

int sd_batch_first_separate(sd_batch* b, int n_frames, const int32_t* slots, const double* boxes, const int32_t* n_boxes,
                            const int32_t* box_idx, void* stream_)
{
    if (!b || n_frames < 0 || n_frames > b->maxImages || (n_frames > 0 && (!slots || !boxes || !n_boxes || !box_idx)))
        return set_err(SD_ERR_INVALID, "bad first_separate arguments");
    if (!b->cullOk) return set_err(SD_ERR_UNSUPPORTED, "more key points per image than the dynamic-object kernels' LDS tables hold (about 6,800)");
    hipStream_t s = stream_ ? (hipStream_t)stream_ : b->lastStream;
    b->lastStream = s;
    if (n_frames == 0) return SD_OK;
    std::vector<SdFrameBoxes>& h = b->hostBoxes;
    h.resize(n_frames);
    for (int f = 0; f < n_frames; f++) {
        if (!slot_ok(b, slots[f])) return set_err(SD_ERR_STATE, "first_separate: slot holds no results");
        if (n_boxes[f] < 0 || n_boxes[f] > SD_MAXB) return set_err(SD_ERR_CAPACITY, "more than SD_MAX_BOXES boxes in a frame");
        memset(&h[f], 0, sizeof(SdFrameBoxes));
        h[f].nb = n_boxes[f];
        for (int j = 0; j < n_boxes[f]; j++) {
            for (int k = 0; k < 4; k++) h[f].boxes[j][k] = boxes[((size_t)f * SD_MAXB + j) * 4 + k];
            h[f].box_idx[j] = box_idx[(size_t)f * SD_MAXB + j];
            h[f].box_status[j] = -1;
        }
    }
    HIPCHK(hipMemcpyAsync(b->d_fbStage, h.data(), (size_t)n_frames * sizeof(SdFrameBoxes), hipMemcpyHostToDevice, s));      // one copy; the workgroups scatter
    HIPCHK(hipMemcpyAsync(b->d_slots, slots, (size_t)n_frames * 4, hipMemcpyHostToDevice, s));
    {
        ProfScope ps(b, s, K_BOXSEP);
        hipLaunchKernelGGL(k_box_separate, dim3(n_frames), dim3(256), sd_box_separate_lds(b->plan.kpCap), s, cull_ptrs(b), b->d_slots, b->d_fbStage);
    }
    LAUNCH_CHECK("k_box_separate");
    return SD_OK;
}

int sd_batch_download_boxes(sd_batch* b, int slot, int* nb, double* boxes, int32_t* box_idx, int32_t* box_status, int32_t* kept_orig,
                            int32_t* box_start, int32_t* box_items, int items_cap, int* n_all, int* n_static)
{
    if (!b || !slot_ok(b, slot)) return SD_ERR_INVALID;
    int rc = sd_batch_sync(b);
    if (rc != SD_OK) return rc;
    SdFrameBoxes h;
    HIPCHK(hipMemcpy(&h, b->d_fb + slot, sizeof(h), hipMemcpyDeviceToHost));
    if (nb) *nb = h.nb;
    if (n_all) *n_all = h.nAll;
    if (n_static) *n_static = h.nOri;
    for (int j = 0; j < h.nb; j++) {
        if (boxes) for (int k = 0; k < 4; k++) boxes[4 * j + k] = h.boxes[j][k];
        if (box_idx) box_idx[j] = h.box_idx[j];
        if (box_status) box_status[j] = h.box_status[j];
        if (kept_orig) kept_orig[j] = h.keptOrig[j];
    }
    if (box_start) for (int j = 0; j <= h.nb; j++) box_start[j] = h.boxStart[j];
    if (box_items) {
        const int n = h.boxStart[h.nb];
        if (n > items_cap) return set_err(SD_ERR_CAPACITY, "box item buffer too small");
        if (n > 0) HIPCHK(hipMemcpy(box_items, b->d_boxItems + (size_t)slot * b->itemsCap, (size_t)n * 4, hipMemcpyDeviceToHost));
    }
    return SD_OK;
}

static_assert(sizeof(sd_frame_boxes) == sizeof(SdFrameBoxes) && offsetof(sd_frame_boxes, box_status) == offsetof(SdFrameBoxes, box_status) &&
              offsetof(sd_frame_boxes, box_start) == offsetof(SdFrameBoxes, boxStart), "sd_frame_boxes is the public face of SdFrameBoxes");
int sd_batch_boxes_device(sd_batch* b, sd_frame_boxes** d_frame_boxes)
{
    if (!b || !d_frame_boxes) return SD_ERR_INVALID;
    *d_frame_boxes = (sd_frame_boxes*)b->d_fb;
    return SD_OK;
}

int sd_batch_download_dynamic(sd_batch* b, int slot, sd_keypoint* kp, uint8_t* desc, float* uright, float* depth, int cap, int* n)
{
    if (!b || !slot_ok(b, slot) || !n) return SD_ERR_INVALID;
    int rc = sd_batch_sync(b);
    if (rc != SD_OK) return rc;
    SdFrameBoxes h;
    HIPCHK(hipMemcpy(&h, b->d_fb + slot, sizeof(h), hipMemcpyDeviceToHost));
    *n = h.nDyn;
    if (h.nDyn > cap) return set_err(SD_ERR_CAPACITY, "dynamic keypoint buffer too small");
    const size_t off = (size_t)slot * b->plan.kpCap;
    if (h.nDyn > 0) {
        if (kp) HIPCHK(hipMemcpy(kp, b->d_kpD + off, (size_t)h.nDyn * sizeof(sd_keypoint), hipMemcpyDeviceToHost));
        if (desc) HIPCHK(hipMemcpy(desc, b->d_descD + off * 32, (size_t)h.nDyn * 32, hipMemcpyDeviceToHost));
        if (uright) HIPCHK(hipMemcpy(uright, b->d_urD + off, (size_t)h.nDyn * 4, hipMemcpyDeviceToHost));
        if (depth) HIPCHK(hipMemcpy(depth, b->d_depD + off, (size_t)h.nDyn * 4, hipMemcpyDeviceToHost));
    }
    return SD_OK;
}

static int separate_impl(sd_batch* b, int n_pairs, const int32_t* cur_index, const int32_t* ref_index, const float* HorF,
                         const int32_t* flag, const int32_t* last_box_idx, const int32_t* last_box_status, const int32_t* n_last,
                         void* stream_, const int32_t* last_slot, const int* d_active);
int sd_batch_separate(sd_batch* b, int n_pairs, const int32_t* cur_index, const int32_t* ref_index, const float* HorF,
                      const int32_t* flag, const int32_t* last_box_idx, const int32_t* last_box_status, const int32_t* n_last,
                      void* stream_)
{
    return separate_impl(b, n_pairs, cur_index, ref_index, HorF, flag, last_box_idx, last_box_status, n_last, stream_, nullptr, nullptr);
}

// last_slot (host, nullable): slot of mLastFrame per pair, replaces last_box_idx / last_box_status / n_last; d_active: see SdSepArgs
static int separate_impl(sd_batch* b, int n_pairs, const int32_t* cur_index, const int32_t* ref_index, const float* HorF,
                         const int32_t* flag, const int32_t* last_box_idx, const int32_t* last_box_status, const int32_t* n_last,
                         void* stream_, const int32_t* last_slot, const int* d_active)
{
    if (!b || n_pairs < 0 || n_pairs > b->maxImages ||
        (n_pairs > 0 && (!cur_index || !ref_index || (!HorF) != (!flag) || (!last_slot && (!last_box_idx || !last_box_status || !n_last)))))
        return set_err(SD_ERR_INVALID, "bad separate arguments");
    if (!b->cullOk) return set_err(SD_ERR_UNSUPPORTED, "more key points per image than the dynamic-object kernels' LDS tables hold (about 6,800)");
    const bool fromMotion = n_pairs > 0 && !HorF;          // HorF == flag == NULL: pair p uses the model fit of pair p
    if (fromMotion && b->nMotion < n_pairs) return set_err(SD_ERR_STATE, "separate: no sd_batch_estimate_motion results for these pairs");
    hipStream_t s = stream_ ? (hipStream_t)stream_ : b->lastStream;
    b->lastStream = s;
    b->nSepPairs = 0;
    if (n_pairs == 0) return SD_OK;
    std::vector<int2>& idx = b->hostPairs;
    idx.resize(n_pairs);
    for (int p = 0; p < n_pairs; p++) {
        if (!slot_ok(b, cur_index[p]) || !slot_ok(b, ref_index[p])) return set_err(SD_ERR_STATE, "separate: slot holds no results");
        if (!fromMotion && flag[p] != 1 && flag[p] != 2) return set_err(SD_ERR_INVALID, "separate: flag must be 1 (H) or 2 (F)");
        if (last_slot ? !slot_ok(b, last_slot[p]) : (n_last[p] < 0 || n_last[p] > SD_MAXB)) return set_err(SD_ERR_INVALID, "separate: bad n_last / last slot");
        idx[p] = make_int2(cur_index[p], ref_index[p]);
    }
    HIPCHK(hipMemcpyAsync(b->d_sepPairs, idx.data(), (size_t)n_pairs * sizeof(int2), hipMemcpyHostToDevice, s));
    if (fromMotion) {
        hipLaunchKernelGGL(k_motion_to_sep, dim3((n_pairs + 63) / 64), dim3(64), 0, s, b->d_moRes, b->d_HorF, b->d_sepFlag, n_pairs, d_active);
        LAUNCH_CHECK("k_motion_to_sep");
    } else {
        HIPCHK(hipMemcpyAsync(b->d_HorF, HorF, (size_t)n_pairs * 36, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(b->d_sepFlag, flag, (size_t)n_pairs * 4, hipMemcpyHostToDevice, s));
    }
    if (last_slot) {
        HIPCHK(hipMemcpyAsync(b->d_nLast, last_slot, (size_t)n_pairs * 4, hipMemcpyHostToDevice, s));      // d_nLast doubles as the slot list
    } else {
        HIPCHK(hipMemcpyAsync(b->d_lastIdx, last_box_idx, (size_t)n_pairs * SD_MAXB * 4, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(b->d_lastStatus, last_box_status, (size_t)n_pairs * SD_MAXB * 4, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(b->d_nLast, n_last, (size_t)n_pairs * 4, hipMemcpyHostToDevice, s));
    }
    SdSepArgs G;
    G.pairIdx = b->d_sepPairs; G.HorF = b->d_HorF; G.flag = b->d_sepFlag; G.lastIdx = b->d_lastIdx; G.lastStatus = b->d_lastStatus;
    G.nLast = b->d_nLast; G.dynStart = b->d_dynStart; G.dynStatus = b->d_dynStatus; G.matches = b->d_sepMatches; G.ret = b->d_sepRet;
    G.lastSlot = last_slot ? b->d_nLast : nullptr; G.active = d_active;
    b->sepActive = d_active;
    {
        ProfScope ps(b, s, K_SEPARATE);
        hipLaunchKernelGGL(k_separate, dim3(n_pairs), dim3(256), sd_separate_lds(b->plan.kpCap), s, cull_ptrs(b), G);
    }
    LAUNCH_CHECK("k_separate");
    b->nSepPairs = n_pairs;
    return SD_OK;
}

int sd_batch_download_separate(sd_batch* b, int pair, int32_t* ret, int32_t* dyn_start, int32_t* dyn_status, int32_t* matches, int cap)
{
    if (!b || pair < 0 || pair >= b->nSepPairs) return SD_ERR_INVALID;
    int rc = sd_batch_sync(b);
    if (rc != SD_OK) return rc;
    int ds[SD_MAXB + 1];
    HIPCHK(hipMemcpy(ds, b->d_dynStart + (size_t)pair * (SD_MAXB + 1), sizeof(ds), hipMemcpyDeviceToHost));
    if (dyn_start) memcpy(dyn_start, ds, sizeof(ds));
    const int n = ds[SD_MAXB];
    if (n > cap) return set_err(SD_ERR_CAPACITY, "dynStatus buffer too small");
    if (ret) HIPCHK(hipMemcpy(ret, b->d_sepRet + pair, 4, hipMemcpyDeviceToHost));
    if (n > 0) {
        if (dyn_status) HIPCHK(hipMemcpy(dyn_status, b->d_dynStatus + (size_t)pair * b->itemsCap, (size_t)n * 4, hipMemcpyDeviceToHost));
        if (matches) HIPCHK(hipMemcpy(matches, b->d_sepMatches + (size_t)pair * b->itemsCap * 2, (size_t)n * 8, hipMemcpyDeviceToHost));
    }
    return SD_OK;
}

int sd_batch_update_frame(sd_batch* b, int only_if_static, void* stream_)
{
    if (!b) return SD_ERR_INVALID;
    if (b->nSepPairs <= 0) return set_err(SD_ERR_STATE, "update_frame needs a preceding separate");
    hipStream_t s = stream_ ? (hipStream_t)stream_ : b->lastStream;
    b->lastStream = s;
    SdSepArgs G;
    G.pairIdx = b->d_sepPairs; G.HorF = b->d_HorF; G.flag = b->d_sepFlag; G.lastIdx = b->d_lastIdx; G.lastStatus = b->d_lastStatus;
    G.nLast = b->d_nLast; G.dynStart = b->d_dynStart; G.dynStatus = b->d_dynStatus; G.matches = b->d_sepMatches; G.ret = b->d_sepRet;
    G.lastSlot = nullptr; G.active = b->sepActive;
    {
        ProfScope ps(b, s, K_UPDATE);
        const size_t lds = (size_t)b->itemsCap * 4 + (size_t)b->plan.kpCap + 64;
        hipLaunchKernelGGL(k_update_frame, dim3(b->nSepPairs), dim3(256), lds, s, cull_ptrs(b), G,
                           only_if_static ? (const int*)b->d_sepRet : (const int*)nullptr);
    }
    LAUNCH_CHECK("k_update_frame");
    return SD_OK;
}


// ---------------------------------------------------------------- reference-frame queue (Tracking.cc:620-666, 952-959)
struct sd_refqueue {
    struct E { double t; int slot; int hasBoxes; };
    std::vector<E> q;      // front = oldest
};

int sd_refqueue_create(sd_refqueue** out) { if (!out) return SD_ERR_INVALID; *out = new sd_refqueue(); return SD_OK; }
int sd_refqueue_destroy(sd_refqueue* q) { delete q; return SD_OK; }
int sd_refqueue_clear(sd_refqueue* q) { if (!q) return SD_ERR_INVALID; q->q.clear(); return SD_OK; }   // :602-605
int sd_refqueue_size(const sd_refqueue* q, int* n) { if (!q || !n) return SD_ERR_INVALID; *n = (int)q->q.size(); return SD_OK; }

// The `while(mCurrentFrame.mTimeStamp - q_frame.front().mTimeStamp > 0.2f)` head of the loop (:623-631): drops
// box-less fronts, then yields the oldest frame more than 0.2 s older than the current one, or -1.
int sd_refqueue_candidate(sd_refqueue* q, double cur_timestamp, int cur_has_boxes, int* slot)
{
    if (!q || !slot) return SD_ERR_INVALID;
    *slot = -1;
    if (!cur_has_boxes) return SD_OK;                              // `!mCurrentFrame.objects.empty()` (:622)
    while (!q->q.empty() && cur_timestamp - q->q.front().t > 0.2f) {
        if (!q->q.front().hasBoxes) { q->q.erase(q->q.begin()); continue; }   // reference: no emptiness re-check (UB)
        *slot = q->q.front().slot;
        return SD_OK;
    }
    return SD_OK;
}

// TrackHomo failed on the candidate (:655-661): pop it unless it is the last element; *again = loop continues.
int sd_refqueue_reject(sd_refqueue* q, int* again)
{
    if (!q || !again) return SD_ERR_INVALID;
    *again = 0;
    if (q->q.size() <= 1) return SD_OK;
    q->q.erase(q->q.begin());
    *again = 1;
    return SD_OK;
}

// After tracking (mState == OK): `if(q_frame.size() >= mMaxFrames * 0.3) q_frame.pop(); q_frame.push(cur)` (:952-959).
// *evicted_slot = slot of the popped frame (its batch slot can be reused) or -1.
int sd_refqueue_push(sd_refqueue* q, double timestamp, int slot, int has_boxes, int max_frames, int* evicted_slot)
{
    if (!q) return SD_ERR_INVALID;
    if (evicted_slot) *evicted_slot = -1;
    if ((double)q->q.size() >= max_frames * 0.3 && !q->q.empty()) {
        if (evicted_slot) *evicted_slot = q->q.front().slot;
        q->q.erase(q->q.begin());
    }
    q->q.push_back({timestamp, slot, has_boxes});
    return SD_OK;
}


// ---------------------------------------------------------------- profiling
int sd_batch_set_profiling(sd_batch* b, int enabled)
{
    if (!b) return SD_ERR_INVALID;
    b->profiling = enabled != 0;
    return SD_OK;
}
int sd_batch_kernel_count(const sd_batch* b, int* n)
{
    if (!b || !n) return SD_ERR_INVALID;
    *n = K_COUNT;
    return SD_OK;
}
int sd_batch_kernel_times(sd_batch* b, int index, const char** name, double* total_ms, int64_t* launches)
{
    if (!b || index < 0 || index >= K_COUNT) return SD_ERR_INVALID;
    drain_profile(b);
    if (name) *name = kKernelNames[index];
    if (total_ms) *total_ms = b->totalMs[index];
    if (launches) *launches = b->launches[index];
    return SD_OK;
}
int sd_batch_reset_kernel_times(sd_batch* b)
{
    if (!b) return SD_ERR_INVALID;
    drain_profile(b);
    for (int i = 0; i < K_COUNT; i++) { b->totalMs[i] = 0; b->launches[i] = 0; }
    return SD_OK;
}

// PointCloudMapping::generatePointCloud (src/pointcloudmapping.cc:59-103) for frame slots of the batch
int sd_batch_backproject_dense(sd_batch* b, int n_frames, const int32_t* slots, const uint8_t* d_color, size_t color_stride,
                               size_t color_pitch, const uint16_t* d_depth, size_t depth_stride_elems, size_t depth_pitch_elems,
                               float depth_factor, const uint8_t* d_mask, size_t mask_stride, size_t mask_pitch, const sd_camera* cam,
                               const double* Twc_host, sd_cloud_point* d_points, int cap_points, int32_t* d_counts, void* stream_)
{
    if (!b || n_frames < 0 || n_frames > b->maxImages || !cam_ok(cam) || (n_frames > 0 && (!slots || !d_color || !d_depth || !Twc_host || !d_points || !d_counts)))
        return set_err(SD_ERR_INVALID, "bad backproject_dense arguments");
    const int W = b->plan.W, H = b->plan.H;
    const int cols3 = (W + 2) / 3, rows3 = (H + 2) / 3, words = (cols3 + 63) / 64;
    if (cap_points < cols3 * rows3) return set_err(SD_ERR_CAPACITY, "cap_points must be at least ceil(W/3) * ceil(H/3)");
    if (words > 64) return set_err(SD_ERR_UNSUPPORTED, "image wider than 12288 pixels");
    if (color_stride < (size_t)3 * W || depth_stride_elems < (size_t)W || (d_mask && mask_stride < (size_t)W)) return set_err(SD_ERR_INVALID, "stride smaller than width");
    hipStream_t s = stream_ ? (hipStream_t)stream_ : b->lastStream;
    b->lastStream = s;
    if (n_frames == 0) return SD_OK;
    for (int f = 0; f < n_frames; f++) if (!slot_ok(b, slots[f])) return set_err(SD_ERR_STATE, "backproject_dense: slot holds no frame");
    const size_t need = (size_t)b->maxImages * rows3 * words;
    if (b->cloudCap < need) {
        HIPCHK(hipStreamSynchronize(s));
        if (b->d_cloudBits) { (void)hipFree(b->d_cloudBits); (void)hipFree(b->d_cloudRows); (void)hipFree(b->d_cloudT); (void)hipFree(b->d_cloudSlots); b->d_cloudBits = nullptr; b->d_cloudRows = nullptr; b->d_cloudT = nullptr; b->d_cloudSlots = nullptr; }
        HIPCHK(hipMalloc((void**)&b->d_cloudBits, need * 8));
        HIPCHK(hipMalloc((void**)&b->d_cloudRows, (size_t)b->maxImages * rows3 * 2 * 4));
        HIPCHK(hipMalloc((void**)&b->d_cloudT, (size_t)b->maxImages * 16 * 8));
        HIPCHK(hipMalloc((void**)&b->d_cloudSlots, (size_t)b->maxImages * 4));
        b->cloudCap = need;
    }
    HIPCHK(hipMemcpyAsync(b->d_cloudT, Twc_host, (size_t)n_frames * 128, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(b->d_cloudSlots, slots, (size_t)n_frames * 4, hipMemcpyHostToDevice, s));
    SdCloudArgs A;
    A.fb = b->d_fb; A.slots = b->d_cloudSlots; A.color = d_color; A.colorStride = color_stride; A.colorPitch = color_pitch;
    A.depth = d_depth; A.depthStride = depth_stride_elems; A.depthPitch = depth_pitch_elems; A.depthFactor = depth_factor;
    A.mask = d_mask; A.maskStride = mask_stride; A.maskPitch = mask_pitch; A.fx = cam->fx; A.fy = cam->fy; A.cx = cam->cx; A.cy = cam->cy;
    A.Twc = b->d_cloudT; A.W = W; A.H = H; A.cols3 = cols3; A.rows3 = rows3; A.words = words; A.bits = b->d_cloudBits; A.rowCount = b->d_cloudRows;
    A.points = (sd_cloud_point_dev*)d_points; A.capPoints = cap_points; A.counts = d_counts;
    hipLaunchKernelGGL(k_cloud_mark, dim3(rows3, n_frames), dim3(256), 0, s, A);
    LAUNCH_CHECK("k_cloud_mark");
    hipLaunchKernelGGL(k_cloud_emit, dim3(rows3, n_frames), dim3(256), 0, s, A);
    LAUNCH_CHECK("k_cloud_emit");
    return SD_OK;
}

#include "sd_tracker.inc"

} // extern "C"
