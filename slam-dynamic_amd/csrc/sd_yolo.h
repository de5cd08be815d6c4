// Host side of the detector behind the C ABI (included by sd_api.hip): network description, Darknet weight
// loading with batch-norm folding, forward pass orchestration, and the reference's post-processing.
//   yolov3Segment::yolov3Segment / readNetFromDarknet     src/yolo.cc:15-31
//   yolov3Segment::Segmentation_                          src/yolo.cc:60-77
//   yolov3Segment::postprocess_ + rectCenterScale         src/yolo.cc:142-206
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>
#include "k_yolo.h"
#include "k_yolo32.h"
#include "k_yolo32w.h"
#include "k_yolo32b.h"

struct sd_yolo {
    std::vector<sd_yolo_layer> L;
    struct Rt { int H = 0, W = 0, C = 0; int cinPad = 0, coutPad = 0; size_t wOff = 0, bOff = 0; _Float16* out = nullptr; int outC = 0; bool alias = false;
                bool wino = false; size_t wOffW = 0;
                bool b3 = false, b3flat = false; int b3wm = 2; size_t wOffB = 0; };          // SD_YOLO_F32X3: this layer runs on bf16 limbs (k_yolo32b.h), its split weights at d_wgtB + wOffB (16-byte units)        // SD_YOLO_F32W: this layer runs as Winograd F(2x2, 3x3), its transformed weights at d_wgtW + wOffW
    std::vector<Rt> R;
    int netW = 0, netH = 0, classes = 80, maxBatch = 0, nconv = 0;
    int f32 = 0;                   // SD_YOLO_F32: activations / weights / arithmetic in f32 (k_yolo32.h); the `out` pointers then hold floats
    float* d_blob8 = nullptr; float* d_wgt32 = nullptr; bool attrF32 = false, attrNms = false;
    int wino = 0;                  // SD_YOLO_F32W (k_yolo32w.h): f32 mode with the eligible 3 x 3 stride-1 layers as Winograd F(2x2, 3x3)
    float* d_wgtW = nullptr; float* d_V = nullptr; size_t wTotalW = 0; bool attrWino = false;
    int b3 = 0;                    // SD_YOLO_F32X3 (k_yolo32b.h): f32 mode with the >= 128-filter layers on three bf16 limbs per operand
    uint4* d_wgtB = nullptr; size_t wTotalB = 0; bool attrB3 = false;
    double mfmaFlopsBf16 = 0;      // per image: bf16 MFMA FLOPs executed by the limb kernels (six limb products per product)
    double mfmaFlops = 0;          // per image, as executed (Winograd layers: 16 multiplies per 2 x 2 block instead of 36)
    float anchors[18];
    _Float16* d_blob4 = nullptr;   // network input, NHWC f16 x 4 channels
    _Float16* d_wgt = nullptr; float* d_bias = nullptr; _Float16* d_zero = nullptr;
    short4* d_ct = nullptr; short4* d_rt = nullptr;
    SdDet* d_dets = nullptr; int* d_ndet = nullptr; float* d_raw = nullptr;
    uint8_t* d_hostImg = nullptr; size_t hostImgCap = 0; uint8_t* d_hostMask = nullptr; size_t hostMaskCap = 0;      // sd_yolo_forward_host / mask_host
    double* d_nmsBoxes = nullptr; int* d_nmsCls = nullptr; float* d_nmsConf = nullptr; int* d_nmsN = nullptr;      // sd_yolo_boxes_batch
    int detCap = 0, totalRows = 0;
    int tabW = 0, tabH = 0;
    size_t wTotal = 0, bTotal = 0;
    bool weightsLoaded = false;
    bool attrGlds = false, attrFlat3 = false;      // dynamic-LDS limits of the convolution kernels raised on this device
    int lastN = 0;
    hipStream_t stream = nullptr;
    // overlap mode (sd_yolo_set_overlap, f32-class modes): blobFromImage of a pass on sPre, the region decodes on sPost, ordered against the convolution
    // stream by events, so that with two passes enqueued the next pass's convolutions start while this pass's decode / NMS / download still run
    bool overlap = false, haveL0 = false, haveDecoded = false, haveNms = false;
    hipStream_t sPre = nullptr, sPost = nullptr;
    hipEvent_t evBlob = nullptr, evL0 = nullptr, evHead[3] = {nullptr, nullptr, nullptr}, evDecoded = nullptr, evNms = nullptr;
    // 2-input [route]s whose second input is written in place by its producer (f32-class modes): only the up-sampled half is copied
    std::vector<void*> owned;
    double convFlops = 0;     // per image
};

static const float kYoloV3Anchors[18] = {10, 13, 16, 30, 33, 23, 30, 61, 62, 45, 59, 119, 116, 90, 156, 198, 373, 326};

// The layer list of src/yolo/yolov3.cfg (107 layers), restated programmatically.
static void yolo_v3_layers(std::vector<sd_yolo_layer>& L)
{
    auto conv = [&](int filters, int size, int stride, int bn = 1, int leaky = 1) {
        sd_yolo_layer l = {}; l.type = SD_YOLO_CONV; l.filters = filters; l.size = size; l.stride = stride; l.batch_normalize = bn; l.leaky = leaky;
        L.push_back(l);
    };
    auto shortcut = [&](int from) { sd_yolo_layer l = {}; l.type = SD_YOLO_SHORTCUT; l.from[0] = from; l.nfrom = 1; L.push_back(l); };
    auto route = [&](int a, int b = 0, int n = 1) { sd_yolo_layer l = {}; l.type = SD_YOLO_ROUTE; l.from[0] = a; l.from[1] = b; l.nfrom = n; L.push_back(l); };
    auto upsample = [&]() { sd_yolo_layer l = {}; l.type = SD_YOLO_UPSAMPLE; l.stride = 2; L.push_back(l); };
    auto yolo = [&](int m0, int m1, int m2) { sd_yolo_layer l = {}; l.type = SD_YOLO_YOLO; l.mask[0] = m0; l.mask[1] = m1; l.mask[2] = m2; L.push_back(l); };
    auto res = [&](int c, int n) { for (int i = 0; i < n; i++) { conv(c / 2, 1, 1); conv(c, 3, 1); shortcut(-3); } };
    conv(32, 3, 1);
    conv(64, 3, 2); res(64, 1);
    conv(128, 3, 2); res(128, 2);
    conv(256, 3, 2); res(256, 8);
    conv(512, 3, 2); res(512, 8);
    conv(1024, 3, 2); res(1024, 4);
    for (int i = 0; i < 3; i++) { conv(512, 1, 1); conv(1024, 3, 1); }
    conv(255, 1, 1, 0, 0); yolo(6, 7, 8);
    route(-4); conv(256, 1, 1); upsample(); route(-1, 61, 2);
    for (int i = 0; i < 3; i++) { conv(256, 1, 1); conv(512, 3, 1); }
    conv(255, 1, 1, 0, 0); yolo(3, 4, 5);
    route(-4); conv(128, 1, 1); upsample(); route(-1, 36, 2);
    for (int i = 0; i < 3; i++) { conv(128, 1, 1); conv(256, 3, 1); }
    conv(255, 1, 1, 0, 0); yolo(0, 1, 2);
}

static inline int yolo_resolve(int idx, int from) { return from < 0 ? idx + from : from; }

// cv::resize INTER_LINEAR coefficient tables (same fixed-point scheme as the pyramid, sd_plan.h)
static void yolo_resize_tables(int sw, int sh, int dw, int dh, std::vector<int16_t>& ct, std::vector<int16_t>& rt)
{
    const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    ct.assign(4 * (size_t)dw, 0); rt.assign(4 * (size_t)dh, 0);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        ct[4 * dx] = (int16_t)sx; ct[4 * dx + 1] = (int16_t)lrintf((1.f - fx) * 2048); ct[4 * dx + 2] = (int16_t)lrintf(fx * 2048);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        rt[4 * dy] = (int16_t)sy; rt[4 * dy + 1] = (int16_t)lrintf((1.f - fy) * 2048); rt[4 * dy + 2] = (int16_t)lrintf(fy * 2048);
    }
}

struct YRect { int x, y, w, h; };
static inline float yolo_overlap(const YRect& a, const YRect& b)
{
    // 1 - jaccardDistance(a, b) of cv::Rect (integer areas, double ratio)
    const double Aa = (double)a.w * a.h, Ab = (double)b.w * b.h;
    if ((Aa + Ab) <= 2.220446049250313e-16) return 1.f;
    const int x1 = std::max(a.x, b.x), y1 = std::max(a.y, b.y);
    const int x2 = std::min(a.x + a.w, b.x + b.w), y2 = std::min(a.y + a.h, b.y + b.h);
    const double Aab = (x2 > x1 && y2 > y1) ? (double)(x2 - x1) * (y2 - y1) : 0.0;
    return (float)(1. - (1. - Aab / (Aa + Ab - Aab)));
}
