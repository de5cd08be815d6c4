// The detector part of the C ABI (include/sd_frontend.h, sd_yolo_*): its own translation unit, so that a change to a front-end kernel
// does not recompile the convolution stack and vice versa.  Kernels: k_yolo.h (f16 mode), k_yolo32.h (f32 mode).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstddef>
#include <cmath>
#include <string>
#include <vector>
#include "sd_common.h"
#include "sd_yolo.h"

extern "C" {

// ---------------------------------------------------------------- detector (YOLOv3 on MFMA)
int sd_yolo_v3_layers(sd_yolo_layer* layers, int cap, int* n, float anchors[18])
{
    if (!n) return SD_ERR_INVALID;
    std::vector<sd_yolo_layer> L;
    yolo_v3_layers(L);
    *n = (int)L.size();
    if (anchors) memcpy(anchors, kYoloV3Anchors, sizeof(kYoloV3Anchors));
    if (layers) {
        if (cap < (int)L.size()) return set_err(SD_ERR_CAPACITY, "layer buffer too small");
        memcpy(layers, L.data(), L.size() * sizeof(sd_yolo_layer));
    }
    return SD_OK;
}

static void yolo_free(sd_yolo* y)
{
    if (!y) return;
    for (void* p : y->owned) if (p) (void)hipFree(p);
    if (y->d_hostImg) (void)hipFree(y->d_hostImg);
    if (y->d_hostMask) (void)hipFree(y->d_hostMask);
    if (y->stream) (void)hipStreamDestroy(y->stream);
    if (y->sPre) (void)hipStreamDestroy(y->sPre);
    if (y->sPost) (void)hipStreamDestroy(y->sPost);
    hipEvent_t evs[] = {y->evBlob, y->evL0, y->evHead[0], y->evHead[1], y->evHead[2], y->evDecoded, y->evNms};
    for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
    delete y;
}

int sd_yolo_create(sd_yolo** out, const sd_yolo_layer* layers, int n_layers, const float anchors[18], int classes, int net_w,
                   int net_h, int max_batch)
{
    return sd_yolo_create_prec(out, layers, n_layers, anchors, classes, net_w, net_h, max_batch, SD_YOLO_F16);
}

int sd_yolo_create_prec(sd_yolo** out, const sd_yolo_layer* layers, int n_layers, const float anchors[18], int classes, int net_w,
                        int net_h, int max_batch, int precision)
{
    if (!out) return SD_ERR_INVALID;
    *out = nullptr;
    if (precision != SD_YOLO_F16 && precision != SD_YOLO_F32 && precision != SD_YOLO_F32W && precision != SD_YOLO_F32X3)
        return set_err(SD_ERR_INVALID, "precision must be SD_YOLO_F16, SD_YOLO_F32, SD_YOLO_F32W or SD_YOLO_F32X3");
    if (!layers || n_layers < 1 || !anchors || classes != 80 || net_w < 32 || net_h < 32 || (net_w % 32) || (net_h % 32) || max_batch < 1)
        return set_err(SD_ERR_INVALID, "bad detector arguments (classes must be 80, net size a multiple of 32)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return set_err(SD_ERR_NO_DEVICE, "no HIP device: the detector has no CPU fallback");
    sd_yolo* y = new sd_yolo();
    y->L.assign(layers, layers + n_layers);
    y->R.resize(n_layers);
    y->netW = net_w; y->netH = net_h; y->classes = classes; y->maxBatch = max_batch;
    y->f32 = precision == SD_YOLO_F32 || precision == SD_YOLO_F32W || precision == SD_YOLO_F32X3;
    y->wino = precision == SD_YOLO_F32W;
    y->b3 = precision == SD_YOLO_F32X3;
    size_t wOffW = 0, vMax = 0, wOffB = 0;
    const size_t eb = y->f32 ? 4 : 2;                     // bytes per activation element
    memcpy(y->anchors, anchors, sizeof(y->anchors));
    // ---- shapes
    int H = net_h, W = net_w, C = 32;      // blob: 3 channels padded to 32
    size_t wOff = 0, bOff = 0;
    for (int i = 0; i < n_layers; i++) {
        const sd_yolo_layer& l = y->L[i];
        sd_yolo::Rt& r = y->R[i];
        if (l.type == SD_YOLO_CONV) {
            if ((l.size != 1 && l.size != 3) || (l.stride != 1 && l.stride != 2) || l.filters < 1) { delete y; return set_err(SD_ERR_UNSUPPORTED, "convolution size/stride not supported"); }
            if (i == 0 && (l.size != 3 || l.stride != 1 || l.filters > 32)) { delete y; return set_err(SD_ERR_UNSUPPORTED, "first convolution must be 3x3, stride 1, <= 32 filters"); }
            const int cinReal = i == 0 ? 3 : C;
            r.cinPad = i == 0 ? 32 : C;
            if (r.cinPad % 32) { delete y; return set_err(SD_ERR_UNSUPPORTED, "input channels must be a multiple of 32"); }
            const int pad = l.size / 2;
            r.H = (H + 2 * pad - l.size) / l.stride + 1; r.W = (W + 2 * pad - l.size) / l.stride + 1; r.C = l.filters;
            r.outC = (l.filters + 31) / 32 * 32;            // stored channel count (255 -> 256)
            r.coutPad = (l.filters + SD_G3_BM - 1) / SD_G3_BM * SD_G3_BM;        // weight/bias rows are padded to the widest filter tile
            r.wOff = wOff; r.bOff = bOff;
            wOff += (size_t)r.coutPad * l.size * l.size * r.cinPad;
            bOff += r.coutPad;
            y->convFlops += 2.0 * r.H * r.W * (double)l.filters * l.size * l.size * cinReal;
            // Winograd F(2x2, 3x3): 3 x 3, stride 1, >= 64 input channels (the fold runs once per cin channels) and whole 128-filter tiles
            r.wino = y->wino && i > 0 && l.size == 3 && l.stride == 1 && r.cinPad >= 64 && (r.cinPad % 16) == 0 && l.filters >= 128 && (l.filters % 128) == 0;
            if (r.wino) {
                const size_t blocks = (size_t)((r.H + 1) / 2) * ((r.W + 1) / 2);
                r.wOffW = wOffW; wOffW += (size_t)r.coutPad * 16 * r.cinPad;
                vMax = std::max(vMax, blocks * 16 * r.cinPad);
                y->mfmaFlops += 2.0 * blocks * 16.0 * (double)l.filters * cinReal;
            } else if (y->b3 && i > 0 && l.filters >= 64 && (l.filters > 64 || (l.filters % 64) == 0) && (r.cinPad % 16) == 0) {
                // three bf16 limbs per operand: the layers k_conv_f32 runs on 128-filter tiles
                r.b3 = true;
                r.b3flat = l.size == 3 && l.stride == 1 && r.W <= 160 && l.filters > 64;       // k_conv3x3_b3: the nine taps share one staged chunk
                r.b3wm = (r.b3flat && r.W <= 80) || l.filters == 64 ? 1 : 2;                      // 1: k_conv3x3_b3c (64-filter tiles, weights staged per chunk too); wider maps do not fit its LDS
                r.wOffB = wOffB; wOffB += (size_t)(r.coutPad / 128) * (l.size * l.size * (r.cinPad / 16)) * 2 * 6 * 64;      // 16-byte fragments
                y->mfmaFlopsBf16 += 6 * 2.0 * r.H * r.W * (double)l.filters * l.size * l.size * cinReal;
            } else y->mfmaFlops += 2.0 * r.H * r.W * (double)l.filters * l.size * l.size * cinReal;
            y->nconv++;
        } else if (l.type == SD_YOLO_SHORTCUT) {
            const int f = yolo_resolve(i, l.from[0]);
            if (f < 0 || f >= i || y->R[f].H != H || y->R[f].W != W || y->R[f].C != C) { delete y; return set_err(SD_ERR_INVALID, "bad shortcut"); }
            r.H = H; r.W = W; r.C = C; r.outC = C;
        } else if (l.type == SD_YOLO_ROUTE) {
            const int f0 = yolo_resolve(i, l.from[0]);
            if (f0 < 0 || f0 >= i) { delete y; return set_err(SD_ERR_INVALID, "bad route"); }
            r.H = y->R[f0].H; r.W = y->R[f0].W; r.C = y->R[f0].C;
            if (l.nfrom == 2) {
                const int f1 = yolo_resolve(i, l.from[1]);
                if (f1 < 0 || f1 >= i || y->R[f1].H != r.H || y->R[f1].W != r.W) { delete y; return set_err(SD_ERR_INVALID, "bad route"); }
                r.C += y->R[f1].C;
            }
            r.outC = r.C;
        } else if (l.type == SD_YOLO_UPSAMPLE) {
            r.H = 2 * H; r.W = 2 * W; r.C = C; r.outC = C;
        } else if (l.type == SD_YOLO_YOLO) {
            if (C != 3 * (5 + classes)) { delete y; return set_err(SD_ERR_INVALID, "[yolo] input must have 3*(5+classes) channels"); }
            r.H = H; r.W = W; r.C = C; r.outC = C;
            y->totalRows += H * W * 3;
        } else { delete y; return set_err(SD_ERR_INVALID, "unknown layer type"); }
        H = r.H; W = r.W; C = r.C;
        if ((l.type == SD_YOLO_CONV) && (r.C % 4) && r.C != 3 * (5 + classes)) { delete y; return set_err(SD_ERR_UNSUPPORTED, "filters must be a multiple of 4"); }
    }
    y->wTotal = wOff; y->bTotal = bOff; y->wTotalW = wOffW; y->wTotalB = wOffB;
    y->detCap = 8192;
    // ---- device memory
    auto alloc = [&](void** p, size_t bytes) -> bool {
        if (hipMalloc(p, bytes) != hipSuccess) return false;
        y->owned.push_back(*p);
        return true;
    };
    bool ok = true;
    const size_t nB = (size_t)max_batch;
    if (y->f32) ok = ok && alloc((void**)&y->d_blob8, nB * net_h * net_w * 8 * 4);
    else ok = ok && alloc((void**)&y->d_blob4, nB * net_h * net_w * 4 * 2);
    ok = ok && alloc((void**)&y->d_zero, 256);
    if (ok) ok = hipMemset(y->d_zero, 0, 256) == hipSuccess;

    if (!y->f32) {
        ok = ok && alloc((void**)&y->d_wgt, wOff * 2 + 64);                             // f16 weights: the f16 mode only
    } else {
        ok = ok && alloc((void**)&y->d_wgt32, wOff * 4 + 64);
        if (y->wino && wOffW) { ok = ok && alloc((void**)&y->d_wgtW, wOffW * 4 + 64); ok = ok && alloc((void**)&y->d_V, nB * vMax * 4 + 64); }
        if (y->b3 && wOffB) ok = ok && alloc((void**)&y->d_wgtB, wOffB * 16 + 65536);  // slack: the kernels request weight fragments up to two steps past a tile's last
    }
    ok = ok && alloc((void**)&y->d_bias, bOff * 4 + 64);
    ok = ok && alloc((void**)&y->d_dets, nB * y->detCap * sizeof(SdDet));
    ok = ok && alloc((void**)&y->d_ndet, nB * 4);
    ok = ok && alloc((void**)&y->d_raw, (size_t)y->totalRows * (5 + classes) * 4 + 64);
    ok = ok && alloc((void**)&y->d_ct, 8 * 8192);
    ok = ok && alloc((void**)&y->d_rt, 8 * 8192);
    for (int i = 0; ok && i < n_layers; i++) {
        const sd_yolo_layer& l = y->L[i];
        sd_yolo::Rt& r = y->R[i];
        if (l.type == SD_YOLO_CONV) {
            ok = alloc((void**)&r.out, nB * r.H * r.W * r.outC * eb + 64);
            if (ok && r.outC != r.C) ok = hipMemset(r.out, 0, nB * r.H * r.W * r.outC * eb) == hipSuccess;
        } else if (l.type == SD_YOLO_SHORTCUT) {
            // fused into the preceding convolution's epilogue when that output has no other consumer
            bool fuse = i > 0 && y->L[i - 1].type == SD_YOLO_CONV && yolo_resolve(i, l.from[0]) != i - 1;
            for (int j = 0; fuse && j < n_layers; j++) {
                if (j == i) continue;
                const sd_yolo_layer& o = y->L[j];
                if (o.type == SD_YOLO_SHORTCUT || o.type == SD_YOLO_ROUTE)
                    for (int k = 0; k < o.nfrom; k++) if (yolo_resolve(j, o.from[k]) == i - 1) fuse = false;
            }
            if (fuse) { r.out = y->R[i - 1].out; r.alias = true; }
            else ok = alloc((void**)&r.out, nB * r.H * r.W * r.outC * eb + 64);
        } else if (l.type == SD_YOLO_ROUTE && l.nfrom == 1) {
            r.out = y->R[yolo_resolve(i, l.from[0])].out; r.alias = true; r.outC = y->R[yolo_resolve(i, l.from[0])].outC;
        } else if (l.type == SD_YOLO_ROUTE) {
            ok = alloc((void**)&r.out, nB * r.H * r.W * r.outC * eb + 64);
        } else if (l.type == SD_YOLO_UPSAMPLE) {
            // materialised only inside the following 2-input route (k_upsample_concat); stand-alone upsample unsupported
            if (!(i + 1 < n_layers && y->L[i + 1].type == SD_YOLO_ROUTE && y->L[i + 1].nfrom == 2 && yolo_resolve(i + 1, y->L[i + 1].from[0]) == i)) {
                yolo_free(y); return set_err(SD_ERR_UNSUPPORTED, "[upsample] must feed a 2-input [route] as its first input");
            }
        } else if (l.type == SD_YOLO_YOLO) {
            r.out = y->R[i - 1].out; r.alias = true; r.outC = y->R[i - 1].outC;
        }
    }
    // f32-class modes: the second input of a 2-input [route] (yolov3.cfg: layers 61 and 36, the skip connections of the two up-sampling branches) is
    // written IN PLACE -- its producer's output pointer becomes the route buffer at the channel offset, its stride the route's channel count -- so the
    // route copies only the up-sampled half (k_upsample_into_f32).  Possible when the producer is a convolution (with or without a fused shortcut)
    // and its tensor is dense; every consumer reads it through (pointer, channel stride) anyway.
    for (int i = 0; ok && y->f32 && i < n_layers; i++) {
        const sd_yolo_layer& l = y->L[i];
        if (l.type != SD_YOLO_ROUTE || l.nfrom != 2) continue;
        const int fa = yolo_resolve(i, l.from[0]), fb = yolo_resolve(i, l.from[1]);
        const int src = yolo_resolve(fa, -1);
        sd_yolo::Rt& rb = y->R[fb];
        const int conv = y->L[fb].type == SD_YOLO_CONV ? fb : (y->L[fb].type == SD_YOLO_SHORTCUT && rb.alias ? fb - 1 : -1);
        if (conv < 0 || y->L[fa].type != SD_YOLO_UPSAMPLE || rb.outC != rb.C || y->R[conv].outC != y->R[conv].C || (y->R[src].C % 4) || (rb.C % 4)) continue;
        bool soleRoute = true;                             // one route only may own the tensor's storage
        for (int j = 0; j < n_layers; j++)
            if (j != i && y->L[j].type == SD_YOLO_ROUTE)
                for (int k = 0; k < y->L[j].nfrom; k++) if (yolo_resolve(j, y->L[j].from[k]) == fb) soleRoute = false;
        if (!soleRoute) continue;
        float* at = (float*)y->R[i].out + y->R[src].C;
        y->R[conv].out = (_Float16*)at; y->R[conv].outC = y->R[i].C;
        rb.out = (_Float16*)at; rb.outC = y->R[i].C;
        y->R[i].alias = true;                              // marks the route: its second input is already in place
    }
    if (ok) ok = hipStreamCreateWithFlags(&y->stream, hipStreamNonBlocking) == hipSuccess;
    if (!ok) { yolo_free(y); return set_err(SD_ERR_HIP, "detector allocation failed"); }
    *out = y;
    return SD_OK;
}

// Overlap mode (f32-class modes): blobFromImage runs on an internal stream ahead of the pass's first convolution, the three region decodes on
// another one behind their heads' convolutions, ordered by events; sd_yolo_boxes_device (on ITS stream argument) waits for the decodes.  With
// the next pass already enqueued on the convolution stream, its convolutions start while this pass's decode / NMS / download run beside them.
// The stream given to sd_yolo_forward_device then is NOT a completion point for the decoded rows: consume them through sd_yolo_boxes_device /
// sd_yolo_boxes_batch (any stream) or the host forms (they synchronise the device).
int sd_yolo_set_overlap(sd_yolo* y, int on)
{
    if (!y) return SD_ERR_INVALID;
    if (on && !y->f32) return set_err(SD_ERR_UNSUPPORTED, "overlap mode exists for the f32-class detector modes");
    int heads = 0;
    for (const sd_yolo_layer& l : y->L) heads += l.type == SD_YOLO_YOLO;
    if (on && heads > 3) return set_err(SD_ERR_UNSUPPORTED, "overlap mode orders at most three [yolo] heads");
    if (on && !y->sPre) {
        HIPCHK(hipStreamCreateWithFlags(&y->sPre, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&y->sPost, hipStreamNonBlocking));
        hipEvent_t* evs[] = {&y->evBlob, &y->evL0, &y->evHead[0], &y->evHead[1], &y->evHead[2], &y->evDecoded, &y->evNms};
        for (hipEvent_t* e : evs) HIPCHK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    if (!on && y->overlap) HIPCHK(hipDeviceSynchronize());
    y->overlap = on != 0;
    return SD_OK;
}

int sd_yolo_destroy(sd_yolo* y) { if (y) { (void)hipDeviceSynchronize(); yolo_free(y); } return SD_OK; }

int sd_yolo_weight_count(const sd_yolo* y, size_t* n_floats)
{
    if (!y || !n_floats) return SD_ERR_INVALID;
    size_t n = 0;
    int C = 3;
    for (size_t i = 0; i < y->L.size(); i++) {
        const sd_yolo_layer& l = y->L[i];
        if (l.type == SD_YOLO_CONV) {
            const int cin = i == 0 ? 3 : (int)y->R[i].cinPad;
            n += (size_t)l.filters * (l.batch_normalize ? 4 : 1) + (size_t)l.filters * cin * l.size * l.size;
        }
        (void)C;
    }
    *n_floats = n;
    return SD_OK;
}

int sd_yolo_load_darknet_weights(sd_yolo* y, const float* p, size_t n_floats)
{
    if (!y || !p) return SD_ERR_INVALID;
    size_t need = 0;
    sd_yolo_weight_count(y, &need);
    if (n_floats != need) return set_err(SD_ERR_INVALID, "weight payload has " + std::to_string(n_floats) + " floats, the network needs " + std::to_string(need));
    if (y->f32) {
        // f32 mode: [coutPad][taps][cinPad] f32 (the first layer's 3 input channels sit in a K chunk of 8), batch-norm folded in f32
        std::vector<float> w32(y->wTotal, 0.f);
        std::vector<float> b32(y->bTotal, 0.f);
        const float* q = p;
        for (size_t i = 0; i < y->L.size(); i++) {
            const sd_yolo_layer& l = y->L[i];
            if (l.type != SD_YOLO_CONV) continue;
            const sd_yolo::Rt& r = y->R[i];
            const int cin = i == 0 ? 3 : r.cinPad, cinP = i == 0 ? 8 : r.cinPad, F = l.filters, taps = l.size * l.size;
            const float* biases = q; q += F;
            const float *scales = nullptr, *mean = nullptr, *var = nullptr;
            if (l.batch_normalize) { scales = q; q += F; mean = q; q += F; var = q; q += F; }
            const float* wt = q; q += (size_t)F * cin * taps;
            for (int f = 0; f < F; f++) {
                float sc = 1.f, bias = biases[f];
                if (l.batch_normalize) { sc = scales[f] / sqrtf(var[f] + 0.000001f); bias = biases[f] - mean[f] * sc; }
                b32[r.bOff + f] = bias;
                for (int c = 0; c < cin; c++)
                    for (int t = 0; t < taps; t++) {
                        // first layer: K step s = taps 2 s and 2 s + 1, four channels each (k_conv_f32's `pair` mode); else [tap][cinPad]
                        const size_t k = i == 0 ? (size_t)f * 8 * ((taps + 1) / 2) + (size_t)(t / 2) * 8 + (size_t)(t % 2) * 4 + c
                                                : ((size_t)f * taps + t) * cinP + c;
                        w32[r.wOff + k] = wt[((size_t)f * cin + c) * taps + t] * sc;
                    }
            }
        }
        HIPCHK(hipMemcpy(y->d_wgt32, w32.data(), y->wTotal * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(y->d_bias, b32.data(), y->bTotal * 4, hipMemcpyHostToDevice));
        if (y->wTotalW) {
            // SD_YOLO_F32W: U = G g G^T of the folded weights, [coutPad][16][cin]; G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
            std::vector<float> wW(y->wTotalW, 0.f);
            for (size_t i = 0; i < y->L.size(); i++) {
                const sd_yolo::Rt& r = y->R[i];
                if (y->L[i].type != SD_YOLO_CONV || !r.wino) continue;
                const int cin = r.cinPad, F = y->L[i].filters;
                for (int f = 0; f < F; f++)
                    for (int c = 0; c < cin; c++) {
                        float g[3][3], t[4][3];
                        for (int k = 0; k < 9; k++) g[k / 3][k % 3] = w32[r.wOff + ((size_t)f * 9 + k) * cin + c];
                        for (int j = 0; j < 3; j++) {
                            t[0][j] = g[0][j];
                            t[1][j] = 0.5f * ((g[0][j] + g[1][j]) + g[2][j]);
                            t[2][j] = 0.5f * ((g[0][j] - g[1][j]) + g[2][j]);
                            t[3][j] = g[2][j];
                        }
                        for (int a = 0; a < 4; a++) {
                            const float u[4] = {t[a][0], 0.5f * ((t[a][0] + t[a][1]) + t[a][2]), 0.5f * ((t[a][0] - t[a][1]) + t[a][2]), t[a][2]};
                            for (int b = 0; b < 4; b++) wW[r.wOffW + ((size_t)f * 16 + 4 * a + b) * cin + c] = u[b];
                        }
                    }
            }
            HIPCHK(hipMemcpy(y->d_wgtW, wW.data(), y->wTotalW * 4, hipMemcpyHostToDevice));
        }
        if (y->wTotalB) {
            // SD_YOLO_F32X3: every folded weight as three bf16 limbs (round to nearest even, each limb of what the ones before left: the sum
            // of the three is the f32 weight exactly), [coutPad][taps][cin / 4][limb][4]
            auto bf16_rne = [](float v) -> uint16_t {
                uint32_t u; memcpy(&u, &v, 4);
                if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)(u >> 16);
                return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
            };
            auto bf16_val = [](uint16_t b) -> float { uint32_t u = (uint32_t)b << 16; float v; memcpy(&v, &u, 4); return v; };
            // fragment order: [filter tile of 128][K step][wave row (64 filters)][limb][m (32 filters)][lane = filter % 32 + 32 * (k / 8)][k % 8]
            std::vector<uint16_t> wB(y->wTotalB * 8, 0);
            for (size_t i = 0; i < y->L.size(); i++) {
                const sd_yolo::Rt& r = y->R[i];
                if (y->L[i].type != SD_YOLO_CONV || !r.b3) continue;
                const int cin = r.cinPad, taps = y->L[i].size * y->L[i].size, F = y->L[i].filters;
                const size_t ksteps = (size_t)taps * (cin / 16);
                for (int f = 0; f < F; f++)
                    for (int t = 0; t < taps; t++)
                        for (int c = 0; c < cin; c++) {
                            const float v = w32[r.wOff + ((size_t)f * taps + t) * cin + c];
                            const uint16_t hi = bf16_rne(v); const float r1 = v - bf16_val(hi);
                            const uint16_t mid = bf16_rne(r1); const float r2 = r1 - bf16_val(mid);
                            const uint16_t lo = bf16_rne(r2);
                            const size_t ks = r.b3flat ? (size_t)(c / 16) * 9 + t : (size_t)t * (cin / 16) + c / 16;      // k_conv3x3_b3 walks [chunk][tap]
                            const int k = c % 16, lane = f % 32 + 32 * (k / 8), bm = 64 * r.b3wm, wmr = (f % bm) / 64, m = (f % 64) / 32;
                            const size_t frag0 = (((size_t)(f / bm) * ksteps + ks) * r.b3wm + wmr) * 6;
                            const uint16_t limb[3] = {hi, mid, lo};
                            for (int l = 0; l < 3; l++) wB[((r.wOffB + (frag0 + 2 * l + m) * 64 + lane) * 8) + k % 8] = limb[l];
                        }
            }
            HIPCHK(hipMemcpy(y->d_wgtB, wB.data(), y->wTotalB * 16, hipMemcpyHostToDevice));
        }
        y->weightsLoaded = true;
        return SD_OK;
    }
    std::vector<_Float16> w(y->wTotal, (_Float16)0.f);
    std::vector<float> b(y->bTotal, 0.f);
    for (size_t i = 0; i < y->L.size(); i++) {
        const sd_yolo_layer& l = y->L[i];
        if (l.type != SD_YOLO_CONV) continue;
        const sd_yolo::Rt& r = y->R[i];
        const int cin = i == 0 ? 3 : r.cinPad, F = l.filters, taps = l.size * l.size;
        const float* biases = p; p += F;
        const float *scales = nullptr, *mean = nullptr, *var = nullptr;
        if (l.batch_normalize) { scales = p; p += F; mean = p; p += F; var = p; p += F; }
        const float* wt = p; p += (size_t)F * cin * taps;
        for (int f = 0; f < F; f++) {
            // batch-norm folding as cv::dnn's Darknet importer applies it: y = (x - mean) * scale / sqrt(var + 1e-6) + beta
            float s = 1.f, bias = biases[f];
            if (l.batch_normalize) { s = scales[f] / sqrtf(var[f] + 0.000001f); bias = biases[f] - mean[f] * s; }
            b[r.bOff + f] = bias;
            for (int c = 0; c < cin; c++)
                for (int t = 0; t < taps; t++)
                {
                    if (i == 0)                        // k_conv_first: one 48-wide K row per filter, k = tap*4 + c
                        w[r.wOff + (size_t)f * 48 + (size_t)t * 4 + c] = (_Float16)(wt[((size_t)f * cin + c) * taps + t] * s);
                    else
                        w[r.wOff + ((size_t)f * taps + t) * r.cinPad + c] = (_Float16)(wt[((size_t)f * cin + c) * taps + t] * s);
                }
        }
    }
    HIPCHK(hipMemcpy(y->d_wgt, w.data(), y->wTotal * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(y->d_bias, b.data(), y->bTotal * 4, hipMemcpyHostToDevice));
    y->weightsLoaded = true;
    return SD_OK;
}

int sd_yolo_layer_shape(const sd_yolo* y, int layer, int* h, int* w, int* c)
{
    if (!y || layer < 0 || layer >= (int)y->L.size()) return SD_ERR_INVALID;
    if (h) *h = y->R[layer].H;
    if (w) *w = y->R[layer].W;
    if (c) *c = y->R[layer].C;
    return SD_OK;
}

int sd_yolo_flops(const sd_yolo* y, double* flops_per_image)
{
    if (!y || !flops_per_image) return SD_ERR_INVALID;
    *flops_per_image = y->convFlops;
    return SD_OK;
}

int sd_yolo_mfma_flops_bf16(const sd_yolo* y, double* flops_per_image)
{
    if (!y || !flops_per_image) return SD_ERR_INVALID;
    *flops_per_image = y->mfmaFlopsBf16;
    return SD_OK;
}

int sd_yolo_winograd_layers(const sd_yolo* y, int* n_layers)
{
    if (!y || !n_layers) return SD_ERR_INVALID;
    int n = 0;
    for (const sd_yolo::Rt& r : y->R) n += r.wino ? 1 : 0;
    *n_layers = n;
    return SD_OK;
}

int sd_yolo_mfma_flops(const sd_yolo* y, double* flops_per_image)
{
    if (!y || !flops_per_image) return SD_ERR_INVALID;
    *flops_per_image = y->mfmaFlops;
    return SD_OK;
}

// The forward pass in f32 (k_yolo32.h): same graph walk, one generic convolution kernel, f32 activations.
// filter tiles walked back to back on a pixel tile (k_conv_f32's workgroup order): the largest power of two that divides tilesY and
// keeps the group's weights (bm filters x kdim floats per tile) within 2.5 MB of an XCD's 4 MB L2
static int f32_group_y(int tilesY, int bm, int kdim)
{
    int g = 1;
    while (2 * g <= tilesY && tilesY % (2 * g) == 0 && (size_t)(2 * g) * bm * kdim * 4 <= (size_t)2560 * 1024) g *= 2;
    return g;
}

static int yolo_forward_f32(sd_yolo* y, const uint8_t* d_bgr, int width, int height, size_t stride, size_t image_pitch, int n,
                            float conf_threshold, hipStream_t s)
{
    const bool ov = y->overlap;
    hipStream_t sb = ov ? y->sPre : s, sd = ov ? y->sPost : s;          // blobFromImage / region decodes
    if (ov && y->haveL0) HIPCHK(hipStreamWaitEvent(sb, y->evL0, 0));    // the previous pass's first convolution has read the blob
    {
        dim3 blk(64, 4), grd((y->netW + 63) / 64, (y->netH + 3) / 4, n);
        hipLaunchKernelGGL(k_blob_from_image_f32, grd, blk, 0, sb, d_bgr, width, height, stride, image_pitch, y->d_ct, y->d_rt, y->d_blob8, y->netW, y->netH, 1);
    }
    LAUNCH_CHECK("k_blob_from_image_f32");
    if (ov) {
        HIPCHK(hipEventRecord(y->evBlob, sb));
        HIPCHK(hipStreamWaitEvent(s, y->evBlob, 0));
    }
    int head = 0;
    bool headGuard = ov && y->haveDecoded;              // before the first head tensor is overwritten: the previous pass's decodes have read them
    if (!y->attrF32) {
        HIPCHK(hipFuncSetAttribute((const void*)k_conv_f32<32, 2, 2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, SD_F32_LDS(32, 2, 2, 8)));
        HIPCHK(hipFuncSetAttribute((const void*)k_conv_f32<32, 2, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, SD_F32_LDS(32, 2, 2, 4)));
        HIPCHK(hipFuncSetAttribute((const void*)k_conv_f32<16, 1, 2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, SD_F32_LDS(16, 1, 2, 8)));
        HIPCHK(hipFuncSetAttribute((const void*)k_conv_f32<16, 1, 1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, SD_F32_LDS(16, 1, 1, 8)));
        HIPCHK(hipFuncSetAttribute((const void*)k_conv_f32<8, 1, 1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, SD_F32_LDS(8, 1, 1, 8)));
        HIPCHK(hipFuncSetAttribute((const void*)k_conv_f32<16, 1, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, SD_F32_LDS(16, 1, 2, 4)));
        HIPCHK(hipFuncSetAttribute((const void*)k_conv_f32<16, 1, 1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, SD_F32_LDS(16, 1, 1, 4)));
        HIPCHK(hipFuncSetAttribute((const void*)k_conv_f32<16, 2, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, SD_F32_LDS(16, 2, 2, 4)));
        HIPCHK(hipFuncSetAttribute((const void*)k_wino_gemm_f32<16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, SD_WINO_LDS(16, 2)));
        HIPCHK(hipFuncSetAttribute((const void*)k_conv_b3<1>, hipFuncAttributeMaxDynamicSharedMemorySize, SD_B3_LDS(1)));
        HIPCHK(hipFuncSetAttribute((const void*)k_conv_b3<2>, hipFuncAttributeMaxDynamicSharedMemorySize, SD_B3_LDS(2)));
        HIPCHK(hipFuncSetAttribute((const void*)k_conv3x3_b3<3, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, SD_B3F_LDS(160, 128)));
        HIPCHK(hipFuncSetAttribute((const void*)k_conv3x3_b3<4, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, SD_B3F_LDS(160, 128)));
        HIPCHK(hipFuncSetAttribute((const void*)k_conv3x3_b3<5, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, SD_B3F_LDS(160, 128)));
        HIPCHK(hipFuncSetAttribute((const void*)k_conv3x3_b3<8, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, SD_B3F_LDS(160, 128)));
HIPCHK(hipFuncSetAttribute((const void*)k_conv3x3_b3c<5>, hipFuncAttributeMaxDynamicSharedMemorySize, SD_B3C_LDS(80)));
        HIPCHK(hipFuncSetAttribute((const void*)k_conv3x3_b3c<6>, hipFuncAttributeMaxDynamicSharedMemorySize, SD_B3C_LDS(80)));
        y->attrF32 = true;
    }
    // tile variant of the >= 128-filter layers: 3 (default) = 128 x 128 tiles on 4-wave workgroups with 16-channel K steps, 40 KB of LDS and 144
    // VGPRs: THREE independent workgroups per CU (3 waves per SIMD keep the MFMA pipe fed through each other's barriers and staging;
    // +3.3 % over variant 1 on every such layer at batch 128; four per CU -- 8-channel steps, 128 VGPRs without fragment prefetch, or 64 x 128 tiles --
    // were 5-7 % slower); 1 = the same tile with 32-channel steps, two workgroups per CU; 0 = 8-wave 128 x 256 tiles; 2 = variant 1 for the
    // 1x1 layers only.  SD_F32_VARIANT is a developer switch.
    static const int variant = getenv("SD_F32_VARIANT") ? atoi(getenv("SD_F32_VARIANT")) : 3;
    static const int small4 = getenv("SD_F32_SMALL4") ? atoi(getenv("SD_F32_SMALL4")) : 1;       // 4-wave tiles for the <= 64-filter layers too (0.5 % at batch 128); developer switch
    const float* cur = y->d_blob8;
    int H = y->netH, W = y->netW, Cs = 4;
    int rowBase = 0;
    for (size_t i = 0; i < y->L.size(); i++) {
        const sd_yolo_layer& l = y->L[i];
        const sd_yolo::Rt& r = y->R[i];
        if (l.type == SD_YOLO_CONV && r.wino) {
            // Winograd F(2x2, 3x3), k_yolo32w.h: input transform into the scratch V, then one GEMM over K = 16 cin with the output transform folded in
            SdWinoArgs A;
            A.V = y->d_V; A.U = y->d_wgtW + r.wOffW; A.bias = y->d_bias + r.bOff; A.res = nullptr; A.out = (float*)r.out; A.zero = (const float*)y->d_zero;
            A.N = n; A.H = r.H; A.W = r.W; A.th = (r.H + 1) / 2; A.tw = (r.W + 1) / 2;
            A.cin = r.cinPad; A.cout = l.filters; A.outStride = r.outC; A.resStride = 0; A.leaky = l.leaky;
            if (i + 1 < y->L.size() && y->L[i + 1].type == SD_YOLO_SHORTCUT && y->R[i + 1].alias) {
                const int f = yolo_resolve((int)i + 1, y->L[i + 1].from[0]);
                A.res = (const float*)y->R[f].out; A.resStride = y->R[f].outC;
            }
            const size_t nblk = (size_t)n * A.th * A.tw, work = nblk * (r.cinPad / 4);
            hipLaunchKernelGGL(k_wino_input, dim3((unsigned)std::min<size_t>((work + 255) / 256, 65536)), dim3(256), 0, s, cur, n, H, W, r.cinPad, Cs, A.th, A.tw, y->d_V);
            LAUNCH_CHECK("k_wino_input");
            // 128 filters x 64 blocks per workgroup.  Measured on one box against the direct f32 mode's 126.4 ms per 128-image batch: this tile
            // 90.9 ms; 64 x 64 tiles (a wave owns 32 x 32, 126 VGPRs, three workgroups per CU) 93.8 ms with 16-channel steps, 92.4 ms with 32.
            // All filter tiles back to back on a block tile: V is the big operand here (4 x the layer's input; a block tile's 16 cin x 64 floats stay in L2
            // while the filter tiles pass, the weights come from the Infinity Cache) -- with k_conv_f32's rule (a group's weights <= 2.5 MB) the 512-channel
            // layers re-read V eight times: 89.1 / 88.1 ms per 128-image batch against 86.1 on the same box.
            A.tilesX = (int)((nblk + 63) / 64); A.tilesY = r.coutPad / 128; A.groupY = A.tilesY;
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wino_gemm_f32<16, 2>), dim3(SD_F32_GRID(A.tilesX, A.tilesY)), dim3(256), SD_WINO_LDS(16, 2), s, A);
            LAUNCH_CHECK("k_wino_gemm_f32");
        } else if (l.type == SD_YOLO_CONV) {
            if (headGuard && i + 1 < y->L.size() && y->L[i + 1].type == SD_YOLO_YOLO) { HIPCHK(hipStreamWaitEvent(s, y->evDecoded, 0)); headGuard = false; }
            SdConvArgsF A;
            A.in = cur; A.wgt = y->d_wgt32 + r.wOff; A.bias = y->d_bias + r.bOff; A.res = nullptr; A.out = (float*)r.out; A.zero = (const float*)y->d_zero;
            A.N = n; A.H = H; A.W = W; A.cin = i == 0 ? 8 : r.cinPad; A.cinStride = Cs; A.pair = i == 0 ? 1 : 0;
            A.Ho = r.H; A.Wo = r.W; A.cout = l.filters; A.outStride = r.outC; A.resStride = 0;
            A.ksize = l.size; A.stride = l.stride; A.pad = l.size / 2; A.leaky = l.leaky;
            if (i + 1 < y->L.size() && y->L[i + 1].type == SD_YOLO_SHORTCUT && y->R[i + 1].alias) {
                const int f = yolo_resolve((int)i + 1, y->L[i + 1].from[0]);
                A.res = (const float*)y->R[f].out; A.resStride = y->R[f].outC;
            }
            const int npix = n * r.H * r.W;
            if (r.b3) {
                A.tilesX = (npix + 127) / 128; A.tilesY = r.coutPad / 128; A.groupY = f32_group_y(A.tilesY, 128, A.cin * l.size * l.size * 3 / 2);
                if (r.b3flat && r.b3wm == 1) {
                    const uint4* wq = (const uint4*)(y->d_wgtB + r.wOffB);
                    A.tilesX = (npix + 511) / 512; A.tilesY = r.coutPad / 64; A.groupY = f32_group_y(A.tilesY, 64, A.cin * 9 * 3 / 2);
                    const int np = (4 * (512 + 2 * W + 2) + 511) / 512;
                    const dim3 grd(8 * (unsigned)std::min(((A.tilesX + 7) / 8) * A.tilesY, 32));      // persistent: one workgroup per CU, 32 per XCD
                    if (np <= 5) hipLaunchKernelGGL(k_conv3x3_b3c<5>, grd, dim3(512), SD_B3C_LDS(W), s, A, wq);
                    else hipLaunchKernelGGL(k_conv3x3_b3c<6>, grd, dim3(512), SD_B3C_LDS(W), s, A, wq);
                } else if (r.b3flat) {
                    const uint4* wq = (const uint4*)(y->d_wgtB + r.wOffB);
                    const int np = (4 * (128 + 2 * W + 2) + 255) / 256;
                    const size_t lds = SD_B3F_LDS(W, 128);
                    const dim3 grd(SD_F32_GRID(A.tilesX, A.tilesY));
                    if (np <= 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv3x3_b3<3, 2, 2>), grd, dim3(256), lds, s, A, wq);
                    else if (np == 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv3x3_b3<4, 2, 2>), grd, dim3(256), lds, s, A, wq);
                    else if (np == 5) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv3x3_b3<5, 2, 2>), grd, dim3(256), lds, s, A, wq);
                    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv3x3_b3<8, 2, 2>), grd, dim3(256), lds, s, A, wq);
                } else if (r.b3wm == 1) {              // the 64-filter layers
                    A.tilesX = (npix + 255) / 256; A.tilesY = 1; A.groupY = 1;
                    hipLaunchKernelGGL(k_conv_b3<1>, dim3(SD_F32_GRID(A.tilesX, A.tilesY)), dim3(256), SD_B3_LDS(1), s, A, (const uint4*)(y->d_wgtB + r.wOffB));
                } else
                hipLaunchKernelGGL(k_conv_b3<2>, dim3(SD_F32_GRID(A.tilesX, A.tilesY)), dim3(256), SD_B3_LDS(2), s, A, (const uint4*)(y->d_wgtB + r.wOffB));
            } else if (i == 0)                         // 3 (-> 8) input channels, <= 32 filters (round 4: the same tile on 4 waves, two workgroups per CU: 118.4 vs 118.3 ms per 128-image pass, no change)
                { A.tilesX = (npix + 511) / 512; A.tilesY = (l.filters + 31) / 32; A.groupY = f32_group_y(A.tilesY, 32, A.cin * l.size * l.size); hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_f32<8, 1, 1, 8>), dim3(SD_F32_GRID(A.tilesX, A.tilesY)), dim3(512), SD_F32_LDS(8, 1, 1, 8), s, A); }
            else if (l.filters <= 32 && small4)
                { A.tilesX = (npix + 255) / 256; A.tilesY = 1; A.groupY = f32_group_y(A.tilesY, 32, A.cin * l.size * l.size); hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_f32<16, 1, 1, 4>), dim3(SD_F32_GRID(A.tilesX, A.tilesY)), dim3(256), SD_F32_LDS(16, 1, 1, 4), s, A); }
            else if (l.filters <= 32)
                { A.tilesX = (npix + 511) / 512; A.tilesY = 1; A.groupY = f32_group_y(A.tilesY, 32, A.cin * l.size * l.size); hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_f32<16, 1, 1, 8>), dim3(SD_F32_GRID(A.tilesX, A.tilesY)), dim3(512), SD_F32_LDS(16, 1, 1, 8), s, A); }
            else if (l.filters <= 64 && small4)
                { A.tilesX = (npix + 255) / 256; A.tilesY = 1; A.groupY = f32_group_y(A.tilesY, 64, A.cin * l.size * l.size); hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_f32<16, 1, 2, 4>), dim3(SD_F32_GRID(A.tilesX, A.tilesY)), dim3(256), SD_F32_LDS(16, 1, 2, 4), s, A); }
            else if (l.filters <= 64)
                { A.tilesX = (npix + 511) / 512; A.tilesY = 1; A.groupY = f32_group_y(A.tilesY, 64, A.cin * l.size * l.size); hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_f32<16, 1, 2, 8>), dim3(SD_F32_GRID(A.tilesX, A.tilesY)), dim3(512), SD_F32_LDS(16, 1, 2, 8), s, A); }
            else if (variant == 3)
                { A.tilesX = (npix + 127) / 128; A.tilesY = r.coutPad / 128; A.groupY = f32_group_y(A.tilesY, 128, A.cin * l.size * l.size); hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_f32<16, 2, 2, 4>), dim3(SD_F32_GRID(A.tilesX, A.tilesY)), dim3(256), SD_F32_LDS(16, 2, 2, 4), s, A); }
            else if (variant == 1 || (variant == 2 && l.size == 1))
                { A.tilesX = (npix + 127) / 128; A.tilesY = r.coutPad / 128; A.groupY = f32_group_y(A.tilesY, 128, A.cin * l.size * l.size); hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_f32<32, 2, 2, 4>), dim3(SD_F32_GRID(A.tilesX, A.tilesY)), dim3(256), SD_F32_LDS(32, 2, 2, 4), s, A); }
            else
                { A.tilesX = (npix + 255) / 256; A.tilesY = r.coutPad / 128; A.groupY = f32_group_y(A.tilesY, 128, A.cin * l.size * l.size); hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_f32<32, 2, 2, 8>), dim3(SD_F32_GRID(A.tilesX, A.tilesY)), dim3(512), SD_F32_LDS(32, 2, 2, 8), s, A); }
            LAUNCH_CHECK("k_conv_f32");
        } else if (l.type == SD_YOLO_SHORTCUT) {
            if (!r.alias) return set_err(SD_ERR_UNSUPPORTED, "unfused [shortcut] is not implemented");
        } else if (l.type == SD_YOLO_ROUTE && l.nfrom == 2) {
            const int fa = yolo_resolve((int)i, l.from[0]), fb = yolo_resolve((int)i, l.from[1]);
            const int src = yolo_resolve(fa, -1);
            const sd_yolo::Rt& ra = y->R[src]; const sd_yolo::Rt& rb = y->R[fb];
            if (r.alias) {                             // the skip tensor was written in place by its producer: only the up-sampled half moves
                const size_t quads = (size_t)n * r.H * r.W * (ra.C / 4);
                hipLaunchKernelGGL(k_upsample_into_f32, dim3((unsigned)std::min<size_t>((quads + 255) / 256, 4096)), dim3(256), 0, s, (const float*)ra.out, ra.C, ra.outC, ra.H, ra.W,
                                   (float*)r.out, r.C, n);
                LAUNCH_CHECK("k_upsample_into_f32");
            } else {
                if (ra.outC != ra.C || rb.outC != rb.C || (ra.C % 4) || (rb.C % 4)) return set_err(SD_ERR_UNSUPPORTED, "route inputs must be dense, channels % 4 == 0");
                // the copy kernel moves 16-byte pieces: an f32 channel counts as two halfs
                hipLaunchKernelGGL(k_upsample_concat, dim3(2048), dim3(256), 0, s, ra.out, 2 * ra.C, ra.H, ra.W, rb.out, 2 * rb.C, r.out, n);
                LAUNCH_CHECK("k_upsample_concat");
            }
        } else if (l.type == SD_YOLO_YOLO) {
            const float* an = y->anchors;
            const int rows = n * r.H * r.W * 3;
            if (ov && head < 3) {
                HIPCHK(hipEventRecord(y->evHead[head], s));
                HIPCHK(hipStreamWaitEvent(sd, y->evHead[head], 0));
                if (head == 0 && y->haveNms) HIPCHK(hipStreamWaitEvent(sd, y->evNms, 0));     // the previous pass's NMS has read the row lists
            }
            if (head == 0) HIPCHK(hipMemsetAsync(y->d_ndet, 0, (size_t)n * 4, sd));
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_region_decode<float>), dim3((rows + 255) / 256), dim3(256), 0, sd, (const float*)r.out, r.outC, r.H, r.W, n, an[2 * l.mask[0]],
                               an[2 * l.mask[0] + 1], an[2 * l.mask[1]], an[2 * l.mask[1] + 1], an[2 * l.mask[2]], an[2 * l.mask[2] + 1],
                               y->netW, y->netH, conf_threshold, rowBase, y->d_dets, y->d_ndet, y->detCap, n == 1 ? y->d_raw : nullptr);
            LAUNCH_CHECK("k_region_decode");
            rowBase += r.H * r.W * 3;
            head++;
        }
        if (ov && i == 0) { HIPCHK(hipEventRecord(y->evL0, s)); y->haveL0 = true; }
        if (l.type != SD_YOLO_YOLO && l.type != SD_YOLO_UPSAMPLE) { cur = (const float*)r.out; H = r.H; W = r.W; Cs = r.outC; }
        if (l.type == SD_YOLO_YOLO) { cur = (const float*)r.out; }
    }
    if (head == 0) HIPCHK(hipMemsetAsync(y->d_ndet, 0, (size_t)n * 4, sd));          // a network without a [yolo] layer yields no rows
    if (ov) { HIPCHK(hipEventRecord(y->evDecoded, sd)); y->haveDecoded = true; }
    return SD_OK;
}

int sd_yolo_forward_device(sd_yolo* y, const uint8_t* d_bgr, int width, int height, size_t stride, size_t image_pitch, int n,
                           float conf_threshold, void* stream_)
{
    if (!y || !d_bgr || width < 2 || height < 2 || n < 1 || n > y->maxBatch || width > 8192 || height > 8192) return set_err(SD_ERR_INVALID, "bad forward arguments");
    if (!y->weightsLoaded) return set_err(SD_ERR_STATE, "detector weights not loaded");
    hipStream_t s = stream_ ? (hipStream_t)stream_ : y->stream;
    if (y->tabW != width || y->tabH != height) {
        std::vector<int16_t> ct, rt;
        yolo_resize_tables(width, height, y->netW, y->netH, ct, rt);
        HIPCHK(hipMemcpy(y->d_ct, ct.data(), ct.size() * 2, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(y->d_rt, rt.data(), rt.size() * 2, hipMemcpyHostToDevice));
        y->tabW = width; y->tabH = height;
    }
    if (y->f32) {
        int rc = yolo_forward_f32(y, d_bgr, width, height, stride, image_pitch, n, conf_threshold, s);
        if (rc != SD_OK) return rc;
        y->lastN = n;
        if (!stream_) { HIPCHK(hipStreamSynchronize(s)); if (y->overlap) HIPCHK(hipStreamSynchronize(y->sPost)); }
        return SD_OK;
    }
    {
        dim3 blk(64, 4), grd((y->netW + 63) / 64, (y->netH + 3) / 4, n);
        hipLaunchKernelGGL(k_blob_from_image, grd, blk, 0, s, d_bgr, width, height, stride, image_pitch, y->d_ct, y->d_rt, y->d_blob4,
                           y->netW, y->netH, 1);
    }
    LAUNCH_CHECK("k_blob_from_image");
    HIPCHK(hipMemsetAsync(y->d_ndet, 0, (size_t)n * 4, s));
    const _Float16* cur = y->d_blob4;
    int H = y->netH, W = y->netW, Cs = 32;
    int rowBase = 0;
    for (size_t i = 0; i < y->L.size(); i++) {
        const sd_yolo_layer& l = y->L[i];
        const sd_yolo::Rt& r = y->R[i];
        if (l.type == SD_YOLO_CONV && i == 0) {
            const size_t npix0 = (size_t)n * r.H * r.W;
            hipLaunchKernelGGL(k_conv_first, dim3((unsigned)((npix0 + 255) / 256)), dim3(256), 0, s, y->d_blob4, y->d_wgt + r.wOff,
                               y->d_bias + r.bOff, r.out, n, r.H, r.W, l.filters, r.outC, l.leaky);
            LAUNCH_CHECK("k_conv_first");
        } else if (l.type == SD_YOLO_CONV) {
            SdConvArgs A;
            A.zero = y->d_zero; A.in = cur; A.wgt = y->d_wgt + r.wOff; A.bias = y->d_bias + r.bOff; A.res = nullptr; A.out = r.out;
            A.N = n; A.H = H; A.W = W; A.cin = r.cinPad; A.cinStride = Cs;
            A.Ho = r.H; A.Wo = r.W; A.cout = l.filters; A.coutPad = r.coutPad; A.outStride = r.outC; A.outOff = 0; A.resStride = 0;
            A.ksize = l.size; A.stride = l.stride; A.pad = l.size / 2; A.leaky = l.leaky;
            if (i + 1 < y->L.size() && y->L[i + 1].type == SD_YOLO_SHORTCUT && y->R[i + 1].alias) {
                const int f = yolo_resolve((int)i + 1, y->L[i + 1].from[0]);
                A.res = y->R[f].out; A.resStride = y->R[f].outC;
            }
            const int npix = n * r.H * r.W;
            dim3 grd((npix + SD_CV_BN - 1) / SD_CV_BN, (l.filters + SD_CV_BM - 1) / SD_CV_BM);
            const bool flat3 = l.size == 3 && l.stride == 1 && W <= 160 && l.filters % SD_G3_BM == 0 && r.cinPad % 32 == 0 && npix >= SD_G3_BN;
            if (!flat3 && r.cinPad % 32 == 0 && npix >= 512 && (l.size == 1 || l.filters >= SD_G3_BM / 2)) {
                bool& attr = y->attrGlds;          // per detector (= per device): function attributes are per device
                const int lds8 = 3 * (512 * 64 + SD_G3_WBYTES), lds4 = 3 * (256 * 64 + SD_G3_WBYTES);
                if (!attr) {
                    HIPCHK(hipFuncSetAttribute((const void*)k_conv_glds<8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds8));
                    HIPCHK(hipFuncSetAttribute((const void*)k_conv_glds<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds4));
                    HIPCHK(hipFuncSetAttribute((const void*)k_conv_glds<8, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds8));
                    HIPCHK(hipFuncSetAttribute((const void*)k_conv_glds<4, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds4));
                    attr = true;
                }
                const int ct = r.coutPad / SD_G3_BM;
                const bool big = ((npix + 511) / 512) * ct >= 256;
                const dim3 g8((npix + 511) / 512, ct), g4((npix + 255) / 256, ct);
                if (l.size == 1 && big) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_glds<8, 1>), g8, dim3(512), lds8, s, A);
                else if (l.size == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_glds<4, 1>), g4, dim3(256), lds4, s, A);
                else if (big) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_glds<8, 3>), g8, dim3(512), lds8, s, A);
                else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_glds<4, 3>), g4, dim3(256), lds4, s, A);
            } else if (flat3) {
                bool& attr = y->attrFlat3;
                if (!attr) {
                    HIPCHK(hipFuncSetAttribute((const void*)k_conv3x3_glds<80>, hipFuncAttributeMaxDynamicSharedMemorySize, SD_G3_LDS(80)));
                    HIPCHK(hipFuncSetAttribute((const void*)k_conv3x3_glds<160>, hipFuncAttributeMaxDynamicSharedMemorySize, SD_G3_LDS(160)));
                    attr = true;
                }
                const dim3 g3((npix + SD_G3_BN - 1) / SD_G3_BN, l.filters / SD_G3_BM);
                if (W <= 80) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv3x3_glds<80>), g3, dim3(512), SD_G3_LDS(80), s, A);
                else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv3x3_glds<160>), g3, dim3(512), SD_G3_LDS(160), s, A);
            } else if (l.size == 3 && l.stride == 1 && W <= SD_C3_MAXW && l.filters % SD_C3_BM == 0 && r.cinPad % SD_C3_BK == 0)
                hipLaunchKernelGGL(k_conv3x3_flat, dim3((npix + SD_C3_BN - 1) / SD_C3_BN, l.filters / SD_C3_BM), dim3(256), 0, s, A);
            else if (r.cinPad % 64 == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_mfma<64>), grd, dim3(256), 0, s, A);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_mfma<32>), grd, dim3(256), 0, s, A);
            LAUNCH_CHECK("k_conv_mfma");
        } else if (l.type == SD_YOLO_SHORTCUT) {
            if (!r.alias) return set_err(SD_ERR_UNSUPPORTED, "unfused [shortcut] is not implemented");
        } else if (l.type == SD_YOLO_ROUTE && l.nfrom == 2) {
            const int fa = yolo_resolve((int)i, l.from[0]), fb = yolo_resolve((int)i, l.from[1]);
            const int src = yolo_resolve(fa, -1);          // the layer the [upsample] reads
            const sd_yolo::Rt& ra = y->R[src]; const sd_yolo::Rt& rb = y->R[fb];
            if (ra.outC != ra.C || rb.outC != rb.C || (ra.C % 8) || (rb.C % 8)) return set_err(SD_ERR_UNSUPPORTED, "route inputs must be dense, channels % 8 == 0");
            hipLaunchKernelGGL(k_upsample_concat, dim3(2048), dim3(256), 0, s, ra.out, ra.C, ra.H, ra.W, rb.out, rb.C, r.out, n);
            LAUNCH_CHECK("k_upsample_concat");
        } else if (l.type == SD_YOLO_YOLO) {
            const float* an = y->anchors;
            const int rows = n * r.H * r.W * 3;
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_region_decode<_Float16>), dim3((rows + 255) / 256), dim3(256), 0, s, (const _Float16*)r.out, r.outC, r.H, r.W, n, an[2 * l.mask[0]],
                               an[2 * l.mask[0] + 1], an[2 * l.mask[1]], an[2 * l.mask[1] + 1], an[2 * l.mask[2]], an[2 * l.mask[2] + 1],
                               y->netW, y->netH, conf_threshold, rowBase, y->d_dets, y->d_ndet, y->detCap, n == 1 ? y->d_raw : nullptr);
            LAUNCH_CHECK("k_region_decode");
            rowBase += r.H * r.W * 3;
        }
        // the input of the next layer
        if (l.type != SD_YOLO_YOLO && l.type != SD_YOLO_UPSAMPLE) { cur = r.out; H = r.H; W = r.W; Cs = r.outC; }
        if (l.type == SD_YOLO_YOLO) { cur = r.out; }
    }
    y->lastN = n;
    if (!stream_) HIPCHK(hipStreamSynchronize(s));
    return SD_OK;
}

int sd_yolo_download_layer(sd_yolo* y, int layer, int image, uint16_t* out)
{
    if (!y || !out || layer < 0 || layer >= (int)y->L.size() || image < 0 || image >= y->lastN) return SD_ERR_INVALID;
    const sd_yolo::Rt& r = y->R[layer];
    if (!r.out) return set_err(SD_ERR_INVALID, "layer has no materialised output");
    HIPCHK(hipDeviceSynchronize());
    const size_t pix = (size_t)r.H * r.W, eb = y->f32 ? 4 : 2;       // f32 mode: `out` receives floats
    const unsigned char* src = (const unsigned char*)r.out + (size_t)image * pix * r.outC * eb;
    if (r.outC == r.C) {
        HIPCHK(hipMemcpy(out, src, pix * r.C * eb, hipMemcpyDeviceToHost));
    } else {
        HIPCHK(hipMemcpy2D(out, (size_t)r.C * eb, src, (size_t)r.outC * eb, (size_t)r.C * eb, pix, hipMemcpyDeviceToHost));
    }
    return SD_OK;
}

int sd_yolo_precision(const sd_yolo* y, int* precision)
{
    if (!y || !precision) return SD_ERR_INVALID;
    *precision = y->b3 ? SD_YOLO_F32X3 : (y->wino ? SD_YOLO_F32W : (y->f32 ? SD_YOLO_F32 : SD_YOLO_F16));
    return SD_OK;
}

int sd_yolo_download_region(sd_yolo* y, float* rows, int* total_rows)
{
    if (!y || !rows || y->lastN != 1) return set_err(SD_ERR_STATE, "region rows are kept only after a forward with n == 1");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(rows, y->d_raw, (size_t)y->totalRows * (5 + y->classes) * 4, hipMemcpyDeviceToHost));
    if (total_rows) *total_rows = y->totalRows;
    return SD_OK;
}

// shared by Segmentation_ / Segmentation: rows above the threshold -> int boxes -> NMSBoxes -> kept (class-filtered) indices
static int yolo_nms(sd_yolo* y, int image, int frame_cols, int frame_rows, float conf_threshold, float nms_threshold,
                    std::vector<SdDet>& d, std::vector<YRect>& rects, std::vector<int>& kept)
{
    HIPCHK(hipDeviceSynchronize());
    int nd = 0;
    HIPCHK(hipMemcpy(&nd, y->d_ndet + image, 4, hipMemcpyDeviceToHost));
    if (nd > y->detCap) return set_err(SD_ERR_CAPACITY, "more than 8192 rows above the confidence threshold");
    d.resize(nd);
    if (nd) HIPCHK(hipMemcpy(d.data(), y->d_dets + (size_t)image * y->detCap, (size_t)nd * sizeof(SdDet), hipMemcpyDeviceToHost));
    std::sort(d.begin(), d.end(), [](const SdDet& a, const SdDet& b) { return a.row < b.row; });   // cv::dnn row order
    rects.assign(nd, YRect{0, 0, 0, 0});
    for (int i = 0; i < nd; i++) {
        if (!(d[i].conf > conf_threshold)) continue;
        const int centerX = (int)(d[i].cx * frame_cols), centerY = (int)(d[i].cy * frame_rows);
        const int width = (int)(d[i].w * frame_cols), height = (int)(d[i].h * frame_rows);
        rects[i] = YRect{centerX - width / 2, centerY - height / 2, width, height};
    }
    std::vector<int> order;
    for (int i = 0; i < nd; i++) if (d[i].conf > conf_threshold) order.push_back(i);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return d[a].conf > d[b].conf; });
    std::vector<int> keep;
    for (int idx : order) {
        bool k = true;
        for (size_t j = 0; j < keep.size() && k; j++) k = yolo_overlap(rects[idx], rects[keep[j]]) <= nms_threshold;
        if (k) keep.push_back(idx);
    }
    kept.clear();
    for (int idx : keep) {
        const int c = d[idx].cls;     // coco.names: 0 person, 1 bicycle, 2 car, 3 motorbike ("motorcycle" never matches), 5 bus, 7 truck
        if (c == 0 || c == 1 || c == 2 || c == 5 || c == 7) kept.push_back(idx);
    }
    return SD_OK;
}

// yolov3Segment::Segmentation (yolo.cc:34-58): mask = 1 outside the dilated central halves of the kept boxes; all ones
// (and *no_target = 1) when nothing is kept.  d_mask: frame_rows x frame_cols u8 in HBM.
int sd_yolo_mask_device(sd_yolo* y, int image, int frame_cols, int frame_rows, float conf_threshold, float nms_threshold,
                        uint8_t* d_mask, size_t stride, int* no_target, void* stream_)
{
    if (!y || !d_mask || image < 0 || image >= y->lastN || frame_cols < 1 || frame_rows < 1 || stride < (size_t)frame_cols) return SD_ERR_INVALID;
    std::vector<SdDet> d; std::vector<YRect> rects; std::vector<int> kept;
    int rc = yolo_nms(y, image, frame_cols, frame_rows, conf_threshold, nms_threshold, d, rects, kept);
    if (rc != SD_OK) return rc;
    if (kept.size() > SD_MAX_BOXES) return set_err(SD_ERR_CAPACITY, "more than SD_MAX_BOXES kept boxes");
    if (no_target) *no_target = kept.empty();
    SdMaskRects R;
    R.n = (int)kept.size();
    for (int k = 0; k < R.n; k++) {
        const YRect& b = rects[kept[k]];
        R.x0[k] = std::max(0, b.x + b.w / 4); R.x1[k] = std::min(b.x + 3 * b.w / 4, frame_cols);
        R.y0[k] = std::max(0, b.y); R.y1[k] = std::min(b.y + b.h, frame_rows);
    }
    hipStream_t s = stream_ ? (hipStream_t)stream_ : y->stream;
    hipLaunchKernelGGL(k_mask_dilate, dim3((frame_cols + 15) / 16, (frame_rows + 15) / 16), dim3(256), 0, s, R, frame_cols, frame_rows, d_mask, stride);
    LAUNCH_CHECK("k_mask_dilate");
    if (!stream_) HIPCHK(hipStreamSynchronize(s));
    return SD_OK;
}

int sd_yolo_boxes(sd_yolo* y, int image, int frame_cols, int frame_rows, float conf_threshold, float nms_threshold, double* boxes,
                  int32_t* class_ids, float* confidences, int cap, int* n_out)
{
    if (!y || !n_out || image < 0 || image >= y->lastN || frame_cols < 1 || frame_rows < 1) return SD_ERR_INVALID;
    {
        std::vector<SdDet> d; std::vector<YRect> rects; std::vector<int> kept;
        int rc = yolo_nms(y, image, frame_cols, frame_rows, conf_threshold, nms_threshold, d, rects, kept);
        if (rc != SD_OK) return rc;
        int n = 0;
        for (int idx : kept) {
            if (n >= cap) return set_err(SD_ERR_CAPACITY, "box buffer too small");
            const YRect& r = rects[idx];
            // rectCenterScale(box, Size2d(-0.2 w, 0.6 h)): rect += size; rect -= size / 2
            const double sw = -0.2 * (double)r.w, sh = 0.6 * (double)r.h;
            if (boxes) { boxes[4 * n] = (double)r.x - sw / 2.0; boxes[4 * n + 1] = (double)r.y - sh / 2.0; boxes[4 * n + 2] = (double)r.w + sw; boxes[4 * n + 3] = (double)r.h + sh; }
            if (class_ids) class_ids[n] = d[idx].cls;
            if (confidences) confidences[n] = d[idx].conf;
            n++;
        }
        *n_out = n;
        return SD_OK;
    }
}

// Host-image forms for a per-frame caller (yolo->Segmentation_(imLeft) in the example drivers): upload + forward for one image,
// and the Segmentation mask downloaded to host memory.
int sd_yolo_forward_host(sd_yolo* y, const uint8_t* bgr, int width, int height, size_t stride, float conf_threshold)
{
    if (!y || !bgr || width < 1 || height < 1 || stride < (size_t)width * 3) return set_err(SD_ERR_INVALID, "bad yolo_forward_host arguments");
    const size_t bytes = stride * (size_t)height;
    if (bytes > y->hostImgCap) {
        if (y->d_hostImg) (void)hipFree(y->d_hostImg);
        y->d_hostImg = nullptr; y->hostImgCap = 0;
        HIPCHK(hipMalloc((void**)&y->d_hostImg, bytes));
        y->hostImgCap = bytes;
    }
    HIPCHK(hipMemcpyAsync(y->d_hostImg, bgr, bytes, hipMemcpyHostToDevice, y->stream));
    return sd_yolo_forward_device(y, y->d_hostImg, width, height, stride, bytes, 1, conf_threshold, y->stream);
}

int sd_yolo_mask_host(sd_yolo* y, int frame_cols, int frame_rows, float conf_threshold, float nms_threshold, uint8_t* mask, size_t stride,
                      int* no_target)
{
    if (!y || !mask || frame_cols < 1 || frame_rows < 1 || stride < (size_t)frame_cols) return set_err(SD_ERR_INVALID, "bad yolo_mask_host arguments");
    const size_t bytes = (size_t)frame_cols * frame_rows;
    if (bytes > y->hostMaskCap) {
        if (y->d_hostMask) (void)hipFree(y->d_hostMask);
        y->d_hostMask = nullptr; y->hostMaskCap = 0;
        HIPCHK(hipMalloc((void**)&y->d_hostMask, bytes));
        y->hostMaskCap = bytes;
    }
    int rc = sd_yolo_mask_device(y, 0, frame_cols, frame_rows, conf_threshold, nms_threshold, y->d_hostMask, (size_t)frame_cols, no_target, nullptr);
    if (rc != SD_OK) return rc;
    HIPCHK(hipMemcpy2D(mask, stride, y->d_hostMask, (size_t)frame_cols, (size_t)frame_cols, (size_t)frame_rows, hipMemcpyDeviceToHost));
    return SD_OK;
}

// postprocess_ for the first n_images of the last forward pass, entirely on the device (k_yolo_nms): one launch, and
// with the host form one download of n_images x (SD_MAX_BOXES boxes + count) instead of a synchronisation per image.
int sd_yolo_boxes_device(sd_yolo* y, int n_images, int frame_cols, int frame_rows, float conf_threshold, float nms_threshold,
                         double* d_boxes, int32_t* d_class_ids, float* d_confidences, int32_t* d_n_boxes, void* stream_)
{
    if (!y || n_images < 0 || n_images > y->lastN || frame_cols < 1 || frame_rows < 1 || !d_boxes || !d_class_ids || !d_confidences || !d_n_boxes)
        return set_err(SD_ERR_INVALID, "bad yolo_boxes_device arguments");
    if (n_images == 0) return SD_OK;
    hipStream_t s = stream_ ? (hipStream_t)stream_ : y->stream;
    if (!y->attrNms) { HIPCHK(hipFuncSetAttribute((const void*)k_yolo_nms, hipFuncAttributeMaxDynamicSharedMemorySize, SD_NMS_LDS)); y->attrNms = true; }
    if (y->overlap && y->haveDecoded) HIPCHK(hipStreamWaitEvent(s, y->evDecoded, 0));      // the decodes of the last pass (internal stream)
    hipLaunchKernelGGL(k_yolo_nms, dim3(n_images), dim3(256), SD_NMS_LDS, s, y->d_dets, y->d_ndet, y->detCap, frame_cols, frame_rows, conf_threshold,
                       nms_threshold, d_boxes, d_class_ids, d_confidences, d_n_boxes);
    LAUNCH_CHECK("k_yolo_nms");
    if (y->overlap) { HIPCHK(hipEventRecord(y->evNms, s)); y->haveNms = true; }
    return SD_OK;
}

int sd_yolo_boxes_batch(sd_yolo* y, int n_images, int frame_cols, int frame_rows, float conf_threshold, float nms_threshold,
                        double* boxes, int32_t* class_ids, float* confidences, int32_t* n_boxes, void* stream_)
{
    if (!y || !boxes || !n_boxes || n_images < 0 || n_images > y->maxBatch) return set_err(SD_ERR_INVALID, "bad yolo_boxes_batch arguments");
    if (!y->d_nmsBoxes) {
        const size_t nB = (size_t)y->maxBatch;
        HIPCHK(hipMalloc((void**)&y->d_nmsBoxes, nB * SD_MAX_BOXES * 4 * 8)); y->owned.push_back(y->d_nmsBoxes);
        HIPCHK(hipMalloc((void**)&y->d_nmsCls, nB * SD_MAX_BOXES * 4)); y->owned.push_back(y->d_nmsCls);
        HIPCHK(hipMalloc((void**)&y->d_nmsConf, nB * SD_MAX_BOXES * 4)); y->owned.push_back(y->d_nmsConf);
        HIPCHK(hipMalloc((void**)&y->d_nmsN, nB * 4)); y->owned.push_back(y->d_nmsN);
    }
    int rc = sd_yolo_boxes_device(y, n_images, frame_cols, frame_rows, conf_threshold, nms_threshold, y->d_nmsBoxes, y->d_nmsCls, y->d_nmsConf,
                                  y->d_nmsN, stream_);
    if (rc != SD_OK || n_images == 0) return rc;
    hipStream_t s = stream_ ? (hipStream_t)stream_ : y->stream;
    HIPCHK(hipMemcpyAsync(boxes, y->d_nmsBoxes, (size_t)n_images * SD_MAX_BOXES * 4 * 8, hipMemcpyDeviceToHost, s));
    if (class_ids) HIPCHK(hipMemcpyAsync(class_ids, y->d_nmsCls, (size_t)n_images * SD_MAX_BOXES * 4, hipMemcpyDeviceToHost, s));
    if (confidences) HIPCHK(hipMemcpyAsync(confidences, y->d_nmsConf, (size_t)n_images * SD_MAX_BOXES * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(n_boxes, y->d_nmsN, (size_t)n_images * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int i = 0; i < n_images; i++)
        if (n_boxes[i] < 0) return set_err(SD_ERR_CAPACITY, "postprocess on device: more than 4096 rows above the threshold or more than SD_MAX_BOXES kept boxes");
    return SD_OK;
}


}  // extern "C"
