// HIP kernels of the detector (YOLOv3 / Darknet-53 at 640x480, src/yolo.cc:60-77, src/yolo/yolov3.cfg):
// the only GEMM-shaped work on the hot path, so the only place MFMA is used.
//   k_blob_from_image   blobFromImage(1/255, 640x480, swapRB, no crop)            yolo.cc:63
//   k_conv_mfma         convolution (+ folded batch-norm + leaky ReLU + shortcut) as implicit GEMM on
//                       v_mfma_f32_32x32x16_f16: D[cout][pixel] = W[cout][k] * X[k][pixel], f32 accumulate
//   k_upsample_concat   [upsample] x2 nearest + [route] channel concatenation
//   k_region_decode     the [yolo] region layer of cv::dnn + the confidence filter of postprocess_ (yolo.cc:163-183)
// Activations are NHWC f16 (channels innermost: one im2col K-step of 32 channels of one filter tap is 64
// contiguous bytes per pixel); weights are [cout][kh][kw][cin] f16 with batch-norm folded in.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "k_sort.h"

typedef _Float16 sd_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 sd_h4 __attribute__((ext_vector_type(4)));
typedef float sd_f16v __attribute__((ext_vector_type(16)));

struct SdConvArgs {
    const _Float16* in;      // [N][H][W][cinStride]
    const _Float16* wgt;     // [coutPad][taps][cin]
    const float* bias;       // [coutPad]
    const _Float16* res;     // shortcut source, same geometry as out (or null)
    _Float16* out;           // [N][Ho][Wo][outStride] written at channel offset outOff
    const _Float16* zero;    // >= 64 zero bytes (source of padded taps in the LDS-DMA gather)
    int N, H, W, cin, cinStride;
    int Ho, Wo, cout, coutPad, outStride, outOff, resStride;
    int ksize, stride, pad, leaky;
};

#define SD_CV_BM 64          // couts per workgroup
#define SD_CV_BN 256         // pixels per workgroup

// 4 waves, each 64 couts x 64 pixels = 2 x 2 MFMA tiles of 32 x 32.  K is walked in steps of BK channels of one
// filter tap; the global loads of step k+1 are issued into registers before the MFMAs of step k and written to
// LDS after them (one LDS buffer, two barriers per step), so the L2/HBM latency of the im2col gather overlaps
// the matrix work.  BK = 64 when the input channel count allows it, else 32.
template <int BK>
__global__ void __launch_bounds__(256) k_conv_mfma(SdConvArgs A)
{
    constexpr int LD = BK + 8;                 // LDS row length in halfs (pad: ds_read_b128 rows 16 B off the bank period)
    constexpr int CPR = BK / 8;                // 16-B chunks per row
    constexpr int XC = SD_CV_BN * CPR / 256;   // activation chunks per thread per step
    constexpr int WC = SD_CV_BM * CPR / 256;   // weight chunks per thread per step
    __shared__ __align__(16) _Float16 sW[SD_CV_BM * LD];
    __shared__ __align__(16) _Float16 sX[SD_CV_BN * LD];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int pix0 = blockIdx.x * SD_CV_BN, co0 = blockIdx.y * SD_CV_BM;
    const int npix = A.N * A.Ho * A.Wo;
    // the activation chunks this thread stages per step: chunk = tid + 256*i -> pixel chunk / CPR, quarter chunk % CPR
    int pyi[XC], pxi[XC];
    size_t pbase[XC];
    bool pok[XC];
#pragma unroll
    for (int i = 0; i < XC; i++) {
        const int chunk = tid + 256 * i;
        const int p = pix0 + chunk / CPR;
        pok[i] = p < npix;
        const int pp = pok[i] ? p : 0;
        const int n = pp / (A.Ho * A.Wo), r = pp - n * (A.Ho * A.Wo);
        const int yo = r / A.Wo, xo = r - yo * A.Wo;
        pyi[i] = yo * A.stride - A.pad; pxi[i] = xo * A.stride - A.pad;
        pbase[i] = (size_t)n * A.H * A.W;
    }
    const int taps = A.ksize * A.ksize;
    const int cchunks = A.cin / BK;
    const int ksteps = taps * cchunks;
    sd_f16v acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;
    const int r32 = lane & 31, h = lane >> 5;
    uint4 xr[XC], wr[WC];
    int t = 0, c0 = 0, kh = 0, kw = 0;         // (tap, channel offset) of the step being fetched
    auto fetch = [&]() {
#pragma unroll
        for (int i = 0; i < WC; i++) {
            const int chunk = tid + 256 * i;
            wr[i] = *(const uint4*)(A.wgt + ((size_t)(co0 + chunk / CPR) * taps + t) * A.cin + c0 + 8 * (chunk % CPR));
        }
#pragma unroll
        for (int i = 0; i < XC; i++) {
            const int chunk = tid + 256 * i;
            const int yi = pyi[i] + kh, xi = pxi[i] + kw;
            xr[i] = make_uint4(0u, 0u, 0u, 0u);
            if (pok[i] && yi >= 0 && yi < A.H && xi >= 0 && xi < A.W)
                xr[i] = *(const uint4*)(A.in + (pbase[i] + (size_t)yi * A.W + xi) * A.cinStride + c0 + 8 * (chunk % CPR));
        }
        c0 += BK;
        if (c0 == A.cin) { c0 = 0; t++; kw++; if (kw == A.ksize) { kw = 0; kh++; } }
    };
    fetch();
    for (int ks = 0; ks < ksteps; ks++) {
#pragma unroll
        for (int i = 0; i < WC; i++) { const int chunk = tid + 256 * i; *(uint4*)(sW + (chunk / CPR) * LD + 8 * (chunk % CPR)) = wr[i]; }
#pragma unroll
        for (int i = 0; i < XC; i++) { const int chunk = tid + 256 * i; *(uint4*)(sX + (chunk / CPR) * LD + 8 * (chunk % CPR)) = xr[i]; }
        __syncthreads();
        if (ks + 1 < ksteps) fetch();          // in flight during the MFMAs below
        // fragments: A = W[row r32][k = 8h + j], B = X[k = 8h + j][col r32]; BK/16 k-steps of 16
#pragma unroll
        for (int kk = 0; kk < BK / 16; kk++) {
            sd_h8 a[2], b[2];
#pragma unroll
            for (int m = 0; m < 2; m++) a[m] = *(const sd_h8*)(sW + (32 * m + r32) * LD + 16 * kk + 8 * h);
#pragma unroll
            for (int n = 0; n < 2; n++) b[n] = *(const sd_h8*)(sX + (64 * wv + 32 * n + r32) * LD + 16 * kk + 8 * h);
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int n = 0; n < 2; n++)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m], b[n], acc[m][n], 0, 0, 0);
        }
        __syncthreads();
    }
    // ---- epilogue: D col = pixel (lane & 31), row = cout (reg&3) + 8*(reg>>2) + 4*(lane>>5): 4 consecutive couts per group
#pragma unroll
    for (int n = 0; n < 2; n++) {
        const int p = pix0 + 64 * wv + 32 * n + r32;
        if (p >= npix) continue;
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int co = co0 + 32 * m + 8 * g + 4 * h;
                if (co >= A.cout) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float x = acc[m][n][4 * g + e] + A.bias[co + e];
                    if (A.leaky) x = x > 0.f ? x : 0.1f * x;
                    v[e] = (float)(_Float16)x;          // rounded before the shortcut add, as sd_conv_epilogue does: every kernel gives the same bits
                }
                if (A.res) {
                    const sd_h4 rr = *(const sd_h4*)(A.res + (size_t)p * A.resStride + co);
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] += (float)rr[e];
                }
                sd_h4 o;
#pragma unroll
                for (int e = 0; e < 4; e++) o[e] = (_Float16)v[e];
                _Float16* dst = A.out + (size_t)p * A.outStride + A.outOff + co;
                if (co + 3 < A.cout) *(sd_h4*)dst = o;
                else for (int e = 0; e < 4 && co + e < A.cout; e++) dst[e] = o[e];
            }
    }
}

// 3x3 / stride 1 / pad 1 convolution for feature maps up to SD_C3_MAXW wide: the im2col of k_conv_mfma re-reads every
// activation nine times (once per tap) and once more per 64-filter tile, which makes those layers L2-bound.  Here a
// workgroup owns 256 CONSECUTIVE pixels of the flattened [N][H][W] index and 128 filters.  For one 32-channel chunk the
// flattened range [p0 - W - 1, p0 + 256 + W] (every tap's source pixel of every tile pixel) is staged in LDS ONCE and all
// nine taps read their B fragments from it at row offsets kh*W + kw; out-of-image taps read a zero row.  Only the
// 128 x 32 weight tile changes per tap (double-buffered, prefetched through registers, one barrier per tap).
// Wave tile: 128 filters x 64 pixels = 4 x 2 MFMA 32x32x16 tiles.
#define SD_C3_BM 128
#define SD_C3_BN 256
#define SD_C3_BK 32
#define SD_C3_LD 40          // LDS row length in halfs (80 B: conflict-free ds_read_b128 over consecutive rows)
#define SD_C3_MAXW 80
#define SD_C3_XROWS (SD_C3_BN + 2 * SD_C3_MAXW + 2)
#define SD_C3_XCH ((SD_C3_XROWS * 4 + 255) / 256)      // 16-B activation chunks per thread per channel chunk
__global__ void __launch_bounds__(256, 2) k_conv3x3_flat(SdConvArgs A)
{
    __shared__ __align__(16) _Float16 sX[(SD_C3_XROWS + 1) * SD_C3_LD];
    __shared__ __align__(16) _Float16 sW[2][SD_C3_BM * SD_C3_LD];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r32 = lane & 31, h = lane >> 5;
    const int W = A.W, H = A.H;
    const int npix = A.N * H * W;
    const int p0 = blockIdx.x * SD_C3_BN, co0 = blockIdx.y * SD_C3_BM;
    const int xrows = SD_C3_BN + 2 * W + 2, ZR = xrows;          // ZR: the all-zero row
    if (tid < 5) *(uint4*)(sX + ZR * SD_C3_LD + 8 * tid) = make_uint4(0u, 0u, 0u, 0u);
    // B-fragment rows of this lane's two pixels
    int jy[2], jx[2], jrow[2];
    bool jok[2];
#pragma unroll
    for (int n = 0; n < 2; n++) {
        const int j = 64 * wv + 32 * n + r32, p = p0 + j;
        jok[n] = p < npix;
        const int q = (jok[n] ? p : 0) / W;
        jx[n] = (jok[n] ? p : 0) - q * W; jy[n] = q % H; jrow[n] = j;
    }
    sd_f16v acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;
    const int nchunks = A.cin / SD_C3_BK;
    uint4 xr[SD_C3_XCH], wr[2];
    auto fetchX = [&](int c0) {
#pragma unroll
        for (int i = 0; i < SD_C3_XCH; i++) {
            const int chunk = tid + 256 * i, r = chunk >> 2;
            const int g = p0 - W - 1 + r;
            xr[i] = make_uint4(0u, 0u, 0u, 0u);
            if (r < xrows && g >= 0 && g < npix) xr[i] = *(const uint4*)(A.in + (size_t)g * A.cinStride + c0 + 8 * (chunk & 3));
        }
    };
    auto storeX = [&]() {
#pragma unroll
        for (int i = 0; i < SD_C3_XCH; i++) {
            const int chunk = tid + 256 * i, r = chunk >> 2;
            if (r < xrows) *(uint4*)(sX + r * SD_C3_LD + 8 * (chunk & 3)) = xr[i];
        }
    };
    auto fetchW = [&](int tap, int c0) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int chunk = tid + 256 * i;
            wr[i] = *(const uint4*)(A.wgt + ((size_t)(co0 + (chunk >> 2)) * 9 + tap) * A.cin + c0 + 8 * (chunk & 3));
        }
    };
    auto storeW = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; i++) { const int chunk = tid + 256 * i; *(uint4*)(sW[buf] + (chunk >> 2) * SD_C3_LD + 8 * (chunk & 3)) = wr[i]; }
    };
    fetchX(0); fetchW(0, 0);
    storeX(); storeW(0);
    __syncthreads();
    int step = 0;
    for (int ch = 0; ch < nchunks; ch++) {
        const bool moreX = ch + 1 < nchunks;
        if (moreX) fetchX((ch + 1) * SD_C3_BK);            // lands while the nine taps run
#pragma unroll
        for (int tap = 0; tap < 9; tap++, step++) {
            const int buf = step & 1;
            const bool moreW = tap < 8 || moreX;
            if (moreW) fetchW(tap < 8 ? tap + 1 : 0, (tap < 8 ? ch : ch + 1) * SD_C3_BK);
            const int kh = tap / 3, kw = tap % 3;
            int brow[2];
#pragma unroll
            for (int n = 0; n < 2; n++) {
                const int yy = jy[n] + kh - 1, xx = jx[n] + kw - 1;
                brow[n] = (jok[n] && yy >= 0 && yy < H && xx >= 0 && xx < W) ? jrow[n] + kh * W + kw : ZR;
            }
#pragma unroll
            for (int kk = 0; kk < 2; kk++) {
                sd_h8 a[4], b[2];
#pragma unroll
                for (int m = 0; m < 4; m++) a[m] = *(const sd_h8*)(sW[buf] + (32 * m + r32) * SD_C3_LD + 16 * kk + 8 * h);
#pragma unroll
                for (int n = 0; n < 2; n++) b[n] = *(const sd_h8*)(sX + brow[n] * SD_C3_LD + 16 * kk + 8 * h);
#pragma unroll
                for (int m = 0; m < 4; m++)
#pragma unroll
                    for (int n = 0; n < 2; n++)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m], b[n], acc[m][n], 0, 0, 0);
            }
            if (moreW) storeW(buf ^ 1);                    // the other buffer: its readers finished before the last barrier
            if (tap == 8 && moreX) { __syncthreads(); storeX(); }
            __syncthreads();
        }
    }
#pragma unroll
    for (int n = 0; n < 2; n++) {
        const int p = p0 + 64 * wv + 32 * n + r32;
        if (p >= npix) continue;
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int co = co0 + 32 * m + 8 * g + 4 * h;
                if (co >= A.cout) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float x = acc[m][n][4 * g + e] + A.bias[co + e];
                    if (A.leaky) x = x > 0.f ? x : 0.1f * x;
                    v[e] = (float)(_Float16)x;          // rounded before the shortcut add, as sd_conv_epilogue does: every kernel gives the same bits
                }
                if (A.res) {
                    const sd_h4 rr = *(const sd_h4*)(A.res + (size_t)p * A.resStride + co);
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] += (float)rr[e];
                }
                sd_h4 o;
#pragma unroll
                for (int e = 0; e < 4; e++) o[e] = (_Float16)v[e];
                _Float16* dst = A.out + (size_t)p * A.outStride + A.outOff + co;
                if (co + 3 < A.cout) *(sd_h4*)dst = o;
                else for (int e = 0; e < 4 && co + e < A.cout; e++) dst[e] = o[e];
            }
    }
}

// The same flattened 3x3 scheme with every staging copy done by LDS-DMA (global_load_lds_dwordx4), so no VGPR holds
// prefetched tiles and the copies stay in flight across barriers:
//   * 512 threads = 8 waves, tile 128 filters x 512 consecutive pixels, wave tile 128 x 64 (4 x 2 MFMA 32x32x16);
//   * X (activations of one 32-channel chunk, flattened range + halo) is double-buffered: chunk c+1 is requested at tap 1
//     of chunk c and has eight taps to land;
//   * W (128 x 32 weights of one tap) goes through a 5-slot ring, requested three taps ahead; one raw s_barrier per tap,
//     preceded by a COUNTED s_waitcnt vmcnt(N) that only retires the tile about to be read;
//   * an LDS-DMA wave-instruction writes 1 KiB linearly (16 rows x 64 B), so rows cannot be padded: the 16-byte slot a
//     lane fills holds channel group q ^ ((row >> 2) & 3) (swizzle applied on the SOURCE address) and fragment reads apply
//     the same XOR, which keeps every ds_read_b128 lane group on 16 distinct bank quads.
#define SD_G3_BM 128
#define SD_G3_BN 512
#define SD_G3_NW 5
#define SD_G3_WBYTES (SD_G3_BM * 64)
// per maximum map width MAXW (80 for the 80/40/20-wide maps, 160 for the 160-wide ones):
//   ZROW    (512 + 2*MAXW + 2 rows) rounded up to 16; row ZROW is all zero
//   XBYTES  one activation buffer, (ZROW + 1) rows x 64 B rounded to 256 B so both buffers bank alike
//   XPIECES LDS-DMA instructions per wave per activation chunk (ZROW/16 groups of 16 rows over 8 waves)
#define SD_G3_ZROW(MAXW) ((SD_G3_BN + 2 * (MAXW) + 2 + 15) / 16 * 16)
#define SD_G3_XBYTES(MAXW) (((SD_G3_ZROW(MAXW) + 1) * 64 + 255) / 256 * 256)
#define SD_G3_XPIECES(MAXW) ((SD_G3_ZROW(MAXW) / 16 + 7) / 8)
#define SD_G3_LDS(MAXW) (2 * SD_G3_XBYTES(MAXW) + SD_G3_NW * SD_G3_WBYTES)
typedef __attribute__((address_space(3))) void sd_lds_void;
typedef const __attribute__((address_space(1))) void sd_glb_void;
#define SD_GLDS16(gsrc, ldst) __builtin_amdgcn_global_load_lds((sd_glb_void*)(gsrc), (sd_lds_void*)(ldst), 16, 0, 0)
template <int N> __device__ __forceinline__ void sd_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Epilogue shared by the LDS-DMA kernels: wave tile 128 filters x 64 pixels held as acc[4][2] (D layout of
// v_mfma_f32_32x32x16: column = pixel lane&31, rows = filters (reg&3) + 8*(reg>>2) + 4*(lane>>5)).  A lane owns 4
// consecutive filters of ONE pixel, so direct stores would be 8-byte pieces 2*outStride bytes apart.  Instead each wave
// transposes through its private LDS region (the staging buffers are dead by now): bias + leaky ReLU in f32, f16 rows
// of 64 filters (+16 B pad), then every lane moves 16-byte pieces so one wave-instruction covers 8 pixels x 128
// contiguous bytes for the shortcut read and the store alike.  Two passes of 64 filters keep the region at 9 KiB/wave.
#define SD_EP_ROW 144
#define SD_EP_WAVE (64 * SD_EP_ROW)
__device__ __forceinline__ void sd_conv_epilogue(const SdConvArgs& A, sd_f16v (&acc)[4][2], unsigned char* smem, int wv, int pbase,
                                                 int co0, int npix, int lane)
{
    const int r32 = lane & 31, h = lane >> 5;
    unsigned char* reg = smem + wv * SD_EP_WAVE;
    const int cend = (A.cout + 7) & ~7;                       // whole 16-byte pieces may be stored up to here
    const bool wide = cend <= A.outStride - A.outOff;
    __syncthreads();                                           // every wave is done with the staging buffers
#pragma unroll
    for (int mh = 0; mh < 2; mh++) {
        if (co0 + 64 * mh >= A.cout) break;
#pragma unroll
        for (int mm = 0; mm < 2; mm++)
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int cl = 32 * mm + 8 * g + 4 * h, co = co0 + 64 * mh + cl;
                    sd_h4 o;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        float x = acc[2 * mh + mm][n][4 * g + e] + A.bias[co + e];      // bias rows are padded to the filter tile
                        if (A.leaky) x = x > 0.f ? x : 0.1f * x;
                        o[e] = (_Float16)x;
                    }
                    *(sd_h4*)(reg + (32 * n + r32) * SD_EP_ROW + cl * 2) = o;
                }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const int c = lane + 64 * t, px = c >> 3, part = c & 7;
            const int p = pbase + px, co = co0 + 64 * mh + 8 * part;
            if (p >= npix || co >= A.cout) continue;
            sd_h8 v = *(const sd_h8*)(reg + px * SD_EP_ROW + part * 16);
            _Float16* dst = A.out + (size_t)p * A.outStride + A.outOff + co;
            if (wide && co + 8 <= cend) {
                if (A.res) {
                    const sd_h8 rr = *(const sd_h8*)(A.res + (size_t)p * A.resStride + co);
#pragma unroll
                    for (int e = 0; e < 8; e++) v[e] = (_Float16)((float)v[e] + (float)rr[e]);
                }
                *(sd_h8*)dst = v;
            } else {
                for (int e = 0; e < 8 && co + e < A.cout; e++) {
                    float x = (float)v[e];
                    if (A.res) x += (float)A.res[(size_t)p * A.resStride + co + e];
                    dst[e] = (_Float16)x;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();                       // pass 2 overwrites the region
    }
}

template <int MAXW>
__global__ void __launch_bounds__(512, 1) k_conv3x3_glds(SdConvArgs A)
{
    constexpr int ZROW = SD_G3_ZROW(MAXW), XBYTES = SD_G3_XBYTES(MAXW), XPIECES = SD_G3_XPIECES(MAXW);
    extern __shared__ __align__(1024) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r32 = lane & 31, h = lane >> 5;
    const int W = A.W, H = A.H;
    const int npix = A.N * H * W;
    const int p0 = blockIdx.x * SD_G3_BN, co0 = blockIdx.y * SD_G3_BM;
    const int xrows = SD_G3_BN + 2 * W + 2, ngroups = (xrows + 15) >> 4;
    if (tid < 8) *(uint4*)(smem + (tid >> 2) * XBYTES + ZROW * 64 + 16 * (tid & 3)) = make_uint4(0u, 0u, 0u, 0u);
    // ---- LDS-DMA sources of this lane: slot (row = lane>>2, q' = lane&3) of a 16-row group holds channel group q' ^ sw
    const int qsrc = (lane & 3) ^ ((lane >> 4) & 3);
    const _Float16* xsrc[XPIECES];
    int xdst[XPIECES];
#pragma unroll
    for (int i = 0; i < XPIECES; i++) {
        int g = wv + 8 * i;
        if (g >= ngroups) g -= 8;                       // surplus piece: rewrite this wave's previous group with the same bytes
        const int r = 16 * g + (lane >> 2);
        int gp = p0 - W - 1 + r;                        // rows outside [0, npix) are only ever addressed by masked taps
        gp = gp < 0 ? 0 : (gp >= npix ? npix - 1 : gp);
        xsrc[i] = A.in + (size_t)gp * A.cinStride + 8 * qsrc;
        xdst[i] = 1024 * g;
    }
    const _Float16* wsrc = A.wgt + (size_t)(co0 + 16 * wv + (lane >> 2)) * 9 * A.cin + 8 * qsrc;
    unsigned char* const wring = smem + 2 * XBYTES;
    const int nchunks = A.cin / 32;
    auto issueX = [&](int ch) {
        unsigned char* xb = smem + (ch & 1) * XBYTES;
#pragma unroll
        for (int i = 0; i < XPIECES; i++) SD_GLDS16(xsrc[i] + ch * 32, xb + xdst[i]);
    };
    int wtap = 0, wch = 0, wslot = 0;                   // the next weight stage to request
    auto issueW = [&]() {
        SD_GLDS16(wsrc + (size_t)wtap * A.cin + wch * 32, wring + wslot * SD_G3_WBYTES + 1024 * wv);
        wslot = wslot == SD_G3_NW - 1 ? 0 : wslot + 1;
        if (!(wtap == 8 && wch == nchunks - 1)) { wtap++; if (wtap == 9) { wtap = 0; wch++; } }     // saturate: surplus requests re-fetch the last stage
    };
    // ---- fragment addressing
    int jy[2], jx[2], jrow[2];
    bool jok[2];
#pragma unroll
    for (int n = 0; n < 2; n++) {
        const int j = 64 * wv + 32 * n + r32, p = p0 + j;
        jok[n] = p < npix;
        const int q = (jok[n] ? p : 0) / W;
        jx[n] = (jok[n] ? p : 0) - q * W; jy[n] = q % H; jrow[n] = j;
    }
    const int aoff = r32 * 64 + ((h ^ ((r32 >> 2) & 3)) << 4);
    sd_f16v acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;
    __syncthreads();                                    // zero rows written
    issueX(0); issueW(); issueW(); issueW();
    int rslot = 0;
    for (int ch = 0; ch < nchunks; ch++) {
        const unsigned char* xb = smem + (ch & 1) * XBYTES;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            issueW();                                    // stage s+3 -> the slot read at step s-2
            if (tap == 1) issueX(ch + 1 < nchunks ? ch + 1 : ch);
            // retire W(s) (and X(ch) at tap 0): everything requested after it may stay in flight
            if (tap >= 1 && tap <= 4) sd_wait_vmcnt<3 + XPIECES>();
            else sd_wait_vmcnt<3>();
            __builtin_amdgcn_s_barrier();
            const unsigned char* wb = wring + rslot * SD_G3_WBYTES;
            rslot = rslot == SD_G3_NW - 1 ? 0 : rslot + 1;
            const int kh = tap / 3, kw = tap % 3;
            int boff[2];
#pragma unroll
            for (int n = 0; n < 2; n++) {
                const int yy = jy[n] + kh - 1, xx = jx[n] + kw - 1;
                const int row = (jok[n] && yy >= 0 && yy < H && xx >= 0 && xx < W) ? jrow[n] + kh * W + kw : ZROW;
                boff[n] = row * 64 + ((h ^ ((row >> 2) & 3)) << 4);
            }
#pragma unroll
            for (int kk = 0; kk < 2; kk++) {
                sd_h8 a[4], b[2];
#pragma unroll
                for (int m = 0; m < 4; m++) a[m] = *(const sd_h8*)(wb + ((aoff ^ (32 * kk)) + 2048 * m));
#pragma unroll
                for (int n = 0; n < 2; n++) b[n] = *(const sd_h8*)(xb + (boff[n] ^ (32 * kk)));
#pragma unroll
                for (int m = 0; m < 4; m++)
#pragma unroll
                    for (int n = 0; n < 2; n++)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m], b[n], acc[m][n], 0, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // surplus LDS-DMA must not outlive the workgroup's LDS
    sd_conv_epilogue(A, acc, smem, wv, p0 + 64 * wv, co0, npix, lane);
}

// Implicit-GEMM convolution with LDS-DMA staging for everything the flattened 3x3 kernel does not take: 1x1 layers
// (KS = 1, a plain GEMM over pixels), and 3x3 layers with stride 2 or on maps wider than SD_C3_MAXW (KS = 3, each
// stage gathers one filter tap's rows; padded taps read a zero page).  NWAVES waves, tile 128 filters x 64*NWAVES pixels,
// three-stage ring of {X: pixels x 32 channels, W: 128 x 32}; stage s+2 is requested right after the barrier of step s
// (every wave has then finished reading the slot it overwrites), and the counted vmcnt before the next barrier leaves
// exactly that one stage in flight.  The 1x1 layers move cin + cout halfs per pixel for 2*cin*cout flops: they are bound
// by the activation stream, which is read once per 128-filter tile in full 64-byte pieces.
template <int NWAVES, int KS>
__global__ void __launch_bounds__(64 * NWAVES, NWAVES == 8 ? 1 : 2) k_conv_glds(SdConvArgs A)
{
    constexpr int BN = 64 * NWAVES;
    constexpr int XB = BN * 64, STAGE = XB + SD_G3_WBYTES;
    constexpr int WP = 8 / NWAVES;                    // weight pieces per wave per stage
    constexpr int TAPS = KS * KS;
    extern __shared__ __align__(1024) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r32 = lane & 31, h = lane >> 5;
    const int npix = A.N * A.Ho * A.Wo;
    const int p0 = blockIdx.x * BN, co0 = blockIdx.y * SD_G3_BM;
    const int qsrc = (lane & 3) ^ ((lane >> 4) & 3);
    const _Float16* xsrc[4];                          // KS == 1: the pixel's row; KS == 3: the image base
    int xy[4], xx[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int gp = p0 + 16 * (wv + NWAVES * i) + (lane >> 2);
        gp = gp >= npix ? npix - 1 : gp;
        if (KS == 1) { xsrc[i] = A.in + (size_t)gp * A.cinStride + 8 * qsrc; xy[i] = xx[i] = 0; }
        else {
            const int n = gp / (A.Ho * A.Wo), r = gp - n * (A.Ho * A.Wo);
            const int yo = r / A.Wo, xo = r - yo * A.Wo;
            xy[i] = yo * A.stride - A.pad; xx[i] = xo * A.stride - A.pad;
            xsrc[i] = A.in + (size_t)n * A.H * A.W * A.cinStride + 8 * qsrc;
        }
    }
    const _Float16* wsrc[WP];
#pragma unroll
    for (int i = 0; i < WP; i++) wsrc[i] = A.wgt + (size_t)(co0 + 16 * (wv + NWAVES * i) + (lane >> 2)) * TAPS * A.cin + 8 * qsrc;
    const int nsteps = TAPS * (A.cin / 32);
    int it = 0, ic0 = 0, ikh = 0, ikw = 0;            // (tap, channel offset) of the next stage to request
    auto issue = [&](int slot) {
        unsigned char* sb = smem + slot * STAGE;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const _Float16* src;
            if (KS == 1) src = xsrc[i] + ic0;
            else {
                const int yi = xy[i] + ikh, xi = xx[i] + ikw;
                src = (yi >= 0 && yi < A.H && xi >= 0 && xi < A.W) ? xsrc[i] + ((size_t)yi * A.W + xi) * A.cinStride + ic0 : A.zero + 8 * (lane & 3);
            }
            SD_GLDS16(src, sb + 1024 * (wv + NWAVES * i));
        }
#pragma unroll
        for (int i = 0; i < WP; i++) SD_GLDS16(wsrc[i] + (size_t)it * A.cin + ic0, sb + XB + 1024 * (wv + NWAVES * i));
        if (!(it == TAPS - 1 && ic0 + 32 == A.cin)) {  // saturate: surplus requests re-fetch the last stage
            ic0 += 32;
            if (ic0 == A.cin) { ic0 = 0; it++; ikw++; if (ikw == KS) { ikw = 0; ikh++; } }
        }
    };
    const int aoff = r32 * 64 + ((h ^ ((r32 >> 2) & 3)) << 4);
    int boff[2];
#pragma unroll
    for (int n = 0; n < 2; n++) { const int row = 64 * wv + 32 * n + r32; boff[n] = row * 64 + ((h ^ ((row >> 2) & 3)) << 4); }
    sd_f16v acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;
    issue(0); issue(1);
    int slot = 0;
    for (int s = 0; s < nsteps; s++) {
        if (NWAVES == 8) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue(slot == 0 ? 2 : slot - 1);                                       // stage s + 2 -> slot (s + 2) % 3
        const unsigned char* xb = smem + slot * STAGE;
        const unsigned char* wb = xb + XB;
        slot = slot == 2 ? 0 : slot + 1;
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            sd_h8 a[4], b[2];
#pragma unroll
            for (int m = 0; m < 4; m++) a[m] = *(const sd_h8*)(wb + ((aoff ^ (32 * kk)) + 2048 * m));
#pragma unroll
            for (int n = 0; n < 2; n++) b[n] = *(const sd_h8*)(xb + (boff[n] ^ (32 * kk)));
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int n = 0; n < 2; n++)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m], b[n], acc[m][n], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // surplus LDS-DMA must not outlive the workgroup's LDS
    sd_conv_epilogue(A, acc, smem, wv, p0 + 64 * wv, co0, npix, lane);
}

// blobFromImage(image, 1/255, Size(640,480), Scalar(0,0,0), swapRB = true, crop = false): bilinear resize of
// the 8-bit image (OpenCV resize INTER_LINEAR fixed-point path, per channel), swap R and B, scale to [0,1].
// Output NHWC f16, 4 channels (3 + a zero); read directly by k_conv_first.
__global__ void __launch_bounds__(256) k_blob_from_image(const uint8_t* __restrict__ src, int sw, int sh, size_t sstride,
                                                         size_t spitch, const short4* __restrict__ ct,
                                                         const short4* __restrict__ rt, _Float16* __restrict__ dst,
                                                         int dw, int dh, int swapRB)
{
    const int img = blockIdx.z;
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= dw || y >= dh) return;
    const short4 ce = ct[x], re = rt[y];
    const int sx = ce.x, sx1 = min(sx + 1, sw - 1);
    const int r0 = min(max((int)re.x, 0), sh - 1), r1 = min(max((int)re.x + 1, 0), sh - 1);
    const uint8_t* S0 = src + (size_t)img * spitch + (size_t)r0 * sstride;
    const uint8_t* S1 = src + (size_t)img * spitch + (size_t)r1 * sstride;
    float ch[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int h0 = S0[3 * sx + c] * ce.y + S0[3 * sx1 + c] * ce.z;
        const int h1 = S1[3 * sx + c] * ce.y + S1[3 * sx1 + c] * ce.z;
        const int v = ((((int)re.y * (h0 >> 4)) >> 16) + (((int)re.z * (h1 >> 4)) >> 16) + 2) >> 2;
        ch[c] = (float)(v & 255) * (float)(1 / 255.0);
    }
    sd_h4 o;
    o[0] = (_Float16)(swapRB ? ch[2] : ch[0]); o[1] = (_Float16)ch[1]; o[2] = (_Float16)(swapRB ? ch[0] : ch[2]); o[3] = (_Float16)0.f;
    *(sd_h4*)(dst + ((size_t)img * dh * dw + (size_t)y * dw + x) * 4) = o;
}

// First convolution (3 input channels, 3x3, stride 1, pad 1, <= 32 filters) straight from the 4-channel blob: the GEMM K
// axis is k = tap*4 + c (9 taps -> 36, padded to 48 = 3 MFMA k-steps).  With the 32x32x16 fragment layout a lane's 8
// B-operand halfs are exactly two taps of its own pixel, so the im2col gather is two 8-byte loads per k-step and needs no
// LDS; the 32 x 48 weight tile lives in registers.  One wave = 64 pixels x 32 filters.
__global__ void __launch_bounds__(256) k_conv_first(const _Float16* __restrict__ blob4, const _Float16* __restrict__ wgt,
                                                    const float* __restrict__ bias, _Float16* __restrict__ out, int N, int H, int W,
                                                    int cout, int outStride, int leaky)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r32 = lane & 31, h = lane >> 5;
    const size_t npix = (size_t)N * H * W;
    sd_h8 a[3];
#pragma unroll
    for (int kk = 0; kk < 3; kk++) a[kk] = *(const sd_h8*)(wgt + r32 * 48 + 16 * kk + 8 * h);
#pragma unroll
    for (int n = 0; n < 2; n++) {
        const size_t p = (size_t)blockIdx.x * 256 + 64 * wv + 32 * n + r32;
        const bool ok = p < npix;
        const size_t pp = ok ? p : 0;
        const int x = (int)(pp % W);
        const size_t q = pp / W;
        const int y = (int)(q % H);
        const _Float16* img = blob4 + (q / H) * (size_t)H * W * 4;
        sd_f16v acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 3; kk++) {
            sd_h8 b;
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int tp = 4 * kk + 2 * h + e;
                const int ky = (tp * 11) >> 5, kx = tp - 3 * ky;          // tp / 3, tp % 3 for tp < 12
                const int yy = y + ky - 1, xx = x + kx - 1;
                sd_h4 v = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
                if (ok && tp < 9 && yy >= 0 && yy < H && xx >= 0 && xx < W) v = *(const sd_h4*)(img + ((size_t)yy * W + xx) * 4);
                b[4 * e] = v[0]; b[4 * e + 1] = v[1]; b[4 * e + 2] = v[2]; b[4 * e + 3] = v[3];
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kk], b, acc, 0, 0, 0);
        }
        if (!ok) continue;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int co = 8 * g + 4 * h;
            if (co >= cout) continue;
            sd_h4 o;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float v = acc[4 * g + e] + bias[co + e];
                if (leaky) v = v > 0.f ? v : 0.1f * v;
                o[e] = (_Float16)v;
            }
            _Float16* dst = out + p * outStride + co;
            if (co + 3 < cout) *(sd_h4*)dst = o;
            else for (int e = 0; e < 4 && co + e < cout; e++) dst[e] = o[e];
        }
    }
}

// [upsample] stride 2 (nearest) of `a` (C1 channels, h x w) into channels [0, C1) of `out` (2h x 2w, C1 + C2
// channels), and [route] concatenation of `b` (C2 channels, 2h x 2w) into channels [C1, C1 + C2).
__global__ void __launch_bounds__(256) k_upsample_concat(const _Float16* __restrict__ a, int C1, int h, int w,
                                                         const _Float16* __restrict__ b, int C2,
                                                         _Float16* __restrict__ out, int N)
{
    const int Ct = C1 + C2, H2 = 2 * h, W2 = 2 * w;
    const size_t total = (size_t)N * H2 * W2 * (Ct / 8);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c8 = (int)(i % (Ct / 8));
        const size_t p = i / (Ct / 8);
        const int x = (int)(p % W2);
        const size_t q = p / W2;
        const int y = (int)(q % H2), n = (int)(q / H2);
        uint4 v;
        if (c8 * 8 < C1) v = *(const uint4*)(a + (((size_t)n * h + (y >> 1)) * w + (x >> 1)) * C1 + c8 * 8);
        else v = *(const uint4*)(b + (((size_t)n * H2 + y) * W2 + x) * C2 + (c8 * 8 - C1));
        *(uint4*)(out + p * Ct + c8 * 8) = v;
    }
}

// cv::dnn RegionLayer for YOLOv3 (logistic activations, classes scaled by objectness, thresh 0.001 default) fused
// with the confidence filter of yolov3Segment::postprocess_ (yolo.cc:163-183): one thread per (cell, anchor) row.
// Rows are numbered as cv::dnn emits them: head by head, then (y, x, anchor).  Surviving rows are appended to a
// compact list; their order is restored on the host before NMSBoxes (it needs the original row order for ties).
struct SdDet { int row; int cls; float conf; float cx, cy, w, h; };   // 28 B
template <typename T>
__global__ void __launch_bounds__(256) k_region_decode(const T* __restrict__ head, int hs /*channel stride*/,
                                                       int gh, int gw, int N, float aw0, float ah0, float aw1,
                                                       float ah1, float aw2, float ah2, int netW, int netH,
                                                       float confThreshold, int rowBase, SdDet* __restrict__ dets,
                                                       int* __restrict__ ndet, int detCap, float* __restrict__ rawOut)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int perImg = gh * gw * 3;
    const int lane = threadIdx.x & 63;
    const bool valid = i < N * perImg;
    const int ii = valid ? i : 0;
    const int n = ii / perImg, r = ii - n * perImg;
    const int a = r % 3, cell = r / 3;
    const int y = cell / gw, x = cell - y * gw;
    const T* t = head + ((size_t)n * gh * gw + cell) * hs + a * 85;
    auto sig = [](float v) { return 1.f / (1.f + expf(-v)); };
    const float obj = valid ? sig((float)t[4]) : 0.f;
    // class scores are obj * sigmoid(.) <= obj (a product with a factor <= 1 never rounds above obj), so a row whose
    // objectness is not above the threshold cannot pass the filter; only the other rows (all rows when the raw
    // tensor is requested) have their 80 classes scored, and those are scored by the whole wave: lane c takes
    // classes c and c + 64 (one coalesced 160-byte read), then a butterfly keeps the first maximum as minMaxLoc does.
    const bool want = valid && (rawOut != nullptr || obj > confThreshold);
    float best = 0.f;
    int bc = 0;
    unsigned long long todo = __ballot(want);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int si = __shfl(ii, src);
        const float sobj = __shfl(obj, src);
        const int sn = si / perImg, sr = si - sn * perImg;
        const T* st = head + ((size_t)sn * gh * gw + sr / 3) * hs + (sr % 3) * 85 + 5;
        float p0 = sobj * sig((float)st[lane]);
        if (!(p0 > 0.001f)) p0 = 0.f;                 // region layer `thresh`
        float p1 = 0.f;
        if (lane < 16) { p1 = sobj * sig((float)st[64 + lane]); if (!(p1 > 0.001f)) p1 = 0.f; }
        if (rawOut) {
            float* o = rawOut + (size_t)(rowBase + sr) * 85 + 5;
            o[lane] = p0;
            if (lane < 16) o[64 + lane] = p1;
        }
        float m = p0; int mc = lane;
        if (p1 > m) { m = p1; mc = lane + 64; }        // strictly greater: the lower class index wins ties
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float om = __shfl_xor(m, d); const int oc = __shfl_xor(mc, d);
            if (om > m || (om == m && oc < mc)) { m = om; mc = oc; }
        }
        if (lane == src) { best = m; bc = m > 0.f ? mc : 0; }
    }
    // Slots of the surviving rows: counted per WORKGROUP in LDS (its 256 consecutive rows belong to two images, or to a few more of a very small
    // network: at most 256 / 3 + 2), then ONE global atomic per image and workgroup.  One atomic per row -- a thousand of them on one image's counter with the synthetic heads, every one a round trip to L2
    // behind the others -- was what a launch waited for (0.86 ms for the 80 x 60 head of a 256-image batch).  The list's order is free (k_yolo_nms sorts).
    __shared__ int s_cnt[128], s_base[128];
    if (threadIdx.x < 128) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int n0 = (int)(((size_t)blockIdx.x * 256) / (size_t)perImg);
    const bool keep = want && best > confThreshold;
    int local = 0;
    if (keep) local = atomicAdd(&s_cnt[n - n0], 1);
    __syncthreads();
    if (threadIdx.x < 128 && s_cnt[threadIdx.x] > 0) s_base[threadIdx.x] = atomicAdd(&ndet[n0 + threadIdx.x], s_cnt[threadIdx.x]);
    __syncthreads();
    if (!want) return;
    const float aw = a == 0 ? aw0 : a == 1 ? aw1 : aw2, ah = a == 0 ? ah0 : a == 1 ? ah1 : ah2;
    const float cx = (sig((float)t[0]) + (float)x) / (float)gw;
    const float cy = (sig((float)t[1]) + (float)y) / (float)gh;
    const float bw = expf((float)t[2]) * aw / (float)netW;
    const float bh = expf((float)t[3]) * ah / (float)netH;
    if (rawOut) {
        float* o = rawOut + (size_t)(rowBase + r) * 85;
        o[0] = cx; o[1] = cy; o[2] = bw; o[3] = bh; o[4] = obj;
    }
    if (keep) {
        const int slot = s_base[n - n0] + local;
        if (slot < detCap) {
            SdDet d; d.row = rowBase + r; d.cls = bc; d.conf = best; d.cx = cx; d.cy = cy; d.w = bw; d.h = bh;
            dets[(size_t)n * detCap + slot] = d;
        }
    }
}

// yolov3Segment::postprocess_ after the confidence filter (yolo.cc:186-205) on the device, one workgroup per image:
// boxes = (int)(centre * frame dims) etc. as the reference truncates them, cv::dnn::NMSBoxes(conf, nms) = stable sort by
// score (descending; equal scores keep cv::dnn's row order) + greedy keep while the overlap with every kept box is
// <= nms (1 - jaccardDistance on cv::Rect, f64), the class filter {person, bicycle, car, bus, truck} ("motorcycle" never
// matches coco.names' "motorbike") and rectCenterScale(box, (-0.2 w, 0.6 h)).  Output: up to SD_MAX_BOXES boxes per image as
// cv::Rect2d (x, y, w, h f64) in kept order, nOut[image] (or -1 - count when a capacity was exceeded).
#define SD_NMS_MAXDET 4096      // rows above the confidence threshold per image handled on the device
#define SD_NMS_MAXKEEP 512      // boxes NMS may keep before the class filter
#define SD_NMS_LDS (SD_NMS_MAXDET * (8 + 16 + 1))
// Greedy NMS without a barrier per candidate: candidate t survives iff no KEPT earlier candidate overlaps it by more than the
// threshold, so the workgroup iterates over the kept boxes only (tens) -- next unsuppressed candidate by a block-wide minimum, then
// every later candidate tests itself against it in parallel -- instead of over all candidates (a thousand with three barriers each).
__global__ void __launch_bounds__(256) k_yolo_nms(const SdDet* __restrict__ dets, const int* __restrict__ ndet, int detCap, int frameCols,
                                                  int frameRows, float confThreshold, float nmsThreshold, double* __restrict__ boxesOut,
                                                  int* __restrict__ clsOut, float* __restrict__ confOut, int* __restrict__ nOut)
{
    extern __shared__ __align__(16) unsigned char nms_smem[];                                             // SD_NMS_LDS bytes
    unsigned long long* keys = (unsigned long long*)nms_smem;                                             // [SD_NMS_MAXDET]
    int* s_x = (int*)(keys + SD_NMS_MAXDET); int* s_y = s_x + SD_NMS_MAXDET; int* s_w = s_y + SD_NMS_MAXDET; int* s_h = s_w + SD_NMS_MAXDET;   // int boxes of the sorted candidates
    unsigned char* s_dead = (unsigned char*)(s_h + SD_NMS_MAXDET);
    __shared__ int s_kept[SD_NMS_MAXKEEP];
    __shared__ int s_wmin[8];
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nd = ndet[img];
    const SdDet* D = dets + (size_t)img * detCap;
    if (nd > SD_NMS_MAXDET || nd > detCap) { if (tid == 0) nOut[img] = -1 - nd; return; }
    int n2 = 64;
    while (n2 < nd) n2 <<= 1;
    // sort key: score descending, then cv::dnn row ascending (the stable order of NMSBoxes' input); the slot rides in the low bits
    for (int t = tid; t < n2; t += 256) {
        unsigned long long k = ~0ull;
        if (t < nd && D[t].conf > confThreshold) k = ((unsigned long long)(~__float_as_uint(D[t].conf)) << 32) | ((unsigned)D[t].row << 13) | (unsigned)t;
        keys[t] = k;
    }
    __syncthreads();
    sd_block_sort64(keys, n2, tid, 256);
    int nc = 0;                                                // candidates above the threshold (they sort first)
    for (int t = tid; t < nd; t += 256) {
        const bool live = keys[t] != ~0ull;
        if (live) {
            const SdDet d = D[(int)(keys[t] & 0x1FFFu)];
            const int centerX = (int)(d.cx * frameCols), centerY = (int)(d.cy * frameRows);
            const int width = (int)(d.w * frameCols), height = (int)(d.h * frameRows);
            s_x[t] = centerX - width / 2; s_y[t] = centerY - height / 2; s_w[t] = width; s_h[t] = height;
        }
        s_dead[t] = live ? 0 : 1;
        nc += live;
    }
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nc += __shfl_xor(nc, o, 64);
    if (lane == 0) s_wmin[wv] = nc;
    __syncthreads();
    const int ncand = s_wmin[0] + s_wmin[1] + s_wmin[2] + s_wmin[3];
    __syncthreads();
    // Greedy loop, ONE barrier per kept box.  Thread `tid` owns candidates tid, tid + 256, ... (at most 16): whether they are still alive is a bit mask
    // in a register, their areas are registers, so an iteration is: own first alive candidate at or after `from` -> wave minimum -> LDS (double-buffered
    // by the iteration's parity: a wave can be at most one barrier ahead) -> barrier -> block minimum i = the next kept box -> every thread tests its own
    // later candidates against it.  (The first version kept the flags in LDS and took three barriers per kept box: with ~460 boxes kept per image on
    // the synthetic heads that was 0.83 ms of pure latency per 256-image pass.)
    constexpr int NJ = SD_NMS_MAXDET / 256;
    unsigned alive = 0;
#pragma unroll
    for (int j = 0; j < NJ; j++) { const int t = tid + 256 * j; if (t < ncand && !s_dead[t]) alive |= 1u << j; }
    int nk = 0, from = 0, par = 0;
    while (true) {
        const int j0 = from <= tid ? 0 : (from - tid + 255) >> 8;              // own candidates below `from` are out of play
        const unsigned m = j0 < NJ ? alive & (~0u << j0) : 0u;
        int mine = m ? tid + 256 * (__ffs((int)m) - 1) : (1 << 30);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mine = min(mine, __shfl_xor(mine, o, 64));
        if (lane == 0) s_wmin[4 * par + wv] = mine;
        __syncthreads();
        const int i = min(min(s_wmin[4 * par], s_wmin[4 * par + 1]), min(s_wmin[4 * par + 2], s_wmin[4 * par + 3]));
        par ^= 1;
        if (i >= ncand) break;
        if (nk >= SD_NMS_MAXKEEP) { if (tid == 0) nOut[img] = -1 - nd; return; }       // uniform
        if (tid == 0) s_kept[nk] = i;
        nk++;
        const int rx = s_x[i], ry = s_y[i], width = s_w[i], height = s_h[i];
        const double Ab = (double)width * height;
        unsigned rest = alive & (i < tid ? ~0u : (((i - tid) >> 8) + 1 < NJ ? ~0u << (((i - tid) >> 8) + 1) : 0u));      // own candidates after i
        if ((i & 255) == tid) alive &= ~(1u << (i >> 8));
        while (rest) {
            const int j = __ffs((int)rest) - 1;
            rest &= rest - 1;
            const int t = tid + 256 * j;
            // cv::dnn NMSBoxes: keep t only while overlap(t, kept) <= nms for every kept box; overlap = 1 - jaccardDistance on cv::Rect
            const int tx = s_x[t], ty = s_y[t], tw = s_w[t], th = s_h[t];
            const double Aa = (double)tw * th;
            bool kill;
            if ((Aa + Ab) <= 2.220446049250313e-16) kill = !(1.f <= nmsThreshold);
            else {
                const int x1 = max(tx, rx), y1 = max(ty, ry);
                const int x2 = min(tx + tw, rx + width), y2 = min(ty + th, ry + height);
                const double Aab = (x2 > x1 && y2 > y1) ? (double)(x2 - x1) * (y2 - y1) : 0.0;
                // ov = (float)(1 - (1 - Aab / U)) against the threshold.  The quotient is only needed when it lies within 1e-6 of the threshold -- the
                // roundings of the reference's expression move it by < 1e-7 -- and everywhere else two multiplications decide.
                const double U = Aa + Ab - Aab, thr = (double)nmsThreshold;
                if (Aab < (thr - 1e-6) * U) kill = false;
                else if (Aab > (thr + 1e-6) * U) kill = true;
                else kill = !((float)(1. - (1. - Aab / U)) <= nmsThreshold);
            }
            if (kill) alive &= ~(1u << j);
        }
        from = i + 1;
    }
    if (tid == 0) {
        int n = 0;
        for (int k = 0; k < nk; k++) {
            const int t = s_kept[k];
            const SdDet d = D[(int)(keys[t] & 0x1FFFu)];
            const int c = d.cls;
            if (!(c == 0 || c == 1 || c == 2 || c == 5 || c == 7)) continue;
            if (n >= SD_MAX_BOXES) { n = -1 - nk; break; }
            const double sw = -0.2 * (double)s_w[t], sh = 0.6 * (double)s_h[t];
            double* o = boxesOut + ((size_t)img * SD_MAX_BOXES + n) * 4;
            o[0] = (double)s_x[t] - sw / 2.0; o[1] = (double)s_y[t] - sh / 2.0; o[2] = (double)s_w[t] + sw; o[3] = (double)s_h[t] + sh;
            clsOut[img * SD_MAX_BOXES + n] = c; confOut[img * SD_MAX_BOXES + n] = d.conf;
            n++;
        }
        nOut[img] = n;
    }
}

// yolov3Segment::Segmentation (yolo.cc:34-58) after NMS: rasterise the central half-width of every kept box
// (postprocess, yolo.cc:128-131), dilate with cv::getStructuringElement(MORPH_ELLIPSE, 31x31) and return
// 1 - dilated.  One 16x16 pixel tile per workgroup; the rasterised mask of the tile + 15-px halo lives in LDS;
// span[dy] = half-width of the ellipse row (dx such that columns c-dx .. c+dx are set).
struct SdMaskRects { int n; int x0[SD_MAX_BOXES], y0[SD_MAX_BOXES], x1[SD_MAX_BOXES], y1[SD_MAX_BOXES]; };   // filled region [x0,x1) x [y0,y1)
__global__ void __launch_bounds__(256) k_mask_dilate(SdMaskRects R, int cols, int rows, uint8_t* __restrict__ mask, size_t stride)
{
    __shared__ uint8_t t[46][48];
    __shared__ int span[31];
    const int bx = blockIdx.x * 16, by = blockIdx.y * 16, tid = threadIdx.x;
    if (tid < 31) {
        const int r = 15, c = 15, dy = tid - r;
        const double inv_r2 = 1.0 / ((double)r * r);
        span[tid] = (int)lrint(c * sqrt((double)(r * r - dy * dy) * inv_r2));     // saturate_cast<int>(double) = cvRound
    }
    for (int i = tid; i < 46 * 46; i += 256) {
        const int ty = i / 46, tx = i - ty * 46;
        const int x = bx + tx - 15, y = by + ty - 15;
        uint8_t v = 0;
        if (x >= 0 && x < cols && y >= 0 && y < rows)
            for (int k = 0; k < R.n; k++) v |= (x >= R.x0[k] && x < R.x1[k] && y >= R.y0[k] && y < R.y1[k]);
        t[ty][tx] = v;
    }
    __syncthreads();
    const int lx = tid & 15, ly = tid >> 4;
    const int x = bx + lx, y = by + ly;
    if (x >= cols || y >= rows) return;
    int hit = 0;
    for (int dy = -15; dy <= 15 && !hit; dy++) {
        const int dx = span[dy + 15];
        for (int k = -dx; k <= dx; k++) hit |= t[ly + 15 + dy][lx + 15 + k];
    }
    mask[(size_t)y * stride + x] = (uint8_t)(1 - (hit ? 1 : 0));
}
