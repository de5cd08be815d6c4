// HIP kernels of the ORB extractor for gfx950 (CDNA4, wave64).
//
// Each kernel covers the whole batch (all images, all pyramid levels) in one launch so
// that a batch of stereo frames fills the 256 CUs; per-image work is far too small to do
// so on its own.  Reference loops they replace:
//   k_pyr_level0 / k_pyr_level  ComputePyramid            src/ORBextractor.cc:1107-1132
//   k_fast_cells                cv::FAST per 30-px cell    src/ORBextractor.cc:789-829
//   k_quadtree                  DistributeOctTree          src/ORBextractor.cc:539-763
//   k_orient                    IC_Angle + kp finalise     src/ORBextractor.cc:77-104,837-852,1095-1101
//   k_blur                      GaussianBlur 7x7 sigma 2   src/ORBextractor.cc:1085-1086
//   k_describe                  computeOrbDescriptor       src/ORBextractor.cc:107-147
// Compiled with -ffp-contract=off: the f32 expressions must round exactly like the
// reference's un-fused SSE2 code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sd_plan.h"
#include "sd_trig.h"
#include "../../include/sd_frontend.h"

#define SD_WAVE 64

struct SdDevPlan {               // lives in HBM; every kernel gets a pointer to it
    SdLevel lv[SD_MAX_LEVELS];
    int nlevels;
    int iniTh, minTh;
    int cellTotal;
    int cellListCap;
    int kpCapLevels, kpCap;
    unsigned long long pyrImageBytes, blurImageBytes;
    int umax[16];
    int taps[7];
};

__constant__ __attribute__((aligned(4))) signed char c_pattern[1024] = {
#include "orb_pattern.inc"
};

__device__ __forceinline__ int sd_reflect101(int p, int len)
{
    // |p| excursions are at most 19+3 px and len >= 40, so one fold suffices
    if (p < 0) p = -p;
    if (p >= len) p = 2 * (len - 1) - p;
    return p;
}

// XCD-aware work order for per-image kernels.  Workgroups are dispatched round-robin over the 8 XCDs (id % 8) and every
// XCD has its own L2, so a 1-D grid is decoded as (image group, item, XCD): XCD x processes images x, x + 8, ... and the
// items of one image (neighbouring tiles / cells / keypoints, which share cache lines and halos) meet in ONE L2 instead of
// being fetched by up to eight.  Grid size = items_per_image * round_up(n_images, 8).
__device__ __forceinline__ bool sd_xcd_image_item(unsigned id, int perImage, int nImages, int& img, int& item)
{
    const int xcd = (int)(id & 7u);
    const unsigned k = id >> 3;
    item = (int)(k % (unsigned)perImage);
    img = (int)(k / (unsigned)perImage) * 8 + xcd;
    return img < nImages;
}

// Keypoint slots are laid out level after level with every level's slice starting on a multiple of 8 (sd_plan.h), so the 8
// (k_orient) or 4 (k_describe) slots of a workgroup share one level: level, level geometry and the per-level counts are
// wave-uniform SCALAR loads issued together, not per-lane loads chained behind each other.
__device__ __forceinline__ int sd_slot_level(const SdDevPlan& P, int slot0)
{
    int level = 0;
#pragma unroll
    for (int l = 1; l < SD_MAX_LEVELS; l++)
        if (l < P.nlevels && slot0 >= P.lv[l].kpOffset) level = l;
    return level;
}
// lc = this image's per-level counts; the array carries SD_MAX_LEVELS ints of padding so that all loads are unconditional
__device__ __forceinline__ void sd_level_counts(const int* __restrict__ lc, int nlevels, int level, int& before, int& mine, int& total)
{
    before = 0; mine = 0; total = 0;
#pragma unroll
    for (int l = 0; l < SD_MAX_LEVELS; l++) {
        const int ld = lc[l];
        const int c = l < nlevels ? ld : 0;
        total += c;
        before += l < level ? c : 0;
        mine = l == level ? c : mine;
    }
}

typedef uint32_t __attribute__((aligned(1))) sd_u32_una;
typedef unsigned long long __attribute__((aligned(1))) sd_u64_una;
typedef uint32_t sd_u4v __attribute__((ext_vector_type(4)));
typedef sd_u4v sd_u128_unaligned __attribute__((aligned(1)));       // the hardware takes unaligned dwordx4 accesses

// (pyramid level 0 = padded copy of the input with BORDER_REFLECT_101, ORBextractor.cc:1127-1128: k_pyr_level0_gray / _rgb and
//  their _frame kernels in k_fast.h)

// ------------------------------------------------------------------ pyramid, level >= 1
// cv::resize(INTER_LINEAR) of level-1's interior + REFLECT_101 border in one pass: every padded
// pixel is the resized value at its reflected interior position (no second pass, no dependency
// between threads).  Coefficient tables come from the host plan (sd_plan.h) as 8-byte entries.
// Interior groups: the 4 columns' entries are two 16-B loads, the <= 7 source bytes per row one 8-B load.
__device__ __forceinline__ int sd_lerp_px(int s00, int s01, int s10, int s11, int a0, int a1, int b0, int b1)
{
    // all operands are < 2^16 and products < 2^27: full-rate 24-bit multiplies instead of v_mul_lo_u32
    const int h0 = __mul24(s00, a0) + __mul24(s01, a1);
    const int h1 = __mul24(s10, a0) + __mul24(s11, a1);
    return (((__mul24(b0, h0 >> 4) >> 16) + (__mul24(b1, h1 >> 4) >> 16) + 2) >> 2) & 255;
}

#define SD_PYR_ROWS 4     // padded rows per thread: keeps 4x the loads in flight per wave (the kernel is latency-bound)
__global__ void __launch_bounds__(256) k_pyr_level(uint8_t* __restrict__ pyr, const short4* __restrict__ tabs,
                                                   const SdDevPlan* __restrict__ PP, int level)
{
    const SdDevPlan& P = *PP;
    const SdLevel& g = P.lv[level];
    const SdLevel& s = P.lv[level - 1];
    const int img = blockIdx.z;
    const int gx = blockIdx.x * 64 + threadIdx.x;
    const int X0 = -20 + 4 * gx;
    if (X0 > g.W + SD_EDGE - 1) return;
    const int HP = g.H + 2 * SD_EDGE;
    const int Yb = blockIdx.y * (4 * SD_PYR_ROWS) + threadIdx.y;      // rows Yb, Yb+4, Yb+8, Yb+12
    if (Yb >= HP) return;
    const short4* ct = tabs + g.tabOffset;
    const short4* rt = ct + g.W;
    const uint8_t* sbase = pyr + (size_t)img * P.pyrImageBytes + s.pyrOffset + (size_t)SD_EDGE * s.stride + SD_XOFF;
    uint8_t* dbase = pyr + (size_t)img * P.pyrImageBytes + g.pyrOffset + SD_XOFF + X0;
    const int sstride = (int)s.stride, gstride = (int)g.stride;
    short4 re[SD_PYR_ROWS];
#pragma unroll
    for (int r = 0; r < SD_PYR_ROWS; r++) re[r] = rt[sd_reflect101(min(Yb + 4 * r, HP - 1) - SD_EDGE, g.H)];
    // interior groups whose four source columns (+1) fit one 8-byte window (resize ratios up to 2); wider ratios and the
    // groups on the reflected frame take the per-pixel path
    const bool interior = X0 >= 0 && X0 + 3 < g.W;
    if (interior && ct[X0 + 3].x - ct[X0].x <= 6) {
        const short4 c0 = ct[X0], c1 = ct[X0 + 1], c2 = ct[X0 + 2], c3 = ct[X0 + 3];
        const int sx0 = c0.x;
        const int o1 = 8 * (c1.x - sx0), o2 = 8 * (c2.x - sx0), o3 = 8 * (c3.x - sx0);
        unsigned long long w0[SD_PYR_ROWS], w1[SD_PYR_ROWS];
#pragma unroll
        for (int r = 0; r < SD_PYR_ROWS; r++) {
            const int sy0 = re[r].x;
            const int r0 = min(max(sy0, 0), s.H - 1), r1 = min(max(sy0 + 1, 0), s.H - 1);
            // a column at the right edge has a1 == 0, so reading the byte after it (the frame) is harmless
            w0[r] = *(const sd_u64_una*)(sbase + __mul24(r0, sstride) + sx0);     // rows < 2^12, strides < 2^12: 24-bit multiply
            w1[r] = *(const sd_u64_una*)(sbase + __mul24(r1, sstride) + sx0);
        }
#pragma unroll
        for (int r = 0; r < SD_PYR_ROWS; r++) {
            const int Yp = Yb + 4 * r;
            if (Yp < HP) {
                const int b0 = re[r].y, b1 = re[r].z;
                const unsigned long long a = w0[r], b = w1[r];
                const uint32_t p0 = sd_lerp_px((int)(a & 255), (int)((a >> 8) & 255), (int)(b & 255), (int)((b >> 8) & 255), c0.y, c0.z, b0, b1);
                const uint32_t p1 = sd_lerp_px((int)((a >> o1) & 255), (int)((a >> (o1 + 8)) & 255), (int)((b >> o1) & 255), (int)((b >> (o1 + 8)) & 255), c1.y, c1.z, b0, b1);
                const uint32_t p2 = sd_lerp_px((int)((a >> o2) & 255), (int)((a >> (o2 + 8)) & 255), (int)((b >> o2) & 255), (int)((b >> (o2 + 8)) & 255), c2.y, c2.z, b0, b1);
                const uint32_t p3 = sd_lerp_px((int)((a >> o3) & 255), (int)((a >> (o3 + 8)) & 255), (int)((b >> o3) & 255), (int)((b >> (o3 + 8)) & 255), c3.y, c3.z, b0, b1);
                *(uint32_t*)(dbase + __mul24(Yp, gstride)) = p0 | (p1 << 8) | (p2 << 16) | (p3 << 24);
            }
        }
    } else {
        short4 ce[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int X = X0 + k;
            X = X < -SD_EDGE ? -SD_EDGE : (X > g.W + SD_EDGE - 1 ? g.W + SD_EDGE - 1 : X);
            ce[k] = ct[sd_reflect101(X, g.W)];
        }
#pragma unroll
        for (int r = 0; r < SD_PYR_ROWS; r++) {
            const int Yp = Yb + 4 * r;
            if (Yp < HP) {
                const int sy0 = re[r].x, b0 = re[r].y, b1 = re[r].z;
                const int r0 = min(max(sy0, 0), s.H - 1), r1 = min(max(sy0 + 1, 0), s.H - 1);
                const uint8_t* S0 = sbase + __mul24(r0, sstride);
                const uint8_t* S1 = sbase + __mul24(r1, sstride);
                uint32_t pack = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int sx = ce[k].x, sx1 = min(sx + 1, s.W - 1);
                    pack |= (uint32_t)sd_lerp_px(S0[sx], S0[sx1], S1[sx], S1[sx1], ce[k].y, ce[k].z, b0, b1) << (8 * k);
                }
                *(uint32_t*)(dbase + __mul24(Yp, gstride)) = pack;
            }
        }
    }
}

// A pyramid level through LDS, frame included: a 256-thread workgroup produces TW x TH pixels of the PADDED plane (padded
// coordinate (Xp, Yp) <-> interior pixel (reflect101(Xp - 19), reflect101(Yp - 19)), i.e. resize followed by
// copyMakeBorder(BORDER_REFLECT_101) in one pass, as k_pyr_level does).
//   A  the source rows / columns the tile needs are staged with unaligned 16-byte loads (coalesced);
//   B  horizontal pass once per SOURCE row (each is used by ~1.7 destination rows): h = S[sx]*a0 + S[sx+1]*a1, kept as
//      h >> 4 (< 2^15) in a u16 plane; a thread keeps its four column-table entries for all its rows; groups whose four
//      columns ascend within 6 source bytes (all but the ones on the reflected frame) extract from one 64-bit window;
//   C  vertical pass from the u16 plane, 8 pixels per thread, one 8-byte store.
// Same integer arithmetic as sd_lerp_px.  ~4x fewer memory instructions per pixel than k_pyr_level (no per-thread table
// loads, 16-byte source loads, 8-byte stores) and no latency-bound frame strips.
#define SD_PT_XSHIFT 5
template <int TW, int TH>
__global__ void __launch_bounds__(256) k_pyr_level_tiles(uint8_t* __restrict__ pyr, const short4* __restrict__ tabs,
                                                         const SdDevPlan* __restrict__ PP, int level, int srcRowBytes, int srcRowsMax,
                                                         const int* __restrict__ ext)
{
    const SdDevPlan& P = *PP;
    const SdLevel& g = P.lv[level];
    const SdLevel& s = P.lv[level - 1];
    extern __shared__ __align__(16) unsigned char smem[];
    uint8_t* sS = smem;                                                        // [srcRowsMax][srcRowBytes]
    unsigned short* sH = (unsigned short*)(smem + (size_t)srcRowsMax * srcRowBytes);      // [srcRowsMax][TW]
    const int img = blockIdx.z, tid = threadIdx.x;
    // tile origin: padded column -SD_PT_XSHIFT + bx*TW, so that the 8-byte stores of phase C fall on 8-byte addresses (the
    // padded plane starts SD_XOFF - 19 = 13 bytes into the row; columns < 0 land in the unused left margin)
    const int Xp0 = (int)blockIdx.x * TW - SD_PT_XSHIFT, Yp0 = blockIdx.y * TH;
    const int PW = g.W + 2 * SD_EDGE, HP = g.H + 2 * SD_EDGE;
    const short4* ct = tabs + g.tabOffset;
    const short4* rt = ct + g.W;
    const int sstride = s.stride, gstride = g.stride;
    const uint8_t* sbase = pyr + (size_t)img * P.pyrImageBytes + s.pyrOffset + (size_t)SD_EDGE * s.stride + SD_XOFF;
    // extents of the source region, precomputed on the host (the reflection makes them non-monotone on the frame)
    const int sxA = ext[blockIdx.x], syA = ext[gridDim.x + 2 * blockIdx.y], nrows = ext[gridDim.x + 2 * blockIdx.y + 1];
    // The column-table entries of phase B and the row-table entries of phase C are requested here, ahead of the source
    // pixels, so that the workgroup pays ONE memory round trip instead of three (tables, pixels, tables).
    constexpr int NGB = TW / 4, STEPB = 256 / NGB;
    constexpr int NGC = TW / 8, STEPC = 256 / NGC, NYC = (TH + STEPC - 1) / STEPC;
    const int gqB = tid % NGB;
    short4 c[4];
#pragma unroll
    for (int k = 0; k < 4; k++) c[k] = ct[sd_reflect101(min(max(Xp0 + 4 * gqB + k, 0), PW - 1) - SD_EDGE, g.W)];
    short4 reC[NYC];
#pragma unroll
    for (int i = 0; i < NYC; i++) reC[i] = rt[sd_reflect101(min(Yp0 + tid / NGC + STEPC * i, HP - 1) - SD_EDGE, g.H)];
    // ---- A: 16-byte chunks, two requests in flight per thread
    const int cpr = srcRowBytes >> 4;
    const uint32_t cprInv = 0xFFFFFFFFu / (uint32_t)cpr + 1u;                  // c / cpr == umulhi(c, cprInv) for c * cpr < 2^32
    const int nchunks = nrows * cpr;
    for (int c0 = tid; c0 < nchunks; c0 += 512) {
        const int c1 = c0 + 256;
        const int ra = (int)__umulhi((uint32_t)c0, cprInv), qa = c0 - ra * cpr;
        const int rb = (int)__umulhi((uint32_t)c1, cprInv), qb = c1 - rb * cpr;
        const sd_u4v va = *(const sd_u128_unaligned*)(sbase + __mul24(syA + ra, sstride) + sxA + 16 * qa);
        sd_u4v vb = va;
        if (c1 < nchunks) vb = *(const sd_u128_unaligned*)(sbase + __mul24(syA + rb, sstride) + sxA + 16 * qb);
        *(sd_u4v*)(sS + ra * srcRowBytes + 16 * qa) = va;
        if (c1 < nchunks) *(sd_u4v*)(sS + rb * srcRowBytes + 16 * qb) = vb;
    }
    __syncthreads();
    // ---- B
    {
        constexpr int NG = NGB, STEP = STEPB;
        const int gq = gqB;
        const int o0 = c[0].x - sxA;
        const int d1 = c[1].x - c[0].x, d2 = c[2].x - c[0].x, d3 = c[3].x - c[0].x;
        const bool window = d1 >= 0 && d2 >= d1 && d3 >= d2 && d3 <= 6;          // per lane; false only on the reflected frame
        const int sh = o0 & 3;
        // window path: the 8 source bytes of the group sit in two dwords (lo, hi); pixel k needs bytes d_k and d_k + 1 of them.
        // One v_perm spreads the pair into two u16 lanes, one v_dot2_u32_u16 applies (a0, a1): two instructions per pixel
        // (the byte-shift form needed 64-bit shifts and masks, ~4x as many; the kernel is VALU-bound).
        typedef unsigned short sd_us2 __attribute__((ext_vector_type(2)));
        uint32_t sel[4], wk[4];
        const int dk[4] = {0, d1, d2, d3};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            sel[k] = (uint32_t)dk[k] | 0x0C00u | ((uint32_t)(dk[k] + 1) << 16) | 0x0C000000u;
            wk[k] = ((uint32_t)c[k].y & 0xFFFFu) | ((uint32_t)c[k].z << 16);
        }
        for (int r = tid / NG; r < nrows; r += STEP) {
            const uint8_t* row = sS + r * srcRowBytes;
            uint32_t h[4];
            if (window) {
                const uint32_t* p = (const uint32_t*)(row + (o0 & ~3));
                const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
                const uint32_t lo = __builtin_amdgcn_alignbyte(w1, w0, sh), hi = __builtin_amdgcn_alignbyte(w2, w1, sh);
#pragma unroll
                for (int k = 0; k < 4; k++)
                    h[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(sd_us2, __builtin_amdgcn_perm(hi, lo, sel[k])), __builtin_bit_cast(sd_us2, wk[k]), 0u, false);
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int o = c[k].x - sxA;           // the byte after the last source column has weight a1 == 0
                    h[k] = (uint32_t)(__mul24((int)row[o], c[k].y) + __mul24((int)row[o + 1], c[k].z));
                }
            }
            *(uint2*)(sH + r * TW + 4 * gq) = make_uint2((h[0] >> 4) | ((h[1] >> 4) << 16), (h[2] >> 4) | ((h[3] >> 4) << 16));
        }
    }
    __syncthreads();
    // ---- C
    {
        constexpr int NG = TW / 8, STEP = 256 / NG;
        const int gq = tid % NG;
        if (Xp0 + 8 * gq >= PW) return;                                            // no barrier below
        uint8_t* dbase = pyr + (size_t)img * P.pyrImageBytes + g.pyrOffset + (SD_XOFF - SD_EDGE) + Xp0 + 8 * gq;
#pragma unroll
        for (int i = 0; i < NYC; i++) {
            const int Y = tid / NG + STEP * i;
            const int Yp = Yp0 + Y;
            if (Y >= TH || Yp >= HP) break;
            const short4 re = reC[i];
            const int r0 = min(max((int)re.x, 0), s.H - 1) - syA, r1 = min(max((int)re.x + 1, 0), s.H - 1) - syA;
            const int b0 = re.y, b1 = re.z;
            const uint4 u0 = *(const uint4*)(sH + r0 * TW + 8 * gq), u1 = *(const uint4*)(sH + r1 * TW + 8 * gq);
            const uint32_t hh0[4] = {u0.x, u0.y, u0.z, u0.w}, hh1[4] = {u1.x, u1.y, u1.z, u1.w};
            uint32_t out[2] = {0u, 0u};
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int a0 = (int)((hh0[k >> 1] >> (16 * (k & 1))) & 0xFFFFu), a1 = (int)((hh1[k >> 1] >> (16 * (k & 1))) & 0xFFFFu);
                const uint32_t v = (uint32_t)((((__mul24(b0, a0) >> 16) + (__mul24(b1, a1) >> 16) + 2) >> 2) & 255);
                out[k >> 2] |= v << (8 * (k & 3));
            }
            // a group that runs past the padded width spills into the row's slack / the next row's unused left margin
            *(uint2*)(dbase + __mul24(Yp, gstride)) = make_uint2(out[0], out[1]);
        }
    }
}

// ------------------------------------------------------------------ FAST-9/16 per cell
// One workgroup per cell window (<= 64x64 px, staged in LDS).  Instead of OpenCV's
// threshold-dependent row buffers we use the threshold-free form of the same result:
//   s(p)   = max over the 16 contiguous 9-arcs of min |v - ring| on the darker/brighter side, -1
//            (== cornerScore<16>; p is a corner at threshold T  <=>  s(p) >= T)
//   keep_T = s(p) >= T  and  s(p) > s(q) for the 8 neighbours q inside the scanned window
// and the cell uses T = iniTh if keep_iniTh is non-empty, else minTh (ORBextractor.cc:807-816).
#define SD_TILE_S 72   // LDS row stride of the window tile

__device__ __forceinline__ bool sd_has9(unsigned m)
{
    unsigned x = m | (m << 16);
    unsigned y = x & (x >> 1);
    y &= y >> 2;
    y &= y >> 4;          // bit i: ring bits i..i+7 set
    y &= x >> 8;          // bit i: ring bits i..i+8 set
    return (y & 0xFFFFu) != 0;
}

__device__ __forceinline__ int sd_fast_score(const uint8_t* c, int minTh)
{
    const int S = SD_TILE_S;
    const int v = c[0];
    int d[16];
    d[0] = v - c[3 * S];       d[1] = v - c[3 * S + 1];   d[2] = v - c[2 * S + 2];   d[3] = v - c[S + 3];
    d[4] = v - c[3];           d[5] = v - c[-S + 3];      d[6] = v - c[-2 * S + 2];  d[7] = v - c[-3 * S + 1];
    d[8] = v - c[-3 * S];      d[9] = v - c[-3 * S - 1];  d[10] = v - c[-2 * S - 2]; d[11] = v - c[-S - 3];
    d[12] = v - c[-3];         d[13] = v - c[S - 3];      d[14] = v - c[2 * S - 2];  d[15] = v - c[3 * S - 1];
    unsigned dark = 0, bright = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        dark |= (unsigned)(d[k] > minTh) << k;      // ring pixel darker than v - T
        bright |= (unsigned)(d[k] < -minTh) << k;   // ring pixel brighter than v + T
    }
    if (!sd_has9(dark) && !sd_has9(bright)) return 0;
    // sliding 9-window min / max over the cyclic ring by doubling
    int mn2[16], mx2[16], mn4[16], mx4[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { mn2[k] = min(d[k], d[(k + 1) & 15]); mx2[k] = max(d[k], d[(k + 1) & 15]); }
#pragma unroll
    for (int k = 0; k < 16; k++) { mn4[k] = min(mn2[k], mn2[(k + 2) & 15]); mx4[k] = max(mx2[k], mx2[(k + 2) & 15]); }
    int A = -1000, B = 1000;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        int mn9 = min(min(mn4[k], mn4[(k + 4) & 15]), d[(k + 8) & 15]);
        int mx9 = max(max(mx4[k], mx4[(k + 4) & 15]), d[(k + 8) & 15]);
        A = max(A, mn9);
        B = min(B, mx9);
    }
    return max(A, -B) - 1;
}

__global__ void __launch_bounds__(256) k_fast_cells(const uint8_t* __restrict__ pyr, const SdCell* __restrict__ cells,
                                                    uint32_t* __restrict__ cellList, int* __restrict__ cellCount,
                                                    const SdDevPlan* __restrict__ PP)
{
    const SdDevPlan& P = *PP;
    __shared__ uint8_t tile[70 * SD_TILE_S];
    __shared__ uint8_t keep[64 * 64];       // NMS-surviving score per scanned pixel (0 = none)
    __shared__ int s_any, s_wsum[4];
    const int img = blockIdx.y;
    const SdCell c = cells[blockIdx.x];
    const SdLevel& g = P.lv[c.level];
    const int ww = c.x1 - c.x0, wh = c.y1 - c.y0;
    const int sw = ww - 6, sh = wh - 6;                    // scanned area
    const int tid = threadIdx.x;
    if (tid == 0) s_any = 0;
    if (sw <= 0 || sh <= 0) { if (tid == 0) cellCount[(size_t)img * P.cellTotal + blockIdx.x] = 0; return; }
    const uint8_t* src = pyr + (size_t)img * P.pyrImageBytes + g.pyrOffset + (size_t)(SD_EDGE + c.y0) * g.stride +
                         SD_XOFF + c.x0;
    for (int i = tid; i < ww * wh; i += 256) {
        int y = i / ww, x = i - y * ww;
        tile[y * SD_TILE_S + x] = src[(size_t)y * g.stride + x];
    }
    __syncthreads();
    const int npix = sw * sh;
    // scores (0 unless corner at minTh) into keep[] first
    for (int i = tid; i < npix; i += 256) {
        int y = i / sw, x = i - y * sw;
        int s = sd_fast_score(&tile[(y + 3) * SD_TILE_S + x + 3], P.minTh);
        keep[i] = (uint8_t)(s >= P.minTh ? s : 0);
    }
    __syncthreads();
    // 3x3 NMS (strictly greater than all 8 neighbours; outside the scanned area counts as 0)
    uint8_t mine[16];
    int any = 0;
    {
        int k = 0;
        for (int i = tid; i < npix; i += 256, k++) {
            int y = i / sw, x = i - y * sw;
            int s = keep[i];
            int ok = s > 0;
            if (ok) {
#pragma unroll
                for (int dy = -1; dy <= 1; dy++)
#pragma unroll
                    for (int dx = -1; dx <= 1; dx++) {
                        if (dx == 0 && dy == 0) continue;
                        int xx = x + dx, yy = y + dy;
                        int q = (xx >= 0 && xx < sw && yy >= 0 && yy < sh) ? keep[yy * sw + xx] : 0;
                        ok &= (s > q);
                    }
            }
            mine[k] = (uint8_t)(ok ? s : 0);
            any |= (ok && s >= P.iniTh);
        }
    }
    if (any) s_any = 1;
    __syncthreads();
    {
        int k = 0;
        for (int i = tid; i < npix; i += 256, k++) keep[i] = mine[k];
    }
    __syncthreads();
    const int T = s_any ? P.iniTh : P.minTh;
    // ordered (row-major) compaction: thread t owns the contiguous chunk [t*chunk, (t+1)*chunk)
    const int chunk = (npix + 255) / 256;
    const int beg = tid * chunk, end = min(beg + chunk, npix);
    int cnt = 0;
    for (int i = beg; i < end; i++) cnt += (keep[i] >= T);
    // block exclusive scan of cnt
    const int lane = tid & 63, wv = tid >> 6;
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) s_wsum[wv] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wv; w++) base += s_wsum[w];
    const int total = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
    int pos = base + incl - cnt;
    uint32_t* out = cellList + (size_t)img * P.cellListCap + c.listOffset;
    for (int i = beg; i < end; i++) {
        int s = keep[i];
        if (s >= T) {
            int y = i / sw, x = i - y * sw;
            // coordinates relative to (minBorderX, minBorderY): FAST-local + j*wCell (ORBextractor.cc:822-823)
            uint32_t px = (uint32_t)(x + 3 + c.jw), py = (uint32_t)(y + 3 + c.ih);
            if (pos < c.cap) out[pos] = px | (py << 12) | ((uint32_t)s << 24);
            pos++;
        }
    }
    if (tid == 0) cellCount[(size_t)img * P.cellTotal + blockIdx.x] = min(total, c.cap);
}

// ------------------------------------------------------------------ quadtree distribution
// One workgroup per (image, level).  The std::list of the reference is an array in list order;
// a pass divides a set of nodes "in processing order" (list order for the plain passes, descending
// (size, creation) for the sorted passes of ORBextractor.cc:673-738), children are created in that
// order and end up reversed at the head of the list, exactly like repeated push_front.  The early
// `break` of the sorted pass is a prefix-sum cut.  Node arrays live in LDS (list capacity L = quota+3,
// child slots 4L); the candidates (packed x,y,score) and their node ids live in REGISTERS, CPT per
// thread (8 or 32; a global-memory path covers levels with more than 8192 candidates), so a pass
// touches no HBM/L2 at all.
__device__ __forceinline__ int sd_block_excl_scan(int* a, int n, int* wsum /*[>=4]*/)
{
    // in-place exclusive scan of a[0..n) by 256 threads; returns the total
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int chunk = (n + 255) / 256;
    const int beg = min(tid * chunk, n), end = min(beg + chunk, n);
    int sum = 0;
    for (int i = beg; i < end; i++) sum += a[i];
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    __syncthreads();
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wv; w++) base += wsum[w];
    const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    int run = base + incl - sum;
    for (int i = beg; i < end; i++) { int v = a[i]; a[i] = run; run += v; }
    __syncthreads();
    return total;
}

__device__ __forceinline__ int sd_quadrant(short4 r, int x, int y)
{
    const int midx = r.x + ((r.y - r.x + 1) >> 1);     // UL.x + ceil((UR.x-UL.x)/2)
    const int midy = r.z + ((r.w - r.z + 1) >> 1);
    return (x < midx ? 0 : 1) + (y < midy ? 0 : 2);   // n1,n2,n3,n4 -> 0,1,2,3
}

struct SdQtLds {
    unsigned long long* keys;      // [sortP]
    short4* rectA; short4* rectB;  // [L]  x = x0, y = x1, z = y0, w = y1
    int* cntA; int* cntB;          // [L]
    int* tmp;                      // [L]
    unsigned* best;                // [L]
    int* child; int* gidx;         // [4L]
    short* order; short* procRank; short* keptRank;   // [L]
    int* wsum;                     // [8]
    int* scal;                     // [4]
};

// CPT > 0: candidates c = tid + 256*k, k < CPT, in registers.  CPT == 0: candidates in global memory.
template <int CPT>
__device__ __forceinline__ void sd_qt_body(const SdQtLds S, const SdLevel& g, uint32_t* __restrict__ myCand,
                                           uint16_t* __restrict__ myNode, const int M, const int L, const int sortP,
                                           int* __restrict__ errFlag, int& nOutRef)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int R = CPT > 0 ? CPT : 2;
    uint32_t cv[R];
    unsigned nd2[R / 2];          // node ids (< 65536), two per register: keeps the kernel under 128 VGPRs
#define SD_FOR_CAND(BODY)                                                                      \
    if constexpr (CPT > 0) {                                                                   \
        _Pragma("unroll") for (int k_ = 0; k_ < CPT; k_++) {                                   \
            const int c = tid + 256 * k_;                                                      \
            if (c < M) {                                                                       \
                const uint32_t V = cv[k_];                                                     \
                int ND = (int)((nd2[k_ >> 1] >> (16 * (k_ & 1))) & 0xFFFFu);                   \
                BODY                                                                           \
                nd2[k_ >> 1] = (k_ & 1) ? ((nd2[k_ >> 1] & 0x0000FFFFu) | ((unsigned)ND << 16))  \
                                        : ((nd2[k_ >> 1] & 0xFFFF0000u) | (unsigned)ND);       \
            }                                                                                  \
            if ((k_ & 7) == 7) __builtin_amdgcn_sched_barrier(0);   /* bound the live ranges */ \
        }                                                                                      \
    } else {                                                                                   \
        for (int c = tid; c < M; c += 256) {                                                   \
            const uint32_t V = myCand[c]; int ND = myNode[c]; BODY myNode[c] = (uint16_t)ND;   \
        }                                                                                      \
    }
    if constexpr (CPT > 0) {
#pragma unroll
        for (int k = 0; k < CPT; k++) { const int c = tid + 256 * k; cv[k] = c < M ? myCand[c] : 0u; }
#pragma unroll
        for (int k = 0; k < CPT / 2; k++) nd2[k] = 0;
    }
    // ---- initial nodes (ORBextractor.cc:543-588)
    const int N = g.quota, nIni = g.nIni;
    const float hX = g.hX;
    for (int i = tid; i < nIni; i += 256) S.child[i] = 0;
    __syncthreads();
    SD_FOR_CAND({
        int idx = (int)((float)(V & 0xFFF) / hX);
        idx = min(idx, nIni - 1);
        ND = idx;
        atomicAdd(&S.child[idx], 1);
    })
    __syncthreads();
    for (int i = tid; i < nIni; i += 256) S.tmp[i] = S.child[i] > 0;
    __syncthreads();
    int n = sd_block_excl_scan(S.tmp, nIni, S.wsum);
    for (int i = tid; i < nIni; i += 256) {
        if (S.child[i] > 0) {
            short4 r;
            r.x = (short)(int)(hX * (float)i); r.y = (short)(int)(hX * (float)(i + 1));
            r.z = 0; r.w = (short)(g.maxBY - g.minBY);
            S.rectA[S.tmp[i]] = r; S.cntA[S.tmp[i]] = S.child[i];
        }
    }
    __syncthreads();
    SD_FOR_CAND({ ND = S.tmp[ND]; })
    __syncthreads();

    short4* rc = S.rectA; short4* rn = S.rectB; int* cc = S.cntA; int* cn = S.cntB;
    bool sortedPhase = false;
    for (int iter = 0; iter < 96; iter++) {
        const int prev = n;
        // ---- processing order
        for (int i = tid; i < n; i += 256) { S.tmp[i] = cc[i] > 1; S.procRank[i] = -1; }
        if (sortedPhase)
            for (int i = tid; i < sortP; i += 256)
                S.keys[i] = (i < n && cc[i] > 1) ? (((unsigned long long)(unsigned)cc[i] << 32) | (unsigned)(0xFFFF - i)) : 0ull;
        __syncthreads();
        const int m = sd_block_excl_scan(S.tmp, n, S.wsum);
        if (!sortedPhase) {
            for (int i = tid; i < n; i += 256) if (cc[i] > 1) S.order[S.tmp[i]] = (short)i;
        } else {
            // descending order by rank counting (m <= quota: a few hundred keys; one LDS sweep per key, no barriers)
            for (int i = tid; i < n; i += 256) {
                const unsigned long long k = S.keys[i];
                if (k) {
                    int rank = 0;
                    for (int j = 0; j < n; j++) rank += S.keys[j] > k;
                    S.order[rank] = (short)i;
                }
            }
        }
        for (int i = tid; i < 4 * m; i += 256) S.child[i] = 0;
        __syncthreads();
        for (int i = tid; i < m; i += 256) S.procRank[S.order[i]] = (short)i;
        __syncthreads();
        // ---- child counts of every node in `order`
        SD_FOR_CAND({
            const int t = S.procRank[ND];
            if (t >= 0) atomicAdd(&S.child[4 * t + sd_quadrant(rc[ND], V & 0xFFF, (V >> 12) & 0xFFF)], 1);
        })
        __syncthreads();
        // ---- sorted pass: cut at the first division that reaches N nodes (break at :731-732)
        int mEff = m;
        if (sortedPhase) {
            if (tid == 0) S.scal[0] = m;
            for (int i = tid; i < m; i += 256) {
                const int e = (S.child[4 * i] > 0) + (S.child[4 * i + 1] > 0) + (S.child[4 * i + 2] > 0) + (S.child[4 * i + 3] > 0) - 1;
                S.gidx[i] = e; S.tmp[i] = e;
            }
            __syncthreads();
            sd_block_excl_scan(S.tmp, m, S.wsum);
            for (int i = tid; i < m; i += 256)
                if (n + S.tmp[i] + S.gidx[i] >= N) atomicMin(&S.scal[0], i);
            __syncthreads();
            mEff = min(m, S.scal[0] + 1);
            __syncthreads();
            for (int i = tid + mEff; i < m; i += 256) S.procRank[S.order[i]] = -1;
            __syncthreads();
        }
        // ---- creation index of the non-empty children, in creation order
        for (int i = tid; i < 4 * mEff; i += 256) S.gidx[i] = S.child[i] > 0;
        for (int i = tid; i < n; i += 256) S.tmp[i] = S.procRank[i] < 0;
        __syncthreads();
        const int E = sd_block_excl_scan(S.gidx, 4 * mEff, S.wsum);
        const int K = sd_block_excl_scan(S.tmp, n, S.wsum);
        // ---- new list: children reversed at the head, untouched nodes behind in their old order
        int nExp = 0;
        for (int j = tid; j < 4 * mEff; j += 256) {
            const int cj = S.child[j];
            if (cj > 0) {
                const short4 r = rc[S.order[j >> 2]];
                const int q = j & 3;
                const int midx = r.x + ((r.y - r.x + 1) >> 1), midy = r.z + ((r.w - r.z + 1) >> 1);
                short4 o;
                o.x = (q & 1) ? (short)midx : r.x; o.y = (q & 1) ? r.y : (short)midx;
                o.z = (q & 2) ? (short)midy : r.z; o.w = (q & 2) ? r.w : (short)midy;
                const int np = E - 1 - S.gidx[j];
                rn[np] = o; cn[np] = cj;
                nExp += (cj > 1);
            }
        }
        for (int i = tid; i < n; i += 256)
            if (S.procRank[i] < 0) { const int np = E + S.tmp[i]; rn[np] = rc[i]; cn[np] = cc[i]; S.keptRank[i] = (short)S.tmp[i]; }
        __syncthreads();
        SD_FOR_CAND({
            const int t = S.procRank[ND];
            if (t >= 0) ND = E - 1 - S.gidx[4 * t + sd_quadrant(rc[ND], V & 0xFFF, (V >> 12) & 0xFFF)];
            else ND = E + S.keptRank[ND];
        })
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nExp += __shfl_xor(nExp, o, 64);
        if (lane == 0) S.wsum[4 + wv] = nExp;
        __syncthreads();
        const int nToExpand = S.wsum[4] + S.wsum[5] + S.wsum[6] + S.wsum[7];
        n = E + K;
        { short4* t1 = rc; rc = rn; rn = t1; int* t2 = cc; cc = cn; cn = t2; }
        if (n >= N || n == prev) break;                       // ORBextractor.cc:666-669, 735-736
        if (!sortedPhase && (n + nToExpand * 3) > N) sortedPhase = true;   // :670
        if (n + 4 > L) { if (tid == 0) atomicOr(errFlag, 1); break; }
    }
    // ---- best response per node, first in candidate order on ties (ORBextractor.cc:741-760)
    for (int i = tid; i < n; i += 256) S.best[i] = 0;
    __syncthreads();
    SD_FOR_CAND({ atomicMax(&S.best[ND], ((V >> 24) << 24) | (0xFFFFFFu - (unsigned)c)); })
    __syncthreads();
    nOutRef = n;
#undef SD_FOR_CAND
}

__global__ void __launch_bounds__(256, 4) k_quadtree(const uint32_t* __restrict__ cellList,
                                                  const int* __restrict__ cellCount, const SdCell* __restrict__ cells,
                                                  uint32_t* __restrict__ cand, uint16_t* __restrict__ nodeOf,
                                                  int* __restrict__ lvlCount, int* __restrict__ candCount,
                                                  uint32_t* __restrict__ lvlKp, int* __restrict__ errFlag,
                                                  const SdDevPlan* __restrict__ PP, int L, int sortP)
{
    const SdDevPlan& P = *PP;
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int s_wsum[8];
    __shared__ int s_scal[4];
    // longest work first: all images' level 0 (the most candidates), then level 1, ... so that the short high levels fill
    // the slots the level-0 workgroups leave, instead of every image's level 0 heading a round of its own
    const int level = blockIdx.y, img = blockIdx.x;
    const SdLevel& g = P.lv[level];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    SdQtLds S;
    unsigned char* p = smem;
    S.keys = (unsigned long long*)p; p += (size_t)sortP * 8;
    S.rectA = (short4*)p; p += (size_t)L * 8;
    S.rectB = (short4*)p; p += (size_t)L * 8;
    S.cntA = (int*)p; p += (size_t)L * 4;
    S.cntB = (int*)p; p += (size_t)L * 4;
    S.tmp = (int*)p; p += (size_t)L * 4;
    S.best = (unsigned*)p; p += (size_t)L * 4;
    S.child = (int*)p; p += (size_t)L * 16;
    S.gidx = (int*)p; p += (size_t)L * 16;
    S.order = (short*)p; p += (size_t)L * 2;
    S.procRank = (short*)p; p += (size_t)L * 2;
    S.keptRank = (short*)p; p += (size_t)L * 2;
    S.wsum = s_wsum; S.scal = s_scal;

    uint32_t* myCand = cand + (size_t)img * P.cellListCap + g.candOffset;
    uint16_t* myNode = nodeOf + (size_t)img * P.cellListCap + g.candOffset;
    const int* myCellCount = cellCount + (size_t)img * P.cellTotal + g.cell0;
    const uint32_t* myList = cellList + (size_t)img * P.cellListCap;

    // ---- compact the per-cell candidate lists in cell order (row-major cells, row-major pixels)
    // One thread per OUTPUT slot: its cell is found by bisection over the scanned counts in LDS, so all reads of the cell lists
    // are independent and in flight together (a wave-per-cell loop chained ~120 dependent count -> list round trips per wave and
    // was half of this kernel's time on level 0).
    for (int i = tid; i < g.nCells; i += 256) { S.tmp[i] = myCellCount[i]; S.gidx[i] = cells[g.cell0 + i].listOffset; }
    __syncthreads();
    const int M = sd_block_excl_scan(S.tmp, g.nCells, S.wsum);
    for (int j = tid; j < M; j += 256) {
        int lo = 0, hi = g.nCells - 1;                    // largest cell whose first slot is <= j (empty cells share a slot with their successor)
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (S.tmp[mid] <= j) lo = mid; else hi = mid - 1;
        }
        myCand[j] = myList[S.gidx[lo] + (j - S.tmp[lo])];
    }
    __syncthreads();     // workgroup-scope: myCand is re-read below by other threads of this workgroup
    if (tid == 0) candCount[(size_t)img * P.nlevels + level] = M;

    int n = 0;
    if (M <= 256 * 8) sd_qt_body<8>(S, g, myCand, myNode, M, L, sortP, errFlag, n);
    else if (M <= 256 * 32) sd_qt_body<32>(S, g, myCand, myNode, M, L, sortP, errFlag, n);
    else sd_qt_body<0>(S, g, myCand, myNode, M, L, sortP, errFlag, n);

    const int nOut = min(n, g.kpCap);
    if (n > g.kpCap && tid == 0) atomicOr(errFlag, 2);
    uint32_t* out = lvlKp + (size_t)img * P.kpCapLevels + g.kpOffset;
    for (int i = tid; i < nOut; i += 256) {
        const unsigned c = 0xFFFFFFu - (S.best[i] & 0xFFFFFFu);
        out[i] = myCand[c];
    }
    if (tid == 0) lvlCount[(size_t)img * P.nlevels + level] = nOut;
}

// ------------------------------------------------------------------ orientation + final keypoint record
// Half a wave (32 lanes = the 31 columns of the patch) per keypoint.
__device__ __forceinline__ float sd_fast_atan2(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
    const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
    const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
    const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)2.2204460492503131e-16);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)2.2204460492503131e-16);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

__global__ void __launch_bounds__(256) k_orient(const uint8_t* __restrict__ pyr, const uint32_t* __restrict__ lvlKp,
                                                const int* __restrict__ lvlCount, sd_keypoint* __restrict__ kpOut,
                                                float2* __restrict__ rot, int* __restrict__ count, const SdDevPlan* __restrict__ PP,
                                                int nImages, int groupsPerImage)
{
    const SdDevPlan& P = *PP;
    int img, grp;
    if (!sd_xcd_image_item(blockIdx.x, groupsPerImage, nImages, img, grp)) return;
    const int slot = grp * 8 + (threadIdx.x >> 5);
    const int l32 = threadIdx.x & 31;
    if (slot >= P.kpCapLevels) return;
    const int level = sd_slot_level(P, grp * 8);
    const SdLevel& g = P.lv[level];
    const int idx = slot - g.kpOffset;
    int before, mine, tot;
    sd_level_counts(lvlCount + (size_t)img * P.nlevels, P.nlevels, level, before, mine, tot);
    if (slot == 0 && l32 == 0) count[img] = tot;
    if (idx >= mine) return;
    const uint32_t v = lvlKp[(size_t)img * P.kpCapLevels + slot];
    const int px = (int)(v & 0xFFF) + g.minBX, py = (int)((v >> 12) & 0xFFF) + g.minBY;
    // Buffer addressing: wave-uniform descriptor of this image's level plane + one 32-bit per-lane offset (patch row 0) + a SCALAR
    // row offset per load, so the 31 row loads need no vector address arithmetic (per-lane 64-bit pointers cost a chained 64-bit
    // VALU add per load, a quarter of this VALU-bound kernel's instructions).
    const __amdgpu_buffer_rsrc_t plane = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(pyr + (size_t)img * P.pyrImageBytes + g.pyrOffset + SD_XOFF), 0, 0x7FFFFFFF, 0x00020000);
    const int u = l32 - SD_HALF_PATCH;
    const int off0 = __mul24(SD_EDGE + py - SD_HALF_PATCH, g.stride) + px + u;         // >= 0: py >= 16 - 3, EDGE = 19
    int m10 = 0, m01 = 0;
    if (l32 < 31) {
        // umax of ORBextractor.cc:452-470 depends only on HALF_PATCH_SIZE = 15 (checked against the plan on the
        // host); a compile-time table lets all 31 row loads be issued back to back instead of one per L2 round trip.
        constexpr int kUmax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
        const int au = u < 0 ? -u : u;
        int vals[31];
#pragma unroll
        for (int r = 0; r < 31; r++) vals[r] = __builtin_amdgcn_raw_buffer_load_b8(plane, off0, r * g.stride, 0);
#pragma unroll
        for (int r = 0; r < 31; r++) {
            const int vv = r - SD_HALF_PATCH;
            const int val = au <= kUmax[vv < 0 ? -vv : vv] ? vals[r] : 0;
            m10 += val;                              // u is the same for every row of this lane: one multiply after the loop
            m01 += __mul24(vv, val);                 // |vv| <= 15, val <= 255: full-rate 24-bit multiply
        }
        m10 = __mul24(u, m10);
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { m10 += __shfl_xor(m10, o, 64); m01 += __shfl_xor(m01, o, 64); }
    if (l32 == 0) {
        const float angle = sd_fast_atan2((float)m01, (float)m10);
        sd_keypoint k;
        k.x = (float)px * g.scale;      // level 0: scale == 1.0f exactly (ORBextractor.cc:1095-1101)
        k.y = (float)py * g.scale;
        k.size = g.sizeF; k.angle = angle; k.response = (float)(v >> 24);
        k.octave = level; k.class_id = -1;
        kpOut[(size_t)img * P.kpCap + before + idx] = k;
        const float factorPI = (float)(M_PI / 180.f);
        const float ang = angle * factorPI;
        // a,b := correctly rounded f32 of cos/sin of the f32 angle (oracle spec Q3)
        const sd_cs cs = sd_cos_sin_f32((double)ang);
        rot[(size_t)img * P.kpCapLevels + slot] = make_float2(cs.c, cs.s);
    }
}

// ------------------------------------------------------------------ Gaussian blur 7x7 (fixed point 8.8 taps)
// result = (sum_ij k_i k_j p_ij + 0x8000) >> 16; both passes exact, REFLECT_101 comes for free from the
// pyramid's own 19-px border.  Tile 64x16 per 256-thread block, staged through LDS.
__global__ void __launch_bounds__(256) k_blur(const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur, const SdDevPlan* __restrict__ PP)
{
    const SdDevPlan& P = *PP;
    __shared__ uint8_t t_in[22][72];
    __shared__ uint16_t t_h[22][64];
    const int zi = blockIdx.z;
    const int img = zi / P.nlevels, level = zi - img * P.nlevels;
    const SdLevel& g = P.lv[level];
    const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 16;
    if (x0 >= g.W || y0 >= g.H) return;
    const int tid = threadIdx.x;
    const uint8_t* src = pyr + (size_t)img * P.pyrImageBytes + g.pyrOffset + (size_t)SD_EDGE * g.stride + SD_XOFF;
    for (int i = tid; i < 22 * 70; i += 256) {
        int r = i / 70, c = i - r * 70;
        int y = min(y0 + r - 3, g.H + 2), x = min(x0 + c - 3, g.W + 2);
        t_in[r][c] = src[(ptrdiff_t)y * g.stride + x];
    }
    __syncthreads();
    for (int i = tid; i < 22 * 64; i += 256) {
        int r = i >> 6, c = i & 63;
        int s = 0;
#pragma unroll
        for (int k = 0; k < 7; k++) s += P.taps[k] * t_in[r][c + k];
        t_h[r][c] = (uint16_t)s;
    }
    __syncthreads();
    uint8_t* dst = blur + (size_t)img * P.blurImageBytes + g.blurOffset;
    for (int i = tid; i < 16 * 64; i += 256) {
        int r = i >> 6, c = i & 63;
        int x = x0 + c, y = y0 + r;
        if (x < g.W && y < g.H) {
            unsigned s = 0;
#pragma unroll
            for (int k = 0; k < 7; k++) s += (unsigned)P.taps[k] * t_h[r + k][c];
            unsigned v = (s + 0x8000u) >> 16;
            dst[(size_t)y * g.blurStride + x] = (uint8_t)(v > 255 ? 255 : v);
        }
    }
}

// ------------------------------------------------------------------ steered rBRIEF
// One wave per keypoint.  The 37x37 neighbourhood of the blurred plane (pattern radius <= 18.4 px after
// rotation) is staged in LDS with dword loads, 10 per row; the 512 taps are then LDS byte reads.  Lane i
// evaluates pairs i, i+64, i+128, i+192; the four 64-bit wave ballots ARE the descriptor (bit k of the
// descriptor = pair k, LSB first within each byte).
#define SD_DP_W 40    // staged bytes per patch row
#define SD_DP_R 18    // patch radius
typedef uint32_t __attribute__((aligned(1))) sd_u32_ua;
#define SD_DP_KPW 2   // keypoints per wave: their loads are in flight together (the kernel is bound by memory round trips x occupancy)
__global__ void __launch_bounds__(256) k_describe(const uint8_t* __restrict__ blur, const uint32_t* __restrict__ lvlKp,
                                                  const int* __restrict__ lvlCount, const float2* __restrict__ rot,
                                                  uint8_t* __restrict__ descOut, const SdDevPlan* __restrict__ PP, int nImages,
                                                  int groupsPerImage)
{
    const SdDevPlan& P = *PP;
    __shared__ __align__(16) uint8_t patch[4][SD_DP_KPW][37 * SD_DP_W];
    int img, grp;
    if (!sd_xcd_image_item(blockIdx.x, groupsPerImage, nImages, img, grp)) return;
    const int wv = threadIdx.x >> 6;
    const int slot0 = grp * (4 * SD_DP_KPW) + wv * SD_DP_KPW;      // the 8 slots of a workgroup share a level (slices start on multiples of 8)
    const int lane = threadIdx.x & 63;
    if (slot0 >= P.kpCapLevels) return;
    const int level = sd_slot_level(P, grp * (4 * SD_DP_KPW));
    const SdLevel& g = P.lv[level];
    const int idx0 = slot0 - g.kpOffset;
    int before, mine, tot;
    sd_level_counts(lvlCount + (size_t)img * P.nlevels, P.nlevels, level, before, mine, tot);
    if (idx0 >= mine) return;
    const int nk = min(SD_DP_KPW, mine - idx0);                       // keypoints of this wave (wave-uniform)
    const int step = g.blurStride;
    uint32_t v[SD_DP_KPW];
    float2 ab[SD_DP_KPW];
#pragma unroll
    for (int q = 0; q < SD_DP_KPW; q++) {
        const size_t o = (size_t)img * P.kpCapLevels + slot0 + (q < nk ? q : 0);
        v[q] = lvlKp[o]; ab[q] = rot[o];
    }
    uint32_t pqs[4];                                                       // requested before the patches: one round trip for both
#pragma unroll
    for (int r = 0; r < 4; r++) pqs[r] = *(const uint32_t*)(c_pattern + 4 * (lane + 64 * r));       // one dword = (x0, y0, x1, y1) as int8
    // Patch fetch: lane -> (row, dword) of the 37 x 10-dword patch is the same for every keypoint, so the per-lane byte offsets are
    // computed once (incrementally, no division per load) and a keypoint only adds its wave-uniform origin as the SCALAR offset of
    // buffer loads: no vector address arithmetic per load (the kernel is bound by vector instruction issue).
    const __amdgpu_buffer_rsrc_t plane = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(blur + (size_t)img * P.blurImageBytes + g.blurOffset), 0, 0x7FFFFFFF, 0x00020000);
    int offs[6];
    {
        int r = lane / 10, c = lane - r * 10;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            offs[k] = __mul24(r, step) + 4 * c;
            r += 6; c += 4;                                   // + 64 = 6 rows of 10 dwords + 4
            if (c >= 10) { c -= 10; r += 1; }
        }
    }
    uint32_t pix[SD_DP_KPW][6];
#pragma unroll
    for (int q = 0; q < SD_DP_KPW; q++) {
        const uint32_t vq = (uint32_t)__builtin_amdgcn_readfirstlane((int)v[q]);      // the same in every lane: make it scalar
        const int px = (int)(vq & 0xFFF) + g.minBX, py = (int)((vq >> 12) & 0xFFF) + g.minBY;
        const int origin = (py - SD_DP_R) * step + (px - SD_DP_R);                    // >= 0: the patch lies inside the blurred plane
#pragma unroll
        for (int k = 0; k < 6; k++) {
            pix[q][k] = 0;
            if (lane + 64 * k < 370) pix[q][k] = __builtin_amdgcn_raw_buffer_load_b32(plane, offs[k], origin, 0);
        }
    }
#pragma unroll
    for (int q = 0; q < SD_DP_KPW; q++) {
        uint32_t* pw = (uint32_t*)patch[wv][q];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const int i = lane + 64 * k;
            if (i < 370) pw[i] = pix[q][k];
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): this wave's LDS writes are done
#pragma unroll
    for (int q = 0; q < SD_DP_KPW; q++) {
        if (q >= nk) break;
        const float a = ab[q].x, b = ab[q].y;
        const uint8_t* center = patch[wv][q] + SD_DP_R * SD_DP_W + SD_DP_R;
        unsigned long long words[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t pq = pqs[r];
            const float x0 = (float)(signed char)(pq & 255u), y0 = (float)(signed char)((pq >> 8) & 255u);
            const float x1 = (float)(signed char)((pq >> 16) & 255u), y1 = (float)(signed char)(pq >> 24);
            const int iy0 = __float2int_rn(x0 * b + y0 * a), ix0 = __float2int_rn(x0 * a - y0 * b);
            const int iy1 = __float2int_rn(x1 * b + y1 * a), ix1 = __float2int_rn(x1 * a - y1 * b);
            const int t0 = center[__mul24(iy0, SD_DP_W) + ix0];      // v_mad_i32_i24: a plain `iy * W` is a quarter-rate v_mul_lo_u32 here
            const int t1 = center[__mul24(iy1, SD_DP_W) + ix1];
            words[r] = __ballot(t0 < t1);
        }
        if (lane < 4) {
            unsigned long long w = lane == 0 ? words[0] : lane == 1 ? words[1] : lane == 2 ? words[2] : words[3];
            *(unsigned long long*)(descOut + ((size_t)img * P.kpCap + before + idx0 + q) * 32 + 8 * lane) = w;
        }
    }
}
