// The detector's convolutions with every f32 operand carried as THREE bf16 limbs (mode SD_YOLO_F32X3): x = hi + mid + lo exactly (3 x 8
// significant bits = the 24 of an f32, each limb the round-to-nearest bf16 of what the limbs before it left), and a product a * b evaluated as
// the six limb products of weight >= 2^-16 -- hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi -- each of them EXACT in f32 (8 x 8 bits), accumulated
// in f32 by v_mfma_f32_32x32x16_bf16.  What is dropped (mid*lo, lo*mid, lo*lo) is <= 2^-23 of |a b|, the size of the rounding an f32 multiply
// applies to the product itself, so the mode is an f32 evaluation of the network in the sense the f32 and f32w modes are (different rounding of
// the same sums) and is held to the same layer tolerance and box-set tests -- while the matrix cores run 16 x the f32 MFMA's rate: six bf16
// MFMAs of 32 cycles replace eight f32 MFMAs of 64 cycles per 32 x 32 x 16 block, 2.67 x fewer MFMA cycles.
//   host (sd_yolo_load_darknet_weights)   weights split once: [coutPad][taps][cin / 4][3 limbs][4] bf16 (a 4-channel piece's limbs are adjacent)
//   k_conv_b3                              k_conv_f32's loop (activations stay f32 in HBM): the staging pass splits the activations it moves
//                                          (v_cvt_pk_bf16_f32 + an exact subtraction per limb), LDS holds three limb planes per operand
// Activations, bias, leaky ReLU, shortcut and everything after the accumulator are f32 as in k_conv_f32.
#pragma once
#include "k_yolo32.h"

typedef __bf16 sd_b8 __attribute__((ext_vector_type(8)));
typedef __bf16 sd_b2 __attribute__((ext_vector_type(2)));
typedef float sd_f2v __attribute__((ext_vector_type(2)));
typedef unsigned int sd_u4w __attribute__((ext_vector_type(4)));

// two floats -> their three bf16 limbs, packed (element 0 in the low half)
__device__ __forceinline__ void sd_split3(float x0, float x1, uint32_t& hi, uint32_t& mid, uint32_t& lo)
{
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(sd_f2v{x0, x1}, sd_b2));
    const float r0 = x0 - __builtin_bit_cast(float, hi << 16), r1 = x1 - __builtin_bit_cast(float, hi & 0xffff0000u);       // exact
    mid = __builtin_bit_cast(uint32_t, __builtin_convertvector(sd_f2v{r0, r1}, sd_b2));
    const float q0 = r0 - __builtin_bit_cast(float, mid << 16), q1 = r1 - __builtin_bit_cast(float, mid & 0xffff0000u);     // exact, <= 8 bits left
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(sd_f2v{q0, q1}, sd_b2));
}

#define SD_B3_ROWB 48                          // LDS row: 16 bf16 (one MFMA K chunk) + 16 bytes: 16 consecutive rows cover the 64 banks once
#define SD_B3_LDS(WM) (2 * 3 * (256 / (WM)) * SD_B3_ROWB)   // two stages x three limb planes x 128 | 256 pixel rows

// 128 filters x 128 pixels per 4-wave workgroup (wave 64 x 64 = 2 x 2 MFMA tiles), K in steps of 16 channels of one filter tap = ONE bf16 MFMA
// K chunk: 24 MFMAs (768 cycles) per wave and step.  Only the ACTIVATIONS go through LDS (f32 from HBM, split by the staging pass, three limb
// planes, two stages = 36 KB); the weights were split and laid out in FRAGMENT order on the host -- [filter tile][K step][wave row][limb][m]
// [lane] x 16 bytes, the exact register image of an MFMA A operand -- and every wave requests its own six fragments per step straight from
// L2 two steps ahead.  (With the weights staged through LDS like the activations the kernel was bound by the LDS itself: 72 KB moved per
// workgroup and step against 768 MFMA cycles; measured 43 % of the time with the matrix pipe busy.)
template <int WM>      // 2: 128 filters x 128 pixels; 1: 64 x 256 (the 64-filter layers), every wave the same 64 filters
__global__ void __launch_bounds__(256, 2) k_conv_b3(SdConvArgsF A, const uint4* __restrict__ wgt3)
{
    constexpr int NT = 256, WN = 4 / WM, BM = 64 * WM, BN = 64 * WN, BK = 16, CPR = 4;
    constexpr int XC = BN * CPR / NT;                              // 2 four-channel pieces per thread and step
    constexpr int PLANE = BN * SD_B3_ROWB, STAGE = 3 * PLANE;
    extern __shared__ __align__(16) unsigned char smemb[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r32 = lane & 31, h = lane >> 5;
    const int wm = wv % WM, wn = wv / WM;
    const int perXcd = (A.tilesX + 7) >> 3;
    const int slot = blockIdx.x >> 3, perGroup = perXcd * A.groupY;
    const int grp = slot / perGroup, rg = slot - grp * perGroup;
    const int tx = (blockIdx.x & 7) * perXcd + rg / A.groupY, ty = grp * A.groupY + rg % A.groupY;
    if (tx >= A.tilesX) return;
    const int pix0 = tx * BN, co0 = ty * BM;
    const int npix = A.N * A.Ho * A.Wo;
    int pyi[XC], pxi[XC];
    size_t pbase[XC];
    bool pok[XC];
#pragma unroll
    for (int i = 0; i < XC; i++) {
        const int chunk = tid + NT * i;
        const int p = pix0 + chunk / CPR;
        pok[i] = p < npix;
        const int pp = pok[i] ? p : 0;
        const int n = pp / (A.Ho * A.Wo), r = pp - n * (A.Ho * A.Wo);
        const int yo = r / A.Wo, xo = r - yo * A.Wo;
        pyi[i] = yo * A.stride - A.pad; pxi[i] = xo * A.stride - A.pad;
        pbase[i] = (size_t)n * A.H * A.W;
    }
    const int taps = A.ksize * A.ksize;
    const int ksteps = taps * (A.cin / BK);
    sd_f16v acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;
    sd_f4 xr[2][XC];
    sd_b8 fa[3][3][2];                                   // [register set][limb][m]: the weights' fragments of steps s, s + 1, s + 2
    const uint4* aptr = wgt3 + ((size_t)ty * ksteps * WM + wm) * (6 * 64) + lane;      // this wave's six fragments of a step; the next step's are WM x 6 KB on
    const float* xptr[XC];
    int xinc[XC];
    int c0 = 0, kh = 0, kw = 0;
    auto retap = [&]() {
#pragma unroll
        for (int i = 0; i < XC; i++) {
            const int chunk = tid + NT * i;
            const int yi = pyi[i] + kh, xi = pxi[i] + kw;
            const bool ok = pok[i] && yi >= 0 && yi < A.H && xi >= 0 && xi < A.W;
            const float* p = A.in + ((ptrdiff_t)pbase[i] + (ptrdiff_t)yi * A.W + xi) * A.cinStride + 4 * (chunk % CPR);
            xptr[i] = ok ? p : A.zero;
            xinc[i] = ok ? BK : 0;
        }
    };
    retap();
    auto fetchW = [&](const int set) {
#pragma unroll
        for (int l = 0; l < 3; l++)
#pragma unroll
            for (int m = 0; m < 2; m++) fa[set][l][m] = __builtin_bit_cast(sd_b8, aptr[(2 * l + m) * 64]);
        aptr += WM * 6 * 64;
    };
    auto fetchX = [&](const int set) {
#pragma unroll
        for (int i = 0; i < XC; i++) { xr[set][i] = *(const sd_f4*)xptr[i]; xptr[i] += xinc[i]; }
        c0 += BK;
        if (c0 == A.cin) { c0 = 0; kw++; if (kw == A.ksize) { kw = 0; kh++; } retap(); }
    };
    auto store = [&](int buf, const int set) {
        unsigned char* sX = smemb + buf * STAGE;
#pragma unroll
        for (int i = 0; i < XC; i++) {
            const int chunk = tid + NT * i;
            uint2 l3[3];
            sd_split3(xr[set][i][0], xr[set][i][1], l3[0].x, l3[1].x, l3[2].x);
            sd_split3(xr[set][i][2], xr[set][i][3], l3[0].y, l3[1].y, l3[2].y);
#pragma unroll
            for (int l = 0; l < 3; l++) *(uint2*)(sX + l * PLANE + (chunk / CPR) * SD_B3_ROWB + 8 * (chunk % CPR)) = l3[l];
        }
    };
    // activation fragments: lane (r32, h) holds elements k = 8 h .. 8 h + 7 of pixel row r32 of its tile: one 16-byte read per tile and limb
    const int boff = (64 * wn + r32) * SD_B3_ROWB + 16 * h;
    sd_b8 fb[3][2];
    auto frags = [&](int buf, const int l) {
        const unsigned char* base = smemb + buf * STAGE + l * PLANE;
#pragma unroll
        for (int n = 0; n < 2; n++) fb[l][n] = *(const sd_b8*)(base + boff + 32 * n * SD_B3_ROWB);
    };
    auto mfmas = [&](const int set, const int la, const int lb) {       // the four tiles between two products on the same accumulator
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int n = 0; n < 2; n++)
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][la][m], fb[lb][n], acc[m][n], 0, 0, 0);
    };
    // step s computes with weight set s % 3 and activation stage s & 1; the activations of step s + 1 are split and written during step s
    // (from the f32 registers requested during step s - 1), the weights of step s + 2 are requested during step s
    fetchX(0);
    fetchW(0);
    store(0, 0);
    fetchW(1);
    fetchX(1);                                           // unconditional like the weights' (a step past the last tap reads a valid pixel or the zero page)
    fetchX(0);
    __syncthreads();
    auto step = [&](const int ks, const int par, const int ws) {       // par = ks & 1, ws = ks % 3: literals at the call sites
        frags(par, 0);
        frags(par, 1);
        fetchW((ws + 2) % 3);                            // unconditional and pinned, see k_conv3x3_b3
        __builtin_amdgcn_sched_barrier(0);
        mfmas(ws, 0, 0);
        frags(par, 2);
        mfmas(ws, 0, 1);
        mfmas(ws, 1, 0);
        if (ks + 1 < ksteps) store(par ^ 1, par ^ 1);
        mfmas(ws, 1, 1);
        fetchX(par ^ 1);
        mfmas(ws, 0, 2);
        mfmas(ws, 2, 0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
    };
    int ks = 0;
    for (; ks + 5 < ksteps; ks += 6) { step(ks, 0, 0); step(ks + 1, 1, 1); step(ks + 2, 0, 2); step(ks + 3, 1, 0); step(ks + 4, 0, 1); step(ks + 5, 1, 2); }
    // ksteps is a multiple of 3 or of 2 (taps 9 or 1 times cin / 16 with cin a multiple of 32): the tail walks the remaining steps with the same literals
    if (ks < ksteps) { step(ks, 0, 0); ks++; }
    if (ks < ksteps) { step(ks, 1, 1); ks++; }
    if (ks < ksteps) { step(ks, 0, 2); ks++; }
    if (ks < ksteps) { step(ks, 1, 0); ks++; }
    if (ks < ksteps) { step(ks, 0, 1); ks++; }
    // ---- epilogue (k_conv_f32's for the 128-filter tiles): D column = pixel (lane & 31), rows = filters (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    sd_f4 rr[2][2][4];
#pragma unroll
    for (int n = 0; n < 2; n++) {
        const int p = pix0 + 64 * wn + 32 * n + r32;
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int co = co0 + 64 * wm + 32 * m + 8 * g + 4 * h;
                rr[n][m][g] = sd_f4{0.f, 0.f, 0.f, 0.f};
                if (A.res && p < npix && co < A.cout) rr[n][m][g] = *(const sd_f4*)(A.res + (size_t)p * A.resStride + co);
            }
    }
#pragma unroll
    for (int n = 0; n < 2; n++) {
        const int p = pix0 + 64 * wn + 32 * n + r32;
        if (p >= npix) continue;
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int co = co0 + 64 * wm + 32 * m + 8 * g + 4 * h;
                if (co >= A.cout) continue;
                sd_f4 v;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float x = acc[m][n][4 * g + e] + A.bias[co + e];
                    if (A.leaky) x = x > 0.f ? x : 0.1f * x;
                    v[e] = x + rr[n][m][g][e];
                }
                float* dst = A.out + (size_t)p * A.outStride + co;
                if (co + 3 < A.cout) *(sd_f4*)dst = v;
                else for (int e = 0; e < 4 && co + e < A.cout; e++) dst[e] = v[e];
            }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// 3 x 3, stride 1 on limbs: the activations of a 16-channel chunk are split and staged ONCE for all nine taps.  The workgroup's 128
// output pixels are consecutive in the flattened [N][H][W] index; LDS holds that range plus W + 1 pixels on either side (R = 128 + 2 W + 2
// rows of three limb planes) and tap (kh, kw) of pixel p is simply row p + kh W + kw of it -- or a row of zeros where the tap leaves the
// image (decided per lane once, a 9-bit mask per pixel).  Against k_conv_b3, per 24 MFMAs of a wave: no activation load, no split, no LDS
// write (4 - 9 x fewer of each: the halo costs (128 + 2 W + 2) / 128), one barrier pair per 216 MFMAs instead of one per 24; what remains per
// tap is six weight-fragment requests, six fragment reads and two address selects.  K order = [chunk][tap] (the host lays the weight
// fragments out in that order).  NP = activation pieces per thread and chunk = ceil(4 R / 256).
template <int NP, int WM, int WN>
__global__ void __launch_bounds__(64 * WM * WN, WM * WN == 4 ? 2 : 1) k_conv3x3_b3(SdConvArgsF A, const uint4* __restrict__ wgt3)
{
    constexpr int NT = 64 * WM * WN, BM = 64 * WM, BN = 64 * WN, BK = 16;      // <2, 2>: 128 filters x 128 pixels, two workgroups per CU; <2, 4>: 128 x 256 on eight waves, one per CU:
                                                                               // half the weight-fragment bytes and 0.7 x the halo per MFMA
    extern __shared__ __align__(16) unsigned char smemb[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r32 = lane & 31, h = lane >> 5;
    const int wm = wv % WM, wn = wv / WM;
    const int perXcd = (A.tilesX + 7) >> 3;
    const int slot = blockIdx.x >> 3, perGroup = perXcd * A.groupY;
    const int grp = slot / perGroup, rg = slot - grp * perGroup;
    const int tx = (blockIdx.x & 7) * perXcd + rg / A.groupY, ty = grp * A.groupY + rg % A.groupY;
    if (tx >= A.tilesX) return;
    const int pix0 = tx * BN, co0 = ty * BM;
    const int npix = A.N * A.H * A.W;                    // stride 1, pad 1: output geometry = input geometry
    const int W = A.W;
    const int R = BN + 2 * W + 2;                        // staged rows; row R is the zero row
    const int PLANE = (R + 1) * SD_B3_ROWB;
    const int nchunks = A.cin / BK, ksteps = 9 * nchunks;
    // ---- staging map: piece i of this thread = (row, 4-channel quarter); rows are flattened pixels pix0 - W - 1 + row, clamped into the tensor
    // (a clamped row is only ever addressed by a tap that is masked to the zero row)
    const float* xptr[NP];
    int xoff[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const int piece = tid + NT * i;
        const int row = piece >> 2, q = piece & 3;
        const bool on = row < R;
        const int g = min(max(pix0 - W - 1 + row, 0), npix - 1);
        xptr[i] = A.in + (size_t)g * A.cinStride + 4 * q;
        xoff[i] = on ? row * SD_B3_ROWB + 8 * q : -1;
    }
    sd_f4 xr[NP];
    auto fetchX = [&]() {
#pragma unroll
        for (int i = 0; i < NP; i++) { if (xoff[i] >= 0) xr[i] = *(const sd_f4*)xptr[i]; xptr[i] += BK; }
    };
    auto storeX = [&]() {
#pragma unroll
        for (int i = 0; i < NP; i++) {
            if (xoff[i] < 0) continue;
            uint2 l3[3];
            sd_split3(xr[i][0], xr[i][1], l3[0].x, l3[1].x, l3[2].x);
            sd_split3(xr[i][2], xr[i][3], l3[0].y, l3[1].y, l3[2].y);
#pragma unroll
            for (int l = 0; l < 3; l++) *(uint2*)(smemb + l * PLANE + xoff[i]) = l3[l];
        }
    };
    // ---- this lane's two pixels (n = 0, 1): LDS byte address of the centre-tap row minus (W + 1) rows, and which taps stay inside the image
    int bbase[2];
    unsigned tapok[2];
#pragma unroll
    for (int n = 0; n < 2; n++) {
        const int pl = 64 * wn + 32 * n + r32, p = pix0 + pl;
        bbase[n] = pl * SD_B3_ROWB + 16 * h;
        unsigned m = 0;
        if (p < npix) {
            const int r = p % (A.H * W);
            const int y = r / W, x = r - y * W;
#pragma unroll
            for (int t = 0; t < 9; t++) {
                const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                if (yy >= 0 && yy < A.H && xx >= 0 && xx < W) m |= 1u << t;
            }
        }
        tapok[n] = m;
    }
    const int zrow = R * SD_B3_ROWB + 16 * h;
    if (tid < 3 * (SD_B3_ROWB / 4)) ((uint32_t*)(smemb + (tid / (SD_B3_ROWB / 4)) * PLANE + R * SD_B3_ROWB))[tid % (SD_B3_ROWB / 4)] = 0u;
    sd_f16v acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;
    sd_b8 fa[3][3][2];
    const uint4* aptr = wgt3 + ((size_t)ty * ksteps * WM + wm) * (6 * 64) + lane;
    auto fetchW = [&](const int set) {
#pragma unroll
        for (int l = 0; l < 3; l++)
#pragma unroll
            for (int m = 0; m < 2; m++) fa[set][l][m] = __builtin_bit_cast(sd_b8, aptr[(2 * l + m) * 64]);
        aptr += WM * 6 * 64;
    };
    sd_b8 fb[3][2];
    auto frags = [&](const int t, const int l) {        // t literal
        const int off = ((t / 3) * W + t % 3) * SD_B3_ROWB;
#pragma unroll
        for (int n = 0; n < 2; n++) {
            const int a = (tapok[n] >> t) & 1u ? bbase[n] + off : zrow;
            fb[l][n] = *(const sd_b8*)(smemb + l * PLANE + a);
        }
    };
    auto mfmas = [&](const int set, const int la, const int lb) {
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int n = 0; n < 2; n++)
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][la][m], fb[lb][n], acc[m][n], 0, 0, 0);
    };
    fetchX();
    fetchW(0);
    fetchW(1);                                           // ksteps >= 18
    storeX();
    __syncthreads();
    auto tap = [&](const int t, const int ws) {          // literals
        frags(t, 0);
        frags(t, 1);
        fetchW((ws + 2) % 3);                            // UNCONDITIONAL (the last two steps read 24 KB past the tile's weights, into the buffer's slack): behind a
                                                         // branch the compiler must wait as if the request had not been made, i.e. for the newest loads in flight -- the
                                                         // two-step distance collapses to none and every step eats an L2 round trip (measured: 80.7 -> see DESIGN)
        __builtin_amdgcn_sched_barrier(0);               // the requests stay HERE, two steps ahead of their use: left to itself the scheduler sinks them next
                                                         // to their consumers (shorter live ranges) and the two-step distance is gone
        mfmas(ws, 0, 0);
        frags(t, 2);
        mfmas(ws, 0, 1);
        mfmas(ws, 1, 0);
        mfmas(ws, 1, 1);
        mfmas(ws, 0, 2);
        mfmas(ws, 2, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int c = 0; c < nchunks; c++) {
        fetchX();                                        // the next chunk's f32 pieces travel under this chunk's 216 MFMAs (unconditional: past the last chunk it reads
                                                         // the next pixel's first channels, 64 bytes of slack at the very end of the tensor)
        tap(0, 0); tap(1, 1); tap(2, 2); tap(3, 0); tap(4, 1); tap(5, 2); tap(6, 0); tap(7, 1); tap(8, 2);      // 9 steps: the weight-set phase repeats every chunk
        if (c + 1 < nchunks) {
            __syncthreads();                             // every wave has read its last fragments of this chunk
            storeX();
            __syncthreads();
        }
    }
    // ---- epilogue: as k_conv_b3
    sd_f4 rr[2][2][4];
#pragma unroll
    for (int n = 0; n < 2; n++) {
        const int p = pix0 + 64 * wn + 32 * n + r32;
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int co = co0 + 64 * wm + 32 * m + 8 * g + 4 * h;
                rr[n][m][g] = sd_f4{0.f, 0.f, 0.f, 0.f};
                if (A.res && p < npix && co < A.cout) rr[n][m][g] = *(const sd_f4*)(A.res + (size_t)p * A.resStride + co);
            }
    }
#pragma unroll
    for (int n = 0; n < 2; n++) {
        const int p = pix0 + 64 * wn + 32 * n + r32;
        if (p >= npix) continue;
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int co = co0 + 64 * wm + 32 * m + 8 * g + 4 * h;
                if (co >= A.cout) continue;
                sd_f4 v;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float x = acc[m][n][4 * g + e] + A.bias[co + e];
                    if (A.leaky) x = x > 0.f ? x : 0.1f * x;
                    v[e] = x + rr[n][m][g][e];
                }
                float* dst = A.out + (size_t)p * A.outStride + co;
                if (co + 3 < A.cout) *(sd_f4*)dst = v;
                else for (int e = 0; e < 4 && co + e < A.cout; e++) dst[e] = v[e];
            }
    }
}
#define SD_B3F_LDS(W, BN) (3 * ((BN) + 2 * (W) + 2 + 1) * SD_B3_ROWB)


// ---------------------------------------------------------------------------------------------------------------------------------
// k_conv3x3_b3 with the WEIGHTS of a chunk staged too.  What bounds k_conv3x3_b3 is the issue of vector-memory instructions, not their bytes
// or their latency: a 1 KB request costs the issuing wave ~60 cycles of its instruction stream among MFMAs (MI355X_MICROARCH.md, per-
// instruction constants; tools/micro/mfma_operands.hip reproduces the kernel's step: 98 % of the bf16 MFMA rate without the six weight
// requests per 24 MFMAs, 60 % with them, wherever they come from), and six of them per 768 MFMA cycles is half a wave's time.  Here a
// workgroup of EIGHT waves owns 64 filters x 512 consecutive pixels (every wave the same 64 filters, its own 64 pixels) and stages per
// 16-channel chunk BOTH operands once: the activations as in k_conv3x3_b3 (range + halo, split, three limb planes) and the chunk's 9 taps x
// six weight fragments = 54 KB, copied as they are (the host's fragment order IS the LDS image: slot (tap, fragment) = 1 KB, read back with
// ds_read_b128 at lane * 16).  Per thread and chunk: 7 + 6 requests of 16 bytes instead of 54 + 5; per tap nothing but twelve LDS reads and
// 24 MFMAs, no barrier; two barriers per chunk.  One workgroup per CU (152 KB of LDS at W = 80), two waves per SIMD.
#define SD_B3C_WBYTES (9 * 6 * 1024)
#define SD_B3C_LDS(W) (3 * (512 + 2 * (W) + 2 + 1) * SD_B3_ROWB + SD_B3C_WBYTES)
template <int NP>
__global__ void __launch_bounds__(512, 1) k_conv3x3_b3c(SdConvArgsF A, const uint4* __restrict__ wgt3)
{
    constexpr int NT = 512, BM = 64, BN = 512, BK = 16, NWP = (SD_B3C_WBYTES / 16 + NT - 1) / NT;      // 3456 weight pieces: 6.75 -> 7 per thread
    extern __shared__ __align__(16) unsigned char smemb[];
    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6, r32 = lane & 31, h = lane >> 5;
    // PERSISTENT workgroups: the launch is one workgroup per CU (8 XCDs x 32), and workgroup w of an XCD walks that XCD's tiles w, w + 32, ... in
    // k_conv_f32's XCD-aware order.  With one workgroup per CU nothing else covers a tile's launch, first staging and epilogue (measured ~20 % of
    // these layers), so the tiles are chained instead: the last chunk of a tile requests the FIRST chunk of the next one, the epilogue's stores
    // leave while that chunk is split into LDS, and the matrix pipe only stops for the two barriers.
    const int perXcd = (A.tilesX + 7) >> 3, perGroup = perXcd * A.groupY, slots = perXcd * A.tilesY, stride = gridDim.x >> 3;
    const int npix = A.N * A.H * A.W;
    const int W = A.W;
    const int R = BN + 2 * W + 2;
    const int PLANE = (R + 1) * SD_B3_ROWB;
    unsigned char* wlds = smemb + 3 * PLANE;
    const int nchunks = A.cin / BK;
    auto tile_of = [&](int& slot, int& tx, int& ty) -> bool {      // the first valid tile at or after `slot` on this workgroup's list (the last XCD's
        for (; slot < slots; slot += stride) {                       // list has holes: pixel tiles beyond tilesX)
            const int grp = slot / perGroup, rg = slot - grp * perGroup;
            tx = (blockIdx.x & 7) * perXcd + rg / A.groupY; ty = grp * A.groupY + rg % A.groupY;
            if (tx < A.tilesX) return true;
        }
        return false;
    };
    // the staging map does not depend on the tile: piece i of this thread = (row, 4-channel quarter) of the staged pixel range
    int xoff[NP], xrow[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const int piece = tid + NT * i;
        xrow[i] = piece >> 2;
        xoff[i] = (piece >> 2) < R ? (piece >> 2) * SD_B3_ROWB + 8 * (piece & 3) : -1;
    }
    const float* xptr[NP];
    const uint4* wsrc = wgt3;
    auto point_at = [&](const int tx, const int ty) {    // the staging pointers of a tile's first chunk
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const int g = min(max(tx * BN - W - 1 + xrow[i], 0), npix - 1);
            xptr[i] = A.in + (size_t)g * A.cinStride + 4 * (tid & 3);
        }
        wsrc = wgt3 + (size_t)ty * nchunks * (SD_B3C_WBYTES / 16) + tid;
    };
    const bool wlast = tid + NT * (NWP - 1) < SD_B3C_WBYTES / 16;      // the seventh piece exists for the first 384 threads
    sd_f4 xr[NP];
    sd_u4w wr[NWP];
    auto fetch = [&]() {
#pragma unroll
        for (int i = 0; i < NWP; i++) wr[i] = *(const sd_u4w*)(wsrc + NT * i);      // all seven unconditionally (the last 128 threads' seventh piece is the next chunk's, not stored): a
        wsrc += SD_B3C_WBYTES / 16;                               // conditionally written element sends the whole array to scratch and the requests wait right there
#pragma unroll
        for (int i = 0; i < NP; i++) { xr[i] = *(const sd_f4*)xptr[i]; xptr[i] += BK; }
    };
    auto store = [&]() {
#pragma unroll
        for (int i = 0; i < NWP - 1; i++) *(sd_u4w*)(wlds + (tid + NT * i) * 16) = wr[i];
        if (wlast) *(sd_u4w*)(wlds + (tid + NT * (NWP - 1)) * 16) = wr[NWP - 1];
#pragma unroll
        for (int i = 0; i < NP; i++) {
            if (xoff[i] < 0) continue;
            uint2 l3[3];
            sd_split3(xr[i][0], xr[i][1], l3[0].x, l3[1].x, l3[2].x);
            sd_split3(xr[i][2], xr[i][3], l3[0].y, l3[1].y, l3[2].y);
#pragma unroll
            for (int l = 0; l < 3; l++) *(uint2*)(smemb + l * PLANE + xoff[i]) = l3[l];
        }
    };
    int bbase[2];
#pragma unroll
    for (int n = 0; n < 2; n++) bbase[n] = (64 * wn + 32 * n + r32) * SD_B3_ROWB + 16 * h;
    unsigned tapok[2];
    auto masks = [&](const int pix0) {                  // which taps of this lane's two pixels stay inside the image
#pragma unroll
        for (int n = 0; n < 2; n++) {
            const int p = pix0 + 64 * wn + 32 * n + r32;
            unsigned m = 0;
            if (p < npix) {
                const int r = p % (A.H * W);
                const int y = r / W, x = r - y * W;
#pragma unroll
                for (int t = 0; t < 9; t++) {
                    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                    if (yy >= 0 && yy < A.H && xx >= 0 && xx < W) m |= 1u << t;
                }
            }
            tapok[n] = m;
        }
    };
    const int zrow = R * SD_B3_ROWB + 16 * h;
    if (tid < 3 * (SD_B3_ROWB / 4)) ((uint32_t*)(smemb + (tid / (SD_B3_ROWB / 4)) * PLANE + R * SD_B3_ROWB))[tid % (SD_B3_ROWB / 4)] = 0u;
    sd_f16v acc[2][2];
    sd_b8 fa[3][2], fb[3][2];
    auto fragsW = [&](const int t, const int l) {
#pragma unroll
        for (int m = 0; m < 2; m++) fa[l][m] = *(const sd_b8*)(wlds + (t * 6 + 2 * l + m) * 1024 + lane * 16);
    };
    auto fragsX = [&](const int t, const int l) {
        const int off = ((t / 3) * W + t % 3) * SD_B3_ROWB;
#pragma unroll
        for (int n = 0; n < 2; n++) {
            const int a = (tapok[n] >> t) & 1u ? bbase[n] + off : zrow;
            fb[l][n] = *(const sd_b8*)(smemb + l * PLANE + a);
        }
    };
    auto mfmas = [&](const int la, const int lb) {
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int n = 0; n < 2; n++)
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[la][m], fb[lb][n], acc[m][n], 0, 0, 0);
    };
    auto tap = [&](const int t) {
        fragsW(t, 0); fragsX(t, 0);
        fragsW(t, 1); fragsX(t, 1);
        mfmas(0, 0);
        fragsW(t, 2); fragsX(t, 2);
        mfmas(0, 1);
        mfmas(1, 0);
        mfmas(1, 1);
        mfmas(0, 2);
        mfmas(2, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    int slot = blockIdx.x >> 3, tx, ty;
    bool have = tile_of(slot, tx, ty);
    if (!have) return;                                   // (slots of an XCD beyond its last pixel tile sit at the end of its list)
    point_at(tx, ty);
    fetch();
    store();
    __syncthreads();
    while (have) {
        const int pix0 = tx * BN, co0 = ty * BM;
        masks(pix0);
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;
        int ntx, nty, nslot = slot + stride;
        const bool more = tile_of(nslot, ntx, nty);
        for (int c = 0; c < nchunks; c++) {
            if (c + 1 == nchunks && more) point_at(ntx, nty);      // the last chunk of a tile requests the first chunk of the next
            fetch();                                     // 13 requests per thread under this chunk's 216 MFMAs (unconditional: past the very last chunk they land in the
            __builtin_amdgcn_sched_barrier(0);           // weight buffer's slack / the next pixel's channels and are dropped)
            tap(0); tap(1); tap(2); tap(3); tap(4); tap(5); tap(6); tap(7); tap(8);
            if (c + 1 < nchunks || more) {
                __syncthreads();
                store();
                __syncthreads();
            }
        }
        // ---- epilogue: as k_conv_b3, 64 filters (its loads and stores overlap the next tile's first taps)
        sd_f4 rr[2][2][4];
#pragma unroll
        for (int n = 0; n < 2; n++) {
            const int p = pix0 + 64 * wn + 32 * n + r32;
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int co = co0 + 32 * m + 8 * g + 4 * h;
                    rr[n][m][g] = sd_f4{0.f, 0.f, 0.f, 0.f};
                    if (A.res && p < npix && co < A.cout) rr[n][m][g] = *(const sd_f4*)(A.res + (size_t)p * A.resStride + co);
                }
        }
#pragma unroll
        for (int n = 0; n < 2; n++) {
            const int p = pix0 + 64 * wn + 32 * n + r32;
            if (p >= npix) continue;
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int co = co0 + 32 * m + 8 * g + 4 * h;
                    if (co >= A.cout) continue;
                    sd_f4 v;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        float x = acc[m][n][4 * g + e] + A.bias[co + e];
                        if (A.leaky) x = x > 0.f ? x : 0.1f * x;
                        v[e] = x + rr[n][m][g][e];
                    }
                    float* dst = A.out + (size_t)p * A.outStride + co;
                    if (co + 3 < A.cout) *(sd_f4*)dst = v;
                    else for (int e = 0; e < 4 && co + e < A.cout; e++) dst[e] = v[e];
                }
        }
        slot = nslot; tx = ntx; ty = nty; have = more;
    }
}
