// The detector's 3 x 3, stride-1 layers as Winograd F(2 x 2, 3 x 3) in f32 (mode SD_YOLO_F32W): 16 multiplies per 2 x 2 output block and
// (filter, channel) pair instead of 36, i.e. 2.25 x fewer MFMA FLOPs on the layers that hold ~3/4 of the network's arithmetic.  Operands
// and accumulation stay f32; what changes against SD_YOLO_F32 is the summation (sums and differences of inputs and of weights are
// multiplied instead of the inputs and weights themselves), so the results differ from a direct f32 convolution in the last bits --
// the mode is held to the same layer tolerance and the same box-set test as SD_YOLO_F32 (tests/test_gpu_yolo.py).
//   host (sd_yolo_load_darknet_weights)   U = G g G^T per (filter, channel), stored [coutPad][16][cin]
//   k_wino_input                          V = B^T d B per (2 x 2 output block, channel) over its 4 x 4 input patch, stored [block][16][cin]
//   k_wino_gemm_f32                       M_xi = U_xi V_xi for the 16 positions xi as ONE GEMM over K = 16 cin whose accumulator is folded
//                                         into the four outputs Y = A^T M A every cin channels; + bias + leaky ReLU + shortcut
// G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]], A^T = [[1,1,1,0],[0,1,-1,-1]].
#pragma once
#include "k_yolo32.h"

struct SdWinoArgs {
    const float* V; const float* U; const float* bias; const float* res; float* out;
    const float* zero;               // >= 16 zero bytes: the source of block rows beyond the last block
    int N, H, W;                     // images, output (= input) rows and columns
    int th, tw;                      // 2 x 2 blocks per image: (H + 1) / 2 x (W + 1) / 2
    int cin, cout, outStride, resStride, leaky;
    int tilesX, tilesY, groupY;      // block tiles, filter tiles; launch order as k_conv_f32
};

// One thread per (block, 4 channels): 16 16-byte loads of the patch (zero outside the image), 32 + 32 additions, 16 16-byte stores.
// Consecutive threads take consecutive channel pieces of a block, so a block's 16 rows of cin floats are written as whole rows.
__global__ void __launch_bounds__(256) k_wino_input(const float* __restrict__ in, int N, int H, int W, int C, int cstride, int th, int tw,
                                                    float* __restrict__ V)
{
    const int c4n = C >> 2;
    const size_t total = (size_t)N * th * tw * c4n;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c4 = (int)(idx % c4n);
        const size_t t = idx / c4n;
        const int bx = (int)(t % tw);
        const size_t r = t / tw;
        const int by = (int)(r % th), n = (int)(r / th);
        const int y0 = 2 * by - 1, x0 = 2 * bx - 1;
        sd_f4 d[4][4];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int y = y0 + i, x = x0 + j;
                d[i][j] = sd_f4{0.f, 0.f, 0.f, 0.f};
                if (y >= 0 && y < H && x >= 0 && x < W) d[i][j] = *(const sd_f4*)(in + (((size_t)n * H + y) * W + x) * cstride + 4 * c4);
            }
        sd_f4 e[4][4];
#pragma unroll
        for (int j = 0; j < 4; j++) {                   // rows: B^T d
            e[0][j] = d[0][j] - d[2][j];
            e[1][j] = d[1][j] + d[2][j];
            e[2][j] = d[2][j] - d[1][j];
            e[3][j] = d[1][j] - d[3][j];
        }
        float* o = V + t * 16 * (size_t)C + 4 * c4;
#pragma unroll
        for (int i = 0; i < 4; i++) {                   // columns: (B^T d) B
            *(sd_f4*)(o + (size_t)(4 * i + 0) * C) = e[i][0] - e[i][2];
            *(sd_f4*)(o + (size_t)(4 * i + 1) * C) = e[i][1] + e[i][2];
            *(sd_f4*)(o + (size_t)(4 * i + 2) * C) = e[i][2] - e[i][1];
            *(sd_f4*)(o + (size_t)(4 * i + 3) * C) = e[i][1] - e[i][3];
        }
    }
}

// 128 filters x 64 blocks per workgroup of four waves (2 along the filters x 2 along the blocks), a wave owns 64 filters x 32 blocks:
// the running product M (2 MFMA tiles, 32 registers) and the four outputs of a block Y[a][b] (8 tiles, 128 registers).  K is walked as
// in k_conv_f32's 1 x 1 case -- both operands are plain rows of 16 cin floats -- and after the last channel step of position xi the
// wave adds +-M to the Y[a][b] that A^T gives a non-zero coefficient for xi (9 of the 16 (xi, ab) pairs) and clears M.  The fold is
// ~100 VALU instructions against 2 cin / 8 * 4 MFMAs (8192 MFMA cycles at cin = 128) and runs under the MFMAs of the other
// workgroup's wave on the same SIMD.  Two workgroups per CU (<= 256 VGPRs + AGPRs per lane).
template <int BK, int MT>
__global__ void __launch_bounds__(256, MT == 1 ? 3 : 2) __attribute__((amdgpu_waves_per_eu(MT == 1 ? 3 : 2, MT == 1 ? 3 : 2))) k_wino_gemm_f32(SdWinoArgs A)
{
    constexpr int NT = 256, BM = 64 * MT, BN = 64;
    constexpr int LD = BK + 4, CPR = BK / 4;
    constexpr int XC = BN * CPR / NT, WC = BM * CPR / NT;
    constexpr int STAGE = (BM + BN) * LD, NCH = BK / 8;
    static_assert(XC >= 1 && XC * NT == BN * CPR && WC * NT == BM * CPR, "staging map");
    extern __shared__ __align__(16) float smemf[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r32 = lane & 31, h = lane >> 5;
    const int wm = wv & 1, wn = wv >> 1;
    const int perXcd = (A.tilesX + 7) >> 3;
    const int slot = blockIdx.x >> 3, perGroup = perXcd * A.groupY;
    const int grp = slot / perGroup, rg = slot - grp * perGroup;
    const int tx = (blockIdx.x & 7) * perXcd + rg / A.groupY, ty = grp * A.groupY + rg % A.groupY;
    if (tx >= A.tilesX) return;
    const int blk0 = tx * BN, co0 = ty * BM;
    const int nblk = A.N * A.th * A.tw;
    const int wrow = 16 * A.cin;                         // floats per filter and per block
    const int spx = A.cin / BK, ksteps = 16 * spx;
    sd_f16v M[MT], Y[4][MT];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            M[m][r] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; q++) Y[q][m][r] = 0.f;
        }
    sd_f4 xr[2][XC], wr[2][WC];
    const float* wptr[WC];
    const float* xptr[XC];
    int xinc[XC];
#pragma unroll
    for (int i = 0; i < WC; i++) {
        const int chunk = tid + NT * i;
        wptr[i] = A.U + (size_t)(co0 + chunk / CPR) * wrow + 4 * (chunk % CPR);      // weight rows are padded to the filter tile
    }
#pragma unroll
    for (int i = 0; i < XC; i++) {
        const int chunk = tid + NT * i;
        const int t = blk0 + chunk / CPR;
        const bool ok = t < nblk;
        xptr[i] = ok ? A.V + (size_t)t * wrow + 4 * (chunk % CPR) : A.zero;
        xinc[i] = ok ? BK : 0;
    }
    auto fetch = [&](const int set) {
#pragma unroll
        for (int i = 0; i < WC; i++) { wr[set][i] = *(const sd_f4*)wptr[i]; wptr[i] += BK; }
#pragma unroll
        for (int i = 0; i < XC; i++) { xr[set][i] = *(const sd_f4*)xptr[i]; xptr[i] += xinc[i]; }
    };
    auto store = [&](int buf, const int set) {
        float* sW = smemf + buf * STAGE;
        float* sX = sW + BM * LD;
#pragma unroll
        for (int i = 0; i < WC; i++) { const int chunk = tid + NT * i; *(sd_f4*)(sW + (chunk / CPR) * LD + 4 * (chunk % CPR)) = wr[set][i]; }
#pragma unroll
        for (int i = 0; i < XC; i++) { const int chunk = tid + NT * i; *(sd_f4*)(sX + (chunk / CPR) * LD + 4 * (chunk % CPR)) = xr[set][i]; }
    };
    const int aoff = (32 * MT * wm + r32) * LD + 4 * h, boff = BM * LD + (32 * wn + r32) * LD + 4 * h;
    sd_f4 fa[2][MT], fb[2];
    auto frags = [&](int buf, int kc, int s) {
        const float* base = smemf + buf * STAGE;
#pragma unroll
        for (int m = 0; m < MT; m++) fa[s][m] = *(const sd_f4*)(base + aoff + 32 * m * LD + 8 * kc);
        fb[s] = *(const sd_f4*)(base + boff + 8 * kc);
    };
    auto mfmas = [&](int s) {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int m = 0; m < MT; m++)
                M[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][m][j], fb[s][j], M[m], 0, 0, 0);
    };
    // Y[2 a + b] += A^T[a][xi / 4] * A^T[b][xi % 4] * M: the coefficients are 0 or +-1, so the fused multiply-add is an exact add
    auto fold = [&](const int xi) {
        const int xy = xi >> 2, xx = xi & 3;
        const float ra[2] = {xy < 3 ? 1.f : 0.f, xy == 0 ? 0.f : (xy == 1 ? 1.f : -1.f)};
        const float cb[2] = {xx < 3 ? 1.f : 0.f, xx == 0 ? 0.f : (xx == 1 ? 1.f : -1.f)};
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) {
                const float f = ra[a] * cb[b];
                if (f != 0.f) {
#pragma unroll
                    for (int m = 0; m < MT; m++)
#pragma unroll
                        for (int r = 0; r < 16; r++) Y[2 * a + b][m][r] = __builtin_fmaf(f, M[m][r], Y[2 * a + b][m][r]);
                }
            }
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int r = 0; r < 16; r++) M[m][r] = 0.f;
    };
    fetch(0);
    store(0, 0);
    fetch(1);                                           // ksteps >= 16
    fetch(0);
    __syncthreads();
    int left = spx, xi = 0;
    auto step = [&](const int ks, const int par) {
        frags(par, 0, 0);
#pragma unroll
        for (int kc = 0; kc < NCH; kc++) {
            if (kc + 1 < NCH) frags(par, kc + 1, (kc + 1) & 1);
            mfmas(kc & 1);
            if (kc == 0) {
                if (ks + 1 < ksteps) store(par ^ 1, par ^ 1);
                if (ks + 3 < ksteps) fetch(par ^ 1);
            }
        }
        if (--left == 0) { fold(xi); xi++; left = spx; }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int ks = 0; ks < ksteps; ks += 2) { step(ks, 0); step(ks + 1, 1); }       // ksteps is even
    // ---- epilogue: D column = block (lane & 31), rows = filters (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5); a block's four pixels.  The layer's
    // filter count is a multiple of the 128-filter tile (checked on the host), so a lane's eight 16-byte pieces per pixel need no bounds
    // test; the shortcut values of a row of the block (two pixels) are requested together.
    const int t = blk0 + 32 * wn + r32;
    if (t >= nblk) return;
    const int per = A.th * A.tw;
    const int n = t / per, rb = t - n * per;
    const int by = rb / A.tw, bx = rb - by * A.tw;
    const int cob = co0 + 32 * MT * wm + 4 * h;         // piece (m, g) = filters cob + 32 m + 8 g .. + 3
    sd_f4 bias4[MT][4];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int g = 0; g < 4; g++) bias4[m][g] = *(const sd_f4*)(A.bias + cob + 32 * m + 8 * g);
#pragma unroll
    for (int a = 0; a < 2; a++) {
        const int yy = 2 * by + a;
        if (yy >= A.H) continue;
        const size_t p0 = ((size_t)n * A.H + yy) * A.W + 2 * bx;
        const bool second = 2 * bx + 1 < A.W;
        sd_f4 rr[2][MT][4];
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    rr[b][m][g] = sd_f4{0.f, 0.f, 0.f, 0.f};
                    if (A.res && (b == 0 || second)) rr[b][m][g] = *(const sd_f4*)(A.res + (p0 + b) * A.resStride + cob + 32 * m + 8 * g);
                }
#pragma unroll
        for (int b = 0; b < 2; b++) {
            if (b == 1 && !second) continue;
            float* dst = A.out + (p0 + b) * A.outStride + cob;
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    sd_f4 v;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        float x = Y[2 * a + b][m][4 * g + e] + bias4[m][g][e];
                        if (A.leaky) x = x > 0.f ? x : 0.1f * x;
                        v[e] = x + rr[b][m][g][e];
                    }
                    *(sd_f4*)(dst + 32 * m + 8 * g) = v;
                }
        }
    }
}
#define SD_WINO_LDS(BK, MT) (2 * (64 * (MT) + 64) * ((BK) + 4) * 4)
