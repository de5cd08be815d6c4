// The detector in the reference's own arithmetic: f32 operands, f32 accumulation, on v_mfma_f32_32x32x2_f32 (dense peak
// 157.3 TFLOP/s, 1/16 of the f16 rate).  cv::dnn runs YOLOv3 in f32 on the CPU (src/yolo.cc:29 DNN_TARGET_CPU); this mode
// exists so that the box sets can be compared with an f32 forward pass instead of being explained away by a tolerance, and
// so that the f16 mode's disagreements can be counted against it (tests/test_gpu_yolo.py).  Activations NHWC f32, weights
// [coutPad][taps][cin] f32 with batch-norm folded in f32.
//   k_blob_from_image_f32   blobFromImage -> NHWC f32 x 4 channels (R, G, B, 0: one 16-byte piece per pixel; the first layer pairs two
//                           filter taps into one 8-wide K chunk, see `pair` below)
//   k_conv_f32<BK>          implicit-GEMM convolution, 1x1 / 3x3, stride 1 / 2, + bias + leaky ReLU + shortcut
// One f32 MFMA is 64 cycles for 4096 FLOPs and needs ONE float per operand per lane, so the kernel is MFMA-bound with a
// plain structure: 8 waves, K walked in steps of BK channels of one filter tap, tiles staged through registers into LDS.  A lane reads
// its fragments as 16-byte pieces: half h of the wave takes floats [4h, 4h + 4) of an 8-wide K chunk, MFMA j of the chunk
// uses element j on both operands -- which k a (half, j) pair stands for is immaterial as long as A and B agree.
#pragma once
#include "k_yolo.h"

typedef float sd_f4 __attribute__((ext_vector_type(4)));

struct SdConvArgsF {
    const float* in; const float* wgt; const float* bias; const float* res; float* out;
    const float* zero;       // >= 16 zero bytes: the source of padded taps and of rows beyond the last pixel
    int N, H, W, cin, cinStride;
    int Ho, Wo, cout, outStride, resStride;
    int ksize, stride, pad, leaky;
    int pair;                // first layer (BK == 8 only): the input has 4 channels per pixel and K step s holds taps 2 s and 2 s + 1 (tap 9 = zeros):
                             // 5 steps of 8 instead of 9 steps of 8 with five zero channels each; weights [coutPad][5][2][4]
    int tilesX, tilesY, groupY;      // pixel tiles, filter tiles (groupY divides tilesY): the launch is 1-D, SD_F32_GRID(tilesX, tilesY) workgroups
};

// Tile shapes: WM waves along the filters x (8 / WM) waves along the pixels, a wave owns MT x 2 MFMA tiles (32 MT filters x 64
// pixels).  <32, 2, 2>: 128 filters x 256 pixels (every layer with >= 128 filters); <16, 1, 2>: 64 x 512 and <16 | 8, 1, 1>:
// 32 x 512 for the three 64- / 32-filter layers at 320 x 240 and the first layer, which would waste half or three quarters of a
// 128-filter tile.  LDS holds TWO stages: while the MFMAs of stage s run, the tiles of stage s + 1 (fetched into registers one
// step earlier) are written to the other buffer and the global loads of stage s + 2 are issued -- one barrier per step, and
// neither the address arithmetic of the gather nor the LDS writes sit between two barriers with the MFMA pipe idle.
template <int BK, int WM, int MT, int NW>
__global__ void __launch_bounds__(64 * NW, NW == 8 ? 1 : (WM == 2 && BK == 16 ? 3 : 2)) k_conv_f32(SdConvArgsF A)
{
    constexpr int NT = 64 * NW;                         // NW = 8: one workgroup per CU; NW = 4: two independent workgroups per CU
    constexpr int WN = NW / WM, BM = 32 * MT * WM, BN = 64 * WN;
    constexpr int LD = BK + 4;                          // LDS row length in floats (16-byte aligned rows, 2-way conflicts at worst)
    constexpr int CPR = BK / 4;                         // 16-byte chunks per row
    constexpr int XC = (BN * CPR + NT - 1) / NT;        // activation chunks per thread per step
    constexpr int WC = (BM * CPR + NT - 1) / NT;
    constexpr int STAGE = (BM + BN) * LD;               // floats per stage
    constexpr int NCH = BK / 8;                         // 8-wide K chunks per step
    extern __shared__ __align__(16) float smemf[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r32 = lane & 31, h = lane >> 5;
    const int wm = wv % WM, wn = wv / WM;               // wave tile: filters [32 MT wm, +32 MT), pixels [64 wn, +64)
    // XCD-aware order (workgroup L runs on XCD L % 8, each XCD has its own 4 MB L2): XCD x owns a CONTIGUOUS range of pixel tiles (the
    // rows that neighbouring pixel tiles share for the 3 x 3 taps meet in one L2) and walks it once per GROUP of groupY filter tiles,
    // the filter tiles of a group back to back on each pixel tile.  groupY is chosen on the host so that a group's weights stay
    // L2-resident (<= 2.5 MB): the activations are then fetched from HBM tilesY / groupY times instead of tilesY times, without
    // trading them for weight misses (all 8 filter tiles of a 512 -> 1024 3 x 3 layer are 19 MB of weights).  Measured per 128-image batch
    // (FETCH_SIZE, doubled): 246 GB read in plain (pixel tile, filter tile) grid order, 139 GB with this order at the same 124.1 ms;
    // all filter tiles back to back regardless of their weights: 103 GB but 126.4 ms.
    const int perXcd = (A.tilesX + 7) >> 3;
    const int slot = blockIdx.x >> 3, perGroup = perXcd * A.groupY;
    const int grp = slot / perGroup, r = slot - grp * perGroup;
    const int tx = (blockIdx.x & 7) * perXcd + r / A.groupY, ty = grp * A.groupY + r % A.groupY;
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
        pok[i] = chunk < BN * CPR && p < npix;
        const int pp = pok[i] ? p : 0;
        const int n = pp / (A.Ho * A.Wo), r = pp - n * (A.Ho * A.Wo);
        const int yo = r / A.Wo, xo = r - yo * A.Wo;
        pyi[i] = yo * A.stride - A.pad; pxi[i] = xo * A.stride - A.pad;
        pbase[i] = (size_t)n * A.H * A.W;
    }
    const int taps = A.ksize * A.ksize;
    const bool pair = BK == 8 && A.pair;
    const int ksteps = pair ? (taps + 1) / 2 : taps * (A.cin / BK);
    const int wrow = pair ? 8 * ksteps : taps * A.cin;   // floats per filter
    sd_f16v acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;
    // Staging addresses are kept as running pointers: inside a filter tap a step only adds BK floats to each of them; the (y, x)
    // arithmetic and the bounds test of the im2col gather run once per TAP (every cin / BK steps), and a padded tap reads a zero
    // page through the same unconditional load.  (The per-step form of this arithmetic, ~150 VALU instructions with 64-bit
    // multiplies and branches, cost the MFMA pipe a quarter of its time: tools/micro/conv_loop_cost.hip.)
    // DEEP: two register sets, a stage's tiles are requested TWO steps before they go to LDS.  The 64-filter tile <.., 1, 2, ..> stages
    // four activation chunks per thread; a second set would cost it its third wave per SIMD (186 VGPRs) and more than it gains.
    constexpr bool DEEP = !(WM == 1 && MT == 2);
    constexpr int NSET = DEEP ? 2 : 1;
    sd_f4 xr[NSET][XC], wr[NSET][WC];
    const float* wptr[WC];
    const float* xptr[XC];
    int winc[WC], xinc[XC];
#pragma unroll
    for (int i = 0; i < WC; i++) {
        const int chunk = tid + NT * i;
        const bool on = chunk < BM * CPR;
        wptr[i] = on ? A.wgt + (size_t)(co0 + chunk / CPR) * wrow + 4 * (chunk % CPR) : A.zero;
        winc[i] = on ? BK : 0;                          // (tap, channel) is one contiguous axis of a filter's weights
    }
    int c0 = 0, kh = 0, kw = 0, ps = 0;
    auto retap = [&]() {                                // branch-free: the address of a padded tap is computed and discarded
#pragma unroll
        for (int i = 0; i < XC; i++) {
            const int chunk = tid + NT * i;
            int dy = kh, dx = kw, coff = 4 * (chunk % CPR);
            bool tapok = true;
            if (BK == 8 && pair) {                      // piece q of the row = tap 2 ps + q, all four channels of its pixel
                const int tap = 2 * ps + (chunk % CPR);
                dy = tap / A.ksize; dx = tap - dy * A.ksize; coff = 0;
                tapok = tap < taps;
            }
            const int yi = pyi[i] + dy, xi = pxi[i] + dx;
            const bool ok = pok[i] && tapok && yi >= 0 && yi < A.H && xi >= 0 && xi < A.W;
            const float* p = A.in + ((ptrdiff_t)pbase[i] + (ptrdiff_t)yi * A.W + xi) * A.cinStride + coff;
            xptr[i] = ok ? p : A.zero;
            xinc[i] = ok && !pair ? BK : 0;
        }
    };
    retap();
    auto fetch = [&](const int set) {
#pragma unroll
        for (int i = 0; i < WC; i++) { wr[set][i] = *(const sd_f4*)wptr[i]; wptr[i] += winc[i]; }
#pragma unroll
        for (int i = 0; i < XC; i++) { xr[set][i] = *(const sd_f4*)xptr[i]; xptr[i] += xinc[i]; }
        if (BK == 8 && pair) { ps++; retap(); return; }
        c0 += BK;
        if (c0 == A.cin) { c0 = 0; kw++; if (kw == A.ksize) { kw = 0; kh++; } retap(); }
    };
    auto store = [&](int buf, const int set) {
        float* sW = smemf + buf * STAGE;
        float* sX = sW + BM * LD;
#pragma unroll
        for (int i = 0; i < WC; i++) { const int chunk = tid + NT * i; if ((WC * NT == BM * CPR) || chunk < BM * CPR) *(sd_f4*)(sW + (chunk / CPR) * LD + 4 * (chunk % CPR)) = wr[set][i]; }
#pragma unroll
        for (int i = 0; i < XC; i++) { const int chunk = tid + NT * i; if ((XC * NT == BN * CPR) || chunk < BN * CPR) *(sd_f4*)(sX + (chunk / CPR) * LD + 4 * (chunk % CPR)) = xr[set][i]; }
    };
    const int aoff = (32 * MT * wm + r32) * LD + 4 * h, boff = BM * LD + (64 * wn + r32) * LD + 4 * h;
    sd_f4 fa[2][MT], fb[2][2];                          // fragments of two consecutive K chunks: the reads of chunk c + 1 are issued before the MFMAs of chunk c
    auto frags = [&](int buf, int kc, int slot) {
        const float* base = smemf + buf * STAGE;
#pragma unroll
        for (int m = 0; m < MT; m++) fa[slot][m] = *(const sd_f4*)(base + aoff + 32 * m * LD + 8 * kc);
#pragma unroll
        for (int n = 0; n < 2; n++) fb[slot][n] = *(const sd_f4*)(base + boff + 32 * n * LD + 8 * kc);
    };
    auto mfmas = [&](int slot) {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int n = 0; n < 2; n++)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[slot][m][j], fb[slot][n][j], acc[m][n], 0, 0, 0);
    };
    // The two waves that share a SIMD (w and w + 4) do their staging at DIFFERENT K chunks of a step: the ~150 address / LDS-write
    // instructions of one wave then run under the other wave's MFMAs instead of both leaving the MFMA pipe idle together.
    const int myslot = NW == 8 ? (NCH >= 4 ? 2 * (wv >> 2) : (NCH == 2 ? (wv >> 2) : 0)) : 0;
    // DEEP: stage s lives in register set s & 1 from its request (two steps ahead: ~12 000 cycles with three waves per SIMD, against an
    // L2 / HBM latency of several thousand under load -- with ONE step of distance the ds_writes waited on their loads for 11 % of
    // the kernel time) until it is written to LDS buffer s & 1 during step s - 1.
    fetch(0);
    store(0, 0);
    if (ksteps > 1) fetch(NSET - 1);
    if (DEEP && ksteps > 2) fetch(0);
    __syncthreads();
    // Measured around this loop at batch 128 (ms per detector batch; the ablation builds -- no barrier / no staging / no fragment reads -- gave wrong results, existed for timing only and are no longer in this source):
    // as is 121.0-122.5; without the barrier -1 % (122.9 against 124.1, measured before the two-step prefetch); without any staging (no
    // loads, no LDS writes, no address updates) 109.3; staging without the
    // global loads 120.1 -- i.e. what the staging costs is not the memory latency.  Requests issued unconditionally (so that the
    // compiler can wait with vmcnt(7..4) instead of vmcnt(3..0)) 122.2; a staging map whose ds_write_b128 passes hit 16 distinct bank
    // quads (SQ_LDS_BANK_CONFLICT 0 instead of 33 % of the LDS-active cycles, but the LDS is busy only 10-16 % of the time) 123.7; the
    // barrier moved in front of the step's last chunk with the next step's first fragments requested right behind it (the post-barrier
    // LDS round trip under 16 MFMAs) 121.4.  The same tile staged entirely by LDS-DMA (global_load_lds_dwordx4 into unpadded, XOR-
    // swizzled rows, a ring of three 16 KB stages at three workgroups per CU, counted vmcnt(4) + one barrier per step; results
    // identical) 132.6 against 120.3: slower, not kept.  The weight operand kept out of LDS altogether (each lane fetches its own A
    // fragments from L2 one step ahead, only the activations are staged: half the LDS writes and reads; results identical) 136.0
    // against 119.7: slower, not kept -- so the ~9 % that vanish with the staging are not the LDS writes as such.  s_setprio 1 / 3 around every MFMA cluster (round 4): 117.7 / 117.6
    // and 118.0 / 118.0 against 118.1 and 118.0 -- nothing.  PMC: clock 2.37-2.39 GHz in these kernels (no DVFS give-back), MFMA pipe busy
    // 83 % in the 3 x 3 body layers.
    auto step = [&](const int ks, const int par) {     // par = ks & 1, a literal at both call sites
        const int cur = par;
        frags(cur, 0, 0);
#pragma unroll
        for (int kc = 0; kc < NCH; kc++) {
            if (kc + 1 < NCH) frags(cur, kc + 1, (kc + 1) & 1);
            mfmas(kc & 1);
            if (kc == myslot) {                         // stage ks + 1 into the other buffer, stage ks + 3 requested into the set that frees
                if (ks + 1 < ksteps) store(cur ^ 1, DEEP ? par ^ 1 : 0);
                if (ks + 1 + NSET < ksteps) fetch(DEEP ? par ^ 1 : 0);
            }
        }
        // The barrier stays BEHIND the step's last MFMA: hoisted above them (legal, they touch no memory) the two waves of a SIMD
        // would reach it a staging block apart and the earlier one would sit there with its remaining MFMAs unissued.
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
    };
    int ks = 0;
    for (; ks + 1 < ksteps; ks += 2) { step(ks, 0); step(ks + 1, 1); }
    if (ks < ksteps) step(ks, 0);
    if constexpr (WM == 1) {
        // ---- epilogue.  D column = pixel (lane & 31), rows = filters (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5): written straight from the
        // accumulators a store instruction would cover 32 pixels x 32 bytes.  The wave's 32-pixel half tiles go through its own piece of
        // the (now free) LDS instead -- [pixel][32 MT filters] -- and come back with consecutive lanes on consecutive 16-byte pieces of a
        // pixel's filter row: stores and shortcut reads of 128 / 256 contiguous bytes per pixel.  (The first layers write 5 GB per
        // 128-image batch against a few dozen K steps: their time was the scattered stores.)  Only the <= 64-filter tiles (WM == 1) do
        // this: on the 128-filter tiles, whose epilogue is 3 % of a long K loop and overlaps the other workgroups' MFMAs, the LDS round
        // trip made every 3 x 3 body layer 1-6 % slower (measured), so they keep the direct stores below.
        constexpr int TWD = 32 * MT, TLD = TWD + 4;           // floats per pixel row of the transpose tile (+4: rows 16 bytes apart mod 256)
        constexpr int PPI = 64 / (TWD / 4);                   // pixels per store instruction: a pixel row is TWD / 4 pieces
        static_assert(NW * 32 * TLD <= 2 * STAGE, "the transpose tiles of the epilogue must fit the staging buffers");
        float* T = smemf + wv * 32 * TLD;                     // wave-private: no workgroup barrier inside the epilogue
        const int piece = lane % (TWD / 4), prow = lane / (TWD / 4);
        const int cob = co0 + TWD * wm + 4 * piece;           // first of this lane's four output channels
        sd_f4 bias4 = sd_f4{0.f, 0.f, 0.f, 0.f};
        if (cob < A.cout) bias4 = *(const sd_f4*)(A.bias + cob);                       // bias rows are padded to the filter tile
        __syncthreads();                                      // every wave's last fragment reads are done: the staging buffers are free
#pragma unroll
        for (int n = 0; n < 2; n++) {
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int g = 0; g < 4; g++)
                    *(sd_f4*)(T + r32 * TLD + 32 * m + 8 * g + 4 * h) = sd_f4{acc[m][n][4 * g], acc[m][n][4 * g + 1], acc[m][n][4 * g + 2], acc[m][n][4 * g + 3]};
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int pb = pix0 + 64 * wn + 32 * n;
            sd_f4 rr[32 / PPI];
#pragma unroll
            for (int it = 0; it < 32 / PPI; it++) {            // all shortcut reads requested before the first one is used
                const int p = pb + PPI * it + prow;
                rr[it] = sd_f4{0.f, 0.f, 0.f, 0.f};
                if (A.res && p < npix && cob < A.cout) rr[it] = *(const sd_f4*)(A.res + (size_t)p * A.resStride + cob);
            }
#pragma unroll
            for (int it = 0; it < 32 / PPI; it++) {
                const int pl = PPI * it + prow, p = pb + pl;
                if (p >= npix || cob >= A.cout) continue;
                const sd_f4 a = *(const sd_f4*)(T + pl * TLD + 4 * piece);
                sd_f4 v;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float x = a[e] + bias4[e];
                    if (A.leaky) x = x > 0.f ? x : 0.1f * x;
                    v[e] = x + rr[it][e];
                }
                float* dst = A.out + (size_t)p * A.outStride + cob;
                if (cob + 3 < A.cout) *(sd_f4*)dst = v;
                else for (int e = 0; e < 4 && cob + e < A.cout; e++) dst[e] = v[e];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    } else {
        // ---- epilogue: D column = pixel (lane & 31), rows = filters (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).  All shortcut reads are
        // requested before the first one is used (one memory round trip, not one per 16-byte piece).
        sd_f4 rr[2][MT][4];
#pragma unroll
        for (int n = 0; n < 2; n++) {
            const int p = pix0 + 64 * wn + 32 * n + r32;
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int co = co0 + 32 * MT * wm + 32 * m + 8 * g + 4 * h;
                    rr[n][m][g] = sd_f4{0.f, 0.f, 0.f, 0.f};
                    if (A.res && p < npix && co < A.cout) rr[n][m][g] = *(const sd_f4*)(A.res + (size_t)p * A.resStride + co);
                }
        }
#pragma unroll
        for (int n = 0; n < 2; n++) {
            const int p = pix0 + 64 * wn + 32 * n + r32;
            if (p >= npix) continue;
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int co = co0 + 32 * MT * wm + 32 * m + 8 * g + 4 * h;
                    if (co >= A.cout) continue;
                    sd_f4 v;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        float x = acc[m][n][4 * g + e] + A.bias[co + e];            // bias rows are padded to the filter tile
                        if (A.leaky) x = x > 0.f ? x : 0.1f * x;
                        v[e] = x + rr[n][m][g][e];
                    }
                    float* dst = A.out + (size_t)p * A.outStride + co;
                    if (co + 3 < A.cout) *(sd_f4*)dst = v;
                    else for (int e = 0; e < 4 && co + e < A.cout; e++) dst[e] = v[e];
                }
        }
    }
}
#define SD_F32_GRID(tx, ty) (unsigned)((((tx) + 7) / 8) * 8 * (ty))
#define SD_F32_LDS(BK, WM, MT, NW) (2 * (32 * (MT) * (WM) + 64 * ((NW) / (WM))) * ((BK) + 4) * 4)

// blobFromImage as k_blob_from_image, NHWC f32 with 4 channels (R, G, B after swapRB, then a zero)
__global__ void __launch_bounds__(256) k_blob_from_image_f32(const uint8_t* __restrict__ src, int sw, int sh, size_t sstride, size_t spitch,
                                                             const short4* __restrict__ ct, const short4* __restrict__ rt,
                                                             float* __restrict__ dst, int dw, int dh, int swapRB)
{
    const int img = blockIdx.z;
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= dw || y >= dh) return;
    const short4 ce = ct[x], re = rt[y];
    const int sx = ce.x, sx1 = min(sx + 1, sw - 1);
    const int r0 = min(max((int)re.x, 0), sh - 1), r1 = min(max((int)re.x + 1, 0), sh - 1);
    const uint8_t* S0 = src + (size_t)img * spitch + (size_t)r0 * sstride;
    const uint8_t* S1 = src + (size_t)img * spitch + (size_t)r1 * sstride;
    // the two source pixels of a row are six consecutive bytes: one unaligned dword + one unaligned halfword per row instead of six byte loads (the
    // kernel was bound by the issue of its twelve byte loads per output pixel, not by its 16-byte store); the last source column repeats itself
    // (sx1 == sx) and takes the byte path, which never reads past its pixel
    uint32_t a0, a1, b0, b1;                                    // bytes 0-3 and 4-5 of the pair, rows r0 / r1
    if (sx1 == sx + 1) {
        typedef uint32_t __attribute__((aligned(1))) u32u; typedef uint16_t __attribute__((aligned(1))) u16u;
        a0 = *(const u32u*)(S0 + 3 * sx); b0 = *(const u16u*)(S0 + 3 * sx + 4);
        a1 = *(const u32u*)(S1 + 3 * sx); b1 = *(const u16u*)(S1 + 3 * sx + 4);
    } else {
        a0 = (uint32_t)S0[3 * sx] | ((uint32_t)S0[3 * sx + 1] << 8) | ((uint32_t)S0[3 * sx + 2] << 16) | ((uint32_t)S0[3 * sx1] << 24);
        b0 = (uint32_t)S0[3 * sx1 + 1] | ((uint32_t)S0[3 * sx1 + 2] << 8);
        a1 = (uint32_t)S1[3 * sx] | ((uint32_t)S1[3 * sx + 1] << 8) | ((uint32_t)S1[3 * sx + 2] << 16) | ((uint32_t)S1[3 * sx1] << 24);
        b1 = (uint32_t)S1[3 * sx1 + 1] | ((uint32_t)S1[3 * sx1 + 2] << 8);
    }
    const int p0[6] = {(int)(a0 & 255), (int)((a0 >> 8) & 255), (int)((a0 >> 16) & 255), (int)(a0 >> 24), (int)(b0 & 255), (int)((b0 >> 8) & 255)};
    const int p1[6] = {(int)(a1 & 255), (int)((a1 >> 8) & 255), (int)((a1 >> 16) & 255), (int)(a1 >> 24), (int)(b1 & 255), (int)((b1 >> 8) & 255)};
    float ch[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int h0 = p0[c] * ce.y + p0[3 + c] * ce.z;
        const int h1 = p1[c] * ce.y + p1[3 + c] * ce.z;
        const int v = ((((int)re.y * (h0 >> 4)) >> 16) + (((int)re.z * (h1 >> 4)) >> 16) + 2) >> 2;
        ch[c] = (float)(v & 255) * (float)(1 / 255.0);
    }
    float* o = dst + ((size_t)img * dh * dw + (size_t)y * dw + x) * 4;
    *(sd_f4*)o = sd_f4{swapRB ? ch[2] : ch[0], ch[1], swapRB ? ch[0] : ch[2], 0.f};
}


// [upsample] x 2 (nearest) of `a` (C1 channels at stride aStride, h x w) into channels [0, C1) of `out` (2h x 2w, Ct channels per pixel).  The second
// input of the [route] is not touched: in the f32-class modes its producer writes it in place (sd_yolo_create_prec).  One 16-byte piece per thread:
// a piece is read once per output pixel (four times in all, three of them from L2) and written once.
__global__ void __launch_bounds__(256) k_upsample_into_f32(const float* __restrict__ a, int C1, int aStride, int h, int w, float* __restrict__ out, int Ct, int N)
{
    const int q = C1 / 4, H2 = 2 * h, W2 = 2 * w;
    const size_t total = (size_t)N * H2 * W2 * q;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c4 = (int)(i % q);
        const size_t p = i / q;
        const int x = (int)(p % W2);
        const size_t r = p / W2;
        const int y = (int)(r % H2), n = (int)(r / H2);
        *(sd_f4*)(out + p * Ct + 4 * c4) = *(const sd_f4*)(a + (((size_t)n * h + (y >> 1)) * w + (x >> 1)) * aStride + 4 * c4);
    }
}
