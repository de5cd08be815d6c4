// HIP kernels for the Frame-side association steps and image preprocessing (gfx950, wave64).
//   k_stereo_match / k_stereo_filter   Frame::ComputeStereoMatches   src/Frame.cc:874-1048
//   k_rgbd                             Frame::ComputeStereoFromRGBD  src/Frame.cc:1051-1072
//   k_cvt_gray                         cvtColor in GrabImage*        src/Tracking.cc:175-200,256-269
//   k_depth_to_f32                     imDepth.convertTo(CV_32F,f)   src/Tracking.cc:271-272
//   k_hamming_matrix                   ORBmatcher::DescriptorDistance src/ORBmatcher.cc:1804-1820
#pragma once
#include "k_sort.h"
#include "k_extract.h"

__device__ __forceinline__ int sd_hamming256(const uint4 a0, const uint4 a1, const uint4 b0, const uint4 b1)
{
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// Keypoints of BOTH eyes bucketed by integer row (counting sort in LDS, grid = (frames, 2 eyes)): rowStart[y] .. rowStart[y+1] are
// the keypoints of that image with (int)y == y.  The order inside a bucket is irrelevant: the matcher takes the lexicographic minimum of
// (distance, iR).  Tables are indexed by IMAGE (2f = left, 2f + 1 = right).
__global__ void __launch_bounds__(256) k_row_sort(const sd_keypoint* __restrict__ kp, const int* __restrict__ count,
                                                  unsigned short* __restrict__ rowIdx, int* __restrict__ rowStart,
                                                  int cap, int H)
{
    extern __shared__ __align__(16) unsigned char smem[];
    int* start = (int*)smem;                 // [H + 1]
    int* fill = start + H + 8;               // [H]
    __shared__ int s_carry;
    const int img = 2 * blockIdx.x + blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    __shared__ int s_wsum[4];
    const int N = count[img];
    const sd_keypoint* k = kp + (size_t)img * cap;
    for (int y = tid; y < H; y += 256) { start[y] = 0; fill[y] = 0; }
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int i = tid; i < N; i += 256) { int y = (int)k[i].y; y = min(max(y, 0), H - 1); atomicAdd(&start[y], 1); }
    __syncthreads();
    for (int y0 = 0; y0 < H; y0 += 256) {    // exclusive scan over rows, 256 at a time
        const int y = y0 + tid;
        const int v = y < H ? start[y] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (lane == 63) s_wsum[wv] = incl;
        __syncthreads();
        int base = s_carry;
        for (int w = 0; w < wv; w++) base += s_wsum[w];
        if (y < H) start[y] = base + incl - v;
        __syncthreads();
        if (tid == 255) s_carry = base + incl;
        __syncthreads();
    }
    if (tid == 0) start[H] = s_carry;
    __syncthreads();
    for (int i = tid; i < N; i += 256) {
        int y = (int)k[i].y; y = min(max(y, 0), H - 1);
        rowIdx[(size_t)img * cap + start[y] + atomicAdd(&fill[y], 1)] = (unsigned short)i;
    }
    for (int y = tid; y <= H; y += 256) rowStart[(size_t)img * (H + 8) + y] = start[y];
}

// Frame::ComputeStereoMatches (Frame.cc:874-1031) with the descriptor search staged through LDS.  A workgroup owns SD_SR_ROWS image rows
// of ONE frame: the left keypoints of those rows (from the left eye's row table) and every right keypoint that can lie in the band of any of
// them (rows Y0 - bandR .. Y1 + bandR of the right eye's table: position, octave, index and the 256-bit descriptor) are staged ONCE, and the
// half-waves then walk left keypoints against LDS-resident candidates -- instead of every left keypoint fetching its ~70 candidates'
// keypoint records and descriptors through L2 again (round 2: 5.3x the algorithmic bytes, one dependent global round trip per step of the
// search).  Row-band membership (the reference's vRowIndices table, Frame.cc:884-900) is evaluated per (left, right) pair: right keypoint iR is
// a candidate of row yi iff floor(kpY - r) <= yi <= ceil(kpY + r), r = 2*scale[octave]; the reference visits candidates in increasing iR, so
// "first best wins" == lexicographic min of (distance, iR) here, whatever the staging order.  More keypoints than a pass holds (SD_SR_LEFT /
// SD_SR_CAND) are taken in further passes.  The 11 x 11 SAD refinement is per matched keypoint as before (its windows are the algorithmic
// bulk of this kernel's bytes).  1-D grid in XCD-aware order: the chunks of a frame meet in one L2.
#define SD_SR_ROWS 16
#define SD_SR_LEFT 128
#define SD_SR_CAND 256
#define SD_SR_RS 96               // staged slice of the right row table: SD_SR_ROWS + 2 * bandR + 2 entries
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) k_stereo_match(const sd_keypoint* __restrict__ kp,
                                                      const uint8_t* __restrict__ desc, const int* __restrict__ count,
                                                      const uint8_t* __restrict__ pyr, float* __restrict__ uRight,
                                                      float* __restrict__ depthOut, int* __restrict__ sadOut,
                                                      const unsigned short* __restrict__ rowIdx,
                                                      const int* __restrict__ rowStart, int bandR,
                                                      const SdDevPlan* __restrict__ PP, float mbf, float fx, int nFrames, int chunksPerFrame)
{
    const SdDevPlan& P = *PP;
    __shared__ __align__(16) uint8_t s_win[8][2 * (2 * 11 * 12 + 11 * 24) + 16];       // per half-wave: the SAD step's left window (twice) + right strip as u16
    __shared__ uint4 s_ld0[SD_SR_LEFT], s_ld1[SD_SR_LEFT], s_rd0[SD_SR_CAND], s_rd1[SD_SR_CAND];
    __shared__ float s_lx[SD_SR_LEFT], s_ly[SD_SR_LEFT], s_rx[SD_SR_CAND], s_ry[SD_SR_CAND];
    __shared__ unsigned s_lmeta[SD_SR_LEFT], s_rmeta[SD_SR_CAND], s_best[SD_SR_LEFT];      // meta = index | octave << 16
    __shared__ float s_bx[SD_SR_LEFT];                                                      // x of the best candidate so far (kpR.pt.x of Frame.cc:957)
    __shared__ int s_rs[SD_SR_RS];
    // the level geometry the refinement step needs, by LEVEL (a per-lane level would otherwise cost a dependent global load of the plan per key point)
    __shared__ float s_lvInv[SD_MAX_LEVELS], s_lvScale[SD_MAX_LEVELS];
    __shared__ int s_lvW[SD_MAX_LEVELS], s_lvStride[SD_MAX_LEVELS], s_lvOff[SD_MAX_LEVELS];
    int f, chunk;
    if (!sd_xcd_image_item(blockIdx.x, chunksPerFrame, nFrames, f, chunk)) return;
    const int tid = threadIdx.x, lane = tid & 63, hl = lane & 31, grp = tid >> 5;
    const int imgL = 2 * f, imgR = 2 * f + 1;
    const int H0 = P.lv[0].H;
    const int Y0 = chunk * SD_SR_ROWS, Y1 = min(Y0 + SD_SR_ROWS, H0);
    if (Y0 >= H0) return;
    const int* rsL = rowStart + (size_t)imgL * (H0 + 8);
    const int* rsR = rowStart + (size_t)imgR * (H0 + 8);
    const unsigned short* ridxL = rowIdx + (size_t)imgL * P.kpCap;
    const unsigned short* ridxR = rowIdx + (size_t)imgR * P.kpCap;
    const int l0 = rsL[Y0], l1 = rsL[Y1];
    if (l0 == l1) return;                                     // no left keypoint in these rows
    const int ya = max(Y0 - bandR, 0), yb = min(Y1 - 1 + bandR, H0 - 1);
    if (tid <= yb + 1 - ya && tid < SD_SR_RS) s_rs[tid] = rsR[ya + tid];
    if (tid >= 128 && tid < 128 + P.nlevels) {
        const SdLevel& gl = P.lv[tid - 128];
        s_lvInv[tid - 128] = gl.invScale; s_lvScale[tid - 128] = gl.scale; s_lvW[tid - 128] = gl.W; s_lvStride[tid - 128] = gl.stride;
        s_lvOff[tid - 128] = gl.pyrOffset + SD_EDGE * gl.stride + SD_XOFF;
    }
    const int r0 = rsR[ya], r1 = rsR[yb + 1];
    const sd_keypoint* kL_ = kp + (size_t)imgL * P.kpCap;
    const sd_keypoint* kR = kp + (size_t)imgR * P.kpCap;
    const uint8_t* dL = desc + (size_t)imgL * P.kpCap * 32;
    const uint8_t* dR = desc + (size_t)imgR * P.kpCap * 32;
    const float mb = mbf / fx;
    const float minZ = mb, minD = 0.f;
    const float maxD = mbf / minZ;
    // lane l of each half holds scale[l]: the band radius of a right keypoint comes from a cross-lane read
    const float scaleOfLane = hl < P.nlevels ? P.lv[hl].scale : 0.f;
    for (int lb = l0; lb < l1; lb += SD_SR_LEFT) {
        const int nL = min(SD_SR_LEFT, l1 - lb);
        __syncthreads();                                       // the previous pass is done with the left arrays (and s_rs is written)
        if (tid < nL) {
            const int iL = ridxL[lb + tid];
            const sd_keypoint* q = kL_ + iL;
            s_lx[tid] = q->x; s_ly[tid] = q->y; s_lmeta[tid] = (unsigned)iL | ((unsigned)q->octave << 16);
            const uint4* d = (const uint4*)(dL + (size_t)iL * 32);
            s_ld0[tid] = d[0]; s_ld1[tid] = d[1];
            s_best[tid] = ((unsigned)SD_TH_HIGH << 16) | 0xFFFFu; s_bx[tid] = 0.f;
        }
        for (int cb = r0; cb < r1; cb += SD_SR_CAND) {
            const int nC = min(SD_SR_CAND, r1 - cb);
            __syncthreads();                                   // left arrays visible / the previous candidate chunk is consumed
            if (tid < nC) {
                const int iR = ridxR[cb + tid];
                const sd_keypoint* q = kR + iR;
                s_rx[tid] = q->x; s_ry[tid] = q->y; s_rmeta[tid] = (unsigned)iR | ((unsigned)q->octave << 16);
                const uint4* d = (const uint4*)(dR + (size_t)iR * 32);
                s_rd0[tid] = d[0]; s_rd1[tid] = d[1];
            }
            __syncthreads();
            for (int j = grp; j < nL; j += 8) {
                const float uL = s_lx[j], vL = s_ly[j];
                const int levelL = (int)(s_lmeta[j] >> 16);
                const uint4 a0 = s_ld0[j], a1 = s_ld1[j];
                const int yi = (int)vL;
                const float minU = uL - maxD, maxU = uL - minD;
                // a candidate must sit on level levelL - 1 .. levelL + 1, so its band radius is at most 2 * scale[levelL + 1]: rows further
                // than that (+1 for the floor / ceil of the band test) hold no candidate of THIS keypoint; bandR is the all-level bound
                const float rmax = 2.0f * __shfl(scaleOfLane, (lane & 32) + min(levelL + 1, P.nlevels - 1), 64);
                const int band = min(bandR, (int)ceilf(rmax) + 1);
                const bool scan = !(maxU < 0);
                int p0 = s_rs[min(max(yi - band, 0), H0 - 1) - ya];
                int p1 = scan ? s_rs[min(max(yi + band, 0), H0 - 1) + 1 - ya] : p0;
                p0 = max(p0, cb) - cb; p1 = min(p1, cb + nC) - cb;
                unsigned bestKey = 0xFFFFFFFFu;
                float bestX = 0.f;
                for (int pb = p0; pb < p1; pb += 32) {          // uniform inside a half
                    const int p = pb + hl;
                    const bool in = p < p1;
                    const int pp = in ? p : p0;
                    const float kx = s_rx[pp], ky = s_ry[pp];
                    const unsigned meta = s_rmeta[pp];
                    const int koct = (int)(meta >> 16);
                    const float r = 2.0f * __shfl(scaleOfLane, (lane & 32) + koct, 64);
                    const int maxr = (int)ceilf(ky + r), minr = (int)floorf(ky - r);
                    const bool cand = in && !(yi < minr || yi > maxr) && !(koct < levelL - 1 || koct > levelL + 1) && kx >= minU && kx <= maxU;
                    const unsigned dist = (unsigned)sd_hamming256(a0, a1, s_rd0[pp], s_rd1[pp]);
                    const unsigned key = (dist << 16) | (meta & 0xFFFFu);
                    if (cand && key < bestKey) { bestKey = key; bestX = kx; }
                }
                unsigned wkey = bestKey;
#pragma unroll
                for (int s = 16; s > 0; s >>= 1) wkey = min(wkey, (unsigned)__shfl_xor((int)wkey, s, 64));
                // keys are unique (they carry iR): exactly one lane holds the winner; only this half-wave owns entry j
                if (bestKey == wkey && wkey < s_best[j]) { s_best[j] = wkey; s_bx[j] = bestX; }
            }
        }
        __syncthreads();
        // ---- sub-pixel refinement of every left keypoint of this pass (Frame.cc:963-1031), one half-wave per keypoint.  Everything the
        // window addresses need is in LDS (the winning candidate's x was kept beside its key), so the 99 dword requests of keypoint j + 8 are
        // issued BEFORE the 1331 absolute differences of keypoint j are formed: one exposed round trip per half-wave instead of one per keypoint.
        struct Win { uint32_t ld[4]; float scaleduR0; int ok; };
        const uint8_t* pyrF = pyr + (size_t)imgL * P.pyrImageBytes;         // imgR = imgL + 1 follows it
        auto issue = [&](int j) -> Win {
            Win wq; wq.ok = 0; wq.scaleduR0 = 0.f;
#pragma unroll
            for (int it = 0; it < 4; it++) wq.ld[it] = 0;
            if (j >= nL) return wq;
            const unsigned bestKey = s_best[j];
            if ((int)(bestKey >> 16) >= (SD_TH_HIGH + SD_TH_LOW) / 2) return wq;
            const int levelL = (int)(s_lmeta[j] >> 16);
            const float sf = s_lvInv[levelL];
            const int gW = s_lvW[levelL], gStride = s_lvStride[levelL];
            const float scaleduL = roundf(s_lx[j] * sf), scaledvL = roundf(s_ly[j] * sf), scaleduR0 = roundf(s_bx[j] * sf);
            const int w = 5, L = 5;
            const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
            if (iniu < 0 || endu >= (float)gW) return wq;
            const int rowT = (int)(scaledvL - w);
            // 32-bit offsets inside the frame's two pyramid blocks (left image first): one wave-uniform base, per-lane offsets
            const int offL = s_lvOff[levelL] + rowT * gStride + (int)(scaleduL - w);
            const int offR = (int)P.pyrImageBytes + s_lvOff[levelL] + rowT * gStride + (int)(scaleduR0 - w) - L;   // incR = 0 window, columns -5 ..
#pragma unroll
            for (int it = 0; it < 4; it++) {
                const int id = hl + 32 * it;
                if (id < 33) { const int yy = id / 3, q = id - yy * 3; wq.ld[it] = *(const sd_u32_una*)(pyrF + (unsigned)(offL + yy * gStride + 4 * q)); }
                else if (id < 99) { const int jj = id - 33, yy = jj / 6, q = jj - yy * 6; wq.ld[it] = *(const sd_u32_una*)(pyrF + (unsigned)(offR + yy * gStride + 4 * q)); }
            }
            wq.scaleduR0 = scaleduR0; wq.ok = 1;
            return wq;
        };
        Win cur = issue(grp);
        for (int j = grp; j < nL; j += 8) {
            const Win nxt = issue(j + 8);
            const int iL = (int)(s_lmeta[j] & 0xFFFFu), levelL = (int)(s_lmeta[j] >> 16);
            const float uL = s_lx[j];
            const size_t o = (size_t)imgL * P.kpCap + iL;         // outputs are indexed by the LEFT image
            float outU = -1.f, outD = -1.f;
            int outS = -1;
            if (cur.ok) {                                         // uniform inside a half
                const float gScale = s_lvScale[levelL];
                const int L = 5;
                // The SAD of Frame.cc:981-1003 for the 11 shifts incR = -5 .. 5:  dist(k) = sum over the 11 x 11 window of
                // |(IL - cL) - (IR(x + k) - cR(k))| = |(IL + cR(k)) - (IR(x + k) + cL)|, all exact integers.  Lane (k, rg) of the half-wave
                // (22 of 32 lanes) owns shift k for the rows of group rg (0: rows 0-5, 1: rows 6-10) and forms TWO differences per
                // v_sad_u16: the windows lie in LDS as u16 (left: plain and shifted by one pixel, so that an odd shift still reads
                // ALIGNED pixel pairs on both sides; right: + cL already added), one v_pk-style add puts cR(k) on the left pair.
                uint16_t* wLA = (uint16_t*)s_win[grp];                // [11][12]  L[y][x]
                uint16_t* wLB = wLA + 11 * 12;                       // [11][12]  L[y][x + 1]
                uint16_t* wR = wLB + 11 * 12;                        // [11][24]  R[y][c] + cL, c = 0 .. 20 (column c = pixel -5 + c of the incR = 0 window)
                const unsigned cL = (unsigned)(__shfl((int)cur.ld[0], (lane & 32) + 16, 64) >> 8) & 255u;   // L[5][5]: dword id 16 = row 5, bytes 4..7
                const unsigned cL2 = cL * 0x10001u;
                __builtin_amdgcn_wave_barrier();                     // the previous keypoint's window reads are done
#pragma unroll
                for (int it = 0; it < 4; it++) {
                    const int id = hl + 32 * it;
                    const uint32_t v = cur.ld[it];
                    const uint32_t lo = __builtin_amdgcn_perm(0u, v, 0x0C010C00u), hi = __builtin_amdgcn_perm(0u, v, 0x0C030C02u);   // bytes -> u16 pairs
                    if (id < 33) {
                        const int yy = id / 3, q = id - yy * 3;
                        uint32_t* pa = (uint32_t*)(wLA + yy * 12 + 4 * q);
                        pa[0] = lo; pa[1] = hi;
                        uint16_t* pb = wLB + yy * 12 + 4 * q;                                   // LB[c - 1 .. c + 2] = bytes 0 .. 3
                        if (q > 0) pb[-1] = (uint16_t)(v & 255u);
                        *(uint32_t*)pb = __builtin_amdgcn_perm(0u, v, 0x0C020C01u);
                        pb[2] = (uint16_t)(v >> 24);
                    } else if (id < 99) {
                        const int jj = id - 33, yy = jj / 6, q = jj - yy * 6;
                        uint32_t* pr = (uint32_t*)(wR + yy * 24 + 4 * q);
                        pr[0] = lo + cL2; pr[1] = hi + cL2;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0): this wave's LDS writes are done
                const int k = hl < 11 ? hl : (hl < 22 ? hl - 11 : 0), rg = hl >= 11 ? 1 : 0;
                const int e = k & 1;
                const unsigned cRk = (unsigned)wR[L * 24 + L + k] - cL;            // R[5][5 + k]
                const unsigned cR2 = cRk * 0x10001u;
                const uint16_t* lsel = e ? wLB : wLA;
                const int xs = e ? 0 : 10;                                         // the pixel left over beside the five aligned pairs
                unsigned acc = 0;
                const int yBeg = rg ? 6 : 0, yEnd = rg ? 11 : 6;
                for (int yy = yBeg; yy < yEnd; yy++) {
                    const uint32_t* lp = (const uint32_t*)(lsel + yy * 12);
                    const uint32_t* rp = (const uint32_t*)(wR + yy * 24 + k + e);   // k + e is even: aligned pairs R[xx + k], R[xx + 1 + k], xx = e + 2 i
#pragma unroll
                    for (int i = 0; i < 5; i++) acc = __builtin_amdgcn_sad_u16(lp[i] + cR2, rp[i], acc);
                    acc = __builtin_amdgcn_sad_u16((unsigned)wLA[yy * 12 + xs] + cRk, (unsigned)wR[yy * 24 + xs + k], acc);
                }
                if (hl >= 22) acc = 0;
                __builtin_amdgcn_wave_barrier();
                // lanes 0 .. 10 of the half: dist(k) = both row groups
                const int hb = lane & 32;
                unsigned dk = acc + (unsigned)__shfl((int)acc, hb + (hl < 11 ? hl + 11 : hl), 64);
                // `if(dist<bestDist)` over incR ascending: the FIRST minimum = min of (dist << 4 | k)
                unsigned key = hl < 11 ? (dk << 4) | (unsigned)hl : 0xFFFFFFFFu;
#pragma unroll
                for (int s = 8; s > 0; s >>= 1) key = min(key, (unsigned)__shfl_xor((int)key, s, 64));
                key = (unsigned)__shfl((int)key, hb, 64);                          // (lanes 16 .. 31 reduced among themselves)
                const int bestS = (int)(key >> 4), bestinc = (int)(key & 15u) - L;
                if (!(bestinc == -L || bestinc == L)) {
                    const int kb = bestinc + L;
                    const float dist1 = (float)(unsigned)__shfl((int)dk, hb + kb - 1, 64), dist2 = (float)bestS,
                                dist3 = (float)(unsigned)__shfl((int)dk, hb + kb + 1, 64);
                    const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
                    if (!(deltaR < -1 || deltaR > 1)) {
                        float bestuR = gScale * ((float)cur.scaleduR0 + (float)bestinc + deltaR);
                        float disparity = (uL - bestuR);
                        if (disparity >= minD && disparity < maxD) {
                            if (disparity <= 0) {
                                disparity = (float)0.01;
                                bestuR = (float)((double)uL - 0.01);
                            }
                            outD = mbf / disparity;
                            outU = bestuR;
                            outS = bestS;
                        }
                    }
                }
            }
            if (hl == 0) { uRight[o] = outU; depthOut[o] = outD; sadOut[o] = outS; }
            cur = nxt;
        }
    }
}

// Median-of-SAD outlier rejection (Frame.cc:1033-1047): median = element [size/2] of the sorted SAD
// distances; a match survives iff (float)dist < 1.5f*1.4f*median.  One workgroup per frame; the k-th
// smallest value is found by bisection on the 16-bit value range (SAD <= 121*510).
__global__ void __launch_bounds__(256) k_stereo_filter(const int* __restrict__ count, float* __restrict__ uRight,
                                                       float* __restrict__ depthOut, const int* __restrict__ sad,
                                                       const SdDevPlan* __restrict__ PP)
{
    const SdDevPlan& P = *PP;
    __shared__ int s_red[4];
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int N = count[2 * f];
    const size_t base = (size_t)(2 * f) * P.kpCap;
    // number of matches
    int c = 0;
    for (int i = tid; i < N; i += 256) c += sad[base + i] >= 0;
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) c += __shfl_xor(c, s, 64);
    if (lane == 0) s_red[wv] = c;
    __syncthreads();
    const int nm = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    __syncthreads();
    if (nm == 0) return;                 // reference: undefined (Frame.cc:1035 on an empty vector)
    const int k = nm / 2;
    int lo = 0, hi = 65535;              // smallest v with #(sad <= v) >= k+1
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        int cc = 0;
        for (int i = tid; i < N; i += 256) { const int s = sad[base + i]; cc += (s >= 0 && s <= mid); }
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) cc += __shfl_xor(cc, s, 64);
        if (lane == 0) s_red[wv] = cc;
        __syncthreads();
        const int tot = s_red[0] + s_red[1] + s_red[2] + s_red[3];
        __syncthreads();
        if (tot >= k + 1) hi = mid; else lo = mid + 1;
    }
    const float median = (float)lo;
    const float thDist = 1.5f * 1.4f * median;
    for (int i = tid; i < N; i += 256) {
        const int s = sad[base + i];
        if (s >= 0 && !((float)s < thDist)) { uRight[base + i] = -1.f; depthOut[base + i] = -1.f; }
    }
}

// Frame::ComputeStereoFromRGBD with mvKeysUn == mvKeys (zero distortion).  T = uint16_t fuses the
// convertTo(CV_32F, factor) of Tracking.cc:271-272 into the lookup; T = float reads a converted map.
template <typename T>
__global__ void __launch_bounds__(256) k_rgbd(const sd_keypoint* __restrict__ kp, const sd_keypoint* __restrict__ kpUn, const int* __restrict__ count,
                                              const T* __restrict__ depth, size_t strideE, size_t pitchE, float factor,
                                              float mbf, float* __restrict__ uRight, float* __restrict__ depthOut,
                                              const SdDevPlan* __restrict__ PP)
{
    const SdDevPlan& P = *PP;
    const int img = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count[img]) return;
    const sd_keypoint k = kp[(size_t)img * P.kpCap + i];
    const float v = k.y, u = k.x;
    const T raw = depth[(size_t)img * pitchE + (size_t)(int)v * strideE + (int)u];
    float d;
    d = (float)raw * factor;        // convertTo(CV_32F, factor) in f32 for CV_16U and CV_32F alike; factor 1 (CV_32F passed through, Tracking.cc:271-272) is exact
    float ur = -1.f, dd = -1.f;
    if (d > 0) { dd = d; ur = kpUn[(size_t)img * P.kpCap + i].x - mbf / d; }      // kpU.pt.x - mbf / d (Frame.cc:1069); the lookup above is at the DISTORTED position
    uRight[(size_t)img * P.kpCap + i] = ur;
    depthOut[(size_t)img * P.kpCap + i] = dd;
}

// cvtColor(*2GRAY), 8-bit: (R*4899 + G*9617 + B*1868 + 8192) >> 14.  4 pixels per thread.
__global__ void __launch_bounds__(256) k_cvt_gray(const uint8_t* __restrict__ src, int W, int H, size_t sstride,
                                                  size_t spitch, int channels, int rgbOrder, uint8_t* __restrict__ dst,
                                                  size_t dstride, size_t dpitch)
{
    const int img = blockIdx.z;
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (y >= H || x0 >= W) return;
    const uint8_t* s = src + (size_t)img * spitch + (size_t)y * sstride + (size_t)x0 * channels;
    uint8_t* d = dst + (size_t)img * dpitch + (size_t)y * dstride + x0;
    const int n = min(4, W - x0);
    for (int k = 0; k < n; k++) {
        const uint8_t* p = s + k * channels;
        const int r = rgbOrder ? p[0] : p[2], g = p[1], b = rgbOrder ? p[2] : p[0];
        d[k] = (uint8_t)((r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14);
    }
}

__global__ void __launch_bounds__(256) k_depth_to_f32(const uint16_t* __restrict__ src, int W, int H, size_t sstrideE,
                                                      size_t spitchE, float factor, float* __restrict__ dst)
{
    const int img = blockIdx.z;
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= H) return;
    dst[((size_t)img * H + y) * W + x] = (float)src[(size_t)img * spitchE + (size_t)y * sstrideE + x] * factor;
}

__global__ void __launch_bounds__(256) k_hamming_matrix(const uint8_t* __restrict__ a, int na,
                                                        const uint8_t* __restrict__ b, int nb,
                                                        uint16_t* __restrict__ out)
{
    const int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y;
    if (i >= na || j >= nb) return;
    const uint4* pa = (const uint4*)(a + (size_t)i * 32);
    const uint4* pb = (const uint4*)(b + (size_t)j * 32);
    out[(size_t)i * nb + j] = (uint16_t)sd_hamming256(pa[0], pa[1], pb[0], pb[1]);
}

// =====================================================================================
// Frame grid, UnprojectStereo, and the Frame<->Frame projection matcher
//   k_grid_cells        Frame::PosInGrid / AssignFeaturesToGrid       src/Frame.cc:463-478,790-800
//   k_unproject         Frame::UnprojectStereo                        src/Frame.cc:1074-1088
//   k_proj_candidates   ORBmatcher::SearchByProjection, search part   src/ORBmatcher.cc:407-485,1485-1570
//   k_proj_resolve      ... assignment order, rotation histogram      src/ORBmatcher.cc:487-559,1572-1627
// =====================================================================================
#define SD_GRID_COLS 64   // FRAME_GRID_COLS, Frame.h:40
#define SD_GRID_ROWS 48   // FRAME_GRID_ROWS, Frame.h:39
#define SD_HISTO 30       // HISTO_LENGTH, ORBmatcher.cc:39
#define SD_PROJ_K 64      // candidates kept per projected point (one per lane)

struct SdCamera { float fx, fy, cx, cy, mbf, mb, mnMinX, mnMaxX, mnMinY, mnMaxY; };

// cell id = posX*48 + posY, or -1 when the keypoint falls outside the 64x48 grid.  The reference's
// per-cell index lists are ordered by keypoint index, so (cell id, index) is the visiting order of
// GetFeaturesInArea (cells ix-major, iy-minor; insertion order inside a cell).
__global__ void __launch_bounds__(256) k_grid_cells(const sd_keypoint* __restrict__ kp, const int* __restrict__ count,
                                                    short* __restrict__ cellOf, SdCamera cam, int cap, int imgStep)
{
    const int img = blockIdx.y * imgStep, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count[img]) return;
    const sd_keypoint k = kp[(size_t)img * cap + i];
    const float wInv = (float)SD_GRID_COLS / (cam.mnMaxX - cam.mnMinX);
    const float hInv = (float)SD_GRID_ROWS / (cam.mnMaxY - cam.mnMinY);
    const int px = (int)roundf((k.x - cam.mnMinX) * wInv), py = (int)roundf((k.y - cam.mnMinY) * hInv);
    const bool in = !(px < 0 || px >= SD_GRID_COLS || py < 0 || py >= SD_GRID_ROWS);
    cellOf[(size_t)img * cap + i] = (short)(in ? px * SD_GRID_ROWS + py : -1);
}

// The device form of mGrid[64][48]: keypoint indices grouped by cell, ascending index inside a cell (the
// reference's push_back order), + the start of every cell (cellStart[3072] = number of in-grid keypoints).
// One workgroup per image: LDS histogram -> exclusive scan -> scatter -> per-cell insertion sort (cells hold
// ~0.7 keypoints on average, so the sort is a handful of compares).
#define SD_GRID_CELLS (SD_GRID_COLS * SD_GRID_ROWS)
__global__ void __launch_bounds__(256) k_grid_sort(const short* __restrict__ cellOf, const int* __restrict__ count,
                                                   unsigned short* __restrict__ sortedIdx,
                                                   unsigned short* __restrict__ cellStart, int cap, int imgStep)
{
    extern __shared__ __align__(16) unsigned char smem[];
    int* start = (int*)smem;                                   // [3072 + 1]
    int* fill = start + SD_GRID_CELLS + 8;                     // [3072]
    unsigned short* out = (unsigned short*)(fill + SD_GRID_CELLS);   // [cap]
    __shared__ int s_wsum[4];
    const int img = blockIdx.x * imgStep, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int N = count[img];
    const short* cells = cellOf + (size_t)img * cap;
    for (int c = tid; c < SD_GRID_CELLS; c += 256) { start[c] = 0; fill[c] = 0; }
    __syncthreads();
    for (int i = tid; i < N; i += 256) { const int c = cells[i]; if (c >= 0) atomicAdd(&start[c], 1); }
    __syncthreads();
    // exclusive scan of 3072 counts: 12 consecutive cells per thread
    int local[12], sum = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) { local[k] = start[tid * 12 + k]; sum += local[k]; }
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    if (lane == 63) s_wsum[wv] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int w = 0; w < wv; w++) run += s_wsum[w];
    const int total = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
#pragma unroll
    for (int k = 0; k < 12; k++) { start[tid * 12 + k] = run; run += local[k]; }
    if (tid == 0) start[SD_GRID_CELLS] = total;
    __syncthreads();
    for (int i = tid; i < N; i += 256) {
        const int c = cells[i];
        if (c >= 0) out[start[c] + atomicAdd(&fill[c], 1)] = (unsigned short)i;
    }
    __syncthreads();
    for (int c = tid; c < SD_GRID_CELLS; c += 256) {            // insertion sort inside each cell
        const int s0 = start[c], e0 = start[c + 1];
        for (int i = s0 + 1; i < e0; i++) {
            const unsigned short v = out[i];
            int j = i - 1;
            while (j >= s0 && out[j] > v) { out[j + 1] = out[j]; j--; }
            out[j + 1] = v;
        }
    }
    __syncthreads();
    for (int i = tid; i < total; i += 256) sortedIdx[(size_t)img * cap + i] = out[i];
    for (int c = tid; c <= SD_GRID_CELLS; c += 256) cellStart[(size_t)img * (SD_GRID_CELLS + 8) + c] = (unsigned short)start[c];
}

__device__ __forceinline__ void sd_mat3_mul_add(const float* __restrict__ T /*row-major 4x4*/, float x, float y, float z,
                                                float& ox, float& oy, float& oz)
{
    // t = (a0*b0 + a1*b1) + a2*b2 ; d = t + c   (f32, left to right, no contraction)
    float s;
    s = T[0] * x + T[1] * y; s = s + T[2] * z; ox = s + T[3];
    s = T[4] * x + T[5] * y; s = s + T[6] * z; oy = s + T[7];
    s = T[8] * x + T[9] * y; s = s + T[10] * z; oz = s + T[11];
}

__global__ void __launch_bounds__(256) k_unproject(const sd_keypoint* __restrict__ kp, const int* __restrict__ count,
                                                   const float* __restrict__ depth, const float* __restrict__ Twc,
                                                   float* __restrict__ xw, uint8_t* __restrict__ flags, SdCamera cam,
                                                   int cap, int imgStep)
{
    const int img = blockIdx.y * imgStep, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count[img]) return;
    const size_t o = (size_t)img * cap + i;
    const float z = depth[o];
    float X = 0.f, Y = 0.f, Z = 0.f;
    uint8_t f = 0;
    if (z > 0) {
        const sd_keypoint k = kp[o];
        const float invfx = 1.0f / cam.fx, invfy = 1.0f / cam.fy;
        const float x = (k.x - cam.cx) * z * invfx;
        const float y = (k.y - cam.cy) * z * invfy;
        sd_mat3_mul_add(Twc + (size_t)blockIdx.y * 16, x, y, z, X, Y, Z);
        f = 1;
    }
    xw[3 * o] = X; xw[3 * o + 1] = Y; xw[3 * o + 2] = Z;
    flags[o] = f;
}

// wave-wide ascending bitonic sort of one 64-bit key per lane
// (keys occupy lanes 0 .. n-1, the other lanes hold the maximum: a network over the first 2^ceil(log2 n) lanes suffices)
__device__ __forceinline__ unsigned long long sd_wave_sort64(unsigned long long key, int lane, int n = 64)
{
    int m = 2;
    while (m < n) m <<= 1;
    for (int k = 2; k <= m && n > 1; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)key, j, 64);
            const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(key >> 32), j, 64);
            const unsigned long long other = ((unsigned long long)hi << 32) | lo;
            const bool takeMin = (((lane & k) == 0) == ((lane & j) == 0));
            key = takeMin ? (key < other ? key : other) : (key > other ? key : other);
        }
    return key;
}
// merge two ascending 64-key sequences held one per lane, keep the 64 smallest (ascending)
__device__ __forceinline__ unsigned long long sd_wave_merge_low64(unsigned long long a, unsigned long long b, int lane)
{
    const unsigned lo = (unsigned)__shfl((int)(unsigned)b, 63 - lane, 64);
    const unsigned hi = (unsigned)__shfl((int)(unsigned)(b >> 32), 63 - lane, 64);
    const unsigned long long br = ((unsigned long long)hi << 32) | lo;
    unsigned long long key = a < br ? a : br;                 // bitonic: the 64 smallest of both
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) {
        const unsigned l2 = (unsigned)__shfl_xor((int)(unsigned)key, j, 64);
        const unsigned h2 = (unsigned)__shfl_xor((int)(unsigned)(key >> 32), j, 64);
        const unsigned long long other = ((unsigned long long)h2 << 32) | l2;
        key = ((lane & j) == 0) ? (key < other ? key : other) : (key > other ? key : other);
    }
    return key;
}

// Phase A.  Projects every Last-frame map point, scans the Current frame's keypoints for the members of
// GetFeaturesInArea(u, v, radius, level range) and keeps the (<= 64) candidates with Hamming distance <= TH_HIGH sorted by
// (distance, visiting order).  No assignment state is touched here.
// A search window holds ~4-15 grid-cell members, so a point gets a GROUP of 16 lanes (four points per wave): the group
// fetches its cell runs, tests 16 members per step, compacts the hits and sorts them in a 16-lane bitonic network.  A point
// whose window spans more than 16 grid columns or yields more than 16 hits is redone by the whole wave (GW = 64).
struct SdProjArgs {
    const sd_keypoint* kp; const uint8_t* desc; const float* uRight; const int* count; const short* cellOf;
    const unsigned short* sortedIdx; const unsigned short* cellStart; const float* xw; const uint8_t* flags; const uint8_t* dmp;
    const float* Tcw; const float* Tlw; unsigned short* cand; uint8_t* ncand; int* errFlag; const SdDevPlan* P;
    SdCamera cam; float th; int bMono; const int2* pairIdx;
    // Tracker mode (sd_tracker): pairs whose `active` entry is 0 are skipped; with redoBelow > 0 only the pairs whose previous
    // search returned fewer than redoBelow matches are searched again (TrackHomo's `if(nmatches<20)` retry with 2*th).
    const int* active; const int* redoNmatch; int redoBelow;
};

// GW lanes (a group) work on point i of the pair; `live` = the group has a point.  Loop bounds are made wave-uniform with
// __any() so that the cross-lane operations are executed by all lanes.  Returns the hit count (may exceed cap: overflow)
// and leaves the sorted keys in `key` (lane gl of the group holds rank gl).
template <int GW>
__device__ __forceinline__ int sd_proj_point(const SdProjArgs& A, int pair, int i, bool live, int gl, int gshift,
                                             unsigned long long* __restrict__ keys /* LDS, GW slots of this group */,
                                             unsigned long long& keyOut, bool& tooWide)
{
    const SdDevPlan& P = *A.P;
    const SdCamera& cam = A.cam;
    const int lane = threadIdx.x & 63;
    const int imgC = A.pairIdx[pair].x, imgL = A.pairIdx[pair].y;
    const int cap = P.kpCap;
    const size_t oL = (size_t)imgL * cap + (live ? i : 0);
    int n = 0;
    tooWide = false;
    bool ok = live && (A.flags[oL] & 1) != 0;
    float u = 0.f, v = 0.f, invzc = 0.f, radius = 0.f;
    int minLevel = -1, maxLevel = -1;
    const float* T = A.Tcw + (size_t)pair * 16;
    const float* Tl = A.Tlw + (size_t)pair * 16;
    const float scaleOfLane = lane < P.nlevels ? P.lv[lane].scale : 0.f;
    {
        const float X = A.xw[3 * oL], Y = A.xw[3 * oL + 1], Z = A.xw[3 * oL + 2];
        const int nLastOctave = A.kp[oL].octave;
        float xc, yc, zc;
        sd_mat3_mul_add(T, X, Y, Z, xc, yc, zc);
        invzc = 1.0f / zc;
        if (invzc < 0) ok = false;
        u = cam.fx * xc * invzc + cam.cx;
        v = cam.fy * yc * invzc + cam.cy;
        if (u < cam.mnMinX || u > cam.mnMaxX) ok = false;
        if (v < cam.mnMinY || v > cam.mnMaxY) ok = false;
        // bForward / bBackward (ORBmatcher.cc:1497-1510): tlc = Rlw*twc + tlw, twc = -Rcw^T * tcw
        float twx, twy, twz, s_;
        s_ = (-T[0]) * T[3] + (-T[4]) * T[7]; twx = s_ + (-T[8]) * T[11];
        s_ = (-T[1]) * T[3] + (-T[5]) * T[7]; twy = s_ + (-T[9]) * T[11];
        s_ = (-T[2]) * T[3] + (-T[6]) * T[7]; twz = s_ + (-T[10]) * T[11];
        float lx, ly, lz;
        sd_mat3_mul_add(Tl, twx, twy, twz, lx, ly, lz);
        const bool bForward = lz > cam.mb && !A.bMono, bBackward = -lz > cam.mb && !A.bMono;
        radius = A.th * __shfl(scaleOfLane, nLastOctave, 64);         // lane l holds scale[l]: no load chained behind the octave
        if (bForward) { minLevel = nLastOctave; maxLevel = -1; }
        else if (bBackward) { minLevel = 0; maxLevel = nLastOctave; }
        else { minLevel = nLastOctave - 1; maxLevel = nLastOctave + 1; }
    }
    const float wInv = (float)SD_GRID_COLS / (cam.mnMaxX - cam.mnMinX);
    const float hInv = (float)SD_GRID_ROWS / (cam.mnMaxY - cam.mnMinY);
    const int nMinCellX = max(0, (int)floorf((u - cam.mnMinX - radius) * wInv));
    const int nMaxCellX = min(SD_GRID_COLS - 1, (int)ceilf((u - cam.mnMinX + radius) * wInv));
    const int nMinCellY = max(0, (int)floorf((v - cam.mnMinY - radius) * hInv));
    const int nMaxCellY = min(SD_GRID_ROWS - 1, (int)ceilf((v - cam.mnMinY + radius) * hInv));
    if (nMinCellX >= SD_GRID_COLS || nMaxCellX < 0 || nMinCellY >= SD_GRID_ROWS || nMaxCellY < 0) ok = false;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    const uint4* dl = (const uint4*)(A.dmp + oL * 32);
    const uint4 l0 = dl[0], l1 = dl[1];
    const sd_keypoint* kC = A.kp + (size_t)imgC * cap;
    const short* cellC = A.cellOf + (size_t)imgC * cap;
    const float* urC = A.uRight + (size_t)imgC * cap;
    const uint8_t* dC = A.desc + (size_t)imgC * cap * 32;
    const float ur = u - cam.mbf * invzc;
    const unsigned short* sorted = A.sortedIdx + (size_t)imgC * cap;
    const unsigned short* cs = A.cellStart + (size_t)imgC * (SD_GRID_CELLS + 8);
    // GetFeaturesInArea visits cells ix-major / iy-minor: cells (ix, minY..maxY) are one contiguous run of the sorted list.
    // Group lane j < nCols fetches the run of column ix = nMinCellX + j; a group prefix sum turns the runs into one flat range
    // so that GW candidates are tested per step.
    int nColsA = ok ? nMaxCellX - nMinCellX + 1 : 0;
    if (nColsA > GW) { tooWide = true; nColsA = 0; }
    int runS = 0, runN = 0;
    if (gl < nColsA) {
        const int ix = nMinCellX + gl;
        runS = cs[ix * SD_GRID_ROWS + nMinCellY];
        runN = cs[ix * SD_GRID_ROWS + nMaxCellY + 1] - runS;
    }
    int incl = runN;
#pragma unroll
    for (int o = 1; o < GW; o <<= 1) { const int t = __shfl_up(incl, o, GW); if (gl >= o) incl += t; }
    const int total = __shfl(incl, GW - 1, GW);
    const int excl = incl - runN;
    int colsU = nColsA;                                            // wave-uniform bounds
#pragma unroll
    for (int o = GW; o < 64; o <<= 1) colsU = max(colsU, __shfl_xor(colsU, o, 64));
    colsU = __builtin_amdgcn_readfirstlane(colsU);
    // GW == 64 (the whole wave on one point): more than 64 hits are possible (wide windows over dense key points); the 64 SMALLEST keys are
    // kept by a running merge, so that a truncated list is the true head of the full one (k_proj_resolve says so if it ever runs out)
    unsigned long long best = ~0ull;
    int nbuf = 0;
    for (int base = 0; __any(base < total); base += GW) {
        const int t = base + gl;
        bool hit = false;
        unsigned long long key = 0;
        // owner column of flat index t: the last group lane whose exclusive prefix is <= t
        int col = 0;
        for (int j = 1; j < colsU; j++) {
            const int ej = __shfl(excl, j, GW);
            if (j < nColsA && ej <= t) col = j;
        }
        const int cS = __shfl(runS, col, GW), cE = __shfl(excl, col, GW);
        if (t < total) {
            // everything a candidate needs is requested as soon as its index is known (one round trip, not one per test)
            const int i2 = sorted[cS + (t - cE)];
            const sd_keypoint* kq = kC + i2;
            const float kx = kq->x, ky = kq->y;
            const int koct = kq->octave;
            const float r2 = urC[i2];
            const int cellKey = cellC[i2];
            const uint4* dr = (const uint4*)(dC + (size_t)i2 * 32);
            const uint4 d0 = dr[0], d1 = dr[1];
            bool lv = true;
            if (bCheckLevels) {
                if (koct < minLevel) lv = false;
                if (maxLevel >= 0 && koct > maxLevel) lv = false;
            }
            const float distx = kx - u, disty = ky - v;
            bool rOk = true;
            if (r2 > 0) { const float er = fabsf(ur - r2); if (er > radius) rOk = false; }
            const int dist = sd_hamming256(l0, l1, d0, d1);
            if (lv && fabsf(distx) < radius && fabsf(disty) < radius && rOk && dist <= SD_TH_HIGH) {
                hit = true;
                key = ((unsigned long long)dist << 32) | ((unsigned long long)cellKey << 16) | (unsigned)i2;
            }
        }
        const unsigned long long mAll = __ballot(hit);
        const unsigned long long m = GW == 64 ? mAll : (mAll >> gshift) & ((1ull << (GW & 63)) - 1ull);
        if (GW == 64) {
            const int h = __popcll(m);
            if (nbuf + h > 64) {                       // flush the waiting keys into the running best-64
                __builtin_amdgcn_wave_barrier();
                unsigned long long kb = gl < nbuf ? keys[gl] : ~0ull;
                kb = sd_wave_sort64(kb, gl, nbuf);
                best = sd_wave_merge_low64(best, kb, gl);
                nbuf = 0;
                __builtin_amdgcn_wave_barrier();
            }
            if (hit) keys[nbuf + __popcll(m & ((1ull << gl) - 1ull))] = key;
            nbuf += h;
            n += h;
        } else {
            if (hit) {
                const int pos = n + __popcll(m & ((1ull << gl) - 1ull));
                if (pos < GW) keys[pos] = key;
            }
            n += __popcll(m);
        }
    }
    if (GW == 64) {
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        unsigned long long kb = gl < nbuf ? keys[gl] : ~0ull;
        kb = sd_wave_sort64(kb, gl, nbuf);
        keyOut = sd_wave_merge_low64(best, kb, gl);
        return n;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): this wave's LDS writes are done
    // group-wide bitonic sort of <= GW keys (one per lane), ascending.  The keys sit in lanes 0 .. n-1 (the rest hold the
    // maximum), so a network over the first m = 2^ceil(log2 n) lanes is enough: windows hold few candidates, which makes this
    // 3-10 compare-exchange stages instead of 21.
    const int nk = min(n, GW);
    unsigned long long key = gl < nk ? keys[gl] : ~0ull;
    int mU = nk;
#pragma unroll
    for (int o = GW; o < 64; o <<= 1) mU = max(mU, __shfl_xor(mU, o, 64));
    mU = __builtin_amdgcn_readfirstlane(mU);
    int m2 = 2;
    while (m2 < mU) m2 <<= 1;
    for (int k = 2; k <= m2 && mU > 1; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)key, j, 64);
            const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(key >> 32), j, 64);
            const unsigned long long other = ((unsigned long long)hi << 32) | lo;
            const bool up = ((gl & k) == 0);
            const bool lower = ((gl & j) == 0);
            const bool takeMin = (up == lower);
            key = takeMin ? (key < other ? key : other) : (key > other ? key : other);
        }
    keyOut = key;
    return n;
}

__global__ void __launch_bounds__(256) k_proj_candidates(const SdProjArgs A)
{
    __shared__ unsigned long long s_keys[4][SD_PROJ_K];
    const int pair = blockIdx.y;
    if (A.active && !A.active[pair]) return;
    if (A.redoBelow > 0 && A.redoNmatch[pair] >= A.redoBelow) return;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int Nl = A.count[A.pairIdx[pair].y];
    const int cap = A.P->kpCap;
    const int iw = (blockIdx.x * 4 + wv) * 4;                       // first of the wave's four points
    if (iw >= Nl) return;
    const int sub = lane >> 4, gl = lane & 15;
    const int i = iw + sub;
    const bool live = i < Nl;
    unsigned long long key;
    bool wide;
    const int n = sd_proj_point<16>(A, pair, i, live, gl, lane & 48, s_keys[wv] + 16 * sub, key, wide);
    const bool redo = live && (wide || n > 16);
    const size_t oOut = (size_t)pair * cap + i;
    if (live && !redo) {
        if (gl < n) A.cand[oOut * SD_PROJ_K + gl] = (unsigned short)(key & 0xFFFFu);
        if (gl == 0) A.ncand[oOut] = (uint8_t)n;
    }
    unsigned long long redoMask = __ballot(redo && gl == 0);
    while (redoMask) {                                               // rare: the whole wave takes the point
        const int s = (__ffsll((long long)redoMask) - 1) >> 4;
        redoMask &= redoMask - 1;
        __builtin_amdgcn_wave_barrier();
        unsigned long long key64;
        bool wide64;
        int n64 = sd_proj_point<64>(A, pair, iw + s, true, lane, 0, s_keys[wv], key64, wide64);
        if (n64 > SD_PROJ_K) n64 = SD_PROJ_K;                         // the 64 nearest of more: a list of exactly 64 may be truncated (k_proj_resolve)
        const size_t o64 = (size_t)pair * cap + iw + s;
        if (lane < n64) A.cand[o64 * SD_PROJ_K + lane] = (unsigned short)(key64 & 0xFFFFu);
        if (lane == 0) A.ncand[o64] = (uint8_t)n64;
    }
}

// Phase B: one wave per frame pair walks the Last-frame points in index order (the order that decides
// which Current keypoints are already taken, ORBmatcher.cc:462-465) in chunks of 64.  A chunk is
// committed in parallel unless two of its points want the same keypoint and the earlier one carries a
// map point with observations; only then the chunk is replayed serially.
__global__ void __launch_bounds__(64) k_proj_resolve(
    const sd_keypoint* __restrict__ kp, const int* __restrict__ count, const uint8_t* __restrict__ flags,
    const unsigned short* __restrict__ cand, const uint8_t* __restrict__ ncand, const uint8_t* __restrict__ occupied,
    int* __restrict__ matchOut, int* __restrict__ pairsOut, int* __restrict__ npairsOut, int* __restrict__ nmatchOut,
    const SdDevPlan* __restrict__ PP, int checkOrientation, const int2* __restrict__ pairIdx,
    const int* __restrict__ active, int redoBelow, int* __restrict__ errFlag)
{
    if (active && !active[blockIdx.x]) return;                       // tracker mode: see SdProjArgs
    if (redoBelow > 0 && nmatchOut[blockIdx.x] >= redoBelow) return;
    const SdDevPlan& P = *PP;
    extern __shared__ __align__(16) unsigned char smem[];
    const int cap = P.kpCap;
    const int capA = (cap + 15) & ~15;
    int* s_match = (int*)smem;                              // [capA]
    float* s_angL = (float*)(s_match + capA);               // [capA] angles of the Last-frame keypoints
    float* s_angC = s_angL + capA;                          // [capA] angles of the Current-frame keypoints
    unsigned short* s_c0 = (unsigned short*)(s_angC + capA);   // [capA] first candidate of every Last-frame point
    uint8_t* s_taken = (uint8_t*)(s_c0 + capA);             // [capA]
    uint8_t* s_bin = s_taken + capA;                        // [capA] rotation bin of pair p
    uint8_t* s_n = s_bin + capA;                            // [capA] candidate count (<= 64) of every Last-frame point | 0x80 if it has observations
    __shared__ int s_hist[SD_HISTO];
    __shared__ int s_ind[3];
    const int pair = blockIdx.x, lane = threadIdx.x;
    const int imgC = pairIdx[pair].x, imgL = pairIdx[pair].y;
    const int Nl = count[imgL], Nc = count[imgC];
    const sd_keypoint* kC = kp + (size_t)imgC * cap;
    const sd_keypoint* kL = kp + (size_t)imgL * cap;
    const uint8_t* fl = flags + (size_t)imgL * cap;
    const unsigned short* cd = cand + (size_t)pair * cap * SD_PROJ_K;
    const uint8_t* nc = ncand + (size_t)pair * cap;
    int* pairs = pairsOut + (size_t)pair * cap * 2;
    // Everything the serial walk below reads per point is staged in LDS first, with independent requests (eight per lane in
    // flight): the walk itself is one wave per pair, and a global round trip per 64-point chunk was most of its time.
    for (int i0 = lane; i0 < max(Nc, Nl); i0 += 64 * 8) {
        float aC[8], aL[8]; uint8_t tk[8], nn[8]; unsigned short c0[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int i = i0 + 64 * k;
            const int ic = min(i, Nc - 1 < 0 ? 0 : Nc - 1), il = min(i, Nl - 1 < 0 ? 0 : Nl - 1);
            aC[k] = kC[ic].angle; tk[k] = occupied ? occupied[(size_t)pair * cap + ic] : 0;
            aL[k] = kL[il].angle; nn[k] = (uint8_t)(nc[il] | ((fl[il] & 2) ? 0x80 : 0)); c0[k] = cd[(size_t)il * SD_PROJ_K];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int i = i0 + 64 * k;
            if (i < Nc) { s_match[i] = -1; s_taken[i] = tk[k]; s_angC[i] = aC[k]; }
            if (i < Nl) { s_angL[i] = aL[k]; s_n[i] = nn[k]; s_c0[i] = c0[k]; }
        }
    }
    if (lane < SD_HISTO) s_hist[lane] = 0;
    __syncthreads();
    const float factor = 1.0f / SD_HISTO;
    int np = 0;
    for (int base = 0; base < Nl; base += 64) {
        const int i = base + lane;
        const int nf = i < Nl ? s_n[i] : 0;
        const int n = nf & 0x7F;
        const bool obs = (nf & 0x80) != 0;
        // speculative pick against the state at chunk start (the first candidate comes from LDS, the rare later ones from HBM)
        int pick = -1;
        if (n > 0) {
            const int c = s_c0[i];
            if (!s_taken[c]) pick = c;
            else
                for (int j = 1; j < n; j++) {
                    const int c2 = cd[(size_t)i * SD_PROJ_K + j];
                    if (!s_taken[c2]) { pick = c2; break; }
                }
            // a full list may be the head of a longer one (k_proj_candidates keeps the 64 nearest): if every entry is taken the 65th
            // nearest would decide -- say so instead of answering "no match"
            if (pick < 0 && n == SD_PROJ_K) atomicOr(errFlag, 4);
        }
        // conflict: an earlier lane with observations wants the same keypoint
        bool conflict = false;
        const unsigned long long obsMask = __ballot(obs && pick >= 0);
        if (obsMask) {
            for (int s = 1; s < 64; s++) {
                const int src = lane - s;
                const int op = __shfl(pick, src & 63, 64);
                const int oo = __shfl((int)obs, src & 63, 64);
                if (src >= 0 && oo && op >= 0 && op == pick) conflict = true;
            }
        }
        if (__ballot(conflict)) {
            // serial replay of this chunk in index order
            for (int l = 0; l < 64; l++) {
                const int il = base + l;
                if (il >= Nl) break;
                const int nl = nc[il];
                int p = -1;
                for (int j = 0; j < nl; j++) {
                    const int c = cd[(size_t)il * SD_PROJ_K + j];
                    if (!s_taken[c]) { p = c; break; }
                }
                if (p < 0 && nl == SD_PROJ_K && lane == 0) atomicOr(errFlag, 4);
                if (lane == l) pick = p;
                if (p >= 0 && (fl[il] & 2) && lane == 0) s_taken[p] = 1;
                __syncthreads();
            }
        } else {
            if (pick >= 0 && obs) s_taken[pick] = 1;
        }
        // commit in index order: mvpMapPoints[bestIdx2] = pMP (later points overwrite)
        if (pick >= 0) atomicMax(&s_match[pick], i);
        const unsigned long long pm = __ballot(pick >= 0);
        if (pick >= 0) {
            const int p = np + __popcll(pm & ((1ull << lane) - 1ull));
            pairs[2 * p] = i; pairs[2 * p + 1] = pick;
            int bin = 0;
            if (checkOrientation) {
                float rot = s_angL[i] - s_angC[pick];
                if (rot < 0.0f) rot += 360.0f;
                bin = (int)roundf(rot * factor);
                if (bin == SD_HISTO) bin = 0;
                atomicAdd(&s_hist[bin], 1);
            }
            s_bin[p] = (uint8_t)bin;
        }
        np += __popcll(pm);
        __syncthreads();
    }
    int culled = 0;
    if (checkOrientation) {
        if (lane == 0) {
            // ORBmatcher::ComputeThreeMaxima (ORBmatcher.cc:1758-1799)
            int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
            for (int b = 0; b < SD_HISTO; b++) {
                const int s = s_hist[b];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = b; }
                else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = b; }
                else if (s > max3) { max3 = s; ind3 = b; }
            }
            if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
            else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
            s_ind[0] = ind1; s_ind[1] = ind2; s_ind[2] = ind3;
        }
        __syncthreads();
        const int i1 = s_ind[0], i2 = s_ind[1], i3 = s_ind[2];
        for (int p = lane; p < np; p += 64) {
            const int b = s_bin[p];
            if (b != i1 && b != i2 && b != i3) { s_match[pairs[2 * p + 1]] = -1; culled++; }
        }
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) culled += __shfl_xor(culled, s, 64);
        __syncthreads();
    }
    for (int i = lane; i < Nc; i += 64) matchOut[(size_t)pair * cap + i] = s_match[i];
    if (lane == 0) { npairsOut[pair] = np; nmatchOut[pair] = np - culled; }
}

// =====================================================================================
// Local-map search (Tracking::SearchLocalPoints, Tracking.cc:2014-2064)
//   k_local_candidates  Frame::isInFrustum (Frame.cc:677-733) + the search part of
//                       ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th)   ORBmatcher.cc:45-98
//   k_local_resolve     ... best / second best in assignment order, ratio test            ORBmatcher.cc:99-129
// =====================================================================================
struct SdMapPoint { float xw[3]; float normal[3]; float minDistance, maxDistance; unsigned flags; };     // sd_map_point
struct SdTrack { float projX, projY, projXR, viewCos; int level; int inView; };                          // sd_track_info

// std::log(float), taken correctly rounded (the oracle's logf_cr)
__device__ __forceinline__ float sd_logf_cr(float x) { return (float)log((double)x); }


// One wave per local map point.  cand[m][k] = (dist << 16 | keypoint index) of the SD_PROJ_K nearest members of
// GetFeaturesInArea (all distances: the second best may exceed TH_HIGH) ordered by (distance, visiting order);
// ncand[m] = min(count, 64); overflow[m] = 1 when more than 64 keypoints were in the window.
__global__ void __launch_bounds__(256) k_local_candidates(
    const sd_keypoint* __restrict__ kp, const uint8_t* __restrict__ desc, const float* __restrict__ uRight,
    const int* __restrict__ count, const short* __restrict__ cellOf, const unsigned short* __restrict__ sortedIdx,
    const unsigned short* __restrict__ cellStart, const SdMapPoint* __restrict__ mps, const uint8_t* __restrict__ mpDesc,
    const int* __restrict__ frameOf /*[n_frames] image slot*/, const int* __restrict__ ptOff /*[n_frames+1]*/,
    const float* __restrict__ Tcw, SdTrack* __restrict__ track, unsigned* __restrict__ cand, uint8_t* __restrict__ ncand,
    uint8_t* __restrict__ overflow, const SdDevPlan* __restrict__ PP, SdCamera cam, float th, float viewingCosLimit)
{
    const SdDevPlan& P = *PP;
    __shared__ unsigned long long s_keys[4][SD_PROJ_K];
    const int f = blockIdx.y;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m0 = ptOff[f], M = ptOff[f + 1] - m0;
    const int ml = blockIdx.x * 4 + wv;
    if (ml >= M) return;
    const int m = m0 + ml;
    const int img = frameOf[f];
    const int cap = P.kpCap;
    const float* T = Tcw + (size_t)f * 16;
    const SdMapPoint mp = mps[m];
    // ---- Frame::isInFrustum
    SdTrack t; t.projX = t.projY = t.projXR = t.viewCos = 0.f; t.level = 0; t.inView = 0;
    bool ok = (mp.flags & 1u) != 0;
    float invz = 0.f;
    if (ok) {
        float xc, yc, zc;
        sd_mat3_mul_add(T, mp.xw[0], mp.xw[1], mp.xw[2], xc, yc, zc);
        if (zc < 0.0f) ok = false;
        invz = 1.0f / zc;
        const float u = cam.fx * xc * invz + cam.cx;
        const float v = cam.fy * yc * invz + cam.cy;
        if (u < cam.mnMinX || u > cam.mnMaxX) ok = false;
        if (v < cam.mnMinY || v > cam.mnMaxY) ok = false;
        // mOw = -mRcw.t()*mtcw (Frame.cc:667-674)
        float ox, oy, oz, s;
        s = (-T[0]) * T[3] + (-T[4]) * T[7]; ox = s + (-T[8]) * T[11];
        s = (-T[1]) * T[3] + (-T[5]) * T[7]; oy = s + (-T[9]) * T[11];
        s = (-T[2]) * T[3] + (-T[6]) * T[7]; oz = s + (-T[10]) * T[11];
        const float px = mp.xw[0] - ox, py = mp.xw[1] - oy, pz = mp.xw[2] - oz;
        double s2 = (double)px * (double)px; s2 += (double)py * (double)py; s2 += (double)pz * (double)pz;
        const float dist = (float)sqrt(s2);
        if (dist < 0.8f * mp.minDistance || dist > 1.2f * mp.maxDistance) ok = false;
        double dot = (double)px * (double)mp.normal[0]; dot += (double)py * (double)mp.normal[1]; dot += (double)pz * (double)mp.normal[2];
        const float viewCos = (float)(dot / (double)dist);
        if (viewCos < viewingCosLimit) ok = false;
        if (ok) {
            const float ratio = mp.maxDistance / dist;
            const float logScaleFactor = sd_logf_cr(P.lv[1].scale);
            int nScale = (int)ceilf(sd_logf_cr(ratio) / logScaleFactor);
            if (nScale < 0) nScale = 0; else if (nScale >= P.nlevels) nScale = P.nlevels - 1;
            t.inView = 1; t.projX = u; t.projXR = u - cam.mbf * invz; t.projY = v; t.level = nScale; t.viewCos = viewCos;
        }
    }
    if (lane == 0) track[m] = t;
    // ---- candidates
    unsigned long long best = ~0ull;           // lane k holds the k-th smallest key so far
    int total_hits = 0;
    if (ok) {
        float r = (double)t.viewCos > 0.998 ? 2.5f : 4.0f;
        if (th != 1.0f) r *= th;
        const float radius = r * P.lv[t.level].scale;
        const int minLevel = t.level - 1, maxLevel = t.level;
        const float u = t.projX, v = t.projY;
        const float wInv = (float)SD_GRID_COLS / (cam.mnMaxX - cam.mnMinX);
        const float hInv = (float)SD_GRID_ROWS / (cam.mnMaxY - cam.mnMinY);
        const int nMinCellX = max(0, (int)floorf((u - cam.mnMinX - radius) * wInv));
        const int nMaxCellX = min(SD_GRID_COLS - 1, (int)ceilf((u - cam.mnMinX + radius) * wInv));
        const int nMinCellY = max(0, (int)floorf((v - cam.mnMinY - radius) * hInv));
        const int nMaxCellY = min(SD_GRID_ROWS - 1, (int)ceilf((v - cam.mnMinY + radius) * hInv));
        if (!(nMinCellX >= SD_GRID_COLS || nMaxCellX < 0 || nMinCellY >= SD_GRID_ROWS || nMaxCellY < 0)) {
            const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
            const uint4* dl = (const uint4*)(mpDesc + (size_t)m * 32);
            const uint4 l0 = dl[0], l1 = dl[1];
            const sd_keypoint* kC = kp + (size_t)img * cap;
            const short* cellC = cellOf + (size_t)img * cap;
            const float* urC = uRight + (size_t)img * cap;
            const uint8_t* dC = desc + (size_t)img * cap * 32;
            const unsigned short* sorted = sortedIdx + (size_t)img * cap;
            const unsigned short* cs = cellStart + (size_t)img * (SD_GRID_CELLS + 8);
            const int nColsA = nMaxCellX - nMinCellX + 1;
            int runS = 0, runN = 0;
            if (lane < nColsA) {
                const int ix = nMinCellX + lane;
                runS = cs[ix * SD_GRID_ROWS + nMinCellY];
                runN = cs[ix * SD_GRID_ROWS + nMaxCellY + 1] - runS;
            }
            int incl = runN;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int tt = __shfl_up(incl, o, 64); if (lane >= o) incl += tt; }
            const int total = __shfl(incl, 63, 64);
            const int excl = incl - runN;
            int nbuf = 0;                          // keys waiting in s_keys[wv]
            for (int base = 0; base < total; base += 64) {
                const int tt = base + lane;
                bool hit = false;
                unsigned long long key = 0;
                int col = 0;
                for (int j = 1; j < nColsA; j++) { const int ej = __shfl(excl, j, 64); if (ej <= tt) col = j; }
                const int cS = __shfl(runS, col, 64), cE = __shfl(excl, col, 64);
                if (tt < total) {
                    const int i2 = sorted[cS + (tt - cE)];
                    const sd_keypoint k = kC[i2];
                    bool lv = true;
                    if (bCheckLevels) {
                        if (k.octave < minLevel) lv = false;
                        if (maxLevel >= 0 && k.octave > maxLevel) lv = false;
                    }
                    const float distx = k.x - u, disty = k.y - v;
                    if (lv && fabsf(distx) < radius && fabsf(disty) < radius) {
                        bool rOk = true;
                        const float r2 = urC[i2];
                        if (r2 > 0) { const float er = fabsf(t.projXR - r2); if (er > radius) rOk = false; }
                        if (rOk) {
                            const uint4* dr = (const uint4*)(dC + (size_t)i2 * 32);
                            const int dist = sd_hamming256(l0, l1, dr[0], dr[1]);
                            hit = true;
                            key = ((unsigned long long)dist << 32) | ((unsigned long long)cellC[i2] << 16) | (unsigned)i2;
                        }
                    }
                }
                const unsigned long long hm = __ballot(hit);
                const int h = __popcll(hm);
                if (nbuf + h > SD_PROJ_K) {        // flush the waiting keys into the running best-64
                    unsigned long long kb = lane < nbuf ? s_keys[wv][lane] : ~0ull;
                    kb = sd_wave_sort64(kb, lane, nbuf);
                    best = sd_wave_merge_low64(best, kb, lane);
                    nbuf = 0;
                }
                if (hit) s_keys[wv][nbuf + __popcll(hm & ((1ull << lane) - 1ull))] = key;
                nbuf += h;
                total_hits += h;
            }
            unsigned long long kb = lane < nbuf ? s_keys[wv][lane] : ~0ull;
            kb = sd_wave_sort64(kb, lane, nbuf);
            best = sd_wave_merge_low64(best, kb, lane);
        }
    }
    const int n = total_hits < SD_PROJ_K ? total_hits : SD_PROJ_K;
    if (lane < n) cand[(size_t)m * SD_PROJ_K + lane] = (unsigned)((best >> 32) << 16) | (unsigned)(best & 0xFFFFu);
    if (lane == 0) { ncand[m] = (uint8_t)n; overflow[m] = total_hits > SD_PROJ_K; }
}

// One wave per frame walks its local map points in order (ORBmatcher.cc:51) in chunks of 64: lane = map point.  With the
// candidate list sorted by (distance, visiting order), the reference's running best / second best over the keypoints
// that are not taken are the first two free entries of the list.  A chunk is committed in parallel unless an earlier
// point of the chunk (one with observations, whose keypoint becomes taken) picks a keypoint that a later point uses as its
// best or second best; then the chunk is replayed serially.
__global__ void __launch_bounds__(64) k_local_resolve(
    const sd_keypoint* __restrict__ kp, const int* __restrict__ count, const SdMapPoint* __restrict__ mps,
    const int* __restrict__ frameOf, const int* __restrict__ ptOff, const unsigned* __restrict__ cand,
    const uint8_t* __restrict__ ncand, const uint8_t* __restrict__ overflow, const uint8_t* __restrict__ occupied,
    int* __restrict__ mpMatch, int* __restrict__ kpMatch, int* __restrict__ nmatchOut, int* __restrict__ errFlag,
    const SdDevPlan* __restrict__ PP, float nnratio)
{
    const SdDevPlan& P = *PP;
    extern __shared__ __align__(16) unsigned char smem[];
    const int cap = P.kpCap;
    int* s_match = (int*)smem;                              // [cap]
    uint8_t* s_taken = (uint8_t*)(s_match + cap);           // [cap]
    const int f = blockIdx.x, lane = threadIdx.x;
    const int img = frameOf[f];
    const int N = count[img];
    const int m0 = ptOff[f], M = ptOff[f + 1] - m0;
    const sd_keypoint* kF = kp + (size_t)img * cap;
    for (int i = lane; i < N; i += 64) { s_match[i] = -1; s_taken[i] = occupied ? occupied[(size_t)f * cap + i] : 0; }
    __syncthreads();
    int nm = 0;
    // decision of one map point against the current taken[] state; returns the chosen keypoint or -1, c1 = the second free entry
    auto decide = [&](int m, int n, bool ovf, int& c1out) -> int {
        int c0 = -1, c1 = -1, d0 = 256, d1 = 256, j = 0;
        for (; j < n; j++) {
            const unsigned e = cand[(size_t)m * SD_PROJ_K + j];
            const int c = (int)(e & 0xFFFFu);
            if (s_taken[c]) continue;
            if (c0 < 0) { c0 = c; d0 = (int)(e >> 16); }
            else { c1 = c; d1 = (int)(e >> 16); break; }
        }
        if (ovf && c1 < 0) atomicOr(errFlag, 64);           // the truncated list ran out: the 65th+ nearest would matter
        c1out = c1;
        if (c0 < 0 || d0 > SD_TH_HIGH) return -1;
        const int lv0 = kF[c0].octave, lv1 = c1 >= 0 ? kF[c1].octave : -1;
        if (lv0 == lv1 && (float)d0 > nnratio * (float)d1) return -1;
        return c0;
    };
    for (int base = 0; base < M; base += 64) {
        const int ml = base + lane;
        const bool act = ml < M;
        const int m = m0 + (act ? ml : 0);
        const int n = act ? ncand[m] : 0;
        const bool ovf = act && overflow[m];
        const bool obs = act && (mps[m].flags & 2u) != 0;
        int c1 = -1;
        int pick = n > 0 ? decide(m, n, ovf, c1) : -1;
        // the first free entry when the ratio test rejected it still matters to nobody; what matters is whether an
        // earlier taker removes this lane's first or second free entry
        int c0any = -1;
        if (n > 0) { for (int j = 0; j < n; j++) { const int c = (int)(cand[(size_t)m * SD_PROJ_K + j] & 0xFFFFu); if (!s_taken[c]) { c0any = c; break; } } }
        bool conflict = false;
        const unsigned long long takers = __ballot(obs && pick >= 0);
        if (takers) {
            for (int s = 1; s < 64; s++) {
                const int src = lane - s;
                const int op = __shfl(pick, src & 63, 64);
                const int oo = __shfl((int)obs, src & 63, 64);
                if (src >= 0 && oo && op >= 0 && (op == c0any || op == c1)) conflict = true;
            }
        }
        if (__ballot(conflict)) {
            for (int l = 0; l < 64; l++) {
                const int il = base + l;
                if (il >= M) break;
                const int mm = m0 + il;
                int cc;
                const int p = decide(mm, ncand[mm], overflow[mm] != 0, cc);      // every lane computes the same value
                if (lane == l) pick = p;
                if (p >= 0 && (mps[mm].flags & 2u) && lane == 0) s_taken[p] = 1;
                __syncthreads();
            }
        } else {
            if (pick >= 0 && obs) s_taken[pick] = 1;
        }
        if (act) mpMatch[m] = pick;
        if (pick >= 0) atomicMax(&s_match[pick], ml);        // F.mvpMapPoints[bestIdx] = pMP: the later point stays
        nm += __popcll(__ballot(pick >= 0));
        __syncthreads();
    }
    for (int i = lane; i < N; i += 64) kpMatch[(size_t)f * cap + i] = s_match[i];
    if (lane == 0) nmatchOut[f] = nm;
}


// =====================================================================================
// Frame::UndistortKeyPoints (Frame.cc:812-842) for a camera with distortion: cv::undistortPoints(pts, mK, mDistCoef, Mat(), mK)
// [OpenCV-recall: five fixed-point iterations in double, see oracle/frame_oracle.inc].  One thread per keypoint; writes the
// mvKeysUn records (the keypoint with pt replaced).
// =====================================================================================
struct SdDistortion { double fx, fy, cx, cy, k1, k2, p1, p2, k3; };
__device__ __forceinline__ void sd_undistort_pt(const SdDistortion& D, float u, float v, float& uo, float& vo)
{
    const double ifx = 1. / D.fx, ify = 1. / D.fy;
    double x = u, y = v;
    x = (x - D.cx) * ifx; y = (y - D.cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((0. * r2 + 0.) * r2 + 0.) * r2) / (1 + ((D.k3 * r2 + D.k2) * r2 + D.k1) * r2);
        const double deltaX = 2 * D.p1 * x * y + D.p2 * (r2 + 2 * x * x);
        const double deltaY = D.p1 * (r2 + 2 * y * y) + 2 * D.p2 * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    uo = (float)(x * D.fx + D.cx);
    vo = (float)(y * D.fy + D.cy);
}
__global__ void __launch_bounds__(256) k_undistort_points(const float* __restrict__ pts, int n, SdDistortion D, float* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float uo, vo;
    sd_undistort_pt(D, pts[2 * i], pts[2 * i + 1], uo, vo);
    out[2 * i] = uo; out[2 * i + 1] = vo;
}
__global__ void __launch_bounds__(256) k_undistort_keypoints(const sd_keypoint* __restrict__ kp, const int* __restrict__ count, int cap,
                                                             SdDistortion D, int identity, sd_keypoint* __restrict__ kpUn)
{
    const int img = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count[img]) return;
    sd_keypoint k = kp[(size_t)img * cap + i];
    if (!identity) { float uo, vo; sd_undistort_pt(D, k.x, k.y, uo, vo); k.x = uo; k.y = vo; }
    kpUn[(size_t)img * cap + i] = k;
}

// Frame::UndistortKeyPoints for a list of frame slots, the static key points (mvKeys -> mvKeysUn) and the per-box dynamic ones
// (mvdynKeys -> mvdynKeysUn): a pure function of the key point, so it is simply re-run whenever the arrays were permuted
// (firstSeparate) or extended (UpdateFrame).  blockIdx.z = 0: static, 1: dynamic.
__global__ void __launch_bounds__(256) k_undistort_slots(const sd_keypoint* __restrict__ kp, const sd_keypoint* __restrict__ kpD, const int* __restrict__ count,
                                                         const int* __restrict__ nDynOf /* &fb[0].nDyn, stride fbStrideInts */, int fbStrideInts,
                                                         const int* __restrict__ slots, int cap, SdDistortion D, sd_keypoint* __restrict__ kpUn,
                                                         sd_keypoint* __restrict__ kpDUn)
{
    const int slot = slots[blockIdx.y], i = blockIdx.x * 256 + threadIdx.x;
    const bool dyn = blockIdx.z != 0;
    const int n = dyn ? nDynOf[(size_t)slot * fbStrideInts] : count[slot];
    if (i >= n) return;
    sd_keypoint k = (dyn ? kpD : kp)[(size_t)slot * cap + i];
    float uo, vo;
    sd_undistort_pt(D, k.x, k.y, uo, vo);
    k.x = uo; k.y = vo;
    (dyn ? kpDUn : kpUn)[(size_t)slot * cap + i] = k;
}
