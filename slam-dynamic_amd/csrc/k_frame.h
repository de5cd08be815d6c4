// HIP kernels for the Frame-side association steps and image preprocessing (gfx950, wave64).
//   k_stereo_match / k_stereo_filter   Frame::ComputeStereoMatches   src/Frame.cc:874-1048
//   k_rgbd                             Frame::ComputeStereoFromRGBD  src/Frame.cc:1051-1072
//   k_cvt_gray                         cvtColor in GrabImage*        src/Tracking.cc:175-200,256-269
//   k_depth_to_f32                     imDepth.convertTo(CV_32F,f)   src/Tracking.cc:271-272
//   k_hamming_matrix                   ORBmatcher::DescriptorDistance src/ORBmatcher.cc:1804-1820
#pragma once
#include "k_extract.h"

__device__ __forceinline__ int sd_hamming256(const uint4 a0, const uint4 a1, const uint4 b0, const uint4 b1)
{
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// One wave per left keypoint.  Row-band membership (the reference's vRowIndices table, Frame.cc:884-900)
// is evaluated directly per (left, right) pair: right keypoint iR is a candidate of row yi iff
// floor(kpY - r) <= yi <= ceil(kpY + r), r = 2*scale[octave]; candidates are visited in increasing iR
// there, so "first best wins" == lexicographic min of (distance, iR) here.
__global__ void __launch_bounds__(256) k_stereo_match(const sd_keypoint* __restrict__ kp,
                                                      const uint8_t* __restrict__ desc, const int* __restrict__ count,
                                                      const uint8_t* __restrict__ pyr, float* __restrict__ uRight,
                                                      float* __restrict__ depthOut, int* __restrict__ sadOut,
                                                      const SdDevPlan* __restrict__ PP, float mbf, float fx)
{
    const SdDevPlan& P = *PP;
    const int f = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int iL = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int imgL = 2 * f, imgR = 2 * f + 1;
    const int N = count[imgL], Nr = count[imgR];
    if (iL >= N) return;
    const size_t o = (size_t)f * P.kpCap + iL;
    const sd_keypoint kL = kp[(size_t)imgL * P.kpCap + iL];
    const uint4* dl = (const uint4*)(desc + ((size_t)imgL * P.kpCap + iL) * 32);
    const uint4 l0 = dl[0], l1 = dl[1];
    const float mb = mbf / fx;
    const float minZ = mb, minD = 0.f;
    const float maxD = mbf / minZ;
    const float uL = kL.x, vL = kL.y;
    const int yi = (int)vL;
    const float minU = uL - maxD, maxU = uL - minD;
    const int levelL = kL.octave;
    unsigned bestKey = ((unsigned)SD_TH_HIGH << 16) | 0xFFFFu;
    const sd_keypoint* kR = kp + (size_t)imgR * P.kpCap;
    const uint8_t* dR = desc + (size_t)imgR * P.kpCap * 32;
    if (!(maxU < 0)) {
        for (int iR = lane; iR < Nr; iR += 64) {
            const sd_keypoint k = kR[iR];
            const float r = 2.0f * P.lv[k.octave].scale;
            const int maxr = (int)ceilf(k.y + r), minr = (int)floorf(k.y - r);
            if (yi < minr || yi > maxr) continue;
            if (k.octave < levelL - 1 || k.octave > levelL + 1) continue;
            if (k.x >= minU && k.x <= maxU) {
                const uint4* dr = (const uint4*)(dR + (size_t)iR * 32);
                const unsigned dist = (unsigned)sd_hamming256(l0, l1, dr[0], dr[1]);
                const unsigned key = (dist << 16) | (unsigned)iR;
                bestKey = min(bestKey, key);
            }
        }
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) bestKey = min(bestKey, (unsigned)__shfl_xor((int)bestKey, s, 64));
    const int bestDist = (int)(bestKey >> 16);
    const int thOrbDist = (SD_TH_HIGH + SD_TH_LOW) / 2;
    float outU = -1.f, outD = -1.f;
    int outS = -1;
    if (bestDist < thOrbDist) {
        const int bestIdxR = (int)(bestKey & 0xFFFFu);
        const float uR0 = kR[bestIdxR].x;
        const SdLevel& g = P.lv[levelL];
        const float sf = g.invScale;
        const float scaleduL = roundf(kL.x * sf), scaledvL = roundf(kL.y * sf), scaleduR0 = roundf(uR0 * sf);
        const int w = 5, L = 5;
        const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
        if (!(iniu < 0 || endu >= (float)g.W)) {
            const uint8_t* baseL = pyr + (size_t)imgL * P.pyrImageBytes + g.pyrOffset + (size_t)SD_EDGE * g.stride + SD_XOFF;
            const uint8_t* baseR = pyr + (size_t)imgR * P.pyrImageBytes + g.pyrOffset + (size_t)SD_EDGE * g.stride + SD_XOFF;
            const int rowT = (int)(scaledvL - w);
            const uint8_t* IL = baseL + (ptrdiff_t)rowT * g.stride + (int)(scaleduL - w);
            const uint8_t* IR = baseR + (ptrdiff_t)rowT * g.stride + (int)(scaleduR0 - w);   // incR = 0 window
            const int cL = IL[w * g.stride + w];
            int sums[11];
#pragma unroll
            for (int k = 0; k < 11; k++) sums[k] = 0;
            for (int p = lane; p < 121; p += 64) {
                const int yy = p / 11, xx = p - yy * 11;
                const int a = (int)IL[yy * g.stride + xx] - cL;
#pragma unroll
                for (int k = 0; k < 11; k++) {
                    const int inc = k - L;
                    const int cR = IR[w * g.stride + w + inc];
                    const int b = (int)IR[yy * g.stride + xx + inc] - cR;
                    const int d = a - b;
                    sums[k] += d < 0 ? -d : d;
                }
            }
#pragma unroll
            for (int k = 0; k < 11; k++)
#pragma unroll
                for (int s = 32; s > 0; s >>= 1) sums[k] += __shfl_xor(sums[k], s, 64);
            int bestS = 0x7FFFFFFF, bestinc = 0;
#pragma unroll
            for (int k = 0; k < 11; k++)
                if (sums[k] < bestS) { bestS = sums[k]; bestinc = k - L; }
            if (!(bestinc == -L || bestinc == L)) {
                float dist1 = 0.f, dist2 = 0.f, dist3 = 0.f;
#pragma unroll
                for (int k = 1; k < 10; k++)
                    if (k - L == bestinc) { dist1 = (float)sums[k - 1]; dist2 = (float)sums[k]; dist3 = (float)sums[k + 1]; }
                const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
                if (!(deltaR < -1 || deltaR > 1)) {
                    float bestuR = g.scale * ((float)scaleduR0 + (float)bestinc + deltaR);
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
    }
    if (lane == 0) { uRight[o] = outU; depthOut[o] = outD; sadOut[o] = outS; }
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
    const size_t base = (size_t)f * P.kpCap;
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
__global__ void __launch_bounds__(256) k_rgbd(const sd_keypoint* __restrict__ kp, const int* __restrict__ count,
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
    if (sizeof(T) == 2) d = (float)raw * factor; else d = (float)raw;
    float ur = -1.f, dd = -1.f;
    if (d > 0) { dd = d; ur = k.x - mbf / d; }
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
