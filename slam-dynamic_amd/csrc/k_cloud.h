// Dense RGB-D back-projection with the dynamic mask: PointCloudMapping::generatePointCloud (src/pointcloudmapping.cc:59-103),
// the one consumer of the semantic mask in the reference, fed by Tracking::CreateNewKeyFrame (src/Tracking.cc:1999-2007:
// dyn_boxes = the frame's objects whose box_status is 0 or 2).  HBM-bound byte work: every third row / column of depth, mask and
// colour is read once, the kept points are written once, in the reference's push_back order (row-major scan).
//   k_cloud_mark   one workgroup per sampled row: keep / skip per sample -> 64-bit keep masks, row counts, masked_num
//   k_cloud_emit   the same geometry: row offset = sum of the counts above it, then each wave writes its 64 samples in order
// Arithmetic as the reference: p.z = d; p.x = (n - cx) * z / fx; p.y = (m - cy) * z / fy in f32 (no contraction);
// pcl::transformPointCloud(tmp, cloud, T.inverse().matrix()) in f64, left to right, narrowed to f32 [PCL-recall].
#pragma once
#include "k_cull.h"

struct sd_cloud_point_dev { float x, y, z; uint8_t b, g, r, a; };

struct SdCloudArgs {
    const SdFrameBoxes* fb; const int* slots;
    const uint8_t* color; size_t colorStride, colorPitch;
    const uint16_t* depth; size_t depthStride, depthPitch; float depthFactor;
    const uint8_t* mask; size_t maskStride, maskPitch;
    float fx, fy, cx, cy;
    const double* Twc;                 // [frames][16] row-major T^-1
    int W, H, cols3, rows3, words;     // sampled columns / rows, 64-bit words per sampled row
    unsigned long long* bits;          // [frames][rows3][words]
    int* rowCount;                     // [frames][rows3][2]: kept, masked
    sd_cloud_point_dev* points; int capPoints;
    int* counts;                       // [frames][2]: points, masked_num
};

__global__ void __launch_bounds__(256) k_cloud_mark(const SdCloudArgs A)
{
    __shared__ double s_box[SD_MAXB][4];
    __shared__ int s_nb, s_kept, s_masked;
    const int f = blockIdx.y, r = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int m = 3 * r;
    const SdFrameBoxes& F = A.fb[A.slots[f]];
    if (tid == 0) {
        int nb = 0;
        for (int j = 0; j < F.nb; j++)
            if (F.box_status[j] == 2 || F.box_status[j] == 0) { for (int k = 0; k < 4; k++) s_box[nb][k] = F.boxes[j][k]; nb++; }
        s_nb = nb; s_kept = 0; s_masked = 0;
    }
    __syncthreads();
    const int nb = s_nb;
    const uint16_t* drow = A.depth + (size_t)f * A.depthPitch + (size_t)m * A.depthStride;
    const uint8_t* mrow = A.mask ? A.mask + (size_t)f * A.maskPitch + (size_t)m * A.maskStride : nullptr;
    int kept = 0, masked = 0;
    for (int j0 = 0; j0 < A.words * 64; j0 += 256) {
        const int j = j0 + tid, n = 3 * j;
        bool keep = false;
        if (j < A.cols3) {
            bool skip = false;
            const double px = (double)(float)n, py = (double)(float)m;
            for (int k = 0; k < nb && !skip; k++) {
                const double x = s_box[k][0], y = s_box[k][1];
                if (x <= px && px < x + s_box[k][2] && y <= py && py < y + s_box[k][3] && mrow && mrow[n] != 0) skip = true;
            }
            masked += skip;
            const float d = (float)drow[n] * A.depthFactor;
            keep = !((double)d < 0.01 || d > 5.0f || skip);
        }
        const unsigned long long bm = __ballot(keep);
        if (lane == 0 && (j0 + tid) / 64 < A.words) A.bits[((size_t)f * A.rows3 + r) * A.words + (j0 + tid) / 64] = bm;
        kept += keep;
    }
    if (kept) atomicAdd(&s_kept, kept);
    if (masked) atomicAdd(&s_masked, masked);
    __syncthreads();
    if (tid == 0) { A.rowCount[((size_t)f * A.rows3 + r) * 2] = s_kept; A.rowCount[((size_t)f * A.rows3 + r) * 2 + 1] = s_masked; }
}

__global__ void __launch_bounds__(256) k_cloud_emit(const SdCloudArgs A)
{
    __shared__ int s_part[4], s_mpart[4], s_wordOff[64];
    const int f = blockIdx.y, r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int m = 3 * r;
    // offset of this row = kept samples of the rows above it; the last row's workgroup also publishes the frame totals
    int a = 0, am = 0;
    const int upto = (r == A.rows3 - 1) ? A.rows3 : r;
    for (int k = tid; k < upto; k += 256) { a += A.rowCount[((size_t)f * A.rows3 + k) * 2]; am += A.rowCount[((size_t)f * A.rows3 + k) * 2 + 1]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); am += __shfl_xor(am, o, 64); }
    if (lane == 0) { s_part[wv] = a; s_mpart[wv] = am; }
    __syncthreads();
    int rowOff = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    if (r == A.rows3 - 1) {
        if (tid == 0) { A.counts[2 * f] = rowOff; A.counts[2 * f + 1] = s_mpart[0] + s_mpart[1] + s_mpart[2] + s_mpart[3]; }
        rowOff -= A.rowCount[((size_t)f * A.rows3 + r) * 2];
    }
    const unsigned long long* bits = A.bits + ((size_t)f * A.rows3 + r) * A.words;
    if (tid < 64) {                                      // exclusive prefix of the words' popcounts
        const int c = tid < A.words ? __popcll(bits[tid]) : 0;
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        s_wordOff[tid] = incl - c;
    }
    __syncthreads();
    const uint16_t* drow = A.depth + (size_t)f * A.depthPitch + (size_t)m * A.depthStride;
    const uint8_t* crow = A.color + (size_t)f * A.colorPitch + (size_t)m * A.colorStride;
    const double* T = A.Twc + (size_t)f * 16;
    for (int w = wv; w < A.words; w += 4) {
        const unsigned long long bm = bits[w];
        if (!((bm >> lane) & 1ull)) continue;
        const int pos = rowOff + s_wordOff[w] + __popcll(bm & ((1ull << lane) - 1ull));
        if (pos >= A.capPoints) continue;
        const int n = 3 * (w * 64 + lane);
        const float z = (float)drow[n] * A.depthFactor;
        float x = ((float)n - A.cx) * z; x = x / A.fx;
        float y = ((float)m - A.cy) * z; y = y / A.fy;
        double X = T[0] * (double)x + T[1] * (double)y; X = X + T[2] * (double)z; X = X + T[3];
        double Y = T[4] * (double)x + T[5] * (double)y; Y = Y + T[6] * (double)z; Y = Y + T[7];
        double Z = T[8] * (double)x + T[9] * (double)y; Z = Z + T[10] * (double)z; Z = Z + T[11];
        sd_cloud_point_dev p;
        p.x = (float)X; p.y = (float)Y; p.z = (float)Z;
        p.b = crow[3 * n]; p.g = crow[3 * n + 1]; p.r = crow[3 * n + 2]; p.a = 255;
        A.points[(size_t)f * A.capPoints + pos] = p;
    }
}
