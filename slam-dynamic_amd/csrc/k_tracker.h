// Small kernels of the frame-level boundary (sd_tracker, sd_tracker.inc): everything here is per-lane bookkeeping that must
// not cost a host round trip -- the heavy work is in k_extract.h / k_fast.h / k_frame.h / k_motion.h / k_cull.h.
//   k_copy_frames     Frame's copy constructor (src/Frame.cc:39-63) for a list of (src, dst) slots in one launch
//   k_frame_records   the history-free half of prefetched frames <-> fixed-stride records (frames computed on one GPU, consumed on another)
//   k_lane_gate       `if(!mCurrentFrame.objects.empty() && ...)` (src/Tracking.cc:622) evaluated on the device
//   k_reset_boxes     the constructors without boxes: objects.clear(), N_d = 0
//   k_fill_mono       `mvuRight = vector<float>(N,-1); mvDepth = vector<float>(N,-1);` (src/Frame.cc:432-434)
//   k_lane_summary    what the host needs of a frame after a step, packed for ONE device-to-host copy
#pragma once
#include "k_cull.h"
#include "k_motion.h"

#define SD_COPY_SEGS 24
struct SdCopyTable { char* base[SD_COPY_SEGS]; unsigned slotBytes[SD_COPY_SEGS]; int n; };

// grid (blocks, segments, frames): every segment is a per-slot array, slot k at base + k * slotBytes
__global__ void __launch_bounds__(256) k_copy_frames(SdCopyTable T, const int2* __restrict__ srcDst)
{
    const int seg = blockIdx.y;
    const int2 sd = srcDst[blockIdx.z];
    if (sd.x == sd.y) return;
    const unsigned n = T.slotBytes[seg];
    const char* s = T.base[seg] + (size_t)sd.x * n;
    char* d = T.base[seg] + (size_t)sd.y * n;
    const bool a16 = ((((size_t)s) | ((size_t)d)) & 15) == 0;
    const unsigned quads = a16 ? n >> 4 : 0;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < quads; i += gridDim.x * 256) ((uint4*)d)[i] = ((const uint4*)s)[i];
    for (unsigned i = (quads << 4) + blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) d[i] = s[i];
}

// The same between two workspaces with the same row stride (sd_tracker_prefetch's pool -> the lanes' current slots): lane l copies slot
// srcFirst + l * slotStep of the source arrays to slot l * slotStep of the destination arrays.
struct SdCopyTableX { const char* src[SD_COPY_SEGS]; char* dst[SD_COPY_SEGS]; unsigned slotBytes[SD_COPY_SEGS]; int n; };
__global__ void __launch_bounds__(256) k_copy_frames_x(SdCopyTableX T, int srcFirst, int slotStep)
{
    const int seg = blockIdx.y, l = blockIdx.z;
    const unsigned n = T.slotBytes[seg];
    const char* s = T.src[seg] + (size_t)(srcFirst + l * slotStep) * n;
    char* d = T.dst[seg] + (size_t)(l * slotStep) * n;
    const bool a16 = ((((size_t)s) | ((size_t)d)) & 15) == 0;
    const unsigned quads = a16 ? n >> 4 : 0;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < quads; i += gridDim.x * 256) ((uint4*)d)[i] = ((const uint4*)s)[i];
    for (unsigned i = (quads << 4) + blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) d[i] = s[i];
}

// Prefetched frames as self-contained records (sd_tracker_export_prefetched / _import_prefetched: a frame's history-free half computed on one GPU,
// consumed by the GPU that owns the stream).  Record r = the segments of pool entry `first + r` back to back, each padded to 16 bytes; entry e is
// slot e * slotStep of the workspace.  toRecord = 1 packs, 0 unpacks.  grid (blocks, segments, records).
struct SdRecordTable { char* base[SD_COPY_SEGS]; unsigned slotBytes[SD_COPY_SEGS]; unsigned offset[SD_COPY_SEGS]; int n; unsigned recordBytes; };
__global__ void __launch_bounds__(256) k_frame_records(SdRecordTable T, char* __restrict__ records, int first, int slotStep, int toRecord)
{
    const int seg = blockIdx.y, r = blockIdx.z;
    const unsigned n = T.slotBytes[seg];
    char* a = T.base[seg] + (size_t)((first + r) * slotStep) * n;                  // workspace side
    char* c = records + (size_t)r * T.recordBytes + T.offset[seg];                 // record side (16-byte aligned by construction)
    const char* s = toRecord ? a : c;
    char* d = toRecord ? c : a;
    const bool a16 = ((((size_t)s) | ((size_t)d)) & 15) == 0;
    const unsigned quads = a16 ? n >> 4 : 0;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < quads; i += gridDim.x * 256) ((uint4*)d)[i] = ((const uint4*)s)[i];
    for (unsigned i = (quads << 4) + blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) d[i] = s[i];
}

__global__ void k_lane_gate(const SdFrameBoxes* __restrict__ fb, const int* __restrict__ want, int* __restrict__ active, int n, int slotStep)
{
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s < n) active[s] = (want[s] && fb[s * slotStep].nb > 0) ? 1 : 0;
}

__global__ void k_reset_boxes(SdFrameBoxes* __restrict__ fb, const int* __restrict__ count, const int* __restrict__ slots, int n)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const int slot = slots[i];
    SdFrameBoxes& F = fb[slot];
    F.nb = 0; F.nAll = count[slot]; F.nOri = count[slot]; F.nDyn = 0; F.boxStart[0] = 0;
}

__global__ void __launch_bounds__(256) k_fill_mono(const int* __restrict__ count, float* __restrict__ uRight, float* __restrict__ depth, int cap)
{
    const int img = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count[img]) return;
    uRight[(size_t)img * cap + i] = -1.f; depth[(size_t)img * cap + i] = -1.f;
}

struct SdLaneSummary {
    int N, nb, nAll, nOri, nDyn;
    int flag, nH, nF, sepRet, nTrackMatches, nTrackPairs, nLastMatches;
    int box_idx[SD_MAXB], box_status[SD_MAXB], keptOrig[SD_MAXB];
    double boxes[SD_MAXB][4];
};

__global__ void __launch_bounds__(64) k_lane_summary(const SdFrameBoxes* __restrict__ fb, const int* __restrict__ count,
                                                     const SdMotionResult* __restrict__ moRes, const int* __restrict__ sepRet,
                                                     const int* __restrict__ nmatch, const int* __restrict__ npairs, int nLanes, int slotStep,
                                                     int haveLast, SdLaneSummary* __restrict__ out)
{
    const int s = blockIdx.x, tid = threadIdx.x;
    const int slot = s * slotStep;
    const SdFrameBoxes& F = fb[slot];
    SdLaneSummary& O = out[s];
    if (tid < SD_MAXB) {
        const bool in = tid < F.nb;
        O.box_idx[tid] = in ? F.box_idx[tid] : 0; O.box_status[tid] = in ? F.box_status[tid] : 0; O.keptOrig[tid] = in ? F.keptOrig[tid] : 0;
        for (int k = 0; k < 4; k++) O.boxes[tid][k] = in ? F.boxes[tid][k] : 0.0;
    }
    if (tid == 0) {
        O.N = count[slot]; O.nb = F.nb; O.nAll = F.nAll; O.nOri = F.nOri; O.nDyn = F.nDyn;
        O.flag = moRes ? moRes[s].flag : 0; O.nH = moRes ? moRes[s].nH : 0; O.nF = moRes ? moRes[s].nF : 0;
        O.sepRet = sepRet[s]; O.nTrackMatches = nmatch[s]; O.nTrackPairs = npairs[s];
        O.nLastMatches = haveLast ? nmatch[nLanes + s] : -1;
    }
}
