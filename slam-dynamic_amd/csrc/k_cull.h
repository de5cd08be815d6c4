// HIP kernels of the dynamic-object cull (gfx950, wave64).  One workgroup per frame / frame pair: the
// box sets are tiny (<= 64 boxes, tens to hundreds of keypoints each), the batch supplies the parallelism.
//   k_box_separate   Frame::firstSeparate + ctor split      src/Frame.cc:555-604, 336-367
//   k_separate       Tracking::Separate: BFMatcher(crossCheck) per box + classifyH / classifyF + box status
//                                                            src/Tracking.cc:1093-1367
//   k_update_frame   Frame::UpdateFrame                      src/Frame.cc:607-641
#pragma once
#include "k_frame.h"

#define SD_MAXB SD_MAX_BOXES // boxes per frame the device tables hold (64: box masks are one 64-bit word per key point)
#define SD_BF_TCAP 2048     // train descriptors of one box staged in LDS at a time (larger boxes pass through in chunks)
// dynamic LDS of the two kernels whose tables are per key point (cap = key-point slots per image)
static inline size_t sd_separate_lds(int cap) { return (size_t)SD_BF_TCAP * 32 + (size_t)cap * 8 + 64; }       // descriptor chunk + colBest / rowBest
static inline size_t sd_box_separate_lds(int cap) { return (size_t)cap * 20 + 64; }                               // two 64-bit box masks + a partition code per key point

struct SdFrameBoxes {       // per frame slot, lives in HBM
    int nb;                 // boxes (after firstSeparate: the reference's `objects`)
    int nAll;               // keypoints before the split (N_s + N_d)
    int nOri;               // N_ori of UpdateFrame (== N_s)
    int nDyn;               // N_d
    double boxes[SD_MAXB][4];
    int box_idx[SD_MAXB];
    int box_status[SD_MAXB];
    int keptOrig[SD_MAXB];  // original index of every box that survived the empty-box erase
    int boxStart[SD_MAXB + 1];
    int pad;
};

struct SdCullPtrs {
    sd_keypoint* kp; uint8_t* desc; float* uright; float* depth; int* count;
    sd_keypoint* kpT; uint8_t* descT; float* urT; float* depT;      // staging rows [maxImages][cap]
    sd_keypoint* kpD; uint8_t* descD; float* urD; float* depD;      // the dynamic keypoints of a frame (mvdynKeys...), [maxImages][cap]
    const sd_keypoint* kpDUn;                                        // mvdynKeysUn (== kpD for a camera without distortion)
    SdFrameBoxes* fb; int* boxItems;                                 // [maxImages], [maxImages][itemsCap]
    int cap, itemsCap;
    int* errFlag;
};

__device__ __forceinline__ int sd_block_scan256(int v, int* wsum, int& total)
{
    // exclusive scan of one value per thread (256 threads); total returned to all
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    __syncthreads();
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wv; w++) base += wsum[w];
    total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    return base + incl - v;
}

// Input: fb[slot].nb / boxes / box_idx filled by the host (boxTrack output).  One workgroup per frame.
// `stage` (nullable): the host's records of this launch, one per workgroup, uploaded in ONE copy; the workgroup moves its record into its slot first.
__global__ void __launch_bounds__(256) k_box_separate(SdCullPtrs A, const int* __restrict__ slots, const SdFrameBoxes* __restrict__ stage)
{
    extern __shared__ __align__(16) unsigned char smem[];
    typedef unsigned long long bmask;               // bit j = the key point lies in box j (SD_MAXB <= 64)
    bmask* smask = (bmask*)smem;                    // [cap] box mask per keypoint (original order)
    bmask* dmask = smask + A.cap;                   // [cap] box mask per dynamic keypoint (new order)
    unsigned* code = (unsigned*)(dmask + A.cap);    // [cap] partition code: static rank, or 0x80000000 | dynamic rank
    __shared__ int s_wsum[4];
    __shared__ bmask s_has;
    __shared__ int s_nb2, s_empty, s_remap[SD_MAXB], s_kept[SD_MAXB], s_cnt[SD_MAXB], s_start[SD_MAXB + 1];
    __shared__ double s_box[SD_MAXB][4];
    const int slot = slots[blockIdx.x], tid = threadIdx.x;
    SdFrameBoxes& F = A.fb[slot];
    if (stage) {
        const uint32_t* src = (const uint32_t*)(stage + blockIdx.x);
        uint32_t* dst = (uint32_t*)&F;
        for (int i = tid; i < (int)(sizeof(SdFrameBoxes) / 4); i += 256) dst[i] = src[i];
        __threadfence_block();
        __syncthreads();
    }
    const int N = A.count[slot], nb = F.nb;
    const size_t base = (size_t)slot * A.cap;
    if (N == 0) {                                   // `if(mvKeys.empty()) return;` (Frame.cc:160-161, 320-321): the constructor ends BEFORE boxTrack, objects stays empty
        __syncthreads();
        if (tid == 0) { F.nb = 0; F.nAll = 0; F.nOri = 0; F.nDyn = 0; F.boxStart[0] = 0; }
        return;
    }
    if (tid == 0) s_has = 0;
    if (tid < nb) { s_box[tid][0] = F.boxes[tid][0]; s_box[tid][1] = F.boxes[tid][1]; s_box[tid][2] = F.boxes[tid][2]; s_box[tid][3] = F.boxes[tid][3]; }
    __syncthreads();
    // ---- box membership (cv::Rect2d::contains on the f32 keypoint position widened to f64)
    bmask has = 0;
    for (int i = tid; i < N; i += 256) {
        const double px = (double)A.kp[base + i].x, py = (double)A.kp[base + i].y;
        bmask m = 0;
        for (int j = 0; j < nb; j++) {
            const double x = s_box[j][0], y = s_box[j][1];
            if (x <= px && px < x + s_box[j][2] && y <= py && py < y + s_box[j][3]) m |= 1ull << j;
        }
        smask[i] = m;
        has |= m;
    }
    if (has) atomicOr(&s_has, has);
    __syncthreads();
    // ---- empty-box erase, literally (Frame.cc:585-592; hasKpts is not erased alongside)
    if (tid == 0) {
        int kept[SD_MAXB], n2 = nb;
        for (int j = 0; j < nb; j++) kept[j] = j;
        int empty = 0;
        for (int i = 0; i < n2; i++) {
            if ((s_has >> i) & 1ull) continue;
            for (int k = i; k + 1 < n2; k++) kept[k] = kept[k + 1];
            n2--;
            empty = 1;
        }
        s_nb2 = n2; s_empty = empty;
        for (int j = 0; j < n2; j++) s_kept[j] = kept[j];
        int nfalse = 0;
        for (int j = 0; j < nb; j++) {
            if (!((s_has >> j) & 1ull)) nfalse++;
            s_remap[j] = empty ? j - nfalse : j;      // index -= count(hasKpts[0..j] == false)
        }
    }
    __syncthreads();
    // ---- stable partition: static keypoints first, dynamic ones after, both in original order
    int nStaticTotal = 0;
    {
        int carry = 0;
        for (int i0 = 0; i0 < N; i0 += 256) {
            const int i = i0 + tid;
            const int isStatic = (i < N) && smask[i] == 0;
            int tot;
            const int ex = sd_block_scan256(isStatic, s_wsum, tot);
            if (i < N) {
                const int sBefore = carry + ex;                 // static keypoints before i
                // destination is final only once N_s is known: remember the static rank (or the dynamic rank, flagged)
                code[i] = isStatic ? (unsigned)sBefore : (0x80000000u | (unsigned)(i - sBefore));
            }
            carry += tot;
            __syncthreads();
        }
        nStaticTotal = carry;
    }
    const int Ns = nStaticTotal, Nd = N - Ns;
    // static keypoints -> staging rows (then back, compacted); dynamic ones -> the frame's dynamic arrays
    for (int i = tid; i < N; i += 256) {
        const unsigned cd = code[i];
        const bool dynk = (cd & 0x80000000u) != 0;
        const int dst = (int)(cd & 0x7FFFFFFFu);
        if (dynk) dmask[dst] = smask[i];                        // the masks of the dynamic keypoints in their new order
        sd_keypoint k = A.kp[base + i];
        if (dynk) k.class_id = i;                               // class_id = original index (Frame.cc:567-570)
        const uint4* ds = (const uint4*)(A.desc + (base + i) * 32);
        if (dynk) {
            A.kpD[base + dst] = k;
            uint4* dd = (uint4*)(A.descD + (base + dst) * 32);
            dd[0] = ds[0]; dd[1] = ds[1];
            A.urD[base + dst] = A.uright[base + i];
            A.depD[base + dst] = A.depth[base + i];
        } else {
            A.kpT[base + dst] = k;
            uint4* dd = (uint4*)(A.descT + (base + dst) * 32);
            dd[0] = ds[0]; dd[1] = ds[1];
            A.urT[base + dst] = A.uright[base + i];
            A.depT[base + dst] = A.depth[base + i];
        }
    }
    __syncthreads();
    // static part back to the frame's arrays: positions >= N_s are free for UpdateFrame's re-admissions
    for (int i = tid; i < Ns; i += 256) {
        A.kp[base + i] = A.kpT[base + i];
        const uint4* ds = (const uint4*)(A.descT + (base + i) * 32);
        uint4* dd = (uint4*)(A.desc + (base + i) * 32);
        dd[0] = ds[0]; dd[1] = ds[1];
        A.uright[base + i] = A.urT[base + i];
        A.depth[base + i] = A.depT[base + i];
    }
    __syncthreads();
    // ---- per-box keypoint lists (Frame.cc:347-360): thread j owns original box j
    const int nb2 = s_nb2;
    if (tid < SD_MAXB) {
        int c = 0;
        if (tid < nb && ((s_has >> tid) & 1ull))
            for (int i = 0; i < Nd; i++) c += (int)((dmask[i] >> tid) & 1ull);
        s_cnt[tid] = c;
    }
    __syncthreads();
    if (tid == 0) {
        int cntNew[SD_MAXB];
        for (int b = 0; b < SD_MAXB; b++) cntNew[b] = 0;
        for (int j = 0; j < nb; j++)
            if (((s_has >> j) & 1ull) && s_remap[j] >= 0 && s_remap[j] < nb2) cntNew[s_remap[j]] = s_cnt[j];
        int pos = 0;
        for (int b = 0; b < nb2; b++) { s_start[b] = pos; pos += cntNew[b]; }
        s_start[nb2] = pos;
        if (pos > A.itemsCap) atomicOr(A.errFlag, 16);
    }
    __syncthreads();
    if (tid < nb && ((s_has >> tid) & 1ull) && s_remap[tid] >= 0 && s_remap[tid] < nb2) {
        int* items = A.boxItems + (size_t)slot * A.itemsCap;
        int pos = s_start[s_remap[tid]];
        for (int i = 0; i < Nd; i++)
            if ((dmask[i] >> tid) & 1ull) { if (pos < A.itemsCap) items[pos] = i; pos++; }   // index into the dynamic arrays
    }
    __syncthreads();
    // ---- frame record: objects = boxes after the erase; N = N_s
    int o = 0, id = 0;
    if (tid < nb2) { o = s_kept[tid]; id = F.box_idx[o]; }
    __syncthreads();                                              // all reads of F.box_idx before the in-place rewrite
    if (tid < nb2) {
        F.boxes[tid][0] = s_box[o][0]; F.boxes[tid][1] = s_box[o][1]; F.boxes[tid][2] = s_box[o][2]; F.boxes[tid][3] = s_box[o][3];
        F.box_idx[tid] = id;
        F.box_status[tid] = -1;                                   // Frame.cc:370
        F.keptOrig[tid] = o;
    }
    if (tid <= nb2) F.boxStart[tid] = s_start[tid];
    if (tid == 0) { F.nb = nb2; F.nAll = N; F.nOri = Ns; F.nDyn = Nd; A.count[slot] = Ns; }
}

// ---------------------------------------------------------------------------------------------------
struct SdSepArgs {
    const int2* pairIdx;          // (cur slot, ref slot) per pair
    const float* HorF;            // [pairs][9]
    const int* flag;              // [pairs] 1 = H, 2 = F
    const int* lastIdx;           // [pairs][SD_MAXB] mLastFrame.box_idx
    const int* lastStatus;        // [pairs][SD_MAXB]
    const int* nLast;             // [pairs]
    int* dynStart;                // [pairs][SD_MAXB + 1]
    int* dynStatus;               // [pairs][itemsCap]
    int* matches;                 // [pairs][itemsCap][2]
    int* ret;                     // [pairs] 1 = a static box exists
    const int* lastSlot;          // nullable: [pairs] slot of mLastFrame; its box_idx / box_status are read on the device instead of lastIdx / lastStatus
    const int* active;            // nullable: [pairs] 0 = skip the pair, leave its results untouched (sd_tracker)
};

__device__ __forceinline__ void sd_inv3x3(const float* S, float* D)
{
    // cv::Mat::inv() 3x3 f32: double cofactors / determinant, narrowed to f32; singular -> zeros
    const double d0 = (double)S[0] * ((double)S[4] * S[8] - (double)S[5] * S[7]) -
                      (double)S[1] * ((double)S[3] * S[8] - (double)S[5] * S[6]) +
                      (double)S[2] * ((double)S[3] * S[7] - (double)S[4] * S[6]);
    if (d0 == 0.) { for (int i = 0; i < 9; i++) D[i] = 0.f; return; }
    const double d = 1. / d0;
    D[0] = (float)(((double)S[4] * S[8] - (double)S[5] * S[7]) * d);
    D[1] = (float)(((double)S[2] * S[7] - (double)S[1] * S[8]) * d);
    D[2] = (float)(((double)S[1] * S[5] - (double)S[2] * S[4]) * d);
    D[3] = (float)(((double)S[5] * S[6] - (double)S[3] * S[8]) * d);
    D[4] = (float)(((double)S[0] * S[8] - (double)S[2] * S[6]) * d);
    D[5] = (float)(((double)S[2] * S[3] - (double)S[0] * S[5]) * d);
    D[6] = (float)(((double)S[3] * S[7] - (double)S[4] * S[6]) * d);
    D[7] = (float)(((double)S[1] * S[6] - (double)S[0] * S[7]) * d);
    D[8] = (float)(((double)S[0] * S[4] - (double)S[1] * S[3]) * d);
}

__device__ __forceinline__ bool sd_classify_one(const float* M, const float* Mi, int flag, float u1, float v1, float u2, float v2)
{
    if (flag == 1) {
        const float th = (float)5.991;
        const float invSigmaSquare = (float)(1.0 / (1.0f * 1.0f));
        const float w2in1inv = 1.0f / (Mi[6] * u2 + Mi[7] * v2 + Mi[8]);
        const float u2in1 = (Mi[0] * u2 + Mi[1] * v2 + Mi[2]) * w2in1inv;
        const float v2in1 = (Mi[3] * u2 + Mi[4] * v2 + Mi[5]) * w2in1inv;
        const float squareDist1 = (u1 - u2in1) * (u1 - u2in1) + (v1 - v2in1) * (v1 - v2in1);
        const float chiSquare1 = squareDist1 * invSigmaSquare;
        const float w1in2inv = 1.0f / (M[6] * u1 + M[7] * v1 + M[8]);
        const float u1in2 = (M[0] * u1 + M[1] * v1 + M[2]) * w1in2inv;
        const float v1in2 = (M[3] * u1 + M[4] * v1 + M[5]) * w1in2inv;
        const float squareDist2 = (u2 - u1in2) * (u2 - u1in2) + (v2 - v1in2) * (v2 - v1in2);
        const float chiSquare2 = squareDist2 * invSigmaSquare;
        return chiSquare2 <= th && chiSquare1 <= th;
    } else {
        const float th_F = (float)5.841;
        const float invSigmaSquare = (float)(1.0 / (1.0f * 1.0f));
        const float a2 = M[0] * u1 + M[1] * v1 + M[2];
        const float b2 = M[3] * u1 + M[4] * v1 + M[5];
        const float c2 = M[6] * u1 + M[7] * v1 + M[8];
        const float num2 = a2 * u2 + b2 * v2 + c2;
        const float squareDist1 = num2 * num2 / (a2 * a2 + b2 * b2);
        const float chiSquare1 = squareDist1 * invSigmaSquare;
        const float a1 = M[0] * u2 + M[3] * v2 + M[6];
        const float b1 = M[1] * u2 + M[4] * v2 + M[7];
        const float c1 = M[2] * u2 + M[5] * v2 + M[8];
        const float num1 = a1 * u1 + b1 * v1 + c1;
        const float squareDist2 = num1 * num1 / (a1 * a1 + b1 * b1);
        const float chiSquare2 = squareDist2 * invSigmaSquare;
        return chiSquare1 <= th_F && chiSquare2 <= th_F;
    }
}

// One workgroup per (current, reference) pair; boxes in sequence.  Cross-checked brute force: thread = query,
// train descriptors in LDS; the column minima (nearest query of every train) are LDS atomicMin on
// (distance << 16 | query), which is exactly "first nearest wins".
// HorF / flag of pair p from the model fit of the same pair index (sd_batch_estimate_motion -> sd_batch_separate)
__global__ void k_motion_to_sep(const SdMotionResult* __restrict__ res, float* __restrict__ HorF, int* __restrict__ flag, int n,
                                const int* __restrict__ active)
{
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= n || (active && !active[p])) return;
    for (int k = 0; k < 9; k++) HorF[(size_t)p * 9 + k] = res[p].HorF[k];
    flag[p] = res[p].flag;
}

__global__ void __launch_bounds__(256) k_separate(SdCullPtrs A, SdSepArgs G)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint4* tdesc = (uint4*)smem;                                   // [SD_BF_TCAP][2]: one CHUNK of the train descriptors
    unsigned* colBest = (unsigned*)(tdesc + 2 * SD_BF_TCAP);       // [cap] nearest query of every train descriptor
    unsigned* rowBest = colBest + A.cap;                           // [cap] nearest train descriptor of every query
    __shared__ int s_wsum[4];
    __shared__ float s_M[9], s_Mi[9];
    __shared__ int s_num0, s_static;
    const int pair = blockIdx.x, tid = threadIdx.x;
    if (G.active && !G.active[pair]) return;
    const int cs = G.pairIdx[pair].x, rs = G.pairIdx[pair].y;
    SdFrameBoxes& FC = A.fb[cs];
    const SdFrameBoxes& FR = A.fb[rs];
    const int flag = G.flag[pair];
    if (tid < 9) s_M[tid] = G.HorF[(size_t)pair * 9 + tid];
    if (tid == 0) s_static = 0;
    __syncthreads();
    if (tid == 0) sd_inv3x3(s_M, s_Mi);
    __syncthreads();
    const int* itemsC = A.boxItems + (size_t)cs * A.itemsCap;
    const int* itemsR = A.boxItems + (size_t)rs * A.itemsCap;
    const size_t baseC = (size_t)cs * A.cap, baseR = (size_t)rs * A.cap;
    int* dynStart = G.dynStart + (size_t)pair * (SD_MAXB + 1);
    int* dyn = G.dynStatus + (size_t)pair * A.itemsCap;
    int* mt = G.matches + (size_t)pair * A.itemsCap * 2;
    int pos = 0;
    const int nbC = FC.nb, nbR = FR.nb;
    if (flag != 1 && flag != 2) {        // TrackHomo returned 0 (taken from sd_batch_estimate_motion): Separate is not called (Tracking.cc:637)
        if (tid == 0) { for (int b = 0; b <= SD_MAXB; b++) dynStart[b] = 0; G.ret[pair] = 0; }
        return;
    }
    for (int nbx = 0; nbx < nbC; nbx++) {
        if (tid == 0) dynStart[nbx] = pos;
        const int id = FC.box_idx[nbx];
        int ref = -1;
        for (int j = 0; j < nbR; j++) if (FR.box_idx[j] == id) { ref = j; break; }
        if (ref < 0) continue;
        const int q0 = FC.boxStart[nbx], nq = FC.boxStart[nbx + 1] - q0;
        const int t0 = FR.boxStart[ref], nt = FR.boxStart[ref + 1] - t0;
        if (nq == 0 || nt == 0) continue;
        if (nq > A.cap || nt > A.cap) { if (tid == 0) atomicOr(A.errFlag, 32); continue; }      // cannot happen: a box's lists are subsets of a frame's keypoints
        __syncthreads();
        for (int i = tid; i < nq; i += 256) rowBest[i] = 0xFFFFFFFFu;
        // the train descriptors pass through LDS in chunks of SD_BF_TCAP (a box of a 2000-feature frame is one chunk); minima of
        // (distance << 16 | index) are order-free, so "first nearest wins" holds across chunks as it does across lanes
        for (int c0 = 0; c0 < nt; c0 += SD_BF_TCAP) {
            const int nc = min(SD_BF_TCAP, nt - c0);
            for (int j = tid; j < nc; j += 256) {
                const uint4* d = (const uint4*)(A.descD + (baseR + itemsR[t0 + c0 + j]) * 32);
                tdesc[2 * j] = d[0]; tdesc[2 * j + 1] = d[1];
                colBest[c0 + j] = 0xFFFFFFFFu;
            }
            __syncthreads();
            for (int i = tid; i < nq; i += 256) {
                const uint4* d = (const uint4*)(A.descD + (baseC + itemsC[q0 + i]) * 32);
                const uint4 a0 = d[0], a1 = d[1];
                unsigned best = rowBest[i];
                int j = tid % nc;                                 // staggered start: lanes hit different trains
                for (int s = 0; s < nc; s++) {
                    const unsigned dist = (unsigned)sd_hamming256(a0, a1, tdesc[2 * j], tdesc[2 * j + 1]);
                    best = min(best, (dist << 16) | (unsigned)(c0 + j));
                    atomicMin(&colBest[c0 + j], (dist << 16) | (unsigned)i);
                    j++; if (j == nc) j = 0;
                }
                rowBest[i] = best;
            }
            __syncthreads();
        }
        // cross-check + ordered compaction by query index
        int carry = 0;
        for (int i0 = 0; i0 < nq; i0 += 256) {
            const int i = i0 + tid;
            int ok = 0, jj = 0;
            if (i < nq) { jj = (int)(rowBest[i] & 0xFFFFu); ok = (int)(colBest[jj] & 0xFFFFu) == i; }
            int tot;
            const int ex = sd_block_scan256(ok, s_wsum, tot);
            if (ok) { mt[2 * (pos + carry + ex)] = i; mt[2 * (pos + carry + ex) + 1] = jj; }
            carry += tot;
            __syncthreads();
        }
        const int ng = carry;
        if (ng < 3 || (double)ng < 0.2 * (double)nq) continue;      // Tracking.cc:1125
        if (tid == 0) s_num0 = 0;
        __syncthreads();
        int mine = 0;
        for (int m = tid; m < ng; m += 256) {
            const int qi = mt[2 * (pos + m)], ti = mt[2 * (pos + m) + 1];
            const sd_keypoint kc = A.kpDUn[baseC + itemsC[q0 + qi]], kr = A.kpDUn[baseR + itemsR[t0 + ti]];      // classifyH / classifyF run on mvdynKeysUn (Tracking.cc:1131-1133)
            const bool st = sd_classify_one(s_M, s_Mi, flag, kr.x, kr.y, kc.x, kc.y);
            dyn[pos + m] = st ? qi : -1;
            mine += st;
        }
        if (mine) atomicAdd(&s_num0, mine);
        __syncthreads();
        if (tid == 0) {
            const int num0 = s_num0;
            const double lim = 0.2 * (double)ng;
            if ((double)num0 > (lim > 1.0 ? lim : 1.0)) {
                s_static = 1;                                      // `box_status[n_box] == 1;` is a no-op (:1188)
            } else {
                int ls = -1;
                if (G.lastSlot) {                                  // mLastFrame lives in a slot of this batch
                    const SdFrameBoxes& FL = A.fb[G.lastSlot[pair]];
                    for (int j = 0; j < FL.nb; j++) if (FL.box_idx[j] == id) { ls = FL.box_status[j]; break; }
                } else {
                    const int nl = G.nLast[pair];
                    for (int j = 0; j < nl; j++) if (G.lastIdx[(size_t)pair * SD_MAXB + j] == id) { ls = G.lastStatus[(size_t)pair * SD_MAXB + j]; break; }
                }
                FC.box_status[nbx] = (ls == 0 || ls == 2) ? 2 : 0;
            }
        }
        pos += ng;
        __syncthreads();
    }
    if (tid == 0) { for (int b = nbC; b <= SD_MAXB; b++) dynStart[b] = pos; G.ret[pair] = s_static; }
}

// Frame::UpdateFrame: append the re-admitted keypoints (class_id de-duplicated, push_back order) behind the
// static ones and bump N.  One workgroup per frame; the short in-order walk is done by one thread.
__global__ void __launch_bounds__(256) k_update_frame(SdCullPtrs A, SdSepArgs G, const int* __restrict__ doUpdate)
{
    extern __shared__ __align__(16) unsigned char smem[];
    int* list = (int*)smem;                        // [itemsCap] source positions k
    uint8_t* seen = (uint8_t*)(list + A.itemsCap); // [cap] by class_id (= original index)
    __shared__ int s_n;
    const int pair = blockIdx.x, tid = threadIdx.x;
    if (G.active && !G.active[pair]) return;
    if (doUpdate && !doUpdate[pair]) return;
    const int cs = G.pairIdx[pair].x;
    SdFrameBoxes& F = A.fb[cs];
    const size_t base = (size_t)cs * A.cap;
    const int* items = A.boxItems + (size_t)cs * A.itemsCap;
    const int* dynStart = G.dynStart + (size_t)pair * (SD_MAXB + 1);
    const int* dyn = G.dynStatus + (size_t)pair * A.itemsCap;
    for (int i = tid; i < A.cap; i += 256) seen[i] = 0;
    __syncthreads();
    if (tid == 0) {
        int n = 0;
        for (int b = 0; b < F.nb; b++) {
            const int s0 = dynStart[b], e0 = dynStart[b + 1];
            for (int m = s0; m < e0; m++) {
                const int q = dyn[m];
                if (q == -1) continue;
                const int k = items[F.boxStart[b] + q];
                const int cid = A.kpD[base + k].class_id;
                if (cid < 0 || cid >= A.cap || seen[cid]) continue;
                seen[cid] = 1;
                list[n++] = k;
            }
        }
        s_n = n;
    }
    __syncthreads();
    const int n = s_n, Ns = A.count[cs];
    for (int r = tid; r < n; r += 256) {
        const int k = list[r];
        A.kp[base + Ns + r] = A.kpD[base + k];
        const uint4* ds = (const uint4*)(A.descD + (base + k) * 32);
        uint4* dd = (uint4*)(A.desc + (base + Ns + r) * 32);
        dd[0] = ds[0]; dd[1] = ds[1];
        A.uright[base + Ns + r] = A.urD[base + k];
        A.depth[base + Ns + r] = A.depD[base + k];
    }
    if (tid == 0) { F.nOri = Ns; A.count[cs] = Ns + n; }
}
