// Bag-of-words kernels: the DBoW2 vocabulary tree of the reference (vendored, Thirdparty/DBoW2) and
// ORBmatcher::SearchByBoW.
//   k_bow_transform     TemplatedVocabulary::transform(feature, word, weight, nid, levelsup)   TemplatedVocabulary.h:1218-1259
//   k_bow_finalize      ... transform(features, BowVector, FeatureVector, levelsup)           TemplatedVocabulary.h:1127-1203
//                       BowVector::addWeight / addIfNotExist / normalize                      BowVector.cpp:35-85
//                       FeatureVector::addFeature                                             FeatureVector.cpp:30-46
//   k_search_by_bow     ORBmatcher::SearchByBoW(KeyFrame*, Frame&, matches)                   src/ORBmatcher.cc:159-288
// The vocabulary is ONE packed buffer (the object bench.py broadcasts over RCCL at start-up):
//   SdVocabHeader | desc[n][32] u8 | weight[n] f64 | parent[n] i32 | childStart[n+1] i32 | childIdx[n-1] i32 | wordId[n] i32
// Node ids are the reference's (line number of the text file, root = 0); children keep file order, which decides ties.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct SdVocabHeader {
    uint32_t magic, version, k, L, scoring, weighting, nNodes, nWords;
    uint64_t offDesc, offWeight, offParent, offChildStart, offChildIdx, offWordId, totalBytes;
    uint64_t pad[5];
};                                     // 128 bytes
#define SD_VOCAB_MAGIC 0x42564453u     // "SDVB"

struct SdVocabDev {                    // device pointers into the packed buffer
    const uint8_t* desc; const double* weight; const int* childStart; const int* childIdx; const int* wordId;
    int L, scoring, weighting, nNodes;
};

// 16 lanes per feature: lane j scores child j (k <= 20 children: at most two rounds), a 16-lane butterfly keeps the
// first minimum in child order (the reference's strict `d < best_d`).
__global__ void __launch_bounds__(256) k_bow_transform(const uint8_t* __restrict__ desc, const int* __restrict__ count,
                                                       const int* __restrict__ imgOf, SdVocabDev V, int levelsup, int cap,
                                                       unsigned* __restrict__ wordOut, double* __restrict__ weightOut,
                                                       unsigned* __restrict__ nidOut)
{
    const int img = imgOf[blockIdx.y];
    const int sub = threadIdx.x & 15;
    const int i = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int N = count[img];
    const bool act = i < N;                       // whole 16-lane groups go inactive together
    const size_t o = (size_t)img * cap + (act ? i : 0);
    const uint4* df = (const uint4*)(desc + o * 32);
    const uint4 f0 = df[0], f1 = df[1];
    const int nid_level = V.L - levelsup;
    unsigned node = 0, nid = 0;
    bool nidSet = nid_level <= 0;
    int level = 0;
    while (true) {
        const int cs = V.childStart[node], ce = V.childStart[node + 1];
        if (cs == ce) break;                      // leaf
        level++;
        unsigned best = 0xFFFFFFFFu;
        for (int c0 = cs; c0 < ce; c0 += 16) {
            const int idx = c0 + sub;
            unsigned key = 0xFFFFFFFFu;
            if (idx < ce) {
                const unsigned child = (unsigned)V.childIdx[idx];
                const uint4* dn = (const uint4*)(V.desc + (size_t)child * 32);
                key = ((unsigned)sd_hamming256(f0, f1, dn[0], dn[1]) << 16) | (unsigned)(idx - cs);
            }
            best = key < best ? key : best;
        }
#pragma unroll
        for (int d = 8; d > 0; d >>= 1) { const unsigned o2 = (unsigned)__shfl_xor((int)best, d, 16); best = o2 < best ? o2 : best; }
        node = (unsigned)V.childIdx[cs + (int)(best & 0xFFFFu)];
        if (level == nid_level) { nid = node; nidSet = true; }
    }
    if (!nidSet) nid = node;                      // spec Q12: the reference leaves *nid unwritten here
    if (act && sub == 0) { wordOut[o] = (unsigned)V.wordId[node]; weightOut[o] = V.weight[node]; nidOut[o] = nid; }
}

// One workgroup per image.  FeatureVector = the features with weight > 0 sorted by (node id, feature index), + its runs;
// BowVector = the distinct words ascending; a word seen c times holds ((w + w) + ...) summed c times in feature order (all
// terms are the word's own weight) for TF / TF-IDF, w for IDF / BINARY; then the reference's normalisation, summed in
// ascending word order by one lane (double additions do not commute bit-for-bit).
// exclusive scan of s[0..256) in place, total in s[256]; called by all 256 threads between two barriers of the caller
__device__ __forceinline__ void sd_scan256(int* s, int tid)
{
    if (tid < 64) {
        const int a = s[4 * tid], b = s[4 * tid + 1], c = s[4 * tid + 2], d = s[4 * tid + 3];
        const int sum = a + b + c + d;
        int incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (tid >= o) incl += t; }
        const int ex = incl - sum;
        s[4 * tid] = ex; s[4 * tid + 1] = ex + a; s[4 * tid + 2] = ex + a + b; s[4 * tid + 3] = ex + a + b + c;
        if (tid == 63) s[256] = incl;
    }
}

__global__ void __launch_bounds__(256) k_bow_finalize(const int* __restrict__ count, const int* __restrict__ imgOf,
                                                      const unsigned* __restrict__ word, const double* __restrict__ weight,
                                                      const unsigned* __restrict__ nid, int cap, int sortCap, int scoring, int weighting,
                                                      unsigned* __restrict__ fvNode, unsigned* __restrict__ fvFeat,
                                                      int* __restrict__ fvRunStart, unsigned* __restrict__ fvRunNode,
                                                      unsigned* __restrict__ bowWord, double* __restrict__ bowVal,
                                                      int* __restrict__ meta /*[img][4]: nf, nRuns, nb, -*/)
{
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned long long* keys = (unsigned long long*)smem;       // [sortCap]
    int* s_scan = (int*)(keys + sortCap);                        // [256 + 1]
    __shared__ double s_norm;
    const int img = imgOf[blockIdx.x], tid = threadIdx.x;
    const int N = count[img];
    int sortN = 256;                                             // this image's sort size: the power of two above its keypoint count
    while (sortN < N) sortN <<= 1;
    const size_t base = (size_t)img * cap;
    // ---------------- FeatureVector
    for (int t = tid; t < sortN; t += 256) {
        unsigned long long key = ~0ull;
        if (t < N && weight[base + t] > 0) key = ((unsigned long long)nid[base + t] << 32) | (unsigned)t;
        keys[t] = key;
    }
    __syncthreads();
    sd_block_sort64(keys, sortN, tid, 256);
    // nf = number of real keys; runs = distinct node ids.  Chunked scan: thread t owns items [t*per, (t+1)*per)
    const int per = sortN / 256 > 0 ? sortN / 256 : 1;
    {
        int nfLocal = 0, runsLocal = 0;
        for (int q = 0; q < per; q++) {
            const int t = tid * per + q;
            if (t < sortN && keys[t] != ~0ull) {
                nfLocal++;
                if (t == 0 || (keys[t] >> 32) != (keys[t - 1] >> 32)) runsLocal++;
            }
        }
        s_scan[tid] = runsLocal;
        __syncthreads();
        sd_scan256(s_scan, tid);
        __syncthreads();
        int r = s_scan[tid];
        for (int q = 0; q < per; q++) {
            const int t = tid * per + q;
            if (t < sortN && keys[t] != ~0ull) {
                fvNode[base + t] = (unsigned)(keys[t] >> 32); fvFeat[base + t] = (unsigned)(keys[t] & 0xFFFFFFFFu);
                if (t == 0 || (keys[t] >> 32) != (keys[t - 1] >> 32)) { fvRunStart[(size_t)img * (cap + 1) + r] = t; fvRunNode[base + r] = (unsigned)(keys[t] >> 32); r++; }
            }
        }
        const int nRuns = s_scan[256];
        __syncthreads();
        s_scan[tid] = nfLocal;
        __syncthreads();
        if (tid == 0) {
            int nf = 0; for (int k = 0; k < 256; k++) nf += s_scan[k];
            meta[img * 4 + 0] = nf; meta[img * 4 + 1] = nRuns;
            fvRunStart[(size_t)img * (cap + 1) + nRuns] = nf;
        }
        __syncthreads();
    }
    // ---------------- BowVector
    for (int t = tid; t < sortN; t += 256) {
        unsigned long long key = ~0ull;
        if (t < N && weight[base + t] > 0) key = ((unsigned long long)word[base + t] << 32) | (unsigned)t;
        keys[t] = key;
    }
    __syncthreads();
    sd_block_sort64(keys, sortN, tid, 256);
    {
        int local = 0;
        for (int q = 0; q < per; q++) {
            const int t = tid * per + q;
            if (t < sortN && keys[t] != ~0ull && (t == 0 || (keys[t] >> 32) != (keys[t - 1] >> 32))) local++;
        }
        s_scan[tid] = local;
        __syncthreads();
        sd_scan256(s_scan, tid);
        __syncthreads();
        int r = s_scan[tid];
        const int nb = s_scan[256];
        for (int q = 0; q < per; q++) {
            const int t = tid * per + q;
            if (t < sortN && keys[t] != ~0ull && (t == 0 || (keys[t] >> 32) != (keys[t - 1] >> 32))) {
                const unsigned w = (unsigned)(keys[t] >> 32);
                int c = 1;
                while (t + c < sortN && (unsigned)(keys[t + c] >> 32) == w) c++;
                const double wv = weight[base + (unsigned)(keys[t] & 0xFFFFFFFFu)];
                double v = wv;
                if (weighting == 0 || weighting == 1) for (int a = 1; a < c; a++) v += wv;       // addWeight, once per feature
                bowWord[base + r] = w; bowVal[base + r] = v;
                r++;
            }
        }
        __threadfence_block();
        __syncthreads();
        double* vals = (double*)keys;                          // the sorted keys are dead: stage the values for the serial sum
        for (int t = tid; t < nb; t += 256) vals[t] = bowVal[base + t];
        __syncthreads();
        const bool must = scoring != 5;
        if ((weighting == 0 || weighting == 1) && nb > 0 && !must) {
            const double nd = (double)nb;
            for (int t = tid; t < nb; t += 256) vals[t] /= nd;
        }
        __syncthreads();
        if (must) {
            if (tid < 64) {
                // The sum must run in word order (f64 rounding).  Wave 0 fetches 64 values per step (one per lane) and adds them
                // in lane order through v_readlane: ~16 cycles per element instead of an LDS round trip per element.
                double norm = 0.0;
                for (int t0 = 0; t0 < nb; t0 += 64) {
                    const double v = t0 + tid < nb ? vals[t0 + tid] : 0.0;      // +0.0 / 0*0 leave the sum unchanged
                    const double e = scoring != 1 ? fabs(v) : v * v;
                    const unsigned long long bits = __builtin_bit_cast(unsigned long long, e);
                    const int lo = (int)(unsigned)bits, hi = (int)(unsigned)(bits >> 32);
#pragma unroll
                    for (int l = 0; l < 64; l++) {
                        const unsigned long long b = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(hi, l) << 32) | (unsigned)__builtin_amdgcn_readlane(lo, l);
                        norm += __builtin_bit_cast(double, b);
                    }
                }
                if (scoring == 1) norm = sqrt(norm);
                if (tid == 0) s_norm = norm;
            }
            __syncthreads();
            const double norm = s_norm;
            if (norm > 0.0) for (int t = tid; t < nb; t += 256) vals[t] /= norm;
        }
        for (int t = tid; t < nb; t += 256) bowVal[base + t] = vals[t];
        if (tid == 0) meta[img * 4 + 2] = nb;
    }
}

#define SD_BOW_REGF 128          // frame features of a node that the register path holds (two per lane)
// One workgroup per (keyframe, frame) pair; a wave takes a vocabulary node both FeatureVectors share.  A frame feature
// belongs to exactly one node, so waves never contend for a keypoint; inside a node the keyframe features are walked in
// order (the order that decides which frame keypoints are already taken, ORBmatcher.cc:206-207).  Best / second best over
// the not-yet-taken frame features = first and second entry of the (distance, position) order (see k_local_resolve).
#define SD_BOW_WAVES 16
__global__ void __launch_bounds__(64 * SD_BOW_WAVES) k_search_by_bow(
    const sd_keypoint* __restrict__ kp, const uint8_t* __restrict__ desc, const int* __restrict__ count,
    const unsigned* __restrict__ fvFeat, const int* __restrict__ fvRunStart, const unsigned* __restrict__ fvRunNode,
    const int* __restrict__ meta, const uint8_t* __restrict__ kfValid /*nullable [pair][cap]*/, const int2* __restrict__ pairIdx,
    int cap, float nnratio, int checkOrientation, int* __restrict__ matchOut, int* __restrict__ nmatchOut)
{
    extern __shared__ __align__(16) unsigned char smem[];
    int* s_match = (int*)smem;                               // [cap]
    uint8_t* s_bin = (uint8_t*)(s_match + cap);              // [cap]
    __shared__ int s_hist[SD_HISTO];
    __shared__ int s_ind[3];
    __shared__ int s_nm;
    const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int imgK = pairIdx[pair].x, imgF = pairIdx[pair].y;
    const int NF = count[imgF];
    for (int i = tid; i < NF; i += 64 * SD_BOW_WAVES) s_match[i] = -1;
    if (tid < SD_HISTO) s_hist[tid] = 0;
    if (tid == 0) s_nm = 0;
    __syncthreads();
    const int runsK = meta[imgK * 4 + 1], runsF = meta[imgF * 4 + 1];
    const int* rsK = fvRunStart + (size_t)imgK * (cap + 1);
    const int* rsF = fvRunStart + (size_t)imgF * (cap + 1);
    const unsigned* rnK = fvRunNode + (size_t)imgK * cap;
    const unsigned* rnF = fvRunNode + (size_t)imgF * cap;
    const unsigned* ffK = fvFeat + (size_t)imgK * cap;
    const unsigned* ffF = fvFeat + (size_t)imgF * cap;
    const uint8_t* dK = desc + (size_t)imgK * cap * 32;
    const uint8_t* dF = desc + (size_t)imgF * cap * 32;
    const sd_keypoint* kK = kp + (size_t)imgK * cap;
    const sd_keypoint* kF = kp + (size_t)imgF * cap;
    const float factor = 1.0f / SD_HISTO;
    int nm = 0;
    // ---- nodes with up to 128 frame features (the usual case: ~20 x 20 features at level L - 4): the wave keeps the node's
    // frame descriptors in registers (lane l owns positions l and l + 64, and with them their taken flags), gathers the
    // keyframe descriptors 64 at a time and broadcasts one per step by shuffles: no memory access inside the serial loop.
    for (int rk = wv; rk < runsK; rk += SD_BOW_WAVES) {
        const unsigned node = rnK[rk];
        int lo = 0, hi = runsF;                               // lower_bound of node in the frame's runs
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (rnF[mid] < node) lo = mid + 1; else hi = mid; }
        if (lo >= runsF || rnF[lo] != node) continue;
        const int a0 = rsK[rk], a1 = rsK[rk + 1], c0 = rsF[lo], c1 = rsF[lo + 1];
        if (c1 - c0 > SD_BOW_REGF) continue;                  // left to the generic path below
        unsigned iFl[2]; uint4 f0[2], f1[2]; bool freeF[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int c = c0 + lane + 64 * q;
            freeF[q] = c < c1;
            iFl[q] = freeF[q] ? ffF[c] : 0u;
            const uint4* pf = (const uint4*)(dF + (size_t)iFl[q] * 32);
            f0[q] = pf[0]; f1[q] = pf[1];
        }
        for (int kb = a0; kb < a1; kb += 64) {
            const int a = kb + lane;
            const unsigned iKl = a < a1 ? ffK[a] : 0u;
            const int validK = a < a1 && (!kfValid || kfValid[(size_t)pair * cap + iKl]);
            const uint4* pk = (const uint4*)(dK + (size_t)iKl * 32);
            const uint4 kl0 = pk[0], kl1 = pk[1];
            const float angK = kK[iKl].angle;
            const int nstep = a1 - kb < 64 ? a1 - kb : 64;
            for (int t = 0; t < nstep; t++) {
                if (!__builtin_amdgcn_readlane(validK, t)) continue;
                uint4 k0, k1;                                  // t is wave-uniform: v_readlane, no LDS crossbar
                k0.x = (unsigned)__builtin_amdgcn_readlane((int)kl0.x, t); k0.y = (unsigned)__builtin_amdgcn_readlane((int)kl0.y, t);
                k0.z = (unsigned)__builtin_amdgcn_readlane((int)kl0.z, t); k0.w = (unsigned)__builtin_amdgcn_readlane((int)kl0.w, t);
                k1.x = (unsigned)__builtin_amdgcn_readlane((int)kl1.x, t); k1.y = (unsigned)__builtin_amdgcn_readlane((int)kl1.y, t);
                k1.z = (unsigned)__builtin_amdgcn_readlane((int)kl1.z, t); k1.w = (unsigned)__builtin_amdgcn_readlane((int)kl1.w, t);
                unsigned key[2];
#pragma unroll
                for (int q = 0; q < 2; q++)
                    key[q] = freeF[q] ? (((unsigned)sd_hamming256(k0, k1, f0[q], f1[q]) << 16) | (unsigned)(lane + 64 * q)) : 0xFFFFFFFFu;
                // one butterfly for the smallest and the second smallest key (keys are unique: position in the low bits)
                unsigned b1 = key[0] < key[1] ? key[0] : key[1];
                unsigned b2 = key[0] < key[1] ? key[1] : key[0];
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) {
                    const unsigned o1 = (unsigned)__shfl_xor((int)b1, d, 64), o2 = (unsigned)__shfl_xor((int)b2, d, 64);
                    const unsigned hi1 = b1 > o1 ? b1 : o1, lo2 = b2 < o2 ? b2 : o2;
                    b1 = b1 < o1 ? b1 : o1;
                    b2 = hi1 < lo2 ? hi1 : lo2;
                }
                if (b1 == 0xFFFFFFFFu) continue;
                const int bestDist1 = (int)(b1 >> 16), bestDist2 = b2 == 0xFFFFFFFFu ? 256 : (int)(b2 >> 16);
                if (bestDist1 <= SD_TH_LOW && (float)bestDist1 < nnratio * (float)bestDist2) {
                    const int pos = (int)(b1 & 0xFFFFu);
                    const unsigned iKt = (unsigned)__builtin_amdgcn_readlane((int)iKl, t);
                    const float aK = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(angK), t));
                    if ((pos & 63) == lane) {
                        const int q = pos >> 6;
                        const unsigned iF = q ? iFl[1] : iFl[0];
                        if (q) freeF[1] = false; else freeF[0] = false;
                        s_match[iF] = (int)iKt;
                        int bin = 0;
                        if (checkOrientation) {
                            float rot = aK - kF[iF].angle;
                            if (rot < 0.0f) rot += 360.0f;
                            bin = (int)roundf(rot * factor);
                            if (bin == SD_HISTO) bin = 0;
                            atomicAdd(&s_hist[bin], 1);
                        }
                        s_bin[iF] = (uint8_t)bin;
                    }
                    nm++;
                }
            }
        }
    }
    // ---- nodes with more frame features than the registers hold: a wave per node, lanes over the frame's features, from memory
    for (int rk = wv; rk < runsK; rk += SD_BOW_WAVES) {
        const unsigned node = rnK[rk];
        int lo = 0, hi = runsF;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (rnF[mid] < node) lo = mid + 1; else hi = mid; }
        if (lo >= runsF || rnF[lo] != node) continue;
        const int a0 = rsK[rk], a1 = rsK[rk + 1], c0 = rsF[lo], c1 = rsF[lo + 1];
        if (c1 - c0 <= SD_BOW_REGF) continue;
        for (int a = a0; a < a1; a++) {
            const unsigned iK = ffK[a];
            if (kfValid && !kfValid[(size_t)pair * cap + iK]) continue;
            const uint4* pk = (const uint4*)(dK + (size_t)iK * 32);
            const uint4 k0 = pk[0], k1 = pk[1];
            unsigned best1 = 0xFFFFFFFFu;                     // dist << 16 | position in the node's frame list
            for (int c = c0 + lane; c < c1; c += 64) {
                const unsigned iF = ffF[c];
                if (s_match[iF] >= 0) continue;
                const uint4* pf = (const uint4*)(dF + (size_t)iF * 32);
                const unsigned key = ((unsigned)sd_hamming256(k0, k1, pf[0], pf[1]) << 16) | (unsigned)(c - c0);
                best1 = key < best1 ? key : best1;
            }
            unsigned b1 = best1;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) { const unsigned o = (unsigned)__shfl_xor((int)b1, d, 64); b1 = o < b1 ? o : b1; }
            if (b1 == 0xFFFFFFFFu) continue;                  // no free frame feature in this node
            unsigned best2 = 0xFFFFFFFFu;                     // the smallest key other than the winner
            for (int c = c0 + lane; c < c1; c += 64) {
                const unsigned iF = ffF[c];
                if (s_match[iF] >= 0 || (unsigned)(c - c0) == (b1 & 0xFFFFu)) continue;
                const uint4* pf = (const uint4*)(dF + (size_t)iF * 32);
                const unsigned key = ((unsigned)sd_hamming256(k0, k1, pf[0], pf[1]) << 16) | (unsigned)(c - c0);
                best2 = key < best2 ? key : best2;
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) { const unsigned o = (unsigned)__shfl_xor((int)best2, d, 64); best2 = o < best2 ? o : best2; }
            const int bestDist1 = (int)(b1 >> 16), bestDist2 = best2 == 0xFFFFFFFFu ? 256 : (int)(best2 >> 16);
            if (bestDist1 <= SD_TH_LOW && (float)bestDist1 < nnratio * (float)bestDist2) {
                const unsigned iF = ffF[c0 + (int)(b1 & 0xFFFFu)];
                if (lane == 0) {
                    s_match[iF] = (int)iK;
                    int bin = 0;
                    if (checkOrientation) {
                        float rot = kK[iK].angle - kF[iF].angle;
                        if (rot < 0.0f) rot += 360.0f;
                        bin = (int)roundf(rot * factor);
                        if (bin == SD_HISTO) bin = 0;
                        atomicAdd(&s_hist[bin], 1);
                    }
                    s_bin[iF] = (uint8_t)bin;
                }
                nm++;
                __builtin_amdgcn_wave_barrier();
            }
            __threadfence_block();                            // s_match written by lane 0 is read by every lane next round
        }
    }
    if (lane == 0) atomicAdd(&s_nm, nm);
    __syncthreads();
    if (checkOrientation) {
        if (tid == 0) {
            int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;     // ComputeThreeMaxima (ORBmatcher.cc:1758-1799)
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
        int culled = 0;
        for (int i = tid; i < NF; i += 64 * SD_BOW_WAVES)
            if (s_match[i] >= 0) { const int b = s_bin[i]; if (b != i1 && b != i2 && b != i3) { s_match[i] = -1; culled++; } }
        if (culled) atomicSub(&s_nm, culled);
        __syncthreads();
    }
    for (int i = tid; i < NF; i += 64 * SD_BOW_WAVES) matchOut[(size_t)pair * cap + i] = s_match[i];
    if (tid == 0) nmatchOut[pair] = s_nm;
}
