// The model fit of Tracking::TrackHomo (src/Tracking.cc:1026-1075) on the device: homography and fundamental matrix
// from the background matches of SearchByProjection, inlier masks and the reference's choice between the two.
// cv::findHomography / cv::findFundamentalMat cannot be matched bit for bit without OpenCV; the algorithm is this
// build's spec: Hartley normalisation as Initializer::Normalize (src/Initializer.cc:749-795), up to 512 (H) and
// 1024 (F) independent hypotheses from a counter-based sampler, each counted over ALL pairs by one thread, OpenCV's inlier
// criteria (3 px), most inliers wins (lowest index on ties), H refitted on its inliers by the normalised DLT of
// Initializer::ComputeH21 (:246-272).  Like OpenCV's RANSAC the search stops once the best model's inlier ratio w makes more
// samples pointless -- (1 - w^m)^k <= 1 - confidence -- but the test is made at ONE fixed checkpoint per model (after 64 H /
// 128 F hypotheses, as count >= 0.53 N / 0.66 N: confidence 0.995 / 0.99, no transcendental at run time), so that the result
// does not depend on the order in which the parallel hypotheses finish: the winner is the best of the first 64 / 128 when the
// test holds and the best of all 512 / 1024 otherwise.  All arithmetic is f64 in a fixed order (no contraction), so the CPU
// oracle's independent restatement gives the same bits.
//   k_motion_prepare   gather the point pairs, normalisation parameters (serial f64 sums in index order)
//   k_motion_models    one thread per hypothesis: sample -> minimal solver (8 x 9 elimination in registers) -> model
//   k_motion_count     inlier counts, one point pair per lane.  Both run as stage 0 (the hypotheses up to the checkpoint) and
//                      stage 1 (the rest, skipped per pair and model -- decided by the wave itself from the stage-0 counts --
//                      when the checkpoint test holds)
//   k_motion_select    best hypotheses, masks, DLT refit (A^T A per entry; 9x9 Jacobi in round-robin order, four disjoint
//                      rotations at a time), TrackHomo's choice
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SD_MOTION_KH 512
#define SD_MOTION_KF 1024
#define SD_MOTION_K (SD_MOTION_KH + SD_MOTION_KF)
#define SD_MOTION_KH0 64            // checkpoints: hypotheses evaluated before the termination test
#define SD_MOTION_KF0 128
#define SD_MOTION_WH 0.53           // >= (1 - 0.005^(1/64))^(1/4)  = 0.5309: 64 four-point samples reach confidence 0.995 at this inlier ratio
#define SD_MOTION_WF 0.66           // >= (1 - 0.01^(1/128))^(1/8)  = 0.6585: 128 eight-point samples reach confidence 0.99
#define SD_MOTION_CHUNK 256         // pairs whose DLT rows are staged in LDS at a time (36 KB)

struct SdMotionNorm { double meanX1, meanY1, sX1, sY1, meanX2, meanY2, sX2, sY2; int ok; int n; };
struct SdMotionResult { double H[9]; double F[9]; float HorF[9]; int nH, nF, flag, bestH, bestF; };

__device__ __forceinline__ unsigned long long sd_splitmix64(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// draws `need` (<= 8) distinct indices below N for hypothesis h of model m; false when 64 draws do not suffice.  idx stays in
// registers: every access has a compile-time index.
__device__ inline bool sd_motion_sample(int m, int h, int N, int need, int* idx /*[8]*/)
{
    int got = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) idx[j] = -1;
#pragma nounroll
    for (int c = 0; c < 64 && got < need; c++) {
        const int cand = (int)(sd_splitmix64(((unsigned long long)(m * 4096 + h) << 6) + (unsigned long long)c) % (unsigned long long)N);
        bool dup = false;
#pragma unroll
        for (int j = 0; j < 8; j++) dup |= idx[j] == cand;          // unused slots hold -1
        if (!dup) {
#pragma unroll
            for (int j = 0; j < 8; j++) if (j == got) idx[j] = cand;
            got++;
        }
    }
    return got == need;
}

// points_last / points_current of pair p (ORBmatcher.cc:505-506): the pair list of sd_batch_search_by_projection
__global__ void __launch_bounds__(256) k_motion_prepare(const sd_keypoint* __restrict__ kp, const int* __restrict__ pairs,
                                                        const int* __restrict__ npairs, const int2* __restrict__ pairIdx, int cap,
                                                        float* __restrict__ pts /*[pair][cap][4]: x1 y1 x2 y2*/,
                                                        SdMotionNorm* __restrict__ norm, const int* __restrict__ active,
                                                        const int* __restrict__ nmatch, int minMatches)
{
    const int pair = blockIdx.x, tid = threadIdx.x;
    if (active && !active[pair]) return;
    const int imgC = pairIdx[pair].x, imgL = pairIdx[pair].y;
    // TrackHomo's `if(nmatches<20) return 0` (Tracking.cc:1013-1017): no model is fitted, the flag comes out 0
    const int N = (nmatch && nmatch[pair] < minMatches) ? 0 : npairs[pair];
    float* P = pts + (size_t)pair * cap * 4;
    for (int i = tid; i < N; i += 256) {
        const int iL = pairs[((size_t)pair * cap + i) * 2], iC = pairs[((size_t)pair * cap + i) * 2 + 1];
        const sd_keypoint a = kp[(size_t)imgL * cap + iL], b = kp[(size_t)imgC * cap + iC];
        P[4 * i] = a.x; P[4 * i + 1] = a.y; P[4 * i + 2] = b.x; P[4 * i + 3] = b.y;
    }
    __threadfence_block();
    __syncthreads();
    // serial f64 sums in index order (the order the oracle uses), one lane per coordinate, from LDS
    extern __shared__ __align__(16) float sPts[];                 // [N][4]
    __shared__ double s_mean[4], s_dev[4];
    for (int i = tid; i < 4 * N; i += 256) sPts[i] = P[i];
    __syncthreads();
    if (tid < 4) {
        double m = 0;
        for (int i = 0; i < N; i++) m += (double)sPts[4 * i + tid];
        m = m / N;
        double d = 0;
        for (int i = 0; i < N; i++) d += fabs((double)sPts[4 * i + tid] - m);
        d = d / N;
        s_mean[tid] = m; s_dev[tid] = d;
    }
    __syncthreads();
    if (tid == 0) {
        const bool ok = N >= 8 && s_dev[0] > 0 && s_dev[1] > 0 && s_dev[2] > 0 && s_dev[3] > 0;
        SdMotionNorm* nm = norm + pair;
        nm->meanX1 = s_mean[0]; nm->meanY1 = s_mean[1]; nm->meanX2 = s_mean[2]; nm->meanY2 = s_mean[3];
        nm->sX1 = ok ? 1.0 / s_dev[0] : 0; nm->sY1 = ok ? 1.0 / s_dev[1] : 0; nm->sX2 = ok ? 1.0 / s_dev[2] : 0; nm->sY2 = ok ? 1.0 / s_dev[3] : 0;
        nm->n = N; nm->ok = ok ? 1 : 0;
    }
}

__device__ inline void sd_mat3_mul_d(const double* A, const double* B, double* C)
{
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += A[3 * i + k] * B[3 * k + j]; C[3 * i + j] = s; }
}
__device__ inline void sd_h_denormalize(const double* Hn, const SdMotionNorm& n, double* H)
{
    const double T1[9] = {n.sX1, 0, -n.meanX1 * n.sX1, 0, n.sY1, -n.meanY1 * n.sY1, 0, 0, 1};
    const double T2inv[9] = {1.0 / n.sX2, 0, n.meanX2, 0, 1.0 / n.sY2, n.meanY2, 0, 0, 1};
    double t[9];
    sd_mat3_mul_d(Hn, T1, t);
    sd_mat3_mul_d(T2inv, t, H);
    if (fabs(H[8]) > 1e-300) { const double s = 1.0 / H[8]; for (int k = 0; k < 9; k++) H[k] *= s; }
}
__device__ inline void sd_f_denormalize(const double* Fn, const SdMotionNorm& n, double* F)
{
    const double T1[9] = {n.sX1, 0, -n.meanX1 * n.sX1, 0, n.sY1, -n.meanY1 * n.sY1, 0, 0, 1};
    const double T2t[9] = {n.sX2, 0, 0, 0, n.sY2, 0, -n.meanX2 * n.sX2, -n.meanY2 * n.sY2, 1};
    double t[9];
    sd_mat3_mul_d(Fn, T1, t);
    sd_mat3_mul_d(T2t, t, F);
}
// The 3-px predicates in cross-multiplied form (no f64 division in the loop that dominates k_motion_hyp: the same inequalities,
// |x2 - H x1|^2 <= 9 multiplied through by w^2 and d^2 / (a^2 + b^2) <= 9 by the line's norm); the oracle evaluates the same
// products and sums in the same order.
__device__ __forceinline__ bool sd_h_inlier(const double* H, double x1, double y1, double x2, double y2)
{
    const double w = H[6] * x1 + H[7] * y1 + H[8];
    const double A = H[0] * x1 + H[1] * y1 + H[2], B = H[3] * x1 + H[4] * y1 + H[5];
    const double dx = A - x2 * w, dy = B - y2 * w;
    return dx * dx + dy * dy <= 9.0 * (w * w);
}
__device__ __forceinline__ bool sd_f_inlier(const double* F, double x1, double y1, double x2, double y2)
{
    double a = F[0] * x1 + F[1] * y1 + F[2], b = F[3] * x1 + F[4] * y1 + F[5], c = F[6] * x1 + F[7] * y1 + F[8];
    const double n2 = a * a + b * b, d2 = x2 * a + y2 * b + c;
    a = F[0] * x2 + F[3] * y2 + F[6]; b = F[1] * x2 + F[4] * y2 + F[7]; c = F[2] * x2 + F[5] * y2 + F[8];
    const double n1 = a * a + b * b, d1 = x1 * a + y1 * b + c;
    return d1 * d1 <= 9.0 * n1 && d2 * d2 <= 9.0 * n2;
}

#define SD_N1X(P, i, n) (((double)(P)[4 * (i)] - (n).meanX1) * (n).sX1)
#define SD_N1Y(P, i, n) (((double)(P)[4 * (i) + 1] - (n).meanY1) * (n).sY1)
#define SD_N2X(P, i, n) (((double)(P)[4 * (i) + 2] - (n).meanX2) * (n).sX2)
#define SD_N2Y(P, i, n) (((double)(P)[4 * (i) + 3] - (n).meanY2) * (n).sY2)

// The minimal solvers pivot at run time; with the matrix in a private array that is scratch memory (a trip to the L2 per access
// along one dependent chain: the solvers, not the inlier counts, then set the kernel time).  Here every loop is unrolled and the
// pivoting is done with selects -- "swap rows c and piv" becomes, for each row r > c, a conditional exchange under piv == r -- so
// the 8 x 9 matrix stays in registers.  Values and operation order are those of the plain elimination (the oracle's): entries the
// plain form swaps or updates but never reads again (columns < c of the rows below, column c after its elimination) are left out.
__device__ __forceinline__ void sd_xchg_if(bool c, double& a, double& b) { const double t = a; a = c ? b : a; b = c ? t : b; }

__device__ inline bool sd_h_from_4(const float* P, const SdMotionNorm& n, const int* idx, double* Hn)
{
    double M[8][9];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const double x = SD_N1X(P, idx[k], n), y = SD_N1Y(P, idx[k], n), u = SD_N2X(P, idx[k], n), v = SD_N2Y(P, idx[k], n);
        double* r0 = M[2 * k]; double* r1 = M[2 * k + 1];
        r0[0] = x; r0[1] = y; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -u * x; r0[7] = -u * y; r0[8] = u;
        r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = x; r1[4] = y; r1[5] = 1; r1[6] = -v * x; r1[7] = -v * y; r1[8] = v;
    }
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        int piv = c;
        double pv = fabs(M[c][c]);
#pragma unroll
        for (int r = c + 1; r < 8; r++) { const double a = fabs(M[r][c]); const bool g = a > pv; pv = g ? a : pv; piv = g ? r : piv; }
        ok = ok && !(pv < 1e-12);
#pragma unroll
        for (int r = c + 1; r < 8; r++) {
            const bool sw = piv == r;
#pragma unroll
            for (int k = c; k < 9; k++) sd_xchg_if(sw, M[c][k], M[r][k]);
        }
        const double d = M[c][c];
#pragma unroll
        for (int r = c + 1; r < 8; r++) {
            const double f = M[r][c] / d;
#pragma unroll
            for (int k = c + 1; k < 9; k++) M[r][k] -= f * M[c][k];
        }
    }
    if (!ok) return false;                              // (a singular system ran through with infinities; nothing was stored)
#pragma unroll
    for (int c = 7; c >= 0; c--) {
        double sum = M[c][8];
#pragma unroll
        for (int k = c + 1; k < 8; k++) sum -= M[c][k] * Hn[k];
        Hn[c] = sum / M[c][c];
    }
    Hn[8] = 1.0;
    return true;
}

// serial cyclic Jacobi for the 3x3 case (one thread)
__device__ inline void sd_jacobi3(double* a, double* v, int sweeps)
{
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) v[i * 3 + j] = i == j ? 1.0 : 0.0;
#pragma nounroll
    for (int s = 0; s < sweeps; s++)
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int q = p + 1; q < 3; q++) {
                const double apq = a[p * 3 + q];
                if (fabs(apq) < 1e-300) continue;
                const double theta = (a[q * 3 + q] - a[p * 3 + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < 3; k++) { const double akp = a[k * 3 + p], akq = a[k * 3 + q]; a[k * 3 + p] = c * akp - sn * akq; a[k * 3 + q] = sn * akp + c * akq; }
                for (int k = 0; k < 3; k++) { const double apk = a[p * 3 + k], aqk = a[q * 3 + k]; a[p * 3 + k] = c * apk - sn * aqk; a[q * 3 + k] = sn * apk + c * aqk; }
                for (int k = 0; k < 3; k++) { const double vkp = v[k * 3 + p], vkq = v[k * 3 + q]; v[k * 3 + p] = c * vkp - sn * vkq; v[k * 3 + q] = sn * vkp + c * vkq; }
            }
}

__device__ inline bool sd_f_from_8(const float* P, const SdMotionNorm& n, const int* idx, double* Fn)
{
    double M[8][9];
    int col[9];
#pragma unroll
    for (int k = 0; k < 9; k++) col[k] = k;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const double u1 = SD_N1X(P, idx[k], n), v1 = SD_N1Y(P, idx[k], n), u2 = SD_N2X(P, idx[k], n), v2 = SD_N2Y(P, idx[k], n);
        double* r = M[k];
        r[0] = u2 * u1; r[1] = u2 * v1; r[2] = u2; r[3] = v2 * u1; r[4] = v2 * v1; r[5] = v2; r[6] = u1; r[7] = v1; r[8] = 1;
    }
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        int pr = c, pc = c;
        double best = -1;
#pragma unroll
        for (int r = c; r < 8; r++)
#pragma unroll
            for (int k = c; k < 9; k++) { const double a = fabs(M[r][k]); const bool g = a > best; best = g ? a : best; pr = g ? r : pr; pc = g ? k : pc; }
        ok = ok && !(best < 1e-12);
#pragma unroll
        for (int r = c + 1; r < 8; r++) {
            const bool sw = pr == r;
#pragma unroll
            for (int k = c; k < 9; k++) sd_xchg_if(sw, M[c][k], M[r][k]);
        }
#pragma unroll
        for (int k = c + 1; k < 9; k++) {
            const bool sw = pc == k;
#pragma unroll
            for (int r = 0; r < 8; r++) sd_xchg_if(sw, M[r][c], M[r][k]);
            const int t = col[c]; col[c] = sw ? col[k] : col[c]; col[k] = sw ? t : col[k];
        }
        const double d = M[c][c];
#pragma unroll
        for (int r = c + 1; r < 8; r++) {
            const double f = M[r][c] / d;
#pragma unroll
            for (int k = c + 1; k < 9; k++) M[r][k] -= f * M[c][k];
        }
    }
    if (!ok) return false;
    double x[9];
    x[8] = 1.0;
#pragma unroll
    for (int c = 7; c >= 0; c--) {
        double sum = -M[c][8] * x[8];
#pragma unroll
        for (int k = c + 1; k < 8; k++) sum -= M[c][k] * x[k];
        x[c] = sum / M[c][c];
    }
    double f[9], nrm = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) f[j] = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int j = 0; j < 9; j++) f[j] = col[k] == j ? x[k] : f[j];       // f[col[k]] = x[k]
        nrm += x[k] * x[k];
    }
    nrm = 1.0 / sqrt(nrm);
#pragma unroll
    for (int k = 0; k < 9; k++) f[k] *= nrm;
    double G[9], V[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) { double sum = 0; for (int k = 0; k < 3; k++) sum += f[3 * k + i] * f[3 * k + j]; G[3 * i + j] = sum; }
    sd_jacobi3(G, V, 10);
    // smallest eigenvalue's column of V without a run-time register index
    const bool s1 = G[4] < G[0];
    const double g01 = s1 ? G[4] : G[0];
    const bool s2 = G[8] < g01;
    const double v[3] = {s2 ? V[2] : (s1 ? V[1] : V[0]), s2 ? V[5] : (s1 ? V[4] : V[3]), s2 ? V[8] : (s1 ? V[7] : V[6])};
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const double fv = f[3 * i] * v[0] + f[3 * i + 1] * v[1] + f[3 * i + 2] * v[2];
#pragma unroll
        for (int j = 0; j < 3; j++) Fn[3 * i + j] = f[3 * i + j] - fv * v[j];
    }
    return true;
}

// builds hypothesis `hyp` (0..511: H, 512..1535: F); false when the sample is degenerate
__device__ inline bool sd_motion_model(const float* P, const SdMotionNorm& n, int hyp, double* Mdl)
{
    int idx[8];
    if (hyp < SD_MOTION_KH) {
        double Hn[9];
        if (!sd_motion_sample(0, hyp, n.n, 4, idx) || !sd_h_from_4(P, n, idx, Hn)) return false;
        sd_h_denormalize(Hn, n, Mdl);
    } else {
        double Fn[9];
        if (!sd_motion_sample(1, hyp - SD_MOTION_KH, n.n, 8, idx) || !sd_f_from_8(P, n, idx, Fn)) return false;
        sd_f_denormalize(Fn, n, Mdl);
    }
    return true;
}

// Best hypothesis of counts[lo, hi) as one key per wave: (count << 11) | (2047 - index) for count > 0, else 0 -- the maximum is
// the serial scan's `if (c > best)` winner (most inliers, lowest index on ties); every lane returns it.
__device__ __forceinline__ int sd_motion_wave_best(const int* cn, int lo, int hi, int lane)
{
    int key = 0;
    for (int h = lo + lane; h < hi; h += 64) { const int c = cn[h]; key = max(key, c > 0 ? (c << 11) | (2047 - h) : 0); }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) key = max(key, __shfl_xor(key, o, 64));
    return key;
}
// The checkpoint test of model m (0: H, 1: F) for one pair, evaluated by a whole wave from the stage-0 counts; key = the winner
// over the hypotheses that count.  Returns true when the search stops at the checkpoint.
__device__ __forceinline__ bool sd_motion_checkpoint(const int* cnModel, int m, int N, int lane, int& key)
{
    key = sd_motion_wave_best(cnModel, 0, m ? SD_MOTION_KF0 : SD_MOTION_KH0, lane);
    return (double)(key >> 11) >= (m ? SD_MOTION_WF : SD_MOTION_WH) * (double)N;
}

// The hypotheses of a stage in two kernels, so that both parts keep their lanes busy:
//   k_motion_models   one THREAD per hypothesis: sample -> minimal solver -> denormalised model (9 f64) to HBM; count = -1 for a
//                     degenerate sample, else 0.  64 hypotheses of one model per wave; in stage 1 the wave first evaluates the
//                     checkpoint test of its model from the stage-0 counts and leaves when the search has stopped.
//   k_motion_count    one WORKGROUP per SD_MOTION_HB hypotheses of one model of one pair, ONE POINT PAIR PER LANE: the points sit
//                     in registers (f64) for all hypotheses of the block, the model is uniform (scalar loads), a hypothesis'
//                     count is the sum of the lanes' counts.  (One thread per hypothesis walking all pairs -- the round-1 form --
//                     is a ~50 000-instruction dependent f64 loop per wave, on a few hundred waves.)
#define SD_MOTION_HB 16
#define SD_MOTION_MAXJ 8                                // point pairs per lane and pass: a pass covers 4 waves x 64 lanes x 8 = 2048 pairs
#define SD_MOTION_N0 (SD_MOTION_KH0 + SD_MOTION_KF0)
#define SD_MOTION_N1 (SD_MOTION_KH - SD_MOTION_KH0 + SD_MOTION_KF - SD_MOTION_KF0)
// hypothesis t of a stage -> (model, index inside the model); both stages list their H hypotheses first
__device__ __forceinline__ void sd_motion_stage_hyp(int stage, int t, int& m, int& h)
{
    const int nH = stage ? SD_MOTION_KH - SD_MOTION_KH0 : SD_MOTION_KH0;
    m = t < nH ? 0 : 1;
    h = m ? (stage ? SD_MOTION_KF0 : 0) + t - nH : (stage ? SD_MOTION_KH0 : 0) + t;
}

// grid (SD_MOTION_N0 | SD_MOTION_N1 / 64, pairs), 64 threads
__global__ void __launch_bounds__(64) k_motion_models(const float* __restrict__ pts, const SdMotionNorm* __restrict__ norm, int cap,
                                                      int* __restrict__ counts /*[pair][SD_MOTION_K]*/, double* __restrict__ models /*[pair][SD_MOTION_K][9]*/,
                                                      const int* __restrict__ active, int stage)
{
    const int pair = blockIdx.y, lane = threadIdx.x;
    if (active && !active[pair]) return;
    int m, h;
    sd_motion_stage_hyp(stage, blockIdx.x * 64 + lane, m, h);            // m is uniform: the H counts of both stages are multiples of 64
    const int hyp = (m ? SD_MOTION_KH : 0) + h;
    const SdMotionNorm n = norm[pair];
    int* cn = counts + (size_t)pair * SD_MOTION_K;
    if (!n.ok) { cn[hyp] = -1; return; }
    if (stage == 1) {
        int key;
        if (sd_motion_checkpoint(cn + (m ? SD_MOTION_KH : 0), m, n.n, lane, key)) return;     // decided: nobody reads anything beyond the checkpoint
    }
    double Mdl[9];
    const bool ok = sd_motion_model(pts + (size_t)pair * cap * 4, n, hyp, Mdl);
    double* out = models + ((size_t)pair * SD_MOTION_K + hyp) * 9;
#pragma unroll
    for (int k = 0; k < 9; k++) out[k] = Mdl[k];
    cn[hyp] = ok ? 0 : -1;
}

// grid (SD_MOTION_N0 | SD_MOTION_N1 / SD_MOTION_HB, pairs), 256 threads
__global__ void __launch_bounds__(256) k_motion_count(const float* __restrict__ pts, const SdMotionNorm* __restrict__ norm, int cap,
                                                      int* __restrict__ counts, const double* __restrict__ models,
                                                      const int* __restrict__ active, int stage)
{
    __shared__ int s_cnt[SD_MOTION_HB];
    const int pair = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    if (active && !active[pair]) return;
    int m, h0;
    sd_motion_stage_hyp(stage, blockIdx.x * SD_MOTION_HB, m, h0);
    const int hyp0 = (m ? SD_MOTION_KH : 0) + h0;
    const SdMotionNorm n = norm[pair];
    if (!n.ok) return;
    int* cn = counts + (size_t)pair * SD_MOTION_K;
    if (stage == 1) {
        int key;
        if (sd_motion_checkpoint(cn + (m ? SD_MOTION_KH : 0), m, n.n, lane, key)) return;     // every wave finds the same
    }
    if (tid < SD_MOTION_HB) s_cnt[tid] = 0;
    __syncthreads();
    const float* P = pts + (size_t)pair * cap * 4;
    const double* Mp = models + ((size_t)pair * SD_MOTION_K + hyp0) * 9;
    for (int base = 0; base < n.n; base += 256 * SD_MOTION_MAXJ) {
        double x1[SD_MOTION_MAXJ], y1[SD_MOTION_MAXJ], x2[SD_MOTION_MAXJ], y2[SD_MOTION_MAXJ];
        const int nj = min(SD_MOTION_MAXJ, (n.n - base + 255) >> 8);             // uniform
#pragma unroll
        for (int j = 0; j < SD_MOTION_MAXJ; j++) {
            const int i = base + 256 * j + tid;
            const float4 q = j < nj && i < n.n ? *(const float4*)(P + 4 * (size_t)i) : make_float4(0.f, 0.f, 0.f, 0.f);
            x1[j] = q.x; y1[j] = q.y; x2[j] = q.z; y2[j] = q.w;
        }
#pragma nounroll
        for (int h = 0; h < SD_MOTION_HB; h++) {
            if (cn[hyp0 + h] < 0) continue;                                      // degenerate sample (k_motion_models); uniform
            double Mdl[9];
#pragma unroll
            for (int k = 0; k < 9; k++) Mdl[k] = Mp[9 * h + k];
            int c = 0;
#pragma unroll
            for (int j = 0; j < SD_MOTION_MAXJ; j++) {
                if (j < nj) {
                    const bool in = base + 256 * j + tid < n.n && (m == 0 ? sd_h_inlier(Mdl, x1[j], y1[j], x2[j], y2[j]) : sd_f_inlier(Mdl, x1[j], y1[j], x2[j], y2[j]));
                    c += in ? 1 : 0;
                }
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o, 64);
            if (lane == 0 && c) atomicAdd(&s_cnt[h], c);
        }
    }
    __syncthreads();
    if (tid < SD_MOTION_HB && cn[hyp0 + tid] >= 0) cn[hyp0 + tid] = s_cnt[tid];
}

// Round-robin order of the 36 index pairs of a 9x9 Jacobi sweep: round r holds the four pairs with p + q = r (mod 9); the pairs of
// a round are disjoint, so their rotations are computed from the same matrix and applied together (all column updates, then all
// row updates) -- nine dependent steps per sweep instead of 36.
__device__ const unsigned char sd_jacobi_rr[9][4][2] = {{{1, 8}, {2, 7}, {3, 6}, {4, 5}}, {{0, 1}, {2, 8}, {3, 7}, {4, 6}}, {{0, 2}, {3, 8}, {4, 7}, {5, 6}},
                                                        {{0, 3}, {1, 2}, {4, 8}, {5, 7}}, {{0, 4}, {1, 3}, {5, 8}, {6, 7}}, {{0, 5}, {1, 4}, {2, 3}, {6, 8}},
                                                        {{0, 6}, {1, 5}, {2, 4}, {7, 8}}, {{0, 7}, {1, 6}, {2, 5}, {3, 4}}, {{0, 8}, {1, 7}, {2, 6}, {3, 5}}};

__global__ void __launch_bounds__(256) k_motion_select(const float* __restrict__ pts, const SdMotionNorm* __restrict__ norm,
                                                       const int* __restrict__ counts, int cap, uint8_t* __restrict__ maskH,
                                                       uint8_t* __restrict__ maskF, SdMotionResult* __restrict__ res,
                                                       const int* __restrict__ active)
{
    if (active && !active[blockIdx.x]) return;
    __shared__ int s_best[2], s_cnt[2];
    __shared__ double s_H[9], s_F[9], s_M[81], s_V[81];
    __shared__ double s_rows[SD_MOTION_CHUNK * 18];
    __shared__ int s_okH, s_okF;
    const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const SdMotionNorm n = norm[pair];
    const float* P = pts + (size_t)pair * cap * 4;
    const int* cn = counts + (size_t)pair * SD_MOTION_K;
    SdMotionResult* R = res + pair;
    uint8_t* mH = maskH + (size_t)pair * cap;
    uint8_t* mF = maskF + (size_t)pair * cap;
    for (int i = tid; i < n.n; i += 256) { mH[i] = 0; mF[i] = 0; }
    if (tid < 128) {                             // wave 0: H, wave 1: F.  Winner = the first maximum (strict >) of the serial loop over
        const int m = tid >> 6;                  // the hypotheses that count: up to the checkpoint when its test holds, else all of them
        const int* cm = cn + (m ? SD_MOTION_KH : 0);
        int key = 0;
        if (n.ok && !sd_motion_checkpoint(cm, m, n.n, lane, key))
            key = max(key, sd_motion_wave_best(cm, m ? SD_MOTION_KF0 : SD_MOTION_KH0, m ? SD_MOTION_KF : SD_MOTION_KH, lane));
        if (lane == 0) { s_best[m] = key ? (m ? SD_MOTION_KH : 0) + 2047 - (key & 2047) : -1; s_cnt[m] = key >> 11; }
    }
    __syncthreads();
    if (tid == 0) {
        double Mdl[9];
        s_okH = s_best[0] >= 0 && s_cnt[0] >= 4 && sd_motion_model(P, n, s_best[0], Mdl);
        if (s_okH) for (int k = 0; k < 9; k++) s_H[k] = Mdl[k];
    }
    if (tid == 64) {
        double Mdl[9];
        s_okF = s_best[1] >= 0 && s_cnt[1] >= 8 && sd_motion_model(P, n, s_best[1], Mdl);
        if (s_okF) for (int k = 0; k < 9; k++) s_F[k] = Mdl[k];
    }
    __syncthreads();
    const bool okH = s_okH != 0, okF = s_okF != 0;
    if (okH) for (int i = tid; i < n.n; i += 256) mH[i] = sd_h_inlier(s_H, P[4 * i], P[4 * i + 1], P[4 * i + 2], P[4 * i + 3]) ? 1 : 0;
    if (okF) for (int i = tid; i < n.n; i += 256) mF[i] = sd_f_inlier(s_F, P[4 * i], P[4 * i + 1], P[4 * i + 2], P[4 * i + 3]) ? 1 : 0;
    __threadfence_block();
    __syncthreads();
    // ---- refit H on its inliers: A^T A, one lane per entry, serial over the pairs in index order (Initializer::ComputeH21 rows)
    if (okH) {
        // rows of Initializer::ComputeH21 for a chunk of pairs go to LDS (all threads; rows of non-inliers are zero, and
        // adding +0 never changes the sum), then lane (p, q) adds its products in index order
        const int p = tid / 9, q = tid % 9;
        double acc = 0;
        for (int c0 = 0; c0 < n.n; c0 += SD_MOTION_CHUNK) {
            const int cn_ = n.n - c0 < SD_MOTION_CHUNK ? n.n - c0 : SD_MOTION_CHUNK;
            __syncthreads();
            for (int j = tid; j < cn_; j += 256) {
                const int i = c0 + j;
                double* r0 = s_rows + (size_t)j * 18; double* r1 = r0 + 9;
                if (mH[i]) {
                    const double u1 = SD_N1X(P, i, n), v1 = SD_N1Y(P, i, n), u2 = SD_N2X(P, i, n), v2 = SD_N2Y(P, i, n);
                    r0[0] = 0; r0[1] = 0; r0[2] = 0; r0[3] = -u1; r0[4] = -v1; r0[5] = -1; r0[6] = v2 * u1; r0[7] = v2 * v1; r0[8] = v2;
                    r1[0] = u1; r1[1] = v1; r1[2] = 1; r1[3] = 0; r1[4] = 0; r1[5] = 0; r1[6] = -u2 * u1; r1[7] = -u2 * v1; r1[8] = -u2;
                } else {
                    for (int k = 0; k < 18; k++) r0[k] = 0;
                }
            }
            __syncthreads();
            if (tid < 81)
                for (int j = 0; j < cn_; j++) {
                    const double* r0 = s_rows + (size_t)j * 18;
                    acc += r0[p] * r0[q];
                    acc += r0[9 + p] * r0[9 + q];
                }
        }
        if (tid < 81) { s_M[tid] = acc; s_V[tid] = p == q ? 1.0 : 0.0; }
        __syncthreads();
        // Jacobi, 15 sweeps in round-robin order: lane = 9 g + k, g = which of the round's four rotations, k = the row / column
        // index it updates.  The nine lanes of a rotation each evaluate its (c, s) (same operands, same result).
        if (tid < 64) {
            const int g = lane < 36 ? lane / 9 : 0, k = lane - 9 * g;
            for (int sw = 0; sw < 15; sw++)
                for (int r = 0; r < 9; r++) {
                    const int p = sd_jacobi_rr[r][g][0], q = sd_jacobi_rr[r][g][1];
                    const double apq = s_M[p * 9 + q];
                    const bool rot = lane < 36 && !(fabs(apq) < 1e-300);
                    double c = 1.0, sn = 0.0;
                    if (rot) {
                        const double theta = (s_M[q * 9 + q] - s_M[p * 9 + p]) / (2.0 * apq);
                        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                        c = 1.0 / sqrt(t * t + 1.0); sn = t * c;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (rot) { const double akp = s_M[k * 9 + p], akq = s_M[k * 9 + q]; s_M[k * 9 + p] = c * akp - sn * akq; s_M[k * 9 + q] = sn * akp + c * akq; }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (rot) { const double apk = s_M[p * 9 + k], aqk = s_M[q * 9 + k]; s_M[p * 9 + k] = c * apk - sn * aqk; s_M[q * 9 + k] = sn * apk + c * aqk; }
                    if (rot) { const double vkp = s_V[k * 9 + p], vkq = s_V[k * 9 + q]; s_V[k * 9 + p] = c * vkp - sn * vkq; s_V[k * 9 + q] = sn * vkp + c * vkq; }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
        }
        __syncthreads();
    }
    if (tid == 0) {
        for (int k = 0; k < 9; k++) { R->H[k] = 0; R->F[k] = 0; R->HorF[k] = 0.f; }
        R->bestH = s_best[0]; R->bestF = s_best[1];
        R->nH = okH ? s_cnt[0] : 0; R->nF = okF ? s_cnt[1] : 0;
        if (okH) {
            int sm = 0;
            for (int k = 1; k < 9; k++) if (s_M[10 * k] < s_M[10 * sm]) sm = k;
            double Hn[9], H[9];
            for (int k = 0; k < 9; k++) Hn[k] = s_V[9 * k + sm];
            sd_h_denormalize(Hn, n, H);
            for (int k = 0; k < 9; k++) R->H[k] = H[k];
        }
        if (okF) for (int k = 0; k < 9; k++) R->F[k] = s_F[k];
        int flag = 0;                            // Tracking.cc:1060-1072
        if (R->nF > 10 || R->nH > 10) {
            if (R->nH > R->nF) { flag = 1; for (int k = 0; k < 9; k++) R->HorF[k] = (float)R->H[k]; }
            else { flag = 2; for (int k = 0; k < 9; k++) R->HorF[k] = (float)R->F[k]; }
        }
        R->flag = flag;
    }
}
