// The model fit of Tracking::TrackHomo (src/Tracking.cc:1026-1075) on the device: homography and fundamental matrix
// from the background matches of SearchByProjection, inlier masks and the reference's choice between the two.
// cv::findHomography / cv::findFundamentalMat cannot be matched bit for bit without OpenCV; the algorithm is this
// build's spec: Hartley normalisation as Initializer::Normalize (src/Initializer.cc:749-795), 512 + 1024
// independent hypotheses from a counter-based sampler (all evaluated in parallel: one thread each), OpenCV's inlier
// criteria (3 px), most inliers wins (lowest index on ties), H refitted on its inliers by the normalised DLT of
// Initializer::ComputeH21 (:246-272).  All arithmetic is f64 in a fixed order (no contraction), so the CPU oracle's
// independent restatement gives the same bits.
//   k_motion_prepare   gather the point pairs, normalisation parameters (serial f64 sums in index order)
//   k_motion_hyp       one thread per hypothesis: sample -> minimal solver -> inlier count over all pairs
//   k_motion_select    best hypotheses, masks, DLT refit (A^T A per entry, 9-lane cyclic Jacobi), TrackHomo's choice
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SD_MOTION_KH 512
#define SD_MOTION_KF 1024
#define SD_MOTION_K (SD_MOTION_KH + SD_MOTION_KF)
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

__device__ inline bool sd_motion_sample(int m, int h, int N, int need, int* idx)
{
    int got = 0;
    for (int c = 0; c < 64 && got < need; c++) {
        const int cand = (int)(sd_splitmix64(((unsigned long long)(m * 4096 + h) << 6) + (unsigned long long)c) % (unsigned long long)N);
        bool dup = false;
        for (int j = 0; j < got; j++) dup |= idx[j] == cand;
        if (!dup) idx[got++] = cand;
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

__device__ inline bool sd_h_from_4(const float* P, const SdMotionNorm& n, const int* idx, double* Hn)
{
    double M[8][9];
    for (int k = 0; k < 4; k++) {
        const double x = SD_N1X(P, idx[k], n), y = SD_N1Y(P, idx[k], n), u = SD_N2X(P, idx[k], n), v = SD_N2Y(P, idx[k], n);
        double* r0 = M[2 * k]; double* r1 = M[2 * k + 1];
        r0[0] = x; r0[1] = y; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -u * x; r0[7] = -u * y; r0[8] = u;
        r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = x; r1[4] = y; r1[5] = 1; r1[6] = -v * x; r1[7] = -v * y; r1[8] = v;
    }
    for (int c = 0; c < 8; c++) {
        int piv = c;
        for (int r = c + 1; r < 8; r++) if (fabs(M[r][c]) > fabs(M[piv][c])) piv = r;
        if (fabs(M[piv][c]) < 1e-12) return false;
        if (piv != c) for (int k = 0; k < 9; k++) { const double t = M[c][k]; M[c][k] = M[piv][k]; M[piv][k] = t; }
        for (int r = c + 1; r < 8; r++) {
            const double f = M[r][c] / M[c][c];
            for (int k = c; k < 9; k++) M[r][k] -= f * M[c][k];
        }
    }
    for (int c = 7; c >= 0; c--) {
        double s = M[c][8];
        for (int k = c + 1; k < 8; k++) s -= M[c][k] * Hn[k];
        Hn[c] = s / M[c][c];
    }
    Hn[8] = 1.0;
    return true;
}

// serial cyclic Jacobi for the 3x3 case (one thread)
__device__ inline void sd_jacobi3(double* a, double* v, int sweeps)
{
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) v[i * 3 + j] = i == j ? 1.0 : 0.0;
    for (int s = 0; s < sweeps; s++)
        for (int p = 0; p < 2; p++)
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
    for (int k = 0; k < 9; k++) col[k] = k;
    for (int k = 0; k < 8; k++) {
        const double u1 = SD_N1X(P, idx[k], n), v1 = SD_N1Y(P, idx[k], n), u2 = SD_N2X(P, idx[k], n), v2 = SD_N2Y(P, idx[k], n);
        double* r = M[k];
        r[0] = u2 * u1; r[1] = u2 * v1; r[2] = u2; r[3] = v2 * u1; r[4] = v2 * v1; r[5] = v2; r[6] = u1; r[7] = v1; r[8] = 1;
    }
    for (int c = 0; c < 8; c++) {
        int pr = c, pc = c;
        double best = -1;
        for (int r = c; r < 8; r++) for (int k = c; k < 9; k++) if (fabs(M[r][k]) > best) { best = fabs(M[r][k]); pr = r; pc = k; }
        if (best < 1e-12) return false;
        if (pr != c) for (int k = 0; k < 9; k++) { const double t = M[c][k]; M[c][k] = M[pr][k]; M[pr][k] = t; }
        if (pc != c) { for (int r = 0; r < 8; r++) { const double t = M[r][c]; M[r][c] = M[r][pc]; M[r][pc] = t; } const int t = col[c]; col[c] = col[pc]; col[pc] = t; }
        for (int r = c + 1; r < 8; r++) {
            const double f = M[r][c] / M[c][c];
            for (int k = c; k < 9; k++) M[r][k] -= f * M[c][k];
        }
    }
    double x[9];
    x[8] = 1.0;
    for (int c = 7; c >= 0; c--) {
        double s = -M[c][8] * x[8];
        for (int k = c + 1; k < 8; k++) s -= M[c][k] * x[k];
        x[c] = s / M[c][c];
    }
    double f[9], nrm = 0;
    for (int k = 0; k < 9; k++) { f[col[k]] = x[k]; nrm += x[k] * x[k]; }
    nrm = 1.0 / sqrt(nrm);
    for (int k = 0; k < 9; k++) f[k] *= nrm;
    double G[9], V[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += f[3 * k + i] * f[3 * k + j]; G[3 * i + j] = s; }
    sd_jacobi3(G, V, 10);
    int sm = 0;
    for (int k = 1; k < 3; k++) if (G[4 * k] < G[4 * sm]) sm = k;
    const double v[3] = {V[sm], V[3 + sm], V[6 + sm]};
    for (int i = 0; i < 3; i++) {
        const double fv = f[3 * i] * v[0] + f[3 * i + 1] * v[1] + f[3 * i + 2] * v[2];
        for (int j = 0; j < 3; j++) Fn[3 * i + j] = f[3 * i + j] - fv * v[j];
    }
    return true;
}

// builds hypothesis `hyp` (0..511: H, 512..1535: F); false when the sample is degenerate
__device__ inline bool sd_motion_model(const float* P, const SdMotionNorm& n, int hyp, double* Mdl)
{
    if (hyp < SD_MOTION_KH) {
        int idx[4];
        double Hn[9];
        if (!sd_motion_sample(0, hyp, n.n, 4, idx) || !sd_h_from_4(P, n, idx, Hn)) return false;
        sd_h_denormalize(Hn, n, Mdl);
    } else {
        int idx[8];
        double Fn[9];
        if (!sd_motion_sample(1, hyp - SD_MOTION_KH, n.n, 8, idx) || !sd_f_from_8(P, n, idx, Fn)) return false;
        sd_f_denormalize(Fn, n, Mdl);
    }
    return true;
}

__global__ void __launch_bounds__(256) k_motion_hyp(const float* __restrict__ pts, const SdMotionNorm* __restrict__ norm, int cap,
                                                    int* __restrict__ counts /*[pair][SD_MOTION_K]*/, const int* __restrict__ active)
{
    extern __shared__ __align__(16) float sP[];               // [N][4]
    const int pair = blockIdx.y, hyp = blockIdx.x * 256 + threadIdx.x;
    if (active && !active[pair]) return;
    const SdMotionNorm n = norm[pair];
    const float* P = pts + (size_t)pair * cap * 4;
    if (!n.ok) { if (hyp < SD_MOTION_K) counts[(size_t)pair * SD_MOTION_K + hyp] = -1; return; }
    for (int i = threadIdx.x; i < n.n * 4; i += 256) sP[i] = P[i];
    __syncthreads();
    if (hyp >= SD_MOTION_K) return;
    double Mdl[9];
    int cnt = -1;
    if (sd_motion_model(sP, n, hyp, Mdl)) {
        cnt = 0;
        if (hyp < SD_MOTION_KH) { for (int i = 0; i < n.n; i++) cnt += sd_h_inlier(Mdl, sP[4 * i], sP[4 * i + 1], sP[4 * i + 2], sP[4 * i + 3]) ? 1 : 0; }
        else { for (int i = 0; i < n.n; i++) cnt += sd_f_inlier(Mdl, sP[4 * i], sP[4 * i + 1], sP[4 * i + 2], sP[4 * i + 3]) ? 1 : 0; }
    }
    counts[(size_t)pair * SD_MOTION_K + hyp] = cnt;
}

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
    if (tid < 2) {                               // the first maximum (strict >) = what the serial loop keeps
        const int lo = tid == 0 ? 0 : SD_MOTION_KH, hi = tid == 0 ? SD_MOTION_KH : SD_MOTION_K;
        int best = -1, bc = 0;
        if (n.ok) for (int h = lo; h < hi; h++) { const int c = cn[h]; if (c > bc) { bc = c; best = h; } }
        s_best[tid] = best; s_cnt[tid] = bc;
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
        // cyclic Jacobi, 15 sweeps; the three update loops of a rotation run on 9 lanes (k = lane)
        if (tid < 64) {
            for (int s = 0; s < 15; s++)
                for (int p = 0; p < 8; p++)
                    for (int q = p + 1; q < 9; q++) {
                        const double apq = s_M[p * 9 + q];
                        if (fabs(apq) < 1e-300) continue;                         // wave-uniform (LDS value)
                        const double theta = (s_M[q * 9 + q] - s_M[p * 9 + p]) / (2.0 * apq);
                        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                        const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                        __builtin_amdgcn_wave_barrier();
                        if (lane < 9) { const int k = lane; const double akp = s_M[k * 9 + p], akq = s_M[k * 9 + q]; s_M[k * 9 + p] = c * akp - sn * akq; s_M[k * 9 + q] = sn * akp + c * akq; }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        if (lane < 9) { const int k = lane; const double apk = s_M[p * 9 + k], aqk = s_M[q * 9 + k]; s_M[p * 9 + k] = c * apk - sn * aqk; s_M[q * 9 + k] = sn * apk + c * aqk; }
                        if (lane < 9) { const int k = lane; const double vkp = s_V[k * 9 + p], vkq = s_V[k * 9 + q]; s_V[k * 9 + p] = c * vkp - sn * vkq; s_V[k * 9 + q] = sn * vkp + c * vkq; }
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
