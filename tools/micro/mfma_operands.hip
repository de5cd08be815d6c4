// Developer micro-benchmark: does the RATE of v_mfma_f32_32x32x16_bf16 depend on how often its A / B source registers change?  No memory traffic at all: six
// A and six B fragments sit in registers, four accumulators are cycled (every MFMA is independent of the three before it), the issue order is pinned.
//   P0  every MFMA reads the same A and the same B          P1  A changes every MFMA (six fragments cycled), B constant
//   P2  A and B both change every MFMA                        P3  the real kernel's pattern: (la, lb) pairs over 2 x 2 tiles, A constant for two MFMAs, B alternating
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
template <int P>
__global__ void __launch_bounds__(256, 2) k(const u4* __restrict__ src, float* out, int steps, unsigned long long* clk)
{
    const int tid = threadIdx.x;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    f16v acc[4];
    for (int a = 0; a < 4; a++) for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
    b8 A[6], B[6];
    for (int i = 0; i < 6; i++) { A[i] = __builtin_bit_cast(b8, src[tid + 256 * i]); B[i] = __builtin_bit_cast(b8, src[tid + 256 * (6 + i)]); }
    for (int s = 0; s < steps; s++) {
#pragma unroll
        for (int g = 0; g < 24; g++) {
            const int a = g & 3;
            int ia, ib;
            if (P == 0) { ia = 0; ib = 0; }
            else if (P == 1) { ia = g % 6; ib = 0; }
            else if (P == 2) { ia = g % 6; ib = (g + 3) % 6; }
            else { const int grp = g >> 2; const int la = grp == 0 ? 0 : grp == 1 ? 0 : grp == 2 ? 1 : grp == 3 ? 1 : grp == 4 ? 0 : 2, lb = grp == 0 ? 0 : grp == 1 ? 1 : grp == 2 ? 0 : grp == 3 ? 1 : grp == 4 ? 2 : 0;
                   ia = 2 * la + (a >> 1); ib = 2 * lb + (a & 1); }
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ia], B[ib], acc[a], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float t = 0.f;
    for (int a = 0; a < 4; a++) for (int r = 0; r < 16; r++) t += acc[a][r];
    out[blockIdx.x * 256 + tid] = t;
    if (blockIdx.x == gridDim.x - 1 && tid == 0 && clk) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}
// Q: the limb kernel's step with its operand traffic: six A fragments per step from global memory, requested DIST steps ahead (three register sets), six B
// fragments per step from LDS, read at the step start; MFMA order pinned.  G = 0: no global requests (A constant), L = 0: no LDS reads (B constant)
template <int G, int L, int DIST, int SHARED = 0>
__global__ void __launch_bounds__(256, 2) q(const u4* __restrict__ src, float* out, int steps, unsigned mask)
{
    __shared__ u4 lds[2048];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2048; i += 256) lds[i] = u4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    __syncthreads();
    f16v acc[4];
    for (int a = 0; a < 4; a++) for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
    u4 ring[3][6];
    for (int s2 = 0; s2 < 3; s2++) for (int i = 0; i < 6; i++) ring[s2][i] = src[(tid + 256 * i + 1536 * s2) & mask];
    b8 B[6];
    for (int i = 0; i < 6; i++) B[i] = __builtin_bit_cast(b8, lds[lane + 64 * i]);
    unsigned off = ((SHARED ? 0u : blockIdx.x * 4096u) + (SHARED == 2 ? (tid & 127) : tid)) & mask;      // SHARED: every workgroup walks the same stream (2: and wave pairs the same lines)
    auto step = [&](const int ws, const int sidx) {
        if (L) {
#pragma unroll
            for (int i = 0; i < 4; i++) B[i] = __builtin_bit_cast(b8, lds[(lane + 64 * i + 384 * (sidx & 3)) & 2047]);
        }
        if (G) {
#pragma unroll
            for (int i = 0; i < 6; i++) ring[(ws + DIST) % 3][i] = src[(off + 256u * i) & mask];
            off = (off + 1536u) & mask;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 24; g++) {
            const int a = g & 3, grp = g >> 2;
            const int la = grp == 0 ? 0 : grp == 1 ? 0 : grp == 2 ? 1 : grp == 3 ? 1 : grp == 4 ? 0 : 2, lb = grp == 0 ? 0 : grp == 1 ? 1 : grp == 2 ? 0 : grp == 3 ? 1 : grp == 4 ? 2 : 0;
            if (L && g == 4) {
#pragma unroll
                for (int i = 4; i < 6; i++) B[i] = __builtin_bit_cast(b8, lds[(lane + 64 * i + 384 * (sidx & 3)) & 2047]);
            }
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, ring[ws][2 * la + (a >> 1)]), B[2 * lb + (a & 1)], acc[a], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int s = 0; s < steps; s += 3) { step(0, s); step(1, s + 1); step(2, s + 2); }
    float t = 0.f;
    for (int a = 0; a < 4; a++) for (int r = 0; r < 16; r++) t += acc[a][r];
    out[blockIdx.x * 256 + tid] = t;
}
template <int G, int L, int DIST, int SHARED = 0>
static void runq(const char* name, const u4* src, float* out, unsigned mask)
{
    const int grid = 512 * 8, steps = 3000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((q<G, L, DIST, SHARED>), dim3(grid), dim3(256), 0, 0, src, out, 30, mask);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((q<G, L, DIST, SHARED>), dim3(grid), dim3(256), 0, 0, src, out, steps, mask);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)grid * 4 * steps * 24 * 32768.0;
    printf("%-70s %8.3f ms  %7.1f TFLOP/s bf16  (%.1f %% of 2500)\n", name, ms, fl / (ms * 1e-3) / 1e12, fl / (ms * 1e-3) / 2.5e15 * 100);
}
template <int P>
static void run(const char* name, const u4* src, float* out, int steps = 3000)
{
    const int grid = 512 * 8;
    static unsigned long long* clk = nullptr;
    if (!clk) hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<P>, dim3(grid), dim3(256), 0, 0, src, out, 30, (unsigned long long*)nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<P>, dim3(grid), dim3(256), 0, 0, src, out, steps, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double fl = (double)grid * 4 * steps * 24 * 32768.0;
    printf("%-70s %8.3f ms  %7.1f TFLOP/s bf16  (%.1f %% of 2500)  last workgroup: %.0f MHz\n", name, ms, fl / (ms * 1e-3) / 1e12, fl / (ms * 1e-3) / 2.5e15 * 100, (double)h[0] / (double)h[1] * 100.0);
}
int main()
{
    u4* src; float* out;
    hipMalloc(&src, 256 * 12 * 16); hipMemset(src, 0x3f, 256 * 12 * 16);
    hipMalloc(&out, (size_t)512 * 8 * 256 * 4);
    run<0>("P0 same A, same B", src, out);
    run<1>("P1 A changes every MFMA, B constant", src, out);
    run<2>("P2 A and B change every MFMA", src, out);
    run<3>("P3 the limb kernel's pattern", src, out);
    run<3>("P3, 10 x longer (sustained), operands 0x3f3f...", src, out, 30000);
    {   // random operand bits (finite bf16 values in [1, 2) with random mantissas): data toggling costs power
        std::vector<unsigned> hst(256 * 12 * 4);
        unsigned x = 12345u;
        for (auto& v : hst) { x = x * 1664525u + 1013904223u; const unsigned lo = 0x3f80u | ((x >> 8) & 0x7fu), hi = 0x3f80u | ((x >> 20) & 0x7fu) | ((x >> 31) << 15); v = lo | (hi << 16); }
        hipMemcpy(src, hst.data(), hst.size() * 4, hipMemcpyHostToDevice);
    }
    run<3>("P3, random mantissas / signs", src, out);
    run<3>("P3, random, 10 x longer (sustained)", src, out, 30000);
    run<3>("P3, random, 10 x longer (sustained), again", src, out, 30000);
    const size_t nb = 1u << 21;
    u4* big; hipMalloc(&big, nb * 16 + 65536); hipMemset(big, 0x3f, nb * 16 + 65536);
    const unsigned bmask = (unsigned)nb - 1;
    runq<0, 0, 2>("Q  limb step, A and B constant", big, out, bmask);
    runq<0, 1, 2>("Q  + six B fragments per step from LDS", big, out, bmask);
    runq<1, 0, 2>("Q  + six A fragments per step from global memory, 2 steps ahead", big, out, bmask);
    runq<1, 1, 2>("Q  + both", big, out, bmask);
    runq<1, 1, 1>("Q  + both, A one step ahead", big, out, bmask);
    runq<1, 1, 2, 1>("Q  both; every workgroup walks the SAME 32 MB stream", big, out, bmask);
    runq<1, 1, 2, 1>("Q  both; every workgroup walks the same 2 MB stream", big, out, (1u << 17) - 1);
    runq<1, 1, 2, 2>("Q  both; same 2 MB stream, wave pairs request the same lines", big, out, (1u << 17) - 1);
    runq<1, 1, 2, 0>("Q  both; private offsets inside 2 MB", big, out, (1u << 17) - 1);
    return 0;
}
