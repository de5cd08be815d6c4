// Developer micro-benchmark: which ingredient of k_conv_f32's main loop costs MFMA time.  8 waves, 64 MFMAs per wave per step,
// ingredients switched on one by one: (B) one barrier per step, (W) 6 ds_write_b128 per thread per step into the other LDS stage,
// (G) 6 global_load_dwordx4 per thread per step (register prefetch, consumed one step later), (S) the staging skewed between the two
// waves of a SIMD.  Prints TFLOP/s per variant.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define LD 36
#define STAGE (384 * LD)
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
template <int B, int W, int G, int S>
__global__ void __launch_bounds__(512, 1) k(const float* __restrict__ src, float* out, int steps, size_t wrap)
{
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 2 * STAGE; i += 512) { unsigned h = (unsigned)(i + blockIdx.x * 7919) * 2654435761u; h ^= h >> 15; lds[i] = (float)(h & 0xFFFF) / 65536.0f - 0.5f; }
    __syncthreads();
    f16v acc[2][2];
    for (int m = 0; m < 2; m++) for (int n = 0; n < 2; n++) for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;
    const int wm = wv & 1, wn = wv >> 1;
    const int aoff = (64 * wm + (lane & 31)) * LD + 4 * (lane >> 5), boff = 128 * LD + (64 * wn + (lane & 31)) * LD + 4 * (lane >> 5);
    f4 pre[6];
    for (int i = 0; i < 6; i++) pre[i] = f4{0.f, 0.f, 0.f, 0.f};
    size_t goff = ((size_t)blockIdx.x * 512 + tid) * 4;
    const int myslot = S ? 2 * (wv >> 2) : 0;
    for (int s = 0; s < steps; s++) {
        const float* base = lds + (s & 1) * STAGE;
        float* other = lds + ((s & 1) ^ 1) * STAGE;
#pragma unroll
        for (int kc = 0; kc < 4; kc++) {
            f4 a[2], b[2];
            a[0] = *(const f4*)(base + aoff + 8 * kc); a[1] = *(const f4*)(base + aoff + 32 * LD + 8 * kc);
            b[0] = *(const f4*)(base + boff + 8 * kc); b[1] = *(const f4*)(base + boff + 32 * LD + 8 * kc);
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int m = 0; m < 2; m++)
#pragma unroll
                    for (int n = 0; n < 2; n++) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][j], b[n][j], acc[m][n], 0, 0, 0);
            if (kc == myslot) {
                if (W) {
#pragma unroll
                    for (int i = 0; i < 6; i++) { const int chunk = tid + 512 * i; *(f4*)(other + (chunk >> 3) * LD + 4 * (chunk & 7)) = pre[i]; }
                }
                if (G == 1) {
#pragma unroll
                    for (int i = 0; i < 6; i++) pre[i] = *(const f4*)(src + ((goff + (size_t)i * 512 * 4 * 64) & (wrap - 1)));
                    goff += 3072 * 4 * 64;
                }
                if (G == 2) {                           // LDS-DMA: 6 x 1 KiB per wave per step straight into the other stage
#pragma unroll
                    for (int i = 0; i < 6; i++)
                        __builtin_amdgcn_global_load_lds((glb_void*)(src + ((goff + (size_t)i * 512 * 4 * 64) & (wrap - 1))), (lds_void*)(other + (wv * 6 + i) * 256), 16, 0, 0);
                    goff += 3072 * 4 * 64;
                }
            }
        }
        if (G == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (B) { __builtin_amdgcn_sched_barrier(0); __syncthreads(); __builtin_amdgcn_sched_barrier(0); }
    }
    float sum = 0;
    for (int m = 0; m < 2; m++) for (int n = 0; n < 2; n++) for (int r = 0; r < 16; r++) sum += acc[m][n][r];
    for (int i = 0; i < 6; i++) sum += pre[i][0];
    out[blockIdx.x * 512 + tid] = sum;
}
template <int B, int W, int G, int S> static void run(const char* name, const float* src, float* d, size_t wrap)
{
    const int steps = 288, grid = 4096;
    hipFuncSetAttribute((const void*)k<B, W, G, S>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<B, W, G, S>), dim3(grid), dim3(512), 2 * STAGE * 4, 0, src, d, steps, wrap);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-34s %8.2f ms  %6.1f TFLOP/s\n", name, ms, (double)grid * 8 * steps * 64 * 4096.0 / ms / 1e9);
}
int main()
{
    float *d, *src;
    const size_t wrap = (size_t)64 << 20;      // floats: 256 MB
    hipMalloc(&d, 4096 * 512 * 4); hipMalloc(&src, wrap * 4 + 4096); hipMemset(src, 0, wrap * 4);
    run<0, 0, 0, 0>("MFMA + LDS fragment reads", src, d, wrap);
    run<1, 0, 0, 0>("+ barrier per step", src, d, wrap);
    run<1, 1, 0, 0>("+ 6 ds_write_b128 / thread / step", src, d, wrap);
    run<1, 1, 1, 0>("+ 6 global loads / thread / step", src, d, wrap);
    run<1, 1, 1, 1>("+ staging skewed between SIMD mates", src, d, wrap);
    run<1, 1, 1, 0>("same, 4 MB source (cache-resident)", src, d, (size_t)1 << 20);
    run<1, 1, 1, 1>("same skewed, 4 MB source", src, d, (size_t)1 << 20);
    run<1, 0, 1, 0>("global loads without LDS writes, 4 MB", src, d, (size_t)1 << 20);
    run<1, 0, 2, 0>("LDS-DMA staging, 256 MB source", src, d, wrap);
    run<1, 0, 2, 1>("LDS-DMA staging skewed, 256 MB", src, d, wrap);
    run<1, 0, 2, 1>("LDS-DMA staging skewed, 4 MB", src, d, (size_t)1 << 20);
    return 0;
}
