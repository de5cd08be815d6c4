// Developer micro-benchmark: the rate v_mfma_f32_32x32x2_f32 sustains on this box (a) from registers, (b) with the LDS fragment
// reads of k_conv_f32 (4 ds_read_b128 per 16 MFMAs), 2 waves per SIMD, every CU busy.  Prints TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int LDSREADS, int RANDOM>
__global__ void __launch_bounds__(512, 1) k(float* out, int iters)
{
    __shared__ __align__(16) float lds[384 * 36];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 384 * 36; i += 512) { unsigned h = (unsigned)(i + blockIdx.x * 7919) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; lds[i] = RANDOM ? ((float)(h & 0xFFFFFF) / 8388608.0f - 1.0f) : (float)(i % 13) * 0.01f; }
    __syncthreads();
    f16v acc[2][2];
    for (int m = 0; m < 2; m++) for (int n = 0; n < 2; n++) for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;
    f4 a[2] = {{1.f, 2.f, 3.f, 4.f}, {0.5f, 0.25f, 0.125f, 1.f}}, b[2] = {{1.f, 1.f, 2.f, 2.f}, {3.f, 1.f, 2.f, 1.f}};
    const float* pa = lds + (lane & 31) * 36 + 4 * (lane >> 5);
    for (int it = 0; it < iters; it++) {
        if (LDSREADS) {
            a[0] = *(const f4*)(pa + 8 * (it & 3)); a[1] = *(const f4*)(pa + 32 * 36 + 8 * (it & 3));
            b[0] = *(const f4*)(pa + 128 * 36 + 8 * (it & 3)); b[1] = *(const f4*)(pa + 160 * 36 + 8 * (it & 3));
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int n = 0; n < 2; n++) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][j], b[n][j], acc[m][n], 0, 0, 0);
    }
    float s = 0;
    for (int m = 0; m < 2; m++) for (int n = 0; n < 2; n++) for (int r = 0; r < 16; r++) s += acc[m][n][r];
    out[blockIdx.x * 512 + tid] = s;
}
int main()
{
    float* d; hipMalloc(&d, 4096 * 512 * 4);
    const int iters = 8192, grid = 4096;
    for (int v = 0; v < 3; v++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (v == 0) hipLaunchKernelGGL((k<0, 0>), dim3(grid), dim3(512), 0, 0, d, iters);
            else if (v == 1) hipLaunchKernelGGL((k<1, 0>), dim3(grid), dim3(512), 0, 0, d, iters);
            else hipLaunchKernelGGL((k<1, 1>), dim3(grid), dim3(512), 0, 0, d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = (double)grid * 8 * iters * 16 * 4096.0;
        printf("%s: %.2f ms, %.1f TFLOP/s\n", v == 0 ? "registers only, constant operands" : v == 1 ? "LDS fragment reads, low-entropy operands" : "LDS fragment reads, random operands", ms, fl / ms / 1e9);
    }
    return 0;
}
