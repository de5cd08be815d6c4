// Developer micro-benchmark (round 4): can the packed-f32 vector FMAs (v_pk_fma_f32: 256 FLOPs per wave instruction, 4 cycles -- the same
// 157 TFLOP/s peak as the f32 MFMA) run BESIDE v_mfma_f32_32x32x2_f32 on the same SIMDs?  One 8-wave workgroup per CU: waves 0-3 (one per SIMD) issue
// MFMAs from registers, waves 4-7 (the second wave of each SIMD) issue packed FMAs on 32 independent accumulators.  Modes: MFMA waves alone, VALU waves
// alone, both.  Prints ms and TFLOP/s of each pipe and the shader clock seen by s_memtime (a 100 MHz counter) -- a second pipe bought with clock is no gain.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_valu_coissue tools/micro/mfma_valu_coissue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(512, 1) k(float* out, unsigned long long* clk, int itersM, int itersV, int mode, float seed)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long m0 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    if (wv < 4) {
        if (mode == 1) return;
        f16v acc[2][2];
        for (int m = 0; m < 2; m++) for (int n = 0; n < 2; n++) for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;
        unsigned h = (unsigned)(tid + blockIdx.x * 7919) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        float a[2][4], b[2][4];
        for (int m = 0; m < 2; m++) for (int j = 0; j < 4; j++) { h = h * 1664525u + 1013904223u; a[m][j] = (float)(h & 0xFFFFFF) / 8388608.0f - 1.0f; h = h * 1664525u + 1013904223u; b[m][j] = ((float)(h & 0xFFFFFF) / 8388608.0f - 1.0f) * seed; }
        for (int it = 0; it < itersM; it++) {
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int m = 0; m < 2; m++)
#pragma unroll
                    for (int n = 0; n < 2; n++) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][j], b[n][j], acc[m][n], 0, 0, 0);
        }
        for (int m = 0; m < 2; m++) for (int n = 0; n < 2; n++) for (int r = 0; r < 16; r++) s += acc[m][n][r];
    } else {
        if (mode == 0) return;
        f2 acc[32], mul[4], add[4];
        unsigned h = (unsigned)(tid + blockIdx.x * 104729) * 2654435761u; h ^= h >> 15;
        for (int i = 0; i < 32; i++) { h = h * 1664525u + 1013904223u; acc[i] = f2{(float)(h & 0xFFFF) / 65536.f, (float)(h >> 16) / 65536.f}; }
        for (int i = 0; i < 4; i++) { h = h * 1664525u + 1013904223u; mul[i] = f2{0.99f + (float)(h & 0xFF) * 1e-5f, 0.98f + (float)((h >> 8) & 0xFF) * 1e-5f} * seed; add[i] = f2{0.013f * (i + 1), 0.017f * (i + 1)}; }
        for (int it = 0; it < itersV; it++) {
#pragma unroll
            for (int i = 0; i < 32; i++) acc[i] = __builtin_elementwise_fma(acc[i], mul[i & 3], add[i & 3]);
        }
        for (int i = 0; i < 32; i++) s += acc[i][0] + acc[i][1];
    }
    out[blockIdx.x * 512 + tid] = s;
    if (lane == 0) { clk[(blockIdx.x * 8 + wv) * 2] = __builtin_amdgcn_s_memtime() - t0; clk[(blockIdx.x * 8 + wv) * 2 + 1] = __builtin_amdgcn_s_memrealtime() - m0; }
}
int main()
{
    const int grid = 256 * 4;
    float* d; hipMalloc(&d, grid * 512 * 4);
    unsigned long long* c; hipMalloc(&c, grid * 8 * 2 * 8);
    unsigned long long* hc = new unsigned long long[grid * 16];
    const int itersM = 16384;                      // 16 MFMAs of 64 cycles per iteration
    for (int pass = 0; pass < 2; pass++)
    for (int mode = 0; mode < 4; mode++) {
        // mode 3 = both, with the VALU waves given HALF the issue slots' worth of work (a staging-like duty cycle)
        const int m = mode == 3 ? 2 : mode;
        const int itersV = mode == 3 ? itersM * 4 : itersM * 8;          // 32 packed FMAs of 4 cycles per iteration: 8 iterations = one MFMA iteration's 1024 cycles
        hipMemset(c, 0, grid * 16 * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, d, c, itersM, itersV, m, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(hc, c, grid * 16 * 8, hipMemcpyDeviceToHost);
        double cyc = 0, mt = 0; int nw = 0;
        for (int i = 0; i < grid * 8; i++) if (hc[2 * i + 1]) { cyc += (double)hc[2 * i]; mt += (double)hc[2 * i + 1]; nw++; }
        const double flM = m == 1 ? 0 : (double)grid * 4 * itersM * 16 * 4096.0, flV = m == 0 ? 0 : (double)grid * 4 * itersV * 32 * 256.0;
        if (pass) printf("%-34s %8.2f ms   MFMA %6.1f TFLOP/s   packed-f32 VALU %6.1f TFLOP/s   sum %6.1f   shader clock %.2f GHz\n",
               mode == 0 ? "MFMA waves alone" : mode == 1 ? "VALU waves alone" : mode == 2 ? "both, equal issue time" : "both, VALU waves at half duty", ms, flM / ms / 1e9, flV / ms / 1e9,
               (flM + flV) / ms / 1e9, nw ? cyc / mt * 0.1 : 0.0);
    }
    return 0;
}
