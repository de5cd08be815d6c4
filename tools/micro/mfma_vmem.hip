// Developer micro-benchmark: what do vector-memory reads cost a wave that is issuing bf16 MFMAs back to back?  (k_conv3x3_b3 loses a quarter
// of its time to six 16-byte-per-lane weight requests per 24 MFMAs although nothing waits for them.)  256 threads, two workgroups per CU, per
// step and wave 24 v_mfma_f32_32x32x16_bf16 on four accumulators plus L reads of 16 bytes per lane in one of several forms, from a 1 MB
// (L2-resident) buffer; the values read become MFMA operands two steps later, as in the real kernel.  Prints bf16 TFLOP/s per variant.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_vmem mfma_vmem.hip && ./mfma_vmem
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

// MODE 0 none; 1 global_load_dwordx4, consumed two steps later; 2 the same, never consumed; 3 global_load_dwordx2 x 2L; 4 ds_read_b128 x L from LDS;
// 5 global loads with the SAME address in every wave of the CU (L1 hits); 6 loads issued in ONE burst at the step start instead of spread between MFMAs
template <int L, int MODE>
__global__ void __launch_bounds__(256, 2) k(const u4* __restrict__ src, float* out, int steps, unsigned mask, unsigned long long* clk)
{
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    __shared__ u4 lds[1024];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 1024; i += 256) lds[i] = u4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    __syncthreads();
    f16v acc[4];
    for (int a = 0; a < 4; a++) for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
    u4 ring[3][L > 0 ? L : 1];
    for (int s = 0; s < 3; s++) for (int i = 0; i < (L > 0 ? L : 1); i++) ring[s][i] = u4{0x3f803f80u + tid, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    // MODE 7: every workgroup walks the SAME stream (a weight tensor read by all pixel tiles); MODE 8: the same, each wave of a workgroup a different quarter of the step's lines
    unsigned off = (MODE == 5 ? 0u : MODE == 7 ? (unsigned)lane : MODE == 8 ? (unsigned)tid : (blockIdx.x * 256u + tid)) & mask;
    u4 sink = u4{0, 0, 0, 0};
    auto loads = [&](const int set) {
        if (MODE == 0 || L == 0) return;
#pragma unroll
        for (int i = 0; i < L; i++) {
            if (MODE == 4) ring[set][i] = lds[(lane + 64 * i + off) & 1023];
            else if (MODE == 3) { const u2* p = (const u2*)(src + ((off + 64u * i) & mask)); u2 x = p[0], y = p[1]; ring[set][i] = u4{x[0], x[1], y[0], y[1]}; }
            else ring[set][i] = src[(off + (MODE == 5 ? lane : 0) + (MODE == 8 ? 256u : 64u) * i) & mask];
        }
        off = (off + (MODE == 8 ? 256u : 64u) * L) & mask;
    };
    auto step = [&](const int ws) {            // literal
        const u4* cur = ring[ws];
        b8 A = __builtin_bit_cast(b8, cur[0]);
        b8 B = __builtin_bit_cast(b8, cur[L > 1 ? 1 : 0]);
        b8 A3[3], B3[3][2];
        if (MODE == 10 || MODE == 11) {         // distinct operand registers, no LDS: 10 = B from the ring too, 11 = B prefetched from LDS one step ahead
#pragma unroll
            for (int l = 0; l < 3; l++) { A3[l] = __builtin_bit_cast(b8, cur[2 * l]); B3[l][0] = __builtin_bit_cast(b8, cur[(2 * l + 1) % L]); B3[l][1] = __builtin_bit_cast(b8, ring[(ws + 1) % 3][(2 * l + 1) % L]); }
        }
        if (MODE == 9) {                        // the real kernel's operand traffic: three A limbs from the global ring, six B fragments from LDS per step
#pragma unroll
            for (int l = 0; l < 3; l++) {
                A3[l] = __builtin_bit_cast(b8, cur[2 * l]);
                B3[l][0] = __builtin_bit_cast(b8, lds[(lane + 64 * l + ws * 200) & 1023]);
                B3[l][1] = __builtin_bit_cast(b8, lds[(lane + 64 * l + ws * 200 + 300) & 1023]);
            }
        }
        if (MODE == 6) { loads((ws + 2) % 3); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int g = 0; g < 6; g++) {
#pragma unroll
            for (int a = 0; a < 4; a++) {
                if (MODE == 9 || MODE == 10) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A3[g % 3], B3[(g + a) % 3][a & 1], acc[a], 0, 0, 0);
                else acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[a], 0, 0, 0);
            }
            if (g == 0 && MODE != 6) { loads((ws + 2) % 3); }
        }
        if (MODE == 2) { for (int i = 0; i < (L > 0 ? L : 1); i++) for (int e = 0; e < 4; e++) sink[e] ^= 0; }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int s = 0; s < steps; s += 3) { step(0); step(1); step(2); }
    float t = 0.f;
    for (int a = 0; a < 4; a++) for (int r = 0; r < 16; r++) t += acc[a][r];
    if (MODE != 1 && MODE != 6 && MODE != 7 && MODE != 8 && MODE != 9 && MODE != 10) for (int s = 0; s < 3; s++) for (int i = 0; i < (L > 0 ? L : 1); i++) t += (float)(ring[s][i][0] & 1u);
    out[blockIdx.x * 256 + tid] = t + (float)sink[0];
    if (blockIdx.x == 0 && tid == 0 && clk) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}

template <int L, int MODE>
static void run(const char* name, const u4* src, float* out, unsigned mask)
{
    static unsigned long long* clk = nullptr;
    if (!clk) hipMalloc(&clk, 16);
    const int grid = 512 * 8, steps = 3000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<L, MODE>), dim3(grid), dim3(256), 0, 0, src, out, 30, mask, (unsigned long long*)nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<L, MODE>), dim3(grid), dim3(256), 0, 0, src, out, steps, mask, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)grid * 4 * steps * 24 * 32768.0;
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("%-64s L=%d  %8.3f ms  %7.1f TFLOP/s bf16  (%.1f %% of 2500)  workgroup 0: %.0f shader cycles per 100 MHz tick x 100 = %.0f MHz\n", name, L, ms, fl / (ms * 1e-3) / 1e12,
           fl / (ms * 1e-3) / 2.5e15 * 100, (double)h[0] / (double)h[1], (double)h[0] / (double)h[1] * 100.0);
}

int main()
{
    const size_t n = 1u << 16;                  // 64 K x 16 B = 1 MB
    u4* src; float* out;
    hipMalloc(&src, n * 16 + 65536); hipMemset(src, 0x3f, n * 16 + 65536);
    hipMalloc(&out, (size_t)512 * 8 * 256 * 4);
    const unsigned mask = (unsigned)n - 1;
    const size_t nb = 1u << 21;                 // 2 M x 16 B = 32 MB
    u4* big; hipMalloc(&big, nb * 16 + 65536); hipMemset(big, 0x3f, nb * 16 + 65536);
    const unsigned bmask = (unsigned)nb - 1;
    run<0, 0>("no reads", src, out, mask);
    run<1, 1>("global_load_dwordx4, consumed", src, out, mask);
    run<2, 1>("global_load_dwordx4, consumed", src, out, mask);
    run<6, 1>("global_load_dwordx4, consumed", src, out, mask);
    run<12, 1>("global_load_dwordx4, consumed", src, out, mask);
    run<6, 2>("global_load_dwordx4, not consumed", src, out, mask);
    run<6, 3>("global_load_dwordx2 x 2", src, out, mask);
    run<6, 4>("ds_read_b128 from LDS", src, out, mask);
    run<12, 4>("ds_read_b128 from LDS", src, out, mask);
    run<6, 5>("global_load_dwordx4, same 6 KB in every wave (L1 hits)", src, out, mask);
    run<6, 6>("global_load_dwordx4, one burst at the step start", src, out, mask);
    run<6, 7>("global_load_dwordx4, all workgroups walk the same 1 MB", src, out, mask);
    run<6, 7>("global_load_dwordx4, all workgroups walk the same 32 MB", big, out, bmask);
    run<6, 8>("... the four waves of a workgroup different lines of it", big, out, bmask);
    run<6, 1>("global_load_dwordx4, private offsets in 32 MB", big, out, bmask);
    run<6, 9>("6 global A fragments + 6 LDS B fragments per step", big, out, bmask);
    run<6, 10>("the same operand pattern, B fragments from registers (no LDS)", big, out, bmask);
    return 0;
}
