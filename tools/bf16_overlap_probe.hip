// bf16_overlap_probe.hip -- how much vector work hides under v_mfma_f32_32x32x16_bf16 on gfx950?
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/bf16_overlap_probe.hip -o /tmp/p && /tmp/p
// One workgroup of W waves per CU (64 KB of LDS each, 256 workgroups); every wave runs the SAME loop body: M bf16 MFMAs (four
// independent accumulators) and V independent v_fma_f32 (inline asm, so the compiler neither packs nor removes them), either
// as two blocks (all MFMAs, then all FMAs) or interleaved.  Reported: cycles per iteration and SIMD at the sustained clock
// measured by the MFMA-only row (32 cycles per MFMA).
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define FMA4(x0, x1, x2, x3) asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n" \
                                          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3))

template <int M, int V, bool INTER>
__global__ void __launch_bounds__(1024) probe(int iters, float* out) {
    __shared__ float lds[16384];
    const int lane = threadIdx.x & 63;
    lds[threadIdx.x] = lane;
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k)
        for (int j = 0; j < 16; ++j) acc[k][j] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3f80 + lane); b[j] = (short)(0x3f00 + j); }
    float x0 = lane * 1e-3f, x1 = 0.5f, x2 = 0.25f, x3 = 0.125f;
    for (int i = 0; i < iters; ++i) {
        if constexpr (INTER) {
            constexpr int VPM = M ? V / M : 0;              // FMAs after each MFMA (multiples of 4)
#pragma unroll
            for (int m = 0; m < M; ++m) {
                acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < VPM / 4; ++v) FMA4(x0, x1, x2, x3);
            }
        } else {
#pragma unroll
            for (int m = 0; m < M; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < V / 4; ++v) FMA4(x0, x1, x2, x3);
        }
    }
    float r = x0 + x1 + x2 + x3;
    for (int k = 0; k < 4; ++k) r += acc[k][0] + acc[k][7];
    out[blockIdx.x * 1024 + threadIdx.x] = r + lds[lane];
}

template <int M, int V, bool INTER>
static float run(int waves, int iters, float* d_out) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<M, V, INTER>), dim3(256), dim3(64 * waves), 0, 0, iters, d_out);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<M, V, INTER>), dim3(256), dim3(64 * waves), 0, 0, iters, d_out);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 3 * 1e3f;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int iters = 2048;
    float* d_out;
    (void)hipMalloc(&d_out, 256 * 1024 * sizeof(float));
    for (int waves : {4, 8, 16}) {
        const int wps = waves / 4;
        const float t_m = run<12, 0, false>(waves, iters, d_out);
        const double ghz = (double)iters * 12 * 32 * wps / (t_m * 1e3);
        auto cyc = [&](float us) { return us * 1e3 * ghz / iters; };       // cycles per iteration and SIMD (all its waves)
        printf("%d waves per SIMD: 12 bf16 MFMAs per wave and iteration alone %.1f us -> %.2f GHz at 32 cycles per MFMA\n", wps, t_m, ghz);
        printf("   48 v_fma alone                      %7.0f cycles / iteration / SIMD\n", cyc(run<0, 48, false>(waves, iters, d_out)));
        printf("   96 v_fma alone                      %7.0f\n", cyc(run<0, 96, false>(waves, iters, d_out)));
        printf("   12 MFMA (=%d) then 48 v_fma       %7.0f\n", 384 * wps, cyc(run<12, 48, false>(waves, iters, d_out)));
        printf("   12 MFMA, 48 v_fma interleaved       %7.0f\n", cyc(run<12, 48, true>(waves, iters, d_out)));
        printf("   12 MFMA then 96 v_fma               %7.0f\n", cyc(run<12, 96, false>(waves, iters, d_out)));
        printf("   12 MFMA, 96 v_fma interleaved       %7.0f\n", cyc(run<12, 96, true>(waves, iters, d_out)));
    }
    return 0;
}
