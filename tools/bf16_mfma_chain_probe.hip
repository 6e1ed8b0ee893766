// bf16_mfma_chain_probe.hip -- what does a DEPENDENT chain of v_mfma_f32_32x32x16_bf16 cost?
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/bf16_mfma_chain_probe.hip -o /tmp/p && /tmp/p
// W waves per SIMD, every wave issues 12 MFMAs per iteration over NACC accumulators round-robin (NACC = 1: every MFMA
// accumulates onto the result of the one before it; 2: of the one two before; ...).  Reported: cycles per MFMA and SIMD at
// the clock the 4-accumulator, 4-wave run sustains.
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ void __launch_bounds__(1024) probe(int iters, float* out) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[NACC];
    for (int k = 0; k < NACC; ++k)
        for (int j = 0; j < 16; ++j) acc[k][j] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3f80 + lane); b[j] = (short)(0x3f00 + j); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int m = 0; m < 12; ++m) acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % NACC], 0, 0, 0);
    }
    float r = 0.f;
    for (int k = 0; k < NACC; ++k) r += acc[k][0] + acc[k][7];
    out[blockIdx.x * 1024 + threadIdx.x] = r;
}

template <int NACC>
static float run(int waves, int iters, float* d_out) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<NACC>), dim3(256), dim3(64 * waves), 0, 0, iters, d_out);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<NACC>), dim3(256), dim3(64 * waves), 0, 0, iters, d_out);
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
    const float t_ref = run<4>(16, iters, d_out);
    const double ghz = (double)iters * 12 * 32 * 4 / (t_ref * 1e3);
    printf("4 waves per SIMD, 4 accumulators: %.1f us -> %.2f GHz at 32 cycles per MFMA\n", t_ref, ghz);
    for (int waves : {4, 8, 16}) {
        const int wps = waves / 4;
        auto cyc = [&](float us) { return us * 1e3 * ghz / ((double)iters * 12 * wps); };
        printf("%d wave(s) per SIMD: cycles per MFMA and SIMD with 1 / 2 / 3 / 4 accumulators in rotation: %6.1f %6.1f %6.1f %6.1f\n", wps,
               cyc(run<1>(waves, iters, d_out)), cyc(run<2>(waves, iters, d_out)), cyc(run<3>(waves, iters, d_out)), cyc(run<4>(waves, iters, d_out)));
    }
    return 0;
}
