// mfma_valu_probe.hip -- does an f32 MFMA stream share the SIMD's vector ALU with VALU work of a partner wave?
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_probe.hip -o /tmp/probe && /tmp/probe
// Workgroup = 8 waves (one CU): waves 0-3 (one per SIMD) run role A, waves 4-7 role B (the SIMD partners).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum Role { IDLE = 0, MFMA32 = 1, VALU_FMA = 2, LDS_READ = 3, LDS_WRITE = 4, VALU_EXP = 5, MFMA_BF16 = 6 };
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(512) probe(int roleA, int roleB, int iters, float* out) {
    __shared__ float lds[16384];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int role = wave < 4 ? roleA : roleB;
    float r = 0.0f;
    for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = (float)i * 1e-6f;
    __syncthreads();
    if (role == MFMA32) {
        f32x16 acc[4];
        for (int k = 0; k < 4; ++k)
            for (int j = 0; j < 16; ++j) acc[k][j] = 0.f;
        float a = lane * 1e-3f, b = 1.0f + lane * 1e-4f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k], 0, 0, 0);
        }
        for (int k = 0; k < 4; ++k) r += acc[k][0] + acc[k][7];
    } else if (role == MFMA_BF16) {
        f32x16 acc[4];
        for (int k = 0; k < 4; ++k)
            for (int j = 0; j < 16; ++j) acc[k][j] = 0.f;
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3f80 + lane); b[j] = (short)(0x3f00 + j); }
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
        }
        for (int k = 0; k < 4; ++k) r += acc[k][0] + acc[k][7];
    } else if (role == VALU_FMA) {
        float x[8];
        for (int k = 0; k < 8; ++k) x[k] = lane * 1e-3f + k;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                for (int k = 0; k < 8; ++k) x[k] = __builtin_fmaf(x[k], 0.999f, 1e-3f);     // 32 VALU per iteration
        }
        for (int k = 0; k < 8; ++k) r += x[k];
    } else if (role == VALU_EXP) {
        float x[8];
        for (int k = 0; k < 8; ++k) x[k] = lane * 1e-3f + k;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = __builtin_amdgcn_exp2f(x[k]) * 0.5f;          // 8 transcendental + 8 mul
        }
        for (int k = 0; k < 8; ++k) r += x[k];
    } else if (role == LDS_READ) {
        const float* p = lds + lane;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) r += p[(k * 64 + (i & 63) * 64) & 16383];           // 16 ds_read_b32
        }
    } else if (role == LDS_WRITE) {
        float* p = lds + wave * 1024 + lane;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) p[k * 64] = r + i;                                   // 16 ds_write_b32
        }
        __syncthreads();
        r += p[0];
    }
    if (role != LDS_WRITE) __syncthreads();
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

static float run(int a, int b, int iters, float* d_out, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 0, 0, a, b, iters, d_out);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 0, 0, a, b, iters, d_out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1e3f;
}

int main() {
    const int blocks = 256, iters = 4096;
    float* d_out;
    hipMalloc(&d_out, blocks * 512 * sizeof(float));
    const char* names[] = {"idle", "mfma_f32_32x32x2", "valu_fma", "lds_read", "lds_write", "valu_exp", "mfma_bf16_32x32x16"};
    const int pairs[][2] = {{1, 0}, {2, 0}, {3, 0}, {4, 0}, {5, 0}, {6, 0}, {1, 1}, {1, 2}, {1, 3}, {1, 4}, {1, 5}, {6, 2}, {6, 3}, {2, 2}, {2, 3}};
    printf("%-22s %-22s %10s\n", "waves 0-3", "waves 4-7", "us");
    for (auto& pr : pairs) printf("%-22s %-22s %10.1f\n", names[pr[0]], names[pr[1]], run(pr[0], pr[1], iters, d_out, blocks));
    // reference: 4096 iters x 4 MFMA x 64 cycles = 1.05 M cycles = 437 us at 2.4 GHz
    return 0;
}
