// bf16_issue_probe.hip -- what does one vector instruction of each kind ADD to a v_mfma_f32_32x32x16_bf16 stream on gfx950?
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/bf16_issue_probe.hip -o /tmp/p && /tmp/p
// The shape of the bf16x3 Winograd kernel's channel loop: one workgroup of 16 waves per CU (4 per SIMD), every wave runs the
// SAME body -- 12 bf16 MFMAs (four accumulators) and 96 instructions of ONE kind (inline asm: the compiler neither packs nor
// removes them), as two blocks.  Reported per kind: time alone, time with the MFMAs, and what one instruction adds to the MFMA
// stream in SIMD cycles (at the clock the MFMA-only run sustains, 32 cycles per MFMA).
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));

enum Kind { FMA = 0, PKFMA, AND, PERM, MOV, PKADD, EXP, RCP, ADD, CVTPK, DSREAD64, DOT2, NKINDS };
static const char* kNames[] = {"v_fma_f32", "v_pk_fma_f32", "v_and_b32", "v_perm_b32", "v_mov_b32", "v_pk_add_f32", "v_exp_f32",
                               "v_rcp_f32", "v_add_f32", "v_cvt_pk_bf16_f32", "ds_read_b64", "v_dot2_f32_bf16"};

template <int KIND>
__device__ __forceinline__ void four(float& x0, float& x1, float& x2, float& x3, f2& p0, f2& p1, f2& p2, f2& p3, unsigned laddr) {
    if constexpr (KIND == FMA)
        asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
    else if constexpr (KIND == PKFMA)
        asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
    else if constexpr (KIND == AND)
        asm volatile("v_and_b32 %0, 0xffff0000, %0\n v_and_b32 %1, 0xffff0000, %1\n v_and_b32 %2, 0xffff0000, %2\n v_and_b32 %3, 0xffff0000, %3\n" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
    else if constexpr (KIND == PERM)
        asm volatile("v_perm_b32 %0, %1, %0, %4\n v_perm_b32 %1, %2, %1, %4\n v_perm_b32 %2, %3, %2, %4\n v_perm_b32 %3, %0, %3, %4\n" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "s"(0x07060302u));
    else if constexpr (KIND == MOV)
        asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0\n" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
    else if constexpr (KIND == PKADD)
        asm volatile("v_pk_add_f32 %0, %0, %0\n v_pk_add_f32 %1, %1, %1\n v_pk_add_f32 %2, %2, %2\n v_pk_add_f32 %3, %3, %3\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
    else if constexpr (KIND == EXP)
        asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
    else if constexpr (KIND == RCP)
        asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
    else if constexpr (KIND == ADD)
        asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3\n" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
    else if constexpr (KIND == CVTPK)
        asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %0\n" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
    else if constexpr (KIND == DOT2)
        asm volatile("v_dot2_f32_bf16 %0, %1, %4, %0\n v_dot2_f32_bf16 %1, %2, %4, %1\n v_dot2_f32_bf16 %2, %3, %4, %2\n v_dot2_f32_bf16 %3, %0, %4, %3\n" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "s"(0x0000bf80u));
    else if constexpr (KIND == DSREAD64)
        asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:512\n ds_read_b64 %2, %4 offset:1024\n ds_read_b64 %3, %4 offset:1536\n s_waitcnt lgkmcnt(0)\n"
                     : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3) : "v"(laddr) : "memory");
}

template <int M, int V, int KIND>
__global__ void __launch_bounds__(1024) probe(int iters, float* out) {
    __shared__ float lds[16384];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += 1024) lds[i] = lane;
    __syncthreads();
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k)
        for (int j = 0; j < 16; ++j) acc[k][j] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3f80 + lane); b[j] = (short)(0x3f00 + j); }
    float x0 = lane * 1e-3f, x1 = 0.5f, x2 = 0.25f, x3 = 0.125f;
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x1, x2}, p3 = {x3, x0};
    const unsigned laddr = (unsigned)((threadIdx.x >> 6) * 1024 + lane * 2) * 4u;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int m = 0; m < M; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
        for (int v = 0; v < V / 4; ++v) four<KIND>(x0, x1, x2, x3, p0, p1, p2, p3, laddr);
    }
    float r = x0 + x1 + x2 + x3 + p0.x + p1.y + p2.x + p3.y;
    for (int k = 0; k < 4; ++k) r += acc[k][0] + acc[k][7];
    out[blockIdx.x * 1024 + threadIdx.x] = r + lds[lane];
}

// x - bf16(x) by one v_dot2_f32_bf16: (hi, .) . (-1, 0) + x -- exact?  Compared with the subtraction in fp32
__global__ void dot2_check(const float* x, unsigned* bad, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x0 = x[i], x1 = x[(i + 1) % n];
    unsigned hp;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hp) : "v"(x0), "v"(x1));
    float r0, r1;
    asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(r0) : "v"(hp), "s"(0x0000bf80u), "v"(x0));
    asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(r1) : "v"(hp), "s"(0xbf800000u), "v"(x1));
    const float e0 = x0 - __uint_as_float(hp << 16), e1 = x1 - __uint_as_float(hp & 0xffff0000u);
    if (__float_as_uint(r0) != __float_as_uint(e0) || __float_as_uint(r1) != __float_as_uint(e1)) atomicAdd(bad, 1u);
}

template <int M, int V, int KIND>
static float run(int iters, float* d_out) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<M, V, KIND>), dim3(256), dim3(1024), 0, 0, iters, d_out);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<M, V, KIND>), dim3(256), dim3(1024), 0, 0, iters, d_out);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 3 * 1e3f;
}

template <int KIND>
static void row(int iters, float* d_out, float t_m, double ghz) {
    const float t_a = run<0, 96, KIND>(iters, d_out), t_b = run<12, 96, KIND>(iters, d_out);
    auto cyc = [&](float us) { return us * 1e3 * ghz / iters; };
    printf("%-20s alone %7.1f us (%5.2f cycles / instr / SIMD slot)   with the MFMAs %7.1f us   added per instruction %5.2f cycles   hidden %4.0f %%\n",
           kNames[KIND], t_a, cyc(t_a) / (96 * 4), t_b, cyc(t_b - t_m) / (96 * 4), 100.0 * (1.0 - (t_b - t_m) / t_a));
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int iters = 2048;
    float* d_out;
    (void)hipMalloc(&d_out, 256 * 1024 * sizeof(float));
    const float t_m = run<12, 0, FMA>(iters, d_out);
    const double ghz = (double)iters * 12 * 32 * 4 / (t_m * 1e3);
    printf("4 waves per SIMD, 12 bf16 MFMAs per wave and iteration alone: %.1f us -> %.2f GHz at 32 cycles per MFMA; 96 instructions of a kind per wave and iteration\n", t_m, ghz);
    row<FMA>(iters, d_out, t_m, ghz);
    row<ADD>(iters, d_out, t_m, ghz);
    row<PKFMA>(iters, d_out, t_m, ghz);
    row<PKADD>(iters, d_out, t_m, ghz);
    row<AND>(iters, d_out, t_m, ghz);
    row<PERM>(iters, d_out, t_m, ghz);
    row<MOV>(iters, d_out, t_m, ghz);
    row<CVTPK>(iters, d_out, t_m, ghz);
    row<EXP>(iters, d_out, t_m, ghz);
    row<RCP>(iters, d_out, t_m, ghz);
    row<DSREAD64>(iters, d_out, t_m, ghz);
    row<DOT2>(iters, d_out, t_m, ghz);
    {   // residual by dot2: bit-equal to the fp32 subtraction on 4M values of mixed magnitude?
        const int n = 1 << 22;
        float* hx = new float[n];
        unsigned seed = 12345u;
        for (int i = 0; i < n; ++i) {
            seed = seed * 1664525u + 1013904223u;
            const float u = (float)(seed >> 8) * (1.0f / 16777216.0f) - 0.5f;
            seed = seed * 1664525u + 1013904223u;
            const int e = (int)(seed >> 27) - 16;
            hx[i] = ldexpf(u, e);
        }
        float* dx; unsigned* dbad; unsigned hbad = 0;
        (void)hipMalloc(&dx, n * sizeof(float)); (void)hipMalloc(&dbad, 4);
        (void)hipMemcpy(dx, hx, n * sizeof(float), hipMemcpyHostToDevice); (void)hipMemset(dbad, 0, 4);
        hipLaunchKernelGGL(dot2_check, dim3(n / 256), dim3(256), 0, 0, dx, dbad, n);
        (void)hipMemcpy(&hbad, dbad, 4, hipMemcpyDeviceToHost);
        printf("x - bf16_rne(x) by v_dot2_f32_bf16 against the fp32 subtraction: %u of %d pairs differ\n", hbad, n);
    }
    return 0;
}
