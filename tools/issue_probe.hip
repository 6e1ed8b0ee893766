// issue_probe.hip -- what does one instruction of each kind cost an f32-MFMA stream that shares its SIMD?
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/issue_probe.hip -o /tmp/issue_probe && /tmp/issue_probe
// One workgroup of 8 waves per CU (64 KB of LDS each, 256 workgroups): waves 0-3 (one per SIMD) run role A, waves 4-7 (their
// SIMD partners) role B.  Every role is an inline-asm block of 16 instructions of ONE kind per iteration, so that the
// compiler adds nothing (the round-2 probe's "lds_read" role carried 16 v_add per 16 reads).  Reported per pairing:
// time alone, time beside the MFMA stream, and the time ADDED to the MFMA stream per partner instruction, in SIMD cycles.
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum Role { IDLE = 0, MFMA = 1, VADD = 2, VEXP = 3, DSREAD = 4, DSREAD2ST = 5, DSREADB64X2 = 6, DSWRITE = 7, DSWRITE2ST = 8,
            SALU = 9, VMEM = 10, VMEMX4 = 11, VFMA_SGPR = 12, GLOBAL_SADDR = 13, GLOBAL_VADDR = 14, GLOBAL_X4_SADDR = 15,
            DSREADB128 = 16, DSWRITEB128 = 17, DSWRITEB64 = 18, VMEMX2 = 19, LDSDMA = 20, SLOAD = 21, VPKFMA = 22, VMEM_NOUSE = 23, VMEM_SOFF = 24, VMEM_ADD = 25, VMEM_BUILTIN_NOADD = 26, VMEM_SOFF_L1 = 27,
            NROLES = 28 };
static const char* kNames[] = {"idle", "mfma_f32_32x32x2", "v_add_f32", "v_exp_f32", "ds_read_b32", "ds_read2st64_b32",
                               "ds_read2_b64", "ds_write_b32", "ds_write2st64_b32", "s_add_u32", "buffer_load_dword",
                               "buffer_load_dwordx4", "v_fma_f32 (sgpr)", "global_load_dword saddr", "global_load_dword vaddr",
                               "global_load_dwordx4 saddr", "ds_read_b128", "ds_write_b128", "ds_write_b64", "buffer_load_dwordx2",
                               "buffer_load_dword lds", "s_load_dwordx4", "v_pk_fma_f32", "buffer_load_dword (asm)",
                               "bld asm soffset 16KB/wave", "bld asm + v_add of result", "bld builtin, no add", "bld asm soffset 4KB/wave"};

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP4(x) x x x x

__global__ void __launch_bounds__(512) probe(int roleA, int roleB, int iters, const float* __restrict__ src, float* out) {
    __shared__ float lds[16384];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int role = wave < 4 ? roleA : roleB;
    for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = (float)i * 1e-6f;
    __syncthreads();
    float r = 0.0f;
    const unsigned laddr = (unsigned)(wave * 2048 + lane) * 4u;      // a private 8 KB window per wave
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, -1, 0x00020000);
    if (role == MFMA) {
        f32x16 acc[4];
        for (int k = 0; k < 4; ++k)
            for (int j = 0; j < 16; ++j) acc[k][j] = 0.f;
        float a = lane * 1e-3f, b = 1.0f + lane * 1e-4f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k], 0, 0, 0);
        }
        for (int k = 0; k < 4; ++k) r += acc[k][0] + acc[k][7];
    } else if (role == VADD) {
        float x0 = lane, x1 = 1, x2 = 2, x3 = 3;
        for (int i = 0; i < iters; ++i)
            asm volatile(REP4("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        r = x0 + x1 + x2 + x3;
    } else if (role == VFMA_SGPR) {
        float x0 = lane, x1 = 1, x2 = 2, x3 = 3;
        float s = 1.0f;
        asm volatile("" : "+s"(s));
        for (int i = 0; i < iters; ++i)
            asm volatile(REP4("v_fma_f32 %0, %4, %0, %0\n v_fma_f32 %1, %4, %1, %1\n v_fma_f32 %2, %4, %2, %2\n v_fma_f32 %3, %4, %3, %3\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "s"(s));
        r = x0 + x1 + x2 + x3;
    } else if (role == VEXP) {
        float x0 = lane * 1e-3f, x1 = 0.1f, x2 = 0.2f, x3 = 0.3f;
        for (int i = 0; i < iters; ++i)
            asm volatile(REP4("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        r = x0 + x1 + x2 + x3;
    } else if (role == DSREAD) {
        float x0, x1, x2, x3;
        for (int i = 0; i < iters; ++i)
            asm volatile(REP4("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:256\n ds_read_b32 %2, %4 offset:512\n ds_read_b32 %3, %4 offset:768\n")
                         "s_waitcnt lgkmcnt(0)\n"
                         : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(laddr) : "memory");
        r = x0 + x1 + x2 + x3;
    } else if (role == DSREAD2ST) {
        float x0, x1, x2, x3, x4, x5, x6, x7;
        for (int i = 0; i < iters; ++i)
            asm volatile(REP4("ds_read2st64_b32 %0, %4 offset1:1\n ds_read2st64_b32 %1, %4 offset0:2 offset1:3\n"
                              "ds_read2st64_b32 %2, %4 offset0:4 offset1:5\n ds_read2st64_b32 %3, %4 offset0:6 offset1:7\n")
                         "s_waitcnt lgkmcnt(0)\n"
                         : "=&v"(*(double*)&x0), "=&v"(*(double*)&x2), "=&v"(*(double*)&x4), "=&v"(*(double*)&x6) : "v"(laddr) : "memory");
        r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    } else if (role == DSREADB64X2) {
        typedef float v4 __attribute__((ext_vector_type(4)));
        v4 a, b, c, d;
        const unsigned la = (unsigned)(wave * 2048 + lane * 2) * 4u;
        for (int i = 0; i < iters; ++i)
            asm volatile(REP4("ds_read2_b64 %0, %4 offset1:1\n ds_read2_b64 %1, %4 offset0:2 offset1:3\n"
                              "ds_read2_b64 %2, %4 offset0:4 offset1:5\n ds_read2_b64 %3, %4 offset0:6 offset1:7\n")
                         "s_waitcnt lgkmcnt(0)\n"
                         : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(la) : "memory");
        r = a.x + b.y + c.z + d.w;
    } else if (role == DSWRITE) {
        float x0 = lane;
        for (int i = 0; i < iters; ++i)
            asm volatile(REP4("ds_write_b32 %1, %0\n ds_write_b32 %1, %0 offset:256\n ds_write_b32 %1, %0 offset:512\n ds_write_b32 %1, %0 offset:768\n")
                         "s_waitcnt lgkmcnt(0)\n"
                         : : "v"(x0), "v"(laddr) : "memory");
        r = x0;
    } else if (role == DSWRITE2ST) {
        float x0 = lane, x1 = 1.0f;
        for (int i = 0; i < iters; ++i)
            asm volatile(REP4("ds_write2st64_b32 %2, %0, %1 offset1:1\n ds_write2st64_b32 %2, %0, %1 offset0:2 offset1:3\n"
                              "ds_write2st64_b32 %2, %0, %1 offset0:4 offset1:5\n ds_write2st64_b32 %2, %0, %1 offset0:6 offset1:7\n")
                         "s_waitcnt lgkmcnt(0)\n"
                         : : "v"(x0), "v"(x1), "v"(laddr) : "memory");
        r = x0;
    } else if (role == SALU) {
        unsigned s0 = 1, s1 = 2, s2 = 3, s3 = 4;
        for (int i = 0; i < iters; ++i)
            asm volatile(REP4("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n")
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        r = (float)(s0 + s1 + s2 + s3);
    } else if (role == VMEM) {
        float x0 = 0, x1 = 0, x2 = 0, x3 = 0;
        const unsigned off = (unsigned)lane * 4u;
        for (int i = 0; i < iters; ++i) {       // 4 loads per iteration (an L2-resident 1 KB window per wave)
            x0 += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, (i & 15) * 256, 0));
            x1 += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, (i & 15) * 256 + 4096, 0));
            x2 += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, (i & 15) * 256 + 8192, 0));
            x3 += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, (i & 15) * 256 + 12288, 0));
        }
        r = x0 + x1 + x2 + x3;
    } else if (role == VMEMX4) {
        typedef float v4 __attribute__((ext_vector_type(4)));
        v4 a = {0, 0, 0, 0};
        const unsigned off = (unsigned)lane * 16u;
        for (int i = 0; i < iters; ++i) {
            a += __builtin_bit_cast(v4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, (i & 15) * 1024, 0));
            a += __builtin_bit_cast(v4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, (i & 15) * 1024 + 16384, 0));
            a += __builtin_bit_cast(v4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, (i & 15) * 1024 + 32768, 0));
            a += __builtin_bit_cast(v4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, (i & 15) * 1024 + 49152, 0));
        }
        r = a.x + a.y + a.z + a.w;
    }
    else if (role == GLOBAL_SADDR) {
        float x0, x1, x2, x3;
        const unsigned off = (unsigned)lane * 4u;
        for (int i = 0; i < iters; ++i)
            asm volatile("global_load_dword %0, %4, %5\n global_load_dword %1, %4, %5 offset:1024\n"
                         "global_load_dword %2, %4, %5 offset:2048\n global_load_dword %3, %4, %5 offset:3072\n s_waitcnt vmcnt(0)\n"
                         : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(off), "s"(src) : "memory");
        r = x0 + x1 + x2 + x3;
    } else if (role == GLOBAL_VADDR) {
        float x0, x1, x2, x3;
        const float* pl = src + lane;
        for (int i = 0; i < iters; ++i)
            asm volatile("global_load_dword %0, %4, off\n global_load_dword %1, %4, off offset:1024\n"
                         "global_load_dword %2, %4, off offset:2048\n global_load_dword %3, %4, off offset:3072\n s_waitcnt vmcnt(0)\n"
                         : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(pl) : "memory");
        r = x0 + x1 + x2 + x3;
    } else if (role == GLOBAL_X4_SADDR) {
        typedef float v4 __attribute__((ext_vector_type(4)));
        v4 a, b, c, d;
        const unsigned off = (unsigned)lane * 16u;
        for (int i = 0; i < iters; ++i)
            asm volatile("global_load_dwordx4 %0, %4, %5\n global_load_dwordx4 %1, %4, %5 offset:1024\n"
                         "global_load_dwordx4 %2, %4, %5 offset:2048\n global_load_dwordx4 %3, %4, %5 offset:3072\n s_waitcnt vmcnt(0)\n"
                         : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(off), "s"(src) : "memory");
        r = a.x + b.y + c.z + d.w;
    } else if (role == VMEM_NOUSE) {
        float x0, x1, x2, x3;
        const unsigned off = (unsigned)lane * 4u;
        for (int i = 0; i < iters; ++i)
            asm volatile("buffer_load_dword %0, %4, %5, 0 offen\n buffer_load_dword %1, %4, %5, 0 offen offset:1024\n"
                         "buffer_load_dword %2, %4, %5, 0 offen offset:2048\n buffer_load_dword %3, %4, %5, 0 offen offset:3072\n s_waitcnt vmcnt(0)\n"
                         : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(off), "s"(rs) : "memory");
        r = x0 + x1 + x2 + x3;
    } else if (role == VMEM_SOFF || role == VMEM_SOFF_L1) {
        float x0, x1, x2, x3;
        const unsigned off = (unsigned)lane * 4u;
        const int mask = role == VMEM_SOFF ? 15 : 0;
        for (int i = 0; i < iters; ++i) {
            const int so = __builtin_amdgcn_readfirstlane((i & mask) * 256);
            const int so1 = __builtin_amdgcn_readfirstlane(so + (mask ? 4096 : 1024)), so2 = __builtin_amdgcn_readfirstlane(so + (mask ? 8192 : 2048)),
                      so3 = __builtin_amdgcn_readfirstlane(so + (mask ? 12288 : 3072));
            asm volatile("buffer_load_dword %0, %4, %5, %6 offen\n buffer_load_dword %1, %4, %5, %7 offen\n"
                         "buffer_load_dword %2, %4, %5, %8 offen\n buffer_load_dword %3, %4, %5, %9 offen\n s_waitcnt vmcnt(0)\n"
                         : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3)
                         : "v"(off), "s"(rs), "s"(so), "s"(so1), "s"(so2), "s"(so3) : "memory");
        }
        r = x0 + x1 + x2 + x3;
    } else if (role == VMEM_ADD) {
        float x0, x1, x2, x3, a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        const unsigned off = (unsigned)lane * 4u;
        for (int i = 0; i < iters; ++i)
            asm volatile("buffer_load_dword %0, %8, %9, 0 offen\n buffer_load_dword %1, %8, %9, 0 offen offset:1024\n"
                         "buffer_load_dword %2, %8, %9, 0 offen offset:2048\n buffer_load_dword %3, %8, %9, 0 offen offset:3072\n s_waitcnt vmcnt(0)\n"
                         "v_add_f32 %4, %4, %0\n v_add_f32 %5, %5, %1\n v_add_f32 %6, %6, %2\n v_add_f32 %7, %7, %3\n"
                         : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(off), "s"(rs) : "memory");
        r = a0 + a1 + a2 + a3;
    } else if (role == VMEM_BUILTIN_NOADD) {
        const unsigned off = (unsigned)lane * 4u;
        for (int i = 0; i < iters; ++i) {
            float x0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, (i & 15) * 256, 0));
            float x1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, (i & 15) * 256 + 4096, 0));
            float x2 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, (i & 15) * 256 + 8192, 0));
            float x3 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, (i & 15) * 256 + 12288, 0));
            asm volatile("" : : "v"(x0), "v"(x1), "v"(x2), "v"(x3));
        }
    } else if (role == VMEMX2) {
        typedef float v2 __attribute__((ext_vector_type(2)));
        v2 a, b, c, d;
        const unsigned off = (unsigned)lane * 8u;
        for (int i = 0; i < iters; ++i)
            asm volatile("buffer_load_dwordx2 %0, %4, %5, 0 offen\n buffer_load_dwordx2 %1, %4, %5, 0 offen offset:1024\n"
                         "buffer_load_dwordx2 %2, %4, %5, 0 offen offset:2048\n buffer_load_dwordx2 %3, %4, %5, 0 offen offset:3072\n s_waitcnt vmcnt(0)\n"
                         : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(off), "s"(rs) : "memory");
        r = a.x + b.y + c.x + d.y;
    } else if (role == SLOAD) {
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        u4 a, b, c, d;
        for (int i = 0; i < iters; ++i)
            asm volatile("s_load_dwordx4 %0, %4, 0x0\n s_load_dwordx4 %1, %4, 0x10\n s_load_dwordx4 %2, %4, 0x20\n s_load_dwordx4 %3, %4, 0x30\n s_waitcnt lgkmcnt(0)\n"
                         : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d) : "s"(src) : "memory");
        r = (float)(a.x + b.y + c.z + d.w);
    } else if (role == DSREADB128) {
        typedef float v4 __attribute__((ext_vector_type(4)));
        v4 a, b, c, d;
        const unsigned la = (unsigned)(wave * 2048 + lane * 4) * 4u;
        for (int i = 0; i < iters; ++i)
            asm volatile(REP4("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n")
                         "s_waitcnt lgkmcnt(0)\n"
                         : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(la) : "memory");
        r = a.x + b.y + c.z + d.w;
    } else if (role == DSWRITEB128) {
        typedef float v4 __attribute__((ext_vector_type(4)));
        v4 a = {1, 2, 3, 4};
        const unsigned la = (unsigned)(wave * 2048 + lane * 4) * 4u;
        for (int i = 0; i < iters; ++i)
            asm volatile(REP4("ds_write_b128 %1, %0\n ds_write_b128 %1, %0 offset:1024\n ds_write_b128 %1, %0 offset:2048\n ds_write_b128 %1, %0 offset:3072\n")
                         "s_waitcnt lgkmcnt(0)\n"
                         : : "v"(a), "v"(la) : "memory");
        r = a.x;
    } else if (role == DSWRITEB64) {
        typedef float v2 __attribute__((ext_vector_type(2)));
        v2 a = {1, 2};
        const unsigned la = (unsigned)(wave * 2048 + lane * 2) * 4u;
        for (int i = 0; i < iters; ++i)
            asm volatile(REP4("ds_write_b64 %1, %0\n ds_write_b64 %1, %0 offset:512\n ds_write_b64 %1, %0 offset:1024\n ds_write_b64 %1, %0 offset:1536\n")
                         "s_waitcnt lgkmcnt(0)\n"
                         : : "v"(a), "v"(la) : "memory");
        r = a.x;
    } else if (role == VPKFMA) {
        typedef float v2 __attribute__((ext_vector_type(2)));
        v2 x0 = {1, 2}, x1 = {3, 4}, x2 = {5, 6}, x3 = {7, 8};
        for (int i = 0; i < iters; ++i)
            asm volatile(REP4("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        r = x0.x + x1.y + x2.x + x3.y;
    }
    __syncthreads();
    out[blockIdx.x * 512 + threadIdx.x] = r + lds[lane];
}

static float run(int a, int b, int iters, const float* src, float* d_out, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 0, 0, a, b, iters, src, d_out);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 0, 0, a, b, iters, src, d_out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 3 * 1e3f;
}

int main() {
    const int blocks = 256, iters = 4096;
    setvbuf(stdout, nullptr, _IONBF, 0);
    float *d_out, *src;
    hipMalloc(&d_out, blocks * 512 * sizeof(float));
    hipMalloc(&src, 1 << 20);
    hipMemset(src, 0, 1 << 20);
    const float mfma_us = run(MFMA, IDLE, iters, src, d_out, blocks);
    // 4096 iterations x 4 MFMAs x 64 cycles
    const double ghz = 4096.0 * 4 * 64 / (mfma_us * 1e3);
    printf("mfma alone: %.1f us  -> %.2f GHz if the pipe is saturated; mfma || mfma: %.1f us\n", mfma_us, ghz,
           run(MFMA, MFMA, iters, src, d_out, blocks));
    printf("%-22s %6s %10s %10s %12s %14s\n", "partner stream", "n/it", "alone us", "beside us", "added us", "cycles / instr");
    for (int role = VADD; role < NROLES; ++role) {
        const int n_per_iter = (role == VMEM || role == VMEMX4 || role == GLOBAL_SADDR || role == GLOBAL_VADDR || role == GLOBAL_X4_SADDR ||
                                role == VMEMX2 || role == LDSDMA || role == SLOAD || role >= VMEM_NOUSE) ? 4 : 16;
        if (role == LDSDMA) continue;      // (not assembled here; MI355X_MICROARCH.md puts one LDS-DMA piece at ~60 cycles among bare MFMAs)
        const float alone = run(role, IDLE, iters, src, d_out, blocks);
        const float both = run(MFMA, role, iters, src, d_out, blocks);
        const double added = both - mfma_us;
        printf("%-22s %6d %10.1f %10.1f %12.1f %14.2f   (alone: %.2f cycles / instr)\n", kNames[role], n_per_iter, alone, both, added,
               added * 1e3 * ghz / (4096.0 * n_per_iter), alone * 1e3 * ghz / (4096.0 * n_per_iter));
    }
    return 0;
}
