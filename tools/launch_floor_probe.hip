// What does a kernel launch cost inside a replayed hipGraph on this chip, as a function of the launch's shape?  A chain of N
// dependent kernels that do nothing (or touch one cache line per workgroup), captured once and replayed; time per node.
//   hipcc --offload-arch=gfx950 -O2 tools/launch_floor_probe.hip -o tools/bin/launch_floor_probe && tools/bin/launch_floor_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void empty_kernel(float* p, int touch) {
    extern __shared__ float lds[];
    if (touch && threadIdx.x == 0) {
        lds[0] = p[blockIdx.x * 16];
        p[blockIdx.x * 16] = lds[0] + 1.0f;
    }
}

int main() {
    float* buf;
    CK(hipMalloc(&buf, 1 << 22));
    CK(hipMemset(buf, 0, 1 << 22));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    struct Shape { int grid, block, lds, touch; const char* name; };
    const Shape shapes[] = {
        {256, 1024, 158 * 1024, 0, "256 x 1024 threads, 158 KB LDS (the Winograd kernels' shape), nothing"},
        {256, 1024, 158 * 1024, 1, "256 x 1024 threads, 158 KB LDS, one line read + written per workgroup"},
        {1024, 1024, 158 * 1024, 0, "1024 x 1024 threads, 158 KB LDS (four rounds of workgroups), nothing"},
        {2048, 256, 0, 0, "2048 x 256 threads, no LDS, nothing"},
        {2048, 256, 0, 1, "2048 x 256 threads, no LDS, one line per workgroup"},
        {512, 256, 0, 1, "512 x 256 threads (gn_finalize's shape), one line per workgroup"},
    };
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(empty_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int N = 120, REPS = 50;
    for (const Shape& sh : shapes) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(sh.grid), dim3(sh.block), sh.lds, s, buf, sh.touch);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < REPS; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms = 0.0f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-90s %6.2f us per node (graph of %d, %d replays)\n", sh.name, ms * 1e3 / (REPS * N), N, REPS);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
