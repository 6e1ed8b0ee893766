// fetch_calib.hip -- how does rocprofv3's FETCH_SIZE count streaming reads of 4, 8 and 16 bytes per lane on gfx950?
// (MI355X_MICROARCH.md calibrates the 16-byte case: FETCH_SIZE reports half of the bytes.)  Each kernel sums a 1 GiB
// buffer once, far past the 256 MiB Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o /tmp/fetch_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/fc -o fc -- /tmp/fetch_calib
#include <hip/hip_runtime.h>

#include <cstdio>

template <typename T>
__global__ void __launch_bounds__(256) sum_kernel(const T* __restrict__ x, size_t n, float* out) {
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const T v = x[i];
        const float* f = reinterpret_cast<const float*>(&v);
        for (unsigned k = 0; k < sizeof(T) / 4; ++k) acc += f[k];
    }
    if (acc == 123.456f) out[0] = acc;        // keep the loads
}

int main() {
    const size_t bytes = size_t(1) << 30;
    float* x = nullptr;
    float* out = nullptr;
    hipMalloc(&x, bytes);
    hipMalloc(&out, 4);
    hipMemset(x, 0, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(sum_kernel<float>, dim3(8192), dim3(256), 0, 0, x, bytes / 4, out);
        hipLaunchKernelGGL(sum_kernel<float2>, dim3(8192), dim3(256), 0, 0, reinterpret_cast<const float2*>(x), bytes / 8, out);
        hipLaunchKernelGGL(sum_kernel<float4>, dim3(8192), dim3(256), 0, 0, reinterpret_cast<const float4*>(x), bytes / 16, out);
    }
    hipDeviceSynchronize();
    std::printf("read %zu bytes per launch\n", bytes);
    return 0;
}
