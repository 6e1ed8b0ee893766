// repack.hip -- every derived form of the UNet's weights in three launches.
//
// After an optimizer step (train_diffusion.py:232, scaler.step(optimizer)) every re-layout of every weight is stale: the
// direct kernels' packing, the Winograd-domain filters in both layouts, the transposed tap-flipped filters of the
// backward-data convolutions and THEIR packings, the fused q/k/v matrix, the transposed time-embedding matrices.  As one
// launch per tensor and form that is ~700 launches of a few microseconds each (2.7 ms of a 29 ms step at batch 32, 64x64:
// profiles/r02/train_step_b32_64_kernel_stats.csv, conv_pack / winograd_pack* / transpose* / copyBuffer rows).  Here a job
// table built once per model names (kind, source, destination, shape) of every re-layout; a workgroup finds its job by
// binary search over the jobs' first-block indices and runs the same per-element functions as the stand-alone kernels
// (pack_device.h).  Three phases because of read-after-write: raw -> {packing, U, transposed raw, q/k/v matrix}, then
// the forms derived from those, then the forms derived from the second phase.
#include "common.h"
#include "pack_device.h"
#include "train.h"

namespace sisic {

constexpr int PB_THREADS = 256, PB_ITER = 4;      // elements per workgroup = 1024

__global__ void __launch_bounds__(PB_THREADS) pack_batch_kernel(const PackJob* __restrict__ jobs, int njobs) {
    // last job whose first_block <= blockIdx.x
    int lo = 0, hi = njobs - 1;
    const int blk = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= blk) lo = mid; else hi = mid - 1;
    }
    const PackJob j = jobs[lo];
    const size_t base = (size_t)(blk - j.first_block) * (PB_THREADS * PB_ITER) + threadIdx.x;
#pragma unroll
    for (int it = 0; it < PB_ITER; ++it) {
        const size_t i = base + (size_t)it * PB_THREADS;
        if (i >= j.total) break;
        switch (j.kind) {
            case PackJob::COPY: j.dst[i] = j.src[i]; break;
            case PackJob::TRANSPOSE2D: transpose2d_elem(i, j.src, j.b, j.dst, j.c, j.d); break;
            case PackJob::FLIP: transpose_flip_elem(i, j.src, j.a, j.b, j.c, j.dst); break;
            case PackJob::CONV_PACK: conv_pack_elem(i, j.src, j.a, j.b, j.c, j.d, j.e, j.dst); break;
            case PackJob::WINO_FIRST: winograd_pack_elem(i, j.src, j.a, j.b, j.d, j.e, j.dst); break;
            case PackJob::WINO_WIDE: winograd_pack_wide_elem(i, j.src, j.a, j.b, j.dst); break;
            case PackJob::WINO_BF3: winograd_pack_bf3_elem(i, j.src, j.a, j.b, reinterpret_cast<unsigned*>(j.dst)); break;
        }
    }
}

int pack_job_blocks(const PackJob& j) { return (int)((j.total + PB_THREADS * PB_ITER - 1) / (PB_THREADS * PB_ITER)); }

int launch_pack_batch(sisic_ctx*, const PackJob* dev_jobs, int njobs, int nblocks, hipStream_t s) {
    if (njobs == 0) return SISIC_OK;
    hipLaunchKernelGGL(pack_batch_kernel, dim3(nblocks), dim3(PB_THREADS), 0, s, dev_jobs, njobs);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// ---- job constructors: the arithmetic of the stand-alone launchers (launch_conv_pack, launch_winograd_pack, ...)
PackJob pack_job_copy(const float* src, float* dst, size_t n) {
    PackJob j{}; j.kind = PackJob::COPY; j.src = src; j.dst = dst; j.total = n; return j;
}
PackJob pack_job_transpose2d(const float* in, int rows, int cols, float* out, int out_ld, int out_col0) {
    PackJob j{}; j.kind = PackJob::TRANSPOSE2D; j.src = in; j.dst = out; j.a = rows; j.b = cols; j.c = out_ld; j.d = out_col0;
    j.total = (size_t)rows * cols; return j;
}
PackJob pack_job_flip(const float* w, int Cout, int Cin, int KK, float* wt) {
    PackJob j{}; j.kind = PackJob::FLIP; j.src = w; j.dst = wt; j.a = Cout; j.b = Cin; j.c = KK; j.total = (size_t)Cout * Cin * KK; return j;
}
PackJob pack_job_conv(const float* w, int Cout, int Cin, int k, float* packed) {
    PackJob j{}; j.kind = PackJob::CONV_PACK; j.src = w; j.dst = packed; j.a = Cout; j.b = Cin; j.c = k * k;
    j.d = conv_cin_pad(Cin, k); j.e = conv_cout_pad(Cout); j.total = (size_t)conv_packed_floats(Cout, Cin, k);      // 1x1: all three layouts
    return j;
}
PackJob pack_job_wino_first(const float* w, int Cout, int Cin, float* packed) {
    PackJob j{}; j.kind = PackJob::WINO_FIRST; j.src = w; j.dst = packed; j.a = Cout; j.b = Cin;
    j.d = round_up(Cin, 16); j.e = conv_cout_pad(Cout); j.total = (size_t)j.d * j.e; return j;
}
PackJob pack_job_wino_wide(int Cout, int Cin, float* packed) {          // reads the first layout at `packed`, writes behind it
    PackJob j{}; j.kind = PackJob::WINO_WIDE; j.src = packed; j.dst = packed + winograd_first_floats(Cout, Cin);
    j.a = conv_cout_pad(Cout); j.b = round_up(Cout, 128); j.total = (size_t)round_up(Cin, 16) * 16 * j.b; return j;
}

PackJob pack_job_wino_bf3(int Cout, int Cin, float* packed) {           // reads the first layout at `packed`, writes behind the wide one
    PackJob j{}; j.kind = PackJob::WINO_BF3; j.src = packed;
    const size_t wide = (size_t)round_up(Cin, 16) * 16 * round_up(Cout, 128);
    j.dst = packed + winograd_first_floats(Cout, Cin) + wide;
    j.a = conv_cout_pad(Cout); j.b = round_up(Cout, 128); j.total = wide + wide / 2; return j;
}

}  // namespace sisic
