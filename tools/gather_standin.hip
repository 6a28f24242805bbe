// Development tool: a kernel that stands in for RCCL's all-gather on ONE GPU.  A few persistent
// blocks (as many as a collective has channels) write `bytes` into a buffer in 16-byte stores,
// optionally pausing between stores so that the kernel lasts as long as a link-bound gather
// would.  Built by tools/coresidency_probe.py into tools/lib/libgather_standin.so.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f4 __attribute__((ext_vector_type(4)));

// `regs` floats per lane are kept live so that the block holds registers the way a collective's
// kernel does (0: a lean block that fits beside anything).
template <int REGS>
__global__ __launch_bounds__(256) void standin_kernel(f4* dst, const f4* src, size_t nvec, int pause) {
    f4 keep[REGS > 0 ? REGS / 4 : 1];
#pragma unroll
    for (int i = 0; i < (REGS > 0 ? REGS / 4 : 1); ++i) keep[i] = f4{float(i), 1.0f, 2.0f, float(threadIdx.x)};
    const size_t stride = size_t(gridDim.x) * blockDim.x;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        f4 v = src ? src[i] : keep[0];
#pragma unroll
        for (int k = 0; k < (REGS > 0 ? REGS / 4 : 1); ++k) v += keep[k];
        dst[i] = v;
        for (int p = 0; p < pause; ++p) __builtin_amdgcn_s_sleep(16);
    }
}

extern "C" int standin_launch(void* dst, const void* src, size_t bytes, int blocks, int pause, int regs,
                              void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t nvec = bytes / 16;
    if (regs >= 256)
        hipLaunchKernelGGL(standin_kernel<256>, dim3(blocks), dim3(256), 0, st, static_cast<f4*>(dst),
                           static_cast<const f4*>(src), nvec, pause);
    else if (regs >= 96)
        hipLaunchKernelGGL(standin_kernel<96>, dim3(blocks), dim3(256), 0, st, static_cast<f4*>(dst),
                           static_cast<const f4*>(src), nvec, pause);
    else
        hipLaunchKernelGGL(standin_kernel<0>, dim3(blocks), dim3(256), 0, st, static_cast<f4*>(dst),
                           static_cast<const f4*>(src), nvec, pause);
    return static_cast<int>(hipGetLastError());
}
