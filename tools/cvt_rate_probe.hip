// Development tool: issue rate of the 16-bit conversions next to a plain fp32 add on gfx950
// (the 16-bit aggregation kernels spend three roundings per client element).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/lib/cvt_rate_probe.bin tools/cvt_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void spin(float* out, int iters, float seed) {
    float r[8];
    for (int j = 0; j < 8; ++j) r[j] = seed + threadIdx.x * 0.001f + j;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 0) {
                __asm__ volatile("v_add_f32 %0, %0, %1" : "+v"(r[j]) : "v"(seed));
            } else if (MODE == 1) {
                __asm__ volatile("v_cvt_pk_bf16_f32 %0, 0, %0" : "+v"(r[j]));
            } else if (MODE == 2) {
                __asm__ volatile("v_cvt_f16_f32 %0, %0" : "+v"(r[j]));
            } else if (MODE == 3) {
                __asm__ volatile("v_cvt_f32_f16 %0, %0" : "+v"(r[j]));
            } else if (MODE == 4) {
                __asm__ volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&r[j & ~1])) : "v"(*reinterpret_cast<double*>(&r[j & ~1])));
            } else {
                __asm__ volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(r[j]));
            }
        }
    }
    float s = 0;
    for (int j = 0; j < 8; ++j) s += r[j];
    if (s == 12345.678f) out[0] = s;
}

template <int MODE>
void run(const char* name, float* out) {
    const int iters = 4096, blocks = 256 * 16;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    spin<MODE><<<blocks, 256>>>(out, iters, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    spin<MODE><<<blocks, 256>>>(out, iters, 1.0f);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double lane_ops = double(blocks) * 256 * iters * 8;
    printf("%-22s %8.3f ms  %7.2f T lane-instructions/s\n", name, ms, lane_ops / ms / 1e9);
}

int main() {
    float* out; CK(hipMalloc(&out, 4));
    run<0>("v_add_f32", out);
    run<1>("v_cvt_pk_bf16_f32", out);
    run<2>("v_cvt_f16_f32", out);
    run<3>("v_cvt_f32_f16", out);
    run<4>("v_pk_add_f32", out);
    run<5>("v_and_b32", out);
    return 0;
}
