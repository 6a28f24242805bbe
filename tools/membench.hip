// Development tool: HBM read ceilings on one GPU for the access patterns the reduce kernel uses.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/membench tools/membench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// (a) one long stream, each thread U loads in flight, grid-stride
template <int U, bool NT>
__global__ __launch_bounds__(256) void read1(const f4* __restrict__ p, size_t n4, float* out) {
    f4 acc = {0, 0, 0, 0};
    size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        f4 t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) t[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += t[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

// (b) K rows, a block owns VPT*256 consecutive vectors of every row (the reduce kernel's pattern)
template <int U, int VPT>
__global__ __launch_bounds__(256) void readk(const f4* __restrict__ p, size_t row4, int K, unsigned n4, float* out) {
    f4 acc[VPT];
    for (int q = 0; q < VPT; ++q) acc[q] = f4{0, 0, 0, 0};
    unsigned tiles = n4 / (256 * VPT);
    for (unsigned t = blockIdx.x; t < tiles; t += gridDim.x) {
        unsigned base = t * 256 * VPT + threadIdx.x;
        for (int k = 0; k + U <= K; k += U) {
            f4 x[U][VPT];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int q = 0; q < VPT; ++q) x[u][q] = __builtin_nontemporal_load(p + (size_t)(k + u) * row4 + base + q * 256);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int q = 0; q < VPT; ++q) acc[q] += x[u][q];
        }
    }
    float s = 0;
    for (int q = 0; q < VPT; ++q) s += acc[q].x + acc[q].y + acc[q].z + acc[q].w;
    if (s == 123.456f) out[0] = s;
}

// (c) same pattern with the reduce arithmetic: s += (x - g) * w
template <int U, int VPT, int MODE = 0>
__global__ __launch_bounds__(256) void readk_math(const f4* __restrict__ p, size_t row4, int K, unsigned n4, const f4* __restrict__ gp, f4* __restrict__ gout, float w, unsigned wmask = 0xFFFF) {
    unsigned tiles = n4 / (256 * VPT);
    for (unsigned t = blockIdx.x; t < tiles; t += gridDim.x) {
        unsigned base = t * 256 * VPT + threadIdx.x;
        f4 acc[VPT], g[VPT];
        for (int q = 0; q < VPT; ++q) { acc[q] = f4{0, 0, 0, 0}; g[q] = (MODE >= 2) ? f4{w, w, w, w} : gp[base + q * 256]; }
        for (int k = 0; k + U <= K; k += U) {
            f4 x[U][VPT];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int q = 0; q < VPT; ++q) x[u][q] = __builtin_nontemporal_load(p + (size_t)(k + u) * row4 + base + q * 256);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int q = 0; q < VPT; ++q) acc[q] = acc[q] + (x[u][q] - g[q]) * w;
        }
        if (MODE == 0 || MODE == 3) { for (int q = 0; q < VPT; ++q) gout[base + q * 256] = g[q] + acc[q]; }
        else if (MODE == 4) { for (int q = 0; q < VPT; ++q) gout[(base + q * 256) & wmask] = g[q] + acc[q]; }
        else if (MODE == 5) { for (int q = 0; q < VPT; ++q) __builtin_nontemporal_store(g[q] + acc[q], gout + base + q * 256); }
        else { for (int q = 0; q < VPT; ++q) { f4 r = g[q] + acc[q]; if (r.x == 123.456f) gout[base + q * 256] = r; } }
    }
}

__global__ void fill_rand(unsigned* p, size_t n) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        unsigned long long z = i * 0x9E3779B97F4A7C15ull + 12345;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        float f = (float)((int)(z & 0xFFFF) - 32768) * 1e-5f;
        p[i] = __float_as_uint(f);
    }
}

__global__ __launch_bounds__(256) void copy4(const f4* __restrict__ a, f4* __restrict__ b, size_t n4) {
    size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) b[i] = __builtin_nontemporal_load(a + i);
}

template <typename F>
float timeit(F f, int reps = 10) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int r = 0; r < reps; ++r) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms); }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}
#include <algorithm>
int main() {
    const int K = 64; const size_t row = 25557056; const size_t n = row * K;   // 6.54 GB
    float* p; float* out; CK(hipMalloc(&p, n * 4)); CK(hipMalloc(&out, 4)); CK(hipMemset(p, 1, n * 4));
    float* q; CK(hipMalloc(&q, (size_t)1 << 31));
    size_t n4 = n / 4;
    float ms;
    for (int grid : {1024, 2048, 4096, 8192, 16384}) {
        ms = timeit([&] { read1<8, true><<<grid, 256>>>((const f4*)p, n4, out); });
        printf("read1 U=8 nt grid=%5d: %.3f ms %.0f GB/s\n", grid, ms, n * 4 / ms / 1e6);
    }
    ms = timeit([&] { read1<8, false><<<4096, 256>>>((const f4*)p, n4, out); });
    printf("read1 U=8 plain grid=4096: %.3f ms %.0f GB/s\n", ms, n * 4 / ms / 1e6);
    ms = timeit([&] { read1<16, true><<<2048, 256>>>((const f4*)p, n4, out); });
    printf("read1 U=16 nt grid=2048: %.3f ms %.0f GB/s\n", ms, n * 4 / ms / 1e6);
    ms = timeit([&] { read1<4, true><<<8192, 256>>>((const f4*)p, n4, out); });
    printf("read1 U=4 nt grid=8192: %.3f ms %.0f GB/s\n", ms, n * 4 / ms / 1e6);
    unsigned r4 = row / 4;
    ms = timeit([&] { readk<8, 4><<<r4 / 1024, 256>>>((const f4*)p, r4, K, r4, out); });
    printf("readk U=8 VPT=4 tile/block: %.3f ms %.0f GB/s\n", ms, n * 4 / ms / 1e6);
    ms = timeit([&] { readk<8, 4><<<512, 256>>>((const f4*)p, r4, K, r4, out); });
    printf("readk U=8 VPT=4 grid=512: %.3f ms %.0f GB/s\n", ms, n * 4 / ms / 1e6);
    ms = timeit([&] { readk<8, 1><<<r4 / 256, 256>>>((const f4*)p, r4, K, r4, out); });
    printf("readk U=8 VPT=1 tile/block: %.3f ms %.0f GB/s\n", ms, n * 4 / ms / 1e6);
    ms = timeit([&] { readk<8, 1><<<2048, 256>>>((const f4*)p, r4, K, r4, out); });
    printf("readk U=8 VPT=1 grid=2048: %.3f ms %.0f GB/s\n", ms, n * 4 / ms / 1e6);
    ms = timeit([&] { readk<16, 2><<<r4 / 512, 256>>>((const f4*)p, r4, K, r4, out); });
    printf("readk U=16 VPT=2 tile/block: %.3f ms %.0f GB/s\n", ms, n * 4 / ms / 1e6);
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) { fill_rand<<<4096, 256>>>((unsigned*)p, n); fill_rand<<<4096, 256>>>((unsigned*)q, ((size_t)1 << 31) / 4); CK(hipDeviceSynchronize()); printf("-- random data --\n"); }
        ms = timeit([&] { readk<8, 4><<<r4 / 1024, 256>>>((const f4*)p, r4, K, r4, out); });
        printf("readk U=8 VPT=4 tile/block: %.3f ms %.0f GB/s\n", ms, n * 4 / ms / 1e6);
        ms = timeit([&] { readk_math<8, 4><<<r4 / 1024, 256>>>((const f4*)p, r4, K, r4, (const f4*)q, (f4*)q + r4, 0.015625f); });
        printf("readk_math U=8 VPT=4 tile/block: %.3f ms %.0f GB/s client, %.0f alg\n", ms, n * 4 / ms / 1e6, (n + 2 * row) * 4 / ms / 1e6);
        ms = timeit([&] { readk_math<8, 4, 1><<<r4 / 1024, 256>>>((const f4*)p, r4, K, r4, (const f4*)q, (f4*)q + r4, 0.015625f); });
        printf("readk_math nostore: %.3f ms %.0f GB/s client\n", ms, n * 4 / ms / 1e6);
        ms = timeit([&] { readk_math<8, 4, 2><<<r4 / 1024, 256>>>((const f4*)p, r4, K, r4, (const f4*)q, (f4*)q + r4, 0.015625f); });
        printf("readk_math nostore no-g-load: %.3f ms %.0f GB/s client\n", ms, n * 4 / ms / 1e6);
        ms = timeit([&] { readk_math<8, 4, 3><<<r4 / 1024, 256>>>((const f4*)p, r4, K, r4, (const f4*)q, (f4*)q + r4, 0.015625f); });
        printf("readk_math store, no-g-load: %.3f ms %.0f GB/s client\n", ms, n * 4 / ms / 1e6);
        ms = timeit([&] { readk_math<8, 4, 4><<<r4 / 1024, 256>>>((const f4*)p, r4, K, r4, (const f4*)q, (f4*)q + r4, 0.015625f); });
        printf("readk_math store to 1MB window: %.3f ms %.0f GB/s client\n", ms, n * 4 / ms / 1e6);
        for (unsigned mb : {1u, 4u, 16u, 32u, 64u}) {
            unsigned mask = mb * 65536u - 1u;   // window of mb MiB in 16-byte vectors
            ms = timeit([&] { readk_math<8, 4, 4><<<r4 / 1024, 256>>>((const f4*)p, r4, K, r4, (const f4*)q, (f4*)q + r4, 0.015625f, mask); });
            printf("readk_math store to %u MiB window: %.3f ms %.0f GB/s client\n", mb, ms, n * 4 / ms / 1e6);
        }
        ms = timeit([&] { readk_math<8, 4, 5><<<r4 / 1024, 256>>>((const f4*)p, r4, K, r4, (const f4*)q, (f4*)q + r4, 0.015625f); });
        printf("readk_math nt store: %.3f ms %.0f GB/s client\n", ms, n * 4 / ms / 1e6);
        ms = timeit([&] { readk_math<8, 4, 0><<<512, 256>>>((const f4*)p, r4, K, r4, (const f4*)q, (f4*)q + r4, 0.015625f); });
        printf("readk_math grid=512 store: %.3f ms %.0f GB/s client\n", ms, n * 4 / ms / 1e6);
        ms = timeit([&] { readk_math<8, 4, 5><<<512, 256>>>((const f4*)p, r4, K, r4, (const f4*)q, (f4*)q + r4, 0.015625f); });
        printf("readk_math grid=512 nt store: %.3f ms %.0f GB/s client\n", ms, n * 4 / ms / 1e6);
        ms = timeit([&] { readk_math<8, 4, 1><<<512, 256>>>((const f4*)p, r4, K, r4, (const f4*)q, (f4*)q + r4, 0.015625f); });
        printf("readk_math grid=512 nostore: %.3f ms %.0f GB/s client\n", ms, n * 4 / ms / 1e6);
        ms = timeit([&] { readk_math<8, 2><<<r4 / 512, 256>>>((const f4*)p, r4, K, r4, (const f4*)q, (f4*)q + r4, 0.015625f); });
        printf("readk_math U=8 VPT=2 tile/block: %.3f ms %.0f GB/s client\n", ms, n * 4 / ms / 1e6);
        ms = timeit([&] { readk_math<8, 1><<<r4 / 256, 256>>>((const f4*)p, r4, K, r4, (const f4*)q, (f4*)q + r4, 0.015625f); });
        printf("readk_math U=8 VPT=1 tile/block: %.3f ms %.0f GB/s client\n", ms, n * 4 / ms / 1e6);
        ms = timeit([&] { read1<8, true><<<8192, 256>>>((const f4*)p, n4, out); });
        printf("read1 U=8 nt grid=8192: %.3f ms %.0f GB/s\n", ms, n * 4 / ms / 1e6);
    }
    size_t c4 = ((size_t)1 << 31) / 16;
    ms = timeit([&] { copy4<<<8192, 256>>>((const f4*)p, (f4*)q, c4); });
    printf("copy4 2GiB: %.3f ms %.0f GB/s (r+w)\n", ms, 2.0 * c4 * 16 / ms / 1e6);
    return 0;
}
