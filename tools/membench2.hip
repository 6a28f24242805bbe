// Development tool: how cache-policy bits and store timing change the cost of the g write-back
// that rides on a K-row streaming read.  build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ f4 as_f4(u4 v) { return __builtin_bit_cast(f4, v); }
__device__ __forceinline__ u4 as_u4(f4 v) { return __builtin_bit_cast(u4, v); }

// LAUX: aux bits of the client-row loads, SAUX: aux bits of the result store (1 sc0, 2 nt, 16 sc1)
// EVERY: only tiles with t % EVERY == 0 store (write volume / EVERY)
template <int LAUX, int SAUX, int EVERY>
__global__ __launch_bounds__(256) void rk(const float* __restrict__ p, size_t row, int K, unsigned n4, const float* __restrict__ gp, float* __restrict__ gout, float w) {
    constexpr int U = 8, VPT = 4;
    unsigned tiles = n4 / (256 * VPT);
    __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)gp, 0, (int)(n4 * 16), 0x00020000);
    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)gout, 0, (int)(n4 * 16), 0x00020000);
    for (unsigned t = blockIdx.x; t < tiles; t += gridDim.x) {
        unsigned base = (t * 256 * VPT + threadIdx.x) * 16;
        f4 acc[VPT], g[VPT];
        for (int q = 0; q < VPT; ++q) { acc[q] = f4{0, 0, 0, 0}; g[q] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rg, base + q * 4096, 0, 0)); }
        for (int k = 0; k + U <= K; k += U) {
            f4 x[U][VPT];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p + (size_t)(k + u) * row), 0, (int)(n4 * 16), 0x00020000);
#pragma unroll
                for (int q = 0; q < VPT; ++q) x[u][q] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rx, base + q * 4096, 0, LAUX));
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int q = 0; q < VPT; ++q) acc[q] = acc[q] + (x[u][q] - g[q]) * w;
        }
        if (t % EVERY == 0) {
            for (int q = 0; q < VPT; ++q) __builtin_amdgcn_raw_buffer_store_b128(as_u4(g[q] + acc[q]), ro, base + q * 4096, 0, SAUX);
        } else {
            for (int q = 0; q < VPT; ++q) { f4 r = g[q] + acc[q]; if (r.x == 123.456f) __builtin_amdgcn_raw_buffer_store_b128(as_u4(r), ro, base + q * 4096, 0, 0); }
        }
    }
}

// Results parked in LDS and written in bursts of STASH tiles (one 256-thread block per CU, persistent).
template <int STASH, int SAUX = 0, bool BARRIER = false, int TH = 256, int U = 8, int VPT = 4>
__global__ __launch_bounds__(TH) void rk_stash(const float* __restrict__ p, size_t row, int K, unsigned n4, const float* __restrict__ gp, float* __restrict__ gout, float w, unsigned* counter = nullptr) {
    __shared__ f4 stash[STASH][VPT][TH];
    unsigned tiles = n4 / (TH * VPT);
    unsigned per = (tiles + gridDim.x - 1) / gridDim.x;        // contiguous tile range per block
    unsigned t0 = blockIdx.x * per, t1 = min(tiles, t0 + per);
    __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)gp, 0, (int)(n4 * 16), 0x00020000);
    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)gout, 0, (int)(n4 * 16), 0x00020000);
    unsigned phase = 0;
    for (unsigned tb = t0; tb < t0 + per; tb += STASH) {
        unsigned cnt = tb < t1 ? min((unsigned)STASH, t1 - tb) : 0;
        for (unsigned j = 0; j < cnt; ++j) {
            unsigned base = ((tb + j) * TH * VPT + threadIdx.x) * 16;
            f4 acc[VPT], g[VPT];
            for (int q = 0; q < VPT; ++q) { acc[q] = f4{0, 0, 0, 0}; g[q] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rg, base + q * (TH * 16), 0, 0)); }
            for (int k = 0; k + U <= K; k += U) {
                f4 x[U][VPT];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p + (size_t)(k + u) * row), 0, (int)(n4 * 16), 0x00020000);
#pragma unroll
                    for (int q = 0; q < VPT; ++q) x[u][q] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rx, base + q * (TH * 16), 0, 2));
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int q = 0; q < VPT; ++q) acc[q] = acc[q] + (x[u][q] - g[q]) * w;
            }
            for (int q = 0; q < VPT; ++q) stash[j][q][threadIdx.x] = g[q] + acc[q];
        }
        if (BARRIER) {
            ++phase;
            __syncthreads();
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned spins = 0;
                while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase * gridDim.x && ++spins < 4000000u) __builtin_amdgcn_s_sleep(8);
            }
            __syncthreads();
        }
        for (unsigned j = 0; j < cnt; ++j) {
            unsigned base = ((tb + j) * TH * VPT + threadIdx.x) * 16;
            for (int q = 0; q < VPT; ++q) __builtin_amdgcn_raw_buffer_store_b128(as_u4(stash[j][q][threadIdx.x]), ro, base + q * (TH * 16), 0, SAUX);
        }
    }
}

// Stash kernel with a software-pipelined client loop: two register sets of U clients, the next
// set's loads are issued before the current set is consumed.
template <int STASH, int SAUX, int U, int VPT>
__global__ __launch_bounds__(256) void rk_pipe(const float* __restrict__ p, size_t row, int K, unsigned n4, const float* __restrict__ gp, float* __restrict__ gout, float w) {
    constexpr int TH = 256;
    __shared__ f4 stash[STASH][VPT][TH];
    unsigned tiles = n4 / (TH * VPT);
    unsigned per = (tiles + gridDim.x - 1) / gridDim.x;
    unsigned t0 = blockIdx.x * per, t1 = min(tiles, t0 + per);
    __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)gp, 0, (int)(n4 * 16), 0x00020000);
    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)gout, 0, (int)(n4 * 16), 0x00020000);
    for (unsigned tb = t0; tb < t1; tb += STASH) {
        unsigned cnt = min((unsigned)STASH, t1 - tb);
        for (unsigned j = 0; j < cnt; ++j) {
            unsigned base = ((tb + j) * TH * VPT + threadIdx.x) * 16;
            f4 acc[VPT], g[VPT];
            for (int q = 0; q < VPT; ++q) { acc[q] = f4{0, 0, 0, 0}; g[q] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rg, base + q * (TH * 16), 0, 0)); }
            f4 xa[U][VPT], xb[U][VPT];
            auto issue = [&](f4 (&x)[U][VPT], int k) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p + (size_t)(k + u) * row), 0, (int)(n4 * 16), 0x00020000);
#pragma unroll
                    for (int q = 0; q < VPT; ++q) x[u][q] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rx, base + q * (TH * 16), 0, 2));
                }
            };
            auto consume = [&](f4 (&x)[U][VPT]) {
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int q = 0; q < VPT; ++q) acc[q] = acc[q] + (x[u][q] - g[q]) * w;
            };
            issue(xa, 0);
            int k = 0;
            for (; k + 2 * U < K; k += 2 * U) {   // assumes K % (2U) == 0
                issue(xb, k + U);
                consume(xa);
                issue(xa, k + 2 * U);
                consume(xb);
            }
            issue(xb, k + U);
            consume(xa);
            consume(xb);
            for (int q = 0; q < VPT; ++q) stash[j][q][threadIdx.x] = g[q] + acc[q];
        }
        for (unsigned j = 0; j < cnt; ++j) {
            unsigned base = ((tb + j) * TH * VPT + threadIdx.x) * 16;
            for (int q = 0; q < VPT; ++q) __builtin_amdgcn_raw_buffer_store_b128(as_u4(stash[j][q][threadIdx.x]), ro, base + q * (TH * 16), 0, SAUX);
        }
    }
}

// Adam-shaped traffic: per tile read g, m, v and write g, m, v (cheap stand-in arithmetic).
// LAYOUT 0: three separate arrays; 1: one array interleaved per tile [tile][3][VPT*256 vectors].
template <int STASH, int LAYOUT>
__global__ __launch_bounds__(256) void rk_adam(const float* __restrict__ p, size_t row, int K, unsigned n4, float* __restrict__ st, float w) {
    constexpr int U = 8, VPT = 4, TH = 256, NV = 3;
    __shared__ f4 stash[STASH][NV][VPT][TH];
    unsigned tiles = n4 / (TH * VPT);
    unsigned per = (tiles + gridDim.x - 1) / gridDim.x;
    unsigned t0 = blockIdx.x * per, t1 = min(tiles, t0 + per);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)st, 0, (int)(n4 * 16 * 3), 0x00020000);
    auto off = [&](unsigned tile, int c, int q) -> unsigned {
        if (LAYOUT == 0) return (unsigned)(((size_t)c * n4 + (size_t)tile * TH * VPT + q * TH + threadIdx.x) * 16);
        return (unsigned)((((size_t)tile * NV + c) * TH * VPT + q * TH + threadIdx.x) * 16);
    };
    for (unsigned tb = t0; tb < t1; tb += STASH) {
        unsigned cnt = min((unsigned)STASH, t1 - tb);
        for (unsigned j = 0; j < cnt; ++j) {
            unsigned base = ((tb + j) * TH * VPT + threadIdx.x) * 16;
            f4 acc[VPT], g[VPT], m[VPT], v[VPT];
            for (int q = 0; q < VPT; ++q) {
                acc[q] = f4{0, 0, 0, 0};
                g[q] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rs, off(tb + j, 0, q), 0, 0));
                m[q] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rs, off(tb + j, 1, q), 0, 0));
                v[q] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rs, off(tb + j, 2, q), 0, 0));
            }
            for (int k = 0; k + U <= K; k += U) {
                f4 x[U][VPT];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p + (size_t)(k + u) * row), 0, (int)(n4 * 16), 0x00020000);
#pragma unroll
                    for (int q = 0; q < VPT; ++q) x[u][q] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rx, base + q * (TH * 16), 0, 2));
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int q = 0; q < VPT; ++q) acc[q] = acc[q] + (x[u][q] - g[q]) * w;
            }
            for (int q = 0; q < VPT; ++q) {
                m[q] = m[q] * 0.9f + acc[q] * 0.1f;
                v[q] = v[q] * 0.99f + acc[q] * acc[q] * 0.01f;
                stash[j][0][q][threadIdx.x] = g[q] + m[q] * 0.1f;
                stash[j][1][q][threadIdx.x] = m[q];
                stash[j][2][q][threadIdx.x] = v[q];
            }
        }
        for (unsigned j = 0; j < cnt; ++j)
            for (int c = 0; c < NV; ++c)
                for (int q = 0; q < VPT; ++q) __builtin_amdgcn_raw_buffer_store_b128(as_u4(stash[j][c][q][threadIdx.x]), rs, off(tb + j, c, q), 0, 17);
    }
}

__global__ void fill_rand(unsigned* p, size_t n) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        unsigned long long z = i * 0x9E3779B97F4A7C15ull + 12345;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        p[i] = __float_as_uint((float)((int)(z & 0xFFFF) - 32768) * 1e-5f);
    }
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

const int K = 64; const size_t row = 25557056; const size_t n = row * K;
float *p, *q;
template <int LAUX, int SAUX, int EVERY>
void run(const char* name, int grid = 0) {
    unsigned r4 = row / 4;
    if (!grid) grid = r4 / 1024;
    float ms = timeit([&] { rk<LAUX, SAUX, EVERY><<<grid, 256>>>(p, row, K, r4, q, q + row, 0.015625f); });
    CK(hipGetLastError());
    printf("%-44s %.3f ms %6.0f GB/s client\n", name, ms, n * 4 / ms / 1e6);
}
unsigned* counter;
template <int STASH, int SAUX = 0, bool BARRIER = false, int TH = 256, int U = 8, int VPT = 4>
void run_stash(const char* name, int grid) {
    unsigned r4 = row / 4;
    float ms = timeit([&] { CK(hipMemsetAsync(counter, 0, 4)); rk_stash<STASH, SAUX, BARRIER, TH, U, VPT><<<grid, TH>>>(p, row, K, r4, q, q + row, 0.015625f, counter); });
    CK(hipGetLastError());
    printf("%-44s %.3f ms %6.0f GB/s client\n", name, ms, n * 4 / ms / 1e6);
}

template <int STASH, int SAUX, int U, int VPT>
void run_pipe(const char* name, int grid) {
    unsigned r4 = row / 4;
    float ms = timeit([&] { rk_pipe<STASH, SAUX, U, VPT><<<grid, 256>>>(p, row, K, r4, q, q + row, 0.015625f); });
    CK(hipGetLastError());
    printf("%-44s %.3f ms %6.0f GB/s client\n", name, ms, n * 4 / ms / 1e6);
}

float* st3;
template <int STASH, int LAYOUT>
void run_adam(const char* name, int grid) {
    unsigned r4 = row / 4;
    float ms = timeit([&] { rk_adam<STASH, LAYOUT><<<grid, 256>>>(p, row, K, r4, st3, 0.015625f); });
    CK(hipGetLastError());
    printf("%-44s %.3f ms %6.0f GB/s client, alg %6.0f\n", name, ms, n * 4 / ms / 1e6, (n + 6 * row) * 4 / ms / 1e6);
}

int main() {
    CK(hipMalloc(&p, n * 4)); CK(hipMalloc(&q, row * 8)); CK(hipMalloc(&counter, 256));
    fill_rand<<<4096, 256>>>((unsigned*)p, n); fill_rand<<<4096, 256>>>((unsigned*)q, row * 2); CK(hipDeviceSynchronize());
    CK(hipMalloc(&st3, row * 4 * 3)); CK(hipMemset(st3, 0, row * 4 * 3));
    run_adam<3, 0>("adam traffic, separate g/m/v, stash 3", 256);
    run_adam<3, 1>("adam traffic, interleaved per tile, stash 3", 256);
    run_adam<1, 0>("adam traffic, separate, stash 1, grid 512", 512);
    run_adam<1, 1>("adam traffic, interleaved, stash 1, grid 512", 512);
    run_adam<3, 0>("adam traffic, separate g/m/v, stash 3 (again)", 256);
    run_adam<3, 1>("adam traffic, interleaved, stash 3 (again)", 256);
    run<2, 0, 1>("load nt, store plain");
    run<2, 0, 1000000>("load nt, no store");
    run<2, 0, 2>("load nt, store plain, half volume");
    run<2, 0, 4>("load nt, store plain, quarter volume");
    run<2, 2, 1>("load nt, store nt");
    run<2, 1, 1>("load nt, store sc0");
    run<2, 16, 1>("load nt, store sc1");
    run<2, 17, 1>("load nt, store sc0 sc1");
    run<2, 19, 1>("load nt, store sc0 sc1 nt");
    run<2, 18, 1>("load nt, store sc1 nt");
    run<0, 0, 1>("load plain, store plain");
    run<0, 2, 1>("load plain, store nt");
    run<16, 0, 1>("load sc1, store plain");
    run<17, 0, 1>("load sc0 sc1, store plain");
    run<18, 0, 1>("load sc1 nt, store plain");
    run<19, 0, 1>("load sc0 sc1 nt, store plain");
    run<19, 19, 1>("load sc0 sc1 nt, store sc0 sc1 nt");
    run<3, 0, 1>("load sc0 nt, store plain");
    run<2, 0, 1>("load nt, store plain, grid 256", 256);
    run<2, 0, 1>("load nt, store plain, grid 512", 512);
    run_stash<4>("stash 4 tiles (64KB LDS), grid 512", 512);
    run_stash<8>("stash 8 tiles (128KB LDS), grid 256", 256);
    run_stash<2>("stash 2 tiles (32KB LDS), grid 512", 512);
    run_stash<8, 17>("stash 8 tiles, sc0 sc1 stores, grid 256", 256);
    run_stash<8, 0, true>("stash 8 tiles, chip barrier before writes, grid 256", 256);
    run_stash<8, 17, true>("stash 8, barrier, sc0 sc1, grid 256", 256);
    run_stash<4, 0, true>("stash 4 tiles, barrier, grid 256", 256);
    run_stash<9>("stash 9 tiles (144KB), grid 256", 256);
    run_stash<9, 17>("stash 9 tiles (144KB), sc0sc1, grid 256", 256);
    run_stash<4, 17, false, 512>("512 thr, stash 4 x 32KB, sc0sc1, grid 256", 256);
    run_stash<4, 17, false, 512, 4>("512 thr, U=4, stash 4 x 32KB, sc0sc1", 256);
    run_stash<8, 17, false, 512, 8, 2>("512 thr, vpt 2, stash 8 x 16KB, sc0sc1", 256);
    run_stash<8, 17, false, 512, 4, 2>("512 thr, vpt 2, U=4, stash 8 x 16KB, sc0sc1", 256);
    run_stash<8, 17, false, 1024, 8, 1>("1024 thr, vpt 1, stash 8 x 16KB, sc0sc1", 256);
    run_stash<4, 17, false, 1024, 4, 2>("1024 thr, vpt 2 U=4, stash 4 x 32KB, sc0sc1", 256);
    run_stash<8, 17, false, 256, 16, 2>("256 thr, vpt 2 U=16, stash 8 x 8KB, sc0sc1", 256);
    run_stash<4, 17>("stash 4 (64KB), sc0sc1, grid 512", 512);
    run_stash<2, 17>("stash 2 (32KB), sc0sc1, grid 1024", 1024);
    run_stash<1, 17>("stash 1 (16KB), sc0sc1, grid 1024", 1024);
    run_stash<1, 17>("stash 1 (16KB), sc0sc1, grid 2048", 2048);
    run_stash<4, 17, false, 256, 4, 4>("stash 4, U=4, sc0sc1, grid 512", 512);
    run_stash<2, 17, false, 256, 4, 4>("stash 2, U=4, sc0sc1, grid 1024", 1024);
    run_pipe<8, 17, 4, 4>("pipelined 2x(U=4) vpt 4, stash 8, sc0sc1", 256);
    run_pipe<8, 17, 8, 2>("pipelined 2x(U=8) vpt 2, stash 8(x8KB), sc0sc1", 256);
    run_pipe<8, 17, 2, 4>("pipelined 2x(U=2) vpt 4, stash 8, sc0sc1", 256);
    run_pipe<4, 17, 4, 4>("pipelined 2x(U=4) vpt 4, stash 4, grid 512", 512);
    return 0;
}
