// Development tool: can the result writes of a reduce round be absorbed on chip beyond LDS?
// The K-row reduce (non-temporal 16-byte loads, 4 vectors x 8 clients in flight per lane) stores NV
// result vectors per element -- either where they belong (a stream as long as the rows) or into a
// RING WINDOW of W bytes per result vector that is overwritten cyclically.  If windows larger than
// the L2 (32 MB) but within the Infinity Cache (256 MB) cost as little as an L2-resident one, a
// round could park its results there and write them out in one pure-write pass at the end.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/lib/membench4.bin tools/membench4.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// AUX: cache-policy bits of the store (17 = sc0 sc1 write-through, 0 = plain, 2 = nt)
template <int NV, int AUX>
__global__ __launch_bounds__(256) void rk4(const f4* __restrict__ p, size_t n4, int K, f4* __restrict__ out,
                                           size_t win4 /* vectors per result window */, float w) {
    constexpr int VPT = 4, U = 8;
    constexpr unsigned T = 256 * VPT;
    const unsigned tiles = static_cast<unsigned>(n4 / T);
    for (unsigned t = blockIdx.x; t < tiles; t += gridDim.x) {
        size_t i[VPT];
        for (int q = 0; q < VPT; ++q) i[q] = static_cast<size_t>(t) * T + q * 256 + threadIdx.x;
        f4 acc[VPT];
        for (int q = 0; q < VPT; ++q) acc[q] = f4{0, 0, 0, 0};
        for (int k = 0; k + U <= K; k += U) {
            f4 x[U][VPT];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int q = 0; q < VPT; ++q) x[u][q] = __builtin_nontemporal_load(p + i[q] + (k + u) * n4);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int q = 0; q < VPT; ++q) acc[q] = acc[q] + (x[u][q] - w) * w;
        }
        if (NV == 0) {
            for (int q = 0; q < VPT; ++q) if (acc[q].x == 123.456f) out[i[q]] = acc[q];
        } else {
            // tile t lands in window slot (t mod slots): whole tiles, so the stores stay contiguous
            const size_t slots = win4 / T;
            const size_t base = (static_cast<size_t>(t) % slots) * T;
#pragma unroll
            for (int c = 0; c < NV; ++c) {
                __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(out + c * win4 + base), 0, T * 16, 0x00020000);
#pragma unroll
                for (int q = 0; q < VPT; ++q) {
                    f4 r = acc[q] + float(c);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, r), ro, (q * 256 + threadIdx.x) * 16, 0, AUX);
                }
            }
        }
    }
}

__global__ void fill_rand(unsigned* p, size_t n) {
    size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    size_t s = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (; i < n; i += s) {
        unsigned z = static_cast<unsigned>(i) * 2654435761u + 12345u;
        z ^= z >> 15; z *= 2246822519u; z ^= z >> 13;
        p[i] = 0x3C000000u | (z & 0x007FFFFFu);
    }
}

template <typename F>
float median_ms(F launch, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

template <int NV, int AUX>
float run(const f4* slab, size_t n4, int K, f4* out, size_t win4, int reps) {
    const unsigned grid = static_cast<unsigned>(std::min<size_t>(n4 / 1024, 1u << 20));
    return median_ms([&] { rk4<NV, AUX><<<grid, 256>>>(slab, n4, K, out, win4, 0.01f); }, reps);
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 9;
    struct Case { size_t n; int K; };
    const Case cases[] = {{25557032, 32}, {25557032, 64}, {25557032, 16}, {86567656, 32}};
    for (const Case& c : cases) {
        const size_t n4 = c.n / 4 / 1024 * 1024;
        f4 *slab, *out;
        CK(hipMalloc(&slab, n4 * 16 * c.K)); CK(hipMalloc(&out, n4 * 16 * 3));
        fill_rand<<<4096, 256>>>(reinterpret_cast<unsigned*>(slab), n4 * 4 * c.K);
        CK(hipDeviceSynchronize());
        const float r0 = run<0, 0>(slab, n4, c.K, out, n4, reps);
        printf("== n=%zu floats (%.1f MB rows) K=%d: read-only %.4f ms (%.0f GB/s)\n", n4 * 4, n4 * 16 / 1e6, c.K, r0,
               double(c.K) * n4 * 16 / r0 / 1e6);
        for (size_t win_mb : {100000, 80, 42, 21, 10, 5, 2}) {
            const size_t win4 = std::min(n4, win_mb * 1000000 / 16 / 1024 * 1024);
            const float w1 = run<1, 17>(slab, n4, c.K, out, win4, reps), w3 = run<3, 17>(slab, n4, c.K, out, win4, reps);
            const float p3 = run<3, 0>(slab, n4, c.K, out, win4, reps), t3 = run<3, 2>(slab, n4, c.K, out, win4, reps);
            printf("  window %7.1f MB per result vector: 1 vector write-through +%.4f ms | 3 vectors write-through +%.4f ms, plain +%.4f ms, nt +%.4f ms"
                   "  (3 x %.0f MB stored)\n", win4 * 16 / 1e6, w1 - r0, w3 - r0, p3 - r0, t3 - r0, n4 * 16 / 1e6);
            fflush(stdout);
        }
        CK(hipFree(slab)); CK(hipFree(out));
    }
    return 0;
}
