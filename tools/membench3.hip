// Development tool: does the layout of the K client rows matter to HBM?  Compares the row-major
// slab [K][N] with tile-interleaved slabs [N/C][K][C] (every block reads K*C contiguous vectors)
// for the reduce kernel's access pattern, over vector lengths and client counts.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/bin/membench3 tools/membench3.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// C = vectors per client per layout tile (0: row-major).  STORE: 0 none, 1 g written through.
template <int STORE, int VPT, int U>
__global__ __launch_bounds__(256) void rk3(const f4* __restrict__ p, size_t n4, int K, unsigned C,
                                           const f4* __restrict__ gp, f4* __restrict__ gout, float w) {
    constexpr unsigned T = 256 * VPT;
    const unsigned tiles = static_cast<unsigned>(n4 / T);
    for (unsigned t = blockIdx.x; t < tiles; t += gridDim.x) {
        size_t i[VPT], xo[VPT];
        for (int q = 0; q < VPT; ++q) {
            i[q] = static_cast<size_t>(t) * T + q * 256 + threadIdx.x;
            xo[q] = C ? (i[q] / C) * (static_cast<size_t>(K) * C) + (i[q] % C) : i[q];
        }
        const size_t kstride = C ? C : n4;
        f4 acc[VPT], g[VPT];
        for (int q = 0; q < VPT; ++q) { acc[q] = f4{0, 0, 0, 0}; g[q] = STORE ? __builtin_nontemporal_load(gp + i[q]) : f4{w, w, w, w}; }
        for (int k = 0; k + U <= K; k += U) {
            f4 x[U][VPT];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int q = 0; q < VPT; ++q) x[u][q] = __builtin_nontemporal_load(p + xo[q] + (k + u) * kstride);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int q = 0; q < VPT; ++q) acc[q] = acc[q] + (x[u][q] - g[q]) * w;
        }
        for (int q = 0; q < VPT; ++q) {
            f4 r = g[q] + acc[q];
            if (STORE) {
                __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(gout + static_cast<size_t>(t) * T), 0, T * 16, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, r), ro, (q * 256 + threadIdx.x) * 16, 0, 17);
            } else if (r.x == 123.456f) {
                gout[i[q]] = r;
            }
        }
    }
}

// Write phase of the phase-separation experiment: `nv` result vectors of vectors [v0, v1).
__global__ __launch_bounds__(256) void write_phase(f4* __restrict__ out, size_t n4, size_t v0, size_t v1, int nv, float w) {
    for (size_t i = v0 + static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < v1; i += static_cast<size_t>(gridDim.x) * 256)
        for (int c = 0; c < nv; ++c) {
            f4 r = {w, w + c, w, w};
            __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(out + c * n4 + (i & ~size_t(255))), 0, 4096, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, r), ro, static_cast<unsigned>(i & 255) * 16, 0, 17);
        }
}

// Read phase: the K-row reduce of vectors [v0, v1) with no store.
template <int VPT, int U>
__global__ __launch_bounds__(256) void read_phase(const f4* __restrict__ p, size_t n4, int K, size_t v0, size_t v1, f4* __restrict__ gout, float w) {
    constexpr unsigned T = 256 * VPT;
    const size_t tiles = (v1 - v0) / T;
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        size_t i[VPT];
        for (int q = 0; q < VPT; ++q) i[q] = v0 + t * T + q * 256 + threadIdx.x;
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
        for (int q = 0; q < VPT; ++q) if (acc[q].x == 123.456f) gout[i[q]] = acc[q];
    }
}

__global__ void fill_rand(unsigned* p, size_t n) {
    size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    size_t s = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (; i < n; i += s) {
        unsigned z = static_cast<unsigned>(i) * 2654435761u + 12345u;
        z ^= z >> 15; z *= 2246822519u; z ^= z >> 13;
        p[i] = 0x3C000000u | (z & 0x007FFFFFu);       // floats around 0.01
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

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 9;
    const int only = argc > 2 ? atoi(argv[2]) : -1;          // index of the one case to run
    const int only_layout = argc > 3 ? atoi(argv[3]) : -1;   // C of the one layout to run
    struct Case { size_t n; int K; };
    const Case cases[] = {{11689512, 32}, {11689512, 100}, {25557032, 64}, {86567656, 32}, {162500000, 32}, {162500000, 64}};
    const unsigned layouts[] = {0, 1024, 4096, 16384};
    if (only == -2) {
        // marginal cost of the result write as the client count (the read share) varies
        const size_t n4 = 25557032 / 4 / 16384 * 16384;
        const int Kmax = 128;
        f4 *slab, *g, *out;
        CK(hipMalloc(&slab, n4 * 16 * Kmax)); CK(hipMalloc(&g, n4 * 16)); CK(hipMalloc(&out, n4 * 16));
        fill_rand<<<4096, 256>>>(reinterpret_cast<unsigned*>(slab), n4 * 4 * Kmax);
        fill_rand<<<4096, 256>>>(reinterpret_cast<unsigned*>(g), n4 * 4);
        CK(hipDeviceSynchronize());
        const unsigned grid4 = static_cast<unsigned>(std::min<size_t>(n4 / 1024, 65536));
        for (int K : {8, 16, 32, 64, 128}) {
            float r0 = median_ms([&] { rk3<0, 4, 8><<<grid4, 256>>>(slab, n4, K, 0, g, out, 0.01f); }, reps);
            float r1 = median_ms([&] { rk3<1, 4, 8><<<grid4, 256>>>(slab, n4, K, 0, g, out, 0.01f); }, reps);
            const double wb = double(n4) * 16;
            printf("K=%3d: read-only %.3f ms %5.0f GB/s | with g read+write %.3f ms | extra %.3f ms for %.0f MB read + %.0f MB written = %.0f GB/s marginal, write share %.1f %%\n",
                   K, r0, double(K) * n4 * 16 / r0 / 1e6, r1, r1 - r0, wb / 1e6, wb / 1e6, 2 * wb / (r1 - r0) / 1e6, 100.0 / (K + 2));
        }
        return 0;
    }
    if (only == -3) {
        // phase separation: read-only kernels over a chunk of the vector alternate with write-only
        // kernels of the chunk's results -- what perfectly synchronised write phases would cost
        for (int K : {64, 32}) {
            const size_t n4 = (K == 64 ? 25557032 : 162500000) / 4 / 16384 * 16384;
            f4 *slab, *out;
            CK(hipMalloc(&slab, n4 * 16 * K)); CK(hipMalloc(&out, n4 * 16 * 3));
            fill_rand<<<4096, 256>>>(reinterpret_cast<unsigned*>(slab), n4 * 4 * K);
            CK(hipDeviceSynchronize());
            for (int nv : {1, 3}) {
                for (size_t chunk_mb : {4, 9, 18, 36, 72, 144, 100000}) {
                    size_t cv = std::min(n4, chunk_mb * 1000000 / 16 / 1024 * 1024);
                    float t = median_ms([&] {
                        for (size_t v0 = 0; v0 < n4; v0 += cv) {
                            size_t v1 = std::min(n4, v0 + cv);
                            read_phase<4, 8><<<static_cast<unsigned>(std::min<size_t>((v1 - v0) / 1024, 65536)), 256>>>(slab, n4, K, v0, v1, out, 0.01f);
                            write_phase<<<static_cast<unsigned>(std::min<size_t>((v1 - v0) / 256, 4096)), 256>>>(out, n4, v0, v1, nv, 0.01f);
                        }
                    }, reps);
                    const double alg = double(K + 2 * nv) * n4 * 16;    // counts the state reads this experiment skips
                    printf("K=%d n=%.0fM nv=%d chunk %6zu MB/vector (%zu phases): %.3f ms  alg frac %.3f\n", K, n4 * 4 / 1e6, nv,
                           cv * 16 / 1000000, (n4 + cv - 1) / cv, t, alg / t / 1e6 / 8000);
                    fflush(stdout);
                    if (cv == n4) break;
                }
            }
            CK(hipFree(slab)); CK(hipFree(out));
        }
        return 0;
    }
    int ci = -1;
    for (const Case& c : cases) {
        if (++ci != only && only >= 0) continue;
        const size_t n4 = c.n / 4 / 16384 * 16384;       // whole layout tiles of every size
        f4 *slab, *g, *out;
        CK(hipMalloc(&slab, n4 * 16 * c.K)); CK(hipMalloc(&g, n4 * 16)); CK(hipMalloc(&out, n4 * 16));
        fill_rand<<<4096, 256>>>(reinterpret_cast<unsigned*>(slab), n4 * 4 * c.K);
        fill_rand<<<4096, 256>>>(reinterpret_cast<unsigned*>(g), n4 * 4);
        CK(hipDeviceSynchronize());
        printf("== n=%zu floats (%.1f MB rows) K=%d\n", n4 * 4, n4 * 16 / 1e6, c.K);
        for (unsigned C : layouts) {
            if (only_layout >= 0 && C != static_cast<unsigned>(only_layout)) continue;
            const unsigned grid4 = static_cast<unsigned>(std::min<size_t>(n4 / 1024, 65536));
            float r0 = median_ms([&] { rk3<0, 4, 8><<<grid4, 256>>>(slab, n4, c.K, C, g, out, 0.01f); }, reps);
            float r1 = median_ms([&] { rk3<1, 4, 8><<<grid4, 256>>>(slab, n4, c.K, C, g, out, 0.01f); }, reps);
            float p1 = median_ms([&] { rk3<1, 4, 8><<<512, 256>>>(slab, n4, c.K, C, g, out, 0.01f); }, reps);
            const unsigned grid2 = static_cast<unsigned>(std::min<size_t>(n4 / 512, 65536));
            float v2 = median_ms([&] { rk3<1, 2, 16><<<grid2, 256>>>(slab, n4, c.K, C, g, out, 0.01f); }, reps);
            const double rb = double(c.K) * n4 * 16, sb = double(c.K + 2) * n4 * 16;
            printf("  C=%5u: read-only %.3f ms %5.0f GB/s | +g r/w %.3f ms %5.0f GB/s (frac %.3f) | grid 512: %.3f ms frac %.3f | vpt2 u16: %.3f ms frac %.3f\n",
                   C, r0, rb / r0 / 1e6, r1, sb / r1 / 1e6, sb / r1 / 1e6 / 8000, p1, sb / p1 / 1e6 / 8000, v2, sb / v2 / 1e6 / 8000);
            fflush(stdout);
        }
        CK(hipFree(slab)); CK(hipFree(g)); CK(hipFree(out));
    }
    return 0;
}
