// fedagg_admm.hip -- gfx950 kernels and C ABI of the ADMM server updates (IIADMM / ICEADMM).
//
// Like the FedAvg-family kernels these are HBM-bound streaming kernels: a thread owns whole
// 16-byte vectors of the flat parameter axis and walks the K clients for them in client order,
// so every client byte is read once.  Per client and element IIADMM reads the primal, the
// server-held dual and (for the dual residual) the previous primal and writes the dual back:
// 16 bytes per client and parameter, a quarter of them written -- a copy-like mix, unlike the
// FedAvg round.  ICEADMM reads three streams per client and writes only g.
//
// The two residual norms the aggregators report are sums of squares over all elements: every
// thread keeps float64 partial sums (the squares themselves are fp32, as torch.square's are),
// a block folds them in a fixed order and writes ONE pair per block; a one-block kernel adds the
// pairs in a fixed order.  No atomics: the same launch shape gives the same bits.
//
// Built with -ffp-contract=off: the formulas are sequences of separately rounded fp32 ops in
// the order of iiadmm_aggregator.py:151-162 and iceadmm_aggregator.py:151-166.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/appfl_amd.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

// Launch shapes (vectors per thread, clients whose loads are in flight together).  Short vectors:
// 8-KiB tiles so that every CU gets blocks.  Long ones: longer contiguous runs per row and visit
// -- IIADMM, which rewrites K dual rows in place, does best one client at a time over 32-KiB
// tiles (measured on 25.5 M elements x 16 clients: 0.635 -> 0.653 of peak), ICEADMM with 16-KiB
// tiles and four clients ahead (0.685 -> 0.70).
#ifndef FA_ADMM_AHEAD
#define FA_ADMM_AHEAD 4
#endif
#ifndef FA_ADMM_VPT
#define FA_ADMM_VPT 2
#endif
#ifndef FA_ADMM_VPT_II
#define FA_ADMM_VPT_II 8
#endif
#ifndef FA_ADMM_AHEAD_II
#define FA_ADMM_AHEAD_II 1
#endif
#ifndef FA_ADMM_VPT_ICE
#define FA_ADMM_VPT_ICE 4
#endif
#ifndef FA_ADMM_AHEAD_ICE
#define FA_ADMM_AHEAD_ICE 4
#endif
#ifndef FA_ADMM_DUAL_NT
#define FA_ADMM_DUAL_NT 1
#endif
#ifndef FA_ADMM_WT
#define FA_ADMM_WT 1
#endif
#ifndef FA_ADMM_XCD
#define FA_ADMM_XCD 32
#endif
#ifndef FA_ADMM_GRID
#define FA_ADMM_GRID FA_ADMM_BLOCKS
#endif
constexpr int kThreads = 256;
constexpr unsigned kLongShapeMinTiles = 1024;   // the long-vector shape needs this many of its tiles
constexpr unsigned kMaxBlocks = FA_ADMM_BLOCKS; // one pair of partial sums per block
constexpr size_t kSegment = size_t(1) << 29;    // elements one launch indexes with 32-bit vector indices

enum { kII = 0, kICE = 1 };

struct AdmmArgs {
    const float* x[FA_ADMM_KCHUNK];   // primal rows
    float* d[FA_ADMM_KCHUNK];         // dual rows (IIADMM: updated in place; ICEADMM: read)
    const float* p[FA_ADMM_KCHUNK];   // previous primal rows (only read when has_prev)
    float rho[FA_ADMM_KCHUNK];        // penalty as fp32
    float c1[FA_ADMM_KCHUNK];         // IIADMM: fp32(1/penalty); ICEADMM: fp32(penalty/total)
    float c2;                         // ICEADMM: fp32(1/total)
    float nclients;                   // IIADMM: divisor of the sum
    float* g;
    const float* acc_in;              // sums of earlier client chunks (NULL: start from 0)
    float* acc_out;                   // where a non-final chunk leaves its sum
    const float* res_in;              // same for the dual-residual vector
    float* res_out;
    double* partial;                  // [gridDim.x][2]: this launch's sums of squares per block
    unsigned n;                       // elements of width VEC
    int K;
    int last;
    int has_prev;
    int wt;                           // dual rows leave with write-through stores
};

template <int VEC>
struct Pk {
    float e[VEC];
};

template <int VEC>
__device__ __forceinline__ Pk<VEC> ld(const float* p, unsigned i, bool nt) {
    Pk<VEC> r;
    if constexpr (VEC == 4) {
        const f4* q = reinterpret_cast<const f4*>(p) + i;
        f4 t = nt ? __builtin_nontemporal_load(q) : *q;
        r.e[0] = t.x; r.e[1] = t.y; r.e[2] = t.z; r.e[3] = t.w;
    } else {
        r.e[0] = nt ? __builtin_nontemporal_load(p + i) : p[i];
    }
    return r;
}

template <int VEC>
__device__ __forceinline__ void st(float* p, unsigned i, const Pk<VEC>& r, int wt) {
    if constexpr (VEC == 4) {
        f4 t;
        t.x = r.e[0]; t.y = r.e[1]; t.z = r.e[2]; t.w = r.e[3];
        if (wt) {
            // 4 GiB window: a launch addresses at most kSegment floats = 2 GiB from p
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p, 0, -1, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, t), rs, i * 16u, 0, 17);
        } else {
            reinterpret_cast<f4*>(p)[i] = t;
        }
    } else {
        p[i] = r.e[0];
    }
}

// Sum over the block in a fixed order: lanes by halving shuffles, then the four waves in turn.
__device__ __forceinline__ double block_sum(double v, double* lds) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) lds[wave] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) s += lds[w];
    }
    __syncthreads();
    return s;
}

// One client's contribution to the vectors a thread owns; everything stays in registers.
template <int ALG, int VEC, bool PREV>
__device__ __forceinline__ void admm_fold(const AdmmArgs& a, int k, const Pk<VEC>& g, const Pk<VEC>& xv,
                                          Pk<VEC>& du, const Pk<VEC>& pv, Pk<VEC>& acc, Pk<VEC>& r,
                                          bool ok, double& prim) {
    const float rho = a.rho[k], c1 = a.c1[k];
#ifdef FA_ADMM_COPY_ONLY
    // experiment: the same streams with next to no arithmetic (what the memory system allows)
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        acc.e[j] = acc.e[j] + xv.e[j];
        if (PREV) r.e[j] = r.e[j] + pv.e[j];
        if (ALG != kII) acc.e[j] = acc.e[j] + du.e[j];
    }
    (void)rho; (void)c1; (void)g; (void)ok; (void)prim;
    return;
#endif
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const float x = xv.e[j];
        const float d = g.e[j] - x;
        const float d2 = d * d;
        if (ok) prim += static_cast<double>(d2);
        if (ALG == kII) {
            const float pd = rho * d;
            const float dn = du.e[j] + pd;
            du.e[j] = dn;
            const float tt = c1 * dn;
            const float uu = x - tt;
            acc.e[j] = acc.e[j] + uu;
        } else {
            const float aa = c1 * x;
            const float bb = a.c2 * du.e[j];
            const float ss = aa + bb;
            acc.e[j] = acc.e[j] + ss;
        }
        if (PREV) {
            const float dp = pv.e[j] - x;
            const float tp = rho * dp;
            r.e[j] = r.e[j] + tp;
        }
    }
}

// PREV: the previous-primal rows are read and the dual residual is formed.
template <int ALG, int VEC, int VPT, bool PREV, int kAhead>
__global__ __launch_bounds__(kThreads) void admm_kernel(const AdmmArgs a) {
    __shared__ double lds[kThreads / 64];
    constexpr unsigned kTile = kThreads * VPT;
    const unsigned tiles = (a.n + kTile - 1) / kTile;
    double prim = 0.0, dual2 = 0.0;
    for (unsigned t0 = blockIdx.x; t0 < tiles; t0 += gridDim.x) {
        unsigned t = t0;
        // Blocks go round-robin to the 8 XCDs: inside windows of 8*R tiles every XCD takes R
        // consecutive tiles instead of every eighth one (measured +2-3 %); a ragged last window
        // keeps the plain order.
        if (VEC == 4) {
            constexpr unsigned R = FA_ADMM_XCD;
            const unsigned win = t0 & ~(8u * R - 1u);
            if (win + 8u * R <= tiles) t = win + (t0 & 7u) * R + ((t0 >> 3) & (R - 1u));
        }
        unsigned idx[VPT];
        bool ok[VPT];
#pragma unroll
        for (int q = 0; q < VPT; ++q) {
            const unsigned i = t * kTile + q * kThreads + threadIdx.x;
            ok[q] = i < a.n;
            idx[q] = ok[q] ? i : a.n - 1;   // idle lanes redo the last vector and store nothing
        }
        Pk<VEC> g[VPT], acc[VPT], r[VPT];
#pragma unroll
        for (int q = 0; q < VPT; ++q) {
            g[q] = ld<VEC>(a.g, idx[q], false);
            if (a.acc_in) {
                acc[q] = ld<VEC>(a.acc_in, idx[q], false);
            } else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[q].e[j] = 0.0f;
            }
            if (PREV && a.res_in) {
                r[q] = ld<VEC>(a.res_in, idx[q], false);
            } else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) r[q].e[j] = 0.0f;
            }
        }
        int k = 0;
        // whole batches: the client indices are k .. k + kAhead - 1 with no clamping, so the row
        // pointers of a batch come in with a few wide scalar loads
        for (; k + kAhead <= a.K; k += kAhead) {
            Pk<VEC> xv[kAhead][VPT], dv[kAhead][VPT], pv[kAhead][VPT];
#pragma unroll
            for (int u = 0; u < kAhead; ++u) {
#pragma unroll
                for (int q = 0; q < VPT; ++q) {
                    xv[u][q] = ld<VEC>(a.x[k + u], idx[q], true);
                    dv[u][q] = ld<VEC>(a.d[k + u], idx[q], FA_ADMM_DUAL_NT != 0);
                    if (PREV) pv[u][q] = ld<VEC>(a.p[k + u], idx[q], true);
                }
            }
#pragma unroll
            for (int u = 0; u < kAhead; ++u) {
#pragma unroll
                for (int q = 0; q < VPT; ++q) {
                    admm_fold<ALG, VEC, PREV>(a, k + u, g[q], xv[u][q], dv[u][q], pv[u][q], acc[q], r[q], ok[q], prim);
                    if (ALG == kII && ok[q]) st<VEC>(a.d[k + u], idx[q], dv[u][q], a.wt);
                }
            }
        }
        for (; k < a.K; ++k) {
            Pk<VEC> xv[VPT], dv[VPT], pv[VPT];
#pragma unroll
            for (int q = 0; q < VPT; ++q) {
                xv[q] = ld<VEC>(a.x[k], idx[q], true);
                dv[q] = ld<VEC>(a.d[k], idx[q], FA_ADMM_DUAL_NT != 0);
                if (PREV) pv[q] = ld<VEC>(a.p[k], idx[q], true);
            }
#pragma unroll
            for (int q = 0; q < VPT; ++q) {
                admm_fold<ALG, VEC, PREV>(a, k, g[q], xv[q], dv[q], pv[q], acc[q], r[q], ok[q], prim);
                if (ALG == kII && ok[q]) st<VEC>(a.d[k], idx[q], dv[q], a.wt);
            }
        }
#pragma unroll
        for (int q = 0; q < VPT; ++q) {
            if (!ok[q]) continue;
            if (!a.last) {
                st<VEC>(a.acc_out, idx[q], acc[q], a.wt);
                if (PREV) st<VEC>(a.res_out, idx[q], r[q], a.wt);
                continue;
            }
            Pk<VEC> out;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                out.e[j] = (ALG == kII) ? acc[q].e[j] / a.nclients : acc[q].e[j];
                const float r2 = r[q].e[j] * r[q].e[j];
                if (PREV) dual2 += static_cast<double>(r2);
            }
            st<VEC>(a.g, idx[q], out, a.wt);
        }
    }
    const double ps = block_sum(prim, lds);
    const double ds = block_sum(dual2, lds);
    if (threadIdx.x == 0) {
        a.partial[2 * blockIdx.x] = ps;
        a.partial[2 * blockIdx.x + 1] = ds;
    }
}

// Adds the `blocks` pairs a launch left in ws[2..] to the running totals ws[0..1] (`first`: the
// totals start from zero), in a fixed order; `finish`: also writes the two norms.
__global__ __launch_bounds__(kThreads) void admm_totals_kernel(double* ws, unsigned blocks, int first,
                                                             int finish, double* norms) {
    __shared__ double lds[kThreads / 64];
    double p = 0.0, d = 0.0;
    for (unsigned b = threadIdx.x; b < blocks; b += kThreads) {
        p += ws[2 + 2 * b];
        d += ws[2 + 2 * b + 1];
    }
    const double ps = block_sum(p, lds);
    const double ds = block_sum(d, lds);
    if (threadIdx.x == 0) {
        const double tp = (first ? 0.0 : ws[0]) + ps, td = (first ? 0.0 : ws[1]) + ds;
        ws[0] = tp;
        ws[1] = td;
        if (finish) {
            norms[0] = sqrt(tp);
            norms[1] = sqrt(td);
        }
    }
}

inline int hip_code(hipError_t e) { return e == hipSuccess ? FA_OK : FA_ERR_HIP - static_cast<int>(e); }
inline bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

template <int ALG, bool PREV>
int launch_admm_p(const AdmmArgs& a, bool vec, hipStream_t st) {
    if (vec) {
        constexpr int LV = ALG == kII ? FA_ADMM_VPT_II : FA_ADMM_VPT_ICE;
        constexpr int LA = ALG == kII ? FA_ADMM_AHEAD_II : FA_ADMM_AHEAD_ICE;
        const size_t long_tiles = (static_cast<size_t>(a.n) + kThreads * LV - 1) / (kThreads * LV);
        if (long_tiles >= kLongShapeMinTiles) {
            const unsigned grid = static_cast<unsigned>(long_tiles > FA_ADMM_GRID ? FA_ADMM_GRID : long_tiles);
            hipLaunchKernelGGL((admm_kernel<ALG, 4, LV, PREV, LA>), dim3(grid), dim3(kThreads), 0, st, a);
            return static_cast<int>(grid);
        }
        constexpr int VPT = FA_ADMM_VPT;
        const size_t tiles = (static_cast<size_t>(a.n) + kThreads * VPT - 1) / (kThreads * VPT);
        const unsigned grid = static_cast<unsigned>(tiles > FA_ADMM_GRID ? FA_ADMM_GRID : tiles);
        hipLaunchKernelGGL((admm_kernel<ALG, 4, VPT, PREV, FA_ADMM_AHEAD>), dim3(grid), dim3(kThreads), 0, st, a);
        return static_cast<int>(grid);
    }
    const size_t tiles = (static_cast<size_t>(a.n) + kThreads - 1) / kThreads;
    const unsigned grid = static_cast<unsigned>(tiles > kMaxBlocks ? kMaxBlocks : tiles);
    hipLaunchKernelGGL((admm_kernel<ALG, 1, 1, PREV, FA_ADMM_AHEAD>), dim3(grid), dim3(kThreads), 0, st, a);
    return static_cast<int>(grid);
}

template <int ALG>
int launch_admm(const AdmmArgs& a, bool vec, hipStream_t st) {
    return a.has_prev ? launch_admm_p<ALG, true>(a, vec, st) : launch_admm_p<ALG, false>(a, vec, st);
}

int admm_dispatch(int alg, const float* const* primal, float* const* dual, const float* const* prev,
                  const double* penalty, int K, int num_clients, float* g, size_t n, float* carry_ws,
                  double* res_ws, double* norms, hipStream_t st) {
    if (!primal || !dual || !penalty || !g || !res_ws || !norms || K <= 0) return FA_ERR_INVALID;
    if (alg == kII && num_clients <= 0) return FA_ERR_INVALID;
    if (K > FA_ADMM_KCHUNK && !carry_ws) return FA_ERR_WORKSPACE;
    if (!aligned_to(res_ws, 8) || !aligned_to(norms, 8)) return FA_ERR_ALIGN;
    bool a4 = aligned_to(g, 4) && (!carry_ws || aligned_to(carry_ws, 4));
    bool a16 = aligned_to(g, 16) && (!carry_ws || aligned_to(carry_ws, 16));
    double total = 0.0;
    for (int k = 0; k < K; ++k) {
        if (!primal[k] || !dual[k] || (prev && !prev[k])) return FA_ERR_INVALID;
        a4 = a4 && aligned_to(primal[k], 4) && aligned_to(dual[k], 4) && (!prev || aligned_to(prev[k], 4));
        a16 = a16 && aligned_to(primal[k], 16) && aligned_to(dual[k], 16) && (!prev || aligned_to(prev[k], 16));
        total += penalty[k];   // iceadmm_aggregator.py:147-149: a Python sum in client order
    }
    if (!a4) return FA_ERR_ALIGN;
    hipError_t e = hipSuccess;
    if (n == 0) {
        hipLaunchKernelGGL(admm_totals_kernel, dim3(1), dim3(kThreads), 0, st, res_ws, 0u, 1, 1, norms);
        return hip_code(hipGetLastError());
    }
    // ranges of the flat axis: <= kSegment pieces, each a 128-bit body and a scalar tail
    struct Range { size_t off, cnt; bool vec; };
    Range ranges[2 * 64];
    int nr = 0;
    for (size_t off = 0; off < n; off += kSegment) {
        const size_t len = (n - off < kSegment) ? n - off : kSegment;
        const size_t body = a16 ? (len & ~size_t(3)) : 0;
        if (nr + 2 > 128) return FA_ERR_INVALID;
        if (body) ranges[nr++] = {off, body / 4, true};
        if (len > body) ranges[nr++] = {off + body, len - body, false};
    }
    float* acc_ws = carry_ws;
    float* rvec_ws = carry_ws ? carry_ws + (n + 3) / 4 * 4 : nullptr;   // 16-byte aligned like acc_ws
    int first = 1;
    for (int ri = 0; ri < nr; ++ri) {
        const Range& rg = ranges[ri];
        for (int k0 = 0; k0 < K; k0 += FA_ADMM_KCHUNK) {
            const int kc = (K - k0 < FA_ADMM_KCHUNK) ? K - k0 : FA_ADMM_KCHUNK;
            AdmmArgs a;
            memset(&a, 0, sizeof(a));
            for (int k = 0; k < kc; ++k) {
                a.x[k] = primal[k0 + k] + rg.off;
                a.d[k] = dual[k0 + k] + rg.off;
                a.p[k] = prev ? prev[k0 + k] + rg.off : nullptr;
                a.rho[k] = static_cast<float>(penalty[k0 + k]);
                a.c1[k] = alg == kII ? static_cast<float>(1.0 / penalty[k0 + k])
                                     : static_cast<float>(penalty[k0 + k] / total);
            }
            a.c2 = static_cast<float>(1.0 / total);
            a.nclients = static_cast<float>(num_clients);
            a.g = g + rg.off;
            a.last = (k0 + kc == K);
            a.acc_in = k0 ? acc_ws + rg.off : nullptr;
            a.acc_out = a.last ? nullptr : acc_ws + rg.off;
            a.res_in = k0 ? rvec_ws + rg.off : nullptr;
            a.res_out = a.last ? nullptr : rvec_ws + rg.off;
            a.partial = res_ws + 2;
            a.n = static_cast<unsigned>(rg.cnt);
            a.K = kc;
            a.has_prev = prev ? 1 : 0;
            a.wt = FA_ADMM_WT;
            const int grid = alg == kII ? launch_admm<kII>(a, rg.vec, st) : launch_admm<kICE>(a, rg.vec, st);
            e = hipGetLastError();
            if (e != hipSuccess) return hip_code(e);
            const int finish = (ri == nr - 1 && a.last) ? 1 : 0;
            hipLaunchKernelGGL(admm_totals_kernel, dim3(1), dim3(kThreads), 0, st, res_ws,
                               static_cast<unsigned>(grid), first, finish, norms);
            first = 0;
            e = hipGetLastError();
            if (e != hipSuccess) return hip_code(e);
        }
    }
    return FA_OK;
}

}  // namespace

extern "C" {

int fa_admm_ii(const float* const* primal, float* const* dual, const float* const* prev,
               const double* penalty, int K, int num_clients, float* g, size_t n, float* carry_ws,
               double* res_ws, double* norms, fa_stream_t stream) {
    return admm_dispatch(kII, primal, dual, prev, penalty, K, num_clients, g, n, carry_ws, res_ws,
                         norms, reinterpret_cast<hipStream_t>(stream));
}

int fa_admm_ice(const float* const* primal, const float* const* dual, const float* const* prev,
                const double* penalty, int K, float* g, size_t n, float* carry_ws, double* res_ws,
                double* norms, fa_stream_t stream) {
    return admm_dispatch(kICE, primal, const_cast<float* const*>(dual), prev, penalty, K, 1, g, n,
                         carry_ws, res_ws, norms, reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
