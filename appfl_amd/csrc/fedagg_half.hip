// fedagg_half.hip -- the aggregation kernels for bfloat16 / float16 state-dict entries.
//
// The reference's float path takes any float tensor (misc/memory_utils.py:112-131).  On 16-bit
// float tensors PyTorch's CPU kernels evaluate every single tensor operation in fp32 and round
// its result to the 16-bit type (nearest, ties to even); nothing is fused across operations.  A
// Python scalar that multiplies or divides such a tensor enters as its fp32 value; one that is
// added (tau, tau**2) or passed as `alpha=` of add_ / sub_ is first rounded to the tensor's type.
//
// Everything stays 16 bits wide in HBM -- client rows, global state, moments, steps -- so a
// 16-bit model moves half the bytes of an fp32 one.  A kernel widens what it loads (exact),
// evaluates the fp32 formula of the entry point of the same name with each operation's result
// passed through rnd<HALF>() -- the hardware's round-to-nearest-even conversion to the 16-bit
// type and back -- and narrows what it stores (exact again: every stored value came out of
// rnd).  Results equal the reference's bit for bit.  Streaming kernels: 128-bit loads of eight
// elements where the buffers allow it, eight client loads in flight; the rounds of long client
// lists on bfloat16 run on a persistent kernel that writes its results in chip-wide bursts.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/appfl_amd.h"
#include "internal.h"

namespace {

typedef unsigned u4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 1 << 20;
constexpr int kAhead = 8;   // client loads in flight per element / vector
constexpr int kWide = 8;    // elements of a 128-bit load
constexpr size_t kSegmentH = size_t(1) << 30;   // elements one reduce launch covers at most

#include "grid_meet.h"

// Launch shape of the 128-bit reduce kernel: vectors per thread and client loads in flight per
// vector (build-time knobs for A/B runs; the defaults are the measured best).  ResNet-50 x 64
// clients, FedAdam, bfloat16, kernel time: 1 x 8 0.623 ms, 2 x 8 / 2 x 4 0.637, 4 x 4 0.631,
// 1 x 16 0.639 (2 x 16 and 4 x 8 spill); streaming loads of m / v 1.5 % slower; write-through
// result stores 2.4 % faster (0.620 -> 0.606 ms); XCD runs of 32 or 128 tiles 0.7 % faster; with every rounding
// compiled out 0.611 ms -- the roundings are not what bounds it.  With the loads in flight kept
// packed (so that deeper batches fit in registers): 1 x 8 / 1 x 16 / 1 x 32 0.621, 2 x 4 0.622,
// 2 x 8 0.625, 4 x 8 0.630, 2 x 16 0.634, 4 x 16 0.719: the shape is not what bounds it either;
// this is where the fp32 tile-per-block kernel sits for the same round (0.70-0.72 of peak); long
// client lists on bfloat16 therefore take the persistent kernel further down
// (reduce_late_half_kernel), which parks a phase's results on chip and writes them as one burst.
#ifndef FA_HALF_VPT
#define FA_HALF_VPT 1
#endif
#ifndef FA_HALF_AHEAD
#define FA_HALF_AHEAD 8
#endif
#ifndef FA_HALF_NT_STATE
#define FA_HALF_NT_STATE 0
#endif
#ifndef FA_HALF_XCD_RUN
#define FA_HALF_XCD_RUN 32     // > 0: inside windows of 8 * RUN tiles each XCD takes RUN consecutive tiles
#endif
#ifndef FA_HALF_WT
#define FA_HALF_WT 1           // 1: results leave with write-through stores
#endif

template <int HALF> struct Elem;
template <> struct Elem<FA_HALF_BF16> { typedef __bf16 T; };
template <> struct Elem<FA_HALF_F16> { typedef _Float16 T; };

// fp32 -> 16-bit type (round to nearest even) -> fp32.  The empty asm makes the fp32 value
// opaque to the compiler, so that the conversion stays a conversion of its own.  Left alone, the
// compiler merges a multiply and the narrowing that follows it into one mixed-precision
// instruction (v_fma_mixlo_f16), which rounds the EXACT product once (tools/fma_mix_probe.hip:
// 4 194 304 of 4 194 304 products), while the reference rounds twice, to fp32 and then to fp16.
// The two differ whenever the fp32 rounding lands on an fp16 tie, and with a factor like
// beta1 = 0.9 that is no rarity: 0.9f * m sits just below a tie for every tenth fp16 m, so
// 4 % of the products m * 0.9f came out one unit in the last place apart.
template <int HALF>
__device__ __forceinline__ float rnd(float x) {
#ifdef FA_HALF_DEBUG_NO_ROUNDING      // timing experiment only: wrong results
    return x;
#endif
    if constexpr (HALF == FA_HALF_BF16) {
        // The packed conversion of the pair (0, x) leaves x's bfloat16 in the HIGH half of the
        // dword and zero below it -- which is the rounded value as an fp32 already: one
        // instruction per rounding instead of a conversion and a shift (no instruction fuses a
        // multiply with a conversion to bfloat16, so nothing needs hiding from the compiler here).
        typedef float f2 __attribute__((ext_vector_type(2)));
        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
        const f2 pair = {0.0f, x};
        return __builtin_bit_cast(float, __builtin_convertvector(pair, b2));
    } else {
        __asm__("" : "+v"(x));
        return static_cast<float>(static_cast<_Float16>(x));
    }
}

template <int HALF>
float rnd_host(float x) {
    if (HALF == FA_HALF_BF16) {
        uint32_t u;
        memcpy(&u, &x, 4);
        if ((u & 0x7FFFFFFFu) > 0x7F800000u) return x;   // NaN
        u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
        float r;
        memcpy(&r, &u, 4);
        return r;
    }
    return static_cast<float>(static_cast<_Float16>(x));
}

float rnd_host(float x, int half) { return half == FA_HALF_BF16 ? rnd_host<FA_HALF_BF16>(x) : rnd_host<FA_HALF_F16>(x); }

struct HyperH {
    float lr, b1, omb1, b2, omb2, tau;   // tau already rounded to the 16-bit type (it is ADDED)
};

template <int VEC>
struct PackH {
    float e[VEC];
};

// two 16-bit elements of one dword, widened (exact)
template <int HALF>
__device__ __forceinline__ void widen2(unsigned u, float& lo, float& hi) {
    if constexpr (HALF == FA_HALF_BF16) {
        lo = __builtin_bit_cast(float, u << 16);
        hi = __builtin_bit_cast(float, u & 0xFFFF0000u);
    } else {
        lo = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<unsigned short>(u & 0xFFFFu)));
        hi = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<unsigned short>(u >> 16)));
    }
}

// ... and two values that are representable in the 16-bit type, narrowed into one dword (exact)
template <int HALF>
__device__ __forceinline__ unsigned narrow2(float lo, float hi) {
    if constexpr (HALF == FA_HALF_BF16) {
        return (__builtin_bit_cast(unsigned, lo) >> 16) | (__builtin_bit_cast(unsigned, hi) & 0xFFFF0000u);
    } else {
        const unsigned a = __builtin_bit_cast(unsigned short, static_cast<_Float16>(lo));
        const unsigned b = __builtin_bit_cast(unsigned short, static_cast<_Float16>(hi));
        return a | (b << 16);
    }
}

// What a load of one element of width VEC leaves in registers before it is widened: the four
// dwords of a 128-bit access, or the one 16-bit value.  Loads that wait in flight stay in this
// form (half the registers of the widened one).
template <int VEC> struct Raw { u4 v; };
template <> struct Raw<1> { unsigned short v; };

template <int VEC, bool NT>
__device__ __forceinline__ Raw<VEC> load_raw(const void* p, size_t i) {
    Raw<VEC> r;
    if constexpr (VEC == kWide) {
        const u4* q = static_cast<const u4*>(p) + i;
        r.v = NT ? __builtin_nontemporal_load(q) : *q;
    } else {
        const unsigned short* q = static_cast<const unsigned short*>(p) + i;
        r.v = NT ? __builtin_nontemporal_load(q) : *q;
    }
    return r;
}

template <int HALF, int VEC>
__device__ __forceinline__ PackH<VEC> widen(const Raw<VEC>& t) {
    PackH<VEC> r;
    if constexpr (VEC == kWide) {
        widen2<HALF>(t.v.x, r.e[0], r.e[1]);
        widen2<HALF>(t.v.y, r.e[2], r.e[3]);
        widen2<HALF>(t.v.z, r.e[4], r.e[5]);
        widen2<HALF>(t.v.w, r.e[6], r.e[7]);
    } else {
        float hi;
        widen2<HALF>(t.v, r.e[0], hi);
    }
    return r;
}

// element i of width VEC (8: one 128-bit access, 1: one 16-bit access) of a 16-bit buffer
template <int HALF, int VEC, bool NT>
__device__ __forceinline__ PackH<VEC> load_h(const void* p, size_t i) {
    PackH<VEC> r;
    if constexpr (VEC == kWide) {
        const u4* q = static_cast<const u4*>(p) + i;
        const u4 t = NT ? __builtin_nontemporal_load(q) : *q;
        widen2<HALF>(t.x, r.e[0], r.e[1]);
        widen2<HALF>(t.y, r.e[2], r.e[3]);
        widen2<HALF>(t.z, r.e[4], r.e[5]);
        widen2<HALF>(t.w, r.e[6], r.e[7]);
    } else {
        const unsigned short* q = static_cast<const unsigned short*>(p) + i;
        const unsigned short b = NT ? __builtin_nontemporal_load(q) : *q;
        float hi;
        widen2<HALF>(b, r.e[0], hi);
    }
    return r;
}

template <int HALF, int VEC>
__device__ __forceinline__ void store_h(void* p, size_t i, const PackH<VEC>& r) {
    if constexpr (VEC == kWide) {
        u4 t;
        t.x = narrow2<HALF>(r.e[0], r.e[1]);
        t.y = narrow2<HALF>(r.e[2], r.e[3]);
        t.z = narrow2<HALF>(r.e[4], r.e[5]);
        t.w = narrow2<HALF>(r.e[6], r.e[7]);
        static_cast<u4*>(p)[i] = t;
    } else {
        static_cast<unsigned short*>(p)[i] = static_cast<unsigned short>(narrow2<HALF>(r.e[0], 0.0f) & 0xFFFFu);
    }
}

// Result store of the 128-bit kernels: optionally write-through (sc0 sc1), which keeps result
// lines from lingering dirty in L2 between the streaming reads.
template <int HALF, int VEC>
__device__ __forceinline__ void store_result_h(void* p, size_t i, const PackH<VEC>& r) {
    if constexpr (VEC == kWide && FA_HALF_WT != 0) {
        u4 t;
        t.x = narrow2<HALF>(r.e[0], r.e[1]);
        t.y = narrow2<HALF>(r.e[2], r.e[3]);
        t.z = narrow2<HALF>(r.e[4], r.e[5]);
        t.w = narrow2<HALF>(r.e[6], r.e[7]);
        // the buffer's base is the same for the whole wave (it lives in scalar registers); the
        // lane's element goes into the 32-bit offset, which covers the 4 GiB a launch may address
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p, 0, -1, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(t, rs, static_cast<unsigned>(i) * 16u, 0, 17);
    } else {
        store_h<HALF, VEC>(p, i, r);
    }
}

template <int HALF>
__device__ __forceinline__ float ld(const void* p, size_t i) { return load_h<HALF, 1, false>(p, i).e[0]; }
template <int HALF>
__device__ __forceinline__ float ld_nt(const void* p, size_t i) { return load_h<HALF, 1, true>(p, i).e[0]; }
template <int HALF>
__device__ __forceinline__ void st(void* p, size_t i, float x) {
    PackH<1> r;
    r.e[0] = x;
    store_h<HALF, 1>(p, i, r);
}

__device__ __forceinline__ float sign_h(float d) {
    return static_cast<float>(static_cast<int>(0.0f < d) - static_cast<int>(d < 0.0f));
}

// fedavgm_aggregator.py:48-61, fedadam_aggregator.py:60-99, fedyogi_aggregator.py:60-97,
// fedadagrad_aggregator.py:59-91, one rounding per tensor operation.
template <int OPT, int HALF>
__device__ __forceinline__ float optimizer_step_h(float step, float& m, float& v, const HyperH& h) {
    if (OPT == FA_OPT_NONE) return step;
    if (OPT == FA_OPT_AVGM) {
        float mt = rnd<HALF>(v * h.b1);
        v = rnd<HALF>(mt + step);
        return v;
    }
    float mt = rnd<HALF>(m * h.b1);
    float st = rnd<HALF>(step * h.omb1);
    m = rnd<HALF>(mt + st);
    float s2 = rnd<HALF>(step * step);
    if (OPT == FA_OPT_ADAM) {
        float vt = rnd<HALF>(v * h.b2);
        float sq = rnd<HALF>(s2 * h.omb2);
        v = rnd<HALF>(vt + sq);
    } else if (OPT == FA_OPT_YOGI) {
        float sg = sign_h(rnd<HALF>(v - s2));
        float yu = rnd<HALF>(s2 * sg);
        float ys = rnd<HALF>(yu * h.omb2);
        v = rnd<HALF>(v - ys);
    } else {
        v = rnd<HALF>(v + s2);
    }
    float num = rnd<HALF>(h.lr * m);
    float rt = rnd<HALF>(sqrtf(v));
    float den = rnd<HALF>(rt + h.tau);
    return rnd<HALF>(num / den);
}

struct ReduceHArgs {
    const void* x[FA_KCHUNK];
    float w[FA_KCHUNK];
    void* g;
    const void* g_ro;
    void* m;
    void* v;
    const void* step_in;
    void* step_out;
    size_t n;   // elements of width VEC
    int K, last, apply;
    HyperH h;
    unsigned* sync_word;   // late-epilogue kernel: the counter its blocks meet on (NULL: no meetings)
    unsigned sync_epoch;   // ... and this launch's tag in it
    int sync_groups;       // 2: the odd and the even XCDs' blocks meet apart, half a phase shifted; 1: all together
    int no_persistent;     // host side only: keep this launch on the tile-per-block kernel (FA_ROUTE_NO_PERSISTENT)
};

// One block walks tiles of kThreads * VPT elements of width VEC; thread t owns elements
// t, t + kThreads, ... of the tile, so every load instruction of a wave covers one contiguous KiB.
template <int OPT, int HALF, int VEC, int VPT, int U>
__global__ __launch_bounds__(kThreads) void reduce_half_kernel(const ReduceHArgs a) {
    constexpr bool NTS = FA_HALF_NT_STATE != 0;
    const size_t tile_elems = static_cast<size_t>(kThreads) * VPT;
    const size_t tiles = (a.n + tile_elems - 1) / tile_elems;
    for (size_t t0 = blockIdx.x; t0 < tiles; t0 += gridDim.x) {
        size_t tile = t0;
        if constexpr (FA_HALF_XCD_RUN > 0 && VEC == kWide) {
            // blocks are dispatched round-robin over the 8 XCDs: give each XCD runs of consecutive tiles
            constexpr size_t R = FA_HALF_XCD_RUN;
            const size_t win = t0 & ~(8 * R - 1);
            if (win + 8 * R <= tiles) tile = win + (t0 & 7) * R + ((t0 >> 3) & (R - 1));
        }
        size_t idx[VPT], at_[VPT];
#pragma unroll
        for (int q = 0; q < VPT; ++q) {
            idx[q] = tile * tile_elems + static_cast<size_t>(q) * kThreads + threadIdx.x;
            at_[q] = idx[q] < a.n ? idx[q] : a.n - 1;      // loads of a ragged last tile stay in bounds
        }
        PackH<VEC> g[VPT], s[VPT];
#pragma unroll
        for (int q = 0; q < VPT; ++q) {
            g[q] = load_h<HALF, VEC, false>(a.apply ? a.g : a.g_ro, at_[q]);
            if (a.step_in) {
                s[q] = load_h<HALF, VEC, false>(a.step_in, at_[q]);
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) s[q].e[e] = 0.0f;
            }
        }
        int k = 0;
        for (; k + U <= a.K; k += U) {
            Raw<VEC> xr[U][VPT];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int q = 0; q < VPT; ++q) xr[u][q] = load_raw<VEC, true>(a.x[k + u], at_[q]);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float w = a.w[k + u];
#pragma unroll
                for (int q = 0; q < VPT; ++q) {
                    const PackH<VEC> x = widen<HALF, VEC>(xr[u][q]);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        float d = rnd<HALF>(x.e[e] - g[q].e[e]);
                        float wd = rnd<HALF>(d * w);
                        s[q].e[e] = rnd<HALF>(s[q].e[e] + wd);
                    }
                }
            }
        }
        for (; k < a.K; ++k) {
            const float w = a.w[k];
#pragma unroll
            for (int q = 0; q < VPT; ++q) {
                const PackH<VEC> x = load_h<HALF, VEC, true>(a.x[k], at_[q]);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    float d = rnd<HALF>(x.e[e] - g[q].e[e]);
                    float wd = rnd<HALF>(d * w);
                    s[q].e[e] = rnd<HALF>(s[q].e[e] + wd);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < VPT; ++q) {
            if (idx[q] >= a.n) continue;
            const size_t i = idx[q];
            if (!a.last) {
                store_h<HALF, VEC>(a.step_out, i, s[q]);
                continue;
            }
            if (OPT != FA_OPT_NONE) {
                PackH<VEC> v = load_h<HALF, VEC, NTS>(a.v, i);
                PackH<VEC> m;
                if (OPT != FA_OPT_AVGM) {
                    m = load_h<HALF, VEC, NTS>(a.m, i);
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) m.e[e] = 0.0f;
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e) s[q].e[e] = optimizer_step_h<OPT, HALF>(s[q].e[e], m.e[e], v.e[e], a.h);
                store_result_h<HALF, VEC>(a.v, i, v);
                if (OPT != FA_OPT_AVGM) store_result_h<HALF, VEC>(a.m, i, m);
            }
            if (a.step_out) store_result_h<HALF, VEC>(a.step_out, i, s[q]);
            if (a.apply) {
                PackH<VEC> r;
#pragma unroll
                for (int e = 0; e < VEC; ++e) r.e[e] = rnd<HALF>(g[q].e[e] + s[q].e[e]);
                store_result_h<HALF, VEC>(a.g, i, r);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Persistent variant with a LATE optimizer epilogue (the 16-bit counterpart of the fp32 library's
// kernel of that kind): one block per CU takes every gridDim.x-th tile of 256 x 4 vectors, folds
// its clients into the averaged step alone -- 8 clients x 4 vectors = 32 packed 128-bit loads in
// flight per lane at one wave per SIMD --, parks the step (16 bits per element) in LDS and, every
// kLateTiles tiles, runs the epilogue over the parked tiles: g (for the second time), m and v are
// read, the optimizer formulas evaluated, g, m and v written.  A phase is a long stretch of pure
// reads followed by a copy-like burst instead of three thin result streams among the reads.
// Same operations on the same values in the same order as reduce_half_kernel: same bits.
#ifndef FA_HALF_LATE_TILES
#define FA_HALF_LATE_TILES 10
#endif
#ifndef FA_HALF_LATE_OVERLAP
#define FA_HALF_LATE_OVERLAP 1      /* 0: one buffer, a batch's loads are awaited before the next batch is requested (A/B knob; measured 5-13 % slower at 2 clients a batch, 0-4 % slower at 4) */
#endif
#ifndef FA_HALF_LATE_AHEAD
#define FA_HALF_LATE_AHEAD 2
#endif
constexpr int kLateTiles = FA_HALF_LATE_TILES;
#ifndef FA_HALF_LATE_VPT
#define FA_HALF_LATE_VPT 4
#endif
constexpr int kLateVpt = FA_HALF_LATE_VPT;

// g and the running averaged step of one 16-byte vector (eight elements) in the persistent
// kernel.  bfloat16: widened to fp32, every operation rounded by rnd<>.  float16: kept packed, and
// the two tensor-tensor operations of a client's term -- x - g and step + term -- run as native
// packed float16 instructions: PyTorch evaluates them in fp32 and rounds to float16, and for + and
// - of float16 values that double rounding equals the one correct rounding the instruction does
// (fp32 carries 24 >= 2 * 11 + 2 significant bits; sums that fall into the subnormal range are
// exact).  The product with the fp32 weight keeps its two roundings (fp32, then float16): two
// instructions, with the product made opaque in between so that they are not merged into one
// mixed-precision instruction that rounds once.
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
// (by value: a bit cast straight from a vector's element reads the vector's first element)
__device__ __forceinline__ h2 as_h2(unsigned u) { return __builtin_bit_cast(h2, u); }
__device__ __forceinline__ unsigned as_bits(h2 v) { return __builtin_bit_cast(unsigned, v); }

template <int HALF>
struct LateAcc {
    PackH<kWide> g, s;
    __device__ __forceinline__ void start(const Raw<kWide>& gr) {
        g = widen<HALF, kWide>(gr);
#pragma unroll
        for (int e = 0; e < kWide; ++e) s.e[e] = 0.0f;
    }
    __device__ __forceinline__ void fold(const Raw<kWide>& xr, float w) {
        const PackH<kWide> x = widen<HALF, kWide>(xr);
#pragma unroll
        for (int e = 0; e < kWide; ++e) {
            float d = rnd<HALF>(x.e[e] - g.e[e]);
            float wd = rnd<HALF>(d * w);
            s.e[e] = rnd<HALF>(s.e[e] + wd);
        }
    }
    __device__ __forceinline__ u4 step() const {
        u4 t;
        t.x = narrow2<HALF>(s.e[0], s.e[1]);
        t.y = narrow2<HALF>(s.e[2], s.e[3]);
        t.z = narrow2<HALF>(s.e[4], s.e[5]);
        t.w = narrow2<HALF>(s.e[6], s.e[7]);
        return t;
    }
    __device__ __forceinline__ u4 applied() const {      // g + step
        float r[kWide];
#pragma unroll
        for (int e = 0; e < kWide; ++e) r[e] = rnd<HALF>(g.e[e] + s.e[e]);
        u4 t;
        t.x = narrow2<HALF>(r[0], r[1]);
        t.y = narrow2<HALF>(r[2], r[3]);
        t.z = narrow2<HALF>(r[4], r[5]);
        t.w = narrow2<HALF>(r[6], r[7]);
        return t;
    }
};

#ifndef FA_HALF_BF16_PACKED
#define FA_HALF_BF16_PACKED 1
#endif
#if FA_HALF_BF16_PACKED
// bfloat16: the same operations on PAIRS of elements, so that the three fp32 operations of a
// client's term are packed instructions (v_pk_add_f32 / v_pk_mul_f32: two elements each); every
// result is still rounded on its own (one conversion per element).
template <>
struct LateAcc<FA_HALF_BF16> {
    f2 g[4], s[4];
    static __device__ __forceinline__ f2 wide(unsigned u) {
        return f2{__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xFFFF0000u)};
    }
    static __device__ __forceinline__ f2 rnd2(f2 v) { return f2{rnd<FA_HALF_BF16>(v.x), rnd<FA_HALF_BF16>(v.y)}; }
    static __device__ __forceinline__ unsigned bits(f2 v) { return narrow2<FA_HALF_BF16>(v.x, v.y); }
    __device__ __forceinline__ void start(const Raw<kWide>& gr) {
        g[0] = wide(gr.v.x);
        g[1] = wide(gr.v.y);
        g[2] = wide(gr.v.z);
        g[3] = wide(gr.v.w);
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] = f2{0.0f, 0.0f};
    }
    __device__ __forceinline__ void fold(const Raw<kWide>& xr, float w) {
        const f2 x[4] = {wide(xr.v.x), wide(xr.v.y), wide(xr.v.z), wide(xr.v.w)};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f2 d = rnd2(x[q] - g[q]);
            const f2 wd = rnd2(d * w);
            s[q] = rnd2(s[q] + wd);
        }
    }
    __device__ __forceinline__ u4 step() const {
        u4 t;
        t.x = bits(s[0]);
        t.y = bits(s[1]);
        t.z = bits(s[2]);
        t.w = bits(s[3]);
        return t;
    }
    __device__ __forceinline__ u4 applied() const {
        u4 t;
        t.x = bits(rnd2(g[0] + s[0]));
        t.y = bits(rnd2(g[1] + s[1]));
        t.z = bits(rnd2(g[2] + s[2]));
        t.w = bits(rnd2(g[3] + s[3]));
        return t;
    }
};
#endif

#ifndef FA_HALF_F16_NATIVE
#define FA_HALF_F16_NATIVE 1
#endif
#if FA_HALF_F16_NATIVE
template <>
struct LateAcc<FA_HALF_F16> {
    h2 g[4], s[4];
    __device__ __forceinline__ void start(const Raw<kWide>& gr) {
        g[0] = as_h2(gr.v.x);
        g[1] = as_h2(gr.v.y);
        g[2] = as_h2(gr.v.z);
        g[3] = as_h2(gr.v.w);
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] = h2{static_cast<_Float16>(0.0f), static_cast<_Float16>(0.0f)};
    }
    __device__ __forceinline__ void fold(const Raw<kWide>& xr, float w) {
        const h2 x[4] = {as_h2(xr.v.x), as_h2(xr.v.y), as_h2(xr.v.z), as_h2(xr.v.w)};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const h2 d = x[q] - g[q];
            f2 pr = __builtin_convertvector(d, f2) * w;
            __asm__("" : "+v"(pr));
            s[q] = s[q] + __builtin_convertvector(pr, h2);
        }
    }
    __device__ __forceinline__ u4 step() const {
        u4 t;
        t.x = as_bits(s[0]);
        t.y = as_bits(s[1]);
        t.z = as_bits(s[2]);
        t.w = as_bits(s[3]);
        return t;
    }
    __device__ __forceinline__ u4 applied() const {
        u4 t;
        t.x = as_bits(g[0] + s[0]);
        t.y = as_bits(g[1] + s[1]);
        t.z = as_bits(g[2] + s[2]);
        t.w = as_bits(g[3] + s[3]);
        return t;
    }
};
#endif

template <int OPT, int HALF>
__global__ __launch_bounds__(kThreads, 1) void reduce_late_half_kernel(const ReduceHArgs a) {
    // (plain FedAvg has no epilogue to defer: its NEW g is what gets parked, and a phase ends with
    // the burst of those alone)
    constexpr int VPT = kLateVpt, U = FA_HALF_LATE_AHEAD, P = kLateTiles, VEC = kWide;
    constexpr bool kHasM = OPT >= FA_OPT_ADAM;
    __shared__ u4 stash[P][VPT][kThreads];
    const unsigned G = gridDim.x, n = static_cast<unsigned>(a.n);
    const unsigned per_round = G * kThreads * VPT;
    const unsigned rounds = (n + per_round - 1) / per_round;
    const unsigned slots = G * rounds;
    const unsigned tlen = (((n + slots - 1) / slots) + 63u) & ~63u;     // equal-length tiles, whole wave rows
    const unsigned total = (n + tlen - 1) / tlen;
    unsigned slot = blockIdx.x;
    if (G % 8 == 0) slot = (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8;  // an XCD's blocks take neighbouring tiles
    // chip-wide phases: the blocks meet before their bursts (timing only, bounded waits), the odd
    // and the even XCDs' blocks as two groups half a phase apart
    const unsigned most = (total + G - 1) / G;
    bool phased = a.sync_word != nullptr && most > static_cast<unsigned>(P);
    GridMeet meet = {a.sync_word, 0u, false, G};
    unsigned shift = 0, meetings = 0;
    if (phased && a.sync_groups == 2 && G % 8 == 0) {
        const unsigned group = (blockIdx.x % 8u) % 2u;
        meet.word = a.sync_word + 4u * group;
        meet.blocks = G / 2u;
        shift = group * P / 2u;
    }
    if (phased) {
        meetings = (most + shift + P - 1) / P;
        grid_meet_enter(meet, a.sync_epoch);
    }
    if (slot >= total) {
        if (phased) grid_meet_leave(meet, meetings);
        return;
    }
    const unsigned ntiles = (total - slot + G - 1) / G;
    const unsigned r0 = slot * tlen, tstride = G * tlen;
    auto index = [&](unsigned tile, int p, bool& live) -> unsigned {
        const unsigned off = p * kThreads + threadIdx.x;
        const unsigned i = r0 + tile * tstride + off;
        live = off < tlen && i < n;
        return live ? i : n - 1;      // idle lanes redo the last vector, never store
    };
    // The client loop runs in batches of U clients, double-buffered: the loads of batch b + 1 are
    // in flight while batch b is folded, and the first batch (and g) of the NEXT tile is requested
    // before this tile's step is parked and the epilogue runs -- at one wave per SIMD nothing
    // else would hide the memory latency at a tile's start.
    unsigned idx[VPT], idn[VPT];
    Raw<VEC> gr[VPT], xa[U][VPT], xb[U][VPT];
    auto request = [&](int k0, Raw<VEC> (&buf)[U][VPT], const unsigned (&ix)[VPT]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int p = 0; p < VPT; ++p) buf[u][p] = load_raw<VEC, true>(a.x[k0 + u], ix[p]);
    };
    const int nb = a.K / U;
#pragma unroll
    for (int p = 0; p < VPT; ++p) {
        bool live;
        idx[p] = index(0, p, live);
        gr[p] = load_raw<VEC, false>(a.g, idx[p]);
    }
    if (nb > 0) request(0, xa, idx);
    for (unsigned ct = 0; ct < ntiles; ++ct) {
        LateAcc<HALF> acc[VPT];
#pragma unroll
        for (int p = 0; p < VPT; ++p) acc[p].start(gr[p]);
        auto fold = [&](int k0, const Raw<VEC> (&buf)[U][VPT]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float w = a.w[k0 + u];
#pragma unroll
                for (int p = 0; p < VPT; ++p) acc[p].fold(buf[u][p], w);
            }
        };
        int b = 0;
#if FA_HALF_LATE_OVERLAP
#pragma unroll 1
        for (; b + 2 <= nb; b += 2) {
            request((b + 1) * U, xb, idx);
            fold(b * U, xa);
            if (b + 2 < nb) request((b + 2) * U, xa, idx);
            fold((b + 1) * U, xb);
        }
        if (b < nb) fold(b * U, xa);
#else
#pragma unroll 1
        for (; b < nb; ++b) {
            if (b > 0) request(b * U, xa, idx);
            fold(b * U, xa);
        }
#endif
        for (int k = nb * U; k < a.K; ++k) {
            const float w = a.w[k];
            Raw<VEC> xr[VPT];
#pragma unroll
            for (int p = 0; p < VPT; ++p) xr[p] = load_raw<VEC, true>(a.x[k], idx[p]);
#pragma unroll
            for (int p = 0; p < VPT; ++p) acc[p].fold(xr[p], w);
        }
        if (ct + 1 < ntiles) {
#pragma unroll
            for (int p = 0; p < VPT; ++p) {
                bool live;
                idn[p] = index(ct + 1, p, live);
                gr[p] = load_raw<VEC, false>(a.g, idn[p]);
            }
            if (nb > 0) request(0, xa, idn);
        }
        const int j = static_cast<int>((ct + shift) % P);
#pragma unroll
        for (int p = 0; p < VPT; ++p)      // every thread only touches its own slots: no barrier
            stash[j][p][threadIdx.x] = OPT == FA_OPT_NONE ? acc[p].applied() : acc[p].step();
        if (j == P - 1 || ct + 1 == ntiles) {
            const unsigned first = ct - static_cast<unsigned>(j);
            const int j_lo = (ct < static_cast<unsigned>(P) && static_cast<unsigned>(j) >= ct) ? j - static_cast<int>(ct) : 0;
            if (phased) grid_meet(meet);
            if constexpr (OPT == FA_OPT_NONE) {
                __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.g, 0, -1, 0x00020000);
#pragma unroll 1
                for (int jj = j_lo; jj <= j; ++jj) {
#pragma unroll
                    for (int p = 0; p < VPT; ++p) {
                        bool live;
                        const unsigned i = index(first + jj, p, live);
                        if (live) __builtin_amdgcn_raw_buffer_store_b128(stash[jj][p][threadIdx.x], rs, i * 16u, 0, 17);
                    }
                }
                if (phased) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                goto next_tile;
            }
            // the epilogue over the parked tiles: all of a tile's g / m / v loads are issued at
            // once, the next tile's fly while this tile's formulas are evaluated and stored
            Raw<VEC> gq[2][VPT], mq[2][VPT], vq[2][VPT];
            auto fetch = [&](int jj, int b) {
#pragma unroll
                for (int p = 0; p < VPT; ++p) {
                    bool live;
                    const unsigned i = index(first + jj, p, live);
                    gq[b][p] = load_raw<VEC, false>(a.g, i);
                    vq[b][p] = load_raw<VEC, true>(a.v, i);
                    if (kHasM) mq[b][p] = load_raw<VEC, true>(a.m, i);
                }
            };
            auto finish = [&](int jj, int b) {
#pragma unroll
                for (int p = 0; p < VPT; ++p) {
                    bool live;
                    const unsigned i = index(first + jj, p, live);
                    Raw<VEC> sr;
                    sr.v = stash[jj][p][threadIdx.x];
                    PackH<VEC> st = widen<HALF, VEC>(sr), gg = widen<HALF, VEC>(gq[b][p]), vv = widen<HALF, VEC>(vq[b][p]), mm;
                    if (kHasM) {
                        mm = widen<HALF, VEC>(mq[b][p]);
                    } else {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) mm.e[e] = 0.0f;
                    }
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const float step = optimizer_step_h<OPT, HALF>(st.e[e], mm.e[e], vv.e[e], a.h);
                        gg.e[e] = rnd<HALF>(gg.e[e] + step);
                    }
                    if (live) {
                        store_result_h<HALF, VEC>(a.v, i, vv);
                        if (kHasM) store_result_h<HALF, VEC>(a.m, i, mm);
                        store_result_h<HALF, VEC>(a.g, i, gg);
                    }
                }
            };
            fetch(j_lo, 0);
#pragma unroll 1
            for (int jj = j_lo; jj <= j; jj += 2) {
                if (jj + 1 <= j) fetch(jj + 1, 1);
                finish(jj, 0);
                if (jj + 1 <= j) {
                    if (jj + 2 <= j) fetch(jj + 2, 0);
                    finish(jj + 1, 1);
                }
            }
            if (phased) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the burst before the next phase's loads
        }
    next_tile:
#pragma unroll
        for (int p = 0; p < VPT; ++p) idx[p] = idn[p];
    }
    if (phased) grid_meet_leave(meet, meetings);
}

template <int OPT, int HALF>
__global__ __launch_bounds__(kThreads) void opt_step_half_kernel(void* step, void* m, void* v, HyperH h, size_t n) {
    const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
    for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += stride) {
        float vv = ld<HALF>(v, i);
        float mm = (OPT != FA_OPT_AVGM) ? ld<HALF>(m, i) : 0.0f;
        st<HALF>(step, i, optimizer_step_h<OPT, HALF>(ld<HALF>(step, i), mm, vv, h));
        st<HALF>(v, i, vv);
        if (OPT != FA_OPT_AVGM) st<HALF>(m, i, mm);
    }
}

template <int HALF>
__global__ __launch_bounds__(kThreads) void apply_half_kernel(void* g, const void* step, size_t n) {
    const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
    for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += stride)
        st<HALF>(g, i, rnd<HALF>(ld<HALF>(g, i) + ld<HALF>(step, i)));
}

struct MeanHArgs {
    const void* x[FA_KCHUNK];
    void* g;
    const void* sum_in;
    size_t n;
    int K, last;
    float Kall;
};

// fedavg_aggregator.py:285-295: param_sum.add_(model[name]) per client, then div_(K).
template <int HALF>
__global__ __launch_bounds__(kThreads) void mean_half_kernel(const MeanHArgs a) {
    const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
    for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < a.n; i += stride) {
        float s = a.sum_in ? ld<HALF>(a.sum_in, i) : 0.0f;
        for (int k = 0; k < a.K; ++k) s = rnd<HALF>(s + ld_nt<HALF>(a.x[k], i));
        st<HALF>(a.g, i, a.last ? rnd<HALF>(s / a.Kall) : s);
    }
}

// The step of one arrival (fedasync_aggregator.py:243-264).  MODE 0: model-based a * (x - g);
// 2: gradient-based a * (-x); and the two forms of the optimize_memory branch, which are ONE
// tensor operation each -- zeros.add_(d, alpha=a) and zeros.sub_(x, alpha=a) evaluate 0 + a*d and
// 0 - a*x in fp32 and round once -- MODE 3: model-based, 1: gradient-based.  They differ from
// the plain products in the sign of a zero: a product that underflows the 16-bit type keeps its
// sign through the one rounding, and 0 + (-0) is +0.  (PyTorch itself is of two minds here: the
// vectorised body of its CPU kernel does exactly this, the scalar tail that handles a tensor's
// last few elements rounds the product first and so returns +0 for every underflow.  The body's
// form is the one followed; the difference is the sign of a zero STEP and reaches g only where
// g is -0 itself.)
template <int MODE, int HALF>
__device__ __forceinline__ float async_step_h(float x, float g, float alpha) {
    if (MODE == 0 || MODE == 3) {
        float d = rnd<HALF>(x - g);
        float p = alpha * d;
        return rnd<HALF>(MODE == 3 ? 0.0f + p : p);
    }
    if (MODE == 1) {
        float p = alpha * x;
        return rnd<HALF>(0.0f - p);
    }
    float nx = -x;
    return rnd<HALF>(alpha * nx);
}

// One arrival over 16-bit buffers, VEC elements per thread.  BUFF = false is the FedAsync mix
// (g = g + step, `step` optional output); BUFF = true the FedBuff accumulate (the buffer `step`
// grows by the arrival's step; with flush != 0 also g = g + buffer).
template <int MODE, int HALF, int VEC, bool BUFF>
__global__ __launch_bounds__(kThreads) void arrival_half_kernel(const void* x, float alpha, void* g, void* step, int first, int flush, size_t n) {
    const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
    for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += stride) {
        const PackH<VEC> gv = load_h<HALF, VEC, false>(g, i);
        const PackH<VEC> xv = load_h<HALF, VEC, true>(x, i);
        PackH<VEC> s, r;
        if (BUFF && !first) {
            s = load_h<HALF, VEC, false>(step, i);
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) s.e[e] = 0.0f;
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float d = async_step_h<MODE, HALF>(xv.e[e], gv.e[e], alpha);
            s.e[e] = BUFF ? rnd<HALF>(s.e[e] + d) : d;
            r.e[e] = rnd<HALF>(gv.e[e] + s.e[e]);
        }
        if (BUFF || step) store_h<HALF, VEC>(step, i, s);
        if (!BUFF || flush) store_h<HALF, VEC>(g, i, r);
    }
}

// Weighted mixes of K client vectors that do not subtract g first (FedCompass / FedQueue,
// fedcompass_aggregator.py:152-338, fedqueue_aggregator.py:75-131): the six combine modes of the
// fp32 mix_kernel with every operation rounded.
struct MixHArgs {
    const void* x[FA_KCHUNK];
    float w[FA_KCHUNK];
    void* g;
    const void* acc_in;   // partial sum of earlier client chunks (NULL: 0)
    void* acc_out;        // where a non-final chunk leaves its partial sum
    size_t n;             // elements of width VEC
    int K, mode, last;
    float coeff;
};

template <int HALF, int VEC>
__global__ __launch_bounds__(kThreads) void mix_half_kernel(const MixHArgs a) {
    const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
    for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < a.n; i += stride) {
        PackH<VEC> g, acc;
        const bool reads_g = a.mode != FA_MIX_REPLACE && a.mode != FA_MIX_SUM_SET;
        if (reads_g) {
            g = load_h<HALF, VEC, false>(a.g, i);
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) g.e[e] = 0.0f;
        }
        if (a.mode == FA_MIX_SUBTRACT) {
            acc = g;
        } else if (a.acc_in) {
            acc = load_h<HALF, VEC, false>(a.acc_in, i);
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc.e[e] = (a.mode >= FA_MIX_CONVEX_ONE) ? -0.0f : 0.0f;   // -0 + y == y for every y
        }
        int k = 0;
        for (; k + kAhead <= a.K; k += kAhead) {
            PackH<VEC> x[kAhead];
#pragma unroll
            for (int u = 0; u < kAhead; ++u) x[u] = load_h<HALF, VEC, true>(a.x[k + u], i);
#pragma unroll
            for (int u = 0; u < kAhead; ++u)
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    float xw = rnd<HALF>(x[u].e[e] * a.w[k + u]);
                    acc.e[e] = rnd<HALF>((a.mode == FA_MIX_SUBTRACT) ? acc.e[e] - xw : acc.e[e] + xw);
                }
        }
        for (; k < a.K; ++k) {
            const PackH<VEC> x = load_h<HALF, VEC, true>(a.x[k], i);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float xw = rnd<HALF>(x.e[e] * a.w[k]);
                acc.e[e] = rnd<HALF>((a.mode == FA_MIX_SUBTRACT) ? acc.e[e] - xw : acc.e[e] + xw);
            }
        }
        if (!a.last && a.mode != FA_MIX_SUBTRACT) {
            store_h<HALF, VEC>(a.acc_out, i, acc);
            continue;
        }
        if ((a.mode == FA_MIX_CONVEX || a.mode == FA_MIX_CONVEX_ONE) && a.last) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float gt = rnd<HALF>(g.e[e] * a.coeff);
                acc.e[e] = rnd<HALF>(gt + acc.e[e]);
            }
        }
        if (a.mode == FA_MIX_SUM_ADD && a.last) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc.e[e] = rnd<HALF>(g.e[e] + acc.e[e]);
        }
        store_h<HALF, VEC>(a.g, i, acc);
    }
}

template <int HALF>
__global__ __launch_bounds__(kThreads) void accum_raw_half_kernel(const void* x, void* step, int first, size_t n) {
    const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
    for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += stride) {
        const float s = first ? 0.0f : ld<HALF>(step, i);
        st<HALF>(step, i, rnd<HALF>(s + ld<HALF>(x, i)));
    }
}

template <int HALF>
__global__ __launch_bounds__(kThreads) void flush_mean_half_kernel(void* g, const void* step, float kf, size_t n) {
    const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
    for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += stride)
        st<HALF>(g, i, rnd<HALF>(ld<HALF>(step, i) / kf));
}

template <int HALF>
__global__ __launch_bounds__(kThreads) void fill_half_kernel(void* p, float value, size_t n) {
    const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
    for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += stride) st<HALF>(p, i, value);
}

inline hipStream_t as_stream(fa_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? FA_OK : FA_ERR_HIP - static_cast<int>(e);
}
inline bool aligned2(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 1u) == 0; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline bool known(int half) { return half == FA_HALF_BF16 || half == FA_HALF_F16; }
inline unsigned grid_for(size_t n) {
    size_t b = (n + kThreads - 1) / kThreads;
    return static_cast<unsigned>(b < 1 ? 1 : (b > static_cast<size_t>(kMaxBlocks) ? kMaxBlocks : b));
}
inline const void* at(const void* p, size_t elements) { return p ? static_cast<const char*>(p) + 2 * elements : nullptr; }
inline void* at(void* p, size_t elements) { return p ? static_cast<char*>(p) + 2 * elements : nullptr; }

HyperH make_hyper_h(double lr, double beta1, double beta2, double tau, int half) {
    return HyperH{static_cast<float>(lr), static_cast<float>(beta1), static_cast<float>(1.0 - beta1),
                  static_cast<float>(beta2), static_cast<float>(1.0 - beta2),
                  rnd_host(static_cast<float>(tau), half)};
}

// Benchmark knob of the persistent kernel with the late epilogue, read at every launch:
// APPFL_AMD_HALF_LATE = 0 keeps every round on the tile-per-block kernel, 1 (the default) routes
// to the late-epilogue kernel where it applies, with the odd and the even XCDs' blocks meeting
// as two staggered groups, 2 the same without meetings, 3 with one chip-wide group; 4 forces the
// kernel at every client count and length it can run on.  Same bits whatever the value.
int late_mode() {
    const char* e = getenv("APPFL_AMD_HALF_LATE");
    return e && *e ? atoi(e) : 1;
}

// Where the late-epilogue kernel is ahead of the tile-per-block kernel (same process, kernel time,
// bfloat16, tools/half_late_sweep.py; `tiles` = tiles of 256 x 4 vectors per CU).  Two-moment
// optimizers: 8 M elements (3.8 tiles) x 48 / 64 / 100 clients -3.7 / -5.5 / -9.6 %, 25.6 M (12 tiles)
// x 32 / 48 / 64 / 100 -1.1 / -7.0 / -8.9 / -10.6 %, 47 M (22 tiles) x 24 / 32 / 64 -1.0 / -5.2 / -9.7 %,
// 162.5 M x 24 / 32 -4.1 / -5.4 %; behind at fewer clients (25.6 M x 24 / 16 / 8 +4 / +17 / +42 %:
// at one wave per SIMD a short client list leaves the epilogue nothing to hide behind) and on
// short vectors (4.5 M x 48 +8 %).  FedAvgM (two result streams): from 32 clients at 3.5 tiles
// (8 M x 32 / 64 -2.3 / -9.1 %), 24 at 12 (25.6 M -1.3 %), 16 at 40 (87 M -3.6 %).
// float16 (x - g and step + term as native packed instructions, see LateAcc; with every
// operation rounded through fp32 the kernel was 4-42 % BEHIND at every shape): later break-even
// than bfloat16, its tile-per-block kernel being a little faster -- two-moment optimizers 25.6 M x
// 48 / 64 / 100 -4.2 / -4.1 / -7.4 %, 47 M x 64 -8.5 %, 8-17 M x 64 -0.8...-4.2 %, but 17 M x 48 +4 %;
// FedAvgM 25.6 M x 48 / 64 -3.3 / -5.3 %, 47 M x 32 -3.8 %; plain FedAvg 8-87 M x 48-64 -3.4...-5.9 %.
bool late_wins(int half, int opt, int K, double tiles) {
    if (half != FA_HALF_BF16) {
        if (opt == FA_OPT_NONE) return (K >= 48 && tiles >= 3.5) || (K >= 32 && tiles >= 22.0);
        if (opt == FA_OPT_AVGM) return (K >= 64 && tiles >= 3.5) || (K >= 48 && tiles >= 8.0) || (K >= 32 && tiles >= 22.0);
        return (K >= 64 && tiles >= 3.5) || (K >= 48 && tiles >= 12.0);
    }
    // plain FedAvg (the new g parked, one result stream): 8 M elements x 24 / 32 / 64 / 100 clients
    // -2.4 / -3.6 / -8.2 / -8.1 %, 11.7 M x 100 -10.4 % (0.750 -> 0.837 of peak), 25.6 M x 24 / 64 -2.4 /
    // -7.9 %, 162.5 M x 16 / 32 -2.3 / -6.3 %; behind at 16 clients elsewhere (+0.5...+5 %) and below
    if (opt == FA_OPT_NONE) return (K >= 24 && tiles >= 3.5) || (K >= 16 && tiles >= 70.0);
    if (opt == FA_OPT_AVGM) return (K >= 32 && tiles >= 3.5) || (K >= 24 && tiles >= 12.0) || (K >= 16 && tiles >= 40.0);
    return (K >= 48 && tiles >= 3.5) || (K >= 32 && tiles >= 12.0) || (K >= 24 && tiles >= 22.0);
}

// The late-epilogue kernel takes the final form of a launch (step, epilogue, apply; no carry in,
// no step output) of an optimizer round.
template <int HALF>
bool launch_late_half(int opt, const ReduceHArgs& a0, hipStream_t st) {
    const int mode = late_mode();
    if (!mode || a0.no_persistent || !a0.last || !a0.apply || a0.step_in || a0.step_out) return false;
    const int G = fa_internal::cu_count();
    if (mode != 4 && !late_wins(HALF, opt, a0.K, static_cast<double>(a0.n) / (static_cast<double>(G) * kThreads * kLateVpt)))
        return false;
    ReduceHArgs a = a0;
    a.sync_word = mode == 2 ? nullptr : fa_internal::meeting_word(st, &a.sync_epoch);
    a.sync_groups = mode == 3 ? 1 : 2;
    const dim3 grid(G), block(kThreads);
    switch (opt) {
        case FA_OPT_NONE: hipLaunchKernelGGL((reduce_late_half_kernel<FA_OPT_NONE, HALF>), grid, block, 0, st, a); break;
        case FA_OPT_AVGM: hipLaunchKernelGGL((reduce_late_half_kernel<FA_OPT_AVGM, HALF>), grid, block, 0, st, a); break;
        case FA_OPT_ADAM: hipLaunchKernelGGL((reduce_late_half_kernel<FA_OPT_ADAM, HALF>), grid, block, 0, st, a); break;
        case FA_OPT_YOGI: hipLaunchKernelGGL((reduce_late_half_kernel<FA_OPT_YOGI, HALF>), grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL((reduce_late_half_kernel<FA_OPT_ADAGRAD, HALF>), grid, block, 0, st, a); break;
    }
    return true;
}

template <int HALF, int VEC>
void launch_reduce_half(int opt, const ReduceHArgs& a, hipStream_t st) {
    if constexpr (VEC == kWide) {
        if (launch_late_half<HALF>(opt, a, st)) return;
    }
    constexpr int VPT = VEC == kWide ? FA_HALF_VPT : 1;
    constexpr int U = VEC == kWide ? FA_HALF_AHEAD : kAhead;
    const size_t tiles = (a.n + static_cast<size_t>(kThreads) * VPT - 1) / (static_cast<size_t>(kThreads) * VPT);
#ifdef FA_HALF_GRID_CAP       // A/B knob: a resident grid that walks the tiles grid-stride (256 / 512 / 1024 / 2048 /
                              // 4096 blocks: 0.61 / 0.71 / 0.69 / 0.70 / 0.71 of peak against 0.725 with one tile per block)
    const size_t cap = FA_HALF_GRID_CAP;
#else
    const size_t cap = kMaxBlocks;
#endif
    const dim3 grid(static_cast<unsigned>(tiles > cap ? cap : tiles)), block(kThreads);
    switch (opt) {
        case FA_OPT_NONE: hipLaunchKernelGGL((reduce_half_kernel<FA_OPT_NONE, HALF, VEC, VPT, U>), grid, block, 0, st, a); break;
        case FA_OPT_AVGM: hipLaunchKernelGGL((reduce_half_kernel<FA_OPT_AVGM, HALF, VEC, VPT, U>), grid, block, 0, st, a); break;
        case FA_OPT_ADAM: hipLaunchKernelGGL((reduce_half_kernel<FA_OPT_ADAM, HALF, VEC, VPT, U>), grid, block, 0, st, a); break;
        case FA_OPT_YOGI: hipLaunchKernelGGL((reduce_half_kernel<FA_OPT_YOGI, HALF, VEC, VPT, U>), grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL((reduce_half_kernel<FA_OPT_ADAGRAD, HALF, VEC, VPT, U>), grid, block, 0, st, a); break;
    }
}

// `k_done` > 0: the partial step of clients 0..k_done-1 is in the carry buffer (step_out, else
// step_ws) already; the loop over the clients is taken up from there (same order, same bits).
int reduce_half(const void* const* x, const double* w, int K, void* g, const void* g_ro, void* m, void* v,
                size_t n, int opt, const HyperH& h, void* step_out, void* step_ws, int apply, int half,
                hipStream_t st, int k_done = 0, unsigned flags = 0) {
    if (!x || !w || K <= 0 || !known(half)) return FA_ERR_INVALID;
    if (flags & ~static_cast<unsigned>(FA_ROUTE_NO_PERSISTENT)) return FA_ERR_INVALID;
    if (k_done < 0 || k_done >= K) return FA_ERR_INVALID;
    if (k_done > 0 && !step_out && !step_ws) return FA_ERR_WORKSPACE;
    if (apply ? !g : (!g_ro || !step_out)) return FA_ERR_INVALID;
    if (opt < FA_OPT_NONE || opt > FA_OPT_ADAGRAD) return FA_ERR_INVALID;
    if (opt != FA_OPT_NONE && !v) return FA_ERR_INVALID;
    if (opt >= FA_OPT_ADAM && !m) return FA_ERR_INVALID;
    if (K - k_done > FA_KCHUNK && !step_out && !step_ws) return FA_ERR_WORKSPACE;
    const void* ptrs[] = {g, g_ro, m, v, step_out, step_ws};
    bool wide = true;
    for (const void* p : ptrs) {
        if (p && !aligned2(p)) return FA_ERR_ALIGN;
        wide = wide && (!p || aligned16(p));
    }
    for (int k = 0; k < K; ++k) {
        if (!x[k]) return FA_ERR_INVALID;
        if (!aligned2(x[k])) return FA_ERR_ALIGN;
        wide = wide && aligned16(x[k]);
    }
    if (n == 0) return FA_OK;
    if (n > kSegmentH) {
        // the write-through stores address a buffer through a 32-bit byte offset: one launch
        // covers at most kSegmentH elements (2 GiB) of every buffer
        std::vector<const void*> xs(static_cast<size_t>(K));
        for (size_t off = 0; off < n; off += kSegmentH) {
            for (int k = 0; k < K; ++k) xs[static_cast<size_t>(k)] = at(x[k], off);
            int rc = reduce_half(xs.data(), w, K, at(g, off), at(g_ro, off), at(m, off), at(v, off),
                                 n - off < kSegmentH ? n - off : kSegmentH, opt, h, at(step_out, off),
                                 at(step_ws, off), apply, half, st, k_done, flags);
            if (rc != FA_OK) return rc;
        }
        return FA_OK;
    }
    const int chunks = (K - k_done + FA_KCHUNK - 1) / FA_KCHUNK;
    void* carry = step_out ? step_out : step_ws;
    // 128-bit body over the leading whole vectors when every buffer allows it, the 16-bit-access
    // kernel over the rest
    const size_t body = wide ? n / kWide * kWide : 0;
    for (int ch = 0; ch < chunks; ++ch) {
        ReduceHArgs a;
        memset(&a, 0, sizeof(a));
        const int k0 = k_done + ch * FA_KCHUNK;
        a.K = (K - k0 < FA_KCHUNK) ? K - k0 : FA_KCHUNK;
        a.last = (ch == chunks - 1);
        a.apply = apply;
        a.h = h;
        a.no_persistent = (flags & FA_ROUTE_NO_PERSISTENT) != 0;
        const int o = a.last ? opt : FA_OPT_NONE;
        for (int part = 0; part < 2; ++part) {
            const size_t lo = part ? body : 0, cnt = part ? n - body : body;
            if (cnt == 0) continue;
            for (int k = 0; k < a.K; ++k) {
                a.x[k] = at(x[k0 + k], lo);
                a.w[k] = static_cast<float>(w[k0 + k]);
            }
            a.g = at(g, lo);
            a.g_ro = at(g_ro, lo);
            a.m = at(m, lo);
            a.v = at(v, lo);
            a.step_in = (ch == 0 && k_done == 0) ? nullptr : at(static_cast<const void*>(carry), lo);
            a.step_out = at(a.last ? step_out : carry, lo);
            if (part == 0) {
                a.n = cnt / kWide;
                if (half == FA_HALF_BF16) launch_reduce_half<FA_HALF_BF16, kWide>(o, a, st); else launch_reduce_half<FA_HALF_F16, kWide>(o, a, st);
            } else {
                a.n = cnt;
                if (half == FA_HALF_BF16) launch_reduce_half<FA_HALF_BF16, 1>(o, a, st); else launch_reduce_half<FA_HALF_F16, 1>(o, a, st);
            }
            int rc = launch_status();
            if (rc != FA_OK) return rc;
        }
    }
    return FA_OK;
}

template <int MODE, int HALF, bool BUFF>
void launch_arrival(const void* x, float alpha, void* g, void* step, int first, int flush, size_t n, hipStream_t st) {
    const bool wide = aligned16(x) && aligned16(g) && (!step || aligned16(step));
    const size_t body = wide ? n / kWide * kWide : 0;
    if (body)
        hipLaunchKernelGGL((arrival_half_kernel<MODE, HALF, kWide, BUFF>), dim3(grid_for(body / kWide)), dim3(kThreads), 0,
                           st, x, alpha, g, step, first, flush, body / kWide);
    if (n > body)
        hipLaunchKernelGGL((arrival_half_kernel<MODE, HALF, 1, BUFF>), dim3(grid_for(n - body)), dim3(kThreads), 0, st,
                           at(x, body), alpha, at(g, body), at(step, body), first, flush, n - body);
}

template <int HALF, bool BUFF>
int arrival_half(const void* x, float alpha, int gradient_based, int sub_form, void* g, void* step, int first,
                 int flush, size_t n, hipStream_t st) {
    if (!gradient_based && sub_form)
        launch_arrival<3, HALF, BUFF>(x, alpha, g, step, first, flush, n, st);
    else if (!gradient_based)
        launch_arrival<0, HALF, BUFF>(x, alpha, g, step, first, flush, n, st);
    else if (sub_form)
        launch_arrival<1, HALF, BUFF>(x, alpha, g, step, first, flush, n, st);
    else
        launch_arrival<2, HALF, BUFF>(x, alpha, g, step, first, flush, n, st);
    return launch_status();
}

template <int HALF>
int opt_step_half(void* step, void* m, void* v, size_t n, int opt, const HyperH& h, hipStream_t st) {
    const dim3 grid(grid_for(n)), block(kThreads);
    switch (opt) {
        case FA_OPT_AVGM: hipLaunchKernelGGL((opt_step_half_kernel<FA_OPT_AVGM, HALF>), grid, block, 0, st, step, m, v, h, n); break;
        case FA_OPT_ADAM: hipLaunchKernelGGL((opt_step_half_kernel<FA_OPT_ADAM, HALF>), grid, block, 0, st, step, m, v, h, n); break;
        case FA_OPT_YOGI: hipLaunchKernelGGL((opt_step_half_kernel<FA_OPT_YOGI, HALF>), grid, block, 0, st, step, m, v, h, n); break;
        default: hipLaunchKernelGGL((opt_step_half_kernel<FA_OPT_ADAGRAD, HALF>), grid, block, 0, st, step, m, v, h, n); break;
    }
    return launch_status();
}

}  // namespace

extern "C" {

const char* fa_half_reduce_kernel_name(int K, size_t n, int opt, int half) {
    if (K <= 0 || K > FA_KCHUNK || !late_mode()) return "reduce_half_kernel";
    const double tiles = static_cast<double>(n / kWide) / (static_cast<double>(fa_internal::cu_count()) * kThreads * kLateVpt);
    return (late_mode() == 4 || late_wins(half, opt, K, tiles)) ? "reduce_late_half_kernel" : "reduce_half_kernel";
}

float fa_half_round_scalar(double x, int half) {
    return known(half) ? rnd_host(static_cast<float>(x), half) : static_cast<float>(x);
}

int fa_half_reduce_opt(const void* const* x, const double* w, int K, void* g, void* m, void* v, size_t n,
                       int opt, double lr, double beta1, double beta2, double tau, void* step_out,
                       void* step_ws, int half, fa_stream_t stream) {
    return reduce_half(x, w, K, g, nullptr, m, v, n, opt, make_hyper_h(lr, beta1, beta2, tau, half), step_out,
                       step_ws, 1, half, as_stream(stream));
}

int fa_half_reduce_opt_ex(const void* const* x, const double* w, int K, void* g, void* m, void* v, size_t n,
                          int opt, double lr, double beta1, double beta2, double tau, void* step_out,
                          void* step_ws, int half, unsigned flags, fa_stream_t stream) {
    return reduce_half(x, w, K, g, nullptr, m, v, n, opt, make_hyper_h(lr, beta1, beta2, tau, half), step_out,
                       step_ws, 1, half, as_stream(stream), 0, flags);
}

int fa_half_reduce_step(const void* const* x, const double* w, int K, const void* g, void* step_out, size_t n,
                        int half, fa_stream_t stream) {
    return reduce_half(x, w, K, nullptr, g, nullptr, nullptr, n, FA_OPT_NONE, make_hyper_h(0, 0, 0, 0, half),
                       step_out, nullptr, 0, half, as_stream(stream));
}

int fa_half_reduce_opt_resume(const void* const* x, const double* w, int K, int k_done, void* g, void* m,
                              void* v, size_t n, int opt, double lr, double beta1, double beta2, double tau,
                              void* step_ws, int half, fa_stream_t stream) {
    return reduce_half(x, w, K, g, nullptr, m, v, n, opt, make_hyper_h(lr, beta1, beta2, tau, half), nullptr,
                       step_ws, 1, half, as_stream(stream), k_done);
}

int fa_half_reduce_step_resume(const void* const* x, const double* w, int K, int k_done, const void* g,
                               void* step, size_t n, int half, fa_stream_t stream) {
    return reduce_half(x, w, K, nullptr, g, nullptr, nullptr, n, FA_OPT_NONE, make_hyper_h(0, 0, 0, 0, half),
                       step, nullptr, 0, half, as_stream(stream), k_done);
}

int fa_half_server_opt_step(void* step, void* m, void* v, size_t n, int opt, double lr, double beta1,
                            double beta2, double tau, int half, fa_stream_t stream) {
    if (!known(half)) return FA_ERR_INVALID;
    if (opt == FA_OPT_NONE) return FA_OK;
    if (opt < FA_OPT_NONE || opt > FA_OPT_ADAGRAD) return FA_ERR_INVALID;
    if (!step || !v || (opt >= FA_OPT_ADAM && !m)) return FA_ERR_INVALID;
    if (!aligned2(step) || !aligned2(v) || (m && !aligned2(m))) return FA_ERR_ALIGN;
    if (n == 0) return FA_OK;
    const HyperH h = make_hyper_h(lr, beta1, beta2, tau, half);
    return half == FA_HALF_BF16 ? opt_step_half<FA_HALF_BF16>(step, m, v, n, opt, h, as_stream(stream))
                                : opt_step_half<FA_HALF_F16>(step, m, v, n, opt, h, as_stream(stream));
}

int fa_half_apply_step(void* g, const void* step, size_t n, int half, fa_stream_t stream) {
    if (!g || !step || !known(half)) return FA_ERR_INVALID;
    if (!aligned2(g) || !aligned2(step)) return FA_ERR_ALIGN;
    if (n == 0) return FA_OK;
    const dim3 grid(grid_for(n)), block(kThreads);
    if (half == FA_HALF_BF16)
        hipLaunchKernelGGL((apply_half_kernel<FA_HALF_BF16>), grid, block, 0, as_stream(stream), g, step, n);
    else
        hipLaunchKernelGGL((apply_half_kernel<FA_HALF_F16>), grid, block, 0, as_stream(stream), g, step, n);
    return launch_status();
}

int fa_half_mean(const void* const* x, int K, void* g, size_t n, int half, fa_stream_t stream) {
    if (!x || !g || K <= 0 || !known(half)) return FA_ERR_INVALID;
    if (!aligned2(g)) return FA_ERR_ALIGN;
    for (int k = 0; k < K; ++k) {
        if (!x[k]) return FA_ERR_INVALID;
        if (!aligned2(x[k])) return FA_ERR_ALIGN;
    }
    if (n == 0) return FA_OK;
    const int chunks = (K + FA_KCHUNK - 1) / FA_KCHUNK;
    for (int ch = 0; ch < chunks; ++ch) {
        MeanHArgs a;
        memset(&a, 0, sizeof(a));
        const int k0 = ch * FA_KCHUNK;
        a.K = (K - k0 < FA_KCHUNK) ? K - k0 : FA_KCHUNK;
        for (int k = 0; k < a.K; ++k) a.x[k] = x[k0 + k];
        a.g = g;
        a.sum_in = (ch == 0) ? nullptr : g;
        a.n = n;
        a.last = (ch == chunks - 1);
        a.Kall = static_cast<float>(K);
        const dim3 grid(grid_for(n)), block(kThreads);
        if (half == FA_HALF_BF16)
            hipLaunchKernelGGL((mean_half_kernel<FA_HALF_BF16>), grid, block, 0, as_stream(stream), a);
        else
            hipLaunchKernelGGL((mean_half_kernel<FA_HALF_F16>), grid, block, 0, as_stream(stream), a);
        int rc = launch_status();
        if (rc != FA_OK) return rc;
    }
    return FA_OK;
}

int fa_half_weighted_mix(const void* const* x, const double* w, int K, void* g, double coeff, int mode,
                         void* acc_ws, size_t n, int half, fa_stream_t stream) {
    if (!x || !w || !g || K <= 0 || !known(half)) return FA_ERR_INVALID;
    if (mode < FA_MIX_REPLACE || mode > FA_MIX_SUM_SET) return FA_ERR_INVALID;
    if (mode == FA_MIX_CONVEX_ONE && K != 1) return FA_ERR_INVALID;
    if (K > FA_KCHUNK && mode != FA_MIX_SUBTRACT && !acc_ws) return FA_ERR_WORKSPACE;
    if (!aligned2(g) || (acc_ws && !aligned2(acc_ws))) return FA_ERR_ALIGN;
    bool wide = aligned16(g) && (!acc_ws || aligned16(acc_ws));
    for (int k = 0; k < K; ++k) {
        if (!x[k]) return FA_ERR_INVALID;
        if (!aligned2(x[k])) return FA_ERR_ALIGN;
        wide = wide && aligned16(x[k]);
    }
    if (n == 0) return FA_OK;
    const size_t body = wide ? n / kWide * kWide : 0;
    const int chunks = (K + FA_KCHUNK - 1) / FA_KCHUNK;
    for (int ch = 0; ch < chunks; ++ch) {
        MixHArgs a;
        memset(&a, 0, sizeof(a));
        const int k0 = ch * FA_KCHUNK;
        a.K = (K - k0 < FA_KCHUNK) ? K - k0 : FA_KCHUNK;
        a.mode = mode;
        a.last = (ch == chunks - 1);
        a.coeff = static_cast<float>(coeff);
        for (int part = 0; part < 2; ++part) {
            const size_t lo = part ? body : 0, cnt = part ? n - body : body;
            if (cnt == 0) continue;
            for (int k = 0; k < a.K; ++k) {
                a.x[k] = at(x[k0 + k], lo);
                a.w[k] = static_cast<float>(w[k0 + k]);
            }
            a.g = at(g, lo);
            a.acc_in = (ch == 0 || mode == FA_MIX_SUBTRACT) ? nullptr : at(static_cast<const void*>(acc_ws), lo);
            a.acc_out = at(acc_ws, lo);
            a.n = part ? cnt : cnt / kWide;
            const dim3 grid(grid_for(a.n)), block(kThreads);
            if (part == 0) {
                if (half == FA_HALF_BF16)
                    hipLaunchKernelGGL((mix_half_kernel<FA_HALF_BF16, kWide>), grid, block, 0, as_stream(stream), a);
                else
                    hipLaunchKernelGGL((mix_half_kernel<FA_HALF_F16, kWide>), grid, block, 0, as_stream(stream), a);
            } else {
                if (half == FA_HALF_BF16)
                    hipLaunchKernelGGL((mix_half_kernel<FA_HALF_BF16, 1>), grid, block, 0, as_stream(stream), a);
                else
                    hipLaunchKernelGGL((mix_half_kernel<FA_HALF_F16, 1>), grid, block, 0, as_stream(stream), a);
            }
            int rc = launch_status();
            if (rc != FA_OK) return rc;
        }
    }
    return FA_OK;
}

int fa_half_async_mix(const void* x, double alpha_t, int gradient_based, int sub_form, void* g, void* step_out,
                      size_t n, int half, fa_stream_t stream) {
    if (!x || !g || !known(half)) return FA_ERR_INVALID;
    if (!aligned2(x) || !aligned2(g) || (step_out && !aligned2(step_out))) return FA_ERR_ALIGN;
    if (n == 0) return FA_OK;
    // sub_form = the reference's optimize_memory branch, which passes alpha_t as `alpha=` of
    // add_ / sub_ (fedasync_aggregator.py:246-258): PyTorch casts that scalar to the tensor's type
    float alpha = static_cast<float>(alpha_t);
    if (sub_form) alpha = rnd_host(alpha, half);
    return half == FA_HALF_BF16
               ? arrival_half<FA_HALF_BF16, false>(x, alpha, gradient_based, sub_form, g, step_out, 0, 0, n, as_stream(stream))
               : arrival_half<FA_HALF_F16, false>(x, alpha, gradient_based, sub_form, g, step_out, 0, 0, n, as_stream(stream));
}

int fa_half_buff_accum(const void* x, double alpha_t, int gradient_based, void* g, void* step, int first,
                       int flush, size_t n, int half, fa_stream_t stream) {
    if (!x || !g || !step || !known(half)) return FA_ERR_INVALID;
    if (!aligned2(x) || !aligned2(g) || !aligned2(step)) return FA_ERR_ALIGN;
    if (n == 0) return FA_OK;
    const float alpha = static_cast<float>(alpha_t);
    return half == FA_HALF_BF16
               ? arrival_half<FA_HALF_BF16, true>(x, alpha, gradient_based, 0, g, step, first, flush, n, as_stream(stream))
               : arrival_half<FA_HALF_F16, true>(x, alpha, gradient_based, 0, g, step, first, flush, n, as_stream(stream));
}

int fa_half_buff_accum_raw(const void* x, void* step, int first, size_t n, int half, fa_stream_t stream) {
    if (!x || !step || !known(half)) return FA_ERR_INVALID;
    if (!aligned2(x) || !aligned2(step)) return FA_ERR_ALIGN;
    if (n == 0) return FA_OK;
    const dim3 grid(grid_for(n)), block(kThreads);
    if (half == FA_HALF_BF16)
        hipLaunchKernelGGL((accum_raw_half_kernel<FA_HALF_BF16>), grid, block, 0, as_stream(stream), x, step, first, n);
    else
        hipLaunchKernelGGL((accum_raw_half_kernel<FA_HALF_F16>), grid, block, 0, as_stream(stream), x, step, first, n);
    return launch_status();
}

int fa_half_buff_flush_mean(void* g, const void* step, int K, size_t n, int half, fa_stream_t stream) {
    if (!g || !step || K <= 0 || !known(half)) return FA_ERR_INVALID;
    if (!aligned2(g) || !aligned2(step)) return FA_ERR_ALIGN;
    if (n == 0) return FA_OK;
    const dim3 grid(grid_for(n)), block(kThreads);
    if (half == FA_HALF_BF16)
        hipLaunchKernelGGL((flush_mean_half_kernel<FA_HALF_BF16>), grid, block, 0, as_stream(stream), g, step,
                           static_cast<float>(K), n);
    else
        hipLaunchKernelGGL((flush_mean_half_kernel<FA_HALF_F16>), grid, block, 0, as_stream(stream), g, step,
                           static_cast<float>(K), n);
    return launch_status();
}

int fa_half_fill(void* p, double value, size_t n, int half, fa_stream_t stream) {
    if (!p || !known(half)) return FA_ERR_INVALID;
    if (!aligned2(p)) return FA_ERR_ALIGN;
    if (n == 0) return FA_OK;
    const float v = rnd_host(static_cast<float>(value), half);
    const dim3 grid(grid_for(n)), block(kThreads);
    if (half == FA_HALF_BF16)
        hipLaunchKernelGGL((fill_half_kernel<FA_HALF_BF16>), grid, block, 0, as_stream(stream), p, v, n);
    else
        hipLaunchKernelGGL((fill_half_kernel<FA_HALF_F16>), grid, block, 0, as_stream(stream), p, v, n);
    return launch_status();
}

}  // extern "C"
