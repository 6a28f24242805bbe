// fedagg.hip -- gfx950 kernels and C ABI for the FedAvg-family aggregation hot path.
//
// All kernels are HBM-bound streaming kernels: every thread owns whole 16-byte vectors of the
// flat parameter axis, walks the K client buffers in client order for its vectors and keeps
// the running step in registers, so each client byte is read once and g / m / v are read and
// written once per launch.  Client pointers and weights travel in the kernel-argument block
// and are fetched by scalar loads (wave-uniform), which leaves one `global_load_dwordx4
// v, v_off, s[base]` per client per vector in the inner loop.
//
// Built with -ffp-contract=off: the formulas are sequences of separately rounded fp32 ops.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/appfl_amd.h"
#include "internal.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
// Grid cap; larger inputs are walked grid-stride.  It covers a whole launch segment (2^29
// elements) at one 4-KiB tile per block: with a cap of 2^16 a resident block of a long vector
// hopped between stretches 256 MiB apart in every row, which cost 5-7 % at 2-8 clients on
// 162.5 M elements (fraction of peak 0.745 / 0.705 / 0.674 -> 0.78 / 0.75 / 0.72).
constexpr int kMaxBlocks = 1 << 20;

struct Hyper {
    float lr, b1, omb1, b2, omb2, tau;
};

// Plain-mean entries folded into a reduce launch (fa_round): up to three segments of the client
// rows, addressed relative to x[k].  A few hundred to a few thousand elements (BatchNorm
// statistics, counters): a launch of their own costs more than the work.
constexpr int kTailSegments = 3;
struct MeanTail {
    void* out[kTailSegments];
    long long off[kTailSegments];   // bytes from x[k] to the segment in client k's row
    unsigned n[kTailSegments];
    int dtype[kTailSegments];       // FA_DT_*
    unsigned total;                 // elements over all segments (0: nothing folded)
    unsigned blocks;                // tile-per-block kernel: leading blocks of the grid that do this work
    float Kall;                     // client count as fp32
};

struct ReduceArgs {
    const float* x[FA_KCHUNK];
    float w[FA_KCHUNK];
    float* g;
    float* m;
    float* v;
    const float* step_in;  // partial step of earlier client chunks (NULL: start from 0)
    float* step_out;       // where the partial / final step goes (NULL: nowhere)
    const float* g_ro;     // read-only g for step-only mode
    unsigned n;            // elements of width VEC
    int K;
    int last;   // 1: run the optimizer epilogue and the apply
    int apply;  // 1: write g (0 = step-only mode)
    int wt;     // 1: results leave with write-through (sc0 sc1) stores
    int il;     // persistent kernel: 1 = tiles dealt round-robin to the blocks, 0 = contiguous ranges
    int sync;   // persistent kernel, chip-wide write phases: bit 1 = the blocks meet before their
                // result bursts, bit 0 = and again after the bursts have drained; 0 = no phases;
                // bits 2-3 = groups of XCDs with staggered phases (0: one group)
    int phase_tiles;       // persistent kernel: tiles per phase (one burst); 0 = all the chip can park
    unsigned* sync_word;   // counter the blocks of this launch meet on (sync != 0)
    unsigned sync_epoch;   // this launch's tag in the counter's high bits: the first block to find another tag restarts the count
    Hyper h;
    MeanTail mt;
};

template <int VEC>
struct Pack {
    float e[VEC];
};

// Client bytes are read exactly once: `nt` loads keep them from displacing g / m / v in L2.
template <int VEC, bool NT = true>
__device__ __forceinline__ Pack<VEC> load_stream(const float* p, unsigned i) {
    Pack<VEC> r;
    if constexpr (VEC == 4) {
        const f4* q = reinterpret_cast<const f4*>(p) + i;
        f4 t = NT ? __builtin_nontemporal_load(q) : *q;
        r.e[0] = t.x; r.e[1] = t.y; r.e[2] = t.z; r.e[3] = t.w;
    } else {
        r.e[0] = NT ? __builtin_nontemporal_load(p + i) : p[i];
    }
    return r;
}

template <int VEC>
__device__ __forceinline__ Pack<VEC> load_state(const float* p, unsigned i);
template <>
__device__ __forceinline__ Pack<4> load_state<4>(const float* p, unsigned i) {
    f4 t = reinterpret_cast<const f4*>(p)[i];
    Pack<4> r;
    r.e[0] = t.x; r.e[1] = t.y; r.e[2] = t.z; r.e[3] = t.w;
    return r;
}
template <>
__device__ __forceinline__ Pack<1> load_state<1>(const float* p, unsigned i) {
    Pack<1> r;
    r.e[0] = p[i];
    return r;
}

template <int VEC>
__device__ __forceinline__ void store_state(float* p, unsigned i, const Pack<VEC>& r);
template <>
__device__ __forceinline__ void store_state<4>(float* p, unsigned i, const Pack<4>& r) {
    f4 t;
    t.x = r.e[0]; t.y = r.e[1]; t.z = r.e[2]; t.w = r.e[3];
    reinterpret_cast<f4*>(p)[i] = t;
}
template <>
__device__ __forceinline__ void store_state<1>(float* p, unsigned i, const Pack<1>& r) {
    p[i] = r.e[0];
}

typedef unsigned u4 __attribute__((ext_vector_type(4)));

// Result store of the 128-bit kernels: optionally write-through (sc0 sc1), which keeps the
// result lines from lingering dirty in L2 and trickling out between the streaming reads.
__device__ __forceinline__ void store_result(float* p, unsigned i, const Pack<4>& r, int wt) {
    if (wt) {
        f4 t;
        t.x = r.e[0]; t.y = r.e[1]; t.z = r.e[2]; t.w = r.e[3];
        // 4 GiB window: a launch addresses at most kSegment floats = 2 GiB from p
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p, 0, -1, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, t), rs, i * 16u, 0, 17);
    } else {
        store_state<4>(p, i, r);
    }
}
__device__ __forceinline__ void store_result(float* p, unsigned i, const Pack<1>& r, int) {
    store_state<1>(p, i, r);
}

__device__ __forceinline__ float sign_f32(float d) {
    // torch.sign: (0 < d) - (d < 0); NaN and both zeros give 0.
    return static_cast<float>(static_cast<int>(0.0f < d) - static_cast<int>(d < 0.0f));
}

// Turns the averaged pseudo-gradient into the server-optimizer step; updates m / v in place.
template <int OPT>
__device__ __forceinline__ float optimizer_step(float step, float& m, float& v, const Hyper& h) {
    if (OPT == FA_OPT_NONE) return step;
    if (OPT == FA_OPT_AVGM) {
        float mt = v * h.b1;
        v = mt + step;
        return v;
    }
    float mt = m * h.b1;
    float st = step * h.omb1;
    m = mt + st;
    float s2 = step * step;
    if (OPT == FA_OPT_ADAM) {
        float vt = v * h.b2;
        float sq = s2 * h.omb2;
        v = vt + sq;
    } else if (OPT == FA_OPT_YOGI) {
        float sg = sign_f32(v - s2);
        float yu = s2 * sg;
        float ys = yu * h.omb2;
        v = v - ys;
    } else {  // FA_OPT_ADAGRAD
        v = v + s2;
    }
    float num = h.lr * m;
    // sqrtf and / lower to the correctly rounded sequences (hipcc's default
    // -fhip-fp32-correctly-rounded-divide-sqrt); the __fsqrt_rn intrinsic is the 1-ulp v_sqrt_f32.
    float den = sqrtf(v) + h.tau;
    return num / den;
}

// Plain mean: the sum in client order, then torch.div by the client count (integers: fp32
// true division, truncated back).
__device__ __forceinline__ float mean_finish(float s, float kf) { return s / kf; }
__device__ __forceinline__ int64_t mean_finish(int64_t s, float kf) {
    return static_cast<int64_t>(static_cast<float>(s) / kf);
}
__device__ __forceinline__ int32_t mean_finish(int32_t s, float kf) {
    return static_cast<int32_t>(static_cast<float>(s) / kf);
}

// Folded plain-mean entries.  A handful of elements (the usual case: num_batches_tracked
// counters) go one per WAVE: lane k fetches client k's value, so one memory latency covers all
// clients, and the values are then added in client order through lane reads.  More elements go
// one per thread like mean_kernel, a few clients in flight at a time (a deeper batch would cost
// the host kernels registers they need elsewhere).
constexpr int kTailBatch = 8;
constexpr unsigned kTailByWave = 64;   // up to this many elements: one wave each
constexpr unsigned kWavesPerBlock = kThreads / 64;

template <typename T>
__device__ __forceinline__ const T* tail_src(const ReduceArgs& a, int k, long long off) {
    return reinterpret_cast<const T*>(reinterpret_cast<const char*>(a.x[k]) + off);
}

template <typename T>
__device__ __forceinline__ void mean_tail_element(const ReduceArgs& a, long long off, void* out, unsigned i) {
    T s = static_cast<T>(0);
    int k = 0;
    for (; k + kTailBatch <= a.K; k += kTailBatch) {
        T x[kTailBatch];
#pragma unroll
        for (int u = 0; u < kTailBatch; ++u) x[u] = __builtin_nontemporal_load(tail_src<T>(a, k + u, off) + i);
#pragma unroll
        for (int u = 0; u < kTailBatch; ++u) s = s + x[u];
    }
    for (; k < a.K; ++k) s = s + __builtin_nontemporal_load(tail_src<T>(a, k, off) + i);
    static_cast<T*>(out)[i] = mean_finish(s, a.mt.Kall);
}

template <typename T>
__device__ __forceinline__ void mean_tail_wave(const ReduceArgs& a, long long off, void* out, unsigned i) {
    static_assert(FA_KCHUNK <= 128, "two values per lane cover a launch's clients");
    const int lane = threadIdx.x & 63;
    T x0 = static_cast<T>(0), x1 = static_cast<T>(0);
    if (lane < a.K) x0 = __builtin_nontemporal_load(tail_src<T>(a, lane, off) + i);
    if (lane + 64 < a.K) x1 = __builtin_nontemporal_load(tail_src<T>(a, lane + 64, off) + i);
    T s = static_cast<T>(0);
    for (int k = 0; k < a.K; ++k) s = s + __shfl(k < 64 ? x0 : x1, k & 63, 64);
    if (lane == 0) static_cast<T*>(out)[i] = mean_finish(s, a.mt.Kall);
}

// `worker` = index of this block among the blocks that share the folded entries.
__device__ __forceinline__ void mean_tail(const ReduceArgs& a, unsigned worker) {
    const bool by_wave = a.mt.total <= kTailByWave;
    unsigned e = by_wave ? worker * kWavesPerBlock + threadIdx.x / 64u : worker * kThreads + threadIdx.x;
    if (e >= a.mt.total) return;
#pragma unroll
    for (int sgm = 0; sgm < kTailSegments; ++sgm) {
        if (e < a.mt.n[sgm]) {
            const long long off = a.mt.off[sgm];
            void* out = a.mt.out[sgm];
            if (a.mt.dtype[sgm] == FA_DT_F32) {
                if (by_wave) mean_tail_wave<float>(a, off, out, e); else mean_tail_element<float>(a, off, out, e);
            } else if (a.mt.dtype[sgm] == FA_DT_I64) {
                if (by_wave) mean_tail_wave<int64_t>(a, off, out, e); else mean_tail_element<int64_t>(a, off, out, e);
            } else {
                if (by_wave) mean_tail_wave<int32_t>(a, off, out, e); else mean_tail_element<int32_t>(a, off, out, e);
            }
            return;
        }
        e -= a.mt.n[sgm];
    }
}

// Blocks that share `total` folded elements.
inline unsigned tail_blocks(unsigned total) {
    return total <= kTailByWave ? (total + kWavesPerBlock - 1) / kWavesPerBlock : (total + kThreads - 1) / kThreads;
}

// Registers a thread holds for one tile (VPT vectors of VEC floats each).
template <int VEC, int VPT>
struct TileRegs {
    Pack<VEC> g[VPT], s[VPT], m[VPT], v[VPT];
};

// B clients k .. k+B-1 folded into the running step: all their loads first, then the
// arithmetic in client order.
template <int VEC, int VPT, int B, bool NT>
__device__ __forceinline__ void fold_batch(const ReduceArgs& a, int k, const unsigned (&idx)[VPT],
                                           TileRegs<VEC, VPT>& r) {
    Pack<VEC> xv[B][VPT];
#pragma unroll
    for (int u = 0; u < B; ++u)
#pragma unroll
        for (int p = 0; p < VPT; ++p) xv[u][p] = load_stream<VEC, NT>(a.x[k + u], idx[p]);
#pragma unroll
    for (int u = 0; u < B; ++u) {
        const float wk = a.w[k + u];
#pragma unroll
        for (int p = 0; p < VPT; ++p) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                float d = xv[u][p].e[j] - r.g[p].e[j];
                float wd = d * wk;
                r.s[p].e[j] = r.s[p].e[j] + wd;
            }
        }
    }
}

// The two halves of fold_batch on their own, for callers that keep the next batch's loads in
// flight while they fold the current one.
template <int VEC, int VPT, int B, bool NT>
__device__ __forceinline__ void load_batch(const ReduceArgs& a, int k, const unsigned (&idx)[VPT],
                                           Pack<VEC> (&xv)[B][VPT]) {
#pragma unroll
    for (int u = 0; u < B; ++u)
#pragma unroll
        for (int p = 0; p < VPT; ++p) xv[u][p] = load_stream<VEC, NT>(a.x[k + u], idx[p]);
}
template <int VEC, int VPT, int B>
__device__ __forceinline__ void apply_batch(const ReduceArgs& a, int k, const Pack<VEC> (&xv)[B][VPT],
                                            TileRegs<VEC, VPT>& r) {
#pragma unroll
    for (int u = 0; u < B; ++u) {
        const float wk = a.w[k + u];
#pragma unroll
        for (int p = 0; p < VPT; ++p) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                float d = xv[u][p].e[j] - r.g[p].e[j];
                float wd = d * wk;
                r.s[p].e[j] = r.s[p].e[j] + wd;
            }
        }
    }
}

// The clients k = 0 .. in whole batches of B, two buffers taking turns: the loads of batch b + 1
// are issued before batch b is folded, so a wave's loads never drain between batches.  Same
// additions in the same (client) order as fold_batch after fold_batch.  Returns the clients done.
template <int VEC, int VPT, int B, bool NT>
__device__ __forceinline__ int fold_batches_overlapped(const ReduceArgs& a, const unsigned (&idx)[VPT],
                                                       TileRegs<VEC, VPT>& r) {
    const int nb = a.K / B;
    if (nb == 0) return 0;
    Pack<VEC> xa[B][VPT], xb[B][VPT];
    load_batch<VEC, VPT, B, NT>(a, 0, idx, xa);
    int b = 1;                                   // xa holds batch b - 1
    for (; b + 1 < nb; b += 2) {
        load_batch<VEC, VPT, B, NT>(a, b * B, idx, xb);
        apply_batch<VEC, VPT, B>(a, (b - 1) * B, xa, r);
        load_batch<VEC, VPT, B, NT>(a, (b + 1) * B, idx, xa);
        apply_batch<VEC, VPT, B>(a, b * B, xb, r);
    }
    if (b < nb) {
        load_batch<VEC, VPT, B, NT>(a, b * B, idx, xb);
        apply_batch<VEC, VPT, B>(a, (b - 1) * B, xa, r);
        apply_batch<VEC, VPT, B>(a, b * B, xb, r);
    } else {
        apply_batch<VEC, VPT, B>(a, (b - 1) * B, xa, r);
    }
    return nb * B;
}

// Folds the K clients of this launch into the step of the vectors idx[0..VPT) and, on the last
// client chunk, runs the optimizer epilogue and forms g + step.  Everything stays in registers;
// the caller decides how the results reach memory.
// STEP_ONLY: stop after the clients (the averaged step in r.s, g as read in r.g) whatever a.last
// says -- for callers that run the epilogue themselves, later; g is then read with the caching
// policy, because the caller reads it again soon.
// OVERLAP: batches of U clients with the next batch's loads in flight (U <= 4).
template <int OPT, int VEC, int VPT, int U, bool NT, bool STEP_ONLY = false, bool OVERLAP = false>
__device__ __forceinline__ void fold_tile(const ReduceArgs& a, const unsigned (&idx)[VPT],
                                          TileRegs<VEC, VPT>& r) {
    const float* gsrc = a.apply ? a.g : a.g_ro;
#pragma unroll
    for (int p = 0; p < VPT; ++p) {
        // g, m and v are read once per round like the client rows, but whether the streaming
        // (nt) policy pays differs: measured +2 % for g under plain FedAvg, -1.5 % for g beside
        // two moment streams, +1.5-7 % for m / v of the two-moment optimizers, -1 % for FedAvgM's v.
        r.g[p] = (OPT == FA_OPT_NONE && !STEP_ONLY) ? load_stream<VEC, true>(gsrc, idx[p]) : load_state<VEC>(gsrc, idx[p]);
        if (a.step_in) {
            r.s[p] = load_state<VEC>(a.step_in, idx[p]);
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) r.s[p].e[j] = 0.0f;
        }
    }
    // Whole batches of U clients, then the rest in batches of 8 / 4 / 2 / 1: a client count that
    // is no multiple of U keeps several clients' loads in flight to the end (with one client at
    // a time, 12 clients at U = 16 ran 20 % slower than 16).
    int k = 0;
    if constexpr (OVERLAP) {
        static_assert(U <= 4, "two buffers of U clients x VPT vectors live at once");
        k = fold_batches_overlapped<VEC, VPT, U, NT>(a, idx, r);
    } else {
        for (; k + U <= a.K; k += U) fold_batch<VEC, VPT, U, NT>(a, k, idx, r);
    }
    if constexpr (U > 8) {
        if (k + 8 <= a.K) { fold_batch<VEC, VPT, 8, NT>(a, k, idx, r); k += 8; }
    }
    if constexpr (U > 4) {
        if (k + 4 <= a.K) { fold_batch<VEC, VPT, 4, NT>(a, k, idx, r); k += 4; }
    }
    if (k + 2 <= a.K) { fold_batch<VEC, VPT, 2, NT>(a, k, idx, r); k += 2; }
    if (k < a.K) fold_batch<VEC, VPT, 1, NT>(a, k, idx, r);
    if (STEP_ONLY || !a.last) return;
#pragma unroll
    for (int p = 0; p < VPT; ++p) {
        if (OPT != FA_OPT_NONE) {
            r.v[p] = (OPT >= FA_OPT_ADAM) ? load_stream<VEC, true>(a.v, idx[p]) : load_state<VEC>(a.v, idx[p]);
            if (OPT != FA_OPT_AVGM) r.m[p] = load_stream<VEC, true>(a.m, idx[p]);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                float mj = (OPT != FA_OPT_AVGM) ? r.m[p].e[j] : 0.0f;
                r.s[p].e[j] = optimizer_step<OPT>(r.s[p].e[j], mj, r.v[p].e[j], a.h);
                if (OPT != FA_OPT_AVGM) r.m[p].e[j] = mj;
            }
        }
        if (a.apply) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) r.g[p].e[j] = r.g[p].e[j] + r.s[p].e[j];
        }
    }
}

// One tile = kThreads * VPT vectors; a thread owns vectors tile_base + p * kThreads + tid.
// In the one partial tile (GUARD) out-of-range threads redo the last vector's loads and skip
// the stores, so the instruction stream is the unguarded one.
template <int OPT, int VEC, int VPT, int U, bool NT, bool GUARD>
__device__ __forceinline__ void reduce_tile(const ReduceArgs& a, unsigned base) {
    unsigned idx[VPT];
    bool ok[VPT];
#pragma unroll
    for (int p = 0; p < VPT; ++p) {
        unsigned i = base + p * kThreads + threadIdx.x;
        ok[p] = !GUARD || i < a.n;
        idx[p] = ok[p] ? i : a.n - 1;
    }
    TileRegs<VEC, VPT> r;
    fold_tile<OPT, VEC, VPT, U, NT>(a, idx, r);
#pragma unroll
    for (int p = 0; p < VPT; ++p) {
        if (!ok[p]) continue;
        if (!a.last) {
            store_result(a.step_out, idx[p], r.s[p], a.wt);
            continue;
        }
        if (OPT != FA_OPT_NONE) {
            store_result(a.v, idx[p], r.v[p], a.wt);
            if (OPT != FA_OPT_AVGM) store_result(a.m, idx[p], r.m[p], a.wt);
        }
        if (a.step_out) store_result(a.step_out, idx[p], r.s[p], a.wt);
        if (a.apply) store_result(a.g, idx[p], r.g[p], a.wt);
    }
}

template <int OPT, int VEC, int VPT, int U, bool NT>
__global__ __launch_bounds__(kThreads) void reduce_kernel(const ReduceArgs a) {
    constexpr unsigned kTile = kThreads * VPT;
    const unsigned full = a.n / kTile;  // tiles that need no bounds check
    const unsigned tiles = (a.n + kTile - 1) / kTile;
    // Blocks are dispatched round-robin over the 8 XCDs.  Inside every window of 8*R tiles
    // each XCD takes R consecutive tiles instead of every eighth one (measured +4-5 % for the
    // two-moment optimizers at K = 32..64, +-1 % elsewhere); a ragged last window keeps the
    // plain order.
    constexpr unsigned R = (OPT >= FA_OPT_ADAM) ? 32u : 128u;
    // the first mt.blocks blocks of the grid take the folded plain-mean entries instead of tiles
    const unsigned tb = a.mt.blocks;
    if (blockIdx.x < tb) {
        mean_tail(a, blockIdx.x);
        return;
    }
    const unsigned nblocks = gridDim.x - tb;
    for (unsigned t0 = blockIdx.x - tb; t0 < tiles; t0 += nblocks) {
        unsigned t = t0;
        const unsigned win = t0 & ~(8u * R - 1u);
        if (VEC == 4 && win + 8u * R <= tiles) t = win + (t0 & 7u) * R + ((t0 >> 3) & (R - 1u));
        if (t < full)
            reduce_tile<OPT, VEC, VPT, U, NT, false>(a, t * kTile);
        else
            reduce_tile<OPT, VEC, VPT, U, NT, true>(a, t * kTile);
    }
}

// ---------------------------------------------------------------------------------------------
// Persistent variant: one 256-thread block per CU takes every gridDim.x-th 16-KiB tile of the
// parameter axis (so at any moment the chip works on one contiguous stretch of every client row
// -- measured 1.5-2 % faster than giving each block a contiguous range of its own) and parks each
// tile's results on chip: the first tiles of a phase in LDS (every thread only ever touches its
// own slots, so no barrier is needed), further ones in registers the pipeline leaves unused.  At
// the end of a phase it writes them out in one burst of write-through stores.
//
// Why: what a result byte costs on MI355X depends on how DILUTED the writes are among the reads.
// Measured with read-only and write-only kernels alternating over chunks of the vector (K = 32,
// 650 MB rows): 3.25 ms with 35 MB written per phase, 3.45 ms with 17 MB, 3.9 ms with 8 MB or
// less -- against 3.06 ms for the reads alone; and the same 102 MB of results cost 0.045 ms
// extra when spread over an 8-client round but 0.15 ms over a 128-client round.  So the blocks
// write in PHASES: all of them MEET (grid_meet, a bounded wait that carries no data) before
// their bursts, each drains its own burst (s_waitcnt vmcnt(0) on write-through stores), and only
// then issues the loads of the next phase, whose pipeline therefore starts empty.  Worth
// 6-7 % on a 650 MB vector over bursts that every block fires on its own; meeting again
// after the bursts, meeting between bursts, or dealing the tiles in groups so that a burst is
// contiguous, is not.  A vector that the chip can park whole (up to 18 tiles per CU under plain
// FedAvg) is written in one burst when all its reads are done (+1-2 % over several bursts).

#ifndef FA_FOLD_OVERLAP
#define FA_FOLD_OVERLAP 0          /* persistent kernels: next batch's loads in flight while one is folded (A/B knob; measured 3-6 % slower in the wide-batch kernel, 2-3 % in the late-epilogue kernel) */
#endif
#ifndef FA_PARK_FOLD_AHEAD
#define FA_PARK_FOLD_AHEAD 2    /* clients ahead per vector in the wide-batch persistent kernel (A/B knob) */
#endif
constexpr int kStashVpt = 4;
constexpr int kStashUnroll = 8;
constexpr unsigned kStashTile = kThreads * kStashVpt;                 // vectors per tile
constexpr unsigned kStashVecBytes = kStashTile * 16;                  // 16 KiB per result vector
constexpr unsigned kStashBytes = 160 * 1024;                          // LDS budget of a block

#ifndef FA_STASH_MAX_TILES
#define FA_STASH_MAX_TILES 10
#endif
constexpr int kStashMaxTiles = FA_STASH_MAX_TILES;
#ifndef FA_STASH_REG_BUDGET
#define FA_STASH_REG_BUDGET 128
#endif
#ifndef FA_STASH_ADAM_REGS
#define FA_STASH_ADAM_REGS 0
#endif
constexpr unsigned kPhaseTiles = 4;   // tiles per phase of a vector too long to be parked whole
constexpr int kStashRegBudget = FA_STASH_REG_BUDGET;   // registers per lane that may hold parked results

// What a launch writes back per tile.
enum { kOutFinal = 0, kOutFinalStep = 1, kOutStepOnly = 2 };

template <int OPT, int MODE>
struct StashShape {
    static constexpr int kVectors =
        MODE == kOutStepOnly
            ? 1
            : 1 + (OPT != FA_OPT_NONE) + (OPT >= FA_OPT_ADAM) + (MODE == kOutFinalStep);
    static constexpr int kTiles =
        (kStashBytes / kStashVecBytes / kVectors) > kStashMaxTiles ? kStashMaxTiles : (kStashBytes / kStashVecBytes / kVectors);
    // further tiles of a phase whose results wait in registers (16 per result vector and tile):
    // at one wave per SIMD the pipeline leaves about a quarter of the 512 registers unused
    static constexpr int kRegTiles =
        (OPT == FA_OPT_NONE ? kStashRegBudget : OPT == FA_OPT_AVGM ? kStashRegBudget / 2 : FA_STASH_ADAM_REGS) / (16 * kVectors);
};

__device__ __forceinline__ void burst_store(float* base, unsigned bytes, unsigned idx, const f4& v) {
    // write-through (sc0 sc1) 128-bit store
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, static_cast<int>(bytes), 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), rs, idx * 16u, 0, 17);
}

#include "grid_meet.h"

// The client loop of the persistent kernel is software-pipelined ACROSS tiles: two register
// sets of kStashUnroll clients x kStashVpt vectors are in flight; after a set is consumed the
// loads of the set two steps ahead are issued -- which may belong to the next tile, whose g / m
// / v are fetched with its first set.  The optimizer epilogue, the LDS parking and the burst
// stores of a tile therefore run while the next tile's client bytes are already on their way
// (with one wave per SIMD nothing else would hide them), and at one wave per SIMD the kernel
// may spend up to 512 VGPRs per lane on this.
template <int OPT, int MODE>
__global__ __launch_bounds__(kThreads, 1) void reduce_stash_kernel(const ReduceArgs a) {
    constexpr int NV = StashShape<OPT, MODE>::kVectors;
    constexpr int S = StashShape<OPT, MODE>::kTiles;       // tiles of a phase parked in LDS
    constexpr int RT = StashShape<OPT, MODE>::kRegTiles;   // ... and in registers
    constexpr int PT = S + RT;                             // tiles per phase (one burst)
    constexpr int U = kStashUnroll, VPT = kStashVpt;
    constexpr bool kHasV = MODE != kOutStepOnly && OPT != FA_OPT_NONE;
    constexpr bool kHasM = MODE != kOutStepOnly && OPT >= FA_OPT_ADAM;
    __shared__ f4 stash[S][NV][VPT][kThreads];
    // contiguous range of this block, in units of one wave-row (64 vectors)
    const unsigned units = (a.n + 63u) / 64u;
    const unsigned u0 = static_cast<unsigned>(static_cast<uint64_t>(units) * blockIdx.x / gridDim.x);
    const unsigned u1 = static_cast<unsigned>(static_cast<uint64_t>(units) * (blockIdx.x + 1) / gridDim.x);
    unsigned r0 = u0 * 64u;
    unsigned r1 = (u1 * 64u < a.n) ? u1 * 64u : a.n;
    unsigned ntiles, tstride = kStashTile, tlen = kStashTile;
    // Chip-wide write phases (only with round-robin tiles, where all blocks have equal work).
    bool phased = a.sync != 0 && a.il != 0 && a.sync_word != nullptr;
    unsigned P = static_cast<unsigned>(PT);   // tiles per phase: at most what LDS and the spare registers hold
    GridMeet meet = {a.sync_word, 0u, false, gridDim.x};
    unsigned meetings = 0;
    unsigned shift = 0;   // staggered phases: this block's first phase is `shift` tiles shorter
    if (a.il) {
        // Every block gets the same number of equally long tiles where that is possible: the
        // tile is shortened, in steps of one wave-row (64 vectors = 1 KiB; finer steps cost
        // more in misaligned wave loads than they balance), so that the last round is as full
        // as the others; lanes past the tile's end idle.
        const unsigned per_round = gridDim.x * kStashTile;
        const unsigned rounds = (a.n + per_round - 1) / per_round;
        const unsigned slots = gridDim.x * rounds;
        tlen = (((a.n + slots - 1) / slots) + 63u) & ~63u;
        const unsigned total = (a.n + tlen - 1) / tlen;
        // Blocks are dispatched round-robin over the 8 XCDs; give each XCD a contiguous run of
        // a round's tiles (slot = xcd * blocks_per_xcd + index within the XCD) rather than every
        // eighth one: a consistent +0.5 % (fewer rows open per XCD's slice of the fabric).
        unsigned slot = blockIdx.x;
        if (gridDim.x % 8 == 0) slot = (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8;
        // folded plain-mean entries first, on the blocks with the highest slots: those are the
        // ones with a tile less than the others when the tiles do not divide evenly
        if (a.mt.total) mean_tail(a, gridDim.x - 1u - slot);
        // every block attends as many meetings as the busiest block
        const unsigned most = (total + gridDim.x - 1) / gridDim.x;
        // A vector short enough to be parked whole is written in one burst at the very end, with
        // no meeting; a longer one in phases of kPhaseTiles tiles (measured on a 155-tile-per-CU
        // vector of 32 clients, kernel ms: no phases 3.79, phases of 1 / 2 / 3 / 4 / 5 / 6 / 8 /
        // 12 / 16 tiles 4.17 / 3.73 / 3.60 / 3.56 / 3.59 / 3.60 / 3.62 / 3.68 / 3.69: shorter
        // ones pay the drain-and-refill too often, in longer ones the blocks drift apart).
        if (a.phase_tiles > 0) {
            if (static_cast<unsigned>(a.phase_tiles) < P) P = static_cast<unsigned>(a.phase_tiles);
        } else if (most > P) {
            // Three or four result vectors per tile (the two-moment optimizers): phases of two
            // tiles -- with the two staggered groups below, kernel time against phases of three
            // (all that LDS holds): -2.8 % at 25 M elements x 64 clients, -2.5 % at 25 M x 48,
            // -1...-2 % at 87 M x 64, -2...-3 % at 162 M x 32, -1 % at 12 M x 64 and 25 M x 100.
            const unsigned want = NV >= 3 ? 2u : kPhaseTiles;
            if (P > want) P = want;
        }
        if (most <= P) phased = false;
        if (phased && (a.sync >> 2) && gridDim.x % 8 == 0) {
            // Groups of XCDs whose phases are a fraction of a phase apart: while one group drains
            // its bursts and refills its pipeline, the others keep the memory system reading.
            // Each group meets on a counter of its own (16 bytes apart in the launch's slot).
            // Measured against one chip-wide group (kernel time): odd / even XCDs -1.5...-2.4 % on
            // a 650 MB vector of 32 clients, -1.0...-1.4 % for FedAvgM / FedAvg at 25-47 M
            // elements x 48-64 clients, +-0.5 % for the two-moment optimizers and for K = 100;
            // four groups the same or slightly worse; XCDs 0-3 / 4-7 +1...+4 % (slower than one group).
            const unsigned mode = static_cast<unsigned>(a.sync >> 2) & 3u;   // 1: 2 groups (odd / even XCDs), 2: 4 groups, 3: 2 groups (XCDs 0-3 / 4-7)
            const unsigned xcd = blockIdx.x % 8u;
            const unsigned groups = mode == 2u ? 4u : 2u;
            const unsigned group = mode == 3u ? xcd >> 2 : xcd % groups;
            meet.word = a.sync_word + 4u * group;
            meet.blocks = gridDim.x / groups;
            shift = group * P / groups;
        }
        if (phased) meetings = ((most + shift + P - 1) / P) * (((a.sync & 1) ? 1u : 0u) + ((a.sync & 2) ? 1u : 0u));
        if (phased) grid_meet_enter(meet, a.sync_epoch);
        if (slot >= total) {
            if (phased) grid_meet_leave(meet, meetings);
            return;
        }
        ntiles = (total - slot + gridDim.x - 1) / gridDim.x;
        r0 = slot * tlen;
        r1 = a.n;
        tstride = gridDim.x * tlen;
    } else {
        if (a.mt.total) mean_tail(a, gridDim.x - 1u - blockIdx.x);
        if (r0 >= r1) return;
        ntiles = (r1 - r0 + kStashTile - 1) / kStashTile;
    }
    const int nb = (a.K + U - 1) / U;   // client sets per tile (the host guarantees nb >= 2)
    const unsigned nbytes = a.n * 16u;
    const float* gsrc = a.apply ? a.g : a.g_ro;
    float* dst[4] = {nullptr, nullptr, nullptr, nullptr};
    if (MODE == kOutStepOnly) {
        dst[0] = a.step_out;
    } else {
        int c = 0;
        dst[c++] = a.g;
        if (kHasV) dst[c++] = a.v;
        if (kHasM) dst[c++] = a.m;
        if (MODE == kOutFinalStep) dst[c++] = a.step_out;
    }

    Pack<4> xa[U][VPT], xb[U][VPT];
    Pack<4> g_cur[VPT], s_cur[VPT], m_cur[VPT], v_cur[VPT];
    Pack<4> g_nxt[VPT], s_nxt[VPT], m_nxt[VPT], v_nxt[VPT];
    f4 rpark[RT > 0 ? RT : 1][NV][VPT];   // results of tiles S.. of the phase (static indices only)
    unsigned it = 0, ct = 0;   // tile being issued / consumed
    int ib = 0, cb = 0;        // client set within that tile
    // Phased mode: loads are only issued for the tiles of the current phase (S tiles = one
    // burst), so that the pipeline is empty when the burst goes out; `full_*` tell which
    // register set holds an issued, not yet consumed client set.
    unsigned issue_end = phased ? P - shift : 0xFFFFFFFFu;
    bool full_a = false, full_b = false, burst_done = false;

    auto tile_start = [&](unsigned tile) -> unsigned { return r0 + tile * tstride; };
    auto vec_index = [&](unsigned tile, int p) -> unsigned {
        const unsigned off = p * kThreads + threadIdx.x;
        const unsigned i = r0 + tile * tstride + off;
        return (off < tlen && i < r1) ? i : r1 - 1;   // idle lanes redo the last vector, never store
    };
    auto issue = [&](Pack<4> (&x)[U][VPT], bool& full) {
        if (it >= ntiles || it >= issue_end) return;
        full = true;
        unsigned idx[VPT];
#pragma unroll
        for (int p = 0; p < VPT; ++p) idx[p] = vec_index(it, p);
        if (ib == 0) {
#pragma unroll
            for (int p = 0; p < VPT; ++p) {
                g_nxt[p] = load_state<4>(gsrc, idx[p]);
                if (a.step_in) s_nxt[p] = load_state<4>(a.step_in, idx[p]);
                if (kHasV && a.last) v_nxt[p] = load_state<4>(a.v, idx[p]);
                if (kHasM && a.last) m_nxt[p] = load_state<4>(a.m, idx[p]);
            }
        }
        const int k0 = ib * U;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = (k0 + u < a.K) ? k0 + u : a.K - 1;
            const float* xp = a.x[k];
#pragma unroll
            for (int p = 0; p < VPT; ++p) x[u][p] = load_stream<4, true>(xp, idx[p]);
        }
        if (++ib == nb) {
            ib = 0;
            ++it;
        }
    };
    auto finish_tile = [&]() {
        const int j = static_cast<int>((ct + shift) % P);
#pragma unroll
        for (int p = 0; p < VPT; ++p) {
            if (a.last && MODE != kOutStepOnly) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (OPT != FA_OPT_NONE) {
                        float mj = kHasM ? m_cur[p].e[e] : 0.0f;
                        s_cur[p].e[e] = optimizer_step<OPT>(s_cur[p].e[e], mj, v_cur[p].e[e], a.h);
                        if (kHasM) m_cur[p].e[e] = mj;
                    }
                    g_cur[p].e[e] = g_cur[p].e[e] + s_cur[p].e[e];
                }
            }
            int c = 0;
            auto park = [&](const Pack<4>& v) {
                f4 t;
                t.x = v.e[0]; t.y = v.e[1]; t.z = v.e[2]; t.w = v.e[3];
                if (j < S) {
                    stash[j][c][p][threadIdx.x] = t;
                } else {
                    // j is the same for the whole block: one of these runs, the index is static
#pragma unroll
                    for (int q = 0; q < RT; ++q)
                        if (j - S == q) rpark[q][c][p] = t;
                }
                ++c;
            };
            if (MODE == kOutStepOnly) {
                park(s_cur[p]);
            } else {
                park(g_cur[p]);
                if (kHasV) park(v_cur[p]);
                if (kHasM) park(m_cur[p]);
                if (MODE == kOutFinalStep) park(s_cur[p]);
            }
        }
        if (j == static_cast<int>(P) - 1 || ct + 1 == ntiles) {
            const unsigned first = ct - static_cast<unsigned>(j);   // (modular: tile of slot jj = first + jj)
            // slots below j_lo belong to no tile: the shortened first phase of a staggered block
            const int j_lo = (ct < P && static_cast<unsigned>(j) >= ct) ? j - static_cast<int>(ct) : 0;
            burst_done = true;
            if (phased && (a.sync & 2)) grid_meet(meet);   // nobody writes before everybody has read
#pragma unroll 1
            for (int jj = j_lo; jj <= j && jj < S; ++jj) {
#pragma unroll
                for (int c = 0; c < NV; ++c)
#pragma unroll
                    for (int p = 0; p < VPT; ++p) {
                        const unsigned off = p * kThreads + threadIdx.x;
                        const unsigned i = tile_start(first + jj) + off;
                        if (off < tlen && i < r1) burst_store(dst[c], nbytes, i, stash[jj][c][p][threadIdx.x]);
                    }
            }
#pragma unroll
            for (int q = 0; q < RT; ++q) {
                if (S + q <= j && S + q >= j_lo) {
#pragma unroll
                    for (int c = 0; c < NV; ++c)
#pragma unroll
                        for (int p = 0; p < VPT; ++p) {
                            const unsigned off = p * kThreads + threadIdx.x;
                            const unsigned i = tile_start(first + S + q) + off;
                            if (off < tlen && i < r1) burst_store(dst[c], nbytes, i, rpark[q][c][p]);
                        }
                }
            }
        }
    };
    auto consume = [&](Pack<4> (&x)[U][VPT], bool& full) {
        if (!full) return;
        full = false;
        if (cb == 0) {
#pragma unroll
            for (int p = 0; p < VPT; ++p) {
                g_cur[p] = g_nxt[p];
                if (kHasV) v_cur[p] = v_nxt[p];
                if (kHasM) m_cur[p] = m_nxt[p];
                if (a.step_in) {
                    s_cur[p] = s_nxt[p];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) s_cur[p].e[e] = 0.0f;
                }
            }
        }
        const int k0 = cb * U;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (k0 + u < a.K) {
                const float wk = a.w[k0 + u];
#pragma unroll
                for (int p = 0; p < VPT; ++p) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float d = x[u][p].e[e] - g_cur[p].e[e];
                        float wd = d * wk;
                        s_cur[p].e[e] = s_cur[p].e[e] + wd;
                    }
                }
            }
        }
        if (++cb == nb) {
            cb = 0;
            finish_tile();
            ++ct;
        }
    };

    // After a burst in phased mode: wait until this block's stores have left (they are
    // write-through, so their completion means the bytes are on their way to HBM), meet the
    // other blocks, then open the next phase's tiles for issue.
    auto end_phase = [&]() {
        if (!burst_done) return;
        burst_done = false;
        if (!phased) return;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the pipeline is empty, only stores are out
        if (a.sync & 1) grid_meet(meet);
        if (issue_end != 0xFFFFFFFFu) issue_end += P;
    };
    issue(xa, full_a);
    issue(xb, full_b);
    while (ct < ntiles) {
        consume(xa, full_a);
        end_phase();
        issue(xa, full_a);
        consume(xb, full_b);
        end_phase();
        issue(xb, full_b);
        // nothing in flight and nothing issued: cannot happen while tiles remain, but a wave
        // must never spin here
        if (!full_a && !full_b) break;
    }
    if (phased) grid_meet_leave(meet, meetings);
}

// ---------------------------------------------------------------------------------------------
// Persistent kernel with plain client batches (the "wide-batch" kernel), for plain FedAvg: one
// 256-thread block per CU takes every gridDim.x-th tile like reduce_stash_kernel, but folds a tile
// with fold_tile (FA_PARK_FOLD_AHEAD = 2 clients x 4 vectors = 8 loads in flight per lane, drained
// before the next 8; no pipelining across tiles), parks the tile's result in LDS and writes up to
// ten tiles at a time in the same chip-wide, staggered phases.  Measured with 16 clients in flight,
// against the better of the other two kernels (fraction of peak; the figures with 2 in flight are
// with the routing rule, park_wins): 162.5 M elements x 32 clients 0.75 -> 0.85, x 16 0.765 ->
// 0.80, x 64 0.80 -> 0.83; 120 M x 32 0.75 -> 0.815; 87 M x 16 / 32 / 64 0.77 / 0.75 / 0.79 -> 0.79
// / 0.81 / 0.82; 325 M x 32 0.79 -> 0.82; level at 65 M elements and at 128 clients, behind below
// (47 M x 48: 0.76 against 0.79), and far behind under the optimizers, whose epilogue has
// nothing to hide behind at one wave per SIMD (FedAdam 0.53-0.64 against 0.70-0.75; FedAvgM
// -1...-3 %).  Short client lists gain as much (8 / 12 clients on 162.5 M: 0.72 / 0.75 -> 0.77 /
// 0.80).  Phases of ten tiles (all that LDS holds) beat shorter ones on very long vectors (4 / 6 /
// 8 / 10 tiles on 162.5 M x 32: 0.81 / 0.84 / 0.84 / 0.85); long client lists on vectors of 11-48
// tiles per CU take three (see where P is chosen).  It takes the final form of a launch
// (weighted step and apply, no step output) and, as plain FedAvg whatever the optimizer, the
// step-only forms (a client chunk that is not the last one; fa_reduce_step).
template <int OPT>
__global__ __launch_bounds__(kThreads, 1) void reduce_park_kernel(const ReduceArgs a) {
    constexpr int NV = 1 + (OPT != FA_OPT_NONE) + (OPT >= FA_OPT_ADAM);
    constexpr int S = (kStashBytes / kStashVecBytes / NV) > kStashMaxTiles ? kStashMaxTiles
                                                                           : (kStashBytes / kStashVecBytes / NV);
    constexpr int VPT = kStashVpt;
    __shared__ f4 stash[S][NV][VPT][kThreads];
    const unsigned per_round = gridDim.x * kStashTile;
    const unsigned rounds = (a.n + per_round - 1) / per_round;
    const unsigned slots = gridDim.x * rounds;
    const unsigned tlen = (((a.n + slots - 1) / slots) + 63u) & ~63u;
    const unsigned total = (a.n + tlen - 1) / tlen;
    unsigned slot = blockIdx.x;
    if (gridDim.x % 8 == 0) slot = (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8;
    if (a.mt.total) mean_tail(a, gridDim.x - 1u - slot);
    const unsigned most = (total + gridDim.x - 1) / gridDim.x;
    unsigned P = static_cast<unsigned>(S);
    bool phased = a.sync != 0 && a.sync_word != nullptr;
    if (a.phase_tiles > 0) {
        if (static_cast<unsigned>(a.phase_tiles) < P) P = static_cast<unsigned>(a.phase_tiles);
    } else if (a.K >= 40 && most > P && most <= 48u) {
        // long client lists on vectors of 11-48 tiles per CU: phases of three tiles (kernel time against
        // ten: 11.7 M elements x 64-128 clients -0.5...-1 %, 25.6 M x 48-128 -1.1...-1.4 %, 47 M x 40-100
        // -1.3...-3 %); ten stay ahead from 82 tiles per CU (87-162 M x 64: +0.4...+1.4 % for three) and
        // at shorter client lists (162 M x 6-12: +12...+25 %)
        P = 3u;
    }
    if (most <= P) phased = false;
    GridMeet meet = {a.sync_word, 0u, false, gridDim.x};
    unsigned shift = 0, meetings = 0;
    if (phased && (a.sync >> 2) && gridDim.x % 8 == 0) {
        const unsigned mode = static_cast<unsigned>(a.sync >> 2) & 3u;
        const unsigned xcd = blockIdx.x % 8u;
        const unsigned groups = mode == 2u ? 4u : 2u;
        const unsigned group = mode == 3u ? xcd >> 2 : xcd % groups;
        meet.word = a.sync_word + 4u * group;
        meet.blocks = gridDim.x / groups;
        shift = group * P / groups;
    }
    if (phased) meetings = (most + shift + P - 1) / P;
    if (phased) grid_meet_enter(meet, a.sync_epoch);
    if (slot >= total) {
        if (phased) grid_meet_leave(meet, meetings);
        return;
    }
    const unsigned ntiles = (total - slot + gridDim.x - 1) / gridDim.x;
    const unsigned r0 = slot * tlen, r1 = a.n, tstride = gridDim.x * tlen;
    const unsigned nbytes = a.n * 16u;
    // the step-only forms (a client chunk that is not the last, or no apply) leave the step
    const bool whole = a.last && a.apply;
    float* dst[3] = {whole ? a.g : a.step_out, OPT != FA_OPT_NONE ? a.v : nullptr, OPT >= FA_OPT_ADAM ? a.m : nullptr};
    for (unsigned ct = 0; ct < ntiles; ++ct) {
        unsigned idx[VPT];
#pragma unroll
        for (int p = 0; p < VPT; ++p) {
            const unsigned off = p * kThreads + threadIdx.x;
            const unsigned i = r0 + ct * tstride + off;
            idx[p] = (off < tlen && i < r1) ? i : r1 - 1;   // idle lanes redo the last vector, never store
        }
        TileRegs<4, VPT> r;
        fold_tile<OPT, 4, VPT, FA_PARK_FOLD_AHEAD, true, false, FA_FOLD_OVERLAP != 0>(a, idx, r);
        const int j = static_cast<int>((ct + shift) % P);
#pragma unroll
        for (int p = 0; p < VPT; ++p) {
            auto as_f4 = [](const Pack<4>& v) {
                f4 t;
                t.x = v.e[0]; t.y = v.e[1]; t.z = v.e[2]; t.w = v.e[3];
                return t;
            };
            stash[j][0][p][threadIdx.x] = as_f4(whole ? r.g[p] : r.s[p]);
            if (OPT != FA_OPT_NONE) stash[j][NV > 1 ? 1 : 0][p][threadIdx.x] = as_f4(r.v[p]);
            if (OPT >= FA_OPT_ADAM) stash[j][NV > 2 ? 2 : 0][p][threadIdx.x] = as_f4(r.m[p]);
        }
        if (j == static_cast<int>(P) - 1 || ct + 1 == ntiles) {
            const unsigned first = ct - static_cast<unsigned>(j);
            const int j_lo = (ct < P && static_cast<unsigned>(j) >= ct) ? j - static_cast<int>(ct) : 0;
            if (phased) grid_meet(meet);
#pragma unroll 1
            for (int jj = j_lo; jj <= j; ++jj) {
#pragma unroll
                for (int c = 0; c < NV; ++c)
#pragma unroll
                    for (int p = 0; p < VPT; ++p) {
                        const unsigned off = p * kThreads + threadIdx.x;
                        const unsigned i = r0 + (first + jj) * tstride + off;
                        if (off < tlen && i < r1) burst_store(dst[c], nbytes, i, stash[jj][c][p][threadIdx.x]);
                    }
            }
            if (phased) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain before the next phase's loads
        }
    }
    if (phased) grid_meet_leave(meet, meetings);
}

// ---------------------------------------------------------------------------------------------
// Persistent kernel with a LATE optimizer epilogue, for FedAvgM / FedAdam / FedYogi / FedAdagrad.
// The fused kernels above write g, m and v of every tile as soon as its clients are folded: three
// (two) thin result streams trickling out among the reads, which is what holds the two-moment
// optimizers at 0.69-0.72 of peak at 16-40 clients, and with three result vectors per tile LDS
// can only park 3 tiles per phase.  Here a tile's clients are folded in plain batches
// (FA_LATE_FOLD_AHEAD = 2 clients x 4 vectors in flight per lane) into the AVERAGED STEP alone, which
// is parked in LDS -- one vector per tile, so up to TEN tiles per phase like plain FedAvg's
// wide-batch kernel; three by default, see where P is chosen -- and the
// epilogue runs at the end of the phase over the parked tiles: g, m and v are read (g for the
// second time, a few hundred microseconds after the first: 4 of 4 (K + 6) + 4 bytes per
// parameter), the optimizer formulas evaluated, g, m and v written.  A phase is therefore a long
// stretch of pure reads followed by a copy-like burst (as many bytes read as written, each CU on
// its own tiles), the two mixes HBM serves best, in the same chip-wide staggered phases.
// Same operations on the same values in the same order as the fused kernels: same bits.
#ifndef FA_LATE_BATCH
#define FA_LATE_BATCH 3
#endif
#ifndef FA_LATE_FOLD_AHEAD
#define FA_LATE_FOLD_AHEAD 2    /* clients ahead per vector while a tile is folded (A/B knob) */
#endif
#ifndef FA_LATE_DEBUG
#define FA_LATE_DEBUG 0   /* 1-3: timing experiments that drop parts of the epilogue (wrong results) */
#endif
constexpr int kLateBatch = FA_LATE_BATCH;   // tiles whose g / m / v the epilogue requests at once
constexpr unsigned kLatePhaseTiles = 3;

template <int OPT>
__global__ __launch_bounds__(kThreads, 1) void reduce_late_kernel(const ReduceArgs a) {
    static_assert(OPT != FA_OPT_NONE, "plain FedAvg has no epilogue to defer");
    constexpr int S = (kStashBytes / kStashVecBytes) > kStashMaxTiles ? kStashMaxTiles : (kStashBytes / kStashVecBytes);
    constexpr int VPT = kStashVpt;
    constexpr bool kHasM = OPT >= FA_OPT_ADAM;
    __shared__ f4 stash[S][VPT][kThreads];
    const unsigned per_round = gridDim.x * kStashTile;
    const unsigned rounds = (a.n + per_round - 1) / per_round;
    const unsigned slots = gridDim.x * rounds;
    const unsigned tlen = (((a.n + slots - 1) / slots) + 63u) & ~63u;
    const unsigned total = (a.n + tlen - 1) / tlen;
    unsigned slot = blockIdx.x;
    if (gridDim.x % 8 == 0) slot = (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8;
    if (a.mt.total) mean_tail(a, gridDim.x - 1u - slot);
    const unsigned most = (total + gridDim.x - 1) / gridDim.x;
    // phases of three tiles (kernel time against six: -1...-2 % at 11.7-25.6 M elements x 24-100
    // clients, -2...-4.5 % at 47-87 M; against ten, all that LDS holds, another 1-2 % more; two tiles
    // are better still on long vectors at long client lists but lose 2-3 % at 24 clients); vectors
    // of up to 8 tiles per CU keep six (4.5 M elements, one burst: -2...-4 % against three; 8 M: -1.6 %)
    unsigned P = most <= 8u ? 6u : kLatePhaseTiles;
    bool phased = a.sync != 0 && a.sync_word != nullptr;
    if (a.phase_tiles > 0) P = static_cast<unsigned>(a.phase_tiles) < static_cast<unsigned>(S) ? static_cast<unsigned>(a.phase_tiles) : static_cast<unsigned>(S);
    if (most <= P) phased = false;
    GridMeet meet = {a.sync_word, 0u, false, gridDim.x};
    unsigned shift = 0, meetings = 0;
    if (phased && (a.sync >> 2) && gridDim.x % 8 == 0) {
        const unsigned mode = static_cast<unsigned>(a.sync >> 2) & 3u;
        const unsigned xcd = blockIdx.x % 8u;
        const unsigned groups = mode == 2u ? 4u : 2u;
        const unsigned group = mode == 3u ? xcd >> 2 : xcd % groups;
        meet.word = a.sync_word + 4u * group;
        meet.blocks = gridDim.x / groups;
        shift = group * P / groups;
    }
    if (phased) meetings = (most + shift + P - 1) / P;
    if (phased) grid_meet_enter(meet, a.sync_epoch);
    if (slot >= total) {
        if (phased) grid_meet_leave(meet, meetings);
        return;
    }
    const unsigned ntiles = (total - slot + gridDim.x - 1) / gridDim.x;
    const unsigned r0 = slot * tlen, r1 = a.n, tstride = gridDim.x * tlen;
    const unsigned nbytes = a.n * 16u;
    auto tile_index = [&](unsigned tile, int p, bool& live) -> unsigned {
        const unsigned off = p * kThreads + threadIdx.x;
        const unsigned i = r0 + tile * tstride + off;
        live = off < tlen && i < r1;
        return live ? i : r1 - 1;   // idle lanes redo the last vector, never store
    };
    auto as_f4 = [](const Pack<4>& v) {
        f4 t;
        t.x = v.e[0]; t.y = v.e[1]; t.z = v.e[2]; t.w = v.e[3];
        return t;
    };
    for (unsigned ct = 0; ct < ntiles; ++ct) {
        unsigned idx[VPT];
#pragma unroll
        for (int p = 0; p < VPT; ++p) {
            bool live;
            idx[p] = tile_index(ct, p, live);
        }
        TileRegs<4, VPT> r;
        fold_tile<FA_OPT_NONE, 4, VPT, FA_LATE_FOLD_AHEAD, true, true, FA_FOLD_OVERLAP != 0>(a, idx, r);
        const int j = static_cast<int>((ct + shift) % P);
#pragma unroll
        for (int p = 0; p < VPT; ++p) stash[j][p][threadIdx.x] = as_f4(r.s[p]);
        if (j == static_cast<int>(P) - 1 || ct + 1 == ntiles) {
            const unsigned first = ct - static_cast<unsigned>(j);
            const int j_lo = (ct < P && static_cast<unsigned>(j) >= ct) ? j - static_cast<int>(ct) : 0;
            if (phased) grid_meet(meet);
            // The epilogue over the parked tiles, kLateBatch tiles at a time: all of a batch's
            // g / m / v loads are issued at once (3 x 4 x kLateBatch 16-byte loads in flight per
            // lane, so the burst streams like the client reads do), and the next batch's loads
            // fly while this batch's formulas are evaluated and its results stored.
            constexpr int B = kLateBatch;
            Pack<4> gq[2][B][VPT], mq[2][B][VPT], vq[2][B][VPT];
            auto fetch = [&](int jj0, auto bsel) {
                constexpr int b = decltype(bsel)::value;
#pragma unroll
                for (int t = 0; t < B; ++t) {
                    const int jj = jj0 + t <= j ? jj0 + t : j;   // past the end: redo the last tile's loads
#pragma unroll
                    for (int p = 0; p < VPT; ++p) {
                        bool live;
                        const unsigned i = tile_index(first + jj, p, live);
#if FA_LATE_DEBUG == 3   /* timing experiment: no epilogue loads */
                        gq[b][t][p] = Pack<4>{{1.0f, 2.0f, 3.0f, float(i)}};
                        vq[b][t][p] = gq[b][t][p];
                        if (kHasM) mq[b][t][p] = gq[b][t][p];
#else
                        gq[b][t][p] = load_state<4>(a.g, i);
                        vq[b][t][p] = kHasM ? load_stream<4, true>(a.v, i) : load_state<4>(a.v, i);
                        if (kHasM) mq[b][t][p] = load_stream<4, true>(a.m, i);
#endif
                    }
                }
            };
            auto finish = [&](int jj0, auto bsel) {
                constexpr int b = decltype(bsel)::value;
#pragma unroll
                for (int t = 0; t < B; ++t) {
                    const int jj = jj0 + t;
                    if (jj > j) break;
#pragma unroll
                    for (int p = 0; p < VPT; ++p) {
                        bool live;
                        const unsigned i = tile_index(first + jj, p, live);
                        const f4 st = stash[jj][p][threadIdx.x];
                        float sv[4] = {st.x, st.y, st.z, st.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float mj = kHasM ? mq[b][t][p].e[e] : 0.0f;
                            const float step = optimizer_step<OPT>(sv[e], mj, vq[b][t][p].e[e], a.h);
                            if (kHasM) mq[b][t][p].e[e] = mj;
                            gq[b][t][p].e[e] = gq[b][t][p].e[e] + step;
                        }
#if FA_LATE_DEBUG == 2   /* timing experiment: no epilogue stores (the impossible test keeps the values alive) */
                        if (live && gq[b][t][p].e[0] == 1.2345e38f && vq[b][t][p].e[1] == 5.4321e37f && (!kHasM || mq[b][t][p].e[2] == 3.3e36f)) {
#else
                        if (live) {
#endif
                            burst_store(a.g, nbytes, i, as_f4(gq[b][t][p]));
                            burst_store(a.v, nbytes, i, as_f4(vq[b][t][p]));
                            if (kHasM) burst_store(a.m, nbytes, i, as_f4(mq[b][t][p]));
                        }
                    }
                }
            };
            using B0 = std::integral_constant<int, 0>;
            using B1 = std::integral_constant<int, 1>;
#if FA_LATE_DEBUG == 1   /* timing experiment: no epilogue at all */
            if (a.n == 0xFFFFFFFFu)
#endif
            fetch(j_lo, B0{});
#pragma unroll 1
            for (int jj = j_lo; jj <= j && FA_LATE_DEBUG != 1; jj += 2 * B) {
                if (jj + B <= j) fetch(jj + B, B1{});
                finish(jj, B0{});
                if (jj + B <= j) {
                    if (jj + 2 * B <= j) fetch(jj + 2 * B, B0{});
                    finish(jj + B, B1{});
                }
            }
            if (phased) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain before the next phase's loads
        }
    }
    if (phased) grid_meet_leave(meet, meetings);
}

// ---------------------------------------------------------------------------------------------
// Plain mean over clients (entries that bypass the weighted step).

template <typename T>
struct MeanArgs {
    const T* x[FA_KCHUNK];
    T* g;
    const T* sum_in;  // partial sums of earlier chunks (NULL: 0) -- lives in g between chunks
    unsigned n;
    int K;       // clients in this launch
    int last;    // 1: divide
    float Kall;  // total client count as fp32
};


// The client values of an element are loaded kMeanBatch at a time (independent loads in
// flight) and then added in client order; with few elements the kernel is latency-bound, so the
// batch depth, not the thread count, sets its duration.
constexpr int kMeanBatch = 16;

template <typename T>
__global__ __launch_bounds__(kThreads) void mean_kernel(const MeanArgs<T> a) {
    for (unsigned i = blockIdx.x * kThreads + threadIdx.x; i < a.n; i += gridDim.x * kThreads) {
        T s = a.sum_in ? a.sum_in[i] : static_cast<T>(0);
        int k = 0;
        for (; k + kMeanBatch <= a.K; k += kMeanBatch) {
            T x[kMeanBatch];
#pragma unroll
            for (int u = 0; u < kMeanBatch; ++u) x[u] = __builtin_nontemporal_load(a.x[k + u] + i);
#pragma unroll
            for (int u = 0; u < kMeanBatch; ++u) s = s + x[u];
        }
        for (; k < a.K; ++k) s = s + __builtin_nontemporal_load(a.x[k] + i);
        a.g[i] = a.last ? mean_finish(s, a.Kall) : s;
    }
}

// ---------------------------------------------------------------------------------------------
// Weighted mixes of K client vectors that do not subtract g first (FedCompass / FedQueue).

struct MixArgs {
    const float* x[FA_KCHUNK];
    float w[FA_KCHUNK];
    float* g;
    const float* acc_in;  // partial sum of earlier client chunks (NULL: 0); modes 0 / 1
    float* acc_out;       // where a non-final chunk leaves its partial sum
    unsigned n;           // elements of width VEC
    int K;
    int mode;             // FA_MIX_*
    int last;
    float coeff;
};

// The combine mode is a template parameter (no run-time branches in the client loop).
template <int VEC, int VPT, int U, int MODE>
__global__ __launch_bounds__(kThreads) void mix_kernel(const MixArgs a) {
    constexpr unsigned kTile = kThreads * VPT;
    const unsigned tiles = (a.n + kTile - 1) / kTile;
    // the reduce kernel's XCD-contiguous tile order: inside windows of 8 * R tiles each XCD takes R
    // consecutive tiles instead of every eighth one
    constexpr unsigned R = 128u;
    for (unsigned t0 = blockIdx.x; t0 < tiles; t0 += gridDim.x) {
        unsigned t = t0;
        const unsigned win = t0 & ~(8u * R - 1u);
        if (VEC == 4 && win + 8u * R <= tiles) t = win + (t0 & 7u) * R + ((t0 >> 3) & (R - 1u));
        unsigned idx[VPT];
        bool ok[VPT];
#pragma unroll
        for (int p = 0; p < VPT; ++p) {
            unsigned i = t * kTile + p * kThreads + threadIdx.x;
            ok[p] = i < a.n;
            idx[p] = ok[p] ? i : a.n - 1;
        }
        Pack<VEC> acc[VPT], g[VPT];
#pragma unroll
        for (int p = 0; p < VPT; ++p) {
            if (MODE != FA_MIX_REPLACE && MODE != FA_MIX_SUM_SET) g[p] = load_state<VEC>(a.g, idx[p]);
            if (MODE == FA_MIX_SUBTRACT) {
                acc[p] = g[p];
            } else if (a.acc_in) {
                acc[p] = load_state<VEC>(a.acc_in, idx[p]);
            } else {
                // -0 + y == y for every y, so the single-arrival form adds x*w to nothing
                const float zero = (MODE >= FA_MIX_CONVEX_ONE) ? -0.0f : 0.0f;
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[p].e[j] = zero;
            }
        }
        auto fold = [&](const Pack<VEC>& x, int p, float wk) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                float xw = x.e[j] * wk;
                acc[p].e[j] = (MODE == FA_MIX_SUBTRACT) ? acc[p].e[j] - xw : acc[p].e[j] + xw;
            }
        };
        // B clients: all their loads, then the arithmetic in client order
        auto batch = [&](auto tag, int k0) {
            constexpr int B = decltype(tag)::value;
            Pack<VEC> xv[B][VPT];
#pragma unroll
            for (int u = 0; u < B; ++u)
#pragma unroll
                for (int p = 0; p < VPT; ++p) xv[u][p] = load_stream<VEC, true>(a.x[k0 + u], idx[p]);
#pragma unroll
            for (int u = 0; u < B; ++u)
#pragma unroll
                for (int p = 0; p < VPT; ++p) fold(xv[u][p], p, a.w[k0 + u]);
        };
        // whole batches of U clients, then the rest in batches of 8 / 4 / 2 / 1 (as in fold_tile)
        int k = 0;
        for (; k + U <= a.K; k += U) batch(std::integral_constant<int, U>{}, k);
        if constexpr (U > 8) {
            if (k + 8 <= a.K) { batch(std::integral_constant<int, 8>{}, k); k += 8; }
        }
        if constexpr (U > 4) {
            if (k + 4 <= a.K) { batch(std::integral_constant<int, 4>{}, k); k += 4; }
        }
        if (k + 2 <= a.K) { batch(std::integral_constant<int, 2>{}, k); k += 2; }
        if (k < a.K) batch(std::integral_constant<int, 1>{}, k);
#pragma unroll
        for (int p = 0; p < VPT; ++p) {
            if (!ok[p]) continue;
            if (!a.last && MODE != FA_MIX_SUBTRACT) {
                store_result(a.acc_out, idx[p], acc[p], 1);
                continue;
            }
            if ((MODE == FA_MIX_CONVEX || MODE == FA_MIX_CONVEX_ONE) && a.last) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    float gt = g[p].e[j] * a.coeff;
                    acc[p].e[j] = gt + acc[p].e[j];
                }
            }
            if (MODE == FA_MIX_SUM_ADD && a.last) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[p].e[j] = g[p].e[j] + acc[p].e[j];
            }
            store_result(a.g, idx[p], acc[p], 1);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Elementwise kernels: apply, FedAsync mix, FedBuff accumulate / flush, fills.

template <int VEC>
__global__ __launch_bounds__(kThreads) void apply_kernel(float* g, const float* step, unsigned n) {
    for (unsigned i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        Pack<VEC> gv = load_state<VEC>(g, i), sv = load_stream<VEC>(step, i);
#pragma unroll
        for (int j = 0; j < VEC; ++j) gv.e[j] = gv.e[j] + sv.e[j];
        store_state<VEC>(g, i, gv);
    }
}

// MODE 0: model-based, 1: gradient-based "0 - a*x", 2: gradient-based "a * (-x)".
template <int MODE>
__device__ __forceinline__ float async_step(float x, float g, float alpha) {
    if (MODE == 0) {
        float d = x - g;
        return alpha * d;
    }
    if (MODE == 1) {
        float ax = alpha * x;
        return 0.0f - ax;
    }
    float nx = -x;
    return alpha * nx;
}

// The arrival kernels move 2-3 streams in and 1-2 out per element and nothing else; each thread
// keeps kArrivalVpt vectors of every input stream in flight and results leave write-through.
#ifndef FA_ARRIVAL_VPT
#define FA_ARRIVAL_VPT 4
#endif
constexpr int kArrivalVpt = FA_ARRIVAL_VPT;

template <int VEC>
__device__ __forceinline__ void arrival_indices(unsigned t, unsigned n, unsigned (&idx)[VEC == 4 ? kArrivalVpt : 1],
                                                bool (&ok)[VEC == 4 ? kArrivalVpt : 1]) {
    constexpr int VPT = VEC == 4 ? kArrivalVpt : 1;
#pragma unroll
    for (int p = 0; p < VPT; ++p) {
        const unsigned i = t * (kThreads * VPT) + p * kThreads + threadIdx.x;
        ok[p] = i < n;
        idx[p] = ok[p] ? i : n - 1;
    }
}

template <int VEC, int MODE>
__global__ __launch_bounds__(kThreads) void async_mix_kernel(const float* x, float alpha, float* g,
                                                           float* step_out, unsigned n, int wt) {
    constexpr int VPT = VEC == 4 ? kArrivalVpt : 1;
    const unsigned tiles = (n + kThreads * VPT - 1) / (kThreads * VPT);
    for (unsigned t = blockIdx.x; t < tiles; t += gridDim.x) {
        unsigned idx[VPT];
        bool ok[VPT];
        arrival_indices<VEC>(t, n, idx, ok);
        Pack<VEC> xv[VPT], gv[VPT], sv[VPT];
#pragma unroll
        for (int p = 0; p < VPT; ++p) {
            xv[p] = load_stream<VEC>(x, idx[p]);
            gv[p] = load_stream<VEC>(g, idx[p]);   // every stream here is read once
        }
#pragma unroll
        for (int p = 0; p < VPT; ++p) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                sv[p].e[j] = async_step<MODE>(xv[p].e[j], gv[p].e[j], alpha);
                gv[p].e[j] = gv[p].e[j] + sv[p].e[j];
            }
            if (ok[p]) {
                if (step_out) store_result(step_out, idx[p], sv[p], wt);
                store_result(g, idx[p], gv[p], wt);
            }
        }
    }
}

// FLUSH: also g = g + step after the accumulate.
template <int VEC, int MODE, bool FLUSH>
__global__ __launch_bounds__(kThreads) void buff_accum_kernel(const float* x, float alpha, float* g,
                                                            float* step, int first, unsigned n, int wt) {
    constexpr int VPT = VEC == 4 ? kArrivalVpt : 1;
    const unsigned tiles = (n + kThreads * VPT - 1) / (kThreads * VPT);
    for (unsigned t = blockIdx.x; t < tiles; t += gridDim.x) {
        unsigned idx[VPT];
        bool ok[VPT];
        arrival_indices<VEC>(t, n, idx, ok);
        Pack<VEC> xv[VPT], gv[VPT], sv[VPT];
#pragma unroll
        for (int p = 0; p < VPT; ++p) {
            xv[p] = load_stream<VEC>(x, idx[p]);
            gv[p] = load_stream<VEC>(g, idx[p]);     // streamed like x: +3 % over cached loads
            if (first) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) sv[p].e[j] = 0.0f;
            } else {
                sv[p] = load_stream<VEC>(step, idx[p]);
            }
        }
#pragma unroll
        for (int p = 0; p < VPT; ++p) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                float d = async_step<MODE>(xv[p].e[j], gv[p].e[j], alpha);
                sv[p].e[j] = sv[p].e[j] + d;
                if (FLUSH) gv[p].e[j] = gv[p].e[j] + sv[p].e[j];
            }
            if (ok[p]) {
                store_result(step, idx[p], sv[p], wt);
                if (FLUSH) store_result(g, idx[p], gv[p], wt);
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void accum_raw_kernel(const T* x, T* step, int first,
                                                           unsigned n) {
    for (unsigned i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        T s = first ? static_cast<T>(0) : step[i];
        step[i] = s + x[i];
    }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void flush_mean_kernel(T* g, const T* step, float kf,
                                                            unsigned n) {
    for (unsigned i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads)
        g[i] = mean_finish(step[i], kf);
}

// Server-optimizer epilogue alone: step (in/out), m, v updated in place.
template <int OPT, int VEC>
__global__ __launch_bounds__(kThreads) void opt_step_kernel(float* step, float* m, float* v,
                                                          Hyper h, unsigned n) {
    for (unsigned i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        Pack<VEC> sv = load_state<VEC>(step, i), vv = load_state<VEC>(v, i), mv;
        if (OPT != FA_OPT_AVGM) mv = load_state<VEC>(m, i);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float mj = (OPT != FA_OPT_AVGM) ? mv.e[j] : 0.0f;
            sv.e[j] = optimizer_step<OPT>(sv.e[j], mj, vv.e[j], h);
            if (OPT != FA_OPT_AVGM) mv.e[j] = mj;
        }
        store_state<VEC>(step, i, sv);
        store_state<VEC>(v, i, vv);
        if (OPT != FA_OPT_AVGM) store_state<VEC>(m, i, mv);
    }
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void fill_kernel(float* p, float value, unsigned n) {
    for (unsigned i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        Pack<VEC> r;
#pragma unroll
        for (int j = 0; j < VEC; ++j) r.e[j] = value;
        store_state<VEC>(p, i, r);
    }
}

__device__ __forceinline__ float synth_noise(uint64_t seed, uint64_t i) {
    uint64_t z = seed * 0x9E3779B97F4A7C15ull + i;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    int s = static_cast<int>(z & 0xFFFF) + static_cast<int>((z >> 16) & 0xFFFF) +
            static_cast<int>((z >> 32) & 0xFFFF) + static_cast<int>((z >> 48) & 0xFFFF);
    return static_cast<float>(s - 131070) * 2.6428998e-05f;
}

__global__ __launch_bounds__(kThreads) void synth_kernel(float* p, const float* base, unsigned n,
                                                       uint64_t seed, uint64_t first, float scale) {
    for (unsigned i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        float t = scale * synth_noise(seed, first + i);
        p[i] = base ? base[i] + t : t;
    }
}

// ---------------------------------------------------------------------------------------------
// Host side.

inline hipStream_t as_stream(fa_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int hip_code(hipError_t e) { return e == hipSuccess ? FA_OK : FA_ERR_HIP - static_cast<int>(e); }

inline int launch_status() { return hip_code(hipGetLastError()); }

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

inline unsigned grid_for(size_t work_items) {
    size_t blocks = (work_items + kThreads - 1) / kThreads;
    if (blocks < 1) blocks = 1;
    if (blocks > static_cast<size_t>(kMaxBlocks)) blocks = kMaxBlocks;
    return static_cast<unsigned>(blocks);
}

// Largest element count one launch indexes with 32-bit vector indices (a multiple of 4).
constexpr size_t kSegment = size_t(1) << 29;

// Launch shape of the 128-bit reduce kernel.  The defaults are the measured best on MI355X;
// fa_tune() swaps them at run time, for the CALLING THREAD only, so a benchmark can compare
// shapes inside one process without touching what other threads' aggregators launch.
struct Shape {
    int vpt;         // 16-byte vectors per thread per tile: 1, 2 or 4
    int unroll;      // clients loaded ahead of the arithmetic: 4, 8 or 16
    int nt;          // non-temporal client loads
    int max_blocks;  // grid cap; tiles beyond it are walked grid-stride
    int stash;       // the persistent LDS-stash kernel takes vectors with at least this many
                     // tiles per CU (0: never, < 0: always); it pays off once every CU has
                     // several tiles of its own
};
thread_local Shape g_shape = {0, 8, 1, kMaxBlocks, 2};
// FA_ROUTE_* flags of the call being dispatched on this thread (fa_reduce_opt_ex); 0 otherwise.
thread_local unsigned g_call_flags = 0;
constexpr unsigned kLeanTileVectors = 1u << 20;   // 16 MiB of g
thread_local int g_write_through = 1;   // fa_tune's nt field: bit 1 set = plain result stores
thread_local int g_interleave = 1;      // fa_tune's nt field: bit 2 set = contiguous per-block ranges in the persistent kernel
thread_local int g_phase_tiles = 0;     // fa_tune's nt field, bits 8-12: tiles per phase of the persistent kernel (0: chosen per launch)
thread_local int g_sync = 2 | (1 << 2);   // fa_tune's nt field: chip-wide write phases of the persistent kernel -- default: the
                           // blocks meet before their bursts, in two groups (odd / even XCDs) half a phase
                           // apart; bit 4 set = no phases; bit 5 set = they also meet after the bursts have
                           // drained; bits 6-7: other groupings

// From this many clients on, the tile-per-block kernel on a long vector takes two vectors per
// thread instead of one.
constexpr int kTwoVectorClients = 16;

int device_cu_count() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        cus[dev] = n;
    }
    return cus[dev];
}

}  // namespace

// Meeting counters of the persistent kernel's launches: kSyncSlots slots of one 256-byte line
// each.  A slot belongs to a STREAM: launches on one stream run one after the other, so a slot
// never serves two launches at once as long as a device has at most kSyncSlots streams launching
// phased rounds (beyond that the oldest assignment is taken over and two streams may share a
// slot for a while: their meetings then end early or time out, results are unaffected).  Nothing
// is cleared between launches: every launch carries a tag, and the first of its blocks that
// finds another tag in the word restarts it (grid_meet_enter), so whatever an overlapping or
// aborted launch left behind is gone with the next launch.  (The name is a host-side handle; the
// per-device address comes from hipGetSymbolAddress.)
static __device__ __align__(256) unsigned sync_words[64 * 64];

namespace {

constexpr int kSyncSlots = 64;
constexpr int kMaxDevices = 64;
std::mutex g_symbol_mutex;
unsigned* g_sync_ptr[kMaxDevices] = {nullptr};
struct SyncSlots {
    hipStream_t stream[kSyncSlots];
    unsigned epoch[kSyncSlots];
    int used;       // assigned slots
    int victim;     // next assignment to take over once all are in use
};
SyncSlots g_sync_slots[kMaxDevices];

// Address of the current device's counters (NULL: unavailable, run without meetings).
unsigned* sync_words_of(int dev) {
    if (!g_sync_ptr[dev]) {
        unsigned* p = nullptr;
        if (hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(sync_words)) != hipSuccess || !p) {
            (void)hipGetLastError();
            return nullptr;
        }
        g_sync_ptr[dev] = p;
    }
    return g_sync_ptr[dev];
}

// The meeting counter of stream `st` on the current device and the tag of this launch in it.
unsigned* next_sync_word(hipStream_t st, unsigned* epoch) {
    int dev = -1;
    *epoch = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    std::lock_guard<std::mutex> hold(g_symbol_mutex);
    unsigned* base = sync_words_of(dev);
    if (!base) return nullptr;
    SyncSlots& t = g_sync_slots[dev];
    int slot = -1;
    for (int i = 0; i < t.used; ++i)
        if (t.stream[i] == st) { slot = i; break; }
    if (slot < 0) {
        slot = t.used < kSyncSlots ? t.used++ : (t.victim++ % kSyncSlots);
        t.stream[slot] = st;
    }
    t.epoch[slot] = t.epoch[slot] % (kEpochs - 1u) + 1u;   // 1..kEpochs-1 (the all-ones tag is never handed out)
    *epoch = t.epoch[slot];
    return base + 64 * slot;
}

}  // namespace

namespace fa_internal {
unsigned* meeting_word(hipStream_t st, unsigned* epoch) { return next_sync_word(st, epoch); }
int cu_count() { return device_cu_count(); }
}  // namespace fa_internal

namespace {

template <int OPT, int MODE>
int launch_stash_m(const ReduceArgs& a0, hipStream_t st) {
    ReduceArgs a = a0;
    a.sync_word = (a.sync && a.il) ? next_sync_word(st, &a.sync_epoch) : nullptr;
    hipLaunchKernelGGL((reduce_stash_kernel<OPT, MODE>), dim3(device_cu_count()), dim3(kThreads), 0, st, a);
    return launch_status();
}

template <int OPT>
int launch_park(const ReduceArgs& a0, hipStream_t st) {
    ReduceArgs a = a0;
    a.sync_word = a.sync ? next_sync_word(st, &a.sync_epoch) : nullptr;
    hipLaunchKernelGGL((reduce_park_kernel<OPT>), dim3(device_cu_count()), dim3(kThreads), 0, st, a);
    return launch_status();
}

template <int OPT>
int launch_late(const ReduceArgs& a0, hipStream_t st) {
    if constexpr (OPT == FA_OPT_NONE) {
        return FA_ERR_INVALID;
    } else {
        ReduceArgs a = a0;
        a.sync_word = a.sync ? next_sync_word(st, &a.sync_epoch) : nullptr;
        hipLaunchKernelGGL((reduce_late_kernel<OPT>), dim3(device_cu_count()), dim3(kThreads), 0, st, a);
        return launch_status();
    }
}

// Whether a launch takes the persistent kernel with the late epilogue: the final form (weighted
// step, epilogue, apply; no step output) of an optimizer round.  Its tile fold keeps only TWO
// clients x 4 vectors in flight per lane (FA_LATE_FOLD_AHEAD): at one wave per SIMD fewer live
// registers beat deeper batches -- ResNet-50 size, FedAdam, kernel time against 16 clients ahead:
// 24 / 32 / 64 clients -2.4 / -2.8 / -3.2 % (4 ahead: -2.4 / -2.7 / -1.6 %).  With that it is ahead of
// the tile-per-block kernel AND of the software-pipelined persistent kernel over most of the
// plane (same process, kernel time against the earlier default routing).  Two-moment optimizers:
// 4.5 M elements x 32 / 48 / 100 clients -1.1 / -5.0 / -7.6 %, 8 M x 24 / 32 / 64 -1.5 / -3.1 / -7.4 %,
// 11.7 M x 32 / 64 / 100 -5.9 / -6.0 / -6.7 %, 25.6 M x 20 / 40 / 64 / 100 -3.7 / -6.2 / -7.4 / -8.0 %,
// 47 M x 16 / 24 / 64 -3.3 / -3.5 / -2.3 % (40-48 clients level), 87 M x 24-64 -5.6...-7.4 %, 162.5 M x
// 24-32 -2.6...-2.9 %; behind at fewer clients (11.7-17 M x 16 +5...+6 %, x 20 +2 %).  FedAvgM: from
// 48 clients everywhere (-3.0...-8.4 %) and from 32 on vectors of >= 40 tiles per CU (47 M x 32 / 40
// -3.6 / -5.4 %, 87 M x 40 -4.7 %); below that its wide tile-per-block shape is ahead (8-25 M x
// 16-32 +1...+7 %).  With phases of three tiles the borders moved out a little more: FedAvgM 2.5-3.5 M x
// 48-100 -4.6...-7.5 % (against the pipelined persistent kernel), 11.7-25.6 M x 40 -1.3...-2 %, 47 M x 24
// -2.2 %; two-moment optimizers 35 M x 16 -3.2 %, 2.5 M x 32 -1.4 % (x 24 +5 %: stays).  fa_tune: stash ==
// -3 forces it at every length and client count, stash <= 0 otherwise keeps it off.
bool late_wins(int opt, int K, uint64_t tiles_per_cu) {
    if (opt == FA_OPT_AVGM)
        return (K >= 48 && tiles_per_cu >= 2) || (K >= 40 && tiles_per_cu >= 4) || (K >= 24 && tiles_per_cu >= 40);
    if (opt < FA_OPT_ADAM) return false;
    return (K >= 32 && tiles_per_cu >= 2) || (K >= 24 && tiles_per_cu >= 7) || (K >= 20 && tiles_per_cu >= 24) ||
           (K >= 16 && tiles_per_cu >= 32);
}
bool stash_wins(int opt, int K, uint64_t tiles_per_cu);
bool takes_late_path(int opt, int K, uint64_t nvec) {
    if (opt == FA_OPT_NONE || K > FA_KCHUNK) return false;
    if (g_call_flags & FA_ROUTE_NO_PERSISTENT) return false;
    if (g_shape.stash == -3) return true;
    if (g_shape.stash <= 0) return false;
    const uint64_t per_cu = static_cast<uint64_t>(device_cu_count()) * kStashTile;
    return late_wins(opt, K, nvec / per_cu);      // (ahead of the software-pipelined persistent kernel too)
}

// Whether a launch takes the wide-batch persistent kernel: plain FedAvg, final form (fa_tune:
// stash == -2 forces it at every length and client count, stash == 0 / -1 keep it off).  Its tile
// fold keeps only TWO clients x 4 vectors in flight per lane (FA_PARK_FOLD_AHEAD; at one wave per
// SIMD fewer live registers beat deeper batches: against 16 ahead 87 M elements x 32 clients -7 %,
// 162.5 M x 12 -2.5 %, 25.6 M x 64 -7.5 %, 11.7 M x 100 -3.6 %).  Same process, kernel time:
//  * from 6 clients on vectors of >= 72 tiles per CU, against the tile-per-block kernel (162.5 M
//    elements, 16 ahead, fraction of peak): 2 / 4 clients 0.78 / 0.75 vs 0.73 / 0.74 (behind), 5 / 6 /
//    7 / 8 / 12 / 14 clients 0.755 / 0.73 / 0.73 / 0.72 / 0.75 / 0.74 -> 0.76 / 0.755 / 0.76 / 0.77 /
//    0.80 / 0.80; the same at 87 M, 120 M and 325 M elements; +2 % at 75 M;
//  * from 40 clients on every vector of at least a tile per CU, against the pipelined persistent
//    kernel (>= 48 clients): 2.5-4.5 M x 48-128 -3.8...-7.9 %, 8 M -2.5...-4.3 %, 11.7 M -5.4...-5.8 %,
//    25.6 M -4.6...-5.1 %, 17 M and 47 M level (-1.4...+1.2 %); against the tile-per-block kernel at
//    40 clients -0.5...-6.6 %;
//  * from 24 clients on vectors of up to 4 tiles per CU (1.2-4.5 M x 24-40: -0.7...-6.6 % against
//    the tile-per-block kernel; 8-47 M x 24-32: 0...+4 %, left where they are).
constexpr uint64_t kParkMinTilesPerCu = 72;
constexpr int kParkMinClients = 6;
bool park_wins(int K, uint64_t tiles_per_cu) {
    if (tiles_per_cu >= kParkMinTilesPerCu) return K >= kParkMinClients;
    if (tiles_per_cu < 1) return false;
    return K >= 40 || (K >= 24 && tiles_per_cu <= 4);
}
bool takes_park_path(int opt, int K, uint64_t nvec, bool final_form) {
    if (opt != FA_OPT_NONE || !final_form || K > FA_KCHUNK) return false;
    if (g_call_flags & FA_ROUTE_NO_PERSISTENT) return false;
    if (g_shape.stash == -2) return true;
    if (g_shape.stash <= 0) return false;
    const uint64_t per_cu = static_cast<uint64_t>(device_cu_count()) * kStashTile;
    return park_wins(K, nvec / per_cu);
}

template <int OPT>
int launch_stash(const ReduceArgs& a, hipStream_t st) {
    if (!a.last) return launch_stash_m<FA_OPT_NONE, kOutStepOnly>(a, st);
    if (!a.apply) return launch_stash_m<FA_OPT_NONE, kOutStepOnly>(a, st);
    if (a.step_out) return launch_stash_m<OPT, kOutFinalStep>(a, st);
    return launch_stash_m<OPT, kOutFinal>(a, st);
}

template <int OPT, int VEC, int VPT, int U, bool NT>
int launch_reduce_s(const ReduceArgs& a, hipStream_t st) {
    size_t tiles = (static_cast<size_t>(a.n) + kThreads * VPT - 1) / (kThreads * VPT);
    size_t cap = static_cast<size_t>(g_shape.max_blocks > 0 ? g_shape.max_blocks : kMaxBlocks);
    unsigned grid = static_cast<unsigned>(tiles < 1 ? 1 : (tiles > cap ? cap : tiles)) + a.mt.blocks;
    hipLaunchKernelGGL((reduce_kernel<OPT, VEC, VPT, U, NT>), dim3(grid), dim3(kThreads), 0, st, a);
    return launch_status();
}

template <int OPT, int VPT, int U>
int launch_reduce_nt(const ReduceArgs& a, hipStream_t st) {
    return g_shape.nt ? launch_reduce_s<OPT, 4, VPT, U, true>(a, st)
                      : launch_reduce_s<OPT, 4, VPT, U, false>(a, st);
}

template <int OPT, int VPT>
int launch_reduce_u(const ReduceArgs& a, int unroll, hipStream_t st) {
    switch (unroll) {
        case 4: return launch_reduce_nt<OPT, VPT, 4>(a, st);
        case 16: return launch_reduce_nt<OPT, VPT, 16>(a, st);
        default: return launch_reduce_nt<OPT, VPT, 8>(a, st);
    }
}

// Whether a launch over `nvec` 16-byte vectors of K clients goes to the persistent kernel.  It
// wins where the client list is long: plain FedAvg and FedAvgM from ~48 clients (+1-3 % over the
// wide tile-per-block shape below at 48-100 clients; at 32-40 clients that shape is 2-4 % faster,
// below it more).  Under the two-moment optimizers three result vectors per tile leave LDS room for
// phases of three tiles only, and many resident waves hide their six state streams better than
// one wave per SIMD at short client lists.  With the phases of odd and even XCDs half a phase
// apart it is ahead from 40 clients on vectors of >= 8 tiles per CU (measured against the
// tile-per-block kernel: +3 % at 25 M elements x 40-56 clients, +2.5 % / +6 % at 12 M x 48 / 100,
// +5-6 % at 87 M x 32-48), from 32 clients on vectors of >= 32 tiles per CU (+6-7 % at 35-60 M x
// 32-36) and from 24 clients on vectors of >= 40 tiles per CU (+3-4 % at 47-60 M x 24-28, +6 % at
// 87 M x 24, +4-5 % at 162 M x 32); below that the tile-per-block kernel wins or ties (25 M x
// 24-28: +2-5 %, 25 M x 32-36 and 35 M x 24-28: within 1.5 %, 60 M x 20: +4 %, 12 M x 24-32: +2-8 %,
// 4.5 M x 32: +8 %).
bool stash_wins(int opt, int K, uint64_t tiles_per_cu) {
    if (opt <= FA_OPT_AVGM) return K >= 48 && tiles_per_cu >= 1;
    return (K >= 40 && tiles_per_cu >= 8) || (K >= 32 && tiles_per_cu >= 32) || (K >= 24 && tiles_per_cu >= 40);
}
bool takes_stash_path(int opt, int K, uint64_t nvec) {
    if (g_call_flags & FA_ROUTE_NO_PERSISTENT) return false;
    if (g_shape.stash < 0) return K > kStashUnroll;
    const uint64_t per_cu = static_cast<uint64_t>(device_cu_count()) * kStashTile;
    return g_shape.stash > 0 && nvec >= per_cu * static_cast<uint64_t>(g_shape.stash) &&
           stash_wins(opt, K, nvec / per_cu);
}

template <int OPT, int VEC>
int launch_reduce_t(const ReduceArgs& a, hipStream_t st) {
    if (VEC == 1) return launch_reduce_s<OPT, 1, 1, 8, true>(a, st);
    {
        const bool step_only = (!a.last || !a.apply) && a.step_out;   // no optimizer in these forms
        const bool final_form = a.last && a.apply && !a.step_out;
        if ((step_only || (OPT == FA_OPT_NONE && final_form)) &&
            takes_park_path(FA_OPT_NONE, a.K, a.n, true))
            return launch_park<FA_OPT_NONE>(a, st);
    }
    if (OPT != FA_OPT_NONE && a.last && a.apply && !a.step_out && takes_late_path(OPT, a.K, a.n))
        return launch_late<OPT>(a, st);
    if (takes_stash_path(OPT, a.K, a.n)) return launch_stash<OPT>(a, st);
    int vpt = g_shape.vpt, unroll = g_shape.unroll;
    if (vpt == 0) {
        // Long vectors: few vectors per thread and 16 clients ahead keep more waves resident to
        // cover the state streams and the optimizer epilogue (measured +10-20 % for the
        // two-moment optimizers, +2-4 % for plain FedAvg at K = 2..15); short vectors are launch
        // bound and do better with fewer, fatter blocks.
        const bool lean = a.n >= kLeanTileVectors;
        vpt = !lean ? 4 : (a.K >= kTwoVectorClients ? 2 : 1);
        unroll = lean ? 16 : 8;
        // ... except under the two-moment optimizers, whose six state streams want many small
        // blocks at every size (measured on a 0.58 M-element vector with 32 clients: 21 us with
        // one vector per thread, 40 us with four; 3-5 % at 2 M elements)
        if (!lean && OPT >= FA_OPT_ADAM) vpt = 1;
        // Plain FedAvg and FedAvgM from ~10 clients: the WIDE shape -- four vectors per thread and
        // 16 clients ahead, 64 loads of 16 bytes in flight per lane at one wave per SIMD.  With
        // nothing but streaming to do, bytes in flight per wave beat resident waves: measured
        // against the shapes above +5-8 % at 16-32 clients on 25-162 M elements, +2-5 % on 2-12 M
        // elements, +3-5 % for FedAvgM; below 10 clients (or 16 on vectors under 16 M elements)
        // the shapes above stay ahead.  (Under the two-moment optimizers it is within +-1.5 % of the
        // shapes above at 16-32 clients, and fetching m / v before the client loop instead of after
        // it changes nothing either: their result writes, not the launch shape, bound them.)
        if (OPT <= FA_OPT_AVGM && (a.K >= 16 || (a.K >= 10 && a.n >= 4u * kLeanTileVectors))) {
            vpt = 4;
            unroll = 16;
        }
    }
    switch (vpt) {
        case 1: return launch_reduce_u<OPT, 1>(a, unroll, st);
        case 4: return launch_reduce_u<OPT, 4>(a, unroll, st);
        default: return launch_reduce_u<OPT, 2>(a, unroll, st);
    }
}

template <int VEC>
int launch_reduce(int opt, const ReduceArgs& a, hipStream_t st) {
    switch (opt) {
        case FA_OPT_NONE: return launch_reduce_t<FA_OPT_NONE, VEC>(a, st);
        case FA_OPT_AVGM: return launch_reduce_t<FA_OPT_AVGM, VEC>(a, st);
        case FA_OPT_ADAM: return launch_reduce_t<FA_OPT_ADAM, VEC>(a, st);
        case FA_OPT_YOGI: return launch_reduce_t<FA_OPT_YOGI, VEC>(a, st);
        case FA_OPT_ADAGRAD: return launch_reduce_t<FA_OPT_ADAGRAD, VEC>(a, st);
    }
    return FA_ERR_INVALID;
}

struct ReduceCall {
    const float* const* x;
    const double* w;
    int K;
    float* g;          // written when apply
    const float* g_ro; // read when !apply
    float* m;
    float* v;
    size_t n;
    int opt;
    Hyper h;
    float* step_out;
    float* step_ws;
    int apply;
    const MeanTail* tail = nullptr;   // plain-mean entries to fold into the launch, if one can take them
    bool* tail_done = nullptr;        // set when a launch took them
    int k_ready = 0;                  // 0 < k_ready < K: only the first k_ready clients' rows have landed;
    hipEvent_t rest_event = nullptr;  // ... the launches that read the others wait for this event
    int k_done = 0;                   // clients 0..k_done-1 were folded by an earlier call: the carry holds their partial step
};

// Where the launches of one call cut the client list: every FA_KCHUNK clients, and at
// `k_ready` when the caller says that only the first k_ready clients' rows have landed yet.
int next_cut(const ReduceCall& c, int k0) {
    int k1 = (c.K - k0 < FA_KCHUNK) ? c.K : k0 + FA_KCHUNK;
    if (k0 < c.k_ready && c.k_ready < k1) k1 = c.k_ready;
    return k1;
}

// Runs [off, off+len) of the flat axis: the client chunks carry the partial step through
// `carry` (step_out when given, else step_ws).
int reduce_range(const ReduceCall& c, size_t off, size_t len, bool vec, hipStream_t st) {
    float* carry = c.step_out ? c.step_out : c.step_ws;
    for (int k0 = c.k_done; k0 < c.K;) {
        const int k1 = next_cut(c, k0);
        if (k0 > 0 && k0 == c.k_ready && c.rest_event) {
            // the remaining clients' rows are still on their way: everything launched so far
            // overlaps their transfer, what follows waits for it
            hipError_t e = hipStreamWaitEvent(st, c.rest_event, 0);
            if (e != hipSuccess) return hip_code(e);
        }
        ReduceArgs a;
        memset(&a, 0, sizeof(a));
        a.K = k1 - k0;
        for (int k = 0; k < a.K; ++k) {
            a.x[k] = c.x[k0 + k] + off;
            a.w[k] = static_cast<float>(c.w[k0 + k]);
        }
        a.g = c.g ? c.g + off : nullptr;
        a.g_ro = c.g_ro ? c.g_ro + off : nullptr;
        a.m = c.m ? c.m + off : nullptr;
        a.v = c.v ? c.v + off : nullptr;
        a.last = (k1 == c.K);
        a.apply = c.apply;
        a.step_in = (k0 == 0) ? nullptr : carry + off;
        a.step_out = a.last ? (c.step_out ? c.step_out + off : nullptr) : carry + off;
        a.h = c.h;
        a.wt = g_write_through;
        a.il = g_interleave;
        a.sync = g_sync;
        a.phase_tiles = g_phase_tiles;
        a.n = static_cast<unsigned>(vec ? len / 4 : len);
        if (c.tail && c.tail_done && !*c.tail_done && k0 == 0 && k1 == c.K) {
            // the pointers of the tail are relative to x[k] of the whole vector
            a.mt = *c.tail;
            for (int sgm = 0; sgm < kTailSegments; ++sgm)
                a.mt.off[sgm] -= static_cast<long long>(off * sizeof(float));
            a.mt.blocks = tail_blocks(a.mt.total);
            *c.tail_done = true;
        }
        int rc = vec ? launch_reduce<4>(c.opt, a, st) : launch_reduce<1>(c.opt, a, st);
        if (rc != FA_OK) return rc;
        k0 = k1;
    }
    return FA_OK;
}

int reduce_dispatch(const ReduceCall& c, hipStream_t st) {
    if (!c.x || !c.w || c.K <= 0) return FA_ERR_INVALID;
    if (c.apply ? !c.g : (!c.g_ro || !c.step_out)) return FA_ERR_INVALID;
    if (c.opt < FA_OPT_NONE || c.opt > FA_OPT_ADAGRAD) return FA_ERR_INVALID;
    if (c.opt != FA_OPT_NONE && !c.v) return FA_ERR_INVALID;
    if (c.opt >= FA_OPT_ADAM && !c.m) return FA_ERR_INVALID;
    if (c.k_done < 0 || c.k_done >= c.K) return FA_ERR_INVALID;
    if ((c.k_done > 0 || next_cut(c, 0) < c.K) && !c.step_out && !c.step_ws) return FA_ERR_WORKSPACE;
    if (c.n == 0) return FA_OK;
    const void* gp = c.apply ? static_cast<const void*>(c.g) : static_cast<const void*>(c.g_ro);
    bool a4 = aligned_to(gp, 4), a16 = aligned16(gp);
    const void* others[] = {c.m, c.v, c.step_out, c.step_ws};
    for (const void* p : others)
        if (p) { a4 = a4 && aligned_to(p, 4); a16 = a16 && aligned16(p); }
    for (int k = 0; k < c.K; ++k) {
        if (!c.x[k]) return FA_ERR_INVALID;
        a4 = a4 && aligned_to(c.x[k], 4);
        a16 = a16 && aligned16(c.x[k]);
    }
    if (!a4) return FA_ERR_ALIGN;
    for (size_t off = 0; off < c.n; off += kSegment) {
        size_t len = (c.n - off < kSegment) ? c.n - off : kSegment;
        size_t body = a16 ? (len & ~size_t(3)) : 0;
        if (body) {
            int rc = reduce_range(c, off, body, true, st);
            if (rc != FA_OK) return rc;
        }
        if (len > body) {
            int rc = reduce_range(c, off + body, len - body, false, st);
            if (rc != FA_OK) return rc;
        }
    }
    return FA_OK;
}

Hyper make_hyper(double lr, double beta1, double beta2, double tau) {
    Hyper h;
    h.lr = static_cast<float>(lr);
    h.b1 = static_cast<float>(beta1);
    h.omb1 = static_cast<float>(1.0 - beta1);
    h.b2 = static_cast<float>(beta2);
    h.omb2 = static_cast<float>(1.0 - beta2);
    h.tau = static_cast<float>(tau);
    return h;
}

template <typename T>
int mean_dispatch(const T* const* x, int K, T* g, size_t n, hipStream_t st) {
    if (!x || !g || K <= 0) return FA_ERR_INVALID;
    if (!aligned_to(g, sizeof(T))) return FA_ERR_ALIGN;
    for (int k = 0; k < K; ++k) {
        if (!x[k]) return FA_ERR_INVALID;
        if (!aligned_to(x[k], sizeof(T))) return FA_ERR_ALIGN;
    }
    if (n == 0) return FA_OK;
    const int chunks = (K + FA_KCHUNK - 1) / FA_KCHUNK;
    for (size_t off = 0; off < n; off += kSegment) {
        size_t len = (n - off < kSegment) ? n - off : kSegment;
        for (int ch = 0; ch < chunks; ++ch) {
            MeanArgs<T> a;
            memset(&a, 0, sizeof(a));
            const int k0 = ch * FA_KCHUNK;
            a.K = (K - k0 < FA_KCHUNK) ? K - k0 : FA_KCHUNK;
            for (int k = 0; k < a.K; ++k) a.x[k] = x[k0 + k] + off;
            a.g = g + off;
            a.sum_in = (ch == 0) ? nullptr : g + off;
            a.n = static_cast<unsigned>(len);
            a.last = (ch == chunks - 1);
            a.Kall = static_cast<float>(K);
            hipLaunchKernelGGL((mean_kernel<T>), dim3(grid_for(len)), dim3(kThreads), 0, st, a);
            int rc = launch_status();
            if (rc != FA_OK) return rc;
        }
    }
    return FA_OK;
}

// Splits [0, n) into <= kSegment pieces and, when every pointer is 16-byte aligned, each piece
// into a 128-bit body and a scalar tail; `fn(off, count, vec)` launches one kernel.
template <typename F>
int for_each_range(size_t n, bool a16, F fn) {
    for (size_t off = 0; off < n; off += kSegment) {
        size_t len = (n - off < kSegment) ? n - off : kSegment;
        size_t body = a16 ? (len & ~size_t(3)) : 0;
        if (body) {
            int rc = fn(off, body / 4, true);
            if (rc != FA_OK) return rc;
        }
        if (len > body) {
            int rc = fn(off + body, len - body, false);
            if (rc != FA_OK) return rc;
        }
    }
    return FA_OK;
}

template <typename T>
int accum_raw_dispatch(const T* x, T* step, int first, size_t n, hipStream_t st) {
    if (!x || !step) return FA_ERR_INVALID;
    if (!aligned_to(x, sizeof(T)) || !aligned_to(step, sizeof(T))) return FA_ERR_ALIGN;
    for (size_t off = 0; off < n; off += kSegment) {
        size_t len = (n - off < kSegment) ? n - off : kSegment;
        hipLaunchKernelGGL((accum_raw_kernel<T>), dim3(grid_for(len)), dim3(kThreads), 0, st,
                           x + off, step + off, first, static_cast<unsigned>(len));
        int rc = launch_status();
        if (rc != FA_OK) return rc;
    }
    return FA_OK;
}

template <typename T>
int flush_mean_dispatch(T* g, const T* step, int K, size_t n, hipStream_t st) {
    if (!g || !step || K <= 0) return FA_ERR_INVALID;
    if (!aligned_to(g, sizeof(T)) || !aligned_to(step, sizeof(T))) return FA_ERR_ALIGN;
    for (size_t off = 0; off < n; off += kSegment) {
        size_t len = (n - off < kSegment) ? n - off : kSegment;
        hipLaunchKernelGGL((flush_mean_kernel<T>), dim3(grid_for(len)), dim3(kThreads), 0, st,
                           g + off, step + off, static_cast<float>(K), static_cast<unsigned>(len));
        int rc = launch_status();
        if (rc != FA_OK) return rc;
    }
    return FA_OK;
}

// Grid of the 128-bit arrival kernels: one block per tile of kArrivalVpt vectors per thread.
unsigned arrival_grid(size_t nvec) {
    size_t tiles = (nvec + kThreads * kArrivalVpt - 1) / (kThreads * kArrivalVpt);
    size_t cap = static_cast<size_t>(g_shape.max_blocks > 0 ? g_shape.max_blocks : kMaxBlocks);
    return static_cast<unsigned>(tiles < 1 ? 1 : (tiles > cap ? cap : tiles));
}

template <int MODE>
int async_mix_mode(const float* x, float alpha, float* g, float* step_out, size_t n, bool a16,
                   hipStream_t st) {
    return for_each_range(n, a16, [&](size_t off, size_t cnt, bool vec) {
        float* so = step_out ? step_out + off : nullptr;
        if (vec)
            hipLaunchKernelGGL((async_mix_kernel<4, MODE>), dim3(arrival_grid(cnt)), dim3(kThreads), 0,
                               st, x + off, alpha, g + off, so, static_cast<unsigned>(cnt), g_write_through);
        else
            hipLaunchKernelGGL((async_mix_kernel<1, MODE>), dim3(grid_for(cnt)), dim3(kThreads), 0,
                               st, x + off, alpha, g + off, so, static_cast<unsigned>(cnt), 0);
        return launch_status();
    });
}

template <int MODE, bool FLUSH>
int buff_accum_mode(const float* x, float alpha, float* g, float* step, int first, size_t n,
                    bool a16, hipStream_t st) {
    return for_each_range(n, a16, [&](size_t off, size_t cnt, bool vec) {
        if (vec)
            hipLaunchKernelGGL((buff_accum_kernel<4, MODE, FLUSH>), dim3(arrival_grid(cnt)),
                               dim3(kThreads), 0, st, x + off, alpha, g + off, step + off, first,
                               static_cast<unsigned>(cnt), g_write_through);
        else
            hipLaunchKernelGGL((buff_accum_kernel<1, MODE, FLUSH>), dim3(grid_for(cnt)),
                               dim3(kThreads), 0, st, x + off, alpha, g + off, step + off, first,
                               static_cast<unsigned>(cnt), 0);
        return launch_status();
    });
}

template <bool FLUSH>
int buff_accum_dispatch(const float* x, double alpha_t, int gradient_based, float* g, float* step,
                        int first, size_t n, hipStream_t st) {
    if (!x || !g || !step) return FA_ERR_INVALID;
    if (!aligned_to(x, 4) || !aligned_to(g, 4) || !aligned_to(step, 4)) return FA_ERR_ALIGN;
    const bool a16 = aligned16(x) && aligned16(g) && aligned16(step);
    const float alpha = static_cast<float>(alpha_t);
    if (gradient_based) return buff_accum_mode<2, FLUSH>(x, alpha, g, step, first, n, a16, st);
    return buff_accum_mode<0, FLUSH>(x, alpha, g, step, first, n, a16, st);
}

// ---------------------------------------------------------------------------------------------
// Library-owned carry workspace of fa_reduce_opt_static: device memory every GPU the library runs
// on holds until the process ends.  In host code `scratch` is only a handle of the symbol; the
// device address comes from hipGetSymbolAddress, once per device, and is what the kernels get.

}  // namespace

static __device__ __align__(256) float scratch[FA_STATIC_WS_FLOATS];

namespace {

float* g_scratch_ptr[kMaxDevices] = {nullptr};

// Address of the current device's copy of `scratch` (looked up on the first call per device).
int scratch_for_current_device(float** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_code(e);
    if (dev < 0 || dev >= kMaxDevices) return FA_ERR_INVALID;
    std::lock_guard<std::mutex> hold(g_symbol_mutex);
    if (!g_scratch_ptr[dev]) {
        float* p = nullptr;
        e = hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(scratch));
        if (e != hipSuccess) return hip_code(e);
        if (!p) return FA_ERR_INVALID;
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return FA_OK;
}

// Host threads that copy the parts of a pack job (fa_pack_host).  Started on first use, never
// joined (they end with the process); a process forked from one that had a pool starts its own.
// Between jobs a worker first spins on a ticket of its own for a short while -- the windows of
// one upload follow each other within microseconds, and waking a sleeping thread costs more
// than copying its share of a small window -- and then sleeps on a condition variable.
class PackPool {
public:
    static constexpr int kMaxWorkers = 15;

    explicit PackPool(int workers) {
        if (workers > kMaxWorkers) workers = kMaxWorkers;
        try {
            for (int j = 1; j <= workers; ++j) {
                std::thread([this, j] { loop(j); }).detach();
                workers_ = j;
            }
        } catch (...) {
            // the system refused a thread: the pool is the workers that did start
        }
    }
    int workers() const { return workers_; }
    // fn(0) on the caller, fn(1..parts-1) on the workers; returns when all are done.  One job
    // at a time (the caller holds g_pack_pool_busy).
    void run(int parts, const std::function<void(int)>& fn) {
        job_ = &fn;
        pending_.store(parts - 1, std::memory_order_relaxed);
        const uint64_t ticket = ++issued_;
        for (int j = 1; j < parts; ++j) slot_[j].ticket.store(ticket, std::memory_order_seq_cst);
        { std::lock_guard<std::mutex> hold(m_); }    // a worker between its last look and its sleep gets there first
        wake_.notify_all();
        fn(0);
        if (!spin_until([this] { return pending_.load(std::memory_order_acquire) == 0; }, kCallerSpinUs)) {
            std::unique_lock<std::mutex> hold(m_);
            done_.wait(hold, [this] { return pending_.load(std::memory_order_acquire) == 0; });
        }
        job_ = nullptr;
    }

private:
    static constexpr int kWorkerSpinUs = 150, kCallerSpinUs = 2000;
    struct alignas(64) Slot {
        std::atomic<uint64_t> ticket{0};   // number of the last job handed to this worker
    };

    template <typename Pred>
    static bool spin_until(Pred ready, int micros) {
        const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(micros);
        for (;;) {
            for (int i = 0; i < 64; ++i) {
                if (ready()) return true;
                __builtin_ia32_pause();
            }
            if (std::chrono::steady_clock::now() >= until) return ready();
        }
    }

    void loop(int j) {
        uint64_t seen = 0;
        auto handed = [&] { return slot_[j].ticket.load(std::memory_order_seq_cst) != seen; };
        for (;;) {
            if (!spin_until(handed, kWorkerSpinUs)) {
                std::unique_lock<std::mutex> hold(m_);
                wake_.wait(hold, handed);
            }
            seen = slot_[j].ticket.load(std::memory_order_seq_cst);
            // job_ was written before the ticket and stays put until every part has reported
            (*job_)(j);
            if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                { std::lock_guard<std::mutex> hold(m_); }
                done_.notify_one();
            }
        }
    }
    int workers_ = 0;
    std::mutex m_;
    std::condition_variable wake_, done_;
    const std::function<void(int)>* job_ = nullptr;
    uint64_t issued_ = 0;
    std::atomic<int> pending_{0};
    Slot slot_[kMaxWorkers + 1];
};

std::mutex g_pack_pool_busy;     // one job at a time
PackPool* g_pack_pool = nullptr;
pid_t g_pack_pool_pid = 0;

// Call with g_pack_pool_busy held.
PackPool* pack_pool() {
    const pid_t me = getpid();
    if (!g_pack_pool || g_pack_pool_pid != me) {
        g_pack_pool = new (std::nothrow) PackPool(15);   // (a parent's pool is left behind: its threads are not ours)
        g_pack_pool_pid = me;
    }
    return g_pack_pool;
}

}  // namespace

namespace {

template <int VEC, int VPT, int U>
int launch_mix_m(const MixArgs& a, unsigned grid, hipStream_t st) {
    switch (a.mode) {
        case FA_MIX_REPLACE: hipLaunchKernelGGL((mix_kernel<VEC, VPT, U, FA_MIX_REPLACE>), dim3(grid), dim3(kThreads), 0, st, a); break;
        case FA_MIX_CONVEX: hipLaunchKernelGGL((mix_kernel<VEC, VPT, U, FA_MIX_CONVEX>), dim3(grid), dim3(kThreads), 0, st, a); break;
        case FA_MIX_SUBTRACT: hipLaunchKernelGGL((mix_kernel<VEC, VPT, U, FA_MIX_SUBTRACT>), dim3(grid), dim3(kThreads), 0, st, a); break;
        case FA_MIX_CONVEX_ONE: hipLaunchKernelGGL((mix_kernel<VEC, VPT, U, FA_MIX_CONVEX_ONE>), dim3(grid), dim3(kThreads), 0, st, a); break;
        case FA_MIX_SUM_ADD: hipLaunchKernelGGL((mix_kernel<VEC, VPT, U, FA_MIX_SUM_ADD>), dim3(grid), dim3(kThreads), 0, st, a); break;
        case FA_MIX_SUM_SET: hipLaunchKernelGGL((mix_kernel<VEC, VPT, U, FA_MIX_SUM_SET>), dim3(grid), dim3(kThreads), 0, st, a); break;
        default: return FA_ERR_INVALID;
    }
    return launch_status();
}

// Launch shape of the mix kernel: 16 clients ahead, two vectors per thread from 8 clients on, one
// below.  Measured (fraction of peak, replace / convex; 4 vectors x 8 ahead, the shape before ->
// this rule): 86.6 M elements x 8 / 16 / 32 clients 0.74 / 0.72 / 0.73 -> 0.79 / 0.75 / 0.75, 25.6 M x
// 8 / 16 / 32 / 64 0.78 / 0.76 / 0.77 / 0.78 -> 0.84 / 0.79 / 0.79 / 0.80, 4.5 M x 4 / 16 / 64 0.67 / 0.77 /
// 0.78 -> 0.83 / 0.82 / 0.81.  The reduce kernel's wide shape (4 vectors x 16 ahead) does not carry
// over: 0.57-0.59 from 16 clients on, with the combine mode compiled in or not.
int launch_mix(const MixArgs& a, bool vec, hipStream_t st) {
    if (!vec) return launch_mix_m<1, 1, 8>(a, grid_for(a.n), st);
    auto grid = [&](unsigned vpt) {
        const size_t tiles = (static_cast<size_t>(a.n) + kThreads * vpt - 1) / (kThreads * vpt);
        return static_cast<unsigned>(tiles > static_cast<size_t>(kMaxBlocks) ? kMaxBlocks : (tiles < 1 ? 1 : tiles));
    };
    if (g_shape.vpt != 0) {   // fa_tune: a fixed shape (benchmark knob)
        const int u = g_shape.unroll;
        switch (g_shape.vpt) {
            case 1: return u == 16 ? launch_mix_m<4, 1, 16>(a, grid(1), st) : launch_mix_m<4, 1, 8>(a, grid(1), st);
            case 2: return u == 16 ? launch_mix_m<4, 2, 16>(a, grid(2), st) : launch_mix_m<4, 2, 8>(a, grid(2), st);
            default: return u == 16 ? launch_mix_m<4, 4, 16>(a, grid(4), st) : launch_mix_m<4, 4, 8>(a, grid(4), st);
        }
    }
    return a.K >= 8 ? launch_mix_m<4, 2, 16>(a, grid(2), st) : launch_mix_m<4, 1, 16>(a, grid(1), st);
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C ABI

extern "C" {

int fa_version(void) { return 100; }

int fa_kchunk(void) { return FA_KCHUNK; }

const char* fa_reduce_kernel_name(int K, size_t n, int opt) {
    const int k = K > FA_KCHUNK ? FA_KCHUNK : K;
    const uint64_t seg = n > kSegment ? kSegment : n;
    if (takes_park_path(opt, k, seg / 4, K <= FA_KCHUNK)) return "reduce_park_kernel";
    if (K <= FA_KCHUNK && takes_late_path(opt, k, seg / 4)) return "reduce_late_kernel";
    return takes_stash_path(opt, k, seg / 4) ? "reduce_stash_kernel" : "reduce_kernel";
}

int fa_tune(int vpt, int unroll, int nt, int max_blocks, int stash) {
    if (vpt != 0 && vpt != 1 && vpt != 2 && vpt != 4) return FA_ERR_INVALID;
    if (unroll != 4 && unroll != 8 && unroll != 16) return FA_ERR_INVALID;
    g_shape.vpt = vpt;
    g_shape.unroll = unroll;
    g_shape.nt = (nt & 1) ? 1 : 0;
    g_write_through = (nt & 2) ? 0 : 1;
    g_interleave = (nt & 4) ? 0 : 1;
    g_sync = (nt & 16) ? 0 : ((nt & 32) ? 3 : 2);
    if (g_sync) {
        // bits 6-7: groups of XCDs with staggered phases -- 0: the default (odd / even XCDs),
        // 1: one group, 2: four groups, 3: XCDs 0-3 / 4-7
        const int m = (nt >> 6) & 3;
        g_sync |= (m == 0 ? 1 : m == 1 ? 0 : m) << 2;
    }
    g_phase_tiles = (nt >> 8) & 31;   // experiment encoding: bits 4-5 of nt = sync mode as is
    g_shape.max_blocks = max_blocks > 0 ? max_blocks : kMaxBlocks;
    g_shape.stash = stash;
    return FA_OK;
}

const char* fa_error_string(int code) {
    switch (code) {
        case FA_OK: return "ok";
        case FA_ERR_INVALID: return "invalid argument";
        case FA_ERR_ALIGN: return "buffer not aligned to its element size";
        case FA_ERR_WORKSPACE: return "K > FA_KCHUNK needs a step workspace";
        case FA_ERR_TOO_LARGE: return "n exceeds FA_STATIC_WS_FLOATS, the library-owned workspace";
    }
    if (code <= FA_ERR_HIP) return hipGetErrorString(static_cast<hipError_t>(FA_ERR_HIP - code));
    return "unknown error";
}

int fa_reduce_opt(const float* const* x, const double* w, int K, float* g, float* m, float* v,
                  size_t n, int opt, double lr, double beta1, double beta2, double tau,
                  float* step_out, float* step_ws, fa_stream_t stream) {
    ReduceCall c{x, w, K, g, nullptr, m, v, n, opt, make_hyper(lr, beta1, beta2, tau),
                 step_out, step_ws, 1};
    return reduce_dispatch(c, as_stream(stream));
}

int fa_reduce_opt_ex(const float* const* x, const double* w, int K, float* g, float* m, float* v,
                     size_t n, int opt, double lr, double beta1, double beta2, double tau,
                     float* step_out, float* step_ws, unsigned flags, fa_stream_t stream) {
    if (flags & ~static_cast<unsigned>(FA_ROUTE_NO_PERSISTENT)) return FA_ERR_INVALID;
    ReduceCall c{x, w, K, g, nullptr, m, v, n, opt, make_hyper(lr, beta1, beta2, tau),
                 step_out, step_ws, 1};
    g_call_flags = flags;
    const int rc = reduce_dispatch(c, as_stream(stream));
    g_call_flags = 0;
    return rc;
}

int fa_debug_scribble_meeting_counters(fa_stream_t stream) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_code(e);
    if (dev < 0 || dev >= kMaxDevices) return FA_ERR_INVALID;
    unsigned* base = nullptr;
    {
        std::lock_guard<std::mutex> hold(g_symbol_mutex);
        base = sync_words_of(dev);
    }
    if (!base) return FA_ERR_INVALID;
    e = hipMemsetAsync(base, 0xFF, sizeof(unsigned) * 64 * kSyncSlots, as_stream(stream));
    return e == hipSuccess ? FA_OK : hip_code(e);
}

static int round_body(const float* const* x, const double* w, int K, float* g, float* m, float* v,
                      size_t n, int opt, double lr, double beta1, double beta2, double tau,
                      float* step_out, float* step_ws, const fa_mean_segment* means, int n_means,
                      int k_ready, void* rest_event, int k_done, fa_stream_t stream) {
    if (n_means < 0 || (n_means > 0 && !means) || !x || K <= 0) return FA_ERR_INVALID;
    if (k_ready < 0 || k_ready > K || k_done < 0 || k_done >= K) return FA_ERR_INVALID;
    const bool split = (k_ready > 0 && k_ready < K) || k_done > 0;
    MeanTail tail;
    memset(&tail, 0, sizeof(tail));
    tail.Kall = static_cast<float>(K);
    size_t total = 0;
    bool foldable = n_means <= kTailSegments && K <= FA_KCHUNK && n > 0 && !split;
    for (int i = 0; i < n_means; ++i) {
        const fa_mean_segment& sg = means[i];
        const size_t width = sg.dtype == FA_DT_I64 ? 8 : 4;
        if (sg.dtype != FA_DT_F32 && sg.dtype != FA_DT_I64 && sg.dtype != FA_DT_I32) return FA_ERR_INVALID;
        if (sg.n && !sg.out) return FA_ERR_INVALID;
        if (!aligned_to(sg.out, width) || sg.row_offset % width) return FA_ERR_ALIGN;
        for (int k = 0; k < K; ++k) {
            if (!x[k]) return FA_ERR_INVALID;
            if (!aligned_to(x[k], width)) return FA_ERR_ALIGN;
        }
        total += sg.n;
        if (foldable) {
            tail.out[i] = sg.out;
            tail.off[i] = static_cast<long long>(sg.row_offset);
            tail.n[i] = static_cast<unsigned>(sg.n);
            tail.dtype[i] = sg.dtype;
        }
    }
    // the folded entries ride on at most a quarter of the persistent kernel's blocks
    foldable = foldable && total > 0 && total <= static_cast<size_t>(device_cu_count() / 4) * kThreads;
    tail.total = foldable ? static_cast<unsigned>(total) : 0u;
    bool done = false;
    ReduceCall c{x, w, K, g, nullptr, m, v, n, opt, make_hyper(lr, beta1, beta2, tau),
                 step_out, step_ws, 1};
    if (foldable) {
        c.tail = &tail;
        c.tail_done = &done;
    }
    if (k_ready > 0 && k_ready < K) {
        c.k_ready = k_ready;
        c.rest_event = static_cast<hipEvent_t>(rest_event);
    }
    c.k_done = k_done;
    int rc = n > 0 ? reduce_dispatch(c, as_stream(stream)) : FA_OK;
    if (rc != FA_OK || done) return rc;
    if (k_ready > 0 && k_ready < K && n == 0 && rest_event) {
        // no weighted entries, so nothing has waited yet: the means below read every client's row
        hipError_t e = hipStreamWaitEvent(as_stream(stream), static_cast<hipEvent_t>(rest_event), 0);
        if (e != hipSuccess) return hip_code(e);
    }
    for (int i = 0; i < n_means; ++i) {
        const fa_mean_segment& sg = means[i];
        std::vector<const void*> rows(static_cast<size_t>(K));
        for (int k = 0; k < K; ++k)
            rows[static_cast<size_t>(k)] = reinterpret_cast<const char*>(x[k]) + sg.row_offset;
        if (sg.dtype == FA_DT_F32)
            rc = mean_dispatch(reinterpret_cast<const float* const*>(rows.data()), K,
                               static_cast<float*>(sg.out), sg.n, as_stream(stream));
        else if (sg.dtype == FA_DT_I64)
            rc = mean_dispatch(reinterpret_cast<const int64_t* const*>(rows.data()), K,
                               static_cast<int64_t*>(sg.out), sg.n, as_stream(stream));
        else
            rc = mean_dispatch(reinterpret_cast<const int32_t* const*>(rows.data()), K,
                               static_cast<int32_t*>(sg.out), sg.n, as_stream(stream));
        if (rc != FA_OK) return rc;
    }
    return FA_OK;
}

int fa_round(const float* const* x, const double* w, int K, float* g, float* m, float* v,
             size_t n, int opt, double lr, double beta1, double beta2, double tau, float* step_out,
             float* step_ws, const fa_mean_segment* means, int n_means, int k_ready,
             void* rest_event, fa_stream_t stream) {
    return round_body(x, w, K, g, m, v, n, opt, lr, beta1, beta2, tau, step_out, step_ws, means,
                      n_means, k_ready, rest_event, 0, stream);
}

int fa_round_resume(const float* const* x, const double* w, int K, int k_done, float* g, float* m,
                    float* v, size_t n, int opt, double lr, double beta1, double beta2, double tau,
                    float* step_ws, const fa_mean_segment* means, int n_means,
                    fa_stream_t stream) {
    if (k_done > 0 && !step_ws) return FA_ERR_WORKSPACE;
    return round_body(x, w, K, g, m, v, n, opt, lr, beta1, beta2, tau, nullptr, step_ws, means,
                      n_means, 0, nullptr, k_done, stream);
}

// A round's arguments, kept on the library's side of the boundary between launches.
struct fa_round_plan {
    std::vector<const float*> x;
    std::vector<double> w;
    std::vector<fa_mean_segment> means;
    float *g, *m, *v, *step_ws;
    size_t n;
    int opt;
    double lr, beta1, beta2, tau;
};

int fa_round_plan_create(const float* const* x, const double* w, int K, float* g, float* m, float* v,
                         size_t n, int opt, double lr, double beta1, double beta2, double tau,
                         float* step_ws, const fa_mean_segment* means, int n_means,
                         fa_round_plan** plan) {
    if (!plan || !x || !w || K <= 0 || n_means < 0 || (n_means > 0 && !means)) return FA_ERR_INVALID;
    fa_round_plan* p = new (std::nothrow) fa_round_plan;
    if (!p) return FA_ERR_INVALID;
    p->x.assign(x, x + K);
    p->w.assign(w, w + K);
    p->means.assign(means, means + n_means);
    p->g = g; p->m = m; p->v = v; p->step_ws = step_ws;
    p->n = n; p->opt = opt;
    p->lr = lr; p->beta1 = beta1; p->beta2 = beta2; p->tau = tau;
    *plan = p;
    return FA_OK;
}

int fa_round_plan_launch(const fa_round_plan* p, fa_stream_t stream) {
    if (!p) return FA_ERR_INVALID;
    return round_body(p->x.data(), p->w.data(), static_cast<int>(p->x.size()), p->g, p->m, p->v, p->n, p->opt,
                      p->lr, p->beta1, p->beta2, p->tau, nullptr, p->step_ws,
                      p->means.empty() ? nullptr : p->means.data(), static_cast<int>(p->means.size()), 0,
                      nullptr, 0, stream);
}

void fa_round_plan_destroy(fa_round_plan* p) { delete p; }

int fa_reduce_apply(const float* const* x, const double* w, int K, float* g, size_t n,
                    float* step_ws, fa_stream_t stream) {
    return fa_reduce_opt(x, w, K, g, nullptr, nullptr, n, FA_OPT_NONE, 0, 0, 0, 0, nullptr,
                         step_ws, stream);
}

int fa_reduce_opt_static(const float* const* x, const double* w, int K, float* g, float* m,
                         float* v, size_t n, int opt, double lr, double beta1, double beta2,
                         double tau, float* step_out, fa_stream_t stream) {
    float* ws = nullptr;
    if (K > FA_KCHUNK && !step_out) {
        // the partial step of the first client chunks is carried through the library's array
        if (n > static_cast<size_t>(FA_STATIC_WS_FLOATS)) return FA_ERR_TOO_LARGE;
        int rc = scratch_for_current_device(&ws);
        if (rc != FA_OK) return rc;
    }
    ReduceCall c{x, w, K, g, nullptr, m, v, n, opt, make_hyper(lr, beta1, beta2, tau),
                 step_out, ws, 1};
    return reduce_dispatch(c, as_stream(stream));
}

int fa_reduce_opt_slab(const float* slab, size_t stride, const double* w, int K, float* g,
                       float* m, float* v, size_t n, int opt, double lr, double beta1,
                       double beta2, double tau, float* step_out, float* step_ws,
                       fa_stream_t stream) {
    if (!slab || K <= 0) return FA_ERR_INVALID;
    std::vector<const float*> x(static_cast<size_t>(K));
    for (int k = 0; k < K; ++k) x[static_cast<size_t>(k)] = slab + static_cast<size_t>(k) * stride;
    return fa_reduce_opt(x.data(), w, K, g, m, v, n, opt, lr, beta1, beta2, tau, step_out, step_ws,
                         stream);
}

int fa_reduce_step(const float* const* x, const double* w, int K, const float* g, float* step_out,
                   size_t n, fa_stream_t stream) {
    ReduceCall c{x, w, K, nullptr, g, nullptr, nullptr, n, FA_OPT_NONE, make_hyper(0, 0, 0, 0),
                 step_out, nullptr, 0};
    return reduce_dispatch(c, as_stream(stream));
}

int fa_reduce_step_resume(const float* const* x, const double* w, int K, int k_done,
                          const float* g, float* step, size_t n, fa_stream_t stream) {
    ReduceCall c{x, w, K, nullptr, g, nullptr, nullptr, n, FA_OPT_NONE, make_hyper(0, 0, 0, 0),
                 step, nullptr, 0};
    c.k_done = k_done;
    return reduce_dispatch(c, as_stream(stream));
}

int fa_apply_step(float* g, const float* step, size_t n, fa_stream_t stream) {
    if (!g || !step) return FA_ERR_INVALID;
    if (!aligned_to(g, 4) || !aligned_to(step, 4)) return FA_ERR_ALIGN;
    hipStream_t st = as_stream(stream);
    return for_each_range(n, aligned16(g) && aligned16(step), [&](size_t off, size_t cnt, bool vec) {
        if (vec)
            hipLaunchKernelGGL((apply_kernel<4>), dim3(grid_for(cnt)), dim3(kThreads), 0, st,
                               g + off, step + off, static_cast<unsigned>(cnt));
        else
            hipLaunchKernelGGL((apply_kernel<1>), dim3(grid_for(cnt)), dim3(kThreads), 0, st,
                               g + off, step + off, static_cast<unsigned>(cnt));
        return launch_status();
    });
}

int fa_server_opt_step(float* step, float* m, float* v, size_t n, int opt, double lr,
                       double beta1, double beta2, double tau, fa_stream_t stream) {
    if (opt == FA_OPT_NONE) return FA_OK;
    if (opt < FA_OPT_NONE || opt > FA_OPT_ADAGRAD) return FA_ERR_INVALID;
    if (!step || !v || (opt >= FA_OPT_ADAM && !m)) return FA_ERR_INVALID;
    if (!aligned_to(step, 4) || !aligned_to(v, 4) || (m && !aligned_to(m, 4))) return FA_ERR_ALIGN;
    const bool a16 = aligned16(step) && aligned16(v) && (!m || aligned16(m));
    const Hyper h = make_hyper(lr, beta1, beta2, tau);
    hipStream_t st = as_stream(stream);
    return for_each_range(n, a16, [&](size_t off, size_t cnt, bool vec) {
        float* mo = m ? m + off : nullptr;
        const dim3 grid(grid_for(cnt)), block(kThreads);
        const unsigned c = static_cast<unsigned>(cnt);
#define FA_LAUNCH_OPT(OPT)                                                                     \
    if (vec)                                                                                   \
        hipLaunchKernelGGL((opt_step_kernel<OPT, 4>), grid, block, 0, st, step + off, mo,      \
                           v + off, h, c);                                                     \
    else                                                                                       \
        hipLaunchKernelGGL((opt_step_kernel<OPT, 1>), grid, block, 0, st, step + off, mo,      \
                           v + off, h, c);
        switch (opt) {
            case FA_OPT_AVGM: FA_LAUNCH_OPT(FA_OPT_AVGM) break;
            case FA_OPT_ADAM: FA_LAUNCH_OPT(FA_OPT_ADAM) break;
            case FA_OPT_YOGI: FA_LAUNCH_OPT(FA_OPT_YOGI) break;
            default: FA_LAUNCH_OPT(FA_OPT_ADAGRAD) break;
        }
#undef FA_LAUNCH_OPT
        return launch_status();
    });
}

int fa_weighted_mix(const float* const* x, const double* w, int K, float* g, double coeff,
                    int mode, float* acc_ws, size_t n, fa_stream_t stream) {
    if (!x || !w || !g || K <= 0) return FA_ERR_INVALID;
    if (mode < FA_MIX_REPLACE || mode > FA_MIX_SUM_SET) return FA_ERR_INVALID;
    if (mode == FA_MIX_CONVEX_ONE && K != 1) return FA_ERR_INVALID;
    if (K > FA_KCHUNK && mode != FA_MIX_SUBTRACT && !acc_ws) return FA_ERR_WORKSPACE;
    if (n == 0) return FA_OK;
    bool a4 = aligned_to(g, 4) && (!acc_ws || aligned_to(acc_ws, 4));
    bool a16 = aligned16(g) && (!acc_ws || aligned16(acc_ws));
    for (int k = 0; k < K; ++k) {
        if (!x[k]) return FA_ERR_INVALID;
        a4 = a4 && aligned_to(x[k], 4);
        a16 = a16 && aligned16(x[k]);
    }
    if (!a4) return FA_ERR_ALIGN;
    hipStream_t st = as_stream(stream);
    const int chunks = (K + FA_KCHUNK - 1) / FA_KCHUNK;
    return for_each_range(n, a16, [&](size_t off, size_t cnt, bool vec) {
        for (int ch = 0; ch < chunks; ++ch) {
            MixArgs a;
            memset(&a, 0, sizeof(a));
            const int k0 = ch * FA_KCHUNK;
            a.K = (K - k0 < FA_KCHUNK) ? K - k0 : FA_KCHUNK;
            for (int k = 0; k < a.K; ++k) {
                a.x[k] = x[k0 + k] + off;
                a.w[k] = static_cast<float>(w[k0 + k]);
            }
            a.g = g + off;
            a.acc_in = (ch == 0 || mode == FA_MIX_SUBTRACT) ? nullptr : acc_ws + off;
            a.acc_out = acc_ws ? acc_ws + off : nullptr;
            a.n = static_cast<unsigned>(cnt);
            a.mode = mode;
            a.last = (ch == chunks - 1);
            a.coeff = static_cast<float>(coeff);
            int rc = launch_mix(a, vec, st);
            if (rc != FA_OK) return rc;
        }
        return static_cast<int>(FA_OK);
    });
}

int fa_mean_f32(const float* const* x, int K, float* g, size_t n, fa_stream_t stream) {
    return mean_dispatch<float>(x, K, g, n, as_stream(stream));
}
int fa_mean_i64(const int64_t* const* x, int K, int64_t* g, size_t n, fa_stream_t stream) {
    return mean_dispatch<int64_t>(x, K, g, n, as_stream(stream));
}
int fa_mean_i32(const int32_t* const* x, int K, int32_t* g, size_t n, fa_stream_t stream) {
    return mean_dispatch<int32_t>(x, K, g, n, as_stream(stream));
}

int fa_async_mix(const float* x, double alpha_t, int gradient_based, int sub_form, float* g,
                 float* step_out, size_t n, fa_stream_t stream) {
    if (!x || !g) return FA_ERR_INVALID;
    if (!aligned_to(x, 4) || !aligned_to(g, 4) || (step_out && !aligned_to(step_out, 4)))
        return FA_ERR_ALIGN;
    const bool a16 = aligned16(x) && aligned16(g) && (!step_out || aligned16(step_out));
    const float alpha = static_cast<float>(alpha_t);
    hipStream_t st = as_stream(stream);
    if (!gradient_based) return async_mix_mode<0>(x, alpha, g, step_out, n, a16, st);
    if (sub_form) return async_mix_mode<1>(x, alpha, g, step_out, n, a16, st);
    return async_mix_mode<2>(x, alpha, g, step_out, n, a16, st);
}

int fa_buff_accum(const float* x, double alpha_t, int gradient_based, const float* g, float* step,
                  int first, size_t n, fa_stream_t stream) {
    return buff_accum_dispatch<false>(x, alpha_t, gradient_based, const_cast<float*>(g), step,
                                      first, n, as_stream(stream));
}

int fa_buff_accum_flush(const float* x, double alpha_t, int gradient_based, float* g, float* step,
                        int first, size_t n, fa_stream_t stream) {
    return buff_accum_dispatch<true>(x, alpha_t, gradient_based, g, step, first, n,
                                     as_stream(stream));
}

int fa_buff_accum_raw_f32(const float* x, float* step, int first, size_t n, fa_stream_t stream) {
    return accum_raw_dispatch<float>(x, step, first, n, as_stream(stream));
}
int fa_buff_accum_raw_i64(const int64_t* x, int64_t* step, int first, size_t n,
                          fa_stream_t stream) {
    return accum_raw_dispatch<int64_t>(x, step, first, n, as_stream(stream));
}
int fa_buff_accum_raw_i32(const int32_t* x, int32_t* step, int first, size_t n,
                          fa_stream_t stream) {
    return accum_raw_dispatch<int32_t>(x, step, first, n, as_stream(stream));
}

int fa_buff_flush_mean_f32(float* g, const float* step, int K, size_t n, fa_stream_t stream) {
    return flush_mean_dispatch<float>(g, step, K, n, as_stream(stream));
}
int fa_buff_flush_mean_i64(int64_t* g, const int64_t* step, int K, size_t n, fa_stream_t stream) {
    return flush_mean_dispatch<int64_t>(g, step, K, n, as_stream(stream));
}
int fa_buff_flush_mean_i32(int32_t* g, const int32_t* step, int K, size_t n, fa_stream_t stream) {
    return flush_mean_dispatch<int32_t>(g, step, K, n, as_stream(stream));
}

int fa_fill_f32(float* p, float value, size_t n, fa_stream_t stream) {
    if (!p) return FA_ERR_INVALID;
    if (!aligned_to(p, 4)) return FA_ERR_ALIGN;
    hipStream_t st = as_stream(stream);
    return for_each_range(n, aligned16(p), [&](size_t off, size_t cnt, bool vec) {
        if (vec)
            hipLaunchKernelGGL((fill_kernel<4>), dim3(grid_for(cnt)), dim3(kThreads), 0, st,
                               p + off, value, static_cast<unsigned>(cnt));
        else
            hipLaunchKernelGGL((fill_kernel<1>), dim3(grid_for(cnt)), dim3(kThreads), 0, st,
                               p + off, value, static_cast<unsigned>(cnt));
        return launch_status();
    });
}

int fa_synth_fill(float* p, const float* base, size_t n, uint64_t seed, uint64_t first,
                  float scale, fa_stream_t stream) {
    if (!p) return FA_ERR_INVALID;
    if (!aligned_to(p, 4) || (base && !aligned_to(base, 4))) return FA_ERR_ALIGN;
    hipStream_t st = as_stream(stream);
    for (size_t off = 0; off < n; off += kSegment) {
        size_t len = (n - off < kSegment) ? n - off : kSegment;
        hipLaunchKernelGGL(synth_kernel, dim3(grid_for(len)), dim3(kThreads), 0, st, p + off,
                           base ? base + off : nullptr, static_cast<unsigned>(len), seed,
                           first + off, scale);
        int rc = launch_status();
        if (rc != FA_OK) return rc;
    }
    return FA_OK;
}

int fa_pack_host(void* dst, const void* const* src, const size_t* dst_off, const size_t* nbytes,
                 int T, int threads) {
    if (!dst || !src || !dst_off || !nbytes || T < 0) return FA_ERR_INVALID;
    for (int t = 0; t < T; ++t)
        if (!src[t] && nbytes[t]) return FA_ERR_INVALID;
    size_t total = 0;
    for (int t = 0; t < T; ++t) total += nbytes[t];
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    // Byte-balanced split: thread j copies bytes [total*j/threads, total*(j+1)/threads) of the
    // concatenated payload, so a few large tensors do not serialise the copy.
    if (threads == 1 || total < (size_t(1) << 20)) {
        for (int t = 0; t < T; ++t)
            if (nbytes[t]) memcpy(static_cast<char*>(dst) + dst_off[t], src[t], nbytes[t]);
        return FA_OK;
    }
    auto work = [&](int j) {
        const size_t lo = total / threads * j + (total % threads) * j / threads;
        const size_t hi = (j == threads - 1) ? total
                                             : total / threads * (j + 1) + (total % threads) * (j + 1) / threads;
        size_t pos = 0;
        for (int t = 0; t < T && pos < hi; ++t) {
            const size_t b = pos, e = pos + nbytes[t];
            pos = e;
            const size_t s = b > lo ? b : lo, f = e < hi ? e : hi;
            if (s < f)
                memcpy(static_cast<char*>(dst) + dst_off[t] + (s - b),
                       static_cast<const char*>(src[t]) + (s - b), f - s);
        }
    };
    // The library's sleeping workers take the parts (starting threads per call costs ~0.2 ms,
    // as much as packing a window of a few MiB); a second caller at the same time, or a call
    // that wants more parts than there are workers, starts threads of its own.
    {
        std::unique_lock<std::mutex> mine(g_pack_pool_busy, std::try_to_lock);
        if (mine.owns_lock()) {
            PackPool* pool = pack_pool();
            if (pool && threads <= pool->workers() + 1) {
                pool->run(threads, work);
                return FA_OK;
            }
        }
    }
    std::vector<std::thread> own;
    own.reserve(static_cast<size_t>(threads));
    for (int j = 1; j < threads; ++j) own.emplace_back(work, j);
    work(0);
    for (auto& th : own) th.join();
    return FA_OK;
}

int fa_pack_host_window(void* dst, const void* const* src, const size_t* dst_off,
                        const size_t* nbytes, int T, size_t win_lo, size_t win_hi, int threads) {
    if (!dst || !src || !dst_off || !nbytes || T < 0 || win_hi < win_lo) return FA_ERR_INVALID;
    // the parts of the pieces that fall inside [win_lo, win_hi) of the destination
    std::vector<const void*> s2;
    std::vector<size_t> o2, n2;
    for (int t = 0; t < T; ++t) {
        const size_t b = dst_off[t], e = dst_off[t] + nbytes[t];
        const size_t lo = b > win_lo ? b : win_lo, hi = e < win_hi ? e : win_hi;
        if (lo >= hi) continue;
        if (!src[t]) return FA_ERR_INVALID;
        s2.push_back(static_cast<const char*>(src[t]) + (lo - b));
        o2.push_back(lo);
        n2.push_back(hi - lo);
    }
    if (s2.empty()) return FA_OK;
    return fa_pack_host(dst, s2.data(), o2.data(), n2.data(), static_cast<int>(s2.size()), threads);
}

namespace {

// CRC-32 (the zip / zlib polynomial, reflected) by eight table look-ups per 8 bytes.
uint32_t g_crc_table[8][256];
std::once_flag g_crc_once;

void crc_tables() {
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        g_crc_table[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
        for (int t = 1; t < 8; ++t)
            g_crc_table[t][i] = (g_crc_table[t - 1][i] >> 8) ^ g_crc_table[0][g_crc_table[t - 1][i] & 0xFFu];
}

// Copies [src, src + n) to dst and returns the CRC-32 of the bytes (one pass over the source).
uint32_t copy_crc32(unsigned char* dst, const unsigned char* src, size_t n) {
    uint32_t c = 0xFFFFFFFFu;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, src + i, 8);
        memcpy(dst + i, &w, 8);
        const uint32_t lo = static_cast<uint32_t>(w) ^ c, hi = static_cast<uint32_t>(w >> 32);
        c = g_crc_table[7][lo & 0xFFu] ^ g_crc_table[6][(lo >> 8) & 0xFFu] ^ g_crc_table[5][(lo >> 16) & 0xFFu] ^
            g_crc_table[4][lo >> 24] ^ g_crc_table[3][hi & 0xFFu] ^ g_crc_table[2][(hi >> 8) & 0xFFu] ^
            g_crc_table[1][(hi >> 16) & 0xFFu] ^ g_crc_table[0][hi >> 24];
    }
    for (; i < n; ++i) {
        dst[i] = src[i];
        c = g_crc_table[0][(c ^ src[i]) & 0xFFu] ^ (c >> 8);
    }
    return c ^ 0xFFFFFFFFu;
}

}  // namespace

int fa_host_copy_crc32(void* dst, const void* src, size_t nbytes, size_t chunk, uint32_t* crcs,
                       int threads) {
    if (!dst || !src || !crcs || chunk == 0) return FA_ERR_INVALID;
    if (!nbytes) return FA_OK;
    std::call_once(g_crc_once, crc_tables);
    {
        // the destination is usually a fresh allocation: ask for huge pages before its first touch
        const uintptr_t page = 4096;
        const uintptr_t a = (reinterpret_cast<uintptr_t>(dst) + page - 1) & ~(page - 1);
        const uintptr_t e = (reinterpret_cast<uintptr_t>(dst) + nbytes) & ~(page - 1);
        if (e > a && e - a >= (size_t(4) << 20)) madvise(reinterpret_cast<void*>(a), e - a, MADV_HUGEPAGE);
    }
    const size_t chunks = (nbytes + chunk - 1) / chunk;
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    if (static_cast<size_t>(threads) > chunks) threads = static_cast<int>(chunks);
    std::atomic<size_t> next{0};
    auto work = [&](int) {
        for (;;) {
            const size_t c = next.fetch_add(1, std::memory_order_relaxed);
            if (c >= chunks) return;
            const size_t a = c * chunk, n = (nbytes - a < chunk) ? nbytes - a : chunk;
            crcs[c] = copy_crc32(static_cast<unsigned char*>(dst) + a, static_cast<const unsigned char*>(src) + a, n);
        }
    };
    if (threads == 1) {
        work(0);
        return FA_OK;
    }
    {
        std::unique_lock<std::mutex> mine(g_pack_pool_busy, std::try_to_lock);
        if (mine.owns_lock()) {
            PackPool* pool = pack_pool();
            if (pool && threads <= pool->workers() + 1) {
                pool->run(threads, work);
                return FA_OK;
            }
        }
    }
    std::vector<std::thread> own;
    own.reserve(static_cast<size_t>(threads));
    for (int j = 1; j < threads; ++j) own.emplace_back(work, j);
    work(0);
    for (auto& th : own) th.join();
    return FA_OK;
}

int fa_host_copy_fresh(void* dst, const void* src, size_t nbytes, int threads) {
    if ((!dst || !src) && nbytes) return FA_ERR_INVALID;
    if (!nbytes) return FA_OK;
    // First touch of a fresh allocation is page-fault bound; with 2-MiB pages (where the kernel
    // offers them on request) the same threads fill it about three times faster.
    const uintptr_t page = 4096;
    const uintptr_t a = (reinterpret_cast<uintptr_t>(dst) + page - 1) & ~(page - 1);
    const uintptr_t e = (reinterpret_cast<uintptr_t>(dst) + nbytes) & ~(page - 1);
    if (e > a && e - a >= (size_t(4) << 20)) madvise(reinterpret_cast<void*>(a), e - a, MADV_HUGEPAGE);
    const void* srcs[1] = {src};
    const size_t offs[1] = {0}, lens[1] = {nbytes};
    return fa_pack_host(dst, srcs, offs, lens, 1, threads);
}

}  // extern "C"
