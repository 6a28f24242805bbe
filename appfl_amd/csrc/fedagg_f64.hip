// fedagg_f64.hip -- the aggregation kernels for float64 state-dict entries.
//
// The reference's float path takes any float tensor (misc/memory_utils.py:112-131); for float64
// tensors PyTorch keeps Python scalars in double, so here every scalar (client weight, alpha_t,
// hyper-parameters, 1 - beta) enters the arithmetic as the double it is, and every operation is a
// separately rounded IEEE double operation in the reference's order (-ffp-contract=off, correctly
// rounded / and sqrt).  float64 entries are rare and small next to the fp32 parameters, so these
// are plain one-element-per-thread streaming kernels with eight client loads in flight.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/appfl_amd.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 1 << 20;   // grid cap (covers a launch segment at one tile per block); larger inputs are walked grid-stride
constexpr int kAhead = 8;   // client loads in flight per element

struct Hyper64 {
    double lr, b1, omb1, b2, omb2, tau;
};

struct Reduce64Args {
    const double* x[FA_KCHUNK];
    double w[FA_KCHUNK];
    double* g;
    const double* g_ro;
    double* m;
    double* v;
    const double* step_in;
    double* step_out;
    size_t n;
    int K, last, apply;
    Hyper64 h;
};

__device__ __forceinline__ double sign_f64(double d) {
    return static_cast<double>(static_cast<int>(0.0 < d) - static_cast<int>(d < 0.0));
}

template <int OPT>
__device__ __forceinline__ double optimizer_step64(double step, double& m, double& v, const Hyper64& h) {
    if (OPT == FA_OPT_NONE) return step;
    if (OPT == FA_OPT_AVGM) {
        double mt = v * h.b1;
        v = mt + step;
        return v;
    }
    double mt = m * h.b1;
    double st = step * h.omb1;
    m = mt + st;
    double s2 = step * step;
    if (OPT == FA_OPT_ADAM) {
        double vt = v * h.b2;
        double sq = s2 * h.omb2;
        v = vt + sq;
    } else if (OPT == FA_OPT_YOGI) {
        double sg = sign_f64(v - s2);
        double yu = s2 * sg;
        double ys = yu * h.omb2;
        v = v - ys;
    } else {
        v = v + s2;
    }
    double num = h.lr * m;
    double den = sqrt(v) + h.tau;
    return num / den;
}

template <int OPT>
__global__ __launch_bounds__(kThreads) void reduce64_kernel(const Reduce64Args a) {
    const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
    for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < a.n; i += stride) {
        const double g = a.apply ? a.g[i] : a.g_ro[i];
        double s = a.step_in ? a.step_in[i] : 0.0;
        int k = 0;
        for (; k + kAhead <= a.K; k += kAhead) {
            double x[kAhead];
#pragma unroll
            for (int u = 0; u < kAhead; ++u) x[u] = __builtin_nontemporal_load(a.x[k + u] + i);
#pragma unroll
            for (int u = 0; u < kAhead; ++u) {
                double d = x[u] - g;
                double wd = d * a.w[k + u];
                s = s + wd;
            }
        }
        for (; k < a.K; ++k) {
            double d = __builtin_nontemporal_load(a.x[k] + i) - g;
            double wd = d * a.w[k];
            s = s + wd;
        }
        if (!a.last) {
            a.step_out[i] = s;
            continue;
        }
        if (OPT != FA_OPT_NONE) {
            double v = a.v[i];
            double m = (OPT != FA_OPT_AVGM) ? a.m[i] : 0.0;
            s = optimizer_step64<OPT>(s, m, v, a.h);
            a.v[i] = v;
            if (OPT != FA_OPT_AVGM) a.m[i] = m;
        }
        if (a.step_out) a.step_out[i] = s;
        if (a.apply) a.g[i] = g + s;
    }
}

template <int OPT>
__global__ __launch_bounds__(kThreads) void opt_step64_kernel(double* step, double* m, double* v, Hyper64 h, size_t n) {
    const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
    for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += stride) {
        double vv = v[i];
        double mm = (OPT != FA_OPT_AVGM) ? m[i] : 0.0;
        step[i] = optimizer_step64<OPT>(step[i], mm, vv, h);
        v[i] = vv;
        if (OPT != FA_OPT_AVGM) m[i] = mm;
    }
}

__global__ __launch_bounds__(kThreads) void apply64_kernel(double* g, const double* step, size_t n) {
    const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
    for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += stride) g[i] = g[i] + step[i];
}

struct Mean64Args {
    const double* x[FA_KCHUNK];
    double* g;
    const double* sum_in;
    size_t n;
    int K, last;
    double Kall;
};

__global__ __launch_bounds__(kThreads) void mean64_kernel(const Mean64Args a) {
    const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
    for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < a.n; i += stride) {
        double s = a.sum_in ? a.sum_in[i] : 0.0;
        for (int k = 0; k < a.K; ++k) s = s + __builtin_nontemporal_load(a.x[k] + i);
        a.g[i] = a.last ? s / a.Kall : s;
    }
}

// MODE 0: model-based, 1: gradient-based "0 - a*x", 2: gradient-based "a * (-x)".
template <int MODE>
__device__ __forceinline__ double async_step64(double x, double g, double alpha) {
    if (MODE == 0) {
        double d = x - g;
        return alpha * d;
    }
    if (MODE == 1) {
        double ax = alpha * x;
        return 0.0 - ax;
    }
    double nx = -x;
    return alpha * nx;
}

template <int MODE>
__global__ __launch_bounds__(kThreads) void async_mix64_kernel(const double* x, double alpha, double* g, double* step_out, size_t n) {
    const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
    for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += stride) {
        const double gv = g[i];
        const double s = async_step64<MODE>(x[i], gv, alpha);
        if (step_out) step_out[i] = s;
        g[i] = gv + s;
    }
}

template <int MODE, bool FLUSH>
__global__ __launch_bounds__(kThreads) void buff_accum64_kernel(const double* x, double alpha, double* g, double* step, int first, size_t n) {
    const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
    for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += stride) {
        const double gv = g[i];
        double s = first ? 0.0 : step[i];
        const double d = async_step64<MODE>(x[i], gv, alpha);
        s = s + d;
        step[i] = s;
        if (FLUSH) g[i] = gv + s;
    }
}

__global__ __launch_bounds__(kThreads) void accum_raw64_kernel(const double* x, double* step, int first, size_t n) {
    const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
    for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += stride) {
        const double s = first ? 0.0 : step[i];
        step[i] = s + x[i];
    }
}

__global__ __launch_bounds__(kThreads) void flush_mean64_kernel(double* g, const double* step, double kf, size_t n) {
    const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
    for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += stride) g[i] = step[i] / kf;
}

// Weighted mixes of K client vectors that do not subtract g first (FedCompass / FedQueue):
// the six combine modes of the fp32 mix_kernel, in double.
struct Mix64Args {
    const double* x[FA_KCHUNK];
    double w[FA_KCHUNK];
    double* g;
    const double* acc_in;   // partial sum of earlier client chunks (NULL: 0)
    double* acc_out;        // where a non-final chunk leaves its partial sum
    size_t n;
    int K, mode, last;
    double coeff;
};

__global__ __launch_bounds__(kThreads) void mix64_kernel(const Mix64Args a) {
    const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
    for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < a.n; i += stride) {
        double g = 0.0;
        if (a.mode != FA_MIX_REPLACE && a.mode != FA_MIX_SUM_SET) g = a.g[i];
        double acc;
        if (a.mode == FA_MIX_SUBTRACT)
            acc = g;
        else if (a.acc_in)
            acc = a.acc_in[i];
        else
            acc = (a.mode >= FA_MIX_CONVEX_ONE) ? -0.0 : 0.0;   // -0 + y == y for every y
        int k = 0;
        for (; k + kAhead <= a.K; k += kAhead) {
            double x[kAhead];
#pragma unroll
            for (int u = 0; u < kAhead; ++u) x[u] = __builtin_nontemporal_load(a.x[k + u] + i);
#pragma unroll
            for (int u = 0; u < kAhead; ++u) {
                double xw = x[u] * a.w[k + u];
                acc = (a.mode == FA_MIX_SUBTRACT) ? acc - xw : acc + xw;
            }
        }
        for (; k < a.K; ++k) {
            double xw = __builtin_nontemporal_load(a.x[k] + i) * a.w[k];
            acc = (a.mode == FA_MIX_SUBTRACT) ? acc - xw : acc + xw;
        }
        if (!a.last && a.mode != FA_MIX_SUBTRACT) {
            a.acc_out[i] = acc;
            continue;
        }
        if ((a.mode == FA_MIX_CONVEX || a.mode == FA_MIX_CONVEX_ONE) && a.last) {
            double gt = g * a.coeff;
            acc = gt + acc;
        }
        if (a.mode == FA_MIX_SUM_ADD && a.last) acc = g + acc;
        a.g[i] = acc;
    }
}

__global__ __launch_bounds__(kThreads) void fill64_kernel(double* p, double value, size_t n) {
    const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
    for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += stride) p[i] = value;
}

inline hipStream_t as_stream(fa_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? FA_OK : FA_ERR_HIP - static_cast<int>(e);
}
inline bool aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }
inline unsigned grid_for(size_t n) {
    size_t b = (n + kThreads - 1) / kThreads;
    return static_cast<unsigned>(b < 1 ? 1 : (b > static_cast<size_t>(kMaxBlocks) ? kMaxBlocks : b));
}

Hyper64 make_hyper64(double lr, double beta1, double beta2, double tau) {
    return Hyper64{lr, beta1, 1.0 - beta1, beta2, 1.0 - beta2, tau};
}

int reduce64(const double* const* x, const double* w, int K, double* g, const double* g_ro, double* m,
             double* v, size_t n, int opt, const Hyper64& h, double* step_out, double* step_ws, int apply,
             hipStream_t st) {
    if (!x || !w || K <= 0) return FA_ERR_INVALID;
    if (apply ? !g : (!g_ro || !step_out)) return FA_ERR_INVALID;
    if (opt < FA_OPT_NONE || opt > FA_OPT_ADAGRAD) return FA_ERR_INVALID;
    if (opt != FA_OPT_NONE && !v) return FA_ERR_INVALID;
    if (opt >= FA_OPT_ADAM && !m) return FA_ERR_INVALID;
    if (K > FA_KCHUNK && !step_out && !step_ws) return FA_ERR_WORKSPACE;
    const void* ptrs[] = {g, g_ro, m, v, step_out, step_ws};
    for (const void* p : ptrs)
        if (p && !aligned8(p)) return FA_ERR_ALIGN;
    for (int k = 0; k < K; ++k) {
        if (!x[k]) return FA_ERR_INVALID;
        if (!aligned8(x[k])) return FA_ERR_ALIGN;
    }
    if (n == 0) return FA_OK;
    const int chunks = (K + FA_KCHUNK - 1) / FA_KCHUNK;
    double* carry = step_out ? step_out : step_ws;
    for (int ch = 0; ch < chunks; ++ch) {
        Reduce64Args a;
        memset(&a, 0, sizeof(a));
        const int k0 = ch * FA_KCHUNK;
        a.K = (K - k0 < FA_KCHUNK) ? K - k0 : FA_KCHUNK;
        for (int k = 0; k < a.K; ++k) {
            a.x[k] = x[k0 + k];
            a.w[k] = w[k0 + k];
        }
        a.g = g;
        a.g_ro = g_ro;
        a.m = m;
        a.v = v;
        a.last = (ch == chunks - 1);
        a.apply = apply;
        a.step_in = (ch == 0) ? nullptr : carry;
        a.step_out = a.last ? step_out : carry;
        a.n = n;
        a.h = h;
        const dim3 grid(grid_for(n)), block(kThreads);
        const int o = a.last ? opt : FA_OPT_NONE;
        switch (o) {
            case FA_OPT_NONE: hipLaunchKernelGGL((reduce64_kernel<FA_OPT_NONE>), grid, block, 0, st, a); break;
            case FA_OPT_AVGM: hipLaunchKernelGGL((reduce64_kernel<FA_OPT_AVGM>), grid, block, 0, st, a); break;
            case FA_OPT_ADAM: hipLaunchKernelGGL((reduce64_kernel<FA_OPT_ADAM>), grid, block, 0, st, a); break;
            case FA_OPT_YOGI: hipLaunchKernelGGL((reduce64_kernel<FA_OPT_YOGI>), grid, block, 0, st, a); break;
            default: hipLaunchKernelGGL((reduce64_kernel<FA_OPT_ADAGRAD>), grid, block, 0, st, a); break;
        }
        int rc = launch_status();
        if (rc != FA_OK) return rc;
    }
    return FA_OK;
}

template <bool FLUSH>
int buff64(const double* x, double alpha, int gradient_based, double* g, double* step, int first, size_t n,
           hipStream_t st) {
    if (!x || !g || !step) return FA_ERR_INVALID;
    if (!aligned8(x) || !aligned8(g) || !aligned8(step)) return FA_ERR_ALIGN;
    if (n == 0) return FA_OK;
    const dim3 grid(grid_for(n)), block(kThreads);
    if (gradient_based)
        hipLaunchKernelGGL((buff_accum64_kernel<2, FLUSH>), grid, block, 0, st, x, alpha, g, step, first, n);
    else
        hipLaunchKernelGGL((buff_accum64_kernel<0, FLUSH>), grid, block, 0, st, x, alpha, g, step, first, n);
    return launch_status();
}

}  // namespace

extern "C" {

int fa_reduce_opt_f64(const double* const* x, const double* w, int K, double* g, double* m, double* v,
                      size_t n, int opt, double lr, double beta1, double beta2, double tau,
                      double* step_out, double* step_ws, fa_stream_t stream) {
    return reduce64(x, w, K, g, nullptr, m, v, n, opt, make_hyper64(lr, beta1, beta2, tau), step_out,
                    step_ws, 1, as_stream(stream));
}

int fa_reduce_step_f64(const double* const* x, const double* w, int K, const double* g, double* step_out,
                       size_t n, fa_stream_t stream) {
    return reduce64(x, w, K, nullptr, g, nullptr, nullptr, n, FA_OPT_NONE, make_hyper64(0, 0, 0, 0),
                    step_out, nullptr, 0, as_stream(stream));
}

int fa_server_opt_step_f64(double* step, double* m, double* v, size_t n, int opt, double lr, double beta1,
                           double beta2, double tau, fa_stream_t stream) {
    if (opt == FA_OPT_NONE) return FA_OK;
    if (opt < FA_OPT_NONE || opt > FA_OPT_ADAGRAD) return FA_ERR_INVALID;
    if (!step || !v || (opt >= FA_OPT_ADAM && !m)) return FA_ERR_INVALID;
    if (!aligned8(step) || !aligned8(v) || (m && !aligned8(m))) return FA_ERR_ALIGN;
    if (n == 0) return FA_OK;
    const Hyper64 h = make_hyper64(lr, beta1, beta2, tau);
    const dim3 grid(grid_for(n)), block(kThreads);
    hipStream_t st = as_stream(stream);
    switch (opt) {
        case FA_OPT_AVGM: hipLaunchKernelGGL((opt_step64_kernel<FA_OPT_AVGM>), grid, block, 0, st, step, m, v, h, n); break;
        case FA_OPT_ADAM: hipLaunchKernelGGL((opt_step64_kernel<FA_OPT_ADAM>), grid, block, 0, st, step, m, v, h, n); break;
        case FA_OPT_YOGI: hipLaunchKernelGGL((opt_step64_kernel<FA_OPT_YOGI>), grid, block, 0, st, step, m, v, h, n); break;
        default: hipLaunchKernelGGL((opt_step64_kernel<FA_OPT_ADAGRAD>), grid, block, 0, st, step, m, v, h, n); break;
    }
    return launch_status();
}

int fa_apply_step_f64(double* g, const double* step, size_t n, fa_stream_t stream) {
    if (!g || !step) return FA_ERR_INVALID;
    if (!aligned8(g) || !aligned8(step)) return FA_ERR_ALIGN;
    if (n == 0) return FA_OK;
    hipLaunchKernelGGL(apply64_kernel, dim3(grid_for(n)), dim3(kThreads), 0, as_stream(stream), g, step, n);
    return launch_status();
}

int fa_mean_f64(const double* const* x, int K, double* g, size_t n, fa_stream_t stream) {
    if (!x || !g || K <= 0) return FA_ERR_INVALID;
    if (!aligned8(g)) return FA_ERR_ALIGN;
    for (int k = 0; k < K; ++k) {
        if (!x[k]) return FA_ERR_INVALID;
        if (!aligned8(x[k])) return FA_ERR_ALIGN;
    }
    if (n == 0) return FA_OK;
    const int chunks = (K + FA_KCHUNK - 1) / FA_KCHUNK;
    for (int ch = 0; ch < chunks; ++ch) {
        Mean64Args a;
        memset(&a, 0, sizeof(a));
        const int k0 = ch * FA_KCHUNK;
        a.K = (K - k0 < FA_KCHUNK) ? K - k0 : FA_KCHUNK;
        for (int k = 0; k < a.K; ++k) a.x[k] = x[k0 + k];
        a.g = g;
        a.sum_in = (ch == 0) ? nullptr : g;
        a.n = n;
        a.last = (ch == chunks - 1);
        a.Kall = static_cast<double>(K);
        hipLaunchKernelGGL(mean64_kernel, dim3(grid_for(n)), dim3(kThreads), 0, as_stream(stream), a);
        int rc = launch_status();
        if (rc != FA_OK) return rc;
    }
    return FA_OK;
}

int fa_weighted_mix_f64(const double* const* x, const double* w, int K, double* g, double coeff,
                        int mode, double* acc_ws, size_t n, fa_stream_t stream) {
    if (!x || !w || !g || K <= 0) return FA_ERR_INVALID;
    if (mode < FA_MIX_REPLACE || mode > FA_MIX_SUM_SET) return FA_ERR_INVALID;
    if (mode == FA_MIX_CONVEX_ONE && K != 1) return FA_ERR_INVALID;
    if (K > FA_KCHUNK && mode != FA_MIX_SUBTRACT && !acc_ws) return FA_ERR_WORKSPACE;
    if (!aligned8(g) || (acc_ws && !aligned8(acc_ws))) return FA_ERR_ALIGN;
    for (int k = 0; k < K; ++k) {
        if (!x[k]) return FA_ERR_INVALID;
        if (!aligned8(x[k])) return FA_ERR_ALIGN;
    }
    if (n == 0) return FA_OK;
    const int chunks = (K + FA_KCHUNK - 1) / FA_KCHUNK;
    for (int ch = 0; ch < chunks; ++ch) {
        Mix64Args a;
        memset(&a, 0, sizeof(a));
        const int k0 = ch * FA_KCHUNK;
        a.K = (K - k0 < FA_KCHUNK) ? K - k0 : FA_KCHUNK;
        for (int k = 0; k < a.K; ++k) {
            a.x[k] = x[k0 + k];
            a.w[k] = w[k0 + k];
        }
        a.g = g;
        a.acc_in = (ch == 0 || mode == FA_MIX_SUBTRACT) ? nullptr : acc_ws;
        a.acc_out = acc_ws;
        a.n = n;
        a.mode = mode;
        a.last = (ch == chunks - 1);
        a.coeff = coeff;
        hipLaunchKernelGGL(mix64_kernel, dim3(grid_for(n)), dim3(kThreads), 0, as_stream(stream), a);
        int rc = launch_status();
        if (rc != FA_OK) return rc;
    }
    return FA_OK;
}

int fa_async_mix_f64(const double* x, double alpha_t, int gradient_based, int sub_form, double* g,
                     double* step_out, size_t n, fa_stream_t stream) {
    if (!x || !g) return FA_ERR_INVALID;
    if (!aligned8(x) || !aligned8(g) || (step_out && !aligned8(step_out))) return FA_ERR_ALIGN;
    if (n == 0) return FA_OK;
    const dim3 grid(grid_for(n)), block(kThreads);
    hipStream_t st = as_stream(stream);
    if (!gradient_based)
        hipLaunchKernelGGL((async_mix64_kernel<0>), grid, block, 0, st, x, alpha_t, g, step_out, n);
    else if (sub_form)
        hipLaunchKernelGGL((async_mix64_kernel<1>), grid, block, 0, st, x, alpha_t, g, step_out, n);
    else
        hipLaunchKernelGGL((async_mix64_kernel<2>), grid, block, 0, st, x, alpha_t, g, step_out, n);
    return launch_status();
}

int fa_buff_accum_f64(const double* x, double alpha_t, int gradient_based, const double* g, double* step,
                      int first, size_t n, fa_stream_t stream) {
    return buff64<false>(x, alpha_t, gradient_based, const_cast<double*>(g), step, first, n, as_stream(stream));
}

int fa_buff_accum_flush_f64(const double* x, double alpha_t, int gradient_based, double* g, double* step,
                            int first, size_t n, fa_stream_t stream) {
    return buff64<true>(x, alpha_t, gradient_based, g, step, first, n, as_stream(stream));
}

int fa_buff_accum_raw_f64(const double* x, double* step, int first, size_t n, fa_stream_t stream) {
    if (!x || !step) return FA_ERR_INVALID;
    if (!aligned8(x) || !aligned8(step)) return FA_ERR_ALIGN;
    if (n == 0) return FA_OK;
    hipLaunchKernelGGL(accum_raw64_kernel, dim3(grid_for(n)), dim3(kThreads), 0, as_stream(stream), x, step, first, n);
    return launch_status();
}

int fa_buff_flush_mean_f64(double* g, const double* step, int K, size_t n, fa_stream_t stream) {
    if (!g || !step || K <= 0) return FA_ERR_INVALID;
    if (!aligned8(g) || !aligned8(step)) return FA_ERR_ALIGN;
    if (n == 0) return FA_OK;
    hipLaunchKernelGGL(flush_mean64_kernel, dim3(grid_for(n)), dim3(kThreads), 0, as_stream(stream), g, step,
                       static_cast<double>(K), n);
    return launch_status();
}

int fa_fill_f64(double* p, double value, size_t n, fa_stream_t stream) {
    if (!p) return FA_ERR_INVALID;
    if (!aligned8(p)) return FA_ERR_ALIGN;
    if (n == 0) return FA_OK;
    hipLaunchKernelGGL(fill64_kernel, dim3(grid_for(n)), dim3(kThreads), 0, as_stream(stream), p, value, n);
    return launch_status();
}

}  // extern "C"
