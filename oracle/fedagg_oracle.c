/*
 * fedagg_oracle.c -- CPU restatement of the arithmetic of APPFL's server-side aggregators on
 * flat vectors.  TEST INFRASTRUCTURE ONLY: it is the checker for the HIP kernels (tests/,
 * the smoke run, the benchmark's cpu_baseline leg) and is never part of the product path.
 *
 * Every formula is written as the sequence of separately rounded fp32 operations that the
 * reference's PyTorch CPU code performs (one tensor op = one rounding), evaluated per element
 * in client order.  Build with -ffp-contract=off and without -ffast-math.
 *
 * Pinned bit-for-bit against golden vectors produced by the reference's own aggregator classes
 * (tests/golden/); the reference's test-suite holds no numeric vectors for this path.
 *
 * Reference files followed (relative to src/appfl/):
 *   algorithm/aggregator/fedavg_aggregator.py     :322-368 (step), :276-296 (apply / plain mean)
 *   algorithm/aggregator/fedavgm_aggregator.py    :48-61
 *   algorithm/aggregator/fedadam_aggregator.py    :60-99
 *   algorithm/aggregator/fedyogi_aggregator.py    :60-97
 *   algorithm/aggregator/fedadagrad_aggregator.py :59-91
 *   algorithm/aggregator/fedasync_aggregator.py   :243-264, :156-160
 *   algorithm/aggregator/fedbuff_aggregator.py    :75-88, :145-158
 *   algorithm/aggregator/iiadmm_aggregator.py     :76-86, :137-162
 *   algorithm/aggregator/iceadmm_aggregator.py    :72-82, :133-166
 *   misc/memory_utils.py                          :95-131
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

enum { OPT_NONE = 0, OPT_AVGM = 1, OPT_ADAM = 2, OPT_YOGI = 3, OPT_ADAGRAD = 4 };

static inline float sign_f32(float d) { return (float)((0.0f < d) - (d < 0.0f)); }

/* fedavg_aggregator.py:360-367: diff = x - g; weighted = diff * w; step.add_(weighted). */
static inline float fold_clients(const float* const* x, const float* wf, int K, float g, size_t i) {
    float step = 0.0f;
    for (int k = 0; k < K; ++k) {
        float d = x[k][i] - g;
        float wd = d * wf[k];
        step = step + wd;
    }
    return step;
}

/*
 * Fused round: weighted step, server optimizer, apply.  w holds the Python-double client
 * weights; each is rounded to fp32 once (tensor * python_float).  step_out may be NULL.
 * Returns 0, or -1 for a bad opt code / K.
 */
int or_reduce_opt(const float* const* x, const double* w, int K, float* g, float* m, float* v,
                  size_t n, int opt, double lr, double beta1, double beta2, double tau,
                  float* step_out) {
    if (K <= 0 || K > 65536 || opt < OPT_NONE || opt > OPT_ADAGRAD) return -1;
    float wf[K];
    for (int k = 0; k < K; ++k) wf[k] = (float)w[k];
    const float lr_f = (float)lr, b1 = (float)beta1, omb1 = (float)(1.0 - beta1);
    const float b2 = (float)beta2, omb2 = (float)(1.0 - beta2), tau_f = (float)tau;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        const float gi = g[i];
        float step = fold_clients(x, wf, K, gi, i);
        if (opt == OPT_AVGM) {
            /* fedavgm_aggregator.py:52-61 */
            float mt = v[i] * b1;
            v[i] = mt + step;
            step = v[i];
        } else if (opt != OPT_NONE) {
            /* fedadam_aggregator.py:64-73 (same first moment in Yogi / Adagrad) */
            float mt = m[i] * b1;
            float st = step * omb1;
            m[i] = mt + st;
            float s2 = step * step;
            if (opt == OPT_ADAM) {
                /* fedadam_aggregator.py:76-86 */
                float vt = v[i] * b2;
                float sq = s2 * omb2;
                v[i] = vt + sq;
            } else if (opt == OPT_YOGI) {
                /* fedyogi_aggregator.py:76-84 */
                float sg = sign_f32(v[i] - s2);
                float yu = s2 * sg;
                float ys = yu * omb2;
                v[i] = v[i] - ys;
            } else {
                /* fedadagrad_aggregator.py:75-78 */
                v[i] = v[i] + s2;
            }
            /* fedadam_aggregator.py:89-99 */
            float num = lr_f * m[i];
            float den = sqrtf(v[i]) + tau_f;
            step = num / den;
        }
        if (step_out) step_out[i] = step;
        g[i] = gi + step; /* fedavg_aggregator.py:281-283 */
    }
    return 0;
}

/* Optimizer epilogue alone on an averaged step (same formulas as or_reduce_opt). */
int or_server_opt_step(float* step, float* m, float* v, size_t n, int opt, double lr, double beta1,
                       double beta2, double tau) {
    if (opt == OPT_NONE) return 0;
    if (opt < OPT_NONE || opt > OPT_ADAGRAD) return -1;
    const float lr_f = (float)lr, b1 = (float)beta1, omb1 = (float)(1.0 - beta1);
    const float b2 = (float)beta2, omb2 = (float)(1.0 - beta2), tau_f = (float)tau;
    for (size_t i = 0; i < n; ++i) {
        float st = step[i];
        if (opt == OPT_AVGM) {
            float mt = v[i] * b1;
            v[i] = mt + st;
            step[i] = v[i];
            continue;
        }
        float mt = m[i] * b1;
        float sm = st * omb1;
        m[i] = mt + sm;
        float s2 = st * st;
        if (opt == OPT_ADAM) {
            float vt = v[i] * b2;
            float sq = s2 * omb2;
            v[i] = vt + sq;
        } else if (opt == OPT_YOGI) {
            float sg = sign_f32(v[i] - s2);
            float yu = s2 * sg;
            float ys = yu * omb2;
            v[i] = v[i] - ys;
        } else {
            v[i] = v[i] + s2;
        }
        float num = lr_f * m[i];
        float den = sqrtf(v[i]) + tau_f;
        step[i] = num / den;
    }
    return 0;
}

/* Step only: fedavg_aggregator.py:322-368. */
int or_reduce_step(const float* const* x, const double* w, int K, const float* g, float* step_out,
                   size_t n) {
    if (K <= 0 || K > 65536) return -1;
    float wf[K];
    for (int k = 0; k < K; ++k) wf[k] = (float)w[k];
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) step_out[i] = fold_clients(x, wf, K, g[i], i);
    return 0;
}

void or_apply_step(float* g, const float* step, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) g[i] = g[i] + step[i];
}

/* fedavg_aggregator.py:285-295: zeros, add each client, div by K (memory_utils.py:127-128). */
void or_mean_f32(const float* const* x, int K, float* g, size_t n) {
    const float kf = (float)K;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        float s = 0.0f;
        for (int k = 0; k < K; ++k) s = s + x[k][i];
        g[i] = s / kf;
    }
}

/* Integer branch: torch.div(int, K) is a float32 true division, .type(int) truncates
 * (memory_utils.py:95-110). */
void or_mean_i64(const int64_t* const* x, int K, int64_t* g, size_t n) {
    const float kf = (float)K;
    for (size_t i = 0; i < n; ++i) {
        int64_t s = 0;
        for (int k = 0; k < K; ++k) s = s + x[k][i];
        g[i] = (int64_t)((float)s / kf);
    }
}

void or_mean_i32(const int32_t* const* x, int K, int32_t* g, size_t n) {
    const float kf = (float)K;
    for (size_t i = 0; i < n; ++i) {
        int32_t s = 0;
        for (int k = 0; k < K; ++k) s = (int32_t)((uint32_t)s + (uint32_t)x[k][i]);
        g[i] = (int32_t)((float)s / kf);
    }
}

/* fedcompass_aggregator.py:152-338 / fedqueue_aggregator.py:75-131.
 * mode 0: g = sum x*w;  1: g = g*coeff + sum x*w;  2: g = g - x0*w0 - x1*w1 ...;
 * 3 (K = 1): g = g*coeff + x0*w0 with no zero accumulator (single FedCompass arrival). */
int or_weighted_mix(const float* const* x, const double* w, int K, float* g, double coeff, int mode,
                    size_t n) {
    if (K <= 0 || K > 65536 || mode < 0 || mode > 5 || (mode == 3 && K != 1)) return -1;
    float wf[K];
    for (int k = 0; k < K; ++k) wf[k] = (float)w[k];
    const float c = (float)coeff;
    for (size_t i = 0; i < n; ++i) {
        float acc = (mode == 2) ? g[i] : 0.0f;
        if (mode >= 3) {
            /* modes 4 / 5 (fedavg_aggregator.py:180-217): first product assigned, rest added */
            acc = x[0][i] * wf[0];
            for (int k = 1; k < K; ++k) {
                float xw = x[k][i] * wf[k];
                acc = acc + xw;
            }
            if (mode == 4) acc = g[i] + acc;
        } else {
            for (int k = 0; k < K; ++k) {
                float xw = x[k][i] * wf[k];
                acc = (mode == 2) ? acc - xw : acc + xw;
            }
        }
        if (mode == 1 || mode == 3) {
            float gt = g[i] * c;
            acc = gt + acc;
        }
        g[i] = acc;
    }
    return 0;
}

/* mode 0: alpha*(x-g); mode 1: 0 - alpha*x (sub_ with alpha on zeros); mode 2: alpha*(-x). */
static inline float async_step(float x, float g, float alpha, int mode) {
    if (mode == 0) {
        float d = x - g;
        return alpha * d;
    }
    if (mode == 1) {
        float ax = alpha * x;
        return 0.0f - ax;
    }
    float nx = -x;
    return alpha * nx;
}

static inline int async_mode(int gradient_based, int sub_form) {
    return gradient_based ? (sub_form ? 1 : 2) : 0;
}

/* fedasync_aggregator.py:243-264 then :156-160. */
void or_async_mix(const float* x, double alpha_t, int gradient_based, int sub_form, float* g,
                  float* step_out, size_t n) {
    const float alpha = (float)alpha_t;
    const int mode = async_mode(gradient_based, sub_form);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        float s = async_step(x[i], g[i], alpha, mode);
        if (step_out) step_out[i] = s;
        g[i] = g[i] + s;
    }
}

/* fedbuff_aggregator.py:145-158 (+ :75-88 when flush != 0). */
void or_buff_accum(const float* x, double alpha_t, int gradient_based, float* g, float* step,
                   int first, int flush, size_t n) {
    const float alpha = (float)alpha_t;
    const int mode = gradient_based ? 2 : 0;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        float s = first ? 0.0f : step[i];
        float t = async_step(x[i], g[i], alpha, mode);
        s = s + t;
        step[i] = s;
        if (flush) g[i] = g[i] + s;
    }
}

void or_buff_accum_raw_f32(const float* x, float* step, int first, size_t n) {
    for (size_t i = 0; i < n; ++i) step[i] = (first ? 0.0f : step[i]) + x[i];
}
void or_buff_accum_raw_i64(const int64_t* x, int64_t* step, int first, size_t n) {
    for (size_t i = 0; i < n; ++i) step[i] = (first ? 0 : step[i]) + x[i];
}
void or_buff_accum_raw_i32(const int32_t* x, int32_t* step, int first, size_t n) {
    for (size_t i = 0; i < n; ++i)
        step[i] = (int32_t)((uint32_t)(first ? 0 : step[i]) + (uint32_t)x[i]);
}

/* fedbuff_aggregator.py:84-86: torch.div(step, K).type(dtype). */
void or_buff_flush_mean_f32(float* g, const float* step, int K, size_t n) {
    const float kf = (float)K;
    for (size_t i = 0; i < n; ++i) g[i] = step[i] / kf;
}
void or_buff_flush_mean_i64(int64_t* g, const int64_t* step, int K, size_t n) {
    const float kf = (float)K;
    for (size_t i = 0; i < n; ++i) g[i] = (int64_t)((float)step[i] / kf);
}
void or_buff_flush_mean_i32(int32_t* g, const int32_t* step, int K, size_t n) {
    const float kf = (float)K;
    for (size_t i = 0; i < n; ++i) g[i] = (int32_t)((float)step[i] / kf);
}

/*
 * ADMM server updates on flat vectors (iiadmm_aggregator.py:76-86,137-162;
 * iceadmm_aggregator.py:72-82,133-166).  K clients in call order; penalty holds the Python
 * doubles, each rounded to fp32 where it multiplies a tensor.  prev may be NULL (first round:
 * no dual residual).  res[0] / res[1] receive the primal / dual residual norms; the squares are
 * fp32 like torch.square's, their sums float64 (the reference sums in torch's own order, so the
 * norms are compared under a tolerance).  alg 0: IIADMM (dual rows updated in place, g =
 * sum(primal - (1/penalty)*dual) / num_clients), alg 1: ICEADMM (g = sum((penalty/total)*primal +
 * (1/total)*dual)).
 */
int or_admm(int alg, const float* const* primal, float* const* dual, const float* const* prev,
            const double* penalty, int K, int num_clients, float* g, size_t n, double* res) {
    if (K <= 0 || K > 65536 || alg < 0 || alg > 1) return -1;
    float rho[K], c1[K];
    double total = 0.0;
    for (int k = 0; k < K; ++k) total += penalty[k];
    for (int k = 0; k < K; ++k) {
        rho[k] = (float)penalty[k];
        c1[k] = alg == 0 ? (float)(1.0 / penalty[k]) : (float)(penalty[k] / total);
    }
    const float c2 = (float)(1.0 / total), nc = (float)num_clients;
    double prim = 0.0, du2 = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : prim, du2)
    for (size_t i = 0; i < n; ++i) {
        const float gi = g[i];
        float acc = 0.0f, r = 0.0f;
        for (int k = 0; k < K; ++k) {
            const float x = primal[k][i];
            const float d = gi - x;
            const float d2 = d * d;
            prim += (double)d2;
            if (alg == 0) {
                const float pd = rho[k] * d;
                const float du = dual[k][i] + pd;
                dual[k][i] = du;
                const float t = c1[k] * du;
                const float u = x - t;
                acc = acc + u;
            } else {
                const float a = c1[k] * x;
                const float b = c2 * dual[k][i];
                const float s = a + b;
                acc = acc + s;
            }
            if (prev) {
                const float dp = prev[k][i] - x;
                const float t = rho[k] * dp;
                r = r + t;
            }
        }
        g[i] = alg == 0 ? acc / nc : acc;
        const float r2 = r * r;
        du2 += (double)r2;
    }
    res[0] = sqrt(prim);
    res[1] = prev ? sqrt(du2) : 0.0;
    return 0;
}

/* Counter-based synthetic inputs shared with the device generator. */
static inline float synth_noise(uint64_t seed, uint64_t i) {
    uint64_t z = seed * 0x9E3779B97F4A7C15ull + i;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    int s = (int)(z & 0xFFFF) + (int)((z >> 16) & 0xFFFF) + (int)((z >> 32) & 0xFFFF) +
            (int)((z >> 48) & 0xFFFF);
    return (float)(s - 131070) * 2.6428998e-05f;
}

void or_synth_fill(float* p, const float* base, size_t n, uint64_t seed, uint64_t first,
                   float scale) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        float t = scale * synth_noise(seed, first + i);
        p[i] = base ? base[i] + t : t;
    }
}
