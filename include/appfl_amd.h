/*
 * appfl_amd.h -- C ABI of libappfl_amd.so: the MI355X (gfx950) kernels behind APPFL's
 * server-side Aggregator (FedAvg family, server optimizers, FedAsync / FedBuff mixing).
 *
 * Every pointer named x / g / m / v / step is DEVICE memory unless the comment says "host".
 * Nothing here allocates, frees or synchronises: the caller owns every buffer, every call only
 * enqueues work on `stream` (a hipStream_t passed as void*; NULL = the default stream) and
 * returns.  Calls on one stream execute in order.  The library is thread-compatible
 * (concurrent calls must use different buffers); the only device state it owns are two
 * static arrays per GPU, resident from the moment the library's code object is loaded on that
 * GPU: 16 KiB of meeting counters its persistent reduce kernel uses for timing only (a slot per
 * stream; every launch tags the slot and restarts a counter it finds with another tag, so
 * nothing is cleared between launches and a counter that an aborted or overlapping launch left
 * inconsistent is good again with the next launch; results never depend on them) and the
 * workspace of fa_reduce_opt_static.
 *
 * Arithmetic contract: each documented formula is evaluated as separately rounded IEEE fp32
 * operations in the written order (no FMA contraction, correctly rounded / and sqrt, denormals
 * kept), and client contributions are accumulated in client order k = 0..K-1.  That is the op
 * sequence of the PyTorch CPU code the reference runs, so results are bit-identical to it.
 * Host scalars arrive as double and are rounded to fp32 once, exactly where PyTorch rounds a
 * Python float that multiplies an fp32 tensor.
 *
 * Return value: FA_OK (0) or a negative code; fa_error_string() names it.
 *
 * Paths below are relative to the reference checkout (src/appfl/...).
 */
#ifndef APPFL_AMD_H
#define APPFL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* fa_stream_t;

enum {
    FA_OK = 0,
    FA_ERR_INVALID = -1,   /* NULL pointer, K <= 0, unknown opt code, ...                */
    FA_ERR_ALIGN = -2,     /* a buffer is not aligned to its element size                */
    FA_ERR_WORKSPACE = -3, /* K > FA_KCHUNK needs a step workspace and none was given     */
    FA_ERR_TOO_LARGE = -4, /* n beyond the library-owned workspace of fa_reduce_opt_static */
    FA_ERR_HIP = -1000     /* FA_ERR_HIP - hipError_t: a HIP runtime call failed          */
};

/* Clients folded by one kernel launch; more clients run as several launches that carry the
 * partial step through the caller's workspace (same client order, same bits). */
#define FA_KCHUNK 128

/* Server optimizer applied to the averaged pseudo-gradient. */
enum {
    FA_OPT_NONE = 0,    /* FedAvg:     algorithm/aggregator/fedavg_aggregator.py:276-283      */
    FA_OPT_AVGM = 1,    /* FedAvgM:    algorithm/aggregator/fedavgm_aggregator.py:48-61       */
    FA_OPT_ADAM = 2,    /* FedAdam:    algorithm/aggregator/fedadam_aggregator.py:60-99       */
    FA_OPT_YOGI = 3,    /* FedYogi:    algorithm/aggregator/fedyogi_aggregator.py:60-97       */
    FA_OPT_ADAGRAD = 4  /* FedAdagrad: algorithm/aggregator/fedadagrad_aggregator.py:59-91    */
};

int fa_version(void);
const char* fa_error_string(int code);
/* Number of clients one launch folds (FA_KCHUNK), for hosts that size workspaces. */
int fa_kchunk(void);
/* Name of the device kernel fa_reduce_opt launches for K clients of n 16-byte-aligned fp32
 * elements and optimizer `opt` (FA_OPT_*) on the current device (profilers list it under this
 * name); no counterpart in the reference. */
const char* fa_reduce_kernel_name(int K, size_t n, int opt);
/*
 * Benchmark hook (no counterpart in the reference): selects the launch shape of the 128-bit
 * reduce kernels -- vectors per thread (1|2|4; 0 = chosen per call, the default), clients
 * loaded ahead (4|8|16; ignored when vectors per thread is 0), nt (bit 0:
 * non-temporal client loads, bit 1: plain instead of write-through result stores, bit 2: the
 * persistent kernel gives each block one contiguous range instead of every gridDim-th tile,
 * bit 4: the persistent kernel's blocks do not meet before their result bursts, bit 5: they
 * also meet after the bursts, bits 6-7: how its blocks are grouped into staggered write phases
 * -- 0 odd / even XCDs (the default), 1 one chip-wide group, 2 four groups, 3 XCDs 0-3 / 4-7,
 * bits 8-12: tiles per phase, 0 = chosen per launch),
 * grid cap (<= 0: default) of the tile-per-block kernel, and from how many 16-KiB tiles per CU
 * a launch with a long client list (FedAvg / FedAvgM >= 48 clients; two-moment optimizers >= 40
 * clients on vectors of >= 8 tiles per CU, >= 32 on >= 32 tiles, >= 24 on >= 40 tiles) takes the persistent
 * kernel that parks results in LDS and writes them in bursts (stash: 0 never, < 0 always for
 * more than 8 clients, default 2).  Plain FedAvg from 6 clients on vectors of >= 72 tiles per
 * CU takes the persistent kernel's wide-batch form (16 clients x 4 vectors in flight per lane,
 * phases of ten tiles); stash == -2 forces that form at every length and client count, stash ==
 * -1 and 0 keep it off.  The two-moment optimizers at mid-size client lists (>= 24 clients on
 * vectors of >= 16 tiles per CU, >= 20 on >= 32 tiles, >= 16 on >= 56 tiles, below the client
 * counts above) take a persistent kernel that parks the averaged step alone and runs the
 * optimizer epilogue at the end of each phase of six tiles; stash == -3 forces that kernel for
 * every optimizer, length and client count, stash <= 0 otherwise keeps it off.  Results do not
 * depend on any of this.
 * Applies to launches made by the calling thread only (thread-local state).
 */
int fa_tune(int vpt, int unroll, int nt, int max_blocks, int stash);

/*
 * Weighted FedAvg step fused with the apply and the server-optimizer epilogue.
 * Replaces FedAvgAggregator.compute_steps + the "global += step" loop
 * (algorithm/aggregator/fedavg_aggregator.py:322-368 and :276-283) and, for opt != NONE, the
 * subclass compute_steps bodies listed at the FA_OPT_* codes.  Per element i:
 *     step = 0;  for k in 0..K-1:  step = step + (x[k][i] - g[i]) * (float)w[k]
 *     AVGM:    v = v*b1 + step;                         step = v
 *     ADAM:    m = m*b1 + step*(float)(1-b1);  v = v*b2 + (step*step)*(float)(1-b2)
 *     YOGI:    m as ADAM;  s2 = step*step;  v = v - (s2*sign(v - s2))*(float)(1-b2)
 *     ADAGRAD: m as ADAM;  v = v + step*step
 *     ADAM/YOGI/ADAGRAD:   step = ((float)lr*m) / (sqrt(v) + (float)tau)
 *     g[i] = g[i] + step
 * x: HOST array of K device pointers (each n floats); w: HOST array of K client weights.
 * m, v: moment vectors (n floats each; m unused for AVGM, both unused for NONE -> may be NULL).
 * step_out: optional (may be NULL) n-float buffer that receives the final step (the value the
 * reference leaves in self.step before the apply).
 * step_ws: n-float workspace, only touched when K > FA_KCHUNK (may be NULL otherwise).
 * Buffers must not overlap except where stated; 16-byte alignment of every buffer selects the
 * 128-bit path, anything else falls back to a scalar kernel with the same results.
 */
int fa_reduce_opt(const float* const* x, const double* w, int K, float* g, float* m, float* v,
                  size_t n, int opt, double lr, double beta1, double beta2, double tau,
                  float* step_out, float* step_ws, fa_stream_t stream);

/*
 * fa_reduce_opt with a per-call routing word (no counterpart in the reference).  flags = 0 is
 * fa_reduce_opt.  FA_ROUTE_NO_PERSISTENT keeps this call off the two persistent kernels (one
 * block per CU, all of a CU's LDS and registers each) whatever the length and client count, so
 * that it runs as the tile-per-block kernel of many small blocks: for callers that run the
 * round BESIDE other kernels on the same GPU -- the sharded aggregator's rounds next to RCCL's
 * all-gather of the previous block row -- where a one-block-per-CU grid would wait for the other
 * kernel's blocks to leave.  Same results bit for bit; unknown flag bits return FA_ERR_INVALID.
 */
enum { FA_ROUTE_DEFAULT = 0, FA_ROUTE_NO_PERSISTENT = 1 };
int fa_reduce_opt_ex(const float* const* x, const double* w, int K, float* g, float* m, float* v,
                     size_t n, int opt, double lr, double beta1, double beta2, double tau,
                     float* step_out, float* step_ws, unsigned flags, fa_stream_t stream);

/*
 * Test hook (no counterpart in the reference): overwrites every meeting counter of the current
 * GPU with garbage on `stream`, as an aborted launch might leave them.  The launches that follow
 * must run at their usual speed (they restart the counters they use); results never depended on
 * the counters in the first place.
 */
int fa_debug_scribble_meeting_counters(fa_stream_t stream);

/*
 * One whole synchronous round over K client ROWS of one layout: fa_reduce_opt over the weighted
 * fp32 entries (x[k] = start of client k's weighted segment, n floats) plus the plain mean --
 * fa_mean_f32 / fa_mean_i64 / fa_mean_i32 below -- of every segment listed in `means`, found
 * `row_offset` bytes after x[k] in every client's row.  Replaces the whole loop body of
 * FedAvgAggregator.aggregate (algorithm/aggregator/fedavg_aggregator.py:273-296: step-path
 * entries at :276-283, bypassing entries at :285-296).  Results are those of the separate
 * calls, bit for bit; the point is the launch count: small segments (BatchNorm statistics,
 * num_batches_tracked counters) ride on spare blocks of the reduce launch instead of taking
 * launches of their own, which for small and mid-size models cost more than the work.  The
 * library folds them when K <= FA_KCHUNK, n_means <= 3 and they are few enough elements
 * (some thousands); otherwise it enqueues the separate mean kernels itself.
 * means: HOST array of n_means descriptors (may be NULL when n_means == 0); out: device buffer
 * of n elements of `dtype`.
 * k_ready / rest_event: for a caller that is still uploading the last clients' rows when it
 * starts the round.  With 0 < k_ready < K the rows of clients 0..k_ready-1 must be complete;
 * the library folds them at once in a launch of their own (the partial step travels through
 * step_ws, which is then required: same client order, same bits, as for K > FA_KCHUNK), makes
 * `stream` wait for rest_event (a hipEvent_t recorded after the remaining rows' transfers;
 * NULL: the caller has ordered them before `stream` itself) and only then folds the rest and
 * takes the means -- so the bulk of the round overlaps the last transfers.  k_ready = 0 or K:
 * one fused launch, rest_event ignored.
 */
enum { FA_DT_F32 = 0, FA_DT_I64 = 1, FA_DT_I32 = 2 };
typedef struct fa_mean_segment {
    size_t row_offset; /* bytes from x[k] to this segment in client k's row */
    void* out;         /* device: receives the n means                       */
    size_t n;          /* elements                                           */
    int dtype;         /* FA_DT_*                                            */
} fa_mean_segment;
int fa_round(const float* const* x, const double* w, int K, float* g, float* m, float* v,
             size_t n, int opt, double lr, double beta1, double beta2, double tau, float* step_out,
             float* step_ws, const fa_mean_segment* means, int n_means, int k_ready,
             void* rest_event, fa_stream_t stream);

/*
 * fa_reduce_opt for callers that cannot allocate device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.
 * Same arithmetic, arguments and results as fa_reduce_opt, except that the carry workspace a
 * round of K > FA_KCHUNK clients needs (when step_out is NULL) is not passed in: it is a
 * library-owned array of FA_STATIC_WS_FLOATS floats (128 MiB) that every GPU holds from the
 * moment the library's code object is loaded on it until the process ends -- in every process
 * that loads the library, whether it ever calls this entry point or not.  The call allocates
 * and frees nothing.
 * K > FA_KCHUNK with step_out == NULL and n > FA_STATIC_WS_FLOATS returns FA_ERR_TOO_LARGE
 * before anything is enqueued.  All calls on one GPU share that GPU's array (its contents are
 * whatever the last call left; every launch writes its part before reading it, so nothing has
 * to be cleared): calls that overlap in time on one GPU -- different streams or threads --
 * race on it and must be serialised by the caller.
 */
#define FA_STATIC_WS_FLOATS (32u << 20)
int fa_reduce_opt_static(const float* const* x, const double* w, int K, float* g, float* m,
                         float* v, size_t n, int opt, double lr, double beta1, double beta2,
                         double tau, float* step_out, fa_stream_t stream);

/* FedAvg only: fa_reduce_opt with opt = FA_OPT_NONE and no moments. */
int fa_reduce_apply(const float* const* x, const double* w, int K, float* g, size_t n,
                    float* step_ws, fa_stream_t stream);

/* Same, for K client rows stored in one slab: x[k] = slab + k * stride (elements). */
int fa_reduce_opt_slab(const float* slab, size_t stride, const double* w, int K, float* g,
                       float* m, float* v, size_t n, int opt, double lr, double beta1,
                       double beta2, double tau, float* step_out, float* step_ws,
                       fa_stream_t stream);

/*
 * Step only (no apply, no optimizer): step_out[i] = sum_k (x[k][i] - g[i]) * (float)w[k].
 * Replaces FedAvgAggregator.compute_steps alone (fedavg_aggregator.py:322-368) for callers
 * that post-process self.step themselves.
 */
int fa_reduce_step(const float* const* x, const double* w, int K, const float* g,
                   float* step_out, size_t n, fa_stream_t stream);

/*
 * The weighted step of a round folded a few clients at a time, as they arrive, and the round
 * finished from there.  The reference folds all clients inside aggregate()
 * (fedavg_aggregator.py:359-368, after SyncScheduler.schedule has collected them,
 * scheduler/sync_scheduler.py:46-76); a server that knows the round's weights beforehand can
 * spend the wait for the stragglers on the early clients' terms instead.  The arithmetic is
 * the loop of fa_reduce_opt cut between two clients (the partial step is an fp32 value the loop
 * holds anyway), so the results are the same bits as one call over all K clients.
 *
 * fa_reduce_step_resume: `step` holds the partial step of clients 0..k_done-1 (ignored and
 * overwritten when k_done == 0); on return it holds that of clients 0..K-1.  x, w: all K entries
 * (those before k_done are not read from the device).  g is only read.
 * fa_round_resume: fa_round whose weighted step starts from `step_ws` = the partial step of
 * clients 0..k_done-1 (0 <= k_done < K; k_done == 0: fa_round without k_ready); the plain means
 * still read all K rows.
 */
int fa_reduce_step_resume(const float* const* x, const double* w, int K, int k_done,
                          const float* g, float* step, size_t n, fa_stream_t stream);
int fa_round_resume(const float* const* x, const double* w, int K, int k_done, float* g, float* m,
                    float* v, size_t n, int opt, double lr, double beta1, double beta2, double tau,
                    float* step_ws, const fa_mean_segment* means, int n_means,
                    fa_stream_t stream);

/*
 * A round whose arguments stay on the library's side between launches (no counterpart in the
 * reference): a steady federation repeats the same fa_round call -- same rows of the same slab,
 * same weights, same state vectors -- every round, and for a small model (the MNIST CNN of the
 * reference's examples, tests/resources/model/cnn.py:6-37) handing the 19 arguments across the
 * FFI boundary again costs a host as much as the launch itself.  fa_round_plan_create copies x, w
 * and the mean segments (the device buffers they point to stay the caller's and must outlive the
 * plan); fa_round_plan_launch(plan, stream) is fa_round(...stored arguments..., step_out = NULL,
 * k_ready = 0, stream): same routing, same kernels, same bits.  A plan may be launched any number
 * of times, from one thread at a time.
 */
typedef struct fa_round_plan fa_round_plan;
int fa_round_plan_create(const float* const* x, const double* w, int K, float* g, float* m, float* v,
                         size_t n, int opt, double lr, double beta1, double beta2, double tau,
                         float* step_ws, const fa_mean_segment* means, int n_means,
                         fa_round_plan** plan);
int fa_round_plan_launch(const fa_round_plan* plan, fa_stream_t stream);
void fa_round_plan_destroy(fa_round_plan* plan);

/*
 * Server-optimizer epilogue alone, for callers that computed the averaged step themselves:
 * step (in: averaged pseudo-gradient, out: optimizer step), m, v updated in place with the
 * formulas of fa_reduce_opt.  Replaces the part of the subclass compute_steps bodies after
 * super().compute_steps() (fedavgm_aggregator.py:48-61, fedadam_aggregator.py:60-99,
 * fedyogi_aggregator.py:60-97, fedadagrad_aggregator.py:59-91).  opt = FA_OPT_NONE is a no-op.
 */
int fa_server_opt_step(float* step, float* m, float* v, size_t n, int opt, double lr,
                       double beta1, double beta2, double tau, fa_stream_t stream);

/* g[i] = g[i] + step[i]  (fedavg_aggregator.py:281-283; fedbuff_aggregator.py:88). */
int fa_apply_step(float* g, const float* step, size_t n, fa_stream_t stream);

/*
 * Plain mean of entries that bypass the weighted step (integer tensors, and non-parameter
 * entries when a model was supplied): fedavg_aggregator.py:285-295 with
 * misc/memory_utils.py:95-110.   s = 0; for k: s = s + x[k][i];
 *   f32: g = s / (float)K        i64/i32: g = (int)trunc((float)s / (float)K)
 */
int fa_mean_f32(const float* const* x, int K, float* g, size_t n, fa_stream_t stream);
int fa_mean_i64(const int64_t* const* x, int K, int64_t* g, size_t n, fa_stream_t stream);
int fa_mean_i32(const int32_t* const* x, int K, int32_t* g, size_t n, fa_stream_t stream);

/*
 * Weighted mixes that do not subtract g first -- FedCompass and FedQueue
 * (algorithm/aggregator/fedcompass_aggregator.py:152-338, fedqueue_aggregator.py:75-131):
 *   acc = 0;  for k in 0..K-1:  acc = acc + x[k][i] * (float)w[k]
 *   FA_MIX_REPLACE:  g = acc                              (FedQueue, model-based)
 *   FA_MIX_CONVEX:   g = g * (float)coeff + acc           (FedCompass group; coeff = 1 - sum w)
 *   FA_MIX_SUBTRACT: for k: g = g - x[k][i] * (float)w[k] (gradient_based form of both)
 *   FA_MIX_CONVEX_ONE (K = 1): g = g * (float)coeff + x[0][i] * (float)w[0]  -- FedCompass's
 *                    single arrival, which adds to no zero accumulator (sign of a zero result)
 *   FA_MIX_SUM_ADD / FA_MIX_SUM_SET: acc = x[0]*w[0]; for k >= 1: acc = acc + x[k]*w[k];
 *                    g = g + acc  /  g = acc -- the flat secure-aggregation payload path
 *                    (algorithm/aggregator/fedavg_aggregator.py:180-217)
 * x, w: HOST arrays as in fa_reduce_opt.  acc_ws: n-float workspace, only needed when
 * K > FA_KCHUNK in the first two modes.
 */
enum {
    FA_MIX_REPLACE = 0, FA_MIX_CONVEX = 1, FA_MIX_SUBTRACT = 2, FA_MIX_CONVEX_ONE = 3,
    FA_MIX_SUM_ADD = 4, FA_MIX_SUM_SET = 5
};
int fa_weighted_mix(const float* const* x, const double* w, int K, float* g, double coeff,
                    int mode, float* acc_ws, size_t n, fa_stream_t stream);

/*
 * ADMM server updates on flat fp32 vectors: K clients in call order, rows of K primal vectors,
 * K dual vectors and (from the second round on) K previous primal vectors.
 *
 * fa_admm_ii -- IIADMM (algorithm/aggregator/iiadmm_aggregator.py:151-162; the server keeps the
 * duals): per element, for k in 0..K-1
 *     d = g - primal[k];  dual[k] = dual[k] + (float)penalty[k] * d        (written back)
 *     acc = acc + (primal[k] - (float)(1.0 / penalty[k]) * dual[k])
 * then g = acc / (float)num_clients   (num_clients: the client count of the first round, :54,162).
 * fa_admm_ice -- ICEADMM (algorithm/aggregator/iceadmm_aggregator.py:147-166; clients send their
 * duals): total = sum of penalty[] in client order (a double), per element
 *     acc = acc + ((float)(penalty[k] / total) * primal[k] + (float)(1.0 / total) * dual[k]);  g = acc
 * Both also produce the two residual norms of the round (iiadmm_aggregator.py:76-86,141-149;
 * iceadmm_aggregator.py:72-82,137-145), with the g of BEFORE the update:
 *     norms[0] = sqrt(sum_k sum_i (g - primal[k])^2)
 *     norms[1] = sqrt(sum_i (sum_k (float)penalty[k] * (prev[k] - primal[k]))^2)    (0 when prev == NULL)
 * The inner sums are fp32 in client order and the squares fp32, as the reference's tensor
 * operations are; the sums over i are float64 in a fixed order (the reference adds fp32 tensors
 * with torch.sum, whose order is its own: the norms agree to ~1e-6 relative, not bit for bit).
 *
 * primal, dual, prev, penalty: HOST arrays of K device pointers / K doubles.  g, norms (2
 * doubles), res_ws (FA_ADMM_RES_WS doubles) and carry_ws: device memory.  carry_ws: 2 * ((n + 3)
 * / 4 * 4) floats, only touched when K > FA_ADMM_KCHUNK (may be NULL otherwise).
 */
#define FA_ADMM_KCHUNK 64
#define FA_ADMM_BLOCKS 2048
#define FA_ADMM_RES_WS (2 + 2 * FA_ADMM_BLOCKS)
int fa_admm_ii(const float* const* primal, float* const* dual, const float* const* prev,
               const double* penalty, int K, int num_clients, float* g, size_t n, float* carry_ws,
               double* res_ws, double* norms, fa_stream_t stream);
int fa_admm_ice(const float* const* primal, const float* const* dual, const float* const* prev,
                const double* penalty, int K, float* g, size_t n, float* carry_ws, double* res_ws,
                double* norms, fa_stream_t stream);

/*
 * FedAsync arrival (algorithm/aggregator/fedasync_aggregator.py:243-264 and :156-160).
 *   gradient_based == 0:  step = (float)alpha_t * (x - g)
 *   gradient_based == 1:  step = 0 - (float)alpha_t * x      (sub_form == 1, optimize_memory)
 *                         step = (float)alpha_t * (-x)       (sub_form == 0)
 *   g = g + step
 * The two gradient_based forms differ only in the sign of a zero step.
 * step_out (may be NULL) receives the step.
 */
int fa_async_mix(const float* x, double alpha_t, int gradient_based, int sub_form, float* g,
                 float* step_out, size_t n, fa_stream_t stream);

/*
 * FedBuff arrival (algorithm/aggregator/fedbuff_aggregator.py:145-158): g stays frozen,
 *   step = step + (float)alpha_t * (x - g)     or, gradient_based:  step + (float)alpha_t*(-x)
 * first != 0 treats the incoming step as zero (the buffer was just reset, :116-124).
 */
int fa_buff_accum(const float* x, double alpha_t, int gradient_based, const float* g,
                  float* step, int first, size_t n, fa_stream_t stream);
/* Same arrival when it fills the buffer: also applies the flush g = g + step
 * (fedbuff_aggregator.py:75-88) in the same pass; step is left holding the flushed step. */
int fa_buff_accum_flush(const float* x, double alpha_t, int gradient_based, float* g,
                        float* step, int first, size_t n, fa_stream_t stream);
/* Entries FedBuff sums raw (fedbuff_aggregator.py:146-152): step = step + x. */
int fa_buff_accum_raw_f32(const float* x, float* step, int first, size_t n, fa_stream_t stream);
int fa_buff_accum_raw_i64(const int64_t* x, int64_t* step, int first, size_t n,
                          fa_stream_t stream);
int fa_buff_accum_raw_i32(const int32_t* x, int32_t* step, int first, size_t n,
                          fa_stream_t stream);
/* ... and flushes as a mean (fedbuff_aggregator.py:84-86): g = (T)((float)step / (float)K);
 * for f32 that is g = step / (float)K. */
int fa_buff_flush_mean_f32(float* g, const float* step, int K, size_t n, fa_stream_t stream);
int fa_buff_flush_mean_i64(int64_t* g, const int64_t* step, int K, size_t n,
                           fa_stream_t stream);
int fa_buff_flush_mean_i32(int32_t* g, const int32_t* step, int K, size_t n,
                           fa_stream_t stream);

/*
 * The same steps for float64 entries of a state dict.  The reference's float path takes any
 * float tensor (misc/memory_utils.py:112-131; the formulas are those cited at the fp32 entry
 * points of the same names); for float64 tensors PyTorch keeps Python scalars in double, so w,
 * alpha_t and the hyper-parameters enter as the doubles they are (no rounding to fp32), and
 * every operation is a separately rounded IEEE double operation in the same order.  Buffers
 * must be 8-byte aligned.
 */
int fa_reduce_opt_f64(const double* const* x, const double* w, int K, double* g, double* m, double* v,
                      size_t n, int opt, double lr, double beta1, double beta2, double tau,
                      double* step_out, double* step_ws, fa_stream_t stream);
int fa_reduce_step_f64(const double* const* x, const double* w, int K, const double* g, double* step_out,
                       size_t n, fa_stream_t stream);
int fa_server_opt_step_f64(double* step, double* m, double* v, size_t n, int opt, double lr, double beta1,
                           double beta2, double tau, fa_stream_t stream);
int fa_apply_step_f64(double* g, const double* step, size_t n, fa_stream_t stream);
int fa_mean_f64(const double* const* x, int K, double* g, size_t n, fa_stream_t stream);
int fa_weighted_mix_f64(const double* const* x, const double* w, int K, double* g, double coeff,
                        int mode, double* acc_ws, size_t n, fa_stream_t stream);
int fa_async_mix_f64(const double* x, double alpha_t, int gradient_based, int sub_form, double* g,
                     double* step_out, size_t n, fa_stream_t stream);
int fa_buff_accum_f64(const double* x, double alpha_t, int gradient_based, const double* g, double* step,
                      int first, size_t n, fa_stream_t stream);
int fa_buff_accum_flush_f64(const double* x, double alpha_t, int gradient_based, double* g, double* step,
                            int first, size_t n, fa_stream_t stream);
int fa_buff_accum_raw_f64(const double* x, double* step, int first, size_t n, fa_stream_t stream);
int fa_buff_flush_mean_f64(double* g, const double* step, int K, size_t n, fa_stream_t stream);
int fa_fill_f64(double* p, double value, size_t n, fa_stream_t stream);

/*
 * The same steps for bfloat16 / float16 entries of a state dict (the reference's float path takes
 * any float tensor, misc/memory_utils.py:112-131; formulas as cited at the fp32 entry points of
 * the same names).  On 16-bit float tensors PyTorch's CPU kernels evaluate each tensor operation
 * in fp32 and round its result to the tensor's type (nearest even); a Python scalar that
 * multiplies / divides enters as its fp32 value, one that is added (tau, tau**2) or passed as
 * `alpha=` of add_ / sub_ is rounded to the tensor's type first.  EVERY buffer here -- client
 * rows, g, m, v, steps, workspaces -- holds 16-bit elements of the type `half` names (n counts
 * elements); the kernels widen on load, round every operation's result to that type on the
 * device and narrow on store, so the buffers leave holding exactly the reference's values.
 * w, alpha_t, lr, beta* enter as (float) of the double; tau as that float rounded to the type.
 * Buffers must be 2-byte aligned (16-byte aligned buffers take 128-bit loads of 8 elements).
 */
enum fa_half {
    FA_HALF_BF16 = 1,   /* torch.bfloat16 */
    FA_HALF_F16 = 2     /* torch.float16  */
};
/* (float)x rounded to the 16-bit type, as a float (how PyTorch lands a Python scalar in a 16-bit
 * tensor: zeros_like(...) + tau**2, fedadam_aggregator.py:45-49). */
float fa_half_round_scalar(double x, int half);
/* Name of the device kernel fa_half_reduce_opt launches for K clients of n 16-byte-aligned 16-bit
 * elements and optimizer `opt` on the current device (fa_reduce_kernel_name for the 16-bit entry
 * points): "reduce_half_kernel", one tile per block, or "reduce_late_half_kernel", the persistent
 * kernel that parks a phase's averaged steps on chip and runs the optimizer epilogue of the phase
 * as one burst (long client lists; the environment variable
 * APPFL_AMD_HALF_LATE=0, a benchmark knob read at every launch, keeps every round off it). */
const char* fa_half_reduce_kernel_name(int K, size_t n, int opt, int half);
int fa_half_reduce_opt(const void* const* x, const double* w, int K, void* g, void* m, void* v,
                       size_t n, int opt, double lr, double beta1, double beta2, double tau,
                       void* step_out, void* step_ws, int half, fa_stream_t stream);
/* fa_half_reduce_opt with the per-call routing word of fa_reduce_opt_ex: FA_ROUTE_NO_PERSISTENT keeps
 * this call off the persistent kernel (for rounds that run beside other kernels on the same GPU);
 * same results bit for bit; unknown flag bits return FA_ERR_INVALID.  No counterpart in the reference. */
int fa_half_reduce_opt_ex(const void* const* x, const double* w, int K, void* g, void* m, void* v,
                          size_t n, int opt, double lr, double beta1, double beta2, double tau,
                          void* step_out, void* step_ws, int half, unsigned flags, fa_stream_t stream);
int fa_half_reduce_step(const void* const* x, const double* w, int K, const void* g, void* step_out,
                        size_t n, int half, fa_stream_t stream);
/* The client loop cut between two clients, as fa_reduce_step_resume / fa_round_resume cut it
 * (fedavg_aggregator.py:359-368): `step` / `step_ws` holds the partial step of clients
 * 0..k_done-1 (0 <= k_done < K; 0: nothing folded yet), the call folds the others -- and, for
 * fa_half_reduce_opt_resume, finishes the round (optimizer, apply).  Same order, same bits. */
int fa_half_reduce_step_resume(const void* const* x, const double* w, int K, int k_done, const void* g,
                               void* step, size_t n, int half, fa_stream_t stream);
int fa_half_reduce_opt_resume(const void* const* x, const double* w, int K, int k_done, void* g, void* m,
                              void* v, size_t n, int opt, double lr, double beta1, double beta2,
                              double tau, void* step_ws, int half, fa_stream_t stream);
int fa_half_server_opt_step(void* step, void* m, void* v, size_t n, int opt, double lr, double beta1,
                            double beta2, double tau, int half, fa_stream_t stream);
int fa_half_apply_step(void* g, const void* step, size_t n, int half, fa_stream_t stream);
int fa_half_mean(const void* const* x, int K, void* g, size_t n, int half, fa_stream_t stream);
/* FedCompass / FedQueue mixes (fedcompass_aggregator.py:152-338, fedqueue_aggregator.py:75-131):
 * fa_weighted_mix's modes, each x*w, each running sum and the final combine rounded. */
int fa_half_weighted_mix(const void* const* x, const double* w, int K, void* g, double coeff,
                         int mode, void* acc_ws, size_t n, int half, fa_stream_t stream);
/* sub_form != 0 is the reference's optimize_memory branch (fedasync_aggregator.py:243-258), where
 * alpha_t is the `alpha=` of add_ / sub_ and is therefore rounded to the 16-bit type -- for the
 * model-based step too; sub_form == 0 (:260-264) multiplies by the fp32 alpha_t. */
int fa_half_async_mix(const void* x, double alpha_t, int gradient_based, int sub_form, void* g,
                      void* step_out, size_t n, int half, fa_stream_t stream);
/* fedbuff_aggregator.py:145-158 (and, with flush != 0, :75-88 for the weighted entries). */
int fa_half_buff_accum(const void* x, double alpha_t, int gradient_based, void* g, void* step,
                       int first, int flush, size_t n, int half, fa_stream_t stream);
int fa_half_buff_accum_raw(const void* x, void* step, int first, size_t n, int half,
                           fa_stream_t stream);
int fa_half_buff_flush_mean(void* g, const void* step, int K, size_t n, int half,
                            fa_stream_t stream);
/* p[i] = fa_half_round_scalar(value): moment initialisation of a 16-bit model. */
int fa_half_fill(void* p, double value, size_t n, int half, fa_stream_t stream);

/* p[i] = value: moment initialisation (m = 0, v = (float)(tau**2), fedadam_aggregator.py:45-49). */
int fa_fill_f32(float* p, float value, size_t n, fa_stream_t stream);

/*
 * Synthetic client update, generated on the device with the counter-based integer PRNG the
 * tests and the benchmark share with the CPU oracle (no counterpart in the reference):
 *   noise(seed, i) = (sum of the four 16-bit fields of splitmix64(seed, i) - 131070) * c
 *   p[i] = base ? base[i] + scale * noise(seed, first + i) : scale * noise(seed, first + i)
 * `first` is the global index of p[0], so shards of one vector can be filled independently.
 */
int fa_synth_fill(float* p, const float* base, size_t n, uint64_t seed, uint64_t first,
                  float scale, fa_stream_t stream);

/*
 * Host-side packing of one client's state dict into a row of a (pinned) staging slab:
 * for t in 0..T-1: memcpy(dst + dst_off[t], src[t], nbytes[t]), spread over `threads` threads.
 * All pointers are HOST memory.  Stands where the reference materialises per-tensor CPU
 * tensors (agent/server.py:505-525) before the aggregator sees them.
 */
int fa_pack_host(void* dst, const void* const* src, const size_t* dst_off, const size_t* nbytes,
                 int T, int threads);
/*
 * fa_pack_host restricted to the destination byte window [win_lo, win_hi): lets a host pack a
 * large row chunk by chunk and start the transfer of each chunk while the next is being packed.
 */
int fa_pack_host_window(void* dst, const void* const* src, const size_t* dst_off,
                        const size_t* nbytes, int T, size_t win_lo, size_t win_hi, int threads);
/*
 * Threaded copy of `nbytes` from `src` (e.g. a pinned download buffer) into a FRESHLY allocated
 * pageable buffer `dst` whose pages have not been touched yet; asks the kernel for huge pages
 * first.  Last leg of handing the new global model back as fresh CPU tensors
 * (misc/memory_utils.py:15-36 clone_state_dict_optimized).
 */
int fa_host_copy_fresh(void* dst, const void* src, size_t nbytes, int threads);

/*
 * Threaded copy of `nbytes` from `src` to `dst` that also yields the CRC-32 (zip polynomial) of
 * every `chunk`-byte piece in the same pass: crcs[i] = crc32(src[i*chunk .. min((i+1)*chunk,
 * nbytes))), ceil(nbytes / chunk) values.  All pointers are HOST memory.  For writing the new
 * global model back into a torch.save-compatible archive, whose records carry a CRC-32
 * (comm/mpi/serializer.py:46-50 model_to_byte, comm/grpc/utils.py:20-24 serialize_model: both
 * torch.save, which checksums every record on one thread).
 */
int fa_host_copy_crc32(void* dst, const void* src, size_t nbytes, size_t chunk, uint32_t* crcs,
                       int threads);

#ifdef __cplusplus
}
#endif
#endif /* APPFL_AMD_H */
