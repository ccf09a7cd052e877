/*
 * flashgmm_amd.h — C ABI of libflashgmm_amd.so, the MI355X (gfx950) GMM entropy-coding path.
 *
 * This is the drop-in boundary for ONE path of tokkiwa/FlashGMM: what its pybind11 module `compressai.ans`
 * exposes for Gaussian-mixture conditionals, and the tensor preparation its Python entropy model does right
 * above that call.  Plain pointers and sizes only; no torch / pybind types.  All functions return an
 * fgmm_status (0 = OK) and never throw across the ABI.
 *
 * Reference interfaces replaced (paths relative to the reference repo):
 *   - RansEncoder::encode_with_indexes_gmm<4>   compressai/cpp_exts/rans/rans_interface.cpp:609-617 (-> :458-554, :557-585)
 *   - RansDecoder::decode_with_indexes_gmm<4>   compressai/cpp_exts/rans/rans_interface.cpp:766-883
 *   - pybind signatures / keyword names         compressai/cpp_exts/rans/rans_interface.cpp:977-1004, :1026-1035
 *   - GaussianMixtureConditional.compress       compressai/entropy_models/entropy_models.py:833-867
 *   - GaussianMixtureConditional.decompress     compressai/entropy_models/entropy_models.py:872-910
 *   - reshape_entropy_parameters (+clamp)       compressai/entropy_models/entropy_models.py:810-828
 *   - APPROX_MODE numbering                     compressai/cpp_exts/rans/rans_interface.cpp:224-232
 *
 * Division of labour (BASELINE.json north_star): every floating-point operation — the K=4 mixture CDF in one
 * of three Phi approximations, 16-bit quantisation, bypass detection, the decode-side per-latent edge tables —
 * runs in hand-written HIP kernels; the integer rANS state machine runs on host threads fed by pinned
 * hipMemcpyAsync copies of the GPU-built tables.  There is NO CPU implementation of the float work in this
 * library: with no usable HIP device every entry point that needs one returns FGMM_ERR_NO_DEVICE.
 *
 * Parameter addressing.  A mixture parameter of latent (channel c, position p), component k lives at
 *     base[k*stride_k + c*stride_c + p*stride_p]           (strides in ELEMENTS)
 *   - latent-codec layout [1, K*M, h, w] (channel = k*M + c): stride_k = M*h*w, stride_c = h*w, stride_p = 1
 *   - the reference's (n, K) accessor views:                 one channel, stride_p = stride(0), stride_k = stride(1)
 * `memspace` says where the parameter / symbol pointers live: FGMM_DEVICE pointers are used in place,
 * FGMM_HOST buffers are staged through pinned memory and copied to the GPU first (PCIe-inclusive path).
 */
#ifndef FLASHGMM_AMD_H
#define FLASHGMM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FGMM_ABI_VERSION 6 /* 6: + the parameter head (fgmm_head_*, fgmm_gmc_compress_head_batch); + fgmm_sink and the _to forms of the batched
                              compress calls; FGMM_WORKER_CPUS that cannot be honoured fails fgmm_ctx_create.  5: + fgmm_ctx_call_log; REMOVED (measured, lost, pruned): options tab_place /
                              tab_spin / copy_engine / dec_pair / dec_group, fgmm_rans_decode_tab2 + fgmm_tab_ref, fgmm_ctx_stat index 6 */

typedef enum {
  FGMM_OK = 0,
  FGMM_ERR_INVALID = 1,     /* bad argument (null pointer, K != 4, negative size, unknown mode ...) */
  FGMM_ERR_NO_DEVICE = 2,   /* no HIP device / HIP runtime error at context creation */
  FGMM_ERR_HIP = 3,         /* a HIP call failed; fgmm_last_error() has the text */
  FGMM_ERR_NOMEM = 4,
  FGMM_ERR_STREAM = 5,      /* bitstream shorter than the symbols it is asked to yield (corrupt input) */
  FGMM_ERR_UNSUPPORTED = 6  /* outside the built envelope (e.g. max_bs_value > FGMM_MAX_BS) */
} fgmm_status;

/* Phi approximation, numbered as the reference CODE numbers APPROX_MODE (the README swaps 1 and 2). */
typedef enum { FGMM_MODE_POLYA = 0, FGMM_MODE_AS = 1, FGMM_MODE_LOGISTIC = 2 } fgmm_mode;

typedef enum { FGMM_HOST = 0, FGMM_DEVICE = 1 } fgmm_memspace;

#define FGMM_K 4              /* the reference binds K = 4 only (rans_interface.cpp:60,982,1003,1033) */
#define FGMM_MAX_BS 1073741822 /* largest decoder half-width (abs_max + 1): 2^30 - 2, so that 2*max_bs + 2 fits an int32.
                                 Beyond 16382 headers are 8 bytes and rows are built by the generic two-pass kernels,
                                 which refuse (FGMM_ERR_UNSUPPORTED) a latent whose evaluation window — what is left of
                                 [-max_bs, max_bs+1] between the provably saturated tails — exceeds 2^20 edges */
#define FGMM_MAX_BS_H4 16382  /* largest half-width the 4-byte header form can carry */

typedef struct fgmm_ctx fgmm_ctx; /* one per process per GPU: streams, pinned staging, workspaces, host threads */

int fgmm_abi_version(void);
const char *fgmm_last_error(void); /* thread-local text of the last failure on this thread */

/* What this process may use of the host: *cpus_out = min(CPUs of its affinity mask, CPUs per period of its cgroup quota —
 * cpu.max of cgroup v2 / cpu.cfs_quota_us of v1, ancestors included); *affinity_out / *quota_out the two terms (quota < 0:
 * none).  Any out-pointer may be NULL. */
int fgmm_host_cpu_budget(double *cpus_out, int *affinity_out, double *quota_out);
/* Host rANS workers a context gets by default when `ranks_sharing` processes (one per GPU) share that budget: the share's CPUs,
 * floor(budget / ranks_sharing) - or, where a cgroup quota (CPU TIME per period) is what limits the budget and the affinity mask is
 * wider, up to FGMM_WORKERS_PER_CPU (environment, 1..4, default 3) workers per CPU of the share (never more than the share of the mask):
 * the workers sleep on the copies' events most of a call, so the bursts of a step run on as many cores while the quota is not exhausted
 * (48 workers under a 16-CPU quota use 14-15 CPUs' worth and are 5 % faster than 16: profiles/r05_stall_diagnosis.md).  Within [1, 48]. */
int fgmm_host_thread_budget(int ranks_sharing);

/* device < 0: current HIP device.  n_threads <= 0: fgmm_host_thread_budget(1) host rANS workers. */
int fgmm_ctx_create(int device, int n_threads, fgmm_ctx **out);
void fgmm_ctx_destroy(fgmm_ctx *ctx);
int fgmm_ctx_device(const fgmm_ctx *ctx);
int fgmm_ctx_threads(const fgmm_ctx *ctx);
/* Resizes the context's pool of host rANS workers in place (n_threads <= 0: the default); options, profiling state and
 * buffers are kept.  Must not race with a call on the same context (it takes the context's lock like every call). */
int fgmm_ctx_set_threads(fgmm_ctx *ctx, int n_threads);
/* Where the context's host workers may run, as a Linux cpulist ("" = wherever the thread that created the context may).  The workers
 * stream the decode-side tables through the L3 of the core complex they run on; a calling thread that shares that L3 - a Python
 * interpreter above all - pays for it between the calls (0.9 - 1.5 ms of glue per Kodak step instead of 0.5).  Decided when the context
 * is created, from the environment variable FGMM_WORKER_CPUS:
 *   unset      the creating thread's CPUs minus those that share an L3 with the CPU it is on (if >= 32 CPUs and two per worker remain)
 *   "inherit"  the creating thread's CPUs
 *   a cpulist  exactly these, e.g. "16-127" (what the kernel grants of them to a thread of this process: the list within the process's
 *              cpuset; fgmm_ctx_worker_cpus reports that).  A value that is not a cpulist, or of which no CPU is granted, makes
 *              fgmm_ctx_create fail with FGMM_ERR_INVALID - never a silent fall back onto the creating thread's L3
 * The library never changes the calling thread's own affinity; a caller that wants the full benefit keeps its thread on the CPUs
 * that are NOT in this list (bench.py does, for its timed regions).
 * The DECODE calls run on a second pool of as many workers confined to ONE hardware thread per core of that list (two sequential
 * decoders on one core's two hardware threads cost 14 % more CPU time than on two cores; the encode call wants every hardware thread):
 * FGMM_DECODE_SMT=1 in the environment keeps the decoders on the list as it is. */
int fgmm_ctx_worker_cpus(fgmm_ctx *ctx, char *cpulist_out, size_t cap);

void fgmm_free(void *p); /* releases any buffer this library returned through an out-pointer */
/* Moves `count` buffers this library returned (src[i], len[i] bytes: bitstreams of a batched compress) into the caller's own
 * storage dst[i] and releases them - the copies run on the context's host workers.  What a binding does instead of `count`
 * single-threaded copies when its language wants to own the bytes (Python: 30 MB of bitstreams of eight 4K images, 3 ms). */
int fgmm_ctx_take_buffers(fgmm_ctx *ctx, void *const *dst, void *const *src, const size_t *len, int count);

/* Tuning knobs of a context (defaults in brackets; what was measured behind each: DESIGN.md).  Unknown names return FGMM_ERR_INVALID.
 *   "pieces"      [0]   decode: the tables of every bitstream of a call reach the host in this many pieces, piece-major and shrinking
 *                       (at most 32; 0 = by the call's longest bitstream: 8 up to 147 k latents, 24 from 1.1 M on); the host workers take
 *                       (bitstream, piece) tasks once their tables have landed, the coder state travels with the bitstream from
 *                       worker to worker
 *   "dec_first"   [2]   decode: bitstreams in the first launch of the first round of pieces (doubling from there: first tables early)
 *   "enc_ways"    [0]   encode: consecutive bitstreams one worker codes symbol by symbol in turn (several dependency chains share a
 *                       core): 1..4; 0 = two when the call has more bitstreams than workers, else one
 *   "enc_segs"    [1]   encode: a call in which every bitstream has a host worker of its own (and whose tables are 4 MB and more) lays
 *                       every table out in four segments of compact channels, LAST SEGMENT FIRST across all bitstreams: rANS encodes
 *                       backwards, so the encoders follow the landing.  0 = whole tables, bitstream after bitstream
 *   "scatter_rounds" [1] decode: decoded symbols return to the GPU round by round (one launch per piece index) instead of one launch per
 *                       bitstream when it has finished
 *   "tab_cap_e"   [12288] single-pass table kernel: edges one block keeps in LDS (latents per block = cap / (2*max_bs+2)); at the
 *                       default a call with a half-width that only fits 16384 edges (383 < max_bs <= 511) gets that
 *   "stage_max_mb" [0]  decode: cap of the device staging area for rows in MiB (0: a quarter of the free device memory).  A launch
 *                       whose rows do not fit is re-run with the exact size its cursor reports
 *   "ef_rows"     [0]   decode: 0 = Elias-Fano rows when min(host workers, bitstreams of the call) >= 10 (PCIe is the bottleneck),
 *                       uint16 rows otherwise (the sequential host decoders are); 1 = always, 2 = never
 *   "ef_min"      [33]  decode: rows with at least this many entries are Elias-Fano coded (>= 14, the format's floor)
 *   "gpu_decode"  [0]   decode: checkpointed bitstreams (fgmm_ckpt) are decoded ON THE GPU, a workgroup per segment (no decode-side
 *                       tables at all); a bitstream with a segment the kernel does not settle goes through the table path.  0 = when
 *                       the call has enough segments for the GPU to be the faster decoder (estimated with rates measured on MI355X +
 *                       EPYC 9575F: on another host set 1 or 2), 1 = whenever a bitstream carries valid notes, 2 = never
 *   "ckpt_decode" [0]   decode, table path: checkpointed bitstreams are decoded in segments on all host workers: 0 = when the call
 *                       has fewer bitstreams than workers, 1 = always, 2 = never (the notes are ignored)
 *   "spin_lat"    [400000] decode: calls of at most this many latents wait on polled events instead of sleeping ones (-1: never)
 *   "enc_vec" / "enc_linear"  A/B switches of the encode-side kernel's load width and grid (0 / 1: the defaults)
 *   "trace"       [0]   1: phase timestamps of every batched call on stderr, 2: + per-bitstream job timeline */
int fgmm_ctx_set_option(fgmm_ctx *ctx, const char *name, int64_t value);
int fgmm_ctx_get_option(fgmm_ctx *ctx, const char *name, int64_t *value_out);
/* Releases the context's grown buffers (device workspace and staging, pinned host ranges); they grow again on demand. */
int fgmm_ctx_trim(fgmm_ctx *ctx);

/* Measurement aid (bench.py): when enabled, timing HIP events bracket the table kernels ON THE STREAM THEY ARE
 * LAUNCHED ON; fgmm_ctx_kernel_ms returns the duration of the most recent launch(es) of
 * which = 0: symtab kernel (encode-side CDF), 1: decode-side table kernels of the last call (all launches, first to
 * last), 2: quant_stats kernel, 3: unused (0). */
int fgmm_ctx_set_profiling(fgmm_ctx *ctx, int enable);
/* Counters of the most recent batched call: which = 0 encode tables copied D2H (bytes), 1 decode headers + block offsets
 * + rows copied D2H (bytes), 2 latents those decode tables describe, 3 edges the decode-side kernels evaluated, 4 bitstreams
 * the GPU's segment decoder decoded (checkpointed ones), 5 bitstreams it handed back to the table path. */
int fgmm_ctx_stat(fgmm_ctx *ctx, int which, uint64_t *out);
int fgmm_ctx_kernel_ms(fgmm_ctx *ctx, int which, float *ms_out);
/* The call log: phase marks of the context's most recent batched calls (at most 64 are kept), always recorded - a handful of
 * clock reads per call.  What a caller needs to tell WHICH part of a slow call was slow: the GPU and the bus (marks 1..3), the host
 * workers (mark 4 against mark 3, busy / wait), or the calling thread.  Times in milliseconds.
 *   kind 0 = batched encode: ms[0] kernels and table copies enqueued, [1] kernels done and side information on the host, [2] host
 *            jobs handed out, [3] last table (segment) seen landed by an encoder, [4] last encoder done, [5] call end
 *   kind 1 = batched decode, table path: ms[0] planned and buffers ensured, [1] first table copy queued, [2] last table copy queued,
 *            [3] last table piece seen landed by a decoder, [4] last decoder done, [5] call end (y_hat complete)
 *   kind 2 = batched decode on the GPU (checkpointed bitstreams): ms[0..2] enqueued, [3..4] segments decoded, [5] call end
 *   worker_busy_ms / worker_wait_ms: summed over the host jobs of the call - coding, and waiting for a table copy to land */
typedef struct {
  int32_t kind, count;   /* count: bitstreams of the call */
  double t_begin_ms;     /* steady clock, since the context was created */
  double ms[6];          /* since t_begin_ms */
  double worker_busy_ms, worker_wait_ms;
  double head_ms[3];     /* kind 1, the time before ms[1] in detail: [0] descriptors built, [1] first launches enqueued and the host
                            workers started, [2] the first launch's size seen on the host (its copy is queued next); else 0 */
} fgmm_call_marks;
/* out[0 .. *n_out) = the most recent min(cap, 64, calls so far) calls, oldest first */
int fgmm_ctx_call_log(fgmm_ctx *ctx, fgmm_call_marks *out, int cap, int *n_out);

/* ------------------------------------------------------------------------------------------------------------
 * 1. The reference's native boundary (compressai.ans), same argument meaning and order.
 * ---------------------------------------------------------------------------------------------------------- */

/* RansEncoder.encode_with_indexes_gmm(symbols, scales, means, weights, max_value) -> bytes
 * symbols: int32[n] contiguous.  scales/means/weights: (n, 4) float32 at [i*stride_n + k*stride_k].
 * max_value is accepted and ignored, as in the reference (rans_interface.cpp:462).
 * *out is malloc'ed by the library (fgmm_free); empty input yields the 8-byte flushed state. */
int fgmm_encode_with_indexes_gmm(fgmm_ctx *ctx, const int32_t *symbols, const float *scales, const float *means,
                                 const float *weights, int64_t n, int64_t stride_n, int64_t stride_k, int K,
                                 int mode, int memspace, int32_t max_value, uint8_t **out, size_t *out_len);

/* RansDecoder.decode_with_indexes_gmm(encoded, scales, means, weights, max_bs_value) -> int32[n]
 * out_symbols: HOST int32[n].  Yields exactly what the reference's float bisection yields, including on
 * non-monotone tables and on its pmf==0 fallback (rans_interface.cpp:865-875). */
int fgmm_decode_with_indexes_gmm(fgmm_ctx *ctx, const uint8_t *encoded, size_t encoded_len, const float *scales,
                                 const float *means, const float *weights, int64_t n, int64_t stride_n,
                                 int64_t stride_k, int K, int mode, int memspace, int32_t max_bs_value,
                                 int32_t *out_symbols);

/* ------------------------------------------------------------------------------------------------------------
 * 2. Entropy-model level, fused on the GPU: GaussianMixtureConditional.compress / .decompress for B = 1.
 *    All tensor pointers are DEVICE pointers in the latent codec's layout:
 *      y [M, h*w] float32;   scales/means/weights [K, M, h*w] via (stride_k, stride_c), position stride 1.
 *    `stream` is the hipStream_t the caller's producer kernels were enqueued on (NULL = default stream);
 *    the library orders its own work after it and returns with all outputs complete.
 * ---------------------------------------------------------------------------------------------------------- */

typedef enum { FGMM_F32 = 0, FGMM_F16 = 1 } fgmm_dtype;

typedef struct {
  const void *scales, *means, *weights;  /* device; float32 or IEEE float16 planes, see dtype */
  int64_t stride_k, stride_c;            /* elements */
  int32_t dtype;                         /* fgmm_dtype.  FGMM_F16 (BASELINE configs[4]): each value is widened to
                                            float32 exactly on load, then the float32 path runs unchanged — the result
                                            is that of the reference fed the widened values (the reference itself
                                            rejects half tensors: accessor<float,2>, rans_interface.cpp:478-480) */
  int32_t flags;                         /* FGMM_PARAMS_LOGITS: `weights` holds the parameter head's LOGITS; the kernels compute
                                            pi = softmax over K themselves (the pi plane is never
                                            written and read back), with one fixed binary32 sequence shared by the encode- and
                                            the decode-side kernel — streams coded this way decode this way, on any device;
                                            they differ from streams coded from torch.softmax's pi in the last bit of pi */
} fgmm_params;
#define FGMM_PARAMS_LOGITS 1

/* compress(y, scales, means, weights) -> ((bytes, abs_max, zero_bitmap), y_quantized)
 *   yq_out        device float32[M*hw]  = round(y) (round-half-even)                    entropy_models.py:839
 *   abs_max_out   max(int|max y|, int|min y|) + 1, floored at 1                         entropy_models.py:834-837
 *   zero_bitmap   HOST int64[M]: 1 where the channel has any non-zero quantised latent  entropy_models.py:840-842
 *   scales are clamped to [0.11, 256] when clamp_scales != 0                            entropy_models.py:817
 *   symbols are coded in (non-zero channel, h, w) order                                 entropy_models.py:844-845 */
int fgmm_gmc_compress(fgmm_ctx *ctx, void *stream, const float *y, const fgmm_params *params, int M, int K,
                      int64_t hw, int mode, int clamp_scales, float *yq_out, int32_t *abs_max_out,
                      int64_t *zero_bitmap_out, uint8_t **out, size_t *out_len);

/* decompress(strings, abs_max, zero_bitmap, scales, means, weights) -> y_hat
 *   y_hat_out     device float32[M*hw]: decoded symbols at the non-zero channels, 0 elsewhere (:903-908) */
int fgmm_gmc_decompress(fgmm_ctx *ctx, void *stream, const uint8_t *encoded, size_t encoded_len, int32_t abs_max,
                        const int64_t *zero_bitmap, const fgmm_params *params, int M, int K, int64_t hw, int mode,
                        int clamp_scales, float *y_hat_out);

/* Checkpoints: a SEEKABLE bitstream without touching the bitstream.  A rANS stream decodes sequentially - symbol i needs the
 * coder state symbol i-1 left - which makes ONE bitstream one host thread's work (1.1 - 1.5 ms for a Kodak half, 27 ms for
 * ELIC's largest group of a 4K image) however many threads idle.  The decoder's state before symbol i is the encoder's state
 * after it has encoded symbol i on its reversed walk, so the encoder can note (state, words the decoder has read by then)
 * every `stride` symbols, OUT OF BAND: `bytes` stays the reference's stream, byte for byte.  A decoder handed the notes
 * decodes the segments between them on as many host workers as it has; a stream without notes (the reference's own) decodes
 * sequentially as ever.  The notes are VERIFIED, never trusted: a segment must end exactly in the next note's state and
 * position - then, by induction from the stream's head, it did the sequential decoder's work; one mismatch and the whole
 * stream is decoded sequentially instead.  16 bytes per `stride` symbols (stride 4096: 0.3 % of a Kodak stream). */
typedef struct {
  uint64_t x;   /* coder state before decoding symbol (k + 1) * stride */
  uint64_t pos; /* 32-bit renormalisation words read by then (the stream's 8-byte head not counted) */
} fgmm_ckpt;

/* Batched forms: `count` independent streams (images / checkerboard halves) in one call.  Kernels for all
 * items are enqueued first on the context's HIP streams, tables come back by pinned async copies, and the host
 * worker threads run one rANS state machine per item.  Item i uses y[i], params[i], ... ; outputs as above. */
typedef struct {
  const float *y;          /* device [M*hw] (compress only) */
  fgmm_params params;
  int32_t M, K;
  int64_t hw;
  float *yq_out;           /* compress: device [M*hw];  decompress: y_hat device [M*hw] */
  int64_t *zero_bitmap;    /* HOST int64[M]: compress out / decompress in */
  int32_t abs_max;         /* compress out / decompress in */
  uint8_t *bytes;          /* compress: out (fgmm_free);  decompress: in */
  size_t bytes_len;
  int32_t status;          /* per-item fgmm_status */
  int32_t ckpt_stride;     /* compress in: note a checkpoint every this many symbols (a power of two >= 256; 0 = none).
                              decompress in: the stride `ckpt` was noted with (0 / ckpt NULL: sequential decode) */
  fgmm_ckpt *ckpt;         /* compress out (fgmm_free; NULL when none): (coded symbols - 1) / ckpt_stride entries;
                              decompress in */
  int64_t n_ckpt;          /* compress out / decompress in */
} fgmm_item;

/* A compress call that returns an error returns NO buffer (items[i].bytes / .ckpt are NULL, whatever items[i].status says): a binding
 * may raise on the status without releasing anything. */
int fgmm_gmc_compress_batch(fgmm_ctx *ctx, void *stream, fgmm_item *items, int count, int mode, int clamp_scales);
int fgmm_gmc_decompress_batch(fgmm_ctx *ctx, void *stream, fgmm_item *items, int count, int mode, int clamp_scales);

/* A sink: the bitstreams of a batched compress call written straight into storage of the caller's (a Python `bytes` object created
 * at its final size and filled by the flush - what rans_interface.cpp:557-585 does with its py::bytes - instead of a buffer of the
 * library's that the binding copies once more: 2.5 MB per Kodak batch, 0.1 ms of the calling thread).  When item i's bitstream is
 * complete and its size known, the library calls alloc(user, i, nbytes) - ONCE per item, ON THE CALLING THREAD (it serves the host
 * workers' requests while it waits for them: a binding may take its interpreter's lock in alloc without its workers queueing for
 * it) - and the worker that coded the bitstream copies the nbytes there; items[i].bytes is that address on return, the caller's to
 * keep: fgmm_free / fgmm_ctx_take_buffers must NOT be given it.  alloc returning NULL fails the call with FGMM_ERR_NOMEM.  A failed
 * call returns items[i].bytes == NULL as ever; what the sink had handed out by then stays the caller's to release.  Checkpoints
 * (items[i].ckpt) are returned as without a sink.  sink == NULL: the plain call. */
typedef struct {
  void *(*alloc)(void *user, int item, size_t nbytes);
  void *user;
} fgmm_sink;
int fgmm_gmc_compress_batch_to(fgmm_ctx *ctx, void *stream, fgmm_item *items, int count, int mode, int clamp_scales, const fgmm_sink *sink);

/* ------------------------------------------------------------------------------------------------------------
 * 2b. The parameter head's last layer, on the matrix cores.
 *    Replaces: the final 1x1 convolution of `entropy_parameters`, nn.Conv2d(c_in, 3*K*M, 1) (compressai/models/ckbd_gmm.py:115-121;
 *    c_in = 640, M = 192 there), the chunk(3, 1) into scales | means | weights and the softmax over K
 *    (compressai/latent_codecs/gaussian_mixture_conditional.py:183-202).
 *    out[o][p] = bias[o] + sum_k weight[o][k] * x[k][p], o = t * K*M + k * M + c (t: scales, means, logits), computed on
 *    v_mfma_f32_32x32x2_f32 in ONE fixed order - bit for bit  acc = bias[o]; for k ascending: acc = fmaf(weight[o][k], x[k][p], acc)  -
 *    so that encoder and decoder derive identical parameters from identical weights on any ROCm / torch / MIOpen version (a
 *    BLAS's or MIOpen's summation order is not part of any contract; the reference silently relies on it being the same on both
 *    sides).  Streams coded from these parameters decode from these parameters.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct fgmm_head fgmm_head; /* a head's weights, packed for the kernel, on the context's device */
/* weight: device float32 [3*K*M, c_in] row-major (Conv2d.weight [3*K*M, c_in, 1, 1]); bias: device float32 [3*K*M] or NULL.
 * The weights are copied (packed) before the call returns. */
int fgmm_head_create(fgmm_ctx *ctx, void *stream, const float *weight, const float *bias, int M, int K, int c_in, fgmm_head **out);
/* ... with flags.  FGMM_HEAD_BF16X6: the same layer on the BF16 matrix cores with binary32 accuracy - weights and features split into
 * three bfloat16 parts each, six part products per product, accumulated in binary32 (fgmm_head16.hip): within ~2e-7 * sum |w x| of the
 * exact sum at a third of the matrix-pipe cycles.  Deterministic on gfx950 (the same inputs give the same parameters on every launch,
 * fused or not), but NOT the fmaf chain of the default form and not restatable bit for bit on a CPU: encoder and decoder must both use
 * it, on MI355X. */
#define FGMM_HEAD_BF16X6 1
int fgmm_head_create_ex(fgmm_ctx *ctx, void *stream, const float *weight, const float *bias, int M, int K, int c_in, int flags,
                        fgmm_head **out);
void fgmm_head_destroy(fgmm_head *head);
/* The parameters as tensors, for the decoder (and for anyone who wants them): item i reads x[i] = device float32 [c_in, hw[i]] and
 * writes out[i] = device float32 [3*K*M, hw[i]] - scales | means | LOGITS, each [K*M, hw] with channel k*M + c: three fgmm_params
 * planes with stride_k = M*hw, stride_c = hw, flags FGMM_PARAMS_LOGITS (the softmax over K and the sigma clamp run in the table
 * kernels).  Returns with the outputs complete. */
int fgmm_head_params_batch(fgmm_ctx *ctx, void *stream, const fgmm_head *head, const float *const *x, float *const *out,
                           const int64_t *hw, int count);
/* fgmm_gmc_compress_batch with the head FUSED into the encode-side CDF kernel: items[i].params is ignored, the twelve parameters of
 * a latent go from the MFMA accumulators into the table entry and never exist in HBM.  x[i]: device float32 [c_in, items[i].hw];
 * items[i].M must be the head's M.  The bitstreams are those of fgmm_gmc_compress_batch fed fgmm_head_params_batch's planes with
 * FGMM_PARAMS_LOGITS (same arithmetic, same bytes). */
int fgmm_gmc_compress_head_batch(fgmm_ctx *ctx, void *stream, fgmm_item *items, const float *const *x, int count,
                                 const fgmm_head *head, int mode, int clamp_scales);
/* ... with the bitstreams written into the caller's storage (fgmm_sink, section 2) */
int fgmm_gmc_compress_head_batch_to(fgmm_ctx *ctx, void *stream, fgmm_item *items, const float *const *x, int count,
                                    const fgmm_head *head, int mode, int clamp_scales, const fgmm_sink *sink);

/* ------------------------------------------------------------------------------------------------------------
 * 3. Building blocks ("same tables => same bytes" surfaces; also what the parity tests probe).
 *    SURVEY.md section 8(b) also lists fgmm_build_symtab_host / fgmm_build_cdftab_host: they do not exist on purpose —
 *    this library has no CPU implementation of the float work; the GPU builders below are the only ones.
 * ---------------------------------------------------------------------------------------------------------- */

/* GPU: float mixture-CDF pair per symbol, c1 = cdf(v - 0.5), c2 = cdf(v - 0.5 + 1.0)  (rans_interface.cpp:498-501).
 * v: device int32[n]; params (n,4) device via (stride_n, stride_k); c1/c2: device float32[n]. */
int fgmm_gmm_cdf_hip(fgmm_ctx *ctx, void *stream, const int32_t *v, const float *scales, const float *means,
                     const float *weights, int64_t n, int64_t stride_n, int64_t stride_k, int mode, float *c1,
                     float *c2);

/* GPU: the kernels' softmax over K on (n, 4) rows of logits -> weights (what FGMM_PARAMS_LOGITS computes in place of
 * torch.softmax, gaussian_mixture_conditional.py:198-202; within 2e-7 of it).  Both pointers device, row-major (n, 4). */
int fgmm_softmax4_hip(fgmm_ctx *ctx, void *stream, const float *logits, float *weights, int64_t n);

/* GPU: encode-side symbol table.  packed[i] = start | range << 16 with start = (uint16)(c1*65535),
 * range = (uint16)(end - start); range == 0 marks the reference's bypass escape (rans_interface.cpp:512-517) and
 * the low half then carries the low 16 bits of the symbol.  All pointers device. */
int fgmm_build_symtab_hip(fgmm_ctx *ctx, void *stream, const int32_t *symbols, const float *scales,
                          const float *means, const float *weights, int64_t n, int64_t stride_n, int64_t stride_k,
                          int mode, uint32_t *packed);

/* GPU: decode-side edge tables (format v5).  For latent i the reference's bisection can only ever look at
 *   F_i[v] = (uint16)(cdf_i(v - 0.5) * 65535),  v in [-max_bs, max_bs + 1]      (rans_interface.cpp:826-862).
 * The kernels find, exactly, the window outside which F_i is constant (evaluating F_i everywhere except where
 * every mixture component is provably saturated — fgmm_selftest_saturation) and store it:
 *   header of latent i, one of three forms (chosen per table from max_bs: FGMM_HDR_FORM(max_bs)):
 *     2 bytes  (a + max_bs) | cnt << 8, cnt in [1, 254]            when 2*max_bs + 2 <= 254
 *              cnt field 255: the row begins with a 4-byte header of the next form (non-monotone rows)
 *     4 bytes  int16 a | cnt << 16 (15 bits) | nonmono << 31        when max_bs <= FGMM_MAX_BS_H4
 *     8 bytes  int32 a ; cnt (31 bits) | nonmono << 31              any max_bs <= FGMM_MAX_BS
 *   row i  = F_i[a .. a+cnt), from the first non-zero edge to the start of the trailing constant run;
 *            F_i[v < a] = 0,   F_i[v >= a+cnt] = the row's last entry;  rows are 2-byte aligned:
 *     cnt < 14 or nonmono : uint16[cnt]                                                   (2*cnt bytes)
 *     cnt >= 14, monotone : Elias-Fano with l low bits, l = 12 for cnt <= 48, else 8, as ONE little-endian bit string
 *                           (bit b = bit (b & 7) of byte b >> 3) rounded up to 16 bits:
 *                             bits [0, HB), HB = cnt + (65536 >> l) : bit ((F_j >> l) + j) set for entry j
 *                             bits [LB + j*l, LB + (j+1)*l), LB = HB rounded up to 8 : the low l bits of entry j
 *     (FGMM_TAB_RAW_ROWS: every row in the first form; the batched decoder raises the threshold 14: "ef_min" option)
 *   `nonmono` is set when the row decreases somewhere.
 * fgmm_build_cdftab_hip (generic two-pass kernels, lane = latent, any max_bs <= FGMM_MAX_BS_H4 here): 4-byte headers, rows
 *   in LATENT ORDER with no stored offset (row i+1 starts where row i ends).  hdr: device uint32[n]; pool: device bytes;
 *   pool_used: device uint64[1] = bytes written.  pool_cap >= n * 2 * (2*max_bs + 2) always suffices; a smaller
 *   pool yields FGMM_ERR_NOMEM with *pool_used = the bytes needed.
 * fgmm_build_tab_hip (the single-pass kernel of the batched decode path: parameters staged in LDS, evaluation flattened
 *   over pairs of edges in packed fp32): headers in the form FGMM_HDR_FORM(max_bs) (device, n * form bytes); rows of each
 *   block of *tl_out consecutive latents are contiguous and start at rows + 4 * blk_off[block]  (blocks are placed by an
 *   atomic cursor: any order, each padded to 4 bytes); blk_off: device uint32[ceil(n / tl)], provide ceil(n / 16) entries; rows_used: device
 *   uint64[1].  FGMM_ERR_UNSUPPORTED when 2*max_bs + 2 does not fit the kernel's LDS budget (use the generic kernels),
 *   FGMM_ERR_NOMEM (with *rows_used = the bytes needed) when rows_cap is too small. */
#define FGMM_TAB_NO_PRUNE 1 /* flags: evaluate all of F_i instead of skipping its saturated tails (A/B testing) */
#define FGMM_TAB_CLAMP 2    /* flags: clamp sigma to [0.11, 256] first (the entropy-model path's kernel variant) */
#define FGMM_TAB_RAW_ROWS 4 /* flags: no Elias-Fano rows, every row as uint16 entries (35 % more bytes, 2-3x faster to search:
                               what the batched decoder chooses for calls with fewer bitstreams than host workers) */
#define FGMM_HDR_FORM(max_bs) ((2 * (int64_t)(max_bs) + 2 <= 254) ? 2 : ((max_bs) <= FGMM_MAX_BS_H4 ? 4 : 8))
int fgmm_build_cdftab_hip(fgmm_ctx *ctx, void *stream, const float *scales, const float *means,
                          const float *weights, int64_t n, int64_t stride_n, int64_t stride_k, int mode,
                          int32_t max_bs, int flags, uint32_t *hdr, uint8_t *pool, uint64_t pool_cap,
                          uint64_t *pool_used);
int fgmm_build_tab_hip(fgmm_ctx *ctx, void *stream, const float *scales, const float *means, const float *weights,
                       int64_t n, int64_t stride_n, int64_t stride_k, int mode, int32_t max_bs, int flags, void *hdr,
                       uint32_t *blk_off, uint8_t *rows, uint64_t rows_cap, uint64_t *rows_used, int32_t *tl_out);

/* GPU self-test: exhaustive scan (every binary32 beyond the thresholds) of the saturation lemmas that let the
 * table kernel skip the tails of F_i.  *n_bad_out = number of violating inputs (must be 0). */
int fgmm_selftest_saturation(fgmm_ctx *ctx, int mode, uint64_t *n_bad_out);

/* GPU self-test of the hand-expanded correctly-rounded cores (fgmm_math.h) against the compiler's IEEE '/' and
 * sqrtf: which = 0 sqrt (every binary32 in [0,2]), 1 division by a clamped sigma (n hashed pairs from `seed`),
 * 2 reciprocal of d >= 1 (every binary32 in [1,+inf]); which = 3, 4, 5: the kernels' slimmed Phi evaluation against
 * the literal one (mode which - 3) for every binary32 |z| < 2^48; which = 6: the Markstein-step sqrt of the Polya
 * path for every binary32 in {0} U [2^-24, 1].  *n_bad_out must come back 0. */
int fgmm_selftest_fastmath(fgmm_ctx *ctx, int which, uint64_t n, uint64_t seed, uint64_t *n_bad_out);

/* Host, integer only: symbol table (+ raw symbols, needed only where range == 0 and abs(symbol) >= 32768)
 * -> bitstream.  BufferedRansEncoder::flush semantics (rans_interface.cpp:557-585). */
int fgmm_rans_encode_symtab(const uint32_t *packed, const int32_t *symbols_or_null, int64_t n, uint8_t **out,
                            size_t *out_len);
/* Two independent tables -> two bitstreams, coded by the calling thread in turn symbol by symbol (the batched paths use
 * this when there are more bitstreams than workers: two dependency chains share a core).  out[k] is byte for byte what
 * fgmm_rans_encode_symtab returns for table k. */
int fgmm_rans_encode_symtab2(const uint32_t *packed0, const int32_t *symbols0_or_null, int64_t n0, const uint32_t *packed1,
                             const int32_t *symbols1_or_null, int64_t n1, uint8_t **out0, size_t *out0_len, uint8_t **out1,
                             size_t *out1_len);
/* The same for `ways` (1..4) tables given as arrays: out[k] / out_len[k] receive bitstream k.  On an error no buffer is
 * returned. */
int fgmm_rans_encode_symtab_n(int ways, const uint32_t *const *packed, const int32_t *const *symbols_or_null, const int64_t *n,
                              uint8_t **out, size_t *out_len);

/* ... noting checkpoints every `stride` symbols (a power of two; see fgmm_ckpt): ckpt_out[fgmm_ckpt_count(n, stride)].  The
 * bitstream is the one fgmm_rans_encode_symtab returns. */
int64_t fgmm_ckpt_count(int64_t n, int64_t stride); /* (n - 1) / stride, 0 for stride <= 0 */
int fgmm_rans_encode_symtab_ckpt(const uint32_t *packed, const int32_t *symbols_or_null, int64_t n, int64_t stride, uint8_t **out,
                                 size_t *out_len, fgmm_ckpt *ckpt_out);

/* The same from a table that lies in `n_seg` (1..4) SEGMENTS of `seg_len` entries (the last may be shorter): seg[s] holds the entries
 * [s * seg_len, (s + 1) * seg_len).  What the batched encoder does with the tables of a large call: they cross PCIe in segments, LAST
 * SEGMENT FIRST (rANS encodes backwards), and an encoder follows the landing.  The bitstream (and the checkpoints, stride > 0) are
 * byte for byte those of the one-piece forms. */
int fgmm_rans_encode_symtab_segs(const uint32_t *const *seg, int n_seg, int64_t seg_len, const int32_t *symbols_or_null, int64_t n,
                                 int64_t stride, uint8_t **out, size_t *out_len, fgmm_ckpt *ckpt_out);

/* Host, integer only: edge tables (4-byte headers, rows sequential in latent order, as fgmm_build_cdftab_hip lays them
 * out) -> symbols; the reference's bisection with every float evaluation replaced by a look-up in F_i.  pool_len bounds
 * every row access: a malformed table (cnt = 0, a row past the pool, an inconsistent Elias-Fano row) yields
 * FGMM_ERR_INVALID, never an out-of-bounds read; up to 32 bytes past a row may be read, so keep 32 bytes of slack
 * after the last row inside pool_len. */
int fgmm_rans_decode_cdftab(const uint8_t *encoded, size_t encoded_len, const uint32_t *hdr, const uint8_t *pool,
                            uint64_t pool_len, int64_t n, int32_t max_bs, int flags, int32_t *out_symbols);
/* The same for any header form and block-placed rows (fgmm_build_tab_hip); blk_off may be NULL (sequential rows).
 * flags: FGMM_TAB_RAW_ROWS as the table was built. */
int fgmm_rans_decode_tab(const uint8_t *encoded, size_t encoded_len, const void *hdr, int hdr_form, const uint32_t *blk_off,
                         int32_t tl, const uint8_t *rows, uint64_t rows_len, int64_t n, int32_t max_bs, int flags,
                         int32_t *out_symbols);
/* The same from a checkpointed stream (block-placed rows: blk_off != NULL), segment by segment on the calling thread - the
 * batched decoder runs the segments on its workers.  Every segment is verified against the next checkpoint; the first mismatch
 * (wrong or hostile notes) makes the whole stream a sequential decode: the symbols are fgmm_rans_decode_tab's in every case.
 * *verified_out (may be NULL): 1 when every checkpoint held. */
int fgmm_rans_decode_tab_ckpt(const uint8_t *encoded, size_t encoded_len, const void *hdr, int hdr_form, const uint32_t *blk_off,
                              int32_t tl, const uint8_t *rows, uint64_t rows_len, int64_t n, int32_t max_bs, int flags,
                              const fgmm_ckpt *ckpt, int64_t n_ckpt, int64_t stride, int32_t *out_symbols, int32_t *verified_out);

/* ------------------------------------------------------------------------------------------------------------
 * 4. Table path — the `z` hyper-latent coder: CompressAI's original table rANS, the other
 *    half of `compressai.ans`.  Host only, integer only.  `cdfs` is a row-major int32 [n_cdfs, cdf_stride] matrix
 *    (the reference takes a list of lists, rans_interface.cpp:334-338); every index is validated here (the
 *    reference's asserts are compiled out).
 * ---------------------------------------------------------------------------------------------------------- */

/* RansEncoder.encode_with_indexes(symbols, indexes, cdfs, cdfs_sizes, offsets) -> bytes  (rans_interface.cpp:587-598) */
int fgmm_encode_with_indexes(const int32_t *symbols, const int32_t *indexes, int64_t n, const int32_t *cdfs,
                             int64_t cdf_stride, int32_t n_cdfs, const int32_t *cdfs_sizes, const int32_t *offsets,
                             uint8_t **out, size_t *out_len);
/* RansDecoder.decode_with_indexes(encoded, indexes, cdfs, cdfs_sizes, offsets) -> int32[n]  (rans_interface.cpp:619-688) */
int fgmm_decode_with_indexes(const uint8_t *encoded, size_t encoded_len, const int32_t *indexes, int64_t n,
                             const int32_t *cdfs, int64_t cdf_stride, int32_t n_cdfs, const int32_t *cdfs_sizes,
                             const int32_t *offsets, int32_t *out_symbols);

/* BufferedRansEncoder (rans_interface.hpp:57-86): symbols accumulate over calls — table calls and GMM symbol tables
 * may be mixed — and ONE stream is flushed (rans_interface.cpp:557-585). */
typedef struct fgmm_symbuf fgmm_symbuf;
int fgmm_symbuf_create(fgmm_symbuf **out);
void fgmm_symbuf_destroy(fgmm_symbuf *b);
int64_t fgmm_symbuf_size(const fgmm_symbuf *b); /* entries buffered (symbols + escape nibbles) */
int fgmm_symbuf_append_table(fgmm_symbuf *b, const int32_t *symbols, const int32_t *indexes, int64_t n,
                             const int32_t *cdfs, int64_t cdf_stride, int32_t n_cdfs, const int32_t *cdfs_sizes,
                             const int32_t *offsets);
int fgmm_symbuf_append_symtab(fgmm_symbuf *b, const uint32_t *packed, const int32_t *symbols_or_null, int64_t n);
int fgmm_symbuf_flush(fgmm_symbuf *b, uint8_t **out, size_t *out_len); /* empties the buffer */
/* GPU-built GMM symbols appended to a buffer: BufferedRansEncoder.encode_with_indexes_gmm (rans_interface.cpp:458-554) */
int fgmm_symbuf_append_gmm(fgmm_ctx *ctx, fgmm_symbuf *b, const int32_t *symbols, const float *scales,
                           const float *means, const float *weights, int64_t n, int64_t stride_n, int64_t stride_k,
                           int K, int mode, int memspace);

/* RansDecoder.set_stream / decode_stream (rans_interface.cpp:886-956): a decoder that keeps its state between calls */
typedef struct fgmm_decstream fgmm_decstream;
int fgmm_decstream_create(const uint8_t *encoded, size_t encoded_len, fgmm_decstream **out); /* copies the stream */
void fgmm_decstream_destroy(fgmm_decstream *d);
int fgmm_decstream_decode(fgmm_decstream *d, const int32_t *indexes, int64_t n, const int32_t *cdfs, int64_t cdf_stride,
                          int32_t n_cdfs, const int32_t *cdfs_sizes, const int32_t *offsets, int32_t *out_symbols);

/* ---- section 5: the callers either side of the path ---------------------------------
 * CheckerboardLatentCodec.unembed / embed (compressai/latent_codecs/checkerboard.py:333-377): split a [planes, h, w]
 * device tensor into its two checkerboard halves [2, planes, h, w/2] (half 0 = anchors) and back.  planes = n * c;
 * w must be even; elem_bytes 4 (float32 / int32) or 2 (float16); anchor_odd = 0 for anchor_parity "even" (anchors at
 * (even row, even column) and (odd row, odd column)), 1 for "odd".  Pure data movement: any bit pattern is preserved. */
int fgmm_ckbd_unembed(fgmm_ctx *ctx, void *stream, const void *src, void *dst, int64_t planes, int64_t h, int64_t w,
                      int elem_bytes, int anchor_odd);
int fgmm_ckbd_embed(fgmm_ctx *ctx, void *stream, const void *src, void *dst, int64_t planes, int64_t h, int64_t w,
                    int elem_bytes, int anchor_odd);

/* compressai._CXX.pmf_to_quantized_cdf (compressai/cpp_exts/ops/ops.cpp:40-109): cdf_out has n + 1 entries */
int fgmm_pmf_to_quantized_cdf(const float *pmf, int n, int precision, uint32_t *cdf_out);

#ifdef __cplusplus
}
#endif
#endif /* FLASHGMM_AMD_H */
