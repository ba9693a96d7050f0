/*
 * vits.h — C ABI of the MI355X-native VITS inference library (libvits_hip.so).
 *
 * DROP-IN BOUNDARY. The first five entry points are byte-compatible with the reference's exported C API
 * (/root/reference/src/include/vits.h:87-102, implemented at /root/reference/src/vits.cpp:1193-1232):
 * same names, same argument meaning, same ownership rules. `vits_model` is opaque here (the reference exposes
 * a C++ class with ggml-typed members, vits.h:17-85, which cannot survive without ggml; every reference
 * caller — test/main.cpp:68-78, test/bench_e2e.cpp:68-91, test/bench.cpp:204-206 — only passes the pointer).
 *
 * Differences from the reference, all deliberate:
 *   - no exception / exit(1) crosses the boundary (reference: std::runtime_error from
 *     src/vits_model_data.cpp:102,144 and ASSERT->exit at src/include/debug.h:29-36). Failures return
 *     NULL / {NULL,0}; the message is available from vits_last_error().
 *   - nothing is printed to stdout (reference prints at src/vits.cpp:27,1200 and in the loaders).
 *   - everything below "extensions" is new: id-level and batched entry points (the reference is batch-1,
 *     text-only), pipelined batches on one handle, synthetic model generation, taps, profiling, operator-level
 *     entry points for parity tests.
 *   - threading: one call at a time per model handle, as in the reference (vits_model::process writes member
 *     tensors, vits.h:22-30) — but ENFORCED: an entry point entered while another call on the same handle is in
 *     progress (from a second thread, or from an on_chunk callback) returns its failure value with
 *     vits_last_error() = "model busy: ...". Distinct handles may be used from distinct threads concurrently.
 *
 * Plain C types only: pointers, sizes, PODs. No torch / HIP types appear in any signature; device buffers are
 * passed as `void*` device addresses.
 */
#ifndef VITS_HIP_H
#define VITS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
#define VITS_API extern "C" __attribute__((visibility("default")))
#else
#define VITS_API __attribute__((visibility("default")))
#endif

typedef struct vits_model vits_model; /* opaque */

/* reference: src/include/vits.h:89-92 */
typedef struct vits_result {
    float* data; /* mono fp32 PCM in [-1,1]; owned by the library until vits_free_result */
    size_t size; /* samples */
} vits_result;

/* ---- the reference's five symbols ------------------------------------------------------------------- */

/* reference: vits.h:94, vits.cpp:1205-1215. Copies what it needs; caller may free `bytes` on return. */
VITS_API vits_model* vits_model_load_from_bytes(const char* bytes, size_t size);
/* reference: vits.h:96, vits.cpp:1193-1203 */
VITS_API vits_model* vits_model_load_from_file(const char* path);
/* reference: vits.h:98, vits.cpp:1217-1219. A handle that another thread (or a streaming callback) is inside is NOT freed: the call returns with
 * vits_last_error() = "vits_free_model: model busy ..." and leaves the handle exactly as it was (the busy flag belongs to the call in progress and is
 * released when that call returns) — call vits_free_model again then. The function returns void as in the reference: check vits_last_error(). */
VITS_API void vits_free_model(vits_model* model);
/* reference: vits.h:100, vits.cpp:1221-1223 */
VITS_API void vits_free_result(vits_result result);
/* reference: vits.h:102, vits.cpp:1225-1232 -> vits_model::process vits.cpp:1101-1191.
 * Text is lower-cased, greedily matched against the model vocabulary and interspersed with blanks
 * (src/vits_tokenizer.cpp:182-208). Noise comes from the reference's process-global libstdc++ stream
 * (VITS_NOISE_REFERENCE), mode is the model default (VITS_MODE_REFERENCE unless changed).
 * Two model-file flags decide what the TEXT entry points (this one, vits_model_tokenize, vits_model_file_tokenize) do:
 *   config "phonetic" = "1" (vits_model_data.cpp:92-94): the model expects espeak-ng phonemes. The reference asserts out at load unless built with
 *     VITS_ESPEAK (vits_tokenizer.cpp:176-178); espeak is out of scope here, so the model LOADS and the text entry points fail with
 *     vits_last_error() = "model expects espeak phonemes ... pass ids"; the id entry points are unaffected.
 *   add_blank = 0: the reference's tokenizer returns an EMPTY id list (Q11, vits_tokenizer.cpp:200-208) and its process() an empty result; the same
 *     here ({NULL, 0} with a message that says so; vits_model_tokenize returns 0). */
VITS_API vits_result vits_model_process(vits_model* model, const char* phonemes);

/* ---- extensions ------------------------------------------------------------------------------------- */

/* Thread-local message of the last failure in this thread ("" if none). */
VITS_API const char* vits_last_error(void);

/* Semantics mode:
 *   VITS_MODE_REFERENCE: what /root/reference/src/vits.cpp literally computes: ConvTranspose1d without
 *     crop (Q1, vits.cpp:187), final LeakyReLU slope 0.1 (Q2, :638), spline width affine (Q3, :720),
 *     index -1 wrap on the last token (Q4, ggml-util.h:235), exp(+log_scale) (Q5, :913-918), and the masked get / set
 *     pair of the spline tails as implemented (Q6, :832-849 with custom-ops.h:739-752,829-862: masked_get keeps the
 *     shape, masked_set consumes compacted values — a latent outside [-5, 5] shifts the log-durations of every later
 *     token; the identity permutation while every latent is inside).
 *   VITS_MODE_HF: what transformers.VitsModel (the model the reference ports, vits.cpp:113) computes. */
#define VITS_MODE_DEFAULT (-1)
#define VITS_MODE_REFERENCE 0
#define VITS_MODE_HF 1
VITS_API int vits_model_set_mode(vits_model* model, int mode);
VITS_API int vits_model_get_mode(const vits_model* model);

/* Conv arithmetic. The reference's conv is fp16 x fp16 -> fp32: weights are cast to fp16 by the exporter
 * (scripts/export_vits.py:87) and the activations are rounded to fp16 by the im2col in front of every conv
 * (src/include/custom-ops.h:684-690).
 *   VITS_ARITH_F32 (default): activations stay fp32, weights are the stored 16-bit values widened exactly, fp32 MFMA
 *     (v_mfma_f32_32x32x2_f32) — the exact-arithmetic parity mode the headline is measured in.
 *   VITS_ARITH_F16: the literal Q7 arithmetic: conv inputs rounded to fp16 (round-to-nearest-even) where the tile is staged,
 *     fp16 weights, fp32 accumulation on v_mfma_f32_32x32x16_f16.
 *   VITS_ARITH_BF16: the same with bf16 operands (BASELINE.json configs[4]: bf16 weights) on v_mfma_f32_32x32x16_bf16.
 *   VITS_ARITH_F32_SPLIT (round 6, opt-in): fp32-ACCURATE results on the bf16 matrix cores. The ResBlock convolutions of the vocoder's wide stages
 *     (C >= 128: 64 % of the path's FLOPs) take their fp16-valued weights as the exact sum of two bf16 values and their fp32 activations as the exact sum of
 *     three, and accumulate the five significant cross products in fp32 (csrc/conv_split.hip): the same accuracy against the exact sum as the fp32 fmaf
 *     chain (measured, tools/split_micro.hip), at 1.8x its matrix-core rate — but NOT the same bits (another summation order), so "batch 1 == row of a
 *     batch, bit for bit" is asserted for VITS_ARITH_F32 only. Everything else, stage one included (durations bit-exact), is the VITS_ARITH_F32 path.
 *     The scope setting does not apply. Weights that are not exactly two bf16 pieces (an fp32-stored conv) keep the fp32 kernels.
 * 16-bit modes apply to the Conv1d / ConvTranspose1d of the scope chosen with vits_model_set_arith_scope (never to the Linear
 * layers q/k/v/out, which are ggml_mul_mat on f32 x f32 in the reference, vits.cpp:287-289,358); everything else (layer norms,
 * attention softmax, splines, gates, residual adds, accumulators) stays fp32. Set between calls, not during one. */
#define VITS_ARITH_F32 0
#define VITS_ARITH_BF16 1
#define VITS_ARITH_F16 2
#define VITS_ARITH_F32_SPLIT 3
VITS_API int vits_model_set_arith(vits_model* model, int arith);
VITS_API int vits_model_get_arith(const vits_model* model);
/* Which convolutions a 16-bit arithmetic mode applies to.
 *   VITS_ARITH_SCOPE_FLOW_VOCODER (default): the coupling flow and HiFiGAN (99 % of the FLOPs) run on 16-bit operands; stage one
 *     (text encoder, duration predictor, prior projection: vits.cpp:244-440,927-972) stays EXACT fp32, so the durations — the
 *     path's only integer output, ceil(exp(logw) * length_scale) at vits.cpp:996-1001 — the frame counts and the sample counts
 *     are bit-identical to the fp32 path's (and to the oracle's) in every arithmetic mode.
 *   VITS_ARITH_SCOPE_ALL_CONVS: the literal Q7 arithmetic — every Conv1d / ConvTranspose1d of the path, stage one included
 *     (custom-ops.h:684-690 rounds the im2col of EVERY conv). Durations then carry the mode's rounding noise: a log-duration
 *     within an fp16 ulp of a ceil() boundary may land on either side.
 * Set between calls, not during one. */
#define VITS_ARITH_SCOPE_FLOW_VOCODER 0
#define VITS_ARITH_SCOPE_ALL_CONVS 1
VITS_API int vits_model_set_arith_scope(vits_model* model, int scope);
VITS_API int vits_model_get_arith_scope(const vits_model* model);

/* EMULATED ggml lookup tables — INFERRED from upstream ggerganov/ggml of the reference's era: the maxilevi/ggml fork the
 * reference is built against is absent (empty submodule), so this is a labelled emulation of what `ggml_gelu` (vits.cpp:673,687) and
 * `ggml_soft_max` (vits.cpp:329,719,735) most probably compute there, not a pinned restatement:
 *   ggml_gelu:     y = fp16->fp32( table_gelu_f16[ fp32->fp16(x) ] ), the table holding fp16( 0.5 x (1 + tanh(sqrt(2/pi) x (1 + 0.044715 x^2))) );
 *   ggml_soft_max: e_i = fp16->fp32( table_exp_f16[ fp32->fp16(x_i - max) ] ), the sum in double, p_i = e_i * (float)(1 / sum).
 * The GELU of the duration predictor's DDS layers and the soft-max of the text encoder's attention and of the spline bins go through those tables
 * (built on the host with the C library, as ggml_init does). Default 0: erf-GELU (what transformers.VitsModel computes) and fp32 soft-max — the
 * mode every parity fixture is in. Set between calls.
 *   on = 1: a table turns a last-bit difference of its argument into a 5e-4 step of its value, so in this mode STAGE ONE (text encoder + duration
 *           predictor, ~1 % of the work) runs in ONE fixed order of operations — include/vits_exact_math.h: one device thread per output element,
 *           sequential sums, fp contraction off, polynomial exp / log instead of the device library's — which the oracle shares
 *           (vo_opts.ggml_tables = 1): log-durations, durations, frame and sample counts are BIT-IDENTICAL to the oracle's (GPU test). A
 *           measurement instrument (how far do ggml's tables move the durations: ~0.1 % of the ids), not a serving configuration: stage one takes
 *           several times longer than the throughput kernels; everything behind the durations is the default path. Needs fp32 stage-one
 *           arithmetic (VITS_ARITH_SCOPE_FLOW_VOCODER, the default).
 *   on = 2: the same tables inside the THROUGHPUT kernels (MFMA summation order, the device library's exp / log): agrees with the oracle's loops
 *           (vo_opts.ggml_tables = 2) statistically only — a few durations per ten thousand ids differ. Kept for that comparison. */
VITS_API int vits_model_set_ggml_tables(vits_model* model, int on);
VITS_API int vits_model_get_ggml_tables(const vits_model* model);

/* Noise source for the two N(0,1) draws (vits.cpp:948 [T,2] and :1059 [L,192]). */
#define VITS_NOISE_REFERENCE 0 /* libstdc++ minstd_rand0 + normal_distribution<float>, global, host-serial */
#define VITS_NOISE_COUNTER 1   /* include/vits_synth_noise.h, evaluated on the device                       */
#define VITS_NOISE_EXPLICIT 2  /* caller-provided host buffers                                             */
/* Re-seed the reference noise stream (the reference never does: default seed 1, vits.cpp:31). */
VITS_API void vits_reference_noise_seed(uint32_t seed);

/* Streaming sink: `pcm` points at samples [offset, offset+n) of utterance `utt` (host memory owned by the library,
 * valid during the call only). Chunks of one utterance arrive in order and tile it exactly. */
typedef int (*vits_chunk_callback)(void* user, int32_t utt, size_t offset, const float* pcm, size_t n);

typedef struct vits_process_opts {
    uint32_t struct_size;       /* = sizeof(vits_process_opts) */
    int32_t mode;                /* VITS_MODE_*; VITS_MODE_DEFAULT = model default */
    int32_t noise_kind;          /* VITS_NOISE_* */
    uint64_t noise_seed;         /* VITS_NOISE_COUNTER: utterance u uses seed noise_seed + u */
    const float* noise_dur;      /* VITS_NOISE_EXPLICIT: host [B][2][id_stride] */
    const float* noise_prior;    /* VITS_NOISE_EXPLICIT: host [B][192][noise_prior_stride] */
    int64_t noise_prior_stride;  /* frames between channels in noise_prior */
    int32_t fixed_duration;      /* >0: every id lasts this many frames (pinned-length benchmark run) */
    int32_t collect_taps;        /* 1: keep stage outputs for vits_model_get_tap */
    void* out_device;            /* optional device buffer [B][out_device_stride] fp32 to receive the PCM */
    int64_t out_device_stride;   /* samples; must be >= the longest utterance */
    int32_t skip_host_copy;      /* 1: leave PCM on the device only (requires out_device or tap access) */
    int32_t async;               /* 1: return without synchronising the stream (requires skip_host_copy and
                                    fixed_duration>0, the only case with no data-dependent host read);
                                    call vits_model_sync() before touching out_device */
    /* ---- long-form / streaming (the reference cannot run 1024-id inputs at all: its 512 MB
     * arena at vits.cpp:1145 overflows) ---- */
    int32_t vocoder_chunk_frames; /* >0: run HiFiGAN over windows of this many frames (plus the exact receptive-
                                     field halo on both sides) instead of the whole utterance: activation memory is
                                     bounded by the window, and the PCM is BIT-IDENTICAL to the unchunked result.
                                     Ignored with collect_taps. 0 = whole utterance. */
    int32_t frames_only;          /* 1: run the text encoder and the duration predictor only: the result carries frames[] and
                                     lengths[] (samples) per utterance and no audio. For dispatchers: sizing out_device,
                                     balancing utterance shards by predicted frames. Same noise -> same durations
                                     as the full call. */
    vits_chunk_callback on_chunk; /* optional: called on the calling thread as each window's PCM reaches the host,
                                     while the device already works on the next windows (needs vocoder_chunk_frames>0
                                     and a host copy, i.e. skip_host_copy=0). Non-zero return aborts the call. */
    void* on_chunk_user;
    const int32_t* noise_seed_offsets; /* VITS_NOISE_COUNTER, optional host [B]: utterance b draws from the stream with seed
                                          noise_seed + noise_seed_offsets[b] instead of noise_seed + b. Lets a dispatcher
                                          re-order or re-shard utterances (e.g. balance ranks by frames) without changing
                                          any utterance's audio. */
} vits_process_opts;

typedef struct vits_batch_result {
    float* data;      /* host [batch][stride] PCM (NULL when skip_host_copy) */
    size_t stride;    /* samples between utterances */
    int64_t* lengths; /* [batch] samples per utterance */
    int64_t* frames;  /* [batch] spectrogram frames per utterance (L) */
    size_t batch;
} vits_batch_result;

/* One utterance from ids (already blank-interspersed, i.e. what input_ids_tensor holds at vits.cpp:1109-1111);
 * same noise/mode behaviour as vits_model_process. */
VITS_API vits_result vits_model_process_ids(vits_model* model, const int32_t* ids, size_t n_ids);

/* B independent utterances; ids is host [B][id_stride], id_lengths[b] <= id_stride valid ids in row b.
 * Utterances never interact: results equal B separate batch-1 calls. Returns 0 on success. */
VITS_API int vits_model_process_batch(vits_model* model, const int32_t* ids, const int32_t* id_lengths, int32_t batch,
                                      int32_t id_stride, const vits_process_opts* opts, vits_batch_result* out);
VITS_API void vits_free_batch_result(vits_batch_result* r);
/* Block until everything queued by this model has finished. */
VITS_API int vits_model_sync(vits_model* model);

/* ---- pipelined batches on ONE handle ------------------------------------------------------------------------------------
 * vits_model_process_batch is one call in, one result out: its stage one (text encoder + duration predictor, vits.cpp:244-440,
 * 927-972: ~110 small, latency-bound launches in exact fp32) runs with the matrix cores mostly idle, and the host read of the
 * frame counts (vits.cpp:1133) drains the device once per call. The pair below keeps up to TWO batches in flight on one model
 * handle: vits_model_submit_batch(i + 1) queues stage one of batch i + 1 on the handle's front-end stream (its own stage-one
 * arena), where it runs UNDER the flow / vocoder kernels of batch i, reads its frame counts while the device is busy, queues its
 * flow + vocoder behind batch i's on the main stream and returns; vits_model_wait() returns the results of the OLDEST submitted
 * batch (in submission order). Every kernel is batch-invariant and runs on the same operands as in vits_model_process_batch, so
 * the PCM, lengths and frames are BIT-IDENTICAL to it (GPU test). Typical loop:
 *     submit(0); for (i = 1; i < n; ++i) { submit(i); wait(&r[i-1]); } wait(&r[n-1]);
 * Restrictions (checked; -1 + vits_last_error): VITS_NOISE_COUNTER only (no host noise buffers outlive the call), no collect_taps,
 * on_chunk, frames_only or async; at most two batches in flight; opts.out_device buffers of in-flight batches must be distinct.
 * While batches are in flight the other entry points that use the device (process*, set_arith, taps) refuse with "batches in flight".
 * With the per-kernel profiler enabled the two stages run serialised on the main stream (events need non-overlapping kernels).
 * Both return 0 on success. vits_model_pending = number of submitted batches not yet waited for (0..2). */
VITS_API int vits_model_submit_batch(vits_model* model, const int32_t* ids, const int32_t* id_lengths, int32_t batch, int32_t id_stride,
                                     const vits_process_opts* opts);
VITS_API int vits_model_wait(vits_model* model, vits_batch_result* out);
VITS_API int vits_model_pending(const vits_model* model);

/* Tokenizer only (src/vits_tokenizer.cpp:182-208). Writes up to cap ids, returns the count (0 for a model file with add_blank = 0, as the
 * reference; -1 with vits_last_error() for a phonetic model or a NULL argument). */
VITS_API int64_t vits_model_tokenize(vits_model* model, const char* text, int32_t* ids, size_t cap);

/* Model facts. */
VITS_API int32_t vits_model_sampling_rate(const vits_model* model);
VITS_API int32_t vits_model_vocab_size(const vits_model* model);
VITS_API int64_t vits_model_weight_bytes(const vits_model* model);

/* Stage outputs of the LAST call with collect_taps=1, for utterance `utt`, copied to host as a dense
 * [channels][len] fp32 array. Names: "enc_out" [192][T], "prior_mean" [192][T], "prior_logvar" [192][T],
 * "log_duration" [1][T], "durations" [1][T] (integer-valued), "z_p" [192][L], "z_flow" [192][L],
 * "pre_tanh" [1][S], "waveform" [1][S], plus "noise_dur" [2][T] and "noise_prior" [192][L].
 * Returns the element count (0 = unknown tap), copies min(count, cap) floats. */
VITS_API int64_t vits_model_get_tap(vits_model* model, const char* name, int32_t utt, float* dst, size_t cap);

/* ---- synthetic model files (no real checkpoint is available) --------------------------
 * Builds a complete model file in the reference's on-disk format (writer
 * scripts/export_vits.py:5-70) with the default MMS-TTS architecture (or the tiny test architecture) and
 * deterministic weights derived from `seed`. Conv weights are stored as fp16 like export_vits.py:87. */
#define VITS_SYNTH_FULL 0 /* VitsConfig defaults == facebook/mms-tts-* architecture */
#define VITS_SYNTH_TINY 1 /* hidden 16 / small vocoder: small enough to commit as a fixture */
#define VITS_SYNTH_BF16 0x100 /* OR-ed in: store conv weights as bf16 (tensor type tag 2, an extension of the format) */
VITS_API int vits_synth_model_bytes(uint64_t seed, int32_t arch, char** bytes, size_t* size);
VITS_API void vits_free_bytes(char* bytes);
/* Parse a model file and write it back (host only): byte-exact round trip of the reference's format
 * (reader src/vits_model_data.cpp:29-97 + src/vits_tokenizer.cpp:22-55, writer scripts/export_vits.py:5-70). */
VITS_API int vits_model_file_reserialize(const char* in, size_t in_size, char** out, size_t* out_size);
/* Everything vits_model_load_from_bytes checks before it uploads (container format, hyper-parameters, the shape of every
 * tensor against them), on the host only. 0 = loadable; -1 = rejected (vits_last_error says which tensor and why). */
VITS_API int vits_model_file_validate(const char* bytes, size_t size);
/* Tokenize with the vocabulary stored in a model file, without loading the model onto a device. */
VITS_API int64_t vits_model_file_tokenize(const char* model_bytes, size_t size, const char* text, int32_t* ids, size_t cap);

/* ---- profiling (HIP events on the library's own stream) ------------------------------------------------
 * While enabled every kernel launch is bracketed by hipEventRecord on the launch stream. The report is a
 * JSON object {"kernels":[{"name":..,"calls":..,"ms":..,"flop":..,"bytes":..},...]} written into buf. */
VITS_API int vits_prof_enable(vits_model* model, int32_t on);
VITS_API int vits_prof_reset(vits_model* model);
VITS_API int64_t vits_prof_report(vits_model* model, char* buf, size_t cap);

/* ---- operator-level entry points (host buffers in, host buffers out; for parity tests) ---------------
 * Each mirrors one reference operator so a test can compare the HIP kernel with the oracle's restatement
 * of the same reference lines. All tensors are dense fp32, layout [batch][channels][time], time fastest
 * (== the reference's ggml ne order [time, channels, batch]). lens may be NULL (all = T). */

/* Arithmetic of the operator-level conv entry points below on this thread (VITS_ARITH_*; default fp32). */
VITS_API int vits_op_set_arith(int32_t arith);

/* conv1d_with_bias (vits.cpp:171-176 -> custom-ops.h:680-694) with the fusions the engine uses.
 * y = post( conv(pre(x)) + bias ), pre: 0 none, 1 leaky_relu(slope); post: 0 none, 1 relu,
 * 2 gated tanh*sigmoid over channel halves (vits.cpp:442-450; Cout must be even, output has Cout/2 channels);
 * then y = (y + residual) if residual, y = (y + accum) * out_scale if accum.
 * w is [Cout][Cin][K] (torch layout). pad_left/pad_right zero padding; output length T (same-length conv
 * requires pad_left + pad_right == (K-1)*dilation). */
typedef struct vits_conv1d_desc {
    int32_t batch, cin, cout, t, t_stride;
    int32_t k, dilation, pad_left;
    int32_t pre_act; /* 0 none, 1 leaky_relu */
    float pre_slope;
    int32_t post_act; /* 0 none, 1 relu, 2 gate */
    float out_scale;  /* applied when accum != NULL */
} vits_conv1d_desc;
VITS_API int vits_op_conv1d(const vits_conv1d_desc* d, const float* x, const float* w, const float* bias,
                            const float* residual, const float* accum, const int32_t* lens, float* y);

/* conv_transpose_1d_with_bias (vits.cpp:178-193). w is [Cin][Cout][K] (torch layout), K == 2*stride.
 * crop = (K-stride)/2 in HF mode (HF modeling_vits.py:483-490), 0 in reference mode (vits.cpp:187, Q1).
 * Output length = stride*T + K - stride - 2*crop. pre-activation leaky_relu(slope) is fused (vits.cpp:613). */
typedef struct vits_convt1d_desc {
    int32_t batch, cin, cout, t, t_stride, t_out_stride;
    int32_t k, stride, crop;
    float pre_slope; /* leaky_relu slope applied to x first; 1.0 = none */
} vits_convt1d_desc;
VITS_API int vits_op_conv_transpose1d(const vits_convt1d_desc* d, const float* x, const float* w, const float* bias,
                                      const int32_t* lens, float* y);

/* Relative-position multi-head self-attention core (vits.cpp:296-356):
 * q,k,v [B][H*hd][T] (q already scaled), rel_k/rel_v [2w+1][hd] shared by heads; out [B][H*hd][T]. */
VITS_API int vits_op_rel_attention(int32_t batch, int32_t heads, int32_t head_dim, int32_t t, int32_t t_stride,
                                   int32_t window, const float* q, const float* k, const float* v,
                                   const float* rel_k, const float* rel_v, const int32_t* lens, float* out);

/* layer_norm over channels of (x + residual) (vits.cpp:115-120,365-372). */
VITS_API int vits_op_add_layer_norm(int32_t batch, int32_t channels, int32_t t, int32_t t_stride, float eps,
                                    const float* x, const float* residual, const float* gamma, const float* beta,
                                    float* y);

/* ---- PCM16 / WAV sink (reference driver test/main.cpp:23-63: clamp to [-1,1], * 32767, truncate; 16 kHz mono) ------ */
VITS_API void vits_pcm16_from_float(const float* pcm, size_t n, int16_t* out);
/* The same conversion on the device, row by row: src [rows][src_stride] fp32 -> dst [rows][dst_stride] int16, the first
 * `cols` samples of each row (or lengths[r] of them when `lengths`, a DEVICE int64 array, is given). All pointers are
 * device pointers; runs on `hip_stream` (a hipStream_t, NULL = the default stream) without synchronising. Halves the
 * bytes of the multi-GPU PCM gather and of the host copy. Returns 0 on success. */
VITS_API int vits_pcm16_from_float_device(const float* src, int64_t src_stride, int16_t* dst, int64_t dst_stride,
                                          const int64_t* lengths, int32_t rows, int64_t cols, void* hip_stream);
/* Writes a canonical 44-byte-header RIFF/WAVE file exactly like test/main.cpp:36-60. Returns 0 on success. */
VITS_API int vits_write_wav16(const char* path, const float* pcm, size_t n, int32_t sample_rate);

/* ---- multi-GPU: the path's one exchange for C / C++ / Swift hosts -------------------------------------------------------
 * The reference synthesises one utterance per call (vits.cpp:184,303; callers test/main.cpp:68-78), so a batch shards by utterance:
 * ONE PROCESS PER GPU, each with its own model handle (vits_set_device(local_rank), weights replicated), rank r synthesising its own
 * rows — no data-path collective until the end, where every rank wants the whole batch's PCM. These three calls are that end: a
 * ragged all-gather over RCCL (xGMI inside a node) — the per-utterance lengths first (fixed size), then the rows padded to the longest
 * utterance of any rank. It is what vits.cpp_amd/multi_gpu.py does through torch.distributed, without Python. RCCL is loaded with
 * dlopen on first use (VITS_RCCL_LIB overrides the library name): single-GPU users need nothing, and world == 1 never touches it.
 *   rank 0:     char id[VITS_GATHER_ID_BYTES]; vits_pcm_gather_unique_id(id);  -> hand the 128 bytes to the other ranks (file, socket, MPI ...)
 *   every rank: g = vits_pcm_gather_init(id, sizeof id, rank, world, rows, row_capacity, 4);           (collective: all ranks call it)
 *   per batch:  vits_model_process_batch(..., opts.out_device = pcm, opts.skip_host_copy = 1, &r);
 *               vits_pcm_gather(g, pcm, stride, r.lengths, NULL, &all);                                 (collective)
 *               -> all.data: DEVICE [world * rows][all.stride], rank blocks in rank order; all.lengths: host [world * rows]
 *   end:        vits_pcm_gather_destroy(g);
 * rows is the same on every rank (pad a short shard with zero-length rows); row_capacity >= the longest utterance anywhere, in elements.
 * elem_bytes 4 = fp32 PCM, 2 = PCM16 (vits_pcm16_from_float_device first: half the bytes on the links). `hip_stream` = the stream the PCM was
 * produced on (the exchange is ordered behind it without a host wait), NULL when the producer has been synchronised — vits_model_process_batch
 * without opts.async has. The call returns when the gathered block is complete; it stays valid until the next call on the same object.
 * One call at a time per object. Return 0 / non-NULL on success, else -1 / NULL with vits_last_error().
 * Failure is COLLECTIVE as well: the first all-gather carries every rank's row_capacity and lengths, a rank whose own arguments are unusable
 * (NULL buffer, a negative length, a row longer than its pcm_stride or than row_capacity) sends -1 for the row instead of returning early, and
 * every rank derives the same verdict from the same table — all ranks return -1 with the offending (rank, row) in vits_last_error(), none is left
 * blocked in RCCL, and the object stays usable. pcm_stride need only reach the rank's OWN longest row. A HIP / RCCL error inside an exchange aborts
 * the communicator (ncclCommAbort) and the object refuses further calls: destroy it on every rank. Elements of a gathered row between its length
 * and `stride` are UNSPECIFIED (padding; rows are not zero-filled).
 * vits_pcm_gather_verdict is that decision as a pure host function of the gathered table (per rank: row_capacity, then `rows` lengths with -1 for an
 * unusable row): 0 and the common row width in *stride_out, or -1 with the message in vits_last_error(). */
#define VITS_GATHER_ID_BYTES 128
typedef struct vits_gather_ctx vits_gather_ctx;
typedef struct vits_gather_result {
    const void* data;       /* device [rows_total][stride] elements, owned by the gather object */
    int64_t stride;         /* elements between rows = the longest utterance of any rank */
    const int64_t* lengths; /* host [rows_total]: elements per utterance, rank blocks in rank order */
    int32_t rows_total;     /* world * rows */
} vits_gather_result;
VITS_API int vits_pcm_gather_unique_id(char* id_out /* [VITS_GATHER_ID_BYTES] */);
VITS_API vits_gather_ctx* vits_pcm_gather_init(const char* id, size_t id_bytes, int32_t rank, int32_t world, int32_t rows, int64_t row_capacity,
                                               int32_t elem_bytes);
VITS_API int vits_pcm_gather(vits_gather_ctx* g, const void* pcm_device, int64_t pcm_stride, const int64_t* lengths_host, void* hip_stream,
                             vits_gather_result* out);
VITS_API void vits_pcm_gather_destroy(vits_gather_ctx* g);
VITS_API int vits_pcm_gather_verdict(const int64_t* table, int32_t world, int32_t rows, int64_t* stride_out);

/* Select the HIP device used by subsequent loads on this thread (one process per GPU: pass LOCAL_RANK). */
VITS_API int vits_set_device(int32_t device);

/* Device facts (for the bench's roofline block). */
VITS_API int vits_device_info(char* name, size_t cap, int32_t* cu_count, int32_t* clock_mhz, int64_t* hbm_bytes);

#endif /* VITS_HIP_H */
