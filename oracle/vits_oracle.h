/*
 * vits_oracle.h — C API of the CPU ORACLE (libvits_oracle.so).
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT. This library is a plain-C++ CPU restatement of the algorithm the
 * reference implements for the hot path (/root/reference/src/vits.cpp:115-1191 + src/include/custom-ops.h +
 * src/include/ggml-util.h). Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product library (libvits_hip.so) never links, loads or calls it.
 *
 * PARITY PIN: the reference itself cannot be built here (its ggml fork submodule is an empty directory and
 * its weights are git-LFS pointers), so the oracle is pinned by
 *   (1) golden vectors generated in the build container from transformers.VitsModel — the model the
 *       reference ports (src/vits.cpp:113) and was verified against (scripts/verify_layers.py:25) —
 *       tests/golden/ (npz files), generator tests/golden/make_golden.py; oracle mode VO_MODE_HF must match them;
 *   (2) the reference's own helper-op known-answer vectors (test/test_ggml_utils.cpp:458-606);
 *   (3) the libstdc++ noise-stream known answer.
 * Mode VO_MODE_REFERENCE then applies the reference's literal deviations from HF,
 * each restated from the cited reference lines and pinned by fixtures of a transformers.VitsModel patched with torch
 * restatements of the same lines (tests/golden/: the _refmode_taps and _q6_refmode_taps files) — the
 * ggml arithmetic below the call sites (fp16 im2col, GELU/softmax tables; Q7/Q8) is "parity unpinned".
 */
#ifndef VITS_ORACLE_H
#define VITS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
#define VO_API extern "C" __attribute__((visibility("default")))
#else
#define VO_API __attribute__((visibility("default")))
#endif

typedef struct vo_model vo_model;
typedef struct vo_run vo_run;

#define VO_MODE_REFERENCE 0
#define VO_MODE_HF 1
/* conv arithmetic (Q7): F32 = exact; F16 = the reference's fp16 im2col x fp16 weights -> fp32 (custom-ops.h:684-690); BF16 likewise */
#define VO_ARITH_F32 0
#define VO_ARITH_BF16 1
#define VO_ARITH_F16 2
/* scope of a 16-bit arithmetic: FLOW_VOCODER = the coupling flow and HiFiGAN only (text encoder and duration predictor stay
 * exact fp32, so the integer durations do not depend on the arithmetic); ALL_CONVS = the literal Q7 (every conv of the graph) */
#define VO_SCOPE_FLOW_VOCODER 0
#define VO_SCOPE_ALL_CONVS 1
#define VO_NOISE_REFERENCE 0
#define VO_NOISE_COUNTER 1
#define VO_NOISE_EXPLICIT 2

typedef struct vo_opts {
    int32_t mode;
    int32_t noise_kind;
    uint64_t noise_seed;        /* counter noise: seed for THIS utterance (caller adds the utterance index) */
    const float* noise_dur;     /* explicit: [2][T] */
    const float* noise_prior;   /* explicit: [192][noise_prior_stride] */
    int64_t noise_prior_stride;
    int32_t fixed_duration;     /* >0: pin every id to this many frames */
    int32_t threads;            /* <=0: max(hardware_concurrency, 6) like src/include/common.h:19-21 */
    int32_t arith;              /* VO_ARITH_*: operand rounding of every Conv1d / ConvTranspose1d (not of the Linear layers) */
    int32_t arith_scope;        /* VO_SCOPE_*: which convs `arith` applies to */
    int32_t ggml_tables;        /* EMULATE ggml's fp16 lookup tables (Q8; inferred from upstream ggerganov/ggml, the reference's fork is absent):
                                   ggml_gelu = tanh-GELU through table_gelu_f16 (vits.cpp:673,687), ggml_soft_max = exp through table_exp_f16, double
                                   sum, multiply by (float)(1/sum) (vits.cpp:329,719,735). 0 (default): erf-GELU, fp32 soft-max.
                                   1: stage one in the exact order of include/vits_exact_math.h (vits_oracle_exact.cpp), which the product's
                                      vits_model_set_ggml_tables(model, 1) shares: bit-identical log-durations on both sides.
                                   2: this file's own loops with the table lookups — the independent restatement mode 1 is compared with at
                                      tolerance, and the counterpart of the product's mode 2. */
} vo_opts;

VO_API const char* vo_last_error(void);
VO_API vo_model* vo_load(const char* bytes, size_t size);
VO_API void vo_free(vo_model* m);
VO_API int32_t vo_num_tensors(const vo_model* m);
/* tensor access for the reader tests: copies widened fp32 data, returns element count; dims in file order */
VO_API int64_t vo_tensor(const vo_model* m, const char* name, float* dst, size_t cap, int32_t* dtype, int32_t* rank,
                         int64_t* dims4);
VO_API int64_t vo_config(const vo_model* m, const char* key, char* dst, size_t cap);

/* full forward for one utterance of T ids; returns all stage taps */
VO_API vo_run* vo_process_ids(vo_model* m, const int32_t* ids, int32_t T, const vo_opts* opts);
/* stage one only: log-durations [T] and durations [T] (= ceil(exp(logw) * length_scale), vits.cpp:995-1001) of one utterance */
VO_API int vo_log_durations(vo_model* m, const int32_t* ids, int32_t T, const vo_opts* opts, float* logw_out, float* dur_out);
/* tap names as in include/vits.h vits_model_get_tap */
VO_API int64_t vo_run_tap(const vo_run* r, const char* name, float* dst, size_t cap);
VO_API void vo_run_free(vo_run* r);

/* tokenizer restatement (src/vits_tokenizer.cpp:182-208, deterministic longest-match) */
VO_API int64_t vo_tokenize(const vo_model* m, const char* text, int32_t* ids, size_t cap);

/* reference noise stream (vits.cpp:31, ggml-util.h:187-199) */
VO_API void vo_reference_noise_seed(uint32_t seed);
VO_API void vo_reference_noise_draw(float* dst, size_t n);

/* operator-level restatements; same descriptors as include/vits.h (layout [B][C][T], time fastest) */
typedef struct vo_conv1d_desc {
    int32_t batch, cin, cout, t, t_stride;
    int32_t k, dilation, pad_left;
    int32_t pre_act;
    float pre_slope;
    int32_t post_act;
    float out_scale;
    int32_t arith; /* VO_ARITH_* */
} vo_conv1d_desc;
VO_API int vo_conv1d(const vo_conv1d_desc* d, const float* x, const float* w, const float* bias, const float* residual,
                     const float* accum, const int32_t* lens, float* y, int32_t threads);
typedef struct vo_convt1d_desc {
    int32_t batch, cin, cout, t, t_stride, t_out_stride;
    int32_t k, stride, crop;
    float pre_slope;
    int32_t arith; /* VO_ARITH_* */
} vo_convt1d_desc;
VO_API int vo_conv_transpose1d(const vo_convt1d_desc* d, const float* x, const float* w, const float* bias,
                               const int32_t* lens, float* y);
VO_API int vo_rel_attention(int32_t batch, int32_t heads, int32_t head_dim, int32_t t, int32_t t_stride, int32_t window,
                            const float* q, const float* k, const float* v, const float* rel_k, const float* rel_v,
                            const int32_t* lens, float* out);
VO_API int vo_add_layer_norm(int32_t batch, int32_t channels, int32_t t, int32_t t_stride, float eps, const float* x,
                             const float* residual, const float* gamma, const float* beta, float* y);

/* helper-op restatements on dense ggml-ordered tensors ne=[ne0,ne1,ne2], data[(i2*ne1+i1)*ne0+i0]
 * (src/include/ggml-util.h:16-276, src/include/custom-ops.h:218-395,739-862) — pinned by the reference's
 * known-answer vectors (test/test_ggml_utils.cpp:458-606). Each writes dst and returns the output ne. */
VO_API void vo_pad_3d(const float* src, const int64_t ne[3], const int32_t pads[6], float* dst, int64_t out_ne[3]);
VO_API void vo_slice_3d(const float* src, const int64_t ne[3], const int32_t se[6], float* dst, int64_t out_ne[3]);
VO_API void vo_flip_3d(const float* src, const int64_t ne[3], int32_t along, float* dst);
VO_API void vo_concat_3d(const float* a, const int64_t ane[3], const float* b, const int64_t bne[3], int32_t dim, float* dst,
                         int64_t out_ne[3]);
VO_API void vo_compare(const float* a, const float* b, int64_t n, int32_t op /*0:<,1:>=,2:<=*/, float* dst);
VO_API void vo_per_row_cumsum(const float* src, const int64_t ne[3], float* dst);
VO_API float vo_max(const float* src, int64_t n);
VO_API void vo_binary_not(const float* src, int64_t n, float* dst);
VO_API void vo_index_put_last_dim(float* t, const int64_t ne[3], int32_t index, float value);
VO_API void vo_index_add_last_dim(float* t, const int64_t ne[3], int32_t index, float value);
VO_API void vo_masked_set(const float* t, const float* mask, const float* values, int64_t n, float* dst);
VO_API int64_t vo_masked_get_compact(const float* t, const float* mask, int64_t n, float* dst);
/* masked_get as custom-ops.h:739-762 implements it (shape kept, zeros where the mask is 0): the form vits.cpp:832-840 consumes (Q6) */
VO_API void vo_masked_get(const float* t, const float* mask, int64_t n, float* dst);
/* number of duration-predictor latents that lay outside [-tail_bound, tail_bound] in the last vo_process_ids / vo_log_durations call of
 * the calling thread (summed over the three spline flows): 0 means the Q6 misalignment of vits.cpp:832-849 did not come into play */
VO_API int64_t vo_outside_latents(void);
VO_API void vo_gather0(const float* t, const int64_t ne[3], const float* index, int64_t n_index, float* dst);
VO_API void vo_arange(int32_t end, float* dst);

#endif
