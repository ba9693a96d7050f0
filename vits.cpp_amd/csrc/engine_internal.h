// engine_internal.h — shared by the engine's translation units only (engine.cpp: one call from ids to PCM; engine_load.cpp:
// weights; engine_stage1.cpp: text encoder + duration predictor; engine_flow.cpp: prior sampling + coupling flow;
// engine_vocoder.cpp: HiFiGAN; engine_support.cpp: profiler, arenas, tokenizer, noise, roctx ranges).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "engine.h"

namespace vits {

// ---- roctx ranges (VITS_ROCTX=1): the phases of a call as marker ranges for `rocprofv3 --marker-trace --kernel-trace` ---------
// The marker library is looked up at run time (librocprofiler-sdk-roctx.so, else libroctx64.so): no link-time dependency, and
// nothing at all happens unless the variable is set. Host-side ranges: they bracket the ENQUEUE of a phase's kernels (the call
// is asynchronous up to the one frame-count read-back), which is what a timeline viewer lines up with the kernel trace.
struct RoctxApi {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    RoctxApi();
};
const RoctxApi& roctx_api();
// consecutive phases of one call: phase(n) closes the previous range and opens the next; the destructor closes the last
struct RoctxPhases {
    bool open = false;
    void phase(const char* name) {
        const RoctxApi& a = roctx_api();
        if (!a.push) return;
        if (open) a.pop();
        a.push(name);
        open = true;
    }
    ~RoctxPhases() {
        if (open) roctx_api().pop();
    }
};
// a nested range (one vocoder stage)
struct RoctxRange {
    bool open = false;
    explicit RoctxRange(const char* name) {
        const RoctxApi& a = roctx_api();
        if (a.push) {
            a.push(name);
            open = true;
        }
    }
    ~RoctxRange() {
        if (open) roctx_api().pop();
    }
};

#define HIP_OK(expr)                                                 \
    do {                                                             \
        hipError_t e_ = (expr);                                      \
        if (e_ != hipSuccess) {                                      \
            err = std::string(#expr) + ": " + hipGetErrorString(e_); \
            return -1;                                               \
        }                                                            \
    } while (0)

#define KPROF(name, call)                  \
    do {                                   \
        prof.begin(name, 0, 0, stream);    \
        hipError_t e__ = (call);           \
        prof.end(stream);                  \
        if (e__ != hipSuccess) return e__; \
    } while (0)

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

static inline TensorRef make_ref(float* p, int channels, int stride) {
    TensorRef t;
    t.p = p;
    t.cs = stride;
    t.bs = (int64_t)channels * stride;
    return t;
}
static inline TensorRef sub_rows(TensorRef t, int c0) {
    t.p += (int64_t)c0 * t.cs;
    return t;
}

void reference_noise_fill(float* dst, size_t n);  // the reference's process-global libstdc++ stream (vits.cpp:31)

// ---- everything one vits_model_process_batch call carries from phase to phase -------------------------------------------------
struct Call {
    const vits_process_opts& o;
    std::string& err;
    const int32_t* ids;
    int B, id_stride;
    int md = 0;  // resolved semantics mode
    bool refmode = true;
    int Tmax = 0, ts = 0, n_up = 0;
    std::vector<int> tlen;
    int64_t sum_t = 0;  // (profiler accounting: real work = sum of the utterance lengths)
    RoctxPhases rx;

    // stage one (sized by B x T)
    struct S1 {
        int *ids, *lens, *cum, *frames, *stage_lens, *stage_mul, *stage_add, *seed_off;
        float *x, *qkv, *att, *tmp, *ffn, *stats, *dpx, *dpy, *dpp, *cond, *z, *u, *dur;
        float *ex_scores, *ex_tok;  // emulated-ggml mode 1 only
        uint16_t* x16;
    } s1{};
    std::vector<int> smul, sadd;  // vocoder stage lengths as affine functions of the frame count: len_i = L * smul[i] + sadd[i]
    int c_first = 0;              // physical row of z holding the log-durations after the duration predictor's flips
    RefNoiseAhead* ref_ahead = nullptr;  // batch 1, reference noise: the prior noise drawn while stage one runs (engine.cpp)

    // the one data-dependent shape (vits.cpp:1133)
    std::vector<int> frames;
    int Lmax = 0;
    std::vector<std::vector<int>> slen;  // [stage][utterance]
    std::vector<int> smax;               // [stage] longest
    int64_t sum_frames = 0;

    // vocoder windows (long-form / streaming)
    struct Win {
        int f0, f1, lo, hi;
    };
    std::vector<Win> wins;
    bool windowed = false;
    int Lw_max = 0, M = 1;

    // stage two (sized by B x L after the host read of the frame counts)
    struct S2 {
        float *zp, *noise, *hout, *gate, *h0, *bu, *bul, *by[3], *bt[3], *byl[3], *bs, *bs16, *pre, *wave;
        uint16_t* x16[3];
        uint16_t *sp_u, *sp_t[3], *sp_y[3];  // VITS_ARITH_F32_SPLIT: split planes of the stage input, and per concurrent resblock of t and of the stream
        int* win_lens;  // [window][n_up + 2][B]: stage lengths of each utterance inside the window, then its emit end
    } s2{};
    int ls = 0, lws = 0, S_stride = 0;
    size_t big = 0;        // floats of the largest vocoder activation (of one window)
    std::vector<int> sts;  // [stage] time stride of that stage's buffers
    bool fast16 = false, fuse16 = false;
    const int* d_len_full[8] = {};
    float* wave_dst = nullptr;
    int64_t wave_stride = 0;

    Call(const vits_process_opts& o_, std::string& err_, const int32_t* ids_, int B_, int id_stride_) : o(o_), err(err_), ids(ids_), B(B_), id_stride(id_stride_) {}
};

// one vocoder window, window-local lengths
struct WinCtx {
    size_t wi;
    Call::Win wn;
    int Lw;
    const int* d_len[8];
    std::vector<int> smax;      // window-local maxima per stage
    std::vector<int64_t> ssum;  // (profiler accounting) sum over utterances of the stage lengths inside this window
    const int* emit_hi;
    int emit_lo;
    TensorRef zwin, pre, wv;
    float final_slope;
};

}  // namespace vits
