// engine.h — host orchestrator: owns the device-resident (pre-packed) weights, the activation arenas, the HIP
// stream, and issues the kernel sequence that replaces the reference's two ggml graphs
// (/root/reference/src/vits.cpp:975-1080 build_graph_part_one/two, :1082-1099 execute_graph, :1101-1191 process).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <initializer_list>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/vits.h"
#include "kernels.h"
#include "model_file.h"

namespace vits {

struct Tokenizer {
    std::vector<std::pair<std::string, int32_t>> vocab;  // sorted by descending key length
    int32_t blank_id = 0;
    bool add_blank = true;
    // config key "phonetic" == "1" (vits_model_data.cpp:92-94 -> vits_tokenizer::set_phonetic): the model was trained on espeak-ng phonemes. The
    // reference built without VITS_ESPEAK asserts out at load (vits_tokenizer.cpp:176-178); espeak is out of scope here, so the model
    // loads and the TEXT entry points refuse (tokenize_checked) — the id entry points are what such a model is driven through.
    bool phonetic = false;
    void init(const ModelFile& f);
    std::vector<int32_t> tokenize(const std::string& text) const;
    // the text entry points of the C ABI: false + message for a phonetic model; add_blank == 0 gives the reference's EMPTY id list (Q11,
    // vits_tokenizer.cpp:200-208: tokens_final is only filled under add_blank), which the callers report as "empty input"
    bool tokenize_checked(const std::string& text, std::vector<int32_t>& ids, std::string& err) const;
};

struct Profiler {
    struct Rec {
        int name_id;
        hipEvent_t a, b;
        double flop, bytes;
        bool a_shared;  // a is the previous record's b (back-to-back launches share one event)
        hipEvent_t spare = nullptr;  // dispatch-attached stop event that a multi-launch span replaced by a recorded one
    };
    bool attach = true;  // events ride on the kernel's dispatch packet (kernels.h: LaunchTimer); false: hipEventRecord around every launch
    bool on = false;
    std::vector<std::string> names;
    std::unordered_map<std::string, int> ids;
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    struct Agg {
        long calls = 0;
        double ms = 0, flop = 0, bytes = 0;
    };
    std::vector<Agg> agg;
    hipEvent_t get();
    // chain = true: if the previous timed launch ended on the same stream with nothing in between, its end event doubles as
    // this launch's start (an event is a barrier packet on the queue: half as many of them in the timed region)
    void begin(const char* name, double flop, double bytes, hipStream_t s, bool chain = false);
    void fence() { last_ok = false; }  // something un-timed was queued (or the host waited): do not chain across it
    hipEvent_t last_b = nullptr;
    hipStream_t last_s = nullptr;
    bool last_ok = false;
    void end(hipStream_t s);
    void collect();  // after a stream sync
    void reset();
    std::string report();
    ~Profiler();
};

struct Arena {
    char* base = nullptr;
    size_t cap = 0, off = 0;
    hipError_t reserve(size_t bytes);  // grow-only; invalidates previous contents
    void reset() { off = 0; }
    template <class T>
    T* alloc(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = reinterpret_cast<T*>(base + off);
        off += n * sizeof(T);
        return off <= cap ? p : nullptr;
    }
    ~Arena();
};

struct Tap {
    float* dev = nullptr;  // snapshot [batch][channels][stride]
    int channels = 0, stride = 0;
    std::vector<int> lens;  // per utterance
};

struct EncoderLayerW {
    PackedConv qkv, out, ffn1, ffn2;
    float *rel_k, *rel_v, *ln1_g, *ln1_b, *ln2_g, *ln2_b;
};
struct DdsW {
    std::vector<float*> dw_w, dw_b, n1_g, n1_b, n2_g, n2_b;
    std::vector<PackedConv> pw;
};
struct DpFlowW {
    float *pre_w, *pre_b;
    DdsW dds;
    PackedConv proj;
};
struct FlowLayerW {
    PackedConv pre, post;
    std::vector<PackedConv> in_layers, res_skip;
};
struct ResBlockW {
    std::vector<PackedConv> c1, c2;
    std::vector<int> dil;
    int k;
};
struct UpStageW {
    PackedConv up;
    std::vector<ResBlockW> rbs;
    int channels, stride, k;
};

// Every VITS_* environment knob of the orchestrator, read ONCE when the model is loaded (never on the call path: getenv is
// not thread-safe against setenv, and a per-layer lookup is host time inside the caller's timed region). INTEGRATION.md §9.
struct Knobs {
    int rb_streams = 3;          // VITS_RB_STREAMS (1 serialises the three resblocks of a stage on the main stream)
    int lat16_lazy_tokens = 4096;  // VITS_LAT16_LAZY_TOKENS: a call of at most this many ids (batch x longest utterance) first makes the latency kernels' weight copy (0: never)
    bool lat16_eager = false;      // VITS_LAT16_EAGER: make that copy at load
    int rb16_serial_max_frames = 1300;  // VITS_RB16_SERIAL_MAX_FRAMES: 16-bit vocoder windows of at most this many frames (all utterances) use one stream ...
    int rb32_sum3_max_frames = 2000;   // VITS_RB32_SUM3_MAX_FRAMES: fp32 vocoder: the resblocks of a stage side by side + one sum launch while the call has fewer frames than this (batch 1 ... 8 x 128 ids: 2.80 -> 2.65, 5.06 -> 4.85 (3), 6.14 -> 5.89 (4), 10.82 -> 10.76 ms (8); 0: never)
    int rb16_serial_min_frames = 600;   // VITS_RB16_SERIAL_MIN_FRAMES: ... unless they have fewer than this (one or two 128-id utterances: kernels of 15-60 blocks, three of which side by side fill more of the chip than the fork / join costs — round 6, batch 1 / 2 / 4: 1.70 -> 1.60 / 1.77 -> 1.71 / 2.09 -> 2.17 ms with three streams)
    int lrelu_copy_minc = 128;   // VITS_LRELU_COPY_MINC: stages at least this wide also store leaky_relu(y)
    bool no_dds_fuse = false;    // VITS_NO_DDS_FUSE: DDS layer as three launches
    bool no_wn_fuse = false;     // VITS_NO_WN_FUSE: WaveNet layer as two launches
    bool no_group16 = false;     // VITS_NO_GROUP16: 16-bit vocoder through the fp32-layout (converter) path
    bool no_fuse16 = false;      // VITS_NO_FUSE16: 16-bit resblock conv pairs as two launches
    bool no_rbblock16 = false;   // VITS_NO_RBBLOCK16: 16-bit narrow-stage resblocks as three fused pairs instead of one kernel
    bool no_fuse32 = false;      // VITS_NO_FUSE32: fp32 resblock conv pairs as two launches
    bool no_rbblock32 = false;   // VITS_NO_RBBLOCK32: fp32 3-tap resblocks of the narrow stages as three fused pairs instead of one kernel
    bool no_rb_group = false;    // VITS_NO_RB_GROUP: the resblocks of a stage as separate launches (no grouped launch)
    bool rb_group_always = false;  // VITS_RB_GROUP=1: grouped launches also when the three streams are available
    bool no_flow_fuse = false;   // VITS_NO_FLOW_FUSE: 16-bit modes: a coupling layer of the flow as nine launches instead of one kernel
    bool prof_attach = true;     // VITS_PROF_ATTACH=0: per-kernel profiler with recorded events instead of dispatch-attached ones
    int front_prio = 1;          // VITS_FRONT_PRIO=0: the front-end stream of pipelined batches at normal instead of high priority
    bool keep_stage_sum32 = false;  // VITS_KEEP_STAGE_SUM32: 16-bit vocoder: also store the fp32 resblock sum of a stage's last resblock (nobody reads it)
    int split_min_batch = 128;   // VITS_SPLIT_MIN_BATCH: vits_model_process_batch splits batches of at least this many utterances in two pipelined parts (0: never).
                                 // Measured (f16, 128 ids): B = 64 split 32 + 32 LOSES (14.98 vs 14.34 ms: two half-size vocoder passes cost more than the hidden stage one)
    int split_first_pct = 50;    // VITS_SPLIT_FIRST_PCT: share of the utterances in the first part (its stage one is the exposed one)
    int flow_chains = 2;         // VITS_FLOW_CHAINS: 16-bit modes: the fused coupling layers of the flow as two independent chains of launches over halves of the batch (1: one chain)
    int flow_chain_min_blocks = 256;  // VITS_FLOW_CHAIN_MIN_BLOCKS: ... when a layer has more blocks than this (one block per CU at a time: 256 = one full round)
    int ref_ahead_frames_per_id = 6;  // VITS_REF_AHEAD_FRAMES_PER_ID: batch-1 calls with the reference noise stream draw the prior noise ahead, into a block sized for this many frames per id
                                      // (it grows if the utterance turns out longer; 0 = draw behind stage one as the reference does)
    bool no_pipeline = false;    // VITS_NO_PIPELINE: vits_model_submit_batch queues both stages on the main stream (no overlap)
    KernelKnobs kernel;          // the launch functions' own tuning knobs (kernels.h), installed per call by KernelKnobsScope
    void read();
};

// the reference stream of one batch-1 call drawn on a helper thread (engine_support.cpp): start, duration_noise, finish(n)
class RefNoiseAhead {
  public:
    RefNoiseAhead();
    ~RefNoiseAhead();
    RefNoiseAhead(const RefNoiseAhead&) = delete;
    RefNoiseAhead& operator=(const RefNoiseAhead&) = delete;
    void start(size_t n_dur, float* prior, size_t cap);  // draws n_dur values (the [T, 2] tensor), then the prior stream into `prior` (up to cap values)
    const float* duration_noise();                       // waits for the first tensor
    void rebase(float* prior, size_t cap);               // a larger prior buffer (values [0, drawn) copied by the caller), only when drawn() == capacity()
    void finish(size_t n);   // the prior tensor has n values: the global engine ends where exactly n draws leave it; joins the thread
    bool active() const { return active_; }
    size_t drawn() const;
    size_t capacity() const;

  private:
    struct Impl;
    Impl* impl_;
    bool active_ = false;
};

struct Call;    // engine_internal.h: the state of one process_batch call
struct WinCtx;  // engine_internal.h: one vocoder window

class Engine {
  public:
    ~Engine();
    bool load(const uint8_t* bytes, size_t size, std::string& err);
    // host-only: everything load() checks (format, hyper-parameters, every tensor's shape) without touching a device
    bool validate(const uint8_t* bytes, size_t size, std::string& err);
    int process_batch(const int32_t* ids, const int32_t* id_lens, int batch, int id_stride, const vits_process_opts& o, vits_batch_result* out,
                      std::string& err);
    // pipelined batches (include/vits.h vits_model_submit_batch / vits_model_wait): up to two in flight
    int submit_batch(const int32_t* ids, const int32_t* id_lens, int batch, int id_stride, const vits_process_opts& o, std::string& err);
    int wait_batch(vits_batch_result* out, std::string& err);
    int pending() const { return (int)(submit_seq_.load(std::memory_order_acquire) - wait_seq_.load(std::memory_order_acquire)); }  // (any thread may ask)
    int sync(std::string& err);
    // One call at a time per handle (the reference's contract too: process writes member tensors, src/include/vits.h:22-30). The ABI
    // takes this flag around every entry point that touches the engine; a second thread gets "model busy" instead of a race.
    std::atomic<bool> busy{false};
    int set_arith(int arith, std::string& err);  // VITS_ARITH_*: packs the 16-bit weight fragments on first use
    // EMULATED ggml fp16 lookup tables for ggml_gelu / ggml_soft_max (Q8; inferred from upstream ggml, the fork is absent): builds the two
    // tables on the host as ggml_init does and uploads them on first use
    // mode 1: stage one additionally runs in the exact order of include/vits_exact_math.h (exact_stage1.hip), shared with the oracle: durations are
    // bit-identical to the oracle's. mode 2: the tables inside the throughput kernels (their own summation order: statistical agreement only).
    int set_ggml_tables(int mode, std::string& err);
    int ggml_tables = 0;
    int arith = VITS_ARITH_F32;
    // which convolutions a 16-bit arithmetic mode applies to (include/vits.h VITS_ARITH_SCOPE_*)
    int arith_scope = VITS_ARITH_SCOPE_FLOW_VOCODER;
    Knobs knobs;
    int64_t get_tap(const char* name, int utt, float* dst, size_t cap);

    HParams hp;
    Tokenizer tok;
    int mode = VITS_MODE_REFERENCE;
    int64_t weight_bytes = 0;
    Profiler prof;
    hipStream_t stream = nullptr;

  private:
    // weights
    float* emb_ = nullptr;
    std::vector<EncoderLayerW> enc_;
    PackedConv enc_proj_;
    PackedConv dp_pre_, dp_proj_;
    DdsW dp_dds_;
    float *dp_translate_ = nullptr, *dp_logscale_ = nullptr;
    std::vector<DpFlowW> dp_flows_;  // index f-1 for flows.f, f = 1..dp_flows
    std::vector<FlowLayerW> flow_;
    PackedConv dec_pre_;
    std::vector<UpStageW> ups_;
    float* dec_post_w_ = nullptr;
    int dec_post_cin_ = 0, dec_post_k_ = 0;
    // host copies of every Conv1d / ConvTranspose1d weight (torch layout, after the flip / negation folds, in the file's storage
    // type), kept so that set_arith can pack the 16-bit A fragments on demand. Linear layers (q/k/v/out) are not listed: they stay fp32 (Q7).
    struct PackSrc {
        PackedConv* pc;  // (points into enc_/flow_/ups_...: those vectors are sized before their entries are packed and never resized)
        std::vector<float> w32;     // fp32-stored tensors
        std::vector<uint16_t> w16;  // fp16 / bf16-stored tensors, as stored
        uint32_t dtype;
        int cout, cin, k, epi, ct_stride;
        std::vector<float> widen() const {
            if (dtype != DT_F16 && dtype != DT_BF16) return w32;
            std::vector<float> w(w16.size());
            for (size_t e = 0; e < w.size(); ++e) w[e] = dtype == DT_F16 ? f16_to_f32(w16[e]) : bf16_to_f32(w16[e]);
            return w;
        }
    };
    std::vector<PackSrc> packs_;
    struct Lat16Lazy {
        PackedConv* pc;
        size_t n;  // floats of the packed array
    };
    std::vector<Lat16Lazy> lat16_lazy_;  // layers that get a wp_l16 copy at the first small call (ensure_lat16)
    bool lat16_ready_ = false;
    int ensure_lat16(std::string& err);
    Ref16 x16_[3];           // per-stream scratch for the 16-bit copy of a conv input (transparent 16-bit path)
    size_t x16_cap_[3] = {0, 0, 0};
    bool vocoder_group_ok_ = false;  // every vocoder channel count is a multiple of 8: group-layout fast path available
    hipError_t conv16_transparent(const char* name, const PackedConv& w, const ConvCall& c, hipStream_t stream);
    hipError_t conv16(const char* name, const PackedConv& w, const Conv16Call& c, hipStream_t stream, double bytes);
    std::vector<void*> owned_;  // every device allocation made at load
    bool dry_run_ = false;

    // stage-one arenas: one per pipeline slot (a batch's stage two reads its stage-one results while the next batch's stage one runs)
    Arena a1_[2], a2_;
    int a1_slot_ = 0;
    Arena& a1() { return a1_[a1_slot_]; }
    // A batch submitted with submit_batch and not yet waited for. Its results are known on the host when submit returns (the frame
    // counts were read there); `done` marks the end of its device work, `host` is pinned staging for the PCM when a host copy was asked for.
    struct Pending {
        bool active = false;
        int B = 0;
        size_t stride = 0;
        std::vector<int64_t> lengths, frames;
        bool host_copy = false;
        float* host = nullptr;
        size_t host_cap = 0;  // bytes
        int* frames_pinned = nullptr;
        size_t frames_cap = 0;  // ints
        int* win_pinned = nullptr;  // vocoder-window length table of a windowed batch (host side of its H2D copy)
        size_t win_cap = 0;         // ints
        hipEvent_t s1_done = nullptr, done = nullptr;
    } pend_[2];
    std::atomic<uint64_t> submit_seq_{0}, wait_seq_{0};  // batch n lives in pend_[n & 1]; written under the busy flag, read by vits_model_pending
    hipStream_t front_ = nullptr;             // stage one of pipelined batches (created on first use)
    // batch-1 calls with the reference noise stream: the prior noise is drawn into pinned memory while stage one runs (engine.cpp)
    float* ref_noise_pinned_ = nullptr;
    size_t ref_noise_cap_ = 0;  // floats
    RefNoiseAhead ref_ahead_;
    float* dur_noise_pinned_ = nullptr;  // the [T, 2] duration noise of such a call
    size_t dur_noise_cap_ = 0;
    hipEvent_t dur_noise_ev_ = nullptr;
    bool async_tail_ = false;                 // the last process_batch returned with device work still queued (opts.async)
    hipEvent_t ev_async_ = nullptr;           // orders the front-end stream behind that tail
    int process_split(const int32_t* ids, const int32_t* id_lens, int batch, int id_stride, const vits_process_opts& o, vits_batch_result* out, std::string& err);
    int process_impl(const int32_t* ids, const int32_t* id_lens, int batch, int id_stride, const vits_process_opts& o, vits_batch_result* out, std::string& err,
                     Pending* pend);
    // The three resblocks of a vocoder stage (kernel sizes 3/7/11) are independent chains of six convolutions; they run on
    // three streams so that the tail of one kernel's grid overlaps the head of another's. side_[j-1] carries resblock j.
    hipStream_t side_[2] = {nullptr, nullptr};
    hipEvent_t ev_fork_ = nullptr, ev_done_[3] = {nullptr, nullptr, nullptr};
    int halo_frames_ = 0;      // receptive field of the vocoder in frames, one side (computed at load)
    void* pinned_ = nullptr;   // grow-only pinned staging for streamed PCM
    int* frames_host_ = nullptr;  // pinned [frames_host_cap_]: destination of a synchronous call's frame-count copy (into pageable memory the copy went through a staging buffer: + 15 us at batch 1)
    size_t frames_host_cap_ = 0;
    size_t pinned_cap_ = 0;
    // arithmetic of the convolutions being queued right now: `arith`, or fp32 while stage one runs under
    // VITS_ARITH_SCOPE_FLOW_VOCODER (every conv wrapper and fused kernel reads this one, never `arith` itself)
    int arith_now_ = VITS_ARITH_F32;
    // VITS_ARITH_F32_SPLIT: every dispatch sees VITS_ARITH_F32 (arith_kernels()); only the vocoder's resblock scheduling asks split_on()
    int arith_kernels() const { return arith == VITS_ARITH_F32_SPLIT ? VITS_ARITH_F32 : arith; }
    bool split_on() const { return arith == VITS_ARITH_F32_SPLIT; }
    // emulated-ggml mode 1: the stage-one tensors as the file holds them (host, storage type) and their fp32 device copies (first use)
    std::vector<TensorEntry> exact_src_;
    struct ExactTensor {
        const float* d = nullptr;
        int rank = 0;
        int64_t ne[4] = {1, 1, 1, 1};
    };
    std::map<std::string, ExactTensor> exact_w_;
    int run_stage_one_exact(Call& c);
    GgmlTables ggml_tabs_;              // what the stage-one kernels receive: null pointers unless ggml_tables
    uint16_t* ggml_tab_dev_ = nullptr;  // [2][65536]: gelu, exp
    struct HStage {  // pinned staging of the per-call host header (ids, lengths, stage tables)
        int* p = nullptr;
        size_t cap = 0;
        hipEvent_t ev = nullptr;
        bool pending = false;
    } hstage_[2];
    int hstage_next_ = 0;
    std::map<std::string, Tap> taps_;
    int tap_batch_ = 0;

    float* upload(const std::vector<float>& v);
    struct ConvShape {
        int cout, cin, k;
    };
    float* upload_tensor(const ModelFile& f, const std::string& name, std::string& err, std::initializer_list<int64_t> want);
    bool pack(const ModelFile& f, const std::string& wname, const std::string& bname, int epi, ConvShape want, PackedConv& out, std::string& err,
              int ct_stride = 0, int transform = 0);
    bool load_dds(const ModelFile& f, const std::string& base, DdsW& d, std::string& err);
    hipError_t conv(const char* name, const PackedConv& w, ConvCall c, hipStream_t on = nullptr);  // on == nullptr: the main stream
    hipError_t run_dds(const DdsW& d, TensorRef x, TensorRef y, TensorRef p, const int* lens, int batch, int tmax, int64_t sum_t);
    // The DDS block on the latency kernels (stage1_lat.hip) with the per-token ops around it fused in: the head (a conv flow's 1 -> H conv +
    // conditioning, or an H -> H 1x1 conv of head_x) and the 1x1 conv behind the block (tail -> tail_y). Scratch a, b: [B][H][ts].
    struct DdsEnds {
        const float *head_w = nullptr, *head_b = nullptr;  // conv flow: pre weights / bias, latent z row zc, conditioning
        TensorRef z, cond;
        int zc = 0;
        const PackedConv* head_conv = nullptr;  // or: 1x1 conv of head_x
        TensorRef head_x;
        const PackedConv* tail_conv = nullptr;
        TensorRef tail_y;
    };
    bool dds_lat_ok(const DdsW& d, const DdsEnds& e, int batch, int tmax) const;
    hipError_t run_dds_lat(const DdsW& d, const DdsEnds& e, TensorRef a, TensorRef b, const int* lens, int batch, int tmax, int64_t sum_t);
    void snapshot(const char* name, TensorRef t, int channels, int stride, int batch, const std::vector<int>& lens);
    void clear_taps();
    // the phases of one call (engine_stage1.cpp, engine_flow.cpp, engine_vocoder.cpp); each returns 0 or -1 with c.err set
    int layout_stage_one(Call& c);
    int run_text_encoder(Call& c);
    int run_duration_predictor(Call& c);
    int layout_stage_two(Call& c);
    int run_prior_sampling(Call& c);
    int run_flow(Call& c);
    int run_vocoder_window32(Call& c, WinCtx& w);
    int run_vocoder_window16(Call& c, WinCtx& w);
};

}  // namespace vits

struct vits_model {
    vits::Engine eng;
};
