// kernels.h — launch interface of the hand-written gfx950 kernels (host side sees only these functions).
//
// Data layout everywhere: activations are fp32 [batch][channel][time], TIME CONTIGUOUS (== the reference's ggml
// ne order [time, channel, batch], so a wavefront's 64 lanes read 64 consecutive time steps = 256 B coalesced).
// Every tensor argument carries its own batch stride (bs) and channel stride (cs), in floats. Ragged batches:
// `len[b]` is the number of valid time steps of utterance b; loads beyond it read as ZERO (utterance boundaries
// behave exactly like the zero padding a batch-1 run sees) and stores beyond it are suppressed.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <vector>

namespace vits {

// Tuning knobs of the launch functions: which tile shape / kernel variant a launch takes. None of them changes a bit of a result (GPU test:
// tools/knob_identity.py). They are read from the environment ONCE per model handle, when it is loaded (Engine::knobs.kernel), and installed
// for the calling thread around every engine entry point (KernelKnobsScope in the ABI); the model-free entry points (vits_op_*) and the
// developer harnesses under tools/ see the process defaults, read at first use. Header-only so that the harnesses that include a single
// kernel file link without the engine.
struct KernelKnobs {
    int narrow_tiles = 64;       // VITS_NARROW_TILES: fp32 convs of at most this many rows take 128-column tiles (0: 256-column tiles)
    int tile128 = 1;             // VITS_TILE128=0: no 128 x 128 tiles in conv_mfma
    int min_blocks = 0;          // VITS_MIN_BLOCKS: grid size under which conv_mfma steps down to a smaller tile (0: 1024 for k <= 3, else 512)
    bool no_narrow = false;      // VITS_NO_NARROW: no 32-column strips for tiny grids / 1x1 convs
    int narrow_k1 = 1 << 30;     // VITS_NARROW_K1: longest sequence whose 1x1 convs take the narrow tile (0: small grids only)
    int nbuf = 0;                // VITS_NBUF=2|3: LDS ring depth of conv_mfma (0: chosen per launch)
    bool no_oneshot = false;     // VITS_NO_ONESHOT: no cooperative one-shot fill for latency-bound launches
    int db_min = 1;              // VITS_DB_MIN=2: register-staged kernels for single-chunk inputs
    int t16_tile0 = 0;           // VITS_T16_TILE0: conv16 tile for c_out multiples of 128 (0: 128 x 128)
    bool no_convt16s = false;    // VITS_NO_CONVT16S: 16-bit transposed convs through conv16's polyphase epilogue
    bool convt16s_all = false;   // VITS_CONVT16S_ALL: the one-row-tile streaming kernel also for stride 8
    bool no_convt16l = false;    // VITS_NO_CONVT16L: no four-phase lines kernel for strides that are multiples of 4
    bool no_att_lat = false;     // VITS_NO_ATT_LAT: small attention grids (at most 128 blocks) without the operand prefetch across phases (the eight-wave kernel of round 6's first step)
    bool att_valu = false;       // VITS_ATT_VALU: attention without the matrix cores
    int att_nw = 0;              // VITS_ATT_NW=4|8: waves per attention block (0: by LDS footprint)
    int att_short = 512;         // VITS_ATT_SHORT: longest sequence (tokens) for the short-sequence attention variant (0: off)
    int ln_tw = 32;              // VITS_LN_TW=64: LayerNorm tiles of 64 time steps (sixteen waves)
    bool rbb_c128 = true;        // VITS_RBB_C128=0: C = 128, k = 3 resblocks as fused pairs instead of rbblock16
    int rbb_c64k11 = -1;         // VITS_RBB_C64K11: C = 64, k = 11 resblocks through rbblock16 as well: 1 always, 0 never (three fused pairs), -1 where the grid is large enough for segments of tiles (VITS_RBB_STREAM_*)
    int rbb_stream_tiles = -1;   // VITS_RBB_STREAM_TILES: tiles a block of rbblock16 walks with the left halo taken from the previous tile (-1: per shape, see launch_rbb; 0 / 1: always one tile per block, both halos recomputed; N: N for every shape)
    int rbb_stream_min_blocks = 1536;  // VITS_RBB_STREAM_MIN_BLOCKS: ... while the segments still number at least this many blocks (two rounds of 3 x 256)
    int fuse16_maxc = 256;       // VITS_FUSE16_MAXC: widest stage whose 16-bit conv pairs are fused
    bool fuse32_c128 = true;     // VITS_FUSE32_C128=0: fp32 k = 3 pairs at C = 128 as two launches
    int wn16_ncw = 1;            // VITS_WN16_NCW=2: 16-bit WaveNet layer with six waves, both column tiles each
    int flow_ncw = 2;            // VITS_FLOW_NCW=1: 16-bit coupling-layer kernel with one column tile per wave
    int convt16_r128 = 0;        // VITS_CONVT16_R128: developer override of the streaming transposed conv's shape for 128-row layers (the 128 -> 64 stride-2 upsampler): nr * 100 + csplit * 10 + (rs == 16), e.g. 211 = <2, 1, 16>; 0 = default <4, 1, 16>
    int convt16_split_max = 64;  // VITS_CONVT16_SPLIT_MAX: 16-bit stride-8 upsamplers deal the units of a position tile out over up to four blocks while the launch has at most this many tiles (0: never)
    bool no_rbb_group3_c64 = false;   // VITS_NO_RBB_GROUP3_C64: ... only the C = 32 stage grouped
    bool no_rbb_group3 = false;       // VITS_NO_RBB_GROUP3: the C = 32 stage's three whole-resblock kernels always as three launches (small grids: not one grouped launch)
    bool no_rb_sum3_f32 = false;      // VITS_NO_RB_SUM3_F32: fp32 path: the resblocks of a small-grid stage always chained through the shared sum
    bool rb_sum3_in_order = false;    // VITS_RB_SUM3_IN_ORDER: side-by-side resblocks enqueued first to last, the first on the main stream (until round 6's last step; default: the last — longest — first and on the main stream)
    bool rb_sum3_block_only = false;  // VITS_RB_SUM3_BLOCK_ONLY: the separate sum launch only for stages whose resblocks are all whole-resblock kernels (until round 6's last step: every small-grid stage)
    bool no_rb_sum3 = false;     // VITS_NO_RB_SUM3: the resblocks of an all-whole-resblock stage always chained through the shared sum (no separate sum launch on small grids)
    int rb16_narrow_max = 128;   // VITS_RB16_NARROW_MAX: 16-bit fused pairs at C >= 128 on 64-column blocks while the 128-column tile would give at most this many blocks (0: never; 64 until round 6: batch 4 / 6 -1 ... -2 % with 128)
    int flow_narrow_max = 96;    // VITS_FLOW_NARROW_MAX: 16-bit coupling-layer kernel on 16-frame blocks while the 48-frame tile would give at most this many blocks (0: never)
    bool no_ln_fuse = false;     // VITS_NO_LN_FUSE: the encoder's LayerNorms always as their own launches (never applied on load by the consuming conv_lat16_kernel)
    bool no_dds_lat = false;     // VITS_NO_DDS_LAT: the duration predictor's DDS layers always on dds_layer_kernel (no 16-token latency kernel, no fused 1x1 convs around it)
    int dds_lat_max_blocks = 96; // VITS_DDS_LAT_MAX_BLOCKS: the latency kernel is taken while batch x ceil(tokens / 16) is at most this
    int lat16h_group_shape = 21;  // VITS_LAT16H_GROUP_SHAPE: 10 x row tiles + column tiles per block of conv16_lat_group_kernel (21, 41, 22, 42)
    bool no_lat16h_group = false; // VITS_NO_LAT16H_GROUP: the C = 256 stage's latency convs as 18 launches on three streams (not six grouped launches on one)
    bool no_lat16h_pre = false;  // VITS_NO_LAT16H_PRE: 16-bit modes: the vocoder's conv_pre always as converter launch + conv16_kernel
    bool no_lat16h = false;      // VITS_NO_LAT16H: 16-bit modes: the wide stages' resblock convs on small grids as fused pairs (rbpair16) / conv16_kernel, never conv16_lat_kernel
    int lat16h_max_tiles = 2048; // VITS_LAT16H_MAX_TILES: ... while a C = 256 conv has at most this many 32 x 32 output tiles (batch 1 ... 4 x 128 ids)
    int lat16h_max_tiles_c128 = 0;  // VITS_LAT16H_MAX_TILES_C128: the same for C = 128 (batch 1 = 1792 tiles measured + 6 ... 12 us against the fused pairs: off)
    int lat16h_shape = 21;       // VITS_LAT16H_SHAPE: 10 x waves per block + 32-column tiles per wave of conv16_lat_kernel (21, 22, 42)
    bool no_lat16 = false;       // VITS_NO_LAT16: tiny grids with long K chains on the 128 x 32 tile of conv_mfma instead of conv_lat16_kernel
    int lat16_max_waves = 768;   // VITS_LAT16_MAX_WAVES: standard convs with at most this many 32 x 32 output tiles take conv_lat16_kernel (0: tiny grids only)
    static KernelKnobs from_env() {
        KernelKnobs k;
        auto num = [](const char* name, int& v) {
            if (const char* e = getenv(name)) v = atoi(e);
        };
        auto flag = [](const char* name, bool& v) { v = getenv(name) != nullptr; };
        num("VITS_NARROW_TILES", k.narrow_tiles);
        flag("VITS_NO_DDS_LAT", k.no_dds_lat);
        flag("VITS_NO_LN_FUSE", k.no_ln_fuse);
        flag("VITS_NO_RB_SUM3", k.no_rb_sum3);
        flag("VITS_RB_SUM3_BLOCK_ONLY", k.rb_sum3_block_only);
        flag("VITS_RB_SUM3_IN_ORDER", k.rb_sum3_in_order);
        flag("VITS_NO_RB_SUM3_F32", k.no_rb_sum3_f32);
        flag("VITS_NO_RBB_GROUP3", k.no_rbb_group3);
        flag("VITS_NO_RBB_GROUP3_C64", k.no_rbb_group3_c64);
        num("VITS_DDS_LAT_MAX_BLOCKS", k.dds_lat_max_blocks);
        num("VITS_TILE128", k.tile128);
        num("VITS_MIN_BLOCKS", k.min_blocks);
        flag("VITS_NO_NARROW", k.no_narrow);
        num("VITS_NARROW_K1", k.narrow_k1);
        num("VITS_NBUF", k.nbuf);
        flag("VITS_NO_ONESHOT", k.no_oneshot);
        num("VITS_DB_MIN", k.db_min);
        num("VITS_T16_TILE0", k.t16_tile0);
        flag("VITS_NO_CONVT16S", k.no_convt16s);
        flag("VITS_CONVT16S_ALL", k.convt16s_all);
        flag("VITS_NO_CONVT16L", k.no_convt16l);
        flag("VITS_ATT_VALU", k.att_valu);
        flag("VITS_NO_ATT_LAT", k.no_att_lat);
        num("VITS_ATT_NW", k.att_nw);
        num("VITS_ATT_SHORT", k.att_short);
        num("VITS_LN_TW", k.ln_tw);
        if (const char* e = getenv("VITS_RBB_C128")) k.rbb_c128 = atoi(e) != 0;
        num("VITS_RBB_C64K11", k.rbb_c64k11);
        num("VITS_RBB_STREAM_TILES", k.rbb_stream_tiles);
        num("VITS_RBB_STREAM_MIN_BLOCKS", k.rbb_stream_min_blocks);
        num("VITS_FUSE16_MAXC", k.fuse16_maxc);
        if (const char* e = getenv("VITS_FUSE32_C128")) k.fuse32_c128 = atoi(e) != 0;
        num("VITS_WN16_NCW", k.wn16_ncw);
        num("VITS_FLOW_NCW", k.flow_ncw);
        num("VITS_FLOW_NARROW_MAX", k.flow_narrow_max);
        num("VITS_RB16_NARROW_MAX", k.rb16_narrow_max);
        num("VITS_CONVT16_SPLIT_MAX", k.convt16_split_max);
        num("VITS_CONVT16_R128", k.convt16_r128);
        flag("VITS_NO_LAT16", k.no_lat16);
        flag("VITS_NO_LAT16H", k.no_lat16h);
        flag("VITS_NO_LAT16H_PRE", k.no_lat16h_pre);
        flag("VITS_NO_LAT16H_GROUP", k.no_lat16h_group);
        num("VITS_LAT16H_GROUP_SHAPE", k.lat16h_group_shape);
        num("VITS_LAT16H_MAX_TILES", k.lat16h_max_tiles);
        num("VITS_LAT16H_MAX_TILES_C128", k.lat16h_max_tiles_c128);
        num("VITS_LAT16H_SHAPE", k.lat16h_shape);
        num("VITS_LAT16_MAX_WAVES", k.lat16_max_waves);
        return k;
    }
};
namespace detail {
inline thread_local const KernelKnobs* tl_kernel_knobs = nullptr;
}
inline const KernelKnobs& kernel_knobs() {
    if (detail::tl_kernel_knobs) return *detail::tl_kernel_knobs;
    static const KernelKnobs process_default = KernelKnobs::from_env();
    return process_default;
}
struct KernelKnobsScope {
    const KernelKnobs* prev;
    explicit KernelKnobsScope(const KernelKnobs* k) : prev(detail::tl_kernel_knobs) { detail::tl_kernel_knobs = k; }
    ~KernelKnobsScope() { detail::tl_kernel_knobs = prev; }
    KernelKnobsScope(const KernelKnobsScope&) = delete;
    KernelKnobsScope& operator=(const KernelKnobsScope&) = delete;
};

// hipFuncAttributeMaxDynamicSharedMemorySize is an attribute of a function ON A DEVICE. A kernel instantiation that needs more than
// 64 KB of LDS raises its limit once per device: the once-flag is one bit per device, so a process that loads models on several
// devices (vits_set_device) raises it on each of them (a process-wide flag left the second device at 64 KB: launch failures there).
struct BigLdsOnce {
    std::atomic<uint64_t> mask{0};
    static uint64_t bit() {
        int d = 0;
        (void)hipGetDevice(&d);
        return 1ull << (d & 63);
    }
    bool needed() const { return !(mask.load(std::memory_order_acquire) & bit()); }
    void done() { mask.fetch_or(bit(), std::memory_order_release); }
};

// Per-launch timing without extra queue packets. The engine's per-kernel profiler (bench.py's instrumented region) used to bracket
// every launch with hipEventRecord: a barrier packet each, ~300 of them per step, and durations that include the gap to the next packet.
// When the profiler has armed this thread's timer, the next kernel launch carries the two events on its OWN dispatch packet instead
// (hipExtLaunchKernel: start / end timestamps of the kernel's completion signal — the clock rocprofv3 reads), and only a span that
// launches several kernels falls back to a recorded stop event (Profiler::end). (Step time: the same, 76.8 against 76.9 ms.)
struct LaunchTimer {
    hipEvent_t start = nullptr, stop = nullptr;
    int launches = 0;
};
inline thread_local LaunchTimer vits_launch_timer;
#define VITS_KLAUNCH(kernel, grid, block, lds, stream, ...)                                                          \
    do {                                                                                                            \
        ::vits::LaunchTimer& lt_ = ::vits::vits_launch_timer;                                                       \
        if (lt_.start && lt_.launches++ == 0)                                                                       \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, lt_.start, lt_.stop, 0, __VA_ARGS__);           \
        else                                                                                                        \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                      \
    } while (0)

#ifdef __HIPCC__
// The WaveNet gate tanh(a) * sigmoid(s) (vits.cpp:442-450) of every gated-conv epilogue (conv_mfma.hip, conv16.hip, wavenet32.hip), libm's
// tanhf / expf. (A version on the hardware exp2 / reciprocal — one v_exp_f32 per factor, |error| <= 2e-7 — measured +-0 on a fused WaveNet
// layer: the gate is 1.5 of its 37 us, tools/wn16_micro.hip; the exact functions stay.)
__device__ __forceinline__ float wavenet_gate(float a, float s) { return tanhf(a) * (1.0f / (1.0f + expf(-s))); }
#endif

struct TensorRef {
    float* p = nullptr;
    int64_t bs = 0;  // batch stride (floats)
    int32_t cs = 0;  // channel stride (floats)
};

// ---- MFMA convolution ---------------------------------------------------------------------------------
enum ConvEpilogue : int { EPI_STD = 0, EPI_GATE = 1, EPI_CONVT = 2 };
enum ConvTile : int {
    TILE_128x128 = 0,
    TILE_64x256 = 1,
    TILE_32x256 = 2,
    TILE_64x64 = 3,
    TILE_32x64 = 4,
    TILE_NARROW = 5,  // 128 rows x 32 columns (256 x 32 for the gated conv): tiny grids only, see choose in launch_conv
    TILE_LAT16 = 6    // 64 rows x 16 columns on v_mfma_f32_16x16x4_f32 (conv_lat16_kernel): tiny grids with long K chains
};

// Weights pre-packed at load time in exact MFMA A-fragment order (see conv_mfma.hip), resident in HBM.
struct PackedConv {
    float* wp = nullptr;    // packed weights
    float* bias = nullptr;  // [cout] (original channel order) or nullptr
    int cin = 0, cout = 0, kt = 0;
    int rows = 0;      // GEMM M (cout, or cout*stride for the transposed conv)
    int mtiles = 0;       // number of 32-row tiles in the packed array (padded to a multiple of 4)
    int mtiles_used = 0;  // tiles that hold real rows
    int nchunks = 0;   // ceil(cin / 32)
    int epi = EPI_STD;
    int ct_stride = 0;  // EPI_CONVT: upsampling stride s
    int64_t bytes = 0;
    uint16_t* wp16 = nullptr;  // 16-bit A fragments of the VITS_ARITH_F16 / BF16 path (packed by Engine::set_arith)
    float* wp_l16 = nullptr;   // the same weights as v_mfma_f32_16x16x4_f32 A fragments (repack_conv_weights_l16) for conv_lat16_kernel, or nullptr
    int64_t bytes16 = 0;
    uint16_t* wps = nullptr;   // VITS_ARITH_F32_SPLIT: the weights as TWO bf16 planes of A fragments (conv_split.hip pack_conv_weights_split), or nullptr
    int64_t bytes_s = 0;
};

// 16-bit activation in "group layout" [batch][channel/8][time][8] (conv16.hip): one 16-byte slot = 8 channels of one time step
struct Ref16 {
    uint16_t* p = nullptr;
    int64_t bs = 0;  // batch stride (16-bit elements)
    int32_t ts = 0;  // slots per group row (time stride)
};

// a conv input of VITS_ARITH_F32_SPLIT (conv_split.hip): the three bf16 planes of an fp32 tensor in the group layout, [batch][plane 3][channel/8][time][8]
struct Split3Ref {
    uint16_t* p = nullptr;
    int64_t bs = 0;  // batch stride (16-bit elements)
    int64_t ps = 0;  // plane stride
    int32_t ts = 0;  // slots per group row (time stride)
};

struct ConvCall {
    TensorRef x, y, res, acc;  // res/acc optional (p == nullptr)
    const int* len_in = nullptr;   // per-utterance valid input length (nullptr: t_in)
    const int* len_out = nullptr;  // per-utterance valid output length (nullptr: t_out)
    int batch = 1;
    int t_in = 0, t_out = 0;  // maximum lengths (grid extent)
    // profiler accounting only: sum over the batch of the per-utterance valid input / output lengths (< 0: batch * t_in / t_out).
    // Blocks past an utterance's length exit at once, so work and traffic are counted over the real lengths, not the padded extent.
    int64_t sum_in = -1, sum_out = -1;
    int dil = 1, pad_l = 0;
    int pre_act = 0;  // 1: leaky_relu(slope) on load
    float slope = 0.f;
    int post_act = 0;  // 1: relu; 2: leaky_relu(post_slope) of the stored value (EPI_STD)
    float post_slope = 0.f;
    float* y2 = nullptr;  // EPI_STD, optional: also store leaky_relu(post_slope) of the output here (layout of y)
    float scale = 1.f;  // applied when acc.p: y = (acc + v) * scale, or / scale when scale_div
    int scale_div = 0;
    int ct_crop = 0;  // EPI_CONVT: output crop
    int tile = -1;    // ConvTile override (-1: chosen from the shape)
    // LayerNorm over the input channels applied ON LOAD (round 6; conv_lat16_kernel only — conv_ln_on_load_ok() tells, launch_conv refuses otherwise):
    // x is the UN-normalised tensor (add_layer_norm_kernel's input), every block normalises the columns of its input tile in that kernel's order of
    // operations, and ln_out (any layout) receives the normalised tensor — each column written once, by the blocks of the first row group.
    const float *ln_gamma = nullptr, *ln_beta = nullptr;
    float ln_eps = 0.f;
    TensorRef ln_out;
    // VITS_ARITH_F32_SPLIT (launch_conv_split only): the input as split planes (x is then unused), and — optional — the planes of leaky_relu(ys3_slope) of the
    // result for the next conv (y may then be null: a tensor whose only reader is the next conv is never stored in fp32)
    Split3Ref xs3, ys3;
    float ys3_slope = 1.f;
};
// VITS_ARITH_F32_SPLIT (conv_split.hip): fp32-accurate convs on the bf16 matrix cores by operand splitting — see the file's head comment
bool conv_split_candidate(int epi, int kt, int cin, int cout);  // the layers that get split weight planes at vits_model_set_arith
bool pack_conv_weights_split(const float* w, int cout, int cin, int k, std::vector<uint16_t>& out);  // w: torch layout [cout][cin][k]; false = not exactly two bf16 pieces
bool conv_split_supported(const PackedConv& w, int dil);
hipError_t launch_conv_split(const PackedConv& w, const ConvCall& c, hipStream_t s);
hipError_t launch_split_planes(TensorRef x, int channels, const int* lens, int batch, int tmax, float slope, Split3Ref out, hipStream_t s);
bool conv_ln_on_load_ok(const PackedConv& w, const ConvCall& c);

// host-side packing: w is torch layout [cout][cin][k] (EPI_STD / EPI_GATE) or [cin][cout][k] (EPI_CONVT)
std::vector<float> pack_conv_weights(const float* w, int cout, int cin, int k, int epi, int ct_stride, int* rows, int* mtiles_used, int* mtiles,
                                     int* nchunks);
// conv_lat16_kernel's A fragments from the packed array: [32-row tile][16-row half][quad = 16 input channels of one tap][lane][4] with lane l = row l & 15,
// component s = channel 4 s + (l >> 4) of the quad: one 16-byte load per lane feeds four consecutive MFMAs. Same size as `packed`.
bool conv_lat16_candidate(int epi, int kt, int cin);  // the layers that get a second copy of their fp32 weights for the latency kernel: every STD conv and the 5-tap GATE convs with >= 64 products per output
std::vector<float> repack_conv_weights_l16(const std::vector<float>& packed, int mtiles, int nchunks, int kt);
int choose_conv_tile(int rows, int epi, int t_hint);
int resolve_conv_tile(const PackedConv& w, const ConvCall& c);  // the tile launch_conv will use (small-grid rules included)
hipError_t launch_conv(const PackedConv& w, const ConvCall& c, hipStream_t s);
double conv_flops(const PackedConv& w, const ConvCall& c, int64_t total_cols);
// The same-position convolutions of up to three ResBlocks (11 / 7 / 3 taps, one dilation of {1, 3, 5}, one channel count that is a
// multiple of 128) as ONE launch on the 128 x 128 tile; results are bit-identical to n launch_conv calls.
bool conv_group_supported(const PackedConv& w, int dil);
hipError_t launch_conv_group(const PackedConv* const* w, const ConvCall* c, int n, hipStream_t s);

// ---- 16-bit-operand convolution (conv16.hip): v_mfma_f32_32x32x16_{f16,bf16}, fp32 accumulate --------------------
struct Conv16Call {
    Ref16 x;  // input, group layout (already activated: the writer / converter applies the leaky_relu)
    const int* len_in = nullptr;
    const int* len_out = nullptr;
    int batch = 1, t_in = 0, t_out = 0, dil = 1, pad_l = 0;
    int post_act = 0;  // standard outputs: 1 relu, 2 leaky_relu(post_slope) of the stored value; group outputs: 1 relu
    float post_slope = 0.f;
    float scale = 1.f;
    int scale_div = 0;
    int ct_crop = 0;
    int tile = -1;
    // standard-layout fp32 outputs / inputs ([b][c][t]): used when neither yg nor y16 is set
    TensorRef y, res, acc;
    float* y2 = nullptr;
    // group-layout outputs: fp32 residual stream [b][c/8][g_ts][8] (yg / resg / accg share strides) and the 16-bit copy
    float* yg = nullptr;
    const float* resg = nullptr;
    const float* accg = nullptr;
    int64_t g_bs = 0;
    int g_ts = 0;
    Ref16 y16;
    float y16_slope = 1.f;  // leaky_relu fused into the 16-bit copy (1 = none)
    int64_t sum_in = -1, sum_out = -1;  // profiler accounting (see ConvCall)
};
// One HiFiGAN ResBlock conv pair as a single kernel (rbpair16.hip): y' = y + conv2(leaky_relu(conv1(x) + b1)) + b2 with x =
// the 16-bit leaky_relu(y); the intermediate stays in LDS. C = 32 / 64, k = 3 / 7 / 11, conv1 dilation 1 / 3 / 5.
struct RbPair16Call {
    Ref16 x;
    const int* lens = nullptr;
    int batch = 1, tmax = 0, dil = 1;
    float slope = 0.1f;
    float* yg = nullptr;
    const float* resg = nullptr;
    const float* accg = nullptr;
    int64_t g_bs = 0;
    int g_ts = 0;
    Ref16 y16;
    float y16_slope = 1.f;
    float scale = 1.f;
    int scale_div = 0;
};
bool rbpair16_supported(int channels, int kt, int dil);
// One whole ResBlock (three pairs, dilations 1 / 3 / 5) as a single kernel (rbblock16.hip): the fp32 stream stays in registers across the
// pairs, HBM sees the stage input once and the resblock output once. C = 32 / 64, k = 3 / 7 / 11.
struct RbBlock16Call {
    const float* y0 = nullptr;  // stage input, fp32 group layout [b][C/8][g_ts][8]
    const int* lens = nullptr;
    int batch = 1, tmax = 0;
    float slope = 0.1f;
    float* yg = nullptr;  // output (same strides as y0)
    const float* accg = nullptr;
    int64_t g_bs = 0;
    int g_ts = 0;
    Ref16 y16;
    float y16_slope = 1.f;
    float scale = 1.f;
    int scale_div = 0;
};
bool rbblock16_supported(int channels, int kt, const int* dils, int ndil, int batch, int tmax);  // (batch x tmax: the launch's grid — C = 64, k = 11 only where it is cut into segments)
hipError_t launch_rbblock16(const PackedConv* const* c1, const PackedConv* const* c2, const RbBlock16Call& c, int arith, hipStream_t s);
// the three resblocks (k = 3, 7, 11) of a C = 32 stage as ONE launch (small grids, side-by-side resblocks): c1[m] / c2[m] = member m's three conv pairs
bool rbblock16_group3_supported(int channels, const int* kts, int batch, int tmax);
hipError_t launch_rbblock16_group3(const PackedConv* const (*c1)[3], const PackedConv* const (*c2)[3], const RbBlock16Call* c, int arith, hipStream_t s);
// ((y0 + y1) [+ y2]) * scale (or / scale) over fp32 group-layout tensors, in the order and with the expressions of the resblocks' chained accumulation; the fp32 sum
// (optional) and / or its 16-bit copy behind leaky_relu(y16_slope) — small grids: the resblocks of a stage then need not run one behind the other
hipError_t launch_rb_sum3(const float* y0, const float* y1, const float* y2, int channels, int64_t g_bs, int g_ts, const int* lens, int batch, int tmax, float scale, int scale_div,
                          float* yg, Ref16 y16, float y16_slope, int arith, hipStream_t s);
// fp32 ResBlock conv pair as one kernel (rbpair32.hip): y = x + conv2(leaky_relu(conv1(leaky_relu(x)) + b1)) + b2; y must not alias x
struct RbPair32Call {
    TensorRef x, y, acc;
    const int* lens = nullptr;
    int batch = 1, tmax = 0, dil = 1;
    float slope = 0.1f;
    float scale = 1.f;
    int scale_div = 0;
    int post_act = 0;
    float post_slope = 0.f;
};
bool rbpair32_supported(int channels, int kt, int dil);
// fp32: one WHOLE 3-tap ResBlock (dilations 1 / 3 / 5) of a narrow stage as one kernel (rbblock32.hip): the stream stays in registers across the three
// pairs; bit-identical to three launch_rbpair32 calls. y must not alias x (acc may alias y).
struct RbBlock32Call {
    TensorRef x, y, acc;
    const int* lens = nullptr;
    int batch = 1, tmax = 0;
    float slope = 0.1f;
    float scale = 1.f;
    int scale_div = 0;
    int post_act = 0;
    float post_slope = 0.f;
};
bool rbblock32_supported(int channels, int kt, const int* dils, int ndil);
hipError_t launch_rbblock32(const PackedConv* const* c1, const PackedConv* const* c2, const RbBlock32Call& c, hipStream_t s);
// one WaveNet layer of the flow (gated conv + 1x1 res/skip conv + the two adds) as one fp32 kernel (wavenet32.hip); h_out must not alias h
struct WaveNet32Call {
    TensorRef h, h_out, outputs;
    const int* lens = nullptr;
    int batch = 1, tmax = 0, hidden = 0, dil = 1;
};
bool wavenet32_supported(int hidden, int kt, int dil, const PackedConv& in, const PackedConv& rs);
hipError_t launch_wavenet32(const PackedConv& in, const PackedConv& rs, const WaveNet32Call& c, hipStream_t s);
bool wavenet16_supported(int hidden, int kt, int dil, const PackedConv& in, const PackedConv& rs);  // the same layer on 16-bit operands
hipError_t launch_wavenet16(const PackedConv& in, const PackedConv& rs, const WaveNet32Call& c, int arith, hipStream_t s);
// one whole coupling layer of the flow (pre conv, four WaveNet layers, post conv, x1 += ...) as one kernel, 16-bit-operand modes (wavenet32.hip)
struct FlowCouple16Call {
    TensorRef x0, x1;  // conditioning half (read), updated half (in place): fp32 [b][F/2][t]
    const int* lens = nullptr;
    int batch = 1, tmax = 0, hidden = 0, half = 0;
};
bool flow_couple16_supported(int hidden, int half, int kt, int rate, int layers, const PackedConv& pre, const PackedConv* in, const PackedConv* rs, const PackedConv& post);
hipError_t launch_flow_couple16(const PackedConv& pre, const PackedConv* in, const PackedConv* rs, const PackedConv& post, const FlowCouple16Call& c, int arith,
                                hipStream_t s);
hipError_t launch_rbpair32(const PackedConv& c1, const PackedConv& c2, const RbPair32Call& c, hipStream_t s);
hipError_t launch_rbpair16(const PackedConv& c1, const PackedConv& c2, const RbPair16Call& c, int arith, hipStream_t s);
std::vector<uint16_t> pack_conv_weights16(const float* w, int cout, int cin, int k, int epi, int ct_stride, int arith);
int choose_conv16_tile(int rows, int epi, int ncols_max, int mtiles_used, int batch);
// conv16_lat.hip: the wide stages' group-layout resblock convs on small grids (batch 1 ... 4); launch_conv16 routes to it (profile tile tag T7)
bool conv16_lat_shape_ok(int channels, int kt, int dil, int batch, int tmax);
bool conv16_lat_wanted(const PackedConv& w, const Conv16Call& c);
hipError_t launch_conv16_lat(const PackedConv& w, const Conv16Call& c, int arith, hipStream_t s);
bool conv16_lat_group_wanted(const PackedConv* const* w, const Conv16Call* c);  // the same-position convs of a stage's three resblocks (k = 3, 7, 11) as ONE launch
hipError_t launch_conv16_lat_group(const PackedConv* const* w, const Conv16Call* c, int arith, hipStream_t s);
bool conv16_lat_pre_wanted(const PackedConv& w, int batch, int tmax);  // the vocoder's conv_pre on a small grid, straight from the fp32 flow output (no converter launch)
hipError_t launch_conv16_lat_pre(const PackedConv& w, TensorRef x, const int* lens, int batch, int tmax, Ref16 y16, float y16_slope, int arith, hipStream_t s);
hipError_t launch_conv16(const PackedConv& w, const Conv16Call& c, int arith, hipStream_t s);
// ConvTranspose1d (kernel = 2 x stride) in the group layout as a streaming kernel (convt16.hip): all stride x c_out rows of a tile of input
// positions per block; bit-identical to launch_conv16 on the same call
bool convt16_stream_supported(const PackedConv& w);
hipError_t launch_convt16_stream(const PackedConv& w, const Conv16Call& c, int arith, hipStream_t s);
// tile tag of the instantiation launch_convt16_stream picks for `w` ("SL128" = convt16_lines_kernel<128>, "S4.1.16" = convt16_kernel<4, 1, 16>):
// the profiler label carries it, so that a profiler entry lines up with exactly one rocprofv3 kernel name (tools/pmc_common.py)
void convt16_stream_tag(const PackedConv& w, char* buf, size_t cap);
hipError_t launch_to_group16(TensorRef x, const int* lens, int batch, int channels, int tmax, float slope, Ref16 y, int arith, hipStream_t s);
hipError_t launch_conv_post16(Ref16 x, const float* w, int cin, int k, TensorRef pre, TensorRef wave, const int* lens, int batch, int tmax, int arith, hipStream_t s,
                              int emit_lo = 0, const int* emit_hi = nullptr);

// ---- small kernels ------------------------------------------------------------------------------------
hipError_t launch_embed(const int* ids, int id_stride, const int* lens, const float* table, int hidden, float scale, TensorRef x, int batch, int tmax,
                        hipStream_t s);
hipError_t launch_scale_rows(TensorRef x, int channels, float scale, int batch, int tmax, hipStream_t s);
// EMULATED ggml lookup tables (INFERRED from upstream ggerganov/ggml of the reference's era, the maxilevi/ggml fork is
// absent): device copies of the two 65536-entry fp16 tables ggml builds at init — gelu[i] = fp16(tanh-GELU(fp16 value i)), exp[i] =
// fp16(expf(fp16 value i)) — built on the host with the C library (Engine::set_ggml_tables). Kernels that receive a non-null pointer route
// ggml_gelu (vits.cpp:673,687) / the exponential of ggml_soft_max (vits.cpp:329,719,735) through it; null = erf-GELU / fp32 soft-max.
struct GgmlTables {
    const uint16_t* gelu = nullptr;
    const uint16_t* exp = nullptr;
};
hipError_t launch_rel_attention(TensorRef q, TensorRef k, TensorRef v, const float* rel_k, const float* rel_v, TensorRef out, const int* lens, int batch,
                                int heads, int head_dim, int tmax, int window, float q_scale, hipStream_t s, GgmlTables tabs = GgmlTables());
hipError_t launch_add_layer_norm(TensorRef x, TensorRef res, const float* gamma, const float* beta, TensorRef y, const int* lens, int batch, int channels,
                                 int tmax, float eps, int post_gelu, TensorRef add_to, hipStream_t s, GgmlTables tabs = GgmlTables());
hipError_t launch_dds_depthwise(TensorRef x, TensorRef g, const float* w, const float* bias, const float* gamma, const float* beta, TensorRef y,
                                const int* lens, int batch, int channels, int tmax, int k, int dil, float eps, hipStream_t s, int arith = 0,
                                GgmlTables tabs = GgmlTables());
// one DDS layer (depthwise + LN + gelu + 1x1 conv + LN + gelu + residual) as one kernel; y must not alias x
bool dds_layer_supported(const PackedConv& pw, int channels, int k, int dil, int arith);
hipError_t launch_dds_layer(TensorRef x, TensorRef y, const float* dw_w, const float* dw_b, const float* g1, const float* b1, const PackedConv& pw, const float* g2,
                            const float* b2, const int* lens, int batch, int channels, int tmax, int k, int dil, float eps, int arith, hipStream_t s,
                            GgmlTables tabs = GgmlTables());
// The same layer for latency-bound launches (stage1_lat.hip: 16-token blocks, element-parallel phases, 16x16x4 MFMA tiles; bit-identical), optionally
// with the per-token op in front of the DDS block fused in (head_w: the conv flow's 1 -> H conv of latent row zc + conditioning, vits.cpp:864,651-653;
// head_conv: an H -> H 1x1 conv of x, vits.cpp:939) and the 1x1 conv behind it (tail_conv -> y2; y is then not written). fp32 arithmetic only.
struct DdsLatCall {
    TensorRef x, y;
    const float *dw_w = nullptr, *dw_b = nullptr, *g1 = nullptr, *b1 = nullptr, *g2 = nullptr, *b2 = nullptr;
    const PackedConv* pw = nullptr;
    const float *head_w = nullptr, *head_b = nullptr;
    TensorRef z, cond;
    int zc = 0;
    const PackedConv* head_conv = nullptr;
    const PackedConv* tail_conv = nullptr;
    TensorRef y2;
    const int* lens = nullptr;
    int batch = 1, channels = 0, tmax = 0, k = 3, dil = 1;
    float eps = 1e-5f;
    GgmlTables tabs;
};
bool dds_layer_lat_supported(const PackedConv& pw, int channels, int k, int dil);
hipError_t launch_dds_layer_lat(const DdsLatCall& c, hipStream_t s);
hipError_t launch_pointwise_from1(TensorRef z, int zc, const float* w, const float* bias, TensorRef cond, TensorRef y, const int* lens, int batch,
                                  int channels, int tmax, hipStream_t s, int arith = 0);
hipError_t launch_spline(TensorRef u, TensorRef z, int zc, const int* lens, int batch, int tmax, int bins, float tail, float inv_sqrt, int mode,
                         hipStream_t s, GgmlTables tabs = GgmlTables());
hipError_t launch_affine(TensorRef z, int c_first, const float* translate, const float* log_scale, int sign, const int* lens, int batch, int tmax,
                         hipStream_t s);
// ---- exact-order stage one of the emulated-ggml mode (exact_stage1.hip; element functions: include/vits_exact_math.h) -----------------------
hipError_t launch_exact_conv(TensorRef x, const float* w, const float* bias, TensorRef y, TensorRef res, const int* lens, int batch, int cin, int cout, int K, int dil, int pad_l,
                             int tmax, bool relu, const float* post_scale, hipStream_t s);
hipError_t launch_exact_depthwise(TensorRef x, const float* w, const float* bias, TensorRef y, const int* lens, int batch, int channels, int K, int dil, int pad, int tmax,
                                  hipStream_t s);
hipError_t launch_exact_add(TensorRef x, TensorRef g, const int* lens, int batch, int channels, int tmax, hipStream_t s);
hipError_t launch_exact_layer_norm(TensorRef x, const float* gamma, const float* beta, const int* lens, int batch, int channels, int tmax, float eps, const uint16_t* gelu_tab,
                                   hipStream_t s);
hipError_t launch_exact_attention(TensorRef q, TensorRef k, TensorRef v, const float* ek, const float* ev, TensorRef out, float* scratch, int srow, const int* lens, int batch,
                                  int heads, int hd, int tmax, int window, const uint16_t* exp_tab, hipStream_t s);
hipError_t launch_exact_affine(TensorRef z, int c_first, float t0, float t1, float e0, float e1, const int* lens, int batch, int tmax, hipStream_t s);
hipError_t launch_exact_spline(TensorRef z, int row, TensorRef u, float* res, float* inside, float* tmp, int stride, const int* lens, int batch, int tmax, int nb, float B,
                               float inv_sqrt, float constant, bool refmode, const uint16_t* exp_tab, hipStream_t s);
hipError_t launch_noise_dur(TensorRef z, const int* lens, int batch, int tmax, uint64_t seed, const int* seed_off, float scale, hipStream_t s);
hipError_t launch_durations(TensorRef logw, int c, const int* lens, int batch, int tmax, float length_scale, int fixed, float* dur, int* cum, int* frames,
                            int* stage_lens, int n_stage, const int* stage_mul, const int* stage_add, hipStream_t s, bool exact = false);
hipError_t launch_zp(TensorRef mean, TensorRef logvar, const int* cum, int cum_stride, const int* tok_lens, const int* frames, TensorRef noise, int noise_kind,
                     uint64_t seed, const int* seed_off, float noise_scale, TensorRef zp, int batch, int channels, int lmax, hipStream_t s);
hipError_t launch_fill(float* p, size_t n, float v, hipStream_t s);
hipError_t launch_rb_sum3_std(TensorRef y0, TensorRef y1, TensorRef y2, TensorRef out, int channels, const int* lens, int batch, int tmax, float scale, int scale_div, int post_act,
                              float post_slope, hipStream_t s);  // fp32 [b][c][t]: ((y0 + y1) [+ y2]) scaled [+ leaky_relu]: side-by-side resblocks of the fp32 path (small grids)
hipError_t launch_fill_rows(TensorRef x, int channels, float v, int batch, int tmax, hipStream_t s);
// fp32 -> int16 PCM rows on the device (test/main.cpp:31-33); lens (device, optional) limits each row
hipError_t launch_pcm16(const float* src, int64_t src_stride, int16_t* dst, int64_t dst_stride, const int64_t* lens, int rows, int64_t cols, hipStream_t s);
hipError_t launch_conv_post(TensorRef x, const float* w, int cin, int k, float slope, TensorRef pre_tanh, TensorRef wave, const int* lens, int batch,
                            int tmax, hipStream_t s, int emit_lo = 0, const int* emit_hi = nullptr, int arith = 0);

}  // namespace vits
