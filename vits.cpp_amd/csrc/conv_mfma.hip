// conv_mfma.hip — the hot kernel: Conv1d / ConvTranspose1d as an implicit GEMM on the gfx950 matrix cores,
// exact fp32 (v_mfma_f32_32x32x2_f32: bit-for-bit an fmaf chain, at the 157 TFLOP/s fp32 rate).
//
// Replaces the reference's conv1d = ggml_im2col_1d + ggml_mul_mat (+ separate bias / leaky_relu / add nodes):
//   /root/reference/src/include/custom-ops.h:680-694 (conv1d_impl), :397-431 (bias), :894-896 (leaky_relu),
//   :696-727 (add), called from src/vits.cpp:137-142,171-176 (conv1d[_with_bias]), :545-581 (HiFiGAN ResBlock),
//   :452-498 (WaveNet), :384-403 (encoder FFN), :287-289,358 (Linear), :178-193 (conv_transpose_1d_with_bias).
// Nothing is unfolded in memory: the input tile (with its (K-1)*dilation halo) is staged ONCE in LDS and every
// tap reads it at a shifted column, so each activation is fetched from HBM once per conv instead of K times.
//
// GEMM view:  Y[M = out channel][N = time] = sum_{kappa = (ci, tap)} A[M][kappa] * B[kappa][N]
//   A = weights, pre-packed at model load in exact MFMA A-fragment order -> one coalesced 16 B/lane load feeds
//       4 consecutive MFMAs, straight from L2 (weights of one conv are <= 2.9 MB, shared by every block);
//   B = x[ci][t + tap*dil - pad] read from the LDS tile with ds_read_b32 (lanes 0-31 / 32-63 each read 32
//       consecutive floats: conflict-free for any dilation).
// v_mfma_f32_32x32x2_f32 operand map (cdna guide §3): A lane l -> A[row l&31][k l>>5]; B lane l -> B[k l>>5][col l&31];
// C/D reg r of lane l -> row (r&3) + 8*(r>>2) + 4*(l>>5), col l&31.
//
// Fused around the GEMM (all separate graph nodes = full DRAM passes in the reference):
//   load side : ragged-length masking (zero padding), leaky_relu(slope)               (vits.cpp:554,567,613)
//   store side: + bias, relu (FFN, :397) | tanh*sigmoid gate (:442-450) | residual add (:578) |
//               resblock accumulation and the 1/num_kernels scale (:630,635) | transposed-conv phase scatter.
#include <hip/hip_runtime.h>
#include <type_traits>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "kernels.h"

// The kernel template is instantiated for ~40 (taps, dilation, tile, epilogue) combinations; one translation unit takes
// minutes, so the Makefile compiles this file five times: VITS_CONV_PART 0 = host code + taps 1/2/5, 1 = taps 3, 2 = taps 7, 4 = grouped launch,
// 3 = taps 11 (each part defines launch_conv_k<taps>() for the dispatcher in part 0).
#ifndef VITS_CONV_PART
#define VITS_CONV_PART 0
#endif

namespace vits {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int CK = 32;  // input channels per LDS chunk

struct ConvParams {
    const float* x;
    int64_t x_bs;
    int x_cs;
    float* y;
    int64_t y_bs;
    int y_cs;
    const float* res;
    int64_t r_bs;
    int r_cs;
    const float* acc;
    int64_t a_bs;
    int a_cs;
    const float* wp;
    const float* bias;
    const int* len_in;
    const int* len_out;
    int t_in, t_out;
    int cin, cout, rows, nchunks;
    int mtiles;  // 32-row tiles in the packed weight array (a block may own row tiles past it: their fragments are read from the last one)
    int dil, pad_l, xw, lds_off;
    int nbuf;  // LDS input buffers of the wave-specialised path: 2, or 3 when a chunk of MFMA work is shorter than the DMA latency
    int oneshot;  // small tiles on latency-bound launches: one buffer per chunk, filled by ALL five waves at once (see the kernel)
    int pre_act;
    float slope;
    int post_act;
    float post_slope;
    float* y2;  // optional second output: leaky_relu(post_slope) of what goes to y (same strides as y)
    float scale;
    int scale_div;
    int ct_stride, ct_crop;
    int kt_rt;  // taps (the latency kernel below takes them at run time; every other kernel has them as a template parameter)
    const float* wl16;  // conv_lat16_kernel's A fragments
    int l16_fill4;      // conv_lat16_kernel: rows 16-byte aligned -> float4 fill
    // conv_lat16_kernel: LayerNorm over the input channels on load (ConvCall::ln_gamma)
    const float *ln_gamma, *ln_beta;
    float ln_eps;
    float* ln_out;
    int64_t lo_bs;
    int lo_cs;
};

// DIL > 0 (or < 0 for the transposed conv): compile-time dilation -> the LDS row stride and every tap offset are
// immediates of the ds_read instructions (one base VGPR instead of one per tap). DIL == 0: run-time dilation (generic).
// DB: wave-specialised, multi-buffered (every compile-time-dilation conv). The block has FIVE waves:
//     waves 0-3 only run MFMAs (their only global loads are the L2-resident weight fragments, through a buffer
//     descriptor), wave 4 is a PRODUCER that streams the next 32-channel input tile from HBM straight into another LDS
//     buffer with LDS-DMA (buffer_load_dwordx4 ... lds: no data VGPRs, no ds_write pass). Reason: vmcnt retires in order, so
//     a compute wave that issues HBM tile loads itself makes every later weight load wait a full HBM latency (measured:
//     k=3 72 -> 103 TFLOP/s without those stalls). The K loop of the compute waves has no vector-ALU instruction at all
//     (VALU and MFMA share the issue port); LeakyReLU is normally applied by whoever WROTE the input, otherwise by the
//     producer in LDS (k >= 5) or at the B-operand read (k <= 3); zero padding at the sequence ends is a fix-up by the
//     producer on boundary tiles only. DESIGN.md section 4.1 has the measurements behind each of these choices.
// Developer instrumentation (per-block phase stamps, VITS_PHASE_TIMING) and the compiled-out ablation hooks of tools/conv_micro.hip (VAR_*: what a variant of the
// kernel WITHOUT some part of it costs — DESIGN.md 4.1 quotes them) live in conv_mfma_ablations.inc; every hook below (VITS_STAMP, VITS_ABL_*) expands to nothing in
// the product build.
#include "conv_mfma_ablations.inc"

#ifndef VITS_WAVES_ATTR
#define VITS_WAVES_ATTR
#endif
// Zero the columns of a landed 32-row LDS tile (row pitch XWP, column 0 = global time ts) that lie outside [0, len): the left
// columns [0, -ts) and the right columns [len - ts, XWP). A lane owns a row (two lanes per row) and walks the few columns concerned:
// a boundary tile costs the producer ~(columns outside) / 2 stores. (A predicated store per (row, 64-column piece) — 96 branches for
// mostly one or two active lanes each — took 7.5-14k cycles per chunk: at 128 tokens per utterance EVERY tile is a boundary tile, and
// the compute waves of the encoder's FFN convs waited for that loop half of the time.)
// Each lane starts its walk at its own column (row r begins at outside-column r mod noob): ds_write_b32 is served in two groups of 32
// lanes on banks (address / 4) mod 32, and XWP is a multiple of 16, so 32 rows at ONE column sit on two banks — a 16-way conflict per
// store, which was the whole LDS_BANK_CONFLICT / LDS_IDX_ACTIVE = 0.30-0.37 of the stage-one tiles (round-3 PMC). Rotated, the rows of a
// store hit min(noob, 16) different banks of their half.
template <int XWP>
__device__ __forceinline__ void zero_oob_columns(float* lbase, int ts, int len, int lane) {
    int nl = -ts;
    nl = nl < 0 ? 0 : (nl > XWP ? XWP : nl);
    int rs = len - ts;
    rs = rs < nl ? nl : (rs > XWP ? XWP : rs);
    const int noob = nl + XWP - rs;
    if (noob <= 0) return;
    float* row = lbase + (lane & 31) * XWP;
    int k = VITS_ABL_ZERO_OOB_START((lane & 31) % noob);  // (loop-invariant over the chunks of a block: computed once)
    for (int i = lane >> 5; i < noob; i += 2) {
        int kk = k + i;
        kk = kk >= noob ? kk - noob : kk;
        row[kk < nl ? kk : rs + (kk - nl)] = 0.f;
    }
}

// The kernel body as a device function of the block's (column tile, row-tile group, utterance) coordinates and the block's dynamic
// LDS: conv_mfma_kernel runs it for one convolution; conv_group_kernel (below) runs the bodies of up to three convolutions with
// different tap counts in ONE launch.
template <int KT, int DIL, bool DB, int WM, int WN, int MR, int NR, int EPI>
__device__ __forceinline__ void conv_mfma_body(const ConvParams& p, float* xs, const int bx, const int by, const int bz) {
    constexpr int BN = WN * NR * 32;
    constexpr int SPAN_C = (KT - 1) * (DIL < 0 ? -DIL : DIL);
    constexpr int STEPS = KT * (CK / 8);  // float4 A-fragments (4 MFMA k-steps each) per chunk and row tile
    // DB path: where the fused leaky_relu runs. Short chunks (k <= 3: 12 steps) leave the producer wave no slack, so the
    // compute waves apply it at the B-operand read (max(x, slope*x), ~2 % there); for k >= 5 the producer rewrites the
    // landed tile in place instead (at-read costs 9 % on the k=11 kernels: two VALU ops between every ds_read and its MFMAs)
#ifndef VITS_LRELU_AT_READ_MAXK
#define VITS_LRELU_AT_READ_MAXK 3
#endif
    constexpr bool LRELU_AT_READ = KT <= VITS_LRELU_AT_READ_MAXK;
    // xs: [CK][xw] input tile(s)

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int b = bz;
    const int t0 = bx * BN;
    if constexpr (DB) VITS_ABL_SETPRIO(3);
    VITS_WSTART();
    const int len_in = p.len_in ? p.len_in[b] : p.t_in;
    // number of valid GEMM columns for this utterance
    int ncols;
    if (EPI == EPI_CONVT) ncols = len_in + 1;  // q in [0, L_in]: the last phase group only sees the m=1 tap
    else ncols = p.len_out ? p.len_out[b] : p.t_out;
    if (t0 >= ncols || len_in <= 0) return;  // (an utterance that has no frames in this vocoder window has length 0)
    VITS_STAMP(0);

    const int mt0 = (by * WM + wm) * MR;  // first 32-row tile of this wave
    // LDS row pitch granularity: 16 floats. (64 — whole dword-DMA pieces — fetched 192 columns for a 128 + 2..10 column
    // tile; 144 needs 18 instead of 24 DMA instructions per chunk and moves 25 % less through L2: 85.8 -> 84.1 ms per step.)
#ifndef VITS_XWP_GRAN
#define VITS_XWP_GRAN 16
#endif
    constexpr int XWP = (BN + SPAN_C + 3 + VITS_XWP_GRAN - 1) / VITS_XWP_GRAN * VITS_XWP_GRAN;  // DB: LDS row pitch (+3: 16-byte aligned origin)
    const int xw = DB ? XWP : (DIL != 0 ? BN + SPAN_C : p.xw);
    const int dil = DIL != 0 ? DIL : p.dil;
    const int lds_off = DIL != 0 ? (DIL < 0 ? SPAN_C : 0) : p.lds_off;
    const int tile_start = t0 - p.pad_l - lds_off;  // global time of LDS column 0
    const float* __restrict__ xb = p.x + (int64_t)b * p.x_bs;

    // DB path: 16-byte aligned rows let the producer stream with dwordx4 LDS-DMA into a tile whose column 0 is the
    // 4-aligned time below tile_start; the compute waves then read their B operands `shift` floats further right
    const bool x4 = DB && ((reinterpret_cast<uintptr_t>(xb) & 15) == 0) && ((p.x_cs & 3) == 0);
    const int shift = x4 ? (tile_start & 3) : 0;
    // Latency-bound launches (batch 1 / short inputs: at most ~2 blocks per CU, on the small tiles). Phase timing of the 1x1 convs
    // there: the compute waves wait ~2.4 us at EVERY chunk barrier for 0.5 us of MFMA work — one wave's LDS-DMA stream delivers a
    // 16 KB fill in about that time however many are queued. In this mode there is one LDS buffer per chunk, all five waves issue
    // the fills (chunk c by wave c mod 5) and post-process their own chunks, and the block meets at ONE barrier; the compute waves
    // then run the whole K range unsynchronised. Same accumulation order.
    constexpr bool COOP = DB && MR * NR <= 2;
    const bool oneshot = COOP && p.oneshot;
    if constexpr (DB) {
        if (wid == 4 || oneshot) {
            // ------------------------------- producer wave (every wave in the one-shot mode) ----------------------
            // barrier protocol (n = nchunks, both sides execute n+1 barriers):
            //   producer: fill(0); B0; for c: { fill(c+1) into buffer (c+1)&1; B(c+1) }
            //   compute : B0; for c: { MFMA on buffer c&1; B(c+1) }
            // buffer (c+1)&1 was last read during chunk c-1, which every compute wave finished before B(c).
            // LDS column 0 holds global time ts = tile_start rounded DOWN to a multiple of 4 (when the rows themselves are
            // 16-byte aligned): every DMA source address is then 16-byte aligned and one global_load_lds_dwordx4 moves
            // 1 KB per wave instruction instead of 256 B — the dword version needs 96 issues (~4 us) per 32 x 192 chunk,
            // which puts the producer on the critical path of the k = 3 kernels (5.7 us of MFMA work per chunk)
            const int ts = tile_start - shift;
            const bool interior = ts >= 0 && ts + XWP <= len_in;
            constexpr int NMP = (XWP + 63) / 64;
            // per-lane clamped time offsets of the NMP 64-column pieces of a row (same for every row and chunk)
            int tcl[NMP];
            bool oob[NMP];
#pragma unroll
            for (int m = 0; m < NMP; ++m) {
                const int t = ts + lane + 64 * m;
                tcl[m] = t < 0 ? 0 : (t < len_in ? t : len_in - 1);
                oob[m] = t != tcl[m];
            }
            // dwordx4 pattern: the tile is a linear array of float4; XW4 float4 per row, 64 per instruction -> the
            // (row, column) of a lane repeats every P4 instructions, which cover R4 whole rows
            constexpr int XW4 = XWP / 4;
            constexpr int G4 = XW4 % 64 == 0 ? 64 : XW4 % 32 == 0 ? 32 : XW4 % 16 == 0 ? 16 : XW4 % 8 == 0 ? 8 : 4;  // gcd(XW4, 64); XWP is a multiple of 16
            constexpr int P4 = XW4 / G4, R4 = 64 / G4;
            unsigned off4[P4];  // BYTE offset from the chunk's first row: 32-bit, so the DMA can use the SGPR-base + VGPR-offset form
#pragma unroll
            for (int j = 0; j < P4; ++j) {
                const int g = j * 64 + lane;
                const int r = g / XW4, c4 = g - r * XW4;
                int t = ts + 4 * c4;  // multiple of 4
                const int tlast = (len_in - 1) & ~3;  // last float4 that starts inside the sequence (rows are padded to x4)
                t = t < 0 ? 0 : (t > tlast ? tlast : t);
                off4[j] = (unsigned)(r * p.x_cs + t) * 4u;
            }
            const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, 0x7fffffff, 0x00020000);
            constexpr int NI4 = CK / R4 * P4;  // DMA instructions of one dwordx4 fill
            auto issue = [&](int c, int buf) __attribute__((always_inline)) {
                float* lbase = xs + buf * (CK * XWP);
                // LDS-DMA of the whole 32 x XWP tile from clamped (always valid) addresses ...
                if (x4 && (c + 1) * CK <= p.cin) {
                    // buffer form: descriptor (SGPRs) + per-lane byte offset (loop-invariant VGPR) + scalar row offset — no
                    // vector ALU work per DMA (the global_load_lds form needs a 64-bit VALU add for every address)
                    unsigned soff = (unsigned)(c * CK) * (unsigned)p.x_cs * 4u;
#pragma unroll 2
                    for (int i = 0; i < CK / R4; ++i) {
#pragma unroll
                        for (int j = 0; j < P4; ++j)
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(lbase + (i * P4 + j) * 256), 16, (int)off4[j],
                                                                 (int)soff, 0, 0);
                        soff += (unsigned)R4 * (unsigned)p.x_cs * 4u;
                    }
                } else {
#pragma unroll 4
                    for (int r = 0; r < CK; ++r) {
                        const int ch = c * CK + r;
                        const float* src = xb + (int64_t)(ch < p.cin ? ch : p.cin - 1) * p.x_cs;
#pragma unroll
                        for (int m = 0; m < NMP; ++m)
                            if (XWP % 64 == 0 || 64 * m + lane < XWP)
                                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + tcl[m]),
                                                                 (__attribute__((address_space(3))) void*)(lbase + r * XWP + 64 * m), 4, 0, 0);
                    }
                }
            };
            auto finish = [&](int c, int buf) __attribute__((always_inline)) {
                float* lbase = xs + buf * (CK * XWP);
                // ... leaky_relu in place (the producer has slack; doing it at the B-operand read of the compute waves
                // instead puts two VALU ops between every ds_read and its MFMAs: measured -9 % on the k=11 kernels) ...
                if (!LRELU_AT_READ && p.pre_act) {
                    // whole buffer as a linear array of float4 (XWP is a multiple of 4): 16-byte LDS reads / writes
                    float4* l4 = reinterpret_cast<float4*>(lbase);
                    constexpr int N4 = CK * XWP / 4;
#pragma unroll 4
                    for (int i = lane; i < N4; i += 64) {
                        float4 v = l4[i];
                        v.x = fmaxf(v.x, v.x * p.slope);
                        v.y = fmaxf(v.y, v.y * p.slope);
                        v.z = fmaxf(v.z, v.z * p.slope);
                        v.w = fmaxf(v.w, v.w * p.slope);
                        l4[i] = v;
                    }
                }
                // ... then zero what lies outside the sequence (boundary tiles) or beyond the last input channel
                if ((c + 1) * CK > p.cin) {
#pragma unroll 4
                    for (int r = 0; r < CK; ++r) {
                        const bool chbad = c * CK + r >= p.cin;
#pragma unroll
                        for (int m = 0; m < NMP; ++m)
                            if ((oob[m] || chbad) && (XWP % 64 == 0 || 64 * m + lane < XWP)) lbase[r * XWP + 64 * m + lane] = 0.f;
                    }
                } else if (!interior) {
                    zero_oob_columns<XWP>(lbase, ts, len_in, lane);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // LDS writes of this wave done before the barrier
            };
            // (no barrier after the LAST chunk: nothing reuses its buffer. The producer therefore retires a whole chunk
            // before the block does, and with it the fifth wave that keeps a second block from being placed on this CU —
            // the next block's launch and prologue overlap this block's last chunk instead of following its K loop)
            if (oneshot) {
                for (int c = wid; c < p.nchunks; c += 5) issue(c, c);
                if (wid == 4) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    for (int c = wid; c < p.nchunks; c += 5) finish(c, c);
                    __syncthreads();
                }
                // (compute waves: bias and weight-fragment prologue first, so that its memory latency overlaps the DMA; then the
                // same wait + post-processing and the barrier — "one-shot, compute side" below)
            } else if (p.nbuf == 2) {
                issue(0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                finish(0, 0);
                __syncthreads();
                for (int c = 0; c + 1 < p.nchunks; ++c) {
                    issue(c + 1, (c + 1) & 1);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    finish(c + 1, (c + 1) & 1);
                    __syncthreads();
                }
            } else {
                // three buffers: the DMA runs TWO chunks ahead. Used when a chunk is less MFMA work than one DMA round trip
                // (1x1 convs, k = 3 on narrow tiles: 0.4-2.7 us against ~2.5 us), where one chunk of look-ahead leaves the
                // compute waves waiting for every chunk. vmcnt retires in order: waiting until at most one fill's worth of
                // instructions is outstanding means the OLDER fill has landed (a dword fill has more instructions than NI4,
                // so the wait is then merely conservative).
                const int n = p.nchunks;
                issue(0, 0);
                if (n > 1) issue(1, 1);
                if (n > 1 && x4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI4) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                finish(0, 0);
                __syncthreads();
                int b1 = 1, b2 = 2;  // buffers of chunks c + 1 and c + 2
                VITS_PSTAMP_BEGIN();
                for (int c = 0; c + 1 < n; ++c) {
                    const bool more = c + 2 < n;
                    if (more) issue(c + 2, b2);  // last read during chunk c - 1, which every compute wave left before B(c)
                    PSTAMP(0);
                    if (more && x4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI4) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    PSTAMP(1);
                    finish(c + 1, b1);
                    PSTAMP(2);
                    __syncthreads();
                    PSTAMP(3);
                    b1 = b2;
                    b2 = b2 == 2 ? 0 : b2 + 1;
                }
                VITS_PSTAMP_END();
            }
            if (wid == 4) {
                VITS_WSTAMP();
                return;
            }
        }
    }

    floatx16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Per-lane bias values of the wide (dwordx4) epilogues, fetched NOW so that their latency hides behind the K loop.
    // Loaded inside the epilogue each one costs an s_waitcnt vmcnt(0), which on gfx9 also waits for every store issued
    // before it: measured 12 us of a 16 us epilogue on the 128x128 tiles.
    float bias_w[MR][4];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            int idx;
            if (EPI == EPI_STD) idx = (mt0 + mr) * 32 + 8 * g + (lane & 3) + 4 * (lane >> 5);
            else if (EPI == EPI_GATE) idx = (mt0 / 2) * 32 + 8 * g + (lane & 3) + 4 * (lane >> 5) + (mr == 0 ? 0 : p.cout / 2);
            else idx = (mt0 + mr) * 4 + g;  // stride-8 transposed conv: 8 phases = 8 GEMM rows per output channel
            bias_w[mr][g] = p.bias ? p.bias[idx < p.cout ? idx : p.cout - 1] : 0.f;
        }
    // gated conv on MR = 1 tiles (the 128 x 32 tile): the tanh rows and the sigmoid rows of a channel group are two WAVES (even / odd
    // row tile); the even wave also needs the sigmoid rows' bias
    float bias_s[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI == EPI_GATE && MR == 1) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int idx = (mt0 / 2) * 32 + 8 * g + (lane & 3) + 4 * (lane >> 5) + p.cout / 2;
            bias_s[g] = p.bias ? p.bias[idx < p.cout ? idx : p.cout - 1] : 0.f;
        }
    }

    const float slope_eff = p.pre_act ? p.slope : 1.0f;  // leaky_relu(x) = max(x, slope*x); slope 1 = identity
    const int krow = lane >> 5;
    const float* xrow0 = xs + krow * xw + wn * (NR * 32) + (lane & 31) + lds_off + shift;  // B operand base of this lane
    const size_t tile4 = (size_t)p.nchunks * STEPS * 64;                           // float4 per 32-row tile
    // A fragments come through a buffer descriptor: the address is (SGPR descriptor) + (per-lane byte offset, loop
    // invariant VGPR) + (scalar step offset), so the K loop needs NO vector ALU work for addressing. VALU and MFMA share
    // the issue port: every v_add / v_lshl_add_u64 in the loop is MFMA time (measured ~8 % of the k = 11 K loop).
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wp), 0, 0x7fffffff, 0x00020000);
    int wvoff[MR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) wvoff[mr] = (int)(((size_t)(mt0 + mr < p.mtiles ? mt0 + mr : p.mtiles - 1) * tile4 + lane) * 16);
    typedef float vfloat4 __attribute__((ext_vector_type(4)));
    // (the small single-buffer kernels run 3-5 blocks per CU and are not MFMA-issue bound: plain loads measured 3-6 % faster there)
    const float4* __restrict__ wq = reinterpret_cast<const float4*>(p.wp) + (size_t)(mt0 + MR <= p.mtiles ? mt0 : 0) * tile4 + lane;
    auto load_a = [&](int mr, int step) __attribute__((always_inline)) -> float4 {
        if constexpr (DB) {
            const vfloat4 v = __builtin_bit_cast(vfloat4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff[mr], step * 1024, 0));
            return make_float4(v.x, v.y, v.z, v.w);
        } else {
            return wq[(size_t)mr * tile4 + (size_t)step * 64];
        }
    };
    const int total_steps = p.nchunks * STEPS;

    // Software pipeline. A fragments (float4 = 4 MFMA k-steps per row tile) live in a ring of 4 register sets and
    // are fetched from L2 TWO steps (32 MFMAs, ~2k cycles) ahead of their use; ring position == step index mod 4,
    // and a tap is exactly 4 steps, so the ring needs no register moves. B values are read from LDS one k-step ahead.
    float4 ring[4][MR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
        ring[0][mr] = load_a(mr, 0);
        ring[1][mr] = load_a(mr, total_steps > 1 ? 1 : 0);
    }
    int gstep = 0;
    static_assert(CK / 8 == 4, "one tap must be 4 A-steps");

    // one chunk of MFMA work on the LDS tile whose lane base is `xrow`
    // (lr: apply leaky_relu at the B-operand read — a compile-time flag so that convolutions whose input is already
    // activated pay nothing for it)
    auto compute_chunk = [&](const float* xrow, auto lr) __attribute__((always_inline)) {
        constexpr bool LR = decltype(lr)::value;
        typedef const __attribute__((address_space(3))) float* LdsF;
        typedef std::conditional_t<DB, const volatile __attribute__((address_space(3))) float*, const __attribute__((address_space(3))) float*> LdsVF;
        LdsF xj = (LdsF)xrow;  // advances by `dil` floats per tap
        float b_nxt[NR];
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) b_nxt[nr] = *(LdsVF)(xj + nr * 32);  // (tap 0, pair 0)
        // fully unrolled taps: a taken branch every 64 MFMAs costs ~240 cycles of MFMA issue (measured on k = 11: K-loop
        // efficiency 0.871 -> 0.915); VITS_TAP_ROLLED keeps the rolled loop for comparison
        VITS_ABL_TAP_UNROLL
        for (int j = 0; j < KT; ++j) {
#pragma unroll
            for (int p4 = 0; p4 < 4; ++p4) {
                {
                    const int nstep = gstep + 2 < total_steps ? gstep + 2 : total_steps - 1;  // clamp: stays in bounds
                    VITS_ABL_UNLESS_NOA(_Pragma("unroll") for (int mr = 0; mr < MR; ++mr) ring[(p4 + 2) & 3][mr] = load_a(mr, nstep);)
                }
                // pin the prefetch at the top of the step: hipcc otherwise sinks the loads next to their first use
                // (vmcnt wait right behind the issue) and the L2 latency lands between MFMAs
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float b_cur[NR];
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr) {
                        b_cur[nr] = LR ? fmaxf(b_nxt[nr], b_nxt[nr] * slope_eff) : b_nxt[nr];
                    }
                    {
                        // next k-step: next channel pair of this tap, or pair 0 of the next tap (after the last tap this
                        // reads a few floats past the row: still inside the tile, value unused)
                        const int pair = p4 * 4 + q;
                        LdsF nx = pair + 1 < CK / 2 ? xj + (2 * (pair + 1)) * xw : xj + dil;
                        // (volatile: keeps one ds_read_b32 per value with a 16-bit immediate offset; merged into
                        // ds_read2_b32 — 8-bit offsets — every read needs a v_add_u32 for its base)
                        VITS_ABL_UNLESS_NOB(_Pragma("unroll") for (int nr = 0; nr < NR; ++nr) b_nxt[nr] = *(LdsVF)(nx + nr * 32);)
                    }
                    // pin the LDS read of the NEXT k-step in front of this step's MFMAs: left alone, hipcc reuses the
                    // registers of b_cur for b_nxt and therefore sinks the ds_read behind the last MFMA that reads them —
                    // one MFMA (64 cycles) in front of the s_waitcnt, less than the LDS latency: measured 14 % of the
                    // K loop of the k = 11 kernel idle on lgkmcnt
                    VITS_ABL_UNLESS_NOBPIN(__builtin_amdgcn_sched_barrier(0);)
#pragma unroll
                    for (int mr = 0; mr < MR; ++mr) {
                        const float4 a4 = ring[p4][mr];
                        const float av = q == 0 ? a4.x : q == 1 ? a4.y : q == 2 ? a4.z : a4.w;
#pragma unroll
                        for (int nr = 0; nr < NR; ++nr) {
                            acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b_cur[nr], acc[mr][nr], 0, 0, 0);
                            VITS_ABL_NOPS();
                        }
                    }
                }
                ++gstep;
            }
            xj += dil;
        }
    };

    static_assert(!DB || DIL != 0, "double buffering needs a compile-time row stride");
    // ---- staging: global -> registers -> LDS. One wave per row group (rows wid, wid+4, ...), lanes along time
    // (coalesced 256 B). Loads are BRANCH-FREE (clamped address + select), so all NK*NM loads of a chunk are in flight
    // together; with a per-element bounds branch hipcc serialises them (one HBM latency each).
    constexpr int NM = DB ? 1 : (DIL != 0 ? (BN + SPAN_C + 63) / 64 : BN / 64 + 1);  // 64-column groups per row (generic: span <= 64)
    constexpr int NK = DB ? 1 : CK / 4;
    float st[NK][NM];
    auto stage_load = [&](int c) __attribute__((always_inline)) {
        VITS_ABL_NOSTAGE_RETURN();
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int ch = c * CK + wid + 4 * k;
            const bool chok = ch < p.cin;
            const float* __restrict__ src = xb + (int64_t)(chok ? ch : p.cin - 1) * p.x_cs;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const int t = tile_start + lane + 64 * m;
                const int tc = t < 0 ? 0 : (t < len_in ? t : len_in - 1);
                const float v = src[tc];
                st[k][m] = (chok && t == tc) ? v : 0.f;
            }
        }
    };
    auto stage_store = [&](int buf) __attribute__((always_inline)) {
        VITS_ABL_NOSTAGE_RETURN();
        float* dst = xs + buf * (CK * xw) + wid * xw + lane;
#pragma unroll
        for (int k = 0; k < NK; ++k)
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                float v = st[k][m];
                if (p.pre_act) v = v > 0.f ? v : v * p.slope;
                if (lane + 64 * m < xw) dst[(4 * k) * xw + 64 * m] = v;
            }
    };
    if constexpr (DB) {
        // compute waves of the wave-specialised path (see the producer above)
        if constexpr (COOP) {
            if (oneshot) {
                // one-shot, compute side: what the producer's finish() does, for this wave's chunks
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const int ts = tile_start - shift;
                const bool interior = ts >= 0 && ts + XWP <= len_in;
                constexpr int NMP = (XWP + 63) / 64;
                bool oob[NMP];
#pragma unroll
                for (int m = 0; m < NMP; ++m) {
                    const int t = ts + lane + 64 * m;
                    oob[m] = t < 0 || t >= len_in;
                }
                for (int c = wid; c < p.nchunks; c += 5) {
                    float* lbase = xs + c * (CK * XWP);
                    if (!LRELU_AT_READ && p.pre_act) {
                        float4* l4 = reinterpret_cast<float4*>(lbase);
                        constexpr int N4 = CK * XWP / 4;
#pragma unroll 4
                        for (int i = lane; i < N4; i += 64) {
                            float4 v = l4[i];
                            v.x = fmaxf(v.x, v.x * p.slope);
                            v.y = fmaxf(v.y, v.y * p.slope);
                            v.z = fmaxf(v.z, v.z * p.slope);
                            v.w = fmaxf(v.w, v.w * p.slope);
                            l4[i] = v;
                        }
                    }
                    if ((c + 1) * CK > p.cin) {
#pragma unroll 4
                        for (int r = 0; r < CK; ++r) {
                            const bool chbad = c * CK + r >= p.cin;
#pragma unroll
                            for (int m = 0; m < NMP; ++m)
                                if ((oob[m] || chbad) && (XWP % 64 == 0 || 64 * m + lane < XWP)) lbase[r * XWP + 64 * m + lane] = 0.f;
                        }
                    } else if (!interior) {
                        zero_oob_columns<XWP>(lbase, ts, len_in, lane);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
        __syncthreads();
        VITS_STAMP(1);
        auto k_loop = [&](auto lr) __attribute__((always_inline)) {
            int buf = 0;
            for (int c = 0; c < p.nchunks; ++c) {
                VITS_ABL_YIELD(c + 1 == p.nchunks);
                compute_chunk(xrow0 + buf * (CK * xw), lr);
                buf = buf + 1 == p.nbuf ? 0 : buf + 1;
                VITS_CHUNK_STAMP(2 * c);
                if (c + 1 < p.nchunks && !oneshot) __syncthreads();  // (see the producer: the last chunk needs no barrier)
                VITS_CHUNK_STAMP(2 * c + 1);
            }
        };
        VITS_ABL_SETPRIO(0);
        if (LRELU_AT_READ && p.pre_act) k_loop(std::true_type{});
        else k_loop(std::false_type{});
        VITS_ABL_SETPRIO(3);
        VITS_STAMP(2);
    } else {
        // single LDS buffer (few chunks: nothing to overlap inside the block; other resident blocks hide the latency)
        VITS_STAMP(1);
        stage_load(0);
        for (int c = 0; c < p.nchunks; ++c) {
            if (c > 0) __syncthreads();  // everyone finished reading the previous chunk
            stage_store(0);
            __syncthreads();
            if (c + 1 < p.nchunks) stage_load(c + 1);  // in flight during the MFMA work
            __builtin_amdgcn_sched_barrier(0);
            compute_chunk(xrow0, std::false_type{});
        }
        VITS_STAMP(2);
    }

    // ---- epilogue -----------------------------------------------------------------------------------
    VITS_ABL_NOEPI_RETURN();
    VITS_ESTAMP(0);
    const int colbase = t0 + wn * (NR * 32) + (lane & 31);
    const int rowoff = 4 * (lane >> 5);
    if (EPI == EPI_STD) {
        // no __restrict__: residual / accumulator may alias the output (in-place updates)
        float* yb = p.y + (int64_t)b * p.y_bs;
        float* y2b = p.y2 ? p.y2 + (int64_t)b * p.y_bs : nullptr;
        const float* rb = p.res ? p.res + (int64_t)b * p.r_bs : nullptr;
        const float* ab = p.acc ? p.acc + (int64_t)b * p.a_bs : nullptr;
        // ---- wide path (interior tile, 16-byte aligned rows): the MFMA C layout gives a lane ONE column and 16 rows, i.e.
        // 64 dword stores (+64 dword loads with a residual) per lane and the epilogue becomes store-issue bound (16 % of a
        // k=3 conv). A 4x4 transpose inside each quad of lanes (two DPP quad_perm butterflies, no LDS) gives every lane
        // 4 CONSECUTIVE columns of one row instead -> dwordx4 loads / stores: 4x fewer memory instructions.
        const bool wide = (t0 + BN <= ncols) && ((mt0 + MR) * 32 <= p.cout) && (((p.y_cs | p.r_cs | p.a_cs) & 3) == 0) &&
                          ((((uintptr_t)yb | (uintptr_t)y2b | (uintptr_t)rb | (uintptr_t)ab) & 15) == 0);
        if (wide) {
            const int qi = lane & 3;                     // position in the quad == row offset after the transpose
            const int cq = (lane & 31) & ~3;             // first of this lane's 4 columns within the 32-column tile
            const bool odd1 = lane & 1, odd2 = lane & 2;
            // residual rows are fetched one (mr, nr) sub-tile AHEAD, i.e. before the stores of the current one: a load
            // issued behind stores can only be waited for together with them (one in-order vmcnt on gfx9)
            float4 rv[2][4], av[4];
            auto load_res = [&](int it, float4* dst) __attribute__((always_inline)) {
                const int mr = it / NR, nr = it % NR;
                const int tcol = t0 + wn * (NR * 32) + nr * 32 + cq;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = (mt0 + mr) * 32 + 8 * g + qi + rowoff;
                    dst[g] = *reinterpret_cast<const float4*>(rb + (int64_t)co * p.r_cs + tcol);
                }
            };
            if (rb) load_res(0, rv[0]);
            VITS_ESTAMP(1);
#pragma unroll
            for (int it = 0; it < MR * NR; ++it) {
                const int mr = it / NR, nr = it % NR;
                if (it < 4) VITS_ESTAMP(2 + it);
                {
                    const int tcol = t0 + wn * (NR * 32) + nr * 32 + cq;
                    if (rb && it + 1 < MR * NR) load_res(it + 1, rv[(it + 1) & 1]);
                    if (ab) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int co = (mt0 + mr) * 32 + 8 * g + qi + rowoff;
                            av[g] = *reinterpret_cast<const float4*>(ab + (int64_t)co * p.a_cs + tcol);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);  // keep the look-ahead loads in front of this sub-tile's stores
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float v0 = acc[mr][nr][4 * g + 0], v1 = acc[mr][nr][4 * g + 1], v2 = acc[mr][nr][4 * g + 2], v3 = acc[mr][nr][4 * g + 3];
                        // butterfly 1: exchange with lane ^ 1 (quad_perm [1,0,3,2] = 0xB1)
                        {
                            float s01 = odd1 ? v0 : v1, s23 = odd1 ? v2 : v3;
                            s01 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s01), 0xB1, 0xF, 0xF, true));
                            s23 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s23), 0xB1, 0xF, 0xF, true));
                            if (odd1) { v0 = s01; v2 = s23; } else { v1 = s01; v3 = s23; }
                        }
                        // butterfly 2: exchange with lane ^ 2 (quad_perm [2,3,0,1] = 0x4E)
                        {
                            float s02 = odd2 ? v0 : v2, s13 = odd2 ? v1 : v3;
                            s02 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s02), 0x4E, 0xF, 0xF, true));
                            s13 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s13), 0x4E, 0xF, 0xF, true));
                            if (odd2) { v0 = s02; v1 = s13; } else { v2 = s02; v3 = s13; }
                        }
                        // now (v0..v3) = columns tcol..tcol+3 of row co
                        const int co = (mt0 + mr) * 32 + 8 * g + qi + rowoff;
                        const float bias = bias_w[mr][g];
                        float o[4] = {v0 + bias, v1 + bias, v2 + bias, v3 + bias};
                        const float4 rg = rv[it & 1][g];
                        const float r4[4] = {rg.x, rg.y, rg.z, rg.w}, a4[4] = {av[g].x, av[g].y, av[g].z, av[g].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = o[e];
                            if (p.post_act == 1) v = v > 0.f ? v : 0.f;
                            if (rb) v = r4[e] + v;
                            if (ab) {
                                v = a4[e] + v;
                                v = p.scale_div ? v / p.scale : v * p.scale;
                            }
                            if (p.post_act == 2) v = fmaxf(v, v * p.post_slope);
                            o[e] = v;
                        }
                        *reinterpret_cast<float4*>(yb + (int64_t)co * p.y_cs + tcol) = make_float4(o[0], o[1], o[2], o[3]);
                        if (y2b)
                            *reinterpret_cast<float4*>(y2b + (int64_t)co * p.y_cs + tcol) =
                                make_float4(fmaxf(o[0], o[0] * p.post_slope), fmaxf(o[1], o[1] * p.post_slope), fmaxf(o[2], o[2] * p.post_slope),
                                            fmaxf(o[3], o[3] * p.post_slope));
                    }
                }
            }
        } else
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (mt0 + mr) * 32 + (r & 3) + 8 * (r >> 2) + rowoff;
                if (co >= p.cout) continue;
                const float bias = p.bias ? p.bias[co] : 0.f;
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) {
                    const int t = colbase + nr * 32;
                    if (t >= ncols) continue;
                    float v = acc[mr][nr][r] + bias;
                    if (p.post_act == 1) v = v > 0.f ? v : 0.f;
                    if (rb) v = rb[(int64_t)co * p.r_cs + t] + v;
                    if (ab) {
                        v = ab[(int64_t)co * p.a_cs + t] + v;
                        v = p.scale_div ? v / p.scale : v * p.scale;
                    }
                    if (p.post_act == 2) v = fmaxf(v, v * p.post_slope);
                    yb[(int64_t)co * p.y_cs + t] = v;
                    if (y2b) y2b[(int64_t)co * p.y_cs + t] = fmaxf(v, v * p.post_slope);
                }
            }
        }
    } else if (EPI == EPI_GATE) {
        // packed tile 2i = tanh rows (channels 32i..), tile 2i+1 = sigmoid rows (half + 32i..). MR == 2: both in this wave.
        // MR == 1 (128 x 32 tile, tiny grids: half the MFMA chain per wave): the odd wave hands its accumulators to the even wave
        // of the pair through LDS, lane for lane (both hold the same (row, column) positions), and retires.
        float sreg[NR][16];
        if constexpr (MR == 1 && NR == 1 && WN == 1 && (WM & 1) == 0) {  // (other MR == 1 shapes are never launched for the gated conv)
            float* ex = xs + (wm >> 1) * (16 * 64);
            __syncthreads();  // every compute wave has left the K loop: the input buffers are free
            if (wm & 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) ex[r * 64 + lane] = acc[0][0][r];
            }
            __syncthreads();
            if (wm & 1) return;
#pragma unroll
            for (int r = 0; r < 16; ++r) sreg[0][r] = ex[r * 64 + lane];
        } else {
#pragma unroll
            for (int nr = 0; nr < NR; ++nr)
#pragma unroll
                for (int r = 0; r < 16; ++r) sreg[nr][r] = acc[MR - 1][nr][r];
        }
        float* yb = p.y + (int64_t)b * p.y_bs;
        const int half = p.cout / 2;
        const int chbase = (mt0 / 2) * 32;
        const bool wide = (t0 + BN <= ncols) && (chbase + 32 <= half) && ((p.y_cs & 3) == 0) && (((uintptr_t)yb & 15) == 0);
        if (wide) {
            // same quad transpose as the standard epilogue: 4 consecutive time steps of one channel per lane -> dwordx4 stores
            const int qi = lane & 3, cq = (lane & 31) & ~3;
            const bool odd1 = lane & 1, odd2 = lane & 2;
            auto xpose = [&](float& v0, float& v1, float& v2, float& v3) __attribute__((always_inline)) {
                float s01 = odd1 ? v0 : v1, s23 = odd1 ? v2 : v3;
                s01 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s01), 0xB1, 0xF, 0xF, true));
                s23 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s23), 0xB1, 0xF, 0xF, true));
                if (odd1) { v0 = s01; v2 = s23; } else { v1 = s01; v3 = s23; }
                float s02 = odd2 ? v0 : v2, s13 = odd2 ? v1 : v3;
                s02 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s02), 0x4E, 0xF, 0xF, true));
                s13 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s13), 0x4E, 0xF, 0xF, true));
                if (odd2) { v0 = s02; v1 = s13; } else { v2 = s02; v3 = s13; }
            };
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                const int tcol = t0 + wn * (NR * 32) + nr * 32 + cq;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float a0 = acc[0][nr][4 * g], a1 = acc[0][nr][4 * g + 1], a2 = acc[0][nr][4 * g + 2], a3 = acc[0][nr][4 * g + 3];
                    float s0 = sreg[nr][4 * g], s1 = sreg[nr][4 * g + 1], s2 = sreg[nr][4 * g + 2], s3 = sreg[nr][4 * g + 3];
                    xpose(a0, a1, a2, a3);
                    xpose(s0, s1, s2, s3);
                    const int ch = chbase + 8 * g + qi + rowoff;
                    const float b0 = bias_w[0][g], b1 = MR == 1 ? bias_s[g] : bias_w[MR - 1][g];
                    float4 o;
                    o.x = wavenet_gate(a0 + b0, s0 + b1);
                    o.y = wavenet_gate(a1 + b0, s1 + b1);
                    o.z = wavenet_gate(a2 + b0, s2 + b1);
                    o.w = wavenet_gate(a3 + b0, s3 + b1);
                    *reinterpret_cast<float4*>(yb + (int64_t)ch * p.y_cs + tcol) = o;
                }
            }
        } else
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = chbase + (r & 3) + 8 * (r >> 2) + rowoff;
            if (ch >= half) continue;
            const float b0 = p.bias ? p.bias[ch] : 0.f, b1 = p.bias ? p.bias[ch + half] : 0.f;
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                const int t = colbase + nr * 32;
                if (t >= ncols) continue;
                yb[(int64_t)ch * p.y_cs + t] = wavenet_gate(acc[0][nr][r] + b0, sreg[nr][r] + b1);
            }
        }
    } else {  // EPI_CONVT: GEMM row rho = co*s + phase, column q; output sample n = s*q + phase - crop
        float* __restrict__ yb = p.y + (int64_t)b * p.y_bs;
        float* __restrict__ y2b = p.y2 ? p.y2 + (int64_t)b * p.y_bs : nullptr;  // leaky_relu(post_slope) copy for the resblocks' first convs
        const int s = p.ct_stride;
        const int out_len = p.len_out ? p.len_out[b] : p.t_out;
        // stride 8: the 4 registers of a group are phases 4h..4h+3 of ONE output channel at one input position, i.e. 4
        // consecutive output samples -> one dwordx4 store (interior tiles; crop is 0 or 4 so the address stays 16-B aligned)
        // (boundary tiles too: at ~225 frames per utterance the first upsampler has no interior tile at all, and the scalar path below
        // is 128 dword stores per lane; columns past the last one and the few samples outside [0, out_len) are guarded per group)
        const bool wide8 = s == 8 && (p.ct_crop & 3) == 0 && ((mt0 + MR) * 32 <= p.rows) && ((p.y_cs & 3) == 0) &&
                           ((((uintptr_t)yb | (uintptr_t)y2b) & 15) == 0);
        // stride 2: the 4 registers of a group are (channel, phase) = (c, 0), (c, 1), (c + 1, 0), (c + 1, 1) at one input position, i.e. two
        // consecutive output samples of two channels -> two 8-byte stores
        const bool wide2 = s == 2 && (p.ct_crop & 1) == 0 && ((mt0 + MR) * 32 <= p.rows) && ((p.y_cs & 1) == 0) &&
                           ((((uintptr_t)yb | (uintptr_t)y2b) & 7) == 0);
        auto lr = [&](float v) __attribute__((always_inline)) { return fmaxf(v, v * p.post_slope); };
        if (wide8) {
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = (mt0 + mr) * 4 + g;
                    const float bias = bias_w[mr][g];
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr) {
                        const int q = colbase + nr * 32;
                        if (q >= ncols) continue;
                        const int n = 8 * q + rowoff - p.ct_crop;
                        const float o0 = acc[mr][nr][4 * g] + bias, o1 = acc[mr][nr][4 * g + 1] + bias, o2 = acc[mr][nr][4 * g + 2] + bias,
                                    o3 = acc[mr][nr][4 * g + 3] + bias;
                        if (n >= 0 && n + 3 < out_len) {
                            *reinterpret_cast<float4*>(yb + (int64_t)co * p.y_cs + n) = make_float4(o0, o1, o2, o3);
                            if (y2b) *reinterpret_cast<float4*>(y2b + (int64_t)co * p.y_cs + n) = make_float4(lr(o0), lr(o1), lr(o2), lr(o3));
                        } else {
                            const float o[4] = {o0, o1, o2, o3};
                            for (int e = 0; e < 4; ++e)
                                if (n + e >= 0 && n + e < out_len) {
                                    yb[(int64_t)co * p.y_cs + n + e] = o[e];
                                    if (y2b) y2b[(int64_t)co * p.y_cs + n + e] = lr(o[e]);
                                }
                        }
                    }
                }
        } else if (wide2) {
            float b2[MR][4][2];  // every bias before the first store (a load behind stores waits for them: one in-order vmcnt)
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = (mt0 + mr) * 16 + 4 * g + 2 * (lane >> 5);
                    b2[mr][g][0] = p.bias ? p.bias[co] : 0.f;
                    b2[mr][g][1] = p.bias ? p.bias[co + 1] : 0.f;
                }
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = (mt0 + mr) * 16 + 4 * g + 2 * (lane >> 5);  // rows 32 (mt0 + mr) + 8 g + 4 h + {0..3} = channels co, co + 1
                    const float bias0 = b2[mr][g][0], bias1 = b2[mr][g][1];
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr) {
                        const int q = colbase + nr * 32;
                        if (q >= ncols) continue;
                        const int n = 2 * q - p.ct_crop;
                        const float o[4] = {acc[mr][nr][4 * g] + bias0, acc[mr][nr][4 * g + 1] + bias0, acc[mr][nr][4 * g + 2] + bias1, acc[mr][nr][4 * g + 3] + bias1};
                        if (n >= 0 && n + 1 < out_len) {
#pragma unroll
                            for (int c2 = 0; c2 < 2; ++c2) {
                                *reinterpret_cast<float2*>(yb + (int64_t)(co + c2) * p.y_cs + n) = make_float2(o[2 * c2], o[2 * c2 + 1]);
                                if (y2b) *reinterpret_cast<float2*>(y2b + (int64_t)(co + c2) * p.y_cs + n) = make_float2(lr(o[2 * c2]), lr(o[2 * c2 + 1]));
                            }
                        } else {
                            for (int e = 0; e < 4; ++e)
                                if (n + (e & 1) >= 0 && n + (e & 1) < out_len) {
                                    yb[(int64_t)(co + (e >> 1)) * p.y_cs + n + (e & 1)] = o[e];
                                    if (y2b) y2b[(int64_t)(co + (e >> 1)) * p.y_cs + n + (e & 1)] = lr(o[e]);
                                }
                        }
                    }
                }
        } else
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rho = (mt0 + mr) * 32 + (r & 3) + 8 * (r >> 2) + rowoff;
                if (rho >= p.rows) continue;
                const int co = rho / s, ph = rho - co * s;
                const float bias = p.bias ? p.bias[co] : 0.f;
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) {
                    const int q = colbase + nr * 32;
                    const int n = s * q + ph - p.ct_crop;
                    if (q >= ncols || n < 0 || n >= out_len) continue;
                    const float o = acc[mr][nr][r] + bias;
                    yb[(int64_t)co * p.y_cs + n] = o;
                    if (y2b) y2b[(int64_t)co * p.y_cs + n] = lr(o);
                }
            }
        }
    }
    VITS_ESTAMP(6);
    VITS_WSTAMP();
    VITS_STAMP(3);
}

template <int KT, int DIL, bool DB, int WM, int WN, int MR, int NR, int EPI>
__global__ __launch_bounds__(DB ? 320 : 256) VITS_WAVES_ATTR void conv_mfma_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float xs_dyn[];
    conv_mfma_body<KT, DIL, DB, WM, WN, MR, NR, EPI>(p, xs_dyn, blockIdx.x, blockIdx.y, blockIdx.z);
}


#if VITS_CONV_PART == 0
// ---- latency-bound launches (batch 1, short inputs): 16 x 16 output tiles on v_mfma_f32_16x16x4_f32 ----------------------------------
// A launch of the tiny grids (the encoder's FFN convs, the flow's gated convs at 128 tokens / 225 frames) has far fewer 32 x 32 output
// tiles than the chip has SIMDs (768 -> 192, k = 3, 128 tokens: 24 tiles on 1024 SIMDs), and the time of such a launch is ONE wave's
// dependent MFMA chain over the whole K range: 1152 x 64 cycles = 31 us of a 50 us launch. Splitting K is ruled out (batch 1 must equal
// a row of a batch bit for bit). But v_mfma_f32_16x16x4_f32 is, like v_mfma_f32_32x32x2_f32, bit for bit a sequential fmaf chain over
// its k indices (tools/mfma_bits.hip: both equal the scalar chain on 1,024 outputs x 256 products), at the same 32 MACs per cycle and SIMD:
// the same K order on 16 x 16 tiles gives FOUR times as many waves with a quarter of the chain each — 18k instead of 74k cycles for that
// conv — and the same bits. The price is operand traffic per MAC (A and B fragments feed a quarter of the MACs), which these launches have
// to spare.
//   block = 4 waves = 64 rows x 16 columns: wave w owns the 16-row half (w >> 1) of 32-row tile 2 * blockIdx.y + (w & 1) — for the
//   gated conv that is the tanh tile and the sigmoid tile of one channel group, and the odd wave hands its accumulators to the even one;
//   the whole input tile [c_in][16 + span] is staged in LDS once by all four waves (LeakyReLU and zero padding applied there);
//   A fragments: a second copy of the layer's weights in the operand order of this instruction (repack_conv_weights_l16; only for the
//   layers conv_lat16_candidate names): one 16-byte load per lane feeds four MFMAs = 16 input channels, fetched 14 loads ahead (read out
//   of the 32 x 32 array — half of every float4 unused — the stream was latency-bound: 768 -> 192, k = 3: 38 us against 41 for the 32 x 32
//   tile); K order = (chunk, tap, channel), as everywhere.
typedef float floatx4 __attribute__((ext_vector_type(4)));
template <int EPI, int P>  // P: LDS row pitch (floats) = 16 columns + span + shift (<= 3) rounded up: 24 (span <= 4), 40 (<= 20), 72 (<= 52); a
                           // compile-time constant so that the B reads of a tap are ONE base register + immediate offsets
__global__ __launch_bounds__(256) void conv_lat16_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float xs_dyn[];
    float* xs = xs_dyn;
    const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
    const int b = blockIdx.z, t0 = blockIdx.x * 16;
    const int len_in = p.len_in ? p.len_in[b] : p.t_in;
    const int ncols = p.len_out ? p.len_out[b] : p.t_out;
    if (t0 >= ncols || len_in <= 0) return;
#ifdef VITS_PHASE_TIMING
#define L16_STAMP(k)                                                                                      \
    do {                                                                                                  \
        if (tid == 0) {                                                                                   \
            const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);          \
            if (lin < 65536) vits_phase_buf[8 * lin + (k)] = __builtin_amdgcn_s_memrealtime();            \
        }                                                                                                 \
    } while (0)
#else
#define L16_STAMP(k)
#endif
    L16_STAMP(0);
    const int KT = p.kt_rt, dil = p.dil;
    const int cin_pad = p.nchunks * CK;
    const int tile_start = t0 - p.pad_l;
    const float* __restrict__ xb = p.x + (int64_t)b * p.x_bs;
    const int jg = lane >> 4, col = lane & 15;
    const int mtile = 2 * blockIdx.y + (wm & 1), half = wm >> 1;
    const int G = p.nchunks * KT;  // taps over the whole K range
    const int NQ = 2 * G;          // quads (16 channels of one tap: four MFMAs, one float4 of A per lane)
    // A stream: descriptor + per-lane byte offset (loop-invariant VGPR) + scalar quad offset — no vector ALU work per load; the array
    // carries AHEAD quads of slack at its end (repack_conv_weights_l16), so the look-ahead needs no clamp
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wl16), 0, 0x7fffffff, 0x00020000);
    const int wvoff = (int)((((size_t)(mtile < p.mtiles ? mtile : p.mtiles - 1) * 2 + half) * (size_t)NQ * 64 + lane) * 16);
    typedef float vfloat4 __attribute__((ext_vector_type(4)));
    auto load_q = [&](int q) __attribute__((always_inline)) -> float4 {
        const vfloat4 v = __builtin_bit_cast(vfloat4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, q * 1024, 0));
        return make_float4(v.x, v.y, v.z, v.w);
    };
    constexpr int R = 32, AHEAD = 30;  // A ring: 32 float4, fetched 30 quads (120 MFMAs, ~4k cycles) ahead: the weights of a batch-1 step come from HBM / the MALL, not from L2
    float4 ar[R];
#pragma unroll
    for (int i = 0; i < AHEAD; ++i) ar[i] = load_q(i);  // (in front of the fill: one memory latency for both)
    // the epilogue's operands — bias, residual, accumulator of a sum — requested with everything else (behind the K loop they were one more exposed
    // round trip per launch: 1-1.5 us of a 7-10 us conv, forty to sixty such launches per batch-1 utterance)
    float ep_bias[4], ep_res[4], ep_acc[4], ep_b1[4];
    {
        const int t_e = t0 + col;
        const float* rb_e = p.res ? p.res + (int64_t)b * p.r_bs : nullptr;
        const float* ab_e = p.acc ? p.acc + (int64_t)b * p.a_bs : nullptr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (EPI == EPI_STD) {
                const int co = mtile * 32 + 16 * half + 4 * jg + r;
                const bool ok = co < p.cout && t_e < ncols;
                ep_bias[r] = (p.bias && co < p.cout) ? p.bias[co] : 0.f;
                ep_res[r] = (rb_e && ok) ? rb_e[(int64_t)co * p.r_cs + t_e] : 0.f;
                ep_acc[r] = (ab_e && ok) ? ab_e[(int64_t)co * p.a_cs + t_e] : 0.f;
                ep_b1[r] = 0.f;
            } else {
                const int ch = (int)blockIdx.y * 32 + 16 * half + 4 * jg + r, halfc = p.cout / 2;
                ep_bias[r] = (p.bias && ch < halfc) ? p.bias[ch] : 0.f;
                ep_b1[r] = (p.bias && ch < halfc) ? p.bias[ch + halfc] : 0.f;
                ep_res[r] = ep_acc[r] = 0.f;
            }
        }
    }
    float lng[3], lnb[3];  // LayerNorm on load: this thread's share of gamma / beta (c_in <= 768), requested with everything else
    if (p.ln_gamma) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int c = tid + 256 * u;
            lng[u] = c < p.cin ? p.ln_gamma[c] : 0.f;
            lnb[u] = c < p.cin ? p.ln_beta[c] : 0.f;
        }
    }
    // ---- fill. 16-byte aligned rows (the engine's arenas): the 24 floats from the 4-aligned time below tile_start as six float4 per row,
    // ten rows per wave instruction, every load of a thread in flight at once; LDS column 0 = that aligned time, the B operands are read
    // `shift` floats further right (launch_lat16 sizes the pitch for it). Otherwise element by element.
    L16_STAMP(1);
    int shift = 0;
    if (p.l16_fill4) {  // (set by launch_lat16: rows 16-byte aligned)
        shift = tile_start & 3;
        const int ts = tile_start - shift;
        constexpr int F4 = P / 4, RPW = 64 / F4;  // float4 per row, rows per wave pass (P = 24: 6 and 10, lanes 60-63 idle)
        const int rsub = lane / F4, c4 = lane - F4 * rsub;
        const int t4 = ts + 4 * c4;
        const bool act = lane < F4 * RPW;
        const bool ld = act && t4 + 3 >= 0 && t4 < len_in;  // (t4 is a multiple of 4: t4 >= 0 or t4 <= -4)
        const float* src = xb + (ld ? t4 : 0);
        constexpr int U = 10;
        for (int rb = wm * RPW + rsub; rb < cin_pad; rb += 4 * RPW * U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = rb + 4 * RPW * u;
                v[u] = (ld && r < p.cin) ? *reinterpret_cast<const float4*>(src + (int64_t)r * p.x_cs) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = rb + 4 * RPW * u;
                float4 x = v[u];
                if (t4 + 1 >= len_in) x.y = 0.f;
                if (t4 + 2 >= len_in) x.z = 0.f;
                if (t4 + 3 >= len_in) x.w = 0.f;
                if (p.pre_act) {
                    x.x = fmaxf(x.x, x.x * p.slope);
                    x.y = fmaxf(x.y, x.y * p.slope);
                    x.z = fmaxf(x.z, x.z * p.slope);
                    x.w = fmaxf(x.w, x.w * p.slope);
                }
                if (act && r < cin_pad) *reinterpret_cast<float4*>(xs + r * P + 4 * c4) = x;
            }
        }
    } else {
        constexpr int LPR = P <= 32 ? 32 : (P <= 64 ? 64 : 128), RPP = 256 / LPR;  // lanes per row, rows per block pass
        const int cc = tid & (LPR - 1), r0 = tid / LPR;
        const int t = tile_start + cc;
        const bool tok = cc < P && t >= 0 && t < len_in;
        const float* src = xb + (tok ? t : 0);
        constexpr int U = 16;
        for (int rb = r0; rb < cin_pad; rb += RPP * U) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = rb + RPP * u;
                v[u] = (tok && r < p.cin) ? src[(int64_t)r * p.x_cs] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = rb + RPP * u;
                float x = v[u];
                if (p.pre_act) x = fmaxf(x, x * p.slope);
                if (cc < P && r < cin_pad) xs[r * P + cc] = x;
            }
        }
    }
    L16_STAMP(2);
    __syncthreads();
    if (p.ln_gamma) {
        // LayerNorm over the channels of every column of the tile that lies inside the sequence, in place (the encoder's add + norm nodes, vits.cpp:365-372,
        // 412-418, were a launch of add_layer_norm_kernel in front of this conv: 7 us for microseconds of work at batch 1). Same order of operations as that
        // kernel: sixteen channel groups, group g sums channels g, g + 16, ... in ascending order, the groups are combined in ascending order; variance the
        // same way around the mean; (v - mean) * inv * gamma + beta. Columns outside the sequence stay zero (the conv's padding).
        float* red = xs + ((size_t)p.nchunks + 1) * CK * P;  // [2][16][32] partial sums of up to 32 columns, then gamma[C], beta[C]
        float* gb = red + 2 * 16 * 32;
        const int C = p.cin;
        const int ncol = 16 + (KT - 1) * dil;  // the columns the K loop reads: LDS columns shift .. shift + ncol - 1 (<= 32: conv_ln_on_load_ok)
        const int tc0 = tile_start;            // time of the first of them
#pragma unroll
        for (int u = 0; u < 3; ++u) {  // (the two parameter vectors once, into LDS: a load per element in the last pass was 18 exposed round trips)
            const int c = tid + 256 * u;
            if (c < C) gb[c] = lng[u], gb[C + c] = lnb[u];
        }
        // thread (g = tid / 16, jj = tid % 16): channel group g of column jj — and of column jj + 16 where the tile has one (a 3-tap conv: two of them)
        const int g = tid >> 4, jj = tid & 15;
        const int ncs = jj + 16 < ncol ? 2 : 1;
        const float* colp = xs + shift + jj;
        for (int u = 0; u < ncs; ++u) {
            float s = 0.f;
#pragma unroll 12
            for (int c = g; c < C; c += 16) s += colp[c * P + 16 * u];
            red[g * 32 + jj + 16 * u] = s;
        }
        __syncthreads();
        float mean[2], inv[2];
        for (int u = 0; u < ncs; ++u) {
            float msum = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) msum += red[q * 32 + jj + 16 * u];
            mean[u] = msum / (float)C;
            float vs = 0.f;
#pragma unroll 12
            for (int c = g; c < C; c += 16) {
                const float d = colp[c * P + 16 * u] - mean[u];
                vs += d * d;
            }
            red[(16 + g) * 32 + jj + 16 * u] = vs;
        }
        __syncthreads();
        const bool writer = blockIdx.y == 0;
        float* lob = p.ln_out + (int64_t)b * p.lo_bs;
        for (int u = 0; u < ncs; ++u) {
            float vsum = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) vsum += red[(16 + q) * 32 + jj + 16 * u];
            const float var = vsum / (float)C;
            inv[u] = 1.0f / sqrtf(var + p.ln_eps);
        }
        if (ncs > 1) {  // the (few) columns past the sixteenth: their statistics to LDS, their elements are normalised by the whole block below
            if (g == 0) red[jj + 16] = mean[1], red[32 + jj + 16] = inv[1];
        }
        {
            const int tt = tc0 + jj;
            if (tt >= 0 && tt < len_in) {  // (outside the sequence: the conv's zero padding stays)
                float* cp = xs + shift + jj;
#pragma unroll 12
                for (int c = g; c < C; c += 16) {
                    const float v = (cp[c * P] - mean[0]) * inv[0] * gb[c] + gb[C + c];
                    cp[c * P] = v;
                    if (writer && tt >= t0 && tt < t0 + 16) lob[(int64_t)c * p.lo_cs + tt] = v;
                }
            }
        }
        if (ncol > 16) {
            __syncthreads();  // (block-uniform: ncol is)
            const int nx = ncol - 16;
            for (int idx = tid; idx < C * nx; idx += 256) {
                const int c = idx / nx, j = 16 + idx - c * nx, tt = tc0 + j;
                if (tt < 0 || tt >= len_in) continue;
                float* e = xs + c * P + shift + j;
                const float v = (*e - red[j]) * red[32 + j] * gb[c] + gb[C + c];
                *e = v;
                if (writer && tt >= t0 && tt < t0 + 16) lob[(int64_t)c * p.lo_cs + tt] = v;  // (a left pad puts the block's last own column(s) here)
            }
        }
        __syncthreads();
    }
    L16_STAMP(3);
    typedef const volatile __attribute__((address_space(3))) float* LdsVF;
    LdsVF xl = (LdsVF)(xs + jg * P + col + shift);
    float bq[2][8];
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) bq[0][s8] = xl[(4 * s8) * P];  // tap 0 = (chunk 0, tap 0)
    int cj = 0, toff = 0;  // tap counter within the chunk; LDS offset (floats) of the current tap = chunk * 32 * P + tap * dil
    const int wrap = CK * P - (KT - 1) * dil;
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int q0 = 0; q0 < NQ; q0 += R) {
#pragma unroll
        for (int tp = 0; tp < R / 2; ++tp) {  // (no break in here: the ring and the B registers need compile-time indices)
            if (q0 + 2 * tp < NQ) {
                // B operands of the NEXT tap (behind the last tap: one chunk of slack rows, allocated by launch_lat16, value unused)
                ++cj;
                if (cj == KT) {
                    cj = 0;
                    toff += wrap;
                } else {
                    toff += dil;
                }
                LdsVF nx = xl + toff;
                // issue order inside the tap, pinned: one of the next tap's eight LDS reads (and, twice, a look-ahead A load) behind every
                // MFMA. Left alone, hipcc puts all of them in front of the tap's first MFMA, where they are 60-100 cycles with the matrix
                // pipe idle (48 instead of 32 cycles per MFMA); sched_group_barrier patterns were only half honoured (42).
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int u = 2 * tp + h;
                    const float4 a4 = ar[u];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float av = k == 0 ? a4.x : k == 1 ? a4.y : k == 2 ? a4.z : a4.w;
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bq[tp & 1][4 * h + k], acc, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        bq[(tp + 1) & 1][4 * h + k] = nx[(4 * (4 * h + k)) * P];
                        if (k == 1) ar[(u + AHEAD) % R] = load_q(q0 + u + AHEAD);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
    }
    L16_STAMP(4);
    // ---- epilogue: reg r of lane l = row 4 * (l >> 4) + r, column l & 15 of the 16 x 16 tile
    const int t = t0 + col;
    if (EPI == EPI_STD) {
        float* yb = p.y + (int64_t)b * p.y_bs;
        float* y2b = p.y2 ? p.y2 + (int64_t)b * p.y_bs : nullptr;
        const float* rb = p.res ? p.res + (int64_t)b * p.r_bs : nullptr;
        const float* ab = p.acc ? p.acc + (int64_t)b * p.a_bs : nullptr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = mtile * 32 + 16 * half + 4 * jg + r;
            if (co >= p.cout || t >= ncols) continue;
            float v = acc[r] + ep_bias[r];
            if (p.post_act == 1) v = v > 0.f ? v : 0.f;
            if (rb) v = ep_res[r] + v;
            if (ab) {
                v = ep_acc[r] + v;
                v = p.scale_div ? v / p.scale : v * p.scale;
            }
            if (p.post_act == 2) v = fmaxf(v, v * p.post_slope);
            yb[(int64_t)co * p.y_cs + t] = v;
            if (y2b) y2b[(int64_t)co * p.y_cs + t] = fmaxf(v, v * p.post_slope);
        }
    } else {
        // gated conv: tile 2i = tanh rows of channels 32i.., tile 2i + 1 = their sigmoid rows (pack_conv_weights): the odd wave's
        // accumulators go to the even wave of the pair through LDS, lane for lane
        float* ex = xs + (wm >> 1) * 256;
        __syncthreads();  // every wave has left the K loop: the input tile is free
        if (wm & 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ex[r * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (wm & 1) return;
        float* yb = p.y + (int64_t)b * p.y_bs;
        const int halfc = p.cout / 2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ch = (int)blockIdx.y * 32 + 16 * half + 4 * jg + r;
            if (ch >= halfc || t >= ncols) continue;
            yb[(int64_t)ch * p.y_cs + t] = wavenet_gate(acc[r] + ep_bias[r], ex[r * 64 + lane] + ep_b1[r]);
        }
    }
}

// the launch: TILE_LAT16 from resolve_conv_tile
static int lat16_pitch(int span) { return span <= 4 ? 24 : span <= 20 ? 40 : span <= 52 ? 72 : 0; }
static hipError_t launch_lat16(const PackedConv& w, const ConvParams& p0, int ncols_max, int batch, hipStream_t s) {
    ConvParams p = p0;
    const int span = (w.kt - 1) * p.dil;
    const int pitch = lat16_pitch(span);
    if (!pitch || !w.wp_l16) return hipErrorInvalidValue;
    // 16-byte aligned rows: float4 fill from the 4-aligned time below the tile's first input
    p.l16_fill4 = (reinterpret_cast<uintptr_t>(p.x) & 15) == 0 && (p.x_cs & 3) == 0 && (p.x_bs & 3) == 0;
    p.xw = pitch;
    p.kt_rt = w.kt;
    p.wl16 = w.wp_l16;
    size_t lds = ((size_t)w.nchunks + 1) * CK * pitch * sizeof(float);  // (+ one chunk of slack rows: the look-ahead of the last tap)
    if (p.ln_gamma) lds += ((size_t)2 * 16 * 32 + 2 * (size_t)w.cin) * sizeof(float);  // LayerNorm on load: partial sums of up to 32 columns; gamma, beta
    dim3 grid((ncols_max + 15) / 16, (w.mtiles_used + 1) / 2, batch);
#define VITS_L16(E, PP)                                                                                                                          \
    do {                                                                                                                                         \
        static BigLdsOnce big;                                                                                                                   \
        if (lds > 64 * 1024 && big.needed()) {                                                                                                   \
            if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_lat16_kernel<E, PP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) \
                return e;                                                                                                                        \
            big.done();                                                                                                                          \
        }                                                                                                                                        \
        VITS_KLAUNCH((conv_lat16_kernel<E, PP>), grid, dim3(256), lds, s, p);                                                                    \
    } while (0)
    if (w.epi == EPI_GATE) {
        if (pitch != 24) return hipErrorInvalidValue;
        VITS_L16(EPI_GATE, 24);
    } else if (pitch == 24) {
        VITS_L16(EPI_STD, 24);
    } else if (pitch == 40) {
        VITS_L16(EPI_STD, 40);
    } else {
        VITS_L16(EPI_STD, 72);
    }
#undef VITS_L16
    return hipGetLastError();
}
#endif  // VITS_CONV_PART == 0

// ---- grouped launch: the convolutions of the SAME position in the (up to) three ResBlocks of a vocoder stage (kernel sizes 11 / 7 /
// 3, one dilation, one channel count; vits.cpp:622-635 runs the resblocks on the same input) as ONE launch -----------------------
// The resblocks are independent chains, so conv number i of each can run side by side; launched one by one, each of them ends in a
// partially filled round of blocks (C = 256: 7.2 rounds of 80 us blocks -> 8; three such tails per position). Here blockIdx.z =
// member * batch + utterance with the members ordered by DESCENDING tap count: the k = 11 blocks are dispatched first and the last
// round is made of k = 3 blocks, a quarter as long. Same body per member as conv_mfma_kernel -> bit-identical results.
struct ConvGroupParams {
    ConvParams m[3];  // slot 0: 11 taps, 1: 7 taps, 2: 3 taps
    int zend[3];      // blockIdx.z < zend[i] -> member i (an absent member has zend[i] == zend[i - 1])
};
template <int DIL>
__global__ __launch_bounds__(320) VITS_WAVES_ATTR void conv_group_kernel(const ConvGroupParams g) {
    extern __shared__ __attribute__((aligned(16))) float xs_dyn[];
    const int z = blockIdx.z;
    if (z < g.zend[0]) conv_mfma_body<11, DIL, true, 2, 2, 2, 2, EPI_STD>(g.m[0], xs_dyn, blockIdx.x, blockIdx.y, z);
    else if (z < g.zend[1]) conv_mfma_body<7, DIL, true, 2, 2, 2, 2, EPI_STD>(g.m[1], xs_dyn, blockIdx.x, blockIdx.y, z - g.zend[0]);
    else conv_mfma_body<3, DIL, true, 2, 2, 2, 2, EPI_STD>(g.m[2], xs_dyn, blockIdx.x, blockIdx.y, z - g.zend[1]);
}

// ---- host side --------------------------------------------------------------------------------------------
hipError_t make_conv_params(const PackedConv& w, const ConvCall& c, int tile, ConvParams& p);
struct TileShape {
    int wm, wn, mr, nr;
};
static TileShape tile_shape(int tile) {
    switch (tile) {
        case TILE_128x128: return {2, 2, 2, 2};
        case TILE_64x256: return {1, 4, 2, 2};
        case TILE_32x256: return {1, 4, 1, 2};
        case TILE_64x64: return {1, 4, 2, 1};  // 64 x 128
        case TILE_LAT16:  // (conv_lat16_kernel has its own grid; parameters are set up as for the narrow tile)
        case TILE_NARROW: return {4, 1, 1, 1};  // 128 x 32: four row tiles of ONE 32-column strip
        default: return {1, 4, 1, 1};          // TILE_32x64: 32 x 128
    }
}

#if VITS_CONV_PART == 0
int choose_conv_tile(int rows, int epi, int t_hint) {
    // rows <= 64 (few MFMAs per staged tile): 128-column tiles -> twice as many independent blocks per CU keep more
    // loads in flight (measured 120.5 -> 116.4 ms per step); VITS_NARROW_TILES=0 restores 256-column tiles
    const int narrow = kernel_knobs().narrow_tiles;
    const bool small_t = t_hint <= 128 || (narrow && rows <= narrow);
    if (epi == EPI_GATE) return small_t ? TILE_64x64 : TILE_64x256;
    const int t128 = kernel_knobs().tile128;
    if (rows % 128 == 0 && t128) return TILE_128x128;
    if (rows % 64 == 0) return small_t ? TILE_64x64 : TILE_64x256;
    return small_t ? TILE_32x64 : TILE_32x256;
}

// Packed layout: [mtile][chunk][tap][p4 = pair/4][lane][q = pair%4]; the value for (mtile, chunk c, tap j, pair p,
// lane l) is A[row = mtile*32 + (l&31)][ci = c*32 + 2p + (l>>5)][tap j] — lane l's A operand of the MFMA
// that consumes input channels (2p, 2p+1) of chunk c at tap j.
std::vector<float> pack_conv_weights(const float* w, int cout, int cin, int k, int epi, int ct_stride, int* rows_out, int* mtiles_used_out,
                                     int* mtiles_out, int* nchunks_out) {
    const int bm_tiles = 4;  // padded so that every tile shape (1, 2 or 4 row tiles per block) divides it
    const int half = cout / 2;
    int rows, kt;
    if (epi == EPI_CONVT) {
        rows = cout * ct_stride;
        kt = k / ct_stride;  // == 2 taps per phase
    } else {
        rows = cout;
        kt = k;
    }
    int mtiles = (rows + 31) / 32;
    if (epi == EPI_GATE) mtiles = 2 * ((half + 31) / 32);
    *mtiles_used_out = mtiles;
    mtiles = (mtiles + bm_tiles - 1) / bm_tiles * bm_tiles;
    const int nchunks = (cin + CK - 1) / CK;
    std::vector<float> out((size_t)mtiles * nchunks * kt * (CK / 2) * 64, 0.f);
    for (int mt = 0; mt < mtiles; ++mt)
        for (int c = 0; c < nchunks; ++c)
            for (int j = 0; j < kt; ++j)
                for (int pr = 0; pr < CK / 2; ++pr)
                    for (int l = 0; l < 64; ++l) {
                        const int ci = c * CK + 2 * pr + (l >> 5);
                        const int r = l & 31;
                        float v = 0.f;
                        if (ci < cin) {
                            if (epi == EPI_STD) {
                                const int co = mt * 32 + r;
                                if (co < cout) v = w[((size_t)co * cin + ci) * k + j];
                            } else if (epi == EPI_GATE) {
                                const int ch = (mt / 2) * 32 + r;
                                const int co = (mt & 1) ? half + ch : ch;
                                if (ch < half) v = w[((size_t)co * cin + ci) * k + j];
                            } else {
                                // y[co][s*q + ph - crop] = sum_ci sum_m x[ci][q - m] * W[ci][co][ph + s*m]  
                                const int rho = mt * 32 + r;
                                const int co = rho / ct_stride, ph = rho % ct_stride;
                                if (co < cout) v = w[((size_t)ci * cout + co) * k + ph + ct_stride * j];
                            }
                        }
                        const size_t idx = (((((size_t)mt * nchunks + c) * kt + j) * (CK / 8) + pr / 4) * 64 + l) * 4 + (pr & 3);
                        out[idx] = v;
                    }
    *rows_out = rows;
    *mtiles_out = mtiles;
    *nchunks_out = nchunks;
    return out;
}

bool conv_lat16_candidate(int epi, int kt, int cin) {
    // (>= 64 products per output: everything but the degenerate convs. The long chains — FFN, gated, vocoder — win by the chain (41 -> 17.7 us);
    // the 1x1 convs, whose chain is only 2.6 us, by the fill and the finer grid: 13-16 -> 7 us at batch 1)
    return (epi == EPI_STD || (epi == EPI_GATE && kt == 5)) && (int64_t)kt * cin >= 64;
}

std::vector<float> repack_conv_weights_l16(const std::vector<float>& packed, int mtiles, int nchunks, int kt) {
    std::vector<float> out(packed.size() + 32 * 256, 0.f);  // (+ 32 quads of slack: conv_lat16_kernel's look-ahead reads past the last stream)
    const size_t G = (size_t)nchunks * kt;
    for (int mt = 0; mt < mtiles; ++mt)
        for (int half = 0; half < 2; ++half)
            for (size_t g = 0; g < G; ++g)
                for (int hq = 0; hq < 2; ++hq)  // the two 16-channel quads of a tap's 32-channel chunk
                    for (int l = 0; l < 64; ++l)
                        for (int sc = 0; sc < 4; ++sc) {
                            const int ci = 16 * hq + 4 * sc + (l >> 4);  // channel within the chunk
                            const int pr = ci >> 1, par = ci & 1;
                            const size_t src = (((((size_t)mt * G + g) * (CK / 8)) + pr / 4) * 64 + (16 * half + (l & 15) + 32 * par)) * 4 + (pr & 3);
                            const size_t dst = (((((size_t)mt * 2 + half) * (2 * G)) + 2 * g + hq) * 64 + l) * 4 + sc;
                            out[dst] = packed[src];
                        }
    return out;
}

#endif  // VITS_CONV_PART == 0

template <int KT, int DIL, bool DB, int EPI>
static hipError_t launch_tile(const PackedConv& w, int tile, const ConvParams& p, int ncols_max, int batch, hipStream_t s) {
    const TileShape ts = tile_shape(tile);
    const int bn = ts.wn * ts.nr * 32;
    const int bm_tiles = ts.wm * ts.mr;
    dim3 grid((ncols_max + bn - 1) / bn, (w.mtiles_used + bm_tiles - 1) / bm_tiles, batch);
    const size_t lds = DB ? (size_t)p.nbuf * CK * ((p.xw + 3 + VITS_XWP_GRAN - 1) / VITS_XWP_GRAN * VITS_XWP_GRAN) * sizeof(float) : (size_t)CK * p.xw * sizeof(float);
#define VITS_LAUNCH(WM, WN, MR, NR)                                                                                                   \
    do {                                                                                                                              \
        static BigLdsOnce big_lds_set; /* (atomic: distinct model handles may launch from distinct threads) */         \
        if (lds > 64 * 1024 && big_lds_set.needed()) {                                                                                        \
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<KT, DIL, DB, WM, WN, MR, NR, EPI>),       \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                             \
            if (ea != hipSuccess) return ea;                                                                                          \
            big_lds_set.done();                                                                                                 \
        }                                                                                                                             \
        VITS_KLAUNCH((conv_mfma_kernel<KT, DIL, DB, WM, WN, MR, NR, EPI>), grid, dim3(DB ? 320 : 256), lds, s, p);                             \
    } while (0)
    switch (tile) {
        case TILE_128x128:
            if (EPI == EPI_GATE) return hipErrorInvalidValue;
            VITS_LAUNCH(2, 2, 2, 2);
            break;
        case TILE_64x256: VITS_LAUNCH(1, 4, 2, 2); break;
        case TILE_64x64: VITS_LAUNCH(1, 4, 2, 1); break;
        case TILE_32x256:
            if (EPI == EPI_GATE) return hipErrorInvalidValue;
            VITS_LAUNCH(1, 4, 1, 2);
            break;
        case TILE_NARROW:
            // (only where launch_conv chooses it: encoder / flow convs on the producer-wave path)
            if constexpr (DB && DIL == 1 && ((EPI == EPI_STD && KT <= 3) || (EPI == EPI_GATE && KT == 5))) {
                VITS_LAUNCH(4, 1, 1, 1);  // (gated conv: tanh / sigmoid row tiles on wave pairs, see the epilogue)
                break;
            } else {
                return hipErrorInvalidValue;
            }
        default:
            if (EPI == EPI_GATE) return hipErrorInvalidValue;
            VITS_LAUNCH(1, 4, 1, 1);
            break;
    }
#undef VITS_LAUNCH
    return hipGetLastError();
}

// one launcher per tap count (see VITS_CONV_PART): picks the compile-time dilation and the producer-wave variant
#define VITS_GO(K, D, E)                                                                     \
    do {                                                                                     \
        if ((D) != 0 && db) return launch_tile<K, D, (D) != 0, E>(w, tile, p, ncols_max, batch, s); \
        return launch_tile<K, D, false, E>(w, tile, p, ncols_max, batch, s);                 \
    } while (0)
#define VITS_LAUNCHER(K) hipError_t launch_conv_k##K(const PackedConv& w, int tile, const ConvParams& p, int ncols_max, int batch, hipStream_t s, bool db)
VITS_LAUNCHER(3);
VITS_LAUNCHER(7);
VITS_LAUNCHER(11);
#if VITS_CONV_PART == 1
VITS_LAUNCHER(3) {
    if (p.dil == 1) VITS_GO(3, 1, EPI_STD);
    if (p.dil == 3) VITS_GO(3, 3, EPI_STD);
    if (p.dil == 5) VITS_GO(3, 5, EPI_STD);
    VITS_GO(3, 0, EPI_STD);
}
#endif
#if VITS_CONV_PART == 2
VITS_LAUNCHER(7) {
    if (p.dil == 1) VITS_GO(7, 1, EPI_STD);
    if (p.dil == 3) VITS_GO(7, 3, EPI_STD);
    if (p.dil == 5) VITS_GO(7, 5, EPI_STD);
    VITS_GO(7, 0, EPI_STD);
}
#endif
#if VITS_CONV_PART == 3
VITS_LAUNCHER(11) {
    if (p.dil == 1) VITS_GO(11, 1, EPI_STD);
    if (p.dil == 3) VITS_GO(11, 3, EPI_STD);
    if (p.dil == 5) VITS_GO(11, 5, EPI_STD);
    VITS_GO(11, 0, EPI_STD);
}
#endif

#if VITS_CONV_PART == 4
bool conv_group_supported(const PackedConv& w, int dil) {
    return w.epi == EPI_STD && (w.kt == 11 || w.kt == 7 || w.kt == 3) && (dil == 1 || dil == 3 || dil == 5) && w.rows % 128 == 0 && w.cin % CK == 0;
}
hipError_t launch_conv_group(const PackedConv* const* w, const ConvCall* c, int n, hipStream_t s) {
    if (n < 1 || n > 3) return hipErrorInvalidValue;
    ConvGroupParams g;
    std::memset(&g, 0, sizeof(g));
    const int slot_kt[3] = {11, 7, 3};
    int have[3] = {-1, -1, -1};
    for (int i = 0; i < n; ++i) {
        if (!conv_group_supported(*w[i], c[i].dil) || c[i].dil != c[0].dil || c[i].batch != c[0].batch || w[i]->rows != w[0]->rows || c[i].tile >= 0) return hipErrorInvalidValue;
        const int slot = w[i]->kt == 11 ? 0 : w[i]->kt == 7 ? 1 : 2;
        if (have[slot] >= 0) return hipErrorInvalidValue;
        have[slot] = i;
    }
    int z = 0, ncols_max = 0;
    size_t lds = 0;
    for (int slot = 0; slot < 3; ++slot) {
        if (have[slot] >= 0) {
            const int i = have[slot];
            if (hipError_t e = make_conv_params(*w[i], c[i], TILE_128x128, g.m[slot])) return e;
            if (g.m[slot].oneshot) return hipErrorInvalidValue;  // (never: 128 x 128 tiles have MR * NR = 4)
            g.m[slot].nbuf = 2;
            z += c[i].batch;
            ncols_max = std::max(ncols_max, c[i].t_out);
            const int xwp = (g.m[slot].xw + 3 + VITS_XWP_GRAN - 1) / VITS_XWP_GRAN * VITS_XWP_GRAN;
            lds = std::max(lds, (size_t)2 * CK * xwp * sizeof(float));
        }
        g.zend[slot] = z;
        (void)slot_kt;
    }
    dim3 grid((ncols_max + 127) / 128, w[0]->mtiles_used / 4, z);
#define VITS_GROUP_LAUNCH(D)                                                                                                              \
    do {                                                                                                                                  \
        static BigLdsOnce big_lds_set;                                                                                      \
        if (lds > 64 * 1024 && big_lds_set.needed()) {                                                            \
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_group_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            if (ea != hipSuccess) return ea;                                                                                              \
            big_lds_set.done();                                                                           \
        }                                                                                                                                 \
        VITS_KLAUNCH((conv_group_kernel<D>), grid, dim3(320), lds, s, g);                                                           \
    } while (0)
    if (c[0].dil == 1) VITS_GROUP_LAUNCH(1);
    else if (c[0].dil == 3) VITS_GROUP_LAUNCH(3);
    else VITS_GROUP_LAUNCH(5);
#undef VITS_GROUP_LAUNCH
    return hipGetLastError();
}
#endif

#if VITS_CONV_PART == 0
// The tile a launch will run on: shape rule (choose_conv_tile), then the small-grid steps. Also what the engine's profiler prints.
int resolve_conv_tile(const PackedConv& w, const ConvCall& c) {
    const int ncols_max = w.epi == EPI_CONVT ? c.t_in + 1 : c.t_out;
    int tile = c.tile >= 0 ? c.tile : choose_conv_tile(w.rows, w.epi, ncols_max);
    if (c.tile < 0 && w.epi == EPI_GATE) {
        const TileShape t2 = tile_shape(tile);
        const int64_t nb = (ncols_max + t2.wn * t2.nr * 32 - 1) / (t2.wn * t2.nr * 32);
        if (nb * ((w.mtiles_used + 1) / 2) * c.batch < 512) tile = TILE_64x64;  // 64 x 128 keeps the tanh/sigmoid row pairing
    }
    if (c.tile < 0 && w.epi != EPI_GATE) {
        // small grids (batch 1, short inputs): fewer than ~2 blocks per CU leaves matrix pipes idle -> step down to
        // smaller tiles until the launch has >= 512 blocks (latency case, BASELINE.json config 2)
        auto blocks = [&](int tl) {
            const TileShape t2 = tile_shape(tl);
            const int64_t nb = (ncols_max + t2.wn * t2.nr * 32 - 1) / (t2.wn * t2.nr * 32);
            const int64_t mb = (w.mtiles_used + t2.wm * t2.mr - 1) / (t2.wm * t2.mr);
            return nb * mb * c.batch;
        };
        // (k <= 3: 1024 — a short K loop costs a small tile little, and e.g. the encoder's 192 -> 768 FFN conv at batch 64 x 128 tokens is
        // 768 blocks of 64 x 128 = 1.5 rounds of the 512 resident blocks, but 3 even rounds of 32 x 128: 82 -> 74 us)
        const int64_t min_blocks_env = kernel_knobs().min_blocks;
        const int64_t min_blocks = min_blocks_env > 0 ? min_blocks_env : (w.kt <= 3 ? 1024 : 512);
        if (blocks(tile) < min_blocks && (tile == TILE_128x128 || tile == TILE_64x256)) tile = TILE_64x64;  // 64 x 128
        if (blocks(tile) < min_blocks && (tile == TILE_64x64 || tile == TILE_32x256)) tile = TILE_32x64;    // 32 x 128
    }
    if (c.tile < 0 && w.epi != EPI_CONVT) {
        // tiny grids (the encoder / duration predictor / flow at batch 1: 6-18 blocks of the 128-column tiles): every block of a
        // 128-column tile streams the WHOLE input in through its one producer wave, and that stream, not the MFMA chain, is the
        // launch time (768 -> 192 FFN conv, k = 3, 128 tokens: 78 us for a 31 us chain). Blocks of four row tiles x ONE 32-column
        // strip need a quarter of the input each. Same per-output accumulation order (the tile shape never changes it).
        const bool no_narrow = kernel_knobs().no_narrow;
        const int dil_eff = w.kt == 1 ? 1 : c.dil;
        const bool shape_ok = dil_eff == 1 && ((w.epi == EPI_STD && w.kt <= 3) || (w.epi == EPI_GATE && w.kt == 5));
        const TileShape t2 = tile_shape(tile);
        const int64_t nb = (int64_t)((ncols_max + t2.wn * t2.nr * 32 - 1) / (t2.wn * t2.nr * 32)) * ((w.mtiles_used + t2.wm * t2.mr - 1) / (t2.wm * t2.mr)) * c.batch;
        // 1x1 convs (QKV / output / projection convs of the encoder, the flow's pre / post convs): the narrow tile on large grids too —
        // 192 -> 576 at batch 64 x 128 tokens 34 -> 26 us, 192 -> 192 21 -> 13 us; at 1024 tokens (config 5) the 1x1 convs of a step 0.83 ->
        // 0.65 ms (bf16 run), 1.34 -> 1.05 ms (fp32 run). VITS_NARROW_K1 = longest sequence that takes it (0: small grids only)
        const int narrow_k1 = kernel_knobs().narrow_k1;
        const bool k1_short = narrow_k1 > 0 && w.epi == EPI_STD && w.kt == 1 && ncols_max <= narrow_k1;
        if (!no_narrow && shape_ok && (nb <= 128 || k1_short)) tile = TILE_NARROW;
        // ... and where the launch time is one wave's MFMA chain (a second copy of the weights exists for the layers with >= 512 products
        // per output: not the 1x1 convs, whose chain is 2.6 us of a launch that is bound by its fill), 16 x 16 tiles on
        // v_mfma_f32_16x16x4_f32 (conv_lat16_kernel): the tiny grids of the narrow tile, and any standard conv whose 32 x 32 tiles would
        // not even fill the SIMDs once (batch 1: the C = 256 stage of the vocoder, conv_pre)
        const int pitch16 = lat16_pitch((w.kt - 1) * dil_eff);
        const int64_t waves32 = (int64_t)((ncols_max + 31) / 32) * w.mtiles_used * c.batch;
        const bool tiny = tile == TILE_NARROW && nb <= 128;
        const bool unfilled = w.epi == EPI_STD && w.kt >= 3 && dil_eff >= 1 && waves32 <= kernel_knobs().lat16_max_waves;
        if ((tiny || unfilled) && !kernel_knobs().no_lat16 && w.wp_l16 && pitch16 && (w.epi == EPI_STD || pitch16 == 24) &&
            ((size_t)w.nchunks + 1) * CK * pitch16 * 4 <= 150 * 1024)
            tile = TILE_LAT16;
    }
    return tile;
}

// ConvCall -> kernel parameters for the tile `tile` (also used by the grouped launch, part 4)
hipError_t make_conv_params(const PackedConv& w, const ConvCall& c, int tile, ConvParams& p) {
    p.x = c.x.p;
    p.x_bs = c.x.bs;
    p.x_cs = c.x.cs;
    p.y = c.y.p;
    p.y_bs = c.y.bs;
    p.y_cs = c.y.cs;
    p.res = c.res.p;
    p.r_bs = c.res.bs;
    p.r_cs = c.res.cs;
    p.acc = c.acc.p;
    p.a_bs = c.acc.bs;
    p.a_cs = c.acc.cs;
    p.wp = w.wp;
    p.bias = w.bias;
    p.len_in = c.len_in;
    p.len_out = c.len_out;
    p.t_in = c.t_in;
    p.t_out = c.t_out;
    p.cin = w.cin;
    p.cout = w.cout;
    p.rows = w.rows;
    p.nchunks = w.nchunks;
    p.mtiles = w.mtiles;
    p.pre_act = c.pre_act;
    p.slope = c.slope;
    p.post_act = c.post_act;
    p.post_slope = c.post_slope;
    p.y2 = c.y2;
    p.scale = c.scale;
    p.scale_div = c.scale_div;
    p.ct_stride = w.ct_stride;
    p.ct_crop = c.ct_crop;
    p.kt_rt = w.kt;
    p.wl16 = w.wp_l16;
    p.l16_fill4 = 0;
    p.ln_gamma = c.ln_gamma;
    p.ln_beta = c.ln_beta;
    p.ln_eps = c.ln_eps;
    p.ln_out = c.ln_out.p;
    p.lo_bs = c.ln_out.bs;
    p.lo_cs = c.ln_out.cs;
    const int ncols_max = w.epi == EPI_CONVT ? c.t_in + 1 : c.t_out;
    const TileShape ts = tile_shape(tile);
    const int bn = ts.wn * ts.nr * 32;
    if (w.epi == EPI_CONVT) {
        p.dil = -1;  // tap m reads x[q - m]
        p.pad_l = 0;
    } else {
        p.dil = w.kt == 1 ? 1 : c.dil;
        p.pad_l = c.pad_l;
    }
    const int span = (w.kt - 1) * p.dil;  // signed extent of the taps
    p.lds_off = span < 0 ? -span : 0;
    p.xw = bn + (span < 0 ? -span : span);
    if ((size_t)2 * CK * p.xw * 4 > 160 * 1024) return hipErrorInvalidValue;
    {
        // third LDS buffer (DMA two chunks ahead) where a chunk is less MFMA work than a DMA round trip (~2.5 us = 6k cycles):
        // taps x (MFMAs per k-step) x 16 k-steps x 64 cycles
        const int nbuf_env = kernel_knobs().nbuf;
        const TileShape t3 = ts;
        const bool short_chunk = w.kt * t3.mr * t3.nr * 1024 < 8000 && w.nchunks >= 3 && bn == 128;
        p.nbuf = nbuf_env == 2 || nbuf_env == 3 ? nbuf_env : (short_chunk ? 3 : 2);
        if (w.nchunks < 2 || (size_t)p.nbuf * CK * ((p.xw + 3 + VITS_XWP_GRAN - 1) / VITS_XWP_GRAN * VITS_XWP_GRAN) * 4 > 150 * 1024) p.nbuf = 2;
        // latency-bound launch on a small tile whose whole input fits in LDS: cooperative one-shot fill (see the kernel)
        const bool no_oneshot = kernel_knobs().no_oneshot;
        const int64_t nblocks = (int64_t)((ncols_max + bn - 1) / bn) * ((w.mtiles_used + t3.wm * t3.mr - 1) / (t3.wm * t3.mr)) * c.batch;
        p.oneshot = 0;
        if (!no_oneshot && t3.mr * t3.nr <= 2 && nblocks <= 512 && w.nchunks >= 2 &&
            (size_t)w.nchunks * CK * ((p.xw + 3 + VITS_XWP_GRAN - 1) / VITS_XWP_GRAN * VITS_XWP_GRAN) * 4 <= 150 * 1024) {
            p.oneshot = 1;
            p.nbuf = w.nchunks;
        }
    }
    return hipSuccess;
}

// LayerNorm on load exists in conv_lat16_kernel only (the launches it pays for are the latency-bound ones): a standard conv without an input activation
// whose tile choice is TILE_LAT16, with the statistics' scratch beside the input tile in LDS, and the input lengths = the output lengths (padded 'same' conv)
bool conv_ln_on_load_ok(const PackedConv& w, const ConvCall& c) {
    if (w.epi != EPI_STD || c.pre_act || !w.wp_l16 || c.len_in != c.len_out || c.t_in != c.t_out) return false;
    if (resolve_conv_tile(w, c) != TILE_LAT16) return false;
    const int span = (w.kt - 1) * (w.kt == 1 ? 1 : c.dil);
    const int pitch = lat16_pitch(span);
    return pitch && span <= 16 && (((size_t)w.nchunks + 1) * CK * pitch + 2 * 16 * 32 + 2 * (size_t)w.cin) * sizeof(float) <= 150 * 1024;
}

hipError_t launch_conv(const PackedConv& w, const ConvCall& c, hipStream_t s) {
    ConvParams p;
    const int ncols_max = w.epi == EPI_CONVT ? c.t_in + 1 : c.t_out;
    const int tile = resolve_conv_tile(w, c);
    if (hipError_t e = make_conv_params(w, c, tile, p)) return e;
    if (c.ln_gamma && (tile != TILE_LAT16 || !c.ln_beta || !c.ln_out.p || !conv_ln_on_load_ok(w, c))) return hipErrorInvalidValue;
    if (tile == TILE_LAT16) return launch_lat16(w, p, ncols_max, c.batch, s);
    const int span = (w.kt - 1) * p.dil;  // signed extent of the taps
    if ((span < 0 ? -span : span) > 64) return hipErrorInvalidValue;  // generic kernels stage at most BN + 64 columns
    const int batch = c.batch;
    // compile-time dilation for the combinations the MMS architecture uses; run-time dilation (DIL = 0) otherwise
    // producer-wave path for every compile-time-dilation conv: with dwordx4 LDS-DMA it also wins for single-chunk inputs
    // (c_in = 32: 109 -> 120 TFLOP/s on the k = 11 layers; with dword DMA it lost 8 % there). VITS_DB_MIN=2 restores the
    // register-staged kernels for them.
    const int db_min = kernel_knobs().db_min;
    const bool db = w.nchunks >= db_min;
#ifdef VITS_MICRO_KT  // developer microbenchmark (tools/conv_micro.hip): instantiate a single (taps, dilation) pair
    VITS_GO(VITS_MICRO_KT, VITS_MICRO_DIL, EPI_STD);
#else
    if (w.epi == EPI_CONVT) {
        if (w.kt == 2) VITS_GO(2, -1, EPI_CONVT);
        return hipErrorInvalidValue;
    }
    if (w.epi == EPI_GATE) {
        if (w.kt == 5 && p.dil == 1) VITS_GO(5, 1, EPI_GATE);
        if (w.kt == 5) VITS_GO(5, 0, EPI_GATE);
        return hipErrorInvalidValue;
    }
    switch (w.kt) {
        case 1: VITS_GO(1, 1, EPI_STD);
        case 3: return launch_conv_k3(w, tile, p, ncols_max, batch, s, db);
        case 5:
            if (p.dil == 1) VITS_GO(5, 1, EPI_STD);
            VITS_GO(5, 0, EPI_STD);
        case 7: return launch_conv_k7(w, tile, p, ncols_max, batch, s, db);
        case 11: return launch_conv_k11(w, tile, p, ncols_max, batch, s, db);
        default: break;
    }
#endif
    return hipErrorInvalidValue;
}

double conv_flops(const PackedConv& w, const ConvCall&, int64_t total_cols) {
    // algorithmic MACs: every (row, col) output sums cin*kt products
    return 2.0 * (double)w.rows * (double)w.cin * (double)w.kt * (double)total_cols;
}
#endif  // VITS_CONV_PART == 0
#undef VITS_GO
#undef VITS_LAUNCHER

}  // namespace vits
