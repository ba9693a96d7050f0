// rbblock16.hip — one WHOLE HiFiGAN ResBlock (three conv pairs, dilations D0 / D1 / D2) as a single kernel, 16-bit-operand modes, the
// narrow vocoder stages (C = 32 / 64; C = 128 for k = 3):
//     y_{p+1} = y_p + Conv_{k,1}( leaky_relu( Conv_{k,D_p}( leaky_relu(y_p) ) + b1_p ) ) + b2_p ,  p = 0, 1, 2     (/root/reference/src/vits.cpp:545-581)
//     out     = [sum of the previous resblocks +] y_3 [ * 1/num_kernels ]                                             (vits.cpp:622-635)
// Why: in the 16-bit modes these stages are bound by HBM bytes (and, through the power budget, by the clock those bytes leave the matrix
// cores: tools/rb16_micro.hip — a C = 64, k = 3 pair takes 0.29 ms, 0.085 ms without its epilogue's traffic). As three fused pairs
// (rbpair16.hip) a resblock moves 3 x 12 B per element: every pair reads the 16-bit input and the fp32 residual and writes the fp32 stream
// and its 16-bit copy. Here the fp32 stream lives in REGISTERS across the three pairs (the MFMA C layout: a wave owns all 32 rows of a row
// tile for its 96 columns), the 16-bit conv inputs x_p = round(leaky_relu(y_p)) and t_p live in ONE LDS tile that the phases take turns
// in, and HBM sees the stage input once (4 B, + halo) and the resblock output once (4 B, + 4 B accumulator, + 2 B 16-bit copy on the last
// resblock): 10-14 B per element and resblock instead of 36. The price is the halo: a tile of W = 384 columns yields W - 24 (k - 1) / 2
// outputs (k = 3: 360, 7: 312, 11: 264), i.e. 1.07 / 1.23 / 1.45 x the MFMA work, which these stages have to spare (C = 64 runs 256-column
// tiles on four waves, 1.10 / 1.39 x, so that two blocks share a CU: see launch_rbblock16).
// Same operands, rounding points and k-order of accumulation (chunk, tap, k-half) as rbpair16_kernel / conv16_kernel: bit-identical to
// the pair path (GPU test), which stays for C >= 128 (MFMA-bound: the halo would cost more than the bytes) and behind VITS_NO_RBBLOCK16=1.
// Round 6: on grids of thousands of tiles a block walks a SEGMENT of tiles and takes the halo on its left from the tile before
// (template parameter STREAM; rbb_stream_tiles_for() says where): 1.19 x instead of 1.45 x the MFMA work at k = 11, which also brings the
// C = 64 / k = 11 resblocks to this kernel.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdlib>

#include "../../include/vits.h"
#include "kernels.h"

namespace vits {

namespace rbb {
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef int int4v __attribute__((ext_vector_type(4)));
typedef int int2v __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf2v __attribute__((ext_vector_type(2)));

template <bool BF>
__device__ __forceinline__ unsigned pack16(float a, float b) {
    float2v f = {a, b};
    if constexpr (BF) return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf2v));
    else return __builtin_bit_cast(unsigned, __builtin_convertvector(f, half2v));
}
// Two 8-byte halves of two 16-byte slots per lane -> one whole slot per lane. In the MFMA C layout lane l < 32 holds channels 0-3 and lane
// l + 32 channels 4-7 of a channel group at one column; writing them as two ds_write_b64 at a 16-byte stride is a 4-way bank conflict
// (PMC: 0.08-0.20 of the LDS cycles of the pair kernels). v_permlane32_swap exchanges the upper half of one register with the lower half of
// another: given groups g0 (a) and g1 (b), lanes < 32 end up with the whole slot of g0 and lanes >= 32 with the whole slot of g1.
__device__ __forceinline__ int4v slot_pair(int2v a, int2v b) {
    const auto x = __builtin_amdgcn_permlane32_swap((unsigned)a.x, (unsigned)b.x, false, false);
    const auto y = __builtin_amdgcn_permlane32_swap((unsigned)a.y, (unsigned)b.y, false, false);
    return int4v{(int)x[0], (int)y[0], (int)x[1], (int)y[1]};
}
}  // namespace rbb

struct RbBlockParams {
    const float* y0;  // stage input (the fp32 stream), group layout [b][C/8][g_ts][8]
    const uint16_t* w1[3];
    const uint16_t* w2[3];  // A fragments of the six convs (pack_conv_weights16)
    const float* b1[3];
    const float* b2[3];
    const int* lens;
    int tmax;
    float slope;  // leaky_relu in front of every conv
    float* yg;    // output, fp32 group layout (same strides as y0)
    const float* accg;
    int64_t g_bs;
    int g_ts;
    uint16_t* y16;
    int64_t y16_bs;
    int y16_ts;
    float y16_slope;
    float scale;
    int scale_div;
    int nt;  // STREAM: tiles a block walks (its segment = W - 2 H + (nt - 1) (W - H) output columns)
};

#ifdef VITS_PHASE_TIMING  // developer instrumentation (tools/rb16_micro.hip)
__device__ unsigned long long vits_rbb_phase[16 * 65536];
#define RBB_STAMP(k)                                                                                            \
    do {                                                                                                        \
        if (threadIdx.x == 0) {                                                                                 \
            const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;                                           \
            if (lin < 65536) vits_rbb_phase[16 * lin + (k)] = __builtin_amdgcn_s_memrealtime();                 \
        }                                                                                                       \
    } while (0)
#else
#define RBB_STAMP(k)
#endif

// Block = NSTRIP column strips x C / (32 MRW) row groups of waves; wave (strip, rg) owns the MRW row tiles [MRW rg, MRW rg + MRW) (32 rows each)
// of the NRW 32-column tiles of its strip.
// STREAM (round 6; large grids): the block WALKS a segment of p.nt tiles from left to right and the halo on the LEFT of every tile but the
// first is not recomputed: each of the six convs finds the columns in front of the tile — P2 D_p of x_p, P2 of t_p — where the previous
// tile left them (`hist`, copied out of the LDS tile while that stage was in it and into the padding in front of the tile when the stage is
// written again). A tile then yields W - H outputs instead of W - 2 H (k = 11 at W = 384: 1.19 x the algorithmic MFMA work instead of
// 1.45 x; k = 7: 1.10 / 1.23). The halo on the right stays: without it the fp32 stream in the registers would have to lag from pair to pair
// (a 32-column tile handed from wave to wave through LDS per pair: 16-28 KB, one block per CU fewer). Every element is the same chain of
// operations on the same operands whichever tile computes it, and a history column is the value its tile computed: bit-identical to the
// one-tile form (GPU test), which stays for grids too small to be cut into segments.
// (The body as a device function: rbblock16_kernel runs it for one resblock, rbblock16_group3_kernel for the three resblocks of a stage in ONE launch.)
template <int KT, int C, int NSTRIP, int NRW, int MRW, int D0, int D1, int D2, bool BF, bool STREAM>
__device__ __forceinline__ void rbblock16_body(const RbBlockParams& p, const int b) {
    using namespace rbb;
    constexpr int NCH = C / 32, W = NSTRIP * NRW * 32;  // (LDS tile: C / 8 channel groups x PITCH slots)
    constexpr int P2 = (KT - 1) / 2;
    constexpr int DMAX = D0 > D1 ? (D0 > D2 ? D0 : D2) : (D1 > D2 ? D1 : D2);
    constexpr int H = P2 * (3 + D0 + D1 + D2);  // halo per side: every pair costs P2 (second conv) + P2 * D_p (first conv)
    constexpr int BO = W - 2 * H;               // output columns of a (first) tile
    constexpr int ADV = W - H;                  // STREAM: distance between the tiles of a segment = output columns of every later tile
    constexpr int PADX = P2 * DMAX;             // the first conv of a pair reads up to P2 * D_p columns beyond a tile column
    constexpr int PITCH = (W + 2 * PADX + 7) / 8 * 8;
    constexpr int STEPS = 2 * KT, TOTAL = NCH * STEPS;
    constexpr int G = C / 8;
    static_assert(BO > 0, "tile too narrow for this kernel size");
    extern __shared__ __attribute__((aligned(16))) int4v tile[];  // [G][PITCH] slots of 8 x 16 bit: x_p, then t_p, then x_{p+1}, ...
    // behind the tile: the biases of the six convs, [pair][b1 | b2][C] fp32 — read from LDS where they are added (fetched from memory
    // behind each conv they cost an exposed L2 round trip per conv: six per block, which the short k = 3 blocks notice)
    float* lbias = reinterpret_cast<float*>(tile + G * PITCH);
    // STREAM: behind the biases the histories, [x_0 | t_0 | x_1 | t_1 | x_2 | t_2], stage q as [G][h_q] slots
    int4v* hist = tile + G * PITCH + 6 * C * (int)sizeof(float) / 16;

    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (wave-uniform: strip / row group live in scalar registers)
    const int strip = wid % NSTRIP, rt0 = (wid / NSTRIP) * MRW;
    const int len = p.lens ? p.lens[b] : p.tmax;
    const int nt = STREAM ? p.nt : 1;
    const int seg0 = blockIdx.x * (BO + (nt - 1) * ADV);
    if (seg0 >= len) return;
    RBB_STAMP(0);
    typedef const __attribute__((address_space(3))) int4v* LdsV;
    typedef __attribute__((address_space(3))) int4v* LdsS;
    // (the biases: requested in front of the first stream loads, so that the two round trips overlap)
    for (int i = threadIdx.x; i < 6 * C; i += C / (32 * MRW) * NSTRIP * 64) {  // (the body's own thread count: in a group launch the block may be larger)
        const int pi = i / (2 * C), r = i - pi * 2 * C;
        lbias[i] = r < C ? p.b1[pi][r] : p.b2[pi][r - C];
    }
    constexpr int HX0 = 0, HT0 = HX0 + P2 * D0, HX1 = HT0 + P2, HT1 = HX1 + P2 * D1, HX2 = HT1 + P2, HT2 = HX2 + P2 * D2;  // column offsets of the histories
    static_assert(HT2 + P2 == H, "the histories are the halo of one side");
    static_assert(!STREAM || G * P2 * DMAX <= C / (32 * MRW) * NSTRIP * 64, "one history slot per thread");

  for (int it = 0; it < nt; ++it) {
    // (STREAM: every per-lane address below is derived from an opaque copy of the lane id — left to itself the compiler hoists the
    // invariant halves of ~50 addresses out of the tile loop and spills 120 registers to keep them)
    int lane_v = lane;
    if constexpr (STREAM) asm volatile("" : "+v"(lane_v));
    const int h = lane_v >> 5, col = lane_v & 31;
    const int tid = wid * 64 + lane_v;
    const int u0 = strip * (NRW * 32) + col;  // this lane's tile column of column tile nr: u0 + 32 nr
    const int tg0 = seg0 - H + it * ADV;  // global time of tile column 0
    // history of a stage: `hq` columns in front of the NEXT tile (tile columns [ADV - hq, ADV)) out of the LDS tile, once the stage is
    // complete in it; and back in front of the tile (slots [PADX - hq, PADX)) when the next tile writes that stage
    auto hist_save = [&](const int off, const int hq) __attribute__((always_inline)) {
        if constexpr (STREAM) {
            if (tid < G * hq) {
                const int g = tid / hq, c = tid - g * hq;
                const int4v v = *((LdsV)(tile + g * PITCH + PADX + ADV - hq + c));
                *((LdsS)(hist + G * off + tid)) = v;
            }
        }
    };
    auto hist_restore = [&](const int off, const int hq, const bool on) __attribute__((always_inline)) {
        if constexpr (STREAM) {
            if (on && tid < G * hq) {
                const int g = tid / hq, c = tid - g * hq;
                const int4v v = *((LdsV)(hist + G * off + tid));
                *((LdsS)(tile + g * PITCH + PADX - hq + c)) = v;
            }
        }
    };
    const int ulo = it == 0 ? H : 0;      // the tile's own outputs: columns [ulo, ADV)
    if (tg0 + ulo >= len) break;
    if (it > 0) __syncthreads();  // every wave is done with the previous tile's t_2

    // ---- the fp32 stream of this wave's rows x columns, in the MFMA C layout: register 4 g + e of yv[m][nr] = channel 32 (rt0 + m) + 8 g + 4 h + e ----
    floatx16 yv[MRW][NRW];
    {
        const float* yb = p.y0 + (int64_t)b * p.g_bs;
#pragma unroll
        for (int m = 0; m < MRW; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch0 = (rt0 + m) * 32 + 8 * g + 4 * h;
#pragma unroll
                for (int nr = 0; nr < NRW; ++nr) {
                    const int t = tg0 + u0 + 32 * nr;
                    float4v v = {0.f, 0.f, 0.f, 0.f};
                    if (t >= 0 && t < len) v = *reinterpret_cast<const float4v*>(yb + ((int64_t)(ch0 >> 3) * p.g_ts + t) * 8 + (ch0 & 7));
#pragma unroll
                    for (int e = 0; e < 4; ++e) yv[m][nr][4 * g + e] = v[e];
                }
            }
    }
    // round(leaky_relu(src + bias)) of this wave's rows x columns into the LDS tile as whole 16-byte slots, zero outside the sequence (what a
    // conv sees as padding): x_p from the stream (bias = nullptr), t_p from the first conv's accumulators
    auto write_tile = [&](const floatx16 (&src)[MRW][NRW], const float* bias_p) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < MRW; ++m)
#pragma unroll
            for (int k = 0; k < 2; ++k) {  // channel groups 2k and 2k + 1 of the row tile: one slot each per lane after the half swap
                float4v bias[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                if (bias_p) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) bias[q] = *reinterpret_cast<const float4v*>(bias_p + (rt0 + m) * 32 + 8 * (2 * k + q) + 4 * h);
                }
#pragma unroll
                for (int nr = 0; nr < NRW; ++nr) {
                    const int u = u0 + 32 * nr, t = tg0 + u;
                    int2v w[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = src[m][nr][4 * (2 * k + q) + e];
                            if (bias_p) v[e] = v[e] + bias[q][e];
                            v[e] = fmaxf(v[e], v[e] * p.slope);
                            if (t < 0 || t >= len) v[e] = 0.f;
                        }
                        w[q].x = (int)pack16<BF>(v[0], v[1]);
                        w[q].y = (int)pack16<BF>(v[2], v[3]);
                    }
                    *((LdsS)(tile + ((rt0 + m) * 4 + 2 * k + h) * PITCH + PADX + u)) = slot_pair(w[0], w[1]);
                }
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    auto mfma = [&](int4v a, int4v bq, floatx16 c) __attribute__((always_inline)) -> floatx16 {
        if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bq), c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, bq), c, 0, 0, 0);
    };
    floatx16 acc[MRW][NRW];
    // one conv over the LDS tile: output column u reads slots u + off0 + j * dstep; acc = sum over (chunk, tap, k-half) — the order of
    // rbpair16_kernel / conv16_kernel
    auto conv = [&](const uint16_t* wp, const int off0, const int dstep) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < MRW; ++m)
#pragma unroll
            for (int j = 0; j < NRW; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;
        LdsV base = (LdsV)(tile + h * PITCH + PADX + u0 + off0);
        const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(wp), 0, 0x7fffffff, 0x00020000);
        int wvoff[MRW];
#pragma unroll
        for (int m = 0; m < MRW; ++m) wvoff[m] = (int)(((size_t)(rt0 + m) * TOTAL * 64 + lane_v) * 16);
        auto load_a = [&](int m, int step) __attribute__((always_inline)) -> int4v {
            return __builtin_bit_cast(int4v, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff[m], step * 1024, 0));
        };
        // weight-fragment ring: a step is NRW 32-cycle MFMAs per row tile, an L2 round trip several steps (C = 32, k >= 7: six slots — eight spill at the 168-VGPR cap of three blocks per CU)
        constexpr int RS = MRW == 1 ? ((C == 32 && KT > 3) ? 6 : 8) : 4, RD = RS - 2;
        int4v ring[RS][MRW];
#pragma unroll
        for (int i = 0; i < RD; ++i)
#pragma unroll
            for (int m = 0; m < MRW; ++m) ring[i][m] = load_a(m, i < TOTAL ? i : TOTAL - 1);
        // chunk / tap / k-half nest with a per-chunk operand pointer (the loop shape of rbpair16_kernel: a flat step loop with absolute
        // offsets compiled to more VGPRs there)
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            LdsV xb = base + c * 4 * PITCH;
            int4v b_nxt[NRW];
#pragma unroll
            for (int nr = 0; nr < NRW; ++nr) b_nxt[nr] = xb[nr * 32];
#pragma unroll
            for (int j = 0; j < KT; ++j)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int s = c * STEPS + j * 2 + kk;  // compile time after unrolling
#pragma unroll
                    for (int m = 0; m < MRW; ++m) ring[(s + RD) % RS][m] = load_a(m, s + RD < TOTAL ? s + RD : TOTAL - 1);
                    __builtin_amdgcn_sched_barrier(0);
                    int4v b_cur[NRW];
#pragma unroll
                    for (int nr = 0; nr < NRW; ++nr) b_cur[nr] = b_nxt[nr];
                    {
                        // next step: the other k-half of this tap, or k-half 0 of the next tap (past the last tap: a slot a little further on, unused)
                        const int noff = kk == 0 ? 2 * PITCH + j * dstep : (j + 1) * dstep;
#pragma unroll
                        for (int nr = 0; nr < NRW; ++nr) b_nxt[nr] = xb[noff + nr * 32];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < MRW; ++m)
#pragma unroll
                        for (int nr = 0; nr < NRW; ++nr) acc[m][nr] = mfma(ring[s % RS][m], b_cur[nr], acc[m][nr]);
                }
        }
    };

    // (hx / ht: history offsets of x_p and t_p; hxn, dn: offset and dilation of the next pair's x)
    auto pair = [&](const int pi, const int dil, const bool last, const int hx, const int ht, const int hxn, const int dn) __attribute__((always_inline)) {
        hist_save(hx, P2 * dil);  // (x_p is complete in the tile: what the next tile's first conv reads in front of its columns)
        // conv 1 over x_p: t column u reads x columns u - P2 dil + j dil
        conv(p.w1[pi], -P2 * dil, dil);
        __syncthreads();  // every wave is done with x_p: t_p takes its place
        write_tile(acc, lbias + pi * 2 * C);  // t = round(leaky_relu(conv1 + b1)), zero outside the sequence (the second conv's padding)
        hist_restore(ht, P2, it > 0);
        __syncthreads();
        hist_save(ht, P2);
        // conv 2 over t_p: y column u reads t columns u - P2 + j
        conv(p.w2[pi], -P2, 1);
        // the stream: y_{p+1} = y_p + (conv2 + b2)
#pragma unroll
        for (int m = 0; m < MRW; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch0 = (rt0 + m) * 32 + 8 * g + 4 * h;
                const float4v bias = *reinterpret_cast<const float4v*>(lbias + pi * 2 * C + C + ch0);
#pragma unroll
                for (int nr = 0; nr < NRW; ++nr)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = acc[m][nr][4 * g + e] + bias[e];
                        yv[m][nr][4 * g + e] = yv[m][nr][4 * g + e] + v;
                    }
            }
        if (!last) {
            __syncthreads();  // every wave is done with t_p: x_{p+1} takes its place
            write_tile(yv, nullptr);
            hist_restore(hxn, P2 * dn, it > 0);
            __syncthreads();
        }
    };

    write_tile(yv, nullptr);
    hist_restore(HX0, P2 * D0, it > 0);
    __syncthreads();
    RBB_STAMP(1);
    pair(0, D0, false, HX0, HT0, HX1, D1);
    RBB_STAMP(2);
    pair(1, D1, false, HX1, HT1, HX2, D2);
    RBB_STAMP(3);
    pair(2, D2, true, HX2, HT2, 0, 1);
    RBB_STAMP(4);

    // ---- epilogue (as the last pair's in rbpair16_kernel): resblock sum / scale, fp32 output + 16-bit copy, the BO owned columns only ----
    // Two passes: the accumulator of the previous resblocks is read in place (accg == yg), so a load / add / store per element is a chain of
    // dependent HBM round trips — the compiler may not move a later load above an earlier store that could alias it (the WaveNet kernels
    // spent 15 of 48 us that way). All addends first (the conv accumulators are dead: their registers hold them), then every store.
    {
        // (STREAM: addresses from a second opaque copy of the lane id — shared with the loads at the top of the tile they would be kept, and spilled, across the six convs)
        int lane_e = lane;
        if constexpr (STREAM) asm volatile("" : "+v"(lane_e));
        const int h = lane_e >> 5, u0 = strip * (NRW * 32) + (lane_e & 31);
        float* yg = p.yg ? p.yg + (int64_t)b * p.g_bs : nullptr;  // (null: only the 16-bit copy is wanted — the last resblock of a stage)
        const float* ag = p.accg ? p.accg + (int64_t)b * p.g_bs : nullptr;
        uint16_t* y16 = p.y16 ? p.y16 + (int64_t)b * p.y16_bs : nullptr;
        float4v av[MRW][4][NRW];
        __builtin_amdgcn_sched_barrier(0);  // (not above the last conv: its accumulators have to be dead for these registers)
        if constexpr (STREAM) {  // (defined on every path: left undefined without an accumulator they become loop-carried values of the tile loop, 40 spilled registers)
#pragma unroll
            for (int m = 0; m < MRW; ++m)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int nr = 0; nr < NRW; ++nr) av[m][g][nr] = float4v{0.f, 0.f, 0.f, 0.f};
        }
        if (ag) {
#pragma unroll
            for (int m = 0; m < MRW; ++m)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch0 = (rt0 + m) * 32 + 8 * g + 4 * h;
#pragma unroll
                    for (int nr = 0; nr < NRW; ++nr) {
                        const int u = u0 + 32 * nr, t = tg0 + u;
                        av[m][g][nr] = float4v{0.f, 0.f, 0.f, 0.f};
                        if (u >= ulo && u < ADV && t < len) av[m][g][nr] = *reinterpret_cast<const float4v*>(ag + ((int64_t)(ch0 >> 3) * p.g_ts + t) * 8 + (ch0 & 7));
                    }
                }
        }
#pragma unroll
        for (int m = 0; m < MRW; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch0 = (rt0 + m) * 32 + 8 * g + 4 * h;
#pragma unroll
                for (int nr = 0; nr < NRW; ++nr) {
                    const int u = u0 + 32 * nr, t = tg0 + u;
                    if (u < ulo || u >= ADV || t >= len) continue;
                    const int64_t go = ((int64_t)(ch0 >> 3) * p.g_ts + t) * 8 + (ch0 & 7);
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = yv[m][nr][4 * g + e];
                    if (ag) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = av[m][g][nr][e] + v[e];
                            v[e] = p.scale_div ? v[e] / p.scale : v[e] * p.scale;
                        }
                    }
                    if (yg) *reinterpret_cast<float4v*>(yg + go) = float4v{v[0], v[1], v[2], v[3]};
                    if (y16) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * p.y16_slope);
                        int2v w2;
                        w2.x = (int)pack16<BF>(v[0], v[1]);
                        w2.y = (int)pack16<BF>(v[2], v[3]);
                        *reinterpret_cast<int2v*>(y16 + ((int64_t)(ch0 >> 3) * p.y16_ts + t) * 8 + (ch0 & 7)) = w2;
                    }
                }
            }
    }
  }  // (tiles of the segment)
    RBB_STAMP(5);
}
template <int KT, int C, int NSTRIP, int NRW, int MRW, int D0, int D1, int D2, bool BF, bool STREAM>
__global__ __launch_bounds__(C / (32 * MRW) * NSTRIP * 64, (C == 32 && NSTRIP == 4 && NRW <= 3) ? 3 : (C / (32 * MRW) * NSTRIP <= 4 ? 2 : 1)) void rbblock16_kernel(const RbBlockParams p) {
    rbblock16_body<KT, C, NSTRIP, NRW, MRW, D0, D1, D2, BF, STREAM>(p, (int)blockIdx.y);
}
// The three resblocks of a narrow stage (k = 3, 7, 11; equal block shapes) in ONE launch: blockIdx.y = 3 x utterance + member; blocks past a member's last
// tile leave at once. One or two utterances, side-by-side resblocks (launch_rb_sum3 adds their outputs): one launch on the main stream instead of three on three
// streams behind a fork and in front of a join of 10-30 us each (the C = 32 stage at batch 1: 63 -> see DESIGN 9). A member is the kernel's body.
struct RbBlockGroup3Params {
    RbBlockParams m[3];
};
// member shapes (column strips, 32-column tiles per wave): those of launch_rbblock16 — C = 32: 4 x 3 for every k; C = 64: 2 x 4 for k = 3 / 7 (four waves), 4 x 3 for
// k = 11 (eight waves: the block is eight waves and a four-wave member's upper four leave before its first barrier)
template <int C, bool BF>
__global__ __launch_bounds__(C == 32 ? 256 : 512, C == 32 ? 3 : 2) void rbblock16_group3_kernel(const RbBlockGroup3Params gp) {
    const int member = (int)blockIdx.y % 3, b = (int)blockIdx.y / 3;
    if constexpr (C == 32) {
        if (member == 0) rbblock16_body<3, C, 4, 3, 1, 1, 3, 5, BF, false>(gp.m[0], b);
        else if (member == 1) rbblock16_body<7, C, 4, 3, 1, 1, 3, 5, BF, false>(gp.m[1], b);
        else rbblock16_body<11, C, 4, 3, 1, 1, 3, 5, BF, false>(gp.m[2], b);
    } else {
        if (member == 2) {
            rbblock16_body<11, C, 4, 3, 1, 1, 3, 5, BF, false>(gp.m[2], b);
        } else {
            if (threadIdx.x >= 256) return;
            if (member == 0) rbblock16_body<3, C, 2, 4, 1, 1, 3, 5, BF, false>(gp.m[0], b);
            else rbblock16_body<7, C, 2, 4, 1, 1, 3, 5, BF, false>(gp.m[1], b);
        }
    }
}

// ---- the sum over a stage's resblocks as its own launch (small grids, round 6) ------------------------------------------------------------------------
// The whole-resblock kernel carries the accumulation into the stage's shared sum itself, so the three resblocks of a narrow stage are a CHAIN of launches in
// the reference's order of the additions (RB0, += RB1, (+= RB2) * 1/num_kernels: vits.cpp:622-635) — at batch 1 three 15-25 us kernels of 75-100 blocks and two
// event hand-overs one behind the other (105 us for the C = 32 stage) on a chip that could run them side by side. Here every resblock writes its own fp32
// output and this kernel adds them in that same order with the same expressions (a + v, then the scale, then leaky_relu and the rounding of the 16-bit copy
// the next upsampler / conv_post reads): the same bits as the chained accumulation, element for element.
template <bool BF>
__global__ __launch_bounds__(256) void rb_sum3_kernel(const float* y0, const float* y1, const float* y2, int64_t g_bs, int g_ts, const int* lens, int tmax, float scale,
                                                       int scale_div, float* yg, uint16_t* y16, int64_t y16_bs, int y16_ts, float y16_slope) {
    const int b = blockIdx.z, grp = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    const int len = lens ? lens[b] : tmax;
    if (t >= len) return;
    const int64_t go = (int64_t)b * g_bs + ((int64_t)grp * g_ts + t) * 8;
    float v[8];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const rbb::float4v a = *reinterpret_cast<const rbb::float4v*>(y0 + go + 4 * hh), b1 = *reinterpret_cast<const rbb::float4v*>(y1 + go + 4 * hh);
        rbb::float4v c2 = {0.f, 0.f, 0.f, 0.f};
        if (y2) c2 = *reinterpret_cast<const rbb::float4v*>(y2 + go + 4 * hh);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = a[e] + b1[e];  // (the second resblock's a + v; its scale is 1: v * 1 = v)
            if (y2) x = x + c2[e];
            x = scale_div ? x / scale : x * scale;
            v[4 * hh + e] = x;
        }
    }
    if (yg) {
        *reinterpret_cast<rbb::float4v*>(yg + go) = rbb::float4v{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<rbb::float4v*>(yg + go + 4) = rbb::float4v{v[4], v[5], v[6], v[7]};
    }
    if (y16) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], v[e] * y16_slope);
        rbb::int4v w;
        w.x = (int)rbb::pack16<BF>(v[0], v[1]);
        w.y = (int)rbb::pack16<BF>(v[2], v[3]);
        w.z = (int)rbb::pack16<BF>(v[4], v[5]);
        w.w = (int)rbb::pack16<BF>(v[6], v[7]);
        *reinterpret_cast<rbb::int4v*>(y16 + (int64_t)b * y16_bs + ((int64_t)grp * y16_ts + t) * 8) = w;
    }
}

hipError_t launch_rb_sum3(const float* y0, const float* y1, const float* y2, int channels, int64_t g_bs, int g_ts, const int* lens, int batch, int tmax, float scale, int scale_div,
                          float* yg, Ref16 y16, float y16_slope, int arith, hipStream_t s) {
    if (!y0 || !y1 || (channels & 7) || (!yg && !y16.p)) return hipErrorInvalidValue;
    dim3 grid((tmax + 255) / 256, channels / 8, batch);
    if (arith == VITS_ARITH_BF16) VITS_KLAUNCH(rb_sum3_kernel<true>, grid, dim3(256), 0, s, y0, y1, y2, g_bs, g_ts, lens, tmax, scale, scale_div, yg, y16.p, y16.bs, y16.ts, y16_slope);
    else VITS_KLAUNCH(rb_sum3_kernel<false>, grid, dim3(256), 0, s, y0, y1, y2, g_bs, g_ts, lens, tmax, scale, scale_div, yg, y16.p, y16.bs, y16.ts, y16_slope);
    return hipGetLastError();
}

// ---- host side -----------------------------------------------------------------------------------------------------------
template <int KT, int C, int NSTRIP, int NRW, int MRW, bool BF, bool STREAM>
static hipError_t launch_rbb_grid(const RbBlockParams& p, int seg_out, int batch, hipStream_t s) {
    constexpr int D0 = 1, D1 = 3, D2 = 5;
    constexpr int P2 = (KT - 1) / 2, W = NSTRIP * NRW * 32, H = P2 * (3 + D0 + D1 + D2), PADX = P2 * D2, PITCH = (W + 2 * PADX + 7) / 8 * 8;
    const size_t lds = (size_t)(C / 8) * PITCH * 16 + (size_t)6 * C * sizeof(float) + (STREAM ? (size_t)(C / 8) * H * 16 : 0);
    static BigLdsOnce big_lds_set;
    if (lds > 64 * 1024 && big_lds_set.needed()) {
        hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbblock16_kernel<KT, C, NSTRIP, NRW, MRW, D0, D1, D2, BF, STREAM>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (ea != hipSuccess) return ea;
        big_lds_set.done();
    }
    dim3 grid((p.tmax + seg_out - 1) / seg_out, batch);
    VITS_KLAUNCH((rbblock16_kernel<KT, C, NSTRIP, NRW, MRW, D0, D1, D2, BF, STREAM>), grid, dim3(C / (32 * MRW) * NSTRIP * 64), lds, s, p);
    return hipGetLastError();
}
// Tiles per block for a launch of `batch` sequences of up to `tmax` columns on tiles of W columns.
// Segments of several tiles (STREAM) once the one-tile grid is many rounds of the chip: a segment's first tile pays the halo of both sides,
// and the blocks of the last round run on a partly empty chip — nt grows with the grid up to the shape's limit.
// Which shapes: measured per kernel on the benchmark batch (64 x 128 ids: 10-14 thousand tiles, i.e. 13-18 per resident block, so that long
// segments cost in balance what they save in halo — tools/rbb_micro.hip, profiles/round6_rbb_stream_micro.txt): k = 11 at C = 32
// 1.000 -> 0.892 ms with 6 tiles, k = 7 at C = 64 1.138 -> 1.040 with 3, k = 11 at C = 64 1.584 -> 1.396 with 4; k = 3 (a halo of 12
// columns) and k = 7 at C = 32 lose 0-25 %: one tile per block there.
static int rbb_stream_tiles_for(int kt, int C, int W, int batch, int tmax) {
    const KernelKnobs& kn = kernel_knobs();
    const int H = (kt - 1) / 2 * 12, BO = W - 2 * H;
    const long blocks1 = (long)((tmax + BO - 1) / BO) * batch;
    int want = kn.rbb_stream_tiles;
    if (want < 0) want = (kt == 11 && C == 32) ? 6 : (kt == 7 && C == 64) ? 3 : (kt == 11 && C == 64) ? 4 : 0;
    if (want <= 1 || kn.rbb_stream_min_blocks <= 0 || blocks1 < 2L * kn.rbb_stream_min_blocks) return 1;
    return (int)(blocks1 / kn.rbb_stream_min_blocks < want ? blocks1 / kn.rbb_stream_min_blocks : want);
}
template <int KT, int C, int NSTRIP, int NRW, int MRW, bool BF>
static hipError_t launch_rbb(RbBlockParams p, int batch, hipStream_t s) {
    constexpr int P2 = (KT - 1) / 2, W = NSTRIP * NRW * 32, H = P2 * 12, BO = W - 2 * H, ADV = W - H;
    const int nt = rbb_stream_tiles_for(KT, C, W, batch, p.tmax);
    p.nt = nt;
    if (nt > 1) return launch_rbb_grid<KT, C, NSTRIP, NRW, MRW, BF, true>(p, BO + (nt - 1) * ADV, batch, s);
    return launch_rbb_grid<KT, C, NSTRIP, NRW, MRW, BF, false>(p, BO, batch, s);
}

bool rbblock16_supported(int channels, int kt, const int* dils, int ndil, int batch, int tmax) {
    const bool c128 = kernel_knobs().rbb_c128;
    if (channels == 128) return c128 && kt == 3 && ndil == 3 && dils[0] == 1 && dils[1] == 3 && dils[2] == 5;
    if (!(channels == 32 || channels == 64) || !(kt == 3 || kt == 7 || kt == 11)) return false;
    // C = 64, k = 11: on one tile per block (1.45 x the MFMA work) the whole-resblock kernel is bound by the matrix cores (at the clock
    // the power budget leaves them) and loses to three fused pairs, 1.59 against 1.45 ms per step (batch 64 x 128 ids); on segments of four
    // tiles (1.19 x; round 6) it wins, 13.31 -> 13.19 ms per pipelined batch. VITS_RBB_C64K11=1 runs it on every grid, =0 on none.
    const int c64k11 = kernel_knobs().rbb_c64k11;
    if (channels == 64 && kt == 11 && !(c64k11 > 0 || (c64k11 < 0 && rbb_stream_tiles_for(11, 64, 384, batch, tmax) > 1))) return false;
    return ndil == 3 && dils[0] == 1 && dils[1] == 3 && dils[2] == 5;
}

static hipError_t rbb_params(const PackedConv* const* c1, const PackedConv* const* c2, const RbBlock16Call& c, RbBlockParams& p) {
    const int C = c1[0]->cin, kt = c1[0]->kt;
    for (int i = 0; i < 3; ++i) {
        if (!c1[i]->wp16 || !c2[i]->wp16 || !c1[i]->bias || !c2[i]->bias || c1[i]->cin != C || c1[i]->cout != C || c2[i]->cin != C || c2[i]->cout != C || c1[i]->kt != kt ||
            c2[i]->kt != kt)
            return hipErrorInvalidValue;
        p.w1[i] = c1[i]->wp16;
        p.w2[i] = c2[i]->wp16;
        p.b1[i] = c1[i]->bias;
        p.b2[i] = c2[i]->bias;
    }
    const int dils[3] = {1, 3, 5};
    if (!rbblock16_supported(C, kt, dils, 3, c.batch, c.tmax) || !c.y0 || (!c.yg && !c.y16.p)) return hipErrorInvalidValue;
    p.y0 = c.y0;
    p.lens = c.lens;
    p.tmax = c.tmax;
    p.slope = c.slope;
    p.yg = c.yg;
    p.accg = c.accg;
    p.g_bs = c.g_bs;
    p.g_ts = c.g_ts;
    p.y16 = c.y16.p;
    p.y16_bs = c.y16.bs;
    p.y16_ts = c.y16.ts;
    p.y16_slope = c.y16_slope;
    p.scale = c.scale;
    p.scale_div = c.scale_div;
    p.nt = 1;
    return hipSuccess;
}

// the three resblocks (k = 3, 7, 11) of a C = 32 stage as one launch (small grids: one tile per block); c1[m] / c2[m]: member m's three conv pairs
bool rbblock16_group3_supported(int channels, const int* kts, int batch, int tmax) {
    if (kernel_knobs().no_rbb_group3 || !(channels == 32 || (channels == 64 && !kernel_knobs().no_rbb_group3_c64)) || kts[0] != 3 || kts[1] != 7 || kts[2] != 11) return false;
    const int dils[3] = {1, 3, 5};
    for (int m = 0; m < 2; ++m)
        if (!rbblock16_supported(channels, kts[m], dils, 3, batch, tmax)) return false;
    // (k = 11 at C = 64 is a whole-resblock kernel here whatever VITS_RBB_C64K11 says for the single launches: measured in the group)
    if (channels == 32 && !rbblock16_supported(channels, 11, dils, 3, batch, tmax)) return false;
    return (long)((tmax + 183) / 184) * batch <= 3072;  // (one tile per block: well below rbb_stream_tiles_for's threshold for segments)
}
hipError_t launch_rbblock16_group3(const PackedConv* const (*c1)[3], const PackedConv* const (*c2)[3], const RbBlock16Call* c, int arith, hipStream_t s) {
    RbBlockGroup3Params gp;
    const int C = c1[0][0]->cin;
    const int kts[3] = {c1[0][0]->kt, c1[1][0]->kt, c1[2][0]->kt};
    if (!rbblock16_group3_supported(C, kts, c[0].batch, c[0].tmax)) return hipErrorInvalidValue;
    for (int m = 0; m < 3; ++m) {
        if (C == 64 && m == 2) {  // (rbb_params asks rbblock16_supported, which answers for the single launch)
            KernelKnobs k2 = kernel_knobs();
            k2.rbb_c64k11 = 1;
            KernelKnobsScope scope(&k2);
            if (hipError_t e = rbb_params(c1[m], c2[m], c[m], gp.m[m])) return e;
        } else if (hipError_t e = rbb_params(c1[m], c2[m], c[m], gp.m[m])) return e;
        if (c[m].batch != c[0].batch || c[m].tmax != c[0].tmax) return hipErrorInvalidValue;
    }
    // LDS of the k = 11 member (384 columns + the widest padding); grid.x of the member with the fewest outputs per tile (C = 32: k = 11, 264; C = 64: k = 7 on 256 columns, 184)
    const int PITCH = (384 + 2 * 25 + 7) / 8 * 8, BOmin = C == 32 ? 264 : 184;
    const size_t lds = (size_t)(C / 8) * PITCH * 16 + (size_t)6 * C * sizeof(float);
    dim3 grid((c[0].tmax + BOmin - 1) / BOmin, 3 * c[0].batch);
    const bool bf = arith == VITS_ARITH_BF16;
    if (C == 32) {
        if (bf) VITS_KLAUNCH((rbblock16_group3_kernel<32, true>), grid, dim3(256), lds, s, gp);
        else VITS_KLAUNCH((rbblock16_group3_kernel<32, false>), grid, dim3(256), lds, s, gp);
    } else {
        if (bf) VITS_KLAUNCH((rbblock16_group3_kernel<64, true>), grid, dim3(512), lds, s, gp);
        else VITS_KLAUNCH((rbblock16_group3_kernel<64, false>), grid, dim3(512), lds, s, gp);
    }
    return hipGetLastError();
}

hipError_t launch_rbblock16(const PackedConv* const* c1, const PackedConv* const* c2, const RbBlock16Call& c, int arith, hipStream_t s) {
    const int C = c1[0]->cin, kt = c1[0]->kt;
    RbBlockParams p;
    if (hipError_t e = rbb_params(c1, c2, c, p)) return e;
    const bool bf = arith == VITS_ARITH_BF16;
    // tile shape (column strips x 32-column tiles per wave): C = 32: 4 x 3 = 384 columns, three blocks per CU. Measured alternatives (batch
    // 64 x 128 ids, f16): C = 32 with 4 x 4 = 512 columns (two blocks per CU instead of three) +5...12 %; C = 64 as 4 x 3 (eight waves, one
    // block per CU: round 3's first version), as 6 x 2 (twelve waves) +-0, as 8 x 2 = 512 columns (sixteen waves at 128 VGPRs, spills) -5 %
    // on k = 11 only, as 2 x 5 (spills) worse than 2 x 4.
#define VITS_RBB_GO(K, CC, NS, NRW_, MRW_)                                                              \
    if (kt == K && C == CC) return bf ? launch_rbb<K, CC, NS, NRW_, MRW_, true>(p, c.batch, s) : launch_rbb<K, CC, NS, NRW_, MRW_, false>(p, c.batch, s)
    VITS_RBB_GO(3, 32, 4, 3, 1);
    VITS_RBB_GO(7, 32, 4, 3, 1);
    VITS_RBB_GO(11, 32, 4, 3, 1);
    // C = 64: four waves (two strips of four column tiles, 256 columns) instead of eight (4 x 3): at 236 VGPRs two blocks share a CU and
    // one's loads, tile writes and epilogue overlap the other's MFMA phases — with eight waves a CU ran ONE block at a time, its matrix
    // pipes 30 % busy at 1.4 TB/s: k = 3 0.70 -> 0.55 ms, k = 7 1.09 -> 1.05 (the narrower tile costs 1.39 x instead of 1.23 x the MFMA work there)
    VITS_RBB_GO(3, 64, 2, 4, 1);
    VITS_RBB_GO(7, 64, 2, 4, 1);
    VITS_RBB_GO(11, 64, 4, 3, 1);
    // C = 128, k = 3: the pairs are HBM-bound (4.4 TB/s); eight waves of two row tiles x two column tiles (256-column tiles, 232 outputs).
    // (Eight waves of one row tile on 128-column tiles at 128 VGPRs — two blocks per CU —, sixteen waves on 256 columns, eight on 192: +-2 %.)
    VITS_RBB_GO(3, 128, 4, 2, 2);
#undef VITS_RBB_GO
    return hipErrorInvalidValue;
}

}  // namespace vits
