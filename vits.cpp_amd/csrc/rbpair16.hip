// rbpair16.hip — one HiFiGAN ResBlock conv PAIR as a single kernel, 16-bit-operand modes, narrow stages (C = 32 / 64; 128 opt-in):
//     y' = y + Conv_{k,1}( leaky_relu( Conv_{k,d}( leaky_relu(y) ) + b1 ) ) + b2        (/root/reference/src/vits.cpp:545-581)
// The intermediate t = leaky_relu(conv1 + b1) never goes to HBM: the block computes a 256-column tile of it (the output tile
// plus the (k-1)-column halo the second conv needs), rounds it to the arithmetic type exactly as the unfused path's epilogue
// does, keeps it in LDS in the group layout, and runs the second conv from there. Per element and pair that is
// read x16 (2 B) + residual (4) + write y' (4) + its 16-bit copy (2) = 12 B instead of 16 B — these stages run at the HBM
// roof in the 16-bit modes (conv16.hip), so bytes are time. Same operands, same k-order of accumulation
// (chunk, tap, k-half) and same rounding points as conv16_kernel: the results are bit-identical to the two-kernel path
// (GPU test), which the engine keeps for C >= 128 and as the VITS_NO_FUSE16=1 fallback.
//
// Block = 4 waves, no producer wave: the whole input tile (C/8 groups x (256 + (k-1)(d+1)) slots) is streamed in with LDS-DMA
// by all four waves at once, then conv1 -> t tile (LDS) -> conv2 -> epilogue; 3-4 blocks per CU overlap one block's DMA and
// epilogue traffic with another's MFMAs.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "../../include/vits.h"
#include "kernels.h"

namespace vits {

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
__device__ __forceinline__ unsigned rb_pack16(float a, float b) {
    float2v f = {a, b};
    if constexpr (BF) return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf2v));
    else return __builtin_bit_cast(unsigned, __builtin_convertvector(f, half2v));
}

#ifdef VITS_PHASE_TIMING  // developer instrumentation (tools/rb16_micro.hip): per-block phase timestamps
__device__ unsigned long long vits_rb_phase[16 * 65536];  // [block][0..6] 100 MHz stamps, [7] HW_ID, [8] XCC_ID, [9..10] shader clock around conv1
#define RB_STAMP(k)                                                                                     \
    do {                                                                                                \
        if (threadIdx.x == 0) {                                                                         \
            const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;                                   \
            if (lin < 65536) {                                                                          \
                vits_rb_phase[16 * lin + (k)] = __builtin_amdgcn_s_memrealtime();                       \
                if ((k) == 1 || (k) == 2) vits_rb_phase[16 * lin + 8 + (k)] = __builtin_amdgcn_s_memtime(); \
                if ((k) == 0) {                                                                         \
                    unsigned hw, xcc;                                                                   \
                    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                   \
                    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                 \
                    vits_rb_phase[16 * lin + 7] = hw;                                                   \
                    vits_rb_phase[16 * lin + 8] = xcc;                                                  \
                }                                                                                       \
            }                                                                                           \
        }                                                                                               \
    } while (0)
#else
#define RB_STAMP(k)
#endif

struct RbPairParams {
    const uint16_t* x;  // leaky_relu(y), rounded: group layout [b][C/8][x_ts][8]
    int64_t x_bs;
    int x_ts;
    const uint16_t *w1, *w2;  // A fragments of the two convs (pack_conv_weights16)
    const float *b1, *b2;
    const int* lens;
    int tmax;
    float slope;  // leaky_relu between the convs
    // epilogue (group layout, as conv16's E16_GROUP)
    float* yg;
    const float* resg;
    const float* accg;
    int64_t g_bs;
    int g_ts;
    uint16_t* y16;
    int64_t y16_bs;
    int y16_ts;
    float y16_slope;
    float scale;
    int scale_div;
};

// ROWS = false: every wave owns all C output rows of its own 32 * NR columns (C <= 64: few rows, wide column strips).
// ROWS = true (C = 128): every wave owns ONE 32-row tile for ALL 32 * NR columns of the block, as conv16's 128 x 128 tile does — a wave
// then streams a quarter of the weights instead of all of them (the column split at C = 128 read every A fragment for one MFMA and
// lost to two kernels), and the t tile takes the place of the x tile in LDS (nothing reads x after the first conv: the residual is
// the fp32 stream), which keeps three blocks on a CU.
template <int KT, int DIL, int C, int NR, bool BF, bool ROWS = false>
// (waves per SIMD asked of the compiler = what the LDS tile lets a CU hold: C = 256 blocks are eight waves and 74-94 KB)
__global__ __launch_bounds__(ROWS ? 2 * C : 256, C >= 256 ? ((KT - 1) * DIL <= 32 ? 4 : 2) : (C >= 64 ? 3 : 1)) void rbpair16_kernel(const RbPairParams p) {
    constexpr int G = C / 8;         // channel groups
    constexpr int NCH = C / 32;      // 32-channel chunks
    constexpr int MR = ROWS ? 1 : C / 32;  // row tiles per wave
    constexpr int NW = ROWS ? C / 32 : 4;  // waves per block (row split: one per 32-row tile: 4 at C = 128, 8 at C = 256)
    // NR 32-column tiles per wave. Column split: 4 waves x NR tiles = 256 mid columns for C <= 64 (NR = 2). Row split: NR = 4 tiles = 128.
    constexpr int BM = (ROWS ? 1 : 4) * NR * 32;  // columns of t computed per block
    constexpr int BO = BM - (KT - 1);  // output columns per block
    constexpr int P2 = (KT - 1) / 2, P1 = (KT - 1) * DIL / 2;
    constexpr int XW = BM + (KT - 1) * DIL;
    constexpr int XWP = (XW + 7) / 8 * 8;
    constexpr int TW = (BM + KT - 1 + 7) / 8 * 8;
    constexpr int STEPS = 2 * KT;
    extern __shared__ __attribute__((aligned(16))) int4v lds[];  // x tile [G][XWP] | t tile [G][TW]  (C >= 64: the t tile REPLACES the x tile)
    int4v* xs = lds;
    constexpr bool ALIAS = C >= 64;  // the t tile takes the x tile's place (one more barrier; C = 64: 75 -> 40 KB, three blocks per CU)
    int4v* ts = ALIAS ? lds : lds + G * XWP;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int b = blockIdx.y;
    const int len = p.lens ? p.lens[b] : p.tmax;
    const int t0 = blockIdx.x * BO;
    if (t0 >= len) return;
    RB_STAMP(0);
    const int h = lane >> 5;
    const int rt0 = ROWS ? wid : 0;               // first 32-row tile of this wave
    const int cb = ROWS ? 0 : wid * (NR * 32);    // first mid column of this wave
    typedef const __attribute__((address_space(3))) int4v* LdsV;

    // ---- phase 0: the input tile, all groups, straight into LDS (slot s <-> global time t0 - P2 - P1 + s) -------------------
    {
        const uint16_t* xb = p.x + (int64_t)b * p.x_bs;
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(xb), 0, 0x7fffffff, 0x00020000);
        const int tx0 = t0 - P2 - P1;
        constexpr int NP = (XWP + 63) / 64;
#pragma unroll
        for (int gi = 0; gi < G / NW; ++gi) {
            const int g = wid + NW * gi;
            const unsigned soff = (unsigned)g * (unsigned)p.x_ts * 16u;
#pragma unroll
            for (int m = 0; m < NP; ++m) {
                const int t = tx0 + lane + 64 * m;
                const int tc = t < 0 ? 0 : (t < len ? t : len - 1);
                if (64 * m + lane < XWP)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(xs + g * XWP + 64 * m), 16, tc * 16, (int)soff, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tx0 < 0 || tx0 + XWP > len) {  // sequence ends: zero padding
            const int4v z = {0, 0, 0, 0};
#pragma unroll
            for (int gi = 0; gi < G / NW; ++gi) {
                const int g = wid + NW * gi;
#pragma unroll
                for (int m = 0; m < NP; ++m) {
                    const int t = tx0 + lane + 64 * m;
                    if (64 * m + lane < XWP && (t < 0 || t >= len)) xs[g * XWP + 64 * m + lane] = z;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }

    auto mfma = [&](int4v a, int4v bq, floatx16 c) __attribute__((always_inline)) -> floatx16 {
        if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bq), c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, bq), c, 0, 0, 0);
    };
    floatx16 acc[MR][NR];

    // one conv over the LDS tile `base` (group row pitch `pitch` slots, tap step `dstep` slots): acc = sum over (chunk, tap, k-half)
    auto conv = [&](const uint16_t* wp, LdsV base, const int pitch, const int dstep) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(wp), 0, 0x7fffffff, 0x00020000);
        constexpr int TOTAL = NCH * STEPS;
        int wvoff[MR];
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) wvoff[mr] = (int)(((size_t)(rt0 + mr) * TOTAL * 64 + lane) * 16);
        auto load_a = [&](int mr, int step) __attribute__((always_inline)) -> int4v {
            return __builtin_bit_cast(int4v, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff[mr], step * 1024, 0));
        };
        // weight-fragment ring: RS slots, fetched RD steps ahead. A step is NR 32-cycle MFMAs per row tile (64-128 cycles): two steps of
        // look-ahead do not cover an L2 round trip. Eight slots where they measured faster: C = 128 (-6 %) and C = 32 (-5 %); C = 64
        // (144-149 VGPRs with them) came out 5-10 % slower on k = 3 / 7, and C = 256 is capped at 128 VGPRs (spills)
#ifndef VITS_RB16_RING
#define VITS_RB16_RING 8
#endif
        constexpr int RS = (C == 128 || C == 32) ? VITS_RB16_RING : 4, RD = RS - 2;
        int4v ring[RS][MR];
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int i = 0; i < RD; ++i) ring[i][mr] = load_a(mr, i < TOTAL ? i : TOTAL - 1);
        // (Measured with tools/rb16_micro.hip, C = 128, k = 11 / 3: operands fetched 2-3 steps ahead instead of one, or a deeper residual ring in
        // the epilogue, are +-0 — the pair is not latency-bound: the matrix pipes are 83-98 % busy at the shader clock the power budget leaves,
        // 1.1 GHz in the conv phases with the epilogue's HBM traffic beside them, 1.24 GHz without it; a register-only fp16 MFMA loop sustains
        // 1.93 GHz, tools/mfma16_peak.hip; DESIGN.md section 4.3. A flat fully unrolled step loop with absolute operand offsets instead of
        // the chunk / tap / k-half nest below compiled to 162 instead of 129 VGPRs at C = 256, k = 11, d = 5 and ran 0.58 instead of 0.33 ms.)
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            LdsV xb = base + c * 4 * pitch;
            int4v b_nxt[NR];
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) b_nxt[nr] = xb[nr * 32];
#pragma unroll
            for (int j = 0; j < KT; ++j)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int s = c * STEPS + j * 2 + kk;  // compile time after unrolling
                    {
                        const int nstep = s + RD < TOTAL ? s + RD : TOTAL - 1;
#pragma unroll
                        for (int mr = 0; mr < MR; ++mr) ring[(s + RD) % RS][mr] = load_a(mr, nstep);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    int4v b_cur[NR];
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr) b_cur[nr] = b_nxt[nr];
                    {
                        const int noff = kk == 0 ? 2 * pitch + j * dstep : (j + 1) * dstep;
#ifndef VAR_NOB  // (ablations of tools/rb16_micro.hip: VAR_NOB / VAR_NOEPI drop the LDS operand reads / the epilogue)
#pragma unroll
                        for (int nr = 0; nr < NR; ++nr) b_nxt[nr] = xb[noff + nr * 32];
#endif
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                        for (int nr = 0; nr < NR; ++nr) acc[mr][nr] = mfma(ring[s % RS][mr], b_cur[nr], acc[mr][nr]);
                }
        }
    };

    __syncthreads();
    RB_STAMP(1);
    // ---- phase 1: conv1 over the x tile: mid column i reads x slots i + j*DIL ------------------------------------------------
    conv(p.w1, (LdsV)(xs + h * XWP + cb + (lane & 31)), XWP, DIL);
    RB_STAMP(2);
    if constexpr (ALIAS) __syncthreads();  // every wave is done with the x tile: t takes its place

    // ---- phase 2: t = round(leaky_relu(conv1 + b1)), zero outside the sequence, into LDS (group layout) -------------------
    // (whole 16-byte slots per lane: in the C layout lane l holds channels 0-3 and lane l + 32 channels 4-7 of a group; v_permlane32_swap
    // trades halves between two groups so that lanes < 32 write one group's slots and lanes >= 32 the other's — two ds_write_b64 at a 16-byte
    // stride were a 4-way bank conflict, 0.08-0.20 of this kernel's LDS cycles in the round-2 PMC pass)
    {
        typedef __attribute__((address_space(3))) int4v* LdsS;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            float4v bias[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) bias[g] = *reinterpret_cast<const float4v*>(p.b1 + (rt0 + mr) * 32 + 8 * g + 4 * h);
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                const int i = cb + nr * 32 + (lane & 31);
                const int tm = t0 - P2 + i;
                int2v w[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[mr][nr][4 * g + e] + bias[g][e];
                        v[e] = fmaxf(v[e], v[e] * p.slope);
                        if (tm < 0 || tm >= len) v[e] = 0.f;  // the second conv's zero padding
                    }
                    w[g].x = (int)rb_pack16<BF>(v[0], v[1]);
                    w[g].y = (int)rb_pack16<BF>(v[2], v[3]);
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const auto x = __builtin_amdgcn_permlane32_swap((unsigned)w[2 * k].x, (unsigned)w[2 * k + 1].x, false, false);
                    const auto y = __builtin_amdgcn_permlane32_swap((unsigned)w[2 * k].y, (unsigned)w[2 * k + 1].y, false, false);
                    *((LdsS)(ts + ((rt0 + mr) * 4 + 2 * k + h) * TW + i)) = int4v{(int)x[0], (int)y[0], (int)x[1], (int)y[1]};
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();
    RB_STAMP(3);

    // ---- phase 3: conv2 over the t tile: output column o reads t slots o + j -------------------------------------------------
    conv(p.w2, (LdsV)(ts + h * TW + cb + (lane & 31)), TW, 1);
    RB_STAMP(4);

    // ---- phase 4: epilogue (as conv16's group epilogue): + b2, + residual, resblock sum / scale, fp32 stream + 16-bit copy ----
#ifdef VAR_NOEPI
    {
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
        if (sacc == 12345.678f) p.yg[tid] = sacc;
        RB_STAMP(5);
        return;
    }
#endif
    {
        float* yg = p.yg ? p.yg + (int64_t)b * p.g_bs : nullptr;
        const float* rg = p.resg ? p.resg + (int64_t)b * p.g_bs : nullptr;
        const float* ag = p.accg ? p.accg + (int64_t)b * p.g_bs : nullptr;
        uint16_t* y16 = p.y16 ? p.y16 + (int64_t)b * p.y16_bs : nullptr;
        constexpr int NGR = MR * 4;
#ifndef VITS_RB16_ERD
#define VITS_RB16_ERD (ROWS ? 2 : 3)
#endif
        constexpr int RD = VITS_RB16_ERD;  // residual look-ahead ring (row split: NR = 4 float4 per entry, and 208 VGPRs with three of them)
        // (the accumulator of the resblock sum — last pair of resblocks 1 and 2, read in place: accg == yg — travels in the same look-ahead ring
        // as the residual: read inside the store loop, every read waits behind the previous store it might alias — 16 dependent round trips)
        // (not in the instantiations compiled for four waves per SIMD, C = 256 with short tiles: 128 VGPRs, the ring would spill 50 of them)
        constexpr bool AVRING = !(C >= 256 && (KT - 1) * DIL <= 32);
        float4v rv[RD][NR], av[AVRING ? RD : 1][NR];
        auto col_ok = [&](int nr, int& t) __attribute__((always_inline)) -> bool {
            const int o = cb + nr * 32 + (lane & 31);
            t = t0 + o;
            return o < BO && t < len;
        };
        auto load_res = [&](int it, float4v* dst, float4v* adst) __attribute__((always_inline)) {
            const int ch0 = (rt0 + it / 4) * 32 + 8 * (it & 3) + 4 * h;
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                int t;
                dst[nr] = float4v{0.f, 0.f, 0.f, 0.f};
                if constexpr (AVRING) adst[nr] = float4v{0.f, 0.f, 0.f, 0.f};
                if (!col_ok(nr, t)) continue;
                const int64_t go = ((int64_t)(ch0 >> 3) * p.g_ts + t) * 8 + (ch0 & 7);
                if (rg) dst[nr] = *reinterpret_cast<const float4v*>(rg + go);
                if constexpr (AVRING) {
                    if (ag) adst[nr] = *reinterpret_cast<const float4v*>(ag + go);
                }
            }
        };
#pragma unroll
        for (int i = 0; i < RD - 1; ++i)
            if (i < NGR) load_res(i, rv[i], av[AVRING ? i : 0]);
#pragma unroll
        for (int it = 0; it < NGR; ++it) {
            const int mr = it / 4, g = it & 3;
            if (it + RD - 1 < NGR) load_res(it + RD - 1, rv[(it + RD - 1) % RD], av[AVRING ? (it + RD - 1) % RD : 0]);
            __builtin_amdgcn_sched_barrier(0);
            const int ch0 = (rt0 + mr) * 32 + 8 * g + 4 * h;
            const float4v bias = *reinterpret_cast<const float4v*>(p.b2 + ch0);
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                int t;
                if (!col_ok(nr, t)) continue;
                const int64_t go = ((int64_t)(ch0 >> 3) * p.g_ts + t) * 8 + (ch0 & 7);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[mr][nr][4 * g + e] + bias[e];
                    if (rg) v[e] = rv[it % RD][nr][e] + v[e];
                }
                if (ag) {
                    float4v a4;
                    if constexpr (AVRING) a4 = av[it % RD][nr];
                    else a4 = *reinterpret_cast<const float4v*>(ag + go);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = a4[e] + v[e];
                        v[e] = p.scale_div ? v[e] / p.scale : v[e] * p.scale;
                    }
                }
                if (yg) *reinterpret_cast<float4v*>(yg + go) = float4v{v[0], v[1], v[2], v[3]};
                if (y16) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * p.y16_slope);
                    int2v w2;
                    w2.x = (int)rb_pack16<BF>(v[0], v[1]);
                    w2.y = (int)rb_pack16<BF>(v[2], v[3]);
                    *reinterpret_cast<int2v*>(y16 + ((int64_t)(ch0 >> 3) * p.y16_ts + t) * 8 + (ch0 & 7)) = w2;
                }
            }
        }
    }
    RB_STAMP(5);
}

// ---- host side -----------------------------------------------------------------------------------------------------------
template <int KT, int DIL, int C, bool BF, int NR>
static hipError_t launch_rb_nr(const RbPairParams& p, int batch, hipStream_t s) {
    constexpr bool ROWS = C >= 128;
    constexpr int BM = (ROWS ? 1 : 4) * NR * 32;
    constexpr int BO = BM - (KT - 1);
    constexpr int XWP = (BM + (KT - 1) * DIL + 7) / 8 * 8, TW = (BM + KT - 1 + 7) / 8 * 8;
    const size_t lds = C >= 64 ? (size_t)(C / 8) * (XWP > TW ? XWP : TW) * 16 : (size_t)(C / 8) * (XWP + TW) * 16;
    static BigLdsOnce big_lds_set;
    if (lds > 64 * 1024 && big_lds_set.needed()) {
        hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpair16_kernel<KT, DIL, C, NR, BF, ROWS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (ea != hipSuccess) return ea;
        big_lds_set.done();
    }
    dim3 grid((p.tmax + BO - 1) / BO, batch);
    VITS_KLAUNCH((rbpair16_kernel<KT, DIL, C, NR, BF, ROWS>), grid, dim3(ROWS ? 2 * C : 256), lds, s, p);
    return hipGetLastError();
}
#ifndef VITS_RB16_NARROW_NR
#define VITS_RB16_NARROW_NR 2
#endif
template <int KT, int DIL, int C, bool BF>
static hipError_t launch_rb(const RbPairParams& p, int batch, hipStream_t s) {
    if constexpr (C >= 128) {
        // small grids (batch 1 ... 4): the row-split blocks own 128 columns (NR = 4 tiles per wave), i.e. 16 blocks for the 1,808 frames of an
        // utterance at C = 256 on 256 CUs; blocks of VITS_RB16_NARROW_NR tiles are that many times shorter chains on that many more CUs
        // (the halo costs more: they lose as soon as the chip is full). Same chains per output: bit-identical.
        constexpr int BO4 = 4 * 32 - (KT - 1);
        if ((int64_t)((p.tmax + BO4 - 1) / BO4) * batch <= kernel_knobs().rb16_narrow_max) return launch_rb_nr<KT, DIL, C, BF, VITS_RB16_NARROW_NR>(p, batch, s);
        return launch_rb_nr<KT, DIL, C, BF, 4>(p, batch, s);
    } else {
        return launch_rb_nr<KT, DIL, C, BF, 2>(p, batch, s);
    }
}

template <int KT, int C, bool BF>
static hipError_t launch_rb_dil(int dil, const RbPairParams& p, int batch, hipStream_t s) {
    switch (dil) {
        case 1: return launch_rb<KT, 1, C, BF>(p, batch, s);
        case 3: return launch_rb<KT, 3, C, BF>(p, batch, s);
        case 5: return launch_rb<KT, 5, C, BF>(p, batch, s);
        default: return hipErrorInvalidValue;
    }
}

template <int C, bool BF>
static hipError_t launch_rb_kt(int kt, int dil, const RbPairParams& p, int batch, hipStream_t s) {
    switch (kt) {
        case 3: return launch_rb_dil<3, C, BF>(dil, p, batch, s);
        case 7: return launch_rb_dil<7, C, BF>(dil, p, batch, s);
        case 11: return launch_rb_dil<11, C, BF>(dil, p, batch, s);
        default: return hipErrorInvalidValue;
    }
}

bool rbpair16_supported(int channels, int kt, int dil) {
    // VITS_FUSE16_MAXC=64 keeps the C = 128 pairs on two kernels
    const int maxc = kernel_knobs().fuse16_maxc;
    if (!(kt == 3 || kt == 7 || kt == 11) || channels > maxc) return false;
    if (channels == 32 || channels == 64 || channels == 128 || channels == 256) return dil == 1 || dil == 3 || dil == 5;
    return false;
}

hipError_t launch_rbpair16(const PackedConv& c1, const PackedConv& c2, const RbPair16Call& c, int arith, hipStream_t s) {
    if (!c1.wp16 || !c2.wp16 || c1.cin != c1.cout || c2.cin != c1.cout || c2.cout != c1.cout || c1.kt != c2.kt || !rbpair16_supported(c1.cin, c1.kt, c.dil))
        return hipErrorInvalidValue;
    RbPairParams p;
    p.x = c.x.p;
    p.x_bs = c.x.bs;
    p.x_ts = c.x.ts;
    p.w1 = c1.wp16;
    p.w2 = c2.wp16;
    p.b1 = c1.bias;
    p.b2 = c2.bias;
    if (!p.b1 || !p.b2) return hipErrorInvalidValue;
    p.lens = c.lens;
    p.tmax = c.tmax;
    p.slope = c.slope;
    p.yg = c.yg;
    p.resg = c.resg;
    p.accg = c.accg;
    p.g_bs = c.g_bs;
    p.g_ts = c.g_ts;
    p.y16 = c.y16.p;
    p.y16_bs = c.y16.bs;
    p.y16_ts = c.y16.ts;
    p.y16_slope = c.y16_slope;
    p.scale = c.scale;
    p.scale_div = c.scale_div;
    const bool bf = arith == VITS_ARITH_BF16;
    if (c1.cin == 32) return bf ? launch_rb_kt<32, true>(c1.kt, c.dil, p, c.batch, s) : launch_rb_kt<32, false>(c1.kt, c.dil, p, c.batch, s);
    if (c1.cin == 64) return bf ? launch_rb_kt<64, true>(c1.kt, c.dil, p, c.batch, s) : launch_rb_kt<64, false>(c1.kt, c.dil, p, c.batch, s);
    if (c1.cin == 128) return bf ? launch_rb_kt<128, true>(c1.kt, c.dil, p, c.batch, s) : launch_rb_kt<128, false>(c1.kt, c.dil, p, c.batch, s);
    return bf ? launch_rb_kt<256, true>(c1.kt, c.dil, p, c.batch, s) : launch_rb_kt<256, false>(c1.kt, c.dil, p, c.batch, s);
}

}  // namespace vits
