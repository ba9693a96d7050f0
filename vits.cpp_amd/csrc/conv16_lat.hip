// conv16_lat.hip — the 16-bit-operand ResBlock convs of the WIDE vocoder stages (C = 128 / 256) on SMALL grids (batch 1 ... 4: config 2, the
// reference's own use), round 6.
//     t = round(leaky_relu(Conv_{k,d}(x) + b1))            (group epilogue, 16-bit output)      (/root/reference/src/vits.cpp:556-567)
//     y' = y + Conv_{k,1}(t) + b2 [, resblock sum / scale]  (group epilogue, fp32 stream + copy)  (vits.cpp:568-581, 622-635)
// Why: at one utterance a C = 256 stage is 1,792 columns. As ONE fused kernel per pair (rbpair16.hip) that is 28-32 blocks, each of which
// pulls both convs' weights — 2.9 MB at k = 11 — through ONE compute unit's path to L2 at 57-75 GB/s: 40-48 us per pair, three pairs per
// resblock one behind the other, 220 of the chip's 256 CUs idle (profiles/round6_b1_f16_timeline.txt). conv16_kernel (the throughput
// kernel: LDS ring, producer wave, a barrier per 32-channel chunk) takes 17-21 us per conv on such a grid. Here a conv is dealt out as
// (row tile pair, 32 or 64 columns) blocks — 224-448 of them — so that every CU streams 90-360 KB of weights instead of 2.9 MB:
//   * the whole input tile [C/8 groups][32 NR + (k-1) d slots] goes to LDS in ONE shot (LDS-DMA, all waves), no barrier inside the K loop;
//   * a wave owns one 32-row tile x NR 32-column tiles; its A fragments come straight from memory through a ring of 16 (fetched 14 steps
//     ahead: the weights of a batch-1 step are not in L2);
//   * K order = (chunk, tap, k-half) on v_mfma_f32_32x32x16_{f16,bf16}, epilogue = conv16_kernel's group epilogue expression for
//     expression: bit-identical to conv16_kernel / rbpair16_kernel / rbblock16_kernel (kernel-choice test).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/vits.h"
#include "kernels.h"

namespace vits {

namespace c16l {
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
}  // namespace c16l

struct Conv16LatParams {
    const uint16_t* x;  // group layout [b][C/8][x_ts][8], already activated
    int64_t x_bs;
    int x_ts;
    const float* xf;  // XF32: the input as fp32 [b][c][t] instead — rounded on its way into LDS (launch_to_group16's expression: the conv_pre of the vocoder)
    int64_t xf_bs;
    int xf_cs;
    const uint16_t* wp;  // A fragments (pack_conv_weights16)
    const float* bias;
    const int* lens;
    int tmax;
    int dil, pad_l, pitch;
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

// Block = WM waves = WM consecutive 32-row tiles of the same 32 NR columns. (The body as a device function: conv16_lat_kernel runs it for one conv,
// conv16_lat_group_kernel for the same-position convs of a stage's three resblocks in ONE launch.)
template <int KT, int C, int WM, int NR, bool BF, bool XF32>  // C: input channels (= output channels for the resblock convs; the launch's grid says how many row tiles)
__device__ __forceinline__ void conv16_lat_body(const Conv16LatParams& p, c16l::int4v* xs, const int b) {
    using namespace c16l;
    constexpr int G = C / 8, NCH = C / 32, STEPS = 2 * KT, TOTAL = NCH * STEPS, BN = 32 * NR;
    // xs: [G][pitch] slots of 8 x 16 bit; slot s of a row <-> time t0 - pad_l + s
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t0 = blockIdx.x * BN;
    const int len = p.lens ? p.lens[b] : p.tmax;
    if (t0 >= len) return;
    const int rt = blockIdx.y * WM + wid;  // this wave's 32-row tile
    const int h = lane >> 5;
    const int pitch = p.pitch, dil = p.dil;
    typedef const __attribute__((address_space(3))) int4v* LdsV;

    // ---- the weight stream: requested first, so that its first round trip and the fill's are one ----
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.wp), 0, 0x7fffffff, 0x00020000);
    const int wvoff = (int)(((size_t)rt * TOTAL * 64 + lane) * 16);
    auto load_a = [&](int step) __attribute__((always_inline)) -> int4v {
        return __builtin_bit_cast(int4v, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, step * 1024, 0));
    };
    constexpr int RS = 16, RD = RS - 2;
    int4v ring[RS];
#pragma unroll
    for (int i = 0; i < RD; ++i) ring[i] = load_a(i < TOTAL ? i : TOTAL - 1);

    // ---- the epilogue's operands — residual, resblock sum, bias — requested NOW: behind the K loop they were a second exposed round trip ----
    const int colbase = t0 + (lane & 31);
    const int rowoff = 4 * h;
    float* yg = p.yg ? p.yg + (int64_t)b * p.g_bs : nullptr;
    const float* rg = p.resg ? p.resg + (int64_t)b * p.g_bs : nullptr;
    const float* ag = p.accg ? p.accg + (int64_t)b * p.g_bs : nullptr;
    uint16_t* y16 = p.y16 ? p.y16 + (int64_t)b * p.y16_bs : nullptr;
    float4v rv[4][NR], av[4][NR], bias4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int ch0 = rt * 32 + 8 * g + rowoff;
        bias4[g] = float4v{0.f, 0.f, 0.f, 0.f};
        if (p.bias) bias4[g] = *reinterpret_cast<const float4v*>(p.bias + ch0);
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            const int t = colbase + nr * 32;
            rv[g][nr] = float4v{0.f, 0.f, 0.f, 0.f};
            av[g][nr] = float4v{0.f, 0.f, 0.f, 0.f};
            const int64_t go = ((int64_t)(ch0 >> 3) * p.g_ts + t) * 8 + (ch0 & 7);
            if (rg && t < len) rv[g][nr] = *reinterpret_cast<const float4v*>(rg + go);
            if (ag && t < len) av[g][nr] = *reinterpret_cast<const float4v*>(ag + go);
        }
    }

    // ---- the input tile, all groups, straight into LDS ----
    if constexpr (XF32) {
        // fp32 [c][t] -> rounded 16-byte slots (8 channels of one time step), launch_to_group16's expression with slope 1; zero outside the sequence
        const float* xb = p.xf + (int64_t)b * p.xf_bs;
        const int tx0 = t0 - p.pad_l;
        const int xw = BN + (KT - 1) * dil;
        for (int idx = (int)threadIdx.x; idx < G * xw; idx += WM * 64) {
            const int g = idx / xw, sl = idx - g * xw;
            const int t = tx0 + sl;
            int4v o = {0, 0, 0, 0};
            if (t >= 0 && t < len) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = xb[(int64_t)(g * 8 + e) * p.xf_cs + t];
                    v[e] = fmaxf(f, f * 1.0f);
                }
                o.x = (int)pack16<BF>(v[0], v[1]);
                o.y = (int)pack16<BF>(v[2], v[3]);
                o.z = (int)pack16<BF>(v[4], v[5]);
                o.w = (int)pack16<BF>(v[6], v[7]);
            }
            xs[g * pitch + sl] = o;
        }
    } else {
        const uint16_t* xb = p.x + (int64_t)b * p.x_bs;
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(xb), 0, 0x7fffffff, 0x00020000);
        const int tx0 = t0 - p.pad_l;
        const int xw = BN + (KT - 1) * dil;  // slots a row holds (<= pitch)
        constexpr int NP = 2;  // 64-slot pieces per row: (k - 1) d <= 50. (A literal: hipcc 7.2 drops the kernel's host stub when an array sized by a template-dependent expression meets the LDS-DMA builtin.)
        static_assert((BN + 50 + 63) / 64 <= NP, "row pieces");
        int voff[NP];
        bool in[NP], oob[NP];
#pragma unroll
        for (int m = 0; m < NP; ++m) {
            const int t = tx0 + lane + 64 * m;
            const int tc = t < 0 ? 0 : (t < len ? t : len - 1);
            voff[m] = tc * 16;
            in[m] = 64 * m + lane < xw;
            oob[m] = t != tc;
        }
#pragma unroll
        for (int gi = 0; gi < G / WM; ++gi) {
            const int g = wid + WM * gi;
            const unsigned soff = (unsigned)g * (unsigned)p.x_ts * 16u;
#pragma unroll
            for (int m = 0; m < NP; ++m)
                if (in[m]) __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(xs + g * pitch + 64 * m), 16, voff[m], (int)soff, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tx0 < 0 || tx0 + xw > len) {  // sequence ends: the conv's zero padding
            const int4v z = {0, 0, 0, 0};
#pragma unroll
            for (int gi = 0; gi < G / WM; ++gi) {
                const int g = wid + WM * gi;
#pragma unroll
                for (int m = 0; m < NP; ++m)
                    if (in[m] && oob[m]) xs[g * pitch + 64 * m + lane] = z;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();

    auto mfma = [&](int4v a, int4v bq, floatx16 c) __attribute__((always_inline)) -> floatx16 {
        if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bq), c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, bq), c, 0, 0, 0);
    };
    floatx16 acc[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // output column u reads slots u + j dil of group rows (4 c + 2 kk + h): the (chunk, tap, k-half) nest of conv16_kernel / rbpair16_kernel
    {
        LdsV base = (LdsV)(xs + h * pitch + (lane & 31));
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
                    ring[(s + RD) % RS] = load_a(s + RD < TOTAL ? s + RD : TOTAL - 1);
                    __builtin_amdgcn_sched_barrier(0);
                    int4v b_cur[NR];
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr) b_cur[nr] = b_nxt[nr];
                    {
                        // next step: the other k-half of this tap, or k-half 0 of the next tap (past the last tap of a chunk: a slot a little further on, unused)
                        const int noff = kk == 0 ? 2 * pitch + j * dil : (j + 1) * dil;
#pragma unroll
                        for (int nr = 0; nr < NR; ++nr) b_nxt[nr] = xb[noff + nr * 32];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr) acc[nr] = mfma(ring[s % RS], b_cur[nr], acc[nr]);
                }
        }
    }

    // ---- epilogue: conv16_kernel's group epilogue (MR = 1): this lane owns channels ch0 .. ch0 + 3 of one time step per group ----
    // (no block of this launch reads what another writes: the residual / sum may alias the output element for element only)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int ch0 = rt * 32 + 8 * g + rowoff;
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            const int t = colbase + nr * 32;
            if (t >= len) continue;
            const int64_t go = ((int64_t)(ch0 >> 3) * p.g_ts + t) * 8 + (ch0 & 7);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = acc[nr][4 * g + e] + bias4[g][e];
                if (rg) v[e] = rv[g][nr][e] + v[e];
            }
            if (ag) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = av[g][nr][e] + v[e];
                    v[e] = p.scale_div ? v[e] / p.scale : v[e] * p.scale;
                }
            }
            if (yg) *reinterpret_cast<float4v*>(yg + go) = float4v{v[0], v[1], v[2], v[3]};
            if (y16) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * p.y16_slope);  // slope 1 = identity
                int2v w2;
                w2.x = (int)pack16<BF>(v[0], v[1]);
                w2.y = (int)pack16<BF>(v[2], v[3]);
                *reinterpret_cast<int2v*>(y16 + ((int64_t)(ch0 >> 3) * p.y16_ts + t) * 8 + (ch0 & 7)) = w2;
            }
        }
    }
}

template <int KT, int C, int WM, int NR, bool BF, bool XF32 = false>
__global__ __launch_bounds__(WM * 64) void conv16_lat_kernel(const Conv16LatParams p) {
    extern __shared__ __attribute__((aligned(16))) c16l::int4v xs_dyn16[];
    conv16_lat_body<KT, C, WM, NR, BF, XF32>(p, xs_dyn16, (int)blockIdx.z);
}

// The same-position convs of a stage's three resblocks (k = 3 / 7 / 11, equal dilation) in ONE launch: blockIdx.z = 3 x utterance + member. At one to four
// utterances the three chains of a stage ran on three streams — a fork and a join of 20-45 us of queue hand-over each per stage; as six grouped launches (+ the
// sum launch) on the main stream there is neither (batch 1: the C = 256 stage 106 -> see DESIGN 9). A member is the body above, instruction for instruction.
struct Conv16LatGroupParams {
    Conv16LatParams m[3];  // members in the order k = 3, 7, 11
};
template <int C, int WM, int NR, bool BF>
__global__ __launch_bounds__(WM * 64) void conv16_lat_group_kernel(const Conv16LatGroupParams gp) {
    extern __shared__ __attribute__((aligned(16))) c16l::int4v xs_dyn16g[];
    const int member = (int)blockIdx.z % 3, b = (int)blockIdx.z / 3;
    if (member == 0) conv16_lat_body<3, C, WM, NR, BF, false>(gp.m[0], xs_dyn16g, b);
    else if (member == 1) conv16_lat_body<7, C, WM, NR, BF, false>(gp.m[1], xs_dyn16g, b);
    else conv16_lat_body<11, C, WM, NR, BF, false>(gp.m[2], xs_dyn16g, b);
}

// ---- host side -----------------------------------------------------------------------------------------------------------
// Which convs: the group-layout ResBlock convs (same length in and out, bias, no activation of the stored value) of a C = 128 / 256
// stage with k = 3 / 7 / 11, while the launch has at most VITS_LAT16H_MAX_TILES (C = 256) / VITS_LAT16H_MAX_TILES_C128 32 x 32 output tiles (one to four utterances).
// Measured (f16, ms per batch of 1 / 2 / 3 / 4 x 128 ids; fused pairs -> this kernel): 1.485 -> 1.428, 1.596 -> 1.553, 1.826 -> 1.780, 1.974 -> 1.939; C = 128 at batch 1 (1792 tiles): + 6 ... 12 us.
bool conv16_lat_shape_ok(int channels, int kt, int dil, int batch, int tmax) {
    const KernelKnobs& kn = kernel_knobs();
    if (kn.no_lat16h) return false;
    if (!(channels == 128 || channels == 256) || !(kt == 3 || kt == 7 || kt == 11) || dil < 1 || (kt - 1) * dil > 50) return false;
    const int64_t tiles = (int64_t)((tmax + 31) / 32) * (channels / 32) * batch;
    return tiles <= (channels == 256 ? kn.lat16h_max_tiles : kn.lat16h_max_tiles_c128);
}
bool conv16_lat_wanted(const PackedConv& w, const Conv16Call& c) {
    if (c.tile >= 0 || w.cin != w.cout || !conv16_lat_shape_ok(w.cin, w.kt, c.dil, c.batch, c.t_out)) return false;
    if (w.epi != EPI_STD || !(c.yg || c.y16.p) || !w.wp16 || !w.bias) return false;
    if (c.len_in != c.len_out || c.t_in != c.t_out || c.post_act != 0 || c.ct_crop != 0 || c.pad_l != (w.kt - 1) * c.dil / 2) return false;
    return !(c.y.p || c.res.p || c.acc.p || c.y2);
}

template <int KT, int C, int WM, int NR, bool BF>
static hipError_t launch_c16l(const Conv16LatParams& p, int batch, hipStream_t s) {
    const size_t lds = (size_t)(C / 8) * p.pitch * 16 + 8 * 16;  // (+ the slots the look-ahead of the last tap reads past the tile, value unused)
    static BigLdsOnce big;
    if (lds > 64 * 1024 && big.needed()) {
        if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv16_lat_kernel<KT, C, WM, NR, BF>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) return e;
        big.done();
    }
    dim3 grid((p.tmax + 32 * NR - 1) / (32 * NR), C / 32 / WM, batch);
    VITS_KLAUNCH((conv16_lat_kernel<KT, C, WM, NR, BF>), grid, dim3(WM * 64), lds, s, p);
    return hipGetLastError();
}

// The vocoder's conv_pre (F -> up_init channels, k = 7, vits.cpp:601) on a small grid, straight from the fp32 flow output: the converter launch
// (launch_to_group16) and the throughput kernel's 14 us become one launch of this kernel (batch 1: - 15 us). Same rounding expression, same K order,
// same group epilogue: same bits.
bool conv16_lat_pre_wanted(const PackedConv& w, int batch, int tmax) {
    const KernelKnobs& kn = kernel_knobs();
    if (kn.no_lat16h || kn.no_lat16h_pre || !w.wp16 || !w.bias || w.epi != EPI_STD) return false;
    if (w.cin != 192 || w.kt != 7 || (w.cout % 64) != 0) return false;
    const int64_t tiles = (int64_t)((tmax + 31) / 32) * (w.cout / 32) * batch;
    return tiles <= kn.lat16h_max_tiles;
}
hipError_t launch_conv16_lat_pre(const PackedConv& w, TensorRef x, const int* lens, int batch, int tmax, Ref16 y16, float y16_slope, int arith, hipStream_t s) {
    if (!conv16_lat_pre_wanted(w, batch, tmax) || !x.p || !y16.p) return hipErrorInvalidValue;
    Conv16LatParams p = {};
    p.xf = x.p;
    p.xf_bs = x.bs;
    p.xf_cs = x.cs;
    p.wp = w.wp16;
    p.bias = w.bias;
    p.lens = lens;
    p.tmax = tmax;
    p.dil = 1;
    p.pad_l = (w.kt - 1) / 2;
    p.pitch = (32 + (w.kt - 1) + 7) / 8 * 8;
    p.y16 = y16.p;
    p.y16_bs = y16.bs;
    p.y16_ts = y16.ts;
    p.y16_slope = y16_slope;
    p.scale = 1.f;
    const size_t lds = (size_t)(192 / 8) * p.pitch * 16 + 8 * 16;
    dim3 grid((tmax + 31) / 32, w.cout / 32 / 2, batch);
    if (arith == VITS_ARITH_BF16) VITS_KLAUNCH((conv16_lat_kernel<7, 192, 2, 1, true, true>), grid, dim3(128), lds, s, p);
    else VITS_KLAUNCH((conv16_lat_kernel<7, 192, 2, 1, false, true>), grid, dim3(128), lds, s, p);
    return hipGetLastError();
}

static Conv16LatParams c16l_params(const PackedConv& w, const Conv16Call& c, int nr) {
    Conv16LatParams p = {};
    p.x = c.x.p;
    p.x_bs = c.x.bs;
    p.x_ts = c.x.ts;
    p.wp = w.wp16;
    p.bias = w.bias;
    p.lens = c.len_out;
    p.tmax = c.t_out;
    p.dil = c.dil;
    p.pad_l = c.pad_l;
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
    p.pitch = (32 * nr + (w.kt - 1) * c.dil + 7) / 8 * 8;
    return p;
}

// the group launch: members k = 3, 7, 11 of one stage (C = 256), equal shapes; two row tiles x 32 columns per block
bool conv16_lat_group_wanted(const PackedConv* const* w, const Conv16Call* c) {
    if (kernel_knobs().no_lat16h_group) return false;
    static const int kts[3] = {3, 7, 11};
    for (int i = 0; i < 3; ++i) {
        if (!conv16_lat_wanted(*w[i], c[i]) || w[i]->kt != kts[i] || w[i]->cin != 256) return false;
        if (c[i].batch != c[0].batch || c[i].t_out != c[0].t_out || c[i].len_out != c[0].len_out || c[i].dil != c[0].dil) return false;
    }
    return true;
}
hipError_t launch_conv16_lat_group(const PackedConv* const* w, const Conv16Call* c, int arith, hipStream_t s) {
    if (!conv16_lat_group_wanted(w, c)) return hipErrorInvalidValue;
    Conv16LatGroupParams gp;
    const int shape = kernel_knobs().lat16h_group_shape;  // 10 WM + NR
    const int wm = shape / 10, nr = shape % 10;
    int pitch = 0;
    for (int i = 0; i < 3; ++i) {
        gp.m[i] = c16l_params(*w[i], c[i], nr);
        pitch = gp.m[i].pitch > pitch ? gp.m[i].pitch : pitch;
    }
    const size_t lds = (size_t)(256 / 8) * pitch * 16 + 8 * 16;
    dim3 grid((c[0].t_out + 32 * nr - 1) / (32 * nr), 256 / 32 / wm, 3 * c[0].batch);
    const bool bf = arith == VITS_ARITH_BF16;
#define VITS_C16LG(WM_, NR_)                                                                                                                        \
    if (wm == WM_ && nr == NR_) {                                                                                                                   \
        if (bf) VITS_KLAUNCH((conv16_lat_group_kernel<256, WM_, NR_, true>), grid, dim3(64 * WM_), lds, s, gp);                                      \
        else VITS_KLAUNCH((conv16_lat_group_kernel<256, WM_, NR_, false>), grid, dim3(64 * WM_), lds, s, gp);                                        \
        return hipGetLastError();                                                                                                                   \
    }
    VITS_C16LG(2, 1)
    VITS_C16LG(4, 1)
    VITS_C16LG(2, 2)
    VITS_C16LG(4, 2)
#undef VITS_C16LG
    return hipErrorInvalidValue;
}

hipError_t launch_conv16_lat(const PackedConv& w, const Conv16Call& c, int arith, hipStream_t s) {
    if (!conv16_lat_wanted(w, c)) return hipErrorInvalidValue;
    const bool bf = arith == VITS_ARITH_BF16;
    // shape (VITS_LAT16H_SHAPE = 10 WM + NR): two row tiles x 32 columns per block by default (batch 1: 1.483 ms against 1.498 with 64 columns and 1.492 with four row tiles x 64)
    const int shape = kernel_knobs().lat16h_shape;
    const int wm = shape / 10, nr = shape % 10;
    const Conv16LatParams p = c16l_params(w, c, nr);
#define VITS_C16L_GO(K, CC, WM_, NR_)                                                                                                           \
    if (w.kt == K && w.cin == CC && wm == WM_ && nr == NR_) return bf ? launch_c16l<K, CC, WM_, NR_, true>(p, c.batch, s) : launch_c16l<K, CC, WM_, NR_, false>(p, c.batch, s)
#define VITS_C16L_SHAPES(K, CC) \
    VITS_C16L_GO(K, CC, 2, 1);  \
    VITS_C16L_GO(K, CC, 2, 2);  \
    VITS_C16L_GO(K, CC, 4, 2)
    VITS_C16L_SHAPES(3, 128);
    VITS_C16L_SHAPES(7, 128);
    VITS_C16L_SHAPES(11, 128);
    VITS_C16L_SHAPES(3, 256);
    VITS_C16L_SHAPES(7, 256);
    VITS_C16L_SHAPES(11, 256);
#undef VITS_C16L_SHAPES
#undef VITS_C16L_GO
    return hipErrorInvalidValue;
}

}  // namespace vits
