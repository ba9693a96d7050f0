// conv_split.hip — VITS_ARITH_F32_SPLIT (round 6): fp32-ACCURATE Conv1d on the 16-bit matrix cores by operand splitting.
//
// The exact-fp32 path (conv_mfma.hip) is bound by v_mfma_f32_32x32x2_f32 at 157 TFLOP/s and has been at 0.80 of that since round 3: nothing left in
// that formulation moves the BASELINE metric. The reference's conv weights are fp16 values (/root/reference/scripts/export_vits.py:87);
// an fp16 value is the exact sum of two bf16 values, an fp32 activation the exact sum of three:
//     w = w1 + w2,  a = a1 + a2 + a3        (each piece = round-to-nearest-even of what the previous ones left)
//     w a  =  a1 w1 + a1 w2 + a2 w1 + a2 w2 + a3 w1   + a3 w2,   the dropped a3 w2 <= 2^-24 |a1 w1|
// Every kept product of two bf16 values is exact in fp32, so five v_mfma_f32_32x32x16_bf16 with fp32 accumulation compute the fp32 product sum to the
// accuracy of an fp32 accumulation — NOT bit-identical to conv_mfma's fmaf chain (the 16-bit MFMA adds its sixteen products in its own order), but as close
// to the exact sum as that chain is: tools/split_micro.hip measures max |error| 3.24e-6 against 3.26e-6 for the fmaf chain on the same data, at 228
// TFLOP/s-equivalent against 126-130 (the gate of VERDICT r5 next 4 was 170). 5 x 32 = 160 matrix-pipe cycles per 16 products instead of 8 x 64 = 512.
//
// What runs here: the ResBlock convolutions of the vocoder's wide stages (C >= 128: /root/reference/src/vits.cpp:545-581 through conv1d_impl,
// /root/reference/src/include/custom-ops.h:680-694), 64 % of the path's FLOPs; everything else — stage one (durations stay bit-exact), the flow, the
// upsamplers, the narrow stages' fused fp32 kernels — is the exact-fp32 path unchanged. Opt-in (vits_model_set_arith(model, VITS_ARITH_F32_SPLIT)), never the
// default, and reported as its own sub-result: the default arithmetic keeps "batch 1 == row of a batch, bit for bit" and the fmaf-chain identity.
//
// Data flow (as conv16.hip's group layout, with three planes): a conv input is [batch][plane 3][C/8][time][8] bf16 — one 16-byte slot = 8 channels of one
// time step = one lane's B operand — written by the PRODUCING epilogue (LeakyReLU, then the three roundings: `writers split, readers don't`; the stage input
// by split_planes_kernel); tiles stream HBM -> LDS with LDS-DMA (no VGPRs, no VALU), every tap is a shifted ds_read_b128 per plane. Weights: two planes of
// A fragments [row tile][chunk][tap][k-half][plane 2][lane][8], packed at vits_model_set_arith. The fp32 residual stream, the resblock sum and the activated
// copy stay in the standard [batch][channel][time] layout of the fp32 path (the fused fp32 kernels beside this one read and write them).
// Block = 4 compute waves (row tile w x four 32-column tiles: every A fragment feeds 4 x 5 MFMAs) + 1 producer wave, two LDS buffers, one barrier per chunk.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>
#include <vector>

#include "../../include/vits.h"
#include "kernels.h"
#include "model_file.h"

namespace vits {

// (helpers)
typedef float cs_floatx16 __attribute__((ext_vector_type(16)));
typedef float cs_float2v __attribute__((ext_vector_type(2)));
typedef int cs_int4v __attribute__((ext_vector_type(4)));
typedef int cs_int2v __attribute__((ext_vector_type(2)));
typedef __bf16 cs_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 cs_bf2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned cs_pack(float a, float b) {  // v_cvt_pk_bf16_f32: round to nearest even, low half = a
    cs_float2v f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, cs_bf2v));
}
__device__ __forceinline__ float cs_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float cs_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
// four fp32 values -> their three bf16 planes (8 bytes each): piece k = RNE(what pieces < k left), exact remainders
__device__ __forceinline__ void cs_split4(const float* v, cs_int2v* out) {
    float r[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const unsigned a = cs_pack(r[0], r[1]), b = cs_pack(r[2], r[3]);
        out[p] = cs_int2v{(int)a, (int)b};
        r[0] -= cs_lo(a), r[1] -= cs_hi(a), r[2] -= cs_lo(b), r[3] -= cs_hi(b);
    }
}
// (end helpers)

struct ConvSplitParams {
    const uint16_t* x;  // [b][3][cin/8][x_ts][8]
    int64_t x_bs, x_ps;  // batch / plane stride (16-bit elements)
    int x_ts;
    const uint16_t* wp;  // [row tile][step][2][64][8]
    const float* bias;
    const int* len_in;
    const int* len_out;
    int t_in, t_out;
    int cin, cout, nchunks;
    int pad_l;
    // epilogue, standard fp32 layout (conv_mfma.hip EPI_STD): y optional
    float* y;
    int64_t y_bs;
    int y_cs;
    const float* res;
    int64_t r_bs;
    int r_cs;
    const float* acc;
    int64_t a_bs;
    int a_cs;
    float scale;
    int scale_div;
    int post_act;  // 2: leaky_relu(post_slope) of the stored value
    float post_slope;
    // the three planes of leaky_relu(ys_slope) of the (un-activated) result, for the next conv; optional
    uint16_t* ys;
    int64_t ys_bs, ys_ps;
    int ys_ts;
    float ys_slope;
};

// WM: 32-row tiles per block. 4: 128 rows x 128 columns (c_out multiples of 128). 2: 64 rows x 256 columns (C = 64: the four compute waves are two row tiles x two
// 128-column halves, so that an A fragment still feeds 4 x 5 MFMAs — at 128 columns the two planes of weight fragments would be 51 B / clock / CU from L2, the
// whole L2 -> CU path); its two LDS buffers hold a 64-channel input completely (120 KB: one block per CU).
template <int KT, int DIL, int WM>
__global__ __launch_bounds__(320) void conv_split_kernel(const ConvSplitParams p) {
    constexpr int WN = 4 / WM, BN = 128 * WN, NR = 4, STEPS = KT * 2;
    constexpr int XWP = (BN + (KT - 1) * DIL + 7) / 8 * 8;  // slots per LDS group row
    constexpr int BUF = 3 * 4 * XWP;                         // slots per buffer: [plane][group][XWP]
    extern __shared__ __attribute__((aligned(16))) cs_int4v xs[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int b = blockIdx.z, t0 = blockIdx.x * BN;
    const int len_in = p.len_in ? p.len_in[b] : p.t_in;
    const int ncols = p.len_out ? p.len_out[b] : p.t_out;
    if (t0 >= ncols || len_in <= 0) return;
    const int tile_start = t0 - p.pad_l;
    const uint16_t* xb = p.x + (int64_t)b * p.x_bs;

    if (wid == 4) {
        // ---- producer wave (conv16.hip's protocol): fill(0); B; for c: { fill(c + 1); B } ----
        constexpr int NMP = 5;  // 64-slot pieces per group row (XWP <= 312). NOT (XWP + 63) / 64: with an array whose size depends on the template parameters
                                // captured by the DMA lambda below, hipcc 7.2 silently drops the kernel's HOST definition (undefined symbol at load)
        static_assert(XWP <= 64 * NMP, "pieces");
        const bool interior = tile_start >= 0 && tile_start + XWP <= len_in;
        int voff[NMP];
        bool oob[NMP];
#pragma unroll
        for (int m = 0; m < NMP; ++m) {
            const int t = tile_start + lane + 64 * m;
            const int tc = t < 0 ? 0 : (t < len_in ? t : len_in - 1);
            voff[m] = tc * 16;
            oob[m] = t != tc;
        }
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(xb), 0, 0x7fffffff, 0x00020000);
        auto issue = [&](int c, int buf) __attribute__((always_inline)) {
            cs_int4v* lbase = xs + buf * BUF;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const unsigned soff = (unsigned)(((int64_t)pl * p.x_ps + (int64_t)(c * 4 + g) * p.x_ts * 8) * 2);
#pragma unroll
                    for (int m = 0; m < NMP; ++m)
                        if (64 * m + lane < XWP)
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(lbase + (pl * 4 + g) * XWP + 64 * m), 16, voff[m], (int)soff, 0, 0);
                }
        };
        auto finish = [&](int buf) __attribute__((always_inline)) {
            if (!interior) {
                cs_int4v* lbase = xs + buf * BUF;
                const cs_int4v z = {0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < 12; ++q)
#pragma unroll
                    for (int m = 0; m < NMP; ++m)
                        if (64 * m + lane < XWP && oob[m]) lbase[q * XWP + 64 * m + lane] = z;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        issue(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        finish(0);
        __syncthreads();
        for (int c = 0; c + 1 < p.nchunks; ++c) {
            issue(c + 1, (c + 1) & 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            finish((c + 1) & 1);
            __syncthreads();
        }
        return;
    }

    // ---- compute waves: row tile wm of the block, four column tiles from column 128 wn ----
    const int wm = wid % WM, wn = wid / WM;
    const int mt = blockIdx.y * WM + wm;
    cs_floatx16 acc[NR];
#pragma unroll
    for (int n = 0; n < NR; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    const int h = lane >> 5;
    typedef const __attribute__((address_space(3))) cs_int4v* LdsV;
    const int lane_slot = h * XWP + 128 * wn + (lane & 31);
    const int total_steps = p.nchunks * STEPS;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.wp), 0, 0x7fffffff, 0x00020000);
    const int wvoff = (int)((((size_t)mt * total_steps * 2) * 64 + lane) * 16);
    auto load_a = [&](int step, int pl) __attribute__((always_inline)) -> cs_int4v {
        return __builtin_bit_cast(cs_int4v, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, (step * 2 + pl) * 1024, 0));
    };
    auto mfma = [&](cs_int4v a, cs_int4v bq, cs_floatx16 c) __attribute__((always_inline)) -> cs_floatx16 {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cs_bf16x8, a), __builtin_bit_cast(cs_bf16x8, bq), c, 0, 0, 0);
    };
    // ring of 4 A-fragment sets (two planes each), fetched two steps ahead; a chunk has 2 KT steps, so for odd KT the ring phase of a chunk's first
    // step alternates between 0 and 2 (conv16.hip: the chunk loop walks pairs of chunks, both phases straight-line code)
    cs_int4v ring[4][2];
    ring[0][0] = load_a(0, 0), ring[0][1] = load_a(0, 1);
    ring[1][0] = load_a(total_steps > 1 ? 1 : 0, 0), ring[1][1] = load_a(total_steps > 1 ? 1 : 0, 1);
    int gstep = 0;
    auto compute_chunk = [&](LdsV xbase, auto base_c) __attribute__((always_inline)) {
        constexpr int BASE = decltype(base_c)::value;
#pragma unroll
        for (int j = 0; j < KT; ++j) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int s = j * 2 + kk;
                {
                    const int nstep = gstep + 2 < total_steps ? gstep + 2 : total_steps - 1;
                    ring[(BASE + s + 2) & 3][0] = load_a(nstep, 0);
                    ring[(BASE + s + 2) & 3][1] = load_a(nstep, 1);
                }
                LdsV bp = xbase + 2 * kk * XWP + j * DIL;
                cs_int4v bq[3][NR];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int n = 0; n < NR; ++n) bq[pl][n] = bp[pl * 4 * XWP + 32 * n];
                const cs_int4v w1 = ring[(BASE + s) & 3][0], w2 = ring[(BASE + s) & 3][1];
#pragma unroll
                for (int n = 0; n < NR; ++n) {
                    acc[n] = mfma(w1, bq[2][n], acc[n]);  // a3 w1 (the smallest term first)
                    acc[n] = mfma(w2, bq[1][n], acc[n]);  // a2 w2
                    acc[n] = mfma(w2, bq[0][n], acc[n]);  // a1 w2
                    acc[n] = mfma(w1, bq[1][n], acc[n]);  // a2 w1
                    acc[n] = mfma(w1, bq[0][n], acc[n]);  // a1 w1
                }
                ++gstep;
            }
        }
    };
    __syncthreads();
    {
        int buf = 0;
        auto step_buf = [&](int c) __attribute__((always_inline)) {
            buf ^= 1;
            if (c + 1 < p.nchunks) __syncthreads();
        };
        int c = 0;  // (2 KT is never a multiple of 4 for the odd tap counts this kernel is built for: pairs of chunks)
        for (; c + 1 < p.nchunks; c += 2) {
            compute_chunk((LdsV)(xs + buf * BUF + lane_slot), std::integral_constant<int, 0>{});
            step_buf(c);
            compute_chunk((LdsV)(xs + buf * BUF + lane_slot), std::integral_constant<int, 2>{});
            step_buf(c + 1);
        }
        if (c < p.nchunks) compute_chunk((LdsV)(xs + buf * BUF + lane_slot), std::integral_constant<int, 0>{});
    }

    // ---- epilogue: conv_mfma.hip's EPI_STD expressions on the fp32 [b][c][t] tensors; the split planes of the next conv's input ----
    const int colbase = t0 + 128 * wn + (lane & 31);
    float* yb = p.y ? p.y + (int64_t)b * p.y_bs : nullptr;
    const float* rb = p.res ? p.res + (int64_t)b * p.r_bs : nullptr;
    const float* ab = p.acc ? p.acc + (int64_t)b * p.a_bs : nullptr;
    uint16_t* ysb = p.ys ? p.ys + (int64_t)b * p.ys_bs : nullptr;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int ch0 = mt * 32 + 8 * g + 4 * h;  // this lane: channels ch0 .. ch0 + 3 (registers 4 g .. 4 g + 3) of one time step per column tile
        if (ch0 >= p.cout) continue;
        float bias[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) bias[e] = p.bias ? p.bias[ch0 + e] : 0.f;
        float rv[NR][4];
        if (rb) {
#pragma unroll
            for (int n = 0; n < NR; ++n) {
                const int t = colbase + 32 * n;
#pragma unroll
                for (int e = 0; e < 4; ++e) rv[n][e] = t < ncols ? rb[(int64_t)(ch0 + e) * p.r_cs + t] : 0.f;
            }
        }
#pragma unroll
        for (int n = 0; n < NR; ++n) {
            const int t = colbase + 32 * n;
            if (t >= ncols) continue;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = acc[n][4 * g + e] + bias[e];
                if (rb) v[e] = rv[n][e] + v[e];
                if (ab) {
                    v[e] = ab[(int64_t)(ch0 + e) * p.a_cs + t] + v[e];
                    v[e] = p.scale_div ? v[e] / p.scale : v[e] * p.scale;
                }
            }
            if (ysb) {
                float x[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = fmaxf(v[e], v[e] * p.ys_slope);  // slope 1 = identity
                cs_int2v pl[3];
                cs_split4(x, pl);
                uint16_t* dst = ysb + ((int64_t)(ch0 >> 3) * p.ys_ts + t) * 8 + (ch0 & 7);
#pragma unroll
                for (int q = 0; q < 3; ++q) *reinterpret_cast<cs_int2v*>(dst + q * p.ys_ps) = pl[q];
            }
            if (yb) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float o = v[e];
                    if (p.post_act == 2) o = fmaxf(o, o * p.post_slope);
                    yb[(int64_t)(ch0 + e) * p.y_cs + t] = o;
                }
            }
        }
    }
}

// fp32 [b][c][t] -> the three bf16 planes of leaky_relu(slope) of it, group layout (the stage input of the resblocks: the upsampler is an fp32 kernel)
__global__ __launch_bounds__(256) void split_planes_kernel(const float* x, int64_t x_bs, int x_cs, const int* lens, int tmax, float slope, uint16_t* ys, int64_t ys_bs,
                                                           int64_t ys_ps, int ys_ts) {
    const int b = blockIdx.z, g = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    const int len = lens ? lens[b] : tmax;
    if (t >= len) return;
    const float* xb = x + (int64_t)b * x_bs + (int64_t)(g * 8) * x_cs + t;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = xb[(int64_t)e * x_cs];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], v[e] * slope);
    cs_int2v lo[3], hi[3];
    cs_split4(v, lo);
    cs_split4(v + 4, hi);
    uint16_t* dst = ys + (int64_t)b * ys_bs + ((int64_t)g * ys_ts + t) * 8;
#pragma unroll
    for (int q = 0; q < 3; ++q) *reinterpret_cast<cs_int4v*>(dst + q * ys_ps) = cs_int4v{lo[q].x, lo[q].y, hi[q].x, hi[q].y};
}

hipError_t launch_split_planes(TensorRef x, int channels, const int* lens, int batch, int tmax, float slope, Split3Ref out, hipStream_t s) {
    if (!x.p || !out.p || (channels & 7)) return hipErrorInvalidValue;
    dim3 grid((tmax + 255) / 256, channels / 8, batch);
    VITS_KLAUNCH(split_planes_kernel, grid, dim3(256), 0, s, x.p, x.bs, x.cs, lens, tmax, slope, out.p, out.bs, out.ps, out.ts);
    return hipGetLastError();
}

// ---- host side ------------------------------------------------------------------------------------------------------------
// the two bf16 planes of a conv's weights as A fragments [row tile][chunk][tap][k-half][plane][lane][8] (conv16.hip's fragment, per plane). false if a
// weight is not the exact sum of two bf16 values (never for fp16-stored weights; bf16-stored ones have an all-zero second plane)
bool pack_conv_weights_split(const float* w, int cout, int cin, int k, std::vector<uint16_t>& out) {
    const int mtiles = ((cout + 31) / 32 + 3) / 4 * 4, nchunks = (cin + 31) / 32;
    out.assign((size_t)mtiles * nchunks * k * 2 * 2 * 64 * 8, 0);
    for (int mt = 0; mt < mtiles; ++mt)
        for (int c = 0; c < nchunks; ++c)
            for (int j = 0; j < k; ++j)
                for (int kk = 0; kk < 2; ++kk)
                    for (int l = 0; l < 64; ++l)
                        for (int e = 0; e < 8; ++e) {
                            const int ci = c * 32 + (2 * kk + (l >> 5)) * 8 + e, co = mt * 32 + (l & 31);
                            if (ci >= cin || co >= cout) continue;
                            const float v = w[((size_t)co * cin + ci) * k + j];
                            const uint16_t w1 = f32_to_bf16(v);
                            const uint16_t w2 = f32_to_bf16(v - bf16_to_f32(w1));
                            if (bf16_to_f32(w1) + bf16_to_f32(w2) != v) return false;
                            const size_t step = ((size_t)(mt * nchunks + c) * k + j) * 2 + kk;
                            out[((step * 2 + 0) * 64 + l) * 8 + e] = w1;
                            out[((step * 2 + 1) * 64 + l) * 8 + e] = w2;
                        }
    return true;
}

// (C = 64 was tried on the 64 x 256 tile below (WM = 2): 92 TFLOP/s-equivalent against 124 for the fused fp32 pairs — a 60 KB chunk per buffer leaves one block per CU,
// and ONE producer wave's LDS-DMA stream, ~1 KB per 0.12 us, takes as long per chunk as the chunk's MFMAs; the narrow stages keep their fused fp32 kernels)
bool conv_split_candidate(int epi, int kt, int cin, int cout) {
    return epi == EPI_STD && (kt == 3 || kt == 7 || kt == 11) && cin >= 128 && (cin & 31) == 0 && (cout & 127) == 0;
}
bool conv_split_supported(const PackedConv& w, int dil) {
    return w.wps && conv_split_candidate(w.epi, w.kt, w.cin, w.cout) && (dil == 1 || dil == 3 || dil == 5);
}

hipError_t launch_conv_split(const PackedConv& w, const ConvCall& c, hipStream_t s) {
    if (!conv_split_supported(w, c.dil) || !c.xs3.p || c.pre_act || c.y2 || (c.post_act != 0 && c.post_act != 2) || (!c.y.p && !c.ys3.p)) return hipErrorInvalidValue;
    ConvSplitParams p{};
    p.x = c.xs3.p, p.x_bs = c.xs3.bs, p.x_ps = c.xs3.ps, p.x_ts = c.xs3.ts;
    p.wp = w.wps;
    p.bias = w.bias;
    p.len_in = c.len_in, p.len_out = c.len_out, p.t_in = c.t_in, p.t_out = c.t_out;
    p.cin = w.cin, p.cout = w.cout, p.nchunks = w.nchunks;
    p.pad_l = c.pad_l;
    p.y = c.y.p, p.y_bs = c.y.bs, p.y_cs = c.y.cs;
    p.res = c.res.p, p.r_bs = c.res.bs, p.r_cs = c.res.cs;
    p.acc = c.acc.p, p.a_bs = c.acc.bs, p.a_cs = c.acc.cs;
    p.scale = c.scale, p.scale_div = c.scale_div;
    p.post_act = c.post_act, p.post_slope = c.post_slope;
    p.ys = c.ys3.p, p.ys_bs = c.ys3.bs, p.ys_ps = c.ys3.ps, p.ys_ts = c.ys3.ts, p.ys_slope = c.ys3_slope;
    const int wm = 4;  // (the 64-row tile, WM = 2, is not instantiated: see conv_split_candidate)
    const int bn = 128 * (4 / wm);
    dim3 grid((c.t_out + bn - 1) / bn, w.cout / (32 * wm), c.batch);
#define VITS_CS(K, D, M)                                                                                                                          \
    do {                                                                                                                                          \
        constexpr size_t lds = (size_t)2 * 12 * ((128 * (4 / M) + (K - 1) * D + 7) / 8 * 8) * 16;                                                 \
        static BigLdsOnce big;                                                                                                                    \
        if (lds > 64 * 1024 && big.needed()) {                                                                                                    \
            if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_split_kernel<K, D, M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) return e; \
            big.done();                                                                                                                           \
        }                                                                                                                                         \
        VITS_KLAUNCH((conv_split_kernel<K, D, M>), grid, dim3(320), lds, s, p);                                                                   \
        return hipGetLastError();                                                                                                                 \
    } while (0)
#define VITS_CS_D(K, M)                    \
    do {                                   \
        if (c.dil == 1) VITS_CS(K, 1, M);  \
        if (c.dil == 3) VITS_CS(K, 3, M);  \
        VITS_CS(K, 5, M);                  \
    } while (0)
#define VITS_CS_M(K)         \
    do {                     \
        VITS_CS_D(K, 4);     \
    } while (0)
    if (w.kt == 3) VITS_CS_M(3);
    if (w.kt == 7) VITS_CS_M(7);
    VITS_CS_M(11);
#undef VITS_CS_M
#undef VITS_CS_D
#undef VITS_CS
}

}  // namespace vits
