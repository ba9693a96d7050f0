// conv16.hip — Conv1d / ConvTranspose1d with 16-BIT OPERANDS (fp16 or bf16) and fp32 accumulation on the gfx950 matrix
// cores: v_mfma_f32_32x32x16_f16 / v_mfma_f32_32x32x16_bf16 (16x the rate of the exact-fp32 path in conv_mfma.hip).
//
// This is the reference's own conv arithmetic: conv1d_impl unfolds the input with ggml_im2col_1d into
// an fp16 tensor and multiplies it with fp16 weights, fp32 accumulate (/root/reference/src/include/custom-ops.h:680-694;
// weights cast by /root/reference/scripts/export_vits.py:87); ggml_conv_transpose_1d (src/vits.cpp:188) converts its source to
// fp16 the same way. VITS_ARITH_F16 reproduces it, VITS_ARITH_BF16 is the bf16 variant BASELINE.json configs[4] names.
// Products of two 16-bit values are exact in fp32, so against the CPU oracle in the same mode only the fp32 summation order
// differs.
//
// Data layout of a 16-bit activation ("group layout"): [batch][channel/8][time][8] — the 8 channels of a group are one
// 16-byte slot per time step. That is exactly one lane's B operand of the 32x32x16 MFMA (lane l: column l&31, k = 8*(l>>5) + 0..7),
// so a tile row [group][time] streams from HBM into LDS with LDS-DMA (buffer_load_dwordx4 ... lds: 64 lanes = 64 consecutive
// time steps = 1 KiB, lane-linear) and every tap reads its operand with ONE conflict-free ds_read_b128 at a shifted slot. The
// same layout in fp32 ([batch][channel/8][time][8] floats) carries the vocoder's residual stream, so that the epilogue
// reads/writes 4 consecutive channels of one time step per lane (dwordx4 / dwordx2) straight from the MFMA C layout, with no
// transpose. Weights are pre-packed at load in A-fragment order [row tile][chunk 32 c_in][tap][k-half][lane][8].
//
// Block = 4 compute waves + 1 producer wave (as conv_mfma.hip: the compute waves' only memory traffic is the L2-resident
// weight stream; the producer feeds LDS and fixes up the sequence ends), 2-3 LDS buffers, one barrier per 32-channel chunk.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "../../include/vits.h"
#include "kernels.h"
#include "model_file.h"

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

// round-to-nearest-even pair conversion (v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32), low half = a
template <bool BF>
__device__ __forceinline__ unsigned pack16(float a, float b) {
    float2v f = {a, b};
    if constexpr (BF) return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf2v));
    else return __builtin_bit_cast(unsigned, __builtin_convertvector(f, half2v));
}
template <bool BF>
__device__ __forceinline__ float unpack16(unsigned short h) {
    if constexpr (BF) return __builtin_bit_cast(float, (unsigned)h << 16);
    else return (float)__builtin_bit_cast(_Float16, h);
}

enum Epi16 : int { E16_STD = 0, E16_GATE = 1, E16_CONVT = 2, E16_GROUP = 3, E16_CONVT_GROUP = 4 };

struct Conv16Params {
    const uint16_t* x;  // group layout [b][cin/8][x_ts][8]
    int64_t x_bs;       // batch stride in 16-bit elements
    int x_ts;           // time stride (slots per group row)
    const uint16_t* wp;
    const float* bias;
    const int* len_in;
    const int* len_out;
    int t_in, t_out;
    int cin, cout, rows, nchunks;
    int dil, pad_l, lds_off, xwp, nbuf;
    int post_act;  // standard outputs: 1 relu, 2 leaky_relu(post_slope) of the stored value
    float post_slope;
    float scale;
    int scale_div;
    int ct_stride, ct_crop;
    // standard layout fp32 [b][c][t] (E16_STD / E16_GATE / E16_CONVT)
    float* y;
    int64_t y_bs;
    int y_cs;
    float* y2;  // optional leaky_relu(post_slope) copy of y
    const float* res;
    int64_t r_bs;
    int r_cs;
    const float* acc;
    int64_t a_bs;
    int a_cs;
    // group layout (E16_GROUP / E16_CONVT_GROUP): fp32 residual stream + 16-bit conv input for the next layer
    float* yg;
    const float* resg;
    const float* accg;
    int64_t g_bs;  // floats
    int g_ts;
    uint16_t* y16;
    int64_t y16_bs;
    int y16_ts;
    float y16_slope;  // leaky_relu applied to the 16-bit copy (1 = none)
};

template <int KT, int DIL, int WM, int WN, int MR, int NR, int EPI, bool BF>
__global__ __launch_bounds__(320) void conv16_kernel(const Conv16Params p) {
    constexpr int BN = WN * NR * 32;
    constexpr int ADIL = DIL < 0 ? -DIL : DIL;
    constexpr int STEPS = KT * 2;  // MFMA k-steps (16 channels each) per 32-channel chunk
    extern __shared__ __attribute__((aligned(16))) int4v xs16[];  // [buf][4 groups][xwp] slots of 8 x 16 bit

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * BN;
    const int len_in = p.len_in ? p.len_in[b] : p.t_in;
    int ncols;
    if (EPI == E16_CONVT || EPI == E16_CONVT_GROUP) ncols = len_in + 1;
    else ncols = p.len_out ? p.len_out[b] : p.t_out;
    if (t0 >= ncols || len_in <= 0) return;

    const int dil = DIL != 0 ? DIL : p.dil;  // signed step between taps in slots (transposed conv: -1)
    // slots per LDS group row: a compile-time constant with a compile-time dilation, so that every ds_read offset of the K loop
    // is an instruction immediate (the host computes the same value into p.xwp)
    constexpr int XWPC = (BN + (KT - 1) * ADIL + 7) / 8 * 8;
    const int xwp = DIL != 0 ? XWPC : p.xwp;
    const int nbuf = DIL != 0 ? p.nbuf : 2;  // (three buffers need the per-fill DMA count as an immediate, see the producer)
    const int tile_start = t0 - p.pad_l - p.lds_off;
    const int bufslots = 4 * xwp;
    const uint16_t* xb = p.x + (int64_t)b * p.x_bs;
    const int ngroups = (p.cin + 7) >> 3;

    if (wid == 4) {
        // ------------------------------- producer wave ------------------------------------------------------------
        // protocol as conv_mfma.hip: fill(0); B0; for c: { fill(c+1); B(c+1) } — the last chunk has no barrier.
        constexpr int NMP = 6;  // 64-slot pieces per group row (xwp <= 384)
        const int npieces = (xwp + 63) >> 6;
        const bool interior = tile_start >= 0 && tile_start + xwp <= len_in;
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
            int4v* lbase = xs16 + buf * bufslots;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int G = c * 4 + g;
                G = G < ngroups ? G : ngroups - 1;  // (a group past the last channel is zeroed in finish())
                const unsigned soff = (unsigned)G * (unsigned)p.x_ts * 16u;
#pragma unroll
                for (int m = 0; m < NMP; ++m)
                    if (m < npieces && 64 * m + lane < xwp)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(lbase + g * xwp + 64 * m), 16, voff[m], (int)soff, 0, 0);
            }
        };
        auto finish = [&](int c, int buf) __attribute__((always_inline)) {
            int4v* lbase = xs16 + buf * bufslots;
            const bool tail_groups = (c + 1) * 4 > ngroups;
            if (!interior || tail_groups) {
                const int4v z = {0, 0, 0, 0};
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const bool gbad = c * 4 + g >= ngroups;
#pragma unroll
                    for (int m = 0; m < NMP; ++m)
                        if (m < npieces && 64 * m + lane < xwp && (gbad || oob[m])) lbase[g * xwp + 64 * m + lane] = z;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        if (nbuf == 2) {
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
            // three buffers: the DMA runs TWO chunks ahead (short chunks: a chunk is less MFMA time than one HBM round trip).
            // vmcnt retires in order: "at most one fill's worth of requests outstanding" means the OLDER fill has landed.
            constexpr int NI = 4 * ((XWPC + 63) / 64);  // DMA instructions of one fill (every piece has at least one live lane)
            const int n = p.nchunks;
            issue(0, 0);
            if (n > 1) issue(1, 1);
            if (n > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            finish(0, 0);
            __syncthreads();
            int b1 = 1, b2 = 2;
            for (int c = 0; c + 1 < n; ++c) {
                const bool more = c + 2 < n;
                if (more) issue(c + 2, b2);  // last read during chunk c - 1, which every compute wave left before B(c)
                if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                finish(c + 1, b1);
                __syncthreads();
                b1 = b2;
                b2 = b2 == 2 ? 0 : b2 + 1;
            }
        }
        return;
    }

    // ------------------------------- compute waves ---------------------------------------------------------------
    const int wm = wid / WN, wn = wid % WN;
    const int mt0 = (blockIdx.y * WM + wm) * MR;
    floatx16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int h = lane >> 5;
    // B operand of this lane: slot [group 2*kk + h][col + tap*dil], col = wn*NR*32 + nr*32 + (lane & 31) + lds_off
    typedef const __attribute__((address_space(3))) int4v* LdsV;
    const int lane_slot = h * xwp + wn * (NR * 32) + (lane & 31) + p.lds_off;
    // A fragments: 1 KiB per (row tile, chunk, tap, k-half), through a buffer descriptor (no VALU address arithmetic in the loop)
    const size_t tile_frags = (size_t)p.nchunks * STEPS;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.wp), 0, 0x7fffffff, 0x00020000);
    int wvoff[MR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) wvoff[mr] = (int)(((size_t)(mt0 + mr) * tile_frags * 64 + lane) * 16);
    const int total_steps = p.nchunks * STEPS;
    auto load_a = [&](int mr, int step) __attribute__((always_inline)) -> int4v {
        return __builtin_bit_cast(int4v, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff[mr], step * 1024, 0));
    };
    auto mfma = [&](int4v a, int4v bq, floatx16 c) __attribute__((always_inline)) -> floatx16 {
        if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bq), c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, bq), c, 0, 0, 0);
    };

    // ring of 4 A-fragment sets, fetched 2 steps ahead; slot = global step mod 4. A chunk has 2*KT steps, so for odd KT the
    // slot of a chunk's first step alternates between 0 and 2: two compiled variants of the chunk body (BASE = 0 / 2).
    int4v ring[4][MR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
        ring[0][mr] = load_a(mr, 0);
        ring[1][mr] = load_a(mr, total_steps > 1 ? 1 : 0);
    }
    int gstep = 0;
    auto compute_chunk = [&](LdsV xbase, auto base_c) __attribute__((always_inline)) {
        constexpr int BASE = decltype(base_c)::value;
        int4v b_nxt[NR];
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) b_nxt[nr] = xbase[nr * 32];
#pragma unroll
        for (int j = 0; j < KT; ++j) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int s = j * 2 + kk;  // chunk-local step (compile time after unrolling)
                {
                    const int nstep = gstep + 2 < total_steps ? gstep + 2 : total_steps - 1;
#pragma unroll
                    for (int mr = 0; mr < MR; ++mr) ring[(BASE + s + 2) & 3][mr] = load_a(mr, nstep);
                }
                __builtin_amdgcn_sched_barrier(0);
                int4v b_cur[NR];
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) b_cur[nr] = b_nxt[nr];
                {
                    // next step's operand: the other k-half of this tap, or k-half 0 of the next tap (past the last tap this reads
                    // a slot a little further right: still inside the tile, value unused)
                    const int noff = kk == 0 ? 2 * xwp + j * dil : (j + 1) * dil;
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr) b_nxt[nr] = xbase[noff + nr * 32];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr) acc[mr][nr] = mfma(ring[(BASE + s) & 3][mr], b_cur[nr], acc[mr][nr]);
                ++gstep;
            }
        }
    };

    __syncthreads();
    {
        int buf = 0;
        auto step_buf = [&](int c) __attribute__((always_inline)) {
            buf = buf + 1 == nbuf ? 0 : buf + 1;
            if (c + 1 < p.nchunks) __syncthreads();
        };
        if constexpr ((STEPS & 3) == 0) {
            for (int c = 0; c < p.nchunks; ++c) {
                compute_chunk((LdsV)(xs16 + buf * bufslots + lane_slot), std::integral_constant<int, 0>{});
                step_buf(c);
            }
        } else {
            // odd tap count: chunks alternate between ring phases 0 and 2 — the loop walks PAIRS of chunks so that both
            // variants are straight-line code (a per-chunk branch between them makes hipcc shuttle the accumulators through
            // copies at every join)
            int c = 0;
            for (; c + 1 < p.nchunks; c += 2) {
                compute_chunk((LdsV)(xs16 + buf * bufslots + lane_slot), std::integral_constant<int, 0>{});
                step_buf(c);
                compute_chunk((LdsV)(xs16 + buf * bufslots + lane_slot), std::integral_constant<int, 2>{});
                step_buf(c + 1);
            }
            if (c < p.nchunks) compute_chunk((LdsV)(xs16 + buf * bufslots + lane_slot), std::integral_constant<int, 0>{});
        }
    }

    // ---- epilogue ---------------------------------------------------------------------------------------------------
    const int colbase = t0 + wn * (NR * 32) + (lane & 31);
    const int rowoff = 4 * h;
    if constexpr (EPI == E16_GROUP) {
        // group layout: this lane owns channels ch0..ch0+3 of one time step per (mr, g): 16-byte fp32 / 8-byte 16-bit accesses
        float* yg = p.yg ? p.yg + (int64_t)b * p.g_bs : nullptr;
        const float* rg = p.resg ? p.resg + (int64_t)b * p.g_bs : nullptr;
        const float* ag = p.accg ? p.accg + (int64_t)b * p.g_bs : nullptr;
        uint16_t* y16 = p.y16 ? p.y16 + (int64_t)b * p.y16_bs : nullptr;
        // The residual of group it + 2 is fetched BEFORE group it is stored (a ring of three): the residual may alias the output
        // (in-place update of the stream), so the compiler cannot hoist a later group's loads above an earlier group's stores by
        // itself, and every group would otherwise be one load -> wait -> store round trip to HBM (measured: that, not HBM bandwidth,
        // bounded the C >= 128 layers). The accumulator of the resblock sum (2 of 18 convs per stage) is read in place.
        constexpr int NG = MR * 4;
        float4v rv[3][NR];
        auto load_res = [&](int it, float4v* dst) __attribute__((always_inline)) {
            const int ch0 = (mt0 + it / 4) * 32 + 8 * (it & 3) + rowoff;
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                const int t = colbase + nr * 32;
                dst[nr] = float4v{0.f, 0.f, 0.f, 0.f};
                if (rg && ch0 < p.cout && t < ncols) dst[nr] = *reinterpret_cast<const float4v*>(rg + ((int64_t)(ch0 >> 3) * p.g_ts + t) * 8 + (ch0 & 7));
            }
        };
        load_res(0, rv[0]);
        if (NG > 1) load_res(1, rv[1]);
#pragma unroll
        for (int it = 0; it < NG; ++it) {
            const int mr = it / 4, g = it & 3;
            if (it + 2 < NG) load_res(it + 2, rv[(it + 2) % 3]);
            __builtin_amdgcn_sched_barrier(0);
            const int ch0 = (mt0 + mr) * 32 + 8 * g + rowoff;
            if (ch0 >= p.cout) continue;
            float4v bias = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) bias = *reinterpret_cast<const float4v*>(p.bias + ch0);
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                const int t = colbase + nr * 32;
                if (t >= ncols) continue;
                const int64_t go = ((int64_t)(ch0 >> 3) * p.g_ts + t) * 8 + (ch0 & 7);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[mr][nr][4 * g + e] + bias[e];
                    if (p.post_act == 1) v[e] = v[e] > 0.f ? v[e] : 0.f;
                    if (rg) v[e] = rv[it % 3][nr][e] + v[e];
                }
                if (ag) {
                    const float4v a4 = *reinterpret_cast<const float4v*>(ag + go);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = a4[e] + v[e];
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
    } else if constexpr (EPI == E16_STD) {
        float* yb = p.y + (int64_t)b * p.y_bs;
        float* y2b = p.y2 ? p.y2 + (int64_t)b * p.y_bs : nullptr;
        const float* rb = p.res ? p.res + (int64_t)b * p.r_bs : nullptr;
        const float* ab = p.acc ? p.acc + (int64_t)b * p.a_bs : nullptr;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
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
    } else if constexpr (EPI == E16_GATE) {
        // packed tile 2i = tanh rows, 2i+1 = sigmoid rows (as conv_mfma.hip): MR == 2
        float* yb = p.y + (int64_t)b * p.y_bs;
        const int half = p.cout / 2;
        const int chbase = (mt0 / 2) * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = chbase + (r & 3) + 8 * (r >> 2) + rowoff;
            if (ch >= half) continue;
            const float b0 = p.bias ? p.bias[ch] : 0.f, b1 = p.bias ? p.bias[ch + half] : 0.f;
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                const int t = colbase + nr * 32;
                if (t >= ncols) continue;
                yb[(int64_t)ch * p.y_cs + t] = wavenet_gate(acc[0][nr][r] + b0, acc[MR - 1][nr][r] + b1);
            }
        }
    } else {  // transposed conv: GEMM row rho = phase * cout + co (phase-major), column q; output sample n = s*q + phase - crop
        const int s = p.ct_stride;
        const int out_len = p.len_out ? p.len_out[b] : p.t_out;
        if constexpr (EPI == E16_CONVT) {
            float* yb = p.y + (int64_t)b * p.y_bs;
            float* y2b = p.y2 ? p.y2 + (int64_t)b * p.y_bs : nullptr;
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rho = (mt0 + mr) * 32 + (r & 3) + 8 * (r >> 2) + rowoff;
                    if (rho >= p.rows) continue;
                    const int ph = rho / p.cout, co = rho - ph * p.cout;
                    const float bias = p.bias ? p.bias[co] : 0.f;
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr) {
                        const int q = colbase + nr * 32;
                        const int n = s * q + ph - p.ct_crop;
                        if (q >= ncols || n < 0 || n >= out_len) continue;
                        const float o = acc[mr][nr][r] + bias;
                        yb[(int64_t)co * p.y_cs + n] = o;
                        if (y2b) y2b[(int64_t)co * p.y_cs + n] = fmaxf(o, o * p.post_slope);
                    }
                }
        } else {
            // group layout: registers 4g..4g+3 = channels co0..co0+3 of ONE output sample (cout is a multiple of 8)
            float* yg = p.yg ? p.yg + (int64_t)b * p.g_bs : nullptr;
            uint16_t* y16 = p.y16 ? p.y16 + (int64_t)b * p.y16_bs : nullptr;
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int rho0 = (mt0 + mr) * 32 + 8 * g + rowoff;
                    if (rho0 >= p.rows) continue;
                    const int ph = rho0 / p.cout, co0 = rho0 - ph * p.cout;
                    float4v bias = {0.f, 0.f, 0.f, 0.f};
                    if (p.bias) bias = *reinterpret_cast<const float4v*>(p.bias + co0);
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr) {
                        const int q = colbase + nr * 32;
                        const int n = s * q + ph - p.ct_crop;
                        if (q >= ncols || n < 0 || n >= out_len) continue;
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[mr][nr][4 * g + e] + bias[e];
                        if (yg) *reinterpret_cast<float4v*>(yg + ((int64_t)(co0 >> 3) * p.g_ts + n) * 8 + (co0 & 7)) = float4v{v[0], v[1], v[2], v[3]};
                        if (y16) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * p.y16_slope);
                            int2v w2;
                            w2.x = (int)pack16<BF>(v[0], v[1]);
                            w2.y = (int)pack16<BF>(v[2], v[3]);
                            *reinterpret_cast<int2v*>(y16 + ((int64_t)(co0 >> 3) * p.y16_ts + n) * 8 + (co0 & 7)) = w2;
                        }
                    }
                }
        }
    }
}

#if !defined(VITS_CONV16_PART) || (VITS_CONV16_PART == 0 && !VITS_CONV16_BF)
// ---- small kernels of the 16-bit path -----------------------------------------------------------------------------------
// fp32 [b][c][t] -> 16-bit group layout, with the consumer conv's input leaky_relu fused (slope 1 = none): the reference's
// "leaky_relu node, then fp16 im2col" (vits.cpp:554,567,613 + custom-ops.h:684-690) as one pass.
template <bool BF>
__global__ void to_group16_kernel(const float* __restrict__ x, int64_t x_bs, int x_cs, const int* lens, int tmax, int channels, float slope, uint16_t* y,
                                  int64_t y_bs, int y_ts) {
    const int b = blockIdx.z, g = blockIdx.y;
    const int t = blockIdx.x * 64 + threadIdx.x;
    const int len = lens ? lens[b] : tmax;
    if (t >= len) return;
    const float* xb = x + (int64_t)b * x_bs + (int64_t)(g * 8) * x_cs + t;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float f = g * 8 + e < channels ? xb[(int64_t)e * x_cs] : 0.f;
        v[e] = fmaxf(f, f * slope);
    }
    int4v o;
    o.x = (int)pack16<BF>(v[0], v[1]);
    o.y = (int)pack16<BF>(v[2], v[3]);
    o.z = (int)pack16<BF>(v[4], v[5]);
    o.w = (int)pack16<BF>(v[6], v[7]);
    *reinterpret_cast<int4v*>(y + (int64_t)b * y_bs + ((int64_t)g * y_ts + t) * 8) = o;
}

hipError_t launch_to_group16(TensorRef x, const int* lens, int batch, int channels, int tmax, float slope, Ref16 y, int arith, hipStream_t s) {
    if (tmax <= 0 || batch <= 0) return hipSuccess;
    dim3 grid((tmax + 63) / 64, (channels + 7) / 8, batch);
    if (arith == VITS_ARITH_BF16) VITS_KLAUNCH(to_group16_kernel<true>, grid, dim3(64), 0, s, x.p, x.bs, x.cs, lens, tmax, channels, slope, y.p, y.bs, y.ts);
    else VITS_KLAUNCH(to_group16_kernel<false>, grid, dim3(64), 0, s, x.p, x.bs, x.cs, lens, tmax, channels, slope, y.p, y.bs, y.ts);
    return hipGetLastError();
}

// conv_post on the 16-bit copy of the last vocoder stage (already activated with the final slope): Conv(C -> 1, k) -> tanh
// (vits.cpp:638-642). Weights are rounded to the arithmetic type on load into LDS, eight channels of a tap per 16-byte entry — the shape of an
// activation slot —, and a slot x entry product is four v_dot2c_f32_{f16,bf16} (two exact 16-bit products added into the fp32 sum each): 56
// ds_read_b128 and 112 dot instructions per sample where one LDS read, one conversion and one FMA per product were 700.
template <bool BF>
__global__ __launch_bounds__(256) void conv_post16_kernel(const uint16_t* __restrict__ x, int64_t x_bs, int x_ts, const float* __restrict__ w, int cin, int k, float* pre,
                                                          int64_t pre_bs, float* wave, int64_t wave_bs, const int* lens, int tmax, int emit_lo, const int* emit_hi) {
    extern __shared__ __attribute__((aligned(16))) int4v wq[];  // [cin / 8][k]: channels 8 g .. 8 g + 7 at tap j, rounded
    const int b = blockIdx.y;
    const int G = cin >> 3;
    for (int i = threadIdx.x; i < G * k; i += blockDim.x) {
        const int g = i / k, j = i - g * k;
        unsigned u[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = pack16<BF>(w[(g * 8 + 2 * e) * k + j], w[(g * 8 + 2 * e + 1) * k + j]);
        wq[i] = int4v{(int)u[0], (int)u[1], (int)u[2], (int)u[3]};
    }
    // the block's activation slots, [G][256 + k - 1], zero outside the sequence: every slot comes from memory once (coalesced 16-byte loads)
    // instead of once per tap through the L1
    const int len = lens ? lens[b] : tmax;
    const int pad = k / 2;
    const int XS = (int)blockDim.x + k - 1;
    int4v* xs = wq + G * k;
    const uint16_t* xb = x + (int64_t)b * x_bs;
    const int tb = blockIdx.x * blockDim.x - pad;
    if ((int)(blockIdx.x * blockDim.x) < len) {
        for (int i = threadIdx.x; i < G * XS; i += blockDim.x) {
            const int g = i / XS, c = i - g * XS, tt = tb + c;
            int4v q = {0, 0, 0, 0};
            if (tt >= 0 && tt < len) q = *reinterpret_cast<const int4v*>(xb + ((int64_t)g * x_ts + tt) * 8);
            xs[i] = q;
        }
    }
    __syncthreads();
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int hi = emit_hi ? emit_hi[b] : len;
    if (t >= len || t < emit_lo || t >= hi) return;
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
    auto dot2 = [](int xa, int wa, float acc) __attribute__((always_inline)) -> float {
        if constexpr (BF) return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2v, xa), __builtin_bit_cast(bf2v, wa), acc, false);
        else return __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, xa), __builtin_bit_cast(half2v, wa), acc, false);
    };
    float a = 0.f;
    for (int g = 0; g < G; ++g)
        for (int j = 0; j < k; ++j) {
            const int tt = t + j - pad;
            if (tt < 0 || tt >= len) continue;  // (the slot is zero there; skipped so that the sum has the terms, and the order, of a whole run)
            const int4v q = xs[g * XS + (int)threadIdx.x + j];
            const int4v wv = wq[g * k + j];
            a = dot2(q.x, wv.x, a);
            a = dot2(q.y, wv.y, a);
            a = dot2(q.z, wv.z, a);
            a = dot2(q.w, wv.w, a);
        }
    if (pre) pre[(int64_t)b * pre_bs + t] = a;
    wave[(int64_t)b * wave_bs + t] = tanhf(a);
}

hipError_t launch_conv_post16(Ref16 x, const float* w, int cin, int k, TensorRef pre, TensorRef wave, const int* lens, int batch, int tmax, int arith, hipStream_t s,
                              int emit_lo, const int* emit_hi) {
    if (tmax <= 0) return hipSuccess;
    dim3 grid((tmax + 255) / 256, batch);
    if (cin & 7) return hipErrorInvalidValue;
    const size_t lds = (size_t)(cin / 8) * (k + 256 + k - 1) * 16;
    if (arith == VITS_ARITH_BF16)
        VITS_KLAUNCH(conv_post16_kernel<true>, grid, dim3(256), lds, s, x.p, x.bs, x.ts, w, cin, k, pre.p, pre.bs, wave.p, wave.bs, lens, tmax, emit_lo, emit_hi);
    else
        VITS_KLAUNCH(conv_post16_kernel<false>, grid, dim3(256), lds, s, x.p, x.bs, x.ts, w, cin, k, pre.p, pre.bs, wave.p, wave.bs, lens, tmax, emit_lo, emit_hi);
    return hipGetLastError();
}

// ---- host side ---------------------------------------------------------------------------------------------------------
// Packed layout: [row tile][chunk][tap][k-half kk][lane][8]; lane l's 8 values = A[row = tile*32 + (l&31)][ci = chunk*32 + (2*kk + (l>>5))*8 + e][tap]
std::vector<uint16_t> pack_conv_weights16(const float* w, int cout, int cin, int k, int epi, int ct_stride, int arith) {
    const int half = cout / 2;
    int rows, kt;
    if (epi == EPI_CONVT) {
        rows = cout * ct_stride;
        kt = k / ct_stride;
    } else {
        rows = cout;
        kt = k;
    }
    int mtiles = (rows + 31) / 32;
    if (epi == EPI_GATE) mtiles = 2 * ((half + 31) / 32);
    mtiles = (mtiles + 3) / 4 * 4;
    const int nchunks = (cin + 31) / 32;
    std::vector<uint16_t> out((size_t)mtiles * nchunks * kt * 2 * 64 * 8, 0);
    auto cvt = [&](float v) -> uint16_t { return arith == VITS_ARITH_BF16 ? f32_to_bf16(v) : f32_to_f16(v); };
    for (int mt = 0; mt < mtiles; ++mt)
        for (int c = 0; c < nchunks; ++c)
            for (int j = 0; j < kt; ++j)
                for (int kk = 0; kk < 2; ++kk)
                    for (int l = 0; l < 64; ++l)
                        for (int e = 0; e < 8; ++e) {
                            const int ci = c * 32 + (2 * kk + (l >> 5)) * 8 + e;
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
                                    // transposed conv, PHASE-MAJOR rows (rho = phase * cout + co; conv_mfma.hip uses co * s + phase): four
                                    // consecutive accumulator registers are then four consecutive CHANNELS of one output sample, which is
                                    // what the group layout stores as one 16-byte access
                                    const int rho = mt * 32 + r;
                                    const int ph = rho / cout, co = rho % cout;
                                    if (ph < ct_stride) v = w[((size_t)ci * cout + co) * k + ph + ct_stride * j];
                                }
                            }
                            out[((((((size_t)mt * nchunks + c) * kt + j) * 2 + kk) * 64 + l) * 8) + e] = cvt(v);
                        }
    return out;
}

#endif  // host part

struct Tile16 {
    int wm, wn, mr, nr;
};
static Tile16 tile16_shape(int tile) {
    switch (tile) {
        case 0: return {2, 2, 2, 4};  // 128 x 256
        case 1: return {1, 4, 2, 2};  // 64 x 256
        case 2: return {1, 4, 1, 2};  // 32 x 256
        case 3: return {1, 4, 2, 1};  // 64 x 128
        case 5: return {2, 2, 2, 2};  // 128 x 128
        case 6: return {4, 1, 1, 4};  // 128 x 128, one row tile per wave: every A fragment feeds 4 MFMAs
        default: return {1, 4, 1, 1};  // 32 x 128
    }
}

// The kernel template is instantiated for ~90 (taps, dilation, tile, epilogue) combinations per operand type, so the Makefile
// compiles this file eight times: VITS_CONV16_BF = 0 / 1 (fp16 / bf16) x VITS_CONV16_PART = 0 (standard-layout epilogues,
// transposed convs, host code), 1 (group epilogue, taps 1/3/5), 2 (taps 7), 3 (taps 11). Each part defines one dispatcher.
#ifndef VITS_CONV16_PART
#define VITS_CONV16_PART 0
#endif
#ifndef VITS_CONV16_BF
#define VITS_CONV16_BF 0
#endif
constexpr bool kBF = VITS_CONV16_BF != 0;

template <int KT, int DIL, int EPI>
static hipError_t launch_tile16(int tile, const Conv16Params& p, int mtiles_used, int ncols_max, int batch, hipStream_t s) {
    const Tile16 ts = tile16_shape(tile);
    const int bn = ts.wn * ts.nr * 32;
    const int bm = ts.wm * ts.mr;
    dim3 grid((ncols_max + bn - 1) / bn, (mtiles_used + bm - 1) / bm, batch);
    const size_t lds = (size_t)p.nbuf * 4 * p.xwp * 16;
#define VITS_LAUNCH16(WM, WN, MR, NR)                                                                                                \
    do {                                                                                                                             \
        static BigLdsOnce big_lds_set;                                                                                 \
        if (lds > 64 * 1024 && big_lds_set.needed()) {                                                       \
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv16_kernel<KT, DIL, WM, WN, MR, NR, EPI, kBF>),     \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                            \
            if (ea != hipSuccess) return ea;                                                                                         \
            big_lds_set.done();                                                                      \
        }                                                                                                                            \
        VITS_KLAUNCH((conv16_kernel<KT, DIL, WM, WN, MR, NR, EPI, kBF>), grid, dim3(320), lds, s, p);                          \
    } while (0)
    // which tiles exist for which epilogue: the standard-layout epilogues (stage one, flow, transparent fallback) only come in the
    // 64- and 32-row tiles; the gate needs MR == 2; run-time-dilation variants skip the 128 x 256 tile
    constexpr bool group = EPI == E16_GROUP || EPI == E16_CONVT_GROUP;
    switch (tile) {
        case 0:
            if constexpr (group && DIL != 0) VITS_LAUNCH16(2, 2, 2, 4);
            else return hipErrorInvalidValue;
            break;
        case 1: VITS_LAUNCH16(1, 4, 2, 2); break;
        case 2:
            if constexpr (group) VITS_LAUNCH16(1, 4, 1, 2);
            else return hipErrorInvalidValue;
            break;
        case 3: VITS_LAUNCH16(1, 4, 2, 1); break;
        case 5:
            if constexpr (group && DIL != 0) VITS_LAUNCH16(2, 2, 2, 2);
            else return hipErrorInvalidValue;
            break;
        case 6:
            if constexpr (group && DIL != 0) VITS_LAUNCH16(4, 1, 1, 4);
            else return hipErrorInvalidValue;
            break;
        default:
            if constexpr (EPI != E16_GATE) VITS_LAUNCH16(1, 4, 1, 1);
            else return hipErrorInvalidValue;
            break;
    }
#undef VITS_LAUNCH16
    return hipGetLastError();
}

// does the 128 x 256 tile exist for this (epilogue, taps, dilation)? (mirrors launch_tile16 / the dispatchers below)
static bool conv16_has_tile0(int epi16, int kt, int dil) {
    if (epi16 == E16_CONVT_GROUP) return true;
    if (epi16 != E16_GROUP) return false;
    if (kt == 1) return true;
    if (kt == 3 || kt == 7 || kt == 11) return dil == 1 || dil == 3 || dil == 5;
    return false;
}

#define VITS_DISPATCH16(NAME) hipError_t NAME(int epi16, int kt, int tile, const Conv16Params& p, int mtiles_used, int ncols_max, int batch, hipStream_t s)
#define VITS_T16(KT, DIL, EPI) return launch_tile16<KT, DIL, EPI>(tile, p, mtiles_used, ncols_max, batch, s)
#if VITS_CONV16_BF
#define VITS_FN16(part) conv16_dispatch_bf16_p##part
#else
#define VITS_FN16(part) conv16_dispatch_f16_p##part
#endif
VITS_DISPATCH16(conv16_dispatch_f16_p0);
VITS_DISPATCH16(conv16_dispatch_f16_p1);
VITS_DISPATCH16(conv16_dispatch_f16_p2);
VITS_DISPATCH16(conv16_dispatch_f16_p3);
VITS_DISPATCH16(conv16_dispatch_bf16_p0);
VITS_DISPATCH16(conv16_dispatch_bf16_p1);
VITS_DISPATCH16(conv16_dispatch_bf16_p2);
VITS_DISPATCH16(conv16_dispatch_bf16_p3);

#if VITS_CONV16_PART == 0
VITS_DISPATCH16(VITS_FN16(0)) {
    if (epi16 == E16_CONVT) {
        if (kt == 2) VITS_T16(2, -1, E16_CONVT);
        return hipErrorInvalidValue;
    }
    if (epi16 == E16_CONVT_GROUP) {
        if (kt == 2) VITS_T16(2, -1, E16_CONVT_GROUP);
        return hipErrorInvalidValue;
    }
    if (epi16 == E16_GATE) {
        if (kt == 5) VITS_T16(5, 0, E16_GATE);
        if (kt == 3) VITS_T16(3, 0, E16_GATE);
        return hipErrorInvalidValue;
    }
    if (epi16 == E16_STD) {
        switch (kt) {
            case 1: VITS_T16(1, 1, E16_STD);
            case 3: VITS_T16(3, 0, E16_STD);
            case 5: VITS_T16(5, 0, E16_STD);
            case 7: VITS_T16(7, 0, E16_STD);
            case 11: VITS_T16(11, 0, E16_STD);
            default: return hipErrorInvalidValue;
        }
    }
    return hipErrorInvalidValue;
}
#endif
#if VITS_CONV16_PART == 1
VITS_DISPATCH16(VITS_FN16(1)) {
    if (epi16 != E16_GROUP) return hipErrorInvalidValue;
    if (kt == 1) VITS_T16(1, 1, E16_GROUP);
    if (kt == 3) {
        if (p.dil == 1) VITS_T16(3, 1, E16_GROUP);
        if (p.dil == 3) VITS_T16(3, 3, E16_GROUP);
        if (p.dil == 5) VITS_T16(3, 5, E16_GROUP);
        VITS_T16(3, 0, E16_GROUP);
    }
    if (kt == 5) VITS_T16(5, 0, E16_GROUP);
    return hipErrorInvalidValue;
}
#endif
#if VITS_CONV16_PART == 2
VITS_DISPATCH16(VITS_FN16(2)) {
    if (epi16 != E16_GROUP || kt != 7) return hipErrorInvalidValue;
    if (p.dil == 1) VITS_T16(7, 1, E16_GROUP);
    if (p.dil == 3) VITS_T16(7, 3, E16_GROUP);
    if (p.dil == 5) VITS_T16(7, 5, E16_GROUP);
    VITS_T16(7, 0, E16_GROUP);
}
#endif
#if VITS_CONV16_PART == 3
VITS_DISPATCH16(VITS_FN16(3)) {
    if (epi16 != E16_GROUP || kt != 11) return hipErrorInvalidValue;
    if (p.dil == 1) VITS_T16(11, 1, E16_GROUP);
    if (p.dil == 3) VITS_T16(11, 3, E16_GROUP);
    if (p.dil == 5) VITS_T16(11, 5, E16_GROUP);
    VITS_T16(11, 0, E16_GROUP);
}
#endif
#undef VITS_T16

#if VITS_CONV16_PART == 0 && !VITS_CONV16_BF
int choose_conv16_tile(int rows, int epi, int ncols_max, int mtiles_used, int batch) {
    int tile;
    const bool small_t = ncols_max <= 128;
    // c_out multiple of 128: 128 x 128 tiles (default) read the input tile once per 128 rows at 3 blocks per CU. 128 x 256 tiles
    // (VITS_T16_TILE0=1) hold one block per CU (196 VGPRs: K loop and epilogue run back to back: 33.6 ms per step); 64 x 256 tiles
    // (VITS_T16_TILE0=2: 29.6 ms) overlap epilogue traffic with MFMAs but fetch the input once per 64 rows.
    const int tile0 = kernel_knobs().t16_tile0;
    if (epi == EPI_GATE) tile = small_t ? 3 : 1;
    else if (rows % 128 == 0) tile = small_t ? 3 : (tile0 == 1 ? 0 : tile0 == 2 ? 1 : tile0 == 3 ? 5 : tile0 == 4 ? 3 : 6);
    else if (rows % 64 == 0) tile = small_t ? 3 : 1;
    else tile = small_t ? 4 : 2;
    // small grids (batch 1, short inputs): step down until the launch has >= 512 blocks
    auto blocks = [&](int tl) {
        const Tile16 t2 = tile16_shape(tl);
        const int64_t nb = (ncols_max + t2.wn * t2.nr * 32 - 1) / (t2.wn * t2.nr * 32);
        const int64_t mb = (mtiles_used + t2.wm * t2.mr - 1) / (t2.wm * t2.mr);
        return nb * mb * batch;
    };
    if (blocks(tile) < 512 && (tile == 0 || tile == 1 || tile == 5 || tile == 6)) tile = 3;
    if (epi != EPI_GATE && blocks(tile) < 512 && (tile == 3 || tile == 2)) tile = 4;
    return tile;
}

hipError_t launch_conv16(const PackedConv& w, const Conv16Call& c, int arith, hipStream_t s) {
    if (!w.wp16) return hipErrorInvalidValue;
    if (conv16_lat_wanted(w, c)) return launch_conv16_lat(w, c, arith, s);  // (small grids of the wide vocoder stages: conv16_lat.hip)
    Conv16Params p;
    p.x = c.x.p;
    p.x_bs = c.x.bs;
    p.x_ts = c.x.ts;
    p.wp = w.wp16;
    p.bias = w.bias;
    p.len_in = c.len_in;
    p.len_out = c.len_out;
    p.t_in = c.t_in;
    p.t_out = c.t_out;
    p.cin = w.cin;
    p.cout = w.cout;
    p.rows = w.rows;
    p.nchunks = w.nchunks;
    p.post_act = c.post_act;
    p.post_slope = c.post_slope;
    p.scale = c.scale;
    p.scale_div = c.scale_div;
    p.ct_stride = w.ct_stride;
    p.ct_crop = c.ct_crop;
    p.y = c.y.p;
    p.y_bs = c.y.bs;
    p.y_cs = c.y.cs;
    p.y2 = c.y2;
    p.res = c.res.p;
    p.r_bs = c.res.bs;
    p.r_cs = c.res.cs;
    p.acc = c.acc.p;
    p.a_bs = c.acc.bs;
    p.a_cs = c.acc.cs;
    p.yg = c.yg;
    p.resg = c.resg;
    p.accg = c.accg;
    p.g_bs = c.g_bs;
    p.g_ts = c.g_ts;
    p.y16 = c.y16.p;
    p.y16_bs = c.y16.bs;
    p.y16_ts = c.y16.ts;
    p.y16_slope = c.y16_slope;
    const bool group = c.yg || c.y16.p;
    int epi16;
    if (w.epi == EPI_CONVT) epi16 = group ? E16_CONVT_GROUP : E16_CONVT;
    else if (w.epi == EPI_GATE) epi16 = E16_GATE;
    else epi16 = group ? E16_GROUP : E16_STD;
    if (epi16 == E16_GROUP && (w.cout & 7)) return hipErrorInvalidValue;
    const int ncols_max = w.epi == EPI_CONVT ? c.t_in + 1 : c.t_out;
    if (w.epi == EPI_CONVT) {
        p.dil = -1;
        p.pad_l = 0;
    } else {
        p.dil = w.kt == 1 ? 1 : c.dil;
        p.pad_l = c.pad_l;
    }
    int tile = c.tile >= 0 ? c.tile : choose_conv16_tile(w.rows, w.epi, ncols_max, w.mtiles_used, c.batch);
    if ((tile == 0 || tile == 5 || tile == 6) && !conv16_has_tile0(epi16, w.kt, p.dil)) tile = 1;
    if (tile == 2 && !(epi16 == E16_GROUP || epi16 == E16_CONVT_GROUP)) tile = 4;
    const Tile16 ts = tile16_shape(tile);
    const int bn = ts.wn * ts.nr * 32;
    const int span = (w.kt - 1) * p.dil;
    p.lds_off = span < 0 ? -span : 0;
    p.xwp = (bn + (span < 0 ? -span : span) + 7) / 8 * 8;
    if (p.xwp > 384) return hipErrorInvalidValue;
    {
        // third LDS buffer where a chunk is less MFMA time than an HBM round trip (~6k cycles): taps x 2 k-halves x MR*NR MFMAs x 32 cycles
        const bool short_chunk = w.kt * 2 * ts.mr * ts.nr * 32 < 6000 && w.nchunks >= 3;
        p.nbuf = short_chunk ? 3 : 2;
        if ((size_t)p.nbuf * 4 * p.xwp * 16 > 150 * 1024) p.nbuf = 2;
    }
    const bool bf = arith == VITS_ARITH_BF16;
    const int part = epi16 != E16_GROUP ? 0 : (w.kt == 7 ? 2 : w.kt == 11 ? 3 : 1);
#define VITS_CALL16(P) (bf ? conv16_dispatch_bf16_p##P(epi16, w.kt, tile, p, w.mtiles_used, ncols_max, c.batch, s) \
                           : conv16_dispatch_f16_p##P(epi16, w.kt, tile, p, w.mtiles_used, ncols_max, c.batch, s))
    switch (part) {
        case 0: return VITS_CALL16(0);
        case 1: return VITS_CALL16(1);
        case 2: return VITS_CALL16(2);
        default: return VITS_CALL16(3);
    }
#undef VITS_CALL16
}
#endif

}  // namespace vits
