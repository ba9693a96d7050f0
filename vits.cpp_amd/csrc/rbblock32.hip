// rbblock32.hip — one WHOLE HiFiGAN ResBlock with 3-tap convolutions (three conv pairs, dilations 1 / 3 / 5) as a single kernel, exact-fp32
// arithmetic, narrow vocoder stages (C = 32 / 64):
//     y_{p+1} = y_p + Conv_{3,1}( leaky_relu( Conv_{3,D_p}( leaky_relu(y_p) ) + b1_p ) ) + b2_p ,  p = 0, 1, 2      (/root/reference/src/vits.cpp:545-581)
//     out     = [sum of the previous resblocks +] y_3 [ * 1/num_kernels ]                                             (vits.cpp:622-635)
// Why: as three fused pairs (rbpair32.hip) the k = 3 resblocks of these stages move 3 x 12 B per element through HBM and sit at the HBM roof,
// not the MFMA one (C = 32: 0.62 of the fp32 peak at 3.1 TB/s). Here the fp32 stream lives in REGISTERS across the three pairs (MFMA C layout),
// the conv inputs x_p = leaky_relu(y_p) and t_p take turns in ONE fp32 LDS tile, and HBM sees the stage input once (+ halo) and the output
// once: 8-12 B per element and resblock. The price is the halo, 12 columns per side of a 256-column tile (1.10 x the MFMA work) — affordable
// for 3 taps only: 7- and 11-tap resblocks are MFMA-bound in fp32 and stay on pairs (their halo would be 36 / 60 columns per side).
// Same MFMA chain per output (chunk, tap, channel pair; v_mfma_f32_32x32x2_f32) and the same epilogue expressions as rbpair32_kernel /
// conv_mfma_kernel: bit-identical to the pair path (GPU test), which stays behind VITS_NO_RBBLOCK32=1 (Engine::knobs).
// The whole-resblock kernel of the 16-bit modes (rbblock16.hip) is the model; this is its fp32 counterpart for the layers where bytes bind.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdlib>

#include "../../include/vits.h"
#include "kernels.h"

namespace vits {

typedef float rbb32_floatx16 __attribute__((ext_vector_type(16)));
typedef float rbb32_float4v __attribute__((ext_vector_type(4)));

struct RbBlock32Params {
    const float* x;  // stage input y_0 (fp32, [b][c][t])
    int64_t x_bs;
    int x_cs;
    const float* w1[3];
    const float* w2[3];  // packed A fragments (pack_conv_weights) of the six convs
    const float* b1[3];
    const float* b2[3];
    const int* lens;
    int tmax;
    float slope;  // leaky_relu in front of every conv
    float* y;
    int64_t y_bs;
    int y_cs;
    const float* acc;  // resblock sum so far, or null
    int64_t a_bs;
    int a_cs;
    float scale;
    int scale_div;
    int post_act;  // 2: y = leaky_relu(post_slope) of the result (stage output feeding the next upsampler)
    float post_slope;
};

// Block = four waves: C / 32 row tiles x 4 / (C / 32) column strips of NR 32-column tiles; W = 256 tile columns, BO = W - 24 outputs.
template <int C, int NR>
__global__ __launch_bounds__(256, C >= 64 ? 2 : 3) void rbblock32_kernel(const RbBlock32Params p) {
    constexpr int KT = 3, D0 = 1, D1 = 3, D2 = 5;
    constexpr int NCH = C / 32, WM = C / 32, WN = 4 / WM;
    constexpr int W = WN * NR * 32;
    constexpr int P2 = (KT - 1) / 2;
    constexpr int H = P2 * (3 + D0 + D1 + D2);  // halo per side: every pair costs P2 (second conv) + P2 * D_p (first conv)
    constexpr int BO = W - 2 * H;
    constexpr int PADX = P2 * D2;                // the first conv of a pair reads up to P2 * D_p columns beyond a tile column
    constexpr int PITCH = (W + 2 * PADX + 3) / 4 * 4;
    constexpr int TOTAL = NCH * KT * 4;          // A-fragment steps (float4 = 4 MFMA k-steps) per conv and row tile
    static_assert(BO > 0, "tile too narrow");
    extern __shared__ __attribute__((aligned(16))) float tile[];  // [C][PITCH]: x_p, then t_p, then x_{p+1}, ...
    float* lbias = tile + C * PITCH;                              // [pair][b1 | b2][C]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int b = blockIdx.y;
    const int len = p.lens ? p.lens[b] : p.tmax;
    const int t0 = blockIdx.x * BO;
    if (t0 >= len) return;
    const int krow = lane >> 5, col = lane & 31;
    const int u0 = wn * (NR * 32) + col;  // this lane's tile column of column tile nr: u0 + 32 nr
    const int tg0 = t0 - H;               // global time of tile column 0
    typedef const __attribute__((address_space(3))) float* LdsF;
    const float* xb = p.x + (int64_t)b * p.x_bs;

    // ---- the fp32 stream of this wave's 32 rows x NR column tiles in the MFMA C layout: register r <-> row 8 (r / 4) + 4 krow + r % 4 ----
    rbb32_floatx16 yv[NR];
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
        const int t = tg0 + u0 + 32 * nr;
        const bool inside = t >= 0 && t < len;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * 32 + (r >> 2) * 8 + krow * 4 + (r & 3);
            yv[nr][r] = inside ? xb[(int64_t)row * p.x_cs + t] : 0.f;
        }
    }
    // the biases of the six convs and zeros in the tile's margins (read by the first conv of a pair for the outermost tile columns, whose
    // results lie in the halo; never written otherwise)
    for (int i = tid; i < 6 * C; i += 256) {
        const int pi = i / (2 * C), r = i - pi * 2 * C;
        lbias[i] = r < C ? p.b1[pi][r] : p.b2[pi][r - C];
    }
    for (int i = tid; i < C * (PITCH - W); i += 256) {
        const int row = i / (PITCH - W), m = i - row * (PITCH - W);
        tile[row * PITCH + (m < PADX ? m : W + m)] = 0.f;
    }
    // leaky_relu(src [+ bias]) of this wave's rows x columns into the LDS tile, zero outside the sequence (what a conv sees as padding):
    // x_p from the stream (bias = nullptr), t_p from the first conv's accumulators
    auto write_tile = [&](const rbb32_floatx16 (&src)[NR], const float* bias_p) __attribute__((always_inline)) {
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            const int u = u0 + 32 * nr, t = tg0 + u;
            const bool inside = t >= 0 && t < len;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 32 + (r >> 2) * 8 + krow * 4 + (r & 3);
                float v = src[nr][r];
                if (bias_p) v = v + bias_p[row];
                v = fmaxf(v, v * p.slope);
                tile[row * PITCH + PADX + u] = inside ? v : 0.f;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    rbb32_floatx16 acc[NR];
    // Weight fragments (float4 = 4 MFMA k-steps of the wave's row tile) travel through a ring of four register sets, two steps ahead. The first
    // two fragments of a conv are requested BEFORE the barriers and the tile write in front of it (prefetch): fetched at the top of the conv
    // they cost an exposed L2 round trip six times per block.
    rbb32_float4v ring[4];
    const int wvoff = (int)(((size_t)wm * TOTAL * 64 + lane) * 16);
    auto load_a = [&](const float* wp, int step) __attribute__((always_inline)) -> rbb32_float4v {
        const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp), 0, 0x7fffffff, 0x00020000);
        return __builtin_bit_cast(rbb32_float4v, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, step * 1024, 0));
    };
    auto prefetch = [&](const float* wp) __attribute__((always_inline)) {
        ring[0] = load_a(wp, 0);
        ring[1] = load_a(wp, 1 < TOTAL ? 1 : 0);
    };
    // one conv over the LDS tile: output column u reads columns u + off0 + j * dstep. Order per output: chunk, tap, channel pair — conv_mfma.hip's.
    // (ring[0], ring[1] hold the conv's first two fragments on entry)
    auto conv = [&](const float* wp, const int off0, const int dstep) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        LdsF base = (LdsF)(tile + krow * PITCH + PADX + u0 + off0);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                LdsF xj = base + (c * 32) * PITCH + j * dstep;
#pragma unroll
                for (int p4 = 0; p4 < 4; ++p4) {
                    const int s = (c * KT + j) * 4 + p4;  // compile time after unrolling
                    ring[(s + 2) & 3] = load_a(wp, s + 2 < TOTAL ? s + 2 : TOTAL - 1);
                    __builtin_amdgcn_sched_barrier(0);
                    const rbb32_float4v a4 = ring[s & 3];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int pair = p4 * 4 + q;
                        float bv[NR];
#pragma unroll
                        for (int nr = 0; nr < NR; ++nr) bv[nr] = xj[(2 * pair) * PITCH + nr * 32];
#pragma unroll
                        for (int nr = 0; nr < NR; ++nr) acc[nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q], bv[nr], acc[nr], 0, 0, 0);
                    }
                }
            }
        }
    };

    // (on entry the ring holds the first fragments of w1[pi]; on exit those of the next pair's first conv)
    auto pair = [&](const int pi, const int dil, const bool last) __attribute__((always_inline)) {
        conv(p.w1[pi], -P2 * dil, dil);  // conv 1 over x_p: t column u reads x columns u - P2 dil + j dil
        prefetch(p.w2[pi]);
        __syncthreads();                 // every wave is done with x_p: t_p takes its place
        write_tile(acc, lbias + pi * 2 * C);  // t = leaky_relu(conv1 + b1), zero outside the sequence (the second conv's padding)
        __syncthreads();
        conv(p.w2[pi], -P2, 1);  // conv 2 over t_p: y column u reads t columns u - P2 + j
        if (!last) prefetch(p.w1[pi + 1]);
        // the stream: y_{p+1} = y_p + (conv2 + b2)   (rbpair32: v = acc + b2; v = residual + v)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 32 + (r >> 2) * 8 + krow * 4 + (r & 3);
                const float v = acc[nr][r] + lbias[pi * 2 * C + C + row];
                yv[nr][r] = yv[nr][r] + v;
            }
        if (!last) {
            __syncthreads();  // every wave is done with t_p: x_{p+1} takes its place
            write_tile(yv, nullptr);
            __syncthreads();
        }
    };

    prefetch(p.w1[0]);
    __syncthreads();  // biases and margins in place
    write_tile(yv, nullptr);
    __syncthreads();
    pair(0, D0, false);
    pair(1, D1, false);
    pair(2, D2, true);

    // ---- epilogue (as the last pair's in rbpair32_kernel): resblock sum / scale, activation of a stage output, the BO owned columns only ----
    {
        float* yb = p.y + (int64_t)b * p.y_bs;
        const float* ab = p.acc ? p.acc + (int64_t)b * p.a_bs : nullptr;
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            const int u = u0 + 32 * nr, t = tg0 + u;
            if (u < H || u >= H + BO || t >= len) continue;
            float av[16];
            if (ab) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * 32 + (r >> 2) * 8 + krow * 4 + (r & 3);
                    av[r] = ab[(int64_t)row * p.a_cs + t];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 32 + (r >> 2) * 8 + krow * 4 + (r & 3);
                float v = yv[nr][r];
                if (ab) {
                    v = av[r] + v;
                    v = p.scale_div ? v / p.scale : v * p.scale;
                }
                if (p.post_act == 2) v = fmaxf(v, v * p.post_slope);
                yb[(int64_t)row * p.y_cs + t] = v;
            }
        }
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------------
template <int C, int NR>
static hipError_t launch_rbb32(const RbBlock32Params& p, int batch, hipStream_t s) {
    constexpr int W = (4 / (C / 32)) * NR * 32, H = 12, BO = W - 2 * H, PADX = 5, PITCH = (W + 2 * PADX + 3) / 4 * 4;
    const size_t lds = ((size_t)C * PITCH + 6 * C) * sizeof(float);
    static BigLdsOnce big_lds_set;
    if (lds > 64 * 1024 && big_lds_set.needed()) {
        hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbblock32_kernel<C, NR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (ea != hipSuccess) return ea;
        big_lds_set.done();
    }
    dim3 grid((p.tmax + BO - 1) / BO, batch);
    VITS_KLAUNCH((rbblock32_kernel<C, NR>), grid, dim3(256), lds, s, p);
    return hipGetLastError();
}

bool rbblock32_supported(int channels, int kt, const int* dils, int ndil) {
    if (kt != 3 || !(channels == 32 || channels == 64)) return false;
    return ndil == 3 && dils[0] == 1 && dils[1] == 3 && dils[2] == 5;
}

hipError_t launch_rbblock32(const PackedConv* const* c1, const PackedConv* const* c2, const RbBlock32Call& c, hipStream_t s) {
    const int C = c1[0]->cin, kt = c1[0]->kt;
    RbBlock32Params p;
    for (int i = 0; i < 3; ++i) {
        if (!c1[i]->wp || !c2[i]->wp || !c1[i]->bias || !c2[i]->bias || c1[i]->cin != C || c1[i]->cout != C || c2[i]->cin != C || c2[i]->cout != C || c1[i]->kt != kt ||
            c2[i]->kt != kt)
            return hipErrorInvalidValue;
        p.w1[i] = c1[i]->wp;
        p.w2[i] = c2[i]->wp;
        p.b1[i] = c1[i]->bias;
        p.b2[i] = c2[i]->bias;
    }
    const int dils[3] = {1, 3, 5};
    if (!rbblock32_supported(C, kt, dils, 3) || !c.x.p || !c.y.p || c.x.p == c.y.p) return hipErrorInvalidValue;
    p.x = c.x.p;
    p.x_bs = c.x.bs;
    p.x_cs = c.x.cs;
    p.lens = c.lens;
    p.tmax = c.tmax;
    p.slope = c.slope;
    p.y = c.y.p;
    p.y_bs = c.y.bs;
    p.y_cs = c.y.cs;
    p.acc = c.acc.p;
    p.a_bs = c.acc.bs;
    p.a_cs = c.acc.cs;
    p.scale = c.scale;
    p.scale_div = c.scale_div;
    p.post_act = c.post_act;
    p.post_slope = c.post_slope;
    // (C = 32 on 384-column tiles — 1.07 x instead of 1.10 x the MFMA work, two blocks per CU instead of three — measured 1.25 against 1.22 ms)
    if (C == 32) return launch_rbb32<32, 2>(p, c.batch, s);
    return launch_rbb32<64, 4>(p, c.batch, s);
}

}  // namespace vits
