// rbpair32.hip — one HiFiGAN ResBlock conv PAIR as a single kernel, exact-fp32 arithmetic, narrow stages (C = 32 / 64):
//     y' = y + Conv_{k,1}( leaky_relu( Conv_{k,d}( leaky_relu(y) ) + b1 ) ) + b2        (/root/reference/src/vits.cpp:545-581)
// In fp32 these stages move 20 B per element and pair through HBM as two launches (conv1 reads y, writes t; conv2 reads t and the
// residual, writes y') and the k = 3 / k = 7 layers sit at the HBM roof, not the MFMA one. Here t never
// leaves the CU: the block streams the input tile in once, applies leaky_relu in place, runs conv1 over the mid columns (output tile
// + the (k-1)-column halo of the second conv), writes t = leaky_relu(conv1 + b1) (zero outside the sequence) over the input tile in
// LDS (nothing reads the input after conv1: the residual comes from memory, where the tile just came from), runs conv2 from there and
// stores y' : 8-12 B per element and pair. Same MFMA chain per output as conv_mfma.hip (chunk, tap, channel pair; v_mfma_f32_32x32x2_f32)
// and the same epilogue expressions: bit-identical to the two-launch path (GPU test), which stays for C >= 128 (MFMA-bound there) and
// behind VITS_NO_FUSE32=1.
//
// Block = 4 waves, no producer wave (the whole tile is one fill: all four waves issue the LDS-DMA): C = 32: one row tile, four
// 64-column strips (256 mid columns); C = 64: two row tiles x two 64-column strips (128 mid columns). 40-47 KB of LDS and < 128 VGPRs:
// three blocks per CU overlap one block's fill and epilogue with the others' MFMAs.
// A fused block reads a halo of its neighbours' input columns while other blocks already store their output: the output must not be
// the input buffer (Engine::run_batch ping-pongs the resblock's stream between two buffers).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdlib>

#include "../../include/vits.h"
#include "kernels.h"

namespace vits {

typedef float rb32_floatx16 __attribute__((ext_vector_type(16)));
typedef float rb32_float4v __attribute__((ext_vector_type(4)));

struct RbPair32Params {
    const float* x;  // raw y (fp32, [b][c][t])
    int64_t x_bs;
    int x_cs;
    const float *w1, *w2;  // packed A fragments (pack_conv_weights)
    const float *b1, *b2;
    const int* lens;
    int tmax;
    float slope;  // leaky_relu in front of both convs
    float* y;
    int64_t y_bs;
    int y_cs;
    const float* acc;  // resblock sum so far (last pair of a resblock), or null
    int64_t a_bs;
    int a_cs;
    float scale;
    int scale_div;
    int post_act;  // 2: y = leaky_relu(post_slope) of the result (stage output feeding the next upsampler)
    float post_slope;
};

template <int KT, int DIL, int C>
__global__ __launch_bounds__(256, C >= 128 ? 2 : 3) void rbpair32_kernel(const RbPair32Params p) {
    constexpr int NCH = C / 32;  // 32-channel chunks
    constexpr int WM = C / 32, WN = 4 / WM, NR = C >= 128 ? 4 : 2;  // (C = 128: four row tiles, every wave all 128 mid columns)
    constexpr int BM = WN * NR * 32;   // mid columns (t) per block
    constexpr int BO = BM - (KT - 1);  // output columns per block
    constexpr int P2 = (KT - 1) / 2, P1 = (KT - 1) * DIL / 2;
    constexpr int XWP = (BM + (KT - 1) * DIL + 3 + 3) / 4 * 4;  // x tile row pitch (floats): + up to 3 columns of alignment shift
    constexpr int XW4 = XWP / 4;
    constexpr int TWP = (BM + KT - 1 + 3) / 4 * 4;  // t tile row pitch
    constexpr int TOTAL = NCH * KT * 4;             // A-fragment steps (float4 = 4 MFMA k-steps) per conv and row tile
    static_assert(TWP <= XWP, "the t tile takes the x tile's place");
    extern __shared__ __attribute__((aligned(16))) float lds[];  // x tile [C][XWP], later t tile [C][TWP]
    float* xs = lds;
    float* ts = lds;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int b = blockIdx.y;
    const int len = p.lens ? p.lens[b] : p.tmax;
    const int t0 = blockIdx.x * BO;
    if (t0 >= len) return;
    const int wm = wid / WN, wn = wid % WN;
    const int cb = wn * (NR * 32);  // first mid column of this wave
    const int krow = lane >> 5;

    // ---- phase 0: the input tile, all channels, straight into LDS; LDS column 0 = global time ts0 (16-byte aligned source) ------
    const int tx0 = t0 - P2 - P1;    // global time of mid-conv tap 0 of mid column 0
    const int ts0 = tx0 & ~3;        // (two's complement: rounds down for negative values too)
    const int shift = tx0 - ts0;
    const float* xb = p.x + (int64_t)b * p.x_bs;
    {
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, 0x7fffffff, 0x00020000);
        const int tlast = (len - 1) & ~3;  // last float4 that starts inside the sequence (rows are padded to multiples of 4)
        constexpr int N4 = C * XW4;
        constexpr int NI = (N4 + 63) / 64;  // 1 KB DMA instructions for the tile
#pragma unroll
        for (int n0 = 0; n0 < NI; n0 += 4) {
            const int n = n0 + wid;
            if (n < NI) {
                int g = n * 64 + lane;
                g = g < N4 ? g : N4 - 1;
                const int r = g / XW4, c4 = g - r * XW4;
                int t = ts0 + 4 * c4;
                t = t < 0 ? 0 : (t > tlast ? tlast : t);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(xs + n * 256), 16, (r * p.x_cs + t) * 4, 0, 0, 0);
            }
        }
    }
    // bias rows of this lane (accumulator register r <-> row 8*(r/4) + 4*(lane/32) + r%4 of the wave's row tile), fetched now
    float bias1[16], bias2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 + (r >> 2) * 8 + krow * 4 + (r & 3);
        bias1[r] = p.b1[row];
        bias2[r] = p.b2[row];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // leaky_relu in place + zero padding outside the sequence (what conv_mfma.hip's producer does per chunk)
    {
        rb32_float4v* x4 = reinterpret_cast<rb32_float4v*>(xs);
        constexpr int N4 = C * XW4;
        for (int g = tid; g < N4; g += 256) {
            const int r = g / XW4, c4 = g - r * XW4;
            const int t = ts0 + 4 * c4;
            rb32_float4v v = x4[g];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float u = v[e];
                u = fmaxf(u, u * p.slope);
                v[e] = (t + e < 0 || t + e >= len) ? 0.f : u;
            }
            x4[g] = v;
        }
    }
    __syncthreads();

    rb32_floatx16 acc[NR];
    typedef const __attribute__((address_space(3))) float* LdsF;

    // one conv over the LDS tile: lane base `base` (row krow, this wave's first column), row pitch `pitch`, tap step `dstep`.
    // Order per output: chunk, tap, channel pair — conv_mfma.hip's.
    auto conv = [&](const float* wp, LdsF base, const int pitch, const int dstep) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp), 0, 0x7fffffff, 0x00020000);
        const int wvoff = (int)(((size_t)wm * TOTAL * 64 + lane) * 16);
        auto load_a = [&](int step) __attribute__((always_inline)) -> rb32_float4v {
            return __builtin_bit_cast(rb32_float4v, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, step * 1024, 0));
        };
        rb32_float4v ring[4];
        ring[0] = load_a(0);
        ring[1] = load_a(1 < TOTAL ? 1 : 0);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                LdsF xj = base + (c * 32) * pitch + j * dstep;
#pragma unroll
                for (int p4 = 0; p4 < 4; ++p4) {
                    const int s = (c * KT + j) * 4 + p4;  // compile time after unrolling
                    ring[(s + 2) & 3] = load_a(s + 2 < TOTAL ? s + 2 : TOTAL - 1);
                    __builtin_amdgcn_sched_barrier(0);
                    const rb32_float4v a4 = ring[s & 3];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int pair = p4 * 4 + q;
                        float bv[NR];
#pragma unroll
                        for (int nr = 0; nr < NR; ++nr) bv[nr] = xj[(2 * pair) * pitch + nr * 32];
#pragma unroll
                        for (int nr = 0; nr < NR; ++nr) acc[nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q], bv[nr], acc[nr], 0, 0, 0);
                    }
                }
            }
        }
    };

    // ---- phase 1: conv1 over the x tile: mid column i reads x columns shift + i + j*DIL ----------------------------------------
    conv(p.w1, (LdsF)(xs + krow * XWP + shift + cb + (lane & 31)), XWP, DIL);
    __syncthreads();  // every wave is done with the x tile: t takes its place

    // ---- phase 2: t = leaky_relu(conv1 + b1), zero outside the sequence (the second conv's padding) -------------------------------
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
        const int i = cb + nr * 32 + (lane & 31);
        const int tm = t0 - P2 + i;
        const bool inside = tm >= 0 && tm < len;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * 32 + (r >> 2) * 8 + krow * 4 + (r & 3);
            float v = acc[nr][r] + bias1[r];
            v = fmaxf(v, v * p.slope);
            ts[row * TWP + i] = inside ? v : 0.f;
        }
    }
    __syncthreads();

    // ---- phase 3: conv2 over the t tile: output column o reads t columns o + j ------------------------------------------------------
    conv(p.w2, (LdsF)(ts + krow * TWP + cb + (lane & 31)), TWP, 1);

    // ---- phase 4: + b2, + residual, resblock sum / scale, activation of a stage output ------------------------------------------------
    {
        float* yb = p.y + (int64_t)b * p.y_bs;
        const float* ab = p.acc ? p.acc + (int64_t)b * p.a_bs : nullptr;
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            const int o = cb + nr * 32 + (lane & 31);
            const int t = t0 + o;
            if (o >= BO || t >= len) continue;
            float rv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 32 + (r >> 2) * 8 + krow * 4 + (r & 3);
                rv[r] = xb[(int64_t)row * p.x_cs + t];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 32 + (r >> 2) * 8 + krow * 4 + (r & 3);
                float v = acc[nr][r] + bias2[r];
                v = rv[r] + v;
                if (ab) {
                    v = ab[(int64_t)row * p.a_cs + t] + v;
                    v = p.scale_div ? v / p.scale : v * p.scale;
                }
                if (p.post_act == 2) v = fmaxf(v, v * p.post_slope);
                yb[(int64_t)row * p.y_cs + t] = v;
            }
        }
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------------
template <int KT, int DIL, int C>
static hipError_t launch_rb32(const RbPair32Params& p, int batch, hipStream_t s) {
    constexpr int WN = 4 / (C / 32), BM = WN * (C >= 128 ? 4 : 2) * 32, BO = BM - (KT - 1);
    constexpr int XWP = (BM + (KT - 1) * DIL + 3 + 3) / 4 * 4;
    const size_t ldsz = ((size_t)C * XWP * sizeof(float) + 1023) / 1024 * 1024;  // (the last 1 KB DMA instruction may overhang the tile)
    static BigLdsOnce big_lds_set;
    if (ldsz > 64 * 1024 && big_lds_set.needed()) {
        hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpair32_kernel<KT, DIL, C>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (ea != hipSuccess) return ea;
        big_lds_set.done();
    }
    dim3 grid((p.tmax + BO - 1) / BO, batch);
    VITS_KLAUNCH((rbpair32_kernel<KT, DIL, C>), grid, dim3(256), ldsz, s, p);
    return hipGetLastError();
}

template <int KT, int C>
static hipError_t launch_rb32_dil(int dil, const RbPair32Params& p, int batch, hipStream_t s) {
    switch (dil) {
        case 1: return launch_rb32<KT, 1, C>(p, batch, s);
        case 3: return launch_rb32<KT, 3, C>(p, batch, s);
        case 5: return launch_rb32<KT, 5, C>(p, batch, s);
        default: return hipErrorInvalidValue;
    }
}

template <int C>
static hipError_t launch_rb32_kt(int kt, int dil, const RbPair32Params& p, int batch, hipStream_t s) {
    switch (kt) {
        case 3: return launch_rb32_dil<3, C>(dil, p, batch, s);
        case 7: return launch_rb32_dil<7, C>(dil, p, batch, s);
        case 11: return launch_rb32_dil<11, C>(dil, p, batch, s);
        default: return hipErrorInvalidValue;
    }
}

bool rbpair32_supported(int channels, int kt, int dil) {
    if (!(kt == 3 || kt == 7 || kt == 11)) return false;
    if (!(dil == 1 || dil == 3 || dil == 5)) return false;
    if (channels == 32 || channels == 64) return true;
    // C = 128: only the k = 3 pairs — their 128-column tile with its small halo is 74 KB (two blocks per CU); k = 7 / 11 would be 84-94 KB
    const bool c128 = kernel_knobs().fuse32_c128;
    return channels == 128 && kt == 3 && c128;
}

hipError_t launch_rbpair32(const PackedConv& c1, const PackedConv& c2, const RbPair32Call& c, hipStream_t s) {
    if (!c1.wp || !c2.wp || c1.cin != c1.cout || c2.cin != c1.cout || c2.cout != c1.cout || c1.kt != c2.kt || !rbpair32_supported(c1.cin, c1.kt, c.dil) ||
        !c1.bias || !c2.bias || c.x.p == c.y.p)
        return hipErrorInvalidValue;
    // 16-byte LDS-DMA: aligned rows
    if ((c.x.cs & 3) || (c.x.bs & 3) || (reinterpret_cast<uintptr_t>(c.x.p) & 15)) return hipErrorInvalidValue;
    RbPair32Params p;
    p.x = c.x.p;
    p.x_bs = c.x.bs;
    p.x_cs = c.x.cs;
    p.w1 = c1.wp;
    p.w2 = c2.wp;
    p.b1 = c1.bias;
    p.b2 = c2.bias;
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
    if (c1.cin == 32) return launch_rb32_kt<32>(c1.kt, c.dil, p, c.batch, s);
    if (c1.cin == 64) return launch_rb32_kt<64>(c1.kt, c.dil, p, c.batch, s);
    return launch_rb32_dil<3, 128>(c.dil, p, c.batch, s);
}

}  // namespace vits
