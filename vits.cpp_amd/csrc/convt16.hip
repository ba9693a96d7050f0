// convt16.hip — the HiFiGAN upsamplers (ConvTranspose1d, kernel = 2 x stride; /root/reference/src/vits.cpp:178-193,613-618) in the
// 16-bit-operand modes as a STREAMING kernel: one block owns a tile of input positions q and computes ALL stride x c_out GEMM rows for it.
//
// conv16_kernel runs this layer as a generic 128 x 128 GEMM tile: 5-wave blocks built for long K loops, one per CU, and for stride 8 the
// 8-16 row blocks of a column tile each stream the same input tile in again and write one phase each of every output line. But the
// transposed conv has almost no K loop (K = 2 c_in: 8-64 steps of 16) and a large output (stride x the input): it is a streaming layer
// — 0.2-0.4 of the HBM roof as a GEMM tile, the largest single entry of the long-form configuration (5.5 of 83 ms, BASELINE config 5).
// Here: four waves, no producer; the whole input tile (all c_in, BN + 1 positions) goes into LDS once with LDS-DMA issued by all waves; the
// waves then compute without another barrier, 2-4 blocks per CU overlap fill, MFMAs and stores. What matters is WHEN the pieces of an
// output line reach L2 (fp32 group layout: one 128-byte line = 4 consecutive samples x 8 channels = 4 phases of one input position):
//   convt16_kernel       stride x c_out <= 128 (the stride-2 upsamplers): every wave ONE row tile; the phases of a line are written
//                        by waves of one block at the same time. 4.4 TB/s against 2.4-2.9 as a GEMM tile.
//   convt16_lines_kernel strides that are multiples of 4 (the stride-8 upsamplers): a wave keeps FOUR phases of its 32-channel block for 64
//                        positions (acc[4][2]) and stores the four phases of a line back to back. A first version that walked the
//                        phases one after the other (a wave's row tiles w, w + 4, ...) wrote the quarters of a line tens of microseconds
//                        apart: the half-written lines left L2 before they were complete and the layer ran at 1.3 TB/s, slower than
//                        the GEMM tile; with whole lines 0.41 against 0.53 ms per step for the two stride-8 upsamplers.
// Same A fragments, same k-order (chunk, tap, k-half), same epilogue expressions as conv16_kernel<2, -1, ..., E16_CONVT_GROUP>:
// bit-identical (GPU test); VITS_NO_CONVT16S=1 keeps the GEMM-tile path, VITS_NO_CONVT16L=1 only for the stride-8 layers.
// Batch 64 x 128 ids, f16: the four upsamplers 1.05 -> 0.79 ms per step; config 5 (bf16): 79.8 -> 76.8 ms.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "../../include/vits.h"
#include "kernels.h"

namespace vits {

namespace ct16 {
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
}  // namespace ct16

struct ConvT16Params {
    const uint16_t* x;  // group layout [b][cin/8][x_ts][8], already activated by its writer
    int64_t x_bs;
    int x_ts;
    const uint16_t* wp;  // A fragments, phase-major rows (pack_conv_weights16, EPI_CONVT)
    const float* bias;
    const int* len_in;
    const int* len_out;
    int t_in, t_out;
    int cin, cout, rows;
    int s, crop;
    float* yg;  // fp32 output, group layout [b][cout/8][g_ts][8]
    int64_t g_bs;
    int g_ts;
    uint16_t* y16;  // optional 16-bit copy (leaky_relu(y16_slope) fused)
    int64_t y16_bs;
    int y16_ts;
    float y16_slope;
};

// NR: 32-column tiles per wave; CSPLIT: column groups of waves (1: the four waves split the row tiles; 2: two waves per row tile, half the columns each)
// RS: weight-fragment ring (8 or 16 slots; 2 c_in / 16 must be a multiple of it)
template <int NR, int CSPLIT, int RS, bool BF>
__global__ __launch_bounds__(256, 2) void convt16_kernel(const ConvT16Params p) {
    using namespace ct16;
    constexpr int BN = NR * CSPLIT * 32;       // input positions (GEMM columns) per block
    constexpr int XW = (BN + 1 + 7) / 8 * 8;   // slots per group row: column 0 = position t0 - 1 (tap 1 reads x[q - 1])
    extern __shared__ __attribute__((aligned(16))) int4v xs[];  // [cin/8][XW]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // (wave-uniform: the weight stream is addressed through scalar offsets)
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * BN;
    const int len_in = p.len_in ? p.len_in[b] : p.t_in;
    const int ncols = len_in + 1;  // q in [0, L_in]: the last position only sees the second tap
    if (t0 >= ncols || len_in <= 0) return;
    const int G = p.cin >> 3, nchunks = p.cin >> 5;
    const int h = lane >> 5, col = lane & 31;

    // ---- the input tile, every channel group, straight into LDS (all four waves issue the DMA; positions outside the sequence are zero) ----
    {
        const uint16_t* xb = p.x + (int64_t)b * p.x_bs;
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(xb), 0, 0x7fffffff, 0x00020000);
        const int ts = t0 - 1;
        constexpr int NP = (XW + 63) / 64;
        int voff[NP];
        bool oob[NP];
#pragma unroll
        for (int m = 0; m < NP; ++m) {
            const int t = ts + lane + 64 * m;
            const int tc = t < 0 ? 0 : (t < len_in ? t : len_in - 1);
            voff[m] = tc * 16;
            oob[m] = t != tc;
        }
        for (int g = wid; g < G; g += 4) {
            const unsigned soff = (unsigned)g * (unsigned)p.x_ts * 16u;
#pragma unroll
            for (int m = 0; m < NP; ++m) {
                const int vo = voff[m];  // (a local: hipcc silently drops the host stub of a kernel template that passes an element of a dependent-size array to this builtin)
                if (64 * m + lane < XW) __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(xs + g * XW + 64 * m), 16, vo, (int)soff, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (ts < 0 || ts + XW > len_in) {
            const int4v z = {0, 0, 0, 0};
            for (int g = wid; g < G; g += 4)
#pragma unroll
                for (int m = 0; m < NP; ++m)
                    if (64 * m + lane < XW && oob[m]) xs[g * XW + 64 * m + lane] = z;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();

    auto mfma = [&](int4v a, int4v bq, floatx16 c) __attribute__((always_inline)) -> floatx16 {
        if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bq), c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, bq), c, 0, 0, 0);
    };
    typedef const __attribute__((address_space(3))) int4v* LdsV;
    const int cgrp = wid % CSPLIT;                 // this wave's column group
    const int cbase = cgrp * (NR * 32) + col;      // tile-local position of column tile 0
    const int nrt = p.rows >> 5;
    const int total = nchunks * 4;                 // k-steps per row tile: chunk x {tap 0, tap 1} x {k-half 0, 1}
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.wp), 0, 0x7fffffff, 0x00020000);
    const int out_len = p.len_out ? p.len_out[b] : p.t_out;
    float* yg = p.yg ? p.yg + (int64_t)b * p.g_bs : nullptr;
    uint16_t* y16 = p.y16 ? p.y16 + (int64_t)b * p.y16_bs : nullptr;

    // The weight fragments of ALL this wave's row tiles form one continuous stream (row tile rt, then rt + 4 / CSPLIT, ...), fetched RD
    // steps ahead through a ring of RS register sets: a row tile is only 8-64 steps of 128 cycles, so a prefetch that restarted at
    // every row tile would expose an L2 round trip per tile (measured: the first version of this kernel, 8 slots restarted per tile,
    // took 0.79 ms for the two stride-8 upsamplers of the benchmark batch against 0.45 ms for the GEMM tile).
    constexpr int RD = RS - 2;
    constexpr int RSTEP = 4 / CSPLIT;  // row-tile stride of a wave
    const int rt_first = wid / CSPLIT;
    const int npass = rt_first < nrt ? (nrt - rt_first + RSTEP - 1) / RSTEP : 0;
    const int lanev = lane * 16;
    int lp = 0, ls = 0;  // (pass, step) of the next fragment to fetch
    auto fetch = [&]() __attribute__((always_inline)) -> int4v {
        const int pc = lp < npass ? lp : npass - 1;  // past the end: re-read the last fragment (value unused)
        const int sc = lp < npass ? ls : total - 1;
        const int soff = ((rt_first + pc * RSTEP) * total + sc) * 1024;
        const int4v v = __builtin_bit_cast(int4v, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lanev, soff, 0));
        if (++ls == total) {
            ls = 0;
            ++lp;
        }
        return v;
    };
    int4v ring[RS];
    if (npass > 0) {
#pragma unroll
        for (int i = 0; i < RD; ++i) ring[i] = fetch();
    }
    for (int pass = 0; pass < npass; ++pass) {
        const int rt = rt_first + pass * RSTEP;
        floatx16 acc[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        // this row tile's biases, requested in front of the K loop (inside the epilogue every group's load sat behind the previous group's stores:
        // a memory round trip per group, see convt16_lines_kernel)
        float4v bias4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int rho0 = rt * 32 + 8 * g + 4 * h;
            bias4[g] = float4v{0.f, 0.f, 0.f, 0.f};
            if (p.bias) bias4[g] = *reinterpret_cast<const float4v*>(p.bias + (rho0 - (rho0 / p.cout) * p.cout));
        }
        LdsV xb = (LdsV)(xs + h * XW + cbase + 1);  // + 1: tap j reads position q - j = LDS column (q - t0) + 1 - j
        for (int s0 = 0; s0 < total; s0 += RS) {     // (total is a multiple of RS: the ring phase is the same at every row tile)
#pragma unroll
            for (int u = 0; u < RS; ++u) {
                ring[(u + RD) % RS] = fetch();
                __builtin_amdgcn_sched_barrier(0);
                // step u of this group: chunk (u >> 2), tap (u >> 1) & 1, k-half u & 1 -> groups 2 kk + h of the chunk, column - tap
                const int off = (u >> 2) * 4 * XW + ((u & 1) ? 2 * XW : 0) - ((u >> 1) & 1);
                int4v bq[NR];
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) bq[nr] = xb[off + nr * 32];
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) acc[nr] = mfma(ring[u], bq[nr], acc[nr]);
            }
            xb += (RS / 4) * 4 * XW;  // RS / 4 chunks = RS channel groups
        }
        // ---- epilogue (conv16's E16_CONVT_GROUP): registers 4g..4g+3 = channels co0..co0+3 of ONE output sample n = s q + phase - crop ----
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int rho0 = rt * 32 + 8 * g + 4 * h;
            const int ph = rho0 / p.cout, co0 = rho0 - ph * p.cout;
            const float4v bias = bias4[g];
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                const int q = t0 + cbase + nr * 32;
                const int n = p.s * q + ph - p.crop;
                if (q >= ncols || n < 0 || n >= out_len) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[nr][4 * g + e] + bias[e];
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

// ---- stride a multiple of 4 (the stride-8 upsamplers): four phases per wave, so that whole 128-byte output lines are written at once -------
// In the fp32 group layout one 128-byte line is 4 consecutive samples x the 8 channels of a group, i.e. 4 consecutive PHASES of one input
// position. A wave therefore keeps the accumulators of four phases (row tiles rt(ph) = ph * c_out / 32 + cb, ph = 4 half ... 4 half + 3) of
// its 32-channel block cb for 64 positions — acc[4][2], 128 VGPRs — and stores the four phases of a line back to back: the line is complete
// in L2 within a few hundred cycles. Per k-step: four weight fragments (one per phase) and two LDS operand reads feed eight MFMAs.
// Work units of a block: (channel block, phase half, 64-position column pair), dealt to the four waves round robin.
template <int BN, bool BF>
__global__ __launch_bounds__(256, 2) void convt16_lines_kernel(const ConvT16Params p) {
    using namespace ct16;
    constexpr int XW = (BN + 1 + 7) / 8 * 8;
    constexpr int NR = 2, PH = 4;
    extern __shared__ __attribute__((aligned(16))) int4v xs[];  // [cin/8][XW]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * BN;
    const int len_in = p.len_in ? p.len_in[b] : p.t_in;
    const int ncols = len_in + 1;
    if (t0 >= ncols || len_in <= 0) return;
    const int G = p.cin >> 3, nchunks = p.cin >> 5;
    const int h = lane >> 5, col = lane & 31;
    {
        const uint16_t* xb = p.x + (int64_t)b * p.x_bs;
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(xb), 0, 0x7fffffff, 0x00020000);
        const int ts = t0 - 1;
        constexpr int NP = (XW + 63) / 64;
        int voff[NP];
        bool oob[NP];
#pragma unroll
        for (int m = 0; m < NP; ++m) {
            const int t = ts + lane + 64 * m;
            const int tc = t < 0 ? 0 : (t < len_in ? t : len_in - 1);
            voff[m] = tc * 16;
            oob[m] = t != tc;
        }
        for (int g = wid; g < G; g += 4) {
            const unsigned soff = (unsigned)g * (unsigned)p.x_ts * 16u;
#pragma unroll
            for (int m = 0; m < NP; ++m) {
                const int vo = voff[m];
                if (64 * m + lane < XW) __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(xs + g * XW + 64 * m), 16, vo, (int)soff, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (ts < 0 || ts + XW > len_in) {
            const int4v z = {0, 0, 0, 0};
            for (int g = wid; g < G; g += 4)
#pragma unroll
                for (int m = 0; m < NP; ++m)
                    if (64 * m + lane < XW && oob[m]) xs[g * XW + 64 * m + lane] = z;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();

    auto mfma = [&](int4v a, int4v bq, floatx16 c) __attribute__((always_inline)) -> floatx16 {
        if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bq), c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, bq), c, 0, 0, 0);
    };
    typedef const __attribute__((address_space(3))) int4v* LdsV;
    const int total = nchunks * 4;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.wp), 0, 0x7fffffff, 0x00020000);
    const int out_len = p.len_out ? p.len_out[b] : p.t_out;
    float* yg = p.yg ? p.yg + (int64_t)b * p.g_bs : nullptr;
    uint16_t* y16 = p.y16 ? p.y16 + (int64_t)b * p.y16_bs : nullptr;
    const int ncb = p.cout >> 5;                 // 32-channel blocks
    const int nhalf = p.s / PH;                  // groups of four phases
    constexpr int NCP = BN / 64;                 // 64-position column pairs
    const int nunits = ncb * nhalf * NCP;
    const int lanev = lane * 16;

    // (gridDim.z > 1, small grids: the units of a tile of positions are dealt out over that many blocks — at batch 1 the first upsampler is
    // four tiles of sixteen units each on 256 CUs; every block stages the small input tile itself)
    for (int unit = wid + 4 * (int)blockIdx.z; unit < nunits; unit += 4 * (int)gridDim.z) {
        const int cp = unit % NCP, rest = unit / NCP;
        const int half = rest % nhalf, cb = rest / nhalf;
        floatx16 acc[PH][NR];
#pragma unroll
        for (int k = 0; k < PH; ++k)
#pragma unroll
            for (int j = 0; j < NR; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[k][j][r] = 0.f;
        int sbase[PH];  // scalar byte offset of the first fragment of each phase's row tile
#pragma unroll
        for (int k = 0; k < PH; ++k) sbase[k] = (((half * PH + k) * ncb + cb) * total) * 1024;
        auto load_a = [&](int k, int step) __attribute__((always_inline)) -> int4v {
            return __builtin_bit_cast(int4v, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lanev, sbase[k] + step * 1024, 0));
        };
        // ring of four register sets per phase, two steps (16 MFMAs, 512 cycles) ahead
        int4v ring[4][PH];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int k = 0; k < PH; ++k) ring[i][k] = load_a(k, i < total ? i : total - 1);
        // the unit's biases, requested in front of the K loop: fetched inside the epilogue, every group's load sat behind the previous group's
        // stores, and a load can only be waited for together with the stores issued before it (one in-order vmcnt): a memory round trip per group
        float4v bias4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bias4[g] = float4v{0.f, 0.f, 0.f, 0.f};
            if (p.bias) bias4[g] = *reinterpret_cast<const float4v*>(p.bias + cb * 32 + 8 * g + 4 * h);
        }
        LdsV xb = (LdsV)(xs + h * XW + cp * 64 + col + 1);
        for (int s0 = 0; s0 < total; s0 += 4) {  // one chunk per iteration: {tap 0, tap 1} x {k-half 0, 1}
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = s0 + u;
#pragma unroll
                for (int k = 0; k < PH; ++k) ring[(u + 2) & 3][k] = load_a(k, s + 2 < total ? s + 2 : total - 1);
                __builtin_amdgcn_sched_barrier(0);
                const int off = ((u & 1) ? 2 * XW : 0) - ((u >> 1) & 1);
                int4v bq[NR];
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) bq[nr] = xb[off + nr * 32];
#pragma unroll
                for (int k = 0; k < PH; ++k)
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr) acc[k][nr] = mfma(ring[u][k], bq[nr], acc[k][nr]);
            }
            xb += 4 * XW;
        }
        // ---- epilogue (same expressions as conv16's E16_CONVT_GROUP). In the MFMA C layout a lane holds ONE position q (its column) and, in
        // acc[k], the four phases of it: a store instruction for phase k then touches 32 different 128-byte lines (one per position, 256
        // bytes apart) with 32 bytes each, and every line is completed by four instructions: quarter-line writes, 2.6 TB/s on the 256 -> 128
        // upsampler where coalesced stores reach 6. A 4 x 4 transpose inside each quad of lanes (two DPP quad_perm butterflies per value, as in
        // conv_mfma.hip's wide epilogue) hands lane j of a quad phase j of the quad's four positions: a store instruction then writes the
        // four phases (x two channel halves from lanes l, l + 32) of EIGHT positions = eight complete lines.
#ifndef VITS_CT16L_OLD_EPI
        {
            const int qj = lane & 3;  // phase this lane stores after the transpose
            const bool odd1 = lane & 1, odd2 = lane & 2;
            auto quad_transpose = [&](float& v0, float& v1, float& v2, float& v3) __attribute__((always_inline)) {
                {
                    float s01 = odd1 ? v0 : v1, s23 = odd1 ? v2 : v3;
                    s01 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s01), 0xB1, 0xF, 0xF, true));
                    s23 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s23), 0xB1, 0xF, 0xF, true));
                    if (odd1) { v0 = s01; v2 = s23; } else { v1 = s01; v3 = s23; }
                }
                {
                    float s02 = odd2 ? v0 : v2, s13 = odd2 ? v1 : v3;
                    s02 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s02), 0x4E, 0xF, 0xF, true));
                    s13 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s13), 0x4E, 0xF, 0xF, true));
                    if (odd2) { v0 = s02; v1 = s13; } else { v2 = s02; v3 = s13; }
                }
            };
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co0 = cb * 32 + 8 * g + 4 * h;
                const float4v bias = bias4[g];
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) {
                    // t[i][e]: position (quad base + i), phase qj, channel e of this lane's half slot
                    float t[4][4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v0 = acc[0][nr][4 * g + e], v1 = acc[1][nr][4 * g + e], v2 = acc[2][nr][4 * g + e], v3 = acc[3][nr][4 * g + e];
                        quad_transpose(v0, v1, v2, v3);  // (every lane of the quad takes part: the guards below only cover the stores)
                        t[0][e] = v0;
                        t[1][e] = v1;
                        t[2][e] = v2;
                        t[3][e] = v3;
                    }
                    const int qb = t0 + cp * 64 + nr * 32 + (col & ~3);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int q = qb + i;
                        if (q >= ncols) continue;
                        const int n = p.s * q + half * PH + qj - p.crop;
                        if (n < 0 || n >= out_len) continue;
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = t[i][e] + bias[e];
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
#else
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co0 = cb * 32 + 8 * g + 4 * h;
            const float4v bias = bias4[g];
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                const int q = t0 + cp * 64 + nr * 32 + col;
                if (q >= ncols) continue;
#pragma unroll
                for (int k = 0; k < PH; ++k) {
                    const int n = p.s * q + half * PH + k - p.crop;
                    if (n < 0 || n >= out_len) continue;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[k][nr][4 * g + e] + bias[e];
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
#endif
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------------
bool convt16_stream_supported(const PackedConv& w) {
    const bool off = kernel_knobs().no_convt16s;
    // (VITS_CONVT16S_ALL=1: the one-row-tile-at-a-time kernel also for stride 8 — the slow first version, kept for the comparison)
    const bool all = kernel_knobs().convt16s_all;
    const bool no_lines = kernel_knobs().no_convt16l;  // (the four-phase variant for strides that are multiples of 4)
    if (off || w.epi != EPI_CONVT || w.kt != 2 || !w.wp16 || w.cin % 64 != 0 || w.cout % 32 != 0 || w.rows % 32 != 0 || w.cin > 512) return false;
    if (w.rows <= 128 || all) return true;
    return !no_lines && w.ct_stride % 4 == 0;
}

template <int NR, int CSPLIT, int RS, bool BF>
static hipError_t launch_ct(const ConvT16Params& p, int ncols_max, int batch, hipStream_t s) {
    constexpr int BN = NR * CSPLIT * 32, XW = (BN + 1 + 7) / 8 * 8;
    const size_t lds = (size_t)(p.cin / 8) * XW * 16;
    static BigLdsOnce big_lds_set;
    if (lds > 64 * 1024 && big_lds_set.needed()) {
        hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&convt16_kernel<NR, CSPLIT, RS, BF>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (ea != hipSuccess) return ea;
        big_lds_set.done();
    }
    dim3 grid((ncols_max + BN - 1) / BN, batch);
    VITS_KLAUNCH((convt16_kernel<NR, CSPLIT, RS, BF>), grid, dim3(256), lds, s, p);
    return hipGetLastError();
}

// which instantiation serves a transposed conv (one place: the launch below and the profiler label of the engine both ask here).
// convt16_lines_kernel<BN>: four phases per wave, for row counts above 128 with a stride that is a multiple of 4; 128 positions per block,
// 64 when c_in = 512 (LDS for two blocks per CU). convt16_kernel<NR, CSPLIT, RS>: 128 positions per block; 64 when the input tile of 128
// would not leave room for two blocks per CU (c_in = 512); 256 positions with two waves per row tile when there are only two row tiles (the
// last upsampler: 64 rows); sixteen ring slots where the step count allows.
namespace {
struct CtChoice {
    int lines_bn = 0, nr = 0, csplit = 0, rs = 0;
};
CtChoice ct_choice(const PackedConv& w) {
    const bool all_s = kernel_knobs().convt16s_all;
    CtChoice c;
    if (w.rows > 128 && w.ct_stride % 4 == 0 && !all_s) {
        c.lines_bn = w.cin > 256 ? 64 : 128;
        return c;
    }
    c.rs = w.cin % 128 == 0 ? 16 : 8;
    if (w.rows / 32 <= 2) c.nr = 4, c.csplit = 2;
    else if (w.cin > 256) c.nr = 2, c.csplit = 1;
    else c.nr = 4, c.csplit = 1;
    if (const int ov = kernel_knobs().convt16_r128; ov && w.rows == 128) {  // (developer override: same bits, another block shape)
        c.nr = ov / 100, c.csplit = ov / 10 % 10, c.rs = (ov % 10) ? 16 : 8;
        if (c.rs == 16 && w.cin % 128 != 0) c.rs = 8;
    }
    return c;
}
}  // namespace

hipError_t launch_convt16_stream(const PackedConv& w, const Conv16Call& c, int arith, hipStream_t s) {
    if (!convt16_stream_supported(w) || !c.yg) return hipErrorInvalidValue;
    ConvT16Params p;
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
    p.s = w.ct_stride;
    p.crop = c.ct_crop;
    p.yg = c.yg;
    p.g_bs = c.g_bs;
    p.g_ts = c.g_ts;
    p.y16 = c.y16.p;
    p.y16_bs = c.y16.bs;
    p.y16_ts = c.y16.ts;
    p.y16_slope = c.y16_slope;
    const bool bf = arith == VITS_ARITH_BF16;
    const int ncols_max = c.t_in + 1;
    const CtChoice ch = ct_choice(w);
    if (ch.lines_bn) {
        // four phases per wave: whole output lines per store burst. 128 positions per block, 64 when c_in = 512 (LDS for two blocks per CU)
        auto go = [&](auto bn_c, auto bf_c) -> hipError_t {
            constexpr int BN = decltype(bn_c)::value;
            constexpr bool BFv = decltype(bf_c)::value;
            constexpr int XW = (BN + 1 + 7) / 8 * 8;
            const size_t lds = (size_t)(p.cin / 8) * XW * 16;
            static BigLdsOnce big_lds_set;
            if (lds > 64 * 1024 && big_lds_set.needed()) {
                hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&convt16_lines_kernel<BN, BFv>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (ea != hipSuccess) return ea;
                big_lds_set.done();
            }
            const int64_t tiles = (int64_t)((ncols_max + BN - 1) / BN) * c.batch;
            const int nunits = (p.cout >> 5) * (p.s / 4) * (BN / 64);
            int zsplit = 1;
            if (tiles <= kernel_knobs().convt16_split_max) zsplit = nunits >= 16 ? 4 : (nunits >= 8 ? 2 : 1);
            dim3 grid((ncols_max + BN - 1) / BN, c.batch, zsplit);
            VITS_KLAUNCH((convt16_lines_kernel<BN, BFv>), grid, dim3(256), lds, s, p);
            return hipGetLastError();
        };
        if (ch.lines_bn == 64) return bf ? go(std::integral_constant<int, 64>{}, std::true_type{}) : go(std::integral_constant<int, 64>{}, std::false_type{});
        return bf ? go(std::integral_constant<int, 128>{}, std::true_type{}) : go(std::integral_constant<int, 128>{}, std::false_type{});
    }
#define VITS_CT(NR, CS, RS)                                                                                            \
    if (ch.nr == NR && ch.csplit == CS && ch.rs == RS)                                                                 \
        return bf ? launch_ct<NR, CS, RS, true>(p, ncols_max, c.batch, s) : launch_ct<NR, CS, RS, false>(p, ncols_max, c.batch, s)
    VITS_CT(4, 2, 16);
    VITS_CT(4, 2, 8);
    VITS_CT(2, 1, 16);
    VITS_CT(2, 1, 8);
    VITS_CT(4, 1, 16);
    VITS_CT(4, 1, 8);
#undef VITS_CT
    return hipErrorInvalidValue;
}

void convt16_stream_tag(const PackedConv& w, char* buf, size_t cap) {
    const CtChoice ch = ct_choice(w);
    if (ch.lines_bn) std::snprintf(buf, cap, "SL%d", ch.lines_bn);
    else std::snprintf(buf, cap, "S%d.%d.%d", ch.nr, ch.csplit, ch.rs);
}

}  // namespace vits
