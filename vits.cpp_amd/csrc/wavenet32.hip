// wavenet32.hip — one WaveNet layer of the residual-coupling flow as a single kernel, exact-fp32 arithmetic:
//     acts = tanh(in[0:H]) * sigmoid(in[H:2H]),  in = Conv_{k,1}(h) + b_in                 (/root/reference/src/vits.cpp:470-483)
//     rs   = Conv_{1x1}(acts) + b_rs;  h' = h + rs[0:H];  outputs += rs[H:2H]                (vits.cpp:484-491; last layer: outputs += rs)
// As two launches (gated conv, then the 1x1 res/skip conv) a layer is 0.18 ms at batch 64 x 225 frames for 12.7 GFLOP (0.08 ms at the
// fp32 MFMA peak): few hundred columns per utterance, 64- and 128-column tiles a quarter empty, two prologues / epilogues. Here a block
// owns 64 frames: the h tile goes into LDS once, twelve waves — one per (32-channel group, 32-frame column tile) — run the gated conv
// (the tanh and the sigmoid row tile of the group), the gate is applied in registers, acts take the h tile's place in LDS, the same waves run the 1x1 conv
// from there and the epilogue adds into h / outputs. Same MFMA chain per output as conv_mfma.hip (chunk, tap, channel pair) and the
// same epilogue expressions: bit-identical to the two-launch path (GPU test), which stays for other shapes, for the 16-bit modes
// and behind VITS_NO_WN_FUSE=1.
// A block reads a 2-frame halo of its neighbours' h columns while other blocks already store h': h' goes to ANOTHER buffer
// (Engine::run_batch alternates two); `outputs` has no halo and is updated in place.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "../../include/vits.h"
#include "kernels.h"

namespace vits {

typedef float wn_floatx16 __attribute__((ext_vector_type(16)));
typedef float wn_float4v __attribute__((ext_vector_type(4)));

struct WaveNet32Params {
    const float* h;  // [b][H][t]
    int64_t h_bs;
    int h_cs;
    float* h_out;  // h' (null on the last layer)
    int64_t ho_bs;
    int ho_cs;
    float* outputs;  // skip accumulator [b][H][t]
    int64_t o_bs;
    int o_cs;
    const float *w_in, *b_in;  // gated conv: packed EPI_GATE fragments (tile 2i = tanh rows of channels 32i.., 2i+1 = sigmoid rows)
    const float *w_rs, *b_rs;  // 1x1 res/skip conv: packed EPI_STD fragments, rs_rows = 2H (H on the last layer)
    int rs_rows;
    const int* lens;
    int tmax;
};

template <int H, int KT>
__global__ __launch_bounds__(4 * H, 1) void wavenet32_kernel(const WaveNet32Params p) {
    constexpr int NG = H / 32;   // channel groups of 32 = gate row-tile pairs
    constexpr int NC = 2;        // 32-frame column tiles per block: one wave per (channel group, column tile)
    constexpr int NW = NG * NC;  // 12 waves at H = 192: three per SIMD (six waves on four SIMDs ran (2, 2, 1, 1): 160 us per layer against 110)
    constexpr int NCH = H / 32;  // 32-channel chunks of K
    constexpr int NR = 1;        // 32-frame column tiles per wave
    constexpr int BM = NC * 32;  // frames per block
    constexpr int P = (KT - 1) / 2;
    constexpr int XWP = (BM + KT - 1 + 3 + 3) / 4 * 4;  // h tile row pitch (floats): + up to 3 columns of alignment shift
    constexpr int XW4 = XWP / 4;
    constexpr int TWP = BM;              // acts tile row pitch
    constexpr int TOTAL1 = NCH * KT * 4;  // A-fragment steps of the gated conv per row tile
    constexpr int TOTAL2 = NCH * 4;       // ... of the 1x1 conv
    static_assert(TWP <= XWP, "acts take the h tile's place");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;
    float* ts = lds;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int b = blockIdx.y;
    const int len = p.lens ? p.lens[b] : p.tmax;
    const int t0 = blockIdx.x * BM;
    if (t0 >= len) return;
    const int krow = lane >> 5;
    const int gw = wid % NG, col = (wid / NG) * 32 + (lane & 31);  // this wave's channel group and (block-local) frame

    // ---- the h tile, all channels, straight into LDS; LDS column 0 = global frame ts0 (16-byte aligned source) ----
    const int tx0 = t0 - P;
    const int ts0 = tx0 & ~3;
    const int shift = tx0 - ts0;
    const float* hb = p.h + (int64_t)b * p.h_bs;
    {
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hb), 0, 0x7fffffff, 0x00020000);
        const int tlast = (len - 1) & ~3;
        constexpr int N4 = H * XW4;
        constexpr int NI = (N4 + 63) / 64;
#pragma unroll
        for (int n0 = 0; n0 < NI; n0 += NW) {
            const int n = n0 + wid;
            if (n < NI) {
                int g = n * 64 + lane;
                g = g < N4 ? g : N4 - 1;
                const int r = g / XW4, c4 = g - r * XW4;
                int t = ts0 + 4 * c4;
                t = t < 0 ? 0 : (t > tlast ? tlast : t);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(xs + n * 256), 16, (r * p.h_cs + t) * 4, 0, 0, 0);
            }
        }
    }
    // biases of this lane's rows (accumulator register r <-> channel 32*wid + 8*(r/4) + 4*krow + r%4)
    float bt[16], bs[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ch = gw * 32 + (r >> 2) * 8 + krow * 4 + (r & 3);
        bt[r] = p.b_in[ch];
        bs[r] = p.b_in[H + ch];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {  // zero padding outside the sequence
        wn_float4v* x4 = reinterpret_cast<wn_float4v*>(xs);
        constexpr int N4 = H * XW4;
        if (ts0 < 0 || ts0 + XWP > len) {
            for (int g = tid; g < N4; g += 64 * NW) {
                const int r = g / XW4, c4 = g - r * XW4;
                const int t = ts0 + 4 * c4;
                wn_float4v v = x4[g];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (t + e < 0 || t + e >= len) v[e] = 0.f;
                x4[g] = v;
            }
        }
    }
    __syncthreads();

    typedef const __attribute__((address_space(3))) float* LdsF;
    wn_floatx16 acc[2][NR];

    // one conv over the LDS tile for this wave's TWO row tiles mt0, mt0 + 1 (order per output: chunk, tap, channel pair)
    auto conv = [&](const float* wp, int mt0, auto total_c, auto taps_c, LdsF base, const int pitch) __attribute__((always_inline)) {
        constexpr int TOTAL = decltype(total_c)::value, TAPS = decltype(taps_c)::value;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][nr][r] = 0.f;
        const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp), 0, 0x7fffffff, 0x00020000);
        int wvoff[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) wvoff[m] = (int)(((size_t)(mt0 + m) * TOTAL * 64 + lane) * 16);
        auto load_a = [&](int m, int step) __attribute__((always_inline)) -> wn_float4v {
            return __builtin_bit_cast(wn_float4v, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff[m], step * 1024, 0));
        };
        wn_float4v ring[4][2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            ring[0][m] = load_a(m, 0);
            ring[1][m] = load_a(m, 1 < TOTAL ? 1 : 0);
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int j = 0; j < TAPS; ++j) {
                LdsF xj = base + (c * 32) * pitch + j;
#pragma unroll
                for (int p4 = 0; p4 < 4; ++p4) {
                    const int s = (c * TAPS + j) * 4 + p4;  // compile time after unrolling
#pragma unroll
                    for (int m = 0; m < 2; ++m) ring[(s + 2) & 3][m] = load_a(m, s + 2 < TOTAL ? s + 2 : TOTAL - 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float bv[NR];
#pragma unroll
                        for (int nr = 0; nr < NR; ++nr) bv[nr] = xj[(2 * (p4 * 4 + q)) * pitch + nr * 32];
#pragma unroll
                        for (int m = 0; m < 2; ++m)
#pragma unroll
                            for (int nr = 0; nr < NR; ++nr) acc[m][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[s & 3][m][q], bv[nr], acc[m][nr], 0, 0, 0);
                    }
                }
            }
        }
    };

    // ---- gated conv: packed tiles 2*wid (tanh rows) and 2*wid + 1 (sigmoid rows) ----
    conv(p.w_in, 2 * gw, std::integral_constant<int, TOTAL1>{}, std::integral_constant<int, KT>{}, (LdsF)(xs + krow * XWP + shift + col), XWP);
    __syncthreads();  // every wave is done with the h tile: acts take its place
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
        const bool inside = t0 + nr * 32 + col < len;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = gw * 32 + (r >> 2) * 8 + krow * 4 + (r & 3);
            const float v = wavenet_gate(acc[0][nr][r] + bt[r], acc[1][nr][r] + bs[r]);
            ts[ch * TWP + nr * 32 + col] = inside ? v : 0.f;
        }
    }
    __syncthreads();

    // ---- 1x1 res/skip conv: row tiles 2*wid, 2*wid + 1 of rs_rows / 32 ----
    const int ntiles2 = p.rs_rows >> 5;
    if (2 * gw >= ntiles2) return;  // (last layer: H rows = NW tiles: the upper half of the waves has none)
    conv(p.w_rs, 2 * gw, std::integral_constant<int, TOTAL2>{}, std::integral_constant<int, 1>{}, (LdsF)(ts + krow * TWP + col), TWP);

    // ---- epilogue: rows < H of a 2H-row layer -> h' = h + rs; the other rows -> outputs += rs ----
    // Two passes: every addend (the residual h or the running `outputs`) and bias is loaded FIRST, then everything is stored. Written as
    // load / add / store per element the compiler has to keep each read-modify-write of `outputs` behind the previous store (the pointers
    // may alias for all it knows): 32 dependent HBM round trips per lane, 15 of the 48 us of a layer in the 16-bit kernel (tools/wn16_micro.hip).
    const bool two = p.rs_rows > H;
    float addv[NR][2][16], biasv[2][16];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (2 * gw + m) * 32 + (r >> 2) * 8 + krow * 4 + (r & 3);
            biasv[m][r] = 2 * gw + m < ntiles2 ? p.b_rs[row] : 0.f;
        }
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
        const int t = t0 + nr * 32 + col;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (2 * gw + m) * 32 + (r >> 2) * 8 + krow * 4 + (r & 3);
                float a = 0.f;
                if (t < len && 2 * gw + m < ntiles2)
                    a = (two && row < H) ? hb[(int64_t)row * p.h_cs + t] : p.outputs[(int64_t)b * p.o_bs + (int64_t)(two ? row - H : row) * p.o_cs + t];
                addv[nr][m][r] = a;
            }
    }
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
        const int t = t0 + nr * 32 + col;
        if (t >= len) continue;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            if (2 * gw + m >= ntiles2) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (2 * gw + m) * 32 + (r >> 2) * 8 + krow * 4 + (r & 3);
                float v = acc[m][nr][r] + biasv[m][r];
                v = addv[nr][m][r] + v;
                if (two && row < H) p.h_out[(int64_t)b * p.ho_bs + (int64_t)row * p.ho_cs + t] = v;
                else p.outputs[(int64_t)b * p.o_bs + (int64_t)(two ? row - H : row) * p.o_cs + t] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same layer in the 16-bit-operand modes (VITS_ARITH_F16 / BF16): operands rounded where the two-launch path rounds them (h and
// acts at the converter in front of each conv16 launch: here when the tile is written to LDS), v_mfma_f32_32x32x16_{f16,bf16} on the
// packed 16-bit fragments, fp32 accumulate / bias / gate / residual. The h tile is loaded through registers (it has to be converted
// anyway) into the group layout [c/8][frame][8] — one ds_read_b128 per MFMA operand. Saves two converter launches per layer as well.
// Bit-identical to to_group16 + conv16 (gate) + to_group16 + conv16 (1x1) (GPU test).
// ---------------------------------------------------------------------------------------------------------------------------------
typedef int wn_int4v __attribute__((ext_vector_type(4)));
typedef _Float16 wn_half8 __attribute__((ext_vector_type(8)));
typedef __bf16 wn_bf16x8 __attribute__((ext_vector_type(8)));

struct WaveNet16Params {
    WaveNet32Params f;  // h / h_out / outputs / biases / lens as in the fp32 kernel (w_in / w_rs unused)
    const uint16_t *w_in16, *w_rs16;
};

template <bool BF>
__device__ __forceinline__ uint16_t wn_round16(float v) {
    if constexpr (BF) {
        const __bf16 h = (__bf16)v;
        return __builtin_bit_cast(uint16_t, h);
    } else {
        const _Float16 h = (_Float16)v;
        return __builtin_bit_cast(uint16_t, h);
    }
}

#ifdef VITS_PHASE_TIMING  // developer instrumentation (tools/wn16_micro.hip)
__device__ unsigned long long vits_wn_phase[8 * 65536];
#define WN_STAMP(k)                                                                                      \
    do {                                                                                                 \
        if (threadIdx.x == 0) {                                                                          \
            const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;                                    \
            if (lin < 65536) vits_wn_phase[8 * lin + (k)] = __builtin_amdgcn_s_memrealtime();            \
        }                                                                                                \
    } while (0)
#else
#define WN_STAMP(k)
#endif

// NCW: 32-frame column tiles per wave. 1 = twelve waves, one (channel group, column tile) each; 2 = six waves that own BOTH column tiles of
// their channel group. The gated conv of a block is bound by the weight-fragment traffic through the CU's vector memory path: twelve waves x
// 120 KB = 1.44 MB at 64 B per clock = 11.8 us (measured 15 us for 6 us of MFMA work; tools/wn16_micro.hip); with NCW = 2 every fragment
// feeds two MFMAs per row tile and the traffic halves.
template <int H, int KT, bool BF, int NCW>
__global__ __launch_bounds__(4 * H / NCW, 1) void wavenet16_kernel(const WaveNet16Params pp) {
    const WaveNet32Params& p = pp.f;
    constexpr int NG = H / 32, NC = 2, NW = NG * NC / NCW;
    constexpr int NCH = H / 32;
    constexpr int BM = NC * 32;
    constexpr int P = (KT - 1) / 2;
    constexpr int XS = BM + KT - 1;  // slots per group row of the h tile
    constexpr int TOTAL1 = NCH * KT * 2, TOTAL2 = NCH * 2;  // A-fragment steps per row tile
    extern __shared__ __attribute__((aligned(16))) wn_int4v l16[];
    wn_int4v* xs = l16;  // [G][XS]
    wn_int4v* ts = l16;  // [G][BM]  (acts take the h tile's place)
    static_assert(BM <= XS, "acts take the h tile's place");

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int b = blockIdx.y;
    const int len = p.lens ? p.lens[b] : p.tmax;
    const int t0 = blockIdx.x * BM;
    if (t0 >= len) return;
    WN_STAMP(0);
    const int krow = lane >> 5;
    const int gw = wid % NG, col0 = (wid / NG) * (NCW * 32) + (lane & 31);  // this lane's column of its column tile n: col0 + 32 n
    const float* hb = p.h + (int64_t)b * p.h_bs;

    // ---- the h tile through registers: rounded, group layout; zero outside the sequence. A thread owns whole 16-byte slots (8 channels of
    // one frame: eight coalesced dword loads, one ds_write_b128) — as 2-byte LDS writes this phase was 13k conflicting ds_write_b16 per block ----
    {
        constexpr int NTH = 64 * NW, NSLOT = (H / 8) * XS, PER = (NSLOT + NTH - 1) / NTH;
        float v[PER][8];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int idx = tid + u * NTH;
            const int g = idx / XS, i = idx - g * XS;
            const int t = t0 - P + i;
            const bool ok = idx < NSLOT && t >= 0 && t < len;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[u][e] = ok ? hb[(int64_t)(g * 8 + e) * p.h_cs + t] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int idx = tid + u * NTH;
            if (idx < NSLOT) {
                wn_int4v q;
                q.x = (int)((unsigned)wn_round16<BF>(v[u][0]) | ((unsigned)wn_round16<BF>(v[u][1]) << 16));
                q.y = (int)((unsigned)wn_round16<BF>(v[u][2]) | ((unsigned)wn_round16<BF>(v[u][3]) << 16));
                q.z = (int)((unsigned)wn_round16<BF>(v[u][4]) | ((unsigned)wn_round16<BF>(v[u][5]) << 16));
                q.w = (int)((unsigned)wn_round16<BF>(v[u][6]) | ((unsigned)wn_round16<BF>(v[u][7]) << 16));
                xs[idx] = q;  // (slot index g * XS + i == idx)
            }
        }
    }
    float bt[16], bs[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ch = gw * 32 + (r >> 2) * 8 + krow * 4 + (r & 3);
        bt[r] = p.b_in[ch];
        bs[r] = p.b_in[H + ch];
    }
    __syncthreads();
    WN_STAMP(1);

    typedef const __attribute__((address_space(3))) wn_int4v* LdsV;
    wn_floatx16 acc[2][NCW];
    auto mfma = [&](wn_int4v a, wn_int4v bq, wn_floatx16 c) __attribute__((always_inline)) -> wn_floatx16 {
        if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wn_bf16x8, a), __builtin_bit_cast(wn_bf16x8, bq), c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wn_half8, a), __builtin_bit_cast(wn_half8, bq), c, 0, 0, 0);
    };
    // one conv for this wave's two row tiles mt0, mt0 + 1 and its NCW column tiles: order per output = chunk, tap, k-half (conv16.hip's)
    auto conv = [&](const uint16_t* wp, int mt0, auto total_c, auto taps_c, LdsV base, const int pitch) __attribute__((always_inline)) {
        constexpr int TOTAL = decltype(total_c)::value, TAPS = decltype(taps_c)::value;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NCW; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
        const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(wp), 0, 0x7fffffff, 0x00020000);
        int wvoff[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) wvoff[m] = (int)(((size_t)(mt0 + m) * TOTAL * 64 + lane) * 16);
        auto load_a = [&](int m, int step) __attribute__((always_inline)) -> wn_int4v {
            return __builtin_bit_cast(wn_int4v, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff[m], step * 1024, 0));
        };
        constexpr int RS = 8, RD = 6;
        wn_int4v ring[RS][2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int i = 0; i < RD; ++i) ring[i][m] = load_a(m, i < TOTAL ? i : TOTAL - 1);
        auto bslot = [&](int s) __attribute__((always_inline)) -> int {
            const int kk = s & 1, cj = s >> 1, j = cj % TAPS, c = cj / TAPS;
            return (c * 4 + 2 * kk) * pitch + j;
        };
        wn_int4v b_nxt[NCW];
#pragma unroll
        for (int n = 0; n < NCW; ++n) b_nxt[n] = base[bslot(0) + 32 * n];
#pragma unroll
        for (int s = 0; s < TOTAL; ++s) {
#pragma unroll
            for (int m = 0; m < 2; ++m) ring[(s + RD) % RS][m] = load_a(m, s + RD < TOTAL ? s + RD : TOTAL - 1);
            __builtin_amdgcn_sched_barrier(0);
            wn_int4v bq[NCW];
#pragma unroll
            for (int n = 0; n < NCW; ++n) {
                bq[n] = b_nxt[n];
                b_nxt[n] = base[bslot(s + 1 < TOTAL ? s + 1 : TOTAL - 1) + 32 * n];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < NCW; ++n) acc[m][n] = mfma(ring[s % RS][m], bq[n], acc[m][n]);
        }
    };

    // ---- gated conv ----
    conv(pp.w_in16, 2 * gw, std::integral_constant<int, TOTAL1>{}, std::integral_constant<int, KT>{}, (LdsV)(xs + krow * XS + col0), XS);
    WN_STAMP(2);
    __syncthreads();
    WN_STAMP(6);
    // acts, rounded, as whole slots: lane l holds channels 4 krow .. 4 krow + 3 of each of the 4 channel groups of its row tile at its frame;
    // v_permlane32_swap trades halves between two groups so that every lane writes one 16-byte slot (2-byte writes: an 8-way bank conflict)
#pragma unroll
    for (int n = 0; n < NCW; ++n) {
        const int col = col0 + 32 * n;
        const bool inside = t0 + col < len;
        unsigned w[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            unsigned short q[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g + e;
                float v = wavenet_gate(acc[0][n][r] + bt[r], acc[1][n][r] + bs[r]);
                asm volatile("" : "+v"(v));  // (two roundings, as the two-launch path: fp32 acts to memory, 16-bit at the converter)
                q[e] = inside ? wn_round16<BF>(v) : (unsigned short)0;
            }
            w[g][0] = (unsigned)q[0] | ((unsigned)q[1] << 16);
            w[g][1] = (unsigned)q[2] | ((unsigned)q[3] << 16);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const auto x = __builtin_amdgcn_permlane32_swap(w[2 * k][0], w[2 * k + 1][0], false, false);
            const auto y = __builtin_amdgcn_permlane32_swap(w[2 * k][1], w[2 * k + 1][1], false, false);
            ts[(gw * 4 + 2 * k + krow) * BM + col] = wn_int4v{(int)x[0], (int)y[0], (int)x[1], (int)y[1]};
        }
    }
    WN_STAMP(7);
    __syncthreads();
    WN_STAMP(3);

    // ---- 1x1 res/skip conv ----
    const int ntiles2 = p.rs_rows >> 5;
    if (2 * gw >= ntiles2) return;
    conv(pp.w_rs16, 2 * gw, std::integral_constant<int, TOTAL2>{}, std::integral_constant<int, 1>{}, (LdsV)(ts + krow * BM + col0), BM);
    WN_STAMP(4);

    // (two passes — every addend and bias loaded first, then everything stored: see the fp32 kernel's epilogue)
    const bool two = p.rs_rows > H;
    float addv[NCW][2][16], biasv[2][16];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (2 * gw + m) * 32 + (r >> 2) * 8 + krow * 4 + (r & 3);
            biasv[m][r] = 2 * gw + m < ntiles2 ? p.b_rs[row] : 0.f;
        }
#pragma unroll
    for (int n = 0; n < NCW; ++n) {
        const int t = t0 + col0 + 32 * n;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (2 * gw + m) * 32 + (r >> 2) * 8 + krow * 4 + (r & 3);
                float a = 0.f;
                if (t < len && 2 * gw + m < ntiles2)
                    a = (two && row < H) ? hb[(int64_t)row * p.h_cs + t] : p.outputs[(int64_t)b * p.o_bs + (int64_t)(two ? row - H : row) * p.o_cs + t];
                addv[n][m][r] = a;
            }
    }
#pragma unroll
    for (int n = 0; n < NCW; ++n) {
        const int t = t0 + col0 + 32 * n;
        if (t >= len) continue;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            if (2 * gw + m >= ntiles2) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (2 * gw + m) * 32 + (r >> 2) * 8 + krow * 4 + (r & 3);
                float v = acc[m][n][r] + biasv[m][r];
                v = addv[n][m][r] + v;
                if (two && row < H) p.h_out[(int64_t)b * p.ho_bs + (int64_t)row * p.ho_cs + t] = v;
                else p.outputs[(int64_t)b * p.o_bs + (int64_t)(two ? row - H : row) * p.o_cs + t] = v;
            }
        }
    }
    WN_STAMP(5);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// One whole coupling layer of the flow as ONE kernel, 16-bit-operand modes (vits.cpp:500-517, reverse direction, mean only):
//     h = pre(x0);  4 x { in = Conv_k5(h); acts = tanh . sigmoid; rs = Conv_1x1(acts); h += rs[0:H]; out += rs[H:2H] (last: out += rs) };
//     x1 += post(out)            (the post conv's weights are negated at load: x1 - m, vits.cpp:506,513)
// As launches a coupling layer is nine of them in these modes (converter + pre conv, fill, four WaveNet layers, converter + post conv)
// with 6 us of MFMA work per 37-44 us WaveNet layer at batch 64 x 225 frames: each layer fills its h tile from HBM, adds into h / out in
// HBM and pays a launch gap. Here a block owns 48 frames and computes on 64 columns (the 2-frame halo of each of the four k = 5 convs
// on both sides): the fp32 stream h lives in the registers of the waves that compute the res rows (channel groups 0-2), the skip sum in
// those of the waves that compute the skip rows (groups 3-5), both in the MFMA C layout; the 16-bit conv inputs (round(h), acts, round(x0),
// round(out)) take turns in two LDS tiles; HBM sees x0 once and x1 twice. Every element goes through the same roundings and the same
// MFMA chains (chunk, tap, k-half) and fp32 adds in the same order as the launch-by-launch path: bit-identical (GPU test).
// ---------------------------------------------------------------------------------------------------------------------------------
struct FlowCouple16Params {
    const float* x0;  // conditioning half, fp32 [b][F/2][t]
    int64_t x0_bs;
    int x0_cs;
    float* x1;  // updated half, in place
    int64_t x1_bs;
    int x1_cs;
    const uint16_t* w_pre;
    const float* b_pre;
    const uint16_t* w_in[4];
    const float* b_in[4];
    const uint16_t* w_rs[4];
    const float* b_rs[4];
    const uint16_t* w_post;
    const float* b_post;
    const int* lens;
    int tmax;
};

// NCW: 32-frame column tiles per wave. 1 = twelve waves (channel group, column tile); 2 = six waves that own both column tiles of their channel
// group: every weight fragment then feeds two MFMAs per row tile, which halves the fragment traffic through the CU's vector memory path
// (the bound of the gated conv, see wavenet16_kernel) — without the per-layer fill and epilogue that made it lose there.
// NCT: 32-frame column tiles per BLOCK. 2 = 48 frames on 64 columns (the throughput shape). 1 = 16 frames on 32 columns, for small grids: an
// utterance of 225 frames is FIVE blocks of the wide tile on 256 CUs (87 us per coupling layer at batch 1); fifteen narrow ones compute 1.5 x
// the columns in a third of the time. Same chains and expressions per output: bit-identical.
template <bool BF, int NCW, int NCT>
__global__ __launch_bounds__(64 * 6 * NCT / NCW, 1) void flow_couple16_kernel(const FlowCouple16Params p) {
    constexpr int H = 192, HF = 96, KT = 5, NL = 4, NG = H / 32, BM = 32 * NCT, HALO = 8, BO = BM - 2 * HALO, P = (KT - 1) / 2, XS = BM + KT - 1;
    constexpr int NGRP = H / 8;  // 16-byte channel groups of an H-channel tile
    constexpr int NTH = 64 * NG * NCT / NCW;
    static_assert(NCW <= NCT, "a wave owns at most the block's column tiles");
    static_assert(HALO == NL * P, "one k = 5 halo per WaveNet layer");
    extern __shared__ __attribute__((aligned(16))) wn_int4v l16[];
    wn_int4v* xs = l16;               // [NGRP][XS]  round(h), slot P + column (two zero slots on either side)
    wn_int4v* ts = l16 + NGRP * XS;   // [NGRP][BM]  acts | round(x0) (12 groups) | round(out)
    float* ex = reinterpret_cast<float*>(l16);  // [3][2][16][NCT][64] fp32: the last layer's rs rows on their way to the skip waves (over xs + ts)
    float* lb = reinterpret_cast<float*>(l16 + NGRP * XS + NGRP * BM);  // biases: pre[H] | in[NL][2H] | rs[NL][2H] | post[HF]
    constexpr int LB_IN = H, LB_RS = H + NL * 2 * H, LB_POST = H + 2 * NL * 2 * H, LB_N = LB_POST + HF;
    static_assert(3 * 2 * 16 * NCT * 64 * 4 <= (NGRP * XS + NGRP * BM) * 16, "exchange buffer fits over the two tiles");

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int b = blockIdx.y;
    const int len = p.lens ? p.lens[b] : p.tmax;
    if ((int)blockIdx.x * BO >= len) return;
    const int tb = (int)blockIdx.x * BO - HALO;  // global frame of column 0
    const int krow = lane >> 5;
    const int gw = wid % NG, ct0 = (wid / NG) * NCW;
    const int col0 = ct0 * 32 + (lane & 31);  // this lane's column of its column tile n: col0 + 32 n
    bool inside[NCW];
#pragma unroll
    for (int n = 0; n < NCW; ++n) inside[n] = tb + col0 + 32 * n >= 0 && tb + col0 + 32 * n < len;
    typedef const __attribute__((address_space(3))) wn_int4v* LdsV;

    // ---- x0 tile (rounded, zero outside the sequence) -> ts[0:12][BM]; zero halo slots of xs; biases -> LDS ----
    {
        const float* x0b = p.x0 + (int64_t)b * p.x0_bs;
        constexpr int NSLOT = (HF / 8) * BM, PER = NSLOT / NTH;  // 768 slots
        float v[PER][8];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int idx = tid + u * NTH;
            const int g = idx / BM, i = idx - g * BM;
            const int tt = tb + i;
            const bool ok = tt >= 0 && tt < len;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[u][e] = ok ? x0b[(int64_t)(g * 8 + e) * p.x0_cs + tt] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            wn_int4v q;
            q.x = (int)((unsigned)wn_round16<BF>(v[u][0]) | ((unsigned)wn_round16<BF>(v[u][1]) << 16));
            q.y = (int)((unsigned)wn_round16<BF>(v[u][2]) | ((unsigned)wn_round16<BF>(v[u][3]) << 16));
            q.z = (int)((unsigned)wn_round16<BF>(v[u][4]) | ((unsigned)wn_round16<BF>(v[u][5]) << 16));
            q.w = (int)((unsigned)wn_round16<BF>(v[u][6]) | ((unsigned)wn_round16<BF>(v[u][7]) << 16));
            ts[tid + u * NTH] = q;
        }
        if (tid < NGRP * 2 * P) {
            const int gg = tid / (2 * P), j = tid - gg * (2 * P);
            xs[gg * XS + (j < P ? j : BM + j)] = wn_int4v{0, 0, 0, 0};
        }
        for (int i2 = tid; i2 < LB_N; i2 += NTH) {
            float bv;
            if (i2 < LB_IN) bv = p.b_pre[i2];
            else if (i2 < LB_RS) bv = p.b_in[(i2 - LB_IN) / (2 * H)][(i2 - LB_IN) % (2 * H)];
            else if (i2 < LB_POST) {
                const int l = (i2 - LB_RS) / (2 * H), r = (i2 - LB_RS) % (2 * H);
                bv = (l + 1 < NL || r < H) ? p.b_rs[l][r] : 0.f;  // (the last layer's 1x1 conv has H rows)
            } else bv = p.b_post[i2 - LB_POST];
            lb[i2] = bv;
        }
    }
    __syncthreads();

    wn_floatx16 acc[2][NCW];
    auto mfma = [&](wn_int4v a, wn_int4v bq, wn_floatx16 c) __attribute__((always_inline)) -> wn_floatx16 {
        if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wn_bf16x8, a), __builtin_bit_cast(wn_bf16x8, bq), c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wn_half8, a), __builtin_bit_cast(wn_half8, bq), c, 0, 0, 0);
    };
    // one conv for NM row tiles mt0.. of this wave's NCW column tiles: order per output = chunk, tap, k-half (conv16.hip's, wavenet16_kernel's).
    // In two parts: conv_begin issues the loads of the first RD weight fragments, conv_run multiplies. A block is alone on its CU and runs ten
    // convs separated by barriers and tile writes: begun right behind the previous conv, a conv's first fragments travel while the gate, the
    // tile write and the barrier in front of it run — inside conv_run that L2 round trip was exposed ten times per block.
    constexpr int RS = 8, RD = 6;
    wn_int4v ring[RS][2];
    int wvoff[2];
    auto load_a = [&](const uint16_t* wp, int m, int step) __attribute__((always_inline)) -> wn_int4v {
        const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(wp), 0, 0x7fffffff, 0x00020000);
        const int vo = wvoff[m];
        return __builtin_bit_cast(wn_int4v, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, vo, step * 1024, 0));
    };
    auto conv_begin = [&](const uint16_t* wp, int mt0, auto total_c, auto nm_c) __attribute__((always_inline)) {
        constexpr int TOTAL = decltype(total_c)::value, NM = decltype(nm_c)::value;
#pragma unroll
        for (int m = 0; m < NM; ++m) wvoff[m] = (int)(((size_t)(mt0 + m) * TOTAL * 64 + lane) * 16);
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int i = 0; i < RD; ++i) ring[i][m] = load_a(wp, m, i < TOTAL ? i : TOTAL - 1);
    };
    auto conv_run = [&](const uint16_t* wp, auto total_c, auto taps_c, auto nm_c, LdsV base, const int pitch) __attribute__((always_inline)) {
        constexpr int TOTAL = decltype(total_c)::value, TAPS = decltype(taps_c)::value, NM = decltype(nm_c)::value;
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int n = 0; n < NCW; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
        auto bslot = [&](int s) __attribute__((always_inline)) -> int {
            const int kk = s & 1, cj = s >> 1, j = cj % TAPS, c = cj / TAPS;
            return (c * 4 + 2 * kk) * pitch + j;
        };
        wn_int4v b_nxt[NCW];
#pragma unroll
        for (int n = 0; n < NCW; ++n) b_nxt[n] = base[bslot(0) + 32 * n];
#pragma unroll
        for (int s = 0; s < TOTAL; ++s) {
#pragma unroll
            for (int m = 0; m < NM; ++m) ring[(s + RD) % RS][m] = load_a(wp, m, s + RD < TOTAL ? s + RD : TOTAL - 1);
            __builtin_amdgcn_sched_barrier(0);
            wn_int4v bq[NCW];
#pragma unroll
            for (int n = 0; n < NCW; ++n) {
                bq[n] = b_nxt[n];
                b_nxt[n] = base[bslot(s + 1 < TOTAL ? s + 1 : TOTAL - 1) + 32 * n];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < NM; ++m)
#pragma unroll
                for (int n = 0; n < NCW; ++n) acc[m][n] = mfma(ring[s % RS][m], bq[n], acc[m][n]);
        }
    };
    using C_PRE = std::integral_constant<int, (HF / 32) * 2>;
    using C_IN = std::integral_constant<int, NG * KT * 2>;
    using C_RS = std::integral_constant<int, NG * 2>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using IK = std::integral_constant<int, KT>;
    // 16 values of one row tile (MFMA C layout: register 4 g + e = channel 8 g + 4 krow + e of the tile) -> rounded, whole 16-byte slots of the
    // tile's four channel groups at this lane's column (v_permlane32_swap trades halves between two groups, see wavenet16_kernel)
    auto put_tile = [&](wn_int4v* dst, int grp0, int pitch, int slot, const float* v, bool ok) __attribute__((always_inline)) {
        unsigned w[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            unsigned short q[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) q[e] = ok ? wn_round16<BF>(v[4 * g + e]) : (unsigned short)0;
            w[g][0] = (unsigned)q[0] | ((unsigned)q[1] << 16);
            w[g][1] = (unsigned)q[2] | ((unsigned)q[3] << 16);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const auto x = __builtin_amdgcn_permlane32_swap(w[2 * k][0], w[2 * k + 1][0], false, false);
            const auto y = __builtin_amdgcn_permlane32_swap(w[2 * k][1], w[2 * k + 1][1], false, false);
            dst[(grp0 + 2 * k + krow) * pitch + slot] = wn_int4v{(int)x[0], (int)y[0], (int)x[1], (int)y[1]};
        }
    };
    auto rowof = [&](int tile, int r) __attribute__((always_inline)) -> int { return tile * 32 + (r >> 2) * 8 + krow * 4 + (r & 3); };

    // ---- pre conv (F/2 -> H, 1x1): the waves of channel groups 0-2 compute the rows they will carry as h ----
    // st[m][n][r]: groups 0-2: the fp32 stream h, rows 64 gw + 32 m + ...; groups 3-5: the skip sum `outputs`, rows 64 (gw - 3) + 32 m + ...
    float st[2][NCW][16];
    if (gw < 3) {
        conv_begin(p.w_pre, 2 * gw, C_PRE{}, I2{});
        conv_run(p.w_pre, C_PRE{}, I1{}, I2{}, (LdsV)(ts + krow * BM + col0), BM);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NCW; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[m][n][r] = acc[m][n][r] + lb[rowof(2 * gw + m, r)];
    } else {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NCW; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[m][n][r] = 0.f;  // (launch_fill_rows of the launch-by-launch path)
    }

    conv_begin(p.w_in[0], 2 * gw, C_IN{}, I2{});
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        __syncthreads();  // the previous layer's 1x1 conv (l = 0: the pre conv) has read ts; its gated conv has read xs
        if (gw < 3) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < NCW; ++n) put_tile(xs, (2 * gw + m) * 4, XS, P + col0 + 32 * n, st[m][n], inside[n]);
        }
        __syncthreads();
        // gated conv: tanh tile 2 gw, sigmoid tile 2 gw + 1 of this wave's channel group
        conv_run(p.w_in[l], C_IN{}, IK{}, I2{}, (LdsV)(xs + krow * XS + col0), XS);
        const bool do_rs = l + 1 < NL || gw < 3;
        if (do_rs) conv_begin(p.w_rs[l], 2 * gw, C_RS{}, I2{});
#pragma unroll
        for (int n = 0; n < NCW; ++n) {
            float a16[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = gw * 32 + (r >> 2) * 8 + krow * 4 + (r & 3);
                float v = wavenet_gate(acc[0][n][r] + lb[LB_IN + l * 2 * H + ch], acc[1][n][r] + lb[LB_IN + l * 2 * H + H + ch]);
                asm volatile("" : "+v"(v));  // (two roundings, as the launch-by-launch path: fp32 acts, 16-bit at the converter)
                a16[r] = v;
            }
            put_tile(ts, gw * 4, BM, col0 + 32 * n, a16, inside[n]);
        }
        __syncthreads();
        // 1x1 res/skip conv: tiles 2 gw, 2 gw + 1 of 2H rows (the last layer: H rows, channel groups 0-2 only)
        if (do_rs) {
            conv_run(p.w_rs[l], C_RS{}, I1{}, I2{}, (LdsV)(ts + krow * BM + col0), BM);
            if (l + 1 < NL) conv_begin(p.w_in[l + 1 < NL ? l + 1 : l], 2 * gw, C_IN{}, I2{});
            if (l + 1 < NL) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < NCW; ++n)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float v = acc[m][n][r] + lb[LB_RS + l * 2 * H + rowof(2 * gw + m, r)];
                            st[m][n][r] = st[m][n][r] + v;  // h' = h + rs[0:H] (groups 0-2), outputs += rs[H:2H] (groups 3-5)
                        }
            }
        }
    }
    // ---- the last layer's rs rows (channel groups 0-2) join the skip sum held by groups 3-5: lane for lane through LDS ----
    __syncthreads();  // every wave has left the last 1x1 conv: xs and ts are free
    if (gw < 3) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NCW; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ex[(((gw * 2 + m) * 16 + r) * NCT + ct0 + n) * 64 + lane] = acc[m][n][r] + lb[LB_RS + (NL - 1) * 2 * H + rowof(2 * gw + m, r)];
    }
    if (gw < 3) conv_begin(p.w_post, gw, C_RS{}, I1{});
    __syncthreads();
    if (gw >= 3) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NCW; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[m][n][r] = st[m][n][r] + ex[((((gw - 3) * 2 + m) * 16 + r) * NCT + ct0 + n) * 64 + lane];
    }
    __syncthreads();  // ex read: ts may be written
    if (gw >= 3) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NCW; ++n) put_tile(ts, (2 * (gw - 3) + m) * 4, BM, col0 + 32 * n, st[m][n], inside[n]);
    }
    __syncthreads();
    // ---- post conv (H -> F/2, 1x1) and the coupling: x1 += post(out) on the block's own 48 frames ----
    if (gw < 3) {
        conv_run(p.w_post, C_RS{}, I1{}, I1{}, (LdsV)(ts + krow * BM + col0), BM);
        float addv[NCW][16];
#pragma unroll
        for (int n = 0; n < NCW; ++n) {
            const int col = col0 + 32 * n, t = tb + col;
            const bool mine = col >= HALO && col < HALO + BO && t < len;
#pragma unroll
            for (int r = 0; r < 16; ++r) addv[n][r] = mine ? p.x1[(int64_t)b * p.x1_bs + (int64_t)rowof(gw, r) * p.x1_cs + t] : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NCW; ++n) {
            const int col = col0 + 32 * n, t = tb + col;
            if (!(col >= HALO && col < HALO + BO && t < len)) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[0][n][r] + lb[LB_POST + rowof(gw, r)];
                v = addv[n][r] + v;
                p.x1[(int64_t)b * p.x1_bs + (int64_t)rowof(gw, r) * p.x1_cs + t] = v;
            }
        }
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------------
bool wavenet32_supported(int hidden, int kt, int dil, const PackedConv& in, const PackedConv& rs) {
    if (hidden != 192 || kt != 5 || dil != 1) return false;
    if (!in.wp || !rs.wp || !in.bias || !rs.bias || in.epi != EPI_GATE || rs.epi != EPI_STD) return false;
    if (in.cin != hidden || in.cout != 2 * hidden || in.kt != kt || rs.cin != hidden || rs.kt != 1) return false;
    return rs.cout == 2 * hidden || rs.cout == hidden;
}

hipError_t launch_wavenet32(const PackedConv& in, const PackedConv& rs, const WaveNet32Call& c, hipStream_t s) {
    constexpr int H = 192, KT = 5;
    if (!wavenet32_supported(c.hidden, in.kt, c.dil, in, rs)) return hipErrorInvalidValue;
    if ((c.h.cs & 3) || (c.h.bs & 3) || (reinterpret_cast<uintptr_t>(c.h.p) & 15)) return hipErrorInvalidValue;
    if (rs.cout == 2 * H && (!c.h_out.p || c.h_out.p == c.h.p)) return hipErrorInvalidValue;
    WaveNet32Params p;
    p.h = c.h.p;
    p.h_bs = c.h.bs;
    p.h_cs = c.h.cs;
    p.h_out = c.h_out.p;
    p.ho_bs = c.h_out.bs;
    p.ho_cs = c.h_out.cs;
    p.outputs = c.outputs.p;
    p.o_bs = c.outputs.bs;
    p.o_cs = c.outputs.cs;
    p.w_in = in.wp;
    p.b_in = in.bias;
    p.w_rs = rs.wp;
    p.b_rs = rs.bias;
    p.rs_rows = rs.cout;
    p.lens = c.lens;
    p.tmax = c.tmax;
    constexpr int XWP = (64 + KT - 1 + 3 + 3) / 4 * 4;
    const size_t ldsz = ((size_t)H * XWP * sizeof(float) + 1023) / 1024 * 1024;
    dim3 grid((c.tmax + 63) / 64, c.batch);
    VITS_KLAUNCH((wavenet32_kernel<H, KT>), grid, dim3(4 * H), ldsz, s, p);
    return hipGetLastError();
}

bool wavenet16_supported(int hidden, int kt, int dil, const PackedConv& in, const PackedConv& rs) {
    if (hidden != 192 || kt != 5 || dil != 1) return false;
    if (!in.wp16 || !rs.wp16 || !in.bias || !rs.bias || in.epi != EPI_GATE || rs.epi != EPI_STD) return false;
    if (in.cin != hidden || in.cout != 2 * hidden || in.kt != kt || rs.cin != hidden || rs.kt != 1) return false;
    return rs.cout == 2 * hidden || rs.cout == hidden;
}

hipError_t launch_wavenet16(const PackedConv& in, const PackedConv& rs, const WaveNet32Call& c, int arith, hipStream_t s) {
    constexpr int H = 192, KT = 5;
    if (!wavenet16_supported(c.hidden, in.kt, c.dil, in, rs) || arith == VITS_ARITH_F32) return hipErrorInvalidValue;
    if (rs.cout == 2 * H && (!c.h_out.p || c.h_out.p == c.h.p)) return hipErrorInvalidValue;
    WaveNet16Params p;
    p.f.h = c.h.p;
    p.f.h_bs = c.h.bs;
    p.f.h_cs = c.h.cs;
    p.f.h_out = c.h_out.p;
    p.f.ho_bs = c.h_out.bs;
    p.f.ho_cs = c.h_out.cs;
    p.f.outputs = c.outputs.p;
    p.f.o_bs = c.outputs.bs;
    p.f.o_cs = c.outputs.cs;
    p.f.w_in = nullptr;
    p.f.b_in = in.bias;
    p.f.w_rs = nullptr;
    p.f.b_rs = rs.bias;
    p.f.rs_rows = rs.cout;
    p.f.lens = c.lens;
    p.f.tmax = c.tmax;
    p.w_in16 = in.wp16;
    p.w_rs16 = rs.wp16;
    const size_t ldsz = (size_t)(H / 8) * (64 + KT - 1) * 16;
    dim3 grid((c.tmax + 63) / 64, c.batch);
    const int ncw = kernel_knobs().wn16_ncw;  // (2: six waves, both column tiles each — measured 40.6 vs 38.4 us per layer)
    if (ncw == 1) {
        if (arith == VITS_ARITH_BF16) VITS_KLAUNCH((wavenet16_kernel<H, KT, true, 1>), grid, dim3(4 * H), ldsz, s, p);
        else VITS_KLAUNCH((wavenet16_kernel<H, KT, false, 1>), grid, dim3(4 * H), ldsz, s, p);
    } else {
        if (arith == VITS_ARITH_BF16) VITS_KLAUNCH((wavenet16_kernel<H, KT, true, 2>), grid, dim3(2 * H), ldsz, s, p);
        else VITS_KLAUNCH((wavenet16_kernel<H, KT, false, 2>), grid, dim3(2 * H), ldsz, s, p);
    }
    return hipGetLastError();
}

bool flow_couple16_supported(int hidden, int half, int kt, int rate, int layers, const PackedConv& pre, const PackedConv* in, const PackedConv* rs, const PackedConv& post) {
    if (hidden != 192 || half != 96 || kt != 5 || rate != 1 || layers != 4) return false;
    if (!pre.wp16 || !pre.bias || pre.cin != half || pre.cout != hidden || pre.kt != 1 || pre.epi != EPI_STD) return false;
    if (!post.wp16 || !post.bias || post.cin != hidden || post.cout != half || post.kt != 1 || post.epi != EPI_STD) return false;
    for (int l = 0; l < layers; ++l) {
        if (!wavenet16_supported(hidden, kt, 1, in[l], rs[l])) return false;
        if (rs[l].cout != (l + 1 < layers ? 2 * hidden : hidden)) return false;
    }
    return true;
}

hipError_t launch_flow_couple16(const PackedConv& pre, const PackedConv* in, const PackedConv* rs, const PackedConv& post, const FlowCouple16Call& c, int arith,
                                hipStream_t s) {
    if (arith == VITS_ARITH_F32 || !flow_couple16_supported(c.hidden, c.half, 5, 1, 4, pre, in, rs, post) || !c.x0.p || !c.x1.p) return hipErrorInvalidValue;
    FlowCouple16Params p;
    p.x0 = c.x0.p;
    p.x0_bs = c.x0.bs;
    p.x0_cs = c.x0.cs;
    p.x1 = c.x1.p;
    p.x1_bs = c.x1.bs;
    p.x1_cs = c.x1.cs;
    p.w_pre = pre.wp16;
    p.b_pre = pre.bias;
    for (int l = 0; l < 4; ++l) {
        p.w_in[l] = in[l].wp16;
        p.b_in[l] = in[l].bias;
        p.w_rs[l] = rs[l].wp16;
        p.b_rs[l] = rs[l].bias;
    }
    p.w_post = post.wp16;
    p.b_post = post.bias;
    p.lens = c.lens;
    p.tmax = c.tmax;
    constexpr int H = 192, NGRP = H / 8, LB_N = H + 2 * 4 * 2 * H + 96;
    // small grids (batch 1 ... 4 at 225 frames): 16-frame blocks on one 32-column tile (see the kernel)
    const int64_t wide_blocks = (int64_t)((c.tmax + 47) / 48) * c.batch;
    if (wide_blocks <= kernel_knobs().flow_narrow_max) {
        const size_t ldsz = (size_t)(NGRP * (32 + 4) + NGRP * 32) * 16 + (size_t)LB_N * 4;
        dim3 grid((c.tmax + 15) / 16, c.batch);
        if (arith == VITS_ARITH_BF16) VITS_KLAUNCH((flow_couple16_kernel<true, 1, 1>), grid, dim3(384), ldsz, s, p);
        else VITS_KLAUNCH((flow_couple16_kernel<false, 1, 1>), grid, dim3(384), ldsz, s, p);
        return hipGetLastError();
    }
    const size_t ldsz = (size_t)(NGRP * (64 + 4) + NGRP * 64) * 16 + (size_t)LB_N * 4;
    // six waves owning both column tiles of their channel group (VITS_FLOW_NCW=1: twelve waves, one column tile each)
    const int ncw = kernel_knobs().flow_ncw;
    dim3 grid((c.tmax + 47) / 48, c.batch);
    if (ncw == 1) {
        if (arith == VITS_ARITH_BF16) VITS_KLAUNCH((flow_couple16_kernel<true, 1, 2>), grid, dim3(768), ldsz, s, p);
        else VITS_KLAUNCH((flow_couple16_kernel<false, 1, 2>), grid, dim3(768), ldsz, s, p);
    } else {
        if (arith == VITS_ARITH_BF16) VITS_KLAUNCH((flow_couple16_kernel<true, 2, 2>), grid, dim3(384), ldsz, s, p);
        else VITS_KLAUNCH((flow_couple16_kernel<false, 2, 2>), grid, dim3(384), ldsz, s, p);
    }
    return hipGetLastError();
}

}  // namespace vits
