// stage1_lat.hip — the duration predictor's DDS layers for LATENCY-bound launches (batch 1, a few short utterances): round 6.
//
// At batch 1 x 128 ids the stochastic duration predictor (vits.cpp:927-972) was 26 launches and 0.35 ms of the 0.82 ms stage one: twelve
// dds_layer_kernel launches of 20 us each (per-block phase stamps, tools/dds_micro.hip: loads 3.8 | depthwise + LN1 + gelu 4.7 | 1x1 conv 5.0 |
// LN2 + gelu + out 3.2 us on FOUR blocks), each flanked by small launches — the 1 -> H conv of a conv flow (4.7 us), its projection (6.1 us), the
// 1x1 convs around the first DDS block (6.1 + 6.2 us). A grid-wide barrier costs three kernel boundaries on this chip (tools/grid_barrier_micro.hip:
// 7.9 against 2.55 us per dependent step at 256 blocks), so the answer is not one persistent kernel but FEWER, FASTER launches:
//   * 16 tokens per block instead of 32 (twice the blocks), twelve waves instead of eight;
//   * the element-wise phases are element-parallel: a thread owns FOUR elements of one token column instead of twelve (each gelu is an erff);
//     only the LayerNorm partial sums keep their 16 channel groups, and every sum keeps its order — channel group g adds channels g, g + 16, ...
//     in ascending order, the groups are combined in ascending order — so every float is the one dds_layer_kernel (misc_kernels.hip) computes;
//   * the 1x1 conv runs on v_mfma_f32_16x16x4_f32 tiles (one 16-row tile per wave, 48 dependent MFMAs instead of 96 twice as long): bit for bit
//     the same sequential fmaf chain over the input channels as v_mfma_f32_32x32x2_f32 (tools/mfma_bits.hip), from the layer's second weight copy in
//     that instruction's operand order (repack_conv_weights_l16, conv_mfma.hip), all 12 quads of a wave fetched at the kernel's first instruction;
//   * HEAD: the layer computes its own input where that is a per-token function of something smaller — the conv flow's 1 -> H conv + conditioning
//     (vits.cpp:864 + :651-653, pointwise_from1_kernel) or the predictor's first 1x1 conv (vits.cpp:939, conv_pre) — for its tile AND halo;
//   * TAIL: the layer applies the 1x1 conv that consumes its output (the conv flow's projection H -> 3 bins - 1, vits.cpp:869; the predictor's
//     H -> H projection, :941) from the output tile in LDS, and only that result goes to memory.
// Same expressions as the kernels it replaces, statement by statement (hipcc contracts a * b + c the same way in both): the engine takes this path for
// small grids only (Engine::run_dds, VITS_DDS_LAT_MAX_BLOCKS), and tests/test_gpu_edge_and_scale.py compares the two bit for bit.
// fp32 arithmetic only (stage one is exact fp32 under the default arithmetic scope; VITS_ARITH_SCOPE_ALL_CONVS with 16-bit operands keeps dds_layer_kernel).
#include <hip/hip_runtime.h>

#include <cmath>

#include "../../include/vits.h"
#include "kernels.h"

namespace vits {

namespace {

constexpr int LAT_NT = 16;      // tokens per block = one MFMA column tile
constexpr int LAT_GROUPS = 16;  // LayerNorm channel groups (add_layer_norm_kernel / dds_layer_kernel: LN_GROUPS)
typedef float lat_float4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float lat_table_lookup(const uint16_t* tab, float x) {
    const uint16_t i = __builtin_bit_cast(uint16_t, (_Float16)x);
    return (float)__builtin_bit_cast(_Float16, tab[i]);
}
__device__ __forceinline__ float lat_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float lat_gelu_op(float x, const uint16_t* gelu_tab) { return gelu_tab ? lat_table_lookup(gelu_tab, x) : lat_gelu_erf(x); }

}  // namespace

enum { DDS_HEAD_NONE = 0, DDS_HEAD_FROM1 = 1, DDS_HEAD_CONV = 2 };
enum { DDS_TAIL_NONE = 0, DDS_TAIL_PROJ = 1 };

struct DdsLatParams {
    const float* x;  // HEAD_NONE: layer input [B][H][T]; HEAD_CONV: the head conv's input [B][h_cin][T]
    int64_t x_bs;
    int x_cs;
    const float* z;  // HEAD_FROM1: latent row zc of [B][2][T]
    int64_t z_bs;
    int z_cs, zc;
    const float* cond;  // HEAD_FROM1: conditioning [B][H][T]
    int64_t c_bs;
    int c_cs;
    const float *h_w, *h_b;  // HEAD_FROM1: w [H], bias [H]; HEAD_CONV: bias [H]
    const float* h_wl16;     // HEAD_CONV: 16x16x4 A fragments of the H x h_cin 1x1 conv
    int h_nchunks;
    const float *dw_w, *dw_b, *g1, *b1, *pw_b, *g2, *b2;
    const float* wl16;  // the layer's pointwise conv, 16x16x4 A fragments
    int nchunks;
    float* y;  // TAIL_NONE: layer output [B][H][T]
    int64_t y_bs;
    int y_cs;
    const float *t_wl16, *t_b;  // TAIL_PROJ: fragments, bias of the t_rows x H 1x1 conv; output y2
    int t_rows, t_mtiles;
    float* y2;
    int64_t y2_bs;
    int y2_cs;
    const int* lens;
    int H, tmax, k, dil;
    float eps;
    const uint16_t* gelu_tab;
};

// one 16 x 16 output tile of a 1x1 conv: rows of fragment stream `frag` (= wl16 + ((mtile * 2 + half) * NQ) * 256 floats, NQ = 2 * nchunks quads of
// 64 lanes x float4), B operands from the LDS tile bt[channel][pitch] at column `col`; the chain runs over the channels in ascending order
template <int MAXQ>
__device__ __forceinline__ void lat_load_quads(lat_float4v (&aq)[MAXQ], const float* frag, int nq, int lane) {
    const lat_float4v* f4 = reinterpret_cast<const lat_float4v*>(frag) + lane;
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) aq[q] = f4[(q < nq ? q : 0) * 64];
}
template <int MAXQ>
__device__ __forceinline__ lat_float4v lat_chain(const lat_float4v (&aq)[MAXQ], int nq, const float* bt, int pitch, int jg, int col) {
    lat_float4v acc = {0.f, 0.f, 0.f, 0.f};
    const float* b0 = bt + jg * pitch + col;
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        if (q < nq) {
            const float* bq = b0 + (16 * q) * pitch;
            const float v0 = bq[0], v1 = bq[4 * pitch], v2 = bq[8 * pitch], v3 = bq[12 * pitch];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[q][0], v0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[q][1], v1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[q][2], v2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[q][3], v3, acc, 0, 0, 0);
        }
    }
    return acc;
}

// MAXCH: upper bound of channels / 32 (6: the MMS-TTS architecture's 192 channels; 8: up to 256). Block = H / 16 waves.
template <int HEAD, int TAIL, int MAXCH>
__global__ __launch_bounds__(MAXCH * 2 * 64) void dds_layer_lat_kernel(DdsLatParams p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int NT = LAT_NT, MAXQ = 2 * MAXCH;
    const int H = p.H, nthr = (int)blockDim.x;
    const int pad = (p.k * p.dil - p.dil) / 2;  // vits.cpp:660
    const int xw = NT + 2 * pad;
    const int hcols = HEAD == DDS_HEAD_CONV ? (xw + 15) & ~15 : 0;  // columns the head conv computes (whole MFMA column tiles)
    float* xt = sm;                                  // [H][xw] layer input with halo (also the residual)
    float* ht = xt + ((H * xw + 3) & ~3);            // [H][NT] depthwise -> gelu(LN1) -> pointwise -> output tile (TAIL_PROJ)
    float* red = ht + H * NT;                        // [2][GROUPS][NT]
    float* prm = red + 2 * LAT_GROUPS * NT;          // [6][H] dw_b g1 b1 pw_b g2 b2, then [H][k] dw_w
    float* et = prm + ((6 * H + H * p.k + 3) & ~3);  // HEAD_CONV: [h_cin][hcols] the head conv's input tile
    const int b = blockIdx.y, t0 = blockIdx.x * NT;
    const int len = p.lens ? p.lens[b] : p.tmax;
    if (t0 >= len) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tl = tid & 15, rr = tid >> 4, rstep = nthr >> 4;  // element-parallel phases: column tl, channels rr, rr + rstep, ...
    const int jg = lane >> 4, col = lane & 15;
    // ---- everything this block reads from memory, issued at once: the wave's weight fragments first (they are needed last) ----
    lat_float4v aq[MAXQ];
    const int nq = 2 * p.nchunks;
    lat_load_quads<MAXQ>(aq, p.wl16 + (size_t)wid * nq * 256, nq, lane);
    lat_float4v hq[HEAD == DDS_HEAD_CONV ? MAXQ : 1];
    if constexpr (HEAD == DDS_HEAD_CONV) lat_load_quads<MAXQ>(hq, p.h_wl16 + (size_t)wid * (2 * p.h_nchunks) * 256, 2 * p.h_nchunks, lane);
    {
        float pv[2][6], wv[2] = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int c = tid + r * nthr < H ? tid + r * nthr : 0;
            pv[r][0] = p.dw_b[c];
            pv[r][1] = p.g1[c];
            pv[r][2] = p.b1[c];
            pv[r][3] = p.pw_b[c];
            pv[r][4] = p.g2[c];
            pv[r][5] = p.b2[c];
            if (tid + r * nthr < H * p.k) wv[r] = p.dw_w[tid + r * nthr];
        }
        constexpr int XB = 12;  // elements of the input tile per thread and pass, all loads in flight
        if constexpr (HEAD == DDS_HEAD_NONE) {
            const float* xb = p.x + (int64_t)b * p.x_bs;
            const int total = H * xw;
            for (int base = tid; base < total; base += XB * nthr) {
                float v[XB];
#pragma unroll
                for (int u = 0; u < XB; ++u) {
                    const int e = base + u * nthr, c = e / xw, i = e - c * xw, t = t0 - pad + i;
                    v[u] = (e < total && t >= 0 && t < len) ? xb[(int64_t)c * p.x_cs + t] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < XB; ++u)
                    if (base + u * nthr < total) xt[base + u * nthr] = v[u];
            }
        } else if constexpr (HEAD == DDS_HEAD_FROM1) {
            // conv_pre of a conv flow (vits.cpp:864) + "inputs + global_conditioning" (:651-653): pointwise_from1_kernel's expression, zero outside the sequence
            const float* zb = p.z + (int64_t)b * p.z_bs + (int64_t)p.zc * p.z_cs;
            const float* cb = p.cond + (int64_t)b * p.c_bs;
            const int total = H * xw;
            for (int base = tid; base < total; base += XB * nthr) {
                float zv[XB], cv[XB], ww[XB], bb[XB];
                bool ok[XB];
#pragma unroll
                for (int u = 0; u < XB; ++u) {
                    const int e = base + u * nthr, c = e < total ? e / xw : 0, i = e - c * xw, t = t0 - pad + i;
                    ok[u] = e < total && t >= 0 && t < len;
                    zv[u] = ok[u] ? zb[t] : 0.f;
                    cv[u] = ok[u] ? cb[(int64_t)c * p.c_cs + t] : 0.f;
                    ww[u] = p.h_w[c];
                    bb[u] = p.h_b[c];
                }
#pragma unroll
                for (int u = 0; u < XB; ++u) {
                    if (base + u * nthr >= total) continue;
                    float v = ww[u] * zv[u] + bb[u];
                    v = v + cv[u];
                    xt[base + u * nthr] = ok[u] ? v : 0.f;
                }
            }
        } else {
            // the head conv's input tile [h_cin][hcols]: columns t0 - pad .. (zero outside the sequence and beyond the xw columns that are needed)
            const float* xb = p.x + (int64_t)b * p.x_bs;
            const int cin = p.h_nchunks * 32, total = cin * hcols;
            for (int base = tid; base < total; base += XB * nthr) {
                float v[XB];
#pragma unroll
                for (int u = 0; u < XB; ++u) {
                    const int e = base + u * nthr, c = e / hcols, i = e - c * hcols, t = t0 - pad + i;
                    v[u] = (e < total && c < H && i < xw && t >= 0 && t < len) ? xb[(int64_t)c * p.x_cs + t] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < XB; ++u)
                    if (base + u * nthr < total) et[base + u * nthr] = v[u];
            }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (tid + r * nthr < H) {
#pragma unroll
                for (int q = 0; q < 6; ++q) prm[q * H + tid + r * nthr] = pv[r][q];
            }
            if (tid + r * nthr < H * p.k) prm[6 * H + tid + r * nthr] = wv[r];
        }
        for (int q = tid + 2 * nthr; q < H * p.k; q += nthr) prm[6 * H + q] = p.dw_w[q];
    }
    __syncthreads();
    if constexpr (HEAD == DDS_HEAD_CONV) {
        // x = W_pre . enc + b (vits.cpp:939; conv_lat16_kernel's chain and epilogue), every column tile of the halo'd tile; zero outside the sequence
        const int nct = hcols >> 4;
        for (int ct = 0; ct < nct; ++ct) {
            const lat_float4v acc = lat_chain<MAXQ>(hq, 2 * p.h_nchunks, et, hcols, jg, ct * 16 + col);
            const int i = ct * 16 + col, t = t0 - pad + i;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wid * 16 + 4 * jg + r;
                const float v = acc[r] + p.h_b[row];
                if (i < xw) xt[row * xw + i] = (t >= 0 && t < len) ? v : 0.f;
            }
        }
        __syncthreads();
    }
    const float *dw_b = prm, *g1 = prm + H, *b1 = prm + 2 * H, *pw_b = prm + 3 * H, *g2 = prm + 4 * H, *b2 = prm + 5 * H, *dw_w = prm + 6 * H;
    // ---- depthwise conv (dds_depthwise_kernel's chain: bias, then the taps in ascending order) ----
    for (int c = rr; c < H; c += rstep) {
        float a = dw_b[c];
        for (int j = 0; j < p.k; ++j) a += dw_w[c * p.k + j] * xt[c * xw + tl + j * p.dil];
        ht[c * NT + tl] = a;
    }
    __syncthreads();
    // ---- LayerNorm over the channels of a token: 16 channel-group partial sums in ascending channel order, combined in ascending group order ----
    auto layer_norm_stats = [&](float& mean, float& inv) __attribute__((always_inline)) {
        if (tid < LAT_GROUPS * NT) {
            const int gq = tid >> 4;
            float s = 0.f;
            for (int c = gq; c < H; c += LAT_GROUPS) s += ht[c * NT + tl];
            red[gq * NT + tl] = s;
        }
        __syncthreads();
        float msum = 0.f;
#pragma unroll
        for (int q = 0; q < LAT_GROUPS; ++q) msum += red[q * NT + tl];
        mean = msum / (float)H;
        if (tid < LAT_GROUPS * NT) {
            const int gq = tid >> 4;
            float vs = 0.f;
            for (int c = gq; c < H; c += LAT_GROUPS) {
                const float d = ht[c * NT + tl] - mean;
                vs += d * d;
            }
            red[(LAT_GROUPS + gq) * NT + tl] = vs;
        }
        __syncthreads();
        float vsum = 0.f;
#pragma unroll
        for (int q = 0; q < LAT_GROUPS; ++q) vsum += red[(LAT_GROUPS + q) * NT + tl];
        const float var = vsum / (float)H;
        inv = 1.0f / sqrtf(var + p.eps);
    };
    {
        float mean, inv;
        layer_norm_stats(mean, inv);
        for (int c = rr; c < H; c += rstep) {
            const float hv = ht[c * NT + tl];
            ht[c * NT + tl] = lat_gelu_op((hv - mean) * inv * g1[c] + b1[c], p.gelu_tab);
        }
    }
    __syncthreads();
    // ---- pointwise conv: wave w owns output rows 16w .. 16w + 15 ----
    {
        const lat_float4v acc = lat_chain<MAXQ>(aq, nq, ht, NT, jg, col);
        __syncthreads();  // every wave is done reading ht
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wid * 16 + 4 * jg + r;
            ht[row * NT + col] = acc[r] + pw_b[row];
        }
    }
    // (TAIL_PROJ: the tail conv's fragments travel while LayerNorm 2 runs)
    lat_float4v tq[TAIL == DDS_TAIL_PROJ ? MAXQ : 1];
    const bool tail_wave = TAIL == DDS_TAIL_PROJ && wid < 2 * p.t_mtiles && wid * 16 < p.t_rows;
    if constexpr (TAIL == DDS_TAIL_PROJ) {
        if (tail_wave) lat_load_quads<MAXQ>(tq, p.t_wl16 + (size_t)wid * nq * 256, nq, lane);
    }
    __syncthreads();
    // ---- LayerNorm 2 + gelu + residual (add_layer_norm_kernel with post_gelu and add_to) ----
    {
        float mean, inv;
        layer_norm_stats(mean, inv);
        const int t = t0 + tl;
        for (int c = rr; c < H; c += rstep) {
            float v = (ht[c * NT + tl] - mean) * inv * g2[c] + b2[c];
            v = lat_gelu_op(v, p.gelu_tab);
            asm volatile("" : "+v"(v));  // (the three-launch path adds in a separate statement behind a branch: no fma of gelu's last product with this add)
            const float o = xt[c * xw + pad + tl] + v;
            if constexpr (TAIL == DDS_TAIL_NONE) {
                if (t < len) p.y[(int64_t)b * p.y_bs + (int64_t)c * p.y_cs + t] = o;
            } else {
                ht[c * NT + tl] = o;  // (this thread's own element: nobody else reads it before the barrier)
            }
        }
    }
    if constexpr (TAIL == DDS_TAIL_PROJ) {
        __syncthreads();
        if (tail_wave) {
            const lat_float4v acc = lat_chain<MAXQ>(tq, nq, ht, NT, jg, col);
            const int t = t0 + col;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wid * 16 + 4 * jg + r;
                if (row < p.t_rows && t < len) p.y2[(int64_t)b * p.y2_bs + (int64_t)row * p.y2_cs + t] = acc[r] + p.t_b[row];
            }
        }
    }
}

static size_t dds_lat_lds(int H, int k, int dil, int head, int h_cin) {
    const int xw = LAT_NT + (k * dil - dil);
    size_t f = ((size_t)H * xw + 3) / 4 * 4 + (size_t)H * LAT_NT + 2 * LAT_GROUPS * LAT_NT + ((size_t)H * (6 + k) + 3) / 4 * 4;
    if (head == DDS_HEAD_CONV) f += (size_t)h_cin * ((xw + 15) & ~15);
    return f * sizeof(float);
}

bool dds_layer_lat_supported(const PackedConv& pw, int channels, int k, int dil) {
    if (channels < 32 || (channels & 31) || channels > 256) return false;
    if (pw.cin != channels || pw.cout != channels || pw.kt != 1 || pw.epi != EPI_STD || !pw.bias || !pw.wp_l16) return false;
    if (k < 1 || dil < 1 || ((k * dil - dil) & 1)) return false;
    return dds_lat_lds(channels, k, dil, DDS_HEAD_CONV, channels) <= 150 * 1024;
}

hipError_t launch_dds_layer_lat(const DdsLatCall& c, hipStream_t s) {
    const PackedConv& pw = *c.pw;
    if (!dds_layer_lat_supported(pw, c.channels, c.k, c.dil)) return hipErrorInvalidValue;
    DdsLatParams p{};
    p.H = c.channels;
    p.tmax = c.tmax;
    p.k = c.k;
    p.dil = c.dil;
    p.eps = c.eps;
    p.gelu_tab = c.tabs.gelu;
    p.lens = c.lens;
    p.dw_w = c.dw_w, p.dw_b = c.dw_b, p.g1 = c.g1, p.b1 = c.b1, p.pw_b = pw.bias, p.g2 = c.g2, p.b2 = c.b2;
    p.wl16 = pw.wp_l16;
    p.nchunks = pw.nchunks;
    int head = DDS_HEAD_NONE, tail = DDS_TAIL_NONE;
    if (c.head_w) {  // conv flow: 1 -> H conv of latent row zc + conditioning
        head = DDS_HEAD_FROM1;
        p.z = c.z.p, p.z_bs = c.z.bs, p.z_cs = c.z.cs, p.zc = c.zc;
        p.cond = c.cond.p, p.c_bs = c.cond.bs, p.c_cs = c.cond.cs;
        p.h_w = c.head_w, p.h_b = c.head_b;
        if (!c.z.p || !c.cond.p || !c.head_b) return hipErrorInvalidValue;
    } else if (c.head_conv) {  // H -> H 1x1 conv in front
        const PackedConv& hc = *c.head_conv;
        if (hc.cin != c.channels || hc.cout != c.channels || hc.kt != 1 || hc.epi != EPI_STD || !hc.bias || !hc.wp_l16) return hipErrorInvalidValue;
        head = DDS_HEAD_CONV;
        p.x = c.x.p, p.x_bs = c.x.bs, p.x_cs = c.x.cs;
        p.h_wl16 = hc.wp_l16, p.h_b = hc.bias, p.h_nchunks = hc.nchunks;
    } else {
        p.x = c.x.p, p.x_bs = c.x.bs, p.x_cs = c.x.cs;
    }
    if (c.tail_conv) {
        const PackedConv& tc = *c.tail_conv;
        if (tc.cin != c.channels || tc.cout < 1 || tc.cout > c.channels || tc.kt != 1 || tc.epi != EPI_STD || !tc.bias || !tc.wp_l16 || !c.y2.p) return hipErrorInvalidValue;
        tail = DDS_TAIL_PROJ;
        p.t_wl16 = tc.wp_l16, p.t_b = tc.bias, p.t_rows = tc.cout, p.t_mtiles = tc.mtiles;
        p.y2 = c.y2.p, p.y2_bs = c.y2.bs, p.y2_cs = c.y2.cs;
    } else {
        if (!c.y.p || c.y.p == c.x.p) return hipErrorInvalidValue;
        p.y = c.y.p, p.y_bs = c.y.bs, p.y_cs = c.y.cs;
    }
    const size_t lds = dds_lat_lds(c.channels, c.k, c.dil, head, c.channels);
    dim3 grid((c.tmax + LAT_NT - 1) / LAT_NT, c.batch);
    const dim3 block(c.channels / 16 * 64);
#define VITS_DDSL(HD, TL, M)                                                                                                                 \
    do {                                                                                                                                     \
        static BigLdsOnce big;                                                                                                               \
        if (lds > 64 * 1024 && big.needed()) {                                                                                               \
            if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dds_layer_lat_kernel<HD, TL, M>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) \
                return e;                                                                                                                    \
            big.done();                                                                                                                      \
        }                                                                                                                                    \
        VITS_KLAUNCH((dds_layer_lat_kernel<HD, TL, M>), grid, block, lds, s, p);                                                             \
    } while (0)
#define VITS_DDSL_M(HD, TL)                           \
    do {                                              \
        if (c.channels <= 192) VITS_DDSL(HD, TL, 6);  \
        else VITS_DDSL(HD, TL, 8);                    \
    } while (0)
    if (head == DDS_HEAD_NONE && tail == DDS_TAIL_NONE) VITS_DDSL_M(DDS_HEAD_NONE, DDS_TAIL_NONE);
    else if (head == DDS_HEAD_NONE) VITS_DDSL_M(DDS_HEAD_NONE, DDS_TAIL_PROJ);
    else if (head == DDS_HEAD_FROM1 && tail == DDS_TAIL_NONE) VITS_DDSL_M(DDS_HEAD_FROM1, DDS_TAIL_NONE);
    else if (head == DDS_HEAD_FROM1) VITS_DDSL_M(DDS_HEAD_FROM1, DDS_TAIL_PROJ);
    else if (tail == DDS_TAIL_NONE) VITS_DDSL_M(DDS_HEAD_CONV, DDS_TAIL_NONE);
    else VITS_DDSL_M(DDS_HEAD_CONV, DDS_TAIL_PROJ);
#undef VITS_DDSL_M
#undef VITS_DDSL
    return hipGetLastError();
}

}  // namespace vits
