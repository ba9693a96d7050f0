// misc_kernels.hip — the non-GEMM kernels of the path (all HBM/latency-bound; coalesced along time).
// Each kernel cites the reference lines it replaces (/root/reference/src/...).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>

#include "../../include/vits.h"
#include "../../include/vits_synth_noise.h"
#include "../../include/vits_exact_math.h"
#include "kernels.h"

namespace vits {

// ---------------------------------------------------------------------------------------------------------
// embedding gather * sqrt(hidden)   (vits.cpp:262-264: ggml_get_rows + ggml_scale)
// ---------------------------------------------------------------------------------------------------------
__global__ void embed_kernel(const int* ids, int id_stride, const int* lens, const float* table, int hidden, float scale, float* x, int64_t bs, int cs,
                             int tmax) {
    const int b = blockIdx.z, c = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tmax) return;
    const int len = lens ? lens[b] : tmax;
    float v = 0.f;
    if (t < len) v = table[(size_t)ids[(size_t)b * id_stride + t] * hidden + c] * scale;
    x[(int64_t)b * bs + (int64_t)c * cs + t] = v;
}

hipError_t launch_embed(const int* ids, int id_stride, const int* lens, const float* table, int hidden, float scale, TensorRef x, int batch, int tmax,
                        hipStream_t s) {
    dim3 grid((tmax + 63) / 64, hidden, batch);
    VITS_KLAUNCH(embed_kernel, grid, dim3(64), 0, s, ids, id_stride, lens, table, hidden, scale, x.p, x.bs, x.cs, tmax);
    return hipGetLastError();
}

__global__ void fill_kernel(float* p, size_t n, float v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}
hipError_t launch_fill(float* p, size_t n, float v, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 4096);
    VITS_KLAUNCH(fill_kernel, dim3(blocks), dim3(256), 0, s, p, n, v);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Relative-position self-attention core (vits.cpp:296-356; helpers :195-235).  SURVEY.md App. F1 closed form:
//   s_ij = q_i.k_j + [|j-i|<=w] q_i.Ek[j-i+w];  p = softmax_j;  o_i = sum_j p_ij v_j + sum_{|j-i|<=w} p_ij Ev[j-i+w]
// The reference materialises dense (2T-1)-row relative tables and pads/reshapes them (pure data movement);
// here the 2w+1 relative logits are 9 extra dot products per query.
// One block = (utterance, head, 16 queries); scores for the 16 queries x T keys live in LDS.
// ---------------------------------------------------------------------------------------------------------
constexpr int ATT_Q = 16;
// Threads per block: 1024 (16 waves) on small grids — at batch 1 a launch has ~34 blocks and every phase is a chain of LDS / memory
// latencies that only more waves hide — and 256 on large ones (16-wave blocks pack worse: 73 -> 107 us per launch at batch 64).
// Both give the same bits: scores and outputs are per-element sums in a fixed order, and the softmax always runs on 16 lanes per query.

// ---------------------------------------------------------------------------------------------------------
// EMULATED ggml lookup tables (Q8; kernels.h GgmlTables): when a table pointer is given, GELU / the soft-max exponential go through the
// 65536-entry fp16 table indexed by the fp16 bits of the argument, as upstream ggml's ggml_vec_gelu_f32 / ggml_compute_forward_soft_max_f32
// do (the tables are built on the HOST with the C library's tanhf / expf, like ggml_init). Null pointers = the default arithmetic.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float ggml_table_lookup(const uint16_t* tab, float x) {
    const uint16_t i = __builtin_bit_cast(uint16_t, (_Float16)x);  // GGML_FP32_TO_FP16: round to nearest even
    return (float)__builtin_bit_cast(_Float16, tab[i]);
}
// Which of the block's 16 queries the 16-lane group t16 = tid / 16 normalises. ds_read_b32 / ds_write_b32 are served in groups of 32 lanes
// on banks (address / 4) mod 32, and the score rows are lp = 4 (mod 64) floats apart: two ADJACENT queries in one group of 32 lanes put
// their 16 columns on overlapping bank ranges (2-way conflicts on three passes over every row: the LDS_BANK_CONFLICT / LDS_IDX_ACTIVE = 0.37
// of the round-3 PMC pass). Queries q and q + 4 sit 16 banks apart: pair those. Each query's own 16-lane sums are untouched (same bits).
__device__ __forceinline__ int softmax_query_of(int t16) { return ((t16 & 1) << 2) | ((t16 >> 1) & 3) | (t16 & 8); }

// soft-max of one score row shared by 16 lanes (lane l16 owns columns l16, l16 + 16, ...), in place. exp_tab == nullptr: expf, fp32 sum,
// multiply by 1 / sum. exp_tab: ggml's table, the sum in double (terms are fp16 values <= 1: exact in any order), multiply by (float)(1 / sum).
__device__ __forceinline__ void softmax_row16(float* row, int len, int l16, const uint16_t* exp_tab) {
    float mx = -INFINITY;
    for (int j = l16; j < len; j += 16) mx = fmaxf(mx, row[j]);
    for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 16));
    if (exp_tab) {
        double sum = 0.0;
        for (int j = l16; j < len; j += 16) {
            const float e = ggml_table_lookup(exp_tab, row[j] - mx);
            row[j] = e;
            sum += (double)e;
        }
        for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 16);
        const float inv = (float)(1.0 / sum);
        for (int j = l16; j < len; j += 16) row[j] *= inv;
        return;
    }
    float sum = 0.f;
    for (int j = l16; j < len; j += 16) {
        const float e = expf(row[j] - mx);
        row[j] = e;
        sum += e;
    }
    for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 16);
    const float inv = 1.0f / sum;
    for (int j = l16; j < len; j += 16) row[j] *= inv;
}

template <int ATT_THREADS>
__global__ __launch_bounds__(ATT_THREADS) void rel_attention_kernel(const float* q, int64_t q_bs, int q_cs, const float* k, int64_t k_bs, int k_cs, const float* v,
                                                            int64_t v_bs, int v_cs, const float* rel_k, const float* rel_v, float* out, int64_t o_bs,
                                                            int o_cs, const int* lens, int head_dim, int tmax, int window, float q_scale, int vshift, const uint16_t* exp_tab) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int ATT_PASSES = 512 / ATT_THREADS < 1 ? 1 : 512 / ATT_THREADS;  // (d, 4-query group) items per thread: head_dim <= 128 (256 threads) / 256
    const int b = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * ATT_Q;
    const int len = lens ? lens[b] : tmax;
    if (i0 >= len) return;
    const int hd = head_dim, nrel = 2 * window + 1;
    float* qs = sm;                      // [ATT_Q][hd]
    float* qe = qs + ATT_Q * hd;         // [ATT_Q][nrel]   q_i . Ek[r]
    float* sc = qe + ATT_Q * nrel;       // [ATT_Q][len_pad]
    const int lp = (len + 3) & ~3;
    const int tid = threadIdx.x;
    const float* qb = q + (int64_t)b * q_bs + (int64_t)h * hd * q_cs;
    const float* kb = k + (int64_t)b * k_bs + (int64_t)h * hd * k_cs;
    const float* vb = v + (int64_t)b * v_bs + (int64_t)h * hd * v_cs;
    for (int idx = tid; idx < ATT_Q * hd; idx += ATT_THREADS) {
        const int d = idx / ATT_Q, qi = idx % ATT_Q;
        const int i = i0 + qi;
        qs[qi * hd + d] = i < len ? qb[(int64_t)d * q_cs + i] * q_scale : 0.f;  // scaling: vits.cpp:296-297
    }
    __syncthreads();
    for (int idx = tid; idx < ATT_Q * nrel; idx += ATT_THREADS) {
        const int qi = idx / nrel, r = idx % nrel;
        float a = 0.f;
        for (int d = 0; d < hd; ++d) a += qs[qi * hd + d] * rel_k[r * hd + d];
        qe[idx] = a;
    }
    __syncthreads();
    if constexpr (ATT_THREADS == 256) {
        // large grids: thread -> key j (coalesced along time), 16 running dot products per thread (half the K loads of the split below)
        for (int j = tid; j < len; j += ATT_THREADS) {
            float a[ATT_Q];
#pragma unroll
            for (int qi = 0; qi < ATT_Q; ++qi) a[qi] = 0.f;
            for (int d = 0; d < hd; ++d) {
                const float kv = kb[(int64_t)d * k_cs + j];
#pragma unroll
                for (int qi = 0; qi < ATT_Q; ++qi) a[qi] += qs[qi * hd + d] * kv;
            }
#pragma unroll
            for (int qi = 0; qi < ATT_Q; ++qi) {
                const int r = j - (i0 + qi) + window;
                float s = a[qi];
                if (r >= 0 && r < nrel) s += qe[qi * nrel + r];
                sc[qi * lp + j] = s;
            }
        }
    } else {
        // scores: thread -> (key j, half of the 16 queries): lanes run along time (coalesced), 8 running dot products per thread. (One
        // thread per key left 255 of 256 threads idle in a second pass for T = 257 = 128 interspersed ids.)
        constexpr int QH = ATT_Q / 2;
        for (int idx = tid; idx < 2 * len; idx += ATT_THREADS) {
            const int half = idx >= len ? 1 : 0, j = idx - half * len;
            const float* qh = qs + half * QH * hd;
            float a[QH];
    #pragma unroll
            for (int qi = 0; qi < QH; ++qi) a[qi] = 0.f;
            for (int d = 0; d < hd; ++d) {
                const float kv = kb[(int64_t)d * k_cs + j];
    #pragma unroll
                for (int qi = 0; qi < QH; ++qi) a[qi] += qh[qi * hd + d] * kv;
            }
    #pragma unroll
            for (int qi = 0; qi < QH; ++qi) {
                const int qa = half * QH + qi;
                const int r = j - (i0 + qa) + window;
                float s = a[qi];
                if (r >= 0 && r < nrel) s += qe[qa * nrel + r];
                sc[qa * lp + j] = s;
            }
        }
    }
    __syncthreads();
    // softmax per query: 16 lanes per query (the first 256 threads)
    if (tid < 256) {
        const int qi = softmax_query_of(tid >> 4), l16 = tid & 15;
        softmax_row16(sc + qi * lp, len, l16, exp_tab);
    }
    __syncthreads();
    // o[qi][d] = sum_j p[qi][j] v[d][j] + windowed relative-value term; thread -> (d, 4-query group), up to ATT_PASSES items per thread.
    // V goes through LDS in 64-key chunks (rows of v are time-major: lanes that differ in d would each touch their own cache line
    // per key — 2/3 of this kernel's time at batch 1); the sums still run over j in ascending order.
    float* ob = out + (int64_t)b * o_bs + (int64_t)h * hd * o_cs;
    if constexpr (ATT_THREADS == 256) {
        // large grids: V straight from memory (rows of v are time-major, lanes differ in d: one cache line per lane and key — other
        // resident blocks hide it, and the LDS staging below costs two barriers per 64 keys: 73 vs 94 us per launch at batch 64)
        for (int idx = tid; idx < hd * (ATT_Q / 4); idx += ATT_THREADS) {
            const int d = idx % hd, qg = idx / hd;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            const float* p0 = sc + (qg * 4 + 0) * lp;
            const float* p1 = sc + (qg * 4 + 1) * lp;
            const float* p2 = sc + (qg * 4 + 2) * lp;
            const float* p3 = sc + (qg * 4 + 3) * lp;
            const float* vr = vb + (int64_t)d * v_cs;
            for (int j = 0; j < len; ++j) {
                const float vv = vr[j];
                a0 += p0[j] * vv;
                a1 += p1[j] * vv;
                a2 += p2[j] * vv;
                a3 += p3[j] * vv;
            }
            float acc[4] = {a0, a1, a2, a3};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int qi = qg * 4 + u, i = i0 + qi;
                if (i >= len) continue;
                float rsum = 0.f;
                for (int r = 0; r < nrel; ++r) {
                    const int j = i + r - window;
                    if (j >= 0 && j < len) rsum += sc[qi * lp + j] * rel_v[r * hd + d];
                }
                ob[(int64_t)d * o_cs + i] = acc[u] + rsum;
            }
        }
        return;
    }
    const int vc = 1 << vshift, vp = vc + 1;  // keys per chunk (64 unless a long utterance's scores leave less LDS); odd pitch: lanes differ in d
    float* vt = sc + ATT_Q * lp;       // [hd][vp]
    float acc[ATT_PASSES][4];
#pragma unroll
    for (int ps = 0; ps < ATT_PASSES; ++ps)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[ps][u] = 0.f;
    const int nitems = hd * (ATT_Q / 4);
    for (int j0 = 0; j0 < len; j0 += vc) {
        const int nj = len - j0 < vc ? len - j0 : vc;
        for (int idx = tid; idx < (hd << vshift); idx += ATT_THREADS) {
            const int d = idx >> vshift, jj = idx & (vc - 1);
            vt[d * vp + jj] = jj < nj ? vb[(int64_t)d * v_cs + j0 + jj] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < ATT_PASSES; ++ps) {
            const int idx = tid + ps * ATT_THREADS;
            if (idx < nitems) {
                const int d = idx % hd, qg = idx / hd;
                const float* p0 = sc + (qg * 4 + 0) * lp + j0;
                const float* p1 = sc + (qg * 4 + 1) * lp + j0;
                const float* p2 = sc + (qg * 4 + 2) * lp + j0;
                const float* p3 = sc + (qg * 4 + 3) * lp + j0;
                const float* vr = vt + d * vp;
                float a0 = acc[ps][0], a1 = acc[ps][1], a2 = acc[ps][2], a3 = acc[ps][3];
                for (int j = 0; j < nj; ++j) {
                    const float vv = vr[j];
                    a0 += p0[j] * vv;
                    a1 += p1[j] * vv;
                    a2 += p2[j] * vv;
                    a3 += p3[j] * vv;
                }
                acc[ps][0] = a0;
                acc[ps][1] = a1;
                acc[ps][2] = a2;
                acc[ps][3] = a3;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int ps = 0; ps < ATT_PASSES; ++ps) {
        const int idx = tid + ps * ATT_THREADS;
        if (idx >= nitems) continue;
        const int d = idx % hd, qg = idx / hd;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int qi = qg * 4 + u, i = i0 + qi;
            if (i >= len) continue;
            float rsum = 0.f;
            for (int r = 0; r < nrel; ++r) {
                const int j = i + r - window;
                if (j >= 0 && j < len) rsum += sc[qi * lp + j] * rel_v[r * hd + d];
            }
            ob[(int64_t)d * o_cs + i] = acc[ps][u] + rsum;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// The same attention core on the matrix cores (head_dim a multiple of 16): scores S = Q K^T and O = P V as 16 x 16 x 4 fp32 MFMAs
// (v_mfma_f32_16x16x4_f32: the 16 queries of a block are M), softmax and the 2w+1 relative terms as above. The VALU kernel does
// one LDS read per FMA and, for 1024-id inputs, was the longest single entry of a 16-bit step.
//   operand layouts (lane l): A[m = l % 16][k = l / 16], B[k = l / 16][n = l % 16], D[m = 4 * (l / 16) + r][n = l % 16], r = 0..3
// Block = (utterance, head, 16 queries), NW = 4 or 8 waves (chosen by the launch, see launch_rel_attention). Scores: wave w owns key
// tiles w, w + NW, ... and fetches the K operands of its next TWO tiles while it multiplies the current one (a tile is 24 x 32 cycles of
// MFMA work, a load round trip several times that). P V: wave w owns d tiles w, w + NW and walks the keys in groups of 16: MFMA k-step 4 g + i of group g takes
// keys 16 g + 4 (l / 16) + i, so that a lane's four k-steps are ONE 16-byte load of V (its row, 4 consecutive keys) and ONE
// ds_read_b128 of P, four groups in flight. (Round 2: one K tile of look-ahead, a dword load of V per k-step consumed right behind its
// issue, V staged through LDS for short inputs; per-block stamps at 1024 ids — tools/att_micro.hip — 28 us of scores and 61 us of P V
// for 5 + 7 us of MFMA work, now 21 + 11.)
// Every sum has ONE order for every grid, so results do not depend on the batch.
// LDS: Q^T [hd][16] | q.Ek [16][nrel] | scores [16][lp], lp = 4 mod 64 (conflict-free A reads of P).
// ---------------------------------------------------------------------------------------------------------
typedef float att_float4v __attribute__((ext_vector_type(4)));

__host__ __device__ inline int att_lp(int len) { return (len + 63) / 64 * 64 + 4; }  // >= len + 4, = 4 (mod 64)

#ifdef VITS_PHASE_TIMING  // developer instrumentation (tools/att_micro.hip): per-block phase stamps, 100 MHz clock
__device__ unsigned long long vits_att_phase[8 * 65536];
#define ATT_STAMP(k)                                                                                             \
    do {                                                                                                         \
        if (threadIdx.x == 0) {                                                                                  \
            const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                 \
            if (lin < 65536) vits_att_phase[8 * lin + (k)] = __builtin_amdgcn_s_memrealtime();                   \
        }                                                                                                        \
    } while (0)
#else
#define ATT_STAMP(k)
#endif

// MAXS: k-steps of head_dim the register arrays are sized for (24: head_dim <= 96, the MMS-TTS architecture; 32: <= 128). SHORT: sequences of
// at most a few key tiles per wave (the 128-token utterances of a batch): ONE key tile of look-ahead instead of two and at most 128 VGPRs, so
// that four blocks of four waves share a CU instead of three — 1024 blocks are then one round of the chip instead of 1.33.
// LAT (latency-bound launches: at most 128 blocks, round 6): every operand that does not depend on an earlier phase is REQUESTED at the top of the phase
// before — the wave's first K tile and the relative-key embeddings beside the Q tile, the first four V groups and the relative-value embeddings before
// the softmax — so that a block pays three memory round trips instead of six (per-block stamps at 128 tokens: 13.9 -> see launch_rel_attention). The
// MFMA sequences, and with them every sum, are those of the other variants.
template <int NW, int MAXS, bool SHORT, bool LAT = false>
__global__ __launch_bounds__(64 * NW, LAT ? 2 : (SHORT ? 4 : 3)) void rel_attention_mfma_kernel(const float* q, int64_t q_bs, int q_cs, const float* k, int64_t k_bs, int k_cs, const float* v,
                                                                     int64_t v_bs, int v_cs, const float* rel_k, const float* rel_v, float* out, int64_t o_bs,
                                                                     int o_cs, const int* lens, int head_dim, int tmax, int window, float q_scale, int v16, const uint16_t* exp_tab) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int NT = 64 * NW;
    // XCD-aware order: the dispatcher deals workgroups to the 8 XCDs round-robin by linear id and every XCD has its own L2. All query
    // tiles of one (utterance, head) read the same K and V: every XCD gets a contiguous range of the order (query tile fastest, then
    // head, then utterance), so that its L2 holds the K / V of a few (utterance, head) pairs instead of all of them (1024 ids: -10 %).
    const unsigned gx = gridDim.x, gy = gridDim.y, nblk = gx * gy * gridDim.z;
    const unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned xc = lin & 7, slot = lin >> 3, qd = nblk >> 3, rm = nblk & 7;
    const unsigned lg = xc * qd + (xc < rm ? xc : rm) + slot;
    const int b = lg / (gx * gy), h = (lg / gx) % gy, i0 = (lg % gx) * ATT_Q;
    const int len = lens ? lens[b] : tmax;
    if (i0 >= len) return;
    ATT_STAMP(0);
    const int hd = head_dim, nrel = 2 * window + 1;
    const int lp = att_lp(len);
    float* qt = sm;                 // [hd][16]  (scaled)
    float* qe = qt + hd * ATT_Q;    // [16][nrel]
    float* sc = sm + ((hd * ATT_Q + ATT_Q * nrel + 3) & ~3);  // [16][lp], 16-byte aligned rows (ds_read_b128 of P)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int ln = lane & 15, lk = lane >> 4;
    const float* qb = q + (int64_t)b * q_bs + (int64_t)h * hd * q_cs;
    const float* kb = k + (int64_t)b * k_bs + (int64_t)h * hd * k_cs;
    const float* vb = v + (int64_t)b * v_bs + (int64_t)h * hd * v_cs;
    const int nsteps = hd >> 2;
    const int ntiles = (len + 15) >> 4;
    float k0[MAXS], k1[MAXS], k2[MAXS];  // K operands of this wave's current / next key tiles (scores phase)
    auto load_tile = [&](int n, float* dst) __attribute__((always_inline)) {
        const int nc = n < ntiles ? n : ntiles - 1;  // (past the last tile: a valid address, values unused)
        const int key = nc * 16 + ln;
        const float* kp = kb + (int64_t)lk * k_cs + (key < len ? key : len - 1);
#pragma unroll
        for (int s2 = 0; s2 < MAXS; ++s2) dst[s2] = s2 < nsteps ? kp[(int64_t)(4 * s2) * k_cs] : 0.f;
    };
    float bvp[MAXS];  // LAT: the relative-key operands of this wave's tile of relative positions
    if constexpr (LAT) {
        load_tile(wid, k0);
        const int rcol = wid * 16 + ln;
        const float* rp = rel_k + (int64_t)(rcol < nrel ? rcol : 0) * hd + lk;
#pragma unroll
        for (int s2 = 0; s2 < MAXS; ++s2) bvp[s2] = (wid * 16 < nrel && s2 < nsteps && rcol < nrel) ? rp[4 * s2] : 0.f;
    }
    {
        float tq[8];  // (head_dim <= 128: at most 8 elements per thread, all loads in flight)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = tid + u * NT;
            const int d = idx / ATT_Q, i = i0 + idx % ATT_Q;
            tq[u] = (idx < ATT_Q * hd && i < len) ? qb[(int64_t)d * q_cs + i] * q_scale : 0.f;  // scaling: vits.cpp:296-297
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (tid + u * NT < ATT_Q * hd) qt[tid + u * NT] = tq[u];  // (idx = d * 16 + qi is the Q^T index)
    }
    __syncthreads();
    // this lane's Q operands for every k-step (A[m = ln][k = 4s + lk]): the same for every key tile and for the relative-key product
    float qa[MAXS];
#pragma unroll
    for (int s2 = 0; s2 < MAXS; ++s2) qa[s2] = s2 < nsteps ? qt[(4 * s2 + lk) * ATT_Q + ln] : 0.f;
    // q_i . Ek[r] for the 2w+1 relative positions: one more 16 x 16 product per tile of 16 relative positions (one tile — wave 0 — for
    // windows up to 7, the MMS-TTS architecture has 4; wider windows take further tiles on the other waves)
    for (int ct = wid; ct * 16 < nrel; ct += NW) {
        const int rcol = ct * 16 + ln;
        att_float4v acc = {0.f, 0.f, 0.f, 0.f};
        const float* rp = rel_k + (int64_t)(rcol < nrel ? rcol : 0) * hd + lk;
#pragma unroll
        for (int s0 = 0; s0 < MAXS; s0 += 8) {
            float bv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (LAT && ct == wid) bv[u] = s0 + u < MAXS ? bvp[s0 + u < MAXS ? s0 + u : 0] : 0.f;
                else bv[u] = (s0 + u < nsteps && rcol < nrel) ? rp[4 * (s0 + u)] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (s0 + u < nsteps) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s0 + u], bv[u], acc, 0, 0, 0);
        }
        if (rcol < nrel) {
#pragma unroll
            for (int r = 0; r < 4; ++r) qe[(4 * lk + r) * nrel + rcol] = acc[r];
        }
    }
    if constexpr (!LAT) __syncthreads();  // (LAT: behind the first tile's MFMA chain, below)
    ATT_STAMP(1);
    // ---- scores: S[16 q][16 keys] per key tile, K = hd; the K operands of this wave's next two tiles are in flight ----
    {
        auto tile_chain = [&](const float* kop) __attribute__((always_inline)) -> att_float4v {
            att_float4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < MAXS; ++s2)
                if (s2 < nsteps) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s2], kop[s2], acc, 0, 0, 0);
            return acc;
        };
        auto tile_store = [&](int n, const att_float4v acc) __attribute__((always_inline)) {
            const int key = n * 16 + ln;
            if (key < len) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qi = 4 * lk + r;
                    const int rr = key - (i0 + qi) + window;
                    float sv = acc[r];
                    if (rr >= 0 && rr < nrel) sv += qe[qi * nrel + rr];
                    sc[qi * lp + key] = sv;
                }
            }
        };
        auto tile = [&](int n, const float* kop) __attribute__((always_inline)) { tile_store(n, tile_chain(kop)); };
        int n = wid;
        if constexpr (LAT) {
            // the first tile's MFMA chain runs BESIDE wave 0's relative-key product: the barrier that publishes q.Ek stands behind the chain, in front of the
            // first store (the other variants have it in front of the phase)
            if (n + NW < ntiles) load_tile(n + NW, k1);
            att_float4v a0 = {0.f, 0.f, 0.f, 0.f};
            if (n < ntiles) a0 = tile_chain(k0);
            __syncthreads();
            if (n < ntiles) tile_store(n, a0);
            n += NW;
            // (further tiles — sequences of more than 16 NW tokens on a small grid — one tile of look-ahead, k1 / k0 in turn)
            while (n < ntiles) {
                if (n + NW < ntiles) load_tile(n + NW, k0);
                __builtin_amdgcn_sched_barrier(0);
                tile(n, k1);
                n += NW;
                if (n >= ntiles) break;
                if (n + NW < ntiles) load_tile(n + NW, k1);
                __builtin_amdgcn_sched_barrier(0);
                tile(n, k0);
                n += NW;
            }
        } else
        if constexpr (SHORT) {
            if (n < ntiles) load_tile(n, k0);
            while (n < ntiles) {
                load_tile(n + NW, k1);
                __builtin_amdgcn_sched_barrier(0);
                tile(n, k0);
                n += NW;
                if (n >= ntiles) break;
                load_tile(n + NW, k0);
                __builtin_amdgcn_sched_barrier(0);
                tile(n, k1);
                n += NW;
            }
        } else {
        if (n < ntiles) {
            load_tile(n, k0);
            load_tile(n + NW, k1);
        }
        while (n < ntiles) {
            load_tile(n + 2 * NW, k2);
            __builtin_amdgcn_sched_barrier(0);
            tile(n, k0);
            n += NW;
            if (n >= ntiles) break;
            load_tile(n + 2 * NW, k0);
            __builtin_amdgcn_sched_barrier(0);
            tile(n, k1);
            n += NW;
            if (n >= ntiles) break;
            load_tile(n + 2 * NW, k1);
            __builtin_amdgcn_sched_barrier(0);
            tile(n, k2);
            n += NW;
        }
        }
    }
    // LAT: the first four V groups of this wave's first d tile and its relative-value embeddings, requested before the softmax
    const int ndt = hd >> 4;
    const int ng = (len + 15) >> 4;
    const int klast4 = (len - 1) & ~3;  // the last 16-byte piece of a row that starts inside the sequence (rows are padded to x4)
    auto load_v = [&](int g, const float* vrow, att_float4v& vv) __attribute__((always_inline)) {
        const int gc = g < ng ? g : ng - 1;
        const int key0 = 16 * gc + 4 * lk;
        if (v16) {
            const att_float4v t = *reinterpret_cast<const att_float4v*>(vrow + (key0 < klast4 ? key0 : klast4));
            vv[0] = key0 < len ? t[0] : 0.f;
            vv[1] = key0 + 1 < len ? t[1] : 0.f;
            vv[2] = key0 + 2 < len ? t[2] : 0.f;
            vv[3] = key0 + 3 < len ? t[3] : 0.f;
        } else {  // rows not 16-byte aligned (never in the engine; the operator entry point takes any stride): same values, four loads
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = vrow[key0 + e < len ? key0 + e : len - 1];
                vv[e] = key0 + e < len ? t : 0.f;
            }
        }
    };
    att_float4v vpre[8];  // (eight groups = 128 keys: the whole row of a 128-token utterance)
    float evp[4];
    if constexpr (LAT) {
        if (wid < ndt) {
            const float* vrow = vb + (int64_t)(wid * 16 + ln) * v_cs;
#pragma unroll
            for (int i = 0; i < 8; ++i) load_v(i, vrow, vpre[i]);
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                const int r = 4 * s2 + lk;
                evp[s2] = r < nrel ? rel_v[(int64_t)r * hd + wid * 16 + ln] : 0.f;
            }
        }
    }
    __syncthreads();
    ATT_STAMP(2);
    // ---- softmax per query: 16 lanes per query; the padding columns of P are zeroed (the MFMA k-steps run over whole groups) ----
    if (tid < 16 * ATT_Q) {
        const int qi = softmax_query_of(tid >> 4), l16 = tid & 15;
        softmax_row16(sc + qi * lp, len, l16, exp_tab);
        for (int j = len + l16; j < lp; j += 16) sc[qi * lp + j] = 0.f;
    }
    __syncthreads();
    ATT_STAMP(3);
    // ---- O[16 q][hd] = P V: d tile dt, K = keys in groups of 16 (see the head comment), four groups in flight ----
    att_float4v oacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};  // d tiles wid and wid + NW (NW >= 4, head_dim <= 128)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int dt = wid + NW * u;
        if (dt < ndt) {
            const float* prow = sc + ln * lp + 4 * lk;
            const float* vrow = vb + (int64_t)(dt * 16 + ln) * v_cs;
            att_float4v pr[4], vr[4];
            auto load_group = [&](int g, att_float4v& pa, att_float4v& vv) __attribute__((always_inline)) {
                const int gc = g < ng ? g : ng - 1;
                pa = *reinterpret_cast<const att_float4v*>(prow + 16 * gc);
                load_v(g, vrow, vv);
            };
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (LAT && u == 0) {  // (V requested before the softmax; P exists only now)
                    pr[i] = *reinterpret_cast<const att_float4v*>(prow + 16 * (i < ng ? i : ng - 1));
                    vr[i] = vpre[i];
                } else {
                    load_group(i, pr[i], vr[i]);
                }
            }
            att_float4v a4 = oacc[u];
            for (int g = 0; g < ng; g += 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const att_float4v pa = pr[i], vv = vr[i];
                    if (LAT && u == 0 && g == 0) {  // (groups 4-7: V came with the first four)
                        pr[i] = *reinterpret_cast<const att_float4v*>(prow + 16 * (4 + i < ng ? 4 + i : ng - 1));
                        vr[i] = vpre[4 + i];
                    } else {
                        load_group(g + 4 + i, pr[i], vr[i]);
                    }
                    if (g + i < ng) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[e], vv[e], a4, 0, 0, 0);
                    }
                }
            }
            oacc[u] = a4;
        }
    }
    ATT_STAMP(4);
    // ---- + windowed relative-value term: O += Pwin[16 q][r] . Ev[r][d] with Pwin[q][r] = P[q][i_q + r - w] (zero outside the sequence
    // and for r >= 2w+1): (2w+1+3)/4 more k-steps per d tile ----
    {
        const int rsteps = (nrel + 3) >> 2;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int dt = wid + NW * u;
            if (dt < ndt) {
                att_float4v a4 = oacc[u];
                for (int s = 0; s < rsteps; ++s) {
                    const int r = 4 * s + lk;
                    const int j = i0 + ln + r - window;
                    const float pw = (r < nrel && j >= 0 && j < len && i0 + ln < len) ? sc[ln * lp + j] : 0.f;
                    const float ev = (LAT && u == 0 && s < 4) ? evp[s < 4 ? s : 0] : (r < nrel ? rel_v[(int64_t)r * hd + dt * 16 + ln] : 0.f);
                    a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(pw, ev, a4, 0, 0, 0);
                }
                oacc[u] = a4;
            }
        }
    }
    // ---- store: lane holds queries 4*lk .. 4*lk+3 of channel dt*16 + ln ----
    float* ob = out + (int64_t)b * o_bs + (int64_t)h * hd * o_cs;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int dt = wid + NW * u;
        if (dt >= ndt) continue;
        const int d = dt * 16 + ln;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + 4 * lk + r;
            if (i < len) ob[(int64_t)d * o_cs + i] = oacc[u][r];
        }
    }
    ATT_STAMP(5);
}

hipError_t launch_rel_attention(TensorRef q, TensorRef k, TensorRef v, const float* rel_k, const float* rel_v, TensorRef out, const int* lens, int batch,
                                int heads, int head_dim, int tmax, int window, float q_scale, hipStream_t s, GgmlTables tabs) {
    const bool valu_only = kernel_knobs().att_valu;
    // matrix-core version. The number of waves (= how the key tiles and the d tiles are dealt out) does not change a single sum, so it may
    // depend on the launch: four waves while two or more blocks fit the LDS of a CU (up to ~1200 tokens: 1024 ids 0.178 ms against 0.222
    // with six waves), eight once the scores of a block leave room for one block only (2049 tokens: 0.87 against 1.29 ms with four)
    const bool v_aligned = (v.cs & 3) == 0 && (v.bs & 3) == 0 && (reinterpret_cast<uintptr_t>(v.p) & 15) == 0;
    if (!valu_only && (head_dim & 15) == 0 && head_dim <= 128) {
        const size_t ldsm = sizeof(float) * ((((size_t)ATT_Q * head_dim + ATT_Q * (2 * window + 1) + 3) & ~(size_t)3) + (size_t)ATT_Q * att_lp(tmax));
        if (ldsm <= 160 * 1024) {
            dim3 gridm((tmax + ATT_Q - 1) / ATT_Q, heads, batch);
            const int nw_env = kernel_knobs().att_nw;
            int nw = 2 * ldsm > 160 * 1024 ? 8 : 4;
            // latency-bound launches (at most 128 blocks: up to eight 128-token utterances): eight waves deal the key tiles and the d tiles out one per wave
            // (per-block stamps at 128 tokens, tools/att_micro.hip: P V 5.0 -> 3.0 us, block life 15.2 -> 13.9; batch 1 / 2 / 4 / 8: -1 ... -3 % per call, round 6)
            if ((int64_t)gridm.x * gridm.y * gridm.z <= 128) nw = 8;
            if (nw_env == 4 || nw_env == 8) nw = nw_env;
#define VITS_ATTM_LAUNCH(NW, MS, SH)                                                                                                                   \
    do {                                                                                                                                         \
        if (ldsm > 64 * 1024) {                                                                                                                  \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rel_attention_mfma_kernel<NW, MS, SH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsm); \
            if (e != hipSuccess) return e;                                                                                                       \
        }                                                                                                                                        \
        VITS_KLAUNCH((rel_attention_mfma_kernel<NW, MS, SH>), gridm, dim3(64 * NW), ldsm, s, q.p, q.bs, q.cs, k.p, k.bs, k.cs, v.p, v.bs, v.cs, rel_k, rel_v, out.p, \
                           out.bs, out.cs, lens, head_dim, tmax, window, q_scale, v_aligned ? 1 : 0, tabs.exp);                                            \
    } while (0)
#define VITS_ATTM_LAUNCH_LAT()                                                                                                                          \
    do {                                                                                                                                         \
        if (ldsm > 64 * 1024) {                                                                                                                  \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rel_attention_mfma_kernel<8, 24, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsm); \
            if (e != hipSuccess) return e;                                                                                                       \
        }                                                                                                                                        \
        VITS_KLAUNCH((rel_attention_mfma_kernel<8, 24, false, true>), gridm, dim3(512), ldsm, s, q.p, q.bs, q.cs, k.p, k.bs, k.cs, v.p, v.bs, v.cs, rel_k, rel_v, out.p, \
                           out.bs, out.cs, lens, head_dim, tmax, window, q_scale, v_aligned ? 1 : 0, tabs.exp);                                            \
    } while (0)
            const int short_max = kernel_knobs().att_short;  // tokens; 0 disables the short variant
            // (the long variants keep their arrays at 32 k-steps: sized for 24 the four-wave kernel measured 0.22 against 0.18 ms at 1024 tokens)
            const bool lat = !kernel_knobs().no_att_lat && nw == 8 && head_dim <= 96 && (int64_t)gridm.x * gridm.y * gridm.z <= 128;
            if (lat) VITS_ATTM_LAUNCH_LAT();
            else if (nw == 4 && head_dim <= 96 && tmax <= short_max) VITS_ATTM_LAUNCH(4, 24, true);
            else if (nw == 4) VITS_ATTM_LAUNCH(4, 32, false);
            else VITS_ATTM_LAUNCH(8, 32, false);
#undef VITS_ATTM_LAUNCH
#undef VITS_ATTM_LAUNCH_LAT
            return hipGetLastError();
        }
    }
    const int lp = (tmax + 3) & ~3;
    size_t lds = 0;
    int vshift = 6;
    for (; vshift >= 3; --vshift) {
        lds = sizeof(float) * ((size_t)ATT_Q * head_dim + ATT_Q * (2 * window + 1) + (size_t)ATT_Q * lp + (size_t)head_dim * ((1 << vshift) + 1));
        if (lds <= 150 * 1024) break;
    }
    if (lds > 150 * 1024 || head_dim * (ATT_Q / 4) > 512) return hipErrorInvalidValue;
    dim3 grid((tmax + ATT_Q - 1) / ATT_Q, heads, batch);
    const bool small_grid = (int64_t)grid.x * grid.y * grid.z <= 512;
#define VITS_ATT_LAUNCH(T)                                                                                                                  \
    do {                                                                                                                                    \
        if (lds > 64 * 1024) {                                                                                                              \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rel_attention_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return e;                                                                                                  \
        }                                                                                                                                   \
        VITS_KLAUNCH(rel_attention_kernel<T>, grid, dim3(T), lds, s, q.p, q.bs, q.cs, k.p, k.bs, k.cs, v.p, v.bs, v.cs, rel_k, rel_v, out.p, out.bs, \
                           out.cs, lens, head_dim, tmax, window, q_scale, vshift, tabs.exp);                                                         \
    } while (0)
    if (small_grid) VITS_ATT_LAUNCH(1024);
    else VITS_ATT_LAUNCH(256);
#undef VITS_ATT_LAUNCH
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// y = LayerNorm_channels(x + res) [optionally gelu, optionally add_to += y]
//   encoder: vits.cpp:365-372, 412-418 (ggml_add + ggml_norm + mul + add = 4 nodes)
//   DDS:     vits.cpp:679-688 (permute, cont, norm, permute, cont, gelu, add)
// Block = 64 time steps x all channels; the tile is held in LDS so x is read from HBM once.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// vits.cpp:673,687 ggml_gelu: erf-GELU (HF) by default; with a table, ggml's tanh-GELU through its fp16 lookup table (Q8, emulated)
__device__ __forceinline__ float gelu_op(float x, const uint16_t* gelu_tab) { return gelu_tab ? ggml_table_lookup(gelu_tab, x) : gelu_erf(x); }

// channel groups per block of the two LayerNorm kernels: 16 x 64 threads, every thread walks channels/16 rows. (4 groups
// made each thread chain 48 dependent loads: 26-58 us per launch at batch 1, where these launches have 2 blocks.) The
// partial sums are combined in a fixed order, so results do not depend on the batch.
constexpr int LN_GROUPS = 16;

template <int TW>
__global__ __launch_bounds__(TW * LN_GROUPS) void add_layer_norm_kernel(const float* x, int64_t x_bs, int x_cs, const float* res, int64_t r_bs, int r_cs,
                                                             const float* gamma, const float* beta, float* y, int64_t y_bs, int y_cs, float* addto,
                                                             int64_t a_bs, int a_cs, const int* lens, int channels, int tmax, float eps, int post_gelu, const uint16_t* gelu_tab) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* tile = sm;                    // [channels][TW]
    float* red = sm + channels * TW;     // [LN_GROUPS][TW] x2
    const int b = blockIdx.y, t0 = blockIdx.x * TW;
    const int len = lens ? lens[b] : tmax;
    if (t0 >= len) return;
    const int tl = threadIdx.x % TW, g = threadIdx.x / TW;
    const int t = t0 + tl;
    const bool ok = t < len;
    float s = 0.f;
    // (twelve channels per step, their loads issued before the first LDS store: a load -> store loop exposes one memory latency per
    // channel, and at batch 1 a launch has two blocks. Same values, same order of s.)
    constexpr int UB = 12;
    for (int c0 = g; c0 < channels; c0 += UB * LN_GROUPS) {
        float xv[UB], rv[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int c = c0 + u * LN_GROUPS;
            const bool live = ok && c < channels;
            xv[u] = live ? x[(int64_t)b * x_bs + (int64_t)c * x_cs + t] : 0.f;
            rv[u] = (live && res) ? res[(int64_t)b * r_bs + (int64_t)c * r_cs + t] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int c = c0 + u * LN_GROUPS;
            if (c >= channels) continue;
            float v = 0.f;
            if (ok) {
                v = xv[u];
                if (res) v += rv[u];
            }
            tile[c * TW + tl] = v;
            s += v;
        }
    }
    red[g * TW + tl] = s;
    __syncthreads();
    float msum = 0.f;
#pragma unroll
    for (int q = 0; q < LN_GROUPS; ++q) msum += red[q * TW + tl];
    const float mean = msum / (float)channels;
    float vs = 0.f;
    for (int c = g; c < channels; c += LN_GROUPS) {
        const float d = tile[c * TW + tl] - mean;
        vs += d * d;
    }
    red[(LN_GROUPS + g) * TW + tl] = vs;
    __syncthreads();
    float vsum = 0.f;
#pragma unroll
    for (int q = 0; q < LN_GROUPS; ++q) vsum += red[(LN_GROUPS + q) * TW + tl];
    const float var = vsum / (float)channels;
    const float inv = 1.0f / sqrtf(var + eps);
    if (!ok) return;
    for (int c = g; c < channels; c += LN_GROUPS) {
        float v = (tile[c * TW + tl] - mean) * inv * gamma[c] + beta[c];
        if (post_gelu) v = gelu_op(v, gelu_tab);
        if (addto) {
            float* a = addto + (int64_t)b * a_bs + (int64_t)c * a_cs + t;
            *a = *a + v;
        } else
            y[(int64_t)b * y_bs + (int64_t)c * y_cs + t] = v;
    }
}

hipError_t launch_add_layer_norm(TensorRef x, TensorRef res, const float* gamma, const float* beta, TensorRef y, const int* lens, int batch, int channels,
                                 int tmax, float eps, int post_gelu, TensorRef add_to, hipStream_t s, GgmlTables tabs) {
    // tile width (time steps per block): 32 = eight waves and channels x 128 B of LDS per block (29 KB at 192 channels); VITS_LN_TW=64 = the
    // sixteen-wave, 57 KB blocks of rounds 1-3. Every token's sums are the same either way (its channels are summed by the same sixteen channel
    // groups in the same order). The small block matters when this kernel shares the chip with another batch's vocoder (vits_model_submit_batch):
    // a 57 KB block only finds room in the tail of a vocoder kernel — stage one of a pipelined f16 batch took 10.2 ms of wall with it, 8.8 with
    // the small one (alone: 1.91 -> 1.87 ms).
    const int tw_env = kernel_knobs().ln_tw;
    const int tw = tw_env == 64 ? 64 : 32;
    const size_t lds = sizeof(float) * ((size_t)channels * tw + 2 * tw * LN_GROUPS);
    if (lds > 150 * 1024) return hipErrorInvalidValue;
    dim3 grid((tmax + tw - 1) / tw, batch);
#define VITS_LN_LAUNCH(TW)                                                                                                                           \
    do {                                                                                                                                             \
        if (lds > 64 * 1024) {                                                                                                                       \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(add_layer_norm_kernel<TW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return e;                                                                                                           \
        }                                                                                                                                            \
        VITS_KLAUNCH(add_layer_norm_kernel<TW>, grid, dim3(TW * LN_GROUPS), lds, s, x.p, x.bs, x.cs, res.p, res.bs, res.cs, gamma, beta, y.p, y.bs, y.cs, add_to.p, \
                     add_to.bs, add_to.cs, lens, channels, tmax, eps, post_gelu, tabs.gelu);                                                       \
    } while (0)
    if (tw == 32) VITS_LN_LAUNCH(32);
    else VITS_LN_LAUNCH(64);
#undef VITS_LN_LAUNCH
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// DDS first half: y = gelu(LN_channels(depthwise_conv_k(x [+ g], dilation) + bias))
//   vits.cpp:651-673 with depthwise_conv_with_bias :144-169 (the reference runs `channels` separate 1-channel
//   convolutions = ~2,300 graph nodes per predictor, Q16). If g != null, x <- x + g is also written back
//   (vits.cpp:651-653 happens once before the layer loop; the engine passes g only for layer 0).
// Block = 64 time steps x all channels, input tile with halo in LDS.
// ---------------------------------------------------------------------------------------------------------
// Q7 (custom-ops.h:684-690): in the 16-bit arithmetic modes every conv operand is rounded (nearest even) to fp16 / bf16; the
// products are then exact in fp32. arith: 0 fp32, 1 bf16, 2 fp16 (VITS_ARITH_*).
__device__ __forceinline__ float round_arith(float v, int arith) {
    if (arith == 2) return (float)(_Float16)v;
    if (arith == 1) return (float)(__bf16)v;
    return v;
}

__global__ __launch_bounds__(64 * LN_GROUPS) void dds_depthwise_kernel(float* x, int64_t x_bs, int x_cs, const float* g, int64_t g_bs, int g_cs, const float* w,
                                                            const float* bias, const float* gamma, const float* beta, float* y, int64_t y_bs, int y_cs,
                                                            const int* lens, int channels, int tmax, int k, int dil, float eps, int arith, const uint16_t* gelu_tab) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int pad = (k * dil - dil) / 2;  // vits.cpp:660
    const int xw = 64 + 2 * pad;
    float* xt = sm;                       // [channels][xw]
    float* ht = xt + channels * xw;       // [channels][64]
    float* red = ht + channels * 64;      // [2][LN_GROUPS][64]
    const int b = blockIdx.y, t0 = blockIdx.x * 64;
    const int len = lens ? lens[b] : tmax;
    if (t0 >= len) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int c = wid; c < channels; c += LN_GROUPS) {
        for (int i = lane; i < xw; i += 64) {
            const int t = t0 - pad + i;
            float v = 0.f;
            if (t >= 0 && t < len) {
                v = x[(int64_t)b * x_bs + (int64_t)c * x_cs + t];
                if (g) {
                    v += g[(int64_t)b * g_bs + (int64_t)c * g_cs + t];
                }
            }
            xt[c * xw + i] = v;
        }
    }
    __syncthreads();
    if (g) {  // write back x + g for the centre columns (this block owns them)
        for (int c = wid; c < channels; c += LN_GROUPS) {
            const int t = t0 + lane;
            if (t < len) x[(int64_t)b * x_bs + (int64_t)c * x_cs + t] = xt[c * xw + pad + lane];
        }
    }
    const int tl = lane, gq = wid;
    float s = 0.f;
    for (int c = gq; c < channels; c += LN_GROUPS) {
        float a = bias[c];
        for (int j = 0; j < k; ++j) a += round_arith(w[c * k + j], arith) * round_arith(xt[c * xw + tl + j * dil], arith);
        ht[c * 64 + tl] = a;
        s += a;
    }
    red[gq * 64 + tl] = s;
    __syncthreads();
    float msum = 0.f;
#pragma unroll
    for (int q = 0; q < LN_GROUPS; ++q) msum += red[q * 64 + tl];
    const float mean = msum / (float)channels;
    float vs = 0.f;
    for (int c = gq; c < channels; c += LN_GROUPS) {
        const float d = ht[c * 64 + tl] - mean;
        vs += d * d;
    }
    red[(LN_GROUPS + gq) * 64 + tl] = vs;
    __syncthreads();
    float vsum = 0.f;
#pragma unroll
    for (int q = 0; q < LN_GROUPS; ++q) vsum += red[(LN_GROUPS + q) * 64 + tl];
    const float var = vsum / (float)channels;
    const float inv = 1.0f / sqrtf(var + eps);
    const int t = t0 + tl;
    if (t >= len) return;
    for (int c = gq; c < channels; c += LN_GROUPS) y[(int64_t)b * y_bs + (int64_t)c * y_cs + t] = gelu_op((ht[c * 64 + tl] - mean) * inv * gamma[c] + beta[c], gelu_tab);
}

hipError_t launch_dds_depthwise(TensorRef x, TensorRef g, const float* w, const float* bias, const float* gamma, const float* beta, TensorRef y,
                                const int* lens, int batch, int channels, int tmax, int k, int dil, float eps, hipStream_t s, int arith, GgmlTables tabs) {
    const int pad = (k * dil - dil) / 2;
    const size_t lds = sizeof(float) * ((size_t)channels * (64 + 2 * pad) + (size_t)channels * 64 + 2 * 64 * LN_GROUPS);
    if (lds > 150 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dds_depthwise_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    dim3 grid((tmax + 63) / 64, batch);
    VITS_KLAUNCH(dds_depthwise_kernel, grid, dim3(64 * LN_GROUPS), lds, s, x.p, x.bs, x.cs, g.p, g.bs, g.cs, w, bias, gamma, beta, y.p, y.bs, y.cs, lens, channels, tmax,
                       k, dil, eps, arith, tabs.gelu);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// One whole DDS layer as ONE kernel (vits.cpp:655-691):
//     x_out = x + gelu(LN2(pointwise(gelu(LN1(depthwise_k,dil(x))))))
// i.e. dds_depthwise_kernel + the 1x1 conv (conv_mfma.hip / conv16.hip) + add_layer_norm_kernel(gelu, add_to) of the
// three-launch path, with the two intermediates kept in LDS. Block = 32 time steps x all channels, 16 channel groups of 32
// threads. Every sum is taken in the order of the three-launch path (the LayerNorm partial sums per channel group and
// their fixed-order combination; the MFMA chain over the packed weight fragments, chunk by chunk), so the result is
// bit-identical to it — tests/test_gpu_round2.py compares the two. At batch 1 the three launches are 45-50 us of
// latency per layer (12 layers per utterance); this one is ~10.
// The block reads a halo of its neighbours' columns: x_out must not be x (Engine::run_dds rotates three buffers).
// ---------------------------------------------------------------------------------------------------------
typedef float dds_floatx16 __attribute__((ext_vector_type(16)));
typedef int dds_int4v __attribute__((ext_vector_type(4)));
typedef _Float16 dds_half8 __attribute__((ext_vector_type(8)));
typedef __bf16 dds_bf16x8 __attribute__((ext_vector_type(8)));

struct DdsLayerParams {
    const float* x;
    int64_t x_bs;
    int x_cs;
    float* y;
    int64_t y_bs;
    int y_cs;
    const float *dw_w, *dw_b, *g1, *b1, *pw_b, *g2, *b2;
    const float* wp;      // fp32 A fragments (pack_conv_weights)
    const uint16_t* wp16;  // 16-bit A fragments (pack_conv_weights16)
    const int* lens;
    int channels, tmax, k, dil, nchunks;
    float eps;
    const uint16_t* gelu_tab;  // emulated ggml GELU table (Q8) or nullptr
#ifdef VITS_PHASE_TIMING  // developer instrumentation (tools/dds_micro.hip): per-block phase timestamps, 100 MHz clock
    unsigned long long* dbg;
#endif
};
#ifdef VITS_PHASE_TIMING
#define DDS_STAMP(i)                                                             \
    do {                                                                         \
        if (threadIdx.x == 0 && p.dbg) p.dbg[blockIdx.x * 16 + (i)] = wall_clock64(); \
    } while (0)
#else
#define DDS_STAMP(i) \
    do {             \
    } while (0)
#endif

// MAXCH: upper bound of nchunks = channels / 32 — the wave's whole set of weight fragments is fetched into registers at the START
// of the kernel (at batch 1 they come from HBM: six dependent round trips inside the MFMA loop were 1/3 of the kernel), and the
// per-channel parameters go to LDS with the input tile, so that every later phase runs out of LDS and registers.
template <int ARITH, int MAXCH>
__global__ __launch_bounds__(32 * LN_GROUPS) void dds_layer_kernel(DdsLayerParams p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int NT = 32;
    constexpr int UB = 12;  // channels a thread works on at once (192 channels / 16 groups: one step)
    const int H = p.channels;
    const int pad = (p.k * p.dil - p.dil) / 2;  // vits.cpp:660
    const int xw = NT + 2 * pad;
    float* xt = sm;                        // [H][xw]   input tile with halo (also the residual)
    float* ht = xt + ((H * xw + 3) & ~3);  // [H][NT]   depthwise output -> gelu(LN1) -> pointwise output
    float* red = ht + H * NT;              // [2][LN_GROUPS][NT]
    float* prm = red + 2 * LN_GROUPS * NT;  // [6][H] dw_b, g1, b1, pw_b, g2, b2; then [H][k] dw_w
    dds_int4v* h16 = reinterpret_cast<dds_int4v*>(prm + ((6 * H + H * p.k + 3) & ~3));  // [H/8][NT] 16-bit operand slots (ARITH != 0)
    const int b = blockIdx.y, t0 = blockIdx.x * NT;
    DDS_STAMP(0);
    const int len = p.lens ? p.lens[b] : p.tmax;
    if (t0 >= len) return;
    const int tid = threadIdx.x, tl = tid & 31, gq = tid >> 5;
    const int lane = tid & 63, wid = tid >> 6;
    const int nmt = H >> 5;
    DDS_STAMP(1);
    // ---- everything this block reads from memory, issued at once: parameters and the input tile first, the weight fragments behind
    // them (147 KB per block in fp32: they may still be in flight while the depthwise phase runs — memory returns in order) ----
    float pv[2][7];
    {
        constexpr int NTH = 32 * LN_GROUPS;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int c = tid + r * NTH < H ? tid + r * NTH : 0;
            pv[r][0] = p.dw_b[c];
            pv[r][1] = p.g1[c];
            pv[r][2] = p.b1[c];
            pv[r][3] = p.pw_b[c];
            pv[r][4] = p.g2[c];
            pv[r][5] = p.b2[c];
        }
    }
    float4 af[ARITH == 0 ? MAXCH * 4 : 1];
    dds_int4v ah[ARITH != 0 ? MAXCH * 2 : 1];
    {
        // flat index over [H][xw], up to 24 loads in flight per thread (a load-then-store loop exposed one memory latency per element)
        const float* xb = p.x + (int64_t)b * p.x_bs;
        constexpr int NTH = 32 * LN_GROUPS, XB = 24;
        const int total = H * xw, dq = NTH / xw, dr = NTH % xw;
        int c = tid / xw, i = tid - c * xw;
        for (int base = tid; base < total || base == tid; base += XB * NTH) {
            float v[XB];
#pragma unroll
            for (int u = 0; u < XB; ++u) {
                const int t = t0 - pad + i;
                v[u] = (base + u * NTH < total && t >= 0 && t < len) ? xb[(int64_t)c * p.x_cs + t] : 0.f;
                c += dq;
                i += dr;
                if (i >= xw) {
                    i -= xw;
                    ++c;
                }
            }
            if (base == tid) {
                float wv[2] = {0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 2; ++r)
                    if (tid + r * NTH < H * p.k) wv[r] = p.dw_w[tid + r * NTH];
                if (wid < nmt) {
                    if constexpr (ARITH == 0) {
                        const float4* wp4 = reinterpret_cast<const float4*>(p.wp) + (size_t)wid * p.nchunks * 4 * 64 + lane;
#pragma unroll
                        for (int q = 0; q < MAXCH * 4; ++q) af[q] = wp4[(q < p.nchunks * 4 ? q : 0) * 64];
                    } else {
                        const dds_int4v* wq = reinterpret_cast<const dds_int4v*>(p.wp16) + (size_t)wid * p.nchunks * 2 * 64 + lane;
#pragma unroll
                        for (int q = 0; q < MAXCH * 2; ++q) ah[q] = wq[(q < p.nchunks * 2 ? q : 0) * 64];
                    }
                }
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    if (tid + r * NTH < H) {
#pragma unroll
                        for (int q = 0; q < 6; ++q) prm[q * H + tid + r * NTH] = pv[r][q];
                    }
                    if (tid + r * NTH < H * p.k) prm[6 * H + tid + r * NTH] = wv[r];
                }
                for (int q = tid + 2 * NTH; q < H * p.k; q += NTH) prm[6 * H + q] = p.dw_w[q];  // (k > 5 at 192 channels)
            }
#pragma unroll
            for (int u = 0; u < XB; ++u)
                if (base + u * NTH < total) xt[base + u * NTH] = v[u];
        }
    }
    __syncthreads();
    DDS_STAMP(2);
    const float *dw_b = prm, *g1 = prm + H, *b1 = prm + 2 * H, *pw_b = prm + 3 * H, *g2 = prm + 4 * H, *b2 = prm + 5 * H, *dw_w = prm + 6 * H;
    // ---- depthwise conv + LayerNorm 1 + gelu (dds_depthwise_kernel, same expressions) ----
    {
        // (UB channels per step: the LDS reads of the chains overlap; every chain and the order of s are those of the one-channel loop)
        constexpr int CS = LN_GROUPS;
        float s = 0.f;
        for (int c0 = gq; c0 < H; c0 += UB * CS) {
            float a[UB];
            int cc[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                cc[u] = c0 + u * CS < H ? c0 + u * CS : c0;
                a[u] = dw_b[cc[u]];
            }
            for (int j = 0; j < p.k; ++j) {
#pragma unroll
                for (int u = 0; u < UB; ++u) a[u] += round_arith(dw_w[cc[u] * p.k + j], ARITH) * round_arith(xt[cc[u] * xw + tl + j * p.dil], ARITH);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u)
                if (c0 + u * CS < H) {
                    ht[cc[u] * NT + tl] = a[u];
                    s += a[u];
                }
        }
        red[gq * NT + tl] = s;
        __syncthreads();
        float msum = 0.f;
#pragma unroll
        for (int q = 0; q < LN_GROUPS; ++q) msum += red[q * NT + tl];
        const float mean = msum / (float)H;
        float vs = 0.f;
        for (int c0 = gq; c0 < H; c0 += UB * LN_GROUPS) {
            float hv[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) hv[u] = ht[(c0 + u * LN_GROUPS < H ? c0 + u * LN_GROUPS : c0) * NT + tl];
#pragma unroll
            for (int u = 0; u < UB; ++u)
                if (c0 + u * LN_GROUPS < H) {
                    const float d = hv[u] - mean;
                    vs += d * d;
                }
        }
        red[(LN_GROUPS + gq) * NT + tl] = vs;
        __syncthreads();
        float vsum = 0.f;
#pragma unroll
        for (int q = 0; q < LN_GROUPS; ++q) vsum += red[(LN_GROUPS + q) * NT + tl];
        const float var = vsum / (float)H;
        const float inv = 1.0f / sqrtf(var + p.eps);
        for (int c0 = gq; c0 < H; c0 += UB * LN_GROUPS) {
            float hv[UB], gg[UB], bb[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int c = c0 + u * LN_GROUPS < H ? c0 + u * LN_GROUPS : c0;
                hv[u] = ht[c * NT + tl];
                gg[u] = g1[c];
                bb[u] = b1[c];
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int c = c0 + u * LN_GROUPS;
                if (c >= H) continue;
                float v = gelu_op((hv[u] - mean) * inv * gg[u] + bb[u], p.gelu_tab);
                if constexpr (ARITH == 0) {
                    ht[c * NT + tl] = v;
                } else {
                    // (the three-launch path stores the fp32 value and rounds it in another kernel; left alone, hipcc folds gelu's last
                    // product into the conversion — v_fma_mixlo_f16, ONE rounding instead of two)
                    asm volatile("" : "+v"(v));
                    uint16_t* slot = reinterpret_cast<uint16_t*>(h16 + (c >> 3) * NT + tl);
                    if constexpr (ARITH == 2) {
                        const _Float16 h = (_Float16)v;
                        slot[c & 7] = __builtin_bit_cast(uint16_t, h);
                    } else {
                        const __bf16 h = (__bf16)v;
                        slot[c & 7] = __builtin_bit_cast(uint16_t, h);
                    }
                }
            }
        }
    }
    __syncthreads();
    DDS_STAMP(3);
    // ---- pointwise conv on the matrix cores: wave w owns output rows 32w..32w+31 (the MFMA chain of conv_mfma.hip / conv16.hip) ----
    dds_floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (wid < nmt) {
        if constexpr (ARITH == 0) {
            const float* bcol = ht + (lane >> 5) * NT + (lane & 31);
#pragma unroll
            for (int c = 0; c < MAXCH; ++c) {
                if (c < p.nchunks) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float* bb = bcol + (c * 32 + 8 * g) * NT;
                        const float4 a = af[c * 4 + g];
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bb[0 * NT], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bb[2 * NT], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bb[4 * NT], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bb[6 * NT], acc, 0, 0, 0);
                    }
                }
            }
        } else {
            const dds_int4v* bcol = h16 + (lane >> 5) * NT + (lane & 31);
#pragma unroll
            for (int c = 0; c < MAXCH; ++c) {
                if (c < p.nchunks) {
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        const dds_int4v a = ah[c * 2 + kk];
                        const dds_int4v bq = bcol[(c * 4 + 2 * kk) * NT];
                        if constexpr (ARITH == 1)
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(dds_bf16x8, a), __builtin_bit_cast(dds_bf16x8, bq), acc, 0, 0, 0);
                        else
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(dds_half8, a), __builtin_bit_cast(dds_half8, bq), acc, 0, 0, 0);
                    }
                }
            }
        }
    }
    DDS_STAMP(4);
    __syncthreads();  // (fp32: every wave is done reading ht)
    DDS_STAMP(5);
    if (wid < nmt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wid * 32 + (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3);
            ht[row * NT + (lane & 31)] = acc[r] + pw_b[row];
        }
    }
    __syncthreads();
    DDS_STAMP(6);
    // ---- LayerNorm 2 + gelu + residual (add_layer_norm_kernel with post_gelu and add_to, same expressions) ----
    {
        float s = 0.f;
        for (int c0 = gq; c0 < H; c0 += UB * LN_GROUPS) {
            float hv[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) hv[u] = ht[(c0 + u * LN_GROUPS < H ? c0 + u * LN_GROUPS : c0) * NT + tl];
#pragma unroll
            for (int u = 0; u < UB; ++u)
                if (c0 + u * LN_GROUPS < H) s += hv[u];
        }
        red[gq * NT + tl] = s;
        __syncthreads();
        float msum = 0.f;
#pragma unroll
        for (int q = 0; q < LN_GROUPS; ++q) msum += red[q * NT + tl];
        const float mean = msum / (float)H;
        float vs = 0.f;
        for (int c0 = gq; c0 < H; c0 += UB * LN_GROUPS) {
            float hv[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) hv[u] = ht[(c0 + u * LN_GROUPS < H ? c0 + u * LN_GROUPS : c0) * NT + tl];
#pragma unroll
            for (int u = 0; u < UB; ++u)
                if (c0 + u * LN_GROUPS < H) {
                    const float d = hv[u] - mean;
                    vs += d * d;
                }
        }
        red[(LN_GROUPS + gq) * NT + tl] = vs;
        __syncthreads();
        float vsum = 0.f;
#pragma unroll
        for (int q = 0; q < LN_GROUPS; ++q) vsum += red[(LN_GROUPS + q) * NT + tl];
        const float var = vsum / (float)H;
        const float inv = 1.0f / sqrtf(var + p.eps);
        const int t = t0 + tl;
        if (t >= len) return;
        float* yb = p.y + (int64_t)b * p.y_bs + t;
        for (int c0 = gq; c0 < H; c0 += UB * LN_GROUPS) {
            float hv[UB], gg[UB], bb[UB], xr[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int c = c0 + u * LN_GROUPS < H ? c0 + u * LN_GROUPS : c0;
                hv[u] = ht[c * NT + tl];
                gg[u] = g2[c];
                bb[u] = b2[c];
                xr[u] = xt[c * xw + pad + tl];
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int c = c0 + u * LN_GROUPS;
                if (c >= H) continue;
                float v = (hv[u] - mean) * inv * gg[u] + bb[u];
                v = gelu_op(v, p.gelu_tab);
                asm volatile("" : "+v"(v));  // (the three-launch path adds in a separate statement behind a branch: no fma of gelu's last product with this add)
                yb[(int64_t)c * p.y_cs] = xr[u] + v;
            }
        }
        DDS_STAMP(7);
    }
}

#ifdef VITS_PHASE_TIMING
unsigned long long* g_dds_dbg = nullptr;
#endif
static size_t dds_layer_lds(int channels, int k, int dil) {
    const int xw = 32 + (k * dil - dil);
    return sizeof(float) * (((size_t)channels * xw + 3) / 4 * 4 + (size_t)channels * 32 + 2 * 32 * LN_GROUPS + ((size_t)channels * (6 + k) + 3) / 4 * 4) + (size_t)channels * 64;
}

bool dds_layer_supported(const PackedConv& pw, int channels, int k, int dil, int arith) {
    if (channels <= 0 || (channels & 31) || channels > 32 * (LN_GROUPS / 2)) return false;  // one wave per 32 output rows, 8 waves
    if (pw.cin != channels || pw.cout != channels || pw.kt != 1 || pw.epi != EPI_STD || !pw.bias) return false;
    if (arith == VITS_ARITH_F32 ? !pw.wp : !pw.wp16) return false;
    if (k < 1 || dil < 1 || ((k * dil - dil) & 1)) return false;
    return dds_layer_lds(channels, k, dil) <= 150 * 1024;
}

hipError_t launch_dds_layer(TensorRef x, TensorRef y, const float* dw_w, const float* dw_b, const float* g1, const float* b1, const PackedConv& pw, const float* g2,
                            const float* b2, const int* lens, int batch, int channels, int tmax, int k, int dil, float eps, int arith, hipStream_t s, GgmlTables tabs) {
    if (!dds_layer_supported(pw, channels, k, dil, arith) || x.p == y.p) return hipErrorInvalidValue;
    DdsLayerParams p;
    p.x = x.p;
    p.x_bs = x.bs;
    p.x_cs = x.cs;
    p.y = y.p;
    p.y_bs = y.bs;
    p.y_cs = y.cs;
    p.dw_w = dw_w;
    p.dw_b = dw_b;
    p.g1 = g1;
    p.b1 = b1;
    p.pw_b = pw.bias;
    p.g2 = g2;
    p.b2 = b2;
    p.wp = pw.wp;
    p.wp16 = pw.wp16;
    p.lens = lens;
    p.channels = channels;
    p.tmax = tmax;
    p.k = k;
    p.dil = dil;
    p.nchunks = pw.nchunks;
    p.eps = eps;
    p.gelu_tab = tabs.gelu;
#ifdef VITS_PHASE_TIMING
    p.dbg = g_dds_dbg;
#endif
    const size_t lds = dds_layer_lds(channels, k, dil);
    dim3 grid((tmax + 31) / 32, batch);
#define VITS_DDS_LAUNCH1(A, M)                                                                                                         \
    do {                                                                                                                               \
        static BigLdsOnce big;                                                                                          \
        if (lds > 64 * 1024 && big.needed()) {                                                                 \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dds_layer_kernel<A, M>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            if (e != hipSuccess) return e;                                                                                             \
            big.done();                                                                                \
        }                                                                                                                              \
        VITS_KLAUNCH((dds_layer_kernel<A, M>), grid, dim3(32 * LN_GROUPS), lds, s, p);                                           \
    } while (0)
#define VITS_DDS_LAUNCH(A)                       \
    do {                                         \
        if (pw.nchunks <= 2) VITS_DDS_LAUNCH1(A, 2);      \
        else if (pw.nchunks <= 4) VITS_DDS_LAUNCH1(A, 4); \
        else if (pw.nchunks <= 6) VITS_DDS_LAUNCH1(A, 6); \
        else VITS_DDS_LAUNCH1(A, 8);             \
    } while (0)
    if (arith == VITS_ARITH_F32) VITS_DDS_LAUNCH(0);
    else if (arith == VITS_ARITH_BF16) VITS_DDS_LAUNCH(1);
    else VITS_DDS_LAUNCH(2);
#undef VITS_DDS_LAUNCH1
#undef VITS_DDS_LAUNCH
    return hipGetLastError();
}

// conv_pre of a conv flow: 1 -> channels pointwise conv of latent row zc (vits.cpp:864): y[c][t] = w[c]*z[zc][t] + b[c]
//   optionally + cond[c][t]: the DDS block's "inputs + global_conditioning" (vits.cpp:651-653), same order of additions
__global__ void pointwise_from1_kernel(const float* z, int64_t z_bs, int z_cs, int zc, const float* w, const float* bias, const float* cond, int64_t c_bs,
                                       int c_cs, float* y, int64_t y_bs, int y_cs, const int* lens, int tmax, int arith) {
    const int b = blockIdx.z, c = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
    const int len = lens ? lens[b] : tmax;
    if (t >= len) return;
    float v = round_arith(w[c], arith) * round_arith(z[(int64_t)b * z_bs + (int64_t)zc * z_cs + t], arith) + bias[c];
    if (cond) v = v + cond[(int64_t)b * c_bs + (int64_t)c * c_cs + t];
    y[(int64_t)b * y_bs + (int64_t)c * y_cs + t] = v;
}
hipError_t launch_pointwise_from1(TensorRef z, int zc, const float* w, const float* bias, TensorRef cond, TensorRef y, const int* lens, int batch, int channels,
                                  int tmax, hipStream_t s, int arith) {
    dim3 grid((tmax + 63) / 64, channels, batch);
    VITS_KLAUNCH(pointwise_from1_kernel, grid, dim3(64), 0, s, z.p, z.bs, z.cs, zc, w, bias, cond.p, cond.bs, cond.cs, y.p, y.bs, y.cs, lens, tmax, arith);
    return hipGetLastError();
}

__global__ void fill_rows_kernel(float* x, int64_t bs, int cs, float v, int tmax) {
    const int b = blockIdx.z, c = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tmax) return;
    x[(int64_t)b * bs + (int64_t)c * cs + t] = v;
}
hipError_t launch_fill_rows(TensorRef x, int channels, float v, int batch, int tmax, hipStream_t s) {
    dim3 grid((tmax + 255) / 256, channels, batch);
    VITS_KLAUNCH(fill_rows_kernel, grid, dim3(256), 0, s, x.p, x.bs, x.cs, v, tmax);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Inverse rational-quadratic spline on latent row zc (unconstrained_rational_quadratic_spline, vits.cpp:804-852 + :695-802;
// HF modeling_vits.py:139-163,211-302). u = conv_proj output [3*bins-1][T]. One block per utterance, one thread per token (tokens
// beyond 512 loop).
//   VITS_MODE_HF: identity outside [-B, B], the spline inside (HF:143-151).
//   VITS_MODE_REFERENCE: Q3 (:720), on the LAST token Q4 (ggml-util.h:235-236,252-253), and the masked get / set pair of :832-849
//   LITERALLY (Q6): tensor_masked_get keeps the shape (custom-ops.h:746-749) while tensor_masked_set consumes its values sequentially
//   (custom-ops.h:836-850). Every token goes through the spline — an outside token as input 0 with zeroed widths / heights / padded
//   derivatives (:834-840) —, the k-th INSIDE token receives the result of token k, and the j-th OUTSIDE token receives element j of
//   [x_t if inside_t else 0]. While every latent of the utterance lies inside (all 8,192 ids of the benchmark batch) that is the identity
//   permutation and each token keeps its own spline output: the block only pays one __syncthreads_or for it.
// ---------------------------------------------------------------------------------------------------------
constexpr int MAX_BINS = 16;

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : (float)log(1.0 + exp((double)x)); }  // custom-ops.h:872-879

// rational_quadratic_spline (vits.cpp:695-802) on one token's row; masked: the row's parameters were zeroed by :837-840
// ggml_soft_max over the bins (vits.cpp:719,735), in place: fp32 by default, ggml's fp16 exp table + double sum when exp_tab is given (Q8)
template <bool TAB>
__device__ __forceinline__ void spline_softmax(float* v, int nb, const uint16_t* exp_tab) {
    float mx = -INFINITY;
    for (int i = 0; i < nb; ++i) mx = fmaxf(mx, v[i]);
    if constexpr (TAB) {
        double sum = 0.0;
        for (int i = 0; i < nb; ++i) {
            v[i] = ggml_table_lookup(exp_tab, v[i] - mx);
            sum += (double)v[i];
        }
        const float inv = (float)(1.0 / sum);
        for (int i = 0; i < nb; ++i) v[i] *= inv;
        return;
    }
    float sum = 0.f;
    for (int i = 0; i < nb; ++i) {
        v[i] = expf(v[i] - mx);
        sum += v[i];
    }
    for (int i = 0; i < nb; ++i) v[i] /= sum;
}

template <bool TAB>
__device__ __forceinline__ float spline_row(float x, const float* ub, int u_cs, int nb, float B, float inv_sqrt, int mode, bool q4, bool masked, const uint16_t* exp_tab) {
    const float min_w = 1e-3f, min_h = 1e-3f, min_d = 1e-3f;
    float W[MAX_BINS], H[MAX_BINS], cw[MAX_BINS + 1], ch[MAX_BINS + 1];
    // widths
    {
        for (int i = 0; i < nb; ++i) W[i] = masked ? 0.f : ub[(int64_t)i * u_cs] * inv_sqrt;
        spline_softmax<TAB>(W, nb, exp_tab);
        if (mode == VITS_MODE_REFERENCE) {
            const float sc = min_w + (1 - min_w * nb);  // Q3
            for (int i = 0; i < nb; ++i) W[i] = W[i] * sc;
        } else {
            for (int i = 0; i < nb; ++i) W[i] = min_w + (1 - min_w * nb) * W[i];
        }
        float cum = 0.f;
        cw[0] = 0.f;
        for (int i = 0; i < nb; ++i) {
            cum += W[i];
            cw[i + 1] = cum;
        }
        for (int i = 0; i <= nb; ++i) cw[i] = (B - (-B)) * cw[i] + (-B);
        cw[0] = -B;
        if (!q4) cw[nb] = B;
        for (int i = 0; i < nb; ++i) W[i] = cw[i + 1] - cw[i];
    }
    {
        for (int i = 0; i < nb; ++i) H[i] = masked ? 0.f : ub[(int64_t)(nb + i) * u_cs] * inv_sqrt;
        spline_softmax<TAB>(H, nb, exp_tab);
        for (int i = 0; i < nb; ++i) H[i] = min_h + (1 - min_h * nb) * H[i];
        float cum = 0.f;
        ch[0] = 0.f;
        for (int i = 0; i < nb; ++i) {
            cum += H[i];
            ch[i + 1] = cum;
        }
        for (int i = 0; i <= nb; ++i) ch[i] = (B - (-B)) * ch[i] + (-B);
        ch[0] = -B;
        if (!q4) ch[nb] = B;
        for (int i = 0; i < nb; ++i) H[i] = ch[i + 1] - ch[i];
    }
    int bin = -1;
    for (int i = 0; i <= nb; ++i) {
        float loc = ch[i];
        if (i == nb && !q4) loc += 1e-6f;
        if (x >= loc) bin++;
    }
    bin = min(max(bin, 0), nb - 1);
    const float constant = (float)log(exp(1.0 - (double)min_d) - 1.0);
    auto deriv = [&](int i) -> float {
        float ud;
        if (masked) ud = 0.f;  // (the constants of :829-830 are zeroed by the mask as well, :840)
        else if (i == 0) ud = constant;
        else if (i == nb) ud = q4 ? 0.f : constant;
        else ud = ub[(int64_t)(2 * nb + i - 1) * u_cs];
        return min_d + softplus_f(ud);
    };
    float in_cw = 0.f, in_w = 1.f, in_ch = 0.f, in_h = 1.f;
    for (int i = 0; i < nb; ++i)  // register arrays: select without dynamic indexing
        if (i == bin) {
            in_cw = cw[i];
            in_w = W[i];
            in_ch = ch[i];
            in_h = H[i];
        }
    const float d0 = deriv(bin), d1 = deriv(bin + 1);
    const float delta = in_h / in_w;
    const float i1 = d0 + d1 - 2 * delta;
    const float i2 = x - in_ch;
    const float i3 = i2 * i1;
    const float a = in_h * (delta - d0) + i3;
    const float bq = in_h * d0 - i3;
    const float cc = -delta * i2;
    const float disc = bq * bq - 4 * a * cc;
    const float root = (2 * cc) / (-bq - sqrtf(disc));
    return root * in_w + in_cw;
}

// (TAB: the emulated-ggml soft-max is a separate build, so that the default one carries no table code)
template <int NT, bool TAB>
__global__ __launch_bounds__(NT) void spline_kernel(const float* u, int64_t u_bs, int u_cs, float* z, int64_t z_bs, int z_cs, int zc, const int* lens, int tmax,
                                                      int nb, float B, float inv_sqrt, int mode, const uint16_t* exp_tab) {
    extern __shared__ float spline_lds[];  // [3][tpad]: masked input, spline result, inside flag of every token of the utterance; then [3 nb - 1][blockDim] parameters
    const int b = blockIdx.x;
    const int len = lens ? lens[b] : tmax;
    const int tpad = (tmax + 63) & ~63;
    float* s_val = spline_lds;
    float* s_res = spline_lds + tpad;
    int* s_in = reinterpret_cast<int*>(spline_lds + 2 * tpad);
    float* s_u = spline_lds + 3 * tpad;  // the parameters of the block's current chunk of tokens, [3 nb - 1][nthr]
    float* zrow = z + (int64_t)b * z_bs + (int64_t)zc * z_cs;
    const bool ref = mode == VITS_MODE_REFERENCE;
    int any_outside = 0;
    // A token's 3 nb - 1 parameters are 3 nb - 1 rows of u, one cache line each per token: read through the pointer inside spline_row they were four
    // dependent rounds of strided loads (16 us per launch at 128 tokens, most of it memory latency). The block stages them chunk by chunk with
    // coalesced loads, all rows in flight at once; the values — and every operation on them — are the same.
    const int nthr = (int)blockDim.x, nrow = 3 * nb - 1;
    for (int c0 = 0; c0 < len; c0 += nthr) {
        const int t = c0 + (int)threadIdx.x;
        {
            const float* ub = u + (int64_t)b * u_bs + (t < len ? t : len - 1);
            float v[3 * MAX_BINS - 1];
#pragma unroll
            for (int r = 0; r < 3 * MAX_BINS - 1; ++r) v[r] = r < nrow ? ub[(int64_t)r * u_cs] : 0.f;
#pragma unroll
            for (int r = 0; r < 3 * MAX_BINS - 1; ++r)
                if (r < nrow) s_u[r * nthr + threadIdx.x] = v[r];
        }
        // (each thread reads back only what it wrote: no barrier needed)
        if (t < len) {
            const float x = zrow[t];
            const bool inside = x >= -B && x <= B;
            const float* ub = s_u + threadIdx.x;
            float r = x;  // HF: identity outside the interval (HF:143-151)
            if (inside || ref) r = spline_row<TAB>(inside ? x : 0.f, ub, nthr, nb, B, inv_sqrt, mode, ref && t == len - 1, !inside, exp_tab);
            if (ref) {
                s_val[t] = inside ? x : 0.f;  // tensor_masked_get(inputs, inside_interval_mask): the shape is kept
                s_res[t] = r;
                s_in[t] = inside ? 1 : 0;
                any_outside |= inside ? 0 : 1;
            } else {
                zrow[t] = r;
            }
        }
    }
    if (!ref) return;
    if (!__syncthreads_or(any_outside)) {
        // every latent inside: both masked_set calls are the identity permutation
        for (int t = threadIdx.x; t < len; t += blockDim.x) zrow[t] = s_res[t];
        return;
    }
    // Q6: the j-th outside token takes element j of the masked input (:832), the k-th inside token the result of token k (:849).
    // nout(t) = outside tokens before t: every thread counts a contiguous chunk, one thread scans the per-thread counts (the
    // first version let every thread count from token 0: O(T^2) LDS reads per utterance)
    __shared__ int s_part[NT];
    const int per = (len + nthr - 1) / nthr;
    const int beg = min((int)threadIdx.x * per, len), end = min(beg + per, len);
    int cnt = 0;
    for (int t = beg; t < end; ++t) cnt += 1 - s_in[t];
    s_part[threadIdx.x] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < nthr; ++i) {
            const int v = s_part[i];
            s_part[i] = run;
            run += v;
        }
    }
    __syncthreads();
    int nout = s_part[threadIdx.x];
    for (int t = beg; t < end; ++t) {
        zrow[t] = s_in[t] ? s_res[t - nout] : s_val[nout];
        nout += 1 - s_in[t];
    }
}

hipError_t launch_spline(TensorRef u, TensorRef z, int zc, const int* lens, int batch, int tmax, int bins, float tail, float inv_sqrt, int mode, hipStream_t s,
                         GgmlTables tabs) {
    if (bins > MAX_BINS || tmax > 4096) return hipErrorInvalidValue;
    const int tpad = (tmax + 63) & ~63;
    // 512 threads: one token per thread up to 512 tokens, two rounds for 1024-id inputs (a 1024-thread build is capped at 128 VGPRs and spills
    // the bin arrays: 74 us per launch against 32 at 8 x 1024 ids)
#define VITS_SPLINE_LAUNCH(NT, TAB)                                                                                                                 \
    do {                                                                                                                                            \
        const size_t lds_ = ((size_t)3 * tpad + (size_t)(3 * bins - 1) * (tpad < NT ? tpad : NT)) * sizeof(float);                                  \
        static BigLdsOnce big;                                                                                                                      \
        if (lds_ > 64 * 1024 && big.needed()) {                                                                                                     \
            if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spline_kernel<NT, TAB>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024)) return e; /* (+ the kernel's static s_part) */ \
            big.done();                                                                                                                             \
        }                                                                                                                                           \
        VITS_KLAUNCH((spline_kernel<NT, TAB>), dim3(batch), dim3(tpad < NT ? tpad : NT), lds_, s, u.p, u.bs, u.cs, z.p, z.bs, z.cs, zc, lens, tmax, bins, tail, inv_sqrt, \
                     mode, tabs.exp);                                                                                                               \
    } while (0)
    if (tabs.exp) {
        VITS_SPLINE_LAUNCH(512, true);
    } else {
        VITS_SPLINE_LAUNCH(512, false);
    }
#undef VITS_SPLINE_LAUNCH
    return hipGetLastError();
}

// elementwise affine, reverse (vits.cpp:901-925; Q5 sign): logical channel ch lives in physical row (c_first ? 1-ch : ch)
__global__ void affine_kernel(float* z, int64_t z_bs, int z_cs, int c_first, const float* translate, const float* log_scale, int sign, const int* lens,
                              int tmax) {
    const int b = blockIdx.z, ch = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
    const int len = lens ? lens[b] : tmax;
    if (t >= len) return;
    const int row = c_first ? 1 - ch : ch;
    float* p = z + (int64_t)b * z_bs + (int64_t)row * z_cs + t;
    *p = (*p - translate[ch]) * expf(sign > 0 ? log_scale[ch] : -log_scale[ch]);
}
hipError_t launch_affine(TensorRef z, int c_first, const float* translate, const float* log_scale, int sign, const int* lens, int batch, int tmax,
                         hipStream_t s) {
    dim3 grid((tmax + 63) / 64, 2, batch);
    VITS_KLAUNCH(affine_kernel, grid, dim3(64), 0, s, z.p, z.bs, z.cs, c_first, translate, log_scale, sign, lens, tmax);
    return hipGetLastError();
}

// duration latents: z[c][t] = N(0,1) * noise_scale_duration (vits.cpp:948-949), counter-based stream
__global__ void noise_dur_kernel(float* z, int64_t z_bs, int z_cs, const int* lens, int tmax, uint64_t seed, const int* seed_off, float scale) {
    const int b = blockIdx.z, c = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
    const int len = lens ? lens[b] : tmax;
    if (t >= len) return;
    z[(int64_t)b * z_bs + (int64_t)c * z_cs + t] = vits_counter_normal(seed + (uint64_t)(seed_off ? seed_off[b] : b), VITS_STREAM_NOISE_DUR, (uint64_t)c * len + t) * scale;
}
hipError_t launch_noise_dur(TensorRef z, const int* lens, int batch, int tmax, uint64_t seed, const int* seed_off, float scale, hipStream_t s) {
    dim3 grid((tmax + 63) / 64, 2, batch);
    VITS_KLAUNCH(noise_dur_kernel, grid, dim3(64), 0, s, z.p, z.bs, z.cs, lens, tmax, seed, seed_off, scale);
    return hipGetLastError();
}

__global__ void scale_rows_kernel(float* x, int64_t bs, int cs, float scale, int tmax) {
    const int b = blockIdx.z, c = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tmax) return;
    x[(int64_t)b * bs + (int64_t)c * cs + t] *= scale;
}
hipError_t launch_scale_rows(TensorRef x, int channels, float scale, int batch, int tmax, hipStream_t s) {
    dim3 grid((tmax + 63) / 64, channels, batch);
    VITS_KLAUNCH(scale_rows_kernel, grid, dim3(64), 0, s, x.p, x.bs, x.cs, scale, tmax);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// durations: d = ceil(exp(logw) * length_scale) (vits.cpp:996), per-utterance inclusive cumsum (:1001),
// frames L = max(1, sum) (:999-1000,1133) and the per-stage vocoder lengths. One block per utterance.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void durations_kernel(const float* logw, int64_t l_bs, int l_cs, int c, const int* lens, int tmax, float length_scale,
                                                        int fixed, float* dur, int* cum, int* frames, int* stage_lens, int n_stage, const int* stage_mul,
                                                        const int* stage_add, int batch, int exact) {
    __shared__ int part[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int len = lens ? lens[b] : tmax;
    const int per = (tmax + 255) / 256;
    const int beg = tid * per, end = min(beg + per, len);
    int s = 0;
    for (int t = beg; t < end; ++t) {
        const float lw = logw[(int64_t)b * l_bs + (int64_t)c * l_cs + t];
        // exact: the emulated-ggml mode — exp as the fixed polynomial both sides share (include/vits_exact_math.h) instead of the device library's
        float d = exact ? vx_duration(lw, length_scale) : ceilf(expf(lw) * length_scale);
        if (fixed > 0) d = (float)fixed;
        dur[(int64_t)b * tmax + t] = d;
        s += (int)d;
    }
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < 256; ++i) {
            const int v = part[i];
            part[i] = run;
            run += v;
        }
        const int L = max(1, run);
        frames[b] = L;
        for (int st = 0; st < n_stage; ++st) stage_lens[st * batch + b] = L * stage_mul[st] + stage_add[st];
    }
    __syncthreads();
    int run = part[tid];
    for (int t = beg; t < end; ++t) {
        run += (int)dur[(int64_t)b * tmax + t];
        cum[(int64_t)b * tmax + t] = run;
    }
    for (int t = max(beg, len); t < min(beg + per, tmax); ++t) {
        dur[(int64_t)b * tmax + t] = 0.f;
        cum[(int64_t)b * tmax + t] = 0x7fffffff;
    }
}

// ---- ((y0 + y1) [+ y2]) * scale over fp32 [b][c][t] tensors: the sum over a stage's resblocks as its own launch (fp32 path, small grids) -------------
// The fp32 resblock kernels carry the accumulation into the stage's shared sum in their last launch, so resblock j's last launch waits for resblock
// j - 1's (vits.cpp:622-635's order of the additions). With one or two utterances the three chains are a few dozen blocks per kernel and the waits are
// idle chip: here every resblock writes its own output and this kernel adds them in that order with the epilogue's expressions (a + v; the second
// resblock's scale is 1: v * 1 = v; then the scale; then the next upsampler's leaky_relu where the last resblock's epilogue applied it): same bits.
__global__ __launch_bounds__(256) void rb_sum3_std_kernel(const float* y0, const float* y1, const float* y2, int64_t bs, int cs, const int* lens, int tmax, float scale, int scale_div,
                                                           int post_act, float post_slope, float* out, int64_t o_bs, int o_cs) {
    const int b = blockIdx.z, c = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    const int len = lens ? lens[b] : tmax;
    if (t >= len) return;
    const int64_t go = (int64_t)b * bs + (int64_t)c * cs + t;
    float x = y0[go] + y1[go];
    if (y2) x = x + y2[go];
    x = scale_div ? x / scale : x * scale;
    if (post_act == 2) x = fmaxf(x, x * post_slope);
    out[(int64_t)b * o_bs + (int64_t)c * o_cs + t] = x;
}
hipError_t launch_rb_sum3_std(TensorRef y0, TensorRef y1, TensorRef y2, TensorRef out, int channels, const int* lens, int batch, int tmax, float scale, int scale_div, int post_act,
                              float post_slope, hipStream_t s) {
    if (!y0.p || !y1.p || !out.p || y0.bs != y1.bs || y0.cs != y1.cs || (y2.p && (y2.bs != y0.bs || y2.cs != y0.cs))) return hipErrorInvalidValue;
    dim3 grid((tmax + 255) / 256, channels, batch);
    VITS_KLAUNCH(rb_sum3_std_kernel, grid, dim3(256), 0, s, y0.p, y1.p, y2.p, y0.bs, y0.cs, lens, tmax, scale, scale_div, post_act, post_slope, out.p, out.bs, out.cs);
    return hipGetLastError();
}

hipError_t launch_durations(TensorRef logw, int c, const int* lens, int batch, int tmax, float length_scale, int fixed, float* dur, int* cum, int* frames,
                            int* stage_lens, int n_stage, const int* stage_mul, const int* stage_add, hipStream_t s, bool exact) {
    VITS_KLAUNCH(durations_kernel, dim3(batch), dim3(256), 0, s, logw.p, logw.bs, logw.cs, c, lens, tmax, length_scale, fixed, dur, cum, frames, stage_lens,
                       n_stage, stage_mul, stage_add, batch, exact ? 1 : 0);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// prior sampling through the monotonic alignment, as a GATHER:
//   a(j) = min{ i : j < cum_i };  z_p[c][j] = mean[c][a(j)] + eps[c][j] * exp(logvar[c][a(j)]) * noise_scale
// replaces the dense [L,T] one-hot matrix + two mul_mat (vits.cpp:1028-1057) and :1059-1063.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void zp_kernel(const float* mean, int64_t m_bs, int m_cs, const float* logvar, int64_t v_bs, int v_cs, const int* cum,
                                                 int cum_stride, const int* tok_lens, const int* frames, const float* noise, int64_t n_bs, int n_cs,
                                                 int noise_kind, uint64_t seed, const int* seed_off, float noise_scale, float* zp, int64_t z_bs, int z_cs, int channels, int tmax_tok) {
    __shared__ int tok[256];
    const int b = blockIdx.y, j0 = blockIdx.x * 256, tid = threadIdx.x;
    const int L = frames[b];
    if (j0 >= L) return;
    const int T = tok_lens ? tok_lens[b] : tmax_tok;
    const int j = j0 + tid;
    {
        // binary search: first i with cum[i] > j
        const int* cb = cum + (int64_t)b * cum_stride;
        int lo = 0, hi = T;  // answer in [0, T]
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cb[mid] > j) hi = mid;
            else lo = mid + 1;
        }
        tok[tid] = lo < T ? lo : -1;
    }
    __syncthreads();
    if (j >= L) return;
    const int a = tok[tid];
    // channels are split over blockIdx.z (every element is independent): one block per 256 frames walked all 192 channels
    // serially through the counter-based normal, 100 us at batch 1
    const int cpb = (channels + gridDim.z - 1) / gridDim.z;
    const int c_end = min(channels, (int)(blockIdx.z + 1) * cpb);
    for (int c = blockIdx.z * cpb; c < c_end; ++c) {
        const float mu = a >= 0 ? mean[(int64_t)b * m_bs + (int64_t)c * m_cs + a] : 0.f;
        const float lv = a >= 0 ? logvar[(int64_t)b * v_bs + (int64_t)c * v_cs + a] : 0.f;
        float e;
        if (noise_kind == VITS_NOISE_COUNTER) e = vits_counter_normal(seed + (uint64_t)(seed_off ? seed_off[b] : b), VITS_STREAM_NOISE_PRIOR, (uint64_t)c * L + j);
        else e = noise[(int64_t)b * n_bs + (int64_t)c * n_cs + j];
        float n = e * expf(lv);  // vits.cpp:1060
        n = n * noise_scale;     // :1061
        zp[(int64_t)b * z_bs + (int64_t)c * z_cs + j] = mu + n;
    }
}

hipError_t launch_zp(TensorRef mean, TensorRef logvar, const int* cum, int cum_stride, const int* tok_lens, const int* frames, TensorRef noise, int noise_kind,
                     uint64_t seed, const int* seed_off, float noise_scale, TensorRef zp, int batch, int channels, int lmax, hipStream_t s) {
    dim3 grid((lmax + 255) / 256, batch, 16);
    VITS_KLAUNCH(zp_kernel, grid, dim3(256), 0, s, mean.p, mean.bs, mean.cs, logvar.p, logvar.bs, logvar.cs, cum, cum_stride, tok_lens, frames, noise.p,
                       noise.bs, noise.cs, noise_kind, seed, seed_off, noise_scale, zp.p, zp.bs, zp.cs, channels, cum_stride);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// conv_post: leaky_relu -> Conv1d(C -> 1, k, no bias) -> tanh   (vits.cpp:638-642). One output row: a matrix
// tile would be 31/32 empty, so this is a VALU kernel; HBM-bound (reads C floats per output sample).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_post_kernel(const float* x, int64_t x_bs, int x_cs, const float* w, int cin, int k, float slope, float* pre,
                                                        int64_t p_bs, float* wave, int64_t w_bs, const int* lens, int tmax, int emit_lo, const int* emit_hi, int arith) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // weights [cin][k]
    const int b = blockIdx.y;
    const int len = lens ? lens[b] : tmax;
    // windowed vocoder (engine.cpp): only samples [emit_lo, emit_hi[b]) of this window are the utterance's own; the rest
    // is halo, computed from a cut edge and owned by a neighbouring window
    const int hi = emit_hi ? min(emit_hi[b], len) : len;
    const int t0 = emit_lo + blockIdx.x * 1024;
    if (t0 >= hi) return;
    for (int i = threadIdx.x; i < cin * k; i += 256) sm[i] = round_arith(w[i], arith);
    __syncthreads();
    const int pad = (k - 1) / 2;
    const float* xb = x + (int64_t)b * x_bs;
    // fast path (k = 7, 16-byte aligned rows, away from the sequence ends): one thread = 4 consecutive samples; per channel
    // three aligned float4 loads cover x[t-4 .. t+7] and feed all 28 products (the scalar path below loads every input 7x).
    // Same order of accumulation per sample (channel-major, tap-minor), so both paths give identical values.
    if (k == 7 && (x_cs & 3) == 0 && ((reinterpret_cast<uintptr_t>(xb) & 15) == 0) && (t0 & 3) == 0) {
        const int t = t0 + 4 * threadIdx.x;
        if (t >= hi) return;
        // interior threads read three aligned float4 per channel; a thread at a sequence end reads the same twelve values one by one,
        // zero outside [0, len) (what the padding contributes). The end threads used to walk a scalar loop of 4 x cin x 7 dependent
        // loads: at batch 1 that ONE thread was the whole kernel time (138 us for a 128-id utterance).
        const bool interior = t >= 4 && t + 8 <= len;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        // eight channels per step, their loads issued before the first product. Accumulation order: channel-major, tap-minor.
        constexpr int CB = 8;
        for (int c0 = 0; c0 < cin; c0 += CB) {
            float4 q[CB][3];
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                const int c = c0 + u < cin ? c0 + u : cin - 1;
                const float* xc = xb + (int64_t)c * x_cs;
                if (interior) {
                    const float4* xr = reinterpret_cast<const float4*>(xc + t - 4);
                    q[u][0] = xr[0], q[u][1] = xr[1], q[u][2] = xr[2];
                } else {
                    float e[12];
#pragma unroll
                    for (int i = 0; i < 12; ++i) {
                        const int tt = t - 4 + i;
                        e[i] = (tt >= 0 && tt < len) ? xc[tt] : 0.f;
                    }
                    q[u][0] = make_float4(e[0], e[1], e[2], e[3]);
                    q[u][1] = make_float4(e[4], e[5], e[6], e[7]);
                    q[u][2] = make_float4(e[8], e[9], e[10], e[11]);
                }
            }
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                if (c0 + u >= cin) break;
                float v[12] = {q[u][0].x, q[u][0].y, q[u][0].z, q[u][0].w, q[u][1].x, q[u][1].y,
                               q[u][1].z, q[u][1].w, q[u][2].x, q[u][2].y, q[u][2].z, q[u][2].w};  // v[i] = x[t - 4 + i]
#pragma unroll
                for (int i = 1; i < 11; ++i) v[i] = round_arith(v[i] > 0.f ? v[i] : v[i] * slope, arith);
                const float* wc = sm + (c0 + u) * 7;
#pragma unroll
                for (int j = 0; j < 7; ++j) {  // sample t + e reads x[t + e + j - 3] = v[e + j + 1]
                    const float wj = wc[j];
                    a0 += wj * v[j + 1];
                    a1 += wj * v[j + 2];
                    a2 += wj * v[j + 3];
                    a3 += wj * v[j + 4];
                }
            }
        }
        float* wp = wave + (int64_t)b * w_bs + t;
        if (t + 4 <= hi) {
            if (pre) *reinterpret_cast<float4*>(pre + (int64_t)b * p_bs + t) = make_float4(a0, a1, a2, a3);
            wp[0] = tanhf(a0), wp[1] = tanhf(a1), wp[2] = tanhf(a2), wp[3] = tanhf(a3);
        } else {
            const float av[4] = {a0, a1, a2, a3};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (t + e < hi) {
                    if (pre) pre[(int64_t)b * p_bs + t + e] = av[e];
                    wp[e] = tanhf(av[e]);
                }
        }
        return;
    }
    for (int u = 0; u < 4; ++u) {
        const int t = t0 + u * 256 + threadIdx.x;
        if (t >= hi) continue;
        float a = 0.f;
        for (int c = 0; c < cin; ++c) {
            const float* xr = xb + (int64_t)c * x_cs;
            for (int j = 0; j < k; ++j) {
                const int tt = t + j - pad;
                float v = (tt >= 0 && tt < len) ? xr[tt] : 0.f;
                v = round_arith(v > 0.f ? v : v * slope, arith);
                a += sm[c * k + j] * v;
            }
        }
        if (pre) pre[(int64_t)b * p_bs + t] = a;
        wave[(int64_t)b * w_bs + t] = tanhf(a);
    }
}

hipError_t launch_conv_post(TensorRef x, const float* w, int cin, int k, float slope, TensorRef pre_tanh, TensorRef wave, const int* lens, int batch, int tmax,
                            hipStream_t s, int emit_lo, const int* emit_hi, int arith) {
    dim3 grid((std::max(tmax - emit_lo, 1) + 1023) / 1024, batch);
    VITS_KLAUNCH(conv_post_kernel, grid, dim3(256), sizeof(float) * cin * k, s, x.p, x.bs, x.cs, w, cin, k, slope, pre_tanh.p, pre_tanh.bs, wave.p, wave.bs,
                       lens, tmax, emit_lo, emit_hi, arith);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// fp32 PCM -> int16 PCM on the device (test/main.cpp:31-33: static_cast<short>(clamp(x, -1, 1) * 32767)), so that the
// multi-GPU gather and the host copy move half the bytes. Pure streaming kernel: 8 samples per thread when both rows
// are 16-byte aligned (2 x dwordx4 in, 1 x dwordx4 out), scalar otherwise.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ short pcm16_of(float x) { return static_cast<short>(fmaxf(-1.0f, fminf(1.0f, x)) * 32767); }

__global__ __launch_bounds__(256) void pcm16_kernel(const float* src, int64_t s_bs, short* dst, int64_t d_bs, const int64_t* lens, int64_t cols, int vec) {
    const int b = blockIdx.y;
    const int64_t n = lens ? min(lens[b], cols) : cols;
    const float* sr = src + (int64_t)b * s_bs;
    short* dr = dst + (int64_t)b * d_bs;
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i0 >= n) return;
    if (vec && i0 + 8 <= n) {
        const float4 a = *reinterpret_cast<const float4*>(sr + i0), c = *reinterpret_cast<const float4*>(sr + i0 + 4);
        union {
            short h[8];
            int4 v;
        } o;
        o.h[0] = pcm16_of(a.x), o.h[1] = pcm16_of(a.y), o.h[2] = pcm16_of(a.z), o.h[3] = pcm16_of(a.w);
        o.h[4] = pcm16_of(c.x), o.h[5] = pcm16_of(c.y), o.h[6] = pcm16_of(c.z), o.h[7] = pcm16_of(c.w);
        *reinterpret_cast<int4*>(dr + i0) = o.v;
    } else {
        for (int64_t i = i0; i < min(i0 + 8, n); ++i) dr[i] = pcm16_of(sr[i]);
    }
}

hipError_t launch_pcm16(const float* src, int64_t src_stride, int16_t* dst, int64_t dst_stride, const int64_t* lens, int rows, int64_t cols, hipStream_t s) {
    if (rows <= 0 || cols <= 0) return hipSuccess;
    const int vec = (((uintptr_t)src | (uintptr_t)dst) & 15) == 0 && (src_stride & 3) == 0 && (dst_stride & 7) == 0;
    dim3 grid((unsigned)((cols + 2047) / 2048), rows);
    VITS_KLAUNCH(pcm16_kernel, grid, dim3(256), 0, s, src, src_stride, reinterpret_cast<short*>(dst), dst_stride, lens, cols, vec);
    return hipGetLastError();
}

}  // namespace vits
