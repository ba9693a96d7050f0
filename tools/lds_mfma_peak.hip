// Developer microbenchmark: what a 16-bit MFMA kernel can sustain when every MFMA takes its B operand from LDS — the situation of the narrow vocoder
// stages (C = 32: ONE 32-row tile of output channels, so a 16 x 32 activation fragment feeds exactly one v_mfma_f32_32x32x16; rbblock16.hip), against
// B fragments reused by 2 / 4 MFMAs (two / four row tiles per wave) and against the register-only loop of tools/mfma16_peak.hip. One ds_read_b128 per
// lane is 1 KB per wave; at the MFMA peak (one MFMA per 32 cycles and SIMD) four SIMDs ask for 128 B per cycle and CU — the LDS port's whole rate.
// Operands: random signs, exponents spread over 2^-7 .. 2^0 (what weights and activations look like; the power budget depends on it).
// Build: hipcc --offload-arch=gfx950 -O3 tools/lds_mfma_peak.hip -o tools/bin/lds_mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef int int4v __attribute__((ext_vector_type(4)));
__device__ unsigned long long clk[4];

__device__ inline int gen16x2(unsigned& h) {
    unsigned v = 0;
    for (int k = 0; k < 2; ++k) {
        h = h * 1664525u + 1013904223u;
        const unsigned r = h >> 8;
        const unsigned sign = (r >> 20) & 1u, ex = (r >> 16) & 7u;
        v |= ((sign << 15) | ((8u + ex) << 10) | (r & 0x3ffu)) << (16 * k);
    }
    return (int)v;
}

// REUSE: MFMAs per B fragment read (1 = C 32, 2 = two row tiles per wave, 4 = four; 0 = B from registers, no LDS at all). Three accumulators chains per
// row tile keep the matrix pipe fed like the 3-column-tile waves of rbblock16.
template <int REUSE>
__global__ __launch_bounds__(256) void lds_loop(float* out, int iters) {
    constexpr int PITCH = 440, NSLOT = 4 * PITCH;  // the C = 32 tile of rbblock16: 4 channel groups x 440 slots of 16 bytes = 28 KB
    __shared__ __attribute__((aligned(16))) int4v tile[NSLOT];
    unsigned h = (threadIdx.x + 1) * 2654435761u ^ (blockIdx.x * 40503u);
    for (int i = threadIdx.x; i < NSLOT; i += 256) tile[i] = int4v{gen16x2(h), gen16x2(h), gen16x2(h), gen16x2(h)};
    constexpr int MR = REUSE == 0 ? 1 : REUSE;
    int4v a[MR][2];
    for (int m = 0; m < MR; ++m)
        for (int i = 0; i < 2; ++i) a[m][i] = int4v{gen16x2(h), gen16x2(h), gen16x2(h), gen16x2(h)};
    int4v breg[3] = {tile[threadIdx.x], tile[threadIdx.x + 256], tile[threadIdx.x + 512]};
    floatx16 acc[MR][3];
    for (int m = 0; m < MR; ++m)
        for (int n = 0; n < 3; ++n)
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    typedef const __attribute__((address_space(3))) int4v* LdsV;
    LdsV base = (LdsV)(tile + (lane >> 5) * PITCH + wid * 96 + (lane & 31));  // lanes 0-31: channel group 0, lanes 32-63: group 1, 32 consecutive slots each
    unsigned long long t0 = 0, r0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        t0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
    }
    for (int it = 0; it < iters; ++it) {
        LdsV cur = base + ((it * 5) & 7);  // (a moving window: the reads cannot be hoisted out of the loop)
#pragma unroll
        for (int j = 0; j < 8; ++j) {  // eight taps' worth: offsets 0 .. 7 slots, two k-halves (groups 0/1 and 2/3)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                int4v b[3];
#pragma unroll
                for (int n = 0; n < 3; ++n) b[n] = REUSE == 0 ? breg[n] : cur[kk * 2 * PITCH + j + 32 * n];
#pragma unroll
                for (int m = 0; m < MR; ++m)
#pragma unroll
                    for (int n = 0; n < 3; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a[m][kk]), __builtin_bit_cast(half8, b[n]), acc[m][n], 0, 0, 0);
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = __builtin_amdgcn_s_memtime() - t0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    float s = 0.f;
    for (int m = 0; m < MR; ++m)
        for (int n = 0; n < 3; ++n)
            for (int r = 0; r < 16; ++r) s += acc[m][n][r];
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int REUSE>
static void run(float* d, int blocks_per_cu) {
    constexpr int MR = REUSE == 0 ? 1 : REUSE;
    const int iters = 6000 / MR;
    const int blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        const int launches = 6;
        for (int l = 0; l < launches; ++l) hipLaunchKernelGGL((lds_loop<REUSE>), dim3(blocks), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c[4];
        hipMemcpyFromSymbol(c, HIP_SYMBOL(clk), sizeof(c));
        const double mfmas = (double)launches * blocks * 4 * iters * 16.0 * MR * 3;
        const double fl = mfmas * 32768.0;
        const double lds_bytes = REUSE == 0 ? 0.0 : (double)launches * blocks * 4 * iters * 16.0 * 3 * 1024.0;
        if (rep == 1)
            printf("B fragment feeds %d MFMA(s)%s, %d blocks of 4 waves per CU: %.0f TFLOP/s = %.3f of 2500; LDS reads %.0f B/clk/CU at the measured clock %.3f GHz (port: 128)\n", MR,
                   REUSE == 0 ? " [registers only, no LDS]" : "", blocks_per_cu, fl / ms / 1e9, fl / ms / 1e9 / 2500.0,
                   lds_bytes / (ms * 1e-3) / 256.0 / ((double)c[0] / ((double)c[1] * 10.0) * 1e9), (double)c[0] / ((double)c[1] * 10.0));
    }
}
int main() {
    float* d;
    hipMalloc(&d, 4096);
    for (int bpc : {2, 3}) {
        run<0>(d, bpc);
        run<1>(d, bpc);
        run<2>(d, bpc);
    }
    run<4>(d, 2);
    return 0;
}
