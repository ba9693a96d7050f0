// Developer microbenchmark: the SUSTAINED 16-bit MFMA ceiling of this chip (v_mfma_f32_32x32x16_f16 / _bf16, registers only, no
// memory traffic), random operands, about a second per configuration, with the shader clock read in the kernel (s_memtime against the
// 100 MHz s_memrealtime). The chip clocks to its power budget: this — not 2.5 PFLOP/s at 2.4 GHz — is what a perfectly fed 16-bit
// MFMA kernel can reach. Build: hipcc --offload-arch=gfx950 -O3 tools/mfma16_peak.hip -o tools/bin/mfma16_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int int4v __attribute__((ext_vector_type(4)));
__device__ unsigned long long clk[4];
template <int WAVES, bool BF>
__global__ __launch_bounds__(WAVES * 64) void mfma_loop(float* out, int iters, int zero) {
    floatx16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    unsigned h = (threadIdx.x + 1) * 2654435761u ^ (blockIdx.x * 40503u);
    int4v a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        int* pa = (int*)&a[i];
        int* pb = (int*)&b[i];
        for (int e = 0; e < 4; ++e) {
            // zero == 0: two finite 16-bit values of magnitude ~0.1-0.25 (random mantissa, one sign, one exponent);
            // zero == 2: random sign, magnitudes spread over 2^-7 .. 2^0 (what activations and weights look like)
            auto gen = [&]() -> int {
                h = h * 1664525u + 1013904223u;
                if (zero == 1) return 0;
                if (zero == 0) return (int)((h & 0x03ff03ffu) | 0x30003000u);
                unsigned v = 0;
                for (int k = 0; k < 2; ++k) {
                    h = h * 1664525u + 1013904223u;
                    const unsigned r = h >> 8;
                    const unsigned sign = (r >> 20) & 1u, ex = (r >> 16) & 7u;
                    const unsigned bits = BF ? (sign << 15) | ((120u + ex) << 7) | (r & 0x7fu) : (sign << 15) | ((8u + ex) << 10) | (r & 0x3ffu);
                    v |= bits << (16 * k);
                }
                return (int)v;
            };
            pa[e] = gen();
            pb[e] = gen();
        }
    }
    unsigned long long t0 = 0, r0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        t0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (BF) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[(i + j) & 3]), acc[j], 0, 0, 0);
                else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a[i]), __builtin_bit_cast(half8, b[(i + j) & 3]), acc[j], 0, 0, 0);
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = __builtin_amdgcn_s_memtime() - t0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[threadIdx.x] = s;
}
template <int WAVES, bool BF>
static void run(const char* label, int zero, float* d, int waves_per_cu) {
    const int iters = 40000;  // 16 MFMAs per iteration per wave
    const int blocks = 256 * (waves_per_cu / WAVES);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        const int launches = 8;
        for (int l = 0; l < launches; ++l) hipLaunchKernelGGL((mfma_loop<WAVES, BF>), dim3(blocks), dim3(WAVES * 64), 0, 0, d, iters, zero);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c[4];
        hipMemcpyFromSymbol(c, HIP_SYMBOL(clk), sizeof(c));
        const double fl = (double)launches * blocks * WAVES * iters * 16.0 * 32768.0;
        printf("%s %s waves/CU %d: %.1f ms  %.0f TFLOP/s = %.3f of 2500; clock if 100 %% busy %.3f GHz; s_memtime / s_memrealtime -> %.3f GHz\n", BF ? "bf16" : "f16 ", label,
               waves_per_cu, ms, fl / ms / 1e9, fl / ms / 1e9 / 2500.0, fl / ms / 1e9 / 2500.0 * 2.4, (double)c[0] / ((double)c[1] * 10.0));
    }
}
int main() {
    float* d;
    hipMalloc(&d, 4096);
    run<4, false>("random", 0, d, 4);
    run<4, false>("random", 0, d, 8);
    run<4, false>("random", 0, d, 12);
    run<4, true>("random", 0, d, 8);
    run<4, false>("zeros ", 1, d, 8);
    run<4, false>("signed, spread exponents", 2, d, 8);
    run<4, true>("signed, spread exponents", 2, d, 8);
    return 0;
}
