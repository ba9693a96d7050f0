// Developer microbenchmark: the SUSTAINED fp32 MFMA ceiling of this chip (v_mfma_f32_32x32x2_f32, registers only, no memory
// traffic), with random and with zero operands, back to back for about a second each. The chip clocks to its power budget,
// so this — not 157.3 TFLOP/s at 2.4 GHz — is what a perfectly fed fp32 MFMA kernel can reach on the same data.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/bin/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void mfma_loop(float* out, int iters, float scale) {
    floatx16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // pseudo-random operands per lane (scale 0 -> all zeros)
    unsigned h = (threadIdx.x + 1) * 2654435761u ^ (blockIdx.x * 40503u);
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) {
        h = h * 1664525u + 1013904223u;
        a[i] = scale * ((float)(h >> 8) / 16777216.f - 0.5f);
        h = h * 1664525u + 1013904223u;
        b[i] = scale * ((float)(h >> 8) / 16777216.f - 0.5f);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[(i + j) & 7], acc[j], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[threadIdx.x] = s;
}
template <int WAVES>
static void run(const char* label, float scale, float* d, int waves_per_cu = 8) {
    const int iters = 20000;  // 32 MFMAs per iteration per wave
    const int blocks = 256 * (waves_per_cu / WAVES) * 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        const int launches = 8;
        for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(mfma_loop<WAVES>, dim3(blocks), dim3(WAVES * 64), 0, 0, d, iters, scale);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double fl = (double)launches * blocks * WAVES * iters * 32.0 * 4096.0;
        printf("%s waves/block %d (%d waves/CU): %.1f ms  %.1f TFLOP/s -> effective clock if 100 %% busy %.3f GHz\n", label, WAVES, waves_per_cu, ms, fl / ms / 1e9,
               fl / ms / 1e9 / 157.3 * 2.4);
    }
}
int main() {
    float* d;
    hipMalloc(&d, 4096);
    run<4>("random", 1.0f, d);
    run<4>("random", 1.0f, d, 4);   // ONE wave per SIMD
    run<4>("random", 1.0f, d, 12);  // three
    return 0;
}
