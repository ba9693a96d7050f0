// Developer check: cycles per MFMA of a DEPENDENT chain (one accumulator) and of 2 / 4 interleaved chains, for v_mfma_f32_16x16x4_f32 and
// v_mfma_f32_32x32x2_f32. Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_chain.hip -o tools/bin/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k16(float* out, unsigned long long* cyc, float a, float b) {
    floatx4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = floatx4{0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < 256; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC>
__global__ void k32(float* out, unsigned long long* cyc, float a, float b) {
    floatx16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < 256; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    float* o; unsigned long long* c; hipMalloc(&o, 1024); hipMalloc(&c, 8);
    unsigned long long h;
#define RUN(K, N)                                                                                         \
    K<N><<<1, 64>>>(o, c, 1.0f, 0.5f); hipDeviceSynchronize(); K<N><<<1, 64>>>(o, c, 1.0f, 0.5f);          \
    hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);                                                           \
    printf(#K " %d accumulator(s): %.1f cycles (s_memtime ticks) per MFMA\n", N, (double)h / (256.0 * 16 * N));
    RUN(k16, 1) RUN(k16, 2) RUN(k16, 4) RUN(k32, 1) RUN(k32, 2) RUN(k32, 4)
    return 0;
}
