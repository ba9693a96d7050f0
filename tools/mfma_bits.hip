// Developer check: is a K chain of v_mfma_f32_16x16x4_f32 bit-identical to the same chain of v_mfma_f32_32x32x2_f32 (and to a sequential
// fmaf chain in k order)? Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_bits.hip -o tools/bin/mfma_bits
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int K = 256;
__global__ void k32(const float* A, const float* B, float* C) {  // A [32][K], B [K][32], C [32][32]
    const int l = threadIdx.x;
    floatx16 acc = {};
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l & 31) * K + k + (l >> 5)], B[(k + (l >> 5)) * 32 + (l & 31)], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}
__global__ void k16(const float* A, const float* B, float* C) {
    const int l = threadIdx.x;
    for (int tm = 0; tm < 2; ++tm)
        for (int tn = 0; tn < 2; ++tn) {
            floatx4 acc = {};
            for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(tm * 16 + (l & 15)) * K + k + (l >> 4)], B[(k + (l >> 4)) * 32 + tn * 16 + (l & 15)], acc, 0, 0, 0);
            for (int r = 0; r < 4; ++r) C[(tm * 16 + 4 * (l >> 4) + r) * 32 + tn * 16 + (l & 15)] = acc[r];
        }
}
int main() {
    std::vector<float> A(32 * K), B(K * 32), C1(1024), C2(1024), C0(1024);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 8) % 20001 - 10000) / 7919.0f * std::ldexp(1.0f, (int)((s >> 3) % 9) - 4); };
    for (auto& v : A) v = rnd();
    for (auto& v : B) v = rnd();
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            float a = 0.f;
            for (int k = 0; k < K; ++k) a = fmaf(A[i * K + k], B[k * 32 + j], a);
            C0[i * 32 + j] = a;
        }
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    k32<<<1, 64>>>(dA, dB, dC); hipMemcpy(C1.data(), dC, 4096, hipMemcpyDeviceToHost);
    k16<<<1, 64>>>(dA, dB, dC); hipMemcpy(C2.data(), dC, 4096, hipMemcpyDeviceToHost);
    int d12 = 0, d10 = 0, d20 = 0;
    for (int i = 0; i < 1024; ++i) {
        d12 += std::memcmp(&C1[i], &C2[i], 4) != 0; d10 += std::memcmp(&C1[i], &C0[i], 4) != 0; d20 += std::memcmp(&C2[i], &C0[i], 4) != 0;
    }
    printf("K = %d: 32x32x2 vs 16x16x4: %d of 1024 differ; 32x32x2 vs fmaf chain: %d; 16x16x4 vs fmaf chain: %d (%s)\n", K, d12, d10, d20, hipGetErrorString(hipGetLastError()));
    return 0;
}
