// Developer check: is a K chain of v_mfma_f32_16x16x32_f16 bit-identical to the same chain of v_mfma_f32_32x32x16_f16 (operands with
// random signs and exponents, fp32 accumulate)? Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_bits16.hip -o tools/bin/mfma_bits16
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
constexpr int K = 512;
// 32x32x16: A lane l -> row l & 31, k = 8 * (l >> 5) + i (i = 0..7); B lane l -> col l & 31, same k; C reg r -> row (r & 3) + 8 (r >> 2) + 4 (l >> 5), col l & 31
__global__ void k32(const _Float16* A, const _Float16* B, float* C) {
    const int l = threadIdx.x;
    floatx16 acc = {};
    for (int k = 0; k < K; k += 16) {
        half8 a, b;
        for (int i = 0; i < 8; ++i) {
            a[i] = A[(l & 31) * K + k + 8 * (l >> 5) + i];
            b[i] = B[(k + 8 * (l >> 5) + i) * 32 + (l & 31)];
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}
// 16x16x32: A lane l -> row l & 15, k = 8 * (l >> 4) + i; B lane l -> col l & 15; C reg r -> row 4 (l >> 4) + r, col l & 15
__global__ void k16(const _Float16* A, const _Float16* B, float* C) {
    const int l = threadIdx.x;
    for (int tm = 0; tm < 2; ++tm)
        for (int tn = 0; tn < 2; ++tn) {
            floatx4 acc = {};
            for (int k = 0; k < K; k += 32) {
                half8 a, b;
                for (int i = 0; i < 8; ++i) {
                    a[i] = A[(tm * 16 + (l & 15)) * K + k + 8 * (l >> 4) + i];
                    b[i] = B[(k + 8 * (l >> 4) + i) * 32 + tn * 16 + (l & 15)];
                }
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
            }
            for (int r = 0; r < 4; ++r) C[(tm * 16 + 4 * (l >> 4) + r) * 32 + tn * 16 + (l & 15)] = acc[r];
        }
}
int main() {
    std::vector<_Float16> A(32 * K), B(K * 32);
    std::vector<float> C1(1024), C2(1024);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (_Float16)(((int)(s >> 8) % 2001 - 1000) / 791.0f * std::ldexp(1.0f, (int)((s >> 3) % 7) - 5)); };
    for (auto& v : A) v = rnd();
    for (auto& v : B) v = rnd();
    _Float16 *dA, *dB; float* dC;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, 4096);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    k32<<<1, 64>>>(dA, dB, dC); hipMemcpy(C1.data(), dC, 4096, hipMemcpyDeviceToHost);
    k16<<<1, 64>>>(dA, dB, dC); hipMemcpy(C2.data(), dC, 4096, hipMemcpyDeviceToHost);
    int d12 = 0; double maxrel = 0;
    for (int i = 0; i < 1024; ++i) {
        d12 += std::memcmp(&C1[i], &C2[i], 4) != 0;
        maxrel = std::max(maxrel, (double)std::fabs(C1[i] - C2[i]) / (std::fabs(C1[i]) + 1e-30));
    }
    printf("K = %d f16: 32x32x16 vs 16x16x32: %d of 1024 outputs differ (max relative difference %.2e) (%s)\n", K, d12, maxrel, hipGetErrorString(hipGetLastError()));
    return 0;
}
