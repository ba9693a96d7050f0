// Developer microbenchmark for conv_mfma_kernel (one instantiation, synthetic data, HIP-event timing). Not part of the
// product. Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DKT_=11 -DDIL_=1 -DCIN_=128 [-DVAR_...] tools/conv_micro.hip -o /tmp/conv_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#ifndef KT_
#define KT_ 11
#endif
#ifndef DIL_
#define DIL_ 1
#endif
#define VITS_MICRO_KT KT_
#define VITS_MICRO_DIL DIL_
#include "../vits.cpp_amd/csrc/conv_mfma.hip"
using namespace vits;
#ifndef KT_
#define KT_ 11
#endif
#ifndef DIL_
#define DIL_ 1
#endif
#ifndef CIN_
#define CIN_ 128
#endif
#ifndef T_
#define T_ 14400
#endif
#ifndef B_
#define B_ 64
#endif
int main() {
    const int C = CIN_, K = KT_, T = T_, B = B_;
    std::vector<float> w((size_t)C * C * K, 0.01f), bias(C, 0.1f);
    PackedConv pc; pc.cin = C; pc.cout = C; pc.kt = K; pc.epi = EPI_STD;
    auto packed = pack_conv_weights(w.data(), C, C, K, EPI_STD, 0, &pc.rows, &pc.mtiles_used, &pc.mtiles, &pc.nchunks);
    float *dw, *db, *dx, *dy;
    hipMalloc(&dw, packed.size() * 4); hipMemcpy(dw, packed.data(), packed.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&db, C * 4); hipMemcpy(db, bias.data(), C * 4, hipMemcpyHostToDevice);
    size_t n = (size_t)B * C * T;
    hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4);
    std::vector<float> hx(n); for (size_t i = 0; i < n; ++i) hx[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    pc.wp = dw; pc.bias = db;
    ConvCall c; c.x.p = dx; c.x.cs = T; c.x.bs = (int64_t)C * T; c.y.p = dy; c.y.cs = T; c.y.bs = (int64_t)C * T;
#ifdef WITH_RES
    c.res = c.x;
#endif
    c.batch = B; c.t_in = c.t_out = T; c.dil = DIL_; c.pad_l = (K - 1) * DIL_ / 2; c.pre_act = 1; c.slope = 0.1f;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) launch_conv(pc, c, nullptr);
    hipDeviceSynchronize();
    const int reps = 5;
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < reps; ++i) launch_conv(pc, c, nullptr);
    hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    double fl = 2.0 * C * C * K * (double)B * T;
    printf("C=%d k=%d d=%d T=%d B=%d: %.3f ms  %.1f TFLOP/s  (%s)\n", C, K, DIL_, T, B, ms, fl / ms / 1e9, hipGetErrorString(hipGetLastError()));
    return 0;
}
