// Developer microbenchmark for rbblock16_kernel (one whole ResBlock, 16-bit operands): synthetic data, HIP-event timing and, with
// -DVITS_PHASE_TIMING, per-block phase stamps (stream load + first tile | pair 0 | pair 1 | pair 2 | epilogue) and per-CU residency. Not part
// of the product. Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DKT_=7 -DC_=64 [-DVITS_PHASE_TIMING] tools/rbb_micro.hip -o tools/bin/rbb_micro
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>
#ifndef KT_
#define KT_ 7
#endif
#ifndef C_
#define C_ 64
#endif
#ifndef T_
#define T_ 28934  // frames x 128 of the C = 64 stage of the benchmark batch, per utterance
#endif
#ifndef B_
#define B_ 64
#endif
#include "../vits.cpp_amd/csrc/rbblock16.hip"
using namespace vits;

int main() {
    const int C = C_, K = KT_, T = T_, B = B_;
    const int ts = (T + 31) / 32 * 32;
    const size_t wn = (size_t)C * C * K;
    std::vector<uint16_t> w(wn);
    for (size_t i = 0; i < wn; ++i) w[i] = 0x2000 + (uint16_t)((i * 2654435761u) >> 20 & 0x3ff) + (uint16_t)((i & 1) << 15);
    std::vector<float> bias(C, 0.01f);
    uint16_t* dw;
    float *db, *dy0, *dyg;
    uint16_t* dy16;
    hipMalloc(&dw, wn * 2);
    hipMemcpy(dw, w.data(), wn * 2, hipMemcpyHostToDevice);
    hipMalloc(&db, C * 4);
    hipMemcpy(db, bias.data(), C * 4, hipMemcpyHostToDevice);
    const size_t n = (size_t)B * C * ts;
    hipMalloc(&dy0, n * 4);
    hipMalloc(&dyg, n * 4);
    hipMalloc(&dy16, n * 2);
    std::vector<float> hy(n);
    for (size_t i = 0; i < n; ++i) hy[i] = (float)((i * 2246822519u) >> 12 & 0xfff) / 4096.f - 0.5f;
    hipMemcpy(dy0, hy.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(dyg, 0, n * 4);
    PackedConv pc;
    pc.cin = pc.cout = C;
    pc.kt = K;
    pc.wp16 = dw;
    pc.bias = db;
    const PackedConv* c1[3] = {&pc, &pc, &pc};
    const PackedConv* c2[3] = {&pc, &pc, &pc};
    RbBlock16Call f;
    f.y0 = dy0;
    f.batch = B;
    f.tmax = T;
    f.yg = dyg;
    f.accg = dyg;  // (the second / third resblock of a stage: the sum so far is read in place)
    f.g_bs = (int64_t)C * ts;
    f.g_ts = ts;
    f.scale = 1.f;
#ifdef RB2_  // the LAST resblock of a stage: sum so far read, scaled by 1 / num_kernels, only the 16-bit copy written
    f.yg = nullptr;
    f.y16.p = dy16;
    f.y16.bs = (int64_t)C * ts;
    f.y16.ts = ts;
    f.y16_slope = 0.1f;
    f.scale = 3.f;
    f.scale_div = 1;
#endif
#ifdef RB0_  // the FIRST resblock of a stage: no sum to read
    f.accg = nullptr;
#endif
#ifdef RAGGED_  // lengths between 60 and 100 % of T (the benchmark batch's predicted durations spread like that)
    std::vector<int> hl(B);
    for (int i = 0; i < B; ++i) hl[i] = (int)((0.6 + 0.4 * ((i * 37) % B) / (double)(B - 1)) * T);
    int* dl;
    hipMalloc(&dl, B * 4);
    hipMemcpy(dl, hl.data(), B * 4, hipMemcpyHostToDevice);
    f.lens = dl;
#endif
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) launch_rbblock16(c1, c2, f, VITS_ARITH_F16, nullptr);
    hipDeviceSynchronize();
    const int reps = 5;
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < reps; ++i) launch_rbblock16(c1, c2, f, VITS_ARITH_F16, nullptr);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
#ifdef VITS_PHASE_TIMING
    {
        std::vector<unsigned long long> ph(16 * 65536);
        hipMemcpyFromSymbol(ph.data(), HIP_SYMBOL(vits_rbb_phase), ph.size() * 8);
        double d[5] = {0, 0, 0, 0, 0};
        size_t cnt = 0;
        unsigned long long tmin = ~0ull, tmax = 0;
        for (size_t i = 0; i < 65536; ++i) {
            const unsigned long long* q = &ph[16 * i];
            if (!q[0] || !q[5] || q[5] < q[0]) continue;
            for (int k = 0; k < 5; ++k) d[k] += (double)(q[k + 1] - q[k]);
            tmin = std::min(tmin, q[0]);
            tmax = std::max(tmax, q[5]);
            ++cnt;
        }
        printf("phases over %zu blocks, us: stream load + first tile %.2f | pair 0 %.2f | pair 1 %.2f | pair 2 %.2f | epilogue %.2f ; block life %.2f us; launch span %.1f us\n", cnt,
               d[0] / cnt / 100, d[1] / cnt / 100, d[2] / cnt / 100, d[3] / cnt / 100, d[4] / cnt / 100, (d[0] + d[1] + d[2] + d[3] + d[4]) / cnt / 100, (tmax - tmin) / 100.0);
    }
#endif
    const double fl = 3.0 * 2.0 * 2.0 * C * C * K * (double)B * T;
    printf("rbblock16 C=%d k=%d T=%d B=%d: %.3f ms  %.0f TFLOP/s (algorithmic, six convs)  (%s)\n", C, K, T, B, ms, fl / ms / 1e9, hipGetErrorString(hipGetLastError()));
    return 0;
}
