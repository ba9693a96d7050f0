// Developer microbenchmark for rel_attention_mfma_kernel (windowed relative-position attention of the text encoder): synthetic data, HIP-event
// timing and, with -DVITS_PHASE_TIMING, per-block phase stamps (prologue | scores | softmax | P V | relative values + store). Not part of
// the product. Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DT_=2049 -DB_=8 [-DVITS_PHASE_TIMING] tools/att_micro.hip -o tools/bin/att_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#ifndef T_
#define T_ 2049
#endif
#ifndef B_
#define B_ 8
#endif
#include "../vits.cpp_amd/csrc/misc_kernels.hip"
using namespace vits;

int main() {
    const int T = T_, B = B_, H = 2, HD = 96, C = H * HD, W = 4;
    const int ts = (T + 3) / 4 * 4;
    const size_t n = (size_t)B * C * ts;
    std::vector<float> h(n);
    float *q, *k, *v, *o, *rk, *rv;
    for (float** p : {&q, &k, &v, &o}) hipMalloc(p, n * 4);
    for (int which = 0; which < 3; ++which) {
        for (size_t i = 0; i < n; ++i) h[i] = (float)(((i + which * 977) * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
        hipMemcpy(which == 0 ? q : which == 1 ? k : v, h.data(), n * 4, hipMemcpyHostToDevice);
    }
    std::vector<float> r((2 * W + 1) * HD, 0.05f);
    hipMalloc(&rk, r.size() * 4);
    hipMalloc(&rv, r.size() * 4);
    hipMemcpy(rk, r.data(), r.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(rv, r.data(), r.size() * 4, hipMemcpyHostToDevice);
    TensorRef tq{q, (int64_t)C * ts, ts}, tk{k, (int64_t)C * ts, ts}, tv{v, (int64_t)C * ts, ts}, to{o, (int64_t)C * ts, ts};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 40; ++i) launch_rel_attention(tq, tk, tv, rk, rv, to, nullptr, B, H, HD, T, W, 0.102f, nullptr);
    hipDeviceSynchronize();
    const int reps = 40;
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < reps; ++i) launch_rel_attention(tq, tk, tv, rk, rv, to, nullptr, B, H, HD, T, W, 0.102f, nullptr);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
#ifdef VITS_PHASE_TIMING
    {
        std::vector<unsigned long long> ph(8 * 65536);
        hipMemcpyFromSymbol(ph.data(), HIP_SYMBOL(vits_att_phase), ph.size() * 8);
        double d[5] = {0, 0, 0, 0, 0};
        size_t cnt = 0;
        unsigned long long tmin = ~0ull, tmax = 0;
        for (size_t i = 0; i < 65536; ++i) {
            const unsigned long long* p = &ph[8 * i];
            if (!p[0] || !p[5] || p[5] < p[0]) continue;
            for (int j = 0; j < 5; ++j) d[j] += (double)(p[j + 1] - p[j]);
            tmin = p[0] < tmin ? p[0] : tmin;
            tmax = p[5] > tmax ? p[5] : tmax;
            ++cnt;
        }
        printf("phases over %zu blocks, us: prologue %.2f | scores %.2f | softmax %.2f | P V %.2f | rel values + store %.2f ; block life %.2f; launch span %.1f us\n", cnt,
               d[0] / cnt / 100, d[1] / cnt / 100, d[2] / cnt / 100, d[3] / cnt / 100, d[4] / cnt / 100, (d[0] + d[1] + d[2] + d[3] + d[4]) / cnt / 100, (tmax - tmin) / 100.0);
    }
#endif
    const double fl = 2.0 * 2.0 * (double)T * T * HD * H * B;
    printf("attention T=%d B=%d heads=%d head_dim=%d: %.3f ms  %.1f TFLOP/s (QK^T + PV)  (%s)\n", T, B, H, HD, ms, fl / ms / 1e9, hipGetErrorString(hipGetLastError()));
    return 0;
}
